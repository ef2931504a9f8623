"""Restart intervals as the parallel unit inside one scan (charls_amd/csrc/device/restart_intervals.hip).  GPU only.

Decode side: streams with DRI/RSTm (the reference's fixtures and our own) are decoded interval-parallel; results and
error codes must equal the sequential exact decoder's and the oracle's.
Encode side (extension, the reference cannot emit restart markers): every interval must be exactly the reference's
coding of those rows as an image of their own, joined by FF D0+m; the oracle -- and the reference itself when
oracle/_ref travelled -- must decode the result."""
import os

import numpy as np
import pytest

import common
import jls_container
import oracle_bind as ob
from charls_amd import capi, synth
from charls_amd.capi import JpegLSError

pytestmark = pytest.mark.gpu

REF_LIB = os.path.join(common.ROOT, "oracle", "_ref", "libcharls_ref.so")


@pytest.fixture(scope="module")
def lib():
    L = capi.load_product()
    assert L.lib.charls_amd_device_status() == 0
    return L


def _split_at_restart_markers(data: bytes):
    """Entropy-coded segment -> list of interval byte strings and the marker codes between them."""
    pieces, codes, start, i = [], [], 0, 0
    while i + 1 < len(data):
        if data[i] == 0xFF and 0xD0 <= data[i + 1] <= 0xD7:
            pieces.append(data[start:i])
            codes.append(data[i + 1])
            start = i = i + 2
        else:
            i += 1
    pieces.append(data[start:])
    return pieces, codes


CONFIGS = [  # name, bits, comps, ilv, near, (w, h), intervals to try
    ("gray8", 8, 1, 0, 0, (200, 150), (1, 7, 64, 149, 150, 1000)),
    ("gray16", 16, 1, 0, 0, (129, 70), (5, 32)),
    ("gray12_near2", 12, 1, 0, 2, (96, 64), (9,)),
    ("rgb8_none", 8, 3, 0, 0, (64, 50), (8,)),
    ("rgb8_line", 8, 3, 1, 0, (64, 50), (8, 49)),
    ("rgb8_sample", 8, 3, 2, 0, (64, 50), (16,)),
    ("rgb8_sample_near3", 8, 3, 2, 3, (40, 33), (4,)),
]


def _image(bits, comps, ilv, w, h, seed):
    planes = [synth.frame_numpy(w, h, seed=seed + c, bits=bits, kind="mixed") for c in range(comps)]
    if comps == 1:
        return planes[0]
    return np.stack(planes, axis=0) if ilv == 0 else np.stack(planes, axis=-1)


@pytest.mark.parametrize("name,bits,comps,ilv,near,size,intervals", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_restart_encode_is_the_reference_coding_of_every_interval(lib, name, bits, comps, ilv, near, size, intervals):
    w, h = size
    img = _image(bits, comps, ilv, w, h, seed=len(name))
    kw = dict(bits_per_sample=bits, component_count=comps, interleave_mode=ilv, near_lossless=near)
    for ri in intervals:
        jls = lib.encode(img, restart_interval=ri, **kw)
        cont = jls_container.parse(jls)
        assert cont.restart_interval == ri
        n_int = (h + ri - 1) // ri if ri < h else 1
        assert len(cont.scans) == (comps if ilv == 0 else 1)
        for k, scan in enumerate(cont.scans):
            pieces, codes = _split_at_restart_markers(jls[scan.data_start:scan.data_end])
            assert len(pieces) == n_int
            assert codes == [0xD0 + (j & 7) for j in range(n_int - 1)]
            for j, piece in enumerate(pieces):
                first = j * ri
                count = min(ri, h - first) if ri < h else h
                if comps == 1:
                    sub, sub_kw = img[first:first + count], kw
                elif ilv == 0:  # one single-component scan per plane
                    sub, sub_kw = img[k, first:first + count], dict(kw, component_count=1)
                else:
                    sub, sub_kw = img[first:first + count], kw
                want = ob.encode(np.ascontiguousarray(sub), width=w, height=count, **sub_kw)
                wc = jls_container.parse(want)
                assert piece == want[wc.scans[0].data_start:wc.scans[0].data_end], (ri, k, j)
        # the CPU decoders read it
        got = ob.decode(jls)[1]
        pix = got.view(img.dtype).reshape(img.shape)
        if near == 0:
            assert np.array_equal(pix, img)
        else:
            assert np.abs(pix.astype(np.int64) - img.astype(np.int64)).max() <= near
        if os.path.exists(REF_LIB):
            ref = capi.CharLSLibrary(REF_LIB)
            assert ref.decode(jls)[1].tobytes() == got.tobytes()
        # and so does the product (interval-parallel)
        assert lib.decode(jls)[1].tobytes() == got.tobytes()


def _assemble_dri_stream(img, ri):
    """A DRI stream put together on the CPU the way any encoder would: the oracle's coding of every interval as an image
    of its own, RSTm between them.  Returns the stream and how many intervals end with 0xFF + a stuffed 7-bit byte."""
    h, w = img.shape
    full = ob.encode(img, width=w, height=h)
    scan = jls_container.parse(full).scans[0]
    sos = full.rfind(b"\xff\xda", 0, scan.data_start)
    body, ff_endings, n = b"", 0, (h + ri - 1) // ri
    for j in range(n):
        sub = np.ascontiguousarray(img[j * ri:(j + 1) * ri])
        s = ob.encode(sub, width=w, height=sub.shape[0])
        sc = jls_container.parse(s).scans[0]
        piece = s[sc.data_start:sc.data_end]
        ff_endings += len(piece) >= 2 and piece[-2] == 0xFF
        body += piece + (bytes([0xFF, 0xD0 + (j & 7)]) if j + 1 < n else b"")
    return full[:sos] + b"\xff\xdd\x00\x04" + ri.to_bytes(2, "big") + full[sos:scan.data_start] + body + b"\xff\xd9", ff_endings


def test_cpu_assembled_restart_streams_including_ff_terminated_intervals(lib):
    rng = np.random.default_rng(1)
    ff_total = 0
    for trial in range(120):
        w, h, ri = int(rng.integers(8, 80)), int(rng.integers(4, 40)), int(rng.integers(1, 4))
        img = (rng.integers(0, 256, size=(h, w), dtype=np.uint8) if trial % 2 else
               (rng.integers(0, 8, size=(h, w)) * 31).astype(np.uint8))
        jls, ff = _assemble_dri_stream(img, ri)
        ff_total += ff
        assert ob.decode(jls)[1].tobytes() == img.tobytes()
        assert lib.decode(jls)[1].tobytes() == img.tobytes(), (trial, w, h, ri)
        # and our encoder writes exactly this file (short intervals of noise do not compress: roomy destination)
        assert lib.encode(img, restart_interval=ri, destination_size=len(jls) + 64) == jls, (trial, w, h, ri)
    assert ff_total > 0  # the sample contains intervals whose last data byte is 0xFF (followed by a stuffed byte)


def test_sequential_and_interval_decoders_agree(lib, knobs):
    img = synth.frame_numpy(320, 240, seed=77, kind="mixed")
    jls = lib.encode(img, restart_interval=16)
    fast = lib.decode(jls)[1].tobytes()
    knobs.set("SEQUENTIAL_INTERVALS", 1)
    slow = lib.decode(jls)[1].tobytes()
    assert fast == slow == img.tobytes()


@pytest.mark.parametrize("damage", ["wrong_index", "missing_marker", "extra_byte", "truncated"])
def test_damaged_restart_streams_report_the_sequential_decoders_errc(lib, damage, knobs):
    img = synth.frame_numpy(96, 64, seed=5, kind="mixed")
    jls = bytearray(lib.encode(img, restart_interval=8))
    cont = jls_container.parse(bytes(jls))
    scan = cont.scans[0]
    marks = [i for i in range(scan.data_start, scan.data_end - 1) if jls[i] == 0xFF and 0xD0 <= jls[i + 1] <= 0xD7]
    assert len(marks) == 7
    if damage == "wrong_index":
        jls[marks[3] + 1] = 0xD6
    elif damage == "missing_marker":
        del jls[marks[2]:marks[2] + 2]
    elif damage == "extra_byte":
        jls.insert(marks[4], 0x00)
    else:
        del jls[marks[5] + 1:]
    try:
        want = ("ok", ob.decode(bytes(jls))[1].tobytes())
    except ob.OracleError as e:
        want = ("err", e.errc)

    def run():
        try:
            return ("ok", lib.decode(bytes(jls))[1].tobytes())
        except JpegLSError as e:
            return ("err", e.errc)
    got = run()
    knobs.set("SEQUENTIAL_INTERVALS", 1)
    seq = run()
    assert got == seq == want


def test_batch_api_with_restart_intervals(lib):
    import torch
    from charls_amd import batch
    frames = synth.frames_torch(5, 256, 200, seed0=40, bits=8, device="cuda:0")
    enc = batch.encode_batch(frames, restart_interval=32, lib=lib)
    assert (enc.errcs == 0).all()
    for f in range(frames.shape[0]):
        one = lib.encode(frames[f].cpu().numpy(), restart_interval=32)
        assert enc.streams[f, :int(enc.sizes[f])].cpu().numpy().tobytes() == one
    out = torch.empty_like(frames)
    params, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out, lib=lib)
    assert (errcs == 0).all() and torch.equal(out, frames)
    assert params.restart_interval == 32


def test_destination_too_small_with_restart_intervals(lib):
    img = synth.frame_numpy(128, 96, seed=9, kind="noise")  # every interval re-learns its statistics: larger than one scan
    jls = lib.encode(img, restart_interval=8, destination_size=40000)
    assert len(jls) > len(lib.encode(img))
    assert lib.encode(img, restart_interval=8, destination_size=len(jls)) == jls
    with pytest.raises(JpegLSError) as e:
        lib.encode(img, restart_interval=8, destination_size=len(jls) - 1)
    assert e.value.errc == 3


@pytest.mark.slow
def test_single_frame_latency_with_restart_intervals(lib, capsys):
    """One 4096x4096 frame through the host-pointer ABI: the restart extension turns one chain into 64."""
    import time
    img = synth.frame_numpy(4096, 4096, seed=2, bits=8)
    t = {}
    for ri in (0, 64):
        a = time.perf_counter()
        jls = lib.encode(img, restart_interval=ri)
        b = time.perf_counter()
        px = lib.decode(jls)[1]
        c = time.perf_counter()
        assert px.tobytes() == img.tobytes()
        t[ri] = (b - a, c - b, len(jls))
    with capsys.disabled():
        print(f"\n[restart] 4096x4096 single frame, host-pointer ABI: no DRI encode {t[0][0]:.3f}s decode {t[0][1]:.3f}s "
              f"({t[0][2]} B); DRI=64 encode {t[64][0]:.3f}s decode {t[64][1]:.3f}s ({t[64][2]} B)")
    assert t[64][1] < t[0][1] / 4


@pytest.mark.parametrize("interval", [8, 16])
def test_batch_restart_planar_rgb_with_a_slot_that_is_too_small(lib, interval):
    """A frame that fails in its first plane (destination_too_small) stays a per-frame error in the later planes of an
    ILV_NONE batch with restart intervals: its emptied descriptor must produce empty intervals, not an underflowed row
    count (round-1 advisor finding); the other frames get the bytes of a batch without the bad frame."""
    import torch
    from charls_amd import batch
    w, h, n = 64, 40, 4
    imgs = [synth.frame_numpy(w, h, seed=70 + f, components=3, kind="noise" if f == 1 else "zero", interleaved=False)
            for f in range(n)]
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    roomy = torch.zeros((n, 65536), dtype=torch.uint8, device="cuda:0")  # noise re-learnt per interval outgrows the estimate
    good = batch.encode_batch(frames, component_count=3, interleave_mode=0, restart_interval=interval, streams=roomy, lib=lib)
    assert (good.errcs == 0).all()
    # every slot is too small for the noise frame (frame 1) but large enough for the all-zero frames
    pitch = (int(max(good.sizes[f] for f in (0, 2, 3))) + 255) & ~255
    assert pitch < int(good.sizes[1])
    streams = torch.zeros((n, pitch), dtype=torch.uint8, device="cuda:0")
    enc = batch.encode_batch(frames, component_count=3, interleave_mode=0, restart_interval=interval, streams=streams, lib=lib)
    assert list(enc.errcs) == [0, 3, 0, 0] and int(enc.sizes[1]) == 0
    for f in (0, 2, 3):
        assert enc.streams[f, :int(enc.sizes[f])].cpu().numpy().tobytes() == \
            good.streams[f, :int(good.sizes[f])].cpu().numpy().tobytes()
    out = torch.zeros_like(frames)
    sizes = enc.sizes.copy()
    sizes[1] = 0
    _, errcs, _ = batch.decode_batch(enc.streams, sizes, out, lib=lib)
    assert [int(e) == 0 for e in errcs] == [True, False, True, True]
    for f in (0, 2, 3):
        assert torch.equal(out[f], frames[f])
