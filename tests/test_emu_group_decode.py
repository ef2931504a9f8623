"""Kernel-logic check on the CPU of scan_group_decode.hip (several scans per wavefront): the unmodified kernel source
compiled for the host (tests/emu) against golden vectors, the oracle, corrupt and mutated streams.  Test infrastructure:
it proves the kernel's logic, the product path is covered by tests/test_gpu_*.py on the GPU box."""
import ctypes as C

import numpy as np
import pytest

import common
import emu_bind
import jls_container
import oracle_bind as ob
from charls_amd import synth
from test_emu_serial_kernels import FAST_CASES, _stream_copy

GROUPS = [4, 8, 16, 32]  # lanes per scan: 16, 8, 4, 2 scans per wavefront


def _rotating(groups, cases):
    """Every case with ONE lanes-per-scan setting, the settings taken in turn (all of them occur over the list; running every
    case with every setting made this file half of the CPU suite)."""
    return [(groups[i % len(groups)],) + tuple(c) for i, c in enumerate(cases)]


def _group_eligible(bits, pc):
    return (pc[4] & 0xFF) != 0 and (bits <= 8 or pc[3] <= 1023)


def _launch(L, descs, group):
    n = len(descs)
    arr = (emu_bind.ScanDesc * n)(*descs)
    res = (emu_bind.ScanResult * n)()
    assert L.emu_decode_scans_group(arr, res, n, group) == 0
    return res


@pytest.mark.parametrize("c", FAST_CASES, ids=lambda c: c["name"])
def test_group_decoder_matches_reference_pixels(c):
    """All scans of a golden case in ONE launch (they share the geometry), lanes per scan varied over the cases."""
    L = emu_bind.lib()
    with open(f"{common.GOLDEN}/{c['file']}", "rb") as f:
        jls = f.read()
    cont = jls_container.parse(jls)
    pc = jls_container.validated_pc(cont.pc, cont.bits, 0)
    if not _group_eligible(cont.bits, pc):
        pytest.skip("scan_fast_decode.hip / the exact decoder take this one")
    bps = 1 if cont.bits <= 8 else 2
    w, h = cont.width, cont.height
    keep, outs, descs = [], [], []
    for scan in cont.scans:
        pix = np.zeros(w * bps * h, dtype=np.uint8)
        outs.append(pix)
        descs.append(emu_bind.make_desc(w, h, 1, 0, cont.bits, 0, 0, pc, 0, pix, w * bps, _stream_copy(jls, scan.data_start), keep))
    group = GROUPS[len(c["name"]) % len(GROUPS)]
    res = _launch(L, descs, group)
    for r, scan in zip(res, cont.scans):
        assert (r.errc, r.flags) == (0, 0), "a valid stream must not need the exact decoder"
        assert r.bytes == scan.data_end - scan.data_start
    assert common.sha(b"".join(o.tobytes() for o in outs)) == c["decoded_sha256"]


@pytest.mark.parametrize("group,waves,w,h,bits,kind,count", [(16, 4, 70, 9, 8, "mixed", 37), (32, 4, 130, 6, 8, "hard", 5), (32, 8, 64, 7, 16, "mixed", 19),
                                                              (16, 8, 33, 5, 12, "noise", 70), (16, 4, 40, 4, 8, "zero", 1)])
def test_group_decoder_with_several_wavefronts_per_workgroup(group, waves, w, h, bits, kind, count):
    """decode_scans_group<S, G, 1, W>: W wavefronts per workgroup share the gradient table and nothing else (the launches of
    big batches: one workgroup per CU, a wavefront per SIMD).  Counts that leave whole wavefronts of the last workgroup
    without a scan, and a single scan."""
    L = emu_bind.lib()
    bps = 1 if bits <= 8 else 2
    keep, outs, descs, imgs = [], [], [], []
    for f in range(count):
        img = synth.frame_numpy(w, h, seed=7 * f + bits + waves, bits=bits, kind=kind)
        jls = ob.encode(img, width=w, height=h, bits_per_sample=bits)
        cont = jls_container.parse(jls)
        pc = jls_container.validated_pc(cont.pc, cont.bits, 0)
        pix = np.zeros(w * h * bps, dtype=np.uint8)
        descs.append(emu_bind.make_desc(w, h, 1, 0, bits, 0, 0, pc, 0, pix, w * bps, _stream_copy(jls, cont.scans[0].data_start), keep))
        outs.append(pix)
        imgs.append((img, cont.scans[0].data_end - cont.scans[0].data_start))
    arr = (emu_bind.ScanDesc * count)(*descs)
    res = (emu_bind.ScanResult * count)()
    assert L.emu_decode_scans_group_waves(arr, res, count, group, waves) == 0
    for r, o, (img, nbytes) in zip(res, outs, imgs):
        assert (r.errc, r.flags, r.bytes) == (0, 0, nbytes)
        assert o.tobytes() == img.tobytes()


@pytest.mark.parametrize("group,w,h,bits,kind,count", _rotating(GROUPS, [(64, 20, 8, "mixed", 7), (300, 5, 8, "noise", 5), (33, 9, 16, "mixed", 5),
                                                 (41, 7, 12, "hard", 3), (1, 9, 8, "mixed", 3), (520, 3, 2, "noise", 9),
                                                 (70, 6, 8, "zero", 4)]))
def test_group_decoder_batches_of_different_frames(group, w, h, bits, kind, count):
    """`count` different frames of one geometry per launch (count is not a multiple of the scans per wavefront, so the last
    wavefront has idle lane groups); runs, escapes, several refills of the bit ring and lines of a few samples."""
    L = emu_bind.lib()
    bps = 1 if bits <= 8 else 2
    keep, outs, descs, imgs, ends = [], [], [], [], []
    for f in range(count):
        img = synth.frame_numpy(w, h, seed=11 * f + bits, bits=bits, kind=kind)
        jls = ob.encode(img, width=w, height=h, bits_per_sample=bits)
        cont = jls_container.parse(jls)
        pc = jls_container.validated_pc(cont.pc, cont.bits, 0)
        pix = np.zeros(w * h * bps, dtype=np.uint8)
        scan = cont.scans[0]
        descs.append(emu_bind.make_desc(w, h, 1, 0, bits, 0, 0, pc, 0, pix, w * bps, _stream_copy(jls, scan.data_start), keep))
        outs.append(pix)
        imgs.append(img)
        ends.append(scan.data_end - scan.data_start)
    res = _launch(L, descs, group)
    for f in range(count):
        assert (res[f].errc, res[f].flags, res[f].bytes) == (0, 0, ends[f]), f
        assert outs[f].tobytes() == imgs[f].tobytes(), f


@pytest.mark.parametrize("group,w,h,bits,comps,xform,kind,count", _rotating([4, 8, 16, 32], [(40, 12, 8, 3, 1, "mixed", 5), (300, 4, 8, 3, 2, "noise", 3), (33, 7, 16, 3, 3, "mixed", 3),
                          (1, 5, 8, 2, 0, "mixed", 2), (64, 6, 8, 3, 0, "zero", 3), (90, 8, 8, 4, 0, "hard", 3),
                          (130, 6, 5, 3, 0, "mixed", 3), (57, 9, 12, 2, 0, "gradient", 4)]))
def test_group_decoder_line_interleaved_batches(group, w, h, bits, comps, xform, kind, count):
    """Line-interleaved scans (one line per component in LDS, one RUNindex per component, the ONE set of contexts): `count`
    different frames per launch decode to the source pixels."""
    L = emu_bind.lib()
    bps = 1 if bits <= 8 else 2
    keep, descs, outs, wants, ends = [], [], [], [], []
    for f in range(count):
        img = synth.frame_numpy(w, h, seed=13 * f + bits + comps, bits=bits, components=comps, kind=kind, interleaved=True)
        jls = ob.encode(img, width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=1,
                        color_transformation=xform)
        cont = jls_container.parse(jls)
        pc = jls_container.validated_pc(cont.pc, cont.bits, 0)
        pix = np.zeros(w * h * comps * bps, dtype=np.uint8)
        descs.append(emu_bind.make_desc(w, h, comps, 1, bits, 0, xform, pc, 0, pix, w * comps * bps,
                                        _stream_copy(jls, cont.scans[0].data_start), keep))
        outs.append(pix)
        wants.append(img.tobytes())
        ends.append(cont.scans[0].data_end - cont.scans[0].data_start)
    res = _launch(L, descs, group)
    for f in range(count):
        assert (res[f].errc, res[f].flags, res[f].bytes) == (0, 0, ends[f]), f
        assert outs[f].tobytes() == wants[f], f


def test_group_decoder_leaves_other_thresholds_to_the_exact_decoder():
    """The gradient table is shared by the scans of a wavefront and built from the first one's thresholds: a scan with
    other thresholds reports kFastRetry, its neighbours decode."""
    L = emu_bind.lib()
    w, h = 40, 12
    keep, outs, descs, imgs = [], [], [], []
    presets = [None, None, (255, 9, 20, 60, 64), None, None]
    for f, preset in enumerate(presets):
        img = synth.frame_numpy(w, h, seed=f + 3, bits=8, kind="mixed")
        jls = ob.encode(img, width=w, height=h, preset=preset)
        cont = jls_container.parse(jls)
        pc = jls_container.validated_pc(cont.pc, 8, 0)
        pix = np.zeros(w * h, dtype=np.uint8)
        descs.append(emu_bind.make_desc(w, h, 1, 0, 8, 0, 0, pc, 0, pix, w, _stream_copy(jls, cont.scans[0].data_start), keep))
        outs.append(pix)
        imgs.append(img)
    res = _launch(L, descs, 8)  # eight scans per wavefront: all five share one
    for f in range(len(presets)):
        if presets[f] is None:
            assert (res[f].errc, res[f].flags) == (0, 0) and outs[f].tobytes() == imgs[f].tobytes(), f
        else:
            assert (res[f].errc, res[f].flags) == (0, 4), f


def test_group_decoder_wide_thresholds_first_in_the_wavefront_do_not_overrun_the_table():
    """the host checks the thresholds of ONE scan of a launch; a 16-bit scan whose T3 is beyond the shared
    gradient table (2 x 1023 + 2 bytes) and that happens to lead its wavefront must neither fill the table past its
    region (the context records of the first scan follow it) nor decode on it: every scan of the wavefront reports
    kFastRetry or decodes correctly, and nothing is corrupted."""
    L = emu_bind.lib()
    w, h = 40, 10
    keep, outs, descs, imgs = [], [], [], []
    presets = [(65535, 30, 900, 5000, 64), None, None]
    for f, preset in enumerate(presets):
        img = synth.frame_numpy(w, h, seed=f + 13, bits=16, kind="mixed")
        jls = ob.encode(img, width=w, height=h, bits_per_sample=16, preset=preset)
        cont = jls_container.parse(jls)
        pc = jls_container.validated_pc(cont.pc, 16, 0)
        pix = np.zeros(w * h * 2, dtype=np.uint8)
        descs.append(emu_bind.make_desc(w, h, 1, 0, 16, 0, 0, pc, 0, pix, w * 2, _stream_copy(jls, cont.scans[0].data_start), keep))
        outs.append(pix)
        imgs.append(img)
    res = _launch(L, descs, 16)  # four scans per wavefront: the wide-threshold scan leads
    for f in range(len(presets)):
        assert res[f].errc == 0 and res[f].flags in (0, 4), f
        if res[f].flags == 0:
            assert outs[f].tobytes() == imgs[f].tobytes(), f
    assert res[0].flags == 4


@pytest.mark.parametrize("name", ["fuzzy-input-bad-run-mode-golomb-code.jls", "fuzzy_input_golomb_16.jls",
                                  "fuzzy-input-no-valid-bits-at-the-end.jls", "no_start_byte_after_encoded_scan.jls"])
def test_group_decoder_defers_on_corrupt_streams(name):
    """The speed path never reports an error itself: anything unusual is handed to the exact decoder."""
    L = emu_bind.lib()
    jls = common.refdata(name)
    cont = jls_container.parse(jls)
    scan = cont.scans[0]
    if scan.near != 0 or scan.ilv != 0:
        pytest.skip("not a scan the speed path takes")
    pc = jls_container.validated_pc(cont.pc, cont.bits, 0)
    bps = 1 if cont.bits <= 8 else 2
    keep = []
    pix = np.zeros(cont.width * bps * cont.height, dtype=np.uint8)
    d = emu_bind.make_desc(cont.width, cont.height, 1, 0, cont.bits, 0, 0, pc, 0, pix, cont.width * bps,
                           _stream_copy(jls, scan.data_start), keep)
    res = _launch(L, [d], 16)
    assert (res[0].errc, res[0].flags) == (0, 4)


@pytest.mark.parametrize("bits,kind", [(8, "mixed"), (8, "zero"), (16, "mixed"), (12, "hard")])
def test_group_dispatch_on_mutated_scan_data_matches_the_oracle(bits, kind):
    """What runtime.hip's launch_decode_plain does (group kernel, then ONE launch of the exact wave decoder for the scans
    that reported kFastRetry) on a batch of mutated streams; every frame ends with the oracle's pixels or error code."""
    L = emu_bind.lib()
    w, h = 48, 12
    img = synth.frame_numpy(w, h, seed=bits, bits=bits, kind=kind)
    base = ob.encode(img, width=w, height=h, bits_per_sample=bits)
    cont = jls_container.parse(base)
    scan = cont.scans[0]
    pc = jls_container.validated_pc(cont.pc, cont.bits, 0)
    bps = 1 if bits <= 8 else 2
    rng = np.random.default_rng(bits * 13 + len(kind))
    keep, descs, outs, wants = [], [], [], []
    for k in range(36):
        b = bytearray(base)
        how = int(rng.integers(0, 5))
        i = int(rng.integers(scan.data_start, len(b) - 2))
        if how == 0:
            b[i] ^= 1 << int(rng.integers(0, 8))
        elif how == 1:
            b[i] = int(rng.choice([0x00, 0xFF, 0x7F, 0x80]))
        elif how == 2:
            del b[i:i + int(rng.integers(1, 4))]
        elif how == 3:
            b[i:i] = bytes([int(rng.choice([0x00, 0xFF, 0x55]))])
        data = bytes(b)  # how == 4: untouched
        try:
            wants.append((0, ob.decode(data)[1].tobytes()))
        except ob.OracleError as e:
            wants.append((e.errc, None))
        pix = np.zeros(w * h * bps, dtype=np.uint8)
        outs.append(pix)
        descs.append(emu_bind.make_desc(w, h, 1, 0, bits, 0, 0, pc, 0, pix, w * bps, _stream_copy(data, scan.data_start), keep))
    res = _launch(L, descs, 16)
    retry = [k for k in range(len(descs)) if res[k].flags & 4]
    for k in retry:  # the product gathers them into one launch; the emulated exact decoder takes one geometry per call too
        one = (emu_bind.ScanResult * 1)()
        L.emu_decode_scans_wave((emu_bind.ScanDesc * 1)(descs[k]), one, 1)
        res[k].errc, res[k].flags, res[k].bytes = one[0].errc, one[0].flags, one[0].bytes
    for k, want in enumerate(wants):
        if want[0] == 0:
            assert res[k].errc == 0 and outs[k].tobytes() == want[1], k
        else:
            assert res[k].errc == want[0], (k, res[k].errc, want[0])


@pytest.mark.parametrize("chunk", range(2))
def test_group_decoder_random_parameters(chunk):
    """Random lossless single-component parameter sets (bits 2..16, custom thresholds / RESET, odd sizes)."""
    from test_oracle_vs_reference import _image
    L = emu_bind.lib()
    rng = np.random.default_rng(900 + chunk)
    done = 0
    for it in range(40):
        bits = int(rng.integers(2, 17))
        w, h = int(rng.choice([1, 2, 5, 17, 64, 65, 130])), int(rng.choice([1, 2, 3, 8, 21]))
        maxval = (1 << bits) - 1
        kind = str(rng.choice(["rand", "smooth", "gradient", "mixed", "zero", "hard"]))
        preset = None
        if rng.random() < 0.4:
            t1 = int(rng.integers(1, maxval + 1))
            t2 = int(rng.integers(t1, maxval + 1))
            t3 = int(rng.integers(t2, maxval + 1))
            preset = (0, t1, t2, t3, int(rng.integers(3, max(255, maxval) + 1)))
        pc = jls_container.validated_pc(preset or (0,) * 5, bits, 0)
        if not _group_eligible(bits, pc):
            continue
        count = int(rng.integers(1, 6))
        bps = 1 if bits <= 8 else 2
        keep, descs, outs, imgs = [], [], [], []
        for f in range(count):
            img = _image(rng, w, h, bits, 1, 0, kind, it + f)
            jls = ob.encode(img, width=w, height=h, bits_per_sample=bits, preset=preset)
            cont = jls_container.parse(jls)
            pix = np.zeros(w * h * bps, dtype=np.uint8)
            descs.append(emu_bind.make_desc(w, h, 1, 0, bits, 0, 0, pc, 0, pix, w * bps,
                                            _stream_copy(jls, cont.scans[0].data_start), keep))
            outs.append(pix)
            imgs.append(np.ascontiguousarray(img))
        group = int(rng.choice(GROUPS))
        res = _launch(L, descs, group)
        for f in range(count):
            tag = (chunk, it, f, bits, w, h, kind, preset, group)
            assert (res[f].errc, res[f].flags) == (0, 0) and outs[f].tobytes() == imgs[f].tobytes(), tag
        done += 1
    assert done >= 15


# ---- decode_scans_group<.., kNear = true>: near-lossless single-component and line-interleaved scans (round 6) ----------
NEAR_GROUPS = [8, 16, 32]


def _near_descs(frames, w, h, bits, near, comps=1, ilv=0, preset=None):
    """(descs, outs, wants, ends, keep) of one launch: the oracle's stream of every frame and what the oracle decodes from it."""
    bps = 1 if bits <= 8 else 2
    keep, descs, outs, wants, ends = [], [], [], [], []
    for img in frames:
        jls = ob.encode(img, width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=ilv, near_lossless=near,
                        preset=preset)
        cont = jls_container.parse(jls)
        pc = jls_container.validated_pc(cont.pc, cont.bits, near)
        pix = np.zeros(w * h * comps * bps, dtype=np.uint8)
        scan = cont.scans[0]
        descs.append(emu_bind.make_desc(w, h, comps, ilv, bits, near, 0, pc, 0, pix, w * comps * bps, _stream_copy(jls, scan.data_start), keep))
        outs.append(pix)
        wants.append(ob.decode(jls)[1].tobytes())
        ends.append(scan.data_end - scan.data_start)
    return descs, outs, wants, ends, keep


@pytest.mark.parametrize("group,w,h,bits,near,kind,count", _rotating(NEAR_GROUPS, [
    (64, 20, 8, 2, "mixed", 7), (300, 5, 8, 1, "noise", 5), (33, 9, 16, 3, "mixed", 5), (41, 7, 12, 7, "hard", 3), (1, 9, 8, 2, "mixed", 3),
    (520, 3, 8, 40, "noise", 5), (70, 6, 8, 3, "zero", 4), (130, 10, 8, 2, "gradient", 6), (90, 8, 16, 100, "mixed", 3), (77, 6, 8, 127, "hard", 3),
    (200, 4, 4, 1, "noise", 4), (150, 9, 8, 5, "smooth", 5)]))
def test_group_decoder_near_lossless_batches(group, w, h, bits, near, kind, count):
    """Near-lossless single-component frames, `count` different ones per launch: reconstruction, runs interrupted into the context
    of |Ra - Rb| <= NEAR, small qbpp (long limited-length codes) -- the oracle's reconstruction, no scan handed to the exact decoder."""
    L = emu_bind.lib()
    frames = [synth.frame_numpy(w, h, seed=17 * f + bits + near, bits=bits, kind=kind) for f in range(count)]
    descs, outs, wants, ends, keep = _near_descs(frames, w, h, bits, near)
    res = _launch(L, descs, group)
    for f in range(count):
        assert (res[f].errc, res[f].flags, res[f].bytes) == (0, 0, ends[f]), f
        assert outs[f].tobytes() == wants[f], f


@pytest.mark.parametrize("group,w,h,bits,near,kind,count,comps", _rotating(NEAR_GROUPS, [(40, 12, 8, 2, "mixed", 5, 3), (300, 4, 8, 3, "noise", 3, 3),
                                                                                          (33, 7, 16, 5, "mixed", 3, 3), (64, 6, 8, 1, "gradient", 3, 3),
                                                                                          (50, 9, 8, 2, "mixed", 3, 2), (70, 5, 12, 4, "mixed", 3, 4)]))
def test_group_decoder_near_lossless_line_interleaved(group, w, h, bits, near, kind, count, comps):
    """Two to four lines per pixel row (ILV_LINE) of near-lossless components on the ONE set of contexts."""
    L = emu_bind.lib()
    frames = [synth.frame_numpy(w, h, seed=19 * f + bits + near, bits=bits, components=comps, kind=kind, interleaved=True) for f in range(count)]
    descs, outs, wants, ends, keep = _near_descs(frames, w, h, bits, near, comps=comps, ilv=1)
    res = _launch(L, descs, group)
    for f in range(count):
        assert (res[f].errc, res[f].flags, res[f].bytes) == (0, 0, ends[f]), f
        assert outs[f].tobytes() == wants[f], f


@pytest.mark.parametrize("group,bits,near,count", [(16, 8, 2, 21), (32, 12, 3, 9)])
def test_group_decoder_near_lossless_with_four_wavefronts_per_workgroup(group, bits, near, count):
    L = emu_bind.lib()
    w, h = 70, 8
    frames = [synth.frame_numpy(w, h, seed=23 * f + bits, bits=bits, kind="mixed") for f in range(count)]
    descs, outs, wants, ends, keep = _near_descs(frames, w, h, bits, near)
    arr = (emu_bind.ScanDesc * count)(*descs)
    res = (emu_bind.ScanResult * count)()
    assert L.emu_decode_scans_group_waves(arr, res, count, group, 4) == 0
    for f in range(count):
        assert (res[f].errc, res[f].flags, res[f].bytes) == (0, 0, ends[f]), f
        assert outs[f].tobytes() == wants[f], f


def test_group_decoder_near_lossless_kernel_leaves_lossless_scans_alone_and_the_other_way_round():
    """The instantiation is chosen from the launch's first scan: a scan of the other kind reports kFastRetry."""
    L = emu_bind.lib()
    w, h = 40, 10
    imgs = [synth.frame_numpy(w, h, seed=f + 5, bits=8, kind="mixed") for f in range(3)]
    for nears in ([2, 0, 2], [0, 2, 0]):
        keep, descs, outs, wants = [], [], [], []
        for img, near in zip(imgs, nears):
            d, o, wnt, _, k = _near_descs([img], w, h, 8, near)
            descs += d; outs += o; wants += wnt; keep += k
        res = _launch(L, descs, 8)
        for f, near in enumerate(nears):
            if near == nears[0]:
                assert (res[f].errc, res[f].flags) == (0, 0) and outs[f].tobytes() == wants[f], (nears, f)
            else:
                assert (res[f].errc, res[f].flags) == (0, 4), (nears, f)


@pytest.mark.parametrize("bits,near,kind", [(8, 2, "mixed"), (8, 9, "zero"), (16, 3, "mixed"), (12, 1, "hard")])
def test_group_dispatch_on_mutated_near_lossless_scan_data_matches_the_oracle(bits, near, kind):
    L = emu_bind.lib()
    w, h = 48, 12
    img = synth.frame_numpy(w, h, seed=bits + near, bits=bits, kind=kind)
    base = ob.encode(img, width=w, height=h, bits_per_sample=bits, near_lossless=near)
    cont = jls_container.parse(base)
    scan = cont.scans[0]
    pc = jls_container.validated_pc(cont.pc, cont.bits, near)
    bps = 1 if bits <= 8 else 2
    rng = np.random.default_rng(bits * 17 + near)
    keep, descs, outs, wants = [], [], [], []
    for k in range(36):
        b = bytearray(base)
        how = int(rng.integers(0, 5))
        i = int(rng.integers(scan.data_start, len(b) - 2))
        if how == 0:
            b[i] ^= 1 << int(rng.integers(0, 8))
        elif how == 1:
            b[i] = int(rng.choice([0x00, 0xFF, 0x7F, 0x80]))
        elif how == 2:
            del b[i:i + int(rng.integers(1, 4))]
        elif how == 3:
            b[i:i] = bytes([int(rng.choice([0x00, 0xFF, 0x55]))])
        data = bytes(b)
        try:
            wants.append((0, ob.decode(data)[1].tobytes()))
        except ob.OracleError as e:
            wants.append((e.errc, None))
        pix = np.zeros(w * h * bps, dtype=np.uint8)
        outs.append(pix)
        descs.append(emu_bind.make_desc(w, h, 1, 0, bits, near, 0, pc, 0, pix, w * bps, _stream_copy(data, scan.data_start), keep))
    res = _launch(L, descs, 16)
    retry = [k for k in range(len(descs)) if res[k].flags & 4]
    for k in retry:
        one = (emu_bind.ScanResult * 1)()
        L.emu_decode_scans_wave((emu_bind.ScanDesc * 1)(descs[k]), one, 1)
        res[k].errc, res[k].flags, res[k].bytes = one[0].errc, one[0].flags, one[0].bytes
    for k, want in enumerate(wants):
        if want[0] == 0:
            assert res[k].errc == 0 and outs[k].tobytes() == want[1], k
        else:
            assert res[k].errc == want[0], (k, res[k].errc, want[0])


def test_group_decoder_near_lossless_random_parameters():
    """Random near-lossless single-component parameter sets (bits 2..16, NEAR up to its maximum, custom thresholds / RESET)."""
    from test_oracle_vs_reference import _image
    L = emu_bind.lib()
    rng = np.random.default_rng(977)
    done = 0
    for it in range(50):
        bits = int(rng.integers(2, 17))
        w, h = int(rng.choice([1, 2, 5, 17, 64, 65, 130])), int(rng.choice([1, 2, 3, 8, 21]))
        maxval = (1 << bits) - 1
        near = int(rng.integers(1, min(255, maxval // 2) + 1))
        if rng.random() < 0.5:
            near = min(near, int(rng.integers(1, 8)))
        kind = str(rng.choice(["rand", "smooth", "gradient", "mixed", "zero", "hard"]))
        preset = None
        if rng.random() < 0.3:
            t1 = int(rng.integers(near + 1, maxval + 1))
            t2 = int(rng.integers(t1, maxval + 1))
            t3 = int(rng.integers(t2, maxval + 1))
            preset = (0, t1, t2, t3, int(rng.integers(3, max(255, maxval) + 1)))
        pc = jls_container.validated_pc(preset or (0,) * 5, bits, near)
        if not _group_eligible(bits, pc):
            continue
        count = int(rng.integers(1, 6))
        frames = [np.ascontiguousarray(_image(rng, w, h, bits, 1, 0, kind, it + f)) for f in range(count)]
        descs, outs, wants, ends, keep = _near_descs(frames, w, h, bits, near, preset=preset)
        group = int(rng.choice(NEAR_GROUPS))
        res = _launch(L, descs, group)
        for f in range(count):
            tag = (it, f, bits, near, w, h, kind, preset, group)
            assert (res[f].errc, res[f].flags) == (0, 0) and outs[f].tobytes() == wants[f], tag
        done += 1
    assert done >= 20


# ---- scan_group_pixels.hip: sample-interleaved scans (2..4 components per pixel), lossless and near-lossless ----------
PIXEL_CASES = [c for c in common.cases() if c["errc"] == 0 and "file" in c and c["interleave_mode"] == 2 and
               c["width"] * c["height"] <= 128 * 128]


def _launch_pixels(L, descs, group):
    n = len(descs)
    arr = (emu_bind.ScanDesc * n)(*descs)
    res = (emu_bind.ScanResult * n)()
    assert L.emu_decode_pixels_group(arr, res, n, group) == 0
    return res


@pytest.mark.parametrize("c", PIXEL_CASES, ids=lambda c: c["name"])
def test_pixel_group_decoder_matches_reference_pixels(c):
    L = emu_bind.lib()
    with open(f"{common.GOLDEN}/{c['file']}", "rb") as f:
        jls = f.read()
    cont = jls_container.parse(jls)
    scan = cont.scans[0]
    pc = jls_container.validated_pc(cont.pc, cont.bits, scan.near)
    if not _group_eligible(cont.bits, pc):
        pytest.skip("the exact decoder takes this one")
    bps = 1 if cont.bits <= 8 else 2
    w, h, nc = cont.width, cont.height, scan.components
    keep = []
    pix = np.zeros(w * h * nc * bps, dtype=np.uint8)
    d = emu_bind.make_desc(w, h, nc, 2, cont.bits, scan.near, cont.transform, pc, 0, pix, w * nc * bps,
                           _stream_copy(jls, scan.data_start), keep)
    group = [8, 16, 32][len(c["name"]) % 3]
    res = _launch_pixels(L, [d], group)
    assert (res[0].errc, res[0].flags) == (0, 0), "a valid stream must not need the exact decoder"
    assert res[0].bytes == scan.data_end - scan.data_start
    assert common.sha(pix.tobytes()) == c["decoded_sha256"]


@pytest.mark.parametrize("group,w,h,bits,comps,near,xform,kind,count", _rotating([8, 16, 32], [(40, 12, 8, 3, 0, 1, "mixed", 5), (37, 9, 8, 3, 2, 0, "gradient", 3), (33, 7, 16, 3, 0, 3, "mixed", 3),
                          (33, 7, 8, 4, 1, 0, "mixed", 5), (300, 4, 8, 3, 0, 2, "noise", 3), (1, 5, 8, 2, 0, 0, "mixed", 2),
                          (64, 6, 8, 3, 0, 0, "zero", 3), (20, 6, 12, 3, 3, 0, "hard", 2), (90, 8, 8, 4, 0, 0, "hard", 3),
                          (130, 6, 5, 3, 0, 0, "mixed", 3), (257, 5, 8, 2, 0, 0, "noise", 4), (75, 10, 8, 3, 0, 3, "gradient", 5), (200, 6, 8, 3, 3, 0, "noise", 3),
                          (90, 8, 7, 2, 1, 0, "mixed", 3), (64, 10, 8, 3, 5, 0, "hard", 2), (50, 8, 8, 3, 100, 0, "mixed", 2)]))
def test_pixel_group_decoder_batches(group, w, h, bits, comps, near, xform, kind, count):
    """`count` different frames per launch; what they decode to is what the oracle decodes (near-lossless: the reconstructed
    samples, bit for bit)."""
    L = emu_bind.lib()
    bps = 1 if bits <= 8 else 2
    keep, descs, outs, wants, ends = [], [], [], [], []
    for f in range(count):
        img = synth.frame_numpy(w, h, seed=17 * f + bits + comps, bits=bits, components=comps, kind=kind, interleaved=True)
        jls = ob.encode(img, width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=2,
                        near_lossless=near, color_transformation=xform)
        wants.append(ob.decode(jls)[1].tobytes())
        cont = jls_container.parse(jls)
        pc = jls_container.validated_pc(cont.pc, cont.bits, near)
        pix = np.zeros(w * h * comps * bps, dtype=np.uint8)
        descs.append(emu_bind.make_desc(w, h, comps, 2, bits, near, xform, pc, 0, pix, w * comps * bps,
                                        _stream_copy(jls, cont.scans[0].data_start), keep))
        outs.append(pix)
        ends.append(cont.scans[0].data_end - cont.scans[0].data_start)
    res = _launch_pixels(L, descs, group)
    for f in range(count):
        assert (res[f].errc, res[f].flags, res[f].bytes) == (0, 0, ends[f]), f
        assert outs[f].tobytes() == wants[f], f


@pytest.mark.parametrize("group,w,h,bits,near,kind,count", _rotating([8, 16, 32], [(64, 20, 8, 3, "mixed", 5), (300, 5, 8, 1, "noise", 3), (37, 9, 16, 5, "gradient", 3),
                                                      (1, 9, 8, 2, "mixed", 3), (70, 6, 8, 2, "zero", 4), (41, 7, 12, 2, "hard", 3),
                                                      (130, 6, 5, 1, "mixed", 3), (50, 8, 8, 100, "mixed", 2)]))
def test_pixel_group_decoder_near_lossless_single_component(group, w, h, bits, near, kind, count):
    """Near-lossless single-component scans on the pixel kernel with one component per pixel (run interruptions choose
    between the two run contexts by |Ra - Rb| <= NEAR): what they decode to is what the oracle decodes, bit for bit."""
    L = emu_bind.lib()
    bps = 1 if bits <= 8 else 2
    keep, descs, outs, wants, ends = [], [], [], [], []
    for f in range(count):
        img = synth.frame_numpy(w, h, seed=19 * f + bits + near, bits=bits, kind=kind)
        jls = ob.encode(img, width=w, height=h, bits_per_sample=bits, near_lossless=near)
        wants.append(ob.decode(jls)[1].tobytes())
        cont = jls_container.parse(jls)
        pc = jls_container.validated_pc(cont.pc, cont.bits, near)
        pix = np.zeros(w * h * bps, dtype=np.uint8)
        descs.append(emu_bind.make_desc(w, h, 1, 0, bits, near, 0, pc, 0, pix, w * bps, _stream_copy(jls, cont.scans[0].data_start), keep))
        outs.append(pix)
        ends.append(cont.scans[0].data_end - cont.scans[0].data_start)
    res = _launch_pixels(L, descs, group)
    for f in range(count):
        assert (res[f].errc, res[f].flags, res[f].bytes) == (0, 0, ends[f]), f
        assert outs[f].tobytes() == wants[f], f


@pytest.mark.parametrize("group,w,h,bits,comps,near,kind,count", _rotating([8, 16, 32], [(40, 12, 8, 3, 2, "mixed", 4), (300, 4, 8, 3, 1, "noise", 3), (33, 7, 16, 3, 5, "mixed", 2),
                                                            (1, 5, 8, 2, 2, "mixed", 2), (64, 6, 8, 3, 2, "zero", 3), (90, 8, 8, 4, 3, "hard", 2)]))
def test_pixel_group_decoder_near_lossless_line_interleaved(group, w, h, bits, comps, near, kind, count):
    """Near-lossless LINE-interleaved scans: every component keeps its own pair of lines and its RUNindex."""
    L = emu_bind.lib()
    bps = 1 if bits <= 8 else 2
    keep, descs, outs, wants, ends = [], [], [], [], []
    for f in range(count):
        img = synth.frame_numpy(w, h, seed=23 * f + bits + comps, bits=bits, components=comps, kind=kind, interleaved=True)
        jls = ob.encode(img, width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=1, near_lossless=near)
        wants.append(ob.decode(jls)[1].tobytes())
        cont = jls_container.parse(jls)
        pc = jls_container.validated_pc(cont.pc, cont.bits, near)
        pix = np.zeros(w * h * comps * bps, dtype=np.uint8)
        descs.append(emu_bind.make_desc(w, h, comps, 1, bits, near, 0, pc, 0, pix, w * comps * bps,
                                        _stream_copy(jls, cont.scans[0].data_start), keep))
        outs.append(pix)
        ends.append(cont.scans[0].data_end - cont.scans[0].data_start)
    res = _launch_pixels(L, descs, group)
    for f in range(count):
        assert (res[f].errc, res[f].flags, res[f].bytes) == (0, 0, ends[f]), f
        assert outs[f].tobytes() == wants[f], f


def test_pixel_group_dispatch_on_mutated_scan_data_matches_the_oracle():
    """Group kernel, then the exact wave decoder for the scans that reported kFastRetry, on mutated RGB streams."""
    L = emu_bind.lib()
    w, h = 40, 10
    for near, xform in ((0, 1), (2, 0)):
        img = synth.frame_numpy(w, h, seed=3 + near, components=3, kind="mixed", interleaved=True)
        base = ob.encode(img, width=w, height=h, component_count=3, interleave_mode=2, near_lossless=near, color_transformation=xform)
        cont = jls_container.parse(base)
        scan = cont.scans[0]
        pc = jls_container.validated_pc(cont.pc, cont.bits, near)
        rng = np.random.default_rng(50 + near)
        keep, descs, outs, wants = [], [], [], []
        for k in range(24):
            b = bytearray(base)
            how = int(rng.integers(0, 4))
            i = int(rng.integers(scan.data_start, len(b) - 2))
            if how == 0:
                b[i] ^= 1 << int(rng.integers(0, 8))
            elif how == 1:
                b[i] = int(rng.choice([0x00, 0xFF, 0x7F, 0x80]))
            elif how == 2:
                del b[i:i + int(rng.integers(1, 4))]
            data = bytes(b)
            try:
                wants.append((0, ob.decode(data)[1].tobytes()))
            except ob.OracleError as e:
                wants.append((e.errc, None))
            pix = np.zeros(w * h * 3, dtype=np.uint8)
            outs.append(pix)
            descs.append(emu_bind.make_desc(w, h, 3, 2, 8, near, xform, pc, 0, pix, w * 3, _stream_copy(data, scan.data_start), keep))
        res = _launch_pixels(L, descs, 16)
        for k in range(len(descs)):
            if res[k].flags & 4:
                one = (emu_bind.ScanResult * 1)()
                L.emu_decode_scans_wave((emu_bind.ScanDesc * 1)(descs[k]), one, 1)
                res[k].errc, res[k].flags, res[k].bytes = one[0].errc, one[0].flags, one[0].bytes
        for k, want in enumerate(wants):
            if want[0] == 0:
                assert res[k].errc == 0 and outs[k].tobytes() == want[1], (near, k)
            else:
                assert res[k].errc == want[0], (near, k, res[k].errc, want[0])


# ---- scan_group_encode.hip: the encoder for near-lossless single-component and sample-interleaved scans --------------
def _encode_group(L, descs, group):
    n = len(descs)
    arr = (emu_bind.ScanDesc * n)(*descs)
    res = (emu_bind.ScanResult * n)()
    assert L.emu_encode_pixels_group(arr, res, n, group) == 0
    return res


@pytest.mark.parametrize("group,w,h,bits,comps,near,xform,kind,count", _rotating([8, 16, 32, 64], [(40, 12, 8, 3, 2, 0, "mixed", 3), (64, 20, 8, 1, 3, 0, "mixed", 5), (37, 9, 16, 1, 5, 0, "gradient", 3),
                          (33, 7, 16, 3, 0, 3, "mixed", 2), (33, 7, 8, 4, 1, 0, "mixed", 3), (700, 4, 8, 1, 1, 0, "noise", 3),
                          (300, 4, 8, 3, 2, 0, "noise", 2), (64, 6, 8, 3, 2, 0, "zero", 2), (1, 5, 8, 1, 2, 0, "mixed", 2),
                          (20, 6, 12, 3, 2, 0, "hard", 2), (48, 9, 8, 2, 0, 0, "mixed", 3), (90, 8, 8, 3, 0, 1, "hard", 2),
                          (130, 6, 5, 3, 1, 0, "mixed", 3), (50, 8, 8, 3, 100, 0, "mixed", 2), (257, 5, 8, 1, 0, 0, "noise", 3),
                          (75, 10, 7, 1, 2, 0, "gradient", 3), (300, 5, 8, 4, 3, 0, "noise", 2)]))
def test_group_encoder_matches_reference_scan_bytes(group, w, h, bits, comps, near, xform, kind, count):
    """`count` different frames per launch: every scan's bytes equal the reference's (the oracle's) entropy-coded segment."""
    L = emu_bind.lib()
    ilv = 2 if comps > 1 else 0
    bps = 1 if bits <= 8 else 2
    keep, descs, outs, wants = [], [], [], []
    for f in range(count):
        img = synth.frame_numpy(w, h, seed=19 * f + bits + near, bits=bits, components=comps, kind=kind, interleaved=True)
        jls = ob.encode(img, width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=ilv,
                        near_lossless=near, color_transformation=xform)
        cont = jls_container.parse(jls)
        scan = cont.scans[0]
        wants.append(jls[scan.data_start:scan.data_end])
        pc = jls_container.validated_pc(cont.pc, cont.bits, near)
        pix = np.frombuffer(np.ascontiguousarray(img).tobytes(), dtype=np.uint8).copy()
        out = np.zeros(len(wants[-1]) + 64, dtype=np.uint8)
        outs.append(out)
        descs.append(emu_bind.make_desc(w, h, comps, ilv, bits, near, xform, pc, 0, pix, w * bps * comps, out, keep))
    res = _encode_group(L, descs, group)
    for f in range(count):
        assert res[f].errc == 0 and outs[f][:res[f].bytes].tobytes() == wants[f], f


@pytest.mark.parametrize("group,w,h,bits,comps,near,xform,kind,count", _rotating([8, 16, 32, 64], [(40, 12, 8, 3, 2, 0, "mixed", 3), (300, 4, 8, 3, 1, 0, "noise", 2), (33, 7, 16, 3, 5, 0, "mixed", 2),
                          (1, 5, 8, 2, 2, 0, "mixed", 2), (64, 6, 8, 3, 2, 0, "zero", 2), (90, 8, 8, 4, 3, 0, "hard", 2),
                          (48, 9, 8, 3, 0, 1, "mixed", 2), (57, 6, 12, 2, 3, 0, "gradient", 2)]))
def test_group_encoder_line_interleaved_matches_reference_scan_bytes(group, w, h, bits, comps, near, xform, kind, count):
    """LINE-interleaved scans on the group encoder (a pair of lines and a RUNindex per component, the user's row
    de-interleaved when its first component starts): every scan's bytes equal the reference's entropy-coded segment."""
    L = emu_bind.lib()
    bps = 1 if bits <= 8 else 2
    keep, descs, outs, wants = [], [], [], []
    for f in range(count):
        img = synth.frame_numpy(w, h, seed=29 * f + bits + near, bits=bits, components=comps, kind=kind, interleaved=True)
        jls = ob.encode(img, width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=1,
                        near_lossless=near, color_transformation=xform)
        cont = jls_container.parse(jls)
        scan = cont.scans[0]
        wants.append(jls[scan.data_start:scan.data_end])
        pc = jls_container.validated_pc(cont.pc, cont.bits, near)
        pix = np.frombuffer(np.ascontiguousarray(img).tobytes(), dtype=np.uint8).copy()
        out = np.zeros(len(wants[-1]) + 64, dtype=np.uint8)
        outs.append(out)
        descs.append(emu_bind.make_desc(w, h, comps, 1, bits, near, xform, pc, 0, pix, w * bps * comps, out, keep))
    res = _encode_group(L, descs, group)
    for f in range(count):
        assert res[f].errc == 0 and outs[f][:res[f].bytes].tobytes() == wants[f], f


@pytest.mark.parametrize("comps,near", [(1, 2), (3, 2), (1, 0)])
def test_group_encoder_destination_too_small_boundary(comps, near):
    """The verdict destination_too_small depends on the reference's 32-bit flush history (src/scan_encoder.hpp:117-120): for
    every capacity around the size of the output the group encoder must agree with the kernel that restates that history
    lane by lane (encode_scans_serial, itself pinned to the reference by tests/test_oracle_vs_reference.py)."""
    L = emu_bind.lib()
    w, h = 41, 9
    ilv = 2 if comps > 1 else 0
    img = synth.frame_numpy(w, h, seed=5, components=comps, kind="mixed", interleaved=True)
    jls = ob.encode(img, width=w, height=h, component_count=comps, interleave_mode=ilv, near_lossless=near)
    cont = jls_container.parse(jls)
    scan = cont.scans[0]
    size = scan.data_end - scan.data_start
    pc = jls_container.validated_pc(cont.pc, 8, near)
    pix = np.frombuffer(np.ascontiguousarray(img).tobytes(), dtype=np.uint8).copy()
    for cap in list(range(size - 6, size + 7)) + [0, 1, 3, 4, size // 2]:
        got = []
        for kernel in ("group", "serial"):
            keep = []
            out = np.zeros(size + 64, dtype=np.uint8)
            d = emu_bind.make_desc(w, h, comps, ilv, 8, near, 0, pc, 0, pix, w * comps, out, keep)
            d.stream_capacity = cap
            res = (emu_bind.ScanResult * 1)()
            if kernel == "group":
                assert L.emu_encode_pixels_group((emu_bind.ScanDesc * 1)(d), res, 1, 16) == 0
            else:
                L.emu_encode_scans_serial((emu_bind.ScanDesc * 1)(d), res, 1)
            got.append((res[0].errc, res[0].bytes if res[0].errc == 0 else 0, out[:res[0].bytes].tobytes() if res[0].errc == 0 else b""))
        assert got[0] == got[1], cap
