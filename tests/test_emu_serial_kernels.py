"""Kernel-logic check on the CPU: scan_serial.hip compiled for the host (tests/emu) vs the golden vectors, scan by scan.
This is test infrastructure -- it proves the kernel source's logic, not the product path (see tests/test_gpu_*.py)."""
import ctypes as C

import numpy as np
import pytest

import common
import emu_bind
import jls_container

import os

FULL = os.environ.get("CHARLS_AMD_QUICK_EMU") != "1"  # every golden case through every kernel by default; CHARLS_AMD_QUICK_EMU=1 = subset
CASES = [c for c in common.cases() if c["errc"] == 0 and "file" in c and c["width"] * c["height"] <= 128 * 128]
# The thread-per-lane emulation of the wave-uniform kernels costs seconds per case: by default they run a subset that
# covers every coding mode once; the one-lane kernel (cheap to emulate) always runs everything.
_WAVE_SUBSET = {"gray8_64x48", "gray8_w1", "gray8_h1", "gray8_1x1", "tiny_rgb8_ilv0", "tiny_rgb8_ilv1", "tiny_rgb8_ilv2",
                "tiny_rgb8_ilv2_near2", "tiny_rgb8_ilv1_near1", "tiny_rgb16_line_hp3", "tiny_rgb8_sample_hp1",
                "tiny_c4_ilv2_near1", "tiny_c2_ilv1", "tiny_gray12", "tiny_gray16_noise", "tiny_gray2", "tiny_gray8_near3"}
_FAST_SUBSET = {"gray8_64x48", "gray8_w1", "gray8_h1", "gray8_1x1", "tiny_gray12", "tiny_gray16_noise", "tiny_gray2",
                "tiny_gray8_noise", "tiny_rgb8_ilv0"}
WAVE_CASES = CASES if FULL else [c for c in CASES if c["name"] in _WAVE_SUBSET]


def _scan_views(c, img):
    """Yield (pixels_view_bytes, stride) per scan of the frame in the user's layout."""
    bytes_ps = 1 if c["bits_per_sample"] <= 8 else 2
    raw = np.frombuffer(img.tobytes(), dtype=np.uint8)
    w, h, nc = c["width"], c["height"], c["component_count"]
    if c["interleave_mode"] == 0:
        plane = w * h * bytes_ps
        return [(raw[i * plane:(i + 1) * plane].copy(), w * bytes_ps) for i in range(nc)]
    return [(raw.copy(), w * nc * bytes_ps)]


@pytest.mark.parametrize("c", CASES, ids=lambda c: c["name"])
def test_emulated_encode_kernel_matches_reference_scan_bytes(c):
    L = emu_bind.lib()
    with open(f"{common.GOLDEN}/{c['file']}", "rb") as f:
        jls = f.read()
    cont = jls_container.parse(jls)
    pc = jls_container.validated_pc(tuple(c["preset"]) if c["preset"] else (0,) * 5, c["bits_per_sample"],
                                    c["near_lossless"])
    img = common.case_input(c)
    views = _scan_views(c, img)
    assert len(views) == len(cont.scans)
    keep, descs, outs = [], [], []
    for (pix, stride), scan in zip(views, cont.scans):
        out = np.zeros(scan.data_end - scan.data_start + 64, dtype=np.uint8)
        outs.append(out)
        descs.append(emu_bind.make_desc(c["width"], c["height"], scan.components, scan.ilv, c["bits_per_sample"],
                                        scan.near, c["color_transformation"], pc, 0, pix, stride, out, keep))
    arr = (emu_bind.ScanDesc * len(descs))(*descs)
    res = (emu_bind.ScanResult * len(descs))()
    L.emu_encode_scans_serial(arr, res, len(descs))
    for r, out, scan in zip(res, outs, cont.scans):
        assert r.errc == 0
        assert out[:r.bytes].tobytes() == jls[scan.data_start:scan.data_end]


DECODERS = ["emu_decode_scans_serial", "emu_decode_scans_wave"]


def _stream_copy(jls, start):
    """Source bytes of a scan placed at a deliberately odd offset of a padded buffer (the ring refill aligns down)."""
    raw = np.zeros(len(jls) - start + 64 + 32, dtype=np.uint8)
    view = raw[19:19 + len(jls) - start]
    view[:] = np.frombuffer(jls[start:], dtype=np.uint8)
    return view


def _decode_params():
    out = [pytest.param(c, "emu_decode_scans_serial", id=f"{c['name']}-serial") for c in CASES]
    out += [pytest.param(c, "emu_decode_scans_wave", id=f"{c['name']}-wave") for c in WAVE_CASES]
    return out


@pytest.mark.parametrize("c,kernel", _decode_params())
def test_emulated_decode_kernel_matches_reference_pixels(c, kernel):
    L = emu_bind.lib()
    with open(f"{common.GOLDEN}/{c['file']}", "rb") as f:
        jls = f.read()
    cont = jls_container.parse(jls)
    pc = jls_container.validated_pc(cont.pc, cont.bits, c["near_lossless"])
    bytes_ps = 1 if cont.bits <= 8 else 2
    w, h = cont.width, cont.height
    keep, descs, outs = [], [], []
    for scan in cont.scans:
        stride = w * bytes_ps * (1 if scan.ilv == 0 else scan.components)
        pix = np.zeros(stride * h, dtype=np.uint8)
        outs.append(pix)
        src = _stream_copy(jls, scan.data_start)
        descs.append(emu_bind.make_desc(w, h, scan.components, scan.ilv, cont.bits, scan.near, cont.transform, pc,
                                        cont.restart_interval, pix, stride, src, keep))
    res = (emu_bind.ScanResult * len(descs))()
    for k, dsc in enumerate(descs):  # one launch per scan: the wave kernel is specialised per geometry
        getattr(L, kernel)((emu_bind.ScanDesc * 1)(dsc), C.byref(res[k]), 1)
    got = b"".join(o.tobytes() for o in outs)
    for r, scan in zip(res, cont.scans):
        assert r.errc == 0
        assert r.bytes == scan.data_end - scan.data_start
    assert common.sha(got) == c["decoded_sha256"]


@pytest.mark.parametrize("kernel", DECODERS if FULL else DECODERS[:1])
@pytest.mark.parametrize("name,pnm,ilv", [("test8_ilv_none_rm_7", "test8.ppm", 0), ("test8_ilv_sample_rm_300", "test8.ppm", 2),
                                          ("test8_ilv_line_rm_7", "test8.ppm", 1)])
def test_emulated_decode_restart_markers(name, pnm, ilv, kernel):
    L = emu_bind.lib()
    jls = common.refdata(f"{name}.jls")
    img, _ = common.read_pnm(pnm)
    want = (common.planar(img) if ilv == 0 else img).tobytes()
    cont = jls_container.parse(jls)
    pc = jls_container.validated_pc(cont.pc, cont.bits, 0)
    keep, descs, outs = [], [], []
    for scan in cont.scans:
        stride = cont.width * (1 if scan.ilv == 0 else scan.components)
        pix = np.zeros(stride * cont.height, dtype=np.uint8)
        outs.append(pix)
        src = _stream_copy(jls, scan.data_start)
        descs.append(emu_bind.make_desc(cont.width, cont.height, scan.components, scan.ilv, cont.bits, scan.near,
                                        cont.transform, pc, cont.restart_interval, pix, stride, src, keep))
    arr = (emu_bind.ScanDesc * len(descs))(*descs)
    res = (emu_bind.ScanResult * len(descs))()
    getattr(L, kernel)(arr, res, len(descs))
    assert all(r.errc == 0 for r in res)
    assert b"".join(o.tobytes() for o in outs) == want


@pytest.mark.parametrize("name,errc", [("fuzzy-input-bad-run-mode-golomb-code.jls", 5),
                                       ("fuzzy_input_golomb_16.jls", 5), ("fuzzy-input-no-valid-bits-at-the-end.jls", 5),
                                       ("no_start_byte_after_encoded_scan.jls", 4)])
@pytest.mark.parametrize("kernel", DECODERS)
def test_emulated_decode_corrupt_streams(name, errc, kernel):
    import oracle_bind as ob
    L = emu_bind.lib()
    jls = common.refdata(name)
    p = ob.read_header(jls)
    cont = jls_container.parse(jls)
    pc = jls_container.validated_pc(cont.pc, cont.bits, cont.scans[0].near)
    scan = cont.scans[0]
    bytes_ps = 1 if cont.bits <= 8 else 2
    stride = cont.width * bytes_ps * (1 if scan.ilv == 0 else scan.components)
    keep = []
    pix = np.zeros(stride * cont.height, dtype=np.uint8)
    src = _stream_copy(jls, scan.data_start)
    d = emu_bind.make_desc(cont.width, cont.height, scan.components, scan.ilv, cont.bits, scan.near, cont.transform, pc,
                           cont.restart_interval, pix, stride, src, keep)
    arr = (emu_bind.ScanDesc * 1)(d)
    res = (emu_bind.ScanResult * 1)()
    getattr(L, kernel)(arr, res, 1)
    assert res[0].errc == errc


# ---- speed path of the decoder (scan_fast_decode.hip): lossless single-component scans ---------------------------------
FAST_CASES = [c for c in CASES if c["near_lossless"] == 0 and (c["component_count"] == 1 or c["interleave_mode"] == 0) and
              (FULL or c["name"] in _FAST_SUBSET)]


@pytest.mark.parametrize("c", FAST_CASES, ids=lambda c: c["name"])
def test_emulated_fast_decoder_matches_reference_pixels(c):
    L = emu_bind.lib()
    with open(f"{common.GOLDEN}/{c['file']}", "rb") as f:
        jls = f.read()
    cont = jls_container.parse(jls)
    pc = jls_container.validated_pc(cont.pc, cont.bits, 0)
    bytes_ps = 1 if cont.bits <= 8 else 2
    w, h = cont.width, cont.height
    keep, outs = [], []
    for scan in cont.scans:
        stride = w * bytes_ps
        pix = np.zeros(stride * h, dtype=np.uint8)
        outs.append(pix)
        src = _stream_copy(jls, scan.data_start)
        d = emu_bind.make_desc(w, h, 1, 0, cont.bits, 0, 0, pc, 0, pix, stride, src, keep)
        res = (emu_bind.ScanResult * 1)()
        L.emu_decode_scans_fast((emu_bind.ScanDesc * 1)(d), res, 1)
        assert res[0].errc == 0 and res[0].flags == 0, "a valid stream must not need the exact decoder"
        assert res[0].bytes == scan.data_end - scan.data_start
    assert common.sha(b"".join(o.tobytes() for o in outs)) == c["decoded_sha256"]


@pytest.mark.parametrize("name", ["fuzzy-input-bad-run-mode-golomb-code.jls", "fuzzy_input_golomb_16.jls",
                                  "fuzzy-input-no-valid-bits-at-the-end.jls", "no_start_byte_after_encoded_scan.jls"])
def test_emulated_fast_decoder_defers_on_corrupt_streams(name):
    """The speed path never reports an error itself: anything unusual is handed to the exact decoder."""
    L = emu_bind.lib()
    jls = common.refdata(name)
    cont = jls_container.parse(jls)
    scan = cont.scans[0]
    if scan.near != 0 or scan.ilv != 0:
        pytest.skip("not a scan the speed path takes")
    pc = jls_container.validated_pc(cont.pc, cont.bits, 0)
    bytes_ps = 1 if cont.bits <= 8 else 2
    keep = []
    pix = np.zeros(cont.width * bytes_ps * cont.height, dtype=np.uint8)
    src = _stream_copy(jls, scan.data_start)
    d = emu_bind.make_desc(cont.width, cont.height, 1, 0, cont.bits, 0, 0, pc, 0, pix, cont.width * bytes_ps, src, keep)
    res = (emu_bind.ScanResult * 1)()
    L.emu_decode_scans_fast((emu_bind.ScanDesc * 1)(d), res, 1)
    assert res[0].errc == 0 and res[0].flags == 4


# ---- the decoder dispatch of runtime.hip (speed path, exact wave decoder on kFastRetry) on mutated streams -------------
def _emulated_dispatch(L, jls, cont, scan):
    """What launch_decode does for a single-component lossless scan: returns (errc, pixels or None)."""
    pc = jls_container.validated_pc(cont.pc, cont.bits, 0)
    bps = 1 if cont.bits <= 8 else 2
    keep = []
    pix = np.zeros(cont.width * cont.height * bps, dtype=np.uint8)
    src = _stream_copy(jls, scan.data_start)
    d = emu_bind.make_desc(cont.width, cont.height, 1, 0, cont.bits, 0, 0, pc, 0, pix, cont.width * bps, src, keep)
    res = (emu_bind.ScanResult * 1)()
    L.emu_decode_scans_fast((emu_bind.ScanDesc * 1)(d), res, 1)
    if res[0].flags & 4:
        L.emu_decode_scans_wave((emu_bind.ScanDesc * 1)(d), res, 1)
    return res[0].errc, res[0].bytes, pix


@pytest.mark.parametrize("bits,kind", [(8, "mixed"), (8, "zero"), (16, "mixed"), (12, "hard")])
def test_emulated_dispatch_on_mutated_scan_data_matches_the_oracle(bits, kind):
    import oracle_bind as ob
    from charls_amd import synth
    L = emu_bind.lib()
    w, h = 48, 12
    img = synth.frame_numpy(w, h, seed=bits, bits=bits, kind=kind)
    base = ob.encode(img, width=w, height=h, bits_per_sample=bits)
    cont = jls_container.parse(base)
    scan = cont.scans[0]
    rng = np.random.default_rng(bits * 7 + len(kind))
    for k in range(40):
        b = bytearray(base)
        how = int(rng.integers(0, 4))
        i = int(rng.integers(scan.data_start, len(b) - 2))
        if how == 0:
            b[i] ^= 1 << int(rng.integers(0, 8))
        elif how == 1:
            b[i] = int(rng.choice([0x00, 0xFF, 0x7F, 0x80]))
        elif how == 2:
            del b[i:i + int(rng.integers(1, 4))]
        else:
            b[i:i] = bytes([int(rng.choice([0x00, 0xFF, 0x55]))])
        data = bytes(b)
        try:
            want = (0, ob.decode(data)[1].tobytes())
        except ob.OracleError as e:
            want = (e.errc, None)
        c2 = jls_container.parse(data) if want[0] == 0 else cont
        errc, used, pix = _emulated_dispatch(L, data, cont, scan)
        if want[0] == 0:
            assert errc == 0 and pix.tobytes() == want[1], (k, how, i)
        else:
            assert errc == want[0], (k, how, i, errc, want[0])
