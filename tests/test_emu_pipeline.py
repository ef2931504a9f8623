"""Kernel-logic check on the CPU for the parallel lossless encoder (lossless_pipeline.hip compiled for the host by
tests/emu): every stage runs with the launch geometry of the product, the final scan bytes must equal the reference's."""
import ctypes as C

import numpy as np
import pytest

import common
import emu_bind
import jls_container
import oracle_bind as ob
from charls_amd import synth

import os

FULL = os.environ.get("CHARLS_AMD_QUICK_EMU") != "1"  # every case by default; the quick subset is for slow hosts
_SUBSET = {"gray8_64x48", "gray8_w1", "gray8_h1", "gray8_1x1", "tiny_gray12", "tiny_gray16_noise", "tiny_gray2",
           "tiny_gray8_noise", "tiny_rgb8_ilv0", "gray8_maxval100"}
ELIGIBLE = [c for c in common.cases()
            if c["errc"] == 0 and "file" in c and c["near_lossless"] == 0 and c["width"] * c["height"] <= 128 * 128 and
            (c["component_count"] == 1 or c["interleave_mode"] == 0) and (FULL or c["name"] in _SUBSET)]


def _encode_planes(L, planes, width, height, bits, pc, capacity_slack=64):
    """planes: list of 2-D arrays (one scan each). Returns list of (errc, flags, bytes)."""
    keep, descs, outs = [], [], []
    for pl in planes:
        pix = np.frombuffer(np.ascontiguousarray(pl).tobytes(), dtype=np.uint8).copy()
        out = np.zeros(width * height * 4 + 1024 if capacity_slack is None else capacity_slack, dtype=np.uint8)
        outs.append(out)
        descs.append(emu_bind.make_desc(width, height, 1, 0, bits, 0, 0, pc, 0, pix, width * (1 if bits <= 8 else 2), out, keep))
    arr = (emu_bind.ScanDesc * len(descs))(*descs)
    res = (emu_bind.ScanResult * len(descs))()
    L.emu_encode_pipeline(arr, res, len(descs))
    return [(r.errc, r.flags, o[:r.bytes].tobytes()) for r, o in zip(res, outs)]


@pytest.mark.parametrize("c", ELIGIBLE, ids=lambda c: c["name"])
def test_pipeline_matches_reference_scan_bytes(c):
    L = emu_bind.lib()
    with open(f"{common.GOLDEN}/{c['file']}", "rb") as f:
        jls = f.read()
    cont = jls_container.parse(jls)
    pc = jls_container.validated_pc(tuple(c["preset"]) if c["preset"] else (0,) * 5, c["bits_per_sample"], 0)
    img = common.case_input(c)
    planes = [img] if img.ndim == 2 else [img[i] for i in range(img.shape[0])]
    size = max(s.data_end - s.data_start for s in cont.scans) + 64
    got = _encode_planes(L, planes, c["width"], c["height"], c["bits_per_sample"], pc, capacity_slack=size)
    for (errc, flags, data), scan in zip(got, cont.scans):
        assert errc == 0
        assert data == jls[scan.data_start:scan.data_end]


@pytest.mark.parametrize("kind,bits,w,h,seed", [("mixed", 8, 130, 11, 1), ("zero", 8, 130, 9, 2), ("hard", 12, 65, 9, 4),
                                                ("mixed", 16, 129, 7, 5), ("mixed", 8, 1, 50, 7), ("noise", 16, 40, 12, 9)])
def test_pipeline_batch_of_seeded_frames(kind, bits, w, h, seed):
    """Several frames in one launch (lanes of the chain kernel interleave frames): each equals the oracle."""
    L = emu_bind.lib()
    frames = [synth.frame_numpy(w, h, seed=seed * 10 + f, bits=bits, kind=kind) for f in range(3)]
    pc = jls_container.validated_pc((0,) * 5, bits, 0)
    got = _encode_planes(L, frames, w, h, bits, pc, capacity_slack=w * h * 4 + 1024)
    for img, (errc, flags, data) in zip(frames, got):
        want = ob.encode(img, width=w, height=h, bits_per_sample=bits)
        cont = jls_container.parse(want)
        assert errc == 0
        assert data == want[cont.scans[0].data_start:cont.scans[0].data_end]


@pytest.mark.skipif(not FULL, reason="210k samples through the thread-per-lane emulation (skipped with CHARLS_AMD_QUICK_EMU=1)")
def test_pipeline_long_runs_cross_run_index_31():
    """Runs long enough to walk RUNindex up to 31 and back (J = 15): src/scan_encoder.hpp:53-73."""
    L = emu_bind.lib()
    w, h = 70000, 3  # one line of 70000 equal samples needs 2^15 blocks
    img = np.zeros((h, w), dtype=np.uint8)
    img[1, 50000:] = 9
    img[2, ::2] = 3
    pc = jls_container.validated_pc((0,) * 5, 8, 0)
    (errc, flags, data), = _encode_planes(L, [img], w, h, 8, pc, capacity_slack=w * h * 2 + 1024)
    want = ob.encode(img, width=w, height=h)
    cont = jls_container.parse(want)
    assert errc == 0 and data == want[cont.scans[0].data_start:cont.scans[0].data_end]


def test_pipeline_long_runs_walk_run_index():
    """Runs that walk RUNindex up to 24 and back (J = 8), and runs that end exactly at the line end."""
    L = emu_bind.lib()
    w, h = 3000, 3
    img = np.zeros((h, w), dtype=np.uint8)
    img[1, 2000:] = 9
    img[2, ::2] = 3
    pc = jls_container.validated_pc((0,) * 5, 8, 0)
    (errc, flags, data), = _encode_planes(L, [img], w, h, 8, pc, capacity_slack=w * h * 2 + 1024)
    want = ob.encode(img, width=w, height=h)
    cont = jls_container.parse(want)
    assert errc == 0 and data == want[cont.scans[0].data_start:cont.scans[0].data_end]


def test_pipeline_destination_too_small_and_knife_edge():
    L = emu_bind.lib()
    img = synth.frame_numpy(64, 64, seed=3, kind="mixed")[:16, :40].copy()
    img = np.ascontiguousarray(np.pad(img, ((0, 48), (0, 24))))  # 64x64 with flat borders: cheap to emulate
    want = ob.encode(img, width=64, height=64)
    cont = jls_container.parse(want)
    n = cont.scans[0].data_end - cont.scans[0].data_start
    pc = jls_container.validated_pc((0,) * 5, 8, 0)
    (errc, flags, data), = _encode_planes(L, [img], 64, 64, 8, pc, capacity_slack=n - 1)
    assert errc == 3
    for slack in ((0, 1, 3) if FULL else (0, 3)):
        (errc, flags, data), = _encode_planes(L, [img], 64, 64, 8, pc, capacity_slack=n + slack)
        assert errc == 0 and flags == 2  # host re-runs the exact serial kernel for these
    (errc, flags, data), = _encode_planes(L, [img], 64, 64, 8, pc, capacity_slack=n + 4)
    assert errc == 0 and flags == 0 and len(data) == n


@pytest.mark.parametrize("bits,reset", [(8, 3), (8, 31), (8, 64), (8, 255), (16, 256), (16, 257)] +
                         ([(8, 4), (16, 258), (16, 300)] if FULL else []))
def test_pipeline_reset_values_with_long_chains(bits, reset):
    """Few contexts, chains of thousands of events: the code_events stage crosses many halving points (RESET is stored
    through a uint8 by the reference, so 256/257/258 behave as 0/1/2 -- SURVEY F8)."""
    L = emu_bind.lib()
    w, h = (96, 48) if FULL else (64, 24)
    rng = np.random.default_rng(reset)
    base = (np.arange(w)[None, :] // 7 + np.arange(h)[:, None] // 5) * (3 if bits == 8 else 700)
    noise = rng.integers(-2, 3, size=(h, w)) * (1 if bits == 8 else 900)
    img = np.clip(base + noise + (40 if bits == 8 else 9000), 0, (1 << bits) - 1).astype(np.uint8 if bits == 8 else np.uint16)
    preset = (0, 0, 0, 0, reset)
    pc = jls_container.validated_pc(preset, bits, 0)
    want = ob.encode(img, width=w, height=h, bits_per_sample=bits, preset=preset)
    cont = jls_container.parse(want)
    (errc, flags, data), = _encode_planes(L, [img], w, h, bits, pc, capacity_slack=w * h * 4 + 1024)
    assert errc == 0
    assert data == want[cont.scans[0].data_start:cont.scans[0].data_end]


def _encode_interleaved(L, img, width, height, comps, bits, xform, capacity, ilv=2):
    keep = []
    pix = np.frombuffer(np.ascontiguousarray(img).tobytes(), dtype=np.uint8).copy()
    out = np.zeros(capacity, dtype=np.uint8)
    pc = jls_container.validated_pc((0,) * 5, bits, 0)
    d = emu_bind.make_desc(width, height, comps, ilv, bits, 0, xform, pc, 0, pix, width * comps * (1 if bits <= 8 else 2), out, keep)
    res = (emu_bind.ScanResult * 1)()
    L.emu_encode_pipeline((emu_bind.ScanDesc * 1)(d), res, 1)
    return res[0].errc, res[0].flags, out[:res[0].bytes].tobytes()


@pytest.mark.parametrize("kind,bits,comps,xform,w,h", [("mixed", 8, 3, 0, 33, 9), ("zero", 8, 3, 0, 70, 6), ("mixed", 8, 3, 1, 40, 7),
                                                       ("mixed", 8, 3, 2, 20, 5), ("mixed", 8, 3, 3, 21, 5), ("mixed", 16, 3, 1, 18, 5),
                                                       ("hard", 12, 2, 0, 19, 6), ("mixed", 8, 4, 0, 17, 5), ("mixed", 8, 3, 0, 1, 9),
                                                       ("runs", 8, 3, 0, 90, 8)])
def test_pipeline_sample_interleaved_scans(kind, bits, comps, xform, w, h):
    """ILV_SAMPLE lossless (2..4 components, HP1..3 for RGB): one set of contexts shared by the components, pixel-level
    run mode, run-interruption codes of every component against run context 0."""
    L = emu_bind.lib()
    if kind == "runs":  # flat areas with isolated differing pixels and single differing components
        rng = np.random.default_rng(4)
        img = np.full((h, w, comps), 50, dtype=np.uint8)
        for _ in range(25):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            img[y, x, int(rng.integers(0, comps))] = int(rng.integers(0, 256))
        img[3, 10:30] = (9, 200, 77)
    else:
        planes = [synth.frame_numpy(w, h, seed=60 + c, bits=bits, kind=kind) for c in range(comps)]
        img = np.stack(planes, axis=-1)
    want = ob.encode(img, width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=2,
                     color_transformation=xform)
    cont = jls_container.parse(want)
    errc, flags, data = _encode_interleaved(L, img, w, h, comps, bits, xform, w * h * comps * 4 + 1024)
    assert errc == 0
    assert data == want[cont.scans[0].data_start:cont.scans[0].data_end]


def test_pipeline_sample_interleaved_random_small_images():
    """Many tiny ILV_SAMPLE images with few grey levels: runs of every length, runs ended by a single component,
    components with a zero context inside non-run pixels (regular context 0), all colour transforms."""
    L = emu_bind.lib()
    rng = np.random.default_rng(12)
    for trial in range(40 if FULL else 14):
        comps = int(rng.choice([2, 3, 3, 4]))
        w, h = int(rng.integers(1, 12)), int(rng.integers(1, 6))
        levels = int(rng.choice([2, 3, 256]))
        img = (rng.integers(0, levels, size=(h, w, comps)) * (255 // (levels - 1) if levels < 256 else 1)).astype(np.uint8)
        if trial % 3 == 0:
            img[:, w // 2:] = img[:, w // 2:w // 2 + 1]  # long runs to the end of the line
        xform = int(rng.integers(0, 4)) if comps == 3 else 0
        want = ob.encode(img, width=w, height=h, component_count=comps, interleave_mode=2, color_transformation=xform)
        cont = jls_container.parse(want)
        errc, flags, data = _encode_interleaved(L, img, w, h, comps, 8, xform, w * h * comps * 4 + 1024)
        assert errc == 0 and data == want[cont.scans[0].data_start:cont.scans[0].data_end], (trial, w, h, comps, xform)


def test_pipeline_line_interleaved_scans():
    """ILV_LINE lossless: the components of a pixel row are coded as lines of their own, sharing the contexts but each
    with its own RUNindex."""
    L = emu_bind.lib()
    rng = np.random.default_rng(21)
    cases = [("mixed", 8, 3, 0, 33, 7), ("mixed", 8, 3, 1, 20, 5), ("mixed", 16, 3, 3, 12, 4), ("hard", 12, 2, 0, 19, 5),
             ("mixed", 8, 4, 0, 9, 4)]
    for kind, bits, comps, xform, w, h in cases:
        img = np.stack([synth.frame_numpy(w, h, seed=80 + c, bits=bits, kind=kind) for c in range(comps)], axis=-1)
        want = ob.encode(img, width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=1,
                         color_transformation=xform)
        cont = jls_container.parse(want)
        errc, flags, data = _encode_interleaved(L, img, w, h, comps, bits, xform, w * h * comps * 4 + 1024, ilv=1)
        assert errc == 0 and data == want[cont.scans[0].data_start:cont.scans[0].data_end], (kind, bits, comps, xform)
    for trial in range(24 if FULL else 10):  # few grey levels: runs in every component, RUNindex walks per component
        comps = int(rng.choice([2, 3, 4]))
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 6))
        img = (rng.integers(0, 2, size=(h, w, comps)) * 200).astype(np.uint8)
        img[:, w // 3:, 0] = img[:, w // 3:w // 3 + 1, 0]
        xform = int(rng.integers(0, 4)) if comps == 3 else 0
        want = ob.encode(img, width=w, height=h, component_count=comps, interleave_mode=1, color_transformation=xform)
        cont = jls_container.parse(want)
        errc, flags, data = _encode_interleaved(L, img, w, h, comps, 8, xform, w * h * comps * 4 + 1024, ilv=1)
        assert errc == 0 and data == want[cont.scans[0].data_start:cont.scans[0].data_end], (trial, w, h, comps, xform)


@pytest.mark.parametrize("chunk", range(3))
def test_pipeline_and_fast_decoder_random_parameters(chunk):
    """Random lossless parameter sets (bits 2..16, custom thresholds / RESET, odd sizes, planar and interleaved) through the
    emulated parallel encoder, and the single-component ones back through the emulated speed-path decoder."""
    from test_oracle_vs_reference import _image
    L = emu_bind.lib()
    rng = np.random.default_rng(700 + chunk)
    for it in range(25):
        bits = int(rng.integers(2, 17))
        comps = int(rng.choice([1, 1, 1, 2, 3, 4]))
        ilv = 0 if comps == 1 else int(rng.integers(1, 3))
        w, h = int(rng.choice([1, 2, 5, 17, 64, 65, 130])), int(rng.choice([1, 2, 3, 8, 21]))
        maxval = (1 << bits) - 1
        kind = str(rng.choice(["rand", "smooth", "gradient", "mixed", "zero", "hard"]))
        preset = (0,) * 5
        if rng.random() < 0.4:
            t1 = int(rng.integers(1, maxval + 1))
            t2 = int(rng.integers(t1, maxval + 1))
            t3 = int(rng.integers(t2, maxval + 1))
            preset = (0, t1, t2, t3, int(rng.integers(3, max(255, maxval) + 1)))
        xform = int(rng.integers(0, 4)) if (comps == 3 and bits in (8, 16) and rng.random() < 0.5) else 0
        img = _image(rng, w, h, bits, comps, ilv, kind, it)
        want = ob.encode(img, width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=ilv,
                         color_transformation=xform, preset=preset if any(preset) else None)
        cont = jls_container.parse(want)
        pc = jls_container.validated_pc(preset, bits, 0)
        keep = []
        pix = np.frombuffer(np.ascontiguousarray(img).tobytes(), dtype=np.uint8).copy()
        out = np.zeros(w * h * comps * 5 + 1024, dtype=np.uint8)
        bps = 1 if bits <= 8 else 2
        d = emu_bind.make_desc(w, h, comps, ilv, bits, 0, xform, pc, 0, pix, w * comps * bps, out, keep)
        res = (emu_bind.ScanResult * 1)()
        L.emu_encode_pipeline((emu_bind.ScanDesc * 1)(d), res, 1)
        scan = cont.scans[0]
        tag = (chunk, it, bits, comps, ilv, w, h, kind, preset, xform)
        assert res[0].errc == 0 and out[:res[0].bytes].tobytes() == want[scan.data_start:scan.data_end], tag
        if comps == 1 and (pc[4] & 0xFF) != 0:
            back = np.zeros(w * h * bps, dtype=np.uint8)
            src = np.frombuffer(want[scan.data_start:] + bytes(64), dtype=np.uint8).copy()
            dd = emu_bind.make_desc(w, h, 1, 0, bits, 0, 0, pc, 0, back, w * bps, src, keep)
            dd.stream_capacity = len(want) - scan.data_start
            r2 = (emu_bind.ScanResult * 1)()
            L.emu_decode_scans_fast((emu_bind.ScanDesc * 1)(dd), r2, 1)
            assert (r2[0].errc, r2[0].flags) == (0, 0) and back.tobytes() == np.ascontiguousarray(img).tobytes(), tag
