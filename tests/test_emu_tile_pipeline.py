"""Kernel-logic check on the CPU for the tile pipeline (tile_pipeline.hip compiled for the host by tests/emu): every stage
runs with the launch geometry of the product; the scan bytes must equal the reference's.

The speculative chain stage is exercised on purpose: small job sizes give small images many jobs per chain, and warm-ups
too short to converge (0, 16) make settle_chains walk jobs again -- the bytes have to be the reference's either way."""
import os

import numpy as np
import pytest

import common
import emu_bind
import jls_container
import oracle_bind as ob
from charls_amd import synth

FULL = os.environ.get("CHARLS_AMD_QUICK_EMU") != "1"
_SUBSET = {"gray8_64x48", "gray8_w1", "gray8_h1", "gray8_1x1", "tiny_gray12", "tiny_gray16_noise", "tiny_gray2",
           "tiny_gray8_noise", "tiny_rgb8_ilv0", "gray8_maxval100"}
ELIGIBLE = [c for c in common.cases()
            if c["errc"] == 0 and "file" in c and c["near_lossless"] == 0 and c["width"] * c["height"] <= 128 * 128 and
            (c["component_count"] == 1 or c["interleave_mode"] == 0) and (FULL or c["name"] in _SUBSET)]


RUN_JOBS = (8, 8, 24)  # events per job of the run chain, warm-up, long warm-up of the rarer context: tiny, so that small images have many jobs that do not converge


def _encode_planes(planes, width, height, bits, pc, capacity, job=64, warm=32, runs=RUN_JOBS):
    L = emu_bind.tile_lib()
    keep, descs, outs = [], [], []
    for pl in planes:
        pix = np.frombuffer(np.ascontiguousarray(pl).tobytes(), dtype=np.uint8).copy()
        out = np.zeros(capacity, dtype=np.uint8)
        outs.append(out)
        descs.append(emu_bind.make_desc(width, height, 1, 0, bits, 0, 0, pc, 0, pix, width * (1 if bits <= 8 else 2), out, keep))
    arr = (emu_bind.ScanDesc * len(descs))(*descs)
    res = (emu_bind.ScanResult * len(descs))()
    L.emu_encode_tile_pipeline(arr, res, len(descs), job, warm, runs[0], runs[1], runs[2])
    return [(r.errc, r.flags, o[:r.bytes].tobytes()) for r, o in zip(res, outs)]


def _scan_bytes(jls, index=0):
    cont = jls_container.parse(jls)
    return jls[cont.scans[index].data_start:cont.scans[index].data_end]


@pytest.mark.parametrize("c", ELIGIBLE, ids=lambda c: c["name"])
def test_tile_pipeline_matches_reference_scan_bytes(c):
    with open(f"{common.GOLDEN}/{c['file']}", "rb") as f:
        jls = f.read()
    cont = jls_container.parse(jls)
    pc = jls_container.validated_pc(tuple(c["preset"]) if c["preset"] else (0,) * 5, c["bits_per_sample"], 0)
    img = common.case_input(c)
    planes = [img] if img.ndim == 2 else [img[i] for i in range(img.shape[0])]
    size = max(s.data_end - s.data_start for s in cont.scans) + 64
    got = _encode_planes(planes, c["width"], c["height"], c["bits_per_sample"], pc, size, job=256, warm=128)
    for (errc, flags, data), scan in zip(got, cont.scans):
        assert errc == 0
        assert data == jls[scan.data_start:scan.data_end]


@pytest.mark.parametrize("job,warm", [(16, 0), (64, 32), (1024, 1024)])
@pytest.mark.parametrize("kind,bits,w,h,seed", [("mixed", 8, 130, 11, 1), ("zero", 8, 130, 9, 2), ("hard", 12, 65, 9, 4),
                                                ("mixed", 16, 129, 7, 5), ("mixed", 8, 1, 50, 7), ("noise", 16, 40, 12, 9),
                                                ("gradient", 8, 300, 40, 3), ("noise", 8, 96, 20, 11)])
def test_tile_pipeline_batch_of_seeded_frames(kind, bits, w, h, seed, job, warm):
    """Several frames in one launch, every job / warm-up setting: each frame equals the oracle."""
    frames = [synth.frame_numpy(w, h, seed=seed * 10 + f, bits=bits, kind=kind) for f in range(2)]
    pc = jls_container.validated_pc((0,) * 5, bits, 0)
    got = _encode_planes(frames, w, h, bits, pc, w * h * 4 + 1024, job=job, warm=warm)
    for img, (errc, flags, data) in zip(frames, got):
        assert errc == 0
        assert data == _scan_bytes(ob.encode(img, width=w, height=h, bits_per_sample=bits))


def test_tile_pipeline_several_tiles_and_lines_per_tile():
    """Geometries that give several tiles (a tile is up to 8192 samples, at most 16 lines): pieces of a chain from many
    tiles, the look-back of the pack stage across tiles, a last tile with fewer lines."""
    for w, h, kind, bits in [(700, 30, "mixed", 8), (64, 70, "gradient", 8), (3000, 7, "mixed", 8), (8192, 3, "gradient", 8),
                             (513, 40, "hard", 12)]:
        img = synth.frame_numpy(w, h, seed=w + h, bits=bits, kind=kind)
        pc = jls_container.validated_pc((0,) * 5, bits, 0)
        (errc, flags, data), = _encode_planes([img], w, h, bits, pc, w * h * 3 + 1024, job=512, warm=256)
        assert errc == 0 and data == _scan_bytes(ob.encode(img, width=w, height=h, bits_per_sample=bits)), (w, h, kind)


def test_tile_pipeline_long_runs_walk_run_index():
    """Runs that walk RUNindex up to 24 and back (J = 8), runs that end exactly at the line end, runs across chunks."""
    w, h = 3000, 3
    img = np.zeros((h, w), dtype=np.uint8)
    img[1, 2000:] = 9
    img[2, ::2] = 3
    pc = jls_container.validated_pc((0,) * 5, 8, 0)
    (errc, flags, data), = _encode_planes([img], w, h, 8, pc, w * h * 2 + 1024)
    assert errc == 0 and data == _scan_bytes(ob.encode(img, width=w, height=h))


def test_tile_pipeline_runs_of_every_length_at_every_phase():
    """Runs that start and end at every position relative to the 64-sample chunks (run length from the keys that follow)."""
    rng = np.random.default_rng(5)
    w, h = 200, 24
    img = np.full((h, w), 77, dtype=np.uint8)
    for y in range(1, h):
        x = int(rng.integers(0, 70))
        while x < w:
            n = int(rng.integers(1, 140))
            img[y, x:x + n] = int(rng.integers(0, 256))
            x += n + int(rng.integers(0, 3))
    pc = jls_container.validated_pc((0,) * 5, 8, 0)
    (errc, flags, data), = _encode_planes([img], w, h, 8, pc, w * h * 2 + 1024)
    assert errc == 0 and data == _scan_bytes(ob.encode(img, width=w, height=h))


def test_tile_pipeline_destination_too_small_and_knife_edge(monkeypatch):
    img = synth.frame_numpy(64, 64, seed=3, kind="mixed")[:16, :40].copy()
    img = np.ascontiguousarray(np.pad(img, ((0, 48), (0, 24))))
    want = ob.encode(img, width=64, height=64)
    n = len(_scan_bytes(want))
    pc = jls_container.validated_pc((0,) * 5, 8, 0)
    for stuffing in ("1", "0"):
        monkeypatch.setenv("CHARLS_AMD_BLOCK_STUFFING", stuffing)
        (errc, flags, data), = _encode_planes([img], 64, 64, 8, pc, n - 1)
        assert errc == 3
        for slack in (0, 1, 3):
            (errc, flags, data), = _encode_planes([img], 64, 64, 8, pc, n + slack)
            assert errc == 0 and flags == 2  # the host re-runs the exact serial kernel for these
        (errc, flags, data), = _encode_planes([img], 64, 64, 8, pc, n + 4)
        assert errc == 0 and flags == 0 and len(data) == n


def test_tile_pipeline_raw_stream_overflows_its_buffer(monkeypatch):
    """A destination far smaller than the unstuffed stream (stuff_resolve must not walk chunk tables that
    were never made): destination_too_small with both forms of the stuffing stage, and nothing written out of bounds."""
    img = synth.frame_numpy(96, 64, seed=8, bits=8, kind="noise")
    pc = jls_container.validated_pc((0,) * 5, 8, 0)
    for stuffing in ("1", "0"):
        monkeypatch.setenv("CHARLS_AMD_BLOCK_STUFFING", stuffing)
        (errc, flags, data), = _encode_planes([img], 96, 64, 8, pc, 700)
        assert errc == 3


@pytest.mark.parametrize("bits,reset", [(8, 3), (8, 31), (8, 64), (8, 255), (16, 256), (16, 257), (8, 4), (16, 258), (16, 300)])
def test_tile_pipeline_reset_values_with_long_chains(bits, reset):
    """Few contexts, chains of thousands of events: jobs start at every phase of the halving cycle (N is a closed form of
    the event index; RESET is stored through a uint8 by the reference, so 256/257/258 behave as 0/1/2 -- SURVEY F8)."""
    w, h = 96, 48
    rng = np.random.default_rng(reset)
    base = (np.arange(w)[None, :] // 7 + np.arange(h)[:, None] // 5) * (3 if bits == 8 else 700)
    noise = rng.integers(-2, 3, size=(h, w)) * (1 if bits == 8 else 900)
    img = np.clip(base + noise + (40 if bits == 8 else 9000), 0, (1 << bits) - 1).astype(np.uint8 if bits == 8 else np.uint16)
    preset = (0, 0, 0, 0, reset)
    pc = jls_container.validated_pc(preset, bits, 0)
    want = _scan_bytes(ob.encode(img, width=w, height=h, bits_per_sample=bits, preset=preset))
    for job, warm in ((48, 16), (80, 200)):
        (errc, flags, data), = _encode_planes([img], w, h, bits, pc, w * h * 4 + 1024, job=job, warm=warm)
        assert errc == 0 and data == want, (job, warm)


def test_tile_pipeline_prediction_clipped_at_both_ends_of_wide_samples():
    """Samples wider than 8 bits keep only the distance of the prediction from the nearer end of the range in their record:
    images that sit at 0 and at MAXVAL with a bias that pushes the prediction over the end."""
    rng = np.random.default_rng(17)
    for bits in (9, 12, 16):
        maxval = (1 << bits) - 1
        w, h = 80, 40
        img = np.zeros((h, w), dtype=np.uint16)
        img[:, : w // 2] = (rng.integers(0, 40, size=(h, w // 2)) ** 2 // 30).astype(np.uint16)              # hugging 0, skewed
        img[:, w // 2:] = (maxval - rng.integers(0, 40, size=(h, w - w // 2)) ** 2 // 30).astype(np.uint16)  # hugging MAXVAL
        pc = jls_container.validated_pc((0,) * 5, bits, 0)
        (errc, flags, data), = _encode_planes([img], w, h, bits, pc, w * h * 5 + 1024, job=64, warm=64)
        assert errc == 0 and data == _scan_bytes(ob.encode(img, width=w, height=h, bits_per_sample=bits)), bits


def _encode_line_interleaved(img, width, height, comps, bits, xform, capacity, job=64, warm=32):
    L = emu_bind.tile_lib()
    keep = []
    pix = np.frombuffer(np.ascontiguousarray(img).tobytes(), dtype=np.uint8).copy()
    out = np.zeros(capacity, dtype=np.uint8)
    pc = jls_container.validated_pc((0,) * 5, bits, 0)
    d = emu_bind.make_desc(width, height, comps, 1, bits, 0, xform, pc, 0, pix, width * comps * (1 if bits <= 8 else 2), out, keep)
    res = (emu_bind.ScanResult * 1)()
    L.emu_encode_tile_pipeline((emu_bind.ScanDesc * 1)(d), res, 1, job, warm, 16, 8, 40)
    return res[0].errc, res[0].flags, out[:res[0].bytes].tobytes()


def test_tile_pipeline_line_interleaved_scans():
    """ILV_LINE lossless: the components of a pixel row are coded as lines of their own, sharing the contexts but each
    with its own RUNindex."""
    rng = np.random.default_rng(21)
    cases = [("mixed", 8, 3, 0, 33, 7), ("mixed", 8, 3, 1, 20, 5), ("mixed", 16, 3, 3, 12, 4), ("hard", 12, 2, 0, 19, 5),
             ("mixed", 8, 4, 0, 9, 4), ("mixed", 8, 3, 1, 700, 9)]
    for kind, bits, comps, xform, w, h in cases:
        img = np.stack([synth.frame_numpy(w, h, seed=80 + c, bits=bits, kind=kind) for c in range(comps)], axis=-1)
        want = ob.encode(img, width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=1,
                         color_transformation=xform)
        errc, flags, data = _encode_line_interleaved(img, w, h, comps, bits, xform, w * h * comps * 4 + 1024)
        assert errc == 0 and data == _scan_bytes(want), (kind, bits, comps, xform)
    for trial in range(24 if FULL else 10):  # few grey levels: runs in every component, RUNindex walks per component
        comps = int(rng.choice([2, 3, 4]))
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 6))
        img = (rng.integers(0, 2, size=(h, w, comps)) * 200).astype(np.uint8)
        img[:, w // 3:, 0] = img[:, w // 3:w // 3 + 1, 0]
        xform = int(rng.integers(0, 4)) if comps == 3 else 0
        want = ob.encode(img, width=w, height=h, component_count=comps, interleave_mode=1, color_transformation=xform)
        errc, flags, data = _encode_line_interleaved(img, w, h, comps, 8, xform, w * h * comps * 4 + 1024)
        assert errc == 0 and data == _scan_bytes(want), (trial, w, h, comps, xform)


@pytest.mark.parametrize("chunk", range(3))
def test_tile_pipeline_random_parameters(chunk):
    """Random lossless parameter sets (bits 2..16, custom thresholds / RESET, odd sizes, planar and line-interleaved)."""
    from test_oracle_vs_reference import _image
    L = emu_bind.tile_lib()
    rng = np.random.default_rng(900 + chunk)
    for it in range(30):
        bits = int(rng.integers(2, 17))
        comps = int(rng.choice([1, 1, 1, 2, 3, 4]))
        ilv = 0 if comps == 1 else 1
        w, h = int(rng.choice([1, 2, 5, 17, 64, 65, 130, 300])), int(rng.choice([1, 2, 3, 8, 21, 40]))
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
        pc = jls_container.validated_pc(preset, bits, 0)
        keep = []
        pix = np.frombuffer(np.ascontiguousarray(img).tobytes(), dtype=np.uint8).copy()
        out = np.zeros(w * h * comps * 5 + 1024, dtype=np.uint8)
        bps = 1 if bits <= 8 else 2
        d = emu_bind.make_desc(w, h, comps, ilv, bits, 0, xform, pc, 0, pix, w * comps * bps, out, keep)
        res = (emu_bind.ScanResult * 1)()
        job = int(rng.choice([16, 32, 64, 256, 1024]))
        warm = int(rng.choice([0, 16, 64, 1024]))
        run_job = int(rng.choice([8, 16, 64, 2048]))
        run_warm = int(rng.choice([0, 8, 64, 2048]))
        run_long = int(rng.choice([0, 16, 256]))
        L.emu_encode_tile_pipeline((emu_bind.ScanDesc * 1)(d), res, 1, job, warm, run_job, run_warm, run_long)
        tag = (chunk, it, bits, comps, ilv, w, h, kind, preset, xform, job, warm, run_job, run_warm, run_long)
        assert res[0].errc == 0 and out[:res[0].bytes].tobytes() == _scan_bytes(want), tag


def test_tile_pipeline_scans_shorter_than_the_launch_geometry():
    """The last restart interval of a frame has fewer lines than the intervals it shares a launch with (the launch is sized
    for the first scan): fewer tiles, a last tile with fewer lines, the scan's result still the reference's bytes."""
    L = emu_bind.tile_lib()
    w = 200
    pc = jls_container.validated_pc((0,) * 5, 8, 0)
    for heights in ((64, 22, 1), (90, 41, 40)):  # lines per tile = 16 for this width
        keep, descs, outs, imgs = [], [], [], []
        for i, h in enumerate(heights):
            img = synth.frame_numpy(w, h, seed=50 + i, bits=8, kind="mixed")
            pix = np.frombuffer(np.ascontiguousarray(img).tobytes(), dtype=np.uint8).copy()
            out = np.zeros(w * h * 3 + 1024, dtype=np.uint8)
            imgs.append(img)
            outs.append(out)
            descs.append(emu_bind.make_desc(w, h, 1, 0, 8, 0, 0, pc, 0, pix, w, out, keep))
        arr = (emu_bind.ScanDesc * len(descs))(*descs)
        res = (emu_bind.ScanResult * len(descs))()
        L.emu_encode_tile_pipeline(arr, res, len(descs), 128, 64, 16, 16, 32)
        for img, r, o, h in zip(imgs, res, outs, heights):
            assert r.errc == 0 and o[:r.bytes].tobytes() == _scan_bytes(ob.encode(img, width=w, height=h)), heights


@pytest.mark.parametrize("run_job,run_warm,run_long", [(8, 0, 0), (8, 8, 64), (16, 64, 16), (64, 16, 512), (2048, 2048, 32768)])
def test_tile_pipeline_run_chain_in_jobs(run_job, run_warm, run_long):
    """Images that are mostly runs (thousands of run events, both interruption types, RUNindex wandering): the run chain cut
    into jobs of every size, with warm-ups that converge and warm-ups that do not."""
    rng = np.random.default_rng(31)
    w, h = 160, 48
    img = np.full((h, w), 120, dtype=np.uint8)
    for y in range(h):
        x = 0
        while x < w:
            n = int(rng.choice([1, 1, 2, 3, 5, 9, 17, 40]))
            img[y, x:x + n] = 120 + int(rng.integers(-2, 3)) * int(rng.integers(0, 2))
            x += n
    img[10:14] = 7  # whole lines in one run
    pc = jls_container.validated_pc((0,) * 5, 8, 0)
    (errc, flags, data), = _encode_planes([img], w, h, 8, pc, w * h * 2 + 1024, runs=(run_job, run_warm, run_long))
    assert errc == 0 and data == _scan_bytes(ob.encode(img, width=w, height=h))
    wide = (img.astype(np.uint16) * 200)  # 16-bit records: Errval takes 16 bits of the run record
    pc16 = jls_container.validated_pc((0,) * 5, 16, 0)
    (errc, flags, data), = _encode_planes([wide], w, h, 16, pc16, w * h * 4 + 1024, runs=(run_job, run_warm, run_long))
    assert errc == 0 and data == _scan_bytes(ob.encode(wide, width=w, height=h, bits_per_sample=16))


def test_tile_pipeline_counts_the_jobs_it_had_to_walk_again():
    """The counters behind charls_amd_speculation_counters: with no warm-up every job guesses the initial state, so jobs are
    walked again; with a warm-up longer than the chains nothing is."""
    import ctypes as C
    L = emu_bind.tile_lib()
    img = synth.frame_numpy(300, 40, seed=3, kind="gradient")
    pc = jls_container.validated_pc((0,) * 5, 8, 0)
    want = _scan_bytes(ob.encode(img, width=300, height=40))
    out = (C.c_uint32 * 8)()
    (errc, flags, data), = _encode_planes([img], 300, 40, 8, pc, 300 * 40 * 2 + 1024, job=16, warm=0, runs=(32, 0, 0))
    assert errc == 0 and data == want
    L.emu_tile_counters(out)
    assert out[0] > 100 and out[1] > 10 and out[2] >= 1
    (errc, flags, data), = _encode_planes([img], 300, 40, 8, pc, 300 * 40 * 2 + 1024, job=1024, warm=1 << 20, runs=(2048, 2048, 32768))
    assert errc == 0 and data == want
    L.emu_tile_counters(out)
    assert out[0] > 0 and out[1] == 0 and out[3] == 0


@pytest.mark.parametrize("rare_warm,serial", [(512, 0), (16, 1), (0, 1)])
def test_rarer_run_context_in_segments(rare_warm, serial):
    """The exact walk of the rarer run-interruption context goes in segments of its event list, a lane each, from a warm-up
    counted in ITS events; a segment that starts from a wrong guess makes lane 0 walk the list again.  A frame with some 600
    events of the rarer type: five segments, the oracle's bytes with a warm-up that converges and with ones that cannot, and
    the counters say which of the two happened."""
    import ctypes as C
    L = emu_bind.tile_lib()
    w, h = 700, 300
    img = synth.frame_numpy(w, h, seed=5, kind="mixed")
    pc = jls_container.validated_pc((0,) * 5, 8, 0)
    want = _scan_bytes(ob.encode(img, width=w, height=h))
    (errc, flags, data), = _encode_planes([img], w, h, 8, pc, w * h * 2 + 1024, job=256, warm=256, runs=(64, 512, rare_warm))
    assert errc == 0 and data == want
    out = (C.c_uint32 * 8)()
    L.emu_tile_counters(out)
    assert out[4] >= 3 and out[5] == serial, list(out)


# ---- pixel mode (tile_pixel_mode.hip): sample-interleaved scans, and lines cut into segment tiles ---------------------

def _encode_scan(img, width, height, comps, ilv, bits, ct=0, capacity=None, job=64, warm=32, runs=RUN_JOBS):
    """One scan of any interleave mode through the emulated tile pipeline; returns (errc, flags, bytes)."""
    L = emu_bind.tile_lib()
    keep = []
    pc = jls_container.validated_pc((0,) * 5, bits, 0)
    pix = np.frombuffer(np.ascontiguousarray(img).tobytes(), dtype=np.uint8).copy()
    nbytes = 1 if bits <= 8 else 2
    stride = width * nbytes * (comps if ilv != 0 else 1)
    out = np.zeros(capacity or (width * height * comps * nbytes * 2 + 1024), dtype=np.uint8)
    desc = emu_bind.make_desc(width, height, comps, ilv, bits, 0, ct, pc, 0, pix, stride, out, keep)
    arr = (emu_bind.ScanDesc * 1)(desc)
    res = (emu_bind.ScanResult * 1)()
    L.emu_encode_tile_pipeline(arr, res, 1, job, warm, runs[0], runs[1], runs[2])
    return res[0].errc, res[0].flags, out[:res[0].bytes].tobytes()


def _rgb(w, h, seed, bits=8, comps=3, kind="mixed", flat=0.0):
    img = synth.frame_numpy(w, h, seed=seed, bits=bits, components=comps, kind=kind, interleaved=True)
    if flat:
        rng = np.random.default_rng(seed)
        # flat patches common to ALL components: run mode of a sample-interleaved scan needs every component to match
        for _ in range(int(flat * h)):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            n = int(rng.integers(2, max(3, w // 2)))
            img[y, x:x + n, :] = img[y, x, :]
            if y + 1 < h:
                img[y + 1, x:x + n, :] = img[y, x, :]
    return img


@pytest.mark.parametrize("comps,bits,ct,w,h,kind", [(3, 8, 0, 130, 11, "mixed"), (3, 8, 1, 67, 19, "mixed"), (3, 8, 2, 200, 9, "hard"),
                                                     (3, 8, 3, 64, 33, "gradient"), (2, 8, 0, 129, 7, "mixed"), (4, 8, 0, 65, 12, "mixed"),
                                                     (3, 16, 0, 70, 9, "mixed"), (3, 16, 1, 66, 8, "hard"), (3, 12, 0, 90, 10, "mixed"),
                                                     (3, 8, 0, 1, 20, "mixed"), (3, 8, 0, 50, 1, "mixed"), (3, 8, 0, 96, 14, "zero"),
                                                     (4, 16, 0, 33, 6, "noise"), (3, 5, 0, 40, 10, "noise")])
@pytest.mark.parametrize("cap", [None, "256"])
def test_pixel_mode_sample_interleaved_matches_oracle(monkeypatch, comps, bits, ct, w, h, kind, cap):
    """ILV_SAMPLE scans, whole-line tiles and (cap = 256 samples per tile) lines cut into segment tiles."""
    if cap:
        monkeypatch.setenv("CHARLS_AMD_TILE_SAMPLES", cap)
    img = _rgb(w, h, seed=w + h + comps, bits=bits, comps=comps, kind=kind, flat=0.5)
    kw = dict(width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=2, color_transformation=ct)
    want = _scan_bytes(ob.encode(img, **kw))
    errc, flags, data = _encode_scan(img, w, h, comps, 2, bits, ct)
    assert errc == 0 and data == want


@pytest.mark.parametrize("job,warm,runs", [(16, 0, (32, 0, 0)), (64, 32, RUN_JOBS), (1024, 1024, (2048, 2048, 32768))])
@pytest.mark.parametrize("cap", [None, "192"])
def test_pixel_mode_speculation_settings(monkeypatch, job, warm, runs, cap):
    if cap:
        monkeypatch.setenv("CHARLS_AMD_TILE_SAMPLES", cap)
    w, h = 150, 40
    img = _rgb(w, h, seed=5, flat=1.5)
    kw = dict(width=w, height=h, component_count=3, interleave_mode=2)
    errc, flags, data = _encode_scan(img, w, h, 3, 2, 8, job=job, warm=warm, runs=runs)
    assert errc == 0 and data == _scan_bytes(ob.encode(img, **kw))


def test_pixel_mode_long_runs_across_segments(monkeypatch):
    """Runs that leave their segment tile (their length is counted on from the keys of the tiles that follow), runs that end
    with the line, a flat image (every tile looks back to the start of its line for its run state)."""
    monkeypatch.setenv("CHARLS_AMD_TILE_SAMPLES", "192")
    w, h = 700, 6
    img = np.zeros((h, w, 3), dtype=np.uint8)
    img[1, 100:, :] = (9, 9, 200)
    img[2, ::7, 1] = 3
    img[3, 300:650, :] = (1, 2, 3)
    img[4, 5:, 0] = 77
    kw = dict(width=w, height=h, component_count=3, interleave_mode=2)
    errc, flags, data = _encode_scan(img, w, h, 3, 2, 8)
    assert errc == 0 and data == _scan_bytes(ob.encode(img, **kw))
    flat = np.full((4, 1000, 3), 41, dtype=np.uint8)
    errc, flags, data = _encode_scan(flat, 1000, 4, 3, 2, 8)
    assert errc == 0 and data == _scan_bytes(ob.encode(flat, width=1000, height=4, component_count=3, interleave_mode=2))


@pytest.mark.parametrize("bits,w,h,kind,cap", [(8, 700, 9, "mixed", "256"), (8, 1000, 5, "zero", "128"), (16, 300, 8, "mixed", "64"),
                                               (12, 513, 7, "hard", "192"), (8, 9000, 2, "mixed", None), (16, 5000, 2, "mixed", None),
                                               (8, 130, 11, "mixed", "PIXEL")])
def test_pixel_mode_wide_planar_lines(monkeypatch, bits, w, h, kind, cap):
    """Single-component lines wider than a tile (cut into segment tiles; two run-interruption contexts), and -- PIXEL -- an
    ordinary scan sent through pixel mode with whole-line tiles."""
    if cap == "PIXEL":
        monkeypatch.setenv("CHARLS_AMD_PIXEL_MODE", "1")
    elif cap:
        monkeypatch.setenv("CHARLS_AMD_TILE_SAMPLES", cap)
    img = synth.frame_numpy(w, h, seed=w, bits=bits, kind=kind)
    if kind == "mixed":
        img[h // 2, w // 3:] = img[h // 2, w // 3]  # a run to the end of the line, across segments
        img[h - 1, 10:w - 10] = img[h - 1, 10]
    errc, flags, data = _encode_scan(img, w, h, 1, 0, bits)
    assert errc == 0 and data == _scan_bytes(ob.encode(img, width=w, height=h, bits_per_sample=bits))


@pytest.mark.parametrize("bits,ct,cap", [(8, 0, "128"), (8, 1, "256"), (16, 2, "64"), (8, 0, "PIXEL")])
def test_pixel_mode_wide_line_interleaved(monkeypatch, bits, ct, cap):
    if cap == "PIXEL":
        monkeypatch.setenv("CHARLS_AMD_PIXEL_MODE", "1")
    else:
        monkeypatch.setenv("CHARLS_AMD_TILE_SAMPLES", cap)
    w, h = 333, 7
    img = _rgb(w, h, seed=bits + ct, bits=bits, flat=1.0)
    kw = dict(width=w, height=h, bits_per_sample=bits, component_count=3, interleave_mode=1, color_transformation=ct)
    errc, flags, data = _encode_scan(img, w, h, 3, 1, bits, ct)
    assert errc == 0 and data == _scan_bytes(ob.encode(img, **kw))


@pytest.mark.parametrize("comps,ilv,bits", [(3, 2, 12), (3, 2, 5), (1, 0, 10), (1, 0, 3)])
def test_pixel_mode_masks_bits_above_the_sample_precision(monkeypatch, comps, ilv, bits):
    """Source samples with garbage above bits_per_sample (src/copy_to_line_buffer.hpp masks them; test/jpegls_encoder_test.cpp:1577-1755)."""
    monkeypatch.setenv("CHARLS_AMD_PIXEL_MODE", "1")
    monkeypatch.setenv("CHARLS_AMD_TILE_SAMPLES", "192")
    w, h = 211, 9
    rng = np.random.default_rng(bits)
    clean = _rgb(w, h, seed=bits, bits=bits, comps=comps, flat=1.0) if comps > 1 else synth.frame_numpy(w, h, seed=bits, bits=bits, kind="mixed")
    full = 8 if bits <= 8 else 16
    dirty = (clean.astype(np.uint32) | (rng.integers(0, 1 << (full - bits), size=clean.shape).astype(np.uint32) << bits)).astype(clean.dtype)
    kw = dict(width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=ilv)
    want = ob.encode(dirty, **kw)
    assert want == ob.encode(clean, **kw)
    errc, flags, data = _encode_scan(dirty, w, h, comps, ilv, bits)
    assert errc == 0 and data == _scan_bytes(want)


@pytest.mark.parametrize("bits,ct,w,h,comps", [(8, 0, 700, 14, 3), (8, 1, 130, 40, 3), (16, 3, 129, 9, 3), (12, 0, 65, 12, 4), (8, 0, 64, 33, 2),
                                               (8, 2, 3000, 5, 3)])
def test_line_interleaved_scans_whole_line_tiles(bits, ct, w, h, comps):
    """ILV_LINE scans take pixel mode with tiles of several coded lines (the line above a coded line is `components` lines up)."""
    img = _rgb(w, h, seed=w + bits, bits=bits, comps=comps, flat=1.0)
    kw = dict(width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=1, color_transformation=ct)
    for job, warm in ((32, 0), (256, 128)):
        errc, flags, data = _encode_scan(img, w, h, comps, 1, bits, ct, job=job, warm=warm)
        assert errc == 0 and data == _scan_bytes(ob.encode(img, **kw)), (job, warm)
