"""Pixel mode of the tile pipeline on the MI355X (tile_pixel_mode.hip): sample-interleaved
scans and lines wider than a tile, bytes against the oracle through the C ABI.  GPU only."""
import ctypes as C

import numpy as np
import pytest

import oracle_bind as ob
from charls_amd import batch, capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    L = capi.load_product()
    assert L.lib.charls_amd_device_status() == 0
    L.lib.charls_amd_speculation_counters.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
    L.lib.charls_amd_speculation_counters.restype = C.c_int32
    return L


def _jobs(lib):
    out = (C.c_uint64 * 4)()
    lib.lib.charls_amd_speculation_counters(out, 4)
    return int(out[0])


def _rgb(w, h, seed, bits=8, comps=3, kind="mixed"):
    img = synth.frame_numpy(w, h, seed=seed, bits=bits, components=comps, kind=kind, interleaved=True)
    rng = np.random.default_rng(seed)
    for _ in range(h // 2):  # flat patches common to all components: runs of a sample-interleaved scan
        y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
        n = int(rng.integers(2, max(3, w // 2)))
        img[y, x:x + n, :] = img[y, x, :]
        if y + 1 < h:
            img[y + 1, x:x + n, :] = img[y, x, :]
    return img


@pytest.mark.parametrize("w,h,comps,bits,ct", [(512, 512, 3, 8, 0), (1024, 768, 3, 8, 1), (4096, 64, 3, 8, 2), (3000, 40, 3, 8, 3),
                                               (700, 300, 2, 8, 0), (2100, 50, 4, 8, 0), (1024, 256, 3, 16, 0), (2048, 32, 3, 16, 1),
                                               (640, 480, 3, 12, 0), (5000, 16, 4, 16, 0)])
def test_sample_interleaved_scans_run_on_the_tile_pipeline(lib, w, h, comps, bits, ct):
    img = _rgb(w, h, seed=w + h, bits=bits, comps=comps)
    kw = dict(width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=2, color_transformation=ct)
    before = _jobs(lib)
    got = lib.encode(img, **kw)
    assert _jobs(lib) > before, "not coded by the tile pipeline"
    assert got == ob.encode(img, **kw)
    assert lib.decode(got)[1].tobytes() == img.tobytes()


@pytest.mark.parametrize("w,h,bits", [(16384, 64, 8), (8193, 33, 8), (65535, 8, 8), (100000, 3, 8), (8192, 48, 16), (4097, 40, 12),
                                      (40000, 4, 16)])
def test_lines_wider_than_a_tile_gray(lib, w, h, bits):
    img = synth.frame_numpy(w, h, seed=w, bits=bits, kind="mixed")
    img[h // 2, w // 3:] = img[h // 2, w // 3]    # a run that crosses every segment up to the end of its line
    img[h - 1, 10:w - 10] = img[h - 1, 10]          # and one that is interrupted in the last segment
    kw = dict(width=w, height=h, bits_per_sample=bits)
    before = _jobs(lib)
    got = lib.encode(img, **kw)
    assert _jobs(lib) > before, "not coded by the tile pipeline"
    assert got == ob.encode(img, **kw)
    assert lib.decode(got)[1].tobytes() == img.tobytes()


@pytest.mark.parametrize("w,h,bits,ct", [(9000, 24, 8, 1), (4200, 20, 16, 0)])
def test_lines_wider_than_a_tile_line_interleaved(lib, w, h, bits, ct):
    img = _rgb(w, h, seed=w, bits=bits)
    kw = dict(width=w, height=h, bits_per_sample=bits, component_count=3, interleave_mode=1, color_transformation=ct)
    before = _jobs(lib)
    got = lib.encode(img, **kw)
    assert _jobs(lib) > before
    assert got == ob.encode(img, **kw)


def test_flat_wide_frames_every_tile_looks_back(lib):
    for comps, ilv in ((1, 0), (3, 2)):
        w, h = 20000, 6
        img = np.full((h, w, comps) if comps > 1 else (h, w), 200, dtype=np.uint8)
        kw = dict(width=w, height=h, component_count=comps, interleave_mode=ilv)
        assert lib.encode(img, **kw) == ob.encode(img, **kw)


def test_forced_speculation_in_pixel_mode(lib, knobs):
    for k, v in {"CHARLS_AMD_JOB_EVENTS": "16", "CHARLS_AMD_WARM_EVENTS": "0", "CHARLS_AMD_RUN_JOB_EVENTS": "32",
                 "CHARLS_AMD_RUN_WARM_EVENTS": "0"}.items():
        knobs.set(k, v)
    for w, h, comps, ilv in ((1024, 512, 3, 2), (12000, 40, 1, 0)):
        img = _rgb(w, h, seed=9, comps=comps) if comps > 1 else synth.frame_numpy(w, h, seed=9, kind="mixed")
        kw = dict(width=w, height=h, component_count=comps, interleave_mode=ilv)
        out = (C.c_uint64 * 4)()
        lib.lib.charls_amd_speculation_counters(out, 4)
        before = list(out)
        got = lib.encode(img, **kw)
        lib.lib.charls_amd_speculation_counters(out, 4)
        assert out[1] > before[1] and out[3] > before[3]
        assert got == ob.encode(img, **kw)


def test_batch_of_sample_interleaved_frames(lib):
    import torch
    w, h, n = 1536, 1024, 5
    host = [_rgb(w, h, seed=80 + i) for i in range(n)]
    frames = torch.from_numpy(np.stack(host)).cuda()
    enc = batch.encode_batch(frames, component_count=3, interleave_mode=2, color_transformation=1, lib=lib)
    got = enc.streams.cpu().numpy()
    for i, img in enumerate(host):
        want = ob.encode(img, width=w, height=h, component_count=3, interleave_mode=2, color_transformation=1)
        assert enc.errcs[i] == 0 and got[i, :int(enc.sizes[i])].tobytes() == want, i
    out = torch.empty_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out, lib=lib)
    assert (errcs == 0).all() and torch.equal(out, frames)


@pytest.mark.parametrize("chunk", range(3))
def test_random_lossless_scans_with_small_tiles(lib, knobs, chunk):
    """Random lossless scans of every interleave mode with tiles of 64 - 512 samples (CHARLS_AMD_TILE_SAMPLES, read per call):
    lines cut into segments at every phase, look-backs, runs across segments, tiles of one to sixteen lines -- and, every
    third case, every scan through pixel mode.  Bytes against the oracle."""
    rng = np.random.default_rng(900 + chunk)
    for it in range(40):
        bits = int(rng.choice([8, 8, 8, 12, 16, 5, 2]))
        comps = int(rng.choice([1, 1, 3, 3, 2, 4]))
        ilv = 0 if comps == 1 else int(rng.integers(1, 3))
        w = int(rng.choice([1, 2, 63, 64, 65, 130, 257, 700, 1500]))
        h = int(rng.choice([1, 2, 3, 9, 20]))
        ct = int(rng.integers(1, 4)) if (comps == 3 and bits in (8, 16) and rng.random() < 0.5) else 0
        kind = str(rng.choice(["mixed", "gradient", "hard", "zero", "noise"]))
        knobs.set("TILE_SAMPLES", int(rng.choice([64, 128, 192, 256, 512])))
        if it % 3 == 0:
            knobs.set("PIXEL_MODE", 1)
        else:
            knobs.clear("PIXEL_MODE")
        img = synth.frame_numpy(w, h, seed=1000 * chunk + it, bits=bits, components=comps, kind=kind, interleaved=True)
        if kind == "mixed" and h > 1:  # flat stretches common to all components, across segment boundaries
            x0 = int(rng.integers(0, w))
            img[h // 2, x0:] = img[h // 2, x0]
        kw = dict(width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=ilv, color_transformation=ct)
        want = ob.encode(img, destination_size=4 * img.nbytes + 4096, **kw)
        assert lib.encode(img, destination_size=4 * img.nbytes + 4096, **kw) == want, (kw, kind, it)
