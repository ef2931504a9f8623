"""Live cross-check oracle <-> reference (oracle/_ref/libcharls_ref.so), randomized parameters.  CPU only.
Skipped where the reference build is absent; the committed golden vectors then carry the pin."""
import os

import numpy as np
import pytest

import oracle_bind as ob
from charls_amd import synth
from charls_amd.capi import CharLSLibrary, JpegLSError

pytestmark = pytest.mark.skipif(not os.path.exists(ob.REF_LIB), reason="oracle/_ref not built (make -C oracle ref)")


def _image(rng, w, h, bits, comps, ilv, kind, seed):
    dt = np.uint8 if bits <= 8 else np.uint16
    shape = (h, w) if comps == 1 else ((comps, h, w) if ilv == 0 else (h, w, comps))
    if kind == "rand":
        return rng.integers(0, 1 << bits, size=shape).astype(dt)
    if kind == "smooth":
        base = int(rng.integers(0, 1 << bits))
        axis = -1 if (comps == 1 or ilv == 0) else 1
        return np.clip(base + rng.integers(-2, 3, size=shape).cumsum(axis=axis) // 3, 0, (1 << bits) - 1).astype(dt)
    return synth.frame_numpy(w, h, seed=seed, bits=bits, components=comps, kind=kind, interleaved=(ilv != 0))


@pytest.mark.parametrize("chunk", range(8))
def test_random_parameters(chunk):
    ref = CharLSLibrary(ob.REF_LIB)
    rng = np.random.default_rng(1000 + chunk)
    for it in range(150):
        bits = int(rng.integers(2, 17))
        comps = int(rng.choice([1, 1, 1, 2, 3, 3, 4]))
        ilv = 0 if comps == 1 else int(rng.integers(0, 3))
        w = int(rng.choice([1, 2, 3, 5, 17, 64, 100, 257]))
        h = int(rng.choice([1, 2, 3, 8, 33, 64]))
        maxval = (1 << bits) - 1
        near = 0 if rng.random() < 0.5 else int(rng.integers(0, min(255, maxval // 2) + 1))
        ct = 0
        if comps == 3 and bits in (8, 16) and near == 0 and ilv != 0 and rng.random() < 0.5:
            ct = int(rng.integers(1, 4))
        kind = str(rng.choice(["rand", "smooth", "gradient", "mixed", "zero", "hard"]))
        preset = None
        if rng.random() < 0.25:
            mv = int(rng.integers(max(1, 2 * near), maxval + 1)) if rng.random() < 0.5 else 0
            mvv = mv or maxval
            if near <= min(255, mvv // 2):
                t1 = int(rng.integers(near + 1, mvv + 1))
                t2 = int(rng.integers(t1, mvv + 1))
                t3 = int(rng.integers(t2, mvv + 1))
                preset = (mv, t1, t2, t3, int(rng.integers(3, max(255, mvv) + 1)))
        img = _image(rng, w, h, bits, comps, ilv, kind, it)
        kw = dict(width=w, height=h, bits_per_sample=bits, component_count=comps, near_lossless=near,
                  interleave_mode=ilv, color_transformation=ct, preset=preset)
        try:
            a, ea = ref.encode(img, **kw), 0
        except JpegLSError as e:
            a, ea = None, e.errc
        try:
            b, eb = ob.encode(img, **kw), 0
        except ob.OracleError as e:
            b, eb = None, e.errc
        assert (ea, a) == (eb, b), kw
        if a is None:
            continue
        _, pa = ref.decode(a)
        _, pb = ob.decode(a)
        assert np.array_equal(pa, pb), kw


def test_destination_too_small_boundary():
    """src/scan_encoder.hpp:119-120: the encoder needs >= 4 free bytes at every flush."""
    ref = CharLSLibrary(ob.REF_LIB)
    img = synth.frame_numpy(64, 64, seed=3, kind="mixed")
    full = ref.encode(img, width=64, height=64)
    for cap in range(len(full) - 8, len(full) + 8):
        try:
            a, ea = ref.encode(img, width=64, height=64, destination_size=cap), 0
        except JpegLSError as e:
            a, ea = None, e.errc
        try:
            b, eb = ob.encode(img, width=64, height=64, destination_size=cap), 0
        except ob.OracleError as e:
            b, eb = None, e.errc
        assert (ea, a) == (eb, b), cap
