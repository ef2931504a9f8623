"""Randomised parity sweep on the GPU: the parameter generator of tests/test_oracle_vs_reference.py (which pins the oracle
to the reference on the CPU) drives the product through the C ABI; encoded bytes / error codes and decoded pixels must
equal the oracle's.  Covers every dispatch of the engine: pipeline and exact encoders, fast / wave / serial decoders."""
import numpy as np
import pytest

import oracle_bind as ob
from charls_amd import capi
from charls_amd.capi import JpegLSError
from test_oracle_vs_reference import _image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    L = capi.load_product()
    assert L.lib.charls_amd_device_status() == 0
    return L


@pytest.mark.parametrize("chunk", range(4))
def test_random_parameters_match_the_oracle(lib, chunk):
    rng = np.random.default_rng(5000 + chunk)
    for it in range(60):
        bits = int(rng.integers(2, 17))
        comps = int(rng.choice([1, 1, 1, 2, 3, 3, 4]))
        ilv = 0 if comps == 1 else int(rng.integers(0, 3))
        w = int(rng.choice([1, 2, 3, 5, 17, 64, 65, 100, 257]))
        h = int(rng.choice([1, 2, 3, 8, 33, 64]))
        maxval = (1 << bits) - 1
        near = 0 if rng.random() < 0.6 else int(rng.integers(0, min(255, maxval // 2) + 1))
        ct = 0
        if comps == 3 and bits in (8, 16) and near == 0 and ilv != 0 and rng.random() < 0.5:
            ct = int(rng.integers(1, 4))
        kind = str(rng.choice(["rand", "smooth", "gradient", "mixed", "zero", "hard"]))
        preset = None
        if rng.random() < 0.3:
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
            want, ew = ob.encode(img, **kw), 0
        except ob.OracleError as e:
            want, ew = None, e.errc
        try:
            got, eg = lib.encode(img, **kw), 0
        except JpegLSError as e:
            got, eg = None, e.errc
        assert (eg, got) == (ew, want), kw
        if want is None:
            continue
        assert lib.decode(want)[1].tobytes() == ob.decode(want)[1].tobytes(), kw
