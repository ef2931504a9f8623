"""Parity tests proper: the HIP path, called through the C ABI, against the golden vectors and the oracle.  GPU only."""
import numpy as np
import pytest

import common
import oracle_bind as ob
from charls_amd import capi, synth
from charls_amd.capi import JpegLSError

pytestmark = pytest.mark.gpu

CASES = common.cases()
SMALL = [c for c in CASES if c["width"] * c["height"] <= 512 * 512]


@pytest.fixture(scope="module")
def lib():
    L = capi.load_product()
    assert L.lib.charls_amd_device_status() == 0, "GPU box without a usable device: the product must not fall back"
    return L


@pytest.mark.parametrize("c", SMALL, ids=lambda c: c["name"])
def test_encode_bytes_equal_reference(lib, c):
    img = common.case_input(c)
    if c["errc"] != 0:
        with pytest.raises(JpegLSError) as e:
            lib.encode(img, **common.case_kwargs(c))
        assert e.value.errc == c["errc"]
        return
    jls = lib.encode(img, **common.case_kwargs(c))
    assert (len(jls), common.sha(jls)) == (c["jls_size"], c["jls_sha256"])


@pytest.mark.parametrize("c", [c for c in SMALL if c["errc"] == 0 and "file" in c], ids=lambda c: c["name"])
def test_decode_pixels_equal_reference(lib, c):
    with open(f"{common.GOLDEN}/{c['file']}", "rb") as f:
        jls = f.read()
    before = capi.engine_counters(lib).get("exact_retry_scans", 0)
    hdr, px = lib.decode(jls)
    assert common.sha(px.tobytes()) == c["decoded_sha256"]
    # a valid stream keeps its speed path: none of its scans may need the exact decoder behind it (round 6: a lane that could
    # not decode a long code faked an overflow of A and sent every 16-bit stream with escape codes there -- bytes right, speed gone)
    assert capi.engine_counters(lib).get("exact_retry_scans", 0) == before


def _fixture_roundtrip(lib, jls_name, pnm_name, ilv, near, reencode, bits=None, preset=None):
    jls = common.refdata(jls_name)
    img, _ = common.read_pnm(pnm_name)
    src = common.planar(img) if (img.ndim == 3 and ilv == 0) else img
    hdr, px = lib.decode(jls)
    if near == 0:
        assert px.tobytes() == src.tobytes()
    else:
        d = np.abs(px.view(src.dtype).astype(np.int64) - src.ravel().astype(np.int64))
        assert d.max() <= near  # tolerance = NEAR, per ISO 14495-1
    if reencode:
        out = lib.encode(src, width=img.shape[1], height=img.shape[0], bits_per_sample=bits or hdr.bits_per_sample,
                         component_count=3 if img.ndim == 3 else 1, near_lossless=near, interleave_mode=ilv,
                         preset=preset)
        assert out == jls


@pytest.mark.parametrize("name,ilv,near", [("t8c0e0", 0, 0), ("t8c1e0", 1, 0), ("t8c2e0", 2, 0),
                                           ("t8c0e3", 0, 3), ("t8c1e3", 1, 3), ("t8c2e3", 2, 3)])
def test_iso_conformance_colour(lib, name, ilv, near):
    _fixture_roundtrip(lib, f"{name}.jls", "test8.ppm", ilv, near, True)


def test_iso_conformance_other(lib):
    _fixture_roundtrip(lib, "t8nde0.jls", "test8bs2.pgm", 0, 0, True, preset=(255, 9, 9, 9, 31))
    _fixture_roundtrip(lib, "t8nde3.jls", "test8bs2.pgm", 0, 3, True, preset=(255, 9, 9, 9, 31))
    _fixture_roundtrip(lib, "t16e0.jls", "test16.pgm", 0, 0, True, bits=12)
    _fixture_roundtrip(lib, "t16e3.jls", "test16.pgm", 0, 3, False, bits=12)
    _fixture_roundtrip(lib, "tulips-gray-8bit-512-512-hp-encoder.jls", "tulips-gray-8bit-512-512.pgm", 0, 0, True)


@pytest.mark.parametrize("pnm,bits,size", [("2bit_parrot_150x200.pgm", 2, 2866), ("4bit-monochrome.pgm", 4, 1596),
                                           ("16-bit-640-480-many-dots.pgm", 16, 4138)])
def test_reference_encode_sizes(lib, pnm, bits, size):
    img, _ = common.read_pnm(pnm)
    out = lib.encode(img, width=img.shape[1], height=img.shape[0], bits_per_sample=bits)
    assert len(out) == size
    assert lib.decode(out)[1].tobytes() == img.tobytes()


@pytest.mark.parametrize("name,pnm,ilv", [("test8_ilv_none_rm_7", "test8.ppm", 0), ("test8_ilv_line_rm_7", "test8.ppm", 1),
                                          ("test8_ilv_sample_rm_7", "test8.ppm", 2),
                                          ("test8_ilv_sample_rm_300", "test8.ppm", 2), ("test16_rm_5", "test16.pgm", 0)])
def test_restart_interval_streams(lib, name, pnm, ilv):
    _fixture_roundtrip(lib, f"{name}.jls", pnm, ilv, 0, False)


@pytest.mark.parametrize("name,errc", [("fuzzy-input-bad-run-mode-golomb-code.jls", 5),
                                       ("fuzzy-input-no-valid-bits-at-the-end.jls", 5),
                                       ("fuzzy_input_golomb_16.jls", 5), ("no_start_byte_after_encoded_scan.jls", 4)])
def test_corrupt_streams_report_reference_errc(lib, name, errc):
    with pytest.raises(JpegLSError) as e:
        lib.decode(common.refdata(name))
    assert e.value.errc == errc


def test_colour_transform_fixtures(lib):
    img, _ = common.read_pnm("banny.ppm")
    for ct in (1, 2, 3):
        assert lib.decode(common.refdata(f"banny-hp{ct}.jls"))[1].tobytes() == img.tobytes()


def test_destination_too_small_boundary(lib):
    """Same accept/reject decision as the reference for every destination size around the exact output size."""
    img = synth.frame_numpy(64, 64, seed=3, kind="mixed")
    full = ob.encode(img, width=64, height=64)
    for cap in range(len(full) - 8, len(full) + 8):
        try:
            a, ea = ob.encode(img, width=64, height=64, destination_size=cap), 0
        except ob.OracleError as e:
            a, ea = None, e.errc
        try:
            b, eb = lib.encode(img, width=64, height=64, destination_size=cap), 0
        except JpegLSError as e:
            b, eb = None, e.errc
        assert (ea, a) == (eb, b), cap


def test_stride_and_padding_are_respected(lib):
    img = synth.frame_numpy(50, 20, seed=4, kind="mixed")
    padded = np.full((20, 64), 0xAB, dtype=np.uint8)
    padded[:, :50] = img
    jls = lib.encode(padded, width=50, height=20, stride=64)
    assert jls == ob.encode(img, width=50, height=20)
    hdr, px = lib.decode(jls, stride=64, destination_size=64 * 20 - 14)
    got = np.concatenate([px, np.zeros(14, np.uint8)]).reshape(20, 64)
    assert np.array_equal(got[:, :50], img) and (got[:-1, 50:] == 0).all()  # padding bytes untouched


def test_random_parameters_against_oracle(lib):
    rng = np.random.default_rng(77)
    for it in range(60):
        bits = int(rng.integers(2, 17))
        comps = int(rng.choice([1, 1, 3, 4, 2]))
        ilv = 0 if comps == 1 else int(rng.integers(0, 3))
        w, h = int(rng.choice([1, 3, 17, 64, 129])), int(rng.choice([1, 2, 8, 33]))
        maxval = (1 << bits) - 1
        near = 0 if rng.random() < 0.5 else int(rng.integers(0, min(255, maxval // 2) + 1))
        ct = int(rng.integers(1, 4)) if (comps == 3 and bits in (8, 16) and near == 0 and ilv != 0 and rng.random() < 0.5) else 0
        kind = str(rng.choice(["mixed", "gradient", "hard", "noise", "zero"]))
        img = synth.frame_numpy(w, h, seed=it, bits=bits, components=comps, kind=kind, interleaved=(ilv != 0))
        kw = dict(width=w, height=h, bits_per_sample=bits, component_count=comps, near_lossless=near,
                  interleave_mode=ilv, color_transformation=ct)
        want = ob.encode(img, **kw)
        assert lib.encode(img, **kw) == want, kw
        assert lib.decode(want)[1].tobytes() == ob.decode(want)[1].tobytes(), kw


@pytest.mark.slow
@pytest.mark.parametrize("name", ["cfg2_full", "cfg2_full_mixed", "cfg3_full", "cfg3b_full_12bit", "cfg4_frame0", "full_tulips_tiled_0",
                                  "full_tulips_tiled_1", "full_noise"])
def test_baseline_configs_full_size_hash(lib, name):
    """BASELINE.json configs at full size: bytes identical to the reference (hash committed), decode restores the input.
    full_tulips_tiled_*: the natural image of bench.py's `tulips` rows (tests/golden/make_golden_full_frames.py);
    full_noise: every sample uniform in 0..255 -- the encoder's worst case, 1.06 bytes per sample."""
    c = next(c for c in CASES if c["name"] == name)
    img = common.case_input(c)
    kwargs = common.case_kwargs(c)
    if name == "full_noise":  # (more than charls_jpegls_encoder_get_estimated_destination_size reserves, for CharLS as for this engine)
        kwargs["destination_size"] = 2 * img.size + 1024
    jls = lib.encode(img, **kwargs)
    assert (len(jls), common.sha(jls)) == (c["jls_size"], c["jls_sha256"])
    assert lib.decode(jls)[1].tobytes() == img.tobytes()
