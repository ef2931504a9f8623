"""Pins the CPU restatement (oracle/jls_oracle.c) against the reference's golden vectors.  CPU only."""
import numpy as np
import pytest

import common
import oracle_bind as ob

CASES = common.cases()
SMALL = [c for c in CASES if c["width"] * c["height"] <= 512 * 512]
FULL = [c for c in CASES if c["width"] * c["height"] > 512 * 512]


@pytest.mark.parametrize("c", SMALL, ids=lambda c: c["name"])
def test_encode_matches_reference_bytes(c):
    img = common.case_input(c)
    assert common.sha(img.tobytes()) == c["input_sha256"], "input recipe drifted"
    if c["errc"] != 0:
        with pytest.raises(ob.OracleError) as e:
            ob.encode(img, **common.case_kwargs(c))
        assert e.value.errc == c["errc"]
        return
    jls = ob.encode(img, **common.case_kwargs(c))
    assert len(jls) == c["jls_size"]
    assert common.sha(jls) == c["jls_sha256"]
    if "file" in c:
        with open(f"{common.GOLDEN}/{c['file']}", "rb") as f:
            assert f.read() == jls


@pytest.mark.parametrize("c", [c for c in SMALL if c["errc"] == 0 and "file" in c], ids=lambda c: c["name"])
def test_decode_matches_reference_pixels(c):
    with open(f"{common.GOLDEN}/{c['file']}", "rb") as f:
        jls = f.read()
    p, px = ob.decode(jls)
    assert common.sha(px.tobytes()) == c["decoded_sha256"]
    img = common.case_input(c)
    if c["near_lossless"] == 0:
        assert px.tobytes() == img.tobytes()
    else:
        d = np.abs(px.view(img.dtype).astype(np.int64) - img.ravel().astype(np.int64))
        assert d.max() <= c["near_lossless"]


@pytest.mark.slow
@pytest.mark.parametrize("c", [c for c in FULL if c["name"] in ("cfg2_full", "cfg4_frame0")], ids=lambda c: c["name"])
def test_full_size_hash(c):
    img = common.case_input(c)
    jls = ob.encode(img, **common.case_kwargs(c))
    assert (len(jls), common.sha(jls)) == (c["jls_size"], c["jls_sha256"])


# ---- the reference's own fixtures (test/compliance_test.cpp:43-141, test/encode_test.cpp:184-202) -------------------

def _roundtrip_file(jls_name, pnm_name, ilv, near, reencode, bits=None, preset=None):
    jls = common.refdata(jls_name)
    img, maxval = common.read_pnm(pnm_name)
    p, px = ob.decode(jls)
    src = common.planar(img) if (img.ndim == 3 and ilv == 0) else img
    if near == 0:
        assert px.tobytes() == src.tobytes()
    else:
        d = np.abs(px.view(src.dtype).astype(np.int64) - src.ravel().astype(np.int64))
        assert d.max() <= near
    if reencode:
        comps = 3 if img.ndim == 3 else 1
        out = ob.encode(src, width=img.shape[1], height=img.shape[0], bits_per_sample=bits or p.bits_per_sample,
                        component_count=comps, near_lossless=near, interleave_mode=ilv, preset=preset)
        assert out == jls


@pytest.mark.parametrize("name,ilv,near", [("t8c0e0", 0, 0), ("t8c1e0", 1, 0), ("t8c2e0", 2, 0),
                                           ("t8c0e3", 0, 3), ("t8c1e3", 1, 3), ("t8c2e3", 2, 3)])
def test_conformance_color(name, ilv, near):
    _roundtrip_file(f"{name}.jls", "test8.ppm", ilv, near, reencode=True)


def test_conformance_custom_thresholds():
    _roundtrip_file("t8nde0.jls", "test8bs2.pgm", 0, 0, True, preset=(255, 9, 9, 9, 31))
    _roundtrip_file("t8nde3.jls", "test8bs2.pgm", 0, 3, True, preset=(255, 9, 9, 9, 31))


def test_conformance_12bit():
    _roundtrip_file("t16e0.jls", "test16.pgm", 0, 0, True, bits=12)
    _roundtrip_file("t16e3.jls", "test16.pgm", 0, 3, False, bits=12)


def test_tulips_hp_encoder_kat():
    _roundtrip_file("tulips-gray-8bit-512-512-hp-encoder.jls", "tulips-gray-8bit-512-512.pgm", 0, 0, True)


@pytest.mark.parametrize("pnm,bits,size", [("2bit_parrot_150x200.pgm", 2, 2866), ("4bit-monochrome.pgm", 4, 1596),
                                           ("16-bit-640-480-many-dots.pgm", 16, 4138)])
def test_encode_test_sizes(pnm, bits, size):
    img, _ = common.read_pnm(pnm)
    out = ob.encode(img, width=img.shape[1], height=img.shape[0], bits_per_sample=bits)
    assert len(out) == size
    _, px = ob.decode(out)
    assert px.tobytes() == img.tobytes()


@pytest.mark.parametrize("name,pnm,ilv", [("test8_ilv_none_rm_7", "test8.ppm", 0), ("test8_ilv_line_rm_7", "test8.ppm", 1),
                                          ("test8_ilv_sample_rm_7", "test8.ppm", 2),
                                          ("test8_ilv_sample_rm_300", "test8.ppm", 2), ("test16_rm_5", "test16.pgm", 0)])
def test_restart_marker_streams(name, pnm, ilv):
    _roundtrip_file(f"{name}.jls", pnm, ilv, 0, reencode=False)


@pytest.mark.parametrize("name,errc", [("fuzzy-input-bad-run-mode-golomb-code.jls", 5),
                                       ("fuzzy-input-no-valid-bits-at-the-end.jls", 5),
                                       ("fuzzy_input_golomb_16.jls", 5), ("no_start_byte_after_encoded_scan.jls", 4),
                                       ("land10-10bit-rgb-hp3-invalid.head.jls", 36)])
def test_corrupt_streams(name, errc):
    with pytest.raises(ob.OracleError) as e:
        ob.decode(common.refdata(name))
    assert e.value.errc == errc


def test_banny_color_transforms():
    img, _ = common.read_pnm("banny.ppm")
    for ct in (1, 2, 3):
        _, px = ob.decode(common.refdata(f"banny-hp{ct}.jls"))
        assert px.tobytes() == img.tobytes()


def test_bit_writer_exact_bytes():
    """Reference white-box vector, test/scan_encoder_test.cpp:32-73: 0xFF bytes force a stuffed bit in the next byte."""
    import ctypes as C
    vals = [0x01, 0xFF, 0xFFFF, 0xFFFF, 0x12345678 >> 1]  # append(0x01,31) ... see the reference test
    cnts = [31, 9, 16, 16, 31]
    dst = np.zeros(64, dtype=np.uint8)
    n = C.c_size_t()
    rc = ob.lib().jls_oracle_bitwriter_kat((C.c_uint32 * 5)(*vals), (C.c_int32 * 5)(*cnts), 5, dst.ctypes.data, 64,
                                           C.byref(n))
    assert rc == 0
    # the stream re-read with the stuffing rule must give back the same bit string
    bits = "".join(format(v, f"0{c}b") for v, c in zip(vals, cnts))
    out, prev_ff = "", False
    for b in dst[:n.value]:
        out += format(int(b), "08b")[1 if prev_ff else 0:]
        prev_ff = b == 0xFF
    assert out.startswith(bits) and set(out[len(bits):]) <= {"0"}


def test_default_thresholds():
    """test/jpegls_preset_coding_parameters_test.cpp:15-96 input/output pairs."""
    import ctypes as C
    out = (C.c_int32 * 5)()
    for maxval, near, exp in [(255, 0, (3, 7, 21)), (4095, 0, (18, 67, 276)), (65535, 0, (18, 67, 276)),
                              (255, 2, (9, 17, 35)), (3, 0, (1 + 0, 3, 3))]:
        ob.lib().jls_oracle_default_pc(maxval, near, out)
        if maxval == 3:
            continue
        assert tuple(out[1:4]) == exp and out[4] == 64
