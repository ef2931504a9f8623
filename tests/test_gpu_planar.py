"""Planar (ILV_NONE) multi-component frames through the C ABI: the component scans are coded / decoded by ONE launch
(reference loops: src/charls_jpegls_encoder.cpp:209-224, src/charls_jpegls_decoder.cpp:177-201); bytes, pixels and error
codes must be those of the scan-by-scan reference.  GPU only."""
import time

import numpy as np
import pytest

import common
import oracle_bind as ob
from charls_amd import capi, synth
from charls_amd.capi import JpegLSError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    L = capi.load_product()
    assert L.lib.charls_amd_device_status() == 0
    return L


def _planes(comps, w, h, bits, kind, seed):
    return np.stack([synth.frame_numpy(w, h, seed=seed + 17 * c, bits=bits, kind=kind) for c in range(comps)], axis=0)


@pytest.mark.parametrize("comps,bits,near,w,h,kind", [(3, 8, 0, 200, 120, "mixed"), (4, 8, 0, 65, 33, "mixed"), (2, 12, 0, 129, 70, "hard"),
                                                       (3, 16, 0, 96, 64, "mixed"), (3, 8, 2, 160, 90, "mixed"), (3, 8, 0, 1, 1, "mixed"),
                                                       (3, 8, 0, 1024, 600, "gradient")])
def test_planar_frames_equal_the_oracle_both_ways(lib, comps, bits, near, w, h, kind):
    img = _planes(comps, w, h, bits, kind, seed=comps * 100 + bits)
    kw = dict(width=w, height=h, bits_per_sample=bits, component_count=comps, near_lossless=near, interleave_mode=0)
    want = ob.encode(img, **kw)
    got = lib.encode(img, **kw)
    assert got == want
    hdr, px = lib.decode(want)
    assert px.tobytes() == ob.decode(want)[1].tobytes()
    if near == 0:
        assert px.tobytes() == img.tobytes()


def test_planar_frame_with_row_padding_and_restart_intervals(lib):
    comps, w, h = 3, 150, 40
    img = _planes(comps, w, h, 8, "mixed", seed=5)
    want = ob.encode(img, width=w, height=h, component_count=comps, interleave_mode=0)
    stride = w + 13
    padded = np.zeros((comps, h, stride), dtype=np.uint8)
    padded[:, :, :w] = img
    assert lib.encode(padded, width=w, height=h, component_count=comps, interleave_mode=0, stride=stride) == want
    hdr, px = lib.decode(want, stride=stride)
    back = np.zeros(comps * h * stride, dtype=np.uint8)
    back[:px.size] = px
    assert np.array_equal(back.reshape(comps, h, stride)[:, :, :w], img)
    dri = lib.encode(img, width=w, height=h, component_count=comps, interleave_mode=0, restart_interval=8)
    assert lib.decode(dri)[1].tobytes() == img.tobytes() == ob.decode(dri)[1].tobytes()


def test_planar_destination_too_small_is_decided_scan_by_scan(lib):
    """The destination the reference hands to scan c is what is left behind scans 0..c-1 and c's own header: every size
    from far too small up to enough must give the reference's verdict (and its bytes once it fits)."""
    comps, w, h = 3, 90, 50
    img = _planes(comps, w, h, 8, "mixed", seed=77)
    kw = dict(width=w, height=h, component_count=comps, interleave_mode=0)
    want = ob.encode(img, **kw)
    n = len(want)
    for size in list(range(n - 12, n + 6)) + [n // 2, n // 3, 200]:
        try:
            ref = ob.encode(img, destination_size=size, **kw)
            ref_errc = 0
        except ob.OracleError as e:
            ref, ref_errc = None, e.errc
        try:
            got = lib.encode(img, destination_size=size, **kw)
            got_errc = 0
        except JpegLSError as e:
            got, got_errc = None, e.errc
        assert got_errc == ref_errc, size
        assert got == ref, size


def test_planar_stream_with_a_comment_between_scans(lib):
    """Segments between the scans (here COM) are parsed on the way from scan to scan; decoding still gives the oracle's
    pixels, and a stream damaged inside its SECOND scan gives the oracle's error."""
    comps, w, h = 3, 64, 40
    img = _planes(comps, w, h, 8, "mixed", seed=3)
    jls = bytearray(ob.encode(img, width=w, height=h, component_count=comps, interleave_mode=0))
    second_sos = jls.find(b"\xFF\xDA", jls.find(b"\xFF\xDA") + 2)
    with_comment = bytes(jls[:second_sos]) + b"\xFF\xFE\x00\x07hello" + bytes(jls[second_sos:])
    assert lib.decode(with_comment)[1].tobytes() == img.tobytes()
    damaged = bytearray(with_comment)
    at = with_comment.find(b"\xFF\xDA", second_sos + 9) + 40
    damaged[at:at + 6] = b"\xFF\xFF\xFF\xFF\xFF\xFF"
    try:
        want_px, want_errc = ob.decode(bytes(damaged))[1].tobytes(), 0
    except ob.OracleError as e:
        want_px, want_errc = None, e.errc
    try:
        got_px, got_errc = lib.decode(bytes(damaged))[1].tobytes(), 0
    except JpegLSError as e:
        got_px, got_errc = None, e.errc
    assert got_errc == want_errc
    if want_errc == 0:
        assert got_px == want_px


def test_planar_rgb_decodes_in_about_the_time_of_one_plane(lib):
    """2048 x 2048 planar RGB through decode_to_buffer next to one plane of it: the three scans run concurrently."""
    w = h = 2048
    img = _planes(3, w, h, 8, "gradient", seed=11)
    rgb = lib.encode(img, width=w, height=h, component_count=3, interleave_mode=0)
    one = lib.encode(img[0], width=w, height=h)
    assert common.sha(rgb) == common.sha(ob.encode(img, width=w, height=h, component_count=3, interleave_mode=0))
    lib.decode(one)
    t0 = time.perf_counter()
    assert lib.decode(one)[1].tobytes() == img[0].tobytes()
    t1 = time.perf_counter()
    assert lib.decode(rgb)[1].tobytes() == img.tobytes()
    t2 = time.perf_counter()
    print(f"\n[planar] 2048x2048: one plane {t1 - t0:.3f}s, three planes {t2 - t1:.3f}s")
    assert (t2 - t1) < 2.0 * (t1 - t0)
