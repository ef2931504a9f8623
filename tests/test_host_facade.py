"""The C-ABI facade of libcharls_amd.so on the CPU: exports, argument checks, state machines, marker writer/reader.
No scan is coded here (that needs the GPU: tests/test_gpu_*.py); without a GPU the coding calls must fail loudly."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import abi_scripts as S
import common
from charls_amd import capi, synth
from charls_amd.capi import JpegLSError


@pytest.fixture(scope="module")
def lib():
    from charls_amd import build
    build.build()
    return capi.load_product()


@pytest.fixture(scope="module")
def observed():
    with open(os.path.join(common.GOLDEN, "abi_observations.json")) as f:
        return json.load(f)


def test_exports_every_declared_symbol(lib):
    import re
    header = open(os.path.join(common.ROOT, "include", "charls_amd.h")).read()
    declared = set(re.findall(r"\b(charls_[a-z0-9_]+)\s*\(", header))
    declared -= {"charls_at_comment_handler", "charls_at_application_data_handler"}
    assert set(capi.all_abi_symbols()) <= declared and len(capi.all_abi_symbols()) == 48
    for name in sorted(declared):
        assert hasattr(lib.lib, name), name


def test_struct_sizes():
    # include/charls/public_types.h:1075-1078
    assert (C.sizeof(capi.SpiffHeader), C.sizeof(capi.FrameInfo), C.sizeof(capi.PcParameters),
            C.sizeof(capi.MappingTableInfo)) == (40, 16, 20, 12)


def _normalise(x):
    return json.loads(json.dumps(x))


@pytest.mark.parametrize("script", [f.__name__ for f in S.SCRIPTS if f.__name__ != "encoder_encode_argument_checks"])
def test_abi_script_matches_reference(lib, observed, script):
    got = _normalise(getattr(S, script)(lib))
    want = observed[script]
    if isinstance(want, dict):
        for k in want:
            assert got[k] == want[k], k
    else:
        assert got == want


def test_encode_argument_checks_match_reference(lib, observed):
    """Identical up to the point where a scan would be coded; there the CPU-only product must answer 200, never encode."""
    got = _normalise(S.encoder_encode_argument_checks(lib))
    want = observed["encoder_encode_argument_checks"]
    has_gpu = lib.lib.charls_amd_device_status() == 0
    for g, w in zip(got, want):
        if w[-1] == 0 and not has_gpu:
            assert g[:-1] == w[:-1] and g[-1] == 200
        else:
            assert g == w


def test_spiff_and_callbacks_match_reference(lib, observed):
    streams = {}
    for name in ("spiff_com_app", "com_only", "spiff_entries"):
        with open(os.path.join(common.GOLDEN, "container", name + ".jls"), "rb") as f:
            streams[name] = f.read()
    got = _normalise(S.decoder_spiff_and_callbacks(lib, streams))
    for name, want in observed["decoder_spiff_and_callbacks"].items():
        g = got[name]
        # entries that are the rc of decode_to_buffer with a too-small destination are identical (110); nothing decodes
        assert g == want, name


def test_mapping_tables_match_reference(lib, observed):
    with open(os.path.join(common.GOLDEN, "container", "abbreviated_tables.jls"), "rb") as f:
        data = f.read()
    assert _normalise(S.decoder_mapping_tables(lib, data)) == observed["decoder_mapping_tables"]


@pytest.mark.parametrize("c", [c for c in common.cases() if "file" in c], ids=lambda c: c["name"])
def test_read_header_of_golden_files(lib, c):
    with open(f"{common.GOLDEN}/{c['file']}", "rb") as f:
        h = lib.read_header(f.read())
    assert (h.width, h.height, h.bits_per_sample, h.component_count) == (c["width"], c["height"], c["bits_per_sample"],
                                                                         c["component_count"])
    assert h.near_lossless == c["near_lossless"] and h.color_transformation == c["color_transformation"]
    assert h.interleave_mode == c["interleave_mode"]


def test_reference_fixture_headers(lib):
    assert lib.read_header(common.refdata("t8nde0.jls")).preset == (255, 9, 9, 9, 31)
    with pytest.raises(JpegLSError) as e:
        lib.read_header(common.refdata("land10-10bit-rgb-hp3-invalid.head.jls"))
    assert e.value.errc == 36  # test/jpegls_decoder_test.cpp:1549-1556
    with pytest.raises(JpegLSError) as e:
        lib.read_header(common.refdata("t8c0e0.jls")[:20])
    assert e.value.errc in (4, 15)


def test_no_cpu_fallback(lib):
    """Without a GPU every coding entry point reports CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE (200)."""
    if lib.lib.charls_amd_device_status() == 0:
        pytest.skip("a GPU is present")
    img = synth.frame_numpy(32, 32)
    with pytest.raises(JpegLSError) as e:
        lib.encode(img, width=32, height=32)
    assert e.value.errc == 200
    with open(f"{common.GOLDEN}/small/gray8_64x48.jls", "rb") as f:
        with pytest.raises(JpegLSError) as e:
            lib.decode(f.read())
    assert e.value.errc == 200


def test_restart_interval_setter_is_additive_and_checked(lib):
    """charls_amd_jpegls_encoder_set_restart_interval (include/charls_amd.h part 2): argument / state checks and its only
    CPU-visible effect, the size estimate; with the default (0) nothing differs from the reference's ABI behaviour."""
    L = lib.lib
    fn = L.charls_amd_jpegls_encoder_set_restart_interval
    fn.argtypes = [C.c_void_p, C.c_uint32]
    fn.restype = C.c_int32
    assert fn(None, 8) == 101  # invalid_argument (charls_jpegls_errc 101), as the reference reports a null handle
    L.charls_jpegls_encoder_create.restype = C.c_void_p
    enc = L.charls_jpegls_encoder_create()
    try:
        fi = capi.FrameInfo(640, 480, 8, 3)
        assert L.charls_jpegls_encoder_set_frame_info(C.c_void_p(enc), C.byref(fi)) == 0
        n0, n1, n2 = C.c_size_t(), C.c_size_t(), C.c_size_t()
        L.charls_jpegls_encoder_get_estimated_destination_size.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        assert L.charls_jpegls_encoder_get_estimated_destination_size(enc, C.byref(n0)) == 0
        assert fn(enc, 16) == 0
        assert L.charls_jpegls_encoder_get_estimated_destination_size(enc, C.byref(n1)) == 0
        assert n1.value == n0.value + 6 + 2 * 30 * 3  # DRI segment + one marker per interval and component
        assert fn(enc, 0) == 0
        assert L.charls_jpegls_encoder_get_estimated_destination_size(enc, C.byref(n2)) == 0
        assert n2.value == n0.value
    finally:
        L.charls_jpegls_encoder_destroy.argtypes = [C.c_void_p]
        L.charls_jpegls_encoder_destroy(enc)


def test_every_symbol_the_header_declares_is_exported():
    """include/charls_amd.h is the boundary: every function it declares (the 48 of the reference and the additive
    charls_amd_* entry points) must be an exported symbol of the library -- loads and look-ups only, no compute."""
    import ctypes
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "include", "charls_amd.h")) as f:
        text = f.read()
    declared = set(re.findall(r"CHARLS_AMD_API\s+[\w\s\*]+?\b(charls_\w+)\s*\(", text))
    assert len(declared) >= 48 + 11, sorted(declared)
    assert {"charls_amd_encode_batch_device", "charls_amd_decode_batch_device", "charls_amd_encode_batch_devices",
            "charls_amd_decode_batch_devices"} <= declared
    lib = ctypes.CDLL(os.path.join(root, "charls_amd", "lib", "libcharls_amd.so"))
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing


def test_knobs_and_engine_counters_are_plain_host_calls(lib):
    """charls_amd_debug_set_knob / charls_amd_engine_counters (include/charls_amd.h part 2): a table and a few counters, usable
    without a GPU.  Unknown names are refused; a cleared knob can be set again; the counters come back as ten values."""
    from charls_amd import capi
    L = lib.lib
    L.charls_amd_debug_set_knob.argtypes = [C.c_char_p, C.c_int64]
    L.charls_amd_debug_set_knob.restype = C.c_int32
    assert L.charls_amd_debug_set_knob(b"TILE_SAMPLES", 256) == 0
    assert L.charls_amd_debug_set_knob(b"CHARLS_AMD_TILE_SAMPLES", capi.KNOB_UNSET) == 0  # (with the prefix too; cleared)
    assert L.charls_amd_debug_set_knob(b"NO_SUCH_KNOB", 1) == 101
    assert L.charls_amd_debug_set_knob(None, 1) == 101
    counters = capi.engine_counters(lib)
    assert set(counters) == {"calls", "launches", "merged_calls", "largest_launch", "pipeline_fallback_scans", "split_launches",
                             "idle_pool_bytes", "deferred_free_bytes", "idle_releases", "exact_retry_scans"}
    assert L.charls_amd_debug_set_knob(b"IDLE_RELEASE_MS", 500) == 0 and L.charls_amd_debug_set_knob(b"IDLE_RELEASE_MS", capi.KNOB_UNSET) == 0
    assert all(v >= 0 for v in counters.values())
