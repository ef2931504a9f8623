"""ctypes binding of the CharLS C ABI (``charls_jpegls_encoder_*`` / ``charls_jpegls_decoder_*``).

The binding is library-agnostic on purpose: the same class drives

* ``charls_amd/lib/libcharls_amd.so`` -- this repository's MI355X engine (the product), and
* ``oracle/_ref/libcharls_ref.so``    -- the reference compiled from /root/reference (tests / cpu_baseline only),

because both export the interface declared in the reference's ``include/charls/charls_jpegls_encoder.h:24-316`` and
``include/charls/charls_jpegls_decoder.h:24-293``.  The Python classes mirror the reference's header-only C++ wrappers
(``include/charls/jpegls_encoder.hpp:58-449``, ``jpegls_decoder.hpp:108-570``): same method names, same argument
meaning, errors surface as :class:`JpegLSError` carrying the ``charls_jpegls_errc`` value.

There is no CPU fallback here: if the product library is missing, :func:`load_product` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "lib", "libcharls_amd.so")


class JpegLSError(RuntimeError):
    def __init__(self, errc: int, where: str = ""):
        self.errc = int(errc)
        super().__init__(f"charls_jpegls_errc={self.errc} in {where}")


class FrameInfo(C.Structure):  # include/charls/public_types.h:983-1000 (16 bytes)
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bits_per_sample", C.c_int32),
                ("component_count", C.c_int32)]


class PcParameters(C.Structure):  # include/charls/public_types.h:1003-1021 (20 bytes)
    _fields_ = [("maximum_sample_value", C.c_int32), ("threshold1", C.c_int32), ("threshold2", C.c_int32),
                ("threshold3", C.c_int32), ("reset_value", C.c_int32)]


class SpiffHeader(C.Structure):  # include/charls/public_types.h:934-980 (40 bytes)
    _fields_ = [("profile_id", C.c_int32), ("component_count", C.c_int32), ("height", C.c_uint32),
                ("width", C.c_uint32), ("color_space", C.c_int32), ("bits_per_sample", C.c_int32),
                ("compression_type", C.c_int32), ("resolution_units", C.c_int32),
                ("vertical_resolution", C.c_uint32), ("horizontal_resolution", C.c_uint32)]


class MappingTableInfo(C.Structure):  # include/charls/public_types.h:1024-1034 (12 bytes)
    _fields_ = [("table_id", C.c_int32), ("entry_size", C.c_int32), ("data_size", C.c_uint32)]


ENCODER_SYMBOLS = [
    "create", "destroy", "set_frame_info", "set_near_lossless", "set_encoding_options", "set_interleave_mode",
    "set_preset_coding_parameters", "set_color_transformation", "set_mapping_table_id",
    "get_estimated_destination_size", "set_destination_buffer", "write_standard_spiff_header", "write_spiff_header",
    "write_spiff_entry", "write_spiff_end_of_directory_entry", "write_comment", "write_application_data",
    "write_mapping_table", "encode_from_buffer", "encode_components_from_buffer", "create_abbreviated_format",
    "get_bytes_written", "rewind"]
DECODER_SYMBOLS = [
    "create", "destroy", "set_source_buffer", "read_spiff_header", "read_header", "get_frame_info",
    "get_near_lossless", "get_interleave_mode", "get_preset_coding_parameters", "get_color_transformation",
    "get_destination_size", "decode_to_buffer", "at_comment", "at_application_data"]
DECODER_SYMBOLS2 = [
    "get_compressed_data_format", "get_mapping_table_id", "find_mapping_table_index", "get_mapping_table_count",
    "get_mapping_table_info", "get_mapping_table_data"]
MISC_SYMBOLS = ["charls_get_error_message", "charls_get_jpegls_category", "charls_get_version_string",
                "charls_get_version_number", "charls_validate_spiff_header"]


def all_abi_symbols() -> list[str]:
    """The 48 exported names of the reference (src/charls.version:1-21)."""
    return ([f"charls_jpegls_encoder_{s}" for s in ENCODER_SYMBOLS] +
            [f"charls_jpegls_decoder_{s}" for s in DECODER_SYMBOLS] +
            [f"charls_decoder_{s}" for s in DECODER_SYMBOLS2] + MISC_SYMBOLS)


@dataclass
class Header:
    width: int
    height: int
    bits_per_sample: int
    component_count: int
    near_lossless: int
    interleave_mode: int
    color_transformation: int
    preset: tuple


class CharLSLibrary:
    """One loaded implementation of the CharLS C ABI."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise FileNotFoundError(f"CharLS-ABI library not found: {path}")
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        vp, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int32
        L.charls_jpegls_encoder_create.restype = vp
        L.charls_jpegls_encoder_destroy.argtypes = [vp]
        L.charls_jpegls_encoder_destroy.restype = None
        L.charls_jpegls_decoder_create.restype = vp
        L.charls_jpegls_decoder_destroy.argtypes = [vp]
        L.charls_jpegls_decoder_destroy.restype = None
        sig = {
            "charls_jpegls_encoder_set_frame_info": [vp, C.POINTER(FrameInfo)],
            "charls_jpegls_encoder_set_near_lossless": [vp, i32],
            "charls_jpegls_encoder_set_encoding_options": [vp, u32],
            "charls_jpegls_encoder_set_interleave_mode": [vp, i32],
            "charls_jpegls_encoder_set_preset_coding_parameters": [vp, C.POINTER(PcParameters)],
            "charls_jpegls_encoder_set_color_transformation": [vp, i32],
            "charls_jpegls_encoder_set_mapping_table_id": [vp, i32, i32],
            "charls_jpegls_encoder_get_estimated_destination_size": [vp, C.POINTER(sz)],
            "charls_jpegls_encoder_set_destination_buffer": [vp, vp, sz],
            "charls_jpegls_encoder_write_standard_spiff_header": [vp, i32, i32, u32, u32],
            "charls_jpegls_encoder_write_spiff_header": [vp, C.POINTER(SpiffHeader)],
            "charls_jpegls_encoder_write_spiff_entry": [vp, u32, vp, sz],
            "charls_jpegls_encoder_write_spiff_end_of_directory_entry": [vp],
            "charls_jpegls_encoder_write_comment": [vp, vp, sz],
            "charls_jpegls_encoder_write_application_data": [vp, i32, vp, sz],
            "charls_jpegls_encoder_write_mapping_table": [vp, i32, i32, vp, sz],
            "charls_jpegls_encoder_encode_from_buffer": [vp, vp, sz, u32],
            "charls_jpegls_encoder_encode_components_from_buffer": [vp, vp, sz, i32, u32],
            "charls_jpegls_encoder_create_abbreviated_format": [vp],
            "charls_jpegls_encoder_get_bytes_written": [vp, C.POINTER(sz)],
            "charls_jpegls_encoder_rewind": [vp],
            "charls_jpegls_decoder_set_source_buffer": [vp, vp, sz],
            "charls_jpegls_decoder_read_spiff_header": [vp, C.POINTER(SpiffHeader), C.POINTER(i32)],
            "charls_jpegls_decoder_read_header": [vp],
            "charls_jpegls_decoder_get_frame_info": [vp, C.POINTER(FrameInfo)],
            "charls_jpegls_decoder_get_near_lossless": [vp, i32, C.POINTER(i32)],
            "charls_jpegls_decoder_get_interleave_mode": [vp, i32, C.POINTER(i32)],
            "charls_jpegls_decoder_get_preset_coding_parameters": [vp, i32, C.POINTER(PcParameters)],
            "charls_jpegls_decoder_get_color_transformation": [vp, C.POINTER(i32)],
            "charls_jpegls_decoder_get_destination_size": [vp, u32, C.POINTER(sz)],
            "charls_jpegls_decoder_decode_to_buffer": [vp, vp, sz, u32],
            "charls_jpegls_decoder_at_comment": [vp, vp, vp],
            "charls_jpegls_decoder_at_application_data": [vp, vp, vp],
            "charls_decoder_get_compressed_data_format": [vp, C.POINTER(i32)],
            "charls_decoder_get_mapping_table_id": [vp, i32, C.POINTER(i32)],
            "charls_decoder_find_mapping_table_index": [vp, i32, C.POINTER(i32)],
            "charls_decoder_get_mapping_table_count": [vp, C.POINTER(i32)],
            "charls_decoder_get_mapping_table_info": [vp, i32, C.POINTER(MappingTableInfo)],
            "charls_decoder_get_mapping_table_data": [vp, i32, vp, sz],
            "charls_validate_spiff_header": [C.POINTER(SpiffHeader), C.POINTER(FrameInfo)],
        }
        for name, argtypes in sig.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = i32
        L.charls_get_error_message.argtypes = [i32]
        L.charls_get_error_message.restype = C.c_char_p
        L.charls_get_version_string.restype = C.c_char_p
        L.charls_get_version_number.argtypes = [C.POINTER(i32)] * 3
        L.charls_get_version_number.restype = None

    # -- helpers -----------------------------------------------------------------------------------------------
    def _check(self, rc: int, where: str):
        if rc != 0:
            raise JpegLSError(rc, where)

    @staticmethod
    def _buf(a):
        """(pointer, nbytes, keepalive) of a bytes-like / ndarray."""
        if isinstance(a, np.ndarray):
            a = np.ascontiguousarray(a)
            return a.ctypes.data, a.nbytes, a
        b = (C.c_ubyte * len(a)).from_buffer_copy(bytes(a))
        return C.addressof(b), len(a), b

    def error_message(self, errc: int) -> str:
        return self.lib.charls_get_error_message(errc).decode()

    # -- jpegls_encoder::encode convenience (include/charls/jpegls_encoder.hpp:58-110) --------------------------
    def encode(self, image, *, width=None, height=None, bits_per_sample=8, component_count=1, near_lossless=0,
               interleave_mode=0, color_transformation=0, preset=None, encoding_options=0, stride=0,
               destination_size=None, restart_interval=0, destination=None):
        """Encode `image` (ndarray or bytes, user layout of SURVEY 8a row a20) to a .jls byte string.
        restart_interval != 0 uses charls_amd_jpegls_encoder_set_restart_interval (product library only).
        destination: a preallocated uint8 ndarray to encode into -- the methodology of the reference's cli/benchmark.cpp:60-90
        (destination allocated outside the measurement loop, handle inside); the result is then a VIEW of it, not a copy."""
        L = self.lib
        if isinstance(image, np.ndarray) and (width is None or height is None):
            if interleave_mode == 0 and component_count > 1:
                height, width = image.shape[1], image.shape[2]
            else:
                height, width = image.shape[0], image.shape[1]
        enc = L.charls_jpegls_encoder_create()
        if not enc:
            raise MemoryError
        try:
            fi = FrameInfo(width, height, bits_per_sample, component_count)
            self._check(L.charls_jpegls_encoder_set_frame_info(enc, C.byref(fi)), "set_frame_info")
            self._check(L.charls_jpegls_encoder_set_near_lossless(enc, near_lossless), "set_near_lossless")
            self._check(L.charls_jpegls_encoder_set_interleave_mode(enc, interleave_mode), "set_interleave_mode")
            if color_transformation:
                self._check(L.charls_jpegls_encoder_set_color_transformation(enc, color_transformation),
                            "set_color_transformation")
            if encoding_options:
                self._check(L.charls_jpegls_encoder_set_encoding_options(enc, encoding_options),
                            "set_encoding_options")
            if preset is not None:
                pc = PcParameters(*preset)
                self._check(L.charls_jpegls_encoder_set_preset_coding_parameters(enc, C.byref(pc)), "set_pc")
            if restart_interval:
                fn = L.charls_amd_jpegls_encoder_set_restart_interval
                fn.argtypes = [C.c_void_p, C.c_uint32]
                fn.restype = C.c_int32
                self._check(fn(enc, restart_interval), "set_restart_interval")
            if destination_size is None:
                n = C.c_size_t()
                self._check(L.charls_jpegls_encoder_get_estimated_destination_size(enc, C.byref(n)), "estimate")
                destination_size = n.value
            dst = destination if destination is not None else np.empty(destination_size, dtype=np.uint8)
            self._check(L.charls_jpegls_encoder_set_destination_buffer(enc, dst.ctypes.data, dst.nbytes), "set_dest")
            ptr, nbytes, keep = self._buf(image)
            self._check(L.charls_jpegls_encoder_encode_from_buffer(enc, ptr, nbytes, stride), "encode_from_buffer")
            n = C.c_size_t()
            self._check(L.charls_jpegls_encoder_get_bytes_written(enc, C.byref(n)), "get_bytes_written")
            del keep
            return dst[:n.value] if destination is not None else dst[:n.value].tobytes()
        finally:
            L.charls_jpegls_encoder_destroy(enc)

    # -- jpegls_decoder::decode convenience (include/charls/jpegls_decoder.hpp:108-180) --------------------------
    def read_header(self, data) -> Header:
        L = self.lib
        dec = L.charls_jpegls_decoder_create()
        try:
            ptr, n, keep = self._buf(data)
            self._check(L.charls_jpegls_decoder_set_source_buffer(dec, ptr, n), "set_source_buffer")
            self._check(L.charls_jpegls_decoder_read_header(dec), "read_header")
            return self._header(dec)
        finally:
            L.charls_jpegls_decoder_destroy(dec)

    def _header(self, dec) -> Header:
        L = self.lib
        fi = FrameInfo()
        self._check(L.charls_jpegls_decoder_get_frame_info(dec, C.byref(fi)), "get_frame_info")
        near, ilv, ct = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(L.charls_jpegls_decoder_get_near_lossless(dec, 0, C.byref(near)), "get_near")
        self._check(L.charls_jpegls_decoder_get_interleave_mode(dec, 0, C.byref(ilv)), "get_ilv")
        self._check(L.charls_jpegls_decoder_get_color_transformation(dec, C.byref(ct)), "get_ct")
        pc = PcParameters()
        self._check(L.charls_jpegls_decoder_get_preset_coding_parameters(dec, 0, C.byref(pc)), "get_pc")
        return Header(fi.width, fi.height, fi.bits_per_sample, fi.component_count, near.value, ilv.value, ct.value,
                      (pc.maximum_sample_value, pc.threshold1, pc.threshold2, pc.threshold3, pc.reset_value))

    def decode(self, data, stride=0, destination_size=None, out=None):
        """Decode a .jls byte string. Returns (Header, uint8 ndarray of the raw destination bytes).
        out: a preallocated uint8 ndarray to decode into (cli/benchmark.cpp:40-55: allocated outside the loop)."""
        L = self.lib
        dec = L.charls_jpegls_decoder_create()
        if not dec:
            raise MemoryError
        try:
            ptr, n, keep = self._buf(data)
            self._check(L.charls_jpegls_decoder_set_source_buffer(dec, ptr, n), "set_source_buffer")
            self._check(L.charls_jpegls_decoder_read_header(dec), "read_header")
            hdr = self._header(dec)
            if destination_size is None:
                sz = C.c_size_t()
                self._check(L.charls_jpegls_decoder_get_destination_size(dec, stride, C.byref(sz)), "get_dest_size")
                destination_size = sz.value
            if out is None:
                out = np.zeros(destination_size, dtype=np.uint8)
            self._check(L.charls_jpegls_decoder_decode_to_buffer(dec, out.ctypes.data, out.nbytes, stride),
                        "decode_to_buffer")
            del keep
            return hdr, out
        finally:
            L.charls_jpegls_decoder_destroy(dec)


_product = None


def load_product() -> CharLSLibrary:
    """Load this repository's engine. Raises if it has not been built -- there is no fallback."""
    global _product
    if _product is None:
        if not os.path.exists(PRODUCT_LIB):
            raise RuntimeError(
                f"{PRODUCT_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). charls_amd has no CPU fallback.")
        if "torch" not in sys.modules:
            # torch brings a HIP runtime of its own under the SONAME this library links (/opt/rocm's libamdhip64); the one
            # that is loaded first serves both, and torch does not find its device on the other one.  Every Python user of
            # this binding (tests, bench, tools) uses torch for device memory sooner or later: it goes first.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        _product = CharLSLibrary(PRODUCT_LIB)
    return _product


KNOB_UNSET = -(1 << 63)


def set_knob(name: str, value, lib: CharLSLibrary | None = None) -> None:
    """charls_amd_debug_set_knob: a test / measurement knob of the engine (charls_amd/csrc/device/knobs.h); value None clears
    it.  The environment variables of the same names are read once, when the library first looks at a knob."""
    L = (lib or load_product()).lib
    L.charls_amd_debug_set_knob.argtypes = [C.c_char_p, C.c_int64]
    L.charls_amd_debug_set_knob.restype = C.c_int32
    rc = L.charls_amd_debug_set_knob(name.encode(), KNOB_UNSET if value is None else int(value))
    if rc != 0:
        raise JpegLSError(rc, f"charls_amd_debug_set_knob({name})")


def engine_counters(lib: CharLSLibrary | None = None) -> dict:
    """charls_amd_engine_counters: what the coalescer of the host-pointer ABI did, and pipeline scans without a work area."""
    L = (lib or load_product()).lib
    L.charls_amd_engine_counters.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
    L.charls_amd_engine_counters.restype = C.c_int32
    out = (C.c_uint64 * 10)()
    n = L.charls_amd_engine_counters(out, 10)
    assert n >= 5
    names = ("calls", "launches", "merged_calls", "largest_launch", "pipeline_fallback_scans", "split_launches", "idle_pool_bytes",
             "deferred_free_bytes", "idle_releases", "exact_retry_scans")
    return dict(zip(names[:n], (int(v) for v in out[:n])))
