"""ctypes binding of oracle/_build/libjls_oracle.so (the CPU restatement) -- test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "libjls_oracle.so")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libcharls_ref.so")


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bits_per_sample", C.c_int32),
                ("component_count", C.c_int32), ("near_lossless", C.c_int32), ("interleave_mode", C.c_int32),
                ("color_transformation", C.c_int32), ("maximum_sample_value", C.c_int32), ("threshold1", C.c_int32),
                ("threshold2", C.c_int32), ("threshold3", C.c_int32), ("reset_value", C.c_int32),
                ("encoding_options", C.c_uint32), ("restart_interval", C.c_uint32)]


class OracleError(RuntimeError):
    def __init__(self, errc):
        self.errc = int(errc)
        super().__init__(f"oracle errc={errc}")


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "jls_oracle.c")
        if (not os.path.exists(ORACLE_LIB)) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        L = C.CDLL(ORACLE_LIB)
        L.jls_oracle_encode.argtypes = [C.POINTER(Params), C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_size_t)]
        L.jls_oracle_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(Params)]
        L.jls_oracle_read_header.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Params)]
        L.jls_oracle_default_pc.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        L.jls_oracle_bitwriter_kat.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_int, C.c_void_p,
                                               C.c_size_t, C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def encode(image, *, width, height, bits_per_sample=8, component_count=1, near_lossless=0, interleave_mode=0,
           color_transformation=0, preset=None, encoding_options=0, stride=0, destination_size=None) -> bytes:
    a = np.ascontiguousarray(image) if isinstance(image, np.ndarray) else np.frombuffer(bytes(image), dtype=np.uint8)
    pc = preset or (0, 0, 0, 0, 0)
    p = Params(width, height, bits_per_sample, component_count, near_lossless, interleave_mode, color_transformation,
               pc[0], pc[1], pc[2], pc[3], pc[4], encoding_options, 0)
    if destination_size is None:
        raw = width * height * component_count * ((bits_per_sample + 7) // 8)
        destination_size = raw + raw // 16 + 1024 + 34
    dst = np.empty(destination_size, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().jls_oracle_encode(C.byref(p), a.ctypes.data, a.nbytes, stride, dst.ctypes.data, dst.nbytes, C.byref(n))
    if rc:
        raise OracleError(rc)
    return dst[:n.value].tobytes()


def read_header(data) -> Params:
    b = np.frombuffer(bytes(data), dtype=np.uint8)
    p = Params()
    rc = lib().jls_oracle_read_header(b.ctypes.data, b.nbytes, C.byref(p))
    if rc:
        raise OracleError(rc)
    return p


def decode(data, stride=0):
    b = np.frombuffer(bytes(data), dtype=np.uint8)
    p = read_header(data)
    bytes_ps = (p.bits_per_sample + 7) // 8
    if stride == 0:
        size = p.width * p.height * p.component_count * bytes_ps
    elif p.interleave_mode == 0:
        size = stride * p.component_count * p.height
    else:
        size = stride * p.height
    out = np.zeros(size, dtype=np.uint8)
    q = Params()
    rc = lib().jls_oracle_decode(b.ctypes.data, b.nbytes, out.ctypes.data, out.nbytes, stride, C.byref(q))
    if rc:
        raise OracleError(rc)
    return q, out
