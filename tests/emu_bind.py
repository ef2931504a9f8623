"""Builds and loads tests/_emu_build/libjls_emu.so: the gfx950 kernel sources compiled for the host against the SIMT
emulation header (tests/emu).  TEST-ONLY -- lets the CPU suite exercise kernel logic; never used by the product."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
OUT = os.path.join(ROOT, "tests", "_emu_build", "libjls_emu.so")
DEV = os.path.join(ROOT, "charls_amd", "csrc", "device")


class ScanDesc(C.Structure):  # must mirror charls_amd/csrc/device/scan_types.h
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("components", C.c_int32),
                ("interleave_mode", C.c_int32), ("bits_per_sample", C.c_int32), ("near_lossless", C.c_int32),
                ("color_transformation", C.c_int32), ("t1", C.c_int32), ("t2", C.c_int32), ("t3", C.c_int32),
                ("reset", C.c_int32), ("restart_interval", C.c_uint32), ("pixels", C.c_void_p),
                ("pixel_stride", C.c_uint64), ("stream", C.c_void_p), ("stream_capacity", C.c_uint64),
                ("line_scratch", C.c_void_p)]


class ScanResult(C.Structure):
    _fields_ = [("errc", C.c_uint32), ("flags", C.c_uint32), ("bytes", C.c_uint64)]


_lib = None
_libs = {}


def _build_and_load(driver, out):
    srcs = glob.glob(os.path.join(EMU_DIR, "*")) + glob.glob(os.path.join(EMU_DIR, "hip", "*")) + \
        glob.glob(os.path.join(DEV, "*"))
    newest = max(os.path.getmtime(s) for s in srcs)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    import fcntl
    with open(out + ".lock", "w") as lock:  # pytest-xdist workers must not rebuild / replace the library concurrently
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not os.path.exists(out) or os.path.getmtime(out) < newest:
            tmp = out + f".{os.getpid()}.tmp"
            subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-DJLS_KNOBS_LIVE_ENV", "-I" + EMU_DIR,
                                   "-I" + DEV, "-x", "c++", os.path.join(EMU_DIR, driver), "-o", tmp])
            os.replace(tmp, out)
    L = C.CDLL(out)
    assert L.emu_sizeof_scan_desc() == C.sizeof(ScanDesc)
    return L


def tile_lib():
    """The tile pipeline's kernels alone (a translation unit of its own: builds in seconds)."""
    if "tile" not in _libs:
        _libs["tile"] = _build_and_load("emu_tile_driver.cpp", os.path.join(ROOT, "tests", "_emu_build", "libjls_emu_tile.so"))
    return _libs["tile"]


def profile_lib():
    """scan_group_decode.hip with its path counters on (tools/decode_path_profile.py)."""
    if "profile" not in _libs:
        _libs["profile"] = _build_and_load("emu_profile_driver.cpp", os.path.join(ROOT, "tests", "_emu_build", "libjls_emu_profile.so"))
    return _libs["profile"]


def lib():
    global _lib
    if _lib is None:
        srcs = glob.glob(os.path.join(EMU_DIR, "*")) + glob.glob(os.path.join(EMU_DIR, "hip", "*")) + \
            glob.glob(os.path.join(DEV, "*"))
        newest = max(os.path.getmtime(s) for s in srcs)
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        import fcntl
        with open(OUT + ".lock", "w") as lock:  # pytest-xdist workers must not rebuild / replace the library concurrently
            fcntl.flock(lock, fcntl.LOCK_EX)
            if not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
                tmp = OUT + f".{os.getpid()}.tmp"
                subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-DJLS_KNOBS_LIVE_ENV", "-I" + EMU_DIR,
                                       "-I" + DEV, "-x", "c++", os.path.join(EMU_DIR, "emu_driver.cpp"), "-o", tmp])
                os.replace(tmp, OUT)
        L = C.CDLL(OUT)
        assert L.emu_sizeof_scan_desc() == C.sizeof(ScanDesc)
        _lib = L
    return _lib


def make_desc(width, height, comps, ilv, bits, near, xform, pc, restart, pixels: np.ndarray, pixel_stride,
              stream: np.ndarray, keep: list):
    planes = 1 if ilv == 0 else comps
    scratch = np.zeros(2 * planes * (width + 2), dtype=np.uint16)
    keep.extend([scratch, pixels, stream])
    return ScanDesc(width, height, comps, ilv, bits, near, xform, pc[1], pc[2], pc[3], pc[4] & 0xFF, restart,
                    pixels.ctypes.data, pixel_stride, stream.ctypes.data, stream.nbytes, scratch.ctypes.data)
