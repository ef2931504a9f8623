"""Builds charls_amd/lib/libcharls_amd.so (host C++ facade + gfx950 kernels) with hipcc.  No GPU needed to build."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "libcharls_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# translation units; the *.hip kernel sources are #included by runtime.hip so that launches and kernels share a TU
SOURCES = [
    "device/runtime.hip",
    "host/stream_reader.cpp",
    "host/scan_engine.cpp",
    "host/encoder_api.cpp",
    "host/decoder_api.cpp",
    "host/misc_api.cpp",
    "host/batch_api.cpp",
]


def _newest_source() -> float:
    files = glob.glob(os.path.join(CSRC, "**", "*"), recursive=True) + [os.path.join(ROOT, "include", "charls_amd.h")]
    return max(os.path.getmtime(f) for f in files if os.path.isfile(f))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest_source():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    obj_dir = os.path.join(OUT_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wextra",
              "-Wno-unused-parameter", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace("/", "_") + ".o")
        objs.append(obj)
        cmd = [HIPCC, *common, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src}\n{out.decode()}\n")
        elif verbose and out:
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT, "-Wl,-soname,libcharls_amd.so",
            "-Wl,--no-undefined"]
    subprocess.check_call(link)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
