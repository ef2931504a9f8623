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
# The same objects linked under the reference's SONAME (src/CMakeLists.txt:65-67: libcharls.so.3), so that a program
# linked with -lcharls against CharLS finds this library when charls_amd/lib precedes CharLS on its library path.
OUT_ALIAS = os.path.join(OUT_DIR, "libcharls.so.3")
VERSION_SCRIPT = os.path.join(CSRC, "charls_amd.version")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# translation units; the *.hip kernel sources are #included by the launch code that instantiates them (runtime.hip, and
# units of their own, one per sample width, for the kernels with the most instantiations, so that the build stays parallel)
SOURCES = [
    "device/runtime.hip",
    "device/launch_pixels_u8.hip",
    "device/launch_pixels_u16.hip",
    "device/launch_group_encode_u8.hip",
    "device/launch_group_encode_u16.hip",
    "host/stream_reader.cpp",
    "host/scan_engine.cpp",
    "host/encoder_api.cpp",
    "host/decoder_api.cpp",
    "host/misc_api.cpp",
    "host/batch_api.cpp",
    "host/multi_device.cpp",
]


def _newest_source() -> float:
    files = glob.glob(os.path.join(CSRC, "**", "*"), recursive=True) + [os.path.join(ROOT, "include", "charls_amd.h")]
    return max(os.path.getmtime(f) for f in files if os.path.isfile(f))


LLVM_BIN = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
OFFLOAD_ARCH = "gfx950"  # the one target of this library (also the bundle id the scratch check unpacks)


def kernel_resources(obj: str) -> list[dict]:
    """Kernels of one hipcc object with what their code-object notes say about them: name, private segment (scratch) bytes,
    VGPRs, SGPRs, static LDS.  The gfx950 code object is cut out of the object's .hip_fatbin section."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        fat, dev = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        subprocess.check_call([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--" + OFFLOAD_ARCH, "--output=" + dev])
        notes = subprocess.check_output([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", dev]).decode()
    kernels, cur = [], None
    keys = {".private_segment_fixed_size": "scratch", ".vgpr_count": "vgprs", ".sgpr_count": "sgprs",
            ".group_segment_fixed_size": "lds", ".name": "name"}
    for line in notes.splitlines():
        line = line.strip()
        if line.startswith("- .") or line.startswith("- .agpr_count") or line.startswith("- .args"):
            if line.startswith("- .agpr_count") or line.startswith("- .args"):
                cur = {}
                kernels.append(cur)
            line = line[2:]
        if cur is None or ":" not in line:
            continue
        key, value = line.split(":", 1)
        if key in keys:
            value = value.strip()
            cur[keys[key]] = value if key == ".name" else int(value)
    return [k for k in kernels if "name" in k and "scratch" in k]


# Kernels that may keep a private segment: none.  Every kernel of the library is on some speed path or is the exact
# fallback of one; a private segment means a register array that is indexed dynamically or a lambda the compiler did not
# inline (the scan state of a group kernel in scratch, green on the CPU harness and a fault on the GPU).
SCRATCH_ALLOWED: tuple[str, ...] = ()


def check_no_scratch(objs: list[str]) -> list[dict]:
    """Raises when a kernel of the library has a private segment; returns the resource table otherwise."""
    table = []
    for obj in objs:
        if not obj.endswith(".hip.o"):
            continue
        table.extend(kernel_resources(obj))
    if not table:
        raise RuntimeError("no kernels found in the device objects: the scratch check cannot see them")
    offenders = [k for k in table if k["scratch"] > 0 and not any(a in k["name"] for a in SCRATCH_ALLOWED)]
    if offenders:
        raise RuntimeError("kernels with a private segment (scratch): " +
                           ", ".join(f"{k['name']} ({k['scratch']} B)" for k in offenders))
    return table


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(OUT) and os.path.exists(OUT_ALIAS) and os.path.getmtime(OUT) >= _newest_source():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    obj_dir = os.path.join(OUT_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    common = ["--offload-arch=" + OFFLOAD_ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wextra",
              "-Wno-unused-parameter", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
              *os.environ.get("CHARLS_AMD_CXXFLAGS", "").split()]  # (e.g. -DJLS_PHASE_CLOCKS, tools/phase_clocks.py)
    objs = []
    procs = []
    # An object is compiled again when anything it can include is newer: the device units include device/ only, the host
    # units host/, the headers of device/ and the public header (a change to a kernel does not recompile the facade, a change
    # to the facade not the kernels -- the five device units take minutes).  CHARLS_AMD_CXXFLAGS always recompile.
    def newest(paths):
        return max(os.path.getmtime(f) for f in paths if os.path.isfile(f))
    device_files = glob.glob(os.path.join(CSRC, "device", "*"))
    host_deps = glob.glob(os.path.join(CSRC, "host", "*")) + [f for f in device_files if f.endswith(".h")] + \
        [os.path.join(ROOT, "include", "charls_amd.h"), __file__]
    device_newest, host_newest = newest(device_files + [__file__]), newest(host_deps)
    flags_given = bool(os.environ.get("CHARLS_AMD_CXXFLAGS", "").split())
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace("/", "_") + ".o")
        objs.append(obj)
        stale_after = device_newest if src.startswith("device/") else host_newest
        if not flags_given and os.path.exists(obj) and os.path.getmtime(obj) >= stale_after:
            continue
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
    # The scratch check needs llvm-objcopy / clang-offload-bundler / llvm-readelf of the ROCm LLVM (ROCM_LLVM_BIN).  A kernel with a
    # private segment always fails the build; a toolchain without those tools only loses the check (a warning), and
    # CHARLS_AMD_SKIP_SCRATCH_CHECK=1 skips it.
    if os.environ.get("CHARLS_AMD_SKIP_SCRATCH_CHECK") == "1":
        sys.stderr.write("scratch check skipped (CHARLS_AMD_SKIP_SCRATCH_CHECK=1)\n")
    else:
        try:
            table = check_no_scratch(objs)
            if verbose:
                sys.stderr.write(f"scratch check: {len(table)} kernels, none with a private segment\n")
        except (FileNotFoundError, subprocess.CalledProcessError) as e:
            sys.stderr.write(f"warning: the scratch check could not run ({e}); set ROCM_LLVM_BIN to the ROCm LLVM tools\n")
    for out, soname in ((OUT, "libcharls_amd.so"), (OUT_ALIAS, "libcharls.so.3")):
        link = [HIPCC, "--offload-arch=" + OFFLOAD_ARCH, "-shared", "-fPIC", *objs, "-o", out, "-Wl,-soname," + soname,
                "-Wl,--no-undefined", "-Wl,--version-script=" + VERSION_SCRIPT, "-ldl"]
        subprocess.check_call(link)
    dev_link = os.path.join(OUT_DIR, "libcharls.so")  # what -lcharls resolves at link time
    if os.path.lexists(dev_link):
        os.remove(dev_link)
    os.symlink("libcharls.so.3", dev_link)
    return OUT


C_CALLER_SRC = os.path.join(ROOT, "tests", "c_caller", "roundtrip.c")
C_CALLER_DIR = os.path.join(ROOT, "tests", "c_caller", "build")


def build_c_callers(reference_include: str = "/root/reference/include") -> list[str]:
    """tests/c_caller/roundtrip.c compiled with gcc as C99 against this repository's header and -- where the reference tree
    is present -- against the reference's own <charls/charls.h>, both linked with -lcharls (SONAME libcharls.so.3 of
    charls_amd/lib).  The link test of INTEGRATION.md; the binaries travel to the GPU box and run in smoke()."""
    os.makedirs(C_CALLER_DIR, exist_ok=True)
    outs = []
    variants = [("roundtrip_own_header", ["-I" + os.path.join(ROOT, "include")])]
    if os.path.isdir(os.path.join(reference_include, "charls")):
        variants.append(("roundtrip_reference_headers", ["-DUSE_REFERENCE_HEADERS", "-I" + reference_include]))
    for name, flags in variants:
        out = os.path.join(C_CALLER_DIR, name)
        subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", *flags, C_CALLER_SRC, "-o", out, "-L" + OUT_DIR,
                               "-lcharls", "-Wl,-rpath,$ORIGIN/../../../charls_amd/lib"])
        outs.append(out)
    return outs


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
