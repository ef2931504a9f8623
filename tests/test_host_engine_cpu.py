"""The host layer of the 48-symbol ABI without a GPU: charls_amd/csrc/host/scan_engine.cpp (resource pool, coalescer, launch streams,
housekeeping thread) compiled by g++ against a stand-in runtime (tests/host_stub: "device memory" on the heap, fake launches that
derive every scan's result from that scan's own input) and driven by 24 threads x a handle per call.  Plain, under
AddressSanitizer + UndefinedBehaviorSanitizer, and under ThreadSanitizer (-DJLS_TSAN: timed waits on the system clock, see
charls_amd/csrc/host/coalescer.h).  The reference builds its tests with sanitizers (CMakeLists.txt:53, src/CMakeLists.txt:60-63)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "host_stub")


def _run(tmp_path, name, flags, env=None):
    exe = tmp_path / name
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", *flags, "-I" + STUB, "-I" + os.path.join(ROOT, "charls_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(STUB, "engine_stress.cpp"), os.path.join(STUB, "stub_runtime.cpp"),
                           os.path.join(ROOT, "charls_amd", "csrc", "host", "scan_engine.cpp"), "-o", str(exe)])
    return subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)


def _available(flag):
    return subprocess.run(["g++", flag, "-x", "c++", "-", "-o", os.devnull], input="int main(){return 0;}", capture_output=True,
                          text=True).returncode == 0


def test_engine_under_threads(tmp_path):
    r = _run(tmp_path, "engine_plain", [])
    assert r.returncode == 0 and "engine ok" in r.stdout, r.stdout + r.stderr


def test_engine_under_address_and_undefined_behavior_sanitizers(tmp_path):
    if not _available("-fsanitize=address,undefined"):
        pytest.skip("this g++ has no libasan / libubsan")
    r = _run(tmp_path, "engine_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"],
             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))  # (the pool, the coalescer and the housekeeping state are leaked on purpose)
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "engine ok" in r.stdout, r.stdout + r.stderr


def test_engine_under_thread_sanitizer(tmp_path):
    if not _available("-fsanitize=thread"):
        pytest.skip("this g++ has no libtsan")
    r = _run(tmp_path, "engine_tsan", ["-fsanitize=thread", "-DJLS_TSAN"], env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"))
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    assert r.returncode == 0 and "engine ok" in r.stdout, r.stdout + r.stderr
