"""The coalescer of the host-pointer ABI (charls_amd/csrc/host/coalescer.h) with a fake launch: merged launches, keys and
lanes kept apart, the exclusive lane's group commit, failures, retractions, the cap.  Pure C++, no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_coalescer_merges_announced_calls(tmp_path):
    exe = tmp_path / "coalescer_test"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "charls_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "coalescer", "coalescer_test.cpp"),
                           "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "coalescer ok" in r.stdout and "FAILED" not in r.stdout, r.stdout + r.stderr
