"""tools/jls_bench.py -- the reference's `charls-cli benchmark-encode / benchmark-decode` (cli/benchmark.cpp:32-90) for this
engine -- is run as a program: both ways in (host-pointer ABI, batch API), and the host-ABI loop on the reference's own
library next to it.  What is checked is that the tool runs, verifies its own round trip and reports the reference's
quantities; the numbers themselves are printed for the log.  GPU only."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "jls_bench.py")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libcharls_ref.so")


def _run(*args):
    out = subprocess.run([sys.executable, TOOL, *args, "--json"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert any(l.startswith("Total encoding time is:") for l in lines) and any(l.startswith("Total decoding time is:") for l in lines)
    return json.loads(lines[-1])


def test_host_abi_loop_on_config_1_for_both_libraries():
    ours = _run("--config", "1", "--abi", "host", "--loop", "2")
    assert ours["abi"] == "host" and ours["loop"] == 2 and ours["encoded_bytes"] == 7295349
    print(f"\n[jls_bench] config 1 host ABI, this engine: encode {ours['encode']['ms_per_image']} ms, decode {ours['decode']['ms_per_image']} ms")
    if os.path.exists(REF_LIB):
        ref = _run("--config", "1", "--abi", "host", "--loop", "2", "--library", REF_LIB)
        assert ref["encoded_bytes"] == ours["encoded_bytes"]  # the same .jls size from both libraries
        print(f"[jls_bench] config 1 host ABI, reference on the host CPU: encode {ref['encode']['ms_per_image']} ms, "
              f"decode {ref['decode']['ms_per_image']} ms")


def test_batch_api_loop_on_config_1_and_3():
    one = _run("--config", "1", "--abi", "batch", "--frames", "32", "--loop", "2")
    assert one["frames"] == 32 and one["encode"]["mpix_s"] > 0 and one["decode"]["mpix_s"] > 0
    three = _run("--config", "3", "--abi", "batch", "--frames", "64", "--loop", "2")
    assert three["frames"] == 64
    print(f"\n[jls_bench] batch API: config 1 x 32 frames encode {one['encode']['mpix_s']} / decode {one['decode']['mpix_s']} MPix/s; "
          f"config 3 x 64 frames encode {three['encode']['mpix_s']} / decode {three['decode']['mpix_s']} MPix/s")
