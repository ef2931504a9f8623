"""The measurement tools that include product sources still build (hipcc cross-compiles gfx950 without a GPU): the step loop's
microbenchmark instantiates the macros of charls_amd/csrc/device/scan_group_step.inc with its own compositions, so a change to
their signatures breaks it silently otherwise.  CPU only; nothing is run."""
import os
import shutil
import subprocess

import pytest

import common

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="no hipcc")
@pytest.mark.parametrize("source", ["steploop.hip", "issue_ceiling.hip", "stream_priority.hip"])
def test_microbenchmarks_compile(source, tmp_path):
    src = os.path.join(common.ROOT, "tools", "microbench", source)
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-I" + os.path.join(common.ROOT, "charls_amd", "csrc", "device"), "-c", src,
                          "-o", str(tmp_path / "probe.o")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]


def test_design_tables_tool_reads_the_committed_profiles():
    out = subprocess.run([os.environ.get("PYTHON", "python"), os.path.join(common.ROOT, "tools", "design_tables.py"), "r06"], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "| `value` |" in out.stdout and "### 6.3" in out.stdout and "near-lossless gray (6)" in out.stdout
