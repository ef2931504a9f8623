"""The link test behind INTEGRATION.md: a C99 program written against the CharLS C API (tests/c_caller/roundtrip.c),
compiled against the REFERENCE's own headers, linked with -lcharls and run on libcharls_amd.so through its
libcharls.so.3 SONAME.  The same binary runs on the reference itself when that is first on the library path."""
import os
import subprocess

import pytest

import common

BUILD = os.path.join(common.ROOT, "tests", "c_caller", "build")
LIB_DIR = os.path.join(common.ROOT, "charls_amd", "lib")
REF = os.path.join(common.ROOT, "oracle", "_ref", "libcharls_ref.so")
BINARIES = ["roundtrip_own_header", "roundtrip_reference_headers"]


@pytest.fixture(scope="module")
def binaries():
    from charls_amd import build
    build.build()
    have = [os.path.join(BUILD, b) for b in BINARIES if os.path.exists(os.path.join(BUILD, b))]
    if os.path.isdir("/root/reference/include/charls") or not have:
        have = build.build_c_callers()
    return have


def _run(path, env=None):
    return subprocess.run([path], capture_output=True, text=True, timeout=300, env=env)


def test_library_carries_the_reference_soname_and_only_the_c_api():
    from charls_amd import build, capi
    build.build()
    dyn = subprocess.run(["readelf", "-d", os.path.join(LIB_DIR, "libcharls.so.3")], capture_output=True, text=True).stdout
    assert "Library soname: [libcharls.so.3]" in dyn  # reference src/CMakeLists.txt:65-67
    names = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIB_DIR, "libcharls.so.3")], capture_output=True,
                           text=True).stdout.split("\n")
    exported = {line.split()[-1] for line in names if line.strip()}
    assert exported and all(n.startswith("charls_") for n in exported)  # the version script (src/charls.version:1-21)
    assert set(capi.all_abi_symbols()) <= exported


def test_c_caller_links_against_the_soname(binaries):
    assert len(binaries) >= 1
    for b in binaries:
        dyn = subprocess.run(["readelf", "-d", b], capture_output=True, text=True).stdout
        assert "Shared library: [libcharls.so.3]" in dyn, b


def test_c_caller_fails_loudly_without_a_gpu(binaries):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: see test_c_caller_round_trips_on_the_gpu")
    for b in binaries:
        r = _run(b)
        assert r.returncode == 3 and "errc 200" in r.stdout, r.stdout


def test_the_same_binary_runs_on_the_reference(binaries, tmp_path):
    """LD_LIBRARY_PATH beats the binary's RUNPATH: with the reference under the name libcharls.so.3 the program is an
    ordinary CharLS application (so what it checks is CharLS behaviour, not something this library made up)."""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libcharls_ref.so not built")
    os.symlink(REF, tmp_path / "libcharls.so.3")
    for b in binaries:
        r = _run(b, env=dict(os.environ, LD_LIBRARY_PATH=str(tmp_path)))
        assert r.returncode == 0 and "c_caller ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_caller_round_trips_on_the_gpu():
    have = [os.path.join(BUILD, b) for b in BINARIES if os.path.exists(os.path.join(BUILD, b))]
    assert have, "tests/c_caller/build/* did not travel (built by __graft_entry__.build())"
    for b in have:
        r = _run(b)
        assert r.returncode == 0 and "c_caller ok" in r.stdout, r.stdout + r.stderr
