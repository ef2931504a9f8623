"""Which paths of scan_group_decode.hip a decode takes (CPU harness with the kernel's path counters on, tests/emu/
emu_profile_driver.cpp): the 128-bit refill and the run service of the step loop must be the paths that run on ordinary
streams, with the byte-by-byte refill and the handlers behind the loop left for what they are kept for.  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import common
import emu_bind
import jls_container
import oracle_bind as ob
from charls_amd import synth
from test_emu_serial_kernels import _stream_copy

ROUNDS, REFILLS, STEP_LOOPS, STEPS, GENERAL_RUNS, BYTEWISE, DELETE_TRIPS, PREPARES, WINDOWED_RUNS, SERVED_IN_LOOP = 0, 1, 3, 5, 6, 11, 12, 14, 15, 16


def _decode(frames, width, height, group):
    L = emu_bind.profile_lib()
    keep, outs, descs = [], [], []
    for img in frames:
        jls = ob.encode(img, width=width, height=height, bits_per_sample=8)
        cont = jls_container.parse(jls)
        pc = jls_container.validated_pc(cont.pc, cont.bits, 0)
        pix = np.zeros(width * height, dtype=np.uint8)
        descs.append(emu_bind.make_desc(width, height, 1, 0, 8, 0, 0, pc, 0, pix, width, _stream_copy(jls, cont.scans[0].data_start), keep))
        outs.append(pix)
    n = len(descs)
    arr = (emu_bind.ScanDesc * n)(*descs)
    res = (emu_bind.ScanResult * n)()
    counts = (C.c_ulonglong * 32)()
    assert L.emu_profile_decode_group(arr, res, n, group, counts) == 0
    for r, o, img in zip(res, outs, frames):
        assert (r.errc, r.flags) == (0, 0)
        assert o.tobytes() == np.ascontiguousarray(img).tobytes()
    return list(counts)


@pytest.mark.parametrize("group", [8, 16, 32])
def test_noise_streams_are_unstuffed_by_the_128_bit_refill(group):
    """Noise compresses to nothing: long streams with a 0xFF in every 256 bytes.  Apart from the first bytes of a misaligned
    stream and the bytes around its end every refill is the 128-bit one, and it deletes stuffed bits."""
    w, h = 384, 24
    frames = [synth.frame_numpy(w, h, seed=300 + f, bits=8, kind="noise") for f in range(64 // group)]
    c = _decode(frames, w, h, group)
    assert c[REFILLS] >= 4
    assert c[DELETE_TRIPS] > 0, "no stuffed bit met the 128-bit refill: the test frames are too small"
    assert c[BYTEWISE] <= 4, "the byte-by-byte refill is for the ends of a stream"
    assert c[BYTEWISE] < c[REFILLS]


def test_short_runs_are_served_inside_the_step_loop():
    """The bench's frames and a natural image: nearly every run event -- runs of length 0 first of all -- is served by the step
    loop itself (round 6), so that a wavefront enters the loop once per ~ 50 samples instead of once per event; what leaves the
    loop comes out of one 64-bit window, and the bit-by-bit handler is for runs to the end of a line."""
    w, h = 1024, 6
    frames = [synth.frame_numpy(w, h, seed=1000 + f, bits=8, kind="gradient") for f in range(8)]
    c = _decode(frames, w, h, 8)
    assert c[STEPS] > 0 and c[ROUNDS] > 0
    assert c[SERVED_IN_LOOP] > 8 * (c[WINDOWED_RUNS] + c[GENERAL_RUNS]) and c[SERVED_IN_LOOP] > 0
    assert c[STEPS] > 20 * c[STEP_LOOPS], "a wavefront leaves the loop for events the loop should serve"
    assert 0 < c[PREPARES] <= c[ROUNDS]
    tile, _ = common.read_pnm("tulips-gray-8bit-512-512.pgm")
    frames = [np.ascontiguousarray(np.roll(tile, 31 * f, axis=1)[40 * f:40 * f + 8]) for f in range(4)]
    c = _decode(frames, 512, 8, 16)
    assert c[SERVED_IN_LOOP] > 8 * (c[WINDOWED_RUNS] + c[GENERAL_RUNS])


def test_flat_frames_never_take_it():
    """All zero (the value a frame's surroundings have): every line is one run to its end, nothing is interrupted -- the one
    case the windowed handler leaves to the bit-by-bit one."""
    w, h = 200, 9
    c = _decode([np.zeros((h, w), dtype=np.uint8) for _ in range(2)], w, h, 32)
    assert c[SERVED_IN_LOOP] == 0 and c[WINDOWED_RUNS] == 0 and c[GENERAL_RUNS] > 0


def test_the_path_profile_tool_runs():
    out = subprocess.run([sys.executable, os.path.join(common.ROOT, "tools", "decode_path_profile.py"), "--width", "256", "--lines", "4",
                          "--group", "16"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "run services inside the step loop" in out.stdout and "refills" in out.stdout
