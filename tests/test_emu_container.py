"""container_kernels.hip on the CPU harness (kernels compiled for the host, a thread per lane): the marker search the batch
decoder uses to get from one component scan of a planar frame to the next, and the placement of scans that the batch
encoder coded into private buffers.  The GPU suite runs the same kernels through the batch API (tests/test_gpu_batch.py)."""
import ctypes as C

import numpy as np
import pytest

import emu_bind

NO_MARKER = (1 << 64) - 1
KOK, TOO_SMALL = 0, 3


def _first_marker(b: bytes, start: int, end: int) -> int:
    """The rule of the reference's reader: 0xFF followed by a byte with its high bit set that is not RSTm."""
    for q in range(start, end - 1):
        if b[q] == 0xFF and b[q + 1] >= 0x80 and not (0xD0 <= b[q + 1] <= 0xD7):
            return q
    return NO_MARKER


def _search(buf: np.ndarray, stretches):
    L = emu_bind.lib()
    pairs = np.array(stretches, dtype=np.uint64).reshape(-1, 2)
    found = np.zeros(len(pairs), dtype=np.uint64)
    L.emu_find_scan_end(buf.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p), found.ctypes.data_as(C.c_void_p),
                        C.c_int(len(pairs)))
    return [int(v) for v in found]


def test_find_scan_end_every_position_and_alignment():
    """A marker at every offset around the 64-byte shares of the threads and the 16 KB trips of the workgroup, stretches
    that start at every alignment, stuffed 0xFF bytes and restart markers before it, a 0xFF as the very last byte."""
    rng = np.random.default_rng(1)
    n = 40000
    base = rng.integers(0, 0x80, size=n, dtype=np.uint8)  # nothing that looks like a marker
    ff = rng.integers(0, n - 1, size=300)
    base[ff] = 0xFF                                        # stuffed 0xFF bytes: followed by a byte below 0x80
    base[ff + 1] &= 0x7F
    for at in list(range(0, 136)) + [16383, 16384, 16385, 16447, 16448, 32767, 32768, n - 2]:
        b = base.copy()
        b[at], b[at + 1] = 0xFF, 0xDA
        for start in sorted({0, 1, 3, 7, 63, 64, 65, max(0, at - 1), at}):
            if start > at:
                continue
            want = _first_marker(bytes(b), start, n)
            assert _search(b, [(start, n)]) == [want], (at, start)
    b = base.copy()
    b[5000], b[5001] = 0xFF, 0xD3   # a restart marker is not the end of the segment
    b[9000], b[9001] = 0xFF, 0xFF   # fill bytes before a marker: the first 0xFF ends the segment
    b[9002] = 0xD9
    assert _search(b, [(0, n), (5000, n), (9001, n), (9003, n)]) == [_first_marker(bytes(b), s, n) for s in (0, 5000, 9001, 9003)]
    b = base.copy()
    b[n - 1] = 0xFF                 # no follower: not a marker
    assert _search(b, [(0, n), (n - 1, n), (n, n), (17, 18)]) == [NO_MARKER] * 4


def test_find_scan_end_many_stretches_of_one_buffer():
    rng = np.random.default_rng(2)
    n = 70000
    b = rng.integers(0, 256, size=n, dtype=np.uint8)       # random bytes: markers everywhere
    stretches = [(int(s), int(min(n, s + l))) for s, l in zip(rng.integers(0, n, size=40), rng.integers(0, 40000, size=40))]
    assert _search(b, stretches) == [_first_marker(bytes(b), s, e) for s, e in stretches]


class Cursor(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("errc", C.c_uint32), ("pad", C.c_uint32)]


@pytest.mark.parametrize("seed", range(4))
def test_place_plane_scans_against_a_model(seed):
    """Scans of random sizes in private buffers -> behind their headers in the frames' slots: every byte of the slots, the
    cursors and the again-flags against a model of the walk (a scan that failed or that would leave fewer than 4 bytes
    behind it: the frame is flagged and its cursor left alone; a header that does not fit: destination too small)."""
    L = emu_bind.lib()
    assert L.emu_sizeof_scan_result() == C.sizeof(emu_bind.ScanResult)
    rng = np.random.default_rng(seed)
    frames, rounds, header = 7, int(rng.integers(2, 5)), int(rng.integers(10, 15))
    capacity, pitch = 2048, 4000
    priv = rng.integers(0, 256, size=frames * rounds * capacity, dtype=np.uint8)
    headers = rng.integers(0, 256, size=rounds * header, dtype=np.uint8)
    results = (emu_bind.ScanResult * (frames * rounds))()
    cursors = (Cursor * frames)()
    for f in range(frames):
        cursors[f] = Cursor(int(rng.integers(20, 60)), 0, 0)
        for r in range(rounds):
            results[f * rounds + r] = emu_bind.ScanResult(KOK, 0, int(rng.integers(0, 1500)))
    results[1 * rounds + 1].errc = 5                         # a scan that failed
    results[2 * rounds + 0].bytes = pitch                    # a scan that cannot fit its slot
    cursors[3].errc = 9                                      # a frame that failed before its scans
    cursors[4].offset = pitch - header + 1                   # no room for the first header
    results[5 * rounds + 0].bytes = pitch - cursors[5].offset - header - 3  # leaves 3 bytes: inside the zone of the exact verdict
    results[6 * rounds + 0].bytes = 17                       # (bytes that are not a multiple of 16: the tail of the copy)
    slots = np.full(frames * pitch, 0xEE, dtype=np.uint8)
    want = slots.copy()
    want_cursors, want_redo = [], []
    for f in range(frames):
        off, errc, again, placed = int(cursors[f].offset), int(cursors[f].errc), False, []
        for r in range(rounds):
            if errc != KOK or again:
                break
            if off + header > pitch:
                errc = TOO_SMALL
                break
            res = results[f * rounds + r]
            if res.errc != KOK or res.bytes + 4 > pitch - off - header:
                again = True
                break
            placed.append((r, off, int(res.bytes)))
            off += header + int(res.bytes)
        for r, at, nbytes in ([] if again else placed):
            want[f * pitch + at:f * pitch + at + header] = headers[r * header:(r + 1) * header]
            src = (f * rounds + r) * capacity
            want[f * pitch + at + header:f * pitch + at + header + nbytes] = priv[src:src + nbytes]
        want_cursors.append((int(cursors[f].offset), int(cursors[f].errc)) if again else (off, errc))
        want_redo.append(1 if again else 0)
    redo = np.full(frames, 7, dtype=np.uint32)
    L.emu_place_plane_scans(slots.ctypes.data_as(C.c_void_p), C.c_uint64(pitch), headers.ctypes.data_as(C.c_void_p), C.c_uint32(header),
                            C.c_uint32(rounds), priv.ctypes.data_as(C.c_void_p), C.c_uint64(capacity), results, cursors,
                            redo.ctypes.data_as(C.c_void_p), C.c_uint32(frames))
    assert list(redo) == want_redo
    assert [(int(c.offset), int(c.errc)) for c in cursors] == want_cursors
    assert np.array_equal(slots, want)
