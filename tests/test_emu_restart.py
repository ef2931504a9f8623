"""Kernel-logic check on the CPU for restart_intervals.hip (compiled for the host by tests/emu): the marker search on
hand-made byte strings and on a reference fixture, interval descriptors, the acceptance check and the encode-side join."""
import ctypes as C

import numpy as np
import pytest

import common
import emu_bind
import jls_container


def _desc(stream: np.ndarray, keep, *, width=8, height=16, restart=4, offset=0):
    pix = np.zeros(width * height, dtype=np.uint8)
    d = emu_bind.make_desc(width, height, 1, 0, 8, 0, 0, (255, 3, 7, 21, 64), restart, pix, width, stream, keep)
    d.stream = stream.ctypes.data + offset
    d.stream_capacity = stream.nbytes - offset
    return d


def _find(data: bytes, max_marks: int, offset=0):
    L = emu_bind.lib()
    keep = []
    buf = np.frombuffer(bytes(offset) + data + bytes(64), dtype=np.uint8).copy()
    d = _desc(buf, keep, offset=offset)
    d.stream_capacity = len(data)
    marks = np.full(max(max_marks, 1), 0xEEEEEEEE, dtype=np.uint32)
    counts = np.zeros(1, dtype=np.uint32)
    L.emu_find_restart_markers(C.byref(d), marks.ctypes.data_as(C.c_void_p), C.c_uint32(max_marks),
                               counts.ctypes.data_as(C.c_void_p), 1)
    return int(counts[0]), [int(m) for m in marks[:min(int(counts[0]), max_marks)]] if counts[0] != 0xFFFFFFFF else None


def _python_find(data: bytes):
    out = []
    for i in range(len(data) - 1):
        if data[i] == 0xFF and data[i + 1] >= 0x80:
            if 0xD0 <= data[i + 1] <= 0xD7:
                out.append(i)
            else:
                break
    return out


@pytest.mark.parametrize("offset", [0, 1, 7, 15])
def test_marker_search_matches_a_byte_loop(offset):
    rng = np.random.default_rng(offset)
    body = bytearray(rng.integers(0, 255, size=5000, dtype=np.uint8).tobytes().replace(b"\xff", b"\x7f"))
    # markers at awkward places: start, 16-byte and 1024-byte boundaries, back to back, FF 00 / FF 7F stuffing look-alikes
    for pos, code in [(0, 0xD0), (15, 0xD1), (16 + 15, 0xD2), (1022, 0xD3), (1024, 0xD4), (2047, 0xD5), (2049, 0xD6), (3000, 0xD7),
                      (3002, 0xD0)]:
        body[pos:pos + 2] = bytes([0xFF, code])
    body[4000:4002] = b"\xff\x00"
    body[4100:4102] = b"\xff\x7f"
    body[4500:4502] = b"\xff\xd9"  # EOI ends the segment
    body[4600:4602] = b"\xff\xd1"  # not counted any more
    want = _python_find(bytes(body))
    assert len(want) == 9
    assert _find(bytes(body), 16, offset) == (9, want)
    assert _find(bytes(body), 9, offset) == (9, want)
    assert _find(bytes(body), 8, offset)[0] == 0xFFFFFFFF  # more markers than the caller expects
    assert _find(bytes(body[:1]), 4, offset) == (0, [])
    assert _find(b"\xff\xff\xd0" + bytes(body), 4, offset) == (0, [])  # a fill byte is "another marker": sequential decoder


@pytest.mark.parametrize("name", ["test8_ilv_sample_rm_7.jls", "test16_rm_5.jls"])
def test_marker_search_on_reference_fixture(name):
    jls = common.refdata(name)
    cont = jls_container.parse(jls)
    scan = cont.scans[0]
    body = jls[scan.data_start:]
    want = _python_find(body)
    n_int = (cont.height + cont.restart_interval - 1) // cont.restart_interval
    assert len(want) == n_int - 1
    assert _find(body, n_int - 1, 3) == (n_int - 1, want)


def test_interval_descriptors_check_and_join():
    L = emu_bind.lib()
    keep = []
    # a scan of 10 lines, intervals of 4 lines -> 3 intervals; "coded" bytes are arbitrary here
    pieces = [bytes([1, 2, 3, 4, 5]), bytes([9] * 70), bytes([7, 7])]
    body = pieces[0] + b"\xff\xd0" + pieces[1] + b"\xff\xd1" + pieces[2] + b"\xff\xd9"
    buf = np.frombuffer(body + bytes(64), dtype=np.uint8).copy()
    parent = _desc(buf, keep, width=8, height=10, restart=4)
    parent.stream_capacity = len(body)
    marks = np.array([5, 5 + 2 + 70], dtype=np.uint32)
    subs = (emu_bind.ScanDesc * 3)()
    L.emu_build_decode_intervals(C.byref(parent), marks.ctypes.data_as(C.c_void_p), C.c_uint32(3), subs, 1)
    assert [s.height for s in subs] == [4, 4, 2]
    assert [s.stream - parent.stream for s in subs] == [0, 7, 79]
    assert [s.stream_capacity for s in subs] == [7, 72, len(body) - 79]  # each window includes its terminating marker
    assert [s.pixels - parent.pixels for s in subs] == [0, 4 * 8, 8 * 8] and all(s.restart_interval == 0 for s in subs)

    def check(sub_bytes, sub_flags=(0, 0, 0), sub_errc=(0, 0, 0)):
        sr = (emu_bind.ScanResult * 3)()
        for j in range(3):
            sr[j].errc, sr[j].flags, sr[j].bytes = sub_errc[j], sub_flags[j], sub_bytes[j]
        res = (emu_bind.ScanResult * 1)()
        L.emu_check_intervals(C.byref(parent), marks.ctypes.data_as(C.c_void_p), C.c_uint32(3), sr, res, 1)
        return res[0].errc, res[0].flags, res[0].bytes

    assert check((5, 70, 2)) == (0, 0, 79 + 2)              # consumed = offset of the scan's terminating marker
    assert check((5, 69, 2)) == (0, 8, 0)                   # an interval that stopped short of its marker
    assert check((5, 70, 2), sub_flags=(0, 4, 0)) == (0, 8, 0)
    assert check((5, 70, 2), sub_flags=(1, 1, 1)) == (0, 0, 81)  # 1 = "decoded by the exact wave decoder"
    assert check((5, 70, 2), sub_errc=(0, 0, 5)) == (0, 8, 0)
    buf[5 + 1] = 0xD3                                       # wrong marker number
    assert check((5, 70, 2)) == (0, 8, 0)

    # ---- encode side: join three privately coded intervals
    priv = [np.frombuffer(p + bytes(32), dtype=np.uint8).copy() for p in pieces]
    keep.extend(priv)
    out = np.zeros(128, dtype=np.uint8)
    parent2 = _desc(out, keep, width=8, height=10, restart=4)
    subs2 = (emu_bind.ScanDesc * 3)()
    sr = (emu_bind.ScanResult * 3)()
    for j in range(3):
        subs2[j] = _desc(priv[j], keep)
        sr[j].errc, sr[j].flags, sr[j].bytes = 0, 0, len(pieces[j])
    offsets = np.zeros(3, dtype=np.uint64)
    res = (emu_bind.ScanResult * 1)()
    L.emu_join_intervals(C.byref(parent2), subs2, C.c_uint32(3), sr, offsets.ctypes.data_as(C.c_void_p), res, 1)
    joined = pieces[0] + b"\xff\xd0" + pieces[1] + b"\xff\xd1" + pieces[2]
    assert (res[0].errc, res[0].flags, res[0].bytes) == (0, 0, len(joined))
    assert out[:len(joined)].tobytes() == joined and not out[len(joined):].any()
    parent2.stream_capacity = len(joined) - 1               # the joined size decides destination_too_small
    out[:] = 0
    L.emu_join_intervals(C.byref(parent2), subs2, C.c_uint32(3), sr, offsets.ctypes.data_as(C.c_void_p), res, 1)
    assert (res[0].errc, res[0].bytes) == (3, 0) and not out.any()
    sr[1].errc = 3                                          # a private buffer was too small: repeat, do not report
    parent2.stream_capacity = 128
    L.emu_join_intervals(C.byref(parent2), subs2, C.c_uint32(3), sr, offsets.ctypes.data_as(C.c_void_p), res, 1)
    assert (res[0].errc, res[0].flags) == (0, 8)


@pytest.mark.parametrize("comps,ilv,ri,w,h", [(1, 0, 3, 40, 8), (1, 0, 1, 17, 4), (3, 2, 2, 12, 5), (3, 1, 4, 9, 6)])
def test_restart_interval_encode_end_to_end(comps, ilv, ri, w, h):
    """launch_encode_intervals on the CPU: the joined scan equals the oracle's coding of every interval with FF D0+m between
    them (for ILV_NONE / ILV_LINE / ILV_SAMPLE scans of the parallel pipeline)."""
    import oracle_bind as ob
    from charls_amd import synth
    L = emu_bind.lib()
    planes = [synth.frame_numpy(w, h, seed=90 + c, kind="mixed") for c in range(comps)]
    img = planes[0] if comps == 1 else np.ascontiguousarray(np.stack(planes, axis=-1))
    want, n = b"", (h + ri - 1) // ri
    for j in range(n):
        sub = np.ascontiguousarray(img[j * ri:(j + 1) * ri])
        s = ob.encode(sub, width=w, height=sub.shape[0], component_count=comps, interleave_mode=ilv)
        sc = jls_container.parse(s).scans[0]
        want += s[sc.data_start:sc.data_end] + (bytes([0xFF, 0xD0 + (j & 7)]) if j + 1 < n else b"")
    keep = []
    pix = np.frombuffer(img.tobytes(), dtype=np.uint8).copy()
    out = np.zeros(len(want) + 100, dtype=np.uint8)
    d = emu_bind.make_desc(w, h, comps, ilv, 8, 0, 0, (255, 3, 7, 21, 64), ri, pix, w * comps, out, keep)
    res = (emu_bind.ScanResult * 1)()
    L.emu_encode_with_restart_intervals(C.byref(d), res, 1, C.c_uint64(w * ri * comps * 4 + 256))
    assert (res[0].errc, res[0].flags, res[0].bytes) == (0, 0, len(want))
    assert out[:len(want)].tobytes() == want
