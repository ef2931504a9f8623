"""Stage E of the encoder pipeline in its two forms (stuff_scan, and the block-parallel survey / resolve / emit of
block_stuffing.hip) on raw bit streams of every shape, against a bit-by-bit restatement of JPEG-LS stuffing
(reference src/scan_encoder.hpp:103-180: a byte that follows 0xFF carries seven bits; a final 0xFF is followed by 0x00)."""
import ctypes as C

import numpy as np
import pytest

import emu_bind


def _reference(raw: bytes, total_bits: int) -> bytes:
    padded = bytearray(raw) + b"\0\0\0"
    if total_bits % 8:  # zeros behind the last bit
        padded[total_bits // 8] &= (0xFF00 >> (total_bits % 8)) & 0xFF
        for k in range(total_bits // 8 + 1, len(padded)):
            padded[k] = 0
    out = bytearray()
    pos, short = 0, False
    while pos < total_bits:
        n = 7 if short else 8
        i, s = pos >> 3, pos & 7
        value = ((((padded[i] << 8) | padded[i + 1]) << s) >> (16 - n)) & ((1 << n) - 1)
        out.append(value)
        pos += n
        short = value == 0xFF
    if out and out[-1] == 0xFF:
        out.append(0)
    return bytes(out)


def _run(L, raw: bytes, total_bits: int, capacity: int, blocks: bool):
    raw_bytes = (len(raw) + 64 + 15) // 16 * 16
    buf = np.zeros(raw_bytes, dtype=np.uint8)
    buf[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    if total_bits % 8:  # the pipeline leaves zeros behind the last bit
        buf[total_bits // 8] &= (0xFF00 >> (total_bits % 8)) & 0xFF
        buf[total_bits // 8 + 1:] = 0
    out = np.full(capacity + 16, 0xEE, dtype=np.uint8)
    res = emu_bind.ScanResult()
    L.emu_stuff_raw(buf.ctypes.data_as(C.c_void_p), C.c_uint64(total_bits), C.c_uint64(raw_bytes), out.ctypes.data_as(C.c_void_p),
                    C.c_uint64(capacity), C.c_int(1 if blocks else 0), C.byref(res))
    assert (out[capacity:] == 0xEE).all(), "wrote behind the destination"
    return res.errc, res.flags, res.bytes, out[:min(capacity, res.bytes)].tobytes()


def _streams():
    rng = np.random.default_rng(5)
    cases = [(b"", 0), (b"\xff", 8), (b"\xff", 3), (b"\xff\xff\xff", 24), (b"\x00" * 40, 313), (b"\xff" * 3000, 24000 - 5)]
    for n, p_one in [(17, 0.5), (1023, 0.5), (1024, 0.97), (1025, 0.9), (5000, 0.99), (70000, 0.93), (300000, 0.6)]:
        bits = (rng.random(n * 8) < p_one).astype(np.uint8)
        cases.append((np.packbits(bits).tobytes(), n * 8 - int(rng.integers(0, 8))))
    # long stretches of ones with single zeros at every alignment: every entry state of a chunk occurs
    bits = np.ones(40000, dtype=np.uint8)
    bits[rng.integers(0, 40000, size=400)] = 0
    cases.append((np.packbits(bits).tobytes(), 40000 - 3))
    # more chunks than the resolve step has lanes: every lane composes the tables of several chunks into one map, and with
    # a 0xFF in every second or third byte the maps do not collapse to one exit state
    for n, p_one in [(600000, 0.9), (270001, 0.97)]:
        bits = (rng.random(n * 8) < p_one).astype(np.uint8)
        cases.append((np.packbits(bits).tobytes(), n * 8 - int(rng.integers(0, 8))))
    return cases


@pytest.mark.parametrize("index", range(len(_streams())))
def test_both_forms_equal_the_bitwise_rule(index):
    L = emu_bind.lib()
    raw, total_bits = _streams()[index]
    want = _reference(raw, total_bits)
    for blocks in (False, True):
        errc, flags, size, data = _run(L, raw, total_bits, len(want) + 100, blocks)
        assert (errc, size) == (0, len(want)), (blocks, errc, size, len(want))
        assert data == want, blocks
        assert flags == 0


@pytest.mark.parametrize("index", [3, 8, 9, 12])
def test_capacity_verdicts_agree(index):
    """Too small, exact, and the three-byte zone in which the host re-runs the exact kernel (flags bit 1)."""
    L = emu_bind.lib()
    raw, total_bits = _streams()[index]
    want = _reference(raw, total_bits)
    for capacity in [0, 1, len(want) - 1, len(want), len(want) + 1, len(want) + 3, len(want) + 4]:
        if capacity < 0:
            continue
        a = _run(L, raw, total_bits, capacity, False)
        b = _run(L, raw, total_bits, capacity, True)
        assert a[:3] == b[:3], (capacity, a[:3], b[:3])
        assert a[3] == b[3] == want[:min(capacity, len(want))], capacity


def test_pipeline_through_the_block_form(monkeypatch):
    """The whole encoder pipeline with stage E in its block-parallel form (the switch runtime.hip reads as well)."""
    import test_emu_pipeline as P
    monkeypatch.setenv("CHARLS_AMD_BLOCK_STUFFING", "1")
    P.test_pipeline_batch_of_seeded_frames("mixed", 8, 130, 11, 1)
    P.test_pipeline_batch_of_seeded_frames("noise", 16, 40, 12, 9)
    P.test_pipeline_destination_too_small_and_knife_edge()
