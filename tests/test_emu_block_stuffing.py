"""Stage E of the encoder pipeline in its three forms (stuff_scan; the block-parallel survey / resolve / emit of
block_stuffing.hip; the speculative form of speculative_stuffing.hip, with chunks and warm-ups small enough that guesses fail
and scans are given up) on raw bit streams of every shape, against a bit-by-bit restatement of JPEG-LS stuffing
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


def _run(L, raw: bytes, total_bits: int, capacity: int, blocks):
    """blocks: False / 0 = stuff_scan, True / 1 = block form, 2 = speculative form."""
    raw_bytes = (len(raw) + 64 + 15) // 16 * 16
    buf = np.zeros(raw_bytes, dtype=np.uint8)
    buf[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    if total_bits % 8:  # the pipeline leaves zeros behind the last bit
        buf[total_bits // 8] &= (0xFF00 >> (total_bits % 8)) & 0xFF
        buf[total_bits // 8 + 1:] = 0
    out = np.full(capacity + 16, 0xEE, dtype=np.uint8)
    res = emu_bind.ScanResult()
    L.emu_stuff_raw(buf.ctypes.data_as(C.c_void_p), C.c_uint64(total_bits), C.c_uint64(raw_bytes), out.ctypes.data_as(C.c_void_p),
                    C.c_uint64(capacity), C.c_int(int(blocks)), C.byref(res))
    assert (out[capacity:] == 0xEE).all(), "wrote behind the destination"
    return res.errc, res.flags, res.bytes, out[:min(capacity, res.bytes)].tobytes()


def _streams():
    rng = np.random.default_rng(5)
    cases = [(b"", 0), (b"\xff", 8), (b"\xff", 3), (b"\xff\xff\xff", 24), (b"\x00" * 40, 313), (b"\xff" * 3000, 24000 - 5)]
    for n, p_one in [(17, 0.5), (1023, 0.5), (1024, 0.97), (1025, 0.9), (5000, 0.99), (40000, 0.93), (90000, 0.6)]:
        bits = (rng.random(n * 8) < p_one).astype(np.uint8)
        cases.append((np.packbits(bits).tobytes(), n * 8 - int(rng.integers(0, 8))))
    # long stretches of ones with single zeros at every alignment: every entry state of a chunk occurs
    bits = np.ones(40000, dtype=np.uint8)
    bits[rng.integers(0, 40000, size=400)] = 0
    cases.append((np.packbits(bits).tobytes(), 40000 - 3))
    # more chunks than the resolve step has lanes: every lane composes the tables of several chunks into one map, and with
    # a 0xFF in every second or third byte the maps do not collapse to one exit state
    for n, p_one in [(160000, 0.9), (150001, 0.97)]:
        bits = (rng.random(n * 8) < p_one).astype(np.uint8)
        cases.append((np.packbits(bits).tobytes(), n * 8 - int(rng.integers(0, 8))))
    return cases


SPEC = [None, ("64", "0"), ("64", "64"), ("256", "2048"), ("1024", "512")]  # (chunk, warm-up) of the speculative form


def _spec(monkeypatch, geometry):
    if geometry is None:
        monkeypatch.delenv("CHARLS_AMD_SPEC_CHUNK", raising=False)
        monkeypatch.delenv("CHARLS_AMD_SPEC_WARM", raising=False)
    else:
        monkeypatch.setenv("CHARLS_AMD_SPEC_CHUNK", geometry[0])
        monkeypatch.setenv("CHARLS_AMD_SPEC_WARM", geometry[1])


@pytest.mark.parametrize("index", range(len(_streams())))
def test_all_forms_equal_the_bitwise_rule(index, monkeypatch):
    L = emu_bind.lib()
    raw, total_bits = _streams()[index]
    want = _reference(raw, total_bits)
    # (the harness runs a wavefront as 64 threads: the 64-byte chunks only on the shorter streams)
    spec = SPEC if len(raw) <= 80000 else [None]
    for blocks, geometry in [(0, None), (1, None)] + [(2, g) for g in spec]:
        _spec(monkeypatch, geometry)
        errc, flags, size, data = _run(L, raw, total_bits, len(want) + 100, blocks)
        assert (errc, size) == (0, len(want)), (blocks, geometry, errc, size, len(want))
        assert data == want, (blocks, geometry)
        assert flags == 0


@pytest.mark.parametrize("index", [3, 8, 9, 12])
def test_capacity_verdicts_agree(index, monkeypatch):
    """Too small, exact, and the three-byte zone in which the host re-runs the exact kernel (flags bit 1)."""
    L = emu_bind.lib()
    raw, total_bits = _streams()[index]
    want = _reference(raw, total_bits)
    for capacity in [0, 1, len(want) - 1, len(want), len(want) + 1, len(want) + 3, len(want) + 4]:
        if capacity < 0:
            continue
        a = _run(L, raw, total_bits, capacity, 0)
        for blocks, geometry in [(1, None), (2, None), (2, ("256", "2048"))] + ([(2, ("64", "64"))] if len(raw) <= 2000 else []):
            _spec(monkeypatch, geometry)
            b = _run(L, raw, total_bits, capacity, blocks)
            assert a[:3] == b[:3], (capacity, blocks, geometry, a[:3], b[:3])
            assert a[3] == b[3] == want[:min(capacity, len(want))], (capacity, blocks, geometry)


def test_tile_pipeline_through_every_form(monkeypatch):
    """The whole encoder pipeline with stage E in each of its forms (CHARLS_AMD_BLOCK_STUFFING: 0, 1, and -- a switch of the
    harness -- 2 for the speculative form, here with 64-byte chunks)."""
    import test_emu_tile_pipeline as P
    monkeypatch.setenv("CHARLS_AMD_SPEC_CHUNK", "64")
    monkeypatch.setenv("CHARLS_AMD_SPEC_WARM", "256")
    for form in ("0", "1", "2"):
        monkeypatch.setenv("CHARLS_AMD_BLOCK_STUFFING", form)
        P.test_tile_pipeline_batch_of_seeded_frames("mixed", 8, 130, 11, 1, 64, 32)
        P.test_tile_pipeline_batch_of_seeded_frames("noise", 16, 40, 12, 9, 64, 32)
    monkeypatch.setenv("CHARLS_AMD_BLOCK_STUFFING", "2")
    img = P.synth.frame_numpy(64, 64, seed=3, kind="mixed")
    want = P.ob.encode(img, width=64, height=64)
    n = len(P._scan_bytes(want))
    pc = P.jls_container.validated_pc((0,) * 5, 8, 0)
    (errc, flags, data), = P._encode_planes([img], 64, 64, 8, pc, n - 1)
    assert errc == 3
    (errc, flags, data), = P._encode_planes([img], 64, 64, 8, pc, n + 2)
    assert errc == 0 and flags == 2
    (errc, flags, data), = P._encode_planes([img], 64, 64, 8, pc, n + 4)
    assert errc == 0 and flags == 0 and data == P._scan_bytes(want)
