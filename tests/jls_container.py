"""Tiny .jls marker walker for the tests (locates the entropy-coded segments so kernels can be checked scan by scan)."""
from dataclasses import dataclass, field


@dataclass
class Scan:
    components: int
    near: int
    ilv: int
    data_start: int  # offset of the first entropy-coded byte
    data_end: int    # offset of the marker that terminates the segment


@dataclass
class Container:
    width: int = 0
    height: int = 0
    bits: int = 0
    components: int = 0
    transform: int = 0
    restart_interval: int = 0
    pc: tuple = (0, 0, 0, 0, 0)
    scans: list = field(default_factory=list)


def _next_marker(data: bytes, pos: int, restart_ok: bool) -> int:
    """Offset of the next 0xFF followed by a byte >= 0x80 that is not a restart marker (when restart_ok)."""
    n = len(data)
    while pos < n - 1:
        if data[pos] == 0xFF and data[pos + 1] >= 0x80:  # FF FF = fill bytes of a marker
            if not (restart_ok and 0xD0 <= data[pos + 1] <= 0xD7):
                return pos
        pos += 1
    return n


def parse(data: bytes) -> Container:
    c = Container()
    assert data[0:2] == b"\xff\xd8"
    pos = 2
    while pos < len(data):
        assert data[pos] == 0xFF, hex(pos)
        while pos + 1 < len(data) and data[pos + 1] == 0xFF:
            pos += 1
        if pos + 1 >= len(data):
            break
        m = data[pos + 1]
        if m == 0xD9:
            break
        size = int.from_bytes(data[pos + 2:pos + 4], "big")
        seg = data[pos + 4:pos + 2 + size]
        if m == 0xF7:
            c.bits = seg[0]
            c.height = c.height or int.from_bytes(seg[1:3], "big")
            c.width = c.width or int.from_bytes(seg[3:5], "big")
            c.components = seg[5]
        elif m == 0xF8 and seg[0] == 1:
            c.pc = tuple(int.from_bytes(seg[1 + 2 * i:3 + 2 * i], "big") for i in range(5))
        elif m == 0xF8 and seg[0] == 4:
            w = seg[1]
            c.height = int.from_bytes(seg[2:2 + w], "big")
            c.width = int.from_bytes(seg[2 + w:2 + 2 * w], "big")
        elif m == 0xDD:
            c.restart_interval = int.from_bytes(seg, "big")
        elif m == 0xE8 and len(seg) == 5 and seg[:4] == b"mrfx":
            c.transform = seg[4]
        pos += 2 + size
        if m == 0xDA:
            nc = seg[0]
            near, ilv = seg[1 + 2 * nc], seg[2 + 2 * nc]
            end = _next_marker(data, pos, c.restart_interval != 0)
            c.scans.append(Scan(nc, near, ilv, pos, end))
            pos = end
    return c


def default_pc(maxval: int, near: int):
    """ISO 14495-1 C.2.4.1.1.1 defaults (same table the reference pins in test/jpegls_preset_coding_parameters_test.cpp)."""
    def clamp(i, j):
        return j if (i > maxval or i < j) else i
    if maxval >= 128:
        f = (min(maxval, 4095) + 128) // 256
        t1 = clamp(f * 1 + 2 + 3 * near, near + 1)
        t2 = clamp(f * 4 + 3 + 5 * near, t1)
        t3 = clamp(f * 17 + 4 + 7 * near, t2)
    else:
        f = 256 // (maxval + 1)
        t1 = clamp(max(2, 3 // f + 3 * near), near + 1)
        t2 = clamp(max(3, 7 // f + 5 * near), t1)
        t3 = clamp(max(4, 21 // f + 7 * near), t2)
    return (maxval, t1, t2, t3, 64)


def validated_pc(pc, bits, near):
    bit_max = (1 << bits) - 1
    maxval = pc[0] or bit_max
    d = default_pc(maxval, near)
    return (maxval, pc[1] or d[1], pc[2] or d[2], pc[3] or d[3], pc[4] or d[4])
