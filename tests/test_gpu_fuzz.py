"""Decoder fuzzing against the oracle (reference fuzzing/libfuzzer-decoder/main.cpp:10-30): mutated
streams must give the oracle's pixels or the oracle's error code, through the C ABI, for every decoder the dispatcher
can pick (fast -> exact wave -> sequential, interval-parallel for DRI streams).  GPU only."""
import numpy as np
import pytest

import common
import jls_container
import oracle_bind as ob
from charls_amd import capi, synth
from charls_amd.capi import JpegLSError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    L = capi.load_product()
    assert L.lib.charls_amd_device_status() == 0
    return L


def _bases(lib):
    g8 = synth.frame_numpy(96, 40, seed=11, kind="mixed")
    g16 = synth.frame_numpy(70, 24, seed=12, bits=16, kind="mixed")
    rgb = np.stack([synth.frame_numpy(48, 20, seed=13 + c, kind="mixed") for c in range(3)], axis=-1)
    flat = np.full((30, 64), 77, dtype=np.uint8)
    flat[10:20, 5:50] = synth.frame_numpy(45, 10, seed=3, kind="mixed")
    return {
        "gray8": ob.encode(g8, width=96, height=40),
        "gray16": ob.encode(g16, width=70, height=24, bits_per_sample=16),
        "gray8_near3": ob.encode(g8, width=96, height=40, near_lossless=3),
        "rgb_sample": ob.encode(rgb, width=48, height=20, component_count=3, interleave_mode=2),
        "rgb_line_hp1": ob.encode(rgb, width=48, height=20, component_count=3, interleave_mode=1, color_transformation=1),
        "runs": ob.encode(flat, width=64, height=30),
        "gray8_dri": lib.encode(g8, restart_interval=8),
        "rgb_sample_near2_dri": lib.encode(rgb, component_count=3, interleave_mode=2, near_lossless=2, restart_interval=4),
        "gray16_dri": lib.encode(g16, bits_per_sample=16, restart_interval=5),
        "ref_rm_7": common.refdata("test8_ilv_none_rm_7.jls"),
    }


def _mutate(rng, data: bytes, scan_start: int) -> bytes:
    b = bytearray(data)
    kind = rng.integers(0, 6)
    lo = scan_start  # the entropy-coded part and what follows; header error codes are pinned by tests/test_host_facade.py
    # against the reference itself (the oracle's container parser is minimal and reports fewer distinct codes)
    if kind == 0:      # flip a few bits
        for _ in range(int(rng.integers(1, 4))):
            i = int(rng.integers(lo, len(b)))
            b[i] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:    # overwrite a byte
        b[int(rng.integers(lo, len(b)))] = int(rng.integers(0, 256))
    elif kind == 2:    # truncate
        del b[int(rng.integers(lo, len(b))):]
    elif kind == 3:    # delete a short run
        i = int(rng.integers(lo, len(b) - 1))
        del b[i:i + int(rng.integers(1, 5))]
    elif kind == 4:    # insert bytes (0xFF likes to make markers)
        i = int(rng.integers(lo, len(b)))
        b[i:i] = bytes(rng.choice([0x00, 0xFF, 0x7F, 0xD0, 0x80], size=int(rng.integers(1, 4))).astype(np.uint8))
    else:              # swap two bytes
        i, j = int(rng.integers(lo, len(b))), int(rng.integers(lo, len(b)))
        b[i], b[j] = b[j], b[i]
    return bytes(b)


def _outcome(fn, data):
    try:
        out = fn(data)
        return ("ok", np.asarray(out[1]).tobytes())
    except (JpegLSError, ob.OracleError) as e:
        return ("err", e.errc)


@pytest.mark.parametrize("name", ["gray8", "gray16", "gray8_near3", "rgb_sample", "rgb_line_hp1", "runs", "gray8_dri", "rgb_sample_near2_dri",
                                  "gray16_dri", "ref_rm_7"])
def test_mutated_streams_decode_like_the_oracle(lib, name):
    base = _bases(lib)[name]
    cont = jls_container.parse(base)
    start = cont.scans[0].data_start
    rng = np.random.default_rng(abs(hash(name)) % (1 << 32) if False else sum(name.encode()))
    assert _outcome(lib.decode, base) == _outcome(ob.decode, base)
    mismatches = []
    for k in range(60):
        data = _mutate(rng, base, start)
        want = _outcome(ob.decode, data)
        got = _outcome(lib.decode, data)
        if got != want:
            mismatches.append((k, want[0], want[1] if want[0] == "err" else len(want[1]), got[0],
                               got[1] if got[0] == "err" else len(got[1])))
    assert not mismatches, mismatches


@pytest.mark.parametrize("restart", [0, 6])
def test_mutated_planar_streams_through_the_batch_decoder(lib, knobs, restart):
    """The batch decoder finds the component scans of planar frames by searching for the marker that ends each of them and
    decodes them by one launch; a frame for which anything is out of the ordinary is decoded again scan by scan.  Eighty
    mutated planar RGB streams in ONE batch: every frame's error code equals what part 1 (one stream through the C ABI)
    and the scan-by-scan batch decoder report for it, and its pixels are the oracle's wherever the oracle decodes it."""
    import torch
    from charls_amd import batch
    w, h, n = 48, 20, 80
    rgb = np.stack([synth.frame_numpy(w, h, seed=23 + c, kind="mixed") for c in range(3)])  # planar
    base = lib.encode(rgb, component_count=3, restart_interval=restart) if restart else ob.encode(rgb, width=w, height=h, component_count=3)
    base = bytes(base)
    start = jls_container.parse(base).scans[0].data_start
    rng = np.random.default_rng(77 + restart)
    streams = [base] + [_mutate(rng, base, start) for _ in range(n - 1)]
    pitch = max(len(s) for s in streams) + 16
    host = np.zeros((n, pitch), dtype=np.uint8)
    for f, s in enumerate(streams):
        host[f, :len(s)] = np.frombuffer(s, dtype=np.uint8)
    sizes = np.array([len(s) for s in streams], dtype=np.uint64)
    dev = torch.from_numpy(host).cuda()
    out_a = torch.zeros((n, 3, h, w), dtype=torch.uint8, device="cuda")
    out_b = torch.zeros_like(out_a)
    knobs.clear("BATCH_ROUNDS")
    _, errcs_a, _ = batch.decode_batch(dev, sizes, out_a)
    knobs.set("BATCH_ROUNDS", 1)
    _, errcs_b, _ = batch.decode_batch(dev, sizes, out_b)
    knobs.clear("BATCH_ROUNDS")
    assert list(errcs_a) == list(errcs_b)
    assert errcs_a[0] == 0 and (errcs_a != 0).sum() > 10 and (errcs_a == 0).sum() > 3  # (the mutations do both)
    mismatches = []
    for f, s in enumerate(streams):
        part1 = _outcome(lib.decode, s)
        want = _outcome(ob.decode, s)
        if (part1[1] if part1[0] == "err" else 0) != int(errcs_a[f]):
            mismatches.append((f, "errc", part1, int(errcs_a[f])))
        elif want[0] == "ok" and (errcs_a[f] != 0 or out_a[f].cpu().numpy().tobytes() != want[1] or not torch.equal(out_a[f], out_b[f])):
            mismatches.append((f, "pixels", int(errcs_a[f])))
    assert not mismatches, mismatches
