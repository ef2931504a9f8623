"""Batch API on device-resident frames (charls_amd.h part 2): every frame bit-exact, round-trip properties at scale."""
import os

import numpy as np
import pytest

import common
import oracle_bind as ob
from charls_amd import batch, capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("bits,kind", [(8, "mixed"), (8, "gradient"), (16, "mixed"), (12, "gradient")])
def test_batch_frames_equal_oracle(torch, bits, kind):
    n, w, h = 5, 200, 120
    frames = synth.frames_torch(n, w, h, seed0=40, bits=bits, kind=kind, device="cuda:0")
    enc = batch.encode_batch(frames, bits_per_sample=bits)
    host = enc.streams.cpu().numpy()
    for f in range(n):
        want = ob.encode(synth.frame_numpy(w, h, seed=40 + f, bits=bits, kind=kind), width=w, height=h, bits_per_sample=bits)
        assert enc.errcs[f] == 0
        assert host[f, :int(enc.sizes[f])].tobytes() == want
    out = torch.empty_like(frames)
    p, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert (errcs == 0).all() and torch.equal(out, frames)
    assert (p.frame_info.width, p.frame_info.height, p.frame_info.bits_per_sample) == (w, h, bits)


def test_batch_rgb_modes(torch):
    n, w, h = 3, 96, 64
    for ilv, near, ct in [(0, 0, 0), (1, 0, 0), (2, 0, 1), (2, 2, 0), (1, 0, 3)]:
        imgs = [synth.frame_numpy(w, h, seed=60 + f, components=3, kind="mixed", interleaved=(ilv != 0)) for f in range(n)]
        frames = torch.from_numpy(np.stack(imgs)).cuda()
        enc = batch.encode_batch(frames, component_count=3, interleave_mode=ilv, near_lossless=near, color_transformation=ct)
        host = enc.streams.cpu().numpy()
        for f in range(n):
            want = ob.encode(imgs[f], width=w, height=h, component_count=3, interleave_mode=ilv, near_lossless=near,
                             color_transformation=ct)
            assert enc.errcs[f] == 0 and host[f, :int(enc.sizes[f])].tobytes() == want, (ilv, near, ct, f)
        out = torch.empty_like(frames)
        _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
        assert (errcs == 0).all()
        if near == 0:
            assert torch.equal(out, frames)
        else:
            assert (out.int() - frames.int()).abs().max().item() <= near


@pytest.mark.parametrize("comps,ilv,near,ct,bits", [(1, 0, 2, 0, 8), (3, 1, 2, 0, 8), (3, 1, 0, 1, 8), (3, 2, 3, 0, 8), (1, 0, 3, 0, 12),
                                                   (3, 1, 2, 0, 12), (3, 2, 0, 2, 16), (4, 2, 1, 0, 8)])
def test_batch_midsize_modes_equal_oracle(torch, comps, ilv, near, ct, bits):
    """The lanes-per-scan kernels of the modes without a pipeline form (near-lossless, interleaved) on frames wide and tall
    enough for many refills of the bit ring and drains of the staging ring: the coded bytes are the oracle's, and what they
    decode to is what the oracle decodes (near-lossless: the reconstruction, bit for bit)."""
    n, w, h = 3, 1000, 300
    imgs = [synth.frame_numpy(w, h, seed=90 + f, bits=bits, components=comps, kind="mixed", interleaved=(ilv != 0)) for f in range(n)]
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    if comps > 1 and ilv == 0:
        pytest.skip("planar multi-component frames have their own test")
    enc = batch.encode_batch(frames, bits_per_sample=bits, component_count=comps, interleave_mode=ilv, near_lossless=near,
                             color_transformation=ct)
    host = enc.streams.cpu().numpy()
    out = torch.empty_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert (errcs == 0).all()
    got = out.cpu().numpy()
    for f in range(n):
        want = ob.encode(imgs[f], width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=ilv,
                         near_lossless=near, color_transformation=ct)
        assert enc.errcs[f] == 0 and host[f, :int(enc.sizes[f])].tobytes() == want, f
        assert got[f].tobytes() == ob.decode(want)[1].tobytes(), f


def test_batch_destination_too_small_is_per_frame(torch):
    w, h = 64, 64
    frames = torch.stack([synth.frames_torch(1, w, h, seed0=1, kind="noise", device="cuda:0")[0],
                          synth.frames_torch(1, w, h, seed0=2, kind="zero", device="cuda:0")[0]])
    streams = torch.zeros((2, 2048), dtype=torch.uint8, device="cuda:0")
    enc = batch.encode_batch(frames, streams=streams)
    assert enc.errcs[0] == 3 and enc.errcs[1] == 0  # noise does not fit 2 KiB, zeros do
    assert enc.sizes[0] == 0 and enc.sizes[1] > 0


def test_batch_corrupt_stream_is_per_frame(torch):
    w, h = 64, 48
    frames = synth.frames_torch(3, w, h, seed0=9, kind="mixed", device="cuda:0")
    enc = batch.encode_batch(frames)
    bad = enc.streams.clone()
    bad[1, 40:60] = 0xFF  # destroy part of the entropy-coded segment of frame 1
    out = torch.zeros_like(frames)
    _, errcs, _ = batch.decode_batch(bad, enc.sizes, out)
    assert errcs[0] == 0 and errcs[2] == 0 and errcs[1] != 0
    assert torch.equal(out[0], frames[0]) and torch.equal(out[2], frames[2])


@pytest.mark.slow
@pytest.mark.parametrize("bits", [8, 16])
def test_full_size_roundtrip_properties(torch, bits):
    """BASELINE.json sizes: lossless round trip restores every sample; sizes equal the golden manifest."""
    cases = {c["name"]: c for c in common.cases()}
    c = cases["cfg2_full" if bits == 8 else "cfg3_full"]
    frames = synth.frames_torch(2, 4096, 4096, seed0=c["seed"], bits=bits, device="cuda:0")
    enc = batch.encode_batch(frames, bits_per_sample=bits)
    assert (enc.errcs == 0).all()
    assert int(enc.sizes[0]) == c["jls_size"]
    assert common.sha(enc.streams[0, :int(enc.sizes[0])].cpu().numpy().tobytes()) == c["jls_sha256"]
    out = torch.empty_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert (errcs == 0).all() and torch.equal(out, frames)


@pytest.mark.slow
def test_config4_frames_equal_reference_hashes(torch):
    """BASELINE configs[3]: independent 2048x2048 8-bit frames with per-frame seeds (100 + f); the first four are pinned
    by hashes of the reference's output, all must round-trip."""
    cases = {c["name"]: c for c in common.cases()}
    n = 8
    frames = synth.frames_torch(n, 2048, 2048, seed0=100, device="cuda:0")
    enc = batch.encode_batch(frames)
    assert (enc.errcs == 0).all()
    host = enc.streams.cpu().numpy()
    for f in range(4):
        c = cases[f"cfg4_frame{f}"]
        data = host[f, :int(enc.sizes[f])].tobytes()
        assert (len(data), common.sha(data)) == (c["jls_size"], c["jls_sha256"])
    assert len(set(int(s) for s in enc.sizes)) > 1  # the frames really differ in length (variable-size gather)
    out = torch.empty_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert (errcs == 0).all() and torch.equal(out, frames)


def test_exact_decoder_agrees_with_speed_path(torch, knobs):
    """CHARLS_AMD_EXACT_DECODER=1 forces the reference-policy decoder; both must give the same pixels and byte counts."""
    frames = synth.frames_torch(4, 333, 77, seed0=5, kind="mixed", device="cuda:0")
    enc = batch.encode_batch(frames)
    out_fast = torch.empty_like(frames)
    batch.decode_batch(enc.streams, enc.sizes, out_fast)
    knobs.set("EXACT_DECODER", 1)
    out_exact = torch.empty_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out_exact)
    assert (errcs == 0).all() and torch.equal(out_fast, out_exact) and torch.equal(out_exact, frames)


@pytest.mark.parametrize("group,waves,bits,count,w,h", [(16, 4, 8, 37, 300, 40), (32, 4, 8, 5, 4096, 6), (32, 8, 16, 19, 129, 33), (16, 8, 12, 70, 64, 20),
                                                        (16, 4, 8, 1, 50, 50), (32, 4, 16, 64, 640, 24)])
def test_decoder_workgroups_of_several_wavefronts(torch, knobs, group, waves, bits, count, w, h):
    """decode_scans_group<S, G, 1, W>: the launch shape of big batches (one workgroup per CU, a wavefront per SIMD: runtime.hip,
    decode_group_plan), forced here on small ones -- counts that leave wavefronts of the last workgroup without a scan, a single
    scan, a 4096-sample line -- and compared with the one-wavefront workgroups and the source frames."""
    frames = synth.frames_torch(count, w, h, seed0=90, bits=bits, kind="mixed", device="cuda:0")
    enc = batch.encode_batch(frames, bits_per_sample=bits)
    assert (enc.errcs == 0).all()
    knobs.set("DECODE_GROUP", group)
    knobs.set("DECODE_WORKGROUP_WAVES", waves)
    out = torch.zeros_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert (errcs == 0).all() and torch.equal(out, frames)
    knobs.set("DECODE_WORKGROUP_WAVES", 1)
    again = torch.zeros_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, again)
    assert (errcs == 0).all() and torch.equal(again, frames)


def test_the_launch_rule_takes_workgroups_for_big_batches(torch):
    """More scans than two one-wavefront workgroups per CU can take at 32 lanes per scan (1024 on an MI355X): the library's own
    rule switches to workgroups of four wavefronts.  3000 small frames, every one compared."""
    frames = synth.frames_torch(24, 96, 16, seed0=7, kind="mixed", device="cuda:0").repeat(125, 1, 1).contiguous()
    enc = batch.encode_batch(frames)
    out = torch.zeros_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert (errcs == 0).all() and torch.equal(out, frames)


def test_serial_and_pipeline_encoders_agree(torch):
    frames = synth.frames_torch(3, 257, 65, seed0=8, kind="mixed", device="cuda:0")
    batch.set_encode_engine(1)
    a = batch.encode_batch(frames)
    batch.set_encode_engine(2)
    b = batch.encode_batch(frames)
    batch.set_encode_engine(0)
    assert (a.sizes == b.sizes).all()
    for f in range(3):
        n = int(a.sizes[f])
        assert torch.equal(a.streams[f, :n], b.streams[f, :n])


@pytest.mark.slow
def test_config4_all_256_frames_and_their_exchange(torch):
    """BASELINE configs[3] as stated: 256 independent 2048x2048 8-bit frames (seeds 100 + f).  The first four are pinned by
    hashes of the reference's output, all of them round-trip, their lengths differ, and the whole payload goes through the
    bitstream exchange of the multi-GPU path (RCCL with one rank here: process group, size all-gather, own-rank hand-over)."""
    import torch.distributed as dist
    cases = {c["name"]: c for c in common.cases()}
    n = 256
    frames = synth.frames_torch(n, 2048, 2048, seed0=100, device="cuda:0")
    enc = batch.encode_batch(frames)
    assert (enc.errcs == 0).all()
    for f in range(4):
        c = cases[f"cfg4_frame{f}"]
        data = enc.streams[f, :int(enc.sizes[f])].cpu().numpy().tobytes()
        assert (len(data), common.sha(data)) == (c["jls_size"], c["jls_sha256"])
    assert len(set(int(s) for s in enc.sizes)) > 32  # variable lengths: what the exchange has to cope with
    out = torch.empty_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert (errcs == 0).all() and torch.equal(out, frames)
    # the exchange, with the real payload
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        got = {}
        _, sizes = batch.gather_streams(enc.streams, enc.sizes, dst=0, chunk_frames=32,
                                        sink=lambda r, first, part, sz: got.setdefault(first, (part, sz)))
        assert sorted(got) == list(range(0, n, 32)) and (sizes[0] == enc.sizes).all()
        for first, (part, sz) in got.items():
            for k in (0, len(sz) - 1):
                f = first + k
                assert torch.equal(part[k, :int(sz[k])], enc.streams[f, :int(enc.sizes[f])])
    finally:
        dist.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("name", ["cfg5a_full", "cfg5b_full"])
def test_config5_full_size_equals_reference_hash(torch, name):
    """BASELINE configs[4] at its real size, 4096x4096 RGB sample-interleaved: 5a = HP1 lossless, 5b = NEAR 2 (the literal
    combination HP1 + NEAR 2 is rejected by the reference, see test_gpu_parity).  The stream must hash to the reference's
    (tests/golden/cases.json), the decoder must restore the pixels (5a) or stay within NEAR and equal the oracle's
    reconstruction checksum (5b)."""
    cases = {c["name"]: c for c in common.cases()}
    c = cases[name]
    planes = [synth.frames_torch(1, 4096, 4096, seed0=c["seed"] + 7919 * k, device="cuda:0")[0] for k in range(3)]
    rgb = torch.stack(planes, dim=2).unsqueeze(0).contiguous()  # (1, H, W, 3): synth.frame_numpy's interleaved layout
    assert common.sha(rgb.cpu().numpy().tobytes()) == c["input_sha256"]
    enc = batch.encode_batch(rgb, component_count=3, interleave_mode=2, near_lossless=c["near_lossless"],
                             color_transformation=c["color_transformation"])
    assert enc.errcs[0] == 0 and int(enc.sizes[0]) == c["jls_size"]
    assert common.sha(enc.streams[0, :int(enc.sizes[0])].cpu().numpy().tobytes()) == c["jls_sha256"]
    out = torch.empty_like(rgb)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert errcs[0] == 0
    if c["near_lossless"] == 0:
        assert torch.equal(out, rgb)
    else:
        assert (out.int() - rgb.int()).abs().max().item() <= c["near_lossless"]
    assert common.sha(out.cpu().numpy().tobytes()) == c["decoded_sha256"]


def test_planar_batch_codes_all_component_scans_in_one_launch(torch):
    """the component scans of planar frames are coded by ONE launch (frames x components scans) and
    put in place afterwards.  A 3-plane 2048 x 2048 batch, bytes against the oracle; the time of the batch against the time
    of the same planes as single-component frames."""
    import time
    n, w, h = 4, 2048, 2048
    imgs = [synth.frame_numpy(w, h, seed=300 + f, components=3, kind="mixed", interleaved=False) for f in range(n)]
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    batch.encode_batch(frames, component_count=3)  # warm-up (work areas)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc = batch.encode_batch(frames, component_count=3)
    t_planar = time.perf_counter() - t0
    host = enc.streams.cpu().numpy()
    for f in range(n):
        want = ob.encode(imgs[f], width=w, height=h, component_count=3)
        assert enc.errcs[f] == 0 and host[f, :int(enc.sizes[f])].tobytes() == want, f
    planes = frames.reshape(n * 3, h, w)
    batch.encode_batch(planes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch.encode_batch(planes)
    t_planes = time.perf_counter() - t0
    assert t_planar < 1.5 * t_planes + 0.002, (t_planar, t_planes)  # (in rounds it took three times the launches of a third of the scans)
    out = torch.empty_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert (errcs == 0).all() and torch.equal(out, frames)


@pytest.mark.parametrize("bits,near,comps", [(8, 0, 3), (8, 2, 3), (16, 0, 2), (12, 0, 4)])
def test_planar_batch_every_capacity_around_the_fit(torch, bits, near, comps):
    """Slots that are too small for some component, exactly large enough, or within the 4 bytes in which the reference's
    verdict depends on the capacity it passes to the scan: every frame gets the oracle's errc / bytes (frames whose scans
    cannot simply be put in place are coded again scan by scan)."""
    n, w, h = 3, 80, 40
    imgs = [synth.frame_numpy(w, h, seed=400 + f, bits=bits, components=comps, kind="mixed", interleaved=False) for f in range(n)]
    frames = torch.from_numpy(np.stack(imgs).astype(np.int16 if bits > 8 else np.uint8)).cuda()
    kw = dict(width=w, height=h, bits_per_sample=bits, component_count=comps, near_lossless=near)
    full = [ob.encode(img, **kw) for img in imgs]
    for pitch in sorted({len(full[0]) + d for d in (-40, -5, -3, -2, -1, 0, 1, 2, 3, 4, 5, 64)} | {len(full[1]), len(full[2]) + 1}):
        streams = torch.zeros((n, pitch), dtype=torch.uint8, device="cuda")
        enc = batch.encode_batch(frames, bits_per_sample=bits, component_count=comps, near_lossless=near, streams=streams)
        host = enc.streams.cpu().numpy()
        for f in range(n):
            try:
                want, ew = ob.encode(imgs[f], destination_size=pitch, **kw), 0
            except ob.OracleError as e:
                want, ew = None, e.errc
            assert enc.errcs[f] == ew, (pitch, f, enc.errcs[f], ew)
            if want is not None:
                assert host[f, :int(enc.sizes[f])].tobytes() == want, (pitch, f)


def _decode_both_ways(torch, streams, sizes, like, knobs):
    """The batch decoder with the component scans of planar frames in one launch, and scan by scan (CHARLS_AMD_BATCH_ROUNDS)."""
    out_a, out_b = torch.zeros_like(like), torch.zeros_like(like)
    knobs.clear("BATCH_ROUNDS")
    _, errcs_a, _ = batch.decode_batch(streams, sizes, out_a)
    knobs.set("BATCH_ROUNDS", 1)
    _, errcs_b, _ = batch.decode_batch(streams, sizes, out_b)
    knobs.clear("BATCH_ROUNDS")
    return out_a, errcs_a, out_b, errcs_b


def test_planar_batch_decodes_all_component_scans_in_one_launch(torch, knobs):
    """VERDICT round 3, item 8, the decoder's half: the scans of planar frames are found (the marker that ends a scan is
    searched for on the device, the part-1 reader parses on from there) and decoded by ONE launch.  Pixels against the
    frames, the same pixels and error codes as scan by scan, and the time against the same planes as separate frames."""
    import time
    n, w, h = 6, 1024, 1024
    imgs = [synth.frame_numpy(w, h, seed=500 + f, components=3, kind="mixed", interleaved=False) for f in range(n)]
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    enc = batch.encode_batch(frames, component_count=3)
    assert (enc.errcs == 0).all()
    out_a, errcs_a, out_b, errcs_b = _decode_both_ways(torch, enc.streams, enc.sizes, frames, knobs)
    assert (errcs_a == 0).all() and (errcs_b == 0).all()
    assert torch.equal(out_a, frames) and torch.equal(out_b, frames)
    planes = batch.encode_batch(frames.reshape(n * 3, h, w))
    out_p = torch.zeros((n * 3, h, w), dtype=frames.dtype, device="cuda")
    batch.decode_batch(planes.streams, planes.sizes, out_p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch.decode_batch(planes.streams, planes.sizes, out_p)
    t_planes = time.perf_counter() - t0
    t0 = time.perf_counter()
    batch.decode_batch(enc.streams, enc.sizes, out_a)
    t_planar = time.perf_counter() - t0
    assert t_planar < 1.5 * t_planes + 0.005, (t_planar, t_planes)  # (scan by scan: three rounds of a third of the scans)


@pytest.mark.parametrize("bits,comps,near,restart", [(8, 3, 0, 0), (8, 3, 2, 0), (16, 2, 0, 0), (12, 4, 0, 0), (8, 3, 0, 7)])
def test_planar_batch_decode_modes_and_damage(torch, knobs, bits, comps, near, restart):
    """Planar frames of several widths of sample, near-lossless, with restart intervals; then the same batch with one stream
    cut short, one damaged inside its second scan, one whose end-of-image marker is gone and one with a comment segment THAT
    CONTAINS A START-OF-SCAN MARKER between two scans: pixels and error codes equal what the scan-by-scan decoder gives."""
    n, w, h = 5, 96, 56
    imgs = [synth.frame_numpy(w, h, seed=520 + f, bits=bits, components=comps, kind="mixed", interleaved=False) for f in range(n)]
    frames = torch.from_numpy(np.stack(imgs).astype(np.int16 if bits > 8 else np.uint8)).cuda()
    enc = batch.encode_batch(frames, bits_per_sample=bits, component_count=comps, near_lossless=near, restart_interval=restart)
    assert (enc.errcs == 0).all()
    out_a, errcs_a, out_b, errcs_b = _decode_both_ways(torch, enc.streams, enc.sizes, frames, knobs)
    assert (errcs_a == 0).all() and (errcs_b == 0).all() and torch.equal(out_a, out_b)
    if near == 0:
        assert torch.equal(out_a, frames)
    else:
        assert int((out_a.to(torch.int32) - frames.to(torch.int32)).abs().max()) <= near
    host = enc.streams.cpu().numpy().copy()
    sizes = np.array(enc.sizes, dtype=np.uint64)
    pitch = host.shape[1] + 64
    bad = np.zeros((n, pitch), dtype=np.uint8)
    bad[:, :host.shape[1]] = host

    def scan_starts(f):  # offsets of the SOS markers of frame f
        b = bytes(host[f, :int(sizes[f])])
        at, found = 0, []
        while True:
            at = b.find(b"\xff\xda", at)
            if at < 0:
                return found
            found.append(at)
            at += 2

    sizes[0] = sizes[0] * 2 // 3                      # cut short
    second = scan_starts(1)[1]
    bad[1, second + 30:second + 40] = 0xFF            # damage inside the second scan
    sizes[2] -= 2                                     # no end-of-image marker
    s3 = scan_starts(3)[1]
    comment = bytes([0xFF, 0xFE, 0x00, 0x08, 0xFF, 0xDA, 0x00, 0x08, 0x01, 0x01])  # COM, length 8: FF DA 00 08 01 01
    tail = bytes(bad[3, s3:int(sizes[3])])
    bad[3, s3:s3 + len(comment)] = np.frombuffer(comment, dtype=np.uint8)
    bad[3, s3 + len(comment):s3 + len(comment) + len(tail)] = np.frombuffer(tail, dtype=np.uint8)
    sizes[3] += len(comment)
    streams = torch.from_numpy(bad).cuda()
    out_a, errcs_a, out_b, errcs_b = _decode_both_ways(torch, streams, sizes, frames, knobs)
    assert list(errcs_a) == list(errcs_b), (errcs_a, errcs_b)
    assert errcs_a[0] != 0 and errcs_a[1] != 0 and errcs_a[2] != 0 and errcs_a[3] == 0 and errcs_a[4] == 0
    assert torch.equal(out_a[3], out_b[3]) and torch.equal(out_a[4], out_b[4])
    if near == 0:
        assert torch.equal(out_a[3], frames[3]) and torch.equal(out_a[4], frames[4])
    # the reference decodes the frame with the comment to the same pixels
    _, want = ob.decode(bytes(bad[3, :int(sizes[3])]))
    assert out_a[3].cpu().numpy().tobytes() == want.tobytes()


def _runs_frames(torch, count, w, h, bits, seed, longest):
    """Piecewise-constant lines with a little noise on some segments: run mode every few samples, runs of 0 .. `longest` samples,
    RUNindex moving up and down -- what the step loop's run service (scan_group_step.inc: JLS_STEP_RARE) lives on."""
    rng = np.random.default_rng(seed)
    maxval = (1 << bits) - 1
    frames = np.zeros((count, h, w), dtype=np.uint16 if bits > 8 else np.uint8)
    for f in range(count):
        for y in range(h):
            x = 0
            level = int(rng.integers(0, maxval + 1))
            while x < w:
                n = int(rng.integers(1, longest + 2))
                if rng.random() < 0.5:
                    level = int(np.clip(level + rng.integers(-maxval // 8 - 1, maxval // 8 + 2), 0, maxval))
                seg = np.full(min(n, w - x), level, dtype=np.int64)
                if rng.random() < 0.3:
                    seg = np.clip(seg + rng.integers(-3, 4, size=seg.size), 0, maxval)
                frames[f, y, x:x + seg.size] = seg
                x += seg.size
        if f % 3 == 2 and h > 1:  # every third frame: lines that repeat the line above (runs in the context of RItype 1)
            frames[f, 1::2] = frames[f, 0:-1:2][:frames[f, 1::2].shape[0]]
    t = torch.from_numpy(frames.view(np.int16) if bits > 8 else frames).to("cuda:0")
    return t, frames


@pytest.mark.parametrize("group", [8, 16, 32])
@pytest.mark.parametrize("bits,near,w,h,longest", [(8, 0, 257, 24, 6), (8, 0, 64, 40, 40), (8, 2, 300, 16, 9), (8, 0, 5, 30, 3), (12, 0, 130, 20, 12),
                                                    (16, 3, 96, 18, 5), (8, 1, 1000, 6, 20), (8, 0, 4096, 3, 17)])
def test_run_service_of_the_step_loop_against_the_exact_decoder(torch, knobs, group, bits, near, w, h, longest):
    """The assembly step loop serves short runs itself (a run of at most G samples that is interrupted inside its line, with its
    codes inside the 32-bit window): the CPU harness runs the C++ rendering, which does not -- so this path is pinned here, on
    frames made of short runs, for every lanes-per-scan setting, lossless and near-lossless, against the exact decoder
    (decode_scans_wave: the reference's reader) and the source."""
    count = 2 * (64 // group) + 1
    frames, host = _runs_frames(torch, count, w, h, bits, seed=group + bits + near + w, longest=longest)
    enc = batch.encode_batch(frames, bits_per_sample=bits, near_lossless=near)
    assert (enc.errcs == 0).all()
    knobs.set("DECODE_GROUP", group)
    before = capi.engine_counters().get("exact_retry_scans", 0)
    out = torch.zeros_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert (errcs == 0).all()
    assert capi.engine_counters().get("exact_retry_scans", 0) == before, "a valid stream was handed to the exact decoder"
    knobs.set("EXACT_DECODER", 1)
    exact = torch.zeros_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, exact)
    assert (errcs == 0).all() and torch.equal(out, exact)
    got = out.cpu().numpy().view(host.dtype).astype(np.int64)
    assert np.abs(got - host.astype(np.int64)).max() <= near  # tolerance = NEAR, per ISO 14495-1 (0: lossless)


@pytest.mark.parametrize("comps,bits,near", [(2, 8, 2), (3, 8, 3), (4, 8, 1), (3, 12, 2), (2, 16, 5)])
def test_near_lossless_line_interleaved_scans_on_the_group_kernel(torch, knobs, comps, bits, near):
    """Near-lossless ILV_LINE scans of two to four components decode on decode_scans_group<.., NL, 1, kNear> (round 6; before: the
    pixel kernels): the speed path hands nothing to the exact decoder, agrees with it and with the oracle, and stays within NEAR."""
    host = np.stack([synth.frame_numpy(129, 33, seed=7 * f + comps, bits=bits, components=comps, kind="mixed", interleaved=True) for f in range(6)])
    frames = torch.from_numpy(host.view(np.int16) if bits > 8 else host).to("cuda:0")
    enc = batch.encode_batch(frames, bits_per_sample=bits, component_count=comps, interleave_mode=1, near_lossless=near)
    assert (enc.errcs == 0).all()
    before = capi.engine_counters().get("exact_retry_scans", 0)
    out = torch.zeros_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert (errcs == 0).all()
    assert capi.engine_counters().get("exact_retry_scans", 0) == before
    knobs.set("EXACT_DECODER", 1)
    exact = torch.zeros_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, exact)
    assert (errcs == 0).all() and torch.equal(out, exact)
    first = enc.streams[0, :int(enc.sizes[0])].cpu().numpy().tobytes()
    assert first == ob.encode(host[0], width=129, height=33, bits_per_sample=bits, component_count=comps, interleave_mode=1, near_lossless=near)
    assert out[0].cpu().numpy().tobytes() == ob.decode(first)[1].tobytes()
    got = out.cpu().numpy().view(host.dtype).astype(np.int64)
    assert np.abs(got - host.astype(np.int64)).max() <= near  # tolerance = NEAR, per ISO 14495-1


def test_near_lossless_decode_knob_keeps_the_pixel_kernels_alive(torch, knobs):
    """NEAR_DECODE_PIXELS=1 (the A/B of round 6): near-lossless single-component scans back on decode_pixels_group -- same pixels."""
    frames = synth.frames_torch(5, 300, 40, seed0=21, kind="mixed", device="cuda:0")
    enc = batch.encode_batch(frames, near_lossless=3)
    a = torch.zeros_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, a)
    assert (errcs == 0).all()
    knobs.set("NEAR_DECODE_PIXELS", 1)
    b = torch.zeros_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, b)
    assert (errcs == 0).all() and torch.equal(a, b)
    assert int((a.to(torch.int16) - frames.to(torch.int16)).abs().max()) <= 3  # tolerance = NEAR
