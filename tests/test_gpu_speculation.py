"""The speculative stages of the lossless encoder, forced on the MI355X.

The tile pipeline codes every context chain and the run chain in JOBS that start from a guessed state;
settle_chains / settle_runs check every job boundary and code a job again where the guess was wrong.  With the default
job sizes and warm-ups nearly every guess is right on natural data, so the ordinary parity tests never execute the
re-walks on the GPU.  Here the knobs (read per call, runtime.hip: TileLayout) make every boundary disagree -- jobs of 32
events with no warm-up -- and full-range noise, which never converges, goes through with the DEFAULT knobs.  The bytes must
be the oracle's in every case, and charls_amd_speculation_counters must show that the re-walks really ran.

Reference behaviour matched: src/regular_mode_context.hpp:45-93, src/run_mode_context.hpp:65-83.  GPU only."""
import ctypes as C

import numpy as np
import pytest

import oracle_bind as ob
from charls_amd import batch, capi, synth

pytestmark = pytest.mark.gpu

FORCED = {"CHARLS_AMD_JOB_EVENTS": "32", "CHARLS_AMD_WARM_EVENTS": "0", "CHARLS_AMD_RUN_JOB_EVENTS": "32",
          "CHARLS_AMD_RUN_WARM_EVENTS": "0"}


@pytest.fixture(scope="module")
def lib():
    L = capi.load_product()
    assert L.lib.charls_amd_device_status() == 0
    L.lib.charls_amd_speculation_counters.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
    L.lib.charls_amd_speculation_counters.restype = C.c_int32
    return L


def counters(lib):
    out = (C.c_uint64 * 4)()
    assert lib.lib.charls_amd_speculation_counters(out, 4) == 4
    return np.array(list(out), dtype=np.int64)


def force(knobs, **more):
    for k, v in {**FORCED, **more}.items():
        knobs.set(k, v)


@pytest.mark.parametrize("kind,bits,w,h,seed", [("mixed", 8, 512, 512, 21), ("gradient", 8, 1024, 1024, 22),
                                                ("hard", 8, 1024, 1024, 23), ("mixed", 16, 512, 512, 24),
                                                ("mixed", 12, 1024, 1024, 25), ("hard", 16, 1024, 1024, 26)])
def test_every_job_boundary_disagrees_gray(lib, knobs, kind, bits, w, h, seed):
    force(knobs)
    img = synth.frame_numpy(w, h, seed=seed, bits=bits, kind=kind)
    before = counters(lib)
    got = lib.encode(img, width=w, height=h, bits_per_sample=bits)
    did = counters(lib) - before
    assert got == ob.encode(img, width=w, height=h, bits_per_sample=bits)
    assert did[0] >= w * h // 32 // 2 and did[1] > did[0] // 4, did  # (jobs of 32 events: whole rounds of the walkers) most of them walked again
    if kind == "mixed":
        assert did[2] > 1 and did[3] > 0, did  # flat patches: the run chain has jobs, and they were settled too
    assert lib.decode(got)[1].tobytes() == img.tobytes()


@pytest.mark.parametrize("bits,w,h,ct", [(8, 512, 512, 0), (8, 1024, 1024, 1), (16, 512, 512, 0)])
def test_every_job_boundary_disagrees_line_interleaved_rgb(lib, knobs, bits, w, h, ct):
    force(knobs)
    img = synth.frame_numpy(w, h, seed=31 + bits, bits=bits, components=3, kind="mixed", interleaved=True)
    kw = dict(width=w, height=h, bits_per_sample=bits, component_count=3, interleave_mode=1, color_transformation=ct)
    before = counters(lib)
    got = lib.encode(img, **kw)
    did = counters(lib) - before
    assert got == ob.encode(img, **kw)
    assert did[1] > did[0] // 4 and did[3] > 0, did


def test_partial_warm_up_settles_some_jobs_only(lib, knobs):
    """A warm-up that is too short for SOME chains: right and wrong guesses in the same launch."""
    force(knobs, CHARLS_AMD_JOB_EVENTS="64", CHARLS_AMD_WARM_EVENTS="48", CHARLS_AMD_RUN_JOB_EVENTS="64",
          CHARLS_AMD_RUN_WARM_EVENTS="32")
    img = synth.frame_numpy(1024, 1024, seed=41, kind="mixed")
    before = counters(lib)
    got = lib.encode(img, width=1024, height=1024)
    did = counters(lib) - before
    assert got == ob.encode(img, width=1024, height=1024)
    assert 0 < did[1] < did[0], did


@pytest.mark.parametrize("bits", [8, 16])
def test_full_range_noise_never_converges_default_knobs(lib, bits):
    """|Errval| >> N makes C a random walk that both walks share: jobs disagree whatever the warm-up."""
    w = h = 1024
    img = synth.frame_numpy(w, h, seed=51, bits=bits, kind="noise")
    before = counters(lib)
    got = lib.encode(img, width=w, height=h, bits_per_sample=bits, destination_size=w * h * (bits // 8) * 2 + 4096)
    did = counters(lib) - before
    assert got == ob.encode(img, width=w, height=h, bits_per_sample=bits, destination_size=w * h * (bits // 8) * 2 + 4096)
    assert did[1] > 0, did
    assert lib.decode(got)[1].tobytes() == img.tobytes()


def test_batch_mixing_converging_and_non_converging_frames(lib):
    import torch
    w = h = 1024
    kinds = ["gradient", "noise", "mixed", "noise", "hard", "zero"]
    host = [synth.frame_numpy(w, h, seed=60 + i, kind=k) for i, k in enumerate(kinds)]
    frames = torch.from_numpy(np.stack(host)).cuda()
    streams = torch.empty((len(kinds), 2 * w * h + 4096), dtype=torch.uint8, device="cuda")
    before = counters(lib)
    enc = batch.encode_batch(frames, streams=streams)
    did = counters(lib) - before
    got = enc.streams.cpu().numpy()
    for i, img in enumerate(host):
        want = ob.encode(img, width=w, height=h, destination_size=2 * w * h + 4096)
        assert enc.errcs[i] == 0 and got[i, :int(enc.sizes[i])].tobytes() == want, kinds[i]
    assert 0 < did[1] < did[0], did  # the noise frames' jobs were walked again, the others' were not
    out = torch.empty_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out)
    assert (errcs == 0).all() and torch.equal(out, frames)


def test_batch_with_forced_knobs_and_two_passes(lib, knobs):
    """Several scans per launch AND several passes (a small workspace limit) with every boundary disagreeing."""
    import torch
    force(knobs)
    w = h = 512
    host = [synth.frame_numpy(w, h, seed=70 + i, kind="mixed" if i % 2 else "hard") for i in range(6)]
    frames = torch.from_numpy(np.stack(host)).cuda()
    batch.set_workspace_limit(16 << 20, lib)
    try:
        before = counters(lib)
        enc = batch.encode_batch(frames)
        did = counters(lib) - before
    finally:
        batch.set_workspace_limit(0, lib)
        batch.release_work_areas(lib)
    got = enc.streams.cpu().numpy()
    for i, img in enumerate(host):
        assert enc.errcs[i] == 0 and got[i, :int(enc.sizes[i])].tobytes() == ob.encode(img, width=w, height=h), i
    assert did[1] > did[0] // 4 and did[3] > 0, did


@pytest.mark.parametrize("chunk,warm", [("64", "0"), ("256", "512"), ("4096", "1024")])
def test_speculative_stuffing_with_failing_guesses(lib, knobs, chunk, warm):
    """Stage E in its speculative form (passes of more than 8 scans; speculative_stuffing.hip) with chunks and warm-ups so small
    that entry-state guesses fail: chunks walked again by the resolving wavefront, and -- 64-byte chunks without warm-up --
    scans given up and stuffed in sequence by their first wavefront.  Bytes against the oracle."""
    import torch
    knobs.set("SPEC_CHUNK", chunk)
    knobs.set("SPEC_WARM", warm)
    w = h = 256
    kinds = ["mixed", "noise", "gradient", "hard", "zero", "noise", "mixed", "hard", "gradient", "noise", "mixed", "zero"]
    host = [synth.frame_numpy(w, h, seed=90 + i, kind=k) for i, k in enumerate(kinds)]
    host[1][:] = 255  # all ones in the raw stream: 0xFF bytes everywhere
    frames = torch.from_numpy(np.stack(host)).cuda()
    streams = torch.empty((len(kinds), 2 * w * h + 4096), dtype=torch.uint8, device="cuda")
    enc = batch.encode_batch(frames, streams=streams)
    got = enc.streams.cpu().numpy()
    for i, img in enumerate(host):
        want = ob.encode(img, width=w, height=h, destination_size=2 * w * h + 4096)
        assert enc.errcs[i] == 0 and got[i, :int(enc.sizes[i])].tobytes() == want, (i, kinds[i])


def _six(lib):
    out = (C.c_uint64 * 6)()
    assert lib.lib.charls_amd_speculation_counters(out, 6) == 6
    return np.array(list(out), dtype=np.int64)


@pytest.mark.parametrize("rare_warm,serial", [(None, 0), ("0", 1), ("24", 1)])
def test_rarer_run_context_segments_and_their_fallback(lib, knobs, rare_warm, serial):
    """The exact walk of the rarer run-interruption context goes in segments of its event list from a warm-up counted in
    its own events; a wrong guess makes one lane walk the list again.  Both ways on the MI355X: the oracle's bytes, and the
    counters say which of the two ran."""
    w, h = 1536, 1024
    img = synth.frame_numpy(w, h, seed=41, kind="mixed")
    if rare_warm is not None:
        knobs.set("RARE_WARM_EVENTS", rare_warm)
    before = _six(lib)
    got = lib.encode(img, width=w, height=h)
    did = _six(lib) - before
    assert got == ob.encode(img, width=w, height=h)
    assert did[4] >= 3 and did[5] == serial, did
