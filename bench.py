#!/usr/bin/env python3
"""bench.py -- JPEG-LS encode+decode throughput of the MI355X engine on BASELINE.json's headline workload.

Workload (BASELINE.json configs[1]): 4096x4096 8-bit grayscale frames, lossless.  One "step" = one pass of the hot
path over one batch: encode `--frames` device-resident frames to .jls and decode them back (per GPU).  With N > 1
ranks every rank owns its own frames (weak scaling, frames are the sharding unit) and the encoded
bitstreams are sent to rank 0 over RCCL inside the timed region (exact byte counts, point to point).  Inputs are
resident in HBM before the clock starts; `value` = frames * pixels of all ranks / (max-over-ranks time), in MPixels/s.

`python bench.py --gpus N` starts the N ranks itself (one process per GPU through torch.distributed.run on 127.0.0.1)
when it is not already running under a launcher, and fails when the box has fewer than N GPUs.

Correctness is checked outside the timed region: every frame must round-trip bit-exactly and rank 0's first frame
must hash to the committed golden produced by the reference (tests/golden/cases.json: cfg2_full).

One JSON line is printed by rank 0 (see the driver contract): it carries `roofline` for the dominant kernel
(HIP-event time of that kernel on its own stream, algorithmic bytes per launch) and `cpu_baseline` (the reference's
CPU codec, or the oracle port when oracle/_ref did not travel, timed on this box's host cores: one thread, and one
handle per thread on all cores).  Extra keys outside `value`: `batch_sweep` (throughput against the number
of frames in flight), `data_sweep` (the same engine on other data), `single_frame_ms` (one frame through the host-pointer C ABI),
`batch_api_host_buffers` (the batch API fed from pinned host memory) and `threads_abi` (the 48-symbol ABI under a pool of host
threads -- the harness `cpu_baseline.all_cores` uses for CharLS); all PCIe inclusive, none of them `value`.

`python bench.py --workload cfg4 --gpus N` runs BASELINE configs[3] as stated instead: 256 frames of 2048x2048 split over the N
ranks (strong scaling), bitstreams gathered to rank 0 inside the clock.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

WIDTH = HEIGHT = 4096  # (--workload cfg4 sets 2048)
BITS = 8
CFG4_FRAMES = 256      # BASELINE configs[3]: 256 independent 2048 x 2048 frames, seeds 100 + f (tests/golden/cases.json: cfg4_frame0..3)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
WORKSPACE_BYTES = 96 << 30  # HBM the encoder may keep for its work areas during the bench (charls_amd_set_workspace_limit)


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_description():
    """(model name, hardware threads, physical cores or None)."""
    model, cores, physical = "unknown", os.cpu_count() or 1, None
    try:
        seen, package, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    package = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if package is not None and core is not None:
                        seen.add((package, core))
                    package = core = None
        physical = len(seen) or None
    except OSError:
        pass
    return model, cores, physical


def _cpu_codec():
    from charls_amd import synth
    img = synth.frame_numpy(WIDTH, HEIGHT, seed=2, bits=BITS)
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libcharls_ref.so")
    if os.path.exists(ref_path):
        from charls_amd.capi import CharLSLibrary
        codec = CharLSLibrary(ref_path)
        # destination buffers allocated once, outside the clock (cli/benchmark.cpp:40-55,60-90), handles inside
        dst = np.empty(WIDTH * HEIGHT + WIDTH * HEIGHT // 16 + 2048, dtype=np.uint8)
        px = np.empty(WIDTH * HEIGHT, dtype=np.uint8)
        return ("reference", img, lambda: codec.encode(img, width=WIDTH, height=HEIGHT, bits_per_sample=BITS, destination=dst),
                lambda data: codec.decode(data, out=px),
                "g++ -O3 -flto -DNDEBUG -std=c++17 (oracle/Makefile: the flags of the reference's Release shared build)")
    import oracle_bind as ob
    return ("port", img, lambda: ob.encode(img, width=WIDTH, height=HEIGHT, bits_per_sample=BITS),
            lambda data: ob.decode(data), "gcc -O2 -std=c99 (oracle/Makefile)")


def cpu_baseline(seconds_budget: float = 10.0, all_cores_budget: float = 8.0):
    """Reference CPU codec on a bounded sample of the same workload (frames of cfg2, seed 2): one thread, then one
    handle per thread on every host core (independent frames, the fair comparison for batches)."""
    from concurrent.futures import ThreadPoolExecutor
    kind, img, enc, dec, flags = _cpu_codec()
    jls = enc()  # warm-up
    dec(jls)
    t_enc, t_dec, reps = [], [], 0
    start = time.perf_counter()
    while reps < 3 or (time.perf_counter() - start < seconds_budget and reps < 30):
        a = time.perf_counter()
        jls = enc()
        b = time.perf_counter()
        dec(jls)
        c = time.perf_counter()
        t_enc.append(b - a)
        t_dec.append(c - b)
        reps += 1
    mpix = WIDTH * HEIGHT / 1e6
    best_enc, best_dec = min(t_enc), min(t_dec)
    model, cores, physical = cpu_description()

    # all cores: every thread owns its codec handles (created per call, as cli/benchmark.cpp does) and codes its own copy
    # of the frame; ctypes releases the GIL for the duration of the C calls
    def worker(deadline):
        n_enc = n_dec = 0
        t_e = t_d = 0.0
        while time.perf_counter() < deadline or n_enc == 0:
            a = time.perf_counter()
            data = enc()
            b = time.perf_counter()
            dec(data)
            c = time.perf_counter()
            t_e += b - a
            t_d += c - b
            n_enc += 1
            n_dec += 1
        return n_enc, t_e, n_dec, t_d

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        results = list(pool.map(worker, [t0 + all_cores_budget] * cores))
    wall = time.perf_counter() - t0
    frames_done = sum(r[0] for r in results)
    enc_rate = sum(r[0] / r[1] for r in results) * mpix  # sum of the threads' own rates
    dec_rate = sum(r[2] / r[3] for r in results) * mpix
    return {
        "value": round(mpix / (best_enc + best_dec), 2),
        "unit": "MPixels/s encode+decode",
        "cores": 1,
        "kind": kind,
        "sample": f"{reps} x one {WIDTH}x{HEIGHT} 8-bit frame (seed 2), C ABI in-memory, best of {reps}",
        "encode_mpix_s": round(mpix / best_enc, 2),
        "decode_mpix_s": round(mpix / best_dec, 2),
        "single_frame_ms": {"encode": round(best_enc * 1e3, 1), "decode": round(best_dec * 1e3, 1)},
        "cpu_model": model,
        "host_cores": cores,
        "host_hardware_threads": cores,
        "host_physical_cores": physical,
        "compiler_flags": flags,
        "all_cores": {"threads": cores, "threads_are": "hardware threads (SMT siblings included), one codec handle per thread",
                      "physical_cores": physical, "value": round(frames_done * mpix / wall, 2), "unit": "MPixels/s encode+decode",
                      "encode_mpix_s": round(enc_rate, 2), "decode_mpix_s": round(dec_rate, 2),
                      "sample": f"{frames_done} frames in {wall:.1f} s, one handle per thread, independent frames"},
    }


def relaunch_under_torchrun(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) and return their exit status."""
    import torch
    backend = os.environ.get("CHARLS_AMD_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        have = torch.cuda.device_count()
        if have < args.gpus:
            log(f"--gpus {args.gpus} but this box has {have} GPU(s): refusing to run fewer ranks than asked for")
            return 2
    port = os.environ.get("MASTER_PORT", "29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    log("launching " + " ".join(cmd))
    return subprocess.call(cmd)


def exchange_selftest(world, rank, workload="cfg2"):
    """--selftest-exchange: the launcher, the process group and the exact-size bitstream exchange on FABRICATED streams
    (no codec, no GPU needed with CHARLS_AMD_BENCH_BACKEND=gloo), driven from a side thread exactly as the timed step drives it
    (hand_over in main()).  With --workload cfg4 the ranks own the frames batch.shard_range gives them out of 256, and rank 0
    checks that all 256 arrived, each from the rank that owns it.  Not a measurement."""
    import torch
    import torch.distributed as dist
    from charls_amd import batch
    if workload == "cfg4":
        first_frame, end_frame = batch.shard_range(CFG4_FRAMES, rank, world)
        count = end_frame - first_frame
    else:
        first_frame, count = 0, 5 + rank  # ranks own different numbers of frames
    rng = np.random.default_rng(1234 + rank)
    sizes = rng.integers(1, 5000, size=count).astype(np.uint64)
    pitch = 5120
    streams = torch.zeros((count, pitch), dtype=torch.uint8)
    for f in range(count):
        streams[f, :int(sizes[f])] = torch.from_numpy(rng.integers(0, 256, size=int(sizes[f]), dtype=np.uint8))
        streams[f, 0] = (first_frame + f) & 0xFF  # (the frame's number in the job, for the order check)
    digest = hashlib.sha256()
    for f in range(count):
        digest.update(streams[f, :int(sizes[f])].numpy().tobytes())
    mine = torch.tensor(list(digest.digest()), dtype=torch.uint8)
    all_digests = [torch.zeros(32, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(all_digests, mine)
    received, failure = {}, []

    def hand_over():
        try:
            batch.gather_streams(streams, sizes, dst=0, sink=lambda r, first, frames, sz: received.setdefault(r, []).append((first, frames, sz)))
        except BaseException as e:  # noqa: BLE001
            failure.append(e)

    mover = threading.Thread(target=hand_over, name="bitstream-gather")
    mover.start()
    mover.join()
    if failure:
        raise failure[0]
    ok = True
    if rank == 0:
        frames_seen = 0
        for r in range(world):
            h = hashlib.sha256()
            owner_first = batch.shard_range(CFG4_FRAMES, r, world)[0] if workload == "cfg4" else 0
            for first, frames, sz in sorted(received.get(r, []), key=lambda x: x[0]):
                for f in range(len(frames)):
                    h.update(frames[f][:int(sz[f])].numpy().tobytes())
                    ok = ok and int(frames[f][0]) == (owner_first + first + f) & 0xFF
                    frames_seen += 1
            ok = ok and list(h.digest()) == all_digests[r].tolist()
        if workload == "cfg4":
            ok = ok and frames_seen == CFG4_FRAMES
        line = {"selftest": "exchange", "n_gpus": world, "backend": dist.get_backend(),
                "ranks_seen_by_backend": dist.get_world_size(), "ok": ok}
        if workload == "cfg4":
            line.update(workload="cfg4", frames=frames_seen)
        print(json.dumps(line), flush=True)
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=int(os.environ.get("CHARLS_AMD_BENCH_FRAMES", "4096")),
                    help="frames per GPU per step (decoding is one serial chain per frame: throughput comes from the "
                         "number of frames in flight, see `batch_sweep` in the output; 4096 frames with their bitstreams and "
                         "decoded copies are 210 GB of the 288 GB of HBM)")
    ap.add_argument("--engine", type=int, default=0, help="0 auto, 1 serial kernel, 2 pipeline")
    ap.add_argument("--restart-interval", type=int, default=0,
                    help="NOT the headline: code every N lines as a restart interval (this library's encoder extension; "
                         "the reference's encoder cannot emit restart markers, so the streams are no longer the "
                         "reference's bytes, only decodable by it)")
    ap.add_argument("--distinct", type=int, default=0,
                    help="profiling aid, never for the bench line: synthesise only this many distinct frames and repeat them "
                         "(rocprofv3 --pmc crashes inside torch's per-frame synthesis kernels when thousands of them run under it)")
    ap.add_argument("--workload", choices=["cfg2", "cfg4"], default="cfg2",
                    help="cfg2 (default, the headline): 4096x4096 frames, --frames per GPU, weak scaling.  cfg4: BASELINE configs[3] as "
                         "stated -- 256 frames of 2048x2048 split over the ranks by batch.shard_range (STRONG scaling), bitstreams gathered "
                         "to rank 0 inside the clock")
    ap.add_argument("--threads-abi", default="",
                    help="thread counts of the `threads_abi` extra (host threads x handles through the 48-symbol ABI); default: the "
                         "host's hardware threads and twice that")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip batch_sweep / data_sweep / single_frame_ms / batch_api_host_buffers / threads_abi")
    ap.add_argument("--selftest-exchange", action="store_true", help="launcher + exchange on fabricated streams (no codec)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the arguments disagree")
        sys.exit(2)
    backend = os.environ.get("CHARLS_AMD_BENCH_BACKEND", "nccl")
    # CHARLS_AMD_BENCH_FORCE_GATHER=1 runs the RCCL path (init, barrier, exchange of bitstreams) with a single rank too
    force_gather = os.environ.get("CHARLS_AMD_BENCH_FORCE_GATHER") == "1"
    use_dist = world > 1 or force_gather
    dev = None
    if backend == "nccl":
        if torch.cuda.device_count() <= local_rank:
            log(f"rank {rank}: no GPU {local_rank} on this box ({torch.cuda.device_count()} visible)")
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"),
                              RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if rank == 0:
            log(f"process group up: backend={dist.get_backend()} nranks={dist.get_world_size()}")
    if args.selftest_exchange:
        rc = exchange_selftest(world, rank, args.workload)
        if use_dist:
            dist.destroy_process_group()
        sys.exit(rc)

    from charls_amd import batch, capi, synth
    lib = capi.load_product()
    assert lib.lib.charls_amd_device_status() == 0, "no usable GPU: the product has no CPU fallback"
    batch.set_encode_engine(args.engine, lib)
    batch.set_workspace_limit(WORKSPACE_BYTES, lib)

    global WIDTH, HEIGHT
    frames_n = args.frames
    seed0 = 2 + rank * 100003  # rank 0 frame 0 == golden cfg2_full
    golden_name = "cfg2_full"
    if args.workload == "cfg4":
        WIDTH = HEIGHT = 2048
        first_frame, end_frame = batch.shard_range(CFG4_FRAMES, rank, world)
        frames_n = end_frame - first_frame
        seed0 = 100 + first_frame  # frame f of the job has seed 100 + f on whichever rank it lands
        golden_name = "cfg4_frame0"
    if args.distinct and args.distinct < frames_n:
        base = synth.frames_torch(args.distinct, WIDTH, HEIGHT, seed0=seed0, bits=BITS, device=dev)
        frames = base.repeat((frames_n + args.distinct - 1) // args.distinct, 1, 1)[:frames_n].contiguous()
        del base
    else:
        frames = synth.frames_torch(frames_n, WIDTH, HEIGHT, seed0=seed0, bits=BITS, device=dev)
    pitch = (batch.estimated_destination_size(WIDTH, HEIGHT, BITS, 1) + 255) & ~255
    streams = torch.empty((frames_n, pitch), dtype=torch.uint8, device=dev)
    out = torch.empty_like(frames)
    torch.cuda.synchronize()

    enc_ms, dec_ms, enc_kernel_ms, dec_kernel_ms = [], [], [], []

    # The streams are final when the encoder returns: their hand-over to rank 0 starts right away, on a side stream and a
    # thread of its own, and runs UNDER the decoder (which keeps half of the SIMDs and none of the xGMI links busy); the step
    # ends when both are through.  (Round 3 ran it after the decoder: 7 x 30 GB landing on rank 0 behind 5.5 s of coding.)
    gather_stream = torch.cuda.Stream(device=dev) if (use_dist and backend == "nccl") else None

    def hand_over(enc, failure):
        try:
            if gather_stream is not None:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(gather_stream):
                    batch.gather_streams(enc.streams, enc.sizes, dst=0, sink=lambda r, first, part, sz: None)
                gather_stream.synchronize()
            else:
                batch.gather_streams(enc.streams, enc.sizes, dst=0, sink=lambda r, first, part, sz: None)
        except BaseException as e:  # noqa: BLE001 -- re-raised by the step
            failure.append(e)

    def step(timed: bool):
        t0 = time.perf_counter()
        enc = batch.encode_batch(frames, bits_per_sample=BITS, streams=streams, restart_interval=args.restart_interval, lib=lib)
        t1 = time.perf_counter()
        failure, mover = [], None
        if use_dist:
            mover = threading.Thread(target=hand_over, args=(enc, failure), name="bitstream-gather")
            mover.start()
        _, errcs, dec_t = batch.decode_batch(enc.streams, enc.sizes, out, lib=lib)
        t2 = time.perf_counter()
        if mover is not None:
            mover.join()
            if failure:
                raise failure[0]
        if timed:
            enc_ms.append((t1 - t0) * 1e3)
            dec_ms.append((t2 - t1) * 1e3)
            enc_kernel_ms.append(enc.gpu_ms)
            dec_kernel_ms.append(dec_t)
        return enc, errcs

    for _ in range(args.warmup):
        step(False)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        enc, errcs = step(True)
    barrier()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness, outside the timed region
    assert (enc.errcs == 0).all() and (errcs == 0).all(), "a frame failed"
    for f0 in range(0, frames_n, 128):  # chunked: torch.equal materialises a mask as large as its inputs
        assert torch.equal(out[f0:f0 + 128], frames[f0:f0 + 128]), "round trip is not lossless"
    bit_exact = None
    if rank == 0 and args.restart_interval == 0:
        with open(os.path.join(ROOT, "tests", "golden", "cases.json")) as f:
            golden = {c["name"]: c for c in json.load(f)}[golden_name]
        first = enc.streams[0, :int(enc.sizes[0])].cpu().numpy().tobytes()
        bit_exact = (len(first) == golden["jls_size"] and hashlib.sha256(first).hexdigest() == golden["jls_sha256"])
        assert bit_exact, "frame 0 differs from the reference's .jls (golden hash)"

    if rank == 0:
        pixels = WIDTH * HEIGHT
        mpix = pixels / 1e6
        total_frames = (CFG4_FRAMES if args.workload == "cfg4" else frames_n * world) * args.steps
        value = total_frames * mpix / elapsed
        jls_bytes = float(np.mean(enc.sizes.astype(np.float64)))
        raw_bytes = pixels * ((BITS + 7) // 8)
        # dominant kernel = largest HIP-event time per step (events recorded on the stream the kernels run on)
        stage_names = ["analyze_tiles", "plan_chains+sort_tiles", "walk_jobs+settle_chains (run chain on a side stream)", "pack_tiles",
                       "stuff_scan"]
        stages = (np.mean([k[2:7] for k in enc_kernel_ms], axis=0)
                  if enc_kernel_ms and len(enc_kernel_ms[0]) >= 7 and args.restart_interval == 0 else None)
        dk = float(np.mean([k[1] for k in dec_kernel_ms])) if dec_kernel_ms else 0.0
        dom_name, dom_ms = "decode_scans_group", dk
        if stages is not None and float(stages.max()) > dk:
            dom_name, dom_ms = stage_names[int(stages.argmax())], float(stages.max())
        # algorithmic bytes: every pixel byte and every .jls byte touched once by a pass over the batch
        alg_bytes = frames_n * (raw_bytes + jls_bytes)
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes and instruction counts of the dominant kernel come from rocprofv3 PMC passes of their own (separate runs:
        # profiles/README.md); they are used only when they were taken for THIS kernel instantiation and batch size
        traffic, traffic_source, issue = None, None, None
        try:
            pmc_file = next(p for p in (os.path.join(ROOT, "profiles", f"r0{r}_dominant_kernel_pmc.json") for r in (6, 5, 4, 3)) if os.path.exists(p))
            with open(pmc_file) as f:
                tj = json.load(f)
            if tj.get("kernel") == dom_name and int(tj.get("frames", -1)) == frames_n:
                traffic = int(tj["hbm_bytes_per_frame"] * frames_n)
                traffic_source = tj.get("source")
            if tj.get("kernel") == dom_name and dom_ms > 0 and int(tj.get("frames", -1)) == frames_n:  # (instructions per sample are those of the launch shape of that many frames)
                # instruction-issue roofline.  The ruler is MEASURED (tools/microbench/issue_ceiling.hip, profiles/r06_valu_issue_ceiling.txt):
                # the decoder's instruction mix issues 0.243 wave64 vector instructions per SIMD-cycle with one wavefront per SIMD
                # (the launch shape of this kernel), 0.250 with two, 0.336 with four -- the saturated figure a workgroup can reach;
                # plain two-operand instructions saturate at 0.5, the guide's "2 cycles per wave64 VALU instruction".  `frac` is
                # against the saturated mix; the guide's figure is given beside it.  (Rounds 3 - 5 changed this ruler every round:
                # 4, 2, 4 cycles per instruction.)
                valu = float(tj["valu_wave_instructions_per_sample"]) * frames_n * pixels
                simd_cycles = 1024 * 2.4e9
                peak = simd_cycles * 0.336
                issue = {"bound": "valu_issue", "achieved": round(valu / (dom_ms * 1e-3) / 1e9, 2), "peak": round(peak / 1e9, 1),
                         "unit": "G wave-instructions/s", "frac": round(valu / (dom_ms * 1e-3) / peak, 4),
                         "valu_wave_instructions_per_sample": tj["valu_wave_instructions_per_sample"],
                         "peak_is": "1024 SIMDs x 2.4 GHz x 0.336 wave64 vector instructions per SIMD-cycle: the decoder's instruction mix "
                                    "with four wavefronts per SIMD (profiles/r06_valu_issue_ceiling.txt); one wavefront per SIMD -- this "
                                    "kernel's launch shape -- issues 0.243",
                         "frac_of_one_wavefront_per_simd": round(valu / (dom_ms * 1e-3) / (simd_cycles * 0.243), 4),
                         "frac_of_guide_2_cycles_per_instruction": round(valu / (dom_ms * 1e-3) / (simd_cycles * 0.5), 4),
                         "source": tj.get("instruction_source")}
        except (OSError, ValueError, KeyError, StopIteration):
            pass
        line = {
            "metric": ("MPixels/s encode+decode, 256 x 2048x2048 8-bit gray sharded over the ranks, bit-exact vs CharLS" if args.workload == "cfg4" else
                       "MPixels/s encode+decode, 4096x4096 8-bit gray, bit-exact vs CharLS" if args.restart_interval == 0 else
                       "MPixels/s encode+decode, 4096x4096 8-bit gray, restart-interval extension (CharLS-decodable, not CharLS's bytes)"),
            "value": round(value, 2),
            "unit": "MPixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.workload == "cfg4" else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic (seeded gradient + noise frames, charls_amd/synth.py)" +
                    (f"; PROFILING RUN: {args.distinct} distinct frames repeated" if args.distinct else ""),
            "config": {"workload": ("BASELINE configs[3]: 256 independent 2048x2048 8-bit gray frames, lossless, split over the ranks (batch.shard_range)"
                                    if args.workload == "cfg4" else
                                    "BASELINE configs[1]: 4096x4096 8-bit gray lossless, batch of independent frames"),
                       "frames_per_gpu": frames_n, "jls_bytes_per_frame": int(jls_bytes),
                       "sharding": (f"frames over {world} rank(s), bitstreams sent to rank 0 over "
                                    f"{dist.get_backend() if use_dist else 'nccl'} inside the timed region, under the decoder") if world > 1 else "1 GPU",
                       "ranks_seen_by_backend": dist.get_world_size() if use_dist else 1,
                       "engine": args.engine, "restart_interval": args.restart_interval},
            "bit_exact_vs_reference": bit_exact,
            "encode_mpix_s": round(frames_n * mpix / (np.mean(enc_ms) * 1e-3), 2),
            "decode_mpix_s": round(frames_n * mpix / (np.mean(dec_ms) * 1e-3), 2),
            "encode_ms": round(float(np.mean(enc_ms)), 3),
            "decode_ms": round(float(np.mean(dec_ms)), 3),
            "encode_stage_ms": dict(zip(["analyze", "partition", "chains", "pack", "stuff"],
                                        [round(float(v), 3) for v in np.mean([k[2:7] for k in enc_kernel_ms], axis=0)]))
            if enc_kernel_ms and len(enc_kernel_ms[0]) >= 7 else None,
            "encode_stage_note": "HIP-event time per stage summed over the passes of the batch (tile pipeline: analyze_tiles | "
                                 "plan_chains + sort_tiles | walk_jobs + settle_chains, the run chain beside them | pack_tiles | "
                                 "stuffing); `stuff` runs on a side stream under the next pass's first stages, so the five figures "
                                 "add up to more than encode_ms",
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "kernel_ms_per_launch": round(dom_ms, 3), "algorithmic_bytes_per_launch": int(alg_bytes)},
            "issue_roofline": issue,
        }
        # (context for `value` and the CPU baseline: at N = 1 only -- the other ranks of a multi-GPU run would sit in the process
        # group for minutes while rank 0 measures them)
        if not args.no_extras and args.restart_interval == 0 and args.workload == "cfg2" and world == 1:
            line.update(extras(lib, batch, torch, frames, streams, out, enc, dev, mpix, pitch, args))
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        # ---- context, never `value`: SURVEY 8(d)'s methodology (host buffer in -> host buffer out) and the whole host CPU
        if "batch_api_host_buffers" in line:
            h = line["batch_api_host_buffers"]
            line["value_batch_api_host_buffers"] = {"value": round(1.0 / (1.0 / h["encode_mpix_s"] + 1.0 / h["decode_mpix_s"]), 1), "unit": "MPixels/s",
                                                    "frames": h["frames"], "what": "batch API, encode + decode round trip with the PCIe copies inside the clock"}
        if "cpu_baseline" in line and isinstance(line.get("threads_abi"), dict) and "value" in line["threads_abi"]:
            allc = line["cpu_baseline"]["all_cores"]["value"]
            line["threads_abi_vs_cpu_all_cores"] = {str(r["threads"]): round(r["value"] / allc, 2)
                                                    for r in [line["threads_abi"]] + line["threads_abi"].get("more", [])}
        if "batch_sweep" in line and "cpu_baseline" in line:
            allc = line["cpu_baseline"]["all_cores"]["value"]
            line["vs_cpu_all_cores"] = {str(b["frames"]): round(1.0 / (1.0 / b["encode_mpix_s"] + 1.0 / b["decode_mpix_s"]) / allc, 2)
                                        for b in line["batch_sweep"] if b["frames"] >= 256}
        print(json.dumps(line), flush=True)

    if use_dist:
        dist.destroy_process_group()


def threads_abi(lib, host_frames, pitch, threads, seconds, stagger=0.0):
    """The product under the harness `cpu_baseline.all_cores` uses for CharLS: `threads` host threads, one encoder / decoder
    handle per call (created inside the clock, as cli/benchmark.cpp does), host buffers in and out, every thread coding its
    own frame in a loop until the deadline.  Through the 48-symbol ABI only.  stagger > 0 (tools/threads_abi_probe.py, not the
    bench line): the threads start up to that many seconds apart and idle a random while between their calls -- arrival without
    any alignment."""
    import random
    from concurrent.futures import ThreadPoolExecutor
    from charls_amd import capi
    mpix = WIDTH * HEIGHT / 1e6
    dsts = [np.empty(pitch, dtype=np.uint8) for _ in range(threads)]
    pxs = [np.empty(WIDTH * HEIGHT, dtype=np.uint8) for _ in range(threads)]
    start_line = threading.Barrier(threads)
    lib.encode(host_frames[0], width=WIDTH, height=HEIGHT, bits_per_sample=BITS, destination=dsts[0])  # (module load, first allocations)

    def worker(t):
        img = host_frames[t % len(host_frames)]
        n = 0
        t_e = t_d = 0.0
        ok = True
        start_line.wait()
        rng = random.Random(t)
        if stagger:
            time.sleep(rng.uniform(0, stagger))
        deadline = time.perf_counter() + seconds
        while time.perf_counter() < deadline or n == 0:
            a = time.perf_counter()
            data = lib.encode(img, width=WIDTH, height=HEIGHT, bits_per_sample=BITS, destination=dsts[t])
            b = time.perf_counter()
            lib.decode(data, out=pxs[t])
            c = time.perf_counter()
            if stagger:
                time.sleep(rng.uniform(0, stagger / 10))
            t_e += b - a
            t_d += c - b
            n += 1
        ok = pxs[t].tobytes() == img.tobytes()
        return n, t_e, t_d, ok

    before = capi.engine_counters(lib)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as pool:
        results = list(pool.map(worker, range(threads)))
    wall = time.perf_counter() - t0
    after = capi.engine_counters(lib)
    assert all(r[3] for r in results), "a thread's round trip is not lossless"
    done = sum(r[0] for r in results)
    return {"threads": threads, "value": round(done * mpix / wall, 1), "unit": "MPixels/s encode+decode",
            "encode_mpix_s": round(sum(r[0] / r[1] for r in results) * mpix, 1),
            "decode_mpix_s": round(sum(r[0] / r[2] for r in results) * mpix, 1),
            "sample": f"{done} frames in {wall:.1f} s, one handle per call and thread, {min(threads, len(host_frames))} distinct frames",
            "engine_counters": {k: after[k] - before[k] if k != "largest_launch" else after[k] for k in after}}


DATA_KINDS = ("gradient", "mixed", "hard", "zero", "tulips")


def data_frames(torch, kind, count, dev, into):
    """`count` frames of one kind of data into `into` (a (count, H, W) uint8 device tensor).  tulips: the reference's own
    natural test image (test/conformance material held under tests/golden/refdata) tiled 8 x 8 to 4096 x 4096, rolled by a
    different offset per frame so that the frames differ."""
    from charls_amd import synth
    if kind != "tulips":
        distinct = min(count, 64)
        base = synth.frames_torch(distinct, WIDTH, HEIGHT, seed0=2, bits=BITS, kind=kind, device=dev)
        for f0 in range(0, count, distinct):
            n = min(distinct, count - f0)
            into[f0:f0 + n] = base[:n]
        return f"synth kind={kind}, {distinct} distinct frames repeated"
    import common
    img, _ = common.read_pnm("tulips-gray-8bit-512-512.pgm")
    tile = torch.from_numpy(np.ascontiguousarray(img)).to(dev)
    big = tile.repeat(HEIGHT // tile.shape[0], WIDTH // tile.shape[1])
    for f in range(count):
        into[f] = torch.roll(big, shifts=(17 * (f % 64), 29 * (f % 64)), dims=(0, 1))
    return "tests/golden/refdata/tulips-gray-8bit-512-512.pgm tiled 8 x 8, rolled per frame (64 distinct)"


def data_sweep(lib, batch, torch, frames, streams, out, dev, mpix, count):
    """the same engine on other data -- flat patches (run mode), +-32 noise, an all-zero frame, a natural image.
    Encode / decode MPix/s of `count` frames each and what the encoder's speculation did.  Uses out[:count] for the
    frames and out[count:2 count] for the decoded copies."""
    import ctypes as C
    L = lib.lib
    L.charls_amd_speculation_counters.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
    L.charls_amd_speculation_counters.restype = C.c_int32

    def spec():
        v = (C.c_uint64 * 6)()
        L.charls_amd_speculation_counters(v, 6)
        return np.array(list(v), dtype=np.int64)

    rows = []
    src, dst = out[:count], out[count:2 * count]
    for kind in DATA_KINDS:
        note = data_frames(torch, kind, count, dev, src)
        torch.cuda.synchronize()
        s0 = spec()
        best_e = best_d = None
        for _ in range(2):
            a = time.perf_counter()
            e = batch.encode_batch(src, bits_per_sample=BITS, streams=streams[:count], lib=lib)
            torch.cuda.synchronize()
            b = time.perf_counter()
            _, errcs, _ = batch.decode_batch(e.streams, e.sizes, dst, lib=lib)
            torch.cuda.synchronize()
            c = time.perf_counter()
            best_e = b - a if best_e is None else min(best_e, b - a)
            best_d = c - b if best_d is None else min(best_d, c - b)
        s1 = spec()
        assert (e.errcs == 0).all() and (errcs == 0).all()
        for f0 in range(0, count, 128):
            assert torch.equal(dst[f0:f0 + 128], src[f0:f0 + 128]), f"data_sweep {kind}: round trip is not lossless"
        d = (s1 - s0) // 2
        rows.append({"data": kind, "frames": count, "encode_mpix_s": round(count * mpix / best_e, 1), "decode_mpix_s": round(count * mpix / best_d, 1),
                     "jls_bytes_per_frame": int(np.mean(e.sizes.astype(np.float64))),
                     "speculation": {"jobs": int(d[0]), "jobs_walked_again": int(d[1]), "run_jobs": int(d[2]), "run_jobs_walked_again": int(d[3]),
                                     "rare_segments": int(d[4]), "rare_serial_fallbacks": int(d[5])},
                     "source": note})
    return rows


def near_lossless_row(lib, batch, torch, frames, streams, out, mpix, count, near=2):
    """the same frames coded near-lossless (NEAR = 2).  Both directions are ONE chain per frame here (the
    encoder predicts from reconstructed samples: scan_group_encode.hip; the decoder is decode_scans_group<.., kNear>), so the
    figure is the number of frames in flight times a stream's rate.  One pass; every sample within NEAR of its source."""
    src, dst = frames[:count], out[:count]
    torch.cuda.synchronize()
    a = time.perf_counter()
    e = batch.encode_batch(src, bits_per_sample=BITS, near_lossless=near, streams=streams[:count], lib=lib)
    torch.cuda.synchronize()
    b = time.perf_counter()
    _, errcs, _ = batch.decode_batch(e.streams, e.sizes, dst, lib=lib)
    torch.cuda.synchronize()
    c = time.perf_counter()
    assert (e.errcs == 0).all() and (errcs == 0).all()
    worst = 0
    for f0 in range(0, count, 64):
        worst = max(worst, int((dst[f0:f0 + 64].to(torch.int16) - src[f0:f0 + 64].to(torch.int16)).abs().max().item()))
    assert worst <= near, f"near-lossless round trip: a sample is {worst} away from its source (NEAR = {near})"
    return {"near": near, "frames": count, "encode_mpix_s": round(count * mpix / (b - a), 1), "decode_mpix_s": round(count * mpix / (c - b), 1),
            "jls_bytes_per_frame": int(np.mean(e.sizes.astype(np.float64))), "largest_difference": worst,
            "note": "one pass; |decoded - source| <= NEAR checked for every sample (tolerance = NEAR, ISO 14495-1); not part of `value`"}


def extras(lib, batch, torch, frames, streams, out, enc, dev, mpix, pitch, args):
    """Context for `value` (never part of it): throughput against the number of frames in flight, one frame through the
    host-pointer C ABI (the literal reading of BASELINE configs[1]), a batch with the PCIe copies inside the clock, the
    48-symbol ABI under a pool of host threads, and the same engine on other data."""
    result = {}
    # ---- throughput against the batch size (frames resident in HBM, as in the timed region)
    sweep = []
    for n in (1, 64, 256, 1024, 4096):
        if n > frames.shape[0]:
            continue
        torch.cuda.synchronize()
        a = time.perf_counter()
        e = batch.encode_batch(frames[:n], bits_per_sample=BITS, streams=streams[:n], lib=lib)
        torch.cuda.synchronize()
        b = time.perf_counter()
        batch.decode_batch(e.streams, e.sizes, out[:n], lib=lib)
        torch.cuda.synchronize()
        c = time.perf_counter()
        sweep.append({"frames": n, "encode_mpix_s": round(n * mpix / (b - a), 1), "decode_mpix_s": round(n * mpix / (c - b), 1)})
    result["batch_sweep"] = sweep
    # ---- the same engine on other data (SURVEY 8d, C2 variants)
    if frames.shape[0] >= 512:
        try:
            result["data_sweep"] = data_sweep(lib, batch, torch, frames, streams, out, dev, mpix, min(1024, frames.shape[0] // 2))
        except Exception as e:  # noqa: BLE001 -- an extra must not take the bench line with it
            result["data_sweep"] = {"error": repr(e)}
    if frames.shape[0] >= 1024:
        try:
            result["near_lossless"] = near_lossless_row(lib, batch, torch, frames, streams, out, mpix, 1024)
        except Exception as e:  # noqa: BLE001
            result["near_lossless"] = {"error": repr(e)}
    # ---- one frame through the host-pointer C ABI (handle created inside the clock, as cli/benchmark.cpp does); the
    # batch-sized work areas go first: giving ~90 GB back to the driver takes seconds and is not part of coding a frame
    batch.release_work_areas(lib)
    img = frames[0].cpu().numpy()
    dst = np.empty(pitch, dtype=np.uint8)          # destinations allocated outside the clock, handles inside it: the
    px = np.empty(WIDTH * HEIGHT, dtype=np.uint8)  # methodology of the reference's cli/benchmark.cpp:40-55,60-90
    lib.encode(img, width=WIDTH, height=HEIGHT, bits_per_sample=BITS, destination=dst)  # warm-up (allocations, module load)
    best = 1e9
    for _ in range(5):
        a = time.perf_counter()
        jls = lib.encode(img, width=WIDTH, height=HEIGHT, bits_per_sample=BITS, destination=dst)
        best = min(best, time.perf_counter() - a)
    jls = jls.tobytes()
    b = time.perf_counter()
    lib.decode(jls, out=px)
    c = time.perf_counter()
    assert px.tobytes() == img.tobytes()
    result["single_frame_ms"] = {"encode": round(best * 1e3, 2), "decode": round((c - b) * 1e3, 1),
                                 "path": "charls_jpegls_encoder_encode_from_buffer / charls_jpegls_decoder_decode_to_buffer, "
                                         "host buffers in and out (PCIe inclusive), one 4096x4096 8-bit frame; handle created "
                                         "inside the clock, destination allocated outside (cli/benchmark.cpp); encode: best of 5"}
    # ---- a batch with the PCIe copies inside the clock: pinned host frames -> HBM -> encode -> .jls back to pinned host
    # memory, and the way back for decode.  The batch goes in chunks of 64 frames: the upload of chunk k + 1 and the download
    # of chunk k - 1 run on copy streams under the coding of chunk k (the library call blocks the calling thread, not the
    # copies that were queued before it).  This is the BATCH API fed from host buffers, not the 48-symbol ABI (that is
    # `threads_abi` below).
    chunk = 64
    up, down, main = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.current_stream(dev)

    def host_buffers(n):
        host_frames = torch.empty((n, HEIGHT, WIDTH), dtype=torch.uint8).pin_memory()
        host_frames.copy_(frames[:n])
        host_streams = torch.empty((n, pitch), dtype=torch.uint8).pin_memory()
        batch.encode_batch(out[:chunk], bits_per_sample=BITS, streams=streams[:chunk], lib=lib)  # (work areas of a chunk: allocated once, outside the clock)
        sizes = np.zeros(n, dtype=np.uint64)
        torch.cuda.synchronize()

        def pipelined(upload, code, download, chunks):
            """upload(k) / download(k) queue asynchronous copies of chunk k on the current stream; code(k) blocks."""
            ready = []
            with torch.cuda.stream(up):
                upload(0)
                ready.append(up.record_event())
            for k in range(chunks):
                if k + 1 < chunks:
                    with torch.cuda.stream(up):
                        upload(k + 1)
                        ready.append(up.record_event())
                main.wait_event(ready[k])
                code(k)
                coded = main.record_event()
                with torch.cuda.stream(down):
                    down.wait_event(coded)
                    download(k)
            torch.cuda.synchronize()

        def rng(k):
            return slice(k * chunk, min(n, (k + 1) * chunk))

        def encode_chunk(k):
            e = batch.encode_batch(out[rng(k)], bits_per_sample=BITS, streams=streams[rng(k)], lib=lib)
            sizes[rng(k)] = e.sizes

        def download_streams(k):  # one copy of exactly its bytes per stream (a strided 2-D copy of the chunk's slots takes ten times as long)
            for f in range(k * chunk, min(n, (k + 1) * chunk)):
                host_streams[f, :int(sizes[f])].copy_(streams[f, :int(sizes[f])], non_blocking=True)

        a = time.perf_counter()
        pipelined(lambda k: out[rng(k)].copy_(host_frames[rng(k)], non_blocking=True), encode_chunk, download_streams, -(-n // chunk))
        b = time.perf_counter()

        # (decoding is one serial chain per frame: its rate is the number of frames in flight, so the whole batch is ONE chunk;
        # the decoded frames come back into the pinned frames' own buffer and are compared with the originals on the device)
        def upload_streams(k):
            for f in range(n):
                streams[f, :int(sizes[f])].copy_(host_streams[f, :int(sizes[f])], non_blocking=True)

        host_frames.zero_()
        pipelined(upload_streams,
                  lambda k: batch.decode_batch(streams[:n], sizes, out[:n], lib=lib),
                  lambda k: host_frames.copy_(out[:n], non_blocking=True), 1)
        c = time.perf_counter()
        for f0 in range(0, n, 128):
            assert torch.equal(host_frames[f0:f0 + 128].to(dev), frames[f0:f0 + 128])
        return {"frames": n, "encode_mpix_s": round(n * mpix / (b - a), 1), "decode_mpix_s": round(n * mpix / (c - b), 1)}

    sizes_to_run = [n for n in (256, 1024) if n <= frames.shape[0]] or [frames.shape[0]]
    rows = []
    for n in sizes_to_run:
        try:
            rows.append(host_buffers(n))
        except Exception as e:  # noqa: BLE001 -- e.g. the box cannot pin 35 GB
            rows.append({"frames": n, "error": repr(e)})
    good = [r for r in rows if "error" not in r]
    if good:
        result["batch_api_host_buffers"] = {**good[0], "more": good[1:] + [r for r in rows if "error" in r],
                                            "path": "the BATCH API fed from pinned host buffers: H2D + batch calls + D2H inside the clock; encode in "
                                                    "chunks of 64 frames with the copies of the neighbouring chunks under the coding of a chunk, decode "
                                                    "as one batch (PCIe inclusive; never `value`; not the 48-symbol ABI -- see threads_abi)"}
    # ---- the 48-symbol ABI under a pool of host threads, the harness of cpu_baseline.all_cores (SURVEY 8b "Threading")
    try:
        counts = [int(x) for x in args.threads_abi.split(",") if x] or [os.cpu_count() or 64, 2 * (os.cpu_count() or 64)]
        distinct = min(max(counts), 256, frames.shape[0])
        host = frames[:distinct].cpu().numpy()
        host_list = [host[i] for i in range(distinct)]
        batch.release_work_areas(lib)
        runs = []
        for t in counts:
            runs.append(threads_abi(lib, host_list, pitch, t, seconds=10.0))
        batch.release_work_areas(lib)
        result["threads_abi"] = {**runs[0], "more": runs[1:],
                                 "path": "charls_jpegls_encoder_* / charls_jpegls_decoder_* only, host buffers in and out (PCIe inclusive), T host "
                                         "threads x one handle per call -- the harness of cpu_baseline.all_cores; concurrent calls are merged into "
                                         "shared launches by the library (engine_counters)"}
    except Exception as e:  # noqa: BLE001
        result["threads_abi"] = {"error": repr(e)}
    # ---- `value` on a natural image: the bench's own batch (all frames in flight) of the reference's test image tiled to the frame
    # size -- the synthetic frames are the decoder's friendly case (a run event every 80 samples; the natural image: every 9).
    # Last of the extras: the frames buffer becomes the destination of the decode.
    try:
        n = frames.shape[0]
        batch.release_work_areas(lib)
        note = data_frames(torch, "tulips", n, dev, out)
        torch.cuda.synchronize()
        best_e = best_d = None
        for _ in range(2):
            a = time.perf_counter()
            e = batch.encode_batch(out, bits_per_sample=BITS, streams=streams, lib=lib)
            torch.cuda.synchronize()
            b = time.perf_counter()
            _, errcs, _ = batch.decode_batch(e.streams, e.sizes, frames, lib=lib)
            torch.cuda.synchronize()
            c = time.perf_counter()
            best_e = b - a if best_e is None else min(best_e, b - a)
            best_d = c - b if best_d is None else min(best_d, c - b)
        assert (e.errcs == 0).all() and (errcs == 0).all()
        for f0 in range(0, n, 128):
            assert torch.equal(frames[f0:f0 + 128], out[f0:f0 + 128]), "natural image: round trip is not lossless"
        result["value_natural_image"] = {"value": round(n * mpix / (best_e + best_d), 1), "unit": "MPixels/s encode+decode", "frames": n,
                                         "encode_mpix_s": round(n * mpix / best_e, 1), "decode_mpix_s": round(n * mpix / best_d, 1),
                                         "jls_bytes_per_frame": int(np.mean(e.sizes.astype(np.float64))), "data": note}
    except Exception as e:  # noqa: BLE001
        result["value_natural_image"] = {"error": repr(e)}
    return result


if __name__ == "__main__":
    main()
