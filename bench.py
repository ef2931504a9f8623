#!/usr/bin/env python3
"""bench.py -- JPEG-LS encode+decode throughput of the MI355X engine on BASELINE.json's headline workload.

Workload (BASELINE.json configs[1]): 4096x4096 8-bit grayscale frames, lossless.  One "step" = one pass of the hot
path over one batch: encode `--frames` device-resident frames to .jls and decode them back (per GPU).  With N > 1
ranks every rank owns its own frames (weak scaling, frames are the sharding unit) and the encoded
bitstreams are gathered to rank 0 over RCCL inside the timed region.  Inputs are resident in HBM before the clock
starts; `value` = frames * pixels of all ranks / (max-over-ranks time), in MPixels/s.

Correctness is checked outside the timed region: every frame must round-trip bit-exactly and rank 0's first frame
must hash to the committed golden produced by the reference (tests/golden/cases.json: cfg2_full).

One JSON line is printed by rank 0 (see the driver contract): it carries `roofline` for the dominant kernel
(HIP-event time of that kernel on its own stream, algorithmic bytes per launch) and `cpu_baseline` (the reference's
CPU codec, or the oracle port when oracle/_ref did not travel, timed on this box's host cores).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

WIDTH = HEIGHT = 4096
BITS = 8
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(seconds_budget: float = 12.0):
    """Reference CPU codec (single thread) on a bounded sample of the same workload: frames of cfg2 (seed 2)."""
    from charls_amd import synth
    img = synth.frame_numpy(WIDTH, HEIGHT, seed=2, bits=BITS)
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libcharls_ref.so")
    kind = "reference" if os.path.exists(ref_path) else "port"
    if kind == "reference":
        from charls_amd.capi import CharLSLibrary
        codec = CharLSLibrary(ref_path)
        enc = lambda: codec.encode(img, width=WIDTH, height=HEIGHT, bits_per_sample=BITS)  # noqa: E731
        dec = lambda data: codec.decode(data)  # noqa: E731
    else:
        import oracle_bind as ob
        enc = lambda: ob.encode(img, width=WIDTH, height=HEIGHT, bits_per_sample=BITS)  # noqa: E731
        dec = lambda data: ob.decode(data)  # noqa: E731
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
    return {
        "value": round(mpix / (best_enc + best_dec), 2),
        "unit": "MPixels/s encode+decode",
        "cores": 1,
        "kind": kind,
        "sample": f"{reps} x one 4096x4096 8-bit frame (seed 2), C ABI in-memory, best of {reps}",
        "encode_mpix_s": round(mpix / best_enc, 2),
        "decode_mpix_s": round(mpix / best_dec, 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=int(os.environ.get("CHARLS_AMD_BENCH_FRAMES", "4096")),
                    help="frames per GPU per step (decoding is one serial chain per frame: throughput comes from "
                         "the number of concurrent frames; 4096 = four wavefronts per SIMD = what LDS holds, and with "
                         "their bitstreams and decoded copies 210 GB of the 288 GB of HBM)")
    ap.add_argument("--engine", type=int, default=0, help="0 auto, 1 serial kernel, 2 pipeline")
    ap.add_argument("--restart-interval", type=int, default=0,
                    help="NOT the headline: code every N lines as a restart interval (this library's encoder extension; "
                         "the reference's encoder cannot emit restart markers, so the streams are no longer the "
                         "reference's bytes, only decodable by it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from charls_amd import batch, capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # CHARLS_AMD_BENCH_FORCE_GATHER=1 runs the RCCL path (init, barrier, gather of bitstreams) with a single rank too
    force_gather = os.environ.get("CHARLS_AMD_BENCH_FORCE_GATHER") == "1"
    if world > 1 or force_gather:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"),
                              RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    lib = capi.load_product()
    assert lib.lib.charls_amd_device_status() == 0, "no usable GPU: the product has no CPU fallback"
    batch.set_encode_engine(args.engine, lib)

    frames_n = args.frames
    seed0 = 2 + rank * 100003  # rank 0 frame 0 == golden cfg2_full
    frames = synth.frames_torch(frames_n, WIDTH, HEIGHT, seed0=seed0, bits=BITS, device=dev)
    pitch = (batch.estimated_destination_size(WIDTH, HEIGHT, BITS, 1) + 255) & ~255
    streams = torch.empty((frames_n, pitch), dtype=torch.uint8, device=dev)
    out = torch.empty_like(frames)
    torch.cuda.synchronize()

    enc_ms, dec_ms, enc_kernel_ms, dec_kernel_ms = [], [], [], []

    def step(timed: bool):
        t0 = time.perf_counter()
        enc = batch.encode_batch(frames, bits_per_sample=BITS, streams=streams, restart_interval=args.restart_interval, lib=lib)
        t1 = time.perf_counter()
        _, errcs, dec_t = batch.decode_batch(enc.streams, enc.sizes, out, lib=lib)
        t2 = time.perf_counter()
        if world > 1 or force_gather:
            batch.gather_streams(enc.streams, enc.sizes, dst=0, sink=lambda r, first, part, sz: None)
        if timed:
            enc_ms.append((t1 - t0) * 1e3)
            dec_ms.append((t2 - t1) * 1e3)
            enc_kernel_ms.append(enc.gpu_ms)
            dec_kernel_ms.append(dec_t)
        return enc, errcs

    for _ in range(args.warmup):
        step(False)

    def barrier():
        if world > 1 or force_gather:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        enc, errcs = step(True)
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1 or force_gather:
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
            golden = {c["name"]: c for c in json.load(f)}["cfg2_full"]
        first = enc.streams[0, :int(enc.sizes[0])].cpu().numpy().tobytes()
        bit_exact = (len(first) == golden["jls_size"] and hashlib.sha256(first).hexdigest() == golden["jls_sha256"])
        assert bit_exact, "frame 0 differs from the reference's .jls (golden hash)"

    if rank == 0:
        pixels = WIDTH * HEIGHT
        total_frames = frames_n * world * args.steps
        value = total_frames * pixels / 1e6 / elapsed
        jls_bytes = float(np.mean(enc.sizes.astype(np.float64)))
        raw_bytes = pixels * ((BITS + 7) // 8)
        # dominant kernel = largest HIP-event time per step (events recorded on the stream the kernels run on)
        stage_names = ["analyze_rows", "chain_offsets+scatter_events", "bias_chains+code_events", "sum/scan/write_raw_bits", "stuff_scan"]
        stages = (np.mean([k[2:7] for k in enc_kernel_ms], axis=0)
                  if enc_kernel_ms and len(enc_kernel_ms[0]) >= 7 and args.restart_interval == 0 else None)
        dk = float(np.mean([k[1] for k in dec_kernel_ms])) if dec_kernel_ms else 0.0
        dom_name, dom_ms = "decode_scans_fast", dk
        if stages is not None and float(stages.max()) > dk:
            dom_name, dom_ms = stage_names[int(stages.argmax())], float(stages.max())
        # algorithmic bytes: every pixel byte and every .jls byte touched once by a pass over the batch
        alg_bytes = frames_n * (raw_bytes + jls_bytes)
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        try:  # HBM bytes per frame of the dominant kernel measured with rocprofv3 PMC (profiles/r01_traffic.json)
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                tj = json.load(f)
            if tj.get("kernel") == dom_name:
                traffic = int(tj["hbm_bytes_per_frame"] * frames_n)
        except (OSError, ValueError, KeyError):
            pass
        line = {
            "metric": "MPixels/s encode+decode, 4096x4096 8-bit gray, bit-exact vs CharLS" if args.restart_interval == 0 else
                      "MPixels/s encode+decode, 4096x4096 8-bit gray, restart-interval extension (CharLS-decodable, not CharLS's bytes)",
            "value": round(value, 2),
            "unit": "MPixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic (seeded gradient + noise frames, charls_amd/synth.py)",
            "config": {"workload": "BASELINE configs[1]: 4096x4096 8-bit gray lossless, batch of independent frames",
                       "frames_per_gpu": frames_n, "jls_bytes_per_frame": int(jls_bytes),
                       "sharding": f"frames over {world} rank(s), RCCL gather of bitstreams to rank 0" if world > 1 else "1 GPU",
                       "engine": args.engine, "restart_interval": args.restart_interval},
            "bit_exact_vs_reference": bit_exact,
            "encode_mpix_s": round(frames_n * pixels / 1e6 / (np.mean(enc_ms) * 1e-3), 2),
            "decode_mpix_s": round(frames_n * pixels / 1e6 / (np.mean(dec_ms) * 1e-3), 2),
            "encode_ms": round(float(np.mean(enc_ms)), 3),
            "decode_ms": round(float(np.mean(dec_ms)), 3),
            "encode_stage_ms": dict(zip(["analyze", "partition", "chains", "pack", "stuff"],
                                        [round(float(v), 3) for v in np.mean([k[2:7] for k in enc_kernel_ms], axis=0)]))
            if enc_kernel_ms and len(enc_kernel_ms[0]) >= 7 else None,
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel_ms_per_launch": round(dom_ms, 3), "algorithmic_bytes_per_launch": int(alg_bytes)},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)

    if world > 1 or force_gather:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
