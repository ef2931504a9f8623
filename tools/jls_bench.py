#!/usr/bin/env python3
"""jls_bench -- the reference's `charls-cli benchmark-encode / benchmark-decode` (cli/benchmark.cpp:32-90) for this engine,
parameterised over BASELINE.json's configurations with the batch size as an argument.

Methodology as in the reference's tool: a loop of `--loop` iterations around the PUBLIC calls, codec handle created inside
the loop, destination allocated outside it, total and per-image time printed.  Two ways in:
  --abi host    one frame per call through the host-pointer C ABI (charls_jpegls_encoder_* / charls_jpegls_decoder_*),
                host buffer in -> host buffer out (PCIe inclusive) -- literally what the reference's tool times;
  --abi batch   `--frames` device-resident frames per call through charls_amd_encode_batch_device /
                charls_amd_decode_batch_device (how the engine is meant to be fed: frames are its unit of parallelism).
  --library PATH runs the host-ABI loop on any CharLS-ABI library instead (e.g. oracle/_ref/libcharls_ref.so, on the CPU).

Examples (on the GPU box):
  python tools/jls_bench.py --config 1 --abi host --loop 5
  python tools/jls_bench.py --config 3 --abi batch --frames 256 --loop 3
  python tools/jls_bench.py --config 4a --abi batch --frames 64
  python tools/jls_bench.py --config 1 --abi host --library oracle/_ref/libcharls_ref.so
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from charls_amd import capi, synth  # noqa: E402

# BASELINE.json configs (index as there); 4a / 4b are the measurable variants of configs[4], whose literal combination
# (HP1 + NEAR 2) the reference rejects
CONFIGS = {
    "0": dict(width=512, height=512, bits=8, comps=1, ilv=0, near=0, xform=0, seed=1, what="512x512 8-bit gray, lossless"),
    "1": dict(width=4096, height=4096, bits=8, comps=1, ilv=0, near=0, xform=0, seed=2, what="4096x4096 8-bit gray, lossless"),
    "2": dict(width=4096, height=4096, bits=16, comps=1, ilv=0, near=0, xform=0, seed=4, what="4096x4096 16-bit gray, lossless"),
    "3": dict(width=2048, height=2048, bits=8, comps=1, ilv=0, near=0, xform=0, seed=100, what="2048x2048 8-bit gray frames, lossless"),
    "4a": dict(width=4096, height=4096, bits=8, comps=3, ilv=2, near=0, xform=1, seed=5, what="4096x4096 RGB ILV_SAMPLE HP1, lossless"),
    "4b": dict(width=4096, height=4096, bits=8, comps=3, ilv=2, near=2, xform=0, seed=5, what="4096x4096 RGB ILV_SAMPLE, NEAR=2"),
}


def report(what, seconds, loops, images_per_loop, mpix_per_image):
    per_image = seconds / (loops * images_per_loop)
    print(f"Total {what} time is: {seconds * 1e3:.2f} ms")
    print(f"{what.capitalize()} time per image: {per_image * 1e3:.3f} ms  ({mpix_per_image / per_image:.1f} MPixels/s)")
    return {"total_ms": round(seconds * 1e3, 3), "ms_per_image": round(per_image * 1e3, 4), "mpix_s": round(mpix_per_image / per_image, 1)}


def host_abi(lib, c, loops):
    img = synth.frame_numpy(c["width"], c["height"], seed=c["seed"], bits=c["bits"], components=c["comps"], interleaved=True)
    kw = dict(width=c["width"], height=c["height"], bits_per_sample=c["bits"], component_count=c["comps"],
              interleave_mode=c["ilv"], near_lossless=c["near"], color_transformation=c["xform"])
    mpix = c["width"] * c["height"] / 1e6
    jls = lib.encode(img, **kw)  # warm-up: device context, allocations
    lib.decode(jls)
    print(f"Test encode performance with loop count {loops} and {c['what']} ({len(jls)} bytes encoded)")
    t0 = time.perf_counter()
    for _ in range(loops):
        jls = lib.encode(img, **kw)  # creates and destroys its own handle: the handle is inside the loop
    enc = report("encoding", time.perf_counter() - t0, loops, 1, mpix)
    print(f"Test decode performance with loop count {loops}")
    t0 = time.perf_counter()
    for _ in range(loops):
        _, px = lib.decode(jls)
    dec = report("decoding", time.perf_counter() - t0, loops, 1, mpix)
    worst = int(np.abs(np.frombuffer(px.tobytes(), dtype=img.dtype).astype(np.int64) - img.reshape(-1).astype(np.int64)).max())
    assert worst <= c["near"], worst
    return {"encode": enc, "decode": dec, "encoded_bytes": len(jls)}


def batch_abi(lib, c, loops, frames_n):
    import torch
    from charls_amd import batch
    dev = torch.device("cuda:0")
    if c["comps"] == 1:
        frames = synth.frames_torch(frames_n, c["width"], c["height"], seed0=c["seed"], bits=c["bits"], device=dev)
    else:
        frames = torch.empty((frames_n, c["height"], c["width"], c["comps"]), dtype=torch.uint8, device=dev)
        for k in range(c["comps"]):
            frames[..., k] = synth.frames_torch(frames_n, c["width"], c["height"], seed0=c["seed"] + 7919 * k, bits=c["bits"], device=dev)
    kw = dict(bits_per_sample=c["bits"], component_count=c["comps"], interleave_mode=c["ilv"], near_lossless=c["near"],
              color_transformation=c["xform"], lib=lib)
    mpix = c["width"] * c["height"] / 1e6
    enc = batch.encode_batch(frames, **kw)  # warm-up: work areas
    out = torch.empty_like(frames)
    batch.decode_batch(enc.streams, enc.sizes, out, lib=lib)
    torch.cuda.synchronize()
    print(f"Test encode performance with loop count {loops}, {frames_n} frames per call, {c['what']}")
    t0 = time.perf_counter()
    for _ in range(loops):
        enc = batch.encode_batch(frames, streams=enc.streams, **kw)
    torch.cuda.synchronize()
    e = report("encoding", time.perf_counter() - t0, loops, frames_n, mpix)
    print(f"Test decode performance with loop count {loops}, {frames_n} frames per call")
    t0 = time.perf_counter()
    for _ in range(loops):
        _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out, lib=lib)
    torch.cuda.synchronize()
    d = report("decoding", time.perf_counter() - t0, loops, frames_n, mpix)
    assert (enc.errcs == 0).all() and (errcs == 0).all()
    worst = max(int((out[f0:f0 + 16].int() - frames[f0:f0 + 16].int()).abs().max().item()) for f0 in range(0, frames_n, 16))
    assert worst <= c["near"], worst
    return {"encode": e, "decode": d, "frames": frames_n, "encoded_bytes_per_frame": int(np.mean(enc.sizes))}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", default="1", choices=sorted(CONFIGS))
    ap.add_argument("--abi", default="batch", choices=["host", "batch"])
    ap.add_argument("--frames", type=int, default=256, help="frames per call (--abi batch)")
    ap.add_argument("--loop", type=int, default=3, help="the reference tool's loop count")
    ap.add_argument("--library", default=None, help="any CharLS-ABI shared library for --abi host (default: libcharls_amd.so)")
    ap.add_argument("--json", action="store_true", help="print one JSON line with the numbers as well")
    args = ap.parse_args()
    c = CONFIGS[args.config]
    if args.abi == "batch":
        import torch  # noqa: F401  (before the library: torch brings its own HIP runtime, and the first one loaded serves both)
    lib = capi.CharLSLibrary(os.path.abspath(args.library)) if args.library else capi.load_product()
    if args.abi == "host":
        result = host_abi(lib, c, args.loop)
    else:
        if args.library:
            ap.error("--abi batch is this library's additive API")
        result = batch_abi(lib, c, args.loop, args.frames)
    if args.json:
        print(json.dumps({"config": args.config, "what": c["what"], "abi": args.abi, "loop": args.loop, **result}))


if __name__ == "__main__":
    main()
