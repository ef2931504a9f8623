"""Throughput of the other BASELINE.json configurations through the batch API (not bench lines: context for DESIGN.md).
Run on the GPU box: python tools/measure_configs.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from charls_amd import batch, capi, synth  # noqa: E402

lib = capi.load_product()
dev = torch.device("cuda:0")
# the encoder's work areas: as much as the batches below need to be coded in ONE pass where they fit (the caller's decision,
# charls_amd_set_workspace_limit; the library's default is a quarter of the device)
batch.set_workspace_limit(int(os.environ.get("CHARLS_AMD_MEASURE_WORKSPACE_GIB", "160")) << 30, lib)


def run(name, frames, *, bits, comps=1, ilv=0, near=0, xform=0, restart=0, slot_factor=None):
    torch.cuda.synchronize()
    kw = dict(bits_per_sample=bits, component_count=comps, interleave_mode=ilv, near_lossless=near,
              color_transformation=xform, restart_interval=restart, lib=lib)
    if slot_factor:  # data that does not compress needs more room than charls_jpegls_encoder_get_estimated_destination_size gives it
        kw["streams"] = torch.empty((frames.shape[0], int(frames[0].numel() * frames.element_size() * slot_factor) & ~255), dtype=torch.uint8, device=dev)
    batch.encode_batch(frames, **kw)  # warm-up with the whole batch: the work arena is sized by the call that needs it
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc = batch.encode_batch(frames, **kw)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = torch.empty_like(frames)
    _, errcs, _ = batch.decode_batch(enc.streams, enc.sizes, out, lib=lib)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    assert (enc.errcs == 0).all() and (errcs == 0).all()
    if near:  # (16-bit samples live in int16 tensors: compare them as the unsigned values they are)
        full = (1 << (8 * frames.element_size())) - 1
        diff = ((out[:8].to(torch.int32) & full) - (frames[:8].to(torch.int32) & full)).abs().max().item()
    else:
        diff = int(not torch.equal(out, frames))
    assert diff <= near, diff
    pix = frames[0].numel() // comps * frames.shape[0] / 1e6
    ratio = frames[0].numel() * frames.element_size() / float(np.mean(enc.sizes))
    print(f"{name}: {frames.shape[0]} frames, encode {pix / (t1 - t0):.0f} MPix/s, decode {pix / (t2 - t1):.0f} MPix/s, "
          f"round trip {pix / (t2 - t0):.0f} MPix/s, compression {ratio:.2f}", flush=True)
    del out, enc
    torch.cuda.empty_cache()


import argparse  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="2,3,5a,5b", help="which configurations to measure")
ap.add_argument("--frames16", type=int, default=1024)
ap.add_argument("--rgb-frames", type=int, default=256)
ap.add_argument("--lib", default=None, help="another build of the product library (A/B runs)")
args = ap.parse_args()
only = set(args.only.split(","))
if args.lib:
    capi.PRODUCT_LIB = os.path.abspath(args.lib)


def rgb_frames(count, size, seed0, bits=8):
    """(count, size, size, 3) uint8 / int16: synth.frame_numpy's interleaved layout (plane k of frame f uses seed seed0 + f + 7919 k)."""
    out = torch.empty((count, size, size, 3), dtype=torch.uint8 if bits <= 8 else torch.int16, device=dev)
    for k in range(3):
        out[..., k] = synth.frames_torch(count, size, size, seed0=seed0 + 7919 * k, bits=bits, device=dev)
    return out


if "2" in only:
    f = synth.frames_torch(args.frames16, 4096, 4096, seed0=4, bits=16, device=dev)
    run("config 2: 4096x4096 16-bit gray lossless", f, bits=16)
    del f
    torch.cuda.empty_cache()
    batch.release_work_areas(lib)
if "2b" in only:  # SURVEY 8(d) C3': "medical-style" 12-bit samples in 16-bit containers
    f = synth.frames_torch(args.frames16, 4096, 4096, seed0=4, bits=12, device=dev)
    run("config 2 as 12-bit (2b): 4096x4096 12-bit gray lossless", f, bits=12)
    del f
    torch.cuda.empty_cache()
    batch.release_work_areas(lib)
if "noise" in only:  # full-range noise: the chains' recurrences never forget (|Errval| >> N), every job is walked again by ONE lane per chain
    import ctypes as C
    L = lib.lib
    L.charls_amd_speculation_counters.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
    L.charls_amd_speculation_counters.restype = C.c_int32
    before = (C.c_uint64 * 6)()
    L.charls_amd_speculation_counters(before, 6)
    f = synth.frames_torch(64, 4096, 4096, seed0=2, bits=8, kind="noise", device=dev)
    run("full-range noise (the encoder's worst case): 4096x4096 8-bit gray, kind=noise", f, bits=8, slot_factor=1.3)
    after = (C.c_uint64 * 6)()
    L.charls_amd_speculation_counters(after, 6)
    print("  speculation counters of its two encodes (jobs, walked again, run jobs, walked again, rare segments, rare serial): "
          + ", ".join(str(int(a) - int(b)) for a, b in zip(after, before)), flush=True)
    del f
    torch.cuda.empty_cache()
    batch.release_work_areas(lib)
if "3" in only:
    f = synth.frames_torch(256, 2048, 2048, seed0=100, bits=8, device=dev)
    run("config 3: 256 x 2048x2048 8-bit gray lossless (one GPU)", f, bits=8)
    del f
    torch.cuda.empty_cache()
    batch.release_work_areas(lib)
if "5a" in only:
    rgb = rgb_frames(args.rgb_frames, 4096, 5)
    run("config 4 as 5a: 4096x4096 RGB ILV_SAMPLE HP1 lossless", rgb, bits=8, comps=3, ilv=2, xform=1)
    del rgb
    torch.cuda.empty_cache()
    batch.release_work_areas(lib)
if "6" in only:
    f = synth.frames_torch(1024, 4096, 4096, seed0=2, bits=8, device=dev)
    run("near-lossless gray (6): 4096x4096 8-bit gray NEAR=2", f, bits=8, near=2)
    del f
    torch.cuda.empty_cache()
    batch.release_work_areas(lib)
if "5c" in only:
    rgb = rgb_frames(args.rgb_frames, 4096, 5)
    run("config 4 line-interleaved (5c): 4096x4096 RGB ILV_LINE HP1 lossless", rgb, bits=8, comps=3, ilv=1, xform=1)
    del rgb
    torch.cuda.empty_cache()
    batch.release_work_areas(lib)
if "5p" in only:  # planar RGB: the component scans of every frame in one launch, both directions
    planes = torch.empty((args.rgb_frames, 3, 4096, 4096), dtype=torch.uint8, device=dev)
    for k in range(3):
        planes[:, k] = synth.frames_torch(args.rgb_frames, 4096, 4096, seed0=5 + 7919 * k, bits=8, device=dev)
    run("4096x4096 RGB planar (ILV_NONE) lossless (5p)", planes, bits=8, comps=3)
    capi.set_knob("BATCH_ROUNDS", 1)
    run("the same, scan by scan (CHARLS_AMD_BATCH_ROUNDS=1: rounds of one component scan per frame, as until round 4)", planes, bits=8, comps=3)
    capi.set_knob("BATCH_ROUNDS", None)
    del planes
    torch.cuda.empty_cache()
    batch.release_work_areas(lib)
if "5d" in only:
    rgb = rgb_frames(args.rgb_frames, 4096, 5)
    run("config 4 line-interleaved near-lossless (5d): 4096x4096 RGB ILV_LINE NEAR=2", rgb, bits=8, comps=3, ilv=1, near=2)
    del rgb
    torch.cuda.empty_cache()
    batch.release_work_areas(lib)
if "5e" in only:
    rgb = rgb_frames(max(1, args.rgb_frames // 2), 4096, 5, bits=16)
    run("4096x4096 RGB 16-bit ILV_SAMPLE lossless (5e)", rgb, bits=16, comps=3, ilv=2)
    run("4096x4096 RGB 16-bit ILV_SAMPLE NEAR=2 (5f)", rgb, bits=16, comps=3, ilv=2, near=2)
    del rgb
    torch.cuda.empty_cache()
    batch.release_work_areas(lib)
if "6w" in only:
    f = synth.frames_torch(512, 4096, 4096, seed0=4, bits=16, device=dev)
    run("near-lossless 16-bit gray (6w): 4096x4096 16-bit gray NEAR=2", f, bits=16, near=2)
    del f
    torch.cuda.empty_cache()
    batch.release_work_areas(lib)
if "5b" in only:
    rgb = rgb_frames(args.rgb_frames, 4096, 5)
    run("config 4 as 5b: 4096x4096 RGB ILV_SAMPLE NEAR=2", rgb, bits=8, comps=3, ilv=2, near=2)
    run("config 4 as 5b with 16-line restart intervals (extension)", rgb, bits=8, comps=3, ilv=2, near=2, restart=16)
