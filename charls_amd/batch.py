"""Device-resident batch encode/decode (charls_amd.h part 2) on torch tensors, plus frame sharding across ranks.

torch supplies HBM allocations, the current HIP stream and torch.distributed (backend "nccl" = RCCL over xGMI);
all coding work happens inside libcharls_amd.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi


class CodecParams(C.Structure):  # charls_amd_codec_params (include/charls_amd.h)
    _fields_ = [("frame_info", capi.FrameInfo), ("near_lossless", C.c_int32), ("interleave_mode", C.c_int32),
                ("color_transformation", C.c_int32), ("preset_coding_parameters", capi.PcParameters),
                ("encoding_options", C.c_uint32), ("restart_interval", C.c_uint32)]


class DeviceShard(C.Structure):  # charls_amd_device_shard
    _fields_ = [("device", C.c_int32), ("frame_count", C.c_uint32), ("d_frames", C.c_void_p), ("d_streams", C.c_void_p),
                ("hip_stream", C.c_void_p)]


class Gather(C.Structure):  # charls_amd_gather
    _fields_ = [("root_shard", C.c_uint32), ("d_gathered", C.c_void_p), ("capacity_bytes", C.c_size_t),
                ("offsets", C.POINTER(C.c_uint64)), ("total_bytes", C.POINTER(C.c_uint64)), ("transport", C.c_int32)]


TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_PEER_COPIES = 0, 1, 2


def _bind(lib):
    l = lib.lib
    if getattr(l, "_batch_bound", False):
        return l
    l.charls_amd_encode_batch_device.argtypes = [C.POINTER(CodecParams), C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32,
                                                 C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_int32),
                                                 C.c_void_p]
    l.charls_amd_encode_batch_device.restype = C.c_int32
    l.charls_amd_decode_batch_device.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_void_p,
                                                 C.c_size_t, C.c_uint32, C.POINTER(CodecParams), C.POINTER(C.c_int32),
                                                 C.c_void_p]
    l.charls_amd_decode_batch_device.restype = C.c_int32
    l.charls_amd_last_timings.argtypes = [C.POINTER(C.c_double), C.c_int32]
    l.charls_amd_last_timings.restype = C.c_int32
    l.charls_amd_set_encode_engine.argtypes = [C.c_int32]
    l.charls_amd_set_encode_engine.restype = C.c_int32
    l.charls_amd_set_workspace_limit.argtypes = [C.c_uint64]
    l.charls_amd_set_workspace_limit.restype = C.c_int32
    l.charls_amd_release_work_areas.argtypes = []
    l.charls_amd_release_work_areas.restype = C.c_int32
    l.charls_amd_work_area_bytes.argtypes = []
    l.charls_amd_work_area_bytes.restype = C.c_uint64
    l.charls_amd_encode_batch_devices.argtypes = [C.POINTER(CodecParams), C.c_uint32, C.POINTER(DeviceShard), C.c_size_t,
                                                  C.c_uint32, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_int32),
                                                  C.POINTER(Gather)]
    l.charls_amd_encode_batch_devices.restype = C.c_int32
    l.charls_amd_decode_batch_devices.argtypes = [C.c_uint32, C.POINTER(DeviceShard), C.c_size_t, C.POINTER(C.c_uint64),
                                                  C.c_size_t, C.c_uint32, C.POINTER(CodecParams), C.POINTER(C.c_int32)]
    l.charls_amd_decode_batch_devices.restype = C.c_int32
    l._batch_bound = True
    return l


def estimated_destination_size(width, height, bits, components) -> int:
    """charls_jpegls_encoder_get_estimated_destination_size (reference src/charls_jpegls_encoder.cpp:103-114)."""
    raw = width * height * components * ((bits + 7) // 8)
    return raw + raw // 16 + 1024 + 34


@dataclass
class EncodedBatch:
    streams: "torch.Tensor"  # (frames, pitch) uint8 on the device; frame f's .jls is streams[f, :sizes[f]]
    sizes: np.ndarray        # uint64, host
    errcs: np.ndarray        # int32, host
    gpu_ms: tuple            # (total, dominant kernel)


def last_timings(lib=None):
    l = _bind(lib or capi.load_product())
    buf = (C.c_double * 8)()
    n = l.charls_amd_last_timings(buf, 8)
    return tuple(buf[i] for i in range(n))


def encode_batch(frames, *, bits_per_sample=8, component_count=1, interleave_mode=0, near_lossless=0,
                 color_transformation=0, preset=(0, 0, 0, 0, 0), encoding_options=0, restart_interval=0, streams=None,
                 lib=None) -> EncodedBatch:
    """frames: contiguous torch tensor on the GPU, (F, H, W) / (F, C, H, W) for ILV_NONE or (F, H, W, C) otherwise,
    dtype uint8 (<= 8 bit) or int16/uint16 (9..16 bit).  restart_interval (lines, 0 = none) is this library's extension:
    the intervals of a frame are coded in parallel and separated by RSTm markers."""
    import torch
    lib = lib or capi.load_product()
    l = _bind(lib)
    assert frames.is_cuda and frames.is_contiguous()
    count = frames.shape[0]
    if component_count == 1 or interleave_mode == 0:
        height, width = frames.shape[-2], frames.shape[-1]
    else:
        height, width = frames.shape[1], frames.shape[2]
    frame_pitch = frames[0].numel() * frames.element_size()
    if streams is None:
        pitch = (estimated_destination_size(width, height, bits_per_sample, component_count) + 255) & ~255
        streams = torch.empty((count, pitch), dtype=torch.uint8, device=frames.device)
    assert streams.is_contiguous() and streams.shape[0] == count
    p = CodecParams(capi.FrameInfo(width, height, bits_per_sample, component_count), near_lossless, interleave_mode,
                    color_transformation, capi.PcParameters(*preset), encoding_options, restart_interval)
    sizes = np.zeros(count, dtype=np.uint64)
    errcs = np.zeros(count, dtype=np.int32)
    stream = torch.cuda.current_stream(frames.device).cuda_stream
    rc = l.charls_amd_encode_batch_device(C.byref(p), count, frames.data_ptr(), frame_pitch, 0, streams.data_ptr(),
                                          streams.shape[1], sizes.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          errcs.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(stream))
    if rc != 0:
        raise capi.JpegLSError(rc, "charls_amd_encode_batch_device")
    return EncodedBatch(streams, sizes, errcs, last_timings(lib))


def decode_batch(streams, sizes, out, *, lib=None):
    """streams: (F, pitch) uint8 device tensor; sizes: host uint64 array; out: preallocated device tensor whose [f] slice
    receives frame f in the reference's user layout. Returns (params, errcs, gpu_ms)."""
    import torch
    lib = lib or capi.load_product()
    l = _bind(lib)
    assert streams.is_cuda and streams.is_contiguous() and out.is_cuda and out.is_contiguous()
    count = streams.shape[0]
    sizes = np.ascontiguousarray(sizes, dtype=np.uint64)
    errcs = np.zeros(count, dtype=np.int32)
    p = CodecParams()
    frame_pitch = out[0].numel() * out.element_size()
    stream = torch.cuda.current_stream(streams.device).cuda_stream
    rc = l.charls_amd_decode_batch_device(count, streams.data_ptr(), streams.shape[1],
                                          sizes.ctypes.data_as(C.POINTER(C.c_uint64)), out.data_ptr(), frame_pitch, 0,
                                          C.byref(p), errcs.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(stream))
    if rc != 0:
        raise capi.JpegLSError(rc, "charls_amd_decode_batch_device")
    return p, errcs, last_timings(lib)


def set_workspace_limit(nbytes: int, lib=None):
    """HBM the library may keep for its work areas (process-wide; 0 = a quarter of the device)."""
    _bind(lib or capi.load_product()).charls_amd_set_workspace_limit(int(nbytes))


def release_work_areas(lib=None):
    """Frees the calling thread's work areas (they are re-allocated on demand)."""
    _bind(lib or capi.load_product()).charls_amd_release_work_areas()


def work_area_bytes(lib=None) -> int:
    return int(_bind(lib or capi.load_product()).charls_amd_work_area_bytes())


def set_encode_engine(engine: int, lib=None):
    """0 automatic, 1 one-wavefront-per-scan kernel, 2 parallel lossless pipeline."""
    rc = _bind(lib or capi.load_product()).charls_amd_set_encode_engine(engine)
    if rc:
        raise capi.JpegLSError(rc, "charls_amd_set_encode_engine")


# ---- multi-GPU: frames are the sharding unit; the only exchange is the final bitstream gather -------------

def encode_batch_devices(frame_shards, stream_shards, *, bits_per_sample=8, gather_to=None, transport=TRANSPORT_AUTO, lib=None):
    """One process, several GPUs (charls_amd_encode_batch_devices): frame_shards[s] / stream_shards[s] are contiguous device
    tensors of shard s on ITS device -- (F_s, H, W) single-component frames and (F_s, pitch) uint8 slots, same H, W and
    pitch everywhere.  gather_to = (root shard index, uint8 device tensor on the root's device): the streams of all shards
    are brought together there, back to back in frame order.  Returns (sizes, errcs, offsets or None, total or None)."""
    lib = lib or capi.load_product()
    l = _bind(lib)
    n = len(frame_shards)
    height, width = frame_shards[0].shape[-2], frame_shards[0].shape[-1]
    frame_pitch = int(np.prod(frame_shards[0].shape[1:])) * frame_shards[0].element_size()
    pitch = stream_shards[0].shape[1]
    shards = (DeviceShard * n)()
    total = 0
    for s in range(n):
        f, st = frame_shards[s], stream_shards[s]
        assert f.is_cuda and f.is_contiguous() and st.is_contiguous() and st.shape[1] == pitch and st.device == f.device
        shards[s] = DeviceShard(f.device.index or 0, f.shape[0], f.data_ptr(), st.data_ptr(), None)
        total += f.shape[0]
    p = CodecParams(capi.FrameInfo(width, height, bits_per_sample, 1), 0, 0, 0, capi.PcParameters(0, 0, 0, 0, 0), 0, 0)
    sizes = np.zeros(total, dtype=np.uint64)
    errcs = np.zeros(total, dtype=np.int32)
    offsets = total_bytes = None
    g = None
    if gather_to is not None:
        root, buf = gather_to
        offsets = np.zeros(total, dtype=np.uint64)
        total_bytes = C.c_uint64(0)
        g = Gather(root, buf.data_ptr(), buf.numel(), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), C.pointer(total_bytes), transport)
    rc = l.charls_amd_encode_batch_devices(C.byref(p), n, shards, frame_pitch, 0, pitch, sizes.ctypes.data_as(C.POINTER(C.c_uint64)),
                                           errcs.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(g) if g is not None else None)
    if rc != 0:
        raise capi.JpegLSError(rc, "charls_amd_encode_batch_devices")
    return sizes, errcs, offsets, (int(total_bytes.value) if total_bytes is not None else None)


def decode_batch_devices(stream_shards, sizes, out_shards, *, lib=None):
    """charls_amd_decode_batch_devices: shard s decodes stream_shards[s] (F_s, pitch) into out_shards[s] on its device."""
    lib = lib or capi.load_product()
    l = _bind(lib)
    n = len(stream_shards)
    shards = (DeviceShard * n)()
    total = 0
    for s in range(n):
        st, o = stream_shards[s], out_shards[s]
        shards[s] = DeviceShard(st.device.index or 0, st.shape[0], o.data_ptr(), st.data_ptr(), None)
        total += st.shape[0]
    sizes = np.ascontiguousarray(sizes, dtype=np.uint64)
    errcs = np.zeros(total, dtype=np.int32)
    p = CodecParams()
    frame_pitch = int(np.prod(out_shards[0].shape[1:])) * out_shards[0].element_size()
    rc = l.charls_amd_decode_batch_devices(n, shards, stream_shards[0].shape[1], sizes.ctypes.data_as(C.POINTER(C.c_uint64)),
                                           frame_pitch, 0, C.byref(p), errcs.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise capi.JpegLSError(rc, "charls_amd_decode_batch_devices")
    return p, errcs


def shard_range(total: int, rank: int, world: int):
    """Contiguous block of frame indices owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_streams(streams, sizes, dst=0, group=None, chunk_frames=32, sink=None):
    """Variable-length gather of the encoded frames to rank `dst` (RCCL on GPU tensors, gloo on CPU tensors): an all-gather
    of the per-frame byte counts, then the payload point to point in rounds of `chunk_frames` frames per rank -- every
    frame is ONE send of exactly its bytes (`streams[f, :sizes[f]]` is contiguous, nothing is packed, padded or copied on
    the sending side), the sends / receives of a round are posted together (dist.batch_isend_irecv = one ncclGroup), and
    rank dst receives into a buffer per source rank that is re-used from round to round (the payload of a big batch does
    not have to fit rank dst's HBM at once).  `sink(rank, first_frame, tensor, sizes)` is called on dst for every received
    piece (its own frames are handed over as views, they never move); without a sink the pieces are kept and returned.
    Returns on dst: (list of per-rank lists of (first_frame, uint8 tensor (n, >= max size of the piece)), per-rank size
    arrays); elsewhere: (None, per-rank size arrays)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = streams.device
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(sizes)], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    max_count = max(counts)
    mine = torch.zeros(max(max_count, 1), dtype=torch.int64, device=dev)
    mine[:len(sizes)] = torch.as_tensor(np.asarray(sizes, dtype=np.int64), device=dev)
    all_sizes = [torch.zeros(max(max_count, 1), dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, mine, group=group)
    all_sizes = [s[:c].cpu().numpy().astype(np.uint64) for s, c in zip(all_sizes, counts)]
    kept = [[] for _ in range(world)] if rank == dst else None
    for first in range(0, max_count, chunk_frames):
        if rank != dst:
            ops = [dist.P2POp(dist.isend, streams[f, :int(sizes[f])], dst, group)
                   for f in range(first, min(first + chunk_frames, len(sizes))) if int(sizes[f]) > 0]
            for req in (dist.batch_isend_irecv(ops) if ops else []):
                req.wait()
            continue
        ops, pieces = [], []
        for r in range(world):
            m = max(0, min(chunk_frames, counts[r] - first))
            if m == 0:
                continue
            sz = all_sizes[r][first:first + m]
            if r == rank:
                pieces.append((r, streams[first:first + m], sz))
                continue
            longest = -(-max(int(sz.max()), 1) // 65536) * 65536  # few distinct buffer sizes -> re-used blocks
            part = torch.empty((m, longest), dtype=torch.uint8, device=dev)
            ops += [dist.P2POp(dist.irecv, part[f, :int(sz[f])], r, group) for f in range(m) if int(sz[f]) > 0]
            pieces.append((r, part, sz))
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        for r, part, sz in pieces:
            if sink is not None:
                sink(r, first, part, sz)
            else:
                kept[r].append((first, part))
    return kept, all_sizes
