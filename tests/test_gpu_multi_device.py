"""charls_amd_encode_batch_devices / charls_amd_decode_batch_devices (include/charls_amd.h part 2b) on the one GPU of the test
box: the worker threads, the per-thread work areas, the shard-ordered results and the gather are exercised with two shards
on device 0 (peer copies) and with one shard through RCCL's communicator set-up.  Several devices are the driver's to run."""
import hashlib

import numpy as np
import pytest

import common
import oracle_bind as ob
from charls_amd import batch, capi, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def lib():
    L = capi.load_product()
    assert L.lib.charls_amd_device_status() == 0
    return L


def _frames(count, w, h, seed0):
    return synth.frames_torch(count, w, h, seed0=seed0, bits=8, kind="mixed", device="cuda")


@pytest.mark.parametrize("split,transport", [((5, 3), batch.TRANSPORT_PEER_COPIES), ((0, 4), batch.TRANSPORT_PEER_COPIES),
                                             ((6,), batch.TRANSPORT_RCCL), ((6,), batch.TRANSPORT_AUTO)])
def test_shards_give_the_frames_of_part_one_and_gather_them_in_order(lib, split, transport):
    w, h = 320, 200
    pitch = (batch.estimated_destination_size(w, h, 8, 1) + 255) & ~255
    frame_shards, stream_shards, want = [], [], []
    for s, n in enumerate(split):
        f = _frames(n, w, h, seed0=10 + 100 * s) if n else torch.empty((0, h, w), dtype=torch.uint8, device="cuda")
        frame_shards.append(f)
        stream_shards.append(torch.zeros((n, pitch), dtype=torch.uint8, device="cuda"))
        for i in range(n):
            want.append(ob.encode(f[i].cpu().numpy(), width=w, height=h))
    total = sum(split)
    root = len(split) - 1
    gathered = torch.zeros(total * pitch, dtype=torch.uint8, device="cuda")
    sizes, errcs, offsets, total_bytes = batch.encode_batch_devices(frame_shards, stream_shards, gather_to=(root, gathered),
                                                                    transport=transport, lib=lib)
    assert (errcs == 0).all()
    assert [int(x) for x in sizes] == [len(x) for x in want]
    assert total_bytes == sum(len(x) for x in want)
    flat = gathered[:total_bytes].cpu().numpy().tobytes()
    assert hashlib.sha256(flat).hexdigest() == hashlib.sha256(b"".join(want)).hexdigest()
    assert [int(o) for o in offsets] == list(np.cumsum([0] + [len(x) for x in want[:-1]]))
    # and back: every shard decodes its own streams
    outs = [torch.zeros_like(f) for f in frame_shards]
    params, derr = batch.decode_batch_devices(stream_shards, sizes, outs, lib=lib)
    assert (derr == 0).all()
    for f, o in zip(frame_shards, outs):
        assert torch.equal(f, o)
    assert params.frame_info.width == w and params.frame_info.height == h


def test_a_gather_buffer_that_is_too_small_and_bad_shards_are_refused(lib):
    w, h = 64, 48
    pitch = (batch.estimated_destination_size(w, h, 8, 1) + 255) & ~255
    f = _frames(2, w, h, seed0=3)
    st = torch.zeros((2, pitch), dtype=torch.uint8, device="cuda")
    small = torch.zeros(100, dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.JpegLSError) as e:
        batch.encode_batch_devices([f], [st], gather_to=(0, small), lib=lib)
    assert e.value.errc == 3
    l = batch._bind(lib)
    shards = (batch.DeviceShard * 1)(batch.DeviceShard(99, 2, f.data_ptr(), st.data_ptr(), None))
    p = batch.CodecParams(capi.FrameInfo(w, h, 8, 1), 0, 0, 0, capi.PcParameters(0, 0, 0, 0, 0), 0, 0)
    sizes = np.zeros(2, dtype=np.uint64)
    errcs = np.zeros(2, dtype=np.int32)
    import ctypes as C
    rc = l.charls_amd_encode_batch_devices(C.byref(p), 1, shards, h * w, 0, pitch, sizes.ctypes.data_as(C.POINTER(C.c_uint64)),
                                           errcs.ctypes.data_as(C.POINTER(C.c_int32)), None)
    assert rc != 0


def _devices_api(lib):
    import ctypes as C
    l = batch._bind(lib)
    l.charls_amd_devices_create.restype = C.c_void_p
    l.charls_amd_devices_destroy.argtypes = [C.c_void_p]
    l.charls_amd_devices_destroy.restype = None
    l.charls_amd_devices_work_area_bytes.argtypes = [C.c_void_p]
    l.charls_amd_devices_work_area_bytes.restype = C.c_uint64
    l.charls_amd_devices_release_work_areas.argtypes = [C.c_void_p]
    l.charls_amd_devices_release_work_areas.restype = C.c_int32
    return l


def test_second_call_on_the_context_allocates_nothing_and_costs_what_one_device_costs(lib):
    """the worker threads, their work areas and the communicator belong to a context that lives
    across calls.  Two consecutive calls: the work areas the workers hold do not change, and the second call takes what the
    single-device call takes (it used to start threads, allocate, free and set RCCL up per call)."""
    import time
    l = _devices_api(lib)
    w = h = 1024
    n = 16
    pitch = (batch.estimated_destination_size(w, h, 8, 1) + 255) & ~255
    f = _frames(n, w, h, seed0=40)
    st = torch.zeros((n, pitch), dtype=torch.uint8, device="cuda")
    gathered = torch.zeros(n * pitch, dtype=torch.uint8, device="cuda")
    l.charls_amd_devices_destroy(None)  # a fresh default context
    assert l.charls_amd_devices_work_area_bytes(None) == 0

    def call():
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = batch.encode_batch_devices([f], [st], gather_to=(0, gathered), transport=batch.TRANSPORT_RCCL, lib=lib)
        return time.perf_counter() - t, out

    t_first, first = call()
    held = l.charls_amd_devices_work_area_bytes(None)
    assert held > 0
    t_second, second = call()
    assert l.charls_amd_devices_work_area_bytes(None) == held
    assert (first[0] == second[0]).all() and (second[1] == 0).all()
    # the single-device call on this thread, same frames (its own work areas: warm it up first)
    batch.encode_batch(f, streams=st, lib=lib)
    torch.cuda.synchronize()
    t = time.perf_counter()
    batch.encode_batch(f, streams=st, lib=lib)
    t_single = time.perf_counter() - t
    gather_s = 0.002  # the exchange itself: 16 device-to-device copies of ~0.4 MB
    assert t_second <= 1.1 * t_single + gather_s, (t_first, t_second, t_single)
    # the context gives its memory back on request, and all of it when it is destroyed
    assert l.charls_amd_devices_release_work_areas(None) == 0
    assert l.charls_amd_devices_work_area_bytes(None) == 0
    l.charls_amd_devices_destroy(None)
    batch.release_work_areas(lib)


def test_explicit_contexts_are_independent(lib):
    l = _devices_api(lib)
    import ctypes as C
    w, h = 320, 200
    pitch = (batch.estimated_destination_size(w, h, 8, 1) + 255) & ~255
    f = _frames(3, w, h, seed0=77)
    st = torch.zeros((3, pitch), dtype=torch.uint8, device="cuda")
    l.charls_amd_devices_encode_batch.argtypes = [C.c_void_p] + list(l.charls_amd_encode_batch_devices.argtypes)
    l.charls_amd_devices_encode_batch.restype = C.c_int32
    ctxs = [l.charls_amd_devices_create() for _ in range(2)]
    assert all(ctxs)
    p = batch.CodecParams(capi.FrameInfo(w, h, 8, 1), 0, 0, 0, capi.PcParameters(0, 0, 0, 0, 0), 0, 0)
    for ctx in ctxs:
        shards = (batch.DeviceShard * 2)(batch.DeviceShard(0, 2, f.data_ptr(), st.data_ptr(), None),
                                         batch.DeviceShard(0, 1, f[2:].data_ptr(), st[2:].data_ptr(), None))
        sizes = np.zeros(3, dtype=np.uint64)
        errcs = np.zeros(3, dtype=np.int32)
        rc = l.charls_amd_devices_encode_batch(ctx, C.byref(p), 2, shards, h * w, 0, pitch, sizes.ctypes.data_as(C.POINTER(C.c_uint64)),
                                               errcs.ctypes.data_as(C.POINTER(C.c_int32)), None)
        assert rc == 0 and (errcs == 0).all()
        got = st.cpu().numpy()
        for i in range(3):
            assert got[i, :int(sizes[i])].tobytes() == ob.encode(f[i].cpu().numpy(), width=w, height=h)
        assert l.charls_amd_devices_work_area_bytes(ctx) > 0
    for ctx in ctxs:
        l.charls_amd_devices_destroy(ctx)


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() < 2, reason="one-GPU box")
def test_one_thread_moves_from_device_to_device(lib):
    """work areas AND the pipeline's side streams follow the thread to its new device."""
    w = h = 512
    for device in (0, 1, 0):
        torch.cuda.set_device(device)
        f = synth.frames_torch(4, w, h, seed0=5, bits=8, kind="mixed", device=f"cuda:{device}")
        enc = batch.encode_batch(f, lib=lib)
        got = enc.streams.cpu().numpy()
        for i in range(4):
            assert got[i, :int(enc.sizes[i])].tobytes() == ob.encode(f[i].cpu().numpy(), width=w, height=h)
    torch.cuda.set_device(0)
