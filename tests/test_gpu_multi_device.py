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
