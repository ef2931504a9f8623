// multi_device.cpp -- charls_amd_encode_batch_devices / charls_amd_decode_batch_devices (charls_amd.h part 2b).
//
// frames are the sharding unit, there is no exchange while coding, and the only collective is the hand-over
// of the finished bitstreams to one device.  One process drives several GPUs: a worker thread per shard binds to the
// shard's device, owns its HIP stream and its work areas (they belong to the calling thread and remember their device),
// and runs the single-device batch call of batch_api.cpp on the shard's frames.  The reference has no counterpart; the
// seam is the same one as everywhere else in this library (`encode_scan` / `decode_scan` per scan,
// src/charls_jpegls_encoder.cpp:285-296, src/charls_jpegls_decoder.cpp:186-189).
//
// Gather (encode only, optional): sizes are exchanged first (they are host values of this process already: the
// all-gather of SURVEY 8e is a prefix sum here), then every non-root shard sends each of its streams -- exactly
// sizes[f] bytes -- to its place in the root's buffer.  Transport:
//   * RCCL (ncclCommInitAll over the shards' devices, one grouped ncclSend / ncclRecv pair per frame over xGMI); the
//     library is opened at run time (librccl.so), so that hosts without it can still load this library;
//   * peer copies (hipMemcpyPeerAsync), used when RCCL is not there or the caller asks for them.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "../device/runtime.h"
#include "common.h"

using namespace jls;
using dev::hip_check;

namespace {

// ---- the few RCCL entry points the gather needs (rccl.h: ncclResult_t = int, ncclUint8 = 1)
struct Rccl
{
    using Comm = void*;
    int (*comm_init_all)(Comm*, int, const int*) = nullptr;
    int (*comm_destroy)(Comm) = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    int (*send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    bool ready = false;
};

const Rccl& rccl()
{
    static const Rccl api = [] {
        Rccl r;
        void* h = nullptr;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"})
        {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h != nullptr)
                break;
        }
        if (h == nullptr)
            return r;
        r.comm_init_all = reinterpret_cast<decltype(r.comm_init_all)>(dlsym(h, "ncclCommInitAll"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
        r.group_start = reinterpret_cast<decltype(r.group_start)>(dlsym(h, "ncclGroupStart"));
        r.group_end = reinterpret_cast<decltype(r.group_end)>(dlsym(h, "ncclGroupEnd"));
        r.send = reinterpret_cast<decltype(r.send)>(dlsym(h, "ncclSend"));
        r.recv = reinterpret_cast<decltype(r.recv)>(dlsym(h, "ncclRecv"));
        r.ready = r.comm_init_all && r.comm_destroy && r.group_start && r.group_end && r.send && r.recv;
        return r;
    }();
    return api;
}

constexpr int kNcclUint8 = 1;

void check_shards(uint32_t shard_count, const charls_amd_device_shard* shards)
{
    check_argument(shard_count >= 1 && shard_count <= 64);
    check_pointer(shards);
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess)
        raise(CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE);
    for (uint32_t s = 0; s < shard_count; ++s)
    {
        check_argument(shards[s].device >= 0 && shards[s].device < devices);
        if (shards[s].frame_count != 0)
        {
            check_pointer(shards[s].d_frames);
            check_pointer(shards[s].d_streams);
        }
    }
}

// Runs `work(shard)` for every shard on a thread of its own, bound to the shard's device; the first error wins.
template <typename Work>
charls_jpegls_errc on_every_shard(uint32_t shard_count, const charls_amd_device_shard* shards, Work work)
{
    std::atomic<int32_t> first_error{CHARLS_JPEGLS_ERRC_SUCCESS};
    auto body = [&](uint32_t s) {
        charls_jpegls_errc e = CHARLS_JPEGLS_ERRC_SUCCESS;
        if (hipSetDevice(shards[s].device) != hipSuccess)
            e = CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE;
        else
            e = work(s);
        int32_t expected = CHARLS_JPEGLS_ERRC_SUCCESS;
        if (e != CHARLS_JPEGLS_ERRC_SUCCESS)
            first_error.compare_exchange_strong(expected, static_cast<int32_t>(e));
        // the work areas of this thread die with it: give the memory back while the device is still current
        (void)charls_amd_release_work_areas();
    };
    std::vector<std::thread> workers;
    workers.reserve(shard_count);
    for (uint32_t s = 0; s < shard_count; ++s)
        workers.emplace_back(body, s);
    for (std::thread& t : workers)
        t.join();
    return static_cast<charls_jpegls_errc>(first_error.load());
}

struct StreamGuard
{
    hipStream_t s{};
    StreamGuard() { hip_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~StreamGuard() { (void)hipStreamDestroy(s); }
    StreamGuard(const StreamGuard&) = delete;
    StreamGuard& operator=(const StreamGuard&) = delete;
};

} // namespace

extern "C" charls_jpegls_errc charls_amd_encode_batch_devices(const charls_amd_codec_params* params, uint32_t shard_count,
                                                              const charls_amd_device_shard* shards, size_t frame_pitch_bytes,
                                                              uint32_t stride, size_t stream_pitch_bytes, uint64_t* sizes,
                                                              charls_jpegls_errc* errcs, const charls_amd_gather* gather)
try
{
    check_pointer(params);
    check_pointer(sizes);
    check_pointer(errcs);
    dev::require_device();
    check_shards(shard_count, shards);
    std::vector<uint64_t> first(shard_count + 1, 0);
    for (uint32_t s = 0; s < shard_count; ++s)
        first[s + 1] = first[s] + shards[s].frame_count;
    int caller_device = 0;
    (void)hipGetDevice(&caller_device);

    const charls_jpegls_errc coded = on_every_shard(shard_count, shards, [&](uint32_t s) -> charls_jpegls_errc {
        const charls_amd_device_shard& sh = shards[s];
        if (sh.frame_count == 0)
            return CHARLS_JPEGLS_ERRC_SUCCESS;
        hipStream_t stream = static_cast<hipStream_t>(sh.hip_stream);
        return charls_amd_encode_batch_device(params, sh.frame_count, sh.d_frames, frame_pitch_bytes, stride, sh.d_streams,
                                              stream_pitch_bytes, sizes + first[s], errcs + first[s], stream);
    });
    (void)hipSetDevice(caller_device);
    if (coded != CHARLS_JPEGLS_ERRC_SUCCESS || gather == nullptr)
        return coded;

    // ---- hand-over of the bitstreams to the root shard's device
    check_argument(gather->root_shard < shard_count);
    check_pointer(gather->d_gathered);
    check_pointer(gather->offsets);
    const uint64_t total_frames = first[shard_count];
    uint64_t at = 0;
    for (uint64_t f = 0; f < total_frames; ++f)
    { // (the "all-gather of sizes": every shard's sizes are host values of this process; frames that failed take no room)
        gather->offsets[f] = at;
        at += errcs[f] == CHARLS_JPEGLS_ERRC_SUCCESS ? sizes[f] : 0;
    }
    if (at > gather->capacity_bytes)
        raise(CHARLS_JPEGLS_ERRC_DESTINATION_TOO_SMALL);
    const uint32_t root = gather->root_shard;
    const int root_device = shards[root].device;
    auto* gathered = static_cast<uint8_t*>(gather->d_gathered);
    bool distinct_devices = true;
    for (uint32_t a = 0; a < shard_count; ++a)
        for (uint32_t b = a + 1; b < shard_count; ++b)
            distinct_devices = distinct_devices && shards[a].device != shards[b].device;
    // (with RCCL asked for by name the communicator is set up even for a single shard: that is what a one-GPU box can test)
    const bool use_rccl = gather->transport != CHARLS_AMD_TRANSPORT_PEER_COPIES && rccl().ready && distinct_devices &&
                          (shard_count > 1 || gather->transport == CHARLS_AMD_TRANSPORT_RCCL);
    if (gather->transport == CHARLS_AMD_TRANSPORT_RCCL && !use_rccl)
        raise(CHARLS_AMD_ERRC_DEVICE_FAILURE); // RCCL was asked for and is not usable here

    // the root's own streams: copies on its device
    hip_check(hipSetDevice(root_device));
    {
        StreamGuard own;
        for (uint64_t f = first[root]; f < first[root + 1]; ++f)
            if (errcs[f] == CHARLS_JPEGLS_ERRC_SUCCESS && sizes[f] != 0)
                hip_check(hipMemcpyAsync(gathered + gather->offsets[f],
                                         static_cast<const uint8_t*>(shards[root].d_streams) + (f - first[root]) * stream_pitch_bytes,
                                         sizes[f], hipMemcpyDeviceToDevice, own.s));
        if (use_rccl)
        {
            const Rccl& api = rccl();
            std::vector<int> devices(shard_count);
            for (uint32_t s = 0; s < shard_count; ++s)
                devices[s] = shards[s].device;
            std::vector<Rccl::Comm> comms(shard_count, nullptr);
            if (api.comm_init_all(comms.data(), static_cast<int>(shard_count), devices.data()) != 0)
                raise(CHARLS_AMD_ERRC_DEVICE_FAILURE);
            std::vector<hipStream_t> streams(shard_count, nullptr);
            bool failed = false;
            for (uint32_t s = 0; s < shard_count && !failed; ++s)
            {
                failed = hipSetDevice(devices[s]) != hipSuccess || hipStreamCreateWithFlags(&streams[s], hipStreamNonBlocking) != hipSuccess;
            }
            // one grouped send / receive pair per frame, in rounds of 64 frames per peer (point-to-point over xGMI)
            constexpr uint64_t kRound = 64;
            for (uint64_t r0 = 0; !failed; r0 += kRound)
            {
                bool any = false;
                failed = failed || api.group_start() != 0;
                for (uint32_t s = 0; s < shard_count && !failed; ++s)
                {
                    if (s == root)
                        continue;
                    const uint64_t lo = first[s] + r0, hi = std::min<uint64_t>(first[s + 1], lo + kRound);
                    for (uint64_t f = lo; f < hi && !failed; ++f)
                    {
                        any = true;
                        if (errcs[f] != CHARLS_JPEGLS_ERRC_SUCCESS || sizes[f] == 0)
                            continue;
                        (void)hipSetDevice(devices[s]);
                        failed = failed || api.send(static_cast<const uint8_t*>(shards[s].d_streams) + (f - first[s]) * stream_pitch_bytes,
                                                    sizes[f], kNcclUint8, static_cast<int>(root), comms[s], streams[s]) != 0;
                        (void)hipSetDevice(root_device);
                        failed = failed || api.recv(gathered + gather->offsets[f], sizes[f], kNcclUint8, static_cast<int>(s), comms[root],
                                                    streams[root]) != 0;
                    }
                }
                failed = api.group_end() != 0 || failed;
                if (!any)
                    break;
            }
            for (uint32_t s = 0; s < shard_count; ++s)
            {
                if (streams[s] != nullptr)
                {
                    (void)hipSetDevice(devices[s]);
                    failed = hipStreamSynchronize(streams[s]) != hipSuccess || failed;
                    (void)hipStreamDestroy(streams[s]);
                }
                if (comms[s] != nullptr)
                    (void)api.comm_destroy(comms[s]);
            }
            (void)hipSetDevice(root_device);
            if (failed)
                raise(CHARLS_AMD_ERRC_DEVICE_FAILURE);
        }
        else
        {
            for (uint32_t s = 0; s < shard_count; ++s)
            {
                if (s == root)
                    continue;
                for (uint64_t f = first[s]; f < first[s + 1]; ++f)
                    if (errcs[f] == CHARLS_JPEGLS_ERRC_SUCCESS && sizes[f] != 0)
                        hip_check(hipMemcpyPeerAsync(gathered + gather->offsets[f], root_device,
                                                     static_cast<const uint8_t*>(shards[s].d_streams) + (f - first[s]) * stream_pitch_bytes,
                                                     shards[s].device, sizes[f], own.s));
            }
        }
        hip_check(hipStreamSynchronize(own.s));
    }
    (void)hipSetDevice(caller_device);
    if (gather->total_bytes != nullptr)
        *gather->total_bytes = at;
    return CHARLS_JPEGLS_ERRC_SUCCESS;
}
catch (...)
{
    return current_exception_to_errc();
}

extern "C" charls_jpegls_errc charls_amd_decode_batch_devices(uint32_t shard_count, const charls_amd_device_shard* shards,
                                                              size_t stream_pitch_bytes, const uint64_t* sizes,
                                                              size_t frame_pitch_bytes, uint32_t stride,
                                                              charls_amd_codec_params* params_out, charls_jpegls_errc* errcs)
try
{
    check_pointer(sizes);
    check_pointer(errcs);
    dev::require_device();
    check_shards(shard_count, shards);
    std::vector<uint64_t> first(shard_count + 1, 0);
    for (uint32_t s = 0; s < shard_count; ++s)
        first[s + 1] = first[s] + shards[s].frame_count;
    int caller_device = 0;
    (void)hipGetDevice(&caller_device);
    std::vector<charls_amd_codec_params> found(shard_count);
    const charls_jpegls_errc coded = on_every_shard(shard_count, shards, [&](uint32_t s) -> charls_jpegls_errc {
        const charls_amd_device_shard& sh = shards[s];
        if (sh.frame_count == 0)
            return CHARLS_JPEGLS_ERRC_SUCCESS;
        // (in a decode shard d_streams is the source and d_frames the destination)
        return charls_amd_decode_batch_device(sh.frame_count, sh.d_streams, stream_pitch_bytes, sizes + first[s],
                                              const_cast<void*>(sh.d_frames), frame_pitch_bytes, stride, &found[s], errcs + first[s],
                                              static_cast<hipStream_t>(sh.hip_stream));
    });
    (void)hipSetDevice(caller_device);
    if (params_out != nullptr)
        for (uint32_t s = 0; s < shard_count; ++s)
            if (shards[s].frame_count != 0)
            {
                *params_out = found[s];
                break;
            }
    return coded;
}
catch (...)
{
    return current_exception_to_errc();
}
