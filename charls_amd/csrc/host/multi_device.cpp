// multi_device.cpp -- several GPUs from one process (charls_amd.h part 2b): charls_amd_devices_* and the two
// charls_amd_*_batch_devices entry points that run on the process-wide default context.
//
// frames are the sharding unit, there is no exchange while coding, and the only collective is the hand-over
// of the finished bitstreams to one device.  One process drives several GPUs through a CONTEXT (charls_amd_devices) that
// lives across calls and owns
//   * a worker thread per (device, shard ordinal on that device), bound to its device for life: the encoder's work areas
//     belong to the thread that made them (runtime.hip), so a worker that stays keeps its areas -- the second call
//     allocates nothing (giving ~90 GB back to the driver and asking for it again took seconds per call when every call
//     started its own threads);
//   * the RCCL communicator over the devices of the last gathered call and a stream per device for the exchange
//     (ncclCommInitAll costs hundreds of milliseconds: it runs when the device list changes, not per call).
// The reference has no counterpart; the seam is the same one as everywhere else in this library (`encode_scan` /
// `decode_scan` per scan, src/charls_jpegls_encoder.cpp:285-296, src/charls_jpegls_decoder.cpp:186-189).
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
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
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

// One line on stderr, once per process and message: a transport that silently degrades is a performance bug nobody finds.
void note_once(const char* message)
{
    static std::mutex guard;
    static std::vector<std::string> said;
    std::lock_guard<std::mutex> lock(guard);
    for (const std::string& m : said)
        if (m == message)
            return;
    said.emplace_back(message);
    std::fprintf(stderr, "%s\n", message);
}

// hipMemcpyPeerAsync between two devices goes over xGMI only when each may address the other's memory; otherwise the runtime
// stages the copy through host memory.  Enabled once per pair and direction; where the devices cannot reach each other the
// copies still work (through the host), and that is said once.
void enable_peer_access(int a, int b)
{
    static std::mutex guard;
    static std::vector<std::pair<int, int>> done;
    std::lock_guard<std::mutex> lock(guard);
    int current = 0;
    (void)hipGetDevice(&current);
    for (const auto& [from, to] : {std::pair<int, int>{a, b}, std::pair<int, int>{b, a}})
    {
        bool seen = false;
        for (const auto& d : done)
            seen = seen || d == std::pair<int, int>{from, to};
        if (seen)
            continue;
        done.emplace_back(from, to);
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || can == 0)
        {
            (void)hipGetLastError();
            char text[160];
            std::snprintf(text, sizeof text, "charls_amd: device %d cannot address device %d: peer copies between them go through host memory", from, to);
            note_once(text);
            continue;
        }
        if (hipSetDevice(from) == hipSuccess)
        {
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            {
                char text[160];
                std::snprintf(text, sizeof text, "charls_amd: hipDeviceEnablePeerAccess(%d -> %d) failed (%d): peer copies go through host memory", from, to,
                              static_cast<int>(e));
                note_once(text);
            }
            (void)hipGetLastError();
        }
    }
    (void)hipSetDevice(current);
}

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

// A thread bound to one device for as long as the context lives.  Its thread-local work areas (the encoder's arena, the
// interval buffers) stay allocated between tasks.
class Worker
{
public:
    explicit Worker(int device) : device_(device), thread_([this] { loop(); }) {}
    Worker(const Worker&) = delete;
    Worker& operator=(const Worker&) = delete;
    ~Worker()
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        cv_.notify_all();
        if (thread_.joinable())
            thread_.join();
    }
    void submit(std::function<charls_jpegls_errc()> task)
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            task_ = std::move(task);
            pending_ = true;
            done_ = false;
        }
        cv_.notify_all();
    }
    charls_jpegls_errc wait()
    {
        std::unique_lock<std::mutex> lock(m_);
        cv_.wait(lock, [this] { return done_; });
        return result_;
    }

private:
    void loop()
    {
        const bool bound = hipSetDevice(device_) == hipSuccess;
        for (;;)
        {
            std::function<charls_jpegls_errc()> task;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [this] { return pending_ || stop_; });
                if (!pending_)
                    break; // (stop)
                task = std::move(task_);
                pending_ = false;
            }
            charls_jpegls_errc r = CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE;
            if (bound)
            {
                try
                {
                    r = task();
                }
                catch (...)
                {
                    r = current_exception_to_errc();
                }
            }
            {
                std::lock_guard<std::mutex> lock(m_);
                result_ = r;
                done_ = true;
            }
            cv_.notify_all();
        }
        // this thread's work areas die with it: give the memory back while the device is still current (the thread's OWN
        // areas -- not the process-wide pool of the host-pointer handles)
        if (bound)
            dev::release_work_areas();
    }

    int device_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<charls_jpegls_errc()> task_;
    bool pending_ = false, done_ = true, stop_ = false;
    charls_jpegls_errc result_ = CHARLS_JPEGLS_ERRC_SUCCESS;
    std::thread thread_; // (last: it starts in the constructor)
};

} // namespace

// The context: workers and the exchange's communicator, kept across calls.  One call at a time per context.
struct charls_amd_devices
{
    std::mutex call;
    std::map<std::pair<int, uint32_t>, std::unique_ptr<Worker>> workers; // (device, ordinal among the shards of that device)
    std::vector<int> comm_devices; // devices of the cached communicator, in shard order
    std::vector<Rccl::Comm> comms;
    std::vector<hipStream_t> comm_streams;

    Worker& worker(int device, uint32_t ordinal)
    {
        std::unique_ptr<Worker>& w = workers[{device, ordinal}];
        if (!w)
            w = std::make_unique<Worker>(device);
        return *w;
    }

    void drop_communicator() noexcept
    {
        int current = 0;
        (void)hipGetDevice(&current);
        for (size_t s = 0; s < comm_devices.size(); ++s)
        {
            if (s < comm_streams.size() && comm_streams[s] != nullptr)
            {
                (void)hipSetDevice(comm_devices[s]);
                (void)hipStreamDestroy(comm_streams[s]);
            }
            if (s < comms.size() && comms[s] != nullptr)
                (void)rccl().comm_destroy(comms[s]);
        }
        comm_devices.clear();
        comms.clear();
        comm_streams.clear();
        (void)hipSetDevice(current);
    }

    // Communicator and exchange streams over `devices` (shard order): made when the device list changes.
    void ensure_communicator(const std::vector<int>& devices)
    {
        if (devices == comm_devices && !comms.empty())
            return;
        drop_communicator();
        const Rccl& api = rccl();
        std::vector<Rccl::Comm> made(devices.size(), nullptr);
        if (api.comm_init_all(made.data(), static_cast<int>(devices.size()), devices.data()) != 0)
        {
            for (Rccl::Comm c : made)
                if (c != nullptr)
                    (void)api.comm_destroy(c);
            raise(CHARLS_AMD_ERRC_DEVICE_FAILURE);
        }
        comm_devices = devices;
        comms = made;
        comm_streams.assign(devices.size(), nullptr);
        for (size_t s = 0; s < devices.size(); ++s)
            if (hipSetDevice(devices[s]) != hipSuccess || hipStreamCreateWithFlags(&comm_streams[s], hipStreamNonBlocking) != hipSuccess)
            {
                drop_communicator();
                raise(CHARLS_AMD_ERRC_DEVICE_FAILURE);
            }
    }

    ~charls_amd_devices()
    {
        drop_communicator();
        workers.clear(); // joins the threads; each frees its work areas on its way out
    }
};

namespace {

// The context behind charls_amd_encode_batch_devices / charls_amd_decode_batch_devices and behind a NULL context argument.
// Never destroyed by the runtime (worker threads must not outlive-or-race the HIP runtime's own teardown at exit);
// charls_amd_devices_destroy(NULL) releases what it holds.
std::mutex g_default_guard;
charls_amd_devices* g_default = nullptr;

charls_amd_devices& context_or_default(charls_amd_devices* ctx)
{
    if (ctx != nullptr)
        return *ctx;
    std::lock_guard<std::mutex> lock(g_default_guard);
    if (g_default == nullptr)
        g_default = new charls_amd_devices;
    return *g_default;
}

// Runs `work(shard)` for every shard on the context's worker of the shard's device; the first error wins.  `after(s)` runs on
// the calling thread as soon as shards 0..s have all succeeded -- while the later shards are still at work.
template <typename Work, typename After>
charls_jpegls_errc on_every_shard(charls_amd_devices& ctx, uint32_t shard_count, const charls_amd_device_shard* shards, Work work, After after)
{
    std::vector<Worker*> used(shard_count, nullptr);
    std::map<int, uint32_t> seen; // shards per device so far
    for (uint32_t s = 0; s < shard_count; ++s)
        used[s] = &ctx.worker(shards[s].device, seen[shards[s].device]++); // (may throw bad_alloc / system_error: nothing submitted yet)
    uint32_t submitted = 0;
    charls_jpegls_errc first_error = CHARLS_JPEGLS_ERRC_SUCCESS;
    try
    {
        for (; submitted < shard_count; ++submitted)
            used[submitted]->submit([&work, s = submitted]() -> charls_jpegls_errc { return work(s); });
    }
    catch (...)
    { // (a task could not be handed over: the ones that were are still waited for -- they refer to this frame's locals)
        first_error = current_exception_to_errc();
    }
    for (uint32_t s = 0; s < submitted; ++s)
    {
        const charls_jpegls_errc e = used[s]->wait();
        if (first_error == CHARLS_JPEGLS_ERRC_SUCCESS)
            first_error = e;
        if (first_error == CHARLS_JPEGLS_ERRC_SUCCESS && submitted == shard_count)
        {
            try
            {
                after(s);
            }
            catch (...)
            { // (the remaining shards are still waited for)
                first_error = current_exception_to_errc();
            }
        }
    }
    return first_error;
}

template <typename Work>
charls_jpegls_errc on_every_shard(charls_amd_devices& ctx, uint32_t shard_count, const charls_amd_device_shard* shards, Work work)
{
    return on_every_shard(ctx, shard_count, shards, work, [](uint32_t) {});
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

// ---- the context API -------------------------------------------------------------------------------------------------

extern "C" charls_amd_devices* charls_amd_devices_create(void)
{
    try
    {
        return new charls_amd_devices;
    }
    catch (...)
    {
        return nullptr;
    }
}

extern "C" void charls_amd_devices_destroy(charls_amd_devices* context)
{
    if (context != nullptr)
    {
        { // a call that is running on the context finishes first (destroying a context WHILE starting a call on it stays the
          // caller's bug: nothing can make a pointer that is being deleted safe to pass in)
            std::lock_guard<std::mutex> running(context->call);
        }
        delete context;
        return;
    }
    // NULL: what the default context holds (its threads, their work areas, the communicator)
    charls_amd_devices* d = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_default_guard);
        d = g_default;
        g_default = nullptr;
    }
    if (d != nullptr)
    {
        std::lock_guard<std::mutex> running(d->call);
    }
    delete d;
}

extern "C" uint64_t charls_amd_devices_work_area_bytes(charls_amd_devices* context)
try
{
    charls_amd_devices& ctx = context_or_default(context);
    std::lock_guard<std::mutex> lock(ctx.call);
    std::atomic<uint64_t> total{0};
    for (auto& entry : ctx.workers)
        entry.second->submit([&total]() -> charls_jpegls_errc {
            total.fetch_add(dev::work_area_bytes());
            return CHARLS_JPEGLS_ERRC_SUCCESS;
        });
    for (auto& entry : ctx.workers)
        (void)entry.second->wait();
    return total.load();
}
catch (...)
{
    return 0;
}

extern "C" charls_jpegls_errc charls_amd_devices_release_work_areas(charls_amd_devices* context)
try
{
    charls_amd_devices& ctx = context_or_default(context);
    std::lock_guard<std::mutex> lock(ctx.call);
    for (auto& entry : ctx.workers)
        entry.second->submit([]() -> charls_jpegls_errc {
            dev::release_work_areas();
            return CHARLS_JPEGLS_ERRC_SUCCESS;
        });
    for (auto& entry : ctx.workers)
        (void)entry.second->wait();
    return CHARLS_JPEGLS_ERRC_SUCCESS;
}
catch (...)
{
    return current_exception_to_errc();
}

extern "C" charls_jpegls_errc charls_amd_devices_encode_batch(charls_amd_devices* context, const charls_amd_codec_params* params,
                                                              uint32_t shard_count, const charls_amd_device_shard* shards,
                                                              size_t frame_pitch_bytes, uint32_t stride, size_t stream_pitch_bytes,
                                                              uint64_t* sizes, charls_jpegls_errc* errcs, const charls_amd_gather* gather)
try
{
    check_pointer(params);
    check_pointer(sizes);
    check_pointer(errcs);
    dev::require_device();
    check_shards(shard_count, shards);
    charls_amd_devices& ctx = context_or_default(context);
    std::lock_guard<std::mutex> lock(ctx.call);
    std::vector<uint64_t> first(shard_count + 1, 0);
    for (uint32_t s = 0; s < shard_count; ++s)
        first[s + 1] = first[s] + shards[s].frame_count;
    int caller_device = 0;
    (void)hipGetDevice(&caller_device);

    auto code_shard = [&](uint32_t s) -> charls_jpegls_errc {
        const charls_amd_device_shard& sh = shards[s];
        if (sh.frame_count == 0)
            return CHARLS_JPEGLS_ERRC_SUCCESS;
        hipStream_t stream = static_cast<hipStream_t>(sh.hip_stream);
        return charls_amd_encode_batch_device(params, sh.frame_count, sh.d_frames, frame_pitch_bytes, stride, sh.d_streams,
                                              stream_pitch_bytes, sizes + first[s], errcs + first[s], stream);
    };
    if (gather == nullptr)
        return on_every_shard(ctx, shard_count, shards, code_shard);

    // ---- with a hand-over of the bitstreams to the root shard's device.  The "all-gather of sizes" of SURVEY 8e is a prefix
    // sum here (every shard's sizes are host values of this process), and it is known for shard s as soon as shards 0..s have
    // coded: the streams of shard s start moving then, while the later shards are still coding (round 4 moved everything
    // strictly after all coding).
    check_argument(gather->root_shard < shard_count);
    check_pointer(gather->d_gathered);
    check_pointer(gather->offsets);
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
    if (gather->transport == CHARLS_AMD_TRANSPORT_AUTO && !use_rccl && shard_count > 1 && distinct_devices)
        note_once("charls_amd: RCCL (librccl.so) is not usable in this process: bitstreams are gathered with hipMemcpyPeerAsync");

    struct RestoreDevice
    {
        int device;
        ~RestoreDevice() { (void)hipSetDevice(device); }
    } restore{caller_device};
    std::vector<int> devices(shard_count);
    for (uint32_t s = 0; s < shard_count; ++s)
        devices[s] = shards[s].device;
    if (use_rccl)
        ctx.ensure_communicator(devices); // (cached: ncclCommInitAll runs when the list of devices changes)
    else
        for (uint32_t s = 0; s < shard_count; ++s)
            if (s != root && shards[s].device != root_device)
                enable_peer_access(root_device, shards[s].device);
    hip_check(hipSetDevice(root_device));
    StreamGuard own; // the root's own copies and the peer copies
    hip_check(hipSetDevice(caller_device));
    uint64_t at = 0;
    bool exchange_failed = false;

    auto hand_over = [&](uint32_t s) {
        // offsets of shard s (frames that failed take no room)
        for (uint64_t f = first[s]; f < first[s + 1]; ++f)
        {
            gather->offsets[f] = at;
            at += errcs[f] == CHARLS_JPEGLS_ERRC_SUCCESS ? sizes[f] : 0;
        }
        if (at > gather->capacity_bytes)
            raise(CHARLS_JPEGLS_ERRC_DESTINATION_TOO_SMALL);
        hip_check(hipSetDevice(root_device));
        auto source = [&](uint64_t f) { return static_cast<const uint8_t*>(shards[s].d_streams) + (f - first[s]) * stream_pitch_bytes; };
        if (s == root || shards[s].device == root_device)
        {
            for (uint64_t f = first[s]; f < first[s + 1]; ++f)
                if (errcs[f] == CHARLS_JPEGLS_ERRC_SUCCESS && sizes[f] != 0)
                    hip_check(hipMemcpyAsync(gathered + gather->offsets[f], source(f), sizes[f], hipMemcpyDeviceToDevice, own.s));
        }
        else if (!use_rccl)
        {
            for (uint64_t f = first[s]; f < first[s + 1]; ++f)
                if (errcs[f] == CHARLS_JPEGLS_ERRC_SUCCESS && sizes[f] != 0)
                    hip_check(hipMemcpyPeerAsync(gathered + gather->offsets[f], root_device, source(f), shards[s].device, sizes[f], own.s));
        }
        else
        { // one grouped send / receive pair per frame, in rounds of 64 frames (point-to-point over xGMI)
            const Rccl& api = rccl();
            constexpr uint64_t kRound = 64;
            for (uint64_t lo = first[s]; lo < first[s + 1] && !exchange_failed; lo += kRound)
            {
                const uint64_t hi = std::min<uint64_t>(first[s + 1], lo + kRound);
                exchange_failed = api.group_start() != 0;
                for (uint64_t f = lo; f < hi && !exchange_failed; ++f)
                {
                    if (errcs[f] != CHARLS_JPEGLS_ERRC_SUCCESS || sizes[f] == 0)
                        continue;
                    (void)hipSetDevice(devices[s]);
                    exchange_failed = exchange_failed || api.send(source(f), sizes[f], kNcclUint8, static_cast<int>(root), ctx.comms[s], ctx.comm_streams[s]) != 0;
                    (void)hipSetDevice(root_device);
                    exchange_failed = exchange_failed || api.recv(gathered + gather->offsets[f], sizes[f], kNcclUint8, static_cast<int>(s), ctx.comms[root],
                                                                  ctx.comm_streams[root]) != 0;
                }
                exchange_failed = api.group_end() != 0 || exchange_failed;
            }
            if (exchange_failed)
                raise(CHARLS_AMD_ERRC_DEVICE_FAILURE);
        }
        (void)hipSetDevice(caller_device);
    };

    charls_jpegls_errc result = on_every_shard(ctx, shard_count, shards, code_shard, hand_over);
    // everything that was queued has to land (or be given up) before the buffers go back to the caller
    bool sync_failed = false;
    (void)hipSetDevice(root_device);
    sync_failed = hipStreamSynchronize(own.s) != hipSuccess || sync_failed;
    if (use_rccl)
        for (uint32_t s = 0; s < shard_count; ++s)
        {
            (void)hipSetDevice(devices[s]);
            sync_failed = hipStreamSynchronize(ctx.comm_streams[s]) != hipSuccess || sync_failed;
        }
    (void)hipSetDevice(caller_device);
    if (use_rccl && (exchange_failed || sync_failed))
        ctx.drop_communicator(); // (its state after a failed group is unknown: the next call starts a new one)
    if (result == CHARLS_JPEGLS_ERRC_SUCCESS && sync_failed)
        result = CHARLS_AMD_ERRC_DEVICE_FAILURE;
    if (result == CHARLS_JPEGLS_ERRC_SUCCESS && gather->total_bytes != nullptr)
        *gather->total_bytes = at;
    return result;
}
catch (...)
{
    return current_exception_to_errc();
}

extern "C" charls_jpegls_errc charls_amd_devices_decode_batch(charls_amd_devices* context, uint32_t shard_count,
                                                              const charls_amd_device_shard* shards, size_t stream_pitch_bytes,
                                                              const uint64_t* sizes, size_t frame_pitch_bytes, uint32_t stride,
                                                              charls_amd_codec_params* params_out, charls_jpegls_errc* errcs)
try
{
    check_pointer(sizes);
    check_pointer(errcs);
    dev::require_device();
    check_shards(shard_count, shards);
    charls_amd_devices& ctx = context_or_default(context);
    std::lock_guard<std::mutex> lock(ctx.call);
    std::vector<uint64_t> first(shard_count + 1, 0);
    for (uint32_t s = 0; s < shard_count; ++s)
        first[s + 1] = first[s] + shards[s].frame_count;
    std::vector<charls_amd_codec_params> found(shard_count);
    const charls_jpegls_errc coded = on_every_shard(ctx, shard_count, shards, [&](uint32_t s) -> charls_jpegls_errc {
        const charls_amd_device_shard& sh = shards[s];
        if (sh.frame_count == 0)
            return CHARLS_JPEGLS_ERRC_SUCCESS;
        // (in a decode shard d_streams is the source and d_frames the destination)
        return charls_amd_decode_batch_device(sh.frame_count, sh.d_streams, stream_pitch_bytes, sizes + first[s],
                                              const_cast<void*>(sh.d_frames), frame_pitch_bytes, stride, &found[s], errcs + first[s],
                                              static_cast<hipStream_t>(sh.hip_stream));
    });
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

// ---- the same two calls on the process-wide default context
extern "C" charls_jpegls_errc charls_amd_encode_batch_devices(const charls_amd_codec_params* params, uint32_t shard_count,
                                                              const charls_amd_device_shard* shards, size_t frame_pitch_bytes,
                                                              uint32_t stride, size_t stream_pitch_bytes, uint64_t* sizes,
                                                              charls_jpegls_errc* errcs, const charls_amd_gather* gather)
{
    return charls_amd_devices_encode_batch(nullptr, params, shard_count, shards, frame_pitch_bytes, stride, stream_pitch_bytes, sizes,
                                           errcs, gather);
}

extern "C" charls_jpegls_errc charls_amd_decode_batch_devices(uint32_t shard_count, const charls_amd_device_shard* shards,
                                                              size_t stream_pitch_bytes, const uint64_t* sizes,
                                                              size_t frame_pitch_bytes, uint32_t stride,
                                                              charls_amd_codec_params* params_out, charls_jpegls_errc* errcs)
{
    return charls_amd_devices_decode_batch(nullptr, shard_count, shards, stream_pitch_bytes, sizes, frame_pitch_bytes, stride, params_out,
                                           errcs);
}
