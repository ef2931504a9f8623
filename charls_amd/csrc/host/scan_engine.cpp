// scan_engine.cpp -- see scan_engine.h.
#include "scan_engine.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../device/knobs.h"
#include "coalescer.h"

namespace jls {

using dev::hip_check;

namespace {

constexpr int kMaxDevices = 32;
int slot_of(int device) noexcept
{
    return device >= 0 && device < kMaxDevices ? device : 0;
}

// Idle resource sets.  Never destroyed: at process exit the HIP runtime may already be gone.
struct ResourcePool
{
    std::mutex mutex;
    std::vector<std::unique_ptr<EngineResources>> idle;
    size_t idle_bytes[kMaxDevices]{}; // per device
};
ResourcePool& pool()
{
    static ResourcePool* p = new ResourcePool;
    return *p;
}
// What stays idle: a pool of threads that makes a handle per image (cli/benchmark.cpp does) gives a set back and takes one a
// moment later, and hipFree waits for EVERY kernel on the device -- seconds, while another thread's decoder runs.  So up to
// kMaxIdleSets sets of at most kMaxIdleBytesPerSet each stay, as long as those of a DEVICE add up to less than kMaxIdleBytes (a
// 16th of an MI355X; never more than charls_amd_set_workspace_limit allows).  They do not stay for long: what the host-pointer
// ABI keeps between calls -- these sets, the shared work areas of the merged encoder launches, blocks whose hipFree was put off --
// is given back once no call has come for kIdleReleaseMs (the janitor below), and charls_amd_release_work_areas() gives it back
// at once.  A drop-in library is a guest in the caller's HBM.
constexpr size_t kMaxIdleSets = 1024;
constexpr size_t kMaxIdleBytesPerSet = size_t{512} << 20;
constexpr size_t kMaxIdleBytes = size_t{18} << 30;
constexpr long long kIdleReleaseMs = 2000;

size_t idle_cap() noexcept
{
    const uint64_t limit = dev::workspace_limit();
    return limit != 0 ? std::min<size_t>(kMaxIdleBytes, static_cast<size_t>(limit)) : kMaxIdleBytes;
}

// ---- the janitor: one thread per process, started by the first coding call of the host-pointer ABI.  It wakes a few times a
// second; when no coding call is running and none has ended for kIdleReleaseMs it frees what the calls left behind.
struct Janitor
{
    std::mutex guard;
    std::condition_variable wake;
    std::thread thread;
    std::atomic<bool> started{false};
    bool stop{};
    std::atomic<long long> last_activity_ms{0};
    std::atomic<int> calls_running{0};
    std::atomic<uint64_t> releases{0};
};
Janitor& janitor()
{
    static Janitor* j = new Janitor; // never destroyed (its thread is joined by an atexit handler)
    return *j;
}
long long steady_ms() noexcept
{
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
bool anything_held() noexcept
{
    {
        ResourcePool& p = pool();
        std::lock_guard<std::mutex> lock(p.mutex);
        if (!p.idle.empty())
            return true;
    }
    return dev::shared_work_area_bytes() != 0 || dev::deferred_free_bytes() != 0;
}
void janitor_loop()
{
    Janitor& j = janitor();
    std::unique_lock<std::mutex> lock(j.guard);
    while (!j.stop)
    {
#ifdef JLS_TSAN // (gcc's ThreadSanitizer does not intercept the steady-clock wait: coalescer.h)
        j.wake.wait_until(lock, std::chrono::system_clock::now() + std::chrono::milliseconds(250));
#else
        j.wake.wait_for(lock, std::chrono::milliseconds(250));
#endif
        if (j.stop)
            break;
        const long long quiet_ms = knobs::get_or(knobs::kIdleReleaseMs, kIdleReleaseMs);
        if (quiet_ms <= 0 || j.calls_running.load() != 0 || steady_ms() - j.last_activity_ms.load() < quiet_ms)
            continue;
        lock.unlock();
        if (anything_held())
        {
            release_idle_engine_resources();
            dev::release_shared_work_areas();
            dev::reap_deferred_frees();
            j.releases.fetch_add(1);
        }
        lock.lock();
    }
}
void stop_janitor() noexcept
{
    Janitor& j = janitor();
    {
        std::lock_guard<std::mutex> lock(j.guard);
        if (!j.started.load(std::memory_order_relaxed))
            return;
        j.stop = true;
    }
    j.wake.notify_all();
    if (j.thread.joinable())
        j.thread.join();
}
// A coding call of the host-pointer ABI begins / ends.
struct Activity
{
    Activity() noexcept
    {
        Janitor& j = janitor();
        j.calls_running.fetch_add(1);
        j.last_activity_ms.store(steady_ms());
        if (!j.started.load(std::memory_order_acquire))
        {
            std::lock_guard<std::mutex> lock(j.guard);
            if (!j.started.load(std::memory_order_relaxed) && !j.stop)
            {
                try
                {
                    j.thread = std::thread(janitor_loop);
                    j.started.store(true, std::memory_order_release);
                    std::atexit(stop_janitor);
                }
                catch (...)
                { // (no thread to be had: nothing decays, charls_amd_release_work_areas() still works)
                }
            }
        }
    }
    ~Activity()
    {
        Janitor& j = janitor();
        j.last_activity_ms.store(steady_ms());
        j.calls_running.fetch_sub(1);
    }
    Activity(const Activity&) = delete;
    Activity& operator=(const Activity&) = delete;
};

std::unique_ptr<EngineResources> acquire_resources()
{
    int device = 0;
    hip_check(hipGetDevice(&device));
    {
        ResourcePool& p = pool();
        std::lock_guard<std::mutex> lock(p.mutex);
        for (size_t i = p.idle.size(); i-- > 0;)
            if (p.idle[i]->device == device)
            {
                std::unique_ptr<EngineResources> r = std::move(p.idle[i]);
                p.idle.erase(p.idle.begin() + static_cast<std::ptrdiff_t>(i));
                p.idle_bytes[slot_of(device)] -= std::min(p.idle_bytes[slot_of(device)], r->pooled_bytes);
                return r;
            }
    }
    auto r = std::make_unique<EngineResources>();
    r->device = device;
    hip_check(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    return r;
}

void release_resources(std::unique_ptr<EngineResources> r) noexcept
{
    if (!r)
        return;
    if (r->device_bytes() <= kMaxIdleBytesPerSet)
    {
        ResourcePool& p = pool();
        std::lock_guard<std::mutex> lock(p.mutex);
        size_t& held = p.idle_bytes[slot_of(r->device)];
        if (p.idle.size() < kMaxIdleSets && held + r->device_bytes() <= idle_cap())
        {
            r->pooled_bytes = r->device_bytes();
            held += r->pooled_bytes;
            p.idle.push_back(std::move(r));
            janitor().last_activity_ms.store(steady_ms()); // (what has just come back is not old yet)
            return;
        }
    }
    // (r is destroyed here: too large to keep, or the pool is full)
}

// ---- the streams the shared launches run on.  The device runs the kernels of only a few hardware queues side by side, and
// the runtime deals its queues out per stream PRIORITY: a decoder launch is ONE kernel that runs for seconds (a serial chain
// per scan), and on the handles' own streams it shared its hardware queue with whatever stream came next -- an encoder batch
// of 250 frames that should take 0.1 s waited 3.4 s behind an unrelated decoder kernel (profiles/r05_threads_call_trace.txt).
// So decoder launches run on a few streams of the LOWEST priority (their own queues; they leave most of every CU idle and
// give way), encoder launches on one stream of the HIGHEST priority per device; the handles' streams only carry copies.
struct LaunchStreams
{
    std::mutex guard;
    std::vector<hipStream_t> idle_decode; // lowest priority; made on demand, kept
    hipStream_t encode{};                 // highest priority (the encoder lane runs one batch at a time)
};
LaunchStreams& launch_streams(int device)
{
    static LaunchStreams* all = new LaunchStreams[kMaxDevices]; // never destroyed: the HIP runtime may be gone at exit
    return all[device >= 0 && device < kMaxDevices ? device : 0];
}
void priority_range(int& lowest, int& highest)
{
    lowest = highest = 0;
    if (hipDeviceGetStreamPriorityRange(&lowest, &highest) != hipSuccess)
    {
        (void)hipGetLastError();
        lowest = highest = 0;
    }
}
hipStream_t take_decode_stream(int device)
{
    LaunchStreams& ls = launch_streams(device);
    {
        std::lock_guard<std::mutex> lock(ls.guard);
        if (!ls.idle_decode.empty())
        {
            hipStream_t s = ls.idle_decode.back();
            ls.idle_decode.pop_back();
            return s;
        }
    }
    int lowest = 0, highest = 0;
    priority_range(lowest, highest);
    hipStream_t s{};
    hip_check(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lowest));
    return s;
}
void give_back_decode_stream(int device, hipStream_t s) noexcept
{
    LaunchStreams& ls = launch_streams(device);
    std::lock_guard<std::mutex> lock(ls.guard);
    ls.idle_decode.push_back(s);
}
hipStream_t encode_stream(int device)
{
    LaunchStreams& ls = launch_streams(device);
    std::lock_guard<std::mutex> lock(ls.guard);
    if (ls.encode == nullptr)
    {
        int lowest = 0, highest = 0;
        priority_range(lowest, highest);
        hip_check(hipStreamCreateWithPriority(&ls.encode, hipStreamNonBlocking, highest));
    }
    return ls.encode;
}

Coalescer& coalescer()
{
    static Coalescer* c = new Coalescer; // never destroyed: calls may be in flight at process exit
    return *c;
}

bool coalescing_enabled()
{
    return knobs::get_or(knobs::kCoalesce, 1) != 0;
}

// The coalescer's lanes and the launch streams are tables over devices 0 .. kMaxDevices - 1; a device beyond them (no such box
// exists) codes every call for itself.
static_assert(2 * kMaxDevices <= Coalescer::kLanes, "a lane per device and direction");
bool coalescing_enabled_on(int device)
{
    return coalescing_enabled() && device >= 0 && device < kMaxDevices;
}

bool tracing()
{
    return knobs::get_or(knobs::kTrace, 0) != 0;
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int lane_of(int device, bool decode)
{
    return device * 2 + (decode ? 1 : 0);
}

// How long an announcement counts and the leader of a batch waits for announced calls at most.  What they are late by is their
// host -> device copy (and whatever their thread does between configuring its handle and calling the coding function), so
// the wait is priced in the work it delays: a decoder launch is one serial chain per scan (0.2 us per sample: seconds for
// a large frame), an eighth of that is cheap; an encoder launch of one frame takes about as long as the frame's upload,
// so it waits for up to 64 uploads.  Never more than 400 ms / 50 ms.
uint32_t merge_wait_us(const ScanDesc& d, bool decode)
{
    if (const long long knob = knobs::get(knobs::kCoalesceWaitUs); knob != knobs::kUnset)
        return static_cast<uint32_t>(std::clamp<long long>(knob, 0, 100'000'000));
    const double samples = static_cast<double>(d.width) * d.height * std::max(1, d.components);
    if (decode)
        return static_cast<uint32_t>(std::clamp(samples * 0.2 / 8, 50.0, 400e3));
    const double upload_us = samples * (d.bits_per_sample > 8 ? 2 : 1) / 50e3; // 50 GB/s
    return static_cast<uint32_t>(std::clamp(64 * upload_us, 200.0, 50e3));
}

// Decoder batches of one geometry that may run side by side: the device runs kernels of a few streams at once, and a second
// batch beside the first costs nothing while the first leaves most of the chip idle; beyond that a batch waits (and collects).
constexpr uint32_t kDecodeBatchesAtOnce = 3;
constexpr uint32_t kMaxMergedScans = 16384;

} // namespace

void coalescer_stats(uint64_t out[5]) noexcept
{
    const Coalescer::Stats s = coalescer().stats();
    out[0] = s.calls;
    out[1] = s.launches;
    out[2] = s.merged;
    out[3] = s.largest;
    out[4] = s.split;
}

uint64_t frame_hint(uint32_t width, uint32_t height, int bits_per_sample) noexcept
{
    return Coalescer::geometry_hint(width, height, bits_per_sample);
}

EngineResources::~EngineResources()
{
    if (stream)
        (void)hipStreamDestroy(stream);
}

void release_idle_engine_resources() noexcept
{
    std::vector<std::unique_ptr<EngineResources>> gone;
    {
        ResourcePool& p = pool();
        std::lock_guard<std::mutex> lock(p.mutex);
        gone.swap(p.idle);
        for (size_t& bytes : p.idle_bytes)
            bytes = 0;
    }
}

uint64_t idle_engine_resource_bytes() noexcept
{
    ResourcePool& p = pool();
    std::lock_guard<std::mutex> lock(p.mutex);
    uint64_t total = 0;
    for (const size_t bytes : p.idle_bytes)
        total += bytes;
    return total;
}

uint64_t idle_releases() noexcept
{
    return janitor().releases.load();
}

ScanEngine::~ScanEngine()
{
    end_call();
    release_resources(std::move(r_));
}

void ScanEngine::ensure_stream()
{
    dev::require_device();
    if (!r_)
        r_ = acquire_resources();
}

// How long an announcement made BEFORE the coding call holds a leader up at most: the threads of a pool go from configuring
// their handle to the coding call in microseconds; a caller that only wanted the header costs nobody more than this.
constexpr uint32_t kEarlyAnnouncementUs = 1000;

void ScanEngine::expect_call(bool decode, uint64_t hint, bool uploading) noexcept
{
    if (!coalescing_enabled() || dev::device_status() != CHARLS_JPEGLS_ERRC_SUCCESS)
        return;
    const uint32_t fresh_us = uploading ? Coalescer::kForever : kEarlyAnnouncementUs;
    if (ticket_ != 0)
    { // (announced before: the call has got further, or knows its geometry now)
        if (uploading || hint != 0)
            coalescer().renew(announced_lane_, ticket_, hint, fresh_us);
        return;
    }
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
    {
        (void)hipGetLastError();
        return;
    }
    if (!coalescing_enabled_on(device))
        return;
    announced_lane_ = lane_of(device, decode);
    ticket_ = coalescer().announce(announced_lane_, hint, fresh_us);
}

void ScanEngine::end_call() noexcept
{
    if (trace_.begin_ms != 0 && tracing())
    { // (CHARLS_AMD_TRACE: where the call's time went; submit = waiting for / leading the shared launch, launch = this thread ran it)
        std::fprintf(stderr, "charls_amd trace %s begin=%.1f total=%.1f upload=%.1f sync=%.1f submit=%.1f (launch=%.1f of %u scans) copy_out=%.1f ms\n",
                     trace_.decode ? "decode" : "encode", trace_.begin_ms, now_ms() - trace_.begin_ms, trace_.upload_ms, trace_.sync_ms,
                     trace_.submit_ms, trace_.launch_ms, trace_.launched_scans, trace_.copy_out_ms);
    }
    trace_ = Trace{};
    dev::reap_deferred_frees(); // (blocks that were given up while a decoder launch ran; nothing to do as a rule)
    if (ticket_ != 0)
        coalescer().retract(announced_lane_, ticket_);
    ticket_ = 0;
    announced_lane_ = -1;
}

ScanDesc ScanEngine::make_desc(const ScanSpec& s) const
{
    ScanDesc d{};
    d.width = s.width;
    d.height = s.height;
    d.components = s.components;
    d.interleave_mode = s.interleave_mode;
    d.bits_per_sample = s.bits_per_sample;
    d.near_lossless = s.near_lossless;
    d.color_transformation = s.color_transformation;
    d.t1 = s.pc.threshold1;
    d.t2 = s.pc.threshold2;
    d.t3 = s.pc.threshold3;
    d.reset = static_cast<uint8_t>(s.pc.reset_value); // reference src/scan_codec.hpp:142
    d.restart_interval = s.restart_interval;
    return d;
}

ScanResult ScanEngine::run(const ScanDesc& desc, bool decode)
{
    ScanResult r;
    run_many(&desc, 1, decode, &r);
    return r;
}

// One launch for descs[0, n) on `stream`, with this handle's staging areas.
void ScanEngine::launch(const ScanDesc* descs, uint32_t n, bool decode, ScanResult* results, hipStream_t stream)
{
    const size_t desc_bytes = sizeof(ScanDesc) * n, result_bytes = sizeof(ScanResult) * n;
    // (with room to spare: growing a buffer frees the old one, and hipFree waits for every kernel on the device -- another
    // batch's decoder, seconds)
    auto* staged = static_cast<uint8_t*>(r_->staging.ensure(dev::with_headroom(desc_bytes + result_bytes)));
    std::memcpy(staged, descs, desc_bytes);
    auto* d_descs = static_cast<ScanDesc*>(r_->desc.ensure(dev::with_headroom(desc_bytes)));
    auto* d_results = static_cast<ScanResult*>(r_->result.ensure(dev::with_headroom(result_bytes)));
    hip_check(hipMemcpyAsync(d_descs, staged, desc_bytes, hipMemcpyHostToDevice, stream));
    ScanDesc proto = descs[0];
    if (decode)
        dev::launch_decode(proto, d_descs, d_results, n, stream);
    else
    {
        for (uint32_t i = 1; i < n; ++i) // (an upper bound of every scan's capacity: launch_encode sizes the raw streams by it)
            proto.stream_capacity = std::max(proto.stream_capacity, descs[i].stream_capacity);
        dev::launch_encode(proto, d_descs, d_results, n, stream);
    }
    hip_check(hipMemcpyAsync(staged + desc_bytes, d_results, result_bytes, hipMemcpyDeviceToHost, stream));
    hip_check(hipStreamSynchronize(stream));
    std::memcpy(results, staged + desc_bytes, result_bytes);
}

void ScanEngine::run_many(const ScanDesc* descs, uint32_t count, bool decode, ScanResult* results)
{
    const Activity activity; // (the janitor keeps its hands off while a call runs, and counts the quiet time from its end)
    constexpr size_t kKeepBytes = size_t{1} << 30; // a drop-in library must not sit on gigabytes of the caller's HBM between calls
    if (!coalescing_enabled_on(r_->device))
    {
        if (ticket_ != 0)
            end_call();
        launch(descs, count, decode, results, r_->stream);
        if (dev::thread_work_area_bytes() > kKeepBytes)
            dev::release_thread_work_areas();
        return;
    }
    // What this call uploaded is in HBM before anybody's launch may read it.
    const double t_sync = tracing() ? now_ms() : 0;
    hip_check(hipStreamSynchronize(r_->stream));
    if (tracing())
        trace_.sync_ms += now_ms() - t_sync;
    const int lane = lane_of(r_->device, decode);
    if (ticket_ != 0 && announced_lane_ != lane)
        end_call(); // (announced for another device or direction: nobody waits for that any longer)
    const Coalescer::Ticket ticket = ticket_;
    ticket_ = 0; // (submit takes the announcement back)
    announced_lane_ = -1;
    const uint32_t wait_us = merge_wait_us(descs[0], decode);
    const Coalescer::Launch run_batch = [this, decode](const ScanDesc* all, uint32_t n, ScanResult* out) {
        const double t_launch = tracing() ? now_ms() : 0;
        struct Note
        {
            ScanEngine& e;
            double t0;
            uint32_t n;
            ~Note()
            {
                if (tracing())
                    e.trace_.launch_ms += now_ms() - t0, e.trace_.launched_scans = n;
            }
        } note{*this, t_launch, n};
        if (decode)
        { // decoder launches keep next to nothing between calls and run side by side
            struct Borrowed
            {
                int device;
                hipStream_t s;
                ~Borrowed() { give_back_decode_stream(device, s); }
            } borrowed{r_->device, take_decode_stream(r_->device)};
            struct Running
            { // (hipFree of any thread would wait for this launch: frees are set aside while it runs, runtime.hip)
                Running() { dev::long_kernel_begins(); }
                ~Running() { dev::long_kernel_ends(); }
            } running;
            launch(all, n, true, out, borrowed.s);
            if (dev::thread_work_area_bytes() > kKeepBytes)
                dev::release_thread_work_areas();
            return;
        }
        // encoder launches share the device's work areas: one merged launch at a time (the coalescer's exclusive lane
        // already sees to that; the scope also keeps charls_amd_release_work_areas() of another thread out)
        dev::SharedAreasScope shared;
        launch(all, n, false, out, encode_stream(r_->device));
        // (the shared areas stay: they never grow beyond dev::shared_areas_keep_bytes(), and a pool of threads that codes in
        // rounds would otherwise pay for gigabytes of hipMalloc + hipFree -- seconds -- in every round;
        // charls_amd_release_work_areas() gives them back)
    };
    // (the grace: a decoder batch that had to queue is seconds late already, a quarter of the wait more lets the threads of the
    // batch before it come back and join; the gap: on a lane that several threads are seen to use a decoder batch goes when no
    // call has joined for a quarter of the wait; encoder batches are short and collect what arrives while they queue)
    const Coalescer::Policy policy{wait_us, kMaxMergedScans, decode ? kDecodeBatchesAtOnce : 0u, decode ? wait_us / 4 : 0u, decode ? wait_us / 4 : 0u};
    const double t_submit = tracing() ? now_ms() : 0;
    struct Queued
    {
        ScanEngine& e;
        double t0;
        ~Queued()
        {
            if (tracing())
                e.trace_.submit_ms += now_ms() - t0;
        }
    } queued{*this, t_submit};
    trace_.decode = decode;
    coalescer().submit(lane, merge_key_of(descs[0]), descs, count, results, ticket, policy, run_batch);
}

void ScanEngine::encode_planes(const ScanSpec& spec, uint32_t count, size_t plane_bytes, size_t stride, size_t capacity, ScanResult* results)
{
    ensure_stream();
    const size_t bound = dev::worst_case_scan_bytes(spec.width, spec.height, spec.components, spec.bits_per_sample);
    plane_capacity_ = (std::min(capacity, bound) + 255) & ~size_t{255};
    auto* bits = static_cast<uint8_t*>(r_->bits.ensure(plane_capacity_ * count));
    const size_t scratch_samples = dev::line_scratch_samples(spec.width, spec.interleave_mode, spec.components);
    auto* scratch = static_cast<uint16_t*>(r_->scratch.ensure(scratch_samples * sizeof(uint16_t) * count));
    std::vector<ScanDesc> descs(count, make_desc(spec));
    for (uint32_t c = 0; c < count; ++c)
    {
        descs[c].pixels = r_->pixels.as<uint8_t>() + plane_bytes * c;
        descs[c].pixel_stride = stride;
        descs[c].stream = bits + plane_capacity_ * c;
        descs[c].stream_capacity = std::min(capacity, bound);
        descs[c].line_scratch = scratch + scratch_samples * c;
    }
    run_many(descs.data(), count, false, results);
}

void ScanEngine::fetch_encoded_scan(uint32_t index, uint8_t* destination, size_t bytes)
{
    copy_out(destination, r_->bits.as<uint8_t>() + plane_capacity_ * index, bytes);
}

void ScanEngine::decode_planes(const ScanSpec& spec, const size_t* stream_offsets, uint32_t count, ScanResult* results)
{
    ensure_stream();
    const size_t row_bytes = static_cast<size_t>(spec.width) * bytes_per_sample(spec.bits_per_sample);
    const size_t plane = row_bytes * spec.height;
    auto* pixels = static_cast<uint8_t*>(r_->pixels.ensure(plane * count));
    const size_t scratch_samples = dev::line_scratch_samples(spec.width, spec.interleave_mode, spec.components);
    auto* scratch = static_cast<uint16_t*>(r_->scratch.ensure(scratch_samples * sizeof(uint16_t) * count));
    std::vector<ScanDesc> descs(count, make_desc(spec));
    for (uint32_t c = 0; c < count; ++c)
    {
        descs[c].pixels = pixels + plane * c;
        descs[c].pixel_stride = row_bytes;
        descs[c].stream = r_->bits.as<uint8_t>() + stream_offsets[c];
        descs[c].stream_capacity = stream_bytes_ - stream_offsets[c];
        descs[c].line_scratch = scratch + scratch_samples * c;
    }
    run_many(descs.data(), count, true, results);
}

void ScanEngine::fetch_decoded_plane(const ScanSpec& spec, uint32_t index, uint8_t* destination, size_t stride)
{
    const size_t row_bytes = static_cast<size_t>(spec.width) * bytes_per_sample(spec.bits_per_sample);
    copy_rows_out(destination, stride, r_->pixels.as<uint8_t>() + row_bytes * spec.height * index, row_bytes, spec.height);
}

// Device -> host on the handle's own stream (the null stream would queue the copies of all calling threads behind each other).
void ScanEngine::copy_out(uint8_t* destination, const uint8_t* device_source, size_t bytes)
{
    if (bytes == 0)
        return;
    const double t0 = tracing() ? now_ms() : 0;
    hip_check(hipMemcpyAsync(destination, device_source, bytes, hipMemcpyDeviceToHost, r_->stream));
    hip_check(hipStreamSynchronize(r_->stream));
    if (tracing())
        trace_.copy_out_ms += now_ms() - t0;
}

void ScanEngine::copy_rows_out(uint8_t* destination, size_t stride, const uint8_t* device_source, size_t row_bytes, size_t rows)
{
    if (stride == row_bytes)
    { // packed rows: one copy
        copy_out(destination, device_source, row_bytes * rows);
        return;
    }
    hip_check(hipMemcpy2DAsync(destination, stride, device_source, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, r_->stream));
    hip_check(hipStreamSynchronize(r_->stream));
}

void ScanEngine::upload_pixels(const uint8_t* source, size_t bytes, uint64_t hint)
{
    ensure_stream();
    expect_call(false, hint, true); // (before the copy: the copy is what the others of a batch wait for)
    const double t0 = tracing() ? now_ms() : 0;
    r_->pixels.ensure(bytes);
    pixel_bytes_ = bytes;
    hip_check(hipMemcpyAsync(r_->pixels.as<uint8_t>(), source, bytes, hipMemcpyHostToDevice, r_->stream));
    if (tracing())
        trace_.upload_ms += now_ms() - t0, trace_.begin_ms = trace_.begin_ms == 0 ? t0 : trace_.begin_ms;
}

size_t ScanEngine::encode_scan(const ScanSpec& spec, size_t pixel_offset, size_t stride, uint8_t* destination,
                               size_t destination_size)
{
    ensure_stream();
    // Never allocate more than the scan can possibly produce; a larger destination behaves identically.
    const size_t bound = dev::worst_case_scan_bytes(spec.width, spec.height, spec.components, spec.bits_per_sample);
    const size_t capacity = std::min(destination_size, bound);
    ScanDesc d = make_desc(spec);
    d.pixels = r_->pixels.as<uint8_t>() + pixel_offset;
    d.pixel_stride = stride;
    d.stream = static_cast<uint8_t*>(r_->bits.ensure(capacity));
    d.stream_capacity = capacity;
    d.line_scratch = static_cast<uint16_t*>(
        r_->scratch.ensure(dev::line_scratch_samples(spec.width, spec.interleave_mode, spec.components) * sizeof(uint16_t)));
    const ScanResult r = run(d, false);
    if (r.errc != kOk)
        raise(static_cast<charls_jpegls_errc>(r.errc));
    copy_out(destination, d.stream, r.bytes);
    return r.bytes;
}

void ScanEngine::upload_stream(const uint8_t* source, size_t bytes, uint64_t hint)
{
    ensure_stream();
    expect_call(true, hint, true);
    const double t0 = tracing() ? now_ms() : 0;
    r_->bits.ensure(bytes + 16); // the ring refill of the wave decoder reads whole 16-byte groups
    stream_bytes_ = bytes;
    hip_check(hipMemcpyAsync(r_->bits.as<uint8_t>(), source, bytes, hipMemcpyHostToDevice, r_->stream));
    if (tracing())
        trace_.upload_ms += now_ms() - t0, trace_.begin_ms = trace_.begin_ms == 0 ? t0 : trace_.begin_ms;
}

size_t ScanEngine::decode_scan(const ScanSpec& spec, size_t stream_offset, uint8_t* destination, size_t stride)
{
    ensure_stream();
    const size_t planes = spec.interleave_mode == 0 ? 1 : static_cast<size_t>(spec.components);
    const size_t row_bytes = planes * spec.width * bytes_per_sample(spec.bits_per_sample);
    ScanDesc d = make_desc(spec);
    d.pixels = static_cast<uint8_t*>(r_->pixels.ensure(row_bytes * spec.height)); // device rows are packed
    d.pixel_stride = row_bytes;
    d.stream = r_->bits.as<uint8_t>() + stream_offset;
    d.stream_capacity = stream_bytes_ - stream_offset;
    d.line_scratch = static_cast<uint16_t*>(
        r_->scratch.ensure(dev::line_scratch_samples(spec.width, spec.interleave_mode, spec.components) * sizeof(uint16_t)));
    const ScanResult r = run(d, true);
    if (r.errc != kOk) // the destination content after a failed decode is unspecified in the reference as well
        raise(static_cast<charls_jpegls_errc>(r.errc));
    copy_rows_out(destination, stride, d.pixels, row_bytes, spec.height);
    return r.bytes;
}

} // namespace jls
