// runtime.hip -- device discovery, arenas and kernel launches (see runtime.h).  Compiled for gfx950 only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "knobs.h"
#include "runtime.h"
#include "scan_serial.hip"
#include "container_kernels.hip"
#include "scan_wave_decode.hip"
#include "pipeline_common.hip"
#include "block_stuffing.hip"
#include "speculative_stuffing.hip"
#include "tile_pipeline.hip"
#include "scan_fast_decode.hip"
#include "scan_group_decode.hip"
#include "group_launch.h"
#include "restart_intervals.hip"

namespace jls::dev {

namespace {
std::once_flag g_once;
charls_jpegls_errc g_status = CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE;
std::atomic<int32_t> g_engine{0};
std::atomic<uint64_t> g_workspace_limit{0}; // charls_amd_set_workspace_limit; 0 = a quarter of the device

void probe() noexcept
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    {
        (void)hipGetLastError();
        return;
    }
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
        return;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return;
    // The code objects in this library are gfx950 only; any other device cannot run them.
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && std::getenv("CHARLS_AMD_ALLOW_ANY_ARCH") == nullptr)
        return;
    g_status = CHARLS_JPEGLS_ERRC_SUCCESS;
}
} // namespace

charls_jpegls_errc device_status() noexcept
{
    std::call_once(g_once, probe);
    return g_status;
}

void require_device()
{
    if (device_status() != CHARLS_JPEGLS_ERRC_SUCCESS)
        raise(CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE);
}

// hipFree waits for every kernel that is running on the device.  While one of this library's decoder launches of the
// host-pointer ABI is in flight on a device -- ONE kernel that runs for seconds -- a block of that device that is given up (a
// buffer that grows) is not freed but set aside, and freed once no such launch is running there (profiles/r05_concurrent_kernels.txt:
// an encoder call that grew the shared work areas beside a running decoder took 2.9 s instead of 2 ms).  What is set aside is
// invisible HBM, so it is bounded: beyond kMaxDeferredBytes per device a block is freed on the spot (and waits), and an
// allocation that fails frees everything that was set aside -- waiting for the running launch if it has to -- and tries again.
namespace {
constexpr int kMaxDevices = 32;
constexpr size_t kMaxDeferredBytes = size_t{4} << 30;
struct DeferredFrees
{
    std::mutex guard;
    std::vector<std::pair<void*, size_t>> blocks[kMaxDevices];
    size_t bytes[kMaxDevices]{};
    std::atomic<int> long_kernels[kMaxDevices]{};
};
DeferredFrees& deferred()
{
    static DeferredFrees* d = new DeferredFrees; // never destroyed: a thread may give a buffer up while the process exits
    return *d;
}
int slot_of(int device) noexcept
{
    return device >= 0 && device < kMaxDevices ? device : 0;
}
int current_device() noexcept
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
    {
        (void)hipGetLastError();
        device = 0;
    }
    return device;
}
// Frees what was set aside on `device` (the calling thread's current device is restored).  `even_while_running`: the caller
// needs the memory now and waits for the launch that is in flight.
void reap_device(int device, bool even_while_running) noexcept
{
    DeferredFrees& d = deferred();
    const int slot = slot_of(device);
    if (!even_while_running && d.long_kernels[slot].load() > 0)
        return;
    std::vector<std::pair<void*, size_t>> gone;
    {
        std::lock_guard<std::mutex> lock(d.guard);
        gone.swap(d.blocks[slot]);
        d.bytes[slot] = 0;
    }
    if (gone.empty())
        return;
    const int before = current_device();
    if (before != device)
        (void)hipSetDevice(device);
    for (const auto& block : gone)
        (void)hipFree(block.first);
    if (before != device)
        (void)hipSetDevice(before);
}
} // namespace

void* DeviceBuffer::ensure(size_t bytes)
{
    int device = 0;
    hip_check(hipGetDevice(&device));
    // (memory of another device is of no use to the kernels this thread is about to launch: a thread that moved on to
    // another device starts over there)
    if (bytes <= cap_ && ptr_ && device == device_)
        return ptr_;
    release();
    const size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&ptr_, want);
    if (e == hipErrorOutOfMemory)
    { // what this library set aside is memory too
        (void)hipGetLastError();
        ptr_ = nullptr;
        reap_device(device, true);
        e = hipMalloc(&ptr_, want);
    }
    if (e != hipSuccess)
        ptr_ = nullptr;
    hip_check(e);
    cap_ = want;
    device_ = device;
    return ptr_;
}

void reap_deferred_frees() noexcept
{
    for (int device = 0; device < kMaxDevices; ++device)
        if (deferred().bytes[device] != 0) // (a racy look is fine: whoever set a block aside reaps again when its call ends)
            reap_device(device, false);
}

uint64_t deferred_free_bytes() noexcept
{
    DeferredFrees& d = deferred();
    std::lock_guard<std::mutex> lock(d.guard);
    uint64_t total = 0;
    for (int device = 0; device < kMaxDevices; ++device)
        total += d.bytes[device];
    return total;
}

void long_kernel_begins() noexcept
{
    deferred().long_kernels[slot_of(current_device())].fetch_add(1);
}

void long_kernel_ends() noexcept
{
    deferred().long_kernels[slot_of(current_device())].fetch_sub(1);
}

void DeviceBuffer::release() noexcept
{
    if (ptr_)
    {
        DeferredFrees& d = deferred();
        const int slot = slot_of(device_);
        bool set_aside = false;
        if (d.long_kernels[slot].load() > 0)
        {
            std::lock_guard<std::mutex> lock(d.guard);
            if (d.bytes[slot] + cap_ <= kMaxDeferredBytes)
            {
                d.blocks[slot].emplace_back(ptr_, cap_);
                d.bytes[slot] += cap_;
                set_aside = true;
            }
        }
        if (!set_aside)
        { // (the block may live on another device than the thread's current one: a handle that moved on)
            const int before = current_device();
            if (device_ >= 0 && before != device_)
                (void)hipSetDevice(device_);
            (void)hipFree(ptr_);
            if (device_ >= 0 && before != device_)
                (void)hipSetDevice(before);
        }
    }
    ptr_ = nullptr;
    cap_ = 0;
}

PinnedBuffer::~PinnedBuffer()
{
    if (ptr_)
        (void)hipHostFree(ptr_);
}

void* PinnedBuffer::ensure(size_t bytes)
{
    if (bytes <= cap_ && ptr_)
        return ptr_;
    if (ptr_)
        (void)hipHostFree(ptr_);
    ptr_ = nullptr;
    cap_ = 0;
    hip_check(hipHostMalloc(&ptr_, bytes < 256 ? 256 : bytes, hipHostMallocDefault));
    cap_ = bytes < 256 ? 256 : bytes;
    return ptr_;
}

EncodeEngine encode_engine() noexcept
{
    return static_cast<EncodeEngine>(g_engine.load());
}

void set_encode_engine(EncodeEngine e) noexcept
{
    g_engine.store(static_cast<int32_t>(e));
}

void set_workspace_limit(uint64_t bytes) noexcept
{
    g_workspace_limit.store(bytes);
}

uint64_t workspace_limit() noexcept
{
    return g_workspace_limit.load();
}

Timings& last_timings() noexcept
{
    static thread_local Timings t{};
    return t;
}

namespace {
std::atomic<uint64_t> g_speculation[tile_counter_count]{};
std::atomic<uint64_t> g_exact_retry_scans{0}; // scans a speed-path decoder handed to the exact decoder (it did not end cleanly there)
std::atomic<uint64_t> g_serial_fallback_scans{0}; // scans the tile pipeline was eligible for that ran on the one-wavefront kernel: no work area
void note_pipeline_fallback(uint32_t scans) noexcept
{
    g_serial_fallback_scans.fetch_add(scans);
}
}
uint64_t pipeline_fallback_scans() noexcept
{
    return g_serial_fallback_scans.load();
}
uint64_t exact_retry_scans() noexcept
{
    return g_exact_retry_scans.load();
}
void speculation_counters(uint64_t out[tile_counter_count]) noexcept
{
    for (int i = 0; i < tile_counter_count; ++i)
        out[i] = g_speculation[i].load();
}

void launch_encode_serial(const ScanDesc* d_descs, ScanResult* d_results, uint32_t count, hipStream_t stream)
{
    if (count == 0)
        return;
    hipLaunchKernelGGL(encode_scans_serial, dim3(count), dim3(64), 0, stream, d_descs, d_results);
    hip_check(hipGetLastError());
}

void launch_decode_serial(const ScanDesc* d_descs, ScanResult* d_results, uint32_t count, hipStream_t stream)
{
    if (count == 0)
        return;
    hipLaunchKernelGGL(decode_scans_serial, dim3(count), dim3(64), 0, stream, d_descs, d_results);
    hip_check(hipGetLastError());
}

namespace {

size_t wave_decode_lds(const ScanDesc& d)
{
    const size_t planes = d.interleave_mode == 0 ? 1 : static_cast<size_t>(d.components);
    return wave::kFixedLds + planes * (static_cast<size_t>(d.width) + 2) * (d.bits_per_sample > 8 ? 2 : 1);
}

bool wave_decode_eligible(const ScanDesc& d)
{
    if (wave_decode_lds(d) > kMaxDynamicLds)
        return false; // line does not fit LDS
    if (d.reset == 0)
        return false; // RESET = 256*m is stored as 0 by the reference: N is never halved and outgrows the packed context
    if (d.bits_per_sample > 8 && d.interleave_mode == 0 &&
        ((reinterpret_cast<uintptr_t>(d.pixels) | d.pixel_stride) & 1u) != 0)
        return false; // odd row address for 16-bit samples
    return true;
}

// Scans whose restart intervals are decoded as scans of their own (restart_intervals.hip).
bool interval_decode_candidate(const ScanDesc& d)
{
    return d.restart_interval != 0 && d.restart_interval < d.height && d.height <= 65535 && d.restart_interval <= 65535 &&
           wave_decode_eligible(d) && knobs::get_or(knobs::kSequentialIntervals, 0) == 0;
}

size_t fast_decode_lds(const ScanDesc& d)
{
    const size_t line_bytes = ((static_cast<size_t>(d.width) + 2) * (d.bits_per_sample > 8 ? 2 : 1) + 3) & ~size_t{3};
    return (d.bits_per_sample > 8 ? fast::fixed_lds<uint16_t>() : fast::fixed_lds<uint8_t>()) + line_bytes;
}

// Lines a scan of the group decoder keeps in LDS: one, or one per component of a line-interleaved scan.
uint32_t group_lines(const ScanDesc& d)
{
    return d.interleave_mode == 1 ? static_cast<uint32_t>(d.components) : 1u;
}

size_t group_lds_bytes(const ScanDesc& d, uint32_t scans_per_wave)
{
    return d.bits_per_sample > 8 ? grp::workgroup_lds_bytes<uint16_t>(d.width, scans_per_wave, group_lines(d))
                                 : grp::workgroup_lds_bytes<uint8_t>(d.width, scans_per_wave, group_lines(d));
}

// Lanes per scan (G) and wavefronts per workgroup (W) of the speed path (scan_group_decode.hip) for a launch of `count` scans;
// lanes 0 = the one-scan-per-wavefront kernel (scan_fast_decode.hip).
//
// What a sample costs: a wavefront stops for every event of every one of its scans (run mode, end of line, refill), so the
// fewer scans share a wavefront the less a step takes -- 206 ns with 2 scans per wavefront, 217 with 4, 232 with 8 -- as long
// as the wavefront has a SIMD TO ITSELF: two wavefronts of this kernel on one SIMD take a third longer each (4.8 s instead of
// 3.6 s for a frame's 16.8 M samples: profiles/r05_decode_wavefronts_per_workgroup.txt).  So the rule is one wavefront per
// SIMD, and the fewest scans per wavefront that allows:
//  * up to ONE wavefront per CU, one-wavefront workgroups (W = 1); with two per CU they still find a SIMD each, but a workgroup
//    of four is 4 % faster there too (1024 frames at 32 lanes: 2.91 s against 3.03 s, round 6), so W = 1 beyond one per CU is
//    left to the scans that have no W = 4 instantiation (several lines per pixel row);
//  * beyond two per CU the dispatcher doubles one-wavefront workgroups up on some SIMDs while others idle -- not in every launch:
//    1024 of them decoded 4096 frames in 3.67 s in one call and in 4.89 s in others, at an unchanged 2.39 GHz
//    (profiles/r05_pmc_decode_effective_clock.txt; rounds 3 and 4 took the slow launches for the chip clocking down and never
//    went beyond two wavefronts per CU).  A workgroup of FOUR wavefronts that takes the whole LDS of its CU is dealt out one
//    wavefront per SIMD by construction: 4096 frames at 16 lanes per scan in 3.76 s, every time (8 lanes, W = 1: 4.05 s).
// DECODE_GROUP / DECODE_WORKGROUP_WAVES override (0, 4, 8, 16, 32 lanes; 1, 4, 8 wavefronts).
struct GroupPlan
{
    int lanes, waves;
};
GroupPlan decode_group_plan(const ScanDesc& d, uint32_t count)
{
    if (d.bits_per_sample > 8 && d.t3 > grp::kMaxTableT3)
        return {0, 1};
    const bool one_line = group_lines(d) == 1; // (the W > 1 instantiations exist for single-component scans)
    int forced = static_cast<int>(knobs::get_or(knobs::kDecodeGroup, -1));
    int forced_waves = static_cast<int>(knobs::get_or(knobs::kDecodeWorkgroupWaves, -1));
    if (d.near_lossless != 0)
    { // (the near-lossless instantiations: 8 / 16 / 32 lanes, workgroups of one or four wavefronts)
        forced = forced == 4 ? -1 : forced;
        forced_waves = forced_waves == 8 ? -1 : forced_waves;
    }
    if (forced == 0)
        return {0, 1};
    if (forced == 4 || forced == 8 || forced == 16 || forced == 32)
    {
        const uint32_t per_wave = 64u / static_cast<uint32_t>(forced);
        if ((forced_waves == 4 || forced_waves == 8) && one_line && (forced == 16 || forced == 32) &&
            group_lds_bytes(d, per_wave * static_cast<uint32_t>(forced_waves)) <= kGroupDecodeLds)
            return {forced, forced_waves};
        if (group_lds_bytes(d, per_wave) <= kGroupDecodeLds)
            return {forced, 1};
    }
    static const uint32_t cus = [] {
        hipDeviceProp_t prop;
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess)
            return 256u;
        return static_cast<uint32_t>(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
    }();
    const uint32_t waves_per_cu = static_cast<uint32_t>(std::clamp<long long>(knobs::get_or(knobs::kDecodeWavesPerCu, 2), 1, 16));
    const bool allow_workgroups = one_line && forced_waves != 1;
    int fallback = 0;
    for (int lanes = 32; lanes >= 8; lanes /= 2)
    {
        const uint32_t per_wave = 64u / static_cast<uint32_t>(lanes);
        if (group_lds_bytes(d, per_wave) > kGroupDecodeLds)
            break;
        fallback = lanes;
        const uint32_t waves = (count + per_wave - 1) / per_wave;
        if (waves <= cus)
            return {lanes, 1}; // a CU per wavefront
        // four wavefronts per CU as ONE workgroup (it must be the only one its CU can hold: more than half of the LDS): ahead of
        // two one-wavefront workgroups per CU since round 6 -- 1024 frames at 32 lanes: 2.91 s against 3.03 s
        // (profiles/r06_decode_launch_shapes.txt)
        const uint32_t per_group = 4 * per_wave;
        if (allow_workgroups && lanes >= 16 && group_lds_bytes(d, per_group) <= kGroupDecodeLds && (count + per_group - 1) / per_group <= cus)
            return {lanes, 4};
        if (waves <= waves_per_cu * cus)
            return {lanes, 1};
    }
    return {fallback, 1};
}
int decode_group_lanes(const ScanDesc& d, uint32_t count)
{
    return decode_group_plan(d, count).lanes;
}

constexpr uint32_t kPixelWaves = 512; // wavefronts a launch of the pixel kernels (scan_group_pixels.hip, scan_group_encode.hip) aims at: two per CU

// Lanes per scan of the speed path of sample-interleaved scans, lossless or near-lossless, and of near-lossless
// single-component scans (scan_group_pixels.hip); 0 = the exact decoder.  Packing as in decode_group_lanes.
int pixel_group_lanes(const ScanDesc& d, uint32_t count)
{
    const bool by_sample = d.interleave_mode == 2 && d.components >= 2 && d.components <= 4;
    // (near-lossless single-component and line-interleaved scans have decode_scans_group<.., kNear> since round 6;
    // NEAR_DECODE_PIXELS=1 brings them back here: the A/B)
    const bool near_here = knobs::get_or(knobs::kNearDecodePixels, 0) != 0;
    const bool near_planar = d.interleave_mode == 0 && d.components == 1 && d.near_lossless != 0 && near_here;
    const bool near_by_line = d.interleave_mode == 1 && d.components >= 2 && d.components <= 4 && d.near_lossless != 0 && near_here;
    if ((!by_sample && !near_planar && !near_by_line) || !wave_decode_eligible(d))
        return 0;
    if (by_sample && d.bits_per_sample > 8 && ((reinterpret_cast<uintptr_t>(d.pixels) | d.pixel_stride) & 1u) != 0)
        return 0; // odd row address for 16-bit samples
    if ((d.bits_per_sample > 8 && d.t3 > grp::kMaxTableT3) || knobs::get_or(knobs::kExactDecoder, 0) != 0)
        return 0;
    const int forced = static_cast<int>(knobs::get_or(knobs::kDecodeGroup, -1));
    if (forced == 0)
        return 0;
    if ((forced == 8 || forced == 16 || forced == 32) && pixel_group_lds_bytes(d, 64u / forced) <= kGroupDecodeLds)
        return forced;
    int best = 0;
    for (int lanes = 32; lanes >= 8; lanes /= 2)
    {
        const uint32_t per_wave = 64u / static_cast<uint32_t>(lanes);
        if (pixel_group_lds_bytes(d, per_wave) > kGroupDecodeLds)
            break;
        best = lanes;
        if ((count + per_wave - 1) / per_wave <= kPixelWaves) // (two one-wavefront workgroups per CU find a SIMD each, see decode_group_plan)
            break;
    }
    return best;
}

// Lossless single-component scans take the speed path (scan_group_decode.hip / scan_fast_decode.hip); it defers to the
// exact kernels whenever the scan does not end cleanly.
bool fast_decode_eligible(const ScanDesc& d)
{
    const bool planar = d.interleave_mode == 0 && d.components == 1;
    const bool by_line = d.interleave_mode == 1 && d.components >= 2 && d.components <= 4; // group kernel only
    // (near-lossless scans: the group kernel only)
    const bool near_ok = d.near_lossless == 0 ||
                         (knobs::get_or(knobs::kNearDecodePixels, 0) == 0 && decode_group_lanes(d, 1) != 0);
    return wave_decode_eligible(d) && near_ok && (planar || by_line) &&
           ((planar && fast_decode_lds(d) <= kMaxDynamicLds) || decode_group_lanes(d, 1) != 0) &&
           knobs::get_or(knobs::kExactDecoder, 0) == 0;
}

// Scans the speed path handed back (ScanResult.flags & kFastRetry) are gathered so that ONE launch of the exact decoder
// takes all of them.
__global__ void gather_retries(const ScanDesc* __restrict__ descs, const ScanResult* __restrict__ results, uint32_t count,
                               ScanDesc* __restrict__ retry_descs, uint32_t* __restrict__ retry_index,
                               uint32_t* __restrict__ retry_count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && (results[i].flags & fast::kFastRetry) != 0)
    {
        const uint32_t slot = atomicAdd(retry_count, 1u);
        retry_index[slot] = i;
        retry_descs[slot] = descs[i];
    }
}

__global__ void scatter_retries(const ScanResult* __restrict__ retry_results, const uint32_t* __restrict__ retry_index,
                                uint32_t retries, ScanResult* __restrict__ results)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < retries)
        results[retry_index[s]] = retry_results[s];
}

template <typename S>
void launch_wave_decode(int nc, const ScanDesc* d_descs, ScanResult* d_results, uint32_t count, size_t lds,
                        hipStream_t stream)
{
    switch (nc)
    {
    case 1:
        hipLaunchKernelGGL((decode_scans_wave<S, 1>), dim3(count), dim3(64), lds, stream, d_descs, d_results);
        break;
    case 2:
        hipLaunchKernelGGL((decode_scans_wave<S, 2>), dim3(count), dim3(64), lds, stream, d_descs, d_results);
        break;
    case 3:
        hipLaunchKernelGGL((decode_scans_wave<S, 3>), dim3(count), dim3(64), lds, stream, d_descs, d_results);
        break;
    default:
        hipLaunchKernelGGL((decode_scans_wave<S, 4>), dim3(count), dim3(64), lds, stream, d_descs, d_results);
        break;
    }
}
} // namespace

uint64_t decode_launch_key(const ScanDesc& d) noexcept
{
    // width | components | interleave | wide | eligible : scans with equal keys can share a launch
    const uint64_t base = (static_cast<uint64_t>(d.width) << 16) | (static_cast<uint64_t>(d.components & 0xFF) << 8) |
                          (static_cast<uint64_t>(d.interleave_mode & 3) << 4) |
                          (fast_decode_eligible(d) || pixel_group_lanes(d, 1) != 0 ? 4u : 0u) |
                          (d.bits_per_sample > 8 ? 2u : 0u) | (wave_decode_eligible(d) ? 1u : 0u);
    if (!interval_decode_candidate(d))
        return base;
    // interval-parallel decode also needs equal height and restart interval (both < 2^16 here, as is the width)
    return (uint64_t{1} << 63) | (static_cast<uint64_t>(d.restart_interval) << 47) | (static_cast<uint64_t>(d.height) << 31) |
           (static_cast<uint64_t>(d.width & 0xFFFF) << 15) | (static_cast<uint64_t>(d.components & 7) << 12) |
           (static_cast<uint64_t>(d.interleave_mode & 3) << 10) | (base & 7u);
}

namespace {
// Side streams of the pipeline (per thread and device): the stuffing stage of a pass runs on one of them under the next pass's
// first stages, the run chain on another under the walkers of the regular chains.
constexpr int kMaxPipelineLanes = 2;
struct PipelineLanes
{
    hipStream_t streams[kMaxPipelineLanes]{};
    bool created = false;
    int device = -1; // streams belong to a device: a thread that moved on to another one gets new streams there
    ~PipelineLanes() { destroy(); }
    void destroy() noexcept
    {
        if (!created)
            return;
        int current = 0;
        const bool switched = hipGetDevice(&current) == hipSuccess && current != device && hipSetDevice(device) == hipSuccess;
        for (hipStream_t s : streams)
            (void)hipStreamDestroy(s);
        if (switched)
            (void)hipSetDevice(current);
        created = false;
    }
    void ensure()
    {
        int current = 0;
        hip_check(hipGetDevice(&current));
        if (created && current == device)
            return;
        destroy();
        for (hipStream_t& s : streams)
            hip_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        created = true;
        device = current;
    }
};

// Work areas: the HBM (and the side streams) a thread's launches keep between calls -- the tile pipeline's arena, the private
// stream buffers of the planar batch encoder, the small tables of the restart-interval and retry paths.  Every thread has
// a set of its own (the batch API; the workers of multi_device.cpp); the host-pointer ABI runs its merged launches on ONE
// set per device that is shared by all calling threads (SharedAreasScope), so that a pool of 256 threads does not end up with
// 256 arenas.
constexpr int kIntervalArenas = 8;
struct WorkAreas
{
    DeviceBuffer pipeline, plane, interval[kIntervalArenas];
    PipelineLanes lanes;
    size_t bytes() const noexcept
    {
        size_t total = pipeline.capacity() + plane.capacity();
        for (const DeviceBuffer& b : interval)
            total += b.capacity();
        return total;
    }
    void release() noexcept
    {
        pipeline.release();
        plane.release();
        for (DeviceBuffer& b : interval)
            b.release();
    }
};
thread_local WorkAreas* t_shared_areas = nullptr; // set by SharedAreasScope
WorkAreas& areas()
{
    if (t_shared_areas != nullptr)
        return *t_shared_areas;
    static thread_local WorkAreas mine;
    return mine;
}
DeviceBuffer& interval_arena(int which)
{
    return areas().interval[which];
}
PipelineLanes& pipeline_lanes()
{
    return areas().lanes;
}

void launch_decode_plain(const ScanDesc& proto, const ScanDesc* d_descs, ScanResult* d_results, uint32_t count,
                         hipStream_t stream);

// All `count` scans share proto's geometry and restart interval.
void launch_decode_intervals(const ScanDesc& proto, const ScanDesc* d_descs, ScanResult* d_results, uint32_t count,
                             hipStream_t stream)
{
    const uint32_t lines = proto.restart_interval;
    const uint32_t intervals = (proto.height + lines - 1) / lines;
    const uint32_t max_marks = intervals - 1;
    const size_t subs_n = static_cast<size_t>(count) * intervals;
    auto* d_marks = static_cast<uint32_t*>(interval_arena(0).ensure(sizeof(uint32_t) * count * max_marks));
    auto* d_counts = static_cast<uint32_t*>(interval_arena(1).ensure(sizeof(uint32_t) * count));
    auto* d_subs = static_cast<ScanDesc*>(interval_arena(2).ensure(sizeof(ScanDesc) * subs_n));
    auto* d_sub_results = static_cast<ScanResult*>(interval_arena(3).ensure(sizeof(ScanResult) * subs_n));
    hipLaunchKernelGGL(interval::find_restart_markers, dim3(count), dim3(64), 0, stream, d_descs, d_marks, max_marks, d_counts);
    hip_check(hipGetLastError());
    std::vector<uint32_t> counts(count);
    hip_check(hipMemcpyAsync(counts.data(), d_counts, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, stream));
    hip_check(hipStreamSynchronize(stream));
    bool regular = true;
    for (uint32_t c : counts)
        regular = regular && c == max_marks;
    if (!regular)
    { // a stream without the expected markers: the sequential decoder reports what the reference reports
        launch_decode_plain(proto, d_descs, d_results, count, stream);
        return;
    }
    hipLaunchKernelGGL(interval::build_decode_intervals, dim3(intervals, count), dim3(1), 0, stream, d_descs, d_marks, intervals,
                       d_subs);
    hip_check(hipGetLastError());
    ScanDesc sub_proto = proto;
    sub_proto.height = lines;
    sub_proto.restart_interval = 0;
    launch_decode_plain(sub_proto, d_subs, d_sub_results, static_cast<uint32_t>(subs_n), stream);
    hipLaunchKernelGGL(interval::check_intervals, dim3((count + 63) / 64), dim3(64), 0, stream, d_descs, d_marks, intervals,
                       d_sub_results, d_results, count);
    hip_check(hipGetLastError());
    std::vector<ScanResult> results(count);
    hip_check(hipMemcpyAsync(results.data(), d_results, sizeof(ScanResult) * count, hipMemcpyDeviceToHost, stream));
    hip_check(hipStreamSynchronize(stream));
    for (uint32_t i = 0; i < count; ++i)
        if ((results[i].flags & interval::kIntervalRetry) != 0)
            launch_decode_plain(proto, d_descs + i, d_results + i, 1, stream);
}
} // namespace

void launch_decode(const ScanDesc& proto, const ScanDesc* d_descs, ScanResult* d_results, uint32_t count,
                   hipStream_t stream)
{
    if (count == 0)
        return;
    if (interval_decode_candidate(proto))
        launch_decode_intervals(proto, d_descs, d_results, count, stream);
    else
        launch_decode_plain(proto, d_descs, d_results, count, stream);
}

namespace {
void launch_decode_plain(const ScanDesc& proto, const ScanDesc* d_descs, ScanResult* d_results, uint32_t count,
                         hipStream_t stream)
{
    if (count == 0)
        return;
    if (!wave_decode_eligible(proto))
    {
        launch_decode_serial(d_descs, d_results, count, stream);
        return;
    }
    const int nc = proto.interleave_mode == 2 ? proto.components : 1;
    const size_t lds = wave_decode_lds(proto);
    auto exact = [&](const ScanDesc* descs, ScanResult* results, uint32_t n) {
        if (proto.bits_per_sample > 8)
            launch_wave_decode<uint16_t>(nc, descs, results, n, lds, stream);
        else
            launch_wave_decode<uint8_t>(nc, descs, results, n, lds, stream);
        hip_check(hipGetLastError());
    };
    const int pixel_lanes = pixel_group_lanes(proto, count);
    if (pixel_lanes != 0)
    { // sample-interleaved scans, lossless or near-lossless
        launch_decode_pixels(proto, pixel_lanes, d_descs, d_results, count, stream);
    }
    else if (!fast_decode_eligible(proto))
    {
        exact(d_descs, d_results, count);
        return;
    }
    else
    {
    const int group = decode_group_lanes(proto, count);
    if (group == 0)
    {
        if (proto.bits_per_sample > 8)
            hipLaunchKernelGGL((decode_scans_fast<uint16_t>), dim3(count), dim3(64), fast_decode_lds(proto), stream, d_descs, d_results);
        else
            hipLaunchKernelGGL((decode_scans_fast<uint8_t>), dim3(count), dim3(64), fast_decode_lds(proto), stream, d_descs, d_results);
    }
    else
    {
        const uint32_t per_wave = 64u / static_cast<uint32_t>(group);
        const int wg_waves = decode_group_plan(proto, count).waves;
        const uint32_t per_group = per_wave * static_cast<uint32_t>(wg_waves);
        const dim3 grid((count + per_group - 1) / per_group);
        // (a workgroup of several wavefronts is to have its CU to itself -- one wavefront per SIMD: it asks for more than half
        // of the CU's LDS whatever its scans need)
        const size_t lds = wg_waves > 1 ? std::max<size_t>(group_lds_bytes(proto, per_group), kGroupDecodeLds / 2 + 1024) : group_lds_bytes(proto, per_group);
#define JLS_LAUNCH_GROUP_NWK(S, G, N, W, K)                                                                                  \
    do                                                                                                                       \
    {                                                                                                                        \
        if (lds > kMaxDynamicLds) /* more than the default limit of dynamic LDS per workgroup */                              \
            hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_scans_group<S, G, N, W, K>),                 \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));               \
        hipLaunchKernelGGL((decode_scans_group<S, G, N, W, K>), grid, dim3(64 * W), lds, stream, d_descs, d_results, count); \
    } while (0)
#define JLS_LAUNCH_GROUP_NW(S, G, N, W) JLS_LAUNCH_GROUP_NWK(S, G, N, W, false)
#define JLS_LAUNCH_GROUP_N(S, G, N) JLS_LAUNCH_GROUP_NW(S, G, N, 1)
#define JLS_LAUNCH_GROUP(S, G)                                                                                           \
    do                                                                                                                   \
    {                                                                                                                    \
        const uint32_t nl = group_lines(proto);                                                                          \
        if (nl == 1) JLS_LAUNCH_GROUP_N(S, G, 1);                                                                        \
        else if (nl == 2) JLS_LAUNCH_GROUP_N(S, G, 2);                                                                   \
        else if (nl == 3) JLS_LAUNCH_GROUP_N(S, G, 3);                                                                   \
        else JLS_LAUNCH_GROUP_N(S, G, 4);                                                                                \
    } while (0)
        const bool wide = proto.bits_per_sample > 8;
        if (proto.near_lossless != 0)
        { // near-lossless: 8 / 16 / 32 lanes
            const uint32_t nl = group_lines(proto);
#define JLS_LAUNCH_NEAR(S, G, W)                                                                                         \
    do                                                                                                                   \
    {                                                                                                                    \
        if (nl == 1) JLS_LAUNCH_GROUP_NWK(S, G, 1, W, true);                                                             \
        else if (nl == 2) JLS_LAUNCH_GROUP_NWK(S, G, 2, 1, true);                                                        \
        else if (nl == 3) JLS_LAUNCH_GROUP_NWK(S, G, 3, 1, true);                                                        \
        else JLS_LAUNCH_GROUP_NWK(S, G, 4, 1, true);                                                                     \
    } while (0)
            if (wg_waves == 4 && group == 16)
            {
                if (wide) JLS_LAUNCH_NEAR(uint16_t, 16, 4); else JLS_LAUNCH_NEAR(uint8_t, 16, 4);
            }
            else if (wg_waves == 4 && group == 32)
            {
                if (wide) JLS_LAUNCH_NEAR(uint16_t, 32, 4); else JLS_LAUNCH_NEAR(uint8_t, 32, 4);
            }
            else if (group == 8)
            {
                if (wide) JLS_LAUNCH_NEAR(uint16_t, 8, 1); else JLS_LAUNCH_NEAR(uint8_t, 8, 1);
            }
            else if (group == 16)
            {
                if (wide) JLS_LAUNCH_NEAR(uint16_t, 16, 1); else JLS_LAUNCH_NEAR(uint8_t, 16, 1);
            }
            else
            {
                if (wide) JLS_LAUNCH_NEAR(uint16_t, 32, 1); else JLS_LAUNCH_NEAR(uint8_t, 32, 1);
            }
#undef JLS_LAUNCH_NEAR
        }
        else if (wg_waves == 4 && group == 16)
        {
            if (wide) JLS_LAUNCH_GROUP_NW(uint16_t, 16, 1, 4); else JLS_LAUNCH_GROUP_NW(uint8_t, 16, 1, 4);
        }
        else if (wg_waves == 4 && group == 32)
        {
            if (wide) JLS_LAUNCH_GROUP_NW(uint16_t, 32, 1, 4); else JLS_LAUNCH_GROUP_NW(uint8_t, 32, 1, 4);
        }
        else if (wg_waves == 8 && group == 16)
        {
            if (wide) JLS_LAUNCH_GROUP_NW(uint16_t, 16, 1, 8); else JLS_LAUNCH_GROUP_NW(uint8_t, 16, 1, 8);
        }
        else if (wg_waves == 8 && group == 32)
        {
            if (wide) JLS_LAUNCH_GROUP_NW(uint16_t, 32, 1, 8); else JLS_LAUNCH_GROUP_NW(uint8_t, 32, 1, 8);
        }
        else if (group == 4)
        {
            if (wide) JLS_LAUNCH_GROUP(uint16_t, 4); else JLS_LAUNCH_GROUP(uint8_t, 4);
        }
        else if (group == 8)
        {
            if (wide) JLS_LAUNCH_GROUP(uint16_t, 8); else JLS_LAUNCH_GROUP(uint8_t, 8);
        }
        else if (group == 16)
        {
            if (wide) JLS_LAUNCH_GROUP(uint16_t, 16); else JLS_LAUNCH_GROUP(uint8_t, 16);
        }
        else
        {
            if (wide) JLS_LAUNCH_GROUP(uint16_t, 32); else JLS_LAUNCH_GROUP(uint8_t, 32);
        }
#undef JLS_LAUNCH_GROUP_NW
#undef JLS_LAUNCH_GROUP_NWK
#undef JLS_LAUNCH_GROUP
#undef JLS_LAUNCH_GROUP_N
    }
    }
    hip_check(hipGetLastError());
    // scans that did not end cleanly are decided by the exact decoder (error codes and byte counts of the reference):
    // they are gathered on the device and decoded by one launch
    auto* d_retry_count = static_cast<uint32_t*>(interval_arena(4).ensure(sizeof(uint32_t)));
    hip_check(hipMemsetAsync(d_retry_count, 0, sizeof(uint32_t), stream));
    auto* d_retry_index = static_cast<uint32_t*>(interval_arena(5).ensure(with_headroom(sizeof(uint32_t) * count)));
    auto* d_retry_descs = static_cast<ScanDesc*>(interval_arena(6).ensure(with_headroom(sizeof(ScanDesc) * count)));
    hipLaunchKernelGGL(gather_retries, dim3((count + 255) / 256), dim3(256), 0, stream, d_descs,
                       static_cast<const ScanResult*>(d_results), count, d_retry_descs, d_retry_index, d_retry_count);
    hip_check(hipGetLastError());
    uint32_t retries = 0;
    hip_check(hipMemcpyAsync(&retries, d_retry_count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    hip_check(hipStreamSynchronize(stream));
    if (retries == 0)
        return;
    g_exact_retry_scans.fetch_add(retries); // (charls_amd_engine_counters [9]: a valid stream that lands here is a lost speed path)
    auto* d_retry_results = static_cast<ScanResult*>(interval_arena(7).ensure(with_headroom(sizeof(ScanResult) * retries)));
    exact(d_retry_descs, d_retry_results, retries);
    hipLaunchKernelGGL(scatter_retries, dim3((retries + 255) / 256), dim3(256), 0, stream,
                       static_cast<const ScanResult*>(d_retry_results), static_cast<const uint32_t*>(d_retry_index), retries,
                       d_results);
    hip_check(hipGetLastError());
}
} // namespace

// ---------------------------------------------------------------------------------------------------------------
// Lossless pipeline orchestration.
namespace {

struct StageTimer
{
    hipEvent_t ev[10]{};
    int n = 0;
    hipStream_t s;
    explicit StageTimer(hipStream_t stream) : s(stream) {}
    StageTimer(const StageTimer&) = delete;
    StageTimer& operator=(const StageTimer&) = delete;
    StageTimer(StageTimer&& o) noexcept : n(o.n), s(o.s)
    {
        for (int i = 0; i < n; ++i)
            ev[i] = o.ev[i];
        o.n = 0;
    }
    ~StageTimer()
    {
        for (int i = 0; i < n; ++i)
            (void)hipEventDestroy(ev[i]);
    }
    void mark() { mark_on(s); }
    void mark_on(hipStream_t stream)
    {
        hip_check(hipEventCreate(&ev[n]));
        hip_check(hipEventRecord(ev[n], stream));
        ++n;
    }
    double between(int a, int b)
    {
        float ms = 0;
        hip_check(hipEventSynchronize(ev[b]));
        hip_check(hipEventElapsedTime(&ms, ev[a], ev[b]));
        return ms;
    }
};

// Events of a call, destroyed with it (also when a HIP error is raised half-way through a pass).
struct EventList
{
    std::vector<hipEvent_t> events;
    explicit EventList(size_t n = 0) { grow(n); }
    EventList(const EventList&) = delete;
    EventList& operator=(const EventList&) = delete;
    ~EventList()
    {
        for (hipEvent_t e : events)
            (void)hipEventDestroy(e);
    }
    void grow(size_t n)
    {
        events.reserve(n);
        while (events.size() < n)
        {
            hipEvent_t e{};
            hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            events.push_back(e);
        }
    }
    hipEvent_t operator[](size_t i) const { return events[i]; }
};

size_t align_up(size_t v, size_t a)
{
    return (v + a - 1) / a * a;
}

// Stage E in its block-parallel form (block_stuffing.hip) unless CHARLS_AMD_BLOCK_STUFFING=0 asks for stuff_scan.
bool block_stuffing_enabled()
{
    return knobs::get_or(knobs::kBlockStuffing, 1) != 0;
}

// Stage E in its speculative form (speculative_stuffing.hip) unless CHARLS_AMD_SPEC_STUFFING=0: for the passes whose stuffing
// nothing hides (the last pass of a call) and for streams so long that one wavefront per scan takes longer than the next
// pass's first stages (stuff_scan: 4.2 ns per byte, 99 ms for the 23.5 MB of a 4096 x 4096 RGB frame).
bool spec_stuffing_enabled()
{
    return knobs::get_or(knobs::kSpecStuffing, 1) != 0;
}
constexpr size_t kSpecStuffingAlwaysBytes = size_t{32} << 20; // destination capacity from which every pass takes the speculative form

constexpr uint32_t kBlockStuffingScans = 8; // scans per pass up to which stage E runs in its block-parallel form (it is for latency: every chunk is
                                            // walked from 16 entry states, 2.7 GB of L2 misses per frame when 64 frames do it at once)

DeviceBuffer& pipeline_arena()
{
    return areas().pipeline;
}

constexpr size_t kArenaReserve = size_t{8} << 30; // HBM left to the caller (collective buffers, ...) when the device is nearly full

// Work-area budget of this call: the workspace limit, capped by what is free now (plus what the arena already holds).
size_t arena_budget(size_t held)
{
    size_t free_bytes = 0, total_bytes = 0;
    if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess)
    {
        (void)hipGetLastError();
        return size_t{1} << 30;
    }
    const uint64_t configured = g_workspace_limit.load();
    const size_t limit = configured != 0 ? static_cast<size_t>(configured) : total_bytes / 4;
    const size_t reachable = free_bytes + held;
    const size_t usable = reachable > kArenaReserve ? reachable - kArenaReserve : reachable / 2;
    return std::min(limit, usable);
}

} // namespace

// ensure() that reports failure instead of raising (the arena is released, so the next attempt starts clean).
void* try_ensure(DeviceBuffer& buffer, size_t bytes) noexcept
{
    try
    {
        return buffer.ensure(bytes);
    }
    catch (const error&)
    {
        (void)hipGetLastError();
        return nullptr;
    }
}

namespace {


// ---------------------------------------------------------------------------------------------------------------
// Tile pipeline (tile_pipeline.hip, tile_pixel_mode.hip): every lossless scan the pipeline is eligible for.

// Work area of one scan: 8 B per sample of up to 8 bits (key / slot map 2, records 3, code words 3: 2-byte slots, of which the
// run starts -- at most every second sample -- take two), 10 B per wider sample (2 + 4 + 4), the (tiles + 1) x 367 piece table,
// the job states and the unstuffed stream.
struct TileLayout
{
    tile::TilePlan plan;
    size_t samples, lines, raw_bytes, max_jobs, max_run_jobs;
    uint32_t lines_per_tile, tiles, job_events, warm_events, run_job_events, run_warm_events, rare_warm_events;
    size_t off_keyinv, off_seg, off_total, off_base, off_jobfirst, off_planpart, off_rec, off_code, off_jobs, off_runjobs, off_bbase, off_raw,
        off_bits, off_status, off_stuff, bytes;
    TileLayout(const ScanDesc& d, size_t capacity_hint, uint32_t count)
    {
        plan = tile::plan_tiles(d);
        lines = plan.lines;
        samples = static_cast<size_t>(plan.samples);
        lines_per_tile = plan.lines_per_tile;
        tiles = plan.tiles;
        // Jobs: every job pays warm_events of warm-up, so long jobs are cheaper; but ONE frame needs thousands of lanes to
        // fill the chip.  Aim at a quarter of a million lanes per launch, between 1024 and 8192 events per job.
        const long long knob_job = knobs::get(knobs::kJobEvents), knob_warm = knobs::get(knobs::kWarmEvents);
        uint64_t job = 1024;
        while (job < 8192 && samples * count / (job * 2) >= (uint64_t{1} << 18))
            job *= 2;
        job_events = knob_job != knobs::kUnset ? static_cast<uint32_t>(std::max<long long>(32, knob_job) / 32 * 32) : static_cast<uint32_t>(job); // (whole rounds of the walkers: 32 two-byte slots)
        warm_events = knob_warm != knobs::kUnset ? static_cast<uint32_t>(std::max<long long>(0, knob_warm)) : 1024u;
        max_jobs = samples / job_events + pipe::kChains;
        // the run chain: jobs of 2048 run events with a warm-up of as many (a test frame has 55 000 run events); small batches
        // take smaller jobs -- ONE frame has the whole chip, and the walk of a job and its warm-up is what the frame waits for
        // (up to four 4096 x 4096 frames: 128 events behind a warm-up of 1024 -- 2.35 ms per frame where 256 / 2048 took 2.65;
        // 512 events of warm-up are enough for the test frame and 256 are not: every job walked again, 10.8 ms)
        const long long knob_run_job = knobs::get(knobs::kRunJobEvents), knob_run_warm = knobs::get(knobs::kRunWarmEvents);
        const uint64_t batch_samples = static_cast<uint64_t>(samples) * count;
        const uint32_t run_job_default = batch_samples <= (uint64_t{1} << 26) ? 128u : (batch_samples <= (uint64_t{1} << 29) ? 512u : 2048u);
        run_job_events = knob_run_job != knobs::kUnset ? static_cast<uint32_t>(std::max<long long>(32, knob_run_job) / 32 * 32) : run_job_default;
        run_warm_events = knob_run_warm != knobs::kUnset ? static_cast<uint32_t>(std::max<long long>(0, knob_run_warm)) : (run_job_default == 128u ? 1024u : 2048u);
        max_run_jobs = samples / run_job_events + 2; // (+ the entry of the totals)
        // the exact walk of the rarer run context: events of that type a lane walks before its segment of the list
        rare_warm_events = static_cast<uint32_t>(std::max<long long>(0, knobs::get_or(knobs::kRareWarmEvents, 512)));
        const size_t worst = worst_case_scan_bytes(plan.line_samples, static_cast<uint32_t>(lines), 1, d.bits_per_sample);
        raw_bytes = align_up((capacity_hint < worst ? capacity_hint : worst) + 64, 16);
        size_t o = 0;
        auto take = [&](size_t n) {
            const size_t at = o;
            o = align_up(o + n, 256);
            return at;
        };
        off_keyinv = take(samples * 2);
        off_seg = take(static_cast<size_t>(tiles + 1) * pipe::kChains * 4);
        off_total = take(pipe::kChains * 4);
        off_base = take(pipe::kChains * 4);
        off_jobfirst = take((pipe::kChains + 1) * 4);
        off_planpart = take(static_cast<size_t>(tile::kPlanGroups) * pipe::kChains * 4);
        // records and code words: 2-byte slots for samples of up to 8 bits (a run start takes two), 4-byte slots otherwise
        const uint32_t run_slots = d.bits_per_sample > 8 ? 1u : 2u;
        const size_t slot_bytes = d.bits_per_sample > 8 ? 4 : 2;
        const size_t slots = static_cast<size_t>(tile::slots_capacity(samples, run_slots, lines)) + tile::kSlack;
        off_rec = take(slots * slot_bytes);
        off_code = take(slots * slot_bytes);
        off_jobs = take(max_jobs * sizeof(tile::JobState));
        off_runjobs = take(max_run_jobs * sizeof(tile::RunJob));
        off_bbase = take(static_cast<size_t>(tiles) * 16); // look-back states and tile tails, 8 B each
        off_raw = take(raw_bytes);
        off_bits = take(16);
        off_status = take(8);
        off_stuff = take(std::max(block_stuffing_enabled() ? (raw_bytes / pipe::kStuffChunk + 1) * pipe::kStuffWords * 4 : 0,
                                  spec_stuffing_enabled() ? pipe::stuff_spec_table_words(raw_bytes, pipe::stuff_spec_geometry().chunk_bytes) * 4 : 0));
        bytes = o;
    }
};

// The tile kernels ask for more dynamic LDS than a kernel gets by default.  Function attributes are per DEVICE: they are set
// once for every device this process launches the kernels on.
void ensure_tile_attributes()
{
    static std::mutex guard;
    static uint64_t done[4] = {0, 0, 0, 0}; // bitmap of devices
    int device = 0;
    hip_check(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(guard);
    if (device >= 0 && device < 256 && (done[device >> 6] >> (device & 63) & 1u) != 0)
        return;
    const int lds = 160 * 1024;
    auto set = [&](const void* kernel) { hip_check(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); };
    set(reinterpret_cast<const void*>(tile::analyze_tiles<uint8_t, 0>));
    set(reinterpret_cast<const void*>(tile::analyze_tiles<uint16_t, 0>));
    set(reinterpret_cast<const void*>(tile::pack_tiles<uint8_t>));
    set(reinterpret_cast<const void*>(tile::pack_tiles<uint16_t>));
    set(reinterpret_cast<const void*>(tile::sort_tiles<uint8_t, 0>));
    set(reinterpret_cast<const void*>(tile::sort_tiles<uint16_t, 0>));
    set(reinterpret_cast<const void*>(tile::analyze_pixel_tiles<uint8_t>));
    set(reinterpret_cast<const void*>(tile::analyze_pixel_tiles<uint16_t>));
    set(reinterpret_cast<const void*>(tile::sort_pixel_tiles<uint8_t>));
    set(reinterpret_cast<const void*>(tile::sort_pixel_tiles<uint16_t>));
    if (device >= 0 && device < 256)
        done[device >> 6] |= uint64_t{1} << (device & 63);
}

constexpr size_t kCounterBytes = 256; // tile::kCounters words behind the work areas of a call

template <typename S>
void run_tile_pipeline(const ScanDesc& proto, ScanDesc* d_descs, ScanResult* d_results, uint32_t count, hipStream_t stream)
{
    // (the shared work areas of the host-pointer ABI stay allocated between calls, so they are kept moderate: more frames than
    // fit take more passes)
    const size_t budget = t_shared_areas != nullptr ? std::min(arena_budget(pipeline_arena().capacity()), shared_areas_keep_bytes())
                                                    : arena_budget(pipeline_arena().capacity());
    // the layout depends on the number of scans of a pass only through the job size: settle on it for a full pass
    TileLayout lay(proto, proto.stream_capacity, count);
    size_t per_scan = lay.bytes + 2 * (sizeof(tile::Work) + sizeof(pipe::Work));
    uint32_t resident = static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>(count, budget / per_scan)));
    if (resident < count)
    {
        lay = TileLayout(proto, proto.stream_capacity, resident);
        per_scan = lay.bytes + 2 * (sizeof(tile::Work) + sizeof(pipe::Work));
        resident = static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>(count, budget / per_scan)));
    }
    uint8_t* arena = nullptr;
    // a configured limit that does not even cover ONE work area is honoured: the one-wavefront-per-scan kernel needs none
    const bool over_limit = g_workspace_limit.load() != 0 && budget < per_scan;
    for (; !over_limit;)
    {
        // (the shared areas of the host-pointer ABI see batches of every size: they grow in powers of two, so that they stop
        // growing -- growing frees the old block, and hipFree waits for every kernel on the device)
        uint32_t room = resident;
        if (t_shared_areas != nullptr)
        {
            room = 1;
            while (room < resident)
                room *= 2;
            room = static_cast<uint32_t>(std::max<size_t>(resident, std::min<size_t>(room, budget / per_scan)));
        }
        arena = static_cast<uint8_t*>(try_ensure(pipeline_arena(), per_scan * room + kCounterBytes));
        if (arena != nullptr || resident == 1)
            break;
        resident = (resident + 1) / 2;
    }
    if (arena == nullptr)
    { // no work area to be had: the one-wavefront kernel needs none (counted: charls_amd_engine_counters)
        note_pipeline_fallback(count);
        launch_encode_serial(d_descs, d_results, count, stream);
        last_timings().count = 2;
        return;
    }
    const uint32_t passes = (count + resident - 1) / resident;
    const uint32_t per_pass = (count + passes - 1) / passes; // passes of equal size (a short last pass fills the chip badly)
    // what the speculative stages did (tile::Counter), summed over the scans and passes of this call
    auto* d_counters = reinterpret_cast<uint32_t*>(arena + per_scan * resident);
    hip_check(hipMemsetAsync(d_counters, 0, kCounterBytes, stream));
    std::vector<std::vector<tile::Work>> works(passes);
    std::vector<std::vector<pipe::Work>> stuff_works(passes);
    std::vector<StageTimer> timers;
    timers.reserve(passes);
    // As in run_pipeline: the stuffing stage of a pass runs on a side stream under the next pass's stages A - C.
    const bool overlap_stuffing = passes > 1;
    hipStream_t stuff_stream = stream;
    EventList packed, stuffed;
    pipeline_lanes().ensure();
    if (overlap_stuffing)
    {
        stuff_stream = pipeline_lanes().streams[0];
        packed.grow(passes);
        stuffed.grow(passes);
    }
    hipStream_t runs_stream = pipeline_lanes().streams[1];
    EventList sorted(passes), runs_coded(passes);
    ensure_tile_attributes();

    for (uint32_t pass = 0; pass < passes; ++pass)
    {
        const uint32_t first = pass * per_pass;
        const uint32_t n = std::min(per_pass, count - first);
        hipStream_t s = stream;
        const size_t copy = pass & 1u;
        uint8_t* tables = arena + lay.bytes * per_pass;
        auto* d_works = reinterpret_cast<tile::Work*>(tables) + copy * per_pass;
        auto* d_stuff = reinterpret_cast<pipe::Work*>(tables + 2 * sizeof(tile::Work) * per_pass) + copy * per_pass;
        works[pass].resize(n);
        stuff_works[pass].resize(n);
        for (uint32_t i = 0; i < n; ++i)
        {
            uint8_t* base = arena + lay.bytes * i;
            tile::Work& w = works[pass][i];
            w.keyinv = reinterpret_cast<uint16_t*>(base + lay.off_keyinv);
            w.seg = reinterpret_cast<uint32_t*>(base + lay.off_seg);
            w.plan_part = reinterpret_cast<uint32_t*>(base + lay.off_planpart);
            w.chain_total = reinterpret_cast<uint32_t*>(base + lay.off_total);
            w.chain_base = reinterpret_cast<uint32_t*>(base + lay.off_base);
            w.job_first = reinterpret_cast<uint32_t*>(base + lay.off_jobfirst);
            w.rec = reinterpret_cast<uint32_t*>(base + lay.off_rec);
            w.code = reinterpret_cast<uint32_t*>(base + lay.off_code);
            w.jobs = reinterpret_cast<tile::JobState*>(base + lay.off_jobs);
            w.run_jobs = reinterpret_cast<tile::RunJob*>(base + lay.off_runjobs);
            w.run_job_events = lay.run_job_events;
            w.run_warm_events = lay.run_warm_events;
            w.rare_warm_events = lay.rare_warm_events;
            w.blockbase = reinterpret_cast<uint64_t*>(base + lay.off_bbase);
            w.tile_tail = reinterpret_cast<uint64_t*>(base + lay.off_bbase) + lay.tiles;
            w.raw = reinterpret_cast<uint32_t*>(base + lay.off_raw);
            w.raw_words = lay.raw_bytes / 4;
            w.total_bits = reinterpret_cast<uint64_t*>(base + lay.off_bits) + copy; // (zeroed by plan_chains)
            w.status = reinterpret_cast<uint32_t*>(base + lay.off_status) + copy;
            w.counters = d_counters;
            w.lines_per_tile = lay.lines_per_tile;
            w.tiles = lay.tiles;
            w.job_events = lay.job_events;
            w.warm_events = lay.warm_events;
            w.segs_per_line = lay.plan.segs_per_line;
            w.seg_pixels = lay.plan.seg_pixels;
            w.tile_capacity = lay.plan.tile_capacity;
            w.run_slots = tile::run_slots_of<S>();
            pipe::Work& sw = stuff_works[pass][i];
            std::memset(&sw, 0, sizeof sw);
            sw.raw = w.raw;
            sw.raw_words = w.raw_words;
            sw.total_bits = w.total_bits;
            sw.status = w.status;
            sw.stuff_tables = reinterpret_cast<uint32_t*>(base + lay.off_stuff);
        }
        hip_check(hipMemcpyAsync(d_works, works[pass].data(), sizeof(tile::Work) * n, hipMemcpyHostToDevice, s));
        hip_check(hipMemcpyAsync(d_stuff, stuff_works[pass].data(), sizeof(pipe::Work) * n, hipMemcpyHostToDevice, s));

        const ScanDesc* descs = d_descs + first;
        const uint32_t tiles_grid = 8 * ((lay.tiles + 7) / 8);
        const tile::TilePlan& plan = lay.plan;
        const bool pixel_mode = plan.mode == 2;
        const size_t lds_a = pixel_mode ? tile::analyze_pixel_lds_bytes(plan.lines_per_tile, plan.step, plan.max_pixels, plan.nc, sizeof(S), plan.tile_capacity)
                                        : tile::analyze_lds_bytes(proto.width, lay.lines_per_tile, sizeof(S), proto.interleave_mode);
        const size_t lds_b = pixel_mode ? tile::sort_pixel_lds_bytes(plan.lines_per_tile, plan.step, plan.max_pixels, plan.nc, sizeof(S), plan.tile_capacity)
                                        : tile::sort_lds_bytes(proto.width, lay.lines_per_tile, sizeof(S), proto.interleave_mode);
        timers.emplace_back(s);
        StageTimer& t = timers.back();
        t.mark();
        if (pixel_mode)
            hipLaunchKernelGGL((tile::analyze_pixel_tiles<S>), dim3(tiles_grid, n), dim3(tile::kThreads), lds_a, s, descs, d_works);
        else
            hipLaunchKernelGGL((tile::analyze_tiles<S, 0>), dim3(tiles_grid, n), dim3(tile::kThreads), lds_a, s, descs, d_works);
        t.mark();
        hipLaunchKernelGGL(tile::sum_chains, dim3(tile::kPlanGroups, n), dim3(tile::kPlanThreads), 0, s, descs, d_works);
        hipLaunchKernelGGL(tile::plan_chains, dim3(n), dim3(tile::kPlanThreads), 0, s, descs, d_works);
        hipLaunchKernelGGL(tile::apply_chains, dim3(tile::kPlanGroups, n), dim3(tile::kPlanThreads), 0, s, descs, d_works);
        if (pixel_mode)
            hipLaunchKernelGGL((tile::sort_pixel_tiles<S>), dim3(tiles_grid, n), dim3(tile::kThreads), lds_b, s, descs, d_works);
        else
            hipLaunchKernelGGL((tile::sort_tiles<S, 0>), dim3(tiles_grid, n), dim3(tile::kThreads), lds_b, s, descs, d_works);
        t.mark();
        // The run chain on a side stream under the walkers of the regular chains: it touches the run chain and the
        // interruption chain only, they every other chain.
        hip_check(hipEventRecord(sorted[pass], s));
        hip_check(hipStreamWaitEvent(runs_stream, sorted[pass], 0));
        {
            const uint32_t run_jobs = static_cast<uint32_t>(lay.max_run_jobs);
            const dim3 lanes((static_cast<uint64_t>(run_jobs) * n + 63) / 64);
            const dim3 count_grid(std::min<uint32_t>(run_jobs, std::max<uint32_t>(32, 4096 / n)), n);
            if (pixel_mode)
                hipLaunchKernelGGL((tile::count_runs<S, 1>), count_grid, dim3(64), 0, runs_stream, d_works, plan.nc);
            else
                hipLaunchKernelGGL((tile::count_runs<S, 0>), count_grid, dim3(64), 0, runs_stream, d_works, 1u);
            hipLaunchKernelGGL(tile::scan_runs, dim3(n), dim3(64), 0, runs_stream, d_works);
            if (proto.interleave_mode != 2)
            { // the context of the rarer interruption type, exactly (a sample-interleaved scan has one type only)
                if (pixel_mode)
                    hipLaunchKernelGGL((tile::compact_rare_runs<S, 1>), count_grid, dim3(64), 0, runs_stream, d_works);
                else
                    hipLaunchKernelGGL((tile::compact_rare_runs<S, 0>), count_grid, dim3(64), 0, runs_stream, d_works);
            }
            const uint32_t rare_blocks = proto.interleave_mode != 2 ? n : 0u; // (the walk of the rarer context rides with the warm-ups: a wavefront per scan)
#define JLS_RUN_CHAIN(ILV, FMT)                                                                                          \
    do                                                                                                                   \
    {                                                                                                                    \
        hipLaunchKernelGGL((tile::warm_run_jobs<S, ILV, FMT>), dim3(rare_blocks + lanes.x), dim3(64), 0, runs_stream, descs, d_works, n, rare_blocks); \
        hipLaunchKernelGGL((tile::walk_run_jobs<S, ILV, FMT>), lanes, dim3(64), 0, runs_stream, descs, d_works, n);      \
        hipLaunchKernelGGL((tile::settle_runs<S, ILV, FMT>), dim3(n), dim3(64), 0, runs_stream, descs, d_works, n);  \
    } while (0)
            if (!pixel_mode)
                JLS_RUN_CHAIN(0, 0);
            else if (proto.interleave_mode == 2)
                JLS_RUN_CHAIN(2, 1);
            else if (proto.interleave_mode == 1)
                JLS_RUN_CHAIN(1, 1);
            else
                JLS_RUN_CHAIN(0, 1);
#undef JLS_RUN_CHAIN
        }
        hip_check(hipEventRecord(runs_coded[pass], runs_stream));
        hipLaunchKernelGGL((tile::walk_jobs<S>), dim3(static_cast<uint32_t>((lay.max_jobs + 63) / 64), n), dim3(64), 0, s, descs, d_works);
        hipLaunchKernelGGL((tile::settle_chains<S>), dim3(pipe::kChains, n), dim3(64), 0, s, descs, d_works, n);
        hip_check(hipStreamWaitEvent(s, runs_coded[pass], 0));
        t.mark();
        if (overlap_stuffing && pass > 0)
            hip_check(hipStreamWaitEvent(s, stuffed[pass - 1], 0)); // the raw bits of the pass before have been read
        hipLaunchKernelGGL(tile::clear_pack_state, dim3(1, n), dim3(256), 0, s, d_works,
                           static_cast<uint32_t>(static_cast<size_t>(lay.tiles) * 16));
        hipLaunchKernelGGL((tile::pack_tiles<S>), dim3(lay.tiles, n), dim3(tile::pack_threads_for(lay.plan.tile_capacity)), tile::pack_lds_bytes(lay.plan.tile_capacity, proto.bits_per_sample), s,
                           descs, d_works);
        t.mark();
        if (overlap_stuffing)
        {
            hip_check(hipEventRecord(packed[pass], s));
            hip_check(hipStreamWaitEvent(stuff_stream, packed[pass], 0));
        }
        t.mark_on(stuff_stream);
        // Block-parallel stuffing is what ONE frame needs (a lane per KB of stream instead of a wavefront per frame); it
        // walks every chunk from 16 entry states, so with a wavefront's worth of frames per SIMD pair stuff_scan is cheaper.
        if (block_stuffing_enabled() && n <= kBlockStuffingScans)
        {
            const uint32_t chunk_waves = static_cast<uint32_t>((lay.raw_bytes / pipe::kStuffChunk + 1 + 63) / 64);
            const uint32_t survey_blocks = pipe::stuff_survey_blocks(lay.raw_bytes);
            hipLaunchKernelGGL(pipe::stuff_survey, dim3(survey_blocks, n), dim3(64), 0, stuff_stream, d_stuff);
            hipLaunchKernelGGL(pipe::stuff_resolve, dim3(n), dim3(pipe::kStuffResolveThreads), 0, stuff_stream, d_stuff);
            hipLaunchKernelGGL(pipe::stuff_emit, dim3(chunk_waves, n), dim3(64), 0, stuff_stream, descs, d_stuff, d_results + first);
        }
        else if (spec_stuffing_enabled() && (pass + 1 == passes || lay.raw_bytes >= kSpecStuffingAlwaysBytes))
        { // a wavefront per 64 KB of raw stream, from guessed entry states
            const pipe::SpecGeometry spec = pipe::stuff_spec_geometry();
            const uint32_t waves = static_cast<uint32_t>(lay.raw_bytes / spec.chunk_bytes + 1);
            hipLaunchKernelGGL(pipe::stuff_spec_survey, dim3(waves, n), dim3(64), 0, stuff_stream, d_stuff, spec.chunk_bytes, spec.warm_bytes);
            hipLaunchKernelGGL(pipe::stuff_spec_resolve, dim3(n), dim3(64), 0, stuff_stream, d_stuff, spec.chunk_bytes);
            hipLaunchKernelGGL(pipe::stuff_spec_emit, dim3(waves, n), dim3(64), 0, stuff_stream, descs, d_stuff, d_results + first, spec.chunk_bytes);
        }
        else
            hipLaunchKernelGGL(pipe::stuff_scan, dim3(n), dim3(64), 0, stuff_stream, descs, d_stuff, d_results + first);
        t.mark_on(stuff_stream);
        if (overlap_stuffing)
            hip_check(hipEventRecord(stuffed[pass], stuff_stream));
        hip_check(hipGetLastError());
    }
    if (overlap_stuffing)
        hip_check(hipStreamWaitEvent(stream, stuffed[passes - 1], 0));
    static_assert(tile::kCounters == tile_counter_count, "runtime.h: tile_counter_count");
    uint32_t counters[tile::kCounters] = {};
    hip_check(hipMemcpyAsync(counters, d_counters, sizeof counters, hipMemcpyDeviceToHost, stream));
    hip_check(hipStreamSynchronize(stream)); // the host copies of the work descriptors and the timers go out of scope
    for (uint32_t i = 0; i < tile::kCounters; ++i)
        g_speculation[i].fetch_add(counters[i]);
#ifdef JLS_PHASE_CLOCKS
    { // debug build (tools/phase_clocks.sh): the clocks the wavefronts of the tile kernels spent in each of their phases
        unsigned long long all[(kCounterBytes - 32) / 8] = {};
        hip_check(hipMemcpy(all, reinterpret_cast<const uint8_t*>(d_counters) + 32, sizeof all, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "phase_clocks");
        for (unsigned long long v : all)
            std::fprintf(stderr, " %llu", v);
        std::fprintf(stderr, "\n");
    }
#endif
    Timings& tm = last_timings();
    double stage_ms[5] = {0, 0, 0, 0, 0};
    for (StageTimer& t : timers)
    {
        for (int i = 0; i < 4; ++i)
            stage_ms[i] += t.between(i, i + 1);
        stage_ms[4] += t.between(5, 6);
    }
    for (int i = 0; i < 5; ++i)
        tm.values[2 + i] = stage_ms[i];
    tm.count = 7;
}

} // namespace

bool pipeline_eligible(const ScanDesc& d) noexcept
{
    if (encode_engine() == EncodeEngine::serial)
        return false;
    if (d.near_lossless != 0)
        return false; // the template then holds reconstructed samples: nothing is known ahead of the chain
    const bool planar = d.interleave_mode == 0 && d.components == 1 && d.color_transformation == 0;
    const bool interleaved = d.interleave_mode != 0 && d.components >= 2 && d.components <= 4 &&
                             (d.color_transformation == 0 || d.components == 3);
    if (!planar && !interleaved)
        return false;
    const uint64_t samples = static_cast<uint64_t>(d.width) * d.height * static_cast<uint64_t>(d.components);
    if (samples >= (uint64_t{1} << 31))
        return false;
    if (d.reset == 0 && samples >= (uint64_t{1} << 23))
        return false; // N never halves (RESET = 256 m through the reference's uint8): it would outgrow the 24-bit multiplies of the chain stage
    if (d.bits_per_sample > 8 && ((reinterpret_cast<uintptr_t>(d.pixels) | d.pixel_stride) & 1u) != 0)
        return false;
    return true;
}

namespace {
// Restart-interval encode (extension): every interval is coded as a scan of its own into a private buffer, then the
// pieces are joined with RSTm markers (restart_intervals.hip).  All `count` scans share proto's geometry and interval.
void launch_encode_intervals(const ScanDesc& proto, ScanDesc* d_descs, ScanResult* d_results, uint32_t count,
                             hipStream_t stream)
{
    const uint32_t lines = proto.restart_interval;
    const uint32_t intervals = (proto.height + lines - 1) / lines;
    ScanDesc sub_proto = proto;
    sub_proto.height = lines;
    sub_proto.restart_interval = 0;
    const size_t worst = worst_case_scan_bytes(proto.width, lines, proto.interleave_mode == 0 ? 1 : proto.components,
                                               proto.bits_per_sample);
    const bool needs_scratch = !pipeline_eligible(sub_proto);
    const size_t scratch_samples = needs_scratch ? line_scratch_samples(proto.width, proto.interleave_mode, proto.components) : 0;
    // First attempt: twice the interval's share of the destination (a destination sized by
    // charls_jpegls_encoder_get_estimated_destination_size leaves every interval more than it can use).  An interval
    // that does not fit its private buffer makes its group repeat with worst-case buffers, so the verdict
    // destination_too_small depends on the joined size only.
    const size_t first_capacity = std::min(worst, align_up(2 * (static_cast<size_t>(proto.stream_capacity) / intervals) + 4096, 256));
    constexpr size_t kBufferBudget = size_t{8} << 30; // private interval buffers in flight
    for (uint32_t first = 0; first < count;)
    {
        size_t capacity = first_capacity;
        uint32_t group = 0;
        for (int attempt = 0; attempt < 2; ++attempt)
        {
            group = static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>(count - first, kBufferBudget / (capacity * intervals))));
            const size_t subs_n = static_cast<size_t>(group) * intervals;
            auto* d_subs = static_cast<ScanDesc*>(interval_arena(2).ensure(sizeof(ScanDesc) * subs_n));
            auto* d_sub_results = static_cast<ScanResult*>(interval_arena(3).ensure(sizeof(ScanResult) * subs_n));
            auto* d_offsets = static_cast<uint64_t*>(interval_arena(0).ensure(sizeof(uint64_t) * subs_n));
            auto* d_scratch = needs_scratch
                                  ? static_cast<uint16_t*>(interval_arena(5).ensure(sizeof(uint16_t) * scratch_samples * subs_n))
                                  : nullptr;
            auto* d_buffers = static_cast<uint8_t*>(interval_arena(4).ensure(capacity * subs_n));
            hipLaunchKernelGGL(interval::build_encode_intervals, dim3(intervals, group), dim3(1), 0, stream, d_descs + first,
                               intervals, d_buffers, static_cast<uint64_t>(capacity), d_scratch,
                               static_cast<uint64_t>(scratch_samples), d_subs);
            hip_check(hipGetLastError());
            sub_proto.stream_capacity = capacity;
            launch_encode(sub_proto, d_subs, d_sub_results, static_cast<uint32_t>(subs_n), stream);
            hipLaunchKernelGGL(interval::plan_join, dim3((group + 63) / 64), dim3(64), 0, stream, d_descs + first, intervals,
                               d_sub_results, d_offsets, d_results + first, group);
            hipLaunchKernelGGL(interval::join_intervals, dim3(intervals, group), dim3(256), 0, stream, d_descs + first, d_subs,
                               intervals, d_sub_results, d_offsets, d_results + first);
            hip_check(hipGetLastError());
            std::vector<ScanResult> results(group);
            hip_check(hipMemcpyAsync(results.data(), d_results + first, sizeof(ScanResult) * group, hipMemcpyDeviceToHost, stream));
            hip_check(hipStreamSynchronize(stream)); // the private buffers are reused by the next group
            bool again = false;
            for (const ScanResult& r : results)
                again = again || (r.flags & interval::kIntervalRetry) != 0;
            if (!again || capacity == worst)
                break;
            capacity = worst;
        }
        first += group;
    }
}
} // namespace

namespace {
// Lanes per scan of the group encoder (scan_group_encode.hip) for scans the parallel pipeline cannot take -- near-lossless
// single-component, sample-interleaved and line-interleaved scans; 0 = the one-lane kernel (lines that do not fit LDS).
int group_encode_lanes(const ScanDesc& d, uint32_t count)
{
    const bool shape = (d.interleave_mode != 0 && d.components >= 2 && d.components <= 4) || (d.interleave_mode == 0 && d.components == 1);
    if (!shape || encode_engine() == EncodeEngine::serial)
        return 0;
    int best = 0;
    for (int lanes = 64; lanes >= 8; lanes /= 2)
    {
        const uint32_t per_wave = 64u / static_cast<uint32_t>(lanes);
        if (group_encode_lds_bytes(d, per_wave) > kGroupDecodeLds)
            break;
        best = lanes;
        if ((count + per_wave - 1) / per_wave <= kPixelWaves) // (two one-wavefront workgroups per CU find a SIMD each, see decode_group_plan)
            break;
    }
    return best;
}

} // namespace

void launch_encode(const ScanDesc& proto, ScanDesc* d_descs, ScanResult* d_results, uint32_t count, hipStream_t stream)
{
    if (count == 0)
        return;
    if (proto.restart_interval != 0 && proto.restart_interval < proto.height)
    {
        launch_encode_intervals(proto, d_descs, d_results, count, stream);
        return;
    }
    if (!pipeline_eligible(proto))
    {
        if (encode_engine() == EncodeEngine::pipeline)
            raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT);
        const int lanes = group_encode_lanes(proto, count);
        if (lanes != 0)
            launch_encode_group(proto, lanes, d_descs, d_results, count, stream);
        else
            launch_encode_serial(d_descs, d_results, count, stream);
        return;
    }
    if (proto.bits_per_sample > 8)
        run_tile_pipeline<uint16_t>(proto, d_descs, d_results, count, stream);
    else
        run_tile_pipeline<uint8_t>(proto, d_descs, d_results, count, stream);
    // Scans whose destination is within 3 bytes of their size: the reference's verdict depends on its flush history
    // (src/scan_encoder.hpp:117-120), so those few are re-coded by the kernel that restates that history.
    std::vector<ScanResult> results(count);
    hip_check(hipMemcpyAsync(results.data(), d_results, sizeof(ScanResult) * count, hipMemcpyDeviceToHost, stream));
    hip_check(hipStreamSynchronize(stream));
    for (uint32_t i = 0; i < count; ++i)
        if (results[i].errc == kOk && (results[i].flags & 2u) != 0)
            launch_encode_serial(d_descs + i, d_results + i, 1, stream);
}

// Private stream buffers of the batch encoder's planar path (the component scans of a group of frames are coded into them
// and then put in place): gigabytes, so they are kept between calls like the pipeline's work areas -- allocating and
// freeing 8 GiB per call cost three times the coding of 256 4096 x 4096 RGB frames.
DeviceBuffer& plane_arena()
{
    return areas().plane;
}

size_t work_area_budget() noexcept
{
    return arena_budget(areas().bytes());
}

size_t shared_areas_keep_bytes() noexcept
{
    size_t free_bytes = 0, total_bytes = 0;
    if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess)
    {
        (void)hipGetLastError();
        return size_t{1} << 30;
    }
    const uint64_t configured = g_workspace_limit.load();
    const size_t eighth = total_bytes / 8;
    return configured != 0 ? std::min<size_t>(static_cast<size_t>(configured), eighth) : eighth;
}

// ---- the work areas of the host-pointer ABI: one set per device, used by one merged launch at a time
namespace {
struct SharedAreas
{
    std::mutex turn;
    WorkAreas areas;
    std::atomic<size_t> held{0};
};
std::atomic<SharedAreas*> g_shared[kMaxDevices]{};
SharedAreas& shared_areas(int device)
{
    std::atomic<SharedAreas*>& slot = g_shared[device >= 0 && device < kMaxDevices ? device : 0];
    SharedAreas* s = slot.load(std::memory_order_acquire);
    if (s == nullptr)
    {
        auto* fresh = new SharedAreas; // never destroyed: at process exit the HIP runtime may already be gone
        if (slot.compare_exchange_strong(s, fresh, std::memory_order_acq_rel))
            s = fresh;
        else
            delete fresh;
    }
    return *s;
}
} // namespace

SharedAreasScope::SharedAreasScope()
{
    hip_check(hipGetDevice(&device_));
    SharedAreas& s = shared_areas(device_);
    s.turn.lock();
    t_shared_areas = &s.areas;
}

SharedAreasScope::~SharedAreasScope()
{
    SharedAreas& s = shared_areas(device_);
    s.held.store(s.areas.bytes(), std::memory_order_relaxed);
    t_shared_areas = nullptr;
    s.turn.unlock();
}

size_t SharedAreasScope::bytes() const noexcept
{
    return shared_areas(device_).areas.bytes();
}

void SharedAreasScope::release() noexcept
{
    shared_areas(device_).areas.release();
}

size_t thread_work_area_bytes() noexcept
{
    return areas().bytes();
}

void release_thread_work_areas() noexcept
{
    areas().release();
}

void release_shared_work_areas() noexcept
{
    if (t_shared_areas != nullptr)
        return; // (called from inside a merged launch)
    int current = 0;
    const bool have_device = hipGetDevice(&current) == hipSuccess;
    for (int device = 0; device < kMaxDevices; ++device)
    {
        SharedAreas* s = g_shared[device].load(std::memory_order_acquire);
        if (s == nullptr)
            continue;
        std::lock_guard<std::mutex> turn(s->turn); // (a merged launch that is running finishes first)
        (void)hipSetDevice(device);
        s->areas.lanes.destroy();
        s->areas.release();
        s->held.store(0, std::memory_order_relaxed);
    }
    if (have_device)
        (void)hipSetDevice(current);
}

size_t shared_work_area_bytes() noexcept
{
    size_t total = 0;
    for (int device = 0; device < kMaxDevices; ++device)
        if (SharedAreas* s = g_shared[device].load(std::memory_order_acquire))
            total += s->held.load(std::memory_order_relaxed);
    return total;
}

void release_work_areas() noexcept
{
    areas().release();
    if (t_shared_areas != nullptr)
        return; // (called from inside a merged launch: its areas are the ones just released)
    release_shared_work_areas();
    reap_deferred_frees(); // (blocks that were set aside while a decoder launch of the host-pointer ABI ran)
}

size_t work_area_bytes() noexcept
{
    size_t total = areas().bytes();
    if (t_shared_areas == nullptr)
        for (int device = 0; device < kMaxDevices; ++device)
            if (SharedAreas* s = g_shared[device].load(std::memory_order_acquire))
                total += s->held.load(std::memory_order_relaxed);
    return total;
}


static_assert(sizeof(FrameCursorPod) == sizeof(FrameCursor), "cursor layout");

void launch_place_prologue(uint8_t* slots, uint64_t slot_pitch, const uint8_t* prologue, uint32_t prologue_size,
                           FrameCursorPod* cursors, uint32_t frames, hipStream_t stream)
{
    hipLaunchKernelGGL(place_prologue, dim3(frames), dim3(64), 0, stream, slots, slot_pitch, prologue, prologue_size,
                       reinterpret_cast<FrameCursor*>(cursors), frames);
    hip_check(hipGetLastError());
}

void launch_place_scan_header(uint8_t* slots, uint64_t slot_pitch, const uint8_t* header, uint32_t header_size,
                              FrameCursorPod* cursors, ScanDesc* descs, uint32_t frames, hipStream_t stream)
{
    hipLaunchKernelGGL(place_scan_header, dim3((frames + 63) / 64), dim3(64), 0, stream, slots, slot_pitch, header,
                       header_size, reinterpret_cast<FrameCursor*>(cursors), descs, frames);
    hip_check(hipGetLastError());
}

void launch_advance_cursor(FrameCursorPod* cursors, const ScanResult* results, uint32_t header_size, uint32_t frames,
                           hipStream_t stream)
{
    hipLaunchKernelGGL(advance_cursor, dim3((frames + 63) / 64), dim3(64), 0, stream,
                       reinterpret_cast<FrameCursor*>(cursors), results, header_size, frames);
    hip_check(hipGetLastError());
}

void launch_place_plane_scans(uint8_t* slots, uint64_t slot_pitch, const uint8_t* headers, uint32_t header_size, uint32_t rounds,
                              const uint8_t* private_streams, uint64_t capacity, const ScanResult* results, FrameCursorPod* cursors,
                              uint32_t* redo, uint32_t frames, hipStream_t stream)
{
    hipLaunchKernelGGL(place_plane_scans, dim3(frames * rounds, kPlaceShares), dim3(256), 0, stream, slots, slot_pitch, headers, header_size,
                       rounds, private_streams, capacity, results, reinterpret_cast<const FrameCursor*>(cursors));
    hipLaunchKernelGGL(advance_plane_cursors, dim3((frames + 63) / 64), dim3(64), 0, stream, slot_pitch, header_size, rounds, results,
                       reinterpret_cast<FrameCursor*>(cursors), redo, frames);
    hip_check(hipGetLastError());
}

void launch_find_scan_end(const uint8_t* slots, const MarkerSearch* searches, unsigned long long* found, uint32_t count, hipStream_t stream)
{
    hipLaunchKernelGGL(find_scan_end, dim3(count), dim3(256), 0, stream, slots, searches, found);
    hip_check(hipGetLastError());
}

void launch_place_epilogue(uint8_t* slots, uint64_t slot_pitch, FrameCursorPod* cursors, bool even_size, uint64_t* sizes,
                           uint32_t* errcs, uint32_t frames, hipStream_t stream)
{
    hipLaunchKernelGGL(place_epilogue, dim3((frames + 63) / 64), dim3(64), 0, stream, slots, slot_pitch,
                       reinterpret_cast<FrameCursor*>(cursors), even_size ? 1u : 0u, sizes, errcs, frames);
    hip_check(hipGetLastError());
}

} // namespace jls::dev
