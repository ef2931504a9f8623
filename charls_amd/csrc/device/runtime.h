// runtime.h -- host-side HIP plumbing of the engine: device discovery, per-handle stream + device arenas, kernel launches.
// Compiled by hipcc together with the kernels; the C-ABI facade (host/*.cpp) only sees this interface.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "../host/common.h"
#include "scan_types.h"

namespace jls::dev {

// CHARLS_JPEGLS_ERRC_SUCCESS when a gfx950 device is usable by this process (lazy, thread-safe).
charls_jpegls_errc device_status() noexcept;
void require_device(); // raises CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE

inline void hip_check(hipError_t e)
{
    if (e != hipSuccess)
        raise(e == hipErrorOutOfMemory ? CHARLS_JPEGLS_ERRC_NOT_ENOUGH_MEMORY : CHARLS_AMD_ERRC_DEVICE_FAILURE);
}

// Grow-only device allocation owned by a handle.
class DeviceBuffer
{
public:
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { release(); }
    void* ensure(size_t bytes);
    void release() noexcept;
    template <typename T>
    T* as() const noexcept
    {
        return static_cast<T*>(ptr_);
    }
    size_t capacity() const noexcept { return cap_; }

private:
    void* ptr_{};
    size_t cap_{};
    int device_{-1}; // the device the memory lives on
};

// Pinned host allocation (descriptor upload / result download without staging copies).
class PinnedBuffer
{
public:
    PinnedBuffer() = default;
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    ~PinnedBuffer();
    void* ensure(size_t bytes);
    template <typename T>
    T* as() const noexcept
    {
        return static_cast<T*>(ptr_);
    }

private:
    void* ptr_{};
    size_t cap_{};
};

// A launch that runs for seconds is in flight (a decoder launch of the host-pointer ABI): until it ends, DeviceBuffers that
// give a block up set it aside instead of calling hipFree (which would wait for that launch); reap_deferred_frees() frees what
// was set aside once no such launch is running.
// (The counter is per device, and so is what is set aside: at most 4 GiB per device -- a block beyond that is freed on the spot --
// and an allocation that fails frees everything that was set aside on its device, waiting for the running launch, and retries.)
void long_kernel_begins() noexcept;
void long_kernel_ends() noexcept;
void reap_deferred_frees() noexcept;
uint64_t deferred_free_bytes() noexcept; // what is set aside right now, all devices

enum class EncodeEngine : int32_t
{
    automatic = 0,
    serial = 1,
    pipeline = 2
};
EncodeEngine encode_engine() noexcept;
void set_encode_engine(EncodeEngine e) noexcept;

// Samples of line window needed by one scan: 2 * planes * (width + 2).
inline size_t line_scratch_samples(uint32_t width, int32_t interleave_mode, int32_t components)
{
    return 2 * static_cast<size_t>(interleave_mode == 0 ? 1 : components) * (static_cast<size_t>(width) + 2);
}

// Upper bound of the entropy-coded bytes of one scan: every sample costs at most LIMIT bits (+1 stuffed bit per 8),
// run-length codes at most 32 + 16 bits per run start; plus the end-of-scan padding.
inline size_t worst_case_scan_bytes(uint32_t width, uint32_t height, int32_t components, int32_t bits)
{
    const size_t limit_bits = 2 * static_cast<size_t>(bits + (bits > 8 ? bits : 8));
    const size_t samples = static_cast<size_t>(width) * height * static_cast<size_t>(components);
    const size_t bits_total = samples * (limit_bits + 48);
    return bits_total / 7 + 64;
}

// Kernel launches (descs/results are DEVICE pointers).
void launch_encode_serial(const ScanDesc* d_descs, ScanResult* d_results, uint32_t count, hipStream_t stream);
void launch_decode_serial(const ScanDesc* d_descs, ScanResult* d_results, uint32_t count, hipStream_t stream);

// Decoder dispatch.  All `count` scans must share the geometry / coding mode of `proto` (a HOST copy of one of them):
// the wave-uniform LDS decoder (scan_wave_decode.hip) when the scan qualifies, the global-memory fallback otherwise.
// The key groups scans that can share one launch.
uint64_t decode_launch_key(const ScanDesc& d) noexcept;
void launch_decode(const ScanDesc& proto, const ScanDesc* d_descs, ScanResult* d_results, uint32_t count,
                   hipStream_t stream);

// Encoder dispatch.  All `count` scans share the geometry / coding mode of `proto` (HOST copy of one of them; its
// stream_capacity is an upper bound of every scan's capacity).  Lossless single-component scans go through the
// parallel pipeline (tile_pipeline.hip) unless the engine is forced to serial; everything else, and the rare scan
// whose destination is within 3 bytes of its output size, runs the exact one-wavefront-per-scan kernel.
bool pipeline_eligible(const ScanDesc& proto) noexcept;
void launch_encode(const ScanDesc& proto, ScanDesc* d_descs, ScanResult* d_results, uint32_t count, hipStream_t stream);

// Container placement kernels (container_kernels.hip); all pointers are DEVICE pointers.
struct FrameCursorPod
{
    uint64_t offset;
    uint32_t errc;
    uint32_t pad;
};
void launch_place_prologue(uint8_t* slots, uint64_t slot_pitch, const uint8_t* prologue, uint32_t prologue_size,
                           FrameCursorPod* cursors, uint32_t frames, hipStream_t stream);
void launch_place_scan_header(uint8_t* slots, uint64_t slot_pitch, const uint8_t* header, uint32_t header_size,
                              FrameCursorPod* cursors, ScanDesc* descs, uint32_t frames, hipStream_t stream);
void launch_advance_cursor(FrameCursorPod* cursors, const ScanResult* results, uint32_t header_size, uint32_t frames,
                           hipStream_t stream);
// Where the entropy-coded segments that start at searches[k].from end (the first marker that is not a restart marker;
// kNoMarker: none before searches[k].end): the batch decoder's way from one component scan of a planar frame to the next.
void launch_find_scan_end(const uint8_t* slots, const MarkerSearch* searches, unsigned long long* found, uint32_t count, hipStream_t stream);
void launch_place_plane_scans(uint8_t* slots, uint64_t slot_pitch, const uint8_t* headers, uint32_t header_size, uint32_t rounds,
                              const uint8_t* private_streams, uint64_t capacity, const ScanResult* results, FrameCursorPod* cursors,
                              uint32_t* redo, uint32_t frames, hipStream_t stream);
void launch_place_epilogue(uint8_t* slots, uint64_t slot_pitch, FrameCursorPod* cursors, bool even_size, uint64_t* sizes,
                           uint32_t* errcs, uint32_t frames, hipStream_t stream);

// HBM the library keeps for its own work areas (the lossless pipeline's per-scan work area, restart-interval buffers).
// The limit is process-wide (0 = a quarter of the device's memory); the areas themselves belong to the calling thread,
// grow on demand up to the limit and stay allocated between calls until released.
void set_workspace_limit(uint64_t bytes) noexcept;
uint64_t workspace_limit() noexcept;          // as set (0: the default rule)
void release_shared_work_areas() noexcept;    // the shared sets of the host-pointer ABI's merged launches, all devices (a launch that runs finishes first)
size_t shared_work_area_bytes() noexcept;     // what they hold
DeviceBuffer& plane_arena(); // the calling thread's private stream buffers of the planar batch encoder (a work area)
void* try_ensure(DeviceBuffer& buffer, size_t bytes) noexcept; // ensure() that reports failure (nullptr; the buffer is released) instead of raising
void release_work_areas() noexcept;
size_t work_area_bytes() noexcept;
size_t thread_work_area_bytes() noexcept;   // the calling thread's own areas (not the shared set of the host-pointer ABI)
void release_thread_work_areas() noexcept;
size_t work_area_budget() noexcept; // what the calling thread's work areas may grow to right now (limit, free memory)
// Scans the lossless pipeline was eligible for that were coded by the one-wavefront kernel because no work area could be had.
uint64_t pipeline_fallback_scans() noexcept;
uint64_t exact_retry_scans() noexcept; // scans the speed-path decoders handed to the exact decoder since the library was loaded

// A size with room to spare for buffers that grow with the number of scans of a launch: at least 64 KB, then the next power of
// two.  (Growing a DeviceBuffer frees the old block, and hipFree waits for every kernel that is running on the device.)
inline size_t with_headroom(size_t bytes) noexcept
{
    size_t v = size_t{64} << 10;
    while (v < bytes)
        v <<= 1;
    return v;
}
// What the shared work areas of the host-pointer ABI may keep between calls: an eighth of the device's memory, within the
// workspace limit.
size_t shared_areas_keep_bytes() noexcept;

// The host-pointer ABI's merged launches (host/scan_engine.cpp) run on ONE set of work areas per device, shared by all
// calling threads: while a scope is alive the calling thread's launches use that set (and nobody else does).
class SharedAreasScope
{
public:
    SharedAreasScope();
    ~SharedAreasScope();
    SharedAreasScope(const SharedAreasScope&) = delete;
    SharedAreasScope& operator=(const SharedAreasScope&) = delete;
    size_t bytes() const noexcept;
    void release() noexcept;

private:
    int device_{0};
};

// Per-thread record of the last batch call's GPU time (charls_amd_last_timings).
struct Timings
{
    double values[8];
    int32_t count;
};
Timings& last_timings() noexcept;

// Process-wide totals of what the speculative stages of the lossless encoder did (tile_pipeline.hip, tile::Counter):
// regular-chain jobs, how many of them were walked again by the settling lane, the same two for the run chain.
constexpr int tile_counter_count = 6; // tile::kCounters
void speculation_counters(uint64_t out[tile_counter_count]) noexcept;

} // namespace jls::dev
