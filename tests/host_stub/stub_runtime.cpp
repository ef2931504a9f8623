// TEST-ONLY: the part of charls_amd/csrc/device/runtime.hip that charls_amd/csrc/host/scan_engine.cpp links against, with
// fake launches (see hip/hip_runtime_api.h in this directory).  A "decoder launch" fills every scan's pixel buffer with a
// byte that depends on the scan's OWN coded bytes, an "encoder launch" writes a stream that depends on the scan's OWN pixels --
// so a caller that got somebody else's result, or a buffer that was freed under it, is seen (and AddressSanitizer /
// ThreadSanitizer see the rest).  The shared work areas of the encoder launches are one heap block that grows.
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

#include "device/runtime.h"

namespace jls::dev {

charls_jpegls_errc device_status() noexcept
{
    return CHARLS_JPEGLS_ERRC_SUCCESS;
}
void require_device() {}

void* DeviceBuffer::ensure(size_t bytes)
{
    if (bytes <= cap_ && ptr_)
        return ptr_;
    release();
    hip_check(hipMalloc(&ptr_, bytes < 256 ? 256 : bytes));
    cap_ = bytes < 256 ? 256 : bytes;
    device_ = 0;
    return ptr_;
}
void DeviceBuffer::release() noexcept
{
    if (ptr_)
        (void)hipFree(ptr_);
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
    hip_check(hipHostMalloc(&ptr_, bytes < 256 ? 256 : bytes, hipHostMallocDefault));
    cap_ = bytes < 256 ? 256 : bytes;
    return ptr_;
}

namespace {
std::atomic<int> g_long_kernels{0};
std::mutex g_shared_turn;
DeviceBuffer* g_shared_area = new DeviceBuffer; // never destroyed, like the product's
std::atomic<size_t> g_shared_held{0};
thread_local bool t_in_shared_scope = false;
} // namespace

void long_kernel_begins() noexcept { g_long_kernels.fetch_add(1); }
void long_kernel_ends() noexcept { g_long_kernels.fetch_sub(1); }
void reap_deferred_frees() noexcept {}
uint64_t deferred_free_bytes() noexcept { return 0; }
uint64_t workspace_limit() noexcept { return 0; }
size_t shared_areas_keep_bytes() noexcept { return size_t{1} << 30; }
size_t thread_work_area_bytes() noexcept { return 0; }
void release_thread_work_areas() noexcept {}
size_t shared_work_area_bytes() noexcept { return g_shared_held.load(); }
void release_shared_work_areas() noexcept
{
    if (t_in_shared_scope)
        return;
    std::lock_guard<std::mutex> turn(g_shared_turn); // (a merged launch that is running finishes first)
    g_shared_area->release();
    g_shared_held.store(0);
}

SharedAreasScope::SharedAreasScope()
{
    g_shared_turn.lock();
    t_in_shared_scope = true;
}
SharedAreasScope::~SharedAreasScope()
{
    g_shared_held.store(g_shared_area->capacity());
    t_in_shared_scope = false;
    g_shared_turn.unlock();
}
size_t SharedAreasScope::bytes() const noexcept { return g_shared_area->capacity(); }
void SharedAreasScope::release() noexcept { g_shared_area->release(); }

void launch_decode(const ScanDesc& proto, const ScanDesc* d_descs, ScanResult* d_results, uint32_t count, hipStream_t)
{
    (void)proto;
    std::this_thread::sleep_for(std::chrono::microseconds(300 + 5 * count)); // (a launch takes a while: others arrive meanwhile)
    for (uint32_t s = 0; s < count; ++s)
    {
        const ScanDesc& d = d_descs[s];
        uint8_t tag = 0;
        for (uint64_t b = 0; b < d.stream_capacity; ++b)
            tag = static_cast<uint8_t>(tag * 31 + d.stream[b]);
        for (uint32_t y = 0; y < d.height; ++y)
            std::memset(d.pixels + y * d.pixel_stride, tag, d.width);
        d_results[s] = ScanResult{0, 0, d.stream_capacity};
    }
}

void launch_encode(const ScanDesc& proto, ScanDesc* d_descs, ScanResult* d_results, uint32_t count, hipStream_t)
{
    // (the merged encoder launches of all threads share one work area: touch all of it)
    auto* area = static_cast<uint8_t*>(g_shared_area->ensure(static_cast<size_t>(count) * proto.width * 8));
    std::memset(area, 0xA5, static_cast<size_t>(count) * proto.width * 8);
    std::this_thread::sleep_for(std::chrono::microseconds(200));
    for (uint32_t s = 0; s < count; ++s)
    {
        const ScanDesc& d = d_descs[s];
        uint8_t tag = 0;
        for (uint32_t y = 0; y < d.height; ++y)
            for (uint32_t x = 0; x < d.width; ++x)
                tag = static_cast<uint8_t>(tag * 131 + d.pixels[y * d.pixel_stride + x]);
        const uint64_t bytes = 16;
        if (d.stream_capacity < bytes)
        {
            d_results[s] = ScanResult{CHARLS_JPEGLS_ERRC_DESTINATION_TOO_SMALL, 0, 0};
            continue;
        }
        std::memset(d.stream, tag, bytes);
        d_results[s] = ScanResult{0, 0, bytes};
    }
}

} // namespace jls::dev
