// runtime.hip -- device discovery, arenas and kernel launches (see runtime.h).  Compiled for gfx950 only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "runtime.h"
#include "scan_serial.hip"
#include "container_kernels.hip"

namespace jls::dev {

namespace {
std::once_flag g_once;
charls_jpegls_errc g_status = CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE;
std::atomic<int32_t> g_engine{0};

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

void* DeviceBuffer::ensure(size_t bytes)
{
    if (bytes <= cap_ && ptr_)
        return ptr_;
    release();
    const size_t want = bytes < 256 ? 256 : bytes;
    hip_check(hipMalloc(&ptr_, want));
    cap_ = want;
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

Timings& last_timings() noexcept
{
    static thread_local Timings t{};
    return t;
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

void launch_place_epilogue(uint8_t* slots, uint64_t slot_pitch, FrameCursorPod* cursors, bool even_size, uint64_t* sizes,
                           uint32_t* errcs, uint32_t frames, hipStream_t stream)
{
    hipLaunchKernelGGL(place_epilogue, dim3((frames + 63) / 64), dim3(64), 0, stream, slots, slot_pitch,
                       reinterpret_cast<FrameCursor*>(cursors), even_size ? 1u : 0u, sizes, errcs, frames);
    hip_check(hipGetLastError());
}

} // namespace jls::dev
