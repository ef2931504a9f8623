// TEST-ONLY stand-in for <hip/hip_runtime_api.h>: just what charls_amd/csrc/host/scan_engine.cpp and device/runtime.h name, with
// "device memory" on the host heap and streams that complete at once.  It lets the CPU suite build the host layer of the
// ABI -- resource pool, coalescer, housekeeping thread -- without hipcc and run it under ThreadSanitizer / AddressSanitizer
// (tests/test_host_engine_cpu.py).  Not a HIP implementation, never part of the product.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>

enum hipError_t
{
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2
};
typedef struct stub_stream* hipStream_t;
enum hipMemcpyKind
{
    hipMemcpyHostToHost,
    hipMemcpyHostToDevice,
    hipMemcpyDeviceToHost,
    hipMemcpyDeviceToDevice
};
constexpr unsigned hipStreamNonBlocking = 1, hipHostMallocDefault = 0;

inline hipError_t hipGetDevice(int* device)
{
    *device = 0;
    return hipSuccess;
}
inline hipError_t hipSetDevice(int)
{
    return hipSuccess;
}
inline hipError_t hipGetLastError()
{
    return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t bytes)
{
    *p = std::malloc(bytes);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipFree(void* p)
{
    std::free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned)
{
    return hipMalloc(p, bytes);
}
inline hipError_t hipHostFree(void* p)
{
    return hipFree(p);
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned)
{
    *s = reinterpret_cast<hipStream_t>(std::malloc(8));
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int)
{
    return hipStreamCreateWithFlags(s, flags);
}
inline hipError_t hipStreamCreate(hipStream_t* s)
{
    return hipStreamCreateWithFlags(s, 0);
}
inline hipError_t hipStreamDestroy(hipStream_t s)
{
    std::free(s);
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t)
{
    return hipSuccess;
}
inline hipError_t hipDeviceGetStreamPriorityRange(int* lowest, int* highest)
{
    *lowest = 0;
    *highest = -1;
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t)
{
    std::memcpy(dst, src, bytes);
    return hipSuccess;
}
inline hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind,
                                   hipStream_t)
{
    for (size_t y = 0; y < height; ++y)
        std::memcpy(static_cast<char*>(dst) + y * dpitch, static_cast<const char*>(src) + y * spitch, width);
    return hipSuccess;
}
