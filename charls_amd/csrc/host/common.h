// common.h -- host-side plumbing shared by the C-ABI facade: error transport, argument checks, limits.
// Limits and semantics follow the reference (src/constants.hpp:14-51, src/util.hpp:83-101,200-260).
#pragma once
#include <cstddef>
#include <cstdint>
#include <new>

#include "charls_amd.h"

namespace jls {

// Internal exception; never crosses the C ABI (every thunk converts it to the errc it carries).
struct error
{
    charls_jpegls_errc code;
};

[[noreturn]] inline void raise(charls_jpegls_errc code)
{
    throw error{code};
}

inline void check_argument(bool ok, charls_jpegls_errc code = CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT)
{
    if (!ok)
        raise(code);
}

inline void check_operation(bool ok)
{
    if (!ok)
        raise(CHARLS_JPEGLS_ERRC_INVALID_OPERATION);
}

template <typename T>
inline T* check_pointer(T* p)
{
    if (!p)
        raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT);
    return p;
}

// A (pointer,size) pair is valid when the pointer is non-null or the size is zero (src/util.hpp check_argument(span)).
inline void check_buffer(const void* p, size_t size)
{
    if (!p && size != 0)
        raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT);
}

inline size_t checked_mul(size_t a, size_t b)
{
    size_t r;
    if (__builtin_mul_overflow(a, b, &r))
        raise(CHARLS_JPEGLS_ERRC_PARAMETER_VALUE_NOT_SUPPORTED);
    return r;
}

constexpr uint32_t kMaxDimension = 100000;       // src/constants.hpp:25-28
constexpr int32_t kMinBits = 2, kMaxBits = 16;   // :22-23
constexpr int32_t kMaxComponents = 255;          // :17
constexpr int32_t kMaxComponentsInScan = 4;      // :18
constexpr int32_t kMaxNear = 255;                // :24
constexpr size_t kSegmentMaxData = 65535 - 2;    // :55
constexpr size_t kSpiffEntryMaxData = 65528;     // :47
constexpr size_t kSpiffHeaderSize = 34;          // :45

inline int32_t bit_max_value(int32_t bits) { return (1 << bits) - 1; }
inline size_t bytes_per_sample(int32_t bits) { return static_cast<size_t>((bits + 7) / 8); }
inline int32_t max_near_for(int32_t maxval) { return maxval / 2 < kMaxNear ? maxval / 2 : kMaxNear; } // src/jpegls_algorithm.hpp:46-51

// Converts the in-flight exception to an errc (function-try-block tail of every thunk).
inline charls_jpegls_errc current_exception_to_errc() noexcept
{
    try
    {
        throw;
    }
    catch (const error& e)
    {
        return e.code;
    }
    catch (const std::bad_alloc&)
    {
        return CHARLS_JPEGLS_ERRC_NOT_ENOUGH_MEMORY;
    }
    catch (...)
    {
        return CHARLS_AMD_ERRC_DEVICE_FAILURE;
    }
}

} // namespace jls
