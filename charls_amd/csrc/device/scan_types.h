// scan_types.h -- plain-old-data shared between the host facade and the gfx950 kernels.
//
// One ScanDesc describes one JPEG-LS scan (the unit the reference hands to scan_encoder::encode_scan /
// scan_decoder::decode_scan, /root/reference/src/scan_encoder.hpp:28, src/scan_decoder.hpp:33).  Batches are arrays of
// ScanDesc in device memory; every scan is independent (own contexts, own bitstream), which is what the engine shards
// over wavefronts, CUs and GPUs.
#pragma once
#include <cstddef>
#include <cstdint>

namespace jls {

// charls_jpegls_errc values raised on the hot path (reference include/charls/public_types.h:28-88)
enum : uint32_t
{
    kOk = 0,
    kDestinationTooSmall = 3,
    kNeedMoreData = 4,
    kInvalidData = 5,
    kRestartMarkerNotFound = 23,
};

struct ScanDesc
{
    // geometry / coding parameters (already validated by the facade)
    uint32_t width;
    uint32_t height;
    int32_t components;       // components coded in THIS scan (1 for ILV_NONE, 2..4 for ILV_LINE / ILV_SAMPLE)
    int32_t interleave_mode;  // 0 none, 1 line, 2 sample
    int32_t bits_per_sample;
    int32_t near_lossless;
    int32_t color_transformation;
    int32_t t1, t2, t3;
    int32_t reset;            // RESET after the reference's uint8_t cast (src/scan_codec.hpp:142)
    uint32_t restart_interval; // lines per restart interval, 0 = none (encode: extension, see restart_intervals.hip)
    // memory (all device pointers)
    uint8_t* pixels;          // first row of the scan in the user's layout (source for encode, destination for decode)
    uint64_t pixel_stride;    // bytes between rows
    uint8_t* stream;          // entropy-coded segment: destination for encode, source for decode
    uint64_t stream_capacity; // encode: bytes available; decode: bytes until the end of the source buffer
    uint16_t* line_scratch;   // 2 * planes * (width + 2) samples, planes = ILV_NONE ? 1 : components
};

struct ScanResult
{
    uint32_t errc;   // kOk or one of the codes above
    uint32_t flags;
    uint64_t bytes;  // encode: bytes written; decode: bytes consumed (scan_decoder::get_actual_position)
};

// Derived per-scan constants (reference src/default_traits.hpp:51-59, src/jpegls_algorithm.hpp:124-140).
struct Traits
{
    int32_t maxval, near, range, qbpp, limit, t1, t2, t3, reset, bpp;
};

// find_scan_end (container_kernels.hip): the stretch of a batch's stream slots to search, as offsets from the first slot.
struct MarkerSearch
{
    uint64_t from, end; // offsets from the first slot
};
constexpr unsigned long long kNoMarker = ~0ull;

} // namespace jls
