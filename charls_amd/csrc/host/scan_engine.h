// scan_engine.h -- per-handle bridge between the host-memory C ABI and the device scan kernels.
// It replaces what the reference does with `make_scan_codec<...>()->encode_scan / decode_scan`
// (src/charls_jpegls_encoder.cpp:285-296, src/charls_jpegls_decoder.cpp:177-201): host pointer in, device work, host
// pointer out.  There is no CPU implementation behind it: without a usable GPU every call raises
// CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE.
#pragma once
#include "../device/runtime.h"
#include "common.h"

namespace jls {

struct ScanSpec
{
    uint32_t width, height;
    int32_t components; // in this scan
    int32_t interleave_mode;
    int32_t bits_per_sample;
    int32_t near_lossless;
    int32_t color_transformation;
    charls_jpegls_pc_parameters pc; // validated
    uint32_t restart_interval;
};

class ScanEngine
{
public:
    ScanEngine() = default;
    ScanEngine(const ScanEngine&) = delete;
    ScanEngine& operator=(const ScanEngine&) = delete;
    ~ScanEngine();

    // ---- encode: the frame's pixels are uploaded once, then one call per scan
    void upload_pixels(const uint8_t* source, size_t bytes);
    // Encodes the scan whose first row starts `pixel_offset` bytes into the uploaded pixels; writes the entropy-coded
    // segment to `destination` (host) and returns its size.  Raises what scan_encoder::encode_scan would throw.
    size_t encode_scan(const ScanSpec& spec, size_t pixel_offset, size_t stride, uint8_t* destination,
                       size_t destination_size);

    // ---- decode: the remaining source bytes are uploaded once, then one call per scan
    void upload_stream(const uint8_t* source, size_t bytes);
    // Decodes the scan that starts `stream_offset` bytes into the uploaded stream into `destination` (host, rows
    // `stride` apart); returns the number of source bytes consumed.  Raises what scan_decoder::decode_scan would throw.
    size_t decode_scan(const ScanSpec& spec, size_t stream_offset, uint8_t* destination, size_t stride);

private:
    void ensure_stream();
    ScanDesc make_desc(const ScanSpec& spec) const;
    ScanResult run(const ScanDesc& desc, bool decode);

    hipStream_t stream_{};
    bool have_stream_{};
    dev::DeviceBuffer pixels_, bits_, scratch_, desc_, result_;
    dev::PinnedBuffer staging_;
    size_t pixel_bytes_{}, stream_bytes_{};
};

} // namespace jls
