// scan_engine.cpp -- see scan_engine.h.
#include "scan_engine.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace jls {

using dev::hip_check;

ScanEngine::~ScanEngine()
{
    if (have_stream_)
        (void)hipStreamDestroy(stream_);
}

void ScanEngine::ensure_stream()
{
    dev::require_device();
    if (!have_stream_)
    {
        hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        have_stream_ = true;
    }
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
    auto* staged = static_cast<uint8_t*>(staging_.ensure(sizeof(ScanDesc) + sizeof(ScanResult)));
    std::memcpy(staged, &desc, sizeof desc);
    auto* d_desc = static_cast<ScanDesc*>(desc_.ensure(sizeof(ScanDesc)));
    auto* d_result = static_cast<ScanResult*>(result_.ensure(sizeof(ScanResult)));
    hip_check(hipMemcpyAsync(d_desc, staged, sizeof desc, hipMemcpyHostToDevice, stream_));
    if (decode)
        dev::launch_decode(desc, d_desc, d_result, 1, stream_);
    else
        dev::launch_encode(desc, d_desc, d_result, 1, stream_);
    hip_check(hipMemcpyAsync(staged + sizeof desc, d_result, sizeof(ScanResult), hipMemcpyDeviceToHost, stream_));
    hip_check(hipStreamSynchronize(stream_));
    if (dev::work_area_bytes() > (size_t{1} << 30))
        dev::release_work_areas(); // a drop-in library must not sit on gigabytes of the caller's HBM between calls
    ScanResult r;
    std::memcpy(&r, staged + sizeof desc, sizeof r);
    return r;
}

void ScanEngine::run_many(const ScanDesc* descs, uint32_t count, bool decode, ScanResult* results)
{
    const size_t desc_bytes = sizeof(ScanDesc) * count, result_bytes = sizeof(ScanResult) * count;
    auto* staged = static_cast<uint8_t*>(staging_.ensure(desc_bytes + result_bytes));
    std::memcpy(staged, descs, desc_bytes);
    auto* d_descs = static_cast<ScanDesc*>(desc_.ensure(desc_bytes));
    auto* d_results = static_cast<ScanResult*>(result_.ensure(result_bytes));
    hip_check(hipMemcpyAsync(d_descs, staged, desc_bytes, hipMemcpyHostToDevice, stream_));
    if (decode)
        dev::launch_decode(descs[0], d_descs, d_results, count, stream_);
    else
        dev::launch_encode(descs[0], d_descs, d_results, count, stream_);
    hip_check(hipMemcpyAsync(staged + desc_bytes, d_results, result_bytes, hipMemcpyDeviceToHost, stream_));
    hip_check(hipStreamSynchronize(stream_));
    if (dev::work_area_bytes() > (size_t{1} << 30))
        dev::release_work_areas();
    std::memcpy(results, staged + desc_bytes, result_bytes);
}

void ScanEngine::encode_planes(const ScanSpec& spec, uint32_t count, size_t plane_bytes, size_t stride, size_t capacity, ScanResult* results)
{
    ensure_stream();
    const size_t bound = dev::worst_case_scan_bytes(spec.width, spec.height, spec.components, spec.bits_per_sample);
    plane_capacity_ = (std::min(capacity, bound) + 255) & ~size_t{255};
    auto* bits = static_cast<uint8_t*>(bits_.ensure(plane_capacity_ * count));
    const size_t scratch_samples = dev::line_scratch_samples(spec.width, spec.interleave_mode, spec.components);
    auto* scratch = static_cast<uint16_t*>(scratch_.ensure(scratch_samples * sizeof(uint16_t) * count));
    std::vector<ScanDesc> descs(count, make_desc(spec));
    for (uint32_t c = 0; c < count; ++c)
    {
        descs[c].pixels = pixels_.as<uint8_t>() + plane_bytes * c;
        descs[c].pixel_stride = stride;
        descs[c].stream = bits + plane_capacity_ * c;
        descs[c].stream_capacity = std::min(capacity, bound);
        descs[c].line_scratch = scratch + scratch_samples * c;
    }
    run_many(descs.data(), count, false, results);
}

void ScanEngine::fetch_encoded_scan(uint32_t index, uint8_t* destination, size_t bytes)
{
    hip_check(hipMemcpy(destination, bits_.as<uint8_t>() + plane_capacity_ * index, bytes, hipMemcpyDeviceToHost));
}

void ScanEngine::decode_planes(const ScanSpec& spec, const size_t* stream_offsets, uint32_t count, ScanResult* results)
{
    ensure_stream();
    const size_t row_bytes = static_cast<size_t>(spec.width) * bytes_per_sample(spec.bits_per_sample);
    const size_t plane = row_bytes * spec.height;
    auto* pixels = static_cast<uint8_t*>(pixels_.ensure(plane * count));
    const size_t scratch_samples = dev::line_scratch_samples(spec.width, spec.interleave_mode, spec.components);
    auto* scratch = static_cast<uint16_t*>(scratch_.ensure(scratch_samples * sizeof(uint16_t) * count));
    std::vector<ScanDesc> descs(count, make_desc(spec));
    for (uint32_t c = 0; c < count; ++c)
    {
        descs[c].pixels = pixels + plane * c;
        descs[c].pixel_stride = row_bytes;
        descs[c].stream = bits_.as<uint8_t>() + stream_offsets[c];
        descs[c].stream_capacity = stream_bytes_ - stream_offsets[c];
        descs[c].line_scratch = scratch + scratch_samples * c;
    }
    run_many(descs.data(), count, true, results);
}

void ScanEngine::fetch_decoded_plane(const ScanSpec& spec, uint32_t index, uint8_t* destination, size_t stride)
{
    const size_t row_bytes = static_cast<size_t>(spec.width) * bytes_per_sample(spec.bits_per_sample);
    hip_check(hipMemcpy2D(destination, stride, pixels_.as<uint8_t>() + row_bytes * spec.height * index, row_bytes, row_bytes,
                          spec.height, hipMemcpyDeviceToHost));
}

void ScanEngine::upload_pixels(const uint8_t* source, size_t bytes)
{
    ensure_stream();
    pixels_.ensure(bytes);
    pixel_bytes_ = bytes;
    hip_check(hipMemcpyAsync(pixels_.as<uint8_t>(), source, bytes, hipMemcpyHostToDevice, stream_));
}

size_t ScanEngine::encode_scan(const ScanSpec& spec, size_t pixel_offset, size_t stride, uint8_t* destination,
                               size_t destination_size)
{
    ensure_stream();
    // Never allocate more than the scan can possibly produce; a larger destination behaves identically.
    const size_t bound = dev::worst_case_scan_bytes(spec.width, spec.height, spec.components, spec.bits_per_sample);
    const size_t capacity = std::min(destination_size, bound);
    ScanDesc d = make_desc(spec);
    d.pixels = pixels_.as<uint8_t>() + pixel_offset;
    d.pixel_stride = stride;
    d.stream = static_cast<uint8_t*>(bits_.ensure(capacity));
    d.stream_capacity = capacity;
    d.line_scratch = static_cast<uint16_t*>(
        scratch_.ensure(dev::line_scratch_samples(spec.width, spec.interleave_mode, spec.components) * sizeof(uint16_t)));
    const ScanResult r = run(d, false);
    if (r.errc != kOk)
        raise(static_cast<charls_jpegls_errc>(r.errc));
    hip_check(hipMemcpy(destination, d.stream, r.bytes, hipMemcpyDeviceToHost));
    return r.bytes;
}

void ScanEngine::upload_stream(const uint8_t* source, size_t bytes)
{
    ensure_stream();
    bits_.ensure(bytes + 16); // the ring refill of the wave decoder reads whole 16-byte groups
    stream_bytes_ = bytes;
    hip_check(hipMemcpyAsync(bits_.as<uint8_t>(), source, bytes, hipMemcpyHostToDevice, stream_));
}

size_t ScanEngine::decode_scan(const ScanSpec& spec, size_t stream_offset, uint8_t* destination, size_t stride)
{
    ensure_stream();
    const size_t planes = spec.interleave_mode == 0 ? 1 : static_cast<size_t>(spec.components);
    const size_t row_bytes = planes * spec.width * bytes_per_sample(spec.bits_per_sample);
    ScanDesc d = make_desc(spec);
    d.pixels = static_cast<uint8_t*>(pixels_.ensure(row_bytes * spec.height)); // device rows are packed
    d.pixel_stride = row_bytes;
    d.stream = bits_.as<uint8_t>() + stream_offset;
    d.stream_capacity = stream_bytes_ - stream_offset;
    d.line_scratch = static_cast<uint16_t*>(
        scratch_.ensure(dev::line_scratch_samples(spec.width, spec.interleave_mode, spec.components) * sizeof(uint16_t)));
    const ScanResult r = run(d, true);
    if (r.errc != kOk) // the destination content after a failed decode is unspecified in the reference as well
        raise(static_cast<charls_jpegls_errc>(r.errc));
    hip_check(hipMemcpy2D(destination, stride, d.pixels, row_bytes, row_bytes, spec.height, hipMemcpyDeviceToHost));
    return r.bytes;
}

} // namespace jls
