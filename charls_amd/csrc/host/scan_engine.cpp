// scan_engine.cpp -- see scan_engine.h.
#include "scan_engine.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

namespace jls {

using dev::hip_check;

namespace {

// Idle resource sets.  Never destroyed: at process exit the HIP runtime may already be gone.
struct ResourcePool
{
    std::mutex mutex;
    std::vector<std::unique_ptr<EngineResources>> idle;
};
ResourcePool& pool()
{
    static ResourcePool* p = new ResourcePool;
    return *p;
}
constexpr size_t kMaxIdleSets = 4;
constexpr size_t kMaxIdleBytesPerSet = size_t{512} << 20; // a drop-in library must not sit on gigabytes of the caller's HBM

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
        if (p.idle.size() < kMaxIdleSets)
        {
            p.idle.push_back(std::move(r));
            return;
        }
    }
    // (r is destroyed here: too large to keep, or the pool is full)
}

} // namespace

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
    }
}

ScanEngine::~ScanEngine()
{
    release_resources(std::move(r_));
}

void ScanEngine::ensure_stream()
{
    dev::require_device();
    if (!r_)
        r_ = acquire_resources();
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
    auto* staged = static_cast<uint8_t*>(r_->staging.ensure(sizeof(ScanDesc) + sizeof(ScanResult)));
    std::memcpy(staged, &desc, sizeof desc);
    auto* d_desc = static_cast<ScanDesc*>(r_->desc.ensure(sizeof(ScanDesc)));
    auto* d_result = static_cast<ScanResult*>(r_->result.ensure(sizeof(ScanResult)));
    hip_check(hipMemcpyAsync(d_desc, staged, sizeof desc, hipMemcpyHostToDevice, r_->stream));
    if (decode)
        dev::launch_decode(desc, d_desc, d_result, 1, r_->stream);
    else
        dev::launch_encode(desc, d_desc, d_result, 1, r_->stream);
    hip_check(hipMemcpyAsync(staged + sizeof desc, d_result, sizeof(ScanResult), hipMemcpyDeviceToHost, r_->stream));
    hip_check(hipStreamSynchronize(r_->stream));
    if (dev::work_area_bytes() > (size_t{1} << 30))
        dev::release_work_areas(); // a drop-in library must not sit on gigabytes of the caller's HBM between calls
    ScanResult r;
    std::memcpy(&r, staged + sizeof desc, sizeof r);
    return r;
}

void ScanEngine::run_many(const ScanDesc* descs, uint32_t count, bool decode, ScanResult* results)
{
    const size_t desc_bytes = sizeof(ScanDesc) * count, result_bytes = sizeof(ScanResult) * count;
    auto* staged = static_cast<uint8_t*>(r_->staging.ensure(desc_bytes + result_bytes));
    std::memcpy(staged, descs, desc_bytes);
    auto* d_descs = static_cast<ScanDesc*>(r_->desc.ensure(desc_bytes));
    auto* d_results = static_cast<ScanResult*>(r_->result.ensure(result_bytes));
    hip_check(hipMemcpyAsync(d_descs, staged, desc_bytes, hipMemcpyHostToDevice, r_->stream));
    if (decode)
        dev::launch_decode(descs[0], d_descs, d_results, count, r_->stream);
    else
        dev::launch_encode(descs[0], d_descs, d_results, count, r_->stream);
    hip_check(hipMemcpyAsync(staged + desc_bytes, d_results, result_bytes, hipMemcpyDeviceToHost, r_->stream));
    hip_check(hipStreamSynchronize(r_->stream));
    if (dev::work_area_bytes() > (size_t{1} << 30))
        dev::release_work_areas();
    std::memcpy(results, staged + desc_bytes, result_bytes);
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
    hip_check(hipMemcpy(destination, r_->bits.as<uint8_t>() + plane_capacity_ * index, bytes, hipMemcpyDeviceToHost));
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
    hip_check(hipMemcpy2D(destination, stride, r_->pixels.as<uint8_t>() + row_bytes * spec.height * index, row_bytes, row_bytes,
                          spec.height, hipMemcpyDeviceToHost));
}

void ScanEngine::upload_pixels(const uint8_t* source, size_t bytes)
{
    ensure_stream();
    r_->pixels.ensure(bytes);
    pixel_bytes_ = bytes;
    hip_check(hipMemcpyAsync(r_->pixels.as<uint8_t>(), source, bytes, hipMemcpyHostToDevice, r_->stream));
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
    hip_check(hipMemcpy(destination, d.stream, r.bytes, hipMemcpyDeviceToHost));
    return r.bytes;
}

void ScanEngine::upload_stream(const uint8_t* source, size_t bytes)
{
    ensure_stream();
    r_->bits.ensure(bytes + 16); // the ring refill of the wave decoder reads whole 16-byte groups
    stream_bytes_ = bytes;
    hip_check(hipMemcpyAsync(r_->bits.as<uint8_t>(), source, bytes, hipMemcpyHostToDevice, r_->stream));
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
    hip_check(hipMemcpy2D(destination, stride, d.pixels, row_bytes, row_bytes, spec.height, hipMemcpyDeviceToHost));
    return r.bytes;
}

} // namespace jls
