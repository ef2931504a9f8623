// scan_engine.h -- per-handle bridge between the host-memory C ABI and the device scan kernels.
// It replaces what the reference does with `make_scan_codec<...>()->encode_scan / decode_scan`
// (src/charls_jpegls_encoder.cpp:285-296, src/charls_jpegls_decoder.cpp:177-201): host pointer in, device work, host
// pointer out.  There is no CPU implementation behind it: without a usable GPU every call raises
// CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE.
#pragma once
#include <memory>

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

// What a handle needs on the device: a stream, the device copies of pixels and coded bytes, descriptors, a pinned
// staging area.  The reference's callers create an encoder or decoder per image (cli/benchmark.cpp does, inside its
// clock); creating and destroying these per handle cost more than coding a 4096 x 4096 frame (hipStreamCreate, five
// hipMalloc, hipHostMalloc and their frees), so handles take a set from a process-wide pool and give it back
// (scan_engine.cpp: acquire_resources / release_resources; a few sets of bounded size stay idle).
struct EngineResources
{
    hipStream_t stream{};
    int device{-1};
    dev::DeviceBuffer pixels, bits, scratch, desc, result;
    dev::PinnedBuffer staging;
    EngineResources() = default;
    EngineResources(const EngineResources&) = delete;
    EngineResources& operator=(const EngineResources&) = delete;
    ~EngineResources();
    size_t pooled_bytes{}; // what the pool counted for this set when it came back
    size_t device_bytes() const noexcept
    {
        return pixels.capacity() + bits.capacity() + scratch.capacity() + desc.capacity() + result.capacity();
    }
};

// Frees the idle sets of the pool (charls_amd_release_work_areas calls it, and the janitor after two quiet seconds).
void release_idle_engine_resources() noexcept;
uint64_t idle_engine_resource_bytes() noexcept; // device memory of the idle sets
uint64_t idle_releases() noexcept;              // how often the janitor gave memory back
// What the coalescer of the host-pointer ABI did so far (coalescer.h): calls, launches, calls that shared a launch, scans of
// the largest launch, merged batches that ran out of memory and were run call by call.
void coalescer_stats(uint64_t out[5]) noexcept;
// The frame's Coalescer::geometry_hint (scan_engine.cpp keeps coalescer.h to itself).
uint64_t frame_hint(uint32_t width, uint32_t height, int bits_per_sample) noexcept;

class ScanEngine;
// Ends the engine's part of a coding call of the facade however the call ends (ScanEngine::end_call).
struct CallScope
{
    explicit CallScope(ScanEngine& e) noexcept : engine(e) {}
    CallScope(const CallScope&) = delete;
    CallScope& operator=(const CallScope&) = delete;
    inline ~CallScope();
    ScanEngine& engine;
};

class ScanEngine
{
public:
    ScanEngine() = default;
    ScanEngine(const ScanEngine&) = delete;
    ScanEngine& operator=(const ScanEngine&) = delete;
    ~ScanEngine();

    // ---- encode: the frame's pixels are uploaded once, then one call per scan
    // (`hint`: Coalescer::geometry_hint of the frame -- what kind of call the upload belongs to)
    void upload_pixels(const uint8_t* source, size_t bytes, uint64_t hint = 0);
    // Encodes the scan whose first row starts `pixel_offset` bytes into the uploaded pixels; writes the entropy-coded
    // segment to `destination` (host) and returns its size.  Raises what scan_encoder::encode_scan would throw.
    size_t encode_scan(const ScanSpec& spec, size_t pixel_offset, size_t stride, uint8_t* destination,
                       size_t destination_size);
    // The `count` single-component scans of a planar frame at once (reference src/charls_jpegls_encoder.cpp:209-224 codes
    // them in a plain loop; they share nothing).  Scan c starts c * plane_bytes into the uploaded pixels and is coded
    // into a private device buffer of `capacity` bytes; results[c] is what scan_encoder::encode_scan would have done with a
    // destination of that size.  The caller places the scans with fetch_encoded_scan.
    void encode_planes(const ScanSpec& spec, uint32_t count, size_t plane_bytes, size_t stride, size_t capacity, ScanResult* results);
    void fetch_encoded_scan(uint32_t index, uint8_t* destination, size_t bytes);

    // ---- decode: the remaining source bytes are uploaded once, then one call per scan
    void upload_stream(const uint8_t* source, size_t bytes, uint64_t hint = 0);
    // Decodes the scan that starts `stream_offset` bytes into the uploaded stream into `destination` (host, rows
    // `stride` apart); returns the number of source bytes consumed.  Raises what scan_decoder::decode_scan would throw.
    size_t decode_scan(const ScanSpec& spec, size_t stream_offset, uint8_t* destination, size_t stride);
    // `count` scans of identical geometry and coding parameters at once (the component scans of a planar frame are
    // independent: own contexts, and FF DA cannot occur inside entropy-coded data, src/scan_decoder.hpp:272-284).
    // results[c] is the result of scan c; fetch_decoded_plane copies its rows out.
    void decode_planes(const ScanSpec& spec, const size_t* stream_offsets, uint32_t count, ScanResult* results);
    void fetch_decoded_plane(const ScanSpec& spec, uint32_t index, uint8_t* destination, size_t stride);

    // A coding call is coming on this handle: announced to the coalescer, so that calls of other threads that are about to
    // launch a batch this one can join wait for it (coalescer.h).  Before the coding call (the decoder was given its source,
    // the encoder its frame info -- the caller may never code anything) the announcement counts for a millisecond; inside it
    // (`uploading`: the host -> device copy is on its way) for as long as a leader is prepared to wait.  `hint`: the frame's
    // Coalescer::geometry_hint, 0 while it is not known.
    void expect_call(bool decode, uint64_t hint = 0, bool uploading = false) noexcept;
    // The coding call of the facade is over (normally or not), or the handle goes away: an announcement that was not
    // followed by a scan is taken back.
    void end_call() noexcept;

private:
    void ensure_stream();
    ScanDesc make_desc(const ScanSpec& spec) const;
    ScanResult run(const ScanDesc& desc, bool decode);
    void run_many(const ScanDesc* descs, uint32_t count, bool decode, ScanResult* results);
    void launch(const ScanDesc* descs, uint32_t n, bool decode, ScanResult* results, hipStream_t stream);
    void copy_out(uint8_t* destination, const uint8_t* device_source, size_t bytes);
    void copy_rows_out(uint8_t* destination, size_t stride, const uint8_t* device_source, size_t row_bytes, size_t rows);

    std::unique_ptr<EngineResources> r_; // from the pool at the first device call, back to it with the handle
    size_t pixel_bytes_{}, stream_bytes_{}, plane_capacity_{};
    struct Trace // CHARLS_AMD_TRACE: the phases of the current coding call, in ms
    {
        double begin_ms{}, upload_ms{}, sync_ms{}, submit_ms{}, launch_ms{}, copy_out_ms{};
        uint32_t launched_scans{};
        bool decode{};
    } trace_;
    int announced_lane_{-1};   // the coalescer lane this handle announced itself on (-1: none)
    uint64_t ticket_{};        // of that announcement (0: none)
};

inline CallScope::~CallScope()
{
    engine.end_call();
}

} // namespace jls
