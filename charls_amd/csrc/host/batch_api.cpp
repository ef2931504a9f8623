// batch_api.cpp -- charls_amd_encode_batch_device / charls_amd_decode_batch_device (charls_amd.h part 2).
//
// The batch is the engine's native unit: `frame_count` independent frames resident in HBM, every scan of every frame a
// separate chain of work for the GPU (frames shard with no exchange).  The container bytes around the
// entropy-coded segments are produced by the same StreamWriter the part-1 encoder uses, and parsed by the same
// StreamReader the part-1 decoder uses, so each frame's .jls file / error code is what part 1 gives for that frame.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../device/knobs.h"
#include "../device/runtime.h"
#include "common.h"
#include "preset.h"
#include "stream_reader.h"
#include "stream_writer.h"

using namespace jls;
using dev::hip_check;

namespace {
// One window of every listed stream -> a contiguous staging buffer (then ONE device-to-host copy instead of one small,
// synchronously staged copy per frame: 4096 of those were 60 ms per round of the batch decoder).
struct WindowSpec
{
    uint64_t offset; // from the first slot
    uint32_t bytes;
    uint32_t pad;
};
__global__ void gather_windows(const uint8_t* __restrict__ slots, const WindowSpec* __restrict__ specs, uint8_t* __restrict__ out,
                               uint32_t window)
{
    const WindowSpec w = specs[blockIdx.x];
    const uint8_t* src = slots + w.offset;
    uint8_t* dst = out + (size_t)blockIdx.x * window;
    for (uint32_t b = threadIdx.x; b < w.bytes; b += blockDim.x)
        dst[b] = src[b];
}

struct EventTimer
{
    hipEvent_t a{}, b{};
    hipStream_t s;
    explicit EventTimer(hipStream_t stream) : s(stream)
    {
        hip_check(hipEventCreate(&a));
        hip_check(hipEventCreate(&b));
    }
    ~EventTimer()
    {
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    }
    void start() { hip_check(hipEventRecord(a, s)); }
    void stop() { hip_check(hipEventRecord(b, s)); }
    double ms()
    {
        float v = 0;
        hip_check(hipEventSynchronize(b));
        hip_check(hipEventElapsedTime(&v, a, b));
        return v;
    }
};

// Validation of the coding parameters: the checks charls_jpegls_encoder::encode_components performs before any scan
// is coded (reference src/charls_jpegls_encoder.cpp:182-207, 298-358).
void validate_encode(const charls_amd_codec_params& p, size_t frame_pitch, uint32_t stride_arg, size_t* stride_out,
                     charls_jpegls_pc_parameters* pc_out)
{
    const charls_frame_info& f = p.frame_info;
    check_argument(f.width >= 1 && f.width <= kMaxDimension, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_WIDTH);
    check_argument(f.height >= 1 && f.height <= kMaxDimension, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_HEIGHT);
    check_argument(f.bits_per_sample >= kMinBits && f.bits_per_sample <= kMaxBits,
                   CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_BITS_PER_SAMPLE);
    check_argument(f.component_count >= 1 && f.component_count <= kMaxComponents,
                   CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_COMPONENT_COUNT);
    check_argument(p.near_lossless >= 0 && p.near_lossless <= kMaxNear, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_NEAR_LOSSLESS);
    check_argument(p.interleave_mode >= 0 && p.interleave_mode <= 2, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_INTERLEAVE_MODE);
    check_argument(p.color_transformation >= 0 && p.color_transformation <= 3,
                   CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_COLOR_TRANSFORMATION);
    check_argument(p.encoding_options <= 7u, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_ENCODING_OPTIONS);
    if (f.component_count == 1 && p.interleave_mode != 0)
        raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_INTERLEAVE_MODE);
    if (p.interleave_mode != 0 && f.component_count > kMaxComponentsInScan)
        raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_COMPONENT_COUNT);
    const int32_t bit_maxval = bit_max_value(f.bits_per_sample);
    int32_t maxval = bit_maxval;
    if (p.preset_coding_parameters.maximum_sample_value != 0)
    {
        if (p.preset_coding_parameters.maximum_sample_value < 1 || p.preset_coding_parameters.maximum_sample_value > bit_maxval)
            raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_JPEGLS_PC_PARAMETERS);
        maxval = p.preset_coding_parameters.maximum_sample_value;
    }
    if (p.near_lossless > max_near_for(maxval))
        raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_NEAR_LOSSLESS);
    const size_t row = static_cast<size_t>(f.width) * bytes_per_sample(f.bits_per_sample) *
                       (p.interleave_mode == 0 ? 1u : static_cast<size_t>(f.component_count));
    size_t stride = stride_arg;
    if (stride == 0)
        stride = row;
    else if (stride < row)
        raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_STRIDE);
    const size_t need = (p.interleave_mode == 0 ? checked_mul(stride * static_cast<size_t>(f.component_count), f.height)
                                                : checked_mul(stride, f.height)) -
                        (stride - row);
    if (frame_pitch < need)
        raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE);
    if (!pc_validate(p.preset_coding_parameters, bit_maxval, p.near_lossless, pc_out))
        raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_JPEGLS_PC_PARAMETERS);
    if (p.color_transformation != 0 && !color_transformation_possible(f, p.near_lossless, p.interleave_mode))
        raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_COLOR_TRANSFORMATION);
    *stride_out = stride;
}

ScanDesc base_desc(const charls_frame_info& f, int32_t components, int32_t ilv, int32_t near, int32_t xform,
                   const charls_jpegls_pc_parameters& pc, uint32_t restart)
{
    ScanDesc d{};
    d.width = f.width;
    d.height = f.height;
    d.components = components;
    d.interleave_mode = ilv;
    d.bits_per_sample = f.bits_per_sample;
    d.near_lossless = near;
    d.color_transformation = xform;
    d.t1 = pc.threshold1;
    d.t2 = pc.threshold2;
    d.t3 = pc.threshold3;
    d.reset = static_cast<uint8_t>(pc.reset_value);
    d.restart_interval = restart;
    return d;
}

thread_local bool t_force_rounds = false; // the batch encoder codes the component scans of planar frames one round per component

} // namespace

extern "C" charls_jpegls_errc charls_amd_encode_batch_device(const charls_amd_codec_params* params, uint32_t frame_count,
                                                             const void* d_frames, size_t frame_pitch_bytes,
                                                             uint32_t stride_arg, void* d_streams,
                                                             size_t stream_pitch_bytes, uint64_t* sizes,
                                                             charls_jpegls_errc* errcs, void* hip_stream)
try
{
    check_pointer(params);
    check_pointer(sizes);
    check_pointer(errcs);
    if (frame_count == 0)
        return CHARLS_JPEGLS_ERRC_SUCCESS;
    check_pointer(d_frames);
    check_pointer(d_streams);
    dev::require_device();
    const charls_amd_codec_params& p = *params;
    size_t stride = 0;
    charls_jpegls_pc_parameters pc{};
    validate_encode(p, frame_pitch_bytes, stride_arg, &stride, &pc);
    const charls_frame_info& f = p.frame_info;
    auto stream = static_cast<hipStream_t>(hip_stream);

    // ---- container bytes, produced once on the host by the writer of part 1
    uint8_t prologue[512];
    StreamWriter w;
    w.set_destination(prologue, sizeof prologue);
    w.start_of_image();
    if (p.encoding_options & 2u)
    {
        static const char version[] = "charls 3.0.0";
        w.comment(reinterpret_cast<const uint8_t*>(version), sizeof version);
    }
    if (p.color_transformation != 0)
        w.color_transform(p.color_transformation);
    if (f.component_count * 3 + 32 > 400)
        raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_COMPONENT_COUNT); // batch API: at most ~120 components
    if (w.start_of_frame(f))
        w.oversize_dimensions(f.height, f.width);
    if (!pc_is_default(p.preset_coding_parameters, default_pc(bit_max_value(f.bits_per_sample), p.near_lossless)) ||
        ((p.encoding_options & 4u) && f.bits_per_sample > 12))
        w.preset_coding_parameters(pc);
    if (p.restart_interval != 0)
        w.define_restart_interval(p.restart_interval);
    const uint32_t prologue_size = static_cast<uint32_t>(w.bytes_written());

    const uint32_t rounds = p.interleave_mode == 0 ? static_cast<uint32_t>(f.component_count) : 1u;
    const int32_t comps_per_scan = p.interleave_mode == 0 ? 1 : f.component_count;
    std::vector<std::vector<uint8_t>> sos(rounds);
    {
        std::vector<uint8_t> tmp(static_cast<size_t>(rounds) * 16);
        StreamWriter ws; // one writer for all scans: component identifiers keep counting across scans
        ws.set_destination(tmp.data(), tmp.size());
        for (uint32_t r = 0; r < rounds; ++r)
        {
            const size_t before = ws.bytes_written();
            ws.start_of_scan(comps_per_scan, p.near_lossless, p.interleave_mode);
            sos[r].assign(tmp.data() + before, tmp.data() + ws.bytes_written());
        }
    }

    // ---- device work areas
    const size_t scratch_samples = dev::line_scratch_samples(f.width, p.interleave_mode, comps_per_scan);
    dev::DeviceBuffer d_descs, d_results, d_cursors, d_scratch, d_blob, d_sizes, d_errcs;
    d_descs.ensure(sizeof(ScanDesc) * frame_count);
    d_results.ensure(sizeof(ScanResult) * frame_count);
    d_cursors.ensure(sizeof(dev::FrameCursorPod) * frame_count);
    d_scratch.ensure(scratch_samples * sizeof(uint16_t) * frame_count);
    d_sizes.ensure(sizeof(uint64_t) * frame_count);
    d_errcs.ensure(sizeof(uint32_t) * frame_count);
    size_t blob_bytes = prologue_size;
    for (auto& s : sos)
        blob_bytes += s.size();
    std::vector<uint8_t> blob(blob_bytes);
    std::memcpy(blob.data(), prologue, prologue_size);
    {
        size_t o = prologue_size;
        for (auto& s : sos)
        {
            std::memcpy(blob.data() + o, s.data(), s.size());
            o += s.size();
        }
    }
    d_blob.ensure(blob_bytes);
    hip_check(hipMemcpyAsync(d_blob.as<uint8_t>(), blob.data(), blob_bytes, hipMemcpyHostToDevice, stream));

    std::vector<ScanDesc> descs(frame_count);
    auto* slots = static_cast<uint8_t*>(d_streams);
    const auto* frames = static_cast<const uint8_t*>(d_frames);

    EventTimer total(stream), scans(stream);
    double scan_ms = 0;
    dev::last_timings().count = 0;
    total.start();
    dev::launch_place_prologue(slots, stream_pitch_bytes, d_blob.as<uint8_t>(), prologue_size,
                               d_cursors.as<dev::FrameCursorPod>(), frame_count, stream);
    size_t blob_offset = prologue_size;
    // ---- the component scans of planar frames TOGETHER (reference src/charls_jpegls_encoder.cpp:209-224 codes them in a plain
    // loop; they share nothing): one launch over frames x components into private buffers, then every frame's scans are put in
    // place behind their SOS headers (place_plane_scans).  Frames go in groups that keep the private buffers under 8 GiB.  A
    // frame with a scan that failed or that ends within 4 bytes of its destination is coded again below, scan by scan with
    // the reference's capacities -- its verdict depends on them there (src/scan_encoder.hpp:117-120).
    std::vector<uint8_t> redo_frames; // 1 = code this frame in rounds (all of them when the scans are not coded together)
    const bool equal_headers = [&] {
        for (auto& h : sos)
            if (h.size() != sos[0].size())
                return false;
        return true;
    }();
    // (Worth it while a round alone does not fill the chip: 256 4096 x 4096 RGB frames code at 17.9 GPix/s together and at
    // 18.8 in rounds of 256 scans -- the scans' second trip through memory --, four 2048 x 2048 frames three times faster.)
    constexpr uint32_t kTogetherFrames = 128;
    if (rounds > 1 && p.restart_interval == 0 && equal_headers && !t_force_rounds && frame_count <= kTogetherFrames &&
        knobs::get_or(knobs::kBatchRounds, 0) == 0)
    {
        const size_t worst = dev::worst_case_scan_bytes(f.width, f.height, 1, f.bits_per_sample);
        const size_t capacity = (std::min(stream_pitch_bytes, worst) + 255) & ~size_t{255};
        // The private buffers are a work area like the pipeline's: they come out of the configured workspace
        // (charls_amd_set_workspace_limit) -- half of what this thread may hold, the pipeline's arena wants the rest --, never
        // more than 8 GiB, and when they cannot be had the frames are coded in rounds, which need none.
        const size_t scratch_bytes = (scratch_samples * sizeof(uint16_t) + 255) & ~size_t{255}; // (the exact re-coding of a scan that ends within 3 bytes of its buffer)
        const size_t per_frame = (capacity + scratch_bytes) * rounds;
        const size_t budget = std::min<size_t>(size_t{8} << 30, dev::work_area_budget() / 2);
        uint32_t group = static_cast<uint32_t>(std::min<size_t>(frame_count, budget / per_frame));
        uint8_t* priv = nullptr;
        while (group >= 1 && (priv = static_cast<uint8_t*>(dev::try_ensure(dev::plane_arena(), per_frame * group))) == nullptr) // (kept between calls)
            group /= 2;
        if (priv != nullptr)
        {
        auto* plane_scratch = reinterpret_cast<uint16_t*>(priv + static_cast<size_t>(capacity) * rounds * group);
        dev::DeviceBuffer d_plane_descs, d_plane_results, d_redo;
        d_plane_descs.ensure(sizeof(ScanDesc) * rounds * group);
        d_plane_results.ensure(sizeof(ScanResult) * rounds * group);
        d_redo.ensure(sizeof(uint32_t) * frame_count);
        std::vector<ScanDesc> plane_descs(static_cast<size_t>(rounds) * group);
        const uint32_t sos_size = static_cast<uint32_t>(sos[0].size());
        for (uint32_t first = 0; first < frame_count; first += group)
        {
            const uint32_t n = std::min(group, frame_count - first);
            for (uint32_t i = 0; i < n; ++i)
                for (uint32_t r = 0; r < rounds; ++r)
                {
                    ScanDesc d = base_desc(f, 1, 0, p.near_lossless, p.color_transformation, pc, 0);
                    d.pixels = const_cast<uint8_t*>(frames) + (first + i) * frame_pitch_bytes + r * stride * f.height;
                    d.pixel_stride = stride;
                    d.stream = priv + (static_cast<size_t>(i) * rounds + r) * capacity;
                    d.stream_capacity = std::min(stream_pitch_bytes, worst);
                    d.line_scratch = plane_scratch + (static_cast<size_t>(i) * rounds + r) * (scratch_bytes / sizeof(uint16_t));
                    plane_descs[static_cast<size_t>(i) * rounds + r] = d;
                }
            hip_check(hipMemcpyAsync(d_plane_descs.as<ScanDesc>(), plane_descs.data(), sizeof(ScanDesc) * rounds * n, hipMemcpyHostToDevice, stream));
            hip_check(hipStreamSynchronize(stream)); // plane_descs is reused by the next group
            scans.start();
            {
                ScanDesc proto = plane_descs[0];
                dev::launch_encode(proto, d_plane_descs.as<ScanDesc>(), d_plane_results.as<ScanResult>(), rounds * n, stream);
            }
            scans.stop();
            scan_ms += scans.ms();
            dev::launch_place_plane_scans(slots + first * stream_pitch_bytes, stream_pitch_bytes, d_blob.as<uint8_t>() + prologue_size, sos_size,
                                          rounds, priv, capacity, d_plane_results.as<ScanResult>(),
                                          d_cursors.as<dev::FrameCursorPod>() + first, d_redo.as<uint32_t>() + first, n, stream);
        }
        std::vector<uint32_t> redo(frame_count);
        hip_check(hipMemcpyAsync(redo.data(), d_redo.as<uint32_t>(), sizeof(uint32_t) * frame_count, hipMemcpyDeviceToHost, stream));
        hip_check(hipStreamSynchronize(stream));
        redo_frames.assign(frame_count, 0);
        for (uint32_t i = 0; i < frame_count; ++i)
            redo_frames[i] = redo[i] != 0;
        } // (priv)
    }
    const bool in_rounds = redo_frames.empty();
    for (uint32_t r = 0; r < rounds && in_rounds; ++r)
    {
        for (uint32_t i = 0; i < frame_count; ++i)
        {
            ScanDesc d = base_desc(f, comps_per_scan, p.interleave_mode, p.near_lossless, p.color_transformation, pc,
                                   p.restart_interval);
            d.pixels = const_cast<uint8_t*>(frames) + i * frame_pitch_bytes + (p.interleave_mode == 0 ? r * stride * f.height : 0);
            d.pixel_stride = stride;
            d.line_scratch = d_scratch.as<uint16_t>() + i * scratch_samples;
            descs[i] = d;
        }
        hip_check(hipMemcpyAsync(d_descs.as<ScanDesc>(), descs.data(), sizeof(ScanDesc) * frame_count,
                                 hipMemcpyHostToDevice, stream));
        hip_check(hipStreamSynchronize(stream)); // descs is reused by the next round
        const uint32_t sos_size = static_cast<uint32_t>(sos[r].size());
        dev::launch_place_scan_header(slots, stream_pitch_bytes, d_blob.as<uint8_t>() + blob_offset, sos_size,
                                      d_cursors.as<dev::FrameCursorPod>(), d_descs.as<ScanDesc>(), frame_count, stream);
        scans.start();
        {
            ScanDesc proto = descs[0];
            proto.stream_capacity = stream_pitch_bytes; // upper bound of every frame's remaining capacity
            dev::launch_encode(proto, d_descs.as<ScanDesc>(), d_results.as<ScanResult>(), frame_count, stream);
        }
        scans.stop();
        dev::launch_advance_cursor(d_cursors.as<dev::FrameCursorPod>(), d_results.as<ScanResult>(), sos_size, frame_count,
                                   stream);
        scan_ms += scans.ms();
        blob_offset += sos_size;
    }
    dev::launch_place_epilogue(slots, stream_pitch_bytes, d_cursors.as<dev::FrameCursorPod>(),
                               (p.encoding_options & 1u) != 0, d_sizes.as<uint64_t>(), d_errcs.as<uint32_t>(),
                               frame_count, stream);
    total.stop();
    hip_check(hipMemcpyAsync(sizes, d_sizes.as<uint64_t>(), sizeof(uint64_t) * frame_count, hipMemcpyDeviceToHost, stream));
    static_assert(sizeof(charls_jpegls_errc) == sizeof(uint32_t), "errc size");
    hip_check(hipMemcpyAsync(errcs, d_errcs.as<uint32_t>(), sizeof(uint32_t) * frame_count, hipMemcpyDeviceToHost, stream));
    hip_check(hipStreamSynchronize(stream));
    dev::Timings& t = dev::last_timings();
    t.values[0] = total.ms();
    t.values[1] = scan_ms;
    if (t.count < 2)
        t.count = 2; // the pipeline adds its stage breakdown in values[2..6]
    // frames whose scans could not simply be put in place: scan by scan, with the capacities the reference passes -- every RUN
    // of such frames by one call in rounds (a batch with slots too small for anything flags every frame: one call, not
    // frame_count calls of one frame each); the timings reported are those of this call, not of the re-coding.
    const dev::Timings kept = dev::last_timings();
    for (uint32_t i = 0; i < redo_frames.size();)
    {
        if (!redo_frames[i])
        {
            ++i;
            continue;
        }
        uint32_t last = i + 1;
        while (last < redo_frames.size() && redo_frames[last])
            ++last;
        struct ForceRounds
        {
            ForceRounds() { t_force_rounds = true; }
            ~ForceRounds() { t_force_rounds = false; }
        } force;
        const charls_jpegls_errc rc = charls_amd_encode_batch_device(params, last - i, frames + i * frame_pitch_bytes, frame_pitch_bytes, stride_arg,
                                                                     slots + i * stream_pitch_bytes, stream_pitch_bytes, sizes + i, errcs + i, hip_stream);
        if (rc != CHARLS_JPEGLS_ERRC_SUCCESS)
            return rc;
        i = last;
    }
    if (!redo_frames.empty())
        dev::last_timings() = kept;
    return CHARLS_JPEGLS_ERRC_SUCCESS;
}
catch (...)
{
    return current_exception_to_errc();
}

extern "C" charls_jpegls_errc charls_amd_decode_batch_device(uint32_t frame_count, const void* d_streams,
                                                             size_t stream_pitch_bytes, const uint64_t* sizes,
                                                             void* d_frames, size_t frame_pitch_bytes,
                                                             uint32_t stride_arg, charls_amd_codec_params* params_out,
                                                             charls_jpegls_errc* errcs, void* hip_stream)
try
{
    check_pointer(sizes);
    check_pointer(errcs);
    if (frame_count == 0)
        return CHARLS_JPEGLS_ERRC_SUCCESS;
    check_pointer(d_streams);
    check_pointer(d_frames);
    dev::require_device();
    auto stream = static_cast<hipStream_t>(hip_stream);
    const auto* slots = static_cast<const uint8_t*>(d_streams);
    auto* frames = static_cast<uint8_t*>(d_frames);

    // Every frame's marker segments are parsed on the host by the part-1 reader.  Only a window of each stream is
    // fetched: the first `kWindow` bytes up front, later windows at the position each scan ended.
    constexpr size_t kWindow = 2048;
    struct Frame
    {
        StreamReader reader;
        std::vector<uint8_t> window; // host copy of [window_base, window_base + window.size())
        size_t window_base{};
        size_t cursor{};             // absolute offset of the next unparsed byte
        size_t plane_offset{};       // destination offset of the next scan
        uint32_t decoded_components{};
        charls_jpegls_errc errc{};
        bool done{};
    };
    std::vector<Frame> fr(frame_count);

    auto fetch = [&](Frame& x, uint32_t i, size_t base, size_t bytes) {
        const size_t avail = base < sizes[i] ? static_cast<size_t>(sizes[i]) - base : 0;
        const size_t n = std::min(bytes, avail);
        x.window.resize(n);
        x.window_base = base;
        if (n)
            hip_check(hipMemcpyAsync(x.window.data(), slots + i * stream_pitch_bytes + base, n, hipMemcpyDeviceToHost, stream));
    };
    // the same for many frames at once (bases[k] is the offset of frame which[k]'s window): one gather on the device, one copy
    dev::DeviceBuffer d_specs, d_windows;
    dev::PinnedBuffer h_specs, h_windows;
    auto fetch_many = [&](std::vector<Frame>& set, const std::vector<uint32_t>& which, const std::vector<size_t>& bases) {
        const size_t n = which.size();
        if (n == 0)
            return;
        auto* specs = static_cast<WindowSpec*>(h_specs.ensure(sizeof(WindowSpec) * n));
        for (size_t k = 0; k < n; ++k)
        {
            const uint32_t i = which[k];
            const size_t avail = bases[k] < sizes[i] ? static_cast<size_t>(sizes[i]) - bases[k] : 0;
            specs[k] = WindowSpec{static_cast<uint64_t>(i) * stream_pitch_bytes + bases[k],
                                  static_cast<uint32_t>(std::min(kWindow, avail)), 0};
        }
        d_specs.ensure(sizeof(WindowSpec) * n);
        d_windows.ensure(kWindow * n);
        auto* staged = static_cast<uint8_t*>(h_windows.ensure(kWindow * n));
        hip_check(hipMemcpyAsync(d_specs.as<WindowSpec>(), specs, sizeof(WindowSpec) * n, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(gather_windows, dim3(static_cast<uint32_t>(n)), dim3(256), 0, stream, slots, d_specs.as<const WindowSpec>(),
                           d_windows.as<uint8_t>(), static_cast<uint32_t>(kWindow));
        hip_check(hipGetLastError());
        hip_check(hipMemcpyAsync(staged, d_windows.as<uint8_t>(), kWindow * n, hipMemcpyDeviceToHost, stream));
        hip_check(hipStreamSynchronize(stream));
        for (size_t k = 0; k < n; ++k)
        {
            Frame& x = set[which[k]];
            x.window.assign(staged + k * kWindow, staged + k * kWindow + specs[k].bytes);
            x.window_base = bases[k];
        }
    };
    // A parse that runs off the end of the window while the stream has more bytes is retried with a larger window.
    auto parse = [&](std::vector<Frame>& set, uint32_t i, auto&& body) {
        Frame& x = set[i];
        size_t want = kWindow;
        const StreamReader snapshot = x.reader; // a failed attempt may have advanced the reader's state machine
        for (;;)
        {
            try
            {
                body(x);
                x.errc = CHARLS_JPEGLS_ERRC_SUCCESS;
                return;
            }
            catch (const error& e)
            {
                const bool truncated = x.window_base + x.window.size() < sizes[i];
                if (!truncated || (e.code != CHARLS_JPEGLS_ERRC_NEED_MORE_DATA &&
                                   e.code != CHARLS_JPEGLS_ERRC_INVALID_MARKER_SEGMENT_SIZE &&
                                   e.code != CHARLS_JPEGLS_ERRC_DEFINE_NUMBER_OF_LINES_MARKER_NOT_FOUND))
                {
                    x.errc = e.code;
                    x.done = true;
                    return;
                }
            }
            want *= 8;
            fetch(x, i, x.window_base, want);
            hip_check(hipStreamSynchronize(stream));
            x.reader = snapshot;
        }
    };

    if (stream_pitch_bytes == 0)
        raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE);
    {
        std::vector<uint32_t> all(frame_count);
        for (uint32_t i = 0; i < frame_count; ++i)
        {
            if (sizes[i] > stream_pitch_bytes)
                raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE);
            all[i] = i;
        }
        fetch_many(fr, all, std::vector<size_t>(frame_count, 0));
    }
    for (uint32_t i = 0; i < frame_count; ++i)
        parse(fr, i, [&](Frame& x) {
            x.reader.set_source(x.window.data(), x.window.size());
            x.reader.read_header();
            if (x.reader.end_of_image())
                raise(CHARLS_JPEGLS_ERRC_INVALID_OPERATION); // abbreviated table stream: nothing to decode
            x.cursor = x.window_base + static_cast<size_t>(x.reader.position() - x.window.data());
        });

    dev::DeviceBuffer d_descs, d_results, d_scratch;
    std::vector<ScanDesc> descs;
    std::vector<ScanResult> results;
    std::vector<uint32_t> active;
    EventTimer total(stream);
    double scan_ms = 0;
    uint32_t params_from = UINT32_MAX; // the frame params_out describes so far: the lowest-index frame that got as far as a scan

    // What a frame's scan needs besides its stream position: geometry and coding parameters from the reader (which stands
    // behind the scan's header), the destination of its first plane, the line scratch (an offset until run_scans places it).
    auto scan_desc = [&](uint32_t i, Frame& x, size_t& scratch_total) -> ScanDesc {
        const charls_frame_info& f = x.reader.frame_info();
        const int32_t ilv = x.reader.scan_interleave_mode();
        const uint32_t nc = x.reader.scan_component_count();
        const size_t row = (ilv == 0 ? 1u : nc) * static_cast<size_t>(f.width) * bytes_per_sample(f.bits_per_sample);
        size_t stride = stride_arg;
        if (stride == 0)
            stride = row;
        else if (stride < row)
            raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_STRIDE);
        const size_t need = (ilv == 0 ? stride * nc * f.height : stride * f.height) - (stride - row);
        if (frame_pitch_bytes < x.plane_offset + need)
            raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE);
        ScanDesc d = base_desc(f, static_cast<int32_t>(nc), ilv, x.reader.parameters().near_lossless,
                               x.reader.parameters().transformation, x.reader.validated_pc(),
                               x.reader.parameters().restart_interval);
        d.pixels = frames + i * frame_pitch_bytes + x.plane_offset;
        d.pixel_stride = stride;
        d.stream = const_cast<uint8_t*>(slots) + i * stream_pitch_bytes + x.cursor;
        d.stream_capacity = sizes[i] - x.cursor;
        d.line_scratch = reinterpret_cast<uint16_t*>(scratch_total); // placed by run_scans
        scratch_total += dev::line_scratch_samples(f.width, ilv, static_cast<int32_t>(nc)) * sizeof(uint16_t);
        scratch_total = (scratch_total + 255) & ~size_t{255};
        return d;
    };
    auto report_params = [&](Frame& x, uint32_t index) {
        if (params_out && index < params_from)
        {
            *params_out = charls_amd_codec_params{x.reader.frame_info(), x.reader.parameters().near_lossless,
                                                  x.reader.scan_interleave_mode(), x.reader.parameters().transformation,
                                                  x.reader.preset_coding_parameters(), 0, x.reader.parameters().restart_interval};
            params_from = index;
        }
    };
    // Decodes descs[0, n) (tags[k] says whose scan descs[k] is; both are permuted: scans that can share a kernel
    // specialisation are made contiguous, one launch per group) and leaves results[k] beside descs[k].
    auto run_scans = [&](std::vector<uint32_t>& tags, size_t scratch_total) {
        const uint32_t n = static_cast<uint32_t>(descs.size());
        auto* scratch = static_cast<uint8_t*>(d_scratch.ensure(scratch_total));
        for (ScanDesc& d : descs)
            d.line_scratch = reinterpret_cast<uint16_t*>(scratch + reinterpret_cast<size_t>(d.line_scratch));
        {
            std::vector<uint32_t> order(n);
            for (uint32_t k = 0; k < n; ++k)
                order[k] = k;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
                return dev::decode_launch_key(descs[a]) < dev::decode_launch_key(descs[b]);
            });
            std::vector<ScanDesc> d2(n);
            std::vector<uint32_t> t2(n);
            for (uint32_t k = 0; k < n; ++k)
            {
                d2[k] = descs[order[k]];
                t2[k] = tags[order[k]];
            }
            descs = std::move(d2);
            tags = std::move(t2);
        }
        d_descs.ensure(sizeof(ScanDesc) * n);
        d_results.ensure(sizeof(ScanResult) * n);
        results.resize(n);
        hip_check(hipMemcpyAsync(d_descs.as<ScanDesc>(), descs.data(), sizeof(ScanDesc) * n, hipMemcpyHostToDevice, stream));
        total.start();
        for (uint32_t first = 0; first < n;)
        {
            uint32_t last = first + 1;
            while (last < n && dev::decode_launch_key(descs[last]) == dev::decode_launch_key(descs[first]))
                ++last;
            dev::launch_decode(descs[first], d_descs.as<ScanDesc>() + first, d_results.as<ScanResult>() + first, last - first,
                               stream);
            first = last;
        }
        total.stop();
        hip_check(hipMemcpyAsync(results.data(), d_results.as<ScanResult>(), sizeof(ScanResult) * n, hipMemcpyDeviceToHost, stream));
        hip_check(hipStreamSynchronize(stream)); // (descs and results are reused by the next call)
        scan_ms += total.ms();
    };

    // ---- Planar frames: the component scans of all of them in ONE launch (the decoder's rate is the number of scans in
    // flight: scan by scan, a batch of RGB planes ran three rounds of a third of its scans).  Where scan c + 1 starts is only
    // known once scan c has been decoded -- or once the marker that ends scan c has been found: inside an entropy-coded
    // segment no 0xFF is followed by a byte with its high bit set, so a search finds it (find_scan_end), and the part-1 reader
    // parses on from there, on a COPY of the frame's state.  The copy replaces the frame only if every scan then ends exactly
    // at the marker found behind it and the end of the image parses; any frame that does not is decoded again by the rounds
    // below, which report whatever part 1 reports for it.  src/charls_jpegls_decoder.cpp:211-236 is the scan-by-scan loop.
    if (knobs::get_or(knobs::kBatchRounds, 0) == 0)
    {
        struct Plan
        {
            uint32_t frame;
            std::vector<ScanDesc> scans;   // (line scratch: an offset until run_scans places it)
            std::vector<size_t> starts;    // first byte of scan c
            std::vector<size_t> marker_at; // the marker found behind scan c
            bool ok{true};
        };
        std::vector<Frame> probe;
        std::vector<Plan> plans;
        size_t scratch_total = 0;
        for (uint32_t i = 0; i < frame_count; ++i)
        {
            const Frame& x = fr[i];
            if (x.done || x.reader.component_count() < 2 || x.reader.scan_interleave_mode() != 0 || x.reader.scan_component_count() != 1)
                continue;
            if (probe.empty())
                probe = fr; // (readers, windows and cursors of every frame; only the planned ones are touched)
            Plan p;
            p.frame = i;
            try
            { // every plane must fit behind the first (the rounds raise the error where it belongs otherwise)
                const charls_frame_info& f = x.reader.frame_info();
                const size_t row = static_cast<size_t>(f.width) * bytes_per_sample(f.bits_per_sample);
                const size_t stride = stride_arg == 0 ? row : stride_arg;
                if (stride < row || frame_pitch_bytes < checked_mul(checked_mul(stride, f.height), x.reader.component_count()) - (stride - row))
                    continue;
                p.scans.push_back(scan_desc(i, probe[i], scratch_total));
            }
            catch (const error&)
            {
                continue;
            }
            p.starts.push_back(x.cursor);
            plans.push_back(std::move(p));
        }
        dev::DeviceBuffer d_search, d_found;
        dev::PinnedBuffer h_search, h_found;
        for (uint32_t c = 1; !plans.empty(); ++c)
        {
            std::vector<uint32_t> need; // plans whose frame has a scan c
            for (uint32_t k = 0; k < plans.size(); ++k)
                if (plans[k].ok && probe[plans[k].frame].reader.component_count() > c)
                    need.push_back(k);
            if (need.empty())
                break;
            const size_t n = need.size();
            auto* search = static_cast<MarkerSearch*>(h_search.ensure(sizeof(MarkerSearch) * n));
            auto* found = static_cast<unsigned long long*>(h_found.ensure(sizeof(unsigned long long) * n));
            for (size_t q = 0; q < n; ++q)
            {
                const uint32_t i = plans[need[q]].frame;
                search[q] = MarkerSearch{static_cast<uint64_t>(i) * stream_pitch_bytes + plans[need[q]].starts.back(),
                                         static_cast<uint64_t>(i) * stream_pitch_bytes + sizes[i]};
            }
            d_search.ensure(sizeof(MarkerSearch) * n);
            d_found.ensure(sizeof(unsigned long long) * n);
            hip_check(hipMemcpyAsync(d_search.as<MarkerSearch>(), search, sizeof(MarkerSearch) * n, hipMemcpyHostToDevice, stream));
            dev::launch_find_scan_end(slots, d_search.as<const MarkerSearch>(), d_found.as<unsigned long long>(), static_cast<uint32_t>(n), stream);
            hip_check(hipMemcpyAsync(found, d_found.as<unsigned long long>(), sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, stream));
            hip_check(hipStreamSynchronize(stream));
            std::vector<uint32_t> which;
            std::vector<size_t> bases;
            for (size_t q = 0; q < n; ++q)
            {
                Plan& p = plans[need[q]];
                if (found[q] == kNoMarker)
                {
                    p.ok = false;
                    continue;
                }
                p.marker_at.push_back(static_cast<size_t>(found[q] - static_cast<uint64_t>(p.frame) * stream_pitch_bytes));
                which.push_back(p.frame);
                bases.push_back(p.marker_at.back());
            }
            fetch_many(probe, which, bases);
            for (size_t q = 0; q < n; ++q)
            {
                Plan& p = plans[need[q]];
                if (!p.ok)
                    continue;
                Frame& y = probe[p.frame];
                const charls_frame_info f = y.reader.frame_info();
                const size_t row = static_cast<size_t>(f.width) * bytes_per_sample(f.bits_per_sample);
                y.plane_offset += (stride_arg ? stride_arg : row) * f.height;
                parse(probe, p.frame, [&](Frame& z) {
                    z.reader.continue_on_window(z.window.data(), z.window.size());
                    z.reader.read_next_start_of_scan();
                    z.cursor = z.window_base + static_cast<size_t>(z.reader.position() - z.window.data());
                });
                if (y.done || y.reader.scan_interleave_mode() != 0 || y.reader.scan_component_count() != 1)
                {
                    p.ok = false;
                    continue;
                }
                try
                {
                    p.scans.push_back(scan_desc(p.frame, y, scratch_total));
                    p.starts.push_back(y.cursor);
                }
                catch (const error&)
                {
                    p.ok = false;
                }
            }
        }
        descs.clear();
        std::vector<uint32_t> tags; // plan << 8 | component
        for (uint32_t k = 0; k < plans.size(); ++k)
        {
            Plan& p = plans[k];
            p.ok = p.ok && p.scans.size() == probe[p.frame].reader.component_count() && p.scans.size() < 256;
            for (uint32_t c = 0; p.ok && c < p.scans.size(); ++c)
            {
                descs.push_back(p.scans[c]);
                tags.push_back((k << 8) | c);
            }
        }
        if (!descs.empty())
        {
            run_scans(tags, scratch_total);
            std::vector<std::vector<size_t>> used(plans.size());
            for (Plan& p : plans)
                used[&p - plans.data()].assign(p.scans.size(), 0);
            for (uint32_t k = 0; k < tags.size(); ++k)
            {
                Plan& p = plans[tags[k] >> 8];
                const uint32_t c = tags[k] & 0xFFu;
                if (results[k].errc != kOk)
                    p.ok = false;
                used[tags[k] >> 8][c] = static_cast<size_t>(results[k].bytes);
            }
            std::vector<uint32_t> which;
            std::vector<size_t> bases;
            for (uint32_t k = 0; k < plans.size(); ++k)
            {
                Plan& p = plans[k];
                for (uint32_t c = 0; p.ok && c + 1 < p.scans.size(); ++c)
                    p.ok = p.starts[c] + used[k][c] == p.marker_at[c]; // the scan ends exactly at the marker found behind it
                if (!p.ok)
                    continue;
                probe[p.frame].cursor = p.starts.back() + used[k].back();
                which.push_back(p.frame);
                bases.push_back(probe[p.frame].cursor);
            }
            fetch_many(probe, which, bases);
            for (Plan& p : plans)
            {
                if (!p.ok)
                    continue;
                parse(probe, p.frame, [&](Frame& z) {
                    z.reader.continue_on_window(z.window.data(), z.window.size());
                    z.reader.read_end_of_image();
                    z.cursor = z.window_base + static_cast<size_t>(z.reader.position() - z.window.data());
                });
                Frame& y = probe[p.frame];
                if (y.errc != CHARLS_JPEGLS_ERRC_SUCCESS)
                    continue; // (the rounds decode it again and report this)
                report_params(fr[p.frame], p.frame);
                y.decoded_components = static_cast<uint32_t>(p.scans.size());
                y.done = true;
                fr[p.frame] = std::move(y);
            }
        }
    }


    for (;;)
    {
        active.clear();
        descs.clear();
        size_t scratch_total = 0;
        for (uint32_t i = 0; i < frame_count; ++i)
        {
            Frame& x = fr[i];
            if (x.done)
                continue;
            try
            {
                const ScanDesc d = scan_desc(i, x, scratch_total);
                report_params(x, i);
                descs.push_back(d);
                active.push_back(i);
            }
            catch (const error& e)
            {
                x.errc = e.code;
                x.done = true;
            }
        }
        if (active.empty())
            break;
        const uint32_t n = static_cast<uint32_t>(active.size());
        run_scans(active, scratch_total);

        // advance every frame past its scan; fetch the bytes that follow (next SOS or EOI)
        {
            std::vector<uint32_t> which;
            std::vector<size_t> bases;
            which.reserve(n);
            bases.reserve(n);
            for (uint32_t k = 0; k < n; ++k)
            {
                Frame& x = fr[active[k]];
                if (results[k].errc != kOk)
                {
                    x.errc = static_cast<charls_jpegls_errc>(results[k].errc);
                    x.done = true;
                    continue;
                }
                x.cursor += results[k].bytes;
                which.push_back(active[k]);
                bases.push_back(x.cursor);
            }
            fetch_many(fr, which, bases);
        }
        for (uint32_t k = 0; k < n; ++k)
        {
            const uint32_t i = active[k];
            Frame& x = fr[i];
            if (x.done)
                continue;
            const charls_frame_info f = x.reader.frame_info();
            const uint32_t nc = x.reader.scan_component_count();
            const int32_t ilv = x.reader.scan_interleave_mode();
            x.decoded_components += nc;
            const bool last = x.decoded_components == x.reader.component_count();
            if (!last)
            {
                const size_t row = (ilv == 0 ? 1u : nc) * static_cast<size_t>(f.width) * bytes_per_sample(f.bits_per_sample);
                const size_t stride = stride_arg ? stride_arg : row;
                x.plane_offset += stride * f.height;
            }
            parse(fr, i, [&](Frame& y) {
                // continue the same reader on the freshly fetched window
                y.reader.continue_on_window(y.window.data(), y.window.size());
                if (last)
                    y.reader.read_end_of_image();
                else
                    y.reader.read_next_start_of_scan();
                y.cursor = y.window_base + static_cast<size_t>(y.reader.position() - y.window.data());
            });
            if (last)
                x.done = true;
        }
    }
    for (uint32_t i = 0; i < frame_count; ++i)
        errcs[i] = fr[i].errc;
    dev::Timings& t = dev::last_timings();
    t.values[0] = scan_ms;
    t.values[1] = scan_ms;
    t.count = 2;
    return CHARLS_JPEGLS_ERRC_SUCCESS;
}
catch (...)
{
    return current_exception_to_errc();
}
