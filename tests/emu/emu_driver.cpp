// TEST-ONLY driver: exposes the gfx950 kernels, compiled for the host, to the CPU test-suite (tests/test_emu_*.py).
#include "emu_launch.h"

namespace emu {
BlockState* g_block = nullptr;
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
} // namespace emu

#include "../../charls_amd/csrc/device/scan_serial.hip"
#include "../../charls_amd/csrc/device/pipeline_common.hip"
#include "../../charls_amd/csrc/device/block_stuffing.hip"
#include "../../charls_amd/csrc/device/speculative_stuffing.hip"
#include "../../charls_amd/csrc/device/scan_fast_decode.hip"
#include "../../charls_amd/csrc/device/scan_group_decode.hip"
#include "../../charls_amd/csrc/device/scan_group_pixels.hip"
#include "../../charls_amd/csrc/device/scan_group_encode.hip"
#include "../../charls_amd/csrc/device/restart_intervals.hip"
#include "../../charls_amd/csrc/device/container_kernels.hip"
#include "emu_tile_pipeline.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {

void emu_encode_scans_serial(const jls::ScanDesc* descs, jls::ScanResult* results, int count)
{
    emu::launch(jls::encode_scans_serial, dim3(count), dim3(64), 0, descs, results);
}

void emu_decode_scans_serial(const jls::ScanDesc* descs, jls::ScanResult* results, int count)
{
    emu::launch(jls::decode_scans_serial, dim3(count), dim3(64), 0, descs, results);
}

// NC co-sited components / sample width chosen exactly as the host dispatcher does (runtime.hip: launch_decode).
int emu_decode_scans_wave(const jls::ScanDesc* descs, jls::ScanResult* results, int count)
{
    const jls::ScanDesc& d = descs[0];
    const int planes = d.interleave_mode == 0 ? 1 : d.components;
    const bool wide = d.bits_per_sample > 8;
    const size_t lds = jls::wave::kFixedLds + (size_t)planes * (d.width + 2) * (wide ? 2 : 1);
    const int nc = d.interleave_mode == 2 ? d.components : 1;
#define EMU_CASE(S, N) emu::launch(jls::decode_scans_wave<S, N>, dim3(count), dim3(64), lds, descs, results)
    if (!wide)
    {
        if (nc == 1) EMU_CASE(uint8_t, 1); else if (nc == 2) EMU_CASE(uint8_t, 2); else if (nc == 3) EMU_CASE(uint8_t, 3); else EMU_CASE(uint8_t, 4);
    }
    else
    {
        if (nc == 1) EMU_CASE(uint16_t, 1); else if (nc == 2) EMU_CASE(uint16_t, 2); else if (nc == 3) EMU_CASE(uint16_t, 3); else EMU_CASE(uint16_t, 4);
    }
#undef EMU_CASE
    return 0;
}

} // extern "C"

extern "C" {

int emu_decode_scans_fast(const jls::ScanDesc* descs, jls::ScanResult* results, int count)
{
    const jls::ScanDesc& d = descs[0];
    const bool wide = d.bits_per_sample > 8;
    const size_t line_bytes = (((size_t)d.width + 2) * (wide ? 2 : 1) + 3) & ~size_t{3};
    const size_t lds = (wide ? jls::fast::fixed_lds<uint16_t>() : jls::fast::fixed_lds<uint8_t>()) + line_bytes;
    if (wide)
        emu::launch(jls::decode_scans_fast<uint16_t>, dim3(count), dim3(64), lds, descs, results);
    else
        emu::launch(jls::decode_scans_fast<uint8_t>, dim3(count), dim3(64), lds, descs, results);
    return 0;
}

// Stage E alone on a given raw bit stream (`raw_bytes` readable and zero behind the stream): stuff_scan or, with `blocks`,
// the block-parallel form; both must produce the same bytes and result words.
void emu_stuff_raw(const uint8_t* raw, uint64_t total_bits, uint64_t raw_bytes, uint8_t* out, uint64_t capacity, int blocks,
                   jls::ScanResult* result)
{
    using namespace jls;
    std::vector<uint32_t> words(raw_bytes / 4 + 32, 0);
    std::memcpy(words.data(), raw, (size_t)raw_bytes);
    uint64_t bits = total_bits;
    uint32_t status = 0;
    const pipe::SpecGeometry spec = pipe::stuff_spec_geometry();
    std::vector<uint32_t> tables(std::max((size_t)(raw_bytes / pipe::kStuffChunk + 2) * pipe::kStuffWords,
                                          pipe::stuff_spec_table_words(raw_bytes, spec.chunk_bytes)), 0xA5A5A5A5u);
    pipe::Work w{};
    w.raw = words.data();
    w.raw_words = (uint32_t)(raw_bytes / 4);
    w.total_bits = &bits;
    w.status = &status;
    w.stuff_tables = tables.data();
    ScanDesc d{};
    d.stream = out;
    d.stream_capacity = capacity;
    if (blocks == 2)
    { // the speculative form (speculative_stuffing.hip), grids as in runtime.hip
        const unsigned waves = (unsigned)(raw_bytes / spec.chunk_bytes + 1);
        emu::launch(pipe::stuff_spec_survey, dim3(waves, 1), dim3(64), 0, (const pipe::Work*)&w, spec.chunk_bytes, spec.warm_bytes);
        emu::launch(pipe::stuff_spec_resolve, dim3(1), dim3(64), 0, (const pipe::Work*)&w, spec.chunk_bytes);
        emu::launch(pipe::stuff_spec_emit, dim3(waves, 1), dim3(64), 0, (const ScanDesc*)&d, (const pipe::Work*)&w, result, spec.chunk_bytes);
    }
    else if (blocks)
    {
        const unsigned chunk_waves = (unsigned)((raw_bytes / pipe::kStuffChunk + 1 + 63) / 64);
        const unsigned survey_blocks = pipe::stuff_survey_blocks(raw_bytes);
        emu::launch(pipe::stuff_survey, dim3(survey_blocks, 1), dim3(64), 0, (const pipe::Work*)&w);
        emu::launch(pipe::stuff_resolve, dim3(1), dim3(pipe::kStuffResolveThreads), 0, (const pipe::Work*)&w);
        emu::launch(pipe::stuff_emit, dim3(chunk_waves, 1), dim3(64), 0, (const ScanDesc*)&d, (const pipe::Work*)&w, result);
    }
    else
        emu::launch(pipe::stuff_scan, dim3(1), dim3(64), 0, (const ScanDesc*)&d, (const pipe::Work*)&w, result);
}

// scan_group_decode.hip: `group` lanes per scan, 64 / group scans per workgroup (all scans share descs[0]'s geometry).
int emu_decode_scans_group(const jls::ScanDesc* descs, jls::ScanResult* results, int count, int group)
{
    const jls::ScanDesc& d = descs[0];
    const bool wide = d.bits_per_sample > 8;
    const int per_wave = 64 / group;
    const int nl = d.interleave_mode == 1 ? d.components : 1; // a line-interleaved scan keeps one line per component
    const size_t lds = wide ? jls::grp::workgroup_lds_bytes<uint16_t>(d.width, per_wave, nl) : jls::grp::workgroup_lds_bytes<uint8_t>(d.width, per_wave, nl);
    const dim3 grid((count + per_wave - 1) / per_wave);
#define EMU_GROUP_N(S, G, N) emu::launch(jls::decode_scans_group<S, G, N>, grid, dim3(64), lds, descs, results, (uint32_t)count)
#define EMU_GROUP(S, G)                                     \
    do                                                      \
    {                                                       \
        if (nl == 1) EMU_GROUP_N(S, G, 1);                  \
        else if (nl == 2) EMU_GROUP_N(S, G, 2);             \
        else if (nl == 3) EMU_GROUP_N(S, G, 3);             \
        else EMU_GROUP_N(S, G, 4);                          \
    } while (0)
    if (d.near_lossless != 0)
    { // decode_scans_group<S, G, NL, 1, true>: the product's near-lossless instantiations (runtime.hip: launch_decode_plain)
#define EMU_GROUP_NEAR(S, G)                                                                                                              \
    do                                                                                                                                    \
    {                                                                                                                                     \
        if (nl == 1) emu::launch(jls::decode_scans_group<S, G, 1, 1, true>, grid, dim3(64), lds, descs, results, (uint32_t)count);        \
        else if (nl == 2) emu::launch(jls::decode_scans_group<S, G, 2, 1, true>, grid, dim3(64), lds, descs, results, (uint32_t)count);   \
        else if (nl == 3) emu::launch(jls::decode_scans_group<S, G, 3, 1, true>, grid, dim3(64), lds, descs, results, (uint32_t)count);   \
        else emu::launch(jls::decode_scans_group<S, G, 4, 1, true>, grid, dim3(64), lds, descs, results, (uint32_t)count);                \
    } while (0)
        if (group == 8)
        {
            if (wide) EMU_GROUP_NEAR(uint16_t, 8); else EMU_GROUP_NEAR(uint8_t, 8);
        }
        else if (group == 16)
        {
            if (wide) EMU_GROUP_NEAR(uint16_t, 16); else EMU_GROUP_NEAR(uint8_t, 16);
        }
        else if (group == 32)
        {
            if (wide) EMU_GROUP_NEAR(uint16_t, 32); else EMU_GROUP_NEAR(uint8_t, 32);
        }
        else
            return -1;
#undef EMU_GROUP_NEAR
    }
    else if (group == 4)
    {
        if (wide) EMU_GROUP(uint16_t, 4); else EMU_GROUP(uint8_t, 4);
    }
    else if (group == 8)
    {
        if (wide) EMU_GROUP(uint16_t, 8); else EMU_GROUP(uint8_t, 8);
    }
    else if (group == 16)
    {
        if (wide) EMU_GROUP(uint16_t, 16); else EMU_GROUP(uint8_t, 16);
    }
    else if (group == 32)
    {
        if (wide) EMU_GROUP(uint16_t, 32); else EMU_GROUP(uint8_t, 32);
    }
    else
        return -1;
#undef EMU_GROUP
#undef EMU_GROUP_N
    return 0;
}

// The same kernel with W wavefronts per workgroup (one workgroup per CU, a wavefront per SIMD: the launches of big batches).
int emu_decode_scans_group_waves(const jls::ScanDesc* descs, jls::ScanResult* results, int count, int group, int waves)
{
    const jls::ScanDesc& d = descs[0];
    const bool wide = d.bits_per_sample > 8;
    if ((group != 16 && group != 32) || (waves != 4 && waves != 8) || d.interleave_mode == 1)
        return -1;
    const int per_group = 64 / group * waves;
    const size_t lds = wide ? jls::grp::workgroup_lds_bytes<uint16_t>(d.width, per_group, 1) : jls::grp::workgroup_lds_bytes<uint8_t>(d.width, per_group, 1);
    const dim3 grid((count + per_group - 1) / per_group);
#define EMU_GROUP_W(S, G, W) emu::launch(jls::decode_scans_group<S, G, 1, W>, grid, dim3(64 * W), lds, descs, results, (uint32_t)count)
#define EMU_GROUP_W_NEAR(S, G) emu::launch(jls::decode_scans_group<S, G, 1, 4, true>, grid, dim3(64 * 4), lds, descs, results, (uint32_t)count)
    if (d.near_lossless != 0)
    {
        if (waves != 4)
            return -1;
        if (group == 16)
        {
            if (wide) EMU_GROUP_W_NEAR(uint16_t, 16); else EMU_GROUP_W_NEAR(uint8_t, 16);
        }
        else
        {
            if (wide) EMU_GROUP_W_NEAR(uint16_t, 32); else EMU_GROUP_W_NEAR(uint8_t, 32);
        }
    }
    else if (group == 16 && waves == 4)
    {
        if (wide) EMU_GROUP_W(uint16_t, 16, 4); else EMU_GROUP_W(uint8_t, 16, 4);
    }
    else if (group == 32 && waves == 4)
    {
        if (wide) EMU_GROUP_W(uint16_t, 32, 4); else EMU_GROUP_W(uint8_t, 32, 4);
    }
    else if (group == 16)
    {
        if (wide) EMU_GROUP_W(uint16_t, 16, 8); else EMU_GROUP_W(uint8_t, 16, 8);
    }
    else
    {
        if (wide) EMU_GROUP_W(uint16_t, 32, 8); else EMU_GROUP_W(uint8_t, 32, 8);
    }
#undef EMU_GROUP_W
#undef EMU_GROUP_W_NEAR
    return 0;
}

// scan_group_pixels.hip: sample-interleaved scans, `group` lanes per scan.
int emu_decode_pixels_group(const jls::ScanDesc* descs, jls::ScanResult* results, int count, int group)
{
    const jls::ScanDesc& d = descs[0];
    const bool wide = d.bits_per_sample > 8;
    const int per_wave = 64 / group;
    const int nc = d.interleave_mode == 2 ? d.components : 1;
    const int nl = d.interleave_mode == 1 ? d.components : 1;
    const size_t lds = wide ? jls::grp::pixel_workgroup_lds_bytes<uint16_t>(d.width, nc, per_wave, nl)
                            : jls::grp::pixel_workgroup_lds_bytes<uint8_t>(d.width, nc, per_wave, nl);
    const dim3 grid((count + per_wave - 1) / per_wave);
#define EMU_PIXELS_L(S, G, NLINES) emu::launch(jls::decode_pixels_group<S, G, 1, NLINES>, grid, dim3(64), lds, descs, results, (uint32_t)count)
#define EMU_PIXELS_LG(S, NLINES)                                 \
    do                                                           \
    {                                                            \
        if (group == 8) EMU_PIXELS_L(S, 8, NLINES);              \
        else if (group == 16) EMU_PIXELS_L(S, 16, NLINES);       \
        else if (group == 32) EMU_PIXELS_L(S, 32, NLINES);       \
        else return -1;                                          \
    } while (0)
    if (nl > 1)
    {
        if (!wide)
        {
            if (nl == 2) EMU_PIXELS_LG(uint8_t, 2); else if (nl == 3) EMU_PIXELS_LG(uint8_t, 3); else EMU_PIXELS_LG(uint8_t, 4);
        }
        else
        {
            if (nl == 2) EMU_PIXELS_LG(uint16_t, 2); else if (nl == 3) EMU_PIXELS_LG(uint16_t, 3); else EMU_PIXELS_LG(uint16_t, 4);
        }
        return 0;
    }
#undef EMU_PIXELS_LG
#undef EMU_PIXELS_L
#define EMU_PIXELS(S, G, N) emu::launch(jls::decode_pixels_group<S, G, N>, grid, dim3(64), lds, descs, results, (uint32_t)count)
#define EMU_PIXELS_G(S, N)                                       \
    do                                                           \
    {                                                            \
        if (group == 8) EMU_PIXELS(S, 8, N);                     \
        else if (group == 16) EMU_PIXELS(S, 16, N);              \
        else if (group == 32) EMU_PIXELS(S, 32, N);              \
        else return -1;                                          \
    } while (0)
    if (!wide)
    {
        if (nc == 1) EMU_PIXELS_G(uint8_t, 1); else if (nc == 2) EMU_PIXELS_G(uint8_t, 2); else if (nc == 3) EMU_PIXELS_G(uint8_t, 3); else if (nc == 4) EMU_PIXELS_G(uint8_t, 4); else return -1;
    }
    else
    {
        if (nc == 1) EMU_PIXELS_G(uint16_t, 1); else if (nc == 2) EMU_PIXELS_G(uint16_t, 2); else if (nc == 3) EMU_PIXELS_G(uint16_t, 3); else if (nc == 4) EMU_PIXELS_G(uint16_t, 4); else return -1;
    }
#undef EMU_PIXELS_G
#undef EMU_PIXELS
    return 0;
}

// scan_group_encode.hip: the encoder for near-lossless (and any other) single-component or sample-interleaved scans.
int emu_encode_pixels_group(const jls::ScanDesc* descs, jls::ScanResult* results, int count, int group)
{
    const jls::ScanDesc& d = descs[0];
    const bool wide = d.bits_per_sample > 8;
    const int per_wave = 64 / group;
    const int nc = d.interleave_mode == 2 ? d.components : 1;
    const int nl = d.interleave_mode == 1 ? d.components : 1;
    const size_t lds = wide ? jls::grp::encode_workgroup_lds_bytes<uint16_t>(d.width, nc, per_wave, nl) : jls::grp::encode_workgroup_lds_bytes<uint8_t>(d.width, nc, per_wave, nl);
    const dim3 grid((count + per_wave - 1) / per_wave);
#define EMU_ENC_L(S, G, NLINES) emu::launch(jls::encode_pixels_group<S, G, 1, NLINES>, grid, dim3(64), lds, descs, results, (uint32_t)count)
#define EMU_ENC_LG(S, NLINES)                                  \
    do                                                         \
    {                                                          \
        if (group == 8) EMU_ENC_L(S, 8, NLINES);               \
        else if (group == 16) EMU_ENC_L(S, 16, NLINES);        \
        else if (group == 32) EMU_ENC_L(S, 32, NLINES);        \
        else if (group == 64) EMU_ENC_L(S, 64, NLINES);        \
        else return -1;                                        \
    } while (0)
    if (nl > 1)
    {
        if (!wide)
        {
            if (nl == 2) EMU_ENC_LG(uint8_t, 2); else if (nl == 3) EMU_ENC_LG(uint8_t, 3); else EMU_ENC_LG(uint8_t, 4);
        }
        else
        {
            if (nl == 2) EMU_ENC_LG(uint16_t, 2); else if (nl == 3) EMU_ENC_LG(uint16_t, 3); else EMU_ENC_LG(uint16_t, 4);
        }
        return 0;
    }
#undef EMU_ENC_LG
#undef EMU_ENC_L
#define EMU_ENC(S, G, N) emu::launch(jls::encode_pixels_group<S, G, N>, grid, dim3(64), lds, descs, results, (uint32_t)count)
#define EMU_ENC_G(S, N)                                        \
    do                                                         \
    {                                                          \
        if (group == 8) EMU_ENC(S, 8, N);                      \
        else if (group == 16) EMU_ENC(S, 16, N);               \
        else if (group == 32) EMU_ENC(S, 32, N);               \
        else if (group == 64) EMU_ENC(S, 64, N);               \
        else return -1;                                        \
    } while (0)
    if (!wide)
    {
        if (nc == 1) EMU_ENC_G(uint8_t, 1); else if (nc == 2) EMU_ENC_G(uint8_t, 2); else if (nc == 3) EMU_ENC_G(uint8_t, 3); else EMU_ENC_G(uint8_t, 4);
    }
    else
    {
        if (nc == 1) EMU_ENC_G(uint16_t, 1); else if (nc == 2) EMU_ENC_G(uint16_t, 2); else if (nc == 3) EMU_ENC_G(uint16_t, 3); else EMU_ENC_G(uint16_t, 4);
    }
#undef EMU_ENC_G
#undef EMU_ENC
    return 0;
}

// restart_intervals.hip, decode side: marker search, interval descriptors, and (given the intervals' results) the check.
void emu_find_restart_markers(const jls::ScanDesc* descs, uint32_t* marks, uint32_t max_marks, uint32_t* counts, int count)
{
    emu::launch(jls::interval::find_restart_markers, dim3(count), dim3(64), 0, descs, marks, max_marks, counts);
}

void emu_build_decode_intervals(const jls::ScanDesc* parents, const uint32_t* marks, uint32_t intervals, jls::ScanDesc* subs,
                                int count)
{
    emu::launch(jls::interval::build_decode_intervals, dim3(intervals, count), dim3(1), 0, parents, marks, intervals, subs);
}

void emu_check_intervals(const jls::ScanDesc* parents, const uint32_t* marks, uint32_t intervals,
                         const jls::ScanResult* sub_results, jls::ScanResult* results, int count)
{
    emu::launch(jls::interval::check_intervals, dim3((count + 63) / 64), dim3(64), 0, parents, marks, intervals, sub_results,
                results, (uint32_t)count);
}

// encode side: the join of already coded intervals
void emu_join_intervals(const jls::ScanDesc* parents, const jls::ScanDesc* subs, uint32_t intervals,
                        const jls::ScanResult* sub_results, uint64_t* offsets, jls::ScanResult* results, int count)
{
    emu::launch(jls::interval::plan_join, dim3((count + 63) / 64), dim3(64), 0, parents, intervals, sub_results, offsets,
                results, (uint32_t)count);
    emu::launch(jls::interval::join_intervals, dim3(intervals, count), dim3(256), 0, parents, subs, intervals, sub_results,
                (const uint64_t*)offsets, (const jls::ScanResult*)results);
}

// The whole restart-interval encode of runtime.hip (launch_encode_intervals) for lossless scans, on the host: interval
// descriptors, the pipeline on the intervals, the join.
void emu_encode_with_restart_intervals(const jls::ScanDesc* parents, jls::ScanResult* results, int count, uint64_t capacity)
{
    using namespace jls;
    const uint32_t lines = parents[0].restart_interval;
    const uint32_t intervals = (parents[0].height + lines - 1) / lines;
    const size_t subs_n = (size_t)count * intervals;
    std::vector<ScanDesc> subs(subs_n);
    std::vector<ScanResult> sub_results(subs_n);
    std::vector<uint64_t> offsets(subs_n);
    std::vector<uint8_t> buffers(capacity * subs_n + 64, 0xA5);
    emu::launch(interval::build_encode_intervals, dim3(intervals, count), dim3(1), 0, parents, intervals, buffers.data(),
                capacity, (uint16_t*)nullptr, (uint64_t)0, subs.data());
    // the last interval of a scan may be shorter: the pipeline takes its geometry per scan, as in the product
    for (size_t i = 0; i < subs_n; ++i)
    {
        if (subs[i].bits_per_sample > 8)
            emu_tile_pipeline<uint16_t>(&subs[i], &sub_results[i], 1, 256, 128, 2048, 2048, 32768);
        else
            emu_tile_pipeline<uint8_t>(&subs[i], &sub_results[i], 1, 256, 128, 2048, 2048, 32768);
    }
    emu_join_intervals(parents, subs.data(), intervals, sub_results.data(), offsets.data(), results, count);
}

// container_kernels.hip: the marker that ends an entropy-coded segment (the batch decoder's way from one component scan of
// a planar frame to the next), for `count` stretches of one buffer.
void emu_find_scan_end(const uint8_t* slots, const uint64_t* from_end_pairs, unsigned long long* found, int count)
{
    emu::launch(jls::find_scan_end, dim3(count), dim3(256), 0, slots, reinterpret_cast<const jls::MarkerSearch*>(from_end_pairs), found);
}

// container_kernels.hip: the component scans of planar frames, coded into private buffers, put in place behind their SOS
// headers (place_plane_scans + advance_plane_cursors).  cursors: {offset, errc} pairs of 64 bits + 2 x 32 bits per frame.
void emu_place_plane_scans(uint8_t* slots, uint64_t slot_pitch, const uint8_t* headers, uint32_t header_size, uint32_t rounds,
                           const uint8_t* private_streams, uint64_t capacity, const jls::ScanResult* results, void* cursors,
                           uint32_t* redo, uint32_t frames)
{
    auto* c = static_cast<jls::FrameCursor*>(cursors);
    emu::launch(jls::place_plane_scans, dim3(frames * rounds, jls::kPlaceShares), dim3(256), 0, slots, slot_pitch, headers, header_size, rounds,
                private_streams, capacity, results, (const jls::FrameCursor*)c);
    emu::launch(jls::advance_plane_cursors, dim3((frames + 63) / 64), dim3(64), 0, slot_pitch, header_size, rounds, results, c, redo, frames);
}

size_t emu_sizeof_scan_result() { return sizeof(jls::ScanResult); }
size_t emu_sizeof_scan_desc() { return sizeof(jls::ScanDesc); }
}
