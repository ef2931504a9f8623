// scan_group_encode.hip -- the scan ENCODER for the modes whose causal template holds RECONSTRUCTED samples (near-lossless
// coding: nothing is known ahead of the serial chain), several scans per wavefront.
//
// encode_scans_serial (scan_serial.hip) codes such a scan with ONE lane of a wavefront and its line window in global
// memory -- about 0.25 MPix/s per scan whatever the batch.  Here, as in scan_group_decode.hip / scan_group_pixels.hip,
// the 64 lanes are split into groups of G lanes and every group codes a scan of its own:
//   * the scan's state lives on chip: 365 context records and the two run contexts in LDS, run index and bit writer in
//     registers (replicated over the group's lanes), TWO lines of pixels in LDS, samples interleaved as in the user's row:
//     the reconstructed previous row, and the current row, which starts as the source samples (as the codec sees them:
//     masked, colour-transformed) and turns into their reconstruction pixel by pixel;
//   * the G lanes of a group share the bulk work: the source row is fetched (src/copy_to_line_buffer.hpp:21-262) by all
//     of them at the start of a line, and the coded bytes go from an LDS staging ring to the destination in cooperative
//     copies;
//   * the pixel is the unit of a step (1 component for planar scans, 2..4 for ILV_SAMPLE): contexts of all components,
//     run mode when all are 0, otherwise the components one after the other in regular mode
//     (src/scan_encoder_impl.hpp:109-302).
// Two step forms, as in scan_group_pixels.hip: the pixel loop for 8-bit samples (NEAR >= 0), written for the number of
// instructions it issues -- contexts of the next pixel worked out while this one is coded, in-register forwarding of a
// context record between the components of a pixel, records / reconstructed samples / code words of a pixel committed
// together -- and the general step (wider samples, escape codes, run mode) with the shared inlines of scan_model.h.
// The bit writer is the reference's (32-bit accumulator, four-byte flushes, a 7-bit byte after every 0xFF,
// src/scan_encoder.hpp:75-186) with its capacity accounting, so destination_too_small is raised for exactly the same
// destination sizes; only where the bytes land differs (the staging ring).  Output is byte-identical to
// scan_encoder::encode_scan.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "scan_group_pixels.hip"
#include "scan_model.h"

namespace jls {
namespace grp {

constexpr uint32_t kOutRingBytes = 2048; // coded bytes staged in LDS per scan
constexpr int kEncodeStepsPerCheck = 16; // pixels between two looks at the staging ring

// LDS of a workgroup (one wavefront): the gradient table shared by its scans (as in scan_group_decode.hip), then one
// region per scan: context records (grp::Record), run contexts, staging ring, two lines of pixels (pixel_line_bytes).
template <typename S>
struct EncodeLayout
{
    static constexpr uint32_t kRecords = 0;                 // 366 x 8 B
    static constexpr uint32_t kRun = 2928;                  // 2 x RunCtx
    static constexpr uint32_t kOut = kRun + 32;             // staging ring
    static constexpr uint32_t kAfterOut = kOut + kOutRingBytes;
};

template <typename S>
__host__ __device__ constexpr uint32_t encode_lines_offset(uint32_t components)
{
    return ((EncodeLayout<S>::kAfterOut + components * (uint32_t)sizeof(S) + 15u) & ~15u) - components * (uint32_t)sizeof(S);
}

// `rows` = 1, or the number of components of a LINE-INTERLEAVED scan (then `components` is 1: every component keeps its
// own pair of lines).
template <typename S>
__host__ __device__ constexpr uint32_t encode_region_bytes(uint32_t width, uint32_t components, uint32_t rows = 1)
{
    return bank_spread(encode_lines_offset<S>(components) + 2 * rows * pixel_line_bytes<S>(width, components));
}

template <typename S>
__host__ __device__ constexpr uint32_t encode_workgroup_lds_bytes(uint32_t width, uint32_t components, uint32_t scans_per_wave,
                                                                  uint32_t rows = 1)
{
    return Layout<S>::kLutBytes + scans_per_wave * encode_region_bytes<S>(width, components, rows);
}

// The reference's bit writer (src/scan_encoder.hpp:75-186) onto a staging ring; every member replicated over the lanes.
struct RingWriter
{
    uint8_t* ring;
    uint64_t remaining; // bytes of the destination not yet used
    uint64_t written;
    uint32_t buf;
    int free_bits;
    bool ff;
    uint32_t err;

    JLS_DEV void flush()
    {
        if (remaining < 4)
        {
            err = kDestinationTooSmall;
            free_bits = free_bits < 0 ? 0 : free_bits;
            return;
        }
        for (int k = 0; k < 4; ++k)
        {
            if (free_bits >= 32)
            {
                free_bits = 32;
                break;
            }
            uint32_t v;
            if (ff)
            {
                v = buf >> 25;
                buf <<= 7;
                free_bits += 7;
            }
            else
            {
                v = buf >> 24;
                buf <<= 8;
                free_bits += 8;
            }
            ring[written & (kOutRingBytes - 1)] = (uint8_t)v;
            ff = v == 0xFFu;
            --remaining;
            ++written;
        }
    }

    JLS_DEV void append(uint32_t bits, int count)
    {
        if (err)
            return;
        free_bits -= count;
        if (free_bits >= 0)
        {
            if (count)
                buf |= bits << free_bits;
            return;
        }
        buf |= bits >> -free_bits;
        flush();
        if (err)
            return;
        if (free_bits < 0)
        {
            buf |= bits >> -free_bits;
            flush();
            if (err)
                return;
        }
        if (free_bits < 32)
            buf |= bits << free_bits;
    }

    JLS_DEV void end_scan()
    {
        if (err)
            return;
        flush();
        if (err)
            return;
        if (ff)
            append(0, (free_bits - 1) % 8);
        flush();
    }

    // Limited-length Golomb code, src/scan_encoder_core.hpp:69-103.
    JLS_DEV void golomb(const Traits& t, int k, int m, int limit)
    {
        int hb = m >> k;
        if (hb < limit - t.qbpp - 1)
        {
            if (hb + 1 > 31)
            {
                append(0, hb / 2);
                hb -= hb / 2;
            }
            const int total = hb + 1 + k;
            const uint32_t rem = (uint32_t)m & ((1u << k) - 1u);
            if (total < 32)
                append((1u << k) | rem, total);
            else
            {
                append(1, hb + 1);
                append(rem, k);
            }
            return;
        }
        if (limit - t.qbpp > 31)
        {
            append(0, 31);
            append(1, limit - t.qbpp - 31);
        }
        else
            append(1, limit - t.qbpp);
        append((uint32_t)(m - 1) & ((1u << t.qbpp) - 1u), t.qbpp);
    }
};

} // namespace grp

// Dynamic LDS: grp::encode_workgroup_lds_bytes<S>(width, NC, 64 / G, NL).  NC = 1: a single-component scan (planar);
// NC = 2..4: a sample-interleaved scan of NC components.  NL = 2..4 (with NC = 1): a LINE-interleaved scan of NL components
// (src/scan_encoder_impl.hpp:109-160): the lines of a pixel row one component after the other, each against the line of
// its own component above it and with its own RUNindex, on the one set of contexts; every component keeps its own pair
// of lines, the user's row is de-interleaved into the NL current lines when its first component starts.
template <typename S, int G, int NC, int NL = 1>
__global__ void __launch_bounds__(64) encode_pixels_group(const ScanDesc* __restrict__ descs, ScanResult* __restrict__ results,
                                                          uint32_t count)
{
    using namespace grp;
    using L = EncodeLayout<S>;
    static_assert(G == 8 || G == 16 || G == 32 || G == 64, "lanes per scan");
    static_assert(NC >= 1 && NC <= 4, "components per pixel");
    static_assert(NL >= 1 && NL <= 4 && (NL == 1 || NC == 1), "lines per pixel row");
    constexpr int kScansPerWave = 64 / G;
    constexpr bool kWide = sizeof(S) > 1;
    JLS_DYNAMIC_LDS(smem);
    const int lane = threadIdx.x;
    const int sid = lane / G;
    const int sub = lane % G;
    const uint32_t scan = blockIdx.x * kScansPerWave + (uint32_t)sid;
    const bool live = scan < count;
    const ScanDesc d = descs[live ? scan : count - 1];
    const Traits t = make_traits(d);
    const uint32_t width = d.width;
    const uint32_t line_bytes = pixel_line_bytes<S>(width, NC);
    const uint32_t line_samples = line_bytes / (uint32_t)sizeof(S);
    const int mask = (1 << d.bits_per_sample) - 1;

    unsigned char* region = smem + Layout<S>::kLutBytes + (size_t)sid * encode_region_bytes<S>(width, NC, NL);
    Record* records = reinterpret_cast<Record*>(region + L::kRecords);
    RunCtx* run_ctx = reinterpret_cast<RunCtx*>(region + L::kRun);
    uint8_t* out_ring = region + L::kOut;
    // sample c of pixel j of a line: [j * NC + c]; pixel 0 is the left edge, 1 .. width the row, width + 1 the right edge
    S* line_a = reinterpret_cast<S*>(region + encode_lines_offset<S>(NC));
    S* line_b = reinterpret_cast<S*>(region + encode_lines_offset<S>(NC) + line_bytes);
    // gradient table (quantised gradient + 4) shared by the scans of the wavefront, for the pixel loop of 8-bit samples; a
    // wavefront whose scans do not share thresholds and NEAR codes all of them with the general step
    unsigned char* lut = smem;
    const ScanDesc& d_first = descs[blockIdx.x * kScansPerWave];
    const Traits t_first = make_traits(d_first);
    const bool own_table = t.t1 == t_first.t1 && t.t2 == t_first.t2 && t.t3 == t_first.t3 && t.bpp == t_first.bpp &&
                           t.near == t_first.near;
    const int cap = kWide ? t_first.t3 : 255; // the table covers -cap .. cap
    {
        const Record fresh{(uint32_t)initial_a(t), 1u};
        for (int q = sub; q < 366; q += G)
            records[q] = fresh;
        if (sub < 2)
            run_ctx[sub] = RunCtx{sub, initial_a(t), 1, 0};
        for (int q = lane; q <= 2 * cap && q < (int)Layout<S>::kLutBytes; q += 64)
            lut[q] = (unsigned char)(quantize(t_first, q - cap) + 4);
        for (uint32_t q = sub; q < 2 * NL * line_samples; q += G)
            line_a[q] = 0;
    }
    RingWriter bw;
    bw.ring = out_ring;
    bw.remaining = d.stream_capacity;
    bw.written = 0;
    bw.buf = 0;
    bw.free_bits = 32;
    bw.ff = false;
    bw.err = kOk;
    uint64_t copied = 0; // bytes of the staging ring already in the destination
    JLS_LOCKSTEP();

    enum : int { kLineStart = 0, kInLine, kFinish, kDone };
    int phase = !live ? kDone : (d.height == 0 ? kFinish : kLineStart);
    uint32_t y = 0, i = 1;
    int run_index = 0;
    S* prev = line_a; // the two lines swap after every row
    S* cur = line_b;
    // line-interleaved scans: the component whose line is being coded, which of its two lines is the current one, and the
    // RUNindex of every component
    int comp = 0, flip = 0;
    int run_index_of[NL];
#pragma unroll
    for (int c = 0; c < NL; ++c)
        run_index_of[c] = 0;
    const bool quick = (!kWide || t_first.t3 <= kMaxTableT3) && __all(!live || own_table) && lds_address(smem) == 0; // (see lds_load)

    // staged bytes -> destination, by the lanes of the group (everything written so far, or whole 256-byte pieces)
    auto drain = [&](bool wanted, bool everything) __attribute__((always_inline)) {
        const uint64_t target = everything ? bw.written : bw.written & ~(uint64_t)255;
        uint64_t at = copied + (uint64_t)sub;
        while (__any(wanted && at < target))
        {
            if (wanted && at < target)
                d.stream[at] = out_ring[at & (kOutRingBytes - 1)];
            at += G;
        }
        if (wanted)
            copied = target > copied ? target : copied;
    };

    // One pixel in the general form for the lanes in `todo` (src/scan_encoder_impl.hpp:109-302): regular mode component
    // after component on the one set of contexts, or run mode with its interruption sample(s).
    auto general_pixel = [&](bool todo) __attribute__((always_inline)) {
        const uint32_t at0 = todo ? i : 1u;
        int ra[NC], rb[NC], rc[NC], qs[NC];
        bool all_zero = true;
#pragma unroll
        for (int c = 0; c < NC; ++c)
        {
            ra[c] = (int)cur[(at0 - 1) * NC + c];
            rc[c] = (int)prev[(at0 - 1) * NC + c];
            rb[c] = (int)prev[at0 * NC + c];
            const int rd = (int)prev[(at0 + 1) * NC + c];
            qs[c] = 81 * quantize(t, rd - rb[c]) + 9 * quantize(t, rb[c] - rc[c]) + quantize(t, rc[c] - ra[c]);
            all_zero = all_zero && qs[c] == 0;
        }
        const bool regular = todo && !all_zero;
        const bool in_run = todo && all_zero;
        // -- regular mode, src/scan_encoder_core.hpp:40-67
        int rx[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c)
        {
            const int s = qs[c] >> 31;
            const int idx = (qs[c] ^ s) - s;
            const Record rec = records[idx];
            RegCtx ctx{(int)rec.a, (int)rec.ncb >> 16, (int)(signed char)(rec.ncb >> 8), (int)(rec.ncb & 0xFFu)};
            const int x = (int)cur[at0 * NC + c];
            const int k = regular_k(ctx);
            const int px = clamp_sample(t, med_predict(ra[c], rb[c], rc[c]) + ((ctx.c ^ s) - s));
            const int e = error_value(t, ((x - px) ^ s) - s);
            rx[c] = reconstruct(t, px, (e ^ s) - s);
            if (regular)
            {
                if (k >= 16)
                    bw.err = kInvalidData;
                else
                {
                    bw.golomb(t, k, map_error(error_correction(ctx, k | t.near) ^ e), t.limit);
                    if (!regular_update(ctx, e, t.near, t.reset))
                        bw.err = kInvalidData;
                }
            }
            JLS_LOCKSTEP();
            if (regular)
                records[idx] = Record{(uint32_t)ctx.a, (uint32_t)ctx.n | (((uint32_t)ctx.c & 0xFFu) << 8) | ((uint32_t)ctx.b << 16)};
            JLS_LOCKSTEP();
        }
        if (regular)
        {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                cur[i * NC + c] = (S)rx[c];
            ++i;
        }
        // -- run mode, src/scan_encoder_impl.hpp:249-302, src/scan_encoder.hpp:53-73
        if (__any(in_run))
        {
            const uint32_t remaining = width - (i - 1);
            uint32_t run = 0;
            bool interrupted = false;
            if (in_run)
            {
                for (;;)
                {
                    bool near_all = true;
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        near_all = near_all && is_near(t, (int)cur[(i + run) * NC + c], ra[c]);
                    if (!near_all)
                        break;
                    if (++run == remaining)
                        break;
                }
            }
            JLS_LOCKSTEP();
            { // the run's pixels are reconstructed as Ra, by the lanes of the group
                uint32_t r = (uint32_t)sub;
                while (__any(in_run && r < run))
                {
                    if (in_run && r < run)
                    {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            cur[(i + r) * NC + c] = (S)ra[c];
                    }
                    r += G;
                }
            }
            if (in_run)
            {
                uint32_t left = run;
                while (left >= (1u << run_j(run_index)))
                {
                    bw.append(1, 1);
                    left -= 1u << run_j(run_index);
                    if (run_index < 31)
                        ++run_index;
                }
                if (run == remaining)
                {
                    if (left != 0)
                        bw.append(1, 1);
                    i = width + 1;
                }
                else
                {
                    bw.append(left, run_j(run_index) + 1);
                    interrupted = true;
                }
            }
            const uint32_t at = i + run;
#pragma unroll
            for (int c = 0; c < NC; ++c)
            { // run interruption sample(s): src/scan_encoder_core.hpp:105-138
                const int rb_at = (int)prev[(interrupted ? at : 1u) * NC + c];
                const int x = (int)cur[(interrupted ? at : 1u) * NC + c];
                const int which = (NC == 1 && is_near(t, ra[c], rb_at)) ? 1 : 0;
                const int sg = which ? 1 : ((rb_at - ra[c]) < 0 ? -1 : 1);
                const int e = which ? error_value(t, x - ra[c]) : error_value(t, (x - rb_at) * sg);
                RunCtx ctx = run_ctx[which];
                if (interrupted)
                {
                    const int k = run_k(ctx);
                    const int map = run_map(ctx, e, k);
                    const int em = 2 * (e < 0 ? -e : e) - ctx.ritype - map;
                    bw.golomb(t, k, em, t.limit - run_j(run_index) - 1);
                    run_update(ctx, e, em, t.reset);
                }
                const int rec = which ? reconstruct(t, ra[c], e) : reconstruct(t, rb_at, e * sg);
                JLS_LOCKSTEP();
                if (interrupted)
                {
                    run_ctx[which] = ctx;
                    cur[at * NC + c] = (S)rec;
                }
                JLS_LOCKSTEP();
            }
            if (interrupted)
            {
                if (run_index > 0)
                    --run_index;
                i = at + 1;
            }
        }
        JLS_LOCKSTEP();
    };

    for (;;)
    {
        // ---- staging ring: drained before a burst of pixels could overrun it (a pixel writes at most NC * 8 + 8 bytes)
        {
            const bool pending = phase != kDone && bw.written - copied >= kOutRingBytes / 2;
            if (__any(pending))
            {
                JLS_LOCKSTEP();
                drain(pending, false);
                JLS_LOCKSTEP();
            }
        }
        // ---- a new line: fetch the row (src/copy_to_line_buffer.hpp) into the current line, edge samples
        // (src/scan_codec.hpp:189-195): cur[0] = prev[1]
        {
            const bool starting = phase == kLineStart;
            if (__any(starting))
            {
                const uint8_t* row = d.pixels + (size_t)y * d.pixel_stride;
                if (NL > 1)
                {
                    if (starting)
                    {
                        prev = line_a + (uint32_t)(2 * comp + flip) * line_samples;
                        cur = line_a + (uint32_t)(2 * comp + (flip ^ 1)) * line_samples;
#pragma unroll
                        for (int c = 0; c < NL; ++c)
                            run_index = comp == c ? run_index_of[c] : run_index;
                    }
                    // the first component of a pixel row: the user's row goes, de-interleaved, into the NL current lines
                    const bool fetching = starting && comp == 0;
                    const bool transformed = NL == 3 && d.color_transformation != 0;
                    uint32_t xx = (uint32_t)sub;
                    while (__any(fetching && xx < width))
                    {
                        if (fetching && xx < width)
                        {
                            unsigned v[4] = {0, 0, 0, 0};
#pragma unroll
                            for (int c = 0; c < NL; ++c)
                            {
                                const uint8_t* q = row + ((size_t)xx * NL + c) * sizeof(S);
                                v[c] = kWide ? (unsigned)q[0] | ((unsigned)q[1] << 8) : (unsigned)q[0];
                            }
                            if (transformed)
                                hp_forward(d.color_transformation, kWide, (int)v[0], (int)v[1], (int)v[2], v);
#pragma unroll
                            for (int c = 0; c < NL; ++c)
                                line_a[(uint32_t)(2 * c + (flip ^ 1)) * line_samples + 1 + xx] = (S)(transformed ? v[c] : v[c] & (unsigned)mask);
                        }
                        xx += G;
                    }
                }
                else
                {
                    S* samples = cur + NC; // pixel 1
                    const bool transformed = NC == 3 && d.color_transformation != 0;
                    const bool plain = !transformed && mask == (kWide ? 0xFFFF : 0xFF);
                    const uint32_t row_bytes = width * NC * (uint32_t)sizeof(S);
                    const bool aligned = ((reinterpret_cast<uintptr_t>(d.pixels) | d.pixel_stride) & 15u) == 0;
                    const uint32_t wide_bytes = aligned && plain ? row_bytes & ~15u : 0u;
                    uint32_t off = (uint32_t)sub * 16u;
                    while (__any(starting && off < wide_bytes))
                    {
                        if (starting && off < wide_bytes)
                            *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(samples) + off) =
                                *reinterpret_cast<const uint4*>(row + off);
                        off += G * 16u;
                    }
                    // the samples behind the 16-byte pieces, or the whole row when it needs masking
                    uint32_t ss = wide_bytes / (uint32_t)sizeof(S) + (uint32_t)sub;
                    while (__any(starting && !transformed && ss < width * NC))
                    {
                        if (starting && !transformed && ss < width * NC)
                        {
                            const uint8_t* q = row + (size_t)ss * sizeof(S);
                            const unsigned v = kWide ? (unsigned)q[0] | ((unsigned)q[1] << 8) : (unsigned)q[0];
                            samples[ss] = (S)(v & (unsigned)mask);
                        }
                        ss += G;
                    }
                    // ... or the colour transform
                    uint32_t xx = (uint32_t)sub;
                    while (__any(starting && transformed && xx < width))
                    {
                        if (starting && transformed && xx < width)
                        {
                            unsigned v[3];
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                            {
                                const uint8_t* q = row + ((size_t)xx * NC + (c < NC ? c : 0)) * sizeof(S);
                                v[c] = kWide ? (unsigned)q[0] | ((unsigned)q[1] << 8) : (unsigned)q[0];
                            }
                            hp_forward(d.color_transformation, kWide, (int)v[0], (int)v[1], (int)v[2], v);
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                samples[xx * NC + (c < NC ? c : 0)] = (S)v[c];
                        }
                        xx += G;
                    }
                }
                JLS_LOCKSTEP();
                if (starting && sub < NC)
                    cur[sub] = prev[NC + sub];
                JLS_LOCKSTEP();
                if (starting)
                {
                    i = 1;
                    phase = kInLine;
                }
            }
        }
        // ---- pixels
        bool stepped = false; // the pixel loop ran: the lanes it stopped at take ONE general step
        {
            const bool active = quick && phase == kInLine && bw.err == kOk; // i <= width: the end of a line is handled at once
            const LaneMask active_m = lanes_where(active);
            auto pixel_loop = [&](auto near_tag) __attribute__((always_inline)) {
                constexpr bool kNearLoop = decltype(near_tag)::value;
                const int near = t.near, step_size = 2 * t.near + 1, range = t.range, range_span = t.range * (2 * t.near + 1);
                const int half_range = (t.range + 1) / 2;
                const uint32_t reciprocal = 0xFFFFFFFFu / (uint32_t)(kNearLoop ? step_size : 3) + 1u; // floor(2^32 / step) + 1
                stepped = true;
                const uint32_t rest_of_line = width + 1 - i;
                uint32_t steps = kEncodeStepsPerCheck;
                while (lanes_where(active && rest_of_line < steps) != 0)
                    --steps;
                uint32_t ticker = 1u << (steps - 1);
                const uint32_t limit_v = opaque(active ? (uint32_t)(t.limit - t.qbpp - 1) : 0u);
                const int maxval = t.maxval, reset = t.reset, bpp = t.bpp;
                const uint32_t records_address = opaque(lds_address(records));
                // scans outside a line stay on pixel 1 of their (dead) lines and store nothing
                const S* pp = prev + (active ? i : 1u) * NC; // pixel i of the previous line
                S* cp = cur + (active ? i : 1u) * NC;        // pixel i of the current line: still the source samples
                const uint32_t advance = active ? (uint32_t)NC : 0u;
                int qsu[NC], q1[NC], ra[NC], rb[NC], rc[NC], xs[NC];
                Record rec[NC];
                auto index_of = [&](int q) -> uint32_t { return abs_difference((uint32_t)q, 364u); };
                // the table is at LDS address 0: -255 .. 255 for 8-bit samples, -T3 .. T3 (differences clamped) for wider ones
                auto gradient = [&](int diff) -> int {
                    return (int)lds_load<unsigned char>((uint32_t)((kWide ? med3(diff, -cap, cap) : diff) + cap));
                };
                auto record_at = [&](uint32_t idx) -> Record {
                    const uint32_t at = records_address + (idx << 3);
                    return Record{lds_load<uint32_t>(at), lds_load<uint32_t>(at + 4)};
                };
#pragma unroll
                for (int c = 0; c < NC; ++c)
                {
                    ra[c] = (int)cp[c - NC];
                    rc[c] = (int)pp[c - NC];
                    rb[c] = (int)pp[c];
                    xs[c] = (int)cp[c];
                    const int rd = (int)pp[c + NC];
                    q1[c] = gradient(rd - rb[c]);
                    qsu[c] = mad24(mad24(q1[c], 9, gradient(rb[c] - rc[c])), 9, gradient(rc[c] - ra[c]));
                    rec[c] = record_at(index_of(qsu[c]));
                }
                uint32_t k_seen = 0;
                LaneMask ok_m;
                do
                {
                    // the next pixel's Rb, Rd and source samples: reads that depend on nothing of this pixel go first
                    const S* const pn = pp + advance;
                    int rb_next[NC], rd_next[NC], xs_next[NC], q1_next[NC], q3_next[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                    {
                        rb_next[c] = (int)pn[c];
                        rd_next[c] = (int)pn[c + NC];
                        xs_next[c] = (int)cp[c + advance];
                    }
                    // a pixel in run mode: every component has context 0
                    LaneMask run_m = lanes_where(qsu[0] == 364);
#pragma unroll
                    for (int c = 1; c < NC; ++c)
                        run_m &= lanes_where(qsu[c] == 364);
                    ok_m = ~run_m;
                    uint32_t k_pixel = 0;
                    int rx[NC];
                    uint32_t idx[NC], code[NC], length[NC];
                    Record updated[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                    {
                        idx[c] = index_of(qsu[c]);
                        // the record as the earlier components of this pixel left it
                        Record r = rec[c];
#pragma unroll
                        for (int e = 0; e < c; ++e)
                        {
                            const bool same = idx[c] == idx[e];
                            r.a = same ? updated[e].a : r.a;
                            r.ncb = same ? updated[e].ncb : r.ncb;
                        }
                        const int n = (int)(r.ncb & 0xFFu);
                        const int cc = (int)(signed char)(r.ncb >> 8);
                        const int bb = (int)r.ncb >> 16;
                        // k = min{k : N << k >= A} from the exponents of A and N (see scan_group_decode.hip)
                        const int k_raw = ((int)(float_bits(r.a) - float_bits((uint32_t)n)) + 0x7FFFFF) >> 23;
                        const int k = k_raw < 0 ? 0 : k_raw;
                        const int sgn = qsu[c] < 364 ? -1 : 1;
                        const int px0 = med3(ra[c] + (rb[c] - rc[c]), ra[c], rb[c]);
                        const int px = med3(mad24(cc, sgn, px0), 0, maxval);
                        // Errval: quantised, reduced modulo RANGE (src/default_traits.hpp:77-80,123-139,157-163); Rx
                        const int diff = (xs[c] - px) * sgn;
                        int e;
                        if (kNearLoop)
                        {
                            // (|diff| + NEAR) / (2 NEAR + 1) by the reciprocal: exact for numerators below 2^16
                            const int q = (int)__umulhi(abs_difference((uint32_t)xs[c], (uint32_t)px) + (uint32_t)near, reciprocal);
                            const int neg = diff >> 31;
                            e = (q ^ neg) - neg;
                            e += (e >> 31) & range;
                            e -= e >= half_range ? range : 0;
                            int v = mad24(e * sgn, step_size, px);
                            v += v < -near ? range_span : (v > maxval + near ? -range_span : 0);
                            rx[c] = med3(v, 0, maxval);
                        }
                        else
                        {
                            e = sign_extend(diff, bpp); // RANGE = 2^bpp
                            rx[c] = xs[c];
                        }
                        q3_next[c] = gradient(rb[c] - rx[c]); // the next pixel's Rc - Ra
                        if (c == 0)
                        {
#pragma unroll
                            for (int f = 0; f < NC; ++f)
                                q1_next[f] = gradient(rd_next[f] - rb_next[f]);
                        }
                        // mapped error (src/jpegls_algorithm.hpp:67-73; complemented first when k = 0, NEAR = 0 and
                        // 2B + N - 1 < 0, src/regular_mode_context.hpp:36-42), limited-length Golomb code
                        // (src/scan_encoder_core.hpp:69-103): hb zeros, a one, the k low bits
                        const int corrected = kNearLoop ? e : e ^ (((k - 1) & (2 * bb + n - 1)) >> 31);
                        const int mm = (corrected >> 31) ^ (2 * corrected);
                        const uint32_t hb = (uint32_t)mm >> k;
                        ok_m &= lanes_where(hb < limit_v);
                        code[c] = (uint32_t)mm + ((1u - hb) << k); // (1 << k) | (mm & ((1 << k) - 1))
                        length[c] = hb + 1u + (uint32_t)k;
                        ok_m &= lanes_where(length[c] < 32u);
                        k_pixel |= (uint32_t)k;
                        // A.12 / A.13, src/regular_mode_context.hpp:45-93, in the median form of scan_group_decode.hip
                        const int magnitude = e < 0 ? -e : e;
                        int u_a = (int)r.a + magnitude;
                        int u_n1 = n + 1;
                        int u_tb = kNearLoop ? mad24(e, step_size, bb) : bb + e;
                        const LaneMask halve_m = lanes_where(n == reset);
                        if (__builtin_expect(halve_m != 0, 0))
                        {
                            JLS_RARE_BLOCK();
                            if (lane_of(halve_m))
                            {
                                u_a >>= 1;
                                u_n1 = (n >> 1) + 1;
                                u_tb >>= 1;
                            }
                        }
                        const int minus_delta = 1 - med3(u_tb, 0, 1) - med3(u_tb + u_n1, 0, 1);
                        const int b_new = med3(mad24(minus_delta, u_n1, u_tb), 1 - u_n1, 0);
                        const int c_new = med3(cc - minus_delta, -128, 127);
                        updated[c] = Record{(uint32_t)u_a, ((uint32_t)b_new << 16) | pack_bytes((uint32_t)c_new, (uint32_t)u_n1)};
                    }
                    // the next pixel's contexts (Rc = this pixel's Rb, Ra = this pixel's reconstruction)
                    uint32_t at_next[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                    {
                        qsu[c] = mad24(mad24(q1_next[c], 9, q1[c]), 9, q3_next[c]);
                        q1[c] = q1_next[c];
                        rc[c] = rb[c];
                        rb[c] = rb_next[c];
                        ra[c] = rx[c];
                        xs[c] = xs_next[c];
                        at_next[c] = records_address + (index_of(qsu[c]) << 3);
                    }
                    // the pixel is complete: records, reconstructed samples and code words go out together (a lane that could
                    // not code one of its components stores nothing and keeps its position; the loop ends for everybody)
                    JLS_LOCKSTEP();
                    if (lane_of(ok_m))
                    {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                        {
                            const uint32_t at = records_address + (idx[c] << 3);
                            lds_store<uint32_t>(at, updated[c].a); // in component order: the last of equal contexts stays
                            lds_store<uint32_t>(at + 4, updated[c].ncb);
                            cp[c] = (S)rx[c];
                        }
                        ++i;
                        k_seen |= k_pixel;
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                    { // RingWriter::append (src/scan_encoder.hpp:85-116) with its common case inline
                        const bool mine = lane_of(ok_m);
                        const int free_after = bw.free_bits - (int)length[c];
                        const LaneMask over_m = lanes_where(mine && free_after < 0);
                        if (mine && free_after >= 0)
                        {
                            bw.free_bits = free_after;
                            bw.buf |= code[c] << free_after;
                        }
                        if (__builtin_expect(over_m != 0, 0))
                        {
                            JLS_RARE_BLOCK();
                            if (lane_of(over_m))
                                bw.append(code[c], (int)length[c]);
                        }
                    }
                    JLS_LOCKSTEP_STORES();
                    pp = pn;
                    cp += advance;
                    // the records of the next pixel: read behind the stores
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        rec[c] = Record{lds_load<uint32_t>(at_next[c]), lds_load<uint32_t>(at_next[c] + 4)};
                    // a writer in trouble (destination too small) ends the loop as well
                    ok_m &= ~lanes_where(bw.err != kOk);
                    ticker = tick(ticker, active_m, ok_m);
                } while (ticker != 0);
                if (active && k_seen >= 16u)
                    bw.err = kInvalidData; // src/regular_mode_context.hpp:99-111 (cannot happen with 8-bit samples)
                // one general step for the lanes the loop stopped at
                const bool stopped = active && !lane_of(ok_m) && bw.err == kOk;
                general_pixel(stopped);
            };
            if (active_m != 0)
            {
                if (t_first.near == 0)
                    pixel_loop(std::false_type{});
                else
                    pixel_loop(std::true_type{});
            }
        }
        if (!stepped && __any(phase == kInLine))
        {
            for (int step = 0; step < kEncodeStepsPerCheck; ++step)
            {
                const bool todo = phase == kInLine && i <= width && bw.err == kOk;
                general_pixel(todo);
                if (__any(bw.err != kOk || (phase == kInLine && i > width) ||
                          (phase != kDone && bw.written - copied >= kOutRingBytes / 2)))
                    break;
            }
        }
        if (bw.err != kOk && phase != kDone)
            phase = kDone;
        // ---- end of a line: the right edge of the next line's previous line; the lines swap
        {
            const bool ending = phase == kInLine && i > width;
            if (__any(ending))
            {
                if (ending && sub < NC)
                    cur[(width + 1) * NC + sub] = cur[width * NC + sub];
                JLS_LOCKSTEP();
                if (ending && NL == 1)
                {
                    S* const was_prev = prev;
                    prev = cur;
                    cur = was_prev;
                    ++y;
                    phase = y == d.height ? kFinish : kLineStart;
                }
                if (ending && NL > 1)
                { // the next component of this pixel row, or the first of the next one (lines are chosen at the line start)
#pragma unroll
                    for (int c = 0; c < NL; ++c)
                        run_index_of[c] = comp == c ? run_index : run_index_of[c];
                    if (comp == NL - 1)
                    {
                        comp = 0;
                        flip ^= 1;
                        ++y;
                    }
                    else
                        ++comp;
                    phase = y == d.height ? kFinish : kLineStart;
                }
            }
        }
        // ---- end of the scan: src/scan_encoder.hpp:167-186
        {
            const bool finishing = phase == kFinish;
            if (__any(finishing))
            {
                if (finishing)
                    bw.end_scan();
                JLS_LOCKSTEP();
                drain(finishing && bw.err == kOk, true);
                JLS_LOCKSTEP();
                if (finishing)
                    phase = kDone;
            }
        }
        if (__all(phase == kDone))
            break;
    }
    if (live && sub == 0)
        results[scan] = ScanResult{bw.err, 0, bw.err == kOk ? bw.written : 0};
}

} // namespace jls
