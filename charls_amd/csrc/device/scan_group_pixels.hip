// scan_group_pixels.hip -- speed path of the scan decoder for SAMPLE-INTERLEAVED scans (ILV_SAMPLE: 2..4 components coded
// pixel by pixel, reference src/scan_decoder_impl.hpp:162-261), lossless and near-lossless, several scans per wavefront.
//
// Same organisation as scan_group_decode.hip -- the 64 lanes are split into groups of G lanes, every group decodes a scan
// of its own with all its state replicated over the group's lanes, control flow convergent for the wavefront, the lanes
// of a group sharing the bulk work (un-stuffing, run fills, colour transform and row stores) -- with the pixel as the
// unit of a step:
//   * the context of every component of the pixel comes first (two table look-ups per component: the gradient towards
//     the next sample of the previous line is the next pixel's gradient towards the previous one);
//   * the pixel is in run mode only when ALL its components have context 0 (src/scan_decoder_impl.hpp:196-204);
//     otherwise its components are decoded one after the other in regular mode on the ONE set of contexts, a component
//     with context 0 on context record 0 with positive sign;
//   * the components of a run-interruption pixel are decoded against run context 0 with the sign of Rb - Ra
//     (src/scan_decoder_impl.hpp:300-337, src/scan_decoder_core.hpp:72-100).
// The arithmetic is the general one (NEAR >= 0, RANGE not a power of two: src/default_traits.hpp), written with the
// shared inlines of scan_model.h rather than for instruction count: what this kernel buys is scans per wavefront and
// state in LDS / registers.  The exact one-scan-per-wavefront decoder (scan_wave_decode.hip) decodes such a scan at about
// 0.5 MPix/s whatever the batch.
//
// LDS per scan: 365 context records, two run contexts, the dense bit ring and one line PER COMPONENT (planar, the
// interleaving and the inverse colour transform happen when a finished line goes to the user's row).  As in
// scan_group_decode.hip a result is accepted only when the scan ends cleanly; everything else reports kFastRetry.
#pragma once
#include <hip/hip_runtime.h>

#include "scan_group_decode.hip"
#include "scan_model.h"

namespace jls {
namespace grp {

constexpr int kPixelStepsPerCheck = 8; // pixels between two looks at the producer

template <typename S>
__host__ __device__ constexpr uint32_t pixel_line_samples(uint32_t width)
{
    return (width + 6 + 15) & ~15u; // per component; multiples of 16 samples keep every component's sample 1 aligned alike
}

template <typename S>
__host__ __device__ constexpr uint32_t pixel_region_bytes(uint32_t width, uint32_t components)
{
    return (Layout<S>::kLine + components * pixel_line_samples<S>(width) * (uint32_t)sizeof(S) + 15u) & ~15u;
}

template <typename S>
__host__ __device__ constexpr uint32_t pixel_workgroup_lds_bytes(uint32_t width, uint32_t components, uint32_t scans_per_wave)
{
    return Layout<S>::kLutBytes + scans_per_wave * pixel_region_bytes<S>(width, components);
}

} // namespace grp

// Dynamic LDS: grp::pixel_workgroup_lds_bytes<S>(width, NC, 64 / G).
template <typename S, int G, int NC>
__global__ void __launch_bounds__(64) decode_pixels_group(const ScanDesc* __restrict__ descs, ScanResult* __restrict__ results,
                                                          uint32_t count)
{
    using namespace grp;
    using L = Layout<S>;
    static_assert(G == 8 || G == 16 || G == 32, "lanes per scan");
    static_assert(NC >= 2 && NC <= 4, "components per pixel");
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
    const uint32_t line_samples = pixel_line_samples<S>(width);

    unsigned char* region = smem + L::kLutBytes + (size_t)sid * pixel_region_bytes<S>(width, NC);
    Record* records = reinterpret_cast<Record*>(region + L::kRecords);
    RunCtx* run_ctx = reinterpret_cast<RunCtx*>(region + L::kRun);
    uint32_t* ring = reinterpret_cast<uint32_t*>(region + L::kRing);
    S* lines = reinterpret_cast<S*>(region + L::kLine); // component c: lines + c * line_samples, sample i at [i]
    // gradient table shared by the scans of the wavefront (see scan_group_decode.hip); NEAR is part of it
    unsigned char* lut = smem;
    const ScanDesc& d_first = descs[blockIdx.x * kScansPerWave];
    const Traits t_first = make_traits(d_first);
    const int cap = kWide ? t_first.t3 : 255;
    const bool own_table = t.t1 == t_first.t1 && t.t2 == t_first.t2 && t.t3 == t_first.t3 && t.bpp == t_first.bpp &&
                           t.near == t_first.near;
    {
        const Record fresh{(uint32_t)initial_a(t), 1u};
        for (int q = sub; q < 366; q += G)
            records[q] = fresh;
        if (sub < 2)
            run_ctx[sub] = RunCtx{sub, initial_a(t), 1, 0};
        for (int q = lane; q <= 2 * cap; q += 64)
            lut[q] = (unsigned char)(quantize(t_first, q - cap) + 4);
        for (uint32_t q = sub; q < NC * line_samples; q += G)
            lines[q] = 0;
        for (uint32_t q = sub; q <= kRingWords; q += G)
            ring[q] = 0;
    }
    Producer src;
    {
        const uint64_t mis = (uint64_t)(reinterpret_cast<uintptr_t>(d.stream) & 15u);
        src.gbase = d.stream - mis;
        src.u_begin = mis;
        src.u_next = 0;
        src.u_end = mis + d.stream_capacity;
        src.u_marker = ~0ull;
        src.produced = 0;
        src.prev_byte = 0;
        src.ended = d.stream_capacity == 0;
    }
    JLS_LOCKSTEP();

    enum : int { kLineStart = 0, kInLine, kDrain, kDone };
    int phase = !live || !own_table ? kDone : (d.height == 0 ? kDrain : kLineStart);
    bool retry = live && !own_table;
    uint32_t p = 0; // consumed dense bits
    uint32_t y = 0, i = 1;
    int run_index = 0;
    int a[NC], rc[NC], corner[NC], first[NC]; // Ra, Rc = prev[i - 1], prev[0] of this line, cur[0] of this line, per component
    int q_prev[NC];                           // quantised prev[i] - prev[i - 1] per component (the last pixel's Q1)
#pragma unroll
    for (int c = 0; c < NC; ++c)
        a[c] = rc[c] = corner[c] = first[c] = q_prev[c] = 0;
    const uint32_t margin_bits = (uint32_t)(kPixelStepsPerCheck + 1) * NC * (uint32_t)t.limit + 320u;

    auto quantised = [&](int diff) -> int { // quantised gradient, -4 .. 4
        if (kWide)
            diff = diff < -cap ? -cap : (diff > cap ? cap : diff);
        return (int)lut[diff + cap] - 4;
    };
    auto line_of = [&](int c) -> S* { return lines + (uint32_t)c * line_samples; };

    // Errval of one regular-mode sample of context q (sign s, index idx); false = leave the scan to the exact decoder
    auto decode_regular = [&](int idx, int& errval, RegCtx& ctx, int& c_before) -> bool {
        const Record rec = records[idx];
        ctx = RegCtx{(int)rec.a, (int)rec.ncb >> 16, (int)(signed char)(rec.ncb >> 8), (int)(rec.ncb & 0xFFu)};
        c_before = ctx.c; // the prediction is corrected with C as it was BEFORE this sample's update
        const int k = regular_k(ctx);
        if (k >= 16)
            return false;
        const int u = take_unary(ring, p, 47); // anything longer: let the exact decoder classify it
        if (u < 0)
            return false;
        int mm;
        if (u < t.limit - t.qbpp - 1)
            mm = (u << k) | (int)take_bits(ring, p, k);
        else
            mm = (int)take_bits(ring, p, t.qbpp) + 1;
        int e = unmap_error(mm);
        if ((e < 0 ? -e : e) > 65535)
            return false; // src/scan_decoder_core.hpp:38-69
        if (k == 0)
            e ^= error_correction(ctx, t.near);
        if (!regular_update(ctx, e, t.near, t.reset))
            return false;
        errval = e;
        return true;
    };

    for (;;)
    {
        // ---- producer
        {
            const uint32_t ahead = src.produced - p;
            const bool busy = phase != kDone && !src.ended;
            const bool need = busy && (phase == kDrain ? ahead < 64u : ahead < margin_bits);
            if (__any(need))
            {
                const bool want = busy && ahead <= kRingBits - G * 128u - 128u;
                refill<G>(src, ring, want, lane, sub);
                continue;
            }
        }
        // ---- first pixel of a line (src/scan_codec.hpp:189-195 per component)
        {
            const bool starting = phase == kLineStart;
            if (__any(starting))
            {
                if (starting && sub < NC)
                    line_of(sub)[width + 1] = line_of(sub)[width];
                JLS_LOCKSTEP();
                if (starting)
                {
                    i = 1;
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                    {
                        const int rb = (int)line_of(c)[1];
                        rc[c] = corner[c];            // prev[0]
                        a[c] = rb;                    // cur[0] = prev[1]
                        first[c] = rb;
                        q_prev[c] = quantised(rb - rc[c]);
                    }
                    phase = kInLine;
                }
            }
        }
        // ---- pixels
        bool in_run = false;
        for (int step = 0; step < kPixelStepsPerCheck; ++step)
        {
            const bool active = phase == kInLine && i <= width && !retry;
            int rb[NC], qs[NC], q1[NC];
            bool all_zero = true;
#pragma unroll
            for (int c = 0; c < NC; ++c)
            {
                rb[c] = (int)line_of(c)[active ? i : 0];
                const int rd = (int)line_of(c)[active ? i + 1 : 0];
                q1[c] = quantised(rd - rb[c]);
                qs[c] = 81 * q1[c] + 9 * q_prev[c] + quantised(rc[c] - a[c]);
                all_zero = all_zero && qs[c] == 0;
            }
            in_run = active && all_zero;
            const bool regular = active && !all_zero;
            int x[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c)
            {
                const int s = qs[c] >> 31;
                const int idx = (qs[c] ^ s) - s;
                int e = 0, c_before = 0;
                RegCtx ctx{0, 0, 0, 1};
                bool good = regular && !retry;
                if (good)
                    good = decode_regular(idx, e, ctx, c_before);
                const int px = clamp_sample(t, med_predict(a[c], rb[c], rc[c]) + ((c_before ^ s) - s));
                x[c] = reconstruct(t, px, (e ^ s) - s);
                JLS_LOCKSTEP();
                if (good)
                    records[idx] = Record{(uint32_t)ctx.a, (uint32_t)ctx.n | (((uint32_t)ctx.c & 0xFFu) << 8) | ((uint32_t)ctx.b << 16)};
                else if (regular)
                    retry = true;
                JLS_LOCKSTEP();
            }
            if (regular && !retry)
            {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                {
                    line_of(c)[i] = (S)x[c];
                    a[c] = x[c];
                    rc[c] = rb[c];
                    q_prev[c] = q1[c];
                }
                ++i;
            }
            JLS_LOCKSTEP();
            if (__any(in_run || retry || (phase == kInLine && i > width)))
                break;
        }

        // ---- run mode of a pixel: src/scan_decoder_impl.hpp:264-337
        if (__any(in_run))
        {
            const uint32_t remaining = width - (i - 1);
            uint32_t run = 0;
            bool counting = in_run;
            while (__any(counting))
            {
                const uint32_t bit = peek32(ring, p) & 1u;
                if (counting)
                {
                    ++p;
                    if (bit)
                    {
                        const uint32_t block = 1u << run_j(run_index);
                        const uint32_t count_now = block < remaining - run ? block : remaining - run;
                        run += count_now;
                        if (count_now == block && run_index < 31)
                            ++run_index;
                        if (run == remaining)
                            counting = false;
                    }
                    else
                        counting = false;
                }
            }
            bool interrupted = in_run && run != remaining;
            if (interrupted)
            {
                run += take_bits(ring, p, run_j(run_index));
                if (run > remaining)
                {
                    retry = true;
                    interrupted = false;
                    run = 0;
                }
            }
            JLS_LOCKSTEP();
            {
                uint32_t r = (uint32_t)sub;
                while (__any(in_run && r < run))
                {
                    if (in_run && r < run)
                    {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            line_of(c)[i + r] = (S)a[c];
                    }
                    r += G;
                }
            }
            const uint32_t at = i + run;
            JLS_LOCKSTEP();
            int x[NC], rb_at[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c)
            { // every component against run context 0, in component order (src/scan_decoder_impl.hpp:300-337)
                rb_at[c] = (int)line_of(c)[interrupted ? at : 0]; // prev[at]: not overwritten yet
                RunCtx ctx = run_ctx[0];
                x[c] = 0;
                if (interrupted)
                {
                    const int k = run_k(ctx);
                    const int limit = t.limit - run_j(run_index) - 1;
                    const int u = k > 24 ? -1 : take_unary(ring, p, 47);
                    if (u < 0)
                    {
                        retry = true;
                        interrupted = false;
                    }
                    else
                    {
                        int em;
                        if (u < limit - t.qbpp - 1)
                            em = (u << k) + (int)take_bits(ring, p, k);
                        else
                            em = (int)take_bits(ring, p, t.qbpp) + 1;
                        const int e = run_error_value(ctx, em + ctx.ritype, k);
                        run_update(ctx, e, em, t.reset);
                        x[c] = reconstruct(t, rb_at[c], e * ((rb_at[c] - a[c]) < 0 ? -1 : 1));
                    }
                }
                JLS_LOCKSTEP();
                if (interrupted)
                    run_ctx[0] = ctx;
                JLS_LOCKSTEP();
            }
            if (interrupted)
            {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                {
                    line_of(c)[at] = (S)x[c];
                    a[c] = x[c];
                    rc[c] = rb_at[c];
                }
                if (run_index > 0)
                    --run_index;
                i = at + 1;
            }
            else if (in_run && !retry)
                i = width + 1; // the run reached the end of the line
            JLS_LOCKSTEP();
            if (interrupted && i <= width)
            {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    q_prev[c] = quantised((int)line_of(c)[i] - rc[c]);
            }
        }

        if (retry)
            phase = kDone;

        // ---- finished line -> user's row: interleave, inverse colour transform (src/copy_from_line_buffer.hpp:19-191)
        {
            const bool ending = phase == kInLine && i > width;
            if (__any(ending))
            {
                uint8_t* row = d.pixels + (size_t)y * d.pixel_stride;
                uint32_t xx = (uint32_t)sub;
                while (__any(ending && xx < width))
                {
                    if (ending && xx < width)
                    {
                        unsigned v[4];
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            v[c] = line_of(c)[1 + xx];
                        if (NC == 3 && d.color_transformation != 0)
                            hp_inverse(d.color_transformation, kWide, (int)v[0], (int)v[1], (int)v[2], v);
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                        {
                            uint8_t* q = row + ((size_t)xx * NC + c) * sizeof(S);
                            q[0] = (uint8_t)v[c];
                            if (kWide)
                                q[1] = (uint8_t)(v[c] >> 8);
                        }
                    }
                    xx += G;
                }
                JLS_LOCKSTEP();
                if (ending)
                {
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        corner[c] = first[c];
                    ++y;
                    phase = y == d.height ? kDrain : kLineStart;
                }
            }
        }
        {
            const bool draining = phase == kDrain;
            const uint32_t ahead = src.produced - p;
            if (draining && (src.ended || ahead >= 64u))
                phase = kDone;
        }
        if (__all(phase == kDone))
            break;
    }

    ScanResult r{kOk, 0, 0};
    if (!retry)
    {
        const uint32_t left = src.produced - p; // > 2^31 when the consumer ran past the producer
        const bool clean = src.u_marker != ~0ull && left < 15u && (left == 0 || field(peek32(ring, p), (int)left) == 0);
        if (clean)
            r.bytes = src.u_marker - src.u_begin;
        else
            retry = true;
    }
    if (retry)
        r.flags = fast::kFastRetry;
    if (live && sub == 0)
        results[scan] = r;
}

} // namespace jls
