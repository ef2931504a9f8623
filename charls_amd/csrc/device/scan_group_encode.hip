// scan_group_encode.hip -- the scan ENCODER for the modes whose causal template holds RECONSTRUCTED samples (near-lossless
// coding: nothing is known ahead of the serial chain), several scans per wavefront.
//
// encode_scans_serial (scan_serial.hip) codes such a scan with ONE lane of a wavefront and its line window in global
// memory -- about 0.25 MPix/s per scan whatever the batch.  Here, as in scan_group_decode.hip / scan_group_pixels.hip,
// the 64 lanes are split into groups of G lanes and every group codes a scan of its own:
//   * the scan's state lives on chip: 365 context records and the two run contexts in LDS, Ra / Rc / run index / bit
//     writer in registers (replicated over the group's lanes), ONE line of reconstructed samples per component in LDS
//     (a reconstructed sample overwrites the slot of the sample above it once that one has been read);
//   * the G lanes of a group share the bulk work: the source row is fetched, masked and colour-transformed
//     (src/copy_to_line_buffer.hpp:21-262) into LDS by all of them at the start of a line, and the coded bytes go from
//     an LDS staging ring to the destination in cooperative copies;
//   * the pixel is the unit of a step (1 component for planar scans, 2..4 for ILV_SAMPLE): contexts of all components,
//     run mode when all are 0, otherwise the components one after the other in regular mode
//     (src/scan_encoder_impl.hpp:109-302).
// The bit writer is the reference's (32-bit accumulator, four-byte flushes, a 7-bit byte after every 0xFF,
// src/scan_encoder.hpp:75-186) with its capacity accounting, so destination_too_small is raised for exactly the same
// destination sizes; only where the bytes land differs (the staging ring).  Output is byte-identical to
// scan_encoder::encode_scan.  Arithmetic is the general one of scan_model.h (NEAR >= 0), not tuned for instruction count.
#pragma once
#include <hip/hip_runtime.h>

#include "scan_model.h"

namespace jls {
namespace grp {

constexpr uint32_t kOutRingBytes = 2048; // coded bytes staged in LDS per scan
constexpr int kEncodeStepsPerCheck = 8;  // pixels between two looks at the staging ring

template <typename S>
struct EncodeLayout
{
    static constexpr uint32_t kRecords = 0;                     // 365 x RegCtx (16 B)
    static constexpr uint32_t kRun = 365 * 16 + 8;              // 2 x RunCtx
    static constexpr uint32_t kOut = kRun + 32 + 8;             // staging ring
    static constexpr uint32_t kLines = kOut + kOutRingBytes;    // NC reconstructed lines, then NC source lines
};

template <typename S>
__host__ __device__ constexpr uint32_t encode_line_samples(uint32_t width)
{
    return (width + 6 + 7) & ~7u;
}

template <typename S>
__host__ __device__ constexpr uint32_t encode_region_bytes(uint32_t width, uint32_t components)
{
    return (EncodeLayout<S>::kLines + 2 * components * encode_line_samples<S>(width) * (uint32_t)sizeof(S) + 15u) & ~15u;
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

// Dynamic LDS: (64 / G) * grp::encode_region_bytes<S>(width, NC).  NC = 1: a single-component scan (planar); NC = 2..4:
// a sample-interleaved scan of NC components.
template <typename S, int G, int NC>
__global__ void __launch_bounds__(64) encode_pixels_group(const ScanDesc* __restrict__ descs, ScanResult* __restrict__ results,
                                                          uint32_t count)
{
    using namespace grp;
    using L = EncodeLayout<S>;
    static_assert(G == 8 || G == 16 || G == 32 || G == 64, "lanes per scan");
    static_assert(NC >= 1 && NC <= 4, "components per pixel");
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
    const uint32_t line_samples = encode_line_samples<S>(width);
    const int mask = (1 << d.bits_per_sample) - 1;

    unsigned char* region = smem + (size_t)sid * encode_region_bytes<S>(width, NC);
    RegCtx* records = reinterpret_cast<RegCtx*>(region + L::kRecords);
    RunCtx* run_ctx = reinterpret_cast<RunCtx*>(region + L::kRun);
    uint8_t* out_ring = region + L::kOut;
    S* lines = reinterpret_cast<S*>(region + L::kLines);          // reconstructed: component c at lines + c * line_samples
    S* source = lines + (uint32_t)NC * line_samples;              // this row's samples as the codec sees them, same layout
    auto line_of = [&](int c) -> S* { return lines + (uint32_t)c * line_samples; };
    auto source_of = [&](int c) -> S* { return source + (uint32_t)c * line_samples; };
    {
        const RegCtx fresh{initial_a(t), 0, 0, 1};
        for (int q = sub; q < 365; q += G)
            records[q] = fresh;
        if (sub < 2)
            run_ctx[sub] = RunCtx{sub, initial_a(t), 1, 0};
        for (uint32_t q = sub; q < 2u * NC * line_samples; q += G)
            lines[q] = 0;
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
    int a[NC], rc[NC], corner[NC], first[NC], q_prev[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
        a[c] = rc[c] = corner[c] = first[c] = q_prev[c] = 0;

    // staged bytes -> destination, by the lanes of the group (everything written so far, or whole 256-byte pieces)
    auto drain = [&](bool wanted, bool everything) {
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
        // ---- a new line: fetch the row (src/copy_to_line_buffer.hpp), edge samples (src/scan_codec.hpp:189-195)
        {
            const bool starting = phase == kLineStart;
            if (__any(starting))
            {
                const uint8_t* row = d.pixels + (size_t)y * d.pixel_stride;
                uint32_t xx = (uint32_t)sub;
                while (__any(starting && xx < width))
                {
                    if (starting && xx < width)
                    {
                        unsigned v[4] = {0, 0, 0, 0};
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                        {
                            const uint8_t* q = row + ((size_t)xx * NC + c) * sizeof(S);
                            v[c] = kWide ? (unsigned)q[0] | ((unsigned)q[1] << 8) : (unsigned)q[0];
                        }
                        if (NC == 3 && d.color_transformation != 0)
                            hp_forward(d.color_transformation, kWide, (int)v[0], (int)v[1], (int)v[2], v);
                        else
                        {
#pragma unroll
                            for (int c = 0; c < NC; ++c)
                                v[c] &= (unsigned)mask;
                        }
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            source_of(c)[1 + xx] = (S)v[c];
                    }
                    xx += G;
                }
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
                        rc[c] = corner[c]; // prev[0]
                        a[c] = rb;         // cur[0] = prev[1]
                        first[c] = rb;
                        q_prev[c] = quantize(t, rb - rc[c]);
                    }
                    phase = kInLine;
                }
            }
        }
        // ---- pixels
        for (int step = 0; step < kEncodeStepsPerCheck; ++step)
        {
            const bool active = phase == kInLine && i <= width && bw.err == kOk;
            int rb[NC], qs[NC], q1[NC];
            bool all_zero = true;
#pragma unroll
            for (int c = 0; c < NC; ++c)
            {
                rb[c] = (int)line_of(c)[active ? i : 0];
                const int rd = (int)line_of(c)[active ? i + 1 : 0];
                q1[c] = quantize(t, rd - rb[c]);
                qs[c] = 81 * q1[c] + 9 * q_prev[c] + quantize(t, rc[c] - a[c]);
                all_zero = all_zero && qs[c] == 0;
            }
            const bool regular = active && !all_zero;
            const bool in_run = active && all_zero;
            // -- regular mode, src/scan_encoder_core.hpp:40-67, component after component on the one set of contexts
            int rx[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c)
            {
                const int s = qs[c] >> 31;
                const int idx = (qs[c] ^ s) - s;
                RegCtx ctx = records[idx];
                const int x = (int)source_of(c)[regular ? i : 1];
                const int k = regular_k(ctx);
                const int px = clamp_sample(t, med_predict(a[c], rb[c], rc[c]) + ((ctx.c ^ s) - s));
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
                    records[idx] = ctx;
                JLS_LOCKSTEP();
            }
            if (regular)
            {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                {
                    line_of(c)[i] = (S)rx[c];
                    a[c] = rx[c];
                    rc[c] = rb[c];
                    q_prev[c] = q1[c];
                }
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
                            near_all = near_all && is_near(t, (int)source_of(c)[i + run], a[c]);
                        if (!near_all)
                            break;
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            line_of(c)[i + run] = (S)a[c];
                        if (++run == remaining)
                            break;
                    }
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
                    const int ra = a[c];
                    const int rb_at = (int)line_of(c)[interrupted ? at : 0]; // prev[at]: not overwritten
                    const int x = (int)source_of(c)[interrupted ? at : 1];
                    const int which = (NC == 1 && is_near(t, ra, rb_at)) ? 1 : 0;
                    const int sg = which ? 1 : ((rb_at - ra) < 0 ? -1 : 1);
                    const int e = which ? error_value(t, x - ra) : error_value(t, (x - rb_at) * sg);
                    RunCtx ctx = run_ctx[which];
                    if (interrupted)
                    {
                        const int k = run_k(ctx);
                        const int map = run_map(ctx, e, k);
                        const int em = 2 * (e < 0 ? -e : e) - ctx.ritype - map;
                        bw.golomb(t, k, em, t.limit - run_j(run_index) - 1);
                        run_update(ctx, e, em, t.reset);
                    }
                    const int rec = which ? reconstruct(t, ra, e) : reconstruct(t, rb_at, e * sg);
                    JLS_LOCKSTEP();
                    if (interrupted)
                    {
                        run_ctx[which] = ctx;
                        line_of(c)[at] = (S)rec;
                        a[c] = rec;
                        rc[c] = rb_at;
                    }
                    JLS_LOCKSTEP();
                }
                if (interrupted)
                {
                    if (run_index > 0)
                        --run_index;
                    i = at + 1;
                    if (i <= width)
                    {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            q_prev[c] = quantize(t, (int)line_of(c)[i] - rc[c]);
                    }
                }
            }
            JLS_LOCKSTEP();
            if (__any(bw.err != kOk || (phase == kInLine && i > width) || (phase != kDone && bw.written - copied >= kOutRingBytes / 2)))
                break;
        }
        if (bw.err != kOk && phase != kDone)
            phase = kDone;
        // ---- end of a line
        {
            const bool ending = phase == kInLine && i > width;
            if (ending)
            {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    corner[c] = first[c];
                ++y;
                phase = y == d.height ? kFinish : kLineStart;
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
