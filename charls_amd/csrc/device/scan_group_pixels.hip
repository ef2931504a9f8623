// scan_group_pixels.hip -- speed path of the scan decoder for SAMPLE-INTERLEAVED scans (ILV_SAMPLE: 2..4 components coded
// pixel by pixel, reference src/scan_decoder_impl.hpp:162-261), lossless and near-lossless, several scans per wavefront;
// with one component per pixel also the speed path of NEAR-LOSSLESS single-component scans (the lossless ones have
// scan_group_decode.hip).
//
// Same organisation as scan_group_decode.hip -- the 64 lanes are split into groups of G lanes, every group decodes a scan
// of its own with all its state replicated over the group's lanes, control flow convergent for the wavefront, the lanes
// of a group sharing the bulk work (un-stuffing, run fills, colour transform and row stores) -- with the pixel as the
// unit of a step:
//   * the pixel is in run mode only when ALL its components have context 0 (src/scan_decoder_impl.hpp:196-204);
//     otherwise its components are decoded one after the other in regular mode on the ONE set of contexts, a component
//     with context 0 on context record 0 with positive sign;
//   * the components of a run-interruption pixel are decoded against run context 0 with the sign of Rb - Ra
//     (src/scan_decoder_impl.hpp:300-337, src/scan_decoder_core.hpp:72-100).
//
// LDS per scan: 365 context records, two run contexts, the dense bit ring and TWO lines of pixels (the previous and the
// current one, samples interleaved as in the user's row): nothing of a pixel's neighbourhood lives in registers, so a scan
// is described by its pixel index and its bit position alone, and whatever cannot decode a pixel leaves no state behind.
//
// Two step forms:
//   * the pixel loop for scans of 8-bit samples (lossless or near-lossless), written like the step loop of scan_group_decode.hip for the
//     number of instructions it issues (a lone wavefront issues one instruction every 4.3 - 5 cycles whatever it is):
//     the codes of ALL components of a pixel are cut from one 64-bit window of the bit ring, a component that meets the
//     context of an earlier component of its pixel takes that component's updated record from registers, the context
//     records and samples of a pixel are stored together once every component has decoded, and the contexts of the NEXT
//     pixel are worked out while this one decodes (they do not depend on it but for one gradient per component);
//   * the general step (NEAR >= 0, samples wider than 8 bits, escape codes, long prefixes) with the shared inlines of
//     scan_model.h; it also decodes the pixels the loop above stops at.
// As in scan_group_decode.hip a result is accepted only when the scan ends cleanly; everything else reports kFastRetry.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "scan_group_decode.hip"
#include "scan_model.h"

namespace jls {
namespace grp {

constexpr int kPixelStepsPerCheck = 16; // pixels between two looks at the producer

// Bytes of one line of pixels: pixel 0 is the left edge, 1 .. width the row, width + 1 the right edge, two more are read
// ahead; a multiple of 16 so that both lines place pixel 1 on a 16-byte boundary.
template <typename S>
__host__ __device__ constexpr uint32_t pixel_line_bytes(uint32_t width, uint32_t components)
{
    return ((width + 4) * components * (uint32_t)sizeof(S) + 15u) & ~15u;
}

template <typename S>
__host__ __device__ constexpr uint32_t pixel_lines_offset(uint32_t components)
{
    // first line: pixel 1 on a 16-byte boundary behind the ring (Layout::kRing + ring + its two mirror words)
    const uint32_t after_ring = Layout<S>::kRing + kRingWords * 4 + 16;
    return ((after_ring + components * (uint32_t)sizeof(S) + 15u) & ~15u) - components * (uint32_t)sizeof(S);
}

// `rows` = 1, or the number of components of a LINE-INTERLEAVED scan (then `components` is 1: every component keeps its
// own pair of lines).
template <typename S>
__host__ __device__ constexpr uint32_t pixel_region_bytes(uint32_t width, uint32_t components, uint32_t rows = 1)
{
    return bank_spread(pixel_lines_offset<S>(components) + 2 * rows * pixel_line_bytes<S>(width, components));
}

template <typename S>
__host__ __device__ constexpr uint32_t pixel_workgroup_lds_bytes(uint32_t width, uint32_t components, uint32_t scans_per_wave,
                                                                 uint32_t rows = 1)
{
    return Layout<S>::kLutBytes + scans_per_wave * pixel_region_bytes<S>(width, components, rows);
}

// The three ring words around bit p (see ring_words_at).
struct RingWords3
{
    uint32_t w0, w1, w2;
};
JLS_DEV RingWords3 ring_words3_at(uint32_t ring_address, uint32_t p)
{
    const uint32_t at = ring_address + (bit_field(p, 5, 8) << 2);
    return RingWords3{lds_load<uint32_t>(at), lds_load<uint32_t>(at + 4), lds_load<uint32_t>(at + 8)};
}

} // namespace grp

// Dynamic LDS: grp::pixel_workgroup_lds_bytes<S>(width, NC, 64 / G, NL).  NL = 2..4 (with NC = 1): a LINE-INTERLEAVED scan of
// NL components -- the lines of a pixel row are coded one component after the other, each against the line of its own
// component above it and with its own RUNindex, on the one set of contexts (src/scan_decoder_impl.hpp:62-129); every
// component keeps its own pair of lines, and a finished pixel row goes out interleaved.  (The lossless ones have the same
// arrangement in scan_group_decode.hip; these are the near-lossless ones.)
template <typename S, int G, int NC, int NL = 1>
__global__ void __launch_bounds__(64) decode_pixels_group(const ScanDesc* __restrict__ descs, ScanResult* __restrict__ results,
                                                          uint32_t count)
{
    using namespace grp;
    using L = Layout<S>;
    static_assert(G == 8 || G == 16 || G == 32, "lanes per scan");
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

    unsigned char* region = smem + L::kLutBytes + (size_t)sid * pixel_region_bytes<S>(width, NC, NL);
    Record* records = reinterpret_cast<Record*>(region + L::kRecords);
    RunCtx* run_ctx = reinterpret_cast<RunCtx*>(region + L::kRun);
    uint32_t* ring = reinterpret_cast<uint32_t*>(region + L::kRing);
    // sample c of pixel j of a line: [j * NC + c]
    S* line_a = reinterpret_cast<S*>(region + pixel_lines_offset<S>(NC));
    S* line_b = reinterpret_cast<S*>(region + pixel_lines_offset<S>(NC) + line_bytes);
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
        // (the host checks the thresholds of ONE scan of the launch: a scan with T3 beyond the table that leads its wavefront
        // must not write past the table's region -- it and its neighbours go to the exact decoder, see `usable`)
        for (int q = lane; q <= 2 * cap && q < (int)L::kLutBytes; q += 64)
            lut[q] = (unsigned char)(quantize(t_first, q - cap) + 4);
        for (uint32_t q = sub; q < 2 * NL * line_bytes / (uint32_t)sizeof(S); q += G)
            line_a[q] = 0;
        for (uint32_t q = sub; q <= kRingWords + 1; q += G)
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
    const bool usable = own_table && (!kWide || t_first.t3 <= kMaxTableT3) && lds_address(smem) == 0; // (see lds_load)
    int phase = !live || !usable ? kDone : (d.height == 0 ? kDrain : kLineStart);
    bool retry = live && !usable;
    uint32_t p = 0; // consumed dense bits
    uint32_t y = 0, i = 1;
    int run_index = 0;
    S* prev = line_a; // the two lines swap after every row
    S* cur = line_b;
    // line-interleaved scans: the component whose line is being decoded, which of its two lines is the current one, and the
    // RUNindex of every component
    int comp = 0, flip = 0;
    int run_index_of[NL];
#pragma unroll
    for (int c = 0; c < NL; ++c)
        run_index_of[c] = 0;
    const uint32_t line_samples = line_bytes / (uint32_t)sizeof(S);
    const uint32_t margin_bits = (uint32_t)(kPixelStepsPerCheck + 1) * NC * (uint32_t)t.limit + 320u;
    // the pixel loop takes every scan whose gradient table is the wavefront's (the scans of a wavefront share NEAR: it is part
    // of the table; `usable` above)
    const bool quick = true;

    auto quantised = [&](int diff) -> int { // quantised gradient + 4: 0 .. 8
        if (kWide)
            diff = diff < -cap ? -cap : (diff > cap ? cap : diff);
        return (int)lut[diff + cap];
    };

    // Errval of one regular-mode sample of context index idx; false = leave the scan to the exact decoder
    auto decode_regular = [&](int idx, int& errval, RegCtx& ctx, int& c_before) __attribute__((always_inline)) -> bool {
        const Record rec = records[idx];
        ctx = RegCtx{(int)rec.a, (int)rec.ncb >> 16, (int)(signed char)(rec.ncb >> 8), (int)(rec.ncb & 0xFFu)};
        c_before = ctx.c; // the prediction is corrected with C as it was BEFORE this sample's update
        const int k = regular_k(ctx);
        if (k >= 16)
            return false;
        const int u = take_unary(ring, p, t.limit); // no code has more zeros than LIMIT; anything longer: the exact decoder classifies it
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

    // One pixel in the general form for the lanes in `todo`: returns whether the pixel is in run mode (then nothing has
    // been decoded); otherwise its components are decoded and stored, or `retry` is raised.
    auto general_pixel = [&](bool todo) __attribute__((always_inline)) -> bool {
        const uint32_t at = todo ? i : 1u;
        int ra[NC], rb[NC], rc[NC], qs[NC];
        bool all_zero = true;
#pragma unroll
        for (int c = 0; c < NC; ++c)
        {
            ra[c] = (int)cur[(at - 1) * NC + c];
            rc[c] = (int)prev[(at - 1) * NC + c];
            rb[c] = (int)prev[at * NC + c];
            const int rd = (int)prev[(at + 1) * NC + c];
            qs[c] = 81 * (quantised(rd - rb[c]) - 4) + 9 * (quantised(rb[c] - rc[c]) - 4) + (quantised(rc[c] - ra[c]) - 4);
            all_zero = all_zero && qs[c] == 0;
        }
        const bool regular = todo && !all_zero;
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
            const int px = clamp_sample(t, med_predict(ra[c], rb[c], rc[c]) + ((c_before ^ s) - s));
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
                cur[i * NC + c] = (S)x[c];
            ++i;
        }
        JLS_LOCKSTEP();
        return todo && all_zero;
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
        // ---- first pixel of a line (src/scan_codec.hpp:189-195 per component): cur[0] = prev[1]; prev[0] is what was
        // cur[0] of the line above, i.e. that line's prev[1]
        {
            const bool starting = phase == kLineStart;
            if (__any(starting))
            {
                if (NL > 1 && starting)
                {
                    prev = line_a + (uint32_t)(2 * comp + flip) * line_samples;
                    cur = line_a + (uint32_t)(2 * comp + (flip ^ 1)) * line_samples;
#pragma unroll
                    for (int c = 0; c < NL; ++c)
                        run_index = comp == c ? run_index_of[c] : run_index;
                }
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
        bool in_run = false;
        bool stepped = false; // the pixel loop ran: the lanes it stopped at take ONE general step
        {
            const bool active = quick && phase == kInLine && !retry; // i <= width: the end of a line is handled at once
            const LaneMask active_m = lanes_where(active);
            // the loop in two renderings: lossless, and near-lossless (de-quantised error, reconstruction with the range
            // fix of src/default_traits.hpp:118-141, no k = 0 correction)
            auto pixel_loop = [&](auto near_tag) __attribute__((always_inline)) {
                constexpr bool kNearLoop = decltype(near_tag)::value;
                const int near = t.near, step_size = 2 * t.near + 1, range_span = t.range * (2 * t.near + 1);
                stepped = true;
                const uint32_t rest_of_line = width + 1 - i;
                uint32_t steps = kPixelStepsPerCheck;
                while (lanes_where(active && rest_of_line < steps) != 0)
                    --steps;
                uint32_t ticker = 1u << (steps - 1);
                const uint32_t limit_v = opaque(active ? (uint32_t)(t.limit - t.qbpp - 1) : 0u);
                const int maxval = t.maxval, reset = t.reset;
                const uint32_t ring_address = opaque(lds_address(ring));
                const uint32_t records_address = opaque(lds_address(records));
                // scans outside a line stay on pixel 1 of their (dead) lines and store nothing
                const S* pp = prev + (active ? i : 1u) * NC; // pixel i of the previous line
                S* cp = cur + (active ? i : 1u) * NC;
                const uint32_t advance = active ? (uint32_t)NC : 0u;
                // what a pixel brings into its iteration: per component Q + 364 (the three gradients come with + 4 each),
                // Q1 + 4 (the next pixel's Q2 + 4), Ra, Rb, Rc, and its context record as read before the pixel
                int qsu[NC], q1[NC], ra[NC], rb[NC], rc[NC];
                Record rec[NC];
                auto index_of = [&](int q) -> uint32_t { return abs_difference((uint32_t)q, 364u); };
                // the table is at LDS address 0: -255 .. 255 for 8-bit samples, -T3 .. T3 (differences clamped: one more
                // instruction) for wider ones
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
                    const int rd = (int)pp[c + NC];
                    q1[c] = gradient(rd - rb[c]);
                    qsu[c] = mad24(mad24(q1[c], 9, gradient(rb[c] - rc[c])), 9, gradient(rc[c] - ra[c]));
                    rec[c] = record_at(index_of(qsu[c]));
                }
                RingWords3 words = ring_words3_at(ring_address, p);
                uint32_t k_seen = 0, mm_seen = 0;
                LaneMask ok_m;
                do
                {
                    // the next pixel's Rb and Rd: reads that depend on nothing of this pixel go first
                    const S* const pn = pp + advance;
                    int rb_next[NC], rd_next[NC], q1_next[NC], q3_next[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                    {
                        rb_next[c] = (int)pn[c];
                        rd_next[c] = (int)pn[c + NC];
                    }
                    // the next 64 bits of the stream: bit j of the pair is stream bit p + j (v_alignbit shifts by p mod 32)
                    const uint32_t lo = funnel_shift(words.w1, words.w0, p);
                    const uint32_t hi = funnel_shift(words.w2, words.w1, p);
                    const uint64_t window = ((uint64_t)hi << 32) | lo;
                    // a pixel in run mode: every component has context 0
                    LaneMask run_m = lanes_where(qsu[0] == 364);
#pragma unroll
                    for (int c = 1; c < NC; ++c)
                        run_m &= lanes_where(qsu[c] == 364);
                    ok_m = ~run_m;
                    uint32_t taken = 0; // bits of this pixel's earlier components: at most 32 for the window to hold the next code
                    uint32_t k_pixel = 0, mm_pixel = 0;
                    int x[NC];
                    uint32_t idx[NC];
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
                        const uint32_t win = (uint32_t)(window >> taken);
                        const uint32_t u = lowest_one(win); // length of the unary prefix; 0xFFFFFFFF for an all-zero window
                        const uint32_t u1 = u + 1u;
                        const uint32_t beyond = bit_reverse(win >> (u1 & 31u));
                        ok_m &= lanes_where(u < limit_v);
                        const int n = (int)(r.ncb & 0xFFu);
                        const int cc = (int)(signed char)(r.ncb >> 8);
                        const int bb = (int)r.ncb >> 16;
                        // k = min{k : N << k >= A} from the exponents of A and N (see scan_group_decode.hip)
                        const int k_raw = ((int)(float_bits(r.a) - float_bits((uint32_t)n)) + 0x7FFFFF) >> 23;
                        const int k = k_raw < 0 ? 0 : k_raw;
                        const int mm = (int)((u << k) | (uint32_t)(((uint64_t)beyond << k) >> 32));
                        if (kWide) // the code has to lie inside the 32-bit window (8-bit samples: u < 23 and k <= 9)
                            ok_m &= lanes_where(u1 + (uint32_t)k <= 32u);
                        const int half = mm >> 1;
                        const int odd = kNearLoop ? (mm & 1) : ((mm ^ (((k - 1) & (2 * bb + n - 1)) >> 31)) & 1);
                        const int e = half ^ -odd;
                        const int sgn = qsu[c] < 364 ? -1 : 1;
                        const int px0 = med3(ra[c] + (rb[c] - rc[c]), ra[c], rb[c]);
                        const int px = med3(mad24(cc, sgn, px0), 0, maxval);
                        if (kNearLoop)
                        {
                            int v = mad24(e * sgn, step_size, px);
                            v += v < -near ? range_span : (v > maxval + near ? -range_span : 0);
                            x[c] = med3(v, 0, maxval);
                        }
                        else
                            x[c] = mad24(e, sgn, px) & maxval;
                        q3_next[c] = gradient(rb[c] - x[c]); // the next pixel's Rc - Ra
                        if (c == 0)
                        {
#pragma unroll
                            for (int f = 0; f < NC; ++f)
                                q1_next[f] = gradient(rd_next[f] - rb_next[f]);
                        }
                        taken += u1 + (uint32_t)k;
                        if (c + 1 < NC)
                            ok_m &= lanes_where(taken <= 32u);
                        k_pixel |= (uint32_t)k;
                        mm_pixel |= (uint32_t)mm;
                        // A.12 / A.13, src/regular_mode_context.hpp:45-93, in the median form of scan_group_decode.hip
                        int u_a = (int)r.a + half + odd;
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
                    // the next pixel's contexts (Rc = this pixel's Rb, Ra = this pixel's samples)
                    uint32_t at_next[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                    {
                        qsu[c] = mad24(mad24(q1_next[c], 9, q1[c]), 9, q3_next[c]);
                        q1[c] = q1_next[c];
                        rc[c] = rb[c];
                        rb[c] = rb_next[c];
                        ra[c] = x[c];
                        at_next[c] = records_address + (index_of(qsu[c]) << 3);
                    }
                    // the pixel is complete: its records and samples go to LDS together (a lane that could not decode one of
                    // its components stores nothing and keeps its position; the loop ends for everybody)
                    JLS_LOCKSTEP();
                    if (lane_of(ok_m))
                    {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                        {
                            const uint32_t at = records_address + (idx[c] << 3);
                            lds_store<uint32_t>(at, updated[c].a); // in component order: the last of equal contexts stays
                            lds_store<uint32_t>(at + 4, updated[c].ncb);
                            cp[c] = (S)x[c];
                        }
                        p += taken;
                        ++i;
                        k_seen |= k_pixel;
                        mm_seen |= mm_pixel;
                    }
                    JLS_LOCKSTEP_STORES();
                    pp = pn;
                    cp += advance;
                    // the records and the bit window of the next pixel: read behind the stores
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        rec[c] = Record{lds_load<uint32_t>(at_next[c]), lds_load<uint32_t>(at_next[c] + 4)};
                    words = ring_words3_at(ring_address, p);
                    ticker = tick(ticker, active_m, ok_m);
                } while (ticker != 0);
                // the reference raises invalid_data for k >= 16; valid streams of 8-bit samples keep k <= 9 and the mapped
                // error below RANGE (anything else is left to the exact decoder as a whole)
                if (active && (k_seen >= (kWide ? 16u : 10u) || (mm_seen >> t.qbpp) != 0u))
                    retry = true;
                // one general step for the lanes the loop stopped at
                const bool stopped = active && !lane_of(ok_m) && !retry;
                in_run = general_pixel(stopped);
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
            for (int step = 0; step < kPixelStepsPerCheck; ++step)
            {
                const bool todo = phase == kInLine && i <= width && !retry;
                in_run = general_pixel(todo);
                if (__any(in_run || retry || (phase == kInLine && i > width)))
                    break;
            }
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
            int a[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c)
                a[c] = (int)cur[((in_run ? i : 1u) - 1) * NC + c];
            JLS_LOCKSTEP();
            {
                uint32_t r = (uint32_t)sub;
                while (__any(in_run && r < run))
                {
                    if (in_run && r < run)
                    {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            cur[(i + r) * NC + c] = (S)a[c];
                    }
                    r += G;
                }
            }
            const uint32_t at = i + run;
            JLS_LOCKSTEP();
            int x[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c)
            { // sample-interleaved: every component against run context 0, in component order
              // (src/scan_decoder_impl.hpp:300-337); a single component: context 1 when Ra and Rb are within NEAR
              // (src/scan_decoder_impl.hpp:264-298)
                const int rb_at = (int)prev[(interrupted ? at : 1u) * NC + c];
                const int which = (NC == 1 && is_near(t, a[c], rb_at)) ? 1 : 0;
                RunCtx ctx = run_ctx[which];
                x[c] = 0;
                if (interrupted)
                {
                    const int k = run_k(ctx);
                    const int limit = t.limit - run_j(run_index) - 1;
                    const int u = k > 24 ? -1 : take_unary(ring, p, t.limit);
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
                        x[c] = which ? reconstruct(t, a[c], e) : reconstruct(t, rb_at, e * ((rb_at - a[c]) < 0 ? -1 : 1));
                    }
                }
                JLS_LOCKSTEP();
                if (interrupted)
                    run_ctx[which] = ctx;
                JLS_LOCKSTEP();
            }
            if (interrupted)
            {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    cur[at * NC + c] = (S)x[c];
                if (run_index > 0)
                    --run_index;
                i = at + 1;
            }
            else if (in_run && !retry)
                i = width + 1; // the run reached the end of the line
            JLS_LOCKSTEP();
        }

        if (retry)
            phase = kDone;

        // ---- finished line -> user's row (inverse colour transform: src/copy_from_line_buffer.hpp:19-191); the line
        // becomes the previous one
        if (NL == 1)
        {
            const bool ending = phase == kInLine && i > width;
            if (__any(ending))
            {
                uint8_t* row = d.pixels + (size_t)y * d.pixel_stride;
                const S* samples = cur + NC; // pixel 1
                const bool transformed = NC == 3 && d.color_transformation != 0;
                const uint32_t row_bytes = width * NC * (uint32_t)sizeof(S);
                const bool aligned = ((reinterpret_cast<uintptr_t>(d.pixels) | d.pixel_stride) & 15u) == 0;
                const uint32_t wide_bytes = aligned && !transformed ? row_bytes & ~15u : 0u;
                uint32_t off = (uint32_t)sub * 16u;
                while (__any(ending && off < wide_bytes))
                {
                    if (ending && off < wide_bytes)
                        *reinterpret_cast<uint4*>(row + off) =
                            *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(samples) + off);
                    off += G * 16u;
                }
                if (__any(ending && transformed))
                {
                    uint32_t xx = (uint32_t)sub;
                    while (__any(ending && transformed && xx < width))
                    {
                        if (ending && transformed && xx < width)
                        {
                            unsigned v[3];
                            hp_inverse(d.color_transformation, kWide, (int)samples[xx * NC], (int)samples[xx * NC + 1],
                                       (int)samples[xx * NC + (NC > 2 ? 2 : 0)], v);
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                reinterpret_cast<S*>(row)[xx * NC + (c < NC ? c : 0)] = (S)v[c];
                        }
                        xx += G;
                    }
                }
                uint32_t ss = wide_bytes / (uint32_t)sizeof(S) + (uint32_t)sub;
                while (__any(ending && !transformed && ss < width * NC))
                {
                    if (ending && !transformed && ss < width * NC)
                        reinterpret_cast<S*>(row)[ss] = samples[ss];
                    ss += G;
                }
                if (ending && sub < NC)
                    cur[(width + 1) * NC + sub] = cur[width * NC + sub]; // the right edge of the next line's previous line
                JLS_LOCKSTEP();
                if (ending)
                {
                    S* const was_prev = prev;
                    prev = cur;
                    cur = was_prev;
                    ++y;
                    phase = y == d.height ? kDrain : kLineStart;
                }
            }
        }
        else
        { // a component's line; behind the last component the pixel row goes out, interleaved
            const bool ending = phase == kInLine && i > width;
            if (__any(ending))
            {
                const bool row_done = ending && comp == NL - 1;
                if (__any(row_done))
                {
                    uint8_t* row = d.pixels + (size_t)y * d.pixel_stride;
                    const bool transformed = NL == 3 && d.color_transformation != 0;
                    uint32_t xx = (uint32_t)sub;
                    while (__any(row_done && xx < width))
                    {
                        if (row_done && xx < width)
                        {
                            unsigned v[4];
#pragma unroll
                            for (int c = 0; c < NL; ++c)
                                v[c] = line_a[(uint32_t)(2 * c + (flip ^ 1)) * line_samples + 1 + xx];
                            if (transformed)
                                hp_inverse(d.color_transformation, kWide, (int)v[0], (int)v[1], (int)v[NL > 2 ? 2 : 0], v);
#pragma unroll
                            for (int c = 0; c < NL; ++c)
                            { // (bytes: the user's row of 16-bit pixels need not be aligned)
                                uint8_t* q = row + ((size_t)xx * NL + c) * sizeof(S);
                                q[0] = (uint8_t)v[c];
                                if (kWide)
                                    q[1] = (uint8_t)(v[c] >> 8);
                            }
                        }
                        xx += G;
                    }
                }
                if (ending && sub == 0)
                    cur[width + 1] = cur[width]; // the right edge of the next row's previous line
                JLS_LOCKSTEP();
                if (ending)
                {
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
