// scan_group_decode.hip -- speed path of the scan decoder for lossless single-component scans: SEVERAL scans per wavefront.
//
// Decoding one scan is one dependency chain (bit position -> k -> context -> reconstructed sample, reference
// src/scan_decoder_core.hpp:38-69), so the only parallelism is the number of scans in flight, and what a wavefront pays
// per decoded sample is the number of instructions it has to issue (a wavefront issues at most one instruction every
// four cycles).  scan_fast_decode.hip spends a whole 64-lane instruction on the value of ONE scan; here the 64 lanes are
// split into groups of G lanes and every group decodes a scan of its own, so one instruction advances 64 / G scans:
//
//   * all per-scan state (window of the previous line, Ra, bit position, producer state) lives in vector registers,
//     replicated over the G lanes of the group; nothing is wave-uniform, there is no scalar chain;
//   * control flow stays convergent for the whole wavefront: every step all groups decode one regular-mode sample
//     under a per-lane predicate; a group that meets anything else (run mode, an escape or long code, the end of its
//     line, an empty bit ring) raises an event, the wavefront leaves the step loop, and the event is handled once, out
//     of line, under the predicate of the groups that raised it;
//   * the G lanes of a group share the bulk work of their scan: un-stuffing 16 coded bytes per lane into the dense
//     bit ring (as in scan_fast_decode.hip), run fills, and the 16-byte row stores of every finished line;
//   * LDS per scan: 365 context records (8 B), two run contexts, a 1 KB dense bit ring, the gradient table and ONE line
//     of samples = 8.6 KB for 4096 8-bit samples, 34.5 KB per wavefront at G = 16: four wavefronts per CU, one per SIMD.
//
// Like scan_fast_decode.hip this is not a restatement of the reference's bit reader: a result is accepted only when the
// scan ends cleanly (all samples decoded inside the entropy-coded segment, zero padding, marker next); everything else
// reports flags = kFastRetry and the exact decoder decides (runtime.hip: launch_decode_plain).
#pragma once
#include <hip/hip_runtime.h>

#include "scan_model.h"
#include "scan_fast_decode.hip"

namespace jls {
namespace grp {

constexpr uint32_t kRingWords = 256;               // dense bits resident per scan: 8192 (word kRingWords mirrors word 0)
constexpr uint32_t kRingBits = kRingWords * 32;
constexpr int kStepsPerCheck = 16;                 // regular-mode steps between two looks at the producer
constexpr uint32_t kMarginBits = kStepsPerCheck * 32 + 320; // dense bits the step loop and one event handler may consume
constexpr int kMaxTableT3 = 1023;                  // widest gradient table (2 * T3 + 1 entries) for samples wider than 8 bits

// Per-scan LDS region.  The line starts one sample before a 16-byte boundary so that sample 1 (the first of the row) is
// aligned for the 16-byte row stores.
template <typename S>
struct Layout
{
    static constexpr uint32_t kRecords = 0;                       // 365 x 8 B (+ pad)
    static constexpr uint32_t kRun = 2928;                        // 2 x RunCtx
    static constexpr uint32_t kRing = kRun + 32;                  // kRingWords + 1 words
    static constexpr uint32_t kLut = kRing + kRingWords * 4 + 16; // gradient table
    static constexpr uint32_t kLutBytes = sizeof(S) == 1 ? 512 : 2 * kMaxTableT3 + 2;
    static constexpr uint32_t kLine = kLut + kLutBytes + 16 - sizeof(S);
};

template <typename S>
__host__ __device__ constexpr uint32_t region_bytes(uint32_t width)
{
    return (Layout<S>::kLine + (width + 6) * (uint32_t)sizeof(S) + 15u) & ~15u;
}

// Regular-mode context record: word 0 = A, word 1 = N | (C & 0xFF) << 8 | B << 16.  N <= RESET <= 255, -128 <= C <= 127
// and -N < B <= 0 after A.13 (reference src/regular_mode_context.hpp:45-93), so the record is exact.
struct Record
{
    uint32_t a;
    uint32_t ncb;
};

// 32 dense bits starting at bit p (MSB first).
JLS_DEV uint32_t peek32(const uint32_t* ring, uint32_t p)
{
    const uint32_t wi = (p >> 5) & (kRingWords - 1);
    const uint64_t both = ((uint64_t)ring[wi] << 32) | ring[wi + 1];
    return (uint32_t)((both << (p & 31)) >> 32);
}

JLS_DEV uint32_t take_bits(const uint32_t* ring, uint32_t& p, int n) // 0 <= n <= 32
{
    const uint32_t w = peek32(ring, p);
    p += (uint32_t)n;
    return n == 0 ? 0u : w >> (32 - n);
}

// Producer of one scan's dense bit ring (all members replicated over the lanes of the group).
struct Producer
{
    const uint8_t* gbase; // 16-byte aligned origin of the coded stream
    uint64_t u_next;      // next coded byte to un-stuff (u = offset + misalignment)
    uint64_t u_end;
    uint64_t u_begin;
    uint64_t u_marker;    // position of the terminating marker once seen (else ~0)
    uint32_t produced;    // dense bits written so far (mod 2^32)
    uint32_t prev_byte;
    bool ended;           // marker or end of source reached: `produced` is final
};

// OR `n` (1..8) bits, right aligned in v, at dense bit position p.
JLS_DEV void put_bits(uint32_t* ring, uint32_t p, uint32_t v, int n)
{
    const uint32_t q = (p >> 5) & (kRingWords - 1);
    const int off = (int)(p & 31);
    const int room = 32 - off;
    if (n <= room)
    {
        atomicOr(&ring[q], v << (room - n));
        if (q == 0)
            atomicOr(&ring[kRingWords], v << (room - n));
    }
    else
    {
        atomicOr(&ring[q], v >> (n - room));
        if (q == 0)
            atomicOr(&ring[kRingWords], v >> (n - room));
        const uint32_t q2 = (q + 1) & (kRingWords - 1);
        atomicOr(&ring[q2], v << (32 - (n - room)));
        if (q2 == 0)
            atomicOr(&ring[kRingWords], v << (32 - (n - room)));
    }
}

// Un-stuffs G x 16 coded bytes of every scan whose lanes pass `want` (JPEG-LS stuffing is byte aligned in the coded
// stream: the byte after a 0xFF carries 7 payload bits; a 0xFF followed by a byte >= 0x80 is a marker).  Called by all
// 64 lanes; the shuffles only ever read lanes of the caller's own group.
template <int G>
JLS_DEV void refill(Producer& s, uint32_t* ring, bool want, int lane, int sub)
{
    constexpr uint32_t kChunk = G * 16;
    // 1) clear the words this refill may touch (everything after the word holding `produced`)
    if (want)
    {
        const uint32_t first = (s.produced + 31) >> 5;
        for (uint32_t j = sub; j < kChunk / 4 + 2; j += G)
        {
            const uint32_t q = (first + j) & (kRingWords - 1);
            ring[q] = 0;
            if (q == 0)
                ring[kRingWords] = 0;
        }
    }
    JLS_LOCKSTEP();
    // 2) every lane takes 16 coded bytes
    const uint64_t u0 = s.u_next + (uint64_t)sub * 16;
    uint4 raw = make_uint4(0, 0, 0, 0);
    uint32_t next_first = 0; // coded byte following this lane's 16 (for the marker test of its last byte)
    if (want && u0 < s.u_end)
    {
        raw = *reinterpret_cast<const uint4*>(s.gbase + u0);
        if (u0 + 16 < s.u_end)
            next_first = s.gbase[u0 + 16];
    }
    const uint32_t words[4] = {raw.x, raw.y, raw.z, raw.w};
    const uint32_t last = words[3] >> 24;
    uint32_t before = __shfl_up(last, 1);
    if (sub == 0)
        before = s.prev_byte;
    int nbits[16];
    uint32_t bytes[16];
    int marker_at = 16;
    uint32_t prev = before;
    int total = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j)
    {
        const uint32_t b = (words[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
        const uint64_t u = u0 + (uint64_t)j;
        bytes[j] = b;
        int n = 0;
        if (u == s.u_begin)
            prev = 0; // the first coded byte has no predecessor
        if (want && u >= s.u_begin && u < s.u_end && marker_at == 16)
        {
            const uint32_t nb = j < 15 ? ((words[(j + 1) >> 2] >> (((j + 1) & 3) * 8)) & 0xFFu) : next_first;
            const bool is_marker = b == 0xFFu && (u + 1 >= s.u_end || (nb & 0x80u) != 0);
            if (is_marker)
                marker_at = j;
            else
                n = prev == 0xFFu ? 7 : 8;
        }
        nbits[j] = n;
        total += n;
        prev = b;
    }
    // lanes of the group after its first marker contribute nothing
    const unsigned long long markers = __ballot(marker_at < 16);
    const int group_base = lane - sub;
    const uint32_t group_markers = (uint32_t)(markers >> group_base) & (G == 32 ? 0xFFFFFFFFu : ((1u << (G & 31)) - 1u));
    const unsigned long long group_markers64 = G == 64 ? markers : (unsigned long long)group_markers;
    const int first_marker = group_markers64 ? __ffsll(group_markers64) - 1 : 64; // lane index inside the group
    if (sub > first_marker)
        total = 0;
    // exclusive prefix sum of the group's bit counts
    int inc = total;
    for (int delta = 1; delta < G; delta <<= 1)
    {
        const int up = __shfl_up(inc, delta);
        if (sub >= delta)
            inc += up;
    }
    uint32_t p = s.produced + (uint32_t)(inc - total);
    if (want && sub <= first_marker)
    {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (nbits[j] != 0)
            {
                put_bits(ring, p, nbits[j] == 7 ? (bytes[j] & 0x7Fu) : bytes[j], nbits[j]);
                p += (uint32_t)nbits[j];
            }
    }
    const int chunk_bits = __shfl(inc, group_base + G - 1);
    const uint32_t chunk_last = __shfl(last, group_base + G - 1);
    const int marker_j = __shfl(marker_at, group_base + (first_marker < G ? first_marker : 0));
    JLS_LOCKSTEP();
    if (want)
    {
        s.produced += (uint32_t)chunk_bits;
        s.prev_byte = chunk_last;
        if (first_marker < G)
        {
            s.u_marker = s.u_next + (uint64_t)first_marker * 16 + (uint64_t)marker_j;
            s.ended = true;
        }
        s.u_next += kChunk;
        if (s.u_next >= s.u_end)
            s.ended = true;
    }
}

// Number of zero bits before the next one bit, which is consumed as well; -1 when it exceeds `most`.  Per lane.
JLS_DEV int take_unary(const uint32_t* ring, uint32_t& p, int most)
{
    int total = 0;
    for (;;)
    {
        const uint32_t w = peek32(ring, p);
        const int u = w == 0 ? 32 : __clz((int)w);
        if (u < 32)
        {
            p += (uint32_t)(u + 1);
            total += u;
            return total > most ? -1 : total;
        }
        p += 32;
        total += 32;
        if (total > most)
            return -1;
    }
}

} // namespace grp

// Dynamic LDS: (64 / G) * grp::region_bytes<S>(width).  `count` scans, 64 / G of them per workgroup of one wavefront.
template <typename S, int G>
__global__ void __launch_bounds__(64) decode_scans_group(const ScanDesc* __restrict__ descs, ScanResult* __restrict__ results,
                                                         uint32_t count)
{
    using namespace grp;
    using L = Layout<S>;
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

    unsigned char* region = smem + (size_t)sid * region_bytes<S>(width);
    Record* records = reinterpret_cast<Record*>(region + L::kRecords);
    RunCtx* run_ctx = reinterpret_cast<RunCtx*>(region + L::kRun);
    uint32_t* ring = reinterpret_cast<uint32_t*>(region + L::kRing);
    signed char* lut = reinterpret_cast<signed char*>(region + L::kLut);
    S* line = reinterpret_cast<S*>(region + L::kLine);
    const int cap = kWide ? t.t3 : 255; // the table covers gradients -cap .. cap; beyond that the magnitude is 4

    {
        const Record fresh{(uint32_t)initial_a(t), 1u};
        for (int q = sub; q < 366; q += G)
            records[q] = fresh;
        if (sub < 2)
            run_ctx[sub] = RunCtx{sub, initial_a(t), 1, 0};
        for (int q = sub; q <= 2 * cap; q += G)
            lut[q] = (signed char)quantize(t, q - cap);
        for (uint32_t q = sub; q < width + 6; q += G)
            line[q] = 0;
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

    enum : int { kNone = 0, kRun, kSlow };
    enum : int { kLineStart = 0, kInLine, kDrain, kDone };
    int phase = !live ? kDone : (d.height == 0 ? kDrain : kLineStart);
    bool retry = false;
    uint32_t p = 0;     // consumed dense bits
    uint32_t y = 0, i = 1;
    int corner = 0, first = 0, run_index = 0;
    int a = 0, b = 0, c = 0, dd = 0, dnn = 0; // Ra, Rb = prev[i], Rc = prev[i - 1], prev[i + 1], prev[i + 2]
    int q1 = 0, q2 = 0;                       // quantised prev[i + 1] - prev[i] and prev[i] - prev[i - 1]
    uint32_t a_seen = 0;                      // OR of every updated A (samples wider than 8 bits): 2^24 overflow test
    const int maxval = t.maxval, reset = t.reset;
    const int limit_m = t.limit - t.qbpp - 1;

    auto quantised = [&](int diff) -> int {
        if (kWide)
            diff = diff < -cap ? -cap : (diff > cap ? cap : diff);
        return (int)lut[diff + cap];
    };
    // (re)loads the window of the previous line for sample i; c (= prev[i - 1]) is the caller's
    auto prime = [&]() {
        b = (int)line[i];
        dd = (int)line[i + 1];
        dnn = (int)line[i + 2];
        q2 = quantised(b - c);
        q1 = quantised(dd - b);
    };

    for (;;)
    {
        // ---- producer: keep kMarginBits ahead of the consumer; scans that finished their samples look for the marker
        {
            const uint32_t ahead = src.produced - p;
            const bool busy = phase != kDone && !src.ended;
            const bool need = busy && (phase == kDrain ? ahead < 64u : ahead < kMarginBits);
            if (__any(need))
            {
                const bool want = busy && ahead <= kRingBits - G * 128u - 128u;
                refill<G>(src, ring, want, lane, sub);
                continue;
            }
        }
        // ---- first sample of a line: reference src/scan_codec.hpp:189-195, src/scan_decoder_impl.hpp:62-129
        {
            const bool starting = phase == kLineStart;
            if (__any(starting))
            {
                if (starting && sub == 0)
                    line[width + 1] = line[width];
                JLS_LOCKSTEP();
                if (starting)
                {
                    c = corner;          // prev[0]
                    i = 1;
                    prime();
                    a = b;               // cur[0] = prev[1]
                    first = b;
                    phase = kInLine;
                }
            }
        }
        // ---- step loop: one regular-mode sample per scan and step
        int event = kNone;
        for (int step = 0; step < kStepsPerCheck; ++step)
        {
            const bool active = phase == kInLine && i <= width;
            const uint32_t win = peek32(ring, p);
            const int q1n = quantised(dnn - dd);
            const int q3 = quantised(c - a);
            const int qs = 81 * q1 + 9 * q2 + q3;
            const int s = qs >> 31;
            const int idx = (qs ^ s) - s;
            const Record rec = records[idx]; // idx 0 (run mode) reads a valid, unused record
            const int ctx_a = (int)rec.a;
            const int n = (int)(rec.ncb & 0xFFu);
            const int cc = (int)(signed char)(rec.ncb >> 8);
            const int bb = (int)rec.ncb >> 16;
            int k = __clz(n) - __clz(ctx_a); // N >= 1; A may be 0 (then k = 0)
            k = k < 0 ? 0 : k;
            k += ((n << k) < ctx_a);
            const int u = win == 0 ? 32 : __clz((int)win);
            const bool fits = u < limit_m && u + 1 + k <= 32 && (!kWide || k < 16);
            const bool ok = active && qs != 0 && fits;
            // Golomb code -> mapped error -> Errval (src/scan_decoder_core.hpp:38-69)
            const uint32_t rest = (win << u) << 1;
            const int mm = (u << k) | (int)(uint32_t)(((uint64_t)rest << k) >> 32);
            int e = (mm >> 1) ^ -(mm & 1);
            e ^= ((k - 1) & (2 * bb + n - 1)) >> 31; // k = 0 and 2B + N - 1 < 0: src/regular_mode_context.hpp:36-42
            // MED predictor = median(Ra, Rb, Ra + Rb - Rc), src/jpegls_algorithm.hpp:143-161, plus the bias C
            const int px = med3(med3(a + (b - c), a, b) + ((cc ^ s) - s), 0, maxval);
            const int x = (px + ((e ^ s) - s)) & maxval;
            // A.12 / A.13, src/regular_mode_context.hpp:45-93 (|B| cannot overflow in lossless mode).  With N' the new N
            // and tb = B + Errval (halved at a reset): delta = (tb > 0) - (tb + N' <= 0),
            // B' = median(tb - delta * N', 1 - N', 0), C' = median(C + delta, -128, 127).
            const int a_new = ctx_a + (e < 0 ? -e : e);
            const int sh = n == reset;
            const int n_new = (n >> sh) + 1;
            const int tb = (bb + e) >> sh;
            const int minus_delta = 1 - med3(tb, 0, 1) - med3(tb + n_new, 0, 1);
            const int b_new = med3(tb + minus_delta * n_new, 1 - n_new, 0);
            const int c_new = med3(cc - minus_delta, -128, 127);
            JLS_LOCKSTEP();
            if (ok)
            {
                records[idx] = Record{(uint32_t)(a_new >> sh),
                                      (uint32_t)n_new | (((uint32_t)c_new & 0xFFu) << 8) | ((uint32_t)b_new << 16)};
                line[i] = (S)x;
                if (kWide)
                    a_seen |= (uint32_t)a_new;
                p += (uint32_t)(u + 1 + k);
                a = x;
                c = b;
                b = dd;
                dd = dnn;
                dnn = (int)line[i + 3];
                q2 = q1;
                q1 = q1n;
                ++i;
            }
            else if (active)
                event = qs == 0 ? kRun : kSlow;
            JLS_LOCKSTEP();
            if (__any(event != kNone || (phase == kInLine && i > width)))
                break;
        }

        // ---- run mode: reference src/scan_decoder_impl.hpp:264-337, src/scan_decoder_core.hpp:72-100
        if (__any(event == kRun))
        {
            const bool in_run = event == kRun;
            const uint32_t remaining = width - (i - 1);
            uint32_t run = 0;
            bool counting = in_run;
            while (__any(counting))
            {
                const uint32_t bit = peek32(ring, p) >> 31;
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
                        line[i + r] = (S)a;
                    r += G;
                }
            }
            const uint32_t at = i + run;
            JLS_LOCKSTEP();
            const int b_at = (int)line[interrupted ? at : 0]; // prev[at]: not overwritten yet
            const int which = a == b_at ? 1 : 0;
            RunCtx ctx = run_ctx[which];
            int x = 0;
            if (interrupted)
            {
                const int k = run_k(ctx);
                const int limit = t.limit - run_j(run_index) - 1;
                const int u = k > 24 ? -1 : take_unary(ring, p, 47); // anything longer: let the exact decoder classify it
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
                    x = which ? ((a + e) & t.maxval) : ((b_at + e * ((b_at - a) < 0 ? -1 : 1)) & t.maxval);
                }
            }
            JLS_LOCKSTEP();
            if (interrupted)
            {
                run_ctx[which] = ctx;
                line[at] = (S)x;
                a = x;
                c = b_at; // becomes Rc of the next sample
                if (run_index > 0)
                    --run_index;
                i = at + 1;
            }
            else if (in_run && !retry)
                i = width + 1; // the run reached the end of the line
            JLS_LOCKSTEP();
            if (interrupted && i <= width)
                prime();
        }

        // ---- one regular-mode sample with every case the step loop leaves out (escape codes, long prefixes)
        if (__any(event == kSlow))
        {
            const bool slow = event == kSlow;
            const int qs = 81 * q1 + 9 * q2 + quantised(c - a);
            const int s = qs >> 31;
            const int idx = (qs ^ s) - s;
            const Record rec = records[idx];
            RegCtx ctx{(int)rec.a, (int)rec.ncb >> 16, (int)(signed char)(rec.ncb >> 8), (int)(rec.ncb & 0xFFu)};
            const int k = regular_k(ctx);
            const int px = clamp_sample(t, med_predict(a, b, c) + ((ctx.c ^ s) - s));
            int x = 0;
            bool good = slow && k < 16;
            if (good)
            {
                const int u = take_unary(ring, p, 47);
                if (u < 0)
                    good = false;
                else
                {
                    int mm;
                    if (u < limit_m)
                        mm = (u << k) | (int)take_bits(ring, p, k);
                    else
                        mm = (int)take_bits(ring, p, t.qbpp) + 1;
                    int e = unmap_error(mm);
                    if (k == 0)
                        e ^= error_correction(ctx, 0);
                    if (!regular_update(ctx, e, 0, t.reset))
                        good = false;
                    x = (px + ((e ^ s) - s)) & t.maxval;
                }
            }
            JLS_LOCKSTEP();
            if (good)
            {
                records[idx] = Record{(uint32_t)ctx.a,
                                      (uint32_t)ctx.n | (((uint32_t)ctx.c & 0xFFu) << 8) | ((uint32_t)ctx.b << 16)};
                line[i] = (S)x;
                a = x;
                c = b;
                ++i;
            }
            else if (slow)
                retry = true;
            JLS_LOCKSTEP();
            if (good && i <= width)
                prime();
        }

        if (kWide && a_seen >= (1u << 24))
            retry = true;
        if (retry)
            phase = kDone;

        // ---- finished line -> user's row
        {
            const bool ending = phase == kInLine && i > width;
            if (__any(ending))
            {
                uint8_t* row = d.pixels + (size_t)y * d.pixel_stride;
                const S* samples = line + 1;
                const uint32_t row_bytes = width * (uint32_t)sizeof(S);
                const bool aligned = ((reinterpret_cast<uintptr_t>(d.pixels) | d.pixel_stride) & 15u) == 0;
                const uint32_t wide_bytes = aligned ? row_bytes & ~15u : 0u;
                uint32_t off = (uint32_t)sub * 16u;
                while (__any(ending && off < wide_bytes))
                {
                    if (ending && off < wide_bytes)
                        *reinterpret_cast<uint4*>(row + off) =
                            *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(samples) + off);
                    off += G * 16u;
                }
                uint32_t xx = wide_bytes / (uint32_t)sizeof(S) + (uint32_t)sub;
                while (__any(ending && xx < width))
                {
                    if (ending && xx < width)
                        reinterpret_cast<S*>(row)[xx] = samples[xx];
                    xx += G;
                }
                JLS_LOCKSTEP();
                if (ending)
                {
                    corner = first;
                    ++y;
                    phase = y == d.height ? kDrain : kLineStart;
                }
            }
        }
        // ---- all samples decoded: the scan is clean when only zero padding is left and the marker follows
        {
            const bool draining = phase == kDrain;
            const uint32_t ahead = src.produced - p;
            if (draining && (src.ended || ahead >= 64u))
                phase = kDone;
        }
        if (__all(phase == kDone))
            break;
    }

    // Clean end of scan: nothing consumed past the coded segment, only zero padding left (at most the rest of a byte
    // plus the 7-bit byte that follows a trailing 0xFF), marker found right behind it.
    ScanResult r{kOk, 0, 0};
    if (!retry)
    {
        const uint32_t left = src.produced - p; // > 2^31 when the consumer ran past the producer
        const bool clean = src.u_marker != ~0ull && left < 15u && (left == 0 || (peek32(ring, p) >> (32 - left)) == 0);
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
