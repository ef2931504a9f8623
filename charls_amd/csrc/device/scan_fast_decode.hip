// scan_fast_decode.hip -- speed path of the scan decoder for lossless single-component scans (the BASELINE workload).
//
// Same one-wavefront-per-scan, wave-uniform organisation as scan_wave_decode.hip, with the per-sample dependency chain
// cut down to what is truly serial (bit position -> k -> context -> reconstructed sample):
//
//   * un-stuffing is hoisted out of the chain.  JPEG-LS stuffing is byte aligned in the coded stream (the byte after a
//     0xFF carries 7 payload bits), so each 1 KB refill is un-stuffed by all 64 lanes at once (16 bytes per lane, a
//     wave prefix sum of the 7/8-bit contributions, LDS atomic OR into a dense bit ring); marker detection (0xFF followed
//     by a byte >= 0x80) happens there as well.  The serial reader then only does aligned 64-bit reads of dense bits.
//   * the part of the context that depends on the previous LINE only (81*Q1 + 9*Q2 and the sample Rd) is computed by
//     all lanes for 64 samples at a time, straight from the line buffer (samples at and after the decoding position
//     still hold the previous line), and kept in a vector register; the serial loop reads it with v_readlane and
//     needs one gradient quantisation (Rc - Ra) and the context table per sample.  LDS per scan is the context
//     table, the bit ring and ONE line: 9 KB for 4096 8-bit samples, i.e. four wavefronts per SIMD.
//   * context records are two words (A | N<<24, B | C<<16); the quantised gradient Rc - Ra of 8-bit samples comes from a
//     511-entry LDS table; run mode and every unusual code are kept out of line.
//
// This kernel is not a restatement of the reference's bit reader; it decodes the same bit sequence.  Its result is used
// only when the scan ends cleanly (all samples decoded inside the entropy-coded segment, zero padding, marker next).
// In every other case it reports flags = kFastRetry and the host re-runs the exact decoder (scan_wave_decode.hip /
// scan_serial.hip), which reproduces the reference's error codes and byte counts.
#pragma once
#include <hip/hip_runtime.h>

#include "scan_model.h"
#include "scan_wave_decode.hip"

namespace jls {
namespace fast {

constexpr uint32_t kBitRingBytes = 2048;          // dense (un-stuffed) bits resident in LDS (>= 8256 + 4096 bits, see refill)
constexpr uint32_t kBitRingBits = kBitRingBytes * 8;
constexpr uint32_t kSrcChunk = 1024;              // coded bytes consumed per cooperative refill
constexpr uint32_t kFixedLds = wave::kCtxBytes + wave::kRunBytes + kBitRingBytes;
constexpr uint32_t kGradientLutBytes = 512;       // 8-bit samples only: quantised gradient for d = -255..255
template <typename S>
constexpr uint32_t fixed_lds()
{
    return kFixedLds + (sizeof(S) == 1 ? kGradientLutBytes : 0u);
}
constexpr uint32_t kFastRetry = 4u;               // ScanResult.flags: decode again with the exact kernel

// Producer/consumer state of the dense bit ring (wave-uniform).
struct DenseBits
{
    const uint8_t* gbase; // 16-byte aligned origin of the coded stream
    uint64_t u_next;      // next coded byte to un-stuff (u = offset + misalignment)
    uint64_t u_end;       // end of the source
    uint64_t u_begin;
    uint64_t produced;    // dense bits written so far
    uint64_t u_marker;    // position of the terminating marker once seen (else ~0)
    uint32_t prev_byte;   // last coded byte of the previous refill
    uint32_t* ring;       // kBitRingBytes / 4 words; bit p lives in word ((p >> 5) ^ 1) (64-bit words, MSB first)
    bool ended;           // marker or end of source reached: `produced` is final
    int lane;

    // OR `n` (1..32) bits, right aligned in v, at dense bit position p (MSB first).
    JLS_DEV void put(uint64_t p, uint32_t v, int n)
    {
        const uint32_t q = (uint32_t)((p & (kBitRingBits - 1)) >> 5);
        const int off = (int)(p & 31);
        const int room = 32 - off;
        if (n <= room)
            atomicOr(&ring[q ^ 1u], v << (room - n));
        else
        {
            atomicOr(&ring[q ^ 1u], v >> (n - room));
            const uint32_t q2 = (q + 1) & (kBitRingBits / 32 - 1);
            atomicOr(&ring[q2 ^ 1u], v << (32 - (n - room)));
        }
    }

    // Un-stuffs up to kSrcChunk coded bytes.  All 64 lanes.
    JLS_DEV void refill()
    {
        // 1) clear the words this refill may touch (everything after the word holding `produced`)
        {
            const uint32_t first = (uint32_t)((produced + 31) >> 5);
            for (uint32_t i = lane; i < kSrcChunk * 8 / 32 + 2; i += 64)
            {
                const uint32_t q = (first + i) & (kBitRingBits / 32 - 1);
                ring[q ^ 1u] = 0;
            }
        }
        __syncthreads();
        // 2) every lane takes 16 coded bytes
        const uint64_t u0 = u_next + (uint64_t)lane * 16;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (u0 < u_end)
            raw = *reinterpret_cast<const uint4*>(gbase + u0);
        const uint32_t words[4] = {raw.x, raw.y, raw.z, raw.w};
        uint32_t next_first = 0; // coded byte following this lane's 16 (for the marker test of its last byte)
        if (u0 + 16 < u_end)
            next_first = gbase[u0 + 16];
        uint32_t last = words[3] >> 24;
        uint32_t before = __shfl_up(last, 1);
        if (lane == 0)
            before = prev_byte;
        // bits contributed by each byte and the first marker inside this lane's bytes
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
            if (u == u_begin)
                prev = 0; // the first coded byte has no predecessor
            if (u >= u_begin && u < u_end && marker_at == 16)
            {
                const uint32_t nb = j < 15 ? ((words[(j + 1) >> 2] >> (((j + 1) & 3) * 8)) & 0xFFu) : next_first;
                const bool is_marker = b == 0xFFu && (u + 1 >= u_end || (nb & 0x80u) != 0);
                if (is_marker)
                    marker_at = j;
                else
                    n = prev == 0xFFu ? 7 : 8;
            }
            nbits[j] = n;
            total += n;
            prev = b;
        }
        // lanes after the first marker (or past the end) contribute nothing
        const unsigned long long has_marker = __ballot(marker_at < 16);
        const int first_marker_lane = has_marker ? __ffsll(has_marker) - 1 : 64;
        if (lane > first_marker_lane)
            total = 0;
        // exclusive prefix sum of the lanes' bit counts
        int inc = total;
        for (int delta = 1; delta < 64; delta <<= 1)
        {
            const int up = __shfl_up(inc, delta);
            if (lane >= delta)
                inc += up;
        }
        uint64_t p = produced + (uint64_t)(inc - total);
        if (lane <= first_marker_lane)
        {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (nbits[j] != 0)
                {
                    put(p, nbits[j] == 7 ? (bytes[j] & 0x7Fu) : bytes[j], nbits[j]);
                    p += (uint64_t)nbits[j];
                }
        }
        const int chunk_bits = (int)uniform((uint32_t)__shfl(inc, 63));
        produced += (uint64_t)chunk_bits;
        prev_byte = uniform(__shfl(last, 63));
        if (has_marker)
        {
            const int at = (int)uniform((uint32_t)__shfl(marker_at, first_marker_lane));
            u_marker = u_next + (uint64_t)first_marker_lane * 16 + (uint64_t)at;
            ended = true;
        }
        u_next += kSrcChunk;
        if (u_next >= u_end)
            ended = true;
        __syncthreads();
    }

    JLS_DEV void init(const uint8_t* stream, uint64_t size, uint32_t* ring_, int lane_)
    {
        const uint64_t mis = (uint64_t)(reinterpret_cast<uintptr_t>(stream) & 15u);
        gbase = stream - mis;
        u_begin = mis;
        u_next = 0;
        u_end = mis + size;
        produced = 0;
        u_marker = ~0ull;
        prev_byte = 0;
        ring = ring_;
        ended = size == 0;
        lane = lane_;
        for (uint32_t i = lane; i < kBitRingBits / 32; i += 64)
            ring[i] = 0;
        __syncthreads();
    }

    // 64 dense bits starting at bit p, MSB first (zeros beyond `produced`).
    JLS_DEV uint64_t peek64(uint64_t p) const
    {
        const uint64_t* ring64 = reinterpret_cast<const uint64_t*>(ring);
        const uint32_t w = (uint32_t)((p & (kBitRingBits - 1)) >> 6);
        const uint64_t r0 = ring64[w];
        const uint64_t r1 = ring64[(w + 1) & (kBitRingBits / 64 - 1)];
        const uint64_t w0 = ((uint64_t)uniform((uint32_t)(r0 >> 32)) << 32) | uniform((uint32_t)r0); // scalar registers
        const uint64_t w1 = ((uint64_t)uniform((uint32_t)(r1 >> 32)) << 32) | uniform((uint32_t)r1);
        const int s = (int)(p & 63);
        return s ? ((w0 << s) | (w1 >> (64 - s))) : w0;
    }
};

// Consumer side of the dense ring (wave-uniform, lives in scalar registers): the next `valid` bits MSB-aligned in
// `cache` (bits below are zero), topped up 32 bits at a time from ring word `next_word`.  `safe_words` counts the words
// the producer has completed beyond next_word; the scan loop keeps it >= kHandlerWords before it enters the out-of-line
// handlers (a run or a long code consumes far fewer bits), so these never have to call the producer themselves.
struct BitWindow
{
    uint64_t cache;
    int valid;
    uint32_t next_word;
    uint32_t safe_words;
    bool starved; // a handler ran past the producer: the scan is retried by the exact decoder
};

constexpr uint32_t kRingWords = kBitRingBits / 32;
constexpr uint32_t kHandlerWords = 144; // 64 samples x 2 words + a run / long code (ring: 512 words, refill <= 258)
constexpr uint32_t kUnlimitedWords = 0x40000000u;

JLS_DEV void top_up(BitWindow& w, const uint32_t* ring) // requires valid <= 32
{
    uint32_t word = 0;
    if (w.safe_words != 0)
    {
        word = uniform(ring[(w.next_word & (kRingWords - 1)) ^ 1u]);
        --w.safe_words;
    }
    else
        w.starved = true;
    ++w.next_word;
    w.cache |= (uint64_t)word << (32 - w.valid);
    w.valid += 32;
}

JLS_DEV void fill(BitWindow& w, const uint32_t* ring) // afterwards valid >= 33
{
    while (w.valid <= 32)
        top_up(w, ring);
}

JLS_DEV uint32_t take_bits(BitWindow& w, const uint32_t* ring, int n) // 0 <= n <= 32
{
    fill(w, ring);
    const uint32_t v = (uint32_t)((w.cache >> 1) >> (63 - n));
    w.cache <<= n;
    w.valid -= n;
    return v;
}

// Number of zero bits before the next one bit, which is consumed as well; -1 when it exceeds `most`.
JLS_DEV int take_unary(BitWindow& w, const uint32_t* ring, int most)
{
    int total = 0;
    for (;;)
    {
        fill(w, ring);
        const int u = w.cache == 0 ? 64 : __clzll((long long)w.cache);
        if (u < w.valid)
        {
            w.cache <<= u + 1;
            w.valid -= u + 1;
            total += u;
            return total > most ? -1 : total;
        }
        total += w.valid;
        w.cache = 0;
        w.valid = 0;
        if (total > most)
            return -1;
    }
}

// Per-sample record derived from the previous line: low half = 9*Q1 + Q2 (|.| <= 40, signed), high half = prev[i+1].
// 8-bit samples pack it in 16 bits, wider samples in 32.
template <typename S>
struct AuxOf
{
    static constexpr int kShift = sizeof(S) == 1 ? 8 : 16;
};

template <typename S>
JLS_DEV int aux_rd(uint32_t a)
{
    return (int)(a >> AuxOf<S>::kShift);
}
template <typename S>
JLS_DEV int aux_pre(uint32_t a)
{
    return AuxOf<S>::kShift == 8 ? (int)(signed char)(a & 0xFFu) : (int)(short)(a & 0xFFFFu);
}

// Record of position `pos` (1-based).  line[p] holds the previous line for p >= i (the decoding position); the sample
// left of position i has already been overwritten, its previous-line value is `rc_at_i`.  line[width + 1] replicates
// line[width].  Lanes left of i or right of the line produce records nobody reads.
template <typename S>
JLS_DEV uint32_t chunk_record(const Traits& t, const S* line, uint32_t pos, uint32_t i, uint32_t width, int rc_at_i)
{
    const uint32_t p = pos <= width ? pos : width;
    const int rb = (int)line[p];
    const int rd = (int)line[p + 1];
    const int rc = pos == i ? rc_at_i : (int)line[p - 1];
    const int pre = 9 * quantize(t, rd - rb) + quantize(t, rb - rc);
    return ((uint32_t)pre & ((1u << AuxOf<S>::kShift) - 1u)) | ((uint32_t)rd << AuxOf<S>::kShift);
}

// Regular-mode context record of this kernel: word 0 = A | N << 24, word 1 = (B & 0xFFFF) | C << 16.  A < 2^24 is
// checked at every update, N <= RESET <= 255 (the reference stores RESET through a uint8_t, src/scan_codec.hpp:142),
// -N < B <= 0 and -128 <= C <= 127 after A.13, so the record is exact and packs/unpacks in one or two instructions.
struct CtxRecord
{
    uint32_t an;
    uint32_t bc;
};

JLS_DEV RegCtx open_record(const CtxRecord r)
{
    return RegCtx{(int)(r.an & 0xFFFFFFu), (int)(short)(r.bc & 0xFFFFu), (int)r.bc >> 16, (int)(r.an >> 24)};
}

JLS_DEV CtxRecord close_record(const RegCtx& x)
{
    return CtxRecord{(uint32_t)x.a | ((uint32_t)x.n << 24), ((uint32_t)x.b & 0xFFFFu) | ((uint32_t)x.c << 16)};
}

// Run mode (reference src/scan_decoder_impl.hpp:270-330, src/scan_decoder_core.hpp:71-101).  Returns false when the scan
// must be retried by the exact decoder.
template <typename S>
JLS_DEV bool decode_run(const Traits& t, const wave::WaveModel& m, BitWindow& w, const uint32_t* ring, S* line,
                        uint32_t width, uint32_t& i, int& ra, int& rb, int& run_index, int lane)
{
    const uint32_t remaining = width - (i - 1);
    uint32_t run = 0;
    for (;;)
    {
        if (!take_bits(w, ring, 1))
            break;
        const uint32_t block = 1u << run_j(run_index);
        const uint32_t count = block < remaining - run ? block : remaining - run;
        run += count;
        if (count == block && run_index < 31)
            ++run_index;
        if (run == remaining)
            break;
    }
    if (run != remaining)
    {
        run += take_bits(w, ring, run_j(run_index));
        if (run > remaining)
            return false;
    }
    JLS_LOCKSTEP();
    for (uint32_t r = lane; r < run; r += 64)
        line[i + r] = (S)ra;
    if (run == remaining)
    {
        i = width + 1;
        return true;
    }
    const uint32_t at = i + run;
    JLS_LOCKSTEP();
    const int rb_at = (int)uniform((uint32_t)line[at]); // prev[at]: not overwritten yet
    const int which = ra == rb_at ? 1 : 0;
    JLS_LOCKSTEP();
    RunCtx ctx = m.run[which];
    ctx.ritype = (int)uniform((uint32_t)ctx.ritype);
    ctx.a = (int)uniform((uint32_t)ctx.a);
    ctx.n = (int)uniform((uint32_t)ctx.n);
    ctx.nn = (int)uniform((uint32_t)ctx.nn);
    const int k = run_k(ctx);
    if (k > 24)
        return false;
    const int limit = t.limit - run_j(run_index) - 1;
    const int u = take_unary(w, ring, 47); // anything longer: let the exact decoder classify it
    if (u < 0)
        return false;
    int em;
    if (u < limit - t.qbpp - 1)
        em = (u << k) + (int)take_bits(w, ring, k);
    else
        em = (int)take_bits(w, ring, t.qbpp) + 1;
    const int e = run_error_value(ctx, em + ctx.ritype, k);
    run_update(ctx, e, em, t.reset);
    JLS_LOCKSTEP();
    m.run[which] = ctx;
    const int rx = which ? ((ra + e) & t.maxval) : ((rb_at + e * ((rb_at - ra) < 0 ? -1 : 1)) & t.maxval);
    line[at] = (S)rx;
    ra = rx;
    rb = rb_at; // becomes Rc of the next sample
    if (run_index > 0)
        --run_index;
    i = at + 1;
    return true;
}

// One regular-mode sample with every case the inner loop leaves out (escape codes, prefixes longer than the window):
// src/scan_decoder_core.hpp:38-69, src/scan_decoder.hpp:99-125.  Returns false -> retry with the exact decoder.
JLS_DEV bool decode_regular_slow(const Traits& t, const wave::WaveModel& m, BitWindow& w, const uint32_t* ring, int qs,
                                 int pred, int& x_out)
{
    const int s = qs >> 31;
    const int idx = (qs ^ s) - s;
    JLS_LOCKSTEP();
    CtxRecord* records = reinterpret_cast<CtxRecord*>(m.reg);
    const CtxRecord packed = records[idx];
    RegCtx ctx = open_record(CtxRecord{uniform(packed.an), uniform(packed.bc)});
    const int k = regular_k(ctx);
    if (k >= 16)
        return false;
    const int px = clamp_sample(t, pred + ((ctx.c ^ s) - s));
    const int u = take_unary(w, ring, 47);
    if (u < 0)
        return false;
    int mm;
    if (u < t.limit - t.qbpp - 1)
        mm = (u << k) | (int)take_bits(w, ring, k);
    else
        mm = (int)take_bits(w, ring, t.qbpp) + 1;
    int e = unmap_error(mm);
    if (k == 0)
        e ^= error_correction(ctx, 0);
    if (!regular_update(ctx, e, 0, t.reset))
        return false;
    JLS_LOCKSTEP();
    records[idx] = close_record(ctx);
    x_out = (px + ((e ^ s) - s)) & t.maxval;
    return true;
}

} // namespace fast

// Dynamic LDS: fast::fixed_lds<S>() + (width + 2) * sizeof(S) rounded up to 4.
//
// Control structure: ONE loop whose body visits a 64-sample chunk of the current line.  The chunk's records (aux) are
// computed into a VGPR once and read per sample with v_readlane; decoded samples are collected in a VGPR and
// written back to the line in one LDS store, so the per-sample LDS traffic is the context record only.  The inner loop
// handles nothing but plain regular-mode samples whose code fits the bit window; everything else leaves it with an event
// code and is handled once, out of line: producer refill (the only call site of DenseBits::refill), run mode, long or
// escape codes.  That keeps the inner loop short, straight and free of cold code.
template <typename S>
__global__ void __launch_bounds__(64) decode_scans_fast(const ScanDesc* __restrict__ descs, ScanResult* __restrict__ results)
{
    using namespace fast;
    JLS_DYNAMIC_LDS(smem);
    const int lane = threadIdx.x;
    const ScanDesc d = descs[blockIdx.x];
    const Traits t = make_traits(d);
    const wave::WaveModel m{reinterpret_cast<wave::PackedCtx*>(smem), reinterpret_cast<RunCtx*>(smem + wave::kCtxBytes)};
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + wave::kCtxBytes + wave::kRunBytes);
    const uint32_t width = d.width;
    signed char* gradient_lut = reinterpret_cast<signed char*>(smem + kFixedLds); // 8-bit samples only
    S* line = reinterpret_cast<S*>(smem + fixed_lds<S>());
    CtxRecord* records = reinterpret_cast<CtxRecord*>(m.reg);

    {
        const CtxRecord fresh = close_record(RegCtx{initial_a(t), 0, 0, 1});
        for (int q = lane; q < 365; q += 64)
            records[q] = fresh;
        if (lane < 2)
            m.run[lane] = RunCtx{lane, initial_a(t), 1, 0};
        if (sizeof(S) == 1)
            for (int q = lane; q < 511; q += 64)
                gradient_lut[q] = (signed char)quantize(t, q - 255);
    }
    for (uint32_t i = lane; i < width + 2; i += 64)
        line[i] = 0;
    DenseBits src;
    src.init(d.stream, d.stream_capacity, ring, lane);
    BitWindow w{0, 0, 0, 0, false};

    enum : int { kNone = 0, kChunkEnd, kRun, kSlow, kRetry };
    enum : int { kLineStart = 0, kInLine, kDrain };
    int phase = d.height == 0 ? kDrain : kLineStart;
    int corner = 0, first = 0;
    int run_index = 0;
    bool retry = false;
    const int t1 = t.t1, t2 = t.t2, t3 = t.t3, maxval = t.maxval, reset = t.reset;
    const int limit_m = t.limit - t.qbpp - 1;
    uint32_t y = 0, i = 1;
    int ra = 0, rb = 0, rd = 0;
    const int vz = vector_zero();
    const int v_zero = vz, v_one = vz | 1, v_maxval = vz | maxval, v_cmin = vz | -128, v_cmax = vz | 127; // VGPR constants

    for (;;)
    {
        // ---- producer: the one place where coded bytes are un-stuffed into the ring
        if (!src.ended && (phase == kDrain || w.safe_words < kHandlerWords))
        {
            src.refill();
            w.safe_words = src.ended ? kUnlimitedWords : (uint32_t)(src.produced >> 5) - w.next_word;
            continue;
        }
        if (src.ended)
            w.safe_words = kUnlimitedWords;
        if (phase == kDrain)
            break;
        if (phase == kLineStart)
        {
            if (lane == 0)
                line[width + 1] = line[width];
            __syncthreads();
            rb = corner;                                  // prev[0]
            ra = (int)uniform((uint32_t)line[1]);         // cur[0] = prev[1]
            rd = ra;                                      // prev[1]
            first = ra;
            i = 1;
            phase = kInLine;
        }

        // ---- one visit of the chunk that holds sample i
        const uint32_t chunk_base = (i - 1) & ~63u;
        const uint32_t chunk_last = chunk_base + 64 < width ? chunk_base + 64 : width;
        const uint32_t pos = chunk_base + 1 + lane;
        const uint32_t v_aux = chunk_record<S>(t, line, pos, i, width, rb);
        uint32_t v_out = 0;
        const uint32_t flush_from = i;
        int event = kNone;
        int qs = 0;
        // Work split of the inner loop.  One CU has ONE scalar unit for its four SIMDs but a vector ALU per SIMD, and a
        // loop that is all scalar instructions is bound by that shared unit (measured: two wavefronts per SIMD were barely
        // faster than one).  So only the bit window and the loop control are kept on the scalar unit; the sample
        // arithmetic (gradient, context record, prediction, A/B/C/N update) is done by the vector ALU on values that
        // are equal in all lanes -- Ra and the context record simply stay in vector registers -- and k is the one value
        // that crosses back per sample.  Wavefronts sharing a SIMD then overlap scalar and vector work.
        int ra_v = ra | vz;     // equal in all lanes, kept in a VGPR
        uint32_t a_seen = 0;    // OR of every updated A: the 2^24 overflow test is done once per chunk visit
        // Single-exit loop (exits are taken at the bottom only: the compiler adds guard flags to multi-exit loops).  The
        // producer margin (kHandlerWords) covers the 2 words a sample can consume times the 64 samples of a visit.
        do
        {
            if (w.valid <= 32)
                top_up(w, ring);
            const int sel = (int)((i - 1) & 63u);
            const uint32_t a = from_lane(v_aux, sel);
            const int rd_next = aux_rd<S>(a);
            // Q3 = quantised (Rc - Ra) (reference src/jpegls_algorithm.hpp:173-194): a table look-up for 8-bit samples
            int q3;
            if (sizeof(S) == 1)
                q3 = (int)gradient_lut[(rb + 255) - ra_v];
            else
            {
                const int d3 = rb - ra_v;
                const int ad = d3 < 0 ? -d3 : d3;
                q3 = (ad > 0) + (ad >= t1) + (ad >= t2) + (ad >= t3);
                q3 = d3 < 0 ? -q3 : q3;
            }
            const int qs_v = 9 * aux_pre<S>(a) + q3;
            const int s = qs_v >> 31;
            const int idx = qs_v < 0 ? -qs_v : qs_v;
            JLS_LOCKSTEP();
            const CtxRecord packed = records[idx]; // idx 0 (run mode) reads a valid, unused slot
            if (uniform((uint32_t)qs_v) == 0)
            {
                qs = 0;
                event = kRun;
                continue;
            }
            const int a_acc = (int)(packed.an & 0xFFFFFFu);
            const int n = (int)(packed.an >> 24);
            const int b = (int)(short)(packed.bc & 0xFFFFu);
            const int c = (int)packed.bc >> 16;
            int k_v = __builtin_clz((unsigned)n) - __clz(a_acc); // N >= 1; A may be 0 (then k = 0)
            k_v = k_v < 0 ? 0 : k_v;
            k_v += ((n << k_v) < a_acc);
            const int k = (int)uniform((uint32_t)k_v);
            const int u = w.cache == 0 ? 64 : __clzll((long long)w.cache);
            // 8-bit samples: |Errval| <= 128 keeps A / N < 2^9, so k < 16 and A < 2^24 hold by construction there
            if (u >= limit_m || u + 1 + k > w.valid || (sizeof(S) > 1 && k >= 16))
            {
                qs = (int)uniform((uint32_t)qs_v);
                event = kSlow;
                continue;
            }
            // MED predictor = median of (Ra, Rb, Ra + Rb - Rc): src/jpegls_algorithm.hpp:143-161
            const int px = med3(med3s(ra_v + (rd - rb), ra_v, rd) + ((c ^ s) - s), v_zero, v_maxval);
            const uint64_t after = w.cache << (u + 1);
            const int mm = (u << k) | (int)((after >> 1) >> (63 - k));
            w.cache = after << k;
            w.valid -= u + 1 + k;
            int e = (mm >> 1) ^ -(mm & 1);
            e ^= ((k_v - 1) & (2 * b + n - 1)) >> 31; // k = 0 and 2B + N - 1 < 0: src/regular_mode_context.hpp:36-42
            // A.12/A.13, src/regular_mode_context.hpp:45-93 (|B| cannot overflow in lossless mode).  With N' the new N
            // and t = B + Errval (halved at a reset): delta = (t > 0) - (t + N' <= 0), B' = median(t - delta * N', 1 - N', 0),
            // C' = median(C + delta, -128, 127).
            const int a_new = a_acc + (e < 0 ? -e : e);
            if (sizeof(S) > 1)
                a_seen |= (uint32_t)a_new;
            const int sh = n == reset;
            const int n_new = (n >> sh) + 1;
            const int tb = (b + e) >> sh;
            const int minus_delta = 1 - med3(tb, v_zero, v_one) - med3(tb + n_new, v_zero, v_one);
            const int b_new = med3(tb + __mul24(minus_delta, n_new), 1 - n_new, v_zero);
            const int c_new = med3(c - minus_delta, v_cmin, v_cmax);
            JLS_LOCKSTEP();
            records[idx] = CtxRecord{(uint32_t)(a_new >> sh) | ((uint32_t)n_new << 24),
                                     ((uint32_t)b_new & 0xFFFFu) | ((uint32_t)c_new << 16)};
            const int x = (px + ((e ^ s) - s)) & maxval;
            v_out = lane == sel ? (uint32_t)x : v_out;
            rb = rd;
            rd = rd_next;
            ra_v = x;
            ++i;
            event = (int)((chunk_last - i) >> 31); // kChunkEnd (= 1) once i has passed the chunk, else kNone
        } while (event == kNone);
        ra = (int)uniform((uint32_t)ra_v);
        if (uniform(a_seen) >= (1u << 24))
            event = kRetry;
        // samples decoded by the inner loop -> line
        if (pos >= flush_from && pos < i)
            line[pos] = (S)v_out;
        JLS_LOCKSTEP();
        if (event == kRun)
        {
            // the inner loop had not advanced its neighbourhood yet: Rc = rb, Rb = rd as for any sample at i
            int rb_next = rd;
            rb = rd;
            if (!decode_run<S>(t, m, w, ring, line, width, i, ra, rb_next, run_index, lane))
                retry = true;
            rb = rb_next;
            JLS_LOCKSTEP();
            if (i <= width)
                rd = (int)uniform((uint32_t)line[i]); // prev[i]: Rb of the next sample
        }
        else if (event == kSlow)
        {
            int x = 0;
            if (!decode_regular_slow(t, m, w, ring, qs, med_predict(ra, rd, rb), x))
                retry = true;
            JLS_LOCKSTEP();
            const int rd_after = (int)uniform((uint32_t)line[i + 1]); // prev[i + 1]
            JLS_LOCKSTEP();
            if (lane == 0)
                line[i] = (S)x;
            rb = rd;
            rd = rd_after;
            ra = x;
            ++i;
        }
        else if (event == kRetry)
            retry = true;
        if (retry || w.starved)
        {
            retry = true;
            break;
        }
        if (i > width)
        { // finished line -> user's row
            corner = first;
            __syncthreads();
            uint8_t* row = d.pixels + (size_t)y * d.pixel_stride;
            if (sizeof(S) == 1)
                for (uint32_t x = lane; x < width; x += 64)
                    row[x] = (uint8_t)line[1 + x];
            else
                for (uint32_t x = lane; x < width; x += 64)
                    reinterpret_cast<uint16_t*>(row)[x] = (uint16_t)line[1 + x];
            JLS_LOCKSTEP();
            ++y;
            phase = y == d.height ? kDrain : kLineStart;
        }
    }

    // Clean end of scan: nothing consumed past the coded segment, only zero padding left (at most the rest of a byte
    // plus the 7-bit byte that follows a trailing 0xFF), marker found right behind it.
    ScanResult r{kOk, 0, 0};
    if (!retry)
    {
        const uint64_t consumed = (uint64_t)w.next_word * 32 - (uint64_t)w.valid;
        const bool inside = consumed <= src.produced;
        const uint64_t left = inside ? src.produced - consumed : 0;
        const bool clean = inside && src.u_marker != ~0ull && left < 15 &&
                           (left == 0 || (src.peek64(consumed) >> (64 - left)) == 0);
        if (clean)
            r.bytes = src.u_marker - src.u_begin;
        else
            retry = true;
    }
    if (retry)
        r.flags = kFastRetry;
    if (lane == 0)
        results[blockIdx.x] = r;
}

} // namespace jls
