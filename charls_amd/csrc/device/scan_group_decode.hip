// scan_group_decode.hip -- speed path of the scan decoder for single-component and line-interleaved scans, lossless and
// near-lossless: SEVERAL scans per wavefront.
//
// Decoding one scan is one dependency chain (bit position -> k -> context -> reconstructed sample, reference
// src/scan_decoder_core.hpp:38-69), so the only parallelism is the number of scans in flight, and what a wavefront pays
// per decoded sample is the number of instructions it has to issue (one wavefront issues one instruction every four to five
// cycles: tools/microbench/issue_ceiling.hip).  scan_fast_decode.hip spends a whole 64-lane instruction on the value of ONE
// scan; here the 64 lanes are split into groups of G lanes and every group decodes a scan of its own, so one instruction
// advances 64 / G scans:
//
//   * all per-scan state (Ra, bit position, pointers into the line and into the prepared entries, producer state) lives in
//     vector registers, replicated over the G lanes of the group; nothing is wave-uniform, there is no scalar chain;
//   * control flow stays convergent for the whole wavefront: every step all groups decode one regular-mode sample; what is
//     rare -- a context whose N has reached RESET, run mode, an unusual code -- is ONE mask looked at once per step, the
//     frequent run events are served inside the loop under the EXEC mask of their lanes (scan_group_step.inc), and for the rest
//     (an escape or long code, a long run, the end of a line, an empty bit ring) the wavefront leaves the step loop and the event
//     is handled once, out of line, under the predicate of the groups that raised it;
//   * the G lanes of a group share the bulk work of their scan: un-stuffing 16 coded bytes per lane into the dense bit ring
//     (a lane's 16 bytes as one 128-bit number: refill), preparing what the steps need of the previous line 64 samples ahead
//     (prepare), run fills, and the 16-byte row stores of every finished line;
//   * LDS per scan: 365 context records (8 B), two run contexts, a 1 KB dense bit ring, 192 prepared entries (8 B) and ONE line
//     of samples = 9.7 KB for 4096 8-bit samples (the launch rule: runtime.hip, decode_group_plan).
//
// Like scan_fast_decode.hip this is not a restatement of the reference's bit reader: a result is accepted only when the
// scan ends cleanly (all samples decoded inside the entropy-coded segment, zero padding, marker next); everything else
// reports flags = kFastRetry and the exact decoder decides (runtime.hip: launch_decode_plain).
#pragma once
#include <hip/hip_runtime.h>

#include "scan_model.h"
#include "scan_fast_decode.hip"
#ifndef JLS_EMULATED
#include "scan_group_step.inc"
#endif

namespace jls {
namespace grp {

constexpr uint32_t kRingWords = 256;               // dense bits resident per scan: 8192 (words kRingWords, kRingWords + 1 mirror words 0, 1)
constexpr uint32_t kRingBits = kRingWords * 32;
constexpr int kStepsPerCheck = 64;                 // regular-mode steps between two looks at the producer
constexpr uint32_t kMarginBits = kStepsPerCheck * 32 + 320; // dense bits the step loop and one event handler may consume
constexpr int kMaxTableT3 = 1023;                  // widest gradient table (2 * T3 + 1 entries) for samples wider than 8 bits
// What a step needs of the previous line is prepared ahead of the step loop, 64 samples at a time and by all lanes of the group
// (prepare_line): one 8-byte entry per sample, in a ring of 128 entries whose first 64 are mirrored behind its end, so that a
// step loop of up to 63 steps reads its entries at consecutive addresses wherever in the ring it starts.
constexpr uint32_t kPrepRing = 128;
constexpr uint32_t kPrepSlots = kPrepRing + 64;
constexpr uint32_t kPrepChunk = 64;                // samples prepared per call
constexpr int kStepsPerLoop = 63;                  // the step loop never consumes the last prepared sample (see prepare_line)

// LDS of a workgroup (one wavefront): the gradient table shared by its scans, then one region per scan.  A scan's line
// starts one sample before a 16-byte boundary so that sample 1 (the first of the row) is aligned for the 16-byte row stores.
template <typename S>
struct Layout
{
    static constexpr uint32_t kLutBytes = sizeof(S) == 1 ? 512 : 2 * kMaxTableT3 + 2; // quantised gradient + 4 for -cap .. cap
    static constexpr uint32_t kRecords = 0;                        // 365 x 8 B (+ an unused slot)
    static constexpr uint32_t kRun = 2928;                         // 2 x RunCtx, then what the step loop's run service needs of its call (16 B)
    static constexpr uint32_t kRing = kRun + 48;                   // kRingWords + 2 words
    static constexpr uint32_t kPrep = kRing + kRingWords * 4 + 8;  // kPrepSlots x 8 B: what a step needs of the previous line
    static constexpr uint32_t kLine = kPrep + kPrepSlots * 8 + 24 + 16 - sizeof(S);
    static_assert(kPrep % 8 == 0 && (kLine + sizeof(S)) % 16 == 0, "alignment of the prepared entries and of sample 1 of the line");
};

// Regions follow each other at a stride of 16 bytes more than a multiple of 128: the scans of a wavefront run in step and
// address their regions at equal offsets, and a stride that is a multiple of the 32 banks x 4 bytes (or of half of it)
// puts all of them, or every other one, on the same banks -- every LDS instruction of the step loop then takes its
// bank-conflict cycles, which is what wavefronts sharing a CU queue for.
__host__ __device__ constexpr uint32_t bank_spread(uint32_t bytes)
{
    return ((bytes + 127u) & ~127u) + 16u;
}

// Bytes from one line of a scan to the next (line-interleaved scans keep one line per component).
template <typename S>
__host__ __device__ constexpr uint32_t line_stride_bytes(uint32_t width)
{
    return ((width + 6) * (uint32_t)sizeof(S) + 15u) & ~15u;
}

template <typename S>
__host__ __device__ constexpr uint32_t region_bytes(uint32_t width, uint32_t lines = 1)
{
    return bank_spread(Layout<S>::kLine + lines * line_stride_bytes<S>(width));
}

template <typename S>
__host__ __device__ constexpr uint32_t workgroup_lds_bytes(uint32_t width, uint32_t scans_per_wave, uint32_t lines = 1)
{
    return Layout<S>::kLutBytes + scans_per_wave * region_bytes<S>(width, lines);
}

// Regular-mode context record: word 0 = A, word 1 = N | (C & 0xFF) << 8 | B << 16.  N <= RESET <= 255, -128 <= C <= 127
// and -N < B <= 0 after A.13 (reference src/regular_mode_context.hpp:45-93), so the record is exact.
struct Record
{
    uint32_t a;
    uint32_t ncb;
};

// One prepared entry of the previous line (the kernel's `prepare`) / one 8-byte LDS word.
struct PrepEntry
{
    uint32_t x, y;
};
JLS_DEV PrepEntry load_pair(uint32_t address) // (8-byte aligned)
{
    const uint64_t w = lds_load<uint64_t>(address);
    return PrepEntry{(uint32_t)w, (uint32_t)(w >> 32)};
}
JLS_DEV void store_pair(uint32_t address, const PrepEntry& e)
{
    lds_store<uint64_t>(address, ((uint64_t)e.y << 32) | e.x);
}

#ifndef JLS_EMULATED
JLS_DEV uint32_t abs_difference_plus(uint32_t a, uint32_t b, uint32_t c) // |a - b| + c in one instruction
{
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#else
JLS_DEV uint32_t abs_difference_plus(uint32_t a, uint32_t b, uint32_t c)
{
    return (a > b ? a - b : b - a) + c;
}
#endif

// The dense bit ring is LSB first: stream bit j is bit (j & 31) of word (j >> 5), so that the next 32 bits of the stream are
// one funnel shift (v_alignbit_b32) of two neighbouring words, the unary prefix is a count of TRAILING zeros, and a
// k-bit field of the stream (whose first bit is the most significant) is the bit-reversed low end of the window.
JLS_DEV uint32_t peek32(const uint32_t* ring, uint32_t p)
{
    const uint32_t wi = (p >> 5) & (kRingWords - 1);
    const uint64_t both = ((uint64_t)ring[wi + 1] << 32) | ring[wi];
    return (uint32_t)(both >> (p & 31));
}

// The two ring words around bit p, the ring given by its LDS address held in one register (two address instructions).
JLS_DEV uint64_t ring_words_at(uint32_t ring_address, uint32_t p)
{
    const uint32_t at = ring_address + (bit_field(p, 5, 8) << 2); // word (p >> 5) mod kRingWords
    return ((uint64_t)lds_load<uint32_t>(at + 4) << 32) | lds_load<uint32_t>(at);
}

// Value of the first n bits of window w, first bit most significant (0 <= n <= 32).
JLS_DEV uint32_t field(uint32_t w, int n)
{
    return (uint32_t)(((uint64_t)bit_reverse(w) << n) >> 32);
}

JLS_DEV uint32_t take_bits(const uint32_t* ring, uint32_t& p, int n) // 0 <= n <= 32
{
    const uint32_t w = peek32(ring, p);
    p += (uint32_t)n;
    return field(w, n);
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

// OR `n` (1..8) bits, right aligned in v with the first bit of the stream most significant, at dense bit position p.
JLS_DEV void put_bits(uint32_t* ring, uint32_t p, uint32_t v, int n)
{
    const uint32_t q = (p >> 5) & (kRingWords - 1);
    const int off = (int)(p & 31);
    const uint32_t rv = bit_reverse(v) >> (32 - n); // first bit of the stream at bit 0
    atomicOr(&ring[q], rv << off);
    if (q < 2)
        atomicOr(&ring[kRingWords + q], rv << off);
    if (off + n > 32)
    {
        const uint32_t q2 = (q + 1) & (kRingWords - 1);
        atomicOr(&ring[q2], rv >> (32 - off));
        if (q2 < 2)
            atomicOr(&ring[kRingWords + q2], rv >> (32 - off));
    }
}

// Un-stuffs G x 16 coded bytes of every scan whose lanes pass `want` (JPEG-LS stuffing is byte aligned in the coded
// stream: the byte after a 0xFF carries 7 payload bits; a 0xFF followed by a byte >= 0x80 is a marker).  Called by all
// 64 lanes; the shuffles only ever read lanes of the caller's own group.
//
// Two renderings.  Inside a scan -- every wanted lane's 16 bytes and the byte behind them are coded bytes that have a
// predecessor, and no 0xFF is followed by a byte >= 0x80 -- a lane's bytes are ONE 128-bit number in the ring's bit order
// from which the stuffed bits are deleted (one byte in 256 is followed by one; the wavefront loops as often as its busiest
// lane has them, once or twice) and which then goes to the ring as five words.  The first bytes of a misaligned stream,
// the bytes around its end and anything that looks like a marker take the byte-by-byte rendering (refill_bytewise).
template <int G>
JLS_DEV void refill_bytewise(Producer& s, uint32_t* ring, bool want, int lane, int sub, uint64_t u0, const uint4& raw,
                             uint32_t next_first, uint32_t before)
{
    constexpr uint32_t kChunk = G * 16;
    const uint32_t words[4] = {raw.x, raw.y, raw.z, raw.w};
    const uint32_t last = words[3] >> 24;
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

// Bits 7, 15, 23, 31: the bytes of w that are 0xFF (exact: no carries between the bytes).
JLS_DEV uint32_t ff_bytes(uint32_t w)
{
    const uint32_t v = ~w;
    return ~((((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) | 0x7F7F7F7Fu);
}

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
            if (q < 2)
                ring[kRingWords + q] = 0;
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
    const uint32_t last = raw.w >> 24;
    uint32_t before = __shfl_up(last, 1);
    if (sub == 0)
        before = s.prev_byte; // (0 ahead of the first coded byte)
    // 0xFF bytes, and bytes with their first bit set, as bits 7, 15, 23, 31 of their words
    const uint32_t z0 = ff_bytes(raw.x), z1 = ff_bytes(raw.y), z2 = ff_bytes(raw.z), z3 = ff_bytes(raw.w);
    const uint32_t marker_like = (((z0 << 8) & raw.x) | (((z1 << 8) | (z0 >> 24)) & raw.y) | (((z2 << 8) | (z1 >> 24)) & raw.z) |
                                  (((z3 << 8) | (z2 >> 24)) & raw.w) | ((z3 >> 24) & next_first)) & 0x80808080u;
    const bool inside = u0 >= s.u_begin && u0 + 17 <= s.u_end;
    if (!__all(!want || (inside && marker_like == 0)))
    {
        JLS_PATH(11); // refills byte by byte
        refill_bytewise<G>(s, ring, want, lane, sub, u0, raw, next_first, before);
        return;
    }
    // 3) the lane's bytes in the ring's bit order (the first bit of the stream is bit 0)
    uint32_t d0 = __builtin_bswap32(bit_reverse(raw.x)), d1 = __builtin_bswap32(bit_reverse(raw.y));
    uint32_t d2 = __builtin_bswap32(bit_reverse(raw.z)), d3 = __builtin_bswap32(bit_reverse(raw.w));
    // the stuffed bits: the first bit (bit 8 t of the number) of every byte t that follows a 0xFF
    uint32_t f0 = (((z0 << 8) | (before == 0xFFu ? 0x80u : 0u))) >> 7, f1 = ((z1 << 8) | (z0 >> 24)) >> 7;
    uint32_t f2 = ((z2 << 8) | (z1 >> 24)) >> 7, f3 = ((z3 << 8) | (z2 >> 24)) >> 7;
    const int total = want ? 128 - (__popc(f0) + __popc(f1) + __popc(f2) + __popc(f3)) : 0;
    if (!want)
        f0 = f1 = f2 = f3 = 0;
    while (__any((f0 | f1 | f2 | f3) != 0))
    { // delete the last of them: what lies above it moves down one bit
        JLS_PATH(12); // trips of the refill's delete loop
        if ((f0 | f1 | f2 | f3) != 0)
        {
            const int at = f3 ? 3 : (f2 ? 2 : (f1 ? 1 : 0));
            const uint32_t fw = f3 ? f3 : (f2 ? f2 : (f1 ? f1 : f0));
            const uint32_t bit = 31u - leading_zeros(fw);
            const uint32_t below = (1u << bit) - 1u;
            const uint32_t e0 = (d0 >> 1) | (d1 << 31), e1 = (d1 >> 1) | (d2 << 31), e2 = (d2 >> 1) | (d3 << 31), e3 = d3 >> 1;
            d0 = at == 0 ? (d0 & below) | (e0 & ~below) : d0;
            d1 = at == 1 ? (d1 & below) | (e1 & ~below) : (at < 1 ? e1 : d1);
            d2 = at == 2 ? (d2 & below) | (e2 & ~below) : (at < 2 ? e2 : d2);
            d3 = at == 3 ? (d3 & below) | (e3 & ~below) : e3;
            f0 = at == 0 ? f0 & below : f0;
            f1 = at == 1 ? f1 & below : f1;
            f2 = at == 2 ? f2 & below : f2;
            f3 = at == 3 ? f3 & below : f3;
        }
    }
    // 4) exclusive prefix sum of the group's bit counts, then the number goes to the ring as five words
    int inc = total;
    for (int delta = 1; delta < G; delta <<= 1)
    {
        const int up = __shfl_up(inc, delta);
        if (sub >= delta)
            inc += up;
    }
    const int group_base = lane - sub;
    if (want)
    {
        const uint32_t p = s.produced + (uint32_t)(inc - total);
        const uint32_t o = p & 31u;
        const uint32_t q = p >> 5;
        const uint32_t c[5] = {d0 << o, (uint32_t)((((uint64_t)d1 << 32) | d0) << o >> 32), (uint32_t)((((uint64_t)d2 << 32) | d1) << o >> 32),
                               (uint32_t)((((uint64_t)d3 << 32) | d2) << o >> 32), (uint32_t)(((uint64_t)d3 << o) >> 32)};
#pragma unroll
        for (int j = 0; j < 5; ++j)
        {
            const uint32_t qj = (q + (uint32_t)j) & (kRingWords - 1);
            atomicOr(&ring[qj], c[j]);
            if (qj < 2)
                atomicOr(&ring[kRingWords + qj], c[j]);
        }
    }
    const int chunk_bits = __shfl(inc, group_base + G - 1);
    const uint32_t chunk_last = __shfl(last, group_base + G - 1);
    JLS_LOCKSTEP();
    if (want)
    {
        s.produced += (uint32_t)chunk_bits;
        s.prev_byte = chunk_last;
        s.u_next += kChunk;
        if (s.u_next >= s.u_end)
            s.ended = true;
    }
}

// Samples of the first r blocks of a run-length code, sum of 2^J[q] for q < r (0 <= r <= 32; J: reference src/scan_codec.hpp:18-19
// = {0 x 4, 1 x 4, 2 x 4, 3 x 4, 4, 4, 5, 5, 6, 6, 7, 7, 8, 9, ..., 15}), in closed form: 0 .. 16 by fours, 16 .. 24 by twos, then
// single blocks.
JLS_DEV uint32_t run_prefix(uint32_t r)
{
    const uint32_t fours = ((4u + (r & 3u)) << (r >> 2)) - 4u;                          // r <= 16
    const uint32_t twos = 28u + ((2u + (r & 1u)) << (4u + (((r - 16u) >> 1) & 7u)));    // 16 <= r <= 24
    const uint32_t ones = 284u + (1u << ((r - 16u) & 31u));                             // 24 <= r <= 32
    return r <= 16u ? fours : (r <= 24u ? twos : ones);
}

// Number of zero bits before the next one bit, which is consumed as well; -1 when it exceeds `most`.  Per lane.
JLS_DEV int take_unary(const uint32_t* ring, uint32_t& p, int most)
{
    int total = 0;
    for (;;)
    {
        const uint32_t w = peek32(ring, p);
        const int u = w == 0 ? 32 : __ffs((int)w) - 1;
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

// Dynamic LDS: grp::workgroup_lds_bytes<S>(width, scans per workgroup, NL).  `count` scans, 64 / G of them per wavefront.
//
// The step loop is written for its instruction count (scan_group_step.inc has the budget and what was measured): what a step
// needs of the previous line comes prepared ({Rc | T << 16, Rb} per sample), sign handling is multiply-adds with +-1, and what
// an event needs is worked out after the loop from the state it leaves behind.
//
// NL = 1: a single-component scan.  NL = 2..4: a LINE-INTERLEAVED scan of NL components (reference
// src/scan_decoder_impl.hpp:62-129): the lines of a pixel row are coded one component after the other, each against the
// line of its own component above it and with its own RUNindex, on the ONE set of contexts; a line is decoded exactly like
// a line of a single-component scan, so the step loop is the same code, and what changes is which of the NL lines in LDS
// it works on and that a finished pixel row goes to the user's row interleaved (and through the inverse colour transform).
//
// W: wavefronts of a workgroup.  The wavefronts of a kernel never talk to each other (one barrier, behind the shared gradient
// table); W only decides WHERE they run.  A workgroup's wavefronts are dealt to the SIMDs of its CU in turn, so W = 4 with the
// whole LDS of a CU puts exactly one wavefront on every SIMD of every CU.  Workgroups of ONE wavefront (rounds 2 - 4) are
// placed wherever a slot is free: with four of them per CU, two often share a SIMD while another SIMD idles, and those two
// take the launch's tail with them -- rocprofv3 counted wavefronts resident for 77 % of such a launch on average while the
// shader engines were busy for 94 % of it, at an unchanged 2.39 GHz (profiles/r05_pmc_decode_effective_clock.txt; what
// rounds 3 and 4 read as "the chip clocks down when every SIMD runs this kernel").
//
// kNear: near-lossless scans (NEAR > 0; src/default_traits.hpp).  The chain is the same -- the line holds RECONSTRUCTED samples,
// which is what the decoder works on anyway -- and what differs is local: gradients within +-NEAR quantise to 0 (the table), no
// error correction, Rx = Px + sign Errval (2 NEAR + 1) brought back into range and clamped instead of reduced modulo 2^bpp, B
// moves by Errval (2 NEAR + 1), and a run is interrupted into the context of |Ra - Rb| <= NEAR.
template <typename S, int G, int NL = 1, int W = 1, bool kNear = false>
__global__ void __launch_bounds__(64 * W) decode_scans_group(const ScanDesc* __restrict__ descs, ScanResult* __restrict__ results,
                                                             uint32_t count)
{
    using namespace grp;
    using L = Layout<S>;
    static_assert(G == 4 || G == 8 || G == 16 || G == 32, "lanes per scan");
    static_assert(NL >= 1 && NL <= 4, "lines per pixel row");
    static_assert(W == 1 || W == 2 || W == 4 || W == 8, "wavefronts per workgroup");
    constexpr int kScansPerWave = 64 / G;
    constexpr bool kWide = sizeof(S) > 1;
    constexpr bool kChecked = kWide || kNear; // codes may exceed the 32-bit window, |Errval| 65535, A 2^24: looked at
    JLS_DYNAMIC_LDS(smem);
    const int lane = W == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 63u);
    const int wave = W == 1 ? 0 : (int)(threadIdx.x >> 6);
    const int sid = lane / G;
    const int sub = lane % G;
    const uint32_t scan = (blockIdx.x * (uint32_t)W + (uint32_t)wave) * kScansPerWave + (uint32_t)sid;
    const bool live = scan < count;
    const ScanDesc d = descs[live ? scan : count - 1];
    const Traits t = make_traits(d);
    const uint32_t width = d.width;

    unsigned char* region = smem + L::kLutBytes + (size_t)(wave * kScansPerWave + sid) * region_bytes<S>(width, NL);
    Record* records = reinterpret_cast<Record*>(region + L::kRecords);
    RunCtx* run_ctx = reinterpret_cast<RunCtx*>(region + L::kRun);
    uint32_t* ring = reinterpret_cast<uint32_t*>(region + L::kRing);
    const uint32_t ring_address = opaque(lds_address(ring));
    // The gradient table (quantised gradient + 4 for gradients -cap .. cap; beyond that the magnitude is 4) is shared by
    // the scans of the wavefront: its index is then a difference plus a constant, with no per-scan base to add.  It is
    // built from the thresholds of the workgroup's first scan; a scan with other thresholds (batches mix them only when
    // the streams carry different LSE segments) is left to the exact decoder.
    unsigned char* lut = smem;
    const ScanDesc& d_first = descs[blockIdx.x * (uint32_t)(W * kScansPerWave)]; // (always < count: every workgroup has a live first scan)
    const Traits t_first = make_traits(d_first);
    S* const line0 = reinterpret_cast<S*>(region + L::kLine);
    const uint32_t line_stride = line_stride_bytes<S>(width) / (uint32_t)sizeof(S); // samples from one component's line to the next
    S* line = line0; // the line being decoded
    const int cap = kWide ? t_first.t3 : 255;
    const bool own_table = t.t1 == t_first.t1 && t.t2 == t_first.t2 && t.t3 == t_first.t3 && t.bpp == t_first.bpp && t.near == t_first.near &&
                           (t.near != 0) == kNear;

    {
        const Record fresh{(uint32_t)initial_a(t), 1u};
        // (record 0 belongs to no context -- Q = 0 is run mode -- and is what a step in run mode reads: its N is preset to RESET,
        // so that the step loop's ONE test for "anything rare" -- N has reached RESET -- catches run mode as well)
        for (int q = sub; q < 366; q += G)
            records[q] = q == 0 ? Record{fresh.a, (uint32_t)t.reset} : fresh;
        if (sub < 2)
            run_ctx[sub] = RunCtx{sub, initial_a(t), 1, 0};
        // (the host checks the thresholds of ONE scan of the launch: a scan with T3 beyond the table that leads its wavefront
        // must not write past the table's region -- it and its neighbours go to the exact decoder, see `usable`)
        for (int q = (int)threadIdx.x; q <= 2 * cap && q < (int)L::kLutBytes; q += 64 * W)
            lut[q] = (unsigned char)((quantize(t_first, q - cap) + 4) * 8); // (premultiplied: a context record has 8 bytes)
        for (uint32_t q = sub; q < (NL == 1 ? width + 6 : NL * line_stride); q += G)
            line0[q] = 0;
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
    if (W > 1)
        __syncthreads(); // the one meeting of the workgroup's wavefronts: the gradient table they share is complete
    JLS_LOCKSTEP();

    enum : int { kLineStart = 0, kInLine, kDrain, kDone };
    const bool usable = own_table && (!kWide || t_first.t3 <= kMaxTableT3) && lds_address(smem) == 0; // (see lds_load)
    int phase = !live || !usable ? kDone : (d.height == 0 ? kDrain : kLineStart);
    bool retry = live && !usable;
    uint32_t p = 0;     // consumed dense bits
    uint32_t y = 0, i = 1;
    int corner = 0, first = 0, run_index = 0;
    // line-interleaved scans: the component whose line is being decoded, and what every component keeps from row to row
    int comp = 0;
    int corner_of[NL], run_index_of[NL];
#pragma unroll
    for (int c = 0; c < NL; ++c)
        corner_of[c] = run_index_of[c] = 0;
    int a = 0;           // Ra
    uint32_t a_seen = 0; // OR of every updated A (samples wider than 8 bits): 2^24 overflow test
    const int maxval = t.maxval, reset = t.reset;
    const uint32_t limit_m = (uint32_t)(t.limit - t.qbpp - 1);
    // the longest unary prefix the out-of-line readers follow: LIMIT - qbpp - 1 zeros announce the escape code -- 47 for lossless
    // 16-bit samples, more when NEAR makes qbpp small (LIMIT = 64) -- and anything longer is left to the exact decoder
    constexpr int kLongestPrefix = kNear ? 62 : 47;
    const int near = kNear ? t.near : 0, near_step = 2 * near + 1, range_step = t.range * near_step;
    // the sample a run is interrupted by (src/scan_decoder_impl.hpp:300-337): in the context of RItype 1 -- Ra and Rb within
    // NEAR of each other -- Ra + Errval, else Rb + Errval sign(Rb - Ra); reconstructed as every sample is
    auto same_level = [&](int ra, int rb) -> bool { return kNear ? abs_difference((uint32_t)ra, (uint32_t)rb) <= (uint32_t)near : ra == rb; };
    auto reconstructed = [&](int predicted, int e) -> int {
        if (!kNear)
            return (predicted + e) & maxval;
        int v = mad24(e, near_step, predicted);
        v += v < -near ? range_step : (v > maxval + near ? -range_step : 0);
        return med3(v, 0, maxval);
    };
    auto interruption_sample = [&](bool which, int ra, int rb, int e) -> int {
        return which ? reconstructed(ra, e) : reconstructed(rb, (rb - ra) < 0 ? -e : e);
    };
    const uint32_t records_address = lds_address(records);
    const uint32_t prep_address = lds_address(region + L::kPrep);
    // What the steps need of the previous line is prepared ahead of them (prepare): the entries of samples up to prepped_end
    // exist.  An entry is made while the previous line still holds ALL its samples around it, i.e. before sample j - 1 of
    // the current line is stored; the one exception is the first sample behind an event that jumped ahead of the prepared
    // entries (a line start, a long run): its Rc = prev[i - 1] is kept in rc_over by whoever moved i.
    uint32_t prepped_end = 0;
    int rc_over = 0;

    // (quantised gradient + 4) * 8, 0..64: the three of a context then combine without sign extensions, and the byte offset of
    // a context's record is |8 (81 Q1 + 9 Q2 + Q3)| = |9 (9 q1 + q2) + q3 - 8 * 364| with q = 8 (Q + 4)
    auto quantised = [&](int diff) -> int {
        if (kWide)
            diff = med3(diff, -cap, cap);
        return (int)lds_load<unsigned char>((uint32_t)(diff + cap)); // the table is at LDS address 0
    };
    // Entries of samples from .. from + kPrepChunk - 1 (those of the line): {Rc | T << 16, Rb} with T = 9 (9 q1 + q2), q1 of
    // Rd - Rb, q2 of Rb - Rc.  Every lane of the group takes kPrepChunk / G consecutive samples: their kPer + 2 samples of the
    // previous line, kPer + 1 gradients (q2 of a sample is q1 of its left neighbour).
    auto prepare = [&](bool want, uint32_t from, bool first_is_over) {
        constexpr uint32_t kPer = kPrepChunk / G;
        const uint32_t j0 = from + (uint32_t)sub * kPer;
        const uint32_t last = width + 1; // (prev[width + 1] = prev[width], set at the start of the line)
        int v[kPer + 2];
#pragma unroll
        for (uint32_t c = 0; c < kPer + 2; ++c)
        {
            const uint32_t at = j0 - 1 + c;
            v[c] = (int)line[at < last ? at : last];
        }
        if (first_is_over && sub == 0)
            v[0] = rc_over;
        int g[kPer + 1];
#pragma unroll
        for (uint32_t c = 0; c < kPer + 1; ++c)
            g[c] = quantised(v[c + 1] - v[c]);
        JLS_LOCKSTEP();
#pragma unroll
        for (uint32_t c = 0; c < kPer; ++c)
        {
            const uint32_t j = j0 + c;
            const uint32_t tt = (uint32_t)mad24(g[c + 1], 81, 9 * g[c]);
            const uint32_t slot = j & (kPrepRing - 1);
            const PrepEntry entry{(uint32_t)v[c] | (tt << 16), (uint32_t)v[c + 1]};
            if (want && j <= width)
            {
                store_pair(prep_address + slot * 8u, entry);
                if (slot < kPrepSlots - kPrepRing)
                    store_pair(prep_address + (slot + kPrepRing) * 8u, entry);
            }
        }
        JLS_LOCKSTEP();
    };

    for (;;)
    {
        JLS_PATH(0); // rounds
        // A wavefront whose scans are all inside a line with their bit rings filled needs nothing before its next steps but their
        // entries, and one whose scans all decoded to the end of the step loop nothing behind it: most rounds are such rounds (one
        // in 54 steps on the bench's frames, ~ 2000 cycles each), and they pay ONE test where the blocks below pay one each.
        // (Scans that are through -- or lane groups without a scan -- do not count, as long as one scan is still inside a line.)
        const bool look = !__all(phase == kDone || (phase == kInLine && (src.ended || src.produced - p >= kMarginBits)));
        // ---- producer: keep kMarginBits ahead of the consumer; scans that finished their samples look for the marker
        {
            const uint32_t ahead = src.produced - p;
            const bool busy = phase != kDone && !src.ended;
            const bool need = busy && (phase == kDrain ? ahead < 64u : ahead < kMarginBits);
            if (look && __any(need))
            {
                const bool want = busy && ahead <= kRingBits - G * 128u - 128u;
                JLS_PATH(1); // refills
                refill<G>(src, ring, want, lane, sub);
                continue;
            }
        }
        // ---- first sample of a line: reference src/scan_codec.hpp:189-195, src/scan_decoder_impl.hpp:62-129
        {
            const bool starting = phase == kLineStart;
            if (look && __any(starting))
            {
                JLS_PATH(2); // line starts
                if (NL > 1 && starting)
                {
                    line = line0 + (uint32_t)comp * line_stride;
#pragma unroll
                    for (int c = 0; c < NL; ++c)
                    {
                        corner = comp == c ? corner_of[c] : corner;
                        run_index = comp == c ? run_index_of[c] : run_index;
                    }
                }
                if (starting && sub == 0)
                    line[width + 1] = line[width];
                JLS_LOCKSTEP();
                if (starting)
                {
                    i = 1;
                    prepped_end = 0;
                    rc_over = corner;    // Rc = prev[0]
                    a = (int)line[1];    // cur[0] = prev[1]
                    first = a;
                    phase = kInLine;
                }
            }
        }
        // ---- the previous line, prepared for the next steps: every scan inside a line has the entries of samples i .. i + 63
        // (or to the end of its line) before the step loop; a call makes 64 of them
        {
            const uint32_t reach = i + (uint32_t)kStepsPerLoop < width ? i + (uint32_t)kStepsPerLoop : width;
            const bool need = phase == kInLine && prepped_end < reach;
            if (__any(need))
            {
                JLS_PATH(14); // calls of prepare
                const uint32_t from = prepped_end + 1 > i ? prepped_end + 1 : i;
                prepare(need, from, from == i);
                if (need)
                    prepped_end = from + kPrepChunk - 1;
            }
        }
        // ---- step loop: one regular-mode sample per scan and step
        const bool in_line = phase == kInLine; // i <= width: the end of a line is handled as soon as it is reached
        const LaneMask in_line_m = lanes_where(in_line);
        LaneMask ok_m = ~0ull; // lanes whose last step decoded a sample
        uint32_t qsu8 = 2912;  // 8 (Q + 364) of the last step
        // what the last step of a lane that could not decode its sample had in its registers (the run handler starts from it)
        uint32_t win_stopped = 0;
        PrepEntry entry_stopped{0, 0};
        if (in_line_m != 0)
        {
            // no scan may step past the end of its line: the wavefront takes as many steps as the shortest rest allows
            const uint32_t rest_of_line = width + 1 - i;
            uint32_t steps = kStepsPerLoop;
            for (;;)
            { // (one trip per scan that is close to the end of its line, at most)
                const LaneMask closer = lanes_where(in_line && rest_of_line < steps);
                if (closer == 0)
                    break;
                JLS_PATH(4); // trips of the step-count loop
                steps = value_of_lowest_lane(closer, rest_of_line);
            }
            JLS_PATH(3); // step loops
            // lanes outside their line never pass the `u < limit` test below
            const uint32_t limit_v = opaque(in_line ? limit_m : 0u);
            // The loop is rotated.  An iteration first does the bookkeeping the PREVIOUS step left behind -- bit position,
            // line and entry pointers, the second half of its context update (A.13, k and the error correction of the new
            // state) and the two stores -- and then decodes its own sample; the LDS reads of the new step are issued before
            // the update arithmetic and the context read before the prefix / predictor arithmetic, so the wavefront finds
            // every LDS result waiting.  Nothing in the loop is predicated: when a lane cannot decode its sample the loop
            // ends before that lane's next bookkeeping, and the lanes that did decode theirs get it after the loop.  The
            // first iteration does the bookkeeping of a step that changes nothing (it rewrites cur[i - 1] and an unused
            // context record); scans that are not inside a line (finished or waiting for their marker: their contexts and
            // line are dead) run along on their own dead state without advancing.
            const uint32_t p_kept = p;
            constexpr uint32_t kSz = (uint32_t)sizeof(S);
            const uint32_t line_address = lds_address(line);
            // lm: LDS address of the slot of the sample BEFORE the previous step's sample (a step adds one slot and then
            // stores Ra = the previous step's sample at lm); pp: entry of the previous step's sample
            uint32_t lm = in_line ? line_address + (i - 2) * kSz : line_address + (width + 1) * kSz;
            uint32_t pp = prep_address + ((i & (kPrepRing - 1)) << 3) - (in_line ? 8u : 0u);
            uint32_t where = records_address + 365u * 8u; // the previous step's context record (an unused slot at first)
            // what the previous step leaves for its context update: A + |Errval|, N, B + Errval (all three already halved
            // when N had reached RESET) and C
            int u_a = 0, u_n1 = 2, u_tb = 0, u_cc = 0; // u_n1 = N + 1
            uint32_t u1 = 0, k_last = 0, t_mm = 0;    // the previous step's code: prefix + 1, k, mapped error
            uint32_t k_seen = 0, mm_seen = 0, a_seen_now = 0;
            uint32_t win_now = 0;
#if defined(JLS_EMULATED) || defined(JLS_CXX_STEP_LOOP)
            // (the rendering the CPU harness runs, and the specification of the assembly below)
            const uint32_t lm_step = in_line ? kSz : 0u;
            const uint32_t pp_step = in_line ? 8u : 0u;
            uint64_t ticker = 1ull << (steps - 1); // one-hot step counter, 1 <= steps <= kStepsPerLoop
            const uint32_t pp_first = pp + 8u;     // the entry of the call's first sample
            do
            {
                JLS_PATH(5); // steps
                // -- bookkeeping of the previous step, part 1: registers (Ra was set when the sample was decoded)
                p += u1 + k_last;
                if (kChecked)
                    mm_seen |= t_mm;
                lm += lm_step; // now the slot of the previous step's sample; the slots behind it still hold the previous line
                pp += pp_step;
                // -- this step's LDS reads: the prepared entry {Rc | T << 16, Rb}, the bit window, Q3
                const PrepEntry entry = load_pair(pp);
                const uint64_t ring_words = ring_words_at(ring_address, p);
                const int rc = kWide ? (int)(entry.x & 0xFFFFu) : (int)(entry.x & 0xFFu);
                const int rb = (int)entry.y;
                const int q3 = quantised(rc - a);
                // -- bookkeeping, part 2: A.12 / A.13, src/regular_mode_context.hpp:45-93 (|B| cannot overflow in lossless
                // mode).  With N' the new N and tb = B + Errval (halved at a reset): delta = (tb > 0) - (tb + N' <= 0),
                // B' = median(tb - delta * N', 1 - N', 0), C' = median(C + delta, -128, 127).
                Record updated;
                {
                    const int n_new = u_n1;
                    const int minus_delta = 1 - med3(u_tb, 0, 1) - med3(u_tb + n_new, 0, 1);
                    const int b_new = med3(mad24(minus_delta, n_new, u_tb), 1 - n_new, 0);
                    const int c_new = med3(u_cc - minus_delta, -128, 127);
                    updated = Record{(uint32_t)u_a, ((uint32_t)b_new << 16) | pack_bytes((uint32_t)c_new, (uint32_t)n_new)};
                }
                const uint32_t win = (uint32_t)(ring_words >> (p & 31)); // the next 32 bits of the stream
                win_now = win;
                // -- the chain: Ra -> Q3 -> context -> k -> code -> Errval -> sample
                qsu8 = (entry.x >> 16) + (uint32_t)q3; // 8 (Q + 364)
                const uint32_t here = abs_difference_plus(qsu8, 2912u, records_address); // the record of |Q|; Q = 0 (run mode): an unused one
                const int sgn = qsu8 < 2912u ? -1 : 1;
                JLS_LOCKSTEP();
                store_pair(where, PrepEntry{updated.a, updated.ncb}); // bookkeeping, part 3: the stores
                lds_store<S>(lm, (S)a);
                JLS_LOCKSTEP();
                where = here;
                const PrepEntry rec = load_pair(here); // {A, N | C << 8 | B << 16}
                const uint32_t u = lowest_one(win); // length of the unary prefix; 0xFFFFFFFF for an all-zero window
                u1 = u + 1u; // the window beyond the prefix: u1 = 32 only with k = 0, where no field is read
                const uint32_t beyond = bit_reverse(win >> (u1 & 31u));
                const int px0 = med3(a + (rb - rc), a, rb); // MED predictor = median(Ra, Rb, Ra + Rb - Rc), src/jpegls_algorithm.hpp:143-161
                // A regular-mode sample whose code lies inside the 32-bit window.  8-bit samples: valid streams keep
                // A / N < 2^9, so k <= 9, and u < LIMIT - qbpp - 1 <= 23 then bounds the code by 32 bits and |Errval| by
                // 5888; k (and, for wider samples, the mapped error) are only accumulated here and examined after the
                // loop: a stream that breaks those bounds is invalid and goes to the exact decoder as a whole.
                ok_m = lanes_where(qsu8 != 2912u) & lanes_where(u < limit_v);
                const int n = (int)(rec.y & 0xFFu);
                const int cc = (int)(signed char)(rec.y >> 8);
                const int bb = (int)rec.y >> 16;
                // k = min{k : N << k >= A} (N >= 1; A may be 0, then k = 0).  Scaling a float by 2^k adds k << 23 to its bit
                // pattern and positive floats order like their bit patterns, so N * 2^k >= A <=> k << 23 >= bits(A) - bits(N):
                // k = max(0, ceil((bits(A) - bits(N)) / 2^23)).  A < 2^24 and N <= 255 convert exactly.
                const int k_raw = ((int)(float_bits(rec.x) - float_bits((uint32_t)n)) + 0x7FFFFF) >> 23;
                const int k = k_raw < 0 ? 0 : k_raw;
                if (kChecked)
                    ok_m &= lanes_where(u + (uint32_t)k < 32u);
                // Golomb code -> mapped error -> Errval (src/scan_decoder_core.hpp:38-69): prefix and remainder are ONE 64-bit
                // number {u : the window beyond the prefix}, and the mapped error its upper half after a shift by k
                const int mm = (int)(uint32_t)(((((uint64_t)u << 32) | beyond) << (k & 63)) >> 32);
                // Errval = unmap(mm), complemented when k = 0 and 2B + N - 1 < 0 (src/regular_mode_context.hpp:36-42):
                // e = (mm >> 1) ^ -odd and |e| = (mm >> 1) + odd with odd = the low bit of mm, flipped by the correction
                // (17 bits of it: a mapped error beyond 131071 is invalid data and examined after the loop; what a lane that cannot
                // decode its sample -- an empty window: u = 0xFFFFFFFF -- computes here must not look like an overflow of A)
                const int half = (int)(((uint32_t)mm >> 1) & 0x1FFFFu);
                const int odd = kNear ? (mm & 1) : ((mm ^ (int)((uint32_t)((k - 1) & (2 * bb + n - 1)) >> 31)) & 1);
                const int e = half ^ -odd;
                const int px = med3(mad24(cc, sgn, px0), 0, maxval); // plus the bias C
                int e_scaled = e; // what B moves by
                if (kNear)
                { // Rx = Px + sign Errval (2 NEAR + 1), brought back into -NEAR .. MAXVAL + NEAR, clamped (src/default_traits.hpp:172-184)
                    e_scaled = mad24(e, near_step, 0);
                    int v = mad24(e, sgn < 0 ? -near_step : near_step, px);
                    v += v < -near ? range_step : (v > maxval + near ? -range_step : 0);
                    a = med3(v, 0, maxval);
                }
                else
                    a = mad24(e, sgn, px) & maxval; // every lane: a lane that could not decode reloads its Ra after the loop
                k_last = (uint32_t)k;
                t_mm = (uint32_t)mm;
                // first half of A.12 / A.13 for this step (the rest is the next iteration's): A += |Errval|, B += Errval, and
                // the halving of A, B and N once N has reached RESET
                u_a = (int)rec.x + half + odd;
                if (kChecked)
                    a_seen_now |= (uint32_t)u_a;
                u_n1 = n + 1;
                u_tb = bb + e_scaled;
                u_cc = cc;
                if (n == reset)
                { // once per RESET samples of a context
                    u_a >>= 1;
                    u_n1 = (n >> 1) + 1;
                    u_tb >>= 1;
                }
                k_seen |= (uint32_t)k; // of every lane: a lane that cannot decode still looked at a real context
                // -- run mode inside the loop (the rare path of the assembly: scan_group_step.inc, JLS_STEP_RARE): when the lanes that
                // stopped are all in run mode, a run of at most G samples that is interrupted inside its line, with RUNindex <= 16 behind
                // its complete blocks and all its bits inside this step's window, is served here; its lane then looks like one that
                // decoded a regular sample at the end of the run.  The count of steps is a count of SAMPLES per scan -- the entries
                // prepared for the call reach 63 ahead and a line ends: a run that would take its scan beyond either is not served,
                // and the count shrinks to what the scan that is furthest ahead may still take.
                {
                    const LaneMask run_m = in_line_m & lanes_where(qsu8 == 2912u);
                    const LaneMask unusual_m = in_line_m & ~(lanes_where(u < limit_v) & (kChecked ? lanes_where(u + (uint32_t)k < 32u) : ~0ull));
                    if (run_m != 0 && unusual_m == 0)
                    {
                        const bool mine = lane_of(run_m);
                        const RunCtx ctx0 = run_ctx[0], ctx1 = run_ctx[1];
                        const int ra = (int)lds_load<S>(lm);
                        const uint32_t ones = lowest_one(~win); // complete blocks of the run-length code
                        const uint32_t ri2 = (uint32_t)run_index + ones;
                        bool fits = mine && ones <= 16u && ri2 <= 16u;
                        const uint32_t j = fits ? ri2 >> 2 : 0u; // J[RUNindex] = RUNindex >> 2 up to 16
                        const uint32_t blocks = fits ? ((4u + (ri2 & 3u)) << j) - ((4u + ((uint32_t)run_index & 3u)) << ((uint32_t)run_index >> 2)) : 0u;
                        const uint32_t run = blocks + field(fits ? win >> (ones + 1u) : 0u, (int)j);
                        const uint32_t used = fits ? ones + 1u + j : 0u; // bits of the run-length code (<= 21)
                        const uint32_t taken = ((pp - pp_first) >> 3) + run + 1u; // samples of this call with the event
                        const uint32_t budget = rest_of_line < (uint32_t)kStepsPerLoop ? rest_of_line : (uint32_t)kStepsPerLoop;
                        fits = fits && run <= (uint32_t)G && taken <= budget;
                        const int rb = (int)lds_load<S>(lm + ((fits ? run : 0u) + 1u) * kSz); // the sample above the interruption sample
                        const int which = same_level(ra, rb) ? 1 : 0;
                        RunCtx ctx = which ? ctx1 : ctx0;
                        const uint32_t goal = (uint32_t)ctx.a + (uint32_t)(ctx.n >> 1) * (uint32_t)ctx.ritype;
                        int kr = 0;
                        if (goal > (uint32_t)ctx.n)
                        {
                            kr = (int)leading_zeros((uint32_t)ctx.n) - (int)leading_zeros(goal);
                            kr += (((uint64_t)(uint32_t)ctx.n << kr) < goal) ? 1 : 0;
                        }
                        const uint32_t w3 = win >> used; // the bits behind the run-length code
                        const uint32_t zeros = lowest_one(w3);
                        const int escape_from = t.limit - (int)j - 1 - t.qbpp - 1;
                        const int tail_bits = (int)zeros < escape_from ? kr : t.qbpp;
                        const uint32_t event_bits = used + zeros + 1u + (uint32_t)tail_bits;
                        fits = fits && kr <= 24 && zeros < 32u && event_bits <= 32u;
                        const LaneMask served_m = lanes_where(fits);
                        if (served_m != 0)
                        {
                            JLS_PATH(16); // run services inside the step loop
                            const uint32_t tail = field(fits ? w3 >> ((zeros + 1u) & 31u) : 0u, fits ? tail_bits : 0);
                            const int em = (int)zeros < escape_from ? ((int)zeros << kr) + (int)tail : (int)tail + 1;
                            const int er = run_error_value(ctx, em + ctx.ritype, kr);
                            run_update(ctx, er, em, t.reset);
                            const int x = interruption_sample(which != 0, ra, rb, er);
                            JLS_LOCKSTEP();
                            if (fits)
                            {
                                if ((uint32_t)sub < run)
                                    lds_store<S>(lm + ((uint32_t)sub + 1u) * kSz, (S)ra); // the lanes of the scan fill its run
                                run_ctx[which] = ctx;
                                a = x;
                                u1 = event_bits;
                                k_last = 0;
                                t_mm = 0;
                                where = records_address + 365u * 8u; // (its "context update" goes to the unused record)
                                run_index = ri2 > 0 ? (int)ri2 - 1 : 0;
                                lm += run * kSz;
                                pp += run * 8u;
                            }
                            JLS_LOCKSTEP();
                            ok_m |= served_m;
                            // the count of steps left: no more than any served scan may still take
                            const uint32_t may_take = fits ? budget - taken : ~0u;
                            uint32_t left = 63u - (uint32_t)__builtin_clzll(ticker);
                            for (LaneMask m = served_m; m != 0;)
                            {
                                const uint32_t v = value_of_lowest_lane(m, may_take);
                                left = v < left ? v : left;
                                m &= ~lanes_where(may_take == v);
                            }
                            ticker = 1ull << left;
                        }
                    }
                }
                // one exit: a one-hot counter that an event clears
                ticker = tick(ticker, in_line_m, ok_m);
            } while (ticker != 0);
#else
            // The same loop, written out for gfx950: 72 instructions per step for 8-bit samples (the compiler's rendering of the
            // C++ above: 88), scheduled by hand so that the three LDS round trips of the chain (Q3 <- the gradient table,
            // the context record, and before them the prepared entry, which is requested one step ahead) are covered by the
            // previous step's context update, the bit window and the predictor.  Lanes outside their line are switched off
            // (EXEC) for the whole loop.  Hazards kept by hand: two instructions between a v_cmp and the VALU that reads its mask.
            {
                LaneMask fail_m;
                uint32_t count = steps - 1;
                const int cap_v = cap;
                const uint32_t run_ctx_address = lds_address(run_ctx);
                // (an event's escape prefix: escape_base - J; RESET <= 255: the records keep N in a byte)
                const uint32_t cfg_v = (uint32_t)(t.limit - t.qbpp - 2) | ((uint32_t)t.qbpp << 8) | ((uint32_t)reset << 16);
                const int maxval_s = (int)uniform((uint32_t)maxval); // (the scans of a wavefront that are inside a line share their sample precision: `usable`)
                // what the loop's run service needs of this call: the entry of the call's first sample and the samples left in
                // its line (the count of steps is a count of samples per scan: scan_group_step.inc)
                lds_store<uint64_t>(run_ctx_address + 32u, ((uint64_t)rest_of_line << 32) | (uint64_t)(pp + 8u));
                [[maybe_unused]] const int near_range_s = (int)uniform((uint32_t)near | ((uint32_t)range_step << 8));
                if constexpr (kNear && kWide)
                    JLS_STEP_LOOP_ASM_WIDE_NEAR();
                else if constexpr (kNear)
                    JLS_STEP_LOOP_ASM_NARROW_NEAR();
                else if constexpr (kWide)
                    JLS_STEP_LOOP_ASM_WIDE();
                else
                    JLS_STEP_LOOP_ASM_NARROW();
                (void)cap_v;
                ok_m = in_line_m & ~fail_m;
            }
#endif
            win_stopped = win_now;
            // the bookkeeping owed to the lanes whose last step decoded a sample
            bool owed_last;
            int ra_stopped;
            {
                RegCtx ctx{u_a, u_tb, u_cc, u_n1};
                const int minus_delta = 1 - med3(ctx.b, 0, 1) - med3(ctx.b + ctx.n, 0, 1);
                ctx.b = med3(ctx.b + minus_delta * ctx.n, 1 - ctx.n, 0);
                ctx.c = med3(ctx.c - minus_delta, -128, 127);
                const bool owed = in_line && lane_of(ok_m);
                owed_last = owed;
                JLS_LOCKSTEP();
                ra_stopped = (int)lds_load<S>(lm); // Ra of a lane whose last step did not decode: stored by that step
                if (owed)
                {
                    store_pair(where, PrepEntry{(uint32_t)ctx.a, ((uint32_t)ctx.b << 16) | pack_bytes((uint32_t)ctx.c, (uint32_t)ctx.n)});
                    lm += kSz;
                    lds_store<S>(lm, (S)a);
                    p += u1 + k_last;
                    if (kChecked)
                        mm_seen |= t_mm;
                }
                JLS_LOCKSTEP();
            }
            if (in_line)
            {
                if (!owed_last)
                    a = ra_stopped;
                i = (lm - line_address) / kSz + 1;
                a_seen |= a_seen_now;
                // the reference raises invalid_data for k >= 16 and for |Errval| > 65535
                // (src/regular_mode_context.hpp:99-111, src/scan_decoder_core.hpp:38-69)
                if (k_seen >= (kChecked ? 16u : 10u) || (kChecked && mm_seen > 131071u))
                    retry = true;
            }
            else
                p = p_kept;
            // the entry of the sample a lane stopped at (its Rb is what the run handler needs first)
            if (in_line && !owed_last)
                entry_stopped = load_pair(pp);
        }
        // what stopped a scan that is still inside its line: Q = 0 is run mode, anything else an unusual code
        const int qs8 = (int)qsu8 - 2912;
        const bool stopped = in_line && !lane_of(ok_m);
        const bool in_run = stopped && qs8 == 0 && !retry;
        const bool slow = stopped && qs8 != 0 && !retry;
        // (nothing stopped, no line ended, nothing to hand back: the next round)
        if (in_line_m != 0 && __all(phase == kDone || (in_line && !stopped && !retry && i <= width && !(kChecked && a_seen >= (1u << 24)))))
            continue;

        // ---- run mode: reference src/scan_decoder_impl.hpp:264-337, src/scan_decoder_core.hpp:72-100
        // (what the step loop's own run service does not take; the handler for runs of length 0 that stood here from round 3 on --
        // out of the registers the lane left the loop with -- has nothing left to do: those runs are served inside the loop)
        // A run of any length that is interrupted inside its line, with codes that fit 64 bits of the stream: ONE request for the
        // bits (three ring words), both run contexts read ahead of knowing which, the length of the run from the number of
        // leading ones in closed form (run_prefix) instead of bit by bit, and one dependent LDS trip -- the sample above the
        // interruption sample.  What does not fit -- a run that reaches the end of its line, RUNindex beyond the table, a code
        // longer than the window -- takes the handler below, bit by bit (round 3's).
        bool windowed_runs = false;
        if (__any(in_run))
        {
            const uint32_t remaining = width - (i - 1);
            const uint32_t wi = (p >> 5) & (kRingWords - 1);
            const uint32_t r0 = ring[wi], r1 = ring[wi + 1], r2 = ring[wi + 2]; // (words kRingWords, kRingWords + 1 mirror words 0, 1)
            const RunCtx ctx0 = run_ctx[0], ctx1 = run_ctx[1];
            const uint32_t lo = funnel_shift(r1, r0, p), hi = funnel_shift(r2, r1, p);
            const uint64_t w = ((uint64_t)hi << 32) | lo; // the next 64 bits of the stream
            const uint32_t ones = lowest_one(~lo);        // blocks of the run-length code (0xFFFFFFFF: no zero among 32 bits)
            const uint32_t ri_after = (uint32_t)run_index + ones;
            const bool counted = ones < 32u && ri_after <= 31u;
            const uint32_t ri2 = counted ? ri_after : 0u;
            const uint32_t blocks = run_prefix(ri2) - run_prefix((uint32_t)run_index);
            const int j = run_j((int)ri2);
            const uint32_t after_zero = (uint32_t)(w >> ((ones + 1u) & 63u));
            const uint32_t run = blocks + field(after_zero, j);
            const uint32_t used = ones + 1u + (uint32_t)j; // bits of the run-length code (<= 47)
            const bool inside = counted && blocks < remaining && run < remaining; // (every block complete, interrupted before the end of the line)
            const uint32_t at = inside ? i + run : i;
            JLS_LOCKSTEP();
            const int b_at = (int)line[at]; // prev[at]: not overwritten yet
            const int which = same_level(a, b_at) ? 1 : 0;
            RunCtx ctx = which ? ctx1 : ctx0;
            const uint32_t goal = (uint32_t)ctx.a + (uint32_t)(ctx.n >> 1) * (uint32_t)ctx.ritype;
            int k = 0;
            if (goal > (uint32_t)ctx.n)
            {
                k = (int)leading_zeros((uint32_t)ctx.n) - (int)leading_zeros(goal);
                k += (((uint64_t)(uint32_t)ctx.n << k) < goal) ? 1 : 0;
            }
            const uint32_t w3 = (uint32_t)(w >> (used & 63u)); // the bits behind the run-length code
            const uint32_t zeros = lowest_one(w3);
            const int escape_from = t.limit - j - 1 - t.qbpp - 1;
            const int tail_bits = (int)zeros < escape_from ? k : t.qbpp;
            const uint32_t code_bits = zeros + 1u + (uint32_t)tail_bits;
            const bool fits = inside && k <= 24 && zeros < 32u && used + code_bits <= 64u;
            windowed_runs = __all(!in_run || fits);
            if (windowed_runs)
            {
                JLS_PATH(15); // run handlers out of one 64-bit window
                const uint32_t tail = field((uint32_t)(w >> ((used + zeros + 1u) & 63u)), tail_bits);
                const int em = (int)zeros < escape_from ? ((int)zeros << k) + (int)tail : (int)tail + 1;
                const int e = run_error_value(ctx, em + ctx.ritype, k);
                run_update(ctx, e, em, t.reset);
                const int x = interruption_sample(which != 0, a, b_at, e);
                JLS_LOCKSTEP();
                {
                    uint32_t r = (uint32_t)sub;
                    while (__any(in_run && r < run))
                    {
                        JLS_PATH(8); // trips of the run fill
                        if (in_run && r < run)
                            line[i + r] = (S)a;
                        r += G;
                    }
                }
                JLS_LOCKSTEP();
                if (in_run)
                {
                    run_ctx[which] = ctx;
                    line[at] = (S)x;
                    a = x;
                    run_index = ri2 > 0 ? (int)ri2 - 1 : 0;
                    i = at + 1;
                    p += used + code_bits;
                    rc_over = b_at; // prev[at] is Rc of the next sample, should it have no entry yet
                }
                JLS_LOCKSTEP();
            }
        }
        if (!windowed_runs && __any(in_run))
        {
            JLS_PATH(6); // run handler
            const uint32_t remaining = width - (i - 1);
            uint32_t run = 0;
            // A run event stops all scans of the wavefront, so what it costs is paid 64 / G times per run: the run-length code
            // (a few ones, a zero, J bits) comes out of ONE 32-bit window of the ring, bit by bit in a register.
            uint32_t window = peek32(ring, p);
            uint32_t used = 0;
            // (most run-length codes are the single zero bit of a run shorter than its first block: no trip at all then)
            bool counting = in_run && (window & 1u) != 0;
            if (in_run && !counting)
            {
                window >>= 1;
                used = 1;
            }
            while (__any(counting))
            {
                JLS_PATH(7); // bits of run-length codes
                if (counting)
                {
                    if (used == 32)
                    { // (a code longer than the window)
                        p += 32;
                        used = 0;
                        window = peek32(ring, p);
                    }
                    const uint32_t bit = window & 1u;
                    window >>= 1;
                    ++used;
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
                const int j = run_j(run_index);
                if (used + (uint32_t)j <= 32)
                {
                    run += field(window, j);
                    used += (uint32_t)j;
                }
                else
                {
                    p += used;
                    used = 0;
                    run += take_bits(ring, p, j);
                }
                if (run > remaining)
                {
                    retry = true;
                    interrupted = false;
                    run = 0;
                }
            }
            if (in_run)
                p += used;
            JLS_LOCKSTEP();
            {
                uint32_t r = (uint32_t)sub;
                while (__any(in_run && r < run))
                {
                    JLS_PATH(8); // trips of the run fill
                    if (in_run && r < run)
                        line[i + r] = (S)a;
                    r += G;
                }
            }
            const uint32_t at = i + run;
            JLS_LOCKSTEP();
            const int b_at = (int)line[interrupted ? at : 0]; // prev[at]: not overwritten yet
            const int which = same_level(a, b_at) ? 1 : 0;
            RunCtx ctx = run_ctx[which];
            int x = 0;
            if (interrupted)
            {
                // k = min{k : N << k >= A + (N >> 1) RItype} (src/run_mode_context.hpp:34-62) from the leading zeros
                const uint32_t goal = (uint32_t)ctx.a + (uint32_t)(ctx.n >> 1) * (uint32_t)ctx.ritype;
                int k = 0;
                if (goal > (uint32_t)ctx.n)
                {
                    k = (int)leading_zeros((uint32_t)ctx.n) - (int)leading_zeros(goal);
                    k += (((uint64_t)(uint32_t)ctx.n << k) < goal) ? 1 : 0;
                }
                const int limit = t.limit - run_j(run_index) - 1;
                // prefix and remainder out of one window where they fit (they do unless the code is an unusual one)
                const uint32_t w2 = peek32(ring, p);
                const uint32_t zeros = lowest_one(w2); // 0xFFFFFFFF: no one bit in the window
                int u = -1;
                int em = 0;
                if (k <= 24 && zeros < 32u)
                {
                    u = (int)zeros;
                    const int tail_bits = u < limit - t.qbpp - 1 ? k : t.qbpp;
                    uint32_t tail;
                    if (zeros + 1u + (uint32_t)tail_bits <= 32u)
                    {
                        tail = field(zeros == 31u ? 0u : w2 >> (zeros + 1u), tail_bits);
                        p += zeros + 1u + (uint32_t)tail_bits;
                    }
                    else
                    {
                        p += zeros + 1u;
                        tail = take_bits(ring, p, tail_bits);
                    }
                    em = u < limit - t.qbpp - 1 ? (u << k) + (int)tail : (int)tail + 1;
                }
                else if (k <= 24)
                { // a prefix of 32 zeros or more: the general reader (anything beyond kLongestPrefix: let the exact decoder classify it)
                    u = take_unary(ring, p, kLongestPrefix);
                    if (u >= 0)
                    {
                        if (u < limit - t.qbpp - 1)
                            em = (u << k) + (int)take_bits(ring, p, k);
                        else
                            em = (int)take_bits(ring, p, t.qbpp) + 1;
                    }
                }
                if (u < 0)
                {
                    retry = true;
                    interrupted = false;
                }
                else
                {
                    const int e = run_error_value(ctx, em + ctx.ritype, k);
                    run_update(ctx, e, em, t.reset);
                    x = interruption_sample(which != 0, a, b_at, e);
                }
            }
            JLS_LOCKSTEP();
            if (interrupted)
            {
                run_ctx[which] = ctx;
                line[at] = (S)x;
                a = x;
                if (run_index > 0)
                    --run_index;
                i = at + 1;
                rc_over = b_at; // prev[at] is Rc of the next sample, should it have no entry yet
            }
            else if (in_run && !retry)
                i = width + 1; // the run reached the end of the line
            JLS_LOCKSTEP();
        }

        // ---- one regular-mode sample with every case the step loop leaves out (escape codes, long prefixes)
        if (__any(slow))
        {
            JLS_PATH(9); // unusual codes
            const int rc = kWide ? (int)(entry_stopped.x & 0xFFFFu) : (int)(entry_stopped.x & 0xFFu), rb = (int)entry_stopped.y;
            const int s = qs8 >> 31;
            const int idx = ((qs8 ^ s) - s) >> 3;
            const Record rec = records[idx];
            RegCtx ctx{(int)rec.a, (int)rec.ncb >> 16, (int)(signed char)(rec.ncb >> 8), (int)(rec.ncb & 0xFFu)};
            const int k = regular_k(ctx);
            const int px = clamp_sample(t, med_predict(a, rb, rc) + ((ctx.c ^ s) - s));
            int x = 0;
            bool good = slow && k < 16;
            if (good)
            {
                const int u = take_unary(ring, p, kLongestPrefix);
                if (u < 0)
                    good = false;
                else
                {
                    int mm;
                    if ((uint32_t)u < limit_m)
                        mm = (u << k) | (int)take_bits(ring, p, k);
                    else
                        mm = (int)take_bits(ring, p, t.qbpp) + 1;
                    int e = unmap_error(mm);
                    if (k == 0)
                        e ^= error_correction(ctx, near);
                    if (!regular_update(ctx, e, near, t.reset))
                        good = false;
                    x = reconstructed(px, (e ^ s) - s);
                }
            }
            JLS_LOCKSTEP();
            if (good)
            {
                records[idx] = Record{(uint32_t)ctx.a,
                                      (uint32_t)ctx.n | (((uint32_t)ctx.c & 0xFFu) << 8) | ((uint32_t)ctx.b << 16)};
                line[i] = (S)x;
                a = x;
                ++i;
                rc_over = rb; // prev[i - 1] of the next sample, should it have no entry yet
            }
            else if (slow)
                retry = true;
            JLS_LOCKSTEP();
        }

        if (kChecked && a_seen >= (1u << 24))
            retry = true;
        if (retry)
            phase = kDone;

        // ---- finished line -> user's row
        if (NL == 1)
        {
            const bool ending = phase == kInLine && i > width;
            if (__any(ending))
            {
                JLS_PATH(10); // line ends
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
        else
        { // a component's line; behind the last component the pixel row: interleave, inverse colour transform
          // (src/copy_from_line_buffer.hpp:19-191)
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
                                v[c] = line0[(uint32_t)c * line_stride + 1 + xx];
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
                JLS_LOCKSTEP();
                if (ending)
                {
#pragma unroll
                    for (int c = 0; c < NL; ++c)
                    {
                        corner_of[c] = comp == c ? first : corner_of[c];
                        run_index_of[c] = comp == c ? run_index : run_index_of[c];
                    }
                    if (comp == NL - 1)
                    {
                        comp = 0;
                        ++y;
                    }
                    else
                        ++comp;
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
