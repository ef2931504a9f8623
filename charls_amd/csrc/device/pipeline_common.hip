// pipeline_common.hip -- what the stages of the lossless encoder pipeline share (tile_pipeline.hip, tile_pixel_mode.hip,
// block_stuffing.hip): chain numbering, how an interleaved scan is read ("coded lines", colour transforms), the limited-length
// Golomb code word, the look-back states of the pack stage, the work descriptor of the stuffing stage, and stage E in its
// one-wavefront-per-scan form (stuff_scan).  (Until round 4 this file also held the round-2 pipeline -- line-by-line
// scatter, one lane per chain --, which the tile pipeline has replaced for every scan the pipeline takes.)
//
// In lossless mode the causal template holds SOURCE samples, so everything except the adaptive statistics is a pure
// function of the image (reference src/scan_encoder_impl.hpp:109-144); the stages are described in
// tile_pipeline.hip.  MFMA is not used anywhere: nothing here is a contraction.
#pragma once
#include <hip/hip_runtime.h>

#include "knobs.h"
#include "scan_model.h"


namespace jls {
namespace pipe {

constexpr int kChains = 367;          // 0 = run chain, 1..364 = regular contexts, 365 = slots of run-interruption samples,
                                      // 366 = regular context 0
constexpr int kRegularChains = 365;   // chains coded by code_events: 1..364 and kZeroContextChain
constexpr int kInterruptChain = 365;  // no recurrence of its own: the run chain codes these samples, in the same order
constexpr int kZeroContextChain = 366; // ILV_SAMPLE only: a component whose own gradients are all zero while the pixel
                                       // as a whole is not in run mode is coded with regular context 0
constexpr uint32_t kGradientTable = 512; // LDS bytes of stage A's gradient table (8-bit samples)
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr uint16_t kNoEvent = 0xFFFF; // key of a sample that produces no code of its own
constexpr uint32_t kPackBlock = 4096; // samples per workgroup of stage D (256 threads x 16)
constexpr uint32_t kStatusInvalid = 1u;
constexpr uint32_t kChainPad = 16;    // chains start on multiples of 16 records (four 16-byte groups = one cache line), see bias_chains
constexpr uint32_t kChainSlack = kChains * kChainPad + 64; // spare records of sval/spos: padding + read-ahead

// Work descriptor of the stuffing stage (stuff_scan, block_stuffing.hip), parallel to the ScanDesc array.
struct Work
{
    uint32_t* raw;          // unstuffed bit stream, 32-bit words in big-endian bit order
    uint64_t raw_words;     // capacity of raw
    uint64_t* total_bits;   // [1]
    uint32_t* status;       // [1] kStatusInvalid when the reference would raise invalid_data
    uint32_t* stuff_tables; // block_stuffing.hip: kStuffWords words per chunk of the raw stream
};

// ILV_LINE scans are coded line by line and, within a line of pixels, component by component: "coded line" L is
// component L % components of pixel row L / components, with the contexts shared and ONE RUNindex per component
// (reference src/scan_encoder_impl.hpp:109-144 with component_count lines per row).  The previous line of the same
// component is coded line L - line_step.
JLS_DEV uint32_t coded_lines(const ScanDesc& d)
{
    return d.interleave_mode == 1 ? d.height * (uint32_t)d.components : d.height;
}
JLS_DEV bool sign_fits_record(const ScanDesc& d) // x : 16 | Px : 15 | sign of the context : 1
{
    return d.bits_per_sample <= 15;
}
JLS_DEV uint32_t line_step(const ScanDesc& d)
{
    return d.interleave_mode == 1 ? (uint32_t)d.components : 1u;
}

template <typename S>
JLS_DEV void load_pixel(const ScanDesc& d, uint32_t y, uint32_t x, int mask, int out[4]);

// Sample x of coded line `line` as the codec sees it.  ILV is the scan's interleave mode as a compile-time constant: the
// planar instantiation (the headline path) carries none of the interleaved code.
template <typename S, int ILV>
JLS_DEV int load_sample(const ScanDesc& d, uint32_t line, uint32_t x, int mask)
{
    if (ILV == 1)
    {
        int px[4];
        load_pixel<S>(d, line / (uint32_t)d.components, x, mask, px);
        return px[line % (uint32_t)d.components];
    }
    const S* row = reinterpret_cast<const S*>(d.pixels + (size_t)line * d.pixel_stride);
    return (int)row[x] & mask;
}

// Samples per line as the stages after A see them: ILV_SAMPLE scans are coded pixel by pixel, component by component
// (reference src/scan_encoder_impl.hpp:146-247), so their "line" is width * components samples long and the raster
// index of a sample is (y * width + x) * components + c.
JLS_DEV uint32_t line_samples(const ScanDesc& d)
{
    return d.interleave_mode == 2 ? d.width * (uint32_t)d.components : d.width;
}

// One pixel of an interleaved scan as the codec sees it: masked to the sample precision, colour transform applied
// (src/copy_to_line_buffer.hpp:37-93, src/color_transform.hpp).
template <typename S>
JLS_DEV void load_pixel(const ScanDesc& d, uint32_t y, uint32_t x, int mask, int out[4])
{
    const S* px = reinterpret_cast<const S*>(d.pixels + (size_t)y * d.pixel_stride) + (size_t)x * d.components;
    for (int c = 0; c < d.components; ++c)
        out[c] = (int)px[c] & mask;
    if (d.color_transformation != 0)
    {
        unsigned t[3];
        hp_forward(d.color_transformation, sizeof(S) == 2, out[0], out[1], out[2], t);
        out[0] = (int)t[0];
        out[1] = (int)t[1];
        out[2] = (int)t[2];
    }
}

typedef uint32_t u32x4 __attribute__((vector_size(16)));

struct CodeWord
{
    uint64_t bits;
    int len;
};

// Limited-length Golomb code as (bits, length): src/scan_encoder_core.hpp:69-103.
JLS_DEV CodeWord golomb_word(const Traits& t, int k, int m, int limit)
{
    CodeWord c;
    const int hb = m >> k;
    if (hb < limit - t.qbpp - 1)
    {
        c.len = hb + 1 + k;
        c.bits = (1ull << k) | (uint64_t)((uint32_t)m & ((1u << k) - 1u));
    }
    else
    {
        c.len = limit;
        c.bits = (1ull << t.qbpp) | (uint64_t)((uint32_t)(m - 1) & ((1u << t.qbpp) - 1u));
    }
    return c;
}

// Look-back states of the pack stage (tile_pipeline.hip: pack_tiles): bits 62..63 = state (0 nothing, 1 own bits, 2 bits up to
// and including this tile), bits 0..61 the value; zero before the launch.
constexpr uint64_t kBlockOwn = 1ull << 62, kBlockUpTo = 2ull << 62, kBlockValue = (1ull << 62) - 1ull;

// E: raw bits -> stuffed bytes.  After a 0xFF byte the next byte carries 7 bits (MSB 0); a final 0xFF is followed by 0x00;
// the last partial byte is zero padded (src/scan_encoder.hpp:103-180).
//
// Stuffing is sequential only through the (rare) 0xFF bytes -- one byte in 256 of random bits, one in ~760 of real coded
// data --, so a wavefront speculates: lane l cuts output bytes 8 l .. 8 l + 7 of the next 512 out of the raw bit stream
// assuming no 0xFF occurs before them (64 bits straight from memory, the lanes' windows contiguous: 512 bytes per request),
// finds the first 0xFF among its own eight bytes with three folds, a ballot finds the first lane that has one, everything up
// to and including that byte is final and leaves as 8-byte stores, and the next round starts behind it with a 7-bit byte --
// which is the ordinary case again once the window is read one bit earlier and its first bit cleared.  (Until round 4 a lane
// cut ONE byte per round out of an LDS ring: 4.2 ns per byte, 99 ms for the 23.5 MB of a 4096 x 4096 RGB frame.)
struct StuffCursor
{
    uint64_t bp;       // raw bit at which the next output byte starts
    uint64_t written;  // output bytes so far
    bool short_first;  // that byte follows a 0xFF: 7 payload bits
};

struct __attribute__((packed)) UnalignedQuad
{
    uint64_t v;
};
JLS_DEV uint64_t raw_window(const uint8_t* raw, uint64_t bit) // the 64 raw bits from `bit` on, first bit most significant
{
    const uint8_t* p = raw + (bit >> 3);
    const uint64_t v = __builtin_bswap64(reinterpret_cast<const UnalignedQuad*>(p)->v);
    const uint32_t s = (uint32_t)(bit & 7u);
    return s == 0 ? v : (v << s) | ((uint64_t)p[8] >> (8 - s));
}

// The output bytes that START in [cur.bp, end), by ONE wavefront (all 64 lanes call it).  kEmit: they are stored at
// out + cur.written (bytes at or behind `capacity` are dropped).  raw must be readable 16 bytes beyond bit `end`.
template <bool kEmit>
JLS_DEV void stuff_walk(const uint8_t* raw, StuffCursor& cur, uint64_t end, uint8_t* out, uint64_t capacity)
{
    const uint32_t lane = threadIdx.x & 63u;
    while (cur.bp < end)
    {
        const uint64_t base = cur.bp - (cur.short_first ? 1u : 0u); // (a 0xFF byte lies before a short byte: no underflow)
        const uint64_t pos = base + 64ull * lane;
        uint64_t w = pos < end ? raw_window(raw, pos) : 0ull;
        if (cur.short_first && lane == 0)
            w &= ~(1ull << 63); // the 7-bit byte: its first bit is the stuffed zero
        const uint32_t valid = pos >= end ? 0u : (end - pos >= 64 ? 8u : (uint32_t)((end - pos + 7) / 8)); // bytes of this lane that start before `end`
        uint64_t ones = w & (w >> 1);
        ones &= ones >> 2;
        ones &= ones >> 4;
        ones &= 0x0101010101010101ull;
        const uint32_t first_ff = ones ? (uint32_t)(__builtin_clzll(ones) / 8) : 8u; // in stream order
        const unsigned long long ffm = __ballot(first_ff < valid);
        const uint64_t left = end - base;
        const uint32_t round_bytes = left >= 512 * 8 ? 512u : (uint32_t)((left + 7) / 8);
        uint32_t total = round_bytes;
        if (ffm != 0)
        {
            const int f = (int)__ffsll(ffm) - 1;
            total = 8u * (uint32_t)f + (uint32_t)__shfl((int)first_ff, f) + 1u;
        }
        if (kEmit)
        {
            const uint32_t mine = total > 8u * lane ? (total - 8u * lane >= 8u ? 8u : total - 8u * lane) : 0u;
            const uint64_t at = cur.written + 8ull * lane;
            if (mine == 8 && at + 8 <= capacity)
                reinterpret_cast<UnalignedQuad*>(out + at)->v = __builtin_bswap64(w);
            else
                for (uint32_t k = 0; k < mine; ++k)
                    if (at + k < capacity)
                        out[at + k] = (uint8_t)(w >> (56 - 8 * k));
        }
        cur.written += total;
        cur.bp = base + 8ull * total;
        cur.short_first = ffm != 0;
    }
}

// One wavefront per scan: the whole stream in sequence (the passes whose stuffing runs under the next pass's first stages;
// batches take speculative_stuffing.hip, single frames block_stuffing.hip).  Result flags: bit 1 = the capacity is within 3
// bytes of the output size, where the reference's accept/reject decision depends on its 32-bit flush history; the host then
// re-runs the exact serial kernel.
JLS_DEV ScanResult stuff_stream_sequential(const ScanDesc& d, const Work& w, uint64_t total_bits)
{
    ScanResult r{kOk, 0, 0};
    StuffCursor cur{0, 0, false};
    stuff_walk<true>(reinterpret_cast<const uint8_t*>(w.raw), cur, total_bits, d.stream, d.stream_capacity);
    if (cur.short_first)
    { // src/scan_encoder.hpp:107-112: a trailing 0xFF is followed by a byte of seven zero bits
        if ((threadIdx.x & 63u) == 0 && cur.written < d.stream_capacity)
            d.stream[cur.written] = 0;
        ++cur.written;
    }
    r.bytes = cur.written;
    if (cur.written > d.stream_capacity)
        r.errc = kDestinationTooSmall;
    else if (d.stream_capacity - cur.written < 4)
        r.flags = 2; // undecidable here, see above
    return r;
}

__global__ void __launch_bounds__(64) stuff_scan(const ScanDesc* __restrict__ descs, const Work* __restrict__ works,
                                                 ScanResult* __restrict__ results)
{
    const ScanDesc d = descs[blockIdx.x];
    const Work w = works[blockIdx.x];
    const uint64_t total_bits = *w.total_bits;
    ScanResult r{kOk, 0, 0};
    if ((*w.status & kStatusInvalid) != 0)
        r.errc = kInvalidData;
    else if ((total_bits + 7) / 8 > w.raw_words * 4)
        r.errc = kDestinationTooSmall; // the unstuffed stream alone exceeds the destination
    else
        r = stuff_stream_sequential(d, w, total_bits);
    if (threadIdx.x == 0)
        results[blockIdx.x] = r;
}

} // namespace pipe
} // namespace jls
