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

// E: one wavefront per scan: raw bits -> stuffed bytes.  After a 0xFF byte the next byte carries 7 bits (MSB 0); a final
// 0xFF is followed by 0x00; the last partial byte is zero padded (src/scan_encoder.hpp:103-180).
//
// Stuffing is sequential only through the (rare) 0xFF bytes, so the wavefront speculates: lane l cuts output byte l of
// the next 64 out of the raw bit stream assuming no 0xFF occurs before it; a ballot finds the first 0xFF, everything up
// to and including it is final and is stored with one coalesced write, and the next round starts behind it with a
// 7-bit first byte.  Result flags: bit 1 = the capacity is within 3 bytes of the output size, where the reference's
// accept/reject decision depends on its 32-bit flush history; the host then re-runs the exact serial kernel.
__global__ void __launch_bounds__(64) stuff_scan(const ScanDesc* __restrict__ descs, const Work* __restrict__ works,
                                                 ScanResult* __restrict__ results)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[4096];
    const ScanDesc d = descs[blockIdx.x];
    const Work w = works[blockIdx.x];
    const int lane = threadIdx.x;
    const uint64_t total_bits = *w.total_bits;
    const uint64_t raw_bytes_cap = w.raw_words * 4;
    const uint8_t* raw = reinterpret_cast<const uint8_t*>(w.raw);
    ScanResult r{kOk, 0, 0};

    if ((*w.status & kStatusInvalid) != 0)
        r.errc = kInvalidData;
    else if ((total_bits + 7) / 8 > raw_bytes_cap)
        r.errc = kDestinationTooSmall; // the unstuffed stream alone exceeds the destination
    if (r.errc != kOk)
    {
        if (lane == 0)
            results[blockIdx.x] = r;
        return;
    }

    uint64_t loaded = 0;  // raw bytes [loaded - 4096, loaded) are resident in s_in (ring)
    uint64_t bp = 0;      // next raw bit
    uint64_t written = 0; // output bytes so far
    bool first_short = false; // the next output byte follows a 0xFF: 7 payload bits
    bool last_ff = false;

    while (bp < total_bits)
    {
        JLS_LOCKSTEP();
        // raw bytes needed by this round: 64 output bytes + slack
        while (loaded < (bp >> 3) + 80 && loaded < raw_bytes_cap)
        {
            const uint64_t o = loaded + (uint64_t)lane * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (o + 16 <= raw_bytes_cap)
                v = *reinterpret_cast<const uint4*>(raw + o);
            *reinterpret_cast<uint4*>(s_in + (o & 4095)) = v;
            loaded += 1024;
            __syncthreads();
        }
        // lane l: bits [start, start + n) with n = 7 for a byte that follows a 0xFF
        const int n = (lane == 0 && first_short) ? 7 : 8;
        const uint64_t start = bp + (uint64_t)lane * 8 - ((lane != 0 && first_short) ? 1 : 0);
        const bool active = start < total_bits;
        const uint32_t b0 = s_in[(start >> 3) & 4095];
        const uint32_t b1 = s_in[((start >> 3) + 1) & 4095];
        const uint32_t two = (b0 << 8) | b1; // 16 raw bits, MSB first (zeros beyond the end of the stream)
        const uint32_t byte = (two >> (16 - (int)(start & 7) - n)) & ((1u << n) - 1u);
        const unsigned long long act = __ballot(active);
        const unsigned long long ffm = __ballot(active && byte == 0xFFu);
        const int count = __popcll(act);                              // active lanes are a prefix
        const int upto = ffm ? (int)__ffsll(ffm) : count;             // bytes that are final in this round
        if (lane < upto && written + (uint64_t)lane < d.stream_capacity)
            d.stream[written + lane] = (uint8_t)byte;
        written += (uint64_t)upto;
        bp = bp + (uint64_t)upto * 8 - (first_short ? 1 : 0);
        first_short = ffm != 0 && upto <= count;
        last_ff = ffm != 0;
    }
    if (last_ff)
    { // src/scan_encoder.hpp:107-112: a trailing 0xFF is followed by a byte of seven zero bits
        if (lane == 0 && written < d.stream_capacity)
            d.stream[written] = 0;
        ++written;
    }

    r.bytes = written;
    if (written > d.stream_capacity)
        r.errc = kDestinationTooSmall;
    else if (d.stream_capacity - written < 4)
        r.flags = 2; // undecidable here, see above
    if (lane == 0)
        results[blockIdx.x] = r;
}

} // namespace pipe
} // namespace jls
