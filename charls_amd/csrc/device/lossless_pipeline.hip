// lossless_pipeline.hip -- parallel JPEG-LS encoder for lossless scans: single-component (the BASELINE headline path) and
// the component-interleaved modes ILV_LINE / ILV_SAMPLE with 2..4 components (see analyze_pixels, coded_lines).
//
// In lossless mode the causal template holds SOURCE samples, so everything except the adaptive statistics is a pure
// function of the image (reference src/scan_encoder_impl.hpp:109-144).  The scan is therefore coded in
// stages that each expose the parallelism they really have:
//
//   A  analyze_rows      one wavefront per scan line: context id, sign, MED prediction for every sample (coalesced; bound
//                        by the instructions it issues); run-mode segmentation of the line resolved as a carry chain
//                        over ballot masks
//   B1 chain_offsets     per-line histograms of the 365 statistic chains (364 regular contexts + the run chain) ->
//                        exclusive offsets (a stable counting sort by context, raster order kept inside a chain)
//   B2 scatter_events    events move to their chain, stable ranks from a wave-level sort (deterministic, no atomics on order);
//                        the slot of every sample is recorded (inv) so that codes can stay in chain order until D
//   C1 bias_chains       one LANE per chain: the {B,C,N} recurrence turns the chain's samples into Errval (the only
//                        serial dependency of regular mode); the run chain carries RUNindex and the two run-interruption
//                        contexts and codes its events directly.  Chains of different contexts never interact in
//                        lossless mode, so 365 x scans lanes run concurrently.
//   C2 code_events       one wavefront per chain: A is a segmented prefix sum of |Errval|, N a function of the event
//                        index -> k and the Golomb words of 64 events per step, stored in chain order (coalesced)
//   D  write_raw_bits    codes are gathered in raster order and concatenated MSB-first into the unstuffed bit stream
//                        (a line's samples of one chain are neighbours in chain order: the gathers hit whole cache
//                        lines, where a scatter of 8-byte codes by chain wrote every line many times); the bit offset of
//                        a block comes from a chained look-back scan among the blocks of the scan
//   E  stuff_scan        JPEG-LS 0xFF bit stuffing + end-of-scan padding (src/scan_encoder.hpp:103-180), one wavefront
//                        per scan streaming through LDS; run by runtime.hip on a side stream, under the next pass
//
// Output is byte-identical to scan_encoder::encode_scan.  MFMA is not used anywhere: nothing here is a contraction.
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

// Per-scan work areas (device pointers), parallel to the ScanDesc array.
struct Work
{
    uint16_t* key;         // [H*W] chain | sign << 9, or kNoEvent
    uint32_t* val;         // [H*W] x | Px << 16   (run start: run length | end-of-line << 31)
    uint32_t* hist;        // [H][kChains] events per line and chain -> exclusive prefix over lines
    uint32_t* chain_total; // [kChains]
    uint32_t* chain_base;  // [kChains] offset of the chain in sval/spos
    uint32_t* sval;        // [H*W + kChainSlack] events grouped by chain, raster order inside a chain; 16-byte aligned
    uint32_t* spos;        // [H*W + kChainSlack] raster index of the event | sign << 31; 16-byte aligned
    uint32_t* inv;         // [H*W] slot (index into sval/code/len) of the sample, or kNoSlot
    uint8_t* len;          // [H*W + kChainSlack] code length per slot
    uint64_t* code;        // [H*W + kChainSlack] code bits per slot, right aligned (re-uses key/val, dead after B2)
    uint64_t* blockbase;   // [ceil(H*W / kPackBlock)] look-back states of write_raw_bits
    uint32_t* raw;         // unstuffed bit stream, 32-bit words in big-endian bit order; zeroed before stage D
    uint64_t raw_words;    // capacity of raw
    uint64_t* total_bits;  // [1]
    uint32_t* status;      // [1] kStatusInvalid when the reference would raise invalid_data
    uint32_t* stuff_tables; // block_stuffing.hip: kStuffWords words per chunk of the raw stream
};

// Workgroup -> scan line.  Workgroup b runs on XCD b % 8 (each XCD has its own L2): giving every XCD a contiguous band of
// lines keeps the short per-chain segments that neighbouring lines append to the same cache lines in ONE L2, where they
// merge before they are written back.  Launch with grid.x = 8 * ceil(height / 8); returns height for idle workgroups.
JLS_DEV uint32_t xcd_band_row(uint32_t block, uint32_t height)
{
    const uint32_t band = (height + 7) / 8;
    const uint32_t y = (block & 7u) * band + (block >> 3);
    return (block >> 3) < band && y < height ? y : height;
}

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

template <typename S, int ILV>
__global__ void __launch_bounds__(64) analyze_rows(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    JLS_DYNAMIC_LDS(smem);
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const Traits t = make_traits(d);
    const uint32_t lines = ILV == 1 ? coded_lines(d) : d.height;
    const uint32_t y = xcd_band_row(blockIdx.x, lines); // coded line
    if (y >= lines)
        return;
    const uint32_t step = ILV == 1 ? line_step(d) : 1u;
    const int lane = threadIdx.x;
    const uint32_t width = d.width;
    const uint32_t chunks = (width + 63) / 64;
    uint64_t* s_eq = reinterpret_cast<uint64_t*>(smem);
    uint64_t* s_q0 = s_eq + chunks;
    uint32_t* s_next = reinterpret_cast<uint32_t*>(s_q0 + chunks);
    uint32_t* s_hist = s_next + chunks;
    unsigned char* s_grad = reinterpret_cast<unsigned char*>(s_hist + kChains); // kGradientTable bytes
    const int mask = (1 << d.bits_per_sample) - 1;

    for (int c = lane; c < kChains; c += 64)
        s_hist[c] = 0;
    // 8-bit samples: the quantised gradient (+ 4) of every difference -255 .. 255 from a table -- the arithmetic form of
    // src/jpegls_algorithm.hpp:173-194 is eight comparisons per gradient, three gradients per sample, and this stage is
    // bound by the instructions it issues (143 per 64 samples, of which the table takes 40 away)
    if (sizeof(S) == 1)
    {
        for (int q = lane; q < 511; q += 64)
            s_grad[q] = (unsigned char)(quantize(t, q - 255) + 4);
        __syncthreads();
    }

    // edge samples of the line (src/scan_codec.hpp:189-195 and the two-line ping-pong of src/scan_encoder_impl.hpp:55-106)
    const int edge_a = y >= step ? load_sample<S, ILV>(d, y - step, 0, mask) : 0;     // cur[0]  = prev[1]
    const int edge_c = y >= 2 * step ? load_sample<S, ILV>(d, y - 2 * step, 0, mask) : 0; // prev[0] = two lines up, first sample
    uint16_t* key_row = w.key + (size_t)y * width;
    uint32_t* val_row = w.val + (size_t)y * width;

    // ---- pass 1: every sample as if coded in regular mode; equality / zero-context masks per 64-sample chunk
    for (uint32_t k = 0; k < chunks; ++k)
    {
        const uint32_t x = k * 64 + lane;
        bool eq = false, q0 = false;
        if (x < width)
        {
            const int v = load_sample<S, ILV>(d, y, x, mask);
            const int ra = x > 0 ? load_sample<S, ILV>(d, y, x - 1, mask) : edge_a;
            int rb = 0, rc = 0, rd = 0;
            if (y >= step)
            {
                rb = load_sample<S, ILV>(d, y - step, x, mask);
                rc = x > 0 ? load_sample<S, ILV>(d, y - step, x - 1, mask) : edge_c;
                rd = load_sample<S, ILV>(d, y - step, x + 1 < width ? x + 1 : width - 1, mask);
            }
            else
                rc = x > 0 ? 0 : edge_c;
            const int qs = sizeof(S) == 1
                               ? ((int)s_grad[rd - rb + 255] * 9 + (int)s_grad[rb - rc + 255]) * 9 + (int)s_grad[rc - ra + 255] - 364
                               : context_id(t, ra, rb, rc, rd);
            const int sg = qs >> 31;
            const int ctx = (qs ^ sg) - sg;
            const int px = med3(ra + rb - rc, ra, rb); // MED predictor = median(Ra, Rb, Ra + Rb - Rc), src/jpegls_algorithm.hpp:143-161
            key_row[x] = (uint16_t)(ctx | ((sg & 1) << 9));
            val_row[x] = (uint32_t)v | ((uint32_t)px << 16);
            eq = v == ra;
            q0 = qs == 0;
        }
        const unsigned long long m_eq = __ballot(eq);
        const unsigned long long m_q0 = __ballot(q0);
        if (lane == 0)
        {
            s_eq[k] = m_eq;
            s_q0[k] = m_q0;
        }
    }
    __syncthreads();

    // ---- pass 2 (reverse): column of the first sample at or after the NEXT chunk that differs from its left neighbour
    if (lane == 0)
    {
        uint32_t carry = width;
        for (uint32_t k = chunks; k-- > 0;)
        {
            s_next[k] = carry;
            const uint32_t valid = width - k * 64 >= 64 ? 64 : width - k * 64;
            const unsigned long long vm = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
            const unsigned long long neq = ~s_eq[k] & vm;
            if (neq)
                carry = k * 64 + (uint32_t)__ffsll(neq) - 1;
        }
    }
    __syncthreads();

    // ---- pass 3: run-mode state before every sample.  s' = eq & (s | q0) is a carry chain: generate = eq & q0,
    // propagate = eq, so one 64-bit addition per chunk resolves 64 samples (src/scan_encoder_impl.hpp:249-275).
    unsigned long long carry = 0;
    for (uint32_t k = 0; k < chunks; ++k)
    {
        const unsigned long long a = s_eq[k];
        const unsigned long long b = s_eq[k] & s_q0[k];
        const unsigned long long sum = a + b + carry;
        const unsigned long long st = sum ^ a ^ b; // bit i: in-run state before sample i
        carry = (((a & b) | ((a | b) & st)) >> 63) & 1ull;
        const uint32_t x = k * 64 + lane;
        if (x < width)
        {
            const bool s = (st >> lane) & 1ull;
            const bool q0 = (s_q0[k] >> lane) & 1ull;
            const bool eq = (a >> lane) & 1ull;
            if (!(s || q0))
                atomicAdd(&s_hist[key_row[x] & 0x1FF], 1u); // regular sample, key already written by this lane
            else if (s && eq)
                key_row[x] = kNoEvent; // inside a run
            else if (s)
            { // the sample that ends a run started earlier: coded by the run chain, owns a slot of its own
                key_row[x] = (uint16_t)kInterruptChain;
                atomicAdd(&s_hist[kInterruptChain], 1u);
            }
            else
            { // a run starts here (possibly of length 0)
                uint32_t run = 0, eol = 0;
                if (eq)
                {
                    const uint32_t valid = width - k * 64 >= 64 ? 64 : width - k * 64;
                    const unsigned long long vm = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
                    const unsigned long long neq = (~a & vm) >> lane;
                    const uint32_t end = neq ? x + (uint32_t)__ffsll(neq) - 1 : s_next[k];
                    run = end - x;
                    eol = end == width ? 1u : 0u;
                }
                key_row[x] = 0;
                val_row[x] = run | (eol << 31);
                atomicAdd(&s_hist[0], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* hist_row = w.hist + (size_t)y * kChains;
    for (int c = lane; c < kChains; c += 64)
        hist_row[c] = s_hist[c];
}

// ---------------------------------------------------------------------------------------------------------------
// A for ILV_SAMPLE scans (2..4 components): same launch geometry and LDS as analyze_rows, one LANE per PIXEL.  A pixel is
// in run mode when all its components are (src/scan_encoder_impl.hpp:171-199, 222-247); regular-mode pixels give one
// event per component, in component order, all drawing on the ONE set of contexts.  The component-0 sample of the pixel
// where a run starts carries the run event; the components of the pixel that ends a run are events of
// kInterruptChain, except component 0 of a run of length 0, whose code is merged with the run-length code.
template <typename S>
__global__ void __launch_bounds__(64) analyze_pixels(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    JLS_DYNAMIC_LDS(smem);
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const Traits t = make_traits(d);
    const uint32_t y = xcd_band_row(blockIdx.x, d.height);
    if (y >= d.height)
        return;
    const int lane = threadIdx.x;
    const uint32_t width = d.width;
    const int nc = d.components;
    const uint32_t chunks = (width + 63) / 64;
    uint64_t* s_eq = reinterpret_cast<uint64_t*>(smem);
    uint64_t* s_q0 = s_eq + chunks;
    uint32_t* s_next = reinterpret_cast<uint32_t*>(s_q0 + chunks);
    uint32_t* s_hist = s_next + chunks;
    const int mask = (1 << d.bits_per_sample) - 1;
    for (int c = lane; c < kChains; c += 64)
        s_hist[c] = 0;
    int edge_a[4] = {0, 0, 0, 0}, edge_c[4] = {0, 0, 0, 0}; // cur[0] = prev[1]; prev[0] = first pixel of line y-2
    if (y > 0)
        load_pixel<S>(d, y - 1, 0, mask, edge_a);
    if (y > 1)
        load_pixel<S>(d, y - 2, 0, mask, edge_c);
    uint16_t* key_row = w.key + (size_t)y * width * nc;
    uint32_t* val_row = w.val + (size_t)y * width * nc;

    // ---- pass 1: every component as if coded in regular mode; equality / zero-context masks per 64-pixel chunk
    for (uint32_t k = 0; k < chunks; ++k)
    {
        const uint32_t x = k * 64 + lane;
        bool eq = false, q0 = false;
        if (x < width)
        {
            int v[4], ra[4], rb[4] = {0, 0, 0, 0}, rc[4] = {0, 0, 0, 0}, rd[4] = {0, 0, 0, 0};
            load_pixel<S>(d, y, x, mask, v);
            if (x > 0)
                load_pixel<S>(d, y, x - 1, mask, ra);
            else
                for (int c = 0; c < 4; ++c)
                    ra[c] = edge_a[c];
            if (y > 0)
            {
                load_pixel<S>(d, y - 1, x, mask, rb);
                if (x > 0)
                    load_pixel<S>(d, y - 1, x - 1, mask, rc);
                else
                    for (int c = 0; c < 4; ++c)
                        rc[c] = edge_c[c];
                load_pixel<S>(d, y - 1, x + 1 < width ? x + 1 : width - 1, mask, rd);
            }
            eq = true;
            q0 = true;
            for (int c = 0; c < nc; ++c)
            {
                const int qs = context_id(t, ra[c], rb[c], rc[c], rd[c]);
                const int sg = qs >> 31;
                const int ctx = (qs ^ sg) - sg;
                key_row[(size_t)x * nc + c] = (uint16_t)((ctx == 0 ? kZeroContextChain : ctx) | ((sg & 1) << 9));
                val_row[(size_t)x * nc + c] = (uint32_t)v[c] | ((uint32_t)med_predict(ra[c], rb[c], rc[c]) << 16);
                eq = eq && v[c] == ra[c];
                q0 = q0 && qs == 0;
            }
        }
        const unsigned long long m_eq = __ballot(eq);
        const unsigned long long m_q0 = __ballot(q0);
        if (lane == 0)
        {
            s_eq[k] = m_eq;
            s_q0[k] = m_q0;
        }
    }
    __syncthreads();

    // ---- pass 2 (reverse): column of the first pixel at or after the NEXT chunk that differs from its left neighbour
    if (lane == 0)
    {
        uint32_t carry = width;
        for (uint32_t k = chunks; k-- > 0;)
        {
            s_next[k] = carry;
            const uint32_t valid = width - k * 64 >= 64 ? 64 : width - k * 64;
            const unsigned long long vm = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
            const unsigned long long neq = ~s_eq[k] & vm;
            if (neq)
                carry = k * 64 + (uint32_t)__ffsll(neq) - 1;
        }
    }
    __syncthreads();

    // ---- pass 3: run-mode state before every pixel (the carry chain of analyze_rows)
    unsigned long long carry = 0;
    for (uint32_t k = 0; k < chunks; ++k)
    {
        const unsigned long long a = s_eq[k];
        const unsigned long long b = s_eq[k] & s_q0[k];
        const unsigned long long sum = a + b + carry;
        const unsigned long long st = sum ^ a ^ b;
        carry = (((a & b) | ((a | b) & st)) >> 63) & 1ull;
        const uint32_t x = k * 64 + lane;
        if (x < width)
        {
            const bool s = (st >> lane) & 1ull;
            const bool q0 = (s_q0[k] >> lane) & 1ull;
            const bool eq = (a >> lane) & 1ull;
            uint16_t* keys = key_row + (size_t)x * nc;
            if (!(s || q0))
            {
                for (int c = 0; c < nc; ++c)
                    atomicAdd(&s_hist[keys[c] & 0x1FF], 1u);
            }
            else if (s && eq)
            {
                for (int c = 0; c < nc; ++c)
                    keys[c] = kNoEvent;
            }
            else if (s)
            { // the pixel that ends a run started earlier
                for (int c = 0; c < nc; ++c)
                    keys[c] = (uint16_t)kInterruptChain;
                atomicAdd(&s_hist[kInterruptChain], (uint32_t)nc);
            }
            else
            { // a run starts here (possibly of length 0: then this pixel also ends it)
                uint32_t run = 0, eol = 0;
                if (eq)
                {
                    const uint32_t valid = width - k * 64 >= 64 ? 64 : width - k * 64;
                    const unsigned long long vm = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
                    const unsigned long long neq = (~a & vm) >> lane;
                    const uint32_t end = neq ? x + (uint32_t)__ffsll(neq) - 1 : s_next[k];
                    run = end - x;
                    eol = end == width ? 1u : 0u;
                }
                keys[0] = 0;
                val_row[(size_t)x * nc] = run | (eol << 31);
                atomicAdd(&s_hist[0], 1u);
                for (int c = 1; c < nc; ++c)
                    keys[c] = eq ? kNoEvent : (uint16_t)kInterruptChain;
                if (!eq)
                    atomicAdd(&s_hist[kInterruptChain], (uint32_t)(nc - 1));
            }
        }
    }
    __syncthreads();
    uint32_t* hist_row = w.hist + (size_t)y * kChains;
    for (int c = lane; c < kChains; c += 64)
        hist_row[c] = s_hist[c];
}

// ---------------------------------------------------------------------------------------------------------------
// B1: one workgroup of 384 threads per scan.
__global__ void __launch_bounds__(384) chain_offsets(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    __shared__ uint32_t s_total[384];
    const ScanDesc d = descs[blockIdx.x];
    const Work w = works[blockIdx.x];
    const int c = threadIdx.x;
    uint32_t running = 0;
    if (c < kChains)
    {
        for (uint32_t y = 0; y < coded_lines(d); ++y)
        {
            const uint32_t n = w.hist[(size_t)y * kChains + c];
            w.hist[(size_t)y * kChains + c] = running;
            running += n;
        }
        w.chain_total[c] = running;
    }
    s_total[c] = c < kChains ? running : 0;
    if (c == 383)
    { // the two result words of the later stages start at zero (stages C2 / D write them)
        *w.total_bits = 0;
        *w.status = 0;
    }
    __syncthreads();
    if (c == 0)
    { // 365 values: a serial scan is cheaper than its synchronisation
        uint32_t acc = 0;
        for (int i = 0; i < kChains; ++i)
        {
            w.chain_base[i] = acc;
            acc += (s_total[i] + kChainPad - 1) / kChainPad * kChainPad;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// B2: grid (8 * ceil(height / 8), scans), one wavefront per line; stable scatter of the line's events to their chains.
__global__ void __launch_bounds__(64) scatter_events(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    __shared__ uint32_t s_cnt[kChains];
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const uint32_t y = xcd_band_row(blockIdx.x, coded_lines(d));
    if (y >= coded_lines(d))
        return;
    const int lane = threadIdx.x;
    const uint32_t width = line_samples(d);
    for (int c = lane; c < kChains; c += 64)
        s_cnt[c] = w.chain_base[c] + w.hist[(size_t)y * kChains + c];
    __syncthreads();
    for (uint32_t k = 0; k < (width + 63) / 64; ++k)
    {
        const uint32_t x = k * 64 + lane;
        uint16_t key = kNoEvent;
        uint32_t v = 0;
        if (x < width)
        {
            key = w.key[(size_t)y * width + x];
            v = w.val[(size_t)y * width + x];
        }
        const bool has = key != kNoEvent;
        // Stable rank of every event inside its chain: the 64 (chain, lane) keys are sorted with a bitonic network
        // (21 compare-exchange steps instead of one ballot round per distinct chain of the chunk), a lane's rank is its
        // distance from the start of its run of equal chains, and the slot number travels back with ds_permute.
        uint32_t sorted = has ? ((uint32_t)(key & 0x1FF) << 6) | (uint32_t)lane : 0x7FFF0000u | (uint32_t)lane;
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
            for (int stride = size >> 1; stride > 0; stride >>= 1)
            {
                const uint32_t other = (uint32_t)__shfl_xor((int)sorted, stride);
                const bool keep_min = ((lane & stride) == 0) == ((lane & size) == 0);
                const uint32_t lo = sorted < other ? sorted : other, hi = sorted < other ? other : sorted;
                sorted = keep_min ? lo : hi;
            }
        const bool live = sorted < 0x7FFF0000u;
        const int chain = (int)(sorted >> 6) & 0x1FF;
        const int before = __shfl_up(chain, 1);
        const bool starts_run = live && (lane == 0 || chain != before);
        const unsigned long long starts = __ballot(starts_run);
        const unsigned long long lives = __ballot(live);
        const int run_first = 63 - __clzll((long long)(starts & (~0ull >> (63 - lane)))); // highest start at or below me
        const unsigned long long above = lane == 63 ? 0ull : (starts >> (lane + 1)) << (lane + 1);
        const int run_end = above ? __ffsll(above) - 1 : __popcll(lives);
        uint32_t slot = 0;
        JLS_LOCKSTEP();
        if (live)
            slot = s_cnt[chain] + (uint32_t)(lane - run_first);
        JLS_LOCKSTEP();
        if (starts_run)
            s_cnt[chain] += (uint32_t)(run_end - run_first);
        JLS_LOCKSTEP();
        const uint32_t dest = (uint32_t)__builtin_amdgcn_ds_permute((int)(sorted & 63u) << 2, (int)slot);
        if (has)
        {
            // Samples of up to 15 bits leave bit 31 of the (x, Px) record free for the sign of the context: the regular
            // chains then read ONE array (bias_chains), and only the run chain (chain 0) needs the position.
            const bool packed = sign_fits_record(d) && (key & 0x1FF) != 0;
            w.sval[dest] = packed ? v | ((uint32_t)(key >> 9) << 31) : v;
            if (!packed)
                w.spos[dest] = (uint32_t)((size_t)y * width + x) | ((uint32_t)(key >> 9) << 31);
        }
        if (x < width)
            w.inv[(size_t)y * width + x] = has ? dest : kNoSlot;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C: one lane per (chain, scan).  Thread t codes chain t / scans of scan t % scans, so that the lanes of a wavefront
// work on the SAME context of different frames (similar chain lengths -> little idle time).

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

// The regular-mode chain of one lane (see bias_chains): `lines` cache lines of 16 records, in place.  kPacked: the sign of
// the context sits in bit 31 of the record (samples of up to 15 bits), otherwise in bit 31 of the parallel array `pin`.
// All lanes of the wavefront call this together (lanes without a regular chain with lines = 0).
//
// Memory: a lane reads its chain a whole cache line (64 bytes) per request, and the lanes of a wavefront belong to
// different frames, hundreds of MB apart: such a request takes about 1600 cycles alone and several times that while
// all chains of a pass -- or the wide stages of another pass -- are streaming (tools/microbench/stream_probe.hip), against
// about 1800 cycles of arithmetic per line.  So lines are requested several lines ahead, into register buffers that
// rotate by code position (no copies), and a line's results are stored after the wait for the next line.
typedef uint32_t u32x4 __attribute__((vector_size(16)));

template <bool kPacked>
JLS_DEV void walk_regular_chain(const JLS_GLOBAL_AS u32x4* vin, const JLS_GLOBAL_AS u32x4* pin, JLS_GLOBAL_AS u32x4* vout,
                                uint32_t lines, int reset, int bits, int maxval)
{
    struct Line
    {
        u32x4 v[4];
        u32x4 q[kPacked ? 1 : 4];
    };
    int b = 0, c = 0;
    // one event; n_before = N before it (wave-uniform), halve = N has reached RESET (wave-uniform, rare)
    auto event = [&](uint32_t v, uint32_t ps, int n_before, bool halve) -> uint32_t {
        const int sgn = ((int)(kPacked ? v : ps) >> 31) | 1;
        const int px_raw = kPacked ? (int)((v >> 16) & 0x7FFFu) : (int)(v >> 16);
        const int px = med3(mad24(c, sgn, px_raw), 0, maxval);                      // src/scan_encoder_core.hpp:57-67
        const int err = sign_extend(__mul24((int)(v & 0xFFFFu) - px, sgn), bits); // src/default_traits.hpp:123-139
        const uint32_t out = ((uint32_t)err << 1) | ((uint32_t)(2 * b + n_before - 1) >> 31);
        // A.13 as in the decoder: with t = B + Errval (halved at a reset) and N' the new N,
        // delta = (t > 0) - (t + N' <= 0), B' = median(t - delta * N', 1 - N', 0), C' = median(C + delta, -128, 127)
        int tb = b + err; // |B| < N + RANGE/2: cannot reach 2^24
        int n_new = n_before + 1;
        if (halve)
        {
            tb >>= 1;
            n_new = (n_before >> 1) + 1;
        }
        const int minus_delta = 1 - med3(tb, 0, 1) - med3(tb + n_new, 0, 1);
        b = med3(mad24(minus_delta, n_new, tb), 1 - n_new, 0);
        c = med3(c - minus_delta, -128, 127);
        return out;
    };
    int nn = 1; // wave-uniform
    auto four = [&](const u32x4& v, const u32x4& q) -> u32x4 {
        u32x4 o;
        if (reset == 0 || nn + 3 < reset)
        { // no RESET among these four
            o[0] = event(v[0], q[0], nn, false);
            o[1] = event(v[1], q[1], nn + 1, false);
            o[2] = event(v[2], q[2], nn + 2, false);
            o[3] = event(v[3], q[3], nn + 3, false);
            nn += 4;
        }
        else
        {
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const bool halve = nn == reset;
                o[j] = event(v[j], q[j], nn, halve);
                nn = (halve ? nn >> 1 : nn) + 1;
            }
        }
        return o;
    };
    // Unconditional: a lane past the end of its chain reads on into the next chain / the spare records behind the last one
    // (kChainSlack) and ignores what it gets.  Requests that depend on a condition would force every wait to be a wait for
    // everything outstanding -- it could no longer count on younger requests being there.
    auto request = [&](Line& x, uint32_t line) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            x.v[j] = vin[line * 4 + j];
            if (!kPacked)
                x.q[j] = pin[line * 4 + j];
        }
    };
    auto process = [&](const Line& x, u32x4 (&o)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = four(x.v[j], x.q[kPacked ? 0 : j]);
    };
    auto store = [&](const u32x4 (&o)[4], uint32_t line) {
        if (line < lines)
        {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                vout[line * 4 + j] = o[j];
        }
    };
    auto arrived = [&](const Line& x) {
#ifndef JLS_EMULATED
        if (kPacked)
            asm volatile("" ::"v"(x.v[0]), "v"(x.v[1]), "v"(x.v[2]), "v"(x.v[3]));
        else
            asm volatile("" ::"v"(x.v[0]), "v"(x.v[1]), "v"(x.v[2]), "v"(x.v[3]), "v"(x.q[0]), "v"(x.q[1]), "v"(x.q[2]), "v"(x.q[3]));
#endif
    };
    // kBuffers - 1 lines are in flight while one is processed; the loop body handles kBuffers lines so that every buffer
    // keeps its registers.
    constexpr int kBuffers = kPacked ? 5 : 3;
    Line buffer[kBuffers];
    u32x4 out[kBuffers][4];
#pragma unroll
    for (int k = 0; k < kBuffers; ++k)
    {
        if (kPacked)
            buffer[k].q[0] = u32x4{0, 0, 0, 0};
        request(buffer[k], (uint32_t)k);
    }
    arrived(buffer[0]);
    for (uint32_t line = 0; __any(line < lines); line += kBuffers)
    {
#pragma unroll
        for (int k = 0; k < kBuffers; ++k)
        {
            process(buffer[k], out[k]);                 // line + k
            arrived(buffer[(k + 1) % kBuffers]);        // line + k + 1 is here (requested kBuffers - 1 lines of arithmetic ago)
            store(out[k], line + (uint32_t)k);
            request(buffer[k], line + (uint32_t)(k + kBuffers));
        }
    }
}

// C0: what the run chain needs of the IMAGE, worked out for all run events at once.  In lossless mode the interruption
// sample of a run, its Ra and its Rb are source samples, so the type of the interruption (Ra == Rb or not) and its error
// value are pure functions of the image (src/scan_encoder_core.hpp:105-125); only RUNindex, the two run-interruption
// contexts and the code words are serial.  This kernel replaces the raster position in spos[e] of every run event of
// chain 0 by {Errval : 17 bits | RItype << 17 | (coded line mod components) << 18}, so that the lane that walks the run
// chain in bias_chains touches no pixel and does no division: three dependent global loads per event were what the whole
// stage waited for (55 000 run events of a 4096 x 4096 test frame at about 3 us each).  Planar and line-interleaved scans;
// sample-interleaved scans (several components per interruption) keep the position.  grid (64, scans) x 256 threads.
template <typename S, int ILV>
__global__ void __launch_bounds__(256) prepare_run_events(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    static_assert(ILV == 0 || ILV == 1, "one sample per interruption");
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const Traits t = make_traits(d);
    const uint32_t n = w.chain_total[0];
    const uint32_t* sval = w.sval + w.chain_base[0];
    uint32_t* spos = w.spos + w.chain_base[0];
    const uint32_t step = ILV == 1 ? line_step(d) : 1u;
    const int mask = (1 << d.bits_per_sample) - 1;
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < n; e += gridDim.x * 256u)
    {
        const uint32_t v = sval[e];
        const uint32_t p = spos[e] & 0x7FFFFFFFu;
        const uint32_t y = p / d.width; // coded line
        uint32_t packed = (y % step) << 18;
        if ((v >> 31) == 0)
        { // not an end-of-line run: the interruption sample follows the run
            const uint32_t xi = p - y * d.width + (v & 0x7FFFFFFFu);
            const int xv = load_sample<S, ILV>(d, y, xi, mask);
            const int ra = xi > 0 ? load_sample<S, ILV>(d, y, xi - 1, mask) : (y >= step ? load_sample<S, ILV>(d, y - step, 0, mask) : 0);
            const int rb = y >= step ? load_sample<S, ILV>(d, y - step, xi, mask) : 0;
            const int which = ra == rb ? 1 : 0;
            const int err = which ? error_value(t, xv - ra) : error_value(t, (xv - rb) * ((rb - ra) < 0 ? -1 : 1));
            packed |= ((uint32_t)err & 0x1FFFFu) | ((uint32_t)which << 17);
        }
        spos[e] = packed;
    }
}

template <typename S, int ILV>
__global__ void __launch_bounds__(64) bias_chains(const ScanDesc* __restrict__ descs, const Work* __restrict__ works,
                                                  uint32_t scans)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    // (no early exits: the regular-mode part below uses wavefront-wide votes, so every lane walks through it -- lanes
    // without a regular chain with zero cache lines to do)
    const bool exists = tid < scans * (uint32_t)kChains;
    const uint32_t chain = exists ? tid / scans : 1u;
    const uint32_t frame = exists ? tid % scans : 0u;
    const ScanDesc d = descs[frame];
    const Work w = works[frame];
    const Traits t = make_traits(d);
    const bool regular = exists && chain != 0 && chain != (uint32_t)kInterruptChain;
    const uint32_t n = w.chain_total[chain];
    JLS_GLOBAL_AS uint32_t* sval = (JLS_GLOBAL_AS uint32_t*)(w.sval + w.chain_base[chain]);
    const JLS_GLOBAL_AS uint32_t* spos = (const JLS_GLOBAL_AS uint32_t*)(w.spos + w.chain_base[chain]);

    { // ---- regular mode, serial half: src/scan_encoder_core.hpp:57-67, src/regular_mode_context.hpp:45-93
        // Of {A,B,C,N} only B and C feed back into the VALUE that is coded (through the bias-corrected prediction); N is
        // a function of the event index alone and A is a (periodically halved) running sum of |Errval| that only picks
        // the Golomb parameter.  The serial chain therefore carries just {B,C,N}: it turns each (x, Px, sign) record
        // into Errval plus the sign of 2B+N-1 (the k=0 error-correction condition, src/regular_mode_context.hpp:36-42),
        // in place.  A, k and the code words are then computed 64 events at a time by code_events.
        //
        // Memory: every chain starts on a multiple of kChainPad (= 16) records and the buffers carry kChainSlack spare
        // records, so a lane streams its chain with aligned 16-byte loads/stores (4 records) and no bounds logic: records
        // past the end of a chain are garbage that is processed into the chain's own padding.
        //
        // What a chain costs is what its lane issues per event (a lone wavefront issues an instruction every 4.3 - 5 cycles,
        // profiles/r02_microbench_latency.txt), and the pass waits for the longest chain.  The lanes of a wavefront walk
        // different frames in step, so N -- a function of the event index alone -- is the same in all of them: it lives in
        // scalar registers (no vector work for N, 1 - N or the RESET test), a group of four events without a RESET is
        // straight-line code, and a chain is read and written a whole cache line (16 records) per lane at a time, the
        // next line of both arrays in flight while the current one is processed.
        const int maxval = t.maxval;
        const int bpp = d.bits_per_sample; // RANGE = 2^bpp in lossless mode: modulo RANGE = sign extension of bpp bits
        const bool uniform_parameters = __all(t.reset == (int)uniform((uint32_t)t.reset) && bpp == (int)uniform((uint32_t)bpp)) != 0;
        const JLS_GLOBAL_AS u32x4* vin = (const JLS_GLOBAL_AS u32x4*)sval;
        const JLS_GLOBAL_AS u32x4* pin = (const JLS_GLOBAL_AS u32x4*)spos;
        JLS_GLOBAL_AS u32x4* vout = (JLS_GLOBAL_AS u32x4*)sval;
        int b = 0, c = 0;
        if (uniform_parameters)
        {
            const int reset = (int)uniform((uint32_t)t.reset);
            const int bits = (int)uniform((uint32_t)bpp);
            const uint32_t lines = regular ? (n + kChainPad - 1) / kChainPad : 0u; // cache lines of this lane's chain (the last one runs into its padding)
            if (bits <= 15)
                walk_regular_chain<true>(vin, pin, vout, lines, reset, bits, maxval);
            else
                walk_regular_chain<false>(vin, pin, vout, lines, reset, bits, maxval);
        }
        else
        { // frames with different RESET / precision in one wavefront (no caller of this library makes such batches): per lane
            int nn = 1;
            const int reset = t.reset;
            const int wrap = 32 - bpp;
            const bool packed = sign_fits_record(d);
            auto step = [&](uint32_t v, uint32_t ps) -> uint32_t {
                const int s = (int)(packed ? v : ps) >> 31; // 0 or -1
                const int px = med3((int)((v >> 16) & (packed ? 0x7FFFu : 0xFFFFu)) + ((c ^ s) - s), 0, maxval);
                int err = (((int)(v & 0xFFFFu) - px) ^ s) - s;
                err = (int)((uint32_t)err << wrap) >> wrap;
                const uint32_t out = ((uint32_t)err << 1) | ((uint32_t)(2 * b + nn - 1) >> 31);
                const int sh = nn == reset;
                const int tb = (b + err) >> sh;
                nn = (nn >> sh) + 1;
                const int minus_delta = 1 - med3(tb, 0, 1) - med3(tb + nn, 0, 1);
                b = med3(tb + __mul24(minus_delta, nn), 1 - nn, 0);
                c = med3(c - minus_delta, -128, 127);
                return out;
            };
            const uint32_t groups = regular ? (n + kChainPad - 1) / kChainPad * 4 : 0u;
            for (uint32_t g = 0; g < groups; ++g)
            {
                const u32x4 v = vin[g], q = pin[g];
                u32x4 o;
                o[0] = step(v[0], q[0]);
                o[1] = step(v[1], q[1]);
                o[2] = step(v[2], q[2]);
                o[3] = step(v[3], q[3]);
                vout[g] = o;
            }
        }
    }
    if (exists && chain == 0)
    { // ---- run mode: src/scan_encoder.hpp:53-73, src/scan_encoder_impl.hpp:249-275, src/scan_encoder_core.hpp:105-125
        // Codes go to the slot of the sample they belong to: the run-length code to the run's own slot, the code of the
        // interruption sample to the next slot of chain kInterruptChain (its events are these samples, in this order).
        // (two named records selected by value and the RUNindex values packed into one word: an indexed pair / array of
        // registers lives in scratch -- 48 bytes of private segment per lane until round 3)
        RunCtx rc0{0, initial_a(t), 1, 0}, rc1{1, initial_a(t), 1, 0};
        uint32_t run_indices = 0; // ILV_LINE: one RUNindex per component (src/scan_encoder_impl.hpp:126-137), 8 bits each
        const int mask = (1 << d.bits_per_sample) - 1;
        uint64_t* run_code = w.code + w.chain_base[0];
        uint8_t* run_len = w.len + w.chain_base[0];
        uint64_t* int_code = w.code + w.chain_base[kInterruptChain];
        uint8_t* int_len = w.len + w.chain_base[kInterruptChain];
        uint32_t interruptions = 0;
        for (uint32_t e = 0; e < n; ++e)
        {
            const uint32_t v = sval[e];
            const uint32_t p = spos[e] & (ILV == 2 ? 0x7FFFFFFFu : 0xFFFFFFFFu); // ILV != 2: prepare_run_events' record
            uint32_t run = v & 0x7FFFFFFFu;
            const bool eol = (v >> 31) != 0;
            uint32_t y = 0, x0 = 0;
            if (ILV == 2)
            {
                const uint32_t samples_per_line = line_samples(d);
                y = p / samples_per_line; // coded line
                x0 = (p - y * samples_per_line) / (uint32_t)d.components;
            }
            const uint32_t index_shift = ILV == 1 ? ((p >> 18) & 3u) * 8u : 0u;
            int run_index = (int)((run_indices >> index_shift) & 0xFFu);
            auto keep_run_index = [&] { run_indices = (run_indices & ~(0xFFu << index_shift)) | ((uint32_t)run_index << index_shift); };
            const uint32_t full = run;
            // run-length part: ones for every completed 2^J block, then either the end-of-line one or 0 + remainder
            uint64_t bits = 0;
            int len = 0;
            while (run >= (1u << run_j(run_index)))
            {
                bits = (bits << 1) | 1ull;
                ++len;
                run -= 1u << run_j(run_index);
                if (run_index < 31)
                    ++run_index;
            }
            if (eol)
            {
                if (run != 0)
                {
                    bits = (bits << 1) | 1ull;
                    ++len;
                }
                run_code[e] = bits;
                run_len[e] = (uint8_t)len;
                keep_run_index();
                continue;
            }
            const int jb = run_j(run_index);
            bits = (bits << (jb + 1)) | run;
            len += jb + 1;
            // run interruption sample at x0 + full
            const uint32_t xi = x0 + full;
            if (ILV == 2)
            { // every component against run context 0, in component order (src/scan_encoder_impl.hpp:222-247,
              // src/scan_encoder_core.hpp:127-138); component 0 of a run of length 0 shares the run's slot
                int xv[4], ra[4], rb[4] = {0, 0, 0, 0};
                load_pixel<S>(d, y, xi, mask, xv);
                if (xi > 0)
                    load_pixel<S>(d, y, xi - 1, mask, ra);
                else if (y > 0)
                    load_pixel<S>(d, y - 1, 0, mask, ra);
                else
                    ra[0] = ra[1] = ra[2] = ra[3] = 0;
                if (y > 0)
                    load_pixel<S>(d, y - 1, xi, mask, rb);
                for (int c = 0; c < d.components; ++c)
                {
                    const int sg = (rb[c] - ra[c]) < 0 ? -1 : 1;
                    const int err = error_value(t, (xv[c] - rb[c]) * sg);
                    RunCtx& ctx = rc0;
                    const int k = run_k(ctx);
                    const int map = run_map(ctx, err, k);
                    const int em = 2 * (err < 0 ? -err : err) - ctx.ritype - map;
                    const CodeWord cw = golomb_word(t, k, em, t.limit - jb - 1);
                    run_update(ctx, err, em, t.reset);
                    if (c == 0 && full == 0)
                    {
                        run_code[e] = (bits << cw.len) | cw.bits;
                        run_len[e] = (uint8_t)(len + cw.len);
                    }
                    else
                    {
                        if (c == 0)
                        {
                            run_code[e] = bits;
                            run_len[e] = (uint8_t)len;
                        }
                        int_code[interruptions] = cw.bits;
                        int_len[interruptions] = (uint8_t)cw.len;
                        ++interruptions;
                    }
                }
                if (run_index > 0)
                    --run_index;
                keep_run_index();
                continue;
            }
            // type and error value of the interruption come from prepare_run_events
            const int which = (int)((p >> 17) & 1u);
            const int err = (int)(p << 15) >> 15;
            RunCtx ctx = which ? rc1 : rc0;
            const int k = run_k(ctx);
            const int map = run_map(ctx, err, k);
            const int em = 2 * (err < 0 ? -err : err) - ctx.ritype - map;
            const CodeWord c = golomb_word(t, k, em, t.limit - jb - 1);
            run_update(ctx, err, em, t.reset);
            if (which)
                rc1 = ctx;
            else
                rc0 = ctx;
            if (run_index > 0)
                --run_index;
            keep_run_index();
            if (full == 0)
            { // both codes belong to the same sample: J+1 zero bits followed by the interruption code (<= LIMIT bits)
                run_code[e] = (bits << c.len) | c.bits;
                run_len[e] = (uint8_t)(len + c.len);
            }
            else
            {
                run_code[e] = bits;
                run_len[e] = (uint8_t)len;
                int_code[interruptions] = c.bits;
                int_len[interruptions] = (uint8_t)c.len;
                ++interruptions;
            }
        }
    }
}

// C2: grid (kRegularChains, scans), one wavefront per regular chain: 64 consecutive events of the chain per step.
//
// Before event i of a chain, N_i depends on i only (it counts 1..RESET, then cycles RESET/2+1..RESET) and
// A_i = A_seg + (sum of |Errval| since the last halving), where A_seg changes only at the events with N_i == RESET
// (src/regular_mode_context.hpp:45-63).  Inside a step the sums are a wavefront prefix sum; the halvings (two per 64
// events for RESET = 64) are walked in a short uniform loop.  Then k, the error correction and the limited-length Golomb
// word (src/regular_mode_context.hpp:99-136, src/scan_encoder_core.hpp:57-103) are independent per event.
__global__ void __launch_bounds__(64) code_events(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const Traits t = make_traits(d);
    const uint32_t chain = blockIdx.x < 364 ? blockIdx.x + 1 : (uint32_t)kZeroContextChain;
    const uint32_t n = w.chain_total[chain];
    if (n == 0)
        return;
    const JLS_GLOBAL_AS uint32_t* serr = (const JLS_GLOBAL_AS uint32_t*)(w.sval + w.chain_base[chain]);
    JLS_GLOBAL_AS uint64_t* code_out = (JLS_GLOBAL_AS uint64_t*)(w.code + w.chain_base[chain]);
    JLS_GLOBAL_AS uint8_t* len_out = (JLS_GLOBAL_AS uint8_t*)(w.len + w.chain_base[chain]);
    const uint32_t lane = threadIdx.x;
    const uint32_t reset = (uint32_t)t.reset;          // 0: N never matches (RESET = 256*m stored through uint8)
    const uint32_t half = reset >> 1;
    const uint32_t period = reset - half;              // events between two halvings once N cycles
    uint32_t a_seg = (uint32_t)initial_a(t);           // A before the first event of the current segment (uniform)
    uint32_t bad = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 64)
    {
        const uint32_t i = i0 + lane;
        const bool live = i < n;
        const uint32_t rec = live ? serr[i] : 0;
        const int err = (int)rec >> 1;
        const uint32_t mag = (uint32_t)(err < 0 ? -err : err);
        uint32_t incl = mag; // inclusive prefix sum of |Errval| over the lanes
        for (int delta = 1; delta < 64; delta <<= 1)
        {
            const uint32_t up = __shfl_up(incl, delta);
            if ((int)lane >= delta)
                incl += up;
        }
        // N before event i
        uint32_t n_i = i + 1;
        if (reset != 0 && i >= reset)
            n_i = half + 1 + (i - reset) % period;
        // walk the halving events inside [i0, i0+64): they are at i = reset-1 + j*period
        uint32_t my_a = a_seg, my_base = 0, seg_base = 0;
        if (reset != 0 && i0 + 64 > reset - 1)
        {
            uint32_t r = reset - 1;
            if (i0 > r)
                r += (i0 - r + period - 1) / period * period;
            for (; r < i0 + 64 && r < n; r += period)
            {
                const uint32_t at = (uint32_t)__shfl((int)incl, (int)(r - i0)); // sum up to and including the halving event
                a_seg = (a_seg + at - seg_base) >> 1;
                seg_base = at;
                if (i > r)
                {
                    my_a = a_seg;
                    my_base = at;
                }
            }
        }
        const uint32_t a_i = my_a + (incl - mag - my_base);
        bad |= (uint32_t)(live && a_i + mag >= (1u << 24)); // src/regular_mode_context.hpp:56-61
        if (live)
        {
            const RegCtx ctx{(int)a_i, 0, 0, (int)n_i};
            int k = regular_k(ctx);
            bad |= (uint32_t)(k >= 16);
            k = k > 15 ? 15 : k;
            const int corr = k == 0 ? -(int)(rec & 1u) : 0;
            const CodeWord c = golomb_word(t, k, map_error(corr ^ err), t.limit);
            code_out[i] = c.bits;
            len_out[i] = (uint8_t)c.len;
        }
        // A before the first event of the next step: everything after the last halving of this step
        a_seg += (uint32_t)__shfl((int)incl, 63) - seg_base;
    }
    if (bad)
        atomicOr(w.status, kStatusInvalid);
}

// ---------------------------------------------------------------------------------------------------------------
// D: grid (blocks, scans), 256 threads x 16 samples: concatenate the codes into the raw bit stream.  Where a block's
// bits start is the sum of the code lengths of all blocks before it; the blocks of a scan find that out among themselves
// while they run (a chained scan with look-back: every block publishes the bits of its own samples as soon as it has
// added them up, then the bits of everything up to and including itself once it knows its start; a block reads back
// over its predecessors until it meets one that already knows), so the code lengths and the slot map are read ONCE --
// a separate pass that summed the lengths per block read both a second time (197 MB of 2.1 GB per 4096 x 4096 frame).
// Workgroups start in the order of their index, x fastest: a block only ever waits for blocks that were started before it.
// blockbase[b]: bits 62..63 = state (0 nothing, 1 own bits, 2 bits up to and including b), bits 0..61 the value; zero
// before the launch.
// Zeroes the look-back states and the raw bit stream of every scan of a pass (they are contiguous in a work area): grid
// (any, scans) x 256 threads, 16 bytes per thread and step.
__global__ void __launch_bounds__(256) clear_pack_state(const Work* __restrict__ works, uint32_t bytes_per_scan)
{
    uint4* at = reinterpret_cast<uint4*>(works[blockIdx.y].blockbase);
    const uint32_t groups = bytes_per_scan / 16;
    for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < groups; g += gridDim.x * 256u)
        at[g] = make_uint4(0, 0, 0, 0);
}

constexpr uint64_t kBlockOwn = 1ull << 62, kBlockUpTo = 2ull << 62, kBlockValue = (1ull << 62) - 1ull;

__global__ void __launch_bounds__(256) write_raw_bits(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    __shared__ uint32_t s_scan[256];
    __shared__ uint64_t s_start;
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const uint64_t total = (uint64_t)line_samples(d) * coded_lines(d);
    const uint64_t base = (uint64_t)blockIdx.x * kPackBlock + (uint64_t)threadIdx.x * 16;
    int lens[16];
    uint32_t slots[16];
    uint32_t sum = 0;
    for (int i = 0; i < 16; ++i)
        slots[i] = base + i < total ? w.inv[base + i] : kNoSlot;
    for (int i = 0; i < 16; ++i)
    {
        lens[i] = slots[i] != kNoSlot ? w.len[slots[i]] : 0;
        sum += (uint32_t)lens[i];
    }
    s_scan[threadIdx.x] = sum;
    __syncthreads();
    for (int stride = 1; stride < 256; stride <<= 1) // Hillis-Steele inclusive scan
    {
        const uint32_t add = (int)threadIdx.x >= stride ? s_scan[threadIdx.x - stride] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += add;
        __syncthreads();
    }
    // ---- where this block starts: the first wavefront looks back, 64 predecessors at a time
    if (threadIdx.x < 64)
    {
        const int lane = threadIdx.x;
        const uint64_t own = s_scan[255];
        const uint32_t b = blockIdx.x;
        if (lane == 0)
            store_relaxed(&w.blockbase[b], (b == 0 ? kBlockUpTo : kBlockOwn) | own);
        uint64_t start = 0;
        uint32_t reach = b; // blocks [reach, b) are accounted for in `start`
        while (reach > 0)
        {
            const bool mine = (uint32_t)lane < reach;
            const uint32_t j = mine ? reach - 1 - (uint32_t)lane : 0;
            uint64_t state = 0;
            do
            { // every predecessor in the window has published something
                state = mine ? load_relaxed(&w.blockbase[j]) : kBlockOwn;
            } while (__any((state >> 62) == 0));
            const unsigned long long knows = __ballot(mine && (state >> 62) == 2);
            const int last = knows ? (int)__ffsll(knows) - 1 : 63; // the nearest predecessor that knows its total
            uint64_t part = mine && lane <= last ? state & kBlockValue : 0;
            for (int delta = 32; delta > 0; delta >>= 1)
                part += __shfl_xor(part, delta);
            start += part;
            if (knows)
                break;
            reach = reach > 64 ? reach - 64 : 0;
        }
        if (lane == 0)
        {
            if (b != 0)
                store_relaxed(&w.blockbase[b], kBlockUpTo | (start + own));
            s_start = start;
            if ((uint64_t)(b + 1) * kPackBlock >= total)
                *w.total_bits = start + own; // the last block of the scan
        }
    }
    __syncthreads();
    uint64_t bitpos = s_start + s_scan[threadIdx.x] - sum;
    if (sum == 0)
        return;
    uint64_t word = bitpos >> 5;
    const uint64_t first_word = word;
    int acc_bits = (int)(bitpos & 31); // the leading bits of the first word belong to the previous thread
    uint64_t acc = 0;
    for (int i = 0; i < 16; ++i)
    {
        int left = lens[i];
        if (left == 0)
            continue;
        const uint64_t v = w.code[slots[i]];
        while (left > 0)
        {
            const int room = 64 - acc_bits;
            const int n = left < room ? left : room;
            const uint64_t piece = n == 64 ? v : ((v >> (left - n)) & ((1ull << n) - 1ull));
            acc |= piece << (room - n);
            acc_bits += n;
            left -= n;
            while (acc_bits >= 32)
            {
                const uint32_t out = __builtin_bswap32((uint32_t)(acc >> 32));
                if (word < w.raw_words)
                {
                    if (word == first_word)
                        atomicOr(&w.raw[word], out); // shared with the previous thread's tail
                    else
                        w.raw[word] = out;           // entirely ours
                }
                acc <<= 32;
                acc_bits -= 32;
                ++word;
            }
        }
    }
    if (acc_bits > 0 && word < w.raw_words)
        atomicOr(&w.raw[word], __builtin_bswap32((uint32_t)(acc >> 32))); // tail shared with the next thread
}

// ---------------------------------------------------------------------------------------------------------------
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
