// scan_serial.hip -- one-wavefront-per-scan JPEG-LS kernels for gfx950 (every coding mode the reference supports).
//
// A scan is an inherently serial object in two cases: decoding (symbol i+1 cannot be located before symbol i is
// decoded, reference src/scan_decoder_core.hpp:38-69) and near-lossless / interleaved encoding (the causal template
// holds RECONSTRUCTED samples, src/scan_encoder_impl.hpp:130-134).  For those the engine's parallelism is across scans
// (frames, components of ILV_NONE frames, GPUs), so the unit of work here is one 64-lane wavefront per scan:
//
//   * all 64 lanes move pixels: coalesced row loads/stores between the user's layout and the two-line causal window,
//     including de-interleaving, bit masking and the HP1..HP3 colour transforms (src/copy_{to,from}_line_buffer.hpp),
//   * lane 0 walks the entropy chain of the line: context id, MED prediction, Golomb-Rice code, A/B/C/N update, run
//     mode, bit stuffing (src/scan_encoder_impl.hpp:109-302, src/scan_decoder_impl.hpp:132-337),
//   * the 365 regular + 2 run-interruption contexts live in LDS (5.9 KB per wavefront).
//
// The lossless single-component encoder has a far more parallel formulation (tile_pipeline.hip); this file is the
// general path and the decoder.  Results are bit-exact with the reference, including the error codes of appendix D.
#pragma once
#include <hip/hip_runtime.h>

#include "scan_model.h"

namespace jls {

namespace {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------------------------------
// Bit writer: the reference's 32-bit accumulator and flush policy (src/scan_encoder.hpp:75-186), kept identical so
// that destination_too_small is raised for exactly the same destination sizes.
struct BitWriter
{
    uint8_t* pos;
    uint64_t remaining;
    uint64_t written;
    uint32_t buf;
    int free_bits;
    bool ff;
    uint32_t err;

    JLS_DEV void init(uint8_t* dst, uint64_t capacity)
    {
        pos = dst;
        remaining = capacity;
        written = 0;
        buf = 0;
        free_bits = 32;
        ff = false;
        err = kOk;
    }

    JLS_DEV void flush()
    {
        if (remaining < 4)
        {
            err = kDestinationTooSmall;
            free_bits = free_bits < 0 ? 0 : free_bits; // keep the state sane; the scan is abandoned
            return;
        }
        for (int i = 0; i < 4; ++i)
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
            *pos++ = (uint8_t)v;
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

// ---------------------------------------------------------------------------------------------------------------
// Bit reader: the reference's 64-bit cache with its refill / marker rules (src/scan_decoder.hpp:250-322).
struct BitReader
{
    const uint8_t* pos;
    const uint8_t* end;
    uint64_t cache;
    int valid;
    uint32_t restart_counter;
    uint32_t err;

    JLS_DEV void fill()
    {
        if (err)
            return;
        // The reference's fast refill (src/scan_decoder.hpp:286-308): taken exactly when none of the next 8 bytes is
        // 0xFF.  It may load up to 64 bits where the byte loop stops at 56..63, which is observable through the read
        // position at the end of a scan, so it is restated rather than treated as an optimisation.
        if (end - pos >= 8 && valid >= 0)
        {
            uint64_t v = 0;
            bool any_ff = false;
            for (int i = 0; i < 8; ++i)
            {
                const uint64_t b = pos[i];
                any_ff = any_ff || b == 0xFFu;
                v = (v << 8) | b;
            }
            if (!any_ff)
            {
                cache |= v >> valid;
                const int consumed = (64 - valid) / 8;
                pos += consumed;
                valid += consumed * 8;
                return;
            }
        }
        do
        {
            if (pos >= end)
            {
                if (valid <= 0)
                    err = kInvalidData;
                return;
            }
            const uint64_t b = *pos;
            if (b == 0xFFu && (pos == end - 1 || (pos[1] & 0x80u) != 0))
            {
                if (valid <= 0)
                    err = kInvalidData;
                return;
            }
            const int shift = 56 - valid;
            if (shift < 64)
                cache |= b << shift;
            valid += 8;
            ++pos;
            if (b == 0xFFu)
                --valid;
        } while (valid < 56);
    }

    JLS_DEV void init(const uint8_t* src, uint64_t size)
    {
        pos = src;
        end = src + size;
        cache = 0;
        valid = 0;
        restart_counter = 0;
        err = kOk;
        fill();
    }

    JLS_DEV void skip(int n)
    {
        valid -= n;
        cache = n >= 64 ? 0 : (cache << n);
    }

    JLS_DEV int value(int n) // read_value, src/scan_decoder.hpp:127-142
    {
        if (valid < n)
        {
            fill();
            if (!err && valid < n)
                err = kInvalidData;
            if (err)
                return 0;
        }
        const int v = (int)(cache >> (64 - n));
        skip(n);
        return v;
    }

    JLS_DEV unsigned peek_byte()
    {
        if (valid < 8)
            fill();
        return (unsigned)(cache >> 56);
    }

    JLS_DEV int bit()
    {
        if (valid <= 0)
            fill();
        const int b = (int)(cache >> 63);
        skip(1);
        return b;
    }

    JLS_DEV int unary() // read_unary_code, src/scan_decoder.hpp:176-217
    {
        if (valid < 16)
            fill();
        const int count = cache == 0 ? 64 : __clzll((long long)cache);
        if (count < 16)
        {
            skip(count + 1);
            return count;
        }
        skip(15);
        for (int zeros = 15;; ++zeros)
        {
            if (err)
                return 0;
            if (bit())
                return zeros;
        }
    }

    JLS_DEV int golomb(const Traits& t, int k, int limit) // decode_mapped_error_value, src/scan_decoder.hpp:113-125
    {
        const int u = unary();
        if (u < limit - t.qbpp - 1)
            return k == 0 ? u : (u << k) + value(k);
        return value(t.qbpp) + 1;
    }

    JLS_DEV void end_scan() // src/scan_decoder.hpp:71-89
    {
        if (err)
            return;
        if (pos >= end)
        {
            err = kNeedMoreData;
            return;
        }
        if (*pos != 0xFFu)
        {
            (void)bit();
            if (err)
                return;
            if (pos >= end)
            {
                err = kNeedMoreData;
                return;
            }
            if (*pos != 0xFFu)
            {
                err = kInvalidData;
                return;
            }
        }
        if (cache != 0)
            err = kInvalidData;
    }

    JLS_DEV const uint8_t* actual_position() const // src/scan_decoder.hpp:92-107
    {
        int v = valid;
        const uint8_t* p = pos;
        for (;;)
        {
            const int last = p[-1] == 0xFFu ? 7 : 8;
            if (v < last)
                return p;
            v -= last;
            --p;
        }
    }

    JLS_DEV void restart_marker() // src/scan_decoder.hpp:237-243,335-349
    {
        if (err)
            return;
        const uint32_t expected = 0xD0u + restart_counter;
        if (pos == end)
        {
            err = kNeedMoreData;
            return;
        }
        uint32_t v = *pos++;
        if (v != 0xFFu)
        {
            err = kRestartMarkerNotFound;
            return;
        }
        do
        {
            if (pos == end)
            {
                err = kNeedMoreData;
                return;
            }
            v = *pos++;
        } while (v == 0xFFu);
        if (v != expected)
        {
            err = kRestartMarkerNotFound;
            return;
        }
        restart_counter = (restart_counter + 1) & 7u;
        valid = 0;
        cache = 0;
        fill();
    }
};

// ---------------------------------------------------------------------------------------------------------------
struct Model
{
    RegCtx* reg; // [365] in LDS
    RunCtx* run; // [2]   in LDS
    int run_index;
};

JLS_DEV void init_model(const Traits& t, Model& m, int lane)
{
    const int a0 = initial_a(t);
    for (int i = lane; i < 365; i += kWave)
        m.reg[i] = RegCtx{a0, 0, 0, 1};
    if (lane < 2)
        m.run[lane] = RunCtx{lane, a0, 1, 0};
}

// Regular mode, encoder: src/scan_encoder_core.hpp:40-67.  Returns Rx.
JLS_DEV int encode_regular(const Traits& t, Model& m, BitWriter& bw, int qs, int x, int pred)
{
    const int s = qs >> 31;
    RegCtx ctx = m.reg[(qs ^ s) - s];
    const int k = regular_k(ctx);
    if (k >= 16)
    {
        bw.err = kInvalidData;
        return x;
    }
    const int px = clamp_sample(t, pred + ((ctx.c ^ s) - s));
    const int e = error_value(t, ((x - px) ^ s) - s);
    bw.golomb(t, k, map_error(error_correction(ctx, k | t.near) ^ e), t.limit);
    if (!regular_update(ctx, e, t.near, t.reset))
        bw.err = kInvalidData;
    m.reg[(qs ^ s) - s] = ctx;
    return reconstruct(t, px, (e ^ s) - s);
}

// Regular mode, decoder: src/scan_decoder_core.hpp:38-69 (golomb_lut hit <=> whole code within the first 8 bits).
JLS_DEV int decode_regular(const Traits& t, Model& m, BitReader& br, int qs, int pred)
{
    const int s = qs >> 31;
    RegCtx ctx = m.reg[(qs ^ s) - s];
    const int px = clamp_sample(t, pred + ((ctx.c ^ s) - s));
    const int k = regular_k(ctx);
    if (k >= 16)
    {
        br.err = kInvalidData;
        return 0;
    }
    int e;
    const unsigned top = br.peek_byte();
    const int u = top == 0 ? 8 : (__clz((int)top) - 24);
    if (u + 1 + k <= 8)
    {
        const int mm = (u << k) | (int)((top >> (8 - u - 1 - k)) & ((1u << k) - 1u));
        br.skip(u + 1 + k);
        e = unmap_error(mm);
    }
    else
    {
        e = unmap_error(br.golomb(t, k, t.limit));
        if (e > 65535 || e < -65535)
            br.err = kInvalidData;
    }
    if (br.err)
        return 0;
    if (k == 0)
        e ^= error_correction(ctx, t.near);
    if (!regular_update(ctx, e, t.near, t.reset))
        br.err = kInvalidData;
    m.reg[(qs ^ s) - s] = ctx;
    return reconstruct(t, px, (e ^ s) - s);
}

JLS_DEV void encode_run_error(const Traits& t, Model& m, BitWriter& bw, int which, int e) // src/scan_encoder_core.hpp:105-116
{
    RunCtx ctx = m.run[which];
    const int k = run_k(ctx);
    const int map = run_map(ctx, e, k);
    const int em = 2 * (e < 0 ? -e : e) - ctx.ritype - map;
    bw.golomb(t, k, em, t.limit - run_j(m.run_index) - 1);
    run_update(ctx, e, em, t.reset);
    m.run[which] = ctx;
}

JLS_DEV int decode_run_error(const Traits& t, Model& m, BitReader& br, int which) // src/scan_decoder_core.hpp:72-81
{
    RunCtx ctx = m.run[which];
    const int k = run_k(ctx);
    if (k > 32)
    {
        br.err = kInvalidData;
        return 0;
    }
    const int em = br.golomb(t, k, t.limit - run_j(m.run_index) - 1);
    if (br.err)
        return 0;
    const int e = run_error_value(ctx, em + ctx.ritype, k);
    run_update(ctx, e, em, t.reset);
    m.run[which] = ctx;
    return e;
}

// One line of `nc` co-sited components (nc > 1 only in ILV_SAMPLE).  Lane 0 only.
template <bool kDecode>
JLS_DEV void code_line(const Traits& t, Model& m, BitWriter& bw, BitReader& br, uint16_t* prev, uint16_t* cur, int nc,
                       size_t plane_stride, uint32_t width)
{
    uint32_t i = 1;
    while (i <= width)
    {
        if ((kDecode ? br.err : bw.err) != kOk)
            return;
        int qs[4];
        bool all_zero = true;
        for (int j = 0; j < nc; ++j)
        {
            const uint16_t* p = prev + j * plane_stride;
            const uint16_t* q = cur + j * plane_stride;
            qs[j] = context_id(t, q[i - 1], p[i], p[i - 1], p[i + 1]);
            all_zero = all_zero && qs[j] == 0;
        }
        if (!all_zero)
        {
            for (int j = 0; j < nc; ++j)
            {
                const uint16_t* p = prev + j * plane_stride;
                uint16_t* q = cur + j * plane_stride;
                const int pred = med_predict(q[i - 1], p[i], p[i - 1]);
                q[i] = (uint16_t)(kDecode ? decode_regular(t, m, br, qs[j], pred) : encode_regular(t, m, bw, qs[j], q[i], pred));
            }
            ++i;
            continue;
        }
        // ---- run mode: src/scan_encoder_impl.hpp:249-275, src/scan_decoder_impl.hpp:264-337
        const uint32_t remaining = width - (i - 1);
        uint32_t run = 0;
        if (kDecode)
        {
            while (br.bit())
            {
                if (br.err)
                    return;
                const uint32_t block = 1u << run_j(m.run_index);
                const uint32_t count = block < remaining - run ? block : remaining - run;
                run += count;
                if (count == block && m.run_index < 31)
                    ++m.run_index;
                if (run == remaining)
                    break;
            }
            if (br.err)
                return;
            if (run != remaining)
            {
                const int jb = run_j(m.run_index);
                run += jb > 0 ? (uint32_t)br.value(jb) : 0u;
            }
            if (br.err)
                return;
            if (run > remaining)
            {
                br.err = kInvalidData;
                return;
            }
            for (int j = 0; j < nc; ++j)
            {
                uint16_t* q = cur + j * plane_stride;
                const uint16_t ra = q[i - 1];
                for (uint32_t r = 0; r < run; ++r)
                    q[i + r] = ra;
            }
        }
        else
        {
            for (;;)
            {
                bool near_all = true;
                for (int j = 0; j < nc; ++j)
                {
                    const uint16_t* q = cur + j * plane_stride;
                    near_all = near_all && is_near(t, q[i + run], q[i - 1]);
                }
                if (!near_all)
                    break;
                for (int j = 0; j < nc; ++j)
                {
                    uint16_t* q = cur + j * plane_stride;
                    q[i + run] = q[i - 1];
                }
                if (++run == remaining)
                    break;
            }
            // encode_run_pixels, src/scan_encoder.hpp:53-73
            uint32_t left = run;
            while (left >= (1u << run_j(m.run_index)))
            {
                bw.append(1, 1);
                left -= 1u << run_j(m.run_index);
                if (m.run_index < 31)
                    ++m.run_index;
            }
            if (run == remaining)
            {
                if (left != 0)
                    bw.append(1, 1);
            }
            else
                bw.append(left, run_j(m.run_index) + 1);
        }
        if (run == remaining)
            return;
        // ---- run interruption sample
        const uint32_t at = i + run;
        for (int j = 0; j < nc; ++j)
        {
            const uint16_t* p = prev + j * plane_stride;
            uint16_t* q = cur + j * plane_stride;
            const int ra = q[i - 1];
            const int rb = p[at];
            int rx;
            if (nc == 1 && is_near(t, ra, rb))
            {
                int e;
                if (kDecode)
                    e = decode_run_error(t, m, br, 1);
                else
                {
                    e = error_value(t, q[at] - ra);
                    encode_run_error(t, m, bw, 1, e);
                }
                rx = reconstruct(t, ra, e);
            }
            else
            {
                const int sg = (rb - ra) < 0 ? -1 : 1;
                int e;
                if (kDecode)
                    e = decode_run_error(t, m, br, 0);
                else
                {
                    e = error_value(t, (q[at] - rb) * sg);
                    encode_run_error(t, m, bw, 0, e);
                }
                rx = reconstruct(t, rb, e * sg);
            }
            q[at] = (uint16_t)rx;
        }
        if (m.run_index > 0)
            --m.run_index;
        i = at + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Pixel movement by the whole wavefront.

JLS_DEV unsigned load_sample(const uint8_t* p, bool wide)
{
    return wide ? (unsigned)p[0] | ((unsigned)p[1] << 8) : (unsigned)p[0];
}

JLS_DEV void store_sample(uint8_t* p, bool wide, unsigned v)
{
    p[0] = (uint8_t)v;
    if (wide)
        p[1] = (uint8_t)(v >> 8);
}

// src/copy_to_line_buffer.hpp:21-262
JLS_DEV void row_to_window(const ScanDesc& d, const uint8_t* row, uint16_t* cur, size_t plane_stride, int lane)
{
    const bool wide = d.bits_per_sample > 8;
    const int bytes = wide ? 2 : 1;
    const unsigned mask = (1u << d.bits_per_sample) - 1u;
    if (d.interleave_mode == 0)
    {
        const bool need_mask = d.bits_per_sample != bytes * 8;
        for (uint32_t i = lane; i < d.width; i += kWave)
        {
            const unsigned v = load_sample(row + (size_t)i * bytes, wide);
            cur[1 + i] = (uint16_t)(need_mask ? (v & mask) : v);
        }
        return;
    }
    const int nc = d.components;
    for (uint32_t i = lane; i < d.width; i += kWave)
    {
        unsigned v[4];
        for (int j = 0; j < nc; ++j)
            v[j] = load_sample(row + ((size_t)i * nc + j) * bytes, wide);
        if (d.color_transformation != 0 && nc == 3)
            hp_forward(d.color_transformation, wide, (int)v[0], (int)v[1], (int)v[2], v);
        else
            for (int j = 0; j < nc; ++j)
                v[j] &= mask;
        for (int j = 0; j < nc; ++j)
            cur[j * plane_stride + 1 + i] = (uint16_t)v[j];
    }
}

// src/copy_from_line_buffer.hpp:19-191
JLS_DEV void window_to_row(const ScanDesc& d, const uint16_t* cur, size_t plane_stride, uint8_t* row, int lane)
{
    const bool wide = d.bits_per_sample > 8;
    const int bytes = wide ? 2 : 1;
    if (d.interleave_mode == 0)
    {
        for (uint32_t i = lane; i < d.width; i += kWave)
            store_sample(row + (size_t)i * bytes, wide, cur[1 + i]);
        return;
    }
    const int nc = d.components;
    for (uint32_t i = lane; i < d.width; i += kWave)
    {
        unsigned v[4];
        for (int j = 0; j < nc; ++j)
            v[j] = cur[j * plane_stride + 1 + i];
        if (d.color_transformation != 0 && nc == 3)
            hp_inverse(d.color_transformation, wide, (int)v[0], (int)v[1], (int)v[2], v);
        for (int j = 0; j < nc; ++j)
            store_sample(row + ((size_t)i * nc + j) * bytes, wide, v[j]);
    }
}

JLS_DEV void zero_window(uint16_t* lines, size_t count, int lane)
{
    for (size_t i = lane; i < count; i += kWave)
        lines[i] = 0;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
// grid = number of scans, block = one wavefront.
__global__ void __launch_bounds__(64) encode_scans_serial(const ScanDesc* __restrict__ descs, ScanResult* __restrict__ results)
{
    __shared__ RegCtx s_reg[365];
    __shared__ RunCtx s_run[2];
    const int lane = threadIdx.x;
    const ScanDesc d = descs[blockIdx.x];
    const Traits t = make_traits(d);
    Model m{s_reg, s_run, 0};
    init_model(t, m, lane);

    const size_t ps = (size_t)d.width + 2;
    const int planes = d.interleave_mode == 0 ? 1 : d.components;
    zero_window(d.line_scratch, 2 * (size_t)planes * ps, lane);
    __syncthreads();

    BitWriter bw;
    BitReader br_unused{};
    bw.init(d.stream, d.stream_capacity);
    int run_index[4] = {0, 0, 0, 0};

    for (uint32_t line = 0; line < d.height; ++line)
    {
        uint16_t* prev = d.line_scratch + ((line & 1) ? (size_t)planes * ps : 0);
        uint16_t* cur = d.line_scratch + ((line & 1) ? 0 : (size_t)planes * ps);
        row_to_window(d, d.pixels + (size_t)line * d.pixel_stride, cur, ps, lane);
        __syncthreads();
        if (lane == 0 && bw.err == kOk)
        {
            if (d.interleave_mode == 2)
            {
                for (int j = 0; j < d.components; ++j)
                { // initialize_edge_pixels on whole pixels, src/scan_codec.hpp:189-195
                    prev[j * ps + d.width + 1] = prev[j * ps + d.width];
                    cur[j * ps] = prev[j * ps + 1];
                }
                m.run_index = run_index[0];
                code_line<false>(t, m, bw, br_unused, prev, cur, d.components, ps, d.width);
                run_index[0] = m.run_index;
            }
            else
            {
                for (int j = 0; j < planes; ++j)
                {
                    uint16_t* p = prev + j * ps;
                    uint16_t* q = cur + j * ps;
                    p[d.width + 1] = p[d.width];
                    q[0] = p[1];
                    m.run_index = run_index[j];
                    code_line<false>(t, m, bw, br_unused, p, q, 1, ps, d.width);
                    run_index[j] = m.run_index;
                }
            }
        }
        __syncthreads();
    }
    if (lane == 0)
    {
        bw.end_scan();
        ScanResult r;
        r.errc = bw.err;
        r.flags = 0;
        r.bytes = bw.written;
        results[blockIdx.x] = r;
    }
}

__global__ void __launch_bounds__(64) decode_scans_serial(const ScanDesc* __restrict__ descs, ScanResult* __restrict__ results)
{
    __shared__ RegCtx s_reg[365];
    __shared__ RunCtx s_run[2];
    __shared__ uint32_t s_err;
    const int lane = threadIdx.x;
    const ScanDesc d = descs[blockIdx.x];
    const Traits t = make_traits(d);
    Model m{s_reg, s_run, 0};
    init_model(t, m, lane);

    const size_t ps = (size_t)d.width + 2;
    const int planes = d.interleave_mode == 0 ? 1 : d.components;
    zero_window(d.line_scratch, 2 * (size_t)planes * ps, lane);
    if (lane == 0)
        s_err = kOk;
    __syncthreads();

    BitWriter bw_unused{};
    BitReader br{};
    if (lane == 0)
    {
        br.init(d.stream, d.stream_capacity);
        s_err = br.err;
    }
    int run_index[4] = {0, 0, 0, 0};
    const uint32_t interval = d.restart_interval == 0 ? d.height : d.restart_interval;
    uint32_t lines_left_in_interval = interval;
    __syncthreads();
    uint32_t err = s_err; // every lane reads the published status between two barriers, lane 0 writes it outside
    __syncthreads();

    for (uint32_t line = 0; line < d.height && err == kOk; ++line)
    {
        uint16_t* prev = d.line_scratch + ((line & 1) ? (size_t)planes * ps : 0);
        uint16_t* cur = d.line_scratch + ((line & 1) ? 0 : (size_t)planes * ps);
        if (lane == 0)
        {
            if (d.interleave_mode == 2)
            {
                for (int j = 0; j < d.components; ++j)
                {
                    prev[j * ps + d.width + 1] = prev[j * ps + d.width];
                    cur[j * ps] = prev[j * ps + 1];
                }
                m.run_index = run_index[0];
                code_line<true>(t, m, bw_unused, br, prev, cur, d.components, ps, d.width);
                run_index[0] = m.run_index;
            }
            else
            {
                for (int j = 0; j < planes; ++j)
                {
                    uint16_t* p = prev + j * ps;
                    uint16_t* q = cur + j * ps;
                    p[d.width + 1] = p[d.width];
                    q[0] = p[1];
                    m.run_index = run_index[j];
                    code_line<true>(t, m, bw_unused, br, p, q, 1, ps, d.width);
                    run_index[j] = m.run_index;
                }
            }
            s_err = br.err;
        }
        __syncthreads();
        err = s_err;
        if (err == kOk)
            window_to_row(d, cur, ps, d.pixels + (size_t)line * d.pixel_stride, lane);
        // restart interval boundary: src/scan_decoder_impl.hpp:119-127
        const bool boundary = (--lines_left_in_interval == 0) && (line + 1 < d.height) && err == kOk;
        __syncthreads();
        if (boundary)
        {
            lines_left_in_interval = interval;
            if (lane == 0)
            {
                br.restart_marker(); // a failure is latched in br.err and published by the next line
                for (int j = 0; j < 4; ++j)
                    run_index[j] = 0;
            }
            zero_window(d.line_scratch, 2 * (size_t)planes * ps, lane);
            init_model(t, m, lane);
            __syncthreads();
        }
    }
    if (lane == 0)
    {
        if (br.err == kOk)
            br.end_scan();
        ScanResult r;
        r.errc = br.err;
        r.flags = 0;
        r.bytes = br.err == kOk ? (uint64_t)(br.actual_position() - d.stream) : 0;
        results[blockIdx.x] = r;
    }
}

} // namespace jls
