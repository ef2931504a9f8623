// scan_wave_decode.hip -- the production JPEG-LS scan decoder for gfx950: one wavefront per scan, wave-uniform control.
//
// Decoding a scan is one serial dependency chain (bit position -> Golomb parameter k -> context statistics -> previous
// reconstructed sample; reference src/scan_decoder_core.hpp:38-69), so the chip is filled with MANY scans (frames,
// components, restart intervals), one 64-lane wavefront each, and the per-sample chain is kept as short as the ISA
// allows:
//   * everything the chain touches lives on-chip: the packed context table (365 x 8 B), the run-interruption contexts,
//     ONE line of samples per plane (the causal window is kept in registers: Ra/Rb/Rc/Rd slide, a decoded sample
//     overwrites the slot of the sample above it once that one has been read), and a 2 KB ring of the bitstream -- all
//     in LDS (~7 KB + one line per wavefront, so 16+ wavefronts fit a CU's 160 KB);
//   * control flow is wave-uniform (all lanes compute the same scalar chain), which lets the whole wavefront do the
//     bulk memory work in-line: 1 KB coalesced bitstream refills (16 B per lane), run fills, and the conversion of each
//     finished line to the user's layout (de-interleave, inverse HP1..HP3) with coalesced row stores;
//   * the bit reader keeps the reference's exact refill and marker rules (src/scan_decoder.hpp:250-322) so that bytes
//     consumed and error codes match on valid, truncated and corrupt streams alike.
// decode_scans_serial (scan_serial.hip) remains the fallback for scans whose line does not fit LDS or whose RESET makes
// N exceed 8 bits.
#pragma once
#include <hip/hip_runtime.h>

#include "scan_model.h"

namespace jls {
namespace wave {

constexpr int kLanes = 64;
constexpr uint32_t kRing = 2048; // bytes of bitstream resident in LDS (power of two)
constexpr uint32_t kChunk = 1024; // one cooperative refill: 64 lanes x 16 B
constexpr uint32_t kCtxBytes = 2928; // 365 packed contexts, padded to 16
constexpr uint32_t kRunBytes = 32;
constexpr uint32_t kFixedLds = kCtxBytes + kRunBytes + kRing;

// {A : 32 | -B : 8 | C : 8 | N : 16}.  After A.13 the bias accumulator satisfies -N < B <= 0 and N <= RESET <= 255
// (the reference stores RESET through a uint8_t, src/scan_codec.hpp:142), so 8 bits of magnitude are exact.
struct PackedCtx
{
    uint32_t a;
    uint32_t bcn;
};

JLS_DEV RegCtx unpack(const PackedCtx p)
{
    RegCtx x;
    x.a = (int)p.a;
    x.b = -(int)(p.bcn & 0xFFu);
    x.c = (int)(signed char)((p.bcn >> 8) & 0xFFu);
    x.n = (int)(p.bcn >> 16);
    return x;
}

JLS_DEV PackedCtx pack(const RegCtx& x)
{
    PackedCtx p;
    p.a = (uint32_t)x.a;
    p.bcn = (uint32_t)(-x.b) | (((uint32_t)x.c & 0xFFu) << 8) | ((uint32_t)x.n << 16);
    return p;
}

// ---------------------------------------------------------------------------------------------------------------
// Bit reader over an LDS ring that mirrors the global bitstream.  Addresses are kept in "u" units: u = stream offset +
// misalignment of the stream pointer, so that every refill is a 16-byte aligned global load per lane.
struct RingReader
{
    const uint8_t* gbase; // 16-byte aligned: stream - misalign
    uint8_t* ring;
    uint64_t u_pos;    // next unread byte
    uint64_t u_end;    // one past the last byte of the source
    uint64_t u_loaded; // ring holds [max(u_begin, u_loaded - kRing), u_loaded)
    uint64_t u_begin;
    uint64_t cache;
    int valid;
    uint32_t restart_counter;
    uint32_t err;
    int lane;

    JLS_DEV uint32_t byte_at(uint64_t u) const { return ring[u & (kRing - 1)]; }

    // Keeps >= 512 unread bytes (or everything up to the end) resident.  Wave-uniform.
    JLS_DEV void ensure()
    {
        while (u_loaded < u_end && u_loaded < u_pos + 512)
        {
            const uint64_t u = u_loaded + (uint64_t)lane * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (u < u_end)
                v = *reinterpret_cast<const uint4*>(gbase + u);
            *reinterpret_cast<uint4*>(ring + (u & (kRing - 1))) = v;
            u_loaded += kChunk;
            __syncthreads();
        }
    }

    JLS_DEV void fill() // src/scan_decoder.hpp:252-322
    {
        if (err)
            return;
        ensure();
        if (u_end - u_pos >= 8 && valid >= 0)
        {
            const uint64_t* ring64 = reinterpret_cast<const uint64_t*>(ring);
            const uint64_t w0 = ring64[(u_pos >> 3) & (kRing / 8 - 1)];
            const uint64_t w1 = ring64[((u_pos >> 3) + 1) & (kRing / 8 - 1)];
            const int sh = (int)(u_pos & 7) * 8;
            const uint64_t le = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
            const uint64_t inv = ~le; // a 0xFF byte becomes a zero byte
            const bool any_ff = ((inv - 0x0101010101010101ull) & ~inv & 0x8080808080808080ull) != 0;
            if (!any_ff)
            {
                cache |= __builtin_bswap64(le) >> valid;
                const int consumed = (64 - valid) / 8;
                u_pos += consumed;
                valid += consumed * 8;
                return;
            }
        }
        do
        {
            if (u_pos >= u_end)
            {
                if (valid <= 0)
                    err = kInvalidData;
                return;
            }
            const uint64_t b = byte_at(u_pos);
            if (b == 0xFFu && (u_pos == u_end - 1 || (byte_at(u_pos + 1) & 0x80u) != 0))
            {
                if (valid <= 0)
                    err = kInvalidData;
                return;
            }
            const int shift = 56 - valid;
            if (shift < 64)
                cache |= b << shift;
            valid += 8;
            ++u_pos;
            if (b == 0xFFu)
                --valid;
        } while (valid < 56);
    }

    JLS_DEV void init(const uint8_t* stream, uint64_t size, uint8_t* ring_, int lane_)
    {
        const uint64_t mis = (uint64_t)(reinterpret_cast<uintptr_t>(stream) & 15u);
        gbase = stream - mis;
        ring = ring_;
        lane = lane_;
        u_begin = mis;
        u_pos = mis;
        u_end = mis + size;
        u_loaded = 0; // aligned origin
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

    JLS_DEV int value(int n)
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

    JLS_DEV int unary()
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

    JLS_DEV int golomb(const Traits& t, int k, int limit)
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
        ensure();
        if (u_pos >= u_end)
        {
            err = kNeedMoreData;
            return;
        }
        if (byte_at(u_pos) != 0xFFu)
        {
            (void)bit();
            if (err)
                return;
            if (u_pos >= u_end)
            {
                err = kNeedMoreData;
                return;
            }
            if (byte_at(u_pos) != 0xFFu)
            {
                err = kInvalidData;
                return;
            }
        }
        if (cache != 0)
            err = kInvalidData;
    }

    JLS_DEV uint64_t consumed_bytes() const // get_actual_position, src/scan_decoder.hpp:92-107
    {
        int v = valid;
        uint64_t u = u_pos;
        for (;;)
        {
            const int last = byte_at(u - 1) == 0xFFu ? 7 : 8;
            if (v < last)
                return u - u_begin;
            v -= last;
            --u;
        }
    }

    JLS_DEV void restart_marker() // src/scan_decoder.hpp:237-243,335-349
    {
        if (err)
            return;
        ensure();
        const uint32_t expected = 0xD0u + restart_counter;
        if (u_pos == u_end)
        {
            err = kNeedMoreData;
            return;
        }
        uint32_t v = byte_at(u_pos++);
        if (v != 0xFFu)
        {
            err = kRestartMarkerNotFound;
            return;
        }
        do
        {
            ensure();
            if (u_pos == u_end)
            {
                err = kNeedMoreData;
                return;
            }
            v = byte_at(u_pos++);
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

struct WaveModel
{
    PackedCtx* reg;
    RunCtx* run;
};

JLS_DEV void init_model(const Traits& t, const WaveModel& m, int lane)
{
    RegCtx x{initial_a(t), 0, 0, 1};
    const PackedCtx p = pack(x);
    for (int i = lane; i < 365; i += kLanes)
        m.reg[i] = p;
    if (lane < 2)
        m.run[lane] = RunCtx{lane, initial_a(t), 1, 0};
}

JLS_DEV int decode_regular(const Traits& t, const WaveModel& m, RingReader& br, int qs, int pred) // src/scan_decoder_core.hpp:38-69
{
    const int s = qs >> 31;
    const int idx = (qs ^ s) - s;
    JLS_LOCKSTEP();
    RegCtx ctx = unpack(m.reg[idx]);
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
    { // the reference's golomb_lut hit: the whole code is inside the first byte (src/golomb_lut.cpp:24-63)
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
    JLS_LOCKSTEP();
    m.reg[idx] = pack(ctx);
    return reconstruct(t, px, (e ^ s) - s);
}

JLS_DEV int decode_run_error(const Traits& t, const WaveModel& m, RingReader& br, int which, int run_index) // src/scan_decoder_core.hpp:72-81
{
    JLS_LOCKSTEP();
    RunCtx ctx = m.run[which];
    const int k = run_k(ctx);
    if (k > 32)
    {
        br.err = kInvalidData;
        return 0;
    }
    const int em = br.golomb(t, k, t.limit - run_j(run_index) - 1);
    if (br.err)
        return 0;
    const int e = run_error_value(ctx, em + ctx.ritype, k);
    run_update(ctx, e, em, t.reset);
    JLS_LOCKSTEP();
    m.run[which] = ctx;
    return e;
}

// Run length of a run that starts with `remaining` samples left on the line: src/scan_decoder_impl.hpp:301-337.
JLS_DEV uint32_t decode_run_length(RingReader& br, int& run_index, uint32_t remaining)
{
    uint32_t run = 0;
    while (br.bit())
    {
        if (br.err)
            return 0;
        const uint32_t block = 1u << run_j(run_index);
        const uint32_t count = block < remaining - run ? block : remaining - run;
        run += count;
        if (count == block && run_index < 31)
            ++run_index;
        if (run == remaining)
            break;
    }
    if (br.err)
        return 0;
    if (run != remaining)
    {
        const int jb = run_j(run_index);
        run += jb > 0 ? (uint32_t)br.value(jb) : 0u;
    }
    if (!br.err && run > remaining)
        br.err = kInvalidData;
    return run;
}

// One line of NC co-sited components (NC > 1: ILV_SAMPLE).  `buf` holds the previous line on entry and the decoded line
// on exit; plane j lives at buf + j * ps.  corner[j] is prev[0] of plane j (the reference's line buffer element 0).
template <typename S, int NC>
JLS_DEV void decode_line(const Traits& t, const WaveModel& m, RingReader& br, S* buf, uint32_t ps, uint32_t width, int* corner,
                         int& run_index, int lane)
{
    int ra[NC], rb[NC], rd[NC], first[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j)
    {
        S* p = buf + j * ps;
        first[j] = p[1];
        if (lane == 0)
            p[width + 1] = p[width]; // prev[W+1] = prev[W]   (src/scan_codec.hpp:189-195)
        ra[j] = first[j];            // cur[0] = prev[1]
        rb[j] = corner[j];           // will become Rc of the first sample
        rd[j] = first[j];            // will become Rb of the first sample
    }
    __syncthreads();

    uint32_t i = 1;
    while (i <= width && br.err == kOk)
    {
        JLS_LOCKSTEP();
        int rc[NC], qs[NC];
        bool all_zero = true;
#pragma unroll
        for (int j = 0; j < NC; ++j)
        {
            rc[j] = rb[j];
            rb[j] = rd[j];
            rd[j] = buf[j * ps + i + 1];
            qs[j] = context_id(t, ra[j], rb[j], rc[j], rd[j]);
            all_zero = all_zero && qs[j] == 0;
        }
        if (!all_zero)
        {
#pragma unroll
            for (int j = 0; j < NC; ++j)
            {
                const int x = decode_regular(t, m, br, qs[j], med_predict(ra[j], rb[j], rc[j]));
                buf[j * ps + i] = (S)x;
                ra[j] = x;
            }
            ++i;
            continue;
        }
        // ---- run mode (src/scan_decoder_impl.hpp:264-337)
        const uint32_t remaining = width - (i - 1);
        const uint32_t run = decode_run_length(br, run_index, remaining);
        if (br.err)
            return;
        JLS_LOCKSTEP(); // every lane has read Rd of this sample before the fill may overwrite it
#pragma unroll
        for (int j = 0; j < NC; ++j)
        {
            S* p = buf + j * ps;
            const S v = (S)ra[j];
            for (uint32_t r = lane; r < run; r += kLanes) // the whole wavefront fills the run
                p[i + r] = v;
        }
        if (run == remaining)
            break;
        const uint32_t at = i + run;
        int rb_at[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j)
            rb_at[j] = at == i ? rb[j] : (int)buf[j * ps + at]; // prev[at], not yet overwritten
#pragma unroll
        for (int j = 0; j < NC; ++j)
        {
            int rx;
            if (NC == 1 && is_near(t, ra[j], rb_at[j]))
            {
                const int e = decode_run_error(t, m, br, 1, run_index);
                rx = reconstruct(t, ra[j], e);
            }
            else
            {
                const int sg = (rb_at[j] - ra[j]) < 0 ? -1 : 1;
                const int e = decode_run_error(t, m, br, 0, run_index);
                rx = reconstruct(t, rb_at[j], e * sg);
            }
            if (br.err)
                return;
            rd[j] = buf[j * ps + at + 1]; // prev[at+1]: next iteration's Rb
            rb[j] = rb_at[j];             // prev[at]:   next iteration's Rc
            buf[j * ps + at] = (S)rx;
            ra[j] = rx;
        }
        if (run_index > 0)
            --run_index;
        i = at + 1;
    }
#pragma unroll
    for (int j = 0; j < NC; ++j)
        corner[j] = first[j];
    __syncthreads();
}

// Finished line -> user's row (src/copy_from_line_buffer.hpp:19-191), all lanes.
template <typename S>
JLS_DEV void line_to_row(const ScanDesc& d, const S* buf, uint32_t ps, uint8_t* row, int lane)
{
    const bool wide = d.bits_per_sample > 8;
    if (d.interleave_mode == 0)
    {
        if (!wide)
        {
            for (uint32_t i = lane; i < d.width; i += kLanes)
                row[i] = (uint8_t)buf[1 + i];
        }
        else
        {
            uint16_t* out = reinterpret_cast<uint16_t*>(row); // rows of 16-bit images are 2-byte aligned (checked by the host)
            for (uint32_t i = lane; i < d.width; i += kLanes)
                out[i] = (uint16_t)buf[1 + i];
        }
        return;
    }
    const int nc = d.components;
    const int bytes = wide ? 2 : 1;
    for (uint32_t i = lane; i < d.width; i += kLanes)
    {
        unsigned v[4];
        for (int j = 0; j < nc; ++j)
            v[j] = buf[j * ps + 1 + i];
        if (d.color_transformation != 0 && nc == 3)
            hp_inverse(d.color_transformation, wide, (int)v[0], (int)v[1], (int)v[2], v);
        for (int j = 0; j < nc; ++j)
        {
            uint8_t* q = row + ((size_t)i * nc + j) * bytes;
            q[0] = (uint8_t)v[j];
            if (wide)
                q[1] = (uint8_t)(v[j] >> 8);
        }
    }
}

} // namespace wave

// Dynamic LDS: wave::kFixedLds + planes * (width + 2) * sizeof(S), planes = ILV_NONE ? 1 : components.
template <typename S, int NC>
__global__ void __launch_bounds__(64) decode_scans_wave(const ScanDesc* __restrict__ descs, ScanResult* __restrict__ results)
{
    using namespace wave;
    JLS_DYNAMIC_LDS(smem);
    const int lane = threadIdx.x;
    const ScanDesc d = descs[blockIdx.x];
    const Traits t = make_traits(d);
    const WaveModel m{reinterpret_cast<PackedCtx*>(smem), reinterpret_cast<RunCtx*>(smem + kCtxBytes)};
    uint8_t* ring = smem + kCtxBytes + kRunBytes;
    S* line = reinterpret_cast<S*>(smem + kFixedLds);
    const uint32_t ps = d.width + 2;
    const int planes = d.interleave_mode == 0 ? 1 : d.components;

    init_model(t, m, lane);
    for (uint32_t i = lane; i < (uint32_t)planes * ps; i += kLanes)
        line[i] = 0;
    __syncthreads();

    RingReader br;
    br.init(d.stream, d.stream_capacity, ring, lane);
    int corner[4] = {0, 0, 0, 0};
    int run_index[4] = {0, 0, 0, 0};
    const uint32_t interval = d.restart_interval == 0 ? d.height : d.restart_interval;
    uint32_t left_in_interval = interval;

    for (uint32_t y = 0; y < d.height && br.err == kOk; ++y)
    {
        if (NC > 1)
            decode_line<S, NC>(t, m, br, line, ps, d.width, corner, run_index[0], lane);
        else
            for (int j = 0; j < planes; ++j) // ILV_LINE: one sub-line per component, shared contexts, own RUNindex
                decode_line<S, 1>(t, m, br, line + j * ps, ps, d.width, corner + j, run_index[j], lane);
        if (br.err != kOk)
            break;
        line_to_row<S>(d, line, ps, d.pixels + (size_t)y * d.pixel_stride, lane);
        JLS_LOCKSTEP();
        if (--left_in_interval == 0 && y + 1 < d.height)
        { // restart interval boundary: src/scan_decoder_impl.hpp:119-127
            left_in_interval = interval;
            br.restart_marker();
            __syncthreads();
            for (int j = 0; j < 4; ++j)
            {
                run_index[j] = 0;
                corner[j] = 0;
            }
            for (uint32_t i = lane; i < (uint32_t)planes * ps; i += kLanes)
                line[i] = 0;
            init_model(t, m, lane);
            __syncthreads();
        }
    }
    if (br.err == kOk)
        br.end_scan();
    if (lane == 0)
    {
        ScanResult r;
        r.errc = br.err;
        r.flags = 1;
        r.bytes = br.err == kOk ? br.consumed_bytes() : 0;
        results[blockIdx.x] = r;
    }
}

} // namespace jls
