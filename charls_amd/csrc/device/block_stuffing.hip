// block_stuffing.hip -- stage E of the lossless pipeline in block-parallel form: raw bits -> stuffed bytes for all chunks of
// a scan at once.
//
// JPEG-LS stuffing (a byte that follows 0xFF carries 7 bits, src/scan_encoder.hpp:103-180) is sequential only through the
// position at which every output byte STARTS in the raw bit stream: byte i starts at r_i, takes w_i bits (8, or 7 behind a
// 0xFF byte) and the next one starts at r_i + w_i.  stuff_scan (pipeline_common.hip) walks that chain with one wavefront
// per scan: 27 ms for the 7 MB of a 4096 x 4096 frame -- a quarter of the time ONE frame takes to encode, and the reason the
// stage of a pass has to hide under the next pass.  Here the raw stream is cut into chunks of kStuffChunk bytes; a chunk
// owns the output bytes that start inside it, and what it needs from its predecessors is only the state in which it is
// entered: the offset (0..7) of its first byte from its first bit, and whether that byte is a 7-bit one -- 16 states:
//
//   survey   one LANE per (chunk, entry state) walks the chunk from that state (between two 0xFF bytes the walk is a
//            search for an all-ones byte at a fixed bit phase, 64 bits at a time) and records how many output bytes the
//            chunk owns and in which state it hands over;
//   resolve  one workgroup per scan composes the chunks' tables from chunk 0 (state 0) on: every lane composes the
//            tables of a run of chunks into one map (entry state -> exit state, owned bytes), one lane walks the 256 maps,
//            every lane walks its chunks again from the state it is really entered in and leaves every chunk its entry
//            state and the index of its first output byte;
//   emit     one lane per chunk walks once more, from its real entry state, and stores its bytes.
//
// (Until the end of round 3 the survey was one lane per chunk -- sixteen walks in a row -- and the composition one lane per
// scan: 0.68 + 0.53 ms of the 6 ms ONE 4096 x 4096 frame takes through the host-pointer ABI, on 111 wavefronts and on one
// lane.)
//
// Output and result words are those of stuff_scan (a trailing 0xFF is followed by 0x00, the last partial byte is zero
// padded, flags bit 1 when the capacity is within 3 bytes of the size).  The default form of stage E since round 3
// (CHARLS_AMD_BLOCK_STUFFING=0 selects stuff_scan): equal to stuff_scan byte for byte on the CPU harness
// (tests/test_emu_block_stuffing.py) and in the GPU suite.
#pragma once
#include <hip/hip_runtime.h>

#include "pipeline_common.hip"

namespace jls {
namespace pipe {

constexpr uint32_t kStuffChunk = 1024;             // raw bytes per chunk
constexpr uint32_t kStuffChunkBits = kStuffChunk * 8;
constexpr uint32_t kStuffStates = 16;              // offset of the first byte (0..7) | 8 when it is a 7-bit byte
constexpr uint32_t kStuffResolveThreads = 256;     // lanes of the resolve step: each composes ceil(chunks / 256) tables

// Words per chunk in Work::stuff_tables: kStuffStates table entries (owned bytes | exit state << 16), then the entry state
// and the first output byte index (two words) as resolve leaves them.
constexpr uint32_t kStuffWords = kStuffStates + 4;

struct __attribute__((packed)) UnalignedU64
{
    uint64_t v;
};
JLS_DEV uint64_t raw_bits_at(const uint8_t* raw, uint64_t bit) // the 64 raw bits from `bit` on, first bit most significant
{
    const uint8_t* p = raw + (bit >> 3);
    // (one load of eight bytes at any address -- gfx950 takes unaligned global loads -- instead of eight of one byte)
    const uint64_t v = __builtin_bswap64(reinterpret_cast<const UnalignedU64*>(p)->v);
    const uint32_t s = (uint32_t)(bit & 7u);
    return s == 0 ? v : (v << s) | ((uint64_t)p[8] >> (8 - s));
}

// One pass over the output bytes that start in [from, end): from a byte of `width` bits at `from`.  emit(byte, index) is
// called for every byte when kEmit.  Returns the number of bytes; `from` / `width` are left at the first byte at or behind
// `end`; last_ff says whether the last byte was 0xFF.
template <bool kEmit, typename Emit>
JLS_DEV uint32_t walk_chunk(const uint8_t* raw, uint64_t& from, uint32_t& width, uint64_t end, bool& last_ff, Emit emit)
{
    uint32_t count = 0;
    uint64_t r = from;
    uint32_t wd = width;
    last_ff = false;
    while (r < end)
    {
        if (wd == 7)
        { // the byte behind a 0xFF: seven bits, never 0xFF itself
            if (kEmit)
                emit((uint32_t)(raw_bits_at(raw, r) >> 57), count);
            ++count;
            r += 7;
            wd = 8;
            last_ff = false;
            continue;
        }
        // eight bytes at the current phase; which of them are 0xFF (bit 0 of every byte of `ones` after the folds)
        const uint64_t w = raw_bits_at(raw, r);
        uint64_t ones = w & (w >> 1);
        ones &= ones >> 2;
        ones &= ones >> 4;
        ones &= 0x0101010101010101ull;
        const uint64_t left = end - r;
        const uint32_t fit = left >= 64 ? 8u : (uint32_t)((left + 7) / 8); // bytes of this window that start before `end`
        const uint32_t first_ff = ones ? (uint32_t)(__builtin_clzll(ones) / 8) : 8u; // in stream order
        const uint32_t take = first_ff < fit ? first_ff + 1 : fit;
        if (kEmit)
            for (uint32_t b = 0; b < take; ++b)
                emit((uint32_t)(w >> (56 - 8 * b)) & 0xFFu, count + b);
        count += take;
        r += (uint64_t)take * 8;
        last_ff = first_ff < fit;
        wd = last_ff ? 7 : 8;
    }
    from = r;
    width = wd;
    return count;
}

// Workgroups (of 64 lanes) of the survey of a raw stream of at most `raw_bytes` bytes: one lane per (chunk, entry state).
__host__ __device__ constexpr uint32_t stuff_survey_blocks(uint64_t raw_bytes)
{
    return (uint32_t)(((raw_bytes / kStuffChunk + 1) * kStuffStates + 63) / 64);
}

// grid (stuff_survey_blocks, scans) x 64 lanes.
__global__ void __launch_bounds__(64) stuff_survey(const Work* __restrict__ works)
{
    const Work w = works[blockIdx.y];
    const uint64_t total_bits = *w.total_bits;
    const uint32_t chunks = (uint32_t)((total_bits + kStuffChunkBits - 1) / kStuffChunkBits);
    const uint32_t item = blockIdx.x * 64u + threadIdx.x;
    const uint32_t chunk = item / kStuffStates, state = item % kStuffStates;
    if (chunk >= chunks || (uint64_t)(total_bits + 7) / 8 > (uint64_t)w.raw_words * 4)
        return;
    const uint8_t* raw = reinterpret_cast<const uint8_t*>(w.raw);
    const uint64_t begin = (uint64_t)chunk * kStuffChunkBits;
    const uint64_t end = begin + kStuffChunkBits < total_bits ? begin + kStuffChunkBits : total_bits;
    uint64_t r = begin + (state & 7u);
    uint32_t wd = (state & 8u) ? 7u : 8u;
    bool last_ff;
    const uint32_t count = walk_chunk<false>(raw, r, wd, end, last_ff, [](uint32_t, uint32_t) {});
    const uint32_t exit_state = (uint32_t)((r - (begin + kStuffChunkBits)) & 7u) | (wd == 7 ? 8u : 0u);
    w.stuff_tables[(size_t)chunk * kStuffWords + state] = count | (exit_state << 16);
}

// grid (scans) x kStuffResolveThreads lanes.
__global__ void __launch_bounds__(kStuffResolveThreads) stuff_resolve(const Work* __restrict__ works)
{
    __shared__ uint32_t s_row[kStuffResolveThreads * kStuffStates];   // the table of the chunk a lane is at
    __shared__ uint32_t s_count[kStuffResolveThreads * kStuffStates]; // bytes a lane's chunks own, by the state they are entered in
    __shared__ uint64_t s_exit[kStuffResolveThreads];                 // ... and the state they hand over in, four bits per entry state
    __shared__ uint32_t s_entry[kStuffResolveThreads];                // the state a lane's chunks are really entered in
    __shared__ uint64_t s_first[kStuffResolveThreads];                // output bytes before them
    const Work w = works[blockIdx.x];
    const uint64_t total_bits = *w.total_bits;
    // Nothing was surveyed for a scan that is invalid or whose raw stream did not fit its buffer (write_raw_bits clamps
    // its stores but still reports the full bit count): its chunk count would run past the tables of this work area.
    if ((*w.status & kStatusInvalid) != 0 || (uint64_t)(total_bits + 7) / 8 > (uint64_t)w.raw_words * 4)
        return;
    const uint32_t chunks = (uint32_t)((total_bits + kStuffChunkBits - 1) / kStuffChunkBits);
    const uint32_t lane = threadIdx.x;
    const uint32_t per_lane = (chunks + kStuffResolveThreads - 1) / kStuffResolveThreads;
    const uint32_t from = lane * per_lane < chunks ? lane * per_lane : chunks;
    const uint32_t to = from + per_lane < chunks ? from + per_lane : chunks;
    uint32_t* row = s_row + lane * kStuffStates; // (a lane's own slot: a table is looked up by state, which registers cannot be)
    const u32x4* tables = reinterpret_cast<const u32x4*>(w.stuff_tables); // kStuffWords words = five quads per chunk
    u32x4* row4 = reinterpret_cast<u32x4*>(row);
    constexpr uint32_t kQuads = kStuffWords / 4;
    // 1) the chunks of the lane as ONE map: exits = state after them, counts = bytes they own, for each of the 16 entry states
    uint64_t exits = 0xFEDCBA9876543210ull;
    uint32_t counts[kStuffStates];
#pragma unroll
    for (uint32_t st = 0; st < kStuffStates; ++st)
        counts[st] = 0;
    u32x4 ahead[4];
    if (from < to)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            ahead[q] = tables[(size_t)from * kQuads + q];
    for (uint32_t c = from; c < to; ++c)
    {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            row4[q] = ahead[q];
        if (c + 1 < to)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                ahead[q] = tables[(size_t)(c + 1) * kQuads + q];
        uint64_t next = 0;
#pragma unroll
        for (uint32_t st = 0; st < kStuffStates; ++st)
        {
            const uint32_t t = row[(uint32_t)(exits >> (4 * st)) & 15u];
            counts[st] += t & 0xFFFFu;
            next |= (uint64_t)(t >> 16) << (4 * st);
        }
        exits = next;
    }
    s_exit[lane] = exits;
#pragma unroll
    for (uint32_t st = 0; st < kStuffStates; ++st)
        s_count[lane * kStuffStates + st] = counts[st];
    __syncthreads();
    // 2) one lane walks the maps: the first byte of the stream starts at bit 0 and is an 8-bit one
    if (lane == 0)
    {
        uint32_t state = 0;
        uint64_t index = 0;
        for (uint32_t l = 0; l < kStuffResolveThreads; ++l)
        {
            s_entry[l] = state;
            s_first[l] = index;
            index += s_count[l * kStuffStates + state];
            state = (uint32_t)(s_exit[l] >> (4 * state)) & 15u;
        }
    }
    __syncthreads();
    // 3) every lane walks its chunks again, from the state they are really entered in
    uint32_t state = s_entry[lane];
    uint64_t index = s_first[lane];
    if (from < to)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            ahead[q] = tables[(size_t)from * kQuads + q];
    for (uint32_t c = from; c < to; ++c)
    {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            row4[q] = ahead[q];
        if (c + 1 < to)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                ahead[q] = tables[(size_t)(c + 1) * kQuads + q];
        uint32_t* out = w.stuff_tables + (size_t)c * kStuffWords + kStuffStates;
        out[0] = state;
        out[1] = (uint32_t)index;
        out[2] = (uint32_t)(index >> 32);
        const uint32_t t = row[state];
        index += t & 0xFFFFu;
        state = t >> 16;
    }
}

// grid (max(1, ceil(chunks / 64)), scans) x 64 lanes; writes the scan's result (stuff_scan's words).
__global__ void __launch_bounds__(64) stuff_emit(const ScanDesc* __restrict__ descs, const Work* __restrict__ works,
                                                 ScanResult* __restrict__ results)
{
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const uint64_t total_bits = *w.total_bits;
    const uint32_t chunks = (uint32_t)((total_bits + kStuffChunkBits - 1) / kStuffChunkBits);
    const uint32_t chunk = blockIdx.x * 64u + threadIdx.x;
    const bool invalid = (*w.status & kStatusInvalid) != 0;
    const bool overflow = (uint64_t)(total_bits + 7) / 8 > (uint64_t)w.raw_words * 4;
    if (invalid || overflow || chunks == 0)
    { // nothing to stuff: the first lane of the scan reports, with stuff_scan's words (an empty stream still meets the
      // capacity rules: flags 2 when fewer than 4 bytes are left)
        if (chunk == 0)
        {
            ScanResult res{invalid ? kInvalidData : (overflow ? kDestinationTooSmall : kOk), 0, 0};
            if (res.errc == kOk && d.stream_capacity < 4)
                res.flags = 2;
            results[blockIdx.y] = res;
        }
        return;
    }
    if (chunk >= chunks)
        return;
    const uint8_t* raw = reinterpret_cast<const uint8_t*>(w.raw);
    const uint32_t* mine = w.stuff_tables + (size_t)chunk * kStuffWords + kStuffStates;
    const uint32_t state = mine[0];
    const uint64_t first = (uint64_t)mine[1] | ((uint64_t)mine[2] << 32);
    const uint64_t begin = (uint64_t)chunk * kStuffChunkBits;
    const uint64_t end = begin + kStuffChunkBits < total_bits ? begin + kStuffChunkBits : total_bits;
    uint64_t r = begin + (state & 7u);
    uint32_t wd = (state & 8u) ? 7u : 8u;
    bool last_ff;
    uint8_t* out = d.stream;
    const uint64_t capacity = d.stream_capacity;
    const uint32_t count = walk_chunk<true>(raw, r, wd, end, last_ff, [&](uint32_t byte, uint32_t k) {
        if (first + k < capacity)
            out[first + k] = (uint8_t)byte;
    });
    if (chunk + 1 == chunks)
    { // the last chunk closes the scan: src/scan_encoder.hpp:107-112, a trailing 0xFF is followed by seven zero bits
        uint64_t written = first + count;
        if (last_ff)
        {
            if (written < capacity)
                out[written] = 0;
            ++written;
        }
        ScanResult res{kOk, 0, written};
        if (written > capacity)
            res.errc = kDestinationTooSmall;
        else if (capacity - written < 4)
            res.flags = 2; // the reference's verdict depends on its flush history here: the host re-runs the exact kernel
        results[blockIdx.y] = res;
    }
}

} // namespace pipe
} // namespace jls
