// block_stuffing.hip -- stage E of the lossless pipeline in block-parallel form: raw bits -> stuffed bytes for all chunks of
// a scan at once.
//
// JPEG-LS stuffing (a byte that follows 0xFF carries 7 bits, src/scan_encoder.hpp:103-180) is sequential only through the
// position at which every output byte STARTS in the raw bit stream: byte i starts at r_i, takes w_i bits (8, or 7 behind a
// 0xFF byte) and the next one starts at r_i + w_i.  stuff_scan (lossless_pipeline.hip) walks that chain with one wavefront
// per scan: 27 ms for the 7 MB of a 4096 x 4096 frame -- a quarter of the time ONE frame takes to encode, and the reason the
// stage of a pass has to hide under the next pass.  Here the raw stream is cut into chunks of kStuffChunk bytes; a chunk
// owns the output bytes that start inside it, and what it needs from its predecessors is only the state in which it is
// entered: the offset (0..7) of its first byte from its first bit, and whether that byte is a 7-bit one -- 16 states:
//
//   survey   one LANE per chunk walks its chunk from each of the 16 entry states (between two 0xFF bytes the walk is a
//            search for an all-ones byte at a fixed bit phase, 64 bits at a time) and records, per state, how many output
//            bytes the chunk owns and in which state it hands over;
//   resolve  one wavefront per scan composes the chunks' tables from chunk 0 (state 0) on -- a table look-up per chunk out
//            of LDS -- and leaves every chunk its entry state and the index of its first output byte;
//   emit     one lane per chunk walks once more, from its real entry state, and stores its bytes.
//
// Output and result words are those of stuff_scan (a trailing 0xFF is followed by 0x00, the last partial byte is zero
// padded, flags bit 1 when the capacity is within 3 bytes of the size).  The default form of stage E since round 3
// (CHARLS_AMD_BLOCK_STUFFING=0 selects stuff_scan): equal to stuff_scan byte for byte on the CPU harness
// (tests/test_emu_block_stuffing.py) and in the GPU suite.
#pragma once
#include <hip/hip_runtime.h>

#include "lossless_pipeline.hip"

namespace jls {
namespace pipe {

constexpr uint32_t kStuffChunk = 1024;             // raw bytes per chunk
constexpr uint32_t kStuffChunkBits = kStuffChunk * 8;
constexpr uint32_t kStuffStates = 16;              // offset of the first byte (0..7) | 8 when it is a 7-bit byte
constexpr uint32_t kStuffTile = 256;               // chunks whose tables the resolve step holds in LDS at a time

// Words per chunk in Work::stuff_tables: kStuffStates table entries (owned bytes | exit state << 16), then the entry state
// and the first output byte index (two words) as resolve leaves them.
constexpr uint32_t kStuffWords = kStuffStates + 4;

JLS_DEV uint64_t raw_bits_at(const uint8_t* raw, uint64_t bit) // the 64 raw bits from `bit` on, first bit most significant
{
    const uint8_t* p = raw + (bit >> 3);
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i)
        v = (v << 8) | p[i];
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

// grid (ceil(chunks / 64), scans) x 64 lanes.
__global__ void __launch_bounds__(64) stuff_survey(const Work* __restrict__ works)
{
    const Work w = works[blockIdx.y];
    const uint64_t total_bits = *w.total_bits;
    const uint32_t chunks = (uint32_t)((total_bits + kStuffChunkBits - 1) / kStuffChunkBits);
    const uint32_t chunk = blockIdx.x * 64u + threadIdx.x;
    if (chunk >= chunks || (uint64_t)(total_bits + 7) / 8 > (uint64_t)w.raw_words * 4)
        return;
    const uint8_t* raw = reinterpret_cast<const uint8_t*>(w.raw);
    const uint64_t begin = (uint64_t)chunk * kStuffChunkBits;
    const uint64_t end = begin + kStuffChunkBits < total_bits ? begin + kStuffChunkBits : total_bits;
    uint32_t* table = w.stuff_tables + (size_t)chunk * kStuffWords;
    for (uint32_t state = 0; state < kStuffStates; ++state)
    {
        uint64_t r = begin + (state & 7u);
        uint32_t wd = (state & 8u) ? 7u : 8u;
        bool last_ff;
        const uint32_t count = walk_chunk<false>(raw, r, wd, end, last_ff, [](uint32_t, uint32_t) {});
        const uint32_t exit_state = (uint32_t)((r - (begin + kStuffChunkBits)) & 7u) | (wd == 7 ? 8u : 0u);
        table[state] = count | (exit_state << 16);
    }
}

// grid (scans) x 64 lanes.
__global__ void __launch_bounds__(64) stuff_resolve(const Work* __restrict__ works)
{
    __shared__ uint32_t s_table[kStuffTile * kStuffStates];
    __shared__ uint32_t s_entry[kStuffTile];
    __shared__ uint32_t s_first[kStuffTile * 2];
    const Work w = works[blockIdx.x];
    const uint64_t total_bits = *w.total_bits;
    // Nothing was surveyed for a scan that is invalid or whose raw stream did not fit its buffer (write_raw_bits clamps
    // its stores but still reports the full bit count): its chunk count would run past the tables of this work area.
    if ((*w.status & kStatusInvalid) != 0 || (uint64_t)(total_bits + 7) / 8 > (uint64_t)w.raw_words * 4)
        return;
    const uint32_t chunks = (uint32_t)((total_bits + kStuffChunkBits - 1) / kStuffChunkBits);
    const int lane = threadIdx.x;
    uint32_t state = 0;  // the first byte of the stream starts at bit 0 and is an 8-bit one
    uint64_t index = 0;  // output bytes before the chunk
    for (uint32_t c0 = 0; c0 < chunks; c0 += kStuffTile)
    {
        const uint32_t n = chunks - c0 < kStuffTile ? chunks - c0 : kStuffTile;
        for (uint32_t k = lane; k < n * kStuffStates; k += 64)
            s_table[k] = w.stuff_tables[(size_t)(c0 + k / kStuffStates) * kStuffWords + k % kStuffStates];
        __syncthreads();
        if (lane == 0)
            for (uint32_t c = 0; c < n; ++c)
            {
                s_entry[c] = state;
                s_first[2 * c] = (uint32_t)index;
                s_first[2 * c + 1] = (uint32_t)(index >> 32);
                const uint32_t t = s_table[c * kStuffStates + state];
                index += t & 0xFFFFu;
                state = t >> 16;
            }
        __syncthreads();
        for (uint32_t c = lane; c < n; c += 64)
        {
            uint32_t* out = w.stuff_tables + (size_t)(c0 + c) * kStuffWords + kStuffStates;
            out[0] = s_entry[c];
            out[1] = s_first[2 * c];
            out[2] = s_first[2 * c + 1];
        }
        __syncthreads();
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
