// speculative_stuffing.hip -- stage E of the lossless pipeline for batches: raw bits -> stuffed bytes, the chunks of a scan
// walked in parallel from GUESSED entry states.
//
// The state in which a chunk of the raw stream is entered -- the offset (0..7) of its first output byte from its first bit,
// and whether that byte follows a 0xFF and carries 7 bits (block_stuffing.hip) -- is a serial recurrence over the 0xFF bytes
// of everything before it.  But like the context statistics (tile_pipeline.hip) the recurrence FORGETS: two walks of the
// same bits from different states stay apart only until their byte boundaries coincide, every 0xFF that either of them
// meets moves its boundaries by one bit against the other's, so their distance is a random walk on eight positions that is
// absorbed at zero -- within a few KB of ordinary coded data (one byte in 256 is 0xFF).  So:
//
//   survey   one WAVEFRONT per chunk (64 KB of raw stream) walks from `warm` bytes before its chunk, from state 0, to its
//            chunk's first bit -- a guess of the entry state -- and on through the chunk: bytes owned, exit state
//            (stuff_walk: 512 bytes per round);
//   resolve  one wavefront per scan checks every chunk boundary (the guess must equal the predecessor's exit state; chunk 0
//            is entered in state 0 by definition), walks a chunk whose guess was wrong again from the true state, and leaves
//            every chunk its entry state and the index of its first output byte (a prefix sum).  A scan with more wrong
//            guesses than a handful (coded data without 0xFF bytes: nothing makes walks meet) is handed to the sequential
//            form instead;
//   emit     one wavefront per chunk walks once more, from its real entry state, and stores its bytes; the last chunk closes
//            the scan.  For a scan that resolve gave up on, the first wavefront walks the whole stream.
//
// Measured on real coded data (tools: a 1024 x 1024 test frame has one 0xFF in 760 bytes): with a warm-up of 16 KB three
// guesses in a hundred are wrong, with 64 KB none in 170.  Against stuff_scan (one wavefront per scan, whatever the number
// of scans in the pass) this reads the raw stream 2 + warm / chunk times, with a wavefront per 64 KB; against the block form
// (block_stuffing.hip: exact tables for all 16 entry states of every 1 KB chunk, 17 reads of the stream, a LANE per walk) it
// is what a pass of hundreds of scans can afford.  Output and result words are those of stuff_scan, byte for byte
// (tests/test_emu_block_stuffing.py, GPU suite).  CHARLS_AMD_SPEC_STUFFING=0 switches it off; CHARLS_AMD_SPEC_CHUNK /
// CHARLS_AMD_SPEC_WARM (bytes) size it (the tests use tiny values so that guesses fail).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "block_stuffing.hip"

namespace jls {
namespace pipe {

// Words per chunk in Work::stuff_tables: survey {guess, owned bytes, exit state, -}, resolve {entry state, first byte index
// (two words), -}.  Word 0 of the whole table region's LAST entry slot is not used; the give-up flag of a scan lives in word 3
// of chunk 0.
constexpr uint32_t kSpecWords = 8;
constexpr uint32_t kSpecMaxFixes = 16;    // wrong guesses resolve repairs itself before it gives the scan up
constexpr uint32_t kSpecGiveUp = 0xFFFFFFFFu;

JLS_DEV uint32_t entry_state_of(const StuffCursor& cur, uint64_t boundary) // cur.bp: first byte start at or behind `boundary`
{
    return (uint32_t)((cur.bp - boundary) & 7u) | (cur.short_first ? 8u : 0u);
}
JLS_DEV StuffCursor cursor_from(uint32_t state, uint64_t boundary)
{
    return StuffCursor{boundary + (state & 7u), 0, (state & 8u) != 0};
}

// grid (max_chunks, scans) x 64 lanes: one wavefront per chunk.
__global__ void __launch_bounds__(64) stuff_spec_survey(const Work* __restrict__ works, uint32_t chunk_bytes, uint32_t warm_bytes)
{
    const Work w = works[blockIdx.y];
    const uint64_t total_bits = *w.total_bits;
    const uint64_t chunk_bits = (uint64_t)chunk_bytes * 8;
    const uint32_t chunks = (uint32_t)((total_bits + chunk_bits - 1) / chunk_bits);
    const uint32_t chunk = blockIdx.x;
    if (chunk >= chunks || (*w.status & kStatusInvalid) != 0 || (uint64_t)(total_bits + 7) / 8 > (uint64_t)w.raw_words * 4)
        return;
    const uint8_t* raw = reinterpret_cast<const uint8_t*>(w.raw);
    const uint64_t begin = (uint64_t)chunk * chunk_bits;
    const uint64_t end = begin + chunk_bits < total_bits ? begin + chunk_bits : total_bits;
    uint32_t guess = 0;
    if (chunk != 0)
    { // (a warm-up that reaches back to bit 0 starts in the true state: its guess is exact)
        const uint64_t warm_bits = (uint64_t)warm_bytes * 8;
        StuffCursor warm{begin > warm_bits ? begin - warm_bits : 0, 0, false};
        stuff_walk<false>(raw, warm, begin, nullptr, 0);
        guess = entry_state_of(warm, begin);
    }
    StuffCursor cur = cursor_from(guess, begin);
    stuff_walk<false>(raw, cur, end, nullptr, 0);
    if (threadIdx.x == 0)
    {
        uint32_t* mine = w.stuff_tables + (size_t)chunk * kSpecWords;
        mine[0] = guess;
        mine[1] = (uint32_t)cur.written;
        mine[2] = entry_state_of(cur, begin + chunk_bits);
        mine[3] = 0;
    }
}

// grid (scans) x 64 lanes.
__global__ void __launch_bounds__(64) stuff_spec_resolve(const Work* __restrict__ works, uint32_t chunk_bytes)
{
    const Work w = works[blockIdx.x];
    const uint64_t total_bits = *w.total_bits;
    if ((*w.status & kStatusInvalid) != 0 || (uint64_t)(total_bits + 7) / 8 > (uint64_t)w.raw_words * 4)
        return;
    const uint64_t chunk_bits = (uint64_t)chunk_bytes * 8;
    const uint32_t chunks = (uint32_t)((total_bits + chunk_bits - 1) / chunk_bits);
    const uint8_t* raw = reinterpret_cast<const uint8_t*>(w.raw);
    const int lane = threadIdx.x;
    uint32_t carry_exit = 0; // exit state of the chunk before this group of 64 (chunk 0 is entered in state 0)
    uint64_t carry_index = 0;
    uint32_t fixes = 0;
    for (uint32_t c0 = 0; c0 < chunks; c0 += 64)
    {
        const uint32_t c = c0 + (uint32_t)lane;
        const bool live = c < chunks;
        uint32_t* mine = w.stuff_tables + (size_t)(live ? c : 0) * kSpecWords;
        uint32_t guess = live ? mine[0] : 0u, count = live ? mine[1] : 0u, exit_state = live ? mine[2] : 0u;
        // a chunk whose guess differs from what its predecessor hands over is walked again (by the whole wavefront); chunks in
        // order, because the repaired chunk's own exit state is what its successor has to be checked against
        for (;;)
        {
            uint32_t before = __shfl_up(exit_state, 1);
            if (lane == 0)
                before = carry_exit;
            const unsigned long long wrong = __ballot(live && guess != before);
            if (wrong == 0)
                break;
            const int f = (int)__ffsll(wrong) - 1;
            ++fixes;
            if (fixes > kSpecMaxFixes)
                break;
            const uint32_t true_state = (uint32_t)__shfl((int)before, f);
            const uint64_t begin = (uint64_t)(c0 + (uint32_t)f) * chunk_bits;
            const uint64_t end = begin + chunk_bits < total_bits ? begin + chunk_bits : total_bits;
            StuffCursor cur = cursor_from(true_state, begin);
            stuff_walk<false>(raw, cur, end, nullptr, 0);
            if (lane == f)
            {
                count = (uint32_t)cur.written;
                exit_state = entry_state_of(cur, begin + chunk_bits);
                guess = true_state;
            }
        }
        if (fixes > kSpecMaxFixes)
            break;
        // first output byte of every chunk: exclusive prefix sum of the owned bytes
        uint64_t incl = count;
        for (int delta = 1; delta < 64; delta <<= 1)
        {
            const uint64_t up = __shfl_up(incl, delta);
            if (lane >= delta)
                incl += up;
        }
        if (live)
        {
            const uint64_t first = carry_index + incl - count;
            mine[4] = guess;
            mine[5] = (uint32_t)first;
            mine[6] = (uint32_t)(first >> 32);
        }
        carry_index += __shfl(incl, 63);
        const int last_lane = chunks - c0 >= 64 ? 63 : (int)(chunks - c0) - 1;
        carry_exit = __shfl(exit_state, last_lane);
    }
    if (lane == 0 && chunks != 0)
        w.stuff_tables[3] = fixes > kSpecMaxFixes ? kSpecGiveUp : 0u;
}

// grid (max(1, max_chunks), scans) x 64 lanes: one wavefront per chunk; writes the scan's result (stuff_scan's words).
__global__ void __launch_bounds__(64) stuff_spec_emit(const ScanDesc* __restrict__ descs, const Work* __restrict__ works,
                                                      ScanResult* __restrict__ results, uint32_t chunk_bytes)
{
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const uint64_t total_bits = *w.total_bits;
    const uint64_t chunk_bits = (uint64_t)chunk_bytes * 8;
    const uint32_t chunks = (uint32_t)((total_bits + chunk_bits - 1) / chunk_bits);
    const uint32_t chunk = blockIdx.x;
    const bool invalid = (*w.status & kStatusInvalid) != 0;
    const bool overflow = (uint64_t)(total_bits + 7) / 8 > (uint64_t)w.raw_words * 4;
    if (invalid || overflow || chunks == 0)
    {
        if (chunk == 0 && threadIdx.x == 0)
        {
            ScanResult res{invalid ? kInvalidData : (overflow ? kDestinationTooSmall : kOk), 0, 0};
            if (res.errc == kOk && d.stream_capacity < 4)
                res.flags = 2;
            results[blockIdx.y] = res;
        }
        return;
    }
    if (w.stuff_tables[3] == kSpecGiveUp)
    { // too many wrong guesses: the whole stream in sequence, by the scan's first wavefront
        if (chunk == 0)
        {
            const ScanResult res = stuff_stream_sequential(d, w, total_bits);
            if (threadIdx.x == 0)
                results[blockIdx.y] = res;
        }
        return;
    }
    if (chunk >= chunks)
        return;
    const uint8_t* raw = reinterpret_cast<const uint8_t*>(w.raw);
    const uint32_t* mine = w.stuff_tables + (size_t)chunk * kSpecWords + 4;
    const uint64_t first = (uint64_t)mine[1] | ((uint64_t)mine[2] << 32);
    const uint64_t begin = (uint64_t)chunk * chunk_bits;
    const uint64_t end = begin + chunk_bits < total_bits ? begin + chunk_bits : total_bits;
    StuffCursor cur = cursor_from(mine[0], begin);
    cur.written = first;
    const uint64_t capacity = d.stream_capacity;
    stuff_walk<true>(raw, cur, end, d.stream, capacity);
    if (chunk + 1 == chunks)
    { // the last chunk closes the scan: src/scan_encoder.hpp:107-112, a trailing 0xFF is followed by seven zero bits
        uint64_t written = cur.written;
        if (cur.short_first)
        {
            if (threadIdx.x == 0 && written < capacity)
                d.stream[written] = 0;
            ++written;
        }
        ScanResult res{kOk, 0, written};
        if (written > capacity)
            res.errc = kDestinationTooSmall;
        else if (capacity - written < 4)
            res.flags = 2;
        if (threadIdx.x == 0)
            results[blockIdx.y] = res;
    }
}

// Chunk and warm-up of the speculative form (bytes of raw stream): 64 KB each; CHARLS_AMD_SPEC_CHUNK / CHARLS_AMD_SPEC_WARM
// override them (multiples of 8 bytes; the tests use tiny values so that guesses fail).
struct SpecGeometry
{
    uint32_t chunk_bytes, warm_bytes;
};
inline SpecGeometry stuff_spec_geometry()
{
    SpecGeometry g{65536, 65536};
    if (const long long knob = knobs::get(knobs::kSpecChunk); knob != knobs::kUnset)
        g.chunk_bytes = (uint32_t)std::max<long long>(64, std::min<long long>(knob, 1 << 30)) / 8 * 8;
    if (const long long knob = knobs::get(knobs::kSpecWarm); knob != knobs::kUnset)
        g.warm_bytes = (uint32_t)std::max<long long>(0, std::min<long long>(knob, 1 << 30)) / 8 * 8;
    return g;
}

// Words of Work::stuff_tables the speculative form needs for a raw stream of at most raw_bytes bytes.
__host__ __device__ constexpr size_t stuff_spec_table_words(uint64_t raw_bytes, uint32_t chunk_bytes)
{
    return (size_t)(raw_bytes / chunk_bytes + 2) * kSpecWords;
}

} // namespace pipe
} // namespace jls
