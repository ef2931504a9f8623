// tile_pipeline.hip -- the parallel JPEG-LS encoder for lossless scans (every interleave mode, every line width).
//
// In lossless mode everything except the adaptive statistics is a pure function of the image (reference
// src/scan_encoder_impl.hpp:109-144).  The pipeline is built around two observations (round 2 had the same idea as a
// line-by-line scatter with one lane per chain; it is gone):
//
//  1. HBM traffic.  The round-2 pipeline moved 1.48 GB per 4096 x 4096 frame (61 x the algorithmic bytes): 6 B/sample of
//     analysis results written and read back, 4-byte records scattered line by line to ~30 chains per 64 samples (every
//     32-byte sector written several times), 8-byte codes + 1-byte lengths + a 4-byte slot map gathered back.  Here the
//     image is cut into TILES of up to kMaxTileSamples samples (whole lines, or segments of lines that do not fit:
//     plan_tiles); a tile's events are sorted by chain INSIDE LDS and leave as contiguous pieces (one per chain, hundreds of
//     bytes), the slot map is a 2-byte tile-local index that overwrites the 2-byte key in place, and records and code words
//     live in SLOTS of 2 bytes for samples of up to 8 bits (4 otherwise; see Slot).  8 B/sample of work area instead of 21.
//
//  2. The chain floor.  {A,B,C,N} of a context is a serial recurrence over the context's samples; round 2
//     walked every chain with ONE lane (58 ms for the 1.09 M events of the longest chain of a test frame, whatever the
//     batch).  But the recurrence FORGETS: N is a function of the event index alone, C is a feedback loop that pulls B into
//     (-N, 0], A is halved every RESET/2 events.  Two walks of the same events from different states meet after a few
//     hundred events and are identical from then on.  So a chain is cut into JOBS of job_events events; every job is
//     walked by its own lane from a guessed state warm_events before its first event (no output), records the state in
//     which it reaches its first event and the state in which it ends; settle_chains then checks, chain by chain, that
//     every job started in exactly the state its predecessor ended in.  Where that holds -- everywhere, on anything but
//     noise-like data (tools/spec_convergence.c: 0 of 16 253 jobs of the test frame disagree with a warm-up of 1024
//     events) -- the job's codes are the sequential ones by construction; a job that disagrees is walked again from the
//     true state by the settling lane, so the result is exact in every case and only the time depends on the data (and is
//     counted: Counter).  One walker computes k, the error correction and the Golomb word as well.
//
// Stages:
//   A  analyze_tiles   (planar scans whose lines fit a tile; every other scan: analyze_pixel_tiles, tile_pixel_mode.hip)
//                      one workgroup per tile: chain id + sign of every sample (key, 2 B), run-mode segmentation as a
//                      carry chain over ballot masks, events per (tile, chain)
//   B1 sum_chains / plan_chains / apply_chains   per scan: exclusive prefix over tiles per chain -> where each tile's piece
//                      of each chain goes (in slots); chain bases; jobs per chain (the tiles in 64 groups, a workgroup each)
//   B2 sort_tiles      (sort_pixel_tiles) one workgroup per tile: stable ranks through LDS mask tables, records into LDS in
//                      (chain, line, column) order, out as pieces; key -> tile-local slot
//   C1 walk_jobs       one LANE per job: warm-up, then record -> code word, in chain order
//   C2 settle_chains   one wavefront per chain: job boundaries checked 64 at a time, disagreeing jobs re-walked
//   C3 count_runs / scan_runs / compact_rare_runs / warm_run_jobs / walk_run_jobs / settle_runs   the run chain (RUNindex,
//                      the two run-interruption contexts) cut into jobs like the regular chains; the context of the rarer
//                      interruption type is computed exactly, beside the jobs' warm-ups
//   D  pack_tiles      one workgroup per tile: the tile's codes back into LDS piece by piece, gathered in raster order
//                      through the 2-byte slots, concatenated MSB-first; bit offset of a tile by chained look-back
//   E  stuff_scan (pipeline_common.hip) / speculative_stuffing.hip / block_stuffing.hip
//
// Output is byte-identical to scan_encoder::encode_scan.  No MFMA: nothing here is a contraction.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "pipeline_common.hip"

namespace jls {
namespace tile {

using pipe::kChains;
using pipe::kInterruptChain;
using pipe::kNoEvent;
using pipe::kStatusInvalid;
using pipe::kZeroContextChain;


constexpr uint32_t kMaxTileSamples = 8192; // samples of a tile at most (whole lines); also the widest line the pipeline takes
constexpr uint32_t kTileLines = 16;        // lines of a tile at most
constexpr uint32_t kSegments = 16;         // pieces of lines that the wavefronts of a tile workgroup work on, at most
constexpr uint32_t kThreads = 512;         // workgroup of analyze_tiles / sort_tiles
constexpr uint32_t kWaves = kThreads / 64;
constexpr uint32_t kPackThreads = 512;     // workgroup of pack_tiles (16 consecutive samples per thread at most)
constexpr uint16_t kNoLocalSlot = 0xFFFF;
// Records and code words live in SLOTS: 4 bytes for samples wider than 8 bits, 2 bytes otherwise (an 8-bit record is {x, Px}
// with the sign of the context folded in by reflection, an 8-bit code word {length : 6 | bits : 10}: round 3 moved 4 + 4 bytes
// per sample for 17 + 16 bits of information, and the walkers' time is their bytes).  An entry of the RUN chain (chain 0:
// run records and run-length code words need 32 bits) takes run_slots slots: two of the 2-byte ones.
template <typename S>
struct SlotOf
{
    typedef uint32_t type;
};
template <>
struct SlotOf<uint8_t>
{
    typedef uint16_t type;
};
template <typename S>
using Slot = typename SlotOf<S>::type;
template <typename S>
constexpr uint32_t run_slots_of()
{
    return sizeof(Slot<S>) == 2 ? 2u : 1u;
}
constexpr uint32_t kRowChainWords = 288;    // LDS words of the sort stage's row table: a tile has at most (12288 + 16) / 64 + 367 rows of 64 slots (2 bytes each)
constexpr uint32_t kChainPad = 32;         // chains start on multiples of 32 slots (64 or 128 bytes: whole lines)
constexpr uint32_t kSlack = kChains * kChainPad + 128; // spare slots behind rec / code: padding + read-ahead of the walkers
// Slots the chains of `samples` samples in `lines` lines can take at most: every sample has at most one event, a run start
// takes run_slots of them, and two run starts of a line are never neighbours (a run of length 0 is interrupted by a sample
// that differs from the line above it, so the next sample's gradients are not all zero): at most samples / 2 + lines run starts.
__host__ __device__ inline uint64_t slots_capacity(uint64_t samples, uint32_t run_slots, uint64_t lines)
{
    const uint64_t runs = samples / 2 + lines < samples ? samples / 2 + lines : samples;
    return samples + (run_slots > 1 ? runs : 0);
}
// Bytes of the slots a tile of `tile_capacity` samples can fill at most (sorted records in the sort stage, code words in the
// pack stage): 4-byte slots, or 2-byte ones of which a run start takes two.
__host__ __device__ inline size_t stage_bytes(uint32_t tile_capacity, uint32_t sample_bytes)
{
    return sample_bytes == 1 ? (size_t)slots_capacity(tile_capacity, 2, kTileLines) * 2 : (size_t)tile_capacity * 4;
}
constexpr uint32_t kPlanGroups = 64;       // sum_chains / apply_chains take the tiles of a scan in this many groups
#define JLS_HOST_DEV_EARLY __host__ __device__ inline
constexpr uint32_t kRunTag = 1u << 31;     // code word of a run-length code: ones : 6 | tail length : 5 | tail : 20

// What the speculative stages did in a call (charls_amd_speculation_counters): jobs of the regular chains / of the run chain,
// and how many of them the settling lane had to walk again because their predecessor did not end in the state they assumed.
enum Counter : uint32_t
{
    kCountJobs = 0,
    kCountJobsRewalked = 1,
    kCountRunJobs = 2,
    kCountRunJobsRewalked = 3,
    kCountRareSegments = 4, // segments of the rarer run context's event lists that started from a guessed state
    kCountRareSerial = 5,   // scans whose list had to be walked again, serially, because a guess was wrong
    kCounters = 6
};
// Debug build (-DJLS_PHASE_CLOCKS, tools/phase_clocks.sh): the shader clocks every wavefront of the tile kernels
// of every 64th workgroup spends between the marks (all of them queue up behind their own atomics), summed per mark in 64-bit words behind the counters of the call -- which phase of a kernel the
// time goes to, where rocprofv3 only has the kernel's total.  The marks compile to nothing otherwise.
#ifdef JLS_PHASE_CLOCKS
#define JLS_PHASE_BEGIN() uint64_t phase_t = __builtin_amdgcn_s_memtime()
#define JLS_PHASE(slot)                                                                                                  \
    do                                                                                                                   \
    {                                                                                                                    \
        const uint64_t phase_now = __builtin_amdgcn_s_memtime();                                                         \
        if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 0 && phase_now - phase_t < (1ull << 32)) /* (a sample) */   \
            atomicAdd(reinterpret_cast<unsigned long long*>(w.counters + 8) + (slot), (unsigned long long)(phase_now - phase_t)); \
        phase_t = phase_now;                                                                                             \
    } while (0)
#else
#define JLS_PHASE_BEGIN() \
    do                    \
    {                     \
    } while (0)
#define JLS_PHASE(slot) \
    do                  \
    {                   \
    } while (0)
#endif

// State of a job at its first event (after the warm-up) and behind its last one (N is a function of the event index);
// bad: an event of the job would make the reference raise invalid_data.
struct JobState
{
    int32_t in_a, in_b, in_c, out_a, out_b, out_c;
    uint32_t bad, pad;
};

struct Work
{
    uint16_t* keyinv;      // [lines * width] A: chain | sign << 9 or kNoEvent; after B2: tile-local slot or kNoLocalSlot
    uint32_t* seg;         // [(tiles + 1) * kChains] A: events per (tile, chain); B1: first slot of the tile's piece; row `tiles` = chain ends
    uint32_t* chain_total; // [kChains]
    uint32_t* chain_base;  // [kChains]
    uint32_t* job_first;   // [kChains + 1] first job of the chain (chains coded by walk_jobs), number of jobs at the end
    uint32_t* rec;         // [slots_capacity + kSlack] slots: records in chain order (Slot<S>; the run chain's entries are 32-bit)
    uint32_t* code;        // [slots_capacity + kSlack] slots: code words in chain order
    JobState* jobs;        // [samples / job_events + kChains]
    uint32_t* plan_part;   // [kPlanGroups][kChains] sum_chains: slots of a group of tiles; plan_chains: where its first tile's pieces start
    struct RunJob* run_jobs; // [samples / run_job_events + 2] jobs of the run chain, and an entry behind the last one (totals)
    uint64_t* blockbase;   // [tiles] look-back states of pack_tiles; followed by tile_tail (cleared together)
    uint64_t* tile_tail;   // [tiles] pack_tiles: the bits of the tile's last, partial word | valid << 63
    uint32_t* raw;
    uint64_t raw_words;
    uint64_t* total_bits;
    uint32_t* status;
    uint32_t* counters;    // [kCounters] of the CALL (shared by its scans): what the speculation did, see Counter
    // the launch's geometry (all scans of a launch share width, sample type and interleave mode; a scan may have FEWER
    // lines than the launch was sized for -- the last restart interval of a frame -- and then has fewer tiles)
    uint32_t lines_per_tile, tiles, job_events, warm_events;
    uint32_t run_job_events, run_warm_events; // (run_job_events: a multiple of 8)
    uint32_t rare_warm_events;                // walk_rare_context: events OF THE RARER TYPE a lane walks before its segment
    // lines that do not fit a tile are cut into segs_per_line tiles of seg_pixels pixels (a multiple of 64; the last one
    // shorter), lines_per_tile is then 1; otherwise segs_per_line = 1.  tile_capacity: samples of a tile at most.
    uint32_t segs_per_line, seg_pixels, tile_capacity;
    uint32_t run_slots;    // slots an entry of the run chain takes (run_slots_of<S>())
};
template <typename S>
JLS_DEV Slot<S>* rec_slots(const Work& w)
{
    return reinterpret_cast<Slot<S>*>(w.rec);
}
template <typename S>
JLS_DEV Slot<S>* code_slots(const Work& w)
{
    return reinterpret_cast<Slot<S>*>(w.code);
}

JLS_DEV uint32_t tile_of_block(uint32_t block, uint32_t tiles) // XCD-aware: workgroup b runs on XCD b % 8; each XCD gets a band of tiles
{
    const uint32_t band = (tiles + 7) / 8;
    const uint32_t t = (block & 7u) * band + (block >> 3);
    return (block >> 3) < band && t < tiles ? t : tiles;
}

JLS_DEV uint32_t scan_lines(const ScanDesc& d)
{
    return d.interleave_mode == 1 ? pipe::coded_lines(d) : d.height;
}
JLS_DEV uint32_t scan_tiles(const ScanDesc& d, const Work& w) // tiles of THIS scan (<= w.tiles)
{
    return (scan_lines(d) + w.lines_per_tile - 1) / w.lines_per_tile * w.segs_per_line;
}
// Samples per pixel of a coded line: the components of a sample-interleaved scan are coded pixel by pixel, component by
// component (src/scan_encoder_impl.hpp:147-246), so its "line" is width * components samples long.
JLS_DEV uint32_t samples_per_pixel(const ScanDesc& d)
{
    return d.interleave_mode == 2 ? (uint32_t)d.components : 1u;
}
// The samples of a tile in raster order of the scan: [first, first + count).
struct TileSpan
{
    uint64_t first;
    uint32_t count;
};
JLS_DEV TileSpan tile_span(const ScanDesc& d, const Work& w, uint32_t tile)
{
    const uint32_t nc = samples_per_pixel(d), line_samples = d.width * nc, lines = scan_lines(d);
    TileSpan s;
    if (w.segs_per_line == 1)
    {
        const uint32_t first_line = tile * w.lines_per_tile;
        const uint32_t tile_lines = lines - first_line < w.lines_per_tile ? lines - first_line : w.lines_per_tile;
        s.first = (uint64_t)first_line * line_samples;
        s.count = tile_lines * line_samples;
    }
    else
    {
        const uint32_t line = tile / w.segs_per_line, px0 = (tile % w.segs_per_line) * w.seg_pixels;
        const uint32_t pixels = d.width - px0 < w.seg_pixels ? d.width - px0 : w.seg_pixels;
        s.first = (uint64_t)line * line_samples + (uint64_t)px0 * nc;
        s.count = pixels * nc;
    }
    return s;
}

// How the wavefronts of a tile workgroup share a tile: every line is cut into `pieces` runs of chunks (64 samples), a
// "segment" is one piece of one line, segments are numbered in raster order.  Tiles of few long lines get several pieces
// per line so that all wavefronts have work.
struct TileGeometry
{
    uint32_t first_line, tile_lines, width, chunks, pieces, chunks_per_piece, segments;
};
JLS_DEV TileGeometry tile_geometry(const ScanDesc& d, const Work& w, uint32_t tile)
{
    TileGeometry g;
    const uint32_t lines = scan_lines(d);
    g.first_line = tile * w.lines_per_tile;
    g.tile_lines = lines - g.first_line < w.lines_per_tile ? lines - g.first_line : w.lines_per_tile;
    g.width = d.width;
    g.chunks = (d.width + 63) / 64;
    g.pieces = w.lines_per_tile >= kWaves ? 1u : kWaves / w.lines_per_tile;
    g.chunks_per_piece = (g.chunks + g.pieces - 1) / g.pieces;
    g.segments = g.tile_lines * g.pieces;
    return g;
}

// Exclusive prefix sum of up to 2 * kThreads values held one or two per thread (index threadIdx.x and threadIdx.x +
// blockDim.x); s_tmp: one word per wavefront.  All threads of the workgroup call it.
JLS_DEV void block_exclusive_scan(uint32_t& lo, uint32_t& hi, uint32_t* s_tmp)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = (int)(blockDim.x >> 6);
    uint32_t carry = 0;
    for (int half = 0; half < 2; ++half)
    {
        uint32_t& v = half == 0 ? lo : hi;
        uint32_t incl = v;
        for (int delta = 1; delta < 64; delta <<= 1)
        {
            const uint32_t up = __shfl_up(incl, delta);
            if (lane >= delta)
                incl += up;
        }
        __syncthreads(); // s_tmp free again
        if (lane == 63)
            s_tmp[wave] = incl;
        __syncthreads();
        uint32_t before = carry, all = 0;
        for (int w2 = 0; w2 < waves; ++w2)
        {
            const uint32_t n = s_tmp[w2];
            before += w2 < wave ? n : 0;
            all += n;
        }
        v = before + incl - v;
        carry += all;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The lines of a tile and the line above it, staged in LDS (planar scans): analysis reads every sample five times (x, Ra,
// Rb, Rc, Rd); out of LDS those reads cost an LDS latency instead of a trip to the L2, and the staging itself is wide
// coalesced loads issued back to back.  (analyze_tiles / sort_tiles are the specialisation for PLANAR scans whose lines fit
// a tile -- the headline path; every other scan takes tile_pixel_mode.hip.  The template parameter ILV is 0.)
template <typename S, int ILV>
struct Samples
{
    const ScanDesc& d;
    const S* rows;       // LDS: line first_line - 1, then the tile's lines
    uint32_t first_line; // of the tile
    int mask;
    JLS_DEV int operator()(uint32_t line, uint32_t x) const
    {
        static_assert(ILV == 0, "planar scans only");
        return (int)rows[(size_t)(line + 1 - first_line) * d.width + x] & mask;
    }
};

template <typename S, int ILV>
JLS_DEV void stage_lines(const ScanDesc& d, const TileGeometry& g, S* rows)
{
    const uint32_t width = g.width;
    const uint32_t bytes = width * (uint32_t)sizeof(S);
    const uint32_t first = g.first_line == 0 ? 1u : 0u; // line 0 of the staging area lies above the scan: zeros
    if (first)
        for (uint32_t x = threadIdx.x; x < width; x += blockDim.x)
            rows[x] = 0; // src/scan_encoder_impl.hpp:55-70
    const uint8_t* from = d.pixels + (size_t)(g.first_line + first - 1) * d.pixel_stride;
    uint8_t* to = reinterpret_cast<uint8_t*>(rows) + (size_t)first * bytes;
    const uint32_t lines = g.tile_lines + 1 - first;
    if (((reinterpret_cast<uintptr_t>(from) | d.pixel_stride | bytes) & 3u) == 0 && (reinterpret_cast<uintptr_t>(to) & 3u) == 0)
    { // words: all loads of a thread are requested before the first one is stored (a tile and the line above it are at
      // most 16 KB: eight words per thread)
        const uint32_t words_per_line = bytes / 4, total = lines * words_per_line;
        constexpr int kMost = (kMaxTileSamples * 2 / 4 + kThreads - 1) / kThreads;
        // word i of the staging area = word (i % words_per_line) of line i / words_per_line; a thread's words are kThreads
        // apart, so line and word advance by constants (one division per thread, not one per word)
        const uint32_t step_lines = kThreads / words_per_line, step_words = kThreads % words_per_line;
        uint32_t line = threadIdx.x / words_per_line, word = threadIdx.x % words_per_line;
        uint32_t held[kMost];
#pragma unroll
        for (int j = 0; j < kMost; ++j)
        {
            held[j] = line < lines ? reinterpret_cast<const uint32_t*>(from + (size_t)line * d.pixel_stride)[word] : 0u;
            line += step_lines;
            word += step_words;
            if (word >= words_per_line)
            {
                word -= words_per_line;
                ++line;
            }
        }
#pragma unroll
        for (int j = 0; j < kMost; ++j)
        {
            const uint32_t i = threadIdx.x + (uint32_t)j * kThreads;
            if (i < total)
                reinterpret_cast<uint32_t*>(to)[i] = held[j];
        }
        for (uint32_t i = threadIdx.x + kMost * kThreads; i < total; i += kThreads) // (tiles of many short lines of wide samples)
        {
            const uint32_t l2 = i / words_per_line, w2 = i - l2 * words_per_line;
            reinterpret_cast<uint32_t*>(to)[i] = reinterpret_cast<const uint32_t*>(from + (size_t)l2 * d.pixel_stride)[w2];
        }
    }
    else
        for (uint32_t r = 0; r < lines; ++r)
        {
            const S* from1 = reinterpret_cast<const S*>(from + (size_t)r * d.pixel_stride);
            S* to1 = reinterpret_cast<S*>(to + (size_t)r * bytes);
            for (uint32_t x = threadIdx.x; x < width; x += blockDim.x)
                to1[x] = from1[x];
        }
}

JLS_HOST_DEV_EARLY uint32_t sort_segments(uint32_t lines_per_tile) // segments of a full tile (tile_geometry)
{
    return lines_per_tile >= kWaves ? lines_per_tile : lines_per_tile * (kWaves / lines_per_tile);
}

// LDS carve-up shared by analyze_tiles and sort_tiles (byte offsets; every region 16-byte aligned).
struct TileLds
{
    uint32_t rows, keys, masks, table, end;
};
template <typename S, int ILV>
JLS_DEV TileLds tile_lds(uint32_t width, uint32_t lines_per_tile, bool with_keys)
{
    auto up = [](uint32_t v) { return (v + 15u) & ~15u; };
    TileLds l;
    const uint32_t chunks = (width + 63) / 64;
    l.rows = 0;
    l.keys = up((lines_per_tile + 1) * width * (uint32_t)sizeof(S));
    l.masks = l.keys + (with_keys ? up(lines_per_tile * width * 2u) : 0u);
    l.table = l.masks + up(lines_per_tile * chunks * 16u);
    l.end = l.table;
    return l;
}

// ---------------------------------------------------------------------------------------------------------------
// A: grid (8 * ceil(tiles / 8), scans) x 512.
// LDS: lines | keys[tile] u16 | per line and chunk: eq, q0 masks | hist[kChains + 1] | gradient table (512 B)
template <typename S, int ILV>
__global__ void __launch_bounds__(kThreads) analyze_tiles(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    JLS_DYNAMIC_LDS(smem);
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const Traits t = make_traits(d);
    const uint32_t tile = tile_of_block(blockIdx.x, scan_tiles(d, w));
    if (tile >= scan_tiles(d, w))
        return;
    const TileGeometry g = tile_geometry(d, w, tile);
    constexpr uint32_t step = 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t width = g.width, chunks = g.chunks;
    const TileLds lds = tile_lds<S, ILV>(width, w.lines_per_tile, true);
    S* s_rows = reinterpret_cast<S*>(smem + lds.rows);
    uint16_t* s_key = reinterpret_cast<uint16_t*>(smem + lds.keys);
    uint64_t* s_eq = reinterpret_cast<uint64_t*>(smem + lds.masks); // [line][chunk]
    uint64_t* s_q0 = s_eq + (size_t)w.lines_per_tile * chunks;
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + lds.table);
    unsigned char* s_grad = reinterpret_cast<unsigned char*>(s_hist + kChains + 1);
    const int mask = (1 << d.bits_per_sample) - 1;
    const Samples<S, ILV> sample{d, s_rows, g.first_line, mask};

    stage_lines<S, ILV>(d, g, s_rows);
    for (uint32_t c = threadIdx.x; c < (uint32_t)kChains; c += kThreads)
        s_hist[c] = 0;
    if (sizeof(S) == 1)
        for (uint32_t q = threadIdx.x; q < 511; q += kThreads)
            s_grad[q] = (unsigned char)(quantize(t, (int)q - 255) + 4);
    __syncthreads();

    // ---- pass 1: every sample as if coded in regular mode; equality / zero-context masks per 64-sample chunk
    for (uint32_t sgm = wave; sgm < g.segments; sgm += kWaves)
    {
        const uint32_t r = sgm / g.pieces, piece = sgm % g.pieces;
        const uint32_t y = g.first_line + r; // coded line
        const uint32_t k0 = piece * g.chunks_per_piece;
        const uint32_t k1 = k0 + g.chunks_per_piece < chunks ? k0 + g.chunks_per_piece : chunks;
        // edge samples of the line (src/scan_codec.hpp:189-195 and the two-line ping-pong of src/scan_encoder_impl.hpp:55-106)
        const int edge_a = y >= step ? sample(y - step, 0) : 0; // cur[0]  = prev[1]
        const int edge_c = y >= 2 * step ? (r >= 1 ? sample(y - 2 * step, 0) : pipe::load_sample<S, 0>(d, y - 2, 0, mask)) : 0; // prev[0]
        for (uint32_t k = k0; k < k1; ++k)
        {
            const uint32_t x = k * 64 + lane;
            bool eq = false, q0 = false;
            if (x < width)
            {
                const int v = sample(y, x);
                const int ra = x > 0 ? sample(y, x - 1) : edge_a;
                int rb = 0, rc = 0, rd = 0;
                if (y >= step)
                {
                    rb = sample(y - step, x);
                    rc = x > 0 ? sample(y - step, x - 1) : edge_c;
                    rd = sample(y - step, x + 1 < width ? x + 1 : width - 1);
                }
                else
                    rc = x > 0 ? 0 : edge_c;
                const int qs = sizeof(S) == 1
                                   ? ((int)s_grad[rd - rb + 255] * 9 + (int)s_grad[rb - rc + 255]) * 9 + (int)s_grad[rc - ra + 255] - 364
                                   : context_id(t, ra, rb, rc, rd);
                const int sg = qs >> 31;
                const int ctx = (qs ^ sg) - sg;
                s_key[r * width + x] = (uint16_t)(ctx | ((sg & 1) << 9));
                eq = v == ra;
                q0 = qs == 0;
            }
            const unsigned long long m_eq = __ballot(eq);
            const unsigned long long m_q0 = __ballot(q0);
            if (lane == 0)
            {
                s_eq[r * chunks + k] = m_eq;
                s_q0[r * chunks + k] = m_q0;
            }
        }
    }
    __syncthreads();
    // ---- pass 2: run-mode state before every sample.  s' = eq & (s | q0) is a carry chain: generate = eq & q0,
    // propagate = eq, so one 64-bit addition per chunk resolves 64 samples (src/scan_encoder_impl.hpp:249-275).  A piece
    // that does not start its line first runs the (scalar) chain over the chunks before it.
    for (uint32_t sgm = wave; sgm < g.segments; sgm += kWaves)
    {
        const uint32_t r = sgm / g.pieces, piece = sgm % g.pieces;
        const uint32_t y = g.first_line + r;
        const uint32_t k0 = piece * g.chunks_per_piece;
        const uint32_t k1 = k0 + g.chunks_per_piece < chunks ? k0 + g.chunks_per_piece : chunks;
        uint16_t* key_row = w.keyinv + (size_t)y * width;
        unsigned long long carry = 0;
        for (uint32_t k = 0; k < k1; ++k)
        {
            const unsigned long long a = s_eq[r * chunks + k];
            const unsigned long long b = a & s_q0[r * chunks + k];
            const unsigned long long sum = a + b + carry;
            const unsigned long long st = sum ^ a ^ b; // bit i: in-run state before sample i
            carry = (((a & b) | ((a | b) & st)) >> 63) & 1ull;
            const uint32_t x = k * 64 + lane;
            if (k >= k0 && x < width)
            {
                const bool s = (st >> lane) & 1ull;
                const bool q0 = (s_q0[r * chunks + k] >> lane) & 1ull;
                const bool eq = (a >> lane) & 1ull;
                uint16_t key = s_key[r * width + x];
                if (!(s || q0))
                    atomicAdd(&s_hist[key & 0x1FF], 1u); // regular sample
                else if (s && eq)
                    key = kNoEvent; // inside a run
                else if (s)
                { // the sample that ends a run started earlier: coded by the run lane, owns a slot of its own
                    key = (uint16_t)kInterruptChain;
                    atomicAdd(&s_hist[kInterruptChain], 1u);
                }
                else
                { // a run starts here (possibly of length 0); its length is the number of kNoEvent keys that follow
                    key = 0;
                    atomicAdd(&s_hist[0], 1u);
                }
                key_row[x] = key;
            }
        }
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < (uint32_t)kChains; c += kThreads)
        w.seg[(size_t)tile * kChains + c] = s_hist[c];
}

// ---------------------------------------------------------------------------------------------------------------
// B1: column-wise exclusive prefix of the (tiles x kChains) count matrix: where the piece of every (tile, chain) starts in
// rec / code.  Three launches: sum_chains adds up the tiles of a scan in kPlanGroups groups (a workgroup per group and scan),
// plan_chains (a workgroup per scan) turns the 64 x 367 sums into the chains' totals, their places and job numbers and every
// group's starting offsets, apply_chains walks the groups' tiles again and leaves the offsets.  (One workgroup per scan did all
// of it until the end of round 4: 6 MB through ONE CU, 264 us of the 2.1 ms ONE frame took.)
// sum_chains / apply_chains: grid (kPlanGroups, scans) x 384.  plan_chains: grid (scans) x 384.
constexpr uint32_t kPlanThreads = 384; // >= kChains
JLS_DEV void plan_group(const ScanDesc& d, const Work& w, uint32_t g, uint32_t& t0, uint32_t& t1)
{
    const uint32_t tiles = scan_tiles(d, w);
    const uint32_t per_group = (tiles + kPlanGroups - 1) / kPlanGroups;
    t0 = g * per_group < tiles ? g * per_group : tiles;
    t1 = t0 + per_group < tiles ? t0 + per_group : tiles;
}
__global__ void __launch_bounds__(kPlanThreads) sum_chains(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const uint32_t g = blockIdx.x, c = threadIdx.x;
    if (c >= (uint32_t)kChains)
        return;
    uint32_t t0, t1;
    plan_group(d, w, g, t0, t1);
    // (everything below is in SLOTS: an event of the run chain -- chain 0 -- takes w.run_slots of them, every other event one;
    // chain_total alone counts EVENTS)
    uint32_t sum = 0;
    for (uint32_t t = t0; t < t1; ++t)
        sum += w.seg[(size_t)t * kChains + c];
    w.plan_part[g * kChains + c] = sum * (c == 0 ? w.run_slots : 1u);
}
__global__ void __launch_bounds__(kPlanThreads) plan_chains(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    __shared__ uint32_t s_base[kChains + 1];
    const Work w = works[blockIdx.x];
    uint32_t total = 0;
    if (threadIdx.x < (uint32_t)kChains)
    {
        const uint32_t c = threadIdx.x;
        for (uint32_t g = 0; g < kPlanGroups; ++g)
            total += w.plan_part[g * kChains + c];
        w.chain_total[c] = total / (c == 0 ? w.run_slots : 1u);
        s_base[c] = total;
    }
    if (threadIdx.x == kPlanThreads - 1)
    { // the two result words of the later stages start at zero
        *w.total_bits = 0;
        *w.status = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    { // 367 values: a serial scan is cheaper than its synchronisation
        uint32_t acc = 0, jobs = 0;
        for (int c = 0; c < kChains; ++c)
        {
            const uint32_t n = s_base[c]; // slots
            s_base[c] = acc;
            w.chain_base[c] = acc;
            acc += (n + kChainPad - 1) / kChainPad * kChainPad;
            w.job_first[c] = jobs;
            if (c != 0 && c != kInterruptChain)
                jobs += (n + w.job_events - 1) / w.job_events;
        }
        w.job_first[kChains] = jobs;
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)kChains)
    { // where the pieces of every group's first tile start
        const uint32_t c = threadIdx.x;
        uint32_t running = s_base[c];
        for (uint32_t g = 0; g < kPlanGroups; ++g)
        {
            const uint32_t n = w.plan_part[g * kChains + c];
            w.plan_part[g * kChains + c] = running;
            running += n;
        }
    }
}
__global__ void __launch_bounds__(kPlanThreads) apply_chains(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const uint32_t g = blockIdx.x, c = threadIdx.x;
    if (c >= (uint32_t)kChains)
        return;
    const uint32_t tiles = scan_tiles(d, w);
    uint32_t t0, t1;
    plan_group(d, w, g, t0, t1);
    uint32_t running = w.plan_part[g * kChains + c];
    for (uint32_t t = t0; t < t1; ++t)
    {
        const uint32_t n = w.seg[(size_t)t * kChains + c] * (c == 0 ? w.run_slots : 1u);
        w.seg[(size_t)t * kChains + c] = running;
        running += n;
    }
    if (t1 == tiles && t0 < t1)
        w.seg[(size_t)tiles * kChains + c] = running;
}

// ---------------------------------------------------------------------------------------------------------------
// Records.  Samples of up to 8 bits: see make_record.  Wider samples: what the walker needs of x and Px is
// d = sign * (x - Px) mod 2^16 and how far Px is from the end of the sample range it is nearer to (the bias C only ever
// moves the prediction by up to 128, so only one end can clip it): d | room << 16 | side << 24 | sign << 31 with
// room = min(distance, 255), side = 1 for the upper end.  src/scan_encoder_core.hpp:57-67.
template <typename S>
JLS_DEV uint32_t make_record(int x, int px, int sign_bit, int maxval)
{
    if (sizeof(S) == 1)
    { // {x, Px} reflected about the sample range when the sign of the context is negative: sign (x - clamp(Px + sign C)) =
      // (maxval - x) - clamp((maxval - Px) + C), so the walker computes every event as one of positive sign and the sign
      // needs no bit -- 16 bits per record
        const int xr = sign_bit ? maxval - x : x, pr = sign_bit ? maxval - px : px;
        return (uint32_t)xr | ((uint32_t)pr << 8);
    }
    const int d = sign_bit ? px - x : x - px;
    const int lo = px, hi = maxval - px;
    const int side = hi < lo ? 1 : 0;
    int room = side ? hi : lo;
    room = room > 255 ? 255 : room;
    return ((uint32_t)d & 0xFFFFu) | ((uint32_t)room << 16) | ((uint32_t)side << 24) | ((uint32_t)sign_bit << 31);
}

// Code word of a sample in a slot: 8-bit samples {length : 6 | bits : 10} (a Golomb code of an 8-bit sample has at most LIMIT =
// 32 bits, of which at most 9 -- the one and the k <= 8 low bits, or the escape's one and qbpp = 8 bits -- are not leading
// zeros: A / N stays below 132), wider samples {length : 8 | bits : 24}.
template <typename S>
JLS_DEV Slot<S> pack_code(int len, uint32_t bits)
{
    if (sizeof(S) == 1)
        return (Slot<S>)(((uint32_t)len << 10) | (bits & 0x3FFu));
    return (Slot<S>)(((uint32_t)len << 24) | bits);
}

// Record of a run start: everything the run chain needs of the image.  The length of the run; whether it ends with the line;
// otherwise type and error value of the interruption sample (src/scan_encoder_core.hpp:105-125: functions of the image in
// lossless mode) and, in a line-interleaved scan, the component whose RUNindex counts.  One word: a run that does not end
// with its line is shorter than the line (8192 / 4096 samples at most for samples of one / two bytes), Errval has as
// many bits as the samples.
template <typename S>
struct RunRecord
{
    static constexpr uint32_t kRunBits = sizeof(S) == 1 ? 13 : 12, kErrBits = sizeof(S) == 1 ? 9 : 16;
    static JLS_DEV uint32_t end_of_line(uint32_t run, uint32_t component) { return run | (component << 15) | (1u << 31); }
    static JLS_DEV uint32_t interrupted(uint32_t run, int err, int which, uint32_t component)
    {
        return run | (((uint32_t)err & ((1u << kErrBits) - 1u)) << kRunBits) | ((uint32_t)which << (kRunBits + kErrBits)) |
               (component << (kRunBits + kErrBits + 1));
    }
    static JLS_DEV bool is_end_of_line(uint32_t v) { return (v >> 31) != 0; }
    static JLS_DEV uint32_t run(uint32_t v) { return is_end_of_line(v) ? v & 0x7FFFu : v & ((1u << kRunBits) - 1u); }
    static JLS_DEV int err(uint32_t v) { return (int)(v << (32 - kRunBits - kErrBits)) >> (32 - kErrBits); }
    static JLS_DEV int which(uint32_t v) { return (int)((v >> (kRunBits + kErrBits)) & 1u); }
    static JLS_DEV uint32_t component(uint32_t v) { return (v >> (is_end_of_line(v) ? 15u : kRunBits + kErrBits + 1)) & 3u; }
};

// B2: grid (8 * ceil(tiles / 8), scans) x 512.
// LDS: lines | keys[tile] u16 | noev[line][chunk] u64, lead[line][chunk + 1] u32 | segoff[kSegments][kChains] u32 |
//      tileoff, count, global [kChains + 1] | scan scratch | same[kWaves][kChains + 1] | stage[tile] u32
template <typename S, int ILV>
__global__ void __launch_bounds__(kThreads) sort_tiles(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    JLS_DYNAMIC_LDS(smem);
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const Traits t = make_traits(d);
    const uint32_t tile = tile_of_block(blockIdx.x, scan_tiles(d, w));
    if (tile >= scan_tiles(d, w))
        return;
    const TileGeometry g = tile_geometry(d, w, tile);
    constexpr uint32_t step = 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t width = g.width, chunks = g.chunks;
    const TileLds lds = tile_lds<S, ILV>(width, w.lines_per_tile, false);
    S* s_rows = reinterpret_cast<S*>(smem + lds.rows);
    uint64_t* s_noev = reinterpret_cast<uint64_t*>(smem + lds.masks);                                // [line][chunk]
    uint32_t* s_lead = reinterpret_cast<uint32_t*>(s_noev + (size_t)w.lines_per_tile * chunks);       // [line][chunk + 1] (fits: 16 B per (line, chunk) reserved)
    uint32_t* s_segoff = reinterpret_cast<uint32_t*>(smem + lds.table);                               // [kSegments][kChains]
    uint32_t* s_tileoff = s_segoff + sort_segments(w.lines_per_tile) * kChains; // first local slot of the chain
    uint32_t* s_count = s_tileoff + kChains + 1;          // events of the chain in this tile
    uint32_t* s_global = s_count + kChains + 1;           // first global slot of the tile's piece
    uint32_t* s_tmp = s_global + kChains + 1;             // kWaves words (+ padding to 16 words)
    uint32_t* s_same = s_tmp + 16;                        // [kWaves][kChains + 1] lanes of a chunk per chain, see P2; zero between uses
    uint32_t* s_rowbase = s_same + kWaves * (kChains + 1); // [kChains + 1] first row of the chain's piece (P3)
    uint16_t* s_rowchain = reinterpret_cast<uint16_t*>(s_rowbase + kChains + 1); // [slots of a tile / 64 + kChains + 1] (kRowChainWords words)
    Slot<S>* s_stage = reinterpret_cast<Slot<S>*>(s_rowbase + kChains + 1 + kRowChainWords);
    const int mask = (1 << d.bits_per_sample) - 1;
    const Samples<S, ILV> sample{d, s_rows, g.first_line, mask};
    const uint16_t* key_tile = w.keyinv + (size_t)g.first_line * width;

    JLS_PHASE_BEGIN();
    stage_lines<S, ILV>(d, g, s_rows);
    JLS_PHASE(0);
    for (uint32_t i = threadIdx.x; i < g.segments * (uint32_t)kChains; i += kThreads)
        s_segoff[i] = 0;
    for (uint32_t i = threadIdx.x; i < kWaves * ((uint32_t)kChains + 1); i += kThreads)
        s_same[i] = 0;
    __syncthreads();
    JLS_PHASE(1);
    // ---- P1: keys into LDS, events per (segment, chain), samples inside runs per chunk
    for (uint32_t sgm = wave; sgm < g.segments; sgm += kWaves)
    {
        const uint32_t r = sgm / g.pieces, piece = sgm % g.pieces;
        const uint32_t k0 = piece * g.chunks_per_piece;
        const uint32_t k1 = k0 + g.chunks_per_piece < chunks ? k0 + g.chunks_per_piece : chunks;
        for (uint32_t kb = k0; kb < k1; kb += 16)
        { // the keys of up to 16 chunks are requested before the first one is used (one trip to memory, not sixteen)
            uint16_t held[16];
#pragma unroll
            for (int j = 0; j < 16; ++j)
            {
                const uint32_t x = (kb + j) * 64 + lane;
                held[j] = kb + j < k1 && x < width ? key_tile[(size_t)r * width + x] : kNoEvent;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
            {
                const uint32_t k = kb + j;
                if (k < k1) // (uniform)
                {
                    const uint32_t x = k * 64 + lane;
                    const uint16_t key = held[j];
                    if (x < width && key != kNoEvent)
                        atomicAdd(&s_segoff[sgm * kChains + (key & 0x1FF)], (key & 0x1FF) == 0 ? run_slots_of<S>() : 1u); // (slots)
                    const unsigned long long m = __ballot(x < width && key == kNoEvent);
                    if (lane == 0)
                        s_noev[r * chunks + k] = m;
                }
            }
        }
    }
    JLS_PHASE(2);
    __syncthreads();
    JLS_PHASE(3);
    // ---- offsets: chains in order, inside a chain the segments in (raster) order; per line, the number of samples
    // inside runs from the first sample of every chunk on
    {
        uint32_t n[2] = {0, 0};
        for (int half = 0; half < 2; ++half)
        {
            const uint32_t c = threadIdx.x + (uint32_t)half * kThreads;
            if (c < (uint32_t)kChains)
                for (uint32_t sgm = 0; sgm < g.segments; ++sgm)
                    n[half] += s_segoff[sgm * kChains + c];
        }
        uint32_t off[2] = {n[0], n[1]};
        block_exclusive_scan(off[0], off[1], s_tmp);
        for (int half = 0; half < 2; ++half)
        {
            const uint32_t c = threadIdx.x + (uint32_t)half * kThreads;
            if (c < (uint32_t)kChains)
            {
                s_tileoff[c] = off[half];
                s_count[c] = n[half];
                s_global[c] = w.seg[(size_t)tile * kChains + c];
                uint32_t running = off[half];
                for (uint32_t sgm = 0; sgm < g.segments; ++sgm)
                {
                    const uint32_t m = s_segoff[sgm * kChains + c];
                    s_segoff[sgm * kChains + c] = running;
                    running += m;
                }
            }
        }
        // the pieces cut into rows of 64 records for the way out (P3): row q belongs to chain s_rowchain[q]; the interruption
        // chain has no records
        uint32_t rows[2], row_base[2];
        for (int half = 0; half < 2; ++half)
        {
            const uint32_t c = threadIdx.x + (uint32_t)half * kThreads;
            rows[half] = row_base[half] = c < (uint32_t)kChains && c != (uint32_t)kInterruptChain ? (n[half] + 63) / 64 : 0u;
        }
        block_exclusive_scan(row_base[0], row_base[1], s_tmp);
        for (int half = 0; half < 2; ++half)
        {
            const uint32_t c = threadIdx.x + (uint32_t)half * kThreads;
            if (c < (uint32_t)kChains)
            {
                s_rowbase[c] = row_base[half];
                for (uint32_t j = 0; j < rows[half]; ++j)
                    s_rowchain[row_base[half] + j] = (uint16_t)c;
                if (c == (uint32_t)kChains - 1)
                    s_rowbase[kChains] = row_base[half] + rows[half];
            }
        }
        if (threadIdx.x >= kThreads - g.tile_lines)
        { // (the last threads: they have no chain of their own to scan)
            const uint32_t r = kThreads - 1 - threadIdx.x;
            uint32_t lead = 0;
            s_lead[r * (chunks + 1) + chunks] = 0;
            for (uint32_t k = chunks; k-- > 0;)
            {
                const unsigned long long m = s_noev[r * chunks + k];
                lead = m == ~0ull ? 64 + lead : (uint32_t)__ffsll(~m) - 1;
                s_lead[r * (chunks + 1) + k] = lead;
            }
        }
    }
    JLS_PHASE(4);
    __syncthreads();
    JLS_PHASE(5);
    // ---- P2: ranks, records.  The lanes of a chunk that share a chain find each other through a table in LDS: every lane
    // ORs its own bit into the word of its chain (the result does not depend on the order in which the LDS serves the
    // lanes), reads the word back and clears it -- 32 lanes at a time, six LDS instructions per chunk where a ballot per
    // key bit took some sixty vector and scalar ones, and this stage is bound by the instructions it issues.
    for (uint32_t sgm = wave; sgm < g.segments; sgm += kWaves)
    {
        const uint32_t r = sgm / g.pieces, piece = sgm % g.pieces;
        const uint32_t y = g.first_line + r;
        const uint32_t k0 = piece * g.chunks_per_piece;
        const uint32_t k1 = k0 + g.chunks_per_piece < chunks ? k0 + g.chunks_per_piece : chunks;
        uint32_t* segoff = s_segoff + sgm * kChains;
        uint32_t* same_of = s_same + (uint32_t)wave * (kChains + 1);
        uint16_t* inv_row = w.keyinv + (size_t)y * width; // the key of a sample is read here and its slot written in its place
        // (the keys are not kept in LDS -- 16 KB that decide between one and two workgroups per CU: they are read again,
        // two chunks ahead of their use)
        auto key_at = [&](uint32_t k) -> uint16_t {
            const uint32_t x = k * 64 + lane;
            return k < k1 && x < width ? inv_row[x] : kNoEvent;
        };
        uint16_t key_1 = key_at(k0), key_2 = key_at(k0 + 1);
        const S* cur = s_rows + (r + 1) * width; // (planar scans: the line in LDS, the line above it `width` samples before)
        const int edge_a = y >= step ? sample(y - step, 0) : 0;
        const int edge_c = y >= 2 * step ? (r >= 1 ? sample(y - 2 * step, 0) : pipe::load_sample<S, 0>(d, y - 2, 0, mask)) : 0;
        const uint32_t lane_bit = 1u << (lane & 31), below = lane_bit - 1u;
        const bool upper = lane >= 32;
        for (uint32_t k = k0; k < k1; ++k)
        {
            const uint32_t x = k * 64 + lane;
            const bool inside = x < width;
            const uint16_t key = key_1;
            key_1 = key_2;
            key_2 = key_at(k + 2);
            const bool has = key != kNoEvent;
            const uint32_t chain = key & 0x1FFu;
            if (has && !upper)
                atomicOr(&same_of[chain], lane_bit);
            JLS_LOCKSTEP();
            const uint32_t lo = has ? same_of[chain] : 0u;
            JLS_LOCKSTEP();
            if (has && !upper)
                same_of[chain] = 0;
            JLS_LOCKSTEP();
            if (has && upper)
                atomicOr(&same_of[chain], lane_bit);
            JLS_LOCKSTEP();
            const uint32_t hi = has ? same_of[chain] : 0u;
            const uint32_t base = has ? segoff[chain] : 0u;
            JLS_LOCKSTEP();
            if (has && upper)
                same_of[chain] = 0;
            const uint32_t rank = upper ? (uint32_t)__popc(lo) + (uint32_t)__popc(hi & below) : (uint32_t)__popc(lo & below);
            const uint32_t per_event = chain == 0 ? run_slots_of<S>() : 1u; // slots an event of this chain takes
            if (has && rank == 0)
                segoff[chain] = base + ((uint32_t)__popc(lo) + (uint32_t)__popc(hi)) * per_event;
            JLS_LOCKSTEP();
            const uint32_t slot = base + rank * per_event;
            // the record of a regular sample, worked out for every lane (no divergence; lanes without an event discard it)
            uint32_t record = 0;
            if (inside)
            {
                const int v = (int)cur[x] & mask;
                const int ra = x > 0 ? (int)cur[x - 1] & mask : edge_a;
                const int rb = (int)(cur - width)[x] & mask;
                const int rc = x > 0 ? (int)(cur - width)[x - 1] & mask : edge_c;
                record = make_record<S>(v, med3(ra + rb - rc, ra, rb), (key >> 9) & 1, t.maxval);
            }
            if (__any(has && chain == 0))
            { // (rare) run starts
                if (has && chain == 0)
                {
                    const int v = sample(y, x);
                    const int ra = x > 0 ? sample(y, x - 1) : edge_a;
                    uint32_t run = 0;
                    if (v == ra)
                    {
                        const unsigned long long after = lane == 63 ? 0ull : s_noev[r * chunks + k] >> (lane + 1);
                        const uint32_t rest = 63u - (uint32_t)lane;
                        uint32_t n = (uint32_t)__ffsll(~after) - 1;
                        if (n >= rest)
                            n = rest + s_lead[r * (chunks + 1) + k + 1];
                        run = 1 + n;
                    }
                    const uint32_t xi = x + run;
                    if (xi >= width)
                        record = RunRecord<S>::end_of_line(run, 0u);
                    else
                    {
                        const int xv = sample(y, xi);
                        const int ia = xi > 0 ? sample(y, xi - 1) : edge_a;
                        const int ib = y >= step ? sample(y - step, xi) : 0;
                        const int which = ia == ib ? 1 : 0;
                        const int err = which ? error_value(t, xv - ia) : error_value(t, (xv - ib) * ((ib - ia) < 0 ? -1 : 1));
                        record = RunRecord<S>::interrupted(run, err, which, 0u);
                    }
                }
            }
            if (has)
            {
                if (run_slots_of<S>() == 2 && chain == 0)
                { // a run record: 32 bits in two slots
                    s_stage[slot] = (Slot<S>)(record & 0xFFFFu);
                    s_stage[slot + 1] = (Slot<S>)(record >> 16);
                }
                else
                    s_stage[slot] = (Slot<S>)record;
            }
            if (inside)
                inv_row[x] = has ? (uint16_t)slot : kNoLocalSlot;
        }
    }
    JLS_PHASE(6);
    __syncthreads();
    JLS_PHASE(7);
    // ---- P3: pieces out (the interruption chain has no records)
    const uint32_t total_rows = s_rowbase[kChains];
    constexpr int kRows = 4;
    for (uint32_t q0 = (uint32_t)wave * kRows; q0 < total_rows; q0 += kWaves * kRows)
    { // four rows at a time: their LDS reads overlap (eight: 1.5 % slower)
        uint32_t to[kRows];
        Slot<S> held[kRows];
        bool live[kRows];
#pragma unroll
        for (int j = 0; j < kRows; ++j)
        {
            const uint32_t q = q0 + (uint32_t)j;
            const uint32_t c = q < total_rows ? s_rowchain[q] : 0u;
            const uint32_t i = (q - s_rowbase[c]) * 64 + (uint32_t)lane;
            live[j] = q < total_rows && i < s_count[c];
            to[j] = s_global[c] + i;
            held[j] = live[j] ? s_stage[s_tileoff[c] + i] : (Slot<S>)0;
        }
#pragma unroll
        for (int j = 0; j < kRows; ++j)
            if (live[j])
                rec_slots<S>(w)[to[j]] = held[j];
    }
    JLS_PHASE(8);
}

} // namespace tile
} // namespace jls
#include "tile_pixel_mode.hip"
namespace jls {
namespace tile {

// ---------------------------------------------------------------------------------------------------------------
// C: the regular-mode recurrence, src/scan_encoder_core.hpp:40-103 + src/regular_mode_context.hpp:45-136, one event.
struct Chain
{
    int a, b, c, n;
    uint32_t bad;
};

// N before event i of a chain (it counts 1..RESET, then cycles RESET/2 + 1 .. RESET).
JLS_DEV int chain_n_before(uint32_t i, uint32_t reset)
{
    if (reset == 0 || i < reset)
        return (int)(i + 1);
    const uint32_t half = reset >> 1, period = reset - half;
    return (int)(half + 1 + (i - reset) % period);
}

template <typename S>
JLS_DEV Slot<S> code_event(Chain& s, uint32_t rec, const Traits& t)
{
    int err;
    if (sizeof(S) == 1)
    { // (the record is reflected where the context's sign is negative: every event is one of positive sign, see make_record)
        const int px = med3(s.c + (int)(rec >> 8), 0, t.maxval);
        err = sign_extend((int)(rec & 0xFFu) - px, t.bpp);
    }
    else
    {
        const int sgn = ((int)rec >> 31) | 1;
        const int room = (int)((rec >> 16) & 0xFFu);
        const int sc = __mul24(s.c, sgn);
        const int delta = (rec >> 24) & 1u ? (sc < room ? sc : room) : (sc > -room ? sc : -room);
        err = sign_extend((int)(rec & 0xFFFFu) - __mul24(delta, sgn), t.bpp);
    }
    int k = regular_k(RegCtx{s.a, 0, 0, s.n});
    s.bad |= (uint32_t)(k >= 16);
    k = k > 15 ? 15 : k;
    const int corr = k == 0 ? ((2 * s.b + s.n - 1) >> 31) : 0;
    const pipe::CodeWord cw = pipe::golomb_word(t, k, map_error(corr ^ err), t.limit);
    // A.12 / A.13 in median form
    s.a += err < 0 ? -err : err;
    s.bad |= (uint32_t)(s.a >= (1 << 24));
    int tb = s.b + err;
    if (s.n == t.reset)
    {
        s.a >>= 1;
        tb >>= 1;
        s.n >>= 1;
    }
    s.n += 1;
    const int minus_delta = 1 - med3(tb, 0, 1) - med3(tb + s.n, 0, 1);
    s.b = med3(mad24(minus_delta, s.n, tb), 1 - s.n, 0);
    s.c = med3(s.c - minus_delta, -128, 127);
    return pack_code<S>(cw.len, (uint32_t)cw.bits);
}

typedef uint32_t u32x4 __attribute__((vector_size(16)));

// Which chain job `job` of a scan belongs to (job_first is non-decreasing; chains without jobs repeat their successor's value).
JLS_DEV uint32_t chain_of_job(const uint32_t* job_first, uint32_t job)
{
    uint32_t lo = 0, hi = kChains; // job_first[lo] <= job < job_first[hi]
    while (hi - lo > 1)
    {
        const uint32_t mid = (lo + hi) / 2;
        if (job_first[mid] <= job)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// C1: grid (ceil(max_jobs / 64), scans) x 64; max_jobs = samples / job_events + kChains.
//
// Memory: a lane walks its own job, 64 bytes (16 records) per round, and the 64 jobs of a wavefront lie tens of KB apart.
// Fetched by the lanes themselves that is 64 requests of 16 bytes per instruction, each a quarter of a 64-byte segment --
// the code words went out the same way, and the stage moved twice the bytes it had to (PMC, round 3).  So the wavefront
// fetches and stores TOGETHER: four neighbouring lanes cover one job's 64 bytes (16 jobs per instruction, whole segments),
// and a 5 KB table in LDS turns "16 bytes of 16 jobs" into "64 bytes of my job" and back.  The next round's records are
// requested before the current round is coded.
template <typename S>
__global__ void __launch_bounds__(64) walk_jobs(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    constexpr uint32_t kRow = 20; // words from one lane's 16 words to the next lane's (80 bytes: 16-byte aligned, banks spread)
    constexpr uint32_t kPer = 64 / (uint32_t)sizeof(Slot<S>); // events of a round: 64 bytes of records in, 64 bytes of code words out
    constexpr bool kNarrow = sizeof(Slot<S>) == 2;
    __shared__ uint32_t s_first[kChains + 1];
    __shared__ uint64_t s_in[64], s_out[64];       // where the lane's first round of records / code words is
    __shared__ uint32_t s_rounds[64], s_quiet[64]; // rounds of the lane; rounds of warm-up before its first stored one
    __shared__ __attribute__((aligned(16))) uint32_t s_swap[64 * kRow];
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const Traits t = make_traits(d);
    const uint32_t jobs = w.job_first[kChains];
    if (blockIdx.x * 64u >= jobs)
        return;
    const int lane = threadIdx.x;
    for (int c = lane; c <= kChains; c += 64)
        s_first[c] = w.job_first[c];
    __syncthreads();
    const uint32_t job = blockIdx.x * 64u + (uint32_t)lane;
    const bool live = job < jobs;
    const uint32_t chain = live ? chain_of_job(s_first, job) : 1u;
    const uint32_t n = live ? w.chain_total[chain] : 0u;
    const uint32_t start = live ? (job - s_first[chain]) * w.job_events : 0u; // multiple of kPer
    const uint32_t end = start + w.job_events < n ? start + w.job_events : n;
    const uint32_t warm = start > w.warm_events ? (start - w.warm_events) & ~(kPer - 1u) : 0u;
    const Slot<S>* in = rec_slots<S>(w) + w.chain_base[chain];
    Slot<S>* out = code_slots<S>(w) + w.chain_base[chain];
    const uint32_t rounds = end / kPer - warm / kPer, quiet = start / kPer - warm / kPer;
    s_in[lane] = (uint64_t)reinterpret_cast<uintptr_t>(in + warm);
    s_out[lane] = (uint64_t)reinterpret_cast<uintptr_t>(out + warm);
    s_rounds[lane] = rounds;
    s_quiet[lane] = quiet;
    __syncthreads();
    uint32_t most = rounds;
    for (int delta = 32; delta > 0; delta >>= 1)
    {
        const uint32_t other = __shfl_xor(most, delta);
        most = other > most ? other : most;
    }
    // the four (job, quarter) pairs this lane moves for the wavefront
    const uint32_t part = (uint32_t)lane & 3u;
    uint64_t f_in[4], f_out[4];
    uint32_t f_rounds[4], f_quiet[4], f_row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
        const uint32_t j = (uint32_t)q * 16 + ((uint32_t)lane >> 2);
        f_in[q] = s_in[j] + part * 16;
        f_out[q] = s_out[j] + part * 16;
        f_rounds[q] = s_rounds[j];
        f_quiet[q] = s_quiet[j];
        f_row[q] = j * kRow + part * 4;
    }
    auto fetch = [&](uint32_t r, u32x4 (&v)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            v[q] = r < f_rounds[q] ? *reinterpret_cast<const JLS_GLOBAL_AS u32x4*>((uintptr_t)(f_in[q] + (uint64_t)r * 64)) : u32x4{0, 0, 0, 0};
    };
    Chain s{initial_a(t), 0, 0, chain_n_before(warm, (uint32_t)t.reset), 0};
    JobState st{};
    bool started = false; // the warm-up is over: the state at the job's first event has been noted
    auto note_start = [&] {
        s.bad = 0;
        st.in_a = s.a;
        st.in_b = s.b;
        st.in_c = s.c;
        started = true;
    };
    u32x4 coming[4];
    fetch(0, coming);
    for (uint32_t r = 0; r < most; ++r)
    {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<u32x4*>(&s_swap[f_row[q]]) = coming[q];
        JLS_LOCKSTEP();
        u32x4 cur[4], o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            cur[j] = *reinterpret_cast<const u32x4*>(&s_swap[(uint32_t)lane * kRow + (uint32_t)j * 4]);
        JLS_LOCKSTEP();
        fetch(r + 1, coming);
        if (r == quiet && !started)
            note_start();
        if (r < rounds)
        {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
                    if (kNarrow)
                    { // two records in a word, two code words out
                        const uint32_t lo = (uint32_t)code_event<S>(s, cur[j][e] & 0xFFFFu, t);
                        const uint32_t hi = (uint32_t)code_event<S>(s, cur[j][e] >> 16, t);
                        o[j][e] = lo | (hi << 16);
                    }
                    else
                        o[j][e] = (uint32_t)code_event<S>(s, cur[j][e], t);
                }
        }
        if (__any(r >= quiet && r < rounds))
        { // (uniform) some lane has code words to store
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<u32x4*>(&s_swap[(uint32_t)lane * kRow + (uint32_t)j * 4]) = o[j];
            JLS_LOCKSTEP();
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r >= f_quiet[q] && r < f_rounds[q])
                    *reinterpret_cast<JLS_GLOBAL_AS u32x4*>((uintptr_t)(f_out[q] + (uint64_t)r * 64)) = *reinterpret_cast<const u32x4*>(&s_swap[f_row[q]]);
            JLS_LOCKSTEP();
        }
    }
    if (!live)
        return;
    if (!started)
        note_start();
    // ---- the last, partial group of a chain (its tail is the chain's padding)
    for (uint32_t i = end & ~(kPer - 1u); i < end; ++i)
        out[i] = code_event<S>(s, (uint32_t)in[i], t);
    st.out_a = s.a;
    st.out_b = s.b;
    st.out_c = s.c;
    st.bad = s.bad;
    st.pad = 0;
    w.jobs[job] = st;
}

// C2: grid (kChains, scans) x 64, a wavefront per (chain, scan).  The lanes compare every job's entry state with the state
// its predecessor ended in, 64 boundaries at a time; only where one differs does lane 0 go through the chain job by job and
// walk a job whose predecessor did not end in the state the job assumed again from the true state.  (A lane per chain
// looked at its boundaries one after the other until the end of round 4: a trip to memory per job, 0.3 ms for the longest
// chain of ONE frame.)
template <typename S>
__global__ void __launch_bounds__(64) settle_chains(const ScanDesc* __restrict__ descs, const Work* __restrict__ works, uint32_t scans)
{
    const uint32_t chain = blockIdx.x, frame = blockIdx.y;
    if (frame >= scans || chain == 0 || chain == (uint32_t)kInterruptChain)
        return;
    const ScanDesc d = descs[frame];
    const Work w = works[frame];
    const Traits t = make_traits(d);
    const uint32_t j0 = w.job_first[chain], j1 = w.job_first[chain + 1];
    if (j0 == j1)
        return;
    const uint32_t n = w.chain_total[chain];
    const Slot<S>* in = rec_slots<S>(w) + w.chain_base[chain];
    Slot<S>* out = code_slots<S>(w) + w.chain_base[chain];
    {
        bool differs = false;
        uint32_t any_bad = 0;
        for (uint32_t j = j0 + threadIdx.x; j < j1; j += 64)
        {
            const JobState cur = w.jobs[j];
            any_bad |= cur.bad;
            if (j > j0)
            {
                const JobState before = w.jobs[j - 1];
                differs = differs || cur.in_a != before.out_a || cur.in_b != before.out_b || cur.in_c != before.out_c;
            }
        }
        const bool some_bad = __any(any_bad != 0);
        if (!__any(differs))
        { // (the ordinary case)
            if (threadIdx.x == 0)
            {
                if (some_bad)
                    atomicOr(w.status, kStatusInvalid);
                atomicAdd(&w.counters[kCountJobs], j1 - j0);
            }
            return;
        }
        if (threadIdx.x != 0)
            return;
    }
    uint32_t bad = 0, rewalked = 0;
    JobState prev = w.jobs[j0];
    bad |= prev.bad;
    for (uint32_t j = j0 + 1; j < j1; ++j)
    {
        JobState cur = w.jobs[j];
        if (cur.in_a != prev.out_a || cur.in_b != prev.out_b || cur.in_c != prev.out_c)
        { // walk the job again from the state its predecessor really ended in
            ++rewalked;
            const uint32_t start = (j - j0) * w.job_events;
            const uint32_t end = start + w.job_events < n ? start + w.job_events : n;
            Chain s{prev.out_a, prev.out_b, prev.out_c, chain_n_before(start, (uint32_t)t.reset), 0};
            for (uint32_t i = start; i < end; ++i)
                out[i] = code_event<S>(s, (uint32_t)in[i], t);
            cur.out_a = s.a;
            cur.out_b = s.b;
            cur.out_c = s.c;
            cur.bad = s.bad;
        }
        bad |= cur.bad;
        prev = cur;
    }
    if (bad)
        atomicOr(w.status, kStatusInvalid);
    atomicAdd(&w.counters[kCountJobs], j1 - j0);
    if (rewalked)
        atomicAdd(&w.counters[kCountJobsRewalked], rewalked);
}

// C3: the run chain.  RUNindex, the two run-interruption contexts and the slot counter of the interruption samples are a
// serial recurrence over the run events of a scan (src/scan_encoder.hpp:53-73, src/scan_encoder_impl.hpp:249-275,
// src/scan_encoder_core.hpp:105-125): one lane took 19.5 ms for the 55 000 runs of a test frame -- three quarters of the
// time ONE frame takes to encode.  It is cut into jobs exactly like the regular chains:
//   * what is a function of the event INDEX is computed, not guessed: the number of interruptions of either type before
//     an event (-> N of the two contexts, chain_n_before) and the number of interruption samples with a slot of their own
//     (count_runs / scan_runs: per-job counts, exclusive prefix);
//   * what FORGETS is guessed: A and Nn of both contexts are halved every RESET/2 interruptions, RUNindex is pinned at 0 by
//     every short run (and at 31 by long ones); a job starts run_warm_events earlier from the initial values;
//   * settle_runs checks every job boundary and codes a job again, serially, where a guess was wrong.
struct RunState
{
    uint32_t index;   // RUNindex, 8 bits per component of a line-interleaved scan (src/scan_encoder_impl.hpp:126-137)
    int32_t a0, nn0;  // run-interruption context 0 (Ra != Rb)
    int32_t a1, nn1;  // context 1
};
struct RunJob
{
    uint32_t type0, type1, own_slot, pad; // count_runs: events of the job; scan_runs: events before the job
    RunState in, out;
    int32_t rare_a, rare_nn; // walk_rare_context: A and Nn of the context of the RARER interruption type when the job starts (exact)
};

JLS_DEV bool same_state(const RunState& x, const RunState& y)
{
    return x.index == y.index && x.a0 == y.a0 && x.nn0 == y.nn0 && x.a1 == y.a1 && x.nn1 == y.nn1;
}

// Code word of a run-length code: `ones` one-bits, then `tail_len` bits holding `tail`.
JLS_DEV uint32_t run_word(int ones, int tail_len, uint32_t tail)
{
    return kRunTag | ((uint32_t)ones << 25) | ((uint32_t)tail_len << 20) | tail;
}

// One lane's walk over run events [from, to) of a scan.  counts = {type-0, type-1, own-slot} interruptions before `from`.
// kStore: write the code words (the run-length code to the slot of the sample where the run starts, the code of the
// interruption sample to the next slot of chain kInterruptChain: its events are these samples, in this order).
// FMT = 1: the records of pixel mode (RunRecord2): error value and type of an interruption sample with a slot of its own are
// the record of that slot (int_rec, in the order of the slots); a sample-interleaved scan (ILV = 2) codes the `nc`
// components of an interruption pixel one after the other on run context 0 (src/scan_encoder_impl.hpp:277-302).
template <typename S, int ILV, bool kStore, int FMT = 0>
JLS_DEV void walk_runs(const Traits& t, const uint32_t* runs, uint32_t* run_code, Slot<S>* int_code, uint32_t from, uint32_t to,
                       RunState& s, uint32_t type0, uint32_t type1, uint32_t own_slot, const Slot<S>* int_rec = nullptr, uint32_t nc = 1)
{
    RunCtx rc0{0, s.a0, chain_n_before(type0, (uint32_t)t.reset), s.nn0};
    RunCtx rc1{1, s.a1, chain_n_before(type1, (uint32_t)t.reset), s.nn1};
    uint32_t run_index_packed = s.index;
    auto one = [&](uint32_t v) -> uint32_t {
        if (FMT == 1)
        {
            const bool eol = RunRecord2::is_end_of_line(v), zero = RunRecord2::is_zero_run(v);
            uint32_t run = RunRecord2::run(v);
            const uint32_t shift = ILV == 1 ? RunRecord2::component(v) * 8u : 0u;
            int run_index = (int)((run_index_packed >> shift) & 0xFFu);
            int ones = 0;
            while (run >= (1u << run_j(run_index)))
            {
                ++ones;
                run -= 1u << run_j(run_index);
                if (run_index < 31)
                    ++run_index;
            }
            uint32_t word;
            if (eol)
            {
                if (run != 0)
                    ++ones;
                word = run_word(ones, 0, 0);
            }
            else
            {
                const int jb = run_j(run_index);
                word = run_word(ones, jb + 1, run);
                for (uint32_t c = 0; c < (ILV == 2 ? nc : 1u); ++c)
                {
                    const bool shares = zero && c == 0; // the run's own sample: both codes in one word
                    const int which = ILV == 2 ? 0 : RunRecord2::which(v);
                    const int err = shares ? RunRecord2::err(v) : interruption_err<S>(int_rec[own_slot]);
                    RunCtx ctx = which ? rc1 : rc0;
                    const int k = run_k_of_encoder(ctx);
                    const int map = run_map(ctx, err, k);
                    const int em = 2 * (err < 0 ? -err : err) - ctx.ritype - map;
                    const pipe::CodeWord cw = pipe::golomb_word(t, k, em, t.limit - jb - 1);
                    run_update(ctx, err, em, t.reset);
                    if (which)
                        rc1 = ctx;
                    else
                        rc0 = ctx;
                    if (shares) // J + 1 zero bits, then the interruption code (<= LIMIT bits in all)
                        word = ((uint32_t)(jb + 1 + cw.len) << 24) | (uint32_t)cw.bits;
                    else
                    {
                        if (kStore)
                            int_code[own_slot] = pack_code<S>(cw.len, (uint32_t)cw.bits);
                        ++own_slot;
                    }
                }
                if (run_index > 0)
                    --run_index;
            }
            run_index_packed = (run_index_packed & ~(0xFFu << shift)) | ((uint32_t)run_index << shift);
            return word;
        }
        uint32_t run = RunRecord<S>::run(v);
        const bool eol = RunRecord<S>::is_end_of_line(v);
        const uint32_t shift = ILV == 1 ? RunRecord<S>::component(v) * 8u : 0u;
        int run_index = (int)((run_index_packed >> shift) & 0xFFu);
        const uint32_t full = run;
        int ones = 0;
        while (run >= (1u << run_j(run_index)))
        {
            ++ones;
            run -= 1u << run_j(run_index);
            if (run_index < 31)
                ++run_index;
        }
        uint32_t word;
        if (eol)
        {
            if (run != 0)
                ++ones;
            word = run_word(ones, 0, 0);
        }
        else
        {
            const int jb = run_j(run_index);
            const int which = RunRecord<S>::which(v);
            const int err = RunRecord<S>::err(v);
            RunCtx ctx = which ? rc1 : rc0; // (selected by value: an indexed pair of records lives in scratch)
            const int k = run_k_of_encoder(ctx);
            const int map = run_map(ctx, err, k);
            const int em = 2 * (err < 0 ? -err : err) - ctx.ritype - map;
            const pipe::CodeWord c = pipe::golomb_word(t, k, em, t.limit - jb - 1);
            run_update(ctx, err, em, t.reset);
            if (which)
                rc1 = ctx;
            else
                rc0 = ctx;
            if (run_index > 0)
                --run_index;
            if (full == 0) // both codes belong to the same sample: J + 1 zero bits, then the interruption code (<= LIMIT bits in all)
                word = ((uint32_t)(jb + 1 + c.len) << 24) | (uint32_t)c.bits;
            else
            {
                word = run_word(ones, jb + 1, run);
                if (kStore)
                    int_code[own_slot] = pack_code<S>(c.len, (uint32_t)c.bits);
                ++own_slot;
            }
        }
        run_index_packed = (run_index_packed & ~(0xFFu << shift)) | ((uint32_t)run_index << shift);
        return word;
    };
    // The events are read a group at a time, the next group requested before the current one is coded: eight events, or
    // thirty-two where only one bit of most records is looked at (little work per event: the requests have to be further
    // ahead).  Chains start on 64-byte boundaries and are followed by kSlack records, so whole groups can be read; `from` is
    // the start of a job, a multiple of the group size.
    constexpr uint32_t kQuads = 2; // 16-byte quads per group
    const JLS_GLOBAL_AS u32x4* runs4 = (const JLS_GLOBAL_AS u32x4*)runs;
    JLS_GLOBAL_AS u32x4* code4 = (JLS_GLOBAL_AS u32x4*)run_code;
    for (; from % (kQuads * 4) != 0 && from < to; ++from) // (jobs smaller than a group: the CPU tests)
    {
        const uint32_t word = one(runs[from]);
        if (kStore)
            run_code[from] = word;
    }
    uint32_t g = from / (kQuads * 4);
    u32x4 nv[kQuads];
#pragma unroll
    for (uint32_t q = 0; q < kQuads; ++q)
        nv[q] = runs4[g * kQuads + q];
    for (; g < to / (kQuads * 4); ++g)
    {
        u32x4 cv[kQuads], o[kQuads];
#pragma unroll
        for (uint32_t q = 0; q < kQuads; ++q)
        {
            cv[q] = nv[q];
            nv[q] = runs4[(g + 1) * kQuads + q];
        }
#pragma unroll
        for (uint32_t q = 0; q < kQuads; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[q][j] = one(cv[q][j]);
        if (kStore)
#pragma unroll
            for (uint32_t q = 0; q < kQuads; ++q)
                code4[g * kQuads + q] = o[q];
    }
    for (uint32_t e = g * kQuads * 4; e < to; ++e)
    {
        const uint32_t word = one(runs[e]);
        if (kStore)
            run_code[e] = word;
    }
    s.index = run_index_packed;
    s.a0 = rc0.a;
    s.nn0 = rc0.nn;
    s.a1 = rc1.a;
    s.nn1 = rc1.nn;
}

// grid (up to 32, scans) x 64: a wavefront counts the events of a job (and of every gridDim.x-th job after it).
template <typename S, int FMT = 0>
__global__ void __launch_bounds__(64) count_runs(const Work* __restrict__ works, uint32_t nc)
{
    const Work w = works[blockIdx.y];
    const uint32_t n = w.chain_total[0];
    const uint32_t* runs = reinterpret_cast<const uint32_t*>(rec_slots<S>(w) + w.chain_base[0]);
    for (uint32_t job = blockIdx.x; job * w.run_job_events < n; job += gridDim.x)
    {
        const uint32_t from = job * w.run_job_events;
        const uint32_t to = from + w.run_job_events < n ? from + w.run_job_events : n;
        uint32_t type0 = 0, type1 = 0, own = 0;
        for (uint32_t e = from + threadIdx.x; e < to; e += 64)
        {
            const uint32_t v = runs[e];
            if (FMT == 1)
            { // pixel mode; the `nc` components of an interruption pixel of a sample-interleaved scan all code on context 0
                const bool interrupted = !RunRecord2::is_end_of_line(v), zero = RunRecord2::is_zero_run(v);
                const int which = RunRecord2::which(v);
                type0 += interrupted ? (nc > 1 ? nc : (which == 0 ? 1u : 0u)) : 0u;
                type1 += interrupted && nc == 1 && which != 0;
                own += interrupted ? (zero ? nc - 1 : nc) : 0u;
                continue;
            }
            const bool interrupted = !RunRecord<S>::is_end_of_line(v);
            type0 += interrupted && RunRecord<S>::which(v) == 0;
            type1 += interrupted && RunRecord<S>::which(v) != 0;
            own += interrupted && RunRecord<S>::run(v) != 0;
        }
        for (int delta = 32; delta > 0; delta >>= 1)
        {
            type0 += __shfl_xor(type0, delta);
            type1 += __shfl_xor(type1, delta);
            own += __shfl_xor(own, delta);
        }
        if (threadIdx.x == 0)
        {
            RunJob& j = w.run_jobs[job];
            j.type0 = type0;
            j.type1 = type1;
            j.own_slot = own;
        }
    }
}

// grid (scans) x 64: counts of the jobs -> counts before the jobs.
__global__ void __launch_bounds__(64) scan_runs(const Work* __restrict__ works)
{
    const Work w = works[blockIdx.x];
    const uint32_t n = w.chain_total[0];
    const uint32_t jobs = (n + w.run_job_events - 1) / w.run_job_events;
    const int lane = threadIdx.x;
    uint32_t carry[3] = {0, 0, 0};
    for (uint32_t j0 = 0; j0 < jobs; j0 += 64)
    {
        const uint32_t j = j0 + (uint32_t)lane;
        uint32_t v[3] = {0, 0, 0};
        if (j < jobs)
        {
            v[0] = w.run_jobs[j].type0;
            v[1] = w.run_jobs[j].type1;
            v[2] = w.run_jobs[j].own_slot;
        }
        uint32_t incl[3] = {v[0], v[1], v[2]};
        for (int delta = 1; delta < 64; delta <<= 1)
            for (int q = 0; q < 3; ++q)
            {
                const uint32_t up = __shfl_up(incl[q], delta);
                if (lane >= delta)
                    incl[q] += up;
            }
        if (j < jobs)
        {
            w.run_jobs[j].type0 = carry[0] + incl[0] - v[0];
            w.run_jobs[j].type1 = carry[1] + incl[1] - v[1];
            w.run_jobs[j].own_slot = carry[2] + incl[2] - v[2];
        }
        for (int q = 0; q < 3; ++q)
            carry[q] += __shfl(incl[q], 63);
    }
    if (lane == 0)
    { // the totals, in the entry behind the last job
        w.run_jobs[jobs].type0 = carry[0];
        w.run_jobs[jobs].type1 = carry[1];
        w.run_jobs[jobs].own_slot = carry[2];
    }
}

// The context of the RARER of the two interruption types (Ra == Rb or not: 4 % of the interruptions of a test frame are of the
// rarer type) needs thousands of run events before two walks of it meet -- a warm-up of 32 768 events per job until round 4,
// 3.5 ms of the 4.9 ms ONE frame took to encode.  Its events are few, so it is computed exactly instead: compact_rare_runs
// gathers the error values of the rarer type in event order (a wavefront per job; the temporary list lives in the code words
// of the run chain, which nobody writes before walk_run_jobs), walk_rare_context walks them with ONE lane per scan and leaves
// every job the state in which the context is when the job starts.  (A sample-interleaved scan has one type only.)
JLS_DEV bool rarer_is_type1(const Work& w, uint32_t jobs)
{
    return w.run_jobs[jobs].type1 < w.run_jobs[jobs].type0;
}

// grid (up to 32, scans) x 64: a wavefront compacts a job (and every gridDim.x-th job after it).
template <typename S, int FMT>
__global__ void __launch_bounds__(64) compact_rare_runs(const Work* __restrict__ works)
{
    const Work w = works[blockIdx.y];
    const uint32_t n = w.chain_total[0];
    const uint32_t jobs = (n + w.run_job_events - 1) / w.run_job_events;
    if (jobs == 0)
        return;
    const uint32_t* runs = reinterpret_cast<const uint32_t*>(rec_slots<S>(w) + w.chain_base[0]);
    const Slot<S>* int_rec = rec_slots<S>(w) + w.chain_base[kInterruptChain]; // (pixel mode)
    int32_t* rare = reinterpret_cast<int32_t*>(w.code); // (the run chain's code words: chain 0 starts the array, one 32-bit entry per run event)
    const int rare_type = rarer_is_type1(w, jobs) ? 1 : 0;
    const int lane = threadIdx.x;
    for (uint32_t job = blockIdx.x; job < jobs; job += gridDim.x)
    {
        const uint32_t from = job * w.run_job_events;
        const uint32_t to = from + w.run_job_events < n ? from + w.run_job_events : n;
        const RunJob mine = w.run_jobs[job];
        uint32_t at = rare_type ? mine.type1 : mine.type0, own_at = mine.own_slot;
        for (uint32_t e0 = from; e0 < to; e0 += 64)
        {
            const uint32_t e = e0 + (uint32_t)lane;
            const uint32_t v = e < to ? runs[e] : 0u;
            bool is_rare, own;
            if (FMT == 1)
            {
                const bool interrupted = e < to && !RunRecord2::is_end_of_line(v);
                is_rare = interrupted && RunRecord2::which(v) == rare_type;
                own = interrupted && !RunRecord2::is_zero_run(v);
            }
            else
            {
                const bool interrupted = e < to && !RunRecord<S>::is_end_of_line(v);
                is_rare = interrupted && RunRecord<S>::which(v) == rare_type;
                own = interrupted && RunRecord<S>::run(v) != 0;
            }
            const unsigned long long rare_m = __ballot(is_rare), own_m = __ballot(own);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (is_rare)
            {
                int err;
                if (FMT == 1)
                    err = own ? interruption_err<S>(int_rec[own_at + (uint32_t)__popcll(own_m & below)]) : RunRecord2::err(v);
                else
                    err = RunRecord<S>::err(v);
                rare[at + (uint32_t)__popcll(rare_m & below)] = err;
            }
            at += (uint32_t)__popcll(rare_m);
            own_at += (uint32_t)__popcll(own_m);
        }
    }
}

// The context of the rarer interruption type, exactly, event by event: ONE lane walking a scan's events from the first to
// the last.  (Three ways of fetching the values ahead of their use were measured and are all SLOWER than this loop -- 0.48 ms
// for the test frame: 64 at a time with a broadcast per event 0.51, eight at a time in registers with the jobs' counts one job
// ahead 0.58, 1024 at a time through LDS 0.64 -- the loop is not waiting for memory.)
JLS_DEV void walk_rare_context_serially(const Traits& t, const Work& w, uint32_t jobs, bool rare1)
{
    const int32_t* rare = reinterpret_cast<const int32_t*>(w.code);
    RunCtx ctx{rare1 ? 1 : 0, initial_a(t), 1, 0};
    uint32_t i = 0;
    for (uint32_t j = 0; j < jobs; ++j)
    {
        w.run_jobs[j].rare_a = ctx.a;
        w.run_jobs[j].rare_nn = ctx.nn;
        const uint32_t upto = rare1 ? w.run_jobs[j + 1].type1 : w.run_jobs[j + 1].type0; // (entry `jobs` holds the totals)
        for (; i < upto; ++i)
        {
            const int err = rare[i];
            const int k = run_k_of_encoder(ctx);
            const int em = 2 * (err < 0 ? -err : err) - ctx.ritype - run_map(ctx, err, k);
            run_update(ctx, err, em, t.reset);
        }
    }
}

// A wavefront per scan.  Counted in ITS OWN events this context forgets as fast as any other (A and Nn are halved every
// RESET events of the type; N is a function of the event's index): it is the run events between them that made a warm-up
// expensive.  So the gathered list is cut into segments of at least kRareSegment events, a lane each, every lane warms up over
// w.rare_warm_events events of the list before its segment and notes the state its jobs start in; then every lane's entry
// state is compared with the state its predecessor ended in, and if one differs, lane 0 walks the whole list again, serially
// -- the result is exact either way.
constexpr uint32_t kRareSegment = 128;
JLS_DEV void walk_rare_context(const ScanDesc& d, const Work& w)
{
    const uint32_t lane = threadIdx.x;
    const Traits t = make_traits(d);
    const uint32_t n = w.chain_total[0];
    const uint32_t jobs = (n + w.run_job_events - 1) / w.run_job_events;
    if (jobs == 0)
        return;
    const bool rare1 = rarer_is_type1(w, jobs);
    auto starts_at = [&](uint32_t j) -> uint32_t { return rare1 ? w.run_jobs[j].type1 : w.run_jobs[j].type0; }; // events of the type before job j (entry `jobs`: all)
    const uint32_t total = starts_at(jobs);
    const uint32_t per_lane = (total + 63) / 64;
    const uint32_t segment = per_lane > kRareSegment ? per_lane : kRareSegment;
    const uint32_t s = lane * segment < total ? lane * segment : total;
    const uint32_t e = s + segment < total ? s + segment : total;
    const bool active = lane == 0 || s < total; // (lane 0 has the jobs of a scan without events of the type)
    const bool last = e == total;               // the jobs that start behind the last event are this lane's too
    const int32_t* rare = reinterpret_cast<const int32_t*>(w.code);
    const uint32_t warm_from = s > w.rare_warm_events ? s - w.rare_warm_events : 0u;
    RunCtx ctx{rare1 ? 1 : 0, initial_a(t), chain_n_before(warm_from, (uint32_t)t.reset), 0};
    auto one = [&](uint32_t i) {
        const int err = rare[i];
        const int k = run_k_of_encoder(ctx);
        const int em = 2 * (err < 0 ? -err : err) - ctx.ritype - run_map(ctx, err, k);
        run_update(ctx, err, em, t.reset);
    };
    int in_a = 0, in_nn = 0;
    if (active)
    {
        for (uint32_t i = warm_from; i < s; ++i)
            one(i);
        in_a = ctx.a;
        in_nn = ctx.nn;
        // the first job that starts at or behind event s (the counts do not decrease from job to job)
        uint32_t lo = 0, hi = jobs;
        while (lo < hi)
        {
            const uint32_t mid = (lo + hi) / 2;
            if (starts_at(mid) < s)
                lo = mid + 1;
            else
                hi = mid;
        }
        uint32_t j = lo, next = j < jobs ? starts_at(j) : ~0u;
        for (uint32_t i = s;; ++i)
        {
            while (j < jobs && next == i && (i < e || last))
            { // job j starts before event i of the type: it finds the context as it is now
                w.run_jobs[j].rare_a = ctx.a;
                w.run_jobs[j].rare_nn = ctx.nn;
                ++j;
                next = j < jobs ? starts_at(j) : ~0u;
            }
            if (i >= e)
                break;
            one(i);
        }
    }
    // does every segment start in the state the one before it ended in?  (N is right by construction.)
    const int before_a = __shfl_up(ctx.a, 1), before_nn = __shfl_up(ctx.nn, 1);
    const bool differs = active && lane != 0 && (in_a != before_a || in_nn != before_nn);
    const unsigned long long guessed = __ballot(active && lane != 0);
    const bool again = __any(differs);
    if (lane == 0)
    {
        if (guessed)
            atomicAdd(&w.counters[kCountRareSegments], (uint32_t)__popcll(guessed));
        if (again)
        {
            atomicAdd(&w.counters[kCountRareSerial], 1u);
            walk_rare_context_serially(t, w, jobs, rare1);
        }
    }
}

// The run chain's two speculative ingredients in ONE launch, side by side: the first `rare_blocks` wavefronts walk the
// context of the rarer interruption type (a wavefront per scan, exact), the others warm up a job each (a lane per (job, scan):
// RUNindex and the context of the more frequent type from run_warm_events events before the job -- what the rarer
// context is meanwhile does not matter to either, and the job takes it from the exact walk afterwards).  The two were one
// after the other until the end of round 4: 0.4 ms each of the 2.7 ms ONE frame took.
// grid (rare_blocks + ceil(max_run_jobs * scans / 64)) x 64.
template <typename S, int ILV, int FMT = 0>
__global__ void __launch_bounds__(64) warm_run_jobs(const ScanDesc* __restrict__ descs, const Work* __restrict__ works, uint32_t scans,
                                                    uint32_t rare_blocks)
{
    if (blockIdx.x < rare_blocks)
    {
        walk_rare_context(descs[blockIdx.x], works[blockIdx.x]);
        return;
    }
    const uint32_t tid = (blockIdx.x - rare_blocks) * 64u + threadIdx.x;
    const uint32_t job = tid / scans, frame = tid % scans;
    const ScanDesc d = descs[frame];
    const Work w = works[frame];
    const uint32_t n = w.chain_total[0];
    const uint32_t from = job * w.run_job_events;
    if (from >= n)
        return;
    const Traits t = make_traits(d);
    const uint32_t* runs = reinterpret_cast<const uint32_t*>(rec_slots<S>(w) + w.chain_base[0]); // (chain 0 starts the arrays: aligned)
    uint32_t* run_code = reinterpret_cast<uint32_t*>(code_slots<S>(w) + w.chain_base[0]);
    Slot<S>* int_code = code_slots<S>(w) + w.chain_base[kInterruptChain];
    const Slot<S>* int_rec = rec_slots<S>(w) + w.chain_base[kInterruptChain]; // (pixel mode)
    // The warm-up starts at a job boundary (that is where the counts are known).
    const uint32_t warm_jobs = (w.run_warm_events + w.run_job_events - 1) / w.run_job_events;
    const uint32_t warm_job = job > warm_jobs ? job - warm_jobs : 0u;
    RunState s{0, initial_a(t), 0, initial_a(t), 0};
    if (warm_job < job)
    {
        const RunJob before = w.run_jobs[warm_job];
        walk_runs<S, ILV, false, FMT>(t, runs, run_code, int_code, warm_job * w.run_job_events, from, s, before.type0, before.type1, before.own_slot, int_rec,
                                      samples_per_pixel(d));
    }
    w.run_jobs[job].in = s; // (only this field: the rarer context's walk writes others of the same entry meanwhile)
}

// grid (ceil(max_run_jobs * scans / 64)) x 64: one lane per (job, scan), lanes of a wavefront = the same job of different scans.
template <typename S, int ILV, int FMT = 0>
__global__ void __launch_bounds__(64) walk_run_jobs(const ScanDesc* __restrict__ descs, const Work* __restrict__ works, uint32_t scans)
{
    const uint32_t tid = blockIdx.x * 64u + threadIdx.x;
    const uint32_t job = tid / scans, frame = tid % scans;
    const ScanDesc d = descs[frame];
    const Work w = works[frame];
    const uint32_t n = w.chain_total[0];
    const uint32_t from = job * w.run_job_events;
    if (from >= n)
        return;
    const Traits t = make_traits(d);
    const uint32_t to = from + w.run_job_events < n ? from + w.run_job_events : n;
    const uint32_t* runs = reinterpret_cast<const uint32_t*>(rec_slots<S>(w) + w.chain_base[0]); // (chain 0 starts the arrays: aligned)
    uint32_t* run_code = reinterpret_cast<uint32_t*>(code_slots<S>(w) + w.chain_base[0]);
    Slot<S>* int_code = code_slots<S>(w) + w.chain_base[kInterruptChain];
    const Slot<S>* int_rec = rec_slots<S>(w) + w.chain_base[kInterruptChain]; // (pixel mode)
    const uint32_t nc = samples_per_pixel(d);
    const uint32_t jobs = (n + w.run_job_events - 1) / w.run_job_events;
    // the state warm_run_jobs guessed for the start of this job; the context of the rarer type is the exact one
    RunJob mine = w.run_jobs[job];
    RunState s = mine.in;
    if (ILV != 2)
    {
        if (rarer_is_type1(w, jobs))
        {
            s.a1 = mine.rare_a;
            s.nn1 = mine.rare_nn;
        }
        else
        {
            s.a0 = mine.rare_a;
            s.nn0 = mine.rare_nn;
        }
    }
    mine.in = s;
    walk_runs<S, ILV, true, FMT>(t, runs, run_code, int_code, from, to, s, mine.type0, mine.type1, mine.own_slot, int_rec, nc);
    mine.out = s;
    w.run_jobs[job] = mine;
}

// grid (scans) x 64: a wavefront per scan checks its jobs' boundaries, 64 at a time; lane 0 goes through them one by one
// only if one of them does not fit.
template <typename S, int ILV, int FMT = 0>
__global__ void __launch_bounds__(64) settle_runs(const ScanDesc* __restrict__ descs, const Work* __restrict__ works, uint32_t scans)
{
    const uint32_t frame = blockIdx.x;
    if (frame >= scans)
        return;
    const ScanDesc d = descs[frame];
    const Work w = works[frame];
    const Traits t = make_traits(d);
    const uint32_t n = w.chain_total[0];
    const uint32_t jobs = (n + w.run_job_events - 1) / w.run_job_events;
    {
        bool differs = false;
        for (uint32_t j = 1 + threadIdx.x; j < jobs; j += 64)
            differs = differs || !same_state(w.run_jobs[j].in, w.run_jobs[j - 1].out);
        if (!__any(differs))
        { // (the ordinary case)
            if (threadIdx.x == 0)
                atomicAdd(&w.counters[kCountRunJobs], jobs);
            return;
        }
        if (threadIdx.x != 0)
            return;
    }
    const uint32_t* runs = reinterpret_cast<const uint32_t*>(rec_slots<S>(w) + w.chain_base[0]);
    uint32_t* run_code = reinterpret_cast<uint32_t*>(code_slots<S>(w) + w.chain_base[0]);
    Slot<S>* int_code = code_slots<S>(w) + w.chain_base[kInterruptChain];
    const Slot<S>* int_rec = rec_slots<S>(w) + w.chain_base[kInterruptChain];
    const uint32_t nc = samples_per_pixel(d);
    RunState prev = w.run_jobs[0].out;
    uint32_t rewalked = 0;
    for (uint32_t j = 1; j < jobs; ++j)
    {
        const RunJob cur = w.run_jobs[j];
        RunState out = cur.out;
        if (!same_state(cur.in, prev))
        { // code the job again from the state its predecessor really ended in
            ++rewalked;
            out = prev;
            const uint32_t from = j * w.run_job_events;
            const uint32_t to = from + w.run_job_events < n ? from + w.run_job_events : n;
            walk_runs<S, ILV, true, FMT>(t, runs, run_code, int_code, from, to, out, cur.type0, cur.type1, cur.own_slot, int_rec, nc);
        }
        prev = out;
    }
    atomicAdd(&w.counters[kCountRunJobs], jobs);
    if (rewalked)
        atomicAdd(&w.counters[kCountRunJobsRewalked], rewalked);
}

// ---------------------------------------------------------------------------------------------------------------
// D: grid (tiles, scans) x 256; tiles in index order (a tile only waits for tiles that were started before it).
// LDS: codes[tile] u32 | tileoff / count / global [kChains + 1] each | scan[256] | tmp
#define JLS_HOST_DEV __host__ __device__ inline
// LDS of pack_tiles: where the bit buffer starts (behind the code words, the piece tables and the row table)
JLS_HOST_DEV uint32_t pack_bits_offset(uint32_t tile_capacity, uint32_t sample_bytes) // 16-byte aligned
{
    const uint32_t codes = (uint32_t)stage_bytes(tile_capacity, sample_bytes);
    const uint32_t head = ((codes + 3u) & ~3u) + 4 * ((uint32_t)kChains + 1) * 4 + kPackThreads * 4 + 16 * 4 + kRowChainWords * 4;
    return (head + 15u) & ~15u;
}

// Words of the bit buffer of pack_tiles: a sample's code has at most LIMIT = 2 (bpp + max(8, bpp)) bits (a run-length code
// longer than that stands for as many samples without a code of their own), plus the two shared boundary words.
JLS_HOST_DEV uint32_t pack_bits_words(uint32_t tile_capacity, int32_t bits_per_sample)
{
    const uint32_t limit = 2u * (uint32_t)(bits_per_sample + (bits_per_sample > 8 ? bits_per_sample : 8));
    return tile_capacity * limit / 32 + 2;
}

JLS_DEV void expand_code(uint32_t word, uint64_t& bits, int& len)
{
    if (word & kRunTag)
    {
        const int ones = (int)((word >> 25) & 63u), tail_len = (int)((word >> 20) & 31u);
        bits = ((((uint64_t)1 << ones) - 1ull) << tail_len) | (uint64_t)(word & 0xFFFFFu);
        len = ones + tail_len;
    }
    else
    {
        bits = word & 0xFFFFFFu;
        len = (int)(word >> 24);
    }
}

struct __attribute__((packed)) UnalignedQuadPair
{
    uint64_t lo, hi;
};
template <typename S>
__global__ void __launch_bounds__(kPackThreads) pack_tiles(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    JLS_DYNAMIC_LDS(smem);
    constexpr bool kNarrow = sizeof(Slot<S>) == 2;
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const uint32_t tile = blockIdx.x;
    const uint32_t tiles = scan_tiles(d, w);
    if (tile >= tiles)
        return;
    const uint32_t tile_capacity = w.tile_capacity;
    const uint32_t threads = blockDim.x; // pack_threads_for(tile_capacity): 512, fewer when a tile holds fewer than 16 samples per thread of 512
    const uint32_t per_thread = ((tile_capacity + threads - 1) / threads + 7u) & ~7u; // consecutive samples of a thread
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t kPackWaves = threads / 64;
    Slot<S>* s_code = reinterpret_cast<Slot<S>*>(smem); // the tile's code words, in slots
    const uint32_t code_bytes = (uint32_t)stage_bytes(tile_capacity, (uint32_t)sizeof(S));
    uint32_t* s_tileoff = reinterpret_cast<uint32_t*>(smem + ((code_bytes + 3u) & ~3u));
    uint32_t* s_count = s_tileoff + kChains + 1;
    uint32_t* s_global = s_count + kChains + 1;
    uint32_t* s_scan = s_global + kChains + 1; // 256
    uint32_t* s_tmp = s_scan + kPackThreads;   // one word per wavefront (16 reserved)
    uint32_t* s_rowbase = s_tmp + 16;          // [kChains + 1] first row of the chain's piece
    uint16_t* s_rowchain = reinterpret_cast<uint16_t*>(s_rowbase + kChains + 1); // [slots of the tile / 64 + kChains + 1]: kRowChainWords words
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(smem + pack_bits_offset(tile_capacity, (uint32_t)sizeof(S))); // [pack_bits_words] the tile's bits
    const TileSpan span = tile_span(d, w, tile);
    const uint32_t tile_samples = span.count;
    const uint16_t* inv = w.keyinv + span.first;

    // the slot map of this thread's samples: 16 bytes (eight slots) per load, straight into registers and requested before
    // anything else (lane by lane and two bytes at a time these were 32 requests of one cache line each per wavefront
    // instruction; round 3 staged the map through LDS instead: sixteen 2-byte loads and stores per thread)
    constexpr int kGroups = (int)(kMaxTileSamples / kPackThreads / 8); // 16 samples at most
    uint4 mine[kGroups];
    JLS_PHASE_BEGIN();
#pragma unroll
    for (int q = 0; q < kGroups; ++q)
    {
        const uint32_t at = threadIdx.x * per_thread + (uint32_t)q * 8;
        mine[q] = make_uint4(~0u, ~0u, ~0u, ~0u);
        if ((uint32_t)q * 8 < per_thread && at + 8 <= tile_samples)
        {
            const UnalignedQuadPair v = *reinterpret_cast<const UnalignedQuadPair*>(inv + at);
            mine[q] = make_uint4((uint32_t)v.lo, (uint32_t)(v.lo >> 32), (uint32_t)v.hi, (uint32_t)(v.hi >> 32));
        }
        else if ((uint32_t)q * 8 < per_thread && at < tile_samples)
        { // (the tile's last, partial group)
            uint32_t e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                e[j] = at + (uint32_t)j < tile_samples ? inv[at + j] : kNoLocalSlot;
            mine[q] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        }
    }
    { // the tile's pieces, chain by chain, in the order sort_tiles laid them out
        uint32_t n[2] = {0, 0}, g[2] = {0, 0};
        for (int half = 0; half < 2; ++half)
        {
            const uint32_t c = threadIdx.x + (uint32_t)half * threads;
            if (c < (uint32_t)kChains)
            {
                g[half] = w.seg[(size_t)tile * kChains + c];
                n[half] = w.seg[(size_t)(tile + 1) * kChains + c] - g[half];
            }
        }
        uint32_t off[2] = {n[0], n[1]};
        block_exclusive_scan(off[0], off[1], s_tmp);
        for (int half = 0; half < 2; ++half)
        {
            const uint32_t c = threadIdx.x + (uint32_t)half * threads;
            if (c < (uint32_t)kChains)
            {
                s_tileoff[c] = off[half];
                s_count[c] = n[half];
                s_global[c] = g[half];
            }
        }
    }
    JLS_PHASE(9);
    __syncthreads();
    JLS_PHASE(10);
    { // ---- the tile's code words into LDS.  The pieces are cut into rows of 64 words; row q belongs to chain s_rowchain[q].
      // A wavefront takes sixteen rows at a time and requests them together: fetched piece by piece (a piece is ~80 events on
      // average, a few are hundreds), it spent its time waiting for one trip to memory per row.
        uint32_t rows[2] = {0, 0};
        for (int half = 0; half < 2; ++half)
        {
            const uint32_t c = threadIdx.x + (uint32_t)half * threads;
            rows[half] = c < (uint32_t)kChains ? (s_count[c] + 63) / 64 : 0u;
        }
        uint32_t row_base[2] = {rows[0], rows[1]};
        block_exclusive_scan(row_base[0], row_base[1], s_tmp);
        for (int half = 0; half < 2; ++half)
        {
            const uint32_t c = threadIdx.x + (uint32_t)half * threads;
            if (c < (uint32_t)kChains)
            {
                s_rowbase[c] = row_base[half];
                for (uint32_t j = 0; j < rows[half]; ++j)
                    s_rowchain[row_base[half] + j] = (uint16_t)c;
                if (c == (uint32_t)kChains - 1)
                    s_rowbase[kChains] = row_base[half] + rows[half];
            }
        }
        __syncthreads();
        JLS_PHASE(11);
        const uint32_t total_rows = s_rowbase[kChains];
        constexpr int kRows = 16; // rows a wavefront requests together
        for (uint32_t q0 = (uint32_t)wave * kRows; q0 < total_rows; q0 += kPackWaves * kRows)
        {
            uint32_t to[kRows];
            Slot<S> held[kRows];
            bool live[kRows];
#pragma unroll
            for (int j = 0; j < kRows; ++j)
            {
                const uint32_t q = q0 + (uint32_t)j;
                const uint32_t c = q < total_rows ? s_rowchain[q] : 0u;
                const uint32_t i = (q - s_rowbase[c]) * 64 + (uint32_t)lane;
                live[j] = q < total_rows && i < s_count[c];
                to[j] = s_tileoff[c] + i;
                held[j] = live[j] ? code_slots<S>(w)[s_global[c] + i] : (Slot<S>)0;
            }
#pragma unroll
            for (int j = 0; j < kRows; ++j)
                if (live[j])
                    s_code[to[j]] = held[j];
        }
    }
    JLS_PHASE(12);
    __syncthreads();
    JLS_PHASE(13);

    // ---- bits of this thread's samples (their slots are in `mine`)
    auto slot_of = [&](int q, int j) -> uint32_t { // sample 8 q + j of this thread
        const uint32_t word = (j >> 1) == 0 ? mine[q].x : (j >> 1) == 1 ? mine[q].y : (j >> 1) == 2 ? mine[q].z : mine[q].w;
        return (j & 1) ? word >> 16 : word & 0xFFFFu;
    };
    // The code word of the sample in `slot`, in the 32-bit form (length : 8 | bits : 24, or a run-length code): the entries of
    // the run chain come first in a tile's local order and take two of the 2-byte slots.
    const uint32_t run_slots_end = kNarrow ? s_count[0] : 0u;
    auto word_of = [&](uint32_t slot) -> uint32_t {
        if (slot == kNoLocalSlot)
            return 0u; // (no bits)
        if (!kNarrow)
            return (uint32_t)s_code[slot];
        if (slot < run_slots_end)
            return (uint32_t)s_code[slot] | ((uint32_t)s_code[slot + 1] << 16);
        const uint32_t c = (uint32_t)s_code[slot];
        return ((c >> 10) << 24) | (c & 0x3FFu);
    };
    // (the code words of a thread's samples stay in registers for the second pass: one LDS read per sample, not two)
    uint32_t words[kGroups * 8];
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < kGroups; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            const uint32_t word = word_of(slot_of(q, j));
            words[q * 8 + j] = word;
            sum += word & kRunTag ? ((word >> 25) & 63u) + ((word >> 20) & 31u) : word >> 24;
        }
    JLS_PHASE(14);
    { // inclusive scan of the threads' bit counts: inside a wavefront with shuffles, the wavefronts' totals through LDS (two
      // barriers; the Hillis-Steele scan through LDS that stood here until round 4 took eighteen)
        uint32_t incl = sum;
        for (int delta = 1; delta < 64; delta <<= 1)
        {
            const uint32_t up = __shfl_up(incl, delta);
            if (lane >= delta)
                incl += up;
        }
        if (lane == 63)
            s_tmp[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w2 = 0; w2 < kPackWaves; ++w2)
            before += (int)w2 < wave ? s_tmp[w2] : 0u;
        s_scan[threadIdx.x] = before + incl;
        __syncthreads();
    }
    JLS_PHASE(15);
    // ---- where this tile starts: the first wavefront looks back, 64 predecessors at a time (see write_raw_bits)
    if (threadIdx.x < 64)
    {
        const uint64_t own = s_scan[threads - 1];
        const uint32_t b = tile;
        if (lane == 0)
            store_relaxed(&w.blockbase[b], (b == 0 ? pipe::kBlockUpTo : pipe::kBlockOwn) | own);
        uint64_t start = 0;
        uint32_t reach = b; // tiles [reach, b) are accounted for in `start`
        while (reach > 0)
        {
            const bool in_window = (uint32_t)lane < reach;
            const uint32_t j = in_window ? reach - 1 - (uint32_t)lane : 0;
            uint64_t state = 0;
            do
            {
                state = in_window ? load_relaxed(&w.blockbase[j]) : pipe::kBlockOwn;
            } while (__any((state >> 62) == 0));
            const unsigned long long knows = __ballot(in_window && (state >> 62) == 2);
            const int last = knows ? (int)__ffsll(knows) - 1 : 63;
            uint64_t part = in_window && lane <= last ? state & pipe::kBlockValue : 0;
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
                store_relaxed(&w.blockbase[b], pipe::kBlockUpTo | (start + own));
            s_tmp[12] = (uint32_t)start; // (two words of the scan scratch: the kernel has no static LDS, its dynamic
            s_tmp[13] = (uint32_t)(start >> 32); // region may then be as large as the CU's)
            if (b + 1 == tiles)
                *w.total_bits = start + own;
        }
    }
    JLS_PHASE(16);
    __syncthreads();
    JLS_PHASE(17);
    // ---- the tile's bits are put together in LDS and leave as whole
    // words, coalesced.  Nothing of the raw stream is cleared beforehand and no word is written twice: the last, partial
    // word of a tile is not stored by that tile but PUBLISHED (tile_tail: the bits and a valid flag in one 64-bit word, like
    // the look-back states -- the word carries everything, no fence), and the next tile, whose first bits complete it, ORs
    // it into its own first word before storing that.  A tile publishes as soon as its bits are in LDS and only then waits
    // for its predecessor (the waits do not form a chain, except through tiles of less than a word).  (Round 2 cleared the
    // whole buffer, 17.8 MB per frame for 7.3 MB of stream, and every thread wrote its two or three words with atomics:
    // 48 MB of write traffic.)
    const uint64_t tile_start = (uint64_t)s_tmp[12] | ((uint64_t)s_tmp[13] << 32);
    const uint32_t tile_bits = s_scan[threads - 1];
    const uint32_t head = (uint32_t)(tile_start & 31);
    const uint32_t tile_words = (head + tile_bits + 31) / 32; // (<= pack_bits_words: a code has at most LIMIT bits per sample it stands for)
    for (uint32_t i = threadIdx.x; i < tile_words + 1; i += threads)
        s_bits[i] = 0;
    __syncthreads();
    if (sum != 0)
    {
        const uint32_t bitpos = head + s_scan[threadIdx.x] - sum;
        uint32_t word_at = bitpos >> 5;
        const uint32_t first_word = word_at;
        uint32_t pending = bitpos & 31; // the leading bits of the first word belong to the previous thread
        uint64_t acc = 0;               // bits [63 - pending, ...) downwards are ours
        // n <= 32 bits of `part` behind what is pending; a full 32-bit word leaves at once
        auto put = [&](uint32_t part, uint32_t n) {
            acc |= (uint64_t)part << ((64u - pending - n) & 63u);
            pending += n;
            if (pending >= 32)
            {
                const uint32_t o = (uint32_t)(acc >> 32);
                if (word_at == first_word)
                    atomicOr(&s_bits[word_at], o); // shared with the previous thread's tail
                else
                    s_bits[word_at] = o; // entirely ours
                acc <<= 32;
                pending -= 32;
                ++word_at;
            }
        };
#pragma unroll
        for (int q = 0; q < kGroups; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                const uint32_t word = words[q * 8 + j];
                if (word == 0)
                    continue; // (a sample without a code of its own)
                uint64_t v;
                int len;
                expand_code(word, v, len);
                if (len > 32)
                { // (codes of wide samples, long runs: rare)
                    put((uint32_t)(v >> 32), (uint32_t)len - 32u);
                    len = 32;
                }
                put((uint32_t)v, (uint32_t)len);
            }
        if (pending > 0)
            atomicOr(&s_bits[word_at], (uint32_t)(acc >> 32)); // tail shared with the next thread
    }
    JLS_PHASE(18);
    __syncthreads();
    JLS_PHASE(19);
    const bool last_tile = tile + 1 == tiles;
    const bool shared_first = head != 0;                                     // my first word starts in the tile before
    const bool partial_last = ((head + tile_bits) & 31u) != 0 && !last_tile; // my last word is completed by the next tile
    constexpr uint64_t kTailValid = 1ull << 63;
    if (threadIdx.x == 0)
    {
        // (with no word of its own -- tile_words 0, or one word that neither starts nor ends here -- a tile passes on what its
        // predecessor left, plus its own bits)
        const bool own_tail = tile_words >= 2 || !shared_first;
        if (own_tail)
            store_relaxed(&w.tile_tail[tile], kTailValid | (partial_last ? s_bits[tile_words - 1] : 0u));
        if (shared_first)
        { // (tile 0 starts at bit 0)
            uint64_t before;
            do
                before = load_relaxed(&w.tile_tail[tile - 1]);
            while ((before & kTailValid) == 0);
            s_bits[0] |= (uint32_t)before;
        }
        if (!own_tail)
            store_relaxed(&w.tile_tail[tile], kTailValid | (partial_last ? s_bits[0] : 0u));
    }
    __syncthreads();
    JLS_PHASE(20);
    const uint64_t first_global = tile_start >> 5;
    const uint32_t stored_words = partial_last ? tile_words - 1 : tile_words;
    for (uint32_t i = threadIdx.x; i < stored_words; i += threads)
    {
        const uint64_t at = first_global + i;
        if (at < w.raw_words)
            w.raw[at] = __builtin_bswap32(s_bits[i]);
    }
    if (last_tile && threadIdx.x < 4)
    { // the stuffing stage reads a few bytes past the last bit: zeros
        const uint64_t at = first_global + tile_words + threadIdx.x;
        if (at < w.raw_words)
            w.raw[at] = 0;
    }
    JLS_PHASE(21);
}

// Zeroes the look-back states and the tile tails of every scan of a pass (contiguous in a work area).
__global__ void __launch_bounds__(256) clear_pack_state(const Work* __restrict__ works, uint32_t bytes_per_scan)
{
    uint4* at = reinterpret_cast<uint4*>(works[blockIdx.y].blockbase);
    const uint32_t groups = bytes_per_scan / 16;
    for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < groups; g += gridDim.x * 256u)
        at[g] = make_uint4(0, 0, 0, 0);
}

// LDS bytes of the tile kernels for lines of `width` samples of `sample_bytes` bytes, `lines_per_tile` lines per tile.
inline size_t tile_common_lds_bytes(uint32_t width, uint32_t lines_per_tile, uint32_t sample_bytes, int interleave_mode, bool with_keys)
{
    auto up = [](size_t v) { return (v + 15) & ~size_t{15}; };
    const size_t chunks = (width + 63) / 64;
    return up(interleave_mode == 1 ? 0 : (size_t)(lines_per_tile + 1) * width * sample_bytes) +
           (with_keys ? up((size_t)lines_per_tile * width * 2) : 0) + up((size_t)lines_per_tile * chunks * 16);
}
inline size_t analyze_lds_bytes(uint32_t width, uint32_t lines_per_tile, uint32_t sample_bytes, int interleave_mode)
{
    return tile_common_lds_bytes(width, lines_per_tile, sample_bytes, interleave_mode, true) + ((size_t)kChains + 1) * 4 + pipe::kGradientTable;
}
inline size_t sort_lds_bytes(uint32_t width, uint32_t lines_per_tile, uint32_t sample_bytes, int interleave_mode)
{
    return tile_common_lds_bytes(width, lines_per_tile, sample_bytes, interleave_mode, false) + (size_t)sort_segments(lines_per_tile) * kChains * 4 +
           4 * ((size_t)kChains + 1) * 4 + 16 * 4 + (size_t)kWaves * (kChains + 1) * 4 + kRowChainWords * 4 +
           stage_bytes((uint32_t)lines_per_tile * width, sample_bytes);
}
// Threads of a pack_tiles workgroup: 512 with 8 or 16 consecutive samples each; a tile that would leave more than a fifth of
// those samples empty (6144 samples of a 4096-pixel RGB line cut in two) gets fewer threads instead (never fewer than 192: the
// 367 chains are dealt to the threads two at a time).
inline uint32_t pack_threads_for(uint32_t tile_capacity)
{
    const uint32_t per_thread = ((tile_capacity + kPackThreads - 1) / kPackThreads + 7u) & ~7u;
    if ((uint64_t)per_thread * kPackThreads * 4 <= (uint64_t)tile_capacity * 5)
        return kPackThreads;
    const uint32_t fewer = ((tile_capacity + per_thread - 1) / per_thread + 63u) / 64u * 64u;
    return fewer < 192u ? 192u : (fewer > kPackThreads ? kPackThreads : fewer);
}
inline size_t pack_lds_bytes(uint32_t tile_capacity, int32_t bits_per_sample)
{
    return (size_t)pack_bits_offset(tile_capacity, bits_per_sample > 8 ? 2u : 1u) + (size_t)pack_bits_words(tile_capacity, bits_per_sample) * 4;
}
// How a scan is cut into tiles.  A tile holds up to `cap` samples (8192 of one byte, 4096 of two: the sort stage keeps the
// lines, the sorted records and its tables of a tile in LDS; CHARLS_AMD_TILE_SAMPLES lowers it -- more workgroups per CU for
// measurements, segment tiles on small images for tests).  Lines that fit: as many whole lines as fit, at most kTileLines.
// Lines that do not: every line is cut into segs_per_line segments of seg_pixels pixels (a multiple of 64, the last one
// shorter).  mode: 0 = planar whole lines (analyze_tiles<S, 0>), 1 = line-interleaved whole lines (<S, 1>), 2 = pixel mode
// (tile_pixel_mode.hip: sample-interleaved scans, and any scan whose lines do not fit a tile; CHARLS_AMD_PIXEL_MODE=1 sends
// every scan there -- for tests).
struct TilePlan
{
    uint32_t mode, nc, step, lines, line_samples, lines_per_tile, segs_per_line, seg_pixels, tiles, tile_capacity, max_pixels;
    uint64_t samples;
};
inline TilePlan plan_tiles(const ScanDesc& d)
{
    TilePlan p{};
    const uint32_t sample_bytes = d.bits_per_sample > 8 ? 2u : 1u;
    uint32_t cap = sample_bytes == 1 ? kMaxTileSamples : kMaxTileSamples / 2;
    if (const long long knob = knobs::get(knobs::kTileSamples); knob != knobs::kUnset)
        cap = std::min<uint32_t>(cap, (uint32_t)std::max<long long>(64, std::min<long long>(knob, cap)));
    const bool force_pixel_mode = knobs::get_or(knobs::kPixelMode, 0) != 0;
    p.nc = d.interleave_mode == 2 ? (uint32_t)d.components : 1u;
    p.step = d.interleave_mode == 1 ? (uint32_t)d.components : 1u; // distance of a coded line to the line above it
    p.lines = d.height * (d.interleave_mode == 1 ? (uint32_t)d.components : 1u);
    p.line_samples = d.width * p.nc;
    p.samples = (uint64_t)p.line_samples * p.lines;
    if (p.line_samples <= cap)
    {
        const uint32_t n = cap / p.line_samples;
        p.lines_per_tile = n < 1 ? 1 : (n > kTileLines ? kTileLines : n);
        p.segs_per_line = 1;
        p.seg_pixels = d.width;
        p.tile_capacity = p.lines_per_tile * p.line_samples;
        p.mode = d.interleave_mode != 0 || force_pixel_mode ? 2u : 0u;
    }
    else
    {
        p.lines_per_tile = 1;
        const uint32_t fit = std::min(cap, p.line_samples) / p.nc / 64 * 64;
        const uint32_t max_px = fit < 64 ? 64 : fit;
        p.segs_per_line = (d.width + max_px - 1) / max_px;
        p.seg_pixels = ((d.width + p.segs_per_line - 1) / p.segs_per_line + 63) / 64 * 64;
        p.tile_capacity = p.seg_pixels * p.nc;
        p.mode = 2;
    }
    p.tiles = (p.lines + p.lines_per_tile - 1) / p.lines_per_tile * p.segs_per_line;
    p.max_pixels = p.segs_per_line == 1 ? d.width : p.seg_pixels;
    return p;
}
} // namespace tile
} // namespace jls
