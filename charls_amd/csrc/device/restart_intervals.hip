// restart_intervals.hip -- restart intervals as the unit of parallelism INSIDE one scan.
//
// At a restart marker the reference's decoder resets everything a scan starts with: contexts, RUNindex, the previous
// line (zeros) and the bit reader (src/scan_decoder_impl.hpp:119-127, src/scan_decoder.hpp:237-243,335-349).  The lines
// of one restart interval are therefore coded exactly like an independent scan of that many lines, and the intervals of
// a scan can be coded by different wavefronts at the same time.
//
//   decode:  find_restart_markers  one wavefront per scan walks the entropy-coded bytes once (0xFF followed by a byte
//                                  >= 0x80 can only be a marker: that is what bit stuffing guarantees) and lists the
//                                  RSTm positions up to the first other marker
//            build_decode_intervals one ScanDesc per interval (rows, stream window up to and including its RSTm)
//            ... the ordinary decoders run on the intervals ...
//            check_intervals        every interval must have ended exactly at its marker and the markers must count
//                                   0..7 cyclically; anything else is left to the sequential exact decoder, which
//                                   reproduces the reference's error codes
//   encode (extension, the reference's encoder cannot emit restart markers):
//            build_encode_intervals intervals are coded into private buffers,
//            plan_join / join_intervals  then concatenated with FF D0+m between them
#pragma once
#include <hip/hip_runtime.h>

#include "scan_model.h"

namespace jls {
namespace interval {

constexpr uint32_t kIntervalRetry = 8u; // ScanResult.flags: decode this scan sequentially
constexpr uint32_t kTooManyMarkers = 0xFFFFFFFFu;
constexpr uint32_t kUndecided = 4u | kIntervalRetry; // fast::kFastRetry or kIntervalRetry left in an interval's result

// One wavefront per scan.  marks[scan * max_marks + j] = offset (from desc.stream) of the 0xFF of the j-th RSTm marker;
// counts[scan] = number of RSTm markers before the first other marker (kTooManyMarkers when more than max_marks).
__global__ void __launch_bounds__(64) find_restart_markers(const ScanDesc* __restrict__ descs, uint32_t* __restrict__ marks,
                                                           uint32_t max_marks, uint32_t* __restrict__ counts)
{
    const ScanDesc d = descs[blockIdx.x];
    const int lane = threadIdx.x;
    const uint64_t mis = (uint64_t)(reinterpret_cast<uintptr_t>(d.stream) & 15u);
    const uint8_t* gbase = d.stream - mis; // 16-byte aligned
    const uint64_t u_begin = mis, u_end = mis + d.stream_capacity;
    uint32_t* out = marks + (size_t)blockIdx.x * max_marks;
    uint32_t found = 0;
    bool overflow = false;
    for (uint64_t u0 = 0; u0 < u_end; u0 += 1024)
    {
        const uint64_t mine = u0 + (uint64_t)lane * 16;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (mine < u_end)
            raw = *reinterpret_cast<const uint4*>(gbase + mine);
        const uint32_t words[4] = {raw.x, raw.y, raw.z, raw.w};
        uint32_t next_first = 0; // the byte after this lane's 16
        if (mine + 16 < u_end)
            next_first = gbase[mine + 16];
        uint32_t rst_mask = 0, other_mask = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            const uint32_t b = (words[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
            const uint32_t nb = j < 15 ? ((words[(j + 1) >> 2] >> (((j + 1) & 3) * 8)) & 0xFFu) : next_first;
            const uint64_t u = mine + (uint64_t)j;
            const bool inside = u >= u_begin && u + 1 < u_end;
            const bool is_marker = inside && b == 0xFFu && nb >= 0x80u;
            const bool is_rst = is_marker && (nb & 0xF8u) == 0xD0u;
            rst_mask |= (uint32_t)is_rst << j;
            other_mask |= (uint32_t)(is_marker && !is_rst) << j;
        }
        // the first non-RST marker ends the entropy-coded segment
        const unsigned long long lanes_other = __ballot(other_mask != 0);
        int stop_lane = 64, stop_bit = 16;
        if (lanes_other)
        {
            stop_lane = __ffsll(lanes_other) - 1;
            stop_bit = __ffs((int)__shfl((int)other_mask, stop_lane)) - 1;
        }
        if (lane > stop_lane)
            rst_mask = 0;
        else if (lane == stop_lane)
            rst_mask &= (1u << stop_bit) - 1u;
        // ordered append
        const int mine_count = __popc(rst_mask);
        int inc = mine_count;
        for (int delta = 1; delta < 64; delta <<= 1)
        {
            const int up = __shfl_up(inc, delta);
            if (lane >= delta)
                inc += up;
        }
        uint32_t at = found + (uint32_t)(inc - mine_count);
        uint32_t m = rst_mask;
        while (m)
        {
            const int j = __ffs((int)m) - 1;
            m &= m - 1;
            if (at < max_marks)
                out[at] = (uint32_t)(mine + (uint64_t)j - u_begin);
            ++at;
        }
        found += (uint32_t)__shfl(inc, 63);
        if (found > max_marks)
            overflow = true;
        if (lanes_other)
            break;
    }
    if (lane == 0)
        counts[blockIdx.x] = overflow ? kTooManyMarkers : found;
}

// grid (intervals, scans), one thread: the interval as a scan of its own.
__global__ void build_decode_intervals(const ScanDesc* __restrict__ parents, const uint32_t* __restrict__ marks,
                                       uint32_t intervals, ScanDesc* __restrict__ subs)
{
    const uint32_t j = blockIdx.x, s = blockIdx.y;
    ScanDesc d = parents[s];
    const uint32_t* mk = marks + (size_t)s * (intervals - 1);
    const uint32_t lines = d.restart_interval;
    const uint64_t start = j == 0 ? 0 : (uint64_t)mk[j - 1] + 2;
    const uint64_t end = j + 1 < intervals ? (uint64_t)mk[j] + 2 : d.stream_capacity; // its own RSTm terminates an interval
    d.pixels += (uint64_t)j * lines * d.pixel_stride;
    // rows of this interval; a parent without rows (a frame that already failed) has only empty intervals
    const uint64_t before = (uint64_t)j * lines;
    d.height = d.height > before ? (uint32_t)(d.height - before < lines ? d.height - before : lines) : 0u;
    d.stream += start;
    d.stream_capacity = end > start ? end - start : 0;
    d.restart_interval = 0;
    subs[(size_t)s * intervals + j] = d;
}

// One thread per scan.
__global__ void check_intervals(const ScanDesc* __restrict__ parents, const uint32_t* __restrict__ marks, uint32_t intervals,
                                const ScanResult* __restrict__ sub_results, ScanResult* __restrict__ results, uint32_t scans)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= scans)
        return;
    const ScanDesc d = parents[s];
    const uint32_t* mk = marks + (size_t)s * (intervals - 1);
    const ScanResult* sr = sub_results + (size_t)s * intervals;
    bool ok = true;
    uint64_t start = 0;
    for (uint32_t j = 0; j < intervals; ++j)
    {
        ok = ok && sr[j].errc == kOk && (sr[j].flags & kUndecided) == 0;
        if (j + 1 < intervals)
        { // consumed exactly up to its marker, and the marker is the expected one
            ok = ok && sr[j].bytes == (uint64_t)mk[j] - start && d.stream[(uint64_t)mk[j] + 1] == 0xD0u + (j & 7u);
            start = (uint64_t)mk[j] + 2;
        }
    }
    ScanResult r{kOk, 0, 0};
    if (ok)
        r.bytes = start + sr[intervals - 1].bytes;
    else
        r.flags = kIntervalRetry;
    results[s] = r;
}

// ---- encode -------------------------------------------------------------------------------------------------------

// grid (intervals, scans), one thread.  `buffers` holds scans * intervals private streams of `capacity` bytes;
// `scratch` (may be null) scans * intervals line windows of `scratch_samples` samples for the global-memory kernels.
__global__ void build_encode_intervals(const ScanDesc* __restrict__ parents, uint32_t intervals, uint8_t* __restrict__ buffers,
                                       uint64_t capacity, uint16_t* __restrict__ scratch, uint64_t scratch_samples,
                                       ScanDesc* __restrict__ subs)
{
    const uint32_t j = blockIdx.x, s = blockIdx.y;
    ScanDesc d = parents[s];
    const uint32_t lines = d.restart_interval;
    const size_t at = (size_t)s * intervals + j;
    d.pixels += (uint64_t)j * lines * d.pixel_stride;
    // rows of this interval.  A parent with height 0 is a frame that failed earlier (container_kernels.hip:
    // place_scan_header empties its descriptor): all its intervals are empty scans with no room, which fail fast without
    // touching pixels or work areas -- the geometry of the launch (`intervals`) comes from the other frames.
    const uint64_t before = (uint64_t)j * lines;
    const bool dead = d.height == 0 || d.stream_capacity == 0;
    d.height = !dead && d.height > before ? (uint32_t)(d.height - before < lines ? d.height - before : lines) : 0u;
    d.stream = buffers + at * capacity;
    d.stream_capacity = dead ? 0 : capacity;
    d.restart_interval = 0;
    if (scratch != nullptr)
        d.line_scratch = scratch + at * scratch_samples;
    subs[at] = d;
}

// One thread per scan: where every interval goes in the scan's own stream; the scan's result.
// flags bit kIntervalRetry: an interval did not fit ITS private buffer (the caller repeats with worst-case buffers).
__global__ void plan_join(const ScanDesc* __restrict__ parents, uint32_t intervals, const ScanResult* __restrict__ sub_results,
                          uint64_t* __restrict__ offsets, ScanResult* __restrict__ results, uint32_t scans)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= scans)
        return;
    const ScanDesc d = parents[s];
    const ScanResult* sr = sub_results + (size_t)s * intervals;
    uint64_t* off = offsets + (size_t)s * intervals;
    if (d.height == 0 || d.stream_capacity == 0)
    { // a frame that failed before this scan: nothing to join, and no reason to repeat its group with larger buffers
        for (uint32_t j = 0; j < intervals; ++j)
            off[j] = 0;
        results[s] = ScanResult{kDestinationTooSmall, 0, 0};
        return;
    }
    ScanResult r{kOk, 0, 0};
    uint64_t at = 0;
    for (uint32_t j = 0; j < intervals; ++j)
    {
        if (sr[j].errc == kDestinationTooSmall)
            r.flags |= kIntervalRetry;
        else if (sr[j].errc != kOk && r.errc == kOk)
            r.errc = sr[j].errc;
        off[j] = at;
        at += sr[j].bytes + (j + 1 < intervals ? 2 : 0);
    }
    if (r.errc == kOk && r.flags == 0 && at > d.stream_capacity)
        r.errc = kDestinationTooSmall;
    r.bytes = r.errc == kOk && r.flags == 0 ? at : 0;
    results[s] = r;
}

// grid (intervals, scans) x 256 threads: copy + marker.
__global__ void __launch_bounds__(256) join_intervals(const ScanDesc* __restrict__ parents, const ScanDesc* __restrict__ subs,
                                                      uint32_t intervals, const ScanResult* __restrict__ sub_results,
                                                      const uint64_t* __restrict__ offsets, const ScanResult* __restrict__ results)
{
    const uint32_t j = blockIdx.x, s = blockIdx.y;
    const ScanResult r = results[s];
    if (r.errc != kOk || r.flags != 0)
        return;
    const size_t at = (size_t)s * intervals + j;
    const uint8_t* src = subs[at].stream;
    uint8_t* dst = parents[s].stream + offsets[at];
    const uint64_t n = sub_results[at].bytes;
    for (uint64_t i = threadIdx.x; i < n; i += 256)
        dst[i] = src[i];
    if (threadIdx.x == 0 && j + 1 < intervals)
    {
        dst[n] = 0xFF;
        dst[n + 1] = (uint8_t)(0xD0u + (j & 7u));
    }
}

} // namespace interval
} // namespace jls
