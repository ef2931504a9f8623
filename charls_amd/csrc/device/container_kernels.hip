// container_kernels.hip -- placement of marker segments around device-resident entropy-coded segments (batch API).
//
// Every frame of a batch owns one fixed-pitch slot that will hold its complete .jls file.  A per-frame byte cursor lives
// in device memory, so scans are written straight to their final position with exactly the capacity the reference's
// writer would hand to scan_encoder::encode_scan (src/charls_jpegls_encoder.cpp:285-296: `writer_.remaining_destination()`):
//   place_prologue : SOI .. LSE bytes (identical for all frames of a batch)            -> cursor = prologue size
//   place_scan_header (per scan round): SOS bytes at the cursor, patch ScanDesc.stream/.stream_capacity
//   <scan kernels>
//   advance_cursor : cursor += SOS + entropy bytes, latch the first error of the frame
//   place_epilogue : optional 0xFF fill + EOI (src/jpeg_stream_writer.cpp:26-35)       -> sizes[f]
#pragma once
#include <hip/hip_runtime.h>

#include "scan_types.h"

namespace jls {

struct FrameCursor
{
    uint64_t offset; // bytes of the slot already written
    uint32_t errc;   // first error of this frame (charls_jpegls_errc), 0 while fine
    uint32_t pad;
};

__global__ void place_prologue(uint8_t* __restrict__ slots, uint64_t slot_pitch, const uint8_t* __restrict__ prologue,
                               uint32_t prologue_size, FrameCursor* __restrict__ cursors, uint32_t frames)
{
    const uint32_t f = blockIdx.x;
    if (f >= frames)
        return;
    if (prologue_size > slot_pitch)
    { // the reference's writer refuses a segment that does not fit (src/jpeg_stream_writer.cpp:229-243)
        if (threadIdx.x == 0)
            cursors[f] = FrameCursor{0, kDestinationTooSmall, 0};
        return;
    }
    uint8_t* dst = slots + (uint64_t)f * slot_pitch;
    for (uint32_t i = threadIdx.x; i < prologue_size; i += blockDim.x)
        dst[i] = prologue[i];
    if (threadIdx.x == 0)
        cursors[f] = FrameCursor{prologue_size, kOk, 0};
}

// descs[f] describes the scan of this round for frame f; header = SOS segment bytes (same for all frames).
__global__ void place_scan_header(uint8_t* __restrict__ slots, uint64_t slot_pitch, const uint8_t* __restrict__ header,
                                  uint32_t header_size, FrameCursor* __restrict__ cursors, ScanDesc* __restrict__ descs,
                                  uint32_t frames)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames)
        return;
    FrameCursor c = cursors[f];
    uint8_t* slot = slots + (uint64_t)f * slot_pitch;
    if (c.errc == kOk && c.offset + header_size > slot_pitch)
    {
        c.errc = kDestinationTooSmall;
        cursors[f] = c;
    }
    if (c.errc != kOk)
    { // make the scan kernel a no-op that fails fast without touching memory
        descs[f].stream = slot;
        descs[f].stream_capacity = 0;
        descs[f].height = 0;
        return;
    }
    for (uint32_t i = 0; i < header_size; ++i)
        slot[c.offset + i] = header[i];
    descs[f].stream = slot + c.offset + header_size;
    descs[f].stream_capacity = slot_pitch - c.offset - header_size;
}

__global__ void advance_cursor(FrameCursor* __restrict__ cursors, const ScanResult* __restrict__ results,
                               uint32_t header_size, uint32_t frames)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames)
        return;
    FrameCursor c = cursors[f];
    if (c.errc != kOk)
        return;
    const ScanResult r = results[f];
    if (r.errc != kOk)
        c.errc = r.errc;
    else
        c.offset += header_size + r.bytes;
    cursors[f] = c;
}

// The component scans of planar frames coded TOGETHER (one launch for all of them, into private buffers of `capacity` bytes:
// scan r of frame f at private + (f * rounds + r) * capacity) are put where the reference's writer would have had them coded:
// one workgroup per frame walks its scans in order -- SOS header (header_size bytes, the r-th of `headers`), then the scan's
// bytes -- and stops at the first scan that failed or that leaves fewer than 4 bytes behind it: there the reference's verdict
// depends on the capacity it passes to THAT scan (src/scan_encoder.hpp:117-120), so the frame is marked (redo[f] = 1) and
// coded again scan by scan by the host.
struct __attribute__((packed)) UnalignedWord
{
    uint32_t v;
};
// Where the scans of frame f go: scan r behind its header at `at[r]`, or the frame is coded again (`again`).  The same
// walk for every workgroup of the frame and for the kernel that advances the cursor afterwards.
struct PlanePlan
{
    FrameCursor end; // the cursor behind the last scan that is placed
    bool again;
};
__device__ inline PlanePlan plan_plane_scans(const FrameCursor& start, uint64_t slot_pitch, uint32_t header_size, uint32_t rounds,
                                             const ScanResult* __restrict__ results, uint32_t upto, uint64_t& at_upto)
{
    PlanePlan p{start, false};
    at_upto = ~0ull;
    for (uint32_t r = 0; r < rounds && p.end.errc == kOk && !p.again; ++r)
    {
        if (p.end.offset + header_size > slot_pitch)
        {
            p.end.errc = kDestinationTooSmall;
            break;
        }
        const ScanResult res = results[r];
        const uint64_t remaining = slot_pitch - p.end.offset - header_size;
        if (res.errc != kOk || res.bytes + 4 > remaining)
        { // (a scan that failed in its private buffer may still fail differently -- or not at all -- in place)
            p.again = true;
            break;
        }
        if (r == upto)
            at_upto = p.end.offset;
        p.end.offset += header_size + res.bytes;
    }
    return p;
}
struct __attribute__((packed)) UnalignedQuad
{
    uint64_t lo, hi;
};
// grid (frames * rounds, kPlaceShares) x 256: workgroup (f, r, share) copies its share of scan r of frame f from the private
// buffer to its place, 16 bytes per lane and trip (share 0 also writes the scan's header).  A frame one of whose scans
// cannot be placed is left alone -- it is coded again, scan by scan -- and flagged by advance_plane_cursors.  (One
// workgroup per FRAME copying word by word took 20 % of the coding time of 256 4096 x 4096 RGB frames.)
constexpr uint32_t kPlaceShares = 16;
__global__ void __launch_bounds__(256) place_plane_scans(uint8_t* __restrict__ slots, uint64_t slot_pitch, const uint8_t* __restrict__ headers,
                                                         uint32_t header_size, uint32_t rounds, const uint8_t* __restrict__ private_streams,
                                                         uint64_t capacity, const ScanResult* __restrict__ results,
                                                         const FrameCursor* __restrict__ cursors)
{
    const uint32_t f = blockIdx.x / rounds, r = blockIdx.x % rounds;
    uint64_t at;
    const PlanePlan plan = plan_plane_scans(cursors[f], slot_pitch, header_size, rounds, results + (uint64_t)f * rounds, r, at);
    if (plan.again || at == ~0ull)
        return;
    uint8_t* slot = slots + (uint64_t)f * slot_pitch;
    if (blockIdx.y == 0)
        for (uint32_t i = threadIdx.x; i < header_size; i += blockDim.x)
            slot[at + i] = headers[(uint64_t)r * header_size + i];
    const uint64_t bytes = results[(uint64_t)f * rounds + r].bytes;
    const uint8_t* from = private_streams + ((uint64_t)f * rounds + r) * capacity;
    uint8_t* to = slot + at + header_size;
    const uint64_t quads = bytes / 16;
    for (uint64_t i = (uint64_t)blockIdx.y * blockDim.x + threadIdx.x; i < quads; i += (uint64_t)gridDim.y * blockDim.x)
        reinterpret_cast<UnalignedQuad*>(to)[i] = reinterpret_cast<const UnalignedQuad*>(from)[i];
    if (blockIdx.y == 0)
        for (uint64_t i = quads * 16 + threadIdx.x; i < bytes; i += blockDim.x)
            to[i] = from[i];
}
// One lane per frame, after place_plane_scans: the cursor moves behind the frame's last scan, or the frame is flagged.
__global__ void __launch_bounds__(64) advance_plane_cursors(uint64_t slot_pitch, uint32_t header_size, uint32_t rounds,
                                                            const ScanResult* __restrict__ results, FrameCursor* __restrict__ cursors,
                                                            uint32_t* __restrict__ redo, uint32_t frames)
{
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= frames)
        return;
    uint64_t at;
    const PlanePlan plan = plan_plane_scans(cursors[f], slot_pitch, header_size, rounds, results + (uint64_t)f * rounds, ~0u, at);
    if (!plan.again)
        cursors[f] = plan.end;
    redo[f] = plan.again ? 1u : 0u;
}

__global__ void place_epilogue(uint8_t* __restrict__ slots, uint64_t slot_pitch, FrameCursor* __restrict__ cursors,
                               uint32_t even_size, uint64_t* __restrict__ sizes, uint32_t* __restrict__ errcs,
                               uint32_t frames)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames)
        return;
    FrameCursor c = cursors[f];
    uint8_t* slot = slots + (uint64_t)f * slot_pitch;
    if (c.errc == kOk)
    {
        if (even_size && (c.offset & 1u))
        {
            if (c.offset + 1 > slot_pitch)
                c.errc = kDestinationTooSmall;
            else
                slot[c.offset++] = 0xFF;
        }
    }
    if (c.errc == kOk)
    {
        if (c.offset + 2 > slot_pitch)
            c.errc = kDestinationTooSmall;
        else
        {
            slot[c.offset++] = 0xFF;
            slot[c.offset++] = 0xD9;
        }
    }
    sizes[f] = c.errc == kOk ? c.offset : 0;
    errcs[f] = c.errc;
}


// Where the entropy-coded segment that starts at `from` ends: the first 0xFF that is followed by a byte with its high
// bit set and is not a restart marker (inside a segment the byte after a 0xFF starts with a stuffed zero bit, T.87 A.1 /
// src/jpeg_stream_reader.cpp: read_next_marker_code).  One workgroup per stream, 16 KB per trip, first position by atomicMin.
struct __attribute__((packed)) UnalignedU64
{
    uint64_t v;
};
__device__ inline bool ends_segment(uint32_t byte, uint32_t next)
{
    return byte == 0xFFu && next >= 0x80u && !(next >= 0xD0u && next <= 0xD7u);
}
__global__ void __launch_bounds__(256) find_scan_end(const uint8_t* __restrict__ slots, const MarkerSearch* __restrict__ specs,
                                                     unsigned long long* __restrict__ found)
{
    __shared__ unsigned long long first;
    const MarkerSearch s = specs[blockIdx.x];
    if (threadIdx.x == 0)
        first = kNoMarker;
    __syncthreads();
    constexpr uint64_t kPer = 64; // bytes of a thread per trip: a 0xFF at [at, at + kPer) is this thread's, its follower is read with it
    for (uint64_t base = s.from; base < s.end; base += 256 * kPer)
    {
        const uint64_t at = base + threadIdx.x * kPer;
        if (at + kPer + 8 <= s.end)
        { // eight bytes at a time (gfx950 takes the loads at any address); words without a 0xFF byte are passed over
            uint64_t v[9];
#pragma unroll
            for (int j = 0; j < 9; ++j)
                v[j] = reinterpret_cast<const UnalignedU64*>(slots + at + 8 * j)->v;
            bool mine = false;
#pragma unroll
            for (int j = 0; j < 8 && !mine; ++j)
            {
                const uint64_t inv = ~v[j];
                if (((inv - 0x0101010101010101ull) & ~inv & 0x8080808080808080ull) == 0)
                    continue; // (no byte of v[j] is 0xFF)
                const uint64_t after = (v[j] >> 8) | (v[j + 1] << 56);
                for (int q = 0; q < 8; ++q)
                    if (ends_segment((uint32_t)(v[j] >> (8 * q)) & 0xFFu, (uint32_t)(after >> (8 * q)) & 0xFFu))
                    {
                        atomicMin(&first, static_cast<unsigned long long>(at + 8 * j + q));
                        mine = true;
                        break;
                    }
            }
        }
        else if (at + 1 < s.end)
        { // the last bytes of the stream, one by one
            const uint64_t last = at + kPer < s.end - 1 ? at + kPer : s.end - 1;
            uint32_t byte = slots[at];
            for (uint64_t q = at; q < last; ++q)
            {
                const uint32_t next = slots[q + 1];
                if (ends_segment(byte, next))
                {
                    atomicMin(&first, static_cast<unsigned long long>(q));
                    break;
                }
                byte = next;
            }
        }
        __syncthreads();
        if (first != kNoMarker)
            break; // (every thread reads the same value: nothing writes between the two barriers of a trip)
        __syncthreads();
    }
    if (threadIdx.x == 0)
        found[blockIdx.x] = first;
}

} // namespace jls
