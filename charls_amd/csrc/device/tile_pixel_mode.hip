// tile_pixel_mode.hip -- stages A and B2 of the tile pipeline in "pixel mode": sample-interleaved scans (2 - 4 components,
// HP1 - HP3) of any width, and planar / line-interleaved scans whose lines are wider than a tile.  Included by
// tile_pipeline.hip; stages B1, C1, C2, D and E are the ones of the whole-line modes, the run chain (C3) reads the
// record format of this file (RunRecord2).
//
// What is different from the whole-line modes (analyze_tiles / sort_tiles):
//
//  * The unit of the run rule is the PIXEL: a sample-interleaved scan is in run mode only where the gradients of ALL
//    components quantise to zero, a component whose own gradients are zero while the pixel is not in run mode codes on
//    regular context 0 (chain kZeroContextChain), and every component of an interruption pixel codes on run context 0
//    (src/scan_encoder_impl.hpp:188-207,277-302).  Stage A therefore works with one lane per pixel; the sort stage, whose
//    work is per sample, with one lane per sample of the line (pixel-major, component-minor: the order of the chains).
//  * A tile is either some whole lines (narrow scans) or a SEGMENT of one line (seg_pixels pixels, a multiple of 64).  The
//    lines of a tile and the line above are staged in LDS with a pixel of margin on either side, colour transform applied
//    once, the edge rules of src/scan_codec.hpp:189-195 materialised in the margins -- so the neighbourhood of a sample
//    is four LDS reads without a branch.  The run state at the first pixel of a segment is found by looking BACK along the
//    line (run_state_at): it is set iff the streak of pixels equal to their left neighbour that ends at the segment holds a
//    pixel with all-zero gradients, so the look-back stops at the first pixel that differs from its neighbour -- one round
//    of 64 pixels on anything but flat images.  The length of a run that leaves its segment is counted on from the keys of
//    the following segments (a key inside a run is kNoEvent before AND after the sort stage of that tile overwrites keys
//    with slots).
//  * Run records carry no sample of another tile: the error value (and type) of an interruption sample with a slot of
//    its own is the RECORD of that slot, made by the lane that owns the sample; only the interruption sample of a run of
//    length 0, which shares the run's slot, has it in the run record.
#pragma once

namespace jls {
namespace tile {

// Run-chain records of pixel mode: one word per run event -- its length, the type of its interruption (Ra == Rb or not; always
// 0 in a sample-interleaved scan) and, for a run of length 0, the error value of the interruption sample, which shares the
// run's slot -- and one per interruption sample with a slot of its own: its error value.
struct RunRecord2
{
    static constexpr uint32_t kEol = 1u << 31, kZero = 1u << 30;
    static JLS_DEV uint32_t end_of_line(uint32_t run, uint32_t component) { return run | (component << 28) | kEol; }
    static JLS_DEV uint32_t interrupted(uint32_t run, int which, uint32_t component) { return run | ((uint32_t)which << 27) | (component << 28); }
    static JLS_DEV uint32_t zero_run(int err, int which, uint32_t component)
    {
        return ((uint32_t)err & 0x1FFFFu) | ((uint32_t)which << 27) | (component << 28) | kZero;
    }
    static JLS_DEV uint32_t interruption(int err) { return (uint32_t)err & 0x1FFFFu; }
    static JLS_DEV bool is_end_of_line(uint32_t v) { return (v >> 31) != 0; }
    static JLS_DEV bool is_zero_run(uint32_t v) { return ((v >> 30) & 1u) != 0; }
    static JLS_DEV uint32_t run(uint32_t v) { return is_zero_run(v) ? 0u : v & 0x7FFFFFFu; }
    static JLS_DEV uint32_t component(uint32_t v) { return (v >> 28) & 3u; }
    static JLS_DEV int err(uint32_t v) { return (int)(v << 15) >> 15; }
    static JLS_DEV int which(uint32_t v) { return (int)((v >> 27) & 1u); }
};
// The error value in the record of an interruption sample with a slot of its own (a slot holds its low bits).
template <typename S>
JLS_DEV int interruption_err(Slot<S> v)
{
    return sizeof(Slot<S>) == 2 ? (int)(int16_t)v : RunRecord2::err((uint32_t)v);
}

// A tile of pixel mode.
struct PixelTile
{
    uint32_t first_line, tile_lines; // coded lines
    uint32_t px0, pixels;            // first pixel of the tile's piece of its lines, pixels of it
    uint32_t nc, step;               // samples per pixel; distance to the line above (ILV_LINE: the components)
    uint32_t line_samples;           // width * nc
    uint32_t slots;                  // pixels + 2: a staged row holds pixel px0 - 1 .. px0 + pixels
    uint32_t pieces;                 // pieces a row is cut into for the wavefronts of the workgroup
};
JLS_DEV PixelTile pixel_tile(const ScanDesc& d, const Work& w, uint32_t tile)
{
    PixelTile g;
    const uint32_t lines = scan_lines(d);
    g.nc = samples_per_pixel(d);
    g.step = pipe::line_step(d);
    g.line_samples = d.width * g.nc;
    if (w.segs_per_line == 1)
    {
        g.first_line = tile * w.lines_per_tile;
        g.tile_lines = lines - g.first_line < w.lines_per_tile ? lines - g.first_line : w.lines_per_tile;
        g.px0 = 0;
        g.pixels = d.width;
    }
    else
    {
        g.first_line = tile / w.segs_per_line;
        g.tile_lines = 1;
        g.px0 = (tile % w.segs_per_line) * w.seg_pixels;
        g.pixels = d.width - g.px0 < w.seg_pixels ? d.width - g.px0 : w.seg_pixels;
    }
    g.slots = g.pixels + 2;
    g.pieces = w.lines_per_tile >= kWaves ? 1u : kWaves / w.lines_per_tile;
    return g;
}

// Pixel x of coded line `line` as the codec sees it: masked to the sample precision, colour transform applied; nc samples.
template <typename S>
JLS_DEV void load_coded_pixel(const ScanDesc& d, uint32_t line, uint32_t x, int mask, int out[4])
{
    if (d.interleave_mode == 2)
    {
        pipe::load_pixel<S>(d, line, x, mask, out);
        return;
    }
    if (d.interleave_mode == 1)
    {
        int px[4];
        const uint32_t comps = (uint32_t)d.components, c = line % comps;
        pipe::load_pixel<S>(d, line / comps, x, mask, px);
        out[0] = c == 0 ? px[0] : c == 1 ? px[1] : c == 2 ? px[2] : px[3];
        return;
    }
    const S* row = reinterpret_cast<const S*>(d.pixels + (size_t)line * d.pixel_stride);
    out[0] = (int)row[x] & mask;
}

// Rows of the staging area: row r = coded line first_line - step + r -- `step` lines above the tile (one; the components of a
// line-interleaved scan, whose line above is the line of the same component of the pixel row above), then the tile's lines;
// slot j of a row = pixel px0 - 1 + j.  Margins: left of pixel 0 sits pixel 0 of the line above (cur[0] = prev[1], src/scan_codec.hpp:189-195),
// right of the last pixel of a line that pixel again (prev[w + 1] = prev[w]); lines above the scan are zeros.
//
// Pixel by pixel (line-interleaved scans: the staged line holds ONE component of the pixels in memory).
template <typename S>
JLS_DEV void stage_pixel_rows_generic(const ScanDesc& d, const PixelTile& g, S* rows, int mask)
{
    const uint32_t total = (g.tile_lines + g.step) * g.slots;
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x)
    {
        const uint32_t row = i / g.slots, slot = i - row * g.slots;
        int64_t line = (int64_t)g.first_line - (int64_t)g.step + row;
        int64_t p = (int64_t)g.px0 + slot - 1;
        if (p < 0)
        { // left margin of the line's first pixel
            p = 0;
            line -= g.step;
        }
        else if (p >= (int64_t)d.width)
            p = (int64_t)d.width - 1;
        int px[4] = {0, 0, 0, 0};
        if (line >= 0)
            load_coded_pixel<S>(d, (uint32_t)line, (uint32_t)p, mask, px);
        S* to = rows + (size_t)i * g.nc;
        for (uint32_t c = 0; c < g.nc; ++c)
            to[c] = (S)(c == 0 ? px[0] : c == 1 ? px[1] : c == 2 ? px[2] : px[3]);
    }
}

struct __attribute__((packed)) UnalignedU32
{
    uint32_t v;
};

// Line-interleaved scans: the staged rows are COMPONENTS of pixel rows (coded line L = component L % components of pixel row
// L / components), so every source pixel under the tile is loaded and colour-transformed ONCE and its components go to the
// staged rows they belong to (pixel by pixel -- stage_pixel_rows_generic -- every staged sample loaded a whole pixel and
// transformed it: 2.5 pixels per coded sample for a tile of two lines and the three lines above them).
template <typename S>
JLS_DEV void stage_line_interleaved_rows(const ScanDesc& d, const PixelTile& g, S* rows, int mask)
{
    const uint32_t comps = g.step;                            // (the distance to the line above IS the number of components)
    const int64_t first = (int64_t)g.first_line - (int64_t)g.step; // coded line of staged row 0; lines above the scan are zeros
    const uint32_t R = g.tile_lines + g.step;
    const uint32_t a = g.px0 > 0 ? g.px0 - 1 : 0;             // first and last pixel of a row that are real neighbours in memory
    const uint32_t b = g.px0 + g.pixels < d.width ? g.px0 + g.pixels : d.width - 1;
    const uint32_t per = b - a + 1;
    const uint32_t zero_rows = first < 0 ? (uint32_t)(-first) : 0u; // (a multiple of the components or the whole margin above line 0)
    for (uint32_t i = threadIdx.x; i < zero_rows * g.slots; i += blockDim.x)
        rows[i] = 0;
    const uint32_t y0 = first > 0 ? (uint32_t)first / comps : 0u;
    const uint32_t y1 = (uint32_t)(first + (int64_t)R - 1) / comps;
    for (uint32_t i = threadIdx.x; i < (y1 - y0 + 1) * per; i += blockDim.x)
    {
        const uint32_t y = y0 + i / per, x = a + i % per;
        int px[4];
        pipe::load_pixel<S>(d, y, x, mask, px);
        for (uint32_t c = 0; c < comps; ++c)
        {
            const int64_t r = (int64_t)y * comps + c - first;
            if (r >= 0 && r < (int64_t)R)
                rows[(size_t)r * g.slots + (x + 1 - g.px0)] = (S)(c == 0 ? px[0] : c == 1 ? px[1] : c == 2 ? px[2] : px[3]);
        }
    }
    if (threadIdx.x < 2 * R)
    { // the margins at the edges of the scan: left of pixel 0 sits pixel 0 of the line above, right of the last pixel that pixel again
        const uint32_t row = threadIdx.x / 2;
        const bool right = (threadIdx.x & 1u) != 0;
        const int64_t line = first + row;
        if (right ? g.px0 + g.pixels == d.width : g.px0 == 0)
        {
            const int64_t from_line = right ? line : line - (int64_t)g.step;
            int px[4] = {0, 0, 0, 0};
            if (from_line >= 0)
                load_coded_pixel<S>(d, (uint32_t)from_line, right ? d.width - 1 : 0u, mask, px);
            rows[(size_t)row * g.slots + (right ? g.slots - 1 : 0u)] = (S)px[0];
        }
    }
}

// Planar and sample-interleaved scans: the pixels of a staged row are the bytes of the source row, so they come as whole
// words (gfx950 takes a 4-byte global load at any address; the LDS side is kept aligned: up to three bytes at either end
// of a row go one by one), masked to the sample precision on the way; a colour transform then runs over the staged pixels
// in place, and the margins that are not neighbours in memory (the edges of the scan) are filled pixel by pixel.  Every
// sample is loaded once and transformed once, where the stages read it five times.
template <typename S>
JLS_DEV void stage_pixel_rows(const ScanDesc& d, const PixelTile& g, S* rows, int mask)
{
    if (d.interleave_mode == 1)
    {
        stage_line_interleaved_rows<S>(d, g, rows, mask);
        return;
    }
    const uint32_t pb = g.nc * (uint32_t)sizeof(S), row_bytes = g.slots * pb;
    uint8_t* lds = reinterpret_cast<uint8_t*>(rows); // (16-byte aligned)
    const uint32_t a = g.px0 > 0 ? g.px0 - 1 : 0; // first and last pixel that are copied
    const uint32_t b = g.px0 + g.pixels < d.width ? g.px0 + g.pixels : d.width - 1;
    const uint32_t first_slot = a + 1 - g.px0;
    const uint32_t n = (b - a + 1) * pb; // bytes of a row that are copied
    const uint32_t mask_word = sizeof(S) == 1 ? (uint32_t)mask * 0x01010101u : (uint32_t)mask * 0x00010001u;
    const uint32_t R = g.tile_lines + 1;
    const uint32_t max_words = n / 4 + 1;
    auto row_at = [&](uint32_t row, uint32_t& lead, uint32_t& words) -> uint32_t { // offset of the copied bytes in LDS
        const uint32_t at = row * row_bytes + first_slot * pb;
        lead = (4u - (at & 3u)) & 3u;
        lead = lead < n ? lead : n;
        words = (n - lead) / 4;
        return at;
    };
    for (uint32_t i = threadIdx.x; i < R * max_words; i += blockDim.x)
    {
        const uint32_t row = i / max_words, wi = i - row * max_words;
        uint32_t lead, words;
        const uint32_t at = row_at(row, lead, words);
        if (wi < words)
        {
            uint32_t v = 0;
            if (g.first_line + row >= 1)
            {
                const uint8_t* from = d.pixels + (size_t)(g.first_line + row - 1) * d.pixel_stride + (size_t)a * pb + lead + 4 * wi;
                v = reinterpret_cast<const UnalignedU32*>(from)->v & mask_word;
            }
            *reinterpret_cast<uint32_t*>(lds + at + lead + 4 * wi) = v;
        }
    }
    for (uint32_t i = threadIdx.x; i < R * 8; i += blockDim.x)
    { // the bytes before the first and behind the last whole word of a row
        const uint32_t row = i / 8, j = i % 8;
        uint32_t lead, words;
        const uint32_t at = row_at(row, lead, words);
        const uint32_t tail = n - lead - 4 * words;
        const bool head = j < 4;
        const uint32_t o = head ? j : lead + 4 * words + (j - 4);
        if (head ? j < lead : (j - 4) < tail)
        {
            uint32_t v = 0;
            if (g.first_line + row >= 1)
            {
                const uint32_t byte_mask = sizeof(S) == 1 ? (uint32_t)mask : ((o & 1u) ? (uint32_t)mask >> 8 : (uint32_t)mask & 0xFFu);
                v = (uint32_t)(d.pixels + (size_t)(g.first_line + row - 1) * d.pixel_stride + (size_t)a * pb)[o] & byte_mask;
            }
            lds[at + o] = (uint8_t)v;
        }
    }
    if (d.color_transformation != 0)
    { // (three components of 8 or 16 bits, src/color_transform.hpp:26-117; every pixel belongs to one thread)
        __syncthreads();
        const uint32_t per_row = b - a + 1;
        for (uint32_t i = threadIdx.x; i < R * per_row; i += blockDim.x)
        {
            const uint32_t row = i / per_row, k = i - row * per_row;
            if (g.first_line + row < 1)
                continue;
            S* px = reinterpret_cast<S*>(lds + row * row_bytes + (first_slot + k) * pb);
            unsigned t[3];
            hp_forward(d.color_transformation, sizeof(S) == 2, (int)px[0], (int)px[1], (int)px[2], t);
            px[0] = (S)t[0];
            px[1] = (S)t[1];
            px[2] = (S)t[2];
        }
    }
    if (threadIdx.x < 2 * R)
    { // the margins at the edges of the scan
        const uint32_t row = threadIdx.x / 2;
        const bool right = (threadIdx.x & 1u) != 0;
        const int64_t line = (int64_t)g.first_line + row - 1;
        if (right ? g.px0 + g.pixels == d.width : g.px0 == 0)
        {
            const int64_t from_line = right ? line : line - 1;
            int px[4] = {0, 0, 0, 0};
            if (from_line >= 0)
                load_coded_pixel<S>(d, (uint32_t)from_line, right ? d.width - 1 : 0u, mask, px);
            S* to = reinterpret_cast<S*>(lds + row * row_bytes + (right ? g.slots - 1 : 0u) * pb);
            for (uint32_t c = 0; c < g.nc; ++c)
                to[c] = (S)(c == 0 ? px[0] : c == 1 ? px[1] : c == 2 ? px[2] : px[3]);
        }
    }
}

// eq (the pixel equals its left neighbour in every component) and q0 (the gradients of every component quantise to zero)
// of pixel x of a line, from memory: for pixels that lie before a segment tile (run_state_at).
template <typename S>
JLS_DEV void pixel_flags(const ScanDesc& d, const Traits& t, uint32_t line, uint32_t x, int mask, bool& eq, bool& q0)
{
    const uint32_t step = pipe::line_step(d), nc = samples_per_pixel(d);
    int v[4], ra[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0}, rc[4] = {0, 0, 0, 0}, rd[4] = {0, 0, 0, 0};
    load_coded_pixel<S>(d, line, x, mask, v);
    if (x > 0)
        load_coded_pixel<S>(d, line, x - 1, mask, ra);
    else if (line >= step)
        load_coded_pixel<S>(d, line - step, 0, mask, ra);
    if (line >= step)
    {
        load_coded_pixel<S>(d, line - step, x, mask, rb);
        if (x > 0)
            load_coded_pixel<S>(d, line - step, x - 1, mask, rc);
        else if (line >= 2 * step)
            load_coded_pixel<S>(d, line - 2 * step, 0, mask, rc);
        load_coded_pixel<S>(d, line - step, x + 1 < d.width ? x + 1 : d.width - 1, mask, rd);
    }
    eq = true;
    q0 = true;
    for (uint32_t c = 0; c < nc; ++c)
    {
        const int vv = c == 0 ? v[0] : c == 1 ? v[1] : c == 2 ? v[2] : v[3];
        const int a = c == 0 ? ra[0] : c == 1 ? ra[1] : c == 2 ? ra[2] : ra[3];
        const int b = c == 0 ? rb[0] : c == 1 ? rb[1] : c == 2 ? rb[2] : rb[3];
        const int cc = c == 0 ? rc[0] : c == 1 ? rc[1] : c == 2 ? rc[2] : rc[3];
        const int dd = c == 0 ? rd[0] : c == 1 ? rd[1] : c == 2 ? rd[2] : rd[3];
        eq = eq && vv == a;
        q0 = q0 && context_id(t, a, b, cc, dd) == 0;
    }
}

// Run-mode state before pixel px0 of a line (all lanes of ONE wavefront call; the result is the same in all of them).
// s(x + 1) = eq(x) & (s(x) | q0(x)), s(0) = 0: s(px0) is set iff some x0 < px0 has q0(x0) and eq(x0 .. px0 - 1).  Looks back
// 64 pixels a round, lane i at pixel end - 1 - i, and stops at the first pixel that differs from its left neighbour.
template <typename S>
JLS_DEV bool run_state_at(const ScanDesc& d, const Traits& t, uint32_t line, uint32_t px0, int mask)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t end = px0;
    while (end > 0)
    {
        const uint32_t n = end < 64 ? end : 64;
        bool eq = false, q0 = false;
        if (lane < n)
            pixel_flags<S>(d, t, line, end - 1 - lane, mask, eq, q0);
        const unsigned long long m_eq = __ballot(eq), m_q0 = __ballot(q0);
        const uint32_t streak = m_eq == ~0ull ? 64u : (uint32_t)__ffsll(~m_eq) - 1u;
        const unsigned long long in_streak = streak >= 64 ? ~0ull : ((1ull << streak) - 1ull);
        if ((m_q0 & in_streak) != 0)
            return true;
        if (streak < n)
            return false;
        end -= n;
    }
    return false;
}

// LDS carve-up of the two kernels (byte offsets, every region 16-byte aligned).
struct PixelLds
{
    uint32_t rows, keys, masks, table;
};
JLS_HOST_DEV_EARLY uint32_t pixel_rows_bytes(uint32_t lines_per_tile, uint32_t step, uint32_t max_pixels, uint32_t nc, uint32_t sample_bytes)
{
    return ((lines_per_tile + step) * (max_pixels + 2) * nc * sample_bytes + 15u) & ~15u;
}
JLS_HOST_DEV_EARLY PixelLds pixel_lds(uint32_t lines_per_tile, uint32_t step, uint32_t max_pixels, uint32_t nc, uint32_t sample_bytes,
                                      uint32_t tile_capacity, bool with_keys, uint32_t chunks)
{
    PixelLds l;
    l.rows = 0;
    l.keys = pixel_rows_bytes(lines_per_tile, step, max_pixels, nc, sample_bytes);
    l.masks = l.keys + (with_keys ? ((tile_capacity * 2u + 15u) & ~15u) : 0u);
    l.table = l.masks + ((lines_per_tile * chunks * 16u + 15u) & ~15u);
    return l;
}
// pixels of a tile row at most / 64-pixel and 64-sample chunks of it
JLS_HOST_DEV_EARLY uint32_t tile_row_pixels(uint32_t width, uint32_t segs_per_line, uint32_t seg_pixels)
{
    return segs_per_line == 1 ? width : seg_pixels;
}

// ---------------------------------------------------------------------------------------------------------------
// A (pixel mode): grid (8 * ceil(tiles / 8), scans) x 512.
// LDS: rows | keys[tile] u16 | per row and pixel chunk: eq, q0 masks | hist[kChains + 1] | gradient table (512 B) | run state in
template <typename S>
__global__ void __launch_bounds__(kThreads) analyze_pixel_tiles(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    JLS_DYNAMIC_LDS(smem);
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const Traits t = make_traits(d);
    const uint32_t tile = tile_of_block(blockIdx.x, scan_tiles(d, w));
    if (tile >= scan_tiles(d, w))
        return;
    const PixelTile g = pixel_tile(d, w, tile);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nc = g.nc, P = g.pixels;
    const uint32_t max_pixels = tile_row_pixels(d.width, w.segs_per_line, w.seg_pixels);
    const uint32_t max_chunks = (max_pixels + 63) / 64, chunks = (P + 63) / 64;
    const uint32_t chunks_per_piece = (chunks + g.pieces - 1) / g.pieces;
    const uint32_t segments = g.tile_lines * g.pieces;
    const PixelLds lds = pixel_lds(w.lines_per_tile, g.step, max_pixels, nc, (uint32_t)sizeof(S), w.tile_capacity, true, max_chunks);
    S* s_rows = reinterpret_cast<S*>(smem + lds.rows);
    uint16_t* s_key = reinterpret_cast<uint16_t*>(smem + lds.keys); // [row][pixel][component]
    uint64_t* s_eq = reinterpret_cast<uint64_t*>(smem + lds.masks); // [row][chunk]
    uint64_t* s_q0 = s_eq + (size_t)w.lines_per_tile * max_chunks;
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + lds.table);
    unsigned char* s_grad = reinterpret_cast<unsigned char*>(s_hist + kChains + 1);
    uint32_t* s_in = reinterpret_cast<uint32_t*>(s_grad + pipe::kGradientTable);
    const int mask = (1 << d.bits_per_sample) - 1;

    stage_pixel_rows<S>(d, g, s_rows, mask);
    for (uint32_t c = threadIdx.x; c < (uint32_t)kChains; c += kThreads)
        s_hist[c] = 0;
    if (sizeof(S) == 1)
        for (uint32_t q = threadIdx.x; q < 511; q += kThreads)
            s_grad[q] = (unsigned char)(quantize(t, (int)q - 255) + 4);
    if (wave == 0)
    { // run-mode state at the tile's first pixel (segments of lines only)
        const bool in_run = g.px0 != 0 && run_state_at<S>(d, t, g.first_line, g.px0, mask);
        if (lane == 0)
            *s_in = in_run ? 1u : 0u;
    }
    __syncthreads();

    // ---- pass 1: every component as if coded in regular mode; equality / zero-context masks per 64-pixel chunk
    for (uint32_t sgm = wave; sgm < segments; sgm += kWaves)
    {
        const uint32_t r = sgm / g.pieces, piece = sgm % g.pieces;
        const uint32_t k0 = piece * chunks_per_piece;
        const uint32_t k1 = k0 + chunks_per_piece < chunks ? k0 + chunks_per_piece : chunks;
        for (uint32_t k = k0; k < k1; ++k)
        {
            const uint32_t xl = k * 64 + lane; // pixel of the tile row
            bool eq = false, q0 = false;
            if (xl < P)
            {
                const S* cur = s_rows + ((size_t)(r + g.step) * g.slots + xl + 1) * nc;
                const S* above = cur - (size_t)g.step * g.slots * nc;
                uint16_t* keys = s_key + ((size_t)r * P + xl) * nc;
                eq = true;
                q0 = true;
                for (uint32_t c = 0; c < nc; ++c)
                {
                    const int v = (int)cur[c], ra = (int)cur[(int)c - (int)nc];
                    const int rb = (int)above[c], rc = (int)above[(int)c - (int)nc], rd = (int)above[c + nc];
                    const int qs = sizeof(S) == 1
                                       ? ((int)s_grad[rd - rb + 255] * 9 + (int)s_grad[rb - rc + 255]) * 9 + (int)s_grad[rc - ra + 255] - 364
                                       : context_id(t, ra, rb, rc, rd);
                    const int sg = qs >> 31;
                    const int ctx = (qs ^ sg) - sg;
                    keys[c] = (uint16_t)((ctx == 0 ? kZeroContextChain : ctx) | ((sg & 1) << 9));
                    eq = eq && v == ra;
                    q0 = q0 && qs == 0;
                }
            }
            const unsigned long long m_eq = __ballot(eq);
            const unsigned long long m_q0 = __ballot(q0);
            if (lane == 0)
            {
                s_eq[r * max_chunks + k] = m_eq;
                s_q0[r * max_chunks + k] = m_q0;
            }
        }
    }
    __syncthreads();
    // ---- pass 2: run-mode state before every pixel (the carry chain of analyze_tiles; a tile that is a segment of a line
    // starts it with the state run_state_at found)
    const unsigned long long carry_in = *s_in;
    for (uint32_t sgm = wave; sgm < segments; sgm += kWaves)
    {
        const uint32_t r = sgm / g.pieces, piece = sgm % g.pieces;
        const uint32_t k0 = piece * chunks_per_piece;
        const uint32_t k1 = k0 + chunks_per_piece < chunks ? k0 + chunks_per_piece : chunks;
        unsigned long long carry = carry_in;
        for (uint32_t k = 0; k < k1; ++k)
        {
            const unsigned long long a = s_eq[r * max_chunks + k];
            const unsigned long long b = a & s_q0[r * max_chunks + k];
            const unsigned long long sum = a + b + carry;
            const unsigned long long st = sum ^ a ^ b; // bit i: in-run state before pixel i
            carry = (((a & b) | ((a | b) & st)) >> 63) & 1ull;
            const uint32_t xl = k * 64 + lane;
            if (k >= k0 && xl < P)
            {
                const bool s = (st >> lane) & 1ull;
                const bool q0 = (s_q0[r * max_chunks + k] >> lane) & 1ull;
                const bool eq = (a >> lane) & 1ull;
                uint16_t* keys = s_key + ((size_t)r * P + xl) * nc;
                if (!(s || q0))
                { // regular pixel
                    for (uint32_t c = 0; c < nc; ++c)
                        atomicAdd(&s_hist[keys[c] & 0x1FF], 1u);
                }
                else if (s && eq)
                { // inside a run
                    for (uint32_t c = 0; c < nc; ++c)
                        keys[c] = kNoEvent;
                }
                else if (s)
                { // the pixel that ends a run started earlier: coded by the run lane, every component owns a slot
                    for (uint32_t c = 0; c < nc; ++c)
                        keys[c] = (uint16_t)kInterruptChain;
                    atomicAdd(&s_hist[kInterruptChain], nc);
                }
                else
                { // a run starts here (possibly of length 0: then this pixel also ends it and its first component shares the run's slot)
                    keys[0] = 0;
                    atomicAdd(&s_hist[0], 1u);
                    for (uint32_t c = 1; c < nc; ++c)
                        keys[c] = eq ? kNoEvent : (uint16_t)kInterruptChain;
                    if (!eq && nc > 1)
                        atomicAdd(&s_hist[kInterruptChain], nc - 1);
                }
            }
        }
    }
    __syncthreads();
    // ---- keys out, coalesced (a row of the tile is contiguous in the key array)
    const uint32_t row_samples = P * nc;
    for (uint32_t r = 0; r < g.tile_lines; ++r)
    {
        uint16_t* to = w.keyinv + (size_t)(g.first_line + r) * g.line_samples + (size_t)g.px0 * nc;
        for (uint32_t q = threadIdx.x; q < row_samples; q += kThreads)
            to[q] = s_key[(size_t)r * row_samples + q];
    }
    for (uint32_t c = threadIdx.x; c < (uint32_t)kChains; c += kThreads)
        w.seg[(size_t)tile * kChains + c] = s_hist[c];
}

// ---------------------------------------------------------------------------------------------------------------
// B2 (pixel mode): grid (8 * ceil(tiles / 8), scans) x 512; one lane per SAMPLE of a tile row (pixel-major).
// LDS: rows | noev[row][chunk] u64, lead[row][chunk + 1] u32 | the tables of sort_tiles
template <typename S>
__global__ void __launch_bounds__(kThreads) sort_pixel_tiles(const ScanDesc* __restrict__ descs, const Work* __restrict__ works)
{
    JLS_DYNAMIC_LDS(smem);
    const ScanDesc d = descs[blockIdx.y];
    const Work w = works[blockIdx.y];
    const Traits t = make_traits(d);
    const uint32_t tile = tile_of_block(blockIdx.x, scan_tiles(d, w));
    if (tile >= scan_tiles(d, w))
        return;
    const PixelTile g = pixel_tile(d, w, tile);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nc = g.nc;
    const uint32_t Ps = g.pixels * nc; // samples of a tile row
    const uint32_t max_pixels = tile_row_pixels(d.width, w.segs_per_line, w.seg_pixels);
    const uint32_t max_chunks = (max_pixels * nc + 63) / 64, chunks = (Ps + 63) / 64;
    const uint32_t chunks_per_piece = (chunks + g.pieces - 1) / g.pieces;
    const uint32_t segments = g.tile_lines * g.pieces;
    const PixelLds lds = pixel_lds(w.lines_per_tile, g.step, max_pixels, nc, (uint32_t)sizeof(S), w.tile_capacity, false, max_chunks);
    S* s_rows = reinterpret_cast<S*>(smem + lds.rows);
    uint64_t* s_noev = reinterpret_cast<uint64_t*>(smem + lds.masks);                               // [row][chunk]
    uint32_t* s_lead = reinterpret_cast<uint32_t*>(s_noev + (size_t)w.lines_per_tile * max_chunks); // [row][chunk + 1]
    uint32_t* s_segoff = reinterpret_cast<uint32_t*>(smem + lds.table);                             // [kSegments][kChains]
    uint32_t* s_tileoff = s_segoff + sort_segments(w.lines_per_tile) * kChains;
    uint32_t* s_count = s_tileoff + kChains + 1;
    uint32_t* s_global = s_count + kChains + 1;
    uint32_t* s_tmp = s_global + kChains + 1;
    uint32_t* s_same = s_tmp + 16;
    uint32_t* s_rowbase = s_same + kWaves * (kChains + 1);
    uint16_t* s_rowchain = reinterpret_cast<uint16_t*>(s_rowbase + kChains + 1);
    Slot<S>* s_stage = reinterpret_cast<Slot<S>*>(s_rowbase + kChains + 1 + kRowChainWords);
    const int mask = (1 << d.bits_per_sample) - 1;
    auto key_row = [&](uint32_t r) -> uint16_t* { return w.keyinv + (size_t)(g.first_line + r) * g.line_samples + (size_t)g.px0 * nc; };

    stage_pixel_rows<S>(d, g, s_rows, mask);
    for (uint32_t i = threadIdx.x; i < segments * (uint32_t)kChains; i += kThreads)
        s_segoff[i] = 0;
    for (uint32_t i = threadIdx.x; i < kWaves * ((uint32_t)kChains + 1); i += kThreads)
        s_same[i] = 0;
    __syncthreads();
    // ---- P1: events per (segment, chain), samples inside runs per chunk
    for (uint32_t sgm = wave; sgm < segments; sgm += kWaves)
    {
        const uint32_t r = sgm / g.pieces, piece = sgm % g.pieces;
        const uint32_t k0 = piece * chunks_per_piece;
        const uint32_t k1 = k0 + chunks_per_piece < chunks ? k0 + chunks_per_piece : chunks;
        const uint16_t* keys = key_row(r);
        for (uint32_t kb = k0; kb < k1; kb += 16)
        {
            uint16_t held[16];
#pragma unroll
            for (int j = 0; j < 16; ++j)
            {
                const uint32_t q = (kb + j) * 64 + lane;
                held[j] = kb + j < k1 && q < Ps ? keys[q] : kNoEvent;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
            {
                const uint32_t k = kb + j;
                if (k < k1) // (uniform)
                {
                    const uint32_t q = k * 64 + lane;
                    const uint16_t key = held[j];
                    if (q < Ps && key != kNoEvent)
                        atomicAdd(&s_segoff[sgm * kChains + (key & 0x1FF)], (key & 0x1FF) == 0 ? run_slots_of<S>() : 1u); // (slots)
                    const unsigned long long m = __ballot(q < Ps && key == kNoEvent);
                    if (lane == 0)
                        s_noev[r * max_chunks + k] = m;
                }
            }
        }
    }
    __syncthreads();
    { // ---- offsets (as sort_tiles; the interruption chain has records here)
        uint32_t n[2] = {0, 0};
        for (int half = 0; half < 2; ++half)
        {
            const uint32_t c = threadIdx.x + (uint32_t)half * kThreads;
            if (c < (uint32_t)kChains)
                for (uint32_t sgm = 0; sgm < segments; ++sgm)
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
                for (uint32_t sgm = 0; sgm < segments; ++sgm)
                {
                    const uint32_t m = s_segoff[sgm * kChains + c];
                    s_segoff[sgm * kChains + c] = running;
                    running += m;
                }
            }
        }
        uint32_t rows[2], row_base[2];
        for (int half = 0; half < 2; ++half)
        {
            const uint32_t c = threadIdx.x + (uint32_t)half * kThreads;
            rows[half] = row_base[half] = c < (uint32_t)kChains ? (n[half] + 63) / 64 : 0u;
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
        { // samples inside runs from the first sample of every chunk on
            const uint32_t r = kThreads - 1 - threadIdx.x;
            uint32_t lead = 0;
            s_lead[r * (max_chunks + 1) + chunks] = 0;
            for (uint32_t k = chunks; k-- > 0;)
            {
                const unsigned long long m = s_noev[r * max_chunks + k];
                lead = m == ~0ull ? 64 + lead : (uint32_t)__ffsll(~m) - 1;
                s_lead[r * (max_chunks + 1) + k] = lead;
            }
        }
    }
    __syncthreads();
    // ---- P2: ranks (the LDS mask table of sort_tiles), records
    for (uint32_t sgm = wave; sgm < segments; sgm += kWaves)
    {
        const uint32_t r = sgm / g.pieces, piece = sgm % g.pieces;
        const uint32_t y = g.first_line + r;
        const uint32_t k0 = piece * chunks_per_piece;
        const uint32_t k1 = k0 + chunks_per_piece < chunks ? k0 + chunks_per_piece : chunks;
        uint32_t* segoff = s_segoff + sgm * kChains;
        uint32_t* same_of = s_same + (uint32_t)wave * (kChains + 1);
        uint16_t* inv_row = key_row(r);
        auto key_at = [&](uint32_t k) -> uint16_t {
            const uint32_t q = k * 64 + lane;
            return k < k1 && q < Ps ? inv_row[q] : kNoEvent;
        };
        uint16_t key_1 = key_at(k0), key_2 = key_at(k0 + 1);
        const S* cur_row = s_rows + ((size_t)(r + g.step) * g.slots + 1) * nc; // sample 0 of the tile row; its left neighbour nc before
        const S* above_row = cur_row - (size_t)g.step * g.slots * nc;
        const uint32_t lane_bit = 1u << (lane & 31), below = lane_bit - 1u;
        const bool upper = lane >= 32;
        for (uint32_t k = k0; k < k1; ++k)
        {
            const uint32_t q = k * 64 + lane;
            const bool inside = q < Ps;
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
            // the neighbourhood of the sample, for every lane (no divergence; lanes without an event discard it)
            uint32_t record = 0;
            int v = 0, ra = 0, rb = 0;
            if (inside)
            {
                v = (int)cur_row[q];
                ra = (int)cur_row[(int)q - (int)nc];
                rb = (int)above_row[q];
                const int rc = (int)above_row[(int)q - (int)nc];
                record = make_record<S>(v, med3(ra + rb - rc, ra, rb), (key >> 9) & 1, t.maxval);
            }
            if (__any(has && (chain == 0 || chain == (uint32_t)kInterruptChain)))
            { // (rare) run starts and interruption samples
                const int which = nc == 1 && ra == rb ? 1 : 0; // (several components: always run context 0, src/scan_encoder_core.hpp:127-138)
                const int err = which ? error_value(t, v - ra) : error_value(t, (v - rb) * ((rb - ra) < 0 ? -1 : 1));
                if (has && chain == (uint32_t)kInterruptChain)
                    record = RunRecord2::interruption(err);
                if (has && chain == 0)
                {
                    // samples inside the run that follow in this tile row ...
                    const unsigned long long after = lane == 63 ? 0ull : s_noev[r * max_chunks + k] >> (lane + 1);
                    const uint32_t rest = 63u - (uint32_t)lane;
                    uint32_t n = (uint32_t)__ffsll(~after) - 1;
                    if (n >= rest)
                        n = rest + s_lead[r * (max_chunks + 1) + k + 1];
                    // ... and, where the run leaves the tile, in the segments of the line that follow (kNoEvent before and
                    // after the sort stage of those tiles)
                    if (q + 1 + n >= Ps && g.px0 + g.pixels < d.width)
                    {
                        const uint16_t* line_keys = w.keyinv + (size_t)y * g.line_samples;
                        uint32_t at = (g.px0 + g.pixels) * nc;
                        while (at < g.line_samples && line_keys[at] == kNoEvent)
                        {
                            ++at;
                            ++n;
                        }
                    }
                    const bool eq = nc == 1 ? v == ra : n > 0;
                    const uint32_t run = eq ? (n + 1) / nc : 0u;
                    const uint32_t x = g.px0 + q / nc;
                    const uint32_t component = d.interleave_mode == 1 ? y % g.step : 0u;
                    if (x + run >= d.width)
                        record = RunRecord2::end_of_line(run, component);
                    else if (run == 0)
                        record = RunRecord2::zero_run(err, which, component);
                    else
                    { // the type of the interruption: Ra is the run's value, Rb the sample above the interruption sample -- which
                      // may lie in another tile
                        int above[4] = {0, 0, 0, 0};
                        if (nc == 1 && y >= g.step)
                            load_coded_pixel<S>(d, y - g.step, x + run, mask, above);
                        record = RunRecord2::interrupted(run, nc == 1 && v == above[0] ? 1 : 0, component);
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
                inv_row[q] = has ? (uint16_t)slot : kNoLocalSlot;
        }
    }
    __syncthreads();
    // ---- P3: pieces out
    const uint32_t total_rows = s_rowbase[kChains];
    for (uint32_t q0 = (uint32_t)wave * 4; q0 < total_rows; q0 += kWaves * 4)
    {
        uint32_t to[4];
        Slot<S> held[4];
        bool live[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const uint32_t q = q0 + (uint32_t)j;
            const uint32_t c = q < total_rows ? s_rowchain[q] : 0u;
            const uint32_t i = (q - s_rowbase[c]) * 64 + (uint32_t)lane;
            live[j] = q < total_rows && i < s_count[c];
            to[j] = s_global[c] + i;
            held[j] = live[j] ? s_stage[s_tileoff[c] + i] : (Slot<S>)0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (live[j])
                rec_slots<S>(w)[to[j]] = held[j];
    }
}

inline size_t analyze_pixel_lds_bytes(uint32_t lines_per_tile, uint32_t step, uint32_t max_pixels, uint32_t nc, uint32_t sample_bytes, uint32_t tile_capacity)
{
    const PixelLds l = pixel_lds(lines_per_tile, step, max_pixels, nc, sample_bytes, tile_capacity, true, (max_pixels + 63) / 64);
    return (size_t)l.table + ((size_t)kChains + 1) * 4 + pipe::kGradientTable + 16;
}
inline size_t sort_pixel_lds_bytes(uint32_t lines_per_tile, uint32_t step, uint32_t max_pixels, uint32_t nc, uint32_t sample_bytes, uint32_t tile_capacity)
{
    const PixelLds l = pixel_lds(lines_per_tile, step, max_pixels, nc, sample_bytes, tile_capacity, false, (max_pixels * nc + 63) / 64);
    return (size_t)l.table + (size_t)sort_segments(lines_per_tile) * kChains * 4 + 4 * ((size_t)kChains + 1) * 4 + 16 * 4 +
           (size_t)kWaves * (kChains + 1) * 4 + kRowChainWords * 4 + stage_bytes(tile_capacity, sample_bytes);
}

} // namespace tile
} // namespace jls
