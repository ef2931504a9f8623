// TEST-ONLY: the tile pipeline (tile_pipeline.hip, tile_pixel_mode.hip) on the host, kernel by kernel, in the order and with
// the launch geometry runtime.hip uses.  Shared by the two emulator libraries (emu_tile_driver.cpp, emu_driver.cpp).
#pragma once
#include "emu_launch.h"

#include "../../charls_amd/csrc/device/tile_pipeline.hip"
#include "../../charls_amd/csrc/device/block_stuffing.hip"
#include "../../charls_amd/csrc/device/speculative_stuffing.hip"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

static uint32_t g_counters[jls::tile::kCounters]; // of the last emu_encode_tile_pipeline call

// The tile pipeline (tile_pipeline.hip), kernel by kernel, in the order and with the launch geometry runtime.hip uses.
// job_events / warm_events as given: the tests use small values so that small images have many jobs, and warm-ups too
// short to converge so that settle_chains has to walk jobs again.
template <typename S>
static void emu_tile_pipeline(const jls::ScanDesc* descs, jls::ScanResult* results, int count, uint32_t job_events, uint32_t warm_events,
                              uint32_t run_job_events, uint32_t run_warm_events, uint32_t run_long_warm_events)
{
    using namespace jls;
    job_events = std::max<uint32_t>(32, job_events / 32 * 32); // (whole rounds of the walkers, as runtime.hip rounds it)
    const ScanDesc& p = descs[0];
    const tile::TilePlan plan = tile::plan_tiles(p);
    const size_t samples = (size_t)plan.samples;
    const uint32_t lines_per_tile = plan.lines_per_tile;
    const uint32_t tiles = plan.tiles;
    const size_t max_jobs = samples / job_events + pipe::kChains;
    const size_t max_run_jobs = samples / run_job_events + 2;
    std::vector<tile::Work> works(count);
    std::vector<pipe::Work> stuff(count);
    std::vector<void*> allocs;
    auto zalloc = [&](size_t bytes) {
        void* q = std::calloc(bytes + 64, 1);
        allocs.push_back(q);
        return q;
    };
    auto galloc = [&](size_t bytes) { // work areas the product does not clear are filled with garbage here
        void* q = std::malloc(bytes + 64);
        std::memset(q, 0xA5, bytes + 64);
        allocs.push_back(q);
        return q;
    };
    std::memset(g_counters, 0, sizeof g_counters);
    for (int i = 0; i < count; ++i)
    {
        tile::Work& w = works[i];
        const size_t raw_bytes = ((size_t)descs[i].stream_capacity + 64 + 15) / 16 * 16;
        w.keyinv = (uint16_t*)galloc(samples * 2);
        w.seg = (uint32_t*)galloc((size_t)(tiles + 1) * pipe::kChains * 4);
        w.chain_total = (uint32_t*)galloc(pipe::kChains * 4);
        w.chain_base = (uint32_t*)galloc(pipe::kChains * 4);
        w.job_first = (uint32_t*)galloc((pipe::kChains + 1) * 4);
        w.plan_part = (uint32_t*)galloc((size_t)tile::kPlanGroups * pipe::kChains * 4);
        const size_t slots = (size_t)tile::slots_capacity(samples, tile::run_slots_of<S>(), plan.lines) + tile::kSlack;
        w.rec = (uint32_t*)galloc(slots * sizeof(tile::Slot<S>));
        w.code = (uint32_t*)galloc(slots * sizeof(tile::Slot<S>));
        w.run_slots = tile::run_slots_of<S>();
        w.jobs = (tile::JobState*)galloc(max_jobs * sizeof(tile::JobState));
        w.run_jobs = (tile::RunJob*)galloc(max_run_jobs * sizeof(tile::RunJob));
        w.run_job_events = run_job_events;
        w.run_warm_events = run_warm_events;
        w.rare_warm_events = run_long_warm_events; // (the third knob of the run chain: the warm-up of the exact walk's segments)
        uint8_t* pack_state = (uint8_t*)zalloc((size_t)tiles * 16 + 16);
        w.blockbase = (uint64_t*)pack_state;
        w.tile_tail = w.blockbase + tiles;
        w.raw = (uint32_t*)galloc(raw_bytes); // (not cleared by the product either)
        w.raw_words = raw_bytes / 4;
        w.total_bits = (uint64_t*)galloc(8);
        w.status = (uint32_t*)galloc(4);
        w.counters = g_counters;
        w.lines_per_tile = lines_per_tile;
        w.tiles = tiles;
        w.job_events = job_events;
        w.warm_events = warm_events;
        w.segs_per_line = plan.segs_per_line;
        w.seg_pixels = plan.seg_pixels;
        w.tile_capacity = plan.tile_capacity;
        pipe::Work& sw = stuff[i];
        std::memset(&sw, 0, sizeof sw);
        sw.raw = w.raw;
        sw.raw_words = w.raw_words;
        sw.total_bits = w.total_bits;
        sw.status = w.status;
        sw.stuff_tables = (uint32_t*)galloc(std::max((raw_bytes / pipe::kStuffChunk + 2) * pipe::kStuffWords,
                                                     pipe::stuff_spec_table_words(raw_bytes, pipe::stuff_spec_geometry().chunk_bytes)) * 4);
    }
    const tile::Work* wk = works.data();
    const unsigned tiles_grid = 8 * ((tiles + 7) / 8);
    const bool pixel_mode = plan.mode == 2;
    const size_t lds_a = pixel_mode ? tile::analyze_pixel_lds_bytes(plan.lines_per_tile, plan.step, plan.max_pixels, plan.nc, (uint32_t)sizeof(S), plan.tile_capacity)
                                    : tile::analyze_lds_bytes(p.width, lines_per_tile, (uint32_t)sizeof(S), p.interleave_mode);
    const size_t lds_b = pixel_mode ? tile::sort_pixel_lds_bytes(plan.lines_per_tile, plan.step, plan.max_pixels, plan.nc, (uint32_t)sizeof(S), plan.tile_capacity)
                                    : tile::sort_lds_bytes(p.width, lines_per_tile, (uint32_t)sizeof(S), p.interleave_mode);
    if (pixel_mode)
        emu::launch(tile::analyze_pixel_tiles<S>, dim3(tiles_grid, count), dim3(tile::kThreads), lds_a, descs, wk);
    else
        emu::launch(tile::analyze_tiles<S, 0>, dim3(tiles_grid, count), dim3(tile::kThreads), lds_a, descs, wk);
    emu::launch(tile::sum_chains, dim3(tile::kPlanGroups, count), dim3(tile::kPlanThreads), 0, descs, wk);
    emu::launch(tile::plan_chains, dim3(count), dim3(tile::kPlanThreads), 0, descs, wk);
    emu::launch(tile::apply_chains, dim3(tile::kPlanGroups, count), dim3(tile::kPlanThreads), 0, descs, wk);
    if (pixel_mode)
        emu::launch(tile::sort_pixel_tiles<S>, dim3(tiles_grid, count), dim3(tile::kThreads), lds_b, descs, wk);
    else
        emu::launch(tile::sort_tiles<S, 0>, dim3(tiles_grid, count), dim3(tile::kThreads), lds_b, descs, wk);
    emu::launch(tile::walk_jobs<S>, dim3((unsigned)((max_jobs + 63) / 64), count), dim3(64), 0, descs, wk);
    emu::launch(tile::settle_chains<S>, dim3(pipe::kChains, count), dim3(64), 0, descs, wk, (uint32_t)count);
    const dim3 count_grid((unsigned)std::min<size_t>(max_run_jobs, 32), count), lanes((unsigned)((max_run_jobs * count + 63) / 64));
    if (pixel_mode)
        emu::launch(tile::count_runs<S, 1>, count_grid, dim3(64), 0, wk, plan.nc);
    else
        emu::launch(tile::count_runs<S, 0>, count_grid, dim3(64), 0, wk, 1u);
    emu::launch(tile::scan_runs, dim3(count), dim3(64), 0, wk);
    if (p.interleave_mode != 2)
    {
        if (pixel_mode)
            emu::launch(tile::compact_rare_runs<S, 1>, count_grid, dim3(64), 0, wk);
        else
            emu::launch(tile::compact_rare_runs<S, 0>, count_grid, dim3(64), 0, wk);
    }
    const unsigned rare_blocks = p.interleave_mode != 2 ? (unsigned)count : 0u;
#define EMU_RUN_CHAIN(ILV, FMT)                                                                                  \
    do                                                                                                           \
    {                                                                                                            \
        emu::launch(tile::warm_run_jobs<S, ILV, FMT>, dim3(rare_blocks + lanes.x), dim3(64), 0, descs, wk, (uint32_t)count, (uint32_t)rare_blocks); \
        emu::launch(tile::walk_run_jobs<S, ILV, FMT>, lanes, dim3(64), 0, descs, wk, (uint32_t)count);          \
        emu::launch(tile::settle_runs<S, ILV, FMT>, dim3((unsigned)count), dim3(64), 0, descs, wk, (uint32_t)count);       \
    } while (0)
    if (!pixel_mode)
        EMU_RUN_CHAIN(0, 0);
    else if (p.interleave_mode == 2)
        EMU_RUN_CHAIN(2, 1);
    else if (p.interleave_mode == 1)
        EMU_RUN_CHAIN(1, 1);
    else
        EMU_RUN_CHAIN(0, 1);
#undef EMU_RUN_CHAIN
    emu::launch(tile::pack_tiles<S>, dim3(tiles, count), dim3(tile::pack_threads_for(plan.tile_capacity)), tile::pack_lds_bytes(plan.tile_capacity, p.bits_per_sample), descs, wk);
    const pipe::Work* sk = stuff.data();
    if (const char* env = std::getenv("CHARLS_AMD_BLOCK_STUFFING"); env != nullptr && std::atoi(env) == 2)
    { // the speculative form (CHARLS_AMD_BLOCK_STUFFING=2 is a switch of this harness only: the product picks by batch size)
        const pipe::SpecGeometry spec = pipe::stuff_spec_geometry();
        size_t most = 0;
        for (int i = 0; i < count; ++i)
            most = std::max(most, (size_t)works[i].raw_words * 4);
        const unsigned waves = (unsigned)(most / spec.chunk_bytes + 1);
        emu::launch(pipe::stuff_spec_survey, dim3(waves, count), dim3(64), 0, sk, spec.chunk_bytes, spec.warm_bytes);
        emu::launch(pipe::stuff_spec_resolve, dim3(count), dim3(64), 0, sk, spec.chunk_bytes);
        emu::launch(pipe::stuff_spec_emit, dim3(waves, count), dim3(64), 0, descs, sk, results, spec.chunk_bytes);
    }
    else if (env == nullptr || std::atoi(env) != 0)
    {
        size_t most = 0;
        for (int i = 0; i < count; ++i)
            most = std::max(most, (size_t)works[i].raw_words * 4);
        const unsigned chunk_waves = (unsigned)((most / pipe::kStuffChunk + 1 + 63) / 64);
        const unsigned survey_blocks = pipe::stuff_survey_blocks(most);
        emu::launch(pipe::stuff_survey, dim3(survey_blocks, count), dim3(64), 0, sk);
        emu::launch(pipe::stuff_resolve, dim3(count), dim3(pipe::kStuffResolveThreads), 0, sk);
        emu::launch(pipe::stuff_emit, dim3(chunk_waves, count), dim3(64), 0, descs, sk, results);
    }
    else
        emu::launch(pipe::stuff_scan, dim3(count), dim3(64), 0, descs, sk, results);
    for (void* q : allocs)
        std::free(q);
}

