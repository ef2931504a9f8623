// TEST-ONLY driver of the tile pipeline (tile_pipeline.hip compiled for the host), a library of its own so that it
// builds in seconds (tests/test_emu_tile_pipeline.py).
#include "emu_launch.h"

namespace emu {
BlockState* g_block = nullptr;
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
} // namespace emu

#include "emu_tile_pipeline.h"

extern "C" {

void emu_encode_tile_pipeline(const jls::ScanDesc* descs, jls::ScanResult* results, int count, uint32_t job_events, uint32_t warm_events,
                              uint32_t run_job_events, uint32_t run_warm_events, uint32_t run_long_warm_events)
{
    if (descs[0].bits_per_sample > 8)
        emu_tile_pipeline<uint16_t>(descs, results, count, job_events, warm_events, run_job_events, run_warm_events, run_long_warm_events);
    else
        emu_tile_pipeline<uint8_t>(descs, results, count, job_events, warm_events, run_job_events, run_warm_events, run_long_warm_events);
}

void emu_tile_counters(uint32_t* out)
{
    for (uint32_t i = 0; i < jls::tile::kCounters; ++i)
        out[i] = g_counters[i];
}

size_t emu_sizeof_scan_desc() { return sizeof(jls::ScanDesc); }
}
