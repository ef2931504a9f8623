// TEST-ONLY driver: scan_group_decode.hip compiled for the host with its path counters switched on
// (tools/decode_path_profile.py).  One wavefront at a time, so the counters of one launch belong to that launch.
#define JLS_PATH_PROFILE 1
#include "emu_launch.h"

namespace emu {
BlockState* g_block = nullptr;
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
} // namespace emu

extern "C" unsigned long long jls_path_counts[32] = {};

#include "../../charls_amd/csrc/device/scan_group_decode.hip"

#include <cstring>

extern "C" {

int emu_sizeof_scan_desc() { return (int)sizeof(jls::ScanDesc); }

// 8-bit single-component scans, `group` lanes per scan; counts[32] receives the path counters of the launch.
int emu_profile_decode_group(const jls::ScanDesc* descs, jls::ScanResult* results, int count, int group,
                             unsigned long long* counts)
{
    const jls::ScanDesc& d = descs[0];
    if (d.bits_per_sample > 8 || d.interleave_mode != 0)
        return -1;
    const int per_wave = 64 / group;
    const size_t lds = jls::grp::workgroup_lds_bytes<uint8_t>(d.width, per_wave, 1);
    const dim3 grid((count + per_wave - 1) / per_wave);
    std::memset(jls_path_counts, 0, sizeof jls_path_counts);
#define EMU_PROFILE(G)                                                                                                                   \
    do                                                                                                                                   \
    {                                                                                                                                    \
        if (d.near_lossless != 0) emu::launch(jls::decode_scans_group<uint8_t, G, 1, 1, true>, grid, dim3(64), lds, descs, results, (uint32_t)count); \
        else emu::launch(jls::decode_scans_group<uint8_t, G, 1>, grid, dim3(64), lds, descs, results, (uint32_t)count);                  \
    } while (0)
    if (group == 8)
        EMU_PROFILE(8);
    else if (group == 16)
        EMU_PROFILE(16);
    else if (group == 32)
        EMU_PROFILE(32);
    else
        return -1;
#undef EMU_PROFILE
    std::memcpy(counts, jls_path_counts, sizeof jls_path_counts);
    return 0;
}

} // extern "C"
