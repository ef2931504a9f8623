// TEST-ONLY driver: exposes the gfx950 kernels, compiled for the host, to the CPU test-suite (tests/test_emu_*.py).
#include "emu_launch.h"

namespace emu {
BlockState* g_block = nullptr;
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
} // namespace emu

#include "../../charls_amd/csrc/device/scan_serial.hip"

extern "C" {

void emu_encode_scans_serial(const jls::ScanDesc* descs, jls::ScanResult* results, int count)
{
    emu::launch(jls::encode_scans_serial, dim3(count), dim3(64), 0, descs, results);
}

void emu_decode_scans_serial(const jls::ScanDesc* descs, jls::ScanResult* results, int count)
{
    emu::launch(jls::decode_scans_serial, dim3(count), dim3(64), 0, descs, results);
}

size_t emu_sizeof_scan_desc() { return sizeof(jls::ScanDesc); }
}
