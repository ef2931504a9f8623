// group_launch.h -- launches of the lanes-per-scan kernels that live in translation units of their own (their template
// instantiations are most of the device code: compiling them next to runtime.hip instead of inside it keeps the build
// parallel).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "runtime.h"
#include "scan_types.h"

namespace jls::dev {

constexpr size_t kMaxDynamicLds = 64 * 1024;   // what a kernel may ask for without hipFuncSetAttribute
constexpr size_t kGroupDecodeLds = 160 * 1024; // a workgroup of the group kernels may take the whole LDS of a CU

// scan_group_pixels.hip (launch_pixels.hip)
size_t pixel_group_lds_bytes(const ScanDesc& d, uint32_t scans_per_wave);
void launch_decode_pixels(const ScanDesc& proto, int lanes, const ScanDesc* d_descs, ScanResult* d_results, uint32_t count,
                          hipStream_t stream);

// scan_group_encode.hip (launch_group_encode.hip)
size_t group_encode_lds_bytes(const ScanDesc& d, uint32_t scans_per_wave);
void launch_encode_group(const ScanDesc& proto, int lanes, const ScanDesc* d_descs, ScanResult* d_results, uint32_t count,
                         hipStream_t stream);

} // namespace jls::dev
