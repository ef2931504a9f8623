// How many wave64 vector instructions a gfx950 SIMD issues per cycle at saturation, by instruction kind and by the number of
// wavefronts that share the SIMD (1, 2, 4: one workgroup of 4, 8, 16 wavefronts).  Every
// wavefront runs four INDEPENDENT chains of the instruction, so nothing but issue limits it.  The ruler of bench.py's
// `issue_roofline` is the saturated figure of the decoder's instruction mix.  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench/issue_ceiling.hip -o tools/microbench/build/issue_ceiling
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

__device__ __forceinline__ uint64_t now()
{
    uint64_t t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

constexpr int kIterations = 64, kCopies = 16; // x (instructions per copy) instructions per wavefront

#define PROBE(NAME, BODY)                                                                                                         \
    __global__ void __launch_bounds__(1024) NAME(uint64_t* out, int* sink)                                                       \
    {                                                                                                                             \
        int v0 = threadIdx.x, v1 = 3, v2 = 5, v3 = 7, v4 = sink[0] + 9, v5 = 11;                                                  \
        const uint64_t t0 = now();                                                                                                \
        for (int it = 0; it < kIterations; ++it)                                                                                  \
            asm volatile(REP16(BODY) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5));                                \
        const uint64_t t1 = now();                                                                                                \
        if ((threadIdx.x & 63) == 0)                                                                                              \
            out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                                                  \
        sink[1 + (threadIdx.x & 63)] = v0 + v1 + v2 + v3;                                                                         \
    }

PROBE(p_add, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
PROBE(p_mad24, "v_mad_i32_i24 %0, %0, %4, %5\n v_mad_i32_i24 %1, %1, %4, %5\n v_mad_i32_i24 %2, %2, %4, %5\n v_mad_i32_i24 %3, %3, %4, %5\n")
PROBE(p_med3, "v_med3_i32 %0, %0, %4, %5\n v_med3_i32 %1, %1, %4, %5\n v_med3_i32 %2, %2, %4, %5\n v_med3_i32 %3, %3, %4, %5\n")
PROBE(p_alignbit, "v_alignbit_b32 %0, %0, %4, %5\n v_alignbit_b32 %1, %1, %4, %5\n v_alignbit_b32 %2, %2, %4, %5\n v_alignbit_b32 %3, %3, %4, %5\n")
PROBE(p_add3, "v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5\n")
PROBE(p_sdwa, "v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
              "v_add_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
              "v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
              "v_add_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n")
// the decoder's step loop in miniature: of its 62 vector instructions 26 are VOP3 (three sources), 9 SDWA, 27 plain VOP1 / VOP2
PROBE(p_mix, "v_add_u32 %0, %0, %4\n v_med3_i32 %1, %1, %4, %5\n v_add_u32 %2, %2, %4\n v_mad_i32_i24 %3, %3, %4, %5\n"
             "v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_xor_b32 %1, %1, %4\n"
             "v_add3_u32 %2, %2, %4, %5\n v_lshrrev_b32 %3, 1, %3\n")

struct Probe
{
    const char* name;
    void (*fn)(uint64_t*, int*);
    int per_copy;
};

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint64_t* d_out;
    int* d_sink;
    (void)hipMalloc(&d_out, sizeof(uint64_t) * 16 * 2 * cus);
    (void)hipMalloc(&d_sink, 4096);
    (void)hipMemset(d_sink, 0, 4096);
    const Probe probes[] = {{"v_add_u32 (VOP2)", p_add, 4},       {"v_mad_i32_i24 (VOP3)", p_mad24, 4}, {"v_med3_i32 (VOP3)", p_med3, 4},
                            {"v_alignbit_b32 (VOP3)", p_alignbit, 4}, {"v_add3_u32 (VOP3)", p_add3, 4},     {"v_add_u32_sdwa", p_sdwa, 4},
                            {"decoder mix (8 instructions)", p_mix, 8}};
    printf("%d CUs; wave64 vector instructions per SIMD and cycle (four independent chains per wavefront); in brackets: cycles per\n"
           "instruction of ONE wavefront\n", cus);
    printf("%-32s %18s %18s %18s\n", "instruction", "1 wave / SIMD", "2 waves / SIMD", "4 waves / SIMD");
    for (const Probe& p : probes)
    {
        printf("%-32s", p.name);
        for (int per_simd : {1, 2, 4})
        {
            const int waves = per_simd >= 4 ? 16 : per_simd * 4; // per workgroup
            const int groups = per_simd == 8 ? 2 * cus : cus;     // two workgroups of 16 wavefronts per CU
            std::vector<double> cycles;
            for (int r = 0; r < 3; ++r)
            {
                if (per_simd == 8)
                { // a cooperative launch: all 2 x CUs workgroups of 1024 threads are resident at once, i.e. two on every CU
                    void* args[] = {&d_out, &d_sink};
                    if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(p.fn), dim3(groups), dim3(64 * waves), args, 0, 0) != hipSuccess)
                    {
                        (void)hipGetLastError();
                        cycles.clear();
                        break;
                    }
                }
                else
                    hipLaunchKernelGGL(p.fn, dim3(groups), dim3(64 * waves), 0, 0, d_out, d_sink);
                std::vector<uint64_t> h(16 * groups);
                (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
                cycles.clear();
                for (int g = 0; g < groups; ++g)
                    for (int w = 0; w < waves; ++w)
                        cycles.push_back((double)h[g * 16 + w]);
            }
            if (cycles.empty())
            {
                printf("  %-18s", "(no co-residency)");
                continue;
            }
            std::sort(cycles.begin(), cycles.end());
            const double median = cycles[cycles.size() / 2];
            const double instructions = (double)kIterations * kCopies * p.per_copy;
            printf("  %6.3f (%5.2f)    ", per_simd * instructions / median, median / instructions);
        }
        printf("\n");
    }
    printf("(more than 4 wavefronts per SIMD need two workgroups on one CU; neither a grid of 2 x CUs workgroups nor a cooperative launch of it\n"
           " made them run side by side on this box -- every wavefront took what it takes with 4 per SIMD -- so that column is not reported)\n");
    return 0;
}
