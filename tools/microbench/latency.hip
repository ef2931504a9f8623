// Single-wavefront latency / issue-rate probes for gfx950 (s_memtime ticks = shader cycles).  Not part of the product:
// the numbers calibrate the per-sample cost model of the decoders.
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench/latency.hip -o tools/microbench/build/latency
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__device__ __forceinline__ uint64_t now()
{
    uint64_t t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// Every probe: 64 copies of a sequence inside a 16-iteration loop; cycles per copy are reported.
#define PROBE(NAME, SETUP, BODY, ...)                                                                                   \
    __global__ void NAME(uint64_t* out, int* sink)                                                                      \
    {                                                                                                                   \
        extern __shared__ int lds[];                                                                                    \
        for (int i = threadIdx.x; i < 4096; i += 64)                                                                    \
            lds[i] = ((i * 4 + 64) & 16383);                                                                           \
        __syncthreads();                                                                                                \
        int v0 = threadIdx.x & 1, v1 = 3, v2 = 5, v3 = 7, v4 = 9;                                                       \
        int s0 = sink[0], s1 = 3;                                                                                       \
        SETUP;                                                                                                          \
        const uint64_t t0 = now();                                                                                      \
        for (int it = 0; it < 16; ++it)                                                                                 \
        {                                                                                                               \
            asm volatile(REP64(BODY) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+s"(s0), "+s"(s1)::"memory", "vcc", ##__VA_ARGS__); \
        }                                                                                                               \
        const uint64_t t1 = now();                                                                                      \
        if (threadIdx.x == 0)                                                                                           \
            out[0] = t1 - t0;                                                                                           \
        sink[1 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + s0 + s1;                                                       \
    }

PROBE(p_empty, , "")
PROBE(p_vadd_dep, , "v_add_u32 %0, %0, %1\n")
PROBE(p_vadd_ind4, , "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
PROBE(p_vadd_dep2x, , "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n")
PROBE(p_vmad24_dep, , "v_mad_i32_i24 %0, %0, %1, %2\n")
PROBE(p_vmullo_dep, , "v_mul_lo_u32 %0, %0, %1\n")
PROBE(p_vlshl64_dep, , "v_lshlrev_b64 v[100:101], %1, v[100:101]\n", "v100", "v101")
PROBE(p_vffbh_dep, , "v_ffbh_u32 %0, %0\n")
PROBE(p_vmed3_dep, , "v_med3_i32 %0, %0, %1, %2\n")
PROBE(p_valign_dep, , "v_alignbit_b32 %0, %0, %1, 3\n")
PROBE(p_sadd_dep, , "s_add_u32 %5, %5, %6\n")
PROBE(p_sadd_ind2, , "s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n")
PROBE(p_vcmp_cndmask, , "v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n")
PROBE(p_vcmp_sgpr_cnd, , "v_cmp_lt_i32 s[40:41], %0, %1\n v_cndmask_b32 %0, %0, %2, s[40:41]\n", "s40", "s41")
PROBE(p_readlane_rt, , "v_readlane_b32 %5, %0, 3\n v_add_u32 %0, %5, %1\n")
PROBE(p_readfirst_rt, , "v_readfirstlane_b32 %5, %0\n v_add_u32 %0, %5, %1\n")
PROBE(p_ds_read_dep, v0 = (threadIdx.x * 4) & 255, "ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n")
PROBE(p_ds_read64_dep, v0 = (threadIdx.x * 8) & 255, "ds_read_b64 v[100:101], %0\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v100\n", "v100", "v101")
PROBE(p_ds_readu8_dep, v0 = 0, "ds_read_u8 %0, %0\n s_waitcnt lgkmcnt(0)\n")
PROBE(p_ds_read_same_dep, v0 = 64, "ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") /* all lanes one address (broadcast) */
PROBE(p_ds_read_ind4, v0 = 0; v1 = 64; v2 = 128; v3 = 192,
      "ds_read_b32 %0, %0\n ds_read_b32 %1, %1\n ds_read_b32 %2, %2\n ds_read_b32 %3, %3\n s_waitcnt lgkmcnt(0)\n")
PROBE(p_ds_write_read, v0 = 64, "ds_write_b32 %0, %0\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n")
PROBE(p_ds_read_then_valu8, v0 = 64,
      "ds_read_b32 %0, %0\n v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4\n"
      "v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4\n s_waitcnt lgkmcnt(0)\n")
PROBE(p_ds_read_then_valu24, v0 = 64,
      "ds_read_b32 %0, %0\n" REP4("v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4\n"
                                  "v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4\n") "s_waitcnt lgkmcnt(0)\n")
PROBE(p_bpermute_dep, v0 = threadIdx.x * 4; v1 = 0, "ds_bpermute_b32 %1, %0, %1\n s_waitcnt lgkmcnt(0)\n")
PROBE(p_dpp_bcast_dep, , "v_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
PROBE(p_dpp_shr_dep, , "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")
PROBE(p_saveexec, , "s_and_saveexec_b64 s[40:41], vcc\n s_or_b64 exec, exec, s[40:41]\n", "s40", "s41")
PROBE(p_vadd_sadd_mix, , "v_add_u32 %0, %0, %1\n s_add_u32 %5, %5, 1\n")
PROBE(p_branch_taken, , "s_branch 0\n")
PROBE(p_cbranch_not_taken, , "s_cmp_eq_u32 %5, %5\n s_cbranch_scc0 0\n")
PROBE(p_cbranch_taken, , "s_cmp_eq_u32 %5, %5\n s_cbranch_scc1 0\n")
PROBE(p_branch_taken_far, , "s_branch 15\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
      "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
      "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n")
PROBE(p_waitcnt_idle, , "s_waitcnt lgkmcnt(0)\n")
PROBE(p_snop, , "s_nop 0\n")
PROBE(p_vperm, , "v_perm_b32 %0, %0, %1, %2\n")
PROBE(p_vcvt, , "v_cvt_f32_u32 %0, %0\n")
PROBE(p_vbfrev, , "v_bfrev_b32 %0, %0\n")
PROBE(p_ds_write_b8, v0 = 64, "ds_write_b8 %0, %1\n")
PROBE(p_ds_write2, v0 = 64, "ds_write2_b32 %0, %1, %2 offset1:1\n")
PROBE(p_sdwa_dep, , "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n")

// Does the LDS address VGPR + immediate offset wrap modulo 2^32 (a negative register with a larger positive offset)?
__global__ void negative_lds_address(int* out)
{
    extern __shared__ int lds[];
    for (int i = threadIdx.x; i < 4096; i += 64)
        lds[i] = i * 3 + 1;
    __syncthreads();
    int addr = -8 - 4 * (int)threadIdx.x; // bytes
    int v;
    asm volatile("ds_read_b32 %0, %1 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x] = v; // expected lds[(1024 - 8 - 4 t) / 4]
}

struct Probe
{
    const char* name;
    void (*fn)(uint64_t*, int*);
    int ops; // instructions per copy (for the per-instruction column)
};

int main(int argc, char** argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 1; // co-resident wavefronts on ONE CU (one workgroup)
    uint64_t* d_out;
    int* d_sink;
    (void)hipMalloc(&d_out, 64);
    (void)hipMalloc(&d_sink, 4096);
    (void)hipMemset(d_sink, 0, 4096);
    const Probe probes[] = {
        {"empty loop", p_empty, 1},
        {"v_add dependent", p_vadd_dep, 1},
        {"v_add 4 independent chains", p_vadd_ind4, 4},
        {"v_add 2 independent chains", p_vadd_dep2x, 2},
        {"v_mad_i32_i24 dependent", p_vmad24_dep, 1},
        {"v_mul_lo_u32 dependent", p_vmullo_dep, 1},
        {"v_lshlrev_b64 dependent", p_vlshl64_dep, 1},
        {"v_ffbh_u32 dependent", p_vffbh_dep, 1},
        {"v_med3_i32 dependent", p_vmed3_dep, 1},
        {"v_alignbit_b32 dependent", p_valign_dep, 1},
        {"s_add dependent", p_sadd_dep, 1},
        {"s_add 2 independent", p_sadd_ind2, 2},
        {"v_cmp vcc -> v_cndmask (pair)", p_vcmp_cndmask, 2},
        {"v_cmp sgpr -> v_cndmask (pair)", p_vcmp_sgpr_cnd, 2},
        {"v_readlane -> v_add (pair)", p_readlane_rt, 2},
        {"v_readfirstlane -> v_add (pair)", p_readfirst_rt, 2},
        {"ds_read_b32 dependent (per-lane addr)", p_ds_read_dep, 1},
        {"ds_read_u8 dependent", p_ds_readu8_dep, 1},
        {"ds_read_b64 dependent (+ v_mov)", p_ds_read64_dep, 2},
        {"ds_read_b32 dependent (one address)", p_ds_read_same_dep, 1},
        {"4 x ds_read_b32 independent + wait", p_ds_read_ind4, 4},
        {"ds_write_b32 + ds_read_b32 same addr", p_ds_write_read, 2},
        {"ds_read + 8 v_add + wait", p_ds_read_then_valu8, 9},
        {"ds_read + 24 v_add + wait", p_ds_read_then_valu24, 25},
        {"ds_bpermute dependent", p_bpermute_dep, 1},
        {"v_mov_dpp row_newbcast dependent", p_dpp_bcast_dep, 1},
        {"v_add_dpp row_shr dependent", p_dpp_shr_dep, 1},
        {"s_and_saveexec + s_or exec (pair)", p_saveexec, 2},
        {"v_add + s_add alternating (pair)", p_vadd_sadd_mix, 2},
        {"v_add_sdwa dependent", p_sdwa_dep, 1},
        {"s_branch to the next instruction", p_branch_taken, 1},
        {"s_cmp + s_cbranch not taken (pair)", p_cbranch_not_taken, 2},
        {"s_cmp + s_cbranch taken (pair)", p_cbranch_taken, 2},
        {"s_branch over 15 instructions", p_branch_taken_far, 1},
        {"s_waitcnt with nothing outstanding", p_waitcnt_idle, 1},
        {"s_nop 0", p_snop, 1},
        {"v_perm_b32 dependent", p_vperm, 1},
        {"v_cvt_f32_u32 dependent", p_vcvt, 1},
        {"v_bfrev_b32 dependent", p_vbfrev, 1},
        {"ds_write_b8 back to back", p_ds_write_b8, 1},
        {"ds_write2_b32 back to back", p_ds_write2, 1},
    };
    printf("wavefronts in the workgroup: %d\n", waves);
    printf("%-42s %10s %10s\n", "probe", "cyc/copy", "cyc/instr");
    for (const Probe& p : probes)
    {
        uint64_t best = ~0ull;
        for (int r = 0; r < 3; ++r)
        {
            hipLaunchKernelGGL(p.fn, dim3(1), dim3(64 * waves), 16384, 0, d_out, d_sink);
            uint64_t t = 0;
            (void)hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost);
            if (t < best)
                best = t;
        }
        const double per = (double)best / (16.0 * 64.0);
        printf("%-42s %10.2f %10.2f\n", p.name, per, per / p.ops);
    }
    {
        int* d_v;
        (void)hipMalloc(&d_v, 256);
        hipLaunchKernelGGL(negative_lds_address, dim3(1), dim3(64), 16384, 0, d_v);
        int h[64];
        (void)hipMemcpy(h, d_v, 256, hipMemcpyDeviceToHost);
        int good = 0;
        for (int t = 0; t < 64; ++t)
            good += h[t] == ((1024 - 8 - 4 * t) / 4) * 3 + 1;
        printf("negative LDS address register + offset wraps correctly for %d of 64 lanes (lane 0 read %d, expected %d)\n", good,
               h[0], ((1024 - 8) / 4) * 3 + 1);
    }
    return 0;
}
