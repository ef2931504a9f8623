// The decoder's step loop (charls_amd/csrc/device/scan_group_step.inc) alone, on synthetic LDS contents that keep every lane
// decoding: cycles per step of the product's text and of other compositions of its pieces (stores by all lanes, one step per
// trip, ...; round 6's first form and its ablations: tools/microbench/steploop_variants.py at commit 2de007a,
// profiles/r06_steploop_v1_ablation.txt), with the
// product's launch shape -- four wavefronts per workgroup (one per SIMD), four scans per wavefront, the product's LDS layout.
// Not part of the product.  Build: hipcc --offload-arch=gfx950 -O2 -Icharls_amd/csrc/device tools/microbench/steploop.hip -o tools/microbench/build/steploop
// Run: tools/microbench/build/steploop [workgroups]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../charls_amd/csrc/device/scan_group_step.inc"

#define WINDOW_FIRST JLS_STEP_ORDER_WINDOW_FIRST
#define UPDATE_FIRST JLS_STEP_ORDER_UPDATE_FIRST
#define PREDICTOR_LATE JLS_STEP_ORDER_PREDICTOR_LATE
// one step per trip: the one copy requests the next entry into its own pair (the predictor has to come before the request)
#define RARE(SFX, ENEXT) JLS_STEP_RARE(SFX, "ds_read_u8", "ds_write_b8", "1", ENEXT, JLS_RUN_WHICH_LOSSLESS, JLS_RUN_RECON_LOSSLESS)
#define LOOP1(ORDER, STORES, KSEEN)                                                                                               \
    JLS_STEP_PROLOGUE JLS_STEP_BODY_X("a", "s_branch L_stepa%=\n", "v118", "v119", "v[118:119]", "BYTE_0", "1",                   \
                                      ORDER("BYTE_0", "1", "", "v118", "v119"), STORES, "", " offset:255", "", "", "", KSEEN,       \
                                      JLS_STEP_LOSSLESS)                                                                           \
        RARE("a", "v[118:119]") JLS_STEP_EPILOGUE("")
#define LOOP2(ORDER, STORES, KSEEN)                                                                                               \
    JLS_STEP_PROLOGUE JLS_STEP_BODY_X("a", "", "v118", "v119", "v[106:107]", "BYTE_0", "1", ORDER("BYTE_0", "1", "", "v118", "v119"), \
                                      STORES, "", " offset:255", "", "", "", KSEEN, JLS_STEP_LOSSLESS)                              \
        JLS_STEP_BODY_X("b", "s_branch L_stepa%=\n", "v106", "v107", "v[118:119]", "BYTE_0", "1",                                  \
                        ORDER("BYTE_0", "1", "", "v106", "v107"), STORES, "", " offset:255", "", "", "", KSEEN, JLS_STEP_LOSSLESS)   \
            RARE("a", "v[106:107]") RARE("b", "v[118:119]") JLS_STEP_EPILOGUE("")
#define NO_STORES "", "1", "", "0"

#define STEPLOOP_NAME_0 "one step per trip, both stores early"
#define STEPLOOP_TEXT_0 LOOP1(WINDOW_FIRST, JLS_STEP_STORES_EARLY("ds_write_b8"), JLS_STEP_KSEEN)
#define STEPLOOP_NAME_1 "one step per trip, sample store late"
#define STEPLOOP_TEXT_1 LOOP1(WINDOW_FIRST, JLS_STEP_STORES_LATE("ds_write_b8"), JLS_STEP_KSEEN)
#define STEPLOOP_NAME_2 "two steps per trip, both stores early"
#define STEPLOOP_TEXT_2 LOOP2(WINDOW_FIRST, JLS_STEP_STORES_EARLY("ds_write_b8"), JLS_STEP_KSEEN)
#define STEPLOOP_NAME_3 "PRODUCT: two steps per trip, sample late"
#define STEPLOOP_TEXT_3 LOOP2(WINDOW_FIRST, JLS_STEP_STORES_LATE("ds_write_b8"), JLS_STEP_KSEEN)
#define STEPLOOP_NAME_4 "product without k_seen"
#define STEPLOOP_TEXT_4 LOOP2(WINDOW_FIRST, JLS_STEP_STORES_LATE("ds_write_b8"), "")
#define STEPLOOP_NAME_5 "product without the two stores"
#define STEPLOOP_TEXT_5 LOOP2(WINDOW_FIRST, NO_STORES, JLS_STEP_KSEEN)
#define STEPLOOP_NAME_6 "A.13 and the record store first"
#define STEPLOOP_TEXT_6 LOOP2(UPDATE_FIRST, JLS_STEP_STORES_SAMPLE_LATE("ds_write_b8"), JLS_STEP_KSEEN)
#define STEPLOOP_NAME_7 "predictor behind the record request"
#define STEPLOOP_TEXT_7 LOOP2(PREDICTOR_LATE, JLS_STEP_STORES_LATE("ds_write_b8"), JLS_STEP_KSEEN)
#define STEPLOOP_VARIANTS 8

constexpr uint32_t kRegion = 9744, kRecords = 0, kRun = 2928, kRing = 2976, kPrep = 4008, kLine = 5584, kLut = 512;
constexpr int kBurst = 60;

__device__ __forceinline__ uint64_t now()
{
    uint64_t t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__device__ int quantize(int d)
{
    const int pos = (d > 0) + (d >= 3) + (d >= 7) + (d >= 21), neg = (d < 0) + (d <= -3) + (d <= -7) + (d <= -21);
    return pos - neg;
}

#define KERNEL(V)                                                                                                                 \
    __global__ void __launch_bounds__(256) steploop_##V(uint64_t* out, uint32_t* sink, int repeats)                              \
    {                                                                                                                             \
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                                                      \
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sid = lane / 16, sub = lane % 16;                             \
        unsigned char* region = smem + kLut + (size_t)(wave * 4 + sid) * kRegion;                                                 \
        for (int q = threadIdx.x; q < 511; q += 256)                                                                              \
            smem[q] = (unsigned char)((quantize(q - 255) + 4) * 8);                                                               \
        uint32_t* records = reinterpret_cast<uint32_t*>(region + kRecords);                                                       \
        for (int q = sub; q < 366; q += 16)                                                                                       \
        {                                                                                                                         \
            records[2 * q] = 4;                                                                                                   \
            records[2 * q + 1] = q == 0 ? 64 : 1; /* record 0: run mode, N preset to RESET */                                     \
        }                                                                                                                         \
        uint32_t* ring = reinterpret_cast<uint32_t*>(region + kRing);                                                             \
        uint32_t seed = 12345u + 977u * (blockIdx.x * 16 + wave * 4 + sid);                                                       \
        for (int q = sub; q < 258; q += 16)                                                                                       \
        {                                                                                                                         \
            uint32_t x = seed + 2654435761u * (uint32_t)q;                                                                        \
            x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;                                                                         \
            ring[q] = x | 0x00010001u; /* a one bit in every 16: no prefix is longer than 15 */                                   \
        }                                                                                                                         \
        uint32_t* prep = reinterpret_cast<uint32_t*>(region + kPrep);                                                             \
        for (int e = sub; e < 192; e += 16)                                                                                       \
        {                                                                                                                         \
            const uint32_t q1 = (e % 2) ? 24 : 40, q2 = 24 + 8 * (uint32_t)((e / 2) % 3); /* never the run-mode context */        \
            prep[2 * e] = (100u + (uint32_t)(e % 3)) | ((9u * (9u * q1 + q2)) << 16);                                             \
            prep[2 * e + 1] = 101;                                                                                                \
        }                                                                                                                         \
        __syncthreads();                                                                                                          \
        const uint32_t ring_address = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(region + kRing);             \
        const uint32_t records_address = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(region + kRecords);       \
        const uint32_t prep_address = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(region + kPrep);             \
        const uint32_t line_address = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(region + kLine);             \
        int a = 100, u_a = 0, u_n1 = 2, u_tb = 0, u_cc = 0;                                                                       \
        uint32_t p = 0, u1 = 0, k_last = 0, k_seen = 0, qsu8 = 0, win_now = 0, where = records_address + 365 * 8;                 \
        const uint32_t limit_v = 23, cfg_v = 22u | (8u << 8) | (64u << 16); /* escape_base | qbpp << 8 | RESET << 16 */           \
        const uint32_t run_ctx_address = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(region + kRun);           \
        const int maxval_s = 255;                                                                                                 \
        int run_index = 0;                                                                                                        \
        const unsigned long long in_line_m = ~0ull;                                                                               \
        unsigned long long fail_m = 0, failed = 0;                                                                                \
        const uint64_t t0 = now();                                                                                                \
        for (int r = 0; r < repeats; ++r)                                                                                         \
        {                                                                                                                         \
            uint32_t lm = line_address + ((uint32_t)(r * 64) & 2047u);                                                            \
            uint32_t pp = prep_address + (((uint32_t)r * 24u) & 127u) * 8u - 8u;                                                  \
            uint32_t count = kBurst - 1;                                                                                          \
            asm volatile(STEPLOOP_TEXT_##V                                                                                        \
                         : [a] "+v"(a), [p] "+v"(p), [lm] "+v"(lm), [pp] "+v"(pp), [where] "+v"(where), [ua] "+v"(u_a),            \
                           [un1] "+v"(u_n1), [utb] "+v"(u_tb), [cc] "+v"(u_cc), [u1] "+v"(u1), [k] "+v"(k_last),                   \
                           [kseen] "+v"(k_seen), [qsu] "+v"(qsu8), [win] "+v"(win_now), [cnt] "+s"(count), [fail] "=&s"(fail_m),    \
                           [ri] "+v"(run_index)                                                                                   \
                         : [ring] "v"(ring_address), [recbase] "v"(records_address), [limitv] "v"(limit_v), [cfg] "v"(cfg_v),     \
                           [inl] "s"(in_line_m), [smax] "s"(maxval_s), [rctx] "v"(run_ctx_address), [gl] "n"(16)                   \
                         : JLS_STEP_LOOP_CLOBBERS);                                                                                             \
            failed |= fail_m;                                                                                                     \
        }                                                                                                                         \
        const uint64_t t1 = now();                                                                                                \
        if (lane == 0)                                                                                                            \
        {                                                                                                                         \
            out[2 * (blockIdx.x * 4 + wave)] = t1 - t0;                                                                           \
            out[2 * (blockIdx.x * 4 + wave) + 1] = failed;                                                                        \
        }                                                                                                                         \
        sink[threadIdx.x] = (uint32_t)a + p + k_seen + qsu8 + win_now + (uint32_t)u_a;                                            \
    }

KERNEL(0) KERNEL(1) KERNEL(2) KERNEL(3) KERNEL(4) KERNEL(5) KERNEL(6) KERNEL(7)
static_assert(STEPLOOP_VARIANTS == 8, "one KERNEL() per variant");

typedef void (*Kernel)(uint64_t*, uint32_t*, int);

int main(int argc, char** argv)
{
    const int groups = argc > 1 ? atoi(argv[1]) : 256;
    const int repeats = 2000;
    const Kernel kernels[] = {steploop_0, steploop_1, steploop_2, steploop_3, steploop_4, steploop_5, steploop_6, steploop_7};
    const char* names[] = {STEPLOOP_NAME_0, STEPLOOP_NAME_1, STEPLOOP_NAME_2, STEPLOOP_NAME_3, STEPLOOP_NAME_4, STEPLOOP_NAME_5, STEPLOOP_NAME_6,
                           STEPLOOP_NAME_7};
    uint64_t* d_out;
    uint32_t* d_sink;
    (void)hipMalloc(&d_out, sizeof(uint64_t) * 2 * 4 * groups);
    (void)hipMalloc(&d_sink, 4096);
    const size_t lds = kLut + 16 * kRegion;
    printf("step loop alone: %d workgroups of 4 wavefronts (one per SIMD), 4 scans per wavefront, %zu bytes of LDS, %d bursts of %d steps\n",
           groups, lds, repeats, kBurst);
    printf("%-44s %12s %12s %12s %s\n", "variant", "cyc/step min", "median", "max", "lanes that stopped");
    for (int v = 0; v < STEPLOOP_VARIANTS; ++v)
    {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[v]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        std::vector<double> per;
        unsigned long long failed = 0;
        for (int r = 0; r < 2; ++r)
        {
            hipLaunchKernelGGL(kernels[v], dim3(groups), dim3(256), lds, 0, d_out, d_sink, repeats);
            std::vector<uint64_t> h(2 * 4 * groups);
            if (hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
            {
                printf("%s: launch failed: %s\n", names[v], hipGetErrorString(hipGetLastError()));
                break;
            }
            per.clear();
            failed = 0;
            for (int w = 0; w < 4 * groups; ++w)
            {
                per.push_back((double)h[2 * w] / ((double)repeats * kBurst));
                failed |= h[2 * w + 1];
            }
        }
        if (per.empty())
            continue;
        std::sort(per.begin(), per.end());
        printf("%-44s %12.1f %12.1f %12.1f %016llx\n", names[v], per.front(), per[per.size() / 2], per.back(), failed);
    }
    return 0;
}
