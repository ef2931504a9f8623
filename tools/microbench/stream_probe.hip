// How fast does a lane stream 64-byte lines when the 64 lanes of a wavefront read from regions far apart (the layout of
// the encoder's chain stage: one lane per frame, frames hundreds of MB apart) compared with one dense region per wavefront?
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench/stream_probe.hip -o tools/microbench/build/stream_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((vector_size(16)));

// layout 0: lane l of wave w reads frame l's region at chain offset w; layout 1: wave w owns a dense block, lane l its own
// 64-byte column of every 4 KB row.
__global__ void __launch_bounds__(64) stream(const uint8_t* base, uint64_t frame_stride, uint64_t chain_stride, uint32_t lines,
                                             int layout, int busy_instructions, uint64_t* cycles, uint32_t* sink)
{
    const uint32_t w = blockIdx.x, l = threadIdx.x;
    const uint8_t* p = layout == 0 ? base + (uint64_t)l * frame_stride + (uint64_t)w * chain_stride
                                   : base + (uint64_t)w * lines * 4096ull + (uint64_t)l * 64;
    const uint64_t step = layout == 0 ? 64 : 4096;
    u32x4 a[4], b[4];
    uint32_t acc = 0;
    for (int j = 0; j < 4; ++j)
        a[j] = ((const u32x4*)p)[j];
    uint64_t t0 = 0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (uint32_t line = 0; line < lines; line += 2)
    {
        asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
        for (int j = 0; j < 4; ++j)
            b[j] = ((const u32x4*)(p + (uint64_t)(line + 1) * step))[j];
        for (int j = 0; j < 4; ++j)
            acc += a[j][0] ^ a[j][3];
        for (int k = 0; k < busy_instructions; ++k)
            asm volatile("v_add_u32 %0, %0, 1" : "+v"(acc));
        asm volatile("" ::"v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
        for (int j = 0; j < 4; ++j)
            a[j] = ((const u32x4*)(p + (uint64_t)(line + 2) * step))[j];
        for (int j = 0; j < 4; ++j)
            acc += b[j][1] ^ b[j][2];
        for (int k = 0; k < busy_instructions; ++k)
            asm volatile("v_add_u32 %0, %0, 1" : "+v"(acc));
    }
    uint64_t t1 = 0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (l == 0)
        cycles[w] = t1 - t0;
    sink[w * 64 + l] = acc;
}

int main()
{
    const uint64_t frame_stride = 376ull << 20; // one frame's work area
    const uint64_t bytes = frame_stride * 64 + (64ull << 20);
    uint8_t* base;
    if (hipMalloc(&base, bytes) != hipSuccess)
    {
        printf("hipMalloc of %.1f GB failed\n", bytes / 1e9);
        return 1;
    }
    (void)hipMemset(base, 1, bytes);
    uint64_t* d_cycles;
    uint32_t* d_sink;
    (void)hipMalloc(&d_cycles, 8 * 4096);
    (void)hipMalloc(&d_sink, 4 * 64 * 4096);
    const uint32_t lines = 4096; // 256 KB per lane
    printf("%-44s %8s %12s\n", "layout / wavefronts / arithmetic per line", "waves", "cyc per line");
    for (int busy : {0, 400})
        for (int layout : {0, 1})
            for (uint32_t waves : {1u, 16u, 256u, 1300u})
            {
                const uint64_t chain_stride = (frame_stride - lines * 64ull) / 1300 / 64 * 64;
                uint64_t best = ~0ull;
                for (int rep = 0; rep < 2; ++rep)
                {
                    hipLaunchKernelGGL(stream, dim3(waves), dim3(64), 0, 0, base, frame_stride, chain_stride, lines, layout, busy, d_cycles,
                                       d_sink);
                    uint64_t c[4096];
                    (void)hipMemcpy(c, d_cycles, 8 * waves, hipMemcpyDeviceToHost);
                    uint64_t worst = 0;
                    for (uint32_t i = 0; i < waves; ++i)
                        worst = c[i] > worst ? c[i] : worst;
                    best = worst < best ? worst : best;
                }
                char name[96];
                snprintf(name, sizeof name, "%s, %d v_add per line", layout == 0 ? "lane = frame (376 MB apart)" : "dense block per wavefront", busy);
                printf("%-44s %8u %12.1f\n", name, waves, (double)best / lines);
            }
    return 0;
}
