// Does the priority of a stream change how fast a LONE kernel runs on it?  One wavefront of dependent integer work (the shape of
// the decoder: one chain, nothing to overlap) on a stream made by hipStreamCreateWithFlags and on streams of every priority the
// device offers.  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench/stream_priority.hip -o tools/microbench/build/stream_priority
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void chain(unsigned* out, int n)
{
    unsigned x = threadIdx.x + 1;
    for (int i = 0; i < n; ++i)
        x = x * 1664525u + 1013904223u + (x >> 7);
    out[threadIdx.x] = x;
}

static double run(hipStream_t s, unsigned* d, int n)
{
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s, d, 1000); // warm
    (void)hipStreamSynchronize(s);
    const auto a = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s, d, n);
    (void)hipStreamSynchronize(s);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
}

int main()
{
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    printf("hipDeviceGetStreamPriorityRange: least %d, greatest %d\n", least, greatest);
    unsigned* d;
    (void)hipMalloc(&d, 256);
    const int n = 60000000;
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int p = 99;
    (void)hipStreamGetPriority(s, &p);
    for (int r = 0; r < 2; ++r)
        printf("hipStreamCreateWithFlags (priority %d): %.1f ms\n", p, run(s, d, n));
    const int lo = least < greatest ? least : greatest, hi = least < greatest ? greatest : least;
    for (int q = lo - 1; q <= hi + 1; ++q)
    {
        hipStream_t t;
        if (hipStreamCreateWithPriority(&t, hipStreamNonBlocking, q) != hipSuccess)
        {
            printf("priority %d: not created\n", q);
            (void)hipGetLastError();
            continue;
        }
        int got = 99;
        (void)hipStreamGetPriority(t, &got);
        for (int r = 0; r < 2; ++r)
            printf("hipStreamCreateWithPriority(%d) -> priority %d: %.1f ms\n", q, got, run(t, d, n));
    }
    printf("null stream: %.1f ms\n", run(nullptr, d, n));
    return 0;
}
