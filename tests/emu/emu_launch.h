// TEST-ONLY: runs a __global__ function (compiled for the host against tests/emu/hip/hip_runtime.h) block by block,
// one OS thread per GPU thread.
#pragma once
#include <hip/hip_runtime.h>

#include <thread>
#include <vector>

namespace emu {

template <typename Kernel, typename... Args>
void launch(Kernel kernel, dim3 grid, dim3 block, size_t dyn_shared_bytes, Args... args)
{
    // One set of OS threads per launch; the blocks of the grid run one after the other on it (a barrier between blocks).
    const int n = (int)(block.x * block.y * block.z);
    std::vector<unsigned char> dyn_store(dyn_shared_bytes + 64);
    unsigned char* dyn_aligned = dyn_store.data() + (64 - (reinterpret_cast<uintptr_t>(dyn_store.data()) & 63)) % 64;
    BlockState st;
    st.nthreads = n;
    st.dyn_shared = dyn_aligned;
    st.block_barrier.init(n);
    const int waves = (n + kWave - 1) / kWave;
    for (int w = 0; w < waves; ++w)
    {
        const int in_wave = (w == waves - 1) ? n - w * kWave : kWave;
        st.wave_barrier[w].init(in_wave);
    }
    SpinBarrier between;
    between.init(n);
    g_block = &st;
    std::vector<std::thread> threads;
    threads.reserve(n);
    for (int t = 0; t < n; ++t)
        threads.emplace_back([=, &between]() {
            t_threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            t_blockDim = block;
            t_gridDim = grid;
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx)
                    {
                        t_blockIdx = dim3(bx, by, bz);
                        kernel(args...);
                        between.wait();
                    }
        });
    for (auto& th : threads)
        th.join();
}

} // namespace emu
