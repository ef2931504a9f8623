// tests/emu/hip/hip_runtime.h -- TEST-ONLY stand-in for <hip/hip_runtime.h>.
//
// There is no GPU in the build container.  To debug the gfx950 kernels' LOGIC before spending GPU minutes, the tests
// compile the unmodified kernel sources (charls_amd/csrc/device/*.hip) with g++ against this header and run every
// workgroup as a set of OS threads with real barriers (tests/emu/emu_driver.cpp).  It models: thread/block indices,
// __shared__ (one workgroup runs at a time, so `static` is the right storage), __syncthreads, 64-lane wave shuffles /
// ballots, and the atomics the kernels use.  It is never part of the product and says nothing about performance.
#pragma once
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define JLS_GLOBAL_AS
#define JLS_DYNAMIC_LDS(name) unsigned char* name = emu::g_block->dyn_shared
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__

struct dim3
{
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
constexpr int kWave = 64;

// Sense-reversing barrier that yields instead of sleeping: the emulated lanes hit a barrier every few instructions and
// there are many more of them than cores, so futex sleeps/wake-ups dominated the run time of the CPU test suite.
struct SpinBarrier
{
    std::atomic<int> waiting{0};
    std::atomic<int> generation{0};
    int parties{0};
    void init(int n)
    {
        parties = n;
        waiting.store(0);
        generation.store(0);
    }
    void wait()
    {
        const int gen = generation.load(std::memory_order_acquire);
        if (waiting.fetch_add(1, std::memory_order_acq_rel) + 1 == parties)
        {
            waiting.store(0, std::memory_order_relaxed);
            generation.fetch_add(1, std::memory_order_release);
            return;
        }
        while (generation.load(std::memory_order_acquire) == gen)
            sched_yield();
    }
};

struct BlockState
{
    SpinBarrier block_barrier;
    SpinBarrier wave_barrier[16];
    unsigned long long xchg[16][kWave]; // per-wave exchange slots for shuffles / ballots
    int nthreads;
    unsigned char* dyn_shared;
};
extern BlockState* g_block;
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
} // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)
static const int warpSize = 64;

inline void __syncthreads()
{
    emu::g_block->block_barrier.wait();
}

inline int __clz(int x)
{
    return x == 0 ? 32 : __builtin_clz((unsigned)x);
}
inline int __clzll(long long x)
{
    return x == 0 ? 64 : __builtin_clzll((unsigned long long)x);
}
inline int __popc(unsigned x)
{
    return __builtin_popcount(x);
}
inline int __popcll(unsigned long long x)
{
    return __builtin_popcountll(x);
}
inline int __ffs(int x)
{
    return __builtin_ffs(x);
}
inline int __ffsll(unsigned long long x)
{
    return __builtin_ffsll((long long)x);
}

namespace emu {
inline int lane_id()
{
    return (int)(t_threadIdx.x % kWave);
}
inline int wave_id()
{
    return (int)(t_threadIdx.x / kWave);
}
inline void wave_sync()
{
    g_block->wave_barrier[wave_id()].wait();
}
// All 64 lanes of the wave must call these convergently (the kernels are written that way).
inline unsigned long long wave_exchange(unsigned long long mine, int src_lane)
{
    auto& slots = g_block->xchg[wave_id()];
    slots[lane_id()] = mine;
    wave_sync();
    const unsigned long long v = slots[src_lane & (kWave - 1)];
    wave_sync();
    return v;
}
} // namespace emu

// ds_permute_b32: lane (byte_address / 4) receives this lane's value ("push"); the kernels send to distinct lanes.
inline int __builtin_amdgcn_ds_permute(int byte_address, int value)
{
    auto& slots = emu::g_block->xchg[emu::wave_id()];
    emu::wave_sync();
    slots[(byte_address >> 2) & (emu::kWave - 1)] = (unsigned long long)(unsigned)value;
    emu::wave_sync();
    const int v = (int)(unsigned)slots[emu::lane_id()];
    emu::wave_sync();
    return v;
}

template <typename T>
inline T __shfl(T v, int src_lane, int = 64)
{
    unsigned long long raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    raw = emu::wave_exchange(raw, src_lane);
    T out;
    std::memcpy(&out, &raw, sizeof(T));
    return out;
}
template <typename T>
inline T __shfl_up(T v, unsigned delta, int = 64)
{
    const int l = emu::lane_id();
    const int src = l - (int)delta;
    const T got = __shfl(v, src < 0 ? l : src);
    return src < 0 ? v : got;
}
template <typename T>
inline T __shfl_down(T v, unsigned delta, int = 64)
{
    const int l = emu::lane_id();
    const int src = l + (int)delta;
    const T got = __shfl(v, src > 63 ? l : src);
    return src > 63 ? v : got;
}
template <typename T>
inline T __shfl_xor(T v, int mask, int = 64)
{
    return __shfl(v, emu::lane_id() ^ mask);
}
inline unsigned long long __ballot(int pred)
{
    auto& slots = emu::g_block->xchg[emu::wave_id()];
    slots[emu::lane_id()] = pred ? 1ull : 0ull;
    emu::wave_sync();
    unsigned long long m = 0;
    for (int i = 0; i < emu::kWave; ++i)
        m |= slots[i] << i;
    emu::wave_sync();
    return m;
}
inline int __any(int pred)
{
    return __ballot(pred) != 0;
}
inline int __all(int pred)
{
    return __ballot(pred) == ~0ull;
}

template <typename T>
inline T atomicAdd(T* p, T v)
{
    return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
}
template <typename T>
inline T atomicOr(T* p, T v)
{
    return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST);
}
template <typename T>
inline T atomicMax(T* p, T v)
{
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST))
    {
    }
    return old;
}
template <typename T>
inline T atomicMin(T* p, T v)
{
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST))
    {
    }
    return old;
}
template <typename T>
inline T atomicExch(T* p, T v)
{
    return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST);
}
#define __HIP_MEMORY_SCOPE_AGENT 0
template <typename T>
inline T __hip_atomic_load(const T* p, int order, int)
{
    return __atomic_load_n(p, order);
}
template <typename T, typename V>
inline void __hip_atomic_store(T* p, V v, int order, int)
{
    __atomic_store_n(p, (T)v, order);
}
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

inline void __builtin_amdgcn_wave_barrier()
{
    emu::wave_sync();
}

struct uint4
{
    unsigned x, y, z, w;
};
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w)
{
    return uint4{x, y, z, w};
}

inline int __builtin_amdgcn_readlane(int v, int lane)
{
    return __shfl(v, lane);
}

#define JLS_EMULATED 1
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline int __mul24(int a, int b)
{
    return a * b;
}

inline unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned offset, unsigned width)
{
    return width == 0 ? 0u : (v >> (offset & 31u)) & (width >= 32 ? ~0u : ((1u << width) - 1u));
}
inline unsigned __builtin_amdgcn_readfirstlane(unsigned v)
{
    return __shfl(v, 0);
}
