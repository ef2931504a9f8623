// scan_model.h -- JPEG-LS sample arithmetic and context model as device inlines (ISO/IEC 14495-1 annex A as the
// reference implements it).  Shared by the serial scan kernels and the parallel lossless pipeline.  All int32, no floats.
#pragma once
#include <hip/hip_runtime.h>

#include "scan_types.h"

namespace jls {

#define JLS_DEV __device__ __forceinline__

// The wave-uniform kernels rely on the 64 lanes of a wavefront executing in lockstep (all lanes read a shared LDS word
// before any lane of the same wavefront overwrites it).  On gfx950 that is how a wavefront executes; this marker only
// stops the compiler from reordering across it (no instruction is emitted).  The CPU test harness maps it to a real
// barrier between the lane threads.
#define JLS_LOCKSTEP() __builtin_amdgcn_wave_barrier()
// Behind a sequence of stores that every lane of a group issues identically (the later of two stores to one address
// must stay): lockstep execution orders them without the compiler's help, so nothing is needed on the GPU, not even a
// scheduling boundary; the lane threads of the CPU test harness have to meet before any of them reads the result.
#ifdef JLS_EMULATED
#define JLS_LOCKSTEP_STORES() __builtin_amdgcn_wave_barrier()
#else
#define JLS_LOCKSTEP_STORES() ((void)0)
#endif

// Path counters of the CPU test harness (tools/decode_path_profile.py: how often a wavefront takes each path of a kernel,
// to be priced with the instruction counts of the compiled code).  Nothing in the product build.
#ifdef JLS_PATH_PROFILE
extern "C" unsigned long long jls_path_counts[32];
#define JLS_PATH(n) do { if (threadIdx.x == 0) ++jls_path_counts[n]; } while (0)
#define JLS_PATH_ADD(n, v) do { if (threadIdx.x == 0) jls_path_counts[n] += (unsigned long long)(v); } while (0)
#else
#define JLS_PATH(n) ((void)0)
#define JLS_PATH_ADD(n, v) ((void)0)
#endif

// Pointers that are loaded from a descriptor in memory are "generic" to the compiler, which then emits flat_* memory
// instructions; those tick both wait counters and force conservative s_waitcnt 0.  Declaring them global lets the
// compiler use global_* instructions and count outstanding loads exactly.  (Empty in the CPU test harness.)
#ifndef JLS_GLOBAL_AS
#define JLS_GLOBAL_AS __attribute__((address_space(1)))
#endif

// Launch-time sized LDS (Guideline 17 of the CDNA guide: everything carved from one 16-byte aligned dynamic region).
// The CPU test harness pre-defines this to point at its per-workgroup buffer.
#ifndef JLS_DYNAMIC_LDS
#define JLS_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

// Value of lane 0, as a scalar: tells the compiler a value read from LDS / memory is wave-uniform.
JLS_DEV uint32_t uniform(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// Value held by lane `l` (wave-uniform l), as a scalar.
JLS_DEV uint32_t from_lane(uint32_t v, int l)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}

// Median of three signed integers in one vector instruction (the compiler only forms v_med3_i32 for constant bounds).
// med3s takes its last operand from a scalar register.
#ifndef JLS_EMULATED
JLS_DEV int med3(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
JLS_DEV int med3s(int a, int b, int c_scalar)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_scalar));
    return r;
}
#else
JLS_DEV int med3(int a, int b, int c)
{
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}
JLS_DEV int med3s(int a, int b, int c)
{
    return med3(a, b, c);
}
#endif

// Zero in a vector register that the compiler cannot see through.  OR-ing it into a wave-uniform value keeps that value
// and everything computed from it on the vector ALU (the compiler would otherwise move uniform work to the scalar unit).
#ifndef JLS_EMULATED
JLS_DEV int vector_zero()
{
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
#else
JLS_DEV int vector_zero()
{
    return 0;
}
#endif

// Bit tricks with the exact semantics of the gfx950 instructions (the C builtins are undefined for 0).
#ifndef JLS_EMULATED
JLS_DEV uint32_t bit_reverse(uint32_t v)
{
    return __builtin_bitreverse32(v);
}
JLS_DEV uint32_t lowest_one(uint32_t v) // v_ffbl_b32: index of the lowest set bit; 0xFFFFFFFF for v = 0
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
JLS_DEV uint32_t leading_zeros(uint32_t v) // v_ffbh_u32: number of leading zero bits; 0xFFFFFFFF for v = 0
{
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
#else
JLS_DEV uint32_t bit_reverse(uint32_t v)
{
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i)
        r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
JLS_DEV uint32_t lowest_one(uint32_t v)
{
    return v == 0 ? 0xFFFFFFFFu : (uint32_t)__builtin_ctz(v);
}
JLS_DEV uint32_t leading_zeros(uint32_t v)
{
    return v == 0 ? 0xFFFFFFFFu : (uint32_t)__builtin_clz(v);
}
#endif

// Lane predicates as 64-bit masks in scalar registers: combining them is scalar arithmetic, testing them a scalar branch
// (the compiler keeps `bool`s that way too, but only the ballot of a single comparison stays free of a detour through a
// vector register).  mad24 = a * b + c on the low 24 bits in one instruction; pack_bytes = s1.byte0 | s0.byte0 << 8; a
// `rare` marker keeps the compiler from turning a seldom-taken block into unconditional selects.
typedef unsigned long long LaneMask;
#ifndef JLS_EMULATED
JLS_DEV LaneMask lanes_where(bool p)
{
    return __builtin_amdgcn_ballot_w64(p);
}
JLS_DEV bool lane_of(LaneMask m)
{
    return __builtin_amdgcn_inverse_ballot_w64(m);
}
JLS_DEV uint32_t value_of_lowest_lane(LaneMask m, uint32_t v) // v of the first lane of m (m != 0): s_ff1, v_readlane
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_ctzll(m));
}
JLS_DEV int mad24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
JLS_DEV uint32_t pack_bytes(uint32_t s0, uint32_t s1)
{
    return __builtin_amdgcn_perm(s0, s1, 0x0C0C0400u);
}
JLS_DEV uint32_t bit_field(uint32_t v, uint32_t offset, uint32_t width) // (v >> offset) & (2^width - 1), constants
{
    uint32_t r;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "n"(offset), "n"(width));
    return r;
}
JLS_DEV int sign_extend(int v, int bits) // the low `bits` bits of v as a signed number (v_bfe_i32, one instruction)
{
    return __builtin_amdgcn_sbfe(v, 0, bits);
}
JLS_DEV uint32_t abs_difference(uint32_t a, uint32_t b) // |a - b| of unsigned values in one instruction
{
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#define JLS_RARE_BLOCK() asm volatile("; rarely taken")
// Nothing is scheduled across this point (used to keep the arithmetic that covers an LDS latency ahead of the first use
// of the loaded values; the machine scheduler otherwise pulls single users up next to their loads).
#define JLS_SCHEDULE_FENCE() __builtin_amdgcn_sched_barrier(0)
// One-hot loop counter: shifted right once per step, cleared when any lane of `busy` is missing from `ok` (then the
// caller's `while (ticker != 0)` ends the loop).  Three scalar instructions; the compiler's own rendering of the same
// condition takes nine, because it has to merge the loop's exits.
JLS_DEV uint32_t tick(uint32_t ticker, LaneMask busy, LaneMask ok)
{
    LaneMask missing;
    asm volatile("s_andn2_b64 %1, %2, %3\n\ts_cselect_b32 %0, 0, %0\n\ts_lshr_b32 %0, %0, 1"
                 : "+s"(ticker), "=&s"(missing)
                 : "s"(busy), "s"(ok)
                 : "scc");
    return ticker;
}
JLS_DEV uint64_t tick(uint64_t ticker, LaneMask busy, LaneMask ok) // up to 64 steps
{
    LaneMask missing;
    asm volatile("s_andn2_b64 %1, %2, %3\n\ts_cselect_b64 %0, 0, %0\n\ts_lshr_b64 %0, %0, 1"
                 : "+s"(ticker), "=&s"(missing)
                 : "s"(busy), "s"(ok)
                 : "scc");
    return ticker;
}
// Hides how a per-lane value was computed (the compiler would otherwise fold a select into the predicates that use it).
JLS_DEV uint32_t opaque(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return v;
}
JLS_DEV uint32_t float_bits(uint32_t v) // bit pattern of (float)v; exact below 2^24
{
    return __float_as_uint((float)v);
}
// LDS by absolute address (the dynamic LDS region of a kernel without static LDS starts at 0; kernels that use these
// check lds_address(their region) == 0).  An index that is "difference + constant" then needs no base register: the
// constant goes into the instruction's offset field, and the hardware adds register and offset modulo 2^32, so a
// negative difference is fine (checked on gfx950: tools/microbench/latency.hip).
#define JLS_LDS_AS __attribute__((address_space(3)))
JLS_DEV uint32_t lds_address(const void* p)
{
    return (uint32_t)(uintptr_t)(const JLS_LDS_AS void*)p;
}
template <typename T>
JLS_DEV T lds_load(uint32_t address)
{
    return *(const JLS_LDS_AS T*)(uintptr_t)address;
}
template <typename T>
JLS_DEV void lds_store(uint32_t address, T v)
{
    *(JLS_LDS_AS T*)(uintptr_t)address = v;
}
// A 64-bit word other workgroups poll / publish (device scope, no ordering: the word carries everything).
JLS_DEV uint64_t load_relaxed(const uint64_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
JLS_DEV void store_relaxed(uint64_t* p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Bits [shift mod 32, shift mod 32 + 32) of the 64-bit value hi:lo (v_alignbit_b32 takes the low five bits of the shift).
JLS_DEV uint32_t funnel_shift(uint32_t hi, uint32_t lo, uint32_t shift)
{
    return __builtin_amdgcn_alignbit(hi, lo, shift);
}
#else
JLS_DEV LaneMask lanes_where(bool p)
{
    return __ballot(p);
}
JLS_DEV bool lane_of(LaneMask m)
{
    return ((m >> emu::lane_id()) & 1ull) != 0;
}
JLS_DEV uint32_t value_of_lowest_lane(LaneMask m, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_ctzll(m));
}
JLS_DEV int mad24(int a, int b, int c)
{
    return a * b + c;
}
JLS_DEV uint32_t pack_bytes(uint32_t s0, uint32_t s1)
{
    return (s1 & 0xFFu) | ((s0 & 0xFFu) << 8);
}
JLS_DEV uint32_t bit_field(uint32_t v, uint32_t offset, uint32_t width)
{
    return (v >> offset) & ((1u << width) - 1u);
}
JLS_DEV int sign_extend(int v, int bits)
{
    const int sh = 32 - bits;
    return (int)((uint32_t)v << sh) >> sh;
}
JLS_DEV uint32_t abs_difference(uint32_t a, uint32_t b)
{
    return a > b ? a - b : b - a;
}
#define JLS_RARE_BLOCK() ((void)0)
#define JLS_SCHEDULE_FENCE() ((void)0)
JLS_DEV uint32_t tick(uint32_t ticker, LaneMask busy, LaneMask ok)
{
    return (busy & ~ok) != 0 ? 0u : ticker >> 1;
}
JLS_DEV uint64_t tick(uint64_t ticker, LaneMask busy, LaneMask ok)
{
    return (busy & ~ok) != 0 ? 0ull : ticker >> 1;
}
JLS_DEV uint32_t opaque(uint32_t v)
{
    return v;
}
JLS_DEV uint32_t lds_address(const void* p)
{
    return (uint32_t)(static_cast<const unsigned char*>(p) - emu::g_block->dyn_shared);
}
template <typename T>
JLS_DEV T lds_load(uint32_t address)
{
    T v;
    std::memcpy(&v, emu::g_block->dyn_shared + address, sizeof(T));
    return v;
}
template <typename T>
JLS_DEV void lds_store(uint32_t address, T v)
{
    std::memcpy(emu::g_block->dyn_shared + address, &v, sizeof(T));
}
JLS_DEV uint32_t funnel_shift(uint32_t hi, uint32_t lo, uint32_t shift)
{
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (shift & 31u));
}
JLS_DEV uint64_t load_relaxed(const uint64_t* p)
{
    return __atomic_load_n(p, __ATOMIC_RELAXED);
}
JLS_DEV void store_relaxed(uint64_t* p, uint64_t v)
{
    __atomic_store_n(p, v, __ATOMIC_RELAXED);
}
JLS_DEV uint32_t float_bits(uint32_t v)
{
    const float f = (float)v;
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
#endif

// J[] run-length order table (reference src/scan_codec.hpp:18-19), 4 bits per entry packed into two 64-bit words.
JLS_DEV int run_j(int run_index)
{
    // {0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3} and {4,4,5,5,6,6,7,7,8,9,10,11,12,13,14,15}
    const unsigned long long lo = 0x3333222211110000ull;
    const unsigned long long hi = 0xFEDCBA9877665544ull;
    const unsigned long long w = run_index < 16 ? lo : hi;
    return (int)((w >> ((run_index & 15) * 4)) & 15u);
}

JLS_DEV int log2_ceiling(int n)
{
    int x = 0;
    while (n > (1 << x))
        ++x;
    return x;
}

// reference src/make_scan_codec.cpp:40-156: RANGE/qbpp/LIMIT always derive from 2^bpp - 1.
JLS_DEV Traits make_traits(const ScanDesc& d)
{
    Traits t;
    t.bpp = d.bits_per_sample;
    t.near = d.near_lossless;
    t.maxval = (1 << d.bits_per_sample) - 1;
    t.range = (t.maxval + 2 * t.near) / (2 * t.near + 1) + 1;
    t.qbpp = log2_ceiling(t.range);
    t.limit = 2 * (t.bpp + (t.bpp > 8 ? t.bpp : 8));
    t.t1 = d.t1;
    t.t2 = d.t2;
    t.t3 = d.t3;
    t.reset = d.reset;
    return t;
}

JLS_DEV int initial_a(const Traits& t) // reference src/jpegls_algorithm.hpp:56-60
{
    const int a = (t.range + 32) / 64;
    return a > 2 ? a : 2;
}

// Gradient quantisation, branch-free form of reference src/jpegls_algorithm.hpp:173-194 (note the asymmetric
// <= / < on the two sides).
JLS_DEV int quantize(const Traits& t, int d)
{
    const int pos = (d > t.near) + (d >= t.t1) + (d >= t.t2) + (d >= t.t3);
    const int neg = (d < -t.near) + (d <= -t.t1) + (d <= -t.t2) + (d <= -t.t3);
    return pos - neg;
}

JLS_DEV int context_id(const Traits& t, int ra, int rb, int rc, int rd) // src/jpegls_algorithm.hpp:165-168
{
    return (quantize(t, rd - rb) * 9 + quantize(t, rb - rc)) * 9 + quantize(t, rc - ra);
}

JLS_DEV int med_predict(int ra, int rb, int rc) // src/jpegls_algorithm.hpp:143-161
{
    const int lo = ra < rb ? ra : rb;
    const int hi = ra < rb ? rb : ra;
    const int grad = ra + rb - rc;
    return rc >= hi ? lo : (rc <= lo ? hi : grad);
}

JLS_DEV int clamp_sample(const Traits& t, int v) // correct_prediction, src/default_traits.hpp:110-116
{
    return v < 0 ? 0 : (v > t.maxval ? t.maxval : v);
}

JLS_DEV int map_error(int e) // src/jpegls_algorithm.hpp:67-73
{
    return (e >> 30) ^ (2 * e);
}

JLS_DEV int unmap_error(int m) // src/jpegls_algorithm.hpp:80-86
{
    return (m >> 1) ^ -(m & 1);
}

// Errval quantisation + reduction modulo RANGE, src/default_traits.hpp:77-80,123-139,157-163.
JLS_DEV int error_value(const Traits& t, int e)
{
    if (t.near != 0)
    {
        const int d = 2 * t.near + 1;
        e = e > 0 ? (e + t.near) / d : -((t.near - e) / d);
    }
    if (e < 0)
        e += t.range;
    if (e >= (t.range + 1) / 2)
        e -= t.range;
    return e;
}

// Rx, src/default_traits.hpp:83-87,172-184.
JLS_DEV int reconstruct(const Traits& t, int predicted, int e)
{
    const int step = 2 * t.near + 1;
    int v = predicted + e * step;
    if (v < -t.near)
        v += t.range * step;
    else if (v > t.maxval + t.near)
        v -= t.range * step;
    return clamp_sample(t, v);
}

JLS_DEV bool is_near(const Traits& t, int a, int b)
{
    const int d = a - b;
    return (d < 0 ? -d : d) <= t.near;
}

// ---- regular-mode context {A,B,C,N}: src/regular_mode_context.hpp ----------------------------------------------
struct RegCtx
{
    int a, b, c, n;
};

// k = min{k : N<<k >= A}; returns 16 when the reference raises invalid_data (src/regular_mode_context.hpp:99-136).
JLS_DEV int regular_k(const RegCtx& x)
{
    int k = __clz(x.n) - __clz(x.a); // both > 0
    k = k < 0 ? 0 : k;
    k += ((x.n << k) < x.a);
    return k > 16 ? 16 : k;
}

JLS_DEV int error_correction(const RegCtx& x, int k_or_near) // src/regular_mode_context.hpp:36-42
{
    return k_or_near != 0 ? 0 : ((2 * x.b + x.n - 1) >> 31);
}

// A.12/A.13 update; returns false when the reference raises invalid_data (src/regular_mode_context.hpp:45-93).
JLS_DEV bool regular_update(RegCtx& x, int e, int near, int reset)
{
    x.a += e < 0 ? -e : e;
    x.b += e * (2 * near + 1);
    const int ab = x.b < 0 ? -x.b : x.b;
    if (x.a >= (1 << 24) || ab >= (1 << 24))
        return false;
    if (x.n == reset)
    {
        x.a >>= 1;
        x.b >>= 1;
        x.n >>= 1;
    }
    ++x.n;
    if (x.b + x.n <= 0)
    {
        x.b += x.n;
        if (x.b <= -x.n)
            x.b = -x.n + 1;
        if (x.c > -128)
            --x.c;
    }
    else if (x.b > 0)
    {
        x.b -= x.n;
        if (x.b > 0)
            x.b = 0;
        if (x.c < 127)
            ++x.c;
    }
    return true;
}

// ---- run-interruption context {RItype,A,N,Nn}: src/run_mode_context.hpp ----------------------------------------
struct RunCtx
{
    int ritype, a, n, nn;
};

JLS_DEV int run_k(const RunCtx& x) // src/run_mode_context.hpp:34-62 (k > 32 -> 33 = invalid for the decoder)
{
    const long long temp = (long long)x.a + (long long)(x.n >> 1) * x.ritype;
    long long n_test = x.n;
    int k = 0;
    while (n_test < temp && k <= 33)
    {
        n_test <<= 1;
        ++k;
    }
    return k;
}

// The same k for the states an ENCODER can be in (A, N >= 1 and the sum below 2^31: A grows by at most the largest mapped
// error per event and is halved every RESET events), without the loop: N 2^k >= temp first holds for k = floor(log2 temp) -
// floor(log2 N) or the k after it.  The walkers of the tile pipeline's run chain are one dependent chain per lane.
JLS_DEV int run_k_of_encoder(const RunCtx& x)
{
    const uint32_t temp = (uint32_t)x.a + (uint32_t)(x.n >> 1) * (uint32_t)x.ritype, n = (uint32_t)x.n;
    if (temp <= n)
        return 0;
    const int k = __builtin_clz(n) - __builtin_clz(temp);
    return (n << k) < temp ? k + 1 : k;
}

JLS_DEV int run_map(const RunCtx& x, int e, int k) // src/run_mode_context.hpp:103-115
{
    return (k == 0 && e > 0 && 2 * x.nn < x.n) || (e < 0 && 2 * x.nn >= x.n) || (e < 0 && k != 0);
}

JLS_DEV void run_update(RunCtx& x, int e, int em, int reset) // src/run_mode_context.hpp:65-83
{
    if (e < 0)
        ++x.nn;
    x.a += (em + 1 - x.ritype) >> 1;
    if (x.n == reset)
    {
        x.a >>= 1;
        x.n >>= 1;
        x.nn >>= 1;
    }
    ++x.n;
}

JLS_DEV int run_error_value(const RunCtx& x, int temp, int k) // src/run_mode_context.hpp:86-99
{
    const int map = temp & 1;
    const int ea = (temp + map) / 2;
    const int neg = (k != 0 || (2 * x.nn >= x.n)) ? 1 : 0;
    return neg == map ? -ea : ea;
}

// ---- colour transforms, src/color_transform.hpp:26-117 (modulo 2^(8*sizeof(sample))) -----------------------------
JLS_DEV void hp_forward(int xform, bool wide, int r, int g, int b, unsigned out[3])
{
    const int range = wide ? 65536 : 256;
    const int bias = range / 2;
    const unsigned m = (unsigned)range - 1u;
    if (xform == 1)
    {
        out[0] = (unsigned)(r - g + bias) & m;
        out[1] = (unsigned)g & m;
        out[2] = (unsigned)(b - g + bias) & m;
    }
    else if (xform == 2)
    {
        out[0] = (unsigned)(r - g + bias) & m;
        out[1] = (unsigned)g & m;
        out[2] = (unsigned)(b - ((r + g) / 2) + bias) & m;
    }
    else
    {
        const int v2 = (int)((unsigned)(b - g + bias) & m);
        const int v3 = (int)((unsigned)(r - g + bias) & m);
        out[0] = (unsigned)(g + ((v2 + v3) >> 2) - range / 4) & m;
        out[1] = (unsigned)v2;
        out[2] = (unsigned)v3;
    }
}

JLS_DEV void hp_inverse(int xform, bool wide, int v1, int v2, int v3, unsigned out[3])
{
    const int range = wide ? 65536 : 256;
    const int bias = range / 2;
    const unsigned m = (unsigned)range - 1u;
    if (xform == 1)
    {
        out[0] = (unsigned)(v1 + v2 - bias) & m;
        out[1] = (unsigned)v2 & m;
        out[2] = (unsigned)(v3 + v2 - bias) & m;
    }
    else if (xform == 2)
    {
        const int r = (int)((unsigned)(v1 + v2 - bias) & m);
        out[0] = (unsigned)r;
        out[1] = (unsigned)v2 & m;
        out[2] = (unsigned)(v3 + ((r + (int)((unsigned)v2 & m)) >> 1) - bias) & m;
    }
    else
    {
        const int g = v1 - ((v3 + v2) >> 2) + range / 4;
        out[0] = (unsigned)(v3 + g - bias) & m;
        out[1] = (unsigned)g & m;
        out[2] = (unsigned)(v2 + g - bias) & m;
    }
}

} // namespace jls
