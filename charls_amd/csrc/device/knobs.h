// knobs.h -- the engine's test and measurement knobs, in ONE table.
//
// Until round 4 every knob was a std::getenv() on the product's hot path: a stray CHARLS_AMD_* variable in a production
// environment changed an encoder's speed by 10 x, silently, on every call.  Now the environment is read ONCE -- the first
// time any knob is looked at -- into this table; after that only charls_amd_debug_set_knob() (include/charls_amd.h, additive;
// what the GPU tests and the measurement tools use) changes a value.  A knob is either unset (the engine's own rule applies)
// or an integer.  Reads are one relaxed atomic load.
//
// The CPU harness (tests/emu: kernel sources compiled for the host, TEST-ONLY) defines JLS_KNOBS_LIVE_ENV: there a knob
// that was not set through the table is looked up in the environment on every read, which is what its tests rely on.
#pragma once
#include <atomic>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace jls::knobs {

enum Knob : int
{
    kDecodeGroup = 0,       // CHARLS_AMD_DECODE_GROUP: lanes per scan of the group decoders (0 = one scan per wavefront)
    kExactDecoder,          // CHARLS_AMD_EXACT_DECODER: no speed path
    kSequentialIntervals,   // CHARLS_AMD_SEQUENTIAL_INTERVALS: restart intervals decoded one after the other
    kBlockStuffing,         // CHARLS_AMD_BLOCK_STUFFING: 0 = never the block-parallel form of stage E
    kSpecStuffing,          // CHARLS_AMD_SPEC_STUFFING: 0 = never the speculative form of stage E
    kJobEvents,             // CHARLS_AMD_JOB_EVENTS: events per job of a regular chain
    kWarmEvents,            // CHARLS_AMD_WARM_EVENTS: events of warm-up before a job
    kRunJobEvents,          // CHARLS_AMD_RUN_JOB_EVENTS
    kRunWarmEvents,         // CHARLS_AMD_RUN_WARM_EVENTS
    kRareWarmEvents,        // CHARLS_AMD_RARE_WARM_EVENTS
    kTileSamples,           // CHARLS_AMD_TILE_SAMPLES: samples per tile (lowers the cap)
    kPixelMode,             // CHARLS_AMD_PIXEL_MODE: every scan through pixel mode
    kSpecChunk,             // CHARLS_AMD_SPEC_CHUNK: chunk of the speculative stuffing, bytes
    kSpecWarm,              // CHARLS_AMD_SPEC_WARM: its warm-up, bytes
    kBatchRounds,           // CHARLS_AMD_BATCH_ROUNDS: planar batches scan by scan
    kCoalesce,              // CHARLS_AMD_COALESCE: 0 = concurrent calls of the host-pointer ABI are never merged into one launch
    kCoalesceWaitUs,        // CHARLS_AMD_COALESCE_WAIT_US: longest time a call waits for calls that announced themselves
    kDecodeWavesPerCu,      // CHARLS_AMD_DECODE_WAVES_PER_CU: wavefronts per CU the decoder's packing rule aims at
    kDecodeWorkgroupWaves,  // CHARLS_AMD_DECODE_WORKGROUP_WAVES: wavefronts per workgroup of the group decoder (1, 4, 8; with DECODE_GROUP)
    kTrace,                 // CHARLS_AMD_TRACE: one line on stderr per coding call of the host-pointer ABI (where its time went)
    kIdleReleaseMs,         // CHARLS_AMD_IDLE_RELEASE_MS: what the host-pointer ABI keeps between calls is freed after this long without a call (0 = kept)
    kNearDecodePixels,      // CHARLS_AMD_NEAR_DECODE_PIXELS: 1 = near-lossless single-component / line-interleaved scans on the pixel kernels (until round 6)
    kCount
};

constexpr long long kUnset = LLONG_MIN;

inline const char* name_of(int k)
{
    static const char* const names[kCount] = {"DECODE_GROUP", "EXACT_DECODER", "SEQUENTIAL_INTERVALS", "BLOCK_STUFFING", "SPEC_STUFFING",
                                              "JOB_EVENTS", "WARM_EVENTS", "RUN_JOB_EVENTS", "RUN_WARM_EVENTS", "RARE_WARM_EVENTS",
                                              "TILE_SAMPLES", "PIXEL_MODE", "SPEC_CHUNK", "SPEC_WARM", "BATCH_ROUNDS", "COALESCE",
                                              "COALESCE_WAIT_US", "DECODE_WAVES_PER_CU", "DECODE_WORKGROUP_WAVES", "TRACE", "IDLE_RELEASE_MS", "NEAR_DECODE_PIXELS"};
    return k >= 0 && k < kCount ? names[k] : nullptr;
}

struct Table
{
    std::atomic<long long> value[kCount];
};

inline long long from_environment(int k)
{
    char name[64] = "CHARLS_AMD_";
    std::strncat(name, name_of(k), sizeof name - std::strlen(name) - 1);
    const char* env = std::getenv(name);
    if (env == nullptr)
        return kUnset;
    // (a variable that is set but empty counts as 1: CHARLS_AMD_EXACT_DECODER= used to be tested for presence only)
    return *env == '\0' ? 1 : std::atoll(env);
}

inline Table& table()
{
    static Table* t = [] {
        auto* fresh = new Table; // never destroyed: knobs are read by threads that may outlive static destruction
        for (int k = 0; k < kCount; ++k)
#ifdef JLS_KNOBS_LIVE_ENV
            fresh->value[k].store(kUnset, std::memory_order_relaxed);
#else
            fresh->value[k].store(from_environment(k), std::memory_order_relaxed);
#endif
        return fresh;
    }();
    return *t;
}

// The knob's value, or kUnset.
inline long long get(Knob k)
{
    const long long v = table().value[k].load(std::memory_order_relaxed);
#ifdef JLS_KNOBS_LIVE_ENV
    if (v == kUnset)
        return from_environment(k);
#endif
    return v;
}

inline bool is_set(Knob k)
{
    return get(k) != kUnset;
}

// value when set, otherwise `otherwise`
inline long long get_or(Knob k, long long otherwise)
{
    const long long v = get(k);
    return v == kUnset ? otherwise : v;
}

// By name without the CHARLS_AMD_ prefix; value kUnset clears.  False for a name that is not a knob.
inline bool set(const char* name, long long value)
{
    if (name == nullptr)
        return false;
    if (std::strncmp(name, "CHARLS_AMD_", 11) == 0)
        name += 11;
    for (int k = 0; k < kCount; ++k)
        if (std::strcmp(name, name_of(k)) == 0)
        {
            table().value[k].store(value, std::memory_order_relaxed);
            return true;
        }
    return false;
}

} // namespace jls::knobs
