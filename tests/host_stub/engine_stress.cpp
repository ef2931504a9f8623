// TEST-ONLY: charls_amd/csrc/host/scan_engine.cpp (resource pool, coalescer, launch streams, housekeeping thread) built for the
// host against the stand-in runtime of this directory and driven by many threads -- a handle per call, as the reference's callers
// do it (cli/benchmark.cpp).  Meant to run under ThreadSanitizer and AddressSanitizer + UBSan (tests/test_host_engine_cpu.py);
// prints "engine ok" when every call got the result of ITS scan and what the calls left behind was given back.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "device/knobs.h"
#include "host/scan_engine.h"

using namespace jls;

static ScanSpec spec_of(uint32_t width, uint32_t height)
{
    ScanSpec s{};
    s.width = width;
    s.height = height;
    s.components = 1;
    s.interleave_mode = 0;
    s.bits_per_sample = 8;
    s.pc = charls_jpegls_pc_parameters{255, 3, 7, 21, 64};
    return s;
}

int main()
{
    knobs::set("IDLE_RELEASE_MS", 150);
    constexpr int kThreads = 24, kLoops = 12;
    std::atomic<int> wrong{0}, raised{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < kThreads; ++t)
        pool.emplace_back([&, t] {
            for (int k = 0; k < kLoops; ++k)
            {
                const uint32_t width = (t + k) % 3 == 0 ? 96 : 64, height = 8 + (t % 4); // (mixed geometries: several batches)
                const ScanSpec spec = spec_of(width, height);
                try
                {
                    // decode: the "stream" is 40 bytes that only this call has
                    {
                        ScanEngine engine;
                        const CallScope call(engine);
                        std::vector<uint8_t> stream(40, static_cast<uint8_t>(t * 16 + k));
                        uint8_t tag = 0;
                        for (uint8_t b : stream)
                            tag = static_cast<uint8_t>(tag * 31 + b);
                        engine.expect_call(true); // (set_source_buffer)
                        engine.upload_stream(stream.data(), stream.size(), frame_hint(width, height, 8));
                        std::vector<uint8_t> out(static_cast<size_t>(width) * height, 0);
                        const size_t used = engine.decode_scan(spec, 0, out.data(), width);
                        if (used != stream.size())
                            ++wrong;
                        for (uint8_t v : out)
                            if (v != tag)
                            {
                                ++wrong;
                                break;
                            }
                    }
                    // encode: the pixels are this call's
                    {
                        ScanEngine engine;
                        const CallScope call(engine);
                        std::vector<uint8_t> pixels(static_cast<size_t>(width) * height, static_cast<uint8_t>(t * 7 + k * 3 + 1));
                        uint8_t tag = 0;
                        for (uint8_t b : pixels)
                            tag = static_cast<uint8_t>(tag * 131 + b);
                        engine.expect_call(false, frame_hint(width, height, 8)); // (set_frame_info)
                        engine.upload_pixels(pixels.data(), pixels.size(), frame_hint(width, height, 8));
                        uint8_t coded[64] = {};
                        const size_t n = engine.encode_scan(spec, 0, width, coded, sizeof coded);
                        if (n != 16 || coded[0] != tag || coded[15] != tag)
                            ++wrong;
                    }
                    if ((t + k) % 5 == 0)
                    { // a handle that announces a call and goes away without making it
                        ScanEngine engine;
                        engine.expect_call(true);
                    }
                }
                catch (const error&)
                {
                    ++raised;
                }
                if (k % 4 == 3)
                    std::this_thread::sleep_for(std::chrono::milliseconds(3));
            }
        });
    for (auto& th : pool)
        th.join();
    uint64_t stats[5];
    coalescer_stats(stats);
    std::printf("calls %llu launches %llu merged %llu largest %llu split %llu wrong %d raised %d idle pool %llu bytes\n", (unsigned long long)stats[0],
                (unsigned long long)stats[1], (unsigned long long)stats[2], (unsigned long long)stats[3], (unsigned long long)stats[4], wrong.load(),
                raised.load(), (unsigned long long)idle_engine_resource_bytes());
    bool ok = wrong == 0 && raised == 0 && stats[0] == 2ull * kThreads * kLoops && stats[1] < stats[0];
    // what the calls left behind decays (150 ms here)
    const bool kept = idle_engine_resource_bytes() != 0 || dev::shared_work_area_bytes() != 0;
    for (int i = 0; i < 100 && (idle_engine_resource_bytes() != 0 || dev::shared_work_area_bytes() != 0); ++i)
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    std::printf("kept after the burst: %s; after the quiet time: idle pool %llu bytes, shared areas %llu bytes, releases %llu\n", kept ? "yes" : "no",
                (unsigned long long)idle_engine_resource_bytes(), (unsigned long long)dev::shared_work_area_bytes(), (unsigned long long)idle_releases());
    ok = ok && kept && idle_engine_resource_bytes() == 0 && dev::shared_work_area_bytes() == 0 && idle_releases() >= 1;
    // and the engine simply works again afterwards
    {
        ScanEngine engine;
        const CallScope call(engine);
        std::vector<uint8_t> stream(40, 9);
        engine.upload_stream(stream.data(), stream.size());
        std::vector<uint8_t> out(64 * 8, 0);
        ok = ok && engine.decode_scan(spec_of(64, 8), 0, out.data(), 64) == 40;
    }
    std::printf(ok ? "engine ok\n" : "engine FAILED\n");
    return ok ? 0 : 1;
}
