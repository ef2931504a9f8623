// coalescer_test.cpp -- drives charls_amd/csrc/host/coalescer.h with a fake launch (no HIP, no GPU).  Built and run by
// tests/test_coalescer_cpu.py.  Every check prints a line; the last line is "coalescer ok".
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "host/coalescer.h"

using namespace jls;
using Clock = std::chrono::steady_clock;

static int g_failures = 0;
#define CHECK(cond, what)                                                  \
    do                                                                     \
    {                                                                      \
        const bool ok_ = (cond);                                           \
        std::printf("%s: %s\n", ok_ ? "ok" : "FAILED", what);              \
        if (!ok_)                                                          \
            ++g_failures;                                                  \
    } while (0)

static ScanDesc desc_of(uint32_t width, uint32_t tag)
{
    ScanDesc d{};
    d.width = width;
    d.height = 8;
    d.components = 1;
    d.bits_per_sample = 8;
    d.stream_capacity = tag; // (the fake launch echoes it: every caller must get ITS result back)
    return d;
}

static double ms_since(Clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
}

int main()
{
    // 1. a caller that is alone launches at once, whatever the wait
    {
        Coalescer c;
        int launches = 0;
        const ScanDesc d = desc_of(64, 7);
        ScanResult r{};
        const auto t0 = Clock::now();
        const Coalescer::Ticket mine = c.announce(1);
        c.submit(1, merge_key_of(d), &d, 1, &r, mine, Coalescer::Policy{500000, 1024, 2, 125000, 0}, [&](const ScanDesc* all, uint32_t n, ScanResult* out) {
            ++launches;
            for (uint32_t i = 0; i < n; ++i)
                out[i] = ScanResult{0, 0, all[i].stream_capacity * 3};
        });
        CHECK(launches == 1 && r.bytes == 21 && ms_since(t0) < 100, "a lone call launches at once and gets its result");
        CHECK(c.idle(1, 500000), "the lane is idle afterwards");
    }
    // 2. N announced calls end up in ONE launch, every caller gets its own result
    {
        Coalescer c;
        constexpr int kThreads = 48;
        std::atomic<int> launches{0}, largest{0}, wrong{0};
        std::vector<Coalescer::Ticket> tickets;
        for (int i = 0; i < kThreads; ++i)
            tickets.push_back(c.announce(3));
        std::vector<std::thread> threads;
        for (int i = 0; i < kThreads; ++i)
            threads.emplace_back([&, i] {
                std::this_thread::sleep_for(std::chrono::microseconds(200 * (i % 7))); // (uploads of different length)
                ScanDesc d[2] = {desc_of(512, 1000 + i), desc_of(512, 2000 + i)};
                ScanResult r[2]{};
                c.submit(3, merge_key_of(d[0]), d, 2, r, tickets[i], Coalescer::Policy{2000000, 1024, 2, 500000, 0}, [&](const ScanDesc* all, uint32_t n, ScanResult* out) {
                    ++launches;
                    largest = std::max<int>(largest, (int)n);
                    for (uint32_t k = 0; k < n; ++k)
                        out[k] = ScanResult{0, 0, all[k].stream_capacity + 5};
                });
                if (r[0].bytes != 1005u + i || r[1].bytes != 2005u + i)
                    ++wrong;
            });
        for (auto& t : threads)
            t.join();
        CHECK(launches == 1 && largest == 2 * kThreads && wrong == 0, "48 announced calls of two scans share one launch");
        const Coalescer::Stats s = c.stats();
        CHECK(s.calls == kThreads && s.launches == 1 && s.merged == kThreads && s.largest == 2 * kThreads, "the counters say so");
    }
    // 3. different keys never share a launch; equal keys on different lanes neither
    {
        Coalescer c;
        std::atomic<int> launches{0}, mixed{0};
        std::vector<std::thread> threads;
        std::vector<Coalescer::Ticket> tickets;
        for (int i = 0; i < 24; ++i)
            tickets.push_back(c.announce(i % 2 ? 5 : 7));
        for (int i = 0; i < 24; ++i)
            threads.emplace_back([&, i] {
                const ScanDesc d = desc_of(i % 3 == 0 ? 100 : 200, (uint32_t)i);
                ScanResult r{};
                c.submit(i % 2 ? 5 : 7, merge_key_of(d), &d, 1, &r, tickets[i], Coalescer::Policy{300000, 1024, 2, 75000, 0}, [&](const ScanDesc* all, uint32_t n, ScanResult* out) {
                    ++launches;
                    for (uint32_t k = 0; k < n; ++k)
                    {
                        if (all[k].width != all[0].width || (all[k].stream_capacity % 2) != (all[0].stream_capacity % 2))
                            ++mixed;
                        out[k] = ScanResult{};
                    }
                });
            });
        for (auto& t : threads)
            t.join();
        CHECK(mixed == 0 && launches >= 4, "keys and lanes are kept apart");
    }
    // 4. an exclusive lane runs one batch at a time, and whoever arrives while one runs joins the NEXT one (group commit)
    {
        Coalescer c;
        std::atomic<int> running{0}, overlap{0}, launches{0};
        auto fake = [&](const ScanDesc*, uint32_t n, ScanResult* out) {
            if (++running > 1)
                ++overlap;
            ++launches;
            std::this_thread::sleep_for(std::chrono::milliseconds(60));
            for (uint32_t k = 0; k < n; ++k)
                out[k] = ScanResult{};
            --running;
        };
        std::vector<std::thread> threads;
        for (int i = 0; i < 12; ++i)
            threads.emplace_back([&, i] {
                if (i > 0)
                    std::this_thread::sleep_for(std::chrono::milliseconds(10 + i)); // the first one is running by then
                const ScanDesc d = desc_of(300, (uint32_t)i);
                ScanResult r{};
                c.submit(2, merge_key_of(d), &d, 1, &r, 0, Coalescer::Policy{0, 1024, 0, 0, 0}, fake);
            });
        for (auto& t : threads)
            t.join();
        CHECK(overlap == 0 && launches == 2, "exclusive: no two batches at once, 11 late calls in the second launch");
    }
    // 5. a launch that fails fails every call of its batch with its code
    {
        Coalescer c;
        std::atomic<int> codes{0};
        std::vector<std::thread> threads;
        std::vector<Coalescer::Ticket> tickets;
        for (int i = 0; i < 6; ++i)
            tickets.push_back(c.announce(9));
        for (int i = 0; i < 6; ++i)
            threads.emplace_back([&, i] {
                const ScanDesc d = desc_of(77, 0);
                ScanResult r{};
                try
                {
                    c.submit(9, merge_key_of(d), &d, 1, &r, tickets[i], Coalescer::Policy{500000, 1024, 2, 125000, 0},
                             [&](const ScanDesc*, uint32_t, ScanResult*) { raise(CHARLS_JPEGLS_ERRC_NOT_ENOUGH_MEMORY); });
                }
                catch (const error& e)
                {
                    if (e.code == CHARLS_JPEGLS_ERRC_NOT_ENOUGH_MEMORY)
                        ++codes;
                }
            });
        for (auto& t : threads)
            t.join();
        CHECK(codes == 6, "all six calls of a failed launch report its error");
        CHECK(c.idle(9, 500000), "and the lane is idle again");
    }
    // 6. retracting an announcement lets the leader go; an announced call that never comes costs at most the wait; a STALE
    //    announcement (a handle that was configured long ago) costs nothing
    {
        Coalescer c;
        const Coalescer::Ticket other = c.announce(11);
        std::thread quitter([&] {
            std::this_thread::sleep_for(std::chrono::milliseconds(30));
            c.retract(11, other);
        });
        const ScanDesc d = desc_of(10, 0);
        ScanResult r{};
        auto t0 = Clock::now();
        const Coalescer::Ticket mine = c.announce(11);
        c.submit(11, merge_key_of(d), &d, 1, &r, mine, Coalescer::Policy{3000000, 1024, 2, 750000, 0}, [&](const ScanDesc*, uint32_t, ScanResult* out) { out[0] = ScanResult{}; });
        const double waited = ms_since(t0);
        quitter.join();
        CHECK(waited >= 20 && waited < 1500, "the leader waits for an announced call and stops waiting when it is retracted");
        const Coalescer::Ticket never = c.announce(11); // never submits
        t0 = Clock::now();
        c.submit(11, merge_key_of(d), &d, 1, &r, 0, Coalescer::Policy{50000, 1024, 2, 12500, 0}, [&](const ScanDesc*, uint32_t, ScanResult* out) { out[0] = ScanResult{}; });
        const double capped = ms_since(t0);
        CHECK(capped >= 40 && capped < 1000, "an announced call that never comes costs the wait and no more");
        std::this_thread::sleep_for(std::chrono::milliseconds(80)); // the announcement is older than the wait of the next call
        t0 = Clock::now();
        c.submit(11, merge_key_of(d), &d, 1, &r, 0, Coalescer::Policy{50000, 1024, 2, 12500, 0}, [&](const ScanDesc*, uint32_t, ScanResult* out) { out[0] = ScanResult{}; });
        CHECK(ms_since(t0) < 30, "a stale announcement holds nobody up");
        CHECK(c.idle(11, 50000) && !c.idle(11, 60000000), "idle() applies the same notion of freshness");
        c.retract(11, never);
    }
    // 7. a batch never grows beyond max_scans
    {
        Coalescer c;
        std::atomic<int> largest{0}, total{0};
        std::vector<std::thread> threads;
        std::vector<Coalescer::Ticket> tickets;
        for (int i = 0; i < 20; ++i)
            tickets.push_back(c.announce(13));
        for (int i = 0; i < 20; ++i)
            threads.emplace_back([&, i] {
                const ScanDesc d = desc_of(40, 0);
                ScanResult r{};
                c.submit(13, merge_key_of(d), &d, 1, &r, tickets[i], Coalescer::Policy{200000, 8, 4, 50000, 0}, [&](const ScanDesc*, uint32_t n, ScanResult* out) {
                    largest = std::max<int>(largest, (int)n);
                    total += (int)n;
                    for (uint32_t k = 0; k < n; ++k)
                        out[k] = ScanResult{};
                });
            });
        for (auto& t : threads)
            t.join();
        CHECK(largest <= 8 && total == 20, "batches are capped at max_scans and nobody is lost");
    }
    // 8. max_running batches of a key side by side, the next one waits and collects; other keys are not held up
    {
        Coalescer c;
        std::atomic<int> running{0}, most{0}, launches{0}, other_key_ms{0};
        auto fake = [&](const ScanDesc*, uint32_t n, ScanResult* out) {
            const int now = ++running;
            most = std::max<int>(most, now);
            ++launches;
            std::this_thread::sleep_for(std::chrono::milliseconds(80));
            for (uint32_t k = 0; k < n; ++k)
                out[k] = ScanResult{};
            --running;
        };
        std::vector<std::thread> threads;
        for (int i = 0; i < 10; ++i)
            threads.emplace_back([&, i] {
                std::this_thread::sleep_for(std::chrono::milliseconds(5 * i)); // one after the other: nobody is announced when the first two launch
                const ScanDesc d = desc_of(300, (uint32_t)i);
                ScanResult r{};
                c.submit(21, merge_key_of(d), &d, 1, &r, 0, Coalescer::Policy{20000, 1024, 2, 5000, 0}, fake);
            });
        threads.emplace_back([&] {
            std::this_thread::sleep_for(std::chrono::milliseconds(30)); // two batches of the other key are running
            const ScanDesc d = desc_of(999, 0);
            ScanResult r{};
            const auto t0 = Clock::now();
            c.submit(21, merge_key_of(d), &d, 1, &r, 0, Coalescer::Policy{20000, 1024, 2, 5000, 0}, [&](const ScanDesc*, uint32_t, ScanResult* out) { out[0] = ScanResult{}; });
            other_key_ms = (int)ms_since(t0);
        });
        for (auto& t : threads)
            t.join();
        CHECK(most == 2 && launches == 3, "two batches of a key side by side, the eight late calls in a third");
        CHECK(other_key_ms < 40, "a call of another key does not wait for them");
    }
    // 9. two threads that loop over images fall into step: after the first rounds every launch carries both
    {
        Coalescer c;
        std::atomic<int> launches{0}, pairs{0};
        auto fake = [&](const ScanDesc*, uint32_t n, ScanResult* out) {
            ++launches;
            if (n == 2)
                ++pairs;
            std::this_thread::sleep_for(std::chrono::milliseconds(40));
            for (uint32_t k = 0; k < n; ++k)
                out[k] = ScanResult{};
        };
        auto looper = [&](int offset_ms) {
            std::this_thread::sleep_for(std::chrono::milliseconds(offset_ms));
            for (int round = 0; round < 8; ++round)
            {
                const Coalescer::Ticket t = c.announce(23);
                std::this_thread::sleep_for(std::chrono::milliseconds(2)); // its upload
                const ScanDesc d = desc_of(128, 0);
                ScanResult r{};
                c.submit(23, merge_key_of(d), &d, 1, &r, t, Coalescer::Policy{40000, 1024, 1, 10000, 0}, fake);
            }
        };
        std::thread a(looper, 0), b(looper, 17);
        a.join();
        b.join();
        CHECK(pairs >= 5 && launches <= 11, "two looping threads end up sharing their launches");
    }
    // 10. a shared lane: calls that trickle in without announcing themselves still end up in one batch while they keep coming
    {
        Coalescer c;
        std::atomic<int> launches{0}, largest{0};
        auto fake = [&](const ScanDesc*, uint32_t n, ScanResult* out) {
            ++launches;
            largest = std::max<int>(largest, (int)n);
            std::this_thread::sleep_for(std::chrono::milliseconds(30));
            for (uint32_t k = 0; k < n; ++k)
                out[k] = ScanResult{};
        };
        std::vector<std::thread> threads;
        for (int i = 0; i < 20; ++i)
            threads.emplace_back([&, i] {
                std::this_thread::sleep_for(std::chrono::milliseconds(i == 0 ? 0 : 5 + 8 * i)); // 8 ms apart, after a first pair
                const ScanDesc d = desc_of(222, (uint32_t)i);
                ScanResult r{};
                const Coalescer::Ticket t = i < 2 ? c.announce(25) : 0; // (the first two overlap: the lane is seen to be shared)
                c.submit(25, merge_key_of(d), &d, 1, &r, t, Coalescer::Policy{400000, 1024, 3, 0, 40000}, fake);
            });
        for (auto& t : threads)
            t.join();
        CHECK(launches <= 2 && largest >= 19, "calls that keep joining a shared lane's batch go in one launch");
        // and a caller that is alone still launches at once
        const ScanDesc d = desc_of(222, 0);
        ScanResult r{};
        const auto t0 = Clock::now();
        c.submit(25, merge_key_of(d), &d, 1, &r, 0, Coalescer::Policy{400000, 1024, 3, 0, 40000}, [&](const ScanDesc*, uint32_t, ScanResult* out) { out[0] = ScanResult{}; });
        CHECK(ms_since(t0) < 20, "a caller that is alone does not wait for joiners");
    }
    // 11. announcements say what is coming: a leader does not wait for announcements that can never join its batch (a pool that
    //     codes images of mixed sizes), but it does wait for those that can, and for those that do not know yet
    {
        Coalescer c;
        const uint64_t small = Coalescer::geometry_hint(64, 8, 8), large = Coalescer::geometry_hint(4096, 8, 8);
        auto fake = [&](const ScanDesc*, uint32_t n, ScanResult* out) {
            for (uint32_t k = 0; k < n; ++k)
                out[k] = ScanResult{};
        };
        const Coalescer::Policy policy{300000, 1024, 2, 0, 0};
        const Coalescer::Ticket other = c.announce(27, large); // somebody is uploading a LARGE frame
        {
            const ScanDesc d = desc_of(64, 1);
            ScanResult r{};
            const auto t0 = Clock::now();
            c.submit(27, merge_key_of(d), &d, 1, &r, c.announce(27, small), policy, fake);
            CHECK(ms_since(t0) < 50, "a leader does not wait for an announcement of another geometry");
        }
        {
            const ScanDesc d = desc_of(4096, 2);
            ScanResult r{};
            const auto t0 = Clock::now();
            std::thread late([&] {
                std::this_thread::sleep_for(std::chrono::milliseconds(80));
                c.retract(27, other);
            });
            c.submit(27, merge_key_of(d), &d, 1, &r, c.announce(27, large), policy, fake);
            late.join();
            const double waited = ms_since(t0);
            CHECK(waited >= 60 && waited < 250, "it waits for one of its own geometry (until that is retracted)");
        }
        {
            const Coalescer::Ticket unknown = c.announce(27); // (a decoder that has not read its header yet)
            const ScanDesc d = desc_of(64, 3);
            ScanResult r{};
            const auto t0 = Clock::now();
            std::thread late([&] {
                std::this_thread::sleep_for(std::chrono::milliseconds(60));
                c.renew(27, unknown, large); // now it knows: not ours
            });
            c.submit(27, merge_key_of(d), &d, 1, &r, c.announce(27, small), policy, fake);
            late.join();
            const double waited = ms_since(t0);
            CHECK(waited >= 40 && waited < 200, "an announcement of unknown geometry holds a leader up until it is known to be another's");
            c.retract(27, unknown);
        }
    }
    // 12. an announcement made before the coding call (the caller may only want the header) is fresh for as long as it says:
    //     a lone read-header-only handle costs another thread's call a millisecond, not the leader's whole wait
    {
        Coalescer c;
        const Coalescer::Ticket idle_handle = c.announce(29, 0, 1000); // set_source_buffer, and then nothing
        const ScanDesc d = desc_of(512, 1);
        ScanResult r{};
        const auto t0 = Clock::now();
        c.submit(29, merge_key_of(d), &d, 1, &r, 0, Coalescer::Policy{400000, 1024, 3, 100000, 100000},
                 [&](const ScanDesc*, uint32_t, ScanResult* out) { out[0] = ScanResult{}; });
        const double waited = ms_since(t0);
        CHECK(waited < 20, "an early announcement holds a call up for a millisecond at most");
        std::printf("   (waited %.2f ms)\n", waited);
        c.retract(29, idle_handle);
    }
    // 13. a merged launch that runs out of memory is run again call by call: every caller gets the outcome of ITS scans
    {
        Coalescer c;
        std::atomic<int> launches{0}, failed_calls{0}, good_calls{0};
        auto fake = [&](const ScanDesc* all, uint32_t n, ScanResult* out) {
            ++launches;
            if (n > 2)
                raise(CHARLS_JPEGLS_ERRC_NOT_ENOUGH_MEMORY); // (the merged launch does not fit)
            if (all[0].stream_capacity == 5)
                raise(CHARLS_JPEGLS_ERRC_INVALID_DATA); // (one call is bad on its own account)
            for (uint32_t k = 0; k < n; ++k)
                out[k] = ScanResult{0, 0, all[k].stream_capacity + 1000};
        };
        std::vector<Coalescer::Ticket> tickets;
        for (int i = 0; i < 8; ++i)
            tickets.push_back(c.announce(31));
        std::vector<std::thread> threads;
        for (int i = 0; i < 8; ++i)
            threads.emplace_back([&, i] {
                const ScanDesc d = desc_of(640, (uint32_t)i);
                ScanResult r{};
                try
                {
                    c.submit(31, merge_key_of(d), &d, 1, &r, tickets[i], Coalescer::Policy{500000, 1024, 2, 0, 0}, fake);
                    if (r.bytes == 1000u + i)
                        ++good_calls;
                }
                catch (const error& e)
                {
                    if (i == 5 && e.code == CHARLS_JPEGLS_ERRC_INVALID_DATA)
                        ++failed_calls;
                }
            });
        for (auto& t : threads)
            t.join();
        CHECK(good_calls == 7 && failed_calls == 1 && launches == 9 && c.stats().split == 1,
              "out of memory in a merged launch: the calls run one by one, only the bad one fails");
    }
    std::printf(g_failures == 0 ? "coalescer ok\n" : "coalescer FAILED\n");
    return g_failures == 0 ? 0 : 1;
}
