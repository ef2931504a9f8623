// coalescer.h -- merges concurrent calls of the host-pointer C ABI into one kernel launch.
//
// The reference has no shared mutable state (src/golomb_lut.cpp:65-69, src/quantization_lut.cpp:40-62), every coding call
// builds its own codec (src/charls_jpegls_decoder.cpp:177-201), and its callers scale by threads x handles (SURVEY 8b
// "Threading").  On this engine a scan that is launched ALONE has a whole kernel to itself -- one decoder chain on one of
// 1024 SIMDs at 5 MPix/s, an encoder pipeline that fills the chip with one frame's tiles -- and what makes the GPU fast is
// the number of scans in a launch (8 scans share a decoder wavefront; a pass of the encoder takes hundreds of frames; and the
// hardware runs only a handful of kernels of different streams side by side).  So calls that arrive together are put into
// ONE launch:
//
//  * a handle ANNOUNCES itself on its lane (device x direction) as soon as it is clear that a coding call is coming -- the
//    decoder when it is given its source, the encoder its frame info -- and SUBMITS its scans when its host -> device copy is
//    through.  An announcement says WHAT is coming as far as the handle knows (a hash of the frame's geometry; 0 = not known
//    yet) and for how long it counts: one made before the coding call (the caller may only want the header) is FRESH for a
//    millisecond, one made inside the coding call -- its upload is on the way -- for the leader's `wait_us`.  A leader only
//    waits for announcements that can join ITS batch (same geometry, or not known yet): in a pool that codes images of mixed
//    sizes there is almost always a fresh announcement of SOME kind, and waiting for those made every call wait its full
//    `wait_us` and still launch alone;
//  * the first submitter of a kind (key = geometry and coding parameters) leads a batch: it waits while fresh announcements
//    are outstanding (never longer than `wait_us`; a caller that is alone has nobody to wait for and launches at once), then
//    runs the launch for everybody on its own stream and hands the results out;
//  * on a lane that is visibly shared (somebody joined the batch already, or another batch of the lane is running) the leader
//    also waits while calls KEEP JOINING -- the batch goes when none has joined for `gap_us`: a pool of threads that comes out
//    of its encoder calls over a few hundred milliseconds decodes as one batch, not as three small ones and a big one that
//    waits for them (profiles/r05_threads_call_trace.txt);
//  * a launch-level failure of a merged batch that each call could have survived alone (out of memory: the staging areas
//    of 16384 merged scans) is not everybody's failure: the leader runs the calls of the batch one by one and every caller gets
//    the outcome of ITS scans;
//  * at most `max_running` batches of a key run at a time (an encoder lane: ONE batch of any key -- the work areas of the
//    pipeline are shared).  A batch that has to wait for its turn stays OPEN meanwhile -- group commit -- and when its turn
//    comes it may give the callers of the batch that just finished a moment (`grace_us`) to come back: decoder threads that
//    loop over images fall into step that way instead of taking turns.
//
// Pure C++ (no HIP): tests/test_coalescer_cpu.py drives it with a fake launch.
#pragma once
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../device/scan_types.h"
#include "common.h"

namespace jls {

struct MergeKey
{
    uint64_t words[6];
    bool operator<(const MergeKey& o) const noexcept { return std::memcmp(words, o.words, sizeof words) < 0; }
};

// Scans may share a launch when everything the launch code derives from its `proto` is equal: geometry, coding
// parameters, sample alignment (device/runtime.h: launch_decode / launch_encode).
inline MergeKey merge_key_of(const ScanDesc& d) noexcept
{
    MergeKey k{};
    k.words[0] = (static_cast<uint64_t>(d.width) << 32) | d.height;
    k.words[1] = (static_cast<uint64_t>(static_cast<uint32_t>(d.components)) << 32) | (static_cast<uint64_t>(static_cast<uint32_t>(d.interleave_mode)) << 16) |
                 static_cast<uint32_t>(d.bits_per_sample);
    k.words[2] = (static_cast<uint64_t>(static_cast<uint32_t>(d.near_lossless)) << 32) | static_cast<uint32_t>(d.color_transformation);
    k.words[3] = (static_cast<uint64_t>(static_cast<uint32_t>(d.t1)) << 32) | static_cast<uint32_t>(d.t2);
    k.words[4] = (static_cast<uint64_t>(static_cast<uint32_t>(d.t3)) << 32) | static_cast<uint32_t>(d.reset);
    k.words[5] = (static_cast<uint64_t>(d.restart_interval) << 32) | ((reinterpret_cast<uintptr_t>(d.pixels) | d.pixel_stride) & 1u);
    return k;
}

class Coalescer
{
public:
    using Clock = std::chrono::steady_clock;
    // Runs ONE launch for descs[0, n) and fills results[0, n); may raise jls::error (every call of the batch then fails
    // with that code).
    using Launch = std::function<void(const ScanDesc* descs, uint32_t n, ScanResult* results)>;

    struct Stats
    {
        uint64_t calls;    // submissions
        uint64_t launches; // batches run
        uint64_t merged;   // submissions that shared their launch with another one
        uint64_t largest;  // scans in the largest batch
        uint64_t split;    // merged batches whose launch ran out of memory and whose calls were then run one by one
    };

    struct Policy
    {
        uint32_t wait_us;     // how long announcements are fresh, and the longest a leader waits for them
        uint32_t max_scans;   // scans of a batch at most
        uint32_t max_running; // batches of ONE key that may run at a time (0: the lane runs one batch of ANY key at a time)
        uint32_t grace_us;    // what a batch that had to wait for its turn gives the callers of the finished batch to come back
        uint32_t gap_us;      // on a lane that is visibly shared: the batch goes when no call has joined for this long
    };

    static constexpr int kLanes = 64; // device * 2 + (decode ? 1 : 0), devices 0..31
    using Ticket = uint64_t;          // of an announcement; 0 = none
    static constexpr uint32_t kForever = 0xFFFFFFFFu;

    // `hint`: what kind of call is coming (geometry_hint; 0 = not known yet: it may join anything); `fresh_us`: for how long the
    // announcement holds anybody up at most (kForever: as long as the leader's own wait).  announce(lane) is the announcement
    // of a call whose upload is on the way.
    Ticket announce(int lane, uint64_t hint = 0, uint32_t fresh_us = kForever)
    {
        std::lock_guard<std::mutex> lock(mutex_);
        const Ticket ticket = ++last_ticket_;
        const Clock::time_point now = Clock::now();
        announced_[lane % kLanes].emplace(ticket, Announcement{now, fresh_us == kForever ? Clock::time_point::max() : now + std::chrono::microseconds(fresh_us), hint});
        return ticket;
    }

    // The announced call knows more now (its geometry) or has got further (into its coding call: fresh from now on).
    void renew(int lane, Ticket ticket, uint64_t hint, uint32_t fresh_us = kForever) noexcept
    {
        if (ticket == 0)
            return;
        std::lock_guard<std::mutex> lock(mutex_);
        auto& list = announced_[lane % kLanes];
        const auto it = list.find(ticket);
        if (it == list.end())
            return;
        const Clock::time_point now = Clock::now();
        it->second = Announcement{now, fresh_us == kForever ? Clock::time_point::max() : now + std::chrono::microseconds(fresh_us), hint};
        arrival_.notify_all();
    }

    // What an announcement tells about the kind of call: width, height and sample width of the frame (calls that differ in more
    // than that -- components per scan, coding parameters -- still wait for each other; they are rare).
    static uint64_t geometry_hint(uint32_t width, uint32_t height, int bits_per_sample) noexcept
    {
        uint64_t h = (static_cast<uint64_t>(width) << 32) ^ height ^ (static_cast<uint64_t>(static_cast<uint32_t>(bits_per_sample)) << 56);
        h ^= h >> 29;
        h *= 0x9E3779B97F4A7C15ull;
        h ^= h >> 32;
        return h == 0 ? 1 : h;
    }

    // An announced call that will not submit after all (its handle went away, or the call failed before it got that far).
    void retract(int lane, Ticket ticket) noexcept
    {
        if (ticket == 0)
            return;
        std::lock_guard<std::mutex> lock(mutex_);
        announced_[lane % kLanes].erase(ticket);
        arrival_.notify_all();
    }

    // Blocks until the `count` scans are done.  `ticket`: this call's announcement on the lane (0: it made none).  `launch`
    // runs on the calling thread when it leads the batch, and not at all when it joined somebody else's.
    void submit(int lane, const MergeKey& key, const ScanDesc* descs, uint32_t count, ScanResult* results, Ticket ticket, const Policy& policy,
                const Launch& launch)
    {
        lane %= kLanes;
        const auto wait = std::chrono::microseconds(policy.wait_us);
        std::unique_lock<std::mutex> lock(mutex_);
        ++stats_.calls;
        if (ticket != 0)
            announced_[lane].erase(ticket);
        const auto id = std::make_pair(lane, key);
        auto open = open_.find(id);
        if (open != open_.end() && open->second->descs.size() + count <= policy.max_scans)
        { // join
            std::shared_ptr<Batch> b = open->second;
            const size_t first = b->descs.size();
            const size_t call = b->call_first.size();
            b->descs.insert(b->descs.end(), descs, descs + count);
            b->call_first.push_back(static_cast<uint32_t>(first));
            b->call_failure.push_back(CHARLS_JPEGLS_ERRC_SUCCESS);
            ++b->calls;
            b->last_join = Clock::now();
            arrival_.notify_all(); // the leader looks again
            b->finished.wait(lock, [&] { return b->done; });
            if (b->call_failure[call] != CHARLS_JPEGLS_ERRC_SUCCESS)
                raise(b->call_failure[call]);
            std::memcpy(results, b->results.data() + first, sizeof(ScanResult) * count);
            return;
        }
        // lead
        auto b = std::make_shared<Batch>();
        b->descs.assign(descs, descs + count);
        b->call_first.push_back(0);
        b->call_failure.push_back(CHARLS_JPEGLS_ERRC_SUCCESS);
        b->calls = 1;
        const uint64_t hint = descs[0].width == 0 ? 0 : geometry_hint(descs[0].width, descs[0].height, descs[0].bits_per_sample);
        b->last_join = Clock::now();
        const bool published = open == open_.end();
        if (published)
            open_[id] = b; // (otherwise a full batch of this kind is still open: this one stays private and runs on its own)
        arrival_.notify_all();
        auto running_now = [&]() -> uint32_t& { return policy.max_running == 0 ? lane_running_[lane] : key_running_[id]; };
        const uint32_t limit = policy.max_running == 0 ? 1u : policy.max_running;
        auto deadline = Clock::now() + wait;
        Clock::time_point grace{}; // set when the batch had to wait for its turn
        bool had_to_wait = false;
        for (;;)
        {
            const Clock::time_point now = Clock::now();
            const bool my_turn = running_now() < limit;
            if (!my_turn)
            { // (the batch stays open while another one runs: whoever arrives meanwhile joins it)
                had_to_wait = true;
                arrival_.wait(lock);
                continue;
            }
            if (had_to_wait)
            { // the callers of the batch that just finished get a moment to come back and join
                had_to_wait = false;
                grace = now + std::chrono::microseconds(policy.grace_us);
                deadline = now + wait;
            }
            const bool room = published && b->descs.size() < policy.max_scans;
            const bool in_grace = room && now < grace;
            // Others are on their way while an announcement is fresh -- or, where the lane is visibly shared (somebody joined
            // this batch already, or another batch of the lane is running), while calls keep joining: the batch goes when
            // none has joined for `gap_us`.  (A caller that is alone sees neither and launches at once.)
            const bool shared = b->calls > 1 || lane_busy(lane);
            const Clock::time_point quiet = b->last_join + std::chrono::microseconds(policy.gap_us);
            const bool still_joining = shared && now < quiet;
            const bool others_coming = room && now < deadline && (fresh_announcements(lane, now, wait, hint) || still_joining);
            if (!in_grace && !others_coming)
                break;
            Clock::time_point until = deadline;
            if (in_grace && !others_coming)
                until = grace;
            else if (!fresh_announcements(lane, now, wait, hint) && still_joining)
                until = std::min(deadline, quiet);
            else
                until = std::min(until, next_expiry(lane, now, hint)); // (an announcement that goes stale wakes nobody by itself)
            wait_until(lock, until);
        }
        if (published)
            open_.erase(id);
        ++running_now();
        // From here on whatever happens -- an allocation that fails, a launch that throws -- the running count goes down again
        // and everybody who joined is told: a count that leaks blocks the lane for good, a batch that is never `done` its callers.
        charls_jpegls_errc failure = CHARLS_JPEGLS_ERRC_SUCCESS;
        bool was_split = false;
        try
        {
            ++stats_.launches;
            if (b->calls > 1)
                stats_.merged += b->calls;
            stats_.largest = std::max<uint64_t>(stats_.largest, b->descs.size());
            b->results.resize(b->descs.size());
            lock.unlock();
            try
            {
                launch(b->descs.data(), static_cast<uint32_t>(b->descs.size()), b->results.data());
            }
            catch (...)
            {
                failure = current_exception_to_errc();
            }
            if (failure == CHARLS_JPEGLS_ERRC_NOT_ENOUGH_MEMORY && b->calls > 1)
            { // what a call could have survived alone is not everybody's failure: the calls of the batch one by one
                failure = CHARLS_JPEGLS_ERRC_SUCCESS;
                for (size_t call = 0; call < b->call_first.size(); ++call)
                {
                    const uint32_t first = b->call_first[call];
                    const uint32_t end = call + 1 < b->call_first.size() ? b->call_first[call + 1] : static_cast<uint32_t>(b->descs.size());
                    try
                    {
                        launch(b->descs.data() + first, end - first, b->results.data() + first);
                    }
                    catch (...)
                    {
                        b->call_failure[call] = current_exception_to_errc();
                    }
                }
                was_split = true;
            }
        }
        catch (...)
        {
            failure = current_exception_to_errc();
        }
        if (!lock.owns_lock())
            lock.lock();
        if (was_split)
            ++stats_.split;
        if (--running_now() == 0 && policy.max_running != 0)
            key_running_.erase(id);
        if (failure != CHARLS_JPEGLS_ERRC_SUCCESS)
            std::fill(b->call_failure.begin(), b->call_failure.end(), failure);
        b->done = true;
        b->finished.notify_all();
        arrival_.notify_all(); // the leaders that wait for their turn
        if (b->call_failure[0] != CHARLS_JPEGLS_ERRC_SUCCESS)
            raise(b->call_failure[0]);
        std::memcpy(results, b->results.data(), sizeof(ScanResult) * count);
    }

    // No fresh announcement and no open batch on the lane (and, unless asked otherwise, nothing running): whoever holds
    // shared work areas for it may give them back.
    bool idle(int lane, uint32_t fresh_us, bool but_for_the_running_batch = false)
    {
        lane %= kLanes;
        std::lock_guard<std::mutex> lock(mutex_);
        if (fresh_announcements(lane, Clock::now(), std::chrono::microseconds(fresh_us), 0))
            return false;
        if (!but_for_the_running_batch)
        {
            if (lane_running_[lane] != 0)
                return false;
            for (const auto& entry : key_running_)
                if (entry.first.first == lane)
                    return false;
        }
        for (const auto& entry : open_)
            if (entry.first.first == lane)
                return false;
        return true;
    }

    Stats stats()
    {
        std::lock_guard<std::mutex> lock(mutex_);
        return stats_;
    }

private:
    struct Batch
    {
        std::vector<ScanDesc> descs;
        std::vector<ScanResult> results;
        std::vector<uint32_t> call_first;               // first scan of every call of the batch (call 0: the leader)
        std::vector<charls_jpegls_errc> call_failure;   // what every call is told
        uint32_t calls{};
        Clock::time_point last_join{}; // when the latest call joined (the leader counts)
        bool done{};
        std::condition_variable finished;
    };

    struct Announcement
    {
        Clock::time_point at;      // when it was made or renewed
        Clock::time_point expires; // its own freshness (a handle that was configured and then left alone holds nobody up)
        uint64_t hint;             // geometry_hint of the call that is coming; 0 = not known yet
    };

    // std::condition_variable::wait_until on a steady clock is pthread_cond_clockwait, which gcc's ThreadSanitizer does not
    // intercept (it then reports every access under the mutex as a race); the sanitizer build waits on the system clock.
    void wait_until(std::unique_lock<std::mutex>& lock, Clock::time_point until)
    {
#ifdef JLS_TSAN
        const auto left = until - Clock::now();
        if (left > Clock::duration::zero())
            arrival_.wait_until(lock, std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::system_clock::duration>(left));
#else
        arrival_.wait_until(lock, until);
#endif
    }

    bool lane_busy(int lane) const
    {
        if (lane_running_[lane] != 0)
            return true;
        for (const auto& entry : key_running_)
            if (entry.first.first == lane && entry.second != 0)
                return true;
        return false;
    }

    // Is a call on its way to this lane that could join a batch of kind `hint` (0: of any kind)?  An announcement older than
    // `wait`, or past its own freshness, is stale -- a handle that was configured and then left alone -- and holds nobody up.
    bool fresh_announcements(int lane, Clock::time_point now, std::chrono::microseconds wait, uint64_t hint) const
    {
        for (const auto& entry : announced_[lane])
        {
            const Announcement& a = entry.second;
            if (now - a.at <= wait && now < a.expires && (hint == 0 || a.hint == 0 || a.hint == hint))
                return true;
        }
        return false;
    }

    // When the first of the announcements a leader of kind `hint` waits for stops being fresh by its own clock.
    Clock::time_point next_expiry(int lane, Clock::time_point now, uint64_t hint) const
    {
        Clock::time_point first = Clock::time_point::max();
        for (const auto& entry : announced_[lane])
        {
            const Announcement& a = entry.second;
            if (now < a.expires && (hint == 0 || a.hint == 0 || a.hint == hint))
                first = std::min(first, a.expires);
        }
        return first;
    }

    std::mutex mutex_;
    std::condition_variable arrival_; // an announced call submitted or retracted; a running batch finished
    std::map<std::pair<int, MergeKey>, std::shared_ptr<Batch>> open_;
    std::map<std::pair<int, MergeKey>, uint32_t> key_running_;
    std::map<Ticket, Announcement> announced_[kLanes];
    uint32_t lane_running_[kLanes]{};
    Ticket last_ticket_{};
    Stats stats_{};
};

} // namespace jls
