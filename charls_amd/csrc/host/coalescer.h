// coalescer.h -- merges concurrent calls of the host-pointer C ABI into one kernel launch.
//
// The reference has no shared mutable state (src/golomb_lut.cpp:65-69, src/quantization_lut.cpp:40-62), every coding call
// builds its own codec (src/charls_jpegls_decoder.cpp:177-201), and its callers scale by threads x handles (SURVEY 8b
// "Threading").  On this engine a scan that is launched ALONE has a whole kernel to itself -- one decoder chain on one of
// 1024 SIMDs at 5 MPix/s, an encoder pipeline that fills the chip with one frame's tiles -- and what makes the GPU fast is
// the number of scans in a launch (8 scans share a decoder wavefront; a pass of the encoder takes hundreds of frames).  So
// calls that arrive together are put into ONE launch:
//
//  * a call ANNOUNCES itself when its host->device copy starts (ScanEngine::upload_*) and SUBMITS its scans when the copy is
//    through;
//  * the first submitter of a kind (lane = device x direction, key = geometry and coding parameters) becomes the leader of
//    a batch: it waits while announced calls are still on their way (never longer than `wait_us`; a caller that is alone
//    has nobody to wait for and launches at once), then runs the launch for everybody on its own stream and hands the
//    results out;
//  * an EXCLUSIVE lane (encode: the work areas of the pipeline are shared) runs one batch at a time, and the next batch
//    stays open while the current one runs -- group commit.
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
    // Runs ONE launch for descs[0, n) and fills results[0, n); may raise jls::error (every call of the batch then fails
    // with that code).
    using Launch = std::function<void(const ScanDesc* descs, uint32_t n, ScanResult* results)>;

    struct Stats
    {
        uint64_t calls;    // submissions
        uint64_t launches; // batches run
        uint64_t merged;   // submissions that shared their launch with another one
        uint64_t largest;  // scans in the largest batch
    };

    static constexpr int kLanes = 64; // device * 2 + (decode ? 1 : 0), devices 0..31

    void announce(int lane) noexcept
    {
        std::lock_guard<std::mutex> lock(mutex_);
        ++expected_[lane % kLanes];
    }

    // An announced call that will not submit after all (it failed before it got that far).
    void retract(int lane) noexcept
    {
        std::lock_guard<std::mutex> lock(mutex_);
        if (expected_[lane % kLanes] > 0)
            --expected_[lane % kLanes];
        arrival_.notify_all();
    }

    // Blocks until the `count` scans are done.  `announced`: this call announced itself on the lane.  `launch` runs on
    // the calling thread when it leads the batch, and not at all when it joined somebody else's.
    void submit(int lane, const MergeKey& key, const ScanDesc* descs, uint32_t count, ScanResult* results, bool announced, bool exclusive,
                uint32_t wait_us, uint32_t max_scans, const Launch& launch)
    {
        lane %= kLanes;
        std::unique_lock<std::mutex> lock(mutex_);
        ++stats_.calls;
        if (announced && expected_[lane] > 0)
            --expected_[lane];
        const auto id = std::make_pair(lane, key);
        auto open = open_.find(id);
        if (open != open_.end() && open->second->descs.size() + count <= max_scans)
        { // join
            std::shared_ptr<Batch> b = open->second;
            const size_t first = b->descs.size();
            b->descs.insert(b->descs.end(), descs, descs + count);
            ++b->calls;
            arrival_.notify_all(); // the leader looks at `expected_` again
            b->finished.wait(lock, [&] { return b->done; });
            if (b->failure != CHARLS_JPEGLS_ERRC_SUCCESS)
                raise(b->failure);
            std::memcpy(results, b->results.data() + first, sizeof(ScanResult) * count);
            return;
        }
        // lead
        auto b = std::make_shared<Batch>();
        b->descs.assign(descs, descs + count);
        b->calls = 1;
        const bool published = open == open_.end();
        if (published)
            open_[id] = b; // (a full batch of this kind is still open: this one stays private and runs on its own)
        else
            arrival_.notify_all();
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(wait_us);
        for (;;)
        {
            const bool others_coming = published && expected_[lane] > 0 && b->descs.size() < max_scans &&
                                       std::chrono::steady_clock::now() < deadline;
            const bool not_my_turn = exclusive && busy_[lane];
            if (!others_coming && !not_my_turn)
                break;
            if (not_my_turn) // (the batch stays open while another one runs: whoever arrives meanwhile joins it)
                arrival_.wait(lock);
            else
                arrival_.wait_until(lock, deadline);
        }
        if (published)
            open_.erase(id);
        if (exclusive)
            busy_[lane] = true;
        ++stats_.launches;
        if (b->calls > 1)
            stats_.merged += b->calls;
        stats_.largest = std::max<uint64_t>(stats_.largest, b->descs.size());
        b->results.resize(b->descs.size());
        lock.unlock();
        charls_jpegls_errc failure = CHARLS_JPEGLS_ERRC_SUCCESS;
        try
        {
            launch(b->descs.data(), static_cast<uint32_t>(b->descs.size()), b->results.data());
        }
        catch (...)
        {
            failure = current_exception_to_errc();
        }
        lock.lock();
        if (exclusive)
            busy_[lane] = false;
        b->failure = failure;
        b->done = true;
        b->finished.notify_all();
        arrival_.notify_all(); // the leader of the next batch of an exclusive lane
        if (failure != CHARLS_JPEGLS_ERRC_SUCCESS)
            raise(failure);
        std::memcpy(results, b->results.data(), sizeof(ScanResult) * count);
    }

    // Nothing announced, open or running on the lane: whoever holds shared work areas for it may give them back.
    bool idle(int lane, bool but_for_the_running_batch = false)
    {
        lane %= kLanes;
        std::lock_guard<std::mutex> lock(mutex_);
        if (expected_[lane] > 0 || (busy_[lane] && !but_for_the_running_batch))
            return false;
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
        uint32_t calls{};
        bool done{};
        charls_jpegls_errc failure{CHARLS_JPEGLS_ERRC_SUCCESS};
        std::condition_variable finished;
    };

    std::mutex mutex_;
    std::condition_variable arrival_; // an announced call submitted or retracted; a batch of an exclusive lane finished
    std::map<std::pair<int, MergeKey>, std::shared_ptr<Batch>> open_;
    uint32_t expected_[kLanes]{};
    bool busy_[kLanes]{};
    Stats stats_{};
};

} // namespace jls
