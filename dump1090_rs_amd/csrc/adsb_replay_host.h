// adsb_replay_host.h -- the host-only half of the library: the ordered replay every demod call ends with and the
// small helpers around it.  Nothing here touches HIP (adsb_replay_host.cpp builds with plain g++ as well).
#pragma once
#include <cstddef>
#include <cstdint>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/adsb_hip.h"
#include "adsb_record.h"
#include "mode_s_host.hpp"

// The tail of every extern "C" function whose body can allocate (`int f(...) try { ... } ADSB_ABI_CATCH`): nothing is
// thrown across the ABI (include/adsb_hip.h) -- a Rust host would abort on an unwinding foreign frame.
#define ADSB_ABI_CATCH                                        \
    catch (const std::bad_alloc &) { return ADSB_ERR_NOMEM; } \
    catch (...) { return ADSB_ERR_HIP; }

namespace adsb {
namespace host {

// Ordered replay (src/demod_2400.rs:149-207 with src/mode_s/mod.rs:34-139 scoring against src/icao_filter.rs):
// records in any order, replayed by (chunk, j, try_phase) and left where they are; `chunk_offset` is added to
// the messages' chunk.  host_sorts: counted up when the records had to be put in order here.
void replay(IcaoFilter &filter, const Crc24 &crc, const TrialRecord *rec, size_t n, uint64_t chunk_offset,
            std::vector<adsb_msg> &out, uint64_t *host_sorts = nullptr);

// ... records known to be in replay order (no check)
void replay_sorted(IcaoFilter &filter, const Crc24 &crc, const TrialRecord *rec, size_t n, uint64_t chunk_offset, std::vector<adsb_msg> &out);

bool replay_order(const TrialRecord *rec, size_t n, std::vector<uint32_t> &order_out);
bool sort_records(const TrialRecord *rec, size_t n, std::vector<TrialRecord> &sorted_out);

// ---------------------------------------------------------------------------------------------------------------
// The same replay by several threads at once (adsb_multi's collector: a busy sky leaves tens of thousands of records
// per capture and the ordered replay was its one serial stage).
//
// Scoring reads and writes the filter (src/mode_s/mod.rs:71,80-84,97-104,115,130), but between two flushes the filter only
// grows, and WHEN a value enters it is a property of the records alone: a value is added by the first record in replay
// order that can add it -- a DF17 or a DF11 / IID 0 with a clean CRC adds its address whenever the filter does not
// hold it, whether or not the trial wins its position (the reference scores all five phases, mod.rs is called from
// demod_2400.rs:158-182).  So "is address a in the filter when record number p (in replay order) is scored" is
//     a was in the filter when the capture began   OR   first_adder(a) < p
// and with the first adders known every record can be scored independently of every other:
//   plan        the runs (each in replay order, ascending) are cut into parts at position boundaries
//   scan_part   every part finds the first adder of each value it can add                    (parallel)
//   merge       the parts' tables into one; refuses when the 4096-slot table could fill up (then add() gives up,
//               src/icao_filter.rs:46-62, and membership is no longer a set's: the caller replays serially)
//   score_part  every part scores its records against (filter as it was, first adders)        (parallel)
//   finish      the parts' messages in order; the new values enter the filter in the order of their first adders,
//               which is the order the serial replay inserts them in (same table, slot for slot)
// DF18 adds address | 1 << 25 (mod.rs:100-104), which no test ever asks for: it only takes a slot, when its first
// DF18 record finds the plain address unknown.
struct RecordRun {
    const TrialRecord *rec;
    size_t n;
    uint64_t chunk_offset;
};

class ParallelReplay {
  public:
    // false: not worth it or not possible (too few records, a run out of order): replay serially
    // (runs_in_order: the caller has checked every run's own order already.  run_adders: per run, what first_adders()
    // found in it -- whoever had the run in its cache before has done the scan stage's work already, scan_needed() says no)
    typedef std::vector<std::pair<uint32_t, uint64_t>> Adders;   // (value as added, index in the run of its first adder)
    bool plan(const IcaoFilter &filter, const Crc24 &crc, const std::vector<RecordRun> &runs, int parts, bool runs_in_order = false,
              const std::vector<const Adders *> *run_adders = nullptr);
    int parts() const { return (int)part_.size(); }
    bool scan_needed() const { return run_adders_.empty(); }
    void scan_part(int i);
    bool merge();
    void score_part(int i);
    void finish(IcaoFilter &filter, std::vector<adsb_msg> &out);
    // ... or, instead of finish: the parts' messages straight to where they are wanted, every part by whoever holds it in
    // its cache (message_count() of them from dst on; streaming stores: the destination is somebody else's to read),
    // then apply_adds
    size_t message_count() const;
    void copy_to(adsb_msg *dst);
    void copy_part(int i);
    // the parts' messages, for a caller that copies them out itself (in order: part 0, 1, ...)
    const std::vector<adsb_msg> &messages(int i) const { return part_[(size_t)i].out; }
    void apply_adds(IcaoFilter &filter) const;

    typedef uint64_t Pos;            // a record's number in replay order over all runs (two records of one (buffer, j,
                                     // try_phase) -- the device never makes them, a caller's own records may -- keep theirs)
    struct FirstAdds {               // open addressing, value -> number of its first adder
        std::vector<uint32_t> key;   // value + 1 (0: empty)
        std::vector<Pos> pos;
        uint32_t mask = 0, used = 0;
        void reset(uint32_t capacity_pow2);
        void put_min(uint32_t value, Pos p);
        bool put_first(uint32_t value, Pos p);   // only when absent (records scanned in order: the first seen is the first)
        Pos get(uint32_t value) const;   // ~0 when absent
    };

  private:
    struct Piece {
        RecordRun run;
        Pos first;            // the number of run.rec[0]
    };
    struct alignas(128) Part {   // (a part has one writer at a time: no two parts in one cache line)
        std::vector<Piece> runs;
        FirstAdds adds;
        std::vector<std::pair<uint32_t, Pos>> found;   // what `adds` holds, as a list (for the merge)
        std::vector<adsb_msg> out;
        size_t out_at = 0;    // copy_to: where the part's messages go
    };
    const IcaoFilter *filter_ = nullptr;
    const Crc24 *crc_ = nullptr;
    std::vector<Part> part_;
    FirstAdds all_;
    size_t n_records_ = 0;
    adsb_msg *dst_ = nullptr;
    std::vector<std::pair<const Adders *, Pos>> run_adders_;   // (the run's list, the number of the run's first record)
    std::vector<std::pair<Pos, uint32_t>> new_values_;   // (first adder, value as added), in insertion order
};

// The first record (by index) of a run in replay order that can add each value: what ParallelReplay's scan stage finds,
// for one whole run (mode_s/mod.rs:80-84 DF11 / IID 0, :97-104 DF17 / DF18 with a clean CRC).
void first_adders(const Crc24 &crc, const TrialRecord *rec, size_t n, ParallelReplay::Adders &out);

// The threads that score a capture's records side by side (adsb_replay_host.h: ParallelReplay).  A job is a stage of
// one capture's replay -- parts handed out by a counter to whoever is awake, the caller included -- and is done when
// every part is; a thread that wakes up late finds the counter of ITS job used up and goes back to waiting.
class ReplayPool {
  public:
    // (on_start(k): run by worker k before anything else -- adsb_multi places it on a device's NUMA node; hot_us: how long
    // a worker keeps spinning for the next job before it goes to sleep -- 0 for a host that is short of CPUs)
    explicit ReplayPool(int workers, std::function<void(int)> on_start = {}, int hot_us = 1500) : hot_us_(hot_us)
    {
        try {
            th_.reserve((size_t)workers);
            for (int k = 0; k < workers; k++) th_.emplace_back([this, k, on_start] { work(k, on_start); });
        } catch (...) {
            shutdown();   // (a thread that could not be started: the ones that were must be joined before the vector goes)
            throw;
        }
    }
    ~ReplayPool() { shutdown(); }
    void shutdown() noexcept
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto &t : th_)
            if (t.joinable()) t.join();
    }
    int threads() const { return (int)th_.size() + 1; }
    void run(ParallelReplay &pr, void (ParallelReplay::*stage)(int))
    {
        auto job = std::make_shared<Job>();
        job->pr = &pr;
        job->stage = stage;
        job->parts = pr.parts();
        job->claimed.reset(new std::atomic<uint8_t>[(size_t)job->parts]);
        for (int i = 0; i < job->parts; i++) job->claimed[(size_t)i].store(0, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = job;
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        take(*job, (int)th_.size());
        while (job->done.load(std::memory_order_acquire) < job->parts) __builtin_ia32_pause();
        // (a stage that ran out of memory on a worker: an exception must not leave a thread function -- it is caught where
        // it happens and raised again here, on the caller's thread, whose callers turn it into a status)
        if (job->failed.load(std::memory_order_acquire)) throw std::bad_alloc();
    }

  private:
    struct Job {
        ParallelReplay *pr = nullptr;
        void (ParallelReplay::*stage)(int) = nullptr;
        int parts = 0;
        std::unique_ptr<std::atomic<uint8_t>[]> claimed;
        std::atomic<int> done{0};
        std::atomic<bool> failed{false};
    };
    // Thread `me` of T takes parts me, me + T, ... first -- the same ones in both stages of a capture, so the second
    // stage finds its records in the cache the first left them in -- and then whatever nobody has claimed (a thread
    // that woke up late, or is not running at all, holds nobody up).
    void take(Job &job, int me) const
    {
        const int T = threads();
        auto claim = [&](int i) {
            if (job.claimed[(size_t)i].exchange(1, std::memory_order_acq_rel)) return;
            try {
                (job.pr->*job.stage)(i);
            } catch (...) {
                job.failed.store(true, std::memory_order_release);
            }
            job.done.fetch_add(1, std::memory_order_release);
        };
        for (int i = me; i < job.parts; i += T) claim(i);
        for (int i = 0; i < job.parts; i++) claim(i);
    }
    void work(int me, const std::function<void(int)> &on_start)
    {
        if (on_start) on_start(me);
        uint64_t seen = 0;
        for (;;) {
            // a capture's second stage follows its first within microseconds, a busy stream's next capture within a
            // millisecond: stay hot that long, then sleep
            const auto t0 = std::chrono::steady_clock::now();
            while (gen_.load(std::memory_order_acquire) == seen && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(hot_us_))
                __builtin_ia32_pause();
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_.load(std::memory_order_relaxed) != seen; });
                seen = gen_.load(std::memory_order_relaxed);
                if (stop_) return;
                job = job_;
            }
            if (job) take(*job, me);
        }
    }
    const int hot_us_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<uint64_t> gen_{0};
    bool stop_ = false;
    std::shared_ptr<Job> job_;
};

// The sorted union of several sorted, duplicate-free address lists (the shards' learned addresses), appended to
// `out` (cleared first); what is already in `known` (sorted, duplicate-free) is left out.
void union_sorted(const std::vector<const std::vector<uint32_t> *> &lists, const std::vector<uint32_t> &known,
                  std::vector<uint32_t> &out);

// mode_s/mod.rs:80-84 (DF11, IID 0) and :97-99 (DF17): the addresses the replay of these records can add.
void learned_addresses(const Crc24 &crc, const TrialRecord *rec, size_t n, std::vector<uint32_t> &addrs);

}  // namespace host
}  // namespace adsb
