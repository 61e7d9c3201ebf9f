// adsb_multi.cpp -- one capture over N GPUs from ONE process, behind the C ABI (adsb_multi_*).
//
// The reference is one process, one loop, one process-global ICAO filter (dump1090_rs/src/main.rs:154-167,
// src/icao_filter.rs:8-9).  The multi-GPU form keeps exactly that shape for its caller: one handle, one filter,
// one ordered message list -- and inside, one context and one host thread per device:
//
//   submit    the capture is cut into contiguous buffer ranges, one per device; every device thread enqueues
//             phase 1 of its shard (adsb_shard.cpp: scan + the records of the self-validating hits) and goes
//             on polling -- nothing blocks, the shards of the next capture can be enqueued behind it;
//   exchange  the device thread that sees the LAST phase 1 of a capture land forms the union of the shards'
//             learned addresses in memory (minus what every device has been given since the last flush)
//             and hands every device thread phase 2 (set those addresses, match, records) -- captures in order;
//   collect   the caller's thread waits for the last phase 2 of the oldest capture and replays the shards'
//             records, device by device = in global (buffer, j, try_phase) order, through the ONE filter.
//
// No torch, no process group, no collective: the exchange is a few hundred u32 between threads of one process.
// Consecutive captures overlap (up to ADSB_MAX_IN_FLIGHT in flight: a slot of each context per capture), so the
// scans run back to back while the previous capture is exchanged, matched and replayed.
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <thread>

#include "adsb_ctx.h"

using namespace adsb::host;

namespace {

constexpr int kMultiSteps = ADSB_MAX_IN_FLIGHT;   // captures in flight (a slot of every context each)

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Cmd {
    enum Kind { kPhase1, kPhase2, kStop } kind;
    uint64_t step;
};

struct StepDev {   // one device's share of one capture
    const void *src = nullptr;          // the shard's samples on the device
    const int16_t *host_src = nullptr;  // ... or on the host (copied to the device's staging buffer first)
    uint64_t n_samples = 0, chunk_base = 0;
    std::vector<uint32_t> learned;
    const TrialRecord *rec = nullptr;
    size_t n_rec = 0;
    std::vector<TrialRecord> sorted;    // the shard's records in replay order (sorted by its device thread), when they were not
    // a shard the device scored (a dense stream's; adsb_shard.cpp): its messages (chunk already the capture's) and the
    // values its replay hands to icao_filter_add, in order -- in the slot's mapped memory
    bool scored = false;
    adsb_msg *msgs = nullptr;
    size_t n_msgs = 0, n_adds = 0, n_hits = 0;
    const uint32_t *adds = nullptr;
    ParallelReplay::Adders adders;      // the first record of the shard that can add each value (the device thread's pass over
    bool has_adders = false;            // its records, for shards of up to kDeviceThreadScanMax: the replay's scan stage, done)
    int rc = 0;
    adsb_stats st{};
    double t_p1_issue = 0, t_p1_done = 0, t_p2_issue = 0, t_p2_done = 0;
};

enum StepState : int { kFree = 0, kPhase1Out, kPhase2Out, kDone };

struct Step {
    std::atomic<uint64_t> id{0};
    std::atomic<int> state{kFree};
    bool flush_before = false;
    std::vector<StepDev> dev;
    std::atomic<int> p1_left{0}, p2_left{0};
    std::vector<uint32_t> fresh;   // what phase 2 hands every device
    std::vector<std::vector<uint32_t>> earlier;   // per device: what the shards BEFORE it add (for a scored shard: ScoreDev::earlier)
    uint64_t n_samples = 0;
    double t_submit = 0, t_exchange0 = 0, t_exchange1 = 0, t_done = 0;
};

// CPUs this process can really use: its affinity mask, cut by the cgroup's CPU quota where there is one (cpu.max)
int usable_cpus()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n > 0 ? n : CPU_COUNT(&set), CPU_COUNT(&set));
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
            n = std::min<long long>(n, std::max<long long>(1, quota / period));
        std::fclose(f);
    }
    return std::max(1, n);
}

// The thread of a device onto the host cores of that device's NUMA node (sysfs; best effort: a box without the
// entries, or a process already confined elsewhere, is left alone).  Eight threads that each spend a step in
// launches and polling must not pile onto one socket.
void pin_to_device_numa(int device)
{
    char bus[32] = {};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return;
    for (char &ch : bus)
        if (ch >= 'A' && ch <= 'F') ch = (char)(ch - 'A' + 'a');
    char path[128];
    std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = std::fopen(path, "r");
    if (!f) return;
    int node = -1;
    const int got = std::fscanf(f, "%d", &node);
    std::fclose(f);
    if (got != 1 || node < 0) return;
    std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    f = std::fopen(path, "r");
    if (!f) return;
    char list[4096] = {};
    const bool ok = std::fgets(list, sizeof(list), f) != nullptr;
    std::fclose(f);
    if (!ok) return;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    int n_want = 0;
    for (const char *s = list; *s;) {   // the kernel's cpulist format: "0-3,8,10-11"
        char *end = nullptr;
        const long a = std::strtol(s, &end, 10);
        if (end == s) break;
        long b = a;
        if (*end == '-') b = std::strtol(end + 1, &end, 10);
        for (long cpu = a; cpu <= b && cpu < CPU_SETSIZE; cpu++)
            if (cpu >= 0 && CPU_ISSET(cpu, &allowed)) {
                CPU_SET(cpu, &want);
                n_want++;
            }
        s = *end == ',' ? end + 1 : end;
        if (*end != ',') break;
    }
    if (n_want > 0) (void)sched_setaffinity(0, sizeof(want), &want);
}

// a device thread scans its own shard for first adders up to this many records (~5 ns each, between two polls of its
// device); a larger shard -- one device with a busy sky to itself -- is left to the pool's threads
constexpr size_t kDeviceThreadScanMax = 49152;
constexpr size_t kParallelReplayMin = 8192;   // records in a capture from which its replay is worth fanning out

}  // namespace

struct adsb_multi {
    struct Dev {
        int index = 0, device = 0;
        adsb_ctx *ctx = nullptr;
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Cmd> q;
        std::atomic<uint32_t> q_count{0};   // commands pushed so far (the thread spins on it without the lock)
        bool sleeping = false;              // under mu
        void *d_stage[kMultiSteps] = {};
        size_t stage_bytes[kMultiSteps] = {};
    };
    int n = 0;
    size_t max_chunks = 0;
    std::vector<std::unique_ptr<Dev>> dev;
    Step step[kMultiSteps];
    uint64_t submitted = 0, collected = 0;
    bool flush_pending = true;
    // the exchange: phase 2 is handed out capture by capture, by whichever thread saw the last phase 1 land
    std::mutex dispatch_mu;
    uint64_t next_p2 = 0;
    std::vector<uint32_t> known;   // sorted: what every device's superset has been given since the last flush
    std::mutex done_mu;
    std::condition_variable done_cv;
    IcaoFilter filter;
    Crc24 crc;
    adsb_multi_stats stats{};
    std::vector<adsb_msg> undelivered;
    std::vector<adsb_msg> msgs;   // the capture being collected (kept: a busy sky's list is megabytes, page by page when new)
    bool has_undelivered = false;
    std::string last_error;
    std::vector<void *> host_blocks;   // adsb_multi_host_alloc
    std::atomic<uint64_t> shards_sorted_on_host{0};
    std::unique_ptr<ReplayPool> pool;   // (created with the first capture that is large enough to want it)
    ParallelReplay parallel;
    uint64_t parallel_scored = 0, scored_shards_used = 0, scored_shards_refused = 0;
    size_t parallel_min = kParallelReplayMin;
    int score_mode = 0;   // 0: dense shards are scored on their devices; 1: never; 2: scored, and the result refused (tests)
    size_t held_before = 0;   // (collect_capture: values in the filter table when the capture's replay began)
#ifdef ADSB_TUNING
    double t_stage[5] = {};   // plan, scan, merge, score, finish (ADSB_HOST_TIMES=1: printed at destroy)
#endif
};

namespace {

void push_cmd(adsb_multi::Dev &d, Cmd c)
{
    bool wake;
    {
        std::lock_guard<std::mutex> lk(d.mu);
        d.q.push_back(c);
        d.q_count.fetch_add(1, std::memory_order_release);
        wake = d.sleeping;
    }
    if (wake) d.cv.notify_one();
}

// Hand phase 2 to every device, capture by capture in submission order (the known-address set and the bitmaps'
// flush rotation are histories): called by whoever saw a phase 1 land last.
void dispatch_ready(adsb_multi *m)
{
    std::lock_guard<std::mutex> lk(m->dispatch_mu);
    for (;;) {
        Step &s = m->step[m->next_p2 % kMultiSteps];
        if (s.state.load(std::memory_order_acquire) != kPhase1Out || s.id.load(std::memory_order_relaxed) != m->next_p2 ||
            s.p1_left.load(std::memory_order_acquire) != 0)
            return;
        s.t_exchange0 = now_s();
        if (s.flush_before) m->known.clear();   // the bitmaps this capture matches against started clean
        std::vector<const std::vector<uint32_t> *> lists;
        for (const StepDev &sd : s.dev) lists.push_back(&sd.learned);
        union_sorted(lists, m->known, s.fresh);
        if (!s.fresh.empty()) {
            std::vector<uint32_t> merged(m->known.size() + s.fresh.size());
            std::merge(m->known.begin(), m->known.end(), s.fresh.begin(), s.fresh.end(), merged.begin());
            m->known.swap(merged);
        }
        // (for the shards the devices score themselves: what is in the filter for every trial of shard k whoever adds it
        // first = the additions of shards 0 .. k - 1, whole lists, not only what is new to the devices)
        s.earlier.assign((size_t)m->n, {});
        for (int k = 1; k < m->n; k++) {
            const std::vector<uint32_t> &before = s.earlier[(size_t)k - 1], &add = s.dev[(size_t)k - 1].learned;
            std::vector<uint32_t> &u = s.earlier[(size_t)k];
            u.resize(before.size() + add.size());
            u.erase(std::set_union(before.begin(), before.end(), add.begin(), add.end(), u.begin()), u.end());
        }
        s.p2_left.store(m->n, std::memory_order_relaxed);
        s.state.store(kPhase2Out, std::memory_order_release);
        for (auto &d : m->dev) push_cmd(*d, Cmd{Cmd::kPhase2, m->next_p2});
        s.t_exchange1 = now_s();
        m->next_p2++;
    }
}

void device_thread(adsb_multi *m, adsb_multi::Dev *d)
{
    (void)hipSetDevice(d->device);
    pin_to_device_numa(d->device);
    adsb_ctx *c = d->ctx;
    std::deque<uint64_t> w1, w2;   // captures whose phase 1 / phase 2 is out on this device, oldest first
    uint32_t seen = 0;
#ifdef ADSB_TUNING
    // where this thread's time goes, printed when it ends (ADSB_HOST_TIMES=1): issuing the phases, reading the learned
    // addresses out of phase 1's records, taking phase 2's records (checksum), putting them in order
    struct Spent {
        double issue1 = 0, issue2 = 0, learned = 0, records = 0, sort = 0;
        uint64_t captures = 0, n_rec = 0;
        int index;
        ~Spent()
        {
            if (tuning_env("ADSB_HOST_TIMES") && captures)
                std::fprintf(stderr, "adsb_multi device thread %d: %llu captures, %.1f records each; us per capture: issue phase 1 %.1f, learned %.1f, "
                             "issue phase 2 %.1f, records %.1f, order %.1f\n", index, (unsigned long long)captures, (double)n_rec / captures,
                             issue1 / captures * 1e6, learned / captures * 1e6, issue2 / captures * 1e6, records / captures * 1e6, sort / captures * 1e6);
        }
    } spent;
    spent.index = d->index;
#define SPENT(field, expr) do { const double t_ = now_s(); expr; spent.field += now_s() - t_; } while (0)
#else
#define SPENT(field, expr) do { expr; } while (0)
#endif
    double last_progress = now_s();
    for (;;) {
        bool progressed = false;
        if (d->q_count.load(std::memory_order_acquire) != seen) {
            std::deque<Cmd> todo;
            {
                std::lock_guard<std::mutex> lk(d->mu);
                todo.swap(d->q);
                seen = d->q_count.load(std::memory_order_relaxed);
            }
            for (const Cmd &cmd : todo) {
                if (cmd.kind == Cmd::kStop) return;
                Step &s = m->step[cmd.step % kMultiSteps];
                StepDev &sd = s.dev[(size_t)d->index];
                const int k = (int)(cmd.step % kMultiSteps);
                if (cmd.kind == Cmd::kPhase1) {
                    sd.t_p1_issue = now_s();
                    c->flush_pending = s.flush_before;   // (set either way: a shard that failed before it began must not leave its flush to the next capture)
                    const void *src = sd.src;
                    if (sd.host_src && sd.n_samples) {
                        // the host-pointer form: the shard's samples into this capture's staging buffer, on the
                        // stream its scan will run on (shard_begin's own rule), in front of it
                        const size_t bytes = sd.n_samples * 4;
                        if (bytes > d->stage_bytes[k]) {
                            if (d->d_stage[k]) (void)hipFree(d->d_stage[k]);
                            d->d_stage[k] = nullptr;
                            d->stage_bytes[k] = 0;
                            if (hipMalloc(&d->d_stage[k], bytes) == hipSuccess) d->stage_bytes[k] = bytes;
                        }
                        if (d->stage_bytes[k] >= bytes &&
                            hipMemcpyAsync(d->d_stage[k], sd.host_src, bytes, hipMemcpyHostToDevice, c->scan_stream[c->shard_jobs % 2]) == hipSuccess)
                            src = d->d_stage[k];
                        else
                            sd.rc = ADSB_ERR_NOMEM;
                    }
                    if (sd.rc == ADSB_OK) SPENT(issue1, sd.rc = shard_begin(c, k, src, sd.n_samples, true));
                    w1.push_back(cmd.step);
                } else {
                    sd.t_p2_issue = now_s();
                    if (sd.rc == ADSB_OK) SPENT(issue2, sd.rc = shard_match(c, k, s.fresh.data(), s.fresh.size(), s.earlier[(size_t)d->index].data(), s.earlier[(size_t)d->index].size()));
                    w2.push_back(cmd.step);
                }
            }
            progressed = true;
        }
        if (!w1.empty()) {
            const uint64_t id = w1.front();
            const int k = (int)(id % kMultiSteps);
            Step &s = m->step[k];
            StepDev &sd = s.dev[(size_t)d->index];
            if (sd.rc != ADSB_OK || shard_phase_landed(c, k)) {
                if (sd.rc == ADSB_OK) SPENT(learned, sd.rc = shard_learned(c, k, sd.learned));
                if (sd.rc != ADSB_OK) sd.learned.clear();
                sd.t_p1_done = now_s();
                w1.pop_front();
                if (s.p1_left.fetch_sub(1, std::memory_order_acq_rel) == 1) dispatch_ready(m);
                progressed = true;
            }
        }
        if (!w2.empty()) {
            const uint64_t id = w2.front();
            const int k = (int)(id % kMultiSteps);
            Step &s = m->step[k];
            StepDev &sd = s.dev[(size_t)d->index];
            if (sd.rc != ADSB_OK || shard_phase_landed(c, k)) {
                if (sd.rc == ADSB_OK) SPENT(records, sd.rc = shard_records(c, k, &sd.rec, &sd.n_rec));
                if (sd.rc == ADSB_OK && c->shard[k].result_scored) {
                    sd.n_hits = c->stats.n_records;
                    if (shard_scored_result(c, k, &sd.msgs, &sd.n_msgs, &sd.adds, &sd.n_adds)) {
                        sd.scored = true;
                        for (size_t i = 0; i < sd.n_msgs; i++) sd.msgs[i].chunk += sd.chunk_base;
                    } else {
                        SPENT(records, sd.rc = shard_fetch_records(c, k, &sd.rec, &sd.n_rec));   // (a result that did not add up)
                    }
                }
                // (in replay order before they are handed over: the shards' sorts then run side by side, on the device
                // threads, instead of one after the other on the caller's)
                if (sd.rc == ADSB_OK && sd.n_rec > 1) SPENT(sort, if (sort_records(sd.rec, sd.n_rec, sd.sorted)) { sd.rec = sd.sorted.data(); m->shards_sorted_on_host.fetch_add(1, std::memory_order_relaxed); });
                if (sd.rc == ADSB_OK && !sd.scored && sd.n_rec <= kDeviceThreadScanMax) {
                    SPENT(sort, first_adders(c->crc, sd.rec, sd.n_rec, sd.adders));
                    sd.has_adders = true;
                }
#ifdef ADSB_TUNING
                spent.captures++;
                spent.n_rec += sd.n_rec;
#endif
                if (sd.rc != ADSB_OK) {
                    sd.rec = nullptr;
                    sd.n_rec = 0;
                    c->shard[k].active = c->shard[k].waiting = false;   // the slot is free again whatever happened
                }
                sd.st = c->stats;
                sd.t_p2_done = now_s();
                w2.pop_front();
                if (s.p2_left.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                    s.t_done = now_s();
                    {
                        std::lock_guard<std::mutex> lk(m->done_mu);
                        s.state.store(kDone, std::memory_order_release);
                    }
                    m->done_cv.notify_all();
                }
                progressed = true;
            }
        }
        if (progressed) {
            last_progress = now_s();
            continue;
        }
        const double idle = now_s() - last_progress;
        if (w1.empty() && w2.empty()) {
            // nothing out on the device: stay hot for a moment (the next capture of a pipelined caller is
            // microseconds away), then sleep until a command arrives
            if (idle < 200e-6) {
                __builtin_ia32_pause();
                continue;
            }
            std::unique_lock<std::mutex> lk(d->mu);
            d->sleeping = true;
            d->cv.wait(lk, [&] { return d->q_count.load(std::memory_order_relaxed) != seen; });
            d->sleeping = false;
            last_progress = now_s();
        } else if (idle < 5e-3) {
            __builtin_ia32_pause();
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(50));   // a long kernel (or a stuck one): stop burning a core
        }
    }
}

int submit_capture(adsb_multi *m, const void *const *device_iq, const int16_t *host_iq, bool host_form, const size_t *n_samples)
{
    if (m->submitted - m->collected >= (uint64_t)kMultiSteps) return ADSB_ERR_BUSY;
    // contiguous ranges: every shard but the last non-empty one is whole buffers, and none exceeds its context
    uint64_t base = 0, total = 0;
    bool ragged_seen = false;
    for (int k = 0; k < m->n; k++) {
        const size_t n = n_samples[k];
        if (n && ragged_seen) return ADSB_ERR_INVALID;
        if (n && !host_form && (!device_iq[k] || ((uintptr_t)device_iq[k] & 15u))) return ADSB_ERR_INVALID;
        if ((n + kChunkSamples - 1) / kChunkSamples > m->max_chunks) return ADSB_ERR_INVALID;
        if (n % kChunkSamples) ragged_seen = true;
        total += n;
    }
    Step &s = m->step[m->submitted % kMultiSteps];
    if (s.state.load(std::memory_order_acquire) != kFree) return ADSB_ERR_BUSY;
    s.flush_before = m->flush_pending;
    m->flush_pending = false;
    s.dev.assign((size_t)m->n, StepDev{});
    for (int k = 0; k < m->n; k++) {
        StepDev &sd = s.dev[(size_t)k];
        sd.n_samples = n_samples[k];
        sd.chunk_base = base;
        if (host_form) sd.host_src = sd.n_samples ? host_iq + 2 * (size_t)base * kChunkSamples : nullptr;
        else sd.src = device_iq[k];
        base += (n_samples[k] + kChunkSamples - 1) / kChunkSamples;
    }
    s.fresh.clear();
    s.n_samples = total;
    s.p1_left.store(m->n, std::memory_order_relaxed);
    s.t_submit = now_s();
    s.id.store(m->submitted, std::memory_order_relaxed);
    s.state.store(kPhase1Out, std::memory_order_release);
    for (auto &d : m->dev) push_cmd(*d, Cmd{Cmd::kPhase1, m->submitted});
    m->submitted++;
    return ADSB_OK;
}

// (direct: the caller's own array -- a capture scored by several threads goes straight into it when it fits, *direct_n then
// says how many messages it got and `out` stays empty)
int collect_capture(adsb_multi *m, std::vector<adsb_msg> &out, adsb_msg *direct = nullptr, size_t direct_cap = 0, size_t *direct_n = nullptr)
{
    if (m->collected == m->submitted) return ADSB_ERR_INVALID;
    Step &s = m->step[m->collected % kMultiSteps];
    {
        const double t0 = now_s();
        while (s.state.load(std::memory_order_acquire) != kDone) {
            __builtin_ia32_pause();
            if (now_s() - t0 > 2e-3) {
                std::unique_lock<std::mutex> lk(m->done_mu);
                m->done_cv.wait(lk, [&] { return s.state.load(std::memory_order_acquire) == kDone; });
            }
        }
    }
    int rc = ADSB_OK;
    adsb_multi_stats st{};
    st.n_samples = s.n_samples;
    st.n_devices = (uint32_t)m->n;
    st.n_addrs_exchanged = s.fresh.size();
    if (s.flush_before) m->filter.flush();   // icao_flush() took effect before this capture
    double p1_first = 0, p1_last = 0, p2_first = 0, p2_last = 0, p1_max = 0, p2_max = 0;
    bool any_scored = false;
    for (int k = 0; k < m->n; k++) {
        const StepDev &sd = s.dev[(size_t)k];
        if (sd.rc != ADSB_OK && rc == ADSB_OK) {
            rc = sd.rc;
            m->last_error = "device " + std::to_string(m->dev[(size_t)k]->device) + ": " + m->dev[(size_t)k]->ctx->last_error;
        }
        st.n_chunks += sd.st.n_chunks;
        st.n_candidates += sd.st.n_candidates;
        st.n_ap_entries += sd.st.n_ap_entries;
        st.n_records += sd.scored ? sd.n_hits : sd.n_rec;
        st.retries += sd.st.retries;
        any_scored = any_scored || sd.scored;
        p1_first = k ? std::min(p1_first, sd.t_p1_issue) : sd.t_p1_issue;
        p1_last = std::max(p1_last, sd.t_p1_done);
        p2_first = k ? std::min(p2_first, sd.t_p2_issue) : sd.t_p2_issue;
        p2_last = std::max(p2_last, sd.t_p2_done);
        p1_max = std::max(p1_max, sd.t_p1_done - sd.t_p1_issue);
        p2_max = std::max(p2_max, sd.t_p2_done - sd.t_p2_issue);
    }
    const double tr0 = now_s();
    bool direct_done = false;
    if (rc == ADSB_OK) {
        // the shards are contiguous ascending buffer ranges, each in replay order (its device thread saw to that):
        // device by device IS global (buffer, j, try_phase) order
        bool done = false;
        if (any_scored) {
            // Shards the devices scored themselves (a dense stream's): their messages and additions are taken as they are,
            // shard by shard in order; a shard that was not scored (sparse, overflowed, empty) is replayed here in its
            // place.  A scored shard's result stands on the filter being a SET -- k_score asks "was it there, or added
            // before me" -- which ends where the 4096-slot table could fill up (add() gives up, src/icao_filter.rs:46-62):
            // then its records are fetched from its device and replayed here like anyone's.
            size_t held = 0;
            for (uint32_t v : m->filter.table()) held += v != 0;
            m->held_before = held;
            const uint64_t before = m->filter.inserts();
            bool all_direct = direct && direct_n && out.empty();
            size_t total = 0;
            for (int k = 0; k < m->n && all_direct; k++) {
                const StepDev &sd = s.dev[(size_t)k];
                all_direct = sd.scored ? true : sd.n_rec == 0;
                total += sd.n_msgs;
                held += sd.n_adds;
            }
            all_direct = all_direct && total <= direct_cap && held + 64 < IcaoFilter::kSize && m->score_mode != 2;
            if (all_direct) {   // every shard scored: straight into the caller's array
                size_t at = 0;
                for (int k = 0; k < m->n; k++) {
                    const StepDev &sd = s.dev[(size_t)k];
                    if (sd.n_msgs) std::memcpy(direct + at, sd.msgs, sd.n_msgs * sizeof(adsb_msg));
                    at += sd.n_msgs;
                    for (size_t i = 0; i < sd.n_adds; i++) m->filter.add(sd.adds[i], IcaoFilter::hash(sd.adds[i] & 0xFFFFFFu));
                }
                *direct_n = total;
                direct_done = true;
                for (int k = 0; k < m->n; k++) m->scored_shards_used += s.dev[(size_t)k].scored ? 1u : 0u;
            } else {
                for (int k = 0; k < m->n && rc == ADSB_OK; k++) {
                    StepDev &sd = s.dev[(size_t)k];
                    const size_t now_held = (size_t)(m->filter.inserts() - before) + m->held_before;
                    if (sd.scored && now_held + sd.n_adds + 64 < IcaoFilter::kSize && m->score_mode != 2) {
                        out.insert(out.end(), sd.msgs, sd.msgs + sd.n_msgs);
                        for (size_t i = 0; i < sd.n_adds; i++) m->filter.add(sd.adds[i], IcaoFilter::hash(sd.adds[i] & 0xFFFFFFu));
                        m->scored_shards_used++;
                    } else if (sd.scored) {
                        // (the chunk offsets already put into the messages do not matter: they are dropped)
                        adsb_multi::Dev &d = *m->dev[(size_t)k];
                        DeviceGuard on_device(d.device);
                        const TrialRecord *rec = nullptr;
                        size_t n = 0;
                        rc = shard_fetch_records(d.ctx, (int)(m->collected % kMultiSteps), &rec, &n);
                        if (rc == ADSB_OK && n) replay(m->filter, m->crc, rec, n, sd.chunk_base, out);
                        m->scored_shards_refused++;
                    } else if (sd.n_rec) {
                        replay_sorted(m->filter, m->crc, sd.rec, sd.n_rec, sd.chunk_base, out);
                    }
                }
            }
            done = true;
        }
        if (!done && st.n_records >= m->parallel_min) {
            // a busy sky: tens of thousands of records -- scored by several threads at once, each record against the
            // filter as it was plus the positions at which the capture's new addresses enter it (adsb_replay_host.h)
            if (!m->pool) {
                const unsigned hw = std::thread::hardware_concurrency();
                std::vector<int> devs;
                for (auto &d : m->dev) devs.push_back(d->device);
                // six threads beside the caller, two per device from four devices on (the records to score grow with the
                // devices that found them), never more than a quarter of the host's cores
                int workers = (int)std::min<unsigned>(std::max(6u, std::min(16u, 2u * (unsigned)m->dev.size())), std::max(1u, hw / 4));
                // ... nor more than the CPUs this process may use leave beside the device threads and the caller (a
                // container's quota: threads beyond it only get the whole process throttled)
                const int room = usable_cpus() - (int)m->dev.size() - 2;
                workers = std::max(1, std::min(workers, room));
                if (const char *e = tuning_env("ADSB_POOL_WORKERS")) workers = std::max(1, std::atoi(e));   // (tuning build only)
                // (worker k on the host cores of devices[k % n]'s NUMA node: the records it reads sit in that node's memory)
                m->pool.reset(new ReplayPool(workers, [devs](int k) { pin_to_device_numa(devs[(size_t)k % devs.size()]); }));
            }
            std::vector<RecordRun> runs;
            std::vector<const ParallelReplay::Adders *> adders;
            for (int k = 0; k < m->n; k++) {
                const StepDev &sd = s.dev[(size_t)k];
                if (!sd.n_rec) continue;
                runs.push_back({sd.rec, sd.n_rec, sd.chunk_base});
                adders.push_back(sd.has_adders ? &sd.adders : nullptr);
            }
            ParallelReplay &pr = m->parallel;
#ifdef ADSB_TUNING
            double t[6] = {now_s()};
#define STAGE(k) t[k] = now_s()
#else
#define STAGE(k) (void)0
#endif
            if (pr.plan(m->filter, m->crc, runs, 4 * m->pool->threads(), true, &adders)) {
                STAGE(1);
                if (pr.scan_needed()) m->pool->run(pr, &ParallelReplay::scan_part);
                STAGE(2);
                if (pr.merge()) {
                    STAGE(3);
                    m->pool->run(pr, &ParallelReplay::score_part);
                    STAGE(4);
                    const size_t n_msgs = pr.message_count();
                    if (direct && direct_n && out.empty() && n_msgs <= direct_cap) {
                        pr.copy_to(direct);
                        m->pool->run(pr, &ParallelReplay::copy_part);
                        pr.apply_adds(m->filter);
                        *direct_n = n_msgs;
                        direct_done = true;
                    } else {
                        pr.finish(m->filter, out);
                    }
                    STAGE(5);
                    m->parallel_scored++;
                    done = true;
#ifdef ADSB_TUNING
                    for (int k = 0; k < 5; k++) m->t_stage[k] += t[k + 1] - t[k];
#endif
                }
            }
#undef STAGE
        }
        if (!done)
            for (int k = 0; k < m->n; k++) {
                const StepDev &sd = s.dev[(size_t)k];
                if (sd.n_rec) replay_sorted(m->filter, m->crc, sd.rec, sd.n_rec, sd.chunk_base, out);
            }
    }
    const double tr1 = now_s();
    st.n_messages = direct_done ? *direct_n : out.size();
    st.ms_wall = (float)((s.t_done - s.t_submit) * 1e3);
    st.ms_phase1_max = (float)(p1_max * 1e3);
    st.ms_phase2_max = (float)(p2_max * 1e3);
    st.ms_phase1_span = (float)((p1_last - p1_first) * 1e3);
    st.ms_phase2_span = (float)((p2_last - p2_first) * 1e3);
    st.ms_exchange = (float)((s.t_exchange1 - s.t_exchange0) * 1e3);
    st.ms_replay = (float)((tr1 - tr0) * 1e3);
    m->stats = st;
    s.state.store(kFree, std::memory_order_release);
    m->collected++;
    return rc;
}

int deliver_multi(adsb_multi *m, std::vector<adsb_msg> &msgs, adsb_msg *out, size_t cap, size_t *n_out)
{
    const size_t n = std::min(cap, msgs.size());
    if (n && out) std::memcpy(out, msgs.data(), n * sizeof(adsb_msg));
    if (n_out) *n_out = msgs.size();
    m->has_undelivered = msgs.size() > cap;
    if (!m->has_undelivered) {
        m->undelivered.clear();
        return ADSB_OK;
    }
    m->undelivered.swap(msgs);   // the capture is consumed and the filter has moved on: keep what it produced
    return ADSB_ERR_CAPACITY;
}

}  // namespace

extern "C" {

int adsb_multi_create(adsb_multi **out, const int *devices, int n_devices, size_t max_chunks_per_device)
{
    if (!out) return ADSB_ERR_INVALID;
    *out = nullptr;
    if (!devices || n_devices <= 0 || n_devices > 64) return ADSB_ERR_INVALID;
    if (max_chunks_per_device == 0) max_chunks_per_device = 1;
    adsb_multi *m = new (std::nothrow) adsb_multi;
    if (!m) return ADSB_ERR_NOMEM;
    m->n = n_devices;
    m->max_chunks = max_chunks_per_device;
    for (int k = 0; k < n_devices; k++) {
        auto d = std::make_unique<adsb_multi::Dev>();
        d->index = k;
        d->device = devices[k];
        // (created by a thread on the device's NUMA node: the context's pinned host memory -- where its records land and
        // the replay reads them -- is then that node's, whichever node the caller runs on)
        int rc = ADSB_ERR_HIP;
        std::thread([&] {
            pin_to_device_numa(devices[k]);
            rc = adsb_create(&d->ctx, devices[k], max_chunks_per_device);
        }).join();
        if (rc != ADSB_OK) {
            for (auto &e : m->dev) adsb_destroy(e->ctx);
            delete m;
            return rc;
        }
        // (the context's own per-pass timing is not read here: no events on the shards' streams)
        (void)adsb_set_profiling(d->ctx, 0);
        d->ctx->shard_scoring = true;   // (a dense stream's shards are scored on their devices: adsb_shard.cpp)
        m->dev.push_back(std::move(d));
    }
    for (auto &d : m->dev) d->th = std::thread(device_thread, m, d.get());
    *out = m;
    return ADSB_OK;
}

void adsb_multi_destroy(adsb_multi *m)
{
    if (!m) return;
    // what is still in flight is finished first (its kernels write into the contexts' memory)
    std::vector<adsb_msg> drop;
    while (m->collected < m->submitted) {
        drop.clear();
        (void)collect_capture(m, drop);
    }
#ifdef ADSB_TUNING
    if (tuning_env("ADSB_HOST_TIMES") && m->parallel_scored)
        std::fprintf(stderr, "adsb_multi parallel replay: %llu captures; us per capture: plan %.1f, scan %.1f, merge %.1f, score %.1f, finish %.1f\n",
                     (unsigned long long)m->parallel_scored, m->t_stage[0] / m->parallel_scored * 1e6, m->t_stage[1] / m->parallel_scored * 1e6,
                     m->t_stage[2] / m->parallel_scored * 1e6, m->t_stage[3] / m->parallel_scored * 1e6, m->t_stage[4] / m->parallel_scored * 1e6);
#endif
    for (auto &d : m->dev) push_cmd(*d, Cmd{Cmd::kStop, 0});
    for (auto &d : m->dev)
        if (d->th.joinable()) d->th.join();
    {
        DeviceGuard on_device(m->dev[0]->device);
        for (void *p : m->host_blocks) (void)hipHostFree(p);
    }
    for (auto &d : m->dev) {
        DeviceGuard on_device(d->device);
        for (void *p : d->d_stage)
            if (p) (void)hipFree(p);
        adsb_destroy(d->ctx);
    }
    delete m;
}

int adsb_multi_host_alloc(adsb_multi *m, size_t bytes, void **out)
{
    if (!m || !out || bytes == 0) return ADSB_ERR_INVALID;
    *out = nullptr;
    DeviceGuard on_device(m->dev[0]->device);
    void *p = nullptr;
    // pinned for every device of the process (portable): each device thread's copy of its range out of it is
    // a DMA over that device's own link
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        m->last_error = "hipHostMalloc (pinned, portable) failed";
        return ADSB_ERR_NOMEM;
    }
    m->host_blocks.push_back(p);
    *out = p;
    return ADSB_OK;
}

int adsb_multi_host_free(adsb_multi *m, void *p)
{
    if (!m || !p) return ADSB_ERR_INVALID;
    if (m->submitted != m->collected) return ADSB_ERR_BUSY;   // a capture in flight may still be read out of it
    auto it = std::find(m->host_blocks.begin(), m->host_blocks.end(), p);
    if (it == m->host_blocks.end()) return ADSB_ERR_INVALID;
    DeviceGuard on_device(m->dev[0]->device);
    (void)hipHostFree(p);
    m->host_blocks.erase(it);
    return ADSB_OK;
}

int adsb_multi_submit_iq(adsb_multi *m, const int16_t *iq_re_im, size_t n_samples)
{
    if (!m || !iq_re_im || n_samples == 0) return ADSB_ERR_INVALID;
    if ((n_samples + kChunkSamples - 1) / kChunkSamples > (size_t)m->n * m->max_chunks) return ADSB_ERR_INVALID;
    std::vector<size_t> n((size_t)m->n);
    for (int k = 0; k < m->n; k++) (void)adsb_multi_shard_range(n_samples, m->n, k, nullptr, &n[(size_t)k]);
    return submit_capture(m, nullptr, iq_re_im, true, n.data());
}

int adsb_multi_device_count(const adsb_multi *m) { return m ? m->n : 0; }
int adsb_multi_max_in_flight(const adsb_multi *m) { return m ? kMultiSteps : 0; }
int adsb_multi_pending(const adsb_multi *m) { return m ? (int)(m->submitted - m->collected) : 0; }

int adsb_multi_shard_range(size_t n_samples, int n_devices, int k, size_t *first_sample, size_t *n_samples_k)
{
    if (n_devices <= 0 || k < 0 || k >= n_devices) return ADSB_ERR_INVALID;
    // contiguous ranges of whole buffers whose sizes differ by at most one; the capture's ragged end
    // belongs to whoever holds its last buffer
    const uint64_t chunks = ((uint64_t)n_samples + kChunkSamples - 1) / kChunkSamples;
    const uint64_t base = chunks / (uint64_t)n_devices, extra = chunks % (uint64_t)n_devices;
    const uint64_t first = (uint64_t)k * base + std::min<uint64_t>((uint64_t)k, extra);
    const uint64_t last = first + base + ((uint64_t)k < extra ? 1 : 0);
    const uint64_t a = std::min<uint64_t>(first * kChunkSamples, n_samples), b = std::min<uint64_t>(last * kChunkSamples, n_samples);
    if (first_sample) *first_sample = (size_t)a;
    if (n_samples_k) *n_samples_k = (size_t)(b - a);
    return ADSB_OK;
}

int adsb_multi_icao_flush(adsb_multi *m)
{
    if (!m) return ADSB_ERR_INVALID;
    m->flush_pending = true;   // applies to the captures submitted after it, like adsb_icao_flush
    return ADSB_OK;
}

int adsb_multi_submit_iq_device(adsb_multi *m, const void *const *device_iq, const size_t *n_samples)
{
    if (!m || !device_iq || !n_samples) return ADSB_ERR_INVALID;
    return submit_capture(m, device_iq, nullptr, false, n_samples);
}

int adsb_multi_collect(adsb_multi *m, adsb_msg *out, size_t cap, size_t *n_out)
{
    if (!m || (!out && cap)) return ADSB_ERR_INVALID;
    m->msgs.clear();
    size_t direct_n = ~(size_t)0;
    if (int rc = collect_capture(m, m->msgs, out, cap, &direct_n)) return rc;
    if (direct_n != ~(size_t)0) {   // (the messages are in `out` already)
        if (n_out) *n_out = direct_n;
        m->has_undelivered = false;
        m->undelivered.clear();
        return ADSB_OK;
    }
    return deliver_multi(m, m->msgs, out, cap, n_out);
}

int adsb_multi_demod_iq_device(adsb_multi *m, const void *const *device_iq, const size_t *n_samples, adsb_msg *out,
                               size_t cap, size_t *n_out)
{
    if (!m || !device_iq || !n_samples || (!out && cap)) return ADSB_ERR_INVALID;
    if (m->submitted != m->collected) return ADSB_ERR_BUSY;
    if (int rc = submit_capture(m, device_iq, nullptr, false, n_samples)) return rc;
    m->msgs.clear();
    size_t direct_n = ~(size_t)0;
    if (int rc = collect_capture(m, m->msgs, out, cap, &direct_n)) return rc;
    if (direct_n != ~(size_t)0) {   // (the messages are in `out` already)
        if (n_out) *n_out = direct_n;
        m->has_undelivered = false;
        m->undelivered.clear();
        return ADSB_OK;
    }
    return deliver_multi(m, m->msgs, out, cap, n_out);
}

int adsb_multi_demod_iq(adsb_multi *m, const int16_t *iq_re_im, size_t n_samples, adsb_msg *out, size_t cap, size_t *n_out)
{
    if (!m || (!iq_re_im && n_samples) || (!out && cap)) return ADSB_ERR_INVALID;
    if (m->submitted != m->collected) return ADSB_ERR_BUSY;
    // a capture of any length: in pieces of at most what the devices' contexts hold together, each piece cut
    // into contiguous ranges (consecutive pieces are consecutive captures through the one filter)
    const size_t piece = (size_t)m->n * m->max_chunks * kChunkSamples;
    std::vector<adsb_msg> msgs;
    adsb_multi_stats total{};
    std::vector<size_t> n((size_t)m->n);
    for (size_t off = 0; off < n_samples || (off == 0 && n_samples == 0); off += piece) {
        const size_t len = std::min(piece, n_samples - off);
        for (int k = 0; k < m->n; k++) (void)adsb_multi_shard_range(len, m->n, k, nullptr, &n[(size_t)k]);
        if (int rc = submit_capture(m, nullptr, n_samples ? iq_re_im + 2 * off : nullptr, true, n.data())) return rc;
        std::vector<adsb_msg> part;
        if (int rc = collect_capture(m, part)) return rc;
        const uint64_t chunk0 = off / kChunkSamples;
        for (auto &msg : part) {
            msg.chunk += chunk0;
            msgs.push_back(msg);
        }
        total.n_samples += m->stats.n_samples, total.n_chunks += m->stats.n_chunks, total.n_candidates += m->stats.n_candidates;
        total.n_ap_entries += m->stats.n_ap_entries, total.n_records += m->stats.n_records, total.retries += m->stats.retries;
        total.n_addrs_exchanged += m->stats.n_addrs_exchanged, total.ms_wall += m->stats.ms_wall;
        total.ms_replay += m->stats.ms_replay, total.ms_exchange += m->stats.ms_exchange;
        total.ms_phase1_max += m->stats.ms_phase1_max, total.ms_phase2_max += m->stats.ms_phase2_max;
        total.ms_phase1_span += m->stats.ms_phase1_span, total.ms_phase2_span += m->stats.ms_phase2_span;
        if (n_samples == 0) break;
    }
    total.n_devices = (uint32_t)m->n;
    total.n_messages = msgs.size();
    m->stats = total;
    return deliver_multi(m, msgs, out, cap, n_out);
}

int adsb_multi_fetch_messages(adsb_multi *m, adsb_msg *out, size_t cap, size_t *n_out)
{
    if (!m || (!out && cap) || !m->has_undelivered) return ADSB_ERR_INVALID;
    const size_t n = std::min(cap, m->undelivered.size());
    if (n) std::memcpy(out, m->undelivered.data(), n * sizeof(adsb_msg));
    if (n_out) *n_out = m->undelivered.size();
    return m->undelivered.size() > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
}

int adsb_multi_get_stats(const adsb_multi *m, adsb_multi_stats *out)
{
    if (!m || !out) return ADSB_ERR_INVALID;
    *out = m->stats;
    return ADSB_OK;
}

int adsb_multi_filter_table(const adsb_multi *m, uint32_t *out4096)
{
    if (!m || !out4096) return ADSB_ERR_INVALID;
    if (m->submitted != m->collected) return ADSB_ERR_BUSY;
    m->filter.store(out4096);
    return ADSB_OK;
}

int adsb_multi_selftest_tune(adsb_multi *m, uint32_t fresh_cap, uint32_t parallel_min, uint32_t score_mode)
{
    if (!m || score_mode > 2) return ADSB_ERR_INVALID;
    if (m->submitted != m->collected) return ADSB_ERR_BUSY;
    for (auto &d : m->dev) {   // (the device threads are idle: nothing in flight)
        d->ctx->shard_fresh_cap = fresh_cap;
        d->ctx->shard_scoring = score_mode != 1;
    }
    m->parallel_min = parallel_min ? parallel_min : kParallelReplayMin;
    m->score_mode = (int)score_mode;
    return ADSB_OK;
}

int adsb_multi_selftest_counters(const adsb_multi *m, uint64_t *out8)
{
    if (!m || !out8) return ADSB_ERR_INVALID;
    if (m->submitted != m->collected) return ADSB_ERR_BUSY;
    for (int k = 0; k < 8; k++) out8[k] = 0;
    out8[0] = m->shards_sorted_on_host.load(std::memory_order_relaxed);
    for (auto &d : m->dev) {
        out8[1] += d->ctx->shard_fresh_fallbacks;
        out8[2] += d->ctx->shard_device_ordered;
        out8[4] += d->ctx->shard_device_scored;
    }
    out8[3] = m->parallel_scored;
    out8[5] = m->scored_shards_used;
    out8[6] = m->scored_shards_refused;
    return ADSB_OK;
}

const char *adsb_multi_last_error(const adsb_multi *m) { return m ? m->last_error.c_str() : ""; }

}  // extern "C"
