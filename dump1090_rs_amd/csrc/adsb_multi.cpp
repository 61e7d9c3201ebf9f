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
//
// Failure.  Every wait is bounded: a device thread that has not seen a phase land for 2 ms asks the phase's streams
// (shard_phase_check), a stream error fails the shard, and a phase that is still out after the handle's timeout
// (adsb_multi_set_timeout_ms, 30 s) fails it and marks the device dead (nothing more is enqueued on it; its context is
// leaked at destroy rather than waited for).  A capture with a failed shard returns the shard's status from collect and
// POISONS the handle: the one filter, `known` and the devices' supersets have missed that capture's additions, so
// nothing computed after it would be the single stream's -- captures in flight behind it and later submissions return
// ADSB_ERR_POISONED until adsb_multi_icao_flush (nothing in flight) resets every context (shard_reset), the filter and
// `known`, and the stream starts over from an empty filter.  No exception leaves the library: the state machine itself
// (command rings, phase counters) never allocates, every other step runs under a catch that fails the shard or the
// capture, and every extern "C" body is wrapped (abi_guard).
// The host side is x86-64 only (the spin loops are `pause`).
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
    enum Kind { kPhase1, kPhase2, kFetch, kReset, kStop } kind;
    uint64_t step;
};

// A timed wait on a condition variable.  (std::condition_variable::wait_for is pthread_cond_clockwait, which the
// ThreadSanitizer runtime of GCC 11 does not know: it then believes the mutex is still held and reports every later lock
// as a double lock.  Under the sanitizer -- tests/test_multi_orchestration.py -- the wait goes by the system clock.)
template <class Dur, class Pred>
bool timed_wait(std::condition_variable &cv, std::unique_lock<std::mutex> &lk, Dur d, Pred pred)
{
#ifdef __SANITIZE_THREAD__
    return cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::microseconds>(d), pred);
#else
    return cv.wait_for(lk, d, pred);
#endif
}

// what every extern "C" body runs under (include/adsb_hip.h: nothing is thrown across the ABI)
template <class F>
int abi_guard(F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return ADSB_ERR_NOMEM;
    } catch (...) {
        return ADSB_ERR_HIP;
    }
}

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
    std::string error;                  // what failed, when rc says something did (the context's last_error moves on)
    adsb_stats st{};
    double t_p1_issue = 0, t_p1_done = 0, t_p2_issue = 0, t_p2_done = 0;
    // the collector asks the shard's device thread for the records of a scored shard it cannot use (Cmd::kFetch)
    std::atomic<int> fetched{0};
    int fetch_rc = 0;
    void reset()   // (keeps the vectors' storage: a steady stream allocates nothing per capture)
    {
        src = nullptr, host_src = nullptr, n_samples = chunk_base = 0;
        learned.clear(), sorted.clear(), adders.clear(), error.clear();
        rec = nullptr, n_rec = 0, scored = false, msgs = nullptr, n_msgs = n_adds = n_hits = 0, adds = nullptr;
        has_adders = false, rc = 0, st = adsb_stats{}, t_p1_issue = t_p1_done = t_p2_issue = t_p2_done = 0;
        fetched.store(0, std::memory_order_relaxed), fetch_rc = 0;
    }
};

enum StepState : int { kFree = 0, kPhase1Out, kPhase2Out, kDone };

struct Step {
    std::atomic<uint64_t> id{0};
    std::atomic<int> state{kFree};
    bool flush_before = false;
    std::unique_ptr<StepDev[]> dev;   // one per device, made at create
    int exchange_rc = 0;              // the address union failed (out of memory): every shard of the capture fails
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
constexpr double kCheckAfterS = 2e-3;         // a phase that has been out this long without landing: ask its streams
constexpr uint32_t kDefaultTimeoutMs = 30000; // ... and this long: the device is given up (adsb_multi_set_timeout_ms)

// a ring of commands that never allocates (at most a phase-1, a phase-2 and a fetch per capture in flight, a reset, a stop)
struct CmdRing {
    static constexpr uint32_t kCap = 32;
    Cmd slot[kCap];
    uint32_t head = 0, tail = 0;   // under the device's mutex
    bool empty() const { return head == tail; }
    bool push(Cmd c)
    {
        if (tail - head == kCap) return false;
        slot[tail++ % kCap] = c;
        return true;
    }
    Cmd pop() { return slot[head++ % kCap]; }
};

// captures whose phase is out on a device, oldest first (at most kMultiSteps of them)
struct StepFifo {
    uint64_t id[kMultiSteps] = {};
    int head = 0, n = 0;
    bool empty() const { return n == 0; }
    uint64_t front() const { return id[head]; }
    void push(uint64_t v) { id[(head + n++) % kMultiSteps] = v; }
    void pop() { head = (head + 1) % kMultiSteps, n--; }
};

}  // namespace

struct adsb_multi {
    struct Dev {
        int index = 0, device = 0;
        adsb_ctx *ctx = nullptr;
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        CmdRing q;
        std::atomic<uint32_t> q_count{0};   // commands pushed so far (the thread spins on it without the lock)
        bool sleeping = false;              // under mu
        void *d_stage[kMultiSteps] = {};    // the host-pointer form's staging buffers: max_chunks buffers each, on first use
        std::atomic<bool> dead{false};      // a phase timed out on it: nothing more is enqueued, its context is not waited for
        std::atomic<int> reset_done{0};     // Cmd::kReset: 0 pending, 1 done (reset_rc)
        int reset_rc = 0;
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
    // a capture failed: see the head of this file
    bool poisoned = false;
    std::string poison_error;
    // how the threads wait (adsb_multi_set_wait): `block` is what the device threads and the collector read
    int wait_setting = ADSB_WAIT_AUTO;
    std::atomic<bool> block{false};
    std::atomic<uint32_t> timeout_ms{kDefaultTimeoutMs};
    // adsb_multi_selftest_fail: capture number, shard, kind (0: none)
    std::atomic<uint64_t> fault_capture{~0ull};
    std::atomic<int> fault_shard{-1}, fault_kind{0};
#ifdef ADSB_TUNING
    double t_stage[5] = {};   // plan, scan, merge, score, finish (ADSB_HOST_TIMES=1: printed at destroy)
#endif
};

namespace {

// (noexcept: the ring holds every command a device can be owed; a full ring would be a bug in the counts above)
void push_cmd(adsb_multi::Dev &d, Cmd c) noexcept
{
    bool wake;
    {
        std::lock_guard<std::mutex> lk(d.mu);
        if (!d.q.push(c)) std::abort();   // unreachable: see CmdRing
        d.q_count.fetch_add(1, std::memory_order_release);
        wake = d.sleeping;
    }
    if (wake) d.cv.notify_one();
}

int fault_for(const adsb_multi *m, uint64_t capture, int shard)
{
    if (m->fault_capture.load(std::memory_order_relaxed) != capture || m->fault_shard.load(std::memory_order_relaxed) != shard) return 0;
    return m->fault_kind.load(std::memory_order_relaxed);
}

// Hand phase 2 to every device, capture by capture in submission order (the known-address set and the bitmaps'
// flush rotation are histories): called by whoever saw a phase 1 land last.
void dispatch_ready(adsb_multi *m) noexcept
{
    std::lock_guard<std::mutex> lk(m->dispatch_mu);
    for (;;) {
        Step &s = m->step[m->next_p2 % kMultiSteps];
        if (s.state.load(std::memory_order_acquire) != kPhase1Out || s.id.load(std::memory_order_relaxed) != m->next_p2 ||
            s.p1_left.load(std::memory_order_acquire) != 0)
            return;
        s.t_exchange0 = now_s();
        s.exchange_rc = ADSB_OK;
        try {
            if (s.flush_before) m->known.clear();   // the bitmaps this capture matches against started clean
            std::vector<const std::vector<uint32_t> *> lists;
            for (int k = 0; k < m->n; k++) lists.push_back(&s.dev[k].learned);
            union_sorted(lists, m->known, s.fresh);
            if (!s.fresh.empty()) {
                std::vector<uint32_t> merged(m->known.size() + s.fresh.size());
                std::merge(m->known.begin(), m->known.end(), s.fresh.begin(), s.fresh.end(), merged.begin());
                m->known.swap(merged);
            }
            // (for the shards the devices score themselves: what is in the filter for every trial of shard k whoever adds it
            // first = the additions of shards 0 .. k - 1, whole lists, not only what is new to the devices)
            s.earlier.resize((size_t)m->n);
            for (auto &e : s.earlier) e.clear();
            for (int k = 1; k < m->n; k++) {
                const std::vector<uint32_t> &before = s.earlier[(size_t)k - 1], &add = s.dev[k - 1].learned;
                std::vector<uint32_t> &u = s.earlier[(size_t)k];
                u.resize(before.size() + add.size());
                u.erase(std::set_union(before.begin(), before.end(), add.begin(), add.end(), u.begin()), u.end());
            }
        } catch (...) {
            s.exchange_rc = ADSB_ERR_NOMEM;   // (every shard of the capture fails in its second phase; the handle is poisoned at collect)
        }
        s.p2_left.store(m->n, std::memory_order_relaxed);
        s.state.store(kPhase2Out, std::memory_order_release);
        for (auto &d : m->dev) push_cmd(*d, Cmd{Cmd::kPhase2, m->next_p2});
        s.t_exchange1 = now_s();
        m->next_p2++;
    }
}

void shard_failed(adsb_multi::Dev *d, StepDev &sd, int rc, const char *what = nullptr)
{
    if (sd.rc == ADSB_OK) sd.rc = rc;
    try {
        if (sd.error.empty()) sd.error = what ? std::string(what) : d->ctx->last_error;
    } catch (...) {
    }
}

void device_thread(adsb_multi *m, adsb_multi::Dev *d) noexcept
{
    (void)hipSetDevice(d->device);
    pin_to_device_numa(d->device);
    adsb_ctx *c = d->ctx;
    StepFifo w1, w2;   // captures whose phase 1 / phase 2 is out on this device, oldest first
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
    // phase 1 of a shard: its samples (copied to this capture's staging buffer first, in the host-pointer form) and its scan
    auto issue1 = [&](uint64_t id) {
        Step &s = m->step[id % kMultiSteps];
        StepDev &sd = s.dev[d->index];
        const int k = (int)(id % kMultiSteps);
        sd.t_p1_issue = now_s();
        try {
            const int fault = fault_for(m, id, d->index);
            if (d->dead.load(std::memory_order_relaxed)) {
                shard_failed(d, sd, ADSB_ERR_HIP, "the device stopped answering earlier (a shard phase timed out): destroy the adsb_multi");
            } else if (fault == ADSB_FAULT_PHASE1) {
                shard_failed(d, sd, ADSB_ERR_HIP, "injected: phase 1 failed (adsb_multi_selftest_fail)");
            } else {
                c->flush_pending = s.flush_before;   // (set either way: a shard that failed before it began must not leave its flush to the next capture)
                const void *src = sd.src;
                if (sd.host_src && sd.n_samples) {
                    // the host-pointer form: the shard's samples into this capture's staging buffer, on the stream its scan
                    // will run on (shard_begin's own rule), in front of it.  The buffer holds the largest shard the
                    // context takes and is made once (hipFree would wait for everything in flight on the device)
                    const size_t bytes = sd.n_samples * 4;
                    if (!d->d_stage[k] && hipMalloc(&d->d_stage[k], m->max_chunks * (size_t)kChunkSamples * 4) != hipSuccess) {
                        (void)hipGetLastError();
                        d->d_stage[k] = nullptr;
                        shard_failed(d, sd, ADSB_ERR_NOMEM, "hipMalloc of a staging buffer for host samples failed");
                    } else if (hipMemcpyAsync(d->d_stage[k], sd.host_src, bytes, hipMemcpyHostToDevice, c->scan_stream[c->shard_jobs % 2]) != hipSuccess) {
                        (void)hipGetLastError();
                        shard_failed(d, sd, ADSB_ERR_HIP, "hipMemcpyAsync of a shard's host samples failed");
                    } else {
                        src = d->d_stage[k];
                    }
                }
                if (sd.rc == ADSB_OK) {
                    int rc = ADSB_OK;
                    SPENT(issue1, rc = shard_begin(c, k, src, sd.n_samples, true));
                    if (rc != ADSB_OK) shard_failed(d, sd, rc);
                }
            }
        } catch (...) {
            shard_failed(d, sd, ADSB_ERR_NOMEM, "out of memory while a shard's first phase was enqueued");
        }
        w1.push(id);
    };
    auto issue2 = [&](uint64_t id) {
        Step &s = m->step[id % kMultiSteps];
        StepDev &sd = s.dev[d->index];
        const int k = (int)(id % kMultiSteps);
        sd.t_p2_issue = now_s();
        try {
            if (s.exchange_rc != ADSB_OK) shard_failed(d, sd, s.exchange_rc, "out of memory in the address exchange");
            else if (sd.rc == ADSB_OK && fault_for(m, id, d->index) == ADSB_FAULT_PHASE2)
                shard_failed(d, sd, ADSB_ERR_HIP, "injected: phase 2 failed (adsb_multi_selftest_fail)");
            if (sd.rc == ADSB_OK) {
                int rc = ADSB_OK;
                SPENT(issue2, rc = shard_match(c, k, s.fresh.data(), s.fresh.size(), s.earlier[(size_t)d->index].data(), s.earlier[(size_t)d->index].size()));
                if (rc != ADSB_OK) shard_failed(d, sd, rc);
            }
        } catch (...) {
            shard_failed(d, sd, ADSB_ERR_NOMEM, "out of memory while a shard's second phase was enqueued");
        }
        w2.push(id);
    };
    // has the phase at the front of `w` landed (true: also when it failed)?  A phase that has been out for a while is
    // looked into -- its streams' status -- and given up after the handle's timeout.
    double last_check = 0;
    auto landed = [&](uint64_t id, double issued_at) -> bool {
        Step &s = m->step[id % kMultiSteps];
        StepDev &sd = s.dev[d->index];
        const int k = (int)(id % kMultiSteps);
        if (sd.rc != ADSB_OK) return true;
        if (d->dead.load(std::memory_order_relaxed)) {   // (given up on while this phase was out)
            shard_failed(d, sd, ADSB_ERR_HIP, "the device stopped answering (an earlier shard phase timed out)");
            return true;
        }
        const bool hang = fault_for(m, id, d->index) == ADSB_FAULT_HANG;   // (the phase lands, the thread pretends it never does)
        if (!hang && shard_phase_landed(c, k)) return true;
        const double t = now_s();
        if (t - issued_at < kCheckAfterS || t - last_check < kCheckAfterS) return false;
        last_check = t;
        const int st = hang ? 0 : shard_phase_check(c, k);
        if (st > 0) return true;
        if (st < 0) {
            shard_failed(d, sd, st);
            return true;
        }
        if ((t - issued_at) * 1e3 > (double)m->timeout_ms.load(std::memory_order_relaxed)) {
            d->dead.store(true, std::memory_order_relaxed);
            shard_failed(d, sd, ADSB_ERR_HIP, "a shard phase did not finish within the adsb_multi's timeout: the device is given up");
            return true;
        }
        return false;
    };
    auto land1 = [&]() {
        const uint64_t id = w1.front();
        const int k = (int)(id % kMultiSteps);
        Step &s = m->step[k];
        StepDev &sd = s.dev[d->index];
        try {
            if (sd.rc == ADSB_OK) {
                int rc = ADSB_OK;
                SPENT(learned, rc = shard_learned(c, k, sd.learned));
                if (rc != ADSB_OK) shard_failed(d, sd, rc);
            }
        } catch (...) {
            shard_failed(d, sd, ADSB_ERR_NOMEM, "out of memory while a shard's learned addresses were read");
        }
        if (sd.rc != ADSB_OK) sd.learned.clear();
        sd.t_p1_done = now_s();
        w1.pop();
        if (s.p1_left.fetch_sub(1, std::memory_order_acq_rel) == 1) dispatch_ready(m);
    };
    auto land2 = [&]() {
        const uint64_t id = w2.front();
        const int k = (int)(id % kMultiSteps);
        Step &s = m->step[k];
        StepDev &sd = s.dev[d->index];
        try {
            if (sd.rc == ADSB_OK) {
                int rc = ADSB_OK;
                SPENT(records, rc = shard_records(c, k, &sd.rec, &sd.n_rec));
                if (rc == ADSB_OK && fault_for(m, id, d->index) == ADSB_FAULT_RECORDS) {
                    rc = ADSB_ERR_HIP;
                    c->last_error = "injected: the shard's records did not add up (adsb_multi_selftest_fail)";
                }
                if (rc != ADSB_OK) shard_failed(d, sd, rc);
            }
            if (sd.rc == ADSB_OK && c->shard[k].result_scored) {
                sd.n_hits = c->stats.n_records;
                if (shard_scored_result(c, k, &sd.msgs, &sd.n_msgs, &sd.adds, &sd.n_adds)) {
                    sd.scored = true;
                    for (size_t i = 0; i < sd.n_msgs; i++) sd.msgs[i].chunk += sd.chunk_base;
                } else {
                    int rc = ADSB_OK;
                    SPENT(records, rc = shard_fetch_records(c, k, &sd.rec, &sd.n_rec));   // (a result that did not add up)
                    if (rc != ADSB_OK) shard_failed(d, sd, rc);
                }
            }
            // (in replay order before they are handed over: the shards' sorts then run side by side, on the device
            // threads, instead of one after the other on the caller's)
            if (sd.rc == ADSB_OK && sd.n_rec > 1) SPENT(sort, if (sort_records(sd.rec, sd.n_rec, sd.sorted)) { sd.rec = sd.sorted.data(); m->shards_sorted_on_host.fetch_add(1, std::memory_order_relaxed); });
            if (sd.rc == ADSB_OK && !sd.scored && sd.n_rec <= kDeviceThreadScanMax) {
                SPENT(sort, first_adders(c->crc, sd.rec, sd.n_rec, sd.adders));
                sd.has_adders = true;
            }
        } catch (...) {
            shard_failed(d, sd, ADSB_ERR_NOMEM, "out of memory while a shard's records were taken");
        }
#ifdef ADSB_TUNING
        spent.captures++;
        spent.n_rec += sd.n_rec;
#endif
        if (sd.rc != ADSB_OK) {
            sd.rec = nullptr;
            sd.n_rec = 0;
            sd.scored = false;
            c->shard[k].active = c->shard[k].waiting = false;   // the slot is free again whatever happened (shard_reset puts the rest right)
        }
        sd.st = c->stats;
        sd.t_p2_done = now_s();
        w2.pop();
        if (s.p2_left.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            s.t_done = now_s();
            {
                std::lock_guard<std::mutex> lk(m->done_mu);
                s.state.store(kDone, std::memory_order_release);
            }
            m->done_cv.notify_all();
        }
    };
    double last_progress = now_s();
    for (;;) {
        bool progressed = false;
        if (d->q_count.load(std::memory_order_acquire) != seen) {
            Cmd todo[CmdRing::kCap];
            int n_todo = 0;
            {
                std::lock_guard<std::mutex> lk(d->mu);
                while (!d->q.empty()) todo[n_todo++] = d->q.pop();
                seen = d->q_count.load(std::memory_order_relaxed);
            }
            for (int i = 0; i < n_todo; i++) {
                const Cmd &cmd = todo[i];
                if (cmd.kind == Cmd::kStop) return;
                if (cmd.kind == Cmd::kPhase1) issue1(cmd.step);
                else if (cmd.kind == Cmd::kPhase2) issue2(cmd.step);
                else if (cmd.kind == Cmd::kFetch) {
                    // a scored shard whose result the collector cannot use (a filter table about to fill up): its records
                    // out of device memory, by this thread -- the context is this thread's while captures are in flight
                    StepDev &sd = m->step[cmd.step % kMultiSteps].dev[d->index];
                    try {
                        sd.fetch_rc = d->dead.load(std::memory_order_relaxed) ? (int)ADSB_ERR_HIP
                                                                             : shard_fetch_records(c, (int)(cmd.step % kMultiSteps), &sd.rec, &sd.n_rec);
                        if (sd.fetch_rc != ADSB_OK) shard_failed(d, sd, sd.fetch_rc);
                    } catch (...) {
                        sd.fetch_rc = ADSB_ERR_NOMEM;
                    }
                    {
                        std::lock_guard<std::mutex> lk(m->done_mu);
                        sd.fetched.store(1, std::memory_order_release);
                    }
                    m->done_cv.notify_all();
                } else if (cmd.kind == Cmd::kReset) {
                    try {
                        d->reset_rc = d->dead.load(std::memory_order_relaxed) ? (int)ADSB_ERR_HIP : shard_reset(c);
                    } catch (...) {
                        d->reset_rc = ADSB_ERR_NOMEM;
                    }
                    {
                        std::lock_guard<std::mutex> lk(m->done_mu);
                        d->reset_done.store(1, std::memory_order_release);
                    }
                    m->done_cv.notify_all();
                }
            }
            progressed = true;
        }
        if (!w1.empty() && landed(w1.front(), m->step[w1.front() % kMultiSteps].dev[d->index].t_p1_issue)) {
            land1();
            progressed = true;
        }
        if (!w2.empty() && landed(w2.front(), m->step[w2.front() % kMultiSteps].dev[d->index].t_p2_issue)) {
            land2();
            progressed = true;
        }
        if (progressed) {
            last_progress = now_s();
            continue;
        }
        const double idle = now_s() - last_progress;
        const bool block = m->block.load(std::memory_order_relaxed);
        if (w1.empty() && w2.empty()) {
            // nothing out on the device: stay hot for a moment (the next capture of a pipelined caller is
            // microseconds away; not in the blocking mode), then sleep until a command arrives
            if (!block && idle < 200e-6) {
                __builtin_ia32_pause();
                continue;
            }
            std::unique_lock<std::mutex> lk(d->mu);
            d->sleeping = true;
            d->cv.wait(lk, [&] { return d->q_count.load(std::memory_order_relaxed) != seen; });
            d->sleeping = false;
            last_progress = now_s();
        } else if (block) {
            // ADSB_WAIT_BLOCK: asleep between looks at the device -- a timed wait on the command queue, so a command
            // still wakes the thread at once.  25 us while a phase is young (a shard's scan is ~100 us), longer as it ages.
            const auto nap = std::chrono::microseconds(idle < 1e-3 ? 25 : (idle < 5e-3 ? 100 : 1000));
            std::unique_lock<std::mutex> lk(d->mu);
            d->sleeping = true;
            timed_wait(d->cv, lk, nap, [&] { return d->q_count.load(std::memory_order_relaxed) != seen; });
            d->sleeping = false;
        } else if (idle < 5e-3) {
            __builtin_ia32_pause();
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(50));   // a long kernel (or a stuck one): stop burning a core
        }
    }
}

int refuse_poisoned(adsb_multi *m)
{
    m->last_error = "an earlier capture failed (" + m->poison_error + "): adsb_multi_icao_flush, with nothing in flight, starts the stream over from an empty filter";
    return ADSB_ERR_POISONED;
}

int submit_capture(adsb_multi *m, const void *const *device_iq, const int16_t *host_iq, bool host_form, const size_t *n_samples)
{
    if (m->poisoned) return refuse_poisoned(m);
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
    for (int k = 0; k < m->n; k++) {
        StepDev &sd = s.dev[k];
        sd.reset();
        sd.n_samples = n_samples[k];
        sd.chunk_base = base;
        if (host_form) sd.host_src = sd.n_samples ? host_iq + 2 * (size_t)base * kChunkSamples : nullptr;
        else sd.src = device_iq[k];
        base += (n_samples[k] + kChunkSamples - 1) / kChunkSamples;
    }
    s.fresh.clear();
    s.exchange_rc = ADSB_OK;
    s.n_samples = total;
    s.p1_left.store(m->n, std::memory_order_relaxed);
    s.t_submit = now_s();
    s.id.store(m->submitted, std::memory_order_relaxed);
    s.state.store(kPhase1Out, std::memory_order_release);
    for (auto &d : m->dev) push_cmd(*d, Cmd{Cmd::kPhase1, m->submitted});
    m->submitted++;
    return ADSB_OK;
}

// wait (spin first unless the handle blocks) until `ready` says so; false: the handle's timeout -- twice over, the device
// threads give up first -- has passed
template <class Pred>
bool wait_done(adsb_multi *m, Pred ready)
{
    const double t0 = now_s();
    if (!m->block.load(std::memory_order_relaxed))
        while (!ready() && now_s() - t0 < 2e-3) __builtin_ia32_pause();
    if (ready()) return true;
    std::unique_lock<std::mutex> lk(m->done_mu);
    const auto limit = std::chrono::milliseconds(2 * (uint64_t)m->timeout_ms.load(std::memory_order_relaxed) + 1000);
    return timed_wait(m->done_cv, lk, limit, ready);
}

// the replay of a capture whose shards all arrived (rc == ADSB_OK); may throw (out of memory): the caller poisons the handle then
void replay_capture(adsb_multi *m, Step &s, adsb_multi_stats &st, bool any_scored, std::vector<adsb_msg> &out, adsb_msg *direct,
                    size_t direct_cap, size_t *direct_n, bool &direct_done, int &rc)
{
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
        // how many slots a shard's additions can take: the list has one entry per add() CALL (a DF18 whose plain address is
        // unknown re-adds address | NT on every frame and phase), the table one per distinct value -- counted only when the
        // cheap bound (every call a new value) says the table might fill up
        std::vector<uint32_t> scratch;
        auto fits = [&](size_t now_held, const StepDev &sd) {
            if (now_held + sd.n_adds + 64 < IcaoFilter::kSize) return true;
            scratch.assign(sd.adds, sd.adds + sd.n_adds);
            std::sort(scratch.begin(), scratch.end());
            const size_t distinct = (size_t)(std::unique(scratch.begin(), scratch.end()) - scratch.begin());
            return now_held + distinct + 64 < IcaoFilter::kSize;
        };
        bool all_direct = direct && direct_n && out.empty() && m->score_mode != 2;
        size_t total = 0;
        for (int k = 0; k < m->n && all_direct; k++) {
            const StepDev &sd = s.dev[k];
            all_direct = sd.scored ? fits(held, sd) : sd.n_rec == 0;
            total += sd.n_msgs;
            held += sd.n_adds;   // (the cheap bound again for the shards behind: a capture this close to a full table is rare)
        }
        all_direct = all_direct && total <= direct_cap;
        if (all_direct) {   // every shard scored: straight into the caller's array
            size_t at = 0;
            for (int k = 0; k < m->n; k++) {
                const StepDev &sd = s.dev[k];
                if (sd.n_msgs) std::memcpy(direct + at, sd.msgs, sd.n_msgs * sizeof(adsb_msg));
                at += sd.n_msgs;
                for (size_t i = 0; i < sd.n_adds; i++) m->filter.add(sd.adds[i], IcaoFilter::hash(sd.adds[i] & 0xFFFFFFu));
            }
            *direct_n = total;
            direct_done = true;
            for (int k = 0; k < m->n; k++) m->scored_shards_used += s.dev[k].scored ? 1u : 0u;
        } else {
            for (int k = 0; k < m->n && rc == ADSB_OK; k++) {
                StepDev &sd = s.dev[k];
                const size_t now_held = (size_t)(m->filter.inserts() - before) + m->held_before;
                if (sd.scored && m->score_mode != 2 && fits(now_held, sd)) {
                    out.insert(out.end(), sd.msgs, sd.msgs + sd.n_msgs);
                    for (size_t i = 0; i < sd.n_adds; i++) m->filter.add(sd.adds[i], IcaoFilter::hash(sd.adds[i] & 0xFFFFFFu));
                    m->scored_shards_used++;
                } else if (sd.scored) {
                    // (the chunk offsets already put into the messages do not matter: they are dropped.)  The records come
                    // through the shard's own device thread: its context is that thread's while captures are in flight.
                    push_cmd(*m->dev[(size_t)k], Cmd{Cmd::kFetch, m->collected});
                    if (!wait_done(m, [&] { return sd.fetched.load(std::memory_order_acquire) != 0; })) {
                        rc = ADSB_ERR_HIP;
                        sd.error = "the device thread did not answer a request for a shard's records";
                    } else {
                        rc = sd.fetch_rc;
                    }
                    if (rc != ADSB_OK) {
                        m->last_error = "device " + std::to_string(m->dev[(size_t)k]->device) + ": " + sd.error;
                        break;
                    }
                    if (sd.n_rec) replay(m->filter, m->crc, sd.rec, sd.n_rec, sd.chunk_base, out);
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
            // container's quota: threads beyond it only get the whole process throttled); device threads that block
            // between looks take next to nothing
            const bool block = m->block.load(std::memory_order_relaxed);
            const int room = usable_cpus() - (block ? 0 : (int)m->dev.size()) - 2;
            workers = std::max(1, std::min(workers, room));
            if (const char *e = tuning_env("ADSB_POOL_WORKERS")) workers = std::max(1, std::atoi(e));   // (tuning build only)
            // (worker k on the host cores of devices[k % n]'s NUMA node: the records it reads sit in that node's memory;
            // a blocking handle's workers do not stay hot between jobs)
            m->pool.reset(new ReplayPool(workers, [devs](int k) { pin_to_device_numa(devs[(size_t)k % devs.size()]); }, block ? 0 : 1500));
        }
        std::vector<RecordRun> runs;
        std::vector<const ParallelReplay::Adders *> adders;
        for (int k = 0; k < m->n; k++) {
            const StepDev &sd = s.dev[k];
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
            const StepDev &sd = s.dev[k];
            if (sd.n_rec) replay_sorted(m->filter, m->crc, sd.rec, sd.n_rec, sd.chunk_base, out);
        }
}

// (direct: the caller's own array -- a capture scored by several threads goes straight into it when it fits, *direct_n then
// says how many messages it got and `out` stays empty)
int collect_capture(adsb_multi *m, std::vector<adsb_msg> &out, adsb_msg *direct = nullptr, size_t direct_cap = 0, size_t *direct_n = nullptr) noexcept
{
    if (m->collected == m->submitted) return ADSB_ERR_INVALID;
    Step &s = m->step[m->collected % kMultiSteps];
    int rc = ADSB_OK;
    bool direct_done = false;
    adsb_multi_stats st{};
    double tr0 = 0, tr1 = 0;
    if (!wait_done(m, [&] { return s.state.load(std::memory_order_acquire) == kDone; })) {
        // (the device threads bound their own waits, so this is a thread that died or a clock that jumped: the capture's
        // state is unknown and stays unfreed -- nothing more can be done with the handle but destroy it)
        m->poisoned = true;
        try {
            m->poison_error = "a capture was never finished by the device threads";
            m->last_error = m->poison_error;
        } catch (...) {
        }
        return ADSB_ERR_HIP;
    }
    try {
        st.n_samples = s.n_samples;
        st.n_devices = (uint32_t)m->n;
        st.n_addrs_exchanged = s.fresh.size();
        double p1_first = 0, p1_last = 0, p2_first = 0, p2_last = 0, p1_max = 0, p2_max = 0;
        bool any_scored = false;
        for (int k = 0; k < m->n; k++) {
            const StepDev &sd = s.dev[k];
            if (sd.rc != ADSB_OK && rc == ADSB_OK) {
                rc = sd.rc;
                m->last_error = "device " + std::to_string(m->dev[(size_t)k]->device) + " (shard " + std::to_string(k) + "): " + sd.error;
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
        st.ms_phase1_max = (float)(p1_max * 1e3);
        st.ms_phase2_max = (float)(p2_max * 1e3);
        st.ms_phase1_span = (float)((p1_last - p1_first) * 1e3);
        st.ms_phase2_span = (float)((p2_last - p2_first) * 1e3);
        if (m->poisoned) {
            // (computed against a filter history that an earlier capture's failure broke: dropped)
            rc = refuse_poisoned(m);
        } else if (rc == ADSB_OK) {
            if (s.flush_before) m->filter.flush();   // icao_flush() took effect before this capture
            tr0 = now_s();
            replay_capture(m, s, st, any_scored, out, direct, direct_cap, direct_n, direct_done, rc);
            tr1 = now_s();
        }
        if (rc != ADSB_OK && !m->poisoned) {
            m->poisoned = true;
            m->poison_error = m->last_error;
        }
    } catch (...) {
        // (out of memory in the middle of the replay: the filter may hold half the capture's additions)
        rc = ADSB_ERR_NOMEM;
        m->poisoned = true;
        try {
            m->poison_error = "out of memory while a capture was replayed";
            m->last_error = m->poison_error;
        } catch (...) {
        }
    }
    st.n_messages = rc != ADSB_OK ? 0 : (direct_done ? *direct_n : out.size());
    st.ms_wall = (float)((s.t_done - s.t_submit) * 1e3);
    st.ms_exchange = (float)((s.t_exchange1 - s.t_exchange0) * 1e3);
    st.ms_replay = (float)((tr1 - tr0) * 1e3);
    m->stats = st;
    if (rc != ADSB_OK) out.clear();
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

void resolve_wait(adsb_multi *m)
{
    // AUTO: the device threads spin while anything is out on their device -- a phase's end is seen within a fraction of a
    // microsecond, which is worth 4-7 % to a caller with ONE capture in flight and nothing to a pipelined one -- where that
    // is cheap: at most half of the CPUs the process may use (affinity mask, cgroup quota) go to threads that spin or score
    // (devices + caller + a few pool workers).  Where it is not, spinning is ruinous: threads beyond a cgroup's quota get the
    // whole process throttled, threads beyond the affinity mask take turns with the one that has work (measured on eight
    // contexts under four CPUs: +43-54 % per capture; profiles/r6_wait_policy.txt)
    bool block = m->wait_setting == ADSB_WAIT_BLOCK;
    if (m->wait_setting == ADSB_WAIT_AUTO) block = usable_cpus() < 2 * (m->n + 3);
    m->block.store(block, std::memory_order_relaxed);
}

void destroy_multi(adsb_multi *m) noexcept
{
    // what is still in flight is finished first (its kernels write into the contexts' memory); every wait in there is bounded
    std::vector<adsb_msg> drop;
    while (m->collected < m->submitted) {
        drop.clear();
        const uint64_t before = m->collected;
        (void)collect_capture(m, drop);
        if (m->collected == before) break;   // (a capture the device threads never finished: its contexts are leaked below)
    }
    const bool stuck = m->collected < m->submitted;
#ifdef ADSB_TUNING
    if (tuning_env("ADSB_HOST_TIMES") && m->parallel_scored)
        std::fprintf(stderr, "adsb_multi parallel replay: %llu captures; us per capture: plan %.1f, scan %.1f, merge %.1f, score %.1f, finish %.1f\n",
                     (unsigned long long)m->parallel_scored, m->t_stage[0] / m->parallel_scored * 1e6, m->t_stage[1] / m->parallel_scored * 1e6,
                     m->t_stage[2] / m->parallel_scored * 1e6, m->t_stage[3] / m->parallel_scored * 1e6, m->t_stage[4] / m->parallel_scored * 1e6);
#endif
    for (auto &d : m->dev)
        if (d->th.joinable()) push_cmd(*d, Cmd{Cmd::kStop, 0});
    for (auto &d : m->dev)
        if (d->th.joinable()) {
            if (stuck) d->th.detach();   // (whatever it is stuck in; the handle's memory is leaked with it)
            else d->th.join();
        }
    if (stuck) return;
    m->pool.reset();
    if (!m->dev.empty()) {
        DeviceGuard on_device(m->dev[0]->device);
        for (void *p : m->host_blocks) (void)hipHostFree(p);
    }
    for (auto &d : m->dev) {
        // a device that stopped answering: freeing its memory would wait for the kernel that never ends -- leaked, with
        // the context (the process is expected to end, or to reset the device)
        if (d->dead.load(std::memory_order_relaxed) || !d->ctx) continue;
        DeviceGuard on_device(d->device);
        for (void *p : d->d_stage)
            if (p) (void)hipFree(p);
        adsb_destroy(d->ctx);
    }
    delete m;
}

int finish_collect(adsb_multi *m, int rc, size_t direct_n, adsb_msg *out, size_t cap, size_t *n_out)
{
    if (rc != ADSB_OK) return rc;
    if (direct_n != ~(size_t)0) {   // (the messages are in `out` already)
        if (n_out) *n_out = direct_n;
        m->has_undelivered = false;
        m->undelivered.clear();
        return ADSB_OK;
    }
    return deliver_multi(m, m->msgs, out, cap, n_out);
}

}  // namespace

extern "C" {

int adsb_multi_create(adsb_multi **out, const int *devices, int n_devices, size_t max_chunks_per_device)
{
    if (!out) return ADSB_ERR_INVALID;
    *out = nullptr;
    if (!devices || n_devices <= 0 || n_devices > 64) return ADSB_ERR_INVALID;
    return abi_guard([&]() -> int {
        if (max_chunks_per_device == 0) max_chunks_per_device = 1;
        adsb_multi *m = new adsb_multi;
        m->n = n_devices;
        m->max_chunks = max_chunks_per_device;
        auto undo = [&](int rc) {
            for (auto &e : m->dev)
                if (e->ctx) adsb_destroy(e->ctx);
            delete m;
            return rc;
        };
        try {
            for (Step &s : m->step) {
                s.dev.reset(new StepDev[(size_t)n_devices]);
                s.earlier.resize((size_t)n_devices);
            }
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
                if (rc != ADSB_OK) return undo(rc);
                // (the context's own per-pass timing is not read here: no events on the shards' streams)
                (void)adsb_set_profiling(d->ctx, 0);
                d->ctx->shard_scoring = true;   // (a dense stream's shards are scored on their devices: adsb_shard.cpp)
                m->dev.push_back(std::move(d));
            }
            resolve_wait(m);
            for (auto &d : m->dev) d->th = std::thread(device_thread, m, d.get());
        } catch (...) {
            for (auto &d : m->dev)
                if (d->th.joinable()) {
                    push_cmd(*d, Cmd{Cmd::kStop, 0});
                    d->th.join();
                }
            return undo(ADSB_ERR_NOMEM);
        }
        *out = m;
        return ADSB_OK;
    });
}

void adsb_multi_destroy(adsb_multi *m)
{
    if (m) destroy_multi(m);
}

int adsb_multi_host_alloc(adsb_multi *m, size_t bytes, void **out)
{
    if (!m || !out || bytes == 0) return ADSB_ERR_INVALID;
    *out = nullptr;
    return abi_guard([&]() -> int {
        DeviceGuard on_device(m->dev[0]->device);
        void *p = nullptr;
        // pinned for every device of the process (portable): each device thread's copy of its range out of it is
        // a DMA over that device's own link
        if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            m->last_error = "hipHostMalloc (pinned, portable) failed";
            return ADSB_ERR_NOMEM;
        }
        try {
            m->host_blocks.push_back(p);
        } catch (...) {
            (void)hipHostFree(p);
            throw;
        }
        *out = p;
        return ADSB_OK;
    });
}

int adsb_multi_host_free(adsb_multi *m, void *p)
{
    if (!m || !p) return ADSB_ERR_INVALID;
    return abi_guard([&]() -> int {
        if (m->submitted != m->collected) return ADSB_ERR_BUSY;   // a capture in flight may still be read out of it
        auto it = std::find(m->host_blocks.begin(), m->host_blocks.end(), p);
        if (it == m->host_blocks.end()) return ADSB_ERR_INVALID;
        DeviceGuard on_device(m->dev[0]->device);
        (void)hipHostFree(p);
        m->host_blocks.erase(it);
        return ADSB_OK;
    });
}

int adsb_multi_submit_iq(adsb_multi *m, const int16_t *iq_re_im, size_t n_samples)
{
    if (!m || !iq_re_im || n_samples == 0) return ADSB_ERR_INVALID;
    return abi_guard([&]() -> int {
        if ((n_samples + kChunkSamples - 1) / kChunkSamples > (size_t)m->n * m->max_chunks) return ADSB_ERR_INVALID;
        size_t n[64];
        for (int k = 0; k < m->n; k++) (void)adsb_multi_shard_range(n_samples, m->n, k, nullptr, &n[k]);
        return submit_capture(m, nullptr, iq_re_im, true, n);
    });
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
    return abi_guard([&]() -> int {
        if (m->poisoned) {
            // the restart: every context back to what adsb_create left (its own device thread does it), the one filter and
            // the known set empty
            if (m->submitted != m->collected) return ADSB_ERR_BUSY;
            for (auto &d : m->dev) {
                d->reset_done.store(0, std::memory_order_relaxed);
                push_cmd(*d, Cmd{Cmd::kReset, 0});
            }
            int rc = ADSB_OK;
            for (auto &d : m->dev) {
                if (!wait_done(m, [&] { return d->reset_done.load(std::memory_order_acquire) != 0; })) {
                    m->last_error = "device " + std::to_string(d->device) + ": its thread did not answer the reset";
                    return ADSB_ERR_HIP;
                }
                if (d->reset_rc != ADSB_OK && rc == ADSB_OK) {
                    rc = d->reset_rc;
                    m->last_error = "device " + std::to_string(d->device) + " could not be reset (" + d->ctx->last_error +
                                    "): the adsb_multi stays poisoned, destroy it";
                }
            }
            if (rc != ADSB_OK) return rc;
            m->filter.flush();
            m->known.clear();
            m->poisoned = false;
            m->poison_error.clear();
        }
        m->flush_pending = true;   // applies to the captures submitted after it, like adsb_icao_flush
        return ADSB_OK;
    });
}

int adsb_multi_submit_iq_device(adsb_multi *m, const void *const *device_iq, const size_t *n_samples)
{
    if (!m || !device_iq || !n_samples) return ADSB_ERR_INVALID;
    return abi_guard([&] { return submit_capture(m, device_iq, nullptr, false, n_samples); });
}

int adsb_multi_collect(adsb_multi *m, adsb_msg *out, size_t cap, size_t *n_out)
{
    if (!m || (!out && cap)) return ADSB_ERR_INVALID;
    return abi_guard([&]() -> int {
        m->msgs.clear();
        size_t direct_n = ~(size_t)0;
        const int rc = collect_capture(m, m->msgs, out, cap, &direct_n);
        return finish_collect(m, rc, direct_n, out, cap, n_out);
    });
}

int adsb_multi_demod_iq_device(adsb_multi *m, const void *const *device_iq, const size_t *n_samples, adsb_msg *out,
                               size_t cap, size_t *n_out)
{
    if (!m || !device_iq || !n_samples || (!out && cap)) return ADSB_ERR_INVALID;
    return abi_guard([&]() -> int {
        if (m->submitted != m->collected) return ADSB_ERR_BUSY;
        if (int rc = submit_capture(m, device_iq, nullptr, false, n_samples)) return rc;
        m->msgs.clear();
        size_t direct_n = ~(size_t)0;
        const int rc = collect_capture(m, m->msgs, out, cap, &direct_n);
        return finish_collect(m, rc, direct_n, out, cap, n_out);
    });
}

int adsb_multi_demod_iq(adsb_multi *m, const int16_t *iq_re_im, size_t n_samples, adsb_msg *out, size_t cap, size_t *n_out)
{
    if (!m || (!iq_re_im && n_samples) || (!out && cap)) return ADSB_ERR_INVALID;
    return abi_guard([&]() -> int {
        if (m->submitted != m->collected) return ADSB_ERR_BUSY;
        // a capture of any length: in pieces of at most what the devices' contexts hold together, each piece cut
        // into contiguous ranges (consecutive pieces are consecutive captures through the one filter)
        const size_t piece = (size_t)m->n * m->max_chunks * kChunkSamples;
        std::vector<adsb_msg> msgs;
        adsb_multi_stats total{};
        size_t n[64];
        for (size_t off = 0; off < n_samples || (off == 0 && n_samples == 0); off += piece) {
            const size_t len = std::min(piece, n_samples - off);
            for (int k = 0; k < m->n; k++) (void)adsb_multi_shard_range(len, m->n, k, nullptr, &n[k]);
            if (int rc = submit_capture(m, nullptr, n_samples ? iq_re_im + 2 * off : nullptr, true, n)) return rc;
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
    });
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

int adsb_multi_set_wait(adsb_multi *m, int mode)
{
    if (!m || (mode != ADSB_WAIT_AUTO && mode != ADSB_WAIT_SPIN && mode != ADSB_WAIT_BLOCK)) return ADSB_ERR_INVALID;
    if (m->submitted != m->collected) return ADSB_ERR_BUSY;
    m->wait_setting = mode;
    resolve_wait(m);
    m->pool.reset();   // (sized and kept hot for the mode it was made in: the next busy capture makes another)
    return ADSB_OK;
}

int adsb_multi_get_wait(const adsb_multi *m)
{
    if (!m) return ADSB_ERR_INVALID;
    return m->block.load(std::memory_order_relaxed) ? ADSB_WAIT_BLOCK : ADSB_WAIT_SPIN;
}

int adsb_multi_set_timeout_ms(adsb_multi *m, uint32_t ms)
{
    if (!m) return ADSB_ERR_INVALID;
    m->timeout_ms.store(ms ? ms : kDefaultTimeoutMs, std::memory_order_relaxed);
    return ADSB_OK;
}

int adsb_multi_selftest_fail(adsb_multi *m, uint32_t captures_from_now, int shard, int kind)
{
    if (!m || shard < 0 || shard >= m->n || kind < 0 || kind > ADSB_FAULT_RECORDS) return ADSB_ERR_INVALID;
    m->fault_kind.store(0, std::memory_order_relaxed);
    m->fault_capture.store(m->submitted + captures_from_now, std::memory_order_relaxed);
    m->fault_shard.store(shard, std::memory_order_relaxed);
    m->fault_kind.store(kind, std::memory_order_relaxed);
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
    out8[7] = m->poisoned ? 1 : 0;
    return ADSB_OK;
}

const char *adsb_multi_last_error(const adsb_multi *m) { return m ? m->last_error.c_str() : ""; }

}  // extern "C"
