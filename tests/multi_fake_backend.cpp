// tests/multi_fake_backend.cpp -- TEST INFRASTRUCTURE: adsb_multi's orchestration (csrc/adsb_multi.cpp: the device
// threads, the step state machine, the address exchange, the collector, the poisoned-handle rules) on the CPU, under
// ThreadSanitizer and under AddressSanitizer + UBSan, with no GPU and no HIP runtime.  Built and run by
// tests/test_multi_orchestration.py.
//
// csrc/adsb_multi.cpp and csrc/adsb_replay_host.cpp are compiled as they are.  What they call below them is faked here:
//
//   * the HIP calls adsb_multi.cpp makes itself (hipSetDevice, hipMalloc, hipMemcpyAsync, hipHostMalloc ...): malloc / memcpy;
//   * adsb_create / adsb_destroy and the shard_* entry points of csrc/adsb_shard.cpp: a "device" that slices its shard
//     with the ORACLE (orc_to_mag + orc_all_trials: every trial of every position the gates let through), keeps an address
//     superset per flush epoch exactly as the real context does (own learned addresses at phase 1, the exchange's at
//     phase 2), hands back the self-validating trials plus the address/parity trials whose value the superset holds --
//     so an address the exchange fails to deliver costs a record and shows up as a wrong frame list --, that scores about
//     half of its shards itself as k_score / k_emit do (the oracle's score_modes_message against "exact bitmap" + what the
//     shards before it add), and whose phases "land" from ANOTHER thread after random delays (0 us ... 6 ms), like a
//     kernel's write to mapped memory;
//
// and main() drives random sequences of captures (1-8 devices, 1-4 captures in flight, flushes, the host and the device
// forms, spin and block waits, injected failures of every kind with the recovery that follows) and compares every
// capture's messages with ONE oracle stream (orc_demod_iq) fed the same sequence.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <random>
#include <set>
#include <thread>
#include <unordered_map>

#include "../dump1090_rs_amd/csrc/adsb_ctx.h"
#include "../oracle/dump1090_oracle.h"

using namespace adsb::host;

// ------------------------------------------------------------------------------------------------------------------
// operator new that can be told to fail: "the n-th allocation from now, on whichever thread, throws std::bad_alloc"
// (include/adsb_hip.h: nothing is thrown across the ABI, nothing aborts -- main() arms it around single library calls)
// ------------------------------------------------------------------------------------------------------------------
// (GCC sees malloc / free inside the replaced operators inlined next to new-expressions and calls that a mismatch: it is the
// pair the replacement defines)
#pragma GCC diagnostic ignored "-Wmismatched-new-delete"
static std::atomic<long> g_new_countdown{0};   // <= 0: off
static std::atomic<long> g_new_failures{0};
void *operator new(std::size_t n)
{
    if (g_new_countdown.load(std::memory_order_relaxed) > 0 && g_new_countdown.fetch_sub(1, std::memory_order_relaxed) == 1) {
        g_new_failures.fetch_add(1, std::memory_order_relaxed);
        throw std::bad_alloc();
    }
    void *p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new[](std::size_t n) { return operator new(n); }
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }

// ------------------------------------------------------------------------------------------------------------------
// HIP, as far as adsb_multi.cpp and the inline helpers of adsb_ctx.h use it
// ------------------------------------------------------------------------------------------------------------------
static thread_local int t_device = 0;
extern "C" {
hipError_t hipSetDevice(int d) { t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = 8; return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char *, int, int) { return hipErrorInvalidDevice; }   // (no NUMA pinning in the tests)
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "fake HIP error"; }
hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(dst, src, n); return hipSuccess; }
}

// ------------------------------------------------------------------------------------------------------------------
// the fake GPU: phases land from its thread
// ------------------------------------------------------------------------------------------------------------------
namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

class FakeGpu {
  public:
    FakeGpu() : th_([this] { run(); }) {}
    ~FakeGpu()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void land_at(double when, std::atomic<uint32_t> *flag, uint32_t seq)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            q_.push(Ev{when, flag, seq});
        }
        cv_.notify_all();
    }

  private:
    struct Ev {
        double when;
        std::atomic<uint32_t> *flag;
        uint32_t seq;
        bool operator<(const Ev &o) const { return when > o.when; }
    };
    void run()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            if (stop_) return;
            if (q_.empty()) {
                cv_.wait(lk);
                continue;
            }
            const double t = now_s();
            if (q_.top().when <= t) {
                Ev e = q_.top();
                q_.pop();
                e.flag->store(e.seq, std::memory_order_release);   // "the summary lands in mapped memory"
                continue;
            }
            const double wait = q_.top().when - t;
            // (wait_until on the system clock: GCC 11's ThreadSanitizer does not know wait_for's pthread_cond_clockwait)
            if (wait > 80e-6) cv_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds((long)((wait - 50e-6) * 1e6)));
            else {   // close: spin with the lock dropped
                lk.unlock();
                while (now_s() < t + wait) {}
                lk.lock();
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::priority_queue<Ev> q_;
    bool stop_ = false;
    std::thread th_;
};
FakeGpu *g_gpu = nullptr;

// every trial of one buffer (131072 samples or the capture's ragged end), by content: the oracle slices each distinct
// buffer once
struct TrialCache {
    std::mutex mu;
    std::unordered_map<uint64_t, std::shared_ptr<std::vector<TrialRecord>>> map;
    std::shared_ptr<std::vector<TrialRecord>> get(const int16_t *iq, size_t n)
    {
        uint64_t h = 1469598103934665603ull ^ n;
        const uint64_t *w = reinterpret_cast<const uint64_t *>(iq);
        for (size_t i = 0; i < n / 2; i++) h = (h ^ w[i]) * 1099511628211ull + (h >> 29);
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = map.find(h);
            if (it != map.end()) return it->second;
        }
        auto mag = std::make_unique<orc_magbuf>();
        if (orc_to_mag(iq, n, mag.get()) != 0) std::abort();
        auto out = std::make_shared<std::vector<TrialRecord>>(5 * 131072 / 8);
        static_assert(sizeof(orc_trial) == sizeof(TrialRecord), "record layout");
        const size_t k = orc_all_trials(mag.get(), 0, reinterpret_cast<orc_trial *>(out->data()), out->size());
        if (k > out->size()) std::abort();
        out->resize(k);
        std::lock_guard<std::mutex> lk(mu);
        map[h] = out;
        return out;
    }
} g_trials;

typedef std::set<uint32_t> AddrSet;

struct FakeShard {
    std::shared_ptr<AddrSet> bitmap;      // the superset this shard matches against (its flush epoch's)
    bool exact_flush = false;             // an icao_flush precedes the shard: its scoring starts from an empty "exact bitmap"
    bool scored = false;                  // the "device" scored the shard itself (a dense stream's: csrc/adsb_shard.cpp)
    std::vector<adsb_msg> scored_msgs;    // ... its messages (chunk = buffer within the shard) and the values its replay adds
    std::vector<uint32_t> scored_adds;
    std::vector<TrialRecord> all;         // every trial of the shard, chunk = buffer within the shard
    std::vector<TrialRecord> out;         // what the second phase hands over
    std::vector<uint32_t> learned;
    std::atomic<uint32_t> landed{0};      // written by the fake GPU's thread
    uint32_t seq = 0;
    uint64_t n_samples = 0;
};

struct FakeCtx {
    std::shared_ptr<AddrSet> cur;
    AddrSet exact;                        // the filter as the device believes it stood before the capture (the exact bitmap)
    FakeShard shard[kSlots];
    std::mt19937_64 rng;
    uint32_t next_seq = 1;
};
FakeCtx *fake_of(adsb_ctx *c) { return reinterpret_cast<FakeCtx *>(c->h_block); }

// a failure the fake itself injects: the n-th call (counted over all contexts) of one of its entry points fails
std::atomic<int> g_live_contexts{0};    // adsb_create - adsb_destroy
std::atomic<int> g_fail_call{-1};      // 0 shard_begin, 1 shard_learned, 2 shard_match, 3 shard_records, 4 shard_reset, 5 adsb_create
std::atomic<int> g_fail_countdown{0};
bool fake_fails(adsb_ctx *c, int call)
{
    if (g_fail_call.load(std::memory_order_relaxed) != call) return false;
    if (g_fail_countdown.fetch_sub(1, std::memory_order_relaxed) != 1) return false;
    g_fail_call.store(-1, std::memory_order_relaxed);
    c->last_error = "fake backend: injected failure of call kind " + std::to_string(call);
    return true;
}

uint32_t addr_of(const uint8_t *m) { return uint32_t(m[1]) << 16 | uint32_t(m[2]) << 8 | m[3]; }

void schedule_landing(FakeCtx *f, FakeShard &sh)
{
    // mostly a few hundred microseconds; now and then at once, now and then long enough for the device thread to ask
    // the streams (csrc/adsb_multi.cpp: kCheckAfterS)
    const uint64_t r = f->rng() % 100;
    const double delay = r < 10 ? 0.0 : (r < 93 ? (double)(f->rng() % 400) * 1e-6 : (2.5e-3 + (double)(f->rng() % 3500) * 1e-6));
    sh.seq = f->next_seq++;
    if (f->next_seq == 0) f->next_seq = 1;
    g_gpu->land_at(now_s() + delay, &sh.landed, sh.seq);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// the context and the shard entry points (csrc/adsb_ctx.h), faked
// ------------------------------------------------------------------------------------------------------------------
extern "C" {

int adsb_create(adsb_ctx **out, int device, size_t max_chunks)
{
    *out = nullptr;
    if (g_fail_call.load(std::memory_order_relaxed) == 5 && g_fail_countdown.fetch_sub(1, std::memory_order_relaxed) == 1) {
        g_fail_call.store(-1, std::memory_order_relaxed);
        return ADSB_ERR_NOMEM;   // (the k-th context of an adsb_multi cannot be made: the ones before it must be undone)
    }
    adsb_ctx *c = nullptr;
    FakeCtx *f = nullptr;
    try {   // (the real adsb_create returns ADSB_ERR_NOMEM where this would throw: csrc/adsb_context.cpp)
        c = new adsb_ctx;
        f = new FakeCtx;
        f->cur = std::make_shared<AddrSet>();
        f->cur->insert(0);
    } catch (...) {
        delete c;
        delete f;
        return ADSB_ERR_NOMEM;
    }
    c->device = device;
    c->max_chunks = max_chunks;
    f->rng.seed(0x5EED0000u + (uint64_t)device * 977 + max_chunks);
    c->h_block = reinterpret_cast<char *>(f);
    c->flush_pending = true;
    g_live_contexts.fetch_add(1, std::memory_order_relaxed);
    *out = c;
    return ADSB_OK;
}

void adsb_destroy(adsb_ctx *c)
{
    if (!c) return;
    g_live_contexts.fetch_sub(1, std::memory_order_relaxed);
    delete fake_of(c);
    delete c;
}

int adsb_set_profiling(adsb_ctx *, int) { return ADSB_OK; }

}  // extern "C"

namespace adsb {
namespace host {

int shard_begin(adsb_ctx *c, int k, const void *d_iq, uint64_t n_samples, bool)
{
    FakeCtx *f = fake_of(c);
    FakeShard &sh = f->shard[k];
    adsb_ctx::ShardJob &job = c->shard[k];
    if (job.active) return ADSB_ERR_BUSY;
    if (fake_fails(c, 0)) return ADSB_ERR_HIP;
    sh.exact_flush = c->flush_pending;
    if (c->flush_pending) {
        f->cur = std::make_shared<AddrSet>();
        f->cur->insert(0);   // (address 0 always tests true: src/icao_filter.rs:71-80)
        c->flush_pending = false;
    }
    sh.scored = false;
    sh.bitmap = f->cur;
    sh.n_samples = n_samples;
    sh.all.clear();
    sh.out.clear();
    sh.learned.clear();
    const int16_t *iq = static_cast<const int16_t *>(d_iq);
    for (uint64_t off = 0, ch = 0; off < n_samples; off += kChunkSamples, ch++) {
        const size_t n = (size_t)std::min<uint64_t>(kChunkSamples, n_samples - off);
        auto trials = g_trials.get(iq + 2 * off, n);
        for (TrialRecord r : *trials) {
            r.chunk = (uint32_t)ch;
            sh.all.push_back(r);
        }
    }
    // the scan: the addresses the shard's self-validating trials can add (mode_s/mod.rs:80-84, 97-99) -- into its own
    // superset at once, and to the exchange
    for (const TrialRecord &r : sh.all) {
        const unsigned df = r.msg[0] >> 3;
        if (df == 17 && orc_modes_checksum(r.msg, 112) == 0) sh.learned.push_back(addr_of(r.msg));
        if (df == 11 && orc_modes_checksum(r.msg, 56) == 0) sh.learned.push_back(addr_of(r.msg));
    }
    std::sort(sh.learned.begin(), sh.learned.end());
    sh.learned.erase(std::unique(sh.learned.begin(), sh.learned.end()), sh.learned.end());
    for (uint32_t a : sh.learned) sh.bitmap->insert(a);
    job.active = true;
    job.waiting = false;
    job.result_scored = false;
    c->shard_jobs++;
    if (n_samples) {
        job.waiting = true;
        schedule_landing(f, sh);
    }
    return ADSB_OK;
}

bool shard_phase_landed(adsb_ctx *c, int k)
{
    adsb_ctx::ShardJob &job = c->shard[k];
    if (!job.waiting) return true;
    FakeShard &sh = fake_of(c)->shard[k];
    if (sh.landed.load(std::memory_order_acquire) != sh.seq) return false;
    job.waiting = false;
    return true;
}

int shard_phase_check(adsb_ctx *c, int k) { return shard_phase_landed(c, k) ? 1 : 0; }

int shard_learned(adsb_ctx *c, int k, std::vector<uint32_t> &addrs)
{
    adsb_ctx::ShardJob &job = c->shard[k];
    addrs.clear();
    if (!job.active || job.waiting) return ADSB_ERR_INVALID;
    if (fake_fails(c, 1)) return ADSB_ERR_HIP;
    addrs = fake_of(c)->shard[k].learned;
    return ADSB_OK;
}

// A shard scored where it is, as k_score / k_emit do it -- here by the ORACLE: the shard's records in replay order through
// score_modes_message (src/mode_s/mod.rs:34-139) and the best-of-5 selection (src/demod_2400.rs:184-207) against a filter
// that holds what the exact bitmap holds plus what the shards before this one add.
void score_shard(FakeCtx *f, FakeShard &sh, const uint32_t *earlier, size_t n_earlier)
{
    auto filt = std::make_unique<orc_filter>();
    orc_icao_flush(filt.get());
    for (uint32_t a : f->exact) orc_icao_filter_add(filt.get(), a);
    for (size_t i = 0; i < n_earlier; i++) orc_icao_filter_add(filt.get(), earlier[i]);
    std::vector<const TrialRecord *> order;
    for (const TrialRecord &r : sh.out) order.push_back(&r);
    std::sort(order.begin(), order.end(), [](const TrialRecord *a, const TrialRecord *b) {
        if (a->chunk != b->chunk) return a->chunk < b->chunk;
        if ((a->j_tp & 0xFFFFFFu) != (b->j_tp & 0xFFFFFFu)) return (a->j_tp & 0xFFFFFFu) < (b->j_tp & 0xFFFFFFu);
        return (a->j_tp >> 24) < (b->j_tp >> 24);
    });
    sh.scored_msgs.clear();
    sh.scored_adds.clear();
    for (size_t i = 0; i < order.size();) {
        const uint32_t chunk = order[i]->chunk, j = order[i]->j_tp & 0xFFFFFFu;
        const TrialRecord *best = nullptr;
        int32_t best_score = -2;
        int best_len = 7;
        for (; i < order.size() && order[i]->chunk == chunk && (order[i]->j_tp & 0xFFFFFFu) == j; i++) {
            const TrialRecord &r = *order[i];
            const unsigned df = r.msg[0] >> 3;
            const bool adder = ((df == 17 || df == 18) && orc_modes_checksum(r.msg, 112) == 0) || (df == 11 && orc_modes_checksum(r.msg, 56) == 0);
            if (adder && !orc_icao_filter_test(filt.get(), addr_of(r.msg)))   // (what score_modes_message is about to hand icao_filter_add)
                sh.scored_adds.push_back(df == 18 ? (addr_of(r.msg) | ORC_ICAO_FILTER_ADSB_NT) : addr_of(r.msg));
            int len = 0;
            int32_t score = 0;
            if (!orc_score_modes_message(filt.get(), r.msg, 14, &len, &score)) continue;
            if (score > best_score) best = &r, best_score = score, best_len = len;
        }
        if (!best || best_score < 0) continue;
        adsb_msg m{};
        std::memcpy(m.msg, best->msg, 14);
        m.len = (uint8_t)best_len;
        m.try_phase = (uint8_t)(best->j_tp >> 24);
        m.score = best_score;
        m.j = j;
        m.chunk = chunk;
        m.signal_level = (double)(best->power & ((1ull << 40) - 1)) / 65535.0 / 65535.0 / 33.0;
        sh.scored_msgs.push_back(m);
    }
}

int shard_match(adsb_ctx *c, int k, const uint32_t *extra, size_t n_extra, const uint32_t *earlier, size_t n_earlier)
{
    FakeCtx *f = fake_of(c);
    FakeShard &sh = f->shard[k];
    adsb_ctx::ShardJob &job = c->shard[k];
    if (!job.active || job.waiting) return ADSB_ERR_INVALID;
    if (fake_fails(c, 2)) return ADSB_ERR_HIP;
    for (size_t i = 0; i < n_extra; i++) sh.bitmap->insert(extra[i]);
    // match + records: what can score >= 0 or add to the filter, given the superset
    for (const TrialRecord &r : sh.all) {
        const unsigned df = r.msg[0] >> 3;
        bool keep = false;
        if (df == 17 || df == 18) keep = orc_modes_checksum(r.msg, 112) == 0;
        else if (df == 11) {
            const uint32_t crc = orc_modes_checksum(r.msg, 56);
            keep = (crc & 0xFFFF80u) == 0 && ((crc & 0x7Fu) == 0 || sh.bitmap->count(addr_of(r.msg)));
        } else if (df == 0 || df == 4 || df == 5) keep = sh.bitmap->count(orc_modes_checksum(r.msg, 56)) != 0;
        else if (df == 16 || df == 20 || df == 21 || df >= 24) keep = sh.bitmap->count(orc_modes_checksum(r.msg, 112)) != 0;
        if (keep) sh.out.push_back(r);
    }
    // (now and then out of replay order, as records of a sparse shard come: the device thread sorts them)
    if (sh.out.size() > 2 && f->rng() % 3 == 0) std::swap(sh.out[0], sh.out[sh.out.size() - 1]);
    // the exact bitmap's life (csrc/adsb_shard.cpp: exact_side): cleared in front of a flushed capture's scoring, this
    // shard scored against it, then the capture's additions -- every shard's, the exchange has them -- committed to it
    if (sh.exact_flush) f->exact.clear();
    if (c->shard_scoring && sh.n_samples && f->rng() % 2 == 0) {
        score_shard(f, sh, earlier, n_earlier);
        sh.scored = true;
    }
    for (size_t i = 0; i < n_extra; i++) f->exact.insert(extra[i]);
    if (sh.n_samples) {
        job.waiting = true;
        schedule_landing(f, sh);
    }
    return ADSB_OK;
}

int shard_records(adsb_ctx *c, int k, const TrialRecord **rec, size_t *n_out)
{
    FakeShard &sh = fake_of(c)->shard[k];
    adsb_ctx::ShardJob &job = c->shard[k];
    *rec = nullptr;
    *n_out = 0;
    if (!job.active || job.waiting) return ADSB_ERR_INVALID;
    job.active = false;
    if (fake_fails(c, 3)) return ADSB_ERR_HIP;
    adsb_stats st{};
    st.n_samples = sh.n_samples;
    st.n_chunks = (sh.n_samples + kChunkSamples - 1) / kChunkSamples;
    st.n_records = sh.out.size();
    c->stats = st;
    job.result_scored = sh.scored;   // (a scored shard's records stay on the "device": shard_fetch_records brings them over)
    *rec = sh.scored ? nullptr : sh.out.data();
    *n_out = sh.scored ? 0 : sh.out.size();
    return ADSB_OK;
}

bool shard_scored_result(adsb_ctx *c, int k, adsb_msg **msgs, size_t *n_msgs, const uint32_t **adds, size_t *n_adds)
{
    FakeShard &sh = fake_of(c)->shard[k];
    if (!c->shard[k].result_scored) return false;
    *msgs = sh.scored_msgs.data();
    *n_msgs = sh.scored_msgs.size();
    *adds = sh.scored_adds.data();
    *n_adds = sh.scored_adds.size();
    return true;
}

int shard_fetch_records(adsb_ctx *c, int k, const TrialRecord **rec, size_t *n_out)
{
    FakeShard &sh = fake_of(c)->shard[k];
    *rec = sh.out.data();
    *n_out = sh.out.size();
    return ADSB_OK;
}

int shard_reset(adsb_ctx *c)
{
    if (fake_fails(c, 4)) return ADSB_ERR_HIP;
    FakeCtx *f = fake_of(c);
    f->cur = std::make_shared<AddrSet>();
    f->cur->insert(0);
    f->exact.clear();
    for (int k = 0; k < kSlots; k++) {
        c->shard[k].active = c->shard[k].waiting = false;
        f->shard[k].bitmap.reset();
    }
    c->flush_pending = false;
    return ADSB_OK;
}

}  // namespace host
}  // namespace adsb

// ------------------------------------------------------------------------------------------------------------------
// the driver
// ------------------------------------------------------------------------------------------------------------------
namespace {

bool same(const adsb_msg *got, size_t n_got, const std::vector<orc_msg> &want, const char *what, int seq_no, int step)
{
    bool ok = n_got == want.size();
    for (size_t i = 0; ok && i < n_got; i++) {
        const adsb_msg &g = got[i];
        const orc_msg &w = want[i];
        ok = std::memcmp(g.msg, w.msg, 14) == 0 && g.len == w.len && g.try_phase == w.try_phase && g.score == w.score && g.j == w.j &&
             g.chunk == w.chunk && g.signal_level == w.signal_level;
        if (!ok) std::fprintf(stderr, "sequence %d step %d (%s): message %zu differs: chunk %llu/%llu j %u/%u score %d/%d\n", seq_no, step, what, i,
                              (unsigned long long)g.chunk, (unsigned long long)w.chunk, g.j, w.j, g.score, w.score);
    }
    if (n_got != want.size()) std::fprintf(stderr, "sequence %d step %d (%s): %zu messages, the oracle has %zu\n", seq_no, step, what, n_got, want.size());
    return ok;
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <arena.iq: whole 131072-sample buffers of {re, im} int16> <sequences> [seed]\n", argv[0]);
        return 2;
    }
    std::vector<int16_t> arena;
    {
        FILE *fp = std::fopen(argv[1], "rb");
        if (!fp) return 2;
        std::fseek(fp, 0, SEEK_END);
        const long bytes = std::ftell(fp);
        std::fseek(fp, 0, SEEK_SET);
        arena.resize((size_t)bytes / 2);
        if (std::fread(arena.data(), 1, (size_t)bytes, fp) != (size_t)bytes) return 2;
        std::fclose(fp);
    }
    const size_t arena_chunks = arena.size() / 2 / kChunkSamples;
    const int sequences = std::atoi(argv[2]);
    std::mt19937_64 rng(argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 20261003ull);
    FakeGpu gpu;
    g_gpu = &gpu;
    const size_t cap = 1 << 16;
    std::vector<adsb_msg> got(cap);
    std::vector<orc_msg> scratch(cap);
    auto filt = std::make_unique<orc_filter>();
    size_t total_captures = 0, total_msgs = 0, failures_injected = 0, recoveries = 0, dead_handles = 0, blocked = 0, parallel = 0, poisoned_returns = 0;
    size_t scored_used = 0, scored_refused = 0, failed_resets = 0, failed_creates = 0;

    for (int seq_no = 0; seq_no < sequences; seq_no++) {
        const int n_dev = 1 + (int)(rng() % 8);
        const size_t per = 1 + rng() % 3;
        int devices[8];
        for (int k = 0; k < n_dev; k++) devices[k] = (int)(rng() % 8);
        adsb_multi *m = nullptr;
        if (rng() % 16 == 0) {   // a context that cannot be made: create says so and leaves nothing behind
            g_fail_countdown.store(1 + (int)(rng() % n_dev), std::memory_order_relaxed);
            g_fail_call.store(5, std::memory_order_relaxed);
            const int live_before = g_live_contexts.load();   // (handles whose device was given up keep that device's context: by design)
            if (adsb_multi_create(&m, devices, n_dev, per) != ADSB_ERR_NOMEM || m != nullptr || g_live_contexts.load() != live_before) {
                std::fprintf(stderr, "sequence %d: a failed adsb_multi_create returned success or left contexts behind\n", seq_no);
                return 1;
            }
            failed_creates++;
        }
        if (adsb_multi_create(&m, devices, n_dev, per) != ADSB_OK) return 1;
        const int wait_mode = (int)(rng() % 3);
        if (adsb_multi_set_wait(m, wait_mode) != ADSB_OK) return 1;
        blocked += adsb_multi_get_wait(m) == ADSB_WAIT_BLOCK;
        // (half of the sequences: every capture's records scored by the pool's threads, whatever their number)
        // ... and who scores a shard: its "device" for about half of them (score mode 0), never (1), or the device with the
        // collector refusing every result and asking the shard's thread for the records instead (2)
        static const uint32_t score_modes[] = {0, 0, 1, 2};
        if (adsb_multi_selftest_tune(m, 0, rng() % 2 ? 1 : 0, score_modes[rng() % 4]) != ADSB_OK) return 1;
        const bool with_fault = rng() % 3 == 0;
        const bool hang = with_fault && rng() % 6 == 0;
        if (hang) (void)adsb_multi_set_timeout_ms(m, 500);   // (well above what a phase of the fake takes on a loaded box)
        orc_icao_flush(filt.get());
        const int depth = 1 + (int)(rng() % 4);
        const int steps = 3 + (int)(rng() % 9);
        const int fault_at = with_fault ? (int)(rng() % steps) : -1;
        // what the captures in flight should return: the oracle's list (expect 0), the injected failure (1), ADSB_ERR_POISONED (2)
        std::deque<std::vector<orc_msg>> pending;
        std::deque<int> pending_expect;
        bool broken = false;           // the failing capture has been submitted
        bool fail_collected = false;   // ... and collected: the library knows it is poisoned
        bool dead = false, bad = false;
        std::vector<void *> pinned;

        auto collect_one = [&](int step) {
            size_t n = 0;
            const int rc = adsb_multi_collect(m, got.data(), cap, &n);
            const std::vector<orc_msg> w = std::move(pending.front());
            const int expect = pending_expect.front();
            pending.pop_front();
            pending_expect.pop_front();
            if (expect == 0) {
                if (rc != ADSB_OK) {
                    std::fprintf(stderr, "sequence %d step %d: collect returned %d (%s)\n", seq_no, step, rc, adsb_multi_last_error(m));
                    bad = true;
                } else if (!same(got.data(), n, w, "collect", seq_no, step)) bad = true;
                total_msgs += n;
            } else if (expect == 1) {
                fail_collected = true;
                if (rc == ADSB_OK || rc == ADSB_ERR_POISONED) {
                    std::fprintf(stderr, "sequence %d step %d: the capture with the injected failure returned %d\n", seq_no, step, rc);
                    bad = true;
                }
            } else {
                poisoned_returns++;
                if (rc != ADSB_ERR_POISONED) {
                    std::fprintf(stderr, "sequence %d step %d: a capture behind a failed one returned %d, not ADSB_ERR_POISONED\n", seq_no, step, rc);
                    bad = true;
                }
            }
        };

        for (int step = 0; step < steps && !bad && !dead; step++) {
            // a window of the arena: at most what the contexts hold together (the blocking host form: sometimes more)
            const size_t room = (size_t)n_dev * per;
            int form = (int)(rng() % 4);   // 0 host blocking, 1 device blocking, 2 submit device, 3 submit host
            size_t chunks = rng() % 10 == 0 ? 0 : 1 + rng() % room;
            if (form == 0 && rng() % 4 == 0) chunks = room + 1 + rng() % room;
            chunks = std::min(chunks, arena_chunks);
            if (chunks > room) form = 0;
            const size_t first = rng() % (arena_chunks - chunks + 1);
            size_t n_samples = chunks * kChunkSamples;
            if (chunks && rng() % 3 == 0) n_samples -= (rng() % (kChunkSamples - 400)) / 4 * 4;
            if (n_samples == 0) form = 0;   // (the other forms want samples)
            const int16_t *iq = arena.data() + 2 * first * kChunkSamples;
            const bool inject = step == fault_at && !broken;
            // a fault of the fake's own is counted in calls: it is this capture's only if nothing else is in flight
            const bool fake_fault = inject && !hang && rng() % 2 == 0;
            // (... and only if nothing is submitted behind it before it has made its calls: device threads run ahead of
            // each other, the next capture's first call on one device can come before this one's on another)
            if (fake_fault && form >= 2) form = n_samples ? 1 : 0;
            if (form == 0 || form == 1)
                while (!pending.empty() && !bad) collect_one(step);   // the blocking forms want nothing in flight
            while ((int)pending.size() >= depth && !bad) collect_one(step);
            if (bad) break;
            if (broken && pending.empty()) {
                // now and then the reset of one context fails first: the flush says so, the handle stays poisoned, and the
                // next flush -- nothing wrong any more -- restarts it
                if (!hang && rng() % 4 == 0) {
                    g_fail_countdown.store(1 + (int)(rng() % n_dev), std::memory_order_relaxed);
                    g_fail_call.store(4, std::memory_order_relaxed);
                    uint64_t c8[8] = {};
                    if (adsb_multi_icao_flush(m) == ADSB_OK || adsb_multi_selftest_counters(m, c8) != ADSB_OK || c8[7] != 1 ||
                        adsb_multi_submit_iq(m, arena.data(), kChunkSamples) != ADSB_ERR_POISONED) {
                        std::fprintf(stderr, "sequence %d step %d: a restart whose reset failed did not leave the handle poisoned\n", seq_no, step);
                        bad = true;
                        break;
                    }
                    failed_resets++;
                }
                // the restart: flush, and the oracle's filter with it
                const int rc = adsb_multi_icao_flush(m);
                if (hang) {
                    if (rc == ADSB_OK) {
                        std::fprintf(stderr, "sequence %d: a handle with a dead device accepted the restart\n", seq_no);
                        bad = true;
                    }
                    dead = true;
                    dead_handles++;
                    break;
                }
                if (rc != ADSB_OK) {
                    std::fprintf(stderr, "sequence %d step %d: the restart returned %d (%s)\n", seq_no, step, rc, adsb_multi_last_error(m));
                    bad = true;
                    break;
                }
                orc_icao_flush(filt.get());
                broken = fail_collected = false;
                recoveries++;
            } else if (!broken && rng() % 4 == 0) {
                if (adsb_multi_icao_flush(m) != ADSB_OK) bad = true;
                orc_icao_flush(filt.get());
            } else if (broken && rng() % 4 == 0) {
                // a flush while poisoned captures are still in flight: refused, nothing changes
                if (fail_collected && adsb_multi_icao_flush(m) != ADSB_ERR_BUSY) {
                    std::fprintf(stderr, "sequence %d step %d: a restart with captures in flight was not refused\n", seq_no, step);
                    bad = true;
                }
            }
            const int expect = broken ? 2 : (inject ? 1 : 0);
            if (inject) {
                failures_injected++;
                if (hang) {
                    (void)adsb_multi_selftest_fail(m, 0, (int)(rng() % n_dev), ADSB_FAULT_HANG);
                } else if (!fake_fault) {
                    static const int kinds[] = {ADSB_FAULT_PHASE1, ADSB_FAULT_PHASE2, ADSB_FAULT_RECORDS};
                    (void)adsb_multi_selftest_fail(m, 0, (int)(rng() % n_dev), kinds[rng() % 3]);
                } else {
                    // (every capture -- every piece of a long host capture -- makes n_dev calls of each kind)
                    g_fail_countdown.store(1 + (int)(rng() % n_dev), std::memory_order_relaxed);
                    g_fail_call.store((int)(rng() % 4), std::memory_order_relaxed);
                }
            }
            std::vector<orc_msg> w;
            if (expect == 0) {
                const size_t k = orc_demod_iq(filt.get(), iq, n_samples, scratch.data(), cap, nullptr);
                if (k > cap) return 1;
                w.assign(scratch.begin(), scratch.begin() + (long)k);
            }
            total_captures++;
            // the shards of the device forms: the even split, pointers into the arena
            const void *ptrs[8];
            size_t ns[8];
            for (int k = 0; k < n_dev; k++) {
                size_t a = 0;
                (void)adsb_multi_shard_range(n_samples, n_dev, k, &a, &ns[k]);
                ptrs[k] = ns[k] ? iq + 2 * a : nullptr;
            }
            size_t n = 0;
            int rc = ADSB_OK;
            if (form == 0) rc = adsb_multi_demod_iq(m, iq, n_samples, got.data(), cap, &n);
            else if (form == 1) rc = adsb_multi_demod_iq_device(m, ptrs, ns, got.data(), cap, &n);
            else if (form == 2) rc = adsb_multi_submit_iq_device(m, ptrs, ns);
            else if (rng() % 2) {   // out of memory every device "reads by DMA"
                void *p = nullptr;
                if (adsb_multi_host_alloc(m, n_samples * 4, &p) != ADSB_OK) return 1;
                std::memcpy(p, iq, n_samples * 4);
                pinned.push_back(p);
                rc = adsb_multi_submit_iq(m, static_cast<const int16_t *>(p), n_samples);
            } else {
                rc = adsb_multi_submit_iq(m, iq, n_samples);
            }
            if (form == 0 || form == 1) {
                // (nothing was in flight, and a broken handle was restarted above: expect is 0 or 1 here)
                if (expect == 0) {
                    if (rc != ADSB_OK) {
                        std::fprintf(stderr, "sequence %d step %d: blocking call returned %d (%s)\n", seq_no, step, rc, adsb_multi_last_error(m));
                        bad = true;
                    } else if (!same(got.data(), n, w, "blocking", seq_no, step)) bad = true;
                    total_msgs += n;
                } else {
                    if (rc == ADSB_OK || rc == ADSB_ERR_POISONED) {
                        std::fprintf(stderr, "sequence %d step %d: the blocking capture with the injected failure returned %d\n", seq_no, step, rc);
                        bad = true;
                    }
                    broken = fail_collected = true;
                }
            } else if (expect == 2 && fail_collected) {
                poisoned_returns++;
                if (rc != ADSB_ERR_POISONED) {
                    std::fprintf(stderr, "sequence %d step %d: a submission to a poisoned handle returned %d\n", seq_no, step, rc);
                    bad = true;
                }
            } else if (rc != ADSB_OK) {
                std::fprintf(stderr, "sequence %d step %d: submit returned %d (%s)\n", seq_no, step, rc, adsb_multi_last_error(m));
                bad = true;
            } else {
                pending.push_back(std::move(w));
                pending_expect.push_back(expect);
                if (expect == 1) broken = true;
            }
            if (g_fail_call.load() >= 0 && (form == 0 || form == 1)) {
                std::fprintf(stderr, "sequence %d step %d: the fake's injected failure was never reached\n", seq_no, step);
                bad = true;
            }
        }
        while (!pending.empty() && !bad) collect_one(steps);
        uint64_t ctr[8] = {};
        if (!dead && adsb_multi_selftest_counters(m, ctr) == ADSB_OK) parallel += ctr[3], scored_used += ctr[5], scored_refused += ctr[6];
        if (!dead)
            for (void *p : pinned)
                if (adsb_multi_host_free(m, p) != ADSB_OK) bad = true;
        adsb_multi_destroy(m);   // (a handle with a dead device leaks that device's context, by design; it must still return)
        g_fail_call.store(-1);
        if (bad) {
            std::fprintf(stderr, "FAILED in sequence %d (%d devices, %zu buffers each, depth %d, wait mode %d, fault at step %d%s)\n", seq_no, n_dev, per,
                         depth, wait_mode, fault_at, hang ? ", a hang" : "");
            return 1;
        }
    }
    // ---- allocation failures: around ONE library call at a time the n-th operator new from now throws, on whichever thread
    // it falls (the caller's, a device thread's, a pool worker's, the fake's own code below the library).  Whatever the call
    // and the captures in flight then return, nothing may throw across the ABI, terminate or hang; and once what is in flight
    // has been collected and the stream restarted (adsb_multi_icao_flush, as often as it takes), the handle gives the
    // oracle's list again.
    size_t alloc_rounds = 0, alloc_calls_failed = 0;
    for (int seq_no = 0; seq_no < std::max(8, sequences / 4); seq_no++) {
        const int n_dev = 1 + (int)(rng() % 8);
        const size_t per = 1 + rng() % 3;
        int devices[8];
        for (int k = 0; k < n_dev; k++) devices[k] = (int)(rng() % 8);
        adsb_multi *m = nullptr;
        g_new_countdown.store(rng() % 3 == 0 ? 1 + (long)(rng() % 40) : 0);   // (a third of the creates meet one too)
        int rc = adsb_multi_create(&m, devices, n_dev, per);
        g_new_countdown.store(0);
        if (rc != ADSB_OK) {
            if (m != nullptr || rc != ADSB_ERR_NOMEM) {
                std::fprintf(stderr, "alloc sequence %d: adsb_multi_create returned %d with a handle %p\n", seq_no, rc, (void *)m);
                return 1;
            }
            alloc_calls_failed++;
            if (adsb_multi_create(&m, devices, n_dev, per) != ADSB_OK) return 1;
        }
        (void)adsb_multi_set_wait(m, (int)(rng() % 3));
        (void)adsb_multi_selftest_tune(m, 0, rng() % 2 ? 1 : 0, (uint32_t)(rng() % 3));
        const size_t room = (size_t)n_dev * per;
        bool bad = false;
        for (int round = 0; round < 6 && !bad; round++) {
            const size_t chunks = std::min<size_t>(1 + rng() % room, arena_chunks);
            const size_t first = rng() % (arena_chunks - chunks + 1);
            const size_t n_samples = chunks * kChunkSamples;
            const int16_t *iq = arena.data() + 2 * first * kChunkSamples;
            const void *ptrs[8];
            size_t ns[8];
            for (int k = 0; k < n_dev; k++) {
                size_t a = 0;
                (void)adsb_multi_shard_range(n_samples, n_dev, k, &a, &ns[k]);
                ptrs[k] = ns[k] ? iq + 2 * a : nullptr;
            }
            // a few captures in flight, then ONE call with the countdown armed
            const int in_flight = (int)(rng() % 4);
            for (int k = 0; k < in_flight; k++) (void)adsb_multi_submit_iq_device(m, ptrs, ns);
            size_t n = 0;
            const int which = (int)(rng() % 6);
            g_new_countdown.store(1 + (long)(rng() % 60));
            if (which == 0) rc = adsb_multi_submit_iq_device(m, ptrs, ns);
            else if (which == 1) rc = adsb_multi_pending(m) ? adsb_multi_collect(m, got.data(), cap, &n) : adsb_multi_demod_iq(m, iq, n_samples, got.data(), cap, &n);
            else if (which == 2) rc = adsb_multi_pending(m) ? adsb_multi_collect(m, got.data(), cap, &n) : adsb_multi_demod_iq_device(m, ptrs, ns, got.data(), cap, &n);
            else if (which == 3) rc = adsb_multi_submit_iq(m, iq, n_samples);
            else if (which == 4) rc = adsb_multi_icao_flush(m);
            else {
                void *p = nullptr;
                rc = adsb_multi_host_alloc(m, 4096, &p);
                if (rc == ADSB_OK && adsb_multi_pending(m) == 0) (void)adsb_multi_host_free(m, p);
            }
            // (let the captures in flight meet the rest of the countdown on their device threads, then switch it off)
            std::this_thread::sleep_for(std::chrono::microseconds(rng() % 1500));
            g_new_countdown.store(0);
            alloc_rounds++;
            alloc_calls_failed += rc != ADSB_OK;
            // drain, restart, and the stream must be the oracle's again
            for (int guard = 0; adsb_multi_pending(m) > 0; guard++) {
                (void)adsb_multi_collect(m, got.data(), cap, &n);
                if (guard > 16) {
                    std::fprintf(stderr, "alloc sequence %d round %d: captures stay in flight for ever\n", seq_no, round);
                    bad = true;
                    break;
                }
            }
            int tries = 0;
            while (!bad && (rc = adsb_multi_icao_flush(m)) != ADSB_OK)
                if (++tries > 3) {
                    std::fprintf(stderr, "alloc sequence %d round %d: the restart keeps returning %d (%s)\n", seq_no, round, rc, adsb_multi_last_error(m));
                    bad = true;
                }
            if (bad) break;
            orc_icao_flush(filt.get());
            const size_t k = orc_demod_iq(filt.get(), iq, n_samples, scratch.data(), cap, nullptr);
            std::vector<orc_msg> want(scratch.begin(), scratch.begin() + (long)k);
            rc = adsb_multi_demod_iq_device(m, ptrs, ns, got.data(), cap, &n);
            if (rc != ADSB_OK || !same(got.data(), n, want, "after an allocation failure and the restart", seq_no, round)) {
                std::fprintf(stderr, "alloc sequence %d round %d: after the restart the capture returned %d (%s)\n", seq_no, round, rc, adsb_multi_last_error(m));
                bad = true;
            }
        }
        adsb_multi_destroy(m);
        if (bad) return 1;
    }
    std::printf("allocation failures ok: %zu armed calls, %ld allocations failed, %zu calls returned an error\n", alloc_rounds,
                g_new_failures.load(), alloc_calls_failed);
    std::printf("multi orchestration ok: %d sequences, %zu captures, %zu messages, %zu failures injected, %zu restarts, %zu dead handles, "
                "%zu poisoned returns, %zu blocking handles, %zu captures scored by the pool, %zu shards scored by their device used, %zu refused, %zu failed resets, %zu failed creates\n",
                sequences, total_captures, total_msgs, failures_injected, recoveries, dead_handles, poisoned_returns, blocked, parallel, scored_used,
                scored_refused, failed_resets, failed_creates);
    return 0;
}
