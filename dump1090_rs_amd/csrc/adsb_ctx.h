// adsb_ctx.h -- what the host-side units of libadsb_hip.so share: the context, a pass slot, and the
// internal entry points between them.  Not part of the ABI (include/adsb_hip.h is).
//
//   adsb_context.cpp   create / destroy, the stream pool, settings, diagnostics
//   adsb_pass.cpp      one device pass: what is enqueued on which stream, and every cross-stream edge
//                      (DESIGN.md section 5b lists them), submit, the blocking entry points
//   adsb_collect.cpp   waiting for a pass, checksums, the overflow fallback
//   adsb_replay_host.cpp  the ordered host replay and the other host-only entry points (no HIP: also built by g++ under sanitizers)
//   adsb_ring.cpp      the pinned streaming ring
//   adsb_shard.cpp     the two-phase shard calls
//   adsb_selftest.cpp  stage lists and digests for the tests
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/adsb_hip.h"
#include "adsb_device.h"
#include "adsb_scan_geometry.h"
#include "adsb_replay_host.h"
#include "adsb_tables.h"
#include "mode_s_host.hpp"

using namespace adsb;

// One submission in flight: what was asked, and the pinned host side of its results.
struct Slot {
    bool busy = false;
    bool flush_before = false;  // an icao_flush precedes this pass (host filter flushed at collect)
    bool from_mag = false;
    const void *src = nullptr;
    uint64_t n_samples = 0;
    uint32_t n_chunks = 0;
    // pinned, mapped host memory the records kernel writes straight into (no copy commands
    // on the stream): *_dev are the device-side addresses of the same allocations
    Summary *h_sum = nullptr, *h_sum_dev = nullptr;
    TrialRecord *h_rec = nullptr, *h_rec_dev = nullptr;  // hits_cap entries
    uint32_t hits_cap = 0;  // entries in d_hits / h_rec (the fallback's lists are larger than the slot's own)
    // device side of the slot: its own counters, AP list and hit list, so that the match /
    // records tail of this pass (tail stream) can run while the next pass's scan (scan
    // stream) fills the other slot's
    Counters *d_ctr = nullptr;
    uint64_t *d_ap = nullptr, *d_hits = nullptr;
    uint32_t *d_hit_fields = nullptr;  // per hit-list slot: the scan's five bit-class fields of a self-validating hit (ScanParams::hit_fields)
    // device-side ordering of the hit list: per-buffer counts and their prefix (max_chunks + 1
    // each), and the second list the counting sort scatters into
    uint32_t *d_order_cnt = nullptr, *d_order_base = nullptr;
    uint64_t *d_order_tmp = nullptr;
    // device-side scoring: the messages, the filter additions and their summary, in mapped host memory
    adsb_msg *h_msgs = nullptr, *h_msgs_dev = nullptr;
    uint32_t *h_adds = nullptr, *h_adds_dev = nullptr;
    ScoreSummary *h_ssum = nullptr, *h_ssum_dev = nullptr;
    // a pass the library finished ahead of the caller's adsb_collect (park_pending): its result waits here
    bool parked = false;
    int park_rc = 0;
    std::vector<adsb_msg> parked_msgs;
    adsb_stats parked_stats{};
    ScoreDev score{};             // this slot's scoring buffers (the exact bitmap in it is the context's)
    hipEvent_t recorded = nullptr;  // this pass's records kernel has finished (k_score may start; the superset
                                    // bitmap it matched against may be cleared)
    hipStream_t tail_q = nullptr;   // the stream its match / order / records ran on
    hipStream_t scan_q = nullptr;   // ... and its scan
    bool device_scored = false;   // this pass went through k_score / k_emit
    uint64_t score_epoch = 0;     // ... against the filter history of this epoch
    uint32_t *d_carry = nullptr;   // carry-over mode: the kCarrySamples samples before this pass's input
                                   // (kept until the slot is reused: the overflow fallback re-reads it)
    hipEvent_t scanned = nullptr;  // scan stream: this pass's scan has finished
    uint32_t seq = 0;   // what the records kernel writes into h_sum->seq (sanity check)
    uint64_t scan_seq = 0;  // running number of the pass (ms_scan_exclusive: was the previous scan the previous pass?)
    // A pass of a few buffers is ONE launch (k_scan_fast<.., FUSED>: scan, match, records): no event is
    // recorded behind it -- the host sees its summary land in mapped memory -- and it does not wait for a
    // one-launch pass still running on the other scan stream.  `unsynced_from`: the oldest pass that was in
    // flight then (0: none); if any pass from there on turns out to have taught the filter a new address,
    // this pass's match may have missed it and the pass is redone through the three-launch path.
    bool fused = false;
    uint64_t unsynced_from = 0;
    hipStream_t fused_q = nullptr;  // the stream the slot's latest one-launch pass ran on: its summary reaches the host a
                                    // moment before the launch retires, so a pass that reuses the slot's lists from
                                    // another stream orders itself behind that stream first
    hipEvent_t done = nullptr;  // no timing, no system fence: results are written through
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int profiled = 0;  // profiling level the pass was enqueued with
    bool redo = false;  // enqueued by collect_oldest (overflow fallback, rematch): keeps the pass's number, times itself with
                        // the context's redo events, and stays out of ms_scan_exclusive's frontier
};

// Passes in flight: 4 and the device never waits for the host between large passes (3 do for sparse
// streams; a dense one has a longer tail).  A context created for passes of a few buffers gets 8: its
// passes are one launch each and mostly latency (a link read, one round of tiles, a one-workgroup tail),
// four run side by side on the four scan streams and four more are queued behind them, so that a launch
// never waits for the host (adsb_ctx::n_slots).
constexpr int kSlots = ADSB_MAX_IN_FLIGHT_SMALL;  // what the arrays hold
constexpr int kBitmaps = kSlots + 1;
#ifndef ADSB_SCAN_STREAMS
#define ADSB_SCAN_STREAMS 4
#endif
constexpr int kScanStreams = ADSB_SCAN_STREAMS;
constexpr int kScanEvRing = kSlots + 3;  // scan start / stop event pairs in rotation (finish_pass: ms_scan_exclusive)

constexpr size_t kTimelineWords = (size_t)adsb::kApSegments * 8 * 8;  // 8 waves x 8 counters per workgroup

struct adsb_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    bool private_streams = false;   // tuning builds only: this context created (and destroys) its streams itself
    hipStream_t stream = nullptr;
    bool own_stream_dirty = false;  // the library has enqueued something on own_stream that the next pass reads (a host-pointer call's copy)
    int profiling = 1;  // 0: no events, 1: around the scan kernel, 2: around every kernel
    bool flush_pending = true;  // consumed by the next pass: it switches to the clean spare bitmap
    uint32_t stagger_ticks = 0;
    int debug_stop = 0;  // ADSB_DEBUG_STOP: profiling aid, breaks results when non-zero
    unsigned long long *d_timeline = nullptr;  // ADSB_TIMELINE=1: 8 blocks x 8 tiles x 8 stamps
    size_t max_chunks = 0;

    char *h_block = nullptr, *h_block_dev = nullptr;   // the host side of every slot (summaries, records, messages): one pinned allocation
    void *d_stage = nullptr;  // IQ staging for host-pointer calls (lazy)
    size_t stage_bytes = 0;
    void *h_stage = nullptr, *h_stage_dev = nullptr;  // ... pinned and mapped, for calls of a few buffers: the pass reads it in place
    size_t h_stage_bytes = 0;
    // Address bitmaps in rotation (one more than passes in flight): icao_flush moves on to the
    // next (clean) one, the retired one is cleared by that pass's records kernel and comes back
    // into use kSlots flushes later -- a pass that far ahead cannot even be submitted before the
    // pass that cleared it has been collected, so neither a reset launch nor a cross-stream wait
    // is ever needed.
    uint32_t *d_bitmap[kBitmaps] = {};
    int cur_bitmap = 0;
    uint32_t bitmap_lg = kFullBitmapLg;   // kSmallBitmapLg in a context for passes of a few buffers: folded bitmaps, no device-side scoring
    hipStream_t score_stream = nullptr;  // k_score / k_emit of the device-scored passes, in pass order
    hipStream_t tail_stream = nullptr;  // match + records of pass i run here, beside scan i+1
    // The scans run on two internal streams, alternating between consecutive pipelined passes:
    // those do not depend on each other (own lists and counters per slot; bits another scan
    // adds to the bitmap meanwhile only widen the superset), so the next scan's workgroups
    // fill the CUs as the previous scan's persistent grid drains instead of waiting ~13 us
    // behind an in-order queue's end-of-kernel barrier.  `stream` (the caller's) only orders
    // the input: each scan waits for the point `stream` had reached at submit.
    // (four of them: three-launch passes alternate between the first two; one-launch passes of a few
    // buffers, whose kernels are mostly latency -- a link read, one round of tiles, a one-workgroup tail --
    // rotate over all four so that three or four of them overlap)
    hipStream_t scan_stream[kScanStreams] = {};
    int n_scan_streams = 2;  // 4 for a context of a few buffers per pass; a large one keeps the two it always had
                             // (every further stream is another hardware queue for the tail streams to share)
    hipEvent_t prev_scanned = nullptr;      // the latest submission's scan-end event and the stream it is on
    hipStream_t prev_scan_stream = nullptr;
    bool prev_inline = false;               // ... and whether its match ran there rather than on the tail stream
    bool prev_fused = false;                // ... as part of the scan's own launch (no event behind it: prev_scanned is stale)
    hipEvent_t lazy_ev = nullptr;           // recorded on a stream at the moment somebody has to wait for a one-launch pass on it
    uint64_t last_new_insert_seq = 0;       // the latest pass whose replay put a NEW address into the filter
    // folded bitmaps (contexts for passes of a few buffers): the pass behind the latest icao_flush clears the next
    // bitmap itself when it starts; a pass submitted after it shares that bitmap, so while the flushed pass is still
    // in flight on ANOTHER stream the later one is ordered behind it (it must not set a bit the clear then wipes).
    // epoch_first_seq: that pass's number -- what older passes teach the filter is no business of this epoch's.
    hipStream_t fresh_q = nullptr;
    uint64_t fresh_seq = 0, epoch_first_seq = 0;
    int fresh_slot = -1;
    uint64_t rematches = 0;                 // one-launch passes redone because a pass in flight beside them did
    uint32_t order_polls = 200;             // ScanParams::order_polls (adsb_selftest_set_order_polls)
    const unsigned long long *next_src_ready = nullptr;  // ScanParams::src_ready of the next pass enqueued (adsb_demod_iq)
    // adsb_host_register: the caller's buffers, pinned and mapped (adsb_demod_iq reads samples inside them in place)
    struct HostRange {
        char *base = nullptr;
        size_t bytes = 0;
        char *dev = nullptr;
    };
    std::vector<HostRange> host_ranges;
    hipStream_t input_on_stream = nullptr;   // the next pass's input is a copy queued on this stream (the ring): it must run behind it
    bool next_src_host = false;    // ... reads host memory in place (ScanParams::src_host)
    hipEvent_t input_ready[kScanStreams] = {};  // per scan stream: `stream` at submit (the caller's IQ is complete)
    uint32_t *d_tables = nullptr;
    uint32_t hits_cap = 0, ap_cap = 0, seg_cap = 0;
    // Lists that hold the worst case of one buffer (every position sliced, five trials each), for
    // the buffer-by-buffer fallback through the reference-shaped kernel.  58 MB, most of it pinned
    // host memory: allocated the first time a pass overflows the normal lists -- a receiver's
    // stream never gets there, and a process with hundreds of small contexts stays small.
    struct Fallback {
        uint64_t *d_hits = nullptr, *d_dap = nullptr;
        TrialRecord *h_rec = nullptr, *h_rec_dev = nullptr;
    } fb;

    Slot slot[kSlots];
    int n_slots = ADSB_MAX_IN_FLIGHT;  // slots in use (ADSB_MAX_IN_FLIGHT, or _SMALL for contexts of a few buffers)
    int n_bitmaps = ADSB_MAX_IN_FLIGHT + 1;
    // start / stop events of the scans, in a ring one longer than the passes in flight: when pass N
    // is collected the stop event of pass N-1 is still its own (ms_scan_exclusive)
    hipEvent_t scan_ev[kScanEvRing][2] = {};
    hipEvent_t redo_ev[2] = {};      // start / stop of a pass collect_oldest runs again (blocking: one pair is enough)
    hipEvent_t last_stop = nullptr;  // stop event of the pass collected last, and its number
    uint64_t last_scan_seq = 0;
    uint64_t scan_counter = 0;
    uint64_t submitted = 0, collected = 0;  // passes enqueued / finished (replayed) by the library
    uint64_t delivered = 0;                 // passes handed to the caller (<= collected: park_pending)
    // Dense input (thousands of trial records per pass) is ordered and scored on the device; sparse
    // input is not worth the extra launches on the tail stream, the host does it in microseconds.
    // Decided from the last pass finished (a stream's density changes slowly).
    bool dense_mode = false;
    uint32_t next_seq = 1;

    // streaming ring (adsb_ring_*): per slot a pinned host buffer the caller fills and a device staging
    // buffer; a slot is read in place by its pass or copied on that pass's scan stream in front of it
    // (adsb_ring.cpp: adsb_ring_submit) while the passes on the other scan streams compute
    struct RingSlot {
        int16_t *h_iq = nullptr;
        void *h_iq_dev = nullptr;  // the same pinned buffer as the device addresses it
        void *d_iq = nullptr;
    } ring[kSlots];
    size_t ring_samples = 0;
    char *ring_h_block = nullptr, *ring_d_block = nullptr;   // all slots' pinned / staging buffers: one allocation each

    bool carry_over = false;  // adsb_set_carry_over: opt-in, not the reference's semantics
    uint32_t *d_carry_next = nullptr;  // the end of the latest submission's input: the next one's lead-in

    // sharded capture: adsb_shard_scan / adsb_shard_finish park one shard (in slot 0) between its two phases;
    // the multi-GPU entry points (adsb_multi.cpp) keep up to n_slots shards of consecutive captures in flight,
    // one per slot (adsb_shard.cpp: shard_begin ... shard_records)
    bool shard_active = false;
    struct ShardJob {
        bool active = false;      // phase 1 enqueued, phase 2 not collected yet
        bool by_chunk = false;    // the shard overflowed the fast scan's lists: both phases go buffer by buffer
        bool waiting = false;     // a phase's launches are out and its summary has not been seen yet
        bool ran = false;         // Slot::recorded has been recorded behind a second phase of this slot
        ScanParams p{};
        uint32_t *retired = nullptr;   // the bitmap an icao_flush in front of this shard retired (phase 2 cleans it)
        std::vector<TrialRecord> chunk_records;   // by_chunk: the second phase's records
        uint64_t n_cand = 0, n_ap = 0;
        hipStream_t scan_q = nullptr;
        uint32_t *h_addrs = nullptr, *h_addrs_dev = nullptr;   // the other shards' addresses, in mapped host memory
                                                               // (k_set_addresses reads them in place: no copy command)
        bool fresh_list = false;  // phase 1 is the scan alone: it lists the addresses its trials can add (ScanParams::fresh)
        uint32_t *h_fresh = nullptr, *h_fresh_dev = nullptr;   // ... here (mapped host memory, kShardAddrCap of them)
        uint32_t *d_fresh_seen = nullptr;                      // ... each once: 2^24 bits, cleared in front of the scan
        // device-side scoring of a dense stream's shards (adsb_multi): the second phase ends with k_score / k_emit against
        // the context's exact bitmap + `earlier` (d_fresh_seen again, refilled with what the shards before this one add)
        bool scored = false;          // k_score / k_emit follow this shard's records kernel
        bool wait_score = false;      // ... and have been launched: the second phase ends with k_emit's summary
        bool result_scored = false;   // ... and took the shard (ScoreSummary::scored): its records stayed in device memory
        bool exact_flush = false;     // an icao_flush precedes the shard: its second phase starts from the other, cleared, exact bitmap
        uint32_t *exact = nullptr;    // the exact bitmap the shard scores against and its capture's additions go into
        uint32_t *h_earlier = nullptr, *h_earlier_dev = nullptr;   // the earlier shards' additions, in mapped host memory
        int addr_half = 0;            // h_addrs holds two lists: the commit of one use may still read while the next is written
        hipEvent_t addr_read[2] = {nullptr, nullptr};   // ... recorded behind the commit that reads half h (score stream)
    } shard[kSlots];
    uint64_t shard_jobs = 0;      // shards begun (their scans alternate between the first two scan streams)
    uint32_t shard_fresh_cap = 0; // how many fresh addresses a shard may list before it falls back to reading them out of
                                  // its records (0: kShardAddrCap; smaller only in tests of that fallback)
    uint64_t shard_fresh_fallbacks = 0, shard_device_ordered = 0, shard_device_scored = 0;   // (counters for the tests: adsb_multi_selftest_counters)
    bool shard_scoring = false;   // adsb_multi's contexts: dense shards are scored on the device (off: adsb_multi_selftest_tune)
    bool shard_dense = false;     // the shards of this context leave >= 8 records per buffer: their second phase hands the
                                  // records over in replay order (device-side ordering, as dense single-stream passes do)
    uint32_t *d_addrs = nullptr;
    size_t addrs_cap = 0;

    IcaoFilter filter;
    Crc24 crc;
    adsb_stats stats{};
    std::string last_error;
    // the messages of a call whose `out` was too small (ADSB_ERR_CAPACITY): the pass is consumed
    // and the filter has moved on, so they are kept for adsb_fetch_messages
    uint64_t host_sorts = 0;  // passes whose records the host had to put in order itself
    uint64_t host_replays = 0;  // passes the host scored itself (small passes, fallbacks, full filter ...)
    // Device-side scoring (adsb_device.h: ScoreDev).  The exact bitmap follows the filter pass by pass
    // on the tail stream; the host's own filter follows at collect time from the additions each pass
    // reports.  Whenever the host scores a pass itself the two part ways: `score_epoch` moves on, which
    // disowns the device results of passes already in flight, and device scoring resumes once the
    // context is idle and the bitmap has been rebuilt from the host's table.
    ScoreDev score{};
    uint32_t *exact_bm[2] = {nullptr, nullptr};  // the exact bitmap in use and the clean one an icao_flush switches to
    int cur_exact = 0;
    bool exact_valid = false;
    uint64_t score_epoch = 0;
#ifdef ADSB_TUNING
    double t_wait = 0, t_replay = 0, t_enqueue = 0;  // host seconds (ADSB_HOST_TIMES prints them at destroy)
    double ht_s[16] = {};                            // ... and call by call (HT() below)
    uint64_t ht_n[16] = {};
#endif
    std::vector<adsb_msg> undelivered;
    bool has_undelivered = false;
};

namespace adsb {
namespace host {

constexpr uint32_t kWorstPerChunk = 5u * kChunkSamples;  // every j sliced, 5 trials each
constexpr uint32_t kInlineTailChunks = 16;               // passes this small keep their tail on the scan stream

inline int fail(adsb_ctx *c, hipError_t e, const char *what)
{
    if (c) {
        c->last_error = std::string(what) + ": " + hipGetErrorString(e);
    }
    return ADSB_ERR_HIP;
}

#define HIP_TRY(ctx, call)                                                \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return ::adsb::host::fail((ctx), e_, #call); \
    } while (0)

// Every entry point runs on its context's device and leaves the calling thread's current device as it found it: a
// host with several GPUs (torch, or a Rust main loop that drives a context per device from one thread) must not find
// its current device moved under it by a library call.
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    bool restore = false;
    explicit DeviceGuard(int want)
    {
        // (a process that sees one device has nothing to switch: the two runtime calls were a fifth of what a
        // one-buffer pass costs the submitting thread)
        static const int n_devices = [] {
            int n = 0;
            return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
        }();
        if (n_devices == 1 && want == 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) {
            err = hipSetDevice(want);
            restore = err == hipSuccess && prev >= 0;
        }
    }
    ~DeviceGuard()
    {
        if (restore) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define ADSB_ON_DEVICE(ctx)                                    \
    ::adsb::host::DeviceGuard dev_guard_((ctx)->device);       \
    if (dev_guard_.err != hipSuccess) return ::adsb::host::fail((ctx), dev_guard_.err, "hipSetDevice")

// Host seconds per kind of HIP call of a pass (tuning builds only; ADSB_HOST_TIMES=1 prints the table when the
// context is destroyed: tools/hosttime.py).  `{ HT(c, HT_SCAN_LAUNCH); launch ...; }`
enum HtKey { HT_RING_MEMCPY, HT_RING_EVENT, HT_IN_READY, HT_SCAN_LAUNCH, HT_EV_SCANNED, HT_MATCH_LAUNCH, HT_RECORDS_LAUNCH,
             HT_EV_DONE, HT_SYNC, HT_VERIFY, HT_REPLAY, HT_COUNT };
#ifdef ADSB_TUNING
struct HostTimer {
    adsb_ctx *c;
    int k;
    std::chrono::steady_clock::time_point t0;
    HostTimer(adsb_ctx *c_, int k_) : c(c_), k(k_), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer()
    {
        c->ht_s[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        c->ht_n[k]++;
    }
};
#define HT(c, k) ::adsb::host::HostTimer ht_timer_##k((c), ::adsb::host::k)
#else
#define HT(c, k) do {} while (0)
#endif

// adsb_pass.cpp
int ensure_fallback(adsb_ctx *c);
int fallback_slot(adsb_ctx *c, const Slot &sl, Slot &tmp);
// input_done: the event behind which the input is complete; kInputReadyNow: it already is (host-visible
// pinned memory the caller has filled); null: wherever `stream` stands now.  no_fuse: three launches even
// for a pass of a few buffers.
inline hipEvent_t input_ready_now() { return reinterpret_cast<hipEvent_t>(static_cast<uintptr_t>(1)); }
bool one_launch_pass(const adsb_ctx *c, uint32_t n_chunks);
hipStream_t next_scan_stream(const adsb_ctx *c, uint32_t n_chunks);
int enqueue_pass(adsb_ctx *c, Slot &sl, const void *d_src, bool from_mag, uint64_t n_samples, uint32_t n_chunks,
                 bool inline_tail = false, bool lead_from_src = false, bool advance_carry = true,
                 bool force_simple = false, hipEvent_t input_done = nullptr, bool no_fuse = false);
int wait_for_tail_of(adsb_ctx *c, hipStream_t waiter, Slot &other);
// The blocking entry points that launch on `stream` with slot 0's lists and counters without enqueue_pass (shard
// phases, self-tests): behind the one-launch pass that used the slot last, whose summary reaches the host a
// moment before its last workgroup has zeroed the counters (edge 0 of DESIGN.md section 5b).
int order_behind_fused(adsb_ctx *c, Slot &sl, hipStream_t waiter);
inline int order_behind_slot0(adsb_ctx *c) { return order_behind_fused(c, c->slot[0], c->stream); }
int resync_exact(adsb_ctx *c);
int reseed_bitmap_from_filter(adsb_ctx *c);
int submit(adsb_ctx *c, const void *d_src, bool from_mag, uint64_t n_samples, bool inline_tail = false,
           hipEvent_t input_done = nullptr);
int run_sync(adsb_ctx *c, const void *d_src, bool from_mag, uint64_t n_samples, std::vector<adsb_msg> &out,
             hipEvent_t input_done = nullptr);
int demod_device(adsb_ctx *c, const void *d_iq, uint64_t n_samples, std::vector<adsb_msg> &out);
int ensure_stage(adsb_ctx *c, size_t bytes);
int ensure_host_stage(adsb_ctx *c, size_t bytes);

// adsb_collect.cpp
int verify_records(adsb_ctx *c, const Summary *sum, const TrialRecord *rec, size_t n);
int finish_pass(adsb_ctx *c, Slot &sl, uint64_t chunk_offset, adsb_stats &st, std::vector<adsb_msg> &out);
int collect_oldest(adsb_ctx *c, std::vector<adsb_msg> &out);
int park_pending(adsb_ctx *c);
int collect_next(adsb_ctx *c, std::vector<adsb_msg> &out);
int deliver(adsb_ctx *c, std::vector<adsb_msg> &msgs, adsb_msg *out, size_t cap, size_t *n_out);
bool summary_landed(const Summary *s, uint32_t seq);

// adsb_shard.cpp: one shard of a capture in slot k, phase by phase, nothing blocking but shard_phase_wait
// (and the buffer-by-buffer fallback of a shard that overflows the lists).  Order of calls per slot:
// shard_begin -> [landed] -> shard_learned -> shard_match -> [landed] -> shard_records.
constexpr size_t kShardAddrCap = 16384;   // addresses per launch of k_set_addresses from mapped memory
int shard_begin(adsb_ctx *c, int k, const void *d_iq, uint64_t n_samples, bool fresh_list = false);
bool shard_phase_landed(adsb_ctx *c, int k);
int shard_phase_wait(adsb_ctx *c, int k);
// for a caller that polls (adsb_multi's device threads, once a phase has been out for a while): 1 the phase has
// landed, 0 its launches are still running, negative: a stream reports an error, or every stream the phase ran on
// is idle and no whole summary ever arrived (c->last_error says which)
int shard_phase_check(adsb_ctx *c, int k);
// every shard slot, every address superset and the device-side copy of the filter back to what adsb_create left
// (blocking; nothing of this context may be in flight): how an adsb_multi whose capture failed starts over
int shard_reset(adsb_ctx *c);
int shard_learned(adsb_ctx *c, int k, std::vector<uint32_t> &addrs);
int shard_match(adsb_ctx *c, int k, const uint32_t *extra, size_t n_extra, const uint32_t *earlier = nullptr, size_t n_earlier = 0);
// after phase 2 has landed, a shard the device scored: its messages (chunk = buffer index within the shard) and the values
// its replay hands to icao_filter_add, in order -- in the slot's mapped memory, valid until its next shard_begin.
// false: the shard was not scored (or its result cannot be used): take its records.
bool shard_scored_result(adsb_ctx *c, int k, adsb_msg **msgs, size_t *n_msgs, const uint32_t **adds, size_t *n_adds);
// ... and, when a scored shard's result cannot be used after all (a filter table about to fill up): its records out of
// device memory into the slot's host buffer (blocking)
int shard_fetch_records(adsb_ctx *c, int k, const TrialRecord **rec, size_t *n_out);
int shard_records(adsb_ctx *c, int k, const TrialRecord **rec, size_t *n);

}  // namespace host
}  // namespace adsb
