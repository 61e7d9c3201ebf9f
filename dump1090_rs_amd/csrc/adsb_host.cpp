// adsb_host.cpp -- context, C ABI (include/adsb_hip.h) and the ordered host replay.
//
// The functions here mirror the reference's library API for the path
// (src/utils.rs:43 to_mag, src/demod_2400.rs:115 demodulate2400,
// src/icao_filter.rs:11 icao_flush); what each one replaces is listed in the header.
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
    bool device_scored = false;   // this pass went through k_score / k_emit
    uint64_t score_epoch = 0;     // ... against the filter history of this epoch
    uint32_t *d_carry = nullptr;   // carry-over mode: the kCarrySamples samples before this pass's input
                                   // (kept until the slot is reused: the overflow fallback re-reads it)
    hipEvent_t scanned = nullptr;  // scan stream: this pass's scan has finished
    uint32_t seq = 0;   // what the records kernel writes into h_sum->seq (sanity check)
    uint64_t scan_seq = 0;  // running number of the pass (ms_scan_exclusive: was the previous scan the previous pass?)
    hipEvent_t done = nullptr;  // no timing, no system fence: results are written through
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int profiled = 0;  // profiling level the pass was enqueued with
};

constexpr int kSlots = ADSB_MAX_IN_FLIGHT;  // 4: the device never waits for the host between passes (3 do for sparse streams; a dense one has a longer tail)
constexpr int kBitmaps = kSlots + 1;
constexpr int kScanEvRing = kSlots + 3;  // scan start / stop event pairs in rotation (finish_pass: ms_scan_exclusive)

constexpr size_t kTimelineWords = (size_t)adsb::kApSegments * 8 * 8;  // 8 waves x 8 counters per workgroup

struct adsb_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int profiling = 1;  // 0: no events, 1: around the scan kernel, 2: around every kernel
    bool flush_pending = true;  // consumed by the next pass: it switches to the clean spare bitmap
    uint32_t stagger_ticks = 0;
    int debug_stop = 0;  // ADSB_DEBUG_STOP: profiling aid, breaks results when non-zero
    unsigned long long *d_timeline = nullptr;  // ADSB_TIMELINE=1: 8 blocks x 8 tiles x 8 stamps
    size_t max_chunks = 0;

    void *d_stage = nullptr;  // IQ staging for host-pointer calls (lazy)
    size_t stage_bytes = 0;
    uint16_t *d_mag = nullptr;  // one MagnitudeBuffer.data
    // Address bitmaps in rotation (one more than passes in flight): icao_flush moves on to the
    // next (clean) one, the retired one is cleared by that pass's records kernel and comes back
    // into use kSlots flushes later -- a pass that far ahead cannot even be submitted before the
    // pass that cleared it has been collected, so neither a reset launch nor a cross-stream wait
    // is ever needed.
    uint32_t *d_bitmap[kBitmaps] = {};
    int cur_bitmap = 0;
    hipStream_t score_stream = nullptr;  // k_score / k_emit of the device-scored passes, in pass order
    hipStream_t tail_stream = nullptr;  // match + records of pass i run here, beside scan i+1
    // The scans run on two internal streams, alternating between consecutive pipelined passes:
    // those do not depend on each other (own lists and counters per slot; bits another scan
    // adds to the bitmap meanwhile only widen the superset), so the next scan's workgroups
    // fill the CUs as the previous scan's persistent grid drains instead of waiting ~13 us
    // behind an in-order queue's end-of-kernel barrier.  `stream` (the caller's) only orders
    // the input: each scan waits for the point `stream` had reached at submit.
    hipStream_t scan_stream[2] = {nullptr, nullptr};
    hipEvent_t prev_scanned = nullptr;      // the latest submission's scan-end event and the stream it is on
    hipStream_t prev_scan_stream = nullptr;
    bool prev_inline = false;               // ... and whether its match ran there rather than on the tail stream
    hipEvent_t input_ready[2] = {nullptr, nullptr};  // per slot: `stream` at submit (the caller's IQ is complete)
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
    // start / stop events of the scans, in a ring one longer than the passes in flight: when pass N
    // is collected the stop event of pass N-1 is still its own (ms_scan_exclusive)
    hipEvent_t scan_ev[kScanEvRing][2] = {};
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

    // streaming ring (adsb_ring_*): per slot a pinned host buffer the caller fills and a
    // device staging buffer; the H2D copy of one slot runs on its own stream while the
    // other slot's pass computes
    struct RingSlot {
        int16_t *h_iq = nullptr;
        void *d_iq = nullptr;
        hipEvent_t copied = nullptr;
    } ring[kSlots];
    size_t ring_samples = 0;
    hipStream_t copy_stream = nullptr;
    hipStream_t copy_stream_spare = nullptr;  // a pooled copy stream this context has not needed (yet)

    bool carry_over = false;  // adsb_set_carry_over: opt-in, not the reference's semantics
    uint32_t *d_carry_next = nullptr;  // the end of the latest submission's input: the next one's lead-in

    // sharded capture (adsb_shard_scan / adsb_shard_finish): the pass parked between its two phases
    bool shard_active = false;
    bool shard_by_chunk = false;  // the shard overflowed the fast scan's lists: both phases go chunk by chunk
    ScanParams shard_params{};
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
#endif
    std::vector<adsb_msg> undelivered;
    bool has_undelivered = false;
};

namespace {

constexpr uint32_t kWorstPerChunk = 5u * kChunkSamples;  // every j sliced, 5 trials each
constexpr uint32_t kInlineTailChunks = 16;               // passes this small keep their tail on the scan stream

int fail(adsb_ctx *c, hipError_t e, const char *what)
{
    if (c) {
        c->last_error = std::string(what) + ": " + hipGetErrorString(e);
    }
    return ADSB_ERR_HIP;
}

#define HIP_TRY(ctx, call)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return fail((ctx), e_, #call); \
    } while (0)

// Ordered replay (src/demod_2400.rs:149-207 with mode_s scoring): records sorted by
// (chunk, j, try_phase); per (chunk, j) the best trial by strictly-greater score
// starting from -2 wins and is emitted when its score is >= 0.
inline uint64_t replay_key(const TrialRecord &r)
{
    return (uint64_t)r.chunk << 32 | (uint64_t)(r.j_tp & 0xFFFFFFu) << 8 | (r.j_tp >> 24);
}

void replay(IcaoFilter &filter, const Crc24 &crc, TrialRecord *rec, size_t n, uint64_t chunk_offset,
            std::vector<adsb_msg> &out, uint64_t *host_sorts = nullptr)
{
    // order = (chunk, j, try_phase).  Large passes arrive in that order from the device; anything
    // else is put in order here -- the records stay where they are (they may sit in mapped host
    // memory), only 16-byte (key, index) pairs are sorted.
    struct Ref {
        uint64_t key;
        uint32_t idx;
    };
    bool sorted = true;
    for (size_t i = 1; i < n && sorted; i++) sorted = replay_key(rec[i - 1]) <= replay_key(rec[i]);
    std::vector<Ref> order;
    if (!sorted) {
        if (host_sorts) ++*host_sorts;
        order.resize(n);
        uint64_t all_or = 0;
        for (size_t i = 0; i < n; i++) {
            order[i] = {replay_key(rec[i]), (uint32_t)i};
            all_or |= order[i].key;
        }
        // LSD radix sort, 11 bits a pass, skipping digits no key uses (a device pass has
        // chunk < 2^19, j < 2^18, try_phase < 16: four passes); stable
        std::vector<Ref> tmp(n);
        Ref *src = order.data(), *dst = tmp.data();
        for (int shift = 0; shift < 64; shift += 11) {
            if (((all_or >> shift) & 0x7FFu) == 0) continue;
            uint32_t count[2048] = {0};
            for (size_t i = 0; i < n; i++) count[(src[i].key >> shift) & 0x7FFu]++;
            uint32_t at = 0;
            for (uint32_t &c : count) {
                const uint32_t k = c;
                c = at;
                at += k;
            }
            for (size_t i = 0; i < n; i++) dst[count[(src[i].key >> shift) & 0x7FFu]++] = src[i];
            std::swap(src, dst);
        }
        if (src != order.data()) order.swap(tmp);
    }
    auto at = [&](size_t i) -> const TrialRecord & { return sorted ? rec[i] : rec[order[i].idx]; };
    size_t i = 0;
    while (i < n) {
        const uint64_t pos = replay_key(at(i)) >> 8;  // (chunk, j)
        const TrialRecord *best = nullptr;
        Score best_score{false, (int)ADSB_MODES_SHORT_MSG_BYTES, -2};
        for (; i < n; i++) {
            const TrialRecord &r = at(i);
            if ((replay_key(r) >> 8) != pos) break;
            // records built on the device bring the CRC residual along (pad bit 0) and the filter
            // hash of the value their DF asks about (pad bit 1, hash in bits 4..15)
            const Score s = (r.pad & 1) ? score_modes_message(filter, (uint32_t)(r.power >> 40), r.msg,
                                                              (r.pad & 2) ? (int)(r.pad >> 4) : -1)
                                        : score_modes_message(filter, crc, r.msg);
            if (!s.some || s.value <= best_score.value) continue;
            best = &r;
            best_score = s;
        }
        if (!best || best_score.value < 0) continue;
        adsb_msg m{};
        std::memcpy(m.msg, best->msg, 14);
        m.len = (uint8_t)best_score.len;
        m.score = best_score.value;
        m.try_phase = (uint8_t)(best->j_tp >> 24);
        // demod_2400.rs:191-198: signal_len = 14*12/5 = 33 (the same three divisions, in this order)
        const double signal_power = (double)(best->power & ((1ull << 40) - 1)) / 65535.0 / 65535.0;
        m.signal_level = signal_power / 33.0;
        m.j = (uint32_t)(pos & 0xFFFFFFu);
        m.chunk = chunk_offset + (pos >> 24);
        out.push_back(m);
    }
}

int ensure_fallback(adsb_ctx *c)
{
    auto &fb = c->fb;
    if (fb.h_rec_dev) return ADSB_OK;
    if (!fb.d_hits) HIP_TRY(c, hipMalloc((void **)&fb.d_hits, (size_t)kWorstPerChunk * sizeof(uint64_t)));
    if (!fb.d_dap) HIP_TRY(c, hipMalloc((void **)&fb.d_dap, (size_t)kWorstPerChunk * sizeof(uint64_t)));
    if (!fb.h_rec)
        HIP_TRY(c, hipHostMalloc((void **)&fb.h_rec, (size_t)kWorstPerChunk * sizeof(TrialRecord), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(c, hipHostGetDevicePointer((void **)&fb.h_rec_dev, fb.h_rec, 0));
    return ADSB_OK;
}

// `sl` with the worst-case lists in place of its own: what a one-buffer fallback pass runs on
// (same counters, AP list, summary, events and carry as the pass it redoes).
int fallback_slot(adsb_ctx *c, const Slot &sl, Slot &tmp)
{
    if (int rc = ensure_fallback(c)) return rc;
    tmp = sl;
    tmp.d_hits = c->fb.d_hits;
    tmp.h_rec = c->fb.h_rec;
    tmp.h_rec_dev = c->fb.h_rec_dev;
    tmp.hits_cap = kWorstPerChunk;
    return ADSB_OK;
}

int resync_exact(adsb_ctx *c);
int park_pending(adsb_ctx *c);

// Enqueue one device pass over n_chunks chunks starting at d_src into `sl`:
// reset -> scan -> dense -> match -> records -> D2H of the summary and the first records.
int enqueue_pass(adsb_ctx *c, Slot &sl, const void *d_src, bool from_mag, uint64_t n_samples,
                 uint32_t n_chunks, bool inline_tail = false, bool lead_from_src = false,
                 bool advance_carry = true, bool force_simple = false, hipEvent_t input_done = nullptr)
{
    // Passes of many buffers of a dense stream hand their hits over in (buffer, j, try_phase) order and
    // scored; a small pass is all launch overhead and a sparse one leaves a few hundred records that
    // the host sorts and scores in no time; the worst-case lists of the fallback are the host's too.
    const bool order_on_device = !force_simple && n_chunks > kInlineTailChunks && sl.hits_cap == c->hits_cap && c->dense_mode;
    if (order_on_device && c->score.si && !c->exact_valid) {
        // the device's copy of the filter can only be rebuilt from the host's once every pass in
        // flight has been replayed: finish them now (their results wait for adsb_collect)
        if (int rc = park_pending(c)) return rc;
        if (int rc = resync_exact(c)) return rc;
    }
    ScanParams p{};
    p.src = d_src;
    p.n_samples = n_samples;
    p.n_chunks = n_chunks;
    p.clean_bitmap = nullptr;
    if (c->flush_pending) {  // icao_flush: retire the bitmap in use, continue on the next clean one
        p.clean_bitmap = c->d_bitmap[c->cur_bitmap];
        c->cur_bitmap = (c->cur_bitmap + 1) % kBitmaps;
    }
    p.bitmap = c->d_bitmap[c->cur_bitmap];
    p.hits = sl.d_hits;
    p.hits_cap = sl.hits_cap;
    p.ap = sl.d_ap;
    p.ap_cap = c->ap_cap;
    p.seg_cap = c->seg_cap;
    p.dap = c->fb.d_dap;  // only the reference-shaped kernel writes it (force_simple: fallback_slot() came first)
    p.dap_cap = c->fb.d_dap ? kWorstPerChunk : 0;
    p.tables = c->d_tables;
    p.ctr = sl.d_ctr;
    p.summary = sl.h_sum_dev;
    p.stagger_ticks = c->stagger_ticks;
    p.debug_stop = c->debug_stop;
    p.timeline = c->d_timeline;
    p.carry = c->carry_over && !from_mag ? sl.d_carry : nullptr;
    p.lead_from_src = lead_from_src ? 1u : 0u;
    p.order_cnt = order_on_device ? sl.d_order_cnt : nullptr;
    p.order_base = order_on_device ? sl.d_order_base : nullptr;
    p.order_tmp = order_on_device ? sl.d_order_tmp : nullptr;
    sl.device_scored = false;
    if (order_on_device && c->score.si) {
        if (c->exact_valid) {
            p.score = sl.score;
            p.score.exact_retired = nullptr;
            if (c->flush_pending) {  // icao_flush: this pass starts from the clean bitmap
                p.score.exact_retired = c->exact_bm[c->cur_exact];
                c->cur_exact ^= 1;
            }
            p.score.exact = c->exact_bm[c->cur_exact];
            p.score.out_msgs = sl.h_msgs_dev;
            p.score.out_adds = sl.h_adds_dev;
            p.score.summary = sl.h_ssum_dev;
            sl.device_scored = true;
            sl.score_epoch = c->score_epoch;
        }
    }

    sl.src = d_src;
    sl.from_mag = from_mag;
    sl.n_samples = n_samples;
    sl.n_chunks = n_chunks;
    sl.flush_before = c->flush_pending;
    sl.profiled = c->profiling;
    const int prof = sl.profiled;
    sl.seq = c->next_seq++;
    if (c->next_seq == 0) c->next_seq = 1;
    sl.scan_seq = ++c->scan_counter;
    sl.ev[0] = c->scan_ev[sl.scan_seq % kScanEvRing][0];
    sl.ev[1] = c->scan_ev[sl.scan_seq % kScanEvRing][1];
    sl.h_sum->seq = 0;  // the records kernel overwrites it, last, with sl.seq
    p.seq = sl.seq;
    if (sl.device_scored) {
        p.score.seq = sl.seq;
        sl.h_ssum->seq = 0;
    }
    // level 1: the scan launch stamps its own begin/end (no extra packets on the stream);
    // level 2: classic event records between all kernels
    static const bool ext_events = !tuning_env("ADSB_NO_EXT_EVENTS");
    const bool fast = !from_mag && !force_simple;
    p.ev_start = ext_events && prof == 1 && fast ? sl.ev[0] : nullptr;
    p.ev_stop = ext_events && prof == 1 && fast ? sl.ev[1] : nullptr;

    c->flush_pending = false;
    const bool classic = prof > 1 || (prof == 1 && (!fast || !ext_events));
    // odd slots scan on the second stream, unless something orders consecutive passes (the
    // carry hand-off) or the pass is a one-off (fallback, caller-supplied magnitudes)
    static const bool one_scan_stream = tuning_env("ADSB_ONE_SCAN_STREAM") != nullptr;
    const bool second = fast && !p.carry && advance_carry && !one_scan_stream && (c->submitted & 1u) != 0;
    hipStream_t ss = c->scan_stream[second ? 1 : 0];
    // the input is complete at `input_done` (the ring's copy) or where `stream` stands now
    hipEvent_t ready = input_done;
    if (!ready) {
        ready = c->input_ready[second ? 1 : 0];
        HIP_TRY(c, hipEventRecord(ready, c->stream));
    }
    HIP_TRY(c, hipStreamWaitEvent(ss, ready, 0));
    if (classic) HIP_TRY(c, hipEventRecord(sl.ev[0], ss));
    if (p.carry && advance_carry)  // this pass's lead-in: where the previous submission ended
        HIP_TRY(c, hipMemcpyAsync(sl.d_carry, c->d_carry_next, kCarrySamples * sizeof(uint32_t),
                                  hipMemcpyDeviceToDevice, ss));
    if (int e = force_simple ? launch_scan_simple(p, from_mag, ss) : launch_scan(p, from_mag, ss))
        return fail(c, (hipError_t)e, "launch_scan");
    if (classic) HIP_TRY(c, hipEventRecord(sl.ev[1], ss));
    if (p.carry && advance_carry) {
        // the next submission starts from the end of this one's input (taken now: the caller
        // may reuse the buffer as soon as this pass is collected)
        if (int e = launch_update_carry(sl.d_carry, d_src, n_samples, c->d_carry_next, ss))
            return fail(c, (hipError_t)e, "launch_update_carry");
    }
    // the tail runs on its own stream behind the scan: the next pass's scan does not wait
    // for it (it works on the other slot's lists and counters)
    // (a blocking call has nothing to overlap with: its tail stays on the scan stream and
    // saves the cross-stream hand-off)
    // A small pass is all launch overhead: its tail stays on its scan stream too (the two scan
    // streams still let consecutive passes overlap), which saves the cross-stream hand-off.
    if (n_chunks <= kInlineTailChunks) inline_tail = true;
    hipStream_t ts = inline_tail ? ss : c->tail_stream;
    if (p.clean_bitmap && fast) {
        // this pass's records kernel clears the bitmap the previous passes matched against: not
        // before the pass still in flight (whichever stream its tail is on) is through with it
        // (a pass whose tail ran on this same in-order stream is already behind us: only tails that
        // ran elsewhere -- small passes keep theirs on their scan stream -- need the event; it is the
        // one behind their records kernel, not `done`, which device-scored passes record later, on
        // the score stream)
        for (Slot &other : c->slot)
            if (&other != &sl && other.busy && other.tail_q != ts) HIP_TRY(c, hipStreamWaitEvent(ts, other.recorded, 0));
    }
    // The match must see every bit the scans of this and of all earlier passes set in the bitmap
    // (addresses their clean DF11 / DF17 frames will add).  Behind its own scan it is in stream order or
    // waits for `scanned`.  Behind the previous pass's scan it is in order when both matches run on the
    // tail stream (that pass's match waited for its scan); but a small pass matches on its own scan
    // stream, beside the other one -- where the previous pass may still be scanning, or, the other way
    // round, where a small previous pass may not even have started (its input still being copied) when
    // this one's scan is over.  Then the match waits for the previous scan explicitly.  (Scans before the
    // previous one are behind this pass's scan or the previous pass's, on the same two streams.)
    HIP_TRY(c, hipEventRecord(sl.scanned, ss));
    if (!inline_tail) HIP_TRY(c, hipStreamWaitEvent(ts, sl.scanned, 0));
    if (c->prev_scanned && c->prev_scan_stream != ss && (inline_tail || c->prev_inline))
        HIP_TRY(c, hipStreamWaitEvent(ts, c->prev_scanned, 0));
    c->prev_scanned = sl.scanned;
    c->prev_scan_stream = ss;
    c->prev_inline = inline_tail;
    if (prof > 1) HIP_TRY(c, hipEventRecord(sl.ev[2], ts));
    static const bool skip_match = tuning_env("ADSB_SKIP_MATCH") != nullptr;  // measurement aid (tuning build only): wrong results
    if (!skip_match)
        if (int e = launch_match(p, ts)) return fail(c, (hipError_t)e, "launch_match");
    if (int e = launch_order_hits(p, ts)) return fail(c, (hipError_t)e, "launch_order_hits");
    if (prof > 1) HIP_TRY(c, hipEventRecord(sl.ev[3], ts));
    // the records kernel writes the records and the summary into the slot's mapped host
    // memory with write-through stores; `done` only has to say the kernel has drained
    if (int e = launch_records(p, from_mag, sl.h_rec_dev, ts))
        return fail(c, (hipError_t)e, "launch_records");
    sl.tail_q = ts;
    // (always: a later pass whose own tail runs on another stream -- a small one behind an icao_flush --
    // waits for this event before its records kernel clears the bitmap this pass matched against)
    HIP_TRY(c, hipEventRecord(sl.recorded, ts));
    if (sl.device_scored) {
        // Scoring runs on its own in-order stream behind this pass's records kernel, so that the next
        // pass's match / order / records (tail stream) overlap it: every kernel beside the persistent
        // scan is latency, and one chain of eight would be longer than the scan it hides behind.
        // The score stream's order is the filter's order: k_score(i+1) reads the exact bitmap after
        // k_emit(i) has committed pass i's additions to it.
        hipStream_t qs = c->score_stream;
        HIP_TRY(c, hipStreamWaitEvent(qs, sl.recorded, 0));
        if (int e = launch_score(p, qs)) return fail(c, (hipError_t)e, "launch_score");
        ts = qs;
    }
    if (prof > 1) HIP_TRY(c, hipEventRecord(sl.ev[4], ts));
    HIP_TRY(c, hipEventRecord(sl.done, ts));
    return ADSB_OK;
}

// The records and the summary travel to host memory as separate posted writes; the summary's
// sequence word says the pass is done, this says every one of its records has landed whole: the
// 64-bit sum of all their u64 words, as the records kernel added them up.  (Records that are still
// in flight when the completion event has fired would be a platform fault: give them a moment,
// then fail loudly rather than replay something torn.)
int verify_records(adsb_ctx *c, const Summary *sum, const TrialRecord *rec, size_t n)
{
    const uint64_t want = (uint64_t)sum->rec_sum_hi << 32 | sum->rec_sum_lo;
    for (int attempt = 0; attempt < 200; attempt++) {
        uint64_t got = 0;
        const uint64_t *w = reinterpret_cast<const uint64_t *>(rec);
        for (size_t i = 0; i < 4 * n; i++) got += __atomic_load_n(&w[i], __ATOMIC_RELAXED);
        if (got == want) return ADSB_OK;
        for (volatile int spin = 0; spin < 2000; spin++) {}
    }
    if (c) c->last_error = "trial records in host memory do not add up to the checksum of the pass that wrote them";
    return ADSB_ERR_HIP;
}

// The exact bitmap rebuilt from the host's filter table (only while nothing is in flight).
int resync_exact(adsb_ctx *c)
{
    std::vector<uint32_t> addrs;
    for (uint32_t a : c->filter.table())
        if (a != 0 && a <= 0xFFFFFFu) addrs.push_back(a);
    hipStream_t ts = c->score_stream;
    for (uint32_t *bm : c->exact_bm) HIP_TRY(c, hipMemsetAsync(bm, 0, kBitmapAllocWords * sizeof(uint32_t), ts));
    if (!addrs.empty()) {
        if (addrs.size() > c->addrs_cap) {
            if (c->d_addrs) (void)hipFree(c->d_addrs);
            c->d_addrs = nullptr;
            c->addrs_cap = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_addrs, IcaoFilter::kSize * sizeof(uint32_t)));
            c->addrs_cap = IcaoFilter::kSize;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_addrs, addrs.data(), addrs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ts));
        if (int e = launch_set_addresses(c->d_addrs, (uint32_t)addrs.size(), c->exact_bm[c->cur_exact], ts))
            return fail(c, (hipError_t)e, "launch_set_addresses");
    }
    HIP_TRY(c, hipStreamSynchronize(ts));  // (rare: only after the host scored a pass itself)
    c->exact_valid = true;
    return ADSB_OK;
}

// The pass's messages as the device scored them, when they can be taken as they are: scored in the
// current epoch, whole (checksum), and with the filter nowhere near full -- the one situation whose
// reference behaviour (icao_filter_add gives up on a full table, src/icao_filter.rs:46-62) the parallel
// formulation does not reproduce.  Applies the pass's additions to the host's filter, in order.
bool take_device_result(adsb_ctx *c, Slot &sl, uint64_t chunk_offset, std::vector<adsb_msg> &out)
{
    if (!sl.device_scored || sl.score_epoch != c->score_epoch) return false;
    const ScoreSummary *ss = sl.h_ssum;
    if (__atomic_load_n(&ss->seq, __ATOMIC_ACQUIRE) != sl.seq || !ss->scored) return false;
    const size_t nm = ss->n_msgs, na = ss->n_adds;
    if (nm > c->score.cap || na > c->score.cap) return false;
    size_t held = 0;
    for (uint32_t a : c->filter.table()) held += a != 0;
    if (held + na + 64 >= IcaoFilter::kSize) return false;
    const uint64_t want = (uint64_t)ss->msg_sum_hi << 32 | ss->msg_sum_lo;
    bool whole = false;
    for (int attempt = 0; attempt < 200 && !whole; attempt++) {
        uint64_t got = 0;
        const uint64_t *w = reinterpret_cast<const uint64_t *>(sl.h_msgs);
        for (size_t i = 0; i < 5 * nm; i++) got += __atomic_load_n(&w[i], __ATOMIC_RELAXED);
        whole = got == want;
    }
    if (!whole) return false;
    const size_t at = out.size();
    out.insert(out.end(), sl.h_msgs, sl.h_msgs + nm);
    if (chunk_offset)
        for (size_t i = at; i < out.size(); i++) out[i].chunk += chunk_offset;
    for (size_t i = 0; i < na; i++) c->filter.add(sl.h_adds[i]);
    return true;
}

// Wait for the pass in `sl` and replay it.  Returns 1 when a device list overflowed
// (caller re-runs in smaller pieces), 0 on success, < 0 on error.
int finish_pass(adsb_ctx *c, Slot &sl, uint64_t chunk_offset, adsb_stats &st, std::vector<adsb_msg> &out)
{
#ifdef ADSB_TUNING
    const auto tw0 = std::chrono::steady_clock::now();
#endif
    HIP_TRY(c, hipEventSynchronize(sl.done));
#ifdef ADSB_TUNING
    c->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
#endif
    if (__atomic_load_n(&sl.h_sum->seq, __ATOMIC_ACQUIRE) != sl.seq) {
        c->last_error = "pass completed without publishing its summary";
        return ADSB_ERR_HIP;
    }
    if (sl.h_sum->overflow) return 1;
    const size_t n = sl.h_sum->n_hits;
    // (k_records' own test: a pass that k_score took over has left its records in device memory)
    const bool rec_on_device = sl.device_scored && n <= c->score.cap;
    if (!rec_on_device)
        if (int rc = verify_records(c, sl.h_sum, sl.h_rec, n)) return rc;
    if (sl.profiled) {
        float ms = 0;
        HIP_TRY(c, hipEventElapsedTime(&ms, sl.ev[0], sl.ev[1]));
        st.ms_scan += ms;
        // The device time this launch adds: the part of it after the latest scan end seen so far (the
        // "frontier": normally the previous pass's; scans on the two scan streams can also finish out of
        // order, and one that ended before the frontier adds nothing -- the union of the launches'
        // intervals is what is being summed).  The frontier's events are intact for kScanEvRing - kSlots
        // passes back: the ring is that much longer than what can be in flight.
        float excl = ms;
        bool advance = true;
        if (c->last_stop && sl.scan_seq - c->last_scan_seq <= (uint64_t)(kScanEvRing - kSlots)) {
            float since = 0;
            if (hipEventElapsedTime(&since, c->last_stop, sl.ev[1]) == hipSuccess) {
                if (since <= 0) {
                    excl = 0;
                    advance = false;
                } else if (since < excl) {
                    excl = since;
                }
            }
        }
        st.ms_scan_exclusive += excl;
        if (advance) {
            c->last_stop = sl.ev[1];
            c->last_scan_seq = sl.scan_seq;
        }
        if (sl.profiled > 1) {  // per-kernel split of the tail (events cost a few us each)
            HIP_TRY(c, hipEventElapsedTime(&ms, sl.ev[2], sl.ev[3]));
            st.ms_match += ms;
            HIP_TRY(c, hipEventElapsedTime(&ms, sl.ev[3], sl.ev[4]));
            st.ms_records += ms;
            HIP_TRY(c, hipEventElapsedTime(&ms, sl.ev[0], sl.ev[4]));
            st.ms_total_device += ms;
        }
    }
    st.n_candidates += sl.h_sum->n_cand_total;
    st.n_ap_entries += sl.h_sum->n_ap_total;
    st.n_records += n;
    // Density is records per buffer (8 and more: device-side order + score; under 2: the host does
    // it), so that a context of 64 buffers decides like one of 512.  Passes too small to be ordered on
    // the device anyway (and the fallback's one-buffer passes) say nothing about the stream: a small
    // pass between large dense ones must not flip the mode, each flip drains the pipeline.
    if (sl.hits_cap == c->hits_cap && sl.n_chunks > kInlineTailChunks) {
        if (n >= 8u * (size_t)sl.n_chunks) c->dense_mode = true;
        else if (n < 2u * (size_t)sl.n_chunks) c->dense_mode = false;
    }
    if (take_device_result(c, sl, chunk_offset, out)) return 0;
    if (rec_on_device && n) {
        HIP_TRY(c, hipMemcpy(sl.h_rec, sl.score.rec, n * sizeof(TrialRecord), hipMemcpyDeviceToHost));
        if (int rc = verify_records(c, sl.h_sum, sl.h_rec, n)) return rc;
    }
    c->host_replays++;
    c->score_epoch++;        // passes in flight were scored on the device without what this replay adds
    c->exact_valid = false;
    static const bool skip_replay = tuning_env("ADSB_SKIP_REPLAY") != nullptr;  // measurement aid (tuning build only)
#ifdef ADSB_TUNING
    const auto tr0 = std::chrono::steady_clock::now();
#endif
    if (!skip_replay) replay(c->filter, c->crc, sl.h_rec, n, chunk_offset, out, &c->host_sorts);
#ifdef ADSB_TUNING
    c->t_replay += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
#endif
    return 0;
}

// The overflow fallback re-runs a pass against the bitmap in use NOW.  Passes submitted after the
// overflowed one may have rotated the bitmaps (an icao_flush in between) and the retired one has been
// cleared, so the addresses the filter held before this pass would be missing from the superset:
// put them back.  At this point the host filter is exactly the state that preceded the pass (later
// passes have not been replayed yet), and extra bits only widen the superset for later passes.
int reseed_bitmap_from_filter(adsb_ctx *c)
{
    std::vector<uint32_t> addrs;
    for (uint32_t a : c->filter.table())
        if (a != 0 && a <= 0xFFFFFFu) addrs.push_back(a);  // DF18 entries (addr | 1 << 25) match no 24-bit residual
    if (addrs.empty()) return ADSB_OK;
    if (addrs.size() > c->addrs_cap) {
        if (c->d_addrs) (void)hipFree(c->d_addrs);
        c->d_addrs = nullptr;
        c->addrs_cap = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_addrs, IcaoFilter::kSize * sizeof(uint32_t)));
        c->addrs_cap = IcaoFilter::kSize;
    }
    HIP_TRY(c, hipMemcpy(c->d_addrs, addrs.data(), addrs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (int e = launch_set_addresses(c->d_addrs, (uint32_t)addrs.size(), c->d_bitmap[c->cur_bitmap], c->scan_stream[0]))
        return fail(c, (hipError_t)e, "launch_set_addresses");
    return ADSB_OK;
}

// Finish the oldest submission: replay it, or -- when a device list overflowed (far
// denser input than the lists were sized for) -- drain the stream and go chunk by
// chunk, where the worst case always fits.  Bitmap bits set by the aborted pass or by
// later passes are a harmless superset in time.
int collect_oldest(adsb_ctx *c, std::vector<adsb_msg> &out)
{
    Slot &sl = c->slot[c->collected % kSlots];
    adsb_stats st{};
    st.n_samples = sl.n_samples;
    st.n_chunks = sl.n_chunks;
    if (sl.flush_before) c->filter.flush();  // icao_flush() took effect before this pass
    int rc = finish_pass(c, sl, 0, st, out);
    if (rc > 0 && sl.from_mag) {  // a caller-supplied buffer denser than the fast scan's lists: again, the slow way
        st.retries++;
        for (hipStream_t q : c->scan_stream) HIP_TRY(c, hipStreamSynchronize(q));
        HIP_TRY(c, hipStreamSynchronize(c->tail_stream));
        HIP_TRY(c, hipStreamSynchronize(c->score_stream));
        const bool keep_flush = c->flush_pending;
        c->flush_pending = false;
        Slot tmp;
        rc = fallback_slot(c, sl, tmp);
        if (rc == 0) rc = reseed_bitmap_from_filter(c);
        if (rc == 0) rc = enqueue_pass(c, tmp, sl.src, true, sl.n_samples, 1, false, false, false, true);
        if (rc == 0) rc = finish_pass(c, tmp, 0, st, out);
        c->flush_pending = keep_flush;
    } else if (rc > 0) {
        st.retries++;
        for (hipStream_t q : c->scan_stream) HIP_TRY(c, hipStreamSynchronize(q));  // later passes have their results on the host
        HIP_TRY(c, hipStreamSynchronize(c->tail_stream));
        HIP_TRY(c, hipStreamSynchronize(c->score_stream));
        const bool keep_flush = c->flush_pending;
        c->flush_pending = false;
        Slot tmp;  // same counters, summary and events; one chunk at a time into the worst-case lists
        rc = fallback_slot(c, sl, tmp);
        if (rc == 0) rc = reseed_bitmap_from_filter(c);
        for (uint64_t ch = 0; ch < sl.n_chunks && rc == 0; ch++) {
            const uint64_t off = ch * kChunkSamples;
            const uint64_t n = std::min<uint64_t>(kChunkSamples, sl.n_samples - off);
            // (carry-over mode: buffers after the first find their lead-in in src itself; the
            // carry for the next call was already taken when the pass was first enqueued)
            // the reference-shaped kernel: its lists hold the worst case of a chunk
            rc = enqueue_pass(c, tmp, (const uint32_t *)sl.src + off, false, n, 1, false, ch > 0, false, true);
            if (rc == 0) rc = finish_pass(c, tmp, ch, st, out);
        }
        c->flush_pending = keep_flush;
    }
    sl.busy = false;
    c->collected++;
    if (rc > 0) {
        c->last_error = "device lists overflowed on a single chunk";
        return ADSB_ERR_HIP;
    }
    if (rc < 0) return rc;
    c->stats = st;
    return ADSB_OK;
}

// Finish every pass in flight now; the caller still gets them from adsb_collect, in order.
int park_pending(adsb_ctx *c)
{
    while (c->collected < c->submitted) {
        Slot &sl = c->slot[c->collected % kSlots];
        sl.parked_msgs.clear();
        sl.park_rc = collect_oldest(c, sl.parked_msgs);
        sl.parked_stats = c->stats;
        sl.parked = true;
    }
    return ADSB_OK;
}

// adsb_collect: the oldest pass the caller has not had yet
int collect_next(adsb_ctx *c, std::vector<adsb_msg> &out)
{
    if (c->delivered < c->collected) {
        Slot &sl = c->slot[c->delivered % kSlots];
        out.swap(sl.parked_msgs);
        sl.parked_msgs.clear();
        c->stats = sl.parked_stats;
        sl.parked = false;
        c->delivered++;
        return sl.park_rc;
    }
    const int rc = collect_oldest(c, out);
    c->delivered++;
    return rc;
}

int submit(adsb_ctx *c, const void *d_src, bool from_mag, uint64_t n_samples, bool inline_tail = false,
           hipEvent_t input_done = nullptr)
{
    const uint64_t n_chunks = from_mag ? 1 : (n_samples + kChunkSamples - 1) / kChunkSamples;
    if (n_chunks == 0 || n_chunks > kMaxChunks || n_chunks > c->max_chunks) return ADSB_ERR_INVALID;
    Slot &sl = c->slot[c->submitted % kSlots];
    if (sl.busy || sl.parked || c->shard_active) return ADSB_ERR_BUSY;
#ifdef ADSB_TUNING
    const auto te0 = std::chrono::steady_clock::now();
#endif
    int rc = enqueue_pass(c, sl, d_src, from_mag, n_samples, (uint32_t)n_chunks, inline_tail, false, true, false,
                          input_done);
#ifdef ADSB_TUNING
    c->t_enqueue += std::chrono::duration<double>(std::chrono::steady_clock::now() - te0).count();
#endif
    if (rc) return rc;
    sl.busy = true;
    c->submitted++;
    return ADSB_OK;
}

// synchronous pass: everything pending is finished first, in order
int run_sync(adsb_ctx *c, const void *d_src, bool from_mag, uint64_t n_samples, std::vector<adsb_msg> &out)
{
    if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
    int rc = submit(c, d_src, from_mag, n_samples, true);
    if (rc) return rc;
    return collect_next(c, out);
}

// IQ stream of any length resident on the device.
int demod_device(adsb_ctx *c, const void *d_iq, uint64_t n_samples, std::vector<adsb_msg> &out)
{
    if (n_samples == 0) {
        c->stats = adsb_stats{};
        return ADSB_OK;
    }
    adsb_stats total{};
    // a device pass takes at most max_chunks buffers (what the context's lists were sized for;
    // never more than kMaxChunks: entry packing): longer streams go in pieces, which is what
    // consecutive calls would be -- buffers are independent but for the filter
    const uint64_t piece = std::min<uint64_t>(kMaxChunks, c->max_chunks) * (uint64_t)kChunkSamples;
    for (uint64_t off = 0; off < n_samples; off += piece) {
        const uint64_t n = std::min<uint64_t>(piece, n_samples - off);
        std::vector<adsb_msg> part;
        int rc = run_sync(c, (const uint32_t *)d_iq + off, false, n, part);
        if (rc) return rc;
        const uint64_t chunk0 = off / kChunkSamples;
        for (auto &m : part) {
            m.chunk += chunk0;
            out.push_back(m);
        }
        total.n_chunks += c->stats.n_chunks;
        total.n_candidates += c->stats.n_candidates;
        total.n_ap_entries += c->stats.n_ap_entries;
        total.n_records += c->stats.n_records;
        total.ms_scan += c->stats.ms_scan;
        total.ms_scan_exclusive += c->stats.ms_scan_exclusive;
        total.ms_match += c->stats.ms_match;
        total.ms_records += c->stats.ms_records;
        total.ms_total_device += c->stats.ms_total_device;
        total.retries += c->stats.retries;
    }
    total.n_samples = n_samples;
    c->stats = total;
    return ADSB_OK;
}

int deliver(adsb_ctx *c, std::vector<adsb_msg> &msgs, adsb_msg *out, size_t cap,
            size_t *n_out)
{
    c->stats.n_messages = msgs.size();
    const size_t n = std::min(cap, msgs.size());
    if (n && out) std::memcpy(out, msgs.data(), n * sizeof(adsb_msg));
    if (n_out) *n_out = msgs.size();
    c->has_undelivered = msgs.size() > cap;
    if (!c->has_undelivered) {
        c->undelivered.clear();
        return ADSB_OK;
    }
    c->undelivered.swap(msgs);  // the pass is consumed: keep what it produced (adsb_fetch_messages)
    return ADSB_ERR_CAPACITY;
}

int ensure_stage(adsb_ctx *c, size_t bytes)
{
    if (bytes <= c->stage_bytes) return ADSB_OK;
    if (c->d_stage) HIP_TRY(c, hipFree(c->d_stage));
    c->d_stage = nullptr;
    c->stage_bytes = 0;
    HIP_TRY(c, hipMalloc(&c->d_stage, bytes));
    c->stage_bytes = bytes;
    return ADSB_OK;
}

}  // namespace

extern "C" {

// The internal streams of destroyed contexts, kept for the next context on the same device.  The
// runtime gives every new stream of a priority a new hardware queue until it has four of that priority
// and never gives one back, so a process that creates, destroys and re-creates contexts ends up with its
// streams spread over a different set of queues each time -- measured: a dense stream in the second
// context alternates 88 / 270 us per pass (0.175 ms mean) where the first one holds 0.13.  Re-using the
// same streams keeps every context of a process on the queues the first one got.
struct StreamSet {
    int device = -1;
    hipStream_t own = nullptr, scan[2] = {nullptr, nullptr}, tail = nullptr, score = nullptr, copy = nullptr;
};
std::mutex g_stream_pool_mu;
std::vector<StreamSet> g_stream_pool;

bool take_stream_set(int device, StreamSet &out)
{
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    for (size_t i = 0; i < g_stream_pool.size(); i++)
        if (g_stream_pool[i].device == device) {
            out = g_stream_pool[i];
            g_stream_pool.erase(g_stream_pool.begin() + (long)i);
            return true;
        }
    return false;
}

int adsb_create(adsb_ctx **out, int device, size_t max_chunks)
{
    if (!out) return ADSB_ERR_INVALID;
    *out = nullptr;
    if (device < 0) return ADSB_ERR_NO_DEVICE;  // no CPU backend by design
    if (max_chunks == 0) max_chunks = 1;
    if (max_chunks > kMaxChunks) return ADSB_ERR_INVALID;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device >= count) return ADSB_ERR_NO_DEVICE;

    adsb_ctx *c = new (std::nothrow) adsb_ctx;
    if (!c) return ADSB_ERR_NOMEM;
    c->device = device;
    c->max_chunks = max_chunks;
    if (const char *ds = tuning_env("ADSB_DEBUG_STOP")) c->debug_stop = std::atoi(ds);
    if (const char *st = tuning_env("ADSB_STAGGER")) c->stagger_ticks = (uint32_t)std::atoi(st);
    // The fast scan's AP list: one private segment per wave of every persistent workgroup (a pass
    // of n buffers runs min(17 n, resident grid) workgroups of four waves, so a small context only gets
    // the segments it can ever use), each sized for ~5x the rate pure noise produces (2.3 % of
    // samples become address/parity entries).  Denser input falls back to buffer-by-buffer
    // passes through the reference-shaped kernel, whose list (dap) and the hit list hold one
    // buffer's worst case: every position sliced, five trials each.
    (void)hipSetDevice(device);  // scan_resident_blocks() asks the current device
    const uint64_t used_segs = 4 * std::min<uint64_t>((uint64_t)scan_resident_blocks(), max_chunks * (uint64_t)fastgeo::kTilesPerChunk);
    c->seg_cap = (uint32_t)std::max<uint64_t>(1024, (max_chunks * (uint64_t)kChunkSamples / 8 + used_segs - 1) / used_segs);
    c->ap_cap = (uint32_t)(used_segs * c->seg_cap);
    // hit list: ~5x what a busy airspace produces (a frame leaves 3-4 trial records; 1000 frames/s
    // are ~55 per buffer); more than that is the fallback's business too
    c->hits_cap = (uint32_t)(4096 + max_chunks * 1024);

    int rc = ADSB_OK;
    auto body = [&]() -> int {
        HIP_TRY(c, hipSetDevice(device));
        StreamSet pooled;
        const bool reuse = !tuning_env("ADSB_STREAM_PRIO") && !tuning_env("ADSB_SCORE_PRIO") && take_stream_set(device, pooled);
        if (reuse) {
            c->own_stream = pooled.own;
            c->scan_stream[0] = pooled.scan[0];
            c->scan_stream[1] = pooled.scan[1];
            c->tail_stream = pooled.tail;
            c->score_stream = pooled.score;
            c->copy_stream_spare = pooled.copy;
        } else {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        }
        c->stream = c->own_stream;
        HIP_TRY(c, hipMalloc((void **)&c->d_mag, kMagDataLen * sizeof(uint16_t)));
        if (!reuse) {
            // The runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by
            // default, per priority) and two streams on one queue run strictly one after the other.
            // Both scan streams take the highest priority: that pool holds nothing else of this
            // process (the null stream, torch's and the caller's streams are of normal priority), so
            // the two get a queue each and consecutive scans can overlap.  They must have the SAME
            // priority: with different ones, whenever two scans are pending at once (after any hiccup
            // of the host) the higher one's starts first, its successor on that stream is then free
            // earlier too, and the stream settles into finishing passes in the order 2, 1, 4, 3, ...
            // for thousands of passes, 8-10 % slower (passes are collected in order), until another
            // hiccup flips it back; measured over 22 000 passes: (mid, high) spends a third of the
            // time in that mode, (high, high) and (mid, mid) none.
            int least = 0, greatest = 0;
            HIP_TRY(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
            int pt = least, p0 = greatest, p1 = greatest;
            if (const char *e = tuning_env("ADSB_STREAM_PRIO")) {  // measurement aid: "tail,scan0,scan1" as 0 (least) .. 2
                int a = 0, b = 1, d = 2;
                if (std::sscanf(e, "%d,%d,%d", &a, &b, &d) == 3) {
                    const int lv[3] = {least, (least + greatest) / 2, greatest};
                    pt = lv[a % 3], p0 = lv[b % 3], p1 = lv[d % 3];
                }
            }
            HIP_TRY(c, hipStreamCreateWithPriority(&c->tail_stream, hipStreamNonBlocking, pt));
            HIP_TRY(c, hipStreamCreateWithPriority(&c->scan_stream[0], hipStreamNonBlocking, p0));
            HIP_TRY(c, hipStreamCreateWithPriority(&c->scan_stream[1], hipStreamNonBlocking, p1));
            if (tuning_env("ADSB_TIMELINE")) std::fprintf(stderr, "stream priorities: least %d greatest %d\n", least, greatest);
        }
        for (auto &e : c->input_ready)
            HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
        for (auto &b : c->d_bitmap) HIP_TRY(c, hipMalloc((void **)&b, kBitmapAllocWords * sizeof(uint32_t)));
        for (Slot &sl : c->slot) {
            HIP_TRY(c, hipMalloc((void **)&sl.d_ctr, sizeof(Counters)));
            sl.hits_cap = c->hits_cap;
            HIP_TRY(c, hipMalloc((void **)&sl.d_hits, (size_t)c->hits_cap * sizeof(uint64_t)));
            HIP_TRY(c, hipMalloc((void **)&sl.d_ap, (size_t)c->ap_cap * sizeof(uint64_t)));
            HIP_TRY(c, hipMalloc((void **)&sl.d_order_cnt, (max_chunks + 1) * sizeof(uint32_t)));
            HIP_TRY(c, hipMemset(sl.d_order_cnt, 0, (max_chunks + 1) * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc((void **)&sl.d_order_base, (max_chunks + 1) * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc((void **)&sl.d_order_tmp, (size_t)c->hits_cap * sizeof(uint64_t)));
            HIP_TRY(c, hipEventCreateWithFlags(&sl.scanned, hipEventDisableTiming | hipEventDisableSystemFence));
            HIP_TRY(c, hipMalloc((void **)&sl.d_carry, kCarrySamples * sizeof(uint32_t)));
            HIP_TRY(c, hipMemset(sl.d_carry, 0, kCarrySamples * sizeof(uint32_t)));
        }
        HIP_TRY(c, hipMalloc((void **)&c->d_carry_next, kCarrySamples * sizeof(uint32_t)));
        HIP_TRY(c, hipMemset(c->d_carry_next, 0, kCarrySamples * sizeof(uint32_t)));
        {
            // device-side scoring state (shared by the passes: they go through it one after the other
            // on the tail stream); passes of more hits than `cap` are scored on the host
            ScoreDev &cd = c->score;   // cap / hash_mask / exact: the context's; the rest per slot
            cd.cap = std::min<uint32_t>(c->hits_cap, 131072u);
            uint32_t hsize = 1;
            while (hsize < 2 * cd.cap) hsize <<= 1;
            cd.hash_mask = hsize - 1;
            for (auto &bm : c->exact_bm) {
                HIP_TRY(c, hipMalloc((void **)&bm, kBitmapAllocWords * sizeof(uint32_t)));
                HIP_TRY(c, hipMemset(bm, 0, kBitmapAllocWords * sizeof(uint32_t)));
            }
            cd.exact = c->exact_bm[0];
            cd.si = reinterpret_cast<uint32_t *>(cd.exact);  // (non-null: "scoring is available")
            if (!c->score_stream) {
                int least = 0, greatest = 0;
                HIP_TRY(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
                int ps = least;  // with the tail stream's priority: its own hardware queue in that pool
                if (const char *e = tuning_env("ADSB_SCORE_PRIO")) ps = std::atoi(e) == 2 ? greatest : (std::atoi(e) == 1 ? (least + greatest) / 2 : least);
                HIP_TRY(c, hipStreamCreateWithPriority(&c->score_stream, hipStreamNonBlocking, ps));
            }
            for (Slot &sl : c->slot) {
                ScoreDev &sd = sl.score;
                sd = cd;
                HIP_TRY(c, hipMalloc((void **)&sd.si, (size_t)sd.cap * sizeof(uint32_t)));
                HIP_TRY(c, hipMalloc((void **)&sd.rec, (size_t)sd.cap * sizeof(TrialRecord)));
                HIP_TRY(c, hipMalloc((void **)&sd.flag, (size_t)sd.cap * sizeof(uint32_t)));
                HIP_TRY(c, hipMalloc((void **)&sd.pos, (size_t)sd.cap * sizeof(unsigned long long)));
                HIP_TRY(c, hipMalloc((void **)&sd.slot, (size_t)sd.cap * sizeof(uint32_t)));
                HIP_TRY(c, hipMalloc((void **)&sd.hash, (size_t)hsize * sizeof(unsigned long long)));
                HIP_TRY(c, hipMemset(sd.hash, 0xFF, (size_t)hsize * sizeof(unsigned long long)));
                HIP_TRY(c, hipMalloc((void **)&sd.blk, 2 * kScoreBlocks * sizeof(uint32_t)));
                HIP_TRY(c, hipMalloc((void **)&sd.state, sizeof(ScoreState)));
                HIP_TRY(c, hipMemset(sd.state, 0, sizeof(ScoreState)));
                HIP_TRY(c, hipEventCreateWithFlags(&sl.recorded, hipEventDisableTiming | hipEventDisableSystemFence));
            }
            const ScoreDev &sd = cd;
            for (Slot &sl : c->slot) {
                HIP_TRY(c, hipHostMalloc((void **)&sl.h_msgs, (size_t)sd.cap * sizeof(adsb_msg), hipHostMallocMapped | hipHostMallocCoherent));
                HIP_TRY(c, hipHostMalloc((void **)&sl.h_adds, (size_t)sd.cap * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
                HIP_TRY(c, hipHostMalloc((void **)&sl.h_ssum, sizeof(ScoreSummary), hipHostMallocMapped | hipHostMallocCoherent));
                HIP_TRY(c, hipHostGetDevicePointer((void **)&sl.h_msgs_dev, sl.h_msgs, 0));
                HIP_TRY(c, hipHostGetDevicePointer((void **)&sl.h_adds_dev, sl.h_adds, 0));
                HIP_TRY(c, hipHostGetDevicePointer((void **)&sl.h_ssum_dev, sl.h_ssum, 0));
                std::memset(sl.h_ssum, 0, sizeof(ScoreSummary));
            }
        }
        HIP_TRY(c, hipMalloc((void **)&c->d_tables, kTabWords * sizeof(uint32_t)));
        {
            std::vector<uint32_t> tab = build_gf_tables();
            const std::vector<uint32_t> r16 = build_r16(), ft = build_field_table(fast_plane_bytes()),
                                        bits = build_bit_residuals();
            tab.insert(tab.end(), r16.begin(), r16.end());
            tab.insert(tab.end(), ft.begin(), ft.end());
            tab.insert(tab.end(), bits.begin(), bits.end());
            HIP_TRY(c, hipMemcpy(c->d_tables, tab.data(), tab.size() * sizeof(uint32_t),
                                 hipMemcpyHostToDevice));
        }
        for (Slot &sl : c->slot) {
            // (mapped + coherent: the records kernel's write-through stores are visible to the host
            // when its completion event fires, whatever the runtime's default for pinned memory)
            HIP_TRY(c, hipHostMalloc((void **)&sl.h_sum, sizeof(Summary), hipHostMallocMapped | hipHostMallocCoherent));
            HIP_TRY(c, hipHostMalloc((void **)&sl.h_rec, (size_t)c->hits_cap * sizeof(TrialRecord),
                                     hipHostMallocMapped | hipHostMallocCoherent));
            HIP_TRY(c, hipHostGetDevicePointer((void **)&sl.h_sum_dev, sl.h_sum, 0));
            HIP_TRY(c, hipHostGetDevicePointer((void **)&sl.h_rec_dev, sl.h_rec, 0));
            // timing-only events: no system-scope fence when they complete (~10 us each otherwise)
            for (int k = 2; k < 5; k++) HIP_TRY(c, hipEventCreateWithFlags(&sl.ev[k], hipEventDisableSystemFence));
            const bool fenced = tuning_env("ADSB_DONE_FENCE") != nullptr;  // measurement aid only
            HIP_TRY(c, hipEventCreateWithFlags(&sl.done, fenced ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence)));
        }
        for (auto &pair : c->scan_ev)
            for (auto &e : pair) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
        if (tuning_env("ADSB_TIMELINE")) {
            // 1: stamps of 8 blocks x 8 tiles; 2 (with ADSB_DEBUG_STOP=100): per-wave phase totals
            HIP_TRY(c, hipMalloc((void **)&c->d_timeline, kTimelineWords * sizeof(unsigned long long)));
            HIP_TRY(c, hipMemset(c->d_timeline, 0, kTimelineWords * sizeof(unsigned long long)));
        }
        // both bitmaps clean and both counter blocks zero to start with; from then on each
        // pass cleans up for the next (the first pass needs no flush of its own)
        for (int k = 0; k < kBitmaps; k++)
            if (int e = launch_reset(c->slot[k % kSlots].d_ctr, c->d_bitmap[k], c->stream))
                return fail(c, (hipError_t)e, "launch_reset");
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->flush_pending = false;
        return (int)ADSB_OK;
    };
    rc = body();
    if (rc != ADSB_OK) {
        std::fprintf(stderr, "adsb_create: %s\n", c->last_error.c_str());
        adsb_destroy(c);
        return rc;
    }
    *out = c;
    return ADSB_OK;
}

void adsb_destroy(adsb_ctx *c)
{
    if (!c) return;
#ifdef ADSB_TUNING
    if (tuning_env("ADSB_HOST_TIMES"))
        std::fprintf(stderr, "host times over %llu passes: enqueue %.1f us, wait %.1f us, replay %.1f us per pass\n",
                     (unsigned long long)c->collected, 1e6 * c->t_enqueue / (c->collected ? c->collected : 1),
                     1e6 * c->t_wait / (c->collected ? c->collected : 1), 1e6 * c->t_replay / (c->collected ? c->collected : 1));
#endif
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (auto &pair : c->scan_ev)
        for (auto &e : pair)
            if (e) (void)hipEventDestroy(e);
    for (Slot &sl : c->slot) {
        for (int k = 2; k < 5; k++)
            if (sl.ev[k]) (void)hipEventDestroy(sl.ev[k]);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.scanned) (void)hipEventDestroy(sl.scanned);
        if (sl.d_ctr) (void)hipFree(sl.d_ctr);
        if (sl.d_hits) (void)hipFree(sl.d_hits);
        if (sl.d_ap) (void)hipFree(sl.d_ap);
        if (sl.d_order_cnt) (void)hipFree(sl.d_order_cnt);
        if (sl.d_order_base) (void)hipFree(sl.d_order_base);
        if (sl.d_order_tmp) (void)hipFree(sl.d_order_tmp);
        if (sl.h_msgs) (void)hipHostFree(sl.h_msgs);
        if (sl.h_adds) (void)hipHostFree(sl.h_adds);
        if (sl.h_ssum) (void)hipHostFree(sl.h_ssum);
        if (sl.d_carry) (void)hipFree(sl.d_carry);
        if (sl.h_sum) (void)hipHostFree(sl.h_sum);
        if (sl.h_rec) (void)hipHostFree(sl.h_rec);
    }
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_mag) (void)hipFree(c->d_mag);
    for (auto &b : c->d_bitmap)
        if (b) (void)hipFree(b);
    for (hipStream_t q : c->scan_stream)
        if (q) (void)hipStreamSynchronize(q);
    for (hipEvent_t e : c->input_ready)
        if (e) (void)hipEventDestroy(e);
    if (c->tail_stream) (void)hipStreamSynchronize(c->tail_stream);
    for (Slot &sl : c->slot) {
        for (void *q : {(void *)sl.score.si, (void *)sl.score.rec, (void *)sl.score.flag, (void *)sl.score.slot, (void *)sl.score.pos,
                        (void *)sl.score.hash, (void *)sl.score.blk, (void *)sl.score.state})
            if (q && q != (void *)c->score.exact) (void)hipFree(q);
        if (sl.recorded) (void)hipEventDestroy(sl.recorded);
    }
    for (uint32_t *bm : c->exact_bm)
        if (bm) (void)hipFree(bm);
    if (c->score_stream) (void)hipStreamSynchronize(c->score_stream);
    if (c->fb.d_hits) (void)hipFree(c->fb.d_hits);
    if (c->fb.d_dap) (void)hipFree(c->fb.d_dap);
    if (c->fb.h_rec) (void)hipHostFree(c->fb.h_rec);
    if (c->d_tables) (void)hipFree(c->d_tables);
    for (auto &r : c->ring) {
        if (r.copied) (void)hipEventDestroy(r.copied);
        if (r.h_iq) (void)hipHostFree(r.h_iq);
        if (r.d_iq) (void)hipFree(r.d_iq);
    }
    if (c->d_addrs) (void)hipFree(c->d_addrs);
    if (c->d_carry_next) (void)hipFree(c->d_carry_next);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->d_timeline && c->debug_stop == 100) {
        // profiling aid: phase / barrier-wait totals of the last scan, summed over all waves
        std::vector<unsigned long long> tl(kTimelineWords);
        if (hipMemcpy(tl.data(), c->d_timeline, kTimelineWords * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            double sum[8] = {0};
            int nw = 0;
            for (size_t w = 0; w < kTimelineWords / 8; w++) {
                double tot = 0;
                for (int k = 0; k < 8; k++) tot += (double)tl[w * 8 + k];
                if (tot == 0) continue;
                nw++;
                for (int k = 0; k < 8; k++) sum[k] += (double)tl[w * 8 + k];
            }
            double all = 0;
            for (double v : sum) all += v;
            static const char *name[8] = {"P1", "wait B1", "P2", "wait B2", "P3-5", "wait B3", "epilogue", "wait B4"};
            std::fprintf(stderr, "phase accounting over %d waves (clock64 ticks per wave, share):\n", nw);
            for (int k = 0; k < 8; k++)
                std::fprintf(stderr, "  %-9s %10.0f  %5.1f %%\n", name[k], sum[k] / (nw ? nw : 1), 100.0 * sum[k] / (all ? all : 1));
        }
        (void)hipFree(c->d_timeline);
    } else if (c->d_timeline) {
        // profiling aid: dump the stamps of the last scan on the way out
        unsigned long long tl[512];
        if (hipMemcpy(tl, c->d_timeline, sizeof(tl), hipMemcpyDeviceToHost) == hipSuccess)
            for (int b = 0; b < 8; b++)
                for (int it = 0; it < 8; it++) {
                    const unsigned long long *r = tl + (b * 8 + it) * 8;
                    if (!r[0]) continue;
                    std::fprintf(stderr, "timeline block %d tile %d: start %8lld |", b * 128, it,
                                 (long long)(r[0] - tl[0]));
                    for (int k = 1; k < 7; k++) std::fprintf(stderr, " %6lld", (long long)(r[k] - r[k - 1]));
                    std::fprintf(stderr, "  total %lld\n", (long long)(r[6] - r[0]));
                }
        (void)hipFree(c->d_timeline);
    }
    {
        // the streams go back to the pool (a context whose creation failed half-way has no full set:
        // its streams are simply destroyed)
        StreamSet set;
        set.device = c->device;
        set.own = c->own_stream;
        set.scan[0] = c->scan_stream[0];
        set.scan[1] = c->scan_stream[1];
        set.tail = c->tail_stream;
        set.score = c->score_stream;
        set.copy = c->copy_stream ? c->copy_stream : c->copy_stream_spare;
        if (set.own && set.scan[0] && set.scan[1] && set.tail && set.score) {
            std::lock_guard<std::mutex> lk(g_stream_pool_mu);
            g_stream_pool.push_back(set);
        } else {
            for (hipStream_t q : {set.own, set.scan[0], set.scan[1], set.tail, set.score, set.copy})
                if (q) (void)hipStreamDestroy(q);
        }
    }
    delete c;
}

int adsb_set_stream(adsb_ctx *c, void *hip_stream)
{
    if (!c) return ADSB_ERR_INVALID;
    if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return ADSB_OK;
}

int adsb_set_profiling(adsb_ctx *c, int enabled)
{
    if (!c) return ADSB_ERR_INVALID;
    c->profiling = enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled);
    return ADSB_OK;
}

int adsb_set_carry_over(adsb_ctx *c, int enabled)
{
    if (!c) return ADSB_ERR_INVALID;
    if (c->submitted != c->delivered || c->shard_active) return ADSB_ERR_BUSY;
    HIP_TRY(c, hipSetDevice(c->device));
    c->carry_over = enabled != 0;
    // the stream starts here: nothing precedes the next call
    for (Slot &sl : c->slot) HIP_TRY(c, hipMemsetAsync(sl.d_carry, 0, kCarrySamples * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_carry_next, 0, kCarrySamples * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ADSB_OK;
}

int adsb_icao_flush(adsb_ctx *c)
{
    if (!c) return ADSB_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    // Takes effect for everything submitted after this call: the next pass's reset kernel
    // clears the device bitmap (stream-ordered), and the host filter is flushed when that
    // pass is collected, after the passes before it have been replayed.
    c->flush_pending = true;
    return ADSB_OK;
}

int adsb_to_mag(adsb_ctx *c, const int16_t *iq, size_t n, uint16_t *data_out, size_t *length_out)
{
    if (!c || (!iq && n) || !data_out) return ADSB_ERR_INVALID;
    if (n > kChunkSamples) return ADSB_ERR_TOO_LONG;  // reference: index panic, lib.rs:48
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_stage(c, (size_t)kChunkSamples * 4);
    if (rc) return rc;
    if (n) HIP_TRY(c, hipMemcpyAsync(c->d_stage, iq, n * 4, hipMemcpyHostToDevice, c->stream));
    if (int e = launch_to_mag(c->d_stage, (uint32_t)n, c->d_mag, c->stream))
        return fail(c, (hipError_t)e, "launch_to_mag");
    HIP_TRY(c, hipMemcpyAsync(data_out, c->d_mag, kMagDataLen * sizeof(uint16_t),
                              hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (length_out) *length_out = n;
    return ADSB_OK;
}

int adsb_demodulate2400(adsb_ctx *c, const uint16_t *data, size_t length, adsb_msg *out, size_t cap,
                        size_t *n_out)
{
    if (!c || !data || (!out && cap)) return ADSB_ERR_INVALID;
    if (length > kChunkSamples) return ADSB_ERR_TOO_LONG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
    c->stats = adsb_stats{};
    c->stats.n_samples = length;
    c->stats.n_chunks = 1;
    std::vector<adsb_msg> msgs;
    if (length) {
        HIP_TRY(c, hipMemcpyAsync(c->d_mag, data, kMagDataLen * sizeof(uint16_t),
                                  hipMemcpyHostToDevice, c->stream));
        int rc = run_sync(c, c->d_mag, true, length, msgs);
        if (rc) return rc;
    }
    return deliver(c, msgs, out, cap, n_out);
}

int adsb_demod_iq_device(adsb_ctx *c, const void *d_iq, size_t n_samples, adsb_msg *out, size_t cap,
                         size_t *n_out)
{
    if (!c || (!d_iq && n_samples) || (!out && cap)) return ADSB_ERR_INVALID;
    if (((uintptr_t)d_iq & 15u) != 0) return ADSB_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<adsb_msg> msgs;
    int rc = demod_device(c, d_iq, n_samples, msgs);
    if (rc) return rc;
    return deliver(c, msgs, out, cap, n_out);
}

int adsb_submit_iq_device(adsb_ctx *c, const void *d_iq, size_t n_samples)
{
    if (!c || !d_iq || n_samples == 0) return ADSB_ERR_INVALID;
    if (((uintptr_t)d_iq & 15u) != 0) return ADSB_ERR_INVALID;
    if ((n_samples + kChunkSamples - 1) / kChunkSamples > std::min<uint64_t>(kMaxChunks, c->max_chunks))
        return ADSB_ERR_INVALID;  // more buffers than the context was created for
    HIP_TRY(c, hipSetDevice(c->device));
    return submit(c, d_iq, false, n_samples);
}

int adsb_collect(adsb_ctx *c, adsb_msg *out, size_t cap, size_t *n_out)
{
    if (!c || (!out && cap)) return ADSB_ERR_INVALID;
    if (c->submitted == c->delivered) return ADSB_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<adsb_msg> msgs;
    int rc = collect_next(c, msgs);
    if (rc) return rc;
    return deliver(c, msgs, out, cap, n_out);
}

int adsb_pending(const adsb_ctx *c) { return c ? (int)(c->submitted - c->delivered) : 0; }

int adsb_fetch_messages(adsb_ctx *c, adsb_msg *out, size_t cap, size_t *n_out)
{
    if (!c || (!out && cap) || !c->has_undelivered) return ADSB_ERR_INVALID;
    const size_t n = std::min(cap, c->undelivered.size());
    if (n) std::memcpy(out, c->undelivered.data(), n * sizeof(adsb_msg));
    if (n_out) *n_out = c->undelivered.size();
    return c->undelivered.size() > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
}

int adsb_ring_create(adsb_ctx *c, size_t samples_per_slot)
{
    if (!c || samples_per_slot == 0 || c->ring_samples) return ADSB_ERR_INVALID;
    if ((samples_per_slot + kChunkSamples - 1) / kChunkSamples > c->max_chunks) return ADSB_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->copy_stream_spare) {
        c->copy_stream = c->copy_stream_spare;
        c->copy_stream_spare = nullptr;
    } else {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    }
    for (auto &r : c->ring) {
        HIP_TRY(c, hipHostMalloc((void **)&r.h_iq, samples_per_slot * 4, hipHostMallocDefault));
        HIP_TRY(c, hipMalloc(&r.d_iq, samples_per_slot * 4));
        HIP_TRY(c, hipEventCreateWithFlags(&r.copied, hipEventDisableTiming));
    }
    c->ring_samples = samples_per_slot;
    return ADSB_OK;
}

int adsb_ring_acquire(adsb_ctx *c, int16_t **host_iq, size_t *capacity_samples)
{
    if (!c || !host_iq || !c->ring_samples) return ADSB_ERR_INVALID;
    if (c->slot[c->submitted % kSlots].busy || c->slot[c->submitted % kSlots].parked) return ADSB_ERR_BUSY;  // collect the oldest pass first
    *host_iq = c->ring[c->submitted % kSlots].h_iq;
    if (capacity_samples) *capacity_samples = c->ring_samples;
    return ADSB_OK;
}

int adsb_ring_submit(adsb_ctx *c, size_t n_samples)
{
    if (!c || !c->ring_samples || n_samples == 0 || n_samples > c->ring_samples) return ADSB_ERR_INVALID;
    if (c->slot[c->submitted % kSlots].busy || c->slot[c->submitted % kSlots].parked) return ADSB_ERR_BUSY;
    HIP_TRY(c, hipSetDevice(c->device));
    auto &r = c->ring[c->submitted % kSlots];
    // H2D on the copy stream; the pass on the compute stream waits for it, so this slot's
    // transfer overlaps the other slot's kernels
    HIP_TRY(c, hipMemcpyAsync(r.d_iq, r.h_iq, n_samples * 4, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipEventRecord(r.copied, c->copy_stream));
    return submit(c, r.d_iq, false, n_samples, false, r.copied);
}

int adsb_demod_iq(adsb_ctx *c, const int16_t *iq, size_t n_samples, adsb_msg *out, size_t cap,
                  size_t *n_out)
{
    if (!c || (!iq && n_samples) || (!out && cap)) return ADSB_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    // stage through the device in pieces of at most max_chunks chunks
    std::vector<adsb_msg> msgs;
    adsb_stats total{};
    const size_t piece = c->max_chunks * (size_t)kChunkSamples;
    int rc = ensure_stage(c, std::min(piece, std::max<size_t>(n_samples, 1)) * 4);
    if (rc) return rc;
    for (size_t off = 0; off < n_samples; off += piece) {
        const size_t n = std::min(piece, n_samples - off);
        HIP_TRY(c, hipMemcpyAsync(c->d_stage, iq + 2 * off, n * 4, hipMemcpyHostToDevice, c->stream));
        std::vector<adsb_msg> part;
        rc = demod_device(c, c->d_stage, n, part);
        if (rc) return rc;
        const uint64_t chunk0 = off / kChunkSamples;
        for (auto &m : part) {
            m.chunk += chunk0;
            msgs.push_back(m);
        }
        total.n_chunks += c->stats.n_chunks;
        total.n_candidates += c->stats.n_candidates;
        total.n_ap_entries += c->stats.n_ap_entries;
        total.n_records += c->stats.n_records;
        total.ms_scan += c->stats.ms_scan;
        total.ms_scan_exclusive += c->stats.ms_scan_exclusive;
        total.ms_match += c->stats.ms_match;
        total.ms_records += c->stats.ms_records;
        total.ms_total_device += c->stats.ms_total_device;
        total.retries += c->stats.retries;
    }
    total.n_samples = n_samples;
    c->stats = total;
    return deliver(c, msgs, out, cap, n_out);
}

int adsb_read_test_data(const char *path, int16_t *iq, size_t max_samples, size_t *n_out)
{
    if (!path || !iq) return ADSB_ERR_INVALID;
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return ADSB_ERR_INVALID;
    size_t k = 0;
    unsigned char b[4];
    while (k < max_samples && std::fread(b, 1, 4, fp) == 4) {
        // file: [im lo][im hi][re lo][re hi]  (src/utils.rs:29-31) -> memory {re, im}
        iq[2 * k] = (int16_t)(b[2] | (b[3] << 8));
        iq[2 * k + 1] = (int16_t)(b[0] | (b[1] << 8));
        k++;
    }
    std::fclose(fp);
    if (n_out) *n_out = k;
    return ADSB_OK;
}

int adsb_selftest_mag_digest(adsb_ctx *c, uint32_t first_bits, uint32_t count, uint64_t *sum_out,
                             uint64_t *xor_out)
{
    if (!c || !sum_out || !xor_out) return ADSB_ERR_INVALID;
    if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
    HIP_TRY(c, hipSetDevice(c->device));
    // the counters block doubles as the 16-byte result area
    static_assert(sizeof(Counters) >= 16, "digest result fits the counters block");
    // the counters block the next pass will use doubles as the 16-byte result area; it is
    // zeroed again afterwards
    Counters *scratch = c->slot[c->submitted % kSlots].d_ctr;
    HIP_TRY(c, hipMemsetAsync(scratch, 0, sizeof(Counters), c->stream));
    if (int e = launch_mag_digest(first_bits, count, (unsigned long long *)scratch, c->stream))
        return fail(c, (hipError_t)e, "launch_mag_digest");
    uint64_t res[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(res, scratch, sizeof(res), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemsetAsync(scratch, 0, sizeof(Counters), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *sum_out = res[0];
    *xor_out = res[1];
    return ADSB_OK;
}

namespace {

// One blocking pass of the self-test instantiation of the fast scan: the gate-stage position list
// (every pattern match that is a preamble by the reference's own tests, with its stage and the
// production gates' verdict) and the address/parity trial list.  The context's filter is not touched.
int selftest_pass(adsb_ctx *c, const void *d_iq, size_t n_samples, std::vector<uint64_t> *pre, std::vector<uint64_t> *snr,
                  std::vector<uint64_t> *cands, std::vector<uint64_t> *aps)
{
    if (!c || !d_iq || n_samples == 0) return ADSB_ERR_INVALID;
    if (c->submitted != c->delivered || c->shard_active) return ADSB_ERR_BUSY;
    if ((uintptr_t)d_iq % 16) return ADSB_ERR_INVALID;
    const uint64_t n_chunks = (n_samples + kChunkSamples - 1) / kChunkSamples;
    if (n_chunks > c->max_chunks || n_chunks > kMaxChunks) return ADSB_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    Slot &sl = c->slot[0];
    const uint32_t dev_cap = (uint32_t)std::min<uint64_t>(n_samples, 1u << 26);  // a list entry per position at most
    uint64_t *d_cand = nullptr;
    uint32_t *d_count = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_cand, (size_t)dev_cap * sizeof(uint64_t)));
    if (hipMalloc((void **)&d_count, sizeof(uint32_t)) != hipSuccess) {
        (void)hipFree(d_cand);
        return ADSB_ERR_NOMEM;
    }
    auto body = [&]() -> int {
        HIP_TRY(c, hipMemsetAsync(d_count, 0, sizeof(uint32_t), c->stream));
        ScanParams p{};
        p.src = d_iq;
        p.n_samples = n_samples;
        p.n_chunks = (uint32_t)n_chunks;
        p.bitmap = c->d_bitmap[c->cur_bitmap];  // learned addresses only widen the superset
        p.hits = sl.d_hits;
        p.hits_cap = sl.hits_cap;
        p.ap = sl.d_ap;
        p.ap_cap = c->ap_cap;
        p.seg_cap = c->seg_cap;
        p.tables = c->d_tables;
        p.ctr = sl.d_ctr;
        p.summary = sl.h_sum_dev;
        p.cand_out = d_cand;
        p.cand_count = d_count;
        p.cand_cap = dev_cap;
        if (int e = launch_scan(p, false, c->stream)) return fail(c, (hipError_t)e, "launch_scan");
        Counters ctr;
        uint32_t count = 0;
        HIP_TRY(c, hipMemcpyAsync(&ctr, sl.d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&count, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const bool overflow = ctr.overflow != 0 || count > dev_cap;
        bool inconsistent = false;
        if (!overflow) {
            std::vector<uint64_t> all(count);
            if (count) HIP_TRY(c, hipMemcpy(all.data(), d_cand, (size_t)count * sizeof(uint64_t), hipMemcpyDeviceToHost));
            for (uint64_t e : all) {
                const uint32_t stage = (uint32_t)(e >> 28) & 3u;
                const bool gate = ((e >> 30) & 1u) != 0;
                const uint64_t pos = (e >> 32) << 32 | (e & 0x0FFFFFFFu);
                if (stage >= 1 && pre) pre->push_back(pos);
                if (stage >= 2 && snr) snr->push_back(pos);
                if (gate && cands) cands->push_back(pos);
                // the production gates (gate_eval) and the reference's own sequence (preamble_stage) must agree
                inconsistent = inconsistent || (gate != (stage == 3));
            }
            for (auto *v : {pre, snr, cands})
                if (v) std::sort(v->begin(), v->end());
            if (aps) {
                const std::vector<uint32_t> tab = build_gf_tables();
                const uint32_t *x56 = tab.data() + kTabX56 * 256;
                std::vector<uint64_t> seg(c->seg_cap);
                for (int g = 0; g < kApWaveSegs; g++) {
                    const uint32_t k = ctr.seg_ap[g];
                    if (!k) continue;
                    HIP_TRY(c, hipMemcpy(seg.data(), sl.d_ap + (size_t)g * c->seg_cap, (size_t)k * sizeof(uint64_t),
                                         hipMemcpyDeviceToHost));
                    for (uint32_t i = 0; i < k; i++) {
                        const uint64_t e = seg[i];
                        const uint32_t code = entry_code(e);
                        uint32_t v = entry_value(e);
                        if (code >= 5 && code < 10) v = x56[v & 255u] ^ x56[256 + ((v >> 8) & 255u)] ^ x56[512 + (v >> 16)];
                        aps->push_back(pack_entry(v, entry_tp(e), entry_j(e), entry_chunk(e)));
                    }
                }
                std::sort(aps->begin(), aps->end());
            }
        }
        // put the slot back: the records kernel zeroes this pass's counters on its way out
        p.cand_out = nullptr;
        sl.seq = c->next_seq++;
        if (c->next_seq == 0) c->next_seq = 1;
        sl.h_sum->seq = 0;
        p.seq = sl.seq;
        if (int e = launch_records(p, false, sl.h_rec_dev, c->stream)) return fail(c, (hipError_t)e, "launch_records");
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (overflow) {
            c->last_error = "selftest: the pass overflowed the fast scan's lists";
            return ADSB_ERR_HIP;
        }
        if (inconsistent) {
            c->last_error = "selftest: the scan's gates and the reference's sequence of tests disagree on a position";
            return ADSB_ERR_HIP;
        }
        return ADSB_OK;
    };
    const int rc = body();
    (void)hipFree(d_cand);
    (void)hipFree(d_count);
    return rc;
}

int hand_out(const std::vector<uint64_t> &a, uint64_t *out_a, size_t cap_a, size_t *n_a, const std::vector<uint64_t> &b,
             uint64_t *out_b, size_t cap_b, size_t *n_b)
{
    if (n_a) *n_a = a.size();
    if (n_b) *n_b = b.size();
    if (a.size() > cap_a || b.size() > cap_b) return ADSB_ERR_CAPACITY;
    if (!a.empty()) std::memcpy(out_a, a.data(), a.size() * sizeof(uint64_t));
    if (!b.empty()) std::memcpy(out_b, b.data(), b.size() * sizeof(uint64_t));
    return ADSB_OK;
}

}  // namespace

int adsb_selftest_stage_lists(adsb_ctx *c, const void *d_iq, size_t n_samples, uint64_t *cand, size_t cand_cap,
                              size_t *n_cand, uint64_t *ap, size_t ap_cap, size_t *n_ap)
{
    if ((!cand && cand_cap) || (!ap && ap_cap)) return ADSB_ERR_INVALID;
    std::vector<uint64_t> cands, aps;
    if (int rc = selftest_pass(c, d_iq, n_samples, nullptr, nullptr, &cands, &aps)) return rc;
    return hand_out(cands, cand, cand_cap, n_cand, aps, ap, ap_cap, n_ap);
}

int adsb_selftest_gate_stages(adsb_ctx *c, const void *d_iq, size_t n_samples, uint64_t *preamble, size_t preamble_cap,
                              size_t *n_preamble, uint64_t *snr, size_t snr_cap, size_t *n_snr)
{
    if ((!preamble && preamble_cap) || (!snr && snr_cap)) return ADSB_ERR_INVALID;
    std::vector<uint64_t> pre, sn;
    if (int rc = selftest_pass(c, d_iq, n_samples, &pre, &sn, nullptr, nullptr)) return rc;
    return hand_out(pre, preamble, preamble_cap, n_preamble, sn, snr, snr_cap, n_snr);
}

int adsb_selftest_crc_table(uint32_t *out256)
{
    if (!out256) return ADSB_ERR_INVALID;
    static const Crc24 crc;  // the table the host replay scores with (mode_s_host.hpp)
    std::memcpy(out256, crc.t, sizeof(crc.t));
    return ADSB_OK;
}

static_assert(sizeof(adsb_trial) == sizeof(TrialRecord), "adsb_trial mirrors TrialRecord");

// ---------------------------------------------------------------------------------
// Sharded capture: one capture cut into contiguous ranges of buffers, one
// range per GPU.  The only thing that couples the shards is the order-dependent ICAO
// filter, so a shard runs in two phases around a tiny host-side exchange:
//   adsb_shard_scan    scan the shard; return the addresses its self-validating frames
//                      will add to the filter (DF11 with IID 0, DF17)
//   (exchange)         every shard receives the union of all shards' addresses
//   adsb_shard_finish  add them to the shard's superset bitmap, match the address/parity
//                      trials against it, return the raw trial records
// and whoever holds all records replays them once, in global (chunk, j, try_phase) order,
// through one filter (adsb_replay_records).  The union is a superset in time of what the
// filter can hold at any point of the capture, so the result is the single-stream one.
// ---------------------------------------------------------------------------------
namespace {

// One 131072-sample buffer of a parked shard through the reference-shaped kernel, whose lists
// hold the worst case of a buffer: scan (+ match) + records, synchronously.  The records land
// in the fallback's host buffer (c->fb.h_rec) with chunk = 0; *n_out = how many.
int shard_chunk_pass(adsb_ctx *c, ScanParams p, uint64_t ch, bool with_match, uint32_t *clean, size_t *n_out)
{
    HIP_TRY(c, hipSetDevice(c->device));
    Slot &sl = c->slot[0];
    if (int rc = ensure_fallback(c)) return rc;
    const uint64_t off = ch * kChunkSamples;
    p.src = (const uint32_t *)p.src + off;
    p.n_samples = std::min<uint64_t>(kChunkSamples, p.n_samples - off);
    p.n_chunks = 1;
    p.keep_counters = 0;
    p.clean_bitmap = clean;
    p.hits = c->fb.d_hits;
    p.hits_cap = kWorstPerChunk;
    p.dap = c->fb.d_dap;
    p.dap_cap = kWorstPerChunk;
    sl.seq = c->next_seq++;
    if (c->next_seq == 0) c->next_seq = 1;
    sl.h_sum->seq = 0;
    p.seq = sl.seq;
    if (int e = launch_scan_simple(p, false, c->stream)) return fail(c, (hipError_t)e, "launch_scan_simple");
    if (with_match)
        if (int e = launch_match(p, c->stream)) return fail(c, (hipError_t)e, "launch_match");
    if (int e = launch_records(p, false, c->fb.h_rec_dev, c->stream)) return fail(c, (hipError_t)e, "launch_records");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (__atomic_load_n(&sl.h_sum->seq, __ATOMIC_ACQUIRE) != sl.seq || sl.h_sum->overflow) {
        c->last_error = "shard: a single buffer overflowed the worst-case lists";
        return ADSB_ERR_HIP;
    }
    *n_out = sl.h_sum->n_hits;
    return verify_records(c, sl.h_sum, c->fb.h_rec, *n_out);
}

// mode_s/mod.rs:80-84 (DF11, IID 0) and :97-99 (DF17): the addresses the replay will add
void learned_addresses(const adsb_ctx *c, const TrialRecord *rec, size_t n, std::vector<uint32_t> &addrs)
{
    for (size_t i = 0; i < n; i++) {
        const uint8_t *m = rec[i].msg;
        const uint32_t df = m[0] >> 3;
        const bool adds = df == 17 || (df == 11 && c->crc.residual(m, 7) == 0);
        if (adds) addrs.push_back(uint32_t(m[1]) << 16 | uint32_t(m[2]) << 8 | m[3]);
    }
}

}  // namespace

int adsb_shard_scan(adsb_ctx *c, const void *device_iq, size_t n_samples, uint32_t *addrs_out, size_t cap,
                    size_t *n_addrs)
{
    if (!c || (!device_iq && n_samples) || (!addrs_out && cap)) return ADSB_ERR_INVALID;
    if (c->submitted != c->delivered || c->shard_active) return ADSB_ERR_BUSY;
    if (n_addrs) *n_addrs = 0;
    const uint64_t n_chunks = (n_samples + kChunkSamples - 1) / kChunkSamples;
    if (n_chunks > c->max_chunks || n_chunks > kMaxChunks) return ADSB_ERR_INVALID;
    if ((uintptr_t)device_iq % 16) return ADSB_ERR_INVALID;
    // (the caller may be a worker thread whose current device is not this context's: sharding.ShardPipeline)
    HIP_TRY(c, hipSetDevice(c->device));
    Slot &sl = c->slot[0];
    ScanParams p{};
    p.src = device_iq;
    p.n_samples = n_samples;
    p.n_chunks = (uint32_t)n_chunks;
    p.clean_bitmap = nullptr;
    uint32_t *retired = nullptr;
    if (c->flush_pending) {
        retired = c->d_bitmap[c->cur_bitmap];
        c->cur_bitmap = (c->cur_bitmap + 1) % kBitmaps;
        c->filter.flush();
        c->flush_pending = false;
        // the device-side copy of the filter (exact bitmap, k_score) still holds the addresses from
        // before the flush and was not rotated here: it is rebuilt from the (now empty) host table
        // before the next device-scored pass -- the context is idle, nothing in flight to disown
        c->exact_valid = false;
        ++c->score_epoch;
    }
    p.bitmap = c->d_bitmap[c->cur_bitmap];
    p.hits = sl.d_hits;
    p.hits_cap = sl.hits_cap;
    p.ap = sl.d_ap;
    p.ap_cap = c->ap_cap;
    p.seg_cap = c->seg_cap;
    p.dap = nullptr;  // the reference-shaped kernel's list: shard_chunk_pass() fills it in
    p.dap_cap = 0;
    p.tables = c->d_tables;
    p.ctr = sl.d_ctr;
    p.summary = sl.h_sum_dev;
    p.keep_counters = 1;
    sl.seq = c->next_seq++;
    if (c->next_seq == 0) c->next_seq = 1;
    sl.h_sum->seq = 0;
    p.seq = sl.seq;
    size_t n_hits = 0;
    bool by_chunk = false;
    if (n_chunks) {
        if (int e = launch_scan(p, false, c->stream)) return fail(c, (hipError_t)e, "launch_scan");
        if (int e = launch_records(p, false, sl.h_rec_dev, c->stream)) return fail(c, (hipError_t)e, "launch_records");
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (__atomic_load_n(&sl.h_sum->seq, __ATOMIC_ACQUIRE) != sl.seq) {
            c->last_error = "shard scan completed without publishing its summary";
            return ADSB_ERR_HIP;
        }
        by_chunk = sl.h_sum->overflow != 0;
        n_hits = sl.h_sum->n_hits;
        if (!by_chunk)
            if (int rc = verify_records(c, sl.h_sum, sl.h_rec, n_hits)) return rc;
    }
    std::vector<uint32_t> addrs;
    if (by_chunk) {
        // Far denser than the fast scan's lists are sized for: zero this pass's counters (the
        // records kernel does that on its way out), then both phases go buffer by buffer
        // through the reference-shaped kernel, whose lists hold a buffer's worst case.
        ScanParams q = p;
        q.keep_counters = 0;
        if (int e = launch_records(q, false, sl.h_rec_dev, c->stream)) return fail(c, (hipError_t)e, "launch_records");
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (uint64_t ch = 0; ch < n_chunks; ch++) {
            size_t k = 0;
            if (int rc = shard_chunk_pass(c, p, ch, false, nullptr, &k)) return rc;
            learned_addresses(c, c->fb.h_rec, k, addrs);
        }
    } else {
        learned_addresses(c, sl.h_rec, n_hits, addrs);
    }
    c->shard_by_chunk = by_chunk;
    std::sort(addrs.begin(), addrs.end());
    addrs.erase(std::unique(addrs.begin(), addrs.end()), addrs.end());
    p.clean_bitmap = retired;
    c->shard_params = p;
    c->shard_active = true;
    if (n_addrs) *n_addrs = addrs.size();
    const size_t k = std::min(cap, addrs.size());
    if (k) std::memcpy(addrs_out, addrs.data(), k * sizeof(uint32_t));
    return addrs.size() > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
}

int adsb_shard_finish(adsb_ctx *c, const uint32_t *extra_addrs, size_t n_extra, adsb_trial *records_out,
                      size_t cap, size_t *n_records)
{
    if (!c || (!extra_addrs && n_extra) || (!records_out && cap)) return ADSB_ERR_INVALID;
    if (!c->shard_active) return ADSB_ERR_INVALID;
    if (n_records) *n_records = 0;
    HIP_TRY(c, hipSetDevice(c->device));
    Slot &sl = c->slot[0];
    ScanParams p = c->shard_params;
    p.keep_counters = 0;
    c->shard_active = false;
    if (n_extra) {
        if (n_extra > c->addrs_cap) {
            if (c->d_addrs) (void)hipFree(c->d_addrs);
            c->d_addrs = nullptr;
            c->addrs_cap = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_addrs, n_extra * sizeof(uint32_t)));
            c->addrs_cap = n_extra;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_addrs, extra_addrs, n_extra * sizeof(uint32_t), hipMemcpyHostToDevice,
                                  c->stream));
        if (int e = launch_set_addresses(c->d_addrs, (uint32_t)n_extra, p.bitmap, c->stream))
            return fail(c, (hipError_t)e, "launch_set_addresses");
    }
    bool by_chunk = c->shard_by_chunk;
    c->shard_by_chunk = false;
    size_t n = 0;
    if (!by_chunk && p.n_chunks) {
        sl.seq = c->next_seq++;
        if (c->next_seq == 0) c->next_seq = 1;
        sl.h_sum->seq = 0;
        p.seq = sl.seq;
        if (int e = launch_match(p, c->stream)) return fail(c, (hipError_t)e, "launch_match");
        if (int e = launch_records(p, false, sl.h_rec_dev, c->stream)) return fail(c, (hipError_t)e, "launch_records");
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (__atomic_load_n(&sl.h_sum->seq, __ATOMIC_ACQUIRE) != sl.seq) {
            c->last_error = "shard finish completed without publishing its summary";
            return ADSB_ERR_HIP;
        }
        // the matched address/parity trials did not fit the hit list (a large union of addresses
        // over a dense shard): buffer by buffer, like a shard whose scan overflowed.  (The
        // records kernel has zeroed the counters and cleaned the retired bitmap on its way out.)
        if (sl.h_sum->overflow) {
            by_chunk = true;
            p.clean_bitmap = nullptr;
        }
        n = sl.h_sum->n_hits;
        if (!by_chunk)
            if (int rc = verify_records(c, sl.h_sum, sl.h_rec, n)) return rc;
    }
    if (by_chunk) {
        std::vector<TrialRecord> all;
        uint64_t cand = 0, ap = 0;
        for (uint64_t ch = 0; ch < p.n_chunks; ch++) {
            size_t k = 0;
            uint32_t *clean = ch + 1 == p.n_chunks ? p.clean_bitmap : nullptr;
            if (int rc = shard_chunk_pass(c, p, ch, true, clean, &k)) return rc;
            for (size_t i = 0; i < k; i++) {
                TrialRecord r = c->fb.h_rec[i];
                r.chunk = (uint32_t)ch;
                all.push_back(r);
            }
            cand += sl.h_sum->n_cand_total;
            ap += sl.h_sum->n_ap_total;
        }
        adsb_stats st{};
        st.n_samples = p.n_samples;
        st.n_chunks = p.n_chunks;
        st.n_candidates = cand;
        st.n_ap_entries = ap;
        st.n_records = all.size();
        st.retries = 1;
        c->stats = st;
        if (n_records) *n_records = all.size();
        const size_t k = std::min(cap, all.size());
        if (k) std::memcpy(records_out, all.data(), k * sizeof(adsb_trial));
        return all.size() > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
    }
    adsb_stats st{};
    st.n_samples = p.n_samples;
    st.n_chunks = p.n_chunks;
    st.n_candidates = p.n_chunks ? sl.h_sum->n_cand_total : 0;
    st.n_ap_entries = p.n_chunks ? sl.h_sum->n_ap_total : 0;
    st.n_records = n;
    c->stats = st;
    if (n_records) *n_records = n;
    const size_t k = std::min(cap, n);
    static_assert(sizeof(adsb_trial) == sizeof(TrialRecord), "record layout is the ABI's");
    if (k) std::memcpy(records_out, sl.h_rec, k * sizeof(adsb_trial));
    return n > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
}

int adsb_replay_records(uint32_t *filter_table, adsb_trial *records, size_t n, adsb_msg *out,
                        size_t cap, size_t *n_out)
{
    if (!filter_table || (!records && n) || (!out && cap)) return ADSB_ERR_INVALID;
    static const Crc24 crc;
    IcaoFilter filter;
    filter.load(filter_table);
    std::vector<adsb_msg> msgs;
    replay(filter, crc, reinterpret_cast<TrialRecord *>(records), n, 0, msgs);
    filter.store(filter_table);
    const size_t k = std::min(cap, msgs.size());
    if (k) std::memcpy(out, msgs.data(), k * sizeof(adsb_msg));
    if (n_out) *n_out = msgs.size();
    return msgs.size() > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
}

int adsb_format_raw(const adsb_msg *m, char *out, size_t out_size)
{
    if (!m || !out || (m->len != ADSB_MODES_SHORT_MSG_BYTES && m->len != ADSB_MODES_LONG_MSG_BYTES))
        return ADSB_ERR_INVALID;
    const size_t need = 2u * m->len + 3u;  // '*', hex, ';', '\n'
    if (out_size < need + 1) return ADSB_ERR_CAPACITY;
    static const char digits[] = "0123456789abcdef";  // hex::encode is lowercase
    char *w = out;
    *w++ = '*';
    for (int i = 0; i < m->len; i++) {
        *w++ = digits[m->msg[i] >> 4];
        *w++ = digits[m->msg[i] & 15];
    }
    *w++ = ';';
    *w++ = '\n';
    *w = 0;
    return (int)need;
}

uint64_t adsb_host_sorts(const adsb_ctx *c) { return c ? c->host_sorts : 0; }
uint64_t adsb_host_replays(const adsb_ctx *c) { return c ? c->host_replays : 0; }

int adsb_get_stats(const adsb_ctx *c, adsb_stats *out)
{
    if (!c || !out) return ADSB_ERR_INVALID;
    *out = c->stats;
    return ADSB_OK;
}

const char *adsb_strerror(int status)
{
    switch (status) {
    case ADSB_OK: return "ok";
    case ADSB_ERR_INVALID: return "invalid argument";
    case ADSB_ERR_NO_DEVICE: return "no usable HIP device (libadsb_hip has no CPU fallback)";
    case ADSB_ERR_HIP: return "HIP runtime error";
    case ADSB_ERR_TOO_LONG: return "more than 131072 samples for a single MagnitudeBuffer";
    case ADSB_ERR_CAPACITY: return "output array too small";
    case ADSB_ERR_NOMEM: return "out of memory";
    case ADSB_ERR_BUSY: return "submissions are pending (collect them first) or too many are in flight";
    default: return "unknown status";
    }
}

const char *adsb_last_error(const adsb_ctx *c) { return c ? c->last_error.c_str() : ""; }

const char *adsb_version(void) { return "adsb_hip 0.15 gfx950 scan=v8-gate-reads tail=v3-buckets"; }

}  // extern "C"
