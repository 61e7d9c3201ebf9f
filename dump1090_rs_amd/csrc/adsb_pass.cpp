// adsb_pass.cpp -- one device pass: what is enqueued on which stream (every cross-stream event edge is here;
// DESIGN.md section 5b is the table to review it against), submit, and the blocking entry points
// (reference: src/utils.rs:43 to_mag, src/demod_2400.rs:115 demodulate2400, dump1090_rs/src/main.rs:166-167).
#include "adsb_ctx.h"

using namespace adsb::host;

namespace adsb {
namespace host {

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

// `waiter` waits until `other`'s pass is through matching (its records kernel has finished): the event
// behind a three-launch pass, or -- a one-launch pass records none -- an event put on its stream now.
int wait_for_tail_of(adsb_ctx *c, hipStream_t waiter, Slot &other)
{
    if (other.fused) {
        HIP_TRY(c, hipEventRecord(c->lazy_ev, other.tail_q));
        HIP_TRY(c, hipStreamWaitEvent(waiter, c->lazy_ev, 0));
    } else {
        HIP_TRY(c, hipStreamWaitEvent(waiter, other.recorded, 0));
    }
    return ADSB_OK;
}

int order_behind_fused(adsb_ctx *c, Slot &sl, hipStream_t waiter)
{
    if (sl.fused_q && sl.fused_q != waiter) {
        HIP_TRY(c, hipEventRecord(c->lazy_ev, sl.fused_q));
        HIP_TRY(c, hipStreamWaitEvent(waiter, c->lazy_ev, 0));
    }
    sl.fused_q = nullptr;
    return ADSB_OK;
}

// Enqueue one device pass over n_chunks chunks starting at d_src into `sl`:
// reset -> scan -> dense -> match -> records -> D2H of the summary and the first records.
// Whether a plain pass of n_chunks buffers submitted now goes out as one launch.
bool one_launch_pass(const adsb_ctx *c, uint32_t n_chunks)
{
    static const bool never_fuse = tuning_env("ADSB_NO_FUSE") != nullptr;
    return !never_fuse && c->profiling <= 1 && n_chunks <= (uint32_t)kInlineTailChunks;
}

// The scan stream the next plain IQ pass of n_chunks buffers will run on (enqueue_pass's own rule).
hipStream_t next_scan_stream(const adsb_ctx *c, uint32_t n_chunks)
{
    static const int fused_streams = tuning_env("ADSB_FUSED_STREAMS") ? std::atoi(tuning_env("ADSB_FUSED_STREAMS")) : kScanStreams;
    static const bool one_scan_stream = tuning_env("ADSB_ONE_SCAN_STREAM") != nullptr;
    if (c->carry_over || one_scan_stream) return c->scan_stream[0];
    const int period = one_launch_pass(c, n_chunks) ? std::max(1, std::min(fused_streams, c->n_scan_streams)) : 2;
    return c->scan_stream[c->submitted % (uint64_t)period];
}

int enqueue_pass(adsb_ctx *c, Slot &sl, const void *d_src, bool from_mag, uint64_t n_samples,
                 uint32_t n_chunks, bool inline_tail, bool lead_from_src,
                 bool advance_carry, bool force_simple, hipEvent_t input_done, bool no_fuse)
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
    p.bitmap_lg = c->bitmap_lg;
    p.bitmap_fresh = 0;
    if (c->flush_pending) {  // icao_flush: retire the bitmap in use, continue on the next one
        // Full bitmaps (2 MiB): the next one IS clean, and this pass's records kernel cleans the retired one behind the
        // passes still matching against it (edge 1).  Folded ones (64 KB, contexts for passes of a few buffers): the
        // retired one is left as it is and this pass clears the NEXT one itself before it first touches it -- in its
        // own launch (k_scan_fast<FUSED>: bitmap_fresh) or with a reset launch in front of its scan.  Whoever used
        // that bitmap has been collected (one bitmap more than passes in flight), so the pass waits for nobody:
        // an icao_flush before every pass, the reference's own benchmark shape (benches/demod_benchmark.rs:9), used
        // to cost a pipelined one-buffer pass 99 us instead of 5.7 (profiles/r4_v18_hosttime_ring.txt).
        if (c->bitmap_lg == kFullBitmapLg) p.clean_bitmap = c->d_bitmap[c->cur_bitmap];
        else p.bitmap_fresh = 1;
        c->cur_bitmap = (c->cur_bitmap + 1) % c->n_bitmaps;
    }
    p.bitmap = c->d_bitmap[c->cur_bitmap];
    p.hits = sl.d_hits;
    p.hits_cap = sl.hits_cap;
    // (the slot's own hit list only: the fallback's worst-case list is filled by the reference-shaped kernel)
    static const bool no_fields = tuning_env("ADSB_NO_HIT_FIELDS") != nullptr;
    p.hit_fields = nullptr;   // (set below, once it is known whether the pass is a dense stream's or one launch)
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
    // A pass collect_oldest runs again (overflow fallback buffer by buffer, rematch: `advance_carry` is false for
    // exactly those) keeps its number -- a fresh one per buffer would walk through the event ring under the
    // passes still in flight and move last_new_insert_seq ahead of them -- and times itself with its own pair.
    sl.redo = !advance_carry;
    if (!sl.redo) {
        sl.scan_seq = ++c->scan_counter;
        sl.ev[0] = c->scan_ev[sl.scan_seq % kScanEvRing][0];
        sl.ev[1] = c->scan_ev[sl.scan_seq % kScanEvRing][1];
    } else {
        sl.ev[0] = c->redo_ev[0];
        sl.ev[1] = c->redo_ev[1];
    }
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
    // A pass of a few buffers is all launch overhead and event traffic: it goes out as ONE launch whose
    // last workgroup matches, builds the records and publishes the summary (k_scan_fast<.., FUSED>), with
    // no event behind it.  (Level-2 profiling wants the three kernels apart.)
    const bool fused = !force_simple && !no_fuse && one_launch_pass(c, n_chunks);
    sl.fused = fused;
    sl.unsynced_from = 0;
    p.fused_rec = fused ? sl.h_rec_dev : nullptr;
    p.order_polls = c->order_polls;
    p.src_ready = fused && !from_mag ? c->next_src_ready : nullptr;
    c->next_src_ready = nullptr;
    static const bool no_trickle = tuning_env("ADSB_NO_TRICKLE") != nullptr;
    p.src_host = fused && !from_mag && c->next_src_host && !no_trickle ? 1u : 0u;
    c->next_src_host = false;
    // the scan hands the bit fields of its self-validating hits to the record builder: where the record
    // builder's instructions matter (dense streams: it shares the vector pipes with the next scan) and in
    // one-launch passes; a sparse stream's scan stays the lean instantiation
    if (fused || (order_on_device && !no_fields)) p.hit_fields = sl.d_hit_fields;
    // (a one-launch pass times itself on the device's wall clock and reports it with its summary)
    p.ev_start = ext_events && prof == 1 && fast && !fused ? sl.ev[0] : nullptr;
    p.ev_stop = ext_events && prof == 1 && fast && !fused ? sl.ev[1] : nullptr;

    c->flush_pending = false;
    const bool classic = !fused && (prof > 1 || (prof == 1 && (!fast || !ext_events)));
    // odd slots scan on the second stream, unless something orders consecutive passes (the
    // carry hand-off) or the pass is a one-off (fallback, caller-supplied magnitudes)
    static const bool one_scan_stream = tuning_env("ADSB_ONE_SCAN_STREAM") != nullptr;
    static const int fused_streams = tuning_env("ADSB_FUSED_STREAMS") ? std::atoi(tuning_env("ADSB_FUSED_STREAMS")) : kScanStreams;
    const bool rotate = fast && !p.carry && advance_carry && !one_scan_stream;
    // (a slot's passes of one kind always land on the same stream: the slot count is a multiple of both periods)
    const int si = !rotate ? 0 : (int)(c->submitted % (uint64_t)(fused ? std::max(1, std::min(fused_streams, c->n_scan_streams)) : 2));
    hipStream_t ss = c->scan_stream[si];
    // the input is complete at `input_done` (the ring's copy), already (input_ready_now: pinned memory the
    // host has filled), or where `stream` stands now
    if (input_done != input_ready_now()) {
        HT(c, HT_IN_READY);
        hipEvent_t ready = input_done;
        // No event at all when `stream` is the context's own and the library has put nothing on it that this
        // pass could depend on (own_stream_dirty: the copies of the host-pointer entry points): the record +
        // wait pair is two thirds of what a one-launch pass costs the submitting thread.  (hipStreamQuery is
        // no substitute: on a stream that has seen work it took ~25 us, measured.)  A caller's stream
        // (adsb_set_stream) is always waited for.
        const bool nothing_to_wait_for = !ready && c->stream == c->own_stream && !c->own_stream_dirty;
        if (!ready && !nothing_to_wait_for) {
            ready = c->input_ready[si];
            HIP_TRY(c, hipEventRecord(ready, c->stream));
            c->own_stream_dirty = false;   // (what was on it is now behind this pass)
        }
        if (ready) HIP_TRY(c, hipStreamWaitEvent(ss, ready, 0));
    }
    // (a ring slot's copy was queued on the stream adsb_ring_submit expected this pass to take -- the same rule as
    // above, in next_scan_stream(); should the two ever disagree, the pass waits for that stream)
    if (c->input_on_stream && c->input_on_stream != ss) {
        HIP_TRY(c, hipEventRecord(c->lazy_ev, c->input_on_stream));
        HIP_TRY(c, hipStreamWaitEvent(ss, c->lazy_ev, 0));
    }
    // (0) the slot's lists and counters: a one-launch pass that used them last on another stream may still
    //     be zeroing them (the host goes by its summary, which it writes just before)
    if (sl.fused_q && sl.fused_q != ss) {
        HIP_TRY(c, hipEventRecord(c->lazy_ev, sl.fused_q));
        HIP_TRY(c, hipStreamWaitEvent(ss, c->lazy_ev, 0));
    }
    sl.fused_q = fused ? ss : nullptr;
    if (n_chunks <= kInlineTailChunks) inline_tail = true;
    // ---- cross-stream edges (DESIGN.md section 5b is the table of them) -------------------------------
    // (1) A bitmap an icao_flush retired is cleared by this pass (its records kernel, or every workgroup
    //     of a one-launch pass): not before the passes still in flight, whichever stream their tail is on,
    //     are through matching against it.  A tail on this same in-order stream is behind us already.
    // (2) The match must see every address bit the scans of this and of all earlier passes set.  Behind
    //     its own scan it is in stream order or waits for `scanned`.  Behind the previous pass's scan it is
    //     in order when both matches run on the tail stream (that pass's match waited for its scan); a pass
    //     whose match runs on its own scan stream waits for the previous scan explicitly when that ran on
    //     the other one.  (Scans before the previous one are behind this pass's scan or the previous
    //     pass's, on the same two streams.)
    // (3) One-launch passes record no event and do not wait for each other across the two scan streams:
    //     a three-launch pass that needs one of them behind it records an event on that stream now; a
    //     one-launch pass notes from which pass on its match is unsynchronised (Slot::unsynced_from) and
    //     the host redoes it if one of those turns out to have taught the filter a new address.
    bool behind_fresh = false;   // this launch has been ordered behind the pass that opened the epoch
    if (c->bitmap_lg != kFullBitmapLg) {
        const long my_slot = &sl - c->slot;
        if (p.bitmap_fresh) {   // this pass opens the filter's next epoch and clears its bitmap
            c->fresh_q = ss;
            c->fresh_seq = c->epoch_first_seq = sl.scan_seq;
            c->fresh_slot = my_slot >= 0 && my_slot < kSlots ? (int)my_slot : -1;
        } else if (c->fresh_q && c->fresh_q != ss && c->fresh_slot >= 0 && c->fresh_slot != my_slot) {
            const Slot &f = c->slot[c->fresh_slot];
            if (f.busy && f.scan_seq == c->fresh_seq) {   // still in flight: behind it (rare: the first passes after a flush)
                HIP_TRY(c, hipEventRecord(c->lazy_ev, c->fresh_q));
                HIP_TRY(c, hipStreamWaitEvent(ss, c->lazy_ev, 0));
                behind_fresh = true;
            }
        }
    }
    if (fused) {
        HT(c, HT_EV_SCANNED);
        if (p.clean_bitmap)
            for (Slot &other : c->slot)
                if (&other != &sl && other.busy && other.tail_q != ss)
                    if (int rc = wait_for_tail_of(c, ss, other)) return rc;
        bool other_stream_synced = false;
        if (c->prev_scan_stream && c->prev_scan_stream != ss && !c->prev_fused && c->prev_scanned) {
            HIP_TRY(c, hipStreamWaitEvent(ss, c->prev_scanned, 0));
            other_stream_synced = true;   // (in stream order behind it: every earlier pass on that stream)
        }
        // (every pass in flight whose scan is not in stream order before this launch: not on this stream, and not
        // on the stream of the three-launch pass just waited for -- that wait covers what ran on ITS stream only)
        // (... and that belongs to the filter's current epoch: what a pass from before the latest icao_flush taught
        // the filter is gone when this pass is replayed -- with a flush before every pass, the reference's benchmark
        // shape, every pass used to be redone because its predecessor had "taught a new address")
        for (Slot &other : c->slot)
            if (&other != &sl && other.busy && other.scan_q != ss && other.scan_seq >= c->epoch_first_seq &&
                !(other_stream_synced && other.scan_q == c->prev_scan_stream) && !(behind_fresh && other.scan_seq == c->fresh_seq))
                sl.unsynced_from = sl.unsynced_from ? std::min(sl.unsynced_from, other.scan_seq) : other.scan_seq;
    }
    sl.scan_q = ss;
    if (p.bitmap_fresh && !fused) {
        // (three launches in a context of folded bitmaps -- level-2 profiling, a caller's MagnitudeBuffer that
        // overflowed -- : the clear as a launch of its own in front of the scan; the counters it also zeroes are zero)
        if (int e = launch_reset(sl.d_ctr, p.bitmap, p.bitmap_lg, ss)) return fail(c, (hipError_t)e, "launch_reset");
        p.bitmap_fresh = 0;
    }
    if (classic) HIP_TRY(c, hipEventRecord(sl.ev[0], ss));
    if (p.carry && advance_carry)  // this pass's lead-in: where the previous submission ended
        HIP_TRY(c, hipMemcpyAsync(sl.d_carry, c->d_carry_next, kCarrySamples * sizeof(uint32_t),
                                  hipMemcpyDeviceToDevice, ss));
    {
        HT(c, HT_SCAN_LAUNCH);
        if (int e = fused ? launch_pass_fused(p, from_mag, ss)
                          : (force_simple ? launch_scan_simple(p, from_mag, ss) : launch_scan(p, from_mag, ss)))
            return fail(c, (hipError_t)e, "launch_scan");
    }
    if (classic) HIP_TRY(c, hipEventRecord(sl.ev[1], ss));
    if (p.carry && advance_carry) {
        // the next submission starts from the end of this one's input (taken now: the caller
        // may reuse the buffer as soon as this pass is collected)
        if (int e = launch_update_carry(sl.d_carry, d_src, n_samples, c->d_carry_next, ss))
            return fail(c, (hipError_t)e, "launch_update_carry");
    }
    if (fused) {
        sl.tail_q = ss;
        c->prev_scanned = nullptr;
        c->prev_scan_stream = ss;
        c->prev_inline = true;
        c->prev_fused = true;
        return ADSB_OK;
    }
    // the tail runs on its own stream behind the scan: the next pass's scan does not wait
    // for it (it works on the other slot's lists and counters)
    // (a blocking call has nothing to overlap with: its tail stays on the scan stream and
    // saves the cross-stream hand-off)
    // A small pass is all launch overhead: its tail stays on its scan stream too (the two scan
    // streams still let consecutive passes overlap), which saves the cross-stream hand-off.
    hipStream_t ts = inline_tail ? ss : c->tail_stream;
    if (p.clean_bitmap && fast) {
        // edge (1): the event behind the other passes' records kernels, not `done`, which device-scored
        // passes record later, on the score stream
        for (Slot &other : c->slot)
            if (&other != &sl && other.busy && other.tail_q != ts)
                if (int rc = wait_for_tail_of(c, ts, other)) return rc;
    }
    {
        HT(c, HT_EV_SCANNED);
        HIP_TRY(c, hipEventRecord(sl.scanned, ss));
        if (!inline_tail) HIP_TRY(c, hipStreamWaitEvent(ts, sl.scanned, 0));
        // edge (2)
        if (c->prev_scanned && !c->prev_fused && c->prev_scan_stream != ss && (inline_tail || c->prev_inline))
            HIP_TRY(c, hipStreamWaitEvent(ts, c->prev_scanned, 0));
        // edge (3): one-launch passes in flight on a stream this match is not behind
        for (hipStream_t q : c->scan_stream) {
            if (!q || q == ts || (q == ss && !inline_tail)) continue;   // (behind its own scan: behind everything on that stream)
            bool any = false;
            for (Slot &other : c->slot) any = any || (&other != &sl && other.busy && other.fused && other.tail_q == q);
            if (!any) continue;
            HIP_TRY(c, hipEventRecord(c->lazy_ev, q));
            HIP_TRY(c, hipStreamWaitEvent(ts, c->lazy_ev, 0));
        }
    }
    c->prev_scanned = sl.scanned;
    c->prev_scan_stream = ss;
    c->prev_inline = inline_tail;
    c->prev_fused = false;
    if (prof > 1) HIP_TRY(c, hipEventRecord(sl.ev[2], ts));
    static const bool skip_match = tuning_env("ADSB_SKIP_MATCH") != nullptr;  // measurement aid (tuning build only): wrong results
    {
        HT(c, HT_MATCH_LAUNCH);
        if (!skip_match)
            if (int e = launch_match(p, ts)) return fail(c, (hipError_t)e, "launch_match");
        if (int e = launch_order_hits(p, ts)) return fail(c, (hipError_t)e, "launch_order_hits");
    }
    if (prof > 1) HIP_TRY(c, hipEventRecord(sl.ev[3], ts));
    // the records kernel writes the records and the summary into the slot's mapped host
    // memory with write-through stores; `done` only has to say the kernel has drained
    {
        HT(c, HT_RECORDS_LAUNCH);
        if (int e = launch_records(p, from_mag, sl.h_rec_dev, ts))
            return fail(c, (hipError_t)e, "launch_records");
    }
    sl.tail_q = ts;
    // (always: a later pass whose own tail runs on another stream -- a small one behind an icao_flush --
    // waits for this event before its records kernel clears the bitmap this pass matched against)
    {
        HT(c, HT_EV_DONE);
        HIP_TRY(c, hipEventRecord(sl.recorded, ts));
    }
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
    {
        HT(c, HT_EV_DONE);
        HIP_TRY(c, hipEventRecord(sl.done, ts));
    }
    return ADSB_OK;
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
        if (int e = launch_set_addresses(c->d_addrs, (uint32_t)addrs.size(), c->exact_bm[c->cur_exact], kFullBitmapLg, ts))
            return fail(c, (hipError_t)e, "launch_set_addresses");
    }
    HIP_TRY(c, hipStreamSynchronize(ts));  // (rare: only after the host scored a pass itself)
    c->exact_valid = true;
    return ADSB_OK;
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
    if (int e = launch_set_addresses(c->d_addrs, (uint32_t)addrs.size(), c->d_bitmap[c->cur_bitmap], c->bitmap_lg, c->scan_stream[0]))
        return fail(c, (hipError_t)e, "launch_set_addresses");
    return ADSB_OK;
}

int submit(adsb_ctx *c, const void *d_src, bool from_mag, uint64_t n_samples, bool inline_tail,
           hipEvent_t input_done)
{
    const uint64_t n_chunks = from_mag ? 1 : (n_samples + kChunkSamples - 1) / kChunkSamples;
    if (n_chunks == 0 || n_chunks > kMaxChunks || n_chunks > c->max_chunks) return ADSB_ERR_INVALID;
    Slot &sl = c->slot[c->submitted % (uint64_t)c->n_slots];
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
int run_sync(adsb_ctx *c, const void *d_src, bool from_mag, uint64_t n_samples, std::vector<adsb_msg> &out,
             hipEvent_t input_done)
{
    if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
    int rc = submit(c, d_src, from_mag, n_samples, true, input_done);
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

// pinned, mapped staging for host-pointer calls of a few buffers: the pass reads it in place over the
// link, which saves the copy command and its event
int ensure_host_stage(adsb_ctx *c, size_t bytes)
{
    if (bytes <= c->h_stage_bytes) return ADSB_OK;
    if (c->h_stage) HIP_TRY(c, hipHostFree(c->h_stage));
    c->h_stage = c->h_stage_dev = nullptr;
    c->h_stage_bytes = 0;
    HIP_TRY(c, hipHostMalloc(&c->h_stage, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(c, hipHostGetDevicePointer(&c->h_stage_dev, c->h_stage, 0));
    c->h_stage_bytes = bytes;
    return ADSB_OK;
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

}  // namespace host
}  // namespace adsb

extern "C" {

int adsb_to_mag(adsb_ctx *c, const int16_t *iq, size_t n, uint16_t *data_out, size_t *length_out)
try {
    if (!c || (!iq && n) || !data_out) return ADSB_ERR_INVALID;
    if (n > kChunkSamples) return ADSB_ERR_TOO_LONG;  // reference: index panic, lib.rs:48
    ADSB_ON_DEVICE(c);
    // Through pinned, mapped memory both ways: one host copy in, the kernel reads the samples and writes
    // the 131398 magnitudes in place over the link, one host copy out -- instead of two copy commands
    // from / to pageable memory with their staging inside the runtime (62 -> ~36 us for the reference's
    // own buffer size).
    const size_t in_bytes = (size_t)kChunkSamples * 4, out_bytes = (size_t)kMagDataLen * sizeof(uint16_t);
    const size_t out_off = (in_bytes + 255) & ~(size_t)255;
    if (int rc = ensure_host_stage(c, out_off + out_bytes)) return rc;
    if (n) std::memcpy(c->h_stage, iq, n * 4);
    uint16_t *h_mag = reinterpret_cast<uint16_t *>((char *)c->h_stage + out_off);
    uint16_t *h_mag_dev = reinterpret_cast<uint16_t *>((char *)c->h_stage_dev + out_off);
    if (int e = launch_to_mag(c->h_stage_dev, (uint32_t)n, h_mag_dev, c->stream))
        return fail(c, (hipError_t)e, "launch_to_mag");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::memcpy(data_out, h_mag, out_bytes);
    if (length_out) *length_out = n;
    return ADSB_OK;
} ADSB_ABI_CATCH

int adsb_demodulate2400(adsb_ctx *c, const uint16_t *data, size_t length, adsb_msg *out, size_t cap,
                        size_t *n_out)
try {
    if (!c || !data || (!out && cap)) return ADSB_ERR_INVALID;
    if (length > kChunkSamples) return ADSB_ERR_TOO_LONG;
    ADSB_ON_DEVICE(c);
    if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
    c->stats = adsb_stats{};
    c->stats.n_samples = length;
    c->stats.n_chunks = 1;
    std::vector<adsb_msg> msgs;
    if (length) {
        // the caller's MagnitudeBuffer into pinned memory; the pass (one launch) reads it in place
        const size_t bytes = (size_t)kMagDataLen * sizeof(uint16_t);
        if (int rc = ensure_host_stage(c, bytes)) return rc;
        std::memcpy(c->h_stage, data, bytes);
        int rc = run_sync(c, c->h_stage_dev, true, length, msgs, input_ready_now());
        if (rc) return rc;
    }
    return deliver(c, msgs, out, cap, n_out);
} ADSB_ABI_CATCH

int adsb_demod_iq_device(adsb_ctx *c, const void *d_iq, size_t n_samples, adsb_msg *out, size_t cap,
                         size_t *n_out)
try {
    if (!c || (!d_iq && n_samples) || (!out && cap)) return ADSB_ERR_INVALID;
    if (((uintptr_t)d_iq & 15u) != 0) return ADSB_ERR_INVALID;
    ADSB_ON_DEVICE(c);
    std::vector<adsb_msg> msgs;
    int rc = demod_device(c, d_iq, n_samples, msgs);
    if (rc) return rc;
    return deliver(c, msgs, out, cap, n_out);
} ADSB_ABI_CATCH

int adsb_submit_iq_device(adsb_ctx *c, const void *d_iq, size_t n_samples)
try {
    if (!c || !d_iq || n_samples == 0) return ADSB_ERR_INVALID;
    if (((uintptr_t)d_iq & 15u) != 0) return ADSB_ERR_INVALID;
    if ((n_samples + kChunkSamples - 1) / kChunkSamples > std::min<uint64_t>(kMaxChunks, c->max_chunks))
        return ADSB_ERR_INVALID;  // more buffers than the context was created for
    ADSB_ON_DEVICE(c);
    return submit(c, d_iq, false, n_samples);
} ADSB_ABI_CATCH

int adsb_demod_iq(adsb_ctx *c, const int16_t *iq, size_t n_samples, adsb_msg *out, size_t cap,
                  size_t *n_out)
try {
    if (!c || (!iq && n_samples) || (!out && cap)) return ADSB_ERR_INVALID;
    ADSB_ON_DEVICE(c);
    // stage through the device in pieces of at most max_chunks chunks
    std::vector<adsb_msg> msgs;
    adsb_stats total{};
    const size_t piece = c->max_chunks * (size_t)kChunkSamples;
    // A call of a few buffers (the reference's own call shape, benches/demod_benchmark.rs:10-11: one
    // 131072-sample buffer) is one launch that reads the samples in place from pinned host memory: one
    // host copy into it instead of a copy command, its staging inside the runtime and an event.
    // Samples inside a buffer the caller registered (adsb_host_register) are pinned and mapped already: one launch
    // that reads them where they are, no host copy at all.
    if (n_samples && n_samples <= std::min<size_t>(piece, (size_t)kInlineTailChunks * kChunkSamples) && !c->carry_over &&
        ((uintptr_t)iq & 15u) == 0) {
        const char *b = reinterpret_cast<const char *>(iq);
        for (const auto &r : c->host_ranges)
            if (b >= r.base && b + n_samples * 4 <= r.base + r.bytes) {
                if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
                int rc = run_sync(c, r.dev + (b - r.base), false, n_samples, msgs, input_ready_now());
                if (rc) return rc;
                c->stats.n_samples = n_samples;
                return deliver(c, msgs, out, cap, n_out);
            }
    }
    const bool in_place = n_samples <= (size_t)kInlineTailChunks * kChunkSamples && !c->carry_over;
    int rc = in_place ? ensure_host_stage(c, std::max<size_t>(n_samples, 1) * 4 + 512)   // (+ the progress word)
                      : ensure_stage(c, std::min(piece, std::max<size_t>(n_samples, 1)) * 4);
    if (rc) return rc;
    if (in_place && n_samples && n_samples <= piece &&
        one_launch_pass(c, (uint32_t)((n_samples + kChunkSamples - 1) / kChunkSamples))) {
        // One pass of one launch: launch it FIRST and copy the samples into the pinned buffer while the launch is on its way
        // (dispatch latency ~5 us, the copy ~10): each workgroup waits for the host's progress word to pass the
        // end of its tile (ScanParams::src_ready), so the copy and the first tiles overlap instead of adding up.
        const size_t ready_off = (n_samples * 4 + 255) & ~(size_t)255;
        if (int rc2 = ensure_host_stage(c, ready_off + 64)) return rc2;
        unsigned long long *ready = reinterpret_cast<unsigned long long *>((char *)c->h_stage + ready_off);
        __atomic_store_n(ready, 0ull, __ATOMIC_RELEASE);
        if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
        c->next_src_ready = reinterpret_cast<const unsigned long long *>((char *)c->h_stage_dev + ready_off);
        rc = submit(c, c->h_stage_dev, false, n_samples, true, input_ready_now());
        c->next_src_ready = nullptr;
        // (whatever submit said, the copy is finished before anything else: a pass that was launched reads it)
        constexpr size_t kStep = 16384;   // samples per progress update: 64 KB, two tiles
        for (size_t done = 0; done < n_samples;) {
            const size_t k = std::min(kStep, n_samples - done);
            std::memcpy((char *)c->h_stage + done * 4, iq + 2 * done, k * 4);
            done += k;
            __atomic_store_n(ready, (unsigned long long)done, __ATOMIC_RELEASE);
        }
        if (rc) return rc;
        rc = collect_next(c, msgs);
        if (rc) return rc;
        c->stats.n_samples = n_samples;
        return deliver(c, msgs, out, cap, n_out);
    }
    if (in_place && n_samples) {
        std::memcpy(c->h_stage, iq, n_samples * 4);
        for (size_t off = 0; off < n_samples; off += piece) {
            const size_t n = std::min(piece, n_samples - off);
            std::vector<adsb_msg> part;
            rc = run_sync(c, (const uint32_t *)c->h_stage_dev + off, false, n, part, input_ready_now());
            if (rc) return rc;
            const uint64_t chunk0 = off / kChunkSamples;
            for (auto &m : part) {
                m.chunk += chunk0;
                msgs.push_back(m);
            }
            const adsb_stats &st = c->stats;
            total.n_chunks += st.n_chunks, total.n_candidates += st.n_candidates, total.n_ap_entries += st.n_ap_entries;
            total.n_records += st.n_records, total.ms_scan += st.ms_scan, total.ms_scan_exclusive += st.ms_scan_exclusive;
            total.ms_match += st.ms_match, total.ms_records += st.ms_records, total.ms_total_device += st.ms_total_device;
            total.retries += st.retries;
        }
        total.n_samples = n_samples;
        c->stats = total;
        return deliver(c, msgs, out, cap, n_out);
    }
    for (size_t off = 0; off < n_samples; off += piece) {
        const size_t n = std::min(piece, n_samples - off);
        HIP_TRY(c, hipMemcpyAsync(c->d_stage, iq + 2 * off, n * 4, hipMemcpyHostToDevice, c->stream));
        c->own_stream_dirty = true;
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
} ADSB_ABI_CATCH

}  // extern "C"
