// adsb_shard.cpp -- sharded capture.
// ---------------------------------------------------------------------------------
// One capture cut into contiguous ranges of buffers, one range per GPU.  The only thing that
// couples the shards is the order-dependent ICAO filter (the reference is one loop with one
// process-global filter: dump1090_rs/src/main.rs:154-167, src/icao_filter.rs:8-9), so a shard runs
// in two phases around a tiny host-side exchange:
//   phase 1   scan the shard; return the addresses its self-validating frames will add to the
//             filter (DF11 with IID 0, DF17: src/mode_s/mod.rs:80-84, 97-99)
//   exchange  every shard receives the union of all shards' addresses
//   phase 2   add them to the shard's superset bitmap, match the address/parity trials against it,
//             return the raw trial records
// and whoever holds all records replays them once, in global (chunk, j, try_phase) order, through
// one filter.  The union is a superset in time of what the filter can hold at any point of the
// capture, so the result is the single-stream one.
//
// The phases are written per slot and do not block: shard_begin enqueues phase 1 of a shard into
// slot k on one of the two scan streams -- the scan, which lists the addresses its trials can add
// itself (adsb_multi's shards in contexts of more than 16 buffers: ScanParams::fresh), or the scan
// and the records of its self-validating hits, out of which the host reads them -- and
// returns; the host sees the phase finish by its summary landing in mapped memory
// (shard_phase_landed); shard_match enqueues phase 2 on the tail stream (a dense stream's shards
// hand their records over in replay order: device-side ordering, as dense single-stream passes do).  Shards of consecutive
// captures sit in consecutive slots, so the scan of capture i + 1 runs while capture i is being
// exchanged and matched -- that is what adsb_multi.cpp (one process, N GPUs, a thread per device)
// is built on.  adsb_shard_scan / adsb_shard_finish are the same phases, blocking, in slot 0.
// ---------------------------------------------------------------------------------
#include "adsb_ctx.h"

using namespace adsb::host;

namespace {

uint32_t next_seq(adsb_ctx *c)
{
    const uint32_t s = c->next_seq++;
    if (c->next_seq == 0) c->next_seq = 1;
    return s;
}

// One 131072-sample buffer of a parked shard through the reference-shaped kernel, whose lists
// hold the worst case of a buffer: scan (+ match) + records, synchronously.  The records land
// in the fallback's host buffer (c->fb.h_rec) with chunk = 0; *n_out = how many.
int shard_chunk_pass(adsb_ctx *c, Slot &sl, ScanParams p, uint64_t ch, bool with_match, uint32_t *clean, size_t *n_out)
{
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
    p.fresh = nullptr;           // (the reference-shaped kernel's passes: host-ordered records, addresses read out of them)
    p.order_cnt = p.order_base = nullptr;
    p.order_tmp = nullptr;
    p.hit_fields = nullptr;
    p.score = ScoreDev{};
    sl.seq = next_seq(c);
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

// the rare, blocking path starts from idle streams (the slot's own launches are done by then; what other
// slots still have on the tail stream may be matching against the bitmap this path is about to clean)
int drain_for_chunk_path(adsb_ctx *c, const adsb_ctx::ShardJob &job)
{
    if (job.scan_q) HIP_TRY(c, hipStreamSynchronize(job.scan_q));
    HIP_TRY(c, hipStreamSynchronize(c->tail_stream));
    return ADSB_OK;
}

}  // namespace

namespace adsb {
namespace host {

int shard_begin(adsb_ctx *c, int k, const void *d_iq, uint64_t n_samples, bool fresh_list)
{
    Slot &sl = c->slot[k];
    adsb_ctx::ShardJob &job = c->shard[k];
    if (job.active || sl.busy || sl.parked) return ADSB_ERR_BUSY;
    const uint64_t n_chunks = (n_samples + kChunkSamples - 1) / kChunkSamples;
    if (n_chunks > c->max_chunks || n_chunks > kMaxChunks) return ADSB_ERR_INVALID;
    if (!job.h_addrs) {   // (two lists: ShardJob::addr_half)
        HIP_TRY(c, hipHostMalloc((void **)&job.h_addrs, 2 * kShardAddrCap * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&job.h_addrs_dev, job.h_addrs, 0));
    }
    // (adsb_multi's shards in contexts of more than 16 buffers: the scan lists the addresses its trials can add -- all the
    // exchange needs -- and phase 1 is the scan alone; adsb_device.h: ScanParams::fresh.  Contexts for passes of a few
    // buffers read them out of their handful of records.)
    fresh_list = fresh_list && c->bitmap_lg == kFullBitmapLg;
    if (fresh_list && !job.h_fresh) {
        HIP_TRY(c, hipHostMalloc((void **)&job.h_fresh, kShardAddrCap * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&job.h_fresh_dev, job.h_fresh, 0));
        HIP_TRY(c, hipMalloc((void **)&job.d_fresh_seen, kBitmapAllocWords * sizeof(uint32_t)));   // (also the `earlier` set of a scored shard)
        HIP_TRY(c, hipHostMalloc((void **)&job.h_earlier, kShardAddrCap * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&job.h_earlier_dev, job.h_earlier, 0));
    }
    job.fresh_list = fresh_list;
    ScanParams p{};
    p.src = d_iq;
    p.n_samples = n_samples;
    p.n_chunks = (uint32_t)n_chunks;
    p.clean_bitmap = nullptr;
    job.retired = nullptr;
    bool fresh = false;
    p.bitmap_lg = c->bitmap_lg;
    if (c->flush_pending) {
        // (full bitmaps: the retired one is cleaned by the second phase's records kernel; folded ones: the next one
        // is cleared in front of this shard's scan -- enqueue_pass has the argument)
        if (c->bitmap_lg == kFullBitmapLg) job.retired = c->d_bitmap[c->cur_bitmap];
        else fresh = true;
        c->cur_bitmap = (c->cur_bitmap + 1) % c->n_bitmaps;
        c->filter.flush();
        c->flush_pending = false;
        // the device-side copy of the filter (exact bitmap, k_score) still holds the addresses from
        // before the flush and was not rotated here: it is rebuilt from the (now empty) host table
        // before the next device-scored pass
        c->exact_valid = false;
        ++c->score_epoch;
    }
    // (adsb_multi's contexts keep their exact bitmap themselves -- every capture's additions are committed to it behind
    // the capture's scoring, shard_match -- and an icao_flush switches to the other one, cleared in the second phase)
    job.exact_flush = fresh_list && c->score.si && (job.retired != nullptr);
    if (job.exact_flush) c->cur_exact ^= 1;
    job.exact = fresh_list && c->score.si ? c->exact_bm[c->cur_exact] : nullptr;
    job.scored = job.wait_score = job.result_scored = false;
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
    if (fresh_list) {
        p.fresh = job.h_fresh_dev;
        p.fresh_seen = job.d_fresh_seen;
        p.fresh_cap = c->shard_fresh_cap ? std::min<uint32_t>(c->shard_fresh_cap, (uint32_t)kShardAddrCap) : (uint32_t)kShardAddrCap;
        // a dense stream's shards: hits straight into their buffers' buckets, the second phase's records in replay
        // order (enqueue_pass has the rules; only where phase 1 has no records kernel of its own to disturb the buckets)
        if (c->shard_dense && n_chunks > kInlineTailChunks && sl.hits_cap == c->hits_cap) {
            p.order_cnt = sl.d_order_cnt;
            p.order_base = sl.d_order_base;
            p.order_tmp = sl.d_order_tmp;
            p.hit_fields = sl.d_hit_fields;
            c->shard_device_ordered++;
            // ... and scored on the device: k_score / k_emit behind the second phase's records kernel, against the
            // context's exact bitmap and the additions of the shards before this one (ScoreDev::earlier)
            if (c->shard_scoring && c->score.si) {
                job.scored = true;
                p.score = sl.score;
                p.score.exact = job.exact;
                p.score.exact_retired = nullptr;
                p.score.earlier = job.d_fresh_seen;
                p.score.out_msgs = sl.h_msgs_dev;
                p.score.out_adds = sl.h_adds_dev;
                p.score.summary = sl.h_ssum_dev;
            }
        }
    }
    sl.seq = next_seq(c);
    sl.h_sum->seq = 0;
    p.seq = sl.seq;
    job.p = p;
    job.by_chunk = false;
    job.chunk_records.clear();
    job.n_cand = job.n_ap = 0;
    job.scan_q = nullptr;
    job.waiting = false;
    job.active = true;
    if (!n_chunks) {   // (an empty shard: nothing to wait for, nothing learned)
        if (fresh) {
            if (int e = launch_reset(sl.d_ctr, p.bitmap, p.bitmap_lg, c->tail_stream)) return fail(c, (hipError_t)e, "launch_reset");
            HIP_TRY(c, hipStreamSynchronize(c->tail_stream));
        }
        return ADSB_OK;
    }
    // consecutive shards' scans alternate between the first two scan streams, like consecutive passes
    hipStream_t ss = c->scan_stream[c->shard_jobs++ % 2];
    job.scan_q = ss;
    // the caller's samples are complete where `stream` stands now (nothing to wait for on the context's own,
    // idle stream: see enqueue_pass)
    if (!(c->stream == c->own_stream && !c->own_stream_dirty)) {
        hipEvent_t ready = c->input_ready[ss == c->scan_stream[0] ? 0 : 1];
        HIP_TRY(c, hipEventRecord(ready, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(ss, ready, 0));
        c->own_stream_dirty = false;
    }
    if (int rc = order_behind_fused(c, sl, ss)) return rc;
    // Phase 1 stays on its scan stream, no event: the tail stream is in order, and the first-phase records of the
    // NEXT shards (enqueued microseconds after this one's, each waiting for its own scan) would sit in front of
    // this shard's second phase there -- which the host only enqueues when this phase has landed -- so that every
    // capture's match waited for the scans of the three captures behind it and the pipeline ran in bursts of four
    // (measured: 0.99 ms per 2 GiB capture against 0.76 of scan; profiles/r5_multi_overhead.txt).
    // (the slot's previous shard: its second phase ran on the tail stream, and its summary reached the host a
    // moment before that launch retired -- it was still zeroing the counters this scan is about to fill)
    if (job.ran) HIP_TRY(c, hipStreamWaitEvent(ss, sl.recorded, 0));
    if (fresh)
        if (int e = launch_reset(sl.d_ctr, p.bitmap, p.bitmap_lg, ss)) return fail(c, (hipError_t)e, "launch_reset");
    if (fresh_list) HIP_TRY(c, hipMemsetAsync(job.d_fresh_seen, 0, (size_t(1) << 24) / 8, ss));
    if (int e = launch_scan(p, false, ss)) return fail(c, (hipError_t)e, "launch_scan");
    if (fresh_list) {
        if (int e = launch_shard_summary(p, ss)) return fail(c, (hipError_t)e, "launch_shard_summary");
    } else {
        if (int e = launch_records(p, false, sl.h_rec_dev, ss)) return fail(c, (hipError_t)e, "launch_records");
    }
    HIP_TRY(c, hipEventRecord(sl.scanned, ss));   // (the second phase, on the tail stream, orders itself behind this launch)
    job.waiting = true;
    return ADSB_OK;
}

bool shard_phase_landed(adsb_ctx *c, int k)
{
    adsb_ctx::ShardJob &job = c->shard[k];
    if (!job.waiting) return true;
    if (!summary_landed(c->slot[k].h_sum, c->slot[k].seq)) return false;
    // (a scored shard's second phase ends with k_emit's summary)
    if (job.wait_score && __atomic_load_n(&c->slot[k].h_ssum->seq, __ATOMIC_ACQUIRE) != c->slot[k].seq) return false;
    job.waiting = false;
    return true;
}

int shard_phase_wait(adsb_ctx *c, int k)
{
    adsb_ctx::ShardJob &job = c->shard[k];
    if (!job.waiting) return ADSB_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; spin++) {
        if (shard_phase_landed(c, k)) return ADSB_OK;
        __builtin_ia32_pause();
        if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    if (job.scan_q) HIP_TRY(c, hipStreamSynchronize(job.scan_q));
    HIP_TRY(c, hipStreamSynchronize(c->tail_stream));
    if (job.wait_score) HIP_TRY(c, hipStreamSynchronize(c->score_stream));
    for (int attempt = 0; attempt < 200; attempt++) {
        if (shard_phase_landed(c, k)) return ADSB_OK;
        for (volatile int spin = 0; spin < 2000; spin++) {}
    }
    c->last_error = "shard phase completed without publishing a whole summary";
    return ADSB_ERR_HIP;
}

int shard_phase_check(adsb_ctx *c, int k)
{
    if (shard_phase_landed(c, k)) return 1;
    adsb_ctx::ShardJob &job = c->shard[k];
    // the streams either phase can have launches on (the tail and score streams are shared with the other slots'
    // shards -- and, in a process with several contexts on the device, with theirs: busy means "look again later")
    hipStream_t qs[3] = {job.scan_q, c->tail_stream, job.wait_score ? c->score_stream : nullptr};
    for (hipStream_t q : qs) {
        if (!q) continue;
        const hipError_t e = hipStreamQuery(q);
        if (e == hipErrorNotReady) {
            (void)hipGetLastError();
            return 0;
        }
        if (e != hipSuccess) return fail(c, e, "hipStreamQuery (a shard phase that has not landed)");
    }
    // idle streams: the summary was written before the launch retired
    for (int attempt = 0; attempt < 200; attempt++) {
        if (shard_phase_landed(c, k)) return 1;
        for (volatile int spin = 0; spin < 2000; spin++) {}
    }
    c->last_error = "shard phase completed without publishing a whole summary";
    return ADSB_ERR_HIP;
}

int shard_reset(adsb_ctx *c)
{
    for (int k = 0; k < c->n_scan_streams; k++) HIP_TRY(c, hipStreamSynchronize(c->scan_stream[k]));
    HIP_TRY(c, hipStreamSynchronize(c->tail_stream));
    HIP_TRY(c, hipStreamSynchronize(c->score_stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipStream_t q = c->own_stream;
    for (int k = 0; k < c->n_bitmaps; k++)
        if (int e = launch_reset(c->slot[0].d_ctr, c->d_bitmap[k], c->bitmap_lg, q)) return fail(c, (hipError_t)e, "launch_reset");
    for (uint32_t *bm : c->exact_bm)
        if (bm) HIP_TRY(c, hipMemsetAsync(bm, 0, kBitmapAllocWords * sizeof(uint32_t), q));
    for (int si = 0; si < c->n_slots; si++) {
        Slot &sl = c->slot[si];
        adsb_ctx::ShardJob &job = c->shard[si];
        // (what a phase that never ran its records kernel / k_order_prefix / k_emit leaves behind)
        HIP_TRY(c, hipMemsetAsync(sl.d_ctr, 0, sizeof(Counters), q));
        HIP_TRY(c, hipMemsetAsync(sl.d_order_cnt, 0, (c->max_chunks * fastgeo::kTilesPerChunk + 1) * sizeof(uint32_t), q));
        if (sl.score.hash) HIP_TRY(c, hipMemsetAsync(sl.score.hash, 0xFF, ((size_t)sl.score.hash_mask + 1) * sizeof(unsigned long long), q));
        if (sl.score.state) HIP_TRY(c, hipMemsetAsync(sl.score.state, 0, sizeof(ScoreState), q));
        sl.h_sum->seq = 0;
        if (sl.h_ssum) sl.h_ssum->seq = 0;
        job.active = job.waiting = job.by_chunk = job.ran = false;
        job.scored = job.wait_score = job.result_scored = job.exact_flush = false;
        job.retired = nullptr;
        job.exact = nullptr;
        job.chunk_records.clear();
    }
    HIP_TRY(c, hipStreamSynchronize(q));
    c->shard_active = false;
    c->cur_exact = 0;
    c->exact_valid = false;
    ++c->score_epoch;
    c->filter.flush();
    c->flush_pending = false;   // (everything is clean: the capture behind this starts from an empty filter as it is)
    c->shard_dense = false;
    return ADSB_OK;
}

// after phase 1 has landed: the addresses this shard's replay can add (from the list its scan made, or out of its
// records), sorted, no duplicates
int shard_learned(adsb_ctx *c, int k, std::vector<uint32_t> &addrs)
{
    Slot &sl = c->slot[k];
    adsb_ctx::ShardJob &job = c->shard[k];
    addrs.clear();
    if (!job.active || job.waiting) return ADSB_ERR_INVALID;
    const ScanParams &p = job.p;
    if (!p.n_chunks) return ADSB_OK;
    job.by_chunk = sl.h_sum->overflow != 0;
    bool from_records = !job.fresh_list;
    if (!job.by_chunk && job.fresh_list) {
        const size_t n_fresh = sl.h_sum->n_dap;   // (k_shard_summary: count in n_dap, sum of the addresses in rec_sum_lo)
        if (n_fresh <= p.fresh_cap) {
            // the list and the summary are separate posted writes: the list is whole when it adds up
            bool whole = false;
            for (int attempt = 0; attempt < 200 && !whole; attempt++) {
                uint32_t sum = 0;
                for (size_t i = 0; i < n_fresh; i++) sum += __atomic_load_n(&job.h_fresh[i], __ATOMIC_RELAXED);
                whole = sum == sl.h_sum->rec_sum_lo;
                if (!whole)
                    for (volatile int spin = 0; spin < 2000; spin++) {}
            }
            if (!whole) {
                c->last_error = "shard: the fresh addresses in host memory do not add up to the sum of the scan that listed them";
                return ADSB_ERR_HIP;
            }
            addrs.assign(job.h_fresh, job.h_fresh + n_fresh);
        } else {
            // more aircraft than the list holds: the records of the self-validating hits after all, behind the scan, and
            // the addresses out of those
            c->shard_fresh_fallbacks++;
            sl.seq = next_seq(c);
            sl.h_sum->seq = 0;
            job.p.seq = sl.seq;
            ScanParams q1 = job.p;
            q1.score = ScoreDev{};   // (the first phase's records go to the host, whoever scores the second's)
            if (int e = launch_order_hits(q1, job.scan_q)) return fail(c, (hipError_t)e, "launch_order_hits");   // (device-ordered: the buckets' places)
            if (int e = launch_records(q1, false, sl.h_rec_dev, job.scan_q)) return fail(c, (hipError_t)e, "launch_records");
            HIP_TRY(c, hipEventRecord(sl.scanned, job.scan_q));
            job.waiting = true;
            if (int rc = shard_phase_wait(c, k)) return rc;
            from_records = true;
        }
    }
    if (!job.by_chunk && from_records) {
        const size_t n_hits = sl.h_sum->n_hits;
        if (int rc = verify_records(c, sl.h_sum, sl.h_rec, n_hits)) return rc;
        learned_addresses(c->crc, sl.h_rec, n_hits, addrs);
    } else if (job.by_chunk) {
        // Far denser than the fast scan's lists are sized for: zero this pass's counters (the
        // records kernel does that on its way out; a device-ordered pass's bucket counts: k_order_prefix), then both
        // phases go buffer by buffer through the reference-shaped kernel, whose lists hold a buffer's worst case.
        if (int rc = drain_for_chunk_path(c, job)) return rc;
        ScanParams q = p;
        q.keep_counters = 0;
        q.score = ScoreDev{};
        job.scored = false;
        job.p.score = ScoreDev{};
        if (int e = launch_order_hits(q, c->stream)) return fail(c, (hipError_t)e, "launch_order_hits");
        if (int e = launch_records(q, false, sl.h_rec_dev, c->stream)) return fail(c, (hipError_t)e, "launch_records");
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (uint64_t ch = 0; ch < p.n_chunks; ch++) {
            size_t n = 0;
            if (int rc = shard_chunk_pass(c, sl, p, ch, false, nullptr, &n)) return rc;
            learned_addresses(c->crc, c->fb.h_rec, n, addrs);
        }
    }
    std::sort(addrs.begin(), addrs.end());
    addrs.erase(std::unique(addrs.begin(), addrs.end()), addrs.end());
    return ADSB_OK;
}

int shard_match(adsb_ctx *c, int k, const uint32_t *extra, size_t n_extra, const uint32_t *earlier, size_t n_earlier)
{
    Slot &sl = c->slot[k];
    adsb_ctx::ShardJob &job = c->shard[k];
    if (!job.active || job.waiting) return ADSB_ERR_INVALID;
    ScanParams &p = job.p;
    hipStream_t ts = c->tail_stream, qs = c->score_stream;
    p.keep_counters = 0;
    p.clean_bitmap = job.retired;
    job.wait_score = job.result_scored = false;
    // the other shards' addresses join this shard's superset: read by the kernel from the slot's mapped host
    // buffer, no copy command (a capture teaches a few hundred at most).  The buffer holds two lists, used in turn: the
    // same list is committed to the exact bitmap behind this shard's scoring, on the score stream, and that launch may
    // still be reading when the slot's next shard writes its own (the one after that is behind it in stream order).
    const uint32_t *d_list = nullptr;
    int used_half = -1;
    if (n_extra && n_extra <= kShardAddrCap) {
        const int half = job.addr_half;
        uint32_t *h_list = job.h_addrs + (size_t)half * kShardAddrCap;
        d_list = job.h_addrs_dev + (size_t)half * kShardAddrCap;
        job.addr_half ^= 1;
        if (job.addr_read[half]) HIP_TRY(c, hipEventSynchronize(job.addr_read[half]));   // (two uses ago: long done)
        used_half = half;
        std::memcpy(h_list, extra, n_extra * sizeof(uint32_t));
        if (int e = launch_set_addresses(d_list, (uint32_t)n_extra, p.bitmap, p.bitmap_lg, ts))
            return fail(c, (hipError_t)e, "launch_set_addresses");
    } else if (n_extra) {
        // (more than a capture can teach: a caller's own union through adsb_shard_finish -- a device buffer, blocking)
        if (n_extra > c->addrs_cap) {
            if (c->d_addrs) (void)hipFree(c->d_addrs);
            c->d_addrs = nullptr;
            c->addrs_cap = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_addrs, n_extra * sizeof(uint32_t)));
            c->addrs_cap = n_extra;
        }
        HIP_TRY(c, hipStreamSynchronize(ts));   // (d_addrs may still be read by an earlier shard's launches)
        HIP_TRY(c, hipStreamSynchronize(qs));
        HIP_TRY(c, hipMemcpy(c->d_addrs, extra, n_extra * sizeof(uint32_t), hipMemcpyHostToDevice));
        d_list = c->d_addrs;
        if (int e = launch_set_addresses(c->d_addrs, (uint32_t)n_extra, p.bitmap, p.bitmap_lg, ts))
            return fail(c, (hipError_t)e, "launch_set_addresses");
    }
    // What the score stream gets, in its order = the filter's order: [an icao_flush: the exact bitmap this capture starts
    // from, cleared] [this shard's scoring against it] [the capture's additions -- every shard's, the exchange has them --
    // committed to it, for the captures behind this one].
    auto exact_side = [&](bool behind_records) -> int {
        if (!job.exact) return ADSB_OK;
        if (behind_records) HIP_TRY(c, hipStreamWaitEvent(qs, sl.recorded, 0));
        if (job.exact_flush) HIP_TRY(c, hipMemsetAsync(job.exact, 0, kBitmapAllocWords * sizeof(uint32_t), qs));
        if (job.scored && behind_records) {
            HIP_TRY(c, hipMemsetAsync(job.d_fresh_seen, 0, kBitmapAllocWords * sizeof(uint32_t), qs));
            if (n_earlier) {
                std::memcpy(job.h_earlier, earlier, n_earlier * sizeof(uint32_t));
                if (int e = launch_set_addresses(job.h_earlier_dev, (uint32_t)n_earlier, job.d_fresh_seen, kFullBitmapLg, qs))
                    return fail(c, (hipError_t)e, "launch_set_addresses");
            }
            sl.h_ssum->seq = 0;
            p.score.seq = sl.seq;
            if (int e = launch_score(p, qs)) return fail(c, (hipError_t)e, "launch_score");
            job.wait_score = true;
            c->shard_device_scored++;
        }
        if (d_list) {
            if (int e = launch_set_addresses(d_list, (uint32_t)n_extra, job.exact, kFullBitmapLg, qs))
                return fail(c, (hipError_t)e, "launch_set_addresses");
            if (used_half >= 0) {
                if (!job.addr_read[used_half])
                    HIP_TRY(c, hipEventCreateWithFlags(&job.addr_read[used_half], hipEventDisableTiming | hipEventDisableSystemFence));
                HIP_TRY(c, hipEventRecord(job.addr_read[used_half], qs));
            }
        }
        return ADSB_OK;
    };
    if (job.scored && n_earlier > kShardAddrCap) {   // (more aircraft before this shard than the list holds: the host scores it)
        job.scored = false;
        p.score = ScoreDev{};
    }
    if (!p.n_chunks) {
        // an empty shard has no records kernel to clean the bitmap a flush retired
        if (job.retired)
            if (int e = launch_reset(sl.d_ctr, job.retired, p.bitmap_lg, ts)) return fail(c, (hipError_t)e, "launch_reset");
        if (int rc = exact_side(false)) return rc;
        if (n_extra || job.retired) HIP_TRY(c, hipStreamSynchronize(ts));   // (nothing else to wait for)
        return ADSB_OK;
    }
    if (!job.by_chunk) {
        sl.seq = next_seq(c);
        sl.h_sum->seq = 0;
        p.seq = sl.seq;
        // (the first phase's records kernel published its summary a moment before it retired: it resets the
        // block counter this phase's records kernel counts in)
        HIP_TRY(c, hipStreamWaitEvent(ts, sl.scanned, 0));
        if (int e = launch_match(p, ts)) return fail(c, (hipError_t)e, "launch_match");
        if (int e = launch_order_hits(p, ts)) return fail(c, (hipError_t)e, "launch_order_hits");   // (device-ordered shards only)
        if (int e = launch_records(p, false, sl.h_rec_dev, ts)) return fail(c, (hipError_t)e, "launch_records");
        HIP_TRY(c, hipEventRecord(sl.recorded, ts));
        job.ran = true;
        if (int rc = exact_side(true)) return rc;
        job.waiting = true;
    } else {
        if (int rc = exact_side(false)) return rc;
    }
    return ADSB_OK;
}

// after phase 2 has landed: the shard's raw trial records (chunk = buffer index within the shard); they stay
// valid until the slot's next shard_begin
int shard_records(adsb_ctx *c, int k, const TrialRecord **rec, size_t *n_out)
{
    Slot &sl = c->slot[k];
    adsb_ctx::ShardJob &job = c->shard[k];
    *rec = nullptr;
    *n_out = 0;
    if (!job.active || job.waiting) return ADSB_ERR_INVALID;
    ScanParams &p = job.p;
    job.active = false;
    adsb_stats st{};
    st.n_samples = p.n_samples;
    st.n_chunks = p.n_chunks;
    if (!p.n_chunks) {
        c->stats = st;
        return ADSB_OK;
    }
    if (!job.by_chunk) {
        // the matched address/parity trials did not fit the hit list (a large union of addresses
        // over a dense shard): buffer by buffer, like a shard whose scan overflowed.  (The
        // records kernel has zeroed the counters and cleaned the retired bitmap on its way out.)
        if (sl.h_sum->overflow) {
            job.by_chunk = true;
            p.clean_bitmap = nullptr;
        } else {
            const size_t n = sl.h_sum->n_hits;
            // (a shard that k_score took has left its records in device memory: shard_scored_result has what it made of
            // them, shard_fetch_records brings them over if that cannot be used)
            job.result_scored = job.wait_score && sl.h_ssum->scored != 0 && n <= c->score.cap;
            if (!job.result_scored)
                if (int rc = verify_records(c, sl.h_sum, sl.h_rec, n)) return rc;
            st.n_candidates = sl.h_sum->n_cand_total;
            st.n_ap_entries = sl.h_sum->n_ap_total;
            st.n_records = n;
            c->stats = st;
            *rec = job.result_scored ? nullptr : sl.h_rec;
            *n_out = job.result_scored ? 0 : n;
            // (density = records per buffer, with the single stream's thresholds and hysteresis: adsb_collect.cpp)
            if (p.n_chunks > kInlineTailChunks) {
                if (n >= 8u * (size_t)p.n_chunks) c->shard_dense = true;
                else if (n < 2u * (size_t)p.n_chunks) c->shard_dense = false;
            }
            return ADSB_OK;
        }
    }
    if (int rc = drain_for_chunk_path(c, job)) return rc;
    std::vector<TrialRecord> &all = job.chunk_records;
    all.clear();
    for (uint64_t ch = 0; ch < p.n_chunks; ch++) {
        size_t n = 0;
        uint32_t *clean = ch + 1 == p.n_chunks ? p.clean_bitmap : nullptr;
        if (int rc = shard_chunk_pass(c, sl, p, ch, true, clean, &n)) return rc;
        for (size_t i = 0; i < n; i++) {
            TrialRecord r = c->fb.h_rec[i];
            r.chunk = (uint32_t)ch;
            all.push_back(r);
        }
        st.n_candidates += sl.h_sum->n_cand_total;
        st.n_ap_entries += sl.h_sum->n_ap_total;
    }
    st.n_records = all.size();
    st.retries = 1;
    c->stats = st;
    *rec = all.data();
    *n_out = all.size();
    return ADSB_OK;
}

bool shard_scored_result(adsb_ctx *c, int k, adsb_msg **msgs, size_t *n_msgs, const uint32_t **adds, size_t *n_adds)
{
    Slot &sl = c->slot[k];
    const adsb_ctx::ShardJob &job = c->shard[k];
    if (!job.result_scored) return false;
    const ScoreSummary *ss = sl.h_ssum;
    const size_t nm = ss->n_msgs, na = ss->n_adds;
    if (nm > c->score.cap || na > c->score.cap) return false;
    // the messages and the summary are separate posted writes: the list is whole when it adds up (adsb_collect.cpp)
    const uint64_t want = (uint64_t)ss->msg_sum_hi << 32 | ss->msg_sum_lo;
    bool whole = false;
    for (int attempt = 0; attempt < 200 && !whole; attempt++) {
        uint64_t got = 0;
        const uint64_t *w = reinterpret_cast<const uint64_t *>(sl.h_msgs);
        for (size_t i = 0; i < 5 * nm; i++) got += __atomic_load_n(&w[i], __ATOMIC_RELAXED);
        whole = got == want;
    }
    if (!whole) return false;
    *msgs = sl.h_msgs;
    *n_msgs = nm;
    *adds = sl.h_adds;
    *n_adds = na;
    return true;
}

int shard_fetch_records(adsb_ctx *c, int k, const TrialRecord **rec, size_t *n_out)
{
    Slot &sl = c->slot[k];
    const size_t n = sl.h_sum->n_hits;
    *rec = nullptr;
    *n_out = 0;
    if (n > c->score.cap) return ADSB_ERR_INVALID;
    if (n) {
        HIP_TRY(c, hipMemcpy(sl.h_rec, sl.score.rec, n * sizeof(TrialRecord), hipMemcpyDeviceToHost));
        if (int rc = verify_records(c, sl.h_sum, sl.h_rec, n)) return rc;
    }
    *rec = sl.h_rec;
    *n_out = n;
    return ADSB_OK;
}

}  // namespace host
}  // namespace adsb

extern "C" {

int adsb_shard_scan(adsb_ctx *c, const void *device_iq, size_t n_samples, uint32_t *addrs_out, size_t cap,
                    size_t *n_addrs)
try {
    if (!c || (!device_iq && n_samples) || (!addrs_out && cap)) return ADSB_ERR_INVALID;
    if (c->submitted != c->delivered || c->shard_active || c->shard[0].active) return ADSB_ERR_BUSY;
    if (n_addrs) *n_addrs = 0;
    if ((uintptr_t)device_iq % 16) return ADSB_ERR_INVALID;
    // (the caller may be a worker thread whose current device is not this context's: sharding.ShardPipeline)
    ADSB_ON_DEVICE(c);
    if (int rc = shard_begin(c, 0, device_iq, n_samples)) return rc;
    std::vector<uint32_t> addrs;
    int rc = shard_phase_wait(c, 0);
    if (rc == ADSB_OK) rc = shard_learned(c, 0, addrs);
    if (rc != ADSB_OK) {
        c->shard[0].active = c->shard[0].waiting = false;
        return rc;
    }
    c->shard_active = true;
    if (n_addrs) *n_addrs = addrs.size();
    const size_t k = std::min(cap, addrs.size());
    if (k) std::memcpy(addrs_out, addrs.data(), k * sizeof(uint32_t));
    return addrs.size() > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
} ADSB_ABI_CATCH

int adsb_shard_finish(adsb_ctx *c, const uint32_t *extra_addrs, size_t n_extra, adsb_trial *records_out,
                      size_t cap, size_t *n_records)
try {
    if (!c || (!extra_addrs && n_extra) || (!records_out && cap)) return ADSB_ERR_INVALID;
    if (!c->shard_active) return ADSB_ERR_INVALID;
    if (n_records) *n_records = 0;
    ADSB_ON_DEVICE(c);
    c->shard_active = false;
    const TrialRecord *rec = nullptr;
    size_t n = 0;
    int rc = shard_match(c, 0, extra_addrs, n_extra);
    if (rc == ADSB_OK) rc = shard_phase_wait(c, 0);
    if (rc == ADSB_OK) rc = shard_records(c, 0, &rec, &n);
    if (rc != ADSB_OK) {
        c->shard[0].active = c->shard[0].waiting = false;
        return rc;
    }
    if (n_records) *n_records = n;
    const size_t k = std::min(cap, n);
    static_assert(sizeof(adsb_trial) == sizeof(TrialRecord), "record layout is the ABI's");
    if (k) std::memcpy(records_out, rec, k * sizeof(adsb_trial));
    return n > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
} ADSB_ABI_CATCH

}  // extern "C"
