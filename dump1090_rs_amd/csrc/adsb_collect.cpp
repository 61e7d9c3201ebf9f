// adsb_collect.cpp -- waiting for a pass, its checksums, the ordered host replay (src/demod_2400.rs:149-207 with
// src/mode_s/mod.rs:34-139 scoring against src/icao_filter.rs), and the overflow fallback.
#include "adsb_ctx.h"

using namespace adsb::host;

namespace {

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
    const uint64_t before = c->filter.inserts();
    for (size_t i = 0; i < na; i++) c->filter.add(sl.h_adds[i]);
    if (c->filter.inserts() != before) c->last_new_insert_seq = sl.scan_seq;
    return true;
}

}  // namespace

namespace adsb {
namespace host {

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

// The summary is ten separate posted writes with the sequence word issued last; a host that polls for it
// takes the summary only when its own checksum covers what it reads (a word that overtook its neighbours
// on the way would otherwise pair this pass's sequence number with the previous pass's counts).
bool summary_landed(const Summary *s, uint32_t seq)
{
    if (__atomic_load_n(&s->seq, __ATOMIC_ACQUIRE) != seq) return false;
    uint32_t w[10];
    const uint32_t *src = reinterpret_cast<const uint32_t *>(s);
    for (int k = 0; k < 10; k++) w[k] = __atomic_load_n(&src[k], __ATOMIC_RELAXED);
    return w[7] == seq && summary_check(w) == w[9];
}

// Wait for the pass in `sl` and replay it.  Returns 1 when a device list overflowed
// (caller re-runs in smaller pieces), 0 on success, < 0 on error.
int finish_pass(adsb_ctx *c, Slot &sl, uint64_t chunk_offset, adsb_stats &st, std::vector<adsb_msg> &out)
{
#ifdef ADSB_TUNING
    const auto tw0 = std::chrono::steady_clock::now();
#endif
    {
        HT(c, HT_SYNC);
        if (sl.fused) {
            // A one-launch pass has no event behind it: its last workgroup writes the summary into mapped
            // host memory (after the records, with their checksum), and that is what the host watches for.
            // Give up spinning after a while (a pass queued behind long ones) and block on its stream.
            const auto t0 = std::chrono::steady_clock::now();
            bool seen = false;
            for (uint32_t spin = 0; !seen; spin++) {
                seen = summary_landed(sl.h_sum, sl.seq);
                if (seen) break;
                __builtin_ia32_pause();
                if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
            }
            if (!seen) HIP_TRY(c, hipStreamSynchronize(sl.tail_q));
        } else {
            HIP_TRY(c, hipEventSynchronize(sl.done));
        }
    }
#ifdef ADSB_TUNING
    c->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
#endif
    {
        // (behind an event or a stream synchronisation every word has landed; the retries are for the
        // polled case that fell through to the stream wait with the last words still on their way)
        bool whole = false;
        for (int attempt = 0; attempt < 200 && !whole; attempt++) {
            whole = summary_landed(sl.h_sum, sl.seq);
            if (!whole) for (volatile int spin = 0; spin < 2000; spin++) {}
        }
        if (!whole) {
            c->last_error = "pass completed without publishing a whole summary";
            return ADSB_ERR_HIP;
        }
    }
    if (sl.h_sum->overflow) return 1;
    // A one-launch pass did not wait for the passes that were in flight on the other scan stream.  If one
    // of those (all replayed by now: passes are collected in order) taught the filter a NEW address, this
    // pass's address/parity trials may have been matched before its bit was set: again, through the three
    // launches (every earlier pass is complete now, so nothing can be missed a second time).
    if (sl.fused && sl.unsynced_from && c->last_new_insert_seq >= sl.unsynced_from) return 2;
    const size_t n = sl.h_sum->n_hits;
    // (k_records' own test: a pass that k_score took over has left its records in device memory)
    const bool rec_on_device = sl.device_scored && n <= c->score.cap;
    if (!rec_on_device) {
        HT(c, HT_VERIFY);
        if (int rc = verify_records(c, sl.h_sum, sl.h_rec, n)) return rc;
    }
    if (sl.profiled && sl.fused) {
        // a one-launch pass timed itself (device wall clock, 100 MHz): no events, nothing to wait for
        const float ms = (float)sl.h_sum->ticks * 1e-5f;
        st.ms_scan += ms;
        st.ms_scan_exclusive += ms;
    } else if (sl.profiled) {
        float ms = 0;
        HIP_TRY(c, hipEventElapsedTime(&ms, sl.ev[0], sl.ev[1]));
        st.ms_scan += ms;
        // The device time this launch adds: the part of it after the latest scan end seen so far (the
        // "frontier": normally the previous pass's; scans on the two scan streams can also finish out of
        // order, and one that ended before the frontier adds nothing -- the union of the launches'
        // intervals is what is being summed).  The frontier's events are intact for kScanEvRing - n_slots
        // passes back: the ring is that much longer than what can be in flight.
        float excl = ms;
        bool advance = !sl.redo;   // (a pass run again is blocking and on events of its own: no part of the frontier)
        if (!sl.redo && c->last_stop && sl.scan_seq - c->last_scan_seq <= (uint64_t)(kScanEvRing - c->n_slots)) {
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
    {
        HT(c, HT_REPLAY);
        const uint64_t before = c->filter.inserts();
        if (!skip_replay) replay(c->filter, c->crc, sl.h_rec, n, chunk_offset, out, &c->host_sorts);
        if (c->filter.inserts() != before) c->last_new_insert_seq = sl.scan_seq;
    }
#ifdef ADSB_TUNING
    c->t_replay += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
#endif
    return 0;
}

// Finish the oldest submission: replay it, or -- when a device list overflowed (far
// denser input than the lists were sized for) -- drain the stream and go chunk by
// chunk, where the worst case always fits.  Bitmap bits set by the aborted pass or by
// later passes are a harmless superset in time.
int collect_oldest(adsb_ctx *c, std::vector<adsb_msg> &out)
{
    Slot &sl = c->slot[c->collected % (uint64_t)c->n_slots];
    adsb_stats st{};
    st.n_samples = sl.n_samples;
    st.n_chunks = sl.n_chunks;
    if (sl.flush_before) c->filter.flush();  // icao_flush() took effect before this pass
    int rc = finish_pass(c, sl, 0, st, out);
    if (rc == 2) {
        // a one-launch pass that a pass in flight beside it may have invalidated (finish_pass): once more
        // through the three launches.  Passes submitted after it may have rotated the bitmaps behind an
        // icao_flush: as for the overflow fallback, drain the streams and put the filter's addresses (the
        // state that preceded this pass) back into the bitmap in use.
        c->rematches++;
        for (hipStream_t q : c->scan_stream)
            if (q) HIP_TRY(c, hipStreamSynchronize(q));
        HIP_TRY(c, hipStreamSynchronize(c->tail_stream));
        HIP_TRY(c, hipStreamSynchronize(c->score_stream));
        const bool keep_flush = c->flush_pending;
        c->flush_pending = false;
        rc = reseed_bitmap_from_filter(c);
        if (rc == 0)
            rc = enqueue_pass(c, sl, sl.src, sl.from_mag, sl.n_samples, sl.n_chunks, true, false, false, false,
                              input_ready_now(), true);
        if (rc == 0) rc = finish_pass(c, sl, 0, st, out);
        c->flush_pending = keep_flush;
    }
    if (rc > 0 && sl.from_mag) {  // a caller-supplied buffer denser than the fast scan's lists: again, the slow way
        st.retries++;
        for (hipStream_t q : c->scan_stream)
            if (q) HIP_TRY(c, hipStreamSynchronize(q));
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
        for (hipStream_t q : c->scan_stream)
            if (q) HIP_TRY(c, hipStreamSynchronize(q));  // later passes have their results on the host
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
        Slot &sl = c->slot[c->collected % (uint64_t)c->n_slots];
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
        Slot &sl = c->slot[c->delivered % (uint64_t)c->n_slots];
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

}  // namespace host
}  // namespace adsb

extern "C" {

int adsb_collect(adsb_ctx *c, adsb_msg *out, size_t cap, size_t *n_out)
try {
    if (!c || (!out && cap)) return ADSB_ERR_INVALID;
    if (c->submitted == c->delivered) return ADSB_ERR_INVALID;
    ADSB_ON_DEVICE(c);
    std::vector<adsb_msg> msgs;
    int rc = collect_next(c, msgs);
    if (rc) return rc;
    return deliver(c, msgs, out, cap, n_out);
} ADSB_ABI_CATCH

int adsb_pending(const adsb_ctx *c) { return c ? (int)(c->submitted - c->delivered) : 0; }

int adsb_fetch_messages(adsb_ctx *c, adsb_msg *out, size_t cap, size_t *n_out)
try {
    if (!c || (!out && cap) || !c->has_undelivered) return ADSB_ERR_INVALID;
    const size_t n = std::min(cap, c->undelivered.size());
    if (n) std::memcpy(out, c->undelivered.data(), n * sizeof(adsb_msg));
    if (n_out) *n_out = c->undelivered.size();
    return c->undelivered.size() > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
} ADSB_ABI_CATCH

}  // extern "C"
