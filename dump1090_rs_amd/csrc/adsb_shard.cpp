// adsb_shard.cpp -- sharded capture.
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
#include "adsb_ctx.h"

using namespace adsb::host;

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

}  // namespace

extern "C" {

static_assert(sizeof(adsb_trial) == sizeof(TrialRecord), "adsb_trial mirrors TrialRecord");

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
    if (int rc = order_behind_slot0(c)) return rc;
    Slot &sl = c->slot[0];
    ScanParams p{};
    p.src = device_iq;
    p.n_samples = n_samples;
    p.n_chunks = (uint32_t)n_chunks;
    p.clean_bitmap = nullptr;
    uint32_t *retired = nullptr;
    if (c->flush_pending) {
        retired = c->d_bitmap[c->cur_bitmap];
        c->cur_bitmap = (c->cur_bitmap + 1) % c->n_bitmaps;
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
            learned_addresses(c->crc, c->fb.h_rec, k, addrs);
        }
    } else {
        learned_addresses(c->crc, sl.h_rec, n_hits, addrs);
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

}  // extern "C"
