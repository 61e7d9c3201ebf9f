// adsb_selftest.cpp -- device self-tests: stage lists and digests that let the tests compare stages, not only frames.
#include "adsb_ctx.h"

using namespace adsb::host;

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
    ADSB_ON_DEVICE(c);
    if (int rc = order_behind_slot0(c)) return rc;
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
        p.bitmap_lg = c->bitmap_lg;
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

extern "C" {

int adsb_selftest_mag_digest(adsb_ctx *c, uint32_t first_bits, uint32_t count, uint64_t *sum_out,
                             uint64_t *xor_out)
try {
    if (!c || !sum_out || !xor_out) return ADSB_ERR_INVALID;
    if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
    ADSB_ON_DEVICE(c);
    // the counters block doubles as the 16-byte result area
    static_assert(sizeof(Counters) >= 16, "digest result fits the counters block");
    // the counters block the next pass will use doubles as the 16-byte result area; it is
    // zeroed again afterwards
    Counters *scratch = c->slot[c->submitted % (uint64_t)c->n_slots].d_ctr;
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
} ADSB_ABI_CATCH

int adsb_selftest_stage_lists(adsb_ctx *c, const void *d_iq, size_t n_samples, uint64_t *cand, size_t cand_cap,
                              size_t *n_cand, uint64_t *ap, size_t ap_cap, size_t *n_ap)
try {
    if ((!cand && cand_cap) || (!ap && ap_cap)) return ADSB_ERR_INVALID;
    std::vector<uint64_t> cands, aps;
    if (int rc = selftest_pass(c, d_iq, n_samples, nullptr, nullptr, &cands, &aps)) return rc;
    return hand_out(cands, cand, cand_cap, n_cand, aps, ap, ap_cap, n_ap);
} ADSB_ABI_CATCH

int adsb_selftest_gate_stages(adsb_ctx *c, const void *d_iq, size_t n_samples, uint64_t *preamble, size_t preamble_cap,
                              size_t *n_preamble, uint64_t *snr, size_t snr_cap, size_t *n_snr)
try {
    if ((!preamble && preamble_cap) || (!snr && snr_cap)) return ADSB_ERR_INVALID;
    std::vector<uint64_t> pre, sn;
    if (int rc = selftest_pass(c, d_iq, n_samples, &pre, &sn, nullptr, nullptr)) return rc;
    return hand_out(pre, preamble, preamble_cap, n_preamble, sn, snr, snr_cap, n_snr);
} ADSB_ABI_CATCH

int adsb_selftest_set_order_polls(adsb_ctx *c, uint32_t polls)
{
    if (!c) return ADSB_ERR_INVALID;
    if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
    c->order_polls = polls;
    return ADSB_OK;
}

}  // extern "C"
