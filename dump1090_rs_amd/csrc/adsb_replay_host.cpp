// adsb_replay_host.cpp -- the host-only half of libadsb_hip.so: the ordered replay (src/demod_2400.rs:149-207 with
// src/mode_s/mod.rs:34-139 scoring against src/icao_filter.rs:11-97 and src/crc.rs:263-282), the address union of the
// sharded capture, and the ABI's host-only entry points (adsb_replay_records, adsb_format_raw, adsb_read_test_data,
// adsb_selftest_crc_table, adsb_strerror).  No HIP in this unit: besides the library build it is compiled by plain
// g++ with -fsanitize=address,undefined and fed every trial the CPU checker slices plus adversarial records
// (tests/test_host_sanitizers.py) -- the GPU pool offers no device sanitizer, the host side needs none.
#include <immintrin.h>

#include <thread>

#include "adsb_replay_host.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <iterator>

using namespace adsb;

namespace {

// Ordered replay (src/demod_2400.rs:149-207 with mode_s scoring): records sorted by
// (chunk, j, try_phase); per (chunk, j) the best trial by strictly-greater score
// starting from -2 wins and is emitted when its score is >= 0.
inline uint64_t replay_key(const TrialRecord &r)
{
    return (uint64_t)r.chunk << 32 | (uint64_t)(r.j_tp & 0xFFFFFFu) << 8 | (r.j_tp >> 24);
}

}  // namespace

namespace adsb {
namespace host {

// The order records are replayed in: false and `order` = their indices by (chunk, j, try_phase) when they are not in
// it already, true (and `order` untouched) when they are.  Insertion for a handful, LSD radix otherwise; stable.
bool replay_order(const TrialRecord *rec, size_t n, std::vector<uint32_t> &order_out)
{
    struct Ref {
        uint64_t key;
        uint32_t idx;
    };
    bool sorted = true;
    for (size_t i = 1; i < n && sorted; i++) sorted = replay_key(rec[i - 1]) <= replay_key(rec[i]);
    if (sorted) return true;
    std::vector<Ref> order(n);
    uint64_t all_or = 0;
    for (size_t i = 0; i < n; i++) {
        order[i] = {replay_key(rec[i]), (uint32_t)i};
        all_or |= order[i].key;
    }
    if (n <= 96) {
        // a pass of a buffer or two (its workgroups write their records as they find them): by insertion,
        // stable, nothing to allocate or to count
        for (size_t a = 1; a < n; a++) {
            const Ref r = order[a];
            size_t b = a;
            for (; b > 0 && order[b - 1].key > r.key; b--) order[b] = order[b - 1];
            order[b] = r;
        }
        all_or = 0;   // (sorted: the passes below all skip)
    }
    // LSD radix sort, 11 bits a pass, skipping digits no key uses (a device pass has
    // chunk < 2^19, j < 2^18, try_phase < 16: four passes); stable
    std::vector<Ref> tmp(all_or ? n : 0);
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
    order_out.resize(n);
    for (size_t i = 0; i < n; i++) order_out[i] = src[i].idx;
    return false;
}

// A copy of the records in replay order (what a shard's device thread hands the one replaying thread: the sort is
// the larger half of a replay's time and the shards' sorts run side by side); false: they are in order as they are.
bool sort_records(const TrialRecord *rec, size_t n, std::vector<TrialRecord> &sorted_out)
{
    std::vector<uint32_t> order;
    if (replay_order(rec, n, order)) return false;
    sorted_out.resize(n);
    for (size_t i = 0; i < n; i++) sorted_out[i] = rec[order[i]];
    return true;
}

namespace {

struct NoPosition {
    void operator()(const TrialRecord &) const {}
};

// demod_2400.rs:149-207 over records taken in replay order through `at`: per (chunk, j) the best of the trial phases,
// strict >, from -2; `before` is told every record just before it is scored (the position-aware view wants its place)
template <class Filter, class At, class Before>
void replay_in_order(Filter &filter, const Crc24 &crc, size_t n, uint64_t chunk_offset, std::vector<adsb_msg> &out, At at, Before before)
{
    size_t i = 0;
    while (i < n) {
        const uint64_t pos = replay_key(at(i)) >> 8;  // (chunk, j)
        const TrialRecord *best = nullptr;
        Score best_score{false, (int)ADSB_MODES_SHORT_MSG_BYTES, -2};
        for (; i < n; i++) {
            const TrialRecord &r = at(i);
            if ((replay_key(r) >> 8) != pos) break;
            before(r);
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

inline uint32_t record_residual(const Crc24 &crc, const TrialRecord &r)
{
    return (r.pad & 1) ? (uint32_t)(r.power >> 40) : crc.residual(r.msg, (r.msg[0] & 0x80) ? 14 : 7);
}

// (chunk_offset + chunk, j, try_phase) of a record, for comparing the end of one run with the start of the next
inline unsigned __int128 global_key(const TrialRecord &r, uint64_t chunk_offset)
{
    return (unsigned __int128)(chunk_offset + r.chunk) << 32 | (uint32_t)((r.j_tp & 0xFFFFFFu) << 8 | (r.j_tp >> 24));
}

// the filter as it was when the capture began + the first adders: what score_modes_message asks at `now`
struct FilterView {
    const IcaoFilter &filter;
    const ParallelReplay::FirstAdds &adds;
    ParallelReplay::Pos now = 0;
    bool test(uint32_t addr, uint32_t start) const { return filter.test(addr, start) || adds.get(addr) < now; }
    void add(uint32_t, uint32_t) {}   // (ParallelReplay::finish: in the order of the first adders)
};

}  // namespace

void replay_sorted(IcaoFilter &filter, const Crc24 &crc, const TrialRecord *rec, size_t n, uint64_t chunk_offset, std::vector<adsb_msg> &out)
{
    replay_in_order(filter, crc, n, chunk_offset, out, [&](size_t i) -> const TrialRecord & { return rec[i]; }, NoPosition{});
}

void replay(IcaoFilter &filter, const Crc24 &crc, const TrialRecord *rec, size_t n, uint64_t chunk_offset,
            std::vector<adsb_msg> &out, uint64_t *host_sorts)
{
    // order = (chunk, j, try_phase).  Large passes arrive in that order from the device; anything
    // else is put in order here -- the records stay where they are (they may sit in mapped host
    // memory), only 16-byte (key, index) pairs are sorted.
    std::vector<uint32_t> order;
    const bool sorted = replay_order(rec, n, order);
    if (!sorted && host_sorts) ++*host_sorts;
    if (sorted) replay_in_order(filter, crc, n, chunk_offset, out, [&](size_t i) -> const TrialRecord & { return rec[i]; }, NoPosition{});
    else replay_in_order(filter, crc, n, chunk_offset, out, [&](size_t i) -> const TrialRecord & { return rec[order[i]]; }, NoPosition{});
}

void ParallelReplay::FirstAdds::reset(uint32_t capacity_pow2)
{
    key.assign(capacity_pow2, 0u);
    pos.assign(capacity_pow2, ~(Pos)0);   // (what get() answers for an absent value)
    mask = capacity_pow2 - 1;
    used = 0;
}

void ParallelReplay::FirstAdds::put_min(uint32_t value, Pos p)
{
    if (2 * (used + 1) > mask + 1) {   // half full: twice the size
        FirstAdds bigger;
        bigger.reset(2 * (mask + 1));
        for (uint32_t i = 0; i <= mask; i++)
            if (key[i]) bigger.put_min(key[i] - 1, pos[i]);
        *this = std::move(bigger);
    }
    uint32_t h = (value * 2654435761u) >> 7 & mask;
    while (key[h] && key[h] != value + 1) h = (h + 1) & mask;
    if (!key[h]) {
        key[h] = value + 1;
        used++;
    }
    if (p < pos[h]) pos[h] = p;
}

bool ParallelReplay::FirstAdds::put_first(uint32_t value, Pos p)
{
    uint32_t h = (value * 2654435761u) >> 7 & mask;
    while (key[h] && key[h] != value + 1) h = (h + 1) & mask;
    if (key[h]) return false;
    put_min(value, p);   // (may grow the table)
    return true;
}

ParallelReplay::Pos ParallelReplay::FirstAdds::get(uint32_t value) const
{
    uint32_t h = (value * 2654435761u) >> 7 & mask;
    while (key[h] && key[h] != value + 1) h = (h + 1) & mask;
    return pos[h];   // (an empty slot holds ~0)
}

void first_adders(const Crc24 &crc, const TrialRecord *rec, size_t n, ParallelReplay::Adders &out)
{
    out.clear();
    ParallelReplay::FirstAdds seen;
    seen.reset(256);
    uint32_t last_value = ~0u;
    for (size_t q = 0; q < n; q++) {
        const TrialRecord &r = rec[q];
        const uint32_t df = r.msg[0] >> 3;
        if (df != 17 && df != 18 && df != 11) continue;
        if (record_residual(crc, r) != 0) continue;
        const uint32_t addr = uint32_t(r.msg[1]) << 16 | uint32_t(r.msg[2]) << 8 | r.msg[3];
        const uint32_t value = df == 18 ? (addr | IcaoFilter::kAdsbNt) : addr;
        if (value == last_value) continue;
        last_value = value;
        if (seen.put_first(value, q)) out.push_back({value, (uint64_t)q});
    }
}

bool ParallelReplay::plan(const IcaoFilter &filter, const Crc24 &crc, const std::vector<RecordRun> &runs, int parts, bool runs_in_order,
                          const std::vector<const Adders *> *run_adders)
{
    filter_ = &filter;
    crc_ = &crc;
    // (the parts keep their vectors' memory from capture to capture: tens of thousands of messages each time)
    for (Part &p : part_) {
        p.runs.clear();
        p.out.clear();
    }
    new_values_.clear();
    run_adders_.clear();
    size_t total = 0;
    unsigned __int128 last = 0;
    bool first = true;
    for (const RecordRun &r : runs) {
        total += r.n;
        if (!r.n) continue;
        // in replay order throughout: inside every run (unless the caller has seen to that), and from one run to the next
        for (size_t i = 1; i < r.n && !runs_in_order; i++)
            if (replay_key(r.rec[i - 1]) > replay_key(r.rec[i])) return false;
        if (!first && global_key(r.rec[0], r.chunk_offset) < last) return false;
        last = global_key(r.rec[r.n - 1], r.chunk_offset);
        first = false;
    }
    if (parts < 2 || total < (size_t)parts) return false;
    part_.resize((size_t)parts);
    n_records_ = total;
    if (run_adders && run_adders->size() == runs.size()) {
        bool all = true;
        for (size_t k = 0; k < runs.size(); k++) all = all && (runs[k].n == 0 || (*run_adders)[k] != nullptr);
        Pos first_number = 0;
        for (size_t k = 0; k < runs.size() && all; k++) {
            if (runs[k].n) run_adders_.push_back({(*run_adders)[k], first_number});
            first_number += runs[k].n;
        }
    }
    const size_t per = (total + (size_t)parts - 1) / (size_t)parts;
    size_t k = 0, in_part = 0;
    Pos number = 0;
    for (const RecordRun &r : runs) {
        size_t at = 0;
        while (at < r.n) {
            const size_t take = in_part < per ? std::min(r.n - at, per - in_part) : r.n - at;   // (the last part takes what is left)
            // a position's trial phases stay together: the cut moves to the next (chunk, j)
            size_t end = at + take;
            while (end < r.n && (replay_key(r.rec[end]) >> 8) == (replay_key(r.rec[end - 1]) >> 8)) end++;
            part_[k].runs.push_back({{r.rec + at, end - at, r.chunk_offset}, number});
            number += end - at;
            in_part += end - at;
            at = end;
            if (in_part >= per && k + 1 < part_.size()) {
                k++;
                in_part = 0;
            }
        }
    }
    return true;
}

void ParallelReplay::scan_part(int i)
{
    Part &p = part_[(size_t)i];
    // (worked on in locals and handed back at the end: the vectors' own size fields are written on every insertion)
    FirstAdds adds = std::move(p.adds);
    std::vector<std::pair<uint32_t, Pos>> found = std::move(p.found);
    adds.reset(256);
    found.clear();
    uint32_t last_value = ~0u;   // (a frame leaves a record per trial phase that decodes it: the same value several times running)
    for (const Piece &piece : p.runs)
        for (size_t q = 0; q < piece.run.n; q++) {
            const TrialRecord &r = piece.run.rec[q];
            const uint32_t df = r.msg[0] >> 3;
            if (df != 17 && df != 18 && df != 11) continue;
            if (record_residual(*crc_, r) != 0) continue;   // mod.rs:80-84 (IID 0: the whole residual is zero), :97-99
            const uint32_t addr = uint32_t(r.msg[1]) << 16 | uint32_t(r.msg[2]) << 8 | r.msg[3];
            const uint32_t value = df == 18 ? (addr | IcaoFilter::kAdsbNt) : addr;
            if (value == last_value) continue;
            last_value = value;
            if (adds.put_first(value, piece.first + q)) found.push_back({value, piece.first + q});   // (in order: the first seen is the part's first adder)
        }
    p.adds = std::move(adds);
    p.found = std::move(found);
}

bool ParallelReplay::merge()
{
    uint32_t total = 0;
    if (scan_needed()) {
        for (const Part &p : part_) total = std::max(total, (uint32_t)p.found.size());   // (the parts mostly find the same values)
    } else {
        for (const auto &ra : run_adders_) total = std::max(total, (uint32_t)ra.first->size());
    }
    uint32_t cap = 256;
    while (cap < 4 * total) cap <<= 1;
    all_.reset(cap);   // (grows by itself)
    if (scan_needed()) {
        for (const Part &p : part_)
            for (const auto &f : p.found) all_.put_min(f.first, f.second);
    } else {
        for (const auto &ra : run_adders_)
            for (const auto &f : *ra.first) all_.put_min(f.first, ra.second + f.second);
    }
    // what will newly enter the table, and when
    size_t held = 0;
    std::vector<uint32_t> tagged;   // the DF18 values the table holds already
    for (uint32_t v : filter_->table()) {
        held += v != 0;
        if (v & IcaoFilter::kAdsbNt) tagged.push_back(v);
    }
    new_values_.clear();
    for (uint32_t i = 0; i <= all_.mask; i++) {
        if (!all_.key[i]) continue;
        const uint32_t value = all_.key[i] - 1;
        const Pos at = all_.pos[i];
        if (value & IcaoFilter::kAdsbNt) {
            // DF18: tests the plain address, adds the tagged value -- at its first record that finds the address unknown;
            // knowledge only grows, so that is its first record or none
            const uint32_t addr = value & 0xFFFFFFu;
            if (filter_->test(addr) || all_.get(addr) < at) continue;
            if (std::find(tagged.begin(), tagged.end(), value) == tagged.end()) new_values_.push_back({at, value});
        } else if (!filter_->test(value)) {
            new_values_.push_back({at, value});
        }
    }
    std::sort(new_values_.begin(), new_values_.end());
    // add() gives up on a full table and test() then walks all of it: membership stops being a set's
    return held + new_values_.size() + 64 < IcaoFilter::kSize;
}

void ParallelReplay::score_part(int i)
{
    Part &p = part_[(size_t)i];
    std::vector<adsb_msg> out = std::move(p.out);   // (see scan_part)
    out.clear();
    size_t mine = 0;
    for (const Piece &piece : p.runs) mine += piece.run.n;
    out.reserve(mine / 2);   // (a frame leaves ~3 records)
    FilterView view{*filter_, all_};
    for (const Piece &piece : p.runs) {
        const TrialRecord *rec = piece.run.rec;
        replay_in_order(view, *crc_, piece.run.n, piece.run.chunk_offset, out, [&](size_t q) -> const TrialRecord & { return rec[q]; },
                        [&](const TrialRecord &r) { view.now = piece.first + (Pos)(&r - rec); });
    }
    p.out = std::move(out);
}

size_t ParallelReplay::message_count() const
{
    size_t n = 0;
    for (const Part &p : part_) n += p.out.size();
    return n;
}

void ParallelReplay::copy_to(adsb_msg *dst)
{
    dst_ = dst;
    size_t at = 0;
    for (Part &p : part_) {
        p.out_at = at;
        at += p.out.size();
    }
}

void ParallelReplay::copy_part(int i)
{
    const Part &p = part_[(size_t)i];
    static_assert(sizeof(adsb_msg) % 8 == 0 && alignof(adsb_msg) >= 8, "messages are copied as 8-byte words");
    // (not through the cache: the destination's lines are in the reader's, and taking them over one by one is what a
    // plain copy from here would spend its time on)
    const long long *src = reinterpret_cast<const long long *>(p.out.data());
    long long *dst = reinterpret_cast<long long *>(dst_ + p.out_at);
    const size_t words = p.out.size() * (sizeof(adsb_msg) / 8);
    for (size_t w = 0; w < words; w++) _mm_stream_si64(dst + w, src[w]);
    _mm_sfence();
}

void ParallelReplay::apply_adds(IcaoFilter &filter) const
{
    for (const auto &nv : new_values_) filter.add(nv.second, IcaoFilter::hash(nv.second & 0xFFFFFFu));
}

void ParallelReplay::finish(IcaoFilter &filter, std::vector<adsb_msg> &out)
{
    size_t total = 0;
    for (const Part &p : part_) total += p.out.size();
    out.reserve(out.size() + total);
    for (const Part &p : part_) out.insert(out.end(), p.out.begin(), p.out.end());
    apply_adds(filter);
}

// mode_s/mod.rs:80-84 (DF11, IID 0) and :97-99 (DF17): the addresses the replay will add
void learned_addresses(const Crc24 &crc, const TrialRecord *rec, size_t n, std::vector<uint32_t> &addrs)
{
    for (size_t i = 0; i < n; i++) {
        const uint8_t *m = rec[i].msg;
        const uint32_t df = m[0] >> 3;
        const bool adds = df == 17 || (df == 11 && crc.residual(m, 7) == 0);
        if (adds) addrs.push_back(uint32_t(m[1]) << 16 | uint32_t(m[2]) << 8 | m[3]);
    }
}

void union_sorted(const std::vector<const std::vector<uint32_t> *> &lists, const std::vector<uint32_t> &known,
                  std::vector<uint32_t> &out)
{
    out.clear();
    for (const auto *l : lists)
        if (l) out.insert(out.end(), l->begin(), l->end());
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    if (!known.empty()) {
        std::vector<uint32_t> fresh;
        std::set_difference(out.begin(), out.end(), known.begin(), known.end(), std::back_inserter(fresh));
        out.swap(fresh);
    }
}

}  // namespace host
}  // namespace adsb

using namespace adsb;
using namespace adsb::host;

extern "C" {

static_assert(sizeof(adsb_trial) == sizeof(TrialRecord), "adsb_trial mirrors TrialRecord");

int adsb_replay_records(uint32_t *filter_table, adsb_trial *records, size_t n, adsb_msg *out,
                        size_t cap, size_t *n_out)
try {
    if (!filter_table || (!records && n) || (!out && cap)) return ADSB_ERR_INVALID;
    static const Crc24 crc;
    IcaoFilter filter;
    filter.load(filter_table);
    std::vector<adsb_msg> msgs;
    replay(filter, crc, reinterpret_cast<const TrialRecord *>(records), n, 0, msgs);
    filter.store(filter_table);
    const size_t k = std::min(cap, msgs.size());
    if (k) std::memcpy(out, msgs.data(), k * sizeof(adsb_msg));
    if (n_out) *n_out = msgs.size();
    return msgs.size() > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
} ADSB_ABI_CATCH

int adsb_read_test_data(const char *path, int16_t *iq, size_t max_samples, size_t *n_out)
try {
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
} ADSB_ABI_CATCH

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

int adsb_selftest_learned_union(const adsb_trial *records, size_t n, const uint32_t *known, size_t n_known, uint32_t *out,
                                 size_t cap, size_t *n_out)
try {
    if ((!records && n) || (!known && n_known) || (!out && cap)) return ADSB_ERR_INVALID;
    static const Crc24 crc;
    std::vector<uint32_t> learned, kn(known, known + n_known), fresh;
    learned_addresses(crc, reinterpret_cast<const TrialRecord *>(records), n, learned);
    std::sort(learned.begin(), learned.end());
    learned.erase(std::unique(learned.begin(), learned.end()), learned.end());
    std::sort(kn.begin(), kn.end());
    kn.erase(std::unique(kn.begin(), kn.end()), kn.end());
    union_sorted({&learned}, kn, fresh);
    if (n_out) *n_out = fresh.size();
    const size_t k = std::min(cap, fresh.size());
    if (k) std::memcpy(out, fresh.data(), k * sizeof(uint32_t));
    return fresh.size() > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
} ADSB_ABI_CATCH

int adsb_selftest_parallel_replay(uint32_t *filter_table, const adsb_trial *records, size_t n, int runs, int parts, int threads,
                                  adsb_msg *out, size_t cap, size_t *n_out, int *went_parallel)
try {
    if (!filter_table || (!records && n) || (!out && cap) || runs < 1 || parts < 1 || threads < 1) return ADSB_ERR_INVALID;
    static const Crc24 crc;
    IcaoFilter filter;
    filter.load(filter_table);
    // the records in replay order, cut into `runs` runs at changes of the buffer index (what shards are), each handed
    // over with the buffers before it taken out of its `chunk` and put into its chunk_offset
    std::vector<TrialRecord> sorted;
    const TrialRecord *rec = reinterpret_cast<const TrialRecord *>(records);
    if (sort_records(rec, n, sorted)) rec = sorted.data();
    else sorted.assign(rec, rec + n), rec = sorted.data();
    std::vector<RecordRun> rr;
    size_t at = 0;
    for (int k = 0; k < runs && at < n; k++) {
        size_t end = k + 1 == runs ? n : std::min(n, at + (n + (size_t)runs - 1) / (size_t)runs);
        while (end < n && end > at && sorted[end].chunk == sorted[end - 1].chunk) end++;
        const uint32_t base = sorted[at].chunk;
        for (size_t i = at; i < end; i++) sorted[i].chunk -= base;
        rr.push_back({rec + at, end - at, base});
        at = end;
    }
    std::vector<adsb_msg> msgs;
    ParallelReplay pr;
    // (every other size: the runs bring their first adders along, as a device thread of adsb_multi hands them over)
    std::vector<ParallelReplay::Adders> adders(rr.size());
    std::vector<const ParallelReplay::Adders *> adders_of;
    if (n & 2)
        for (size_t k = 0; k < rr.size(); k++) {
            first_adders(crc, rr[k].rec, rr[k].n, adders[k]);
            adders_of.push_back(&adders[k]);
        }
    bool parallel = pr.plan(filter, crc, rr, parts, false, adders_of.empty() ? nullptr : &adders_of);
    if (parallel && (n & 2) && pr.scan_needed()) return ADSB_ERR_INVALID;   // (the lists were complete: they must have been taken)
    if (parallel) {
        auto fan_out = [&](void (ParallelReplay::*stage)(int)) {
            std::vector<std::thread> th;
            for (int t = 0; t < threads; t++)
                th.emplace_back([&, t] {
                    for (int i = t; i < pr.parts(); i += threads) (pr.*stage)(i);
                });
            for (auto &x : th) x.join();
        };
        if (pr.scan_needed()) fan_out(&ParallelReplay::scan_part);
        parallel = pr.merge();
        if (parallel) {
            fan_out(&ParallelReplay::score_part);
            if (pr.message_count() <= cap && (n & 1)) {   // (either way out: straight into the caller's array, or through a list)
                pr.copy_to(out);
                fan_out(&ParallelReplay::copy_part);
                pr.apply_adds(filter);
                if (went_parallel) *went_parallel = 1;
                if (n_out) *n_out = pr.message_count();
                filter.store(filter_table);
                return ADSB_OK;
            }
            pr.finish(filter, msgs);
        }
    }
    if (!parallel)
        for (const RecordRun &r : rr) replay_sorted(filter, crc, r.rec, r.n, r.chunk_offset, msgs);
    if (went_parallel) *went_parallel = parallel ? 1 : 0;
    filter.store(filter_table);
    const size_t k = std::min(cap, msgs.size());
    if (k) std::memcpy(out, msgs.data(), k * sizeof(adsb_msg));
    if (n_out) *n_out = msgs.size();
    return msgs.size() > cap ? ADSB_ERR_CAPACITY : ADSB_OK;
} ADSB_ABI_CATCH

int adsb_selftest_crc_table(uint32_t *out256)
{
    if (!out256) return ADSB_ERR_INVALID;
    static const Crc24 crc;  // the table the host replay scores with (mode_s_host.hpp)
    std::memcpy(out256, crc.t, sizeof(crc.t));
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
    case ADSB_ERR_POISONED: return "an earlier capture of this adsb_multi failed: adsb_multi_icao_flush starts the stream over";
    default: return "unknown status";
    }
}

}  // extern "C"
