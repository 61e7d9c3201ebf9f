// adsb_context.cpp -- context lifetime, the stream pool, settings and diagnostics (include/adsb_hip.h).
//
// The functions of the library mirror the reference's library API for the path (src/utils.rs:43 to_mag,
// src/demod_2400.rs:115 demodulate2400, src/icao_filter.rs:11 icao_flush); what each one replaces is
// listed in the header.
#include "adsb_ctx.h"

using namespace adsb::host;

extern "C" {

// Streams.  The runtime multiplexes streams onto hardware queues: GPU_MAX_HW_QUEUES (4 by default) per priority, gives
// every new stream of a priority a new queue until it has four of that priority, never gives one back, and streams
// beyond that SHARE queues, i.e. run one after the other.  Measured in earlier rounds: a process that creates, destroys
// and re-creates contexts ends up with its streams on a different set of queues each time (a dense stream in the second
// context alternated 88 / 270 us per pass); a process with a large context (two scan streams) AND contexts for passes of
// a few buffers (four) held six high-priority streams, two of which shared a queue -- the one-buffer ring ran at 6.8
// instead of 8.8 Gsample/s, or the large stream at 0.129 instead of 0.097 ms per step, depending on who came first --
// unless the process set GPU_MAX_HW_QUEUES=8 before the runtime started.
// So the library never holds more than four streams of a priority per device, whatever its users create
// (profiles/r5_stream_pool_ab.txt).  Per device, for the life of the process:
//   * TWO highest-priority streams that every large context's scans alternate between.  Exactly two: with four in
//     existence -- even unused -- the large sparse stream's step went from 0.096 to 0.125 ms (consecutive scans no
//     longer overlapped);
//   * FOUR normal-priority streams, made when the first context for passes of a few buffers is, that every such
//     context's one-launch passes rotate over: 10.3-10.6 Gsample/s for the one-buffer ring in a process of its own
//     (four high-priority ones of its own, round 4: 8.7 on the same box; the two shared high-priority ones alone: 8.9;
//     two high + two normal: 6.3 -- mixed priorities finish passes out of order);
//   * TWO lowest-priority ones for the contexts' tail and score chains.
// Contexts that share a stream are simply several in-order users of it (every wait is for an event recorded before it
// was asked for: no cycle).  No environment variable is needed by anybody.
struct DeviceStreams {
    int device = -1;
    int least = 0, greatest = 0;
    hipStream_t high[2] = {};              // highest priority: the scans of large contexts
    hipStream_t small[kScanStreams] = {};  // the four scan streams of contexts for passes of a few buffers
    hipStream_t low[4] = {};               // lowest priority: tail and score chains
    int n_low = 0;
    unsigned next_low = 0;
    std::vector<hipStream_t> free_own;     // normal priority: a context's own stream (input order, host-pointer copies)
};
std::mutex g_streams_mu;
std::vector<DeviceStreams *> g_device_streams;   // (never freed: the streams live as long as the process)

// A stream with a hardware queue of its own.  The runtime maps the streams of a process onto at most four hardware queues
// per priority (GPU_MAX_HW_QUEUES): the fifth stream of a priority shares the queue of an earlier one -- whichever has the
// fewest users at that moment -- and two streams on one queue run their work in order.  Which of the library's streams
// ended up sharing therefore depended on how many streams of that priority the HOST had made before (torch's null stream,
// the contexts' own streams): the four scan streams of a context for passes of a few buffers came out as two, three or
// four queues, and the one-buffer ring ran at 9.0, 10.8 or 11.5 Gsample/s (profiles/r6_stream_queues.txt).  A stream
// created with a CU mask never enters that pool -- the runtime gives it a queue of its own -- and a mask with every CU
// enabled restricts nothing.
hipError_t dedicated_stream(int device, hipStream_t *out)
{
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return e;
    std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, 0xFFFFFFFFu);
    if (prop.multiProcessorCount % 32) mask.back() = (1u << (prop.multiProcessorCount % 32)) - 1u;
    return hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data());
}

// under g_streams_mu, the device current
int device_streams(adsb_ctx *c, int device, bool small, DeviceStreams **out)
{
    DeviceStreams *d = nullptr;
    for (DeviceStreams *e : g_device_streams)
        if (e->device == device) d = e;
    if (!d) {
        d = new (std::nothrow) DeviceStreams;
        if (!d) return ADSB_ERR_NOMEM;
        d->device = device;
        HIP_TRY(c, hipDeviceGetStreamPriorityRange(&d->least, &d->greatest));
        g_device_streams.push_back(d);
    }
    // The two scan streams take the highest priority: that pool holds nothing else of this process (the null stream,
    // torch's and the caller's streams are of normal priority), so each gets a queue and consecutive scans overlap.
    // They must have the SAME priority: with different ones, whenever two scans are pending at once the higher one
    // starts first, its successor on that stream is then free earlier too, and the stream settles into finishing
    // passes in the order 2, 1, 4, 3, ... for thousands of passes, 8-10 % slower (measured over 22 000 passes).
    // (measurement aid, tuning builds: ADSB_POOL_LARGE=1 the two scan streams of large contexts with a hardware queue each
    // instead of the highest priority, =2 the tail and score streams as well)
    const int large_variant = tuning_env("ADSB_POOL_LARGE") ? std::atoi(tuning_env("ADSB_POOL_LARGE")) : 0;
    if (!d->high[0])
        for (auto &q : d->high) {
            if (large_variant >= 1) HIP_TRY(c, dedicated_stream(device, &q));
            else HIP_TRY(c, hipStreamCreateWithPriority(&q, hipStreamNonBlocking, d->greatest));
        }
    if (!d->n_low) {
        const int n = 2;
        for (int k = 0; k < 4; k++) {
            if (k >= n) d->low[k] = d->low[k % n];
            else if (large_variant >= 2) HIP_TRY(c, dedicated_stream(device, &d->low[k]));
            else HIP_TRY(c, hipStreamCreateWithPriority(&d->low[k], hipStreamNonBlocking, d->least));
        }
        d->n_low = n;
    }
    if (small && !d->small[0]) {
        // (measurement aid, tuning builds: 0 the two high-priority streams + two more, 1 those two + two of normal
        // priority, 2 four of normal priority, 3 four high-priority ones of its own, 4 four of the lowest priority,
        // 5 four streams with a hardware queue each: hipExtStreamCreateWithCUMask, every CU enabled)
        const int variant = tuning_env("ADSB_POOL_SMALL") ? std::atoi(tuning_env("ADSB_POOL_SMALL")) : 5;
        const int mid = (d->least + d->greatest) / 2;
        for (int k = 0; k < kScanStreams; k++) {
            const bool share = (variant == 0 || variant == 1) && k < 2;
            const int prio = variant == 0 || variant == 3 ? d->greatest : (variant == 4 ? d->least : mid);
            if (share) d->small[k] = d->high[k];
            else if (variant == 5) {
                // (a runtime that refuses the mask: the pool stream this used to be -- slower when the host's own streams
                // crowd the pool, never wrong)
                if (dedicated_stream(device, &d->small[k]) != hipSuccess) {
                    (void)hipGetLastError();
                    HIP_TRY(c, hipStreamCreateWithPriority(&d->small[k], hipStreamNonBlocking, mid));
                }
            }
            else HIP_TRY(c, hipStreamCreateWithPriority(&d->small[k], hipStreamNonBlocking, prio));
        }
    }
    *out = d;
    return ADSB_OK;
}

int adsb_create(adsb_ctx **out, int device, size_t max_chunks)
try {
    if (!out) return ADSB_ERR_INVALID;
    *out = nullptr;
    if (device < 0) return ADSB_ERR_NO_DEVICE;  // no CPU backend by design
    if (max_chunks == 0) max_chunks = 1;
    if (max_chunks > kMaxChunks) return ADSB_ERR_INVALID;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device >= count) return ADSB_ERR_NO_DEVICE;
    {
        // the kernels are gfx950 code objects and nothing else: any other device is "no usable device" here, not a
        // failed launch later (include/adsb_hip.h: ADSB_ERR_NO_DEVICE)
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return ADSB_ERR_NO_DEVICE;
    }

    adsb_ctx *c = new (std::nothrow) adsb_ctx;
    if (!c) return ADSB_ERR_NOMEM;
    c->device = device;
    c->max_chunks = max_chunks;
    // (a context for passes of a few buffers: one launch each, eight in flight -- adsb_ctx.h)
    c->n_slots = max_chunks <= kInlineTailChunks ? ADSB_MAX_IN_FLIGHT_SMALL : ADSB_MAX_IN_FLIGHT;
    c->n_bitmaps = c->n_slots + 1;
    c->n_scan_streams = c->n_slots == ADSB_MAX_IN_FLIGHT_SMALL ? kScanStreams : 2;
    c->bitmap_lg = c->n_slots == ADSB_MAX_IN_FLIGHT_SMALL ? kSmallBitmapLg : kFullBitmapLg;
    if (const char *ds = tuning_env("ADSB_DEBUG_STOP")) c->debug_stop = std::atoi(ds);
    if (const char *st = tuning_env("ADSB_STAGGER")) c->stagger_ticks = (uint32_t)std::atoi(st);
    // The fast scan's AP list: one private segment per wave of every persistent workgroup (a pass
    // of n buffers runs min(17 n, resident grid) workgroups of four waves, so a small context only gets
    // the segments it can ever use), each sized for ~5x the rate pure noise produces (2.3 % of
    // samples become address/parity entries).  Denser input falls back to buffer-by-buffer
    // passes through the reference-shaped kernel, whose list (dap) and the hit list hold one
    // buffer's worst case: every position sliced, five trials each.
    DeviceGuard on_device(device);  // (scan_resident_blocks() asks the current device; the caller's is put back on the way out)
    if (on_device.err != hipSuccess) {
        delete c;
        return ADSB_ERR_NO_DEVICE;
    }
    const uint64_t used_segs = 4 * std::min<uint64_t>((uint64_t)scan_resident_blocks(), max_chunks * (uint64_t)fastgeo::kTilesPerChunk);
    c->seg_cap = (uint32_t)std::max<uint64_t>(1024, (max_chunks * (uint64_t)kChunkSamples / 8 + used_segs - 1) / used_segs);
    c->ap_cap = (uint32_t)(used_segs * c->seg_cap);
    // hit list: ~5x what a busy airspace produces (a frame leaves 3-4 trial records; 1000 frames/s
    // are ~55 per buffer); more than that is the fallback's business too
    c->hits_cap = (uint32_t)(4096 + max_chunks * 1024);

    int rc = ADSB_OK;
    auto body = [&]() -> int {
        HIP_TRY(c, hipSetDevice(device));
        if (tuning_env("ADSB_STREAM_PRIO") || tuning_env("ADSB_SCORE_PRIO")) {
            // measurement aid (tuning builds): streams of this context's own, "tail,scan0,scan1" as 0 (least) .. 2
            int least = 0, greatest = 0;
            HIP_TRY(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
            int pt = least, p0 = greatest, p1 = greatest, ps = least;
            if (const char *e = tuning_env("ADSB_STREAM_PRIO")) {
                int a = 0, b = 1, d = 2;
                if (std::sscanf(e, "%d,%d,%d", &a, &b, &d) == 3) {
                    const int lv[3] = {least, (least + greatest) / 2, greatest};
                    pt = lv[a % 3], p0 = lv[b % 3], p1 = lv[d % 3];
                }
            }
            if (const char *e = tuning_env("ADSB_SCORE_PRIO")) ps = std::atoi(e) == 2 ? greatest : (std::atoi(e) == 1 ? (least + greatest) / 2 : least);
            c->private_streams = true;
            HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
            HIP_TRY(c, hipStreamCreateWithPriority(&c->tail_stream, hipStreamNonBlocking, pt));
            for (int k = 0; k < c->n_scan_streams; k++)
                HIP_TRY(c, hipStreamCreateWithPriority(&c->scan_stream[k], hipStreamNonBlocking, k & 1 ? p1 : p0));
            HIP_TRY(c, hipStreamCreateWithPriority(&c->score_stream, hipStreamNonBlocking, ps));
        } else {
            std::lock_guard<std::mutex> lk(g_streams_mu);
            DeviceStreams *ds = nullptr;
            const bool small = c->n_scan_streams == kScanStreams;
            if (int rc2 = device_streams(c, device, small, &ds)) return rc2;
            for (int k = 0; k < c->n_scan_streams; k++) c->scan_stream[k] = small ? ds->small[k] : ds->high[k];
            c->tail_stream = ds->low[ds->next_low++ & 3u];
            c->score_stream = ds->low[ds->next_low++ & 3u];
            if (!ds->free_own.empty()) {
                c->own_stream = ds->free_own.back();
                ds->free_own.pop_back();
            } else {
                HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
            }
        }
        c->stream = c->own_stream;
        for (auto &e : c->input_ready)
            HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
        HIP_TRY(c, hipEventCreateWithFlags(&c->lazy_ev, hipEventDisableTiming | hipEventDisableSystemFence));
        for (int k = 0; k < c->n_bitmaps; k++) HIP_TRY(c, hipMalloc((void **)&c->d_bitmap[k], bitmap_alloc_words(c->bitmap_lg) * sizeof(uint32_t)));
        for (int si = 0; si < c->n_slots; si++) {
            Slot &sl = c->slot[si];
            HIP_TRY(c, hipMalloc((void **)&sl.d_ctr, sizeof(Counters)));
            sl.hits_cap = c->hits_cap;
            HIP_TRY(c, hipMalloc((void **)&sl.d_hits, (size_t)c->hits_cap * sizeof(uint64_t)));
            HIP_TRY(c, hipMalloc((void **)&sl.d_hit_fields, (size_t)c->hits_cap * kHitFieldWords * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc((void **)&sl.d_ap, (size_t)c->ap_cap * sizeof(uint64_t)));
            HIP_TRY(c, hipMalloc((void **)&sl.d_order_cnt, (max_chunks * fastgeo::kTilesPerChunk + 1) * sizeof(uint32_t)));
            HIP_TRY(c, hipMemset(sl.d_order_cnt, 0, (max_chunks * fastgeo::kTilesPerChunk + 1) * sizeof(uint32_t)));
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
            // (a context for passes of a few buffers never orders or scores on the device -- its passes are one
            // launch each, replayed by the host in microseconds -- and carries none of this: no exact bitmaps, no
            // scoring buffers, no message slots in its pinned block)
            const bool scoring = c->n_slots != ADSB_MAX_IN_FLIGHT_SMALL;
            cd.cap = scoring ? std::min<uint32_t>(c->hits_cap, 262144u) : 0u;   // (a 2 GiB shard of a busy sky leaves 135 000)
            uint32_t hsize = 1;
            while (hsize < 2 * cd.cap) hsize <<= 1;
            cd.hash_mask = hsize - 1;
            if (scoring) {
                for (auto &bm : c->exact_bm) {
                    HIP_TRY(c, hipMalloc((void **)&bm, kBitmapAllocWords * sizeof(uint32_t)));
                    HIP_TRY(c, hipMemset(bm, 0, kBitmapAllocWords * sizeof(uint32_t)));
                }
                cd.exact = c->exact_bm[0];
                cd.si = reinterpret_cast<uint32_t *>(cd.exact);  // (non-null: "scoring is available")
            }
            for (int si = 0; si < c->n_slots; si++) {
                Slot &sl = c->slot[si];
                ScoreDev &sd = sl.score;
                sd = cd;
                HIP_TRY(c, hipEventCreateWithFlags(&sl.recorded, hipEventDisableTiming | hipEventDisableSystemFence));
                if (!scoring) continue;
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
            }
            // The host side of every slot -- summary, score summary, additions, records, messages: mapped, coherent,
            // written by the kernels with write-through stores -- as ONE pinned allocation per context, cut up here.
            // (five separate allocations per slot were forty small pinned regions in a one-buffer context)
            const ScoreDev &sd = cd;
            auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
            const size_t o_ssum = up(sizeof(Summary)), o_adds = o_ssum + up(sizeof(ScoreSummary)),
                         o_rec = o_adds + up((size_t)sd.cap * sizeof(uint32_t)),
                         o_msgs = o_rec + up((size_t)c->hits_cap * sizeof(TrialRecord)),
                         per_slot = o_msgs + up((size_t)sd.cap * sizeof(adsb_msg));
            HIP_TRY(c, hipHostMalloc((void **)&c->h_block, per_slot * (size_t)c->n_slots, hipHostMallocMapped | hipHostMallocCoherent));
            HIP_TRY(c, hipHostGetDevicePointer((void **)&c->h_block_dev, c->h_block, 0));
            for (int si = 0; si < c->n_slots; si++) {
                Slot &sl = c->slot[si];
                char *h = c->h_block + per_slot * (size_t)si, *d = c->h_block_dev + per_slot * (size_t)si;
                sl.h_sum = reinterpret_cast<Summary *>(h), sl.h_sum_dev = reinterpret_cast<Summary *>(d);
                sl.h_ssum = reinterpret_cast<ScoreSummary *>(h + o_ssum), sl.h_ssum_dev = reinterpret_cast<ScoreSummary *>(d + o_ssum);
                sl.h_adds = reinterpret_cast<uint32_t *>(h + o_adds), sl.h_adds_dev = reinterpret_cast<uint32_t *>(d + o_adds);
                sl.h_rec = reinterpret_cast<TrialRecord *>(h + o_rec), sl.h_rec_dev = reinterpret_cast<TrialRecord *>(d + o_rec);
                sl.h_msgs = reinterpret_cast<adsb_msg *>(h + o_msgs), sl.h_msgs_dev = reinterpret_cast<adsb_msg *>(d + o_msgs);
                std::memset(sl.h_sum, 0, sizeof(Summary));
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
        for (int si = 0; si < c->n_slots; si++) {
            Slot &sl = c->slot[si];
            // (the slot's summary and records live in the context's one pinned block: mapped + coherent, so the records
            // kernel's write-through stores are visible to the host when its completion event fires)
            // timing-only events: no system-scope fence when they complete (~10 us each otherwise)
            for (int k = 2; k < 5; k++) HIP_TRY(c, hipEventCreateWithFlags(&sl.ev[k], hipEventDisableSystemFence));
            const bool fenced = tuning_env("ADSB_DONE_FENCE") != nullptr;  // measurement aid only
            HIP_TRY(c, hipEventCreateWithFlags(&sl.done, fenced ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence)));
        }
        for (auto &pair : c->scan_ev)
            for (auto &e : pair) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
        for (auto &e : c->redo_ev) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
        if (tuning_env("ADSB_TIMELINE")) {
            // 1: stamps of 8 blocks x 8 tiles; 2 (with ADSB_DEBUG_STOP=100): per-wave phase totals
            HIP_TRY(c, hipMalloc((void **)&c->d_timeline, kTimelineWords * sizeof(unsigned long long)));
            HIP_TRY(c, hipMemset(c->d_timeline, 0, kTimelineWords * sizeof(unsigned long long)));
        }
        // both bitmaps clean and both counter blocks zero to start with; from then on each
        // pass cleans up for the next (the first pass needs no flush of its own)
        for (int k = 0; k < c->n_bitmaps; k++)
            if (int e = launch_reset(c->slot[k % (uint64_t)c->n_slots].d_ctr, c->d_bitmap[k], c->bitmap_lg, c->stream))
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
} ADSB_ABI_CATCH

void adsb_destroy(adsb_ctx *c)
{
    if (!c) return;
#ifdef ADSB_TUNING
    if (tuning_env("ADSB_HOST_TIMES"))
        std::fprintf(stderr, "host times over %llu passes: enqueue %.1f us, wait %.1f us, replay %.1f us per pass\n",
                     (unsigned long long)c->collected, 1e6 * c->t_enqueue / (c->collected ? c->collected : 1),
                     1e6 * c->t_wait / (c->collected ? c->collected : 1), 1e6 * c->t_replay / (c->collected ? c->collected : 1));
    if (tuning_env("ADSB_HOST_TIMES")) {
        static const char *name[HT_COUNT] = {"ring: hipMemcpyAsync", "ring: (unused)", "input-ready record + wait",
                                             "scan launch", "scanned record + waits", "match (+ order) launch", "records launch",
                                             "recorded / done records", "collect: wait for the pass", "collect: record checksum",
                                             "collect: replay"};
        double all = 0;
        for (int k = 0; k < HT_COUNT; k++) all += c->ht_s[k];
        for (int k = 0; k < HT_COUNT; k++)
            if (c->ht_n[k])
                std::fprintf(stderr, "  %-32s %8.2f us per pass  (%llu calls, %.2f us each)  %5.1f %%\n", name[k],
                             1e6 * c->ht_s[k] / (c->collected ? c->collected : 1), (unsigned long long)c->ht_n[k],
                             1e6 * c->ht_s[k] / c->ht_n[k], 100.0 * c->ht_s[k] / (all > 0 ? all : 1));
    }
#endif
    DeviceGuard on_device(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (auto &pair : c->scan_ev)
        for (auto &e : pair)
            if (e) (void)hipEventDestroy(e);
    for (auto &e : c->redo_ev)
        if (e) (void)hipEventDestroy(e);
    for (Slot &sl : c->slot) {
        for (int k = 2; k < 5; k++)
            if (sl.ev[k]) (void)hipEventDestroy(sl.ev[k]);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.scanned) (void)hipEventDestroy(sl.scanned);
        if (sl.d_ctr) (void)hipFree(sl.d_ctr);
        if (sl.d_hits) (void)hipFree(sl.d_hits);
        if (sl.d_hit_fields) (void)hipFree(sl.d_hit_fields);
        if (sl.d_ap) (void)hipFree(sl.d_ap);
        if (sl.d_order_cnt) (void)hipFree(sl.d_order_cnt);
        if (sl.d_order_base) (void)hipFree(sl.d_order_base);
        if (sl.d_order_tmp) (void)hipFree(sl.d_order_tmp);
        if (sl.d_carry) (void)hipFree(sl.d_carry);
    }
    if (c->d_stage) (void)hipFree(c->d_stage);
    for (auto &b : c->d_bitmap)
        if (b) (void)hipFree(b);
    for (hipStream_t q : c->scan_stream)
        if (q) (void)hipStreamSynchronize(q);
    for (hipEvent_t e : c->input_ready)
        if (e) (void)hipEventDestroy(e);
    if (c->lazy_ev) (void)hipEventDestroy(c->lazy_ev);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_block) (void)hipHostFree(c->h_block);
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
    for (const auto &r : c->host_ranges) (void)hipHostUnregister(r.base);
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->ring_h_block) (void)hipHostFree(c->ring_h_block);
    if (c->ring_d_block) (void)hipFree(c->ring_d_block);
    if (c->d_addrs) (void)hipFree(c->d_addrs);
    for (auto &j : c->shard) {
        if (j.h_addrs) (void)hipHostFree(j.h_addrs);
        if (j.h_fresh) (void)hipHostFree(j.h_fresh);
        if (j.d_fresh_seen) (void)hipFree(j.d_fresh_seen);
        if (j.h_earlier) (void)hipHostFree(j.h_earlier);
        for (hipEvent_t e : j.addr_read)
            if (e) (void)hipEventDestroy(e);
    }
    if (c->d_carry_next) (void)hipFree(c->d_carry_next);
    if (c->d_timeline && tuning_env("ADSB_TIMELINE") && std::atoi(tuning_env("ADSB_TIMELINE")) == 3) {
        // profiling aid: the stamps of the last one-launch pass (100 MHz wall clock)
        unsigned long long t[10] = {};
        if (hipMemcpy(t, c->d_timeline + 448, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess && t[0]) {
            if (t[8] > t[4] && t[7] > t[8])
                std::fprintf(stderr, "  one-launch pass, second look: fill counts in LDS +%.2f us, entries compared +%.2f us, fence +%.2f us\n",
                             (double)(long long)(t[8] - t[4]) / 100.0, (double)(long long)(t[7] - t[8]) / 100.0,
                             (double)(long long)(t[5] - t[7]) / 100.0);
            static const char *name[7] = {"entry", "tables in LDS, first tile requested", "tiles done", "own entries matched",
                                          "counted in (last workgroup from here on)", "second look done", "records + summary + counters"};
            for (int k = 1; k < 7; k++)
                std::fprintf(stderr, "  one-launch pass: %-45s +%6.2f us  (at %6.2f)\n", name[k],
                             (double)(long long)(t[k] - t[k - 1]) / 100.0, (double)(long long)(t[k] - t[0]) / 100.0);
        }
        // ... and of the last sixteen, workgroup by workgroup: when the first and the last workgroup passed each
        // stage, from the pass's first entry (passes in flight side by side: how they stretch each other)
        std::vector<unsigned long long> w(16 * 32 * 8);
        if (hipMemcpy(w.data(), c->d_timeline + 1024, w.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long origin = ~0ull;
            for (unsigned long long v : w) if (v && v < origin) origin = v;
            for (int ps = 0; ps < 16; ps++) {
                unsigned long long lo[8], hi[8] = {};
                for (auto &v : lo) v = ~0ull;
                for (int g = 0; g < 32; g++)
                    for (int k = 0; k < 8; k++) {
                        const unsigned long long v = w[(ps * 32 + g) * 8 + k];
                        if (!v) continue;
                        lo[k] = std::min(lo[k], v);
                        hi[k] = std::max(hi[k], v);
                    }
                if (!hi[0]) continue;
                std::fprintf(stderr, "  pass %2d entered at %8.2f us:", ps, (double)(lo[0] - origin) / 100.0);
                for (int k = 0; k < 7; k++)
                    if (hi[k]) std::fprintf(stderr, "  s%d %.1f..%.1f", k, (double)(lo[k] - lo[0]) / 100.0, (double)(hi[k] - lo[0]) / 100.0);
                std::fprintf(stderr, "\n");
            }
        }
        (void)hipFree(c->d_timeline);
    } else if (c->d_timeline && c->debug_stop == 100) {
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
            // ... and how evenly the work fell: a workgroup's total (its waves agree to a barrier wait) and its
            // wave-private stage, over the workgroups -- the launch ends with the slowest one
            std::vector<double> tot, late;
            for (size_t w = 0; w + 3 < kTimelineWords / 8; w += 4) {
                double t = 0, l = 0;
                for (int k = 0; k < 8; k++) t += (double)tl[w * 8 + k];
                for (size_t v = 0; v < 4; v++) l = std::max(l, (double)tl[(w + v) * 8 + 4]);
                if (t == 0) continue;
                tot.push_back(t);
                late.push_back(l);
            }
            auto pct = [](std::vector<double> &v, double q) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[(size_t)(q * (double)(v.size() - 1))]; };
            double mt = 0, ml = 0;
            for (double v : tot) mt += v;
            for (double v : late) ml += v;
            mt /= tot.empty() ? 1 : (double)tot.size();
            ml /= late.empty() ? 1 : (double)late.size();
            std::fprintf(stderr, "  per workgroup (wave 0's total): mean %.0f  p50 %.0f  p99 %.0f  max %.0f  (max / mean %.3f)\n", mt, pct(tot, 0.5),
                         pct(tot, 0.99), pct(tot, 1.0), mt > 0 ? pct(tot, 1.0) / mt : 0.0);
            std::fprintf(stderr, "  per workgroup (its slowest wave's P3-5): mean %.0f  p50 %.0f  p99 %.0f  max %.0f  (max / mean %.3f)\n", ml,
                         pct(late, 0.5), pct(late, 0.99), pct(late, 1.0), ml > 0 ? pct(late, 1.0) / ml : 0.0);
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
    if (c->private_streams) {
        for (hipStream_t q : {c->own_stream, c->tail_stream, c->score_stream})
            if (q) (void)hipStreamDestroy(q);
        for (hipStream_t q : c->scan_stream)
            if (q) (void)hipStreamDestroy(q);
    } else if (c->own_stream) {
        // the scan / tail / score streams are the device's (shared, never destroyed); the context's own goes back
        std::lock_guard<std::mutex> lk(g_streams_mu);
        for (DeviceStreams *d : g_device_streams)
            if (d->device == c->device) d->free_own.push_back(c->own_stream);
    }
    delete c;
}

int adsb_set_stream(adsb_ctx *c, void *hip_stream)
try {
    if (!c) return ADSB_ERR_INVALID;
    if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return ADSB_OK;
} ADSB_ABI_CATCH

int adsb_set_profiling(adsb_ctx *c, int enabled)
try {
    if (!c) return ADSB_ERR_INVALID;
    // (the level decides whether a pass of a few buffers is one launch or three, and the cross-stream edges of a
    // pass are worked out from what the passes in flight are: like the other settings, only between passes)
    if (c->submitted != c->delivered || c->shard_active) return ADSB_ERR_BUSY;
    c->profiling = enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled);
    return ADSB_OK;
} ADSB_ABI_CATCH

int adsb_set_carry_over(adsb_ctx *c, int enabled)
try {
    if (!c) return ADSB_ERR_INVALID;
    if (c->submitted != c->delivered || c->shard_active) return ADSB_ERR_BUSY;
    ADSB_ON_DEVICE(c);
    c->carry_over = enabled != 0;
    // the stream starts here: nothing precedes the next call
    for (int si = 0; si < c->n_slots; si++)
        HIP_TRY(c, hipMemsetAsync(c->slot[si].d_carry, 0, kCarrySamples * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_carry_next, 0, kCarrySamples * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ADSB_OK;
} ADSB_ABI_CATCH

int adsb_icao_flush(adsb_ctx *c)
{
    if (!c) return ADSB_ERR_INVALID;
    // Takes effect for everything submitted after this call: the next pass's reset kernel
    // clears the device bitmap (stream-ordered), and the host filter is flushed when that
    // pass is collected, after the passes before it have been replayed.
    c->flush_pending = true;
    return ADSB_OK;
}

uint64_t adsb_host_sorts(const adsb_ctx *c) { return c ? c->host_sorts : 0; }
uint64_t adsb_host_replays(const adsb_ctx *c) { return c ? c->host_replays : 0; }
uint64_t adsb_host_rematches(const adsb_ctx *c) { return c ? c->rematches : 0; }
int adsb_max_in_flight(const adsb_ctx *c) { return c ? c->n_slots : 0; }

int adsb_get_stats(const adsb_ctx *c, adsb_stats *out)
{
    if (!c || !out) return ADSB_ERR_INVALID;
    *out = c->stats;
    return ADSB_OK;
}

const char *adsb_last_error(const adsb_ctx *c) { return c ? c->last_error.c_str() : ""; }

const char *adsb_version(void) { return "adsb_hip 0.21 gfx950 scan=v9-tile-buckets tail=v7-folded-supersets multi=v3-bounded-waits streams=v2-own-queues"; }

}  // extern "C"
