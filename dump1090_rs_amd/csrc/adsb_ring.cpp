// adsb_ring.cpp -- the pinned streaming ring (dump1090_rs/src/main.rs:154-167: one read, one demodulation).
#include "adsb_ctx.h"

using namespace adsb::host;

namespace {
// Slots up to this many buffers are always read in place; up to the one-launch size when nothing else is in flight.
constexpr uint32_t kRingInPlaceChunks = 2;
}

extern "C" {

int adsb_ring_create(adsb_ctx *c, size_t samples_per_slot)
try {
    if (!c || samples_per_slot == 0 || c->ring_samples) return ADSB_ERR_INVALID;
    if ((samples_per_slot + kChunkSamples - 1) / kChunkSamples > c->max_chunks) return ADSB_ERR_INVALID;
    ADSB_ON_DEVICE(c);
    auto body = [&]() -> int {
        // every slot's pinned buffer from ONE allocation, and every staging buffer from one (slot starts 4 KB aligned):
        // mapped and coherent -- slots of a few buffers are read in place by the pass itself
        const size_t stride = (samples_per_slot * 4 + 4095) & ~(size_t)4095;
        char *h_dev = nullptr;
        HIP_TRY(c, hipHostMalloc((void **)&c->ring_h_block, stride * (size_t)c->n_slots, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&h_dev, c->ring_h_block, 0));
        HIP_TRY(c, hipMalloc((void **)&c->ring_d_block, stride * (size_t)c->n_slots));
        for (int k = 0; k < c->n_slots; k++) {
            auto &r = c->ring[k];
            r.h_iq = reinterpret_cast<int16_t *>(c->ring_h_block + stride * (size_t)k);
            r.h_iq_dev = h_dev + stride * (size_t)k;
            r.d_iq = c->ring_d_block + stride * (size_t)k;
        }
        // One small copy per slot now, on an idle stream.  The runtime sets something up on the first copy between a
        // pair of buffers; left to the first pipelined submit (a ring that starts with its slots read in place) every
        // later hipMemcpyAsync of the ring took 12-19 us of the submitting thread instead of 2-5 (measured: 16-buffer
        // slots, eight in flight, 12.5 Gsample/s instead of 13.2; profiles/r4_ring_copy_ab.txt).
        const size_t warm = std::min<size_t>(samples_per_slot * 4, 64 << 10);
        for (int k = 0; k < c->n_slots; k++)
            HIP_TRY(c, hipMemcpyAsync(c->ring[k].d_iq, c->ring[k].h_iq, warm, hipMemcpyHostToDevice, c->scan_stream[k % c->n_scan_streams]));
        for (int k = 0; k < c->n_scan_streams; k++) HIP_TRY(c, hipStreamSynchronize(c->scan_stream[k]));
        return (int)ADSB_OK;
    };
    const int rc = body();
    if (rc != ADSB_OK) {
        // nothing half-made stays behind: a retry starts from scratch instead of overwriting (and leaking) these
        for (int k = 0; k < c->n_scan_streams; k++) (void)hipStreamSynchronize(c->scan_stream[k]);
        if (c->ring_h_block) (void)hipHostFree(c->ring_h_block);
        if (c->ring_d_block) (void)hipFree(c->ring_d_block);
        c->ring_h_block = c->ring_d_block = nullptr;
        for (auto &r : c->ring) r = adsb_ctx::RingSlot{};
        return rc;
    }
    c->ring_samples = samples_per_slot;
    return ADSB_OK;
} ADSB_ABI_CATCH

int adsb_ring_acquire(adsb_ctx *c, int16_t **host_iq, size_t *capacity_samples)
try {
    if (!c || !host_iq || !c->ring_samples) return ADSB_ERR_INVALID;
    if (c->slot[c->submitted % (uint64_t)c->n_slots].busy || c->slot[c->submitted % (uint64_t)c->n_slots].parked) return ADSB_ERR_BUSY;  // collect the oldest pass first
    *host_iq = c->ring[c->submitted % (uint64_t)c->n_slots].h_iq;
    if (capacity_samples) *capacity_samples = c->ring_samples;
    return ADSB_OK;
} ADSB_ABI_CATCH

int adsb_ring_submit(adsb_ctx *c, size_t n_samples)
try {
    if (!c || !c->ring_samples || n_samples == 0 || n_samples > c->ring_samples) return ADSB_ERR_INVALID;
    if (c->slot[c->submitted % (uint64_t)c->n_slots].busy || c->slot[c->submitted % (uint64_t)c->n_slots].parked) return ADSB_ERR_BUSY;
    ADSB_ON_DEVICE(c);
    auto &r = c->ring[c->submitted % (uint64_t)c->n_slots];
    const uint32_t n_chunks = (uint32_t)((n_samples + kChunkSamples - 1) / kChunkSamples);
    // Two ways for a slot to reach the pass (dump1090_rs/src/main.rs:161-167 reads and demodulates 131072
    // samples at a time; a host may batch more per slot):
    //  * read in place over the link by the pass's one launch (no copy command, no staging: the host side is
    //    a single launch).  A kernel pulls ~39 GB/s through the link and the pass is as long as that transfer:
    //    the shortest way from a filled slot to its frames, and the fastest at one or two buffers per slot
    //    whatever is in flight (a copy command costs the submitting thread more than the launch itself);
    //  * copied by the copy engine (~52 GB/s) on the pass's OWN scan stream, in order in front of its first
    //    launch -- no event, no copy stream: while it runs, the passes on the other scan streams compute.  From
    //    three buffers per slot with another pass in flight this is the faster pipeline (4 / 8 / 16 buffers,
    //    three in flight: 10.9 / 11.0 / 13.1 Gsample/s against 9.1 / 9.5 / 10.2 in place; one in flight: 6.3 /
    //    8.1 / 9.7 against 7.6 / 9.4 / 10.1 -- profiles/r4_ring_copy_ab.txt), and the only way for slots of more
    //    than one launch.
    bool in_place = one_launch_pass(c, n_chunks) && !c->carry_over &&
                    (n_chunks <= kRingInPlaceChunks || c->submitted == c->delivered);
    if (const char *e = tuning_env("ADSB_RING_COPY")) {   // measurement aid (tuning build only): 0 in place where possible, else copy
        const int mode = std::atoi(e);
        const bool can = one_launch_pass(c, n_chunks) && !c->carry_over;
        in_place = can && mode == 0;
    }
    if (in_place) {
        c->next_src_host = true;
        const int rc = submit(c, r.h_iq_dev, false, n_samples, false, input_ready_now());
        c->next_src_host = false;
        return rc;
    }
    hipStream_t q = next_scan_stream(c, n_chunks);
    {
        HT(c, HT_RING_MEMCPY);
        HIP_TRY(c, hipMemcpyAsync(r.d_iq, r.h_iq, n_samples * 4, hipMemcpyHostToDevice, q));
    }
    // (enqueue_pass checks that the pass does land on q, and orders it behind q with an event if it ever does not)
    c->input_on_stream = q;
    const int rc = submit(c, r.d_iq, false, n_samples, false, input_ready_now());
    c->input_on_stream = nullptr;
    return rc;
} ADSB_ABI_CATCH

int adsb_host_register(adsb_ctx *c, void *host_ptr, size_t bytes)
try {
    if (!c || !host_ptr || bytes == 0) return ADSB_ERR_INVALID;
    char *b = static_cast<char *>(host_ptr);
    for (const auto &r : c->host_ranges)
        if (b < r.base + r.bytes && r.base < b + bytes) return ADSB_ERR_INVALID;
    ADSB_ON_DEVICE(c);
    HIP_TRY(c, hipHostRegister(host_ptr, bytes, hipHostRegisterMapped));
    void *dev = nullptr;
    if (hipError_t e = hipHostGetDevicePointer(&dev, host_ptr, 0); e != hipSuccess) {
        (void)hipHostUnregister(host_ptr);
        return fail(c, e, "hipHostGetDevicePointer");
    }
    adsb_ctx::HostRange r;
    r.base = b;
    r.bytes = bytes;
    r.dev = static_cast<char *>(dev);
    c->host_ranges.push_back(r);
    return ADSB_OK;
} ADSB_ABI_CATCH

int adsb_host_unregister(adsb_ctx *c, void *host_ptr)
try {
    if (!c || !host_ptr) return ADSB_ERR_INVALID;
    if (c->submitted != c->delivered) return ADSB_ERR_BUSY;
    for (size_t k = 0; k < c->host_ranges.size(); k++)
        if (c->host_ranges[k].base == static_cast<char *>(host_ptr)) {
            ADSB_ON_DEVICE(c);
            HIP_TRY(c, hipHostUnregister(host_ptr));
            c->host_ranges.erase(c->host_ranges.begin() + (long)k);
            return ADSB_OK;
        }
    return ADSB_ERR_INVALID;
} ADSB_ABI_CATCH

}  // extern "C"
