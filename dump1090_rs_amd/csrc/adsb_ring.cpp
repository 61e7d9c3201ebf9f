// adsb_ring.cpp -- the pinned streaming ring (dump1090_rs/src/main.rs:154-167: one read, one demodulation).
#include "adsb_ctx.h"

using namespace adsb::host;

namespace {
constexpr uint64_t kRingInPlaceChunks = kInlineTailChunks;  // every one-launch size
}

extern "C" {

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
    for (int k = 0; k < c->n_slots; k++) {
        auto &r = c->ring[k];
        // (mapped and coherent: slots of a few buffers are read in place by the pass itself)
        HIP_TRY(c, hipHostMalloc((void **)&r.h_iq, samples_per_slot * 4, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(c, hipHostGetDevicePointer(&r.h_iq_dev, r.h_iq, 0));
        HIP_TRY(c, hipMalloc(&r.d_iq, samples_per_slot * 4));
        HIP_TRY(c, hipEventCreateWithFlags(&r.copied, hipEventDisableTiming));
    }
    c->ring_samples = samples_per_slot;
    return ADSB_OK;
}

int adsb_ring_acquire(adsb_ctx *c, int16_t **host_iq, size_t *capacity_samples)
{
    if (!c || !host_iq || !c->ring_samples) return ADSB_ERR_INVALID;
    if (c->slot[c->submitted % (uint64_t)c->n_slots].busy || c->slot[c->submitted % (uint64_t)c->n_slots].parked) return ADSB_ERR_BUSY;  // collect the oldest pass first
    *host_iq = c->ring[c->submitted % (uint64_t)c->n_slots].h_iq;
    if (capacity_samples) *capacity_samples = c->ring_samples;
    return ADSB_OK;
}

int adsb_ring_submit(adsb_ctx *c, size_t n_samples)
{
    if (!c || !c->ring_samples || n_samples == 0 || n_samples > c->ring_samples) return ADSB_ERR_INVALID;
    if (c->slot[c->submitted % (uint64_t)c->n_slots].busy || c->slot[c->submitted % (uint64_t)c->n_slots].parked) return ADSB_ERR_BUSY;
    HIP_TRY(c, hipSetDevice(c->device));
    auto &r = c->ring[c->submitted % (uint64_t)c->n_slots];
    // A slot of a few buffers (the reference reads and demodulates 131072 samples at a time,
    // dump1090_rs/src/main.rs:161-167) is one launch that reads the pinned buffer in place over the link:
    // no copy command, no event, no staging -- the pass is as long as the transfer either way, and the
    // host side of it is a single launch.  Larger slots are copied while the slots before them compute.
    static const bool always_copy = tuning_env("ADSB_RING_COPY") != nullptr;
    // (measured with eight in flight: 8.5 / 9.7 / 10.1 / 10.0 / 10.2 Gsample/s at 1 / 2 / 4 / 8 / 16 buffers per slot
    // read in place; copied first and then one launch: 8.5 at 8, 10.2 at 16, and far less below)
    if (!always_copy && (n_samples + kChunkSamples - 1) / kChunkSamples <= kRingInPlaceChunks && !c->carry_over)
        return submit(c, r.h_iq_dev, false, n_samples, false, input_ready_now());
    // H2D on the copy stream; the pass on the compute stream waits for it, so this slot's
    // transfer overlaps the other slot's kernels
    {
        HT(c, HT_RING_MEMCPY);
        HIP_TRY(c, hipMemcpyAsync(r.d_iq, r.h_iq, n_samples * 4, hipMemcpyHostToDevice, c->copy_stream));
    }
    {
        HT(c, HT_RING_EVENT);
        HIP_TRY(c, hipEventRecord(r.copied, c->copy_stream));
    }
    return submit(c, r.d_iq, false, n_samples, false, r.copied);
}

}  // extern "C"
