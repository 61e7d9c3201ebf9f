/*
 * include/adsb_hip.h -- C ABI of libadsb_hip.so, the MI355X (gfx950) drop-in for the
 * demod_2400 hot path of rsadsb/dump1090_rs v0.8.1.
 *
 * The reference has no FFI or plugin interface of its own: the boundary it offers
 * is its Rust library API.  Each entry point below replaces one of those items; a
 * Rust host declares them in an `extern "C"` block and keeps its own
 * to_mag / demodulate2400 / ModeSMessage wrappers (INTEGRATION.md shows the stub).
 *
 *   reference item (file:line)                              replaced by
 *   ------------------------------------------------------  -------------------------
 *   utils::to_mag                  src/utils.rs:43-58       adsb_to_mag
 *   MagnitudeBuffer / push         src/lib.rs:30-51         the (data[131398], length) pair
 *   demod_2400::demodulate2400     src/demod_2400.rs:115    adsb_demodulate2400
 *   ModeSMessage / buffer()        src/demod_2400.rs:92-112 adsb_msg (msg, len)
 *   icao_filter::icao_flush        src/icao_filter.rs:11    adsb_icao_flush
 *   to_mag + demodulate2400 per buffer, as composed at
 *     dump1090_rs/src/main.rs:166-167 and
 *     benches/demod_benchmark.rs:10-11                      adsb_demod_iq / _device
 *
 * Conventions: every function returns ADSB_OK (0) or a negative adsb_status and
 * never throws, aborts or panics across the ABI (an allocation that fails inside
 * comes back as ADSB_ERR_NOMEM); the caller owns every host
 * buffer; all pointers are plain host pointers except where the name says
 * `device`.  After ADSB_ERR_HIP or ADSB_ERR_NOMEM from a demodulating call the
 * pass it was working on is lost and the filter may have seen part of it: an
 * adsb_ctx is then best destroyed and made anew (or adsb_icao_flush'ed, if starting
 * from an empty filter is acceptable); an adsb_multi says so itself
 * (ADSB_ERR_POISONED) and restarts on adsb_multi_icao_flush.  One adsb_ctx per host thread / per GPU: the ICAO address filter
 * (process-global statics in the reference, src/icao_filter.rs:8-9) lives in the
 * context, so contexts are independent streams.  There is no CPU fallback:
 * adsb_create fails with ADSB_ERR_NO_DEVICE when no gfx950 device is usable.
 * The host side of the library is built for x86-64 Linux hosts (its spin loops are
 * the `pause` instruction; thread placement reads sysfs): the hosts MI355X boards sit in.
 */
#ifndef ADSB_HIP_H
#define ADSB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/lib.rs:22-26 */
#define ADSB_MODES_MAG_BUF_SAMPLES 131072u
#define ADSB_TRAILING_SAMPLES 326u
#define ADSB_MODES_LONG_MSG_BYTES 14u
#define ADSB_MODES_SHORT_MSG_BYTES 7u
/* length of MagnitudeBuffer.data, src/lib.rs:31 */
#define ADSB_MAG_DATA_LEN (ADSB_TRAILING_SAMPLES + ADSB_MODES_MAG_BUF_SAMPLES)

typedef enum {
    ADSB_OK = 0,
    ADSB_ERR_INVALID = -1,   /* null pointer / bad argument */
    ADSB_ERR_NO_DEVICE = -2, /* no usable HIP device (there is no CPU fallback) */
    ADSB_ERR_HIP = -3,       /* a HIP runtime call failed; see adsb_last_error */
    ADSB_ERR_TOO_LONG = -4,  /* more than 131072 samples handed to a one-buffer call:
                                the reference panics here (src/lib.rs:48) */
    ADSB_ERR_CAPACITY = -5,  /* `out` too small; *n_out holds the required count and the
                                first `cap` messages were written.  The pass itself is done (the
                                filter has advanced, as after demodulate2400 returned its Vec):
                                do not repeat the call -- adsb_fetch_messages hands out the whole
                                list, which the context keeps until its next demod call */
    ADSB_ERR_NOMEM = -6,
    ADSB_ERR_BUSY = -7,      /* submissions pending where none are allowed, or too many in flight */
    ADSB_ERR_POISONED = -8   /* an adsb_multi one of whose captures failed: what it computed after that capture is
                                not the single stream's any more, so nothing further is accepted or returned until
                                adsb_multi_icao_flush (with nothing in flight) has started the stream over */
} adsb_status;

#define ADSB_MAX_IN_FLIGHT 4
/* ... and for a context created with max_chunks <= 16, whose passes are a single launch each
 * (adsb_max_in_flight tells which a context got) */
#define ADSB_MAX_IN_FLIGHT_SMALL 8

typedef struct adsb_ctx adsb_ctx;

/* Mirrors ModeSMessage (src/demod_2400.rs:92-102) plus provenance.
 * buffer() == msg[0..len]. */
typedef struct {
    uint8_t msg[ADSB_MODES_LONG_MSG_BYTES];
    uint8_t len;       /* 7 (MsgLen::Short) or 14 (MsgLen::Long) */
    uint8_t try_phase; /* winning trial phase, 4..8 (demod_2400.rs:158) */
    int32_t score;     /* src/mode_s/mod.rs score of the winning trial */
    uint32_t j;        /* preamble index into MagnitudeBuffer.data */
    uint64_t chunk;    /* index of the 131072-sample buffer inside this call */
    double signal_level;
} adsb_msg;

/* Counters of the most recent demod call (diagnostics / bench). */
typedef struct {
    uint64_t n_samples;
    uint64_t n_chunks;
    uint64_t n_candidates;  /* j that passed preamble + SNR + quiet gates */
    uint64_t n_ap_entries;  /* address/parity trials deferred to the filter match */
    uint64_t n_records;     /* trial records replayed on the host */
    uint64_t n_messages;
    float ms_scan;          /* scan kernel (IQ -> candidates -> trials), HIP events */
    float ms_match;         /* address/parity match kernel */
    float ms_records;       /* record builder kernel */
    float ms_total_device;  /* first launch -> last kernel end */
    uint32_t retries;       /* device-list overflow fallbacks taken */
    float ms_scan_exclusive; /* the part of ms_scan after the previous pass's scan had finished: consecutive
                              * pipelined scans overlap (the next one's workgroups fill the CUs as the previous
                              * grid drains), so the sum of ms_scan over passes counts the overlap twice while
                              * the sum of ms_scan_exclusive is the device time the scans took altogether */
} adsb_stats;

/* Create a context on HIP device `device` (>= 0), sized to demodulate up to
 * `max_chunks` 131072-sample buffers per call (host-pointer calls stage through
 * a device buffer of that size; device-pointer calls only size the lists).
 * A context created with max_chunks <= 16 is one for the reference's own call shape (one read, one
 * demodulation, dump1090_rs/src/main.rs:161-167): each of its passes is a single kernel launch, and it
 * keeps ADSB_MAX_IN_FLIGHT_SMALL (8) of them in flight instead of ADSB_MAX_IN_FLIGHT (4).
 * Footprint (each pass in flight has its own lists):
 *   device   the address supersets the match tests against, one more than passes in flight in rotation: 2 MiB each
 *            (bit a = address a) in a large context, plus two for the device-side copy of the filter; 64 KB each
 *            (2^19 bits, address folded: a superset is all the match needs) in a context created with
 *            max_chunks <= 16, which never scores on the device.  Per pass in flight the address/parity list, the
 *            hit list with the scan's bit fields per hit, and (large contexts) the scoring buffers: ~0.8 MB each for
 *            max_chunks = 1 (~7 MB in all, 0.6 of them bitmaps), ~109 MB each for 512 (~450 MB in all);
 *   pinned host (mapped, written by the kernels; one allocation per context)  per pass in flight 32 B per trial
 *            record (4096 + 1024 max_chunks of them) + in large contexts 44 B per scored message slot
 *            (min(that, 262144)): ~1.3 MB in all for max_chunks = 1, ~115 MB for 512; host-pointer calls of a few
 *            buffers add a pinned staging buffer of their size, the ring its slots.
 * Input denser than the lists are sized for (several times a busy airspace) is still
 * demodulated exactly, buffer by buffer through worst-case lists allocated on first use
 * (another 10 MB of device and 20 MB of pinned memory; stats.retries). */
int adsb_create(adsb_ctx **out, int device, size_t max_chunks);
void adsb_destroy(adsb_ctx *ctx);

/* Order the context's work behind the caller's HIP stream (e.g. torch's current stream)
 * instead of the context's own: whatever that stream has been given before a call (the
 * kernels or copies that produce the IQ) is complete before the pass reads its input.  The
 * passes themselves run on the context's internal streams; their results are handed over
 * by the blocking call or by adsb_collect, not through the stream.  Pass NULL to go back to
 * the private stream. */
int adsb_set_stream(adsb_ctx *ctx, void *hip_stream);
/* HIP-event timing of the kernels: 0 = off, 1 = ms_scan only (default; two events),
 * 2 = also ms_match / ms_records / ms_total_device (an event costs the
 * stream several microseconds, so level 2 slows a call down noticeably). */
int adsb_set_profiling(adsb_ctx *ctx, int level);

/* == icao_filter::icao_flush() (src/icao_filter.rs:11-17) for this context.  Stream-ordered: costs nothing by
 * itself; the next pass starts on the next address superset of the rotation (a large context's is clean already,
 * a context for passes of a few buffers has that pass clear its 64 KB when it starts) and waits for no pass in
 * flight -- a flush before every pipelined call, the reference's benchmark shape (benches/demod_benchmark.rs:9),
 * costs the same as none. */
int adsb_icao_flush(adsb_ctx *ctx);

/* == utils::to_mag (src/utils.rs:43-58).  iq_re_im is the in-memory
 * Complex<i16> order: first i16 = re, second = im.  data_out must hold
 * ADSB_MAG_DATA_LEN u16: [0,326) zero, [326,326+n) magnitudes, rest zero.
 * n > 131072 -> ADSB_ERR_TOO_LONG. */
int adsb_to_mag(adsb_ctx *ctx, const int16_t *iq_re_im, size_t n, uint16_t *data_out,
                size_t *length_out);

/* == demod_2400::demodulate2400 (src/demod_2400.rs:115-212) on one
 * MagnitudeBuffer given as (data[ADSB_MAG_DATA_LEN], length).  Messages come
 * out in ascending j, exactly the reference's Vec order. */
int adsb_demodulate2400(adsb_ctx *ctx, const uint16_t *data, size_t length, adsb_msg *out,
                        size_t cap, size_t *n_out);

/* to_mag + demodulate2400 over consecutive 131072-sample buffers of a host IQ
 * stream of any length (the last buffer may be short), the filter persisting
 * from buffer to buffer as in dump1090_rs/src/main.rs:154-167.  Output order:
 * ascending (chunk, j). */
int adsb_demod_iq(adsb_ctx *ctx, const int16_t *iq_re_im, size_t n_samples, adsb_msg *out,
                  size_t cap, size_t *n_out);

/* Same, the IQ already resident in device memory (16-byte aligned). */
int adsb_demod_iq_device(adsb_ctx *ctx, const void *device_iq_re_im, size_t n_samples,
                         adsb_msg *out, size_t cap, size_t *n_out);

/* Asynchronous form of adsb_demod_iq_device for a host that keeps the GPU fed: enqueue one
 * pass (kernels + result copy) and return at once; at most ADSB_MAX_IN_FLIGHT passes
 * (adsb_max_in_flight) may be pending.  adsb_collect waits for the OLDEST pending pass, replays it through the
 * filter and returns its messages, so results come back in submission order and the
 * host replay of pass i overlaps the device scan of pass i+1.  adsb_icao_flush applies
 * to the passes submitted after it.  The synchronous calls return ADSB_ERR_BUSY while
 * anything is pending.  device_iq must stay valid and unchanged until its collect.  One
 * submission holds at most the max_chunks buffers the context was created for
 * (ADSB_ERR_INVALID beyond that; adsb_demod_iq_device cuts longer inputs into such passes
 * itself). */
int adsb_submit_iq_device(adsb_ctx *ctx, const void *device_iq_re_im, size_t n_samples);
int adsb_collect(adsb_ctx *ctx, adsb_msg *out, size_t cap, size_t *n_out);
int adsb_pending(const adsb_ctx *ctx);
/* How many passes this context lets be in flight: ADSB_MAX_IN_FLIGHT, or ADSB_MAX_IN_FLIGHT_SMALL for a
 * context created for at most 16 buffers per pass (the ring of such a context has that many slots). */
int adsb_max_in_flight(const adsb_ctx *ctx);

/* The complete message list of the most recent call that returned ADSB_ERR_CAPACITY
 * (adsb_demodulate2400, adsb_demod_iq[_device], adsb_collect): that call already ran the pass and
 * updated the filter, so repeating it would score against a different filter; this returns what it
 * produced.  The list is kept until the next of those calls.  ADSB_ERR_INVALID when nothing is held. */
int adsb_fetch_messages(adsb_ctx *ctx, adsb_msg *out, size_t cap, size_t *n_out);

/* Streaming ring for a host that produces IQ (an SDR read loop, dump1090_rs/src/main.rs:
 * 154-167): adsb_max_in_flight() (4 or 8) pinned host buffers of `samples_per_slot` samples (4 bytes
 * each) with a device staging buffer each.  Fill the buffer adsb_ring_acquire hands out (e.g.
 * read the SDR straight into it), adsb_ring_submit(n) starts the slot's pass -- one launch that
 * reads a slot of one or two buffers in place over the link; larger slots are copied first, on the
 * pass's own stream --, adsb_collect returns the oldest pass's messages.  While one slot's pass
 * runs, the next slots' transfers are in flight.  samples_per_slot may not exceed the
 * context's max_chunks buffers; adsb_ring_acquire returns ADSB_ERR_BUSY until the pass
 * that last used the slot has been collected. */
int adsb_ring_create(adsb_ctx *ctx, size_t samples_per_slot);
int adsb_ring_acquire(adsb_ctx *ctx, int16_t **host_iq_re_im, size_t *capacity_samples);
int adsb_ring_submit(adsb_ctx *ctx, size_t n_samples);

/* For a host that keeps its own sample buffer (the Vec the SDR reads land in, dump1090_rs/src/main.rs:
 * 154-167, is allocated once): pin it and map it for the device, so that adsb_demod_iq on samples
 * INSIDE it (16-byte aligned start, at most 16 buffers) reads them in place over the link -- one kernel
 * launch, no copy of the samples on the host.  The memory stays the caller's; it must not be freed
 * before adsb_host_unregister (adsb_destroy unregisters what is left).  Samples outside any registered
 * range take the usual path.  ADSB_ERR_INVALID for a range that overlaps a registered one. */
int adsb_host_register(adsb_ctx *ctx, void *host_ptr, size_t bytes);
int adsb_host_unregister(adsb_ctx *ctx, void *host_ptr);

/* src/utils.rs:23-40 read_test_data: file pairs are [im][re] little-endian;
 * writes in-memory {re, im}.  Returns samples read via *n_out. */
int adsb_read_test_data(const char *path, int16_t *iq_re_im, size_t max_samples, size_t *n_out);

/* The raw output line of dump1090_rs/src/main.rs:172-176 for one message:
 * "*" + lowercase hex of buffer() + ";\n" (the format port 30002 clients such as adsb_deku's
 * radar read).  Writes 17 or 31 characters plus a terminating NUL into out (>= 32 bytes);
 * returns the length without the NUL, or a negative status.  Host only. */
int adsb_format_raw(const adsb_msg *msg, char *out, size_t out_size);

/* One trial message as the device hands it to the host replay. */
typedef struct {
    uint64_t power;   /* bits 0..39: sum of the 33 squared magnitudes from j+19 (demod_2400.rs:
                       * 191-196; below 2^38); bits 40..63: see pad */
    uint32_t chunk;
    uint32_t j_tp;    /* j | try_phase << 24 */
    uint8_t msg[ADSB_MODES_LONG_MSG_BYTES];
    uint16_t pad;     /* 0: nothing more.  Records from adsb_shard_finish set bit 0: bits 40..63 of
                       * `power` hold the CRC residual of msg over its own length (src/crc.rs:263-282),
                       * which spares adsb_replay_records the walk over the bytes; and bit 1: bits 4..15
                       * hold icao_hash (src/icao_filter.rs:19-43) of the value the message's DF asks the
                       * filter about -- the residual for DF 0,4,5,16,20,21,24-31, else the address */
} adsb_trial;

/* Carry-over mode -- opt-in and NOT the reference's semantics.  dump1090_rs starts every
 * MagnitudeBuffer with 326 zero samples (src/lib.rs:24,36-44; the constant is what upstream
 * C dump1090 uses to carry the end of the previous buffer over), so a frame that straddles
 * two buffers is lost.  With carry-over enabled the lead-in of every 131072-sample buffer
 * holds the 326 samples that preceded it in the stream -- within a call and from the
 * previous call on this context -- and such frames are decoded in the later buffer
 * (j < 326 there).  Applies to the IQ entry points (adsb_demod_iq*, submit/collect, the
 * ring); adsb_to_mag / adsb_demodulate2400 keep working on the caller's buffer as is.
 * Enabling or disabling restarts the stream (nothing precedes the next call).  Returns
 * ADSB_ERR_BUSY while passes are pending. */
int adsb_set_carry_over(adsb_ctx *ctx, int enabled);

/* Sharded capture: one capture cut into contiguous ranges of 131072-sample buffers, one
 * range per GPU (BASELINE config 4; the reference's loop dump1090_rs/src/main.rs:161-167
 * is one stream with one process-global filter, src/icao_filter.rs:8-9).  Shards run
 * independently except for that filter, so each runs in two phases around a small host-side
 * exchange:
 *   adsb_shard_scan    scans the shard (device_iq: this shard's samples, 16-byte aligned)
 *                      and returns the 24-bit addresses its clean DF11 (IID 0) / DF17 frames
 *                      will add to the filter (src/mode_s/mod.rs:80-84, 97-99), sorted;
 *   the caller hands every shard the union of all shards' lists (a few KB);
 *   adsb_shard_finish  adds them to the shard's address superset, matches the shard's
 *                      address/parity trials against it and returns the raw trial records
 *                      (chunk = buffer index within the shard).
 * Whoever gathers all records adds each shard's first buffer index to `chunk` and replays
 * them once with adsb_replay_records: the result is the single-stream one.  Between the two
 * calls the context accepts no other work (ADSB_ERR_BUSY).  `cap` too small ->
 * ADSB_ERR_CAPACITY with the required count in *n_addrs / *n_records; for
 * adsb_shard_scan the shard stays parked and adsb_shard_finish may still be called. */
int adsb_shard_scan(adsb_ctx *ctx, const void *device_iq, size_t n_samples, uint32_t *addrs_out,
                    size_t cap, size_t *n_addrs);
int adsb_shard_finish(adsb_ctx *ctx, const uint32_t *extra_addrs, size_t n_extra,
                      adsb_trial *records_out, size_t cap, size_t *n_records);

/* ---------------------------------------------------------------------------------------------------------
 * One capture over several GPUs from ONE process (BASELINE config 4).  The reference is one process with
 * one loop and one process-global filter (dump1090_rs/src/main.rs:154-167, src/icao_filter.rs:8-9); an
 * adsb_multi keeps that shape for its caller -- one handle, one ICAO filter, one message list in the
 * reference's order -- over N devices: the capture is cut into contiguous ranges of 131072-sample buffers,
 * one per device; inside, a context and a host thread per device run the two shard phases above, the
 * learned addresses are united in memory between them, and the caller's thread replays all shards' trial
 * records once, in global (buffer, j, try_phase) order, through the one filter (a dense stream's shards are scored
 * on their devices instead -- against the filter as it stood plus what the shards before them add -- and only
 * concatenated here; a capture of >= 8192 records that the host does score --
 * a busy sky -- with the help of up to six more threads, created when the first such capture is collected: every
 * record is scored against the filter as it was plus the record numbers at which the capture's new addresses
 * enter it, which is the ordered replay's answer, computed side by side).  No process group, no
 * collective, nothing outside this library.  The result is the single-stream one, bit for bit.
 *
 *   adsb_multi_create        a context on each of devices[0..n) (a device may appear more than once: then
 *                            its shards share it), each for up to max_chunks_per_device buffers per capture
 *   adsb_multi_icao_flush    == icao_flush() for the ONE filter; applies to the captures submitted after it
 *   adsb_multi_demod_iq      a host capture of any length: cut, copied to the devices, demodulated (blocking)
 *   adsb_multi_demod_iq_device  the shards already resident: device_iq[k] / n_samples[k] = device k's range
 *                            (16-byte aligned; whole buffers except the capture's last non-empty shard;
 *                            adsb_multi_shard_range gives the even split), blocking
 *   adsb_multi_submit_iq_device / adsb_multi_collect   the same, asynchronously: up to
 *                            adsb_multi_max_in_flight (4) captures in flight, results in submission order;
 *                            the scans of capture i + 1 run while capture i is exchanged, matched and
 *                            replayed.  The samples must be complete (their producer synchronised) before
 *                            the call and unchanged until the capture is collected.
 *   adsb_multi_submit_iq     the asynchronous form for a HOST capture of at most n_devices x
 *                            max_chunks_per_device buffers: every device thread copies its range to its device in
 *                            front of its scan.  Out of memory from adsb_multi_host_alloc (pinned for every
 *                            device) the copies are DMAs, one per device over its own link, and overlap the
 *                            scans of the captures in flight; out of ordinary memory they go through the
 *                            runtime's staging buffers (a third of the rate).  The samples stay the caller's
 *                            and must not change before the capture is collected.  (Device side: a staging buffer of
 *                            max_chunks_per_device buffers per capture in flight and device, allocated the first
 *                            time the host form uses that slot and kept: up to 4 x 256 MiB per device at 512.)
 * adsb_msg.chunk is the buffer's index in the whole capture.  Errors and ADSB_ERR_CAPACITY behave as for
 * the one-device calls (adsb_multi_fetch_messages hands out the whole list of a capture whose `out` was too
 * small).  One adsb_multi is driven by one host thread at a time.  Like every entry point of this header, these leave
 * the calling thread's current HIP device as they found it.
 *
 * When a capture fails.  Every wait inside is bounded: a shard phase that has been out for 2 ms has its streams asked
 * (a stream error fails the shard), and one that is still out after the handle's timeout (adsb_multi_set_timeout_ms,
 * 30 s by default) fails it and gives the device up -- nothing more is enqueued on it, and adsb_multi_destroy leaks
 * that device's context instead of waiting for a kernel that never ends.  The capture's collect (or blocking call)
 * returns the failed shard's status, adsb_multi_last_error names the device, and the handle is POISONED: the one
 * filter and the devices' address supersets have missed that capture's additions, so the captures in flight behind it
 * and every later submission return ADSB_ERR_POISONED -- never a silently different frame list.  Collect what is in
 * flight (each returns ADSB_ERR_POISONED), then adsb_multi_icao_flush: it resets every device's context, the filter
 * and the exchange state, and the stream starts over from an empty filter, as after icao_flush().  If a device cannot
 * be reset (it was given up, or the reset itself fails) the flush returns that error and the handle stays poisoned:
 * destroy it.  adsb_multi_destroy never blocks on a device. */
typedef struct adsb_multi adsb_multi;
typedef struct {
    uint64_t n_samples;
    uint64_t n_chunks;
    uint64_t n_candidates;
    uint64_t n_ap_entries;
    uint64_t n_records;          /* trial records replayed on the host */
    uint64_t n_messages;
    uint64_t n_addrs_exchanged;  /* learned addresses handed to every device between the phases */
    uint32_t n_devices;
    uint32_t retries;            /* shards that went buffer by buffer (list overflow) */
    float ms_wall;               /* submit -> the last device's records on the host (host clock) */
    float ms_phase1_max;         /* slowest device: phase 1 issued -> its summary seen (scan + records of learned frames) */
    float ms_phase2_max;         /* slowest device: phase 2 issued -> its summary seen (set addresses, match, records) */
    float ms_phase1_span;        /* first device's phase-1 issue -> last device's phase-1 summary */
    float ms_phase2_span;        /* ... the same for phase 2: ms_wall - the two spans = what the orchestration adds */
    float ms_exchange;           /* the union of the learned addresses + handing phase 2 to every device thread */
    float ms_replay;             /* the ordered replay in adsb_multi_collect */
    float reserved;
} adsb_multi_stats;

int adsb_multi_create(adsb_multi **out, const int *devices, int n_devices, size_t max_chunks_per_device);
void adsb_multi_destroy(adsb_multi *m);
int adsb_multi_device_count(const adsb_multi *m);
int adsb_multi_max_in_flight(const adsb_multi *m);
/* Device k's contiguous share of a capture of n_samples cut n_devices ways: whole buffers, sizes that differ by at
 * most one buffer, the ragged end with whoever holds the last buffer.  Host only. */
int adsb_multi_shard_range(size_t n_samples, int n_devices, int k, size_t *first_sample, size_t *n_samples_k);
int adsb_multi_icao_flush(adsb_multi *m);
int adsb_multi_demod_iq(adsb_multi *m, const int16_t *iq_re_im, size_t n_samples, adsb_msg *out, size_t cap,
                        size_t *n_out);
int adsb_multi_demod_iq_device(adsb_multi *m, const void *const *device_iq, const size_t *n_samples, adsb_msg *out,
                               size_t cap, size_t *n_out);
int adsb_multi_submit_iq_device(adsb_multi *m, const void *const *device_iq, const size_t *n_samples);
int adsb_multi_submit_iq(adsb_multi *m, const int16_t *iq_re_im, size_t n_samples);
/* Pinned host memory every device of the adsb_multi reads by DMA (hipHostMalloc, portable).  Freed by
 * adsb_multi_host_free (ADSB_ERR_BUSY while captures are in flight) or by adsb_multi_destroy. */
int adsb_multi_host_alloc(adsb_multi *m, size_t bytes, void **out);
int adsb_multi_host_free(adsb_multi *m, void *host_ptr);
int adsb_multi_collect(adsb_multi *m, adsb_msg *out, size_t cap, size_t *n_out);
int adsb_multi_pending(const adsb_multi *m);
int adsb_multi_fetch_messages(adsb_multi *m, adsb_msg *out, size_t cap, size_t *n_out);
/* Counters and host-clock timings of the capture collected last. */
int adsb_multi_get_stats(const adsb_multi *m, adsb_multi_stats *out);
/* Table A of the one filter (4096 u32, src/icao_filter.rs:8), for inspection.  ADSB_ERR_BUSY while captures are
 * in flight. */
int adsb_multi_filter_table(const adsb_multi *m, uint32_t *out4096);
const char *adsb_multi_last_error(const adsb_multi *m);
/* How the handle's threads wait for their devices.
 *   ADSB_WAIT_SPIN   a device thread polls its shard's summary in mapped memory while anything is out on its device
 *                    (a phase's end is seen within a microsecond; costs a CPU per device while captures are in flight),
 *                    the collector spins for 2 ms before it sleeps, the replay pool's workers stay hot for 1.5 ms;
 *   ADSB_WAIT_BLOCK  the device threads sleep between looks (a timed wait on their command queue: 25 us while a
 *                    phase is young, up to 1 ms as it ages; a command wakes them at once), the collector and the pool's
 *                    workers sleep on their condition variables right away: a quarter of the CPU time, the same
 *                    throughput for a caller with several captures in flight, 4-7 % more latency for one that has a
 *                    single capture in flight (profiles/r6_wait_policy.txt).  Where the process is short of CPUs
 *                    (a cgroup quota, an affinity mask) it is the only sane form: spinning threads take the CPU from
 *                    the one that has work, or get the whole process throttled (+43-54 % per capture under 4 CPUs);
 *   ADSB_WAIT_AUTO   (default) BLOCK when the CPUs the process may use (affinity mask, cgroup cpu.max) are fewer
 *                    than 2 x (devices + 3), else SPIN; decided at create and again by this call.
 * ADSB_ERR_BUSY while captures are in flight.  adsb_multi_get_wait returns what is in effect (SPIN or BLOCK). */
#define ADSB_WAIT_AUTO 0
#define ADSB_WAIT_SPIN 1
#define ADSB_WAIT_BLOCK 2
int adsb_multi_set_wait(adsb_multi *m, int mode);
int adsb_multi_get_wait(const adsb_multi *m);
/* After this long without a shard phase finishing, the device it runs on is given up (see "When a capture fails").
 * 0 = the default, 30 000 ms.  May be called at any time. */
int adsb_multi_set_timeout_ms(adsb_multi *m, uint32_t ms);

/* Host only, no device needed: the ordered replay every demod call ends with
 * (score_modes_message src/mode_s/mod.rs:34-139 + best-of-5 selection
 * src/demod_2400.rs:149-207 + icao_filter src/icao_filter.rs), exposed so the
 * sequential logic can be exercised on its own.  `records` may come in any order (they are
 * replayed by (chunk, j, try_phase) and left as they are).  `filter_table` is table A of the filter (4096 u32,
 * src/icao_filter.rs:8), read and updated. */
int adsb_replay_records(uint32_t *filter_table, adsb_trial *records, size_t n, adsb_msg *out,
                        size_t cap, size_t *n_out);

/* Device self-test: digest of the magnitude tail (sqrt, *65535+0.5, saturating
 * cast; src/utils.rs:54-55) over `count` consecutive f32 bit patterns of
 * X = im^2 + rn(re^2) starting at `first_bits`.  sum = sum of the u16 outputs,
 * xor = XOR of (out+1)*(2*bits+1).  A test sweeps all X in [0, 2^31] with it. */
int adsb_selftest_mag_digest(adsb_ctx *ctx, uint32_t first_bits, uint32_t count,
                             uint64_t *sum_out, uint64_t *xor_out);

/* Device self-test: the intermediate lists of one blocking pass over device-resident IQ, so that a
 * test can compare stages, not only frames, with the CPU restatement of the reference:
 *   cand[]  every position the gates let through (check_preamble + 3.5 dB + quiet samples,
 *           src/demod_2400.rs:127-146), as buffer << 32 | j, ascending;
 *   ap[]    every address/parity trial (DF 0,4,5,16,20,21,24-31, src/mode_s/mod.rs:56-72,110-135) with
 *           its CRC residual (src/crc.rs:263-282), as buffer << 45 | j << 28 | try_phase << 24 | residual,
 *           ascending.
 * The context's filter is not touched.  ADSB_ERR_CAPACITY with the required counts when a list does
 * not fit; ADSB_ERR_BUSY while passes are pending. */
int adsb_selftest_stage_lists(adsb_ctx *ctx, const void *device_iq_re_im, size_t n_samples, uint64_t *cand,
                              size_t cand_cap, size_t *n_cand, uint64_t *ap, size_t ap_cap, size_t *n_ap);

/* Device self-test, the two stages in front of cand[]: every position at which check_preamble returns
 * Some (src/demod_2400.rs:215-321) -> preamble[], and those that also pass the 3.5 dB test (:129) ->
 * snr[]; both as buffer << 32 | j, ascending (cand[] above is the subset of snr[] that also passes the
 * quiet samples, :135-146).  The kernel writes them from the reference's own sequence of tests applied
 * to its bit-parallel pattern matches, next to the verdict of its production gates; the call fails with
 * ADSB_ERR_HIP if the two ever disagree.  Same conventions as adsb_selftest_stage_lists. */
int adsb_selftest_gate_stages(adsb_ctx *ctx, const void *device_iq_re_im, size_t n_samples, uint64_t *preamble,
                              size_t preamble_cap, size_t *n_preamble, uint64_t *snr, size_t snr_cap, size_t *n_snr);

/* Test hook for the one-launch pass (passes of at most 16 buffers): inside such a pass a workgroup matches
 * its address/parity trials once every tile before its own has published its address bits, and waits for
 * that at most `polls` polls (200 by default, ~0.2 ms); a workgroup that gives up leaves the decision to a
 * second look by the pass's last workgroup.  0 makes every workgroup give up at once, so that a test can
 * run the second look on purpose; results never depend on the value.  ADSB_ERR_BUSY while passes are pending. */
int adsb_selftest_set_order_polls(adsb_ctx *ctx, uint32_t polls);

/* Test hooks for adsb_multi (0 = the default; results never depend on any of them; ADSB_ERR_BUSY while captures are in flight):
 * fresh_cap     a shard's scan lists the addresses its trials can add to the filter, at most this many (16384); a
 *               capture with more aircraft falls back to reading them out of its trial records -- a small value lets a
 *               test take that fallback on purpose;
 * parallel_min  captures of at least this many trial records (8192) are scored by several host threads at once;
 * score_mode    0: a dense stream's shards (contexts of more than 16 buffers) are scored on their devices; 1: never;
 *               2: scored, but the collector refuses the result and fetches the records instead (the path a filter
 *               table about to fill up takes). */
int adsb_multi_selftest_tune(adsb_multi *m, uint32_t fresh_cap, uint32_t parallel_min, uint32_t score_mode);
/* Fault injection for the tests of "When a capture fails": shard `shard` of the capture submitted `captures_from_now`
 * submissions after this call (0 = the next one) fails in the given way; kind 0 disarms.
 *   ADSB_FAULT_PHASE1   its first phase is refused as if the launch had failed (ADSB_ERR_HIP);
 *   ADSB_FAULT_PHASE2   ... its second phase;
 *   ADSB_FAULT_HANG     its first phase is treated as never finishing: the timeout path, the device is given up;
 *   ADSB_FAULT_RECORDS  its records are refused when the second phase has landed, as if their checksum had failed. */
#define ADSB_FAULT_PHASE1 1
#define ADSB_FAULT_PHASE2 2
#define ADSB_FAULT_HANG 3
#define ADSB_FAULT_RECORDS 4
int adsb_multi_selftest_fail(adsb_multi *m, uint32_t captures_from_now, int shard, int kind);
/* ... and what the shards did so far: out8[0] shards whose records the host had to put in order, [1] shards that took
 * the fresh_cap fallback, [2] shards whose second phase ordered its records on the device (a dense stream's), [3] captures
 * whose records were scored by several host threads at once, [4] shards scored on their device, [5] such results the
 * collector used, [6] ... and refused; [7] 1 while the handle is poisoned. */
int adsb_multi_selftest_counters(const adsb_multi *m, uint64_t *out8);

/* Host only, no context: the ordered replay done by several threads at once, as adsb_multi_collect does it for captures
 * of tens of thousands of records (csrc/adsb_replay_host.h: ParallelReplay) -- the records are put in order, cut into
 * `runs` runs at buffer boundaries (the shards), planned into `parts` parts and scanned / scored by `threads` threads.
 * Same arguments and results as adsb_replay_records; *went_parallel = 0 when the plan was refused (too few records, a
 * filter table that could fill up) and the records were replayed serially.  For the tests that pin it to the serial
 * replay and to the oracle. */
int adsb_selftest_parallel_replay(uint32_t *filter_table, const adsb_trial *records, size_t n, int runs, int parts, int threads,
                                  adsb_msg *out, size_t cap, size_t *n_out, int *went_parallel);

/* The 256-entry CRC-24 table the host replay scores with (src/crc.rs:3-260 CRC_TABLE): for the test that
 * pins it against the reference's constants.  Host only, no context. */
int adsb_selftest_crc_table(uint32_t *out256);

/* Host only, no context: the address exchange of a sharded capture as the library does it -- the 24-bit
 * addresses the replay of `records` can add to the filter (clean DF11 with IID 0, DF17: src/mode_s/mod.rs:80-84,
 * 97-99), sorted, without duplicates and without those in `known` (any order).  For the tests that run the
 * host-only code under sanitizers. */
int adsb_selftest_learned_union(const adsb_trial *records, size_t n, const uint32_t *known, size_t n_known,
                                uint32_t *out, size_t cap, size_t *n_out);

int adsb_get_stats(const adsb_ctx *ctx, adsb_stats *out);
/* Diagnostic: how many collected passes handed the host their trial records out of
 * (buffer, j, try_phase) order, so that the host replay had to sort them first.  Passes of more
 * than 16 buffers are put in order on the device; small passes and the overflow fallback are not. */
uint64_t adsb_host_sorts(const adsb_ctx *ctx);
/* Diagnostic: how many collected passes the host scored itself (the ordered replay of
 * score_modes_message / best-of-5, src/mode_s/mod.rs:34-139, src/demod_2400.rs:184-207) instead of
 * taking the messages the device scored.  Passes of more than 16 buffers in a steady pipeline are
 * scored on the device against a device-resident copy of the ICAO filter; small passes, fallbacks,
 * a filter close to its 4096 entries and the passes in flight behind any of those are not. */
uint64_t adsb_host_replays(const adsb_ctx *ctx);
/* Diagnostic: how many passes of a few buffers were run a second time.  Such a pass is a single launch
 * that does not wait for the passes still in flight beside it; when one of those turns out to have added
 * a NEW address to the filter (src/icao_filter.rs:46-62) -- rare once a receiver has seen the aircraft
 * around it -- the later pass's address/parity trials may have been matched too early, and it is redone
 * behind everything else so that the result stays the reference's. */
uint64_t adsb_host_rematches(const adsb_ctx *ctx);
const char *adsb_strerror(int status);
/* Text of the last HIP failure on this context ("" if none). */
const char *adsb_last_error(const adsb_ctx *ctx);
/* Library / kernel generation string, e.g. "adsb_hip 0.1 gfx950". */
const char *adsb_version(void);

#ifdef __cplusplus
}
#endif
#endif
