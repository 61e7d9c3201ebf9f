/* tests/abi_host.c -- the drop-in boundary from a COMPILED host, no Python and no ctypes in between.
 *
 * Plain C over include/adsb_hip.h, doing exactly what the reference's own test routine does
 * (rsadsb/dump1090_rs tests/test.rs:7-17):
 *
 *     icao_filter::icao_flush();
 *     let buf = utils::read_test_data(filename);
 *     let outbuf = utils::to_mag(&buf);
 *     let data = demod_2400::demodulate2400(&outbuf).unwrap();
 *     for (a, b) in data.iter().zip(expected_data.iter()) { assert_eq_hex!(a.buffer(), b); }
 *
 * -- but exact in count and order (the reference's zip() only checks a prefix).
 *
 *     abi_host <capture.iq> <expected hex frame> ...
 *
 * Exit status 0: the frames are exactly the expected ones.  Built with gcc by __graft_entry__.build()
 * and run by tests/test_gpu_parity.py on the GPU box with the frames of tests/golden/reference_frames.json
 * (= tests/test.rs:22-28,35-40,49-56).  What a Rust host's `extern "C"` block binds is this same header.
 */
#define _POSIX_C_SOURCE 200809L   /* clock_gettime under -std=c11 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <stddef.h>

#include "adsb_hip.h"

/* The layouts a non-C host binds (integration/rust/src/hip_ffi.rs, dump1090_rs_amd/_lib.py): pinned here
 * from the C side, so a change to a struct in the header fails this build; tests/test_rust_shim.py
 * checks the Rust and ctypes declarations against the same numbers. */
_Static_assert(sizeof(adsb_msg) == 40, "adsb_msg is 40 bytes");
_Static_assert(offsetof(adsb_msg, msg) == 0 && offsetof(adsb_msg, len) == 14 && offsetof(adsb_msg, try_phase) == 15 &&
               offsetof(adsb_msg, score) == 16 && offsetof(adsb_msg, j) == 20 && offsetof(adsb_msg, chunk) == 24 &&
               offsetof(adsb_msg, signal_level) == 32, "adsb_msg field offsets");
_Static_assert(sizeof(adsb_trial) == 32, "adsb_trial is 32 bytes");
_Static_assert(offsetof(adsb_trial, power) == 0 && offsetof(adsb_trial, chunk) == 8 && offsetof(adsb_trial, j_tp) == 12 &&
               offsetof(adsb_trial, msg) == 16 && offsetof(adsb_trial, pad) == 30, "adsb_trial field offsets");
_Static_assert(sizeof(adsb_stats) == 72, "adsb_stats is 72 bytes");
_Static_assert(offsetof(adsb_stats, n_messages) == 40 && offsetof(adsb_stats, ms_scan) == 48 &&
               offsetof(adsb_stats, retries) == 64 && offsetof(adsb_stats, ms_scan_exclusive) == 68, "adsb_stats field offsets");
_Static_assert(sizeof(adsb_multi_stats) == 96, "adsb_multi_stats is 96 bytes");
_Static_assert(offsetof(adsb_multi_stats, n_addrs_exchanged) == 48 && offsetof(adsb_multi_stats, n_devices) == 56 &&
               offsetof(adsb_multi_stats, retries) == 60 && offsetof(adsb_multi_stats, ms_wall) == 64 &&
               offsetof(adsb_multi_stats, ms_replay) == 88, "adsb_multi_stats field offsets");
_Static_assert(ADSB_OK == 0 && ADSB_ERR_INVALID == -1 && ADSB_ERR_NO_DEVICE == -2 && ADSB_ERR_HIP == -3 && ADSB_ERR_TOO_LONG == -4 &&
               ADSB_ERR_CAPACITY == -5 && ADSB_ERR_NOMEM == -6 && ADSB_ERR_BUSY == -7, "status codes");
_Static_assert(ADSB_MAG_DATA_LEN == 131398 && ADSB_MAX_IN_FLIGHT == 4 && ADSB_MAX_IN_FLIGHT_SMALL == 8, "buffer geometry");

static int check_frames(const adsb_msg *msgs, size_t n, int n_want, char **want_hex)
{
    int rc = (n == (size_t)n_want) ? 0 : 1;
    for (size_t i = 0; i < n; i++) {
        char hex[2 * ADSB_MODES_LONG_MSG_BYTES + 1];
        for (unsigned k = 0; k < msgs[i].len; k++) sprintf(hex + 2 * k, "%02x", msgs[i].msg[k]);  /* buffer() */
        const char *want = i < (size_t)n_want ? want_hex[i] : "(none)";
        const int same = strcmp(hex, want) == 0;
        printf("%s %s\n", hex, same ? "ok" : want);
        if (!same) rc = 1;
    }
    if (n != (size_t)n_want) fprintf(stderr, "%zu frames, %d expected\n", n, n_want);
    return rc;
}

/* abi_host --multi N <capture.iq> frames...: the same capture as ONE capture over N contexts on device 0 through
 * adsb_multi_* -- the single-process multi-GPU entry points from a compiled host, no Python, no process group:
 * each of the N shards holds a contiguous share of the 131072 samples... of a capture made of N copies of the
 * file, so every shard has whole buffers: the frames of the file come out N times, buffer after buffer. */
static int run_multi(int n_dev, int argc, char **argv)
{
    int devices[64];
    if (n_dev < 1 || n_dev > 64 || argc < 1) return 2;
    for (int k = 0; k < n_dev; k++) devices[k] = 0;
    adsb_multi *m = NULL;
    int st = adsb_multi_create(&m, devices, n_dev, 2);
    if (st != ADSB_OK) {
        fprintf(stderr, "adsb_multi_create: %s\n", adsb_strerror(st));
        return 3;
    }
    const size_t per = ADSB_MODES_MAG_BUF_SAMPLES;
    int16_t *iq = malloc(per * 2 * sizeof(int16_t) * (size_t)n_dev);
    adsb_msg *msgs = malloc(sizeof(adsb_msg) * 256 * (size_t)n_dev);
    size_t n_iq = 0, n = 0;
    int rc = 1;
    if (!iq || !msgs) goto out;
    st = adsb_read_test_data(argv[0], iq, per, &n_iq);
    if (st != ADSB_OK || n_iq != per) goto out;
    for (int k = 1; k < n_dev; k++) memcpy(iq + 2 * per * (size_t)k, iq, per * 2 * sizeof(int16_t));
    st = adsb_multi_icao_flush(m);
    if (st == ADSB_OK) st = adsb_multi_demod_iq(m, iq, per * (size_t)n_dev, msgs, 256 * (size_t)n_dev, &n);
    if (st != ADSB_OK) {
        fprintf(stderr, "%s (%s)\n", adsb_strerror(st), adsb_multi_last_error(m));
        goto out;
    }
    /* the first copy of the file: exactly the reference's frames (tests/test.rs), in buffer 0 */
    size_t n0 = 0;
    while (n0 < n && msgs[n0].chunk == 0) n0++;
    rc = check_frames(msgs, n0, argc - 1, argv + 1);
    /* every later copy is scored by a filter that already knows the capture's aircraft: at least as many frames,
     * in ascending buffer order */
    for (size_t i = 1; i < n; i++)
        if (msgs[i].chunk < msgs[i - 1].chunk || msgs[i].chunk >= (uint64_t)n_dev) rc = 1;
    adsb_multi_stats ms;
    if (adsb_multi_get_stats(m, &ms) != ADSB_OK || ms.n_devices != (uint32_t)n_dev || ms.n_messages != n) rc = 1;
    printf("multi: %d devices, %zu frames, %llu records, %llu addresses exchanged\n", n_dev, n,
           (unsigned long long)ms.n_records, (unsigned long long)ms.n_addrs_exchanged);
    /* the asynchronous host form out of pinned memory: the same capture three times in flight behind one flush each --
     * adsb_multi_host_alloc / adsb_multi_submit_iq / adsb_multi_collect; every collect gives the blocking call's list */
    {
        void *pinned = NULL;
        const size_t bytes = per * 2 * sizeof(int16_t) * (size_t)n_dev;
        adsb_msg *again = malloc(sizeof(adsb_msg) * 256 * (size_t)n_dev);
        if (!again || adsb_multi_host_alloc(m, bytes, &pinned) != ADSB_OK) rc = 1;
        else {
            memcpy(pinned, iq, bytes);
            for (int k = 0; k < 3 && rc == 0; k++)
                if (adsb_multi_icao_flush(m) != ADSB_OK || adsb_multi_submit_iq(m, pinned, per * (size_t)n_dev) != ADSB_OK) rc = 1;
            if (rc == 0 && (adsb_multi_pending(m) != 3 || adsb_multi_host_free(m, pinned) != ADSB_ERR_BUSY)) rc = 1;
            for (int k = 0; k < 3 && rc == 0; k++) {
                size_t n2 = 0;
                if (adsb_multi_collect(m, again, 256 * (size_t)n_dev, &n2) != ADSB_OK || n2 != n ||
                    memcmp(again, msgs, n * sizeof(adsb_msg)) != 0)
                    rc = 1;
            }
            if (rc == 0 && adsb_multi_host_free(m, pinned) != ADSB_OK) rc = 1;
            if (rc == 0) printf("multi: three pinned host captures in flight, each equal to the blocking call\n");
        }
        /* "When a capture fails" (include/adsb_hip.h) as a compiled host meets it: the last shard of the second of three
         * captures in flight fails -> the first is fine, the second returns the shard's error, the third and a further
         * submission ADSB_ERR_POISONED, the restart with nothing in flight succeeds and the stream gives the blocking
         * call's list again; all under the blocking wait policy */
        if (rc == 0 && again) {
            size_t n2 = 0;
            if (adsb_multi_set_wait(m, ADSB_WAIT_BLOCK) != ADSB_OK || adsb_multi_get_wait(m) != ADSB_WAIT_BLOCK) rc = 1;
            if (adsb_multi_selftest_fail(m, 1, n_dev - 1, ADSB_FAULT_PHASE2) != ADSB_OK) rc = 1;
            for (int k = 0; k < 3 && rc == 0; k++)
                if (adsb_multi_icao_flush(m) != ADSB_OK || adsb_multi_submit_iq(m, iq, per * (size_t)n_dev) != ADSB_OK) rc = 1;
            if (rc == 0 && (adsb_multi_collect(m, again, 256 * (size_t)n_dev, &n2) != ADSB_OK || n2 != n ||
                            memcmp(again, msgs, n * sizeof(adsb_msg)) != 0))
                rc = 1;
            if (rc == 0 && adsb_multi_collect(m, again, 256 * (size_t)n_dev, &n2) != ADSB_ERR_HIP) rc = 1;
            if (rc == 0 && !strstr(adsb_multi_last_error(m), "injected")) rc = 1;
            if (rc == 0 && adsb_multi_submit_iq(m, iq, per * (size_t)n_dev) != ADSB_ERR_POISONED) rc = 1;
            if (rc == 0 && adsb_multi_icao_flush(m) != ADSB_ERR_BUSY) rc = 1;
            if (rc == 0 && adsb_multi_collect(m, again, 256 * (size_t)n_dev, &n2) != ADSB_ERR_POISONED) rc = 1;
            if (rc == 0 && (adsb_multi_pending(m) != 0 || adsb_multi_icao_flush(m) != ADSB_OK)) rc = 1;
            if (rc == 0 && (adsb_multi_demod_iq(m, iq, per * (size_t)n_dev, again, 256 * (size_t)n_dev, &n2) != ADSB_OK || n2 != n ||
                            memcmp(again, msgs, n * sizeof(adsb_msg)) != 0))
                rc = 1;
            if (rc == 0) printf("multi: a failed shard poisoned the handle, the restart gave the blocking call's list again\n");
        }
        free(again);
    }
out:
    free(iq);
    free(msgs);
    adsb_multi_destroy(m);
    return rc;
}

/* abi_host --live <stream.bin> <frames.out>: the live receiver's loop (dump1090_rs/src/main.rs:154-167) from a compiled
 * host -- one read of 131072 samples into a pinned ring slot (here: a memcpy out of the file's image, {re, im} pairs in
 * memory order), one demodulation, for ever, the ICAO filter never flushed, every slot in flight.  Writes the frames
 * (buffer index, j, try_phase, score, hex, signal_level bits) to frames.out and prints the loop's own time: what
 * bench.py's live_receiver leg reports and compares with the oracle's ONE stream over the same bytes. */
static int run_live(const char *path, const char *out_path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    const size_t per = ADSB_MODES_MAG_BUF_SAMPLES, passes = (size_t)bytes / (per * 4);
    int16_t *iq = malloc((size_t)bytes);
    if (!iq || fread(iq, 1, (size_t)bytes, f) != (size_t)bytes || passes == 0) return 2;
    fclose(f);
    adsb_ctx *ctx = NULL;
    int st = adsb_create(&ctx, 0, 1);
    if (st == ADSB_OK) st = adsb_ring_create(ctx, per);
    if (st == ADSB_OK) st = adsb_icao_flush(ctx);
    if (st != ADSB_OK) {
        fprintf(stderr, "%s\n", adsb_strerror(st));
        return 3;
    }
    const int depth = adsb_max_in_flight(ctx);
    const size_t cap = 4096;
    adsb_msg *msgs = malloc(sizeof(adsb_msg) * cap), *all = malloc(sizeof(adsb_msg) * cap * 64);
    size_t n_all = 0, done = 0;
    int rc = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t b = 0; b <= passes && rc == 0; b++) {
        /* finish the oldest pass when every slot is out, and everything at the end */
        while (rc == 0 && done < b && (b == passes || b - done >= (size_t)depth)) {
            size_t n = 0;
            if (adsb_collect(ctx, msgs, cap, &n) != ADSB_OK || n_all + n > cap * 64) rc = 1;
            for (size_t i = 0; i < n && rc == 0; i++) {
                msgs[i].chunk = done;
                all[n_all++] = msgs[i];
            }
            done++;
        }
        if (b == passes) break;
        int16_t *slot = NULL;
        size_t slot_cap = 0;
        if (adsb_ring_acquire(ctx, &slot, &slot_cap) != ADSB_OK || slot_cap < per) rc = 1;
        else {
            memcpy(slot, iq + 2 * per * b, per * 4);   /* (an SDR driver's read would land here) */
            if (adsb_ring_submit(ctx, per) != ADSB_OK) rc = 1;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    FILE *o = rc == 0 ? fopen(out_path, "w") : NULL;
    if (!o) rc = 1;
    for (size_t i = 0; i < n_all && rc == 0; i++) {
        unsigned long long bits;
        memcpy(&bits, &all[i].signal_level, 8);
        fprintf(o, "%llu %u %u %d ", (unsigned long long)all[i].chunk, all[i].j, all[i].try_phase, all[i].score);
        for (int k = 0; k < all[i].len; k++) fprintf(o, "%02x", all[i].msg[k]);
        fprintf(o, " %016llx\n", bits);
    }
    if (o) fclose(o);
    if (rc == 0) printf("live: %zu passes, %zu frames, %.6f s, %d in flight\n", passes, n_all, secs, depth);
    adsb_destroy(ctx);
    free(iq);
    free(msgs);
    free(all);
    return rc;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s [--multi N | --live stream.bin frames.out] <capture.iq> [expected hex frames...]\n", argv[0]);
        return 2;
    }
    if (strcmp(argv[1], "--multi") == 0) return argc < 4 ? 2 : run_multi(atoi(argv[2]), argc - 3, argv + 3);
    if (strcmp(argv[1], "--live") == 0) return argc < 4 ? 2 : run_live(argv[2], argv[3]);
    adsb_ctx *ctx = NULL;
    int st = adsb_create(&ctx, 0, 1);
    if (st != ADSB_OK) {
        fprintf(stderr, "adsb_create: %s\n", adsb_strerror(st));
        return 3;
    }
    int16_t *iq = malloc((size_t)ADSB_MODES_MAG_BUF_SAMPLES * 2 * sizeof(int16_t));
    uint16_t *data = malloc((size_t)ADSB_MAG_DATA_LEN * sizeof(uint16_t));
    adsb_msg msgs[256];
    size_t n_iq = 0, length = 0, n = 0;
    int rc = 1;
    if (!iq || !data) goto out;

    st = adsb_icao_flush(ctx);                                                       /* test.rs:9  */
    if (st == ADSB_OK) st = adsb_read_test_data(argv[1], iq, ADSB_MODES_MAG_BUF_SAMPLES, &n_iq); /* :10 */
    if (st == ADSB_OK) st = adsb_to_mag(ctx, iq, n_iq, data, &length);               /* test.rs:11 */
    if (st == ADSB_OK) st = adsb_demodulate2400(ctx, data, length, msgs, 256, &n);   /* test.rs:12 */
    if (st != ADSB_OK) {
        fprintf(stderr, "%s (%s)\n", adsb_strerror(st), adsb_last_error(ctx));
        goto out;
    }
    rc = check_frames(msgs, n, argc - 2, argv + 2);
out:
    free(iq);
    free(data);
    adsb_destroy(ctx);
    return rc;
}
