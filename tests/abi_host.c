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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
_Static_assert(ADSB_OK == 0 && ADSB_ERR_INVALID == -1 && ADSB_ERR_NO_DEVICE == -2 && ADSB_ERR_HIP == -3 && ADSB_ERR_TOO_LONG == -4 &&
               ADSB_ERR_CAPACITY == -5 && ADSB_ERR_NOMEM == -6 && ADSB_ERR_BUSY == -7, "status codes");
_Static_assert(ADSB_MAG_DATA_LEN == 131398 && ADSB_MAX_IN_FLIGHT == 4 && ADSB_MAX_IN_FLIGHT_SMALL == 8, "buffer geometry");

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s <capture.iq> [expected hex frames...]\n", argv[0]);
        return 2;
    }
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
    rc = (n == (size_t)(argc - 2)) ? 0 : 1;
    for (size_t i = 0; i < n; i++) {
        char hex[2 * ADSB_MODES_LONG_MSG_BYTES + 1];
        for (unsigned k = 0; k < msgs[i].len; k++) sprintf(hex + 2 * k, "%02x", msgs[i].msg[k]);  /* buffer() */
        const char *want = i + 2 < (size_t)argc ? argv[i + 2] : "(none)";
        const int same = strcmp(hex, want) == 0;
        printf("%s %s\n", hex, same ? "ok" : want);
        if (!same) rc = 1;
    }
    if (n != (size_t)(argc - 2)) fprintf(stderr, "%zu frames, %d expected\n", n, argc - 2);
out:
    free(iq);
    free(data);
    adsb_destroy(ctx);
    return rc;
}
