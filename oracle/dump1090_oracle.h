/*
 * oracle/dump1090_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the demod_2400 hot path of
 * rsadsb/dump1090_rs v0.8.1 (reference @ /root/reference).  It exists so the
 * HIP path can be checked bit-for-bit; it is NOT part of the product.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this oracle
 * against every known-answer vector the reference holds for the path
 * (reference tests/test.rs:19-59, the three test_iq fixtures -> 16 frames,
 * exact count and order).  The reference is Rust and cannot be compiled in
 * this image (no cargo/rustc), so there is no oracle/_ref build.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef DUMP1090_ORACLE_H
#define DUMP1090_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/lib.rs:22-26 */
#define ORC_MODES_MAG_BUF_SAMPLES 131072
#define ORC_TRAILING_SAMPLES 326
#define ORC_MODES_LONG_MSG_BYTES 14
#define ORC_MODES_SHORT_MSG_BYTES 7
#define ORC_MAG_DATA_LEN (ORC_TRAILING_SAMPLES + ORC_MODES_MAG_BUF_SAMPLES)

/* src/icao_filter.rs:5-6 */
#define ORC_ICAO_FILTER_SIZE 4096u
#define ORC_ICAO_FILTER_ADSB_NT (1u << 25)

/* src/lib.rs:30-34 MagnitudeBuffer */
typedef struct {
    uint16_t data[ORC_MAG_DATA_LEN];
    size_t length;
    size_t first_sample_timestamp_12mhz;
} orc_magbuf;

/* src/icao_filter.rs:8-9: the two process-global tables, made per-context */
typedef struct {
    uint32_t a[ORC_ICAO_FILTER_SIZE];
    uint32_t b[ORC_ICAO_FILTER_SIZE];
} orc_filter;

/* src/demod_2400.rs:92-102 ModeSMessage, plus provenance (chunk, j, try_phase) */
typedef struct {
    uint8_t msg[ORC_MODES_LONG_MSG_BYTES];
    uint8_t len;       /* 7 or 14 == buffer().len(), demod_2400.rs:106-111 */
    uint8_t try_phase; /* 4..8, the winning phase */
    int32_t score;
    uint32_t j;        /* index into MagnitudeBuffer.data */
    uint64_t chunk;    /* which 131072-sample buffer (0 for single-buffer calls) */
    double signal_level;
} orc_msg;

/* per-buffer stage statistics */
typedef struct {
    uint64_t preamble_pass; /* check_preamble returned Some */
    uint64_t snr_pass;      /* ... and passed the 3.5 dB gate */
    uint64_t quiet_pass;    /* ... and the quiet gate (= j that get sliced) */
    uint64_t trials;        /* scored trial messages (5 per sliced j) */
    uint64_t frames;        /* emitted */
} orc_stats;

/* src/icao_filter.rs */
void orc_icao_flush(orc_filter *f);                 /* :11-17 */
uint32_t orc_icao_hash(uint32_t a32);               /* :19-43 */
void orc_icao_filter_add(orc_filter *f, uint32_t addr);  /* :46-62 */
int orc_icao_filter_test(const orc_filter *f, uint32_t addr); /* :65-97 */

/* src/crc.rs */
uint32_t orc_crc_table_entry(unsigned i);           /* CRC_TABLE :3-260, regenerated */
uint32_t orc_modes_checksum(const uint8_t *message, size_t bits); /* :263-282 */

/* src/mode_s/mod.rs */
size_t orc_getbits(const uint8_t *data, size_t firstbit_1idx, size_t lastbit_1idx); /* :14-30 */
/* :34-139; returns 0 for None, else 1 and fills *msglen (7|14) and *score */
int orc_score_modes_message(orc_filter *f, const uint8_t *msg, size_t msg_len,
                            int *msglen, int32_t *score);

/* src/utils.rs:43-58 to_mag; iq is in-memory Complex<i16> order {re, im} per sample.
 * n must be <= 131072 (the reference panics beyond, lib.rs:48): returns -1 then. */
int orc_to_mag(const int16_t *iq_re_im, size_t n, orc_magbuf *out);
/* one sample of the above: the exact f32 pipeline */
uint16_t orc_mag_sample(int16_t re, int16_t im);
/* the part of it after mag_sqr (utils.rs:54-55) */
uint16_t orc_mag_from_sqr(float mag_sqr);
/* digest of orc_mag_from_sqr(X * 2^-30) over consecutive f32 bit patterns of X */
uint64_t orc_mag_x_digest(uint32_t first_bits, uint32_t count, uint64_t *xor_out);

/* src/demod_2400.rs:215-321; returns 0 for None */
int orc_check_preamble(const uint16_t *p14, int32_t *high, uint32_t *base_signal,
                       uint32_t *base_noise);
/* src/demod_2400.rs:158-182: slice the 14 bytes for one try_phase at preamble j */
void orc_slice_phase(const uint16_t *data, size_t j, int try_phase, uint8_t msg[14]);

/* src/demod_2400.rs:115-212.  Returns number of messages found; writes at most cap.
 * chunk is copied into each message.  stats may be NULL. */
size_t orc_demodulate2400(orc_filter *f, const orc_magbuf *mag, uint64_t chunk,
                          orc_msg *out, size_t cap, orc_stats *stats);

/* same layout as adsb_trial (include/adsb_hip.h) */
typedef struct {
    uint64_t power;
    uint32_t chunk;
    uint32_t j_tp; /* j | try_phase << 24 */
    uint8_t msg[ORC_MODES_LONG_MSG_BYTES];
    uint16_t pad;
} orc_trial;
/* all 5 trials of every gate-passing j, unscored, in (j, try_phase) order */
size_t orc_all_trials(const orc_magbuf *mag, uint64_t chunk, orc_trial *out, size_t cap);

/* The benches/demod_benchmark.rs:7-12 / main.rs:166-167 composition over a long
 * IQ stream: split into 131072-sample buffers, to_mag + demodulate2400 each, the
 * filter persisting across buffers.  Returns total messages found. */
size_t orc_demod_iq(orc_filter *f, const int16_t *iq_re_im, size_t n_samples,
                    orc_msg *out, size_t cap, orc_stats *stats);

/* NOT reference behaviour -- the opt-in "carry-over" extension of SURVEY.md 8(f)-3, restated
 * here only so that the product's carry-over mode has a checker.  As orc_demod_iq, but the
 * 326-sample lead-in of every buffer (src/lib.rs:24,36-44 leaves it zero) holds the
 * magnitudes of the 326 samples that preceded the buffer in the stream, as upstream C
 * dump1090 does, so frames that straddle a buffer edge are found (in the later buffer).
 * `carry` is the stream state: the last 326 IQ samples seen (652 int16, {re,im} pairs,
 * zero-initialised by the caller), read for the first buffer and updated on return. */
size_t orc_demod_iq_carry(orc_filter *f, const int16_t *iq_re_im, size_t n_samples,
                          orc_msg *out, size_t cap, orc_stats *stats, int16_t *carry);

/* orc_demod_iq over `threads` host threads (dump1090_oracle_mt.c): workers run to_mag, the
 * gates and the slicer per buffer, one thread replays the trials in order through the
 * filter.  Same result as orc_demod_iq; stats->preamble_pass / snr_pass count only the
 * positions that were sliced. */
size_t orc_demod_iq_mt(orc_filter *f, const int16_t *iq_re_im, size_t n_samples,
                       orc_msg *out, size_t cap, orc_stats *stats, int threads);

/* src/utils.rs:23-40 read_test_data: file order is [im][re] little-endian i16;
 * writes in-memory {re, im} pairs.  Returns samples read or -1. */
long orc_read_test_data(const char *path, int16_t *iq_re_im, size_t max_samples);

#ifdef __cplusplus
}
#endif
#endif
