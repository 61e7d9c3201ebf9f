/*
 * oracle/dump1090_oracle.c -- TEST INFRASTRUCTURE ONLY (see dump1090_oracle.h).
 *
 * Plain-C CPU restatement of rsadsb/dump1090_rs v0.8.1's demod_2400 hot path.
 * Written from the behaviour of the cited reference lines, not translated
 * line by line: the bit slicer uses the closed form of the Phase state
 * machine, the CRC table is regenerated from the polynomial, the filter is
 * per-context instead of process-global.
 *
 * Build with -ffp-contract=off: the float pipeline in orc_mag_sample must not
 * be re-fused by the compiler.
 */
#include "dump1090_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* icao_filter (src/icao_filter.rs)                                    */
/* ------------------------------------------------------------------ */

/* src/icao_filter.rs:11-17 */
void orc_icao_flush(orc_filter *f) { memset(f, 0, sizeof(*f)); }

/* src/icao_filter.rs:19-43: Jenkins one-at-a-time over the 3 low bytes, with
 * u64 intermediates, truncated to u32 and masked to the table size. */
uint32_t orc_icao_hash(uint32_t a32)
{
    uint64_t a = a32, h = 0;
    for (int byte = 0; byte < 3; byte++) {
        h += (a >> (8 * byte)) & 0xff;
        h += h << 10;
        h ^= h >> 6;
    }
    h += h << 3;
    h ^= h >> 11;
    h += h << 15;
    return (uint32_t)h & (ORC_ICAO_FILTER_SIZE - 1);
}

/* src/icao_filter.rs:46-62 */
void orc_icao_filter_add(orc_filter *f, uint32_t addr)
{
    uint32_t h0 = orc_icao_hash(addr), h = h0;
    while (f->a[h] != 0 && f->a[h] != addr) {
        h = (h + 1) & (ORC_ICAO_FILTER_SIZE - 1);
        if (h == h0)
            return; /* table full (:52-55: message on stderr, no insert) */
    }
    if (f->a[h] == 0)
        f->a[h] = addr;
}

/* src/icao_filter.rs:65-97: probe table A, then table B (B is never written
 * except by flush, :9,:15-16).  An empty slot compares equal to addr 0. */
static int probe(const uint32_t *t, uint32_t addr)
{
    uint32_t h0 = orc_icao_hash(addr), h = h0;
    while (t[h] != 0 && t[h] != addr) {
        h = (h + 1) & (ORC_ICAO_FILTER_SIZE - 1);
        if (h == h0)
            break;
    }
    return t[h] == addr;
}

int orc_icao_filter_test(const orc_filter *f, uint32_t addr)
{
    return probe(f->a, addr) || probe(f->b, addr);
}

/* ------------------------------------------------------------------ */
/* crc (src/crc.rs)                                                    */
/* ------------------------------------------------------------------ */

/* CRC_TABLE (src/crc.rs:3-260) regenerated: entry i is i<<16 pushed through 8
 * MSB-first steps of the Mode-S generator 0xFFF409, kept to 24 bits.
 * Pinned by T[1]=0xFFF409 (:5) and T[255]=0xFA0480 (:259) in the tests. */
static uint32_t g_crc_table[256];
static int g_crc_ready;

static void crc_init(void)
{
    for (unsigned i = 0; i < 256; i++) {
        uint32_t c = (uint32_t)i << 16;
        for (int k = 0; k < 8; k++)
            c = (c & 0x800000) ? ((c << 1) ^ 0xFFF409u) : (c << 1);
        g_crc_table[i] = c & 0xFFFFFF;
    }
    g_crc_ready = 1;
}

uint32_t orc_crc_table_entry(unsigned i)
{
    if (!g_crc_ready)
        crc_init();
    return g_crc_table[i & 255];
}

/* src/crc.rs:263-282 */
uint32_t orc_modes_checksum(const uint8_t *message, size_t bits)
{
    if (!g_crc_ready)
        crc_init();
    size_t n = bits / 8;
    uint32_t rem = 0;
    for (size_t i = 0; i + 3 < n; i++)
        rem = ((rem << 8) ^ g_crc_table[message[i] ^ ((rem >> 16) & 0xff)]) & 0xFFFFFF;
    rem ^= (uint32_t)message[n - 3] << 16 | (uint32_t)message[n - 2] << 8 | message[n - 1];
    return rem;
}

/* ------------------------------------------------------------------ */
/* mode_s (src/mode_s/mod.rs)                                          */
/* ------------------------------------------------------------------ */

/* src/mode_s/mod.rs:14-30: bits are numbered from 1, MSB of byte 0 first */
size_t orc_getbits(const uint8_t *data, size_t firstbit_1idx, size_t lastbit_1idx)
{
    size_t ans = 0;
    for (size_t b = firstbit_1idx - 1; b <= lastbit_1idx - 1; b++)
        ans = (ans << 1) | ((data[b >> 3] >> (7 - (b & 7))) & 1u);
    return ans;
}

/* src/mode_s/mod.rs:34-139 */
int orc_score_modes_message(orc_filter *f, const uint8_t *msg, size_t msg_len, int *msglen,
                            int32_t *score)
{
    size_t validbits = msg_len * 8;
    if (validbits < ORC_MODES_SHORT_MSG_BYTES * 8) /* :37-39 */
        return 0;

    unsigned df = (unsigned)orc_getbits(msg, 1, 5);                                /* :41 */
    size_t msgbits = (df & 0x10) ? ORC_MODES_LONG_MSG_BYTES * 8 : ORC_MODES_SHORT_MSG_BYTES * 8;
    if (validbits < msgbits) /* :48-50 */
        return 0;

    int all_zero = 1; /* :51-53: over the whole slice handed in, not just msgbits */
    for (size_t i = 0; i < msg_len; i++)
        if (msg[i]) {
            all_zero = 0;
            break;
        }
    if (all_zero)
        return 0;

    int32_t res;
    if (df == 0 || df == 4 || df == 5) { /* :56-72 address/parity, short */
        uint32_t crc = orc_modes_checksum(msg, msgbits);
        res = orc_icao_filter_test(f, crc) ? 1000 : -1;
    } else if (df == 11) { /* :73-90 all-call reply */
        uint32_t crc = orc_modes_checksum(msg, msgbits);
        uint32_t iid = crc & 0x7f;
        crc &= 0xFFFF80;
        uint32_t addr = (uint32_t)orc_getbits(msg, 9, 32);
        int known = orc_icao_filter_test(f, addr); /* evaluated before any add */
        if (crc != 0)
            res = -2;
        else if (iid == 0 && known)
            res = 1600;
        else if (iid == 0) {
            orc_icao_filter_add(f, addr);
            res = 750;
        } else
            res = known ? 1000 : -1;
    } else if (df == 17 || df == 18) { /* :91-109 extended squitter */
        uint32_t addr = (uint32_t)orc_getbits(msg, 9, 32);
        uint32_t crc = orc_modes_checksum(msg, msgbits);
        int known = orc_icao_filter_test(f, addr);
        if (crc != 0)
            res = -2;
        else if (known)
            res = 1800;
        else {
            orc_icao_filter_add(f, df == 17 ? addr : (addr | ORC_ICAO_FILTER_ADSB_NT));
            res = 1400;
        }
    } else if (df == 16 || df == 20 || df == 21 || df >= 24) { /* :110-135 */
        uint32_t crc = orc_modes_checksum(msg, ORC_MODES_LONG_MSG_BYTES * 8);
        res = orc_icao_filter_test(f, crc) ? 1000 : -2;
    } else {
        res = -2; /* :136 */
    }
    *msglen = (int)(msgbits / 8);
    *score = res;
    return 1;
}

/* ------------------------------------------------------------------ */
/* to_mag (src/utils.rs:43-58, src/lib.rs:36-51)                       */
/* ------------------------------------------------------------------ */

/* The tail of src/utils.rs:53-55 as a function of mag_sqr alone:
 * sqrt (correctly rounded), fused *65535 + 0.5, then Rust's `as u16`
 * (truncating, saturating, NaN -> 0). */
uint16_t orc_mag_from_sqr(float mag_sqr)
{
    float mag = sqrtf(mag_sqr);
    float o = fmaf(mag, 65535.0f, 0.5f);
    if (!(o > 0.0f))
        return 0;
    if (o >= 65535.0f)
        return 65535;
    return (uint16_t)o;
}

/* src/utils.rs:47-55.  fi comes from .im, fq from .re; the squares are not
 * symmetric: fq*fq is rounded on its own, fi*fi is fused into the add. */
uint16_t orc_mag_sample(int16_t re, int16_t im)
{
    float fi = (float)im / 32768.0f;
    float fq = (float)re / 32768.0f;
    float t = fq * fq;                   /* one rounded multiply (-ffp-contract=off) */
    return orc_mag_from_sqr(fmaf(fi, fi, t)); /* f32::mul_add = IEEE fused */
}

/* Digest of orc_mag_from_sqr over `count` consecutive f32 bit patterns starting at
 * `first_bits`, each scaled by 2^-30 (= the mag_sqr an integer-valued
 * X = im^2 + rn(re^2) stands for).  Lets a test sweep every representable X in
 * [0, 2^31] against the device's folded-constant form.  Returns the sum of the
 * outputs; *xor_out gets an order-independent hash. */
uint64_t orc_mag_x_digest(uint32_t first_bits, uint32_t count, uint64_t *xor_out)
{
    uint64_t sum = 0, h = 0;
    for (uint32_t i = 0; i < count; i++) {
        uint32_t bits = first_bits + i;
        float x;
        memcpy(&x, &bits, 4);
        uint16_t u = orc_mag_from_sqr(x * 0x1p-30f);
        sum += u;
        h ^= ((uint64_t)u + 1) * (2 * (uint64_t)bits + 1);
    }
    if (xor_out)
        *xor_out = h;
    return sum;
}

int orc_to_mag(const int16_t *iq_re_im, size_t n, orc_magbuf *out)
{
    if (n > ORC_MODES_MAG_BUF_SAMPLES)
        return -1; /* lib.rs:48 would index out of bounds and panic */
    memset(out, 0, sizeof(*out)); /* MagnitudeBuffer::default(), lib.rs:36-44 */
    for (size_t k = 0; k < n; k++) /* push(), lib.rs:47-50: lands at 326 + k */
        out->data[ORC_TRAILING_SAMPLES + k] = orc_mag_sample(iq_re_im[2 * k], iq_re_im[2 * k + 1]);
    out->length = n;
    return 0;
}

/* ------------------------------------------------------------------ */
/* demod_2400 (src/demod_2400.rs)                                      */
/* ------------------------------------------------------------------ */

/* src/demod_2400.rs:215-321.  Five overlapping patterns tried in order; the
 * first that matches decides (high, base_signal, base_noise).  high is always
 * the sum / 4, also when five or six terms are summed. */
int orc_check_preamble(const uint16_t *p, int32_t *high, uint32_t *base_signal,
                       uint32_t *base_noise)
{
    if (!(p[0] < p[1] && p[12] > p[13])) /* :221-224 */
        return 0;

    if (p[1] > p[2] && p[2] < p[3] && p[3] > p[4] && p[8] < p[9] && p[9] > p[10] &&
        p[10] < p[11]) { /* :227-241 peaks 1,3,9,11-12 */
        *high = ((int32_t)p[1] + p[3] + p[9] + p[11] + p[12]) / 4;
        *base_signal = (uint32_t)p[1] + p[3] + p[9];
        *base_noise = (uint32_t)p[5] + p[6] + p[7];
    } else if (p[1] > p[2] && p[2] < p[3] && p[3] > p[4] && p[8] < p[9] && p[9] > p[10] &&
               p[11] < p[12]) { /* :242-261 peaks 1,3,9,12 */
        *high = ((int32_t)p[1] + p[3] + p[9] + p[12]) / 4;
        *base_signal = (uint32_t)p[1] + p[3] + p[9] + p[12];
        *base_noise = (uint32_t)p[5] + p[6] + p[7] + p[8];
    } else if (p[1] > p[2] && p[2] < p[3] && p[4] > p[5] && p[8] < p[9] && p[10] > p[11] &&
               p[11] < p[12]) { /* :262-279 peaks 1,3-4,9-10,12 */
        *high = ((int32_t)p[1] + p[3] + p[4] + p[9] + p[10] + p[12]) / 4;
        *base_signal = (uint32_t)p[1] + p[12];
        *base_noise = (uint32_t)p[6] + p[7];
    } else if (p[1] > p[2] && p[3] < p[4] && p[4] > p[5] && p[9] < p[10] && p[10] > p[11] &&
               p[11] < p[12]) { /* :280-299 peaks 1,4,10,12 */
        *high = ((int32_t)p[1] + p[4] + p[10] + p[12]) / 4;
        *base_signal = (uint32_t)p[1] + p[4] + p[10] + p[12];
        *base_noise = (uint32_t)p[5] + p[6] + p[7] + p[8];
    } else if (p[2] > p[3] && p[3] < p[4] && p[4] > p[5] && p[9] < p[10] && p[10] > p[11] &&
               p[11] < p[12]) { /* :300-317 peaks 1-2,4,10,12 */
        *high = ((int32_t)p[1] + p[2] + p[4] + p[10] + p[12]) / 4;
        *base_signal = (uint32_t)p[4] + p[10] + p[12];
        *base_noise = (uint32_t)p[6] + p[7] + p[8];
    } else {
        return 0; /* :318-320 */
    }
    return 1;
}

/* src/demod_2400.rs:72-83 Phase::calculate_bit as a coefficient row per phase
 * (the fourth tap is only non-zero for Phase::Four) */
static const int32_t k_slice_coef[5][4] = {
    {5, -3, -2, 0}, {4, -1, -3, 0}, {3, 1, -4, 0}, {2, 3, -5, 0}, {1, 5, -5, -1}};

/* src/demod_2400.rs:158-182 with the Phase state machine (:22-70) in closed
 * form: every bit advances the 5x-oversampled position by 12 (2.4 samples),
 * starting at 5*(j+19) + try_phase; sample = position / 5, phase = position % 5.
 * (next(): phase += 2 mod 5 with the index stepping 2, or 3 on wrap-around;
 * next_start(): the byte-start phase += 1, the same thing as 8 bits * 12 = 96 =
 * 19*5 + 1.)  The furthest sample read is data[j + 290]. */
void orc_slice_phase(const uint16_t *data, size_t j, int try_phase, uint8_t msg[14])
{
    size_t s = j + 19 + (size_t)(try_phase / 5);
    int ph = try_phase % 5;
    for (int byte = 0; byte < 14; byte++) {
        unsigned acc = 0;
        for (int bit = 0; bit < 8; bit++) {
            const uint16_t *m = data + s;
            const int32_t *c = k_slice_coef[ph];
            int32_t v = c[0] * m[0] + c[1] * m[1] + c[2] * m[2] + c[3] * m[3];
            acc = (acc << 1) | (v > 0);
            ph += 2;
            s += 2;
            if (ph >= 5) {
                ph -= 5;
                s += 1;
            }
        }
        msg[byte] = (uint8_t)acc;
    }
}

/* src/demod_2400.rs:115-212 */
size_t orc_demodulate2400(orc_filter *f, const orc_magbuf *mag, uint64_t chunk, orc_msg *out,
                          size_t cap, orc_stats *stats)
{
    const uint16_t *data = mag->data;
    size_t found = 0;

    /* :120-125: skip_count is never set non-zero, so every j is examined */
    for (size_t j = 0; j < mag->length; j++) {
        int32_t high;
        uint32_t sig, noise;
        if (!orc_check_preamble(data + j, &high, &sig, &noise)) /* :127 */
            continue;
        if (stats)
            stats->preamble_pass++;
        if (sig * 2 < 3 * noise) /* :129-132, about 3.5 dB */
            continue;
        if (stats)
            stats->snr_pass++;
        /* :135-146 the "quiet" samples must stay below high */
        static const int quiet[9] = {5, 6, 7, 8, 14, 15, 16, 17, 18};
        int loud = 0;
        for (int q = 0; q < 9; q++)
            if ((int32_t)data[j + quiet[q]] >= high)
                loud = 1;
        if (loud)
            continue;
        if (stats)
            stats->quiet_pass++;

        /* :149-200 best of the five trial phases, strictly-greater wins */
        orc_msg best;
        memset(&best, 0, sizeof(best));
        best.score = -2;
        best.len = ORC_MODES_SHORT_MSG_BYTES;
        for (int try_phase = 4; try_phase < 9; try_phase++) {
            uint8_t msg[14];
            orc_slice_phase(data, j, try_phase, msg);
            int msglen;
            int32_t score;
            if (stats)
                stats->trials++;
            if (!orc_score_modes_message(f, msg, 14, &msglen, &score)) /* :184 */
                continue;
            if (score > best.score) { /* :185 */
                memcpy(best.msg, msg, 14);
                best.len = (uint8_t)msglen;
                best.score = score;
                best.try_phase = (uint8_t)try_phase;
                /* :191-198 signal_len = 14*12/5 = 33 samples from j+19 */
                uint64_t p = 0;
                size_t signal_len = 14 * 12 / 5;
                for (size_t k = 0; k < signal_len; k++)
                    p += (uint64_t)data[j + 19 + k] * data[j + 19 + k];
                double signal_power = (double)p / 65535.0 / 65535.0;
                best.signal_level = signal_power / (double)signal_len;
            }
        }
        if (best.score < 0) /* :203-205 */
            continue;
        best.j = (uint32_t)j;
        best.chunk = chunk;
        if (found < cap)
            out[found] = best;
        found++;
        if (stats)
            stats->frames++;
    }
    return found;
}

/* Every trial message of a buffer, unscored: for each j that passes the gates
 * (src/demod_2400.rs:127-146) the five sliced messages (:158-182) with the
 * 33-sample power (:191-196), in (j, try_phase) order.  Feeds the host-replay
 * test of the product library (same layout as adsb_trial). */
size_t orc_all_trials(const orc_magbuf *mag, uint64_t chunk, orc_trial *out, size_t cap)
{
    const uint16_t *data = mag->data;
    size_t n = 0;
    for (size_t j = 0; j < mag->length; j++) {
        int32_t high;
        uint32_t sig, noise;
        if (!orc_check_preamble(data + j, &high, &sig, &noise) || sig * 2 < 3 * noise)
            continue;
        static const int quiet[9] = {5, 6, 7, 8, 14, 15, 16, 17, 18};
        int loud = 0;
        for (int q = 0; q < 9; q++)
            if ((int32_t)data[j + quiet[q]] >= high)
                loud = 1;
        if (loud)
            continue;
        uint64_t p = 0;
        for (size_t k = 0; k < 33; k++)
            p += (uint64_t)data[j + 19 + k] * data[j + 19 + k];
        for (int tp = 4; tp < 9; tp++, n++) {
            if (n >= cap)
                continue;
            memset(&out[n], 0, sizeof(out[n]));
            out[n].power = p;
            out[n].chunk = (uint32_t)chunk;
            out[n].j_tp = (uint32_t)j | ((uint32_t)tp << 24);
            orc_slice_phase(data, j, tp, out[n].msg);
        }
    }
    return n;
}

/* main.rs:166-167 / benches/demod_benchmark.rs:10-11 applied buffer after
 * buffer; the filter is NOT flushed in between (main.rs never flushes). */
size_t orc_demod_iq(orc_filter *f, const int16_t *iq_re_im, size_t n_samples, orc_msg *out,
                    size_t cap, orc_stats *stats)
{
    orc_magbuf *mb = (orc_magbuf *)malloc(sizeof(orc_magbuf)); /* 263 KiB: off the stack */
    if (!mb)
        return 0;
    size_t found = 0;
    uint64_t chunk = 0;
    for (size_t off = 0; off < n_samples; off += ORC_MODES_MAG_BUF_SAMPLES, chunk++) {
        size_t n = n_samples - off;
        if (n > ORC_MODES_MAG_BUF_SAMPLES)
            n = ORC_MODES_MAG_BUF_SAMPLES;
        orc_to_mag(iq_re_im + 2 * off, n, mb);
        size_t room = found < cap ? cap - found : 0;
        found += orc_demodulate2400(f, mb, chunk, out ? out + (found < cap ? found : cap) : out,
                                    room, stats);
    }
    free(mb);
    return found;
}

/* Carry-over extension (see the header: NOT reference behaviour). */
size_t orc_demod_iq_carry(orc_filter *f, const int16_t *iq_re_im, size_t n_samples, orc_msg *out,
                          size_t cap, orc_stats *stats, int16_t *carry)
{
    orc_magbuf *mb = (orc_magbuf *)malloc(sizeof(orc_magbuf));
    if (!mb)
        return 0;
    size_t found = 0;
    uint64_t chunk = 0;
    for (size_t off = 0; off < n_samples; off += ORC_MODES_MAG_BUF_SAMPLES, chunk++) {
        size_t n = n_samples - off;
        if (n > ORC_MODES_MAG_BUF_SAMPLES)
            n = ORC_MODES_MAG_BUF_SAMPLES;
        orc_to_mag(iq_re_im + 2 * off, n, mb);
        /* lead-in data[326 - d] = magnitude of the sample d before this buffer (d = 1..326) */
        for (size_t d = 1; d <= ORC_TRAILING_SAMPLES; d++) {
            const int16_t *s = off >= d ? iq_re_im + 2 * (off - d)
                                        : carry + 2 * (ORC_TRAILING_SAMPLES - (d - off));
            mb->data[ORC_TRAILING_SAMPLES - d] = orc_mag_sample(s[0], s[1]);
        }
        size_t room = found < cap ? cap - found : 0;
        found += orc_demodulate2400(f, mb, chunk, out ? out + (found < cap ? found : cap) : out,
                                    room, stats);
    }
    /* the last 326 samples of the stream so far */
    if (n_samples >= ORC_TRAILING_SAMPLES) {
        memcpy(carry, iq_re_im + 2 * (n_samples - ORC_TRAILING_SAMPLES),
               2 * ORC_TRAILING_SAMPLES * sizeof(int16_t));
    } else if (n_samples) {
        memmove(carry, carry + 2 * n_samples, 2 * (ORC_TRAILING_SAMPLES - n_samples) * sizeof(int16_t));
        memcpy(carry + 2 * (ORC_TRAILING_SAMPLES - n_samples), iq_re_im, 2 * n_samples * sizeof(int16_t));
    }
    free(mb);
    return found;
}

/* src/utils.rs:23-40: each file pair is [im][re], little-endian i16 */
long orc_read_test_data(const char *path, int16_t *iq_re_im, size_t max_samples)
{
    FILE *fp = fopen(path, "rb");
    if (!fp)
        return -1;
    size_t k = 0;
    uint8_t b[4];
    while (k < max_samples && fread(b, 1, 4, fp) == 4) {
        int16_t im = (int16_t)(b[0] | (b[1] << 8));
        int16_t re = (int16_t)(b[2] | (b[3] << 8));
        iq_re_im[2 * k] = re;
        iq_re_im[2 * k + 1] = im;
        k++;
    }
    fclose(fp);
    return (long)k;
}
