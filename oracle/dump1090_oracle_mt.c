/*
 * dump1090_oracle_mt.c -- the oracle over N host threads.  TEST INFRASTRUCTURE ONLY
 * (see dump1090_oracle.h): used by bench.py's cpu_baseline leg and by tests.
 *
 * "N threads over chunks with ordered replay".  The only state
 * demodulate2400 carries from one position to the next is the ICAO filter
 * (reference src/mode_s/mod.rs:71,80-84,97-104,115,130), so the capture is cut into the
 * reference's 131072-sample MagnitudeBuffers (src/lib.rs:22-51) and worker threads run
 * to_mag + gates + the five slicer phases of every surviving j (orc_all_trials: no filter
 * involved) -- and, since round 4, the DF class and the CRC residual of every trial too, so
 * that the one serial stage only sees trials that can matter.  A trial can change the output
 * only if it can score >= 0 or add to the filter (src/mode_s/mod.rs:56-135):
 *   - clean DF11 (residual & 0xFFFF80 == 0) and clean DF17 / DF18 (residual == 0): kept;
 *     DF11 with IID 0 and DF17 will add their address (:80-84, :97-99): its bit is set in
 *     a 2^24-bit set shared by the workers (DF18 adds addr | 1 << 25, which no 24-bit
 *     residual equals);
 *   - address/parity DFs (0, 4, 5, 16, 20, 21, 24..31) score 1000 iff their residual is in
 *     the filter at that moment: kept iff it is in that set once every worker of the round
 *     has finished its buffers -- a superset in time of the filter (what the filter held on
 *     entry is put into the set first; address 0 always tests true, src/icao_filter.rs:71-80);
 *   - everything else scores -2 / -1 / None, is never emitted and never beats a score >= 0.
 * The survivors (a handful per buffer) are replayed in (buffer, j, try_phase) order through
 * orc_score_modes_message and the strict-greater selection of src/demod_2400.rs:149-207 by
 * the calling thread.  This is the argument of DESIGN.md section 3 applied to host threads;
 * the result is identical to orc_demod_iq on one thread (tests/test_oracle_golden.py checks
 * that on the reference captures and on synthetic IQ, bench.py on every run).
 *
 * The capture is processed in rounds of threads * 4 buffers: classify (parallel), barrier,
 * keep / drop the address/parity trials (parallel), barrier, replay (serial, microseconds).
 */
#define _POSIX_C_SOURCE 200809L /* pthread barriers under -std=c11 */
#include "dump1090_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    orc_trial *trials; /* kept trials of the buffer, (j, try_phase) ascending; pad: 1 = address/parity */
    uint32_t *residual; /* ... and their CRC residual */
    size_t n, sliced;   /* kept; all trials the buffer sliced (statistics) */
    int owned;          /* malloc'ed for this buffer (the round's slab was full), else a piece of the slab */
} chunk_result;

typedef struct {
    const int16_t *iq;
    size_t n_samples, n_chunks;
    chunk_result *res;          /* one per buffer of the current round */
    size_t round_first, round_n;
    atomic_size_t next1, next2; /* work counters of the two parallel phases */
    _Atomic uint32_t *seen;     /* 2^24 bits: addresses the filter may hold (superset in time) */
    /* one slab for the kept trials of a round, handed out with an atomic bump: 256 threads that each malloc
     * and free ~125 KB per buffer spend their time in the allocator's mprotect calls (a write lock on the
     * process's address space that every page fault of every thread then waits for) */
    orc_trial *slab;
    uint32_t *slab_res;
    size_t slab_cap;
    atomic_size_t slab_used;
    pthread_barrier_t bar;
    atomic_int failed, quit;
} mt_job;

static inline void seen_set(mt_job *job, uint32_t addr)
{
    atomic_fetch_or_explicit(&job->seen[(addr & 0xFFFFFFu) >> 5], 1u << (addr & 31), memory_order_relaxed);
}

static inline int seen_test(const mt_job *job, uint32_t addr)
{
    return (atomic_load_explicit(&job->seen[(addr & 0xFFFFFFu) >> 5], memory_order_relaxed) >> (addr & 31)) & 1u;
}

/* phase 1: one buffer -> its candidate trials */
static void classify_chunk(mt_job *job, size_t c, orc_magbuf *mb, orc_trial **scratch, size_t *scratch_cap,
                           uint32_t **res_tmp, size_t *res_cap)
{
    chunk_result *r = &job->res[c - job->round_first];
    const size_t off = c * (size_t)ORC_MODES_MAG_BUF_SAMPLES;
    size_t n = job->n_samples - off;
    if (n > ORC_MODES_MAG_BUF_SAMPLES)
        n = ORC_MODES_MAG_BUF_SAMPLES;
    orc_to_mag(job->iq + 2 * off, n, mb);
    size_t got = orc_all_trials(mb, c, *scratch, *scratch_cap);
    if (got > *scratch_cap) { /* denser than expected: size exactly and redo */
        free(*scratch);
        *scratch_cap = got;
        *scratch = (orc_trial *)malloc(got * sizeof(orc_trial));
        if (!*scratch) {
            *scratch_cap = 0;
            atomic_store(&job->failed, 1);
            return;
        }
        got = orc_all_trials(mb, c, *scratch, *scratch_cap);
    }
    r->sliced = got;
    /* in place: kept trials move to the front, order preserved */
    orc_trial *t = *scratch;
    size_t keep = 0;
    if (got > *res_cap) {
        free(*res_tmp);
        *res_cap = got;
        *res_tmp = (uint32_t *)malloc(got * sizeof(uint32_t));
        if (!*res_tmp) {
            *res_cap = 0;
            atomic_store(&job->failed, 1);
            return;
        }
    }
    uint32_t *res = *res_tmp;
    for (size_t i = 0; i < got; i++) {
        const uint8_t *m = t[i].msg;
        const unsigned df = m[0] >> 3;                       /* mod.rs:41 */
        const size_t bits = (df & 0x10) ? 112 : 56;          /* :42-46 */
        const uint32_t crc = orc_modes_checksum(m, bits);
        const uint32_t addr = (uint32_t)m[1] << 16 | (uint32_t)m[2] << 8 | m[3];
        int kind = -1; /* 0 self-validating, 1 address/parity */
        switch (df) {
        case 0: case 4: case 5: case 16: case 20: case 21:
        case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31:
            kind = 1;                                        /* :56-72, :110-135 */
            break;
        case 11:
            if ((crc & 0xFFFF80u) == 0) {                    /* :74-92 */
                kind = 0;
                if ((crc & 0x7Fu) == 0)
                    seen_set(job, addr);
            }
            break;
        case 17:
        case 18:
            if (crc == 0) {                                  /* :94-108 */
                kind = 0;
                if (df == 17)
                    seen_set(job, addr);
            }
            break;
        default:
            break;
        }
        if (kind < 0)
            continue;
        if (keep != i)
            t[keep] = t[i];
        t[keep].pad = (uint16_t)kind;
        res[keep] = crc;
        keep++;
    }
    const size_t at = atomic_fetch_add(&job->slab_used, keep);
    if (at + keep <= job->slab_cap) {
        r->trials = job->slab + at;
        r->residual = job->slab_res + at;
        r->owned = 0;
    } else { /* a round far denser than the slab was sized for */
        r->trials = (orc_trial *)malloc((keep ? keep : 1) * sizeof(orc_trial));
        r->residual = (uint32_t *)malloc((keep ? keep : 1) * sizeof(uint32_t));
        r->owned = 1;
        if (!r->trials || !r->residual) {
            atomic_store(&job->failed, 1);
            return;
        }
    }
    memcpy(r->trials, t, keep * sizeof(orc_trial));
    memcpy(r->residual, res, keep * sizeof(uint32_t));
    r->n = keep;
}

/* phase 2: drop the address/parity trials whose residual no address of the set equals */
static void filter_chunk(mt_job *job, size_t c)
{
    chunk_result *r = &job->res[c - job->round_first];
    size_t keep = 0;
    for (size_t i = 0; i < r->n; i++) {
        if (r->trials[i].pad == 1 && !seen_test(job, r->residual[i]))
            continue;
        if (keep != i)
            r->trials[keep] = r->trials[i];
        r->trials[keep].pad = 0;
        keep++;
    }
    r->n = keep;
}

static void *worker(void *arg)
{
    mt_job *job = (mt_job *)arg;
    orc_magbuf *mb = (orc_magbuf *)malloc(sizeof(orc_magbuf));
    size_t scratch_cap = 16384, res_cap = 16384;
    orc_trial *scratch = (orc_trial *)malloc(scratch_cap * sizeof(orc_trial));
    uint32_t *res_tmp = (uint32_t *)malloc(res_cap * sizeof(uint32_t));
    if (!mb || !scratch || !res_tmp)
        atomic_store(&job->failed, 1);
    for (;;) {
        pthread_barrier_wait(&job->bar); /* round start (the caller has set it up) */
        if (atomic_load(&job->quit))
            break;
        for (;;) {
            const size_t k = atomic_fetch_add(&job->next1, 1);
            if (k >= job->round_n)
                break;
            if (!atomic_load(&job->failed))
                classify_chunk(job, job->round_first + k, mb, &scratch, &scratch_cap, &res_tmp, &res_cap);
        }
        pthread_barrier_wait(&job->bar); /* every address of the round is in the set */
        for (;;) {
            const size_t k = atomic_fetch_add(&job->next2, 1);
            if (k >= job->round_n)
                break;
            if (!atomic_load(&job->failed))
                filter_chunk(job, job->round_first + k);
        }
        pthread_barrier_wait(&job->bar); /* the caller replays */
    }
    free(res_tmp);
    free(scratch);
    free(mb);
    return NULL;
}

/* src/demod_2400.rs:149-207 over one buffer's kept trials ((j, try_phase) ascending) */
static size_t replay_chunk(orc_filter *f, const orc_trial *t, size_t n, uint64_t chunk, orc_msg *out,
                           size_t cap, size_t found, orc_stats *stats)
{
    size_t i = 0;
    while (i < n) {
        const uint32_t j = t[i].j_tp & 0xFFFFFFu;
        orc_msg best;
        memset(&best, 0, sizeof(best));
        best.score = -2;
        best.len = ORC_MODES_SHORT_MSG_BYTES;
        for (; i < n && (t[i].j_tp & 0xFFFFFFu) == j; i++) {
            const orc_trial *tr = &t[i];
            int msglen;
            int32_t score;
            if (!orc_score_modes_message(f, tr->msg, 14, &msglen, &score))
                continue;
            if (score > best.score) {
                memcpy(best.msg, tr->msg, 14);
                best.len = (uint8_t)msglen;
                best.score = score;
                best.try_phase = (uint8_t)(tr->j_tp >> 24);
                const double signal_power = (double)tr->power / 65535.0 / 65535.0;
                best.signal_level = signal_power / 33.0;
            }
        }
        if (best.score < 0)
            continue;
        best.j = j;
        best.chunk = chunk;
        if (found < cap)
            out[found] = best;
        found++;
        if (stats)
            stats->frames++;
    }
    return found;
}

size_t orc_demod_iq_mt(orc_filter *f, const int16_t *iq_re_im, size_t n_samples, orc_msg *out,
                       size_t cap, orc_stats *stats, int threads)
{
    if (threads < 1)
        threads = 1;
    mt_job job;
    memset(&job, 0, sizeof(job));
    job.iq = iq_re_im;
    job.n_samples = n_samples;
    job.n_chunks = (n_samples + ORC_MODES_MAG_BUF_SAMPLES - 1) / ORC_MODES_MAG_BUF_SAMPLES;
    if (job.n_chunks == 0)
        return 0;
    if ((size_t)threads > job.n_chunks)
        threads = (int)job.n_chunks;
    const size_t round_max = (size_t)threads * 4 < job.n_chunks ? (size_t)threads * 4 : job.n_chunks;
    job.res = (chunk_result *)calloc(round_max, sizeof(chunk_result));
    job.slab_cap = round_max * 4096; /* noise keeps ~3 200 of a buffer's ~7 000 trials until the round's addresses are known */
    job.slab = (orc_trial *)malloc(job.slab_cap * sizeof(orc_trial));
    job.slab_res = (uint32_t *)malloc(job.slab_cap * sizeof(uint32_t));
    job.seen = (_Atomic uint32_t *)calloc(1u << 19, sizeof(uint32_t));
    pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    if (!job.res || !job.seen || !th || !job.slab || !job.slab_res ||
        pthread_barrier_init(&job.bar, NULL, (unsigned)threads + 1) != 0) {
        free(job.res);
        free((void *)job.seen);
        free(th);
        free(job.slab);
        free(job.slab_res);
        return 0;
    }
    /* what the filter holds on entry can match from the first sample on; 0 always tests true */
    seen_set(&job, 0);
    for (size_t i = 0; i < ORC_ICAO_FILTER_SIZE; i++)
        if (f->a[i] != 0 && f->a[i] <= 0xFFFFFFu)
            seen_set(&job, f->a[i]);
    int started = 0;
    for (; started < threads; started++)
        if (pthread_create(&th[started], NULL, worker, &job) != 0)
            break;
    size_t found = 0;
    if (started == threads) {
        for (size_t first = 0; first < job.n_chunks && !atomic_load(&job.failed); first += round_max) {
            job.round_first = first;
            job.round_n = job.n_chunks - first < round_max ? job.n_chunks - first : round_max;
            atomic_store(&job.next1, 0);
            atomic_store(&job.next2, 0);
            atomic_store(&job.slab_used, 0);
            memset(job.res, 0, round_max * sizeof(chunk_result));
            const double t0 = now_s();
            pthread_barrier_wait(&job.bar); /* start */
            pthread_barrier_wait(&job.bar); /* classified */
            const double t1 = now_s();
            pthread_barrier_wait(&job.bar); /* filtered */
            const double t2 = now_s();
            size_t kept = 0;
            for (size_t k = 0; k < job.round_n; k++) kept += job.res[k].n;
            for (size_t k = 0; k < job.round_n; k++) {
                chunk_result *r = &job.res[k];
                if (!atomic_load(&job.failed)) {
                    found = replay_chunk(f, r->trials, r->n, first + k, out, cap, found, stats);
                    if (stats) { /* only the sliced positions are known here */
                        stats->trials += r->sliced;
                        stats->preamble_pass += r->sliced / 5;
                        stats->snr_pass += r->sliced / 5;
                        stats->quiet_pass += r->sliced / 5;
                    }
                }
                if (r->owned) {
                    free(r->trials);
                    free(r->residual);
                }
            }
            if (getenv("ORC_MT_TIMES")) /* (diagnostic: where a round's time goes) */
                fprintf(stderr, "orc_demod_iq_mt: round of %zu buffers: classify %.1f ms, filter %.1f ms, replay of %zu trials %.1f ms\n",
                        job.round_n, 1e3 * (t1 - t0), 1e3 * (t2 - t1), kept, 1e3 * (now_s() - t2));
        }
        atomic_store(&job.quit, 1);
        pthread_barrier_wait(&job.bar);
    } else {
        /* could not start every thread: the barrier would never fill; run what started to the end */
        atomic_store(&job.failed, 1);
        atomic_store(&job.quit, 1);
        /* the started workers wait at the barrier for threads + 1 parties: cancel them */
        for (int i = 0; i < started; i++)
            pthread_cancel(th[i]);
    }
    for (int i = 0; i < started; i++)
        pthread_join(th[i], NULL);
    const int failed = atomic_load(&job.failed);
    pthread_barrier_destroy(&job.bar);
    free(th);
    free(job.res);
    free((void *)job.seen);
    free(job.slab);
    free(job.slab_res);
    return failed ? 0 : found;
}
