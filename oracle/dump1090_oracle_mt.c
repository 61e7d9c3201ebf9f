/*
 * dump1090_oracle_mt.c -- the oracle over N host threads.  TEST INFRASTRUCTURE ONLY
 * (see dump1090_oracle.h): used by bench.py's cpu_baseline leg and by tests.
 *
 * "N threads over chunks with ordered replay".  The only state
 * demodulate2400 carries from one position to the next is the ICAO filter
 * (reference src/mode_s/mod.rs:71,80-84,97-104,115,130), so the buffers are cut into
 * the reference's 131072-sample MagnitudeBuffers (src/lib.rs:22-51), worker threads run
 * to_mag + gates + the five slicer phases of every surviving j (orc_all_trials: no filter
 * involved), and one thread replays the trials buffer by buffer, in (j, try_phase) order,
 * through orc_score_modes_message and the strict-greater selection of
 * src/demod_2400.rs:149-207.  The result is identical to orc_demod_iq on one thread
 * (tests/test_oracle_golden.py checks that on the reference captures and on synthetic IQ).
 */
#include "dump1090_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    orc_trial *trials; /* malloc'ed by the worker, freed by the replay */
    size_t n;
    atomic_int ready;
} chunk_result;

typedef struct {
    const int16_t *iq;
    size_t n_samples, n_chunks;
    chunk_result *res;
    atomic_size_t next;        /* next chunk a worker takes */
    atomic_size_t replayed;    /* chunks the replay has consumed (bounds the run-ahead) */
    size_t window;             /* workers stay within this many chunks of the replay */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    atomic_int failed;
} mt_job;

static void *worker(void *arg)
{
    mt_job *job = (mt_job *)arg;
    orc_magbuf *mb = (orc_magbuf *)malloc(sizeof(orc_magbuf));
    if (!mb) {
        atomic_store(&job->failed, 1);
        return NULL;
    }
    for (;;) {
        const size_t c = atomic_fetch_add(&job->next, 1);
        if (c >= job->n_chunks)
            break;
        /* bounded run-ahead: trial lists of a dense capture are large */
        pthread_mutex_lock(&job->mu);
        while (c >= atomic_load(&job->replayed) + job->window && !atomic_load(&job->failed))
            pthread_cond_wait(&job->cv, &job->mu);
        pthread_mutex_unlock(&job->mu);

        const size_t off = c * (size_t)ORC_MODES_MAG_BUF_SAMPLES;
        size_t n = job->n_samples - off;
        if (n > ORC_MODES_MAG_BUF_SAMPLES)
            n = ORC_MODES_MAG_BUF_SAMPLES;
        orc_to_mag(job->iq + 2 * off, n, mb);
        size_t cap = 16384;
        orc_trial *t = (orc_trial *)malloc(cap * sizeof(orc_trial));
        size_t got = t ? orc_all_trials(mb, c, t, cap) : 0;
        if (t && got > cap) { /* denser than expected: size exactly and redo */
            free(t);
            cap = got;
            t = (orc_trial *)malloc(cap * sizeof(orc_trial));
            got = t ? orc_all_trials(mb, c, t, cap) : 0;
        }
        if (!t)
            atomic_store(&job->failed, 1);
        job->res[c].trials = t;
        job->res[c].n = got;
        pthread_mutex_lock(&job->mu);
        atomic_store(&job->res[c].ready, 1);
        pthread_cond_broadcast(&job->cv);
        pthread_mutex_unlock(&job->mu);
    }
    free(mb);
    return NULL;
}

/* src/demod_2400.rs:149-207 over one buffer's trials (5 per j, try_phase ascending) */
static size_t replay_chunk(orc_filter *f, const orc_trial *t, size_t n, uint64_t chunk, orc_msg *out,
                           size_t cap, size_t found, orc_stats *stats)
{
    for (size_t i = 0; i + 5 <= n; i += 5) {
        orc_msg best;
        memset(&best, 0, sizeof(best));
        best.score = -2;
        best.len = ORC_MODES_SHORT_MSG_BYTES;
        if (stats) {
            stats->preamble_pass++; /* only the sliced positions are known here */
            stats->snr_pass++;
            stats->quiet_pass++;
        }
        for (int k = 0; k < 5; k++) {
            const orc_trial *tr = &t[i + k];
            int msglen;
            int32_t score;
            if (stats)
                stats->trials++;
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
        best.j = t[i].j_tp & 0xFFFFFFu;
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
    job.res = (chunk_result *)calloc(job.n_chunks, sizeof(chunk_result));
    if (!job.res)
        return 0;
    job.window = (size_t)threads * 4;
    pthread_mutex_init(&job.mu, NULL);
    pthread_cond_init(&job.cv, NULL);
    pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    int started = 0;
    for (; th && started < threads; started++)
        if (pthread_create(&th[started], NULL, worker, &job) != 0)
            break;
    size_t found = 0;
    if (started > 0) {
        for (size_t c = 0; c < job.n_chunks; c++) {
            pthread_mutex_lock(&job.mu);
            while (!atomic_load(&job.res[c].ready) && !atomic_load(&job.failed))
                pthread_cond_wait(&job.cv, &job.mu);
            pthread_mutex_unlock(&job.mu);
            if (atomic_load(&job.failed))
                break;
            found = replay_chunk(f, job.res[c].trials, job.res[c].n, c, out, cap, found, stats);
            free(job.res[c].trials);
            job.res[c].trials = NULL;
            pthread_mutex_lock(&job.mu);
            atomic_store(&job.replayed, c + 1);
            pthread_cond_broadcast(&job.cv);
            pthread_mutex_unlock(&job.mu);
        }
    }
    /* on failure: release workers waiting for the window, then join */
    pthread_mutex_lock(&job.mu);
    if (atomic_load(&job.failed) || started == 0)
        atomic_store(&job.failed, 1);
    atomic_store(&job.replayed, job.n_chunks);
    pthread_cond_broadcast(&job.cv);
    pthread_mutex_unlock(&job.mu);
    for (int i = 0; i < started; i++)
        pthread_join(th[i], NULL);
    for (size_t c = 0; c < job.n_chunks; c++)
        free(job.res[c].trials);
    free(th);
    free(job.res);
    pthread_cond_destroy(&job.cv);
    pthread_mutex_destroy(&job.mu);
    return found;
}
