/*
 * lzo_bench.c -- TEST INFRASTRUCTURE ONLY: the timing harness of bench.py's cpu_baseline leg.
 *
 * Times the oracle (oracle/lzfse_oracle.c, the C restatement of lzfse_rust's slice path) on host threads with no
 * interpreter in the loop: n_threads pthreads, thread t owns streams t, t + n_threads, ... (cyclically over the given
 * sample, so every thread has work when there are more threads than streams) and keeps encoding -- then decoding -- its
 * own streams into its own buffers until the time budget is over. What is reported is bytes processed / wall time of
 * the phase, the shape of Criterion's throughput figure (bench/src/bench.rs:195-209) summed over threads.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lzfse_oracle.h"

typedef struct {
    const uint8_t *const *raws;
    const size_t *raw_lens;
    const uint8_t *const *encs;
    const size_t *enc_lens;
    size_t n_streams, first, stride;
    double budget;
    int decode;
    uint64_t bytes; /* raw bytes processed */
    double seconds; /* this thread's own wall time */
    int status;
    pthread_barrier_t *start;
} job_t;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    size_t max_raw = 0;
    for (size_t i = 0; i < j->n_streams; i++)
        if (j->raw_lens[i] > max_raw) max_raw = j->raw_lens[i];
    const size_t cap = lzo_encode_bound(max_raw) + 64;
    uint8_t *buf = (uint8_t *)malloc(cap > max_raw + 64 ? cap : max_raw + 64);
    if (!buf) j->status = LZO_IO;
    pthread_barrier_wait(j->start);
    if (j->status) return NULL;
    const double t0 = now_s();
    size_t k = j->first % j->n_streams;
    for (;;) {
        size_t out = 0;
        int st = j->decode ? lzo_decode(j->encs[k], j->enc_lens[k], buf, j->raw_lens[k], &out, NULL)
                           : lzo_encode(j->raws[k], j->raw_lens[k], buf, cap, &out, NULL);
        if (st) {
            j->status = st;
            break;
        }
        j->bytes += j->raw_lens[k];
        k = (k + j->stride) % j->n_streams;
        if (now_s() - t0 >= j->budget) break;
    }
    j->seconds = now_s() - t0;
    free(buf);
    return NULL;
}

/* One phase (decode = 0: lzo_encode of raws, 1: lzo_decode of encs) on n_threads threads for about `budget` seconds.
 * Returns 0 and *mbps = 10^6 bytes of raw data per second over all threads (each thread's bytes / its own time, summed:
 * a thread that finishes its last stream late does not dilute the others). */
int lzo_bench_threads(const uint8_t *const *raws, const size_t *raw_lens, const uint8_t *const *encs, const size_t *enc_lens,
                      size_t n_streams, int n_threads, double budget, int decode, double *mbps, uint64_t *total_bytes) {
    if (!n_streams || n_threads < 1) return LZO_IO;
    job_t *jobs = (job_t *)calloc((size_t)n_threads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    pthread_barrier_t start;
    if (!jobs || !th || pthread_barrier_init(&start, NULL, (unsigned)n_threads)) {
        free(jobs);
        free(th);
        return LZO_IO;
    }
    int started = 0;
    for (int t = 0; t < n_threads; t++) {
        job_t *j = &jobs[t];
        j->raws = raws; j->raw_lens = raw_lens; j->encs = encs; j->enc_lens = enc_lens;
        j->n_streams = n_streams; j->first = (size_t)t; j->stride = (size_t)n_threads;
        j->budget = budget; j->decode = decode; j->start = &start;
        if (pthread_create(&th[t], NULL, worker, j)) break;
        started++;
    }
    int status = started == n_threads ? 0 : LZO_IO;
    if (status) { /* could not start them all: the barrier would never open */
        for (int t = 0; t < started; t++) pthread_cancel(th[t]);
    }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    double rate = 0;
    uint64_t bytes = 0;
    for (int t = 0; t < started && !status; t++) {
        if (jobs[t].status) status = jobs[t].status;
        if (jobs[t].seconds > 0) rate += (double)jobs[t].bytes / jobs[t].seconds;
        bytes += jobs[t].bytes;
    }
    pthread_barrier_destroy(&start);
    free(jobs);
    free(th);
    if (!status) {
        *mbps = rate / 1e6;
        *total_bytes = bytes;
    }
    return status;
}
