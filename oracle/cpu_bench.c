/*
 * cpu_bench.c -- CPU ORACLE side of bench.py's cpu_baseline leg (test infrastructure, see
 * cryo_oracle.h).  Times, on the host cores of the machine the GPU benchmark runs on,
 *   kind "port": this repo's restatement (oracle/lz4_oracle.c, zstd_dec_oracle.c), or
 *   the stock library the reference links (liblz4.so.1 / libzstd.so.1 via dlopen), called
 *   exactly as the reference calls it (compression.c:84 / :116),
 * over a set of compressed blocks, with T threads: thread t decodes blocks i = t mod T into
 * a private buffer until the time budget ends.
 */
#define _GNU_SOURCE
#include "cryo_oracle.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*lz4_dec_fn)(const char *, char *, int, int);
typedef size_t (*zstd_dec_fn)(void *, size_t, const void *, size_t);

typedef struct {
    const uint8_t *base;
    const uint64_t *off;
    const uint32_t *size;
    uint32_t n, B;
    int t, T, method, stock;
    double seconds;
    lz4_dec_fn lz4;
    zstd_dec_fn zstd;
    uint64_t blocks_done;
    int failed;
} job;

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *arg)
{
    job *j = arg;
    uint8_t *out = malloc((size_t)j->B + 64);
    const double t_end = now() + j->seconds;
    uint64_t done = 0;
    if (!out) { j->failed = 1; return NULL; }
    do {
        uint32_t i;
        for (i = (uint32_t)j->t; i < j->n; i += (uint32_t)j->T) {
            const uint8_t *src = j->base + j->off[i];
            long r;
            if (j->stock) {
                if (j->method == 0) r = j->lz4((const char *)src, (char *)out, (int)j->size[i], (int)j->B);
                else r = (long)j->zstd(out, j->B, src, j->size[i]);
            } else {
                r = j->method == 0 ? cryo_oracle_lz4_decompress(src, j->size[i], out, j->B)
                                   : cryo_oracle_zstd_decompress(src, j->size[i], out, j->B);
            }
            if (r != (long)j->B) j->failed = 1;
            done++;
        }
    } while (now() < t_end && !j->failed);
    j->blocks_done = done;
    free(out);
    return NULL;
}

/* returns uncompressed GB/s (1e9), or a negative value: -1 library missing, -2 decode mismatch */
double cryo_oracle_cpu_decode_bench(int method, int stock, const uint8_t *base, const uint64_t *off,
                                    const uint32_t *size, uint32_t n, uint32_t B, int threads, double seconds,
                                    char *version, size_t version_cap)
{
    lz4_dec_fn lz4 = NULL;
    zstd_dec_fn zstd = NULL;
    pthread_t *th;
    job *jobs;
    double t0, dt;
    uint64_t total = 0;
    int t, failed = 0;
    if (version && version_cap) version[0] = 0;
    if (threads < 1) threads = 1;
    if (stock) {
        if (method == 0) {
            void *h = dlopen("liblz4.so.1", RTLD_NOW);
            const char *(*ver)(void);
            if (!h) return -1.0;
            lz4 = (lz4_dec_fn)dlsym(h, "LZ4_decompress_safe");
            ver = (const char *(*)(void))dlsym(h, "LZ4_versionString");
            if (!lz4) return -1.0;
            if (ver && version) strncpy(version, ver(), version_cap - 1);
        } else {
            void *h = dlopen("libzstd.so.1", RTLD_NOW);
            const char *(*ver)(void);
            if (!h) return -1.0;
            zstd = (zstd_dec_fn)dlsym(h, "ZSTD_decompress");
            ver = (const char *(*)(void))dlsym(h, "ZSTD_versionString");
            if (!zstd) return -1.0;
            if (ver && version) strncpy(version, ver(), version_cap - 1);
        }
    } else if (method != 0 && threads > 1) {
        threads = 1; /* the zstd oracle keeps its context in a static: single-threaded by design */
    }
    th = calloc((size_t)threads, sizeof *th);
    jobs = calloc((size_t)threads, sizeof *jobs);
    if (!th || !jobs) return -1.0;
    t0 = now();
    for (t = 0; t < threads; t++) {
        job *j = &jobs[t];
        j->base = base; j->off = off; j->size = size; j->n = n; j->B = B;
        j->t = t; j->T = threads; j->method = method; j->stock = stock; j->seconds = seconds;
        j->lz4 = lz4; j->zstd = zstd;
        pthread_create(&th[t], NULL, worker, j);
    }
    for (t = 0; t < threads; t++) { pthread_join(th[t], NULL); total += jobs[t].blocks_done; failed |= jobs[t].failed; }
    dt = now() - t0;
    free(th); free(jobs);
    if (failed) return -2.0;
    return (double)total * (double)B / dt / 1e9;
}
