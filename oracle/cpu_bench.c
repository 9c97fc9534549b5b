/*
 * cpu_bench.c -- CPU ORACLE side of bench.py's cpu_baseline leg (test infrastructure, see
 * cryo_oracle.h).  Times, on the host cores of the machine the GPU benchmark runs on,
 *   kind "port": this repo's restatement (oracle/lz4_oracle.c, zstd_dec_oracle.c), or
 *   the stock library the reference links (liblz4.so.1 / libzstd.so.1 via dlopen), called
 *   exactly as the reference calls it (compression.c:84 / :116),
 * over a set of blocks, with T threads pinned to the CPUs the caller lists (bench.py: one per physical core,
 * SURVEY.md 8d): thread t handles blocks i = t mod T with a private output buffer.
 */
#define _GNU_SOURCE
#include "cryo_oracle.h"
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*lz4_dec_fn)(const char *, char *, int, int);
typedef size_t (*zstd_dec_fn)(void *, size_t, const void *, size_t);

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------------
 * `inner` passes over n DISTINCT blocks per timed region (1 for one thread; more with many threads, whose single pass
 * over 4096 blocks is a millisecond: thread wake-up jitter then dominated, the all-core figure moved +-15 % between
 * runs), repeated `reps` times, median region time: thread t
 * handles blocks i = t mod T with a private output buffer; a pass is timed from a common start to the
 * last thread's finish.  direction 0 = decode (src = compressed blocks at off[i], size[i] bytes each),
 * 1 = encode (src = raw blocks at off[i], B bytes each; `param` = lz4 acceleration / zstd level).
 * stock = 1: the library the reference links, called as reference compression.c:70-72,84,102-104,116
 * does; stock = 0: this repo's restatement (zstd: single-threaded, its context is static).  cpus (or NULL): the CPU
 * each thread pins itself to.
 * Returns uncompressed GB/s (1e9) or a negative value: -1 library missing, -2 wrong result.
 * ------------------------------------------------------------------------------------------------ */
typedef int (*lz4_enc_fn)(const char *, char *, int, int, int);
typedef size_t (*zstd_enc_fn)(void *, size_t, const void *, size_t, int);

typedef struct {
    const uint8_t *base;
    const uint64_t *off;
    const uint32_t *size;
    uint32_t n, B;
    int t, T, method, stock, encode, param, reps, inner;
    lz4_dec_fn lz4d; zstd_dec_fn zstdd; lz4_enc_fn lz4e; zstd_enc_fn zstde;
    int cpu;        /* >= 0: the CPU this thread pins itself to */
    pthread_barrier_t *bar;
    double *pass_s; /* written by thread 0 */
    uint64_t out_bytes;
    int failed;
} job2;

static void *worker2(void *arg)
{
    job2 *j = arg;
    const size_t cap = (size_t)j->B + (size_t)j->B / 128 + 1024;
    uint8_t *out = malloc(cap);
    int r;
    if (!out) j->failed = 1;
    if (j->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(j->cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set); /* a refused pin only costs repeatability */
    }
    /* With several threads every thread first copies its share of the input into memory it allocates and touches
     * itself, after pinning: the caller's array sits on the NUMA node of the thread that filled it, and 128 cores
     * reading one node's memory measured that node's controllers (88-122 GB/s, 301 in the one run in nine where the
     * kernel had migrated the pages), not the library. */
    uint8_t *local = NULL;
    uint64_t *loff = NULL;
    if (out && j->T > 1) {
        uint64_t tot = 0;
        uint32_t k = 0, i;
        for (i = (uint32_t)j->t; i < j->n; i += (uint32_t)j->T) { tot += (j->encode ? j->B : j->size[i]) + 64u; k++; }
        local = malloc(tot + 64u);
        loff = malloc(((size_t)k + 1u) * sizeof *loff);
        if (local && loff) {
            uint64_t p = 0;
            k = 0;
            for (i = (uint32_t)j->t; i < j->n; i += (uint32_t)j->T) {
                const size_t len = j->encode ? j->B : j->size[i];
                memcpy(local + p, j->base + j->off[i], len);
                loff[k++] = p;
                p += len + 64u;
            }
        } else { free(local); free(loff); local = NULL; loff = NULL; }
    }
    for (r = 0; r < j->reps; r++) {
        uint32_t i;
        double t0;
        int pass;
        pthread_barrier_wait(j->bar);
        t0 = now();
        for (pass = 0; pass < j->inner; pass++) {
        uint32_t k = 0;
        for (i = (uint32_t)j->t; out && i < j->n; i += (uint32_t)j->T, k++) {
            const uint8_t *src = local ? local + loff[k] : j->base + j->off[i];
            long res;
            if (!j->encode) {
                if (j->stock) res = j->method == 0 ? (long)j->lz4d((const char *)src, (char *)out, (int)j->size[i], (int)j->B)
                                                   : (long)j->zstdd(out, j->B, src, j->size[i]);
                else res = j->method == 0 ? cryo_oracle_lz4_decompress(src, j->size[i], out, j->B)
                                          : cryo_oracle_zstd_decompress(src, j->size[i], out, j->B);
                if (res != (long)j->B) j->failed = 1;
            } else {
                if (j->stock) res = j->method == 0 ? (long)j->lz4e((const char *)src, (char *)out, (int)j->B, (int)cap, j->param)
                                                   : (long)j->zstde(out, cap, src, j->B, j->param);
                else res = j->method == 0 ? (long)cryo_oracle_lz4_compress(src, j->B, out, cap, j->param)
                                          : (long)cryo_oracle_zstd_compress(src, j->B, out, cap, j->param);
                if (res <= 0 || (size_t)res > cap) j->failed = 1;
                if (r == 0 && pass == 0) j->out_bytes += (uint64_t)res;
            }
        }
        }
        pthread_barrier_wait(j->bar);
        if (j->t == 0) j->pass_s[r] = now() - t0;
    }
    free(out);
    free(local);
    free(loff);
    return NULL;
}

static int cmp_double(const void *a, const void *b) { return (*(const double *)a > *(const double *)b) - (*(const double *)a < *(const double *)b); }

double cryo_oracle_cpu_pass_bench(int method, int encode, int stock, int param, const uint8_t *base, const uint64_t *off,
                                  const uint32_t *size, uint32_t n, uint32_t B, int threads, const int *cpus, int reps, int inner,
                                  uint64_t *out_bytes, char *version, size_t version_cap)
{
    job2 proto;
    pthread_t *th;
    job2 *jobs;
    pthread_barrier_t bar;
    double *pass_s, med;
    int t, failed = 0;
    memset(&proto, 0, sizeof proto);
    if (version && version_cap) version[0] = 0;
    if (threads < 1) threads = 1;
    if (reps < 1) reps = 1;
    if (inner < 1) inner = 1;
    if (stock) {
        void *h = dlopen(method == 0 ? "liblz4.so.1" : "libzstd.so.1", RTLD_NOW);
        const char *(*ver)(void);
        if (!h) return -1.0;
        if (method == 0) {
            proto.lz4d = (lz4_dec_fn)dlsym(h, "LZ4_decompress_safe");
            proto.lz4e = (lz4_enc_fn)dlsym(h, "LZ4_compress_fast");
            ver = (const char *(*)(void))dlsym(h, "LZ4_versionString");
            if (!proto.lz4d || !proto.lz4e) return -1.0;
        } else {
            proto.zstdd = (zstd_dec_fn)dlsym(h, "ZSTD_decompress");
            proto.zstde = (zstd_enc_fn)dlsym(h, "ZSTD_compress");
            ver = (const char *(*)(void))dlsym(h, "ZSTD_versionString");
            if (!proto.zstdd || !proto.zstde) return -1.0;
        }
        if (ver && version) strncpy(version, ver(), version_cap - 1);
    } else if (method != 0 && threads > 1) {
        threads = 1;
    }
    th = calloc((size_t)threads, sizeof *th);
    jobs = calloc((size_t)threads, sizeof *jobs);
    pass_s = calloc((size_t)reps, sizeof *pass_s);
    if (!th || !jobs || !pass_s) return -1.0;
    pthread_barrier_init(&bar, NULL, (unsigned)threads);
    for (t = 0; t < threads; t++) {
        jobs[t] = proto;
        jobs[t].base = base; jobs[t].off = off; jobs[t].size = size; jobs[t].n = n; jobs[t].B = B;
        jobs[t].t = t; jobs[t].T = threads; jobs[t].method = method; jobs[t].stock = stock; jobs[t].encode = encode;
        jobs[t].param = param; jobs[t].reps = reps; jobs[t].bar = &bar; jobs[t].pass_s = pass_s;
        jobs[t].cpu = cpus ? cpus[t] : -1;
        jobs[t].inner = inner;
        pthread_create(&th[t], NULL, worker2, &jobs[t]);
    }
    if (out_bytes) *out_bytes = 0;
    for (t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        failed |= jobs[t].failed;
        if (out_bytes) *out_bytes += jobs[t].out_bytes;
    }
    pthread_barrier_destroy(&bar);
    qsort(pass_s, (size_t)reps, sizeof *pass_s, cmp_double);
    med = pass_s[reps / 2];
    free(th); free(jobs); free(pass_s);
    if (failed) return -2.0;
    return (double)n * (double)B * (double)inner / med / 1e9;
}
