/*
 * lzf_cpu_bench.c — native CPU-baseline driver for bench.py's `cpu_baseline` leg.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (same rule as lzf_oracle.h): the product never links this.
 *
 * Times, on the host's cores and with native threads (pthreads, no Python in the timed region):
 *   (1) the C restatement of lz-fear (lzf_oracle.c: compress2 with a fresh U32Table per block,
 *       src/raw/compress/mod.rs:165-238 as src/framed/compress.rs:243 calls it; decompress_raw,
 *       src/raw/decompress.rs:58-138 as src/framed/decompress.rs:248 calls it), and
 *   (2) liblz4 when `liblz4.so.1` can be dlopen'ed — the C implementation the reference's README
 *       compares itself with (README.md:11,18): LZ4_compress_fast_continue on a fresh stream per
 *       block (byte-identical output regime) and LZ4_decompress_safe.
 * Blocks are handed to the threads through one shared atomic counter (block-parallel, the way
 * independent-blocks frames parallelise); every thread owns its output buffer and table.  Each
 * measurement is one pass over all blocks; the caller asks for `reps` passes and gets every pass's
 * wall time (CLOCK_MONOTONIC) so it can report the median.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "lzf_oracle.h"

typedef struct {
    /* work description */
    int what;                       /* 0 oracle compress, 1 oracle decompress, 2 liblz4 compress, 3 liblz4 decompress */
    const uint8_t* const* in;       /* per block input */
    const uint64_t* in_len;
    const uint64_t* out_len;        /* expected output length (decompress) / capacity (compress) */
    uint32_t n_blocks;
    uint64_t max_out;               /* per-thread output buffer size */
    /* shared state */
    atomic_uint next;
    atomic_int failed;
    pthread_barrier_t* bar;
} work_t;

/* liblz4 entry points (optional) */
static void* g_lz4;
static void* (*p_createStream)(void);
static int (*p_freeStream)(void*);
static void (*p_resetStream)(void*);
static int (*p_fast_continue)(void*, const char*, char*, int, int, int);
static int (*p_decompress_safe)(const char*, char*, int, int);
static const char* (*p_version)(void);

static int load_lz4(void) {
    if (g_lz4) return 1;
    g_lz4 = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!g_lz4) return 0;
    p_createStream = (void* (*)(void))dlsym(g_lz4, "LZ4_createStream");
    p_freeStream = (int (*)(void*))dlsym(g_lz4, "LZ4_freeStream");
    p_resetStream = (void (*)(void*))dlsym(g_lz4, "LZ4_resetStream");
    p_fast_continue = (int (*)(void*, const char*, char*, int, int, int))dlsym(g_lz4, "LZ4_compress_fast_continue");
    p_decompress_safe = (int (*)(const char*, char*, int, int))dlsym(g_lz4, "LZ4_decompress_safe");
    p_version = (const char* (*)(void))dlsym(g_lz4, "LZ4_versionString");
    if (!p_createStream || !p_freeStream || !p_resetStream || !p_fast_continue || !p_decompress_safe) { dlclose(g_lz4); g_lz4 = NULL; return 0; }
    return 1;
}

const char* lzfo_bench_liblz4_version(void) {
    if (!load_lz4()) return NULL;
    return p_version ? p_version() : "unknown";
}

static void* worker(void* arg) {
    work_t* w = (work_t*)arg;
    uint8_t* out = (uint8_t*)malloc(w->max_out + 64);
    lzfo_u32_table* table = (lzfo_u32_table*)malloc(sizeof(lzfo_u32_table));
    void* stream = (w->what == 2) ? p_createStream() : NULL;
    if (!out || !table) atomic_store(&w->failed, 1);
    else memset(out, 0, w->max_out + 64);          /* fault the pages in before the timed region */
    pthread_barrier_wait(w->bar);                   /* start of the timed region (the caller reads the clock) */
    if (out && table) for (;;) {
        const uint32_t i = atomic_fetch_add(&w->next, 1u);
        if (i >= w->n_blocks) break;
        size_t got = 0;
        int ok = 1;
        switch (w->what) {
        case 0:
            memset(table, 0, sizeof *table);        /* U32Table::default() per block (framed/compress.rs:270) */
            (void)lzfo_compress2(w->in[i], (size_t)w->in_len[i], 0, LZFO_TABLE_U32, table, out, (size_t)w->out_len[i], &got);
            /* (OutputFull = the frame layer stores the block raw: still a complete compress attempt) */
            break;
        case 1:
            ok = lzfo_decompress_raw(w->in[i], (size_t)w->in_len[i], NULL, 0, out, &got, (size_t)w->out_len[i] + 64, (size_t)w->out_len[i]) == LZFO_OK
                 && got == (size_t)w->out_len[i];
            break;
        case 2:
            p_resetStream(stream);                  /* fresh LZ4_stream_t per block */
            (void)p_fast_continue(stream, (const char*)w->in[i], (char*)out, (int)w->in_len[i], (int)w->max_out, 1);
            break;
        case 3:
            ok = p_decompress_safe((const char*)w->in[i], (char*)out, (int)w->in_len[i], (int)w->out_len[i]) == (int)w->out_len[i];
            break;
        }
        if (!ok) atomic_store(&w->failed, 1);
    }
    pthread_barrier_wait(w->bar);                   /* end of the timed region */
    if (stream) p_freeStream(stream);
    free(out); free(table);
    return NULL;
}

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* One measurement series.  what: 0/1 oracle compress/decompress, 2/3 liblz4 compress/decompress.
 * in[i]/in_len[i]: the block to process; out_len[i]: decompress: the exact decoded size; compress: the writer
 * capacity (= in_len for the framed contract).  seconds[reps] receives the wall time of every pass.
 * Returns 0, -1 when liblz4 is missing, -2 on a failed block / allocation. */
int lzfo_bench_run(int what, const uint8_t* const* in, const uint64_t* in_len, const uint64_t* out_len, uint32_t n_blocks,
                   uint32_t threads, uint32_t reps, double* seconds) {
    if (what >= 2 && !load_lz4()) return -1;
    if (threads == 0) threads = 1;
    uint64_t max_out = 0;
    for (uint32_t i = 0; i < n_blocks; ++i) {
        uint64_t need = what == 2 ? in_len[i] + in_len[i] / 255 + 64 : out_len[i];
        if (need > max_out) max_out = need;
    }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    if (!th) return -2;
    int rc = 0;
    for (uint32_t r = 0; r < reps; ++r) {
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, NULL, threads + 1);
        work_t w;
        w.what = what; w.in = in; w.in_len = in_len; w.out_len = out_len; w.n_blocks = n_blocks; w.max_out = max_out;
        atomic_init(&w.next, 0u); atomic_init(&w.failed, 0); w.bar = &bar;
        for (uint32_t t = 0; t < threads; ++t) pthread_create(&th[t], NULL, worker, &w);
        pthread_barrier_wait(&bar);
        const double t0 = now_s();
        pthread_barrier_wait(&bar);
        seconds[r] = now_s() - t0;
        for (uint32_t t = 0; t < threads; ++t) pthread_join(th[t], NULL);
        pthread_barrier_destroy(&bar);
        if (atomic_load(&w.failed)) rc = -2;
    }
    free(th);
    return rc;
}
