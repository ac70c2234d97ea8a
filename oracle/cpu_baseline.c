/* oracle/cpu_baseline.c — multi-threaded CPU timing harness + batch helpers around the CPU checkers.
 * TEST/BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg, tests' input preparation).
 *
 * kind "reference": dlopen()s oracle/_ref/libzstd_ref.so (the reference's own libzstd 1.5.7) and
 * drives ZSTD_compress2 / ZSTD_decompressDCtx exactly as zstd-jni's JNI glue does
 * (/root/reference/src/main/native/jni_fast_zstd.c:606-607, :798-799): one reused CCtx/DCtx per
 * thread, session reset before every call.   kind "port": the restatement in this directory.
 */
#define _GNU_SOURCE
#include "zstd_oracle.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    void* h;
    void* (*createCCtx)(void); size_t (*freeCCtx)(void*);
    size_t (*setParam)(void*, int, int); size_t (*reset)(void*, int);
    size_t (*compress2)(void*, void*, size_t, const void*, size_t);
    void* (*createDCtx)(void); size_t (*freeDCtx)(void*);
    size_t (*dreset)(void*, int);
    size_t (*decompressDCtx)(void*, void*, size_t, const void*, size_t);
    unsigned (*isError)(size_t);
} RefLib;

/* The handle is opened once and never dlclose()d: unloading a library that worker threads ran in
 * upsets tools that hook thread exit (rocprofv3 segfaulted on it). */
static int ref_open(RefLib* r, const char* path) {
    static void* cached = NULL;
    memset(r, 0, sizeof(*r));
    if (!path) return 0;
    if (!cached) cached = dlopen(path, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);     /* its own ZSTD_* first, whatever libzstd the process already has */
    r->h = cached;
    if (!r->h) return -1;
#define SYM(field, name) *(void**)(&r->field) = dlsym(r->h, name); if (!r->field) return -1;
    SYM(createCCtx, "ZSTD_createCCtx") SYM(freeCCtx, "ZSTD_freeCCtx") SYM(setParam, "ZSTD_CCtx_setParameter")
    SYM(reset, "ZSTD_CCtx_reset") SYM(compress2, "ZSTD_compress2") SYM(createDCtx, "ZSTD_createDCtx")
    SYM(freeDCtx, "ZSTD_freeDCtx") SYM(dreset, "ZSTD_DCtx_reset") SYM(decompressDCtx, "ZSTD_decompressDCtx")
    SYM(isError, "ZSTD_isError")
#undef SYM
    return 0;
}

typedef struct {
    const RefLib* lib; int level; int mode;          /* mode 0 compress, 1 decompress */
    const unsigned char* src; const size_t* srcOff;  /* n+1 offsets */
    unsigned char* dst; const size_t* dstOff;        /* n+1 offsets (capacity = diff) */
    size_t* outSize; size_t lo, hi; int failed;
} Job;

static void* worker(void* arg) {
    Job* j = (Job*)arg; size_t i;
    void* cctx = NULL; void* dctx = NULL;
    if (j->lib->h) {
        if (j->mode == 0) { cctx = j->lib->createCCtx(); j->lib->setParam(cctx, 100 /*ZSTD_c_compressionLevel*/, j->level); }
        else dctx = j->lib->createDCtx();
    }
    for (i = j->lo; i < j->hi; i++) {
        const unsigned char* s = j->src + j->srcOff[i]; size_t const sn = j->srcOff[i + 1] - j->srcOff[i];
        unsigned char* d = j->dst + j->dstOff[i]; size_t const dn = j->dstOff[i + 1] - j->dstOff[i];
        size_t r;
        if (j->lib->h) {
            if (j->mode == 0) { j->lib->reset(cctx, 1 /*session_only*/); r = j->lib->compress2(cctx, d, dn, s, sn); }
            else { j->lib->dreset(dctx, 1); r = j->lib->decompressDCtx(dctx, d, dn, s, sn); }
            if (j->lib->isError(r)) j->failed = 1;
        } else {
            r = (j->mode == 0) ? zso_compress(d, dn, s, sn, j->level, 0) : zso_decompress(d, dn, s, sn);
            if (zso_is_error(r)) j->failed = 1;
        }
        j->outSize[i] = r;
    }
    if (cctx) j->lib->freeCCtx(cctx);
    if (dctx) j->lib->freeDCtx(dctx);
    return NULL;
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* Runs one pass of `mode` over n buffers with `threads` threads (static contiguous partition).
 * Returns wall seconds, or -1 on failure. */
static double run_pass(const RefLib* lib, int mode, int level, const unsigned char* src, const size_t* srcOff,
                       unsigned char* dst, const size_t* dstOff, size_t* outSize, size_t n, int threads) {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    Job* jobs = (Job*)calloc((size_t)threads, sizeof(Job));
    int t, failed = 0; double t0, t1;
    t0 = now_s();
    for (t = 0; t < threads; t++) {
        Job* j = &jobs[t];
        j->lib = lib; j->level = level; j->mode = mode; j->src = src; j->srcOff = srcOff; j->dst = dst; j->dstOff = dstOff;
        j->outSize = outSize; j->lo = n * (size_t)t / (size_t)threads; j->hi = n * (size_t)(t + 1) / (size_t)threads;
        pthread_create(&th[t], NULL, worker, j);
    }
    for (t = 0; t < threads; t++) { pthread_join(th[t], NULL); failed |= jobs[t].failed; }
    t1 = now_s();
    free(th); free(jobs);
    return failed ? -1.0 : t1 - t0;
}

/* Batch compress (mode 0) / decompress (mode 1) helper for tests and bench input preparation.
 * libpath NULL -> the port.  Returns 0 or -1. */
int zso_batch(const char* libpath, int mode, int level, const void* src, const size_t* srcOff,
              void* dst, const size_t* dstOff, size_t* outSize, size_t n, int threads) {
    RefLib lib; double s;
    if (ref_open(&lib, libpath)) return -1;
    s = run_pass(&lib, mode, level, (const unsigned char*)src, srcOff, (unsigned char*)dst, dstOff, outSize, n, threads < 1 ? 1 : threads);
    return s < 0 ? -1 : 0;
}

/* CPU baseline: n buffers of bufSize bytes; warm-up pass + best of `reps` for compress and for
 * decompress.  out[0] compress seconds (best), out[1] decompress seconds (best), out[2] total
 * compressed bytes, out[3] 1.0 if every round trip was byte-exact.  Returns 0 or -1. */
int zso_cpu_baseline(const char* libpath, const void* data, size_t bufSize, size_t n, int level, int threads, int reps, double* out) {
    RefLib lib; size_t i; size_t const bound = zso_compress_bound(bufSize);
    size_t* srcOff = (size_t*)malloc(sizeof(size_t) * (n + 1)); size_t* cOff = (size_t*)malloc(sizeof(size_t) * (n + 1));
    size_t* cSize = (size_t*)malloc(sizeof(size_t) * n); size_t* dSize = (size_t*)malloc(sizeof(size_t) * n);
    size_t* pOff = (size_t*)malloc(sizeof(size_t) * (n + 1));
    unsigned char* comp = (unsigned char*)malloc(bound * n); unsigned char* packed = (unsigned char*)malloc(bound * n);
    unsigned char* back = (unsigned char*)malloc(bufSize * n);
    double bestC = 1e30, bestD = 1e30, s; int r, rc = -1; size_t total = 0;
    if (ref_open(&lib, libpath)) goto done;
    if (threads < 1) threads = 1;
    for (i = 0; i <= n; i++) { srcOff[i] = i * bufSize; cOff[i] = i * bound; }
    for (r = 0; r <= reps; r++) {            /* r == 0 is the warm-up */
        s = run_pass(&lib, 0, level, (const unsigned char*)data, srcOff, comp, cOff, cSize, n, threads);
        if (s < 0) goto done;
        if (r > 0 && s < bestC) bestC = s;
    }
    pOff[0] = 0;
    for (i = 0; i < n; i++) { memcpy(packed + pOff[i], comp + cOff[i], cSize[i]); pOff[i + 1] = pOff[i] + cSize[i]; }
    total = pOff[n];
    for (r = 0; r <= reps; r++) {
        s = run_pass(&lib, 1, level, packed, pOff, back, srcOff, dSize, n, threads);
        if (s < 0) goto done;
        if (r > 0 && s < bestD) bestD = s;
    }
    out[0] = bestC; out[1] = bestD; out[2] = (double)total; out[3] = (memcmp(back, data, bufSize * n) == 0) ? 1.0 : 0.0;
    rc = 0;
done:
    free(srcOff); free(cOff); free(cSize); free(dSize); free(pOff); free(comp); free(packed); free(back);
    return rc;
}

/* ---- round 2: persistent threads, barrier start -----------------------------------------------
 * zso_cpu_baseline's passes create their threads and contexts inside the timed window; with hundreds of host threads and a
 * few milliseconds of work each that measures thread start-up.  Here every thread exists, owns its reused CCtx / DCtx (and
 * shares one CDict / DDict when a dictionary is given) BEFORE the clock starts; a pass is released by a barrier, ends at a
 * second barrier, and passes repeat until `minSeconds` of timed work have been done (at least two); the best pass counts.
 *   srcOff: n+1 byte offsets into data, or NULL for n buffers of bufSize bytes
 *   out[0] best compress seconds, out[1] best decompress seconds, out[2] compressed bytes, out[3] 1.0 if byte-exact,
 *   out[4] / out[5] timed passes compress / decompress, out[6] / out[7] mean pass seconds compress / decompress */
typedef struct {
    void* (*createCDict)(const void*, size_t, int); size_t (*freeCDict)(void*); size_t (*refCDict)(void*, const void*);
    void* (*createDDict)(const void*, size_t); size_t (*freeDDict)(void*); size_t (*refDDict)(void*, const void*);
} DictFns;
typedef struct Pool {
    const RefLib* lib; DictFns df; void* cdict; void* ddict; int level, hashLog, chainLog;
    const unsigned char* src; const size_t* srcOff; unsigned char* dst; const size_t* dstOff; size_t* outSize; size_t n;
    int threads; volatile int mode;                 /* 0 compress, 1 decompress, -1 quit */
    pthread_barrier_t start, end; volatile int failed;
} Pool;
typedef struct { Pool* p; int id; } PoolArg;
static void* pool_worker(void* a) {
    Pool* p = ((PoolArg*)a)->p; int const id = ((PoolArg*)a)->id; size_t i;
    size_t const lo = p->n * (size_t)id / (size_t)p->threads, hi = p->n * (size_t)(id + 1) / (size_t)p->threads;
    void* cctx = p->lib->createCCtx(); void* dctx = p->lib->createDCtx();
    p->lib->setParam(cctx, 100 /*ZSTD_c_compressionLevel*/, p->level);
    if (p->hashLog) p->lib->setParam(cctx, 102 /*ZSTD_c_hashLog*/, p->hashLog);
    if (p->chainLog) p->lib->setParam(cctx, 103 /*ZSTD_c_chainLog*/, p->chainLog);
    if (p->cdict) p->df.refCDict(cctx, p->cdict);
    if (p->ddict) p->df.refDDict(dctx, p->ddict);
    for (;;) {
        pthread_barrier_wait(&p->start);
        if (p->mode < 0) break;
        for (i = lo; i < hi; i++) {
            const unsigned char* s = p->src + p->srcOff[i]; size_t const sn = p->srcOff[i + 1] - p->srcOff[i];
            unsigned char* d = p->dst + p->dstOff[i]; size_t const dn = p->dstOff[i + 1] - p->dstOff[i];
            size_t r;
            if (p->mode == 0) { p->lib->reset(cctx, 1 /*session_only: parameters and dictionary stay*/); r = p->lib->compress2(cctx, d, dn, s, sn); }
            else { p->lib->dreset(dctx, 1); r = p->lib->decompressDCtx(dctx, d, dn, s, sn); }
            if (p->lib->isError(r)) p->failed = 1;
            p->outSize[i] = r;
        }
        pthread_barrier_wait(&p->end);
    }
    p->lib->freeCCtx(cctx); p->lib->freeDCtx(dctx);
    return NULL;
}
static double pool_pass(Pool* p, int mode, const unsigned char* src, const size_t* srcOff, unsigned char* dst, const size_t* dstOff, size_t* outSize) {
    double t0, t1;
    p->mode = mode; p->src = src; p->srcOff = srcOff; p->dst = dst; p->dstOff = dstOff; p->outSize = outSize;
    pthread_barrier_wait(&p->start);
    t0 = now_s();
    pthread_barrier_wait(&p->end);
    t1 = now_s();
    return t1 - t0;
}
/* keepPacked / keepCap / keepSizes (optional): the reference's frames of the last compress pass, back to back, and their sizes — bench.py compares EVERY frame of
 * the GPU's batch with them (outside the timed region).  keepCap too small: nothing is copied and out[2] still says how many bytes there were. */
int zso_cpu_baseline3(const char* libpath, const void* data, const size_t* srcOffIn, size_t bufSize, size_t n, int level, int hashLog, int chainLog,
                      int threads, double minSeconds, const void* dict, size_t dictSize, double* out, unsigned char* keepPacked, size_t keepCap, size_t* keepSizes) {
    RefLib lib; Pool p; size_t i, total = 0, maxSrc = bufSize; int t, rc = -1, passes; double best, sum, s;
    size_t* srcOff = (size_t*)malloc(sizeof(size_t) * (n + 1)); size_t* cOff = (size_t*)malloc(sizeof(size_t) * (n + 1));
    size_t* cSize = (size_t*)malloc(sizeof(size_t) * n); size_t* dSize = (size_t*)malloc(sizeof(size_t) * n); size_t* pOff = (size_t*)malloc(sizeof(size_t) * (n + 1));
    unsigned char *comp = NULL, *packed = NULL, *back = NULL; pthread_t* th = NULL; PoolArg* args = NULL;
    memset(&p, 0, sizeof p);
    if (!libpath || ref_open(&lib, libpath)) goto done;                /* the reference library only: the port has no reusable contexts */
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = (int)n;
    for (i = 0; i <= n; i++) srcOff[i] = srcOffIn ? srcOffIn[i] : i * bufSize;
    for (i = 0; i < n; i++) if (srcOff[i + 1] - srcOff[i] > maxSrc) maxSrc = srcOff[i + 1] - srcOff[i];
    cOff[0] = 0; for (i = 0; i < n; i++) cOff[i + 1] = cOff[i] + zso_compress_bound(srcOff[i + 1] - srcOff[i]);
    comp = (unsigned char*)malloc(cOff[n] + 16); back = (unsigned char*)malloc(srcOff[n] + 16);
    if (!comp || !back) goto done;
    p.lib = &lib; p.level = level; p.hashLog = hashLog; p.chainLog = chainLog; p.n = n; p.threads = threads;
    if (dict && dictSize) {
#define DSYM(field, name) *(void**)(&p.df.field) = dlsym(lib.h, name); if (!p.df.field) goto done;
        DSYM(createCDict, "ZSTD_createCDict") DSYM(freeCDict, "ZSTD_freeCDict") DSYM(refCDict, "ZSTD_CCtx_refCDict")
        DSYM(createDDict, "ZSTD_createDDict") DSYM(freeDDict, "ZSTD_freeDDict") DSYM(refDDict, "ZSTD_DCtx_refDDict")
#undef DSYM
        p.cdict = p.df.createCDict(dict, dictSize, level); p.ddict = p.df.createDDict(dict, dictSize);
        if (!p.cdict || !p.ddict) goto done;
    }
    pthread_barrier_init(&p.start, NULL, (unsigned)threads + 1); pthread_barrier_init(&p.end, NULL, (unsigned)threads + 1);
    th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads); args = (PoolArg*)malloc(sizeof(PoolArg) * (size_t)threads);
    for (t = 0; t < threads; t++) { args[t].p = &p; args[t].id = t; pthread_create(&th[t], NULL, pool_worker, &args[t]); }
    /* compress: one warm-up pass, then timed passes */
    pool_pass(&p, 0, (const unsigned char*)data, srcOff, comp, cOff, cSize);
    for (passes = 0, best = 1e30, sum = 0; passes < 2 || sum < minSeconds; passes++) {
        s = pool_pass(&p, 0, (const unsigned char*)data, srcOff, comp, cOff, cSize);
        sum += s; if (s < best) best = s;
        if (passes > 1000) break;
    }
    out[0] = best; out[4] = passes; out[6] = sum / passes;
    pOff[0] = 0; for (i = 0; i < n; i++) pOff[i + 1] = pOff[i] + (lib.isError(cSize[i]) ? 0 : cSize[i]);
    total = pOff[n];
    packed = (unsigned char*)malloc(total + 16);
    if (!packed || p.failed) { p.failed = 1; }
    else {
        for (i = 0; i < n; i++) memcpy(packed + pOff[i], comp + cOff[i], pOff[i + 1] - pOff[i]);
        if (keepPacked && keepCap >= total) memcpy(keepPacked, packed, total);
        if (keepSizes) for (i = 0; i < n; i++) keepSizes[i] = cSize[i];
        pool_pass(&p, 1, packed, pOff, back, srcOff, dSize);
        for (passes = 0, best = 1e30, sum = 0; passes < 2 || sum < minSeconds; passes++) {
            s = pool_pass(&p, 1, packed, pOff, back, srcOff, dSize);
            sum += s; if (s < best) best = s;
            if (passes > 1000) break;
        }
        out[1] = best; out[5] = passes; out[7] = sum / passes;
    }
    p.mode = -1; pthread_barrier_wait(&p.start);
    for (t = 0; t < threads; t++) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&p.start); pthread_barrier_destroy(&p.end);
    if (!p.failed) { out[2] = (double)total; out[3] = (memcmp(back, data, srcOff[n]) == 0) ? 1.0 : 0.0; rc = 0; }
done:
    if (p.cdict) p.df.freeCDict(p.cdict);
    if (p.ddict) p.df.freeDDict(p.ddict);
    free(srcOff); free(cOff); free(cSize); free(dSize); free(pOff); free(comp); free(packed); free(back); free(th); free(args);
    return rc;
}
int zso_cpu_baseline2(const char* libpath, const void* data, const size_t* srcOffIn, size_t bufSize, size_t n, int level, int hashLog, int chainLog,
                      int threads, double minSeconds, const void* dict, size_t dictSize, double* out) {
    return zso_cpu_baseline3(libpath, data, srcOffIn, bufSize, n, level, hashLog, chainLog, threads, minSeconds, dict, dictSize, out, NULL, 0, NULL);
}
