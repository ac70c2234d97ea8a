/* zjni_shim.c — the JNI side of the drop-in: libzstd-jni's natives, with the one-shot hot path bound to libzjni_amd.so.
 *
 * zstd-jni loads ONE native library (J/util/Native.java:90-180) and its Java classes call per-buffer natives
 * (reference src/main/native/jni_fast_zstd.c, jni_zstd.c, "N/").  This library exports every
 * Java_com_github_luben_zstd_* symbol the reference's library exports (149; tests/test_jni_shim.py diffs the two
 * symbol tables) plus three batch natives:
 *   - the one-shot hot path (ZstdCompressCtx.compress*0, ZstdDecompressCtx.decompress*0, Zstd.compressUnsafe /
 *     decompressUnsafe, the dictionary classes, the parameter natives that shape a one-shot frame) is defined HERE
 *     and sends buffers to the GPU library through its C-ABI (include/zjni_amd.h) instead of calling ZSTD_compress2 /
 *     ZSTD_decompressDCtx.  Argument checks, their order and the returned codes are the reference's
 *     (N/jni_fast_zstd.c:586-640, :777-905; N/jni_zstd.c:50-63, :230-267);
 *   - the stream classes (DirectByteBuffer and heap-array, sections at the end of this file) buffer a stream and make its frame with one zjni_compress_stream call;
 *   - frame inspection (content size, frame extent, dictionary ids) and the constants of class Zstd are host-side code at the end of this file;
 *   - what is left (context streams, pledged size / progression, training) is a trampoline (zjni_forward.c) into the bundled CPU
 *     library named by $ZSTD_JNI_CPU_LIB, looked up with dlsym — the CPU code stays where it is, this file contains none.
 *
 * Handles.  A context's nativePtr is the BUNDLED library's own ZSTD_CCtx* / ZSTD_DCtx* whenever that library is present,
 * so a native this file does not define reaches the bundled library with the pointer it expects; what the GPU path
 * needs to know about a context (level, flags, table sizes, dictionary digest) lives in a side table keyed by that
 * pointer — the arrangement the dictionary classes always had.  Without the bundled library the handle is a private
 * token and the trampolines return 0.
 *
 * What the GPU path does not take (levels > 3, inputs > 128 KiB, parameters it cannot honour such as windowLog,
 * strategy or magicless frames, byte[] dictionaries on the compress side) is forwarded to the bundled library's native of
 * the same name.  Without that library such calls return the zjni error code.
 *
 * Built against the JDK's <jni.h>; this image has no JDK, so the build (Makefile next to this file) uses the copy
 * zstd-jni vendors under /root/reference/jni when it is present and the prebuilt .so travels to the GPU box.
 * Heap-array natives copy through Get/SetByteArrayRegion instead of pinning with GetPrimitiveArrayCritical: a GPU
 * round trip inside a critical section would stall the collector (SURVEY.md §8b "Ownership").
 */
#include <jni.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "../../include/zjni_amd.h"

#define P(name) Java_com_github_luben_zstd_##name
#define PS(name) "Java_com_github_luben_zstd_" name
#define E_DST ((jlong)-70)   /* -ZSTD_error_dstSize_tooSmall */
#define E_SRC ((jlong)-72)   /* -ZSTD_error_srcSize_wrong */
#define E_MEM ((jlong)-64)   /* -ZSTD_error_memory_allocation */
#define E_DICT ((jlong)-32)  /* -ZSTD_error_dictionary_wrong */

/* ---- the bundled CPU library (optional) ------------------------------------------------------------- */
static void* g_cpu;
static pthread_once_t g_cpu_once = PTHREAD_ONCE_INIT;
static void cpu_open(void) {
    const char* p = getenv("ZSTD_JNI_CPU_LIB");
    if (p && *p) g_cpu = dlopen(p, RTLD_NOW | RTLD_LOCAL);
}
void* zjni_shim_cpu_sym(const char* name) {                  /* also used by the trampolines (zjni_forward.c) */
    pthread_once(&g_cpu_once, cpu_open);
    return g_cpu ? dlsym(g_cpu, name) : NULL;
}
#define cpu_sym zjni_shim_cpu_sym
static int gpu_on(void) {
    static int state = -1;
    if (state < 0) state = (zjni_device_count() > 0 && zjni_init(0) == 0) ? 1 : 0;
    return state;
}
/* A single buffer cannot amortise a GPU launch (a 64 KiB buffer takes milliseconds on a device built for 65 536 at a
 * time), so when the bundled CPU library is available the per-buffer natives go there and the GPU serves the batch
 * natives; ZSTD_JNI_GPU_PER_BUFFER=1 (or no CPU library, as in the tests) sends per-buffer calls to the GPU as well. */
static int per_buffer_on_gpu(void) {
    static int state = -1;
    if (state < 0) { const char* e = getenv("ZSTD_JNI_GPU_PER_BUFFER"); state = ((e && *e == '1') || !cpu_sym(PS("Zstd_compressBound"))) ? 1 : 0; }
    return state && gpu_on();
}
/* ZSTD_JNI_GPU_AGGREGATE=<microseconds>: concurrent per-buffer calls of plain contexts (no dictionary, default frame layout, no explicit
 * table sizes) are batched across threads (zjni_aggregator_*): the first caller waits that long for company.
 * Unset / 0: every per-buffer call is a batch of one. */
static zjni_aggregator* g_agg;
static pthread_once_t g_agg_once = PTHREAD_ONCE_INIT;
static void agg_open(void) {
    const char* e = getenv("ZSTD_JNI_GPU_AGGREGATE"); long const us = e ? atol(e) : 0;
    g_agg = us > 0 ? zjni_createAggregator(0, 4096, (unsigned)us) : NULL;
}
static zjni_aggregator* aggregator(void) { pthread_once(&g_agg_once, agg_open); return g_agg; }   /* (two first callers used to be able to make two) */

/* ---- who served a call: the integrator cannot see it from Java --------------------------------------
 * zjni_shim_stats(out[4]): hot-path natives [0] answered by the GPU path (a size or a genuine libzstd error code), [1] sent to the bundled
 * library by policy without asking the GPU (per-buffer natives while ZSTD_JNI_CPU_LIB is loaded and ZSTD_JNI_GPU_PER_BUFFER is not 1; contexts
 * carrying a parameter or dictionary only the CPU understands; sizes outside the GPU path), [2] sent there after the GPU path answered 200 / 201 /
 * 40 / 42 ("not mine"), [3] of those, the answers 200 (no device, launch failure).  Relaxed atomics: counters, not a ledger. */
static unsigned long long g_stats[4];
static __thread int t_gpu_declined;                              /* the GPU path has just said "not mine" on this thread: the forward that follows is [2], not [1] */
JNIEXPORT void zjni_shim_stats(unsigned long long* out4) { int i; for (i = 0; i < 4; i++) out4[i] = __atomic_load_n(&g_stats[i], __ATOMIC_RELAXED); }
static int stat_verdict(int final, size_t r) {
    if (final) __atomic_fetch_add(&g_stats[0], 1, __ATOMIC_RELAXED);
    else {
        t_gpu_declined = 1;
        __atomic_fetch_add(&g_stats[2], 1, __ATOMIC_RELAXED);
        if (zjni_isError(r) && zjni_getErrorCode(r) == ZJNI_ERROR_no_device) __atomic_fetch_add(&g_stats[3], 1, __ATOMIC_RELAXED);
    }
    return final;
}
static void stat_forward(void) {                                  /* a hot-path native is about to call the bundled library */
    if (t_gpu_declined) t_gpu_declined = 0; else __atomic_fetch_add(&g_stats[1], 1, __ATOMIC_RELAXED);
}
static int result_final(size_t r) {           /* sizes and genuine libzstd error codes are final; 200/201 mean "not for the GPU path" */
    return !(zjni_isError(r) && zjni_getErrorCode(r) >= 200);
}
static int gpu_result_final(size_t r) { return stat_verdict(result_final(r), r); }

/* ---- per-context state of the GPU path, keyed by nativePtr ------------------------------------------ */
typedef struct CtxState {
    struct CtxState* next; jlong key; int kind;            /* kind 'C' compress, 'D' decompress */
    /* compress: what ZSTD_CCtx_setParameter calls have said so far (defaults = ZSTD_CCtx_reset's) */
    int level, checksum, contentSize, dictIDFlag, hashLog, chainLog;
    int cpuOnly;                                           /* a parameter the GPU path cannot honour is set (windowLog, strategy, magicless, workers, ...) */
    int cpuDict;                                           /* a dictionary is loaded that has no GPU digest (byte[] dictionary, level > 3, digest table full, ...) */
    zjni_cdict* cdict;                                     /* digest of the loaded ZstdDictCompress (owned by the dictionary object) */
    zjni_ddict* ddict;                                     /* digest of the loaded ZstdDictDecompress (owned by the dictionary object) ... */
    zjni_ddict* ddictOwned;                                /* ... or of the bytes given to loadDDict0 (owned by the context) */
    /* ZstdCompressCtx.loadDict(byte[]) = ZSTD_CCtx_loadDictionary: the reference keeps the bytes and digests them into a CDict of its own at
     * the first compress call, with the level set by then, and keeps that CDict until the dictionary is replaced (ZSTD_initLocalDict,
     * N/compress/zstd_compress.c:1246-1290) — frames equal ZSTD_CCtx_refCDict's (checked against the reference: every size up to one block). */
    void* rawDict; size_t rawDictSize; zjni_cdict* localCdict;
    /* the context used as a stream (compress*Stream0 / decompressDirectByteBufferStream0, "the context streams" near the end of this file) */
    struct StreamState* cx;                                /* compress: the frame being written (NULL: none yet) */
    int hasPledged; unsigned long long pledged;            /* ZSTD_CCtx_setPledgedSrcSize for the next frame */
    int cpuInFrame;                                        /* compress: a frame is being written by the bundled library's context (calls passed on as they came) */
    int dInFrame;                                          /* decompress: the bundled library's stream is inside a frame */
} CtxState;
static void cx_drop(CtxState* c);
static void st_drop_local_dict(CtxState* s) {
    if (s->localCdict) { zjni_freeCDict(s->localCdict); s->localCdict = NULL; }
    if (s->rawDict) { free(s->rawDict); s->rawDict = NULL; s->rawDictSize = 0; }
}
#define ST_BUCKETS 4096
static CtxState* g_st[ST_BUCKETS];
static pthread_mutex_t g_st_mu = PTHREAD_MUTEX_INITIALIZER;
static unsigned st_bucket(jlong key) { uint64_t x = (uint64_t)key; x ^= x >> 17; x *= 0x9E3779B97F4A7C15ull; return (unsigned)(x >> 52) & (ST_BUCKETS - 1); }
static void st_defaults(CtxState* s) {
    s->level = 3; s->checksum = 0; s->contentSize = 1; s->dictIDFlag = 1; s->hashLog = 0; s->chainLog = 0; s->cpuOnly = 0; s->cpuDict = 0; s->cdict = NULL; s->ddict = NULL;
}
static CtxState* st_new(jlong key, int kind) {
    CtxState* s = (CtxState*)calloc(1, sizeof(CtxState));
    if (!s) return NULL;
    s->key = key; s->kind = kind; st_defaults(s);
    pthread_mutex_lock(&g_st_mu);
    s->next = g_st[st_bucket(key)]; g_st[st_bucket(key)] = s;
    pthread_mutex_unlock(&g_st_mu);
    return s;
}
/* A context is used by one thread at a time (J/ZstdCompressCtx.java:32-34), so the state itself needs no lock — the table does. */
static CtxState* st_get(jlong key, int kind) {
    CtxState* s;
    if (!key) return NULL;
    pthread_mutex_lock(&g_st_mu);
    for (s = g_st[st_bucket(key)]; s && !(s->key == key && s->kind == kind); s = s->next) {}
    pthread_mutex_unlock(&g_st_mu);
    return s;
}
static CtxState* st_take(jlong key, int kind) {
    CtxState** pp; CtxState* s = NULL;
    pthread_mutex_lock(&g_st_mu);
    for (pp = &g_st[st_bucket(key)]; *pp; pp = &(*pp)->next) if ((*pp)->key == key && (*pp)->kind == kind) { s = *pp; *pp = s->next; break; }
    pthread_mutex_unlock(&g_st_mu);
    return s;
}

/* ---- context lifecycle (N/jni_fast_zstd.c:253-270, :651-668) --------------------------------------- */
static jlong ctx_init(JNIEnv* env, jclass cls, const char* name, int kind) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(name);
    jlong h = f ? f(env, cls) : (jlong)(intptr_t)calloc(1, 16);      /* the bundled library's handle, or a private token */
    if (h && !st_new(h, kind)) { /* no state: the context works, on the CPU path only */ }
    return h;
}
static void ctx_free(JNIEnv* env, jclass cls, jlong ptr, const char* name, int kind) {
    void (*f)(JNIEnv*, jclass, jlong) = (void (*)(JNIEnv*, jclass, jlong))cpu_sym(name);
    CtxState* s = st_take(ptr, kind);
    if (s) { if (s->ddictOwned) zjni_freeDDict(s->ddictOwned); st_drop_local_dict(s); cx_drop(s); free(s); }
    if (!ptr) return;
    if (f) f(env, cls, ptr); else free((void*)(intptr_t)ptr);
}
JNIEXPORT jlong JNICALL P(ZstdCompressCtx_init)(JNIEnv* env, jclass cls) { return ctx_init(env, cls, PS("ZstdCompressCtx_init"), 'C'); }
JNIEXPORT void JNICALL P(ZstdCompressCtx_free)(JNIEnv* env, jclass cls, jlong ptr) { ctx_free(env, cls, ptr, PS("ZstdCompressCtx_free"), 'C'); }
JNIEXPORT jlong JNICALL P(ZstdDecompressCtx_init)(JNIEnv* env, jclass cls) { return ctx_init(env, cls, PS("ZstdDecompressCtx_init"), 'D'); }
JNIEXPORT void JNICALL P(ZstdDecompressCtx_free)(JNIEnv* env, jclass cls, jlong ptr) { ctx_free(env, cls, ptr, PS("ZstdDecompressCtx_free"), 'D'); }

/* ---- parameters: recorded for the GPU path, and passed on to the bundled library's context -------- */
#define FWD_VOID(name, T, ptr, v) do { void (*f_)(JNIEnv*, jclass, jlong, T) = (void (*)(JNIEnv*, jclass, jlong, T))cpu_sym(PS(name)); if (f_) f_(env, cls, ptr, v); } while (0)
JNIEXPORT void JNICALL P(ZstdCompressCtx_setLevel0)(JNIEnv* env, jclass cls, jlong ptr, jint level) {
    CtxState* s = st_get(ptr, 'C'); if (s) s->level = level;
    FWD_VOID("ZstdCompressCtx_setLevel0", jint, ptr, level);
}
JNIEXPORT void JNICALL P(ZstdCompressCtx_setChecksum0)(JNIEnv* env, jclass cls, jlong ptr, jboolean flag) {
    CtxState* s = st_get(ptr, 'C'); if (s) s->checksum = (flag == JNI_TRUE);
    FWD_VOID("ZstdCompressCtx_setChecksum0", jboolean, ptr, flag);
}
JNIEXPORT void JNICALL P(ZstdCompressCtx_setContentSize0)(JNIEnv* env, jclass cls, jlong ptr, jboolean flag) {     /* N/jni_fast_zstd.c:301-308 */
    CtxState* s = st_get(ptr, 'C'); if (s) s->contentSize = (flag == JNI_TRUE);
    FWD_VOID("ZstdCompressCtx_setContentSize0", jboolean, ptr, flag);
}
JNIEXPORT void JNICALL P(ZstdCompressCtx_setDictID0)(JNIEnv* env, jclass cls, jlong ptr, jboolean flag) {          /* N/jni_fast_zstd.c:313-320 */
    CtxState* s = st_get(ptr, 'C'); if (s) s->dictIDFlag = (flag == JNI_TRUE);
    FWD_VOID("ZstdCompressCtx_setDictID0", jboolean, ptr, flag);
}
/* ZSTD_CCtx_reset(session_and_parameters) (N/jni_fast_zstd.c:364-368): parameters back to their defaults, dictionary dropped */
JNIEXPORT jlong JNICALL P(ZstdCompressCtx_reset0)(JNIEnv* env, jclass cls, jlong ptr) {
    jlong (*f)(JNIEnv*, jclass, jlong) = (jlong (*)(JNIEnv*, jclass, jlong))cpu_sym(PS("ZstdCompressCtx_reset0"));
    CtxState* s = st_get(ptr, 'C'); if (s) { st_defaults(s); st_drop_local_dict(s); cx_drop(s); s->hasPledged = 0; s->cpuInFrame = 0; }      /* ZSTD_reset_session_and_parameters clears the dictionaries too, and ends a frame being streamed */
    return f ? f(env, cls, ptr) : 0;
}
/* ZSTD_DCtx_reset(session_and_parameters) (N/jni_fast_zstd.c:712-716) */
JNIEXPORT jlong JNICALL P(ZstdDecompressCtx_reset0)(JNIEnv* env, jclass cls, jlong ptr) {
    jlong (*f)(JNIEnv*, jclass, jlong) = (jlong (*)(JNIEnv*, jclass, jlong))cpu_sym(PS("ZstdDecompressCtx_reset0"));
    CtxState* s = st_get(ptr, 'D');
    if (s) { if (s->ddictOwned) { zjni_freeDDict(s->ddictOwned); s->ddictOwned = NULL; } s->ddict = NULL; s->cpuOnly = 0; s->cpuDict = 0; s->dInFrame = 0; cx_drop(s); }
    return f ? f(env, cls, ptr) : 0;
}
/* class Zstd's parameter natives take a raw context pointer (N/jni_zstd.c:349-570); it may be one of the contexts above or a stream
 * class's.  Recorded when the GPU path honours the parameter, otherwise the context is marked for the CPU path; always passed on. */
static int ss_note_parameter(jlong stream, int what, jint v);       /* the handle may be a stream class's (below): 1 when it is and the stream route honours the parameter, -1 when it is inside a frame that cannot take it */
static jint zstd_setter(JNIEnv* env, jclass cls, jlong stream, jint v, const char* name, int what) {
    jint (*f)(JNIEnv*, jclass, jlong, jint) = (jint (*)(JNIEnv*, jclass, jlong, jint))cpu_sym(name);
    CtxState* s = st_get(stream, what == 'd' ? 'D' : 'C');
    int const streamOk = ss_note_parameter(stream, what, v);
    if (streamOk < 0) return -(jint)60;                 /* ZSTD_error_stage_wrong, as ZSTD_CCtx_setParameter answers past the init stage; nothing is forwarded */
    if (streamOk == 2) return v == 0 ? 3 : (v > 0 ? v : 0);   /* a level inside a frame made here: the next frame's — the bundled context hears of it when this frame is out or replayed (ss_forward_level), ZSTD_CCtx_setParameter's own answer meanwhile */
    if (s) switch (what) {
        case 'l': s->level = v; break;
        case 'k': s->checksum = (v & 0xFF) != 0; break;
        case 'h': s->hashLog = v; break;
        case 'c': s->chainLog = v; break;
        default: s->cpuOnly = 1; break;                    /* 'x' / 'd': the GPU path does not implement it */
    }
    if (f) return f(env, cls, stream, v);
    if (!((s && what != 'x' && what != 'd') || streamOk)) return -(jint)ZJNI_ERROR_unsupported;
    /* no bundled library: ZSTD_CCtx_setParameter's own answer, the value now in force (C/zstd_compress.c ZSTD_CCtxParams_setParameter: the level — 3 for 0, 0 for a negative one —, the flag, the log) */
    if (what == 'l') return v == 0 ? 3 : (v > 0 ? v : 0);
    if (what == 'k') return (v & 0xFF) != 0;
    return v > 0 ? v : 0;
}
#define SETTER(name, T, what) JNIEXPORT jint JNICALL P(name)(JNIEnv* env, jclass cls, jlong stream, T v) { return zstd_setter(env, cls, stream, (jint)v, PS(#name), what); }
SETTER(Zstd_setCompressionLevel, jint, 'l')
SETTER(Zstd_setCompressionChecksums, jboolean, 'k')
SETTER(Zstd_setCompressionHashLog, jint, 'h')              /* ZstdCompressCtx.setHashLog: ZSTD_c_hashLog */
SETTER(Zstd_setCompressionChainLog, jint, 'c')             /* ZstdCompressCtx.setChainLog: ZSTD_c_chainLog */
SETTER(Zstd_setCompressionMagicless, jboolean, 'x')
SETTER(Zstd_setCompressionLong, jint, 'x')
SETTER(Zstd_setCompressionWorkers, jint, 'x')
SETTER(Zstd_setCompressionJobSize, jint, 'x')
SETTER(Zstd_setCompressionOverlapLog, jint, 'x')
SETTER(Zstd_setCompressionWindowLog, jint, 'x')
SETTER(Zstd_setCompressionSearchLog, jint, 'x')
SETTER(Zstd_setCompressionMinMatch, jint, 'x')
SETTER(Zstd_setCompressionTargetLength, jint, 'x')
SETTER(Zstd_setCompressionStrategy, jint, 'x')
SETTER(Zstd_setEnableLongDistanceMatching, jint, 'x')
SETTER(Zstd_setValidateSequences, jint, 'x')
SETTER(Zstd_setSequenceProducerFallback, jboolean, 'x')
SETTER(Zstd_setSearchForExternalRepcodes, jint, 'x')
SETTER(Zstd_setDecompressionMagicless, jboolean, 'd')
SETTER(Zstd_setDecompressionLongMax, jint, 'd')            /* ZSTD_d_windowLogMax (N/jni_zstd.c:400-404): a limit the GPU path does not apply, so the context decodes on the CPU */
SETTER(Zstd_setRefMultipleDDicts, jboolean, 'd')           /* ZSTD_d_refMultipleDDicts (N/jni_zstd.c:522-526): the dictionary set lives in the bundled library's context */
JNIEXPORT void JNICALL P(Zstd_registerSequenceProducer)(JNIEnv* env, jclass cls, jlong stream, jlong state, jlong fn) {
    void (*f)(JNIEnv*, jclass, jlong, jlong, jlong) = (void (*)(JNIEnv*, jclass, jlong, jlong, jlong))cpu_sym(PS("Zstd_registerSequenceProducer"));
    CtxState* s = st_get(stream, 'C'); if (s) s->cpuOnly = (fn != 0);
    if (f) f(env, cls, stream, state, fn);
}

/* ---- ZstdDictCompress / ZstdDictDecompress (N/jni_fast_zstd.c:13-125) --------------------------------
 * The Java object keeps ONE long (nativePtr), and the bundled library's natives read it as their own ZSTD_CDict* / ZSTD_DDict*.
 * So nativePtr stays what the bundled library put there (when it is present), and the GPU digest of the same
 * dictionary lives in a side table keyed by that value; without the bundled library nativePtr is the table key
 * itself (a private handle). */
typedef struct { jlong key; void* gpu; } DictEnt;                       /* gpu: zjni_cdict* or zjni_ddict* (keys are distinct heap addresses) */
#define DICT_MAX 4096
static DictEnt g_dicts[DICT_MAX];
static pthread_mutex_t g_dict_mu = PTHREAD_MUTEX_INITIALIZER;
static jfieldID g_cdict_field, g_ddict_field;                           /* written once each (the value is the same whoever writes it) */
static jfieldID native_ptr_field(JNIEnv* env, jobject obj, jfieldID* cache) {
    jfieldID f = __atomic_load_n(cache, __ATOMIC_ACQUIRE);
    if (!f) { f = (*env)->GetFieldID(env, (*env)->GetObjectClass(env, obj), "nativePtr", "J"); __atomic_store_n(cache, f, __ATOMIC_RELEASE); }
    return f;
}
static int dict_put(jlong key, void* gpu) {                       /* 0 when the table is full (the caller keeps the CPU path for this dictionary) */
    int ok = 0;
    pthread_mutex_lock(&g_dict_mu);
    for (int i = 0; i < DICT_MAX; i++) if (!g_dicts[i].key) { g_dicts[i].key = key; g_dicts[i].gpu = gpu; ok = 1; break; }
    pthread_mutex_unlock(&g_dict_mu);
    return ok;
}
static void* dict_get(jlong key, int remove) {
    void* r = NULL;
    pthread_mutex_lock(&g_dict_mu);
    for (int i = 0; i < DICT_MAX; i++) if (g_dicts[i].key == key && key) { r = g_dicts[i].gpu; if (remove) { g_dicts[i].key = 0; g_dicts[i].gpu = NULL; } break; }
    pthread_mutex_unlock(&g_dict_mu);
    return r;
}
/* after the bundled library's init: attach the GPU digest to whatever handle the object now carries */
static void dict_register(JNIEnv* env, jobject obj, jfieldID field, void* gpu, int compressSide, int bundledInit) {
    jlong key = (*env)->GetLongField(env, obj, field);
    if (!key) {
        /* nativePtr is 0.  With a bundled init that means ZSTD_createCDict / DDict FAILED there: Java's `nativePtr == 0` check must see it, and the
         * trampolines hand nativePtr to the bundled library as its own pointer type — no private handle may go in.  Only without any bundled
         * init (the GPU-only configuration) is the handle ours. */
        if (bundledInit || !gpu) {
            if (gpu) { if (compressSide) zjni_freeCDict((zjni_cdict*)gpu); else zjni_freeDDict((zjni_ddict*)gpu); }
            return;
        }
        key = (jlong)(intptr_t)gpu;
        (*env)->SetLongField(env, obj, field, key);
    }
    if (gpu && !dict_put(key, gpu)) {
        if (compressSide) zjni_freeCDict((zjni_cdict*)gpu); else zjni_freeDDict((zjni_ddict*)gpu);
        if (key == (jlong)(intptr_t)gpu) (*env)->SetLongField(env, obj, field, 0);
    }
}
static void* dict_digest(const void* bytes, size_t size, jint level, int compressSide) {
    if (!gpu_on()) return NULL;
    if (compressSide) return (level >= 1 && level <= 3) ? (void*)zjni_createCDict(bytes, size, level) : NULL;
    return (void*)zjni_createDDict(bytes, size);
}
JNIEXPORT void JNICALL P(ZstdDictCompress_init)(JNIEnv* env, jobject obj, jbyteArray dict, jint dict_offset, jint dict_size, jint level) {
    void (*f)(JNIEnv*, jobject, jbyteArray, jint, jint, jint) = (void (*)(JNIEnv*, jobject, jbyteArray, jint, jint, jint))cpu_sym(PS("ZstdDictCompress_init"));
    jfieldID const field = native_ptr_field(env, obj, &g_cdict_field);
    if (NULL == dict) return;
    if (f) f(env, obj, dict, dict_offset, dict_size, level);
    if (dict_size >= 0) {
        jbyte* copy = (jbyte*)malloc((size_t)dict_size + 1);
        if (!copy) return;
        (*env)->GetByteArrayRegion(env, dict, dict_offset, dict_size, copy);
        if ((*env)->ExceptionCheck(env)) { free(copy); return; }        /* region out of bounds: nothing was copied, nothing to digest */
        dict_register(env, obj, field, dict_digest(copy, (size_t)dict_size, level, 1), 1, f != NULL);
        free(copy);
    }
}
JNIEXPORT void JNICALL P(ZstdDictCompress_initDirect)(JNIEnv* env, jobject obj, jobject dict, jint dict_offset, jint dict_size, jint level, jint byReference) {
    void (*f)(JNIEnv*, jobject, jobject, jint, jint, jint, jint) = (void (*)(JNIEnv*, jobject, jobject, jint, jint, jint, jint))cpu_sym(PS("ZstdDictCompress_initDirect"));
    jfieldID const field = native_ptr_field(env, obj, &g_cdict_field);
    if (NULL == dict) return;
    if (f) f(env, obj, dict, dict_offset, dict_size, level, byReference);
    {   char* p = (char*)(*env)->GetDirectBufferAddress(env, dict);
        if (p && dict_size >= 0) dict_register(env, obj, field, dict_digest(p + dict_offset, (size_t)dict_size, level, 1), 1, f != NULL); }   /* the device keeps its own copy either way */
}
JNIEXPORT void JNICALL P(ZstdDictCompress_free)(JNIEnv* env, jobject obj) {
    void (*f)(JNIEnv*, jobject) = (void (*)(JNIEnv*, jobject))cpu_sym(PS("ZstdDictCompress_free"));
    zjni_cdict* gpu = (zjni_cdict*)dict_get((*env)->GetLongField(env, obj, native_ptr_field(env, obj, &g_cdict_field)), 1);
    if (gpu) zjni_freeCDict(gpu);
    if (f) f(env, obj);
}
JNIEXPORT void JNICALL P(ZstdDictDecompress_init)(JNIEnv* env, jobject obj, jbyteArray dict, jint dict_offset, jint dict_size) {
    void (*f)(JNIEnv*, jobject, jbyteArray, jint, jint) = (void (*)(JNIEnv*, jobject, jbyteArray, jint, jint))cpu_sym(PS("ZstdDictDecompress_init"));
    jfieldID const field = native_ptr_field(env, obj, &g_ddict_field);
    if (NULL == dict) return;
    if (f) f(env, obj, dict, dict_offset, dict_size);
    if (dict_size >= 0) {
        jbyte* copy = (jbyte*)malloc((size_t)dict_size + 1);
        if (!copy) return;
        (*env)->GetByteArrayRegion(env, dict, dict_offset, dict_size, copy);
        if ((*env)->ExceptionCheck(env)) { free(copy); return; }
        dict_register(env, obj, field, dict_digest(copy, (size_t)dict_size, 0, 0), 0, f != NULL);
        free(copy);
    }
}
JNIEXPORT void JNICALL P(ZstdDictDecompress_initDirect)(JNIEnv* env, jobject obj, jobject dict, jint dict_offset, jint dict_size, jint byReference) {
    void (*f)(JNIEnv*, jobject, jobject, jint, jint, jint) = (void (*)(JNIEnv*, jobject, jobject, jint, jint, jint))cpu_sym(PS("ZstdDictDecompress_initDirect"));
    jfieldID const field = native_ptr_field(env, obj, &g_ddict_field);
    if (NULL == dict) return;
    if (f) f(env, obj, dict, dict_offset, dict_size, byReference);
    {   char* p = (char*)(*env)->GetDirectBufferAddress(env, dict);
        if (p && dict_size >= 0) dict_register(env, obj, field, dict_digest(p + dict_offset, (size_t)dict_size, 0, 0), 0, f != NULL); }
}
JNIEXPORT void JNICALL P(ZstdDictDecompress_free)(JNIEnv* env, jobject obj) {
    void (*f)(JNIEnv*, jobject) = (void (*)(JNIEnv*, jobject))cpu_sym(PS("ZstdDictDecompress_free"));
    zjni_ddict* gpu = (zjni_ddict*)dict_get((*env)->GetLongField(env, obj, native_ptr_field(env, obj, &g_ddict_field)), 1);
    if (gpu) zjni_freeDDict(gpu);
    if (f) f(env, obj);
}
/* ZstdCompressCtx.loadDict(ZstdDictCompress) -> ZSTD_CCtx_refCDict (N/jni_fast_zstd.c:325-336) */
JNIEXPORT jlong JNICALL P(ZstdCompressCtx_loadCDictFast0)(JNIEnv* env, jclass cls, jlong ptr, jobject dict) {
    jlong (*f)(JNIEnv*, jclass, jlong, jobject) = (jlong (*)(JNIEnv*, jclass, jlong, jobject))cpu_sym(PS("ZstdCompressCtx_loadCDictFast0"));
    CtxState* s = st_get(ptr, 'C');
    if (s) { s->cdict = NULL; s->cpuDict = 0; st_drop_local_dict(s); }
    if (dict != NULL) {
        jlong const key = (*env)->GetLongField(env, dict, native_ptr_field(env, dict, &g_cdict_field));
        if (!key) return E_DICT;
        if (s) { s->cdict = (zjni_cdict*)dict_get(key, 0); s->cpuDict = (s->cdict == NULL); }   /* no digest (level > 3, table full, < 8 bytes): the bundled library has it */
    }
    return f ? f(env, cls, ptr, dict) : 0;
}
/* ZstdCompressCtx.loadDict(byte[]) -> ZSTD_CCtx_loadDictionary (N/jni_fast_zstd.c:343-357): the bytes are kept; the first compress call
 * digests them at the level set by then (CtxState.localCdict) */
JNIEXPORT jlong JNICALL P(ZstdCompressCtx_loadCDict0)(JNIEnv* env, jclass cls, jlong ptr, jbyteArray dict) {
    jlong (*f)(JNIEnv*, jclass, jlong, jbyteArray) = (jlong (*)(JNIEnv*, jclass, jlong, jbyteArray))cpu_sym(PS("ZstdCompressCtx_loadCDict0"));
    CtxState* s = st_get(ptr, 'C');
    if (s) {
        s->cdict = NULL; s->cpuDict = 0; st_drop_local_dict(s);
        if (dict != NULL) {
            jsize const n = (*env)->GetArrayLength(env, dict);
            s->rawDict = n > 0 ? malloc((size_t)n) : NULL;
            if (s->rawDict) { (*env)->GetByteArrayRegion(env, dict, 0, n, (jbyte*)s->rawDict); s->rawDictSize = (size_t)n; }
            else s->cpuDict = 1;                              /* (an empty dictionary is "no dictionary" to the reference: its own business) */
        }
    }
    if (f) return f(env, cls, ptr, dict);
    return 0;
}
JNIEXPORT jlong JNICALL P(ZstdDecompressCtx_loadDDictFast0)(JNIEnv* env, jclass cls, jlong ptr, jobject dict) {
    jlong (*f)(JNIEnv*, jclass, jlong, jobject) = (jlong (*)(JNIEnv*, jclass, jlong, jobject))cpu_sym(PS("ZstdDecompressCtx_loadDDictFast0"));
    CtxState* s = st_get(ptr, 'D');
    if (s) { if (s->ddictOwned) { zjni_freeDDict(s->ddictOwned); s->ddictOwned = NULL; } s->ddict = NULL; s->cpuDict = 0; }
    if (dict != NULL) {
        jlong const key = (*env)->GetLongField(env, dict, native_ptr_field(env, dict, &g_ddict_field));
        if (!key) return E_DICT;
        if (s) { s->ddict = (zjni_ddict*)dict_get(key, 0); s->cpuDict = (s->ddict == NULL); }
    }
    return f ? f(env, cls, ptr, dict) : 0;
}
/* ZstdDecompressCtx.loadDict(byte[]) -> ZSTD_DCtx_loadDictionary (N/jni_fast_zstd.c:691-705): same frames as with a ZSTD_DDict */
JNIEXPORT jlong JNICALL P(ZstdDecompressCtx_loadDDict0)(JNIEnv* env, jclass cls, jlong ptr, jbyteArray dict) {
    jlong (*f)(JNIEnv*, jclass, jlong, jbyteArray) = (jlong (*)(JNIEnv*, jclass, jlong, jbyteArray))cpu_sym(PS("ZstdDecompressCtx_loadDDict0"));
    CtxState* s = st_get(ptr, 'D');
    jlong r = 0;
    if (s) { if (s->ddictOwned) { zjni_freeDDict(s->ddictOwned); s->ddictOwned = NULL; } s->ddict = NULL; s->cpuDict = 0; }
    if (f) r = f(env, cls, ptr, dict);
    if (s && dict != NULL && r == 0) {
        jsize const n = (*env)->GetArrayLength(env, dict);
        if (n > 0 && gpu_on()) {                                        /* (an empty dictionary = none, ZSTD_DCtx_loadDictionary) */
            jbyte* copy = (jbyte*)malloc((size_t)n);
            if (copy) { (*env)->GetByteArrayRegion(env, dict, 0, n, copy); s->ddictOwned = zjni_createDDict(copy, (size_t)n); free(copy); }
            if (!s->ddictOwned) { if (f) s->cpuDict = 1; else r = -30; }   /* the reference answers a corrupted dictionary here (ZSTD_error_dictionary_corrupted) */
        } else if (n > 0) s->cpuDict = 1;
    }
    return r;
}

/* ---- compress: ZSTD_CCtx_reset + ZSTD_compress2 (N/jni_fast_zstd.c:606-607, :633-635) ----------------- */
static int gpu_takes_when(const CtxState* s, jint srcSize, int on) {
    if (!s || s->cpuOnly || s->cpuDict || !on) return 0;
    if (s->rawDict && !s->localCdict && !s->cpuDict) {                 /* first compress after loadDict(byte[]): the local CDict, at the level of this moment */
        CtxState* w = (CtxState*)s;
        int const lvl = s->level == 0 ? 3 : s->level;
        w->localCdict = (lvl >= 1 && lvl <= 3) ? zjni_createCDict(s->rawDict, s->rawDictSize, lvl) : NULL;
        if (!w->localCdict) w->cpuDict = 1;                            /* no digest (level > 3, under 8 bytes, refused): the bundled library has it */
        if (w->cpuDict) return 0;
    }
    if (s->cdict || s->localCdict) return s->contentSize;              /* sizes the library does not take come back as 40 / 201 and are forwarded */
    if (s->level >= 4 && s->level <= 8) return (size_t)srcSize <= (s->level == 4 ? ZJNI_LEVEL4_MAX : ZJNI_LAZY_MAX) && !(s->hashLog | s->chainLog);   /* one block, no explicit table sizes */
    return s->level >= 0 && s->level <= 3 && (size_t)srcSize <= ZJNI_FRAME_MAX;       /* beyond the level's window the library answers 201 and the call is forwarded */
}
static int gpu_takes(const CtxState* s, jint srcSize) {           /* the one-shot natives: ZSTD_CCtx_reset(session_only) comes first, and forgets a pledged size */
    if (s) ((CtxState*)s)->hasPledged = 0;
    return gpu_takes_when(s, srcSize, per_buffer_on_gpu());
}
static int frame_flags(const CtxState* s) {
    return (s->checksum ? ZJNI_FRAME_CHECKSUM : 0) | (s->contentSize ? 0 : ZJNI_FRAME_NO_CONTENTSIZE) | (s->dictIDFlag ? 0 : ZJNI_FRAME_NO_DICTID);
}
static size_t gpu_compress(const CtxState* s, void* dst, size_t dstCap, const void* src, size_t srcSize) {
    size_t res = 0; const void* sp = src; void* dp = dst; size_t r;
    if (s->cdict || s->localCdict) r = zjni_compress_batch_usingCDict(&sp, &srcSize, &dp, &dstCap, &res, 1, s->cdict ? s->cdict : s->localCdict, frame_flags(s));
    else if (aggregator() && !(s->hashLog | s->chainLog) && (frame_flags(s) & ~ZJNI_FRAME_CHECKSUM) == 0)
        return zjni_aggregator_compress(aggregator(), dst, dstCap, src, srcSize, s->level, s->checksum ? 1 : 0);
    else r = zjni_compress_batch_advanced(&sp, &srcSize, &dp, &dstCap, &res, 1, s->level, frame_flags(s), s->hashLog, s->chainLog);   /* explicit table sizes: level 3 only (40 otherwise -> forwarded) */
    return zjni_isError(r) ? r : res;
}
static int gpu_compress_final(const CtxState* s, size_t r, int haveCpu) {   /* 40 / 42 = outside what the GPU path takes (attach range, table sizes): forward when possible */
    return stat_verdict(result_final(r) && !(zjni_isError(r) && (zjni_getErrorCode(r) == 40 || zjni_getErrorCode(r) == 42) && (s->cdict || s->localCdict || s->hashLog || s->chainLog) && haveCpu), r);
}
typedef jlong (*cbuf_fn)(JNIEnv*, jclass, jlong, jobject, jint, jint, jobject, jint, jint);
static jlong buf_forward(const char* name, jlong none, JNIEnv* env, jclass cls, jlong ptr, jobject dst, jint doff, jint dsize, jobject src, jint soff, jint ssize) {
    cbuf_fn f = (cbuf_fn)cpu_sym(name);
    if (f) stat_forward(); else t_gpu_declined = 0;
    return f ? f(env, cls, ptr, dst, doff, dsize, src, soff, ssize) : none;
}

JNIEXPORT jlong JNICALL P(ZstdCompressCtx_compressDirectByteBuffer0)
  (JNIEnv* env, jclass cls, jlong ptr, jobject dst, jint dst_offset, jint dst_size, jobject src, jint src_offset, jint src_size) {
    CtxState* s = st_get(ptr, 'C');
    if (NULL == dst) return E_DST;
    if (NULL == src) return E_SRC;
    if (0 > dst_offset) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_size) return E_SRC;
    if (dst_offset + dst_size > (*env)->GetDirectBufferCapacity(env, dst)) return E_DST;
    if (src_offset + src_size > (*env)->GetDirectBufferCapacity(env, src)) return E_SRC;
    {   char* d = (char*)(*env)->GetDirectBufferAddress(env, dst);
        char* sb = (char*)(*env)->GetDirectBufferAddress(env, src);
        if (d == NULL || sb == NULL) return E_MEM;
        if (gpu_takes(s, src_size)) {
            size_t const r = gpu_compress(s, d + dst_offset, (size_t)dst_size, sb + src_offset, (size_t)src_size);
            if (gpu_compress_final(s, r, cpu_sym(PS("ZstdCompressCtx_compressDirectByteBuffer0")) != NULL)) return (jlong)r;
        }
    }
    return buf_forward(PS("ZstdCompressCtx_compressDirectByteBuffer0"), -(jlong)ZJNI_ERROR_unsupported, env, cls, ptr, dst, dst_offset, dst_size, src, src_offset, src_size);
}
JNIEXPORT jlong JNICALL P(ZstdCompressCtx_compressByteArray0)
  (JNIEnv* env, jclass cls, jlong ptr, jbyteArray dst, jint dst_offset, jint dst_size, jbyteArray src, jint src_offset, jint src_size) {
    CtxState* s = st_get(ptr, 'C');
    if (0 > dst_offset) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_size) return E_SRC;
    if (src_offset + src_size > (*env)->GetArrayLength(env, src)) return E_SRC;
    if (dst_offset + dst_size > (*env)->GetArrayLength(env, dst)) return E_DST;
    if (gpu_takes(s, src_size)) {
        jbyte* sb = (jbyte*)malloc((size_t)src_size + 1); jbyte* d = (jbyte*)malloc((size_t)dst_size + 1);
        size_t r = (size_t)E_MEM;
        if (sb && d) {
            (*env)->GetByteArrayRegion(env, src, src_offset, src_size, sb);
            r = gpu_compress(s, d, (size_t)dst_size, sb, (size_t)src_size);
            if (!zjni_isError(r)) (*env)->SetByteArrayRegion(env, dst, dst_offset, (jsize)r, d);
        }
        free(sb); free(d);
        if (gpu_compress_final(s, r, cpu_sym(PS("ZstdCompressCtx_compressByteArray0")) != NULL)) return (jlong)r;
    }
    return buf_forward(PS("ZstdCompressCtx_compressByteArray0"), -(jlong)ZJNI_ERROR_unsupported, env, cls, ptr, dst, dst_offset, dst_size, src, src_offset, src_size);
}

/* ---- decompress: ZSTD_DCtx_reset + ZSTD_decompressDCtx (N/jni_fast_zstd.c:798-799, :825-826, :858-860, :892-894) */
static int gpu_dec_takes(const CtxState* s) { return s && !s->cpuOnly && !s->cpuDict && per_buffer_on_gpu(); }
static size_t gpu_decompress(const CtxState* s, void* dst, size_t dstCap, const void* src, size_t srcSize) {
    zjni_ddict* const dd = s->ddictOwned ? s->ddictOwned : s->ddict;
    if (!dd && aggregator()) return zjni_aggregator_decompress(aggregator(), dst, dstCap, src, srcSize);
    return dd ? zjni_decompress_usingDDict(dst, dstCap, src, srcSize, dd) : zjni_decompress(dst, dstCap, src, srcSize);
}
JNIEXPORT jlong JNICALL P(ZstdDecompressCtx_decompressDirectByteBuffer0)
  (JNIEnv* env, jclass cls, jlong ptr, jobject dst, jint dst_offset, jint dst_size, jobject src, jint src_offset, jint src_size) {
    CtxState* s = st_get(ptr, 'D');
    if (NULL == dst) return E_DST;
    if (NULL == src) return E_SRC;
    if (0 > dst_offset) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_size) return E_SRC;
    if (dst_offset + dst_size > (*env)->GetDirectBufferCapacity(env, dst)) return E_DST;
    if (src_offset + src_size > (*env)->GetDirectBufferCapacity(env, src)) return E_SRC;
    {   char* d = (char*)(*env)->GetDirectBufferAddress(env, dst);
        char* sb = (char*)(*env)->GetDirectBufferAddress(env, src);
        if (d == NULL || sb == NULL) return E_MEM;
        if (gpu_dec_takes(s)) {
            size_t const r = gpu_decompress(s, d + dst_offset, (size_t)dst_size, sb + src_offset, (size_t)src_size);
            if (gpu_result_final(r)) return (jlong)r;
        }
    }
    return buf_forward(PS("ZstdDecompressCtx_decompressDirectByteBuffer0"), -(jlong)ZJNI_ERROR_no_device, env, cls, ptr, dst, dst_offset, dst_size, src, src_offset, src_size);
}
JNIEXPORT jlong JNICALL P(ZstdDecompressCtx_decompressByteArray0)
  (JNIEnv* env, jclass cls, jlong ptr, jbyteArray dst, jint dst_offset, jint dst_size, jbyteArray src, jint src_offset, jint src_size) {
    CtxState* s = st_get(ptr, 'D');
    if (0 > dst_offset) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_size) return E_SRC;
    if (src_offset + src_size > (*env)->GetArrayLength(env, src)) return E_SRC;
    if (dst_offset + dst_size > (*env)->GetArrayLength(env, dst)) return E_DST;
    if (gpu_dec_takes(s)) {
        jbyte* sb = (jbyte*)malloc((size_t)src_size + 1); jbyte* d = (jbyte*)malloc((size_t)dst_size + 1);
        size_t r = (size_t)E_MEM;
        if (sb && d) {
            (*env)->GetByteArrayRegion(env, src, src_offset, src_size, sb);
            r = gpu_decompress(s, d, (size_t)dst_size, sb, (size_t)src_size);
            if (!zjni_isError(r)) (*env)->SetByteArrayRegion(env, dst, dst_offset, (jsize)r, d);
        }
        free(sb); free(d);
        if (gpu_result_final(r)) return (jlong)r;
    }
    return buf_forward(PS("ZstdDecompressCtx_decompressByteArray0"), -(jlong)ZJNI_ERROR_no_device, env, cls, ptr, dst, dst_offset, dst_size, src, src_offset, src_size);
}
/* byte[] source into a direct destination (N/jni_fast_zstd.c:838-866) */
JNIEXPORT jlong JNICALL P(ZstdDecompressCtx_decompressByteArrayToDirectByteBuffer0)
  (JNIEnv* env, jclass cls, jlong ptr, jobject dst, jint dst_offset, jint dst_size, jbyteArray src, jint src_offset, jint src_size) {
    CtxState* s = st_get(ptr, 'D');
    if (NULL == dst) return E_DST;
    if (NULL == src) return E_SRC;
    if (0 > dst_offset) return E_DST;
    if (0 > dst_size) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_size) return E_SRC;
    if (src_offset > (*env)->GetArrayLength(env, src) - src_size) return E_SRC;
    if (dst_offset > (jint)(*env)->GetDirectBufferCapacity(env, dst) - dst_size) return E_DST;
    {   char* d = (char*)(*env)->GetDirectBufferAddress(env, dst);
        if (d == NULL) return E_MEM;
        if (gpu_dec_takes(s)) {
            jbyte* sb = (jbyte*)malloc((size_t)src_size + 1);
            size_t r = (size_t)E_MEM;
            if (sb) { (*env)->GetByteArrayRegion(env, src, src_offset, src_size, sb); r = gpu_decompress(s, d + dst_offset, (size_t)dst_size, sb, (size_t)src_size); }
            free(sb);
            if (gpu_result_final(r)) return (jlong)r;
        }
    }
    return buf_forward(PS("ZstdDecompressCtx_decompressByteArrayToDirectByteBuffer0"), -(jlong)ZJNI_ERROR_no_device, env, cls, ptr, dst, dst_offset, dst_size, src, src_offset, src_size);
}
/* direct source into a byte[] destination (N/jni_fast_zstd.c:872-900) */
JNIEXPORT jlong JNICALL P(ZstdDecompressCtx_decompressDirectByteBufferToByteArray0)
  (JNIEnv* env, jclass cls, jlong ptr, jbyteArray dst, jint dst_offset, jint dst_size, jobject src, jint src_offset, jint src_size) {
    CtxState* s = st_get(ptr, 'D');
    if (NULL == dst) return E_DST;
    if (NULL == src) return E_SRC;
    if (0 > dst_offset) return E_DST;
    if (0 > dst_size) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_size) return E_SRC;
    if (dst_offset > (*env)->GetArrayLength(env, dst) - dst_size) return E_DST;
    if (src_offset > (jint)(*env)->GetDirectBufferCapacity(env, src) - src_size) return E_SRC;
    {   char* sb = (char*)(*env)->GetDirectBufferAddress(env, src);
        if (sb == NULL) return E_MEM;
        if (gpu_dec_takes(s)) {
            jbyte* d = (jbyte*)malloc((size_t)dst_size + 1);
            size_t r = (size_t)E_MEM;
            if (d) { r = gpu_decompress(s, d, (size_t)dst_size, sb + src_offset, (size_t)src_size); if (!zjni_isError(r)) (*env)->SetByteArrayRegion(env, dst, dst_offset, (jsize)r, d); }
            free(d);
            if (gpu_result_final(r)) return (jlong)r;
        }
    }
    return buf_forward(PS("ZstdDecompressCtx_decompressDirectByteBufferToByteArray0"), -(jlong)ZJNI_ERROR_no_device, env, cls, ptr, dst, dst_offset, dst_size, src, src_offset, src_size);
}

/* ---- class Zstd: helpers with the reference's semantics (N/jni_zstd.c:230-267, :50-63, :86-96) -------- */
JNIEXPORT jlong JNICALL P(Zstd_compressBound)(JNIEnv* env, jclass cls, jlong size) { (void)env; (void)cls; return (jlong)zjni_compressBound((size_t)size); }
JNIEXPORT jboolean JNICALL P(Zstd_isError)(JNIEnv* env, jclass cls, jlong code) { (void)env; (void)cls; return zjni_isError((size_t)code) != 0; }
JNIEXPORT jstring JNICALL P(Zstd_getErrorName)(JNIEnv* env, jclass cls, jlong code) { (void)cls; return (*env)->NewStringUTF(env, zjni_getErrorName((size_t)code)); }
JNIEXPORT jlong JNICALL P(Zstd_getErrorCode)(JNIEnv* env, jclass cls, jlong code) { (void)env; (void)cls; return (jlong)zjni_getErrorCode((size_t)code); }
/* Zstd.getFrameContentSize(byte[], offset, limit, magicless) -> ZSTD_getFrameContentSize (N/jni_zstd.c:30-43, :86-96) */
static size_t frame_content_size(const uint8_t* p, size_t n, int magicless);      /* "frame inspection", at the end of this file */
static int frame_is_legacy(const uint8_t* p, size_t n);
JNIEXPORT jlong JNICALL P(Zstd_getFrameContentSize0)(JNIEnv* env, jclass cls, jbyteArray src, jint offset, jint limit, jboolean magicless) {
    jbyte head[18]; jint const n = limit < 18 ? (limit < 0 ? 0 : limit) : 18;              /* a frame header is at most 18 bytes */
    (*env)->GetByteArrayRegion(env, src, offset, n, head);
    if (frame_is_legacy((const uint8_t*)head, (size_t)n)) {
        jlong (*f)(JNIEnv*, jclass, jbyteArray, jint, jint, jboolean) = (jlong (*)(JNIEnv*, jclass, jbyteArray, jint, jint, jboolean))cpu_sym(PS("Zstd_getFrameContentSize0"));
        if (f) return f(env, cls, src, offset, limit, magicless);
    }
    return (jlong)frame_content_size((const uint8_t*)head, (size_t)n, magicless == JNI_TRUE);
}
JNIEXPORT jlong JNICALL P(Zstd_compressUnsafe)
  (JNIEnv* env, jclass cls, jlong dst, jlong dst_size, jlong src, jlong src_size, jint level, jboolean checksumFlag) {
    if (per_buffer_on_gpu() && level >= 0 && (level <= 3 ? (size_t)src_size <= ZJNI_FRAME_MAX : (level <= 8 && (size_t)src_size <= (level == 4 ? ZJNI_LEVEL4_MAX : ZJNI_LAZY_MAX)))) {
        size_t const r = zjni_compress2((void*)(intptr_t)dst, (size_t)dst_size, (const void*)(intptr_t)src, (size_t)src_size, level, checksumFlag == JNI_TRUE);
        if (gpu_result_final(r)) return (jlong)r;
    }
    {   jlong (*f)(JNIEnv*, jclass, jlong, jlong, jlong, jlong, jint, jboolean) =
            (jlong (*)(JNIEnv*, jclass, jlong, jlong, jlong, jlong, jint, jboolean))cpu_sym(PS("Zstd_compressUnsafe"));
        if (f) { stat_forward(); return f(env, cls, dst, dst_size, src, src_size, level, checksumFlag); }
    }
    t_gpu_declined = 0;
    return -(jlong)ZJNI_ERROR_unsupported;
}
JNIEXPORT jlong JNICALL P(Zstd_decompressUnsafe)(JNIEnv* env, jclass cls, jlong dst, jlong dst_size, jlong src, jlong src_size) {
    if (per_buffer_on_gpu()) {
        size_t const r = zjni_decompress((void*)(intptr_t)dst, (size_t)dst_size, (const void*)(intptr_t)src, (size_t)src_size);
        if (gpu_result_final(r)) return (jlong)r;
    }
    {   jlong (*f)(JNIEnv*, jclass, jlong, jlong, jlong, jlong) = (jlong (*)(JNIEnv*, jclass, jlong, jlong, jlong, jlong))cpu_sym(PS("Zstd_decompressUnsafe"));
        if (f) { stat_forward(); return f(env, cls, dst, dst_size, src, src_size); }
    }
    t_gpu_declined = 0;
    return -(jlong)ZJNI_ERROR_no_device;
}

/* ---- class Zstd: the one-shot natives over a ZstdDictCompress / ZstdDictDecompress object (N/jni_fast_zstd.c:133-244) --------
 * = ZSTD_compress_usingCDict / ZSTD_decompress_usingDDict on a fresh context: default frame layout, the dictionary's level.  What
 * Zstd.compress(dst, src, ZstdDictCompress) / Zstd.decompress(dst, src, ZstdDictDecompress) and their ByteBuffer overloads call.
 * Argument checks in the reference's order; a dictionary without a GPU digest, or a source the dictionary pipeline answers 40 / 201
 * for, goes to the bundled library's native of the same name. */
static size_t fastdict_compress(zjni_cdict* cd, void* dst, size_t dstCap, const void* src, size_t srcSize) {
    size_t res = 0; const void* sp = src; void* dp = dst;
    size_t const r = zjni_compress_batch_usingCDict(&sp, &srcSize, &dp, &dstCap, &res, 1, cd, 0);
    return zjni_isError(r) ? r : res;
}
static int fastdict_final(size_t r, int haveCpu) {
    return stat_verdict(result_final(r) && !(zjni_isError(r) && (zjni_getErrorCode(r) == 40 || zjni_getErrorCode(r) == 42) && haveCpu), r);
}
typedef jlong (*fd_arr_fn)(JNIEnv*, jclass, jbyteArray, jint, jbyteArray, jint, jint, jobject);
typedef jlong (*fd_buf_fn)(JNIEnv*, jclass, jobject, jint, jint, jobject, jint, jint, jobject);
static jlong fastdict_array(JNIEnv* env, jclass cls, jbyteArray dst, jint dst_offset, jbyteArray src, jint src_offset, jint src_length, jobject dict, int compress, const char* name) {
    fd_arr_fn f = (fd_arr_fn)cpu_sym(name);
    jlong key;
    if (NULL == dict) return E_DICT;
    key = (*env)->GetLongField(env, dict, native_ptr_field(env, dict, compress ? &g_cdict_field : &g_ddict_field));
    if (!key) return E_DICT;
    if (NULL == dst) return E_DST;
    if (NULL == src) return E_SRC;
    if (0 > dst_offset) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_length) return E_SRC;
    {   jsize dst_size = (*env)->GetArrayLength(env, dst); jsize const src_size = (*env)->GetArrayLength(env, src);
        void* const gpu = per_buffer_on_gpu() ? dict_get(key, 0) : NULL;
        if (dst_offset > dst_size) return E_DST;
        if (src_size < src_offset + src_length) return E_SRC;
        dst_size -= dst_offset;
        if (gpu) {
            jbyte* sb = (jbyte*)malloc((size_t)src_length + 1); jbyte* d = (jbyte*)malloc((size_t)dst_size + 1);
            size_t r = (size_t)E_MEM;
            if (sb && d) {
                (*env)->GetByteArrayRegion(env, src, src_offset, src_length, sb);
                r = compress ? fastdict_compress((zjni_cdict*)gpu, d, (size_t)dst_size, sb, (size_t)src_length)
                             : zjni_decompress_usingDDict(d, (size_t)dst_size, sb, (size_t)src_length, (zjni_ddict*)gpu);
                if (!zjni_isError(r)) (*env)->SetByteArrayRegion(env, dst, dst_offset, (jsize)r, d);
            }
            free(sb); free(d);
            if (compress ? fastdict_final(r, f != NULL) : gpu_result_final(r)) return (jlong)r;
        }
    }
    if (f) stat_forward(); else t_gpu_declined = 0;
    return f ? f(env, cls, dst, dst_offset, src, src_offset, src_length, dict) : -(jlong)ZJNI_ERROR_unsupported;
}
static jlong fastdict_direct(JNIEnv* env, jclass cls, jobject dst, jint dst_offset, jint dst_size, jobject src, jint src_offset, jint src_size, jobject dict, int compress, const char* name) {
    fd_buf_fn f = (fd_buf_fn)cpu_sym(name);
    jlong key;
    if (NULL == dict) return E_DICT;
    key = (*env)->GetLongField(env, dict, native_ptr_field(env, dict, compress ? &g_cdict_field : &g_ddict_field));
    if (!key) return E_DICT;
    if (NULL == dst) return E_DST;
    if (NULL == src) return E_SRC;
    if (0 > dst_offset) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_size) return E_SRC;
    {   char* const d = (char*)(*env)->GetDirectBufferAddress(env, dst); char* const sb = (char*)(*env)->GetDirectBufferAddress(env, src);
        void* const gpu = (d && sb && dst_size >= 0 && per_buffer_on_gpu()) ? dict_get(key, 0) : NULL;   /* (the reference does not test the addresses: its own business) */
        if (gpu) {
            size_t const r = compress ? fastdict_compress((zjni_cdict*)gpu, d + dst_offset, (size_t)dst_size, sb + src_offset, (size_t)src_size)
                                      : zjni_decompress_usingDDict(d + dst_offset, (size_t)dst_size, sb + src_offset, (size_t)src_size, (zjni_ddict*)gpu);
            if (compress ? fastdict_final(r, f != NULL) : gpu_result_final(r)) return (jlong)r;
        }
    }
    if (f) stat_forward(); else t_gpu_declined = 0;
    return f ? f(env, cls, dst, dst_offset, dst_size, src, src_offset, src_size, dict) : -(jlong)ZJNI_ERROR_unsupported;
}
JNIEXPORT jlong JNICALL P(Zstd_compressFastDict0)(JNIEnv* env, jclass cls, jbyteArray dst, jint dst_offset, jbyteArray src, jint src_offset, jint src_length, jobject dict) {
    return fastdict_array(env, cls, dst, dst_offset, src, src_offset, src_length, dict, 1, PS("Zstd_compressFastDict0"));
}
JNIEXPORT jlong JNICALL P(Zstd_decompressFastDict0)(JNIEnv* env, jclass cls, jbyteArray dst, jint dst_offset, jbyteArray src, jint src_offset, jint src_length, jobject dict) {
    return fastdict_array(env, cls, dst, dst_offset, src, src_offset, src_length, dict, 0, PS("Zstd_decompressFastDict0"));
}
JNIEXPORT jlong JNICALL P(Zstd_compressDirectByteBufferFastDict0)(JNIEnv* env, jclass cls, jobject dst, jint dst_offset, jint dst_size, jobject src, jint src_offset, jint src_size, jobject dict) {
    return fastdict_direct(env, cls, dst, dst_offset, dst_size, src, src_offset, src_size, dict, 1, PS("Zstd_compressDirectByteBufferFastDict0"));
}
JNIEXPORT jlong JNICALL P(Zstd_decompressDirectByteBufferFastDict0)(JNIEnv* env, jclass cls, jobject dst, jint dst_offset, jint dst_size, jobject src, jint src_offset, jint src_size, jobject dict) {
    return fastdict_direct(env, cls, dst, dst_offset, dst_size, src, src_offset, src_size, dict, 0, PS("Zstd_decompressDirectByteBufferFastDict0"));
}

/* ---- new, additive: batch natives over arrays of direct ByteBuffers ----------------
 * static native long compressBatch0(ByteBuffer[] srcs, ByteBuffer[] dsts, long[] results, int level, boolean checksum);
 * static native long decompressBatch0(ByteBuffer[] srcs, ByteBuffer[] dsts, long[] results);
 * static native long compressBatchDict0(ByteBuffer[] srcs, ByteBuffer[] dsts, long[] results, ZstdDictCompress dict, boolean checksum);
 * Each buffer is taken from position 0 to its capacity and must be a direct buffer.  results[i] = size or the error code
 * compress*0 / decompress*0 would have returned for that buffer; the return value is 0 or a call-level error
 * (srcSize_wrong / dstSize_tooSmall for a null or non-direct element, as the per-buffer natives answer). */
static jlong batch(JNIEnv* env, jobjectArray srcs, jobjectArray dsts, jlongArray results, int compress, int level, int checksum, const zjni_cdict* cdict) {
    jsize n; const void** sp; void** dp; size_t* ss; size_t* dc; size_t* res; jlong* out; size_t r; jsize i; jlong bad = 0;
    if (srcs == NULL) return E_SRC;
    if (dsts == NULL || results == NULL) return E_DST;
    n = (*env)->GetArrayLength(env, srcs);
    if ((*env)->GetArrayLength(env, dsts) != n || (*env)->GetArrayLength(env, results) < n) return E_SRC;
    if (n == 0) return 0;
    sp = (const void**)malloc(n * sizeof(*sp)); dp = (void**)malloc(n * sizeof(*dp));
    ss = (size_t*)malloc(n * sizeof(*ss)); dc = (size_t*)malloc(n * sizeof(*dc)); res = (size_t*)malloc(n * sizeof(*res));
    out = (jlong*)malloc(n * sizeof(*out));
    if (!sp || !dp || !ss || !dc || !res || !out) { free(sp); free(dp); free(ss); free(dc); free(res); free(out); return E_MEM; }
    for (i = 0; i < n && !bad; i++) {
        jobject s = (*env)->GetObjectArrayElement(env, srcs, i), d = (*env)->GetObjectArrayElement(env, dsts, i);
        jlong const sl = s ? (*env)->GetDirectBufferCapacity(env, s) : -1, dl = d ? (*env)->GetDirectBufferCapacity(env, d) : -1;
        sp[i] = s ? (*env)->GetDirectBufferAddress(env, s) : NULL; dp[i] = d ? (*env)->GetDirectBufferAddress(env, d) : NULL;
        if (s == NULL || sl < 0 || (sp[i] == NULL && sl > 0)) bad = E_SRC;            /* null, or not a direct buffer */
        else if (d == NULL || dl < 0 || (dp[i] == NULL && dl > 0)) bad = E_DST;
        ss[i] = (size_t)(sl < 0 ? 0 : sl); dc[i] = (size_t)(dl < 0 ? 0 : dl);
        if (s && (*env)->DeleteLocalRef) (*env)->DeleteLocalRef(env, s);              /* 2n local references would overflow the frame's table on large batches */
        if (d && (*env)->DeleteLocalRef) (*env)->DeleteLocalRef(env, d);
    }
    if (bad) r = (size_t)bad;
    else r = cdict ? zjni_compress_batch_usingCDict(sp, ss, dp, dc, res, (size_t)n, cdict, checksum)
           : compress ? zjni_compress_batch2(sp, ss, dp, dc, res, (size_t)n, level, checksum) : zjni_decompress_batch(sp, ss, dp, dc, res, (size_t)n);
    if (!zjni_isError(r)) { for (i = 0; i < n; i++) out[i] = (jlong)res[i]; (*env)->SetLongArrayRegion(env, results, 0, n, out); }
    free(sp); free(dp); free(ss); free(dc); free(res); free(out);
    return (jlong)r;
}
JNIEXPORT jlong JNICALL P(Zstd_compressBatch0)(JNIEnv* env, jclass cls, jobjectArray srcs, jobjectArray dsts, jlongArray results, jint level, jboolean checksum) {
    (void)cls;
    if (!gpu_on()) return -(jlong)ZJNI_ERROR_no_device;
    return batch(env, srcs, dsts, results, 1, level, checksum == JNI_TRUE, NULL);
}
JNIEXPORT jlong JNICALL P(Zstd_decompressBatch0)(JNIEnv* env, jclass cls, jobjectArray srcs, jobjectArray dsts, jlongArray results) {
    (void)cls;
    if (!gpu_on()) return -(jlong)ZJNI_ERROR_no_device;
    return batch(env, srcs, dsts, results, 0, 0, 0, NULL);
}
JNIEXPORT jlong JNICALL P(Zstd_compressBatchDict0)(JNIEnv* env, jclass cls, jobjectArray srcs, jobjectArray dsts, jlongArray results, jobject dict, jboolean checksum) {
    zjni_cdict* g;
    (void)cls;
    if (!gpu_on()) return -(jlong)ZJNI_ERROR_no_device;
    if (dict == NULL) return E_DICT;
    g = (zjni_cdict*)dict_get((*env)->GetLongField(env, dict, native_ptr_field(env, dict, &g_cdict_field)), 0);
    if (!g) return E_DICT;
    return batch(env, srcs, dsts, results, 1, 0, checksum == JNI_TRUE, g);
}

/* ---- the same, asynchronous (round 6): two Java-side batches in flight from ONE thread ------------------------
 * static native long compressBatchBegin0(ByteBuffer[] srcs, ByteBuffer[] dsts, int level, boolean checksum);   // -> job handle; <= 0: the error the blocking native would return (0 - code), nothing begun
 * static native long decompressBatchBegin0(ByteBuffer[] srcs, ByteBuffer[] dsts);
 * static native long batchFinish0(long job, long[] results);                                                    // waits, writes results[0 .. n), frees the job; returns what the blocking native returns
 * zstd-jni's natives block (N/jni_fast_zstd.c:586-640: one ZSTD_compress2 per call) and so do the batch natives above; a single host batch is a chain — gather + H2D,
 * kernels, D2H + scatter — and the overlap comes from the NEXT batch (zjni_compress_batch_begin / zjni_batch_finish, include/zjni_amd.h: two staging slots per
 * device).  Begin takes the buffers' addresses and capacities as the blocking natives do and holds GLOBAL references on the two arrays until Finish: the arrays and
 * their direct buffers belong to the job meanwhile (the array keeps its elements reachable; overwriting an element before Finish is the caller's error, as handing a
 * ZstdDirectBufferCompressingStream's target to someone else is, N/jni_directbuffercompress_zstd.c:97-161).  A job that is never finished leaks its references. */
typedef struct BatchJob { zjni_batch_job* job; jobject srcsRef, dstsRef; const void** sp; void** dp; size_t* ss; size_t* dc; size_t* res; jsize n; } BatchJob;
static void batch_job_free(JNIEnv* env, BatchJob* j) {
    if (!j) return;
    if (j->srcsRef && (*env)->DeleteGlobalRef) (*env)->DeleteGlobalRef(env, j->srcsRef);
    if (j->dstsRef && (*env)->DeleteGlobalRef) (*env)->DeleteGlobalRef(env, j->dstsRef);
    free(j->sp); free(j->dp); free(j->ss); free(j->dc); free(j->res); free(j);
}
static jlong batch_begin(JNIEnv* env, jobjectArray srcs, jobjectArray dsts, int compress, int level, int checksum) {
    jsize n, i; jlong bad = 0; BatchJob* j;
    if (srcs == NULL) return E_SRC;
    if (dsts == NULL) return E_DST;
    n = (*env)->GetArrayLength(env, srcs);
    if ((*env)->GetArrayLength(env, dsts) != n) return E_SRC;
    j = (BatchJob*)calloc(1, sizeof(*j));
    if (!j) return E_MEM;
    j->n = n;
    j->sp = (const void**)malloc((n + 1) * sizeof(*j->sp)); j->dp = (void**)malloc((n + 1) * sizeof(*j->dp));
    j->ss = (size_t*)malloc((n + 1) * sizeof(*j->ss)); j->dc = (size_t*)malloc((n + 1) * sizeof(*j->dc)); j->res = (size_t*)calloc((size_t)n + 1, sizeof(*j->res));
    if (!j->sp || !j->dp || !j->ss || !j->dc || !j->res) { batch_job_free(env, j); return E_MEM; }
    for (i = 0; i < n && !bad; i++) {
        jobject s = (*env)->GetObjectArrayElement(env, srcs, i), d = (*env)->GetObjectArrayElement(env, dsts, i);
        jlong const sl = s ? (*env)->GetDirectBufferCapacity(env, s) : -1, dl = d ? (*env)->GetDirectBufferCapacity(env, d) : -1;
        j->sp[i] = s ? (*env)->GetDirectBufferAddress(env, s) : NULL; j->dp[i] = d ? (*env)->GetDirectBufferAddress(env, d) : NULL;
        if (s == NULL || sl < 0 || (j->sp[i] == NULL && sl > 0)) bad = E_SRC;
        else if (d == NULL || dl < 0 || (j->dp[i] == NULL && dl > 0)) bad = E_DST;
        j->ss[i] = (size_t)(sl < 0 ? 0 : sl); j->dc[i] = (size_t)(dl < 0 ? 0 : dl);
        if (s && (*env)->DeleteLocalRef) (*env)->DeleteLocalRef(env, s);
        if (d && (*env)->DeleteLocalRef) (*env)->DeleteLocalRef(env, d);
    }
    if (bad) { batch_job_free(env, j); return bad; }
    if ((*env)->NewGlobalRef) { j->srcsRef = (*env)->NewGlobalRef(env, srcs); j->dstsRef = (*env)->NewGlobalRef(env, dsts); }
    if (n == 0) return (jlong)(intptr_t)j;                               /* an empty batch: a job that finishes at once */
    j->job = compress ? zjni_compress_batch_begin(j->sp, j->ss, j->dp, j->dc, j->res, (size_t)n, level, checksum)
                      : zjni_decompress_batch_begin(j->sp, j->ss, j->dp, j->dc, j->res, (size_t)n);
    if (!j->job) { batch_job_free(env, j); return -(jlong)ZJNI_ERROR_no_device; }
    return (jlong)(intptr_t)j;
}
JNIEXPORT jlong JNICALL P(Zstd_compressBatchBegin0)(JNIEnv* env, jclass cls, jobjectArray srcs, jobjectArray dsts, jint level, jboolean checksum) {
    (void)cls;
    if (!gpu_on()) return -(jlong)ZJNI_ERROR_no_device;
    return batch_begin(env, srcs, dsts, 1, level, checksum == JNI_TRUE);
}
JNIEXPORT jlong JNICALL P(Zstd_decompressBatchBegin0)(JNIEnv* env, jclass cls, jobjectArray srcs, jobjectArray dsts) {
    (void)cls;
    if (!gpu_on()) return -(jlong)ZJNI_ERROR_no_device;
    return batch_begin(env, srcs, dsts, 0, 0, 0);
}
JNIEXPORT jlong JNICALL P(Zstd_batchFinish0)(JNIEnv* env, jclass cls, jlong job, jlongArray results) {
    BatchJob* const j = (BatchJob*)(intptr_t)job; size_t r = 0; jlong ret;
    (void)cls;
    if (job <= 0) return E_SRC;                                           /* not a job: Begin's error codes are <= 0 */
    if (j->job) r = zjni_batch_finish(j->job);                            /* always waited for and freed, whatever `results` is */
    ret = (jlong)r;
    if (!zjni_isError(r)) {
        if (results == NULL || (*env)->GetArrayLength(env, results) < j->n) ret = E_DST;
        else if (j->n) {
            jlong* const out = (jlong*)malloc((size_t)j->n * sizeof(*out)); jsize i;
            if (!out) ret = E_MEM;
            else { for (i = 0; i < j->n; i++) out[i] = (jlong)j->res[i]; (*env)->SetLongArrayRegion(env, results, 0, j->n, out); free(out); }
        }
    }
    batch_job_free(env, j);
    return ret;
}

/* ==== ZstdDirectBufferCompressingStreamNoFinalizer (N/jni_directbuffercompress_zstd.c) on the GPU ==========================================
 * The reference feeds ZSTD_compressStream / ZSTD_flushStream / ZSTD_endStream.  The GPU route BUFFERS the stream until it is closed (at most the level's
 * unknown-size window: 512 KiB / 1 MiB / 2 MiB at levels 1 / 2 / 3) and makes the frame ZSTD_compressStream2 would have made, byte for byte, in one
 * zjni_compress_stream call: compressDirectByteBuffer only takes the bytes in, flushStream returns the frame's next part up to here (the call is
 * repeated over everything written so far and only the new bytes are handed out: every decision depends only on the bytes before it), endStream
 * the rest and the epilogue.  Output the caller's buffer has no room for waits in the stream state and the native says how much, as the reference does.
 * A stream that outgrows the window, carries a dictionary, uses a level above 3 or meets a GPU that declines is REPLAYED into the bundled library's
 * stream (its natives are called with direct buffers made for the purpose) and stays there.  Like the per-buffer natives a single stream cannot amortise
 * a launch, so with the bundled library present streams go there unless ZSTD_JNI_GPU_STREAMS=1 (or ZSTD_JNI_GPU_PER_BUFFER=1); ZSTD_JNI_GPU_STREAMS=0
 * keeps them on the CPU in every configuration that has one. */
typedef struct StreamState {
    struct StreamState* next; jlong key;
    int level, checksum, cpuMode, started, finished;
    int levelNext, hasLevelNext;                                /* a level set inside a frame made here: the next frame's (ss_note_parameter) */
    int fwdLevel, fwdLevelPending;                              /* ... and not yet passed on to the bundled context: it would recompress a REPLAYED frame at the new level — passed on when the frame is out, or right behind a replay (ss_forward_level) */
    int levelCpu, paramCpu;                                     /* sticky across sessions (ZstdOutputStream sets parameters once, then resets per frame): a level or a parameter only the bundled library serves */
    unsigned char* buf; size_t total, cap;                      /* everything written so far */
    uint32_t* flushAt; size_t nFlush, flushCap;
    size_t emitted;                                             /* bytes of the frame already made (handed out or pending) */
    uint64_t madeHash;                                          /* FNV-1a over those bytes: a replay into the bundled library must reproduce them (ss_same_prefix) */
    unsigned char* out; size_t outLen, outPos, outCap;          /* made, not yet taken by the caller */
} StreamState;
/* The prefix a GPU-route stream has handed out is followed, after a replay, by what the BUNDLED library's stream writes behind its own version of that prefix.  The two
 * are the same bytes when the bundled library is the release this route reproduces; another libzstd behind $ZSTD_JNI_CPU_LIB may cut its blocks elsewhere, and its later
 * blocks would then continue a prefix the caller does not hold.  So the made bytes are hashed as they are made and the replay compares. */
static uint64_t ss_hash(uint64_t h, const unsigned char* p, size_t n) { size_t i; for (i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001B3ull; } return h; }
#define SS_HASH0 0xCBF29CE484222325ull
static StreamState* g_ss[256];
static pthread_mutex_t g_ss_mu = PTHREAD_MUTEX_INITIALIZER;
static jfieldID g_cs_consumed, g_cs_produced, g_ds_consumed, g_ds_produced;
static StreamState* ss_get(jlong key, int create, int take) {
    StreamState** pp; StreamState* s = NULL;
    if (!key) return NULL;
    pthread_mutex_lock(&g_ss_mu);
    for (pp = &g_ss[st_bucket(key) & 255]; *pp; pp = &(*pp)->next) if ((*pp)->key == key) { s = *pp; if (take) *pp = s->next; break; }
    if (!s && create) { s = (StreamState*)calloc(1, sizeof *s); if (s) { s->key = key; s->level = 3; s->next = g_ss[st_bucket(key) & 255]; g_ss[st_bucket(key) & 255] = s; } }
    pthread_mutex_unlock(&g_ss_mu);
    return s;
}
static void ss_reset(StreamState* s, int level) {
    s->level = level; s->checksum = 0; s->cpuMode = 0; s->started = 0; s->finished = 0; s->total = 0; s->nFlush = 0; s->emitted = 0; s->madeHash = SS_HASH0; s->outLen = s->outPos = 0;
}
static int ss_inside_gpu_frame(const StreamState* s) { return !s->cpuMode && s->started && !s->finished; }     /* started: the first compress / flush call, where ZSTD_compressStream2 leaves zcss_init */
/* a frame made here is out: the stream is as ZSTD_endStream leaves it — parameters kept, a level set meanwhile now in force */
/* the bundled context gets a level that was set inside a frame made here: behind a replay (the bundled stream is then inside the frame at the OLD level, where ZSTD_CCtx_setParameter
 * accepts a level for the next frame) or when the frame is out */
static void ss_forward_level(JNIEnv* env, StreamState* s) {
    jint (*f)(JNIEnv*, jclass, jlong, jint);
    if (!s->fwdLevelPending) return;
    s->fwdLevelPending = 0;
    f = (jint (*)(JNIEnv*, jclass, jlong, jint))cpu_sym(PS("Zstd_setCompressionLevel"));
    if (f) (void)f(env, NULL, s->key, (jint)s->fwdLevel);
}
static void ss_next_frame(StreamState* s) {
    int const lv = s->hasLevelNext ? s->levelNext : s->level, ck = s->checksum, pc = s->paramCpu;
    int const lc = s->hasLevelNext ? (lv < 0 || lv > 3) : s->levelCpu;
    ss_reset(s, lv); s->checksum = ck; s->levelCpu = lc; s->paramCpu = pc; s->hasLevelNext = 0;
    if (lc) s->cpuMode = 1;
}
static void ss_free(StreamState* s) { if (s) { free(s->buf); free(s->flushAt); free(s->out); free(s); } }
static int ss_note_parameter(jlong stream, int what, jint v) {       /* class Zstd's parameter natives on a stream handle: level and checksum are honoured, anything else is the bundled library's */
    StreamState* s = ss_get(stream, 0, 0);
    if (!s) return 0;
    if (ss_inside_gpu_frame(s)) {                                   /* the bytes written so far wait HERE; the bundled context has not seen them and would accept anything */
        if (what != 'l') return -1;                                 /* ZSTD_CCtx_setParameter past zcss_init: only the update-authorised parameters (C/zstd_compress.c ZSTD_isUpdateAuthorized) */
        s->levelNext = v == 0 ? 3 : v; s->hasLevelNext = 1;         /* the level is one of them: this frame keeps its parameters, the next one starts with the new level */
        s->fwdLevel = v; s->fwdLevelPending = 1;
        return 2;
    }
    if (what == 'l') { s->level = v == 0 ? 3 : v; s->levelCpu = (v < 0 || v > 3); if (s->levelCpu) s->cpuMode = 1; }
    else if (what == 'k') s->checksum = (v & 0xFF) != 0;
    else { s->cpuMode = 1; s->paramCpu = 1; return 0; }
    return what == 'k' || !s->levelCpu;
}
static int streams_on_gpu(void) {
    static int state = -1;
    if (state < 0) {
        const char* e = getenv("ZSTD_JNI_GPU_STREAMS");
        if (e && *e == '0' && cpu_sym(PS("Zstd_compressBound"))) state = 0;
        else state = ((e && *e == '1') || per_buffer_on_gpu()) ? 1 : 0;
    }
    return state && gpu_on();
}
static size_t ss_window(int level) { return (size_t)1 << (18 + (level < 1 ? 3 : level)); }
static int ss_out_room(StreamState* s, size_t more) {
    if (s->outPos == s->outLen) s->outPos = s->outLen = 0;
    if (s->outLen + more > s->outCap) { size_t const c = (s->outLen + more) * 2 + 4096; unsigned char* p = (unsigned char*)realloc(s->out, c); if (!p) return 0; s->out = p; s->outCap = c; }
    return 1;
}
static size_t ss_deliver(StreamState* s, char* dst, size_t room) {      /* pending output into the caller's buffer: bytes copied */
    size_t k = s->outLen - s->outPos; if (k > room) k = room;
    if (k) { memcpy(dst, s->out + s->outPos, k); s->outPos += k; }
    return k;
}
static size_t ds_buffered(StreamState* s, char* dst, size_t room, const char* src, size_t avail, size_t* produced, size_t* consumed);      /* frames in pieces without a bundled library: defined at the end of this file */
typedef jlong (*cs_compress_fn)(JNIEnv*, jobject, jlong, jobject, jint, jint, jobject, jint, jint);
typedef jlong (*cs_end_fn)(JNIEnv*, jobject, jlong, jobject, jint, jint);
/* Hand everything buffered so far to the bundled library's stream (flushes where the caller flushed), collecting what it writes; afterwards the stream is the
 * bundled library's.  -1: no bundled library or no way to wrap a buffer; the caller answers with an error code. */
static int ss_replay_to_cpu(JNIEnv* env, jobject obj, StreamState* s) {
    cs_compress_fn cf = (cs_compress_fn)cpu_sym(PS("ZstdDirectBufferCompressingStreamNoFinalizer_compressDirectByteBuffer"));
    cs_end_fn ff = (cs_end_fn)cpu_sym(PS("ZstdDirectBufferCompressingStreamNoFinalizer_flushStream"));
    size_t at = 0, fi = 0, delivered = 0; size_t const scratchCap = 256u << 10;
    unsigned char* scratch; jobject dbuf;
    if (!cf || !ff || !(*env)->NewDirectByteBuffer) return -1;
    scratch = (unsigned char*)malloc(scratchCap); if (!scratch) return -1;
    dbuf = (*env)->NewDirectByteBuffer(env, scratch, (jlong)scratchCap);
    if (!dbuf) { free(scratch); return -1; }
    delivered = s->emitted - (s->outLen - s->outPos);         /* what the caller has already taken of the frame */
    s->outLen = s->outPos = 0;
    while (at < s->total || fi < s->nFlush) {
        size_t const upto = fi < s->nFlush ? s->flushAt[fi] : s->total;
        while (at < upto) {
            size_t const n = upto - at > (1u << 30) ? (1u << 30) : upto - at;
            jobject sbuf = (*env)->NewDirectByteBuffer(env, s->buf + at, (jlong)n);
            jlong r; jint consumed, produced;
            if (!sbuf) { free(scratch); return -1; }
            r = cf(env, obj, s->key, dbuf, 0, (jint)scratchCap, sbuf, 0, (jint)n);
            consumed = (*env)->GetIntField(env, obj, g_cs_consumed); produced = (*env)->GetIntField(env, obj, g_cs_produced);
            if ((*env)->DeleteLocalRef) (*env)->DeleteLocalRef(env, sbuf);
            if (r < 0 || !ss_out_room(s, (size_t)produced)) { free(scratch); return -1; }
            memcpy(s->out + s->outLen, scratch, (size_t)produced); s->outLen += (size_t)produced;
            at += (size_t)consumed;
            if (consumed == 0 && produced == 0) { free(scratch); return -1; }
        }
        if (fi < s->nFlush) {
            for (;;) {
                jlong const r = ff(env, obj, s->key, dbuf, 0, (jint)scratchCap); jint const produced = (*env)->GetIntField(env, obj, g_cs_produced);
                if (r < 0 || !ss_out_room(s, (size_t)produced)) { free(scratch); return -1; }
                memcpy(s->out + s->outLen, scratch, (size_t)produced); s->outLen += (size_t)produced;
                if (r == 0) break;
            }
            fi++;
        }
    }
    if ((*env)->DeleteLocalRef) (*env)->DeleteLocalRef(env, dbuf);
    free(scratch);
    /* What was flushed out earlier came from the GPU route; the bundled stream has just made the same bytes again (the route is byte-identical, and a flushed
     * prefix depends on nothing behind it): they are skipped, what follows is new. */
    if (s->outLen < s->emitted || ss_hash(SS_HASH0, s->out, s->emitted) != s->madeHash) return -1;      /* (not the bytes the caller already holds: no continuation to offer) */
    s->outPos = delivered; s->emitted = 0; s->madeHash = SS_HASH0;
    s->cpuMode = 1; s->total = 0; s->nFlush = 0;
    ss_forward_level(env, s);
    __atomic_fetch_add(&g_stats[2], 1, __ATOMIC_RELAXED);
    return 0;
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferCompressingStreamNoFinalizer_createCStream)(JNIEnv* env, jclass cls) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(PS("ZstdDirectBufferCompressingStreamNoFinalizer_createCStream"));
    jlong const h = f ? f(env, cls) : (jlong)(intptr_t)calloc(1, 16);
    if (h) (void)ss_get(h, 1, 0);
    return h;
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferCompressingStreamNoFinalizer_freeCStream)(JNIEnv* env, jclass cls, jlong stream) {
    jlong (*f)(JNIEnv*, jclass, jlong) = (jlong (*)(JNIEnv*, jclass, jlong))cpu_sym(PS("ZstdDirectBufferCompressingStreamNoFinalizer_freeCStream"));
    ss_free(ss_get(stream, 0, 1));
    if (!stream) return 0;
    if (f) return f(env, cls, stream);
    free((void*)(intptr_t)stream); return 0;
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferCompressingStreamNoFinalizer_recommendedCOutSize)(JNIEnv* env, jclass cls) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(PS("ZstdDirectBufferCompressingStreamNoFinalizer_recommendedCOutSize"));
    return f ? f(env, cls) : (jlong)zjni_compressBound(128u << 10) + 3 + 4;          /* ZSTD_CStreamOutSize(): a block's bound + its header + the checksum */
}
static void cs_fields(JNIEnv* env, jobject obj) {
    jclass const clazz = (*env)->GetObjectClass(env, obj);
    g_cs_consumed = (*env)->GetFieldID(env, clazz, "consumed", "I"); g_cs_produced = (*env)->GetFieldID(env, clazz, "produced", "I");
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferCompressingStreamNoFinalizer_initCStream)(JNIEnv* env, jobject obj, jlong stream, jint level) {
    jlong (*f)(JNIEnv*, jobject, jlong, jint) = (jlong (*)(JNIEnv*, jobject, jlong, jint))cpu_sym(PS("ZstdDirectBufferCompressingStreamNoFinalizer_initCStream"));
    StreamState* s = ss_get(stream, 1, 0);
    cs_fields(env, obj);
    if (s) { ss_reset(s, level == 0 ? 3 : level); s->hasLevelNext = 0; if (level < 0 || level > 3 || !streams_on_gpu()) s->cpuMode = 1; }
    if (!f && (level < 0 || level > 3)) return -(jlong)ZJNI_ERROR_unsupported;     /* no bundled library and a level this route does not make: said here, not at the first write */
    return f ? f(env, obj, stream, level) : 0;
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferCompressingStreamNoFinalizer_initCStreamWithDict)(JNIEnv* env, jobject obj, jlong stream, jbyteArray dict, jint dict_size, jint level) {
    jlong (*f)(JNIEnv*, jobject, jlong, jbyteArray, jint, jint) = (jlong (*)(JNIEnv*, jobject, jlong, jbyteArray, jint, jint))cpu_sym(PS("ZstdDirectBufferCompressingStreamNoFinalizer_initCStreamWithDict"));
    StreamState* s = ss_get(stream, 1, 0);
    cs_fields(env, obj);
    if (s) { ss_reset(s, level); s->cpuMode = 1; }              /* streams with a dictionary: the bundled library's */
    return f ? f(env, obj, stream, dict, dict_size, level) : -(jlong)ZJNI_ERROR_unsupported;
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferCompressingStreamNoFinalizer_initCStreamWithFastDict)(JNIEnv* env, jobject obj, jlong stream, jobject dict) {
    jlong (*f)(JNIEnv*, jobject, jlong, jobject) = (jlong (*)(JNIEnv*, jobject, jlong, jobject))cpu_sym(PS("ZstdDirectBufferCompressingStreamNoFinalizer_initCStreamWithFastDict"));
    StreamState* s = ss_get(stream, 1, 0);
    cs_fields(env, obj);
    if (s) { ss_reset(s, 3); s->cpuMode = 1; }
    return f ? f(env, obj, stream, dict) : -(jlong)ZJNI_ERROR_unsupported;
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferCompressingStreamNoFinalizer_compressDirectByteBuffer)(JNIEnv* env, jobject obj, jlong stream, jobject dst_buf, jint dst_offset, jint dst_size, jobject src_buf, jint src_offset, jint src_size) {
    cs_compress_fn f = (cs_compress_fn)cpu_sym(PS("ZstdDirectBufferCompressingStreamNoFinalizer_compressDirectByteBuffer"));
    StreamState* s = ss_get(stream, 0, 0);
    jlong dst_cap, src_cap; char* dp; char* sp;
    if (s == NULL || (s->cpuMode && s->outPos == s->outLen)) return f ? f(env, obj, stream, dst_buf, dst_offset, dst_size, src_buf, src_offset, src_size) : E_MEM;
    dst_cap = (*env)->GetDirectBufferCapacity(env, dst_buf);                        /* the reference's checks, in its order (N/jni_directbuffercompress_zstd.c:103-110) */
    if (dst_offset + dst_size > dst_cap) return E_DST;
    src_cap = (*env)->GetDirectBufferCapacity(env, src_buf);
    if (src_offset + src_size > src_cap) return E_SRC;
    dp = (char*)(*env)->GetDirectBufferAddress(env, dst_buf); if (dp == NULL) return E_MEM;
    sp = (char*)(*env)->GetDirectBufferAddress(env, src_buf); if (sp == NULL) return E_MEM;
    if (s->cpuMode) {                                                               /* replayed earlier: what the bundled stream wrote then goes out first */
        size_t const k = ss_deliver(s, dp + dst_offset, (size_t)dst_size);
        (*env)->SetIntField(env, obj, g_cs_consumed, 0); (*env)->SetIntField(env, obj, g_cs_produced, (jint)k);
        return 1;
    }
    s->started = 1;
    if (s->total + (size_t)src_size > ss_window(s->level)) {                         /* outgrows the window: the bundled library's stream takes over */
        if (ss_replay_to_cpu(env, obj, s) != 0) return -(jlong)ZJNI_ERROR_unsupported;
        {   size_t const k = ss_deliver(s, dp + dst_offset, (size_t)dst_size);
            if (s->outPos < s->outLen) { (*env)->SetIntField(env, obj, g_cs_consumed, 0); (*env)->SetIntField(env, obj, g_cs_produced, (jint)k); return 1; }
            {   jlong const r = f(env, obj, stream, dst_buf, dst_offset + (jint)k, dst_size - (jint)k, src_buf, src_offset, src_size);
                (*env)->SetIntField(env, obj, g_cs_produced, (*env)->GetIntField(env, obj, g_cs_produced) + (jint)k);
                return r; } }
    }
    if (s->total + (size_t)src_size > s->cap) {
        size_t c = (s->total + (size_t)src_size) * 2 + 65536; unsigned char* p;
        if (c > ss_window(s->level)) c = ss_window(s->level);
        p = (unsigned char*)realloc(s->buf, c); if (!p) return E_MEM;
        s->buf = p; s->cap = c;
    }
    if (src_size > 0) memcpy(s->buf + s->total, sp + src_offset, (size_t)src_size);
    s->total += (size_t)src_size;
    (*env)->SetIntField(env, obj, g_cs_consumed, src_size); (*env)->SetIntField(env, obj, g_cs_produced, 0);
    return (jlong)((128u << 10) - (s->total & ((128u << 10) - 1)));                  /* a hint for the next write, as ZSTD_compressStream gives one */
}
static jlong cs_flush_or_end(JNIEnv* env, jobject obj, jlong stream, jobject dst_buf, jint dst_offset, jint dst_size, int end) {
    const char* const name = end ? PS("ZstdDirectBufferCompressingStreamNoFinalizer_endStream") : PS("ZstdDirectBufferCompressingStreamNoFinalizer_flushStream");
    cs_end_fn f = (cs_end_fn)cpu_sym(name);
    StreamState* s = ss_get(stream, 0, 0);
    jlong dst_cap; char* dp; size_t k = 0;
    if (s == NULL || (s->cpuMode && s->outPos == s->outLen)) return f ? f(env, obj, stream, dst_buf, dst_offset, dst_size) : E_MEM;
    dst_cap = (*env)->GetDirectBufferCapacity(env, dst_buf);
    if (dst_offset + dst_size > dst_cap) return E_DST;
    dp = (char*)(*env)->GetDirectBufferAddress(env, dst_buf);
    if (dp == NULL) return E_MEM;
    if (!s->cpuMode && s->outPos == s->outLen && !s->finished) {                     /* nothing pending: make the frame's next part */
        int const knownEmpty = end && !s->started && s->total == 0;
        size_t const cap = s->total + (s->total >> 8) + 4096 + 64 * (s->nFlush + 4);
        unsigned char* tmp; size_t r;
        if (!end) {
            s->started = 1;
            if (s->total > (s->nFlush ? s->flushAt[s->nFlush - 1] : 0)) {            /* something was written since the last flush */
                if (s->nFlush == s->flushCap) { size_t const c = s->flushCap * 2 + 16; uint32_t* p = (uint32_t*)realloc(s->flushAt, c * sizeof *p); if (!p) return E_MEM; s->flushAt = p; s->flushCap = c; }
                s->flushAt[s->nFlush++] = (uint32_t)s->total;
            } else { (*env)->SetIntField(env, obj, g_cs_produced, 0); return 0; }  /* a flush with nothing buffered writes nothing */
        }
        tmp = (unsigned char*)malloc(cap); if (!tmp) return E_MEM;
        r = zjni_compress_stream(tmp, cap, s->buf, s->total, s->level, s->checksum, s->flushAt, s->nFlush, end, knownEmpty);
        if (zjni_isError(r)) {
            free(tmp);
            if (zjni_getErrorCode(r) >= 200 && ss_replay_to_cpu(env, obj, s) == 0) {   /* the GPU declined: the bundled stream makes the frame from the start */
                k = ss_deliver(s, dp + dst_offset, (size_t)dst_size);
                if (s->outPos < s->outLen) { (*env)->SetIntField(env, obj, g_cs_produced, (jint)k); return (jlong)(s->outLen - s->outPos); }
                {   jlong const rr = f(env, obj, stream, dst_buf, dst_offset + (jint)k, dst_size - (jint)k);
                    (*env)->SetIntField(env, obj, g_cs_produced, (*env)->GetIntField(env, obj, g_cs_produced) + (jint)k);
                    return rr; }
            }
            return (jlong)r;
        }
        if (r < s->emitted || !ss_out_room(s, r - s->emitted)) { free(tmp); return E_MEM; }
        memcpy(s->out + s->outLen, tmp + s->emitted, r - s->emitted); s->madeHash = ss_hash(s->madeHash, tmp + s->emitted, r - s->emitted); s->outLen += r - s->emitted; s->emitted = r;
        free(tmp);
        if (end) { s->finished = 1; __atomic_fetch_add(&g_stats[0], 1, __ATOMIC_RELAXED); }
    }
    k = ss_deliver(s, dp + dst_offset, (size_t)dst_size);
    (*env)->SetIntField(env, obj, g_cs_produced, (jint)k);
    if (s->outPos < s->outLen) return (jlong)(s->outLen - s->outPos);                /* bytes still to be taken: the caller comes back with room */
    if (s->cpuMode) {                                                               /* the replayed part is out: the bundled stream does this flush / end itself */
        jlong const rr = f(env, obj, stream, dst_buf, dst_offset + (jint)k, dst_size - (jint)k);
        (*env)->SetIntField(env, obj, g_cs_produced, (*env)->GetIntField(env, obj, g_cs_produced) + (jint)k);
        return rr;
    }
    if (s->finished) { ss_forward_level(env, s); ss_next_frame(s); }                /* the frame is out: the next write starts a new one, as ZSTD_endStream leaves the stream */
    return 0;
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferCompressingStreamNoFinalizer_flushStream)(JNIEnv* env, jobject obj, jlong stream, jobject dst_buf, jint dst_offset, jint dst_size) {
    return cs_flush_or_end(env, obj, stream, dst_buf, dst_offset, dst_size, 0);
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferCompressingStreamNoFinalizer_endStream)(JNIEnv* env, jobject obj, jlong stream, jobject dst_buf, jint dst_offset, jint dst_size) {
    return cs_flush_or_end(env, obj, stream, dst_buf, dst_offset, dst_size, 1);
}

/* ==== ZstdDirectBufferDecompressingStreamNoFinalizer.decompressStreamNative (N/jni_directbufferdecompress_zstd.c:58-79) ==========================
 * At a frame boundary, when the source buffer holds a COMPLETE zstd frame and the destination has room for all it can decode to, the frame goes to the
 * batch decoder in one piece (zjni_decompress) — also frames without a content size, whose bound comes from their block headers — and the native
 * answers as ZSTD_decompressStream does after a frame's last byte: consumed = the frame, produced = its content, return 0.  Anything else (a frame in pieces,
 * a small destination, skippable frames, a dictionary) is the bundled library's stream, and once it is inside a frame it stays there until the frame ends. */
typedef jlong (*ds_fn)(JNIEnv*, jobject, jlong, jobject, jint, jint, jobject, jint, jint);
JNIEXPORT jlong JNICALL P(ZstdDirectBufferDecompressingStreamNoFinalizer_createDStreamNative)(JNIEnv* env, jclass cls) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(PS("ZstdDirectBufferDecompressingStreamNoFinalizer_createDStreamNative"));
    jlong const h = f ? f(env, cls) : (jlong)(intptr_t)calloc(1, 16);
    if (h) (void)ss_get(h, 1, 0);
    return h;
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferDecompressingStreamNoFinalizer_freeDStreamNative)(JNIEnv* env, jclass cls, jlong stream) {
    jlong (*f)(JNIEnv*, jclass, jlong) = (jlong (*)(JNIEnv*, jclass, jlong))cpu_sym(PS("ZstdDirectBufferDecompressingStreamNoFinalizer_freeDStreamNative"));
    ss_free(ss_get(stream, 0, 1));
    if (!stream) return 0;
    if (f) return f(env, cls, stream);
    free((void*)(intptr_t)stream); return 0;
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferDecompressingStreamNoFinalizer_initDStreamNative)(JNIEnv* env, jobject obj, jlong stream) {
    jlong (*f)(JNIEnv*, jobject, jlong) = (jlong (*)(JNIEnv*, jobject, jlong))cpu_sym(PS("ZstdDirectBufferDecompressingStreamNoFinalizer_initDStreamNative"));
    StreamState* s = ss_get(stream, 1, 0);
    jclass const clazz = (*env)->GetObjectClass(env, obj);
    g_ds_consumed = (*env)->GetFieldID(env, clazz, "consumed", "I"); g_ds_produced = (*env)->GetFieldID(env, clazz, "produced", "I");
    if (s) { ss_reset(s, 3); s->cpuMode = streams_on_gpu() ? 0 : 1; }               /* (started = 1 below: the bundled stream is inside a frame) */
    return f ? f(env, obj, stream) : 0;
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferDecompressingStreamNoFinalizer_decompressStreamNative)(JNIEnv* env, jobject obj, jlong stream, jobject dst_buf, jint dst_offset, jint dst_size, jobject src_buf, jint src_offset, jint src_size) {
    ds_fn f = (ds_fn)cpu_sym(PS("ZstdDirectBufferDecompressingStreamNoFinalizer_decompressStreamNative"));
    StreamState* s = ss_get(stream, 0, 0);
    if (s && !s->cpuMode && !s->started && src_size > 0) {
        jlong const dst_cap = (*env)->GetDirectBufferCapacity(env, dst_buf), src_cap = (*env)->GetDirectBufferCapacity(env, src_buf);
        char* const dp = (char*)(*env)->GetDirectBufferAddress(env, dst_buf); char* const sp = (char*)(*env)->GetDirectBufferAddress(env, src_buf);
        if (dst_offset + dst_size > dst_cap) return E_DST;
        if (src_offset + src_size > src_cap) return E_SRC;
        if (dp && sp) {
            unsigned long long content = 0, bound = 0;
            size_t const ext = zjni_frame_extent(sp + src_offset, (size_t)src_size, &content, &bound);
            if (ext && (bound <= (unsigned long long)dst_size || bound <= (64ull << 20))) {
                /* a frame without a content size is bounded by its block headers only (128 KiB per compressed block): decoded aside when that bound exceeds the
                 * caller's room, and handed over if what it really holds fits */
                int const aside = bound > (unsigned long long)dst_size;
                char* const to = aside ? (char*)malloc((size_t)bound + 1) : dp + dst_offset;
                size_t const r = to ? zjni_decompress(to, aside ? (size_t)bound : (size_t)dst_size, sp + src_offset, ext) : (size_t)E_MEM;
                if (!zjni_isError(r) && r <= (size_t)dst_size) {
                    if (aside) { memcpy(dp + dst_offset, to, r); free(to); }
                    (*env)->SetIntField(env, obj, g_ds_consumed, (jint)ext); (*env)->SetIntField(env, obj, g_ds_produced, (jint)r);
                    __atomic_fetch_add(&g_stats[0], 1, __ATOMIC_RELAXED);
                    return 0;
                }
                if (aside) free(to);
                if (zjni_isError(r) && zjni_getErrorCode(r) < 200 && zjni_getErrorCode(r) != 70) return (jlong)r;     /* the frame is damaged: libzstd's code */
            }
        }
    }
    if (!f) {                                                                       /* no bundled stream: the frame is collected and decoded here (ds_buffered) */
        jlong const dst_cap = (*env)->GetDirectBufferCapacity(env, dst_buf), src_cap = (*env)->GetDirectBufferCapacity(env, src_buf);
        char* const dp = (char*)(*env)->GetDirectBufferAddress(env, dst_buf); char* const sp = (char*)(*env)->GetDirectBufferAddress(env, src_buf);
        size_t produced, consumed, r;
        if (!s || !gpu_on()) return -(jlong)ZJNI_ERROR_unsupported;
        if (dst_offset + dst_size > dst_cap) return E_DST;
        if (src_offset + src_size > src_cap) return E_SRC;
        if (!dp || !sp) return E_MEM;
        r = ds_buffered(s, dp + dst_offset, (size_t)dst_size, sp + src_offset, (size_t)src_size, &produced, &consumed);
        (*env)->SetIntField(env, obj, g_ds_consumed, (jint)consumed); (*env)->SetIntField(env, obj, g_ds_produced, (jint)produced);
        return (jlong)r;
    }
    {   jlong const r = f(env, obj, stream, dst_buf, dst_offset, dst_size, src_buf, src_offset, src_size);
        if (s) s->started = (r > 0);                                                /* > 0: inside a frame (more input or more room wanted); 0: at a boundary again */
        return r; }
}

/* ==== ZstdOutputStreamNoFinalizer / ZstdInputStreamNoFinalizer (N/jni_outputstream_zstd.c, N/jni_inputstream_zstd.c): the heap-array twins of the classes above ====
 * Same route, same state: a compress stream is buffered until it is flushed or closed and zjni_compress_stream makes the frame ZSTD_compressStream2 would have
 * made; srcPos / dstPos (long fields of the Java object) are what consumed / produced are there.  Parameters arrive through class Zstd's natives on the stream
 * handle before the first write (ss_note_parameter: level and checksum are honoured, anything else — and Zstd.loadDictCompress / loadFastDictCompress — makes the
 * stream the bundled library's for good); resetCStream starts a frame and keeps them, as ZSTD_CCtx_reset(session_only) does.  The arrays are copied with
 * Get / SetByteArrayRegion, never pinned across a GPU call.  A stream that outgrows the window or meets a GPU that declines is replayed into the bundled
 * library's stream through ITS natives (byte arrays made for the purpose). */
static jfieldID g_os_src, g_os_dst, g_is_src, g_is_dst;
typedef jint (*os_compress_fn)(JNIEnv*, jobject, jlong, jbyteArray, jint, jbyteArray, jint);
typedef jint (*os_end_fn)(JNIEnv*, jobject, jlong, jbyteArray, jint);
JNIEXPORT jlong JNICALL P(ZstdOutputStreamNoFinalizer_createCStream)(JNIEnv* env, jclass cls) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(PS("ZstdOutputStreamNoFinalizer_createCStream"));
    jlong const h = f ? f(env, cls) : (jlong)(intptr_t)calloc(1, 16);
    if (h) (void)ss_get(h, 1, 0);
    return h;
}
JNIEXPORT jint JNICALL P(ZstdOutputStreamNoFinalizer_freeCStream)(JNIEnv* env, jclass cls, jlong stream) {
    jint (*f)(JNIEnv*, jclass, jlong) = (jint (*)(JNIEnv*, jclass, jlong))cpu_sym(PS("ZstdOutputStreamNoFinalizer_freeCStream"));
    ss_free(ss_get(stream, 0, 1));
    if (!stream) return 0;
    if (f) return f(env, cls, stream);
    free((void*)(intptr_t)stream); return 0;
}
JNIEXPORT jlong JNICALL P(ZstdOutputStreamNoFinalizer_recommendedCOutSize)(JNIEnv* env, jclass cls) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(PS("ZstdOutputStreamNoFinalizer_recommendedCOutSize"));
    return f ? f(env, cls) : (jlong)zjni_compressBound(128u << 10) + 3 + 4;
}
JNIEXPORT jint JNICALL P(ZstdOutputStreamNoFinalizer_resetCStream)(JNIEnv* env, jobject obj, jlong stream) {
    jint (*f)(JNIEnv*, jobject, jlong) = (jint (*)(JNIEnv*, jobject, jlong))cpu_sym(PS("ZstdOutputStreamNoFinalizer_resetCStream"));
    StreamState* s = ss_get(stream, 1, 0);
    jclass const clazz = (*env)->GetObjectClass(env, obj);
    g_os_src = (*env)->GetFieldID(env, clazz, "srcPos", "J"); g_os_dst = (*env)->GetFieldID(env, clazz, "dstPos", "J");
    if (s) {                                                    /* a new frame; level, checksum and what only the bundled library serves stay */
        ss_forward_level(env, s); ss_next_frame(s);
        if (s->level == 0) s->level = 3;
        if (s->levelCpu || s->paramCpu || !streams_on_gpu()) s->cpuMode = 1;
    }
    return f ? f(env, obj, stream) : 0;
}
/* the stream marked as the bundled library's before anything is forwarded: dictionaries on a stream handle */
static int ss_mark_cpu(jlong stream) {                          /* -1: inside a frame made here — ZSTD_CCtx_loadDictionary / refCDict answer stage_wrong there (C/zstd_compress.c: "Can't load a dictionary when cctx is not in init stage") */
    StreamState* s = ss_get(stream, 0, 0);
    if (s && ss_inside_gpu_frame(s)) return -1;
    if (s) { s->paramCpu = 1; s->cpuMode = 1; }
    return 0;
}
JNIEXPORT jint JNICALL P(Zstd_loadDictCompress)(JNIEnv* env, jclass cls, jlong stream, jbyteArray dict, jint dict_size) {
    jint (*f)(JNIEnv*, jclass, jlong, jbyteArray, jint) = (jint (*)(JNIEnv*, jclass, jlong, jbyteArray, jint))cpu_sym(PS("Zstd_loadDictCompress"));
    if (ss_mark_cpu(stream) < 0) return -(jint)60;
    return f ? f(env, cls, stream, dict, dict_size) : -(jint)ZJNI_ERROR_unsupported;
}
JNIEXPORT jint JNICALL P(Zstd_loadFastDictCompress)(JNIEnv* env, jclass cls, jlong stream, jobject dict) {
    jint (*f)(JNIEnv*, jclass, jlong, jobject) = (jint (*)(JNIEnv*, jclass, jlong, jobject))cpu_sym(PS("Zstd_loadFastDictCompress"));
    if (ss_mark_cpu(stream) < 0) return -(jint)60;
    return f ? f(env, cls, stream, dict) : -(jint)ZJNI_ERROR_unsupported;
}
JNIEXPORT jint JNICALL P(Zstd_loadDictDecompress)(JNIEnv* env, jclass cls, jlong stream, jbyteArray dict, jint dict_size) {
    jint (*f)(JNIEnv*, jclass, jlong, jbyteArray, jint) = (jint (*)(JNIEnv*, jclass, jlong, jbyteArray, jint))cpu_sym(PS("Zstd_loadDictDecompress"));
    if (ss_mark_cpu(stream) < 0) return -(jint)60;
    return f ? f(env, cls, stream, dict, dict_size) : -(jint)ZJNI_ERROR_unsupported;
}
JNIEXPORT jint JNICALL P(Zstd_loadFastDictDecompress)(JNIEnv* env, jclass cls, jlong stream, jobject dict) {
    jint (*f)(JNIEnv*, jclass, jlong, jobject) = (jint (*)(JNIEnv*, jclass, jlong, jobject))cpu_sym(PS("Zstd_loadFastDictDecompress"));
    if (ss_mark_cpu(stream) < 0) return -(jint)60;
    return f ? f(env, cls, stream, dict) : -(jint)ZJNI_ERROR_unsupported;
}
/* pending output into dst[0, dst_size): bytes copied (dstPos) */
static size_t os_deliver(JNIEnv* env, StreamState* s, jbyteArray dst, jint dst_size) {
    size_t k = s->outLen - s->outPos; if (k > (size_t)dst_size) k = (size_t)dst_size;
    if (k) { (*env)->SetByteArrayRegion(env, dst, 0, (jsize)k, (const jbyte*)(s->out + s->outPos)); s->outPos += k; }
    return k;
}
/* everything buffered so far into the bundled library's stream through its own natives, collecting what it writes (ss_replay_to_cpu's twin): -1 when that is impossible */
static int os_replay_to_cpu(JNIEnv* env, jobject obj, StreamState* s) {
    os_compress_fn cf = (os_compress_fn)cpu_sym(PS("ZstdOutputStreamNoFinalizer_compressStream"));
    os_end_fn ff = (os_end_fn)cpu_sym(PS("ZstdOutputStreamNoFinalizer_flushStream"));
    size_t at = 0, fi = 0, delivered; jint const scratchCap = 256 << 10, pieceCap = 1 << 20;
    jbyteArray darr, sarr; unsigned char* tmp;
    if (!cf || !ff || !(*env)->NewByteArray) return -1;
    darr = (*env)->NewByteArray(env, scratchCap); sarr = (*env)->NewByteArray(env, pieceCap); tmp = (unsigned char*)malloc((size_t)scratchCap);
    if (!darr || !sarr || !tmp) { free(tmp); return -1; }
    delivered = s->emitted - (s->outLen - s->outPos);
    s->outLen = s->outPos = 0;
    while (at < s->total || fi < s->nFlush) {
        size_t const upto = fi < s->nFlush ? s->flushAt[fi] : s->total;
        while (at < upto) {
            jint const n = upto - at > (size_t)pieceCap ? pieceCap : (jint)(upto - at);
            jint r; jlong sp, dp;
            (*env)->SetByteArrayRegion(env, sarr, 0, n, (const jbyte*)(s->buf + at));
            (*env)->SetLongField(env, obj, g_os_src, 0);
            r = cf(env, obj, s->key, darr, scratchCap, sarr, n);
            sp = (*env)->GetLongField(env, obj, g_os_src); dp = (*env)->GetLongField(env, obj, g_os_dst);
            if (r < 0 || !ss_out_room(s, (size_t)dp)) { free(tmp); return -1; }
            if (dp) { (*env)->GetByteArrayRegion(env, darr, 0, (jsize)dp, (jbyte*)tmp); memcpy(s->out + s->outLen, tmp, (size_t)dp); s->outLen += (size_t)dp; }
            at += (size_t)sp;
            if (sp == 0 && dp == 0) { free(tmp); return -1; }
        }
        if (fi < s->nFlush) {
            for (;;) {
                jint const r = ff(env, obj, s->key, darr, scratchCap); jlong const dp = (*env)->GetLongField(env, obj, g_os_dst);
                if (r < 0 || !ss_out_room(s, (size_t)dp)) { free(tmp); return -1; }
                if (dp) { (*env)->GetByteArrayRegion(env, darr, 0, (jsize)dp, (jbyte*)tmp); memcpy(s->out + s->outLen, tmp, (size_t)dp); s->outLen += (size_t)dp; }
                if (r == 0) break;
            }
            fi++;
        }
    }
    if ((*env)->DeleteLocalRef) { (*env)->DeleteLocalRef(env, darr); (*env)->DeleteLocalRef(env, sarr); }
    free(tmp);
    if (s->outLen < s->emitted || ss_hash(SS_HASH0, s->out, s->emitted) != s->madeHash) return -1;      /* (the flushed prefix came from the GPU route and the bundled stream has made the same bytes again: skipped — or it has not: no continuation to offer) */
    s->outPos = delivered; s->emitted = 0; s->madeHash = SS_HASH0;
    s->cpuMode = 1; s->total = 0; s->nFlush = 0;
    ss_forward_level(env, s);
    __atomic_fetch_add(&g_stats[2], 1, __ATOMIC_RELAXED);
    return 0;
}
JNIEXPORT jint JNICALL P(ZstdOutputStreamNoFinalizer_compressStream)(JNIEnv* env, jobject obj, jlong stream, jbyteArray dst, jint dst_size, jbyteArray src, jint src_size) {
    os_compress_fn f = (os_compress_fn)cpu_sym(PS("ZstdOutputStreamNoFinalizer_compressStream"));
    StreamState* s = ss_get(stream, 0, 0);
    jlong src_pos; size_t take;
    if (s == NULL || !g_os_src || (s->cpuMode && s->outPos == s->outLen)) return f ? f(env, obj, stream, dst, dst_size, src, src_size) : (jint)E_MEM;
    src_pos = (*env)->GetLongField(env, obj, g_os_src);
    if (src_pos < 0 || src_pos > src_size || src_size > (*env)->GetArrayLength(env, src) || dst_size > (*env)->GetArrayLength(env, dst)) return (jint)E_SRC;
    if (s->cpuMode) {                                                               /* replayed earlier: what the bundled stream wrote then goes out first, nothing is consumed */
        size_t const k = os_deliver(env, s, dst, dst_size);
        (*env)->SetLongField(env, obj, g_os_dst, (jlong)k);
        return 1;
    }
    take = (size_t)(src_size - src_pos);
    s->started = 1;
    if (s->total + take > ss_window(s->level)) {                                    /* outgrows the window: the bundled library's stream takes over */
        if (os_replay_to_cpu(env, obj, s) != 0) return -(jint)ZJNI_ERROR_unsupported;
        (*env)->SetLongField(env, obj, g_os_src, src_pos);                          /* (the replay used the field) */
        {   size_t const k = os_deliver(env, s, dst, dst_size);
            if (s->outPos < s->outLen || !f) { (*env)->SetLongField(env, obj, g_os_dst, (jlong)k); return 1; }
            (*env)->SetLongField(env, obj, g_os_dst, (jlong)k); return 1; }         /* the caller's loop comes back with the same source: the bundled stream takes it then */
    }
    if (s->total + take > s->cap) {
        size_t c = (s->total + take) * 2 + 65536; unsigned char* p;
        if (c > ss_window(s->level)) c = ss_window(s->level);
        p = (unsigned char*)realloc(s->buf, c); if (!p) return (jint)E_MEM;
        s->buf = p; s->cap = c;
    }
    if (take) (*env)->GetByteArrayRegion(env, src, (jsize)src_pos, (jsize)take, (jbyte*)(s->buf + s->total));
    s->total += take;
    (*env)->SetLongField(env, obj, g_os_src, (jlong)src_size); (*env)->SetLongField(env, obj, g_os_dst, 0);
    return (jint)((128u << 10) - (s->total & ((128u << 10) - 1)));
}
static jint os_flush_or_end(JNIEnv* env, jobject obj, jlong stream, jbyteArray dst, jint dst_size, int end) {
    const char* const name = end ? PS("ZstdOutputStreamNoFinalizer_endStream") : PS("ZstdOutputStreamNoFinalizer_flushStream");
    os_end_fn f = (os_end_fn)cpu_sym(name);
    StreamState* s = ss_get(stream, 0, 0);
    size_t k = 0;
    if (s == NULL || !g_os_dst || (s->cpuMode && s->outPos == s->outLen)) return f ? f(env, obj, stream, dst, dst_size) : (jint)E_MEM;
    if (dst_size > (*env)->GetArrayLength(env, dst)) return (jint)E_DST;
    if (!s->cpuMode && s->outPos == s->outLen && !s->finished) {                     /* nothing pending: make the frame's next part */
        int const knownEmpty = end && !s->started && s->total == 0;
        size_t const cap = s->total + (s->total >> 8) + 4096 + 64 * (s->nFlush + 4);
        unsigned char* tmp; size_t r;
        if (!end) {
            s->started = 1;
            if (s->total > (s->nFlush ? s->flushAt[s->nFlush - 1] : 0)) {
                if (s->nFlush == s->flushCap) { size_t const c = s->flushCap * 2 + 16; uint32_t* p = (uint32_t*)realloc(s->flushAt, c * sizeof *p); if (!p) return (jint)E_MEM; s->flushAt = p; s->flushCap = c; }
                s->flushAt[s->nFlush++] = (uint32_t)s->total;
            } else { (*env)->SetLongField(env, obj, g_os_dst, 0); return 0; }      /* a flush with nothing buffered writes nothing */
        }
        tmp = (unsigned char*)malloc(cap); if (!tmp) return (jint)E_MEM;
        r = zjni_compress_stream(tmp, cap, s->buf, s->total, s->level, s->checksum, s->flushAt, s->nFlush, end, knownEmpty);
        if (zjni_isError(r)) {
            free(tmp);
            if (zjni_getErrorCode(r) >= 200 && os_replay_to_cpu(env, obj, s) == 0) {   /* the GPU declined: the bundled stream makes the frame from the start */
                k = os_deliver(env, s, dst, dst_size);
                (*env)->SetLongField(env, obj, g_os_dst, (jlong)k);
                if (s->outPos < s->outLen) return (jint)(s->outLen - s->outPos);
                {   jbyteArray rest = dst; jint room = dst_size - (jint)k; jint rr;
                    if (k == 0) { rr = f(env, obj, stream, dst, dst_size); return rr; }
                    /* part of the caller's array is taken: the bundled stream writes the rest through an array of its own */
                    rest = (*env)->NewByteArray(env, room > 0 ? room : 1); if (!rest) return (jint)E_MEM;
                    rr = f(env, obj, stream, rest, room);
                    {   jlong const dp = (*env)->GetLongField(env, obj, g_os_dst);
                        if (dp > 0) { jbyte* b = (jbyte*)malloc((size_t)dp); if (!b) return (jint)E_MEM; (*env)->GetByteArrayRegion(env, rest, 0, (jsize)dp, b); (*env)->SetByteArrayRegion(env, dst, (jsize)k, (jsize)dp, b); free(b); }
                        (*env)->SetLongField(env, obj, g_os_dst, dp + (jlong)k); }
                    if ((*env)->DeleteLocalRef) (*env)->DeleteLocalRef(env, rest);
                    return rr; }
            }
            return (jint)r;
        }
        if (r < s->emitted || !ss_out_room(s, r - s->emitted)) { free(tmp); return (jint)E_MEM; }
        memcpy(s->out + s->outLen, tmp + s->emitted, r - s->emitted); s->madeHash = ss_hash(s->madeHash, tmp + s->emitted, r - s->emitted); s->outLen += r - s->emitted; s->emitted = r;
        free(tmp);
        if (end) { s->finished = 1; __atomic_fetch_add(&g_stats[0], 1, __ATOMIC_RELAXED); }
    }
    k = os_deliver(env, s, dst, dst_size);
    (*env)->SetLongField(env, obj, g_os_dst, (jlong)k);
    if (s->outPos < s->outLen) return (jint)(s->outLen - s->outPos);                 /* bytes still to be taken: the caller's loop comes back */
    if (s->cpuMode) {                                                               /* the replayed part is out: the bundled stream does this flush / end itself — on the caller's next turn when bytes went out now */
        if (k) return 1;
        return f ? f(env, obj, stream, dst, dst_size) : 0;
    }
    if (s->finished) { ss_forward_level(env, s); ss_next_frame(s); }
    return 0;
}
JNIEXPORT jint JNICALL P(ZstdOutputStreamNoFinalizer_flushStream)(JNIEnv* env, jobject obj, jlong stream, jbyteArray dst, jint dst_size) { return os_flush_or_end(env, obj, stream, dst, dst_size, 0); }
JNIEXPORT jint JNICALL P(ZstdOutputStreamNoFinalizer_endStream)(JNIEnv* env, jobject obj, jlong stream, jbyteArray dst, jint dst_size) { return os_flush_or_end(env, obj, stream, dst, dst_size, 1); }
/* ---- ZstdInputStreamNoFinalizer.decompressStream (N/jni_inputstream_zstd.c:68-93): at a frame boundary, a COMPLETE frame in src[srcPos, src_size) whose content
 * fits dst[dstPos, dst_size) goes to the batch decoder in one piece (as in ZstdDirectBufferDecompressingStreamNoFinalizer above); anything else is the bundled
 * library's stream, which keeps the frame once it is inside one. */
typedef jint (*is_fn)(JNIEnv*, jobject, jlong, jbyteArray, jint, jbyteArray, jint);
JNIEXPORT jlong JNICALL P(ZstdInputStreamNoFinalizer_createDStream)(JNIEnv* env, jclass cls) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(PS("ZstdInputStreamNoFinalizer_createDStream"));
    jlong const h = f ? f(env, cls) : (jlong)(intptr_t)calloc(1, 16);
    if (h) (void)ss_get(h, 1, 0);
    return h;
}
JNIEXPORT jint JNICALL P(ZstdInputStreamNoFinalizer_freeDStream)(JNIEnv* env, jclass cls, jlong stream) {
    jint (*f)(JNIEnv*, jclass, jlong) = (jint (*)(JNIEnv*, jclass, jlong))cpu_sym(PS("ZstdInputStreamNoFinalizer_freeDStream"));
    ss_free(ss_get(stream, 0, 1));
    if (!stream) return 0;
    if (f) return f(env, cls, stream);
    free((void*)(intptr_t)stream); return 0;
}
JNIEXPORT jlong JNICALL P(ZstdInputStreamNoFinalizer_recommendedDInSize)(JNIEnv* env, jclass cls) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(PS("ZstdInputStreamNoFinalizer_recommendedDInSize"));
    return f ? f(env, cls) : (jlong)((128u << 10) + 3);                              /* ZSTD_DStreamInSize(): a block and its header */
}
JNIEXPORT jlong JNICALL P(ZstdInputStreamNoFinalizer_recommendedDOutSize)(JNIEnv* env, jclass cls) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(PS("ZstdInputStreamNoFinalizer_recommendedDOutSize"));
    return f ? f(env, cls) : (jlong)(128u << 10);                                    /* ZSTD_DStreamOutSize() */
}
JNIEXPORT jint JNICALL P(ZstdInputStreamNoFinalizer_initDStream)(JNIEnv* env, jobject obj, jlong stream) {
    jint (*f)(JNIEnv*, jobject, jlong) = (jint (*)(JNIEnv*, jobject, jlong))cpu_sym(PS("ZstdInputStreamNoFinalizer_initDStream"));
    StreamState* s = ss_get(stream, 1, 0);
    jclass const clazz = (*env)->GetObjectClass(env, obj);
    g_is_src = (*env)->GetFieldID(env, clazz, "srcPos", "J"); g_is_dst = (*env)->GetFieldID(env, clazz, "dstPos", "J");
    if (s) { int const pc = s->paramCpu; ss_reset(s, 3); s->paramCpu = pc; s->cpuMode = (pc || !streams_on_gpu()) ? 1 : 0; }
    return f ? f(env, obj, stream) : 0;
}
JNIEXPORT jint JNICALL P(ZstdInputStreamNoFinalizer_decompressStream)(JNIEnv* env, jobject obj, jlong stream, jbyteArray dst, jint dst_size, jbyteArray src, jint src_size) {
    is_fn f = (is_fn)cpu_sym(PS("ZstdInputStreamNoFinalizer_decompressStream"));
    StreamState* s = ss_get(stream, 0, 0);
    if (s && g_is_src && !s->cpuMode && !s->paramCpu && !s->started) {
        jlong const sp = (*env)->GetLongField(env, obj, g_is_src), dp = (*env)->GetLongField(env, obj, g_is_dst);
        if (sp >= 0 && sp < src_size && dp >= 0 && dp <= dst_size && src_size <= (*env)->GetArrayLength(env, src) && dst_size <= (*env)->GetArrayLength(env, dst)) {
            size_t const avail = (size_t)(src_size - sp), room = (size_t)(dst_size - dp);
            unsigned char* const in = (unsigned char*)malloc(avail);
            if (in) {
                unsigned long long content = 0, bound = 0; size_t ext;
                (*env)->GetByteArrayRegion(env, src, (jsize)sp, (jsize)avail, (jbyte*)in);
                ext = zjni_frame_extent(in, avail, &content, &bound);
                if (ext && (bound <= room || bound <= (64ull << 20))) {
                    unsigned char* const to = (unsigned char*)malloc((size_t)bound + 1);
                    size_t const r = to ? zjni_decompress(to, (size_t)bound, in, ext) : (size_t)E_MEM;
                    if (!zjni_isError(r) && r <= room) {
                        if (r) (*env)->SetByteArrayRegion(env, dst, (jsize)dp, (jsize)r, (const jbyte*)to);
                        free(to); free(in);
                        (*env)->SetLongField(env, obj, g_is_src, sp + (jlong)ext); (*env)->SetLongField(env, obj, g_is_dst, dp + (jlong)r);
                        __atomic_fetch_add(&g_stats[0], 1, __ATOMIC_RELAXED);
                        return 0;
                    }
                    free(to);
                    if (zjni_isError(r) && zjni_getErrorCode(r) < 200 && zjni_getErrorCode(r) != 70) { free(in); return (jint)r; }     /* the frame is damaged: libzstd's code */
                }
                free(in);
            }
        }
    }
    if (!f) {                                                                       /* no bundled stream: ds_buffered */
        jlong const sp = g_is_src ? (*env)->GetLongField(env, obj, g_is_src) : -1, dp = g_is_dst ? (*env)->GetLongField(env, obj, g_is_dst) : -1;
        size_t produced, consumed, r, avail, room; char* in; char* out;
        if (!s || !gpu_on() || sp < 0 || dp < 0) return -(jint)ZJNI_ERROR_unsupported;
        if (src_size > (*env)->GetArrayLength(env, src) || sp > src_size) return (jint)E_SRC;
        if (dst_size > (*env)->GetArrayLength(env, dst) || dp > dst_size) return (jint)E_DST;
        avail = (size_t)(src_size - sp); room = (size_t)(dst_size - dp);
        in = (char*)malloc(avail + 1); out = (char*)malloc(room + 1);
        if (!in || !out) { free(in); free(out); return (jint)E_MEM; }
        if (avail) (*env)->GetByteArrayRegion(env, src, (jsize)sp, (jsize)avail, (jbyte*)in);
        r = ds_buffered(s, out, room, in, avail, &produced, &consumed);
        if (produced) (*env)->SetByteArrayRegion(env, dst, (jsize)dp, (jsize)produced, (const jbyte*)out);
        free(in); free(out);
        (*env)->SetLongField(env, obj, g_is_src, sp + (jlong)consumed); (*env)->SetLongField(env, obj, g_is_dst, dp + (jlong)produced);
        return zjni_isError(r) ? (jint)r : (r > 0x7FFFFFFFu ? 0x7FFFFFFF : (jint)r);
    }
    {   jint const r = f(env, obj, stream, dst, dst_size, src, src_size);
        if (s) s->started = (r > 0);
        return r; }
}
/* ---- ZstdBufferDecompressingStreamNoFinalizer.decompressStreamNative (N/jni_bufferdecompress_zstd.c:58-87): the heap-ByteBuffer twin of the DirectByteBuffer
 * decompressing stream — byte[] + offsets in, consumed / produced int fields out; same policy: a complete frame at a frame boundary that fits the target goes to
 * the batch decoder, anything else is the bundled library's stream.  The reference's argument checks in its order. */
static jfieldID g_bs_consumed, g_bs_produced;
typedef jlong (*bs_fn)(JNIEnv*, jobject, jlong, jbyteArray, jint, jint, jbyteArray, jint, jint);
JNIEXPORT jlong JNICALL P(ZstdBufferDecompressingStreamNoFinalizer_createDStreamNative)(JNIEnv* env, jclass cls) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(PS("ZstdBufferDecompressingStreamNoFinalizer_createDStreamNative"));
    jlong const h = f ? f(env, cls) : (jlong)(intptr_t)calloc(1, 16);
    if (h) (void)ss_get(h, 1, 0);
    return h;
}
JNIEXPORT jlong JNICALL P(ZstdBufferDecompressingStreamNoFinalizer_freeDStreamNative)(JNIEnv* env, jclass cls, jlong stream) {
    jlong (*f)(JNIEnv*, jclass, jlong) = (jlong (*)(JNIEnv*, jclass, jlong))cpu_sym(PS("ZstdBufferDecompressingStreamNoFinalizer_freeDStreamNative"));
    ss_free(ss_get(stream, 0, 1));
    if (!stream) return 0;
    if (f) return f(env, cls, stream);
    free((void*)(intptr_t)stream); return 0;
}
JNIEXPORT jlong JNICALL P(ZstdBufferDecompressingStreamNoFinalizer_recommendedDOutSizeNative)(JNIEnv* env, jclass cls) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(PS("ZstdBufferDecompressingStreamNoFinalizer_recommendedDOutSizeNative"));
    return f ? f(env, cls) : (jlong)(128u << 10);
}
JNIEXPORT jlong JNICALL P(ZstdBufferDecompressingStreamNoFinalizer_initDStreamNative)(JNIEnv* env, jobject obj, jlong stream) {
    jlong (*f)(JNIEnv*, jobject, jlong) = (jlong (*)(JNIEnv*, jobject, jlong))cpu_sym(PS("ZstdBufferDecompressingStreamNoFinalizer_initDStreamNative"));
    StreamState* s = ss_get(stream, 1, 0);
    jclass const clazz = (*env)->GetObjectClass(env, obj);
    g_bs_consumed = (*env)->GetFieldID(env, clazz, "consumed", "I"); g_bs_produced = (*env)->GetFieldID(env, clazz, "produced", "I");
    if (s) { int const pc = s->paramCpu; ss_reset(s, 3); s->paramCpu = pc; s->cpuMode = (pc || !streams_on_gpu()) ? 1 : 0; }
    return f ? f(env, obj, stream) : 0;
}
JNIEXPORT jlong JNICALL P(ZstdBufferDecompressingStreamNoFinalizer_decompressStreamNative)(JNIEnv* env, jobject obj, jlong stream, jbyteArray dst, jint dst_offset, jint dst_size, jbyteArray src, jint src_offset, jint src_size) {
    bs_fn f = (bs_fn)cpu_sym(PS("ZstdBufferDecompressingStreamNoFinalizer_decompressStreamNative"));
    StreamState* s = ss_get(stream, 0, 0);
    if (s && g_bs_consumed && !s->cpuMode && !s->paramCpu && !s->started && src_size > 0) {
        if (NULL == dst) return E_DST;
        if (NULL == src) return E_SRC;
        if (0 > dst_offset) return E_DST;
        if (0 > src_offset) return E_SRC;
        if (0 > dst_size) return E_DST;
        if (src_offset + src_size > (*env)->GetArrayLength(env, src)) return E_SRC;
        if (dst_offset + dst_size > (*env)->GetArrayLength(env, dst)) return E_DST;
        {   unsigned char* const in = (unsigned char*)malloc((size_t)src_size);
            if (in) {
                unsigned long long content = 0, bound = 0; size_t ext;
                (*env)->GetByteArrayRegion(env, src, src_offset, src_size, (jbyte*)in);
                ext = zjni_frame_extent(in, (size_t)src_size, &content, &bound);
                if (ext && (bound <= (unsigned long long)dst_size || bound <= (64ull << 20))) {
                    unsigned char* const to = (unsigned char*)malloc((size_t)bound + 1);
                    size_t const r = to ? zjni_decompress(to, (size_t)bound, in, ext) : (size_t)E_MEM;
                    if (!zjni_isError(r) && r <= (size_t)dst_size) {
                        if (r) (*env)->SetByteArrayRegion(env, dst, dst_offset, (jsize)r, (const jbyte*)to);
                        free(to); free(in);
                        (*env)->SetIntField(env, obj, g_bs_consumed, (jint)ext); (*env)->SetIntField(env, obj, g_bs_produced, (jint)r);
                        __atomic_fetch_add(&g_stats[0], 1, __ATOMIC_RELAXED);
                        return 0;
                    }
                    free(to);
                    if (zjni_isError(r) && zjni_getErrorCode(r) < 200 && zjni_getErrorCode(r) != 70) { free(in); return (jlong)r; }
                }
                free(in);
            }
        }
    }
    if (!f) {                                                                       /* no bundled stream: ds_buffered */
        size_t produced, consumed, r; char* in; char* out;
        if (!s || !gpu_on() || !g_bs_consumed) return -(jlong)ZJNI_ERROR_unsupported;
        if (NULL == dst) return E_DST;
        if (NULL == src) return E_SRC;
        if (0 > dst_offset || 0 > dst_size || dst_offset + dst_size > (*env)->GetArrayLength(env, dst)) return E_DST;
        if (0 > src_offset || 0 > src_size || src_offset + src_size > (*env)->GetArrayLength(env, src)) return E_SRC;
        in = (char*)malloc((size_t)src_size + 1); out = (char*)malloc((size_t)dst_size + 1);
        if (!in || !out) { free(in); free(out); return E_MEM; }
        if (src_size) (*env)->GetByteArrayRegion(env, src, src_offset, src_size, (jbyte*)in);
        r = ds_buffered(s, out, (size_t)dst_size, in, (size_t)src_size, &produced, &consumed);
        if (produced) (*env)->SetByteArrayRegion(env, dst, dst_offset, (jsize)produced, (const jbyte*)out);
        free(in); free(out);
        (*env)->SetIntField(env, obj, g_bs_consumed, (jint)consumed); (*env)->SetIntField(env, obj, g_bs_produced, (jint)produced);
        return (jlong)r;
    }
    {   jlong const r = f(env, obj, stream, dst, dst_offset, dst_size, src, src_offset, src_size);
        if (s) s->started = (r > 0);
        return r; }
}

/* ==== the context streams: ZstdCompressCtx.compress*Stream0 (N/jni_fast_zstd.c:392-579), setPledgedSrcSize0 (:383-390), getFrameProgression0 (:370-381) and
 * ZstdDecompressCtx.decompressDirectByteBufferStream0 (:720-769) =======================================================================================
 * ZSTD_compressStream2(cctx, out, in, endOp) on the context, in the four heap / direct combinations.  The route is the stream classes': a frame is buffered
 * while it is written (ZSTD_e_continue takes the bytes in), ZSTD_e_flush hands out the frame's next part up to here, ZSTD_e_end the rest — one
 * zjni_compress_stream call each, byte for byte what the reference's context writes over the same directives.  Two cases are one-shot frames and go where
 * compress*0 goes: a frame whose FIRST call is ZSTD_e_end (libzstd takes the input's size as pledged: the frame ZSTD_compress2 makes, content size and
 * all — N/compress/zstd_compress.c:6366, :6450-6560), whatever size was pledged.  A pledged size with more than one call, a dictionary, explicit
 * table sizes, levels above 3, a frame that outgrows the level's window or a GPU that declines: the bundled library's context takes the frame — from its
 * first byte (what was buffered is replayed into it through ITS native, output already handed out is skipped) — and keeps it until it ends.  The result word
 * is the reference's: error -> 1 << 31 | code; else dstPos << 32 | srcPos, bit 63 when nothing is left to flush (:423-433).
 * What differs from the reference, as for the stream classes: WHEN bytes come out (nothing before a flush or the end), never which bytes. */
#define CX_ERR(code) ((jlong)((1ULL << 31) | (unsigned)(code)))
static jlong cx_word(int done, size_t dpos, size_t spos) { uint64_t r = ((uint64_t)(uint32_t)dpos << 32) | (uint32_t)spos; if (done) r |= 1ULL << 63; return (jlong)r; }
typedef jlong (*cx_dd_fn)(JNIEnv*, jclass, jlong, jobject, jint, jint, jobject, jint, jint, jint);
static void cx_drop(CtxState* c) { if (c->cx) { ss_free(c->cx); c->cx = NULL; } }
static StreamState* cx_session(CtxState* c) {
    if (!c->cx) { c->cx = (StreamState*)calloc(1, sizeof(StreamState)); if (c->cx) ss_reset(c->cx, 3); }
    return c->cx;
}
/* one call of the bundled library's direct-buffer stream native over raw memory: 0 and (*produced, *consumed, *done) or *err; -1: no such library */
static int cx_cpu_call(JNIEnv* env, jclass cls, jlong ptr, char* dst, size_t room, const char* src, size_t n, int op, size_t* produced, size_t* consumed, int* done, unsigned* err) {
    cx_dd_fn f = (cx_dd_fn)cpu_sym(PS("ZstdCompressCtx_compressDirectByteBufferStream0"));
    static char nothing[8];
    jobject dbuf, sbuf; uint64_t r;
    if (!f || !(*env)->NewDirectByteBuffer) return -1;
    dbuf = (*env)->NewDirectByteBuffer(env, room ? dst : nothing, (jlong)room); sbuf = (*env)->NewDirectByteBuffer(env, n ? (void*)src : (void*)nothing, (jlong)n);
    if (!dbuf || !sbuf) return -1;
    r = (uint64_t)f(env, cls, ptr, dbuf, 0, (jint)room, sbuf, 0, (jint)n, op);
    if ((*env)->DeleteLocalRef) { (*env)->DeleteLocalRef(env, dbuf); (*env)->DeleteLocalRef(env, sbuf); }
    *err = 0; *produced = *consumed = 0; *done = 0;
    if (r & 0x80000000u) { *err = (unsigned)(r & 0x7FFFFFFFu); return 0; }
    *produced = (size_t)((r >> 32) & 0x7FFFFFFFu); *consumed = (size_t)(r & 0x7FFFFFFFu); *done = (int)(r >> 63);
    return 0;
}
/* everything buffered goes to the bundled library's context (flushes where the caller flushed); what it writes is collected, and the part the caller already
 * has is skipped.  Afterwards the frame is the bundled library's (cpuMode). */
static int cx_replay_to_cpu(JNIEnv* env, jclass cls, jlong ptr, StreamState* s) {
    size_t at = 0, fi = 0; size_t const scratchCap = 256u << 10;
    size_t const delivered = s->emitted - (s->outLen - s->outPos);
    char* scratch = (char*)malloc(scratchCap);
    if (!scratch) return -1;
    scratch[0] = 0;
    s->outLen = s->outPos = 0;
    if (s->total == 0) {                                           /* opened by an empty directive: the bundled context opens its frame the same way (no pledged size) */
        size_t produced, consumed; int done; unsigned err;
        if (cx_cpu_call(env, cls, ptr, scratch, scratchCap, scratch, 0, 0, &produced, &consumed, &done, &err) != 0 || err) { free(scratch); return -1; }
    }
    while (at < s->total || fi < s->nFlush) {
        size_t const upto = fi < s->nFlush ? s->flushAt[fi] : s->total;
        int const flushing = at >= upto;                           /* the bytes up to the flush are in: now the directive itself, until it is done */
        size_t produced, consumed; int done; unsigned err;
        if (cx_cpu_call(env, cls, ptr, scratch, scratchCap, (const char*)s->buf + at, flushing ? 0 : upto - at, flushing ? 1 : 0, &produced, &consumed, &done, &err) != 0 || err
            || !ss_out_room(s, produced)) { free(scratch); return -1; }
        memcpy(s->out + s->outLen, scratch, produced); s->outLen += produced;
        at += consumed;
        if (flushing) { if (done) fi++; }
        else if (!consumed && !produced) { free(scratch); return -1; }
    }
    free(scratch);
    if (s->outLen < s->emitted || ss_hash(SS_HASH0, s->out, s->emitted) != s->madeHash) return -1;      /* (see ss_hash: the bundled stream must have made the caller's prefix again) */
    s->outPos = delivered; s->emitted = 0; s->madeHash = SS_HASH0; s->cpuMode = 1; s->total = 0; s->nFlush = 0;
    ss_forward_level(env, s);
    __atomic_fetch_add(&g_stats[2], 1, __ATOMIC_RELAXED);
    return 0;
}
static int cx_plain(const CtxState* c) {                            /* what zjni_compress_stream can express: a level and the checksum flag */
    int const level = c->level == 0 ? 3 : c->level;
    return !c->cpuOnly && !c->cpuDict && !c->cdict && !c->localCdict && !c->rawDict && !(c->hashLog | c->chainLog) && level >= 1 && level <= 3;
}
/* One directive over raw memory: dst has `room` bytes free, src holds `n` unread bytes.  1: answered (*produced, *consumed, *done, or *err = a libzstd code);
 * 0: this frame is not for the GPU route and nothing has been touched — the caller passes the call on as it came. */
static int cx_stream(JNIEnv* env, jclass cls, jlong ptr, CtxState* c, char* dst, size_t room, const char* src, size_t n, int op,
                     size_t* produced, size_t* consumed, int* done, unsigned* err) {
    StreamState* s = c ? c->cx : NULL;
    int const fresh = !s || (!s->started && !s->cpuMode && s->total == 0 && s->outPos == s->outLen);
    *produced = *consumed = 0; *done = 0; *err = 0;
    if (!c || op < 0 || op > 2 || c->cpuInFrame) return 0;
    if (fresh) {
        if (!streams_on_gpu()) return 0;
        if (op == 2) {                                              /* the whole frame in one directive: a one-shot frame */
            size_t cap, r; unsigned char* tmp;
            if (n > 0x7FFFFFFFu || !gpu_takes_when(c, (jint)n, 1)) return 0;
            /* libzstd compresses such a frame in one piece only when the target can take its bound (N/compress/zstd_compress.c:6145-6152); into a smaller target it
             * goes through the stream's 128 KiB buffer, where a pre-split block is followed by the REST of its piece instead of a full block — other bytes for inputs
             * above one block.  That form (the stream's pieces with the known size's parameters and header) is not on the GPU route. */
            if (n > ZJNI_BLOCKSIZE_MAX && room < zjni_compressBound(n)) return 0;
            /* (a pledged size does not matter here: ending at the first directive, libzstd replaces it by the input's size — N/compress/zstd_compress.c:6366) */
            if (!(s = cx_session(c))) { *err = 64; return 1; }
            cap = zjni_compressBound(n) + 64;
            tmp = (unsigned char*)malloc(cap); if (!tmp) { *err = 64; return 1; }
            r = gpu_compress(c, tmp, cap, src, n);
            if (!gpu_compress_final(c, r, cpu_sym(PS("ZstdCompressCtx_compressDirectByteBufferStream0")) != NULL)) { free(tmp); return 0; }
            if (zjni_isError(r)) { free(tmp); if (zjni_getErrorCode(r) >= 200) return 0; *err = (unsigned)zjni_getErrorCode(r); return 1; }
            ss_reset(s, c->level);
            if (!ss_out_room(s, r)) { free(tmp); *err = 64; return 1; }
            memcpy(s->out, tmp, r); s->outLen = r; s->emitted = r; s->madeHash = ss_hash(SS_HASH0, tmp, r); s->finished = 1; s->started = 1;
            free(tmp);
            *consumed = n;
        } else {
            if (c->hasPledged || !cx_plain(c)) return 0;
            if (!(s = cx_session(c))) { *err = 64; return 1; }
            ss_reset(s, c->level == 0 ? 3 : c->level); s->checksum = c->checksum;
        }
    }
    if (s->cpuMode) {                                               /* replayed earlier: what the bundled context wrote then goes out first, then the call is its own */
        size_t const k = ss_deliver(s, dst, room); size_t p2 = 0;
        *produced = k;
        if (s->outPos < s->outLen) return 1;
        if (cx_cpu_call(env, cls, ptr, dst + k, room - k, src, n, op, &p2, consumed, done, err) != 0) { *err = ZJNI_ERROR_unsupported; return 1; }
        *produced = k + p2;
        if (*err || (op == 2 && *done)) { ss_reset(s, 3); c->hasPledged = 0; }
        return 1;
    }
    if (!s->finished) {
        /* the same one-piece path inside a stream (:6145-6152: the end directive brings ALL the frame's bytes — nothing is buffered, an empty directive opened the
         * frame — and the target can take their bound): blocks as ZSTD_compress2 cuts them, under the stream's header; above one block not the stream route's pieces */
        if (op == 2 && n > ZJNI_BLOCKSIZE_MAX && s->total == 0 && room >= zjni_compressBound(n)) {
            if (cx_replay_to_cpu(env, cls, ptr, s) != 0) { *err = ZJNI_ERROR_unsupported; return 1; }
            return cx_stream(env, cls, ptr, c, dst, room, src, n, op, produced, consumed, done, err);
        }
        if (s->total + n > ss_window(s->level)) {                   /* outgrows the window */
            if (cx_replay_to_cpu(env, cls, ptr, s) != 0) { *err = ZJNI_ERROR_unsupported; return 1; }
            return cx_stream(env, cls, ptr, c, dst, room, src, n, op, produced, consumed, done, err);
        }
        if (s->total + n > s->cap) {
            size_t cc = (s->total + n) * 2 + 65536; unsigned char* p;
            if (cc > ss_window(s->level)) cc = ss_window(s->level);
            p = (unsigned char*)realloc(s->buf, cc); if (!p) { *err = 64; return 1; }
            s->buf = p; s->cap = cc;
        }
        if (n) memcpy(s->buf + s->total, src, n);
        s->total += n; *consumed = n; s->started = 1;
    }
    for (;;) {
        *produced += ss_deliver(s, dst + *produced, room - *produced);
        if (s->outPos < s->outLen) break;                           /* the caller comes back with room */
        if (s->finished || op == 0 || (op == 1 && s->total <= (s->nFlush ? s->flushAt[s->nFlush - 1] : 0))) { *done = 1; break; }   /* (a flush with nothing new writes nothing) */
        /* flush / end with nothing pending: the frame's next part */
        if (op == 1) {
            if (s->nFlush == s->flushCap) { size_t const fc = s->flushCap * 2 + 16; uint32_t* p = (uint32_t*)realloc(s->flushAt, fc * sizeof *p); if (!p) { *err = 64; return 1; } s->flushAt = p; s->flushCap = fc; }
            s->flushAt[s->nFlush++] = (uint32_t)s->total;
        }
        {   size_t const cap = s->total + (s->total >> 8) + 4096 + 64 * (s->nFlush + 4);
            unsigned char* tmp = (unsigned char*)malloc(cap); size_t r;
            if (!tmp) { *err = 64; return 1; }
            r = zjni_compress_stream(tmp, cap, s->buf, s->total, s->level, s->checksum, s->flushAt, s->nFlush, op == 2, 0);
            if (zjni_isError(r)) {
                free(tmp);
                if (zjni_getErrorCode(r) < 200) { *err = (unsigned)zjni_getErrorCode(r); return 1; }
                if (op == 1) s->nFlush--;                           /* the directive is repeated below, on the bundled context */
                if (cx_replay_to_cpu(env, cls, ptr, s) != 0) { *err = ZJNI_ERROR_unsupported; return 1; }
                {   size_t p2 = 0, c2 = 0; size_t const had = *produced;
                    int const ok = cx_stream(env, cls, ptr, c, dst + had, room - had, src, 0, op, &p2, &c2, done, err);
                    *produced = had + p2; return ok; }
            }
            if (r < s->emitted || !ss_out_room(s, r - s->emitted)) { free(tmp); *err = 64; return 1; }
            memcpy(s->out + s->outLen, tmp + s->emitted, r - s->emitted); s->madeHash = ss_hash(s->madeHash, tmp + s->emitted, r - s->emitted); s->outLen += r - s->emitted; s->emitted = r;
            free(tmp);
            if (op == 2) { s->finished = 1; __atomic_fetch_add(&g_stats[0], 1, __ATOMIC_RELAXED); }
        }
    }
    if (s->finished && *done) { ss_reset(s, 3); c->hasPledged = 0; }   /* the frame is out: the next directive starts a new one */
    return 1;
}
static jlong cx_passed_on(CtxState* c, int op, jlong r) {          /* a call the bundled context answered: is it inside a frame now? */
    t_gpu_declined = 0;
    if (c) { int const ended = ((uint64_t)r & 0x80000000u) || (op == 2 && ((uint64_t)r >> 63)); c->cpuInFrame = !ended; if (ended) c->hasPledged = 0; }
    return r;
}
static jlong cx_answer(size_t dpos, size_t spos, size_t produced, size_t consumed, int done, unsigned err) {
    return err ? CX_ERR(err) : cx_word(done, dpos + produced, spos + consumed);
}
static int cx_array_ok(JNIEnv* env, jbyteArray a, jint array_offset, jint size) {      /* is_valid_array_stream_buffer (N/jni_fast_zstd.c:399-404) */
    jsize cap;
    if (0 > array_offset || 0 > size) return 0;
    cap = (*env)->GetArrayLength(env, a);
    return array_offset <= cap && size <= cap - array_offset;
}
/* ZSTD_CCtx_setPledgedSrcSize applies to the NEXT frame.  Where the GPU route may take that frame the pledge is kept here and handed to the bundled context only
 * when the frame is (cx_hand_pledge, before the first call passed on): a pledge given to the bundled context for a frame the GPU route then made would wait there
 * for a later frame. */
JNIEXPORT jlong JNICALL P(ZstdCompressCtx_setPledgedSrcSize0)(JNIEnv* env, jclass cls, jlong ptr, jlong src_size) {
    jlong (*f)(JNIEnv*, jclass, jlong, jlong) = (jlong (*)(JNIEnv*, jclass, jlong, jlong))cpu_sym(PS("ZstdCompressCtx_setPledgedSrcSize0"));
    CtxState* c = st_get(ptr, 'C');
    if (src_size < 0) return E_SRC;
    if (c && streams_on_gpu()) {
        if (c->cpuInFrame) return f ? f(env, cls, ptr, src_size) : (jlong)-60;
        if (c->cx && (c->cx->started || c->cx->cpuMode) && !c->cx->finished) return (jlong)-60;      /* inside a frame: ZSTD_error_stage_wrong */
        c->hasPledged = 1; c->pledged = (unsigned long long)src_size;
        return 0;
    }
    return f ? f(env, cls, ptr, src_size) : 0;
}
static void cx_hand_pledge(JNIEnv* env, jclass cls, jlong ptr, CtxState* c) {          /* a frame is about to start on the bundled context: its pledge goes with it */
    if (c && c->hasPledged && !c->cpuInFrame) {
        jlong (*f)(JNIEnv*, jclass, jlong, jlong) = (jlong (*)(JNIEnv*, jclass, jlong, jlong))cpu_sym(PS("ZstdCompressCtx_setPledgedSrcSize0"));
        if (f) (void)f(env, cls, ptr, (jlong)c->pledged);
        c->hasPledged = 0;
    }
}
/* ZSTD_getFrameProgression: while a frame is buffered here everything taken in is "ingested", nothing "consumed" yet; flushed = handed out */
JNIEXPORT jobject JNICALL P(ZstdCompressCtx_getFrameProgression0)(JNIEnv* env, jclass cls, jlong ptr) {
    jobject (*f)(JNIEnv*, jclass, jlong) = (jobject (*)(JNIEnv*, jclass, jlong))cpu_sym(PS("ZstdCompressCtx_getFrameProgression0"));
    CtxState* c = st_get(ptr, 'C'); StreamState* s = c ? c->cx : NULL;
    if (f && !(s && !s->cpuMode && (s->started || s->total))) return f(env, cls, ptr);
    {   jclass const k = (*env)->FindClass(env, "com/github/luben/zstd/ZstdFrameProgression");
        jmethodID const m = k ? (*env)->GetMethodID(env, k, "<init>", "(JJJJII)V") : NULL;
        jlong const in = s ? (jlong)s->total : 0, made = s ? (jlong)s->emitted : 0, out = s ? (jlong)(s->emitted - (s->outLen - s->outPos)) : 0;
        return m ? (*env)->NewObject(env, k, m, in, made ? in : (jlong)0, made, out, (jint)0, (jint)0) : NULL;
    }
}
JNIEXPORT jlong JNICALL P(ZstdCompressCtx_compressDirectByteBufferStream0)
  (JNIEnv* env, jclass cls, jlong ptr, jobject dst, jint dst_offset, jint dst_size, jobject src, jint src_offset, jint src_size, jint end_op) {
    cx_dd_fn f = (cx_dd_fn)cpu_sym(PS("ZstdCompressCtx_compressDirectByteBufferStream0"));
    CtxState* c = st_get(ptr, 'C');
    char* d; char* sb; size_t produced, consumed; int done; unsigned err;
    if (NULL == dst) return CX_ERR(70);
    if (NULL == src) return CX_ERR(72);
    if (0 > dst_offset || dst_offset > dst_size) return CX_ERR(70);
    if (0 > src_offset || src_offset > src_size) return CX_ERR(72);
    if (dst_size > (*env)->GetDirectBufferCapacity(env, dst)) return CX_ERR(70);
    if (src_size > (*env)->GetDirectBufferCapacity(env, src)) return CX_ERR(72);
    d = (char*)(*env)->GetDirectBufferAddress(env, dst); sb = (char*)(*env)->GetDirectBufferAddress(env, src);
    if (d == NULL || sb == NULL) return CX_ERR(64);
    if (cx_stream(env, cls, ptr, c, d + dst_offset, (size_t)(dst_size - dst_offset), sb + src_offset, (size_t)(src_size - src_offset), end_op, &produced, &consumed, &done, &err))
        return cx_answer((size_t)dst_offset, (size_t)src_offset, produced, consumed, done, err);
    return f ? (cx_hand_pledge(env, cls, ptr, c), cx_passed_on(c, end_op, f(env, cls, ptr, dst, dst_offset, dst_size, src, src_offset, src_size, end_op))) : CX_ERR(ZJNI_ERROR_unsupported);
}
JNIEXPORT jlong JNICALL P(ZstdCompressCtx_compressByteArrayToDirectByteBufferStream0)
  (JNIEnv* env, jclass cls, jlong ptr, jobject dst, jint dst_offset, jint dst_size, jbyteArray src, jint src_array_offset, jint src_offset, jint src_size, jint end_op) {
    jlong (*f)(JNIEnv*, jclass, jlong, jobject, jint, jint, jbyteArray, jint, jint, jint, jint) =
        (jlong (*)(JNIEnv*, jclass, jlong, jobject, jint, jint, jbyteArray, jint, jint, jint, jint))cpu_sym(PS("ZstdCompressCtx_compressByteArrayToDirectByteBufferStream0"));
    CtxState* c = st_get(ptr, 'C');
    char* d; char* in; size_t produced, consumed, n; int done, mine; unsigned err;
    if (NULL == dst) return CX_ERR(70);
    if (NULL == src) return CX_ERR(72);
    if (0 > dst_offset || dst_offset > dst_size) return CX_ERR(70);
    if (0 > src_offset || src_offset > src_size) return CX_ERR(72);
    if (dst_size > (*env)->GetDirectBufferCapacity(env, dst)) return CX_ERR(70);
    if (!cx_array_ok(env, src, src_array_offset, src_size)) return CX_ERR(72);
    d = (char*)(*env)->GetDirectBufferAddress(env, dst);
    if (d == NULL) return CX_ERR(64);
    n = (size_t)(src_size - src_offset);
    in = (char*)malloc(n + 1); if (!in) return CX_ERR(64);
    if (n) (*env)->GetByteArrayRegion(env, src, src_array_offset + src_offset, (jsize)n, (jbyte*)in);
    mine = cx_stream(env, cls, ptr, c, d + dst_offset, (size_t)(dst_size - dst_offset), in, n, end_op, &produced, &consumed, &done, &err);
    free(in);
    if (mine) return cx_answer((size_t)dst_offset, (size_t)src_offset, produced, consumed, done, err);
    return f ? (cx_hand_pledge(env, cls, ptr, c), cx_passed_on(c, end_op, f(env, cls, ptr, dst, dst_offset, dst_size, src, src_array_offset, src_offset, src_size, end_op))) : CX_ERR(ZJNI_ERROR_unsupported);
}
JNIEXPORT jlong JNICALL P(ZstdCompressCtx_compressDirectByteBufferToByteArrayStream0)
  (JNIEnv* env, jclass cls, jlong ptr, jbyteArray dst, jint dst_array_offset, jint dst_offset, jint dst_size, jobject src, jint src_offset, jint src_size, jint end_op) {
    jlong (*f)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint, jint, jobject, jint, jint, jint) =
        (jlong (*)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint, jint, jobject, jint, jint, jint))cpu_sym(PS("ZstdCompressCtx_compressDirectByteBufferToByteArrayStream0"));
    CtxState* c = st_get(ptr, 'C');
    char* sb; char* out; size_t produced, consumed, room; int done, mine; unsigned err;
    if (NULL == dst) return CX_ERR(70);
    if (NULL == src) return CX_ERR(72);
    if (0 > dst_offset || dst_offset > dst_size) return CX_ERR(70);
    if (0 > src_offset || src_offset > src_size) return CX_ERR(72);
    if (!cx_array_ok(env, dst, dst_array_offset, dst_size)) return CX_ERR(70);
    if (src_size > (*env)->GetDirectBufferCapacity(env, src)) return CX_ERR(72);
    sb = (char*)(*env)->GetDirectBufferAddress(env, src);
    if (sb == NULL) return CX_ERR(64);
    room = (size_t)(dst_size - dst_offset);
    out = (char*)malloc(room + 1); if (!out) return CX_ERR(64);
    mine = cx_stream(env, cls, ptr, c, out, room, sb + src_offset, (size_t)(src_size - src_offset), end_op, &produced, &consumed, &done, &err);
    if (mine && produced) (*env)->SetByteArrayRegion(env, dst, dst_array_offset + dst_offset, (jsize)produced, (const jbyte*)out);
    free(out);
    if (mine) return cx_answer((size_t)dst_offset, (size_t)src_offset, produced, consumed, done, err);
    return f ? (cx_hand_pledge(env, cls, ptr, c), cx_passed_on(c, end_op, f(env, cls, ptr, dst, dst_array_offset, dst_offset, dst_size, src, src_offset, src_size, end_op))) : CX_ERR(ZJNI_ERROR_unsupported);
}
JNIEXPORT jlong JNICALL P(ZstdCompressCtx_compressByteArrayStream0)
  (JNIEnv* env, jclass cls, jlong ptr, jbyteArray dst, jint dst_array_offset, jint dst_offset, jint dst_size, jbyteArray src, jint src_array_offset, jint src_offset, jint src_size, jint end_op) {
    jlong (*f)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint, jint, jbyteArray, jint, jint, jint, jint) =
        (jlong (*)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint, jint, jbyteArray, jint, jint, jint, jint))cpu_sym(PS("ZstdCompressCtx_compressByteArrayStream0"));
    CtxState* c = st_get(ptr, 'C');
    char* in; char* out; size_t produced, consumed, room, n; int done, mine; unsigned err;
    if (NULL == dst) return CX_ERR(70);
    if (NULL == src) return CX_ERR(72);
    if (0 > dst_offset || dst_offset > dst_size) return CX_ERR(70);
    if (0 > src_offset || src_offset > src_size) return CX_ERR(72);
    if (!cx_array_ok(env, dst, dst_array_offset, dst_size)) return CX_ERR(70);
    if (!cx_array_ok(env, src, src_array_offset, src_size)) return CX_ERR(72);
    room = (size_t)(dst_size - dst_offset); n = (size_t)(src_size - src_offset);
    in = (char*)malloc(n + 1); out = (char*)malloc(room + 1);
    if (!in || !out) { free(in); free(out); return CX_ERR(64); }
    if (n) (*env)->GetByteArrayRegion(env, src, src_array_offset + src_offset, (jsize)n, (jbyte*)in);
    mine = cx_stream(env, cls, ptr, c, out, room, in, n, end_op, &produced, &consumed, &done, &err);
    if (mine && produced) (*env)->SetByteArrayRegion(env, dst, dst_array_offset + dst_offset, (jsize)produced, (const jbyte*)out);
    free(in); free(out);
    if (mine) return cx_answer((size_t)dst_offset, (size_t)src_offset, produced, consumed, done, err);
    return f ? (cx_hand_pledge(env, cls, ptr, c), cx_passed_on(c, end_op, f(env, cls, ptr, dst, dst_array_offset, dst_offset, dst_size, src, src_array_offset, src_offset, src_size, end_op))) : CX_ERR(ZJNI_ERROR_unsupported);
}
/* ZSTD_decompressStream on the context: at a frame boundary a COMPLETE frame in the source with room for all it decodes to goes to the batch decoder in one piece
 * (as ZstdDirectBufferDecompressingStreamNoFinalizer.decompressStreamNative above); anything else is the bundled library's, until that frame ends */
JNIEXPORT jlong JNICALL P(ZstdDecompressCtx_decompressDirectByteBufferStream0)
  (JNIEnv* env, jclass cls, jlong ptr, jobject dst, jint dst_offset, jint dst_size, jobject src, jint src_offset, jint src_size) {
    jlong (*f)(JNIEnv*, jclass, jlong, jobject, jint, jint, jobject, jint, jint) = (jlong (*)(JNIEnv*, jclass, jlong, jobject, jint, jint, jobject, jint, jint))cpu_sym(PS("ZstdDecompressCtx_decompressDirectByteBufferStream0"));
    CtxState* c = st_get(ptr, 'D');
    char* d; char* sb;
    if (NULL == dst) return CX_ERR(70);
    if (NULL == src) return CX_ERR(72);
    if (0 > dst_offset) return CX_ERR(70);
    if (0 > src_offset) return CX_ERR(72);
    if (0 > dst_size) return CX_ERR(70);
    if (0 > src_size) return CX_ERR(72);
    if (dst_size > (*env)->GetDirectBufferCapacity(env, dst)) return CX_ERR(70);
    if (src_size > (*env)->GetDirectBufferCapacity(env, src)) return CX_ERR(72);
    d = (char*)(*env)->GetDirectBufferAddress(env, dst); if (d == NULL) return CX_ERR(64);
    sb = (char*)(*env)->GetDirectBufferAddress(env, src); if (sb == NULL) return CX_ERR(64);
    if (c && !c->dInFrame && !c->cpuOnly && !c->cpuDict && !c->ddict && !c->ddictOwned && streams_on_gpu() && src_offset < src_size && dst_offset <= dst_size) {
        size_t const room = (size_t)(dst_size - dst_offset);
        unsigned long long content = 0, bound = 0;
        size_t const ext = zjni_frame_extent(sb + src_offset, (size_t)(src_size - src_offset), &content, &bound);
        if (ext && (bound <= room || bound <= (64ull << 20))) {
            int const aside = bound > room;
            char* const to = aside ? (char*)malloc((size_t)bound + 1) : d + dst_offset;
            size_t const r = to ? zjni_decompress(to, aside ? (size_t)bound : room, sb + src_offset, ext) : (size_t)E_MEM;
            if (!zjni_isError(r) && r <= room) {
                if (aside) { memcpy(d + dst_offset, to, r); free(to); }
                __atomic_fetch_add(&g_stats[0], 1, __ATOMIC_RELAXED);
                return cx_word(1, (size_t)dst_offset + r, (size_t)src_offset + ext);
            }
            if (aside) free(to);
            if (zjni_isError(r) && zjni_getErrorCode(r) < 200 && zjni_getErrorCode(r) != 70) return CX_ERR(zjni_getErrorCode(r));
        }
    }
    if (!f) {                                                                       /* no bundled context: ds_buffered, its state beside the context's */
        size_t produced, consumed, r; StreamState* ds;
        if (!c || !gpu_on() || c->cpuOnly || c->cpuDict || c->ddict || c->ddictOwned || src_offset > src_size || dst_offset > dst_size || !(ds = cx_session(c))) return CX_ERR(ZJNI_ERROR_unsupported);
        r = ds_buffered(ds, d + dst_offset, (size_t)(dst_size - dst_offset), sb + src_offset, (size_t)(src_size - src_offset), &produced, &consumed);
        c->dInFrame = ds->started;
        if (zjni_isError(r)) return CX_ERR(zjni_getErrorCode(r));
        return cx_word(r == 0, (size_t)dst_offset + produced, (size_t)src_offset + consumed);
    }
    {   uint64_t const r = (uint64_t)f(env, cls, ptr, dst, dst_offset, dst_size, src, src_offset, src_size);
        if (c) c->dInFrame = !(r & 0x80000000u) && !(r >> 63);                       /* bit 63: ZSTD_decompressStream returned 0, a frame has just ended */
        return (jlong)r; }
}

/* ---- frame inspection and constants: host-side arithmetic on a few header bytes, answered here whether or not a bundled library is loaded ------------
 * Zstd.decompressedSize / getFrameContentSize / findFrameCompressedSize / getDictIdFromFrame / getDictIdFromDict in their byte[] and direct-buffer
 * forms (N/jni_zstd.c:30-43, :70-227) and the constants of class Zstd (N/jni_zstd.c:573-667).  The layout parsed is the format's (RFC 8878 section 3.1.1:
 * magic, frame header descriptor, window descriptor, dictionary id, content size; 3.1.2 skippable frames; 3.1.1.2 block headers); the answers for
 * short, foreign and damaged inputs are libzstd's (N/decompress/zstd_decompress.c:447-545 header, :569-585 content size, :587-602 and :734-795 frame
 * extent, :1624-1650 dictionary ids; N/decompress/zstd_decompress_block.c:63-78 block header) and tests/jni/harness.c compares every one of them with the
 * reference library's native of the same name.  Frames of the pre-1.0 formats (magic 0xFD2FB522 .. 0xFD2FB527, which zstd-jni's build still reads) are
 * the bundled library's: without it they are "not a zstd frame", as they are to the GPU path. */
typedef struct { unsigned long long fcs; uint32_t dictID, headerSize; int skippable, checksum; } FrameHead;
static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
#define FH_ERR(code) ((size_t)0 - (size_t)(code))
#define FH_IS_ERR(r) ((r) > (size_t)0 - 120)
/* 0: *h filled; > 0 (not an error): that many bytes are needed; an error code otherwise */
static size_t frame_head(FrameHead* h, const uint8_t* p, size_t n, int magicless) {
    size_t const minIn = magicless ? 1 : 5;
    if (n < minIn) {
        if (n > 0 && !magicless) {                                         /* what is there must at least begin like a magic number */
            uint8_t a[4] = { 0x28, 0xB5, 0x2F, 0xFD }, b[4] = { 0x50, 0x2A, 0x4D, 0x18 };
            size_t const k = n < 4 ? n : 4;
            memcpy(a, p, k); memcpy(b, p, k);
            if (le32(a) != 0xFD2FB528u && (le32(b) & 0xFFFFFFF0u) != 0x184D2A50u) return FH_ERR(10);
        }
        return minIn;
    }
    memset(h, 0, sizeof *h);
    if (!magicless && le32(p) != 0xFD2FB528u) {
        if ((le32(p) & 0xFFFFFFF0u) != 0x184D2A50u) return FH_ERR(10);       /* prefix_unknown */
        if (n < 8) return 8;
        h->skippable = 1; h->dictID = le32(p) - 0x184D2A50u; h->headerSize = 8; h->fcs = le32(p + 4);
        return 0;
    }
    {   uint8_t const fhd = p[minIn - 1];
        unsigned const didCode = fhd & 3, single = (fhd >> 5) & 1, fcsId = fhd >> 6;
        static const uint8_t didSize[4] = { 0, 1, 2, 4 }, fcsSize[4] = { 0, 2, 4, 8 };
        size_t const fhsize = minIn + !single + didSize[didCode] + fcsSize[fcsId] + (single && !fcsId);
        size_t pos = minIn; int i;
        if (n < fhsize) return fhsize;
        h->headerSize = (uint32_t)fhsize;
        if (fhd & 8) return FH_ERR(14);                                    /* frameParameter_unsupported: the reserved bit */
        if (!single) { if ((unsigned)(p[pos++] >> 3) + 10 > 31) return FH_ERR(16); }      /* frameParameter_windowTooLarge */
        for (i = 0; i < didSize[didCode]; i++) h->dictID |= (uint32_t)p[pos + i] << (8 * i);
        pos += didSize[didCode];
        h->fcs = ~0ull;
        if (fcsId == 0) { if (single) h->fcs = p[pos]; }
        else { h->fcs = 0; for (i = 0; i < fcsSize[fcsId]; i++) h->fcs |= (unsigned long long)p[pos + i] << (8 * i); if (fcsId == 1) h->fcs += 256; }
        h->checksum = (fhd >> 2) & 1;
    }
    return 0;
}
static int frame_is_legacy(const uint8_t* p, size_t n) { return n >= 4 && le32(p) >= 0xFD2FB522u && le32(p) <= 0xFD2FB527u; }
/* JNI_ZSTD_decompressedSize (N/jni_zstd.c:32-43) */
static size_t frame_content_size(const uint8_t* p, size_t n, int magicless) {
    FrameHead h;
    if (magicless) return frame_head(&h, p, n, 1) != 0 ? 0 : (size_t)h.fcs;
    if (frame_head(&h, p, n, 0) != 0) return (size_t)-2;                     /* ZSTD_CONTENTSIZE_ERROR */
    return h.skippable ? 0 : (size_t)h.fcs;
}
/* ZSTD_findFrameCompressedSize */
static size_t frame_compressed_size(const uint8_t* p, size_t n) {
    FrameHead h; size_t pos, r;
    if (n >= 8 && (le32(p) & 0xFFFFFFF0u) == 0x184D2A50u) {
        uint32_t const sz = le32(p + 4);
        if ((uint32_t)(sz + 8) < sz) return FH_ERR(14);
        return (size_t)sz + 8 > n ? FH_ERR(72) : (size_t)sz + 8;
    }
    r = frame_head(&h, p, n, 0);
    if (FH_IS_ERR(r)) return r;
    if (r > 0) return FH_ERR(72);                                           /* srcSize_wrong */
    pos = h.headerSize;
    for (;;) {
        uint32_t bh, type; size_t body;
        if (n - pos < 3) return FH_ERR(72);
        bh = (uint32_t)p[pos] | (uint32_t)p[pos + 1] << 8 | (uint32_t)p[pos + 2] << 16;
        type = (bh >> 1) & 3;
        if (type == 3) return FH_ERR(20);                                   /* corruption_detected: the reserved block type */
        body = type == 1 ? 1 : bh >> 3;
        if (3 + body > n - pos) return FH_ERR(72);
        pos += 3 + body;
        if (bh & 1) break;
    }
    if (h.checksum) { if (n - pos < 4) return FH_ERR(72); pos += 4; }
    return pos;
}
static unsigned frame_dict_id(const uint8_t* p, size_t n) { FrameHead h; memset(&h, 0, sizeof h); return FH_IS_ERR(frame_head(&h, p, n, 0)) ? 0 : h.dictID; }
static unsigned dict_dict_id(const uint8_t* p, size_t n) { return (n < 8 || le32(p) != 0xEC30A437u) ? 0 : le32(p + 4); }

/* byte[] forms: the reference reads the array in place and checks nothing (the Java side has: J/Zstd.java), so an absurd (offset, limit) is the caller's */
static uint8_t* array_bytes(JNIEnv* env, jbyteArray a, jint offset, jint n) {      /* a copy of a[offset, offset + n): inspection reads a header or walks block headers, never the payload twice */
    uint8_t* b = (uint8_t*)malloc((size_t)(n > 0 ? n : 0) + 1);
    if (b && n > 0) (*env)->GetByteArrayRegion(env, a, offset, n, (jbyte*)b);
    return b;
}
#define LEGACY_TO_CPU(T_ARGS, CALL_ARGS, name, p_, n_) do { if (frame_is_legacy(p_, n_)) { jlong (*f_) T_ARGS = (jlong (*) T_ARGS)cpu_sym(PS(name)); if (f_) { jlong const r_ = f_ CALL_ARGS; free(own); return r_; } } } while (0)
JNIEXPORT jlong JNICALL P(Zstd_decompressedSize0)(JNIEnv* env, jclass cls, jbyteArray src, jint offset, jint limit, jboolean magicless) {
    jint const n = limit < 18 ? limit : 18;                                 /* a frame header is at most 18 bytes */
    uint8_t* own = array_bytes(env, src, offset, n); size_t r;
    if (!own) return E_MEM;
    LEGACY_TO_CPU((JNIEnv*, jclass, jbyteArray, jint, jint, jboolean), (env, cls, src, offset, limit, magicless), "Zstd_decompressedSize0", own, (size_t)(n > 0 ? n : 0));
    r = frame_content_size(own, (size_t)(n > 0 ? n : 0), magicless == JNI_TRUE); free(own);
    return (jlong)r;                                                        /* (N/jni_zstd.c:77 compares an unsigned size with 0: unknown and error come back as -1 and -2, as from getFrameContentSize0) */
}
JNIEXPORT jlong JNICALL P(Zstd_findFrameCompressedSize0)(JNIEnv* env, jclass cls, jbyteArray src, jint offset, jint limit) {
    uint8_t* own = array_bytes(env, src, offset, limit); size_t r;
    if (!own) return E_MEM;
    LEGACY_TO_CPU((JNIEnv*, jclass, jbyteArray, jint, jint), (env, cls, src, offset, limit), "Zstd_findFrameCompressedSize0", own, (size_t)(limit > 0 ? limit : 0));
    r = frame_compressed_size(own, (size_t)(limit > 0 ? limit : 0)); free(own);
    return (jlong)r;
}
JNIEXPORT jlong JNICALL P(Zstd_getDictIdFromFrame)(JNIEnv* env, jclass cls, jbyteArray src) {
    jsize const len = (*env)->GetArrayLength(env, src); jint const n = len < 18 ? len : 18;
    uint8_t* own = array_bytes(env, src, 0, n); unsigned id;
    (void)cls;
    if (!own) return 0;
    id = frame_dict_id(own, (size_t)n); free(own);
    return (jlong)id;
}
JNIEXPORT jlong JNICALL P(Zstd_getDictIdFromDict)(JNIEnv* env, jclass cls, jbyteArray src) {
    jsize const len = (*env)->GetArrayLength(env, src); jint const n = len < 8 ? len : 8;
    uint8_t* own = array_bytes(env, src, 0, n); unsigned id;
    (void)cls;
    if (!own) return 0;
    id = dict_dict_id(own, (size_t)n); free(own);
    return (jlong)id;
}
/* direct-buffer forms: range checked against the buffer's capacity, -ZSTD_error_GENERIC when outside (N/jni_zstd.c:104-114, :198-227) */
static const uint8_t* direct_range(JNIEnv* env, jobject buf, jint offset, jint size, jlong* err) {
    jlong const cap = (*env)->GetDirectBufferCapacity(env, buf); const uint8_t* p;
    if (offset < 0 || size < 0 || offset > cap - size) { *err = -1; return NULL; }
    p = (const uint8_t*)(*env)->GetDirectBufferAddress(env, buf);
    if (!p) { *err = E_MEM; return NULL; }
    return p + offset;
}
JNIEXPORT jlong JNICALL P(Zstd_findDirectByteBufferFrameCompressedSize)(JNIEnv* env, jclass cls, jobject src, jint offset, jint size) {
    jlong err = 0; const uint8_t* p = direct_range(env, src, offset, size, &err); uint8_t* own = NULL;
    if (!p) return err;
    LEGACY_TO_CPU((JNIEnv*, jclass, jobject, jint, jint), (env, cls, src, offset, size), "Zstd_findDirectByteBufferFrameCompressedSize", p, (size_t)size);
    return (jlong)frame_compressed_size(p, (size_t)size);
}
JNIEXPORT jlong JNICALL P(Zstd_decompressedDirectByteBufferSize)(JNIEnv* env, jclass cls, jobject src, jint offset, jint size, jboolean magicless) {
    jlong err = 0; const uint8_t* p = direct_range(env, src, offset, size, &err); uint8_t* own = NULL;
    if (!p) return err;
    LEGACY_TO_CPU((JNIEnv*, jclass, jobject, jint, jint, jboolean), (env, cls, src, offset, size, magicless), "Zstd_decompressedDirectByteBufferSize", p, (size_t)size);
    return (jlong)frame_content_size(p, (size_t)size, magicless == JNI_TRUE);      /* (as decompressedSize0: the unsigned comparison at N/jni_zstd.c:211 lets -1 and -2 through) */
}
JNIEXPORT jlong JNICALL P(Zstd_getDirectByteBufferFrameContentSize)(JNIEnv* env, jclass cls, jobject src, jint offset, jint size, jboolean magicless) {
    jlong err = 0; const uint8_t* p = direct_range(env, src, offset, size, &err); uint8_t* own = NULL;
    if (!p) return err;
    LEGACY_TO_CPU((JNIEnv*, jclass, jobject, jint, jint, jboolean), (env, cls, src, offset, size, magicless), "Zstd_getDirectByteBufferFrameContentSize", p, (size_t)size);
    return (jlong)frame_content_size(p, (size_t)size, magicless == JNI_TRUE);
}
JNIEXPORT jlong JNICALL P(Zstd_getDictIdFromFrameBuffer)(JNIEnv* env, jclass cls, jobject src) {
    jlong const cap = (*env)->GetDirectBufferCapacity(env, src); const uint8_t* p;
    (void)cls;
    if (cap <= 0) return 0;
    p = (const uint8_t*)(*env)->GetDirectBufferAddress(env, src);
    return p ? (jlong)frame_dict_id(p, (size_t)cap) : 0;
}
JNIEXPORT jlong JNICALL P(Zstd_getDictIdFromDictDirect)(JNIEnv* env, jclass cls, jobject src, jint offset, jint size) {
    const uint8_t* p = (const uint8_t*)(*env)->GetDirectBufferAddress(env, src);
    (void)cls;
    return p ? (jlong)dict_dict_id(p + offset, (size_t)size) : 0;
}
JNIEXPORT jlong JNICALL P(ZstdDirectBufferDecompressingStreamNoFinalizer_recommendedDOutSizeNative)(JNIEnv* env, jclass cls) {
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym(PS("ZstdDirectBufferDecompressingStreamNoFinalizer_recommendedDOutSizeNative"));
    return f ? f(env, cls) : (jlong)(128u << 10);                                    /* ZSTD_DStreamOutSize() */
}
/* the constants of zstd.h as this library's format code was written against them (a 64-bit build: N/zstd.h:133-148, :1263-1275; levels N/compress/clevels.h,
 * ZSTD_minCLevel = -ZSTD_TARGETLENGTH_MAX) and the error enumeration (N/zstd_errors.h:60-96, the values zjni_getErrorCode returns) */
#define CONST_INT(name, v) JNIEXPORT jint JNICALL P(Zstd_##name)(JNIEnv* env, jclass cls) { (void)env; (void)cls; return (jint)(v); }
CONST_INT(windowLogMin, 10) CONST_INT(windowLogMax, 31) CONST_INT(chainLogMin, 6) CONST_INT(chainLogMax, 30) CONST_INT(hashLogMin, 6) CONST_INT(hashLogMax, 30)
CONST_INT(searchLogMin, 1) CONST_INT(searchLogMax, 30) CONST_INT(magicNumber, 0xFD2FB528u) CONST_INT(blockSizeMax, 1 << 17)
CONST_INT(defaultCompressionLevel, 3) CONST_INT(minCompressionLevel, -(1 << 17)) CONST_INT(maxCompressionLevel, 22)
#define CONST_ERR(name, v) JNIEXPORT jlong JNICALL P(Zstd_err##name)(JNIEnv* env, jclass cls) { (void)env; (void)cls; return (jlong)(v); }
CONST_ERR(NoError, 0) CONST_ERR(Generic, 1) CONST_ERR(PrefixUnknown, 10) CONST_ERR(VersionUnsupported, 12) CONST_ERR(FrameParameterUnsupported, 14)
CONST_ERR(FrameParameterWindowTooLarge, 16) CONST_ERR(CorruptionDetected, 20) CONST_ERR(ChecksumWrong, 22) CONST_ERR(DictionaryCorrupted, 30)
CONST_ERR(DictionaryWrong, 32) CONST_ERR(DictionaryCreationFailed, 34) CONST_ERR(ParameterUnsupported, 40) CONST_ERR(ParameterOutOfBound, 42)
CONST_ERR(TableLogTooLarge, 44) CONST_ERR(MaxSymbolValueTooLarge, 46) CONST_ERR(MaxSymbolValueTooSmall, 48) CONST_ERR(StageWrong, 60)
CONST_ERR(InitMissing, 62) CONST_ERR(MemoryAllocation, 64) CONST_ERR(WorkSpaceTooSmall, 66) CONST_ERR(DstSizeTooSmall, 70) CONST_ERR(SrcSizeWrong, 72)
CONST_ERR(DstBufferNull, 74)

/* ---- decompress streams WITHOUT a bundled library behind them: a frame that arrives in pieces, or into a target smaller than its content ---------------------------
 * With the bundled library such frames are its stream's (ZSTD_decompressStream keeps a window, not the frame).  Without it the frame is collected here — exactly
 * the frame's bytes, never a byte of what follows it: the frame header says where the first block header is, every block header where the next one is — decoded in
 * one piece by zjni_decompress when its last byte has arrived, and handed out as the caller brings room.  Answers follow ZSTD_decompressStream: 0 after a frame's last
 * byte is out, else a positive hint; input is not taken while output is pending.  Frames to 256 MiB and 1 GiB of content. */
#define DS_MAX_FRAME ((size_t)256 << 20)
#define DS_MAX_CONTENT ((unsigned long long)1 << 30)
/* how many bytes the collected frame must reach before more can be said (> have), or 0 with *total = the frame's size when [0, have) holds all of it; an error code */
static size_t ds_need(const uint8_t* p, size_t have, size_t* total) {
    FrameHead h; size_t r, pos;
    *total = 0;
    if (have >= 8 && (p[0] & 0xF0) == 0x50 && p[1] == 0x2A && p[2] == 0x4D && p[3] == 0x18) {       /* a skippable frame: 8 bytes and what they announce */
        size_t const t = 8 + (size_t)((uint32_t)p[4] | (uint32_t)p[5] << 8 | (uint32_t)p[6] << 16 | (uint32_t)p[7] << 24);
        if (t > DS_MAX_FRAME) return FH_ERR(ZJNI_ERROR_unsupported);
        if (have < t) return t;
        *total = t; return 0;
    }
    r = frame_head(&h, p, have, 0);
    if (FH_IS_ERR(r)) return r;
    if (r > 0) return r;
    pos = h.headerSize;
    for (;;) {
        uint32_t bh, type; size_t body;
        if (have < pos + 3) return pos + 3;
        bh = (uint32_t)p[pos] | (uint32_t)p[pos + 1] << 8 | (uint32_t)p[pos + 2] << 16;
        type = (bh >> 1) & 3;
        if (type == 3) return FH_ERR(20);
        body = type == 1 ? 1 : bh >> 3;
        pos += 3 + body;
        if (pos > DS_MAX_FRAME) return FH_ERR(ZJNI_ERROR_unsupported);
        if (bh & 1) break;
        if (have < pos) return pos + 3 > have ? pos + 3 : pos;          /* the block and the next block's header */
    }
    if (h.checksum) pos += 4;
    if (have < pos) return pos;
    *total = pos; return 0;
}
/* one call of a decompress stream on raw memory: dst has `room` bytes, src `avail` unread bytes.  Returns ZSTD_decompressStream's answer (0, a hint, or an error code) */
static size_t ds_buffered(StreamState* s, char* dst, size_t room, const char* src, size_t avail, size_t* produced, size_t* consumed) {
    *produced = *consumed = 0;
    if (s->outPos == s->outLen) {
        for (;;) {                                                  /* collect the frame */
            size_t total = 0, take; size_t const need = ds_need(s->buf, s->total, &total);
            if (zjni_isError(need)) { s->total = 0; s->started = 0; return need; }      /* (libzstd's codes and this library's 201) */
            if (need == 0) {                                        /* all of it is here: decode */
                unsigned long long content = 0, bound = 0; size_t r = 0;
                size_t const ext = zjni_frame_extent(s->buf, total, &content, &bound);
                s->outLen = s->outPos = 0;
                if (ext) {
                    if (bound > DS_MAX_CONTENT || !ss_out_room(s, (size_t)bound + 1)) { s->total = 0; s->started = 0; return FH_ERR(ZJNI_ERROR_unsupported); }
                    r = zjni_decompress(s->out, (size_t)bound, s->buf, total);
                    if (zjni_isError(r)) { s->total = 0; s->started = 0; return r; }
                    s->outLen = r;
                    __atomic_fetch_add(&g_stats[0], 1, __ATOMIC_RELAXED);
                } else if (!(total >= 8 && (s->buf[0] & 0xF0) == 0x50)) { s->total = 0; s->started = 0; return FH_ERR(20); }       /* (a skippable frame has nothing to hand out) */
                s->total = 0;
                break;
            }
            take = need - s->total; if (take > avail - *consumed) take = avail - *consumed;
            if (take == 0) { s->started = 1; return need - s->total; }      /* the caller comes back with more */
            if (s->total + take > s->cap) {
                size_t const c = (s->total + take) * 2 + 4096; unsigned char* q = (unsigned char*)realloc(s->buf, c);
                if (!q) return (size_t)E_MEM;
                s->buf = q; s->cap = c;
            }
            memcpy(s->buf + s->total, src + *consumed, take); s->total += take; *consumed += take;
        }
    }
    *produced = ss_deliver(s, dst, room);
    s->started = s->outPos < s->outLen;
    return s->started ? s->outLen - s->outPos : 0;
}
