/* zjni_shim.c — the JNI side of the drop-in: the hot-path natives of zstd-jni, bound to libzjni_amd.so.
 *
 * zstd-jni's Java classes call per-buffer natives (reference src/main/native/jni_fast_zstd.c, jni_zstd.c, "N/").
 * This file defines the SAME Java_com_github_luben_zstd_* symbols for the one-shot hot path and sends the buffers
 * to the GPU library through its C-ABI (include/zjni_amd.h) instead of calling ZSTD_compress2 /
 * ZSTD_decompressDCtx.  Argument checks, their order and the returned error codes are the reference's
 * (N/jni_fast_zstd.c:586-640, :777-905; N/jni_zstd.c:50-63, :230-267), so the Java/Scala layer above cannot
 * tell the difference except by speed.  Two batch natives are added (no Java signature changes elsewhere).
 *
 * What the GPU path does not take (levels > 3, inputs > 128 KiB, no device) is FORWARDED to the bundled CPU
 * library's own native of the same name, looked up with dlsym in the library named by $ZSTD_JNI_CPU_LIB — the CPU
 * code stays where it is, this file contains none.  Without that library such calls return the zjni error code.
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

#define E_DST ((jlong)-70)   /* -ZSTD_error_dstSize_tooSmall */
#define E_SRC ((jlong)-72)   /* -ZSTD_error_srcSize_wrong */
#define E_MEM ((jlong)-64)   /* -ZSTD_error_memory_allocation */

/* ---- the bundled CPU library (optional) ------------------------------------------------------------- */
static void* g_cpu;
static int g_cpu_tried;
static void* cpu_sym(const char* name) {
    if (!g_cpu_tried) {
        const char* p = getenv("ZSTD_JNI_CPU_LIB");
        g_cpu_tried = 1;
        if (p && *p) g_cpu = dlopen(p, RTLD_NOW | RTLD_LOCAL);
    }
    return g_cpu ? dlsym(g_cpu, name) : NULL;
}
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
    if (state < 0) { const char* e = getenv("ZSTD_JNI_GPU_PER_BUFFER"); state = ((e && *e == '1') || !cpu_sym("Java_com_github_luben_zstd_Zstd_compressBound")) ? 1 : 0; }
    return state && gpu_on();
}
static int gpu_result_final(size_t r) {       /* sizes and genuine libzstd error codes are final; 200/201 mean "not for the GPU path" */
    return !(zjni_isError(r) && zjni_getErrorCode(r) >= 200);
}

/* ---- contexts: what ZstdCompressCtx / ZstdDecompressCtx keep in nativePtr ---------------------------- */
typedef struct { int level; int checksum; int hashLog; int chainLog; jlong cpu; zjni_cdict* gdict; } ZCtx;     /* cpu = the bundled library's own ZSTD_CCtx handle, 0 if absent;
                                                                                       gdict = the GPU digest of the loaded ZstdDictCompress */
typedef struct { jlong cpu; zjni_ddict* gdict; } ZDCtx;                              /* gdict = the GPU digest of the loaded ZstdDictDecompress */

/* The parameter natives of class Zstd (setCompressionHashLog, ...) receive a raw context pointer that may also belong to a
 * stream class of the bundled library; the shim's own contexts are told apart by this registry. */
#define CTX_MAX 65536
static jlong g_ctxs[CTX_MAX];
static pthread_mutex_t g_ctx_mu = PTHREAD_MUTEX_INITIALIZER;
static int ctx_track(jlong p, int add) {            /* 0: registry full (init then fails: an untracked context could be mistaken for a stream's) */
    int ok = 0;
    pthread_mutex_lock(&g_ctx_mu);
    for (int i = 0; i < CTX_MAX; i++) if (g_ctxs[i] == (add ? 0 : p)) { g_ctxs[i] = add ? p : 0; ok = 1; break; }
    pthread_mutex_unlock(&g_ctx_mu);
    return ok;
}
static int ctx_is_ours(jlong p) {
    int r = 0;
    pthread_mutex_lock(&g_ctx_mu);
    for (int i = 0; i < CTX_MAX && !r; i++) r = (g_ctxs[i] == p && p != 0);
    pthread_mutex_unlock(&g_ctx_mu);
    return r;
}
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_ZstdCompressCtx_init(JNIEnv* env, jclass cls) {
    ZCtx* c = (ZCtx*)calloc(1, sizeof(ZCtx));
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym("Java_com_github_luben_zstd_ZstdCompressCtx_init");
    if (!c) return 0;
    c->level = 3;                                   /* ZSTD_CLEVEL_DEFAULT */
    if (!ctx_track((jlong)(intptr_t)c, 1)) { free(c); return 0; }
    if (f) c->cpu = f(env, cls);
    return (jlong)(intptr_t)c;
}
/* ZstdCompressCtx.setHashLog / setChainLog -> Zstd.setCompressionHashLog / ChainLog (N/jni_zstd.c:462-475): ZSTD_c_hashLog / ZSTD_c_chainLog */
static jint set_log(JNIEnv* env, jclass cls, jlong stream, jint v, int chain, const char* name) {
    jint (*f)(JNIEnv*, jclass, jlong, jint) = (jint (*)(JNIEnv*, jclass, jlong, jint))cpu_sym(name);
    if (ctx_is_ours(stream)) {
        ZCtx* c = (ZCtx*)(intptr_t)stream;
        if (chain) c->chainLog = v; else c->hashLog = v;
        return (f && c->cpu) ? f(env, cls, c->cpu, v) : 0;
    }
    return f ? f(env, cls, stream, v) : -(jint)ZJNI_ERROR_unsupported;   /* a stream class's context: the bundled library's business */
}
JNIEXPORT jint JNICALL Java_com_github_luben_zstd_Zstd_setCompressionHashLog(JNIEnv* env, jclass cls, jlong stream, jint hashLog) {
    return set_log(env, cls, stream, hashLog, 0, "Java_com_github_luben_zstd_Zstd_setCompressionHashLog");
}
JNIEXPORT jint JNICALL Java_com_github_luben_zstd_Zstd_setCompressionChainLog(JNIEnv* env, jclass cls, jlong stream, jint chainLog) {
    return set_log(env, cls, stream, chainLog, 1, "Java_com_github_luben_zstd_Zstd_setCompressionChainLog");
}
JNIEXPORT void JNICALL Java_com_github_luben_zstd_ZstdCompressCtx_free(JNIEnv* env, jclass cls, jlong ptr) {
    ZCtx* c = (ZCtx*)(intptr_t)ptr;
    void (*f)(JNIEnv*, jclass, jlong) = (void (*)(JNIEnv*, jclass, jlong))cpu_sym("Java_com_github_luben_zstd_ZstdCompressCtx_free");
    if (!c) return;
    ctx_track(ptr, 0);
    if (f && c->cpu) f(env, cls, c->cpu);
    free(c);
}
JNIEXPORT void JNICALL Java_com_github_luben_zstd_ZstdCompressCtx_setLevel0(JNIEnv* env, jclass cls, jlong ptr, jint level) {
    ZCtx* c = (ZCtx*)(intptr_t)ptr;
    void (*f)(JNIEnv*, jclass, jlong, jint) = (void (*)(JNIEnv*, jclass, jlong, jint))cpu_sym("Java_com_github_luben_zstd_ZstdCompressCtx_setLevel0");
    c->level = level;
    if (f && c->cpu) f(env, cls, c->cpu, level);
}
JNIEXPORT void JNICALL Java_com_github_luben_zstd_ZstdCompressCtx_setChecksum0(JNIEnv* env, jclass cls, jlong ptr, jboolean flag) {
    ZCtx* c = (ZCtx*)(intptr_t)ptr;
    void (*f)(JNIEnv*, jclass, jlong, jboolean) = (void (*)(JNIEnv*, jclass, jlong, jboolean))cpu_sym("Java_com_github_luben_zstd_ZstdCompressCtx_setChecksum0");
    c->checksum = (flag == JNI_TRUE);
    if (f && c->cpu) f(env, cls, c->cpu, flag);
}
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_ZstdDecompressCtx_init(JNIEnv* env, jclass cls) {
    ZDCtx* c = (ZDCtx*)calloc(1, sizeof(ZDCtx));
    jlong (*f)(JNIEnv*, jclass) = (jlong (*)(JNIEnv*, jclass))cpu_sym("Java_com_github_luben_zstd_ZstdDecompressCtx_init");
    if (!c) return 0;
    if (f) c->cpu = f(env, cls);
    return (jlong)(intptr_t)c;
}
JNIEXPORT void JNICALL Java_com_github_luben_zstd_ZstdDecompressCtx_free(JNIEnv* env, jclass cls, jlong ptr) {
    ZDCtx* c = (ZDCtx*)(intptr_t)ptr;
    void (*f)(JNIEnv*, jclass, jlong) = (void (*)(JNIEnv*, jclass, jlong))cpu_sym("Java_com_github_luben_zstd_ZstdDecompressCtx_free");
    if (!c) return;
    if (f && c->cpu) f(env, cls, c->cpu);
    free(c);
}

/* ---- ZstdDictCompress (N/jni_fast_zstd.c:13-66) -------------------------------------------------------
 * The Java object keeps ONE long (nativePtr), and the bundled library's natives read it as their own ZSTD_CDict*.
 * So nativePtr stays what the bundled library put there (when it is present), and the GPU digest of the same
 * dictionary lives in a side table keyed by that value; without the bundled library nativePtr is the table key
 * itself (a private handle). */
typedef struct { jlong key; void* gpu; int level; } DictEnt;              /* gpu: zjni_cdict* or zjni_ddict* (keys are distinct heap addresses) */
#define DICT_MAX 4096
static DictEnt g_dicts[DICT_MAX];
static pthread_mutex_t g_dict_mu = PTHREAD_MUTEX_INITIALIZER;
static jfieldID g_cdict_field;
static int dict_put(jlong key, void* gpu, int level) {           /* 0 when the table is full (the caller keeps the CPU path for this dictionary) */
    int ok = 0;
    pthread_mutex_lock(&g_dict_mu);
    for (int i = 0; i < DICT_MAX; i++) if (!g_dicts[i].key) { g_dicts[i].key = key; g_dicts[i].gpu = gpu; g_dicts[i].level = level; ok = 1; break; }
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
static void dict_register(JNIEnv* env, jobject obj, const void* bytes, size_t size, jint level) {
    jlong key = (*env)->GetLongField(env, obj, g_cdict_field);          /* the bundled library's ZSTD_CDict*, if it made one */
    zjni_cdict* gpu = (gpu_on() && level >= 1 && level <= 3) ? zjni_createCDict(bytes, size, level) : NULL;
    if (!key) {                                                         /* no bundled library: the handle is ours */
        if (!gpu) return;                                               /* nativePtr stays 0: "ZSTD_createCDict failed" on the Java side */
        key = (jlong)(intptr_t)gpu;
        (*env)->SetLongField(env, obj, g_cdict_field, key);
    }
    if (gpu && !dict_put(key, gpu, level)) {
        zjni_freeCDict(gpu);
        if (key == (jlong)(intptr_t)gpu) (*env)->SetLongField(env, obj, g_cdict_field, 0);
    }
}
JNIEXPORT void JNICALL Java_com_github_luben_zstd_ZstdDictCompress_init
  (JNIEnv* env, jobject obj, jbyteArray dict, jint dict_offset, jint dict_size, jint level) {
    void (*f)(JNIEnv*, jobject, jbyteArray, jint, jint, jint) = (void (*)(JNIEnv*, jobject, jbyteArray, jint, jint, jint))cpu_sym("Java_com_github_luben_zstd_ZstdDictCompress_init");
    jclass clazz = (*env)->GetObjectClass(env, obj);
    g_cdict_field = (*env)->GetFieldID(env, clazz, "nativePtr", "J");
    if (NULL == dict) return;
    if (f) f(env, obj, dict, dict_offset, dict_size, level);
    if (dict_size >= 0) {
        jbyte* copy = (jbyte*)malloc((size_t)dict_size + 1);
        if (!copy) return;
        (*env)->GetByteArrayRegion(env, dict, dict_offset, dict_size, copy);
        dict_register(env, obj, copy, (size_t)dict_size, level);
        free(copy);
    }
}
JNIEXPORT void JNICALL Java_com_github_luben_zstd_ZstdDictCompress_initDirect
  (JNIEnv* env, jobject obj, jobject dict, jint dict_offset, jint dict_size, jint level, jint byReference) {
    void (*f)(JNIEnv*, jobject, jobject, jint, jint, jint, jint) = (void (*)(JNIEnv*, jobject, jobject, jint, jint, jint, jint))cpu_sym("Java_com_github_luben_zstd_ZstdDictCompress_initDirect");
    jclass clazz = (*env)->GetObjectClass(env, obj);
    g_cdict_field = (*env)->GetFieldID(env, clazz, "nativePtr", "J");
    if (NULL == dict) return;
    if (f) f(env, obj, dict, dict_offset, dict_size, level, byReference);
    {   char* p = (char*)(*env)->GetDirectBufferAddress(env, dict);
        if (p && dict_size >= 0) dict_register(env, obj, p + dict_offset, (size_t)dict_size, level); }   /* the device keeps its own copy either way */
}
JNIEXPORT void JNICALL Java_com_github_luben_zstd_ZstdDictCompress_free(JNIEnv* env, jobject obj) {
    void (*f)(JNIEnv*, jobject) = (void (*)(JNIEnv*, jobject))cpu_sym("Java_com_github_luben_zstd_ZstdDictCompress_free");
    if (g_cdict_field) {
        jlong const key = (*env)->GetLongField(env, obj, g_cdict_field);
        zjni_cdict* gpu = (zjni_cdict*)dict_get(key, 1);
        if (gpu) zjni_freeCDict(gpu);
    }
    if (f) f(env, obj);
}
/* ZstdCompressCtx.loadDict(ZstdDictCompress) -> ZSTD_CCtx_refCDict (N/jni_fast_zstd.c:325-336) */
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_ZstdCompressCtx_loadCDictFast0(JNIEnv* env, jclass cls, jlong ptr, jobject dict) {
    ZCtx* c = (ZCtx*)(intptr_t)ptr;
    jlong (*f)(JNIEnv*, jclass, jlong, jobject) = (jlong (*)(JNIEnv*, jclass, jlong, jobject))cpu_sym("Java_com_github_luben_zstd_ZstdCompressCtx_loadCDictFast0");
    jlong r = 0;
    c->gdict = NULL;
    if (dict != NULL) {
        jlong const key = g_cdict_field ? (*env)->GetLongField(env, dict, g_cdict_field) : 0;
        if (!key) return -32;                                           /* -ZSTD_error_dictionary_wrong */
        c->gdict = (zjni_cdict*)dict_get(key, 0);
    }
    if (f && c->cpu) r = f(env, cls, c->cpu, dict);
    return r;
}

/* ---- compress: ZSTD_CCtx_reset + ZSTD_compress2 (N/jni_fast_zstd.c:606-607, :633-635) ----------------- */
static int gpu_takes(const ZCtx* c, jint srcSize) {
    if (c->gdict) return per_buffer_on_gpu();                           /* sizes beyond the attach range come back as 40 and are forwarded */
    return per_buffer_on_gpu() && c->level >= 1 && c->level <= 3 && (size_t)srcSize <= ZJNI_BLOCKSIZE_MAX;
}
static size_t gpu_compress(const ZCtx* c, void* dst, size_t dstCap, const void* src, size_t srcSize) {
    if (c->gdict) {
        size_t res = 0; const void* s = src; void* d = dst;
        size_t const r = zjni_compress_batch_usingCDict(&s, &srcSize, &d, &dstCap, &res, 1, c->gdict, c->checksum);
        return zjni_isError(r) ? r : res;
    }
    if (c->hashLog || c->chainLog) {                /* explicit table sizes: level 3 only on the GPU (40 otherwise -> forwarded) */
        size_t res = 0; const void* s = src; void* d = dst;
        size_t const r = zjni_compress_batch_advanced(&s, &srcSize, &d, &dstCap, &res, 1, c->level, c->checksum, c->hashLog, c->chainLog);
        return zjni_isError(r) ? r : res;
    }
    return zjni_compress2(dst, dstCap, src, srcSize, c->level, c->checksum);
}
static int gpu_compress_final(const ZCtx* c, size_t r) {   /* 40 = a dictionary frame outside the attach range / table sizes the GPU path does not take: forward when possible */
    return gpu_result_final(r) && !((c->gdict || c->hashLog || c->chainLog) && zjni_isError(r) && (zjni_getErrorCode(r) == 40 || zjni_getErrorCode(r) == 42) && c->cpu);
}
typedef jlong (*cbuf_fn)(JNIEnv*, jclass, jlong, jobject, jint, jint, jobject, jint, jint);

JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_ZstdCompressCtx_compressDirectByteBuffer0
  (JNIEnv* env, jclass cls, jlong ptr, jobject dst, jint dst_offset, jint dst_size, jobject src, jint src_offset, jint src_size) {
    ZCtx* c = (ZCtx*)(intptr_t)ptr;
    if (NULL == dst) return E_DST;
    if (NULL == src) return E_SRC;
    if (0 > dst_offset) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_size) return E_SRC;
    if (dst_offset + dst_size > (*env)->GetDirectBufferCapacity(env, dst)) return E_DST;
    if (src_offset + src_size > (*env)->GetDirectBufferCapacity(env, src)) return E_SRC;
    {   char* d = (char*)(*env)->GetDirectBufferAddress(env, dst);
        char* s = (char*)(*env)->GetDirectBufferAddress(env, src);
        if (d == NULL || s == NULL) return E_MEM;
        if (gpu_takes(c, src_size)) {
            size_t const r = gpu_compress(c, d + dst_offset, (size_t)dst_size, s + src_offset, (size_t)src_size);
            if (gpu_compress_final(c, r)) return (jlong)r;
        }
    }
    {   cbuf_fn f = (cbuf_fn)cpu_sym("Java_com_github_luben_zstd_ZstdCompressCtx_compressDirectByteBuffer0");
        if (f && c->cpu) return f(env, cls, c->cpu, dst, dst_offset, dst_size, src, src_offset, src_size);
    }
    return -(jlong)ZJNI_ERROR_unsupported;
}

JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_ZstdCompressCtx_compressByteArray0
  (JNIEnv* env, jclass cls, jlong ptr, jbyteArray dst, jint dst_offset, jint dst_size, jbyteArray src, jint src_offset, jint src_size) {
    ZCtx* c = (ZCtx*)(intptr_t)ptr;
    if (0 > dst_offset) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_size) return E_SRC;
    if (src_offset + src_size > (*env)->GetArrayLength(env, src)) return E_SRC;
    if (dst_offset + dst_size > (*env)->GetArrayLength(env, dst)) return E_DST;
    if (gpu_takes(c, src_size)) {
        jbyte* s = (jbyte*)malloc((size_t)src_size + 1); jbyte* d = (jbyte*)malloc((size_t)dst_size + 1);
        size_t r = (size_t)E_MEM;
        if (s && d) {
            (*env)->GetByteArrayRegion(env, src, src_offset, src_size, s);
            r = gpu_compress(c, d, (size_t)dst_size, s, (size_t)src_size);
            if (!zjni_isError(r)) (*env)->SetByteArrayRegion(env, dst, dst_offset, (jsize)r, d);
        }
        free(s); free(d);
        if (gpu_compress_final(c, r)) return (jlong)r;
    }
    {   cbuf_fn f = (cbuf_fn)cpu_sym("Java_com_github_luben_zstd_ZstdCompressCtx_compressByteArray0");
        if (f && c->cpu) return f(env, cls, c->cpu, dst, dst_offset, dst_size, src, src_offset, src_size);
    }
    return -(jlong)ZJNI_ERROR_unsupported;
}

/* ---- decompress: ZSTD_DCtx_reset + ZSTD_decompressDCtx (N/jni_fast_zstd.c:798-799, :825-826, :858-860, :892-894) */
/* ---- ZstdDictDecompress (N/jni_fast_zstd.c:68-125) + ZstdDecompressCtx.loadDict (:673-684): same arrangement as ZstdDictCompress */
static jfieldID g_ddict_field;
static void ddict_register(JNIEnv* env, jobject obj, const void* bytes, size_t size) {
    jlong key = (*env)->GetLongField(env, obj, g_ddict_field);
    zjni_ddict* gpu = gpu_on() ? zjni_createDDict(bytes, size) : NULL;
    if (!key) {
        if (!gpu) return;
        key = (jlong)(intptr_t)gpu;
        (*env)->SetLongField(env, obj, g_ddict_field, key);
    }
    if (gpu && !dict_put(key, gpu, 0)) {
        zjni_freeDDict(gpu);
        if (key == (jlong)(intptr_t)gpu) (*env)->SetLongField(env, obj, g_ddict_field, 0);
    }
}
JNIEXPORT void JNICALL Java_com_github_luben_zstd_ZstdDictDecompress_init(JNIEnv* env, jobject obj, jbyteArray dict, jint dict_offset, jint dict_size) {
    void (*f)(JNIEnv*, jobject, jbyteArray, jint, jint) = (void (*)(JNIEnv*, jobject, jbyteArray, jint, jint))cpu_sym("Java_com_github_luben_zstd_ZstdDictDecompress_init");
    jclass clazz = (*env)->GetObjectClass(env, obj);
    g_ddict_field = (*env)->GetFieldID(env, clazz, "nativePtr", "J");
    if (NULL == dict) return;
    if (f) f(env, obj, dict, dict_offset, dict_size);
    if (dict_size >= 0) {
        jbyte* copy = (jbyte*)malloc((size_t)dict_size + 1);
        if (!copy) return;
        (*env)->GetByteArrayRegion(env, dict, dict_offset, dict_size, copy);
        ddict_register(env, obj, copy, (size_t)dict_size);
        free(copy);
    }
}
JNIEXPORT void JNICALL Java_com_github_luben_zstd_ZstdDictDecompress_initDirect(JNIEnv* env, jobject obj, jobject dict, jint dict_offset, jint dict_size, jint byReference) {
    void (*f)(JNIEnv*, jobject, jobject, jint, jint, jint) = (void (*)(JNIEnv*, jobject, jobject, jint, jint, jint))cpu_sym("Java_com_github_luben_zstd_ZstdDictDecompress_initDirect");
    jclass clazz = (*env)->GetObjectClass(env, obj);
    g_ddict_field = (*env)->GetFieldID(env, clazz, "nativePtr", "J");
    if (NULL == dict) return;
    if (f) f(env, obj, dict, dict_offset, dict_size, byReference);
    {   char* p = (char*)(*env)->GetDirectBufferAddress(env, dict);
        if (p && dict_size >= 0) ddict_register(env, obj, p + dict_offset, (size_t)dict_size); }
}
JNIEXPORT void JNICALL Java_com_github_luben_zstd_ZstdDictDecompress_free(JNIEnv* env, jobject obj) {
    void (*f)(JNIEnv*, jobject) = (void (*)(JNIEnv*, jobject))cpu_sym("Java_com_github_luben_zstd_ZstdDictDecompress_free");
    if (g_ddict_field) {
        zjni_ddict* gpu = (zjni_ddict*)dict_get((*env)->GetLongField(env, obj, g_ddict_field), 1);
        if (gpu) zjni_freeDDict(gpu);
    }
    if (f) f(env, obj);
}
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_ZstdDecompressCtx_loadDDictFast0(JNIEnv* env, jclass cls, jlong ptr, jobject dict) {
    ZDCtx* c = (ZDCtx*)(intptr_t)ptr;
    jlong (*f)(JNIEnv*, jclass, jlong, jobject) = (jlong (*)(JNIEnv*, jclass, jlong, jobject))cpu_sym("Java_com_github_luben_zstd_ZstdDecompressCtx_loadDDictFast0");
    jlong r = 0;
    c->gdict = NULL;
    if (dict != NULL) {
        jlong const key = g_ddict_field ? (*env)->GetLongField(env, dict, g_ddict_field) : 0;
        if (!key) return -32;                                           /* -ZSTD_error_dictionary_wrong */
        c->gdict = (zjni_ddict*)dict_get(key, 0);
    }
    if (f && c->cpu) r = f(env, cls, c->cpu, dict);
    return r;
}
static size_t gpu_decompress(const ZDCtx* c, void* dst, size_t dstCap, const void* src, size_t srcSize) {
    return c->gdict ? zjni_decompress_usingDDict(dst, dstCap, src, srcSize, c->gdict) : zjni_decompress(dst, dstCap, src, srcSize);
}

static jlong dec_forward(const char* name, JNIEnv* env, jclass cls, ZDCtx* c, jobject dst, jint doff, jint dsize, jobject src, jint soff, jint ssize) {
    cbuf_fn f = (cbuf_fn)cpu_sym(name);
    if (f && c->cpu) return f(env, cls, c->cpu, dst, doff, dsize, src, soff, ssize);
    return -(jlong)ZJNI_ERROR_no_device;
}
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_ZstdDecompressCtx_decompressDirectByteBuffer0
  (JNIEnv* env, jclass cls, jlong ptr, jobject dst, jint dst_offset, jint dst_size, jobject src, jint src_offset, jint src_size) {
    ZDCtx* c = (ZDCtx*)(intptr_t)ptr;
    if (NULL == dst) return E_DST;
    if (NULL == src) return E_SRC;
    if (0 > dst_offset) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_size) return E_SRC;
    if (dst_offset + dst_size > (*env)->GetDirectBufferCapacity(env, dst)) return E_DST;
    if (src_offset + src_size > (*env)->GetDirectBufferCapacity(env, src)) return E_SRC;
    {   char* d = (char*)(*env)->GetDirectBufferAddress(env, dst);
        char* s = (char*)(*env)->GetDirectBufferAddress(env, src);
        if (d == NULL || s == NULL) return E_MEM;
        if (per_buffer_on_gpu()) {
            size_t const r = gpu_decompress(c, d + dst_offset, (size_t)dst_size, s + src_offset, (size_t)src_size);
            if (gpu_result_final(r)) return (jlong)r;
        }
    }
    return dec_forward("Java_com_github_luben_zstd_ZstdDecompressCtx_decompressDirectByteBuffer0", env, cls, c, dst, dst_offset, dst_size, src, src_offset, src_size);
}
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_ZstdDecompressCtx_decompressByteArray0
  (JNIEnv* env, jclass cls, jlong ptr, jbyteArray dst, jint dst_offset, jint dst_size, jbyteArray src, jint src_offset, jint src_size) {
    ZDCtx* c = (ZDCtx*)(intptr_t)ptr;
    if (0 > dst_offset) return E_DST;
    if (0 > src_offset) return E_SRC;
    if (0 > src_size) return E_SRC;
    if (src_offset + src_size > (*env)->GetArrayLength(env, src)) return E_SRC;
    if (dst_offset + dst_size > (*env)->GetArrayLength(env, dst)) return E_DST;
    if (per_buffer_on_gpu()) {
        jbyte* s = (jbyte*)malloc((size_t)src_size + 1); jbyte* d = (jbyte*)malloc((size_t)dst_size + 1);
        size_t r = (size_t)E_MEM;
        if (s && d) {
            (*env)->GetByteArrayRegion(env, src, src_offset, src_size, s);
            r = gpu_decompress(c, d, (size_t)dst_size, s, (size_t)src_size);
            if (!zjni_isError(r)) (*env)->SetByteArrayRegion(env, dst, dst_offset, (jsize)r, d);
        }
        free(s); free(d);
        if (gpu_result_final(r)) return (jlong)r;
    }
    return dec_forward("Java_com_github_luben_zstd_ZstdDecompressCtx_decompressByteArray0", env, cls, c, dst, dst_offset, dst_size, src, src_offset, src_size);
}

/* ---- class Zstd: helpers with the reference's semantics (N/jni_zstd.c:230-267, :50-63) ---------------- */
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_Zstd_compressBound(JNIEnv* env, jclass cls, jlong size) {
    (void)env; (void)cls; return (jlong)zjni_compressBound((size_t)size);
}
JNIEXPORT jboolean JNICALL Java_com_github_luben_zstd_Zstd_isError(JNIEnv* env, jclass cls, jlong code) {
    (void)env; (void)cls; return zjni_isError((size_t)code) != 0;
}
JNIEXPORT jstring JNICALL Java_com_github_luben_zstd_Zstd_getErrorName(JNIEnv* env, jclass cls, jlong code) {
    (void)cls; return (*env)->NewStringUTF(env, zjni_getErrorName((size_t)code));
}
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_Zstd_getErrorCode(JNIEnv* env, jclass cls, jlong code) {
    (void)env; (void)cls; return (jlong)zjni_getErrorCode((size_t)code);
}
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_Zstd_compressUnsafe
  (JNIEnv* env, jclass cls, jlong dst, jlong dst_size, jlong src, jlong src_size, jint level, jboolean checksumFlag) {
    if (per_buffer_on_gpu() && level >= 1 && level <= 3 && (size_t)src_size <= ZJNI_BLOCKSIZE_MAX) {
        size_t const r = zjni_compress2((void*)(intptr_t)dst, (size_t)dst_size, (const void*)(intptr_t)src, (size_t)src_size, level, checksumFlag == JNI_TRUE);
        if (gpu_result_final(r)) return (jlong)r;
    }
    {   jlong (*f)(JNIEnv*, jclass, jlong, jlong, jlong, jlong, jint, jboolean) =
            (jlong (*)(JNIEnv*, jclass, jlong, jlong, jlong, jlong, jint, jboolean))cpu_sym("Java_com_github_luben_zstd_Zstd_compressUnsafe");
        if (f) return f(env, cls, dst, dst_size, src, src_size, level, checksumFlag);
    }
    return -(jlong)ZJNI_ERROR_unsupported;
}
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_Zstd_decompressUnsafe
  (JNIEnv* env, jclass cls, jlong dst, jlong dst_size, jlong src, jlong src_size) {
    if (per_buffer_on_gpu()) {
        size_t const r = zjni_decompress((void*)(intptr_t)dst, (size_t)dst_size, (const void*)(intptr_t)src, (size_t)src_size);
        if (gpu_result_final(r)) return (jlong)r;
    }
    {   jlong (*f)(JNIEnv*, jclass, jlong, jlong, jlong, jlong) =
            (jlong (*)(JNIEnv*, jclass, jlong, jlong, jlong, jlong))cpu_sym("Java_com_github_luben_zstd_Zstd_decompressUnsafe");
        if (f) return f(env, cls, dst, dst_size, src, src_size);
    }
    return -(jlong)ZJNI_ERROR_no_device;
}

/* ---- new, additive: batch natives over arrays of direct ByteBuffers ----------------
 * static native long compressBatch0(ByteBuffer[] srcs, ByteBuffer[] dsts, long[] results, int level, boolean checksum);
 * static native long decompressBatch0(ByteBuffer[] srcs, ByteBuffer[] dsts, long[] results);
 * Each buffer is taken from position 0 to its capacity.  results[i] = size or the error code compress*0 /
 * decompress*0 would have returned for that buffer; the return value is 0 or a launch-level error. */
static jlong batch(JNIEnv* env, jobjectArray srcs, jobjectArray dsts, jlongArray results, int compress, int level, int checksum, const zjni_cdict* cdict) {
    jsize const n = (*env)->GetArrayLength(env, srcs);
    const void** sp; void** dp; size_t* ss; size_t* dc; size_t* res; jlong* out; size_t r; jsize i;
    if ((*env)->GetArrayLength(env, dsts) != n || (*env)->GetArrayLength(env, results) < n) return E_SRC;
    if (n == 0) return 0;
    sp = (const void**)malloc(n * sizeof(*sp)); dp = (void**)malloc(n * sizeof(*dp));
    ss = (size_t*)malloc(n * sizeof(*ss)); dc = (size_t*)malloc(n * sizeof(*dc)); res = (size_t*)malloc(n * sizeof(*res));
    out = (jlong*)malloc(n * sizeof(*out));
    if (!sp || !dp || !ss || !dc || !res || !out) { free(sp); free(dp); free(ss); free(dc); free(res); free(out); return E_MEM; }
    for (i = 0; i < n; i++) {
        jobject s = (*env)->GetObjectArrayElement(env, srcs, i), d = (*env)->GetObjectArrayElement(env, dsts, i);
        sp[i] = (*env)->GetDirectBufferAddress(env, s); ss[i] = (size_t)(*env)->GetDirectBufferCapacity(env, s);
        dp[i] = (*env)->GetDirectBufferAddress(env, d); dc[i] = (size_t)(*env)->GetDirectBufferCapacity(env, d);
    }
    r = cdict ? zjni_compress_batch_usingCDict(sp, ss, dp, dc, res, (size_t)n, cdict, checksum)
      : compress ? zjni_compress_batch2(sp, ss, dp, dc, res, (size_t)n, level, checksum) : zjni_decompress_batch(sp, ss, dp, dc, res, (size_t)n);
    if (!zjni_isError(r)) { for (i = 0; i < n; i++) out[i] = (jlong)res[i]; (*env)->SetLongArrayRegion(env, results, 0, n, out); }
    free(sp); free(dp); free(ss); free(dc); free(res); free(out);
    return (jlong)r;
}
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_Zstd_compressBatch0
  (JNIEnv* env, jclass cls, jobjectArray srcs, jobjectArray dsts, jlongArray results, jint level, jboolean checksum) {
    (void)cls;
    if (!gpu_on()) return -(jlong)ZJNI_ERROR_no_device;
    return batch(env, srcs, dsts, results, 1, level, checksum == JNI_TRUE, NULL);
}
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_Zstd_decompressBatch0
  (JNIEnv* env, jclass cls, jobjectArray srcs, jobjectArray dsts, jlongArray results) {
    (void)cls;
    if (!gpu_on()) return -(jlong)ZJNI_ERROR_no_device;
    return batch(env, srcs, dsts, results, 0, 0, 0, NULL);
}
/* static native long compressBatchDict0(ByteBuffer[] srcs, ByteBuffer[] dsts, long[] results, ZstdDictCompress dict, boolean checksum); */
JNIEXPORT jlong JNICALL Java_com_github_luben_zstd_Zstd_compressBatchDict0
  (JNIEnv* env, jclass cls, jobjectArray srcs, jobjectArray dsts, jlongArray results, jobject dict, jboolean checksum) {
    zjni_cdict* g;
    (void)cls;
    if (!gpu_on()) return -(jlong)ZJNI_ERROR_no_device;
    if (dict == NULL || !g_cdict_field) return -32;
    g = (zjni_cdict*)dict_get((*env)->GetLongField(env, dict, g_cdict_field), 0);
    if (!g) return -32;
    return batch(env, srcs, dsts, results, 1, 0, checksum == JNI_TRUE, g);
}
