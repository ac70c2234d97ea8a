/* tests/jni/harness.c — drives JNI natives the way the JVM would, through a hand-built JNIEnv (only the ~12 env
 * functions the one-shot hot path uses), against TWO libraries:
 *   the reference's own JNI library (oracle/_ref/libzstd-jni-ref.so, built from the reference sources in place)
 *   the GPU shim (zstd-jni_amd/lib/libzstd-jni-amd.so)
 * and checks that the hot-path natives return the same values and write the same bytes.  TEST INFRASTRUCTURE.
 * usage: harness <ref-jni.so> <shim.so>        prints "JNI-HARNESS OK checks=N" or the first mismatches; exit 0/1 */
#include <jni.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct Obj { int kind; char* data; jsize len; struct Obj** elems; jlong field; jlong srcPos, dstPos; jint consumed, produced; int borrowed; } Obj;   /* kind 1 direct buffer, 2 byte[], 3 Object[], 4 long[], 5 string, 6 object with one long field (nativePtr) */
static Obj* mk(int kind, jsize len) { Obj* o = (Obj*)calloc(1, sizeof(Obj)); o->kind = kind; o->len = len; o->data = (char*)calloc((size_t)len + 16, kind == 4 ? 8 : 1); return o; }

static void* JNICALL f_GetDirectBufferAddress(JNIEnv* e, jobject b) { (void)e; return (b && ((Obj*)b)->kind == 1) ? ((Obj*)b)->data : NULL; }
static jlong JNICALL f_GetDirectBufferCapacity(JNIEnv* e, jobject b) { (void)e; return (b && ((Obj*)b)->kind == 1) ? ((Obj*)b)->len : -1; }
static jsize JNICALL f_GetArrayLength(JNIEnv* e, jarray a) { (void)e; return ((Obj*)a)->len; }
static void* JNICALL f_GetPrimitiveArrayCritical(JNIEnv* e, jarray a, jboolean* c) { (void)e; if (c) *c = JNI_FALSE; return ((Obj*)a)->data; }
static void JNICALL f_ReleasePrimitiveArrayCritical(JNIEnv* e, jarray a, void* p, jint m) { (void)e; (void)a; (void)p; (void)m; }
static void JNICALL f_GetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize s, jsize l, jbyte* buf) { (void)e; memcpy(buf, ((Obj*)a)->data + s, (size_t)l); }
static void JNICALL f_SetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize s, jsize l, const jbyte* buf) { (void)e; memcpy(((Obj*)a)->data + s, buf, (size_t)l); }
static jobject JNICALL f_GetObjectArrayElement(JNIEnv* e, jobjectArray a, jsize i) { (void)e; return (jobject)((Obj*)a)->elems[i]; }
static void JNICALL f_SetLongArrayRegion(JNIEnv* e, jlongArray a, jsize s, jsize l, const jlong* buf) { (void)e; memcpy(((Obj*)a)->data + 8 * (size_t)s, buf, 8 * (size_t)l); }
static jclass JNICALL f_GetObjectClass(JNIEnv* e, jobject o) { (void)e; return (jclass)o; }
static jfieldID JNICALL f_GetFieldID(JNIEnv* e, jclass c, const char* n, const char* sig) { (void)e; (void)c; (void)sig; return !strcmp(n, "nativePtr") ? (jfieldID)(intptr_t)1 : (!strcmp(n, "consumed") ? (jfieldID)(intptr_t)2 : (!strcmp(n, "produced") ? (jfieldID)(intptr_t)3 : (!strcmp(n, "srcPos") ? (jfieldID)(intptr_t)4 : (!strcmp(n, "dstPos") ? (jfieldID)(intptr_t)5 : NULL)))); }
static jint JNICALL f_GetIntField(JNIEnv* e, jobject o, jfieldID f) { (void)e; return (intptr_t)f == 2 ? ((Obj*)o)->consumed : ((Obj*)o)->produced; }
static void JNICALL f_SetIntField(JNIEnv* e, jobject o, jfieldID f, jint v) { (void)e; if ((intptr_t)f == 2) ((Obj*)o)->consumed = v; else ((Obj*)o)->produced = v; }
static jobject JNICALL f_NewDirectByteBuffer(JNIEnv* e, void* addr, jlong cap) { (void)e; Obj* o = (Obj*)calloc(1, sizeof(Obj)); o->kind = 1; o->len = (jsize)cap; o->data = (char*)addr; o->borrowed = 1; return (jobject)o; }
static jlong JNICALL f_GetLongField(JNIEnv* e, jobject o, jfieldID f) { (void)e; return (intptr_t)f == 4 ? ((Obj*)o)->srcPos : ((intptr_t)f == 5 ? ((Obj*)o)->dstPos : ((Obj*)o)->field); }
static void JNICALL f_SetLongField(JNIEnv* e, jobject o, jfieldID f, jlong v) { (void)e; if ((intptr_t)f == 4) ((Obj*)o)->srcPos = v; else if ((intptr_t)f == 5) ((Obj*)o)->dstPos = v; else ((Obj*)o)->field = v; }
static jbyteArray JNICALL f_NewByteArray(JNIEnv* e, jsize n) { (void)e; return (jbyteArray)mk(2, n); }
static jstring JNICALL f_NewStringUTF(JNIEnv* e, const char* s) { (void)e; Obj* o = mk(5, (jsize)strlen(s) + 1); strcpy(o->data, s); return (jstring)o; }

static jclass JNICALL f_FindClass(JNIEnv* e, const char* n) { (void)e; (void)n; return (jclass)mk(7, 0); }
static jmethodID JNICALL f_GetMethodID(JNIEnv* e, jclass c, const char* n, const char* sig) { (void)e; (void)c; (void)n; (void)sig; return (jmethodID)(intptr_t)2; }
static jobject JNICALL f_NewObject(JNIEnv* e, jclass c, jmethodID m, ...) { (void)e; (void)c; (void)m; return (jobject)mk(7, 0); }
static jbyte* JNICALL f_GetByteArrayElements(JNIEnv* e, jbyteArray a, jboolean* c) { (void)e; if (c) *c = JNI_FALSE; return (jbyte*)((Obj*)a)->data; }
static void JNICALL f_ReleaseByteArrayElements(JNIEnv* e, jbyteArray a, jbyte* p, jint m) { (void)e; (void)a; (void)p; (void)m; }
static jboolean JNICALL f_ExceptionCheck(JNIEnv* e) { (void)e; return JNI_FALSE; }
static int g_deleted;
static void JNICALL f_DeleteLocalRef(JNIEnv* e, jobject o) { (void)e; (void)o; g_deleted++; }
static int g_globals = 0;                                 /* global references alive: the asynchronous batch natives hold two per job until batchFinish0 */
static jobject JNICALL f_NewGlobalRef(JNIEnv* e, jobject o) { (void)e; if (o) g_globals++; return o; }
static void JNICALL f_DeleteGlobalRef(JNIEnv* e, jobject o) { (void)e; if (o) g_globals--; }
static struct JNINativeInterface_ g_fn;
static const struct JNINativeInterface_* g_envp = &g_fn;
static JNIEnv* env(void) {
    g_fn.GetDirectBufferAddress = f_GetDirectBufferAddress; g_fn.GetDirectBufferCapacity = f_GetDirectBufferCapacity;
    g_fn.GetArrayLength = f_GetArrayLength; g_fn.GetPrimitiveArrayCritical = f_GetPrimitiveArrayCritical;
    g_fn.ReleasePrimitiveArrayCritical = f_ReleasePrimitiveArrayCritical; g_fn.GetByteArrayRegion = f_GetByteArrayRegion;
    g_fn.SetByteArrayRegion = f_SetByteArrayRegion; g_fn.GetObjectArrayElement = f_GetObjectArrayElement;
    g_fn.SetLongArrayRegion = f_SetLongArrayRegion; g_fn.NewStringUTF = f_NewStringUTF;
    g_fn.GetByteArrayElements = f_GetByteArrayElements; g_fn.ReleaseByteArrayElements = f_ReleaseByteArrayElements;
    g_fn.FindClass = f_FindClass; g_fn.GetMethodID = f_GetMethodID; g_fn.NewObject = f_NewObject; g_fn.DeleteLocalRef = f_DeleteLocalRef; g_fn.NewGlobalRef = f_NewGlobalRef; g_fn.DeleteGlobalRef = f_DeleteGlobalRef;
    g_fn.ExceptionCheck = f_ExceptionCheck; g_fn.GetIntField = f_GetIntField; g_fn.SetIntField = f_SetIntField; g_fn.NewDirectByteBuffer = f_NewDirectByteBuffer;
    g_fn.GetObjectClass = f_GetObjectClass; g_fn.GetFieldID = f_GetFieldID; g_fn.GetLongField = f_GetLongField; g_fn.SetLongField = f_SetLongField; g_fn.NewByteArray = f_NewByteArray;
    return (JNIEnv*)&g_envp;
}

/* the natives of one library */
typedef struct {
    jlong (*cinit)(JNIEnv*, jclass); void (*cfree)(JNIEnv*, jclass, jlong);
    void (*setLevel)(JNIEnv*, jclass, jlong, jint); void (*setChecksum)(JNIEnv*, jclass, jlong, jboolean);
    jlong (*cDirect)(JNIEnv*, jclass, jlong, jobject, jint, jint, jobject, jint, jint);
    jlong (*cArray)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint, jbyteArray, jint, jint);
    jlong (*dinit)(JNIEnv*, jclass); void (*dfree)(JNIEnv*, jclass, jlong);
    jlong (*dDirect)(JNIEnv*, jclass, jlong, jobject, jint, jint, jobject, jint, jint);
    jlong (*dArray)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint, jbyteArray, jint, jint);
    jlong (*bound)(JNIEnv*, jclass, jlong); jboolean (*isError)(JNIEnv*, jclass, jlong);
    jstring (*errName)(JNIEnv*, jclass, jlong); jlong (*errCode)(JNIEnv*, jclass, jlong);
    jlong (*cUnsafe)(JNIEnv*, jclass, jlong, jlong, jlong, jlong, jint, jboolean);
    jlong (*dUnsafe)(JNIEnv*, jclass, jlong, jlong, jlong, jlong);
    jint (*setHashLog)(JNIEnv*, jclass, jlong, jint); jint (*setChainLog)(JNIEnv*, jclass, jlong, jint);     /* reference only */
    jlong (*cBatch)(JNIEnv*, jclass, jobjectArray, jobjectArray, jlongArray, jint, jboolean);                 /* shim only */
    jlong (*dBatch)(JNIEnv*, jclass, jobjectArray, jobjectArray, jlongArray);
    jlong (*cBatchBegin)(JNIEnv*, jclass, jobjectArray, jobjectArray, jint, jboolean);                         /* shim only (round 6) */
    jlong (*dBatchBegin)(JNIEnv*, jclass, jobjectArray, jobjectArray);
    jlong (*batchFinish)(JNIEnv*, jclass, jlong, jlongArray);
    void (*dictInit)(JNIEnv*, jobject, jbyteArray, jint, jint, jint); void (*dictInitDirect)(JNIEnv*, jobject, jobject, jint, jint, jint, jint);
    void (*dictFree)(JNIEnv*, jobject); jlong (*loadCDict)(JNIEnv*, jclass, jlong, jobject);
    void (*ddictInit)(JNIEnv*, jobject, jbyteArray, jint, jint); void (*ddictInitDirect)(JNIEnv*, jobject, jobject, jint, jint, jint);
    void (*ddictFree)(JNIEnv*, jobject); jlong (*loadDDict)(JNIEnv*, jclass, jlong, jobject);
    jlong (*cBatchDict)(JNIEnv*, jclass, jobjectArray, jobjectArray, jlongArray, jobject, jboolean);                /* shim only */
    void* h;
} Lib;
#define P "Java_com_github_luben_zstd_"
static int load(Lib* L, const char* path, int isRef) {
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen %s: %s\n", path, dlerror()); return 0; }
    L->h = h;
#define S(field, name) *(void**)&L->field = dlsym(h, P name)
    S(cinit, "ZstdCompressCtx_init"); S(cfree, "ZstdCompressCtx_free"); S(setLevel, "ZstdCompressCtx_setLevel0"); S(setChecksum, "ZstdCompressCtx_setChecksum0");
    S(cDirect, "ZstdCompressCtx_compressDirectByteBuffer0"); S(cArray, "ZstdCompressCtx_compressByteArray0");
    S(dinit, "ZstdDecompressCtx_init"); S(dfree, "ZstdDecompressCtx_free");
    S(dDirect, "ZstdDecompressCtx_decompressDirectByteBuffer0"); S(dArray, "ZstdDecompressCtx_decompressByteArray0");
    S(bound, "Zstd_compressBound"); S(isError, "Zstd_isError"); S(errName, "Zstd_getErrorName"); S(errCode, "Zstd_getErrorCode");
    S(cUnsafe, "Zstd_compressUnsafe"); S(dUnsafe, "Zstd_decompressUnsafe");
    S(setHashLog, "Zstd_setCompressionHashLog"); S(setChainLog, "Zstd_setCompressionChainLog");
    S(cBatch, "Zstd_compressBatch0"); S(dBatch, "Zstd_decompressBatch0");
    S(cBatchBegin, "Zstd_compressBatchBegin0"); S(dBatchBegin, "Zstd_decompressBatchBegin0"); S(batchFinish, "Zstd_batchFinish0");
    S(dictInit, "ZstdDictCompress_init"); S(dictInitDirect, "ZstdDictCompress_initDirect"); S(dictFree, "ZstdDictCompress_free");
    S(loadCDict, "ZstdCompressCtx_loadCDictFast0"); S(cBatchDict, "Zstd_compressBatchDict0");
    S(ddictInit, "ZstdDictDecompress_init"); S(ddictInitDirect, "ZstdDictDecompress_initDirect"); S(ddictFree, "ZstdDictDecompress_free"); S(loadDDict, "ZstdDecompressCtx_loadDDictFast0");
#undef S
    if (!L->cinit || !L->cDirect || !L->cArray || !L->dDirect || !L->dArray || !L->bound || !L->errName || !L->cUnsafe) { printf("%s: hot-path natives missing\n", path); return 0; }
    if (!L->setHashLog || !L->setChainLog) { printf("%s: setCompressionHashLog/ChainLog missing\n", path); return 0; }
    if (!isRef && (!L->cBatch || !L->dBatch || !L->cBatchDict || !L->cBatchBegin || !L->dBatchBegin || !L->batchFinish)) { printf("%s: batch natives missing\n", path); return 0; }
    if (!L->dictInit || !L->dictInitDirect || !L->dictFree || !L->loadCDict || !L->ddictInit || !L->ddictInitDirect || !L->ddictFree || !L->loadDDict) { printf("%s: ZstdDictCompress natives missing\n", path); return 0; }
    return 1;
}

/* deterministic test data: text-like / low-entropy / random, the classes of the benchmark generator */
static uint64_t g_x = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { g_x ^= g_x << 13; g_x ^= g_x >> 7; g_x ^= g_x << 17; return (uint32_t)(g_x >> 11); }
static void fill(char* p, jsize n, int cls) {
    static const char* words[8] = {"alpha ", "bravo ", "compress ", "zstandard ", "delta ", "wavefront ", "entropy ", "table "};
    jsize i = 0;
    if (cls == 0) { while (i < n) { const char* w = words[rnd() & 7]; while (*w && i < n) p[i++] = *w++; } }
    else if (cls == 1) { for (; i < n; i++) p[i] = (i > 64 && (rnd() & 7)) ? p[i - 1 - (rnd() & 63)] : (char)(rnd() & 15); }
    else { for (; i < n; i++) p[i] = (char)rnd(); }
}

static int g_checks, g_bad;
#define CHECK(cond, ...) do { g_checks++; if (!(cond)) { if (g_bad++ < 12) { printf("MISMATCH: "); printf(__VA_ARGS__); printf("\n"); } } } while (0)

int main(int argc, char** argv) {
    Lib R, G; JNIEnv* e = env();
    jsize const sizes[] = {0, 1, 17, 100, 4096, 20000, 65536, 131072};
    setvbuf(stdout, NULL, _IONBF, 0);
    if (argc < 3) { printf("usage: harness <ref-jni.so> <shim.so>\n"); return 2; }
#define STAGE(name) do { if (getenv("HARNESS_VERBOSE")) printf("stage: %s (checks so far %d, bad %d)\n", name, g_checks, g_bad); } while (0)
    memset(&R, 0, sizeof R); memset(&G, 0, sizeof G);
    if (!load(&R, argv[1], 1) || !load(&G, argv[2], 0)) return 2;

    STAGE("class Zstd helpers");
    /* class Zstd helpers */
    {   jlong const probes[] = {0, 1, 255, 4096, 65536, 131072, 1 << 20};
        for (unsigned i = 0; i < sizeof probes / sizeof *probes; i++) CHECK(R.bound(e, NULL, probes[i]) == G.bound(e, NULL, probes[i]), "compressBound(%lld)", (long long)probes[i]);
        jlong const codes[] = {-70, -72, -20, -10, -64, -22, -32, 5};
        for (unsigned i = 0; i < sizeof codes / sizeof *codes; i++) {
            CHECK(R.isError(e, NULL, codes[i]) == G.isError(e, NULL, codes[i]), "isError(%lld)", (long long)codes[i]);
            CHECK(R.errCode(e, NULL, codes[i]) == G.errCode(e, NULL, codes[i]), "getErrorCode(%lld)", (long long)codes[i]);
            CHECK(!strcmp(((Obj*)R.errName(e, NULL, codes[i]))->data, ((Obj*)G.errName(e, NULL, codes[i]))->data), "getErrorName(%lld)", (long long)codes[i]);
        }
    }
    STAGE("class Zstd: frame inspection and constants");
    /* decompressedSize / getFrameContentSize / findFrameCompressedSize / getDictIdFromFrame / getDictIdFromDict, byte[] and direct forms, and the constants:
     * host-side natives of the shim (no GPU, no bundled library), compared with the reference's on real frames, every truncation of their beginnings,
     * every single-bit change of their headers, hand-made headers (dictionary ids, 8-byte sizes, window descriptors, skippable and pre-1.0 magics),
     * magicless headers, noise, and direct-buffer ranges outside the buffer */
    {   typedef jlong (*insp_a5)(JNIEnv*, jclass, jbyteArray, jint, jint, jboolean); typedef jlong (*insp_a4)(JNIEnv*, jclass, jbyteArray, jint, jint);
        typedef jlong (*insp_a1)(JNIEnv*, jclass, jbyteArray); typedef jlong (*insp_b5)(JNIEnv*, jclass, jobject, jint, jint, jboolean);
        typedef jlong (*insp_b4)(JNIEnv*, jclass, jobject, jint, jint); typedef jlong (*insp_b1)(JNIEnv*, jclass, jobject);
        typedef jint (*const_i)(JNIEnv*, jclass); typedef jlong (*const_l)(JNIEnv*, jclass);
        static const char* ints[] = {"windowLogMin", "windowLogMax", "chainLogMin", "chainLogMax", "hashLogMin", "hashLogMax", "searchLogMin", "searchLogMax", "magicNumber",
                                     "blockSizeMax", "defaultCompressionLevel", "minCompressionLevel", "maxCompressionLevel"};
        static const char* errs[] = {"NoError", "Generic", "PrefixUnknown", "VersionUnsupported", "FrameParameterUnsupported", "FrameParameterWindowTooLarge", "CorruptionDetected",
                                     "ChecksumWrong", "DictionaryCorrupted", "DictionaryWrong", "DictionaryCreationFailed", "ParameterUnsupported", "ParameterOutOfBound",
                                     "TableLogTooLarge", "MaxSymbolValueTooLarge", "MaxSymbolValueTooSmall", "StageWrong", "InitMissing", "MemoryAllocation", "WorkSpaceTooSmall",
                                     "DstSizeTooSmall", "SrcSizeWrong", "DstBufferNull"};
        char nm[160];
        for (unsigned i = 0; i < sizeof ints / sizeof *ints; i++) {
            snprintf(nm, sizeof nm, P "Zstd_%s", ints[i]);
            const_i r = (const_i)dlsym(R.h, nm), g = (const_i)dlsym(G.h, nm);
            CHECK(r && g && r(e, NULL) == g(e, NULL), "Zstd.%s: ref %d ours %d", ints[i], r ? r(e, NULL) : -1, g ? g(e, NULL) : -1);
        }
        for (unsigned i = 0; i < sizeof errs / sizeof *errs; i++) {
            snprintf(nm, sizeof nm, P "Zstd_err%s", errs[i]);
            const_l r = (const_l)dlsym(R.h, nm), g = (const_l)dlsym(G.h, nm);
            CHECK(r && g && r(e, NULL) == g(e, NULL), "Zstd.err%s: ref %lld ours %lld", errs[i], r ? (long long)r(e, NULL) : -1, g ? (long long)g(e, NULL) : -1);
        }
#define II(T, name) T r_##name = (T)dlsym(R.h, P "Zstd_" #name), g_##name = (T)dlsym(G.h, P "Zstd_" #name)
        II(insp_a5, decompressedSize0); II(insp_a5, getFrameContentSize0); II(insp_a4, findFrameCompressedSize0); II(insp_a1, getDictIdFromFrame); II(insp_a1, getDictIdFromDict);
        II(insp_b5, decompressedDirectByteBufferSize); II(insp_b5, getDirectByteBufferFrameContentSize); II(insp_b4, findDirectByteBufferFrameCompressedSize);
        II(insp_b1, getDictIdFromFrameBuffer); II(insp_b4, getDictIdFromDictDirect);
#undef II
        CHECK(g_decompressedSize0 && g_getFrameContentSize0 && g_findFrameCompressedSize0 && g_getDictIdFromFrame && g_getDictIdFromDict && g_decompressedDirectByteBufferSize &&
              g_getDirectByteBufferFrameContentSize && g_findDirectByteBufferFrameCompressedSize && g_getDictIdFromFrameBuffer && g_getDictIdFromDictDirect, "inspection natives exported");
        /* the cases: (bytes, length) pairs collected in one pool */
        enum { POOL = 1 << 22, MAXCASE = 6000 };
        char* pool = (char*)malloc(POOL); size_t used = 0; size_t at[MAXCASE]; jsize len[MAXCASE]; int nCase = 0;
#define ADD(ptr, n_) do { if (nCase < MAXCASE && used + (size_t)(n_) + 32 <= POOL) { memcpy(pool + used, (ptr), (size_t)(n_)); memset(pool + used + (n_), 0xA5, 32); at[nCase] = used; len[nCase++] = (jsize)(n_); used += (size_t)(n_) + 32; } } while (0)
        {   jlong rc = R.cinit(e, NULL);
            jsize const fsz[] = {0, 1, 100, 255, 256, 300, 65535, 65536, 70000, 131072, 200000, 400000};
            for (unsigned si = 0; si < sizeof fsz / sizeof *fsz; si++) for (int ck = 0; ck < 2; ck++) {
                jsize const n = fsz[si], cap = (jsize)R.bound(e, NULL, n);
                Obj* src = mk(2, n); Obj* dst = mk(2, cap);
                fill(src->data, n, (int)(si % 3));
                R.setLevel(e, NULL, rc, 1 + (int)(si & 1)); R.setChecksum(e, NULL, rc, ck ? JNI_TRUE : JNI_FALSE);
                jlong const z = R.cArray(e, NULL, rc, (jbyteArray)dst, 0, cap, (jbyteArray)src, 0, n);
                if (z <= 0) { CHECK(0, "reference compress for the inspection cases"); continue; }
                ADD(dst->data, z);
                for (jsize t = 0; t < 24 && t < z; t++) ADD(dst->data, t);                      /* beginnings */
                ADD(dst->data, z - 1); ADD(dst->data, z - 3); if (z > 5) ADD(dst->data, z - 5);
                {   char two[1 << 12]; jsize const h = z < 2000 ? (jsize)z : 2000; memcpy(two, dst->data, (size_t)h); memcpy(two + h, dst->data, (size_t)h); ADD(two, 2 * h); }      /* a second frame (or noise) behind */
                if (si < 6) for (int bit = 0; bit < 14 * 8; bit++) { if (bit / 8 < z) { dst->data[bit / 8] ^= (char)(1 << (bit & 7)); ADD(dst->data, z < 600 ? z : 600); dst->data[bit / 8] ^= (char)(1 << (bit & 7)); } }
                ADD(dst->data + 4, z - 4);                                                      /* the same frame without its magic number: for the magicless reads */
                for (jsize t = 0; t < 16 && t + 4 < z; t++) ADD(dst->data + 4, t);
            }
            R.cfree(e, NULL, rc);
        }
        {   /* hand-made headers: every descriptor byte with the reserved bit clear and set, followed by enough bytes for the longest header and an empty last raw block */
            for (int fhd = 0; fhd < 256; fhd++) for (int wl = 0; wl < 3; wl++) {
                unsigned char h[40]; int k = 0;
                h[k++] = 0x28; h[k++] = 0xB5; h[k++] = 0x2F; h[k++] = 0xFD; h[k++] = (unsigned char)fhd;
                if (!((fhd >> 5) & 1)) h[k++] = (unsigned char)(wl == 0 ? 0x00 : (wl == 1 ? 0xAF : 0xB0));          /* window logs 10, 31 (+7/8), 32 */
                else if (wl) continue;
                {   static const int dsz[4] = {0, 1, 2, 4}, fsz2[4] = {0, 2, 4, 8};
                    int const d = dsz[fhd & 3], f = fsz2[fhd >> 6] + (((fhd >> 5) & 1) && !(fhd >> 6));
                    for (int i = 0; i < d; i++) h[k++] = (unsigned char)(0x11 * (i + 1) + fhd);
                    for (int i = 0; i < f; i++) h[k++] = (unsigned char)(i == f - 1 ? 0x01 : 0x80 + i); }
                h[k++] = 0x01; h[k++] = 0x00; h[k++] = 0x00;                                            /* raw, last, empty */
                if (fhd & 4) { h[k++] = 0x99; h[k++] = 0xE9; h[k++] = 0xD8; h[k++] = 0x51; }
                ADD(h, k); ADD(h, k - 1); ADD(h, k - 4); ADD(h + 4, k - 4);
            }
            {   static const unsigned char blocks[][12] = { {0x28,0xB5,0x2F,0xFD,0x20,0x05, 0x03,0x00,0x00, 'x', 0,0},          /* an RLE block of 0 (last) */
                                                            {0x28,0xB5,0x2F,0xFD,0x20,0x05, 0x07,0x00,0x00, 0,0,0},             /* the reserved block type */
                                                            {0x28,0xB5,0x2F,0xFD,0x20,0x05, 0x28,0x00,0x00, 1,2,3} };          /* raw 5, not last, and the input ends */
                for (unsigned i = 0; i < 3; i++) for (int t = 6; t <= 12; t++) ADD(blocks[i], t); }
            for (int v = 0; v < 16; v += 5) {                                                           /* skippable frames */
                unsigned char sk[64]; memset(sk, 0x33, sizeof sk);
                sk[0] = (unsigned char)(0x50 + v); sk[1] = 0x2A; sk[2] = 0x4D; sk[3] = 0x18;
                unsigned const szs[] = {0, 1, 40, 56, 57, 0xFFFFFFF7u, 0xFFFFFFF8u, 0xFFFFFFFFu};
                for (unsigned i = 0; i < sizeof szs / sizeof *szs; i++) {
                    sk[4] = (unsigned char)szs[i]; sk[5] = (unsigned char)(szs[i] >> 8); sk[6] = (unsigned char)(szs[i] >> 16); sk[7] = (unsigned char)(szs[i] >> 24);
                    ADD(sk, 64); ADD(sk, 8); ADD(sk, 7); ADD(sk, 4); ADD(sk, 3); ADD(sk, 1);
                }
            }
            for (unsigned m = 0xFD2FB520u; m <= 0xFD2FB52Au; m++) { unsigned char lg[24]; memset(lg, 0, sizeof lg); lg[0] = (unsigned char)m; lg[1] = 0xB5; lg[2] = 0x2F; lg[3] = 0xFD; ADD(lg, 24); ADD(lg, 4); }
            for (int i = 0; i < 300; i++) { char noise[48]; jsize const n = (jsize)(rnd() % 48); fill(noise, n, 2); ADD(noise, n); }
            {   unsigned char d[16] = {0x37, 0xA4, 0x30, 0xEC, 0x78, 0x56, 0x34, 0x12, 9, 9, 9, 9, 9, 9, 9, 9};             /* dictionaries: the magic and an id */
                for (int t = 0; t <= 16; t++) ADD(d, t);
                d[0] ^= 1; ADD(d, 16); }
        }
        for (int c = 0; c < nCase; c++) {
            jsize const n = len[c], off = 3;
            Obj* whole = mk(2, n); Obj* shifted = mk(2, n + off + 5); Obj* dir = mk(1, n + off + 5);
            memcpy(whole->data, pool + at[c], (size_t)n); memcpy(shifted->data + off, pool + at[c], (size_t)n + 5); memcpy(dir->data + off, pool + at[c], (size_t)n + 5);
            for (int ml = 0; ml < 2; ml++) {
                jboolean const m = ml ? JNI_TRUE : JNI_FALSE;
                jlong a = r_decompressedSize0(e, NULL, (jbyteArray)shifted, off, n, m), b = g_decompressedSize0(e, NULL, (jbyteArray)shifted, off, n, m);
                CHECK(a == b, "decompressedSize0 case %d (n=%d magicless=%d): ref %lld ours %lld", c, n, ml, (long long)a, (long long)b);
                a = r_getFrameContentSize0(e, NULL, (jbyteArray)shifted, off, n, m); b = g_getFrameContentSize0(e, NULL, (jbyteArray)shifted, off, n, m);
                CHECK(a == b, "getFrameContentSize0 case %d (n=%d magicless=%d): ref %lld ours %lld", c, n, ml, (long long)a, (long long)b);
                a = r_decompressedDirectByteBufferSize(e, NULL, dir, off, n, m); b = g_decompressedDirectByteBufferSize(e, NULL, dir, off, n, m);
                CHECK(a == b, "decompressedDirectByteBufferSize case %d (n=%d magicless=%d): ref %lld ours %lld", c, n, ml, (long long)a, (long long)b);
                a = r_getDirectByteBufferFrameContentSize(e, NULL, dir, off, n, m); b = g_getDirectByteBufferFrameContentSize(e, NULL, dir, off, n, m);
                CHECK(a == b, "getDirectByteBufferFrameContentSize case %d (n=%d magicless=%d): ref %lld ours %lld", c, n, ml, (long long)a, (long long)b);
            }
            {   jlong a = r_findFrameCompressedSize0(e, NULL, (jbyteArray)shifted, off, n), b = g_findFrameCompressedSize0(e, NULL, (jbyteArray)shifted, off, n);
                CHECK(a == b, "findFrameCompressedSize0 case %d (n=%d): ref %lld ours %lld", c, n, (long long)a, (long long)b);
                a = r_findDirectByteBufferFrameCompressedSize(e, NULL, dir, off, n); b = g_findDirectByteBufferFrameCompressedSize(e, NULL, dir, off, n);
                CHECK(a == b, "findDirectByteBufferFrameCompressedSize case %d (n=%d): ref %lld ours %lld", c, n, (long long)a, (long long)b);
                a = r_getDictIdFromFrame(e, NULL, (jbyteArray)whole); b = g_getDictIdFromFrame(e, NULL, (jbyteArray)whole);
                CHECK(a == b, "getDictIdFromFrame case %d (n=%d): ref %lld ours %lld", c, n, (long long)a, (long long)b);
                a = r_getDictIdFromDict(e, NULL, (jbyteArray)whole); b = g_getDictIdFromDict(e, NULL, (jbyteArray)whole);
                CHECK(a == b, "getDictIdFromDict case %d (n=%d): ref %lld ours %lld", c, n, (long long)a, (long long)b);
                a = r_getDictIdFromDictDirect(e, NULL, dir, off, n); b = g_getDictIdFromDictDirect(e, NULL, dir, off, n);
                CHECK(a == b, "getDictIdFromDictDirect case %d (n=%d): ref %lld ours %lld", c, n, (long long)a, (long long)b);
                {   Obj* exact = mk(1, n); memcpy(exact->data, pool + at[c], (size_t)n);
                    a = r_getDictIdFromFrameBuffer(e, NULL, exact); b = g_getDictIdFromFrameBuffer(e, NULL, exact);
                    CHECK(a == b, "getDictIdFromFrameBuffer case %d (n=%d): ref %lld ours %lld", c, n, (long long)a, (long long)b);
                    free(exact->data); free(exact); }
            }
            if (c < 40) {                                                                           /* ranges outside the direct buffer */
                jint const bad[][2] = {{-1, 4}, {0, -1}, {n + off + 5, 1}, {1, n + off + 5}, {n + off + 5, 0}, {0x7FFFFFFF, 0x7FFFFFFF}};
                for (unsigned i = 0; i < sizeof bad / sizeof *bad; i++) {
                    CHECK(r_findDirectByteBufferFrameCompressedSize(e, NULL, dir, bad[i][0], bad[i][1]) == g_findDirectByteBufferFrameCompressedSize(e, NULL, dir, bad[i][0], bad[i][1]), "findDirect... range %d/%d", bad[i][0], bad[i][1]);
                    CHECK(r_decompressedDirectByteBufferSize(e, NULL, dir, bad[i][0], bad[i][1], JNI_FALSE) == g_decompressedDirectByteBufferSize(e, NULL, dir, bad[i][0], bad[i][1], JNI_FALSE), "decompressedDirect... range %d/%d", bad[i][0], bad[i][1]);
                    CHECK(r_getDirectByteBufferFrameContentSize(e, NULL, dir, bad[i][0], bad[i][1], JNI_FALSE) == g_getDirectByteBufferFrameContentSize(e, NULL, dir, bad[i][0], bad[i][1], JNI_FALSE), "getDirect...ContentSize range %d/%d", bad[i][0], bad[i][1]);
                }
            }
            free(whole->data); free(whole); free(shifted->data); free(shifted); free(dir->data); free(dir);
        }
        printf("JNI-HARNESS INSPECTION cases=%d\n", nCase);
        free(pool);
#undef ADD
    }
    if (getenv("HARNESS_ONLY_HELPERS")) {
        if (g_bad) { printf("JNI-HARNESS FAILED bad=%d checks=%d\n", g_bad, g_checks); return 1; }
        printf("JNI-HARNESS OK checks=%d\n", g_checks);
        return 0;
    }
    /* ZstdCompressCtx / ZstdDecompressCtx one-shot natives, direct buffers and byte[] */
    int const maxLevel = getenv("HARNESS_MAX_LEVEL") ? atoi(getenv("HARNESS_MAX_LEVEL")) : 3;
    int const plainMax = getenv("HARNESS_PLAIN_MAX_LEVEL") ? atoi(getenv("HARNESS_PLAIN_MAX_LEVEL")) : maxLevel;   /* the one-shot natives without a dictionary: levels 4-8 too on the GPU */
    for (int level = 1; level <= plainMax; level++) for (int ck = 0; ck < 2; ck++) {
        jlong rc = R.cinit(e, NULL), gc = G.cinit(e, NULL), rd = R.dinit(e, NULL), gd = G.dinit(e, NULL);
        R.setLevel(e, NULL, rc, level); G.setLevel(e, NULL, gc, level);
        R.setChecksum(e, NULL, rc, ck ? JNI_TRUE : JNI_FALSE); G.setChecksum(e, NULL, gc, ck ? JNI_TRUE : JNI_FALSE);
        for (unsigned si = 0; si < sizeof sizes / sizeof *sizes; si++) for (int cls = 0; cls < 3; cls++) {
            jsize const n = sizes[si], off = 5, cap = (jsize)R.bound(e, NULL, n) + 40;
            for (int kind = 1; kind <= 2; kind++) {
                Obj* src = mk(kind, n + off + 3); Obj* rdst = mk(kind, cap); Obj* gdst = mk(kind, cap);
                fill(src->data + off, n, cls);
                jlong const rr = kind == 1 ? R.cDirect(e, NULL, rc, rdst, 7, cap - 7, src, off, n) : R.cArray(e, NULL, rc, (jbyteArray)rdst, 7, cap - 7, (jbyteArray)src, off, n);
                jlong const gr = kind == 1 ? G.cDirect(e, NULL, gc, gdst, 7, cap - 7, src, off, n) : G.cArray(e, NULL, gc, (jbyteArray)gdst, 7, cap - 7, (jbyteArray)src, off, n);
                CHECK(rr == gr, "compress L%d ck%d n=%d cls=%d kind=%d: ref %lld gpu %lld", level, ck, n, cls, kind, (long long)rr, (long long)gr);
                if (rr == gr && rr > 0) CHECK(!memcmp(rdst->data, gdst->data, (size_t)rr + 7), "compressed bytes L%d ck%d n=%d cls=%d kind=%d", level, ck, n, cls, kind);   /* (past the frame the reference may leave scratch bytes) */
                if (rr > 0) {       /* decompress the reference's frame with both */
                    Obj* rout = mk(kind, n + 9); Obj* gout = mk(kind, n + 9);
                    jlong const a = kind == 1 ? R.dDirect(e, NULL, rd, rout, 3, n, rdst, 7, (jint)rr) : R.dArray(e, NULL, rd, (jbyteArray)rout, 3, n, (jbyteArray)rdst, 7, (jint)rr);
                    jlong const b = kind == 1 ? G.dDirect(e, NULL, gd, gout, 3, n, rdst, 7, (jint)rr) : G.dArray(e, NULL, gd, (jbyteArray)gout, 3, n, (jbyteArray)rdst, 7, (jint)rr);
                    CHECK(a == b && a == n, "decompress n=%d cls=%d kind=%d: ref %lld gpu %lld", n, cls, kind, (long long)a, (long long)b);
                    CHECK(!memcmp(rout->data, gout->data, (size_t)n + 9) && !memcmp(gout->data + 3, src->data + off, (size_t)n), "decompressed bytes n=%d cls=%d kind=%d", n, cls, kind);
                    if (n > 0) {    /* destination one byte short, truncated source */
                        jlong const a2 = kind == 1 ? R.dDirect(e, NULL, rd, rout, 0, n - 1, rdst, 7, (jint)rr) : R.dArray(e, NULL, rd, (jbyteArray)rout, 0, n - 1, (jbyteArray)rdst, 7, (jint)rr);
                        jlong const b2 = kind == 1 ? G.dDirect(e, NULL, gd, gout, 0, n - 1, rdst, 7, (jint)rr) : G.dArray(e, NULL, gd, (jbyteArray)gout, 0, n - 1, (jbyteArray)rdst, 7, (jint)rr);
                        CHECK(a2 == b2, "decompress short dst n=%d kind=%d: ref %lld gpu %lld", n, kind, (long long)a2, (long long)b2);
                        jlong const a3 = kind == 1 ? R.dDirect(e, NULL, rd, rout, 0, n, rdst, 7, (jint)rr - 2) : R.dArray(e, NULL, rd, (jbyteArray)rout, 0, n, (jbyteArray)rdst, 7, (jint)rr - 2);
                        jlong const b3 = kind == 1 ? G.dDirect(e, NULL, gd, gout, 0, n, rdst, 7, (jint)rr - 2) : G.dArray(e, NULL, gd, (jbyteArray)gout, 0, n, (jbyteArray)rdst, 7, (jint)rr - 2);
                        CHECK(a3 == b3, "decompress truncated n=%d kind=%d: ref %lld gpu %lld", n, kind, (long long)a3, (long long)b3);
                    }
                }
            }
        }
        /* argument checks, in the reference's order (N/jni_fast_zstd.c:588-600, :617-623) */
        {   Obj* s = mk(1, 100); Obj* d = mk(1, 200); Obj* sa = mk(2, 100); Obj* da = mk(2, 200);
            fill(s->data, 100, 0); memcpy(sa->data, s->data, 100);
            CHECK(R.cDirect(e, NULL, rc, NULL, 0, 10, s, 0, 10) == G.cDirect(e, NULL, gc, NULL, 0, 10, s, 0, 10), "null dst");
            CHECK(R.cDirect(e, NULL, rc, d, 0, 10, NULL, 0, 10) == G.cDirect(e, NULL, gc, d, 0, 10, NULL, 0, 10), "null src");
            CHECK(R.cDirect(e, NULL, rc, d, -1, 10, s, 0, 10) == G.cDirect(e, NULL, gc, d, -1, 10, s, 0, 10), "negative dst offset");
            CHECK(R.cDirect(e, NULL, rc, d, 0, 10, s, -1, 10) == G.cDirect(e, NULL, gc, d, 0, 10, s, -1, 10), "negative src offset");
            CHECK(R.cDirect(e, NULL, rc, d, 0, 10, s, 0, -1) == G.cDirect(e, NULL, gc, d, 0, 10, s, 0, -1), "negative src size");
            CHECK(R.cDirect(e, NULL, rc, d, 150, 100, s, 0, 10) == G.cDirect(e, NULL, gc, d, 150, 100, s, 0, 10), "dst range");
            CHECK(R.cDirect(e, NULL, rc, d, 0, 100, s, 50, 60) == G.cDirect(e, NULL, gc, d, 0, 100, s, 50, 60), "src range");
            CHECK(R.cDirect(e, NULL, rc, d, 0, 5, s, 0, 100) == G.cDirect(e, NULL, gc, d, 0, 5, s, 0, 100), "dst too small");
            CHECK(R.cArray(e, NULL, rc, (jbyteArray)da, -1, 10, (jbyteArray)sa, 0, 10) == G.cArray(e, NULL, gc, (jbyteArray)da, -1, 10, (jbyteArray)sa, 0, 10), "array negative dst offset");
            CHECK(R.cArray(e, NULL, rc, (jbyteArray)da, 0, 100, (jbyteArray)sa, 50, 60) == G.cArray(e, NULL, gc, (jbyteArray)da, 0, 100, (jbyteArray)sa, 50, 60), "array src range");
            CHECK(R.cArray(e, NULL, rc, (jbyteArray)da, 150, 100, (jbyteArray)sa, 0, 10) == G.cArray(e, NULL, gc, (jbyteArray)da, 150, 100, (jbyteArray)sa, 0, 10), "array dst range");
            CHECK(R.cArray(e, NULL, rc, (jbyteArray)da, 0, 5, (jbyteArray)sa, 0, 100) == G.cArray(e, NULL, gc, (jbyteArray)da, 0, 5, (jbyteArray)sa, 0, 100), "array dst too small");
            CHECK(R.dDirect(e, NULL, rd, d, 0, 100, s, 0, 100) == G.dDirect(e, NULL, gd, d, 0, 100, s, 0, 100), "decompress garbage");
            CHECK(R.dDirect(e, NULL, rd, NULL, 0, 100, s, 0, 100) == G.dDirect(e, NULL, gd, NULL, 0, 100, s, 0, 100), "decompress null dst");
            CHECK(R.dArray(e, NULL, rd, (jbyteArray)da, 0, 100, (jbyteArray)sa, 90, 20) == G.dArray(e, NULL, gd, (jbyteArray)da, 0, 100, (jbyteArray)sa, 90, 20), "decompress array src range");
        }
        R.cfree(e, NULL, rc); G.cfree(e, NULL, gc); R.dfree(e, NULL, rd); G.dfree(e, NULL, gd);
    }
    STAGE("explicit table sizes");
    /* ZstdCompressCtx.setHashLog / setChainLog: with the level's own table sizes the shim's frames are the reference's PLAIN level 3 */
    if (maxLevel >= 3) {
        jlong rc = R.cinit(e, NULL), gc = G.cinit(e, NULL);
        R.setLevel(e, NULL, rc, 3); G.setLevel(e, NULL, gc, 3);
        CHECK(G.setHashLog(e, NULL, gc, 16) == 16 && G.setChainLog(e, NULL, gc, 15) == 15, "setCompressionHashLog/ChainLog on a shim context (ZSTD_CCtx_setParameter answers the value in force)");
        for (unsigned si = 0; si < sizeof sizes / sizeof *sizes; si++) for (int cls = 0; cls < 3; cls++) {
            jsize const n = sizes[si], cap = (jsize)R.bound(e, NULL, n) + 8;
            Obj* src = mk(1, n); Obj* rdst = mk(1, cap); Obj* gdst = mk(1, cap);
            fill(src->data, n, cls);
            jlong const rr = R.cDirect(e, NULL, rc, rdst, 0, cap, src, 0, n), gr = G.cDirect(e, NULL, gc, gdst, 0, cap, src, 0, n);
            CHECK(rr == gr && rr > 0 && !memcmp(rdst->data, gdst->data, (size_t)rr), "plain level 3 (16/15) n=%d cls=%d: ref %lld gpu %lld", n, cls, (long long)rr, (long long)gr);
        }
        /* setHashLog(14).setChainLog(13): the LDS-sized tables (wave-per-frame matcher / fused kernel on the GPU) = the reference given the same two */
        {   jint const gh = G.setHashLog(e, NULL, gc, 14), gl = G.setChainLog(e, NULL, gc, 13), rh = R.setHashLog(e, NULL, rc, 14), rl = R.setChainLog(e, NULL, rc, 13);
            CHECK(gh == rh && gl == rl && rh == 14 && rl == 13, "setCompressionHashLog/ChainLog (14/13): shim %d / %d, reference %d / %d", (int)gh, (int)gl, (int)rh, (int)rl); }
        for (unsigned si = 0; si < sizeof sizes / sizeof *sizes; si++) for (int cls = 0; cls < 3; cls++) {
            jsize const n = sizes[si], cap = (jsize)R.bound(e, NULL, n) + 8;
            Obj* src = mk(1, n); Obj* rdst = mk(1, cap); Obj* gdst = mk(1, cap);
            fill(src->data, n, cls);
            jlong const rr = R.cDirect(e, NULL, rc, rdst, 0, cap, src, 0, n), gr = G.cDirect(e, NULL, gc, gdst, 0, cap, src, 0, n);
            CHECK(rr == gr && rr > 0 && !memcmp(rdst->data, gdst->data, (size_t)rr), "level 3 with hashLog 14 / chainLog 13 n=%d cls=%d: ref %lld gpu %lld", n, cls, (long long)rr, (long long)gr);
        }
        R.cfree(e, NULL, rc); G.cfree(e, NULL, gc);
    }
    STAGE("raw-content dictionary");
    /* ZstdDictCompress + ZstdCompressCtx.loadDict (N/jni_fast_zstd.c:13-66, :325-336): a raw-content dictionary, sources inside the
     * attach range, byte[] and direct-buffer constructors */
    for (int level = 1; level <= maxLevel; level++) for (int direct = 0; direct < 2; direct++) {
        jsize const dlen = 20000; jsize const srcSizes[] = {0, 100, 1000, 4096, 8000};
        Obj* darr = mk(direct ? 1 : 2, dlen + 11); Obj* robj = mk(6, 0); Obj* gobj = mk(6, 0); Obj* rdobj = mk(6, 0); Obj* gdobj = mk(6, 0);
        jlong rc = R.cinit(e, NULL), gc = G.cinit(e, NULL), rd = R.dinit(e, NULL), gd = G.dinit(e, NULL);
        fill(darr->data + 11, dlen, 0);
        if (direct) { R.dictInitDirect(e, robj, darr, 11, dlen, level, 0); G.dictInitDirect(e, gobj, darr, 11, dlen, level, 0); }
        else { R.dictInit(e, robj, (jbyteArray)darr, 11, dlen, level); G.dictInit(e, gobj, (jbyteArray)darr, 11, dlen, level); }
        CHECK(robj->field != 0 && gobj->field != 0, "ZstdDictCompress init L%d direct=%d", level, direct);
        CHECK(R.loadCDict(e, NULL, rc, robj) == G.loadCDict(e, NULL, gc, gobj), "loadCDictFast0 L%d", level);
        if (direct) { R.ddictInitDirect(e, rdobj, darr, 11, dlen, 0); G.ddictInitDirect(e, gdobj, darr, 11, dlen, 0); }
        else { R.ddictInit(e, rdobj, (jbyteArray)darr, 11, dlen); G.ddictInit(e, gdobj, (jbyteArray)darr, 11, dlen); }
        CHECK(rdobj->field != 0 && gdobj->field != 0, "ZstdDictDecompress init direct=%d", direct);
        CHECK(R.loadDDict(e, NULL, rd, rdobj) == G.loadDDict(e, NULL, gd, gdobj), "loadDDictFast0");
        for (unsigned si = 0; si < sizeof srcSizes / sizeof *srcSizes; si++) for (int kind = 1; kind <= 2; kind++) {
            jsize const n = srcSizes[si], cap = (jsize)R.bound(e, NULL, n) + 16;
            Obj* src = mk(kind, n + 4); Obj* rdst = mk(kind, cap); Obj* gdst = mk(kind, cap);
            fill(src->data + 2, n, 0);
            jlong const rr = kind == 1 ? R.cDirect(e, NULL, rc, rdst, 3, cap - 3, src, 2, n) : R.cArray(e, NULL, rc, (jbyteArray)rdst, 3, cap - 3, (jbyteArray)src, 2, n);
            jlong const gr = kind == 1 ? G.cDirect(e, NULL, gc, gdst, 3, cap - 3, src, 2, n) : G.cArray(e, NULL, gc, (jbyteArray)gdst, 3, cap - 3, (jbyteArray)src, 2, n);
            CHECK(rr == gr && rr > 0 && !memcmp(rdst->data, gdst->data, (size_t)rr + 3), "dict compress L%d direct=%d n=%d kind=%d: ref %lld gpu %lld", level, direct, n, kind, (long long)rr, (long long)gr);
            if (rr > 0) {       /* and back, with the dictionary loaded on the decompress contexts */
                Obj* rout = mk(kind, n + 5); Obj* gout = mk(kind, n + 5);
                jlong const a = kind == 1 ? R.dDirect(e, NULL, rd, rout, 1, n, rdst, 3, (jint)rr) : R.dArray(e, NULL, rd, (jbyteArray)rout, 1, n, (jbyteArray)rdst, 3, (jint)rr);
                jlong const b = kind == 1 ? G.dDirect(e, NULL, gd, gout, 1, n, rdst, 3, (jint)rr) : G.dArray(e, NULL, gd, (jbyteArray)gout, 1, n, (jbyteArray)rdst, 3, (jint)rr);
                CHECK(a == b && a == n && !memcmp(gout->data + 1, src->data + 2, (size_t)n), "dict decompress L%d n=%d kind=%d: ref %lld gpu %lld", level, n, kind, (long long)a, (long long)b);
            }
        }
        CHECK(R.loadCDict(e, NULL, rc, NULL) == G.loadCDict(e, NULL, gc, NULL), "loadCDictFast0(null)");
        {   Obj* src = mk(1, 3000); Obj* rdst = mk(1, 4000); Obj* gdst = mk(1, 4000); fill(src->data, 3000, 0);
            R.setLevel(e, NULL, rc, 1); G.setLevel(e, NULL, gc, 1);
            jlong const rr = R.cDirect(e, NULL, rc, rdst, 0, 4000, src, 0, 3000), gr = G.cDirect(e, NULL, gc, gdst, 0, 4000, src, 0, 3000);
            CHECK(rr == gr && !memcmp(rdst->data, gdst->data, (size_t)rr), "compress after the dictionary was removed"); }
        if (!getenv("HARNESS_SKIP_BATCH")) {                      /* the dictionary batch native == the reference's per-buffer native */
            enum { ND = 100 }; Obj* srcs = mk(3, ND); Obj* dsts = mk(3, ND); Obj* res = mk(4, ND);
            srcs->elems = (Obj**)calloc(ND, sizeof(Obj*)); dsts->elems = (Obj**)calloc(ND, sizeof(Obj*));
            for (int i = 0; i < ND; i++) { jsize const n = (jsize)(rnd() % 8000); srcs->elems[i] = mk(1, n); fill(srcs->elems[i]->data, n, i % 3 == 2 ? 1 : 0); dsts->elems[i] = mk(1, (jsize)R.bound(e, NULL, n)); }
            CHECK(G.cBatchDict(e, NULL, (jobjectArray)srcs, (jobjectArray)dsts, (jlongArray)res, gobj, JNI_FALSE) == 0, "compressBatchDict0");
            R.loadCDict(e, NULL, rc, robj);
            for (int i = 0; i < ND; i++) {
                Obj* one = mk(1, dsts->elems[i]->len);
                jlong const rr = R.cDirect(e, NULL, rc, one, 0, one->len, srcs->elems[i], 0, srcs->elems[i]->len);
                CHECK(rr == ((jlong*)res->data)[i] && !memcmp(one->data, dsts->elems[i]->data, (size_t)rr), "dict batch buffer %d L%d: ref %lld gpu %lld", i, level, (long long)rr, (long long)((jlong*)res->data)[i]);
            }
        }
        {   /* the one-shot natives of class Zstd over the dictionary objects (N/jni_fast_zstd.c:133-244): Zstd.compress(dst, src, ZstdDictCompress) etc.;
             * sizes inside and beyond the attach range, both buffer kinds, the reference's argument checks */
            typedef jlong (*fda_fn)(JNIEnv*, jclass, jbyteArray, jint, jbyteArray, jint, jint, jobject);
            typedef jlong (*fdb_fn)(JNIEnv*, jclass, jobject, jint, jint, jobject, jint, jint, jobject);
            fda_fn rCA = (fda_fn)dlsym(R.h, P "Zstd_compressFastDict0"), gCA = (fda_fn)dlsym(G.h, P "Zstd_compressFastDict0");
            fda_fn rDA = (fda_fn)dlsym(R.h, P "Zstd_decompressFastDict0"), gDA = (fda_fn)dlsym(G.h, P "Zstd_decompressFastDict0");
            fdb_fn rCB = (fdb_fn)dlsym(R.h, P "Zstd_compressDirectByteBufferFastDict0"), gCB = (fdb_fn)dlsym(G.h, P "Zstd_compressDirectByteBufferFastDict0");
            fdb_fn rDB = (fdb_fn)dlsym(R.h, P "Zstd_decompressDirectByteBufferFastDict0"), gDB = (fdb_fn)dlsym(G.h, P "Zstd_decompressDirectByteBufferFastDict0");
            jsize const fdSizes[] = {0, 1, 300, 5000, 8192, 20000, 70000};
            CHECK(rCA && gCA && rDA && gDA && rCB && gCB && rDB && gDB, "FastDict0 natives exported");
            for (unsigned si = 0; si < sizeof fdSizes / sizeof *fdSizes && gCA && gDA && gCB && gDB; si++) for (int kind = 1; kind <= 2; kind++) {
                jsize const n = fdSizes[si], cap = (jsize)R.bound(e, NULL, n) + 16;
                Obj* src = mk(kind, n + 4); Obj* rdst = mk(kind, cap); Obj* gdst = mk(kind, cap); Obj* rout = mk(kind, n + 5); Obj* gout = mk(kind, n + 5);
                fill(src->data + 2, n, si & 1);
                jlong const rr = kind == 1 ? rCB(e, NULL, rdst, 3, cap - 3, src, 2, n, robj) : rCA(e, NULL, (jbyteArray)rdst, 3, (jbyteArray)src, 2, n, robj);
                jlong const gr = kind == 1 ? gCB(e, NULL, gdst, 3, cap - 3, src, 2, n, gobj) : gCA(e, NULL, (jbyteArray)gdst, 3, (jbyteArray)src, 2, n, gobj);
                CHECK(rr == gr && rr > 0 && !memcmp(rdst->data, gdst->data, (size_t)rr + 3), "compressFastDict0 L%d n=%d kind=%d: ref %lld gpu %lld", level, n, kind, (long long)rr, (long long)gr);
                if (rr > 0) {
                    jlong const a = kind == 1 ? rDB(e, NULL, rout, 1, n, rdst, 3, (jint)rr, rdobj) : rDA(e, NULL, (jbyteArray)rout, 1, (jbyteArray)rdst, 3, (jint)rr, rdobj);
                    jlong const b = kind == 1 ? gDB(e, NULL, gout, 1, n, rdst, 3, (jint)rr, gdobj) : gDA(e, NULL, (jbyteArray)gout, 1, (jbyteArray)rdst, 3, (jint)rr, gdobj);
                    CHECK(a == b && a == n && !memcmp(gout->data + 1, src->data + 2, (size_t)n), "decompressFastDict0 L%d n=%d kind=%d: ref %lld gpu %lld", level, n, kind, (long long)a, (long long)b);
                    if (n > 10) {
                        jlong const a2 = kind == 1 ? rDB(e, NULL, rout, 1, n - 4, rdst, 3, (jint)rr, rdobj) : rDA(e, NULL, (jbyteArray)rout, 9, (jbyteArray)rdst, 3, (jint)rr, rdobj);
                        jlong const b2 = kind == 1 ? gDB(e, NULL, gout, 1, n - 4, rdst, 3, (jint)rr, gdobj) : gDA(e, NULL, (jbyteArray)gout, 9, (jbyteArray)rdst, 3, (jint)rr, gdobj);
                        CHECK(a2 == b2, "decompressFastDict0 short destination n=%d kind=%d: ref %lld gpu %lld", n, kind, (long long)a2, (long long)b2);
                    }
                }
            }
            if (gCA && gDA && gCB && gDB) {
                Obj* s = mk(2, 100); Obj* d = mk(2, 200); Obj* sb = mk(1, 100); Obj* db = mk(1, 200); Obj* zero = mk(6, 0);
                CHECK(rCA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 0, 100, NULL) == gCA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 0, 100, NULL), "compressFastDict0 null dictionary");
                CHECK(rCA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 0, 100, zero) == gCA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 0, 100, zero), "compressFastDict0 closed dictionary");
                CHECK(rCA(e, NULL, NULL, 0, (jbyteArray)s, 0, 100, robj) == gCA(e, NULL, NULL, 0, (jbyteArray)s, 0, 100, gobj), "compressFastDict0 null dst");
                CHECK(rCA(e, NULL, (jbyteArray)d, 0, NULL, 0, 100, robj) == gCA(e, NULL, (jbyteArray)d, 0, NULL, 0, 100, gobj), "compressFastDict0 null src");
                CHECK(rCA(e, NULL, (jbyteArray)d, -1, (jbyteArray)s, 0, 100, robj) == gCA(e, NULL, (jbyteArray)d, -1, (jbyteArray)s, 0, 100, gobj), "compressFastDict0 negative dst offset");
                CHECK(rCA(e, NULL, (jbyteArray)d, 201, (jbyteArray)s, 0, 100, robj) == gCA(e, NULL, (jbyteArray)d, 201, (jbyteArray)s, 0, 100, gobj), "compressFastDict0 dst offset beyond the array");
                CHECK(rCA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 50, 60, robj) == gCA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 50, 60, gobj), "compressFastDict0 src range");
                CHECK(rCA(e, NULL, (jbyteArray)d, 195, (jbyteArray)s, 0, 100, robj) == gCA(e, NULL, (jbyteArray)d, 195, (jbyteArray)s, 0, 100, gobj), "compressFastDict0 dst too small");
                CHECK(rDA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 0, 100, NULL) == gDA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 0, 100, NULL), "decompressFastDict0 null dictionary");
                CHECK(rDA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 0, 100, rdobj) == gDA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 0, 100, gdobj), "decompressFastDict0 garbage");
                CHECK(rDA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 0, -1, rdobj) == gDA(e, NULL, (jbyteArray)d, 0, (jbyteArray)s, 0, -1, gdobj), "decompressFastDict0 negative length");
                CHECK(rCB(e, NULL, db, 0, 200, sb, 0, 100, NULL) == gCB(e, NULL, db, 0, 200, sb, 0, 100, NULL), "compressDirectByteBufferFastDict0 null dictionary");
                CHECK(rCB(e, NULL, NULL, 0, 200, sb, 0, 100, robj) == gCB(e, NULL, NULL, 0, 200, sb, 0, 100, gobj), "compressDirectByteBufferFastDict0 null dst");
                CHECK(rCB(e, NULL, db, 0, 200, sb, -1, 100, robj) == gCB(e, NULL, db, 0, 200, sb, -1, 100, gobj), "compressDirectByteBufferFastDict0 negative src offset");
                CHECK(rCB(e, NULL, db, 0, 5, sb, 0, 100, robj) == gCB(e, NULL, db, 0, 5, sb, 0, 100, gobj), "compressDirectByteBufferFastDict0 dst too small");
                CHECK(rDB(e, NULL, db, 0, 200, sb, 0, 100, rdobj) == gDB(e, NULL, db, 0, 200, sb, 0, 100, gdobj), "decompressDirectByteBufferFastDict0 garbage");
                CHECK(rDB(e, NULL, db, 0, 200, NULL, 0, 100, rdobj) == gDB(e, NULL, db, 0, 200, NULL, 0, 100, gdobj), "decompressDirectByteBufferFastDict0 null src");
            }
        }
        {   /* without the dictionary the frame of a non-empty source no longer decodes: same error from both */
            Obj* src = mk(1, 1000); Obj* fr = mk(1, 2000); Obj* o1 = mk(1, 1000); Obj* o2 = mk(1, 1000); fill(src->data, 1000, 0);
            R.loadCDict(e, NULL, rc, robj);
            jlong const rr = R.cDirect(e, NULL, rc, fr, 0, 2000, src, 0, 1000);
            CHECK(R.loadDDict(e, NULL, rd, NULL) == G.loadDDict(e, NULL, gd, NULL), "loadDDictFast0(null)");
            CHECK(R.dDirect(e, NULL, rd, o1, 0, 1000, fr, 0, (jint)rr) == G.dDirect(e, NULL, gd, o2, 0, 1000, fr, 0, (jint)rr), "decompress a dictionary frame without the dictionary"); }
        R.cfree(e, NULL, rc); G.cfree(e, NULL, gc); R.dfree(e, NULL, rd); G.dfree(e, NULL, gd);
        R.dictFree(e, robj); G.dictFree(e, gobj); R.ddictFree(e, rdobj); G.ddictFree(e, gdobj);
    }
    /* Zstd.compressUnsafe / decompressUnsafe (levels 1-2: the reference cannot be given the level-3 table sizes here) */
    for (int level = 1; level <= 2; level++) for (int ck = 0; ck < 2; ck++) {
        jsize const n = 50000; jsize const cap = (jsize)R.bound(e, NULL, n);
        char* s = (char*)malloc(n); char* a = (char*)calloc(cap, 1); char* b = (char*)calloc(cap, 1); char* o = (char*)malloc(n);
        fill(s, n, 0);
        jlong const rr = R.cUnsafe(e, NULL, (jlong)(intptr_t)a, cap, (jlong)(intptr_t)s, n, level, ck ? JNI_TRUE : JNI_FALSE);
        jlong const gr = G.cUnsafe(e, NULL, (jlong)(intptr_t)b, cap, (jlong)(intptr_t)s, n, level, ck ? JNI_TRUE : JNI_FALSE);
        CHECK(rr == gr && rr > 0 && !memcmp(a, b, (size_t)rr), "compressUnsafe L%d ck%d: ref %lld gpu %lld", level, ck, (long long)rr, (long long)gr);
        CHECK(G.dUnsafe(e, NULL, (jlong)(intptr_t)o, n, (jlong)(intptr_t)a, rr) == n && !memcmp(o, s, n), "decompressUnsafe L%d", level);
        free(s); free(a); free(b); free(o);
    }
    /* the batch natives: 300 direct buffers at once == the per-buffer native of the reference, buffer by buffer */
    if (!getenv("HARNESS_SKIP_BATCH")) {   enum { NB = 300 }; jlong rc = R.cinit(e, NULL); R.setLevel(e, NULL, rc, 1);
        Obj* srcs = mk(3, NB); Obj* dsts = mk(3, NB); Obj* res = mk(4, NB); Obj* outs = mk(3, NB); Obj* res2 = mk(4, NB);
        srcs->elems = (Obj**)calloc(NB, sizeof(Obj*)); dsts->elems = (Obj**)calloc(NB, sizeof(Obj*)); outs->elems = (Obj**)calloc(NB, sizeof(Obj*));
        for (int i = 0; i < NB; i++) { jsize const n = (jsize)(rnd() % 70000); srcs->elems[i] = mk(1, n); fill(srcs->elems[i]->data, n, i % 3); dsts->elems[i] = mk(1, (jsize)R.bound(e, NULL, n)); outs->elems[i] = mk(1, n); }
        jlong const r = G.cBatch(e, NULL, (jobjectArray)srcs, (jobjectArray)dsts, (jlongArray)res, 1, JNI_FALSE);
        CHECK(r == 0, "compressBatch0 returned %lld", (long long)r);
        Obj* fr = mk(3, NB); fr->elems = (Obj**)calloc(NB, sizeof(Obj*));
        for (int i = 0; i < NB; i++) {
            Obj* one = mk(1, dsts->elems[i]->len);
            jlong const rr = R.cDirect(e, NULL, rc, one, 0, one->len, srcs->elems[i], 0, srcs->elems[i]->len);
            jlong const gr = ((jlong*)res->data)[i];
            CHECK(rr == gr && !memcmp(one->data, dsts->elems[i]->data, (size_t)rr), "batch buffer %d: ref %lld gpu %lld", i, (long long)rr, (long long)gr);
            fr->elems[i] = mk(1, (jsize)gr); memcpy(fr->elems[i]->data, dsts->elems[i]->data, (size_t)gr);
        }
        jlong const r2 = G.dBatch(e, NULL, (jobjectArray)fr, (jobjectArray)outs, (jlongArray)res2);
        CHECK(r2 == 0, "decompressBatch0 returned %lld", (long long)r2);
        for (int i = 0; i < NB; i++) CHECK(((jlong*)res2->data)[i] == srcs->elems[i]->len && !memcmp(outs->elems[i]->data, srcs->elems[i]->data, (size_t)srcs->elems[i]->len), "batch round trip %d", i);
        /* the asynchronous natives (round 6): THREE compress jobs begun before any is finished (two staging slots: the third waits for one), finished in order — the same
         * frames as the blocking native; then two decompress jobs in flight; global references held per job and released by batchFinish0; the refusals */
        {   enum { NJ = 3 }; Obj* d2[NJ]; Obj* r3[NJ]; jlong job[NJ];
            int const g0 = g_globals;
            for (int k = 0; k < NJ; k++) {
                d2[k] = mk(3, NB); d2[k]->elems = (Obj**)calloc(NB, sizeof(Obj*)); r3[k] = mk(4, NB);
                for (int i = 0; i < NB; i++) d2[k]->elems[i] = mk(1, dsts->elems[i]->len);
                job[k] = G.cBatchBegin(e, NULL, (jobjectArray)srcs, (jobjectArray)d2[k], 1, JNI_FALSE);
                CHECK(job[k] > 0, "compressBatchBegin0 job %d: %lld", k, (long long)job[k]);
            }
            CHECK(g_globals == g0 + 2 * NJ, "global references held by %d jobs: %d", NJ, g_globals - g0);
            for (int k = 0; k < NJ; k++) {
                jlong const fr3 = job[k] > 0 ? G.batchFinish(e, NULL, job[k], (jlongArray)r3[k]) : -1;
                CHECK(fr3 == 0, "batchFinish0 job %d: %lld", k, (long long)fr3);
                for (int i = 0; i < NB; i++) CHECK(((jlong*)r3[k]->data)[i] == ((jlong*)res->data)[i] && !memcmp(d2[k]->elems[i]->data, dsts->elems[i]->data, (size_t)((jlong*)res->data)[i]), "asynchronous job %d buffer %d differs from the blocking native", k, i);
            }
            CHECK(g_globals == g0, "global references released: %d left", g_globals - g0);
            {   Obj* o2[2]; Obj* r4[2]; jlong dj[2];
                for (int k = 0; k < 2; k++) { o2[k] = mk(3, NB); o2[k]->elems = (Obj**)calloc(NB, sizeof(Obj*)); r4[k] = mk(4, NB); for (int i = 0; i < NB; i++) o2[k]->elems[i] = mk(1, srcs->elems[i]->len);
                    dj[k] = G.dBatchBegin(e, NULL, (jobjectArray)fr, (jobjectArray)o2[k]); CHECK(dj[k] > 0, "decompressBatchBegin0 job %d: %lld", k, (long long)dj[k]); }
                for (int k = 0; k < 2; k++) { CHECK(dj[k] > 0 && G.batchFinish(e, NULL, dj[k], (jlongArray)r4[k]) == 0, "batchFinish0 of decompress job %d", k);
                    for (int i = 0; i < NB; i++) CHECK(((jlong*)r4[k]->data)[i] == srcs->elems[i]->len && !memcmp(o2[k]->elems[i]->data, srcs->elems[i]->data, (size_t)srcs->elems[i]->len), "asynchronous round trip job %d buffer %d", k, i); }
            }
            /* refusals: what the blocking natives answer, as Begin's return (<= 0: nothing begun); a results array too short still finishes and frees the job */
            CHECK(G.cBatchBegin(e, NULL, NULL, (jobjectArray)dsts, 1, JNI_FALSE) == -72, "compressBatchBegin0(null srcs)");
            CHECK(G.cBatchBegin(e, NULL, (jobjectArray)srcs, NULL, 1, JNI_FALSE) == -70, "compressBatchBegin0(null dsts)");
            {   Obj* few = mk(3, 2); few->elems = (Obj**)calloc(2, sizeof(Obj*)); few->elems[0] = mk(1, 10); few->elems[1] = mk(1, 10);
                CHECK(G.cBatchBegin(e, NULL, (jobjectArray)srcs, (jobjectArray)few, 1, JNI_FALSE) == -72, "compressBatchBegin0(array lengths differ)");
                Obj* holes = mk(3, 2); holes->elems = (Obj**)calloc(2, sizeof(Obj*)); holes->elems[0] = mk(1, 10);
                CHECK(G.cBatchBegin(e, NULL, (jobjectArray)holes, (jobjectArray)few, 1, JNI_FALSE) == -72, "compressBatchBegin0(null element)");
                CHECK(G.dBatchBegin(e, NULL, (jobjectArray)few, (jobjectArray)holes) == -70, "decompressBatchBegin0(null destination element)");
                Obj* shortRes = mk(4, 1);
                jlong const j2 = G.cBatchBegin(e, NULL, (jobjectArray)few, (jobjectArray)few, 1, JNI_FALSE);      /* (10-byte destinations: per-buffer dstSize_tooSmall, the call itself is fine) */
                CHECK(j2 > 0 && G.batchFinish(e, NULL, j2, (jlongArray)shortRes) == -70, "batchFinish0(results too short)");
                Obj* res5 = mk(4, 2); jlong const j3 = G.cBatchBegin(e, NULL, (jobjectArray)few, (jobjectArray)few, 1, JNI_FALSE);
                CHECK(j3 > 0 && G.batchFinish(e, NULL, j3, (jlongArray)res5) == 0 && ((jlong*)res5->data)[0] == -70 && ((jlong*)res5->data)[1] == -70, "per-buffer codes of an asynchronous job");
                Obj* none = mk(3, 0); Obj* res0 = mk(4, 0); jlong const j0 = G.cBatchBegin(e, NULL, (jobjectArray)none, (jobjectArray)none, 1, JNI_FALSE);
                CHECK(j0 > 0 && G.batchFinish(e, NULL, j0, (jlongArray)res0) == 0, "an empty asynchronous batch");
                CHECK(G.batchFinish(e, NULL, 0, (jlongArray)res5) == -72 && G.batchFinish(e, NULL, -64, (jlongArray)res5) == -72, "batchFinish0 of no job");
            }
            CHECK(g_globals == g0, "global references after the refusals: %d left", g_globals - g0);
        }
        R.cfree(e, NULL, rc);
    }
    STAGE("frame parameters");
    /* ---- the natives that shape or reuse a context beyond level + checksum (N/jni_fast_zstd.c:296-390, :686-716, :832-905) ---- */
#define SYM(L, T, name) ((T)dlsym((L).h, P name))
    typedef void (*setflag_fn)(JNIEnv*, jclass, jlong, jboolean);
    typedef jlong (*ctx1_fn)(JNIEnv*, jclass, jlong);
    typedef jlong (*loadbytes_fn)(JNIEnv*, jclass, jlong, jbyteArray);
    typedef jlong (*buf_fn)(JNIEnv*, jclass, jlong, jobject, jint, jint, jobject, jint, jint);
    {   setflag_fn rCS = SYM(R, setflag_fn, "ZstdCompressCtx_setContentSize0"), gCS = SYM(G, setflag_fn, "ZstdCompressCtx_setContentSize0");
        ctx1_fn rReset = SYM(R, ctx1_fn, "ZstdCompressCtx_reset0"), gReset = SYM(G, ctx1_fn, "ZstdCompressCtx_reset0");
        CHECK(rCS && gCS && rReset && gReset, "setContentSize0 / reset0 exported");
        for (int level = 1; level <= maxLevel && gCS && gReset; level += 2) {
            jlong rc = R.cinit(e, NULL), gc = G.cinit(e, NULL);
            R.setLevel(e, NULL, rc, level); G.setLevel(e, NULL, gc, level);
            for (int pass = 0; pass < 3; pass++) {           /* content size off, on again, then after reset0 */
                if (pass == 0) { rCS(e, NULL, rc, JNI_FALSE); gCS(e, NULL, gc, JNI_FALSE); }
                if (pass == 1) { rCS(e, NULL, rc, JNI_TRUE); gCS(e, NULL, gc, JNI_TRUE); R.setChecksum(e, NULL, rc, JNI_TRUE); G.setChecksum(e, NULL, gc, JNI_TRUE); }
                if (pass == 2) { CHECK(rReset(e, NULL, rc) == gReset(e, NULL, gc), "reset0 result"); R.setLevel(e, NULL, rc, 1); G.setLevel(e, NULL, gc, 1); }   /* reset dropped the checksum flag */
                for (unsigned si = 0; si < sizeof sizes / sizeof *sizes; si++) {
                    jsize const n = sizes[si], cap = (jsize)R.bound(e, NULL, n) + 8;
                    Obj* src = mk(1, n); Obj* rdst = mk(1, cap); Obj* gdst = mk(1, cap);
                    fill(src->data, n, (int)(si % 3));
                    jlong const rr = R.cDirect(e, NULL, rc, rdst, 0, cap, src, 0, n), gr = G.cDirect(e, NULL, gc, gdst, 0, cap, src, 0, n);
                    CHECK(rr == gr && rr > 0 && !memcmp(rdst->data, gdst->data, (size_t)rr), "frame parameters pass %d L%d n=%d: ref %lld gpu %lld", pass, level, n, (long long)rr, (long long)gr);
                    if (rr > 0 && pass == 0) {               /* a frame without content size decodes through the shim too */
                        jlong dd = G.dinit(e, NULL); Obj* out = mk(1, n + 1);
                        CHECK(G.dDirect(e, NULL, dd, out, 0, n, rdst, 0, (jint)rr) == n && !memcmp(out->data, src->data, (size_t)n), "decode frame without content size n=%d", n);
                        G.dfree(e, NULL, dd);
                    }
                }
            }
            R.cfree(e, NULL, rc); G.cfree(e, NULL, gc);
        }
    }
    STAGE("zstd-format dictionary");
    /* a zstd-format dictionary (with an ID): setDictID0(false), loadDDict0(byte[]), loadCDict0(byte[]), the two mixed decompress natives */
    if (getenv("HARNESS_DICT_FILE")) {
        FILE* fp = fopen(getenv("HARNESS_DICT_FILE"), "rb"); Obj* darr = mk(2, 200000); jsize dlen = 0;
        if (fp) { dlen = (jsize)fread(darr->data, 1, 200000, fp); fclose(fp); }
        darr->len = dlen;
        CHECK(dlen > 100, "dictionary file");
        setflag_fn rDI = SYM(R, setflag_fn, "ZstdCompressCtx_setDictID0"), gDI = SYM(G, setflag_fn, "ZstdCompressCtx_setDictID0");
        loadbytes_fn rLD = SYM(R, loadbytes_fn, "ZstdDecompressCtx_loadDDict0"), gLD = SYM(G, loadbytes_fn, "ZstdDecompressCtx_loadDDict0");
        loadbytes_fn rLC = SYM(R, loadbytes_fn, "ZstdCompressCtx_loadCDict0"), gLC = SYM(G, loadbytes_fn, "ZstdCompressCtx_loadCDict0");
        buf_fn rA2D = SYM(R, buf_fn, "ZstdDecompressCtx_decompressByteArrayToDirectByteBuffer0"), gA2D = SYM(G, buf_fn, "ZstdDecompressCtx_decompressByteArrayToDirectByteBuffer0");
        buf_fn rD2A = SYM(R, buf_fn, "ZstdDecompressCtx_decompressDirectByteBufferToByteArray0"), gD2A = SYM(G, buf_fn, "ZstdDecompressCtx_decompressDirectByteBufferToByteArray0");
        ctx1_fn rDReset = SYM(R, ctx1_fn, "ZstdDecompressCtx_reset0"), gDReset = SYM(G, ctx1_fn, "ZstdDecompressCtx_reset0");
        CHECK(gDI && gLD && gLC && gA2D && gD2A && gDReset, "setDictID0 / loadDDict0 / loadCDict0 / mixed decompress natives / reset0 exported");
        for (int level = 1; level <= maxLevel && gDI && gLD && gA2D && gD2A; level += 2) {
            Obj* robj = mk(6, 0); Obj* gobj = mk(6, 0);
            jlong rc = R.cinit(e, NULL), gc = G.cinit(e, NULL), rd = R.dinit(e, NULL), gd = G.dinit(e, NULL);
            R.dictInit(e, robj, (jbyteArray)darr, 0, dlen, level); G.dictInit(e, gobj, (jbyteArray)darr, 0, dlen, level);
            CHECK(R.loadCDict(e, NULL, rc, robj) == G.loadCDict(e, NULL, gc, gobj), "loadCDictFast0 (zstd-format dictionary)");
            CHECK(rLD(e, NULL, rd, (jbyteArray)darr) == gLD(e, NULL, gd, (jbyteArray)darr), "loadDDict0");
            for (int noID = 0; noID < 2; noID++) {
                jsize const srcSizes[] = {0, 300, 3000, 8000};
                rDI(e, NULL, rc, noID ? JNI_FALSE : JNI_TRUE); gDI(e, NULL, gc, noID ? JNI_FALSE : JNI_TRUE);
                for (unsigned si = 0; si < sizeof srcSizes / sizeof *srcSizes; si++) {
                    jsize const n = srcSizes[si], cap = (jsize)R.bound(e, NULL, n) + 16;
                    Obj* src = mk(1, n); Obj* rdst = mk(1, cap); Obj* gdst = mk(1, cap);
                    fill(src->data, n, 0);
                    jlong const rr = R.cDirect(e, NULL, rc, rdst, 0, cap, src, 0, n), gr = G.cDirect(e, NULL, gc, gdst, 0, cap, src, 0, n);
                    CHECK(rr == gr && rr > 0 && !memcmp(rdst->data, gdst->data, (size_t)rr), "dictID flag %d L%d n=%d: ref %lld gpu %lld", !noID, level, n, (long long)rr, (long long)gr);
                    if (rr > 0) {                               /* back through the byte[]-dictionary contexts, by the two mixed natives */
                        Obj* farr = mk(2, (jsize)rr + 4); Obj* out1 = mk(1, n + 3); Obj* out2 = mk(1, n + 3); Obj* oarr1 = mk(2, n + 3); Obj* oarr2 = mk(2, n + 3);
                        memcpy(farr->data + 2, rdst->data, (size_t)rr);
                        jlong const a = rA2D(e, NULL, rd, out1, 1, n, farr, 2, (jint)rr), b = gA2D(e, NULL, gd, out2, 1, n, farr, 2, (jint)rr);
                        CHECK(a == b && a == n && !memcmp(out2->data + 1, src->data, (size_t)n), "decompressByteArrayToDirectByteBuffer0 n=%d: ref %lld gpu %lld", n, (long long)a, (long long)b);
                        jlong const c = rD2A(e, NULL, rd, oarr1, 2, n, rdst, 0, (jint)rr), d = gD2A(e, NULL, gd, oarr2, 2, n, rdst, 0, (jint)rr);
                        CHECK(c == d && c == n && !memcmp(oarr2->data + 2, src->data, (size_t)n), "decompressDirectByteBufferToByteArray0 n=%d: ref %lld gpu %lld", n, (long long)c, (long long)d);
                        CHECK(rA2D(e, NULL, rd, out1, 1, n, farr, 2, (jint)rr + 9) == gA2D(e, NULL, gd, out2, 1, n, farr, 2, (jint)rr + 9), "mixed native: src range");
                        CHECK(rD2A(e, NULL, rd, oarr1, 5, n, rdst, 0, (jint)rr) == gD2A(e, NULL, gd, oarr2, 5, n, rdst, 0, (jint)rr), "mixed native: dst range");
                        CHECK(rA2D(e, NULL, rd, NULL, 1, n, farr, 2, (jint)rr) == gA2D(e, NULL, gd, NULL, 1, n, farr, 2, (jint)rr), "mixed native: null dst");
                    }
                }
            }
            /* the decompress context after reset0 has no dictionary any more: same answer from both */
            {   Obj* src = mk(1, 2000); Obj* fr = mk(1, 3000); Obj* o1 = mk(1, 2000); Obj* o2 = mk(1, 2000); fill(src->data, 2000, 0);
                jlong const rr = R.cDirect(e, NULL, rc, fr, 0, 3000, src, 0, 2000);
                CHECK(rDReset(e, NULL, rd) == gDReset(e, NULL, gd), "ZstdDecompressCtx.reset0");
                CHECK(R.dDirect(e, NULL, rd, o1, 0, 2000, fr, 0, (jint)rr) == G.dDirect(e, NULL, gd, o2, 0, 2000, fr, 0, (jint)rr), "dictionary frame after reset0"); }
            /* a byte[] dictionary on the compress side (ZSTD_CCtx_loadDictionary): digested at the first compress call, attach and copy ranges */
            {   jlong const a = rLC(e, NULL, rc, (jbyteArray)darr), b = gLC(e, NULL, gc, (jbyteArray)darr);
                jsize const tsizes[] = {0, 17, 3000, 8192, 20000, 65536, 131071};
                CHECK(a == b, "loadCDict0: ref %lld shim %lld", (long long)a, (long long)b);
                for (unsigned ti = 0; ti < sizeof tsizes / sizeof *tsizes; ti++) for (int cls2 = 0; cls2 < 2; cls2++) {
                    jsize const n2 = tsizes[ti], cap2 = (jsize)R.bound(e, NULL, n2) + 16;
                    Obj* src = mk(1, n2 + 1); Obj* rdst = mk(1, cap2); Obj* gdst = mk(1, cap2); fill(src->data, n2, cls2);
                    jlong const rr = R.cDirect(e, NULL, rc, rdst, 0, cap2, src, 0, n2), gr = G.cDirect(e, NULL, gc, gdst, 0, cap2, src, 0, n2);
                    CHECK(rr == gr && rr > 0 && !memcmp(rdst->data, gdst->data, (size_t)rr), "compress with a byte[] dictionary L%d n=%d cls=%d: ref %lld shim %lld", level, n2, cls2, (long long)rr, (long long)gr);
                }
                CHECK(rLC(e, NULL, rc, NULL) == gLC(e, NULL, gc, NULL), "loadCDict0(null)");
                {   Obj* src = mk(1, 3000); Obj* rdst = mk(1, 4000); Obj* gdst = mk(1, 4000); fill(src->data, 3000, 0);       /* ... and gone again */
                    jlong const rr = R.cDirect(e, NULL, rc, rdst, 0, 4000, src, 0, 3000), gr = G.cDirect(e, NULL, gc, gdst, 0, 4000, src, 0, 3000);
                    CHECK(rr == gr && rr > 0 && !memcmp(rdst->data, gdst->data, (size_t)rr), "compress after loadCDict0(null): ref %lld shim %lld", (long long)rr, (long long)gr); }
            }
            R.cfree(e, NULL, rc); G.cfree(e, NULL, gc); R.dfree(e, NULL, rd); G.dfree(e, NULL, gd);
            R.dictFree(e, robj); G.dictFree(e, gobj);
        }
    }
    STAGE("getFrameContentSize0");
    /* Zstd.getFrameContentSize0 (N/jni_zstd.c:86-96) */
    {   typedef jlong (*fcs_fn)(JNIEnv*, jclass, jbyteArray, jint, jint, jboolean);
        fcs_fn rF = SYM(R, fcs_fn, "Zstd_getFrameContentSize0"), gF = SYM(G, fcs_fn, "Zstd_getFrameContentSize0");
        jlong rc = R.cinit(e, NULL);
        CHECK(rF && gF, "getFrameContentSize0 exported");
        for (unsigned si = 0; si < sizeof sizes / sizeof *sizes && gF; si++) {
            jsize const n = sizes[si], cap = (jsize)R.bound(e, NULL, n) + 8;
            Obj* src = mk(2, n); Obj* fr = mk(2, cap + 3); fill(src->data, n, 0);
            jlong const rr = R.cArray(e, NULL, rc, (jbyteArray)fr, 3, cap, (jbyteArray)src, 0, n);
            CHECK(rF(e, NULL, (jbyteArray)fr, 3, (jint)rr, JNI_FALSE) == gF(e, NULL, (jbyteArray)fr, 3, (jint)rr, JNI_FALSE), "getFrameContentSize0 n=%d", n);
            CHECK(rF(e, NULL, (jbyteArray)fr, 3, 4, JNI_FALSE) == gF(e, NULL, (jbyteArray)fr, 3, 4, JNI_FALSE), "getFrameContentSize0 short n=%d", n);
            CHECK(rF(e, NULL, (jbyteArray)fr, 4, (jint)rr - 1, JNI_FALSE) == gF(e, NULL, (jbyteArray)fr, 4, (jint)rr - 1, JNI_FALSE), "getFrameContentSize0 garbage n=%d", n);
        }
        R.cfree(e, NULL, rc);
    }
    STAGE("every context native");
    /* With the bundled library behind the shim: EVERY native of the two context classes goes through the shim's export and must
     * behave like the reference's own (the handle in nativePtr is the bundled library's, whoever defines the native) */
    if (getenv("ZSTD_JNI_CPU_LIB")) {
        typedef jlong (*pledge_fn)(JNIEnv*, jclass, jlong, jlong);
        typedef jobject (*prog_fn)(JNIEnv*, jclass, jlong);
        typedef jlong (*sdd_fn)(JNIEnv*, jclass, jlong, jobject, jint, jint, jobject, jint, jint, jint);
        typedef jlong (*sad_fn)(JNIEnv*, jclass, jlong, jobject, jint, jint, jbyteArray, jint, jint, jint, jint);
        typedef jlong (*sda_fn)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint, jint, jobject, jint, jint, jint);
        typedef jlong (*saa_fn)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint, jint, jbyteArray, jint, jint, jint, jint);
        pledge_fn rP = SYM(R, pledge_fn, "ZstdCompressCtx_setPledgedSrcSize0"), gP = SYM(G, pledge_fn, "ZstdCompressCtx_setPledgedSrcSize0");
        prog_fn gProg = SYM(G, prog_fn, "ZstdCompressCtx_getFrameProgression0");
        sdd_fn rS1 = SYM(R, sdd_fn, "ZstdCompressCtx_compressDirectByteBufferStream0"), gS1 = SYM(G, sdd_fn, "ZstdCompressCtx_compressDirectByteBufferStream0");
        sad_fn rS2 = SYM(R, sad_fn, "ZstdCompressCtx_compressByteArrayToDirectByteBufferStream0"), gS2 = SYM(G, sad_fn, "ZstdCompressCtx_compressByteArrayToDirectByteBufferStream0");
        sda_fn rS3 = SYM(R, sda_fn, "ZstdCompressCtx_compressDirectByteBufferToByteArrayStream0"), gS3 = SYM(G, sda_fn, "ZstdCompressCtx_compressDirectByteBufferToByteArrayStream0");
        saa_fn rS4 = SYM(R, saa_fn, "ZstdCompressCtx_compressByteArrayStream0"), gS4 = SYM(G, saa_fn, "ZstdCompressCtx_compressByteArrayStream0");
        buf_fn rDS = SYM(R, buf_fn, "ZstdDecompressCtx_decompressDirectByteBufferStream0"), gDS = SYM(G, buf_fn, "ZstdDecompressCtx_decompressDirectByteBufferStream0");
        CHECK(gP && gProg && gS1 && gS2 && gS3 && gS4 && gDS, "stream / progression / pledged-size natives exported");
        if (gP && gProg && gS1 && gS2 && gS3 && gS4 && gDS) {
            jsize const n = 30000, cap = 40000;
            jlong rc = R.cinit(e, NULL), gc = G.cinit(e, NULL), rd = R.dinit(e, NULL), gd = G.dinit(e, NULL);
            Obj* sd = mk(1, n); Obj* sa = mk(2, n); fill(sd->data, n, 0); memcpy(sa->data, sd->data, (size_t)n);
            R.setLevel(e, NULL, rc, 2); G.setLevel(e, NULL, gc, 2);
            for (int v = 0; v < 4; v++) {
                Obj* rdd = mk(1, cap); Obj* gdd = mk(1, cap); Obj* rda = mk(2, cap); Obj* gda = mk(2, cap); jlong a = 0, b = 0;
                CHECK(rP(e, NULL, rc, n) == gP(e, NULL, gc, n), "setPledgedSrcSize0");
                if (v == 0) { a = rS1(e, NULL, rc, rdd, 0, cap, sd, 0, n, 2); b = gS1(e, NULL, gc, gdd, 0, cap, sd, 0, n, 2); }
                if (v == 1) { a = rS2(e, NULL, rc, rdd, 0, cap, (jbyteArray)sa, 0, 0, n, 2); b = gS2(e, NULL, gc, gdd, 0, cap, (jbyteArray)sa, 0, 0, n, 2); }
                if (v == 2) { a = rS3(e, NULL, rc, (jbyteArray)rda, 0, 0, cap, sd, 0, n, 2); b = gS3(e, NULL, gc, (jbyteArray)gda, 0, 0, cap, sd, 0, n, 2); }
                if (v == 3) { a = rS4(e, NULL, rc, (jbyteArray)rda, 0, 0, cap, (jbyteArray)sa, 0, 0, n, 2); b = gS4(e, NULL, gc, (jbyteArray)gda, 0, 0, cap, (jbyteArray)sa, 0, 0, n, 2); }
                jsize const produced = (jsize)((a >> 32) & 0x7FFFFFFF);
                CHECK(a == b && produced > 0 && !memcmp(v < 2 ? rdd->data : rda->data, v < 2 ? gdd->data : gda->data, (size_t)produced), "compress stream native %d: ref %llx shim %llx", v, (long long)a, (long long)b);
                CHECK(gProg(e, NULL, gc) != NULL, "getFrameProgression0");
                if (v == 0) {           /* and the streaming decompress native on that frame */
                    Obj* o1 = mk(1, n); Obj* o2 = mk(1, n);
                    jlong const c = rDS(e, NULL, rd, o1, 0, n, rdd, 0, produced), d = gDS(e, NULL, gd, o2, 0, n, rdd, 0, produced);
                    CHECK(c == d && !memcmp(o1->data, o2->data, (size_t)n) && !memcmp(o2->data, sd->data, (size_t)n), "decompressDirectByteBufferStream0: ref %llx shim %llx", (long long)c, (long long)d);
                }
            }
            R.cfree(e, NULL, rc); G.cfree(e, NULL, gc); R.dfree(e, NULL, rd); G.dfree(e, NULL, gd);
        }
    }
    /* ---- the context streams: ZstdCompressCtx.compress*Stream0 in the four heap / direct combinations and ZstdDecompressCtx.decompressDirectByteBufferStream0
     * (N/jni_fast_zstd.c:392-579, :720-769).  The same directives (continue per write, flush now and then, end) on both libraries, whatever room the caller's
     * target has: the bytes that come out must be the same bytes, every input byte must be taken, the word's error bit never set; frames whose first directive is
     * the end (one-shot frames, also pledged, also into a small target, levels up to HARNESS_PLAIN_MAX_LEVEL); then the frames back through the decompress native */
    if (!getenv("HARNESS_SKIP_STREAMS")) {
        typedef jlong (*xdd_fn)(JNIEnv*, jclass, jlong, jobject, jint, jint, jobject, jint, jint, jint);
        typedef jlong (*xad_fn)(JNIEnv*, jclass, jlong, jobject, jint, jint, jbyteArray, jint, jint, jint, jint);
        typedef jlong (*xda_fn)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint, jint, jobject, jint, jint, jint);
        typedef jlong (*xaa_fn)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint, jint, jbyteArray, jint, jint, jint, jint);
        typedef jlong (*xpl_fn)(JNIEnv*, jclass, jlong, jlong); typedef jobject (*xpr_fn)(JNIEnv*, jclass, jlong);
        typedef jlong (*xds_fn)(JNIEnv*, jclass, jlong, jobject, jint, jint, jobject, jint, jint);
        struct { xdd_fn dd; xad_fn ad; xda_fn da; xaa_fn aa; xpl_fn pledge; xpr_fn prog; xds_fn ds; } X[2];
        Lib* libs[2] = {&R, &G};
        jsize const totals[] = {0, 1, 100, 5000, 70000, 131072, 200000, 300000, 600000, 2097152, 2200000};
        int const streamMax = getenv("HARNESS_STREAM_MAX") ? atoi(getenv("HARNESS_STREAM_MAX")) : (1 << 30);
        STAGE("context streams");
        for (int k = 0; k < 2; k++) {
#define XS(field, name) *(void**)&X[k].field = dlsym(libs[k]->h, P name)
            XS(dd, "ZstdCompressCtx_compressDirectByteBufferStream0"); XS(ad, "ZstdCompressCtx_compressByteArrayToDirectByteBufferStream0");
            XS(da, "ZstdCompressCtx_compressDirectByteBufferToByteArrayStream0"); XS(aa, "ZstdCompressCtx_compressByteArrayStream0");
            XS(pledge, "ZstdCompressCtx_setPledgedSrcSize0"); XS(prog, "ZstdCompressCtx_getFrameProgression0"); XS(ds, "ZstdDecompressCtx_decompressDirectByteBufferStream0");
#undef XS
            CHECK(X[k].dd && X[k].ad && X[k].da && X[k].aa && X[k].pledge && X[k].prog && X[k].ds, "context stream natives of library %d", k);
        }
        /* one directive, in combination `comb` (0 direct/direct, 1 array source, 2 array target, 3 both arrays), repeated until it reports done (or, for continue, until the
         * source is taken); what comes out is appended to out */
#define DIRECTIVE(k, comb, ctx, srcObjD, srcObjA, sBase, sFrom, sTo, op) do { \
            jint spos = (sFrom); int guard_ = 0; \
            for (;;) { \
                jlong w_ = 0; jint const aoffD = 3, aoffS = 2; \
                if ((comb) == 0) w_ = X[k].dd(e, NULL, ctx, (jobject)dstD, 0, room, (jobject)(srcObjD), spos, (sTo), (op)); \
                if ((comb) == 1) w_ = X[k].ad(e, NULL, ctx, (jobject)dstD, 0, room, (jbyteArray)(srcObjA), aoffS, spos, (sTo), (op)); \
                if ((comb) == 2) w_ = X[k].da(e, NULL, ctx, (jbyteArray)dstA, aoffD, 0, room, (jobject)(srcObjD), spos, (sTo), (op)); \
                if ((comb) == 3) w_ = X[k].aa(e, NULL, ctx, (jbyteArray)dstA, aoffD, 0, room, (jbyteArray)(srcObjA), aoffS, spos, (sTo), (op)); \
                if ((uint64_t)w_ & 0x80000000u) { worst[k] = (jlong)((uint64_t)w_ & 0xFFFFFFFFu); break; } \
                {   jint const dp_ = (jint)(((uint64_t)w_ >> 32) & 0x7FFFFFFFu), sp_ = (jint)((uint64_t)w_ & 0x7FFFFFFFu); \
                    if (sp_ < spos || sp_ > (sTo) || dp_ > room) { worst[k] = -999; break; } \
                    if (n + (size_t)dp_ > cap) { worst[k] = -998; break; } \
                    memcpy(out + n, ((comb) & 2) ? dstA->data + aoffD : dstD->data, (size_t)dp_); n += (size_t)dp_; \
                    spos = sp_; } \
                if (((op) == 0 && spos == (sTo)) || ((op) != 0 && ((uint64_t)w_ >> 63) && spos == (sTo))) break; \
                if (++guard_ > 200000) { worst[k] = -997; break; } \
            } } while (0)
        for (unsigned ti = 0; ti < sizeof totals / sizeof *totals; ti++) for (int variant = 0; variant < 4; variant++) {
            jsize const total = totals[ti];
            jint const level = 1 + (jint)((ti + (unsigned)variant) % 3u);
            jsize const chunk = variant == 0 ? 50000 : (variant == 1 ? 131072 : (variant == 2 ? 7000 : 300000));
            int const flushEvery = variant == 2 ? 3 : (variant == 3 ? 1 : 0);
            jint const room = variant == 1 ? 900 : (1 << 22);
            int const comb = (int)((ti + (unsigned)variant) & 3u), ck = (int)(ti & 1u);
            if ((long long)total > (1ll << (18 + level)) && total > streamMax) continue;
            if (getenv("HARNESS_TOTAL_MAX") && total > atoi(getenv("HARNESS_TOTAL_MAX"))) continue;      /* (a quick pass: the long streams left out) */
            Obj* srcD = mk(1, total > 0 ? total : 1); Obj* srcA = mk(2, total + 2); fill(srcD->data, total, (int)(ti % 3u)); memcpy(srcA->data + 2, srcD->data, (size_t)total);
            char* outs[2]; size_t lens[2] = {0, 0}; jlong worst[2] = {0, 0};
            for (int k = 0; k < 2; k++) {
                Obj* dstD = mk(1, room); Obj* dstA = mk(2, room + 3);
                jlong const ctx = libs[k]->cinit(e, NULL);
                size_t const cap = (size_t)total + (size_t)total / 64 + (1u << 16); char* out = (char*)malloc(cap); size_t n = 0; int calls = 0;
                libs[k]->setLevel(e, NULL, ctx, level); libs[k]->setChecksum(e, NULL, ctx, ck ? JNI_TRUE : JNI_FALSE);
                for (jsize at = 0; at < total && worst[k] == 0; ) {
                    jsize const len = total - at < chunk ? total - at : chunk;
                    DIRECTIVE(k, comb, ctx, srcD, srcA, 0, at, at + len, 0);
                    at += len; calls++;
                    if (flushEvery && calls % flushEvery == 0 && worst[k] == 0) DIRECTIVE(k, comb, ctx, srcD, srcA, 0, at, at, 1);
                    if (calls == 2 && worst[k] == 0) CHECK(X[k].prog(e, NULL, ctx) != NULL, "getFrameProgression0 inside a frame (library %d)", k);
                }
                if (worst[k] == 0) DIRECTIVE(k, comb, ctx, srcD, srcA, 0, total, total, 2);
                /* the context is at a frame boundary again: a second, small frame through it */
                if (worst[k] == 0 && total > 100) { DIRECTIVE(k, comb, ctx, srcD, srcA, 0, 0, 60, 0); if (worst[k] == 0) DIRECTIVE(k, comb, ctx, srcD, srcA, 0, 60, 100, 2); }
                libs[k]->cfree(e, NULL, ctx);
                outs[k] = out; lens[k] = n;
                free(dstD->data); free(dstD); free(dstA->data); free(dstA);
            }
            CHECK(worst[0] == 0 && worst[1] == 0 && lens[0] == lens[1] && !memcmp(outs[0], outs[1], lens[0]), "context stream of %d bytes, level %d ck %d, writes of %d, flush every %d, room %d, combination %d: ref %zu bytes (%lld), shim %zu bytes (%lld)",
                  (int)total, (int)level, ck, (int)chunk, flushEvery, (int)room, comb, lens[0], (long long)worst[0], lens[1], (long long)worst[1]);
            free(outs[0]); free(outs[1]); free(srcD->data); free(srcD); free(srcA->data); free(srcA);
        }
        /* an empty directive opens the frame, the end directive brings all of it: libzstd's one-piece path under a stream header (the bundled library's above one block) */
        if (streamMax != 0) for (int level = 1; level <= 3; level++) for (int opener = 0; opener < 2; opener++) for (int roomy = 0; roomy < 2; roomy++) {
            jsize const total = 330000; jint const room = roomy ? (1 << 20) : 5000; int const comb = (level + opener + roomy) & 3;
            Obj* srcD = mk(1, total); Obj* srcA = mk(2, total + 2); g_x = 0x1234567ull + (unsigned)level; fill(srcD->data, total, 1); memcpy(srcA->data + 2, srcD->data, (size_t)total);
            char* outs[2]; size_t lens[2] = {0, 0}; jlong worst[2] = {0, 0};
            for (int k = 0; k < 2; k++) {
                Obj* dstD = mk(1, room); Obj* dstA = mk(2, room + 3);
                jlong const ctx = libs[k]->cinit(e, NULL);
                size_t const cap = (size_t)total + (1u << 16); char* out = (char*)malloc(cap); size_t n = 0;
                libs[k]->setLevel(e, NULL, ctx, level);
                DIRECTIVE(k, comb, ctx, srcD, srcA, 0, 0, 0, opener);
                if (worst[k] == 0) DIRECTIVE(k, comb, ctx, srcD, srcA, 0, 0, total, 2);
                libs[k]->cfree(e, NULL, ctx);
                outs[k] = out; lens[k] = n;
                free(dstD->data); free(dstD); free(dstA->data); free(dstA);
            }
            CHECK(worst[0] == 0 && worst[1] == 0 && lens[0] == lens[1] && !memcmp(outs[0], outs[1], lens[0]), "empty %s, then the end with %d bytes, level %d, room %d: ref %zu bytes (%lld), shim %zu bytes (%lld)",
                  opener ? "flush" : "write", (int)total, level, (int)room, lens[0], (long long)worst[0], lens[1], (long long)worst[1]);
            free(outs[0]); free(outs[1]); free(srcD->data); free(srcD); free(srcA->data); free(srcA);
        }
        /* frames whose first directive is the end: with and without a pledged size, a pledge that does not match (libzstd replaces it by the input's size), a roomy and a tiny target, the four combinations */
        {   int const plainMax2 = getenv("HARNESS_PLAIN_MAX_LEVEL") ? atoi(getenv("HARNESS_PLAIN_MAX_LEVEL")) : (getenv("HARNESS_MAX_LEVEL") ? atoi(getenv("HARNESS_MAX_LEVEL")) : 3);
            jsize const ones[] = {0, 1, 300, 30000, 131072, 131073, 500000};
            for (unsigned oi = 0; oi < sizeof ones / sizeof *ones; oi++) for (int level = 1; level <= plainMax2; level++) for (int mode = 0; mode < 4; mode++) {
                jsize const total = ones[oi];
                jint const room = (mode == 1) ? 700 : (1 << 20);
                int const comb = (int)((oi + (unsigned)level + (unsigned)mode) & 3u);
                if (level > 3 && total > 131072) continue;
                if (mode == 1 && total > 131072 && streamMax == 0) continue;      /* above one block into a target below the frame's bound: libzstd's buffered form, the bundled library's (none in this leg) */
                Obj* srcD = mk(1, total > 0 ? total : 1); Obj* srcA = mk(2, total + 2); fill(srcD->data, total, (int)(oi % 3u)); memcpy(srcA->data + 2, srcD->data, (size_t)total);
                char* outs[2]; size_t lens[2] = {0, 0}; jlong worst[2] = {0, 0};
                for (int k = 0; k < 2; k++) {
                    Obj* dstD = mk(1, room); Obj* dstA = mk(2, room + 3);
                    jlong const ctx = libs[k]->cinit(e, NULL);
                    size_t const cap = (size_t)total + (size_t)total / 64 + (1u << 16); char* out = (char*)malloc(cap); size_t n = 0;
                    libs[k]->setLevel(e, NULL, ctx, level); libs[k]->setChecksum(e, NULL, ctx, (oi & 1) ? JNI_TRUE : JNI_FALSE);
                    if (mode == 2) worst[k] = X[k].pledge(e, NULL, ctx, total);
                    if (mode == 3) worst[k] = X[k].pledge(e, NULL, ctx, (jlong)total + 5);
                    if (worst[k] == 0) DIRECTIVE(k, comb, ctx, srcD, srcA, 0, 0, total, 2);
                    if (worst[k] == 0 && total >= 300) { DIRECTIVE(k, comb, ctx, srcD, srcA, 0, 0, 300, 2); }       /* and the next frame through the same context */
                    libs[k]->cfree(e, NULL, ctx);
                    outs[k] = out; lens[k] = n;
                    free(dstD->data); free(dstD); free(dstA->data); free(dstA);
                }
                CHECK(worst[0] == 0 && worst[1] == 0 && lens[0] == lens[1] && !memcmp(outs[0], outs[1], lens[0]), "one-directive frame of %d bytes, level %d, mode %d, room %d, combination %d: ref %zu bytes (%lld), shim %zu bytes (%lld)",
                           (int)total, level, mode, (int)room, comb, lens[0], (long long)worst[0], lens[1], (long long)worst[1]);
                /* back through the decompress native of both libraries: the whole frame in the source, room for all of it; then a target one byte short (an error or more calls, never wrong bytes) */
                if (mode == 0 && worst[0] == 0 && lens[0] > 0) {
                    jsize const flen = (jsize)lens[0];
                    Obj* fr = mk(1, flen + 7); memcpy(fr->data + 4, outs[0], (size_t)flen);
                    for (int k = 0; k < 2; k++) {
                        jlong const dctx = libs[k]->dinit(e, NULL); Obj* back = mk(1, total + 300 + 9);
                        jint dpos = 5, spos = 4; int guard = 0; uint64_t w = 0; int frames = 0;
                        while (spos < 4 + flen && guard++ < 1000) {
                            w = (uint64_t)X[k].ds(e, NULL, dctx, (jobject)back, dpos, total + 300 + 9, (jobject)fr, spos, 4 + flen);
                            if (w & 0x80000000u) break;
                            dpos = (jint)((w >> 32) & 0x7FFFFFFFu); spos = (jint)(w & 0x7FFFFFFFu); if (w >> 63) frames++;
                        }
                        CHECK(!(w & 0x80000000u) && (w >> 63) && spos == 4 + flen && dpos == 5 + total + (total >= 300 ? 300 : 0) && !memcmp(back->data + 5, srcD->data, (size_t)total),
                              "decompressDirectByteBufferStream0 (library %d) of the %d-byte frame(s), level %d: word %llx, %d out, %d consumed, %d frame ends", k, (int)total, level, (unsigned long long)w, (int)dpos - 5, (int)spos - 4, frames);
                        libs[k]->dfree(e, NULL, dctx); free(back->data); free(back);
                    }
                    free(fr->data); free(fr);
                }
                free(outs[0]); free(outs[1]); free(srcD->data); free(srcD); free(srcA->data); free(srcA);
            }
        }
        /* HARNESS_FUZZ=<seed>,<iterations>: random scripts of directives on a context — one to three frames per context, writes of random sizes (empty ones too), flushes
         * (repeated ones too) at random, a frame that is a single end directive now and then (with a pledged size that may or may not match), any of the four heap / direct
         * combinations, a target of a few bytes or a roomy one.  The bytes both libraries hand out over the whole script must be the same; then the concatenated frames go
         * back through the decompress native (whole, or in random pieces when there is a bundled library to take pieces). */
        if (getenv("HARNESS_FUZZ")) {
            unsigned long long seed = 1; int iters = 100; int const haveCpu = getenv("ZSTD_JNI_CPU_LIB") != NULL;
            sscanf(getenv("HARNESS_FUZZ"), "%llu,%d", &seed, &iters);
            g_x = 0x9E3779B97F4A7C15ull ^ (seed * 0xD1B54A32D192ED03ull); if (!g_x) g_x = 1;
            int const fuzzMaxLevel = haveCpu ? 5 : 3; int const trace = getenv("HARNESS_FUZZ_TRACE") ? atoi(getenv("HARNESS_FUZZ_TRACE")) : -1;
#define TRACE(...) do { if (it == trace && k == 0) { printf("  script: "); printf(__VA_ARGS__); printf("\n"); } } while (0)
            STAGE("context streams: random scripts");
            for (int it = 0; it < iters; it++) {
                int const nFrames = 1 + (int)(rnd() % 3u), comb = (int)(rnd() & 3u), level = 1 + (int)(rnd() % (unsigned)fuzzMaxLevel), ck = (int)(rnd() & 1u), cls = (int)(rnd() % 3u);
                jint const room = (rnd() & 1u) ? (jint)(1 + rnd() % 2000u) : (1 << 20);
                jsize const poolN = 400000; jsize frameLen[3]; int oneShot[3]; jlong pledge[3]; jsize grand = 0;
                unsigned long long const script = g_x;                                      /* both libraries replay the same random script */
                for (int f = 0; f < nFrames; f++) {
                    unsigned const pick = rnd() % 10u;
                    frameLen[f] = pick == 0 ? 0 : (pick < 4 ? (jsize)(rnd() % 3000u) : (pick < 8 ? (jsize)(rnd() % 140000u) : (jsize)(rnd() % 330000u)));
                    oneShot[f] = (rnd() % 4u) == 0; pledge[f] = -1;
                    if ((oneShot[f] || (haveCpu && (rnd() & 3u) == 0)) && (rnd() & 1u)) pledge[f] = (rnd() & 1u) ? frameLen[f] : (jlong)(rnd() % 500000u);      /* (a pledged frame in several directives is the bundled library's; a broken pledge must fail alike) */
                    if (!oneShot[f] && level > 3 && !haveCpu) frameLen[f] = 0;
                    if (oneShot[f] && frameLen[f] > 131072 && room < (1 << 20) && !haveCpu) { oneShot[f] = 0; pledge[f] = -1; }      /* (the buffered form of a one-directive frame: see the shim) */
                    grand += frameLen[f];
                }
                Obj* srcD = mk(1, poolN); Obj* srcA = mk(2, poolN + 2); fill(srcD->data, poolN, cls); memcpy(srcA->data + 2, srcD->data, (size_t)poolN);
                unsigned long long const afterFill = g_x;
                char* outs[2]; size_t lens[2] = {0, 0}; jlong worst[2] = {0, 0};
                (void)script;
                for (int k = 0; k < 2; k++) {
                    Obj* dstD = mk(1, room); Obj* dstA = mk(2, room + 3);
                    jlong const ctx = libs[k]->cinit(e, NULL);
                    size_t const cap = (size_t)grand + (size_t)grand / 32 + (1u << 16) + 4096u * (size_t)nFrames; char* out = (char*)malloc(cap); size_t n = 0;
                    g_x = afterFill;
                    libs[k]->setLevel(e, NULL, ctx, level); libs[k]->setChecksum(e, NULL, ctx, ck ? JNI_TRUE : JNI_FALSE);
                    for (int f = 0; f < nFrames && worst[k] == 0; f++) {
                        jsize const base = (jsize)(rnd() % (unsigned)(poolN - frameLen[f] + 1)), end = base + frameLen[f];
                        TRACE("frame %d: %d bytes at %d, oneShot %d, pledge %lld, level %d ck %d cls %d", f, (int)frameLen[f], (int)base, oneShot[f], (long long)pledge[f], level, ck, cls);
                        if (pledge[f] >= 0) { jlong const pr = X[k].pledge(e, NULL, ctx, pledge[f]); if (pr != 0) { worst[k] = pr; break; } }
                        if (oneShot[f]) { DIRECTIVE(k, comb, ctx, srcD, srcA, 0, base, end, 2); continue; }
                        for (jsize at = base; worst[k] == 0; ) {
                            unsigned const r = rnd() % 16u;
                            jsize len = r == 0 ? 0 : (r < 6 ? (jsize)(rnd() % 2000u) : (r < 13 ? (jsize)(rnd() % 70000u) : (jsize)(rnd() % 300000u)));
                            if (len > end - at) len = end - at;
                            TRACE("write %d (to %d)", (int)len, (int)(at + len - base));
                            DIRECTIVE(k, comb, ctx, srcD, srcA, 0, at, at + len, 0);
                            at += len;
                            if (worst[k] == 0 && (rnd() % 10u) < 3u) { TRACE("flush at %d", (int)(at - base)); DIRECTIVE(k, comb, ctx, srcD, srcA, 0, at, at, 1); if (worst[k] == 0 && (rnd() & 3u) == 0) { TRACE("flush again"); DIRECTIVE(k, comb, ctx, srcD, srcA, 0, at, at, 1); } }
                            if (at >= end && (len > 0 || (rnd() & 1u))) break;
                        }
                        if (worst[k] == 0) DIRECTIVE(k, comb, ctx, srcD, srcA, 0, end, end, 2);
                    }
                    libs[k]->cfree(e, NULL, ctx);
                    outs[k] = out; lens[k] = n;
                    free(dstD->data); free(dstD); free(dstA->data); free(dstA);
                }
                CHECK(worst[0] == worst[1] && lens[0] == lens[1] && !memcmp(outs[0], outs[1], lens[0]), "random script %d (seed %llu): %d frames, level %d ck %d, room %d, combination %d: ref %zu bytes (%lld), shim %zu bytes (%lld)",
                      it, seed, nFrames, level, ck, (int)room, comb, lens[0], (long long)worst[0], lens[1], (long long)worst[1]);
                if (worst[0] == 0 && lens[0] > 0) {                                             /* the frames back, through both decompress natives */
                    jsize const flen = (jsize)lens[0];
                    Obj* fr = mk(1, flen + 1); memcpy(fr->data, outs[0], (size_t)flen);
                    unsigned long long const cut = g_x;
                    for (int k = 0; k < 2; k++) {
                        jlong const dctx = libs[k]->dinit(e, NULL); Obj* back = mk(1, grand + 16);
                        jint dpos = 0, spos = 0; int guard = 0; uint64_t w = 0;
                        g_x = cut;
                        while (spos < flen && guard++ < 100000) {
                            jint const upto = ((haveCpu || getenv("HARNESS_PIECES")) && (rnd() & 1u)) ? spos + 1 + (jint)(rnd() % (unsigned)(flen - spos)) : flen;      /* pieces need a stream behind the GPU route */
                            jint const before = spos;
                            w = (uint64_t)X[k].ds(e, NULL, dctx, (jobject)back, dpos, grand + 16, (jobject)fr, spos, upto);
                            if (w & 0x80000000u) break;
                            dpos = (jint)((w >> 32) & 0x7FFFFFFFu); spos = (jint)(w & 0x7FFFFFFFu);
                            if (spos == before && upto == flen) break;
                        }
                        CHECK(!(w & 0x80000000u) && spos == flen && dpos == grand, "random script %d (seed %llu): decompress native of library %d: word %llx, %d of %d out, %d of %d consumed", it, seed, k, (unsigned long long)w, (int)dpos, (int)grand, (int)spos, (int)flen);
                        libs[k]->dfree(e, NULL, dctx); free(back->data); free(back);
                    }
                    free(fr->data); free(fr);
                }
                free(outs[0]); free(outs[1]); free(srcD->data); free(srcD); free(srcA->data); free(srcA);
            }
#undef TRACE
            printf("JNI-HARNESS FUZZ seed=%llu scripts=%d\n", seed, iters);
        }
#undef DIRECTIVE
    }
    /* ---- the DirectByteBuffer stream classes (N/jni_directbuffercompress_zstd.c, jni_directbufferdecompress_zstd.c): the same writes, flushes and close on
     * both libraries, whatever room the caller's target buffer has — the bytes that come out must be the same bytes; then the frames through both decompress streams */
    if (!getenv("HARNESS_SKIP_STREAMS")) {
        typedef jlong (*create_fn)(JNIEnv*, jclass); typedef jlong (*free_fn)(JNIEnv*, jclass, jlong); typedef jlong (*init_fn)(JNIEnv*, jobject, jlong, jint);
        typedef jlong (*comp_fn)(JNIEnv*, jobject, jlong, jobject, jint, jint, jobject, jint, jint); typedef jlong (*end_fn)(JNIEnv*, jobject, jlong, jobject, jint, jint);
        typedef jlong (*dinit_fn)(JNIEnv*, jobject, jlong);
        struct { create_fn create; free_fn free_; init_fn init; comp_fn comp; end_fn flush, end; create_fn dcreate; free_fn dfree; dinit_fn dinit; comp_fn dstream; } S[2];
        Lib* libs[2] = {&R, &G};
        jsize const totals[] = {0, 1, 100, 5000, 70000, 131072, 200000, 300000, 600000, 2097152, 2200000};
        int const streamMax = getenv("HARNESS_STREAM_MAX") ? atoi(getenv("HARNESS_STREAM_MAX")) : (1 << 30);     /* the GPU-only leg has no bundled stream to outgrow into */
        STAGE("DirectByteBuffer streams");
        for (int k = 0; k < 2; k++) {
#define SS(field, name) *(void**)&S[k].field = dlsym(libs[k]->h, P "ZstdDirectBufferCompressingStreamNoFinalizer_" name)
            SS(create, "createCStream"); SS(free_, "freeCStream"); SS(init, "initCStream"); SS(comp, "compressDirectByteBuffer"); SS(flush, "flushStream"); SS(end, "endStream");
#undef SS
#define SS(field, name) *(void**)&S[k].field = dlsym(libs[k]->h, P "ZstdDirectBufferDecompressingStreamNoFinalizer_" name)
            SS(dcreate, "createDStreamNative"); SS(dfree, "freeDStreamNative"); SS(dinit, "initDStreamNative"); SS(dstream, "decompressStreamNative");
#undef SS
            CHECK(S[k].create && S[k].free_ && S[k].init && S[k].comp && S[k].flush && S[k].end && S[k].dcreate && S[k].dfree && S[k].dinit && S[k].dstream, "stream natives of library %d", k);
        }
        for (unsigned ti = 0; ti < sizeof totals / sizeof *totals; ti++) for (int variant = 0; variant < 4; variant++) {
            jsize const total = totals[ti];
            jint const level = 1 + (jint)((ti + (unsigned)variant) % 3u);
            jsize const chunk = variant == 0 ? 50000 : (variant == 1 ? 131072 : (variant == 2 ? 7000 : 300000));
            int const flushEvery = variant == 2 ? 3 : (variant == 3 ? 1 : 0);
            jsize const room = variant == 1 ? 900 : (1 << 22);                       /* a target buffer far too small: the natives must say how much is pending */
            if ((long long)total > (1ll << (18 + level)) && total > streamMax) continue;
            if (getenv("HARNESS_TOTAL_MAX") && total > atoi(getenv("HARNESS_TOTAL_MAX"))) continue;      /* (a quick pass: the long streams left out) */
            Obj* src = mk(1, total > 0 ? total : 1); fill(src->data, total, (int)(ti % 3u));
            char* outs[2]; size_t lens[2] = {0, 0}; jlong worst[2] = {0, 0};
            for (int k = 0; k < 2; k++) {
                Obj* self = mk(7, 0); Obj* dst = mk(1, room);
                jlong const h = S[k].create(e, NULL); jlong r = S[k].init(e, (jobject)self, h, level);
                size_t cap = (size_t)total + (size_t)total / 64 + (1u << 16); char* out = (char*)malloc(cap); size_t n = 0; int calls = 0;
                if (r < 0) worst[k] = r;
                for (jsize at = 0; at < total && worst[k] == 0; ) {
                    jsize const len = total - at < chunk ? total - at : chunk; jsize done = 0; int guard = 0;
                    while (done < len && guard++ < 100000) {
                        self->consumed = self->produced = 0;
                        r = S[k].comp(e, (jobject)self, h, (jobject)dst, 0, room, (jobject)src, at + done, len - done);
                        if (r < 0) { worst[k] = r; break; }
                        memcpy(out + n, dst->data, (size_t)self->produced); n += (size_t)self->produced; done += self->consumed;
                    }
                    at += len; calls++;
                    if (flushEvery && calls % flushEvery == 0 && worst[k] == 0) {
                        int guard2 = 0;
                        do { self->produced = 0; r = S[k].flush(e, (jobject)self, h, (jobject)dst, 0, room); if (r < 0) { worst[k] = r; break; } memcpy(out + n, dst->data, (size_t)self->produced); n += (size_t)self->produced; } while (r > 0 && guard2++ < 100000);
                    }
                }
                if (worst[k] == 0) { int guard3 = 0; do { self->produced = 0; r = S[k].end(e, (jobject)self, h, (jobject)dst, 0, room); if (r < 0) { worst[k] = r; break; } memcpy(out + n, dst->data, (size_t)self->produced); n += (size_t)self->produced; } while (r > 0 && guard3++ < 100000); }
                S[k].free_(e, NULL, h);
                outs[k] = out; lens[k] = n;
            }
            CHECK(worst[0] == 0 && worst[1] == 0 && lens[0] == lens[1] && !memcmp(outs[0], outs[1], lens[0]), "stream of %d bytes, level %d, writes of %d, flush every %d, room %d: ref %zu bytes (%lld), shim %zu bytes (%lld)",
                  (int)total, (int)level, (int)chunk, flushEvery, (int)room, lens[0], (long long)worst[0], lens[1], (long long)worst[1]);
            /* the frame back through both decompress streams: whole frame in the source buffer, room for all of it */
            if (worst[0] == 0 && variant != 1) {
                Obj* fr = mk(1, (jsize)lens[0] + 1); memcpy(fr->data, outs[0], lens[0]);
                for (int k = 0; k < 2; k++) {
                    Obj* self = mk(7, 0); Obj* back = mk(1, total + 64);
                    jlong const h = S[k].dcreate(e, NULL); jlong r = S[k].dinit(e, (jobject)self, h);
                    jsize got = 0, used = 0; int guard = 0;
                    do { self->consumed = self->produced = 0;
                         r = S[k].dstream(e, (jobject)self, h, (jobject)back, got, total + 64 - got, (jobject)fr, used, (jsize)lens[0] - used);
                         got += self->produced; used += self->consumed; } while (r > 0 && guard++ < 1000);
                    CHECK(r == 0 && got == total && used == (jsize)lens[0] && !memcmp(back->data, src->data, (size_t)total), "decompress stream (library %d) of the %d-byte stream frame: ret %lld, %d bytes out, %d consumed", k, (int)total, (long long)r, (int)got, (int)used);
                    S[k].dfree(e, NULL, h);
                }
                /* ... and in pieces of 4 000 bytes into a target of 3 000: the bundled stream's case, or (no bundled library) the frame collected by the shim */
                if (streamMax != 0 || getenv("HARNESS_PIECES")) for (int k = 0; k < 2; k++) {
                    Obj* self = mk(7, 0); Obj* back = mk(1, 3000); char* all = (char*)malloc((size_t)total + 64); jsize got = 0;
                    jlong const h = S[k].dcreate(e, NULL); jlong r = S[k].dinit(e, (jobject)self, h);
                    jsize const flen = (jsize)lens[0]; jsize used = 0, fed = 0; int guard = 0;
                    do {
                        if (used == fed && fed < flen) fed = fed + 4000 < flen ? fed + 4000 : flen;
                        self->consumed = self->produced = 0;
                        r = S[k].dstream(e, (jobject)self, h, (jobject)back, 0, 3000, (jobject)fr, used, fed - used);
                        if (r >= 0 && got + self->produced <= total + 64) { memcpy(all + got, back->data, (size_t)self->produced); got += self->produced; }
                        used += self->consumed;
                    } while (r >= 0 && (used < flen || r > 0) && guard++ < 400000);
                    CHECK(r == 0 && got == total && used == flen && !memcmp(all, src->data, (size_t)total), "decompress stream in pieces (library %d) of the %d-byte stream frame: ret %lld, %d bytes out, %d consumed", k, (int)total, (long long)r, (int)got, (int)used);
                    S[k].dfree(e, NULL, h); free(all);
                }
                /* ... and with a skippable frame in front and one behind (ZSTD_decompressStream passes over them), fed in pieces of 7 bytes at first */
                if ((streamMax != 0 || getenv("HARNESS_PIECES")) && variant == 0) {
                    static const unsigned char skip[13] = {0x5A, 0x2A, 0x4D, 0x18, 5, 0, 0, 0, 'h', 'e', 'l', 'l', 'o'};
                    jsize const flen = (jsize)lens[0] + 26;
                    Obj* fr2 = mk(1, flen + 1); memcpy(fr2->data, skip, 13); memcpy(fr2->data + 13, outs[0], lens[0]); memcpy(fr2->data + 13 + lens[0], skip, 13);
                    jlong rets[2] = {0, 0}; jsize gots[2] = {0, 0}, useds[2] = {0, 0};
                    for (int k = 0; k < 2; k++) {
                        Obj* self = mk(7, 0); Obj* back = mk(1, total + 64);
                        jlong const h = S[k].dcreate(e, NULL); jlong r = S[k].dinit(e, (jobject)self, h);
                        jsize used = 0, fed = 0, got = 0; int guard = 0;
                        do {
                            if (used == fed && fed < flen) fed = fed + (fed < 40 ? 7 : 50000) < flen ? fed + (fed < 40 ? 7 : 50000) : flen;
                            self->consumed = self->produced = 0;
                            r = S[k].dstream(e, (jobject)self, h, (jobject)back, got, total + 64 - got, (jobject)fr2, used, fed - used);
                            got += self->produced; used += self->consumed;
                        } while (r >= 0 && (used < flen || r > 0) && guard++ < 400000);
                        CHECK(r == 0 && got == total && used == flen && !memcmp(back->data, src->data, (size_t)total), "decompress stream with skippable frames around (library %d), %d bytes: ret %lld, %d out, %d of %d consumed", k, (int)total, (long long)r, (int)got, (int)used, (int)flen);
                        rets[k] = r; gots[k] = got; useds[k] = used;
                        S[k].dfree(e, NULL, h); free(back->data); free(back);
                    }
                    CHECK(rets[0] == rets[1] && gots[0] == gots[1] && useds[0] == useds[1], "skippable frames: both libraries alike");
                    free(fr2->data); free(fr2);
                }
            }
            free(outs[0]); free(outs[1]);
        }
    }
    /* ZstdOutputStreamNoFinalizer / ZstdInputStreamNoFinalizer: the heap-array stream classes driven as their Java code drives them (write loop on srcPos,
     * flush / end loops on the return value, read loop on dstPos), level and checksum through class Zstd's natives on the stream handle: the same bytes from both libraries */
    if (!getenv("HARNESS_SKIP_STREAMS")) {
        typedef jlong (*create_fn)(JNIEnv*, jclass); typedef jint (*free_fn)(JNIEnv*, jclass, jlong); typedef jint (*reset_fn)(JNIEnv*, jobject, jlong);
        typedef jint (*comp_fn)(JNIEnv*, jobject, jlong, jbyteArray, jint, jbyteArray, jint); typedef jint (*end_fn)(JNIEnv*, jobject, jlong, jbyteArray, jint);
        typedef jint (*set_fn)(JNIEnv*, jclass, jlong, jint); typedef jint (*setb_fn)(JNIEnv*, jclass, jlong, jboolean);
        struct { create_fn create; free_fn free_; reset_fn reset; comp_fn comp; end_fn flush, end; set_fn level; setb_fn checksum; create_fn dcreate; free_fn dfree; reset_fn dinit; comp_fn dstream; } S[2];
        Lib* libs[2] = {&R, &G};
        jsize const totals[] = {0, 1, 100, 5000, 70000, 131072, 200000, 300000, 600000, 2097152, 2200000};
        int const streamMax = getenv("HARNESS_STREAM_MAX") ? atoi(getenv("HARNESS_STREAM_MAX")) : (1 << 30);
        STAGE("heap-array streams (ZstdOutputStream / ZstdInputStream)");
        for (int k = 0; k < 2; k++) {
#define SS(field, name) *(void**)&S[k].field = dlsym(libs[k]->h, P name)
            SS(create, "ZstdOutputStreamNoFinalizer_createCStream"); SS(free_, "ZstdOutputStreamNoFinalizer_freeCStream"); SS(reset, "ZstdOutputStreamNoFinalizer_resetCStream");
            SS(comp, "ZstdOutputStreamNoFinalizer_compressStream"); SS(flush, "ZstdOutputStreamNoFinalizer_flushStream"); SS(end, "ZstdOutputStreamNoFinalizer_endStream");
            SS(level, "Zstd_setCompressionLevel"); SS(checksum, "Zstd_setCompressionChecksums");
            SS(dcreate, "ZstdInputStreamNoFinalizer_createDStream"); SS(dfree, "ZstdInputStreamNoFinalizer_freeDStream"); SS(dinit, "ZstdInputStreamNoFinalizer_initDStream"); SS(dstream, "ZstdInputStreamNoFinalizer_decompressStream");
#undef SS
            CHECK(S[k].create && S[k].free_ && S[k].reset && S[k].comp && S[k].flush && S[k].end && S[k].level && S[k].checksum && S[k].dcreate && S[k].dfree && S[k].dinit && S[k].dstream, "heap-array stream natives of library %d", k);
        }
        for (unsigned ti = 0; ti < sizeof totals / sizeof *totals; ti++) for (int variant = 0; variant < 4; variant++) {
            jsize const total = totals[ti];
            jint const level = 1 + (jint)((ti + (unsigned)variant) % 3u);
            jsize const chunk = variant == 0 ? 50000 : (variant == 1 ? 131072 : (variant == 2 ? 7000 : 300000));
            int const flushEvery = variant == 2 ? 3 : (variant == 3 ? 1 : 0);
            jsize const room = variant == 1 ? 900 : 131591;                           /* variant 1: a target array far too small; else ZSTD_CStreamOutSize() as the Java class uses */
            jboolean const ck = (ti + (unsigned)variant) % 2u ? JNI_TRUE : JNI_FALSE;
            int const frames = variant == 0 && total <= 70000 ? 2 : 1;                /* two frames through one stream object: reset keeps level and checksum */
            if ((long long)total > (1ll << (18 + level)) && total > streamMax) continue;
            if (getenv("HARNESS_TOTAL_MAX") && total > atoi(getenv("HARNESS_TOTAL_MAX"))) continue;      /* (a quick pass: the long streams left out) */
            Obj* src = mk(2, total > 0 ? total : 1); fill(src->data, total, (int)(ti % 3u));
            char* outs[2]; size_t lens[2] = {0, 0}; jlong worst[2] = {0, 0};
            for (int k = 0; k < 2; k++) {
                Obj* self = mk(7, 0); Obj* dst = mk(2, room);
                jlong const h = S[k].create(e, NULL); jint r;
                size_t cap = ((size_t)total + (size_t)total / 64 + (1u << 16)) * (size_t)frames; char* out = (char*)malloc(cap); size_t n = 0;
                r = S[k].level(e, NULL, h, level); if (r < 0) worst[k] = r;
                r = S[k].checksum(e, NULL, h, ck); if (r < 0) worst[k] = r;
                for (int fr = 0; fr < frames && worst[k] == 0; fr++) {
                    int calls = 0;
                    r = S[k].reset(e, (jobject)self, h); if (r < 0) { worst[k] = r; break; }
                    for (jsize at = 0; at < total && worst[k] == 0; ) {           /* ZstdOutputStreamNoFinalizer.write(src, at, len) */
                        jsize const len = total - at < chunk ? total - at : chunk; int guard = 0;
                        self->srcPos = at;
                        while (self->srcPos < at + len && guard++ < 100000) {
                            r = S[k].comp(e, (jobject)self, h, (jbyteArray)dst, room, (jbyteArray)src, at + len);
                            if (r < 0) { worst[k] = r; break; }
                            memcpy(out + n, dst->data, (size_t)self->dstPos); n += (size_t)self->dstPos;
                        }
                        at += len; calls++;
                        if (flushEvery && calls % flushEvery == 0 && worst[k] == 0) {
                            int guard2 = 0;
                            do { r = S[k].flush(e, (jobject)self, h, (jbyteArray)dst, room); if (r < 0) { worst[k] = r; break; } memcpy(out + n, dst->data, (size_t)self->dstPos); n += (size_t)self->dstPos; } while (r > 0 && guard2++ < 100000);
                        }
                    }
                    if (worst[k] == 0) { int guard3 = 0; do { r = S[k].end(e, (jobject)self, h, (jbyteArray)dst, room); if (r < 0) { worst[k] = r; break; } memcpy(out + n, dst->data, (size_t)self->dstPos); n += (size_t)self->dstPos; } while (r > 0 && guard3++ < 100000); }
                }
                S[k].free_(e, NULL, h);
                outs[k] = out; lens[k] = n;
            }
            CHECK(worst[0] == 0 && worst[1] == 0 && lens[0] == lens[1] && !memcmp(outs[0], outs[1], lens[0]), "output stream of %d bytes x %d frames, level %d, checksum %d, writes of %d, flush every %d, room %d: ref %zu bytes (%lld), shim %zu bytes (%lld)",
                  (int)total, frames, (int)level, (int)ck, (int)chunk, flushEvery, (int)room, lens[0], (long long)worst[0], lens[1], (long long)worst[1]);
            /* the frame(s) back through both input streams: everything in the source array, room for all of it; then in pieces of 4 000 bytes (the bundled stream's case) */
            if (worst[0] == 0 && variant != 1) for (int pieces = 0; pieces < ((streamMax == 0 && !getenv("HARNESS_PIECES")) ? 1 : 2); pieces++) {      /* (the GPU-only leg has no bundled stream for a frame that arrives in pieces) */
                Obj* fr = mk(2, (jsize)lens[0] + 1); memcpy(fr->data, outs[0], lens[0]);
                for (int k = 0; k < 2; k++) {
                    Obj* self = mk(7, 0); Obj* back = mk(2, total * frames + 64);
                    jlong const h = S[k].dcreate(e, NULL); jint r = S[k].dinit(e, (jobject)self, h);
                    jsize const all = (jsize)lens[0]; jsize fed = pieces ? 0 : all; int guard = 0;
                    self->srcPos = 0; self->dstPos = 0;
                    do {
                        if (pieces && self->srcPos == fed && fed < all) fed = fed + 4000 < all ? fed + 4000 : all;
                        r = S[k].dstream(e, (jobject)self, h, (jbyteArray)back, total * frames + 64, (jbyteArray)fr, fed);
                    } while (r >= 0 && (self->srcPos < all) && guard++ < 100000);
                    CHECK(r == 0 && self->dstPos == (jlong)total * frames && self->srcPos == all && !memcmp(back->data, src->data, (size_t)total) && (frames == 1 || !memcmp(back->data + total, src->data, (size_t)total)),
                          "input stream (library %d, %s) of the %d-byte x %d stream: ret %d, %lld bytes out, %lld consumed", k, pieces ? "pieces" : "whole", (int)total, frames, (int)r, (long long)self->dstPos, (long long)self->srcPos);
                    S[k].dfree(e, NULL, h);
                }
            }
            /* ... and through ZstdBufferDecompressingStreamNoFinalizer (byte[] + offsets, consumed / produced): whole, then in pieces */
            if (worst[0] == 0 && variant != 1) for (int pieces = 0; pieces < ((streamMax == 0 && !getenv("HARNESS_PIECES")) ? 1 : 2); pieces++) {
                typedef jlong (*bcreate_fn)(JNIEnv*, jclass); typedef jlong (*bfree_fn)(JNIEnv*, jclass, jlong); typedef jlong (*binit_fn)(JNIEnv*, jobject, jlong);
                typedef jlong (*bdec_fn)(JNIEnv*, jobject, jlong, jbyteArray, jint, jint, jbyteArray, jint, jint);
                Obj* fr = mk(2, (jsize)lens[0] + 9); memcpy(fr->data + 5, outs[0], lens[0]);          /* the frame at offset 5 of its array */
                for (int k = 0; k < 2; k++) {
                    bcreate_fn bc = (bcreate_fn)dlsym(libs[k]->h, P "ZstdBufferDecompressingStreamNoFinalizer_createDStreamNative");
                    bfree_fn bf = (bfree_fn)dlsym(libs[k]->h, P "ZstdBufferDecompressingStreamNoFinalizer_freeDStreamNative");
                    binit_fn bi = (binit_fn)dlsym(libs[k]->h, P "ZstdBufferDecompressingStreamNoFinalizer_initDStreamNative");
                    bdec_fn bd = (bdec_fn)dlsym(libs[k]->h, P "ZstdBufferDecompressingStreamNoFinalizer_decompressStreamNative");
                    CHECK(bc && bf && bi && bd, "buffer-decompress natives of library %d", k);
                    if (!(bc && bf && bi && bd)) continue;
                    Obj* self = mk(7, 0); Obj* back = mk(2, total * frames + 64 + 3);
                    jlong const h = bc(e, NULL); jlong r = bi(e, (jobject)self, h);
                    jsize const all = (jsize)lens[0]; jsize used = 0, got = 0, fed = pieces ? 0 : all; int guard = 0;
                    do {
                        if (pieces && used == fed && fed < all) fed = fed + 4000 < all ? fed + 4000 : all;
                        self->consumed = self->produced = 0;
                        r = bd(e, (jobject)self, h, (jbyteArray)back, 3 + got, total * frames + 64 - got, (jbyteArray)fr, 5 + used, fed - used);
                        used += self->consumed; got += self->produced;
                    } while (r >= 0 && used < all && guard++ < 100000);
                    CHECK(r == 0 && got == total * frames && used == all && !memcmp(back->data + 3, src->data, (size_t)total) && (frames == 1 || !memcmp(back->data + 3 + total, src->data, (size_t)total)),
                          "buffer-decompress stream (library %d, %s) of the %d-byte x %d stream: ret %lld, %d bytes out, %d consumed", k, pieces ? "pieces" : "whole", (int)total, frames, (long long)r, (int)got, (int)used);
                    bf(e, NULL, h);
                }
            }
            free(outs[0]); free(outs[1]);
        }
        /* parameters set INSIDE a frame (class Zstd's natives on the raw handle; the Java stream classes refuse it, the natives do not — ADVICE r04): the checksum flag and a
         * dictionary answer ZSTD_error_stage_wrong, the level is accepted and belongs to the NEXT frame (C/zstd_compress.c ZSTD_CCtx_setParameter past zcss_init); both libraries
         * give the same answers and the same two frames */
        {
            typedef jint (*dict_fn)(JNIEnv*, jclass, jlong, jbyteArray, jint);
            jsize const total = 50000; Obj* src = mk(2, total); Obj* dict = mk(2, 4096);
            char* outs[2]; size_t lens[2] = {0, 0}; jint answers[2][3] = {{1, 1, 1}, {1, 1, 1}}; jlong worst[2] = {0, 0};
            STAGE("heap-array streams: parameters inside a frame");
            fill(src->data, total, 1); fill(dict->data, 4096, 1);
            for (int k = 0; k < 2; k++) {
                dict_fn loadDict = (dict_fn)dlsym(libs[k]->h, P "Zstd_loadDictCompress");
                Obj* self = mk(7, 0); Obj* dst = mk(2, 131591);
                jlong const h = S[k].create(e, NULL); jint r;
                char* out = (char*)malloc(4 * (size_t)total); size_t n = 0;
                CHECK(loadDict != NULL, "Zstd_loadDictCompress of library %d", k);
                r = S[k].level(e, NULL, h, 1); if (r < 0) worst[k] = r;
                for (int fr = 0; fr < 2 && worst[k] == 0; fr++) {
                    int guard = 0;
                    r = S[k].reset(e, (jobject)self, h); if (r < 0) { worst[k] = r; break; }
                    self->srcPos = 0;
                    while (self->srcPos < total && guard++ < 100000) {
                        r = S[k].comp(e, (jobject)self, h, (jbyteArray)dst, 131591, (jbyteArray)src, total); if (r < 0) { worst[k] = r; break; }
                        memcpy(out + n, dst->data, (size_t)self->dstPos); n += (size_t)self->dstPos;
                    }
                    if (fr == 0 && worst[k] == 0) {
                        answers[k][0] = S[k].checksum(e, NULL, h, JNI_TRUE);
                        answers[k][1] = S[k].level(e, NULL, h, 3);
                        answers[k][2] = loadDict ? loadDict(e, NULL, h, (jbyteArray)dict, 4096) : 1;
                    }
                    if (worst[k] == 0) { int guard3 = 0; do { r = S[k].end(e, (jobject)self, h, (jbyteArray)dst, 131591); if (r < 0) { worst[k] = r; break; } memcpy(out + n, dst->data, (size_t)self->dstPos); n += (size_t)self->dstPos; } while (r > 0 && guard3++ < 100000); }
                }
                S[k].free_(e, NULL, h);
                outs[k] = out; lens[k] = n;
            }
            CHECK(answers[0][0] == -60 && answers[0][1] == 3 && answers[0][2] == -60, "the reference inside a frame: checksum %d, level %d, dictionary %d", (int)answers[0][0], (int)answers[0][1], (int)answers[0][2]);
            CHECK(!memcmp(answers[0], answers[1], sizeof answers[0]), "parameters inside a frame: the shim answers checksum %d, level %d, dictionary %d", (int)answers[1][0], (int)answers[1][1], (int)answers[1][2]);
            CHECK(worst[0] == 0 && worst[1] == 0 && lens[0] == lens[1] && !memcmp(outs[0], outs[1], lens[0]), "two frames around parameters set inside the first: ref %zu bytes (%lld), shim %zu bytes (%lld)", lens[0], (long long)worst[0], lens[1], (long long)worst[1]);
            free(outs[0]); free(outs[1]);
        }
        if (getenv("ZSTD_JNI_CPU_LIB")) {
            /* a level set inside a frame the GPU route is buffering, and the stream then OUTGROWS the level's window (level 1: 512 KiB) — the frame is replayed into the
             * bundled library, which must recompress it at the level the frame STARTED with (the setter is held back until the replay is done); the next frame runs at the new level */
            jsize const first = 50000, more = 700000; Obj* src = mk(2, first); Obj* big = mk(2, more);
            char* outs[2]; size_t lens[2] = {0, 0}; jlong worst[2] = {0, 0}; jint ans[2] = {-1, -1};
            STAGE("heap-array streams: a level set inside a frame that is replayed later");
            fill(src->data, first, 1); fill(big->data, more, 2);
            for (int k = 0; k < 2; k++) {
                Obj* self = mk(7, 0); Obj* dst = mk(2, 131591);
                jlong const h = S[k].create(e, NULL); jint r;
                char* out = (char*)malloc(4 * (size_t)(first + more)); size_t n = 0;
                r = S[k].level(e, NULL, h, 1); if (r < 0) worst[k] = r;
                for (int fr = 0; fr < 2 && worst[k] == 0; fr++) {
                    int guard = 0;
                    r = S[k].reset(e, (jobject)self, h); if (r < 0) { worst[k] = r; break; }
                    self->srcPos = 0;
                    while (self->srcPos < first && guard++ < 100000) {
                        r = S[k].comp(e, (jobject)self, h, (jbyteArray)dst, 131591, (jbyteArray)src, first); if (r < 0) { worst[k] = r; break; }
                        memcpy(out + n, dst->data, (size_t)self->dstPos); n += (size_t)self->dstPos;
                    }
                    if (fr == 0 && worst[k] == 0) {
                        ans[k] = S[k].level(e, NULL, h, 3);
                        self->srcPos = 0; guard = 0;
                        while (self->srcPos < more && guard++ < 100000) {
                            r = S[k].comp(e, (jobject)self, h, (jbyteArray)dst, 131591, (jbyteArray)big, more); if (r < 0) { worst[k] = r; break; }
                            memcpy(out + n, dst->data, (size_t)self->dstPos); n += (size_t)self->dstPos;
                        }
                    }
                    if (worst[k] == 0) { int guard3 = 0; do { r = S[k].end(e, (jobject)self, h, (jbyteArray)dst, 131591); if (r < 0) { worst[k] = r; break; } memcpy(out + n, dst->data, (size_t)self->dstPos); n += (size_t)self->dstPos; } while (r > 0 && guard3++ < 100000); }
                }
                S[k].free_(e, NULL, h);
                outs[k] = out; lens[k] = n;
            }
            CHECK(ans[0] == 3 && ans[1] == 3, "level inside the frame: ref %d shim %d", (int)ans[0], (int)ans[1]);
            CHECK(worst[0] == 0 && worst[1] == 0 && lens[0] == lens[1] && !memcmp(outs[0], outs[1], lens[0]), "a replayed frame keeps the level it started with: ref %zu bytes (%lld), shim %zu bytes (%lld)", lens[0], (long long)worst[0], lens[1], (long long)worst[1]);
            free(outs[0]); free(outs[1]);
        }
        /* HARNESS_FUZZ: random scripts on one ZstdOutputStreamNoFinalizer object — one to three frames (resetCStream between them), writes of random sizes through the
         * Java class's loop on srcPos, flushes (repeated ones too), a target array of a few bytes or ZSTD_CStreamOutSize(); what both libraries hand out must be the same
         * bytes; then the frames back through ZstdInputStreamNoFinalizer, whole or (with a bundled stream behind) in random pieces into a target of random size */
        if (getenv("HARNESS_FUZZ") && !getenv("HARNESS_FUZZ_SKIP_HEAP")) {
            unsigned long long seed = 1; int iters = 100; int const haveCpu = getenv("ZSTD_JNI_CPU_LIB") != NULL;
            sscanf(getenv("HARNESS_FUZZ"), "%llu,%d", &seed, &iters);
            g_x = 0xC2B2AE3D27D4EB4Full ^ (seed * 0x9E3779B97F4A7C15ull); if (!g_x) g_x = 1;
            STAGE("heap-array streams: random scripts");
            for (int it = 0; it < iters; it++) {
                int const nFrames = 1 + (int)(rnd() % 3u), level = 1 + (int)(rnd() % (haveCpu ? 4u : 3u)), cls = (int)(rnd() % 3u);
                jboolean const ck = (rnd() & 1u) ? JNI_TRUE : JNI_FALSE;
                jsize const room = (rnd() & 1u) ? (jsize)(1 + rnd() % 3000u) : 131591;
                jsize const poolN = 400000; jsize frameLen[3]; jsize grand = 0;
                for (int f = 0; f < nFrames; f++) {
                    unsigned const pick = rnd() % 10u;
                    frameLen[f] = pick == 0 ? 0 : (pick < 4 ? (jsize)(rnd() % 3000u) : (pick < 8 ? (jsize)(rnd() % 140000u) : (jsize)(rnd() % 330000u)));
                    grand += frameLen[f];
                }
                Obj* src = mk(2, poolN); fill(src->data, poolN, cls);
                unsigned long long const afterFill = g_x;
                char* outs[2]; size_t lens[2] = {0, 0}; jlong worst[2] = {0, 0}; jsize bases[3] = {0, 0, 0};
                for (int k = 0; k < 2; k++) {
                    Obj* self = mk(7, 0); Obj* dst = mk(2, room);
                    jlong const h = S[k].create(e, NULL); jint r;
                    size_t const cap = (size_t)grand + (size_t)grand / 32 + (1u << 16) + 4096u * (size_t)nFrames; char* out = (char*)malloc(cap); size_t n = 0;
                    g_x = afterFill;
                    r = S[k].level(e, NULL, h, level); if (r < 0) worst[k] = r;
                    r = S[k].checksum(e, NULL, h, ck); if (r < 0) worst[k] = r;
#define TAKE() do { if (n + (size_t)self->dstPos > cap) { worst[k] = -998; break; } memcpy(out + n, dst->data, (size_t)self->dstPos); n += (size_t)self->dstPos; } while (0)
                    for (int f = 0; f < nFrames && worst[k] == 0; f++) {
                        jsize const base = (jsize)(rnd() % (unsigned)(poolN - frameLen[f] + 1)), end = base + frameLen[f];
                        bases[f] = base;
                        r = S[k].reset(e, (jobject)self, h); if (r < 0) { worst[k] = r; break; }
                        for (jsize at = base; worst[k] == 0; ) {
                            unsigned const q = rnd() % 16u; int guard = 0;
                            jsize len = q == 0 ? 0 : (q < 6 ? (jsize)(rnd() % 2000u) : (q < 13 ? (jsize)(rnd() % 70000u) : (jsize)(rnd() % 300000u)));
                            if (len > end - at) len = end - at;
                            self->srcPos = at;
                            while (self->srcPos < at + len && guard++ < 200000) {
                                r = S[k].comp(e, (jobject)self, h, (jbyteArray)dst, room, (jbyteArray)src, at + len);
                                if (r < 0) { worst[k] = r; break; }
                                TAKE();
                            }
                            at += len;
                            if (worst[k] == 0 && (rnd() % 10u) < 3u) for (int again = 0; again < 1 + (int)((rnd() & 3u) == 0) && worst[k] == 0; again++) {
                                int guard2 = 0;
                                do { r = S[k].flush(e, (jobject)self, h, (jbyteArray)dst, room); if (r < 0) { worst[k] = r; break; } TAKE(); } while (r > 0 && guard2++ < 200000);
                            }
                            if (at >= end && (len > 0 || (rnd() & 1u))) break;
                        }
                        if (worst[k] == 0) { int guard3 = 0; do { r = S[k].end(e, (jobject)self, h, (jbyteArray)dst, room); if (r < 0) { worst[k] = r; break; } TAKE(); } while (r > 0 && guard3++ < 200000); }
                    }
#undef TAKE
                    S[k].free_(e, NULL, h);
                    outs[k] = out; lens[k] = n;
                    free(dst->data); free(dst);
                }
                CHECK(worst[0] == 0 && worst[1] == 0 && lens[0] == lens[1] && !memcmp(outs[0], outs[1], lens[0]), "random output-stream script %d (seed %llu): %d frames, level %d ck %d, room %d: ref %zu bytes (%lld), shim %zu bytes (%lld)",
                      it, seed, nFrames, level, (int)ck, (int)room, lens[0], (long long)worst[0], lens[1], (long long)worst[1]);
                if (worst[0] == 0 && lens[0] > 0) {
                    jsize const all = (jsize)lens[0];
                    Obj* fr = mk(2, all + 1); memcpy(fr->data, outs[0], lens[0]);
                    unsigned long long const cut = g_x;
                    for (int k = 0; k < 2; k++) {
                        jsize const backN = (haveCpu || getenv("HARNESS_PIECES")) && (rnd() & 1u) ? (jsize)(1 + rnd() % 5000u) : grand + 64;      /* a target smaller than the content: the read loop empties it and comes back */
                        Obj* self = mk(7, 0); Obj* back = mk(2, backN); char* got = (char*)malloc((size_t)grand + 64); size_t gotN = 0;
                        jlong const h = S[k].dcreate(e, NULL); jint r = S[k].dinit(e, (jobject)self, h);
                        jsize fed = 0; int guard = 0;
                        g_x = cut; (void)rnd();
                        self->srcPos = 0; self->dstPos = 0;
                        do {
                            if (self->srcPos == fed && fed < all) fed = ((haveCpu || getenv("HARNESS_PIECES")) && (rnd() & 1u)) ? fed + 1 + (jsize)(rnd() % (unsigned)(all - fed)) : all;
                            self->dstPos = 0;
                            r = S[k].dstream(e, (jobject)self, h, (jbyteArray)back, backN, (jbyteArray)fr, fed);
                            if (r >= 0 && gotN + (size_t)self->dstPos <= (size_t)grand + 64) { memcpy(got + gotN, back->data, (size_t)self->dstPos); gotN += (size_t)self->dstPos; }
                        } while (r >= 0 && (self->srcPos < all || r > 0) && guard++ < 400000);
                        {   int same = gotN == (size_t)grand; size_t o = 0;
                            for (int f = 0; f < nFrames && same; f++) { same = !memcmp(got + o, src->data + bases[f], (size_t)frameLen[f]); o += (size_t)frameLen[f]; }
                            CHECK(r == 0 && same && self->srcPos == all, "random script %d (seed %llu): input stream of library %d: ret %d, %zu of %d bytes out, %lld of %d consumed", it, seed, k, (int)r, gotN, (int)grand, (long long)self->srcPos, (int)all); }
                        S[k].dfree(e, NULL, h); free(back->data); free(back); free(got);
                    }
                    free(fr->data); free(fr);
                }
                free(outs[0]); free(outs[1]); free(src->data); free(src);
            }
            printf("JNI-HARNESS FUZZ (heap-array streams) seed=%llu scripts=%d\n", seed, iters);
        }
    }
    /* batch natives refuse what the per-buffer natives refuse: a null or non-direct element is an error code, not a crash */
    if (!getenv("HARNESS_SKIP_BATCH")) {
        Obj* srcs = mk(3, 3); Obj* dsts = mk(3, 3); Obj* res = mk(4, 3);
        srcs->elems = (Obj**)calloc(3, sizeof(Obj*)); dsts->elems = (Obj**)calloc(3, sizeof(Obj*));
        for (int i = 0; i < 3; i++) { srcs->elems[i] = mk(1, 100); fill(srcs->elems[i]->data, 100, 0); dsts->elems[i] = mk(1, 200); }
        g_deleted = 0;
        CHECK(G.cBatch(e, NULL, (jobjectArray)srcs, (jobjectArray)dsts, (jlongArray)res, 1, JNI_FALSE) == 0 && g_deleted == 6, "batch of 3 (local references released: %d)", g_deleted);
        srcs->elems[1] = NULL;
        CHECK(G.cBatch(e, NULL, (jobjectArray)srcs, (jobjectArray)dsts, (jlongArray)res, 1, JNI_FALSE) == -72, "batch with a null source element");
        srcs->elems[1] = mk(1, 100); dsts->elems[2] = NULL;
        CHECK(G.cBatch(e, NULL, (jobjectArray)srcs, (jobjectArray)dsts, (jlongArray)res, 1, JNI_FALSE) == -70, "batch with a null destination element");
        dsts->elems[2] = mk(1, 200);
        {   Obj* heap = mk(2, 100); Obj* keep = srcs->elems[0]; srcs->elems[0] = heap;      /* a heap buffer: GetDirectBufferAddress NULL, capacity -1 */
            CHECK(G.cBatch(e, NULL, (jobjectArray)srcs, (jobjectArray)dsts, (jlongArray)res, 1, JNI_FALSE) == -72, "batch with a non-direct source element");
            srcs->elems[0] = keep; }
    }
    /* who served the hot-path natives (zjni_shim_stats): with nothing to forward to (the GPU test) every call must have been answered by the GPU
     * path; without a GPU (the CPU test) none may have been */
    {   typedef void (*stats_fn)(unsigned long long*);
        stats_fn st = (stats_fn)dlsym(G.h, "zjni_shim_stats");
        unsigned long long v[4] = {0, 0, 0, 0};
        CHECK(st != NULL, "zjni_shim_stats exported");
        if (st) {
            st(v);
            printf("JNI-HARNESS STATS served_by_gpu=%llu forwarded_by_policy=%llu forwarded_after_gpu_declined=%llu of_those_no_device=%llu\n", v[0], v[1], v[2], v[3]);
            if (getenv("HARNESS_EXPECT") && !strcmp(getenv("HARNESS_EXPECT"), "gpu")) CHECK(v[0] > 0 && v[1] == 0 && v[2] == 0, "GPU run: %llu served, %llu + %llu forwarded", v[0], v[1], v[2]);
            if (getenv("HARNESS_EXPECT") && !strcmp(getenv("HARNESS_EXPECT"), "cpu")) CHECK(v[0] == 0 && v[1] > 0, "CPU-only run: %llu served by the GPU, %llu forwarded by policy", v[0], v[1]);
        }
    }
    if (g_bad) { printf("JNI-HARNESS FAILED bad=%d checks=%d\n", g_bad, g_checks); return 1; }
    printf("JNI-HARNESS OK checks=%d\n", g_checks);
    return 0;
}
