// emu_abi.cpp — TEST DOUBLE of libzjni_amd.so for the JNI library's tests on machines without a GPU (test infrastructure, never shipped, never on the product path).
//
// The JNI library (zstd-jni_amd/jni/zjni_shim.c) reaches the GPU through 22 entries of include/zjni_amd.h.  This file defines exactly those entries over the
// kernel BODIES compiled lane-serial under g++ (tests/emu/libzjni_emu.so: the same zj_encode.h / zj_decode.h / zj_cdict.h sources the HIP kernels run, one
// emulated lane or wave at a time), so that a copy of libzstd-jni-amd.so placed beside it (RUNPATH $ORIGIN) runs its whole "GPU route" — one-shot natives,
// dictionaries, the five stream classes, the context streams, replays into the bundled library — on the CPU, and tests/jni/harness.c can compare every native
// with the reference's JNI library in `pytest -m "not gpu"`.  What it does not emulate: launches, batching, scratch, routes (every call is a batch of one frame
// after another) — the GPU legs of tests/test_jni_shim.py cover those.  The few host-side helpers of the real library (error names, bound, frame extent) are
// restated from zj_kernels.hip, which needs hipcc.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <mutex>
#include "../../include/zjni_amd.h"

typedef unsigned long long u64;
extern "C" {
u64 emu_compress(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level);
u64 emu_compress_multi(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level);
u64 emu_compress_stream_flush(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level, const unsigned* flushAt, unsigned nFlush);
u64 emu_decompress(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap);
u64 emu_decompress_dict(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, const unsigned char* dict, unsigned dictSize);
void* emu_cdict_create(const unsigned char* dict, unsigned dictSize, unsigned level);
void emu_cdict_free(void* cd);
u64 emu_compress_cdict(const void* p, const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned flags);
}
#define ERR(code) ((size_t)0 - (size_t)(code))
static unsigned char g_nothing[16];
static const unsigned char* in_(const void* p) { return p ? (const unsigned char*)p : g_nothing; }
static unsigned char* out_(void* p) { return p ? (unsigned char*)p : g_nothing; }
static unsigned cap32(size_t c) { return c > 0xFFFFFFF0u ? 0xFFFFFFF0u : (unsigned)c; }

struct zjni_cdict { void* emu; int level; };
struct zjni_ddict { unsigned char* bytes; size_t size; };

extern "C" {
int zjni_device_count(void) { return getenv("ZJNI_EMU_NO_DEVICE") ? 0 : 1; }
int zjni_init(int device) { return device == 0 ? 0 : -1; }
unsigned zjni_isError(size_t r) { return r > ERR(256) ? 1u : 0u; }
int zjni_getErrorCode(size_t r) { return zjni_isError(r) ? (int)(0 - r) : 0; }
const char* zjni_getErrorName(size_t r) {           // libzstd's strings (N/common/error_private.c:15-65), as the real library carries them
    switch (zjni_getErrorCode(r)) {
    case 0: return "No error detected";
    case 1: return "Error (generic)";
    case 10: return "Unknown frame descriptor";
    case 14: return "Unsupported frame parameter";
    case 16: return "Frame requires too much memory for decoding";
    case 20: return "Data corruption detected";
    case 22: return "Restored data doesn't match checksum";
    case 24: return "Header of Literals' block doesn't respect format specification";
    case 30: return "Dictionary is corrupted";
    case 32: return "Dictionary mismatch";
    case 40: return "Unsupported parameter";
    case 42: return "Parameter is out of bound";
    case 44: return "tableLog requires too much memory : unsupported";
    case 64: return "Allocation error : not enough memory";
    case 70: return "Destination buffer is too small";
    case 72: return "Src size is incorrect";
    case 200: return "zjni: no gfx950 device available";
    case 201: return "zjni: input outside the GPU path (use the CPU path)";
    default: return "Unspecified error code";
    }
}
size_t zjni_compressBound(size_t s) { return s + (s >> 8) + (s < (128u << 10) ? (((128u << 10) - s) >> 11) : 0); }

static size_t one_compress(void* dst, size_t cap, const void* src, size_t n, int level, int flags, int hashLog, int chainLog) {
    if (level == 0) level = 3;
    if (level < 1 || level > 8) return ERR(42);
    if ((hashLog | chainLog) && level != 3) return ERR(40);
    if (n > (level <= 3 ? ZJNI_FRAME_MAX : (level == 4 ? ZJNI_LEVEL4_MAX : ZJNI_LAZY_MAX))) return ERR(ZJNI_ERROR_unsupported);
    if (level <= 2 && n <= ZJNI_BLOCKSIZE_MAX)          // levels 1-2, one block: the fused kernel's body (tables in LDS); level 3 takes the wave route below, whose tables in HBM have the reference's sizes
        return (size_t)emu_compress(in_(src), (unsigned)n, out_(dst), cap32(cap), (unsigned)level | ((unsigned)(flags & 7) << 8));
    if (level == 3 && n <= ZJNI_BLOCKSIZE_MAX && !(hashLog | chainLog)) { hashLog = 16; chainLog = 15; }      // level 3 with nothing set: the reference's own tables (zj_level3_word in zj_kernels.hip)
    return (size_t)emu_compress_multi(in_(src), (unsigned)n, out_(dst), cap32(cap), (unsigned)level | ((unsigned)(flags & 7) << 8) | ((unsigned)hashLog << 16) | ((unsigned)chainLog << 24));
}
size_t zjni_compress_batch_advanced(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity, size_t* result, size_t n,
                                    int level, int flags, int hashLog, int chainLog) {
    for (size_t i = 0; i < n; i++) result[i] = one_compress(dst[i], dstCapacity[i], src[i], srcSize[i], level, flags, hashLog, chainLog);
    return 0;
}
size_t zjni_compress_batch2(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity, size_t* result, size_t n, int level, int checksum) {
    return zjni_compress_batch_advanced(src, srcSize, dst, dstCapacity, result, n, level, checksum ? ZJNI_FRAME_CHECKSUM : 0, 0, 0);
}
size_t zjni_compress2(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, int checksum) {
    return one_compress(dst, dstCapacity, src, srcSize, level, checksum ? ZJNI_FRAME_CHECKSUM : 0, 0, 0);
}
size_t zjni_compress_stream(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, int checksum, const uint32_t* flushAt, size_t nFlush, int final_, int knownEmpty) {
    if (level == 0) level = 3;
    if (level < 1 || level > 3) return ERR(42);
    if (srcSize > ((size_t)1 << (18 + level))) return ERR(ZJNI_ERROR_unsupported);
    return (size_t)emu_compress_stream_flush(in_(src), (unsigned)srcSize, out_(dst), cap32(dstCapacity),
                                             (unsigned)level | (checksum ? 0x100u : 0u) | (final_ ? 0u : 0x10000u) | (knownEmpty ? 0x20000u : 0u), flushAt, (unsigned)nFlush);
}

size_t zjni_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize) {
    return (size_t)emu_decompress(in_(src), (unsigned)srcSize, out_(dst), cap32(dstCapacity));
}
size_t zjni_decompress_batch(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity, size_t* result, size_t n) {
    for (size_t i = 0; i < n; i++) result[i] = zjni_decompress(dst[i], dstCapacity[i], src[i], srcSize[i]);
    return 0;
}

zjni_cdict* zjni_createCDict(const void* dict, size_t dictSize, int level) {
    if (!dict || dictSize < 8 || level < 1 || level > 3) return nullptr;
    void* e = emu_cdict_create((const unsigned char*)dict, (unsigned)dictSize, (unsigned)level);
    if (!e) return nullptr;
    zjni_cdict* c = (zjni_cdict*)calloc(1, sizeof *c); c->emu = e; c->level = level;
    return c;
}
size_t zjni_freeCDict(zjni_cdict* c) { if (c) { emu_cdict_free(c->emu); free(c); } return 0; }
size_t zjni_compress_batch_usingCDict(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity, size_t* result, size_t n,
                                      const zjni_cdict* cdict, int flags) {
    if (!cdict) return ERR(32);
    for (size_t i = 0; i < n; i++)
        result[i] = srcSize[i] > ZJNI_BLOCKSIZE_MAX ? ERR(ZJNI_ERROR_unsupported) : (size_t)emu_compress_cdict(cdict->emu, in_(src[i]), (unsigned)srcSize[i], out_(dst[i]), cap32(dstCapacity[i]), (unsigned)flags & 7u);
    return 0;
}
zjni_ddict* zjni_createDDict(const void* dict, size_t dictSize) {
    static const unsigned char empty[9] = {0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00, 0x01, 0x00, 0x00};
    unsigned char sink[8];
    if (!dict || dictSize < 8) return nullptr;
    size_t const probe = (size_t)emu_decompress_dict(empty, 9, sink, 8, (const unsigned char*)dict, (unsigned)dictSize);      // digests the dictionary: a damaged one answers 30
    if (zjni_isError(probe)) return nullptr;
    zjni_ddict* d = (zjni_ddict*)calloc(1, sizeof *d);
    d->bytes = (unsigned char*)malloc(dictSize); memcpy(d->bytes, dict, dictSize); d->size = dictSize;
    return d;
}
size_t zjni_freeDDict(zjni_ddict* d) { if (d) { free(d->bytes); free(d); } return 0; }
size_t zjni_decompress_usingDDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const zjni_ddict* ddict) {
    if (!ddict) return zjni_decompress(dst, dstCapacity, src, srcSize);
    return (size_t)emu_decompress_dict(in_(src), (unsigned)srcSize, out_(dst), cap32(dstCapacity), ddict->bytes, (unsigned)ddict->size);
}

// no aggregation in the double: ZSTD_JNI_GPU_AGGREGATE is a launch-sharing device, and there are no launches here
// the asynchronous entries (include/zjni_amd.h: a job that runs the blocking entry on a thread of the library): here a std::thread over the double's blocking entries
struct zjni_batch_job { std::thread th; size_t code; };
static std::mutex g_emu_mu;                     // the lane-serial bodies keep their "LDS" and scratch in statics: one job's frames at a time (the real library's kernels of two jobs follow each other too)
zjni_batch_job* zjni_compress_batch_begin(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity, size_t* result, size_t n, int level, int checksum) {
    if (zjni_device_count() <= 0) return nullptr;
    zjni_batch_job* j = new zjni_batch_job(); j->code = 0;
    j->th = std::thread([=]() { std::lock_guard<std::mutex> g(g_emu_mu); j->code = zjni_compress_batch2(src, srcSize, dst, dstCapacity, result, n, level, checksum); });
    return j;
}
zjni_batch_job* zjni_decompress_batch_begin(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity, size_t* result, size_t n) {
    if (zjni_device_count() <= 0) return nullptr;
    zjni_batch_job* j = new zjni_batch_job(); j->code = 0;
    j->th = std::thread([=]() { std::lock_guard<std::mutex> g(g_emu_mu); j->code = zjni_decompress_batch(src, srcSize, dst, dstCapacity, result, n); });
    return j;
}
size_t zjni_batch_finish(zjni_batch_job* j) {
    if (!j) return ERR(ZJNI_ERROR_no_device);
    j->th.join(); size_t const c = j->code; delete j; return c;
}
zjni_aggregator* zjni_createAggregator(int, size_t, unsigned) { return nullptr; }
size_t zjni_aggregator_compress(zjni_aggregator*, void*, size_t, const void*, size_t, int, int) { return ERR(ZJNI_ERROR_no_device); }
size_t zjni_aggregator_decompress(zjni_aggregator*, void*, size_t, const void*, size_t) { return ERR(ZJNI_ERROR_no_device); }

// zjni_frame_extent: as zj_kernels.hip (host_frame_extent + the bound of N/decompress/zstd_decompress.c:739-850)
static uint32_t ld32(const unsigned char* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
size_t zjni_frame_extent(const void* srcv, size_t n, unsigned long long* content, unsigned long long* bound) {
    const unsigned char* p = (const unsigned char*)srcv; u64 c = ~(u64)0, b = 0;
    if (content) *content = c;
    if (bound) *bound = 0;
    if (!p || n < 9 || ld32(p) != 0xFD2FB528u) return 0;
    unsigned const fhd = p[4], didc = fhd & 3, single = (fhd >> 5) & 1, fcsid = fhd >> 6, cks = (fhd >> 2) & 1;
    if (fhd & 8) return 0;
    unsigned const didSz = didc == 3 ? 4 : didc, fcsSz = fcsid == 0 ? single : (1u << fcsid);
    size_t pos = 5 + !single + didSz;
    if (n < pos + fcsSz + 3) return 0;
    if (fcsSz) { c = 0; for (unsigned i = 0; i < fcsSz; i++) c |= (u64)p[pos + i] << (8 * i); if (fcsid == 1) c += 256; }
    pos += fcsSz;
    u64 blockMax = 128u << 10;
    if (!single) { unsigned const wl = (p[5] >> 3) + 10u; if (wl > 31u) return 0; u64 const w = ((u64)1 << wl) + (((u64)1 << wl) >> 3) * (p[5] & 7u); if (w < blockMax) blockMax = w; }
    else if (c < blockMax) blockMax = c;
    for (;;) {
        if (pos + 3 > n) return 0;
        unsigned const bh = (unsigned)p[pos] | ((unsigned)p[pos + 1] << 8) | ((unsigned)p[pos + 2] << 16), type = (bh >> 1) & 3, bs = bh >> 3;
        if (type == 3) return 0;
        b += type == 2 ? blockMax : bs;
        pos += 3 + (type == 1 ? 1 : bs);
        if (pos > n) return 0;
        if (bh & 1) break;
    }
    pos += cks ? 4 : 0;
    if (pos > n) return 0;
    if (content) *content = c;
    if (bound) *bound = (c != ~(u64)0) ? c : b;
    return pos;
}
}
