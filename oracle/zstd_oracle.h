/* oracle/zstd_oracle.h — CPU restatement of the zstd path zstd-jni reaches through
 * ZSTD_compress2 / ZSTD_decompressDCtx (reference: /root/reference/src/main/native, "N/").
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may link or dlopen this.  The product library (zstd-jni_amd/csrc) never includes, links or calls
 * anything in oracle/.
 *
 * Parity status: PINNED.
 *   - decoder: checked byte-for-byte against the reference's golden frames
 *     (N/../../test/resources/xml-{1,3,9}.zst, xml-advanced.zst, xml-sized-combined.zst,
 *     xmlsmall-sized.zst; copies under tests/golden/) and against oracle/_ref (the reference's own
 *     libzstd 1.5.7 compiled from its sources) on seeded inputs — tests/test_oracle.py.
 *   - encoder: checked BYTE-IDENTICAL against oracle/_ref's ZSTD_compress2 on seeded inputs and on
 *     xmlsmall -> xmlsmall-sized.zst — tests/test_oracle.py.
 */
#ifndef ZSTD_ORACLE_H
#define ZSTD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error convention = the reference's: (size_t)(0 - ZSTD_ErrorCode), N/zstd_errors.h:60-98. */
#define ZSO_ERR(code) ((size_t)0 - (size_t)(code))
enum {
    ZSO_error_GENERIC = 1,
    ZSO_error_prefix_unknown = 10,
    ZSO_error_frameParameter_unsupported = 14,
    ZSO_error_frameParameter_windowTooLarge = 16,
    ZSO_error_corruption_detected = 20,
    ZSO_error_checksum_wrong = 22,
    ZSO_error_literals_headerWrong = 24,
    ZSO_error_dictionary_corrupted = 30,
    ZSO_error_dictionary_wrong = 32,
    ZSO_error_parameter_unsupported = 40,
    ZSO_error_tableLog_tooLarge = 44,
    ZSO_error_maxSymbolValue_tooLarge = 46,
    ZSO_error_maxSymbolValue_tooSmall = 48,
    ZSO_error_dstSize_tooSmall = 70,
    ZSO_error_srcSize_wrong = 72,
    ZSO_error_maxCode = 120
};
static inline int zso_is_error(size_t r) { return r > ZSO_ERR(ZSO_error_maxCode); }

/* ---- decode (oracle/zstd_oracle_dec.c) ---- */
/* Decodes every frame in [src, src+srcSize) (skippable frames skipped), like ZSTD_decompressDCtx
 * (N/decompress/zstd_decompress.c:1070-1168). */
size_t zso_decompress(void* dst, size_t dstCap, const void* src, size_t srcSize);
/* ZSTD_getFrameContentSize: returns content size, (u64)-1 unknown, (u64)-2 error. */
unsigned long long zso_frame_content_size(const void* src, size_t srcSize);
/* ZSTD_findFrameCompressedSize */
size_t zso_find_frame_compressed_size(const void* src, size_t srcSize);

/* ---- encode (oracle/zstd_oracle_enc.c) ---- */
size_t zso_compress_bound(size_t srcSize);
/* One-shot ZSTD_compress2 restatement for level in [1,3], no dictionary, checksum optional,
 * srcSize <= 128 KiB (one block; larger inputs return ZSO_error_parameter_unsupported).
 * Output is byte-identical to the reference's.  Returns compressed size or error. */
size_t zso_compress(void* dst, size_t dstCap, const void* src, size_t srcSize, int level, int checksum);
/* Same with ZSTD_c_hashLog / ZSTD_c_chainLog overrides (0 = level default). */
size_t zso_compress_ex(void* dst, size_t dstCap, const void* src, size_t srcSize, int level, int checksum, int hashLog, int chainLog);

/* XXH64 (N/common/xxhash.h) — used for the optional frame checksum */
uint64_t zso_xxh64(const void* p, size_t len, uint64_t seed);
/* FSE_readNCount restated (N/common/entropy_common.c:42-188), exposed for tests */
size_t zso_read_ncount(short* norm, unsigned* maxSV, unsigned* tableLog, const void* src, size_t srcSize);

#ifdef __cplusplus
}
#endif
#endif
