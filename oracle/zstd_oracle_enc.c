/* placeholder until the encoder restatement lands (next commit) */
#include "zstd_oracle.h"
size_t zso_compress_bound(size_t s) { return s + (s >> 8) + (s < (128u << 10) ? (((128u << 10) - s) >> 11) : 0); }
size_t zso_compress(void* dst, size_t dstCap, const void* src, size_t srcSize, int level, int checksum) {
    (void)dst; (void)dstCap; (void)src; (void)srcSize; (void)level; (void)checksum; return ZSO_ERR(ZSO_error_GENERIC);
}
