// zj_encode.h — placeholder until the encoder lands
#pragma once
#include "zj_common.h"
#define ZE_SCRATCH_BYTES 1024u
struct ZEncShared { u32 x; };
template <class G>
ZJ_DEV u64 ze_compress(const G& g, ZEncShared& sh, const u8* src, u32 srcSize, u8* dst, u32 dstCap, u32 level, u8* ws) {
    (void)g; (void)sh; (void)src; (void)srcSize; (void)dst; (void)dstCap; (void)level; (void)ws;
    return ZJ_ERR64(201);
}
