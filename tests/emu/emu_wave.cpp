// tests/emu/emu_wave.cpp — explicit-SIMT build of the wave-per-frame matcher (zstd-jni_amd/csrc/zj_match_wave.h) for the
// CPU-side parity tests: 64 emulated lanes, lockstep between the cross-lane points.  Built twice: lanes visited in ascending
// order, and (-DZW_EMU_REVERSE) in descending order — where several lanes store to one LDS address in the same step the GPU
// lets an unspecified lane win; the two builds let the lowest and the highest lane win, and the frames must not depend on it.
// The descending build also reads the frame the way the product library does (ZW_FRAME_IN_LDS=0: clamped reads from global
// memory), the ascending one from the staged copy with unclamped reads.
// TEST INFRASTRUCTURE ONLY: never linked into libzjni_amd.so.
#include "../../zstd-jni_amd/csrc/zj_encode.h"
#include "../../zstd-jni_amd/csrc/zj_match_wave.h"
#include <stdlib.h>
#include <string.h>

// level word as in emu_compress_split (level | checksum << 8); returns ~0 when the wave matcher does not take the frame
extern "C" unsigned long long emu_compress_wave(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level) {
#ifdef EMU_EXACT       // the sanitizer build: exact-size heap copies, so that a stray read or write lands in a red zone
    struct Exact { u8* s; u8* d; u8* user; unsigned cap; ~Exact() { if (cap) memcpy(user, d, cap); free(s); free(d); } } ex;
    ex.s = (u8*)malloc(srcSize ? srcSize : 1); if (srcSize) memcpy(ex.s, src, srcSize);
    ex.d = (u8*)malloc(dstCap ? dstCap : 1); if (dstCap) memcpy(ex.d, dst, dstCap);
    ex.user = dst; ex.cap = dstCap; src = ex.s; dst = ex.d;
#endif
    Grp<1> g;
    u32 const flags = (level >> 8) & ZE_FLAG_MASK; level &= 0xFFu;
    if (level != 3 || srcSize > 65536u || !zw_takes(ze_params_of(level, srcSize), srcSize)) return ~0ull;
    ZEncShared* sh = (ZEncShared*)calloc(1, sizeof(ZEncShared));
    u8* lds = (u8*)calloc(1, 160 * 1024);
    u8* ws = (u8*)malloc(ZE_SCRATCH_BYTES);
    ZWLds* wl = (ZWLds*)malloc(sizeof(ZWLds)); memset(wl, 0xA5, sizeof(ZWLds));      // the matcher clears what it uses
    u8* fs = (u8*)malloc(ZE_FRAME_STRIDE(65536u));
    u32 meta[3];
    zw_match_frame(*wl, src, srcSize, level, fs, 65536u, meta);
    ZEPre pre; pre.seqs = (ZESeq*)fs; pre.litOff = (const u32*)(fs + (size_t)ZE_FRAME_MAXSEQ(65536u) * 16u); pre.meta = meta;
    ZjProf pf; pf.start(nullptr);
    u64 r = ze_compress(g, *sh, lds, src, srcSize, dst, dstCap, level, ws, pf, &pre, flags, nullptr, 160u * 1024u);
    free(fs); free(wl); free(ws); free(lds); free(sh);
    return r;
}
extern "C" unsigned emu_wave_lds_bytes() { return (unsigned)sizeof(ZWLds); }
