// zj_decode.h — batched zstd frame decoder for gfx950: one frame per wavefront.
//
// Replaces, for batches of independent frames, what zstd-jni reaches through
//   ZstdDecompressCtx.decompress*0 -> ZSTD_decompressDCtx          (reference N/jni_fast_zstd.c:777-905)
// i.e. N/decompress/zstd_decompress.c:953-1066 (frame), N/decompress/zstd_decompress_block.c:134-340
// (literals), :695-782 (sequence headers), :485-603 (FSE tables), :1229-1347 + :1001-1096
// (sequence decode + LZ77 execute), N/decompress/huf_decompress.c:385-518 (Huffman table) and
// N/common/entropy_common.c:42-188,243-305 (NCount / weights).   N/ = src/main/native/.
//
// Data flow per frame (all compressed bytes are read from HBM exactly once, all output bytes written
// exactly once, plus one round trip of the Huffman literals through a per-workgroup HBM scratch that
// stays L2-resident):
//   HBM src --(16 B/lane coalesced)--> LDS windows --> serial tANS / Huffman lanes --> LDS batches
//   --> wave-cooperative LZ77 copies --> HBM dst
// LDS per workgroup (= 1 wave): 3 tANS tables (5 KiB, 4-byte cells), Huffman table (4 KiB),
// bitstream windows (1 KiB), sequence batch + scratch (~2.5 KiB)  => ~13 KiB => 12 frames in flight
// per CU.
#pragma once
#include "zj_common.h"

#define ZD_BLOCK_MAX (1u << 17)
#define ZD_HUF_LOG_MAX 12u          // deepest literals Huffman table the reference's decoder takes (HUF_TABLELOG_MAX, N/common/huf.h:37; ZSTD_HUFFDTABLE_CAPACITY_LOG,
                                    // N/decompress/zstd_decompress_internal.h:78) — its encoder stops at 11 (LitHufLog, N/common/zstd_internal.h:101), other writers need not
#define ZD_HUF_CELLS_LOG 11u        // ... in 2^11 cells of LDS whatever the depth: a 12-bit table is kept as its PAIRS of slots (zd_huf_fill)
#define ZD_HWIN 256u                // bytes of bitstream staged per Huffman stream per round
#define ZD_HSYM 128u                // symbols decoded per stream per round (128*11 bits <= 176 B < ZD_HWIN)
#define ZD_HWIN_STRIDE (ZD_HWIN + 16u)
#define ZD_SWIN (4u * ZD_HWIN_STRIDE - 16u)   // sequence bitstream window reuses the 4 Huffman windows
#define ZD_SEQ_BATCH 64u
#define ZD_LIT_SCRATCH (ZD_BLOCK_MAX + 64u)

// tANS decode cell: next[0:15] | nbBits[16:19] | symbol[20:25] | extraBits[26:30]
#define ZD_CELL(next, nb, sym, extra) ((u32)(next) | ((u32)(nb) << 16) | ((u32)(sym) << 20) | ((u32)(extra) << 26))
#define ZD_CELL_NEXT(c) ((c) & 0xFFFFu)
#define ZD_CELL_NB(c) (((c) >> 16) & 0xFu)
#define ZD_CELL_SYM(c) (((c) >> 20) & 0x3Fu)
#define ZD_CELL_EXTRA(c) (((c) >> 26) & 0x1Fu)
// the split pipeline's 2-byte form of a cell (zj_decode_split.h): symbol | tANS counter << 6, counter = (next + tableSize) >> nbBits
ZJ_DEV u16 zd_cell16(u32 c4, u32 log) { return (u16)(ZD_CELL_SYM(c4) | (((ZD_CELL_NEXT(c4) + (1u << log)) >> ZD_CELL_NB(c4)) << 6)); }

struct ZDecShared {
    u16 huf[1u << ZD_HUF_CELLS_LOG];
    u32 llBase[36];                 // LL_base (N/decompress/zstd_decompress_internal.h) per code
    u32 mlBase[53];
    u8 win[4 * ZD_HWIN_STRIDE];     // bitstream windows
    u8 hstage[4][ZD_HSYM];          // Huffman output staging
    u32 sLit[ZD_SEQ_BATCH];         // per-batch sequences
    u32 sMl[ZD_SEQ_BATCH];
    u32 sOff[ZD_SEQ_BATCH];
    u8 weights[256];
    u32 hrank[32];
    u16 symPos[256];                // Huffman: first table cell of each symbol
    // --- uniforms published by lane 0 ---
    u32 err;
    u32 llLog, mlLog, ofLog, hufLog, hufValid, seqValid;
    u32 hufX2;                      // the reference would decode with its two-code-per-cell table (matters only for corrupted streams)
    u32 hufW1;                      // 12-bit table: its weight-1 symbols (the slots below this hold one symbol each, two to a cell; zd_huf_fill), else 0
    u32 rep[3];
    u32 blkType, blkSize, blkLast;
    u32 hdrSize, windowSize, hasChecksum, blockSizeMax;
    u64 contentSize;
    u32 litType, litSize, litCSize, litHdr, litStreams, litSrcOff;  // literals section
    u32 hA[4], hS0[4], hN[4], hDone[4], hDst[4], hLo[4], hCnt[4], hA0[4];    // Huffman stream state (bit positions rel. to block)
    u32 nbSeq, seqOff;              // sequences section start (rel. to block)
    u32 bN, bLitStart, bOutStart, bLitTotal, bOutTotal, seqDone, winLo;
    u32 tblOff[3], tblMode[3], tblLog[3], tblMax[3];
    u32 dictSize;                   // bytes of dictionary content before the frame's output (0 = none)
    u32 hpAgain;                    // zd_huf_streams_wave: a lane's start moved / the streams check out
    u32 hufGcd;                     // greatest common divisor of the table's code lengths (1 for a table that came from a dictionary)
    // --- tANS tables last: the execute-only kernel of the split pipeline allocates the struct without them ---
    short norm[3][64];              // NCount of the LL / OF / ML table being described
    u16 symNext[64 * 3];
    u32 ll[512];
    u32 ml[512];
    u32 of[256];
};
#define ZD_SHARED_NO_FSE (sizeof(ZDecShared) - (512u + 512u + 256u) * 4u - 3u * 64u * 2u - 64u * 3u * 2u)
#define ZD_HP_LDS_OFF ((ZD_SHARED_NO_FSE + 15u) & ~15u)      // ... and behind it the lanes' windows of zd_huf_streams_wave
#define ZD_EXEC_LDS (ZD_HP_LDS_OFF + 64u * 72u)             // = ZD_HP_LDS (defined with the pass below)

// format constants (N/common/zstd_internal.h:113-165, N/decompress/zstd_decompress_internal.h:28-58)
#define ZD_LL_BASE_INIT { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,0x80,0x100,0x200,0x400,0x800,0x1000,0x2000,0x4000,0x8000,0x10000 }
#define ZD_LL_BITS_INIT { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 }
#define ZD_ML_BASE_INIT { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003 }
#define ZD_ML_BITS_INIT { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 }
#define ZD_LL_DEFNORM_INIT { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 }
#define ZD_ML_DEFNORM_INIT { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 }
#define ZD_OF_DEFNORM_INIT { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 }

#if ZJ_ON_GPU
#define ZD_CONST static __device__ const
#else
#define ZD_CONST static const
#endif
ZD_CONST u8 zd_k_ll_bits[36] = ZD_LL_BITS_INIT;
ZD_CONST u8 zd_k_ml_bits[53] = ZD_ML_BITS_INIT;
ZD_CONST u32 zd_k_ll_base[36] = ZD_LL_BASE_INIT;
ZD_CONST u32 zd_k_ml_base[53] = ZD_ML_BASE_INIT;
ZD_CONST short zd_k_ll_defnorm[36] = ZD_LL_DEFNORM_INIT;
ZD_CONST short zd_k_ml_defnorm[53] = ZD_ML_DEFNORM_INIT;
ZD_CONST short zd_k_of_defnorm[29] = ZD_OF_DEFNORM_INIT;
ZJ_DEV u32 zd_ll_bits(u32 c) { return zd_k_ll_bits[c]; }
ZJ_DEV u32 zd_ml_bits(u32 c) { return zd_k_ml_bits[c]; }

// ------------------------------------------------------------------ wide cooperative copy ----
template <class G>
ZJ_DEV void zd_fill(const G& g, u8* dst, u8 v, u32 n) {
    u64 const vv = 0x0101010101010101ull * v;
    u32 const n16 = n >> 4;
    GRP_FOR(g, i, n16) { st64(dst + 16 * i, vv); st64(dst + 16 * i + 8, vv); }
    GRP_FOR(g, i, n & 15u) dst[(n16 << 4) + i] = v;
}

// stage `len` (<= capacity) bytes of src[from, from+len) into LDS at `w`, zero past srcEnd
template <class G>
ZJ_DEV void zd_stage(const G& g, u8* w, const u8* src, u32 from, u32 len, u32 srcEnd) {
    GRP_FOR(g, c, (len + 15u) >> 4) {
        u32 const o = from + 16 * c;
        u64 a = 0, b = 0;
        if (o + 16 <= srcEnd) { a = ld64(src + o); b = ld64(src + o + 8); }
        else {
            for (u32 k = 0; k < 8; k++) { if (o + k < srcEnd) a |= (u64)src[o + k] << (8 * k); }
            for (u32 k = 0; k < 8; k++) { if (o + 8 + k < srcEnd) b |= (u64)src[o + 8 + k] << (8 * k); }
        }
        st64(w + 16 * c, a); st64(w + 16 * c + 8, b);
    }
}

// ------------------------------------------------------------------ NCount (lane 0) ---------
// N/common/entropy_common.c:42-188.  Reads from global memory [src, src+srcSize). Returns header
// bytes or 0 on error (an NCount header is never 0 bytes).
// The cursor is the reference's own — a byte position, a bit count and a 32-bit word re-read after every field — because its
// behaviour at the end of the buffer is part of the contract: the position is pinned 4 bytes before the end and the bit count
// taken modulo 32 (:146-153, :170-177), so a description that runs past its buffer WRAPS AROUND on the last four bytes
// instead of failing; only the bit count after the last field is tested (:184).  Damaged blocks reach that, and whether they
// are then refused — and with which code — depends on it.  Buffers under 8 bytes are parsed from a zero-padded copy (:62-72).
template <class RD32>
ZJ_DEV u32 zd_read_ncount_body(short* norm, u32* maxSV, u32* tableLog, RD32 rd32, u32 hbSize) {      // hbSize >= 8
    u32 const maxSV1 = *maxSV + 1;
    i32 const iend = (i32)hbSize;
    i32 ip = 0, bitCount = 4;
    u32 charnum = 0, previous0 = 0;
    u32 bitStream = rd32(0);
    i32 nbBits = (i32)(bitStream & 0xF) + 5, remaining, threshold;
    if (nbBits > 15) return 0;
    ZJ_NO_UNROLL             // a constant limit makes the compiler unroll this into a register-hungry store burst
    for (u32 s = 0; s < maxSV1; s++) norm[s] = 0;
    bitStream >>= 4;
    *tableLog = (u32)nbBits;
    remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;
    for (;;) {
        if (previous0) {
            // pairs of 1-bits visible in the current word: each is three more zero-probability symbols
            u32 repeats = (u32)zj_ctz32(~bitStream | 0x80000000u) >> 1;
            while (repeats >= 12) {
                charnum += 3 * 12;
                if (ip <= iend - 7) ip += 3;
                else { bitCount -= 8 * (iend - 7 - ip); bitCount &= 31; ip = iend - 4; }
                bitStream = rd32((u32)ip) >> bitCount;
                repeats = (u32)zj_ctz32(~bitStream | 0x80000000u) >> 1;
            }
            charnum += 3 * repeats;
            bitStream >>= 2 * repeats; bitCount += (i32)(2 * repeats);
            charnum += bitStream & 3; bitCount += 2;
            if (charnum >= maxSV1) break;
            if (ip <= iend - 7 || ip + (bitCount >> 3) <= iend - 4) { ip += bitCount >> 3; bitCount &= 7; }
            else { bitCount -= 8 * (iend - 4 - ip); bitCount &= 31; ip = iend - 4; }
            bitStream = rd32((u32)ip) >> bitCount;
        }
        {   i32 const max = (2 * threshold - 1) - remaining;
            i32 count;
            if ((bitStream & (u32)(threshold - 1)) < (u32)max) { count = (i32)(bitStream & (u32)(threshold - 1)); bitCount += nbBits - 1; }
            else { count = (i32)(bitStream & (u32)(2 * threshold - 1)); if (count >= threshold) count -= max; bitCount += nbBits; }
            count--;
            if (count >= 0) remaining -= count; else remaining += count;
            norm[charnum++] = (short)count;
            previous0 = !count;
            if (remaining < threshold) {
                if (remaining <= 1) break;
                nbBits = (i32)zj_hibit((u32)remaining) + 1; threshold = 1 << (nbBits - 1);
            }
            if (charnum >= maxSV1) break;
            if (ip <= iend - 7 || ip + (bitCount >> 3) <= iend - 4) { ip += bitCount >> 3; bitCount &= 7; }
            else { bitCount -= 8 * (iend - 4 - ip); bitCount &= 31; ip = iend - 4; }
            bitStream = rd32((u32)ip) >> bitCount;
        }
    }
    if (remaining != 1 || charnum > maxSV1 || bitCount > 32) return 0;
    *maxSV = charnum - 1;
    return (u32)(ip + ((bitCount + 7) >> 3));
}
ZJ_DEV u32 zd_read_ncount(short* norm, u32* maxSV, u32* tableLog, const u8* src, u32 srcSize) {
    if (srcSize >= 8) return zd_read_ncount_body(norm, maxSV, tableLog, [&](u32 p) { return ld32(src + p); }, srcSize);
    u64 pad = 0;
    for (u32 k = 0; k < srcSize; k++) pad |= (u64)src[k] << (8 * k);
    u32 const h = zd_read_ncount_body(norm, maxSV, tableLog, [&](u32 p) { return (u32)(pad >> (8 * p)); }, 8);
    return h > srcSize ? 0 : h;
}

// ------------------------------------------------------------------ tANS table (one lane) ---
// N/decompress/zstd_decompress_block.c:485-603.  kind: 0 LL, 1 OF, 2 ML (extra-bits lookup).
ZJ_DEV u32 zd_extra_bits(u32 kind, u32 sym) { return kind == 1 ? sym : (kind == 0 ? zd_ll_bits(sym) : zd_ml_bits(sym)); }

// In two parts: the spread (low-probability symbols to the table's end, the others along the walk pos += step: cells[u] = the symbol of slot u), and the pass that turns
// a slot's symbol into its cell — the slot's rank among its symbol's slots decides the state it leads to (nextState = symbolNext[symbol]++ in ascending slot order).
ZJ_DEV bool zd_fse_spread(u32* cells, const short* norm, u16* symNext, u32 maxSV, u32 tableLog) {
    u32 const size = 1u << tableLog, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    u32 high = size - 1, pos = 0;
    for (u32 s = 0; s <= maxSV; s++) {
        if (norm[s] == -1) { cells[high--] = s; symNext[s] = 1; } else symNext[s] = (u16)norm[s];
    }
    for (u32 s = 0; s <= maxSV; s++) {
        for (i32 i = 0; i < norm[s]; i++) {
            cells[pos] = s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    return pos == 0;
}
ZJ_DEV void zd_fse_finish(u32* cells, u16* symNext, u32 tableLog, u32 kind) {
    u32 const size = 1u << tableLog;
    for (u32 u = 0; u < size; u++) {
        u32 const sym = cells[u];
        u32 const ns = symNext[sym]++;
        u32 const nb = tableLog - zj_hibit(ns);
        cells[u] = ZD_CELL((ns << nb) - size, nb, sym, zd_extra_bits(kind, sym));
    }
}
ZJ_DEV bool zd_build_fse(u32* cells, const short* norm, u16* symNext, u32 maxSV, u32 tableLog, u32 kind) {
    if (!zd_fse_spread(cells, norm, symNext, maxSV, tableLog)) return false;
    zd_fse_finish(cells, symNext, tableLog, kind);
    return true;
}
// The second part by the whole wave (round 5).  On one lane it is a chain of 512 dependent LDS read-modify-writes (symbolNext[symbol]++) — ~70 us of the ~100 a frame
// spends in zj_dec_prep_kernel; here every slot finds its rank by counting: a bitmap of its slots per symbol (bm: 54 symbols x 16 words, zeroed here), the rank of slot u =
// the set bits below u in its symbol's bitmap.  Same cells (tests/test_emu_decode.py::test_emu_fse_table_by_the_wave).  symNext keeps the spread's starting values.
#define ZD_FSE_BM_WORDS (54u * 16u)
static_assert(ZD_FSE_BM_WORDS * 4u <= (2u << ZD_HUF_CELLS_LOG), "the bitmaps fit the Huffman table's room (ZDecShared::huf), which stage 1 lends them");
#if ZJ_ON_GPU
ZJ_DEV void zd_or32(u32* p, u32 v) { atomicOr(p, v); }
#else
static inline void zd_or32(u32* p, u32 v) { *p |= v; }
#endif
template <class G>
ZJ_DEV void zd_fse_finish_wave(const G& g, u32* cells, const u16* symNext, u32 tableLog, u32 kind, u32* bm) {
    u32 const size = 1u << tableLog;
    GRP_FOR(g, i, ZD_FSE_BM_WORDS) bm[i] = 0;
    g.sync();
    GRP_FOR(g, u, size) { u32 const sym = cells[u] & 63u; zd_or32(&bm[sym * 16u + (u >> 5)], 1u << (u & 31u)); }
    g.sync();
    GRP_FOR(g, u, size) {
        u32 const sym = cells[u] & 63u; const u32* const b = bm + sym * 16u; u32 const w = u >> 5;
        u32 rank = (u32)__builtin_popcount(b[w] & ((1u << (u & 31u)) - 1u));
        for (u32 x = 0; x < w; x++) rank += (u32)__builtin_popcount(b[x]);
        u32 const ns = (u32)symNext[sym] + rank;
        u32 const nb = tableLog - zj_hibit(ns);
        cells[u] = ZD_CELL((ns << nb) - size, nb, sym, zd_extra_bits(kind, sym));
    }
    g.sync();
}

// ------------------------------------------------------------------ private lane-0 state -----
struct ZDecSeqPriv {
    u32 sLL, sOF, sML;     // tANS states
    i32 A;                 // unread bits end position (bit index rel. to block start)
    i32 S0;                // first bit of the sequence bitstream
    u32 rep0, rep1, rep2;
    u32 i;                 // sequences decoded so far
    u32 opos;              // output bytes produced so far in this frame (incl. previous blocks)
    u32 lpos;              // literals consumed in this block
};

ZJ_DEV u32 zd_bits(const u8* win, u32 winLo, i32 p, u32 n) {   // n <= 32 bits at bit index p
    u64 const w = ld64(win + ((u32)p >> 3) - winLo) >> ((u32)p & 7);
    return (u32)w & (u32)(((u64)1 << n) - 1);
}

// COLD PATH — entered only when the sequence bit stream runs dry before the last sequence (corrupted input): the refusal is
// certain, the question is its CODE.  The reference does not stop there: it keeps reading from the exhausted 64-bit container
// (by then the stream's first 8 bytes, or all of a shorter stream; N/common/bitstream.h:254-300, :370-420) with shifts that
// wrap at 64 — BIT_readBitsFast = (C << (consumed & 63)) >> (64 - nb) (:347-356), BIT_readBits = (C >> ((64 - consumed - nb)
// & 63)) & mask(nb) (:303-343), consumed = 64 - bits left — and a later sequence trips a check of ZSTD_execSequence
// (dstSize_tooSmall before corruption_detected, zstd_decompress_block.c:919-934), or the end-of-stream test does (:1581).
// Walks the remaining sequences (from state p; init: the three initial states are still to be read) with exactly those bits
// and returns that code.  Nothing is written.  The window already reaches the stream's first byte here.
ZJ_DEV u32 zd_seq_dry_code(ZDecShared& sh, ZDecSeqPriv p, bool init, u32 dstCap) {
    u32 const nbSeq = sh.nbSeq, litSize = sh.litSize, winLo = sh.winLo;
    u32 const startByte = (u32)p.S0 >> 3, len = sh.blkSize - startByte;
    u64 C = 0;
    for (u32 k = 0; k < 8u && k < len; k++) C |= (u64)sh.win[startByte - winLo + k] << (8u * k);
    i32 R = p.A - p.S0;                                        // valid bits left; goes negative
    auto fast = [&](u32 nb) -> u32 {                           // nb >= 1
        u32 v;
        if ((i32)nb <= R) v = zd_bits(sh.win, winLo, p.S0 + R - (i32)nb, nb);
        else { u32 const c = (u32)(64 - R); v = (u32)((C << (c & 63u)) >> (64u - nb)); }
        R -= (i32)nb; return v; };
    auto slow = [&](u32 nb) -> u32 {
        u32 v;
        if ((i32)nb <= R) v = nb ? zd_bits(sh.win, winLo, p.S0 + R - (i32)nb, nb) : 0u;
        else { u32 const c = (u32)(64 - R); v = (u32)(C >> ((64u - c - nb) & 63u)) & (u32)(((u64)1 << nb) - 1u); }
        R -= (i32)nb; return v; };
    if (init) { p.sLL = slow(sh.llLog); p.sOF = slow(sh.ofLog); p.sML = slow(sh.mlLog); }
    for (u32 i = p.i; i < nbSeq; i++) {
        u32 const cl = sh.ll[p.sLL], co = sh.of[p.sOF], cm = sh.ml[p.sML];
        u32 const ofx = ZD_CELL_EXTRA(co), mlx = ZD_CELL_EXTRA(cm), llx = ZD_CELL_EXTRA(cl);
        u32 const llBase = sh.llBase[ZD_CELL_SYM(cl)];
        u32 offset;
        if (ofx > 1) { offset = (1u << ofx) - 3u + fast(ofx); p.rep2 = p.rep1; p.rep1 = p.rep0; p.rep0 = offset; }
        else {
            u32 const ll0 = (llBase == 0);
            if (ofx == 0) { if (ll0) { offset = p.rep1; p.rep1 = p.rep0; p.rep0 = offset; } else offset = p.rep0; }
            else {
                u32 const idx = 1 + ll0 + fast(1);
                u32 t = (idx == 3) ? p.rep0 - 1 : (idx == 1 ? p.rep1 : p.rep2);
                t -= !t;
                if (idx != 1) p.rep2 = p.rep1;
                p.rep1 = p.rep0; p.rep0 = t; offset = t;
            }
        }
        u32 const mlen = sh.mlBase[ZD_CELL_SYM(cm)] + (mlx ? fast(mlx) : 0u);
        u32 const llen = llBase + (llx ? fast(llx) : 0u);
        if (i + 1 < nbSeq) {
            p.sLL = ZD_CELL_NEXT(cl) + slow(ZD_CELL_NB(cl));
            p.sML = ZD_CELL_NEXT(cm) + slow(ZD_CELL_NB(cm));
            p.sOF = ZD_CELL_NEXT(co) + slow(ZD_CELL_NB(co));
        }
        if ((u64)p.opos + llen + mlen > dstCap) return ZJ_E_DSTSIZE_TOO_SMALL;
        if (llen > litSize - p.lpos) return ZJ_E_CORRUPTION;
        if ((u64)offset > (u64)p.opos + llen + sh.dictSize) return ZJ_E_CORRUPTION;
        p.opos += llen + mlen; p.lpos += llen;
    }
    return ZJ_E_CORRUPTION;                                     // the stream did not end on its first bit
}

// Decode up to ZD_SEQ_BATCH sequences into sh.sLit/sMl/sOff.  Runs on lane 0 only.
// N/decompress/zstd_decompress_block.c:1229-1347.  The chain that bounds this kernel is
// state -> cell (LDS) -> bit counts -> next state; everything else is kept off it: the 64 bits below the
// read position are fetched (two 8-byte LDS reads whose address only depends on the previous sequence)
// together with the three cells, all fields are then cut from that register top-down, and validity is
// accumulated and tested once per batch.
ZJ_DEV void zd_seq_batch(ZDecShared& sh, ZDecSeqPriv& p, u32 dstCap) {
    u32 const nbSeq = sh.nbSeq, winLo = sh.winLo;
    u32 n = 0, litTotal = 0, outTotal = 0, err = 0, bad = 0;
    bool dry = false;
    u32 const litSize = sh.litSize;
    u32 const seqStartByte = (u32)p.S0 >> 3;
    sh.bLitStart = p.lpos; sh.bOutStart = p.opos;
    while (n < ZD_SEQ_BATCH && p.i < nbSeq) {
        // window must cover the 136 bits below A unless it already reaches the stream start
        if (winLo > seqStartByte && ((p.A - 136) >> 3) < (i32)winLo) break;
        u32 const e = ((u32)p.A + 7) >> 3, shf = 8 * e - (u32)p.A;
        bool const wide = e >= winLo + 16;                       // false only within 16 bytes of the stream start
        u32 const wbase = wide ? (e - 16) - winLo : 0;
        u64 const wlo = ld64(sh.win + wbase), whi = ld64(sh.win + wbase + 8);
        u32 const cl = sh.ll[p.sLL], co = sh.of[p.sOF], cm = sh.ml[p.sML];
        u64 v = shf ? ((whi << shf) | (wlo >> (64 - shf))) : whi;         // bits [A-64, A), MSB = bit A-1
        u32 const ofx = ZD_CELL_EXTRA(co), mlx = ZD_CELL_EXTRA(cm), llx = ZD_CELL_EXTRA(cl);
        bool const last = (p.i + 1 == nbSeq);
        u32 const nl = last ? 0 : ZD_CELL_NB(cl), nm = last ? 0 : ZD_CELL_NB(cm), no = last ? 0 : ZD_CELL_NB(co);
        u32 const T = ofx + mlx + llx + nl + nm + no;
        u32 ofv, mlv, llv, vl, vm, vo;
        if (p.A - (i32)T < p.S0) { dry = true; break; }           // p is still the state before this sequence
        if (T <= 56 && wide) {
#define ZD_TAKE(nb) ((u32)((v >> 1) >> (63 - (nb))))
            ofv = ZD_TAKE(ofx); v <<= ofx;
            mlv = ZD_TAKE(mlx); v <<= mlx;
            llv = ZD_TAKE(llx); v <<= llx;
            vl = ZD_TAKE(nl); v <<= nl;
            vm = ZD_TAKE(nm); v <<= nm;
            vo = ZD_TAKE(no);
#undef ZD_TAKE
            p.A -= (i32)T;
        } else {
            p.A -= (i32)T;
            i32 q = p.A;
            vo = zd_bits(sh.win, winLo, q, no); q += (i32)no;
            vm = zd_bits(sh.win, winLo, q, nm); q += (i32)nm;
            vl = zd_bits(sh.win, winLo, q, nl); q += (i32)nl;
            llv = zd_bits(sh.win, winLo, q, llx); q += (i32)llx;
            mlv = zd_bits(sh.win, winLo, q, mlx); q += (i32)mlx;
            ofv = zd_bits(sh.win, winLo, q, ofx);
        }
        if (!last) { p.sLL = ZD_CELL_NEXT(cl) + vl; p.sML = ZD_CELL_NEXT(cm) + vm; p.sOF = ZD_CELL_NEXT(co) + vo; }
        u32 const llen = sh.llBase[ZD_CELL_SYM(cl)] + llv;
        u32 const mlen = sh.mlBase[ZD_CELL_SYM(cm)] + mlv;
        u32 offset;
        if (ofx > 1) {
            offset = (1u << ofx) - 3u + ofv;
            p.rep2 = p.rep1; p.rep1 = p.rep0; p.rep0 = offset;
        } else {
            u32 const ll0 = (llen == 0);
            if (ofx == 0) {
                if (ll0) { offset = p.rep1; p.rep1 = p.rep0; p.rep0 = offset; } else offset = p.rep0;
            } else {
                u32 const idx = 1 + ll0 + ofv;
                u32 t = (idx == 3) ? p.rep0 - 1 : (idx == 1 ? p.rep1 : p.rep2);
                t -= !t;
                if (idx != 1) p.rep2 = p.rep1;
                p.rep1 = p.rep0; p.rep0 = t; offset = t;
            }
        }
        // validity is accumulated (64-bit so nothing wraps) and tested after the batch; nothing is executed on a bad batch
        bad |= (llen > litSize - p.lpos) ? 1u : 0u;
        bad |= ((u64)p.opos + llen + mlen > dstCap) ? 2u : 0u;
        bad |= ((u64)offset > (u64)p.opos + llen + sh.dictSize) ? 1u : 0u;     // may reach into the dictionary content
        if (bad) break;
        sh.sLit[n] = llen; sh.sMl[n] = mlen; sh.sOff[n] = offset;
        p.lpos += llen; p.opos += llen + mlen; litTotal += llen; outTotal += llen + mlen;
        n++; p.i++;
    }
    if (dry) err = zd_seq_dry_code(sh, p, false, dstCap);           // cold: the code the reference's garbage sequences end on
    if (!err && bad) err = (bad & 2u) ? ZJ_E_DSTSIZE_TOO_SMALL : ZJ_E_CORRUPTION;     // ZSTD_execSequence tests the destination first (zstd_decompress_block.c:919-920, :967-968)
    if (!err && p.i == nbSeq && p.A != p.S0) err = ZJ_E_CORRUPTION;
    sh.bN = n; sh.bLitTotal = litTotal; sh.bOutTotal = outTotal;
    sh.seqDone = (p.i == nbSeq);
    if (err) sh.err = err;
    // next window request: bytes ending at the byte holding bit A-1 (+8 slack for the 64-bit load)
    {   u32 const hi = ((u32)p.A >> 3) + 9;
        u32 lo = hi > ZD_SWIN ? hi - ZD_SWIN : 0;
        if (lo < seqStartByte) lo = seqStartByte;
        sh.winLo = lo; }
}

// The same batch assembled in LDS.  In text-like data most matches copy bytes that an earlier sequence of the
// same batch has just produced, so the batch resolves in ~10 dependency rounds; through HBM/L2 every round is a
// store -> load round trip of thousands of cycles, in LDS it is a hundred.  The batch's output window
// [op0, op0 + outTot) (<= ZD_STAGE_BYTES, else the global-memory path below is used) is built in `stage`: literals
// land there from the literal buffer, match bytes come from `stage` when their source lies in the window and
// from the already final output in HBM when it lies before it, and the finished window goes out as one
// coalesced copy.
#define ZD_STAGE_BYTES 4096u
#if ZJ_ON_GPU
template <bool DICT, class G>
ZJ_DEV void zd_execute_staged(const G& g, ZDecShared& sh, u8* out, const u8* lit, u32 litAvail, u8* stage, bool valid, u32 ll, u32 ml, u32 off,
                              u32 lp, u32 op, u32 op0, u32 outTot, const u8* dictEnd) {
    u32 const k = g.lane();
    u32 const so = op - op0;                      // window offset of this sequence's literals
    u32 const mp = op + ll, md = mp - op0;        // match destination: absolute / in the window
    // ---- literals: every lane fetches its (<= 32-byte) run in one go, long runs go through the whole wave ----
    if (ll && ll <= 32 && lp + 32u > litAvail) {  // within 32 bytes of the end of what may be read: byte by byte
        for (u32 j = 0; j < ll; j++) stage[so + j] = lit[lp + j];
    } else if (ll && ll <= 32) {
        const u8* const s = lit + lp;
        u64 const a = ld64(s), b = ll > 8 ? ld64(s + 8) : 0, c = ll > 16 ? ld64(s + 16) : 0, d = ll > 24 ? ld64(s + 24) : 0;
        u8* const t = stage + so;
        u32 const full = ll >> 3, rest = ll & 7u;
        if (full > 0) st64(t, a);
        if (full > 1) st64(t + 8, b);
        if (full > 2) st64(t + 16, c);
        if (full > 3) st64(t + 24, d);
        u64 const last = full == 0 ? a : (full == 1 ? b : (full == 2 ? c : d));
        for (u32 j = 0; j < rest; j++) t[8 * full + j] = (u8)(last >> (8 * j));
    }
    {   u64 m = __ballot(ll > 32);
        while (m) {
            u32 const q = (u32)__builtin_ctzll(m); m &= m - 1;
            u32 const qll = ZJ_UNI(__shfl(ll, q, 64)), qlp = ZJ_UNI(__shfl(lp, q, 64)), qso = ZJ_UNI(__shfl(so, q, 64));
            grp_copy_wide(g, stage + qso, lit + qlp, qll);
        }
    }
    // ---- matches: dependency rounds (same rule as the global-memory path) ----
    sh.sLit[k] = mp; sh.sMl[k] = mp + ml;
    g.sync();
    // source position: negative = in the dictionary content (dn bytes come from there), then the final output
    // before the window, then the window itself
    i32 const sp = (i32)mp - (i32)off;
    u32 const dn = (DICT && sp < 0) ? zj_min(ml, (u32)(-sp)) : 0u;
    u32 const ms = (DICT && sp < 0) ? 0u : (u32)sp;
    u32 const me = zj_min(ms + (ml - dn), mp);
    u64 dep = 0;
    if (valid && ml > dn) {
        u32 lo = 0, hi = k;
        while (lo < hi) { u32 const mid = (lo + hi) >> 1; if (sh.sMl[mid] > ms) hi = mid; else lo = mid + 1; }
        u32 const jlo = lo;
        lo = jlo; hi = k;
        while (lo < hi) { u32 const mid = (lo + hi) >> 1; if (sh.sLit[mid] >= me) hi = mid; else lo = mid + 1; }
        u32 const jhi = lo;
        if (jhi > jlo) dep = ((jhi >= 64 ? ~0ull : ((1ull << jhi) - 1)) & ~((1ull << jlo) - 1));
    }
    u64 pending = __ballot(valid && ml > 0);
    while (pending) {
        bool const mine = (pending >> k) & 1;
        bool const ready = mine && ((dep & pending) == 0);
        u64 const readyMask = __ballot(ready);
        u64 big = __ballot(ready && ml > 64);
        while (big) {                             // long matches: the whole wave, byte j of the match from source byte j mod offset
            u32 const q = (u32)__builtin_ctzll(big); big &= big - 1;
            u32 qml = ZJ_UNI(__shfl(ml, q, 64)), qmp = ZJ_UNI(__shfl(mp, q, 64)); u32 const qoff = ZJ_UNI(__shfl(off, q, 64));
            if (DICT && qoff > qmp) {             // the part inside the dictionary first; the rest is an ordinary match
                u32 const used = zj_min(qml, qoff - qmp);
                grp_copy_wide(g, stage + (qmp - op0), dictEnd - (qoff - qmp), used);
                g.sync();
                qmp += used; qml -= used;
            }
            u32 const qms = qmp - qoff, qmd = qmp - op0;
            GRP_FOR(g, j, qml) {
                u32 const spq = qms + (qoff >= qml ? j : j % qoff);
                stage[qmd + j] = spq < op0 ? out[spq] : stage[spq - op0];
            }
            g.sync();
        }
        if (ready && ml <= 64) {
            u8* const d = stage + md;
            u32 j = 0;
            if (DICT && dn) { const u8* const dm = dictEnd + sp; for (; j + 8 <= dn; j += 8) st64(d + j, ld64(dm + j)); for (; j < dn; j++) d[j] = dm[j]; }
            if (ms + (j - dn) < op0 && j < ml) {  // then the part from the final output before the window
                u32 const gc = j + zj_min(ml - j, op0 - (ms + (j - dn)));
                const u8* const m = out + ms - dn;                   // m[j] = output byte of source position ms + (j - dn)
                for (; j + 8 <= gc; j += 8) st64(d + j, ld64(m + j));
                for (; j < gc; j++) d[j] = m[j];
            }
            u32 const wo = ms + (j - dn) >= op0 ? ms + (j - dn) - op0 : 0u;
            const u8* const w = stage + wo - j;                      // w[j] = window byte of source position ms + (j - dn)
            if (off >= 8) { for (; j + 8 <= ml; j += 8) st64(d + j, ld64(w + j)); }
            for (; j < ml; j++) d[j] = w[j];
        }
        g.sync();
        pending &= ~readyMask;
    }
    // ---- the finished window goes out in one coalesced copy ----
    grp_copy_wide(g, out + op0, stage, outTot);
    zj_mem_order();
    g.sync();
}
#endif

// Execute one batch of n <= 64 decoded sequences (N/decompress/zstd_decompress_block.c:1001-1096).
// GPU: one sequence per lane.  Output positions come from a wave prefix sum; all literal runs are copied
// first (they only read the literal buffer); a match may read bytes that an earlier match of the same batch
// produces, so matches run in rounds — a lane copies once no still-pending earlier lane overlaps its source
// range (the lowest pending lane is always ready).  Long runs/matches are copied by the whole wave.
// CHECK (multi-block frames, zj_decode_split.h): the sequences come from a stage that could not test an offset against the frame's output position — a match
// that starts before the output (or an offset of zero) sets sh.err and nothing is written.
template <bool DICT = false, class G, bool CHECK = false>
ZJ_DEV void zd_execute_batch(const G& g, ZDecShared& sh, u8* out, const u8* lit, u32 n, u32 lp0, u32 op0, u32& litTot, u32& outTot, u8* stage = nullptr, u32 litAvail = 0, const u8* dictEnd = nullptr) {
#if ZJ_ON_GPU
    u32 const k = g.lane();
    bool const valid = k < n;
    u32 const ll = valid ? sh.sLit[k] : 0, ml = valid ? sh.sMl[k] : 0, off = valid ? sh.sOff[k] : 1;
    u32 sl = ll, so = ll + ml;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 const a = __shfl_up(sl, d, 64), b = __shfl_up(so, d, 64);
        if ((int)k >= d) { sl += a; so += b; }
    }
    litTot = ZJ_UNI(__shfl(sl, 63, 64)); outTot = ZJ_UNI(__shfl(so, 63, 64));
    u32 const lp = lp0 + sl - ll;                 // literal source
    u32 const op = op0 + so - ll - ml;            // output position of this sequence's literals
    u32 const mp = op + ll;                       // match destination
    if (CHECK) {
        if (__ballot(valid && (off == 0u || off > mp)) != 0ull) { if (k == 0) sh.err = ZJ_E_CORRUPTION; g.sync(); return; }
    }
    if (stage && outTot <= ZD_STAGE_BYTES) { zd_execute_staged<DICT>(g, sh, out, lit, litAvail, stage, valid, ll, ml, off, lp, op, op0, outTot, dictEnd); return; }
    // ---- literals: short runs per lane, long runs by the whole wave ----
    if (ll <= 32) {
        u32 j = 0;
        for (; j + 8 <= ll; j += 8) st64(out + op + j, ld64(lit + lp + j));
        for (; j < ll; j++) out[op + j] = lit[lp + j];
    }
    {   u64 m = __ballot(ll > 32);
        while (m) {
            u32 const q = (u32)__builtin_ctzll(m); m &= m - 1;
            u32 const qll = ZJ_UNI(__shfl(ll, q, 64)), qlp = ZJ_UNI(__shfl(lp, q, 64)), qop = ZJ_UNI(__shfl(op, q, 64));
            grp_copy_wide(g, out + qop, lit + qlp, qll);
        }
    }
    zj_mem_order();
    // ---- matches: dependency rounds ----
    sh.sLit[k] = mp; sh.sMl[k] = mp + ml;          // reuse as mStart / mEnd arrays (LDS)
    g.sync();
    // a source position before the frame's output lies in the dictionary content (ZSTD_execSequence's extDict branch,
    // N/decompress/zstd_decompress_block.c:1051-1068): dn bytes come from there, the rest from the output
    i32 const sp = (i32)mp - (i32)off;
    u32 const dn = (DICT && sp < 0) ? zj_min(ml, (u32)(-sp)) : 0u;      // DICT = false: the sequence checks guarantee sp >= 0
    u32 const ms = (DICT && sp < 0) ? 0u : (u32)sp;         // first source byte inside the output
    u32 const me = zj_min(ms + (ml - dn), mp);    // source bytes at/after mp are produced by this match itself
    // earlier lanes whose match output [mStart_j, mEnd_j) intersects [ms, me): a contiguous lane range
    u64 dep = 0;
    if (valid && ml > dn) {
        u32 lo = 0, hi = k;                       // first j in [0,k) with mEnd_j > ms
        while (lo < hi) { u32 const mid = (lo + hi) >> 1; if (sh.sMl[mid] > ms) hi = mid; else lo = mid + 1; }
        u32 const jlo = lo;
        lo = jlo; hi = k;                         // first j in [jlo,k) with mStart_j >= me
        while (lo < hi) { u32 const mid = (lo + hi) >> 1; if (sh.sLit[mid] >= me) hi = mid; else lo = mid + 1; }
        u32 const jhi = lo;                       // deps = [jlo, jhi)
        if (jhi > jlo) dep = ((jhi >= 64 ? ~0ull : ((1ull << jhi) - 1)) & ~((1ull << jlo) - 1));
    }
    u64 pending = __ballot(valid && ml > 0);
    while (pending) {
        bool const mine = (pending >> k) & 1;
        bool const ready = mine && ((dep & pending) == 0);
        u64 const readyMask = __ballot(ready);
        // long matches: whole wave, one at a time (offset >= length or periodic pattern)
        u64 big = __ballot(ready && ml > 64);
        while (big) {
            u32 const q = (u32)__builtin_ctzll(big); big &= big - 1;
            u32 qml = ZJ_UNI(__shfl(ml, q, 64)), qmp = ZJ_UNI(__shfl(mp, q, 64)); u32 const qoff = ZJ_UNI(__shfl(off, q, 64));
            if (DICT && qoff > qmp) {             // starts in the dictionary content: that part first, the rest is an ordinary match
                u32 const used = zj_min(qml, qoff - qmp);
                grp_copy_wide(g, out + qmp, dictEnd - (qoff - qmp), used);
                zj_mem_order();
                qmp += used; qml -= used;
            }
            const u8* const m = out + qmp - qoff;
            if (qml == 0) { }
            else if (qoff >= qml) grp_copy_wide(g, out + qmp, m, qml);
            else if (qoff >= 64) { for (u32 base = 0; base < qml; base += 64) { u32 const j = base + k; if (j < qml) out[qmp + j] = m[j]; zj_mem_order(); } }
            else { GRP_FOR(g, j, qml) out[qmp + j] = m[j % qoff]; }
            zj_mem_order();
        }
        if (ready && ml <= 64) {
            u8* const d = out + mp; const u8* const m = d - off;
            u32 j = 0;
            if (DICT && dn) { const u8* const dm = dictEnd + sp; for (; j + 8 <= dn; j += 8) st64(d + j, ld64(dm + j)); for (; j < dn; j++) d[j] = dm[j]; }
            if (off >= 8) { for (; j + 8 <= ml; j += 8) st64(d + j, ld64(m + j)); }
            for (; j < ml; j++) d[j] = m[j];
        }
        zj_mem_order();
        pending &= ~readyMask;
    }
    g.sync();
#else
    u32 lp = lp0, op = op0;
    litTot = 0; outTot = 0;
    if (CHECK) {
        u32 p = op0;
        for (u32 k = 0; k < n; k++) { p += sh.sLit[k]; if (sh.sOff[k] == 0u || sh.sOff[k] > p) { sh.err = ZJ_E_CORRUPTION; return; } p += sh.sMl[k]; }
    }
    for (u32 k = 0; k < n; k++) {
        u32 const ll = sh.sLit[k], ml = sh.sMl[k], off = sh.sOff[k];
        litTot += ll; outTot += ll + ml;
        for (u32 j = 0; j < ll; j++) out[op + j] = lit[lp + j];
        lp += ll; op += ll;
        for (u32 j = 0; j < ml; j++) { i32 const v = (i32)op - (i32)off + (i32)j; out[op + j] = (DICT && v < 0) ? dictEnd[v] : out[v]; }
        op += ml;
    }
    (void)g;
#endif
}

// ------------------------------------------------------------------ Huffman -----------------
// Weights (lane 0): N/common/entropy_common.c:243-305 + N/common/fse_decompress.c:166-236.
// Returns header bytes, 0 on error.  Publishes sh.weights[0..nbSym), sh.hufLog, and nbSym via out.
ZJ_DEV u32 zd_huf_read_weights(ZDecShared& sh, const u8* src, u32 srcSize, u32* nbSymOut) {
    u32 iSize, oSize, total = 0;
    u32* const rank = sh.hrank;            // [ZD_HUF_LOG_MAX + 2] in LDS (dynamic indexing)
    u32* const start = sh.hrank + 16;
    if (!srcSize) return 0;
    iSize = src[0];
    if (iSize >= 128) {
        oSize = iSize - 127; iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize) return 0;
        for (u32 n = 0; n < oSize; n += 2) { u32 const b = src[1 + n / 2]; sh.weights[n] = (u8)(b >> 4); if (n + 1 < 256) sh.weights[n + 1] = (u8)(b & 15); }
    } else {
        // FSE-compressed weights, 2 interleaved states, tableLog <= 6.  The description may name more symbols than the 13 weight
        // values: the reference reads it with a limit of 255 and then refuses it only if table + build workspace for (tableLog,
        // maxSymbol) exceed the workspace HUF_readStats owns, FSE_DECOMPRESS_WKSP_SIZE_U32(6, 11) = 219 words
        // (N/common/fse_decompress.c:273, fse.h:267-273) — i.e. maxSymbol <= 11 at tableLog 6 and <= 91 at tableLog 5; weights
        // above 12 are refused after decoding.  Reading with a limit of 92 symbols gives the same verdicts.  The 64-cell table
        // lives in the (idle) Huffman staging area, norm[] and symNext[] in the (idle) bit-stream windows.
        u32 maxSV = 92, tl;
        u32* const wtab = (u32*)&sh.hstage[0][0];              // 64 cells = 256 B
        short* const norm = (short*)&sh.win[0];                // 93 shorts
        u16* const symNext = (u16*)&sh.win[256];               // 93 u16
        if (iSize + 1 > srcSize) return 0;
        {   u32 const h = zd_read_ncount(norm, &maxSV, &tl, src + 1, iSize);
            if (!h || tl > 6 || h > iSize) return 0;
            if ((1u + (1u << tl)) + 1u + ((2u * (maxSV + 1u) + (1u << tl) + 8u + 3u) >> 2) + 129u > 219u) return 0;
            if (!zd_build_fse(wtab, norm, symNext, maxSV, tl, 1)) return 0;
            {   const u8* bs = src + 1 + h; u32 const bn = iSize - h;
                i32 A; u32 s1, s2; u32 n = 0;
                if (bn == 0 || bs[bn - 1] == 0) return 0;
                A = (i32)((bn - 1) * 8 + zj_hibit(bs[bn - 1]));
                // small stream (<= 127 bytes): bits are read straight from global memory / L1
                auto rdbits = [&](u32 nb) -> u32 {
                    A -= (i32)nb;
                    if (A < 0 || nb == 0) return 0;
                    u32 const by = (u32)A >> 3; u32 w = 0;
                    for (u32 k = 0; k < 3; k++) { if (by + k < bn) w |= (u32)bs[by + k] << (8 * k); }
                    return (w >> ((u32)A & 7)) & ((1u << nb) - 1);
                };
                s1 = rdbits(tl); s2 = rdbits(tl);
                if (A < 0) return 0;
                for (;;) {
                    u32 c;
                    if (n + 2 > 255) return 0;
                    c = wtab[s1]; sh.weights[n++] = (u8)ZD_CELL_SYM(c); s1 = ZD_CELL_NEXT(c) + rdbits(ZD_CELL_NB(c));
                    if (A < 0) { sh.weights[n++] = (u8)ZD_CELL_SYM(wtab[s2]); break; }
                    if (n + 2 > 255) return 0;
                    c = wtab[s2]; sh.weights[n++] = (u8)ZD_CELL_SYM(c); s2 = ZD_CELL_NEXT(c) + rdbits(ZD_CELL_NB(c));
                    if (A < 0) { sh.weights[n++] = (u8)ZD_CELL_SYM(wtab[s1]); break; }
                }
                oSize = n;
            }
        }
    }
    for (u32 w = 0; w < ZD_HUF_LOG_MAX + 2; w++) rank[w] = 0;
    for (u32 n = 0; n < oSize; n++) {
        u32 const w = sh.weights[n];
        if (w > ZD_HUF_LOG_MAX) return 0;
        rank[w]++; total += (1u << w) >> 1;
    }
    if (total == 0) return 0;
    {   u32 const tl = zj_hibit(total) + 1;
        if (tl > ZD_HUF_LOG_MAX) return 0;
        u32 const rest = (1u << tl) - total, last = zj_hibit(rest) + 1;
        if ((1u << zj_hibit(rest)) != rest) return 0;
        sh.weights[oSize] = (u8)last; rank[last]++;
        if (rank[1] < 2 || (rank[1] & 1)) return 0;
        // first cell per symbol: weights ascending, symbols ascending inside a weight
        u32 cur = 0;
        for (u32 w = 1; w <= tl; w++) { start[w] = cur; cur += rank[w] << (w - 1); }
        for (u32 n = 0; n <= oSize; n++) {
            u32 const w = sh.weights[n];
            if (w) { sh.symPos[n] = (u16)start[w]; start[w] += 1u << (w - 1); }
        }
        sh.hufLog = tl; sh.hufW1 = tl > ZD_HUF_CELLS_LOG ? rank[1] : 0u;
        {   u32 gcd = 0;                                       // of the code lengths in use (zd_huf_streams_wave: codes start on that lattice)
            for (u32 w = 1; w <= tl; w++) { if (rank[w]) { u32 a = tl + 1u - w, b = gcd; while (b) { u32 const r = a % b; a = b; b = r; } gcd = a; } }
            sh.hufGcd = gcd ? gcd : 1u;
        }
    }
    *nbSymOut = oSize + 1;
    return iSize + 1;
}

// Huffman decode table from the weights zd_huf_read_weights left in sh (all lanes): cell = nbBits << 8 | symbol.
// A table 12 bits deep (HUF_TABLELOG_MAX; 4 096 slots in the reference, N/decompress/huf_decompress.c:385-518) is kept in the same 2 048 cells: slots are laid out
// by ascending weight, the weight-1 symbols (one slot each, an even number of them: N/common/entropy_common.c:300) first, so from slot hufW1 on a symbol's 2^(w-1)
// slots are whole pairs and cell j serves slots 2j and 2j + 1; below it cell j holds the two weight-1 symbols of its slots, one per byte (12 bits each).
template <class G>
ZJ_DEV void zd_huf_fill(const G& g, ZDecShared& sh, u32 nbSym) {
    u32 const log = ZJ_UNI(sh.hufLog);
    bool const deep = log > ZD_HUF_CELLS_LOG;
    GRP_FOR(g, s, nbSym) {
        u32 const w = sh.weights[s];
        if (w) {
            u32 const p0 = sh.symPos[s];
            if (deep && w == 1u) ((u8*)sh.huf)[p0] = (u8)s;
            else {
                u32 const len = deep ? 1u << (w - 2) : 1u << (w - 1), c0 = deep ? p0 >> 1 : p0;
                u16 const cell = (u16)(((log + 1 - w) << 8) | s);
                for (u32 k = 0; k < len; k++) sh.huf[c0 + k] = cell;
            }
        }
    }
}
// the cell of slot `idx` (the next hufLog bits of the stream) of a table deeper than ZD_HUF_CELLS_LOG
ZJ_DEV u32 zd_huf_cell_deep(const ZDecShared& sh, u32 idx) {
    u32 const c = sh.huf[idx >> 1];
    return idx < sh.hufW1 ? (ZD_HUF_LOG_MAX << 8) | ((c >> (8u * (idx & 1u))) & 0xFFu) : c;
}
ZJ_DEV u32 zd_huf_cell(const ZDecShared& sh, u32 idx, u32 log) { return log > ZD_HUF_CELLS_LOG ? zd_huf_cell_deep(sh, idx) : sh.huf[idx]; }

// ------------------------------------------------------------------ dictionaries -------------
// A digested dictionary in HBM = what ZSTD_createDDict keeps (N/decompress/zstd_ddict.c:36-130,
// ZSTD_loadDEntropy N/decompress/zstd_decompress.c:1448-1533): entropy tables in the decoder's own cell
// formats, the three start repcodes, and where the content starts in the raw dictionary bytes.
struct ZDDictDev {
    u32 status;                     // 0 ok, else ZJ_E_* (dictionary_corrupted)
    u32 dictID, contentOff, contentSize, hasEntropy;
    u32 rep[3];
    u32 hufLog, llLog, ofLog, mlLog;
    u32 hufW1;                      // ZDecShared::hufW1 of the table below
    u16 huf[1u << ZD_HUF_CELLS_LOG];
    u32 ll[512], of[256], ml[512];
    u16 c16[1280];                  // the same three tables as the split pipeline's 2-byte cells (LL | OF at 512 | ML at 768; zd_cell16): a frame whose
                                    // three tables are all "repeat" decodes straight from here
};
// Runs on one workgroup; `sh` is scratch.  Raw-content dictionaries (no magic) have no entropy section.
template <class G>
ZJ_DEV void zd_ddict_digest(const G& g, ZDecShared& sh, const u8* dict, u32 dictSize, ZDDictDev* out) {
    GRP_SERIAL(g) {
        sh.err = 0; sh.bN = 0; sh.blkType = 0;            // blkType: has entropy
        sh.hdrSize = 0;                                    // content offset
        sh.rep[0] = 1; sh.rep[1] = 4; sh.rep[2] = 8;
        sh.windowSize = 0;                                 // dictID
        if (dictSize >= 8 && ld32(dict) == 0xEC30A437u) {
            u32 pos = 8, err = 0, nbSym = 0;
            sh.windowSize = ld32(dict + 4);
            {   u32 const h = zd_huf_read_weights(sh, dict + pos, dictSize - pos, &nbSym);
                if (!h || h > dictSize - pos) err = ZJ_E_DICT_CORRUPTED; else pos += h; }
            sh.bN = nbSym;
            for (u32 t = 0; t < 3 && !err; t++) {          // order in the dictionary: OF, ML, LL
                u32 const kind = t == 0 ? 1u : (t == 1 ? 2u : 0u);
                u32 max = kind == 0 ? 35u : (kind == 1 ? 31u : 52u), tl = 0;
                u32 const h = zd_read_ncount(sh.norm[0], &max, &tl, dict + pos, dictSize - pos);
                if (!h || h > dictSize - pos || tl > (kind == 1 ? 8u : 9u)) { err = ZJ_E_DICT_CORRUPTED; break; }
                u32* const cells = kind == 0 ? sh.ll : (kind == 1 ? sh.of : sh.ml);
                if (!zd_build_fse(cells, sh.norm[0], sh.symNext, max, tl, kind)) { err = ZJ_E_DICT_CORRUPTED; break; }
                if (kind == 0) sh.llLog = tl; else if (kind == 1) sh.ofLog = tl; else sh.mlLog = tl;
                pos += h;
            }
            if (!err && pos + 12 > dictSize) err = ZJ_E_DICT_CORRUPTED;
            if (!err) {
                u32 const content = dictSize - (pos + 12);
                for (u32 i = 0; i < 3; i++) { u32 const r = ld32(dict + pos + 4 * i); if (r == 0 || r > content) err = ZJ_E_DICT_CORRUPTED; sh.rep[i] = r; }
                sh.hdrSize = pos + 12; sh.blkType = 1;
            }
            if (err) sh.err = err;
        }
    }
    g.sync();
    if (!ZJ_UNI(sh.err) && ZJ_UNI(sh.blkType)) { zd_huf_fill(g, sh, ZJ_UNI(sh.bN)); g.sync(); }
    GRP_FOR(g, i, 1u << ZD_HUF_CELLS_LOG) out->huf[i] = sh.huf[i];
    GRP_FOR(g, i, 512) { out->ll[i] = sh.ll[i]; out->ml[i] = sh.ml[i]; }
    GRP_FOR(g, i, 256) out->of[i] = sh.of[i];
    if (ZJ_UNI(sh.blkType) && !ZJ_UNI(sh.err)) {
        GRP_FOR(g, i, 512) { out->c16[i] = (u16)(i < (1u << sh.llLog) ? zd_cell16(sh.ll[i], sh.llLog) : 0u); out->c16[768u + i] = (u16)(i < (1u << sh.mlLog) ? zd_cell16(sh.ml[i], sh.mlLog) : 0u); }
        GRP_FOR(g, i, 256) out->c16[512u + i] = (u16)(i < (1u << sh.ofLog) ? zd_cell16(sh.of[i], sh.ofLog) : 0u);
    }
    GRP_SERIAL(g) {
        out->status = sh.err; out->dictID = sh.windowSize; out->contentOff = sh.hdrSize; out->contentSize = dictSize - sh.hdrSize;
        out->hasEntropy = sh.blkType; out->rep[0] = sh.rep[0]; out->rep[1] = sh.rep[1]; out->rep[2] = sh.rep[2];
        out->hufLog = sh.hufLog; out->hufW1 = sh.hufW1; out->llLog = sh.llLog; out->ofLog = sh.ofLog; out->mlLog = sh.mlLog;
    }
    zj_mem_order();
    g.sync();
}

// The dictionary's entropy tables into LDS (a frame that uses a dictionary starts from them): 4 KiB Huffman table
// and/or 5 KiB of tANS cells, copied as words, not unrolled — a handful of loads in flight is enough and keeps the
// kernels' register budget where it is without dictionaries.
template <class G>
ZJ_DEV void zd_load_dict_entropy(const G& g, ZDecShared& sh, const ZDDictDev* dd, bool huf, bool fse) {
    if (huf) {
        const u32* const s32 = (const u32*)dd->huf; u32* const h32 = (u32*)sh.huf;
#if ZJ_ON_GPU
#pragma clang loop unroll(disable)
#endif
        for (u32 i = g.lane(); i < (1u << ZD_HUF_CELLS_LOG) / 2u; i += (u32)g.W) h32[i] = s32[i];
    }
    if (fse) {
#if ZJ_ON_GPU
#pragma clang loop unroll(disable)
#endif
        for (u32 i = g.lane(); i < 512u; i += (u32)g.W) { sh.ll[i] = dd->ll[i]; sh.ml[i] = dd->ml[i]; }
#if ZJ_ON_GPU
#pragma clang loop unroll(disable)
#endif
        for (u32 i = g.lane(); i < 256u; i += (u32)g.W) sh.of[i] = dd->of[i];
    }
}

// Decode <= ZD_HSYM symbols of stream t from its LDS window (one lane per stream).
// N/decompress/huf_decompress.c:721-835 (4X1 loop), :600-640 (1X1).
ZJ_DEV void zd_huf_stream_round(ZDecShared& sh, u32 t) {
    u32 const log = sh.hufLog;
    u32 const todo = zj_min(ZD_HSYM, sh.hN[t] - sh.hDone[t]);
    i32 A = (i32)sh.hA[t]; i32 const S0 = (i32)sh.hS0[t];
    u32 const lo = sh.hLo[t];
    const u8* const win = sh.win + t * ZD_HWIN_STRIDE;
    u32 const startByte = (u32)S0 >> 3;
    u32 k = 0;
    if (log > ZD_HUF_CELLS_LOG) {                              // a table 12 bits deep (no libzstd encoder writes one): 4 symbols per refill, cells through zd_huf_cell_deep
        while (k < todo) {
            u64 c;
            u32 const byteEnd = ((u32)A + 7) >> 3;
            if (A <= S0) c = 0;
            else if (byteEnd >= startByte + 8) c = ld64(win + (byteEnd - 8 - lo)) << (8 * byteEnd - (u32)A);
            else c = ld64(win + (startByte - lo)) << (64 - (u32)(A - S0));
            u32 const m = zj_min(4u, todo - k);
            for (u32 j = 0; j < m; j++) {
                u32 const cell = zd_huf_cell_deep(sh, (u32)(c >> (64 - ZD_HUF_LOG_MAX)));
                u32 const nb = cell >> 8;
                sh.hstage[t][k + j] = (u8)cell;
                c <<= nb; A -= (i32)nb;
            }
            k += m;
        }
        sh.hA[t] = (u32)A; sh.hCnt[t] = todo;
        return;
    }
    while (k < todo) {
        // refill a 64-bit container whose MSB is bit A-1 (>= 57 valid bits, zeros below S0)
        u64 c;
        u32 const byteEnd = ((u32)A + 7) >> 3;
        if (A <= S0) c = 0;
        else if (byteEnd >= startByte + 8) c = ld64(win + (byteEnd - 8 - lo)) << (8 * byteEnd - (u32)A);
        else c = ld64(win + (startByte - lo)) << (64 - (u32)(A - S0));
        u32 const m = zj_min(5u, todo - k);
        for (u32 j = 0; j < m; j++) {
            u32 const cell = sh.huf[(u32)(c >> (64 - log))];
            u32 const nb = cell >> 8;
            sh.hstage[t][k + j] = (u8)cell;
            c <<= nb; A -= (i32)nb;
        }
        k += m;
    }
    sh.hA[t] = (u32)A; sh.hCnt[t] = todo;
}

// Which table shape the reference picks for a 4-stream literals section (HUF_selectDecoder, N/decompress/huf_decompress.c:
// 1793-1843: {table build, per 256 symbols} costs of the one-code / two-code tables by compression-ratio bucket).  Valid streams
// decode to the same bytes with either; zd_huf_x2_accepts below is where the choice shows.
ZD_CONST u16 zd_k_huf_algo_time[16][4] = {
    {0,0,1,1}, {0,0,1,1}, {150,216,381,119}, {170,205,514,112}, {177,199,539,110}, {197,194,644,107}, {221,192,735,107},
    {256,189,881,106}, {359,188,1167,109}, {582,187,1570,114}, {688,187,1712,122}, {825,186,1965,136}, {976,185,2131,150},
    {1180,186,2070,175}, {1377,185,1731,202}, {1412,185,1695,202} };
ZJ_DEV u32 zd_huf_select_x2(u32 dstSize, u32 cSrcSize) {
    u32 const q = cSrcSize >= dstSize ? 15u : cSrcSize * 16u / dstSize, d256 = dstSize >> 8;
    u32 const t0 = zd_k_huf_algo_time[q][0] + zd_k_huf_algo_time[q][1] * d256;
    u32 t1 = zd_k_huf_algo_time[q][2] + zd_k_huf_algo_time[q][3] * d256;
    t1 += t1 >> 5;
    return t1 < t0 ? 1u : 0u;
}

// COLD PATH — runs only for a stream that did not end exactly on its first bit, i.e. only on corrupted input.  The reference's
// two-code table walks the stream one CELL at a time (HUF_decodeStreamX2, huf_decompress.c:1308-1349): the 11-bit cell under the
// cursor holds two codes when both fit (k1 + k2 <= 11, HUF_fillDTableX2 :1117-1177).  When one byte is left to produce,
// HUF_decodeLastSymbolX2 (:1275-1290) emits the cell's first code and, for a two-code cell, skips k1 + k2 bits clamped to the
// stream's end — so up to k2 left-over bits pass its end-of-stream check — and with no bit left at all it indexes the table
// with the top of the last-loaded container (the stream's first 8 bytes) and skips nothing.  Re-walks stream t by cells with
// those rules (one lane, straight from the block's bytes); true = the reference accepts it.  The bytes already decoded stand
// except possibly the last one, rewritten here.
ZJ_DEV bool zd_huf_x2_accepts(ZDecShared& sh, const u8* bsrc, u32 t, u8* out) {
    u32 const log = sh.hufLog, D = log > 11u ? log : 11u, n = sh.hN[t];
    i32 A = (i32)sh.hA0[t]; i32 const S0 = (i32)sh.hS0[t];
    u32 const startByte = (u32)S0 >> 3, len = (sh.hA0[t] >> 3) + 1u - startByte;
    for (u32 i = 0; i < n; ) {
        bool const last = (i + 1u == n);
        i32 const R = A - S0;
        if (R < 0 || (!last && R == 0)) return false;
        u32 idx = 0;
        if (R > 0) {                                           // D bits below A, zeros below the stream's first bit
            for (u32 k = 0; k < D; k++) { i32 const bit = A - 1 - (i32)k; idx <<= 1; if (bit >= S0) idx |= (bsrc[(u32)bit >> 3] >> ((u32)bit & 7u)) & 1u; }
        } else {
            u64 c = 0; for (u32 k = 0; k < 8u && k < len; k++) c |= (u64)bsrc[startByte + k] << (8u * k);
            idx = (u32)(c >> (64u - D));
        }
        u32 const c1 = zd_huf_cell(sh, idx >> (D - log), log), k1 = c1 >> 8;
        u32 const c2 = zd_huf_cell(sh, ((idx << k1) & ((1u << D) - 1u)) >> (D - log), log), k2 = c2 >> 8;
        bool const two = k1 + k2 <= D;
        if (last) {
            out[i] = (u8)c1;
            if (!two) return (u32)R == k1;
            return !(R > 0 && (i32)(k1 + k2) < R);
        }
        A -= (i32)(two ? k1 + k2 : k1); i += two ? 2u : 1u;
    }
    return A == S0;
}

// ------------------------------------------------------------------ the four streams by the whole wave (round 6) ----
// A Huffman stream is a chain — a code's length says where the next code starts — and the rounds of zd_block_literals run it on ONE lane per stream: 4 of a wave's 64
// lanes walk ~7 000 (text) to 16 000 (few distinct bytes) dependent LDS look-ups each for a 64 KiB frame while 60 idle (the literal pass of the batch decode: 9 ms of a
// 20 ms call).  But a prefix code RE-SYNCHRONISES: a decoder started at an arbitrary bit falls into step with the true parse within a few codes.  So a stream's bits
// are cut into ZD_HP_LANES spans with a lane each (4 x 16 = the wave):
//   pass G  every lane but a stream's first starts ZD_HP_SYNC bits ABOVE its span and decodes down to the span's edge; where that lands is its guess of the first code
//           of the true parse inside the span (the stream's first lane starts at the stream's own first code).  Codes start a multiple of the code lengths' common
//           divisor below the stream's first code, so the guess starts on that lattice: a table of codes of ONE length — what few equally likely byte values give,
//           and where no decoder ever falls into step from the wrong phase — is guessed exactly;
//   pass V  every lane decodes its span from its start — counting, no stores — and notes where it crossed into the next span.  A lane whose start is not where its
//           predecessor really ended takes that end as its start and decodes again; the first lane's start is exact, so by induction every lane's is once nothing
//           changes any more (ZD_HP_TRIES repeats at most, then the section is left to the rounds);
//   check   the last lane ends exactly on the stream's first bit and the counts add up to the stream's regenerated size: what the reference's decoders ask of a valid
//           stream (N/decompress/huf_decompress.c:697, :830).  Anything else is NOT judged here — the rounds decode the section again and answer as the reference does
//           (which decoder it would have picked matters there: zd_huf_x2_accepts);
//   pass W  the counts' prefix sums are the lanes' places in the output; every lane decodes its span once more and stores, 8 symbols to a store.
// A lane's chain is (ZD_HP_SYNC / code length) + 2 x (stream / 16) look-ups instead of the stream's length.  Each pass runs in rounds like the one-lane decode's: the
// wave stages the next ZD_HP_WIN bytes under every lane's cursor in LDS (coalesced 16-byte loads; 64 lanes refilling straight from memory, a cache line each, made the
// sequence decode beside this pass wait: measured, profiles/r06), then every lane decodes out of its window.  Tables to ZD_HUF_CELLS_LOG bits; deeper ones (no libzstd
// encoder writes them) stay with the rounds, and so does every caller that has no LDS to spare for the windows (hpWin = nullptr).
#define ZD_HP_LANES 16u
#define ZD_HP_SYNC 192u
#define ZD_HP_TRIES 4u
#define ZD_HP_MIN_LIT 2048u          // smaller sections: the passes' fixed costs outweigh the chain
#define ZD_HP_WIN 64u                // bytes staged per lane and round
#define ZD_HP_STRIDE (ZD_HP_WIN + 8u)
#define ZD_HP_LDS (64u * ZD_HP_STRIDE)
static_assert(ZD_EXEC_LDS == ZD_HP_LDS_OFF + ZD_HP_LDS, "ZD_EXEC_LDS");
static_assert(ZD_HP_LDS <= (512u + 512u + 256u) * 4u, "the multi-block execution stage lends its staging area");
#if !ZJ_ON_GPU
static unsigned long long zd_hp_stats[5];        // lane-serial build (tests/emu): sections taken, left at the check, left after ZD_HP_TRIES, repeated passes V, lanes that decoded again
#define ZD_HP_STAT(i, n) (zd_hp_stats[i] += (n))
#else
#define ZD_HP_STAT(i, n) ((void)0)
#endif
// the edge of span j of a stream: span j is (edge(j + 1), edge(j)]
ZJ_DEV i32 zd_hp_edge(i32 A0, i32 S0, u32 j) { return A0 - (i32)((u64)(u32)(A0 - S0) * j / ZD_HP_LANES); }
// One pass.  Lane l decodes from bit position pA[l] (exclusive top) while the position is above its limit and it has decoded fewer than its cap; pA / pK run along.
//   mode 0 (G): lanes 1.. of a stream, limit = the span's upper edge, cap ZD_HP_SYNC      mode 1 (V): lanes flagged in pC, limit = the span's lower edge, cap = the stream's size + 1
//   mode 2 (W): every lane, no limit, cap = pC[l] codes, symbols to out + pO[l]
#ifndef ZD_HP_PASS_ATTR
#define ZD_HP_PASS_ATTR ZJ_DEV
#endif
template <class G>
ZD_HP_PASS_ATTR void zd_hp_pass(const G& g, ZDecShared& sh, const u8* bsrc, u32 bsize, u8* hpWin, u8* out, u32 mode) {
    u32 const log = ZJ_UNI(sh.hufLog);
    u32* const pC = sh.sOff; u32* const pA = (u32*)&sh.hstage[0][0]; u32* const pK = pA + 64; u32* const pO = (u32*)&sh.win[0];
    u32* const pLim = pO + 64; u32* const pCap = pO + 128;
    GRP_FOR(g, l, 64u) {
        u32 const t = l / ZD_HP_LANES, j = l % ZD_HP_LANES;
        i32 const A0 = (i32)sh.hA0[t], S0 = (i32)sh.hS0[t];
        bool const on = mode == 0u ? j != 0u : (mode == 1u ? (pC[l] & 0x80000000u) != 0u : true);
        pLim[l] = mode == 0u ? (u32)zd_hp_edge(A0, S0, j) : (mode == 1u ? (u32)zd_hp_edge(A0, S0, j + 1u) : 0xFFFFFFFFu);        // (-1: below every position)
        pCap[l] = !on ? 0u : (mode == 0u ? ZD_HP_SYNC : (mode == 1u ? sh.hN[t] + 1u : pC[l]));
        pK[l] = 0;
    }
    g.sync();
    for (;;) {
        GRP_SERIAL(g) { sh.hpAgain = 0; }
        g.sync();
        // stage: the 64 bytes under every working lane's cursor, never below its stream's first byte
        GRP_FOR(g, i, 64u * (ZD_HP_WIN / 16u)) {
            u32 const l = i / (ZD_HP_WIN / 16u), c = i % (ZD_HP_WIN / 16u);
            if ((i32)pA[l] > (i32)pLim[l] && pK[l] < pCap[l]) {
                u32 const startByte = sh.hS0[l / ZD_HP_LANES] >> 3, be = (pA[l] + 7u) >> 3;
                u32 lo = be > ZD_HP_WIN ? be - ZD_HP_WIN : 0u; if (lo < startByte) lo = startByte;
                u32 const o = lo + 16u * c;
                u64 a = 0, b = 0;
                if (o + 16u <= bsize) { a = ld64(bsrc + o); b = ld64(bsrc + o + 8); }
                else { for (u32 k = 0; k < 16u; k++) { if (o + k < bsize) { if (k < 8u) a |= (u64)bsrc[o + k] << (8u * k); else b |= (u64)bsrc[o + k] << (8u * (k - 8u)); } } }
                st64(hpWin + l * ZD_HP_STRIDE + 16u * c, a); st64(hpWin + l * ZD_HP_STRIDE + 16u * c + 8u, b);
                if (c == 0) sh.hpAgain = 1;
            }
        }
        g.sync();
        if (!ZJ_UNI(sh.hpAgain)) break;
        GRP_FOR(g, l, 64u) {
            i32 a = (i32)pA[l]; i32 const lim = (i32)pLim[l]; u32 k = pK[l]; u32 const cap = pCap[l];
            if (a > lim && k < cap) {
                i32 const S0 = (i32)sh.hS0[l / ZD_HP_LANES];
                u32 const startByte = (u32)S0 >> 3, be = ((u32)a + 7u) >> 3;
                u32 lo = be > ZD_HP_WIN ? be - ZD_HP_WIN : 0u; if (lo < startByte) lo = startByte;
                const u8* const win = hpWin + l * ZD_HP_STRIDE;
                u8* const dst = mode == 2u ? out + pO[l] : nullptr;
                u32 k0 = k; u64 acc = 0;                                  // W: symbols k0.. of this lane gather in acc, 8 to a store
                while (a > lim && k < cap) {
                    u32 const byteEnd = ((u32)a + 7u) >> 3;
                    u64 c;
                    if (byteEnd >= lo + 8u) c = ld64(win + (byteEnd - 8u - lo)) << (8u * byteEnd - (u32)a);       // >= 57 bits below a
                    else if (lo == startByte) c = ld64(win) << (64u - (u32)(a - S0));                              // fewer than 64 bits of the stream left: zeros below its first bit
                    else break;                                                                                    // the window is used up
                    for (u32 q = 0; q < 5u && a > lim && k < cap; q++) {
                        u32 const cell = sh.huf[(u32)(c >> (64u - log))], nb = cell >> 8;
                        c <<= nb; a -= (i32)nb;
                        if (mode == 2u) {
                            acc |= (u64)(cell & 0xFFu) << (8u * ((k - k0) & 7u));
                            if (((k - k0) & 7u) == 7u) { st64(dst + (k - 7u), acc); acc = 0; }
                        }
                        k++;
                    }
                }
                if (mode == 2u) { for (u32 r = k - ((k - k0) & 7u); r < k; r++) { dst[r] = (u8)acc; acc >>= 8; } }
                pA[l] = (u32)a; pK[l] = k;
            }
        }
        g.sync();
    }
}
// true (wave-uniform): out[0, litSize) holds the section's literals.  false: nothing decided, the caller's rounds run as if this had not been tried.
// hpWin: ZD_HP_LDS bytes of LDS for the lanes' windows.
template <class G>
ZJ_DEV bool zd_huf_streams_wave(const G& g, ZDecShared& sh, const u8* bsrc, u32 bsize, u8* out, u8* hpWin) {
    if (!hpWin || ZJ_UNI(sh.litStreams) != 4u || ZJ_UNI(sh.hufLog) > ZD_HUF_CELLS_LOG || ZJ_UNI(sh.litSize) < ZD_HP_MIN_LIT) return false;
    u32 const gcd = ZJ_UNI(sh.hufGcd);
    u32* const pS = sh.sLit; u32* const pE = sh.sMl; u32* const pC = sh.sOff;          // per lane: start, end, codes | 1 << 31 "decode (again)"   (the execution stage's arrays: idle here)
    u32* const pA = (u32*)&sh.hstage[0][0]; u32* const pK = pA + 64; u32* const pO = (u32*)&sh.win[0];       // running position, running count, place in the output (idle too: the rounds' staging)
    // pass G
    GRP_FOR(g, l, 64u) {
        u32 const t = l / ZD_HP_LANES, j = l % ZD_HP_LANES;
        i32 const A0 = (i32)sh.hA0[t], S0 = (i32)sh.hS0[t];
        i32 const edge = zd_hp_edge(A0, S0, j);
        i32 from = edge + (i32)ZD_HP_SYNC < A0 ? edge + (i32)ZD_HP_SYNC : A0;
        from = A0 - (i32)((u32)(A0 - from) / gcd * gcd);                                // on the lattice of the code lengths' common divisor
        pA[l] = (u32)(j ? from : A0);
    }
    g.sync();
    zd_hp_pass(g, sh, bsrc, bsize, hpWin, nullptr, 0u);
    GRP_FOR(g, l, 64u) { pS[l] = pA[l]; pC[l] = 0x80000000u; }
    g.sync();
    // pass V until nothing changes
    for (u32 tries = 0; ; tries++) {
        GRP_FOR(g, l, 64u) { if (pC[l] & 0x80000000u) pA[l] = pS[l]; }
        g.sync();
        zd_hp_pass(g, sh, bsrc, bsize, hpWin, nullptr, 1u);
        GRP_FOR(g, l, 64u) { if (pC[l] & 0x80000000u) { pE[l] = pA[l]; pC[l] = pK[l]; } }
        GRP_SERIAL(g) { sh.hpAgain = 0; }
        g.sync();
        GRP_FOR(g, l, 64u) {
            if ((l % ZD_HP_LANES) && pE[l - 1u] != pS[l]) { pS[l] = pE[l - 1u]; pC[l] = 0x80000000u; sh.hpAgain = 1; ZD_HP_STAT(4, 1); }
        }
        g.sync();
        if (!ZJ_UNI(sh.hpAgain)) break;
        if (tries == ZD_HP_TRIES) { ZD_HP_STAT(2, 1); return false; }
        ZD_HP_STAT(3, 1);
    }
    // check
    GRP_SERIAL(g) {
        u32 ok = 1;
        for (u32 t = 0; t < 4u; t++) {
            u32 sum = 0; for (u32 j = 0; j < ZD_HP_LANES; j++) sum += pC[t * ZD_HP_LANES + j];
            if (sum != sh.hN[t] || pE[t * ZD_HP_LANES + ZD_HP_LANES - 1u] != sh.hS0[t]) ok = 0;
        }
        sh.hpAgain = ok;
    }
    g.sync();
    if (!ZJ_UNI(sh.hpAgain)) { ZD_HP_STAT(1, 1); return false; }
    ZD_HP_STAT(0, 1);
    // pass W
    GRP_FOR(g, l, 64u) {
        u32 const t = l / ZD_HP_LANES, j = l % ZD_HP_LANES;
        u32 at = sh.hDst[t]; for (u32 q = 0; q < j; q++) at += pC[t * ZD_HP_LANES + q];
        pO[l] = at; pA[l] = pS[l];
    }
    g.sync();
    zd_hp_pass(g, sh, bsrc, bsize, hpWin, out, 2u);
    zj_mem_order();
    g.sync();
    return true;
}

// ------------------------------------------------------------------ block --------------------
// Literals section of one compressed block: parses its header and regenerates the literals (raw: in place;
// RLE / Huffman: into litScratch).  Returns where they are; sets sh.err and returns nullptr on error.
// Publishes sh.litSize / litHdr / litCSize.  room = bytes left in the frame's destination.
template <class G>
// preLit: the block's Huffman-coded literals were regenerated there by an earlier pass (zd_lit_frame, zj_decode_split.h) — the header is
// parsed as always, the table and the streams are not touched again.
// tableOnly: build the block's Huffman table (sh.huf, sh.hufValid, sh.hufX2) and stop — for a later treeless block whose literals are decoded apart from this one's
// (zj_decode_split.h, multi-block frames); returns bsrc then.
ZJ_DEV const u8* zd_block_literals(const G& g, ZDecShared& sh, const u8* bsrc, u32 bsize, u8* litScratch, ZjProf& pf, u32 room = ~0u, const u8* preLit = nullptr, bool tableOnly = false, u8* hpWin = nullptr) {
    // ---- literals section header (lane 0) : N/decompress/zstd_decompress_block.c:134-340
    GRP_SERIAL(g) {
        u32 err = 0;
        if (bsize < 2) err = ZJ_E_CORRUPTION;
        else {
            u32 const b0 = bsrc[0], type = b0 & 3, fmt = (b0 >> 2) & 3;
            u32 lh = 0, n = 0, c = 0, streams = 1;
            if (type < 2) {
                if (fmt == 0 || fmt == 2) { lh = 1; n = b0 >> 3; }
                else if (fmt == 1) { lh = 2; n = ld16(bsrc) >> 4; }
                else { if (bsize < 3) err = ZJ_E_CORRUPTION; else { lh = 3; n = ld24(bsrc) >> 4; } }
                c = (type == 0) ? n : 1;
            } else if (type == 3 && !sh.hufValid) err = ZJ_E_DICT_CORRUPTED;
            else if (bsize < 5) err = ZJ_E_CORRUPTION;
            else {
                u32 const lhc = ld32(bsrc);
                if (fmt < 2) { streams = fmt ? 4 : 1; lh = 3; n = (lhc >> 4) & 0x3FF; c = (lhc >> 14) & 0x3FF; }
                else if (fmt == 2) { streams = 4; lh = 4; n = (lhc >> 4) & 0x3FFF; c = lhc >> 18; }
                else { streams = 4; lh = 5; n = (lhc >> 4) & 0x3FFFF; c = (lhc >> 22) + ((u32)bsrc[4] << 10); }
            }
            // the reference's order of refusals (zstd_decompress_block.c:150-183, :250-276, :300-326): literals larger than a
            // block, 4 streams for < 6 literals, section larger than the block — and literals larger than the room left in the
            // destination before (raw / RLE) or after (Huffman) the section-size test
            if (!err && type == 1 && lh + 1 > bsize) err = ZJ_E_CORRUPTION;     // RLE: the byte must be there before anything else (:310, :315)
            if (!err && n > sh.blockSizeMax) err = ZJ_E_CORRUPTION;
            if (!err && type >= 2 && streams == 4 && n < 6) err = ZJ_E_LITERALS_HEADER;
            if (!err && type < 2 && n > room) err = ZJ_E_DSTSIZE_TOO_SMALL;
            if (!err && lh + c > bsize) err = ZJ_E_CORRUPTION;
            if (!err && n > room) err = ZJ_E_DSTSIZE_TOO_SMALL;
            sh.litType = type; sh.litSize = n; sh.litCSize = c; sh.litHdr = lh; sh.litStreams = streams;
        }
        if (err) sh.err = err;
    }
    g.sync();
    if (ZJ_UNI(sh.err)) return nullptr;

    pf.mark(0);
    u32 const litType = ZJ_UNI(sh.litType), litSize = ZJ_UNI(sh.litSize), litHdr = ZJ_UNI(sh.litHdr), litCSize = ZJ_UNI(sh.litCSize);
    const u8* lit = litScratch;
    if (tableOnly && litType != 2) return nullptr;
    if (litType == 0) lit = bsrc + litHdr;                      // raw: read in place
    else if (litType == 1) { zd_fill(g, litScratch, bsrc[litHdr], litSize); zj_mem_order(); }
    else if (preLit) lit = preLit;
    else {
        // ---- Huffman table (lane 0 reads weights, all lanes fill) ----
        if (litType == 2) {
            GRP_SERIAL(g) {
                u32 nbSym = 0;
                u32 const h = zd_huf_read_weights(sh, bsrc + litHdr, litCSize, &nbSym);
                if (!h || h > litCSize) sh.err = ZJ_E_CORRUPTION;
                sh.litSrcOff = litHdr + h; sh.bN = nbSym;
                // table shape the reference builds here: HUF_decompress4X_hufOnly_wksp asks HUF_selectDecoder (huf_decompress.c:1930);
                // a single stream always takes the one-code table (zstd_decompress_block.c:219)
                sh.hufX2 = sh.litStreams == 4 ? zd_huf_select_x2(litSize, litCSize) : 0u;
            }
            g.sync();
            if (ZJ_UNI(sh.err)) return nullptr;
            zd_huf_fill(g, sh, ZJ_UNI(sh.bN));
            GRP_SERIAL(g) { sh.hufValid = 1; }
            if (tableOnly) { g.sync(); return bsrc; }
        } else { GRP_SERIAL(g) { sh.litSrcOff = litHdr; } }
        g.sync();
        pf.mark(1);
        // ---- stream descriptors (lane 0) ----
        GRP_SERIAL(g) {
            u32 const off = sh.litSrcOff, end = litHdr + litCSize;   // compressed literal bytes [off, end)
            u32 err = 0;
            if (sh.litStreams == 1) {
                if (end <= off || bsrc[end - 1] == 0) err = ZJ_E_CORRUPTION;
                else { sh.hS0[0] = off * 8; sh.hA[0] = sh.hA0[0] = (end - 1) * 8 + zj_hibit(bsrc[end - 1]); sh.hN[0] = litSize; sh.hDst[0] = 0; }
                for (u32 t = 1; t < 4; t++) { sh.hN[t] = 0; sh.hA[t] = 0; sh.hS0[t] = 0; sh.hDst[t] = 0; }
            } else if (end < off + 10) err = ZJ_E_CORRUPTION;
            else {
                u32 const l1 = ld16(bsrc + off), l2 = ld16(bsrc + off + 2), l3 = ld16(bsrc + off + 4);
                u32 const seg = (litSize + 3) / 4;
                u32 b = off + 6;
                if (b + l1 + l2 + l3 > end || 3 * seg > litSize) err = ZJ_E_CORRUPTION;
                else {
                    sh.hCnt[0] = l1; sh.hCnt[1] = l2; sh.hCnt[2] = l3; sh.hCnt[3] = end - b - l1 - l2 - l3;
                    for (u32 t = 0; t < 4 && !err; t++) {
                        u32 const len = sh.hCnt[t], e = b + len;
                        if (len == 0 || bsrc[e - 1] == 0) { err = ZJ_E_CORRUPTION; break; }
                        sh.hS0[t] = b * 8; sh.hA[t] = sh.hA0[t] = (e - 1) * 8 + zj_hibit(bsrc[e - 1]);
                        sh.hN[t] = (t < 3) ? seg : litSize - 3 * seg; sh.hDst[t] = t * seg;
                        b = e;
                    }
                }
            }
            for (u32 t = 0; t < 4; t++) sh.hDone[t] = 0;
            if (err) sh.err = err;
        }
        g.sync();
        if (ZJ_UNI(sh.err)) return nullptr;
        // ---- the four streams by the whole wave where that applies (round 6), else ----
        if (zd_huf_streams_wave(g, sh, bsrc, bsize, litScratch, hpWin)) return lit;
        // ---- rounds: stage windows (all lanes) -> decode (<=4 lanes) -> flush (all lanes) ----
        {   u32 const maxN = ZJ_UNI(zj_max(zj_max(sh.hN[0], sh.hN[1]), zj_max(sh.hN[2], sh.hN[3])));
            u32 const rounds = (maxN + ZD_HSYM - 1) / ZD_HSYM;
            for (u32 r = 0; r < rounds; r++) {
                GRP_FOR(g, i, 4u * (ZD_HWIN_STRIDE / 16u)) {
                    u32 const t = i / (ZD_HWIN_STRIDE / 16u), c = i % (ZD_HWIN_STRIDE / 16u);
                    if (sh.hDone[t] < sh.hN[t]) {
                        u32 const startByte = sh.hS0[t] >> 3;
                        u32 const hi = ((sh.hA[t] + 7) >> 3);
                        u32 lo = hi > ZD_HWIN ? hi - ZD_HWIN : 0;
                        if (lo < startByte) lo = startByte;
                        u32 const o = lo + 16 * c;
                        u64 a = 0, b = 0;
                        if (o + 16 <= bsize) { a = ld64(bsrc + o); b = ld64(bsrc + o + 8); }
                        else { for (u32 k = 0; k < 16; k++) { if (o + k < bsize) { if (k < 8) a |= (u64)bsrc[o + k] << (8 * k); else b |= (u64)bsrc[o + k] << (8 * (k - 8)); } } }
                        st64(sh.win + t * ZD_HWIN_STRIDE + 16 * c, a); st64(sh.win + t * ZD_HWIN_STRIDE + 16 * c + 8, b);
                        if (c == 0) sh.hLo[t] = lo;
                    }
                }
                g.sync();
                GRP_FOR(g, t, 4) { if (sh.hDone[t] < sh.hN[t]) zd_huf_stream_round(sh, t); else sh.hCnt[t] = 0; }
                g.sync();
                GRP_FOR(g, i, 4u * ZD_HSYM) {
                    u32 const t = i / ZD_HSYM, j = i % ZD_HSYM;
                    if (j < sh.hCnt[t]) litScratch[sh.hDst[t] + sh.hDone[t] + j] = sh.hstage[t][j];
                }
                g.sync();
                GRP_FOR(g, t, 4) sh.hDone[t] += sh.hCnt[t];
                g.sync();
            }
            GRP_SERIAL(g) {
                for (u32 t = 0; t < sh.litStreams; t++) {           // every stream ends exactly on its first bit (huf_decompress.c:697, :830)
                    if (sh.hA[t] != sh.hS0[t] && !(sh.hufX2 && zd_huf_x2_accepts(sh, bsrc, t, litScratch + sh.hDst[t]))) sh.err = ZJ_E_CORRUPTION;
                }
            }
            g.sync();
            if (ZJ_UNI(sh.err)) return nullptr;
        }
    }
    return lit;
}

// Sequences section header + the three tANS tables into sh.ll/of/ml (N/decompress/zstd_decompress_block.c:695-782,
// :485-603).  Publishes sh.nbSeq, sh.seqOff (first byte of the bitstream), sh.llLog/ofLog/mlLog; sets sh.err.
// `bm` (optional): ZD_FSE_BM_WORDS words of LDS the caller can spare — the tables' second pass then runs on the whole wave (zd_fse_finish_wave)
template <class G>
ZJ_DEV void zd_seq_tables(const G& g, ZDecShared& sh, const u8* bsrc, u32 bsize, u32 seqSecOff, u32* bm = nullptr) {
    GRP_SERIAL(g) {
        u32 err = 0, ip = seqSecOff, nbSeq = 0;
        if (ip >= bsize) err = ZJ_E_SRCSIZE_WRONG;
        else {
            nbSeq = bsrc[ip++];
            if (nbSeq > 0x7F) {
                if (nbSeq == 0xFF) { if (ip + 2 > bsize) err = ZJ_E_SRCSIZE_WRONG; else { nbSeq = ld16(bsrc + ip) + 0x7F00; ip += 2; } }
                else { if (ip >= bsize) err = ZJ_E_SRCSIZE_WRONG; else nbSeq = ((nbSeq - 0x80) << 8) + bsrc[ip++]; }
            }
        }
        if (!err) {
            if (nbSeq == 0) { if (ip != bsize) err = ZJ_E_CORRUPTION; }
            else if (ip + 1 > bsize) err = ZJ_E_SRCSIZE_WRONG;
            else {
                u32 const modes = bsrc[ip++];
                if (modes & 3) err = ZJ_E_CORRUPTION;
                // walk the three table descriptions; NCount parsing is inherently serial (each header starts where the
                // previous one ends), building the tables from the parsed counts is not
                for (u32 t = 0; t < 3 && !err; t++) {
                    u32 const mode = (modes >> (6 - 2 * t)) & 3;
                    sh.tblMode[t] = mode;
                    if (mode == 1) {
                        if (ip >= bsize || bsrc[ip] > (t == 0 ? 35u : (t == 1 ? 31u : 52u))) err = ZJ_E_CORRUPTION; else { sh.tblMax[t] = bsrc[ip]; ip++; }
                    } else if (mode == 2) {
                        u32 max = (t == 0 ? 35u : (t == 1 ? 31u : 52u)), tl = 0;
                        u32 const h = zd_read_ncount(sh.norm[t], &max, &tl, bsrc + ip, bsize - ip);
                        if (!h || h > bsize - ip || tl > (t == 1 ? 8u : 9u)) err = ZJ_E_CORRUPTION;
                        else { sh.tblMax[t] = max; sh.tblLog[t] = tl; ip += h; }     // built below, three tables on three lanes
                    } else if (mode == 3) { if (!sh.seqValid) err = ZJ_E_CORRUPTION; }
                }
                sh.seqValid = 1;
            }
        }
        sh.nbSeq = nbSeq; sh.seqOff = ip;
        if (err) sh.err = err;
    }
    g.sync();
    if (ZJ_UNI(sh.err)) return;
    if (ZJ_UNI(sh.nbSeq)) {
        // the three tables on lanes 0..2: described (from the counts just parsed), predefined or RLE
        GRP_FOR(g, t, 3) {
            u32 const mode = sh.tblMode[t];
            u32* cells = t == 0 ? sh.ll : (t == 1 ? sh.of : sh.ml);
            if (mode == 2) {
                u32 const tl = sh.tblLog[t];
                if (!(bm ? zd_fse_spread(cells, sh.norm[t], sh.symNext + 64 * t, sh.tblMax[t], tl) : zd_build_fse(cells, sh.norm[t], sh.symNext + 64 * t, sh.tblMax[t], tl, t))) sh.err = ZJ_E_CORRUPTION;
                if (t == 0) sh.llLog = tl; else if (t == 1) sh.ofLog = tl; else sh.mlLog = tl;
            } else if (mode == 0) {
                u32 const log = (t == 1) ? 5 : 6;
                const short* const dn = t == 0 ? zd_k_ll_defnorm : (t == 1 ? zd_k_of_defnorm : zd_k_ml_defnorm); u32 const dmax = t == 0 ? 35 : (t == 1 ? 28 : 52);
                if (bm) zd_fse_spread(cells, dn, sh.symNext + 64 * t, dmax, log); else zd_build_fse(cells, dn, sh.symNext + 64 * t, dmax, log, t);
                if (t == 0) sh.llLog = log; else if (t == 1) sh.ofLog = log; else sh.mlLog = log;
            } else if (mode == 1) {
                u32 const sym = sh.tblMax[t];
                cells[0] = ZD_CELL(0, 0, sym, zd_extra_bits(t, sym));
                if (t == 0) sh.llLog = 0; else if (t == 1) sh.ofLog = 0; else sh.mlLog = 0;
            }
        }
        g.sync();
        if (bm && !ZJ_UNI(sh.err)) {                   // the spread tables' second pass, a table at a time on the whole wave
            for (u32 t = 0; t < 3u; t++) {
                u32 const mode = ZJ_UNI(sh.tblMode[t]);
                if (mode != 0u && mode != 2u) continue;
                zd_fse_finish_wave(g, t == 0 ? sh.ll : (t == 1 ? sh.of : sh.ml), sh.symNext + 64 * t, ZJ_UNI(t == 0 ? sh.llLog : (t == 1 ? sh.ofLog : sh.mlLog)), t, bm);
            }
        }
    }
}

// Decodes one compressed block [bsrc, bsrc+bsize) of the frame whose output starts at `out`
// (frame-relative position `opos`; the frame may write out[0..frameCap)).  Returns new opos (or sets sh.err).
template <bool DICT = false, class G>
ZJ_DEV u32 zd_compressed_block(const G& g, ZDecShared& sh, const u8* bsrc, u32 bsize, u8* out, u32 opos, u32 frameCap, u8* litScratch, ZjProf& pf, const u8* dictEnd = nullptr) {
    const u8* const lit = zd_block_literals(g, sh, bsrc, bsize, litScratch, pf, frameCap - opos);
    if (ZJ_UNI(sh.err)) return opos;
    u32 const litSize = ZJ_UNI(sh.litSize), litHdr = ZJ_UNI(sh.litHdr), litCSize = ZJ_UNI(sh.litCSize);
    // How far this block may write.  The reference parks regenerated literals 32 bytes past the largest block the frame allows
    // when the destination has room for that (ZSTD_allocateLiteralsBuffer, zstd_decompress_block.c:86-94) and stops sequences
    // there (:1405, oend = litBuffer); otherwise (little room, or raw literals read in place :283-288) at the destination's end.
    // Either way the refusal is dstSize_tooSmall; a block that ends inside the 32-byte slack goes through.
    u32 dstCap = frameCap;
    {   u32 const room = frameCap - opos, bsm = ZJ_UNI(sh.blockSizeMax);
        bool const inPlace = ZJ_UNI(sh.litType) == 0 && litHdr + litSize + 32u <= bsize;
        if (!inPlace && (u64)room > (u64)bsm + 64u + litSize) dstCap = opos + bsm + 32u; }

    pf.mark(2);
    // ---- sequences header + tables ----
    zd_seq_tables(g, sh, bsrc, bsize, litHdr + litCSize);
    if (ZJ_UNI(sh.err)) return opos;
    u32 const nbSeq = ZJ_UNI(sh.nbSeq);
    u32 litUsed = 0;
    if (nbSeq && frameCap == opos) {                            // sequences but no room at all: zstd_decompress_block.c:2119
        GRP_SERIAL(g) { sh.err = ZJ_E_DSTSIZE_TOO_SMALL; } g.sync(); return opos; }
    if (nbSeq) {
        pf.mark(3);
        ZDecSeqPriv p;
        GRP_SERIAL(g) {
            u32 const s = sh.seqOff;
            u32 err = 0;
            p.i = 0; p.opos = opos; p.lpos = 0; p.rep0 = sh.rep[0]; p.rep1 = sh.rep[1]; p.rep2 = sh.rep[2];
            p.S0 = (i32)(s * 8); p.A = p.S0; p.sLL = p.sOF = p.sML = 0;
            if (s >= bsize || bsrc[bsize - 1] == 0) err = ZJ_E_CORRUPTION;
            else p.A = (i32)((bsize - 1) * 8 + zj_hibit(bsrc[bsize - 1]));
            {   u32 const hi = bsize + 8; u32 lo = hi > ZD_SWIN ? hi - ZD_SWIN : 0; if (lo < s) lo = s; sh.winLo = lo; }
            sh.seqDone = 0;
            if (err) sh.err = err;
        }
        g.sync();
        if (ZJ_UNI(sh.err)) return opos;
        bool first = true;
        for (;;) {
            // stage the bitstream window [winLo, winLo + ZD_SWIN)
            zd_stage(g, sh.win, bsrc, ZJ_UNI(sh.winLo), ZD_SWIN, bsize);
            g.sync();
            GRP_SERIAL(g) {
                if (first) {   // initial states: LL, OF, ML (N/decompress/zstd_decompress_block.c:1640-1642)
                    u32 const a = sh.llLog, b = sh.ofLog, c = sh.mlLog;
                    ZDecSeqPriv const p0 = p;
                    p.A -= (i32)a; p.sLL = zd_bits(sh.win, sh.winLo, p.A, a);
                    p.A -= (i32)b; p.sOF = zd_bits(sh.win, sh.winLo, p.A, b);
                    p.A -= (i32)c; p.sML = zd_bits(sh.win, sh.winLo, p.A, c);
                    if (p.A < p.S0) sh.err = zd_seq_dry_code(sh, p0, true, dstCap);
                }
                if (!sh.err) zd_seq_batch(sh, p, dstCap);
            }
            first = false;
            g.sync();
            pf.mark(4);
            if (ZJ_UNI(sh.err)) return opos;
            // ---- execute the batch: N/decompress/zstd_decompress_block.c:1001-1096 ----
            {   u32 const n = ZJ_UNI(sh.bN);
                u32 const lp = ZJ_UNI(sh.bLitStart), op = ZJ_UNI(sh.bOutStart);
                u32 const lt = ZJ_UNI(sh.bLitTotal), ot = ZJ_UNI(sh.bOutTotal);
                u32 lt2, ot2;
                zd_execute_batch<DICT>(g, sh, out, lit, n, lp, op, lt2, ot2, nullptr, 0, dictEnd);
                litUsed = lp + lt; opos = op + ot;
            }
            pf.mark(5);
            if (ZJ_UNI(sh.seqDone)) break;
            g.sync();
        }
        GRP_SERIAL(g) { sh.rep[0] = p.rep0; sh.rep[1] = p.rep1; sh.rep[2] = p.rep2; }
    }
    // ---- last literals ----
    {   u32 const rest = litSize - litUsed;
        if ((u64)opos + rest > dstCap) { GRP_SERIAL(g) { sh.err = ZJ_E_DSTSIZE_TOO_SMALL; } g.sync(); return opos; }
        grp_copy_wide(g, out + opos, lit + litUsed, rest);
        opos += rest;
        zj_mem_order();
    }
    g.sync();
    pf.mark(6);
    return opos;
}

// ------------------------------------------------------------------ frame --------------------
// Decodes all frames in [src, src+srcSize) into dst[0..dstCap).  Returns decoded size or
// ZJ_ERR64(code) — the reference's size_t error convention (N/common/error_private.h).
template <bool DICT = false, class G>
ZJ_DEV u64 zd_decompress(const G& g, ZDecShared& sh, const u8* src, u32 srcSize, u8* dst, u32 dstCap, u8* litScratch, ZjProf& pf,
                         const ZDDictDev* ddArg = nullptr, const u8* dictRaw = nullptr) {
    const ZDDictDev* const dd = DICT ? ddArg : nullptr;     // the no-dictionary instantiation carries none of the dictionary code
    // dd: digested dictionary (ZSTD_decompress_usingDDict, N/decompress/zstd_decompress.c:1661-1672); every frame of the
    // buffer starts from its entropy tables / repcodes and may copy from its content
    const u8* const dictEnd = dd ? dictRaw + dd->contentOff + dd->contentSize : nullptr;
    GRP_SERIAL(g) {
        sh.err = 0; sh.dictSize = dd ? dd->contentSize : 0u;
    }
    GRP_FOR(g, i, 36) sh.llBase[i] = zd_k_ll_base[i];
    GRP_FOR(g, i, 53) sh.mlBase[i] = zd_k_ml_base[i];
    g.sync();
    u32 ipos = 0, total = 0, nbFrames = 0;
    while (ipos < srcSize) {
        // ---- frame header (lane 0): N/decompress/zstd_decompress.c:447-557 ----
        GRP_SERIAL(g) {
            const u8* p = src + ipos; u32 const left = srcSize - ipos; u32 err = 0;
            sh.blkType = 0;   // re-used as "skippable" flag below
            if (left < 5) err = ZJ_E_SRCSIZE_WRONG;
            else {
                u32 const magic = ld32(p);
                if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
                    if (left < 8 || (u64)8 + ld32(p + 4) > left) err = ZJ_E_SRCSIZE_WRONG;
                    else { sh.blkType = 1; sh.hdrSize = 8 + ld32(p + 4); }
                } else {
                    u32 const fhd = p[4], didc = fhd & 3, single = (fhd >> 5) & 1, fcsid = fhd >> 6;
                    u32 const didSz = didc == 3 ? 4 : didc, fcsSz = fcsid == 0 ? single : (1u << fcsid);
                    u32 const need = 5 + !single + didSz + fcsSz; u32 pos = 5;
                    u64 window = 0, content = ~(u64)0; u32 dictID = 0;
                    // the header's size is taken from its descriptor byte and checked (with the first block header) before the
                    // magic number is looked at (ZSTD_decompressFrame, zstd_decompress.c:966-979); garbage after a complete frame
                    // is srcSize_wrong, not prefix_unknown (ZSTD_decompressMultiFrame :1136-1143)
                    if (left < 9 || left < need + 3) err = ZJ_E_SRCSIZE_WRONG;
                    else if (magic != 0xFD2FB528u) err = nbFrames ? ZJ_E_SRCSIZE_WRONG : ZJ_E_PREFIX_UNKNOWN;
                    else if (fhd & 8) err = ZJ_E_FRAMEPARAM_UNSUPPORTED;
                    else {
                        if (!single) { u32 const wd = p[pos++], wl = (wd >> 3) + 10; if (wl > 31) err = ZJ_E_WINDOW_TOO_LARGE; window = (u64)1 << wl; window += (window >> 3) * (wd & 7); }
                        if (didc == 1) dictID = p[pos]; else if (didc == 2) dictID = ld16(p + pos); else if (didc == 3) dictID = ld32(p + pos);
                        pos += didSz;
                        if (fcsid == 0) { if (single) content = p[pos]; } else if (fcsid == 1) content = ld16(p + pos) + 256; else if (fcsid == 2) content = ld32(p + pos); else content = ld64(p + pos);
                        if (single) window = content;
                        if (!err && dictID && dictID != (dd ? dd->dictID : 0u)) err = ZJ_E_DICT_WRONG;   // zstd_decompress.c:717
                        sh.hdrSize = need; sh.contentSize = content; sh.hasChecksum = (fhd >> 2) & 1;
                        sh.blockSizeMax = window < ZD_BLOCK_MAX ? (u32)window : ZD_BLOCK_MAX;
                    }
                }
            }
            sh.rep[0] = 1; sh.rep[1] = 4; sh.rep[2] = 8; sh.hufValid = 0; sh.seqValid = 0; sh.hufX2 = 0;
            if (dd && dd->hasEntropy) {
                sh.rep[0] = dd->rep[0]; sh.rep[1] = dd->rep[1]; sh.rep[2] = dd->rep[2]; sh.hufValid = 1; sh.seqValid = 1; sh.hufX2 = 1;   // ZSTD_loadDEntropy builds the two-code table (zstd_decompress.c:1473)
                sh.hufLog = dd->hufLog; sh.hufW1 = dd->hufW1; sh.hufGcd = 1; sh.llLog = dd->llLog; sh.ofLog = dd->ofLog; sh.mlLog = dd->mlLog;
            }
            if (err) sh.err = err;
        }
        g.sync();
        if (ZJ_UNI(sh.err)) return ZJ_ERR64(ZJ_UNI(sh.err));
        if (ZJ_UNI(sh.blkType) == 1) { ipos += ZJ_UNI(sh.hdrSize); g.sync(); continue; }
        if (dd && dd->hasEntropy) { zd_load_dict_entropy(g, sh, dd, true, true); g.sync(); }   // litEntropy = fseEntropy = 1 (zstd_decompress.c:1554)
        ipos += ZJ_UNI(sh.hdrSize);
        u8* const fout = dst + total; u32 const fcap = dstCap - total;
        u32 opos = 0;
        for (;;) {
            GRP_SERIAL(g) {
                u32 err = 0;
                if (srcSize - ipos < 3) err = ZJ_E_SRCSIZE_WRONG;
                else {
                    u32 const bh = ld24(src + ipos), type = (bh >> 1) & 3, sz = bh >> 3;
                    sh.blkLast = bh & 1; sh.blkType = type; sh.blkSize = sz;
                    if (type == 3) err = ZJ_E_CORRUPTION;
                    // the one-shot frame loop bounds raw and RLE blocks by the destination only (zstd_decompress.c:1020-1026); a
                    // compressed block larger than blockSizeMax is srcSize_wrong (zstd_decompress_block.c:2081)
                    else if (type == 1) { if (srcSize - ipos - 3 < 1) err = ZJ_E_SRCSIZE_WRONG; else if (sz > fcap - opos) err = ZJ_E_DSTSIZE_TOO_SMALL; }
                    else {
                        if (sz > srcSize - ipos - 3) err = ZJ_E_SRCSIZE_WRONG;
                        else if (type == 0 && sz > fcap - opos) err = ZJ_E_DSTSIZE_TOO_SMALL;
                        else if (type == 2 && sz > sh.blockSizeMax) err = ZJ_E_SRCSIZE_WRONG;      // (a compressed block of exactly blockSizeMax is entered: zstd_decompress_block.c:2073-2081)
                    }
                }
                if (err) sh.err = err;
            }
            g.sync();
            if (ZJ_UNI(sh.err)) return ZJ_ERR64(ZJ_UNI(sh.err));
            ipos += 3;
            u32 const type = sh.blkType, sz = sh.blkSize, last = sh.blkLast;
            if (type == 0) { grp_copy_wide(g, fout + opos, src + ipos, sz); opos += sz; ipos += sz; zj_mem_order(); pf.mark(7); }
            else if (type == 1) { zd_fill(g, fout + opos, src[ipos], sz); opos += sz; ipos += 1; zj_mem_order(); }
            else {
                opos = zd_compressed_block<DICT>(g, sh, src + ipos, sz, fout, opos, fcap, litScratch, pf, dictEnd);
                if (ZJ_UNI(sh.err)) return ZJ_ERR64(ZJ_UNI(sh.err));
                ipos += sz;
            }
            g.sync();
            if (last) break;
        }
        {   u64 const cs = zj_uni64(sh.contentSize); if (cs != ~(u64)0 && cs != opos) return ZJ_ERR64(ZJ_E_CORRUPTION); }
        if (ZJ_UNI(sh.hasChecksum)) {
            // frame content checksum: low 32 bits of XXH64(content, 0), N/decompress/zstd_decompress.c:1050-1060
            if (srcSize - ipos < 4) return ZJ_ERR64(ZJ_E_CHECKSUM_WRONG);
            zj_mem_order(); g.sync();
            if ((u32)zj_xxh64(g, fout, opos) != ZJ_UNI(ld32(src + ipos))) return ZJ_ERR64(ZJ_E_CHECKSUM_WRONG);
            ipos += 4;
        }
        total += opos; nbFrames++;
    }
    return total;
}
