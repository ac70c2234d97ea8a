// zj_encode.h — batched zstd frame encoder for gfx950: one buffer (= one frame = one block) per
// wavefront.
//
// Replaces, for batches of independent buffers, what zstd-jni reaches through
//   ZstdCompressCtx.compress*0 -> ZSTD_CCtx_reset + ZSTD_compress2        (reference N/jni_fast_zstd.c:586-640)
// at levels 1..3, i.e. N/compress/zstd_compress.c:4695-4745 (frame header), :4383-4448 (block),
// :2888-3043 (entropy stage), N/compress/zstd_fast.c:192-423 (level 1-2 match finder),
// N/compress/zstd_double_fast.c:105-323 (level 3), N/compress/zstd_compress_literals.c:129-235 +
// N/compress/huf_compress.c (Huffman literals), N/compress/zstd_compress_sequences.c:157-382 +
// N/compress/fse_compress.c (tANS sequences), N/compress/hist.c.     N/ = src/main/native/.
//
// Output contract: byte-identical to the reference's ZSTD_compress2 for levels 1-3 at every size the batch entries take, with the
// reference's own table sizes for the input (level 3 at 64 KiB: hashLog / chainLog = 16 / 15, tables in HBM).  Explicit
// ZSTD_c_hashLog / ZSTD_c_chainLog (the "level word", ze_params_of) are honoured at level 3 — 14 / 13 are the sizes the LDS-resident
// finders of small batches are built for (byte-identical to ZstdCompressCtx.setHashLog(14).setChainLog(13); SURVEY.md Appendix B.2:
// +0.44 % size on Silesia xml).  With a dictionary (zj_cdict.h) frames are byte-identical to ZSTD_CCtx_refCDict + ZSTD_compress2.
//
// LDS (fused small-batch kernel): the match-finder hash tables (position+1, u16 for buffers <= 64 KiB else u32) own the
// LDS during match finding; the entropy stage (histograms, Huffman tree, tANS tables) overlays the same bytes
// afterwards.  HBM scratch per workgroup: literals, sequence records, block body.  Large batches find the sequences
// beforehand, lane per frame (zj_match_lane.h), and this file's ze_compress_t runs the entropy stage on them (`pre`);
// frames <= 4 KiB are then staged, gathered and assembled in LDS when the launch provides the room.
#pragma once
#include "zj_common.h"
#include "zj_invprob.h"


#define ZE_BLOCK_MAX (1u << 17)
#define ZE_MAX_SEQ ((ZE_BLOCK_MAX / 4u) + 16u)
#define ZE_L3_HASHLOG 14u
#define ZE_L3_CHAINLOG 13u

struct ZESeq { u32 ll; u32 ml; u32 off; u32 pos; };   // ll|llCode<<24, ml(matchLength)|mlCode<<24, offBase|ofCode<<24, literal start
#define ZE_LOW24(x) ((x) & 0xFFFFFFu)

// HBM scratch per workgroup
#define ZE_WS_LIT 0u
#define ZE_WS_SEQ (ZE_BLOCK_MAX + 64u)
#define ZE_WS_BODY (ZE_WS_SEQ + ZE_MAX_SEQ * 16u)
#define ZE_SCRATCH_BYTES (ZE_WS_BODY + ZE_BLOCK_MAX + 2048u)

struct ZENode { u32 count; u16 parent; u8 byte; u8 nbBits; };
struct ZEFseCT { u16 state[512]; i32 deltaFind[64]; u32 deltaNbBits[64]; u32 tableLog; };

struct ZEEntropy {                 // entropy-stage view of the LDS
    u32 hist[4][256];              // per-stream literal histograms (exact stream sizes before encoding)
    u32 count[256];
    ZENode node[516];
    u16 rankBase[192]; u16 rankCurr[192];
    u8 nbBits[256]; u16 val[256]; u8 weight[256];
    u32 scount[64]; short norm[64];
    u16 cumul[260]; u8 tableSymbol[512];
    ZEFseCT ct[3];                 // LL, OF, ML
    i32 qsLow[64]; i32 qsHigh[64]; // explicit quicksort stack
    ZESeq stage[64];               // sequence records staged for the tANS encoder
};

#define ZE_LDS_TABLE_L1_U16 (8192u * 2u)
#define ZE_LDS_BYTES(tableBytes) ((tableBytes) > sizeof(ZEEntropy) ? (tableBytes) : sizeof(ZEEntropy))

struct ZEncShared {                // uniforms, outside the overlay
    u32 err;
    u32 nbSeq, litSize, lastLL;
    u32 windowLog, hashLog, chainLog, minMatch, strategy, searchLog;
    u32 hdrSize, bodySize, litSecSize;
    u32 hufLog, hufMaxSV, hufHdr, litMode, litStreams;
    u32 strBytes[4], strOff[4];
    u32 seqType[3], seqHdr[3], seqLastCount;
    u32 tstate[3];                 // tANS encoder states LL, OF, ML carried across 64-sequence batches
    u32 tmp[8];
    u32 tight, tightHuf;           // the reference would have run out of destination somewhere on the way to this block's bytes (ze_compress_t "tight destinations")
    u32 edge[6];                   // LL/OF/ML codes of the first and of the last sequence
    // dictionary state kept across the frames one workgroup encodes (the kernel clears dictLoaded / ctDict once)
    u32 dictLoaded, dictID, dictStrategy, dictMinMatch, dictHufRep, dictHufMaxSV, dictFseRep[3];
    u32 ctDict[3];                 // e.ct[t] currently holds the dictionary's table
    u32 dictCodes[256];            // Huffman code | nbBits << 16 of the dictionary's literal table — in a multi-block frame: the previous block's
    u32 blkRep[2], blkNextRep[2];  // multi-block frames: repcodes confirmed by the last compressed block / left by the block being encoded
};
// one block of a multi-block frame (ze_compress_multi): block = frameBase[start, start + srcSize) of ze_compress_t
struct ZEBlockArgs { const u8* frameBase; u32 frameSize, start, isFirst, lastBlock; u32* tables; u32 serialParse; u32 paramSize; };   // paramSize: the size the compression parameters are chosen for when it is not the frame's (a stream: unknown size, the level's default row); 0 = frameSize   // serialParse: low bits 0 the wave matchers, 1 the one-lane parses (ZE_FLAG_MULTI_SERIAL), 2 the wave matchers without staged spans (ZE_FLAG_MULTI_NOCARRY); bit 2: levels 1-2 on the one-lane parse (ZE_FLAG_MULTI_FAST_SERIAL)
#define ZE_SMALL_MAX 4096u         /* frames up to this size are staged, gathered and assembled in LDS when the launch provides it */
#define ZE_ALIGN16(x) (((x) + 15u) & ~15u)
#define ZE_ENTROPY_LDS ZE_ALIGN16((u32)sizeof(ZEEntropy))
#define ZE_SMALL_LDS_BYTES (ZE_ENTROPY_LDS + 2u * ZE_SMALL_MAX + 1024u + 128u)

// ------------------------------------------------------------------ constants ---------------
#if ZJ_ON_GPU
#define ZE_CONST static __device__ const
#else
#define ZE_CONST static const
#endif
ZE_CONST u8 ze_k_ll_bits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
ZE_CONST u8 ze_k_ml_bits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
ZE_CONST short ze_k_ll_defnorm[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
ZE_CONST short ze_k_ml_defnorm[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
ZE_CONST short ze_k_of_defnorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };
ZE_CONST u8 ze_k_ll_code[64] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,22,22,22,22,22,22,22,22,
                                 23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24 };
ZE_CONST u8 ze_k_ml_code[128] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,
    32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,
    40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,
    42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42 };
ZE_CONST u32 ze_k_rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };

// ZSTD_LLcode / ZSTD_MLcode (v = ml - 3) and LL_bits / ML_bits in closed form: the tables above as arithmetic, so a lane
// does not wait for a constant-memory load per sequence (checked against the tables for every input:
// tests/test_emu_encode.py::test_code_tables_closed_form)
ZJ_HD u32 ze_ll_code(u32 v) {
    if (v < 16) return v;
    if (v < 24) return 16 + ((v - 16) >> 1);
    if (v < 32) return 20 + ((v - 24) >> 2);
    if (v < 48) return 22 + ((v - 32) >> 3);
    if (v < 64) return 24;
    return zj_hibit(v) + 19;
}
ZJ_HD u32 ze_ml_code(u32 v) {
    if (v < 32) return v;
    if (v < 40) return 32 + ((v - 32) >> 1);
    if (v < 48) return 36 + ((v - 40) >> 2);
    if (v < 64) return 38 + ((v - 48) >> 3);
    if (v < 96) return 40 + ((v - 64) >> 4);
    if (v < 128) return 42;
    return zj_hibit(v) + 36;
}
ZJ_HD u32 ze_ll_bits_of(u32 c) { return c < 16 ? 0u : (c >= 25 ? c - 19u : (u32)((0x433221111ull >> (4u * (c - 16u))) & 0xFu)); }
ZJ_HD u32 ze_ml_bits_of(u32 c) { return c < 32 ? 0u : (c >= 43 ? c - 36u : (u32)((0x54433221111ull >> (4u * (c - 32u))) & 0xFu)); }

// ------------------------------------------------------------------ parameters --------------
// N/compress/clevels.h:81-83,107-109 + ZSTD_adjustCParams_internal (N/compress/zstd_compress.c:1553-1572)
ZJ_HD void ze_adjust(u32& windowLog, u32& chainLog, u32& hashLog, u32 srcSize) {
    u32 const srcLog = (srcSize < 64u) ? 6u : zj_hibit(srcSize - 1) + 1;
    if (windowLog > srcLog) windowLog = srcLog;
    if (hashLog > windowLog + 1) hashLog = windowLog + 1;
    if (chainLog > windowLog) chainLog = windowLog;
    if (windowLog < 10) windowLog = 10;
}
struct ZEParams { u32 windowLog, chainLog, hashLog, minMatch, strategy, searchLog; };      // (window logs > 14 at strategies 3-5: the row-based finder, ze_params_uses_rows)      // strategy (ZSTD_strategy): 1 fast, 2 double-fast, 3 greedy, 4 lazy, 5 lazy2 (3-5 on the hash chain), 0 = not served
// "level" arguments are level words: the level in the low byte, then ZstdCompressCtx.setHashLog / setChainLog
// (ZSTD_c_hashLog / ZSTD_c_chainLog, 0 = not set) — honoured for the double-fast strategy, whose tables live in HBM on
// the lane-per-frame path and can therefore have the level's own sizes (16 / 15 at level 3) or any other.
#define ZE_LW(level, hashLog, chainLog) ((u32)(level) | ((u32)(hashLog) << 8) | ((u32)(chainLog) << 16))
#define ZE_LW_LEVEL(lw) ((lw) & 0xFFu)
#define ZE_LW_HL(lw) (((lw) >> 8) & 0xFFu)
#define ZE_LW_CL(lw) (((lw) >> 16) & 0xFFu)
#define ZE_LW_PERIOD(lw) (((lw) >> 24) & 0xFu)    /* match kernel only: rotation period of the double-fast lane machine, 0 = default */
#define ZE_LW_WAVE_ROUTE (1u << 29)              /* classification only: every single-block frame of the call goes to the wave-per-frame kernel (level 3, batches too small to fill the lane pipeline) */
#define ZE_LW_IMPLICIT (1u << 28)                /* hashLog / chainLog in the word are the level's OWN (16 / 15 at level 3: what a caller who set nothing gets) — multi-block frames, whose blocks take the level's parameters of their size, accept such a word */
// "tuned": table sizes the LDS-resident finders (fused kernel, wave-per-frame matcher) cannot hold — everything but the LDS-sized pair itself
#define ZE_LW_TUNED(lw) ((ZE_LW_HL(lw) | ZE_LW_CL(lw)) != 0u && !(ZE_LW_HL(lw) == ZE_L3_HASHLOG && ZE_LW_CL(lw) == ZE_L3_CHAINLOG))
#define ZE_HASHLOG_CAP 17u       /* 15 tag bits below the index must fit the product's high dword */
#define ZE_CHAINLOG_CAP 16u
ZJ_HD ZEParams ze_params_of(u32 levelWord, u32 srcSize) {
    u32 const level = ZE_LW_LEVEL(levelWord), hl = ZE_LW_HL(levelWord), cl = ZE_LW_CL(levelWord);
    u32 w, c, h, mm, st;
    if (level >= 5 && level <= 8) {
        // clevels.h:111-114, inputs <= 16 KiB: lazy (level 5) and lazy2 (levels 6-8) on the hash chain — the window log stays <= 14, so the
        // reference does not switch to its row-based finder; larger inputs at these levels do, and are not served (strategy 0)
        ZEParams q; q.windowLog = q.chainLog = q.hashLog = q.minMatch = q.strategy = q.searchLog = 0;
        if (srcSize > (128u << 10)) return q;
        if (srcSize <= (16u << 10)) {
            w = 14; c = 14; h = 14;
            q.strategy = level == 5 ? 4u : 5u;
            q.searchLog = level == 5 ? 3u : (level == 6 ? 4u : (level == 7 ? 6u : 8u));
        } else {                                              // clevels.h:85-88: greedy / lazy / lazy2 / lazy2 — on the row-based finder, the window log being > 14
            w = 17; c = 16; h = 17;
            q.strategy = level == 5 ? 3u : (level == 6 ? 4u : 5u);
            q.searchLog = level == 8 ? 4u : 3u;
        }
        ze_adjust(w, c, h, srcSize);
        q.windowLog = w; q.chainLog = c; q.hashLog = h; q.minMatch = 4;
        return q;
    }
    if (level == 4) {
        // clevels.h:84 / :110.  <= 16 KiB: greedy on the hash chain (the window log stays <= 14, where the reference does not switch to its
        // row-based match finder, zstd_compress.c:238-245); <= 128 KiB: double-fast with 2^17-entry tables; beyond: greedy on the row
        // finder — not served (strategy 0)
        ZEParams q; q.searchLog = 0;
        if (srcSize <= (16u << 10)) { w = 14; c = 14; h = 14; mm = 4; st = 3; q.searchLog = 4; }
        else if (srcSize <= (128u << 10)) { w = 17; c = 17; h = 17; mm = 4; st = 2; q.searchLog = 2; }
        else { q.windowLog = q.chainLog = q.hashLog = q.minMatch = q.strategy = 0; return q; }
        ze_adjust(w, c, h, srcSize);
        q.windowLog = w; q.chainLog = c; q.hashLog = h; q.minMatch = mm; q.strategy = st;
        return q;
    }
    if (srcSize <= (16u << 10)) { w = 14; c = 14; h = 15; mm = (level == 1) ? 5 : 4; st = (level == 3) ? 2 : 1; }
    else if (srcSize <= (128u << 10)) {
        if (level == 1) { w = 17; c = 12; h = 13; mm = 6; st = 1; }
        else if (level == 2) { w = 17; c = 13; h = 15; mm = 5; st = 1; }
        else { w = 17; c = 15; h = 16; mm = 5; st = 2; }
    } else if (srcSize <= (256u << 10)) {                     // clevels.h:52-55 (multi-block frames, ze_compress_multi)
        if (level == 1) { w = 18; c = 13; h = 14; mm = 6; st = 1; }
        else if (level == 2) { w = 18; c = 14; h = 14; mm = 5; st = 2; }
        else { w = 18; c = 16; h = 16; mm = 4; st = 2; }
    } else {                                                  // clevels.h:27-30
        if (level == 1) { w = 19; c = 13; h = 14; mm = 7; st = 1; }
        else if (level == 2) { w = 20; c = 15; h = 16; mm = 6; st = 1; }
        else { w = 21; c = 16; h = 17; mm = 5; st = 2; }
    }
    ze_adjust(w, c, h, srcSize);
    if (srcSize > (128u << 10)) {                             // multi-block frames run the level's own table sizes (tables in HBM)
        ZEParams q; q.windowLog = w; q.chainLog = c; q.hashLog = h; q.minMatch = mm; q.strategy = st; q.searchLog = 1;
        return q;
    }
    if (st == 2 && (hl | cl)) {   // explicit ZSTD_c_hashLog / ZSTD_c_chainLog: override, then ZSTD_adjustCParams_internal again (zstd_compress.c:1640-1655)
        if (hl) h = hl;
        if (cl) c = cl;
        ze_adjust(w, c, h, srcSize);
    } else if (st == 2) {     // LDS budget: ZSTD_c_hashLog = 14, ZSTD_c_chainLog = 13, then adjust again
        if (h > ZE_L3_HASHLOG || c > ZE_L3_CHAINLOG) { if (h > ZE_L3_HASHLOG) h = ZE_L3_HASHLOG; if (c > ZE_L3_CHAINLOG) c = ZE_L3_CHAINLOG; ze_adjust(w, c, h, srcSize); }
    }
    ZEParams p; p.windowLog = w; p.chainLog = c; p.hashLog = h; p.minMatch = mm; p.strategy = st; p.searchLog = 1;
    return p;
}
// ZSTD_resolveRowMatchFinderMode (zstd_compress.c:238-245): greedy / lazy / lazy2 search rows of tagged entries instead of the hash
// chain once the window log exceeds 14
ZJ_HD bool ze_params_uses_rows(const ZEParams& p) { return p.strategy >= 3u && p.strategy <= 5u && p.windowLog > 14u; }
ZJ_DEV void ze_params(ZEncShared& sh, u32 level, u32 srcSize) {
    ZEParams const p = ze_params_of(level, srcSize);
    sh.windowLog = p.windowLog; sh.chainLog = p.chainLog; sh.hashLog = p.hashLog; sh.minMatch = p.minMatch; sh.strategy = p.strategy; sh.searchLog = p.searchLog;
}
// bytes of LDS the match-finder tables need for this (level, size, index width)
ZJ_HD u32 ze_table_entries(u32 level, u32 maxSrc) {
    // upper bound over all sizes <= maxSrc: level 1: 2^13 (<=16 KiB inputs use 2^min(15, wlog+1) <= 2^15)
    if (level == 3) return (1u << ZE_L3_HASHLOG) + (1u << ZE_L3_CHAINLOG);
    if (level == 2) return 1u << 15;
    return maxSrc <= (16u << 10) ? (1u << 15) : (1u << 15);   // level 1 small inputs: hashLog 15 -> clamp wlog+1
}

// ------------------------------------------------------------------ match finders (lane 0) --
ZJ_DEV u32 ze_hash(const u8* p, u32 hBits, u32 mls) {   // N/compress/zstd_compress_internal.h:898-960
    switch (mls) {
    default:
    case 4: return (ld32(p) * 2654435761U) >> (32 - hBits);
    case 5: return (u32)(((ld64(p) << 24) * 889523592379ULL) >> (64 - hBits));
    case 6: return (u32)(((ld64(p) << 16) * 227718039650203ULL) >> (64 - hBits));
    case 7: return (u32)(((ld64(p) << 8) * 58295818150454627ULL) >> (64 - hBits));
    case 8: return (u32)((ld64(p) * 0xCF1BBCDCB7A56463ULL) >> (64 - hBits));
    }
}
ZJ_DEV u32 ze_count(const u8* in, const u8* match, const u8* inLimit) {   // ZSTD_count (result only)
    const u8* const start = in;
    while (in + 8 <= inLimit) {
        u64 const d = ld64(in) ^ ld64(match);
        if (d) return (u32)(in - start) + ((u32)__builtin_ctzll(d) >> 3);
        in += 8; match += 8;
    }
    while (in < inLimit && *in == *match) { in++; match++; }
    return (u32)(in - start);
}

struct ZEOut { ZESeq* seqs; u32* litOff; u32 n; u32 lit; };
ZJ_DEV void ze_store(ZEOut& o, u32 litPos, u32 ll, u32 offBase, u32 ml) {
    ZESeq s; s.ll = ll; s.ml = ml; s.off = offBase; s.pos = litPos;
    o.litOff[o.n] = o.lit;                      // where this sequence's literals land in the literal buffer
    o.seqs[o.n++] = s; o.lit += ll;
}

#ifndef ZE_COUNT_ITER
#define ZE_COUNT_ITER() ((void)0)     /* instrumentation hook for tools; no-op in the product */
#endif
// hash of a position from its 8 already-loaded bytes (N/compress/zstd_compress_internal.h:898-960)
ZJ_DEV u32 ze_hash_w(u64 w, u32 hBits, u32 mls) {
    switch (mls) {
    default:
    case 4: return ((u32)w * 2654435761U) >> (32 - hBits);
    case 5: return (u32)(((w << 24) * 889523592379ULL) >> (64 - hBits));
    case 6: return (u32)(((w << 16) * 227718039650203ULL) >> (64 - hBits));
    case 7: return (u32)(((w << 8) * 58295818150454627ULL) >> (64 - hBits));
    case 8: return (u32)((w * 0xCF1BBCDCB7A56463ULL) >> (64 - hBits));
    }
}

// Table entry codecs.  Entries hold position+1 (0 = empty).  The HBM-resident tables of the
// lane-per-frame path additionally carry 15 tag bits computed from exactly the bytes the reference
// compares at that position (4 for the short/fast probe, 8 for the long probe): a tag mismatch proves the
// byte comparison would fail, so the candidate's bytes are not fetched — same decisions, fewer HBM sectors.
struct ZEEnt16 { typedef u16 T; ZJ_DEVM T make(u32 pos1, u32) { return (T)pos1; } ZJ_DEVM u32 pos(u32 e) { return e; } ZJ_DEVM bool maybe(u32 e, u32) { return e != 0; } };
struct ZEEnt32 { typedef u32 T; ZJ_DEVM T make(u32 pos1, u32) { return pos1; } ZJ_DEVM u32 pos(u32 e) { return e; } ZJ_DEVM bool maybe(u32 e, u32) { return e != 0; } };
struct ZEEntTag { typedef u32 T; ZJ_DEVM T make(u32 pos1, u32 tag) { return pos1 | (tag << 17); } ZJ_DEVM u32 pos(u32 e) { return e & 0x1FFFFu; }
                  ZJ_DEVM bool maybe(u32 e, u32 tag) { return (e & 0x1FFFFu) != 0 && (e >> 17) == tag; } };
ZJ_DEV u32 ze_tag4(u32 v) { return (v * 2246822519u) >> 17; }                                   // 15 bits from 4 bytes
ZJ_DEV u32 ze_tag8(u64 w) { return (u32)((w * 0x9E3779B97F4A7C15ull) >> 49); }                    // 15 bits from 8 bytes

// ZSTD_compressBlock_fast_noDict_generic (N/compress/zstd_fast.c:192-423); tables hold position+1.
// Same decisions in the same order as the reference's pipelined loop; what changes is WHEN bytes are
// fetched: every global load an iteration can need (the two new positions, the repcode candidate and
// both table candidates) is issued at the top of the iteration, so one HBM/L2 round trip covers two
// positions instead of five, and the 8 bytes of a position are loaded once and carried in registers as
// it moves from ip2/ip3 to ip0/ip1.  The second table write of an iteration (table[hash1] = ip1) happens
// on every path of the reference, so it is done up front, right after table[hash1] has been read.
template <class E>
ZJ_DEV u32 ze_block_fast(ZEOut& o, const u8* src, u32 srcSize, u32 hlog, u32 mls, typename E::T* table) {
    const u8* const istart = src; const u8* const iend = src + srcSize; const u8* const ilimit = iend - 8;
    const u8* anchor = istart; const u8* ip0 = istart + 1; const u8* ip1; const u8* ip2; const u8* ip3;
    u32 rep1 = 1, rep2 = 0;                       // rep {1,4}: 4 > maxRep == 1 at the first position of a frame
    u32 hash0, hash1, matchE, cur0 = 0, offcode = 0, mLength = 0, step;
    const u8* match0 = istart; const u8* nextStep;
    for (;;) {
        step = 2; nextStep = ip0 + 128;
        ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
        if (ip3 >= ilimit) break;
        u64 w0 = ld64(ip0), w1 = ld64(ip1);
        hash0 = ze_hash_w(w0, hlog, mls); hash1 = ze_hash_w(w1, hlog, mls);
        matchE = table[hash0];
        bool found = false, isRep = false;
        do {
            ZE_COUNT_ITER();
            // ---- issue everything this iteration may read ----
            u32 const t0 = ze_tag4((u32)w0), t1 = ze_tag4((u32)w1);
            bool const m0 = E::maybe(matchE, t0);
            u32 const rval = ld32(ip2 - rep1);
            u64 const w2 = ld64(ip2), w3 = ld64(ip3);
            u32 const c0 = m0 ? ld32(istart + E::pos(matchE) - 1) : ~(u32)w0;
            cur0 = (u32)(ip0 - istart); table[hash0] = E::make(cur0 + 1, t0);
            u32 const matchE1 = table[hash1];
            table[hash1] = E::make((u32)(ip1 - istart) + 1, t1);     // written on every path (see header comment)
            bool const m1 = E::maybe(matchE1, t1);
            u32 const c1 = m1 ? ld32(istart + E::pos(matchE1) - 1) : ~(u32)w1;
            // ---- decisions, reference order ----
            if (((u32)w2 == rval) & (rep1 > 0)) {
                ip0 = ip2; match0 = ip0 - rep1;
                mLength = (ip0[-1] == match0[-1]); ip0 -= mLength; match0 -= mLength;
                offcode = 1; mLength += 4;
                found = true; isRep = true; break;
            }
            if (m0 && c0 == (u32)w0) { found = true; break; }
            matchE = matchE1;
            hash0 = hash1; hash1 = ze_hash_w(w2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            cur0 = (u32)(ip0 - istart);
            if (m1 && c1 == (u32)w1) { if (step <= 4) table[hash1] = E::make((u32)(ip1 - istart) + 1, ze_tag4((u32)w2)); found = true; break; }
            matchE = table[hash1];
            hash0 = hash1; hash1 = ze_hash_w(w3, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
            w0 = w2; w1 = w3;
            if (ip2 >= nextStep) { step++; nextStep += 128; }
        } while (ip3 < ilimit);
        if (!found) break;
        if (!isRep) {
            match0 = istart + E::pos(matchE) - 1;
            rep2 = rep1; rep1 = (u32)(ip0 - match0); offcode = rep1 + 3; mLength = 4;
            while (((ip0 > anchor) & (match0 > istart)) && (ip0[-1] == match0[-1])) { ip0--; match0--; mLength++; }
        }
        mLength += ze_count(ip0 + mLength, match0 + mLength, iend);
        ze_store(o, (u32)(anchor - istart), (u32)(ip0 - anchor), offcode, mLength);
        ip0 += mLength; anchor = ip0;
        if (ip0 <= ilimit) {
            {   u64 const wa = ld64(istart + cur0 + 2), wb = ld64(ip0 - 2);
                table[ze_hash_w(wa, hlog, mls)] = E::make(cur0 + 2 + 1, ze_tag4((u32)wa));
                table[ze_hash_w(wb, hlog, mls)] = E::make((u32)(ip0 - 2 - istart) + 1, ze_tag4((u32)wb)); }
            if (rep2 > 0) {
                while ((ip0 <= ilimit) && (ld32(ip0) == ld32(ip0 - rep2))) {
                    u32 const rLength = ze_count(ip0 + 4, ip0 + 4 - rep2, iend) + 4;
                    { u32 const t = rep2; rep2 = rep1; rep1 = t; }
                    {   u64 const wi = ld64(ip0); table[ze_hash_w(wi, hlog, mls)] = E::make((u32)(ip0 - istart) + 1, ze_tag4((u32)wi)); }
                    ip0 += rLength;
                    ze_store(o, (u32)(anchor - istart), 0, 1, rLength);
                    anchor = ip0;
                }
            }
        }
    }
    return (u32)(iend - anchor);
}

// ZSTD_compressBlock_doubleFast_noDict_generic (N/compress/zstd_double_fast.c:105-323), loads hoisted the
// same way: the bytes of ip1, the repcode candidate and both table candidates are requested together.
template <class E>
ZJ_DEV u32 ze_block_dfast(ZEOut& o, const u8* src, u32 srcSize, u32 hBitsL, u32 hBitsS, u32 mls, typename E::T* hashLong, typename E::T* hashSmall) {
    const u8* const istart = src; const u8* const iend = src + srcSize; const u8* const ilimit = iend - 8;
    const u8* anchor = istart; const u8* ip = istart + 1; const u8* ip1;
    u32 off1 = 1, off2 = 0;                       // rep {1,4}: 4 > maxRep == 1 at frame start
    u32 mLength = 0, offset = 0, curr = 0, step, hl0, hl1 = 0, el0, el1 = 0;
    const u8* nextStep; const u8* matchs0 = istart; const u8* matchl0;
    for (;;) {
        step = 1; nextStep = ip + 256; ip1 = ip + step;
        if (ip1 > ilimit) break;
        // software pipeline: the table entries and input bytes of position p+1 are requested while
        // position p is being decided, so a no-match step costs one memory round trip, not two.
        // Reads for p+1 are issued after p's table writes, exactly where the reference reads them.
        u64 w = ld64(ip), w1 = ld64(ip1);
        hl0 = ze_hash_w(w, hBitsL, 8); u32 hs0 = ze_hash_w(w, hBitsS, mls);
        el0 = hashLong[hl0]; u32 es0 = hashSmall[hs0];
        u32 kind = 0;     // 0 none, 1 repcode stored, 2 long match found, 3 short match -> search next long
        do {
            ZE_COUNT_ITER();
            curr = (u32)(ip - istart);
            u32 const tl = ze_tag8(w), ts = ze_tag4((u32)w);
            hashLong[hl0] = E::make(curr + 1, tl); hashSmall[hs0] = E::make(curr + 1, ts);
            // ---- everything this position may read ----
            bool const ml0 = E::maybe(el0, tl), ms0 = E::maybe(es0, ts);
            u32 const rv = ld32(ip + 1 - off1);
            u64 const cl = ml0 ? ld64(istart + E::pos(el0) - 1) : ~w;
            u32 const cs = ms0 ? ld32(istart + E::pos(es0) - 1) : ~(u32)w;
            // ---- next position: hashes, table entries, and the input word after it ----
            hl1 = ze_hash_w(w1, hBitsL, 8); u32 const hs1 = ze_hash_w(w1, hBitsS, mls);
            el1 = hashLong[hl1]; u32 const es1 = hashSmall[hs1];
            u32 const stepN = step + ((ip1 >= nextStep) ? 1u : 0u);
            const u8* const ip2 = ip1 + stepN;
            u64 const w2 = (ip2 <= ilimit) ? ld64(ip2) : 0;
            // ---- decisions, reference order ----
            if ((off1 > 0) & (rv == (u32)(w >> 8))) {
                mLength = ze_count(ip + 1 + 4, ip + 1 + 4 - off1, iend) + 4;
                ip++;
                ze_store(o, (u32)(anchor - istart), (u32)(ip - anchor), 1, mLength);
                kind = 1; break;
            }
            if (ml0 && cl == w) {
                matchl0 = istart + E::pos(el0) - 1;
                mLength = ze_count(ip + 8, matchl0 + 8, iend) + 8;
                offset = (u32)(ip - matchl0);
                while (((ip > anchor) & (matchl0 > istart)) && (ip[-1] == matchl0[-1])) { ip--; matchl0--; mLength++; }
                kind = 2; break;
            }
            if (ms0 && cs == (u32)w) { matchs0 = istart + E::pos(es0) - 1; kind = 3; break; }
            if (ip1 >= nextStep) { step++; nextStep += 256; }
            ip = ip1; ip1 = ip2;
            hl0 = hl1; hs0 = hs1; el0 = el1; es0 = es1; w = w1; w1 = w2;
        } while (ip1 <= ilimit);
        if (kind == 0) break;
        if (kind == 3) {
            mLength = ze_count(ip + 4, matchs0 + 4, iend) + 4;
            offset = (u32)(ip - matchs0);
            if ((E::pos(el1) > 1) && E::maybe(el1, ze_tag8(w1)) && (ld64(istart + E::pos(el1) - 1) == w1)) {
                const u8* const matchl1 = istart + E::pos(el1) - 1;
                u32 const l1len = ze_count(ip1 + 8, matchl1 + 8, iend) + 8;
                if (l1len > mLength) { ip = ip1; mLength = l1len; offset = (u32)(ip - matchl1); matchs0 = matchl1; }
            }
            while (((ip > anchor) & (matchs0 > istart)) && (ip[-1] == matchs0[-1])) { ip--; matchs0--; mLength++; }
        }
        if (kind >= 2) {
            off2 = off1; off1 = offset;
            if (step < 4) hashLong[hl1] = E::make((u32)(ip1 - istart) + 1, ze_tag8(w1));
            ze_store(o, (u32)(anchor - istart), (u32)(ip - anchor), offset + 3, mLength);
        }
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            {   u32 const ins = curr + 2;
                u64 const wa = ld64(istart + ins), wb = ld64(ip - 2), wc = ld64(ip - 1);
                hashLong[ze_hash_w(wa, hBitsL, 8)] = E::make(ins + 1, ze_tag8(wa));
                hashLong[ze_hash_w(wb, hBitsL, 8)] = E::make((u32)(ip - 2 - istart) + 1, ze_tag8(wb));
                hashSmall[ze_hash_w(wa, hBitsS, mls)] = E::make(ins + 1, ze_tag4((u32)wa));
                hashSmall[ze_hash_w(wc, hBitsS, mls)] = E::make((u32)(ip - 1 - istart) + 1, ze_tag4((u32)wc));
            }
            while ((ip <= ilimit) && ((off2 > 0) & (ld32(ip) == ld32(ip - off2)))) {
                u32 const rLength = ze_count(ip + 4, ip + 4 - off2, iend) + 4;
                u32 const t = off2; off2 = off1; off1 = t;
                {   u64 const wi = ld64(ip);
                    hashSmall[ze_hash_w(wi, hBitsS, mls)] = E::make((u32)(ip - istart) + 1, ze_tag4((u32)wi));
                    hashLong[ze_hash_w(wi, hBitsL, 8)] = E::make((u32)(ip - istart) + 1, ze_tag8(wi)); }
                ze_store(o, (u32)(anchor - istart), 0, 1, rLength);
                ip += rLength; anchor = ip;
            }
        }
    }
    return (u32)(iend - anchor);
}

// ---- greedy parse on the hash chain: ZSTD_compressBlock_greedy = ZSTD_compressBlock_lazy_generic(depth 0, search_hashChain, noDict)
//      (N/compress/zstd_lazy.c:1516-1780) over ZSTD_HcFindBestMatch (:667-723) and ZSTD_insertAndFindFirstIndex_internal (:632-657).
// Table indices are the reference's: a fresh context's window starts at index 2 (ZSTD_WINDOW_START_INDEX), so position p has index
// p + 2, entry 0 means "empty" and the chain table is addressed by (index & chainMask).  `nextToUpdate` and the lazy-skipping
// flag (one insert per search while the parse strides over incompressible data) are the match state's, kept here.
struct ZEChain {
    const u8* base;                 // src - 2
    u32* hashTable; u32* chainTable; u32 hashLog, chainMask, chainSize, nbAttemptsMax, mls, nextToUpdate, lazySkipping;
};
ZJ_DEV u32 ze_hc_insert_find(ZEChain& m, const u8* ip) {
    u32 const target = (u32)(ip - m.base);
    u32 idx = m.nextToUpdate;
    while (idx < target) {
        u32 const h = ze_hash(m.base + idx, m.hashLog, m.mls);
        m.chainTable[idx & m.chainMask] = m.hashTable[h];
        m.hashTable[h] = idx;
        idx++;
        if (m.lazySkipping) break;
    }
    m.nextToUpdate = target;
    return m.hashTable[ze_hash(ip, m.hashLog, m.mls)];
}
// returns the best length (3 = none), *offBase set when a match was taken
ZJ_DEV u32 ze_hc_find_best(ZEChain& m, const u8* ip, const u8* iLimit, u32* offBase) {
    u32 const curr = (u32)(ip - m.base);
    u32 const lowLimit = 2u;                                   // the frame fits its window, no dictionary: window.lowLimit
    u32 const minChain = curr > m.chainSize ? curr - m.chainSize : 0u;
    u32 nbAttempts = m.nbAttemptsMax, ml = 3u;
    u32 matchIndex = ze_hc_insert_find(m, ip);
    for (; (matchIndex >= lowLimit) & (nbAttempts > 0); nbAttempts--) {
        const u8* const match = m.base + matchIndex;
        u32 currentMl = 0;
        if (ld32(match + ml - 3) == ld32(ip + ml - 3)) currentMl = ze_count(ip, match, iLimit);     // potentially better
        if (currentMl > ml) {
            ml = currentMl; *offBase = (curr - matchIndex) + 3u;
            if (ip + currentMl == iLimit) break;               // best possible
        }
        if (matchIndex <= minChain) break;
        matchIndex = m.chainTable[matchIndex & m.chainMask];
    }
    return ml;
}
// ---- the row-based finder (ZSTD_RowFindBestMatch, N/compress/zstd_lazy.c:1141-1330, with ZSTD_row_update_internal :926-950 and
// ZSTD_row_nextIndex :796-801): the hash table is cut into rows of 16 / 32 / 64 entries (rowLog = searchLog clamped to 4..6), a byte
// table of the same shape holds an 8-bit tag per entry and, in each row's byte 0, the row's head; inserts walk the head downwards
// (skipping slot 0), a search compares the tag against the whole row, visits the hits from the newest on and stops at 2^min(searchLog,
// rowLog) candidates.  The reference's SIMD / SWAR mask and its 8-entry hash cache are ways to compute this faster: the cache always
// holds the plain hash of its position (asserted there), so it is not modelled; the salt it mixes into the hash permutes rows and
// tags without changing which positions share them, so it is 0 here.  hashTable: u32[1 << hashLog]; tags: u8[1 << hashLog] behind it.
struct ZERow {
    const u8* base; u32* hashTable; u8* tagTable; u32 rowHashLog, rowLog, rowMask, nbAttemptsMax, mls, nextToUpdate, lazySkipping;
};
ZJ_DEV u32 ze_row_next_index(u8* tagRow, u32 rowMask) {
    u32 next = ((u32)tagRow[0] - 1u) & rowMask;
    next += (next == 0) ? rowMask : 0;                     // skip slot 0 (the head itself)
    tagRow[0] = (u8)next;
    return next;
}
ZJ_DEV void ze_row_insert_range(ZERow& m, u32 from, u32 to) {
    for (u32 idx = from; idx < to; idx++) {
        u32 const hash = ze_hash(m.base + idx, m.rowHashLog + 8u, m.mls);
        u32 const relRow = (hash >> 8) << m.rowLog;
        u8* const tagRow = m.tagTable + relRow;
        u32 const pos = ze_row_next_index(tagRow, m.rowMask);
        tagRow[pos] = (u8)hash; m.hashTable[relRow + pos] = idx;
    }
}
ZJ_DEV void ze_row_update(ZERow& m, u32 target) {            // ZSTD_row_update_internal: after a long gap only its two ends are inserted
    u32 idx = m.nextToUpdate;
    if (target - idx > 384u) { ze_row_insert_range(m, idx, idx + 96u); idx = target - 32u; }
    ze_row_insert_range(m, idx, target);
    m.nextToUpdate = target;
}
ZJ_DEV u32 ze_row_find_best(ZERow& m, const u8* ip, const u8* iLimit, u32* offBase) {
    u32 const curr = (u32)(ip - m.base);
    u32 const lowLimit = 2u;                                   // the frame fits its window, no dictionary
    u32 nbAttempts = m.nbAttemptsMax, ml = 3u;
    if (!m.lazySkipping) ze_row_update(m, curr); else m.nextToUpdate = curr;
    u32 const hash = ze_hash(ip, m.rowHashLog + 8u, m.mls);
    u32 const relRow = (hash >> 8) << m.rowLog, tag = hash & 0xFFu;
    u32* const row = m.hashTable + relRow; u8* const tagRow = m.tagTable + relRow;
    u32 const head = (u32)tagRow[0] & m.rowMask, rowEntries = m.rowMask + 1u;
    u32 matchBuffer[64]; u32 numMatches = 0;
    for (u32 j = 0; j < rowEntries && nbAttempts > 0; j++) {   // the rotated mask, newest entry first
        u32 const matchPos = (head + j) & m.rowMask;
        if (tagRow[matchPos] != (u8)tag) continue;
        if (matchPos == 0) continue;
        u32 const matchIndex = row[matchPos];
        if (matchIndex < lowLimit) break;
        matchBuffer[numMatches++] = matchIndex; --nbAttempts;
    }
    {   u32 const pos = ze_row_next_index(tagRow, m.rowMask);   // the current position goes in right away
        tagRow[pos] = (u8)tag; row[pos] = m.nextToUpdate++; }
    for (u32 k = 0; k < numMatches; k++) {
        u32 const matchIndex = matchBuffer[k];
        const u8* const match = m.base + matchIndex;
        u32 currentMl = 0;
        if (ld32(match + ml - 3) == ld32(ip + ml - 3)) currentMl = ze_count(ip, match, iLimit);
        if (currentMl > ml) {
            ml = currentMl; *offBase = (curr - matchIndex) + 3u;
            if (ip + currentMl == iLimit) break;
        }
    }
    return ml;
}
// search front end of the lazy parser: the hash chain (window log <= 14) or the rows
struct ZEFinder { bool rows; ZEChain c; ZERow r; };
ZJ_DEV u32 ze_find_best(ZEFinder& f, const u8* ip, const u8* iLimit, u32* offBase) { return f.rows ? ze_row_find_best(f.r, ip, iLimit, offBase) : ze_hc_find_best(f.c, ip, iLimit, offBase); }
ZJ_DEV void ze_finder_skip(ZEFinder& f, bool on) { if (f.rows) f.r.lazySkipping = on ? 1u : 0u; else f.c.lazySkipping = on ? 1u : 0u; }

// depth 0 = greedy, 1 = lazy, 2 = lazy2 (strategy - 3).  `second`: the chain table, or with rows the tag table (bytes)
ZJ_DEV u32 ze_block_lazy(ZEOut& o, const u8* src, u32 srcSize, const ZEParams& p, u32* hashTable, u32* second) {
    bool const rows = ze_params_uses_rows(p);
    u32* const chainTable = second;
    const u8* const istart = src; const u8* const iend = src + srcSize; const u8* const ilimit = iend - 8 - (rows ? 8 : 0);   // (ZSTD_ROW_HASH_CACHE_SIZE)
    const u8* ip = istart + 1; const u8* anchor = istart;      // ip += (dictAndPrefixLength == 0)
    u32 off1 = 1, off2 = 0;                                    // rep {1,4}: 4 > maxRep == 1 at frame start (saved; only matters to a next block)
    u32 const depth = p.strategy - 3u;
    ZEFinder m; m.rows = rows;
    u32 const mls = p.minMatch < 4u ? 4u : (p.minMatch > 6u ? 6u : p.minMatch);
    m.c.base = src - 2; m.c.hashTable = hashTable; m.c.chainTable = chainTable; m.c.hashLog = p.hashLog;
    m.c.chainSize = 1u << p.chainLog; m.c.chainMask = m.c.chainSize - 1u; m.c.nbAttemptsMax = 1u << p.searchLog;
    m.c.mls = mls; m.c.nextToUpdate = 2u; m.c.lazySkipping = 0;
    {   u32 const rowLog = p.searchLog < 4u ? 4u : (p.searchLog > 6u ? 6u : p.searchLog);
        m.r.base = src - 2; m.r.hashTable = hashTable; m.r.tagTable = (u8*)second; m.r.rowLog = rowLog; m.r.rowMask = (1u << rowLog) - 1u;
        m.r.rowHashLog = p.hashLog - rowLog; m.r.nbAttemptsMax = 1u << (p.searchLog < rowLog ? p.searchLog : rowLog);
        m.r.mls = mls; m.r.nextToUpdate = 2u; m.r.lazySkipping = 0; }
    while (ip < ilimit) {
        u32 matchLength = 0, offBase = 1u;                     // REPCODE1_TO_OFFBASE
        const u8* start = ip + 1;
        bool store = false;
        if ((off1 > 0) & (ld32(ip + 1 - off1) == ld32(ip + 1))) {                      // repcode at ip + 1: at depth 0 it is taken at once
            matchLength = ze_count(ip + 1 + 4, ip + 1 + 4 - off1, iend) + 4u;
            if (depth == 0) store = true;
        }
        if (!store) {
            {   u32 found = 999999999u;                                              // first search (depth 0)
                u32 const ml2 = ze_find_best(m, ip, iend, &found);
                if (ml2 > matchLength) { matchLength = ml2; start = ip; offBase = found; } }
            if (matchLength < 4u) {
                u32 const step = ((u32)(ip - anchor) >> 8) + 1u;                     // kSearchStrength
                ip += step;
                ze_finder_skip(m, step > 8u);                                          // kLazySkippingStep
                continue;
            }
            if (depth >= 1) while (ip < ilimit) {                                    // is the match that starts one (two) bytes later worth more?
                ip++;
                if ((offBase != 0) && ((off1 > 0) & (ld32(ip) == ld32(ip - off1)))) {
                    u32 const mlRep = ze_count(ip + 4, ip + 4 - off1, iend) + 4u;
                    i32 const gain2 = (i32)(mlRep * 3u), gain1 = (i32)(matchLength * 3u - zj_hibit(offBase) + 1u);
                    if ((mlRep >= 4u) && (gain2 > gain1)) { matchLength = mlRep; offBase = 1u; start = ip; }
                }
                {   u32 cand = 999999999u;
                    u32 const ml2 = ze_find_best(m, ip, iend, &cand);
                    i32 const gain2 = (i32)(ml2 * 4u - zj_hibit(cand)), gain1 = (i32)(matchLength * 4u - zj_hibit(offBase) + 4u);
                    if ((ml2 >= 4u) && (gain2 > gain1)) { matchLength = ml2; offBase = cand; start = ip; continue; } }
                if ((depth == 2) && (ip < ilimit)) {
                    ip++;
                    if ((offBase != 0) && ((off1 > 0) & (ld32(ip) == ld32(ip - off1)))) {
                        u32 const mlRep = ze_count(ip + 4, ip + 4 - off1, iend) + 4u;
                        i32 const gain2 = (i32)(mlRep * 4u), gain1 = (i32)(matchLength * 4u - zj_hibit(offBase) + 1u);
                        if ((mlRep >= 4u) && (gain2 > gain1)) { matchLength = mlRep; offBase = 1u; start = ip; }
                    }
                    {   u32 cand = 999999999u;
                        u32 const ml2 = ze_find_best(m, ip, iend, &cand);
                        i32 const gain2 = (i32)(ml2 * 4u - zj_hibit(cand)), gain1 = (i32)(matchLength * 4u - zj_hibit(offBase) + 7u);
                        if ((ml2 >= 4u) && (gain2 > gain1)) { matchLength = ml2; offBase = cand; start = ip; continue; } }
                }
                break;                                                               // nothing better: store the previous solution
            }
            if (offBase > 3u) {                                                      // a real offset: catch up
                u32 const off = offBase - 3u;
                while ((start > anchor) & (start - off > istart) && (start[-1] == (start - off)[-1])) { start--; matchLength++; }
                off2 = off1; off1 = off;
            }
        }
        ze_store(o, (u32)(anchor - istart), (u32)(start - anchor), offBase, matchLength);
        anchor = ip = start + matchLength;
        ze_finder_skip(m, false);
        while ((ip <= ilimit) & (off2 > 0) && (ld32(ip) == ld32(ip - off2))) {        // immediate repcode
            matchLength = ze_count(ip + 4, ip + 4 - off2, iend) + 4u;
            { u32 const t = off2; off2 = off1; off1 = t; }
            ze_store(o, (u32)(anchor - istart), 0u, 1u, matchLength);
            ip += matchLength; anchor = ip;
        }
    }
    return (u32)(iend - anchor);
}

// ---- the same two parses for one BLOCK of a multi-block frame (ze_compress_multi) ----
// The block is base[start, end); table entries and match positions are relative to `base` (the frame), so matches reach
// back into earlier blocks; literal positions in the records are relative to the block.  rep[] comes from the last block
// that was emitted compressed and is left as the reference leaves it for the next one (zstd_fast.c:244-250, :352-372;
// zstd_double_fast.c:153-163, :238-246: offsets beyond the data seen so far are parked and restored).  The frame fits its
// window (checked by the caller), so every earlier position is a legal candidate.
template <class E>
ZJ_DEV u32 ze_block_fast_x(ZEOut& o, const u8* base, u32 start, u32 end, u32 hlog, u32 mls, typename E::T* table, const u32* repIn, u32* repOut) {
    const u8* const istart = base + start; const u8* const iend = base + end; const u8* const ilimit = iend - 8;
    const u8* anchor = istart; const u8* ip0 = istart + (start == 0 ? 1 : 0); const u8* ip1; const u8* ip2; const u8* ip3;
    u32 rep1 = repIn[0], rep2 = repIn[1], saved1 = 0, saved2 = 0;
    {   u32 const maxRep = (u32)(ip0 - base);
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; } }
    u32 hash0, hash1, matchE, cur0 = 0, offcode = 0, mLength = 0, step;
    const u8* match0 = base; const u8* nextStep;
    for (;;) {
        step = 2; nextStep = ip0 + 128;
        ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
        if (ip3 >= ilimit) break;
        u64 w0 = ld64(ip0), w1 = ld64(ip1);
        hash0 = ze_hash_w(w0, hlog, mls); hash1 = ze_hash_w(w1, hlog, mls);
        matchE = table[hash0];
        bool found = false, isRep = false;
        do {
            u32 const t0 = ze_tag4((u32)w0), t1 = ze_tag4((u32)w1);
            bool const m0 = E::maybe(matchE, t0);
            u32 const rval = ld32(ip2 - rep1);
            u64 const w2 = ld64(ip2), w3 = ld64(ip3);
            u32 const c0 = m0 ? ld32(base + E::pos(matchE) - 1) : ~(u32)w0;
            cur0 = (u32)(ip0 - base); table[hash0] = E::make(cur0 + 1, t0);
            u32 const matchE1 = table[hash1];
            table[hash1] = E::make((u32)(ip1 - base) + 1, t1);
            bool const m1 = E::maybe(matchE1, t1);
            u32 const c1 = m1 ? ld32(base + E::pos(matchE1) - 1) : ~(u32)w1;
            if (((u32)w2 == rval) & (rep1 > 0)) {
                ip0 = ip2; match0 = ip0 - rep1;
                mLength = (ip0[-1] == match0[-1]); ip0 -= mLength; match0 -= mLength;
                offcode = 1; mLength += 4;
                found = true; isRep = true; break;
            }
            if (m0 && c0 == (u32)w0) { found = true; break; }
            matchE = matchE1;
            hash0 = hash1; hash1 = ze_hash_w(w2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            cur0 = (u32)(ip0 - base);
            if (m1 && c1 == (u32)w1) { if (step <= 4) table[hash1] = E::make((u32)(ip1 - base) + 1, ze_tag4((u32)w2)); found = true; break; }
            matchE = table[hash1];
            hash0 = hash1; hash1 = ze_hash_w(w3, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
            w0 = w2; w1 = w3;
            if (ip2 >= nextStep) { step++; nextStep += 128; }
        } while (ip3 < ilimit);
        if (!found) break;
        if (!isRep) {
            match0 = base + E::pos(matchE) - 1;
            rep2 = rep1; rep1 = (u32)(ip0 - match0); offcode = rep1 + 3; mLength = 4;
            while (((ip0 > anchor) & (match0 > base)) && (ip0[-1] == match0[-1])) { ip0--; match0--; mLength++; }
        }
        mLength += ze_count(ip0 + mLength, match0 + mLength, iend);
        ze_store(o, (u32)(anchor - istart), (u32)(ip0 - anchor), offcode, mLength);
        ip0 += mLength; anchor = ip0;
        if (ip0 <= ilimit) {
            {   u64 const wa = ld64(base + cur0 + 2), wb = ld64(ip0 - 2);
                table[ze_hash_w(wa, hlog, mls)] = E::make(cur0 + 2 + 1, ze_tag4((u32)wa));
                table[ze_hash_w(wb, hlog, mls)] = E::make((u32)(ip0 - 2 - base) + 1, ze_tag4((u32)wb)); }
            if (rep2 > 0) {
                while ((ip0 <= ilimit) && (ld32(ip0) == ld32(ip0 - rep2))) {
                    u32 const rLength = ze_count(ip0 + 4, ip0 + 4 - rep2, iend) + 4;
                    { u32 const t = rep2; rep2 = rep1; rep1 = t; }
                    {   u64 const wi = ld64(ip0); table[ze_hash_w(wi, hlog, mls)] = E::make((u32)(ip0 - base) + 1, ze_tag4((u32)wi)); }
                    ip0 += rLength;
                    ze_store(o, (u32)(anchor - istart), 0, 1, rLength);
                    anchor = ip0;
                }
            }
        }
    }
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    repOut[0] = rep1 ? rep1 : saved1; repOut[1] = rep2 ? rep2 : saved2;
    return (u32)(iend - anchor);
}
template <class E>
ZJ_DEV u32 ze_block_dfast_x(ZEOut& o, const u8* base, u32 start, u32 end, u32 hBitsL, u32 hBitsS, u32 mls, typename E::T* hashLong, typename E::T* hashSmall,
                            const u32* repIn, u32* repOut) {
    const u8* const istart = base + start; const u8* const iend = base + end; const u8* const ilimit = iend - 8;
    const u8* anchor = istart; const u8* ip = istart + (start == 0 ? 1 : 0); const u8* ip1;
    u32 off1 = repIn[0], off2 = repIn[1], saved1 = 0, saved2 = 0;
    {   u32 const maxRep = (u32)(ip - base);
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; } }
    u32 mLength = 0, offset = 0, curr = 0, step, hl0, hl1 = 0, el0, el1 = 0;
    const u8* nextStep; const u8* matchs0 = base; const u8* matchl0;
    for (;;) {
        step = 1; nextStep = ip + 256; ip1 = ip + step;
        if (ip1 > ilimit) break;
        u64 w = ld64(ip), w1 = ld64(ip1);
        hl0 = ze_hash_w(w, hBitsL, 8); u32 hs0 = ze_hash_w(w, hBitsS, mls);
        el0 = hashLong[hl0]; u32 es0 = hashSmall[hs0];
        u32 kind = 0;
        do {
            curr = (u32)(ip - base);
            u32 const tl = ze_tag8(w), ts = ze_tag4((u32)w);
            hashLong[hl0] = E::make(curr + 1, tl); hashSmall[hs0] = E::make(curr + 1, ts);
            bool const ml0 = E::maybe(el0, tl), ms0 = E::maybe(es0, ts);
            u32 const rv = ld32(ip + 1 - off1);
            u64 const cl = ml0 ? ld64(base + E::pos(el0) - 1) : ~w;
            u32 const cs = ms0 ? ld32(base + E::pos(es0) - 1) : ~(u32)w;
            hl1 = ze_hash_w(w1, hBitsL, 8); u32 const hs1 = ze_hash_w(w1, hBitsS, mls);
            el1 = hashLong[hl1]; u32 const es1 = hashSmall[hs1];
            u32 const stepN = step + ((ip1 >= nextStep) ? 1u : 0u);
            const u8* const ip2 = ip1 + stepN;
            u64 const w2 = (ip2 <= ilimit) ? ld64(ip2) : 0;
            if ((off1 > 0) & (rv == (u32)(w >> 8))) {
                mLength = ze_count(ip + 1 + 4, ip + 1 + 4 - off1, iend) + 4;
                ip++;
                ze_store(o, (u32)(anchor - istart), (u32)(ip - anchor), 1, mLength);
                kind = 1; break;
            }
            if (ml0 && cl == w) {
                matchl0 = base + E::pos(el0) - 1;
                mLength = ze_count(ip + 8, matchl0 + 8, iend) + 8;
                offset = (u32)(ip - matchl0);
                while (((ip > anchor) & (matchl0 > base)) && (ip[-1] == matchl0[-1])) { ip--; matchl0--; mLength++; }
                kind = 2; break;
            }
            if (ms0 && cs == (u32)w) { matchs0 = base + E::pos(es0) - 1; kind = 3; break; }
            if (ip1 >= nextStep) { step++; nextStep += 256; }
            ip = ip1; ip1 = ip2;
            hl0 = hl1; hs0 = hs1; el0 = el1; es0 = es1; w = w1; w1 = w2;
        } while (ip1 <= ilimit);
        if (kind == 0) break;
        if (kind == 3) {
            mLength = ze_count(ip + 4, matchs0 + 4, iend) + 4;
            offset = (u32)(ip - matchs0);
            if ((E::pos(el1) > 1) && E::maybe(el1, ze_tag8(w1)) && (ld64(base + E::pos(el1) - 1) == w1)) {
                const u8* const matchl1 = base + E::pos(el1) - 1;
                u32 const l1len = ze_count(ip1 + 8, matchl1 + 8, iend) + 8;
                if (l1len > mLength) { ip = ip1; mLength = l1len; offset = (u32)(ip - matchl1); matchs0 = matchl1; }
            }
            while (((ip > anchor) & (matchs0 > base)) && (ip[-1] == matchs0[-1])) { ip--; matchs0--; mLength++; }
        }
        if (kind >= 2) {
            off2 = off1; off1 = offset;
            if (step < 4) hashLong[hl1] = E::make((u32)(ip1 - base) + 1, ze_tag8(w1));
            ze_store(o, (u32)(anchor - istart), (u32)(ip - anchor), offset + 3, mLength);
        }
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            {   u32 const ins = curr + 2;
                u64 const wa = ld64(base + ins), wb = ld64(ip - 2), wc = ld64(ip - 1);
                hashLong[ze_hash_w(wa, hBitsL, 8)] = E::make(ins + 1, ze_tag8(wa));
                hashLong[ze_hash_w(wb, hBitsL, 8)] = E::make((u32)(ip - 2 - base) + 1, ze_tag8(wb));
                hashSmall[ze_hash_w(wa, hBitsS, mls)] = E::make(ins + 1, ze_tag4((u32)wa));
                hashSmall[ze_hash_w(wc, hBitsS, mls)] = E::make((u32)(ip - 1 - base) + 1, ze_tag4((u32)wc));
            }
            while ((ip <= ilimit) && ((off2 > 0) & (ld32(ip) == ld32(ip - off2)))) {
                u32 const rLength = ze_count(ip + 4, ip + 4 - off2, iend) + 4;
                u32 const t = off2; off2 = off1; off1 = t;
                {   u64 const wi = ld64(ip);
                    hashSmall[ze_hash_w(wi, hBitsS, mls)] = E::make((u32)(ip - base) + 1, ze_tag4((u32)wi));
                    hashLong[ze_hash_w(wi, hBitsL, 8)] = E::make((u32)(ip - base) + 1, ze_tag8(wi)); }
                ze_store(o, (u32)(anchor - istart), 0, 1, rLength);
                ip += rLength; anchor = ip;
            }
        }
    }
    saved2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;
    repOut[0] = off1 ? off1 : saved1; repOut[1] = off2 ? off2 : saved2;
    return (u32)(iend - anchor);
}

// ------------------------------------------------------------------ FSE (lane 0) ------------
ZJ_DEV u32 ze_fse_optimal_log(u32 maxTableLog, u32 srcSize, u32 maxSV, u32 minus) {   // fse_compress.c:346-372
    u32 const maxBitsSrc = zj_hibit(srcSize - 1) - minus;
    u32 tableLog = maxTableLog;
    u32 const minBitsSrc = zj_hibit(srcSize) + 1, minBitsSym = zj_hibit(maxSV) + 2;
    u32 const minBits = minBitsSrc < minBitsSym ? minBitsSrc : minBitsSym;
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < 5) tableLog = 5;
    if (tableLog > 12) tableLog = 12;
    return tableLog;
}

ZJ_DEV void ze_or_bits(u32* w, u32 bitpos, u64 lo, u64 hi, u32 nbits) {    // OR nbits (<=128) of {hi:lo} at bitpos
    u32 idx = bitpos >> 5; u32 const sh = bitpos & 31;
    // shift the 128-bit value left by sh (<32) into 160 bits = 5 words
    u32 const v0 = (u32)lo, v1 = (u32)(lo >> 32), v2 = (u32)hi, v3 = (u32)(hi >> 32);
    u32 const o0 = v0 << sh;
    u32 const o1 = sh ? ((v1 << sh) | (v0 >> (32 - sh))) : v1;
    u32 const o2 = sh ? ((v2 << sh) | (v1 >> (32 - sh))) : v2;
    u32 const o3 = sh ? ((v3 << sh) | (v2 >> (32 - sh))) : v3;
    u32 const o4 = sh ? (v3 >> (32 - sh)) : 0;
    u32 const words = (sh + nbits + 31) >> 5;
    if (words > 0 && o0) atomicOr(&w[idx], o0);
    if (words > 1 && o1) atomicOr(&w[idx + 1], o1);
    if (words > 2 && o2) atomicOr(&w[idx + 2], o2);
    if (words > 3 && o3) atomicOr(&w[idx + 3], o3);
    if (words > 4 && o4) atomicOr(&w[idx + 4], o4);
}

// ------------------------------------------------------------------ tANS tables by the wave ----
// What FSE_normalizeCount, FSE_writeNCount and FSE_buildCTable_wksp compute (N/compress/fse_compress.c:465-523, :237-328, :68-224), one symbol
// per lane.  The reference walks the alphabet with running state; here every quantity a symbol needs is a prefix sum or a reduction over the
// alphabet (<= 53 symbols: LL 36, OF 32, ML 53, Huffman weights 13), so the alphabet is handled in a few steps whatever its size:
//   shares     a symbol's share of the 2^tableLog cells from its own count; the cells handed out and the first-largest share by LDS atomics
//   describe   the cells left before a symbol = table size + 1 - (prefix sum of |share|) give its field's width and value in closed form; a
//              group of zero shares is its first zero plus a run-length prefix on the symbol behind it; every lane ORs its field at the
//              prefix sum of the widths
//   table      occurrence j of the spread lands on the j-th position (k * step mod size) that is not reserved for a low-probability symbol;
//              a position's slot in the state table is its symbol's first slot + its rank among that symbol's positions (bitmap popcount)
// `scr`: 192 words of LDS.  All functions are called by every lane; results in LDS are visible when they return.
#define ZE_TANS_SCR 192u
// the reference's rounding table for small shares (fse_compress.c:466)
ZE_CONST u32 ze_k_share_round[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };

// The uncommon outcome of ze_tans_shares — the largest symbol cannot absorb the rounding surplus: shares by thresholds first, the rest
// proportionally (fse_compress.c:377-462).  share[] on entry is overwritten.
template <class G>
ZJ_DEV bool ze_tans_shares_rare(const G& g, short* share, u32 tableLog, const u32* count, u32 total, u32 maxSV, i32 lowShare, u32* scr) {
    u32* const pre = scr; u32* const red = scr + 128;
    u32 const n = maxSV + 1u, size = 1u << tableLog;
    u32 const lowUpTo = total >> tableLog;
    u32 oneUpTo = (u32)(((u64)total * 3u) >> (tableLog + 1u));
    GRP_SERIAL(g) { red[0] = 0; red[1] = 0; red[2] = 0; red[3] = 0; red[4] = 0; }
    g.sync();
    GRP_FOR(g, s, n) {                                              // settled at once: nothing, the low share, one cell; open (-2) otherwise
        u32 const c = count[s]; i32 v = -2;
        if (c == 0) v = 0; else if (c <= lowUpTo) v = lowShare; else if (c <= oneUpTo) v = 1;
        if (c && v != -2) { atomicAdd(&red[0], 1u); atomicAdd(&red[1], c); }
        share[s] = (short)v;
    }
    g.sync();
    u32 settled = ZJ_UNI(red[0]), left = total - ZJ_UNI(red[1]);
    u32 cells = size - settled;
    if (cells == 0) return true;
    if (left / cells > oneUpTo) {                                   // the open symbols average more than the bar: raise it once
        oneUpTo = (u32)(((u64)left * 3u) / (cells * 2u));
        GRP_FOR(g, s, n) { u32 const c = count[s]; if (share[s] == -2 && c <= oneUpTo) { share[s] = 1; atomicAdd(&red[2], 1u); atomicAdd(&red[3], c); } }
        g.sync();
        settled += ZJ_UNI(red[2]); left -= ZJ_UNI(red[3]); cells = size - settled;
    }
    if (settled == n) {                                             // every symbol settled: the first most frequent one takes what is left
        GRP_FOR(g, s, n) atomicMax(&red[4], (count[s] << 6) | (63u - s));
        g.sync();
        GRP_SERIAL(g) { u32 const s = 63u - (red[4] & 63u); share[s] = (short)(share[s] + (i32)cells); }
        g.sync();
        return true;
    }
    if (left == 0) {                                                // nothing open: the cells go round the symbols that have a positive share
        GRP_FOR(g, s, 64u) pre[s] = (s < n && share[s] > 0) ? 1u : 0u;
        g.sync();
        grp_scan_incl(g, pre, 64u);
        u32 const holders = ZJ_UNI(pre[63]);
        GRP_FOR(g, s, n) if (share[s] > 0) { u32 const rank = pre[s] - 1u; share[s] = (short)(share[s] + (i32)(cells / holders) + (rank < cells % holders ? 1 : 0)); }
        g.sync();
        return true;
    }
    {   // open symbols share the cells in proportion: a symbol's share = the cells its interval of the running total covers
        u32 const fracBits = 62u - tableLog; u64 const half = ((u64)1 << (fracBits - 1u)) - 1u;
        u64 const perCount = ((((u64)1 << fracBits) * cells) + half) / left;
        GRP_FOR(g, s, 64u) pre[s] = (s < n && share[s] == -2) ? count[s] : 0u;
        g.sync();
        grp_scan_incl(g, pre, 64u);
        GRP_FOR(g, s, n) if (share[s] == -2) {
            u64 const hi = half + (u64)pre[s] * perCount, lo = half + (u64)(pre[s] - count[s]) * perCount;
            u32 const w = (u32)(hi >> fracBits) - (u32)(lo >> fracBits);
            if (w < 1u) atomicOr(&red[0], 0x80000000u);
            share[s] = (short)w;
        }
        g.sync();
        return !(ZJ_UNI(red[0]) & 0x80000000u);
    }
}

// share[s] of the 2^tableLog cells for every symbol s <= maxSV with count[s] occurrences out of `total` (low: symbols at or below the floor
// get -1, "less than one cell", instead of 1).  false: no valid table (the reference reports an error).
template <class G>
ZJ_DEV bool ze_tans_shares(const G& g, short* share, u32 tableLog, const u32* count, u32 total, u32 maxSV, bool low, u32* scr) {
    u32* const red = scr + 128;
    u32 const n = maxSV + 1u;
    i32 const lowShare = low ? -1 : 1;
    u32 const fracBits = 62u - tableLog, lowUpTo = total >> tableLog;
    u64 const perCount = ((u64)1 << 62) / total, roundUnit = (u64)1 << (fracBits - 20u);
    GRP_SERIAL(g) { red[0] = 0; red[1] = 63u; }                     // cells handed out; largest share << 6 | 63 - its (first) symbol
    g.sync();
    GRP_FOR(g, s, n) {
        u32 const c = count[s]; i32 v = 0;
        if (c != 0 && c <= lowUpTo) { v = lowShare; atomicAdd(&red[0], 1u); }
        else if (c != 0) {
            u64 const x = (u64)c * perCount;
            u32 w = (u32)(x >> fracBits);
            if (w < 8u) w += (x - ((u64)w << fracBits) > roundUnit * ze_k_share_round[w]) ? 1u : 0u;      // small shares round up past a share-dependent bar
            v = (i32)w;
            atomicAdd(&red[0], w); atomicMax(&red[1], (w << 6) | (63u - s));
        }
        share[s] = (short)v;
    }
    g.sync();
    u32 const top = 63u - (ZJ_UNI(red[1]) & 63u);
    i32 const surplus = (i32)(1u << tableLog) - (i32)ZJ_UNI(red[0]);    // what rounding left over (or overdrew): the largest share absorbs it
    if (-surplus >= ((i32)share[top] >> 1)) { g.sync(); return ze_tans_shares_rare(g, share, tableLog, count, total, maxSV, lowShare, scr); }
    g.sync();
    GRP_SERIAL(g) { share[top] = (short)(share[top] + surplus); }
    g.sync();
    return true;
}

// One symbol's field of the table description: `before` = cells left (+1) when the description reaches it.  A symbol behind zeros carries their run length in
// front of its own value (the first zero of the group is a field of its own; the run counts the others: 24 per 0xFFFF, 3 per "11", then two bits).
ZJ_DEV u32 ze_tans_field(const short* share, u32 s, u32 before, u32 size, u64& f) {
    i32 const v = share[s];
    bool const follows0 = s > 0u && share[s - 1u] == 0;
    if (before <= 1u || (v == 0 && follows0)) { f = 0; return 0; }
    u32 const thr = zj_min(size, 1u << zj_hibit(before)), nb = zj_hibit(thr) + 1u, spare = 2u * thr - 1u - before;
    u32 c = (u32)(v + 1);
    if (c >= thr) c += spare;
    u32 l = nb - (c < spare ? 1u : 0u);
    f = c;
    if (follows0) {
        u32 z = s - 1u; while (z > 0u && share[z - 1u] == 0) z--;
        u32 const run = s - 1u - z, ones = 16u * (run / 24u) + 2u * ((run % 24u) / 3u);
        f = (((u64)1 << ones) - 1u) | ((u64)((run % 24u) % 3u) << ones) | (f << (ones + 2u));
        l += ones + 2u;
    }
    return l;
}
// The table description (NCount) of share[0..maxSV] into `out` (LDS, 4-byte aligned, 128 bytes, cleared here).  Returns its size in bytes, 0 when the
// shares do not add up; *need = the bytes of destination the reference's writer asks for (its stores are two bytes wide, the last one decides).
template <class G>
ZJ_DEV u32 ze_tans_describe(const G& g, u8* out, const short* share, u32 maxSV, u32 tableLog, u32* scr, u32* need = nullptr) {
    u32* const pre = scr; u32* const width = scr + 64; u32* const w32 = (u32*)out;
    u32 const n = maxSV + 1u, size = 1u << tableLog;
    GRP_FOR(g, s, 64u) { i32 const v = s < n ? (i32)share[s] : 0; pre[s] = (u32)(v < 0 ? -v : v); }
    GRP_FOR(g, i, 32u) w32[i] = 0;
    g.sync();
    grp_scan_incl(g, pre, 64u);
    GRP_FOR(g, s, 64u) {
        u64 f; u32 l = 0;
        if (s < n) { i32 const v = share[s]; l = ze_tans_field(share, s, size + 1u - (pre[s] - (u32)(v < 0 ? -v : v)), size, f); }
        width[s] = l;
    }
    g.sync();
    grp_scan_incl(g, width, 64u);
    GRP_FOR(g, s, n) {                                              // every field at the prefix sum of the widths, behind the 4 bits of the table log
        u64 f; i32 const v = share[s];
        u32 const l = ze_tans_field(share, s, size + 1u - (pre[s] - (u32)(v < 0 ? -v : v)), size, f);
        if (l) ze_or_bits(w32, 4u + width[s] - l, f, 0, l);
    }
    GRP_SERIAL(g) { ze_or_bits(w32, 0, tableLog - 5u, 0, 4u); }
    g.sync();
    u32 const bitsAll = 4u + ZJ_UNI(width[63]);
    if (size + 1u - ZJ_UNI(pre[63]) != 1u) return 0;              // the shares have to cover the table exactly
    if (need) *need = 2u * ((bitsAll - 1u) / 16u) + 2u;
    return (bitsAll + 7u) >> 3;
}

// The encoding table of share[0..maxSV] (state table + per-symbol transforms).  tableSymbol: 2^tableLog bytes of LDS; bm: LDS words for the rank bitmaps,
// (maxSV + 1) * max(1, 2^tableLog / 32) of them.  `ct` may be in LDS or HBM.
template <class G>
ZJ_DEV void ze_tans_table(const G& g, ZEFseCT& ct, const short* share, u32 maxSV, u32 tableLog, u8* tableSymbol, u32* scr, u32* bm) {
    u32* const all = scr; u32* const pos = scr + 64; u32* const part = scr + 128;      // cells of symbols <= s (low ones count 1) / of the spread symbols <= s
    u32 const n = maxSV + 1u, size = 1u << tableLog, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    u32 const bmWords = size >= 32u ? size >> 5 : 1u;
    GRP_FOR(g, s, 64u) { i32 const v = s < n ? (i32)share[s] : 0; all[s] = (u32)(v < 0 ? 1 : v); pos[s] = (u32)(v > 0 ? v : 0); }
    GRP_FOR(g, i, n * bmWords) bm[i] = 0;
    g.sync();
    grp_scan_incl(g, all, 64u);
    grp_scan_incl(g, pos, 64u);
    u32 const nLow = ZJ_UNI(all[63]) - ZJ_UNI(pos[63]), high = size - 1u - nLow;
    GRP_SERIAL(g) { ct.tableLog = tableLog; }
    // per symbol: its transforms, and a low-probability symbol's reserved position at the top of the table (in symbol order, downwards)
    GRP_FOR(g, s, n) {
        i32 const v = share[s]; u32 const cells = (u32)(v < 0 ? 1 : v), first = all[s] - cells;
        if (v == 0) { ct.deltaNbBits[s] = ((tableLog + 1u) << 16) - size; ct.deltaFind[s] = 0; }
        else if (cells == 1u) { ct.deltaNbBits[s] = (tableLog << 16) - size; ct.deltaFind[s] = (i32)(first - 1u); }
        else { u32 const outBits = tableLog - zj_hibit(cells - 1u); ct.deltaNbBits[s] = (outBits << 16) - (cells << outBits); ct.deltaFind[s] = (i32)(first - cells); }
        if (v == -1) tableSymbol[size - 1u - ((first) - (pos[s]))] = (u8)s;           // first - pos[s] = low symbols before s
    }
    // the spread: visit k * step mod size for k = 0, 1, ...; the j-th visited position at or below `high` takes occurrence j.
    // A lane takes `per` consecutive visits; with reserved positions its first j is the count of accepted visits before its chunk.
    u32 const per = (size + 63u) >> 6;
    if (nLow) {
        GRP_FOR(g, l, 64u) { u32 c = 0; for (u32 k = l * per; k < (l + 1u) * per && k < size; k++) c += (((k * step) & mask) <= high) ? 1u : 0u; part[l] = c; }
        g.sync();
        grp_scan_incl(g, part, 64u);                                // (inclusive: part[l - 1] visits accepted before lane l's chunk)
    }
    g.sync();
    GRP_FOR(g, l, 64u) {
        u32 j = nLow ? (l ? part[l - 1u] : 0u) : l * per;
        for (u32 k = l * per; k < (l + 1u) * per && k < size; k++) {
            u32 const p = (k * step) & mask;
            if (p > high) continue;
            u32 lo = 0, hi = n - 1u;                                // the symbol whose occurrences include j: the first s with pos[s] > j
            while (lo < hi) { u32 const mid = (lo + hi) >> 1; if (pos[mid] > j) hi = mid; else lo = mid + 1u; }
            tableSymbol[p] = (u8)lo;
            j++;
        }
    }
    g.sync();
    // state table: positions in increasing order fill their symbol's slots in turn
    GRP_FOR(g, u, size) { u32 const sy = tableSymbol[u]; atomicOr(&bm[sy * bmWords + (u >> 5)], 1u << (u & 31u)); }
    g.sync();
    GRP_FOR(g, u, size) {
        u32 const sy = tableSymbol[u]; const u32* const b = bm + sy * bmWords;
        u32 rank = (u32)__builtin_popcount(b[u >> 5] & ((1u << (u & 31u)) - 1u));
        for (u32 w = 0; w < (u >> 5); w++) rank += (u32)__builtin_popcount(b[w]);
        i32 const v = share[sy]; u32 const cells = (u32)(v < 0 ? 1 : v);
        ct.state[all[sy] - cells + rank] = (u16)(size + u);
    }
    g.sync();
}

// forward bit writer (BIT_CStream_t semantics, N/common/bitstream.h:180-250) over global memory
struct ZEBitW { u8* p; u64 acc; u32 n; };
ZJ_DEV void ze_bw_add(ZEBitW& b, u64 v, u32 nb) {
    b.acc |= (v & (((u64)1 << nb) - 1)) << b.n; b.n += nb;
    if (b.n >= 32) { st32(b.p, (u32)b.acc); b.p += 4; b.acc >>= 32; b.n -= 32; }
}
ZJ_DEV u32 ze_bw_close(ZEBitW& b, const u8* start) {
    ze_bw_add(b, 1, 1);
    while (b.n > 0) { *b.p++ = (u8)b.acc; b.acc >>= 8; b.n = b.n > 8 ? b.n - 8 : 0; }
    return (u32)(b.p - start);
}
struct ZEFseCS { u32 value; };
ZJ_DEV void ze_fse_init2(ZEFseCS& s, const ZEFseCT& ct, u32 sym) {
    u32 const nbBitsOut = (ct.deltaNbBits[sym] + (1u << 15)) >> 16;
    s.value = (nbBitsOut << 16) - ct.deltaNbBits[sym];
    s.value = ct.state[(i32)(s.value >> nbBitsOut) + ct.deltaFind[sym]];
}
ZJ_DEV void ze_fse_encode(ZEBitW& b, ZEFseCS& s, const ZEFseCT& ct, u32 sym) {
    u32 const nbBitsOut = (s.value + ct.deltaNbBits[sym]) >> 16;
    ze_bw_add(b, s.value, nbBitsOut);
    s.value = ct.state[(i32)(s.value >> nbBitsOut) + ct.deltaFind[sym]];
}

// ------------------------------------------------------------------ Huffman (lane 0) --------
ZJ_DEV u32 ze_huf_bucket(u32 count) { return count < 166 ? count : zj_hibit(count) + 158; }

ZJ_DEV void ze_huf_insertion(ZENode* a, i32 low, i32 high) {
    i32 const size = high - low + 1; a += low;
    for (i32 i = 1; i < size; i++) { ZENode const key = a[i]; i32 j = i - 1; while (j >= 0 && a[j].count < key.count) { a[j + 1] = a[j]; j--; } a[j + 1] = key; }
}
ZJ_DEV i32 ze_huf_partition(ZENode* a, i32 low, i32 high) {
    u32 const pivot = a[high].count; i32 i = low - 1;
    for (i32 j = low; j < high; j++) if (a[j].count > pivot) { i++; ZENode t = a[i]; a[i] = a[j]; a[j] = t; }
    { ZENode t = a[i + 1]; a[i + 1] = a[high]; a[high] = t; }
    return i + 1;
}
// HUF_simpleQuickSort (huf_compress.c:587-607) with the recursion turned into an explicit stack: sub-ranges are disjoint,
// so deferring the "recursive" call does not change the result.  The recursive call applies the insertion-sort threshold
// at its entry, the iterative continuation does not; a deferred range below the threshold is sorted on the spot, so only
// ranges of >= 9 elements are ever stacked (<= 256 / 9 of them) — with equal counts every partition peels one element off
// and would otherwise stack an empty range per element.
ZJ_DEV void ze_huf_quicksort(ZEEntropy& e, ZENode* a, i32 low0, i32 high0) {
    i32 sp = 0; e.qsLow[0] = low0; e.qsHigh[0] = high0; sp = 1;
    while (sp > 0) {
        sp--; i32 low = e.qsLow[sp], high = e.qsHigh[sp];
        if (high - low < 8) { ze_huf_insertion(a, low, high); continue; }
        while (low < high) {
            i32 const idx = ze_huf_partition(a, low, high);
            i32 lo2, hi2;
            if (idx - low < high - idx) { lo2 = low; hi2 = idx - 1; low = idx + 1; }
            else { lo2 = idx + 1; hi2 = high; high = idx - 1; }
            if (hi2 - lo2 < 8) { if (hi2 > lo2) ze_huf_insertion(a, lo2, hi2); }
            else { e.qsLow[sp] = lo2; e.qsHigh[sp] = hi2; sp++; }
        }
    }
}

// HUF_buildCTable_wksp (HUF_sort + HUF_buildTree + HUF_setMaxHeight + HUF_buildCTableFromTree, huf_compress.c:376-754) by the whole wave.
//  * HUF_sort.  The reference drops the symbols into rank buckets (exact counts below 166, log2 classes above), largest bucket first, symbols of a
//    bucket in symbol order, and then quicksorts every bucket from count 164 upwards by count.  The result IS the order "count descending, symbol
//    ascending" — a 256-key bitonic sort, four keys per lane — unless a quicksorted bucket of nine or more members holds two equal counts: its
//    quicksort is not stable (smaller buckets go through its insertion sort, which is), and only the serial sort knows what it does then.
//  * HUF_buildTree's two-queue merge is a dependent chain (one lane, <= 255 steps); the leaves' depths follow in parallel.
//  * HUF_setMaxHeight (a table deeper than maxNbBits: rare) stays the serial repair.
//  * HUF_buildCTableFromTree: codes of a length are handed out in symbol order — a ballot per length and 64 symbols gives every symbol its place.
template <class G>
ZJ_DEV u32 ze_huf_build_wave(const G& g, ZEncShared& sh, ZEEntropy& e, u32 maxSV, u32 maxNbBits) {
    ZENode* const node = e.node + 1;
    const u32* const count = e.count;
    u32* const K = (u32*)e.stage;                          // 256 sort keys (the record stage is idle until the sequences section)
    u32* const bsize = e.scount;                           // members of the quicksorted buckets (164 ..), later: codes per length / first code of a length
    GRP_FOR(g, n, 516) { ZENode z; z.count = 0; z.parent = 0; z.byte = 0; z.nbBits = 0; e.node[n] = z; }
    GRP_FOR(g, r, 64) bsize[r] = 0;
    GRP_SERIAL(g) { sh.tmp[5] = 0; }
    g.sync();
    GRP_FOR(g, n, 256) {
        u32 const c = n <= maxSV ? count[n] : 0u;
        K[n] = (c << 8) | (255u - n);
        if (c >= 164u) atomicAdd(&bsize[ze_huf_bucket(c) - 164u], 1u);
    }
    g.sync();
    for (u32 k = 2; k <= 256u; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            GRP_FOR(g, t, 128) {
                u32 const i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), l = i | j;
                u32 const a = K[i], b = K[l];
                bool const down = (i & k) == 0u;           // this block ends up descending (the last merge: all of it)
                if (down ? a < b : a > b) { K[i] = b; K[l] = a; }
            }
            g.sync();
        }
    }
    GRP_FOR(g, i, 255) {
        u32 const c0 = K[i] >> 8, c1 = K[i + 1] >> 8;
        if (c0 == c1 && c0 >= 164u && bsize[ze_huf_bucket(c0) - 164u] >= 9u) atomicOr(&sh.tmp[5], 1u);
    }
    g.sync();
    if (ZJ_UNI(sh.tmp[5])) {
        GRP_SERIAL(g) {                                    // the reference's own sort, bucket by bucket (ze_huf_build's first part)
            for (u32 n = 0; n < 192; n++) { e.rankBase[n] = 0; e.rankCurr[n] = 0; }
            for (u32 n = 0; n <= maxSV; n++) e.rankBase[ze_huf_bucket(count[n])]++;
            for (u32 n = 191; n > 0; n--) { e.rankBase[n - 1] += e.rankBase[n]; e.rankCurr[n - 1] = e.rankBase[n - 1]; }
            for (u32 n = 0; n <= maxSV; n++) { u32 const r = ze_huf_bucket(count[n]) + 1; u32 const pos = e.rankCurr[r]++; node[pos].count = count[n]; node[pos].byte = (u8)n; }
            for (u32 n = 165; n < 191; n++) { i32 const sz = (i32)e.rankCurr[n] - (i32)e.rankBase[n]; if (sz > 1) ze_huf_quicksort(e, node + e.rankBase[n], 0, sz - 1); }
        }
    } else {
        GRP_FOR(g, pos, maxSV + 1u) { u32 const kv = K[pos]; node[pos].count = kv >> 8; node[pos].byte = (u8)(255u - (kv & 255u)); }
    }
    g.sync();
    GRP_SERIAL(g) {
        ZENode* const node0 = e.node;
        i32 nonNull = (i32)maxSV; while (node[nonNull].count == 0) nonNull--;
        i32 lowS = nonNull, nodeNb = 256, lowN = 256; i32 const nodeRoot = nodeNb + lowS - 1;
        node[nodeNb].count = node[lowS].count + node[lowS - 1].count;
        node[lowS].parent = node[lowS - 1].parent = (u16)nodeNb;
        nodeNb++; lowS -= 2;
        for (i32 n = nodeNb; n <= nodeRoot; n++) node[n].count = 1u << 30;
        node0[0].count = 1u << 31;
        while (nodeNb <= nodeRoot) {
            i32 const n1 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
            i32 const n2 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
            node[nodeNb].count = node[n1].count + node[n2].count;
            node[n1].parent = node[n2].parent = (u16)nodeNb; nodeNb++;
        }
        node[nodeRoot].nbBits = 0;
        for (i32 n = nodeRoot - 1; n >= 256; n--) node[n].nbBits = node[node[n].parent].nbBits + 1;
        sh.tmp[4] = (u32)nonNull;
    }
    g.sync();
    u32 const nonNull = ZJ_UNI(sh.tmp[4]);
    GRP_FOR(g, n, nonNull + 1u) node[n].nbBits = node[node[n].parent].nbBits + 1;
    g.sync();
    {   u32 const largestBits = ZJ_UNI((u32)node[nonNull].nbBits);
        if (largestBits > maxNbBits) {
            GRP_SERIAL(g) {                                // HUF_setMaxHeight (huf_compress.c:376-506)
                i32 totalCost = 0; u32 const baseCost = 1u << (largestBits - maxNbBits); u32 const noSymbol = 0xF0F0F0F0u;
                u32* const rankLast = e.scount;
                i32 k = (i32)nonNull;
                while (node[k].nbBits > maxNbBits) { totalCost += (i32)(baseCost - (1u << (largestBits - node[k].nbBits))); node[k].nbBits = (u8)maxNbBits; k--; }
                while (node[k].nbBits == maxNbBits) --k;
                totalCost >>= (largestBits - maxNbBits);
                for (u32 r = 0; r < 14; r++) rankLast[r] = noSymbol;
                {   u32 cur = maxNbBits;
                    for (i32 pos = k; pos >= 0; pos--) { if (node[pos].nbBits >= cur) continue; cur = node[pos].nbBits; rankLast[maxNbBits - cur] = (u32)pos; } }
                while (totalCost > 0) {
                    u32 nDec = zj_hibit((u32)totalCost) + 1;
                    for (; nDec > 1; nDec--) {
                        u32 const highPos = rankLast[nDec], lowPos = rankLast[nDec - 1];
                        if (highPos == noSymbol) continue;
                        if (lowPos == noSymbol) break;
                        if (node[highPos].count <= 2 * node[lowPos].count) break;
                    }
                    while ((nDec <= 12) && (rankLast[nDec] == noSymbol)) nDec++;
                    totalCost -= 1 << (nDec - 1);
                    node[rankLast[nDec]].nbBits++;
                    if (rankLast[nDec - 1] == noSymbol) rankLast[nDec - 1] = rankLast[nDec];
                    if (rankLast[nDec] == 0) rankLast[nDec] = noSymbol;
                    else { rankLast[nDec]--; if (node[rankLast[nDec]].nbBits != maxNbBits - nDec) rankLast[nDec] = noSymbol; }
                }
                while (totalCost < 0) {
                    if (rankLast[1] == noSymbol) { while (node[k].nbBits == maxNbBits) k--; node[k + 1].nbBits--; rankLast[1] = (u32)(k + 1); totalCost++; continue; }
                    node[rankLast[1] + 1].nbBits--; rankLast[1]++; totalCost++;
                }
            }
            g.sync();
        } else maxNbBits = largestBits;
    }
    {   u32* const nbPerRank = e.scount; u32* const valPerRank = e.scount + 16;
        GRP_FOR(g, r, 32) e.scount[r] = 0;
        g.sync();
        GRP_FOR(g, n, nonNull + 1u) atomicAdd(&nbPerRank[node[n].nbBits], 1u);
        GRP_FOR(g, n, 256) { e.nbBits[n] = 0; e.val[n] = 0; }
        g.sync();
        GRP_SERIAL(g) { u32 min = 0; for (i32 n = (i32)maxNbBits; n > 0; n--) { valPerRank[n] = min; min += nbPerRank[n]; min >>= 1; } }
        GRP_FOR(g, n, maxSV + 1u) e.nbBits[node[n].byte] = node[n].nbBits;
        g.sync();
        for (u32 r = 1; r <= maxNbBits; r++) {             // codes of length r, in symbol order
            u32 next = valPerRank[r];
            for (u32 base = 0; base <= maxSV; base += (u32)G::W) {
                u32 const n = base + g.lane();
                bool const mine = n <= maxSV && e.nbBits[n] == r;
                u64 const m = grp_ballot(g, mine);
                if (mine) e.val[n] = (u16)(next + (u32)__builtin_popcountll(m & (((u64)1 << g.lane()) - 1u)));
                next += (u32)__builtin_popcountll(m);
            }
        }
        g.sync();
    }
    return maxNbBits;
}

// HUF_compressWeights (huf_compress.c:132-176): 0 not compressible, 1 rle, else size.  Called by every lane, the result is wave-uniform: the statistics and the
// tANS table of the weights by the wave (ze_tans_*), the two-state encode of the <= 255 weights on one lane.
// `cap` = the room the reference's call is given (everything left of the block's destination): tight |= 1 when its table description
// does not fit (an error there), tight |= 2 when its bit stream does not (8 bytes of slack: BIT_initCStream / BIT_closeCStream,
// bitstream.h:177-186, 258-267 — "not compressible" there, i.e. the raw weights follow).  The bytes written here do not depend on `cap`.
template <class G>
ZJ_DEV u32 ze_huf_compress_weights(const G& g, ZEEntropy& e, u8* dst, u32 wtSize, u32 cap, u32& tight) {
    const u8* const w = e.weight; u32* const count = e.scount; short* const share = e.norm;
    u32* const scr = (u32*)e.rankBase; u32* const uni = scr + 160;        // (rankBase / rankCurr: the Huffman build is done with them)
    u8* const desc = (u8*)&e.node[0]; u32* const bm = (u32*)&e.node[0] + 32;   // (so is the tree)
    if (wtSize <= 1) return 0;
    GRP_FOR(g, s, 16u) count[s] = 0;
    GRP_SERIAL(g) { uni[0] = 0; uni[1] = 0; uni[2] = 0; uni[3] = 0; }
    g.sync();
    GRP_FOR(g, i, wtSize) atomicAdd(&count[w[i]], 1u);
    g.sync();
    GRP_FOR(g, s, 13u) { u32 const c = count[s]; if (c) { atomicMax(&uni[0], s); atomicMax(&uni[1], c); } }
    g.sync();
    u32 const maxSV = ZJ_UNI(uni[0]), maxCount = ZJ_UNI(uni[1]);
    if (maxCount == wtSize) return 1;
    if (maxCount == 1) return 0;
    u32 const tableLog = ze_fse_optimal_log(6, wtSize, maxSV, 2);
    if (!ze_tans_shares(g, share, tableLog, count, wtSize, maxSV, false, scr)) return 0;
    u32 need = 0;
    u32 const h = ze_tans_describe(g, desc, share, maxSV, tableLog, scr, &need);
    if (!h) return 0;
    GRP_FOR(g, i, h) dst[i] = desc[i];
    if (need > cap) tight |= 1u;
    g.sync();
    ZEFseCT& ct = e.ct[0];
    ze_tans_table(g, ct, share, maxSV, tableLog, e.tableSymbol, scr, bm);
    GRP_SERIAL(g) {                                               // fse_compress.c:549-606: two interleaved states, last weight first
        u8* const bstart = dst + h; const u8* ip = w + wtSize; ZEBitW b; ZEFseCS s1, s2; u32 n = wtSize; u32 out = 0, tg = 0;
        if (n > 2) {
            b.p = bstart; b.acc = 0; b.n = 0;
            if (n & 1) { ze_fse_init2(s1, ct, *--ip); ze_fse_init2(s2, ct, *--ip); ze_fse_encode(b, s1, ct, *--ip); }
            else { ze_fse_init2(s2, ct, *--ip); ze_fse_init2(s1, ct, *--ip); }
            n -= 2;
            if (n & 2) { ze_fse_encode(b, s2, ct, *--ip); ze_fse_encode(b, s1, ct, *--ip); }
            while (ip > w) { ze_fse_encode(b, s2, ct, *--ip); ze_fse_encode(b, s1, ct, *--ip); ze_fse_encode(b, s2, ct, *--ip); ze_fse_encode(b, s1, ct, *--ip); }
            ze_bw_add(b, s2.value, ct.tableLog); ze_bw_add(b, s1.value, ct.tableLog);
            {   u32 const fullBytes = (u32)((((u64)(b.p - bstart)) * 8u + b.n + 1u) >> 3);      // whole bytes before the end mark's byte must stay 8 short of the end
                if (cap < h + 9u || fullBytes + 8u >= cap - h) tg = 2u; }
            out = h + ze_bw_close(b, bstart);
        }
        uni[2] = out; uni[3] = tg;
    }
    g.sync();
    tight |= ZJ_UNI(uni[3]);
    return ZJ_UNI(uni[2]);
}

// HUF_writeCTable_wksp (huf_compress.c:248-290); 0 on failure.  `cap` / `tight`: the reference's maxDstSize, and whether it would have
// run out of it on the way to these bytes (the caller then treats the block as the reference does: not compressible).  By every lane; wave-uniform result.
template <class G>
ZJ_DEV u32 ze_huf_write_ctable(const G& g, ZEEntropy& e, u8* op, u32 maxSV, u32 huffLog, u32 cap, bool& tight) {
    GRP_FOR(g, n, maxSV) e.weight[n] = e.nbBits[n] ? (u8)(huffLog + 1 - e.nbBits[n]) : 0;
    g.sync();
    {   u32 tw = 0;
        u32 const h = ze_huf_compress_weights(g, e, op + 1, maxSV, cap ? cap - 1u : 0u, tw);
        if (cap < 1u || (tw & 1u)) tight = true;
        if ((h > 1) & (h < maxSV / 2)) { if (tw) tight = true; GRP_SERIAL(g) { op[0] = (u8)h; } return h + 1; } }
    if (((maxSV + 1) / 2) + 1 > cap) tight = true;
    if (maxSV > 128) return 0;
    g.sync();
    GRP_SERIAL(g) { op[0] = (u8)(128 + (maxSV - 1)); e.weight[maxSV] = 0; }
    g.sync();
    GRP_FOR(g, k, (maxSV + 1u) / 2u) op[k + 1u] = (u8)((e.weight[2u * k] << 4) + e.weight[2u * k + 1u]);
    return ((maxSV + 1) / 2) + 1;
}

// one Huffman stream, symbols last -> first (huf_compress.c:991-1118); runs on one lane
ZJ_DEV void ze_huf_encode_stream(const ZEEntropy& e, u8* dst, const u8* lit, u32 n) {
    ZEBitW b; b.p = dst; b.acc = 0; b.n = 0;
    u32 i = n;
    while (i >= 8) {                                   // 8 symbols per global load, last symbol first
        u64 const w = ld64(lit + i - 8);
        for (int k = 7; k >= 0; k--) { u32 const s = (u32)(w >> (8 * k)) & 0xFF; ze_bw_add(b, e.val[s], e.nbBits[s]); }
        i -= 8;
    }
    for (; i > 0; i--) { u32 const s = lit[i - 1]; ze_bw_add(b, e.val[s], e.nbBits[s]); }
    ze_bw_close(b, dst);
}


// ------------------------------------------------------------------ parallel bit packing -----
// A forward bitstream (BIT_CStream_t layout) assembled by many lanes: every lane ORs its bits into a
// zero-initialised LDS window at its own bit offset (offsets come from a prefix sum of bit counts), then
// the wave flushes the completed 32-bit words to HBM and carries the partial word over.
struct ZEStageBits { u32* w; u8* dst; u32 flushedWords; u32 carryBits; };   // wave-uniform


// flush the words completed by `addBits` new bits; keeps the partial word as the new word 0
template <class G>
ZJ_DEV void ze_stage_flush(const G& g, ZEncShared& sh, ZEStageBits& st, u32 addBits) {
    u32 const total = st.carryBits + addBits, nW = total >> 5;
    g.sync();
    GRP_FOR(g, i, nW) st32(st.dst + 4 * (st.flushedWords + i), st.w[i]);
    GRP_SERIAL(g) { sh.tmp[4] = st.w[nW]; }
    g.sync();
    u32 const carry = ZJ_UNI(sh.tmp[4]);
    GRP_FOR(g, i, nW + 1) st.w[i] = (i == 0) ? carry : 0u;
    g.sync();
    st.flushedWords += nW; st.carryBits = total & 31;
}
// final partial bytes; returns the stream size in bytes
template <class G>
ZJ_DEV u32 ze_stage_finish(const G& g, ZEStageBits& st) {
    u32 const nbytes = (st.carryBits + 7) >> 3;
    g.sync();
    GRP_FOR(g, i, nbytes) st.dst[4 * st.flushedWords + i] = ((const u8*)st.w)[i];
    g.sync();
    return 4 * st.flushedWords + nbytes;
}

// One Huffman stream (huf_compress.c:991-1118: symbols last -> first, then the end mark) packed by the
// whole wave: lane l takes 32 symbols, a prefix sum of chunk bit lengths gives its offset.
// `codes` = e.val | e.nbBits << 16 per symbol (LDS), `stage` = >= 1024 zeroable LDS words.
#define ZE_HUF_CHUNK 32u
// up to 32 symbols lit[beg .. beg+cnt) into four words (symbol i = byte i); full chunks take four 8-byte loads
ZJ_DEV void ze_chunk_load(const u8* lit, u32 beg, u32 cnt, u64& c0, u64& c1, u64& c2, u64& c3) {
    if (cnt == ZE_HUF_CHUNK) { c0 = ld64(lit + beg); c1 = ld64(lit + beg + 8); c2 = ld64(lit + beg + 16); c3 = ld64(lit + beg + 24); return; }
    c0 = c1 = c2 = c3 = 0;
    for (u32 i = 0; i < cnt; i++) {
        u64 const b = (u64)lit[beg + i] << (8 * (i & 7));
        if (i < 8) c0 |= b; else if (i < 16) c1 |= b; else if (i < 24) c2 |= b; else c3 |= b;
    }
}
ZJ_DEV u32 ze_chunk_byte(u64 c0, u64 c1, u64 c2, u64 c3, u32 i) {      // i is a compile-time constant after unrolling
    u64 const w = i < 8 ? c0 : (i < 16 ? c1 : (i < 24 ? c2 : c3));
    return (u32)(w >> (8 * (i & 7))) & 0xFFu;
}
template <class G>
ZJ_DEV u32 ze_huf_encode_wave(const G& g, ZEncShared& sh, const u32* codes, u32* stage, u32* lb, u8* dst, const u8* lit, u32 n) {
    ZEStageBits st; st.w = stage; st.dst = dst; st.flushedWords = 0; st.carryBits = 0;
    GRP_FOR(g, i, 1024) stage[i] = 0;
    g.sync();
    u32 const R = 64u * ZE_HUF_CHUNK;
    u64 cw0 = 0, cw1 = 0, cw2 = 0, cw3 = 0;               // a lane's 32 symbols: fetched once per round (4 loads), used by both passes
    for (u32 hi = n; hi > 0; ) {
        u32 const take = zj_min(R, hi);
        // pass 1: bit length of every lane's chunk (lane 0 = the last symbols = lowest bit positions)
        GRP_FOR(g, l, 64) {
            u32 const end = hi > l * ZE_HUF_CHUNK ? hi - l * ZE_HUF_CHUNK : 0, beg = hi > (l + 1) * ZE_HUF_CHUNK ? hi - (l + 1) * ZE_HUF_CHUNK : 0;
            ze_chunk_load(lit, beg, end - beg, cw0, cw1, cw2, cw3);
            u32 bits = 0;
#if ZJ_ON_GPU
#pragma unroll
#endif
            for (u32 i = 0; i < ZE_HUF_CHUNK; i++) if (i < end - beg) bits += codes[ze_chunk_byte(cw0, cw1, cw2, cw3, i)] >> 16;
            lb[l] = bits;
        }
        g.sync();
        grp_scan_incl(g, lb, 64);
#if !ZJ_ON_GPU
        g.sync();
#endif
        // pass 2: emit
        GRP_FOR(g, l, 64) {
            u32 const end = hi > l * ZE_HUF_CHUNK ? hi - l * ZE_HUF_CHUNK : 0, beg = hi > (l + 1) * ZE_HUF_CHUNK ? hi - (l + 1) * ZE_HUF_CHUNK : 0;
            if (G::W == 1) ze_chunk_load(lit, beg, end - beg, cw0, cw1, cw2, cw3);     // lane-serial build: one lane plays all 64
            u32 pos = st.carryBits + (l ? lb[l - 1] : 0);
            u32 idx = pos >> 5, nb = pos & 31; u64 acc = 0;
#if ZJ_ON_GPU
#pragma unroll
#endif
            for (u32 r = 0; r < ZE_HUF_CHUNK; r++) {
                u32 const i = ZE_HUF_CHUNK - 1u - r;                 // last symbol of the chunk first
                if (i < end - beg) {
                    u32 const c = codes[ze_chunk_byte(cw0, cw1, cw2, cw3, i)];
                    acc |= (u64)(c & 0xFFFF) << nb; nb += c >> 16;
                    if (nb >= 32) { atomicOr(&stage[idx++], (u32)acc); acc >>= 32; nb -= 32; }
                }
            }
            if (nb && (u32)acc) atomicOr(&stage[idx], (u32)acc);
        }
        g.sync();
        u32 const roundBits = ZJ_UNI(lb[63]);
        ze_stage_flush(g, sh, st, roundBits);
        hi -= take;
    }
    GRP_SERIAL(g) { atomicOr(&stage[st.carryBits >> 5], 1u << (st.carryBits & 31)); }    // end mark
    st.carryBits += 1;
    return ze_stage_finish(g, st);
}

// byte histogram of p[0..cnt) into h[256] (LDS): 8 literals per load
template <class G>
ZJ_DEV void ze_hist_add(const G& g, u32* h, const u8* p, u32 cnt) {
    u32 const words = cnt >> 3;
    GRP_FOR(g, w, words) {
        u64 const v = ld64(p + 8 * w);
#if ZJ_ON_GPU
#pragma unroll
#endif
        for (u32 b = 0; b < 8; b++) atomicAdd(&h[(u32)(v >> (8 * b)) & 0xFFu], 1u);
    }
    GRP_FOR(g, i, cnt & 7u) atomicAdd(&h[p[(words << 3) + i]], 1u);
}

// raw / rle literal sections (zstd_compress_literals.c:39-127)
template <class G>
ZJ_DEV u32 ze_raw_literals(const G& g, u8* dst, const u8* lit, u32 n) {
    u32 const fl = 1 + (n > 31) + (n > 4095);
    GRP_SERIAL(g) { if (fl == 1) dst[0] = (u8)(n << 3); else if (fl == 2) st16(dst, (1u << 2) + (n << 4)); else { st16(dst, ((3u << 2) + (n << 4)) & 0xFFFF); dst[2] = (u8)(((3u << 2) + (n << 4)) >> 16); } }
    GRP_FOR(g, i, n) dst[fl + i] = lit[i];
    return n + fl;
}

// ------------------------------------------------------------------ dictionary (ZSTD_CDict) --
#define ZC_REPEAT_NONE 0u                    /* HUF_repeat / FSE_repeat */
#define ZC_REPEAT_CHECK 1u
#define ZC_REPEAT_VALID 2u
// A digested dictionary in HBM: header, then the tagged tables (hashLong[1 << hashLog], hashSmall[1 << chainLog] for
// double-fast; one table for fast), then the raw dictionary bytes.
struct ZECDictDev {
    u32 status;                         // 0 ok, else ZJ_E_*
    u32 dictID, contentOff, contentSize;
    u32 level, hasEntropy;
    u32 windowLog, chainLog, hashLog, minMatch, strategy;       // the dictionary's own cParams
    u32 fillStart;                      // first content offset inserted into the tables
    u32 rep[3];
    u32 hufRepeat, llRepeat, ofRepeat, mlRepeat;
    u32 hufMaxSV, hufLog;
    u32 tablesOff, rawOff;              // byte offsets from the start of this struct
    u8 hufNbBits[256]; u16 hufVal[256];
    ZEFseCT fse[3];                     // LL, OF, ML encoding tables (FSE_buildCTable_wksp of the dictionary's NCounts)
};

ZJ_DEV const u32* ze_cdict_tables(const ZECDictDev* cd) { return (const u32*)((const u8*)cd + cd->tablesOff); }
ZJ_DEV const u8* ze_cdict_content(const ZECDictDev* cd) { return (const u8*)cd + cd->rawOff + cd->contentOff; }

// ------------------------------------------------------------------ frame -------------------
// `lds` = the overlay region (tables / ZEEntropy), ldsBytes its size.
template <class TIdx> struct ZEEntOf;
template <> struct ZEEntOf<u16> { typedef ZEEnt16 E; };
template <> struct ZEEntOf<u32> { typedef ZEEnt32 E; };

#define ZE_FLAG_CHECKSUM 1u      /* ZSTD_c_checksumFlag: append XXH64(content) & 0xFFFFFFFF */
#define ZE_FLAG_NO_FCS 2u        /* ZSTD_c_contentSizeFlag = 0 (ZstdCompressCtx.setContentSize(false)): no frame content size, window descriptor instead */
#define ZE_FLAG_NO_DICTID 4u     /* ZSTD_c_dictIDFlag = 0 (ZstdCompressCtx.setDictID(false)): the dictionary's ID stays out of the header */
#define ZE_FLAG_MASK 7u
#define ZE_FLAG_MULTI_NOCARRY 16u /* library switch (ZJNI_MULTI_WAVE=2): the wave matcher without the staged spans behind a match — A/B runs */
#define ZE_FLAG_MULTI_FAST_SERIAL 32u /* library switch (ZJNI_MULTI_WAVE_FAST=0): levels 1-2 blocks of multi-block frames on the one-lane parse, level 3 as the other switches say */
#define ZE_FLAG_MULTI_SERIAL 8u /* library switch, not a frame parameter (ZJNI_MULTI_WAVE=0): level-3 blocks of multi-block frames on the one-lane parse instead of zj_match_wavex.h */
// Sequences found ahead of time by the lane-per-frame match-finder kernel (zj_enc_match_kernel)
struct ZEPre { ZESeq* seqs; const u32* litOff; const u32* meta; u32 copyMode = 0; };   // meta = {nbSeq, litSize, lastLL}; copyMode: the records come from the dictionary copy-mode search (zj_cdict.h)

// zj_match_wavex.h (included at the end of this file): the wave-per-frame parse of one level-3 block of a multi-block frame
ZJ_DEV u32 zx_block_dfast_wave(u8* lds, ZEOut& o, const u8* base, u32 frameSize, u32 start, u32 end, u32 hBitsL, u32 hBitsS, u32 mls,
                               u32* hashLong, u32* hashSmall, const u32* repIn, u32* repOut, bool carry);
ZJ_DEV u32 zx_block_fast_wave(u8* lds, ZEOut& o, const u8* base, u32 frameSize, u32 start, u32 end, u32 hBits, u32 mls,
                              u32* table, const u32* repIn, u32* repOut, bool carry);
// `ba` != nullptr: src0[0, srcSize) is ONE BLOCK of a multi-block frame (ze_compress_multi): no frame header or checksum here, the
// match finder runs over the frame's tables and repcodes, the previous compressed block's Huffman table may be repeated, and the
// return value is the size of the block with its 3-byte header.
template <class G, class TIdx>
ZJ_DEV u64 ze_compress_t(const G& g, ZEncShared& sh, u8* lds, const u8* src0, u32 srcSize, u8* dst, u32 dstCap, u32 level, u8* ws, ZjProf& pf, const ZEPre* pre, u32 flags, const ZECDictDev* cd, u32 ldsBytes,
                         const ZEBlockArgs* ba = nullptr, u32* hbmTables = nullptr) {
    u32 const tail = (!ba && (flags & ZE_FLAG_CHECKSUM)) ? 4u : 0u;         // XXH64 low 32 bits after the last block (ZSTD_writeEpilogue)
    // Small frames whose sequences are already found: the source is staged into LDS once, literals are gathered and the
    // block body is assembled there, so the stage's many short dependent steps run at LDS latency, not HBM latency.
    u32 const a16 = ZE_ALIGN16(srcSize);
    bool const small = pre && !ba && srcSize <= ZE_SMALL_MAX && srcSize > 0 && ldsBytes >= ZE_ENTROPY_LDS + 2u * a16 + 1024u + 128u;
    u8* const xl = lds + ZE_ENTROPY_LDS;
    const u8* const src = small ? (const u8*)xl : src0;
    u8* const litBuf = small ? xl + a16 + 1024u + 64u : ws + ZE_WS_LIT;   // [staged source, later the block body | 1 KiB slack][literals]
    ZESeq* const seqs = pre ? pre->seqs : (ZESeq*)(ws + ZE_WS_SEQ);
    ZEEntropy& e = *(ZEEntropy*)lds;
    if (small) grp_copy_wide(g, xl, src0, srcSize);                // in flight while lane 0 writes the header
    if (cd && !ZJ_UNI(sh.dictLoaded)) {                            // once per workgroup: what every frame needs from the dictionary
        GRP_FOR(g, s, 256) sh.dictCodes[s] = (u32)cd->hufVal[s] | ((u32)cd->hufNbBits[s] << 16);
        GRP_SERIAL(g) {
            sh.dictID = cd->dictID; sh.dictStrategy = cd->strategy; sh.dictMinMatch = cd->minMatch; sh.dictHufRep = cd->hufRepeat; sh.dictHufMaxSV = cd->hufMaxSV;
            sh.dictFseRep[0] = cd->llRepeat; sh.dictFseRep[1] = cd->ofRepeat; sh.dictFseRep[2] = cd->mlRepeat;
        }
        g.sync();
        GRP_SERIAL(g) { sh.dictLoaded = 1; }
    }

    // ---- frame header (ZSTD_writeFrameHeader, contentSizeFlag = 1; dictID of the attached dictionary if it has one) ----
    GRP_SERIAL(g) {
        sh.err = 0; sh.tight = 0; sh.tightHuf = 0;
        if (cd) { sh.strategy = sh.dictStrategy; sh.minMatch = sh.dictMinMatch; sh.windowLog = 0; sh.hashLog = 0; sh.chainLog = 0; }
        else { ze_params(sh, level, ba ? (ba->paramSize ? ba->paramSize : ba->frameSize) : srcSize); if (sh.strategy == 0) sh.err = 201; }     // a (level, size) the reference serves with a finder this library does not restate
        if (pre) { sh.nbSeq = pre->meta[0]; sh.litSize = pre->meta[1]; sh.lastLL = pre->meta[2]; }
        if (ba) { sh.hdrSize = 0; if (dstCap < 3u + 2u + 1u) sh.err = ZJ_E_DSTSIZE_TOO_SMALL; }     // zstd_compress.c:4629-4631
        else {
        // ZSTD_writeFrameHeader (zstd_compress.c:4695-4745): with the content size (the default) a frame <= 128 KiB is single-segment;
        // without it (contentSizeFlag = 0) the window descriptor of the adjusted windowLog takes its place
        bool const noFcs = (flags & ZE_FLAG_NO_FCS) != 0;
        u32 const dictID = (cd && !(flags & ZE_FLAG_NO_DICTID)) ? sh.dictID : 0u;
        u32 const didCode = (dictID > 0) + (dictID >= 256) + (dictID >= 65536);
        u32 const didBytes = didCode == 3 ? 4u : didCode;
        u32 const fcsCode = noFcs ? 0u : (srcSize >= 256) + (srcSize >= 65536 + 256);
        u32 const hdr = 5 + didBytes + (noFcs ? 1 : (fcsCode == 0 ? 1 : (fcsCode == 1 ? 2 : 4)));
        sh.hdrSize = hdr;
        // with a dictionary the sequences come from outside (attach-mode search up to the cutoff, copy-mode search — zj_cdict.h — beyond it while
        // the reference still compresses with the dictionary's own parameters: one block, source < 128 KiB or < 6 x the dictionary)
        // (the attach pipeline's entropy kernel still meets such frames — without records: it has to refuse them, the copy-mode kernel serves them)
        if (cd && (!pre || (srcSize > (sh.dictStrategy == 2 ? (16u << 10) : (8u << 10))
                            && !(pre->copyMode && srcSize <= (128u << 10) && (srcSize < (128u << 10) || (u64)srcSize < (u64)cd->contentSize * 6u))))) sh.err = ZJ_E_PARAM_UNSUPPORTED;
        else if (cd && noFcs) sh.err = ZJ_E_PARAM_UNSUPPORTED;               // (the attached dictionary's window descriptor is not restated here)
        // room: ZSTD_writeFrameHeader wants ZSTD_FRAMEHEADERSIZE_MAX = 18 bytes whatever the header's size (zstd_compress.c:4711-4712), the block
        // loop 3 + 2 + 1 bytes before every block (:4629-4631), the epilogue of an empty frame its 3 (+ 4) bytes (:5362, :5371)
        else if (dstCap < 18u || (srcSize ? dstCap - hdr < 6u : dstCap < hdr + 3 + tail)) sh.err = ZJ_E_DSTSIZE_TOO_SMALL;
        else {
            st32(dst, 0xFD2FB528u); dst[4] = (u8)((noFcs ? 0u : (1u << 5)) + (fcsCode << 6) + (tail ? 4u : 0u) + didCode);
            u8* dp = dst + 5;
            if (noFcs) *dp++ = (u8)((sh.windowLog - 10u) << 3);
            if (didBytes == 1) dp[0] = (u8)dictID; else if (didBytes == 2) st16(dp, dictID); else if (didBytes == 4) st32(dp, dictID);
            u8* const fp = dp + didBytes;
            if (noFcs) {} else if (fcsCode == 0) fp[0] = (u8)srcSize; else if (fcsCode == 1) st16(fp, srcSize - 256); else st32(fp, srcSize);
        }
        }
    }
    g.sync();
    if (ZJ_UNI(sh.err)) return ZJ_ERR64(ZJ_UNI(sh.err));
    u32 const hdr = ZJ_UNI(sh.hdrSize);
    if (srcSize == 0) {                                                       // ZSTD_writeEpilogue: empty raw last block
        GRP_SERIAL(g) { dst[hdr] = 1; dst[hdr + 1] = 0; dst[hdr + 2] = 0; if (tail) st32(dst + hdr + 3, (u32)zj_xx_finish(ZJ_XXP5, src, 0)); }
        return hdr + 3 + tail;
    }
    bool compressed = false; u32 cSize = 0;
    // Tight destinations.  The reference compresses a block straight into what is left of the destination (capBody bytes behind the block
    // header) and its writers want working room: literals and table descriptions are refused when they do not fit, every bit stream needs
    // 8 bytes of slack behind its last whole byte (HUF_initCStream / HUF_closeCStream, huf_compress.c:862-871, 975-984; BIT_initCStream /
    // BIT_closeCStream), table descriptions are stored two bytes at a time.  Whenever one of them gives up, ZSTD_entropyCompressSeqStore
    // calls the block "not compressible" if it fits raw and fails with dstSize_tooSmall if not (zstd_compress.c:3024-3030) — every detour
    // in between (raw literals where Huffman found no room, raw weights where their tANS stream found none) ends there too, because what
    // it writes instead is larger.  So the block is assembled as always and `sh.tight` records whether any step of the reference's walk
    // through the same bytes would have given up; a tight block is then emitted raw, or refused.
    u32 const capBody = dstCap - hdr - 3u;
    // body goes straight into dst when even the raw fallback fits, else into HBM scratch first
    bool const direct = !small && dstCap >= hdr + 3 + srcSize + 64;
    u8* const body = small ? xl : (direct ? dst + hdr + 3 : ws + ZE_WS_BODY);   // small: over the staged source, dead once the literals are gathered
    if (srcSize >= 7) {                                                       // ZSTD_buildSeqStore: MIN_CBLOCK_SIZE + 3 + 1 + 1
        // ---- match finding: zero the tables (all lanes), then the sequential parse (lane 0) ----
        u32 const strategy = ZJ_UNI(sh.strategy), hlog = ZJ_UNI(sh.hashLog), clog = ZJ_UNI(sh.chainLog), mls = ZJ_UNI(sh.minMatch);
        if (!pre && ba) {                                  // one block of a frame: its tables (HBM, cleared by the caller before block 0) and repcodes carry on
#if defined(ZJ_TUNING_KERNELS) || !ZJ_ON_GPU          /* ZWaveF: exact, measured slower than the one-lane parse at levels 1-2 (2 048 x 512 KiB: 156 ms against 138) — tuning builds and the emulation only */
            if (strategy == 1 && ba->serialParse != 1u && !(ba->serialParse & 4u)) {     // levels 1-2: the fast strategy on the whole wave (zj_match_wavex.h, ZWaveF)
                ZEOut o; o.seqs = seqs; o.litOff = (u32*)(ws + ZE_WS_BODY); o.n = 0; o.lit = 0;
                u32 const lastLL = zx_block_fast_wave(lds, o, ba->frameBase, ba->frameSize, ba->start, ba->start + srcSize, hlog, mls, ba->tables, sh.blkRep, sh.blkNextRep, (ba->serialParse & 3u) == 0u);
                GRP_SERIAL(g) { sh.nbSeq = o.n; sh.litSize = o.lit + lastLL; sh.lastLL = lastLL; }
            } else
#endif
            if (strategy == 2 && (ba->serialParse & 3u) != 1u) {           // level 3: the whole wave (zj_match_wavex.h); the dynamic LDS is free until the entropy stage
                ZEOut o; o.seqs = seqs; o.litOff = (u32*)(ws + ZE_WS_BODY); o.n = 0; o.lit = 0;
                u32 const lastLL = zx_block_dfast_wave(lds, o, ba->frameBase, ba->frameSize, ba->start, ba->start + srcSize, hlog, clog, mls, ba->tables, ba->tables + (1u << hlog), sh.blkRep, sh.blkNextRep, (ba->serialParse & 3u) == 0u);
                GRP_SERIAL(g) { sh.nbSeq = o.n; sh.litSize = o.lit + lastLL; sh.lastLL = lastLL; }
            } else
            GRP_SERIAL(g) {
                ZEOut o; o.seqs = seqs; o.litOff = (u32*)(ws + ZE_WS_BODY); o.n = 0; o.lit = 0;
                u32 const lastLL = (strategy == 1) ? ze_block_fast_x<ZEEnt32>(o, ba->frameBase, ba->start, ba->start + srcSize, hlog, mls, ba->tables, sh.blkRep, sh.blkNextRep)
                                                   : ze_block_dfast_x<ZEEnt32>(o, ba->frameBase, ba->start, ba->start + srcSize, hlog, clog, mls, ba->tables, ba->tables + (1u << hlog), sh.blkRep, sh.blkNextRep);
                sh.nbSeq = o.n; sh.litSize = o.lit + lastLL; sh.lastLL = lastLL;
            }
            zj_mem_order();
        } else if (!pre && hbmTables) {                    // level 4: tables too large for LDS, one set per resident workgroup in HBM (zj_encode_multi_kernel)
            u32 const entries = (strategy >= 3 && ZJ_UNI(sh.windowLog) > 14u) ? (1u << hlog) + (1u << hlog) / 4u      // rows: u32 entries + a byte of tag each
                                                                                 : (1u << hlog) + (1u << clog);
            GRP_FOR(g, i, entries) hbmTables[i] = 0;
            zj_mem_order();
            g.sync();
            pf.mark(0);
            if (strategy == 2 && srcSize >= 64u && !(flags & ZE_FLAG_MULTI_SERIAL)) {     // level 4 (double-fast): the whole wave parses (zj_match_wavex.h), the frame as one block
                GRP_SERIAL(g) { sh.blkRep[0] = 1; sh.blkRep[1] = 4; }
                g.sync();
                ZEOut o; o.seqs = seqs; o.litOff = (u32*)(ws + ZE_WS_BODY); o.n = 0; o.lit = 0;
                u32 const lastLL = zx_block_dfast_wave(lds, o, src, srcSize, 0u, srcSize, hlog, clog, mls, hbmTables, hbmTables + (1u << hlog), sh.blkRep, sh.blkNextRep, !(flags & ZE_FLAG_MULTI_NOCARRY));
                GRP_SERIAL(g) { sh.nbSeq = o.n; sh.litSize = o.lit + lastLL; sh.lastLL = lastLL; }
            } else
            GRP_SERIAL(g) {
                ZEOut o; o.seqs = seqs; o.litOff = (u32*)(ws + ZE_WS_BODY); o.n = 0; o.lit = 0;
                ZEParams q; q.windowLog = sh.windowLog; q.chainLog = clog; q.hashLog = hlog; q.minMatch = mls; q.strategy = strategy; q.searchLog = sh.searchLog;
                u32 const lastLL = (strategy >= 3) ? ze_block_lazy(o, src, srcSize, q, hbmTables, hbmTables + (1u << hlog))
                                                   : ze_block_dfast<ZEEnt32>(o, src, srcSize, hlog, clog, mls, hbmTables, hbmTables + (1u << hlog));
                sh.nbSeq = o.n; sh.litSize = o.lit + lastLL; sh.lastLL = lastLL;
            }
            zj_mem_order();
        } else if (!pre) {
            u32 const entries = (1u << hlog) + (strategy == 2 ? (1u << clog) : 0u);
            {   u32* const w = (u32*)lds; u32 const words = (entries * (u32)sizeof(TIdx) + 3) / 4;
                GRP_FOR(g, i, words) w[i] = 0; }
            g.sync();
            pf.mark(0);
            GRP_SERIAL(g) {
                ZEOut o; o.seqs = seqs; o.litOff = (u32*)(ws + ZE_WS_BODY); o.n = 0; o.lit = 0;
                TIdx* const t = (TIdx*)lds;
                typedef typename ZEEntOf<TIdx>::E EntLds;
                u32 const lastLL = (strategy == 1) ? ze_block_fast<EntLds>(o, src, srcSize, hlog, mls, t)
                                                   : ze_block_dfast<EntLds>(o, src, srcSize, hlog, clog, mls, t, t + (1u << hlog));
                sh.nbSeq = o.n; sh.litSize = o.lit + lastLL; sh.lastLL = lastLL;
            }
            zj_mem_order();
        }
        g.sync();
        pf.mark(1);
        u32 const nbSeq = ZJ_UNI(sh.nbSeq), litSize = ZJ_UNI(sh.litSize), lastLL = ZJ_UNI(sh.lastLL);
        // ---- gather literals into HBM scratch + sequence codes (all lanes) ----
        {   const u32* const litOff = pre ? pre->litOff : (const u32*)(ws + ZE_WS_BODY);   // body scratch is free until the entropy stage
            for (u32 base = 0; base < nbSeq; base += 2u * (u32)G::W) {
              ZESeq sA, sB; u32 oA = 0, oB = 0;                           // two records per lane per pass: both requested before either is used
              sA.ll = sA.ml = sA.off = sA.pos = 0; sB = sA;
              {   u32 const iA = base + g.lane(), iB = iA + (u32)G::W;
                  if (iA < nbSeq) { sA = seqs[iA]; oA = litOff[iA]; }
                  if (iB < nbSeq) { sB = seqs[iB]; oB = litOff[iB]; } }
              for (u32 half = 0; half < 2u; half++) {
                u32 const i = base + half * (u32)G::W + g.lane();
                u32 ll = 0, o = 0, p = 0;
                if (i < nbSeq) {
                    ZESeq s = half ? sB : sA;
                    ll = s.ll; o = half ? oB : oA; p = s.pos;
                    if (ll <= 64) {                                            // short run: this lane, 8 bytes per load
                        u32 k = 0;
                        for (; k + 8 <= ll; k += 8) st64(litBuf + o + k, ld64(src + p + k));
                        u32 const r = ll - k;
                        if (r) {
                            u64 w = 0;
                            if (p + k + 8 <= srcSize) w = ld64(src + p + k);
                            else for (u32 j = 0; j < r; j++) w |= (u64)src[p + k + j] << (8 * j);
                            for (u32 j = 0; j < r; j++) litBuf[o + k + j] = (u8)(w >> (8 * j));
                        }
                    }
                    s.ll = ll | (ze_ll_code(ll) << 24);
                    s.ml = s.ml | (ze_ml_code(s.ml - 3) << 24);
                    s.off = s.off | (zj_hibit(s.off) << 24);
                    seqs[i] = s;
                    if (i == 0) { sh.edge[0] = s.ll >> 24; sh.edge[1] = s.off >> 24; sh.edge[2] = s.ml >> 24; }
                    if (i == nbSeq - 1) { sh.edge[3] = s.ll >> 24; sh.edge[4] = s.off >> 24; sh.edge[5] = s.ml >> 24; }
                }
                u64 m = grp_ballot(g, ll > 64);                                // long runs: the whole wave, one after the other
                while (m) {
                    u32 const k = (u32)__builtin_ctzll(m); m &= m - 1;
                    u32 const llk = grp_bcast(g, ll, k), ok = grp_bcast(g, o, k), pk = grp_bcast(g, p, k);
                    GRP_FOR(g, q, llk) litBuf[ok + q] = src[pk + q];
                }
              }
            }
            GRP_FOR(g, k, lastLL) litBuf[litSize - lastLL + k] = src[srcSize - lastLL + k];
            zj_mem_order();
            g.sync();
        }
        pf.mark(2);
        // ---- literals section (ZSTD_compressLiterals, first block: no previous table) ----
        u32 const strat = strategy;
        {   u32 const n = litSize;
            u32 const lhSize = 3 + (n >= 1024) + (n >= 16384);
            u32 const hufRep = (cd || ba) ? ZJ_UNI(sh.dictHufRep) : ZC_REPEAT_NONE;   // the dictionary's Huffman table is "the previous block's"; in a multi-block frame it IS the previous block's
            bool const single = (n < 256) || (hufRep == ZC_REPEAT_VALID && lhSize == 3);
            bool const preferRepeat = n <= 1024;                              // HUF_flags_preferRepeat (strategy < lazy)
            u32 const seg = (n + 3) / 4;
            u32 mode = 2;                                                      // 0 raw, 1 rle, 2 compressed (new table), 3 compressed (dictionary's table)
            if (n < (hufRep == ZC_REPEAT_VALID ? 6u : 64u)) mode = 0;           // ZSTD_minLiteralsToCompress (strategy <= 6)
            if (mode == 2) {
                bool const suspect = (nbSeq == 0) || (n / nbSeq >= 20);
                GRP_FOR(g, i, 1024) (&e.hist[0][0])[i] = 0;
                g.sync();
                if (suspect && n >= 40960) {                                   // huf_compress.c:1369-1383 sampling
                    GRP_FOR(g, i, 4096) atomicAdd(&e.hist[0][litBuf[i]], 1u);
                    GRP_FOR(g, i, 4096) atomicAdd(&e.hist[1][litBuf[n - 4096 + i]], 1u);
                    g.sync();
                    GRP_SERIAL(g) { u32 lb = 0, le = 0; for (u32 s = 0; s < 256; s++) { lb = zj_max(lb, e.hist[0][s]); le = zj_max(le, e.hist[1][s]); } sh.tmp[0] = (lb + le <= 68) ? 1 : 0; }
                    g.sync();
                    if (ZJ_UNI(sh.tmp[0])) mode = 0;
                    g.sync();
                    GRP_FOR(g, i, 512) (&e.hist[0][0])[i] = 0;
                    g.sync();
                }
            }
            // the block's tail reads sh.litMode (did this block make a new Huffman table?): every path has to leave its own answer there —
            // a workgroup's `sh` outlives the frame, and a stale "2" would register a table that does not exist as the next block's candidate
            if (mode != 2) { GRP_SERIAL(g) { sh.litMode = mode; } }
            if (mode == 2) {
                if (single) ze_hist_add(g, e.hist[0], litBuf, n);
                else {
                    for (u32 t = 0; t < 4; t++) ze_hist_add(g, e.hist[t], litBuf + t * seg, t < 3 ? seg : n - 3 * seg);
                }
                GRP_SERIAL(g) { sh.tmp[6] = 0; sh.tmp[7] = 0; sh.strBytes[0] = 0; sh.strBytes[1] = 0; sh.strBytes[2] = 0; sh.strBytes[3] = 0; }
                g.sync();
                GRP_FOR(g, s, 256) {
                    u32 const c = e.hist[0][s] + e.hist[1][s] + e.hist[2][s] + e.hist[3][s];
                    e.count[s] = c;
                    if (c) { atomicMax(&sh.tmp[6], s); atomicMax(&sh.tmp[7], c); }
                }
                g.sync();
                // HUF_compress_internal (huf_compress.c:1333-1434): the decisions on one lane, the table itself (when one is built) by the whole wave in between
                GRP_SERIAL(g) {
                    u32 const maxSV = sh.tmp[6], largest = sh.tmp[7];
                    u32 m = 2, rep = hufRep, build = 0;
                    sh.hufMaxSV = maxSV;
                    bool useOld = false;
                    if (preferRepeat && rep == ZC_REPEAT_VALID) useOld = true;    // valid table + small input: no statistics at all
                    else if (largest == n) m = 1;
                    else if (largest <= (n >> 7) + 4) m = 0;
                    else {
                        if (rep == ZC_REPEAT_CHECK) {                             // HUF_validateCTable
                            if (sh.dictHufMaxSV < maxSV) rep = ZC_REPEAT_NONE;
                            else for (u32 s = 0; s <= maxSV; s++) if (e.count[s] && !(sh.dictCodes[s] >> 16)) { rep = ZC_REPEAT_NONE; break; }
                        }
                        if (preferRepeat && rep != ZC_REPEAT_NONE) useOld = true;
                        else build = 1;
                    }
                    sh.tmp[0] = build; sh.tmp[1] = m; sh.tmp[2] = rep; sh.tmp[3] = useOld ? 1u : 0u;
                }
                g.sync();
                u32 huffLogBuilt = 0;
                if (ZJ_UNI(sh.tmp[0])) huffLogBuilt = ze_huf_build_wave(g, sh, e, ZJ_UNI(sh.tmp[6]), ze_fse_optimal_log(11, n, ZJ_UNI(sh.tmp[6]), 1));
                u32 hTab = 0;
                if (ZJ_UNI(sh.tmp[0])) {                                         // the table's description: weights, their tANS table and code by the wave
                    bool tg = false;
                    hTab = ze_huf_write_ctable(g, e, body + lhSize, ZJ_UNI(sh.tmp[6]), huffLogBuilt, capBody > lhSize ? capBody - lhSize : 0u, tg);
                    GRP_SERIAL(g) { if (tg) sh.tightHuf = 1; sh.ctDict[0] = 0; }  // (the weights' tANS table went through e.ct[0])
                    g.sync();
                }
                GRP_SERIAL(g) {
                    u32 const maxSV = sh.tmp[6];
                    u32 m = sh.tmp[1], h = hTab; u32 const rep = sh.tmp[2]; bool useOld = sh.tmp[3] != 0;
                    if (sh.tmp[0]) {
                        if (!h) m = 0;
                        else {
                            if (rep != ZC_REPEAT_NONE) {                      // is the dictionary's table at least as good?
                                u32 oldBits = 0, newBits = 0;
                                for (u32 s = 0; s <= maxSV; s++) { oldBits += e.count[s] * (sh.dictCodes[s] >> 16); newBits += e.count[s] * e.nbBits[s]; }
                                if ((oldBits >> 3) <= h + (newBits >> 3) || h + 12 >= n) useOld = true;
                            }
                            if (!useOld && h + 12 >= n) m = 0;
                        }
                    }
                    if (useOld) { m = 3; h = 0; }
                    sh.litMode = m; sh.hufHdr = h;
                }
                g.sync();
                mode = ZJ_UNI(sh.litMode);
                if (mode >= 2) {                                               // exact bits of every stream under the chosen table (all lanes)
                    u32 const streams = single ? 1u : 4u;
                    GRP_FOR(g, s, 256) {
                        u32 const nb = mode == 3 ? (sh.dictCodes[s] >> 16) : (u32)e.nbBits[s];
                        for (u32 t = 0; t < streams; t++) { u32 const c = e.hist[t][s]; if (c) atomicAdd(&sh.strBytes[t], c * nb); }
                    }
                    g.sync();
                }
                GRP_SERIAL(g) {                                                // sizes + ZSTD_compressLiterals' checks
                    u32 m = sh.litMode; u32 const h = sh.hufHdr, largest = sh.tmp[7];
                    if (m >= 2) {
                        u32 total = h + (single ? 0 : 6); bool tooBig = false;
                        // room left for the streams in the reference's call: HUF_compress4X_usingCTable_internal wants 17 bytes up front and
                        // its jump table, every stream 8 bytes to start and 8 of slack at its end (huf_compress.c:1080-1118, 1179-1213)
                        i64 room = (i64)capBody - (i64)lhSize - (i64)h; bool tg = capBody < lhSize + 1u;
                        if (!single) { if (room < 17) tg = true; room -= 6; }
                        for (u32 t = 0; t < (single ? 1u : 4u); t++) {
                            u32 const bits = sh.strBytes[t];
                            u32 const bytes = (bits + 1 + 7) >> 3;
                            if (room <= 8 || (i64)((bits + 1) >> 3) >= room - 8) tg = true;
                            room -= bytes;
                            sh.strBytes[t] = bytes; sh.strOff[t] = total; total += bytes;
                            if (bytes > 65535) tooBig = true;
                        }
                        if (tg) sh.tightHuf = 1;
                        if (!single && n < 12) tooBig = true;
                        if (tooBig || total >= n - 1 || total >= n - ((n >> 6) + 2)) m = 0;
                        else if (total == 1 && largest == n) m = 1;              // one byte out: rle if all literals are the same byte (n < 8 here)
                        else {
                            u8* const hp = body;
                            if (lhSize == 3) { u32 const lhc = m + ((u32)(!single) << 2) + (n << 4) + (total << 14); st16(hp, lhc & 0xFFFF); hp[2] = (u8)(lhc >> 16); }
                            else if (lhSize == 4) st32(hp, m + (2u << 2) + (n << 4) + (total << 18));
                            else { st32(hp, m + (3u << 2) + (n << 4) + (total << 22)); hp[4] = (u8)(total >> 10); }
                            if (!single) { u8* const jt = body + lhSize + h; st16(jt, sh.strBytes[0]); st16(jt + 2, sh.strBytes[1]); st16(jt + 4, sh.strBytes[2]); }
                            sh.litSecSize = lhSize + total;
                        }
                    }
                    sh.litMode = m;
                    if (m >= 1 && sh.tightHuf) sh.tight = 1;                     // (a Huffman attempt that ends in raw literals anyway took no detour)
                    if (m == 1 && capBody < lhSize + 1u) sh.tight = 1;           // ZSTD_compressLiterals' test before the attempt (zstd_compress_literals.c:163)
                }
                g.sync();
                pf.mark(3);
                mode = ZJ_UNI(sh.litMode);
            }
            if (mode == 0 && n + 1u + (n > 31) + (n > 4095) > capBody) { GRP_SERIAL(g) { sh.tight = 1; } }   // ZSTD_noCompressLiterals (zstd_compress_literals.c:46)
            if (mode >= 2) {
                u32 const streams = single ? 1u : 4u;
                u32* const codes = e.count;                                    // histogram is dead: code | nbBits << 16 per symbol
                if (mode == 3) { GRP_FOR(g, s, 256) codes[s] = sh.dictCodes[s]; }
                else { GRP_FOR(g, s, 256) codes[s] = (u32)e.val[s] | ((u32)e.nbBits[s] << 16); }
                g.sync();
                for (u32 t = 0; t < streams; t++) {
                    u32 const cnt = single ? n : (t < 3 ? seg : n - 3 * seg);
                    ze_huf_encode_wave(g, sh, codes, &e.hist[0][0], e.scount, body + lhSize + ZJ_UNI(sh.strOff[t]), litBuf + t * seg, cnt);
                }
            } else if (mode == 1) {
                GRP_SERIAL(g) {
                    u32 const fl = 1 + (n > 31) + (n > 4095);
                    if (fl == 1) body[0] = (u8)(1 + (n << 3)); else if (fl == 2) st16(body, 1 + (1u << 2) + (n << 4)); else { u32 const v = 1 + (3u << 2) + (n << 4); st16(body, v & 0xFFFF); body[2] = (u8)(v >> 16); }
                    body[fl] = litBuf[0]; sh.litSecSize = fl + 1;
                }
            } else {
                u32 const sz = ze_raw_literals(g, body, litBuf, n);
                GRP_SERIAL(g) { sh.litSecSize = sz; }
            }
            zj_mem_order();
            g.sync();
        }
        pf.mark(4);
        // ---- sequences section (zstd_compress.c:2940-3003) ----
        {   u32 const maxCSize = srcSize - ((srcSize >> 6) + 2);              // ZSTD_minGain
            u32 pos = ZJ_UNI(sh.litSecSize);
            GRP_SERIAL(g) {
                u8* op = body + pos;
                if ((i64)capBody - (i64)pos < 4) sh.tight = 1;               // "Can't fit seq hdr in output buf!" (zstd_compress.c:2939-2940)
                if (nbSeq < 128) *op++ = (u8)nbSeq;
                else if (nbSeq < 0x7F00) { op[0] = (u8)((nbSeq >> 8) + 0x80); op[1] = (u8)nbSeq; op += 2; }
                else { op[0] = 0xFF; st16(op + 1, nbSeq - 0x7F00); op += 3; }
                sh.tmp[1] = (u32)(op - body);
            }
            g.sync();
            pos = ZJ_UNI(sh.tmp[1]);
            bool ok = true;
            if (pos + (nbSeq ? 2u : 0u) >= maxCSize) ok = false;                // cannot win any more (a sequences section is >= 2 bytes): raw block
            if (ok && nbSeq) {
                u32 const seqHead = pos; pos += 1;
                u32* const cnt3 = &e.hist[0][0];                              // three code histograms from one pass over the records
                GRP_FOR(g, s, 192) cnt3[s] = 0;
                g.sync();
                GRP_FOR(g, i, nbSeq) { ZESeq const s = seqs[i]; atomicAdd(&cnt3[s.ll >> 24], 1u); atomicAdd(&cnt3[64 + (s.off >> 24)], 1u); atomicAdd(&cnt3[128 + (s.ml >> 24)], 1u); }
                g.sync();
                // ZSTD_buildSequencesStatistics: the three tables (LL, OF, ML) are independent — shares, table description and encoding table each
                // into scratch of its own in the idle tree area; what depends on the ORDER — where a description lands in the block, the capacity
                // tests there, "last count" — follows.  Each table is built by the whole wave (ze_tans_*: a symbol per lane, prefix sums and
                // reductions in LDS); the choice of the encoding type is a handful of scalar decisions every lane makes alike.  (Round 3 had one
                // lane build all three: 31-39 K cycles a frame; a lane per table cost 50 VGPRs and the co-residency with the match kernel,
                // profiles/r03/i_entropy_vgpr_ab.txt.)
                struct ZESeqScr { short norm[64]; u16 cumul[64]; u8 tableSymbol[512]; u8 ncount[128]; u32 need, h, type, first, max, most; };
                ZESeqScr* const scr = (ZESeqScr*)e.node;                      // 3 x 920 bytes of the 4 128 (the Huffman tree is done with)
                u32* const tscr = (u32*)e.rankBase;                           // ZE_TANS_SCR words (rankBase / rankCurr: done with as well)
                u32* const tbm = &e.hist[1][0];                               // rank bitmaps: hist[1..3] + count, 1 024 words (the literals are encoded)
                for (u32 t = 0; t < 3u; t++) {
                    ZESeqScr& q = scr[t];
                    u32* const scount = cnt3 + 64 * t;
                    u32 const maxSym = t == 0 ? 35u : (t == 1 ? 31u : 52u), fseLog = t == 1 ? 8u : 9u, defLog = t == 1 ? 5u : 6u;
                    const short* const defNorm = t == 0 ? ze_k_ll_defnorm : (t == 1 ? ze_k_of_defnorm : ze_k_ml_defnorm);
                    u32 const defMax = t == 0 ? 35u : (t == 1 ? 28u : 52u);
                    GRP_SERIAL(g) { q.max = 0; q.most = 0; }
                    g.sync();
                    GRP_FOR(g, s, maxSym + 1u) { u32 const c = scount[s]; if (c) { atomicMax(&q.max, s); atomicMax(&q.most, c); } }
                    g.sync();
                    u32 const max = ZJ_UNI(q.max), most = ZJ_UNI(q.most);
                    bool const defaultAllowed = (t != 1) || (max <= 28);
                    u32 const fseRep = cd ? ZJ_UNI(sh.dictFseRep[t]) : ZC_REPEAT_NONE;
                    u32 type;                                              // ZSTD_selectEncodingType, strategy < lazy; 3 = the dictionary's table (set_repeat)
                    if (most == nbSeq) type = (defaultAllowed && nbSeq <= 2) ? 0 : 1;
                    else if (defaultAllowed && fseRep == ZC_REPEAT_VALID && nbSeq < 1000) type = 3;
                    else if (strat < 4u) {
                        u32 const dynMin = ((1u << defLog) * (10 - strat)) >> 3;
                        type = (defaultAllowed && ((nbSeq < dynMin) || (most < (nbSeq >> (defLog - 1))))) ? 0 : 2;
                    } else {
                        // strategy >= lazy: the cheapest of predefined / new table by estimated cost (zstd_compress_sequences.c:196-222; no
                        // previous table here: these levels are served without a dictionary and one block per frame).  An impossible choice
                        // costs ERROR(GENERIC) = all ones there, the same here.
                        u32 hb = 0;                                                                        // ZSTD_NCountCost: the size of the description a new table would need
                        {   u32 const tl = ze_fse_optimal_log(fseLog, nbSeq, max, 2);
                            if (ze_tans_shares(g, q.norm, tl, scount, nbSeq, max, nbSeq >= 2048, tscr)) hb = ze_tans_describe(g, q.ncount, q.norm, max, tl, tscr); }
                        GRP_SERIAL(g) {
                            u64 const none = ~(u64)0;
                            u64 basicCost = none;
                            if (defaultAllowed) {                                                          // ZSTD_crossEntropyCost(defaultNorm, defaultNormLog, count, max)
                                u32 const shift = 8u - defLog; u64 c = 0;
                                for (u32 s = 0; s <= max; s++) { u32 const na = defNorm[s] != -1 ? (u32)defNorm[s] : 1u; c += (u64)scount[s] * ze_k_invprob[na << shift]; }
                                basicCost = c >> 8;
                            }
                            u32 cost = 0;                                                                  // ZSTD_entropyCost
                            for (u32 s = 0; s <= max; s++) { u32 nr = (256u * scount[s]) / nbSeq; if (scount[s] != 0 && nr == 0) nr = 1; cost += scount[s] * ze_k_invprob[nr]; }
                            u64 const compressedCost = ((hb ? (u64)hb : none) << 3) + (u64)(cost >> 8);
                            q.type = (basicCost <= none && basicCost <= compressedCost) ? 0u : 2u;         // (repeatCost = ERROR(GENERIC): basic wins ties against it)
                        }
                        g.sync();
                        type = ZJ_UNI(q.type);
                    }
                    u32 h = 0, need = 0;
                    if (type == 1) { GRP_SERIAL(g) { ZEFseCT& ct = e.ct[t]; ct.tableLog = 0; ct.state[0] = 0; ct.state[1] = 0; ct.deltaNbBits[max] = 0; ct.deltaFind[max] = 0; } h = 1; }
                    else if (type == 0) ze_tans_table(g, e.ct[t], defNorm, defMax, defLog, q.tableSymbol, tscr, tbm);
                    else if (type == 2) {
                        u32 const tableLog = ze_fse_optimal_log(fseLog, nbSeq, max, 2);
                        u32 const lastCode = ZJ_UNI(sh.edge[3 + t]);
                        u32 nbSeq1 = nbSeq;
                        if (ZJ_UNI(scount[lastCode]) > 1u) { g.sync(); GRP_SERIAL(g) { scount[lastCode]--; } nbSeq1--; }     // the last sequence's codes only set the final states
                        g.sync();
                        ze_tans_shares(g, q.norm, tableLog, scount, nbSeq1, max, nbSeq1 >= 2048, tscr);
                        h = ze_tans_describe(g, q.ncount, q.norm, max, tableLog, tscr, &need);
                        ze_tans_table(g, e.ct[t], q.norm, max, tableLog, q.tableSymbol, tscr, tbm);
                    }
                    GRP_SERIAL(g) { q.type = type; q.h = h; q.need = need; q.first = sh.edge[t]; }
                }
                g.sync();
                GRP_SERIAL(g) {
                    u32 at = pos;
                    for (u32 t = 0; t < 3; t++) {                             // LL, OF, ML in stream order
                        ZESeqScr const& q = scr[t];
                        u32 const type = q.type, h = q.h;
                        if (type == 1) { body[at] = (u8)q.first; if (at >= capBody) sh.tight = 1; }   // ZSTD_buildCTable, set_rle: "not enough space" (zstd_compress_sequences.c:255)
                        else if (type == 2) {
                            for (u32 k = 0; k < h; k++) body[at + k] = q.ncount[k];
                            if ((u64)at + q.need > capBody) sh.tight = 1;                              // FSE_writeNCount into what is left (zstd_compress_sequences.c:279-280)
                        }
                        sh.seqType[t] = type; sh.seqHdr[t] = h;
                        sh.tmp[3 + t] = (type == 3 && !sh.ctDict[t]) ? 1u : 0u;    // e.ct[t] has to be (re)loaded from the dictionary
                        sh.ctDict[t] = (type == 3) ? 1u : 0u;
                        if (t == 0) sh.seqLastCount = (type == 2) ? h : 0;
                        else if (type == 2) sh.seqLastCount = h;
                        at += h;
                    }
                    sh.tmp[1] = at;
                }
                g.sync();
                for (u32 t = 0; t < 3; t++) {
                    if (ZJ_UNI(sh.tmp[3 + t])) {
                        const u32* const from = (const u32*)&cd->fse[t]; u32* const to = (u32*)&e.ct[t];
                        GRP_FOR(g, i, (u32)(sizeof(ZEFseCT) / 4)) to[i] = from[i];
                    }
                }
                g.sync();
                pos = ZJ_UNI(sh.tmp[1]);
                pf.mark(5);
                // ---- ZSTD_encodeSequences_body (zstd_compress_sequences.c:291-382), restructured for the wave:
                //      per 64 sequences (last -> first) the records are staged into LDS, the per-symbol
                //      transforms are gathered by all lanes, the three tANS state chains (LL, OF, ML) advance on
                //      three lanes (one dependent LDS read per step), and every lane then ORs its sequence's
                //      bits {OF, ML, LL state bits; LL, ML, OF extra bits} at the prefix-summed bit offset ----
                bool over = false;
                ZEStageBits st; st.w = &e.hist[0][0]; st.dst = body + pos; st.flushedWords = 0; st.carryBits = 0;
                u32* const dn = (u32*)&e.node[0];                  // [3][64] deltaNbBits of each sequence's symbol
                i32* const df = (i32*)(dn + 192);                  // [3][64] deltaFindState
                u32* const ob = (u32*)(df + 192);                  // [3][64] state bits out: value | nbBits << 16
                u32* const lb = e.scount;                          // [64] bit counts (reversed order) -> prefix sums
                GRP_FOR(g, i, 1024) st.w[i] = 0;
                GRP_SERIAL(g) { body[seqHead] = (u8)((sh.seqType[0] << 6) + (sh.seqType[1] << 4) + (sh.seqType[2] << 2)); }
                g.sync();
                ZESeq qNext; qNext.ll = 0; qNext.ml = 0; qNext.off = 0; qNext.pos = 0;   // the batch after this one, requested a batch ahead
                {   u32 const c0 = zj_min(64u, nbSeq), l0 = nbSeq - c0;
                    if (G::W > 1 && g.lane() < c0) qNext = seqs[l0 + g.lane()]; }
                for (u32 hi = nbSeq; hi > 0 && !over; ) {
                    u32 const cnt = zj_min(64u, hi), lo = hi - cnt;
                    bool const firstBatch = (hi == nbSeq);
                    ZESeq const qCur = qNext;
                    if (G::W > 1 && lo > 0) { u32 const cn = zj_min(64u, lo), ln = lo - cn; if (g.lane() < cn) qNext = seqs[ln + g.lane()]; }
                    GRP_FOR(g, k, cnt) {
                        ZESeq const q = (G::W > 1) ? qCur : seqs[lo + k]; e.stage[k] = q;
                        u32 const cLL = q.ll >> 24, cOF = q.off >> 24, cML = q.ml >> 24;
                        dn[k] = e.ct[0].deltaNbBits[cLL]; df[k] = e.ct[0].deltaFind[cLL];
                        dn[64 + k] = e.ct[1].deltaNbBits[cOF]; df[64 + k] = e.ct[1].deltaFind[cOF];
                        dn[128 + k] = e.ct[2].deltaNbBits[cML]; df[128 + k] = e.ct[2].deltaFind[cML];
                    }
                    g.sync();
                    GRP_FOR(g, t, 3) {                             // the three state chains
                        const ZEFseCT& ct = e.ct[t];
                        u32 state = sh.tstate[t];
                        u32 k = cnt;
                        if (firstBatch) {                          // FSE_initCState2 with the last sequence's symbol
                            k--;
                            u32 const d = dn[64 * t + k];
                            u32 const nbBitsOut = (d + (1u << 15)) >> 16;
                            state = (nbBitsOut << 16) - d;
                            state = ct.state[(i32)(state >> nbBitsOut) + df[64 * t + k]];
                            ob[64 * t + k] = 0;
                        }
                        while (k >= 8) {                           // 8 steps per block: the symbols' transforms are fetched together,
                            u32 dnr[8]; i32 dfr[8]; u32 obr[8];     // so a step waits for one dependent LDS read (the state table), not four
#if ZJ_ON_GPU
#pragma unroll
#endif
                            for (u32 j = 0; j < 8; j++) { dnr[j] = dn[64 * t + k - 1 - j]; dfr[j] = df[64 * t + k - 1 - j]; }
#if ZJ_ON_GPU
#pragma unroll
#endif
                            for (u32 j = 0; j < 8; j++) {
                                u32 const nbBitsOut = (state + dnr[j]) >> 16;
                                obr[j] = (state & ((1u << nbBitsOut) - 1)) | (nbBitsOut << 16);
                                state = ct.state[(i32)(state >> nbBitsOut) + dfr[j]];
                            }
#if ZJ_ON_GPU
#pragma unroll
#endif
                            for (u32 j = 0; j < 8; j++) ob[64 * t + k - 1 - j] = obr[j];
                            k -= 8;
                        }
                        while (k-- > 0) {
                            u32 const nbBitsOut = (state + dn[64 * t + k]) >> 16;
                            ob[64 * t + k] = (state & ((1u << nbBitsOut) - 1)) | (nbBitsOut << 16);
                            state = ct.state[(i32)(state >> nbBitsOut) + df[64 * t + k]];
                        }
                        sh.tstate[t] = state;
                    }
                    g.sync();
                    GRP_FOR(g, k, cnt) {                           // bit count per sequence, in emission order r = cnt-1-k
                        ZESeq const q = e.stage[k];
                        u32 const c = (ob[k] >> 16) + (ob[64 + k] >> 16) + (ob[128 + k] >> 16)
                                    + ze_ll_bits_of(q.ll >> 24) + ze_ml_bits_of(q.ml >> 24) + (q.off >> 24);
                        lb[cnt - 1 - k] = c;
                    }
                    GRP_FOR(g, k, 64 - cnt) lb[cnt + k] = 0;
                    g.sync();
                    grp_scan_incl(g, lb, 64);
#if !ZJ_ON_GPU
                    g.sync();
#endif
                    GRP_FOR(g, k, cnt) {
                        ZESeq const q = e.stage[k];
                        u32 const r = cnt - 1 - k;
                        u32 const bitpos = st.carryBits + (r ? lb[r - 1] : 0);
                        u64 lo64 = 0, hi64 = 0; u32 nb = 0;
#define ZE_PUT(val, bits) do { u32 const b_ = (bits); u64 const v_ = (u64)(val) & (((u64)1 << b_) - 1); if (b_) { if (nb < 64) { lo64 |= v_ << nb; if (nb + b_ > 64) hi64 |= v_ >> (64 - nb); } else hi64 |= v_ << (nb - 64); nb += b_; } } while (0)
                        ZE_PUT(ob[64 + k] & 0xFFFF, ob[64 + k] >> 16);         // OF state
                        ZE_PUT(ob[128 + k] & 0xFFFF, ob[128 + k] >> 16);       // ML state
                        ZE_PUT(ob[k] & 0xFFFF, ob[k] >> 16);                   // LL state
                        ZE_PUT(ZE_LOW24(q.ll), ze_ll_bits_of(q.ll >> 24));
                        ZE_PUT(ZE_LOW24(q.ml) - 3, ze_ml_bits_of(q.ml >> 24));
                        ZE_PUT(ZE_LOW24(q.off), q.off >> 24);
#undef ZE_PUT
                        ze_or_bits(st.w, bitpos, lo64, hi64, nb);
                    }
                    g.sync();
                    u32 const batchBits = ZJ_UNI(lb[63]);
                    ze_stage_flush(g, sh, st, batchBits);
                    over = (pos + 4 * st.flushedWords) >= maxCSize;             // block will be emitted raw anyway
                    hi = lo;
                }
                if (!over) {                                       // FSE_flushCState x3 (ML, OF, LL) + end mark
                    GRP_SERIAL(g) {
                        u32 p0 = st.carryBits;
                        ze_or_bits(st.w, p0, sh.tstate[2] & ((1u << e.ct[2].tableLog) - 1), 0, e.ct[2].tableLog); p0 += e.ct[2].tableLog;
                        ze_or_bits(st.w, p0, sh.tstate[1] & ((1u << e.ct[1].tableLog) - 1), 0, e.ct[1].tableLog); p0 += e.ct[1].tableLog;
                        ze_or_bits(st.w, p0, sh.tstate[0] & ((1u << e.ct[0].tableLog) - 1), 0, e.ct[0].tableLog); p0 += e.ct[0].tableLog;
                        ze_or_bits(st.w, p0, 1, 0, 1);
                        sh.tmp[5] = p0 + 1 - st.carryBits;
                    }
                    g.sync();
                    ze_stage_flush(g, sh, st, ZJ_UNI(sh.tmp[5]));
                    {   // ZSTD_encodeSequences: more than 8 bytes to start, 8 bytes of slack behind the last whole byte (zstd_compress_sequences.c:300-302, 377-378)
                        i64 const room = (i64)capBody - (i64)pos; u32 const fullBytes = 4u * st.flushedWords + (st.carryBits >> 3);
                        if (room <= 8 || (i64)fullBytes >= room - 8) { GRP_SERIAL(g) { sh.tight = 1; } } }
                    u32 const bitSize = ze_stage_finish(g, st);
                    GRP_SERIAL(g) { sh.tmp[2] = bitSize; }
                } else { GRP_SERIAL(g) { sh.tmp[2] = 0xFFFFFFFFu; } }
                g.sync();
                pf.mark(6);
                {   u32 const bits = ZJ_UNI(sh.tmp[2]), lastCount = ZJ_UNI(sh.seqLastCount);
                    if (bits == 0xFFFFFFFFu) ok = false;
                    else { if (lastCount && (lastCount + bits) < 4) ok = false; pos += bits; } }
            }
            if (ok && pos < maxCSize) { compressed = true; cSize = pos; }
        }
    }
    zj_mem_order();
    g.sync();
    if (ba) {
        // ZSTD_compressBlock_internal's tail (zstd_compress.c:4422-4447) + the block header of ZSTD_compress_frameChunk (:4651-4661):
        // a block of one repeated byte becomes an RLE block unless it is the frame's first; only a block emitted compressed
        // confirms its repcodes and its Huffman table for the blocks that follow
        if (compressed && ZJ_UNI(sh.tight)) {                  // the reference ran out of room on the way: raw (or RLE) if that fits, else dstSize_tooSmall
            if (srcSize > capBody) return ZJ_ERR64(ZJ_E_DSTSIZE_TOO_SMALL);
            compressed = false;
        }
        u32 type = compressed ? 2u : 0u;
        if (srcSize >= 7 && !ba->isFirst && (compressed ? cSize : 0u) < 25u) {
            GRP_SERIAL(g) { sh.tmp[0] = 1; }
            g.sync();
            u32 const b0 = src0[0];
            GRP_FOR(g, i, srcSize) { if (src0[i] != b0) sh.tmp[0] = 0; }
            g.sync();
            if (ZJ_UNI(sh.tmp[0])) type = 1;
        }
        u32 const bodySize = type == 2 ? cSize : (type == 1 ? 1u : srcSize);
        if (dstCap < 3 + bodySize) return ZJ_ERR64(ZJ_E_DSTSIZE_TOO_SMALL);
        if (type == 2) {
            if (!direct) grp_copy_wide(g, dst + 3, body, cSize);
            if (ZJ_UNI(sh.litMode) == 2) {                     // a new Huffman table: the next block may repeat it (HUF_repeat_check)
                u32 const maxSV = ZJ_UNI(sh.hufMaxSV);
                GRP_FOR(g, s2, 256) sh.dictCodes[s2] = s2 <= maxSV ? ((u32)e.val[s2] | ((u32)e.nbBits[s2] << 16)) : 0u;
                GRP_SERIAL(g) { sh.dictHufRep = ZC_REPEAT_CHECK; sh.dictHufMaxSV = maxSV; }
            }
            GRP_SERIAL(g) { sh.blkRep[0] = sh.blkNextRep[0]; sh.blkRep[1] = sh.blkNextRep[1]; }
        } else if (type == 1) { GRP_SERIAL(g) { dst[3] = src0[0]; } }
        else grp_copy_wide(g, dst + 3, src0, srcSize);
        GRP_SERIAL(g) { u32 const bh = ba->lastBlock + (type << 1) + ((type == 2 ? cSize : srcSize) << 3); dst[0] = (u8)bh; dst[1] = (u8)(bh >> 8); dst[2] = (u8)(bh >> 16); }
        zj_mem_order();
        g.sync();
        return 3 + bodySize;
    }
    // ---- block header + placement ----
    if (compressed && ZJ_UNI(sh.tight)) compressed = false;   // the reference ran out of room on the way: a raw block if that fits, else dstSize_tooSmall
    u32 const bodySize = compressed ? cSize : srcSize;
    if (dstCap < hdr + 3 + bodySize + tail) return ZJ_ERR64(ZJ_E_DSTSIZE_TOO_SMALL);
    if (compressed) {
        if (!direct) grp_copy_wide(g, dst + hdr + 3, body, cSize);
        GRP_SERIAL(g) { u32 const bh = 1 + (2u << 1) + (cSize << 3); dst[hdr] = (u8)bh; dst[hdr + 1] = (u8)(bh >> 8); dst[hdr + 2] = (u8)(bh >> 16); }
    } else {
        grp_copy_wide(g, dst + hdr + 3, src0, srcSize);
        GRP_SERIAL(g) { u32 const bh = 1 + (srcSize << 3); dst[hdr] = (u8)bh; dst[hdr + 1] = (u8)(bh >> 8); dst[hdr + 2] = (u8)(bh >> 16); }
    }
    if (tail) {
        u64 const h = zj_xxh64(g, src0, srcSize);
        GRP_SERIAL(g) { st32(dst + hdr + 3 + bodySize, (u32)h); }
    }
    return hdr + 3 + bodySize + tail;
}

// LDS bytes the match finder needs for (level, srcSize); the entropy stage needs sizeof(ZEEntropy).
ZJ_HD u32 ze_lds_need(u32 level, u32 srcSize) {
    u32 w, c, h, st;
    if (srcSize <= (16u << 10)) { w = 14; c = 14; h = 15; st = (level == 3) ? 2 : 1; }
    else if (level == 1) { w = 17; c = 12; h = 13; st = 1; }
    else if (level == 2) { w = 17; c = 13; h = 15; st = 1; }
    else { w = 17; c = 15; h = 16; st = 2; }
    u32 const srcLog = (srcSize < 64u) ? 6u : zj_hibit(srcSize - 1) + 1;
    if (w > srcLog) w = srcLog;
    if (h > w + 1) h = w + 1;
    if (c > w) c = w;
    if (st == 2) { if (h > ZE_L3_HASHLOG) h = ZE_L3_HASHLOG; if (c > ZE_L3_CHAINLOG) c = ZE_L3_CHAINLOG; if (h > w + 1) h = w + 1; if (c > w) c = w; }
    u32 const entries = (1u << h) + (st == 2 ? (1u << c) : 0u);
    u32 const bytes = entries * (srcSize <= 65536u ? 2u : 4u);
    return bytes > (u32)sizeof(ZEEntropy) ? bytes : (u32)sizeof(ZEEntropy);
}

template <class G>
ZJ_DEV u64 ze_compress(const G& g, ZEncShared& sh, u8* lds, const u8* src, u32 srcSize, u8* dst, u32 dstCap, u32 level, u8* ws, ZjProf& pf, const ZEPre* pre = nullptr, u32 flags = 0,
                       const ZECDictDev* cd = nullptr, u32 ldsBytes = 0) {
    if (srcSize <= 65536u) return ze_compress_t<G, u16>(g, sh, lds, src, srcSize, dst, dstCap, level, ws, pf, pre, flags, cd, ldsBytes);
    return ze_compress_t<G, u32>(g, sh, lds, src, srcSize, dst, dstCap, level, ws, pf, pre, flags, cd, ldsBytes);
}

#include "zj_presplit.h"
// ---- multi-block frames: 128 KiB < srcSize <= ZE_MULTI_MAX (ZSTD_compress2 on a larger input: N/compress/zstd_compress.c:4591-4692) ----
// One wavefront walks the frame block by block — the reference's loop: size of the next block (ZSTD_optimalBlockSize: full blocks
// until the frame has saved 3 bytes, then the pre-split heuristic of zj_presplit.h), parse over the frame-wide tables, entropy stage
// with the previous compressed block's Huffman table as repeat candidate, raw / RLE / compressed block header.  Frames are
// byte-identical to the reference's at levels 1-3 with the level's own parameters.  The frame must fit its window (levels 1 / 2 / 3:
// 512 KiB / 1 MiB / 2 MiB), so no position ever leaves it; larger inputs are refused (201) and stay on the CPU path.
// `tables`: (1 << hashLog) + (1 << chainLog) u32 entries in HBM for this workgroup.
#define ZE_MULTI_MAX (2u << 20)
#define ZE_MULTI_TABLE_BYTES (((1u << 17) + (1u << 17)) * 4u)      /* level 3 frames > 256 KiB: 2^17 + 2^16 entries; level 4 at 128 KiB: 2 x 2^17 */
template <class G>
ZJ_DEV u64 ze_compress_multi(const G& g, ZEncShared& sh, u8* lds, const u8* src, u32 srcSize, u8* dst, u32 dstCap, u32 level, u8* ws, ZjProf& pf, u32 flags, u32* tables, u32 ldsBytes) {
    ZEParams const p = ze_params_of(ZE_LW_LEVEL(level), srcSize);
    if (srcSize <= ZE_BLOCK_MAX || srcSize > ZE_MULTI_MAX || srcSize > (1u << p.windowLog)) return ZJ_ERR64(201);
    u32 const tail = (flags & ZE_FLAG_CHECKSUM) ? 4u : 0u;
    bool const noFcs = (flags & ZE_FLAG_NO_FCS) != 0;
    u32 const hdr = noFcs ? 6u : 9u;                               // magic, descriptor, then the window byte or the 4-byte content size (single segment)
    if (dstCap < 18u) return ZJ_ERR64(ZJ_E_DSTSIZE_TOO_SMALL);     // ZSTD_writeFrameHeader wants ZSTD_FRAMEHEADERSIZE_MAX (zstd_compress.c:4711-4712)
    GRP_SERIAL(g) {
        st32(dst, 0xFD2FB528u);
        if (noFcs) { dst[4] = (u8)(tail ? 4u : 0u); dst[5] = (u8)((p.windowLog - 10u) << 3); }
        else { dst[4] = (u8)((1u << 5) + (2u << 6) + (tail ? 4u : 0u)); st32(dst + 5, srcSize); }
        sh.blkRep[0] = 1; sh.blkRep[1] = 4; sh.dictHufRep = ZC_REPEAT_NONE; sh.dictHufMaxSV = 0;
    }
    {   u32 const entries = (1u << p.hashLog) + (p.strategy == 2 ? (1u << p.chainLog) : 0u);
        GRP_FOR(g, i, entries) tables[i] = 0; }
    zj_mem_order();
    g.sync();
    u32 pos = hdr, at = 0, isFirst = 1; i64 savings = 0;
#if defined(ZX_PROFILE) && ZJ_ON_GPU
    u64 zxSplit = 0, zxBlocks = 0, zxT = __builtin_readcyclecounter();
#define ZX_FRAME_MARK(acc) do { u64 const t_ = __builtin_readcyclecounter(); acc += t_ - zxT; zxT = t_; } while (0)
#else
#define ZX_FRAME_MARK(acc) ((void)0)
#endif
    while (at < srcSize) {
        if (p.strategy == 2 && srcSize - at >= 131072u && savings >= 3) {       // double-fast: the chunk fingerprints, all lanes
            u32 const bs = zp_split_by_chunks_g(g, src + at, (u32*)lds);
            GRP_SERIAL(g) { sh.tmp[0] = bs; }
        } else GRP_SERIAL(g) { sh.tmp[0] = zp_block_size(src + at, srcSize - at, p.strategy, savings, (u32*)lds); }
        g.sync();
        ZX_FRAME_MARK(zxSplit);
        u32 const blockSize = ZJ_UNI(sh.tmp[0]);
        g.sync();
        ZEBlockArgs ba; ba.frameBase = src; ba.frameSize = srcSize; ba.paramSize = 0; ba.start = at; ba.isFirst = isFirst; ba.lastBlock = (at + blockSize == srcSize) ? 1u : 0u; ba.tables = tables; ba.serialParse = ((flags & ZE_FLAG_MULTI_SERIAL) ? 1u : ((flags & ZE_FLAG_MULTI_NOCARRY) ? 2u : 0u)) | ((flags & ZE_FLAG_MULTI_FAST_SERIAL) ? 4u : 0u);
        u64 const r = ze_compress_t<G, u32>(g, sh, lds, src + at, blockSize, dst + pos, dstCap - pos, level, ws, pf, nullptr, 0u, nullptr, ldsBytes, &ba);
        if (r > ZJ_ERR64(256)) return r;
        savings += (i64)blockSize - (i64)r;
        at += blockSize; pos += (u32)r; isFirst = 0;
        ZX_FRAME_MARK(zxBlocks);
    }
#if defined(ZX_PROFILE) && ZJ_ON_GPU
    if (blockIdx.x < 8u && threadIdx.x == 0) printf("zx frame wg %u: %u bytes, block sizing (pre-split) %llu kcycles, blocks (parse + entropy stage) %llu kcycles\n", blockIdx.x, srcSize, (unsigned long long)(zxSplit / 1000ull), (unsigned long long)(zxBlocks / 1000ull));
#endif
    if (dstCap < pos + tail) return ZJ_ERR64(ZJ_E_DSTSIZE_TOO_SMALL);
    if (tail) {
        u64 const h = zj_xxh64(g, src, srcSize);
        GRP_SERIAL(g) { st32(dst + pos, (u32)h); }
    }
    return pos + tail;
}

// ---- stream frames: what ZSTD_compressStream2 produces WITHOUT a pledged size — ZstdDirectBufferCompressingStream, ZstdOutputStream
// (N/jni_directbuffercompress_zstd.c:97-161; N/compress/zstd_compress.c:6103-6300 ZSTD_compressStream_generic, :4591-4692 ZSTD_compress_frameChunk) ----
// The caller hands over everything written so far (the stream is buffered until close(), at most the level's unknown-size window: 512 KiB / 1 MiB / 2 MiB at
// levels 1 / 2 / 3, so no position ever leaves the window) and where it flushed.  What differs from ze_compress_multi, rule by rule:
//   (i)   parameters of an UNKNOWN source size: the level's default row (level 3: window 21, chain 16, hash 17) whatever the total — a parameter size apart from
//         the frame size, which still clamps the matchers' loads;
//   (ii)  header without a content size: descriptor = checksum << 2, window byte (windowLog - 10) << 3.  A stream that was closed before anything else was called on
//         it (`knownEmpty`) is the exception: its first call is ZSTD_e_end, the size (0) is known: single segment, one-byte content size;
//   (iii) the input reaches ZSTD_compress_frameChunk in pieces of 128 KiB (the stream's input buffer), so ZSTD_optimalBlockSize sees what is left of the PIECE and
//         `savings` at a piece's start counts the frame header's bytes as produced;
//   (iv)  flush() ends the piece where the caller stands (flushAt[]: ascending byte counts; a flush with nothing buffered writes nothing), the 128 KiB pieces start
//         again behind it;
//   (v)   close() with nothing buffered (the total a multiple of 128 KiB, or a flush just before) writes an empty raw last block; otherwise the buffered rest is the
//         last piece and its last block the frame's last.
// `final` = 0: the caller flushed but did not close — no epilogue, the output is the frame's beginning up to the last flush (bytes past the last flush position are
// not consumed).  Re-running with more input and final = 1 reproduces those bytes and continues: every decision depends only on the bytes before it.
// Exact: tests/test_emu_stream.py against ZSTD_compressStream2 (oracle/ref.py compress_stream), tests/test_gpu_stream.py.
ZJ_HD u32 ze_stream_window_log(u32 level) { return level == 1u ? 19u : (level == 2u ? 20u : 21u); }      // N/compress/clevels.h:26-30 (the rows of a source above 256 KiB)
template <class G>
ZJ_DEV u64 ze_compress_stream(const G& g, ZEncShared& sh, u8* lds, const u8* src, u32 srcSize, u8* dst, u32 dstCap, u32 level, u8* ws, ZjProf& pf, u32 flags, u32* tables, u32 ldsBytes,
                              const u32* flushAt, u32 nFlush, u32 final, u32 knownEmpty) {
    u32 const lv = ZE_LW_LEVEL(level);
    u32 const wlog = ze_stream_window_log(lv);
    if (lv < 1u || lv > 3u || srcSize > (1u << wlog) || srcSize > ZE_MULTI_MAX) return ZJ_ERR64(201);
    u32 const paramSize = srcSize > (256u << 10) ? srcSize : (256u << 10) + 1u;           // any size above 256 KiB selects the row; hashLog / chainLog do not depend on it further
    ZEParams const p = ze_params_of(lv, paramSize);
    u32 const tail = (flags & ZE_FLAG_CHECKSUM) ? 4u : 0u;
    if (dstCap < 18u) return ZJ_ERR64(ZJ_E_DSTSIZE_TOO_SMALL);
    u32 total = srcSize;                                                                  // what this call consumes: everything when closing, up to the last flush otherwise
    if (!final) { total = 0; for (u32 i = 0; i < nFlush; i++) if (flushAt[i] <= srcSize && flushAt[i] > total) total = flushAt[i]; }
    bool const emptyKnown = final && knownEmpty && srcSize == 0u;
    GRP_SERIAL(g) {
        st32(dst, 0xFD2FB528u);
        if (emptyKnown) { dst[4] = (u8)(0x20u + (tail ? 4u : 0u)); dst[5] = 0; }
        else { dst[4] = (u8)(tail ? 4u : 0u); dst[5] = (u8)((wlog - 10u) << 3); }
        sh.blkRep[0] = 1; sh.blkRep[1] = 4; sh.dictHufRep = ZC_REPEAT_NONE; sh.dictHufMaxSV = 0;
    }
    {   u32 const entries = (1u << p.hashLog) + (p.strategy == 2 ? (1u << p.chainLog) : 0u);
        GRP_FOR(g, i, entries) tables[i] = 0; }
    zj_mem_order();
    g.sync();
    u32 pos = 6, isFirst = 1, fi = 0; bool lastSeen = false;
    if (total == 0u && !final) return 0;                                                  // nothing was flushed yet: the header goes out with the first block
    for (u32 chunk = 0, seg = 0; chunk < total; ) {
        while (fi < nFlush && flushAt[fi] <= seg) fi++;                                   // (flushes with nothing buffered)
        bool const haveFlush = fi < nFlush && flushAt[fi] <= total;
        u32 const segEnd = haveFlush ? flushAt[fi] : total;                               // the next flush() (or the end of what was written)
        u32 const chunkEnd = chunk + 131072u < segEnd ? chunk + 131072u : segEnd;
        bool const flushed = haveFlush && chunkEnd == segEnd;                             // this piece ends where the caller flushed
        bool const endChunk = final && chunkEnd == total && (chunkEnd - chunk) != 131072u && !flushed;     // still buffered at ZSTD_e_end: the last frame chunk
        i64 savings = (i64)chunk - (i64)pos;                                              // consumedSrcSize - producedCSize, the header's bytes included
        for (u32 at = chunk; at < chunkEnd; ) {
            if (p.strategy == 2 && chunkEnd - at >= 131072u && savings >= 3) {            // double-fast: the chunk fingerprints, all lanes
                u32 const bs = zp_split_by_chunks_g(g, src + at, (u32*)lds);
                GRP_SERIAL(g) { sh.tmp[0] = bs; }
            } else GRP_SERIAL(g) { sh.tmp[0] = zp_block_size(src + at, chunkEnd - at, p.strategy, savings, (u32*)lds); }
            g.sync();
            u32 const blockSize = ZJ_UNI(sh.tmp[0]);
            g.sync();
            ZEBlockArgs ba; ba.frameBase = src; ba.frameSize = srcSize; ba.paramSize = paramSize; ba.start = at; ba.isFirst = isFirst; ba.lastBlock = (endChunk && at + blockSize == chunkEnd) ? 1u : 0u; ba.tables = tables;
            ba.serialParse = ((flags & ZE_FLAG_MULTI_SERIAL) ? 1u : ((flags & ZE_FLAG_MULTI_NOCARRY) ? 2u : 0u)) | ((flags & ZE_FLAG_MULTI_FAST_SERIAL) ? 4u : 0u);
            u64 const r = ze_compress_t<G, u32>(g, sh, lds, src + at, blockSize, dst + pos, dstCap - pos, level, ws, pf, nullptr, 0u, nullptr, ldsBytes, &ba);
            if (r > ZJ_ERR64(256)) return r;
            lastSeen = ba.lastBlock != 0;
            savings += (i64)blockSize - (i64)r;
            at += blockSize; pos += (u32)r; isFirst = 0;
        }
        chunk = chunkEnd; if (chunkEnd == segEnd) seg = segEnd;
    }
    if (!final) return pos;
    if (dstCap < pos + 3u + tail) return ZJ_ERR64(ZJ_E_DSTSIZE_TOO_SMALL);
    if (!lastSeen) { GRP_SERIAL(g) { dst[pos] = 1; dst[pos + 1] = 0; dst[pos + 2] = 0; } pos += 3; }      // ZSTD_writeEpilogue: an empty raw last block
    if (tail) {
        u64 const h = zj_xxh64(g, src, srcSize);
        GRP_SERIAL(g) { st32(dst + pos, (u32)h); }
    }
    return pos + tail;
}

// ---- multi-block frames, PIPELINED (round 6): a parse wave one block ahead of an entropy wave ------------------------------------------------------------
// ze_compress_multi is a chain per frame — size of the next block, parse, entropy stage, block by block on one wave: 8 x (16 + 11) ms for a 1 MiB frame, and a
// batch of a thousand such frames (BASELINE config 1) leaves three quarters of the device idle while every frame waits for its own chain.  What block b + 1's
// PARSE needs from block b's ENTROPY STAGE is little (N/compress/zstd_compress.c:4383-4448, :4591-4692): whether b was emitted compressed — only then are its
// repcodes the next block's (ZSTD_blockState_confirmRepcodesAndEntropyTables, :4436-4439) — and whether the frame has saved 3 bytes yet (ZSTD_optimalBlockSize,
// :4552-4581: only then is the pre-splitter asked).  Both follow WITH CERTAINTY from an upper bound on b's compressed size that the parse can compute itself:
//   U(b) = 3 + (litSize + 3) + 3 + 1 + 150 + ceil((26 nbSeq + extra bits + 26 + 15) / 8) + 8
// — raw literals are the literal stage's worst case (ZSTD_compressLiterals falls back to them), 9 + 8 + 9 bits the widest tANS codes of a sequence, 150 bytes three
// table descriptions.  U(b) < blockSize - minGain makes the block compressed whatever the entropy stage finds (ZSTD_entropyCompressSeqStore :3019-3035), provided
// the destination has room to spare (no "tight" detour) and the sequences carry >= 256 extra bits (so the block is neither the "< 4 bytes" oddity of
// ZSTD_entropyCompressSeqStore_internal nor short enough — < 25 bytes — to be turned into an RLE block).  Where the bound says nothing (data that does not
// compress, tiny blocks, tight destinations) the parse wave WAITS for the entropy wave's answer, as the one-wave loop does for every block.
// The two waves share a workgroup: records and literal offsets go through the workgroup's two scratch slots in HBM (block b in slot b & 1), block descriptions,
// results and the two counters through LDS; fences are workgroup-scope (no L2 write-back).  The entropy wave CHECKS every assumption the parse wave made about
// a block when it has that block's real result (type and size against U): a wrong one fails the frame (error 1) instead of writing other bytes.
// Exact: tests/test_emu_multiblock.py (the emulation runs the same two roles, the entropy role stepping whenever the parse role waits).
struct ZEPipeBlk { u32 start, size, nbSeq, litSize, lastLL, isFirst, lastBlock, assumePrev, prevU; };      // (nbSeq, litSize, lastLL: ZEPre::meta)
struct ZEPipe {
    u32 ready;                      // blocks the parse wave has published
    u32 done;                       // blocks the entropy wave has finished
    u32 err;                        // the entropy wave gave the frame up (the parse wave stops)
    u32 resType[2], resSize[2];     // block b's result at [b & 1]: block type (0 raw, 1 RLE, 2 compressed), bytes written (3-byte header included)
    ZEPipeBlk blk[2];               // block b's description at [b & 1]
    u32 sum;                        // scratch of the parse wave (extra-bit total)
};
#if ZJ_ON_GPU
ZJ_DEV u32 ze_pipe_load(const u32* p) { return *(const volatile u32*)p; }
ZJ_DEV void ze_pipe_add(u32* p, u32 v) { atomicAdd(p, v); }
#else
static inline u32 ze_pipe_load(const u32* p) { return *p; }
static inline void ze_pipe_add(u32* p, u32 v) { *p += v; }
#endif
// the entropy wave's state between blocks
struct ZEPipeE { u32 b, pos, finished; u64 result; };
template <class G>
ZJ_DEV void ze_pipe_entropy_init(const G& g, ZEncShared& sh, ZEPipeE& st, u8* dst, u32 dstCap, u32 srcSize, u32 level, u32 flags) {
    ZEParams const p = ze_params_of(ZE_LW_LEVEL(level), srcSize);
    bool const noFcs = (flags & ZE_FLAG_NO_FCS) != 0;
    st.b = 0; st.pos = noFcs ? 6u : 9u; st.finished = 0; st.result = 0;
    if (dstCap < 18u) { st.finished = 1; st.result = ZJ_ERR64(ZJ_E_DSTSIZE_TOO_SMALL); return; }
    GRP_SERIAL(g) {
        u32 const tail = (flags & ZE_FLAG_CHECKSUM) ? 4u : 0u;
        st32(dst, 0xFD2FB528u);
        if (noFcs) { dst[4] = (u8)(tail ? 4u : 0u); dst[5] = (u8)((p.windowLog - 10u) << 3); }
        else { dst[4] = (u8)((1u << 5) + (2u << 6) + (tail ? 4u : 0u)); st32(dst + 5, srcSize); }
        sh.dictHufRep = ZC_REPEAT_NONE; sh.dictHufMaxSV = 0; sh.blkRep[0] = 1; sh.blkRep[1] = 4; sh.blkNextRep[0] = 1; sh.blkNextRep[1] = 4;
    }
    g.sync();
}
// one block: the parse wave has published it (pipe.ready > st.b)
template <class G>
ZJ_DEV void ze_pipe_entropy_step(const G& g, ZEncShared& sh, u8* lds, ZEPipe& pipe, ZEPipeE& st, const u8* src, u32 srcSize, u8* dst, u32 dstCap, u32 level, u32 flags,
                                 u8* ws0, u8* ws1, ZjProf& pf, u32 ldsBytes) {
    u32 const b = st.b;
    ZEPipeBlk* const kb = &pipe.blk[b & 1u];
    u32 const start = ZJ_UNI(kb->start), size = ZJ_UNI(kb->size), lastBlock = ZJ_UNI(kb->lastBlock);
    if (b >= 1u && ZJ_UNI(kb->assumePrev)) {              // what the parse wave took for granted about block b - 1 when it parsed this one
        if (ZJ_UNI(pipe.resType[(b - 1u) & 1u]) != 2u || ZJ_UNI(pipe.resSize[(b - 1u) & 1u]) > ZJ_UNI(kb->prevU)) {
            GRP_SERIAL(g) { pipe.err = 1; } st.finished = 1; st.result = ZJ_ERR64(ZJ_E_GENERIC); zj_mem_order(); g.sync(); return; }
    }
    u8* const ws = (b & 1u) ? ws1 : ws0;
    ZEPre pre; pre.seqs = (ZESeq*)(ws + ZE_WS_SEQ); pre.litOff = (const u32*)(ws + ZE_WS_BODY); pre.meta = &kb->nbSeq;
    ZEBlockArgs ba; ba.frameBase = src; ba.frameSize = srcSize; ba.paramSize = 0; ba.start = start; ba.isFirst = ZJ_UNI(kb->isFirst); ba.lastBlock = lastBlock; ba.tables = nullptr; ba.serialParse = 0;
    u64 const r = ze_compress_t<G, u32>(g, sh, lds, src + start, size, dst + st.pos, dstCap - st.pos, level, ws, pf, &pre, flags, nullptr, ldsBytes, &ba);
    if (r > ZJ_ERR64(256)) { GRP_SERIAL(g) { pipe.err = 1; } st.finished = 1; st.result = r; zj_mem_order(); g.sync(); return; }
    zj_mem_order();
    u32 const type = (ZJ_UNI((u32)dst[st.pos]) >> 1) & 3u;
    GRP_SERIAL(g) { pipe.resType[b & 1u] = type; pipe.resSize[b & 1u] = (u32)r; }
    zj_mem_order(); g.sync();
    GRP_SERIAL(g) { pipe.done = b + 1u; }
    zj_mem_order(); g.sync();
    st.pos += (u32)r; st.b = b + 1u;
    if (lastBlock) {
        u32 const tail = (flags & ZE_FLAG_CHECKSUM) ? 4u : 0u;
        st.finished = 1;
        if (dstCap < st.pos + tail) { st.result = ZJ_ERR64(ZJ_E_DSTSIZE_TOO_SMALL); return; }
        if (tail) { u64 const h = zj_xxh64(g, src, srcSize); GRP_SERIAL(g) { st32(dst + st.pos, (u32)h); } }
        st.result = st.pos + tail;
    }
}
// The parse wave, the whole frame.  wait(k): returns once pipe.done >= k (or the entropy wave has given up) — a poll on the GPU; in the emulation it steps the entropy
// role.  ldsP: >= sizeof(ZXLds) and >= 2 064 bytes (the pre-splitter's histograms).
template <class G, class WaitFn>
ZJ_DEV void ze_pipe_parse_role(const G& g, ZEncShared& sh, u8* lds, ZEPipe& pipe, const u8* src, u32 srcSize, u32 dstCap, u32 level, u32 flags, u32* tables, u8* ws0, u8* ws1, WaitFn wait) {
    ZEParams const p = ze_params_of(ZE_LW_LEVEL(level), srcSize);
    u32 const hdr = (flags & ZE_FLAG_NO_FCS) ? 6u : 9u;
    GRP_SERIAL(g) { ze_params(sh, level, srcSize); sh.blkRep[0] = 1; sh.blkRep[1] = 4; sh.blkNextRep[0] = 1; sh.blkNextRep[1] = 4; sh.err = 0; }
    {   u32 const entries = (1u << p.hashLog) + (p.strategy == 2 ? (1u << p.chainLog) : 0u);
        GRP_FOR(g, i, entries) tables[i] = 0; }
    zj_mem_order();
    g.sync();
    u32 at = 0, b = 0, folded = 0;                 // folded: blocks whose real results are in savExact / posExact
    i64 savExact = 0; u64 posExact = hdr;
    u32 sizeOf[2] = {0, 0}, uOf[2] = {0, 0}; bool prevCertain = false;
    u32 const serial = ((flags & ZE_FLAG_MULTI_SERIAL) ? 1u : ((flags & ZE_FLAG_MULTI_NOCARRY) ? 2u : 0u)) | ((flags & ZE_FLAG_MULTI_FAST_SERIAL) ? 4u : 0u);
#if defined(ZE_PIPE_DEBUG) && ZJ_ON_GPU
    u64 dbgWait = 0, dbgParse = 0, dbgT; u32 dbgAssume = 0, dbgAsk = 0;
#define ZE_PD_T0() (dbgT = wall_clock64())
#define ZE_PD_ADD(acc) (acc += wall_clock64() - dbgT)
#else
#define ZE_PD_T0() ((void)0)
#define ZE_PD_ADD(acc) ((void)0)
#endif
    while (at < srcSize) {
        ZE_PD_T0();
        if (b >= 2u) wait(b - 1u);                 // the slot of block b (block b - 2's) is free, block b - 2's result is in
        ZE_PD_ADD(dbgWait);
        if (ZJ_UNI(ze_pipe_load(&pipe.err))) return;
        while (folded + 1u < b && folded < ZJ_UNI(ze_pipe_load(&pipe.done))) { u32 const r = ZJ_UNI(ze_pipe_load(&pipe.resSize[folded & 1u])); savExact += (i64)sizeOf[folded & 1u] - (i64)r; posExact += r; folded++; }
        bool prevCompressed = false, assume = false;
        if (b >= 1u) {
            bool const asksSavings = srcSize - at >= 131072u;                    // (zp_block_size looks at `savings` for a full block only)
            i64 const savLow = savExact + (i64)sizeOf[(b - 1u) & 1u] - (i64)uOf[(b - 1u) & 1u];
            if (prevCertain && folded + 1u == b && (!asksSavings || savLow >= 3)) { assume = true; prevCompressed = true; }
            else {
                ZE_PD_T0();
                wait(b);
                ZE_PD_ADD(dbgWait);
                if (ZJ_UNI(ze_pipe_load(&pipe.err))) return;
                while (folded < b) { u32 const r = ZJ_UNI(ze_pipe_load(&pipe.resSize[folded & 1u])); savExact += (i64)sizeOf[folded & 1u] - (i64)r; posExact += r; folded++; }
                prevCompressed = ZJ_UNI(ze_pipe_load(&pipe.resType[(b - 1u) & 1u])) == 2u;
            }
        }
        i64 const savings = assume ? 3 : savExact;                              // (assumed: the bound says >= 3 where anything looks at it)
#if defined(ZE_PIPE_DEBUG) && ZJ_ON_GPU
        if (b >= 1u) { if (assume) dbgAssume++; else dbgAsk++; }
        ZE_PD_T0();
#endif
        if (prevCompressed) { GRP_SERIAL(g) { sh.blkRep[0] = sh.blkNextRep[0]; sh.blkRep[1] = sh.blkNextRep[1]; } g.sync(); }
        if (p.strategy == 2 && srcSize - at >= 131072u && savings >= 3) {
            u32 const bs = zp_split_by_chunks_g(g, src + at, (u32*)lds);
            GRP_SERIAL(g) { sh.tmp[0] = bs; }
        } else GRP_SERIAL(g) { sh.tmp[0] = zp_block_size(src + at, srcSize - at, p.strategy, savings, (u32*)lds); }
        g.sync();
        u32 const blockSize = ZJ_UNI(sh.tmp[0]);
        g.sync();
        u8* const ws = (b & 1u) ? ws1 : ws0;
        ZEOut o; o.seqs = (ZESeq*)(ws + ZE_WS_SEQ); o.litOff = (u32*)(ws + ZE_WS_BODY); o.n = 0; o.lit = 0;
        u32 lastLL = blockSize;
        if (blockSize >= 7u) {                                                 // ZSTD_buildSeqStore: MIN_CBLOCK_SIZE + 3 + 1 + 1 (ze_compress_t)
            u32 const strategy = p.strategy, hlog = p.hashLog, clog = p.chainLog, mls = p.minMatch;
#if defined(ZJ_TUNING_KERNELS) || !ZJ_ON_GPU
            if (strategy == 1 && serial != 1u && !(serial & 4u)) lastLL = zx_block_fast_wave(lds, o, src, srcSize, at, at + blockSize, hlog, mls, tables, sh.blkRep, sh.blkNextRep, (serial & 3u) == 0u);
            else
#endif
            if (strategy == 2 && (serial & 3u) != 1u) lastLL = zx_block_dfast_wave(lds, o, src, srcSize, at, at + blockSize, hlog, clog, mls, tables, tables + (1u << hlog), sh.blkRep, sh.blkNextRep, (serial & 3u) == 0u);
            else {
                GRP_SERIAL(g) {
                    u32 const l2 = (strategy == 1) ? ze_block_fast_x<ZEEnt32>(o, src, at, at + blockSize, hlog, mls, tables, sh.blkRep, sh.blkNextRep)
                                                   : ze_block_dfast_x<ZEEnt32>(o, src, at, at + blockSize, hlog, clog, mls, tables, tables + (1u << hlog), sh.blkRep, sh.blkNextRep);
                    sh.tmp[1] = l2; sh.tmp[2] = o.n; sh.tmp[3] = o.lit;
                }
                g.sync();
                lastLL = ZJ_UNI(sh.tmp[1]); o.n = ZJ_UNI(sh.tmp[2]); o.lit = ZJ_UNI(sh.tmp[3]);
            }
            zj_mem_order();
        }
        g.sync();
        u32 const nbSeq = blockSize >= 7u ? o.n : 0u, litSize = blockSize >= 7u ? o.lit + lastLL : blockSize;
        // the bound: extra bits of the block's sequences (the records as the parse left them: raw ll, ml, offBase)
        GRP_SERIAL(g) { pipe.sum = 0; }
        g.sync();
        {   u32 acc = 0;
            GRP_FOR(g, i, nbSeq) { ZESeq const q = o.seqs[i]; acc += ze_ll_bits_of(ze_ll_code(q.ll)) + ze_ml_bits_of(ze_ml_code(q.ml - 3u)) + zj_hibit(q.off); }
            if (acc) ze_pipe_add(&pipe.sum, acc); }
        g.sync();
        u32 const extra = ZJ_UNI(pipe.sum);
        u64 const U64 = 3ull + litSize + 3ull + 3ull + 1ull + 150ull + ((26ull * nbSeq + extra + 26ull + 15ull) >> 3) + 8ull;
        u32 const U = U64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (u32)U64;
        u64 posHigh = posExact;
        for (u32 j = folded; j < b; j++) posHigh += zj_min(uOf[j & 1u], sizeOf[j & 1u] + 3u);
        bool const certain = blockSize >= 7u && nbSeq >= 1u && extra >= 256u && U < blockSize - ((blockSize >> 6) + 2u)
                             && (u64)dstCap >= posHigh + blockSize + 3u + 1024u;
        GRP_SERIAL(g) {
            ZEPipeBlk k; k.start = at; k.size = blockSize; k.nbSeq = nbSeq; k.litSize = litSize; k.lastLL = lastLL; k.isFirst = b == 0u ? 1u : 0u;
            k.lastBlock = (at + blockSize == srcSize) ? 1u : 0u; k.assumePrev = assume ? 1u : 0u; k.prevU = b ? uOf[(b - 1u) & 1u] : 0u;
            pipe.blk[b & 1u] = k;
        }
        zj_mem_order(); g.sync();
        GRP_SERIAL(g) { pipe.ready = b + 1u; }
        zj_mem_order(); g.sync();
        sizeOf[b & 1u] = blockSize; uOf[b & 1u] = U; prevCertain = certain;
        at += blockSize; b++;
        ZE_PD_ADD(dbgParse);
    }
#if defined(ZE_PIPE_DEBUG) && ZJ_ON_GPU
    if (blockIdx.x < 4u && (threadIdx.x & 63u) == 0) printf("pipe P wg %u: %u blocks, assumed %u asked %u, waiting %llu us, parsing %llu us\n", blockIdx.x, b, dbgAssume, dbgAsk, (unsigned long long)(dbgWait / 100ull), (unsigned long long)(dbgParse / 100ull));
#endif
}

#if !ZJ_ON_GPU
// the same two roles one after the other (tests/emu): the entropy role steps whenever the parse role waits for it, and to the end when the parse role is done
template <class G>
static u64 ze_compress_multi_pipe_serial(const G& g, ZEncShared& shP, ZEncShared& shE, u8* ldsP, u8* ldsE, ZEPipe& pipe, const u8* src, u32 srcSize, u8* dst, u32 dstCap, u32 level,
                                         u8* ws0, u8* ws1, ZjProf& pf, u32 flags, u32* tables, u32 ldsBytes) {
    ZEParams const p = ze_params_of(ZE_LW_LEVEL(level), srcSize);
    if (srcSize <= ZE_BLOCK_MAX || srcSize > ZE_MULTI_MAX || srcSize > (1u << p.windowLog)) return ZJ_ERR64(201);
    pipe.ready = 0; pipe.done = 0; pipe.err = 0;
    ZEPipeE st;
    ze_pipe_entropy_init(g, shE, st, dst, dstCap, srcSize, level, flags);
    if (st.finished) return st.result;
    auto step = [&]() { ze_pipe_entropy_step(g, shE, ldsE, pipe, st, src, srcSize, dst, dstCap, level, flags, ws0, ws1, pf, ldsBytes); };
    ze_pipe_parse_role(g, shP, ldsP, pipe, src, srcSize, dstCap, level, flags, tables, ws0, ws1,
                       [&](u32 k) { while (pipe.done < k && !pipe.err && !st.finished && pipe.ready > st.b) step(); });
    while (!st.finished && pipe.ready > st.b) step();
    return st.finished ? st.result : ZJ_ERR64(ZJ_E_GENERIC);
}
#endif

// Per-frame HBM scratch of the lane-per-frame match finder: sequence records then literal offsets.
#define ZE_FRAME_MAXSEQ(maxSrc) (((maxSrc) / 4u) + 16u)
#define ZE_FRAME_STRIDE(maxSrc) (ZE_FRAME_MAXSEQ(maxSrc) * 20u)

// Plain-loop lane match finder (tiny frames; the round-synchronous machines of zj_match_lane.h do the rest).
ZJ_DEV void ze_match_lane_serial(const u8* src, u32 srcSize, u32 level, u8* table, u8* fscratch, u32 maxSrc, u32* meta) {
    ZEOut o; o.seqs = (ZESeq*)fscratch; o.litOff = (u32*)(fscratch + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); o.n = 0; o.lit = 0;
    u32 lastLL = srcSize;
    if (srcSize >= 7) {
        ZEParams const p = ze_params_of(level, srcSize);
        // fast: plain u16 entries (one probe per position; tags cost more in table sectors than they save);
        // double-fast: tagged 4-byte entries (two probes per position, most candidates rejected by tag)
        if (p.strategy == 1) lastLL = ze_block_fast<ZEEnt16>(o, src, srcSize, p.hashLog, p.minMatch, (u16*)table);
        else { u32* const t = (u32*)table; lastLL = ze_block_dfast<ZEEntTag>(o, src, srcSize, p.hashLog, p.chainLog, p.minMatch, t, t + (1u << p.hashLog)); }
    }
    meta[0] = o.n; meta[1] = o.lit + lastLL; meta[2] = lastLL;
}

#include "zj_match_lane.h"

// One frame through the lane machinery, start to finish (emulation and single-frame callers; the kernel
// interleaves 64 of these per wavefront, see zj_enc_match_kernel).  Output: records + meta {nbSeq, litSize, lastLL}.
#include "zj_need.h"
template <class M>
ZJ_DEV void ze_match_lane_t(const u8* src, u32 srcSize, u32 level, u8* table, u8* fscratch, u32 maxSrc, u32* meta, const u8* flags = nullptr) {
    M m; m.init(src, srcSize, ze_params_of(level, srcSize), table, fscratch, maxSrc, flags);
    for (u32 r = 0; m.st != ZL_DONE; r++) m.round(M::phase_of(r));
    meta[0] = m.o.n; meta[1] = m.o.lit + m.lastLL; meta[2] = m.lastLL;
}
#include "zj_match_run.h"
#include "zj_match_wavex.h"
// `wide`: the frame is in the launch whose fast-strategy tables hold 4-byte positions (frames > 64 KiB, and the few
// small level-1/2 frames whose hashLog exceeds the common case)
// `flags` (zj_need.h, level 3): the frame's flag bytes — the gated machine
ZJ_DEV void ze_match_lane(const u8* src, u32 srcSize, u32 level, u8* table, u8* fscratch, u32 maxSrc, u32* meta, bool wide = false, const u8* flags = nullptr) {
    if (srcSize < ZL_MIN_FRAME) ze_match_lane_serial(src, srcSize, level, table, fscratch, maxSrc, meta);
    else if (ZE_LW_LEVEL(level) == 3 && flags) ze_match_lane_t<ZLaneD<ZEEntTag, true> >(src, srcSize, level, table, fscratch, maxSrc, meta, flags);
    else if (ZE_LW_LEVEL(level) == 3) ze_match_lane_t<ZLaneD<ZEEntTag> >(src, srcSize, level, table, fscratch, maxSrc, meta);
    else if (wide) ze_match_lane_t<ZLaneF<ZEEnt32> >(src, srcSize, level, table, fscratch, maxSrc, meta);
    else ze_match_lane_t<ZLaneF<ZEEnt16> >(src, srcSize, level, table, fscratch, maxSrc, meta);
}
// per-frame table bytes of the two lane-machine launches: the common case (<= 64 KiB frames with the level's usual tables)
// and the wide one (any frame <= 128 KiB: fast tables up to hashLog 15 with 4-byte entries)
ZJ_HD u32 ze_lane_table_stride(u32 levelWord, bool wide) {
    u32 const level = ZE_LW_LEVEL(levelWord), hl = ZE_LW_HL(levelWord), cl = ZE_LW_CL(levelWord);
    if (level == 3 && (hl | cl)) return ((1u << (hl ? hl : 16u)) + (1u << (cl ? cl : 15u))) * 4u;   // the one not given keeps the level's own size; adjustment only shrinks
    if (level == 3) return ((1u << ZE_L3_HASHLOG) + (1u << ZE_L3_CHAINLOG)) * 4u;
    if (wide) return (1u << 15) * 4u;
    return level == 1 ? (8192u * 2u) : (32768u * 2u);
}
#define ZE_WIDE_MAX_SRC ZE_BLOCK_MAX
