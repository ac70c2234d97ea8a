// zj_decode_split.h — three-stage decoder for large batches of "simple" frames.
//
// The fused wave-per-frame decoder (zj_decode.h) spends most of its time on one lane: the tANS sequence decode
// (N/decompress/zstd_decompress_block.c:1229-1347) is a dependent chain per frame, a wave issues it with 1 of
// 64 lanes active, and three waves per SIMD already saturate the issue port (profiles/r01*_phase_cycles*).
// For batches of thousands of frames the chain is moved to where it can use all lanes:
//
//   stage 1  zd_prep_frame   wave per frame   headers, NCounts, tANS tables -> HBM (8-byte cells), frame record
//   stage 2  ZDSeqLane       LANE per frame   tANS sequence decode, 64 frames per wave in rounds: each round is
//                                             one memory round trip (3 cells + 16 bitstream bytes) per lane
//   stage 3  zd_exec_frame   wave per frame   literals (Huffman) + LZ77 execution of the decoded sequences
//
// "Simple" = one zstd frame filling its buffer, known content size <= 128 KiB (one block), a single compressed
// block — what the batched compressor emits for its records.  Everything else (multi-block, multi-frame,
// skippable, raw/RLE blocks, any error) is routed to the fused kernel, which stays the reference for behaviour
// and error codes: stages 1-3 never report an error themselves, they hand the frame over.
#pragma once

#if ZJ_ON_GPU
#define ZD_ROUND_FENCE5(a, b, c, d, e) asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(e))
#else
#define ZD_ROUND_FENCE5(a, b, c, d, e) ((void)0)
#endif
// a block's "stage 2 is done" flag: set behind everything the lane wrote for the block; stage 3 polls it when it runs beside stage 2
#if ZJ_ON_GPU
ZJ_DEV void zj_block_ready(u32* flag) { zj_release(); __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
static inline void zj_block_ready(u32* flag) { *flag = 1u; }
#endif
#define ZD_SPLIT_MAX_CONTENT 131072u
#define ZD_SPLIT_MAXSEQ (ZD_SPLIT_MAX_CONTENT / 3u + 2u)      // every sequence emits >= 3 bytes (minimum match)
#define ZD_SPLIT_CELLS 1280u                                  // LL 512 | OF 256 | ML 512
#define ZD_SPLIT_OF 512u
#define ZD_SPLIT_ML 768u
#define ZD_SPLIT_TAB_BYTES (ZD_SPLIT_CELLS * 2u)
#define ZD_SPLIT_SEQ_BYTES (ZD_SPLIT_MAXSEQ * 8u)

// Decode cells go to HBM as 2 bytes: symbol (6 bits) | the tANS "nextState" counter of the table build (10 bits, < 2 * tableSize).
// The reference's cell (ZSTD_seqSymbol, N/decompress/zstd_decompress_internal.h:68-73) follows from it: nbBits = tableLog -
// highbit(counter), nextState base = (counter << nbBits) - tableSize (N/decompress/zstd_decompress_block.c:540-556), and
// baseValue / nbAdditionalBits are functions of the symbol (two small LDS tables).  2.5 KiB per frame instead of 10: the
// lane-per-frame decode reads three random cells per sequence, and what bounds it is how many of those reads leave the L2 /
// Infinity Cache (profiles/r02*), i.e. the bytes of table alive per lane.
// decoded sequence record: ll | ml << 18 | offset << 36 (ll, ml <= 2^17 for content <= 128 KiB; the offset field has 28 bits)
ZJ_DEV u64 zd_seq_pack(u32 ll, u32 ml, u32 off) { return (u64)ll | ((u64)ml << 18) | ((u64)off << 36); }

struct ZDMeta {
    u32 blockOff, blockSize;      // block body within the frame buffer
    u32 seqOff;                   // first byte of the sequence bitstream, relative to the block body
    u32 nbSeq, litSize;
    u32 logs;                     // llLog | ofLog << 8 | mlLog << 16
    u32 contentSize, blockSizeMax;
    u32 status;                   // stage 2: 0 ok, 1 hand over to the fused kernel
    u32 hasChecksum;              // 4 checksum bytes follow the block
    u32 dictTables;               // all three tANS tables are the dictionary's: stage 2 reads them there (shared, cache-resident)
    u32 pad;
};

// ---------------------------------------------------------------------------------------------
// Stage 1.  Returns true (wave-uniform) when the frame is simple and its tables/record were written.
template <bool DICT = false, class G>
ZJ_DEV bool zd_prep_frame(const G& g, ZDecShared& sh, const u8* src, u32 srcSize, u32 dstCap, u16* tab, ZDMeta* meta, const ZDDictDev* ddArg = nullptr) {
    const ZDDictDev* const dd = DICT ? ddArg : nullptr;
    bool const dictEntropy = DICT && dd && dd->hasEntropy;      // the frame may start in repeat / treeless modes
    GRP_SERIAL(g) {
        u32 ok = 0;
        sh.err = 0; sh.seqValid = dictEntropy ? 1u : 0u; sh.hufValid = 0; sh.hufX2 = 0;
        if (dictEntropy) { sh.llLog = dd->llLog; sh.ofLog = dd->ofLog; sh.mlLog = dd->mlLog; }
        // frame header, N/decompress/zstd_decompress.c:447-557
        if (srcSize >= 16 && ld32(src) == 0xFD2FB528u) {
            u32 const fhd = src[4], didc = fhd & 3, single = (fhd >> 5) & 1, fcsid = fhd >> 6;
            u32 const fcsSz = fcsid == 0 ? single : (1u << fcsid);
            u32 const didSz = didc == 3 ? 4u : didc;
            u32 const hdr = 5 + !single + didSz + fcsSz;
            if (!(fhd & 8) && (didc == 0 || DICT) && fcsSz != 0 && fcsSz <= 4 && srcSize >= hdr + 3) {
                u32 pos = 5; u64 window = 0; u32 content = 0; bool wok = true;
                if (!single) { u32 const wd = src[pos++], wl = (wd >> 3) + 10; if (wl > 27) wok = false; window = (u64)1 << wl; window += (window >> 3) * (wd & 7); }
                if (didc) { u32 const id = didc == 1 ? src[pos] : (didc == 2 ? ld16(src + pos) : ld32(src + pos)); if (!dd || id != dd->dictID) wok = false; pos += didSz; }   // a wrong dictionary is the fused kernel's to report
                if (fcsid == 0) content = src[pos]; else if (fcsid == 1) content = ld16(src + pos) + 256; else content = ld32(src + pos);
                if (single) window = content;
                u32 const bh = ld24(src + hdr), last = bh & 1, type = (bh >> 1) & 3, sz = bh >> 3;
                u32 const tail = ((fhd >> 2) & 1) ? 4u : 0u;
                u32 const bmax = window < ZD_BLOCK_MAX ? (u32)window : ZD_BLOCK_MAX;
                if (wok && content <= ZD_SPLIT_MAX_CONTENT && content <= dstCap && last && type == 2 && sz >= 2 && sz <= bmax
                    && (u64)hdr + 3 + sz + tail == srcSize) {
                    // literals section header, N/decompress/zstd_decompress_block.c:134-340 (sizes only)
                    const u8* const b = src + hdr + 3;
                    u32 const b0 = b[0], lt = b0 & 3, fmt = (b0 >> 2) & 3; u32 lh = 0, n = 0, c = 0; bool lok = true;
                    if (lt < 2) {
                        if (fmt == 0 || fmt == 2) { lh = 1; n = b0 >> 3; } else if (fmt == 1) { lh = 2; n = ld16(b) >> 4; }
                        else if (sz < 3) lok = false; else { lh = 3; n = ld24(b) >> 4; }
                        c = (lt == 0) ? n : 1;
                    } else if (sz < 5 || (lt == 3 && !dictEntropy)) lok = false;
                    else {
                        u32 const lhc = ld32(b);
                        if (fmt < 2) { lh = 3; n = (lhc >> 4) & 0x3FF; c = (lhc >> 14) & 0x3FF; }
                        else if (fmt == 2) { lh = 4; n = (lhc >> 4) & 0x3FFF; c = lhc >> 18; }
                        else { lh = 5; n = (lhc >> 4) & 0x3FFFF; c = (lhc >> 22) + ((u32)b[4] << 10); }
                    }
                    if (lok && n <= bmax && lh + c < sz) {
                        ok = 1;
                        sh.hdrSize = hdr + 3; sh.blkSize = sz; sh.litSize = n; sh.litHdr = lh; sh.litCSize = c;
                        sh.contentSize = content; sh.blockSizeMax = bmax; sh.hasChecksum = tail ? 1u : 0u;
                    }
                }
            }
        }
        sh.blkType = ok;
    }
    g.sync();
    if (!ZJ_UNI(sh.blkType)) return false;
    u32 const boff = ZJ_UNI(sh.hdrSize), bsize = ZJ_UNI(sh.blkSize);
    if (dictEntropy) { zd_load_dict_entropy(g, sh, dd, false, true); g.sync(); }      // repeat modes copy the dictionary's tables
    zd_seq_tables(g, sh, src + boff, bsize, ZJ_UNI(sh.litHdr) + ZJ_UNI(sh.litCSize), (u32*)sh.huf);      // (stage 1 decodes no literals: the Huffman table's room is free)
    g.sync();
    u32 const nbSeq = ZJ_UNI(sh.nbSeq);
    if (ZJ_UNI(sh.err) || nbSeq > ZD_SPLIT_MAXSEQ) { GRP_SERIAL(g) { sh.err = 0; } g.sync(); return false; }
    bool const allRepeat = dictEntropy && nbSeq && ZJ_UNI(sh.tblMode[0]) == 3 && ZJ_UNI(sh.tblMode[1]) == 3 && ZJ_UNI(sh.tblMode[2]) == 3;
    if (nbSeq && !allRepeat) {
        u32 const llLog = ZJ_UNI(sh.llLog), ofLog = ZJ_UNI(sh.ofLog), mlLog = ZJ_UNI(sh.mlLog);
        GRP_FOR(g, u, 1u << llLog) tab[u] = zd_cell16(sh.ll[u], llLog);
        GRP_FOR(g, u, 1u << ofLog) tab[ZD_SPLIT_OF + u] = zd_cell16(sh.of[u], ofLog);
        GRP_FOR(g, u, 1u << mlLog) tab[ZD_SPLIT_ML + u] = zd_cell16(sh.ml[u], mlLog);
    }
    GRP_SERIAL(g) {
        ZDMeta m;
        m.blockOff = boff; m.blockSize = bsize; m.seqOff = sh.seqOff; m.nbSeq = nbSeq; m.litSize = sh.litSize;
        m.logs = sh.llLog | (sh.ofLog << 8) | (sh.mlLog << 16);
        m.contentSize = (u32)sh.contentSize; m.blockSizeMax = sh.blockSizeMax; m.status = 0; m.hasChecksum = sh.hasChecksum; m.dictTables = allRepeat ? 1u : 0u; m.pad = 0;
        *meta = m;
    }
    zj_mem_order();
    g.sync();
    return true;
}

// Stage 1, frames of ONE STORED BLOCK (round 6): what the compressors write for data that does not compress — a quarter of the bench batches — is a header and a raw
// (or RLE) block; such a frame is not "simple" (no sequences to decode on a lane) and went through the block stages behind everything else (a 0.9 ms tail of four
// launches for 16 384 copies).  Stage 1 is a wave per frame already: it copies the block where it stands.  Only the plain case is taken — known content size equal to the
// block's, the block the frame's last and within blockSizeMax, the buffer ending with the frame, room in the destination, no dictionary ID; a checksum is verified —
// and everything else (and a checksum that does not match) stays with the paths that report errors.  Returns true (wave-uniform) when dst holds the frame's content.
template <class G>
ZJ_DEV bool zd_prep_frame_stored(const G& g, ZDecShared& sh, const u8* src, u32 srcSize, u8* dst, u64 dstCap, u64* result) {
    GRP_SERIAL(g) {
        u32 ok = 0;
        if (srcSize >= 9 && ld32(src) == 0xFD2FB528u) {
            u32 const fhd = src[4], didc = fhd & 3, single = (fhd >> 5) & 1, fcsid = fhd >> 6;
            u32 const fcsSz = fcsid == 0 ? single : (1u << fcsid);
            u32 const hdr = 5 + !single + fcsSz;
            if (!(fhd & 8) && didc == 0 && fcsSz != 0 && fcsSz <= 4 && srcSize >= hdr + 3) {
                u32 pos = 5; u64 window = 0; u32 content = 0; bool wok = true;
                if (!single) { u32 const wd = src[pos++], wl = (wd >> 3) + 10; if (wl > 27) wok = false; window = (u64)1 << wl; window += (window >> 3) * (wd & 7); }
                if (fcsid == 0) content = src[pos]; else if (fcsid == 1) content = ld16(src + pos) + 256; else content = ld32(src + pos);
                if (single) window = content;
                u32 const bh = ld24(src + hdr), last = bh & 1, type = (bh >> 1) & 3, sz = bh >> 3;
                u32 const tail = ((fhd >> 2) & 1) ? 4u : 0u, body = type == 1 ? 1u : sz;
                u32 const bmax = window < ZD_BLOCK_MAX ? (u32)window : ZD_BLOCK_MAX;
                if (wok && last && type < 2 && sz == content && sz <= bmax && (u64)sz <= dstCap && (u64)hdr + 3 + body + tail == srcSize) {
                    ok = 1; sh.hdrSize = hdr + 3; sh.blkSize = sz; sh.blkType = type; sh.hasChecksum = tail ? 1u : 0u;
                }
            }
        }
        sh.litType = ok;
    }
    g.sync();
    if (!ZJ_UNI(sh.litType)) return false;
    u32 const boff = ZJ_UNI(sh.hdrSize), sz = ZJ_UNI(sh.blkSize), type = ZJ_UNI(sh.blkType);
    if (type == 0) grp_copy_wide(g, dst, src + boff, sz); else zd_fill(g, dst, src[boff], sz);
    zj_mem_order();
    g.sync();
    if (ZJ_UNI(sh.hasChecksum)) {
        if ((u32)zj_xxh64(g, dst, sz) != ZJ_UNI(ld32(src + boff + (type == 1 ? 1u : sz)))) return false;      // (the fused kernel says checksum_wrong)
    }
    GRP_SERIAL(g) { *result = sz; }
    return true;
}

// ---------------------------------------------------------------------------------------------
// Stage 2.  One lane per frame; round() is one memory round trip for every lane of the wave.
// Bit positions are relative to the frame buffer.  N/decompress/zstd_decompress_block.c:1229-1347, :1615-1690.
// the two per-symbol tables of stage 2: baseValue | nbAdditionalBits << 24 (LL: 36 entries, ML: 53)
ZJ_DEV void zd_seq_symtabs(u32* ll, u32* ml, u32 first, u32 stride) {
    for (u32 i = first; i < 36u; i += stride) ll[i] = zd_k_ll_base[i] | ((u32)zd_k_ll_bits[i] << 24);
    for (u32 i = first; i < 53u; i += stride) ml[i] = zd_k_ml_base[i] | ((u32)zd_k_ml_bits[i] << 24);
}
// MB = true: the lane decodes ONE BLOCK of a multi-block frame (below, "multi-block frames"): the repcode history it starts from is the previous block's and
// not known yet, so it is carried SYMBOLICALLY — an entry is a concrete offset or "entry k of the history before this block, minus d" (ZD_SYM) — and so are the
// offsets of the records that use it; the execution stage, which walks the frame's blocks in order, resolves both.  The range check of an offset against the
// frame's output position moves there too (the block's position in the frame is not known here either).
#define ZD_SYM 0x80000000u                                     /* in registers: flag | k << 29 | d */
#define ZD_SYM_REC 0x8000000u                                  /* in a record's 28-bit offset field: flag | k << 25 | d */
struct ZDBlk {                    // one block of a multi-block frame (stage 1 writes, stage 2 completes, stage 3 reads)
    u32 frame;                    // index of the frame in the batch
    u32 blockOff, blockSize;      // block body within the frame buffer (RLE: blockSize = the regenerated size, the body is one byte)
    u32 type;                     // 0 raw, 1 RLE, 2 compressed
    u32 seqOff, nbSeq, litSize, logs;
    u32 regen;                    // bytes the block decodes to (stage 1: raw / RLE / no sequences; stage 2 otherwise)
    u32 rep[3];                   // the repcode history after the block, in terms of the history before it where it has to be (ZD_SYM)
    u32 status;                   // stage 2: 0 ok, 1 hand the frame over to the fused kernel
    u32 blockSizeMax;
    u32 seqLo, seqHi;             // first record of the block in the pool
    u32 litLo, litHi;             // its slot in the literal pool (stage 2b of these frames), ~0: none
    u32 hufBlk;                   // treeless literals: the block whose Huffman table they use (its index in blks[]), ~0: none in this frame
    u32 litReady;                 // stage 2b: the Huffman-coded literals are in the slot
    u32 seqReady;                 // stage 2 is done with the block (1 from stage 1 on for blocks without sequences): stage 3 running beside stage 2 waits for it, block by block
};
struct ZDFrameMB { u32 firstBlk, nBlk, hasChecksum, known; u64 contentSize; u32 frameEnd, pad; };     // known: the header carries the content size
template <bool MB>
struct ZDSeqLaneT {
    const u8* src; const u16* tab; u64* seqs; ZDMeta* meta; ZDBlk* blk; const u32* llBase; const u32* mlBase;   // LDS tables of the kernel: baseValue | nbAdditionalBits << 24 per symbol (zd_seq_symtabs)
    i32 A, S0; u32 sLL, sOF, sML, rep0, rep1, rep2, i, nbSeq, opos, lpos, litSize, cap, logs, endByte;
    u32 st;                       // 0 start, 1 running, 2 done
    u32 bad, dictSize;
    // what the previous round fetched for this one: the three cells of the current states and the 16 bitstream bytes [eAt-16, eAt)
    u64 fLo, fHi; u16 fCl, fCo, fCm; u32 eAt;    // (cells kept 2 bytes wide: widening them here would put a wait for the loads at the end of the round)
    u64 pend; bool havePend;      // the previous sequence's record: stored at the top of the next round, ahead of that round's loads

    ZJ_DEV_MEMBER void init(const u8* s, const u16* t, u64* q, ZDMeta* m, const ZDDictDev* dd = nullptr) {
        src = s; tab = t; seqs = q; meta = m;
        ZDMeta const h = *m;
        if (dd && h.dictTables) tab = dd->c16;
        nbSeq = h.nbSeq; litSize = h.litSize; logs = h.logs;
        cap = zj_min(h.contentSize, h.blockSizeMax);
        S0 = (i32)((h.blockOff + h.seqOff) * 8u); endByte = h.blockOff + h.blockSize; A = S0;
        rep0 = 1; rep1 = 4; rep2 = 8; i = 0; opos = 0; lpos = 0; bad = 0; sLL = sOF = sML = 0; dictSize = 0;
        if (dd) { dictSize = dd->contentSize; if (dd->hasEntropy) { rep0 = dd->rep[0]; rep1 = dd->rep[1]; rep2 = dd->rep[2]; } }
        havePend = false; fLo = fHi = 0; fCl = fCo = fCm = 0; eAt = 0;
        st = nbSeq ? 0u : 2u;
    }
    ZJ_DEV_MEMBER void init_block(const u8* s, const u16* t, u64* q, ZDBlk* b) {       // MB: block b of a multi-block frame (its body inside the frame buffer s)
        src = s; tab = t; seqs = q; meta = nullptr; blk = b;
        ZDBlk const h = *b;
        nbSeq = h.nbSeq; litSize = h.litSize; logs = h.logs; cap = h.blockSizeMax;
        S0 = (i32)((h.blockOff + h.seqOff) * 8u); endByte = h.blockOff + h.blockSize; A = S0;
        rep0 = ZD_SYM; rep1 = ZD_SYM | (1u << 29); rep2 = ZD_SYM | (2u << 29);
        i = 0; opos = 0; lpos = 0; bad = 0; sLL = sOF = sML = 0; dictSize = 0;
        havePend = false; fLo = fHi = 0; fCl = fCo = fCm = 0; eAt = 0;
        st = nbSeq ? 0u : 2u;
    }
    ZJ_DEV_MEMBER void finish() {
        if (MB) {
            u32 const regen = opos + (litSize - lpos);
            if (regen > cap) bad = 1;
            blk->regen = regen; blk->rep[0] = rep0; blk->rep[1] = rep1; blk->rep[2] = rep2; blk->status = bad ? 1u : 0u;
            zj_block_ready(&blk->seqReady);       // (records, sizes and history first: stage 3 may be waiting for this block)
        } else meta->status = bad ? 1u : 0u;
        st = 2;
    }
    // top 64 bits below bit position `at` out of the 16 bytes [e-16, e) (hi = upper 8 bytes); at in (8e-128+63, 8e]
    ZJ_DEVM u64 top64(u64 hi, u64 lo, u32 e, u32 at) {
        u32 const s = 8u * e - at;                  // unused bits above `at`
        if (s == 0) return hi;
        if (s < 64u) return (hi << s) | (lo >> (64u - s));
        return s == 64u ? lo : (lo << (s - 64u));
    }
    // issue the loads the NEXT round consumes (current states, bit position A); nothing waits for them here
    ZJ_DEV_MEMBER void fetch() {
        u32 const e = ((u32)A + 7u) >> 3, wp = e >= 16u ? e - 16u : 0u;
        eAt = e;
        fLo = ld64(src + wp); fHi = ld64(src + wp + 8);
        fCl = tab[sLL]; fCo = tab[ZD_SPLIT_OF + sOF]; fCm = tab[ZD_SPLIT_ML + sML];
    }
    ZJ_DEVM void align16(u64& hi, u64& lo, u32 e) {       // stream within 16 bytes of the buffer start: align [e-16, e) by hand
        if (e < 16u) {
            u32 const k = (16u - e) * 8u;                // shift left by k bits (8..120)
            if (k < 64u) { hi = (hi << k) | (lo >> (64u - k)); lo <<= k; } else { hi = k == 64u ? lo : (lo << (k - 64u)); lo = 0; }
        }
    }
    // One round = one sequence.  The round is a dependent chain — this sequence's cells and bits give the next states and bit
    // position, which address the next cells and bits — and a wave runs alone on its SIMD, so the chain's length is the kernel's
    // time.  The round therefore does only what the next addresses need (bit counts, state update), issues the next round's
    // loads, and computes the sequence itself (values, repcode history, ZSTD_execSequence's checks, the record) while they fly.
    ZJ_DEV_MEMBER void round() {
        if (st == 0) {
            u32 const e = endByte, wp = e >= 16u ? e - 16u : 0u;
            u64 lo = ld64(src + wp), hi = ld64(src + wp + 8);
            ZD_ROUND_FENCE5(lo, hi, e, e, e);
            align16(hi, lo, e);
            // last byte carries the end mark; then the three initial states LL, OF, ML (:1640-1642)
            u32 const lastByte = (u32)(hi >> 56);
            if (lastByte == 0 || (u32)S0 >= 8u * endByte) { bad = 1; finish(); return; }
            A = (i32)(8u * (endByte - 1u) + zj_hibit(lastByte));
            u32 const a = logs & 0xFF, b = (logs >> 8) & 0xFF, c = (logs >> 16) & 0xFF;
            if (A - (i32)(a + b + c) < S0) { bad = 1; finish(); return; }
            u64 v = top64(hi, lo, e, (u32)A);
            sLL = a ? (u32)(v >> (64u - a)) : 0u; v <<= a;
            sOF = b ? (u32)(v >> (64u - b)) : 0u; v <<= b;
            sML = c ? (u32)(v >> (64u - c)) : 0u;
            A -= (i32)(a + b + c);
            st = 1;
            fetch();
            return;
        }
        if (st != 1) return;
        u64 lo = fLo, hi = fHi; u32 const e = eAt;
        ZD_ROUND_FENCE5(lo, hi, fCl, fCo, fCm);
        u32 const cl = fCl, co = fCo, cm = fCm;
        align16(hi, lo, e);
        // ---- what the next round's addresses depend on ----
        u32 const La = logs & 0xFF, Lb = (logs >> 8) & 0xFF, Lc = (logs >> 16) & 0xFF;
        u32 const syl = cl & 63u, syo = co & 63u, sym = cm & 63u, nsl = cl >> 6, nso = co >> 6, nsm = cm >> 6;
        u32 const bbl = llBase[syl], bbm = mlBase[sym];
        u32 const ofx = syo, mlx = bbm >> 24, llx = bbl >> 24;
        bool const last = (i + 1u == nbSeq);
        u32 const nbl = La - zj_hibit(nsl), nbm = Lc - zj_hibit(nsm), nbo = Lb - zj_hibit(nso);       // the cells' nbBits
        u32 const nl = last ? 0u : nbl, nm = last ? 0u : nbm, no = last ? 0u : nbo;
        u32 const T1 = ofx + mlx + llx, T = T1 + nl + nm + no;
        if (A - (i32)T < S0) { bad = 1; finish(); return; }
#define ZD_TAKE(v, nb) ((u32)(((v) >> 1) >> (63u - (nb))))
        u32 const A0 = (u32)A;
        u64 v2 = top64(hi, lo, e, A0 - T1);
        u32 const vl = ZD_TAKE(v2, nl); v2 <<= nl;
        u32 const vm = ZD_TAKE(v2, nm); v2 <<= nm;
        u32 const vo = ZD_TAKE(v2, no);
        A -= (i32)T;
        if (havePend) { seqs[i - 1u] = pend; havePend = false; }      // ahead of the loads: the next round's wait does not include this store's round trip
        if (!last) { sLL = (nsl << nbl) - (1u << La) + vl; sML = (nsm << nbm) - (1u << Lc) + vm; sOF = (nso << nbo) - (1u << Lb) + vo; }
        fetch();                                                         // (after the last sequence too: same cells, a valid bit position — no second code path)
        // ---- the sequence itself ----
        u32 const llb = bbl & 0xFFFFFFu, mlb = bbm & 0xFFFFFFu;
        u32 const ofb = ofx > 1u ? (1u << ofx) - 3u : ofx;                // OF_base of code ofx
        u64 v = top64(hi, lo, e, A0);
        u32 const ofv = ZD_TAKE(v, ofx); v <<= ofx;
        u32 const mlv = ZD_TAKE(v, mlx); v <<= mlx;
        u32 const llv = ZD_TAKE(v, llx);
#undef ZD_TAKE
        u32 const llen = llb + llv, mlen = mlb + mlv;
        u32 offset;
        if (ofx > 1u) {
            offset = ofb + ofv;
            // MB: a concrete offset that does not fit a record's 27 bits must not reach the history — from offset code 31 on it would carry bit 31 and read as a
            // symbolic entry (a crafted block decoded to wrong bytes where the reference reports corruption); such windows are the fused kernel's
            if (MB && offset >= ZD_SYM_REC) { bad = 1; finish(); return; }
            rep2 = rep1; rep1 = rep0; rep0 = offset;
        } else {
            u32 const ll0 = (llen == 0u);
            if (ofx == 0u) { if (ll0) { offset = rep1; rep1 = rep0; rep0 = offset; } else offset = rep0; }
            else {
                u32 const idx = 1u + ll0 + ofv;
                u32 t;
                if (MB) {
                    // "repcode 1 minus one" of a symbolic entry stays symbolic (d + 1); a concrete zero is what the reference forces to -1 and then rejects
                    u32 const h0 = rep0, h1 = rep1, h2 = rep2;                  // (values first: a select between the members themselves kept the whole lane in scratch memory — 29 scratch instructions in zj_dec_seq_mb_kernel)
                    u32 const r = (idx == 3u) ? h0 : (idx == 1u ? h1 : h2);
                    t = (idx == 3u) ? ((r & ZD_SYM) ? r + 1u : r - 1u) : r;
                    if (!(t & ZD_SYM) && t == 0u) { bad = 1; finish(); return; }
                    if ((t & ZD_SYM) && (t & 0x1FFFFFFFu) >= 0x1FFFFFFu) { bad = 1; finish(); return; }     // (d does not fit a record: never in practice, the fused kernel's then)
                } else { t = (idx == 3u) ? rep0 - 1u : (idx == 1u ? rep1 : rep2); t -= !t; }
                if (idx != 1u) rep2 = rep1;
                rep1 = rep0; rep0 = t; offset = t;
            }
        }
        // the checks of ZSTD_execSequence (:1001-1096): literals available, room in the block, offset inside the output (MB: the last one is stage 3's)
        if (llen > litSize - lpos || (u64)opos + llen + mlen > cap || (!MB && offset > opos + llen + dictSize)) { bad = 1; finish(); return; }
        if (MB && !(offset & ZD_SYM) && offset >= ZD_SYM_REC) { bad = 1; finish(); return; }               // (an offset beyond 2^27: windows that large are the fused kernel's)
        u64 const rec = zd_seq_pack(llen, mlen, (MB && (offset & ZD_SYM)) ? (ZD_SYM_REC | (((offset >> 29) & 3u) << 25) | (offset & 0x1FFFFFFu)) : offset);
        lpos += llen; opos += llen + mlen; i++;
        if (i == nbSeq) { seqs[i - 1u] = rec; if (A != S0) bad = 1; finish(); }
        else { pend = rec; havePend = true; }
    }
};

typedef ZDSeqLaneT<false> ZDSeqLane;

// ---------------------------------------------------------------------------------------------
// Stage 2b (beside stage 2).  The Huffman-coded literals of a simple frame do not depend on its sequences: while the lane-per-frame
// decode occupies one wave per SIMD with a chain of memory round trips, this pass regenerates the literals of the same frames into a
// per-frame slot in HBM (N/decompress/zstd_decompress_block.c:134-340, huf_decompress.c:721-835 — zd_block_literals, unchanged) and
// marks the frame record (ZDMeta::pad = 1); stage 3 then starts at the LZ77 execution.  Frames it does not mark — raw / RLE literals,
// literals beyond the slot, treeless literals of dictionary frames, anything wrong with the section — are stage 3's as before.
// Returns true (wave-uniform) when `out[0, litSize)` holds the literals.
template <class G>
ZJ_DEV bool zd_lit_frame(const G& g, ZDecShared& sh, const u8* src, const ZDMeta* meta, u8* out, u32 slot, ZjProf& pf, u8* hpWin = nullptr) {      // hpWin: ZD_HP_LDS bytes of LDS (zd_huf_streams_wave) or none
    GRP_SERIAL(g) {
        ZDMeta const m = *meta;
        sh.err = 0; sh.hufValid = 0; sh.hufX2 = 0;
        sh.hdrSize = m.blockOff; sh.blkSize = m.blockSize; sh.blockSizeMax = m.blockSizeMax;
        sh.blkType = ((src[m.blockOff] & 3u) == 2u && m.litSize <= slot) ? 1u : 0u;      // Huffman with its own table, and it fits
    }
    g.sync();
    if (!ZJ_UNI(sh.blkType)) return false;
    const u8* const lit = zd_block_literals(g, sh, src + ZJ_UNI(sh.hdrSize), ZJ_UNI(sh.blkSize), out, pf, slot, nullptr, false, hpWin);
    bool const ok = lit != nullptr && !ZJ_UNI(sh.err);
    zj_mem_order();
    g.sync();
    return ok;
}

// ---------------------------------------------------------------------------------------------
// Stage 3.  Returns the decoded size, or ~0 (wave-uniform) to hand the frame to the fused kernel.
// preLit / preAvail: the frame's slot of stage 2b and its size; used when the frame record says the literals are there.
template <bool DICT = false, class G>
ZJ_DEV u64 zd_exec_frame(const G& g, ZDecShared& sh, const u8* src, u8* dst, const ZDMeta* meta, const u64* seqs, u8* litScratch, ZjProf& pf,
                         const ZDDictDev* ddArg = nullptr, const u8* dictRaw = nullptr, const u8* preLit = nullptr, u32 preAvail = 0, u8* hpWin = nullptr) {
    const ZDDictDev* const dd = DICT ? ddArg : nullptr;
    const u8* const dictEnd = dd ? dictRaw + dd->contentOff + dd->contentSize : nullptr;
    GRP_SERIAL(g) {
        ZDMeta const m = *meta;
        sh.err = m.status ? (u32)ZJ_E_CORRUPTION : 0u; sh.hufValid = 0; sh.hufX2 = 0;
        sh.hdrSize = m.blockOff; sh.blkSize = m.blockSize; sh.nbSeq = m.nbSeq; sh.blockSizeMax = m.blockSizeMax; sh.contentSize = m.contentSize;
        sh.hasChecksum = m.hasChecksum;
        sh.litStreams = (preLit && m.pad == 1u) ? 1u : 0u;     // (a scratch word here: zd_block_literals sets it from the section header)
    }
    g.sync();
    if (ZJ_UNI(sh.err)) return ~(u64)0;
    bool const havePre = ZJ_UNI(sh.litStreams) != 0u;
    if (DICT && dd && dd->hasEntropy && (src[ZJ_UNI(sh.hdrSize)] & 3u) == 3u) {   // treeless literals decode with the dictionary's Huffman table
        zd_load_dict_entropy(g, sh, dd, true, false);
        GRP_SERIAL(g) { sh.hufValid = 1; sh.hufX2 = 1; sh.hufLog = dd->hufLog; sh.hufW1 = dd->hufW1; sh.hufGcd = 1; }
        g.sync();
    }
    const u8* const bsrc = src + ZJ_UNI(sh.hdrSize); u32 const bsize = ZJ_UNI(sh.blkSize);
    u32 const nbSeq = ZJ_UNI(sh.nbSeq), content = (u32)zj_uni64(sh.contentSize);
    u32 const cap = zj_min(content, ZJ_UNI(sh.blockSizeMax));
    const u8* const lit = zd_block_literals(g, sh, bsrc, bsize, litScratch, pf, ~0u, havePre ? preLit : nullptr, false, hpWin);
    if (ZJ_UNI(sh.err)) return ~(u64)0;
    pf.mark(2);
    u32 const litSize = ZJ_UNI(sh.litSize);
    // bytes that may be read starting at `lit`: the rest of the block for raw literals, the scratch (or the frame's slot) otherwise
    u32 const litAvail = (ZJ_UNI(sh.litType) == 0) ? bsize - ZJ_UNI(sh.litHdr) : ((havePre && ZJ_UNI(sh.litType) >= 2u) ? preAvail : ZD_LIT_SCRATCH);
    u32 lp = 0, op = 0;
    for (u32 base = 0; base < nbSeq; base += ZD_SEQ_BATCH) {
        u32 const cnt = zj_min(ZD_SEQ_BATCH, nbSeq - base);
        GRP_FOR(g, k, cnt) {
            u64 const q = seqs[base + k];
            sh.sLit[k] = (u32)q & 0x3FFFFu; sh.sMl[k] = (u32)(q >> 18) & 0x3FFFFu; sh.sOff[k] = (u32)(q >> 36);
        }
        g.sync();
        u32 lt, ot;
        zd_execute_batch<DICT>(g, sh, dst, lit, cnt, lp, op, lt, ot, (u8*)sh.huf, litAvail, dictEnd);   // the Huffman table is dead once the literals are decoded
        lp += lt; op += ot;
        g.sync();
    }
    pf.mark(5);
    {   u32 const rest = litSize - lp;               // stage 2 checked lp <= litSize
        if ((u64)op + rest > cap) return ~(u64)0;
        grp_copy_wide(g, dst + op, lit + lp, rest);
        op += rest;
        zj_mem_order();
    }
    g.sync();
    pf.mark(6);
    if (op != content) return ~(u64)0;
    if (ZJ_UNI(sh.hasChecksum)) {                    // the fused kernel reports a mismatch (checksum_wrong)
        if ((u32)zj_xxh64(g, dst, op) != ZJ_UNI(ld32(bsrc + bsize))) return ~(u64)0;
    }
    return op;
}


// =============================================================================================
// Multi-block frames (and frames without a content size: what the stream classes write) on the split pipeline.
//
// The fused kernel decodes such a frame on one wave, its sequences on one lane: a 1 MiB frame is a chain of ~40 000 dependent steps
// (bench config 1: 17.7 GiB/s against 37 from 16 host threads).  What keeps the BLOCKS of a frame from being decoded side by side is
// little: the tANS tables of a block in repeat mode are the previous block's (resolved in stage 1, which walks a frame's blocks in
// order anyway), treeless literals use the previous Huffman table (stage 3 walks the blocks in order as well), and the repcode history a
// block starts from is the previous block's last — carried symbolically through stage 2 (ZDSeqLaneT<true>) and resolved in stage 3.
//
//   stage 1  zd_prep_frame_multi   wave per frame    frame header, every block header; per compressed block the literals header (sizes), the sequences
//                                                    header and the three tANS tables -> HBM cells; a ZDBlk per block, a ZDFrameMB per frame
//   stage 2  ZDSeqLaneT<true>      LANE per block    tANS sequence decode into a pool of records; regenerated size and final history per block
//   stage 3  zd_exec_frame_multi   wave per frame    blocks in order: raw / RLE copies, literals (Huffman, treeless included), history resolved,
//                                                    offsets checked against the output position, LZ77 execution, checksum
// N/decompress/zstd_decompress.c:953-1066 (frame loop), N/decompress/zstd_decompress_block.c:2066-2160 (block), :695-782 (repeat modes), :1300-1312 (history).
// Like stages 1-3 of single-block frames these never report an error: anything unusual hands the frame to the fused kernel.
#define ZD_MB_MAX_BLOCKS 1024u

// sizes of a compressed block's literals section from its header (N/decompress/zstd_decompress_block.c:134-340): header bytes, regenerated size, compressed size
ZJ_DEV bool zd_lit_sizes(const u8* b, u32 sz, u32& lh, u32& n, u32& c) {
    if (sz < 2) return false;
    u32 const b0 = b[0], lt = b0 & 3, fmt = (b0 >> 2) & 3;
    if (lt < 2) {
        if (fmt == 0 || fmt == 2) { lh = 1; n = b0 >> 3; } else if (fmt == 1) { lh = 2; n = ld16(b) >> 4; }
        else if (sz < 3) return false; else { lh = 3; n = ld24(b) >> 4; }
        c = (lt == 0) ? n : 1;
        return true;
    }
    if (sz < 5) return false;
    u32 const lhc = ld32(b);
    if (fmt < 2) { lh = 3; n = (lhc >> 4) & 0x3FF; c = (lhc >> 14) & 0x3FF; }
    else if (fmt == 2) { lh = 4; n = (lhc >> 4) & 0x3FFF; c = lhc >> 18; }
    else { lh = 5; n = (lhc >> 4) & 0x3FFFF; c = (lhc >> 22) + ((u32)b[4] << 10); }
    return true;
}

// Stage 1.  Returns true (wave-uniform) when the frame's blocks and record were written.  Blocks get consecutive entries of blks[] (claimed from *blkCounter),
// blocks with sequences consecutive records of the pool (claimed from *seqCounter) and an entry of seqList (claimed from *seqListCount).
template <class G>
ZJ_DEV bool zd_prep_frame_multi(const G& g, ZDecShared& sh, const u8* src, u32 srcSize, u64 dstCap, u32 frameIdx, ZDFrameMB* fr, ZDBlk* blks, u16* tabs,
                                u32* blkCounter, u32 blkCap, unsigned long long* seqCounter, u64 seqCap, u32* seqList, u32* seqListCount, u32 minBlocks = 1u,
                                unsigned long long* litCounter = nullptr, u64 litCap = 0, u32* litList = nullptr, u32* litListCount = nullptr) {
    GRP_SERIAL(g) {
        u32 ok = 0;
        sh.err = 0; sh.seqValid = 0; sh.hufValid = 0; sh.hufX2 = 0;
        if (srcSize >= 9 && ld32(src) == 0xFD2FB528u) {
            u32 const fhd = src[4], didc = fhd & 3, single = (fhd >> 5) & 1, fcsid = fhd >> 6;
            u32 const fcsSz = fcsid == 0 ? single : (1u << fcsid);
            u32 const hdr = 5 + !single + fcsSz;
            if (!(fhd & 8) && didc == 0 && srcSize >= hdr + 3) {
                u32 pos = 5; u64 window = 0, content = ~(u64)0; bool wok = true;
                if (!single) { u32 const wd = src[pos++], wl = (wd >> 3) + 10; if (wl > 27) wok = false; window = (u64)1 << wl; window += (window >> 3) * (wd & 7); }
                if (fcsid == 0) { if (single) content = src[pos]; } else if (fcsid == 1) content = (u64)ld16(src + pos) + 256; else if (fcsid == 2) content = ld32(src + pos); else content = ld64(src + pos);
                if (single) window = content;
                if (content != ~(u64)0 && (content > dstCap || content > 0xFFFFFFFFull)) wok = false;      // does not fit: the fused kernel says so
                u32 const tail = ((fhd >> 2) & 1) ? 4u : 0u;
                u32 const bmax = window < ZD_BLOCK_MAX ? (u32)window : ZD_BLOCK_MAX;
                // every block header; the frame must end exactly where the buffer ends
                u32 at = hdr, nb = 0; bool endOk = false;
                while (wok && at + 3 <= srcSize && nb < ZD_MB_MAX_BLOCKS) {
                    u32 const bh = ld24(src + at), last = bh & 1, type = (bh >> 1) & 3, sz = bh >> 3;
                    if (type == 3 || sz > bmax) break;
                    u32 const body = type == 1 ? 1u : sz;
                    if ((u64)at + 3 + body > srcSize) break;
                    at += 3 + body; nb++;
                    if (last) { endOk = ((u64)at + tail == srcSize); break; }
                }
                if (endOk && nb >= 1 && nb >= minBlocks) {
                    u32 const base = atomicAdd(blkCounter, nb);
                    if ((u64)base + nb <= blkCap) {
                        ok = 1;
                        sh.hdrSize = hdr; sh.bN = nb; sh.winLo = base; sh.contentSize = content; sh.blockSizeMax = bmax; sh.hasChecksum = tail ? 1u : 0u; sh.litSrcOff = at;
                    }
                }
            }
        }
        sh.blkType = ok;
    }
    g.sync();
    if (!ZJ_UNI(sh.blkType)) return false;
    u32 const nb = ZJ_UNI(sh.bN), base = ZJ_UNI(sh.winLo), bmax = ZJ_UNI(sh.blockSizeMax);
    u32 at = ZJ_UNI(sh.hdrSize);
    u32 lastHuf = ~0u;                                                     // the latest block of the frame that describes a Huffman table
    for (u32 b = 0; b < nb; b++) {
        u32 const bh = ZJ_UNI(ld24(src + at)), type = (bh >> 1) & 3, sz = bh >> 3;
        u32 const boff = at + 3;
        ZDBlk* const bk = blks + base + b;
        if (type != 2) {
            GRP_SERIAL(g) {
                ZDBlk k; k.frame = frameIdx; k.blockOff = boff; k.blockSize = sz; k.type = type; k.seqOff = 0; k.nbSeq = 0; k.litSize = 0; k.logs = 0; k.regen = sz;
                k.rep[0] = ZD_SYM; k.rep[1] = ZD_SYM | (1u << 29); k.rep[2] = ZD_SYM | (2u << 29); k.status = 0; k.blockSizeMax = bmax; k.seqLo = 0; k.seqHi = 0;
                k.litLo = k.litHi = ~0u; k.hufBlk = ~0u; k.litReady = 0; k.seqReady = 1;
                *bk = k;
            }
            at += 3 + (type == 1 ? 1u : sz);
            continue;
        }
        const u8* const body = src + boff;
        GRP_SERIAL(g) {
            u32 lh = 0, n = 0, c = 0;
            bool const lok = zd_lit_sizes(body, sz, lh, n, c) && n <= bmax && lh + c < sz;
            sh.litSize = n; sh.litHdr = lh; sh.litCSize = c; sh.litType = lok ? 1u : 0u;
        }
        g.sync();
        if (!ZJ_UNI(sh.litType)) return false;
        zd_seq_tables(g, sh, body, sz, ZJ_UNI(sh.litHdr) + ZJ_UNI(sh.litCSize), (u32*)sh.huf);
        g.sync();
        u32 const nbSeq = ZJ_UNI(sh.nbSeq);
        if (ZJ_UNI(sh.err) || nbSeq > ZD_SPLIT_MAXSEQ) { GRP_SERIAL(g) { sh.err = 0; } g.sync(); return false; }
        GRP_SERIAL(g) {
            unsigned long long const q = nbSeq ? atomicAdd(seqCounter, (unsigned long long)nbSeq) : 0ull;
            sh.tblOff[0] = (u32)q; sh.tblOff[1] = (u32)(q >> 32); sh.tblOff[2] = (q + nbSeq <= seqCap) ? 1u : 0u;
        }
        g.sync();
        if (!ZJ_UNI(sh.tblOff[2])) return false;
        if (nbSeq) {
            u16* const tab = tabs + (size_t)(base + b) * ZD_SPLIT_CELLS;
            u32 const llLog = ZJ_UNI(sh.llLog), ofLog = ZJ_UNI(sh.ofLog), mlLog = ZJ_UNI(sh.mlLog);
            GRP_FOR(g, u, 1u << llLog) tab[u] = zd_cell16(sh.ll[u], llLog);
            GRP_FOR(g, u, 1u << ofLog) tab[ZD_SPLIT_OF + u] = zd_cell16(sh.of[u], ofLog);
            GRP_FOR(g, u, 1u << mlLog) tab[ZD_SPLIT_ML + u] = zd_cell16(sh.ml[u], mlLog);
        }
        GRP_SERIAL(g) {
            ZDBlk k; k.frame = frameIdx; k.blockOff = boff; k.blockSize = sz; k.type = 2; k.seqOff = sh.seqOff; k.nbSeq = nbSeq; k.litSize = sh.litSize;
            k.logs = sh.llLog | (sh.ofLog << 8) | (sh.mlLog << 16); k.regen = sh.litSize;
            k.rep[0] = ZD_SYM; k.rep[1] = ZD_SYM | (1u << 29); k.rep[2] = ZD_SYM | (2u << 29); k.status = 0; k.blockSizeMax = bmax; k.seqLo = sh.tblOff[0]; k.seqHi = sh.tblOff[1];
            // Huffman-coded literals get a slot of the literal pool (stage 2b decodes the blocks' literals side by side); treeless ones name the block whose table they use
            u32 const lt = body[0] & 3u;
            k.litLo = k.litHi = ~0u; k.hufBlk = (lt == 3u) ? lastHuf : ~0u; k.litReady = 0; k.seqReady = nbSeq ? 0u : 1u;
            if (lt >= 2u && litList && (lt == 2u || lastHuf != ~0u)) {
                unsigned long long const q = atomicAdd(litCounter, (unsigned long long)((sh.litSize + 63u) & ~31u));
                if (q + ((sh.litSize + 63u) & ~31u) <= litCap) { k.litLo = (u32)q; k.litHi = (u32)(q >> 32); litList[atomicAdd(litListCount, 1u)] = base + b; }
            }
            *bk = k;
            if (nbSeq) seqList[atomicAdd(seqListCount, 1u)] = base + b;
        }
        g.sync();
        if ((ZJ_UNI(body[0]) & 3u) == 2u) lastHuf = base + b;
        at += 3 + sz;
    }
    GRP_SERIAL(g) {
        ZDFrameMB f; f.firstBlk = base; f.nBlk = nb; f.hasChecksum = sh.hasChecksum; f.known = sh.contentSize != ~(u64)0 ? 1u : 0u; f.contentSize = sh.contentSize; f.frameEnd = at; f.pad = 0;
        *fr = f;
    }
    zj_mem_order();
    g.sync();
    return true;
}

// Stage 2b of these frames: the Huffman-coded literals of ONE block into its slot of the literal pool (a wave per block, all blocks side by side; on the fused kernel
// and in stage 3 a frame's literals are decoded block after block on 4 of 64 lanes).  A treeless block first rebuilds the table of the block that described it.
// Returns true (wave-uniform) when the slot holds the literals; stage 3 decodes what is not marked itself.
template <class G>
ZJ_DEV bool zd_lit_block(const G& g, ZDecShared& sh, const u8* src, const ZDBlk* blks, u32 b, u8* pool, ZjProf& pf, u8* hpWin = nullptr) {
    GRP_SERIAL(g) {
        ZDBlk const k = blks[b];
        sh.err = 0; sh.hufValid = 0; sh.hufX2 = 0;
        sh.hdrSize = k.blockOff; sh.blkSize = k.blockSize; sh.blockSizeMax = k.blockSizeMax; sh.tblOff[0] = k.litLo; sh.tblOff[1] = k.litHi; sh.winLo = k.hufBlk;
        sh.blkType = (k.litLo != ~0u || k.litHi != ~0u) ? 1u : 0u;
        if (k.hufBlk != ~0u) { ZDBlk const h = blks[k.hufBlk]; sh.bLitStart = h.blockOff; sh.bOutStart = h.blockSize; }
    }
    g.sync();
    if (!ZJ_UNI(sh.blkType)) return false;
    u32 const bmaxKeep = ZJ_UNI(sh.blockSizeMax);
    if (ZJ_UNI(sh.winLo) != ~0u) {                                   // treeless: the table first
        const u8* const t = zd_block_literals(g, sh, src + ZJ_UNI(sh.bLitStart), ZJ_UNI(sh.bOutStart), nullptr, pf, ~0u, nullptr, true);
        if (t == nullptr || ZJ_UNI(sh.err)) return false;
        GRP_SERIAL(g) { sh.blockSizeMax = bmaxKeep; }
        g.sync();
    }
    u8* const out = pool + (((u64)ZJ_UNI(sh.tblOff[1]) << 32) | ZJ_UNI(sh.tblOff[0]));
    const u8* const lit = zd_block_literals(g, sh, src + ZJ_UNI(sh.hdrSize), ZJ_UNI(sh.blkSize), out, pf, ~0u, nullptr, false, hpWin);
    bool const ok = lit == out && !ZJ_UNI(sh.err);
    zj_mem_order();
    g.sync();
    return ok;
}

// stage 3 beside stage 2: until the block's flag is set (lane 0 polls, 2 s at most); wave-uniform answer, the block's record is visible behind it
template <class G>
ZJ_DEV bool zd_wait_block(const G& g, ZDecShared& sh, const u32* flag) {
#if ZJ_ON_GPU
    u32 v = 0;
    if (g.lane() == 0) {
        u64 const t0 = wall_clock64();                    // 100 MHz
        for (;;) {
            v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v || wall_clock64() - t0 > 200000000ull) break;
            __builtin_amdgcn_s_sleep(64);
        }
    }
    v = (u32)__builtin_amdgcn_readfirstlane((int)v);
    (void)sh;
    if (!v) return false;
    zj_acquire();
    return true;
#else
    (void)g; (void)sh; return *flag != 0u;
#endif
}
// Stage 3.  Returns the decoded size, or ~0 (wave-uniform) to hand the frame to the fused kernel.  `stage`: ZD_STAGE_BYTES of LDS that are not the Huffman table
// (a treeless block needs the previous block's).
template <class G>
ZJ_DEV u64 zd_exec_frame_multi(const G& g, ZDecShared& sh, const u8* src, u8* dst, u64 dstCap, const ZDFrameMB* fr, const ZDBlk* blks, const u64* pool, u8* litScratch, u8* stage, ZjProf& pf,
                               const u8* litPool = nullptr, bool beside = false) {      // beside: stage 2 may still be at work — wait for every block's seqReady (bounded); ~0 - 1: gave up, nothing decided
    GRP_SERIAL(g) {
        ZDFrameMB const f = *fr;
        sh.err = 0; sh.hufValid = 0; sh.hufX2 = 0;
        sh.winLo = f.firstBlk; sh.bN = f.nBlk; sh.hasChecksum = f.hasChecksum; sh.contentSize = f.known ? f.contentSize : ~(u64)0; sh.tblOff[2] = f.frameEnd;
    }
    g.sync();
    u32 const first = ZJ_UNI(sh.winLo), nb = ZJ_UNI(sh.bN), frameEnd = ZJ_UNI(sh.tblOff[2]), hasChecksum = ZJ_UNI(sh.hasChecksum);      // (read now: the block loop reuses these words)
    u64 const content = zj_uni64(sh.contentSize);
    u64 const room = content != ~(u64)0 ? content : dstCap;               // how far the frame may write
    u32 r0 = 1, r1 = 4, r2 = 8;                                            // the repcode history at the start of the block (wave-uniform)
    u32 curHuf = ~0u;                                                      // the block whose Huffman table sits in sh.huf
    u64 op = 0;
    for (u32 b = 0; b < nb; b++) {
        const ZDBlk* const bk = blks + first + b;
        if (beside) {
            if (!zd_wait_block(g, sh, &bk->seqReady)) return ~(u64)0 - 1u;
        }
        GRP_SERIAL(g) {
            ZDBlk const k = *bk;
            sh.blkType = k.type; sh.hdrSize = k.blockOff; sh.blkSize = k.blockSize; sh.nbSeq = k.nbSeq; sh.blockSizeMax = k.blockSizeMax;
            sh.bLitTotal = k.regen; sh.seqDone = k.status; sh.tblOff[0] = k.seqLo; sh.tblOff[1] = k.seqHi;
            sh.rep[0] = k.rep[0]; sh.rep[1] = k.rep[1]; sh.rep[2] = k.rep[2];
            sh.hLo[0] = k.litLo; sh.hLo[1] = k.litHi; sh.hLo[2] = (litPool && k.litReady) ? 1u : 0u; sh.hLo[3] = k.hufBlk;
            if (k.hufBlk != ~0u) { ZDBlk const h = blks[k.hufBlk]; sh.bLitStart = h.blockOff; sh.bOutStart = h.blockSize; }
        }
        g.sync();
        if (ZJ_UNI(sh.seqDone)) return ~(u64)0;
        u32 const type = ZJ_UNI(sh.blkType), boff = ZJ_UNI(sh.hdrSize), bsize = ZJ_UNI(sh.blkSize), regen = ZJ_UNI(sh.bLitTotal);
        if (op + regen > room || op + regen > 0xFFFFFFFFull) return ~(u64)0;
        u32 const op32 = (u32)op;
        if (type == 0) { grp_copy_wide(g, dst + op, src + boff, bsize); zj_mem_order(); op += bsize; g.sync(); continue; }
        if (type == 1) {
            u32 const v = src[boff]; u64 const w = 0x0101010101010101ull * v;
            GRP_FOR(g, j, regen >> 3) st64(dst + op + 8u * j, w);
            GRP_FOR(g, j, regen & 7u) dst[op + (regen & ~7u) + j] = (u8)v;
            zj_mem_order(); op += regen; g.sync(); continue;
        }
        const u8* const bsrc = src + boff;
        u32 const nbSeq = ZJ_UNI(sh.nbSeq);
        bool const havePre = ZJ_UNI(sh.hLo[2]) != 0u;
        const u8* const preLit = havePre ? litPool + (((u64)ZJ_UNI(sh.hLo[1]) << 32) | ZJ_UNI(sh.hLo[0])) : nullptr;
        u32 const lt0 = ZJ_UNI(bsrc[0]) & 3u, hufBlk = ZJ_UNI(sh.hLo[3]);
        if (!havePre && lt0 == 3u && hufBlk != ~0u && curHuf != hufBlk) {
            // treeless literals stage 2b did not serve, while the table in LDS is not the one they use (its block's literals were served there): rebuild it
            u32 const bmaxKeep = ZJ_UNI(sh.blockSizeMax);
            const u8* const t = zd_block_literals(g, sh, src + ZJ_UNI(sh.bLitStart), ZJ_UNI(sh.bOutStart), nullptr, pf, ~0u, nullptr, true);
            if (t == nullptr || ZJ_UNI(sh.err)) return ~(u64)0;
            GRP_SERIAL(g) { sh.blockSizeMax = bmaxKeep; }
            g.sync();
            curHuf = hufBlk;
        }
        if (havePre && lt0 == 3u) { GRP_SERIAL(g) { sh.hufValid = 1; } g.sync(); }      // (the header parse asks for a table; stage 2b had it, and nothing below reads one)
        const u8* const lit = zd_block_literals(g, sh, bsrc, bsize, litScratch, pf, ~0u, preLit, false, stage);      // (stage: idle until the block is executed; ZD_HP_LDS <= its 5 KiB)
        if (lit == nullptr || ZJ_UNI(sh.err)) return ~(u64)0;
        if (!havePre && lt0 == 2u) curHuf = first + b;                  // (the table this block described is the one in LDS now)
        u32 const litSize = ZJ_UNI(sh.litSize);
        u32 const litAvail = (ZJ_UNI(sh.litType) == 0) ? bsize - ZJ_UNI(sh.litHdr) : ((havePre && ZJ_UNI(sh.litType) >= 2u) ? ((litSize + 63u) & ~31u) : ZD_LIT_SCRATCH);
        const u64* const seqs = pool + (((u64)ZJ_UNI(sh.tblOff[1]) << 32) | ZJ_UNI(sh.tblOff[0]));
        u32 lp = 0, opb = op32;
        for (u32 sb = 0; sb < nbSeq; sb += ZD_SEQ_BATCH) {
            u32 const cnt = zj_min(ZD_SEQ_BATCH, nbSeq - sb);
            GRP_SERIAL(g) { sh.bN = 0; }
            g.sync();
            GRP_FOR(g, k, cnt) {
                u64 const q = seqs[sb + k];
                u32 off = (u32)(q >> 36);
                if (off & ZD_SYM_REC) {                              // an entry of the history this block started from, minus d
                    u32 const slot = (off >> 25) & 3u, d = off & 0x1FFFFFFu, v = slot == 0u ? r0 : (slot == 1u ? r1 : r2);
                    if (v <= d) { sh.bN = 1; off = 1; } else off = v - d;     // (zero or below: the reference's "offset forced to -1")
                }
                sh.sLit[k] = (u32)q & 0x3FFFFu; sh.sMl[k] = (u32)(q >> 18) & 0x3FFFFu; sh.sOff[k] = off;
            }
            g.sync();
            if (ZJ_UNI(sh.bN)) return ~(u64)0;
            u32 lt, ot;
            zd_execute_batch<false, G, true>(g, sh, dst, lit, cnt, lp, opb, lt, ot, stage, litAvail, nullptr);     // (checks every offset against its output position)
            if (ZJ_UNI(sh.err)) return ~(u64)0;
            lp += lt; opb += ot;
            g.sync();
        }
        {   u32 const rest = litSize - lp;               // stage 2 checked lp <= litSize and the block's total
            if ((u64)(opb - op32) + rest != regen) return ~(u64)0;
            grp_copy_wide(g, dst + opb, lit + lp, rest);
            zj_mem_order();
        }
        g.sync();
        op += regen;
        {   // the history after the block: its entries in terms of r0 .. r2 where they are symbolic
            u32 const e0 = ZJ_UNI(sh.rep[0]), e1 = ZJ_UNI(sh.rep[1]), e2 = ZJ_UNI(sh.rep[2]);
            u32 n[3]; u32 const e[3] = {e0, e1, e2}; bool bad = false;
            for (u32 t = 0; t < 3; t++) {
                if (e[t] & ZD_SYM) { u32 const slot = (e[t] >> 29) & 3u, d = e[t] & 0x1FFFFFFFu, v = slot == 0u ? r0 : (slot == 1u ? r1 : r2); if (v <= d) bad = true; n[t] = v - d; }
                else n[t] = e[t];
            }
            if (bad) return ~(u64)0;
            r0 = n[0]; r1 = n[1]; r2 = n[2];
        }
    }
    if (content != ~(u64)0 && op != content) return ~(u64)0;
    if (hasChecksum) {
        if ((u32)zj_xxh64(g, dst, (u32)op) != ZJ_UNI(ld32(src + frameEnd))) return ~(u64)0;
    }
    return op;
}
