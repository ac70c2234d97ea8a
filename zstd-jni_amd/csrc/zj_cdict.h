// zj_cdict.h — compression dictionaries on the GPU: what ZstdDictCompress holds (ZSTD_createCDict,
// reference N/jni_fast_zstd.c:26,47-49) and the attach-mode match finders that search it.
//
// Replaces, for batches of independent small buffers,
//   ZstdCompressCtx.loadDict(ZstdDictCompress) + compress*   -> ZSTD_CCtx_refCDict + ZSTD_compress2   (N/jni_fast_zstd.c:325-336, :586-640)
//   Zstd.compress(dst, src, ZstdDictCompress)                -> ZSTD_compress_usingCDict               (N/jni_fast_zstd.c:171-216)
// i.e. N/compress/zstd_compress.c:5710-5719 (cParams of the dictionary), :5061-5155 (ZSTD_loadCEntropy), :4902-5040
// (ZSTD_loadDictionaryContent), N/compress/zstd_fast.c:16-50 and zstd_double_fast.c:18-53 (tagged table fill),
// zstd_compress.c:2309-2385 (attach decision and working parameters), zstd_fast.c:483-679 and
// zstd_double_fast.c:328-547 (the dictMatchState searches).  N/ = src/main/native/.
//
// Scope: the attach range — sources up to 8 KiB (fast) / 16 KiB (double-fast), where the reference searches the
// dictionary's own tables in place (ZSTD_shouldAttachDict).  Larger sources with a dictionary make the reference copy
// or reload the dictionary into the working tables (extDict search); those get ZJ_E_PARAM_UNSUPPORTED here.
#pragma once
#include "zj_decode.h"
#include "zj_encode.h"

#define ZC_TAG_BITS 8u                       /* ZSTD_SHORT_CACHE_TAG_BITS (N/compress/zstd_compress_internal.h:1482) */

// ZSTD_getCParams_internal(level, CONTENTSIZE_UNKNOWN, dictSize, ZSTD_cpm_createCDict) for levels 1..3
// (zstd_compress.c:7759-7786 row choice: rSize = dictSize + 499; clevels.h:25-130; ZSTD_adjustCParams_internal :1473-1600)
ZJ_HD ZEParams ze_cdict_params(u32 level, u32 dictSize) {
    u64 const rSize = (u64)dictSize + 499u;
    u32 w, c, h, mm, st;
    if (rSize <= (16u << 10)) { w = 14; c = 14; h = 15; mm = (level == 1) ? 5 : 4; st = (level == 3) ? 2 : 1; }
    else if (rSize <= (128u << 10)) { w = 17; if (level == 1) { c = 12; h = 13; mm = 6; st = 1; } else if (level == 2) { c = 13; h = 15; mm = 5; st = 1; } else { c = 15; h = 16; mm = 5; st = 2; } }
    else if (rSize <= (256u << 10)) { w = 18; if (level == 1) { c = 13; h = 14; mm = 6; st = 1; } else if (level == 2) { c = 14; h = 14; mm = 5; st = 2; } else { c = 16; h = 16; mm = 4; st = 2; } }
    else { if (level == 1) { w = 19; c = 13; h = 14; mm = 7; st = 1; } else if (level == 2) { w = 20; c = 15; h = 16; mm = 6; st = 1; } else { w = 21; c = 16; h = 17; mm = 5; st = 2; } }
    {   u32 const tSize = 513u + dictSize;                       // unknown source size -> minSrcSize
        u32 const srcLog = zj_hibit(tSize - 1) + 1;
        if (w > srcLog) w = srcLog;
        u32 dawl = w;                                            // ZSTD_dictAndWindowLog
        if (((u64)1 << w) < (u64)dictSize + 513u) dawl = zj_hibit((u32)(dictSize + (1u << w)) - 1) + 1;
        if (h > dawl + 1) h = dawl + 1;
        if (c > dawl) c = dawl;
        if (w < 10) w = 10;
        if (h > 32 - ZC_TAG_BITS) h = 32 - ZC_TAG_BITS;
        if (c > 32 - ZC_TAG_BITS) c = 32 - ZC_TAG_BITS;
    }
    ZEParams p; p.windowLog = w; p.chainLog = c; p.hashLog = h; p.minMatch = mm; p.strategy = st;
    return p;
}
ZJ_HD u32 ze_cdict_table_entries(const ZEParams& p) { return (1u << p.hashLog) + (p.strategy == 2 ? (1u << p.chainLog) : 0u); }
// largest source the reference compresses against the dictionary's tables in place (attachDictSizeCutoffs, zstd_compress.c:2296-2307)
ZJ_HD u32 ze_attach_cutoff(u32 strategy) { return strategy == 2 ? (16u << 10) : (8u << 10); }
// working-context parameters in attach mode: the dictionary's, resized for the source alone (zstd_compress.c:2338-2347)
ZJ_HD ZEParams ze_attach_params(const ZEParams& cd, u32 srcSize) {
    ZEParams p = cd;
    u32 const srcLog = (srcSize < 64u) ? 6u : zj_hibit(srcSize - 1) + 1;
    if (p.windowLog > srcLog) p.windowLog = srcLog;
    if (p.hashLog > p.windowLog + 1) p.hashLog = p.windowLog + 1;
    if (p.chainLog > p.windowLog) p.chainLog = p.windowLog;
    if (p.windowLog < 10) p.windowLog = 10;
    return p;
}
// per-frame table bytes of the attach-mode search (u16 entries: position + 1 <= 16 385)
#define ZC_MAX_SRC (16u << 10)
#define ZC_TABLE_STRIDE (((1u << 15) + (1u << 14)) * 2u)

// ZSTD_dictNCountRepeat (zstd_compress.c:5047-5059)
ZJ_DEV u32 ze_ncount_repeat(const short* norm, u32 dictMaxSV, u32 maxSV) {
    if (dictMaxSV < maxSV) return ZC_REPEAT_CHECK;
    for (u32 s = 0; s <= maxSV; s++) if (norm[s] == 0) return ZC_REPEAT_CHECK;
    return ZC_REPEAT_VALID;
}

// One workgroup digests the dictionary: ZSTD_initCDict_internal -> ZSTD_compress_insertDictionary (zstd_compress.c:5551-5603,
// :5194-5228).  `out` (header + zeroed tables + raw bytes) is in HBM; `sh`/`e` are LDS scratch.
template <class G>
ZJ_DEV void ze_cdict_digest(const G& g, ZDecShared& sh, ZEEntropy& e, u32 dictSize, u32 level, ZECDictDev* out) {
    const u8* const dict = (const u8*)out + out->rawOff;
    ZEParams const cp = ze_cdict_params(level, dictSize);
    GRP_SERIAL(g) {
        u32 err = 0, contentOff = 0, hasEntropy = 0, dictID = 0;
        out->level = level; out->windowLog = cp.windowLog; out->chainLog = cp.chainLog; out->hashLog = cp.hashLog; out->minMatch = cp.minMatch; out->strategy = cp.strategy;
        out->rep[0] = 1; out->rep[1] = 4; out->rep[2] = 8;                        // ZSTD_reset_compressedBlockState
        out->hufRepeat = ZC_REPEAT_NONE; out->llRepeat = ZC_REPEAT_NONE; out->ofRepeat = ZC_REPEAT_NONE; out->mlRepeat = ZC_REPEAT_NONE;
        out->hufMaxSV = 0; out->hufLog = 0;
        if (dictSize < 8) err = ZJ_E_DICT_WRONG;
        else if (ld32(dict) == 0xEC30A437u) {                                    // ZSTD_loadZstdDictionary -> ZSTD_loadCEntropy
            u32 pos = 8, nbSym = 0;
            dictID = ld32(dict + 4);
            {   u32 const h = zd_huf_read_weights(sh, dict + pos, dictSize - pos, &nbSym);      // HUF_readCTable (huf_compress.c:292-345)
                if (!h || h > dictSize - pos) err = ZJ_E_DICT_CORRUPTED;
                else {
                    u32 const tl = sh.hufLog; bool hasZero = false;
                    u16* const nbPerRank = e.cumul; u16* const valPerRank = e.cumul + 16;
                    for (u32 r = 0; r < 32; r++) e.cumul[r] = 0;
                    for (u32 n = 0; n < 256; n++) { out->hufNbBits[n] = 0; out->hufVal[n] = 0; }
                    for (u32 n = 0; n < nbSym; n++) { u32 const w = sh.weights[n]; if (!w) hasZero = true; out->hufNbBits[n] = w ? (u8)(tl + 1 - w) : 0; nbPerRank[w ? tl + 1 - w : 0]++; }
                    {   u16 min = 0; for (u32 n = tl; n > 0; n--) { valPerRank[n] = min; min += nbPerRank[n]; min >>= 1; } }
                    for (u32 n = 0; n < nbSym; n++) { u32 const nb = out->hufNbBits[n]; out->hufVal[n] = nb ? valPerRank[nb]++ : 0; }
                    out->hufMaxSV = nbSym - 1; out->hufLog = tl;
                    out->hufRepeat = (!hasZero && nbSym == 256) ? ZC_REPEAT_VALID : ZC_REPEAT_CHECK;
                    pos += h;
                }
            }
            u32 ofMaxRead = 31;
            short* const ofNorm = sh.norm[0]; short* norm = sh.norm[1];         // (the three share arrays stay: the encoding tables are built by the wave behind this block)
            if (!err) {                                                           // offset codes: table built over all 32 symbols
                u32 tl = 0; u32 const h = zd_read_ncount(ofNorm, &ofMaxRead, &tl, dict + pos, dictSize - pos);
                if (!h || h > dictSize - pos || tl > 8) err = ZJ_E_DICT_CORRUPTED;
                else { sh.tblMax[1] = 31; sh.tblLog[1] = tl; pos += h; }
            }
            if (!err) {                                                           // match lengths
                u32 max = 52, tl = 0; u32 const h = zd_read_ncount(norm, &max, &tl, dict + pos, dictSize - pos);
                if (!h || h > dictSize - pos || tl > 9) err = ZJ_E_DICT_CORRUPTED;
                else { sh.tblMax[2] = max; sh.tblLog[2] = tl; out->mlRepeat = ze_ncount_repeat(norm, max, 52); pos += h; }
            }
            norm = sh.norm[2];
            if (!err) {                                                           // literal lengths
                u32 max = 35, tl = 0; u32 const h = zd_read_ncount(norm, &max, &tl, dict + pos, dictSize - pos);
                if (!h || h > dictSize - pos || tl > 9) err = ZJ_E_DICT_CORRUPTED;
                else { sh.tblMax[0] = max; sh.tblLog[0] = tl; out->llRepeat = ze_ncount_repeat(norm, max, 35); pos += h; }
            }
            if (!err && pos + 12 > dictSize) err = ZJ_E_DICT_CORRUPTED;
            if (!err) {
                u32 const content = dictSize - (pos + 12);
                u32 const offcodeMax = zj_hibit(content + (128u << 10));
                out->ofRepeat = ze_ncount_repeat(ofNorm, ofMaxRead, offcodeMax < 31 ? offcodeMax : 31);
                for (u32 i = 0; i < 3; i++) { u32 const r = ld32(dict + pos + 4 * i); if (r == 0 || r > content) err = ZJ_E_DICT_CORRUPTED; out->rep[i] = r; }
                contentOff = pos + 12; hasEntropy = 1;
            }
        }
        // ZSTD_loadDictionaryContent: tagged indices leave 24 bits for the position
        u32 content = dictSize - contentOff;
        {   u32 const maxDict = (1u << (32 - ZC_TAG_BITS)) - 2u;
            if (!err && content > maxDict) { contentOff += content - maxDict; content = maxDict; } }
        u32 fillStart = 0;
        {   u32 const lg = zj_max(cp.hashLog + 3, cp.chainLog + 1);
            if (lg < 31 && content > (1u << lg)) fillStart = content - (1u << lg); }
        out->status = err; out->dictID = dictID; out->contentOff = contentOff; out->contentSize = content; out->hasEntropy = hasEntropy; out->fillStart = fillStart;
        sh.err = err; sh.litSize = content; sh.litCSize = fillStart; sh.seqValid = hasEntropy;
    }
    zj_mem_order();
    g.sync();
    if (ZJ_UNI(sh.err)) return;
    if (ZJ_UNI(sh.seqValid)) {                                                   // the dictionary's three encoding tables (ZSTD_loadCEntropy: FSE_buildCTable_wksp x 3), by the wave
        for (u32 t = 0; t < 3u; t++)                                             // fse[0] LL <- norm[2], fse[1] OF <- norm[0], fse[2] ML <- norm[1]
            ze_tans_table(g, out->fse[t], sh.norm[t == 0 ? 2 : (t == 1 ? 0 : 1)], ZJ_UNI(sh.tblMax[t]), ZJ_UNI(sh.tblLog[t]), e.tableSymbol, (u32*)e.rankBase, &e.hist[0][0]);
        zj_mem_order();
        g.sync();
    }
    // ---- table fill (ZSTD_fillHashTableForCDict / ZSTD_fillDoubleHashTableForCDict, dtlm_full): in position order, every
    //      third position always overwrites its buckets; the two positions after it fill a bucket (the long table for
    //      double-fast) only if it is still empty.  Order-free form: a bucket ends with the LAST every-third position that
    //      hashes to it if there is one (atomic max over packed index|tag), else with the FIRST other position (CAS-min). ----
    u32 const content = ZJ_UNI(sh.litSize), fillStart = ZJ_UNI(sh.litCSize);
    if (content - fillStart <= 8) return;                                        // HASH_READ_SIZE
    const u8* const base = dict + out->contentOff;                               // content offset p <-> index p + 2
    u32* const tblL = (u32*)((u8*)out + out->tablesOff);
    u32* const tblS = tblL + (1u << cp.hashLog);
    u32 const last = content - 8;                                                // ip + 2 <= iend
    u32 const groups = (fillStart + 2 <= last) ? (last - 2 - fillStart) / 3 + 1 : 0;
    bool const dfast = cp.strategy == 2;
    u32 const mlsL = dfast ? 8u : cp.minMatch;
    GRP_FOR(g, gi, groups) {
        u32 const p = fillStart + 3 * gi;
        u32 const ht = ze_hash(base + p, cp.hashLog + ZC_TAG_BITS, mlsL);
        atomicMax(&tblL[ht >> ZC_TAG_BITS], ((p + 2) << ZC_TAG_BITS) | (ht & 0xFFu));
        if (dfast) { u32 const hs = ze_hash(base + p, cp.chainLog + ZC_TAG_BITS, cp.minMatch); atomicMax(&tblS[hs >> ZC_TAG_BITS], ((p + 2) << ZC_TAG_BITS) | (hs & 0xFFu)); }
    }
    zj_mem_order();
    g.sync();
    GRP_FOR(g, gi, groups) {
        for (u32 i = 1; i < 3; i++) {
            u32 const p = fillStart + 3 * gi + i;
            u32 const ht = ze_hash(base + p, cp.hashLog + ZC_TAG_BITS, mlsL);
            u32 const mine = ((p + 2) << ZC_TAG_BITS) | (ht & 0xFFu);
            u32* const slot = &tblL[ht >> ZC_TAG_BITS];
            u32 old = *slot;
            for (;;) {
                if (old != 0) { u32 const po = (old >> ZC_TAG_BITS) - 2; if ((po - fillStart) % 3 == 0 || po <= p) break; }
                u32 const prev = atomicCAS(slot, old, mine);
                if (prev == old) break;
                old = prev;
            }
        }
    }
    zj_mem_order();
    g.sync();
}

// ------------------------------------------------------------------ attach-mode searches ----------
// Positions: the dictionary content and the source form one index space (content offset c <-> index c + 2, source
// position p <-> index 2 + contentSize + p; zstd_compress.c:2349-2371 puts the working window right behind the
// dictionary's), so an offset is just the distance in that space.
struct ZEDms { const u8* content; u32 size; const u32* hashLong; const u32* hashSmall; u32 hlogL, hlogS; };
ZJ_DEV ZEDms ze_dms_of(const ZECDictDev* cd) {
    ZEDms d; d.content = ze_cdict_content(cd); d.size = cd->contentSize; d.hashLong = ze_cdict_tables(cd); d.hashSmall = d.hashLong + (1u << cd->hashLog);
    d.hlogL = cd->hashLog; d.hlogS = cd->chainLog; return d;
}
// ZSTD_count_2segments (zstd_compress_internal.h:879-895)
ZJ_DEV u32 ze_count2(const u8* ip, const u8* match, const u8* iEnd, const u8* mEnd, const u8* iStart) {
    const u8* const vEnd = (ip + (mEnd - match) < iEnd) ? ip + (mEnd - match) : iEnd;
    u32 const ml = ze_count(ip, match, vEnd);
    if (match + ml != mEnd) return ml;
    return ml + ze_count(ip + ml, iStart, iEnd);
}

// ZSTD_compressBlock_doubleFast_dictMatchState_generic (zstd_double_fast.c:328-547); own tables hold position + 1.
template <class E>
ZJ_DEV u32 ze_block_dfast_dms(ZEOut& o, const u8* src, u32 srcSize, u32 hBitsL, u32 hBitsS, u32 mls, typename E::T* hashLong, typename E::T* hashSmall,
                              const ZEDms& d, u32 rep0, u32 rep1) {
    const u8* const istart = src; const u8* const iend = src + srcSize; const u8* const ilimit = iend - 8;
    const u8* ip = istart; const u8* anchor = istart;
    const u8* const dictStart = d.content; const u8* const dictEnd = d.content + d.size;
    u32 const dsz = d.size;
    u32 off1 = rep0, off2 = rep1;
    // The bytes of the next position (known in advance when this one finds nothing: ip + ((ip - anchor) >> 8) + 1) are
    // requested together with this position's table entries, so a no-match step is two dependent round trips, not three.
    u64 wNext = (ip < ilimit) ? ld64(ip) : 0; const u8* wNextAt = ip;
    while (ip < ilimit) {
        ZE_COUNT_ITER();
        u32 mLength, offset;
        u64 const w = (wNextAt == ip) ? wNext : ld64(ip);
        const u8* const ipn = ip + ((ip - anchor) >> 8) + 1;
        u32 const h2 = ze_hash_w(w, hBitsL, 8), h = ze_hash_w(w, hBitsS, mls);
        u32 const dhtL = ze_hash_w(w, d.hlogL + ZC_TAG_BITS, 8), dhtS = ze_hash_w(w, d.hlogS + ZC_TAG_BITS, mls);
        u32 const dL = d.hashLong[dhtL >> ZC_TAG_BITS], dS = d.hashSmall[dhtS >> ZC_TAG_BITS];
        bool const tagL = (dL & 0xFFu) == (dhtL & 0xFFu), tagS = (dS & 0xFFu) == (dhtS & 0xFFu);
        u32 const curr = (u32)(ip - istart);
        u32 const eL = hashLong[h2], eS = hashSmall[h];
        const u8* match = istart;
        bool matchInDict = false;
        u32 const v = dsz + curr + 1 - off1;                                       // repcode candidate in the joint space
        const u8* const repMatch = v < dsz ? dictStart + v : istart + (v - dsz);
        u32 const repBytes = ld32(repMatch);
        wNext = (ipn < ilimit) ? ld64(ipn) : 0; wNextAt = ipn;
        hashLong[h2] = E::make(curr + 1, 0); hashSmall[h] = E::make(curr + 1, 0);
        if (((u32)(dsz - 1 - v) >= 3) && (repBytes == (u32)(w >> 8))) {
            const u8* const repEnd = v < dsz ? dictEnd : iend;
            mLength = ze_count2(ip + 1 + 4, repMatch + 4, iend, repEnd, istart) + 4;
            ip++;
            ze_store(o, (u32)(anchor - istart), (u32)(ip - anchor), 1, mLength);
            goto stored;
        }
        if (eL != 0 && ld64(istart + E::pos(eL) - 1) == w) {                        // prefix long match
            const u8* matchLong = istart + E::pos(eL) - 1;
            mLength = ze_count(ip + 8, matchLong + 8, iend) + 8;
            offset = (u32)(ip - matchLong);
            while (((ip > anchor) & (matchLong > istart)) && (ip[-1] == matchLong[-1])) { ip--; matchLong--; mLength++; }
            goto found;
        } else if (tagL) {                                                         // dictionary long match
            u32 const di = dL >> ZC_TAG_BITS;
            const u8* dm = dictStart + (di - 2);
            if (di > 2 && ld64(dm) == w) {
                mLength = ze_count2(ip + 8, dm + 8, iend, dictEnd, istart) + 8;
                offset = curr + 2 + dsz - di;
                while (((ip > anchor) & (dm > dictStart)) && (ip[-1] == dm[-1])) { ip--; dm--; mLength++; }
                goto found;
            }
        }
        if (E::pos(eS) > 1) {                                                      // matchIndexS > prefixLowestIndex
            match = istart + E::pos(eS) - 1;
            if (ld32(match) == (u32)w) goto next_long;
        } else if (tagS) {
            u32 const di = dS >> ZC_TAG_BITS;
            match = dictStart + (di - 2);
            if (di > 2 && ld32(match) == (u32)w) { matchInDict = true; goto next_long; }
        }
        ip = ipn;                                                                  // ip += ((ip - anchor) >> kSearchStrength) + 1
        continue;
next_long:
        {   u64 const w1 = ld64(ip + 1);
            u32 const hl3 = ze_hash_w(w1, hBitsL, 8);
            u32 const dht3 = ze_hash_w(w1, d.hlogL + ZC_TAG_BITS, 8);
            u32 const eL3 = hashLong[hl3];
            u32 const dL3 = d.hashLong[dht3 >> ZC_TAG_BITS];
            bool const tagL3 = (dL3 & 0xFFu) == (dht3 & 0xFFu);
            hashLong[hl3] = E::make(curr + 2, 0);
            if (eL3 != 0 && ld64(istart + E::pos(eL3) - 1) == w1) {
                const u8* m3 = istart + E::pos(eL3) - 1;
                mLength = ze_count(ip + 9, m3 + 8, iend) + 8;
                ip++;
                offset = (u32)(ip - m3);
                while (((ip > anchor) & (m3 > istart)) && (ip[-1] == m3[-1])) { ip--; m3--; mLength++; }
                goto found;
            } else if (tagL3) {
                u32 const di = dL3 >> ZC_TAG_BITS;
                const u8* dm = dictStart + (di - 2);
                if (di > 2 && ld64(dm) == w1) {
                    mLength = ze_count2(ip + 1 + 8, dm + 8, iend, dictEnd, istart) + 8;
                    ip++;
                    offset = curr + 1 + 2 + dsz - di;
                    while (((ip > anchor) & (dm > dictStart)) && (ip[-1] == dm[-1])) { ip--; dm--; mLength++; }
                    goto found;
                }
            }
        }
        if (matchInDict) {                                                         // explore the short match
            mLength = ze_count2(ip + 4, match + 4, iend, dictEnd, istart) + 4;
            offset = curr + dsz - (u32)(match - dictStart);
            while (((ip > anchor) & (match > dictStart)) && (ip[-1] == match[-1])) { ip--; match--; mLength++; }
        } else {
            mLength = ze_count(ip + 4, match + 4, iend) + 4;
            offset = (u32)(ip - match);
            while (((ip > anchor) & (match > istart)) && (ip[-1] == match[-1])) { ip--; match--; mLength++; }
        }
found:
        off2 = off1; off1 = offset;
        ze_store(o, (u32)(anchor - istart), (u32)(ip - anchor), offset + 3, mLength);
stored:
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            {   u32 const ins = curr + 2;
                hashLong[ze_hash(istart + ins, hBitsL, 8)] = E::make(ins + 1, 0);
                hashLong[ze_hash(ip - 2, hBitsL, 8)] = E::make((u32)(ip - 2 - istart) + 1, 0);
                hashSmall[ze_hash(istart + ins, hBitsS, mls)] = E::make(ins + 1, 0);
                hashSmall[ze_hash(ip - 1, hBitsS, mls)] = E::make((u32)(ip - 1 - istart) + 1, 0);
            }
            while (ip <= ilimit) {
                u32 const cur2 = (u32)(ip - istart);
                u32 const v2 = dsz + cur2 - off2;
                const u8* const rm2 = v2 < dsz ? dictStart + v2 : istart + (v2 - dsz);
                if (((u32)(dsz - 1 - v2) >= 3) && (ld32(rm2) == ld32(ip))) {
                    const u8* const repEnd2 = v2 < dsz ? dictEnd : iend;
                    u32 const rLength = ze_count2(ip + 4, rm2 + 4, iend, repEnd2, istart) + 4;
                    u32 const t = off2; off2 = off1; off1 = t;
                    ze_store(o, (u32)(anchor - istart), 0, 1, rLength);
                    hashSmall[ze_hash(ip, hBitsS, mls)] = E::make(cur2 + 1, 0);
                    hashLong[ze_hash(ip, hBitsL, 8)] = E::make(cur2 + 1, 0);
                    ip += rLength; anchor = ip;
                    continue;
                }
                break;
            }
        }
    }
    return (u32)(iend - anchor);
}

// ZSTD_compressBlock_fast_dictMatchState_generic (zstd_fast.c:483-679), stepSize 1 (targetLength 0 at levels 1-2)
template <class E>
ZJ_DEV u32 ze_block_fast_dms(ZEOut& o, const u8* src, u32 srcSize, u32 hlog, u32 mls, typename E::T* table, const ZEDms& d, u32 rep0, u32 rep1) {
    const u8* const istart = src; const u8* const iend = src + srcSize; const u8* const ilimit = iend - 8;
    const u8* ip0 = istart; const u8* ip1 = ip0 + 1; const u8* anchor = istart;
    const u8* const dictStart = d.content; const u8* const dictEnd = d.content + d.size;
    u32 const dsz = d.size;
    u32 off1 = rep0, off2 = rep1;
    while (ip1 <= ilimit) {
        u32 mLength = 0;
        u32 hash0 = ze_hash(ip0, hlog, mls);
        u32 dht0 = ze_hash(ip0, d.hlogL + ZC_TAG_BITS, mls);
        u32 dE = d.hashLong[dht0 >> ZC_TAG_BITS];
        bool dTag = (dE & 0xFFu) == (dht0 & 0xFFu);
        u32 mE = table[hash0];
        u32 curr = (u32)(ip0 - istart);
        u32 step = 1;
        const u8* nextStep = ip0 + 256;
        bool got = false;
        for (;;) {
            ZE_COUNT_ITER();
            u32 const v = dsz + curr + 1 - off1;
            const u8* const repMatch = v < dsz ? dictStart + v : istart + (v - dsz);
            u32 const hash1 = ze_hash(ip1, hlog, mls);
            u32 const dht1 = ze_hash(ip1, d.hlogL + ZC_TAG_BITS, mls);
            table[hash0] = E::make(curr + 1, 0);
            if (((u32)(dsz - 1 - v) >= 3) && (ld32(repMatch) == ld32(ip0 + 1))) {
                const u8* const repEnd = v < dsz ? dictEnd : iend;
                mLength = ze_count2(ip0 + 1 + 4, repMatch + 4, iend, repEnd, istart) + 4;
                ip0++;
                ze_store(o, (u32)(anchor - istart), (u32)(ip0 - anchor), 1, mLength);
                got = true; break;
            }
            if (dTag) {
                u32 const di = dE >> ZC_TAG_BITS;
                const u8* dm = dictStart + (di - 2);
                if (di > 2 && ld32(dm) == ld32(ip0)) {
                    if (E::pos(mE) <= 1) {                                         // matchIndex <= prefixStartIndex
                        u32 const offset = curr + 2 + dsz - di;
                        mLength = ze_count2(ip0 + 4, dm + 4, iend, dictEnd, istart) + 4;
                        while (((ip0 > anchor) & (dm > dictStart)) && (ip0[-1] == dm[-1])) { ip0--; dm--; mLength++; }
                        off2 = off1; off1 = offset;
                        ze_store(o, (u32)(anchor - istart), (u32)(ip0 - anchor), offset + 3, mLength);
                        got = true; break;
                    }
                }
            }
            if (mE != 0 && ld32(istart + E::pos(mE) - 1) == ld32(ip0)) {            // ZSTD_match4Found_cmov: index >= prefixStartIndex
                const u8* match = istart + E::pos(mE) - 1;
                u32 const offset = (u32)(ip0 - match);
                mLength = ze_count(ip0 + 4, match + 4, iend) + 4;
                while (((ip0 > anchor) & (match > istart)) && (ip0[-1] == match[-1])) { ip0--; match--; mLength++; }
                off2 = off1; off1 = offset;
                ze_store(o, (u32)(anchor - istart), (u32)(ip0 - anchor), offset + 3, mLength);
                got = true; break;
            }
            dE = d.hashLong[dht1 >> ZC_TAG_BITS];
            dTag = (dE & 0xFFu) == (dht1 & 0xFFu);
            mE = table[hash1];
            if (ip1 >= nextStep) { step++; nextStep += 256; }
            ip0 = ip1; ip1 = ip1 + step;
            if (ip1 > ilimit) break;
            curr = (u32)(ip0 - istart);
            hash0 = hash1;
        }
        if (!got) break;
        ip0 += mLength; anchor = ip0;
        if (ip0 <= ilimit) {
            table[ze_hash(istart + curr + 2, hlog, mls)] = E::make(curr + 2 + 1, 0);
            table[ze_hash(ip0 - 2, hlog, mls)] = E::make((u32)(ip0 - 2 - istart) + 1, 0);
            while (ip0 <= ilimit) {
                u32 const cur2 = (u32)(ip0 - istart);
                u32 const v2 = dsz + cur2 - off2;
                const u8* const rm2 = v2 < dsz ? dictStart + v2 : istart + (v2 - dsz);
                if (((u32)(dsz - 1 - v2) >= 3) && (ld32(rm2) == ld32(ip0))) {
                    const u8* const repEnd2 = v2 < dsz ? dictEnd : iend;
                    u32 const rLength = ze_count2(ip0 + 4, rm2 + 4, iend, repEnd2, istart) + 4;
                    u32 const t = off2; off2 = off1; off1 = t;
                    ze_store(o, (u32)(anchor - istart), 0, 1, rLength);
                    table[ze_hash(ip0, hlog, mls)] = E::make(cur2 + 1, 0);
                    ip0 += rLength; anchor = ip0;
                    continue;
                }
                break;
            }
        }
        ip1 = ip0 + 1;
    }
    return (u32)(iend - anchor);
}

// One frame's sequences against a dictionary (plain per-lane loop): records + meta {nbSeq, litSize, lastLL}.
// ------------------------------------------------------------------ copy-mode searches (sources beyond the attach range) ----------
// Beyond the attach cutoff the reference COPIES the dictionary's tables into the working context (tags stripped,
// ZSTD_resetCCtx_byCopyingCDict, zstd_compress.c:2402-2468) and parses with the dictionary as an external segment of the window:
// ZSTD_compressBlock_fast_extDict_generic (zstd_fast.c:708-960) / ZSTD_compressBlock_doubleFast_extDict_generic
// (zstd_double_fast.c:608-757).  Same index space as above: dictionary content offset c <-> index c + 2, source position p <-> index
// 2 + dictSize + p; the whole dictionary is valid (loadedDictEnd != 0: lowest index = 2).  Tables: plain u32 indices.
struct ZEExt { const u8* src; const u8* dict; u32 dictSize, prefixStartIndex; };
ZJ_DEV const u8* ze_ext_ptr(const ZEExt& x, u32 idx) { return idx < x.prefixStartIndex ? x.dict + (idx - 2u) : x.src + (idx - x.prefixStartIndex); }
ZJ_DEV bool ze_ext_overlap_ok(u32 prefixStartIndex, u32 repIndex) { return (u32)((prefixStartIndex - 1u) - repIndex) >= 3u; }   // ZSTD_index_overlap_check

ZJ_DEV u32 ze_block_fast_ext(ZEOut& o, const u8* src, u32 srcSize, const u8* dict, u32 dictSize, u32 hlog, u32 mls, u32* hashTable, u32 rep0, u32 rep1) {
    ZEExt x; x.src = src; x.dict = dict; x.dictSize = dictSize; x.prefixStartIndex = 2u + dictSize;
    u32 const dictStartIndex = 2u, prefixStartIndex = x.prefixStartIndex;
    const u8* const istart = src; const u8* const iend = src + srcSize; const u8* const ilimit = iend - 8;
    const u8* const dictStart = dict; const u8* const dictEnd = dict + dictSize; const u8* const prefixStart = src;
    const u8* anchor = istart;
    u32 off1 = rep0, off2 = rep1;
    const u8* ip0 = istart; const u8* ip1; const u8* ip2; const u8* ip3;
    #define ZX_IDX(p) ((u32)((p) - istart) + prefixStartIndex)
    {   u32 const curr = ZX_IDX(ip0), maxRep = curr - dictStartIndex;
        if (off2 >= maxRep) off2 = 0;
        if (off1 >= maxRep) off1 = 0; }
    for (;;) {                                                 // _start
        u32 step = 2; const u8* nextStep = ip0 + 128;          // stepSize = targetLength + !targetLength + 1 with targetLength 0; kStepIncr
        ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
        if (ip3 >= ilimit) break;
        u32 hash0 = ze_hash(ip0, hlog, mls), hash1 = ze_hash(ip1, hlog, mls);
        u32 idx = hashTable[hash0];
        u32 current0 = 0, offcode = 0, mLength = 0; const u8* match0 = nullptr; const u8* matchEnd = nullptr;
        int found = 0;                                         // 1 = repcode (_match), 2 = table match (_offset)
        do {
            {   u32 const current2 = ZX_IDX(ip2), repIndex = current2 - off1;
                u32 rval;
                if (((u32)(prefixStartIndex - repIndex) >= 4u) & (off1 > 0)) rval = ld32(ze_ext_ptr(x, repIndex));
                else rval = ld32(ip2) ^ 1u;
                current0 = ZX_IDX(ip0); hashTable[hash0] = current0;
                if (ld32(ip2) == rval) {
                    ip0 = ip2; match0 = ze_ext_ptr(x, repIndex); matchEnd = repIndex < prefixStartIndex ? dictEnd : iend;
                    mLength = (ip0[-1] == match0[-1]); ip0 -= mLength; match0 -= mLength;
                    offcode = 1u; mLength += 4; found = 1; break;
                } }
            {   u32 const mval = idx >= dictStartIndex ? ld32(ze_ext_ptr(x, idx)) : (ld32(ip0) ^ 1u);
                if (ld32(ip0) == mval) { found = 2; break; } }
            idx = hashTable[hash1];
            hash0 = hash1; hash1 = ze_hash(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            current0 = ZX_IDX(ip0); hashTable[hash0] = current0;
            {   u32 const mval = idx >= dictStartIndex ? ld32(ze_ext_ptr(x, idx)) : (ld32(ip0) ^ 1u);
                if (ld32(ip0) == mval) { found = 2; break; } }
            idx = hashTable[hash1];
            hash0 = hash1; hash1 = ze_hash(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
            if (ip2 >= nextStep) { step++; nextStep += 128; }
        } while (ip3 < ilimit);
        if (!found) break;                                     // _cleanup
        if (found == 2) {                                      // _offset
            u32 const offset = current0 - idx;
            const u8* const lowMatchPtr = idx < prefixStartIndex ? dictStart : prefixStart;
            matchEnd = idx < prefixStartIndex ? dictEnd : iend;
            match0 = ze_ext_ptr(x, idx);
            off2 = off1; off1 = offset; offcode = offset + 3u; mLength = 4;
            while (((ip0 > anchor) & (match0 > lowMatchPtr)) && (ip0[-1] == match0[-1])) { ip0--; match0--; mLength++; }
        }
        mLength += ze_count2(ip0 + mLength, match0 + mLength, iend, matchEnd, prefixStart);       // _match
        ze_store(o, (u32)(anchor - istart), (u32)(ip0 - anchor), offcode, mLength);
        ip0 += mLength; anchor = ip0;
        if (ip1 < ip0) hashTable[hash1] = ZX_IDX(ip1);
        if (ip0 <= ilimit) {
            {   const u8* const p2 = ze_ext_ptr(x, current0 + 2u); hashTable[ze_hash(p2, hlog, mls)] = current0 + 2u; }
            hashTable[ze_hash(ip0 - 2, hlog, mls)] = ZX_IDX(ip0 - 2);
            while (ip0 <= ilimit) {
                u32 const repIndex2 = ZX_IDX(ip0) - off2;
                if ((ze_ext_overlap_ok(prefixStartIndex, repIndex2) & (off2 > 0)) && (ld32(ze_ext_ptr(x, repIndex2)) == ld32(ip0))) {
                    const u8* const repMatch2 = ze_ext_ptr(x, repIndex2);
                    const u8* const repEnd2 = repIndex2 < prefixStartIndex ? dictEnd : iend;
                    u32 const repLength2 = ze_count2(ip0 + 4, repMatch2 + 4, iend, repEnd2, prefixStart) + 4u;
                    { u32 const t = off2; off2 = off1; off1 = t; }
                    ze_store(o, (u32)(anchor - istart), 0u, 1u, repLength2);
                    hashTable[ze_hash(ip0, hlog, mls)] = ZX_IDX(ip0);
                    ip0 += repLength2; anchor = ip0;
                    continue;
                }
                break;
            }
        }
    }
    #undef ZX_IDX
    return (u32)(iend - anchor);
}

ZJ_DEV u32 ze_block_dfast_ext(ZEOut& o, const u8* src, u32 srcSize, const u8* dict, u32 dictSize, u32 hBitsL, u32 hBitsS, u32 mls, u32* hashLong, u32* hashSmall,
                              u32 rep0, u32 rep1) {
    ZEExt x; x.src = src; x.dict = dict; x.dictSize = dictSize; x.prefixStartIndex = 2u + dictSize;
    u32 const dictStartIndex = 2u, prefixStartIndex = x.prefixStartIndex;
    const u8* const istart = src; const u8* const iend = src + srcSize; const u8* const ilimit = iend - 8;
    const u8* const dictStart = dict; const u8* const dictEnd = dict + dictSize; const u8* const prefixStart = src;
    const u8* ip = istart; const u8* anchor = istart;
    u32 off1 = rep0, off2 = rep1;
    #define ZX_IDX(p) ((u32)((p) - istart) + prefixStartIndex)
    while (ip < ilimit) {
        u32 const hSmall = ze_hash(ip, hBitsS, mls), matchIndex = hashSmall[hSmall];
        const u8* match = ze_ext_ptr(x, matchIndex >= dictStartIndex ? matchIndex : dictStartIndex);
        u32 const hLong = ze_hash(ip, hBitsL, 8), matchLongIndex = hashLong[hLong];
        const u8* matchLong = ze_ext_ptr(x, matchLongIndex >= dictStartIndex ? matchLongIndex : dictStartIndex);
        u32 const curr = ZX_IDX(ip), repIndex = curr + 1u - off1;
        u32 mLength;
        hashSmall[hSmall] = hashLong[hLong] = curr;
        if ((ze_ext_overlap_ok(prefixStartIndex, repIndex) & (off1 <= curr + 1u - dictStartIndex)) && (ld32(ze_ext_ptr(x, repIndex)) == ld32(ip + 1))) {
            const u8* const repMatch = ze_ext_ptr(x, repIndex);
            const u8* const repMatchEnd = repIndex < prefixStartIndex ? dictEnd : iend;
            mLength = ze_count2(ip + 1 + 4, repMatch + 4, iend, repMatchEnd, prefixStart) + 4u;
            ip++;
            ze_store(o, (u32)(anchor - istart), (u32)(ip - anchor), 1u, mLength);
        } else {
            if ((matchLongIndex > dictStartIndex) && (ld64(matchLong) == ld64(ip))) {
                const u8* const matchEnd = matchLongIndex < prefixStartIndex ? dictEnd : iend;
                const u8* const lowMatchPtr = matchLongIndex < prefixStartIndex ? dictStart : prefixStart;
                mLength = ze_count2(ip + 8, matchLong + 8, iend, matchEnd, prefixStart) + 8u;
                u32 const offset = curr - matchLongIndex;
                while (((ip > anchor) & (matchLong > lowMatchPtr)) && (ip[-1] == matchLong[-1])) { ip--; matchLong--; mLength++; }
                off2 = off1; off1 = offset;
                ze_store(o, (u32)(anchor - istart), (u32)(ip - anchor), offset + 3u, mLength);
            } else if ((matchIndex > dictStartIndex) && (ld32(match) == ld32(ip))) {
                u32 const h3 = ze_hash(ip + 1, hBitsL, 8), matchIndex3 = hashLong[h3];
                const u8* match3 = ze_ext_ptr(x, matchIndex3 >= dictStartIndex ? matchIndex3 : dictStartIndex);
                u32 offset;
                hashLong[h3] = curr + 1u;
                if ((matchIndex3 > dictStartIndex) && (ld64(match3) == ld64(ip + 1))) {
                    const u8* const matchEnd = matchIndex3 < prefixStartIndex ? dictEnd : iend;
                    const u8* const lowMatchPtr = matchIndex3 < prefixStartIndex ? dictStart : prefixStart;
                    mLength = ze_count2(ip + 9, match3 + 8, iend, matchEnd, prefixStart) + 8u;
                    ip++;
                    offset = curr + 1u - matchIndex3;
                    while (((ip > anchor) & (match3 > lowMatchPtr)) && (ip[-1] == match3[-1])) { ip--; match3--; mLength++; }
                } else {
                    const u8* const matchEnd = matchIndex < prefixStartIndex ? dictEnd : iend;
                    const u8* const lowMatchPtr = matchIndex < prefixStartIndex ? dictStart : prefixStart;
                    mLength = ze_count2(ip + 4, match + 4, iend, matchEnd, prefixStart) + 4u;
                    offset = curr - matchIndex;
                    while (((ip > anchor) & (match > lowMatchPtr)) && (ip[-1] == match[-1])) { ip--; match--; mLength++; }
                }
                off2 = off1; off1 = offset;
                ze_store(o, (u32)(anchor - istart), (u32)(ip - anchor), offset + 3u, mLength);
            } else {
                ip += ((u32)(ip - anchor) >> 8) + 1;
                continue;
            }
        }
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            {   u32 const indexToInsert = curr + 2u; const u8* const pi = ze_ext_ptr(x, indexToInsert);
                hashLong[ze_hash(pi, hBitsL, 8)] = indexToInsert;
                hashLong[ze_hash(ip - 2, hBitsL, 8)] = ZX_IDX(ip - 2);
                hashSmall[ze_hash(pi, hBitsS, mls)] = indexToInsert;
                hashSmall[ze_hash(ip - 1, hBitsS, mls)] = ZX_IDX(ip - 1); }
            while (ip <= ilimit) {
                u32 const current2 = ZX_IDX(ip), repIndex2 = current2 - off2;
                if ((ze_ext_overlap_ok(prefixStartIndex, repIndex2) & (off2 <= current2 - dictStartIndex)) && (ld32(ze_ext_ptr(x, repIndex2)) == ld32(ip))) {
                    const u8* const repMatch2 = ze_ext_ptr(x, repIndex2);
                    const u8* const repEnd2 = repIndex2 < prefixStartIndex ? dictEnd : iend;
                    u32 const repLength2 = ze_count2(ip + 4, repMatch2 + 4, iend, repEnd2, prefixStart) + 4u;
                    { u32 const t = off2; off2 = off1; off1 = t; }
                    ze_store(o, (u32)(anchor - istart), 0u, 1u, repLength2);
                    hashSmall[ze_hash(ip, hBitsS, mls)] = current2;
                    hashLong[ze_hash(ip, hBitsL, 8)] = current2;
                    ip += repLength2; anchor = ip;
                    continue;
                }
                break;
            }
        }
    }
    #undef ZX_IDX
    return (u32)(iend - anchor);
}
// sources the reference compresses with the dictionary's own parameters by copying its tables: beyond the attach cutoff, one block,
// and within ZSTD_compressBegin_internal's "use the CDict's parameters" rule (zstd_compress.c:5308-5320: source < 128 KiB or < 6 x dictionary)
ZJ_HD bool ze_cdict_copy_mode(u32 strategy, u32 srcSize, u32 dictContentSize) {
    return srcSize > ze_attach_cutoff(strategy) && srcSize <= (128u << 10) && (srcSize < (128u << 10) || (u64)srcSize < (u64)dictContentSize * 6u);
}

// copy mode, the two steps a workgroup takes before the entropy stage: (all lanes) the dictionary's tables into its slot, tags stripped
// (ZSTD_copyCDictTableIntoCCtx); (lane 0) the external-segment parse into the workgroup's record scratch.  meta = {nbSeq, litSize, lastLL}
template <class G>
ZJ_DEV void ze_cdict_copy_tables(const G& g, const ZECDictDev* cd, u32* slot) {
    const u32* const t = ze_cdict_tables(cd);
    u32 const entries = (1u << cd->hashLog) + (cd->strategy == 2 ? (1u << cd->chainLog) : 0u);
    GRP_FOR(g, i, entries) slot[i] = t[i] >> ZC_TAG_BITS;
}
ZJ_DEV void ze_cdict_copy_parse(const ZECDictDev* cd, const u8* src, u32 srcSize, u32* slot, u8* ws, u32* meta) {
    ZEOut o; o.seqs = (ZESeq*)(ws + ZE_WS_SEQ); o.litOff = (u32*)(ws + ZE_WS_BODY); o.n = 0; o.lit = 0;
    const u8* const dict = ze_cdict_content(cd);
    u32 const lastLL = cd->strategy == 1 ? ze_block_fast_ext(o, src, srcSize, dict, cd->contentSize, cd->hashLog, cd->minMatch, slot, cd->rep[0], cd->rep[1])
                                         : ze_block_dfast_ext(o, src, srcSize, dict, cd->contentSize, cd->hashLog, cd->chainLog, cd->minMatch, slot, slot + (1u << cd->hashLog), cd->rep[0], cd->rep[1]);
    meta[0] = o.n; meta[1] = o.lit + lastLL; meta[2] = lastLL;
}

// The caller has zeroed `table` (ZC_TABLE_STRIDE bytes) and checked srcSize <= ze_attach_cutoff().
ZJ_DEV void ze_match_lane_dict(const u8* src, u32 srcSize, const ZECDictDev* cd, u8* table, u8* fscratch, u32 maxSrc, u32* meta) {
    ZEOut o; o.seqs = (ZESeq*)fscratch; o.litOff = (u32*)(fscratch + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); o.n = 0; o.lit = 0;
    u32 lastLL = srcSize;
    if (srcSize >= 7) {                                                           // ZSTD_buildSeqStore: MIN_CBLOCK_SIZE + 3 + 1 + 1
        ZEParams cdp; cdp.windowLog = cd->windowLog; cdp.chainLog = cd->chainLog; cdp.hashLog = cd->hashLog; cdp.minMatch = cd->minMatch; cdp.strategy = cd->strategy;
        ZEParams const p = ze_attach_params(cdp, srcSize);
        ZEDms const d = ze_dms_of(cd);
        u16* const t = (u16*)table;
        if (p.strategy == 1) lastLL = ze_block_fast_dms<ZEEnt16>(o, src, srcSize, p.hashLog, p.minMatch, t, d, cd->rep[0], cd->rep[1]);
        else lastLL = ze_block_dfast_dms<ZEEnt16>(o, src, srcSize, p.hashLog, p.chainLog, p.minMatch, t, t + (1u << p.hashLog), d, cd->rep[0], cd->rep[1]);
    }
    meta[0] = o.n; meta[1] = o.lit + lastLL; meta[2] = lastLL;
}
