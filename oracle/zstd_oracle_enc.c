/* oracle/zstd_oracle_enc.c — plain-C restatement of the reference's one-shot compressor for the
 * path zstd-jni's ZstdCompressCtx.compress* reaches: ZSTD_compress2 at levels 1..3 (strategies
 * ZSTD_fast / ZSTD_dfast), no dictionary, inputs of at most one block (<= 128 KiB).
 *
 * TEST INFRASTRUCTURE ONLY (see zstd_oracle.h).  Parity: PINNED — byte-identical to oracle/_ref's
 * ZSTD_compress2 on seeded inputs, edge cases and the reference's xmlsmall -> xmlsmall-sized.zst
 * golden (tests/test_oracle.py).
 *
 * Reference files restated ("N/" = /root/reference/src/main/native/):
 *   parameters    N/compress/clevels.h:25-130, N/compress/zstd_compress.c:1473-1600, :7759-7786
 *   frame/block   N/compress/zstd_compress.c:4695-4745 (frame header), :4591-4692 (frame chunk),
 *                 :4383-4448 (block), :3264-3440 (seq store), :2888-3043 (entropy stage)
 *   fast          N/compress/zstd_fast.c:192-423
 *   double-fast   N/compress/zstd_double_fast.c:105-323
 *   literals      N/compress/zstd_compress_literals.c:129-235, N/compress/huf_compress.c (whole)
 *   sequences     N/compress/zstd_compress_sequences.c:157-382, N/compress/fse_compress.c (whole)
 *   histograms    N/compress/hist.c
 *
 * Scope notes: inputs > 128 KiB return ZSO_error_parameter_unsupported (multi-block frames carry
 * window/entropy state across blocks and use the pre-splitter; they stay on the reference path).
 * hashLog/chainLog overrides follow ZSTD_c_hashLog / ZSTD_c_chainLog semantics (override, then
 * ZSTD_adjustCParams_internal) — used to pin the GPU level-3 variant whose tables must fit LDS.
 */
#include "zstd_oracle.h"
#include <string.h>
#include <stdlib.h>

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64; typedef int32_t i32;

#define ERR(name) ZSO_ERR(ZSO_error_##name)
#define BLOCK_MAX (1u << 17)
#define MAXLL 35
#define MAXML 52
#define MAXOFF 31
#define HUF_LOG_LIT 11

static u32 rd32(const u8* p) { u32 v; memcpy(&v, p, 4); return v; }
static u64 rd64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
static void wr16(u8* p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); }
static void wr24(u8* p, u32 v) { wr16(p, v); p[2] = (u8)(v >> 16); }
static void wr32(u8* p, u32 v) { wr16(p, v); wr16(p + 2, v >> 16); }
static u32 hibit(u32 v) { return 31u - (u32)__builtin_clz(v); }

size_t zso_compress_bound(size_t s) { return s + (s >> 8) + (s < (128u << 10) ? (((128u << 10) - s) >> 11) : 0); }

/* ------------------------------------------------------------------ parameters ------------- */
typedef struct { u32 windowLog, chainLog, hashLog, minMatch, strategy; } CParams;   /* strategy 1 fast, 2 dfast */

static void adjust_cparams(CParams* c, u64 srcSize) {
    /* N/compress/zstd_compress.c:1553-1572 (known srcSize, no dictionary) */
    u32 const tSize = (u32)srcSize;
    u32 const srcLog = (tSize < (1u << 6)) ? 6 : hibit(tSize - 1) + 1;
    if (c->windowLog > srcLog) c->windowLog = srcLog;
    if (c->hashLog > c->windowLog + 1) c->hashLog = c->windowLog + 1;
    if (c->chainLog > c->windowLog) c->chainLog = c->windowLog;       /* cycleLog == chainLog below btlazy2 */
    if (c->windowLog < 10) c->windowLog = 10;                          /* ZSTD_WINDOWLOG_ABSOLUTEMIN */
}

static int get_cparams(CParams* c, int level, u64 srcSize, int hashLogOv, int chainLogOv) {
    /* rows 1..3 of the <=16 KB and <=128 KB tables, N/compress/clevels.h:81-83,107-109 */
    static const CParams t128[3] = { {17,12,13,6,1}, {17,13,15,5,1}, {17,15,16,5,2} };
    static const CParams t16[3]  = { {14,14,15,5,1}, {14,14,15,4,1}, {14,14,15,4,2} };
    if (level < 1 || level > 3 || srcSize > BLOCK_MAX) return -1;
    *c = (srcSize <= (16u << 10)) ? t16[level - 1] : t128[level - 1];
    adjust_cparams(c, srcSize);
    if (hashLogOv) c->hashLog = (u32)hashLogOv;
    if (chainLogOv) c->chainLog = (u32)chainLogOv;
    if (hashLogOv || chainLogOv) adjust_cparams(c, srcSize);           /* ZSTD_getCParamsFromCCtxParams: override then adjust */
    return 0;
}

/* ------------------------------------------------------------------ sequence store --------- */
typedef struct { u32 litLength, matchLength, offBase; } Seq;
typedef struct { Seq* seq; u32 nbSeq; u8* lit; u32 litSize; } SeqStore;

static void store_seq(SeqStore* ss, const u8* literals, u32 litLength, u32 offBase, u32 matchLength) {
    memcpy(ss->lit + ss->litSize, literals, litLength); ss->litSize += litLength;
    ss->seq[ss->nbSeq].litLength = litLength; ss->seq[ss->nbSeq].matchLength = matchLength; ss->seq[ss->nbSeq].offBase = offBase;
    ss->nbSeq++;
}

/* N/compress/zstd_compress_internal.h:898-960 */
static u32 hash_ptr(const u8* p, u32 hBits, u32 mls) {
    switch (mls) {
    default:
    case 4: return (rd32(p) * 2654435761U) >> (32 - hBits);
    case 5: return (u32)(((rd64(p) << 24) * 889523592379ULL) >> (64 - hBits));
    case 6: return (u32)(((rd64(p) << 16) * 227718039650203ULL) >> (64 - hBits));
    case 7: return (u32)(((rd64(p) << 8) * 58295818150454627ULL) >> (64 - hBits));
    case 8: return (u32)((rd64(p) * 0xCF1BBCDCB7A56463ULL) >> (64 - hBits));
    }
}

/* N/compress/zstd_compress_internal.h:854-876 (result only; byte-wise is equivalent) */
static u32 count_match(const u8* in, const u8* match, const u8* inLimit) {
    const u8* const start = in;
    while (in < inLimit && *in == *match) { in++; match++; }
    return (u32)(in - start);
}

/* Hash tables hold position+1 (0 = empty).  The reference holds index = position+2 with
 * prefixStartIndex = 2 and zero-initialised tables, so "idx >= prefixStartIndex" <=> entry != 0 and
 * "idx > prefixLowestIndex" <=> position >= 1. */

/* ZSTD_compressBlock_fast_noDict_generic, N/compress/zstd_fast.c:192-423.  Returns last literals. */
static u32 block_fast(SeqStore* ss, u32 rep[3], const u8* src, u32 srcSize, const CParams* cp, u32* table) {
    u32 const hlog = cp->hashLog, mls = cp->minMatch;
    u32 const stepSize = 2;                       /* targetLength == 0 at levels >= 1 */
    const u8* const istart = src; const u8* const iend = src + srcSize; const u8* const ilimit = iend - 8;
    const u8* anchor = istart; const u8* ip0 = istart; const u8* ip1; const u8* ip2; const u8* ip3;
    u32 rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0;
    u32 hash0, hash1, matchE, cur0 = 0, offcode, mLength, step;
    const u8* match0; const u8* nextStep;
    u32 const kStepIncr = 1u << 7;                /* kSearchStrength - 1 */

    ip0 += 1;                                     /* ip0 == prefixStart */
    {   u32 const maxRep = (u32)(ip0 - istart);
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; } }
_start:
    step = stepSize; nextStep = ip0 + kStepIncr;
    ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
    if (ip3 >= ilimit) goto _cleanup;
    hash0 = hash_ptr(ip0, hlog, mls); hash1 = hash_ptr(ip1, hlog, mls);
    matchE = table[hash0];
    do {
        u32 const rval = rd32(ip2 - rep1);
        cur0 = (u32)(ip0 - istart); table[hash0] = cur0 + 1;
        if ((rd32(ip2) == rval) & (rep1 > 0)) {
            ip0 = ip2; match0 = ip0 - rep1;
            mLength = (ip0[-1] == match0[-1]); ip0 -= mLength; match0 -= mLength;
            offcode = 1; mLength += 4;
            table[hash1] = (u32)(ip1 - istart) + 1;
            goto _match;
        }
        if (matchE && rd32(istart + matchE - 1) == rd32(ip0)) {
            table[hash1] = (u32)(ip1 - istart) + 1;
            goto _offset;
        }
        matchE = table[hash1];
        hash0 = hash1; hash1 = hash_ptr(ip2, hlog, mls);
        ip0 = ip1; ip1 = ip2; ip2 = ip3;
        cur0 = (u32)(ip0 - istart); table[hash0] = cur0 + 1;
        if (matchE && rd32(istart + matchE - 1) == rd32(ip0)) {
            if (step <= 4) table[hash1] = (u32)(ip1 - istart) + 1;
            goto _offset;
        }
        matchE = table[hash1];
        hash0 = hash1; hash1 = hash_ptr(ip2, hlog, mls);
        ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
        if (ip2 >= nextStep) { step++; nextStep += kStepIncr; }
    } while (ip3 < ilimit);
_cleanup:
    saved2 = ((saved1 != 0) && (rep1 != 0)) ? saved1 : saved2;
    rep[0] = rep1 ? rep1 : saved1; rep[1] = rep2 ? rep2 : saved2;
    return (u32)(iend - anchor);
_offset:
    match0 = istart + matchE - 1;
    rep2 = rep1; rep1 = (u32)(ip0 - match0); offcode = rep1 + 3; mLength = 4;
    while (((ip0 > anchor) & (match0 > istart)) && (ip0[-1] == match0[-1])) { ip0--; match0--; mLength++; }
_match:
    mLength += count_match(ip0 + mLength, match0 + mLength, iend);
    store_seq(ss, anchor, (u32)(ip0 - anchor), offcode, mLength);
    ip0 += mLength; anchor = ip0;
    if (ip0 <= ilimit) {
        table[hash_ptr(istart + cur0 + 2, hlog, mls)] = cur0 + 2 + 1;
        table[hash_ptr(ip0 - 2, hlog, mls)] = (u32)(ip0 - 2 - istart) + 1;
        if (rep2 > 0) {
            while ((ip0 <= ilimit) && (rd32(ip0) == rd32(ip0 - rep2))) {
                u32 const rLength = count_match(ip0 + 4, ip0 + 4 - rep2, iend) + 4;
                { u32 const t = rep2; rep2 = rep1; rep1 = t; }
                table[hash_ptr(ip0, hlog, mls)] = (u32)(ip0 - istart) + 1;
                ip0 += rLength;
                store_seq(ss, anchor, 0, 1, rLength);
                anchor = ip0;
            }
        }
    }
    goto _start;
}

/* ZSTD_compressBlock_doubleFast_noDict_generic, N/compress/zstd_double_fast.c:105-323 */
static u32 block_dfast(SeqStore* ss, u32 rep[3], const u8* src, u32 srcSize, const CParams* cp, u32* hashLong, u32* hashSmall) {
    u32 const hBitsL = cp->hashLog, hBitsS = cp->chainLog, mls = cp->minMatch;
    const u8* const istart = src; const u8* const iend = src + srcSize; const u8* const ilimit = iend - 8;
    const u8* anchor = istart; const u8* ip = istart; const u8* ip1;
    u32 off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0;
    u32 mLength, offset, curr = 0, step, hl0, hl1, el0, el1;
    const u8* nextStep; const u8* matchs0; const u8* matchl0;
    u32 const kStepIncr = 1u << 8;                /* kSearchStrength */

    ip += 1;
    {   u32 const maxRep = (u32)(ip - istart);
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; } }
    for (;;) {
        step = 1; nextStep = ip + kStepIncr; ip1 = ip + step;
        if (ip1 > ilimit) goto _cleanup;
        hl0 = hash_ptr(ip, hBitsL, 8); el0 = hashLong[hl0];
        do {
            u32 const hs0 = hash_ptr(ip, hBitsS, mls);
            u32 const es0 = hashSmall[hs0];
            curr = (u32)(ip - istart);
            hashLong[hl0] = hashSmall[hs0] = curr + 1;
            if ((off1 > 0) & (rd32(ip + 1 - off1) == rd32(ip + 1))) {
                mLength = count_match(ip + 1 + 4, ip + 1 + 4 - off1, iend) + 4;
                ip++;
                store_seq(ss, anchor, (u32)(ip - anchor), 1, mLength);
                goto _match_stored;
            }
            hl1 = hash_ptr(ip1, hBitsL, 8);
            if (el0 && rd64(istart + el0 - 1) == rd64(ip)) {
                matchl0 = istart + el0 - 1;
                mLength = count_match(ip + 8, matchl0 + 8, iend) + 8;
                offset = (u32)(ip - matchl0);
                while (((ip > anchor) & (matchl0 > istart)) && (ip[-1] == matchl0[-1])) { ip--; matchl0--; mLength++; }
                goto _match_found;
            }
            el1 = hashLong[hl1];
            if (es0 && rd32(istart + es0 - 1) == rd32(ip)) { matchs0 = istart + es0 - 1; goto _search_next_long; }
            if (ip1 >= nextStep) { step++; nextStep += kStepIncr; }
            ip = ip1; ip1 += step;
            hl0 = hl1; el0 = el1;
        } while (ip1 <= ilimit);
_cleanup:
        saved2 = ((saved1 != 0) && (off1 != 0)) ? saved1 : saved2;
        rep[0] = off1 ? off1 : saved1; rep[1] = off2 ? off2 : saved2;
        return (u32)(iend - anchor);
_search_next_long:
        mLength = count_match(ip + 4, matchs0 + 4, iend) + 4;
        offset = (u32)(ip - matchs0);
        if ((el1 > 1) && (rd64(istart + el1 - 1) == rd64(ip1))) {       /* idxl1 > prefixLowestIndex */
            const u8* const matchl1 = istart + el1 - 1;
            u32 const l1len = count_match(ip1 + 8, matchl1 + 8, iend) + 8;
            if (l1len > mLength) { ip = ip1; mLength = l1len; offset = (u32)(ip - matchl1); matchs0 = matchl1; }
        }
        while (((ip > anchor) & (matchs0 > istart)) && (ip[-1] == matchs0[-1])) { ip--; matchs0--; mLength++; }
_match_found:
        off2 = off1; off1 = offset;
        if (step < 4) hashLong[hl1] = (u32)(ip1 - istart) + 1;
        store_seq(ss, anchor, (u32)(ip - anchor), offset + 3, mLength);
_match_stored:
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            {   u32 const ins = curr + 2;
                hashLong[hash_ptr(istart + ins, hBitsL, 8)] = ins + 1;
                hashLong[hash_ptr(ip - 2, hBitsL, 8)] = (u32)(ip - 2 - istart) + 1;
                hashSmall[hash_ptr(istart + ins, hBitsS, mls)] = ins + 1;
                hashSmall[hash_ptr(ip - 1, hBitsS, mls)] = (u32)(ip - 1 - istart) + 1;
            }
            while ((ip <= ilimit) && ((off2 > 0) & (rd32(ip) == rd32(ip - off2)))) {
                u32 const rLength = count_match(ip + 4, ip + 4 - off2, iend) + 4;
                u32 const t = off2; off2 = off1; off1 = t;
                hashSmall[hash_ptr(ip, hBitsS, mls)] = (u32)(ip - istart) + 1;
                hashLong[hash_ptr(ip, hBitsL, 8)] = (u32)(ip - istart) + 1;
                store_seq(ss, anchor, 0, 1, rLength);
                ip += rLength; anchor = ip;
            }
        }
    }
}

/* ------------------------------------------------------------------ FSE compression side --- */
/* FSE_optimalTableLog_internal, N/compress/fse_compress.c:346-372 */
static u32 fse_optimal_log(u32 maxTableLog, u32 srcSize, u32 maxSV, u32 minus) {
    u32 const maxBitsSrc = hibit(srcSize - 1) - minus;
    u32 tableLog = maxTableLog;
    u32 const minBitsSrc = hibit(srcSize) + 1, minBitsSym = hibit(maxSV) + 2;
    u32 const minBits = minBitsSrc < minBitsSym ? minBitsSrc : minBitsSym;
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < 5) tableLog = 5;
    if (tableLog > 12) tableLog = 12;
    return tableLog;
}

/* FSE_normalizeM2, N/compress/fse_compress.c:377-462 */
static int fse_normalize_m2(short* norm, u32 tableLog, const u32* count, u32 total, u32 maxSV, short lowProb) {
    short const NOT_YET = -2; u32 s, distributed = 0, toDist;
    u32 const lowThreshold = total >> tableLog; u32 lowOne = (u32)(((u64)total * 3) >> (tableLog + 1));
    for (s = 0; s <= maxSV; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProb; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = NOT_YET;
    }
    toDist = (1u << tableLog) - distributed;
    if (toDist == 0) return 0;
    if ((total / toDist) > lowOne) {
        lowOne = (u32)(((u64)total * 3) / (toDist * 2));
        for (s = 0; s <= maxSV; s++) {
            if ((norm[s] == NOT_YET) && (count[s] <= lowOne)) { norm[s] = 1; distributed++; total -= count[s]; }
        }
        toDist = (1u << tableLog) - distributed;
    }
    if (distributed == maxSV + 1) {
        u32 maxV = 0, maxC = 0;
        for (s = 0; s <= maxSV; s++) if (count[s] > maxC) { maxV = s; maxC = count[s]; }
        norm[maxV] += (short)toDist;
        return 0;
    }
    if (total == 0) {
        for (s = 0; toDist > 0; s = (s + 1) % (maxSV + 1)) if (norm[s] > 0) { toDist--; norm[s]++; }
        return 0;
    }
    {   u64 const vStepLog = 62 - tableLog; u64 const mid = (1ULL << (vStepLog - 1)) - 1;
        u64 const rStep = ((((u64)1 << vStepLog) * toDist) + mid) / total;
        u64 tmpTotal = mid;
        for (s = 0; s <= maxSV; s++) {
            if (norm[s] == NOT_YET) {
                u64 const end = tmpTotal + (count[s] * rStep);
                u32 const sStart = (u32)(tmpTotal >> vStepLog), sEnd = (u32)(end >> vStepLog);
                u32 const weight = sEnd - sStart;
                if (weight < 1) return -1;
                norm[s] = (short)weight; tmpTotal = end;
            }
        }
    }
    return 0;
}

/* FSE_normalizeCount, N/compress/fse_compress.c:465-523.  Returns 0 = rle special case, <0 error, else tableLog */
static int fse_normalize(short* norm, u32 tableLog, const u32* count, u32 total, u32 maxSV, int useLowProb) {
    static const u32 rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    short const lowProb = useLowProb ? -1 : 1;
    u64 const scale = 62 - tableLog; u64 const step = ((u64)1 << 62) / total; u64 const vStep = 1ULL << (scale - 20);
    int still = 1 << tableLog; u32 s, largest = 0; short largestP = 0; u32 const lowThreshold = total >> tableLog;
    for (s = 0; s <= maxSV; s++) {
        if (count[s] == total) return 0;
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProb; still--; }
        else {
            short proba = (short)((count[s] * step) >> scale);
            if (proba < 8) { u64 const restToBeat = vStep * rtb[proba]; proba += (count[s] * step) - ((u64)proba << scale) > restToBeat; }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba; still -= proba;
        }
    }
    if (-still >= (norm[largest] >> 1)) { if (fse_normalize_m2(norm, tableLog, count, total, maxSV, lowProb)) return -1; }
    else norm[largest] += (short)still;
    return (int)tableLog;
}

/* FSE_writeNCount_generic, N/compress/fse_compress.c:237-328 (safe-write variant; returns size) */
static size_t fse_write_ncount(u8* out0, const short* norm, u32 maxSV, u32 tableLog) {
    u8* out = out0; int nbBits; int const tableSize = 1 << tableLog; int remaining, threshold;
    u32 bitStream = 0; int bitCount = 0; u32 symbol = 0; u32 const alphabetSize = maxSV + 1; int previousIs0 = 0;
    bitStream += (tableLog - 5) << bitCount; bitCount += 4;
    remaining = tableSize + 1; threshold = tableSize; nbBits = (int)tableLog + 1;
    while ((symbol < alphabetSize) && (remaining > 1)) {
        if (previousIs0) {
            u32 start = symbol;
            while ((symbol < alphabetSize) && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) { start += 24; bitStream += 0xFFFFU << bitCount; out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; }
            while (symbol >= start + 3) { start += 3; bitStream += 3U << bitCount; bitCount += 2; }
            bitStream += (symbol - start) << bitCount; bitCount += 2;
            if (bitCount > 16) { out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
        }
        {   int count = norm[symbol++]; int const max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (u32)count << bitCount; bitCount += nbBits; bitCount -= (count < max);
            previousIs0 = (count == 1);
            if (remaining < 1) return 0;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) { out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
    }
    if (remaining != 1) return 0;
    out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += (bitCount + 7) / 8;
    return (size_t)(out - out0);
}

/* Compression table: N/common/fse.h:423-461 + FSE_buildCTable_wksp, N/compress/fse_compress.c:68-224 */
typedef struct { u32 tableLog; u16 state[512]; i32 deltaFind[256]; u32 deltaNbBits[256]; } FseCT;

static void fse_build_ctable(FseCT* ct, const short* norm, u32 maxSV, u32 tableLog) {
    u32 const size = 1u << tableLog, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    u16 cumul[258]; u8 tableSymbol[512]; u32 high = size - 1, u, s, pos = 0;
    ct->tableLog = tableLog;
    cumul[0] = 0;
    for (u = 1; u <= maxSV + 1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tableSymbol[high--] = (u8)(u - 1); }
        else cumul[u] = cumul[u - 1] + (u16)norm[u - 1];
    }
    cumul[maxSV + 1] = (u16)(size + 1);
    for (s = 0; s <= maxSV; s++) {
        int i; for (i = 0; i < norm[s]; i++) { tableSymbol[pos] = (u8)s; do { pos = (pos + step) & mask; } while (pos > high); }
    }
    for (u = 0; u < size; u++) { u8 const sy = tableSymbol[u]; ct->state[cumul[sy]++] = (u16)(size + u); }
    {   u32 total = 0;
        for (s = 0; s <= maxSV; s++) {
            switch (norm[s]) {
            case 0: ct->deltaNbBits[s] = ((tableLog + 1) << 16) - (1u << tableLog); ct->deltaFind[s] = 0; break;
            case -1: case 1: ct->deltaNbBits[s] = (tableLog << 16) - (1u << tableLog); ct->deltaFind[s] = (i32)(total - 1); total++; break;
            default: { u32 const maxBitsOut = tableLog - hibit((u32)norm[s] - 1); u32 const minStatePlus = (u32)norm[s] << maxBitsOut;
                       ct->deltaNbBits[s] = (maxBitsOut << 16) - minStatePlus; ct->deltaFind[s] = (i32)(total - (u32)norm[s]); total += (u32)norm[s]; }
            }
        }
    }
}
static void fse_build_ctable_rle(FseCT* ct, u32 symbol) {   /* N/compress/fse_compress.c:526-546 */
    ct->tableLog = 0; ct->state[0] = 0; ct->state[1] = 0; ct->deltaNbBits[symbol] = 0; ct->deltaFind[symbol] = 0;
}

/* forward bit writer = BIT_CStream_t (N/common/bitstream.h:180-250) with an unbounded accumulator */
typedef struct { u8* p; u64 acc; u32 n; } BitW;
static void bw_add(BitW* b, u64 v, u32 nb) { if (nb) { b->acc |= (v & (((u64)1 << nb) - 1)) << b->n; b->n += nb; } while (b->n >= 8) { *b->p++ = (u8)b->acc; b->acc >>= 8; b->n -= 8; } }
static size_t bw_close(BitW* b, const u8* start) { bw_add(b, 1, 1); if (b->n) { *b->p++ = (u8)b->acc; b->n = 0; } return (size_t)(b->p - start); }

typedef struct { u32 value; const FseCT* ct; } FseCS;
static void fse_init2(FseCS* s, const FseCT* ct, u32 sym) {            /* FSE_initCState2 */
    u32 const nbBitsOut = (ct->deltaNbBits[sym] + (1u << 15)) >> 16;
    s->ct = ct; s->value = (nbBitsOut << 16) - ct->deltaNbBits[sym];
    s->value = ct->state[(i32)(s->value >> nbBitsOut) + ct->deltaFind[sym]];
}
static void fse_encode(BitW* b, FseCS* s, u32 sym) {                   /* FSE_encodeSymbol */
    u32 const nbBitsOut = (s->value + s->ct->deltaNbBits[sym]) >> 16;
    bw_add(b, s->value, nbBitsOut);
    s->value = s->ct->state[(i32)(s->value >> nbBitsOut) + s->ct->deltaFind[sym]];
}
static void fse_flush(BitW* b, const FseCS* s) { bw_add(b, s->value, s->ct->tableLog); }

/* ------------------------------------------------------------------ Huffman ---------------- */
typedef struct { u32 count; u16 parent; u8 byte; u8 nbBits; } Node;

static u32 huf_bucket(u32 count) { return count < 166 ? count : hibit(count) + 158; }   /* HUF_getIndex, huf_compress.c:513-517 */

static void huf_insertion(Node* a, int low, int high) {
    int i, size = high - low + 1; a += low;
    for (i = 1; i < size; i++) { Node const key = a[i]; int j = i - 1; while (j >= 0 && a[j].count < key.count) { a[j + 1] = a[j]; j--; } a[j + 1] = key; }
}
static int huf_partition(Node* a, int low, int high) {
    u32 const pivot = a[high].count; int i = low - 1, j;
    for (j = low; j < high; j++) if (a[j].count > pivot) { Node t; i++; t = a[i]; a[i] = a[j]; a[j] = t; }
    { Node t = a[i + 1]; a[i + 1] = a[high]; a[high] = t; }
    return i + 1;
}
static void huf_quicksort(Node* a, int low, int high) {               /* HUF_simpleQuickSort, huf_compress.c:574-591 */
    if (high - low < 8) { huf_insertion(a, low, high); return; }
    while (low < high) {
        int const idx = huf_partition(a, low, high);
        if (idx - low < high - idx) { huf_quicksort(a, low, idx - 1); low = idx + 1; }
        else { huf_quicksort(a, idx + 1, high); high = idx - 1; }
    }
}
static void huf_sort(Node* node, const u32* count, u32 maxSV) {      /* HUF_sort, huf_compress.c:603-647 */
    struct { u16 base, curr; } rp[192]; u32 n;
    memset(rp, 0, sizeof(rp));
    for (n = 0; n <= maxSV; n++) rp[huf_bucket(count[n])].base++;
    for (n = 191; n > 0; n--) { rp[n - 1].base += rp[n].base; rp[n - 1].curr = rp[n - 1].base; }
    for (n = 0; n <= maxSV; n++) { u32 const r = huf_bucket(count[n]) + 1; u32 const pos = rp[r].curr++; node[pos].count = count[n]; node[pos].byte = (u8)n; }
    /* RANK_POSITION_DISTINCT_COUNT_CUTOFF = 158 + highbit32(158) = 165 (not the 166 of the reference's comment): slot 165 holds count == 164 */
    for (n = 165; n < 191; n++) { int const sz = rp[n].curr - rp[n].base; if (sz > 1) huf_quicksort(node + rp[n].base, 0, sz - 1); }
}

/* HUF_buildTree + HUF_setMaxHeight + HUF_buildCTableFromTree, huf_compress.c:376-754.
 * Outputs nbBits[sym], code value[sym]; returns table log. */
static u32 huf_build(u8* nbBitsOut, u16* valOut, const u32* count, u32 maxSV, u32 maxNbBits) {
    Node tbl[2 * 256 + 2]; Node* const node0 = tbl; Node* const node = tbl + 1;
    int nonNull, lowS, lowN, nodeNb = 256, n, nodeRoot;
    memset(tbl, 0, sizeof(tbl));
    huf_sort(node, count, maxSV);
    nonNull = (int)maxSV; while (node[nonNull].count == 0) nonNull--;
    lowS = nonNull; nodeRoot = nodeNb + lowS - 1; lowN = nodeNb;
    node[nodeNb].count = node[lowS].count + node[lowS - 1].count;
    node[lowS].parent = node[lowS - 1].parent = (u16)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= nodeRoot; n++) node[n].count = 1u << 30;
    node0[0].count = 1u << 31;
    while (nodeNb <= nodeRoot) {
        int const n1 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        int const n2 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        node[nodeNb].count = node[n1].count + node[n2].count;
        node[n1].parent = node[n2].parent = (u16)nodeNb; nodeNb++;
    }
    node[nodeRoot].nbBits = 0;
    for (n = nodeRoot - 1; n >= 256; n--) node[n].nbBits = node[node[n].parent].nbBits + 1;
    for (n = 0; n <= nonNull; n++) node[n].nbBits = node[node[n].parent].nbBits + 1;
    /* HUF_setMaxHeight */
    {   u32 const largestBits = node[nonNull].nbBits;
        if (largestBits > maxNbBits) {
            int totalCost = 0; u32 const baseCost = 1u << (largestBits - maxNbBits); u32 const noSymbol = 0xF0F0F0F0;
            u32 rankLast[14]; int k = nonNull;
            while (node[k].nbBits > maxNbBits) { totalCost += (int)(baseCost - (1u << (largestBits - node[k].nbBits))); node[k].nbBits = (u8)maxNbBits; k--; }
            while (node[k].nbBits == maxNbBits) --k;
            totalCost >>= (largestBits - maxNbBits);
            memset(rankLast, 0xF0, sizeof(rankLast));
            {   u32 cur = maxNbBits; int pos;
                for (pos = k; pos >= 0; pos--) { if (node[pos].nbBits >= cur) continue; cur = node[pos].nbBits; rankLast[maxNbBits - cur] = (u32)pos; } }
            while (totalCost > 0) {
                u32 nDec = hibit((u32)totalCost) + 1;
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
        } else maxNbBits = largestBits;
    }
    /* HUF_buildCTableFromTree */
    {   u16 nbPerRank[13] = {0}, valPerRank[13] = {0}; u16 min = 0;
        for (n = 0; n <= nonNull; n++) nbPerRank[node[n].nbBits]++;
        for (n = (int)maxNbBits; n > 0; n--) { valPerRank[n] = min; min += nbPerRank[n]; min >>= 1; }
        for (n = 0; n <= (int)maxSV; n++) nbBitsOut[node[n].byte] = node[n].nbBits;
        for (n = 0; n <= (int)maxSV; n++) valOut[n] = nbBitsOut[n] ? valPerRank[nbBitsOut[n]]++ : 0;
    }
    return maxNbBits;
}

/* HUF_compressWeights, huf_compress.c:132-176: returns 0 not compressible, 1 rle, else size */
static size_t huf_compress_weights(u8* dst, const u8* w, u32 wtSize) {
    u32 count[13] = {0}; short norm[13]; u32 maxSV = 0, maxCount = 0, s, tableLog; FseCT ct; u8* op = dst;
    if (wtSize <= 1) return 0;
    for (s = 0; s < wtSize; s++) count[w[s]]++;
    for (s = 0; s <= 12; s++) { if (count[s]) maxSV = s; if (count[s] > maxCount) maxCount = count[s]; }
    if (maxCount == wtSize) return 1;
    if (maxCount == 1) return 0;
    tableLog = fse_optimal_log(6, wtSize, maxSV, 2);
    if (fse_normalize(norm, tableLog, count, wtSize, maxSV, 0) <= 0) return 0;
    {   size_t const h = fse_write_ncount(op, norm, maxSV, tableLog); if (!h) return 0; op += h; }
    fse_build_ctable(&ct, norm, maxSV, tableLog);
    /* FSE_compress_usingCTable_generic, fse_compress.c:549-606 */
    {   const u8* ip = w + wtSize; BitW b; FseCS s1, s2; u32 n = wtSize; u8* const bstart = op;
        if (n <= 2) return 0;
        b.p = op; b.acc = 0; b.n = 0;
        if (n & 1) { fse_init2(&s1, &ct, *--ip); fse_init2(&s2, &ct, *--ip); fse_encode(&b, &s1, *--ip); }
        else { fse_init2(&s2, &ct, *--ip); fse_init2(&s1, &ct, *--ip); }
        n -= 2;
        if (n & 2) { fse_encode(&b, &s2, *--ip); fse_encode(&b, &s1, *--ip); }
        while (ip > w) { fse_encode(&b, &s2, *--ip); fse_encode(&b, &s1, *--ip); fse_encode(&b, &s2, *--ip); fse_encode(&b, &s1, *--ip); }
        fse_flush(&b, &s2); fse_flush(&b, &s1);
        op = bstart + bw_close(&b, bstart);
    }
    return (size_t)(op - dst);
}

/* HUF_writeCTable_wksp, huf_compress.c:248-290.  returns 0 on failure */
static size_t huf_write_ctable(u8* op, const u8* nbBits, u32 maxSV, u32 huffLog) {
    u8 weight[256]; u32 n;
    for (n = 0; n < maxSV; n++) weight[n] = nbBits[n] ? (u8)(huffLog + 1 - nbBits[n]) : 0;
    {   size_t const h = huf_compress_weights(op + 1, weight, maxSV);
        if ((h > 1) & (h < maxSV / 2)) { op[0] = (u8)h; return h + 1; } }
    if (maxSV > 128) return 0;
    op[0] = (u8)(128 + (maxSV - 1));
    weight[maxSV] = 0;
    for (n = 0; n < maxSV; n += 2) op[(n / 2) + 1] = (u8)((weight[n] << 4) + weight[n + 1]);
    return ((maxSV + 1) / 2) + 1;
}

/* one stream: symbols last -> first (huf_compress.c:991-1118), end mark, size */
static size_t huf_encode_1x(u8* dst, const u8* src, u32 n, const u8* nbBits, const u16* val) {
    BitW b; u32 i; b.p = dst; b.acc = 0; b.n = 0;
    for (i = n; i > 0; i--) bw_add(&b, val[src[i - 1]], nbBits[src[i - 1]]);
    return bw_close(&b, dst);
}

/* ZSTD_compressLiterals for a first block (no previous table), zstd_compress_literals.c:129-235 */
static size_t raw_literals(u8* dst, const u8* lit, u32 n) {
    u32 const fl = 1 + (n > 31) + (n > 4095);
    if (fl == 1) dst[0] = (u8)(0 + (n << 3)); else if (fl == 2) wr16(dst, 0 + (1 << 2) + (n << 4)); else wr32(dst, 0 + (3 << 2) + (n << 4));
    memcpy(dst + fl, lit, n); return n + fl;
}
static size_t rle_literals(u8* dst, const u8* lit, u32 n) {
    u32 const fl = 1 + (n > 31) + (n > 4095);
    if (fl == 1) dst[0] = (u8)(1 + (n << 3)); else if (fl == 2) wr16(dst, 1 + (1 << 2) + (n << 4)); else wr32(dst, 1 + (3 << 2) + (n << 4));
    dst[fl] = lit[0]; return fl + 1;
}

static size_t compress_literals(u8* dst, const u8* lit, u32 n, u32 strategy, int suspectUncompressible) {
    u32 const lhSize = 3 + (n >= 1024) + (n >= 16384); int const single = n < 256;
    u32 count[256]; u8 nbBits[256]; u16 val[256]; u32 maxSV = 255, largest = 0, i, huffLog; size_t cLit; u8* const ostart = dst + lhSize; u8* op = ostart;
    u32 const minLit = 8u << (9 - strategy < 3 ? 9 - strategy : 3);      /* ZSTD_minLiteralsToCompress, repeat none */
    if (n < minLit) return raw_literals(dst, lit, n);
    /* HUF_compress_internal, huf_compress.c:1333-1434 */
    if (suspectUncompressible && n >= 4096 * 10) {
        u32 c2[256]; u32 lb = 0, le = 0;
        memset(c2, 0, sizeof(c2)); for (i = 0; i < 4096; i++) c2[lit[i]]++; for (i = 0; i < 256; i++) if (c2[i] > lb) lb = c2[i];
        memset(c2, 0, sizeof(c2)); for (i = 0; i < 4096; i++) c2[lit[n - 4096 + i]]++; for (i = 0; i < 256; i++) if (c2[i] > le) le = c2[i];
        if (lb + le <= ((2 * 4096) >> 7) + 4) return raw_literals(dst, lit, n);
    }
    memset(count, 0, sizeof(count)); for (i = 0; i < n; i++) count[lit[i]]++;
    while (!count[maxSV]) maxSV--;
    for (i = 0; i <= maxSV; i++) if (count[i] > largest) largest = count[i];
    if (largest == n) return rle_literals(dst, lit, n);                 /* cLitSize == 1 with n >= 8 */
    if (largest <= (n >> 7) + 4) return raw_literals(dst, lit, n);
    huffLog = fse_optimal_log(HUF_LOG_LIT, n, maxSV, 1);
    huffLog = huf_build(nbBits, val, count, maxSV, huffLog);
    {   size_t const h = huf_write_ctable(op, nbBits, maxSV, huffLog);
        if (!h || h + 12 >= n) return raw_literals(dst, lit, n);
        op += h; }
    if (single) op += huf_encode_1x(op, lit, n, nbBits, val);
    else {
        u32 const seg = (n + 3) / 4; u8* const jt = op; size_t c;
        if (n < 12) return raw_literals(dst, lit, n);
        op += 6;
        c = huf_encode_1x(op, lit, seg, nbBits, val); if (c > 65535) return raw_literals(dst, lit, n); wr16(jt, (u32)c); op += c;
        c = huf_encode_1x(op, lit + seg, seg, nbBits, val); if (c > 65535) return raw_literals(dst, lit, n); wr16(jt + 2, (u32)c); op += c;
        c = huf_encode_1x(op, lit + 2 * seg, seg, nbBits, val); if (c > 65535) return raw_literals(dst, lit, n); wr16(jt + 4, (u32)c); op += c;
        c = huf_encode_1x(op, lit + 3 * seg, n - 3 * seg, nbBits, val); if (c > 65535) return raw_literals(dst, lit, n); op += c;
    }
    cLit = (size_t)(op - ostart);
    if (cLit >= n - 1) return raw_literals(dst, lit, n);                /* HUF_compressCTable_internal */
    if (cLit >= n - ((n >> 6) + 2)) return raw_literals(dst, lit, n);   /* ZSTD_minGain */
    if (lhSize == 3) wr24(dst, 2 + ((u32)(!single) << 2) + (n << 4) + ((u32)cLit << 14));
    else if (lhSize == 4) wr32(dst, 2 + (2 << 2) + (n << 4) + ((u32)cLit << 18));
    else { wr32(dst, 2 + (3 << 2) + (n << 4) + ((u32)cLit << 22)); dst[4] = (u8)(cLit >> 10); }
    return lhSize + cLit;
}

/* ------------------------------------------------------------------ sequences section ------ */
static const u8 LL_bits[MAXLL + 1] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
static const u8 ML_bits[MAXML + 1] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
static const short LL_defNorm[MAXLL + 1] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
static const short ML_defNorm[MAXML + 1] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
static const short OF_defNorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

static u32 ll_code(u32 v) {   /* ZSTD_LLcode */
    static const u8 t[64] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,22,22,22,22,22,22,22,22,
                              23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24 };
    return v > 63 ? hibit(v) + 19 : t[v];
}
static u32 ml_code(u32 v) {   /* ZSTD_MLcode (v = matchLength - 3) */
    static const u8 t[128] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,
        32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,
        40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,
        42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42 };
    return v > 127 ? hibit(v) + 36 : t[v];
}

/* ZSTD_selectEncodingType for strategy < lazy and no previous tables, zstd_compress_sequences.c:157-203 */
static u32 select_type(u32 mostFrequent, u32 nbSeq, u32 defaultNormLog, int defaultAllowed, u32 strategy) {
    if (mostFrequent == nbSeq) return (defaultAllowed && nbSeq <= 2) ? 0 : 1;
    if (defaultAllowed) {
        u32 const mult = 10 - strategy; u32 const dynMin = ((1u << defaultNormLog) * mult) >> 3;
        if ((nbSeq < dynMin) || (mostFrequent < (nbSeq >> (defaultNormLog - 1)))) return 0;
    }
    return 2;
}

/* one of LL / OF / ML: histogram, mode choice, table description, CTable (zstd_compress.c:2763-2880,
 * zstd_compress_sequences.c:243-285).  Returns bytes written; *type gets the mode. */
static size_t build_seq_table(u8* op, FseCT* ct, u32* type, const u8* codes, u32 nbSeq, u32 maxSym, u32 fseLog,
                              const short* defNorm, u32 defLog, u32 defMax, int isOffset, u32 strategy, size_t* lastCountSize) {
    u32 count[64]; u32 max = 0, most = 0, i; int defaultAllowed = 1;
    memset(count, 0, sizeof(count));
    for (i = 0; i < nbSeq; i++) count[codes[i]]++;
    for (i = 0; i <= maxSym; i++) { if (count[i]) max = i; if (count[i] > most) most = count[i]; }
    if (isOffset) defaultAllowed = (max <= 28);                           /* DefaultMaxOff */
    *type = select_type(most, nbSeq, defLog, defaultAllowed, strategy);
    switch (*type) {
    case 1: fse_build_ctable_rle(ct, max); op[0] = codes[0]; return 1;
    case 0: fse_build_ctable(ct, defNorm, defMax, defLog); return 0;
    default: {
        short norm[64]; u32 nbSeq1 = nbSeq; u32 const tableLog = fse_optimal_log(fseLog, nbSeq, max, 2); size_t h;
        if (count[codes[nbSeq - 1]] > 1) { count[codes[nbSeq - 1]]--; nbSeq1--; }
        fse_normalize(norm, tableLog, count, nbSeq1, max, nbSeq1 >= 2048);
        h = fse_write_ncount(op, norm, max, tableLog);
        fse_build_ctable(ct, norm, max, tableLog);
        *lastCountSize = h;
        return h; }
    }
}

/* ZSTD_entropyCompressSeqStore_internal, zstd_compress.c:2888-3003.  Returns block body size, 0 = "emit raw" */
static size_t entropy_compress(u8* dst, size_t dstCap, const SeqStore* ss, u32 strategy, u32 blockSize) {
    u8* op = dst; u32 const nbSeq = ss->nbSeq; (void)dstCap;
    {   int const suspect = (nbSeq == 0) || (ss->litSize / nbSeq >= 20);
        op += compress_literals(op, ss->lit, ss->litSize, strategy, suspect); }
    if (nbSeq < 128) *op++ = (u8)nbSeq;
    else if (nbSeq < 0x7F00) { op[0] = (u8)((nbSeq >> 8) + 0x80); op[1] = (u8)nbSeq; op += 2; }
    else { op[0] = 0xFF; wr16(op + 1, nbSeq - 0x7F00); op += 3; }
    if (nbSeq) {
        u8* const seqHead = op++; u8* llc = (u8*)malloc(nbSeq * 3); u8* ofc = llc + nbSeq; u8* mlc = ofc + nbSeq;
        FseCT* cts = (FseCT*)malloc(3 * sizeof(FseCT)); FseCT* ctLL = cts; FseCT* ctOF = cts + 1; FseCT* ctML = cts + 2;
        u32 tLL, tOF, tML, i; size_t lastCount = 0, bitSize;
        for (i = 0; i < nbSeq; i++) { llc[i] = (u8)ll_code(ss->seq[i].litLength); ofc[i] = (u8)hibit(ss->seq[i].offBase); mlc[i] = (u8)ml_code(ss->seq[i].matchLength - 3); }
        {   size_t lc = 0, h;
            h = build_seq_table(op, ctLL, &tLL, llc, nbSeq, MAXLL, 9, LL_defNorm, 6, MAXLL, 0, strategy, &lc); if (tLL == 2) lastCount = lc; op += h;
            h = build_seq_table(op, ctOF, &tOF, ofc, nbSeq, MAXOFF, 8, OF_defNorm, 5, 28, 1, strategy, &lc); if (tOF == 2) lastCount = lc; op += h;
            h = build_seq_table(op, ctML, &tML, mlc, nbSeq, MAXML, 9, ML_defNorm, 6, MAXML, 0, strategy, &lc); if (tML == 2) lastCount = lc; op += h;
        }
        *seqHead = (u8)((tLL << 6) + (tOF << 4) + (tML << 2));
        /* ZSTD_encodeSequences_body, zstd_compress_sequences.c:291-382 */
        {   BitW b; FseCS sML, sOF, sLL; u32 n = nbSeq - 1; u8* const bstart = op;
            b.p = op; b.acc = 0; b.n = 0;
            fse_init2(&sML, ctML, mlc[n]); fse_init2(&sOF, ctOF, ofc[n]); fse_init2(&sLL, ctLL, llc[n]);
            bw_add(&b, ss->seq[n].litLength, LL_bits[llc[n]]);
            bw_add(&b, ss->seq[n].matchLength - 3, ML_bits[mlc[n]]);
            bw_add(&b, ss->seq[n].offBase, ofc[n]);
            while (n-- > 0) {
                fse_encode(&b, &sOF, ofc[n]); fse_encode(&b, &sML, mlc[n]); fse_encode(&b, &sLL, llc[n]);
                bw_add(&b, ss->seq[n].litLength, LL_bits[llc[n]]);
                bw_add(&b, ss->seq[n].matchLength - 3, ML_bits[mlc[n]]);
                bw_add(&b, ss->seq[n].offBase, ofc[n]);
            }
            fse_flush(&b, &sML); fse_flush(&b, &sOF); fse_flush(&b, &sLL);
            bitSize = bw_close(&b, bstart); op += bitSize;
        }
        free(llc); free(cts);
        if (lastCount && (lastCount + bitSize) < 4) return 0;
    }
    {   size_t const cSize = (size_t)(op - dst); size_t const maxC = blockSize - ((blockSize >> 6) + 2);
        if (cSize >= maxC) return 0;
        return cSize; }
}

/* ------------------------------------------------------------------ frame ------------------ */
size_t zso_compress_ex(void* dstv, size_t dstCap, const void* srcv, size_t srcSize, int level, int checksum, int hashLogOv, int chainLogOv) {
    u8* const dst = (u8*)dstv; const u8* const src = (const u8*)srcv; CParams cp; size_t pos = 0; u8* tmp;
    if (get_cparams(&cp, level, srcSize, hashLogOv, chainLogOv)) return ERR(parameter_unsupported);
    if (dstCap < 18) return ERR(dstSize_tooSmall);                       /* ZSTD_FRAMEHEADERSIZE_MAX */
    /* ZSTD_writeFrameHeader: contentSizeFlag=1, single segment when windowSize >= srcSize */
    {   u32 const windowSize = 1u << cp.windowLog; u32 const single = windowSize >= srcSize;
        u32 const fcsCode = (srcSize >= 256) + (srcSize >= 65536 + 256);
        wr32(dst, 0xFD2FB528u); pos = 4;
        dst[pos++] = (u8)((checksum ? 4 : 0) + (single << 5) + (fcsCode << 6));
        if (!single) dst[pos++] = (u8)((cp.windowLog - 10) << 3);
        if (fcsCode == 0) { if (single) dst[pos++] = (u8)srcSize; } else if (fcsCode == 1) { wr16(dst + pos, (u32)srcSize - 256); pos += 2; } else { wr32(dst + pos, (u32)srcSize); pos += 4; }
    }
    if (srcSize == 0) {                                                  /* ZSTD_writeEpilogue: empty raw last block */
        if (dstCap < pos + 3 + (checksum ? 4 : 0)) return ERR(dstSize_tooSmall);
        wr24(dst + pos, 1); pos += 3;
    } else {
        size_t cSize = 0; size_t const bound = 2 * srcSize + 1024;   /* Huffman output is checked against srcSize only after encoding */
        tmp = (u8*)malloc(bound);
        if (srcSize >= 7) {                                              /* MIN_CBLOCK_SIZE + blockHeader + 1 + 1 */
            SeqStore ss; u32 rep[3] = { 1, 4, 8 }; u32 lastLL;
            u32* tables = (u32*)calloc(((size_t)1 << cp.hashLog) + ((size_t)1 << cp.chainLog), sizeof(u32));
            ss.seq = (Seq*)malloc(sizeof(Seq) * (srcSize / 3 + 8)); ss.nbSeq = 0; ss.lit = (u8*)malloc(srcSize + 32); ss.litSize = 0;
            lastLL = (cp.strategy == 1) ? block_fast(&ss, rep, src, (u32)srcSize, &cp, tables)
                                        : block_dfast(&ss, rep, src, (u32)srcSize, &cp, tables, tables + ((size_t)1 << cp.hashLog));
            memcpy(ss.lit + ss.litSize, src + srcSize - lastLL, lastLL); ss.litSize += lastLL;
            cSize = entropy_compress(tmp, bound, &ss, cp.strategy, (u32)srcSize);
            free(tables); free(ss.seq); free(ss.lit);
        }
        if (cSize == 0) {                                                /* raw block */
            if (dstCap < pos + 3 + srcSize) { free(tmp); return ERR(dstSize_tooSmall); }
            wr24(dst + pos, 1 + (0 << 1) + ((u32)srcSize << 3)); pos += 3; memcpy(dst + pos, src, srcSize); pos += srcSize;
        } else {
            if (dstCap < pos + 3 + cSize) { free(tmp); return ERR(dstSize_tooSmall); }
            wr24(dst + pos, 1 + (2 << 1) + ((u32)cSize << 3)); pos += 3; memcpy(dst + pos, tmp, cSize); pos += cSize;
        }
        free(tmp);
    }
    if (checksum) { if (dstCap < pos + 4) return ERR(dstSize_tooSmall); wr32(dst + pos, (u32)zso_xxh64(src, srcSize, 0)); pos += 4; }
    return pos;
}

size_t zso_compress(void* dst, size_t dstCap, const void* src, size_t srcSize, int level, int checksum) {
    return zso_compress_ex(dst, dstCap, src, srcSize, level, checksum, 0, 0);
}
