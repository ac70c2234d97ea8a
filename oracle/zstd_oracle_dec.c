/* oracle/zstd_oracle_dec.c — plain-C restatement of the reference's zstd frame decoder.
 *
 * TEST INFRASTRUCTURE ONLY (see zstd_oracle.h).  Parity: PINNED against the reference's golden
 * .zst fixtures and against oracle/_ref (tests/test_oracle.py).
 *
 * Reference files restated here ("N/" = /root/reference/src/main/native/):
 *   frame/blocks   N/decompress/zstd_decompress.c:447-557 (frame header), :953-1066 (frame loop),
 *                  :1070-1168 (multi-frame + skippable)
 *   block header   N/decompress/zstd_decompress_block.c:63-77
 *   literals       N/decompress/zstd_decompress_block.c:134-340
 *   huffman        N/common/entropy_common.c:243-305 (weights), N/decompress/huf_decompress.c:385-518
 *   FSE tables     N/common/entropy_common.c:42-188 (NCount), N/decompress/zstd_decompress_block.c:485-603
 *   sequences      N/decompress/zstd_decompress_block.c:695-782 (headers), :1229-1347 (decode),
 *                  :1001-1096 (execute), :1615-1690 (loop)
 *
 * The restatement is position-based (an explicit "bits left" counter over a backward bitstream)
 * instead of the reference's 64-bit container/reload machinery; results are identical.
 */
#include "zstd_oracle.h"
#include <string.h>
#include <stdlib.h>

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64; typedef int64_t i64;

#define ERR(name) ZSO_ERR(ZSO_error_##name)
#define BLOCK_MAX (1u << 17)
#define MAXLL 35
#define MAXML 52
#define MAXOFF 31
#define LLFSELOG 9
#define MLFSELOG 9
#define OFFFSELOG 8
#define HUFLOG_MAX 12   /* HUF_TABLELOG_MAX, N/common/huf.h */

static u32 rd16(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8); }
static u32 rd24(const u8* p) { return rd16(p) | ((u32)p[2] << 16); }
static u32 rd32(const u8* p) { return rd16(p) | (rd16(p + 2) << 16); }
static u64 rd64(const u8* p) { return (u64)rd32(p) | ((u64)rd32(p + 4) << 32); }
static u32 hibit(u32 v) { u32 r = 0; while (v >>= 1) r++; return r; }   /* v != 0 */

/* ---------------------------------------------------------------- backward bit reader ----- */
typedef struct { const u8* p; size_t n; i64 left; } BitR;   /* left = unread bits; <0 = overflow */

/* N/common/bitstream.h:254-300 BIT_initDStream: last byte carries a 1-bit end marker */
static int bitr_init(BitR* b, const u8* p, size_t n) {
    if (n == 0) return -1;
    if (p[n - 1] == 0) return -1;
    b->p = p; b->n = n; b->left = (i64)(n - 1) * 8 + hibit(p[n - 1]);
    return 0;
}
/* nb <= 32 bits whose lowest bit sits at absolute bit index `pos` (bits below 0 read as 0) */
static u32 bits_at(const BitR* b, i64 pos, u32 nb) {
    u64 acc = 0; i64 i;
    if (nb == 0) return 0;
    if (pos < 0) {
        u32 pad = (u32)(-pos);
        if (pad >= nb) return 0;
        return bits_at(b, 0, nb - pad) << pad;
    }
    {   size_t byte = (size_t)(pos >> 3); u32 sh = (u32)(pos & 7);
        for (i = 0; i < 6 && byte + (size_t)i < b->n; i++) acc |= (u64)b->p[byte + (size_t)i] << (8 * i);
        return (u32)((acc >> sh) & (((u64)1 << nb) - 1));
    }
}
static u32 bitr_read(BitR* b, u32 nb) { b->left -= nb; return bits_at(b, b->left, nb); }
static u32 bitr_peek(const BitR* b, u32 nb) { return bits_at(b, b->left - (i64)nb, nb); }

/* The sequence decoder does not stop when its bit stream runs dry: the reference reads on from the exhausted 64-bit container
 * (by then the stream's first 8 bytes, or all of a shorter stream, N/common/bitstream.h:254-300, :370-420) with shifts that wrap
 * at 64, and some later sequence trips a check (or the end-of-stream test does).  Which check — hence the error CODE — depends
 * on those bits, so they are restated: BIT_readBitsFast = (C << (consumed & 63)) >> (64 - nb)  (:347-356),
 * BIT_readBits = (C >> ((64 - consumed - nb) & 63)) & mask(nb)  (:303-343), consumed = 64 - left. */
static u64 bitr_container(const BitR* b) { u64 c = 0; size_t k; for (k = 0; k < 8 && k < b->n; k++) c |= (u64)b->p[k] << (8 * k); return c; }
static u32 seq_read_fast(BitR* b, u32 nb) {       /* nb >= 1 */
    u32 v;
    if ((i64)nb <= b->left) v = bits_at(b, b->left - (i64)nb, nb);
    else { u32 const c = (u32)(64 - b->left); v = (u32)((bitr_container(b) << (c & 63)) >> (64 - nb)); }
    b->left -= nb; return v;
}
static u32 seq_read(BitR* b, u32 nb) {
    u32 v;
    if ((i64)nb <= b->left) v = bits_at(b, b->left - (i64)nb, nb);
    else { u32 const c = (u32)(64 - b->left); v = (u32)(bitr_container(b) >> ((64u - c - nb) & 63u)) & (u32)(((u64)1 << nb) - 1); }
    b->left -= nb; return v;
}

/* ---------------------------------------------------------------- FSE NCount ------------- */
/* N/common/entropy_common.c:42-188.  Returns header bytes consumed or error. */
/* The cursor is the reference's own (byte position + bit count + a 32-bit word re-read after every field) because its behaviour
 * at the end of the buffer is part of the contract: the position is pinned 4 bytes before the end and the bit count taken
 * modulo 32 (:146-153, :170-177), so a description running past its buffer wraps around on the last four bytes instead of
 * failing; only the bit count after the last field is tested (:184).  Buffers under 8 bytes: zero-padded copy (:62-72). */
static size_t read_ncount_body(short* norm, u32* maxSV, u32* tableLog, const u8* src, size_t hbSize) {   /* hbSize >= 8 */
    u32 const maxSV1 = *maxSV + 1;
    int const iend = (int)hbSize;
    int ip = 0, bitCount = 4, nbBits, remaining, threshold;
    u32 charnum = 0; int previous0 = 0;
    u32 bitStream = rd32(src);
    nbBits = (int)(bitStream & 0xF) + 5;
    if (nbBits > 15) return ERR(tableLog_tooLarge);          /* FSE_TABLELOG_ABSOLUTE_MAX */
    memset(norm, 0, maxSV1 * sizeof(short));
    bitStream >>= 4; *tableLog = (u32)nbBits;
    remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;
    for (;;) {
        if (previous0) {
            u32 repeats = (u32)__builtin_ctz(~bitStream | 0x80000000u) >> 1;
            while (repeats >= 12) {
                charnum += 3 * 12;
                if (ip <= iend - 7) ip += 3;
                else { bitCount -= 8 * (iend - 7 - ip); bitCount &= 31; ip = iend - 4; }
                bitStream = rd32(src + ip) >> bitCount;
                repeats = (u32)__builtin_ctz(~bitStream | 0x80000000u) >> 1;
            }
            charnum += 3 * repeats;
            bitStream >>= 2 * repeats; bitCount += (int)(2 * repeats);
            charnum += bitStream & 3; bitCount += 2;
            if (charnum >= maxSV1) break;
            if (ip <= iend - 7 || ip + (bitCount >> 3) <= iend - 4) { ip += bitCount >> 3; bitCount &= 7; }
            else { bitCount -= 8 * (iend - 4 - ip); bitCount &= 31; ip = iend - 4; }
            bitStream = rd32(src + ip) >> bitCount;
        }
        {   int const max = (2 * threshold - 1) - remaining;
            int count;
            if ((bitStream & (u32)(threshold - 1)) < (u32)max) { count = (int)(bitStream & (u32)(threshold - 1)); bitCount += nbBits - 1; }
            else { count = (int)(bitStream & (u32)(2 * threshold - 1)); if (count >= threshold) count -= max; bitCount += nbBits; }
            count--;
            if (count >= 0) remaining -= count; else remaining += count;
            norm[charnum++] = (short)count;
            previous0 = !count;
            if (remaining < threshold) {
                if (remaining <= 1) break;
                nbBits = (int)hibit((u32)remaining) + 1; threshold = 1 << (nbBits - 1);
            }
            if (charnum >= maxSV1) break;
            if (ip <= iend - 7 || ip + (bitCount >> 3) <= iend - 4) { ip += bitCount >> 3; bitCount &= 7; }
            else { bitCount -= 8 * (iend - 4 - ip); bitCount &= 31; ip = iend - 4; }
            bitStream = rd32(src + ip) >> bitCount;
        }
    }
    if (remaining != 1) return ERR(corruption_detected);
    if (charnum > maxSV1) return ERR(maxSymbolValue_tooSmall);
    if (bitCount > 32) return ERR(corruption_detected);
    *maxSV = charnum - 1;
    return (size_t)(ip + ((bitCount + 7) >> 3));
}
static size_t read_ncount(short* norm, u32* maxSV, u32* tableLog, const u8* src, size_t srcSize) {
    if (srcSize < 8) {
        u8 pad[8] = { 0 }; size_t h;
        memcpy(pad, src, srcSize);
        h = read_ncount_body(norm, maxSV, tableLog, pad, 8);
        if (zso_is_error(h)) return h;
        return h > srcSize ? ERR(corruption_detected) : h;
    }
    return read_ncount_body(norm, maxSV, tableLog, src, srcSize);
}

/* ---------------------------------------------------------------- FSE decode tables ------ */
typedef struct { u16 next; u8 nbBits; u8 sym; } FseCell;     /* generic symbol table */

/* Spread + state assignment shared by the weights table and the sequence tables:
 * N/common/fse_decompress.c:58-160 == N/decompress/zstd_decompress_block.c:485-603 */
static int fse_build(FseCell* t, const short* norm, u32 maxSV, u32 tableLog) {
    u32 const size = 1u << tableLog, mask = size - 1;
    u32 const step = (size >> 1) + (size >> 3) + 3;      /* FSE_TABLESTEP, N/common/fse.h:623 */
    u16 symNext[256];
    u32 high = size - 1, s, pos = 0, u;
    for (s = 0; s <= maxSV; s++) {
        if (norm[s] == -1) { t[high--].sym = (u8)s; symNext[s] = 1; }
        else symNext[s] = (u16)norm[s];
    }
    for (s = 0; s <= maxSV; s++) {
        int i;
        for (i = 0; i < norm[s]; i++) {
            t[pos].sym = (u8)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    if (pos != 0) return -1;
    for (u = 0; u < size; u++) {
        u32 const ns = symNext[t[u].sym]++;
        t[u].nbBits = (u8)(tableLog - hibit(ns));
        t[u].next = (u16)((ns << t[u].nbBits) - size);
    }
    return 0;
}

/* ---------------------------------------------------------------- Huffman ---------------- */
typedef struct { u8 nbBits; u8 sym; } HufCell;
typedef struct { HufCell cell[1 << HUFLOG_MAX]; u32 log; int valid; int x2; } HufTable;   /* x2: the reference decodes it with its double-symbol table */

/* FSE-compressed weights: N/common/fse_decompress.c:166-236 (two interleaved states) */
static size_t fse_decode_weights(u8* out, size_t outCap, const u8* src, size_t srcSize) {
    short norm[256]; u32 maxSV = 255, tl; FseCell tab[64];
    size_t const h = read_ncount(norm, &maxSV, &tl, src, srcSize);
    BitR b; u32 s1, s2; size_t n = 0;
    if (zso_is_error(h)) return h;
    if (tl > 6) return ERR(tableLog_tooLarge);
    /* table + build workspace for (tableLog, maxSymbol) must fit what HUF_readStats owns, FSE_DECOMPRESS_WKSP_SIZE_U32(6, 11) = 219
     * words (N/common/fse_decompress.c:273, fse.h:267-273): maxSymbol <= 11 at tableLog 6, <= 91 at tableLog 5 */
    if ((1u + (1u << tl)) + 1u + ((2u * (maxSV + 1u) + (1u << tl) + 8u + 3u) >> 2) + 129u > 219u) return ERR(tableLog_tooLarge);
    if (h > srcSize) return ERR(corruption_detected);
    if (fse_build(tab, norm, maxSV, tl)) return ERR(GENERIC);
    if (bitr_init(&b, src + h, srcSize - h)) return ERR(corruption_detected);
    s1 = bitr_read(&b, tl); s2 = bitr_read(&b, tl);
    if (b.left < 0) return ERR(corruption_detected);
    for (;;) {
        if (n + 2 > outCap) return ERR(dstSize_tooSmall);
        out[n++] = tab[s1].sym; s1 = tab[s1].next + bitr_read(&b, tab[s1].nbBits);
        if (b.left < 0) { out[n++] = tab[s2].sym; break; }
        if (n + 2 > outCap) return ERR(dstSize_tooSmall);
        out[n++] = tab[s2].sym; s2 = tab[s2].next + bitr_read(&b, tab[s2].nbBits);
        if (b.left < 0) { out[n++] = tab[s1].sym; break; }
    }
    return n;
}

/* N/common/entropy_common.c:243-305 + N/decompress/huf_decompress.c:385-518 */
static size_t huf_read_table(HufTable* ht, const u8* src, size_t srcSize) {
    u8 w[256]; u32 rank[HUFLOG_MAX + 2]; size_t iSize, oSize; u32 total = 0, n, tl;
    if (!srcSize) return ERR(srcSize_wrong);
    iSize = src[0];
    if (iSize >= 128) {
        oSize = iSize - 127; iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize) return ERR(srcSize_wrong);
        if (oSize >= 256) return ERR(corruption_detected);
        for (n = 0; n < oSize; n += 2) { w[n] = src[1 + n / 2] >> 4; w[n + 1] = src[1 + n / 2] & 15; }
    } else {
        if (iSize + 1 > srcSize) return ERR(srcSize_wrong);
        oSize = fse_decode_weights(w, 255, src + 1, iSize);
        if (zso_is_error(oSize)) return oSize;
    }
    memset(rank, 0, sizeof(rank));
    for (n = 0; n < oSize; n++) {
        if (w[n] > HUFLOG_MAX) return ERR(corruption_detected);
        rank[w[n]]++; total += (1u << w[n]) >> 1;
    }
    if (total == 0) return ERR(corruption_detected);
    tl = hibit(total) + 1;
    if (tl > HUFLOG_MAX) return ERR(corruption_detected);
    {   u32 const rest = (1u << tl) - total; u32 const last = hibit(rest) + 1;
        if ((1u << hibit(rest)) != rest) return ERR(corruption_detected);
        w[oSize] = (u8)last; rank[last]++;
    }
    if (rank[1] < 2 || (rank[1] & 1)) return ERR(corruption_detected);
    if (tl > 11 + 1) return ERR(tableLog_tooLarge);   /* literals DTable is sized for log 11 (+1 slack as reference: maxTableLog+1 = 12) */
    /* fill: weights ascending, symbols ascending within a weight */
    {   u32 start[HUFLOG_MAX + 2], cur = 0, wv;
        for (wv = 1; wv <= tl; wv++) { start[wv] = cur; cur += rank[wv] << (wv - 1); }
        for (n = 0; n <= oSize; n++) {
            u32 const ww = w[n]; u32 len, k;
            if (!ww) continue;
            len = 1u << (ww - 1);
            for (k = 0; k < len; k++) { ht->cell[start[ww] + k].sym = (u8)n; ht->cell[start[ww] + k].nbBits = (u8)(tl + 1 - ww); }
            start[ww] += len;
        }
    }
    ht->log = tl; ht->valid = 1;
    return iSize + 1;
}

/* Which of its two table shapes the reference picks for a 4-stream literals section: HUF_selectDecoder,
 * N/decompress/huf_decompress.c:1793-1843 (timing model: {table build, per-256-symbols} for single / double symbol cells,
 * by compression-ratio bucket Q).  Both shapes decode a VALID stream to the same bytes; they differ on how a corrupted
 * stream's last symbol is judged (huf_decode_stream below), so the restatement has to pick the same one. */
static int huf_select_x2(size_t dstSize, size_t cSrcSize) {
    static const u16 t[16][4] = {
        {0,0,1,1}, {0,0,1,1}, {150,216,381,119}, {170,205,514,112}, {177,199,539,110}, {197,194,644,107}, {221,192,735,107},
        {256,189,881,106}, {359,188,1167,109}, {582,187,1570,114}, {688,187,1712,122}, {825,186,1965,136}, {976,185,2131,150},
        {1180,186,2070,175}, {1377,185,1731,202}, {1412,185,1695,202} };
    u32 const q = cSrcSize >= dstSize ? 15u : (u32)(cSrcSize * 16 / dstSize), d256 = (u32)(dstSize >> 8);
    u32 const t0 = t[q][0] + t[q][1] * d256; u32 t1 = t[q][2] + t[q][3] * d256;
    t1 += t1 >> 5;
    return t1 < t0;
}

/* one backward stream, exactly `n` symbols.  Single-symbol table: N/decompress/huf_decompress.c:600-640 (1X1 body): one code per
 * step, and the stream must end exactly (:697).
 * Double-symbol table (x2; HUF_decodeStreamX2 :1308-1349): one CELL per step.  The cell under the cursor (dtLog = 11 bits, zeros
 * below the stream's first bit) holds two codes when both fit (k1 + k2 <= dtLog, HUF_fillDTableX2 :1117-1177), and then both are
 * emitted and k1 + k2 bits skipped.  A valid stream decodes to the same bytes either way; a corrupted one is judged differently:
 *   - a two-code cell whose second code lies in the zero padding overruns the stream (rejected), and
 *   - when ONE byte is left to produce, HUF_decodeLastSymbolX2 (:1275-1290) emits the cell's first code and, for a two-code
 *     cell, skips k1 + k2 bits CLAMPED to the end of the stream — up to k2 left-over bits are accepted; with no bit left at all
 *     the cell index comes from the top of the last-loaded container (BIT_lookBitsFast with a shift of 64 & 63 = 0: the stream's
 *     first 8 bytes) and nothing is skipped. */
static int huf_decode_stream(u8* out, size_t n, const u8* src, size_t srcSize, const HufTable* ht) {
    BitR b; size_t i;
    if (bitr_init(&b, src, srcSize)) return -1;
    if (!ht->x2) {
        for (i = 0; i < n; i++) {
            HufCell const c = ht->cell[bitr_peek(&b, ht->log)];
            out[i] = c.sym; b.left -= c.nbBits;
        }
        return b.left == 0 ? 0 : -1;
    }
    {   u32 const D = ht->log > 11 ? ht->log : 11;                     /* HUF_readDTableX2_wksp:1208 builds the table at >= 11 bits */
        for (i = 0; i < n; ) {
            int const last = (i + 1 == n);
            u32 idx; HufCell c1, c2; int two;
            if (b.left < 0 || (!last && b.left == 0)) return -1;
            if (b.left > 0) idx = bitr_peek(&b, D);
            else { u64 c = 0; size_t k; for (k = 0; k < 8 && k < srcSize; k++) c |= (u64)src[k] << (8 * k); idx = (u32)(c >> (64 - D)); }
            c1 = ht->cell[idx >> (D - ht->log)];
            c2 = ht->cell[((idx << c1.nbBits) & ((1u << D) - 1)) >> (D - ht->log)];
            two = (u32)c1.nbBits + c2.nbBits <= D;
            out[i] = c1.sym;
            if (last) {
                if (!two) return b.left == (i64)c1.nbBits ? 0 : -1;
                return (b.left > 0 && (i64)(c1.nbBits + c2.nbBits) < b.left) ? -1 : 0;
            }
            if (two) { out[i + 1] = c2.sym; b.left -= c1.nbBits + c2.nbBits; i += 2; }
            else { b.left -= c1.nbBits; i += 1; }
        }
        return b.left == 0 ? 0 : -1;
    }
}

/* ---------------------------------------------------------------- sequences -------------- */
static const u32 LL_base[MAXLL + 1] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,
    48,64,0x80,0x100,0x200,0x400,0x800,0x1000,0x2000,0x4000,0x8000,0x10000 };
static const u8 LL_bits[MAXLL + 1] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
static const u32 ML_base[MAXML + 1] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,
    33,34,35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003 };
static const u8 ML_bits[MAXML + 1] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
    1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
/* predefined distributions, N/common/zstd_internal.h:113-165 */
static const short LL_defNorm[MAXLL + 1] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
static const short ML_defNorm[MAXML + 1] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
    1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
static const short OF_defNorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

typedef struct { FseCell cell[512]; u32 log; } SeqTable;

typedef struct {
    HufTable huf;
    SeqTable ll, of, ml;           /* "current" tables (what set_repeat re-uses) */
    int seqValid;                  /* dctx->fseEntropy */
    u32 rep[3];
    u8* lit; size_t litCap;
} DState;

static size_t build_seq_table(SeqTable* t, int type, u32 maxSym, u32 maxLog, const short* defNorm, u32 defMax, u32 defLog,
                              const u8* src, size_t srcSize, int repeatOK) {
    switch (type) {
    case 0: /* set_basic: predefined */
        if (fse_build(t->cell, defNorm, defMax, defLog)) return ERR(GENERIC);
        t->log = defLog; return 0;
    case 1: /* set_rle */
        if (!srcSize) return ERR(srcSize_wrong);
        if (src[0] > maxSym) return ERR(corruption_detected);
        t->cell[0].sym = src[0]; t->cell[0].nbBits = 0; t->cell[0].next = 0; t->log = 0; return 1;
    case 3: /* set_repeat */
        if (!repeatOK) return ERR(corruption_detected);
        return 0;
    default: {
        short norm[64]; u32 max = maxSym, tl;
        size_t const h = read_ncount(norm, &max, &tl, src, srcSize);
        if (zso_is_error(h)) return ERR(corruption_detected);
        if (h > srcSize) return ERR(corruption_detected);
        if (tl > maxLog) return ERR(corruption_detected);
        if (fse_build(t->cell, norm, max, tl)) return ERR(corruption_detected);
        t->log = tl; return h; }
    }
}

/* ---------------------------------------------------------------- one compressed block ---- */
/* room = bytes left in the destination; *inPlace = raw literals read straight from the block (zstd_decompress_block.c:283-288) */
static size_t decode_literals(DState* ds, const u8* src, size_t srcSize, const u8** litPtr, size_t* litSize, size_t blockSizeMax, size_t room, int* inPlace) {
    u32 type, fmt;
    *inPlace = 0;
    if (srcSize < 2) return ERR(corruption_detected);       /* MIN_CBLOCK_SIZE */
    type = src[0] & 3; fmt = (src[0] >> 2) & 3;
    if (type == 0 || type == 1) {                            /* raw / rle */
        size_t lh, n;
        if (fmt == 0 || fmt == 2) { lh = 1; n = src[0] >> 3; }
        else if (fmt == 1) { lh = 2; n = rd16(src) >> 4; }
        else { if (srcSize < 3) return ERR(corruption_detected); lh = 3; n = rd24(src) >> 4; }
        if (type == 1 && lh + 1 > srcSize) return ERR(corruption_detected);   /* RLE: the byte must be there before anything else, :310, :315 */
        if (n > blockSizeMax) return ERR(corruption_detected);
        if (n > room) return ERR(dstSize_tooSmall);             /* expectedWriteSize < litSize, :268 / :318 */
        if (type == 0) {
            if (lh + n > srcSize) return ERR(corruption_detected);
            *inPlace = (lh + n + 32 <= srcSize);
            *litPtr = src + lh; *litSize = n; return lh + n;
        }
        if (lh + 1 > srcSize) return ERR(corruption_detected);
        if (fmt == 1 && srcSize < 3) return ERR(corruption_detected);
        if (fmt == 3 && srcSize < 4) return ERR(corruption_detected);
        memset(ds->lit, src[lh], n); *litPtr = ds->lit; *litSize = n; return lh + 1;
    }
    {   size_t lh, n, c; int single = 0; u32 lhc; const u8* ip; size_t cLeft;
        if (type == 3 && !ds->huf.valid) return ERR(dictionary_corrupted);     /* tested before the size, :150-153 */
        if (srcSize < 5) return ERR(corruption_detected);
        lhc = rd32(src);
        if (fmt < 2) { single = !fmt; lh = 3; n = (lhc >> 4) & 0x3FF; c = (lhc >> 14) & 0x3FF; }
        else if (fmt == 2) { lh = 4; n = (lhc >> 4) & 0x3FFF; c = lhc >> 18; }
        else { lh = 5; n = (lhc >> 4) & 0x3FFFF; c = (lhc >> 22) + ((size_t)src[4] << 10); }
        if (n > blockSizeMax) return ERR(corruption_detected);
        if (!single && n < 6) return ERR(literals_headerWrong);      /* MIN_LITERALS_FOR_4_STREAMS */
        if (c + lh > srcSize) return ERR(corruption_detected);
        if (n > room) return ERR(dstSize_tooSmall);             /* :183 */
        ip = src + lh; cLeft = c;
        if (type == 2) {
            size_t const h = huf_read_table(&ds->huf, ip, cLeft);
            if (zso_is_error(h)) return ERR(corruption_detected);
            if (h > cLeft) return ERR(corruption_detected);
            ip += h; cLeft -= h;
            /* table shape: HUF_decompress4X_hufOnly_wksp (huf_decompress.c:1924-1944) asks HUF_selectDecoder with the section's
             * sizes (tree description included); a single stream always takes the one-symbol table (zstd_decompress_block.c:219) */
            ds->huf.x2 = single ? 0 : huf_select_x2(n, c);
        }
        if (single) {
            if (huf_decode_stream(ds->lit, n, ip, cLeft, &ds->huf)) return ERR(corruption_detected);
        } else {
            size_t l1, l2, l3, l4, seg = (n + 3) / 4;
            if (cLeft < 10) return ERR(corruption_detected);          /* jump table 6 + 4x>=1 */
            l1 = rd16(ip); l2 = rd16(ip + 2); l3 = rd16(ip + 4);
            if (6 + l1 + l2 + l3 > cLeft) return ERR(corruption_detected);
            l4 = cLeft - 6 - l1 - l2 - l3;
            if (3 * seg > n) return ERR(corruption_detected);
            if (huf_decode_stream(ds->lit, seg, ip + 6, l1, &ds->huf)
             || huf_decode_stream(ds->lit + seg, seg, ip + 6 + l1, l2, &ds->huf)
             || huf_decode_stream(ds->lit + 2 * seg, seg, ip + 6 + l1 + l2, l3, &ds->huf)
             || huf_decode_stream(ds->lit + 3 * seg, n - 3 * seg, ip + 6 + l1 + l2 + l3, l4, &ds->huf))
                return ERR(corruption_detected);
        }
        *litPtr = ds->lit; *litSize = n; return lh + c;
    }
}

/* `base` = first byte of this frame's output (window never reaches before it: no dictionary) */
static size_t decode_block(DState* ds, u8* base, u8* op, u8* oend, const u8* src, size_t srcSize, size_t blockSizeMax) {
    const u8* lit; size_t litSize; u8* const ostart = op; int inPlace;
    size_t const lsz = decode_literals(ds, src, srcSize, &lit, &litSize, blockSizeMax, (size_t)(oend - op), &inPlace);
    const u8* ip; const u8* iend = src + srcSize; int nbSeq;
    if (zso_is_error(lsz)) return lsz;
    /* how far the block may write: the reference parks regenerated literals 32 bytes past the largest block when the destination
     * has room (ZSTD_allocateLiteralsBuffer :86-94) and stops sequences there; otherwise at the destination's end */
    if (!inPlace && (size_t)(oend - op) > blockSizeMax + 64 + litSize) oend = op + blockSizeMax + 32;
    ip = src + lsz;
    if (ip >= iend) return ERR(srcSize_wrong);               /* MIN_SEQUENCES_SIZE */
    nbSeq = *ip++;
    if (nbSeq > 0x7F) {
        if (nbSeq == 0xFF) { if (ip + 2 > iend) return ERR(srcSize_wrong); nbSeq = (int)rd16(ip) + 0x7F00; ip += 2; }
        else { if (ip >= iend) return ERR(srcSize_wrong); nbSeq = ((nbSeq - 0x80) << 8) + *ip++; }
    }
    if (nbSeq == 0) {
        if (ip != iend) return ERR(corruption_detected);
    } else {
        u32 modes; size_t h; BitR b; u32 sLL, sOF, sML; int i;
        if (ip + 1 > iend) return ERR(srcSize_wrong);
        modes = *ip++;
        if (modes & 3) return ERR(corruption_detected);
        h = build_seq_table(&ds->ll, modes >> 6, MAXLL, LLFSELOG, LL_defNorm, MAXLL, 6, ip, (size_t)(iend - ip), ds->seqValid);
        if (zso_is_error(h)) return ERR(corruption_detected);
        ip += h;
        h = build_seq_table(&ds->of, (modes >> 4) & 3, MAXOFF, OFFFSELOG, OF_defNorm, 28, 5, ip, (size_t)(iend - ip), ds->seqValid);
        if (zso_is_error(h)) return ERR(corruption_detected);
        ip += h;
        h = build_seq_table(&ds->ml, (modes >> 2) & 3, MAXML, MLFSELOG, ML_defNorm, MAXML, 6, ip, (size_t)(iend - ip), ds->seqValid);
        if (zso_is_error(h)) return ERR(corruption_detected);
        ip += h;
        ds->seqValid = 1;
        if (oend == op) return ERR(dstSize_tooSmall);           /* sequences but no room at all, zstd_decompress_block.c:2119 */
        if (bitr_init(&b, ip, (size_t)(iend - ip))) return ERR(corruption_detected);
        sLL = seq_read(&b, ds->ll.log); sOF = seq_read(&b, ds->of.log); sML = seq_read(&b, ds->ml.log);
        for (i = 0; i < nbSeq; i++) {
            FseCell const cl = ds->ll.cell[sLL], co = ds->of.cell[sOF], cm = ds->ml.cell[sML];
            u32 const ofCode = co.sym, llCode = cl.sym, mlCode = cm.sym;
            u32 llen = LL_base[llCode], mlen = ML_base[mlCode]; size_t offset;
            /* offset: N/decompress/zstd_decompress_block.c:1279-1312 */
            if (ofCode > 1) {
                offset = ((size_t)1 << ofCode) - 3 + seq_read_fast(&b, ofCode);     /* OF_base[c] = 2^c - 3 */
                ds->rep[2] = ds->rep[1]; ds->rep[1] = ds->rep[0]; ds->rep[0] = (u32)offset;
            } else {
                u32 const ll0 = (llen == 0);
                if (ofCode == 0) {
                    offset = ds->rep[ll0]; ds->rep[1] = ds->rep[!ll0]; ds->rep[0] = (u32)offset;
                } else {
                    u32 const idx = 1 + ll0 + seq_read_fast(&b, 1);
                    u32 t = (idx == 3) ? ds->rep[0] - 1 : ds->rep[idx];
                    t -= !t;
                    if (idx != 1) ds->rep[2] = ds->rep[1];
                    ds->rep[1] = ds->rep[0]; ds->rep[0] = t; offset = t;
                }
            }
            if (ML_bits[mlCode]) mlen += seq_read_fast(&b, ML_bits[mlCode]);
            if (LL_bits[llCode]) llen += seq_read_fast(&b, LL_bits[llCode]);
            if (i + 1 < nbSeq) {
                sLL = cl.next + seq_read(&b, cl.nbBits);
                sML = cm.next + seq_read(&b, cm.nbBits);
                sOF = co.next + seq_read(&b, co.nbBits);
            }
            /* no test of b.left here: a stream that ran dry keeps yielding bits (above) until a check below, or the end test, fails */
            /* execute: N/decompress/zstd_decompress_block.c:1001-1096 */
            if ((size_t)(oend - op) < (size_t)llen + mlen) return ERR(dstSize_tooSmall);   /* the destination first, :919-920 */
            if (llen > litSize) return ERR(corruption_detected);
            memcpy(op, lit, llen); op += llen; lit += llen; litSize -= llen;
            if (offset > (size_t)(op - base)) return ERR(corruption_detected);
            { u32 k; const u8* m = op - offset; for (k = 0; k < mlen; k++) op[k] = m[k]; }
            op += mlen;
        }
        if (b.left != 0) return ERR(corruption_detected);
    }
    if (litSize > (size_t)(oend - op)) return ERR(dstSize_tooSmall);
    memcpy(op, lit, litSize); op += litSize;
    return (size_t)(op - ostart);
}

/* ---------------------------------------------------------------- frame ------------------- */
typedef struct { u64 contentSize; u64 windowSize; u32 dictID; int checksum; size_t headerSize; int skippable; u32 blockSizeMax; } FrameHdr;

/* N/decompress/zstd_decompress.c:447-557.  Returns 0, or error, or >0 = bytes needed. */
static size_t parse_frame_header(FrameHdr* fh, const u8* src, size_t srcSize) {
    u32 magic; u8 fhd; u32 dictIDcode, fcsID, single; size_t need, pos = 5;
    memset(fh, 0, sizeof(*fh));
    if (srcSize < 5) return ERR(srcSize_wrong);
    magic = rd32(src);
    if (magic != 0xFD2FB528u) {
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (srcSize < 8) return ERR(srcSize_wrong);
            fh->skippable = 1; fh->contentSize = rd32(src + 4); fh->headerSize = 8; return 0;
        }
        return ERR(prefix_unknown);
    }
    fhd = src[4]; dictIDcode = fhd & 3; fh->checksum = (fhd >> 2) & 1; single = (fhd >> 5) & 1; fcsID = fhd >> 6;
    if (fhd & 0x08) return ERR(frameParameter_unsupported);
    {   static const size_t did[4] = { 0, 1, 2, 4 }, fcs[4] = { 0, 2, 4, 8 };
        need = 5 + !single + did[dictIDcode] + fcs[fcsID] + (single && !fcsID); }
    if (srcSize < need) return ERR(srcSize_wrong);
    fh->headerSize = need;
    if (!single) {
        u8 const wd = src[pos++]; u32 const wl = (u32)(wd >> 3) + 10;
        if (wl > 31) return ERR(frameParameter_windowTooLarge);
        fh->windowSize = (u64)1 << wl; fh->windowSize += (fh->windowSize >> 3) * (wd & 7);
    }
    switch (dictIDcode) { case 1: fh->dictID = src[pos]; pos++; break; case 2: fh->dictID = rd16(src + pos); pos += 2; break;
                          case 3: fh->dictID = rd32(src + pos); pos += 4; break; default: break; }
    fh->contentSize = (u64)-1;
    switch (fcsID) { case 0: if (single) fh->contentSize = src[pos]; break; case 1: fh->contentSize = rd16(src + pos) + 256; break;
                     case 2: fh->contentSize = rd32(src + pos); break; default: fh->contentSize = rd64(src + pos); break; }
    if (single) fh->windowSize = fh->contentSize;
    fh->blockSizeMax = (u32)(fh->windowSize < BLOCK_MAX ? fh->windowSize : BLOCK_MAX);
    return 0;
}

unsigned long long zso_frame_content_size(const void* src, size_t srcSize) {
    FrameHdr fh; size_t const r = parse_frame_header(&fh, (const u8*)src, srcSize);
    if (zso_is_error(r)) return (unsigned long long)-2;
    if (fh.skippable) return 0;
    return fh.contentSize;
}

static size_t decode_frame(u8* dst, size_t dstCap, const u8* src, size_t srcSize, size_t* consumed) {
    FrameHdr fh; size_t r;
    const u8* ip; const u8* const iend = src + srcSize; u8* op = dst; u8* const oend = dst + dstCap;
    DState* ds;
    /* ZSTD_decompressFrame (N/decompress/zstd_decompress.c:966-979) sizes the header from its descriptor byte and wants it plus
     * one block header present BEFORE the magic number is examined */
    if (srcSize >= 5 && (rd32(src) & 0xFFFFFFF0u) != 0x184D2A50u) {
        static const size_t did[4] = { 0, 1, 2, 4 }, fcs[4] = { 0, 2, 4, 8 };
        u8 const fhd = src[4]; u32 const single = (fhd >> 5) & 1;
        size_t const need = 5 + !single + did[fhd & 3] + fcs[fhd >> 6] + (single && !(fhd >> 6));
        if (srcSize < 9 || srcSize < need + 3) return ERR(srcSize_wrong);
    }
    r = parse_frame_header(&fh, src, srcSize);
    if (zso_is_error(r)) return r;
    if (fh.skippable) {
        if (8 + fh.contentSize > srcSize) return ERR(srcSize_wrong);
        *consumed = 8 + (size_t)fh.contentSize; return 0;
    }
    if (fh.dictID) return ERR(dictionary_wrong);
    /* no window-size limit here: ZSTD_MAXWINDOWSIZE_DEFAULT is ZSTD_decompressStream's (:2231), the one-shot call has none */
    ds = (DState*)calloc(1, sizeof(DState));
    if (!ds) return ERR(GENERIC);
    ds->litCap = BLOCK_MAX + 32; ds->lit = (u8*)malloc(ds->litCap);
    ds->rep[0] = 1; ds->rep[1] = 4; ds->rep[2] = 8;         /* N/common/zstd_internal.h:65 */
    ip = src + fh.headerSize;
    for (;;) {
        u32 bh, last, type, sz;
        if ((size_t)(iend - ip) < 3) { r = ERR(srcSize_wrong); goto done; }
        bh = rd24(ip); ip += 3; last = bh & 1; type = (bh >> 1) & 3; sz = bh >> 3;
        if (type == 3) { r = ERR(corruption_detected); goto done; }
        if (type == 1) {                                        /* RLE */
            if (ip >= iend) { r = ERR(srcSize_wrong); goto done; }
            /* raw and RLE blocks are bounded by the destination only in the one-shot frame loop (N/decompress/zstd_decompress.c:1020-1026) */
            if (sz > (size_t)(oend - op)) { r = ERR(dstSize_tooSmall); goto done; }
            memset(op, *ip, sz); op += sz; ip += 1;
        } else {
            if (sz > (size_t)(iend - ip)) { r = ERR(srcSize_wrong); goto done; }
            if (type == 0) {
                if (sz > (size_t)(oend - op)) { r = ERR(dstSize_tooSmall); goto done; }
                memcpy(op, ip, sz); op += sz;
            } else {
                size_t d;
                if (sz > fh.blockSizeMax) { r = ERR(srcSize_wrong); goto done; }    /* zstd_decompress_block.c:2081 */
                /* a compressed block of exactly blockSizeMax is allowed since 1.5.4 (zstd_decompress_block.c:2073-2081): the block is entered and answers for itself */
                d = decode_block(ds, dst, op, oend, ip, sz, fh.blockSizeMax);
                if (zso_is_error(d)) { r = d; goto done; }
                op += d;
            }
            ip += sz;
        }
        if (last) break;
    }
    if (fh.contentSize != (u64)-1 && (u64)(op - dst) != fh.contentSize) { r = ERR(corruption_detected); goto done; }
    if (fh.checksum) {
        if ((size_t)(iend - ip) < 4) { r = ERR(checksum_wrong); goto done; }
        if ((u32)zso_xxh64(dst, (size_t)(op - dst), 0) != rd32(ip)) { r = ERR(checksum_wrong); goto done; }
        ip += 4;
    }
    *consumed = (size_t)(ip - src); r = (size_t)(op - dst);
done:
    free(ds->lit); free(ds);
    return r;
}

size_t zso_decompress(void* dst, size_t dstCap, const void* src, size_t srcSize) {
    const u8* ip = (const u8*)src; u8* op = (u8*)dst; size_t left = srcSize, room = dstCap; u32 frames = 0;
    while (left >= 5 || left > 0) {
        size_t used = 0; size_t const d = decode_frame(op, room, ip, left, &used);
        if (d == ERR(prefix_unknown) && frames) return ERR(srcSize_wrong);   /* garbage after a complete frame, ZSTD_decompressMultiFrame :1136 */
        if (zso_is_error(d)) return d;
        frames += !(left >= 4 && (rd32(ip) & 0xFFFFFFF0u) == 0x184D2A50u);
        op += d; room -= d; ip += used; left -= used;
        if (left == 0) break;
    }
    return (size_t)(op - (u8*)dst);
}

/* N/decompress/zstd_decompress.c:770-850 ZSTD_findFrameSizeInfo (compressed size only) */
size_t zso_find_frame_compressed_size(const void* src, size_t srcSize) {
    FrameHdr fh; const u8* ip = (const u8*)src; const u8* const iend = ip + srcSize;
    size_t const r = parse_frame_header(&fh, ip, srcSize);
    if (zso_is_error(r)) return r;
    if (fh.skippable) return 8 + (size_t)fh.contentSize <= srcSize ? 8 + (size_t)fh.contentSize : ERR(srcSize_wrong);
    ip += fh.headerSize;
    for (;;) {
        u32 bh, sz;
        if ((size_t)(iend - ip) < 3) return ERR(srcSize_wrong);
        bh = rd24(ip); ip += 3; sz = (((bh >> 1) & 3) == 1) ? 1 : (bh >> 3);
        if (((bh >> 1) & 3) == 3) return ERR(corruption_detected);
        if (sz > (size_t)(iend - ip)) return ERR(srcSize_wrong);
        ip += sz;
        if (bh & 1) break;
    }
    if (fh.checksum) { if ((size_t)(iend - ip) < 4) return ERR(srcSize_wrong); ip += 4; }
    return (size_t)(ip - (const u8*)src);
}

/* ---------------------------------------------------------------- XXH64 ------------------- */
/* Restatement of the XXH64 one-shot definition (N/common/xxhash.h, XXH64_endian_align). */
#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P4 0x85EBCA77C2B2AE63ULL
#define P5 0x27D4EB2F165667C5ULL
static u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
static u64 xx_round(u64 acc, u64 in) { acc += in * P2; acc = rotl64(acc, 31); return acc * P1; }
static u64 xx_merge(u64 acc, u64 v) { acc ^= xx_round(0, v); return acc * P1 + P4; }
uint64_t zso_xxh64(const void* data, size_t len, uint64_t seed) {
    const u8* p = (const u8*)data; const u8* const end = p + len; u64 h;
    if (len >= 32) {
        u64 v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const u8* const lim = end - 32;
        do { v1 = xx_round(v1, rd64(p)); v2 = xx_round(v2, rd64(p + 8)); v3 = xx_round(v3, rd64(p + 16)); v4 = xx_round(v4, rd64(p + 24)); p += 32; } while (p <= lim);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xx_merge(h, v1); h = xx_merge(h, v2); h = xx_merge(h, v3); h = xx_merge(h, v4);
    } else h = seed + P5;
    h += (u64)len;
    while (p + 8 <= end) { h ^= xx_round(0, rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (u64)rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p) * P5; h = rotl64(h, 11) * P1; p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

/* the NCount reader on its own, for tests/: header bytes or an error; norm[0..*maxSV], *tableLog filled on success */
size_t zso_read_ncount(short* norm, unsigned* maxSV, unsigned* tableLog, const void* src, size_t srcSize) {
    return read_ncount(norm, maxSV, tableLog, (const u8*)src, srcSize);
}
