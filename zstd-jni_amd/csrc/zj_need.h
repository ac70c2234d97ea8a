// zj_need.h — which hash-table accesses of the double-fast parse can matter at all (one flag byte per position of a frame).
//
// ZSTD_compressBlock_doubleFast (N/compress/zstd_double_fast.c:105-323) probes two tables at every position it visits and
// writes both; with the tables of 65 536 frames in HBM those four random requests per position are what bounds the lane-per-frame
// match kernel.  Most of them cannot change a decision:
//   * a probe at position p yields a match only if the entry's position q holds the bytes the reference compares — 8 for the long
//     table, 4 for the short one — and q sits in p's bucket.  If NO other position of the frame carries p's key (long: its 8 bytes;
//     short: its bucket and its first 4 bytes), the comparison fails whatever the entry holds: the probe need not be made
//     (ZN_NEED_L / ZN_NEED_S clear: the lane machine treats the entry as empty);
//   * an entry is only ever used by a probe that is made, so a bucket no needed probe falls into never has to be written
//     (ZN_INS_L / ZN_INS_S clear).  A bucket that is probed receives every write, so the probes that are made see exactly the
//     entries the reference's see.
// "No other position carries the key" is answered for all positions of a frame at once, order-free, with two blocked Bloom
// filters per table in LDS (64-bit blocks, four bits per key): B1 = keys seen, B2 = keys seen again (a key whose B1 bits were already set when it arrived).  A
// position whose key tests positive in B2 may have a twin (or is a false positive — then a probe is made that cannot match: harmless);
// a position whose key tests negative has none: filters have no false negatives, and the word-wide atomic OR of a blocked filter
// makes "already set" exact whichever of two twins arrives first.  Frame by frame: one workgroup of 1 024 lanes (the filters leave a CU room for one; 512 lanes: the flags of the metric batch arrive later and its compress call takes 148-152 ms instead of 135-143, profiles/r04/l_*), three sweeps.
// On the bench's mixed set 16 % / 27 % of the long / short probes and 37 % / 64 % of the writes remain (exact keys; the filters add
// a few per cent).  Decisions, their order and the frames are unchanged by construction, and checked byte for byte.
// Measured on the metric configuration: flags for the frames zn_worth() picks — match kernel 196 -> 165 ms, flag kernel 13.8 ms.
#pragma once

#define ZN_NEED_L 1u
#define ZN_NEED_S 2u
#define ZN_INS_L 4u
#define ZN_INS_S 8u
#define ZN_MAX_LOG_L 16u                 /* bucket bitmaps cover a long table of up to 2^16 entries (plain level 3: hashLog 16) ... */
#define ZN_MAX_LOG_S 15u                 /* ... and a short table of up to 2^15 (chainLog 15) */
#define ZN_FLAG_SLACK 16u                /* bytes behind a frame's flags that the lane machine may read (never interprets) */
#define ZN_FLAG_STRIDE (65536u + ZN_FLAG_SLACK)   /* flag bytes per frame slot of the match kernel's launch (frames up to 64 KiB) */

// Round 5: the "seen" filters are half as large again (6 144 blocks of 64 bits instead of 4 096: 140 KiB of LDS instead of 108, still one workgroup per CU).  With all 65 536 keys in,
// a 4 096-block filter is 63 % full and says "seen" to 16 % of the keys that were not; measured on the low-entropy class of the bench set the flags asked for a long probe at 5.6 % of
// the positions where 0.2 % have a twin, and for a short one at 19.9 % (exact: 15.9 %; 128 KiB frames: 28.0 % against 24.7 %) — every such probe is a round of the lane machine.
#define ZN_B1_BLOCKS 6144u               /* 3 x 2 048: zn_b1() */
struct ZNLds {
    u64 b1L[ZN_B1_BLOCKS], b2L[2048], b1S[ZN_B1_BLOCKS], b2S[2048];     // blocked Bloom filters: 64-bit blocks, 4 bits per key
    u32 bnL[1u << (ZN_MAX_LOG_L - 5u)], bnS[1u << (ZN_MAX_LOG_S - 5u)];   // buckets some needed probe falls into
};
// block of a key in a "seen" filter of 3 x `third` blocks (third a power of two): 11 / 12 bits pick the block inside a third, 16 more the third (a full-rate 24-bit multiply)
ZJ_DEV u32 zn_b1(u32 a, u32 third) { return (a & (third - 1u)) + third * ((((a >> 12) & 0xFFFFu) * 3u) >> 16); }

// Filter hashes.  32-bit multiplies run at a quarter of the VALU rate on gfx950 and this kernel is nothing but hashing, so the keys are
// hashed with as few of them as the filters tolerate: the long key reuses the product the bucket comes from (3 multiplies) plus one
// more over the folded halves; the short key — the bucket and the FOUR bytes the reference compares, not the five its hash covers — one
// multiply on top of its bucket's three.  24-bit multiplies (full rate) fold the pieces together.
struct ZNHash { u32 a, b; };                                  // a: block index + spare bits, b: the four bit positions
ZJ_DEV u32 zn_mul24(u32 x, u32 c) {
    return (x & 0xFFFFFFu) * (c & 0xFFFFFFu);           // both operands under 2^24: the compiler selects v_mul_u32_u24 (full rate)
}
ZJ_DEV ZNHash zn_hash_long(u32 prodHi, u64 w) {                // prodHi = zl_prod_hi(hL, w): every byte of w is in it
    ZNHash h; u32 const f = ((u32)w ^ ((u32)(w >> 32) * 0x85EBCA77u)) * 0x9E3779B1u;   // (2 multiplies)
    h.a = prodHi; h.b = f ^ (f >> 15) ^ zn_mul24(prodHi >> 8, 0x5BD1E9u);
    return h;
}
ZJ_DEV ZNHash zn_hash_short(u32 bucket, u32 v4) {
    ZNHash h; u32 const f = v4 * 0x9E3779B1u;                  // (1 multiply)
    h.a = (f ^ (f >> 15)) + zn_mul24(bucket, 0x10193u);
    h.b = (f >> 7) ^ zn_mul24((f ^ bucket) & 0xFFFFFFu, 0x5BD1E9u) ^ (bucket << 17);
    return h;
}
ZJ_DEV u64 zn_mask(u32 b) { return ((u64)1 << (b & 63u)) | ((u64)1 << ((b >> 6) & 63u)) | ((u64)1 << ((b >> 12) & 63u)) | ((u64)1 << ((b >> 18) & 63u)); }
ZJ_DEV ZNHash zn_second(ZNHash h) { ZNHash g; g.a = (h.b >> 3) ^ zn_mul24(h.a >> 4, 0x2C1B3Du); g.b = (h.a >> 5) ^ zn_mul24(h.b >> 6, 0x297A2Du) ^ (h.b << 11); return g; }   // for the "seen again" filter
#if !ZJ_ON_GPU
static inline u64 zn_atomic_or(u64* p, u64 v) { u64 const o = *p; *p = o | v; return o; }    // lane-serial build
#else
ZJ_DEV u64 zn_atomic_or(u64* p, u64 v) { return atomicOr((unsigned long long*)p, (unsigned long long)v); }
#endif

// T: the lanes that share the frame — T::count() of them, this one is t.id(); t.sync() is their barrier (LDS and the flag bytes this lane wrote)
template <class T>
ZJ_DEV void zn_flags_frame(const T& t, ZNLds& L, const u8* src, u32 n, u32 hashLog, u32 chainLog, u32 mls, u8* F) {
    u32 const npos = n >= 8u ? n - 7u : 0u;                 // positions with 8 readable bytes: all the parse can probe or insert (ip <= ilimit)
    ZLHash const hL = zl_hash_of(8, hashLog), hS = zl_hash_of(mls, chainLog);
    {   u32* const w = (u32*)&L; u32 const words = (u32)(sizeof(ZNLds) / 4u);
        for (u32 i = t.id(); i < words; i += t.count()) w[i] = 0; }
    t.sync();
    for (u32 p = t.id(); p < npos; p += t.count()) {         // sweep 1: every key into "seen"; a key that was there already into "seen again"
        u64 const w = ld64(src + p);
        ZNHash const kl = zn_hash_long(zl_prod_hi(hL, w), w), ks = zn_hash_short(zl_hash(hS, w), (u32)w);
        {   u64 const m = zn_mask(kl.b), old = zn_atomic_or(&L.b1L[zn_b1(kl.a, 2048u)], m);
            if ((old & m) == m) { ZNHash const g = zn_second(kl); zn_atomic_or(&L.b2L[g.a & 2047u], zn_mask(g.b)); } }
        {   u64 const m = zn_mask(ks.b), old = zn_atomic_or(&L.b1S[zn_b1(ks.a, 2048u)], m);
            if ((old & m) == m) { ZNHash const g = zn_second(ks); zn_atomic_or(&L.b2S[g.a & 2047u], zn_mask(g.b)); } }
    }
    t.sync();
    for (u32 p = t.id(); p < n; p += t.count()) {            // sweep 2: which probes are needed, and the buckets they fall into
        u32 f = 0;
        if (p < npos) {
            u64 const w = ld64(src + p);
            u32 const ph = zl_prod_hi(hL, w), bs = zl_hash(hS, w);
            {   ZNHash const g = zn_second(zn_hash_long(ph, w)); u64 const m = zn_mask(g.b);
                if ((L.b2L[g.a & 2047u] & m) == m) { u32 const b = ph >> hL.rsh; f |= ZN_NEED_L; atomicOr(&L.bnL[b >> 5], 1u << (b & 31u)); } }
            {   ZNHash const g = zn_second(zn_hash_short(bs, (u32)w)); u64 const m = zn_mask(g.b);
                if ((L.b2S[g.a & 2047u] & m) == m) { f |= ZN_NEED_S; atomicOr(&L.bnS[bs >> 5], 1u << (bs & 31u)); } }
        }
        F[p] = (u8)f;
    }
    t.sync();
    for (u32 p = t.id(); p < npos; p += t.count()) {         // sweep 3: which writes are needed (this lane wrote F[p] itself)
        u64 const w = ld64(src + p);
        u32 const bl = zl_hash(hL, w), bs = zl_hash(hS, w);
        u32 f = F[p];
        if ((L.bnL[bl >> 5] >> (bl & 31u)) & 1u) f |= ZN_INS_L;
        if ((L.bnS[bs >> 5] >> (bs & 31u)) & 1u) f |= ZN_INS_S;
        F[p] = (u8)f;
    }
    t.sync();
}
// Frames of 64 KiB + 1 .. 128 KiB (the wide launch): twice the keys, so a filter pair takes the room both pairs have above — the long table's flags and the short
// table's are computed one after the other, each with filters of 12 288 / 4 096 blocks over the same LDS (three sweeps per table; a sweep hashes for one table only,
// so the hashing per position is that of a 64 KiB frame's).  The flag byte is written by the long table's pass and completed by the short table's.
template <class T>
ZJ_DEV void zn_flags_frame_wide(const T& t, ZNLds& L, const u8* src, u32 n, u32 hashLog, u32 chainLog, u32 mls, u8* F) {
    u32 const npos = n >= 8u ? n - 7u : 0u;
    ZLHash const hL = zl_hash_of(8, hashLog), hS = zl_hash_of(mls, chainLog);
    u64* const b1 = (u64*)&L; u64* const b2 = b1 + 2u * ZN_B1_BLOCKS;       // 96 KiB + 32 KiB: the room of b1L .. b2S
    for (u32 table = 0; table < 2u; table++) {                 // 0: long, 1: short
        u32* const bn = table == 0 ? L.bnL : L.bnS;
        {   u32* const w = (u32*)&L; u32 const words = (u32)(sizeof(ZNLds) / 4u);
            for (u32 i = t.id(); i < words; i += t.count()) w[i] = 0; }
        t.sync();
        for (u32 p = t.id(); p < npos; p += t.count()) {       // sweep 1: every key into "seen"; a key that was there already into "seen again"
            u64 const w = ld64(src + p);
            ZNHash const k = table == 0 ? zn_hash_long(zl_prod_hi(hL, w), w) : zn_hash_short(zl_hash(hS, w), (u32)w);
            u64 const m = zn_mask(k.b), old = zn_atomic_or(&b1[zn_b1(k.a, 4096u)], m);
            if ((old & m) == m) { ZNHash const g = zn_second(k); zn_atomic_or(&b2[g.a & 4095u], zn_mask(g.b)); }
        }
        t.sync();
        for (u32 p = t.id(); p < n; p += t.count()) {          // sweep 2: which probes of this table are needed, and the buckets they fall into
            u32 f = table == 0 ? 0u : F[p];
            if (p < npos) {
                u64 const w = ld64(src + p);
                if (table == 0) {
                    u32 const ph = zl_prod_hi(hL, w); ZNHash const g = zn_second(zn_hash_long(ph, w)); u64 const m = zn_mask(g.b);
                    if ((b2[g.a & 4095u] & m) == m) { u32 const b = ph >> hL.rsh; f |= ZN_NEED_L; atomicOr(&bn[b >> 5], 1u << (b & 31u)); }
                } else {
                    u32 const bs = zl_hash(hS, w); ZNHash const g = zn_second(zn_hash_short(bs, (u32)w)); u64 const m = zn_mask(g.b);
                    if ((b2[g.a & 4095u] & m) == m) { f |= ZN_NEED_S; atomicOr(&bn[bs >> 5], 1u << (bs & 31u)); }
                }
            }
            F[p] = (u8)f;
        }
        t.sync();
        for (u32 p = t.id(); p < npos; p += t.count()) {       // sweep 3: which writes of this table are needed (this lane wrote F[p] itself)
            u64 const w = ld64(src + p);
            u32 const b = table == 0 ? zl_hash(hL, w) : zl_hash(hS, w);
            if ((bn[b >> 5] >> (b & 31u)) & 1u) F[p] = (u8)(F[p] | (table == 0 ? ZN_INS_L : ZN_INS_S));
        }
        t.sync();
    }
}
// Is the frame worth its flags?  They cost a pass of hashing over every position; they pay where the parse visits most positions and finds
// little: many distinct 4-byte values (few long matches) over a very small alphabet, where short repeats keep turning up and the reference's
// step rule (one more position skipped per 256 unmatched bytes) never gets going: measured on unstructured data, 16 byte values are searched at
// 0.29 positions per byte, 24 and more at 0.03 (they accelerate and are cheap anyway, like base64 or random bytes); text finds matches and needs
// nearly every probe.  So: at most 16 byte values and at least 1 in 2 four-byte values distinct in a sample of 1 024 positions from the middle of
// the frame — conservative on purpose: a frame picked in vain costs the flag pass (0.8 us amortised), a frame not picked costs nothing new.
// `bm` = 136 zeroable LDS words.  The answer is the same in every lane.  A heuristic: it decides speed only, never bytes.
#define ZN_SAMPLE 1024u
template <class T>
ZJ_DEV bool zn_worth(const T& t, u32* bm, const u8* src, u32 n) {
    if (n < 4096u) return false;
    for (u32 i = t.id(); i < 136u; i += t.count()) bm[i] = 0;
    t.sync();
    u32 const off = (n - 4u - ZN_SAMPLE) >> 1;
    for (u32 p = t.id(); p < ZN_SAMPLE; p += t.count()) {
        u32 const v = ld32(src + off + p), h = (v * 2654435761u) >> 20;
        atomicOr(&bm[h >> 5], 1u << (h & 31u));                    // 4 096-bit set of 4-byte values
        atomicOr(&bm[128u + ((v & 255u) >> 5)], 1u << (v & 31u));  // 256-bit set of byte values
    }
    t.sync();
    u32 grams = 0, bytes = 0;
    for (u32 i = 0; i < 128u; i++) grams += (u32)__builtin_popcount(bm[i]);
    for (u32 i = 128u; i < 136u; i++) bytes += (u32)__builtin_popcount(bm[i]);
    t.sync();
    return grams * 64u >= 32u * ZN_SAMPLE && bytes <= 16u;
}
ZJ_HD bool zn_takes(u32 hashLog, u32 chainLog, u32 srcSize) { return hashLog <= ZN_MAX_LOG_L && chainLog <= ZN_MAX_LOG_S && srcSize >= 64u && srcSize <= 65536u; }
ZJ_HD bool zn_takes_wide(u32 hashLog, u32 chainLog, u32 srcSize) { return hashLog <= ZN_MAX_LOG_L && chainLog <= ZN_MAX_LOG_S && srcSize > 65536u && srcSize <= 131072u; }      // zn_flags_frame_wide
#define ZN_FLAG_STRIDE_WIDE (131072u + ZN_FLAG_SLACK)   /* flag bytes per frame slot of the wide launch */
