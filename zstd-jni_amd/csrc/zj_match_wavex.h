// zj_match_wavex.h — wave-per-frame double-fast match finder for ONE BLOCK of a multi-block frame (level 3, inputs above 128 KiB),
// hash tables and frame in HBM.
//
// A multi-block frame is a chain: block k's parse reads the table entries and repcodes blocks 0 .. k-1 left behind
// (N/compress/zstd_compress.c:4591-4692, N/compress/zstd_double_fast.c:105-323), so a frame cannot be spread over lanes the way a batch
// of small frames is, and a frame's tables (2^17 + 2^16 entries) do not fit the LDS.  Rounds 1-3 ran the reference's loop on one lane of
// the frame's wave: every position a dependent chain of three to four HBM round trips.  Here the whole wave runs it, as zj_match_wave.h
// does for small frames: a WINDOW is the next <= 63 positions the reference's inner loop would visit if none of them matched (stride = the
// current step), one per lane, plus one look-ahead lane; every lane hashes its position, reads both table entries and its candidates'
// bytes, the lowest lane with a hit is where the reference's loop stops, the inserts of the lanes up to it are committed, the match is
// extended by the whole wave (64 x 8 bytes per trip, both directions at once), the complementary inserts and the immediate-repcode loop
// follow.  The decisions, their order and every table write are the reference's, so the records are identical to ze_block_dfast_x's
// (tests/test_emu_multiblock.py, tools/fuzz_emu_multiblock.py on the explicit-SIMT build; tests/test_gpu_multiblock.py on the GPU).
//
// What differs from the LDS matcher, because every table access is an HBM request here:
//   * no tentative inserts and read-backs: "the entry as the reference would find it" — the position of the latest lower lane of the
//     window with the same hash, else the table's content — comes from a scoreboard in LDS (every lane ORs its bit into slot
//     (hash & 511), reads the slot back and walks the few lower lanes named there), the table itself is read once and written once;
//   * the window is as wide as the data asks for: on text a match turns up within a few positions, and 63 speculative lanes would spend
//     250 random requests per sequence on entries nobody looks at (the device serves ~45 G random requests per second, DESIGN.md
//     section 4).  The width starts at 8 lanes, doubles after a window without a hit and falls back to twice the hit lane after one;
//     any width gives the same records;
//   * table writes are made by exactly one lane per address and step (a later lane of the window with the same hash shadows the earlier
//     one; the scalar inserts after a match are all lane 0's, in the reference's order), and a wait for the wave's outstanding stores
//     sits in front of every window's table reads, which bypass the CU's L1: a window reads what the previous one wrote.
#pragma once
#include "zj_simt.h"

#define ZX_SB_SLOTS 512u
#ifndef ZX_TAGS
#define ZX_TAGS 1            /* table entries carry an 11-bit tag of the bytes the reference compares; the window's winner is presumed from the tags */
#endif
// Table entry: position + 1 (21 bits: frames up to 2 MiB) | tag << 21.  The tag is a function of exactly the bytes the reference's probe
// compares (8 for the long table, 4 for the short one), so a tag mismatch proves the comparison fails (no false negatives) and a tag
// match is right but for one probe in 2 048.
#define ZX_POS(e) ((e) & 0x1FFFFFu)
#if ZX_TAGS
#define ZX_ENT(pos1, tag) ((pos1) | ((tag) << 21))
#else
#define ZX_ENT(pos1, tag) (pos1)
#endif
#define ZX_CAP_MIN 8u
#define ZX_CARRY 128u                                // bytes per staged stream
struct ZXLds {
    u64 SL[ZX_SB_SLOTS]; u64 SS[ZX_SB_SLOTS];       // scoreboards: lanes of the current window per hash slot
    u8 shadowL[64]; u8 shadowS[64];                  // commit: the lane is shadowed by a later committed lane with its hash
    // what the trip of a match staged (see run()): the frame's bytes from curr on, the same span one match offset back and one
    // previous-offset back, and the 64 bytes before curr and before the match source
    u8 stA[ZX_CARRY + 16u]; u8 stB[ZX_CARRY + 16u]; u8 stC[ZX_CARRY + 16u]; u8 stKA[64]; u8 stKB[64];
};

#if ZJ_ON_GPU
// table entry as the wave's own earlier stores left it: device-scope load (not served from the CU's L1)
#define ZX_TLOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// every vector-memory operation of the wave so far has completed (gfx9: stores count in vmcnt until they are acknowledged)
#define ZX_STORES_DONE() __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define ZX_TLOAD(p) (*(p))
#define ZX_STORES_DONE() ((void)0)
#endif

struct ZWaveX {
    ZXLds* L; const u8* base; u32 nf, start, n, ilimit; u32* HL; u32* HS; ZLHash hL, hS; ZEOut o;
    bool carryOn, cvalid; u32 cbase, coffB, coffC;      // staged streams (ZXLds::stA ..): valid, position of stA[0], offsets of stB / stC behind it
#if defined(ZX_PROFILE) && ZJ_ON_GPU       /* analysis builds (tools/ab_zxprof.sh): cycles per phase of a window, printed per block by the first workgroups */
    u64 pf[12]; u64 pT;
#define ZX_MARK(i) do { u64 const t_ = __builtin_readcyclecounter(); pf[i] += t_ - pT; pT = t_; } while (0)
#else
#define ZX_MARK(i) ((void)0)
#endif
#ifdef ZX_STATS
    u64 stPasses, stHitPasses, stTrips, stLanes, stSlow;
#define ZX_STAT(x) (x)
#else
#define ZX_STAT(x) ((void)0)
#endif
    // frame bytes [pos, pos + 8): never touches memory outside the frame; bytes before its start or past its end read as zero
    // (they can only lengthen a count beyond its limit, and every count is cut to its limit).  In two halves, so that a lane block
    // can issue all its loads (at()) before it waits for any of them (fix() — no branches: a branch per load puts a wait in
    // front of the next load, one round trip each instead of one for all).
    ZJ_DEV_MEMBER u32 at(u32 pos) const { i32 const p = (i32)pos, hi = (i32)(nf - 8u); return (u32)(p < 0 ? 0 : (p > hi ? hi : p)); }
    ZJ_DEVM u64 fix(u64 v, u32 pos, u32 q) {
        i32 const dl = (i32)pos - (i32)q;                   // > 0: bytes dropped at the low end; < 0: bytes of zero fill at the low end
        u32 const a = dl > 0 ? (u32)dl : 0u, b = dl < 0 ? (u32)(0 - dl) : 0u;
        u64 const r = (v >> ((8u * a) & 63u)) << ((8u * b) & 63u);
        return (a | b) >= 8u ? 0 : r;
    }
    ZJ_DEV_MEMBER u64 fb(u32 pos) const { u32 const q = at(pos); return fix(ld64(base + q), pos, q); }
#define ZX_LOAD2(xa, pa, xb, pb) u64 xa, xb; { u32 const pa_ = (pa), pb_ = (pb), qa_ = at(pa_), qb_ = at(pb_); u64 const ra_ = ld64(base + qa_), rb_ = ld64(base + qb_); \
                                   ZW_FENCE2(ra_, rb_); xa = fix(ra_, pa_, qa_); xb = fix(rb_, pb_, qb_); }
    // length of the common prefix of [a..] and [b..] (b < a), at most n - a: ZSTD_count(a, b, iend)
    ZJ_DEV_MEMBER u32 count_fwd(u32 a, u32 b) {
        u32 const lim = n - a;
        for (u32 total = 0;; total += 512u) {
            ZWV<u64> d; ZWV<bool> ne;
            ZW_LANES(l) {
                ZX_LOAD2(ra, a + total + 8u * l, rb, b + total + 8u * l);
                d[l] = ra ^ rb; ne[l] = d[l] != 0;
            }
            ZX_STAT(stTrips++);
            u64 const m = zw_ballot(ne);
            if (m) { u32 const j = (u32)__builtin_ctzll(m); u32 const c = total + 8u * j + ((u32)__builtin_ctzll(zw_get64(d, j)) >> 3); return ZJ_UNI(c < lim ? c : lim); }
            if (total + 512u >= lim) return ZJ_UNI(lim);
        }
    }
    // number of equal bytes going backwards from (ipos - 1, mpos - 1), at most limit (<= mpos < ipos)
    ZJ_DEV_MEMBER u32 count_back(u32 ipos, u32 mpos, u32 limit) {
        if (limit == 0) return 0;
        for (u32 total = 0;; total += 512u) {
            ZWV<u64> d; ZWV<bool> ne;
            ZW_LANES(l) {
                ZX_LOAD2(ra, ipos - total - 8u - 8u * l, rb, mpos - total - 8u - 8u * l);      // (before the frame: zero on both sides, cut by `limit`)
                d[l] = ra ^ rb; ne[l] = d[l] != 0;
            }
            ZX_STAT(stTrips++);
            u64 const m = zw_ballot(ne);
            if (m) { u32 const j = (u32)__builtin_ctzll(m); u32 const c = total + 8u * j + ((u32)__builtin_clzll(zw_get64(d, j)) >> 3); return ZJ_UNI(c < limit ? c : limit); }
            if (total + 512u >= limit) return ZJ_UNI(limit);
        }
    }
    // Both directions of up to two candidate matches in ONE trip: lanes 0-15 count forwards from (a0, b0), lanes 16-31 backwards from
    // (i0, m0), lanes 32-47 / 48-63 the same for the second candidate (two = false: idle).  128 bytes per direction; a count that
    // runs through all of them continues in the general loops above.
    ZJ_DEV_MEMBER void extend(u32 a0, u32 b0, u32 i0, u32 m0, u32 lim0, bool two, u32 a1, u32 b1, u32 i1, u32 m1, u32 lim1,
                              u32& f0, u32& k0, u32& f1, u32& k1) {
        ZWV<u64> d; ZWV<bool> ne;
        ZW_LANES(l) {
            u32 const j = l & 15u, q = l >> 4;
            u32 pa, pb;
            if (q == 0u) { pa = a0 + 8u * j; pb = b0 + 8u * j; } else if (q == 1u) { pa = i0 - 8u - 8u * j; pb = m0 - 8u - 8u * j; }
            else if (q == 2u) { pa = a1 + 8u * j; pb = b1 + 8u * j; } else { pa = i1 - 8u - 8u * j; pb = m1 - 8u - 8u * j; }
            bool const on = two || q < 2u;
            ZX_LOAD2(xa, on ? pa : 8u, xb, on ? pb : 8u);
            u64 const x = xa ^ xb;
            d[l] = x; ne[l] = x != 0;
        }
        ZX_STAT(stTrips++);
        u64 const m = zw_ballot(ne);
        {   u32 const mm = (u32)m & 0xFFFFu, fl = n - a0;
            if (mm) { u32 const j = (u32)__builtin_ctz(mm); f0 = 8u * j + ((u32)__builtin_ctzll(zw_get64(d, j)) >> 3); if (f0 > fl) f0 = fl; }
            else f0 = fl <= 128u ? fl : 128u + count_fwd(a0 + 128u, b0 + 128u);
        }
        {   u32 const mm = (u32)(m >> 16) & 0xFFFFu;
            if (mm) { u32 const j = (u32)__builtin_ctz(mm); k0 = 8u * j + ((u32)__builtin_clzll(zw_get64(d, 16u + j)) >> 3); if (k0 > lim0) k0 = lim0; }
            else k0 = lim0 <= 128u ? lim0 : 128u + count_back(i0 - 128u, m0 - 128u, lim0 - 128u);
        }
        f1 = k1 = 0;
        if (two) {
            {   u32 const mm = (u32)(m >> 32) & 0xFFFFu, fl = n - a1;
                if (mm) { u32 const j = (u32)__builtin_ctz(mm); f1 = 8u * j + ((u32)__builtin_ctzll(zw_get64(d, 32u + j)) >> 3); if (f1 > fl) f1 = fl; }
                else f1 = fl <= 128u ? fl : 128u + count_fwd(a1 + 128u, b1 + 128u);
            }
            {   u32 const mm = (u32)(m >> 48) & 0xFFFFu;
                if (mm) { u32 const j = (u32)__builtin_ctz(mm); k1 = 8u * j + ((u32)__builtin_clzll(zw_get64(d, 48u + j)) >> 3); if (k1 > lim1) k1 = lim1; }
                else k1 = lim1 <= 128u ? lim1 : 128u + count_back(i1 - 128u, m1 - 128u, lim1 - 128u);
            }
        }
        f0 = ZJ_UNI(f0); k0 = ZJ_UNI(k0); f1 = ZJ_UNI(f1); k1 = ZJ_UNI(k1);      // wave-uniform by construction (ballots, read lanes): scalar registers, scalar branches
    }
    // ONE trip for everything a match at curr (source mpos) and what follows it can need: the frame from curr on (lanes 0-15), the same
    // span from the match source on (16-31) and one previous offset back (32-47: the immediate-repcode candidate), the 64 bytes before curr
    // and before the match source (48-63).
    ZJ_DEV_MEMBER void stage(ZXLds& lds, u32 curr, u32 mpos, u32 offC) {
        ZW_LANES(l) {
            u32 const j = l & 15u, q = l >> 4;
            u32 const p = q == 0u ? curr + 8u * j : (q == 1u ? mpos + 8u * j : (q == 2u ? curr - offC + 8u * j : (j < 8u ? curr - 64u + 8u * j : mpos - 128u + 8u * j)));
            u64 const x = fb(p);
            ZW_FENCE2(x, x);
            u8* const to = q == 0u ? lds.stA + 8u * j : (q == 1u ? lds.stB + 8u * j : (q == 2u ? lds.stC + 8u * j : (j < 8u ? lds.stKA + 8u * j : lds.stKB + 8u * (j - 8u))));
            st64(to, x);
        }
        ZX_STAT(stTrips++);
        ZW_SYNC();
    }
    // literal positions in the records are relative to the block
    ZJ_DEV_MEMBER void store(u32 litPos, u32 ll, u32 offBase, u32 ml) {
        ZW_LANES(l) { if (l == 0) { ZESeq s; s.ll = ll; s.ml = ml; s.off = offBase; s.pos = litPos - start; o.litOff[o.n] = o.lit; o.seqs[o.n] = s; } }
        o.n++; o.lit += ll;
    }

    // long-table bucket and tag from ONE product (bucket = its top hashLog bits, tag = the next 11); short-table tag from the 4 compared bytes
    ZJ_DEV_MEMBER u32 tagL_of(u32 prod) const { return (prod >> (hL.rsh - 11u)) & 0x7FFu; }
    ZJ_DEVM u32 tagS_of(u64 w) { return ((u32)w * 2246822519u) >> 21; }
    ZJ_DEV_MEMBER u32 entL_of(u64 w, u32 pos1, u32& bucket) const { u32 const p = zl_prod_hi(hL, w); bucket = p >> hL.rsh; return ZX_ENT(pos1, tagL_of(p)); }
    ZJ_DEV_MEMBER u32 entS_of(u64 w, u32 pos1, u32& bucket) const { bucket = zl_hash(hS, w); return ZX_ENT(pos1, tagS_of(w)); }
    // the block frame[blkStart, blkEnd); returns the length of the last literal run.  repIn / repOut as ze_block_dfast_x.
    ZJ_DEV_MEMBER u32 run(ZXLds& lds, const u8* frame, u32 frameSize, u32 blkStart, u32 blkEnd, u32 hBitsL, u32 hBitsS, u32 mls,
                          u32* hashLong, u32* hashSmall, const u32* repIn, u32* repOut, bool carry = true) {
        carryOn = carry; cvalid = false; cbase = coffB = coffC = 0;
        L = &lds; base = frame; nf = frameSize; start = blkStart; n = blkEnd; ilimit = blkEnd - 8u; HL = hashLong; HS = hashSmall;
        hL = zl_hash_of(8, hBitsL); hS = zl_hash_of(mls, hBitsS);
        o.n = 0; o.lit = 0;
#ifdef ZX_STATS
        stPasses = stHitPasses = stTrips = stLanes = stSlow = 0;
#endif
        ZW_LANES(l) { for (u32 i = l; i < ZX_SB_SLOTS; i += 64u) { lds.SL[i] = 0; lds.SS[i] = 0; } }
        ZW_SYNC();
#if defined(ZX_PROFILE) && ZJ_ON_GPU
        for (int j = 0; j < 12; j++) pf[j] = 0;
        pT = __builtin_readcyclecounter(); u64 const pStart = pT;
#endif
        u32 ip = blkStart + (blkStart == 0u ? 1u : 0u), anchor = blkStart;
        u32 off1 = ZJ_UNI(repIn[0]), off2 = ZJ_UNI(repIn[1]), saved1 = 0, saved2 = 0;
        {   u32 const maxRep = ip;                                          // zstd_double_fast.c:153-163: offsets beyond the data seen so far are parked
            if (off2 > maxRep) { saved2 = off2; off2 = 0; }
            if (off1 > maxRep) { saved1 = off1; off1 = 0; } }
        u32 step = 1, nextStep = ip + 256u, cap = ZX_CAP_MIN;
        if (blkEnd >= 8u) for (;;) {
            ip = ZJ_UNI(ip); anchor = ZJ_UNI(anchor); off1 = ZJ_UNI(off1); off2 = ZJ_UNI(off2); step = ZJ_UNI(step); nextStep = ZJ_UNI(nextStep); cap = ZJ_UNI(cap);
            o.n = ZJ_UNI(o.n); o.lit = ZJ_UNI(o.lit);                       // wave-uniform by construction; this tells the compiler (scalar registers, scalar branches)
            if (ip + step > ilimit) break;                                // ip1 > ilimit: _cleanup
            // ---- window: lanes 0..nIter-1 are the reference's next iterations (ip = p, ip1 = p + step), lane nIter looks ahead
            u32 kmax, room;
            if (step == 1u) { kmax = nextStep > ip ? nextStep - ip : 1u; room = ilimit - ip; }
            else { kmax = nextStep > ip ? (nextStep - ip + step - 1u) / step : 1u; if (kmax < 1u) kmax = 1u; room = (ilimit - ip) / step; }
            u32 nIter = kmax < cap ? kmax : cap; if (room < nIter) nIter = room;
            ZX_STAT(stPasses++); ZX_STAT(stLanes += nIter);
            ZWV<u32> pos, hl, hs, tgL, tgS, eL, eS, rb, predL, predS; ZWV<u64> w, mL, mS;
            ZWV<u32> cL, cS, kind; ZWV<bool> hit, longHit;
            cvalid = ZJ_UNI(cvalid ? 1u : 0u) != 0u; cbase = ZJ_UNI(cbase); coffB = ZJ_UNI(coffB); coffC = ZJ_UNI(coffC);
            if (cvalid && step == 1u && ip >= cbase && (ip - cbase) + nIter + 8u <= ZX_CARRY && (off1 == coffB || off1 == coffC || off1 == 0u)) {
                // the window lies inside the span the last match staged: its bytes and the repcode stream's come from LDS
                const u8* const rs = (off1 == coffB) ? lds.stB : lds.stC;   // (off1 == 0: never compared)
                ZW_LANES(l) {
                    bool const act = l <= nIter;
                    u32 const pp = act ? ip + l : ip; pos[l] = pp;
                    u64 const ww = ld64(lds.stA + (pp - cbase)); u32 const rr = ld32(rs + (pp + 1u - cbase));
                    ZW_FENCE2(ww, rr);
                    w[l] = ww; rb[l] = rr;
                    { u32 const pr = zl_prod_hi(hL, ww); hl[l] = pr >> hL.rsh; tgL[l] = tagL_of(pr); } hs[l] = zl_hash(hS, ww); tgS[l] = tagS_of(ww);
                    predL[l] = 64u; predS[l] = 64u;
                }
            } else {
                ZW_LANES(l) {
                    bool const act = l <= nIter;
                    u32 const pp = act ? ip + l * step : ip; pos[l] = pp;
                    ZX_LOAD2(ww, pp, rw, pp + 1u - off1);
                    u32 const rr = (u32)rw;
                    w[l] = ww; rb[l] = rr;
                    { u32 const pr = zl_prod_hi(hL, ww); hl[l] = pr >> hL.rsh; tgL[l] = tagL_of(pr); } hs[l] = zl_hash(hS, ww); tgS[l] = tagS_of(ww);
                    predL[l] = 64u; predS[l] = 64u;
                }
                ZX_STAT(stTrips++);
            }
            ZX_MARK(0);
            // ---- the table as the previous windows left it, and who in this window comes before whom
            ZX_STORES_DONE();
            ZX_MARK(9);
            ZW_LANES(l) {
                bool const act = l <= nIter, srch = l < nIter;
                u32 const a = hl[l], b = hs[l];
                u32 const x = act ? ZX_TLOAD(&HL[a]) : 0u, y = srch ? ZX_TLOAD(&HS[b]) : 0u;
                if (act) zw_or64(&lds.SL[a & (ZX_SB_SLOTS - 1u)], 1ull << l);      // (the look-ahead lane only reads its slot; its own bit is above every reader's mask)
                if (srch) zw_or64(&lds.SS[b & (ZX_SB_SLOTS - 1u)], 1ull << l);
                ZW_FENCE2(x, y);
                eL[l] = x; eS[l] = y;
            }
            ZX_STAT(stTrips++);
            ZX_MARK(1);
            ZW_SYNC();
            ZW_LANES(l) {
                u64 const below = (1ull << l) - 1ull;
                mL[l] = (l <= nIter) ? (lds.SL[hl[l] & (ZX_SB_SLOTS - 1u)] & below) : 0ull;
                mS[l] = (l < nIter) ? (lds.SS[hs[l] & (ZX_SB_SLOTS - 1u)] & below) : 0ull;
            }
            ZW_SYNC();
            ZW_LANES(l) { if (l <= nIter) lds.SL[hl[l] & (ZX_SB_SLOTS - 1u)] = 0; if (l < nIter) lds.SS[hs[l] & (ZX_SB_SLOTS - 1u)] = 0; }
            ZW_SYNC();
            for (;;) {                                                       // slots are shared by different hashes, equal hashes always share a slot
                ZWV<u32> jL, jS, gL, gS; ZWV<bool> pend;
                ZW_LANES(l) {
                    jL[l] = mL[l] ? 63u - (u32)__builtin_clzll(mL[l]) : l; jS[l] = mS[l] ? 63u - (u32)__builtin_clzll(mS[l]) : l;
                    pend[l] = (mL[l] | mS[l]) != 0;
                }
                if (!zw_ballot(pend)) break;
                ZX_STAT(stSlow++);
                zw_shfl(gL, hl, jL); zw_shfl(gS, hs, jS);
                ZW_LANES(l) {
                    if (mL[l]) { if (gL[l] == hl[l]) { predL[l] = jL[l]; mL[l] = 0; } else mL[l] &= ~(1ull << jL[l]); }
                    if (mS[l]) { if (gS[l] == hs[l]) { predS[l] = jS[l]; mS[l] = 0; } else mS[l] &= ~(1ull << jS[l]); }
                }
            }
            ZX_MARK(2);
            u64 hm = 0; u32 cnt = nIter; bool needExact = true, staged = false;
#if ZX_TAGS
            if (carryOn) {
                // ---- the window's winner, presumed: a lower lane of the window as entry is compared exactly (its word is in a register), a
                //      table entry by its tag; the repcode test is exact.  No presumed hit below lane K = no hit below lane K.
                ZWV<u32> wlo, whi, pwlo, pwhi, pslo; ZWV<bool> anyPred;
                ZW_LANES(l) { wlo[l] = (u32)w[l]; whi[l] = (u32)(w[l] >> 32); pwlo[l] = pwhi[l] = pslo[l] = 0; anyPred[l] = predL[l] < 64u || predS[l] < 64u; }
                if (zw_ballot(anyPred)) {                                // (most windows have no two lanes with one hash: no words to fetch from other lanes)
                    ZWV<u32> iL, iS;
                    ZW_LANES(l) { iL[l] = predL[l] < 64u ? predL[l] : l; iS[l] = predS[l] < 64u ? predS[l] : l; }
                    zw_shfl(pwlo, wlo, iL); zw_shfl(pwhi, whi, iL); zw_shfl(pslo, wlo, iS); }
                ZW_LANES(l) {
                    bool const act = l <= nIter, srch = l < nIter;
                    bool pL, pS; u32 a, b;
                    if (predL[l] < 64u) { pL = pwlo[l] == wlo[l] && pwhi[l] == whi[l]; a = ip + predL[l] * step; }
                    else { u32 const e = eL[l]; pL = ZX_POS(e) > 1u && (e >> 21) == tgL[l]; a = ZX_POS(e) - 1u; }
                    if (predS[l] < 64u) { pS = pslo[l] == wlo[l]; b = ip + predS[l] * step; }
                    else { u32 const e = eS[l]; pS = ZX_POS(e) > 1u && (e >> 21) == tgS[l]; b = ZX_POS(e) - 1u; }
                    pL = pL && act; pS = pS && srch;
                    cL[l] = a; cS[l] = b;
                    bool const hR = srch && off1 > 0u && rb[l] == (u32)(w[l] >> 8);
                    longHit[l] = pL;
                    kind[l] = hR ? 1u : (pL ? 2u : 3u);
                    hit[l] = srch && (hR || pL || pS);
                }
                u64 const pm = zw_ballot(hit);
                if (!pm) needExact = false;                              // (hm = 0, cnt = nIter)
                else {
                    u32 const K = (u32)__builtin_ctzll(pm), curr = ip + K * step, kd = zw_get(kind, K);
                    if (!(kd == 3u && zw_getb(longHit, K + 1u))) {         // (a presumed long match at ip1 as well: the exact tests decide)
                        u32 const mpos = kd == 2u ? zw_get(cL, K) : (kd == 3u ? zw_get(cS, K) : curr - off1);
                        cvalid = false;                                  // (the staged spans are replaced; valid again once the match is settled)
                        stage(lds, curr, mpos, kd == 1u ? off2 : off1);
                        // the presumption checked on the staged bytes: 8 (long) or 4 (short) at the candidate against those at curr
                        bool ok = true;
                        if (kd == 2u) ok = ZJ_UNI((u32)(ld64(lds.stA) == ld64(lds.stB))) != 0u;
                        else if (kd == 3u) ok = ZJ_UNI((u32)(ld32(lds.stA) == ld32(lds.stB))) != 0u;
                        if (ok) { hm = pm; cnt = K + 1u; needExact = false; staged = true; }
                    }
                }
            }
#endif
            if (needExact) {
                // ---- candidates (entry = position + 1; position 0 is never inserted), their bytes, the three tests of an iteration
                ZW_LANES(l) {
                    bool const act = l <= nIter, srch = l < nIter;
                    u32 const entL = predL[l] < 64u ? ip + predL[l] * step + 1u : ZX_POS(eL[l]), entS = predS[l] < 64u ? ip + predS[l] * step + 1u : ZX_POS(eS[l]);
                    bool const vL = act && entL > 1u, vS = srch && entS > 1u;
                    u32 const a = entL - 1u, b = entS - 1u;
                    cL[l] = a; cS[l] = b;
                    ZX_LOAD2(cl, vL ? a : 0u, csw, vS ? b : 0u);
                    u32 const cs = (u32)csw;
                    bool const hR = srch && off1 > 0u && rb[l] == (u32)(w[l] >> 8);
                    bool const hLg = vL && cl == w[l], hSh = vS && cs == (u32)w[l];
                    longHit[l] = hLg;
                    kind[l] = hR ? 1u : (hLg ? 2u : 3u);
                    hit[l] = srch && (hR || hLg || hSh);
                }
                ZX_STAT(stTrips++);
                hm = zw_ballot(hit);
                cnt = hm ? (u32)__builtin_ctzll(hm) + 1u : nIter;            // lanes whose inserts happen
            }
            ZX_MARK(3);
            // ---- commit: HL[hl] = HS[hs] = position + 1 for lanes < cnt; of several lanes with one hash the last one writes
            {   ZWV<bool> hasPred;
                ZW_LANES(l) { hasPred[l] = l < cnt && (predL[l] < 64u || predS[l] < 64u); }
                if (zw_ballot(hasPred)) {
                    ZW_LANES(l) { lds.shadowL[l] = 0; lds.shadowS[l] = 0; }
                    ZW_SYNC();
                    ZW_LANES(l) { if (l < cnt) { if (predL[l] < 64u) lds.shadowL[predL[l]] = 1; if (predS[l] < 64u) lds.shadowS[predS[l]] = 1; } }
                    ZW_SYNC();
                    ZW_LANES(l) { if (l < cnt) { if (!lds.shadowL[l]) HL[hl[l]] = ZX_ENT(pos[l] + 1u, tgL[l]); if (!lds.shadowS[l]) HS[hs[l]] = ZX_ENT(pos[l] + 1u, tgS[l]); } }
                    ZW_SYNC();
                } else {
                    ZW_LANES(l) { if (l < cnt) { HL[hl[l]] = ZX_ENT(pos[l] + 1u, tgL[l]); HS[hs[l]] = ZX_ENT(pos[l] + 1u, tgS[l]); } }
                }
            }
            ZX_MARK(4);
            if (!hm) {                                                       // nobody matched: the loop's own bookkeeping
                u32 const pN = ip + nIter * step;
                if (pN >= nextStep) { step++; nextStep += 256u; }
                ip = pN;
                cap = cap * 2u < 63u ? cap * 2u : 63u;
                continue;
            }
            ZX_STAT(stHitPasses++);
            {   u32 const c2 = 2u * cnt; cap = c2 < ZX_CAP_MIN ? ZX_CAP_MIN : (c2 < 63u ? c2 : 63u); }
            // ---- the match at lane K, as the reference handles it
            u32 const K = cnt - 1u, curr = ip + K * step, ip1 = curr + step, kd = zw_get(kind, K);
            u32 mip, mLength;
            bool const two = kd == 3u && zw_getb(longHit, K + 1u);           // _search_next_long: the long candidate of ip1 has to be counted as well
            if (carryOn && !two) {
                // ONE trip for everything this match and what follows it can need: the frame from curr on (lanes 0-15), the same span
                // one match offset back (16-31: the match source) and one previous offset back (32-47: the immediate-repcode candidate),
                // the 64 bytes before curr and before the match source (48-63).  Forward and backward counts, the complementary
                // inserts, the repcode test behind the match and the next window are then LDS reads as long as they stay inside.
                u32 const mpos = kd == 2u ? zw_get(cL, K) : (kd == 3u ? zw_get(cS, K) : curr - off1);
                u32 const offN = curr - mpos, offC = kd == 1u ? off2 : off1, dd = kd == 1u ? 5u : (kd == 2u ? 8u : 4u);
                if (!staged) stage(lds, curr, mpos, offC);
                ZWV<u64> d2; ZWV<bool> ne2;
                ZW_LANES(l) {
                    u64 x = 0;
                    if (l < 15u) x = ld64(lds.stA + dd + 8u * l) ^ ld64(lds.stB + dd + 8u * l);
                    else if (l >= 16u && l < 24u) x = ld64(lds.stKA + 56u - 8u * (l - 16u)) ^ ld64(lds.stKB + 56u - 8u * (l - 16u));
                    d2[l] = x; ne2[l] = x != 0;
                }
                ZW_SYNC();
                u64 const m2 = zw_ballot(ne2);
                u32 const a0 = curr + dd, fl = n - a0; u32 f0, k0 = 0;
                {   u32 const mm = (u32)m2 & 0x7FFFu;
                    if (mm) { u32 const j = (u32)__builtin_ctz(mm); f0 = 8u * j + ((u32)__builtin_ctzll(zw_get64(d2, j)) >> 3); if (f0 > fl) f0 = fl; }
                    else f0 = fl <= 120u ? fl : 120u + count_fwd(a0 + 120u, a0 - offN + 120u); }
                f0 = ZJ_UNI(f0);
                if (kd == 1u) {
                    mLength = 4u + f0; mip = curr + 1u;
                    store(anchor, mip - anchor, 1u, mLength);
                } else {
                    u32 const lim0 = zj_min(curr - anchor, mpos), mb = (u32)(m2 >> 16) & 0xFFu;
                    if (mb) { u32 const j = (u32)__builtin_ctz(mb); k0 = 8u * j + ((u32)__builtin_clzll(zw_get64(d2, 16u + j)) >> 3); if (k0 > lim0) k0 = lim0; }
                    else k0 = lim0 <= 64u ? lim0 : 64u + count_back(curr - 64u, mpos - 64u, lim0 - 64u);
                    k0 = ZJ_UNI(k0);
                    mip = curr - k0; mLength = dd + f0 + k0;
                    off2 = off1; off1 = offN;
                    if (step < 4u) { u32 const h1 = zw_get(hl, K + 1u), t1 = zw_get(tgL, K + 1u); ZW_LANES(l) { if (l == 0) HL[h1] = ZX_ENT(ip1 + 1u, t1); } }
                    store(anchor, mip - anchor, offN + 3u, mLength);
                }
                cvalid = true; cbase = curr; coffB = offN; coffC = offC;
                ZX_MARK(5);
            } else if (kd == 1u) {
                cvalid = false;
                u32 f0, k0, f1, k1;
                extend(curr + 5u, curr + 5u - off1, 8u, 8u, 0u, false, 0, 0, 8u, 8u, 0, f0, k0, f1, k1);
                mLength = 4u + f0; mip = curr + 1u;
                store(anchor, mip - anchor, 1u, mLength);
            } else {
                cvalid = false;
                u32 mpos, f0, k0, f1, k1;
                if (kd == 2u) {
                    mpos = zw_get(cL, K); mip = curr;
                    extend(curr + 8u, mpos + 8u, curr, mpos, zj_min(curr - anchor, mpos), false, 0, 0, 8u, 8u, 0, f0, k0, f1, k1);
                    mLength = 8u + f0;
                } else {
                    mpos = zw_get(cS, K); mip = curr;
                    u32 const mpos1 = two ? zw_get(cL, K + 1u) : 8u;
                    extend(curr + 4u, mpos + 4u, curr, mpos, zj_min(curr - anchor, mpos), two, ip1 + 8u, mpos1 + 8u, ip1, mpos1, zj_min(ip1 - anchor, mpos1), f0, k0, f1, k1);
                    mLength = 4u + f0;
                    if (two && 8u + f1 > mLength) { mip = ip1; mLength = 8u + f1; mpos = mpos1; k0 = k1; }
                }
                u32 const offset = mip - mpos;
                mip -= k0; mLength += k0;
                off2 = off1; off1 = offset;
                if (step < 4u) { u32 const h1 = zw_get(hl, K + 1u), t1 = zw_get(tgL, K + 1u); ZW_LANES(l) { if (l == 0) HL[h1] = ZX_ENT(ip1 + 1u, t1); } }
                store(anchor, mip - anchor, offset + 3u, mLength);
            }
            ip = ZJ_UNI(mip + mLength); anchor = ip; off1 = ZJ_UNI(off1); off2 = ZJ_UNI(off2);
            ZX_MARK(6);
            if (ip <= ilimit) {
                // complementary insertion — curr + 2 and ip - 2 (long), curr + 2 and ip - 1 (short), in this order — and the
                // immediate-repcode loop; the bytes of both are requested together.  All of these writes are lane 0's, in the
                // reference's order: two of them may name one bucket.
                for (bool first = true;; first = false) {
                    ZWV<u64> d, wi, wq; ZWV<bool> ne; u32 cov;
                    ip = ZJ_UNI(ip); off1 = ZJ_UNI(off1); off2 = ZJ_UNI(off2); o.n = ZJ_UNI(o.n); o.lit = ZJ_UNI(o.lit);
                    u32 const e = ip - cbase;
                    if (cvalid && e + 8u <= ZX_CARRY && (off2 == coffB || off2 == coffC || off2 == 0u)) {
                        const u8* const xs = (off2 == coffB) ? lds.stB : lds.stC;      // (off2 == 0: not compared)
                        u32 const nv = (ZX_CARRY - e) / 8u;                             // words inside the staged span (>= 1)
                        cov = 8u * nv;
                        ZW_LANES(l) {
                            bool const in = l < nv;
                            u32 const at = in ? e + 8u * l : 0u, q = l == 1 ? e - 2u : (l == 3 ? e - 1u : 2u);
                            u64 const ra = ld64(lds.stA + at), rbb = ld64(xs + at), rq = ld64(lds.stA + q);
                            ZW_FENCE2(ra, rbb); ZW_FENCE2(rq, rq);
                            wi[l] = ra; d[l] = in ? ra ^ rbb : 0ull; ne[l] = d[l] != 0; wq[l] = rq;
                        }
                    } else {
                        cov = 512u;
                        ZW_LANES(l) {
                            u32 const q = l == 1 ? ip - 2u : (l == 3 ? ip - 1u : curr + 2u);
                            u32 const p0 = ip + 8u * l, p1 = ip - off2 + 8u * l, q0_ = at(p0), q1_ = at(p1), q2_ = at(q);
                            u64 const r0_ = ld64(base + q0_), r1_ = ld64(base + q1_), r2_ = ld64(base + q2_);
                            ZW_FENCE2(r0_, r1_); ZW_FENCE2(r2_, r2_);
                            u64 const ra = fix(r0_, p0, q0_), rbb = fix(r1_, p1, q1_), rq = fix(r2_, q, q2_);
                            wi[l] = ra; d[l] = ra ^ rbb; ne[l] = d[l] != 0; wq[l] = rq;
                        }
                        ZX_STAT(stTrips++);
                    }
                    if (first) {
                        // lanes 0 / 1: long entries of curr + 2 and ip - 2; lanes 2 / 3: short entries of curr + 2 and ip - 1 — hashed side by side.
                        // Within a table the later insert wins a shared bucket: the earlier lane then does not store.
                        ZWV<u32> bk, en;
                        ZW_LANES(l) {
                            u64 const x = wq[l];                                                 // (lanes 0 / 2: curr + 2's word, lane 1: ip - 2's, lane 3: ip - 1's — see the loads)
                            u32 const p1 = l == 0 || l == 2 ? curr + 3u : (l == 1 ? ip - 1u : ip);
                            u32 bL, bS; u32 const eL_ = entL_of(x, p1, bL), eS_ = entS_of(x, p1, bS);
                            bk[l] = l < 2u ? bL : bS; en[l] = l < 2u ? eL_ : eS_;
                        }
                        bool const sameL = zw_get(bk, 0) == zw_get(bk, 1), sameS = zw_get(bk, 2) == zw_get(bk, 3);
                        ZW_LANES(l) {
                            if (l == 0 && !sameL) HL[bk[l]] = en[l];
                            if (l == 1) HL[bk[l]] = en[l];
                            if (l == 2 && !sameS) HS[bk[l]] = en[l];
                            if (l == 3) HS[bk[l]] = en[l];
                        }
                    }
                    if (off2 == 0u || (u32)zw_get64(d, 0) != 0u) break;
                    u64 const m = zw_ballot(ne);
                    u32 const lim = n - ip; u32 rLength;
                    if (m) { u32 const j = (u32)__builtin_ctzll(m); rLength = 8u * j + ((u32)__builtin_ctzll(zw_get64(d, j)) >> 3); if (rLength > lim) rLength = lim; }
                    else rLength = lim <= cov ? lim : cov + count_fwd(ip + cov, ip - off2 + cov);
                    rLength = ZJ_UNI(rLength);
                    { u32 const t = off2; off2 = off1; off1 = t; }
                    {   u64 const wi0 = zw_get64(wi, 0);
                        ZW_LANES(l) { if (l == 0) { u32 bs_, bl_; u32 const es_ = entS_of(wi0, ip + 1u, bs_), el_ = entL_of(wi0, ip + 1u, bl_); HS[bs_] = es_; HL[bl_] = el_; } } }
                    store(anchor, 0u, 1u, rLength);
                    ip += rLength; anchor = ip;
                    if (ip > ilimit) break;
                }
            }
            step = 1u; nextStep = ip + 256u;
            ZX_MARK(7);
        }
#if defined(ZX_PROFILE) && ZJ_ON_GPU
        if (blockIdx.x < 8u && threadIdx.x == 0 && o.n > 64u)
            printf("zx wg %u block@%u: %u seqs, %llu kcycles; cycles/seq: window %llu  storewait %llu  tables %llu  scoreboard %llu  candidates %llu  commit %llu  stage+count %llu  slowextend+store %llu  post %llu\n",
                   blockIdx.x, blkStart, o.n, (unsigned long long)((__builtin_readcyclecounter() - pStart) / 1000ull), (unsigned long long)(pf[0] / o.n), (unsigned long long)(pf[9] / o.n), (unsigned long long)(pf[1] / o.n), (unsigned long long)(pf[2] / o.n), (unsigned long long)(pf[3] / o.n), (unsigned long long)(pf[4] / o.n), (unsigned long long)(pf[5] / o.n), (unsigned long long)(pf[6] / o.n), (unsigned long long)(pf[7] / o.n));
#endif
        // zstd_double_fast.c:238-246: a parked offset comes back unless a new one took its place
        saved2 = (saved1 != 0u && off1 != 0u) ? saved1 : saved2;
        ZW_LANES(l) { if (l == 0) { repOut[0] = off1 ? off1 : saved1; repOut[1] = off2 ? off2 : saved2; } }
        ZX_STORES_DONE();                                                  // the records are read by other lanes next
        return n - anchor;
    }
};

// ---------------------------------------------------------------------------------------------
// The same for the fast strategy (levels 1-2): ZSTD_compressBlock_fast_noDict_generic (N/compress/zstd_fast.c:192-423), one table.
// The reference's loop takes two positions per iteration — ip0 = B and ip1 = B + 1 — inserts both, and BEFORE it tests their table
// candidates it tests the repcode at the next iteration's first position ip2 = B + gap (gap = the step when that position was set: 2,
// growing by one every 128 bytes without a match).  Order of the tests of iteration k: repcode at B(k+1), candidate of B(k), candidate
// of B(k) + 1; the first test that passes, iterations ascending, ends the loop.  A window holds K iterations: lane 2k = B(k), lane
// 2k + 1 = B(k) + 1, lane 2K = B(K) for the last repcode test.  The entry the reference finds at a position is the table's, or the
// position of the latest lower lane with its hash (every earlier position of the window is inserted before the position's lookup) —
// the scoreboard of the double-fast matcher; tags, presumed winner, staged spans and the order of the table writes as there.
struct ZWaveF : ZWaveX {
    ZJ_DEV_MEMBER u32 entF_of(u64 w, u32 pos1, u32& bucket) const { bucket = zl_hash(hL, w); return ZX_ENT(pos1, tagS_of(w)); }
    ZJ_DEV_MEMBER u32 run_fast(ZXLds& lds, const u8* frame, u32 frameSize, u32 blkStart, u32 blkEnd, u32 hBits, u32 mls,
                               u32* table, const u32* repIn, u32* repOut, bool carry) {
        carryOn = carry; cvalid = false; cbase = coffB = coffC = 0;
        L = &lds; base = frame; nf = frameSize; start = blkStart; n = blkEnd; ilimit = blkEnd - 8u; HL = table; HS = table;
        hL = zl_hash_of(mls, hBits); hS = hL;
        o.n = 0; o.lit = 0;
#ifdef ZX_STATS
        stPasses = stHitPasses = stTrips = stLanes = stSlow = 0;
#endif
        ZW_LANES(l) { for (u32 i = l; i < ZX_SB_SLOTS; i += 64u) lds.SL[i] = 0; }
        ZW_SYNC();
        u32 ip = blkStart + (blkStart == 0u ? 1u : 0u), anchor = blkStart;
        u32 rep1 = ZJ_UNI(repIn[0]), rep2 = ZJ_UNI(repIn[1]), saved1 = 0, saved2 = 0;
        {   u32 const maxRep = ip;                                          // zstd_fast.c:244-250
            if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
            if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; } }
        u32 step = 2, gap = 2, nextStep = ip + 128u, cap = ZX_CAP_MIN;    // cap: iterations per window
        if (blkEnd >= 12u) for (;;) {
            ip = ZJ_UNI(ip); anchor = ZJ_UNI(anchor); rep1 = ZJ_UNI(rep1); rep2 = ZJ_UNI(rep2); step = ZJ_UNI(step); gap = ZJ_UNI(gap); nextStep = ZJ_UNI(nextStep); cap = ZJ_UNI(cap);
            o.n = ZJ_UNI(o.n); o.lit = ZJ_UNI(o.lit);
            if (ip + gap + 1u >= ilimit) break;                           // ip3 >= ilimit: the iteration at ip does not run (loop header and do-while alike)
            // ---- window: K iterations with bases ip, ip + gap, ip + gap + step, ...; a pending step change (gap != step) runs alone
            u32 K = 1;
            if (gap == step) {
                u32 const q = nextStep > ip ? (nextStep - ip + step - 1u) / step : 0u;        // no step change inside: B(K) < nextStep for the checks of iterations 0 .. K - 2
                u32 const kStep = q > 2u ? q - 1u : 1u;
                u32 const kRoom = (ilimit - 2u - ip) / step;                                     // every iteration's ip3 = B + step + 1 < ilimit
                K = kStep < cap ? kStep : cap; if (kRoom < K) K = kRoom; if (K < 1u) K = 1u; if (K > 31u) K = 31u;
            }
            u32 const nS = 2u * K;                                        // search lanes; lane nS looks ahead (repcode test only)
            ZX_STAT(stPasses++); ZX_STAT(stLanes += nS);
            ZWV<u32> pos, hl, tg, eT, rb, pred; ZWV<u64> w, mM; ZWV<u32> cP; ZWV<bool> hitM, hitR;
            cvalid = ZJ_UNI(cvalid ? 1u : 0u) != 0u; cbase = ZJ_UNI(cbase); coffB = ZJ_UNI(coffB); coffC = ZJ_UNI(coffC);
            bool const consecutive = gap == 2u && step == 2u;            // lane l = position ip + l
            if (cvalid && consecutive && ip >= cbase && (ip - cbase) + nS + 8u <= ZX_CARRY && (rep1 == coffB || rep1 == coffC || rep1 == 0u)) {
                const u8* const rs = (rep1 == coffB) ? lds.stB : lds.stC;
                ZW_LANES(l) {
                    bool const act = l <= nS;
                    u32 const pp = act ? ip + l : ip; pos[l] = pp;
                    u64 const ww = ld64(lds.stA + (pp - cbase)); u32 const rr = ld32(rs + (pp - cbase));
                    ZW_FENCE2(ww, rr);
                    w[l] = ww; rb[l] = rr;
                    hl[l] = zl_hash(hL, ww); tg[l] = tagS_of(ww); pred[l] = 64u;
                }
            } else {
                ZW_LANES(l) {
                    bool const act = l <= nS; u32 const k = l >> 1;
                    u32 const b = k == 0u ? ip : ip + gap + (k - 1u) * step;
                    u32 const pp = act ? b + (l & 1u) : ip; pos[l] = pp;
                    ZX_LOAD2(ww, pp, rw, pp - rep1);
                    w[l] = ww; rb[l] = (u32)rw;
                    hl[l] = zl_hash(hL, ww); tg[l] = tagS_of(ww); pred[l] = 64u;
                }
                ZX_STAT(stTrips++);
            }
            // ---- the table as the previous windows left it, and who in this window comes before whom
            ZX_STORES_DONE();
            ZW_LANES(l) {
                bool const srch = l < nS;
                u32 const x = srch ? ZX_TLOAD(&HL[hl[l]]) : 0u;
                if (srch) zw_or64(&lds.SL[hl[l] & (ZX_SB_SLOTS - 1u)], 1ull << l);
                ZW_FENCE2(x, x);
                eT[l] = x;
            }
            ZX_STAT(stTrips++);
            ZW_SYNC();
            ZW_LANES(l) { mM[l] = (l < nS) ? (lds.SL[hl[l] & (ZX_SB_SLOTS - 1u)] & ((1ull << l) - 1ull)) : 0ull; }
            ZW_SYNC();
            ZW_LANES(l) { if (l < nS) lds.SL[hl[l] & (ZX_SB_SLOTS - 1u)] = 0; }
            ZW_SYNC();
            for (;;) {
                ZWV<u32> jM, gM; ZWV<bool> pend;
                ZW_LANES(l) { jM[l] = mM[l] ? 63u - (u32)__builtin_clzll(mM[l]) : l; pend[l] = mM[l] != 0; }
                if (!zw_ballot(pend)) break;
                zw_shfl(gM, hl, jM);
                ZW_LANES(l) { if (mM[l]) { if (gM[l] == hl[l]) { pred[l] = jM[l]; mM[l] = 0; } else mM[l] &= ~(1ull << jM[l]); } }
            }
            // ---- tests: the repcode ones are exact (even lanes from 2 on), the candidates' presumed from tags (a lower lane as entry: exact)
            u32 const tagsOn = (ZX_TAGS && carryOn) ? 1u : 0u;
            {   ZWV<u32> wlo, pwlo, iP; ZWV<bool> anyPred;
                ZW_LANES(l) { wlo[l] = (u32)w[l]; pwlo[l] = 0; iP[l] = pred[l] < 64u ? pred[l] : l; anyPred[l] = pred[l] < 64u; }
                if (zw_ballot(anyPred)) zw_shfl(pwlo, wlo, iP);
                ZW_LANES(l) {
                    bool const srch = l < nS; bool pM; u32 a;
                    if (pred[l] < 64u) { pM = pwlo[l] == wlo[l]; u32 const k = pred[l] >> 1; a = (k == 0u ? ip : ip + gap + (k - 1u) * step) + (pred[l] & 1u); }
                    else { u32 const e = eT[l]; a = ZX_POS(e) - 1u; pM = ZX_POS(e) > 1u && (!tagsOn || (e >> 21) == tg[l]); }
                    cP[l] = a; hitM[l] = srch && pM;
                    hitR[l] = l >= 2u && l <= nS && !(l & 1u) && rep1 > 0u && rb[l] == wlo[l];
                }
            }
            u64 hm = 0; u32 evK = 0, evT = 0;                              // winner: iteration, type (0 repcode at B(k + 1), 1 candidate of B(k), 2 candidate of B(k) + 1)
            bool staged = false, decided = false;
            for (u32 round = 0; !decided; round++) {
                // round 0: presumed (tags); round 1 (a presumption failed, or no tags): every candidate's bytes fetched and compared
                if (round == 1u || !tagsOn) {
                    ZW_LANES(l) {
                        bool const srch = l < nS;
                        u32 const a = cP[l]; bool const v = srch && (pred[l] < 64u || ZX_POS(eT[l]) > 1u);
                        u64 const c = fb(v ? a : 0u);
                        ZW_FENCE2(c, c);
                        hitM[l] = v && (u32)c == (u32)w[l];
                    }
                    ZX_STAT(stTrips++);
                }
                u64 const rM = zw_ballot(hitR), mAll = zw_ballot(hitM);
                u64 const m0 = mAll & 0x5555555555555555ull, m1 = mAll & 0xAAAAAAAAAAAAAAAAull;
                u32 eR = 0xFFFFFFFFu, e0 = 0xFFFFFFFFu, e1 = 0xFFFFFFFFu;
                if (rM) eR = 3u * (((u32)__builtin_ctzll(rM) >> 1) - 1u);
                if (m0) e0 = 3u * ((u32)__builtin_ctzll(m0) >> 1) + 1u;
                if (m1) e1 = 3u * ((u32)__builtin_ctzll(m1) >> 1) + 2u;
                u32 const ev = zj_min(eR, zj_min(e0, e1));
                if (ev == 0xFFFFFFFFu) { hm = 0; decided = true; break; }
                evK = ev / 3u; evT = ev % 3u; hm = 1;
                if (round == 1u || !tagsOn || evT == 0u) { decided = true; break; }
                // a presumed candidate hit: stage the match's spans and check the four bytes there
                {   u32 const lane = 2u * evK + (evT == 2u ? 1u : 0u), hp = zw_get(pos, lane), mp = zw_get(cP, lane);
                    cvalid = false;
                    stage(lds, hp, mp, rep1);
                    if (ZJ_UNI((u32)(ld32(lds.stA) == ld32(lds.stB))) != 0u) { staged = true; decided = true; }
                }
            }
            u32 const cnt = hm ? 2u * (evK + 1u) : nS;                     // lanes whose inserts happen: both positions of every iteration up to the winner's
            // ---- commit
            {   ZWV<bool> hasPred;
                ZW_LANES(l) { hasPred[l] = l < cnt && pred[l] < 64u; }
                if (zw_ballot(hasPred)) {
                    ZW_LANES(l) { lds.shadowL[l] = 0; }
                    ZW_SYNC();
                    ZW_LANES(l) { if (l < cnt && pred[l] < 64u) lds.shadowL[pred[l]] = 1; }
                    ZW_SYNC();
                    ZW_LANES(l) { if (l < cnt && !lds.shadowL[l]) HL[hl[l]] = ZX_ENT(pos[l] + 1u, tg[l]); }
                    ZW_SYNC();
                } else {
                    ZW_LANES(l) { if (l < cnt) HL[hl[l]] = ZX_ENT(pos[l] + 1u, tg[l]); }
                }
            }
            if (!hm) {                                                       // no test passed: the end of the last iteration, as the reference's loop does it
                u32 const bK = K == 0u ? ip : ip + gap + (K - 1u) * step;
                ip = bK; gap = step;
                if (ip + step >= nextStep) { step++; nextStep += 128u; }
                cap = cap * 2u < 31u ? cap * 2u : 31u;
                continue;
            }
            ZX_STAT(stHitPasses++);
            {   u32 const c2 = 2u * (evK + 1u); cap = c2 < ZX_CAP_MIN ? ZX_CAP_MIN : (c2 < 31u ? c2 : 31u); }
            // ---- the match, as the reference handles it
            u32 const bk = evK == 0u ? ip : ip + gap + (evK - 1u) * step;   // B(k)
            u32 const gk = evK == 0u ? gap : step;                          // B(k + 1) - B(k)
            u32 const cur0 = evT == 2u ? bk + 1u : bk;
            u32 const hlane = evT == 0u ? 2u * evK + 2u : 2u * evK + (evT == 2u ? 1u : 0u);
            u32 const hp = evT == 0u ? bk + gk : cur0;                      // where the match starts before it is extended backwards
            u32 const mpos = evT == 0u ? hp - rep1 : zw_get(cP, hlane);
            u32 const offC = evT == 0u ? rep2 : rep1;                       // the offset the repcode loop behind the match will test
            u32 f0, k0;
            if (carryOn) {
                if (!staged) stage(lds, hp, mpos, offC);
                ZWV<u64> d2; ZWV<bool> ne2;
                ZW_LANES(l) {
                    u64 x = 0;
                    if (l < 15u) x = ld64(lds.stA + 4u + 8u * l) ^ ld64(lds.stB + 4u + 8u * l);
                    else if (l >= 16u && l < 24u) x = ld64(lds.stKA + 56u - 8u * (l - 16u)) ^ ld64(lds.stKB + 56u - 8u * (l - 16u));
                    d2[l] = x; ne2[l] = x != 0;
                }
                ZW_SYNC();
                u64 const m2 = zw_ballot(ne2);
                u32 const a0 = hp + 4u, fl = n - a0, offN = hp - mpos;
                {   u32 const mm = (u32)m2 & 0x7FFFu;
                    if (mm) { u32 const j = (u32)__builtin_ctz(mm); f0 = 8u * j + ((u32)__builtin_ctzll(zw_get64(d2, j)) >> 3); if (f0 > fl) f0 = fl; }
                    else f0 = fl <= 120u ? fl : 120u + count_fwd(a0 + 120u, a0 - offN + 120u); }
                u32 const lim0 = evT == 0u ? 1u : zj_min(hp - anchor, mpos), mb = (u32)(m2 >> 16) & 0xFFu;
                if (mb) { u32 const j = (u32)__builtin_ctz(mb); k0 = 8u * j + ((u32)__builtin_clzll(zw_get64(d2, 16u + j)) >> 3); if (k0 > lim0) k0 = lim0; }
                else k0 = lim0 <= 64u ? lim0 : 64u + count_back(hp - 64u, mpos - 64u, lim0 - 64u);
                cvalid = true; cbase = hp; coffB = offN; coffC = offC;
            } else {
                u32 f1, k1;
                extend(hp + 4u, mpos + 4u, hp, mpos, evT == 0u ? 1u : zj_min(hp - anchor, mpos), false, 0, 0, 8u, 8u, 0, f0, k0, f1, k1);
                cvalid = false;
            }
            f0 = ZJ_UNI(f0); k0 = ZJ_UNI(k0);
            u32 const mip = hp - k0, mLength = 4u + f0 + k0;
            if (evT == 0u) store(anchor, mip - anchor, 1u, mLength);         // (the repcode's one byte backwards is unconditional in the reference: ip2 - 1 >= ip1 > anchor)
            else {
                u32 const offset = hp - mpos;
                rep2 = rep1; rep1 = offset;
                if (evT == 2u && step <= 4u) { u32 const h1 = zw_get(hl, 2u * evK + 2u), t1 = zw_get(tg, 2u * evK + 2u), p1 = bk + gk; ZW_LANES(l) { if (l == 0) HL[h1] = ZX_ENT(p1 + 1u, t1); } }
                store(anchor, mip - anchor, offset + 3u, mLength);
            }
            ip = ZJ_UNI(mip + mLength); anchor = ip; rep1 = ZJ_UNI(rep1); rep2 = ZJ_UNI(rep2);
            if (ip <= ilimit) {
                // complementary insertion (cur0 + 2, then ip - 2) and the immediate-repcode loop; lane 0 writes, in the reference's order
                for (bool first = true;; first = false) {
                    ip = ZJ_UNI(ip); rep1 = ZJ_UNI(rep1); rep2 = ZJ_UNI(rep2); o.n = ZJ_UNI(o.n); o.lit = ZJ_UNI(o.lit);
                    ZWV<u64> d, wi, wq; ZWV<bool> ne; u32 cov;
                    u32 const e = ip - cbase;
                    if (cvalid && e + 8u <= ZX_CARRY && cur0 + 2u >= cbase && (rep2 == coffB || rep2 == coffC || rep2 == 0u)) {
                        const u8* const xs = (rep2 == coffB) ? lds.stB : lds.stC;
                        u32 const nv = (ZX_CARRY - e) / 8u;
                        cov = 8u * nv;
                        ZW_LANES(l) {
                            bool const in = l < nv;
                            u32 const at = in ? e + 8u * l : 0u, q = l == 1 ? e - 2u : cur0 + 2u - cbase;
                            u64 const ra = ld64(lds.stA + at), rbb = ld64(xs + at), rq = ld64(lds.stA + q);
                            ZW_FENCE2(ra, rbb); ZW_FENCE2(rq, rq);
                            wi[l] = ra; d[l] = in ? ra ^ rbb : 0ull; ne[l] = d[l] != 0; wq[l] = rq;
                        }
                    } else {
                        cov = 512u;
                        ZW_LANES(l) {
                            u32 const q = l == 1 ? ip - 2u : cur0 + 2u;
                            u32 const p0 = ip + 8u * l, p1 = ip - rep2 + 8u * l, q0_ = at(p0), q1_ = at(p1), q2_ = at(q);
                            u64 const r0_ = ld64(base + q0_), r1_ = ld64(base + q1_), r2_ = ld64(base + q2_);
                            ZW_FENCE2(r0_, r1_); ZW_FENCE2(r2_, r2_);
                            u64 const ra = fix(r0_, p0, q0_), rbb = fix(r1_, p1, q1_), rq = fix(r2_, q, q2_);
                            wi[l] = ra; d[l] = ra ^ rbb; ne[l] = d[l] != 0; wq[l] = rq;
                        }
                        ZX_STAT(stTrips++);
                    }
                    if (first) {
                        u64 const q0 = zw_get64(wq, 0), q1 = zw_get64(wq, 1);
                        ZW_LANES(l) { if (l == 0) { u32 b0_, b1_; u32 const e0_ = entF_of(q0, cur0 + 3u, b0_), e1_ = entF_of(q1, ip - 1u, b1_); HL[b0_] = e0_; HL[b1_] = e1_; } }
                    }
                    if (rep2 == 0u || (u32)zw_get64(d, 0) != 0u) break;
                    u64 const m = zw_ballot(ne);
                    u32 const lim = n - ip; u32 rLength;
                    if (m) { u32 const j = (u32)__builtin_ctzll(m); rLength = 8u * j + ((u32)__builtin_ctzll(zw_get64(d, j)) >> 3); if (rLength > lim) rLength = lim; }
                    else rLength = lim <= cov ? lim : cov + count_fwd(ip + cov, ip - rep2 + cov);
                    rLength = ZJ_UNI(rLength);
                    { u32 const t = rep2; rep2 = rep1; rep1 = t; }
                    {   u64 const wi0 = zw_get64(wi, 0);
                        ZW_LANES(l) { if (l == 0) { u32 b_; u32 const e_ = entF_of(wi0, ip + 1u, b_); HL[b_] = e_; } } }
                    store(anchor, 0u, 1u, rLength);
                    ip += rLength; anchor = ip;
                    if (ip > ilimit) break;
                }
            }
            step = 2u; gap = 2u; nextStep = ip + 128u;
        }
        saved2 = (saved1 != 0u && rep1 != 0u) ? saved1 : saved2;         // zstd_fast.c:352-372
        ZW_LANES(l) { if (l == 0) { repOut[0] = rep1 ? rep1 : saved1; repOut[1] = rep2 ? rep2 : saved2; } }
        ZX_STORES_DONE();
        return n - anchor;
    }
};
ZJ_DEV u32 zx_block_fast_wave(u8* lds, ZEOut& o, const u8* base, u32 frameSize, u32 start, u32 end, u32 hBits, u32 mls,
                              u32* table, const u32* repIn, u32* repOut, bool carry) {
    ZWaveF m; m.o = o;
    frameSize = ZJ_UNI(frameSize); start = ZJ_UNI(start); end = ZJ_UNI(end); hBits = ZJ_UNI(hBits); mls = ZJ_UNI(mls);
    carry = ZJ_UNI(carry ? 1u : 0u) != 0u;
    u32 const lastLL = m.run_fast(*(ZXLds*)lds, base, frameSize, start, end, hBits, mls, table, repIn, repOut, carry);
    o = m.o;
    return lastLL;
}

// One block of a multi-block frame through the wave matcher (the signature ze_compress_t's block path calls; declared in zj_encode.h).
ZJ_DEV u32 zx_block_dfast_wave(u8* lds, ZEOut& o, const u8* base, u32 frameSize, u32 start, u32 end, u32 hBitsL, u32 hBitsS, u32 mls,
                               u32* hashLong, u32* hashSmall, const u32* repIn, u32* repOut, bool carry) {
    ZWaveX m; m.o = o;
    // every argument is the same in all lanes, but some reach here through memory (the block arguments): said once, so that the parse
    // keeps them in scalar registers and branches on them with scalar branches (the pointers are left alone: an integer round trip would
    // cost them their address space, and every load would become a flat load)
    frameSize = ZJ_UNI(frameSize); start = ZJ_UNI(start); end = ZJ_UNI(end); hBitsL = ZJ_UNI(hBitsL); hBitsS = ZJ_UNI(hBitsS); mls = ZJ_UNI(mls);
    carry = ZJ_UNI(carry ? 1u : 0u) != 0u;
    u32 const lastLL = m.run(*(ZXLds*)lds, base, frameSize, start, end, hBitsL, hBitsS, mls, hashLong, hashSmall, repIn, repOut, carry);
    o = m.o;
    return lastLL;
}
