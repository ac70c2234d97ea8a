// zj_match_wave.h — wave-per-frame double-fast match finder (level 3), hash tables in LDS.
//
// One wavefront owns one frame (<= 64 KiB) and runs the reference's sequential parse
// (ZSTD_compressBlock_doubleFast_noDict_generic, N/compress/zstd_double_fast.c:105-323) 64 positions at a time:
// a WINDOW is the next <= 63 positions the reference's inner loop would visit if none of them matched (stride =
// the current step), one per lane, plus one look-ahead lane.  Every lane hashes its position, probes both tables
// (u16 position+1 entries in LDS: 32 KiB long + 16 KiB short at hashLog 14 / chainLog 13), fetches its repcode
// and table candidates and tests them; the lowest lane with a hit is the position where the reference's loop would
// have stopped.  The table inserts of the lanes up to it are committed, the match is extended by the whole wave
// (64 x 8 bytes per step, forwards and backwards), the complementary inserts and the immediate-repcode loop follow,
// and the next window starts behind the match.  The decisions, their order and every table write are the
// reference's, so the sequences are identical to ze_block_dfast's (tests/test_emu_wave.py, tests/test_gpu_encode.py).
//
// What a lane reads from a table is "the entry as the reference would find it": the reference inserts every visited
// position before it moves on, so lane k's entry is the position of the latest lane j < k of the same window with
// the same hash, or else the table's content.  That predecessor is found through a scoreboard in LDS: every lane
// ORs its bit into slot (hash & 255), reads the slot back, and walks the few lower lanes named there (shuffle of
// their hashes) — slots are shared by different hashes, equal hashes always share a slot.  Inserts are committed
// in one LDS write per table by the lanes nobody in the committed range shadows.
//
// The frame itself is staged into LDS as well (one coalesced copy), so every read of the parse — window bytes,
// candidates, match extension — is an LDS access (~75 cycles) instead of an L2 / Infinity-Cache round trip (300 to
// 1 200 cycles, profiles/r02a_probe_*): 116 KiB per wave, one wave per CU, beside the lane-per-frame kernel's waves.
//
// The kernel body is written in explicit-SIMT form — per-lane variables are ZWV<T>, per-lane code sits in
// ZW_LANES blocks, lanes talk to each other only BETWEEN blocks (ballot / shuffle / LDS) — so the same source
// compiles for the GPU (one thread per lane) and, with a 64-iteration loop per block, under g++ for the CPU-side
// parity tests (tests/emu; test infrastructure only).
#pragma once

#include "zj_simt.h"
#define ZW_HASHLOG_MAX 14u
#define ZW_CHAINLOG_MAX 13u
#define ZW_SB_SLOTS 256u
struct ZWLds {
    u16 HL[1u << ZW_HASHLOG_MAX]; u16 HS[1u << ZW_CHAINLOG_MAX];
    u64 SL[ZW_SB_SLOTS]; u64 SS[ZW_SB_SLOTS];       // scoreboards: lanes of the current window per hash slot (windows with colliding hashes only)
    u8 shadowL[64]; u8 shadowS[64];                  // commit: lane is shadowed by a later committed lane with its hash
    // The frame's bytes (staged once, coalesced): every match-finder read is an LDS read.  The match-extension reads are
    // not clamped to the frame: they may run up to 264 bytes past its end (into the padding) or before its start (into the
    // arrays above) — bytes that can only lengthen a count beyond its limit, and every count is cut to its limit.
#ifndef ZW_FRAME_IN_LDS
#define ZW_FRAME_IN_LDS 1
#endif
#if ZW_FRAME_IN_LDS
    u8 frame[65536u + 544u];
#else
    u8 frame[16];                                    // (experiment: frame bytes from global memory, three waves per CU)
#endif
};
// true when a frame's level-3 parameters fit the LDS tables of this matcher
ZJ_HD bool zw_takes(const ZEParams& p, u32 srcSize) { return p.strategy == 2 && p.hashLog <= ZW_HASHLOG_MAX && p.chainLog <= ZW_CHAINLOG_MAX && srcSize <= 65536u && srcSize >= 64u; }

struct ZWaveD {
    ZWLds* L; const u8* src; u32 n, ilimit; ZLHash hL, hS; ZEOut o;
#ifdef ZW_STATS
    u64 stPasses, stHitPasses, stRepIters, stLoadTrips, stSlow;
#define ZW_STAT(x) (x)
#else
#define ZW_STAT(x) ((void)0)
#endif
#if defined(ZW_PROFILE) && ZJ_ON_GPU      /* tools/micro/wavebench: cycles per phase of a pass */
    u64 pf[16]; u64 pT;
#define ZW_MARK(i) do { u64 const t_ = __builtin_readcyclecounter(); pf[i] += t_ - pT; pT = t_; } while (0)
#define ZW_CNT(i) (pf[i]++)
#else
#define ZW_MARK(i) ((void)0)
#define ZW_CNT(i) ((void)0)
#endif
    // frame bytes [pos, pos + 8); pos may lie up to 520 bytes outside the frame on either side (see ZWLds::frame)
#if ZW_FRAME_IN_LDS
    ZJ_DEV_MEMBER u64 fb(u32 pos) const { return ld64(src + (ptrdiff_t)(i32)pos); }
#else
    ZJ_DEV_MEMBER u64 fb(u32 pos) const {            // never touches memory outside the frame; bytes outside read as zero
        if ((i32)pos < 0) { u32 const k = 0u - pos; u64 const v = ld64(src); return k >= 8u ? 0 : (v << (8u * k)); }
        u32 const q = zl_fwd_at(n, pos); return zl_fwd_fix(ld64(src + q), pos, q);
    }
#endif

    // length of the common prefix of src[a..] and src[b..] (b < a), at most n - a: ZSTD_count(a, b, iend)
    ZJ_DEV_MEMBER u32 count_fwd(u32 a, u32 b) {
        u32 const lim = n - a;
        for (u32 total = 0;; total += 512u) {
            ZWV<u64> d; ZWV<bool> ne;
            ZW_LANES(l) {
                u32 const pa = a + total + 8u * l, pb = b + total + 8u * l;
                u32 const qa = zl_fwd_at(n, pa), qb = zl_fwd_at(n, pb);
                u64 const ra = ld64(src + qa), rb = ld64(src + qb);
                d[l] = zl_fwd_fix(ra, pa, qa) ^ zl_fwd_fix(rb, pb, qb); ne[l] = d[l] != 0;
            }
            ZW_STAT(stLoadTrips++);
            u64 const m = zw_ballot(ne);
            if (m) { u32 const j = (u32)__builtin_ctzll(m); u32 const c = total + 8u * j + ((u32)__builtin_ctzll(zw_get64(d, j)) >> 3); return c < lim ? c : lim; }
            if (total + 512u >= lim) return lim;
        }
    }
    // number of equal bytes going backwards from (ipos - 1, mpos - 1), at most limit (<= mpos < ipos)
    ZJ_DEV_MEMBER u32 count_back(u32 ipos, u32 mpos, u32 limit) {
        if (limit == 0) return 0;
        for (u32 total = 0;; total += 512u) {
            ZWV<u64> d; ZWV<bool> ne;
            ZW_LANES(l) {
                u32 const back = total + 8u * l;
                u32 const pa = ipos > back ? ipos - back : 0u, pb = mpos > back ? mpos - back : 0u;
                u32 const qa = zl_back_at(pa), qb = zl_back_at(pb);
                u64 const ra = ld64(src + qa), rb = ld64(src + qb);
                d[l] = zl_back_fix(ra, pa, qa) ^ zl_back_fix(rb, pb, qb); ne[l] = d[l] != 0;
            }
            ZW_STAT(stLoadTrips++);
            u64 const m = zw_ballot(ne);
            if (m) { u32 const j = (u32)__builtin_ctzll(m); u32 const c = total + 8u * j + ((u32)__builtin_clzll(zw_get64(d, j)) >> 3); return c < limit ? c : limit; }
            if (total + 512u >= limit) return limit;
        }
    }
    // Both directions of up to two candidate matches in ONE round of LDS reads: lanes 0-15 count forwards from (a0, b0),
    // lanes 16-31 backwards from (i0, m0), lanes 32-47 / 48-63 the same for the second candidate (two = false: idle).
    // 128 bytes per direction; a count that runs through all of them continues in the general loops above.
    ZJ_DEV_MEMBER void extend(u32 a0, u32 b0, u32 i0, u32 m0, u32 lim0, bool two, u32 a1, u32 b1, u32 i1, u32 m1, u32 lim1,
                              u32& f0, u32& k0, u32& f1, u32& k1) {
        ZWV<u64> d; ZWV<bool> ne;
        ZW_LANES(l) {
            u32 const j = l & 15u, q = l >> 4;
            u32 pa, pb;
            if (q == 0u) { pa = a0 + 8u * j; pb = b0 + 8u * j; } else if (q == 1u) { pa = i0 - 8u - 8u * j; pb = m0 - 8u - 8u * j; }
            else if (q == 2u) { pa = a1 + 8u * j; pb = b1 + 8u * j; } else { pa = i1 - 8u - 8u * j; pb = m1 - 8u - 8u * j; }
            bool const on = two || q < 2u;
            u64 const xa = fb(on ? pa : 8u), xb = fb(on ? pb : 8u);
            ZW_FENCE2(xa, xb);
            u64 const x = xa ^ xb;
            d[l] = x; ne[l] = x != 0;
        }
        ZW_STAT(stLoadTrips++);
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
    }
    ZJ_DEV_MEMBER void store(u32 litPos, u32 ll, u32 offBase, u32 ml) {
        ZW_LANES(l) { if (l == 0) { ZESeq s; s.ll = ll; s.ml = ml; s.off = offBase; s.pos = litPos; o.litOff[o.n] = o.lit; o.seqs[o.n] = s; } }
        o.n++; o.lit += ll;
    }

    // the whole block; returns the length of the last literal run
    ZJ_DEV_MEMBER u32 run(ZWLds& lds, const u8* s, u32 size, const ZEParams& p, u8* fscratch, u32 maxSrc) {
        L = &lds; src = ZW_FRAME_IN_LDS ? lds.frame : s; n = size; ilimit = size - 8u; hL = zl_hash_of(8, p.hashLog); hS = zl_hash_of(p.minMatch, p.chainLog);
        o.seqs = (ZESeq*)fscratch; o.litOff = (u32*)(fscratch + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); o.n = 0; o.lit = 0;
#ifdef ZW_STATS
        stPasses = stHitPasses = stRepIters = stLoadTrips = stSlow = 0;
#endif
#if defined(ZW_PROFILE) && ZJ_ON_GPU
        for (int j = 0; j < 16; j++) pf[j] = 0;
        pT = __builtin_readcyclecounter(); u64 const pStart = pT;
#endif
        {   // stage the frame, empty the tables and scoreboards
            u32 const words = ((1u << p.hashLog) * 2u) / 8u, wordsS = ((1u << p.chainLog) * 2u) / 8u;
            ZW_LANES(l) {
#if ZW_FRAME_IN_LDS
                for (u32 i = l * 16u; i + 16u <= size; i += 1024u) { u64 const x = ld64(s + i), y = ld64(s + i + 8u); st64(lds.frame + i, x); st64(lds.frame + i + 8u, y); }
                for (u32 i = (size & ~15u) + l; i < size; i += 64u) lds.frame[i] = s[i];
#endif
                u64* const a = (u64*)lds.HL; for (u32 i = l; i < words; i += 64u) a[i] = 0;
                u64* const b = (u64*)lds.HS; for (u32 i = l; i < wordsS; i += 64u) b[i] = 0;
                for (u32 i = l; i < ZW_SB_SLOTS; i += 64u) { lds.SL[i] = 0; lds.SS[i] = 0; }
            }
            ZW_SYNC();
        }
        ZW_MARK(9);
        u32 ip = 1, anchor = 0, off1 = 1, off2 = 0, step = 1, nextStep = 1u + 256u;
        for (;;) {
            ip = ZJ_UNI(ip); anchor = ZJ_UNI(anchor); off1 = ZJ_UNI(off1); off2 = ZJ_UNI(off2); step = ZJ_UNI(step); nextStep = ZJ_UNI(nextStep);
            o.n = ZJ_UNI(o.n); o.lit = ZJ_UNI(o.lit);                       // wave-uniform by construction; this tells the compiler (scalar registers, scalar branches)
            if (ip + step > ilimit) break;                                // ip1 > ilimit: _cleanup
            // ---- window: lanes 0..nIter-1 are the reference's next iterations (ip = p, ip1 = p + step), lane nIter looks ahead
            u32 kmax, room;
            if (step == 1u) { kmax = nextStep > ip ? nextStep - ip : 1u; room = ilimit - ip; }
            else { kmax = nextStep > ip ? (nextStep - ip + step - 1u) / step : 1u; if (kmax < 1u) kmax = 1u; room = (ilimit - ip) / step; }
            u32 nIter = kmax < 63u ? kmax : 63u; if (room < nIter) nIter = room;
            ZW_STAT(stPasses++); ZW_CNT(12);
            ZWV<u32> pos, hl, hs, oldL, oldS, rb, predL, predS; ZWV<u64> w; ZWV<bool> lostL, lostS, lost;
            ZWV<u32> cL, cS, kind; ZWV<bool> hit, longHit;
            ZW_LANES(l) {
                bool const act = l <= nIter, srch = l < nIter;
                u32 const pp = act ? ip + l * step : ip; pos[l] = pp;
                u64 const ww = fb(pp); u32 const rr = (u32)fb(pp + 1u - off1);
                ZW_FENCE2(ww, rr);
                w[l] = ww; rb[l] = rr;
                u32 const a = zl_hash(hL, ww), b = zl_hash(hS, ww); hl[l] = a; hs[l] = b;
                u32 const eL = lds.HL[a], eS = lds.HS[b];
                ZW_FENCE2(eL, eS);
                oldL[l] = srch ? eL : 0u; oldS[l] = srch ? eS : 0u;
                predL[l] = 64u; predS[l] = 64u;
            }
            ZW_STAT(stLoadTrips++);
            ZW_SYNC();
            // ---- the window's inserts, tentatively: the table now reads as after the last of these iterations
            ZW_LANES(l) { if (l < nIter) { lds.HL[hl[l]] = (u16)(pos[l] + 1u); lds.HS[hs[l]] = (u16)(pos[l] + 1u); } }
            ZW_SYNC();
            ZW_MARK(0);
            // ---- candidates from the table as it was, their bytes, the three tests of an iteration (repcode, long, short);
            //      a lane that does not read back its own insert shares a hash with another lane of the window
            ZW_LANES(l) {
                bool const act = l <= nIter, srch = l < nIter;
                u32 const backL = lds.HL[hl[l]], backS = lds.HS[hs[l]];
                u32 const eS = oldS[l], a0 = oldL[l] - 1u, b = eS - 1u;
                u64 const cl0 = fb(oldL[l] > 1u ? a0 : 0u);                 // (the look-ahead lane's candidate comes with its read-back, below)
                u32 const cs = (u32)fb(eS > 1u ? b : 0u);
                ZW_FENCE4(backL, backS, cl0, cs);
                lostL[l] = srch && backL != ((pos[l] + 1u) & 0xFFFFu); lostS[l] = srch && backS != ((pos[l] + 1u) & 0xFFFFu);
                lost[l] = lostL[l] || lostS[l];
                u64 cl = cl0; u32 a = a0;
                if (l == nIter) { oldL[l] = backL; a = backL - 1u; cl = fb(backL > 1u ? a : 0u); }   // it reads after the inserts, as the reference does
                bool const vL = act && oldL[l] > 1u, vS = srch && eS > 1u;
                cL[l] = a; cS[l] = b;
                bool const hR = srch && off1 > 0u && rb[l] == (u32)(w[l] >> 8);
                bool const hLg = vL && cl == w[l], hSh = vS && cs == (u32)w[l];
                longHit[l] = hLg;
                kind[l] = hR ? 1u : (hLg ? 2u : 3u);
                hit[l] = srch && (hR || hLg || hSh);
            }
            ZW_STAT(stLoadTrips++);
            u64 hm = zw_ballot(hit);
            u32 cnt = hm ? (u32)__builtin_ctzll(hm) + 1u : nIter;            // lanes whose inserts happen
            u64 const lm = zw_ballot(lost);
            ZW_MARK(1);
            // The tests above are exact for a lane unless an EARLIER lane of the window has its hash (then that lane's position
            // is the entry the reference would read).  Any such pair leaves at least one of its lanes "lost", so: no lost lane up
            // to the one after the winner (whose long-table entry the short-match path consults) = every decision so far is exact.
            if ((lm & (cnt >= 63u ? ~0ull : ((2ull << cnt) - 1ull))) == 0) {
                // undo the inserts of the iterations that do not happen; one lane per table slot writes (the one that owns it now)
                if (hm) { ZW_LANES(l) { if (l >= cnt && l < nIter) { if (!lostL[l]) lds.HL[hl[l]] = (u16)oldL[l]; if (!lostS[l]) lds.HS[hs[l]] = (u16)oldS[l]; } } ZW_SYNC(); }
            } else {
                // ---- general case: back to the table as it was, then every lane's latest earlier lane with the same hash
                ZW_STAT(stSlow++); ZW_CNT(11);
                ZW_LANES(l) { if (l < nIter) { if (!lostL[l]) lds.HL[hl[l]] = (u16)oldL[l]; if (!lostS[l]) lds.HS[hs[l]] = (u16)oldS[l]; } }
                ZW_SYNC();
                ZW_LANES(l) {
                    if (l == nIter) oldL[l] = lds.HL[hl[l]];
                    if (l < nIter) { zw_or64(&lds.SL[hl[l] & (ZW_SB_SLOTS - 1u)], 1ull << l); zw_or64(&lds.SS[hs[l] & (ZW_SB_SLOTS - 1u)], 1ull << l); }
                }
                ZW_SYNC();
                ZWV<u64> mL, mS;
                ZW_LANES(l) {
                    u64 const below = (1ull << l) - 1ull;
                    mL[l] = (l <= nIter) ? (lds.SL[hl[l] & (ZW_SB_SLOTS - 1u)] & below) : 0ull;
                    mS[l] = (l < nIter) ? (lds.SS[hs[l] & (ZW_SB_SLOTS - 1u)] & below) : 0ull;
                }
                ZW_SYNC();
                ZW_LANES(l) { if (l < nIter) { lds.SL[hl[l] & (ZW_SB_SLOTS - 1u)] = 0; lds.SS[hs[l] & (ZW_SB_SLOTS - 1u)] = 0; } }
                for (;;) {
                    ZWV<u32> jL, jS, gL, gS; ZWV<bool> pend;
                    ZW_LANES(l) {
                        jL[l] = mL[l] ? 63u - (u32)__builtin_clzll(mL[l]) : l; jS[l] = mS[l] ? 63u - (u32)__builtin_clzll(mS[l]) : l;
                        pend[l] = (mL[l] | mS[l]) != 0;
                    }
                    if (!zw_ballot(pend)) break;
                    zw_shfl(gL, hl, jL); zw_shfl(gS, hs, jS);
                    ZW_LANES(l) {
                        if (mL[l]) { if (gL[l] == hl[l]) { predL[l] = jL[l]; mL[l] = 0; } else mL[l] &= ~(1ull << jL[l]); }
                        if (mS[l]) { if (gS[l] == hs[l]) { predS[l] = jS[l]; mS[l] = 0; } else mS[l] &= ~(1ull << jS[l]); }
                    }
                }
                ZW_LANES(l) {
                    bool const act = l <= nIter, srch = l < nIter;
                    bool const vL = act && (predL[l] < 64u || oldL[l] > 1u), vS = srch && (predS[l] < 64u || oldS[l] > 1u);
                    u32 const a = predL[l] < 64u ? ip + predL[l] * step : oldL[l] - 1u, b = predS[l] < 64u ? ip + predS[l] * step : oldS[l] - 1u;
                    cL[l] = a; cS[l] = b;
                    u64 const cl = vL ? fb(a) : ~w[l];
                    u32 const cs = vS ? (u32)fb(b) : ~(u32)w[l];
                    bool const hR = srch && off1 > 0u && rb[l] == (u32)(w[l] >> 8);
                    bool const hLg = vL && cl == w[l], hSh = vS && cs == (u32)w[l];
                    longHit[l] = hLg;
                    kind[l] = hR ? 1u : (hLg ? 2u : 3u);
                    hit[l] = srch && (hR || hLg || hSh);
                }
                ZW_STAT(stLoadTrips++);
                hm = zw_ballot(hit);
                cnt = hm ? (u32)__builtin_ctzll(hm) + 1u : nIter;
                // commit: HL[hl] = HS[hs] = position + 1 for lanes < cnt, the last lane of a hash wins
                ZW_LANES(l) { lds.shadowL[l] = 0; lds.shadowS[l] = 0; }
                ZW_SYNC();
                ZW_LANES(l) { if (l < cnt) { if (predL[l] < 64u) lds.shadowL[predL[l]] = 1; if (predS[l] < 64u) lds.shadowS[predS[l]] = 1; } }
                ZW_SYNC();
                ZW_LANES(l) { if (l < cnt) { if (!lds.shadowL[l]) lds.HL[hl[l]] = (u16)(pos[l] + 1u); if (!lds.shadowS[l]) lds.HS[hs[l]] = (u16)(pos[l] + 1u); } }
                ZW_SYNC();
            }
            ZW_MARK(2);
            if (!hm) {                                                       // nobody matched: the loop's own bookkeeping
                u32 const pN = ip + nIter * step;
                if (pN >= nextStep) { step++; nextStep += 256u; }
                ip = pN;
                continue;
            }
            ZW_STAT(stHitPasses++); ZW_CNT(13);
            // ---- the match at lane K, as the reference handles it
            u32 const K = cnt - 1u, curr = ip + K * step, ip1 = curr + step, kd = zw_get(kind, K);
            u32 mip, mLength;
            if (kd == 1u) {
                u32 f0, k0, f1, k1;
                extend(curr + 5u, curr + 5u - off1, 8u, 8u, 0u, false, 0, 0, 8u, 8u, 0, f0, k0, f1, k1);
                mLength = 4u + f0; mip = curr + 1u;
                ZW_MARK(3);
                store(anchor, mip - anchor, 1u, mLength);
            } else {
                u32 mpos, f0, k0, f1, k1;
                if (kd == 2u) {
                    mpos = zw_get(cL, K); mip = curr;
                    extend(curr + 8u, mpos + 8u, curr, mpos, zj_min(curr - anchor, mpos), false, 0, 0, 8u, 8u, 0, f0, k0, f1, k1);
                    mLength = 8u + f0;
                } else {
                    mpos = zw_get(cS, K); mip = curr;
                    bool const two = zw_getb(longHit, K + 1u);               // _search_next_long: the long candidate of ip1
                    u32 const mpos1 = two ? zw_get(cL, K + 1u) : 8u;
                    extend(curr + 4u, mpos + 4u, curr, mpos, zj_min(curr - anchor, mpos), two, ip1 + 8u, mpos1 + 8u, ip1, mpos1, zj_min(ip1 - anchor, mpos1), f0, k0, f1, k1);
                    mLength = 4u + f0;
                    if (two && 8u + f1 > mLength) { mip = ip1; mLength = 8u + f1; mpos = mpos1; k0 = k1; }
                }
                u32 const offset = mip - mpos;
                mip -= k0; mLength += k0;
                ZW_MARK(3);
                off2 = off1; off1 = offset;
                if (step < 4u) { u32 const h1 = zw_get(hl, K + 1u); ZW_LANES(l) { if (l == 0) lds.HL[h1] = (u16)(ip1 + 1u); } }
                store(anchor, mip - anchor, offset + 3u, mLength);
            }
            ip = mip + mLength; anchor = ip;
            ZW_MARK(4);
            if (ip <= ilimit) {
                // complementary insertion — curr + 2 and ip - 2 (long), curr + 2 and ip - 1 (short), in this order — and the
                // immediate-repcode loop; the bytes of both are requested together
                for (bool first = true;; first = false) {
                    ZWV<u64> d, wi, wq; ZWV<bool> ne;
                    ZW_LANES(l) {
                        u32 const q = l == 0 ? curr + 2u : (l == 1 ? ip - 2u : ip - 1u);
                        u64 const ra = fb(ip + 8u * l), rbb = fb(ip - off2 + 8u * l), rq = fb(q);
                        ZW_FENCE2(ra, rbb); ZW_FENCE2(rq, rq);
                        wi[l] = ra; d[l] = ra ^ rbb; ne[l] = d[l] != 0; wq[l] = rq;
                    }
                    ZW_STAT(stLoadTrips++); ZW_STAT(stRepIters++); ZW_CNT(14);
                    if (first) {
                        ZW_LANES(l) { if (l == 0) { lds.HL[zl_hash(hL, wq[l])] = (u16)(curr + 3u); lds.HS[zl_hash(hS, wq[l])] = (u16)(curr + 3u); } }
                        ZW_SYNC();
                        ZW_LANES(l) { if (l == 1) lds.HL[zl_hash(hL, wq[l])] = (u16)(ip - 1u); if (l == 2) lds.HS[zl_hash(hS, wq[l])] = (u16)ip; }
                        ZW_SYNC();
                    }
                    if (off2 == 0u || (u32)zw_get64(d, 0) != 0u) break;
                    u64 const m = zw_ballot(ne);
                    u32 const lim = n - ip; u32 rLength;
                    if (m) { u32 const j = (u32)__builtin_ctzll(m); rLength = 8u * j + ((u32)__builtin_ctzll(zw_get64(d, j)) >> 3); if (rLength > lim) rLength = lim; }
                    else rLength = lim <= 512u ? lim : 512u + count_fwd(ip + 512u, ip - off2 + 512u);
                    { u32 const t = off2; off2 = off1; off1 = t; }
                    ZW_LANES(l) { if (l == 0) { lds.HS[zl_hash(hS, wi[l])] = (u16)(ip + 1u); lds.HL[zl_hash(hL, wi[l])] = (u16)(ip + 1u); } }
                    ZW_SYNC();
                    store(anchor, 0u, 1u, rLength);
                    ip += rLength; anchor = ip;
                    if (ip > ilimit) break;
                }
            }
            ZW_MARK(5);
            step = 1u; nextStep = ip + 256u;
        }
#if defined(ZW_PROFILE) && ZJ_ON_GPU
        pf[15] = __builtin_readcyclecounter() - pStart;
#endif
        return n - anchor;
    }
};

// One frame through the wave matcher: records + meta {nbSeq, litSize, lastLL}, the layout ze_match_lane writes.
ZJ_DEV void zw_match_frame(ZWLds& lds, const u8* src, u32 srcSize, u32 level, u8* fscratch, u32 maxSrc, u32* meta) {
    ZWaveD m;
    u32 const lastLL = m.run(lds, src, srcSize, ze_params_of(level, srcSize), fscratch, maxSrc);
    ZW_LANES(l) { if (l == 0) { meta[0] = m.o.n; meta[1] = m.o.lit + lastLL; meta[2] = lastLL; } }
}
