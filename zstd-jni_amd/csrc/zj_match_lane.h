// zj_match_lane.h — lane-per-frame match finders as round-synchronous state machines.
//
// One lane owns one frame and runs the reference's sequential parse (ZSTD_compressBlock_fast /
// _doubleFast, N/compress/zstd_fast.c:192-423, N/compress/zstd_double_fast.c:105-323); 64 frames advance
// per wavefront.  Written as straight loops (ze_block_fast/ze_block_dfast in zj_encode.h) a wavefront pays
// for SIMT divergence: as soon as one lane finds a match, all 64 lanes wait through that lane's chain of
// dependent memory round trips (count forward, extend backward, re-insert, repcode loop, restart), and with
// 64 lanes some lane almost always has a match (measured: 5.2 us per search step on a single idle wave,
// 6-8 dependent HBM/L2 round trips of ~0.6 us each; tools/micro/chase.hip).
//
// Here every lane is a small state machine and the wavefront executes ROUNDS: each round every lane names
// the few addresses its current state needs, all loads are issued together at one program point, and after
// one wait every lane consumes its data, does its table/record stores and moves to its next state.  A lane
// that is counting a match length and a lane that is probing the next position share the same round trip.
// The decisions, their order, and every table write are exactly the reference's — the machine only changes
// when bytes are fetched — so frames stay byte-identical (tests/test_emu_encode.py, tests/test_gpu_encode.py).
#pragma once

// bytes [pos, pos+8) of a frame of n >= 8 bytes; bytes at or beyond n read as zero (never touches memory past n)
ZJ_DEV u64 zl_ld_fwd(const u8* s, u32 n, u32 pos) {
    u32 const p = zj_min(pos, n - 8u);
    u64 const v = ld64(s + p);
    u32 const sh = (pos - p) * 8u;
    return sh >= 64u ? 0 : (v >> sh);
}
// bytes [pos-8, pos): byte pos-1 is the most significant; bytes before the frame start read as zero
ZJ_DEV u64 zl_ld_back(const u8* s, u32 pos) {
    u32 const p = pos >= 8u ? pos - 8u : 0u;
    u64 const v = ld64(s + p);
    u32 const sh = (8u - (pos - p)) * 8u;
    return sh >= 64u ? 0 : (v << sh);
}
// The same two loads split into "where to read" and "fix up what was read", so that a round can issue every
// lane's loads back to back, unconditionally (slots a lane does not need read offset 0), and wait once.
ZJ_DEV u32 zl_fwd_at(u32 n, u32 pos) { return zj_min(pos, n - 8u); }
ZJ_DEV u64 zl_fwd_fix(u64 v, u32 pos, u32 at) { u32 const sh = (pos - at) * 8u; return sh >= 64u ? 0 : (v >> sh); }
ZJ_DEV u32 zl_back_at(u32 pos) { return pos >= 8u ? pos - 8u : 0u; }
ZJ_DEV u64 zl_back_fix(u64 v, u32 pos, u32 at) { u32 const sh = (8u - (pos - at)) * 8u; return sh >= 64u ? 0 : (v << sh); }
// All loads of a round are consumed here, at one program point: without it the compiler sinks each load into
// the state branch that uses it and the round degenerates into one memory round trip per state again.
#if ZJ_ON_GPU
#define ZL_ROUND_FENCE9(a, b, c, d, e, f, g, h, i) asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h), "v"(i))
#define ZL_ROUND_FENCE2(a, b) asm volatile("" :: "v"(a), "v"(b))
#else
#define ZL_ROUND_FENCE9(a, b, c, d, e, f, g, h, i) ((void)0)
#define ZL_ROUND_FENCE2(a, b) ((void)0)
#endif
ZJ_DEV u32 zl_common_fwd16(u64 a0, u64 a1, u64 b0, u64 b1) {
    u64 const x0 = a0 ^ b0, x1 = a1 ^ b1;
    if (x0) return (u32)__builtin_ctzll(x0) >> 3;
    return 8u + (x1 ? ((u32)__builtin_ctzll(x1) >> 3) : 8u);
}
ZJ_DEV u32 zl_common_back8(u64 p, u64 q) { u64 const x = p ^ q; return x ? ((u32)__builtin_clzll(x) >> 3) : 8u; }

#ifdef ZL_PROFILE
#define ZL_PROF_MEMBERS u64 pT0 = 0, pT1 = 0, pA = 0, pB = 0, pC = 0, pR = 0;
#define ZL_PROF_T0() do { u64 const t_ = __builtin_readcyclecounter(); if (pT0) pC += t_ - pT1; pT0 = t_; pR++; } while (0)
#define ZL_PROF_T1() do { u64 const t_ = __builtin_readcyclecounter(); pA += t_ - pT0; pT1 = t_; } while (0)
#define ZL_PROF_T2() do { u64 const t_ = __builtin_readcyclecounter(); pB += t_ - pT1; pT1 = t_; } while (0)
#else
#define ZL_PROF_MEMBERS
#define ZL_PROF_T0() ((void)0)
#define ZL_PROF_T1() ((void)0)
#define ZL_PROF_T2() ((void)0)
#endif
enum { ZL_LOADW = 0, ZL_START, ZL_SEARCH, ZL_COUNT, ZL_BACK, ZL_POST, ZL_DONE };
// ZSTD_hashPtr without the switch on minMatch: every variant is ((bytes << s) * prime) >> (64 - hBits) with a
// per-frame (s, prime) — minMatch 4 is the 64-bit form of its 32-bit product — and hBits <= 17 only needs the
// high dword of the product: three 32-bit multiplies, no branches (N/compress/zstd_compress_internal.h:898-960).
struct ZLHash { u32 sh, plo, phi, rsh; };
ZJ_DEV ZLHash zl_hash_of(u32 mls, u32 hBits) {
    ZLHash h; u64 p;
    if (mls == 5) { h.sh = 24; p = 889523592379ULL; } else if (mls == 6) { h.sh = 16; p = 227718039650203ULL; }
    else if (mls == 7) { h.sh = 8; p = 58295818150454627ULL; } else if (mls == 8) { h.sh = 0; p = 0xCF1BBCDCB7A56463ULL; }
    else { h.sh = 32; p = 2654435761ULL; }
    h.plo = (u32)p; h.phi = (u32)(p >> 32); h.rsh = 32u - hBits;
    return h;
}
ZJ_DEV u32 zl_mulhi(u32 a, u32 b) { return (u32)(((u64)a * b) >> 32); }
ZJ_DEV u32 zl_prod_hi(const ZLHash& h, u64 w) {            // high dword of ((w << sh) * prime)
    u64 const x = w << h.sh; u32 const xlo = (u32)x, xhi = (u32)(x >> 32);
    return zl_mulhi(xlo, h.plo) + xlo * h.phi + xhi * h.plo;
}
ZJ_DEV u32 zl_hash(const ZLHash& h, u64 w) { return zl_prod_hi(h, w) >> h.rsh; }
#ifndef ZL_DFAST_PERIOD
#define ZL_DFAST_PERIOD 8u
#endif
enum { ZL_EN_COUNT = 1, ZL_EN_POST = 2, ZL_EN_START = 4 };   // which non-search states a round serves
enum { ZC_REP1 = 0, ZC_LONG, ZC_SHORT, ZC_SHORT_L1, ZC_REPLOOP, ZC_FOUND };

// ---------------------------------------------------------------------------------------------
// double-fast (level 3).  Positions are offsets from the frame start.
// GATED (zj_need.h): a flag byte per position says which table accesses can matter — ZN_NEED_L / ZN_NEED_S: some other position of the frame
// carries the key this probe would match on (no such position: whatever the entry holds, the byte comparison fails — the entry is not read
// and counts as empty); ZN_INS_L / ZN_INS_S: some position of the frame will read this position's bucket (none: the entry is not written).
// A bucket that is read at all receives every write, so a probe that is made sees exactly the entry the reference's probe sees; a probe
// that is skipped could not have matched.  Same decisions, a fraction of the table requests.
// (Large level-3 batches run zj_match_run.h's machine — register windows, runs of flag-quiet positions per round; this one serves the
// wide launch, and ZJNI_LANE_MACHINE=0 keeps it selectable for A/B runs.)
template <class E, bool GATED = false>
struct ZLaneD {
    typedef typename E::T Ent;
    const u8* src; u32 n, ilimit; Ent* HL; Ent* HS; ZLHash hL, hS;
    const u8* F; u32 fI, fN;                          // GATED: the frame's flag bytes; flags of ip and of ip1
    ZEOut o;
    u32 st, cont, lastLL;
    u32 ip, ip1, anchor, off1, off2, step, nextStep, curr;
    u64 w, w1; u32 el0, es0, el1, hl0, hs0, hl1, hs1, tl0, tl1;   // tl: long-table tag of w / w1
    u32 ca, cb, acc;                                  // forward count in progress
    u32 mpos, mpos2, mLength, offset, bk, bk2;       // chosen match / candidate at ip1 / backward extension
    bool more, more2, cvalid, needBack, needCand, chk, haveIns;
    u64 wIns;                                         // the word at curr + 2 (first post-insert) when the search round already holds it


    ZJ_DEV_MEMBER void init(const u8* s, u32 size, const ZEParams& p, u8* table, u8* fscratch, u32 maxSrc, const u8* flags = nullptr) {
        src = s; n = size; ilimit = size - 8u; hL = zl_hash_of(8, p.hashLog); hS = zl_hash_of(p.minMatch, p.chainLog);
        HL = (Ent*)table; HS = HL + (1u << p.hashLog);
        F = flags; fI = 15u; fN = 15u;
        o.seqs = (ZESeq*)fscratch; o.litOff = (u32*)(fscratch + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); o.n = 0; o.lit = 0;
        ip = 1; anchor = 0; off1 = 1; off2 = 0; chk = false; lastLL = size;
        needBack = needCand = more = more2 = cvalid = haveIns = false;
        st = ZL_LOADW;
    }
    // long-table index and tag from ONE product: index = top hashLog bits, tag = the next 15 bits (hashLog <= 17)
    // (returned by value: references to members keep part of the machine on the stack)
    ZJ_DEV_MEMBER u32 prod_long(u64 v) const { return zl_prod_hi(hL, v); }
    ZJ_DEV_MEMBER u32 idx_long(u32 p) const { return p >> hL.rsh; }
    ZJ_DEV_MEMBER u32 tag_long(u32 p) const { return (p >> (hL.rsh - 15u)) & 0x7FFFu; }
    ZJ_DEV_MEMBER void put_long(u64 v, u32 pos1) { u32 const p = prod_long(v); HL[idx_long(p)] = E::make(pos1, tag_long(p)); }
    ZJ_DEV_MEMBER void put_long_if(bool on, u64 v, u32 pos1) { if (!GATED || on) put_long(v, pos1); }
    ZJ_DEV_MEMBER void put_short_if(bool on, u64 v, u32 pos1) { if (!GATED || on) HS[zl_hash(hS, v)] = E::make(pos1, ze_tag4((u32)v)); }
    ZJ_DEV_MEMBER void finish() { lastLL = n - anchor; st = ZL_DONE; }
    // outer-loop header of the reference: reset the step and make sure one more position fits
    ZJ_DEV_MEMBER void outer() {
        step = 1; nextStep = ip + 256u; ip1 = ip + 1u;
        if (ip1 > ilimit) finish(); else st = ZL_START;
    }
    ZJ_DEV_MEMBER void begin_count(u32 a, u32 b, u32 c) { ca = a; cb = b; acc = 0; cont = c; st = ZL_COUNT; }
    ZJ_DEV_MEMBER void advance() { ip += mLength; anchor = ip; if (ip <= ilimit) st = ZL_POST; else finish(); }
    ZJ_DEV_MEMBER void fin() {             // a long/short match is final: apply the backward extension, store it
        ip -= bk; mLength += bk;
        off2 = off1; off1 = offset;
        if (step < 4u && (!GATED || (fN & 4u))) HL[hl1] = E::make(ip1 + 1u, tl1);      // (hl1 / fN: the position that was ip1 when the match was found)
        ze_store(o, anchor, ip - anchor, offset + 3u, mLength);
        advance();
    }
    ZJ_DEV_MEMBER void fin_or_back() { if (more) st = ZL_BACK; else fin(); }

    ZJ_DEVM u32 phase_of(u32 r) { return r % ZL_DFAST_PERIOD; }
    ZJ_DEVM u32 default_period() { return ZL_DFAST_PERIOD; }
    ZJ_DEVM bool takes_flags_late() { return false; }
    ZJ_DEV_MEMBER void take_flags(const u8*) {}
    ZL_PROF_MEMBERS
    // Round r of the wavefront.  A searching lane advances every round; the other states take turns (r mod 8:
    // count/backward, post-insert/reload, restart in consecutive rounds, then five search-only rounds), so a
    // round executes the search code plus at most one other state's code instead of all of them (rounds are
    // instruction-issue bound: a wave runs alone on its SIMD), and a lane that follows the natural order
    // search -> count -> post -> restart -> search meets its slots back to back.  The idle part of the
    // period is deliberate: at full batch the kernel is bound by HBM request rate while every lane is active
    // and by latency once only the frames with the most positions are left; pacing the match-dense frames
    // (which need 3x fewer rounds) leaves request slots to the search-dense ones and evens out the finish
    // times (measured: period 4 -> 8 is -6 % kernel time at 65 536 frames).
    // `r` is the round's phase within the period (0 .. period-1); the period is the caller's choice (ZL_DFAST_PERIOD when
    // one match kernel parses the whole batch; shorter when the search-dense frames go to the wave-per-frame kernel)
    ZJ_DEV_MEMBER void round(u32 r) {
        switch (r) {
        case 0: round_t<ZL_EN_COUNT>(); break;
        case 1: round_t<ZL_EN_POST>(); break;
        case 2: round_t<ZL_EN_START>(); break;
        default: round_t<0>(); break;
        }
    }
    template <int K>
    ZJ_DEV_MEMBER void round_t() {
        ZL_PROF_T0();
        u32 pa0 = 0, pa1 = 0, pa2 = 0, pa3 = 0, pa4 = 0, bp0 = 0, bp1 = 0, ti0 = 0, ti1 = 0;
        bool v0 = false, v1 = false, v2 = false, v3 = false, v4 = false, vb = false, vt = false;
        bool ml0 = false, ms0 = false; u32 ip2 = 0; u64 hw = 0;
        bool const on = (st == ZL_SEARCH) || ((K & ZL_EN_COUNT) && (st == ZL_COUNT || st == ZL_BACK))
                     || ((K & ZL_EN_POST) && (st == ZL_POST || st == ZL_LOADW)) || ((K & ZL_EN_START) && st == ZL_START);
        // ---- phase 1: what does this lane's state need (and the table writes that precede its reads) ----
        if (st == ZL_SEARCH) {
            ZE_COUNT_ITER();
            curr = ip;
            u32 const tl = tl0, ts = ze_tag4((u32)w);
            if (!GATED || (fI & 4u)) HL[hl0] = E::make(curr + 1u, tl);
            if (!GATED || (fI & 8u)) HS[hs0] = E::make(curr + 1u, ts);
            ml0 = E::maybe(el0, tl); ms0 = E::maybe(es0, ts);
            pa0 = ip + 1u - off1; v0 = true;
            pa1 = E::pos(el0) - 1u; v1 = ml0;
            pa2 = E::pos(es0) - 1u; v2 = ms0;
            ip2 = ip1 + step + ((ip1 >= nextStep) ? 1u : 0u);
            pa3 = ip2; v3 = ip2 <= ilimit;
            hw = w1; vt = true;
        } else if ((K & ZL_EN_COUNT) && st == ZL_COUNT) {
            pa0 = ca; pa1 = ca + 8u; pa2 = cb; pa3 = cb + 8u; v0 = v1 = v2 = v3 = true;
            if (needBack) { vb = true; if (cont == ZC_SHORT_L1) { bp0 = ip1; bp1 = mpos2; } else { bp0 = ip; bp1 = mpos; } }
            if (needCand) { v4 = true; pa4 = mpos2; }
        } else if ((K & ZL_EN_COUNT) && st == ZL_BACK) {
            vb = true; bp0 = ip - bk; bp1 = mpos - bk;
        } else if ((K & ZL_EN_POST) && st == ZL_POST) {
            pa0 = curr + 2u; pa1 = ip - 2u; pa2 = ip + 6u; v0 = !haveIns; v1 = v2 = true;
            pa3 = ip - off2; v3 = off2 > 0u;
        } else if ((K & ZL_EN_POST) && st == ZL_LOADW) {
            pa0 = ip; pa1 = ip + 1u; v0 = v1 = true;
            pa3 = ip - off2; v3 = chk && off2 > 0u;
        } else if ((K & ZL_EN_START) && st == ZL_START) {
            hw = w; vt = true;
        }
        // the hashes of the word whose table entries this round reads (search: next position; restart: this one) —
        // computed once here for both states, assigned to the state's own fields after the loads
        u32 const hp = prod_long(hw), nhl = idx_long(hp), ntl = tag_long(hp), nhs = zl_hash(hS, hw);
        ti0 = nhl; ti1 = nhs;
        // ---- phase 2: one batch of loads for all states ----
        ZL_PROF_T1();
        pa0 = v0 ? pa0 : 0; pa1 = v1 ? pa1 : 0; pa2 = v2 ? pa2 : 0; pa3 = v3 ? pa3 : 0; pa4 = v4 ? pa4 : 0;
        if (!vb) { bp0 = 8; bp1 = 8; }
        if (!vt) { ti0 = 0; ti1 = 0; }
        u32 const q0 = zl_fwd_at(n, pa0), q1 = zl_fwd_at(n, pa1), q2 = zl_fwd_at(n, pa2), q3 = zl_fwd_at(n, pa3), q4 = zl_fwd_at(n, pa4);
        u32 const qb0 = zl_back_at(bp0), qb1 = zl_back_at(bp1);
        // predicated: a slot nobody asked for costs no transaction (the fence below keeps the loads together)
        u64 r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, rb0 = 0, rb1 = 0; u32 t0 = 0, t1 = 0;
        u32 fa = 0, fb = 0;                              // GATED: flag bytes fetched this round (the area has 8 bytes of slack behind the frame's)
        u32 const fw = (st == ZL_SEARCH) ? fN : fI;      // whose probes this round issues: the next position's (search) / this one's (restart)
        if (v0) r0 = ld64(src + q0);
        if (on) {
            r3 = ld64(src + q3);
            if (!GATED || (fw & 1u)) t0 = (u32)HL[ti0];
            if (!GATED || (fw & 2u)) t1 = (u32)HS[ti1];
            if (GATED) {
                fa = 15u; fb = 0x0F0F0F0Fu;                                                // a frame without flags (F == nullptr) runs ungated
                if (F) {
                    if (st == ZL_SEARCH) fa = F[q3];                                   // flags of ip2 (q3 == ip2 whenever it is a position the search can reach)
                    else if ((K & ZL_EN_POST) && st == ZL_POST) { fa = F[curr + 2u]; fb = ld32(F + ip - 2u); }
                    else if ((K & ZL_EN_POST) && st == ZL_LOADW) fb = ld16(F + ip) << 16;
                }
            }
        }
        if (v1) r1 = ld64(src + q1);
        if (v2) r2 = ld64(src + q2);
        if ((K & ZL_EN_COUNT) && v4) r4 = ld64(src + q4);
        if ((K & ZL_EN_COUNT) && vb) { rb0 = ld64(src + qb0); rb1 = ld64(src + qb1); }
        ZL_ROUND_FENCE9(r0, r1, r2, r3, r4, rb0, rb1, t0, t1);
        if (GATED) ZL_ROUND_FENCE2(fa, fb);
        ZL_PROF_T2();
        u64 const d0 = zl_fwd_fix(r0, pa0, q0), d1 = zl_fwd_fix(r1, pa1, q1), d2 = zl_fwd_fix(r2, pa2, q2), d3 = zl_fwd_fix(r3, pa3, q3);
        u64 const d4 = zl_fwd_fix(r4, pa4, q4);
        u64 const b0 = zl_back_fix(rb0, bp0, qb0), b1 = zl_back_fix(rb1, bp1, qb1);
        if (!on) return;
        // ---- phase 3: consume ----
        if (st == ZL_SEARCH) {
            u32 const rv = (u32)d0; u32 es1 = t1; el1 = t0; hl1 = nhl; hs1 = nhs; tl1 = ntl;
            wIns = d3; haveIns = v3 && (ip2 == ip + 2u);        // curr + 2 == ip2: its bytes arrived with this round
            if ((off1 > 0u) & (rv == (u32)(w >> 8))) {
                begin_count(ip + 5u, ip + 5u - off1, ZC_REP1); needBack = false; needCand = false;
            } else if (ml0 && d1 == w) {
                mpos = E::pos(el0) - 1u;
                begin_count(ip + 8u, mpos + 8u, ZC_LONG); needBack = true; needCand = false;
            } else if (ms0 && (u32)d2 == (u32)w) {
                mpos = E::pos(es0) - 1u;
                begin_count(ip + 4u, mpos + 4u, ZC_SHORT); needBack = true;
                needCand = (E::pos(el1) > 1u) && E::maybe(el1, tl1); mpos2 = E::pos(el1) - 1u; cvalid = false;
            } else {
                bool const inc = ip1 >= nextStep;
                if (inc) { step++; nextStep += 256u; }
                ip = ip1; ip1 = ip2; w = w1; w1 = d3; el0 = el1; es0 = es1; hl0 = hl1; hs0 = hs1; tl0 = tl1;
                if (GATED) { fI = fN; fN = fa & 15u; }
                if (ip1 > ilimit) finish();
            }
        } else if ((K & ZL_EN_COUNT) && st == ZL_COUNT) {
            u32 const lim = n - ca;
            u32 c = zl_common_fwd16(d0, d1, d2, d3);
            if (c > lim) c = lim;
            acc += c;
            if (needBack) {
                bool const second = (cont == ZC_SHORT_L1);
                u32 const limit = second ? zj_min(ip1 - anchor, mpos2) : zj_min(ip - anchor, mpos);
                u32 e = zl_common_back8(b0, b1); if (e > limit) e = limit;
                bool const m = (e == 8u) && (limit > 8u);
                if (second) { bk2 = e; more2 = m; } else { bk = e; more = m; }
                needBack = false;
            }
            if (needCand) { cvalid = (d4 == w1); needCand = false; }
            if (c == 16u && lim > 16u) { ca += 16u; cb += 16u; }
            else if (cont == ZC_REP1) {
                mLength = acc + 4u; ip += 1u;
                ze_store(o, anchor, ip - anchor, 1u, mLength);
                advance();
            } else if (cont == ZC_LONG) {
                mLength = acc + 8u; offset = ip - mpos; fin_or_back();
            } else if (cont == ZC_SHORT) {
                mLength = acc + 4u; offset = ip - mpos;
                if (cvalid) { begin_count(ip1 + 8u, mpos2 + 8u, ZC_SHORT_L1); needBack = true; }
                else fin_or_back();
            } else if (cont == ZC_SHORT_L1) {
                u32 const l1len = acc + 8u;
                if (l1len > mLength) { ip = ip1; mLength = l1len; mpos = mpos2; offset = ip - mpos; bk = bk2; more = more2; }
                fin_or_back();
            } else {                                   // ZC_REPLOOP: immediate repcode after a match
                u32 const rLength = acc + 4u;
                { u32 const t = off2; off2 = off1; off1 = t; }
                put_short_if((fI & 8u) != 0, w, ip + 1u);
                put_long_if((fI & 4u) != 0, w, ip + 1u);
                ze_store(o, anchor, 0u, 1u, rLength);
                ip += rLength; anchor = ip;
                if (ip <= ilimit) { chk = true; st = ZL_LOADW; } else finish();
            }
        } else if ((K & ZL_EN_POST) && st == ZL_POST) {
            u64 const wa = haveIns ? wIns : d0, q0 = d1, q1 = d2;
            u64 const wb = q0, wc = (q0 >> 8) | (q1 << 56);
            u32 const ins = curr + 2u;
            // GATED: fa = flags of curr + 2, fb = flags of ip - 2, ip - 1, ip, ip + 1 (one byte each)
            put_long_if((fa & 4u) != 0, wa, ins + 1u);
            put_long_if((fb & 4u) != 0, wb, ip - 2u + 1u);
            put_short_if((fa & 8u) != 0, wa, ins + 1u);
            put_short_if((fb & 0x800u) != 0, wc, ip - 1u + 1u);
            if (GATED) { fI = (fb >> 16) & 15u; fN = (fb >> 24) & 15u; }
            w = (q0 >> 16) | (q1 << 48); w1 = (q0 >> 24) | (q1 << 40);
            if ((off2 > 0u) && ((u32)w == (u32)d3)) { begin_count(ip + 4u, ip + 4u - off2, ZC_REPLOOP); needBack = false; needCand = false; }
            else outer();
        } else if ((K & ZL_EN_START) && st == ZL_START) {
            el0 = t0; es0 = t1; hl0 = nhl; hs0 = nhs; tl0 = ntl; st = ZL_SEARCH;
        } else if ((K & ZL_EN_COUNT) && st == ZL_BACK) {
            u32 const limit = zj_min(ip - anchor, mpos) - bk;
            u32 e = zl_common_back8(b0, b1); if (e > limit) e = limit;
            bk += e; more = (e == 8u) && (limit > 8u);
            if (!more) fin();
        } else if ((K & ZL_EN_POST) && st == ZL_LOADW) {
            w = d0; w1 = d1;
            if (GATED) { fI = (fb >> 16) & 15u; fN = (fb >> 24) & 15u; }
            if (chk && (off2 > 0u) && ((u32)w == (u32)d3)) { begin_count(ip + 4u, ip + 4u - off2, ZC_REPLOOP); needBack = false; needCand = false; }
            else outer();
            chk = false;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// fast (levels 1-2), one round per position pair.  The reference's loop body handles the pair (ip0, ip1) and needs, in
// dependency order, the pair's bytes, its two table entries, and the candidates' bytes; the machine keeps that as a
// three-deep pipeline: the round that decides pair k fetches the candidates of pair k (entries already known), the table
// entries of pair k+1 (its bytes already known) and the bytes of pair k+2 (positions follow from the step rule).  Table
// reads issued a round early see exactly the writes the reference's read would see: the pair's own two writes are done
// before the loads, and the one write that falls in between (ip2, when ip3's entry shares its bucket) is forwarded.
template <class E>
struct ZLaneF {
    typedef typename E::T Ent;
    const u8* src; u32 n, ilimit; Ent* T; ZLHash hT;
    ZEOut o;
    u32 st, cont, lastLL;
    u32 ip0, ip1, ip2, ip3, anchor, rep1, rep2, step, nextStep, cur0, period;
    u64 w0, w1, w2, w3, wIns; u32 eX, eY;                 // eX / eY: table entries of ip0 / ip1 as the reference reads them
    u32 ca, cb, acc, mpos, mLength, offcode, bk;
    bool more, needBack, chk, haveIns;               // wIns/haveIns: the word at cur0 + 2 (first post-insert) when the search round already holds it

    ZJ_DEV_MEMBER void init(const u8* s, u32 size, const ZEParams& p, u8* table, u8* fscratch, u32 maxSrc, const u8* = nullptr) {
        src = s; n = size; ilimit = size - 8u; hT = zl_hash_of(p.minMatch, p.hashLog); T = (Ent*)table;
        o.seqs = (ZESeq*)fscratch; o.litOff = (u32*)(fscratch + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); o.n = 0; o.lit = 0;
        ip0 = 1; anchor = 0; rep1 = 1; rep2 = 0; chk = false; lastLL = size; needBack = more = haveIns = false; bk = 0;
        period = p.hashLog >= 15u ? 4u : 5u;          // measured: level 1 (hashLog 13) 84.0 ms at 5 / 85.3 at 4 / 99.0 at 3; level 2 (hashLog 15) 137.3 / 132.1 / 139.2
        st = ZL_LOADW;
    }
    ZJ_DEV_MEMBER void finish() { lastLL = n - anchor; st = ZL_DONE; }
    ZJ_DEV_MEMBER void outer() {
        step = 2; nextStep = ip0 + 128u; ip1 = ip0 + 1u; ip2 = ip0 + 2u; ip3 = ip0 + 3u;
        if (ip3 >= ilimit) finish(); else st = ZL_START;
    }
    ZJ_DEV_MEMBER void begin_count(u32 a, u32 b, u32 c) { ca = a; cb = b; acc = 0; cont = c; st = ZL_COUNT; }
    ZJ_DEV_MEMBER void fin() {
        ip0 -= bk; mLength += bk;
        ze_store(o, anchor, ip0 - anchor, offcode, mLength);
        ip0 += mLength; anchor = ip0;
        if (ip0 <= ilimit) st = ZL_POST; else finish();
    }
    // a table candidate matched at ip0 (entry e): new offset, then extend both ways
    ZJ_DEV_MEMBER void found_at(u32 e) {
        mpos = E::pos(e) - 1u;
        rep2 = rep1; rep1 = ip0 - mpos; offcode = rep1 + 3u; mLength = 4u;
        begin_count(ip0 + 4u, mpos + 4u, ZC_FOUND); needBack = true; bk = 0; more = false;
    }

    ZJ_DEVM u32 phase_of(u32 r) { return r; }
    ZJ_DEVM u32 default_period() { return ZL_DFAST_PERIOD; }      // (unused: the fast machine rotates on the round number itself)
    ZJ_DEVM bool takes_flags_late() { return false; }
    ZJ_DEV_MEMBER void take_flags(const u8*) {}
    ZL_PROF_MEMBERS
    // The search state runs every round; count/backward, post-insert/reload and restart take turns (r mod period = 0, 1, 2;
    // see ZLaneD::round).  With one round per pair the kernel sits at ~80 % of the read+write request plateau
    // (tools/micro/chase sweep), so the period only trades match-handling latency against code per round.
    ZJ_DEV_MEMBER void round(u32 r) {
        switch (period == 4u ? (r & 3u) : (r % 5u)) {           // constant divisors: a run-time modulo would cost a division per round
        case 0: round_t<ZL_EN_COUNT>(); break;
        case 1: round_t<ZL_EN_POST>(); break;
        case 2: round_t<ZL_EN_START>(); break;
        default: round_t<0>(); break;
        }
    }
    template <int K>
    ZJ_DEV_MEMBER void round_t() {
        ZL_PROF_T0();
        u32 pa0 = 0, pa1 = 0, pa2 = 0, pa3 = 0, pa4 = 0, bp0 = 0, bp1 = 0, ti = 0;
        bool v0 = false, v1 = false, v2 = false, v3 = false, v4 = false, vb = false, vt = false, m0 = false, m1 = false;
        u32 t0tag = 0, t1tag = 0;
        u32 pa5 = 0, ti2 = 0, ip4 = 0, ip5 = 0; bool v5 = false, vt2 = false;
        bool const on = (st == ZL_SEARCH) || ((K & ZL_EN_COUNT) && (st == ZL_COUNT || st == ZL_BACK))
                     || ((K & ZL_EN_POST) && (st == ZL_POST || st == ZL_LOADW)) || ((K & ZL_EN_START) && st == ZL_START);
        if (st == ZL_SEARCH) {
            ZE_COUNT_ITER();
            t0tag = ze_tag4((u32)w0); t1tag = ze_tag4((u32)w1);
            cur0 = ip0;
            T[zl_hash(hT, w0)] = E::make(ip0 + 1u, t0tag);       // both writes of the pair (ip1's happens on every path of the reference's
            T[zl_hash(hT, w1)] = E::make(ip1 + 1u, t1tag);       // iteration), in the reference's order, before the next pair's entries are read
            m0 = E::maybe(eX, t0tag); m1 = E::maybe(eY, t1tag);
            pa0 = ip2 - 1u - rep1; v0 = true;          // byte before the repcode candidate + its 4 bytes
            ip4 = ip2 + step; ip5 = ip3 + step;
            pa1 = ip4; pa2 = ip5; v1 = v2 = true;       // bytes of the pair after next
            pa3 = E::pos(eX) - 1u; v3 = m0;
            pa5 = E::pos(eY) - 1u; v5 = m1;
            pa4 = ip2 - 1u; v4 = ip2 - ip0 > 8u;         // the byte before ip2 is in w0 unless the step is large
            ti = zl_hash(hT, w2); ti2 = zl_hash(hT, w3); vt = vt2 = true;   // entries of the next pair
        } else if ((K & ZL_EN_COUNT) && st == ZL_COUNT) {
            pa0 = ca; pa1 = ca + 8u; pa2 = cb; pa3 = cb + 8u; v0 = v1 = v2 = v3 = true;
            if (needBack) { vb = true; bp0 = ip0; bp1 = mpos; }
        } else if ((K & ZL_EN_COUNT) && st == ZL_BACK) {
            vb = true; bp0 = ip0 - bk; bp1 = mpos - bk;
        } else if ((K & ZL_EN_POST) && st == ZL_POST) {
            pa0 = cur0 + 2u; pa1 = ip0 - 2u; pa2 = ip0 + 6u; v0 = !haveIns; v1 = v2 = true;
            pa3 = ip0 - rep2; v3 = rep2 > 0u;
        } else if ((K & ZL_EN_POST) && st == ZL_LOADW) {
            pa0 = ip0; pa1 = ip0 + 1u; v0 = v1 = true;
            pa3 = ip0 - rep2; v3 = chk && rep2 > 0u;
        } else if ((K & ZL_EN_START) && st == ZL_START) {
            ti = zl_hash(hT, w0); ti2 = zl_hash(hT, w1); vt = vt2 = true;
            pa1 = ip2; pa2 = ip3; v1 = v2 = true;
        }
        ZL_PROF_T1();
        pa0 = v0 ? pa0 : 0; pa1 = v1 ? pa1 : 0; pa2 = v2 ? pa2 : 0; pa3 = v3 ? pa3 : 0; pa4 = v4 ? pa4 : 0;
        if (!vb) { bp0 = 8; bp1 = 8; }
        if (!vt) ti = 0;
        if (!vt2) ti2 = 0;
        if (!v5) pa5 = 0;
        u32 const q0 = zl_fwd_at(n, pa0), q1 = zl_fwd_at(n, pa1), q2 = zl_fwd_at(n, pa2), q3 = zl_fwd_at(n, pa3), q4 = zl_fwd_at(n, pa4), q5 = zl_fwd_at(n, pa5);
        u32 const qb0 = zl_back_at(bp0), qb1 = zl_back_at(bp1);
        u64 r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, rb0 = 0, rb1 = 0; u32 t = 0, t2 = 0;
        if (v0) r0 = ld64(src + q0);
        if (v1) r1 = ld64(src + q1);
        if (v2) r2 = ld64(src + q2);
        if (v3) r3 = ld64(src + q3);
        if (v4) r4 = ld64(src + q4);
        if (v5) r5 = ld64(src + q5);
        if ((K & ZL_EN_COUNT) && vb) { rb0 = ld64(src + qb0); rb1 = ld64(src + qb1); }
        if (vt) t = (u32)T[ti];
        if (vt2) t2 = (u32)T[ti2];
        ZL_ROUND_FENCE9(r0, r1, r2, r3, r4, rb0, rb1, t, t2); ZL_ROUND_FENCE9(r5, r5, r5, r5, r5, r5, r5, t, t2);
        ZL_PROF_T2();
        u64 const d0 = zl_fwd_fix(r0, pa0, q0), d1 = zl_fwd_fix(r1, pa1, q1), d2 = zl_fwd_fix(r2, pa2, q2), d3 = zl_fwd_fix(r3, pa3, q3);
        u64 const d4 = zl_fwd_fix(r4, pa4, q4), d5 = zl_fwd_fix(r5, pa5, q5);
        u64 const b0 = zl_back_fix(rb0, bp0, qb0), b1 = zl_back_fix(rb1, bp1, qb1);
        if (!on) return;
        if (st == ZL_SEARCH) {
            u32 const gap = ip2 - ip0;                       // == step, or step - 1 right after a step increment
            wIns = w2; haveIns = (gap == 2u);                // cur0 + 2 == ip2 for a match found at ip0 / the repcode position
            u32 const rval = (u32)(d0 >> 8);
            if (((u32)w2 == rval) & (rep1 > 0u)) {
                ip0 = ip2; mpos = ip0 - rep1;
                u8 const prev = gap > 8u ? (u8)d4 : (u8)(w0 >> (8u * (gap - 1u)));
                u32 const e = (prev == (u8)d0) ? 1u : 0u;
                ip0 -= e; mpos -= e; offcode = 1u; mLength = 4u + e;
                begin_count(ip0 + mLength, mpos + mLength, ZC_FOUND); needBack = false; bk = 0; more = false;
            } else if (m0 && (u32)d3 == (u32)w0) {
                found_at(eX);
            } else if (m1 && (u32)d5 == (u32)w1) {
                // reference: shift the pair by one (ip0 <- ip1, ip1 <- ip2, ip2 <- ip3), ip1's candidate matched
                ip0 = ip1; cur0 = ip0;
                wIns = w3;                                   // cur0 is the old ip1: cur0 + 2 == ip3 when ip2 == old ip0 + 2
                if (step <= 4u) T[ti] = E::make(ip2 + 1u, ze_tag4((u32)w2));
                found_at(eY);
            } else {
                // next pair: its entries as the reference reads them — ip2's after this pair's writes (done above), ip3's also
                // after the write of ip2 that the next round starts with
                u32 const nX = t, nY = (ti2 == ti) ? (u32)E::make(ip2 + 1u, ze_tag4((u32)w2)) : t2;
                ip0 = ip2; ip1 = ip3; w0 = w2; w1 = w3; eX = nX; eY = nY;
                ip2 = ip4; ip3 = ip5; w2 = d1; w3 = d2;
                if (ip2 >= nextStep) { step++; nextStep += 128u; }
                if (ip3 >= ilimit) finish();
            }
        } else if ((K & ZL_EN_COUNT) && st == ZL_COUNT) {
            u32 const lim = n - ca;
            u32 c = zl_common_fwd16(d0, d1, d2, d3);
            if (c > lim) c = lim;
            acc += c;
            if (needBack) {
                u32 const limit = zj_min(ip0 - anchor, mpos);
                u32 e = zl_common_back8(b0, b1); if (e > limit) e = limit;
                bk = e; more = (e == 8u) && (limit > 8u); needBack = false;
            }
            if (c == 16u && lim > 16u) { ca += 16u; cb += 16u; }
            else if (cont == ZC_FOUND) { mLength += acc; if (more) st = ZL_BACK; else fin(); }
            else {                                     // ZC_REPLOOP
                u32 const rLength = acc + 4u;
                { u32 const x = rep2; rep2 = rep1; rep1 = x; }
                T[zl_hash(hT, w0)] = E::make(ip0 + 1u, ze_tag4((u32)w0));
                ip0 += rLength;
                ze_store(o, anchor, 0u, 1u, rLength);
                anchor = ip0;
                if (ip0 <= ilimit) { chk = true; st = ZL_LOADW; } else finish();
            }
        } else if ((K & ZL_EN_POST) && st == ZL_POST) {
            u64 const wa = haveIns ? wIns : d0, q0 = d1, q1 = d2;
            T[zl_hash(hT, wa)] = E::make(cur0 + 2u + 1u, ze_tag4((u32)wa));
            T[zl_hash(hT, q0)] = E::make(ip0 - 2u + 1u, ze_tag4((u32)q0));
            w0 = (q0 >> 16) | (q1 << 48); w1 = (q0 >> 24) | (q1 << 40);
            if ((rep2 > 0u) && ((u32)w0 == (u32)d3)) { begin_count(ip0 + 4u, ip0 + 4u - rep2, ZC_REPLOOP); needBack = false; }
            else outer();
        } else if ((K & ZL_EN_START) && st == ZL_START) {
            eX = t; eY = (ti2 == ti) ? (u32)E::make(ip0 + 1u, ze_tag4((u32)w0)) : t2;     // ip1's entry is read after ip0's write
            w2 = d1; w3 = d2; st = ZL_SEARCH;
        } else if ((K & ZL_EN_COUNT) && st == ZL_BACK) {
            u32 const limit = zj_min(ip0 - anchor, mpos) - bk;
            u32 e = zl_common_back8(b0, b1); if (e > limit) e = limit;
            bk += e; more = (e == 8u) && (limit > 8u);
            if (!more) fin();
        } else if ((K & ZL_EN_POST) && st == ZL_LOADW) {
            w0 = d0; w1 = d1;
            if (chk && (rep2 > 0u) && ((u32)w0 == (u32)d3)) { begin_count(ip0 + 4u, ip0 + 4u - rep2, ZC_REPLOOP); needBack = false; }
            else outer();
            chk = false;
        }
    }
};

// Frames below this size run the plain loops (the machines assume 8-byte loads always fit in the frame).
#define ZL_MIN_FRAME 64u
