// zj_match_run.h — the level-3 lane machine of large batches: double-fast (N/compress/zstd_double_fast.c:105-323) one frame per lane,
// with the frame's sequential byte streams in REGISTER WINDOWS and, where the frame has need flags (zj_need.h), a whole run of
// positions decided per round.
//
// What bounds the lane-per-frame match kernel is the number of memory requests that miss the L2 (the device serves
// ~45 G random requests per second whatever their size, and 65 536 lanes x ~6 lines per round is more than the L2 holds, so even the bytes
// a lane read one round earlier come back from the fabric).  ZLaneD asks for the bytes at ip + 1 - off1 (repcode test), the word of the
// position after next and a flag byte in every search round: for the search-dense class of frames that is two thirds of its requests.
// Here a lane keeps
//     A, B    the 16 source bytes at ip                       (one 8-byte refill per search round)
//     RA, RB  the 16 bytes at ip - off1, the repcode stream    (one refill)
//     FA, FB  the 16 flag bytes of ip .. ip + 15               (one refill, flagged frames only)
// and slides them as ip advances.  Everything a search round has to know before it can issue its loads is then in registers:
//   * the repcode test of ip and of the positions behind it (bytes of A:B against RA:RB, four at a time);
//   * with flags: how many positions behind ip are QUIET — no probe can match there (ZN_NEED_L / ZN_NEED_S clear) and the repcode test
//     fails — so that the reference's iteration at such a position only writes its two entries (where ZN_INS_* asks for them) and moves
//     on.  The round decides ip, commits the run of up to 7 quiet positions behind it (their writes, forwarded to the entries already
//     requested where they share a bucket) and has asked, in the same batch of loads, for the table entries of the first position P
//     after the run.  One round trip per probed position instead of one per position; the reference's decisions in the reference's order.
// Frames without flags run the same machine with every flag set (no quiet positions): windows only.
// A step above 8 (reached after ~9 000 unmatched positions: incompressible data) leaves the windows' reach; such positions take the
// slow route FAR -> START -> SEARCH, and a short match found there fetches ip1's long entry in two extra states (SL1, SL2).
// States COUNT / BACK / POST / LOADW / START and the rotation of the non-search states are ZLaneD's (zj_match_lane.h).
// Exactness: tests/test_emu_encode.py (lane-serial build against the reference), tools/fuzz_emu_need.py, tests/test_gpu_encode.py.
#pragma once

#if ZJ_ON_GPU
#define ZR_ANY(x) (__ballot(x) != 0)          /* does any active lane of the wave see x */
#define ZR_UNROLL _Pragma("unroll")
#else
#define ZR_ANY(x) (x)
#define ZR_UNROLL
#endif
enum { ZL_FAR = ZL_DONE + 1, ZL_SL1, ZL_SL2 };
// Cache-policy hints (ZR_NT, a bit set; GPU build only; 1: table reads, 2: table writes): a table entry is touched once and not again
// for thousands of rounds, while the source lines at ip and at the match source are asked for again within the next three rounds, so the
// streaming hint (nt = 1) on the table traffic might have kept those lines in the L2.  Measured (profiles/r03/j_nontemporal_ab.txt): no
// gain — match kernel 148-150 ms without, 143-155 with both, 158-161 with the sequence records streamed as well.  Off.
#ifndef ZR_NT
#define ZR_NT 0
#endif
#if ZJ_ON_GPU
#define ZR_TLOAD(p) ((ZR_NT & 1) ? __builtin_nontemporal_load(p) : *(p))
#define ZR_TSTORE(p, v) do { if (ZR_NT & 2) __builtin_nontemporal_store((v), (p)); else *(p) = (v); } while (0)
#else
#define ZR_TLOAD(p) (*(p))
#define ZR_TSTORE(p, v) (*(p) = (v))
#endif
#ifndef ZR_JMAX_DEFAULT
#define ZR_JMAX_DEFAULT 5u        /* quiet positions committed per round at most (measured on the metric configuration: 3 / 5 / 7 -> 133 / 123 / 135 ms) */
#endif
#ifndef ZR_PERIOD
#define ZR_PERIOD 4u              /* rotation of the non-search states: count, post-insert, restart, one search-only round */
#endif

ZJ_DEV u64 zr_zero80(u64 x) { u64 const m = 0x7F7F7F7F7F7F7F7FULL; return ~(((x & m) + m) | x | m); }      // 0x80 in every byte of x that is zero
ZJ_DEV u64 zr_shr(u64 lo, u64 hi, u32 k) { return (lo >> (8u * k)) | (hi << (64u - 8u * k)); }                // bytes [k, k + 8) of hi:lo, k = 1 .. 7 (constant)
ZJ_DEV u64 zr_ext(u64 lo, u64 hi, u32 s) {                                                                    // bytes [s, s + 8) of hi:lo, s = 0 .. 8 (per lane)
    u64 const v = (lo >> ((8u * s) & 63u)) | ((hi << 1) << ((63u - 8u * s) & 63u));
    return s >= 8u ? hi : v;
}

template <class E, u32 JMAX = ZR_JMAX_DEFAULT>
struct ZLaneR {
    typedef typename E::T Ent;
    const u8* src; u32 n, ilimit; Ent* HL; Ent* HS; ZLHash hL, hS;
    const u8* F;                                      // the frame's flag bytes, or nullptr: every probe and every write is made
    ZEOut o;
    u32 st, cont, lastLL;
    u32 ip, ip1, anchor, off1, off2, step, nextStep, curr;
    u64 A, B, RA, RB, FA, FB;                         // windows: source at ip, source at ip - off1, flags at ip (16 bytes each; valid in SEARCH)
    u64 w1, wIns;                                     // once a match is found at ip: the word at ip1 (SHORT_L1's comparison), the word at curr + 2 (first post-insert)
    u32 el0, es0, el1, hl0, hs0, tl0, hl1, tl1, fN1, fIns;   // entries of ip as the reference reads them; ip1's long bucket / tag / flags (fin), flags of curr + 2
    u32 ca, cb, acc;                                  // forward count in progress
    u32 mpos, mpos2, mLength, offset, bk, bk2;
    bool more, more2, cvalid, needBack, needCand, chk;

    ZJ_DEV_MEMBER void init(const u8* s, u32 size, const ZEParams& p, u8* table, u8* fscratch, u32 maxSrc, const u8* flags = nullptr) {
        src = s; n = size; ilimit = size - 8u; hL = zl_hash_of(8, p.hashLog); hS = zl_hash_of(p.minMatch, p.chainLog);
        HL = (Ent*)table; HS = HL + (1u << p.hashLog);
        F = flags;
        o.seqs = (ZESeq*)fscratch; o.litOff = (u32*)(fscratch + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); o.n = 0; o.lit = 0;
        ip = 1; anchor = 0; off1 = 1; off2 = 0; chk = false; lastLL = size;
        needBack = needCand = more = more2 = cvalid = false;
        A = B = RA = RB = 0; FA = FB = 0x0F0F0F0F0F0F0F0FULL;
        st = ZL_LOADW;
    }
    ZJ_DEV_MEMBER u32 prod_long(u64 v) const { return zl_prod_hi(hL, v); }
    ZJ_DEV_MEMBER u32 idx_long(u32 p) const { return p >> hL.rsh; }
    ZJ_DEV_MEMBER u32 tag_long(u32 p) const { return (p >> (hL.rsh - 15u)) & 0x7FFFu; }
    ZJ_DEV_MEMBER void put_long_if(bool on, u64 v, u32 pos1) { if (on) { u32 const p = prod_long(v); ZR_TSTORE(&HL[idx_long(p)], E::make(pos1, tag_long(p))); } }
    ZJ_DEV_MEMBER void put_short_if(bool on, u64 v, u32 pos1) { if (on) ZR_TSTORE(&HS[zl_hash(hS, v)], E::make(pos1, ze_tag4((u32)v))); }
    ZJ_DEV_MEMBER void finish() { lastLL = n - anchor; st = ZL_DONE; }
    ZJ_DEV_MEMBER void outer() {                      // outer-loop header of the reference: reset the step, make sure one more position fits
        step = 1; nextStep = ip + 256u; ip1 = ip + 1u;
        if (ip1 > ilimit) finish(); else st = ZL_START;
    }
    ZJ_DEV_MEMBER void begin_count(u32 a, u32 b, u32 c) { ca = a; cb = b; acc = 0; cont = c; st = ZL_COUNT; }
    ZJ_DEV_MEMBER void advance() { ip += mLength; anchor = ip; if (ip <= ilimit) st = ZL_POST; else finish(); }
    ZJ_DEV_MEMBER void fin() {                        // a long / short match is final: apply the backward extension, store it
        ip -= bk; mLength += bk;
        off2 = off1; off1 = offset;
        if (step < 4u && (fN1 & 4u)) ZR_TSTORE(&HL[hl1], E::make(ip1 + 1u, tl1));        // (ip1: the position that was ip1 when the match was found)
        ze_store(o, anchor, ip - anchor, offset + 3u, mLength);
        advance();
    }
    ZJ_DEV_MEMBER void fin_or_back() { if (more) st = ZL_BACK; else fin(); }
    // a match at ip (any kind): what the later states need from the windows
    ZJ_DEV_MEMBER void leave_search() { wIns = zr_ext(A, B, 2u); fIns = (u32)(FA >> 16) & 15u; }

    ZJ_DEVM u32 phase_of(u32 r) { return r % ZR_PERIOD; }
    ZJ_DEVM u32 default_period() { return ZR_PERIOD; }
    ZJ_DEVM bool takes_flags_late() { return true; }                 // flags may arrive while the frame is under way: every use of F tolerates "all set" before
    ZJ_DEV_MEMBER void take_flags(const u8* flags) { F = flags; }
    ZL_PROF_MEMBERS
    ZJ_DEV_MEMBER void round(u32 r) {                 // see ZLaneD::round: search every round, the other states in turns
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
        // load slots: s0 .. s4 source words (forward-clamped), sb0 / sb1 backward words, f0 (8 flag bytes) / f1 (4 flag bytes), t0 / t1 table entries
        u32 pa0 = 0, pa1 = 0, pa2 = 0, pa3 = 0, pa4 = 0, bp0 = 0, bp1 = 0, fa0 = 0, fa1 = 0, ti0 = 0, ti1 = 0;
        bool v0 = false, v1 = false, v2 = false, v3 = false, v4 = false, vb = false, vf0 = false, vf1 = false, vt0 = false, vt1 = false;
        bool ml0 = false, ms0 = false, rep0 = false, far = false; u32 J = 0, adv = 0, fP = 0; u64 hw = 0;
        bool const on = (st == ZL_SEARCH) || ((K & ZL_EN_COUNT) && (st == ZL_COUNT || st == ZL_BACK || st == ZL_SL1 || st == ZL_SL2))
                     || ((K & ZL_EN_POST) && (st == ZL_POST || st == ZL_LOADW || st == ZL_FAR)) || ((K & ZL_EN_START) && st == ZL_START);
        // ---- phase 1: what does this lane's state need (and the table writes that precede its reads) ----
        if (st == ZL_SEARCH) {
            ZE_COUNT_ITER();
            curr = ip;
            u32 const fI = (u32)FA & 15u, ts = ze_tag4((u32)A);
            if (fI & 4u) ZR_TSTORE(&HL[hl0], E::make(ip + 1u, tl0));
            if (fI & 8u) ZR_TSTORE(&HS[hs0], E::make(ip + 1u, ts));
            ml0 = E::maybe(el0, tl0); ms0 = E::maybe(es0, ts);
            pa2 = E::pos(el0) - 1u; v2 = ml0;
            pa3 = E::pos(es0) - 1u; v3 = ms0;
            // repcode tests of ip + j, j = 0 .. 7: bytes j + 1 .. j + 4 of the source window against the repcode window
            u64 const y0 = zr_zero80(A ^ RA), y1 = zr_zero80(B ^ RB);
            u64 H = zr_shr(y0, y1, 1) & zr_shr(y0, y1, 2) & zr_shr(y0, y1, 3) & zr_shr(y0, y1, 4);      // 0x80 in byte j: ip + j has a repcode match at ip + j + 1
            if (off1 == 0u) H = 0;
            rep0 = (H & 0x80u) != 0;
            if (step == 1u) {
                // quiet positions behind ip: no probe needed, no repcode match, still below nextStep and ilimit
                u64 const NP = ~zr_zero80(FA & 0x0303030303030303ULL) & 0x8080808080808080ULL;                 // 0x80 in byte j: ip + j needs a probe
                u64 const NQ = ((NP | H) >> 8) | (0x80ULL << 56);                                              // byte j - 1: ip + j is not quiet (j = 1 .. 7); a stop at j = 8
                J = (u32)__builtin_ctzll(NQ) >> 3;
                u32 const roomStep = nextStep - ip, lim0 = roomStep >= 2u ? roomStep - 2u : 0u, lim1 = ilimit - ip - 1u;      // ip + j + 1 < nextStep, ip + j + 1 <= ilimit
                J = zj_min(zj_min(J, JMAX), zj_min(lim0, lim1));
                adv = J + 1u;
            } else adv = step;
            far = adv > 8u;
            if (!far) {
                hw = zr_ext(A, B, adv); fP = (u32)zr_ext(FA, FB, adv) & 15u;     // the next position to decide, P = ip + adv: its word and flags
                vt0 = (fP & 1u) != 0; vt1 = (fP & 2u) != 0;
                pa0 = ip + 16u; v0 = true;                                        // window refills
                pa1 = ip - off1 + 16u; v1 = true;
                fa0 = ip + 16u; vf0 = F != nullptr;
            }
        } else if ((K & ZL_EN_COUNT) && st == ZL_COUNT) {
            pa0 = ca; pa1 = ca + 8u; pa2 = cb; pa3 = cb + 8u; v0 = v1 = v2 = v3 = true;
            if (needBack) { vb = true; if (cont == ZC_SHORT_L1) { bp0 = ip1; bp1 = mpos2; } else { bp0 = ip; bp1 = mpos; } }
            if (needCand) { v4 = true; pa4 = mpos2; }
        } else if ((K & ZL_EN_COUNT) && st == ZL_BACK) {
            vb = true; bp0 = ip - bk; bp1 = mpos - bk;
        } else if ((K & ZL_EN_COUNT) && st == ZL_SL1) {
            pa0 = ip1; v0 = true; fa1 = ip1; vf1 = F != nullptr;
        } else if ((K & ZL_EN_COUNT) && st == ZL_SL2) {
            hw = w1; vt0 = (fN1 & 1u) != 0;
        } else if ((K & ZL_EN_POST) && st == ZL_POST) {
            pa1 = ip - 2u; pa2 = ip + 6u; v1 = v2 = true;
            pa3 = ip - off2; v3 = off2 > 0u;
            fa0 = ip; fa1 = ip - 2u; vf0 = vf1 = F != nullptr;
        } else if ((K & ZL_EN_POST) && (st == ZL_LOADW || st == ZL_FAR)) {
            pa0 = ip; v0 = true;
            pa3 = ip - off2; v3 = st == ZL_LOADW && chk && off2 > 0u;
            fa0 = ip; vf0 = F != nullptr;
        } else if ((K & ZL_EN_START) && st == ZL_START) {
            hw = A; u32 const fI = (u32)FA & 15u; vt0 = (fI & 1u) != 0; vt1 = (fI & 2u) != 0;
            pa0 = ip + 8u; pa1 = ip - off1; pa2 = ip - off1 + 8u; v0 = v1 = v2 = true;
            fa0 = ip + 8u; vf0 = F != nullptr;
        }
        // the hashes of the word whose table entries this round reads (search: P's; restart: ip's; SL2: ip1's)
        u32 const hp = prod_long(hw), nhl = idx_long(hp), ntl = tag_long(hp), nhs = zl_hash(hS, hw);
        ti0 = vt0 ? nhl : 0u; ti1 = vt1 ? nhs : 0u;
        // ---- phase 2: one batch of loads for all states ----
        ZL_PROF_T1();
        pa0 = v0 ? pa0 : 0; pa1 = v1 ? pa1 : 0; pa2 = v2 ? pa2 : 0; pa3 = v3 ? pa3 : 0; pa4 = v4 ? pa4 : 0;
        fa0 = vf0 ? fa0 : 0; fa1 = vf1 ? fa1 : 0;
        if (!vb) { bp0 = 8; bp1 = 8; }
        u32 const q0 = zl_fwd_at(n, pa0), q1 = zl_fwd_at(n, pa1), q2 = zl_fwd_at(n, pa2), q3 = zl_fwd_at(n, pa3), q4 = zl_fwd_at(n, pa4);
        u32 const qf0 = zl_fwd_at(n, fa0), qf1 = zl_fwd_at(n, fa1);
        u32 const qb0 = zl_back_at(bp0), qb1 = zl_back_at(bp1);
        u64 r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, rb0 = 0, rb1 = 0, rf0 = 0; u32 rf1 = 0, t0 = 0, t1 = 0;
#ifdef ZR_COUNT_SLOT                                          /* analysis builds: which load slots a frame activates (tools, never the product) */
        ZR_COUNT_SLOT(0, v0); ZR_COUNT_SLOT(1, v1); ZR_COUNT_SLOT(2, v2); ZR_COUNT_SLOT(3, v3); ZR_COUNT_SLOT(4, (K & ZL_EN_COUNT) && v4); ZR_COUNT_SLOT(5, (K & ZL_EN_COUNT) && vb);
        ZR_COUNT_SLOT(6, vf0); ZR_COUNT_SLOT(7, (K & (ZL_EN_COUNT | ZL_EN_POST)) && vf1); ZR_COUNT_SLOT(8, vt0); ZR_COUNT_SLOT(9, vt1);
#endif
        if (v0) r0 = ld64(src + q0);
        if (v1) r1 = ld64(src + q1);
        if (v2) r2 = ld64(src + q2);
        if (v3) r3 = ld64(src + q3);
        if (vt0) t0 = (u32)ZR_TLOAD(&HL[ti0]);
        if (vt1) t1 = (u32)ZR_TLOAD(&HS[ti1]);
        if (vf0) rf0 = ld64(F + qf0);
        if ((K & (ZL_EN_COUNT | ZL_EN_POST)) && vf1) rf1 = ld32(F + qf1);
        if ((K & ZL_EN_COUNT) && v4) r4 = ld64(src + q4);
        if ((K & ZL_EN_COUNT) && vb) { rb0 = ld64(src + qb0); rb1 = ld64(src + qb1); }
        ZL_ROUND_FENCE9(r0, r1, r2, r3, r4, rb0, rb1, t0, t1);
        ZL_ROUND_FENCE2(rf0, rf1);
        ZL_PROF_T2();
        u64 const d0 = zl_fwd_fix(r0, pa0, q0), d1 = zl_fwd_fix(r1, pa1, q1), d2 = zl_fwd_fix(r2, pa2, q2), d3 = zl_fwd_fix(r3, pa3, q3);
        u64 const d4 = zl_fwd_fix(r4, pa4, q4);
        u64 const b0 = zl_back_fix(rb0, bp0, qb0), b1 = zl_back_fix(rb1, bp1, qb1);
        u64 const g0 = F ? zl_fwd_fix(rf0, fa0, qf0) : 0x0F0F0F0F0F0F0F0FULL;                                   // 8 flag bytes at fa0 (a frame without flags: all set)
        u32 const g1 = F ? (u32)zl_fwd_fix((u64)rf1, fa1, qf1) : 0x0F0F0F0Fu;                                   // 4 flag bytes at fa1 (never clamped: fa1 + 4 <= n)
        if (!on) return;
        // ---- phase 3: consume ----
        if (st == ZL_SEARCH) {
            bool const near1 = !far && (J == 0u);                  // P is the reference's ip1 (ip + step): its long entry, bucket and tag are this round's
            if (rep0) {
                leave_search();
                begin_count(ip + 5u, ip + 5u - off1, ZC_REP1); needBack = false; needCand = false;
            } else if ((ml0 && d2 == A) || (ms0 && (u32)d3 == (u32)A)) {
                bool const isLong = ml0 && d2 == A;
                leave_search();
                ip1 = ip + step;
                if (near1) { w1 = hw; hl1 = nhl; tl1 = ntl; fN1 = fP; el1 = vt0 ? t0 : 0u; }
                else if (!far) { w1 = zr_ext(A, B, 1u); u32 const p1 = prod_long(w1); hl1 = idx_long(p1); tl1 = tag_long(p1); fN1 = (u32)(FA >> 8) & 15u; el1 = 0; }   // ip + 1 is quiet: its probe cannot match
                if (isLong) {
                    mpos = E::pos(el0) - 1u;
                    begin_count(ip + 8u, mpos + 8u, ZC_LONG); needBack = true; needCand = false;
                    if (far) fN1 = 0;                               // (step > 8: fin() writes nothing for ip1)
                } else {
                    mpos = E::pos(es0) - 1u;
                    if (far) st = ZL_SL1;                           // ip1's word, flags and long entry are not here: two more states fetch them
                    else {
                        begin_count(ip + 4u, mpos + 4u, ZC_SHORT); needBack = true;
                        needCand = (E::pos(el1) > 1u) && E::maybe(el1, tl1); mpos2 = E::pos(el1) - 1u; cvalid = false;
                    }
                }
            } else if (far) {
                // no match, and the next position is out of the windows' reach: reload there
                if (ip + step >= nextStep) { step++; nextStep += 256u; ip += step - 1u; } else ip += step;
                if (ip + step > ilimit) finish(); else st = ZL_FAR;
            } else {
                // no match at ip: commit the quiet run behind it, move to P with the entries this round fetched
                u32 e0 = t0, e1 = t1;
ZR_UNROLL
                for (u32 j = 1; j <= JMAX; j++) {
                    if (!ZR_ANY(j <= J)) break;               // (wave-uniform: no lane's run is this long)
                    if (j <= J) {
                        u64 const wq = zr_shr(A, B, j); u32 const fq = (u32)(FA >> (8u * j));
                        if (fq & 4u) { u32 const p = prod_long(wq), b = idx_long(p); Ent const e = E::make(ip + j + 1u, tag_long(p)); ZR_TSTORE(&HL[b], e); if (b == nhl) e0 = (u32)e; }
                        if (fq & 8u) { u32 const b = zl_hash(hS, wq); Ent const e = E::make(ip + j + 1u, ze_tag4((u32)wq)); ZR_TSTORE(&HS[b], e); if (b == nhs) e1 = (u32)e; }
                    }
                }
                if (!vt0) e0 = 0;
                if (!vt1) e1 = 0;
                if (ip + step >= nextStep) { step++; nextStep += 256u; }       // (only ever with J == 0)
                ip += adv;
                A = zr_ext(A, B, adv); B = zr_ext(B, d0, adv);
                RA = zr_ext(RA, RB, adv); RB = zr_ext(RB, d1, adv);
                FA = zr_ext(FA, FB, adv); FB = zr_ext(FB, g0, adv);
                el0 = e0; es0 = e1; hl0 = nhl; hs0 = nhs; tl0 = ntl;
                if (ip + step > ilimit) finish();
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
            } else {                                   // ZC_REPLOOP: immediate repcode after a match (A = the word at ip, FA = its flags)
                u32 const rLength = acc + 4u;
                { u32 const t = off2; off2 = off1; off1 = t; }
                put_short_if(((u32)FA & 8u) != 0, A, ip + 1u);
                put_long_if(((u32)FA & 4u) != 0, A, ip + 1u);
                ze_store(o, anchor, 0u, 1u, rLength);
                ip += rLength; anchor = ip;
                if (ip <= ilimit) { chk = true; st = ZL_LOADW; } else finish();
            }
        } else if ((K & ZL_EN_COUNT) && st == ZL_BACK) {
            u32 const limit = zj_min(ip - anchor, mpos) - bk;
            u32 e = zl_common_back8(b0, b1); if (e > limit) e = limit;
            bk += e; more = (e == 8u) && (limit > 8u);
            if (!more) fin();
        } else if ((K & ZL_EN_COUNT) && st == ZL_SL1) {
            w1 = d0; fN1 = g1 & 15u; st = ZL_SL2;
        } else if ((K & ZL_EN_COUNT) && st == ZL_SL2) {
            el1 = vt0 ? t0 : 0u; hl1 = nhl; tl1 = ntl;            // (read after ip's own writes, as the reference reads it)
            begin_count(ip + 4u, mpos + 4u, ZC_SHORT); needBack = true;
            needCand = (E::pos(el1) > 1u) && E::maybe(el1, tl1); mpos2 = E::pos(el1) - 1u; cvalid = false;
        } else if ((K & ZL_EN_POST) && st == ZL_POST) {
            u64 const q0w = d1, q1w = d2;
            u64 const wb = q0w, wc = (q0w >> 8) | (q1w << 56);
            u32 const ins = curr + 2u;
            // fIns = flags of curr + 2, g1 = flags of ip - 2, ip - 1, ip, ip + 1 (one byte each)
            put_long_if((fIns & 4u) != 0, wIns, ins + 1u);
            put_long_if((g1 & 4u) != 0, wb, ip - 2u + 1u);
            put_short_if((fIns & 8u) != 0, wIns, ins + 1u);
            put_short_if((g1 & 0x800u) != 0, wc, ip - 1u + 1u);
            A = (q0w >> 16) | (q1w << 48); FA = g0;
            if ((off2 > 0u) && ((u32)A == (u32)d3)) { begin_count(ip + 4u, ip + 4u - off2, ZC_REPLOOP); needBack = false; needCand = false; }
            else outer();
        } else if ((K & ZL_EN_POST) && st == ZL_LOADW) {
            A = d0; FA = g0;
            if (chk && (off2 > 0u) && ((u32)A == (u32)d3)) { begin_count(ip + 4u, ip + 4u - off2, ZC_REPLOOP); needBack = false; needCand = false; }
            else outer();
            chk = false;
        } else if ((K & ZL_EN_POST) && st == ZL_FAR) {
            A = d0; FA = g0; st = ZL_START;
        } else if ((K & ZL_EN_START) && st == ZL_START) {
            el0 = vt0 ? t0 : 0u; es0 = vt1 ? t1 : 0u; hl0 = nhl; hs0 = nhs; tl0 = ntl;
            B = d0; RA = d1; RB = d2; FB = g0;
            st = ZL_SEARCH;
        }
    }
};
