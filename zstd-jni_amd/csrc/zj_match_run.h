// zj_match_run.h — the level-3 lane machine of large batches: double-fast (N/compress/zstd_double_fast.c:105-323) one frame per lane,
// with the frame's sequential byte streams in REGISTER WINDOWS and, where the frame has need flags (zj_need.h), a whole run of
// positions decided per round.
//
// What bounds the lane-per-frame match kernel is the number of memory requests that miss the L2 (the device serves
// ~45 G random requests per second whatever their size, and 65 536 lanes x ~6 lines per round is more than the L2 holds, so even the bytes
// a lane read one round earlier come back from the fabric).  ZLaneD asks for the bytes at ip + 1 - off1 (repcode test), the word of the
// position after next and a flag byte in every search round: for the search-dense class of frames that is two thirds of its requests.
// Here a lane keeps
//     A, B, C     the 16 .. 23 source bytes at ip                       (refilled 8 bytes at a time, only when fewer than 16 would be left: round 4)
//     RA, RB, RC  the bytes at ip - off1, the repcode stream             (the same)
//     FA, FB, FC  the flag bytes of ip ..                                (the same, flagged frames only)
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
#ifndef ZR_SHADOW
#define ZR_SHADOW 1               /* the "no match" step of a lane without a candidate is taken while the round's loads fly (round 5) */
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


// Predicated global accesses without a divergent region (GPU): exec narrowed to the lanes that ask, ONE memory instruction, exec restored — three instructions
// where `if (c) v = *p` compiles to a compare, s_and_saveexec, a skip branch, the access, s_or and the moves that merge the result.  The loaded value of a lane
// that did not ask is undefined (nothing reads it: every consumer sits behind the same predicate).  The loads are invisible to the compiler's wait counting,
// so the round waits for them itself, once, in ZR_WAIT (s_waitcnt vmcnt(0), with the results as operands so that no use can be scheduled above it).
// The predicate of such an access is a LANE MASK (ZRM: an SGPR pair on the GPU, a bool in the lane-serial emulation) built from ballots of simple compares and
// scalar and / or — a ballot of a composite bool makes the compiler materialise 0 / 1 in a VGPR and compare it again (two instructions per access).
#if ZJ_ON_GPU
typedef u64 ZRM;
#define ZRM_OF(c) ((ZRM)__builtin_amdgcn_ballot_w64(c))        /* c: ONE compare */
#define ZRM_NOT(m) (~(m))                                      /* (bits of inactive lanes do not matter: every use is ANDed with exec) */
#define ZRM_NONE ((ZRM)0)
#define ZRM_ANY(m) (((m) & __builtin_amdgcn_ballot_w64(true)) != 0)
ZJ_DEV u64 zr_ld64_m(ZRM m, const u8* p) {
    u64 v, sv;
    asm volatile("s_and_saveexec_b64 %1, %2\n\tglobal_load_dwordx2 %0, %3, off\n\ts_mov_b64 exec, %1" : "=&v"(v), "=&s"(sv) : "s"(m), "v"(p) : "memory", "scc");
    return v;
}
ZJ_DEV u32 zr_ld32_m(ZRM m, const u8* p) {
    u32 v; u64 sv;
    asm volatile("s_and_saveexec_b64 %1, %2\n\tglobal_load_dword %0, %3, off\n\ts_mov_b64 exec, %1" : "=&v"(v), "=&s"(sv) : "s"(m), "v"(p) : "memory", "scc");
    return v;
}
ZJ_DEV void zr_st32_m(ZRM m, void* p, u32 v) {
    u64 sv;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dword %2, %3, off\n\ts_mov_b64 exec, %0" : "=&s"(sv) : "s"(m), "v"(p), "v"(v) : "memory", "scc");
}
#define ZR_WAIT9(a, b, c, d, e, f, g, h, i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i) :: "memory")
#define ZR_WAIT2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#else
typedef bool ZRM;
#define ZRM_OF(c) ((bool)(c))
#define ZRM_NOT(m) (!(m))
#define ZRM_NONE false
#define ZRM_ANY(m) (m)
ZJ_DEV u64 zr_ld64_m(ZRM m, const u8* p) { return m ? ld64(p) : 0; }
ZJ_DEV u32 zr_ld32_m(ZRM m, const u8* p) { return m ? ld32(p) : 0; }
ZJ_DEV void zr_st32_m(ZRM m, void* p, u32 v) { if (m) *(u32*)p = v; }
#define ZR_WAIT9(a, b, c, d, e, f, g, h, i) ((void)0)
#define ZR_WAIT2(a, b) ((void)0)
#endif

template <class E, u32 JMAX = ZR_JMAX_DEFAULT>
struct ZLaneR {
    typedef typename E::T Ent;
    const u8* src; u32 n, ilimit; Ent* HL; Ent* HS; ZLHash hL, hS;
    const u8* F;                                      // the frame's flag bytes, or nullptr: every probe and every write is made
    ZEOut o;
    u32 st, cont, lastLL;
    u32 ip, ip1, anchor, off1, off2, step, nextStep, curr;
    u64 A, B, RA, RB, FA, FB;                         // windows: source at ip, source at ip - off1, flags at ip (valid in SEARCH)
    u64 C, RC, FC; u32 vw;                            // their third words and the number of valid bytes (16 .. 23, the same for the three streams): LAZY refill, below
    u64 w1, wIns;                                     // once a match is found at ip: the word at ip1 (SHORT_L1's comparison), the word at curr + 2 (first post-insert)
    u32 el0, es0, el1, hl0, hs0, tl0, hl1, tl1, fN1, fIns;   // entries of ip as the reference reads them; ip1's long bucket / tag / flags (fin), flags of curr + 2
    u32 ca, cb, acc;                                  // forward count in progress
    u32 mpos, mpos2, mLength, offset, bk, bk2;
    u32 more, more2, cvalid, needBack, needCand, chk;      // (0 / 1 in vector registers: a bool member lives as a lane mask and every divergent assignment is three scalar instructions)

    ZJ_DEV_MEMBER void init(const u8* s, u32 size, const ZEParams& p, u8* table, u8* fscratch, u32 maxSrc, const u8* flags = nullptr) {
        src = s; n = size; ilimit = size - 8u; hL = zl_hash_of(8, p.hashLog); hS = zl_hash_of(p.minMatch, p.chainLog);
        HL = (Ent*)table; HS = HL + (1u << p.hashLog);
        F = flags;
        o.seqs = (ZESeq*)fscratch; o.litOff = (u32*)(fscratch + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); o.n = 0; o.lit = 0;
        ip = 1; anchor = 0; off1 = 1; off2 = 0; chk = false; lastLL = size;
        needBack = needCand = more = more2 = cvalid = false;
        A = B = RA = RB = 0; FA = FB = 0x0F0F0F0F0F0F0F0FULL; C = RC = FC = 0; vw = 16u;
        st = ZL_LOADW;
    }
    ZJ_DEV_MEMBER u32 prod_long(u64 v) const { return zl_prod_hi(hL, v); }
    ZJ_DEV_MEMBER u32 idx_long(u32 p) const { return p >> hL.rsh; }
    ZJ_DEV_MEMBER u32 tag_long(u32 p) const { return (p >> (hL.rsh - 15u)) & 0x7FFFu; }
    ZJ_DEV_MEMBER void put_long_if(ZRM on, u64 v, u32 pos1) { u32 const p = prod_long(v); zr_st32_m(on, &HL[idx_long(p)], E::make(pos1, tag_long(p))); }
    ZJ_DEV_MEMBER void put_short_if(ZRM on, u64 v, u32 pos1) { zr_st32_m(on, &HS[zl_hash(hS, v)], E::make(pos1, ze_tag4((u32)v))); }
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
    ZJ_DEV_MEMBER void round(u32 r) {                 // see ZLaneD::round: search every round, the other states in turns (r: wave-uniform)
#ifdef ZR_ROUND_SWITCH                                /* four copies of the round, one per rotation slot (round 3's shape; A/B builds) */
        switch (r) {
        case 0: round_t<ZL_EN_COUNT>(ZL_EN_COUNT); break;
        case 1: round_t<ZL_EN_POST>(ZL_EN_POST); break;
        case 2: round_t<ZL_EN_START>(ZL_EN_START); break;
        default: round_t<0>(0); break;
        }
#else
        // ONE body for every rotation slot, the slot's states behind wave-uniform branches: four specialised copies cost ~35 register moves per round
        // where their register assignments meet again at the loop's back edge, and four times the instruction-cache footprint
        round_t<-1>(r == 0u ? (u32)ZL_EN_COUNT : (r == 1u ? (u32)ZL_EN_POST : (r == 2u ? (u32)ZL_EN_START : 0u)));
#endif
    }
    // One round, written flat (round 4).  The round of round 3 was ~1 220 wave-instructions (SQ counters: 784 VALU + 413 SALU + 21 VMEM), a third of
    // them exec-mask bookkeeping: every `if (v) r = ld64(..)` was a divergent region of its own, every per-state `if / else if` merged a dozen booleans
    // as lane masks (three SALU operations each), and at one wave per SIMD a scalar instruction costs its issue slot like a vector one.  Here
    //   * what a state asks for is worked out by selects, for every lane, without regions (a lane in another state computes garbage that no mask lets out);
    //   * every load and store is ONE predicated instruction — exec narrowed to the lanes that ask (zr_ld64_m / zr_st32_m: s_and_saveexec, the access,
    //     exec back) — and the round's single wait sits in the fence;
    //   * the clamp-and-fix arithmetic of frame-edge loads runs only in the rounds where some lane of the wave is within 8 bytes of an edge (ZR_ANY);
    //   * each state's consume step is one flat region, and the sequence record is written at one place for every kind of match.
    // States, loads, decisions and their order are round 3's (and the reference's): tests/test_emu_encode.py, tools/fuzz_emu_need.py.
    template <int KT>
    ZJ_DEV_MEMBER void round_t(u32 const KU) {
        u32 const K = KT >= 0 ? (u32)KT : KU;          // (compile-time in the switch build, a scalar otherwise)
        ZL_PROF_T0();
        u32 const s = st;
        bool const kC = (K & ZL_EN_COUNT) != 0u, kP = (K & ZL_EN_POST) != 0u, kT = (K & ZL_EN_START) != 0u;    // (wave-uniform: which states this rotation slot serves)
        bool const isS = s == ZL_SEARCH;
        bool const isC = kC && s == ZL_COUNT, isB = kC && s == ZL_BACK, isL1 = kC && s == ZL_SL1, isL2 = kC && s == ZL_SL2;
        bool const isP = kP && s == ZL_POST, isW = kP && s == ZL_LOADW, isF = kP && s == ZL_FAR;
        bool const isT = kT && s == ZL_START;
        bool const hasF = F != nullptr;
        ZRM const MS = ZRM_OF(s == ZL_SEARCH), MHASF = ZRM_OF(F != nullptr);
        u32 const nm8 = n - 8u;
        // ---- phase 1: the search state's view of ip (every lane computes it; only searching lanes act on it) ----
        u32 const fI = (u32)FA & 15u, ts = ze_tag4((u32)A);
        if (isS) { ZE_COUNT_ITER(); curr = ip; }
        zr_st32_m(MS & ZRM_OF((fI & 4u) != 0u), &HL[hl0], E::make(ip + 1u, tl0));
        zr_st32_m(MS & ZRM_OF((fI & 8u) != 0u), &HS[hs0], E::make(ip + 1u, ts));
        bool const ml0 = isS && E::maybe(el0, tl0), ms0 = isS && E::maybe(es0, ts);
        ZRM const ML0 = MS & ZRM_OF((el0 & 0x1FFFFu) != 0u) & ZRM_OF((el0 >> 17) == tl0), MS0 = MS & ZRM_OF((es0 & 0x1FFFFu) != 0u) & ZRM_OF((es0 >> 17) == ts);   // (E::maybe as masks)
        u32 const cl = E::pos(el0) - 1u, cs = E::pos(es0) - 1u;
        // repcode tests of ip + j, j = 0 .. 7: bytes j + 1 .. j + 4 of the source window against the repcode window
        u64 const y0 = zr_zero80(A ^ RA), y1 = zr_zero80(B ^ RB);
        u64 H = zr_shr(y0, y1, 1) & zr_shr(y0, y1, 2) & zr_shr(y0, y1, 3) & zr_shr(y0, y1, 4);          // 0x80 in byte j: ip + j has a repcode match at ip + j + 1
        H = off1 == 0u ? 0 : H;
        bool const rep0 = ((u32)H & 0x80u) != 0u;
        // quiet positions behind ip (step 1 only): no probe needed, no repcode match, still below nextStep and ilimit
        u64 const NP = ~zr_zero80(FA & 0x0303030303030303ULL) & 0x8080808080808080ULL;                     // 0x80 in byte j: ip + j needs a probe
        u64 const NQ = ((NP | H) >> 8) | (0x80ULL << 56);                                                  // byte j - 1: ip + j is not quiet (j = 1 .. 7); a stop at j = 8
        u32 const roomStep = nextStep - ip, lim0 = roomStep >= 2u ? roomStep - 2u : 0u, lim1 = ilimit - ip - 1u;
        u32 const Jq = zj_min(zj_min((u32)__builtin_ctzll(NQ) >> 3, JMAX), zj_min(lim0, lim1));
        u32 const J = step == 1u ? Jq : 0u, adv = step == 1u ? Jq + 1u : step;
        bool const far = adv > 8u;
        ZRM const MSN = MS & ZRM_OF(adv <= 8u);
        u64 const hwS = zr_ext(A, B, adv); u32 const fP = (u32)zr_ext(FA, FB, adv) & 15u;                  // the next position to decide, P = ip + adv: its word and flags (adv <= 8)
        // ---- what each state asks for: the word whose table entries are read, addresses, masks ----
        u64 hw = hwS; u32 fT = fP;
        ZRM MT0 = MSN, MT1 = MSN;                                   // lanes that read a long / short table entry this round (if the flags of the position say so)
        u32 const ro = ip - off1;
        // LAZY REFILL (round 4): the windows hold vw = 16 .. 23 valid bytes and a searching lane asks for the next 8 bytes of its three streams only when
        // fewer than 16 would be left after this round's advance — one request per 8 bytes of progress instead of one per round (a text frame advances one
        // byte per search round: 3 stream requests per round became 3 per 8 rounds; the kernel's time follows its request count, profiles/r04/c_slopes.txt).
        // The next unread byte of a stream sits vw bytes behind its window's start whatever the advance will be.
        u32 const vwn = vw - adv;                                   // valid bytes after the advance
        ZRM const MREF = MSN & ZRM_OF(vwn < 16u);
        u32 pa0 = ip + vw, pa1 = ro + vw, pa2 = cl, pa3 = cs, fa0 = ip + vw, fa1 = ip - 2u, bp0 = ip - bk, bp1 = mpos - bk;
        u32 hi = ip + vw;                                           // the highest forward address the lane's state reads (frame-edge test below)
        ZRM m0 = MREF, m1 = MREF, m2 = ML0, m3 = MS0, m4 = ZRM_NONE, mb = ZRM_NONE, mf0 = MREF, mf1 = ZRM_NONE, mOn = MREF;
        bool const second = cont == ZC_SHORT_L1;
        if (kC) {
            ZRM const MC = ZRM_OF(s == ZL_COUNT), MB = ZRM_OF(s == ZL_BACK), ML1 = ZRM_OF(s == ZL_SL1), ML2 = ZRM_OF(s == ZL_SL2);
            hw = isL2 ? w1 : hw; fT = isL2 ? fN1 : fT; MT0 = MT0 | ML2;
            pa0 = isC ? ca : (isL1 ? ip1 : pa0); m0 = m0 | MC | ML1;
            pa1 = isC ? ca + 8u : pa1; m1 = m1 | MC;
            pa2 = isC ? cb : pa2; m2 = m2 | MC;
            pa3 = isC ? cb + 8u : pa3; m3 = m3 | MC;
            m4 = MC & ZRM_OF(needCand);
            bp0 = isC ? (second ? ip1 : ip) : bp0; bp1 = isC ? (second ? mpos2 : mpos) : bp1; mb = (MC & ZRM_OF(needBack)) | MB;
            fa1 = isL1 ? ip1 : fa1; mf1 = ML1;
            hi = isC ? ca + 8u : hi; mOn = mOn | MC;
        }
        if (kP) {
            bool const wf = isW || isF;
            ZRM const MP = ZRM_OF(s == ZL_POST), MW = ZRM_OF(s == ZL_LOADW), MWF = MW | ZRM_OF(s == ZL_FAR);
            pa0 = wf ? ip : pa0; m0 = m0 | MWF;
            pa1 = isP ? ip - 2u : pa1; m1 = m1 | MP;
            pa2 = isP ? ip + 6u : pa2; m2 = m2 | MP;
            pa3 = (isP || isW) ? ip - off2 : pa3; m3 = m3 | ((MP | (MW & ZRM_OF(chk))) & ZRM_OF(off2 > 0u));
            fa0 = (isP || wf) ? ip : fa0; mf0 = mf0 | MP | MWF;
            mf1 = MP;
            hi = (isP || wf) ? ip + 6u : hi; mOn = mOn | MP | MWF;
        }
        if (kT) {
            ZRM const MT = ZRM_OF(s == ZL_START);
            hw = isT ? A : hw; fT = isT ? fI : fT; MT0 = MT0 | MT; MT1 = MT1 | MT;
            pa0 = isT ? ip + 8u : pa0; m0 = m0 | MT;
            pa1 = isT ? ro : pa1; m1 = m1 | MT;
            pa2 = isT ? ro + 8u : pa2; m2 = m2 | MT;
            fa0 = isT ? ip + 8u : fa0; mf0 = mf0 | MT;
            hi = isT ? ip + 8u : hi; mOn = mOn | MT;
        }
        u32 const hp = prod_long(hw), nhl = idx_long(hp), ntl = tag_long(hp), nhs = zl_hash(hS, hw);
        bool const vt0 = (fT & 1u) != 0u, vt1 = (fT & 2u) != 0u;                                          // (meaningful for the lanes of MT0 / MT1 only)
        MT0 = MT0 & ZRM_OF((fT & 1u) != 0u); MT1 = MT1 & ZRM_OF((fT & 2u) != 0u);
        mf0 = mf0 & MHASF; mf1 = mf1 & MHASF;
#ifdef ZR_COUNT_SLOT                                          /* analysis builds: which load slots a frame activates (tools, never the product) */
        ZR_COUNT_SLOT(0, m0); ZR_COUNT_SLOT(1, m1); ZR_COUNT_SLOT(2, m2); ZR_COUNT_SLOT(3, m3); ZR_COUNT_SLOT(4, m4); ZR_COUNT_SLOT(5, mb);
        ZR_COUNT_SLOT(6, mf0); ZR_COUNT_SLOT(7, mf1); ZR_COUNT_SLOT(8, MT0); ZR_COUNT_SLOT(9, MT1);
#endif
        // ---- phase 2: one batch of predicated loads ----
        ZL_PROF_T1();
        u32 const q0 = zj_min(pa0, nm8), q1 = zj_min(pa1, nm8), q2 = zj_min(pa2, nm8), q3 = zj_min(pa3, nm8), q4 = zj_min(mpos2, nm8), qf0 = zj_min(fa0, nm8);
        u32 const qb0 = bp0 >= 8u ? bp0 - 8u : 0u, qb1 = bp1 >= 8u ? bp1 - 8u : 0u;
        u64 d0 = zr_ld64_m(m0, src + q0), d1 = zr_ld64_m(m1, src + q1), d2 = zr_ld64_m(m2, src + q2), d3 = zr_ld64_m(m3, src + q3);
        u32 t0 = zr_ld32_m(MT0, (const u8*)&HL[nhl]), t1 = zr_ld32_m(MT1, (const u8*)&HS[nhs]);
        u64 g0 = zr_ld64_m(mf0, F + qf0);
        u32 g1 = 0; u64 d4 = 0, b0 = 0, b1 = 0;
        if (kC || kP) g1 = zr_ld32_m(mf1, F + fa1);                                                        // (fa1 + 4 <= n - 4: never clamped)
        if (kC) { d4 = zr_ld64_m(m4, src + q4); b0 = zr_ld64_m(mb, src + qb0); b1 = zr_ld64_m(mb, src + qb1); }
#ifdef ZR_EXTRA_LOADS                                         /* measurement builds only: N extra stream-like requests per search round (new lines, results unused) — the slope of the kernel time over the request count */
        u64 x0 = zr_ld64_m(m0, src + zj_min(q0 ^ 0x2000u, nm8)), x1 = 0, x2 = 0;
        if (ZR_EXTRA_LOADS > 1) x1 = zr_ld64_m(m1, src + zj_min(q1 ^ 0x4000u, nm8));
        if (ZR_EXTRA_LOADS > 2) x2 = zr_ld64_m(m0, src + zj_min(q0 ^ 0x6000u, nm8));
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
#endif
#if ZR_SHADOW
        // ---- in the SHADOW of the loads (round 5): a searching lane whose entries name no candidate (no tag agrees) and whose repcode test fails cannot
        // match at ip whatever this round's loads bring, so everything its "no match" step does — the quiet run's table writes, the advance to P, the
        // slide of the three windows — is worked out while the loads fly instead of after the wait; what the step needs FROM the loads (P's entries,
        // the refill words) is merged in behind the wait.  A lane with a candidate decides behind the wait as before; if the candidate's bytes differ
        // it only forgets the candidate and takes this path in the next round (tags are 15 bits: once in ~30 000 probes).
        bool const sure = isS && !rep0 && !far && !(ml0 || ms0);
        u32 f0v = 0, f1v = 0, k8S = 0; bool refS = false;
        if (sure) {
ZR_UNROLL
            for (u32 j = 1; j <= JMAX; j++) {
                if (!ZR_ANY(j <= J)) break;               // (wave-uniform: no lane's run is this long)
                bool const inRun = j <= J;
                u64 const wq = zr_shr(A, B, j); u32 const fq = (u32)(FA >> (8u * j));
                u32 const pq = prod_long(wq), bl = idx_long(pq), bs = zl_hash(hS, wq);
                u32 const eL = (u32)E::make(ip + j + 1u, tag_long(pq)), eS = (u32)E::make(ip + j + 1u, ze_tag4((u32)wq));
                bool const wL = inRun && (fq & 4u) != 0u, wS = inRun && (fq & 8u) != 0u;
                ZRM const RUN = ZRM_OF(j <= J);
                zr_st32_m(RUN & ZRM_OF((fq & 4u) != 0u), &HL[bl], eL); zr_st32_m(RUN & ZRM_OF((fq & 8u) != 0u), &HS[bs], eS);
                f0v = (wL && bl == nhl && vt0) ? eL : f0v; f1v = (wS && bs == nhs && vt1) ? eS : f1v;          // a write into the bucket this round already asked for is forwarded
            }
            if (ip + step >= nextStep) { step++; nextStep += 256u; }       // (only ever with J == 0)
            ip += adv;
            refS = vwn < 16u; k8S = (vwn & 7u) * 8u;                        // (refill: vwn = 8 .. 15, the new bytes land at byte vwn — behind the wait)
            u64 const sC = adv >= 8u ? 0 : (C >> (8u * adv)), sRC = adv >= 8u ? 0 : (RC >> (8u * adv)), sFC = adv >= 8u ? 0 : (FC >> (8u * adv));
            A = zr_ext(A, B, adv); B = zr_ext(B, C, adv); C = sC;
            RA = zr_ext(RA, RB, adv); RB = zr_ext(RB, RC, adv); RC = sRC;
            FA = zr_ext(FA, FB, adv); FB = zr_ext(FB, FC, adv); FC = sFC;
            vw = refS ? vwn + 8u : vwn;
            hl0 = nhl; hs0 = nhs; tl0 = ntl;
            if (ip + step > ilimit) finish();
        }
#endif
        ZR_WAIT9(d0, d1, d2, d3, d4, b0, b1, t0, t1);
        ZR_WAIT2(g0, g1);
#ifdef ZR_EXTRA_LOADS
        asm volatile("" :: "v"(x0), "v"(x1), "v"(x2));
#endif
        ZL_PROF_T2();
#ifdef ZR_EXTRA_VALU                                          /* measurement builds only: N dependent vector instructions per round — the slope of the kernel time over the instruction count */
        {   u32 zx = ip; asm volatile(".rept %1\n\tv_add_u32 %0, %0, 1\n\t.endr" : "+v"(zx) : "n"(ZR_EXTRA_VALU)); asm volatile("" :: "v"(zx)); }
#endif
        // A load that was clamped at the frame's end (or start) is shifted into place — only in rounds where some lane of the wave is that close to an edge:
        // `hi` is the highest forward address the lane's state reads (candidates and repcode sources lie below ip), a backward load is clamped below 8.
        if (ZRM_ANY(mOn & ZRM_OF(hi > nm8))) {
            if (pa0 > nm8) d0 = zl_fwd_fix(d0, pa0, q0);
            if (pa1 > nm8) d1 = zl_fwd_fix(d1, pa1, q1);
            if (pa2 > nm8) d2 = zl_fwd_fix(d2, pa2, q2);
            if (pa3 > nm8) d3 = zl_fwd_fix(d3, pa3, q3);
            if (fa0 > nm8) g0 = zl_fwd_fix(g0, fa0, qf0);
        }
        if (kC) {
            if (ZRM_ANY(mb & (ZRM_OF(bp0 < 8u) | ZRM_OF(bp1 < 8u)))) { b0 = zl_back_fix(b0, bp0, qb0); b1 = zl_back_fix(b1, bp1, qb1); }
        }
        if (!hasF) { g0 = 0x0F0F0F0F0F0F0F0FULL; g1 = 0x0F0F0F0Fu; }                                       // a frame without flags: all set
        // ---- phase 3: consume ----
        u32 emit = 0; u32 eLL = 0, eOff = 0, eML = 0;                                                 // the sequence this round completes, if any (stored once, below)
        u32 doFin = 0, doLoop = 0;                                                                // a long / short match became final (COUNT or BACK); an immediate repcode match was counted
        if (isS) {
            bool const isLong = ml0 && d2 == A, isShort = !isLong && ms0 && (u32)d3 == (u32)A;
            bool const found = rep0 || isLong || isShort;
            bool const near1 = !far && J == 0u;                        // P is the reference's ip1 (ip + step): its long entry, bucket and tag are this round's
            if (found) {
                wIns = zr_ext(A, B, 2u); fIns = (u32)(FA >> 16) & 15u;                                     // (leave_search)
                if (!rep0) {
                    ip1 = ip + step;
                    if (near1) { w1 = hw; hl1 = nhl; tl1 = ntl; fN1 = fP; el1 = vt0 ? t0 : 0u; }
                    else if (!far) { w1 = zr_ext(A, B, 1u); u32 const p1 = prod_long(w1); hl1 = idx_long(p1); tl1 = tag_long(p1); fN1 = (u32)(FA >> 8) & 15u; el1 = 0; }   // ip + 1 is quiet: its probe cannot match
                    mpos = isLong ? cl : cs;
                    if (isLong && far) fN1 = 0;                          // (step > 8: fin() writes nothing for ip1)
                }
                u32 const skip = rep0 ? 5u : (isLong ? 8u : 4u);
                ca = ip + skip; cb = (rep0 ? ro : mpos) + skip; acc = 0;
                cont = rep0 ? ZC_REP1 : (isLong ? ZC_LONG : ZC_SHORT);
                needBack = !rep0;
                bool const shortNear = isShort && !rep0 && !far;
                needCand = shortNear && (E::pos(el1) > 1u) && E::maybe(el1, tl1);
                if (shortNear) { mpos2 = E::pos(el1) - 1u; cvalid = false; }
                st = (isShort && !rep0 && far) ? (u32)ZL_SL1 : (u32)ZL_COUNT;      // (far short match: ip1's word, flags and long entry are not here — two more states fetch them)
            } else if (far) {
                // no match, and the next position is out of the windows' reach: reload there
                if (ip + step >= nextStep) { step++; nextStep += 256u; ip += step - 1u; } else ip += step;
                if (ip + step > ilimit) finish(); else st = ZL_FAR;
            }
#if ZR_SHADOW
            else if (sure) {                                   // the step was taken in the loads' shadow: P's entries (or a forwarded write) and the refill words
                el0 = f0v ? f0v : (vt0 ? t0 : 0u); es0 = f1v ? f1v : (vt1 ? t1 : 0u);
                u64 const hiSh = (64u - k8S) & 63u;
                B = refS ? (B | (d0 << k8S)) : B; C = refS ? (k8S ? (d0 >> hiSh) : 0) : C;
                RB = refS ? (RB | (d1 << k8S)) : RB; RC = refS ? (k8S ? (d1 >> hiSh) : 0) : RC;
                FB = refS ? (FB | (g0 << k8S)) : FB; FC = refS ? (k8S ? (g0 >> hiSh) : 0) : FC;
            } else { el0 = 0; es0 = 0; }                       // a candidate whose bytes differ: forgotten; the position takes the path above in the next round
#else
            else {
                // no match at ip: commit the quiet run behind it, move to P with the entries this round fetched
                u32 e0 = vt0 ? t0 : 0u, e1 = vt1 ? t1 : 0u;
ZR_UNROLL
                for (u32 j = 1; j <= JMAX; j++) {
                    if (!ZR_ANY(j <= J)) break;               // (wave-uniform: no lane's run is this long)
                    bool const inRun = j <= J;
                    u64 const wq = zr_shr(A, B, j); u32 const fq = (u32)(FA >> (8u * j));
                    u32 const pq = prod_long(wq), bl = idx_long(pq), bs = zl_hash(hS, wq);
                    u32 const eL = (u32)E::make(ip + j + 1u, tag_long(pq)), eS = (u32)E::make(ip + j + 1u, ze_tag4((u32)wq));
                    bool const wL = inRun && (fq & 4u) != 0u, wS = inRun && (fq & 8u) != 0u;
                    ZRM const RUN = ZRM_OF(j <= J);
                    zr_st32_m(RUN & ZRM_OF((fq & 4u) != 0u), &HL[bl], eL); zr_st32_m(RUN & ZRM_OF((fq & 8u) != 0u), &HS[bs], eS);
                    e0 = (wL && bl == nhl && vt0) ? eL : e0; e1 = (wS && bs == nhs && vt1) ? eS : e1;          // a write into the bucket this round already asked for is forwarded
                }
                if (ip + step >= nextStep) { step++; nextStep += 256u; }       // (only ever with J == 0)
                ip += adv;
                // slide the three windows by adv bytes (bytes beyond the valid ones are zero and stay zero), then put the 8 bytes that were asked for behind them
                {   bool const ref = vwn < 16u; u32 const k8 = (vwn & 7u) * 8u;                     // (refill: vwn = 8 .. 15, the new bytes land at byte vwn)
                    u64 const sC = adv >= 8u ? 0 : (C >> (8u * adv)), sRC = adv >= 8u ? 0 : (RC >> (8u * adv)), sFC = adv >= 8u ? 0 : (FC >> (8u * adv));
                    A = zr_ext(A, B, adv); B = zr_ext(B, C, adv);
                    RA = zr_ext(RA, RB, adv); RB = zr_ext(RB, RC, adv);
                    FA = zr_ext(FA, FB, adv); FB = zr_ext(FB, FC, adv);
                    u64 const hiSh = (64u - k8) & 63u;
                    B = ref ? (B | (d0 << k8)) : B; C = ref ? (k8 ? (d0 >> hiSh) : 0) : sC;
                    RB = ref ? (RB | (d1 << k8)) : RB; RC = ref ? (k8 ? (d1 >> hiSh) : 0) : sRC;
                    FB = ref ? (FB | (g0 << k8)) : FB; FC = ref ? (k8 ? (g0 >> hiSh) : 0) : sFC;
                    vw = ref ? vwn + 8u : vwn;
                }
                el0 = e0; es0 = e1; hl0 = nhl; hs0 = nhs; tl0 = ntl;
                if (ip + step > ilimit) finish();
            }
#endif
        }
        if ((K & ZL_EN_COUNT) && isC) {
            u32 const lim = n - ca;
            u32 c = zl_common_fwd16(d0, d1, d2, d3);
            c = zj_min(c, lim);
            acc += c;
            if (needBack) {
                u32 const limit = second ? zj_min(ip1 - anchor, mpos2) : zj_min(ip - anchor, mpos);
                u32 const e = zj_min(zl_common_back8(b0, b1), limit);
                bool const m = (e == 8u) && (limit > 8u);
                if (second) { bk2 = e; more2 = m; } else { bk = e; more = m; }
                needBack = false;
            }
            if (needCand) { cvalid = (d4 == w1); needCand = false; }
            if (c == 16u && lim > 16u) { ca += 16u; cb += 16u; }
            else if (cont == ZC_REP1) {
                mLength = acc + 4u; ip += 1u;
                emit = 1u; eLL = ip - anchor; eOff = 1u; eML = mLength;
            } else if (cont == ZC_SHORT && cvalid) {
                mLength = acc + 4u; offset = ip - mpos;
                ca = ip1 + 8u; cb = mpos2 + 8u; acc = 0; cont = ZC_SHORT_L1; needBack = true;
            } else if (cont == ZC_REPLOOP) {                   // immediate repcode after a match (A = the word at ip, FA = its flags)
                u32 const rLength = acc + 4u;
                { u32 const t = off2; off2 = off1; off1 = t; }
                put_short_if(ZRM_OF(((u32)FA & 8u) != 0u), A, ip + 1u);
                put_long_if(ZRM_OF(((u32)FA & 4u) != 0u), A, ip + 1u);
                emit = 1u; doLoop = 1u; eLL = 0u; eOff = 1u; eML = rLength;
            } else {
                if (cont == ZC_LONG) { mLength = acc + 8u; offset = ip - mpos; }
                else if (cont == ZC_SHORT) { mLength = acc + 4u; offset = ip - mpos; }
                else {                                         // ZC_SHORT_L1: the long match at ip1 against the short one at ip
                    u32 const l1len = acc + 8u;
                    if (l1len > mLength) { ip = ip1; mLength = l1len; mpos = mpos2; offset = ip - mpos; bk = bk2; more = more2; }
                }
                if (more) st = ZL_BACK; else doFin = 1u;
            }
        }
        if ((K & ZL_EN_COUNT) && isB) {
            u32 const limit = zj_min(ip - anchor, mpos) - bk;
            u32 const e = zj_min(zl_common_back8(b0, b1), limit);
            bk += e; more = (e == 8u) && (limit > 8u);
            doFin = more ? 0u : 1u;
        }
        if ((K & ZL_EN_COUNT) && isL1) { w1 = d0; fN1 = g1 & 15u; st = ZL_SL2; }
        if ((K & ZL_EN_COUNT) && isL2) {
            el1 = vt0 ? t0 : 0u; hl1 = nhl; tl1 = ntl;            // (read after ip's own writes, as the reference reads it)
            ca = ip + 4u; cb = mpos + 4u; acc = 0; cont = ZC_SHORT; st = ZL_COUNT; needBack = true;
            needCand = (E::pos(el1) > 1u) && E::maybe(el1, tl1); mpos2 = E::pos(el1) - 1u; cvalid = false;
        }
        if (K & ZL_EN_COUNT) {
            if (doFin) {                                       // a long / short match is final: apply the backward extension (fin)
                ip -= bk; mLength += bk;
                off2 = off1; off1 = offset;
                zr_st32_m(ZRM_OF(step < 4u) & ZRM_OF((fN1 & 4u) != 0u), &HL[hl1], E::make(ip1 + 1u, tl1));      // (ip1: the position that was ip1 when the match was found)
                emit = 1u; eLL = ip - anchor; eOff = offset + 3u; eML = mLength;
            }
            if (emit) {
                ze_store(o, anchor, eLL, eOff, eML);
                ip += eML; anchor = ip;
                if (ip <= ilimit) { chk = chk | doLoop; st = doLoop ? (u32)ZL_LOADW : (u32)ZL_POST; }
                else finish();
            }
        }
        if ((K & ZL_EN_POST) && isP) {
            u64 const q0w = d1, q1w = d2;
            u64 const wb = q0w, wc = (q0w >> 8) | (q1w << 56);
            u32 const ins = curr + 2u;
            // fIns = flags of curr + 2, g1 = flags of ip - 2, ip - 1, ip, ip + 1 (one byte each)
            put_long_if(ZRM_OF((fIns & 4u) != 0u), wIns, ins + 1u);
            put_long_if(ZRM_OF((g1 & 4u) != 0u), wb, ip - 2u + 1u);
            put_short_if(ZRM_OF((fIns & 8u) != 0u), wIns, ins + 1u);
            put_short_if(ZRM_OF((g1 & 0x800u) != 0u), wc, ip - 1u + 1u);
            A = (q0w >> 16) | (q1w << 48); FA = g0;
            if ((off2 > 0u) && ((u32)A == (u32)d3)) { ca = ip + 4u; cb = ip + 4u - off2; acc = 0; cont = ZC_REPLOOP; st = ZL_COUNT; needBack = false; needCand = false; }
            else outer();
        }
        if ((K & ZL_EN_POST) && isW) {
            A = d0; FA = g0;
            if (chk && (off2 > 0u) && ((u32)A == (u32)d3)) { ca = ip + 4u; cb = ip + 4u - off2; acc = 0; cont = ZC_REPLOOP; st = ZL_COUNT; needBack = false; needCand = false; }
            else outer();
            chk = false;
        }
        if ((K & ZL_EN_POST) && isF) { A = d0; FA = g0; st = ZL_START; }
        if ((K & ZL_EN_START) && isT) {
            el0 = vt0 ? t0 : 0u; es0 = vt1 ? t1 : 0u; hl0 = nhl; hs0 = nhs; tl0 = ntl;
            B = d0; RA = d1; RB = d2; FB = g0; C = RC = FC = 0; vw = 16u;
            st = ZL_SEARCH;
        }
    }
};
