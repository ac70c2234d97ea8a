// tests/emu/emu.cpp — lane-serial (W = 1) build of the kernel bodies in zstd-jni_amd/csrc for
// pre-GPU unit tests in the CPU-only dev container.  TEST INFRASTRUCTURE ONLY: never linked into
// libzjni_amd.so, never reachable from the C-ABI (which fails loudly without a GPU).
#include "../../zstd-jni_amd/csrc/zj_decode.h"
#include "../../zstd-jni_amd/csrc/zj_decode_split.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

// -DEMU_EXACT (the sanitizer build): every entry point works on exact-size heap copies of the caller's source and destination, so
// that a read past the last source byte or a write past the capacity lands in an AddressSanitizer red zone.
#ifdef EMU_EXACT
struct EmuExact {
    u8* s; u8* d; u8* userDst; unsigned cap;
    EmuExact(const unsigned char*& src, unsigned n, unsigned char*& dst, unsigned c) {
        s = (u8*)malloc(n ? n : 1); if (n) memcpy(s, src, n);
        d = (u8*)malloc(c ? c : 1); if (c) memcpy(d, dst, c);
        userDst = dst; cap = c; src = s; dst = d;
    }
    ~EmuExact() { if (cap) memcpy(userDst, d, cap); free(s); free(d); }
};
#define EMU_IO(src, n, dst, c) EmuExact emuExact_(src, n, dst, c)
#else
#define EMU_IO(src, n, dst, c) do {} while (0)
#endif

// The decode kernels are persistent too: `sh` (LDS) and the literal scratch slot outlive a frame and nothing clears them.  One
// poisoned set per process instead of a calloc per frame, so that a decoder which relies on leftovers fails here.
static ZDecShared* emu_dec_sh() {
    static ZDecShared* sh = nullptr;
    if (!sh) { sh = (ZDecShared*)malloc(sizeof(ZDecShared)); memset(sh, 0xA5, sizeof(ZDecShared)); }
    return sh;
}
static u8* emu_dec_lit() {
    static u8* lit = nullptr;
    if (!lit) { lit = (u8*)malloc(ZD_LIT_SCRATCH); memset(lit, 0x3C, ZD_LIT_SCRATCH); }
    return lit;
}
extern "C" unsigned long long emu_decompress(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap) {
    EMU_IO(src, srcSize, dst, dstCap);
    Grp<1> g;
    ZDecShared* sh = emu_dec_sh();
    u8* lit = emu_dec_lit();
    ZjProf pf; pf.start(nullptr);
    u64 r = zd_decompress(g, *sh, src, srcSize, dst, dstCap, lit, pf);
    return r;
}
// zd_huf_streams_wave's counters (zj_decode.h): taken, left at the check, left after the tries, repeated passes, lanes that decoded again
extern "C" void emu_hp_stats(unsigned long long* out5, int reset) { for (int i = 0; i < 5; i++) { out5[i] = zd_hp_stats[i]; if (reset) zd_hp_stats[i] = 0; } }
// split pipeline: prep -> lane sequence decode -> execute; frames the pipeline hands over go through the fused path
extern "C" unsigned long long emu_decompress_split(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, int* usedSplit) {
    EMU_IO(src, srcSize, dst, dstCap);
    Grp<1> g;
    ZDecShared* sh = (ZDecShared*)malloc(sizeof(ZDecShared)); memset(sh, 0xA5, sizeof(ZDecShared));      // a kernel's LDS is whatever the previous frame / kernel left
    u8* lit = (u8*)malloc(ZD_LIT_SCRATCH);
    u16* tab = (u16*)calloc(ZD_SPLIT_CELLS, 2); u64* seqs = (u64*)malloc(ZD_SPLIT_SEQ_BYTES); ZDMeta meta;
    ZjProf pf; pf.start(nullptr);
    u64 r = ~(u64)0;
    if (usedSplit) *usedSplit = 0;
    bool const simple = zd_prep_frame(g, *sh, src, srcSize, dstCap, tab, &meta);
    if (!simple) {                                               // zj_dec_prep_kernel: a frame of one stored block is copied by stage 1 itself (round 6); usedSplit = 4
        u64 res = 0;
        if (zd_prep_frame_stored(g, *sh, src, srcSize, dst, dstCap, &res)) { if (usedSplit) *usedSplit = 4; free(seqs); free(tab); free(lit); free(sh); return res; }
    }
    if (simple) {
        u32 symL[36], symM[53]; zd_seq_symtabs(symL, symM, 0, 1); ZDSeqLane m; m.llBase = symL; m.mlBase = symM; m.init(src, tab, seqs, &meta);
        while (m.st != 2) m.round();
        // stage 2b as the kernels run it: its own workgroup state (poisoned), a slot per frame, the frame record marked
        u32 const slot = getenv("EMU_LIT_SLOT") ? (u32)atoi(getenv("EMU_LIT_SLOT")) : 65536u;
        u8* slotBuf = slot ? (u8*)malloc(slot) : nullptr;
        if (slotBuf) {
            ZDecShared* sh2 = (ZDecShared*)malloc(sizeof(ZDecShared)); memset(sh2, 0x3C, sizeof(ZDecShared));
            u8* hp = (u8*)malloc(ZD_HP_LDS); memset(hp, 0x77, ZD_HP_LDS);
            if (zd_lit_frame(g, *sh2, src, &meta, slotBuf, slot, pf, getenv("EMU_NO_HP") ? nullptr : hp)) meta.pad = 1u;
            free(hp);
            free(sh2);
        }
        {   u8* hp = (u8*)malloc(ZD_HP_LDS); memset(hp, 0x78, ZD_HP_LDS);
            r = zd_exec_frame(g, *sh, src, dst, &meta, seqs, lit, pf, nullptr, nullptr, slotBuf, slot, getenv("EMU_NO_HP") ? nullptr : hp);
            free(hp); }
        if (usedSplit && r != ~(u64)0) *usedSplit = meta.pad ? 3 : 1;
        free(slotBuf);
    }
    if (r == ~(u64)0) { memset(sh, 0x5A, sizeof(*sh)); r = zd_decompress(g, *sh, src, srcSize, dst, dstCap, lit, pf); }
    free(seqs); free(tab); free(lit); free(sh);
    return r;
}
// multi-block frames (and frames without a content size) through the split pipeline's block stages: zd_prep_frame_multi -> ZDSeqLaneT<true> per block ->
// zd_exec_frame_multi; frames the stages hand over go through the fused path.  *used = 1 when the block stages served the frame.
extern "C" unsigned long long emu_decompress_mb(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned long long dstCap, int* used) {
    EMU_IO(src, srcSize, dst, (unsigned)(dstCap > 0xFFFFFFFFull ? 0xFFFFFFFFull : dstCap));
    Grp<1> g;
    ZDecShared* sh = (ZDecShared*)malloc(sizeof(ZDecShared)); memset(sh, 0xA5, sizeof(ZDecShared));
    u8* lit = (u8*)malloc(ZD_LIT_SCRATCH);
    u32 const blkCap = ZD_MB_MAX_BLOCKS; u64 const seqCap = (u64)blkCap * 4096u + (u64)srcSize;      // (a sequence takes at least a few bits: far more than the frame can hold)
    ZDBlk* blks = (ZDBlk*)calloc(blkCap, sizeof(ZDBlk)); u16* tabs = (u16*)calloc((size_t)blkCap * ZD_SPLIT_CELLS, 2);
    u64* pool = (u64*)malloc((size_t)seqCap * 8); u32* seqList = (u32*)calloc(blkCap, 4);
    u32 blkCounter = 0, seqListCount = 0, litListCount = 0; unsigned long long seqCounter = 0, litCounter = 0; ZDFrameMB fr;
    // EMU_MB_LIT: 0 no stage 2b (stage 3 decodes every block's literals), 1 (default) every block with a slot, 2 every other one (treeless blocks meet both cases)
    int const litMode = getenv("EMU_MB_LIT") ? atoi(getenv("EMU_MB_LIT")) : 1;
    u64 const litCap = litMode ? (u64)srcSize * 40u + ((u64)blkCap << 17) : 0;
    u8* litPool = litMode ? (u8*)malloc((size_t)(litCap > ((u64)1 << 30) ? ((u64)1 << 30) : litCap) + 64) : nullptr;
    u64 const litCapUsed = litMode ? (litCap > ((u64)1 << 30) ? ((u64)1 << 30) : litCap) : 0;
    u32* litList = (u32*)calloc(blkCap, 4);
    ZjProf pf; pf.start(nullptr);
    u64 r = ~(u64)0;
    if (used) *used = 0;
    {   u64 res = 0;                                             // (stage 1 tries the stored-block copy before the block stages, as the kernel does)
        if (zd_prep_frame_stored(g, *sh, src, srcSize, dst, dstCap, &res)) { if (used) *used = 2; free(litList); free(litPool); free(seqList); free(pool); free(tabs); free(blks); free(lit); free(sh); return res; } }
    if (zd_prep_frame_multi(g, *sh, src, srcSize, dstCap, 0u, &fr, blks, tabs, &blkCounter, blkCap, &seqCounter, seqCap, seqList, &seqListCount, 1u,
                            &litCounter, litCapUsed, litMode ? litList : nullptr, &litListCount)) {
        for (u32 q = 0; q < litListCount; q++) {                  // stage 2b: a workgroup of its own per block (poisoned LDS)
            if (litMode == 2 && (q & 1u)) continue;
            ZDecShared* sh3 = (ZDecShared*)malloc(sizeof(ZDecShared)); memset(sh3, 0x77, sizeof(ZDecShared));
            u8* hp = (u8*)malloc(ZD_HP_LDS); memset(hp, 0x79, ZD_HP_LDS);
            if (zd_lit_block(g, *sh3, src, blks, litList[q], litPool, pf, getenv("EMU_NO_HP") ? nullptr : hp)) blks[litList[q]].litReady = 1u;
            free(hp);
            free(sh3);
        }
        u32 symL[36], symM[53]; zd_seq_symtabs(symL, symM, 0, 1);
        for (u32 q = seqListCount; q-- > 0; ) {                   // (any order: the blocks do not depend on each other here)
            ZDBlk* const bk = blks + seqList[q];
            ZDSeqLaneT<true> m; m.llBase = symL; m.mlBase = symM;
            m.init_block(src, tabs + (size_t)seqList[q] * ZD_SPLIT_CELLS, pool + (((u64)bk->seqHi << 32) | bk->seqLo), bk);
            while (m.st != 2) m.round();
        }
        ZDecShared* sh2 = (ZDecShared*)malloc(sizeof(ZDecShared)); memset(sh2, 0x3C, sizeof(ZDecShared));      // stage 3 is another kernel: its own LDS
        r = zd_exec_frame_multi(g, *sh2, src, dst, dstCap, &fr, blks, pool, lit, (u8*)sh2->ll, pf, litPool);
        free(sh2);
        if (used && r != ~(u64)0) *used = 1;
    }
    if (r == ~(u64)0) { memset(sh, 0x5A, sizeof(*sh)); r = zd_decompress(g, *sh, src, srcSize, dst, (u32)(dstCap > 0xFFFFFFFFull ? 0xFFFFFFFFull : dstCap), lit, pf); }
    free(litList); free(litPool); free(seqList); free(pool); free(tabs); free(blks); free(lit); free(sh);
    return r;
}
// dictionary decode: digest (ZSTD_createDDict) + ZSTD_decompress_usingDDict
// dictionary frames through the three-stage pipeline
extern "C" unsigned long long emu_decompress_split_dict(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap,
                                                        const unsigned char* dict, unsigned dictSize, int* usedSplit) {
    EMU_IO(src, srcSize, dst, dstCap);
    Grp<1> g;
    ZDecShared* sh = (ZDecShared*)malloc(sizeof(ZDecShared)); memset(sh, 0xA5, sizeof(ZDecShared));      // a kernel's LDS is whatever the previous frame / kernel left
    u8* lit = (u8*)malloc(ZD_LIT_SCRATCH);
    u16* tab = (u16*)calloc(ZD_SPLIT_CELLS, 2); u64* seqs = (u64*)malloc(ZD_SPLIT_SEQ_BYTES); ZDMeta meta;
    ZDDictDev* dd = (ZDDictDev*)calloc(1, sizeof(ZDDictDev));
    ZjProf pf; pf.start(nullptr);
    u64 r = ~(u64)0;
    if (usedSplit) *usedSplit = 0;
    zd_ddict_digest(g, *sh, dict, dictSize, dd);
    if (dd->status) r = ZJ_ERR64(dd->status);
    else {
        memset(sh, 0x5A, sizeof(*sh));
        if (zd_prep_frame<true>(g, *sh, src, srcSize, dstCap, tab, &meta, dd)) {
            u32 symL[36], symM[53]; zd_seq_symtabs(symL, symM, 0, 1); ZDSeqLane m; m.llBase = symL; m.mlBase = symM; m.init(src, tab, seqs, &meta, dd);
            while (m.st != 2) m.round();
            r = zd_exec_frame<true>(g, *sh, src, dst, &meta, seqs, lit, pf, dd, dict);
            if (usedSplit && r != ~(u64)0) *usedSplit = 1;
        }
        if (r == ~(u64)0) { memset(sh, 0x5A, sizeof(*sh)); r = zd_decompress<true>(g, *sh, src, srcSize, dst, dstCap, lit, pf, dd, dict); }
    }
    free(dd); free(seqs); free(tab); free(lit); free(sh);
    return r;
}
extern "C" unsigned long long emu_decompress_dict(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap,
                                                  const unsigned char* dict, unsigned dictSize) {
    EMU_IO(src, srcSize, dst, dstCap);
    Grp<1> g;
    ZDecShared* sh = (ZDecShared*)malloc(sizeof(ZDecShared)); memset(sh, 0xA5, sizeof(ZDecShared));      // a kernel's LDS is whatever the previous frame / kernel left
    u8* lit = (u8*)malloc(ZD_LIT_SCRATCH);
    ZDDictDev* dd = (ZDDictDev*)calloc(1, sizeof(ZDDictDev));
    ZjProf pf; pf.start(nullptr);
    zd_ddict_digest(g, *sh, dict, dictSize, dd);
    u64 r;
    if (dd->status) r = ZJ_ERR64(dd->status);
    else { memset(sh, 0x5A, sizeof(*sh)); r = zd_decompress<true>(g, *sh, src, srcSize, dst, dstCap, lit, pf, dd, dict); }
    free(dd); free(lit); free(sh);
    return r;
}
extern "C" unsigned emu_dec_shared_bytes() { return (unsigned)sizeof(ZDecShared); }

#include "../../zstd-jni_amd/csrc/zj_encode.h"
// The encode kernels are persistent: a workgroup's `sh` (static LDS), its dynamic LDS and its HBM scratch slot outlive the frame, and
// nothing clears them between frames.  The emulation keeps one such workgroup for the life of the process — poisoned at "launch",
// with the few fields a kernel sets before its first frame — so that state leaking from one frame into the next shows up here
// (a stale sh.litMode did, on the GPU only, while every frame here started from calloc).
struct EmuWg { ZEncShared* sh; u8* lds; u8* ws; };
static EmuWg& emu_wg() {
    static EmuWg w = { nullptr, nullptr, nullptr };
    if (!w.sh) {
        w.sh = (ZEncShared*)malloc(sizeof(ZEncShared)); memset(w.sh, 0xA5, sizeof(ZEncShared));
        w.sh->dictLoaded = 0; w.sh->ctDict[0] = 0; w.sh->ctDict[1] = 0; w.sh->ctDict[2] = 0;        // zj_encode_kernel / zj_encode_multi_kernel, before the frame loop
        w.lds = (u8*)malloc(160 * 1024); memset(w.lds, 0x5A, 160 * 1024);
        w.ws = (u8*)malloc(ZE_SCRATCH_BYTES); memset(w.ws, 0xC3, ZE_SCRATCH_BYTES);
    }
    return w;
}
extern "C" unsigned long long emu_compress(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level) {
    EMU_IO(src, srcSize, dst, dstCap);
    Grp<1> g;
    EmuWg& wg = emu_wg(); ZEncShared* sh = wg.sh; u8* lds = wg.lds; u8* ws = wg.ws;
    ZjProf pf; pf.start(nullptr);
    // level & 0xFF = level, bits 8-10 = frame flags (ZE_FLAG_CHECKSUM, ZE_FLAG_NO_FCS, ZE_FLAG_NO_DICTID)
    u64 r = (srcSize > ZE_BLOCK_MAX) ? ZJ_ERR64(201) : ze_compress(g, *sh, lds, src, srcSize, dst, dstCap, level & 0xFFu, ws, pf, nullptr, (level >> 8) & ZE_FLAG_MASK);
    return r;
}
extern "C" unsigned emu_enc_lds_need(unsigned level, unsigned srcSize) { return ze_lds_need(level, srcSize); }
// closed-form code / extra-bit functions against the format's tables, every input; returns the number of differences
extern "C" unsigned emu_check_code_tables() {
    unsigned bad = 0;
    for (u32 v = 0; v < (1u << 17) + 8u; v++) {
        if ((v > 63 ? zj_hibit(v) + 19 : ze_k_ll_code[v]) != ze_ll_code(v)) bad++;
        if ((v > 127 ? zj_hibit(v) + 36 : ze_k_ml_code[v]) != ze_ml_code(v)) bad++;
    }
    for (u32 c = 0; c < 36; c++) if (ze_k_ll_bits[c] != ze_ll_bits_of(c)) bad++;
    for (u32 c = 0; c < 53; c++) if (ze_k_ml_bits[c] != ze_ml_bits_of(c)) bad++;
    return bad;
}

// split pipeline: lane-per-frame match finding into HBM scratch, then the entropy stage; frames the classification
// kernel would put on list B (> 64 KiB, or fast-strategy tables beyond the common size) take the wide launch's layout
static int g_emu_force_gated = 0, g_emu_run = 0, g_emu_late = 0;
extern "C" unsigned long long emu_compress_split(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level) {
    EMU_IO(src, srcSize, dst, dstCap);
    if (srcSize > ZE_BLOCK_MAX) return ZJ_ERR64(201);
    Grp<1> g;
    u32 const flags = (level >> 8) & ZE_FLAG_MASK, hl = (level >> 16) & 0xFFu, cl = (level >> 24) & 0xFFu; level &= 0xFFu;   // test encoding: level | frame flags << 8 | hashLog << 16 | chainLog << 24
    u32 const lw = ZE_LW(level, hl, cl);
    u32 const ldsA = level == 1 ? (8192u * 2u) : (level == 2 ? (32768u * 2u) : (((1u << ZE_L3_HASHLOG) + (1u << ZE_L3_CHAINLOG)) * 2u));
    bool const wide = (hl | cl) ? srcSize > 65536u : ze_lds_need(level, srcSize) > (ldsA > (u32)sizeof(ZEEntropy) ? ldsA : (u32)sizeof(ZEEntropy));
    u32 const maxSrc = wide ? ZE_WIDE_MAX_SRC : 65536u;
    EmuWg& wg = emu_wg(); ZEncShared* sh = wg.sh; u8* lds = wg.lds; u8* ws = wg.ws;
    u8* table = (u8*)calloc(1, ze_lane_table_stride(lw, wide));
    u8* fs = (u8*)malloc(ZE_FRAME_STRIDE(maxSrc));
    u32 meta[3];
    // ZJNI_EMU_NEED=1: the need-gated double-fast machine behind zn_flags_frame (zj_need.h), as zj_enc_need_kernel + zj_enc_match_kernel run it
    u8* nflags = nullptr;
    {   ZEParams const p = ze_params_of(lw, srcSize);
        if (level == 3 && wide && getenv("ZJNI_EMU_NEED") && getenv("ZJNI_EMU_NEED")[0] == '7') g_emu_run = 1;      // the wide launch (frames of 64-128 KiB): the run machine without flags
        bool const wideFlags = level == 3 && wide && getenv("ZJNI_EMU_NEED") && strchr("568", getenv("ZJNI_EMU_NEED")[0]) && zn_takes_wide(p.hashLog, p.chainLog, srcSize);   // ... with flags (zn_flags_frame_wide), as the wide launch runs it
        if ((level == 3 && !wide && getenv("ZJNI_EMU_NEED") && zn_takes(p.hashLog, p.chainLog, srcSize)) || wideFlags) {
            struct One { u32 id() const { return 0; } u32 count() const { return 1; } void sync() const {} } one;
            ZNLds* L = (ZNLds*)malloc(sizeof(ZNLds)); memset(L, 0xA5, sizeof(ZNLds));
            nflags = (u8*)malloc(srcSize + ZN_FLAG_SLACK); memset(nflags, 0xFF, srcSize + ZN_FLAG_SLACK);
            char const nm = getenv("ZJNI_EMU_NEED")[0];         // ZLaneD: 1 flags for every frame, 2 for the picked frames (the others run the gated machine without flags)
            g_emu_run = (nm == '5' || nm == '6' || nm == '7' || nm == '8');  // the run machine (zj_match_run.h): 5 flags for every frame, 6 for the picked frames, 7 for none, 8: for every frame but LATE (taken over after a frame-dependent number of rounds, as the match kernel does when the flag kernel is still at work)
            g_emu_late = (nm == '8');
            bool const take = nm != '7' && ((nm != '2' && nm != '6') || zn_worth(one, (u32*)L, src, srcSize));
            if (take && wideFlags) zn_flags_frame_wide(one, *L, src, srcSize, p.hashLog, p.chainLog, p.minMatch, nflags);
            else if (take) zn_flags_frame(one, *L, src, srcSize, p.hashLog, p.chainLog, p.minMatch, nflags);
            else { free(nflags); nflags = nullptr; g_emu_force_gated = wideFlags ? 0 : 1; }
            if (nflags && getenv("ZJNI_EMU_NEED_STATS")) { unsigned c[4] = {0, 0, 0, 0}; for (u32 i = 0; i < srcSize; i++) for (int b = 0; b < 4; b++) c[b] += (nflags[i] >> b) & 1; fprintf(stderr, "need flags of %u positions: needL %u needS %u insL %u insS %u\n", srcSize, c[0], c[1], c[2], c[3]); }
            free(L);
        } }
    if (g_emu_run && g_emu_late && nflags && srcSize >= ZL_MIN_FRAME) {
        ZLaneR<ZEEntTag> m; m.init(src, srcSize, ze_params_of(lw, srcSize), table, fs, maxSrc, nullptr);
        u32 h = srcSize * 2654435761u; for (u32 i = 0; i < srcSize && i < 64u; i++) h = (h ^ src[i]) * 16777619u;
        u32 const lateRound = (h >> 8) % (srcSize / 2u + 1u);              // anywhere in the frame's first part (a frame takes between srcSize / 8 and srcSize rounds)
        for (u32 r = 0; m.st != ZL_DONE; r++) { m.round(m.phase_of(r)); if (r == lateRound) m.take_flags(nflags); }
        meta[0] = m.o.n; meta[1] = m.o.lit + m.lastLL; meta[2] = m.lastLL;
        g_emu_force_gated = 0; g_emu_late = 0;
    } else if (g_emu_run && srcSize >= ZL_MIN_FRAME) {
        if (getenv("ZJNI_EMU_JMAX")) { if (atoi(getenv("ZJNI_EMU_JMAX")) == 3) ze_match_lane_t<ZLaneR<ZEEntTag, 3u> >(src, srcSize, lw, table, fs, maxSrc, meta, nflags); else ze_match_lane_t<ZLaneR<ZEEntTag, 7u> >(src, srcSize, lw, table, fs, maxSrc, meta, nflags); }
        else ze_match_lane_t<ZLaneR<ZEEntTag> >(src, srcSize, lw, table, fs, maxSrc, meta, nflags);
        g_emu_force_gated = 0;
    } else if (g_emu_force_gated && !nflags && srcSize >= ZL_MIN_FRAME) {
        ze_match_lane_t<ZLaneD<ZEEntTag, true> >(src, srcSize, lw, table, fs, maxSrc, meta, nullptr);
        g_emu_force_gated = 0;
    } else ze_match_lane(src, srcSize, lw, table, fs, maxSrc, meta, wide, nflags);
    g_emu_run = 0;
    free(nflags);
    ZEPre pre; pre.seqs = (ZESeq*)fs; pre.litOff = (const u32*)(fs + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); pre.meta = meta;
    ZjProf pf; pf.start(nullptr);
    u64 r = ze_compress(g, *sh, lds, src, srcSize, dst, dstCap, lw, ws, pf, &pre, flags, nullptr, 160u * 1024u);
    free(fs); free(table);
    return r;
}

// ZSTD_splitBlock by chunks (zj_presplit.h): the one-lane walk and the whole-group version on the same 128 KiB
extern "C" unsigned emu_presplit_chunks(const unsigned char* p, int group) {
    Grp<1> g;
    u32* ev = (u32*)calloc(1024, 4);
    u32 const r = group ? zp_split_by_chunks_g(g, p, ev) : zp_split_by_chunks(p, ev);
    free(ev);
    return r;
}
// multi-block frames (128 KiB < srcSize <= 2 MiB): the frame loop of ze_compress_multi, lane-serial
extern "C" unsigned long long emu_compress_multi(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level) {
    EMU_IO(src, srcSize, dst, dstCap);
    Grp<1> g;
    u32 const flags = (level >> 8) & (ZE_FLAG_MASK | ZE_FLAG_MULTI_SERIAL | ZE_FLAG_MULTI_NOCARRY | ZE_FLAG_MULTI_FAST_SERIAL);
    bool const pipelined = (level & 0x8000u) != 0;                   // 0x8000: the pipelined pair of roles (zj_encode_pipe_kernel) instead of the one-wave loop
    level = ZE_LW(level & 0xFFu, (level >> 16) & 0xFFu, (level >> 24) & 0xFFu);                 // hashLog << 16 | chainLog << 24: the level word of a level-3 frame on the wave route (ZJNI_ROUTE_WAVE_HBM)     // 0x800: the one-lane parse of level-3 blocks instead of the wave matcher; 0x1000: the wave matcher without staged spans
    EmuWg& wg = emu_wg(); ZEncShared* sh = wg.sh; u8* lds = wg.lds; u8* ws = wg.ws;
    u32* tables = (u32*)malloc(ZE_MULTI_TABLE_BYTES);
    memset(tables, 0xA5, ZE_MULTI_TABLE_BYTES);                      // the encoder clears what it uses
    ZjProf pf; pf.start(nullptr);
    // zj_encode_multi_kernel's routing: a single block (level 4: match-finder tables in HBM) or a multi-block frame (levels 1-3)
    u64 r;
    if (srcSize > ZE_BLOCK_MAX && pipelined) {
        // zj_encode_pipe_kernel's two waves (round 6): a parse role and an entropy role with their own uniforms, LDS regions and scratch slots, run one after the other
        static ZEncShared* shP = nullptr; static u8* ldsP = nullptr; static u8* ws1 = nullptr;
        if (!shP) { shP = (ZEncShared*)malloc(sizeof(ZEncShared)); ldsP = (u8*)malloc(160 * 1024); ws1 = (u8*)malloc(ZE_SCRATCH_BYTES); }
        memset(shP, 0xA5, sizeof(ZEncShared)); memset(ldsP, 0x5A, 160 * 1024); memset(ws1, 0xC3, ZE_SCRATCH_BYTES);
        ZEPipe pipe; memset(&pipe, 0xA5, sizeof pipe);
        r = ze_compress_multi_pipe_serial(g, *shP, *sh, ldsP, lds, pipe, src, srcSize, dst, dstCap, level, ws, ws1, pf, flags, tables, 160u * 1024u);
    } else
    r = srcSize <= ZE_BLOCK_MAX ? ze_compress_t<Grp<1>, u32>(g, *sh, lds, src, srcSize, dst, dstCap, level, ws, pf, nullptr, flags, nullptr, 160u * 1024u, nullptr, tables)
                                : ze_compress_multi(g, *sh, lds, src, srcSize, dst, dstCap, level, ws, pf, flags, tables, 160u * 1024u);
    free(tables);
    return r;
}

// EXPERIMENT (not in the product): what ZSTD_compressStream2 without a pledged size makes of `src` fed with ZSTD_e_continue and closed
// with ZSTD_e_end — the stream natives' frames — rebuilt from the multi-block pieces: unknown-size parameters (equal to the one-shot ones above 256 KiB), header
// without content size and the unknown-size window byte, the input taken in chunks of 128 KiB (one pre-split at most per chunk, savings counted with the header's
// bytes), an empty raw last block when the total is a multiple of 128 KiB (an empty stream included).  srcSize <= the level's window.  Checked against oracle/ref.py compress_stream.
static unsigned long long emu_compress_stream_f(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level, const unsigned* flushAt, unsigned nFlush);
extern "C" unsigned long long emu_compress_stream(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level) {
    return emu_compress_stream_f(src, srcSize, dst, dstCap, level, nullptr, 0u);
}
// flushAt[0 .. nFlush): ascending byte counts after which the caller flushed (ZSTD_e_flush: the stream classes' flush()) — the bytes buffered at that moment become
// a block of their own and the 128 KiB chunking starts again behind them; a flush with nothing buffered writes nothing
extern "C" unsigned long long emu_compress_stream_flush(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level, const unsigned* flushAt, unsigned nFlush) {
    return emu_compress_stream_f(src, srcSize, dst, dstCap, level, flushAt, nFlush);
}
static unsigned long long emu_compress_stream_f(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level, const unsigned* flushAt, unsigned nFlush) {
    EMU_IO(src, srcSize, dst, dstCap);
    Grp<1> g;
    // test encoding of `level`: level | frame flags << 8 | 0x10000: not final (flushed, not closed) | 0x20000: closed before anything else was called (known-empty) | 0x40000: the one-lane block parses
    u32 const flags = ((level >> 8) & ZE_FLAG_MASK) | ((level & 0x40000u) ? (ZE_FLAG_MULTI_SERIAL | ZE_FLAG_MULTI_FAST_SERIAL) : 0u);
    u32 const final = (level & 0x10000u) ? 0u : 1u, knownEmpty = (level & 0x20000u) ? 1u : 0u; level &= 0xFFu;
    EmuWg& wg = emu_wg(); ZEncShared& sh = *wg.sh; u8* lds = wg.lds; u8* ws = wg.ws;
    u32* tables = (u32*)malloc(ZE_MULTI_TABLE_BYTES);
    ZjProf pf; pf.start(nullptr);
    u64 const out = ze_compress_stream(g, sh, lds, src, srcSize, dst, dstCap, level, ws, pf, flags, tables, 160u * 1024u, flushAt, nFlush, final, knownEmpty);
    free(tables);
    return out;
}

// levels 4-8, frames <= 16 KiB, as the large-batch route runs them: chain parser per frame (zj_enc_match_chain_kernel's body) into the
// record scratch, then the entropy stage on those records (zj_encode_kernel with `pre`)
extern "C" unsigned long long emu_compress_chain(const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned level) {
    EMU_IO(src, srcSize, dst, dstCap);
    if (srcSize > (16u << 10)) return ZJ_ERR64(201);
    Grp<1> g;
    u32 const flags = (level >> 8) & ZE_FLAG_MASK; level &= 0xFFu;
    EmuWg& wg = emu_wg(); ZEncShared* sh = wg.sh; u8* lds = wg.lds; u8* ws = wg.ws;
    u32 const maxSrc = 16u << 10;
    u32* table = (u32*)calloc(1, ((1u << 14) + (1u << 14)) * 4u);
    u8* fs = (u8*)malloc(ZE_FRAME_STRIDE(maxSrc));
    u32 meta[3];
    ZEOut o; o.seqs = (ZESeq*)fs; o.litOff = (u32*)(fs + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); o.n = 0; o.lit = 0;
    u32 lastLL = srcSize;
    if (srcSize >= 7u) { ZEParams const p = ze_params_of(level, srcSize); lastLL = ze_block_lazy(o, src, srcSize, p, table, table + (1u << p.hashLog)); }
    meta[0] = o.n; meta[1] = o.lit + lastLL; meta[2] = lastLL;
    ZEPre pre; pre.seqs = (ZESeq*)fs; pre.litOff = (const u32*)(fs + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); pre.meta = meta;
    ZjProf pf; pf.start(nullptr);
    u64 r = ze_compress(g, *sh, lds, src, srcSize, dst, dstCap, level, ws, pf, &pre, flags, nullptr, 160u * 1024u);
    free(fs); free(table);
    return r;
}

// dictionary compression: digest (ZSTD_createCDict) + ZSTD_CCtx_refCDict / ZSTD_compress2
#include "../../zstd-jni_amd/csrc/zj_cdict.h"
extern "C" void* emu_cdict_create(const unsigned char* dict, unsigned dictSize, unsigned level) {
    if (dictSize < 8 || level < 1 || level > 3) return nullptr;
    ZEParams const cp = ze_cdict_params(level, dictSize);
    size_t const tablesBytes = (size_t)ze_cdict_table_entries(cp) * 4u;
    size_t const head = (sizeof(ZECDictDev) + 15) & ~(size_t)15;
    u8* buf = (u8*)calloc(1, head + tablesBytes + dictSize + 16);
    ZECDictDev* cd = (ZECDictDev*)buf;
    cd->tablesOff = (u32)head; cd->rawOff = (u32)(head + tablesBytes);
    memcpy(buf + cd->rawOff, dict, dictSize);
    Grp<1> g;
    ZDecShared* sh = (ZDecShared*)calloc(1, sizeof(ZDecShared));
    ZEEntropy* e = (ZEEntropy*)calloc(1, sizeof(ZEEntropy));
    ze_cdict_digest(g, *sh, *e, dictSize, level, cd);
    free(e); free(sh);
    if (cd->status && getenv("EMU_DEBUG")) fprintf(stderr, "emu_cdict_create: digest status %u\n", cd->status);
    if (cd->status) { free(buf); return nullptr; }
    return buf;
}
extern "C" void emu_cdict_free(void* cd) { free(cd); }
extern "C" void emu_cdict_info(const void* p, unsigned* out) {
    const ZECDictDev* cd = (const ZECDictDev*)p;
    out[0] = cd->dictID; out[1] = cd->contentSize; out[2] = cd->windowLog; out[3] = cd->chainLog; out[4] = cd->hashLog; out[5] = cd->minMatch; out[6] = cd->strategy;
    out[7] = cd->hufRepeat; out[8] = cd->llRepeat; out[9] = cd->ofRepeat; out[10] = cd->mlRepeat; out[11] = cd->fillStart;
}
extern "C" unsigned long long emu_compress_cdict(const void* p, const unsigned char* src, unsigned srcSize, unsigned char* dst, unsigned dstCap, unsigned flags) {
    EMU_IO(src, srcSize, dst, dstCap);
    const ZECDictDev* cd = (const ZECDictDev*)p;
    Grp<1> g;
    ZEncShared* sh = (ZEncShared*)calloc(1, sizeof(ZEncShared));
    u8* lds = (u8*)calloc(1, 160 * 1024);
    u8* ws = (u8*)malloc(ZE_SCRATCH_BYTES);
    u8* table = (u8*)calloc(1, ZC_TABLE_STRIDE);
    u8* fs = (u8*)malloc(ZE_FRAME_STRIDE(ZC_MAX_SRC));
    u32 meta[3] = {0, srcSize, srcSize};
    ZEPre pre; pre.seqs = (ZESeq*)fs; pre.litOff = (const u32*)(fs + (size_t)ZE_FRAME_MAXSEQ(ZC_MAX_SRC) * 16u); pre.meta = meta;
    u32* big = nullptr;
    if (srcSize <= ze_attach_cutoff(cd->strategy)) ze_match_lane_dict(src, srcSize, cd, table, fs, ZC_MAX_SRC, meta);
    else if (ze_cdict_copy_mode(cd->strategy, srcSize, cd->contentSize)) {
        // zj_encode_cdict_copy_kernel's body: the dictionary's tables copied without their tags into the workgroup's slot, the
        // external-segment parse on lane 0 into the workgroup's record scratch, then the entropy stage on those records
        big = (u32*)malloc(ZE_MULTI_TABLE_BYTES); memset(big, 0xA5, ZE_MULTI_TABLE_BYTES);
        ze_cdict_copy_tables(g, cd, big);
        ze_cdict_copy_parse(cd, src, srcSize, big, ws, meta);
        pre.seqs = (ZESeq*)(ws + ZE_WS_SEQ); pre.litOff = (const u32*)(ws + ZE_WS_BODY); pre.copyMode = 1u;
    }
    ZjProf pf; pf.start(nullptr);
    u64 r = ze_compress(g, *sh, lds, src, srcSize, dst, dstCap, cd->level, ws, pf, &pre, flags & ZE_FLAG_MASK, cd, 160u * 1024u);
    free(big); free(fs); free(table); free(ws); free(lds); free(sh);
    return r;
}

// debugging aid: Huffman weights described at `hdr` (a literals section's tree description); returns the number of symbols
extern "C" unsigned emu_huf_weights(const unsigned char* hdr, unsigned size, unsigned char* weightsOut, unsigned* logOut) {
    ZDecShared* sh = (ZDecShared*)calloc(1, sizeof(ZDecShared));
    u32 nbSym = 0;
    u32 const h = zd_huf_read_weights(*sh, hdr, size, &nbSym);
    if (h) { memcpy(weightsOut, sh->weights, 256); *logOut = sh->hufLog; }
    free(sh);
    return h ? nbSym : 0;
}

// a tANS decode table both ways: zd_build_fse (one lane) and zd_fse_spread + zd_fse_finish_wave (the second pass by the wave, round 5); bit 0 / bit 1 of the result: each one's verdict
extern "C" int emu_fse_dtable(const short* norm, unsigned maxSV, unsigned tableLog, unsigned kind, unsigned* serialOut, unsigned* waveOut) {
    Grp<1> g;
    static u32 a[512], b[512], bm[ZD_FSE_BM_WORDS]; u16 sn1[64], sn2[64];
    memset(a, 0xA5, sizeof a); memset(b, 0x5A, sizeof b); memset(bm, 0x3C, sizeof bm);
    bool const ok1 = zd_build_fse(a, norm, sn1, maxSV, tableLog, kind);
    bool const ok2 = zd_fse_spread(b, norm, sn2, maxSV, tableLog);
    if (ok2) zd_fse_finish_wave(g, b, sn2, tableLog, kind, bm);
    memcpy(serialOut, a, sizeof(u32) << tableLog); memcpy(waveOut, b, sizeof(u32) << tableLog);
    return (ok1 ? 1 : 0) | (ok2 ? 2 : 0);
}
// the kernels' NCount reader on its own (zd_read_ncount): header bytes or 0; norm[0..*maxSV], *tableLog filled on success
extern "C" unsigned emu_read_ncount(const unsigned char* src, unsigned size, unsigned maxSV, short* normOut, unsigned* maxOut, unsigned* logOut) {
    short norm[256]; u32 mx = maxSV, tl = 0;
    u32 const h = zd_read_ncount(norm, &mx, &tl, src, size);
    if (h) { memcpy(normOut, norm, (mx + 1) * sizeof(short)); *maxOut = mx; *logOut = tl; }
    return h;
}

// rounds the double-fast lane machines take for one frame (ZLaneD: mode 0 plain, 1 gated; the run machine: 3 with flags, 4 without) — analysis aid
extern "C" unsigned emu_lane_rounds(const unsigned char* src, unsigned srcSize, unsigned mode) {
    u32 const lw = 3; ZEParams const p = ze_params_of(lw, srcSize);
    if (srcSize < ZL_MIN_FRAME || srcSize > 65536u) return 0;
    u8* table = (u8*)calloc(1, ze_lane_table_stride(lw, false)); u8* fs = (u8*)malloc(ZE_FRAME_STRIDE(65536u));
    u8* flags = nullptr;
    if (mode) {
        struct One { u32 id() const { return 0; } u32 count() const { return 1; } void sync() const {} } one;
        ZNLds* L = (ZNLds*)malloc(sizeof(ZNLds)); flags = (u8*)malloc(srcSize + ZN_FLAG_SLACK); memset(flags, 0xFF, srcSize + ZN_FLAG_SLACK);
        zn_flags_frame(one, *L, src, srcSize, p.hashLog, p.chainLog, p.minMatch, flags); free(L);
    }
    u32 rounds = 0;
    if (mode == 3 || mode == 4) { ZLaneR<ZEEntTag> m; m.init(src, srcSize, p, table, fs, 65536u, mode == 3 ? flags : nullptr); for (u32 r = 0; m.st != ZL_DONE; r++) { m.round(m.phase_of(r)); rounds++; } }
    else if (mode == 1) { ZLaneD<ZEEntTag, true> m; m.init(src, srcSize, p, table, fs, 65536u, flags); for (u32 r = 0; m.st != ZL_DONE; r++) { m.round(m.phase_of(r)); rounds++; } }
    else { ZLaneD<ZEEntTag> m; m.init(src, srcSize, p, table, fs, 65536u); for (u32 r = 0; m.st != ZL_DONE; r++) { m.round(m.phase_of(r)); rounds++; } }
    free(flags); free(fs); free(table);
    return rounds;
}

// the flag bytes zn_flags_frame leaves for one frame with the level-3 parameters of its size (tests: they must cover the exact answer)
extern "C" unsigned emu_need_flags(const unsigned char* src, unsigned srcSize, unsigned char* out, unsigned* params) {
    ZEParams const p = ze_params_of(3u, srcSize);
    if (zn_takes_wide(p.hashLog, p.chainLog, srcSize)) {
        struct One { u32 id() const { return 0; } u32 count() const { return 1; } void sync() const {} } one;
        ZNLds* L = (ZNLds*)malloc(sizeof(ZNLds)); memset(L, 0x5A, sizeof(ZNLds));
        zn_flags_frame_wide(one, *L, src, srcSize, p.hashLog, p.chainLog, p.minMatch, out);
        free(L);
        params[0] = p.hashLog; params[1] = p.chainLog; params[2] = p.minMatch;
        return 1;
    }
    if (!zn_takes(p.hashLog, p.chainLog, srcSize)) return 0;
    struct One { u32 id() const { return 0; } u32 count() const { return 1; } void sync() const {} } one;
    ZNLds* L = (ZNLds*)malloc(sizeof(ZNLds)); memset(L, 0x5A, sizeof(ZNLds));
    zn_flags_frame(one, *L, src, srcSize, p.hashLog, p.chainLog, p.minMatch, out);
    free(L);
    params[0] = p.hashLog; params[1] = p.chainLog; params[2] = p.minMatch;
    return 1;
}
// bucket of a position's long / short probe, as the lane machine computes it
extern "C" void emu_need_buckets(const unsigned char* src, unsigned srcSize, unsigned pos, unsigned* out) {
    ZEParams const p = ze_params_of(3u, srcSize);
    ZLHash const hL = zl_hash_of(8, p.hashLog), hS = zl_hash_of(p.minMatch, p.chainLog);
    u64 const w = ld64(src + pos);
    out[0] = zl_hash(hL, w); out[1] = zl_hash(hS, w);
}

// the wave's tANS table construction (ze_tans_shares / ze_tans_describe / ze_tans_table) on its own, for tests/test_emu_tans.py: the shares of a histogram, their
// description and the encoding table.  Returns 0 when the shares cannot be formed.
extern "C" int emu_tans(const unsigned* count, unsigned maxSV, unsigned total, unsigned tableLog, int low, short* shareOut, unsigned char* descOut, unsigned* descSize, unsigned* need,
                        unsigned short* stateOut, int* deltaFindOut, unsigned* deltaNbBitsOut) {
    Grp<1> g;
    static u32 cnt[64]; static short share[64]; static u32 scr[ZE_TANS_SCR]; static u32 desc32[40]; static u8 tableSymbol[4096]; static u32 bm[64 * 128]; static ZEFseCT ct;
    memset(scr, 0xA5, sizeof scr); memset(desc32, 0x5A, sizeof desc32); memset(tableSymbol, 0xEE, sizeof tableSymbol); memset(bm, 0x77, sizeof bm); memset(&ct, 0x11, sizeof ct); memset(share, 0x22, sizeof share);
    for (u32 s = 0; s < 64; s++) cnt[s] = s <= maxSV ? count[s] : 0;
    if (!ze_tans_shares(g, share, tableLog, cnt, total, maxSV, low != 0, scr)) return 0;
    memcpy(shareOut, share, 64 * sizeof(short));
    *need = 0;
    *descSize = ze_tans_describe(g, (u8*)desc32, share, maxSV, tableLog, scr, need);
    memcpy(descOut, desc32, 128);
    ze_tans_table(g, ct, share, maxSV, tableLog, tableSymbol, scr, bm);
    memcpy(stateOut, ct.state, sizeof ct.state); memcpy(deltaFindOut, ct.deltaFind, sizeof ct.deltaFind); memcpy(deltaNbBitsOut, ct.deltaNbBits, sizeof ct.deltaNbBits);
    return 1;
}
