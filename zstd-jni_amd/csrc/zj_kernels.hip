// zj_kernels.hip — gfx950 kernels + the C-ABI of include/zjni_amd.h (libzjni_amd.so).
//
// Launch shape: persistent workgroups of ONE wavefront (64 lanes); each pulls frame indices from a
// device-scope atomic counter (frames differ 4x in cost), so the grid is
// (#CUs x resident workgroups/CU), not the batch size.  Block b lands on XCD b % 8; frames are
// independent, so no cross-XCD traffic exists and the per-XCD L2s only see their own frames.
#include <hip/hip_runtime.h>
#include <mutex>
#include <algorithm>
#include <thread>
#include <sched.h>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <map>
#include <string>
#include <system_error>
#include <memory>
#include <vector>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/zjni_amd.h"
#include "zj_decode.h"
#include "zj_decode_split.h"
#include "zj_encode.h"
#include "zj_cdict.h"
#ifndef ZW_FRAME_IN_LDS
#define ZW_FRAME_IN_LDS 0
#endif
#include "zj_match_wave.h"
#include "zj_synth.h"

#define ZJNI_ERR(code) ((size_t)0 - (size_t)(code))

// A/B and debug switches (ZJNI_*): read from the environment ONCE per process and kept — getenv() races with setenv() from other JVM threads, and the
// per-buffer natives would otherwise pay ~30 lookups per call.  ZJNI_DEBUG_LIVE_SWITCHES=1 (set before the library loads; tests/conftest.py does) keeps
// them live so a test can flip a switch between two calls.
static const char* zj_env(const char* name) {
    static bool const live = []() { const char* v = getenv("ZJNI_DEBUG_LIVE_SWITCHES"); return v && atoi(v) == 1; }();
    if (live) return getenv(name);
    static std::mutex m; static std::map<std::string, std::pair<bool, std::string> > cache;     // (map nodes never move: the returned pointers stay valid)
    std::lock_guard<std::mutex> g(m);
    auto it = cache.find(name);
    if (it == cache.end()) { const char* v = getenv(name); it = cache.emplace(name, std::make_pair(v != nullptr, std::string(v ? v : ""))).first; }
    return it->second.first ? it->second.second.c_str() : nullptr;
}

// Experiment knobs (grid sizes, losing routes kept for A/B runs, measurement hooks) exist in a TUNING build only — tools/build_variant.sh <name> -DZJ_TUNING_KERNELS,
// its stamp ends in "+tuning" — and read as unset in the product, which keeps ten documented switches: ZJNI_SPLIT_MIN, ZJNI_DSPLIT_MIN,
// ZJNI_L3_WAVE_MAX, ZJNI_NEED, ZJNI_DEC_LIT, ZJNI_DEC_MB, ZJNI_WIDE_SLICE, ZJNI_HOST_THREADS, ZJNI_HOST_TRACE, ZJNI_DEBUG_SYNC (+ ZJNI_DEBUG_LIVE_SWITCHES above).
#ifdef ZJ_TUNING_KERNELS
static const char* zj_tune(const char* name) { return zj_env(name); }
#else
static constexpr const char* zj_tune(const char*) { return nullptr; }
#endif

// ============================================================================ kernels ==========
// Next work item of a persistent workgroup: one device-scope atomic by lane 0, broadcast through
// v_readfirstlane so the index (and everything derived from it) is wave-uniform.
__device__ __forceinline__ u32 zj_next_index(u32* counter) {
    u32 v = 0;
    if (threadIdx.x == 0) v = atomicAdd(counter, 1u);
    return (u32)__builtin_amdgcn_readfirstlane((int)v);
}

// Fused wave-per-frame decoder: frame i of the batch, or (list != nullptr) the frames a list names.
// DICT = true is the ZSTD_decompress_usingDDict variant (a second kernel, so that the common one keeps its registers).
template <bool DICT>
__global__ __launch_bounds__(64, 4) void zj_decode_kernel_t(const u8* __restrict__ src, const u64* __restrict__ srcOff,
                                                        u8* __restrict__ dst, const u64* __restrict__ dstOff,
                                                        u64* __restrict__ result, u32 n, u32* counter, u8* scratch, unsigned long long* prof,
                                                        const u32* __restrict__ list, const u32* listCount,
                                                        const ZDDictDev* dd, const u8* dictRaw) {
    __shared__ ZDecShared sh;
    ZjProf pf; pf.start(prof);
    Grp<64> g;
    u8* const lit = scratch + (size_t)blockIdx.x * ZD_LIT_SCRATCH;
    u32 const count = list ? ZJ_UNI(*listCount) : n;
    for (;;) {
        u32 const k = zj_next_index(counter);      // wave-uniform (SGPR)
        if (k >= count) break;
        u32 const i = list ? ZJ_UNI(list[k]) : k;
        u64 const s0 = srcOff[i], s1 = srcOff[i + 1], d0 = dstOff[i], d1 = dstOff[i + 1];
        u64 const cap = d1 - d0, len = s1 - s0;
        u64 const r = len > 0xFFFFFFFFull ? ZJ_ERR64(ZJ_E_SRCSIZE_WRONG)      // sizes are 32-bit inside the decoder: no silent truncation
                    : zd_decompress<DICT>(g, sh, src + s0, (u32)len, dst + d0, (u32)(cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : cap), lit, pf, dd, dictRaw);
        pf.mark(8);
        if (threadIdx.x == 0) result[i] = r;
        __syncthreads();
    }
}

extern __shared__ __attribute__((aligned(16))) u8 zj_dyn_lds[];

#define zj_decode_kernel zj_decode_kernel_t<false>
#define zj_dec_prep_kernel zj_dec_prep_kernel_t<false>
#define zj_dec_exec_kernel zj_dec_exec_kernel_t<false>
#define zj_decode_dict_kernel zj_decode_kernel_t<true>

// ZSTD_createDDict on the device: one workgroup digests the raw dictionary at dictRaw into *out
__global__ __launch_bounds__(64) void zj_ddict_digest_kernel(const u8* dictRaw, u32 dictSize, ZDDictDev* out) {
    __shared__ ZDecShared sh;
    Grp<64> g;
    zd_ddict_digest(g, sh, dictRaw, dictSize, out);
}

// ---- split decode pipeline (zj_decode_split.h): prep -> lane-per-frame sequence decode -> execute ----
// append k to a completion queue (the producer's records are visible before the entry): match kernel -> entropy kernel, sequence decode -> execution
__device__ __forceinline__ void zj_publish_done(u32* doneList, u32* doneCount, u32 k) {
    if (!doneList) return;
    zj_release();                                          // records + meta visible before the queue entry
    u32 const slot = atomicAdd(doneCount, 1u);
    __hip_atomic_store(&doneList[slot], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the queue entry alone, behind a release fence the caller has made for several lanes at once (zj_match_run)
__device__ __forceinline__ void zj_queue_done(u32* doneList, u32* doneCount, u32 k) {
    u32 const slot = atomicAdd(doneCount, 1u);
    __hip_atomic_store(&doneList[slot], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#ifndef ZJ_HAND_ROUNDS
#define ZJ_HAND_ROUNDS 256u      /* rounds between a match wave's hand-overs to the entropy kernel (a power of two; 32: -5 %, 256: -7 %, 2048: -2 % of the level-3 match kernel against a fence per frame, profiles/r05/f_) */
#endif
// Multi-block frames (zj_decode_split.h, "multi-block frames"): what stage 1 claims from — mb.ctr[0] blocks, [1] entries of seqList, [2] entries of listM, [4..5] records of the pool (64 bit)
struct ZDMbArgs { ZDFrameMB* frames; ZDBlk* blks; u16* tabs; u32* ctr; u32 blkCap; u32 minBlocks; unsigned long long seqCap; u32* seqList; u32* listM; unsigned long long litCap; u32* litList; };      // ctr: ... [8] lit list, [10..11] literal pool bytes (64 bit), [12] work of the literal pass
template <bool DICT>
__global__ __launch_bounds__(64, 4) void zj_dec_prep_kernel_t(const u8* __restrict__ src, const u64* __restrict__ srcOff, const u64* __restrict__ dstOff,
                                                          u32 n, u32* counter, u16* tabs, ZDMeta* metas, u32* listA, u32* listB, u32* listCounts,
                                                          const ZDDictDev* dd, u32* doneList, u32* procFlag, ZDMbArgs mb, u32 mbOnly, u8* __restrict__ dst, u64* __restrict__ result) {
    __shared__ ZDecShared sh;
    Grp<64> g;
    for (;;) {
        u32 const i = zj_next_index(counter);
        if (i >= n) break;
        u64 const s0 = srcOff[i], s1 = srcOff[i + 1], d0 = dstOff[i], d1 = dstOff[i + 1];
        u64 const cap = d1 - d0;
        bool const simple = !mbOnly && zd_prep_frame<DICT>(g, sh, src + s0, (u32)(s1 - s0), (u32)(cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : cap),
                                                           tabs + (size_t)i * ZD_SPLIT_CELLS, metas + i, dd);
        bool multi = false, stored = false;
        if (!DICT && !simple && dst && s1 - s0 <= 0xFFFFFFFFull) stored = zd_prep_frame_stored(g, sh, src + s0, (u32)(s1 - s0), dst + d0, cap, result + i);      // one raw / RLE block: copied here ([11] counts them)
        if (!DICT && !simple && !stored && mb.frames && s1 - s0 <= 0xFFFFFFFFull) {
            // (minBlocks: a small batch keeps its single-block frames on the fused kernel — one launch instead of three)
            multi = zd_prep_frame_multi(g, sh, src + s0, (u32)(s1 - s0), cap, i, mb.frames + i, mb.blks, mb.tabs, mb.ctr, mb.blkCap, (unsigned long long*)(mb.ctr + 4), mb.seqCap, mb.seqList, mb.ctr + 1, mb.minBlocks,
                                        (unsigned long long*)(mb.ctr + 10), mb.litCap, mb.litList, mb.ctr + 8);
        }
        if (threadIdx.x == 0) {
            if (doneList) { doneList[i] = 0xFFFFFFFFu; procFlag[i] = 0; }       // slot i of the completion queue / frame i's "executed by the side pass" flag
            // |A| and the batch's sequence total share one 64-bit counter ([8] = |A|, [9] = sequences, see zj_dec_heavy): one same-address atomic per frame
            if (simple) listA[(u32)atomicAdd((unsigned long long*)&listCounts[8], 1ull | ((unsigned long long)sh.nbSeq << 32))] = i;
            else if (stored) atomicAdd(&listCounts[11], 1u);
            else if (multi) mb.listM[atomicAdd(mb.ctr + 2, 1u)] = i;
            else listB[atomicAdd(&listCounts[1], 1u)] = i;
        }
        __syncthreads();
    }
}
// Multi-block frames, stage 2b: a WAVE per block regenerates its Huffman-coded literals into the block's slot of the literal pool (zd_lit_block)
__global__ __launch_bounds__(64) void zj_dec_lit_mb_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, const u32* __restrict__ litList, const u32* countPtr, u32* workCounter,
                                                            ZDBlk* blks, u8* litPool) {
    ZDecShared& sh = *(ZDecShared*)zj_dyn_lds;          // allocated without the tANS tables, the windows of zd_huf_streams_wave behind it (ZD_EXEC_LDS)
    ZjProf pf; pf.start(nullptr);
    Grp<64> g;
    u32 const count = ZJ_UNI(*countPtr);
    for (;;) {
        u32 const k = zj_next_index(workCounter);
        if (k >= count) break;
        u32 const b = ZJ_UNI(litList[k]);
        bool const ok = zd_lit_block(g, sh, src + zj_uni64(srcOff[ZJ_UNI(blks[b].frame)]), blks, b, litPool, pf, (u8*)zj_dyn_lds + ZD_HP_LDS_OFF);
        if (threadIdx.x == 0 && ok) blks[b].litReady = 1u;
        __syncthreads();
    }
}
// Multi-block frames, stage 2: a LANE per block (ZDSeqLaneT<true>: repcode history carried symbolically), records into the pool
__global__ __launch_bounds__(64) void zj_dec_seq_mb_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, const u32* __restrict__ seqList, const u32* countPtr, u32* workCounter,
                                                            const u16* tabs, u64* pool, ZDBlk* blks, u32 beside) {
    __shared__ u32 llBase[36], mlBase[53];
    zd_seq_symtabs(llBase, mlBase, threadIdx.x, 64u);
    __syncthreads();
    if (beside) __builtin_amdgcn_s_setprio(3);              // the chain of rounds is the critical path; stage 3's waves beside it (waiting for blocks, executing them) fill the gaps
    u32 const count = *countPtr;
    ZDSeqLaneT<true> m; m.st = 2; m.llBase = llBase; m.mlBase = mlBase;
    for (;;) {
        if (m.st == 2) {
            u32 const k = atomicAdd(workCounter, 1u);
            if (k >= count) break;
            u32 const b = seqList[k];
            ZDBlk* const bk = blks + b;
            m.init_block(src + srcOff[bk->frame], tabs + (size_t)b * ZD_SPLIT_CELLS, pool + (((u64)bk->seqHi << 32) | bk->seqLo), bk);
            continue;
        }
        m.round();
    }
}
// Multi-block frames, stage 3: a wave per frame walks its blocks in order (zd_exec_frame_multi); what it hands over goes to list B (the fused kernel)
__global__ __launch_bounds__(64) void zj_dec_exec_mb_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u8* __restrict__ dst, const u64* __restrict__ dstOff, u64* __restrict__ result,
                                                             const u32* __restrict__ listM, const u32* countPtr, u32* workCounter, const ZDFrameMB* frames, const ZDBlk* blks, const u64* pool,
                                                             u8* scratch, u32* listB, u32* listBCount, const u8* litPool, u32 mode, u32* procFlag) {
    // mode 0: list entry k.  mode 1: the same BESIDE stage 2 — a frame's blocks are executed in order, each as soon as stage 2 has set its seqReady (bounded wait; a
    // frame given up on is left to the mode-2 pass).  mode 2: the list entries mode 1 did not finish.
    __shared__ ZDecShared sh;
    ZjProf pf; pf.start(nullptr);
    Grp<64> g;
    u8* const lit = scratch + (size_t)blockIdx.x * ZD_LIT_SCRATCH;
    u32 const count = ZJ_UNI(*countPtr);
    for (;;) {
        u32 const k = zj_next_index(workCounter);
        if (k >= count) break;
        u32 const i = ZJ_UNI(listM[k]);
        if (mode == 2u && ZJ_UNI(procFlag[i])) continue;
        u64 const s0 = zj_uni64(srcOff[i]), d0 = zj_uni64(dstOff[i]), d1 = zj_uni64(dstOff[i + 1]);
        u64 const r = zd_exec_frame_multi(g, sh, src + s0, dst + d0, d1 - d0, frames + i, blks, pool, lit, (u8*)sh.ll, pf, litPool, mode == 1u);      // (sh.ll .. sh.ml: 4 KiB of LDS this kernel has no tANS tables in)
        if (threadIdx.x == 0 && r != ~(u64)0 - 1u) { if (r == ~(u64)0) listB[atomicAdd(listBCount, 1u)] = i; else result[i] = r; if (mode == 1u) procFlag[i] = 1u; }
        __syncthreads();
    }
}
// The sequence decode and the execution kernel run BESIDE each other when the frames carry enough sequences for the chain of
// decode rounds to dominate (>= 512 per frame on average: 64 KiB buffers have ~2 500, 4 KiB records ~100); for light frames the
// per-frame start-up dominates, every wave slot goes to the sequence decode and the execution kernel follows it.
__device__ __forceinline__ bool zj_dec_heavy(const u32* countA) { return countA[1] >= 512u * countA[0]; }      // countA = &listCounts[8]


__global__ __launch_bounds__(64) void zj_dec_seq_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, const u32* __restrict__ list,
                                                         const u32* countPtr, u32* workCounter, const u16* tabs, u64* seqs, ZDMeta* metas,
                                                         const ZDDictDev* dd, u32* doneList, u32* doneCount, u32 heavyWaves) {
    if (doneList && !zj_dec_heavy(countPtr)) doneList = nullptr;
    if (doneList && blockIdx.x >= heavyWaves) return;        // fewer resident waves = fewer decode cells alive at a time (2.5 KiB per lane); a lane takes more frames instead
    __shared__ u32 llBase[36], mlBase[53];
    zd_seq_symtabs(llBase, mlBase, threadIdx.x, 64u);
    __syncthreads();
#ifndef ZD_SEQ_PRIO
#define ZD_SEQ_PRIO 3
#endif
    if (doneList) __builtin_amdgcn_s_setprio(ZD_SEQ_PRIO);   // the chain of rounds is the critical path; the execution kernel's waves beside it fill the gaps
    u32 const count = *countPtr;
    ZDSeqLane m; m.st = 2; m.llBase = llBase; m.mlBase = mlBase;
    u32 cur = 0xFFFFFFFFu;                                   // the frame this lane is decoding
    for (;;) {
        if (m.st == 2) {
            if (cur != 0xFFFFFFFFu) { zj_publish_done(doneList, doneCount, cur); cur = 0xFFFFFFFFu; }   // the execution kernel runs beside this one and takes frames as they finish
            u32 const k = atomicAdd(workCounter, 1u);
            if (k >= count) break;
            u32 const i = list[k];
            m.init(src + srcOff[i], tabs + (size_t)i * ZD_SPLIT_CELLS, seqs + (size_t)i * ZD_SPLIT_MAXSEQ, metas + i, dd);
            cur = i;
            continue;
        }
        m.round();
    }
}

// Stage 2b: the Huffman-coded literals of list A's frames into their slots (zd_lit_frame), beside the sequence decode and ahead of the execution pass on the same side
// stream.  A kernel of its own since round 6 (it was the execution kernel's mode 3): the four streams are decoded by the whole wave (zd_huf_streams_wave), whose
// windows take ZD_HP_LDS bytes of LDS behind the workgroup state — the execution kernel keeps its registers and its workgroups per CU.
__global__ __launch_bounds__(64) void zj_dec_lit_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, const u32* __restrict__ list, const u32* countPtr,
                                                         u32* workCounter, ZDMeta* metas, u8* litSlots, u32 litSlot, unsigned long long* prof) {
    ZDecShared& sh = *(ZDecShared*)zj_dyn_lds;          // allocated without the tANS tables, the windows behind it (ZD_EXEC_LDS)
    ZjProf pf; pf.start(prof);
    Grp<64> g;
    u32 const count = ZJ_UNI(*countPtr);
    for (;;) {
        u32 const k = zj_next_index(workCounter);
        if (k >= count) break;
        u32 const i = ZJ_UNI(list[k]);
        bool const ok = zd_lit_frame(g, sh, src + srcOff[i], metas + i, litSlots + (size_t)i * litSlot, litSlot, pf, (u8*)zj_dyn_lds + ZD_HP_LDS_OFF);
        if (threadIdx.x == 0 && ok) metas[i].pad = 1u;      // the frame record says where the execution stage finds the literals
        __syncthreads();
    }
}

template <bool DICT>
__global__ __launch_bounds__(64) void zj_dec_exec_kernel_t(const u8* __restrict__ src, const u64* __restrict__ srcOff, u8* __restrict__ dst,
                                                          const u64* __restrict__ dstOff, u64* __restrict__ result, const u32* __restrict__ list,
                                                          const u32* countPtr, u32* workCounter, ZDMeta* metas, const u64* seqs, u8* scratch,
                                                          u32* listB, u32* listBCount, unsigned long long* prof, const ZDDictDev* dd, const u8* dictRaw,
                                                          u32 mode, const u32* doneList, u32* procFlag, u8* litSlots, u32 litSlot, u32* processed) {      // (metas: written by mode 3 only — the literal pass marks the frames it served; processed: frames mode 1 executed)
    // mode 0: list entry k.  mode 1: the k-th frame the sequence-decode kernel finishes while this kernel runs beside it (bounded
    // wait; a workgroup that gives up leaves the rest to the mode-2 pass).  mode 2: list entries mode 1 did not get to.
    // litSlots: slot i holds frame i's literals when its record says so (zj_dec_lit_kernel, beside the sequence decode).
    ZDecShared& sh = *(ZDecShared*)zj_dyn_lds;          // allocated without the tANS tables (ZD_SHARED_NO_FSE)
    ZjProf pf; pf.start(prof);
    Grp<64> g;
    u8* const lit = scratch + (size_t)blockIdx.x * ZD_LIT_SCRATCH;
    u32 const count = ZJ_UNI(*countPtr);
    if (mode == 1 && !zj_dec_heavy(countPtr)) return;        // light frames: everything is the mode-2 pass's
    if (mode == 2 && processed && ZJ_UNI(*processed) == count) return;      // the pass beside the sequence decode took every frame (round 6: walking the list to find that out was 0.7 ms)
    for (;;) {
        u32 const k = zj_next_index(workCounter);
        if (k >= count) break;
        u32 i;
        if (mode == 1) {
            u32 v = 0xFFFFFFFFu;
            if (threadIdx.x == 0) {
                u64 const t0 = wall_clock64();            // 100 MHz
                for (;;) {
                    v = __hip_atomic_load(&doneList[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // relaxed poll, fence below (see zj_encode_kernel)
                    if (v != 0xFFFFFFFFu || wall_clock64() - t0 > 200000000ull) break;     // 2 s
                    __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
                }
            }
            v = (u32)__builtin_amdgcn_readfirstlane((int)v);
            if (v == 0xFFFFFFFFu) break;
            zj_acquire();
            i = v;
        } else {
            i = ZJ_UNI(list[k]);
            if (mode == 2 && ZJ_UNI(procFlag[i])) continue;
        }
        u64 const r = zd_exec_frame<DICT>(g, sh, src + srcOff[i], dst + dstOff[i], metas + i, seqs + (size_t)i * ZD_SPLIT_MAXSEQ, lit, pf, dd, dictRaw,
                                          litSlots ? litSlots + (size_t)i * litSlot : (const u8*)nullptr, litSlot);
        pf.mark(8);
        if (threadIdx.x == 0) { if (r == ~(u64)0) listB[atomicAdd(listBCount, 1u)] = i; else result[i] = r; if (mode == 1) { procFlag[i] = 1u; if (processed) atomicAdd(processed, 1u); } }
        __syncthreads();
    }
}

// Encoder: dynamic LDS = the match-finder tables of the launch's (level, size class); the entropy stage
// overlays them.  A classification kernel splits the batch into two index lists: frames whose tables
// fit the LDS of the common case (list A) and the few that need more (list B; e.g. level-1 inputs of
// 8-16 KiB use hashLog 15, inputs > 64 KiB use 4-byte positions).  The same persistent kernel then runs
// once per list — with 128 KiB of LDS for list B, exiting at once when that list is empty.

__global__ __launch_bounds__(256) void zj_enc_classify_kernel(const u64* __restrict__ srcOff, u64* __restrict__ result, u32 n, u32 level,
                                                               u32 ldsA, u32* counters, u32* listA, u32* listB, u32* listC) {
    u32 const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 const size = srcOff[i + 1] - srcOff[i];
    if (ZE_LW_LEVEL(level) >= 4u) {                   // levels 4-8: frames <= 16 KiB -> list A here = the lane-per-frame chain parsers (zj_enc_match_chain_kernel);
        if (size <= (16u << 10)) listA[atomicAdd(&counters[0], 1u)] = i;                      // up to 128 KiB -> list C (level 4: double-fast, levels 5-8: the row-based finder;
        else if (listC && size <= ZE_BLOCK_MAX) listC[atomicAdd(&counters[4], 1u)] = i;       //   tables in HBM, one lane parses);
        else result[i] = ZJ_ERR64(201);                                                        // larger inputs: multi-block frames of these strategies are not served
        return;
    }
    if (size > ZE_BLOCK_MAX) {                        // multi-block frames (list C, zj_encode_multi_kernel) up to ZE_MULTI_MAX, without explicit table sizes
        if (listC && size <= ZE_MULTI_MAX && (!(ZE_LW_HL(level) | ZE_LW_CL(level)) || (level & ZE_LW_IMPLICIT))) { listC[atomicAdd(&counters[4], 1u)] = i; atomicAdd(&counters[6], 1u); }      // ([6]: the multi-block frames among list C — zj_pipe_route)
        else result[i] = ZJ_ERR64(201);
        return;
    }
    if (listC && (level & ZE_LW_WAVE_ROUTE)) { listC[atomicAdd(&counters[4], 1u)] = i; return; }      // a small level-3 batch: every frame its own wave (zj_match_wavex.h)
    bool const a = ZE_LW_TUNED(level) ? (size <= 65536u)                            // table sizes beyond the LDS: lane pipeline only, split by record width
                                                         : (ze_lds_need(ZE_LW_LEVEL(level), (u32)size) <= ldsA);
    if (a) listA[atomicAdd(&counters[0], 1u)] = i;
    else listB[atomicAdd(&counters[1], 1u)] = i;
}

// Lane-per-frame match finding (large batches): every lane owns one frame and runs the reference's
// sequential parse as a round-synchronous state machine (zj_match_lane.h); a lane whose frame is finished
// pulls the next list entry from a device counter inside the same loop, so the 64 lanes of a wave keep
// sharing memory round trips although frames differ 30x in cost.  Tables and sequence records of list
// entry k live at tables + k*tableStride and fscratch + k*ZE_FRAME_STRIDE(maxSrc) in HBM.
// Completion queue between the match kernel and the entropy kernel running beside it: a lane that has
// finished frame k (records and meta written) appends k; entropy workgroups consume the queue in order, so
// the cheap frames are entropy-coded while the expensive ones are still being parsed.
// Work queue over a list sorted by search density (zj_enc_score_kernel), cut at `split` (the first entry whose score reaches
// the threshold).  Entries [split, count) are the wave-per-frame kernel's alone (a lane would sit on such a frame for ~70 000
// rounds); it takes them from the back with its own counter.  Entries [0, split) are a two-ended queue: the lane-per-frame
// kernel claims from the front, the wave kernel — once its own part is done — from the back, and they meet wherever their
// speeds put them.  One 64-bit counter holds both cursors of that part (front low, back high), so every claim sees a
// consistent pair and exactly `split` claims succeed; a failed claim means the part is exhausted for good.
// Layout at work2: [0] u64 cursor pair, [2] u32 split, [3] u32 cursor of the wave kernel's own part.
__device__ __forceinline__ bool zj_claim_front(unsigned long long* work2, u32& k) {
    u32 const split = ((const u32*)work2)[2];
    unsigned long long const old = atomicAdd(work2, 1ull);
    u32 const head = (u32)old, tail = (u32)(old >> 32);
    k = head;
    return (u64)head + tail < split;
}
__device__ __forceinline__ bool zj_claim_back(unsigned long long* work2, u32 count, u32& k) {
    u32 const split = ((const u32*)work2)[2];
    u32 const t = atomicAdd(&((u32*)work2)[3], 1u);
    if (t < count - split) { k = count - 1u - t; return true; }
    unsigned long long const old = atomicAdd(work2, 1ull << 32);
    u32 const head = (u32)old, tail = (u32)(old >> 32);
    k = split - 1u - tail;
    return (u64)head + tail < split;
}

#ifdef ZL_PROFILE
__device__ unsigned long long zlWaveProf[4 * 2048];      // per workgroup of the last large match launch: cycles in the round loop, XCC_ID << 32 | HW_ID, rounds of lane 0, wall clock (100 MHz) at its start
extern "C" int zjni_debug_wave_profile(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(zlWaveProf), sizeof(zlWaveProf)); }
__device__ u32 zlFrameRounds[131072];                    // per frame of the last large match launch: the round in which its lane finished it (the launch's active lanes over time, profiles/r05/)
extern "C" int zjni_debug_frame_rounds(unsigned* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(zlFrameRounds), sizeof(zlFrameRounds)); }
#endif
template <class M>
__device__ __forceinline__ void zj_match_run(const u8* __restrict__ src, const u64* __restrict__ srcOff, u32 level,
                                             const u32* __restrict__ list, u32 count, u32* workCounter,
                                             u8* tables, u32 tableStride, u8* fscratch, u32 maxSrc, u32* meta, u32* doneList, u32* doneCount,
                                             unsigned long long* work2 = nullptr, const u8* flagsBase = nullptr, const u8* gate = nullptr,
                                             const u32* ready = nullptr, u32 flagStride = ZN_FLAG_STRIDE) {      // ready[k] != 0: the flags of list entry k are written (gate[k] says whether any will come); flagStride: flag bytes per list entry
    M m; m.st = ZL_DONE; m.lastLL = 0; m.o.n = 0; m.o.lit = 0;
#ifdef ZL_PROFILE
    u64 const zlWaveT0 = __builtin_readcyclecounter(); u64 zlRounds = 0; u64 const zlWall0 = wall_clock64();
#endif
    u32 have = 0, pend = 0, late = 0; u32 k = 0; u64 tPend = 0;          // (per-lane flags as 0 / 1 in vector registers: a loop-carried bool is a lane mask and every divergent assignment three scalar instructions)
    // Finished frames go to the entropy kernel's queue at the WAVE's hand-overs, every ZJ_HAND_ROUNDS rounds, behind ONE release fence for all lanes that finished since
    // (round 5).  On this part an agent-scope release writes the XCD's L2 back; a fence per finished frame — 65 536 per launch — was 8 % of the level-3 match kernel
    // (137.5 -> 126.0 ms with the fences taken out for a timing run, profiles/r05/f_).  A lane without a frame left stays (idle) until the whole wave is.
    u32 pubK = 0xFFFFFFFFu, idle = 0;
    u32 const period = ZE_LW_PERIOD(level) ? ZE_LW_PERIOD(level) : M::default_period(); u32 ph = 0;   // double-fast machines: rounds per rotation of the non-search states
    for (u32 r = 0;; r++) {
        if (pend) {                                       // (machines that cannot take their flags late) a frame that will get flags: start it when they are there — or without them when the wait runs out (50 ms)
            bool const rdy = __hip_atomic_load(&ready[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (rdy || wall_clock64() - tPend > 5000000ull) {
                if (rdy) zj_acquire();
                u32 const i = list[k];
                u64 const s0 = srcOff[i]; u32 const size = (u32)(srcOff[i + 1] - s0);
                m.init(src + s0, size, ze_params_of(level, size), tables + (size_t)k * tableStride, fscratch + (size_t)k * ZE_FRAME_STRIDE(maxSrc), maxSrc,
                       rdy ? flagsBase + (size_t)k * flagStride : nullptr);
                pend = 0u; have = 1u;
            }
        } else if (m.st == ZL_DONE && !idle) {
            if (have) {
                u32* const mt = meta + 3 * (size_t)k; mt[0] = m.o.n; mt[1] = m.o.lit + m.lastLL; mt[2] = m.lastLL; have = 0u;
#ifdef ZL_PROFILE
                if (count > 4096u && list[k] < 131072u) zlFrameRounds[list[k]] = r;          // (by frame index: the synthetic set's class is index & 3)
#endif
                if (doneList) {
                    if (pubK != 0xFFFFFFFFu) zj_publish_done(doneList, doneCount, pubK);      // (a second frame finished before the hand-over: with a fence of its own)
                    pubK = k;
                }
            }
            late = 0u;
            bool more;
            if (work2) more = zj_claim_front(work2, k);
            else { k = atomicAdd(workCounter, 1u); more = k < count; }
            if (!more) { if (!doneList) break; idle = 1u; }
            else {
            u32 const i = list[k];
            u64 const s0 = srcOff[i]; u32 const size = (u32)(srcOff[i + 1] - s0);
            u8* const tb = tables + (size_t)k * tableStride; u8* const fs = fscratch + (size_t)k * ZE_FRAME_STRIDE(maxSrc);
            if (size < ZL_MIN_FRAME) { ze_match_lane_serial(src + s0, size, level, tb, fs, maxSrc, meta + 3 * (size_t)k); zj_publish_done(doneList, doneCount, k); continue; }
            bool const flagged = flagsBase && gate[k];
            if (flagged && !M::takes_flags_late()) { pend = 1u; tPend = wall_clock64(); }
            else { m.init(src + s0, size, ze_params_of(level, size), tb, fs, maxSrc, nullptr); have = 1u; late = flagged ? 1u : 0u; }
            }
        }
        // A frame whose flags are still being computed starts WITHOUT them and takes them over when they arrive (flags only ever remove work, and what
        // they say about a position does not depend on when it is asked): once per rotation the lane asks; the request travels with the round's own loads.
        u32 rdyNow = 0;
        if (M::takes_flags_late() && late && ph == 0u) rdyNow = __hip_atomic_load(&ready[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef ZL_PROFILE
        zlRounds++;
#endif
        if (m.st != ZL_DONE) m.round(ZE_LW_LEVEL(level) == 3 ? ZJ_UNI(ph) : ZJ_UNI(r));
        if (M::takes_flags_late() && rdyNow != 0u) { zj_acquire(); m.take_flags(flagsBase + (size_t)k * flagStride); late = 0u; }
        ph = ph + 1u >= period ? 0u : ph + 1u;
        if (doneList) {                                   // the wave's hand-over (wave-uniform test)
            bool const allIdle = __ballot(!idle) == 0;
            if ((r & (ZJ_HAND_ROUNDS - 1u)) == ZJ_HAND_ROUNDS - 1u || allIdle) {
                if (__ballot(pubK != 0xFFFFFFFFu) != 0) {
                    zj_release();                         // records + meta of every frame the wave finished since the last one, before their queue entries
                    if (pubK != 0xFFFFFFFFu) { zj_queue_done(doneList, doneCount, pubK); pubK = 0xFFFFFFFFu; }
                }
                if (allIdle) break;
            }
        }
    }
#ifdef ZL_PROFILE
    if (threadIdx.x == 0 && blockIdx.x < 2048u && count > 4096u) { zlWaveProf[4 * blockIdx.x] = __builtin_readcyclecounter() - zlWaveT0; zlWaveProf[4 * blockIdx.x + 1] = ((u64)(u32)__builtin_amdgcn_s_getreg(63508) << 32) | (u32)__builtin_amdgcn_s_getreg(63492); zlWaveProf[4 * blockIdx.x + 2] = zlRounds; zlWaveProf[4 * blockIdx.x + 3] = zlWall0; }
    if (ZL_PROFILE > 1 && blockIdx.x == 0 && threadIdx.x < 4) printf("match lane profile: lane %u (frame class %u) done after %llu rounds, %llu Mcycles; cycles/round: phase1 %llu, loads %llu, phase3 %llu\n", threadIdx.x, threadIdx.x & 3, (unsigned long long)m.pR, (unsigned long long)((m.pA + m.pB + m.pC) / 1000000ull), (unsigned long long)(m.pA / m.pR), (unsigned long long)(m.pB / m.pR), (unsigned long long)(m.pC / m.pR));
#endif
}
// entries [listBase, listBase + sliceLen) of a list whose length sits in device memory (a slice past its end is empty)
__device__ __forceinline__ u32 zj_slice_count(const u32* countPtr, u32 listBase, u32 sliceLen) {
    u32 const count = *countPtr;
    return count > listBase ? zj_min(count - listBase, sliceLen) : 0u;
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void zj_enc_match_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u32 level,
                                                           const u32* __restrict__ list, const u32* countPtr, u32* workCounter,
                                                           u8* tables, u32 tableStride, u8* fscratch, u32 maxSrc, u32* meta, u32* doneList, u32* doneCount,
                                                           u32 listBase, u32 sliceLen, unsigned long long* work2) {
    u32 const count = zj_slice_count(countPtr, listBase, sliceLen);
    list += listBase;
    if (ZE_LW_LEVEL(level) == 3) zj_match_run<ZLaneD<ZEEntTag> >(src, srcOff, level, list, count, workCounter, tables, tableStride, fscratch, maxSrc, meta, doneList, doneCount, work2);
    else zj_match_run<ZLaneF<ZEEnt16> >(src, srcOff, level, list, count, workCounter, tables, tableStride, fscratch, maxSrc, meta, doneList, doneCount, work2);
}

// ---- need-gated level 3 (zj_need.h; ZJNI_NEED = 2 by default: flags for the frames zn_worth() picks) ----
// zj_enc_worth_kernel decides per frame whether it gets flags (gate[k]); zj_enc_need_kernel — one workgroup of 1 024 lanes per picked frame, Bloom
// filters in LDS — computes the flag bytes (which probes can match, which writes can be read) BESIDE the match kernel.  Every frame starts at once;
// a lane with a picked frame asks for ready[k] once per rotation and takes the flags over mid-frame when they are there (the run machine: what a flag
// says about a position does not depend on when it is asked, and until then every flag counts as set).  Nothing ever waits for the flag kernel: if it
// is late, or never scheduled beside the match kernel, frames simply run longer without flags (148 against 152 ms with the bounded wait this replaced,
// profiles/r03/l_late_flags_ab.txt).  The older gated machine (ZJNI_LANE_MACHINE=0) still waits, bounded by 50 ms, and then runs unflagged.
static u32 zj_need_threads() {        // lanes per frame of the flag kernel (one workgroup per CU: its filters take 108 KiB of LDS): ZJNI_NEED_THREADS, 64 .. 1 024
    u32 t = 1024; if (const char* ov = zj_tune("ZJNI_NEED_THREADS")) { int const v = atoi(ov); if (v >= 64 && v <= 1024) t = (u32)v & ~63u; }
    return t;
}
struct ZNThreads {
    __device__ __forceinline__ u32 id() const { return threadIdx.x; }
    __device__ __forceinline__ u32 count() const { return blockDim.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
};
// gate[k] = 1: list entry k will get flags (its slot number goes to pickList); selective = 0: every frame the lane machine takes
__global__ __launch_bounds__(256) void zj_enc_worth_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, const u32* __restrict__ list, const u32* countPtr,
                                                           u8* gate, u32* ready, u32 selective, u32* pickList, u32* pickCount) {
    __shared__ u32 bm[136];
    ZNThreads t;
    u32 const count = *countPtr;
    for (u32 k = blockIdx.x; k < count; k += gridDim.x) {
        u32 const i = list[k];
        u64 const s0 = srcOff[i]; u32 const size = (u32)(srcOff[i + 1] - s0);
        bool take = size >= ZL_MIN_FRAME;                 // (the plain loops take smaller frames, zj_match_run)
        if (take && selective) take = zn_worth(t, bm, src + s0, size);
        if (threadIdx.x == 0) { gate[k] = take ? 1 : 0; ready[k] = 0; if (take) pickList[atomicAdd(pickCount, 1u)] = k; }
    }
}
__global__ __launch_bounds__(1024) void zj_enc_need_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u32 level,
                                                          const u32* __restrict__ list, const u32* __restrict__ pickList, const u32* pickCount,
                                                          u8* flagsBase, u32* ready, u32* work, u32 flagStride) {
    ZNLds& L = *(ZNLds*)zj_dyn_lds;
    __shared__ u32 next;
    ZNThreads t;
    u32 const count = *pickCount;
    u32 prev = 0xFFFFFFFFu;                               // the slot whose flags this workgroup finished last: published by lane 0 in the SAME divergent region that claims
    for (;;) {                                            // the next one (a separate `if (lane 0)` at the loop's end makes the compiler route lane 0 around the barriers)
        if (threadIdx.x == 0) {                           // a work queue: workgroups differ in speed, and the match kernel's lanes are waiting
            if (prev != 0xFFFFFFFFu) { zj_release(); __hip_atomic_store(&ready[prev], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            next = atomicAdd(work, 1u);
        }
        __syncthreads();
        u32 const q = next;
        __syncthreads();
        if (q >= count) break;
        u32 const k = pickList[q], i = list[k];
        u64 const s0 = srcOff[i]; u32 const size = (u32)(srcOff[i + 1] - s0);
        ZEParams const p = ze_params_of(level, size);
        if (size > 65536u) zn_flags_frame_wide(t, L, src + s0, size, p.hashLog, p.chainLog, p.minMatch, flagsBase + (size_t)k * flagStride);      // (the wide launch's frames: a table at a time)
        else zn_flags_frame(t, L, src + s0, size, p.hashLog, p.chainLog, p.minMatch, flagsBase + (size_t)k * flagStride);      // (ends with a barrier: every lane's flag bytes are written)
        prev = k;
    }
}
// The level-3 match kernel of large batches: the run machine (zj_match_run.h) for every frame of the launch — register windows for the sequential
// streams, and for the frames with flags a run of quiet positions per round; frames without flags run it with every flag set.  Mixed waves on
// purpose: a wave of search-dense frames alone issues four times the requests per round (measured with a role split: 201 ms against 168).
template <u32 JMAX>
__device__ __forceinline__ void zj_enc_match_run_body(const u8* __restrict__ src, const u64* __restrict__ srcOff, u32 level,
                                                       const u32* __restrict__ list, const u32* countPtr, u32* workCounter,
                                                       u8* tables, u32 tableStride, u8* fscratch, u32 maxSrc, u32* meta, u32* doneList, u32* doneCount,
                                                       u32 listBase, u32 sliceLen, const u8* flagsBase, const u8* gate, const u32* ready) {
    u32 const count = zj_slice_count(countPtr, listBase, sliceLen);
    zj_match_run<ZLaneR<ZEEntTag, JMAX> >(src, srcOff, level, list + listBase, count, workCounter, tables, tableStride, fscratch, maxSrc, meta, doneList, doneCount, nullptr, flagsBase, gate, ready);
}
#define ZJ_RUN_KERNEL(NAME, JMAX) \
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void NAME(const u8* __restrict__ src, const u64* __restrict__ srcOff, u32 level, \
                                                           const u32* __restrict__ list, const u32* countPtr, u32* workCounter, \
                                                           u8* tables, u32 tableStride, u8* fscratch, u32 maxSrc, u32* meta, u32* doneList, u32* doneCount, \
                                                           u32 listBase, u32 sliceLen, const u8* flagsBase, const u8* gate, const u32* ready) { \
    zj_enc_match_run_body<JMAX>(src, srcOff, level, list, countPtr, workCounter, tables, tableStride, fscratch, maxSrc, meta, doneList, doneCount, listBase, sliceLen, flagsBase, gate, ready); }
ZJ_RUN_KERNEL(zj_enc_match_run_kernel, ZR_JMAX_DEFAULT)
#ifdef ZJ_TUNING_KERNELS                                      /* run-length sweep for A/B runs (tools/ab.sh, ZJNI_RUN_JMAX) */
ZJ_RUN_KERNEL(zj_enc_match_run3_kernel, 3u)
ZJ_RUN_KERNEL(zj_enc_match_run4_kernel, 4u)
ZJ_RUN_KERNEL(zj_enc_match_run6_kernel, 6u)
ZJ_RUN_KERNEL(zj_enc_match_run7_kernel, 7u)
#endif
#ifdef ZJ_TUNING_KERNELS
// ZJNI_LANE_MACHINE=0: the previous machine (ZLaneD with its table accesses predicated on the flags), kept selectable for A/B runs
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void zj_enc_match_gated_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u32 level,
                                                           const u32* __restrict__ list, const u32* countPtr, u32* workCounter,
                                                           u8* tables, u32 tableStride, u8* fscratch, u32 maxSrc, u32* meta, u32* doneList, u32* doneCount,
                                                           u32 listBase, u32 sliceLen, const u8* flagsBase, const u8* gate, const u32* ready) {
    u32 const count = zj_slice_count(countPtr, listBase, sliceLen);
    list += listBase;
    zj_match_run<ZLaneD<ZEEntTag, true> >(src, srcOff, level, list, count, workCounter, tables, tableStride, fscratch, maxSrc, meta, doneList, doneCount, nullptr, flagsBase, gate, ready);
}
#endif

// Levels 4-8, frames <= 16 KiB: the hash-chain parsers (ze_block_lazy: greedy / lazy / lazy2), one LANE per frame as plain loops —
// 64 frames per wave instead of one lane of 64 busy; the chain walk (up to 2^searchLog dependent candidate fetches per position)
// keeps the lanes apart, so the wave runs the union of their paths, but every memory round trip still serves up to 64 frames.
// Tables: hash then chain, u32, cleared by the host before the launch; records and meta where the entropy kernel expects them.
#define ZE_CHAIN_MAX_SRC (16u << 10)
#define ZE_CHAIN_TABLE_BYTES (((1u << 14) + (1u << 14)) * 4u)
__global__ __launch_bounds__(64) void zj_enc_match_chain_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u32 level,
                                                                 const u32* __restrict__ list, const u32* countPtr, u32* workCounter,
                                                                 u8* tables, u8* fscratch, u32* meta) {
    u32 const count = *countPtr;
    for (;;) {
        u32 const k = atomicAdd(workCounter, 1u);
        if (k >= count) break;
        u32 const i = list[k];
        u64 const s0 = srcOff[i]; u32 const size = (u32)(srcOff[i + 1] - s0);
        u8* const fs = fscratch + (size_t)k * ZE_FRAME_STRIDE(ZE_CHAIN_MAX_SRC); u32* const mt = meta + 3 * (size_t)k;
        ZEOut o; o.seqs = (ZESeq*)fs; o.litOff = (u32*)(fs + (size_t)ZE_FRAME_MAXSEQ(ZE_CHAIN_MAX_SRC) * 16u); o.n = 0; o.lit = 0;
        u32 lastLL = size;
        if (size >= 7u) {
            ZEParams const p = ze_params_of(level, size);
            u32* const t = (u32*)(tables + (size_t)k * ZE_CHAIN_TABLE_BYTES);
            lastLL = ze_block_lazy(o, src + s0, size, p, t, t + (1u << p.hashLog));
        }
        mt[0] = o.n; mt[1] = o.lit + lastLL; mt[2] = lastLL;
    }
}

// Levels 4-8, 16 KiB < frame <= 128 KiB, large batches: the same plain-loop parsers one lane per frame — level 4's double-fast with
// 2^17-entry tables, levels 5-8 on the row-based finder — with a table set (ZE_MULTI_TABLE_BYTES) per LANE SLOT of the launch instead
// of per frame: a wave's lanes each claim a frame, the wave clears the claimed slots' tables together (coalesced), every lane parses
// its frame, repeat.  Records and meta go to slice-relative slots; the entropy kernel follows on the same slice.
__global__ __launch_bounds__(64) void zj_enc_match_big_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u32 level,
                                                               const u32* __restrict__ list, const u32* countPtr, u32* workCounter,
                                                               u32* tables, u8* fscratch, u32* meta, u32 listBase, u32 sliceLen) {
    u32 const count = zj_slice_count(countPtr, listBase, sliceLen);
    list += listBase;
    u32* const t = tables + ((size_t)blockIdx.x * 64u + threadIdx.x) * (ZE_MULTI_TABLE_BYTES / 4u);
    for (;;) {
        u32 const k = atomicAdd(workCounter, 1u);
        bool const have = k < count;
        if (!__ballot(have)) break;
        u32 size = 0, entries = 0; u64 s0 = 0; ZEParams p; p.windowLog = p.chainLog = p.hashLog = p.minMatch = p.strategy = p.searchLog = 0;
        if (have) {
            u32 const i = list[k];
            s0 = srcOff[i]; size = (u32)(srcOff[i + 1] - s0);
            p = ze_params_of(level, size);
            entries = ze_params_uses_rows(p) ? (1u << p.hashLog) + (1u << p.hashLog) / 4u : (1u << p.hashLog) + (1u << p.chainLog);
        }
        for (u64 m = __ballot(have && size >= 7u); m; m &= m - 1) {          // clear the claimed slots' tables, one slot at a time, all lanes
            u32 const j = (u32)__builtin_ctzll(m);
            u32* const tj = tables + ((size_t)blockIdx.x * 64u + j) * (ZE_MULTI_TABLE_BYTES / 4u);
            u32 const ej = (u32)__shfl((int)entries, (int)j, 64);
            for (u32 x = threadIdx.x; x < ej; x += 64u) tj[x] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (have) {
            u8* const fs = fscratch + (size_t)k * ZE_FRAME_STRIDE(ZE_BLOCK_MAX); u32* const mt = meta + 3 * (size_t)k;
            ZEOut o; o.seqs = (ZESeq*)fs; o.litOff = (u32*)(fs + (size_t)ZE_FRAME_MAXSEQ(ZE_BLOCK_MAX) * 16u); o.n = 0; o.lit = 0;
            u32 lastLL = size;
            if (size >= 7u) lastLL = p.strategy >= 3u ? ze_block_lazy(o, src + s0, size, p, t, t + (1u << p.hashLog))
                                                      : ze_block_dfast<ZEEnt32>(o, src + s0, size, p.hashLog, p.chainLog, p.minMatch, t, t + (1u << p.hashLog));
            mt[0] = o.n; mt[1] = o.lit + lastLL; mt[2] = lastLL;
        }
    }
}

// Wave-per-frame match finding (zj_match_wave.h): level-3 frames <= 64 KiB with the tables in LDS, claimed from the back of the
// sorted list.  Records and meta of list entry k go where the lane kernel would put them, completion goes to the same queue.
__global__ __launch_bounds__(64) void zj_enc_match_wave_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u32 level,
                                                                const u32* __restrict__ list, const u32* countPtr, unsigned long long* work2,
                                                                u8* fscratch, u32 maxSrc, u32* meta, u32* doneList, u32* doneCount) {
    ZWLds& lds = *(ZWLds*)zj_dyn_lds;
    u32 const count = ZJ_UNI(*countPtr);
    for (;;) {
        u32 k = 0, ok = 0;
        if (threadIdx.x == 0) ok = zj_claim_back(work2, count, k) ? 1u : 0u;
        ok = ZJ_UNI(ok); k = ZJ_UNI(k);
        if (!ok) break;
        u32 const i = ZJ_UNI(list[k]);
        u64 const s0 = zj_uni64(srcOff[i]); u32 const size = (u32)(zj_uni64(srcOff[i + 1]) - s0);
        u8* const fs = fscratch + (size_t)k * ZE_FRAME_STRIDE(maxSrc); u32* const mt = meta + 3 * (size_t)k;
        if (size >= 64u) zw_match_frame(lds, src + s0, size, level, fs, maxSrc, mt);
        else {                                           // tiny frame: the plain loop on lane 0, tables in LDS
            ZEParams const p = ze_params_of(level, size);
            for (u32 j = threadIdx.x; j < (1u << p.hashLog); j += 64u) lds.HL[j] = 0;
            for (u32 j = threadIdx.x; j < (1u << p.chainLog); j += 64u) lds.HS[j] = 0;
            __syncthreads();
            if (threadIdx.x == 0) {
                ZEOut o; o.seqs = (ZESeq*)fs; o.litOff = (u32*)(fs + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); o.n = 0; o.lit = 0;
                u32 lastLL = size;
                if (size >= 7u) lastLL = ze_block_dfast<ZEEnt16>(o, src + s0, size, p.hashLog, p.chainLog, p.minMatch, lds.HL, lds.HS);
                mt[0] = o.n; mt[1] = o.lit + lastLL; mt[2] = lastLL;
            }
        }
        if (threadIdx.x == 0) zj_publish_done(doneList, doneCount, k);
        __syncthreads();
    }
}

#ifdef ZJ_TUNING_KERNELS      /* the ZJNI_HYBRID experiment (DESIGN history, round 2): it stayed off and lives in tuning builds only */
// Search-density score of each listed frame: distinct 4-byte values among 1 024 consecutive positions from the middle of the
// frame (hashed into a 4 096-bit set), 0..63.  Text-like frames (few distinct values: almost every position starts a match)
// score low, frames that are searched position by position score high.
__global__ __launch_bounds__(64) void zj_enc_score_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, const u32* __restrict__ list,
                                                           const u32* countPtr, u8* score) {
    __shared__ u32 bm[128];
    u32 const count = ZJ_UNI(*countPtr);
    for (u32 k = blockIdx.x; k < count; k += gridDim.x) {
        u32 const i = ZJ_UNI(list[k]);
        u64 const s0 = zj_uni64(srcOff[i]); u32 const size = (u32)(zj_uni64(srcOff[i + 1]) - s0);
        u32 sc = 0;
        if (size >= 64u) {
            u32 const avail = size - 4u, S = avail < 1024u ? avail : 1024u, off = (avail - S) >> 1;
            bm[threadIdx.x] = 0; bm[threadIdx.x + 64u] = 0;
            __syncthreads();
            for (u32 p = threadIdx.x; p < S; p += 64u) {
                u32 const h = (ld32(src + s0 + off + p) * 2654435761u) >> 20;
                atomicOr(&bm[h >> 5], 1u << (h & 31u));
            }
            __syncthreads();
            u32 c = (u32)__popc(bm[threadIdx.x]) + (u32)__popc(bm[threadIdx.x + 64u]);
            for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
            sc = (c * 64u) / S; if (sc > 63u) sc = 63u;
            __syncthreads();
        }
        if (threadIdx.x == 0) score[k] = (u8)sc;
    }
}
// Partition of the list for the two match kernels: frames scoring at least `threshold` go to the back (the wave kernel's own
// part, counted from the end) — at most `cap` of them, the wave kernel is the slower of the two per frame and what it takes
// off the lane kernel is HBM requests — everything else to the front IN LIST ORDER (as far as the atomics keep it): a lane
// wave should hold a mix of cheap and expensive frames, 64 expensive ones in one wave leave it running alone at the end.
// work2: [0] u64 cursor pair, [2] split, [3] cursor of the wave kernel's part, [4] search-dense frames seen, [5] front fill.
__global__ __launch_bounds__(256) void zj_enc_partition_kernel(const u32* __restrict__ list, const u32* countPtr, const u8* __restrict__ score, u32 threshold,
                                                               u32 sharePermille, unsigned long long* work2, u32* sorted) {
    u32 const k = blockIdx.x * blockDim.x + threadIdx.x, count = *countPtr;
    if (k >= count) return;
    u32* const w = (u32*)work2;
    u32 const cap = (u32)(((u64)count * sharePermille) / 1000u);
    if (score[k] >= threshold) {
        u32 const r = atomicAdd(&w[4], 1u);
        if (r < cap) { sorted[count - 1u - r] = list[k]; return; }
    }
    sorted[atomicAdd(&w[5], 1u)] = list[k];
}
__global__ void zj_enc_partition_done_kernel(const u32* countPtr, u32 sharePermille, unsigned long long* work2) {
    u32* const w = (u32*)work2;
    u32 const count = *countPtr, cap = (u32)(((u64)count * sharePermille) / 1000u);
    w[2] = count - (w[4] < cap ? w[4] : cap);
}
#endif

// The wide launch: frames > 64 KiB and the fast-strategy frames whose tables exceed the common size; 4-byte positions.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void zj_enc_match_wide_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u32 level,
                                                           const u32* __restrict__ list, const u32* countPtr, u32* workCounter,
                                                           u8* tables, u32 tableStride, u8* fscratch, u32 maxSrc, u32* meta, u32 listBase, u32 sliceLen, u32* doneList, u32* doneCount,
                                                           const u8* flagsBase, const u8* gate, const u32* ready) {
    u32 const count = zj_slice_count(countPtr, listBase, sliceLen);
    list += listBase;
    // With need flags (level 3, flagsBase != nullptr): the run machine for every frame of the slice, flags of ZN_FLAG_STRIDE_WIDE bytes per entry for the frames
    // zj_enc_worth_kernel picked, taken over when zj_enc_need_kernel has written them (zn_flags_frame_wide) — what zj_enc_match_run_kernel does on the common path.
    if (flagsBase) { zj_match_run<ZLaneR<ZEEntTag> >(src, srcOff, level, list, count, workCounter, tables, tableStride, fscratch, maxSrc, meta, doneList, doneCount, nullptr, flagsBase, gate, ready, ZN_FLAG_STRIDE_WIDE); return; }
    // without flags, level 3: ZLaneD.  The run machine WITHOUT flags was measured here in round 4 and is not faster on 128 KiB frames:
    // 65 536 x 128 KiB 382-441 ms with ZLaneD, 445-492 ms with ZLaneR (profiles/r04/d_, e_); tests/test_emu_encode.py keeps the machine exact at these sizes.
    if (ZE_LW_LEVEL(level) == 3) zj_match_run<ZLaneD<ZEEntTag> >(src, srcOff, level, list, count, workCounter, tables, tableStride, fscratch, maxSrc, meta, doneList, doneCount);
    else zj_match_run<ZLaneF<ZEEnt32> >(src, srcOff, level, list, count, workCounter, tables, tableStride, fscratch, maxSrc, meta, doneList, doneCount);
}
// zero the first `count` slots of `stride` bytes (the table slots of a slice; nothing to do for an empty slice)
__global__ __launch_bounds__(256) void zj_zero_slots_kernel(u8* base, u32 stride, const u32* countPtr, u32 listBase, u32 sliceLen) {
    u32 const count = zj_slice_count(countPtr, listBase, sliceLen);
    size_t const total16 = (size_t)count * (stride / 16u);
    uint4* const w = (uint4*)base;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < total16; j += (size_t)gridDim.x * 256) w[j] = make_uint4(0, 0, 0, 0);
}

// At most 168 VGPRs (three waves a SIMD): this kernel runs BESIDE the match kernels, and what it takes of the register file they
// cannot have — at 223 VGPRs the dictionary match kernel ran 16.5 -> 20.8 ms a slice (profiles/r03/i_entropy_vgpr_ab.txt).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) void zj_encode_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff,
                                                        u8* __restrict__ dst, const u64* __restrict__ dstOff,
                                                        u64* __restrict__ result, u32 level, const u32* __restrict__ list,
                                                        const u32* countPtr, u32* workCounter, u8* scratch, unsigned long long* prof,
                                                        u8* fscratch, u32 maxSrc, const u32* meta,
                                                        u32 mode, const u32* doneList, u32* procFlag, u32 flags, const ZECDictDev* cd, u32 ldsBytes, u32 listBase, u32 sliceLen) {
    // mode 0: list entry k.  mode 1: k-th entry of the completion queue the match kernel fills while this kernel
    // runs (bounded wait; a workgroup that gives up leaves its frame to the mode-2 pass).  mode 2: list entries
    // mode 1 did not finish.
    __shared__ ZEncShared sh;
    ZjProf pf; pf.start(prof);
    Grp<64> g;
    if (threadIdx.x == 0) { sh.dictLoaded = 0; sh.ctDict[0] = 0; sh.ctDict[1] = 0; sh.ctDict[2] = 0; }
    __syncthreads();
    u8* const ws = scratch + (size_t)blockIdx.x * ZE_SCRATCH_BYTES;
    u32 const count = ZJ_UNI(zj_slice_count(countPtr, listBase, sliceLen));
    list += listBase;
    for (;;) {
        u32 k = zj_next_index(workCounter);           // wave-uniform (SGPR)
        if (k >= count) break;
        if (mode == 1) {
            u32 v = 0xFFFFFFFFu;
            if (threadIdx.x == 0) {
                u64 const t0 = wall_clock64();            // 100 MHz
                for (;;) {
                    // relaxed poll: an acquire here would invalidate this CU's L1 every few microseconds, under the
                    // match waves that share it; the fence after the loop orders the reads of the frame's records
                    v = __hip_atomic_load(&doneList[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v != 0xFFFFFFFFu || wall_clock64() - t0 > 200000000ull) break;     // 2 s
                    __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
                }
            }
            v = (u32)__builtin_amdgcn_readfirstlane((int)v);
            if (v == 0xFFFFFFFFu) break;
            zj_acquire();
            k = v;
        } else if (mode == 2) {
            if (ZJ_UNI(procFlag[k])) continue;
        }
        u32 const i = ZJ_UNI(list[k]);
        u64 const s0 = zj_uni64(srcOff[i]), s1 = zj_uni64(srcOff[i + 1]), d0 = zj_uni64(dstOff[i]), d1 = zj_uni64(dstOff[i + 1]);
        u64 const cap = d1 - d0;
        ZEPre pre; const ZEPre* prePtr = nullptr;
        if (fscratch) {                                   // sequences were found by zj_enc_match_kernel
            u8* const fs = fscratch + (size_t)k * ZE_FRAME_STRIDE(maxSrc);
            pre.seqs = (ZESeq*)fs; pre.litOff = (const u32*)(fs + (size_t)ZE_FRAME_MAXSEQ(maxSrc) * 16u); pre.meta = meta + 3 * (size_t)k;
            prePtr = &pre;
        }
        u64 const r = ze_compress(g, sh, zj_dyn_lds, src + s0, (u32)(s1 - s0), dst + d0, (u32)(cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : cap), level, ws, pf, prePtr, flags, cd, ldsBytes);
        pf.mark(7);
        if (threadIdx.x == 0) { result[i] = r; if (mode == 1) procFlag[k] = 1u; }
        __syncthreads();
    }
}

// Which of the two kernels serves list C is decided on the device, by both, from the classify kernel's counters (countPtr = &counters[4]: |C|, work, multi-block frames):
// the pipelined one when every frame is resident at once (|C| <= pipeMax workgroups) and at least half of them are multi-block frames; the other one returns at once.
__device__ __forceinline__ bool zj_pipe_route(const u32* countPtr, u32 pipeMax) { u32 const c = countPtr[0]; return pipeMax != 0u && c <= pipeMax && 2u * countPtr[2] >= c && c != 0u; }
// Multi-block frames (128 KiB < input <= ZE_MULTI_MAX): one wavefront walks a frame block by block (ze_compress_multi); its
// frame-wide hash tables sit in HBM, one set per resident workgroup.
#ifndef ZJ_MULTI_WAVES
#define ZJ_MULTI_WAVES 2       /* resident waves per SIMD the register allocation aims at (the wave matcher's windows want ~220 VGPRs) */
#endif
__global__ __launch_bounds__(64, ZJ_MULTI_WAVES) void zj_encode_multi_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u8* __restrict__ dst, const u64* __restrict__ dstOff,
                                                              u64* __restrict__ result, u32 level, const u32* __restrict__ list, const u32* countPtr, u32* workCounter,
                                                              u8* scratch, u32* tables, u32 flags, u32 ldsBytes, u32 pipeMax) {
    if (zj_pipe_route(countPtr, pipeMax)) return;          // zj_encode_pipe_kernel's batch (queued ahead of this launch)
    __shared__ ZEncShared sh;
#ifdef ZX_PROFILE          /* analysis build: the entropy stage's phase marks of workgroup 0, printed when it is done (tools/ab_call*.sh) */
    __shared__ unsigned long long zxPhase[16];
    if (threadIdx.x < 16) zxPhase[threadIdx.x] = 0;
    __syncthreads();
    ZjProf pf; pf.start(blockIdx.x == 0 ? zxPhase : nullptr);
#else
    ZjProf pf; pf.start(nullptr);
#endif
    Grp<64> g;
    if (threadIdx.x == 0) { sh.dictLoaded = 0; sh.ctDict[0] = 0; sh.ctDict[1] = 0; sh.ctDict[2] = 0; }
    __syncthreads();
    u8* const ws = scratch + (size_t)blockIdx.x * ZE_SCRATCH_BYTES;
    u32* const tb = tables + (size_t)blockIdx.x * (ZE_MULTI_TABLE_BYTES / 4u);
    u32 const count = ZJ_UNI(*countPtr);
    for (;;) {
        u32 const k = zj_next_index(workCounter);
        if (k >= count) break;
        u32 const i = ZJ_UNI(list[k]);
        u64 const s0 = zj_uni64(srcOff[i]), s1 = zj_uni64(srcOff[i + 1]), d0 = zj_uni64(dstOff[i]), d1 = zj_uni64(dstOff[i + 1]);
        u64 const cap = d1 - d0;
        u32 const size = (u32)(s1 - s0), capU = (u32)(cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : cap);
        // level 4 (one block, tables too large for LDS) or a multi-block frame of levels 1-3
        u64 const r = size <= ZE_BLOCK_MAX ? ze_compress_t<Grp<64>, u32>(g, sh, zj_dyn_lds, src + s0, size, dst + d0, capU, level, ws, pf, nullptr, flags, nullptr, ldsBytes, nullptr, tb)
                                           : ze_compress_multi(g, sh, zj_dyn_lds, src + s0, size, dst + d0, capU, level, ws, pf, flags, tb, ldsBytes);
        if (threadIdx.x == 0) result[i] = r;
        __syncthreads();
    }
#ifdef ZX_PROFILE
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("zx phases wg 0 (kcycles): zero+params %llu  match %llu  literals gather+codes %llu  hist+huffman table %llu  huffman encode %llu  sequence tables %llu  sequence encode %llu  block place %llu\n",
               zxPhase[0] / 1000ull, zxPhase[1] / 1000ull, zxPhase[2] / 1000ull, zxPhase[3] / 1000ull, zxPhase[4] / 1000ull, zxPhase[5] / 1000ull, zxPhase[6] / 1000ull, zxPhase[7] / 1000ull);
#endif
}

// Multi-block frames, PIPELINED (round 6; zj_encode.h "multi-block frames, PIPELINED"): a workgroup of TWO waves per frame — wave 1 parses block b + 1 while wave 0
// entropy-codes block b.  For batches that cannot fill the device with one-wave chains (a thousand 1 MiB frames: BASELINE config 1); with every wave slot taken the
// one-wave kernel above does the same work in fewer wave-milliseconds, so large batches stay there (compress_batch_device_impl).  Scratch: two slots of encScratch
// per workgroup (block b in slot b & 1), one table set; dynamic LDS: [ZEEntropy of the entropy wave][ZJ_PIPE_LDS_P bytes of the parse wave].
#define ZJ_PIPE_LDS_P 10240u
static_assert(sizeof(ZXLds) <= ZJ_PIPE_LDS_P && 2064u <= ZJ_PIPE_LDS_P, "the parse wave's LDS holds the wave matcher's scoreboards, or the pre-splitter's histograms");
__global__ __launch_bounds__(128, 2) void zj_encode_pipe_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u8* __restrict__ dst, const u64* __restrict__ dstOff,
                                                             u64* __restrict__ result, u32 level, const u32* __restrict__ list, const u32* countPtr, u32* workCounter,
                                                             u8* scratch, u32* tables, u32 flags, u32 ldsBytesE, u32 pipeMax) {
    if (!zj_pipe_route(countPtr, pipeMax)) return;
    // (ONE LDS object: declared as separate variables the two waves' uniforms came out OVERLAID — group_segment_fixed_size 1 440 instead of 2 700: each is used by one
    //  role's code only, and the LDS lowering does not know that the roles run at the same time on different waves)
    struct PipeShared { ZEncShared e, p; ZEPipe pipe; u32 nextK; };
    __shared__ PipeShared S;
    ZEncShared& shE = S.e; ZEncShared& shP = S.p; ZEPipe& pipe = S.pipe; u32& nextK = S.nextK;
    u32 const role = threadIdx.x >> 6;                     // 0: the entropy wave (and single-block frames of the list), 1: the parse wave
    GrpWave g;
    ZjProf pf; pf.start(nullptr);
    if (threadIdx.x == 0) { shE.dictLoaded = 0; shE.ctDict[0] = 0; shE.ctDict[1] = 0; shE.ctDict[2] = 0; shP.dictLoaded = 0; shP.ctDict[0] = 0; shP.ctDict[1] = 0; shP.ctDict[2] = 0; }
    __syncthreads();
    u8* const ws0 = scratch + (size_t)(2u * blockIdx.x) * ZE_SCRATCH_BYTES; u8* const ws1 = ws0 + ZE_SCRATCH_BYTES;
    u32* const tb = tables + (size_t)blockIdx.x * (ZE_MULTI_TABLE_BYTES / 4u);
    u8* const ldsE = zj_dyn_lds; u8* const ldsP = zj_dyn_lds + ((ldsBytesE + 15u) & ~15u);      // (sizeof(ZEEntropy) is 4 mod 8 and the wave matcher's scoreboards are 64-bit LDS atomics)
    u32 const count = ZJ_UNI(*countPtr);
    for (;;) {
        if (threadIdx.x == 0) nextK = atomicAdd(workCounter, 1u);
        __syncthreads();
        u32 const k = ZJ_UNI(nextK);
        __syncthreads();                                    // (both waves have it before the next round writes it)
        if (k >= count) break;
        u32 const i = ZJ_UNI(list[k]);
        u64 const s0 = zj_uni64(srcOff[i]), s1 = zj_uni64(srcOff[i + 1]), d0 = zj_uni64(dstOff[i]), d1 = zj_uni64(dstOff[i + 1]);
        u64 const cap = d1 - d0;
        u32 const size = (u32)(s1 - s0), capU = (u32)(cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : cap);
        ZEParams const p = ze_params_of(ZE_LW_LEVEL(level), size);
        if (size <= ZE_BLOCK_MAX) {                         // level 4 (one block, tables too large for LDS): the entropy wave alone, as zj_encode_multi_kernel does it
            if (role == 0) { u64 const r = ze_compress_t<GrpWave, u32>(g, shE, ldsE, src + s0, size, dst + d0, capU, level, ws0, pf, nullptr, flags, nullptr, ldsBytesE, nullptr, tb); if ((threadIdx.x & 63u) == 0) result[i] = r; }
        } else if (size > ZE_MULTI_MAX || size > (1u << p.windowLog)) { if (threadIdx.x == 0) result[i] = ZJ_ERR64(201); }
        else {
            if (threadIdx.x == 0) { pipe.ready = 0; pipe.done = 0; pipe.err = 0; }
            __syncthreads();
            if (role == 1) {
                ze_pipe_parse_role(g, shP, ldsP, pipe, src + s0, size, capU, level, flags, tb, ws0, ws1, [&](u32 want) {
                    for (;;) {
                        if (ZJ_UNI(ze_pipe_load(&pipe.done)) >= want || ZJ_UNI(ze_pipe_load(&pipe.err))) break;
                        __builtin_amdgcn_s_sleep(32);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                });
            } else {
                ZEPipeE st;
                ze_pipe_entropy_init(g, shE, st, dst + d0, capU, size, level, flags);
                if (st.finished) { if ((threadIdx.x & 63u) == 0) pipe.err = 1; }
#ifdef ZE_PIPE_DEBUG
                u64 dbgWaitE = 0, dbgWorkE = 0;
#endif
                while (!st.finished) {
#ifdef ZE_PIPE_DEBUG
                    u64 const t0 = wall_clock64();
#endif
                    while (ZJ_UNI(ze_pipe_load(&pipe.ready)) <= st.b) __builtin_amdgcn_s_sleep(32);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef ZE_PIPE_DEBUG
                    u64 const t1 = wall_clock64(); dbgWaitE += t1 - t0;
#endif
                    ze_pipe_entropy_step(g, shE, ldsE, pipe, st, src + s0, size, dst + d0, capU, level, flags, ws0, ws1, pf, ldsBytesE);
#ifdef ZE_PIPE_DEBUG
                    dbgWorkE += wall_clock64() - t1;
#endif
                }
#ifdef ZE_PIPE_DEBUG
                if (blockIdx.x < 4u && (threadIdx.x & 63u) == 0) printf("pipe E wg %u: %u blocks, waiting %llu us, working %llu us\n", blockIdx.x, st.b, (unsigned long long)(dbgWaitE / 100ull), (unsigned long long)(dbgWorkE / 100ull));
#endif
                if ((threadIdx.x & 63u) == 0) result[i] = st.result;
            }
        }
        __syncthreads();
    }
}

// Stream frames (ze_compress_stream, zj_encode.h): what the reference's stream classes produce without a pledged size, one wavefront per stream over the
// multi-block kernel's table slots.  mode[i]: bit 0 final (close()), bit 1 the stream was closed before anything else was called on it; flushAt / flushOff:
// the flush positions of stream i = flushAt[flushOff[i] .. flushOff[i + 1]) (flushOff == nullptr: none).
__global__ __launch_bounds__(64, ZJ_MULTI_WAVES) void zj_encode_stream_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u8* __restrict__ dst, const u64* __restrict__ dstOff,
                                                               u64* __restrict__ result, u32 level, u32 count, u32* workCounter, u8* scratch, u32* tables, u32 flags, u32 ldsBytes,
                                                               const u32* __restrict__ flushAt, const u64* __restrict__ flushOff, const u32* __restrict__ mode) {
    __shared__ ZEncShared sh;
    ZjProf pf; pf.start(nullptr);
    Grp<64> g;
    if (threadIdx.x == 0) { sh.dictLoaded = 0; sh.ctDict[0] = 0; sh.ctDict[1] = 0; sh.ctDict[2] = 0; }
    __syncthreads();
    u8* const ws = scratch + (size_t)blockIdx.x * ZE_SCRATCH_BYTES;
    u32* const tb = tables + (size_t)blockIdx.x * (ZE_MULTI_TABLE_BYTES / 4u);
    for (;;) {
        u32 const i = zj_next_index(workCounter);
        if (i >= count) break;
        u64 const s0 = zj_uni64(srcOff[i]), s1 = zj_uni64(srcOff[i + 1]), d0 = zj_uni64(dstOff[i]), d1 = zj_uni64(dstOff[i + 1]);
        u64 const cap = d1 - d0, size64 = s1 - s0;
        u32 const capU = (u32)(cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : cap);
        u64 const f0 = flushOff ? zj_uni64(flushOff[i]) : 0, f1 = flushOff ? zj_uni64(flushOff[i + 1]) : 0;
        u32 const md = mode ? ZJ_UNI(mode[i]) : 1u;
        u64 const r = size64 > ZE_MULTI_MAX ? ZJ_ERR64(201)
                                            : ze_compress_stream(g, sh, zj_dyn_lds, src + s0, (u32)size64, dst + d0, capU, level, ws, pf, flags, tb, ldsBytes, flushAt + f0, (u32)(f1 - f0), md & 1u, (md >> 1) & 1u);
        if (threadIdx.x == 0) result[i] = r;
        __syncthreads();
    }
}

// ZSTD_createCDict on the device: one workgroup digests the dictionary held in `out` (header, zeroed tables, raw bytes)
__global__ __launch_bounds__(64) void zj_cdict_digest_kernel(u32 dictSize, u32 level, ZECDictDev* out) {
    __shared__ ZDecShared sh;
    __shared__ ZEEntropy e;
    Grp<64> g;
    ze_cdict_digest(g, sh, e, dictSize, level, out);
}

// Attach-mode match finding against a dictionary, lane per frame (zj_cdict.h): the dictionary's tables and content
// are shared by every lane (L2-resident), each frame's own tables sit in its HBM slot.  Frames outside the attach
// range are left to the entropy kernel, which reports them.
__global__ __launch_bounds__(64) void zj_enc_match_dict_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, const ZECDictDev* __restrict__ cd,
                                                                const u32* __restrict__ list, const u32* countPtr, u32* workCounter,
                                                                u8* tables, u8* fscratch, u32* meta) {
    u32 const count = *countPtr;
    u32 const cutoff = ze_attach_cutoff(cd->strategy);
    for (;;) {
        u32 const k = atomicAdd(workCounter, 1u);
        if (k >= count) break;
        u32 const i = list[k];
        u64 const s0 = srcOff[i]; u64 const size = srcOff[i + 1] - s0;
        if (size > cutoff) continue;
        ze_match_lane_dict(src + s0, (u32)size, cd, tables + (size_t)k * ZC_TABLE_STRIDE, fscratch + (size_t)k * ZE_FRAME_STRIDE(ZC_MAX_SRC), ZC_MAX_SRC, meta + 3 * (size_t)k);
    }
}

// Dictionary compression beyond the attach range (copy mode, zj_cdict.h): counted per call, then one wave per frame — the dictionary's
// tables copied (tags stripped) into the workgroup's HBM slot, the external-segment parse on lane 0 into the workgroup's record
// scratch, the entropy stage with the dictionary's tables as "previous block".  Such sources are rare next to the 4-16 KiB records
// dictionaries are made for; the kernel returns at once when the call has none.
__global__ __launch_bounds__(256) void zj_cdict_count_copy_kernel(const u64* __restrict__ srcOff, u32 n, const ZECDictDev* __restrict__ cd, u32* counter) {
    u32 const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 const size = srcOff[i + 1] - srcOff[i];
    if (size <= ZE_BLOCK_MAX && ze_cdict_copy_mode(cd->strategy, (u32)size, cd->contentSize)) atomicAdd(counter, 1u);
}
__global__ __launch_bounds__(64) void zj_encode_cdict_copy_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, u8* __restrict__ dst, const u64* __restrict__ dstOff,
                                                                   u64* __restrict__ result, u32 n, const ZECDictDev* __restrict__ cd, const u32* copyCount, u32* workCounter,
                                                                   u8* scratch, u32* tables, u32 flags, u32 ldsBytes) {
    if (*copyCount == 0) return;
    __shared__ ZEncShared sh;
    __shared__ u32 metaL[4];
    ZjProf pf; pf.start(nullptr);
    Grp<64> g;
    if (threadIdx.x == 0) { sh.dictLoaded = 0; sh.ctDict[0] = 0; sh.ctDict[1] = 0; sh.ctDict[2] = 0; }
    __syncthreads();
    u8* const ws = scratch + (size_t)blockIdx.x * ZE_SCRATCH_BYTES;
    u32* const tb = tables + (size_t)blockIdx.x * (ZE_MULTI_TABLE_BYTES / 4u);
    u32 const strategy = ZJ_UNI(cd->strategy), content = ZJ_UNI(cd->contentSize);
    for (;;) {
        u32 const i = zj_next_index(workCounter);
        if (i >= n) break;
        u64 const s0 = zj_uni64(srcOff[i]), s1 = zj_uni64(srcOff[i + 1]), d0 = zj_uni64(dstOff[i]), d1 = zj_uni64(dstOff[i + 1]);
        if (s1 - s0 > ZE_BLOCK_MAX || !ze_cdict_copy_mode(strategy, (u32)(s1 - s0), content)) continue;
        u32 const size = (u32)(s1 - s0); u64 const cap = d1 - d0;
        ze_cdict_copy_tables(g, cd, tb);
        zj_mem_order();
        __syncthreads();
        if (threadIdx.x == 0) ze_cdict_copy_parse(cd, src + s0, size, tb, ws, metaL);
        zj_mem_order();
        __syncthreads();
        ZEPre pre; pre.seqs = (ZESeq*)(ws + ZE_WS_SEQ); pre.litOff = (const u32*)(ws + ZE_WS_BODY); pre.meta = metaL; pre.copyMode = 1u;
        u64 const r = ze_compress(g, sh, zj_dyn_lds, src + s0, size, dst + d0, (u32)(cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : cap), ZJ_UNI(cd->level), ws, pf, &pre, flags, cd, ldsBytes);
        if (threadIdx.x == 0) result[i] = r;
        __syncthreads();
    }
}

// Clears exactly the part of each frame's table slot its attach-mode parameters use (24 KiB for a 4 KiB record, not the
// 96 KiB slot): one workgroup per list entry, 16 bytes per lane per step.
__global__ __launch_bounds__(256) void zj_cdict_zero_tables_kernel(const u64* __restrict__ srcOff, const ZECDictDev* __restrict__ cd,
                                                                   const u32* __restrict__ list, const u32* countPtr, u8* tables) {
    u32 const count = *countPtr;
    ZEParams cdp; cdp.windowLog = cd->windowLog; cdp.chainLog = cd->chainLog; cdp.hashLog = cd->hashLog; cdp.minMatch = cd->minMatch; cdp.strategy = cd->strategy;
    u32 const cutoff = ze_attach_cutoff(cdp.strategy);
    for (u32 k = blockIdx.x; k < count; k += gridDim.x) {
        u32 const i = list[k];
        u64 const size = srcOff[i + 1] - srcOff[i];
        if (size < 7 || size > cutoff) continue;
        ZEParams const p = ze_attach_params(cdp, (u32)size);
        u32 const bytes = ((1u << p.hashLog) + (p.strategy == 2 ? (1u << p.chainLog) : 0u)) * 2u;
        uint4* const w = (uint4*)(tables + (size_t)k * ZC_TABLE_STRIDE);
        for (u32 j = threadIdx.x; j < bytes / 16u; j += 256) w[j] = make_uint4(0, 0, 0, 0);
    }
}

__global__ void zj_synth_kernel(u8* dst, u32 bufSize, u64 firstIndex, u32 n) {
    u32 const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) zs_fill(dst + (size_t)i * bufSize, bufSize, firstIndex + i);
}

// Tight packing of a batch's variable-size outputs (for the multi-GPU gather): frame i moves from
// src[srcOff[i] .. +size[i]) to dst[dstOff[i] ..).  One workgroup per frame, 16 B per lane.
__global__ __launch_bounds__(256) void zj_pack_kernel(const u8* __restrict__ src, const u64* __restrict__ srcOff, const u64* __restrict__ sizes,
                                                       u8* __restrict__ dst, const u64* __restrict__ dstOff, u32 n) {
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        u64 const sz = sizes[i];
        if (sz > ((u64)1 << 40)) continue;              // error result: nothing to move
        const u8* s = src + srcOff[i]; u8* d = dst + dstOff[i];
        u32 const n16 = (u32)(sz >> 4);                  // (round 6 tried 16 aligned bytes per store behind a byte-wise head: 4.7 ms against 3.6 for the metric batch's 2 GB — not kept)
        for (u32 k = threadIdx.x; k < n16; k += 256) { u64 a = ld64(s + 16 * k), b = ld64(s + 16 * k + 8); st64(d + 16 * k, a); st64(d + 16 * k + 8, b); }
        for (u32 k = (n16 << 4) + threadIdx.x; k < (u32)sz; k += 256) d[k] = s[k];
    }
}

// ============================================================================ host state =======
namespace {
#define ZJ_ENC_LDS_BIG 131072u
// pass-0 LDS per level: the tables of > 16 KiB inputs up to 64 KiB (u16 positions)
#define ZJ_BIG_SLICE ((size_t)16384)    /* frames of 16-128 KiB at levels 4-8 per pass of the lane-per-frame route (655 KiB of records each) */
#define ZJ_LEVEL_MAX 8                 /* levels 1-3 on every path; level 4 (inputs <= 128 KiB) and levels 5-8 (<= 16 KiB), no dictionary, no explicit table sizes, on the HBM-table kernel */
size_t enc_lds_pass0(int level) {
    size_t const need = level == 1 ? (8192u * 2u) : (level == 2 ? (32768u * 2u) : (((1u << ZE_L3_HASHLOG) + (1u << ZE_L3_CHAINLOG)) * 2u));
    return need > sizeof(ZEEntropy) ? need : sizeof(ZEEntropy);
}
enum { ZJ_ROUTE_WAVE_HBM = ZJNI_ROUTE_WAVE_HBM, ZJ_ROUTE_FUSED = ZJNI_ROUTE_FUSED, ZJ_ROUTE_WAVE = ZJNI_ROUTE_WAVE, ZJ_ROUTE_LANE = ZJNI_ROUTE_LANE, ZJ_ROUTE_LANE_GATED = ZJNI_ROUTE_LANE_GATED,
       ZJ_ROUTE_RUN = ZJNI_ROUTE_RUN, ZJ_ROUTE_RUN_FLAGS = ZJNI_ROUTE_RUN_FLAGS, ZJ_ROUTE_HYBRID = ZJNI_ROUTE_HYBRID, ZJ_ROUTE_OTHER = ZJNI_ROUTE_OTHER };
// HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues per device (4 by default), and a hardware queue runs its packets in order: a stream that shares
// one with the stream of a 130 ms persistent kernel (the entropy kernel beside the match kernel, the match kernel itself) does not move until that kernel has left —
// measured with two host batches in flight: the second batch's pack kernel and D2H copies waited 130 ms behind the first's entropy kernel (profiles/r05/e_).  This
// library keeps up to ten streams busy at once (a caller's, three of its own per device, three per staging slot): a deployment that keeps two host batches in flight
// sets GPU_MAX_HW_QUEUES=16 in the environment the process is STARTED with.  The library itself never touches the environment: it is
// dlopen()ed into a running JVM, where setenv() races with every other thread's getenv() (rounds 4-5 called setenv from a static initialiser here).
// The host-pointer entries stage a batch through pinned host memory and a device area of the same layout.  Round 5: TWO such slots per device, each with its own
// streams and events, so that two host-pointer calls (two JVM threads in compressBatch0, or zjni_*_batch_begin twice) are in flight at once: while one call's kernels run,
// the other's sources cross the link one way and a third's frames the other — the lane pipeline wants a whole batch resident, so the overlap a single compress call cannot
// have (H2D 75 + kernels 143 + D2H 35 ms in sequence on the metric batch) comes from the next call.  The kernels of the two calls still follow each other (BatchOrder: they
// share the per-device scratch).  A third concurrent call waits for a slot.
#define ZJ_STAGE_SLOTS 2
struct StageSlot {
    std::mutex mu;
    u8* hPinned = nullptr; size_t hPinnedCap = 0;
    u8* dStage = nullptr; size_t dStageCap = 0;
    hipStream_t hostIn = nullptr, hostK = nullptr, hostOut = nullptr;       // H2D / kernels / D2H (the decompress entry runs slices of one call on all three at once: PCIe is full duplex)
    std::vector<hipEvent_t> stageEv;                  // one event per returned slice (compress)
    std::vector<hipEvent_t> pipeEv;                   // three events per slice: source landed, decoded, returned (decompress)
};
struct DevState {
    bool needLdsSet = false;                      // zj_enc_need_kernel's LDS attribute has been set on this device
    int lastRoute = 0;                            // ZJNI_ROUTE_* of the last large compress call (zjni_last_route)
    int ordinal = -1;
    int numCU = 0;
    int decGrid = 0, decDictGrid = 0, encGrid = 0;          // encGrid = largest encoder grid (level-1 LDS)
    int encGridLvl[4] = {0, 0, 0, 0};     // resident workgroups per level for pass 0
    int encGridBig = 0;                    // pass 1 (128 KiB LDS)
    int encGridSmall = 0;                  // entropy stage with small frames staged in LDS (ZE_SMALL_LDS_BYTES)
    // dictionary compress: slice s's entropy kernel (side stream) runs beside slice s+1's match kernel; two sets of records / lists / counters
    u8* wideBuf = nullptr; size_t wideBufCap = 0;     // lane-per-frame path of list B: [tables][frame scratch][meta] for one slice
    StageSlot* slot[ZJ_STAGE_SLOTS] = {};             // host-pointer entries: their staging areas, streams and events (below)
    std::atomic<unsigned>* slotTicket = nullptr;
    std::mutex* hostDecompMu = nullptr;
    u32* multiTables = nullptr; int multiGrid = 0;    // multi-block frames: frame-wide hash tables, one set per resident workgroup
    u32 lastPipeMax = 0;                              // what the last compress call passed to zj_pipe_route (diagnostics: zjni_last_lists)
    int pipeGrid = 0;                                 // ... and the resident workgroups of the pipelined kernel (two waves, two scratch slots each); 0: not available
    u8* cdBuf = nullptr; size_t cdBufCap = 0; size_t cdSliceCap = 0;
    u32* cdList = nullptr; size_t cdListCap = 0;
    hipEvent_t cdMatchDone[2] = {}, cdEncDone[2] = {};
    u32* counters = nullptr;       // [0] decode, [16] encode (separate cache lines)
    u8* decScratch = nullptr; int dseqHeavy = 0;
    volatile u32* encStat = nullptr; hipEvent_t evLists = nullptr;      // pinned: the classify kernel's list counts of the running compress call ([0] |A|, [1] |B|, [4] |C|), valid behind evLists — asked for only while the wide slice does not exist
    volatile u32* decStat = nullptr;                  // pinned: [0] |A|, [1] sequences of the last split-decode slice that ran (copied back asynchronously, read without waiting)
    u8* encScratch = nullptr;
    // staging for the host-pointer entries
    unsigned long long* prof = nullptr;    // 32 phase counters (16 decode + 16 encode) when ZJNI_PROFILE is set
    u32* encList = nullptr; size_t encListCap = 0;   // two index lists of encListCap entries each
    u8* splitBuf = nullptr; size_t splitBufCap = 0;    // lane-per-frame path: [tables][frame scratch][meta]
    int matchGrid = 0;
    int dseqGrid = 0, dexecGrid = 0, dlitGrid = 0;                  // split decode pipeline
    hipEvent_t tev[12] = {};                          // stage boundaries of the last batch calls (zjni_last_timing): [0,1] match, [2..6] decode stages, [8,9] wide match (last slice)
    bool tevWide = false;
    bool tevCompress = false, tevDecompress = false;
    hipStream_t sideStream = nullptr; hipEvent_t evFork = nullptr, evJoin = nullptr;
    hipStream_t waveStream = nullptr; hipEvent_t evJoinWave = nullptr; int waveGrid = 0;   // wave-per-frame matcher beside the lane-per-frame one
    // Level-3 tables of the lane pipeline (n x 384 KiB: 24 GiB for 65 536 frames) have to be zero when a call's match kernel starts.  Clearing them is
    // 5-6 ms of pure HBM writes in front of a kernel that waits on latency; so a call clears them for the NEXT one as soon as its own match
    // kernel is done, on a stream of its own (beside its entropy tail and whatever the host does next), and the next call over the same range
    // only waits for that.  clearedValid: the range [clearedPtr, + clearedBytes) is zero, or will be when evCleared fires.
    hipStream_t clearStream = nullptr; hipEvent_t evMatchDone = nullptr, evCleared = nullptr; u8* clearedPtr = nullptr; size_t clearedBytes = 0; bool clearedValid = false;
    // batch calls share the per-device scratch: they are enqueued under `enqueueMu`, and each call's kernels wait (on the
    // GPU) for the previous call's last kernel, whatever streams the callers use — many host threads may call at once
    std::mutex* enqueueMu = nullptr; hipEvent_t lastDone = nullptr; bool lastValid = false;
    u8* dsplitBuf = nullptr; size_t dsplitBufCap = 0;  // [tables][sequences][frame records][list A][list B]
    u8* dlitBuf = nullptr; size_t dlitBufCap = 0;      // literal slots of the split decode pipeline (stage 2b), one per frame of a slice
    u8* dmbBuf = nullptr; size_t dmbBufCap = 0;        // multi-block frames on the split pipeline: [block tables][blocks][frames][seq list][list M][record pool]
};
std::mutex g_mu;        // guards g_dev
std::vector<DevState> g_dev;
thread_local int t_dev = -1;

int dev_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

DevState* get_state(int ordinal) {
    std::lock_guard<std::mutex> lk(g_mu);
    int const n = dev_count();
    if (n <= 0 || ordinal < 0 || ordinal >= n) return nullptr;
    if ((int)g_dev.size() < n) g_dev.resize(n);
    DevState& d = g_dev[ordinal];
    if (hipSetDevice(ordinal) != hipSuccess) return nullptr;
    if (d.ordinal < 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess) return nullptr;
        d.numCU = prop.multiProcessorCount;
        int perCU = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, zj_decode_kernel, 64, 0) != hipSuccess || perCU < 1) perCU = 8;
        d.decGrid = d.numCU * perCU;
        {   int p2 = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&p2, zj_decode_dict_kernel, 64, 0) != hipSuccess || p2 < 1) p2 = 4;
            d.decDictGrid = d.numCU * (p2 < perCU ? p2 : perCU); }
        if (const char* ov = zj_tune("ZJNI_DEBUG_WG_PER_CU")) { int const v = atoi(ov); if (v >= 1 && v < perCU) d.decGrid = d.numCU * v; }   // occupancy experiments
        if (hipFuncSetAttribute((const void*)zj_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZJ_ENC_LDS_BIG) != hipSuccess) return nullptr;
        for (int lvl = 1; lvl <= 3; lvl++) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, zj_encode_kernel, 64, enc_lds_pass0(lvl)) != hipSuccess || perCU < 1) perCU = 1;
            d.encGridLvl[lvl] = d.numCU * perCU;
            if (const char* ov = zj_tune("ZJNI_DEBUG_WG_PER_CU")) { int const v = atoi(ov); if (v >= 1 && v < perCU) d.encGridLvl[lvl] = d.numCU * v; }
            if (d.encGridLvl[lvl] > d.encGrid) d.encGrid = d.encGridLvl[lvl];
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, zj_encode_kernel, 64, ZJ_ENC_LDS_BIG) != hipSuccess || perCU < 1) perCU = 1;
        d.encGridBig = d.numCU * perCU;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, zj_encode_kernel, 64, ZE_SMALL_LDS_BYTES) != hipSuccess || perCU < 1) perCU = 1;
        d.encGridSmall = d.numCU * perCU;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, zj_enc_match_kernel, 64, 0) != hipSuccess || perCU < 1) perCU = 4;
        d.matchGrid = d.numCU * perCU;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, zj_dec_seq_kernel, 64, 0) != hipSuccess || perCU < 1) perCU = 4;
        d.dseqGrid = d.numCU * perCU;
        // Resident waves of the sequence decode when the execution kernel runs beside it (zj_dec_heavy).  A lane takes its next frame from
        // a counter, so fewer waves than frames / 64 only means more frames per lane; the pipeline's time is flat from 1 to 2 waves per CU
        // (24.7-25.3 ms per 65 536 x 64 KiB) and the decode cells alive at a time shrink with the wave count: 1.5 waves per CU.
        d.dseqHeavy = d.numCU * 3 / 2;
        if (const char* ov = zj_tune("ZJNI_DSEQ_WAVES")) { int const v = atoi(ov); if (v >= 1) d.dseqHeavy = v; }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, zj_dec_exec_kernel, 64, ZD_SHARED_NO_FSE) != hipSuccess || perCU < 1) perCU = 8;
        d.dexecGrid = d.numCU * perCU;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, zj_dec_lit_kernel, 64, ZD_EXEC_LDS) != hipSuccess || perCU < 1) perCU = 8;
        d.dlitGrid = d.numCU * perCU;
        for (auto& e : d.tev) { if (hipEventCreate(&e) != hipSuccess) return nullptr; }
        d.enqueueMu = new std::mutex(); d.slotTicket = new std::atomic<unsigned>(0); d.hostDecompMu = new std::mutex();
        for (int k = 0; k < ZJ_STAGE_SLOTS; k++) d.slot[k] = new StageSlot();
        if (hipEventCreateWithFlags(&d.lastDone, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipStreamCreateWithFlags(&d.sideStream, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&d.evFork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&d.evJoin, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipMalloc(&d.counters, 1024) != hipSuccess) return nullptr;
        if (hipStreamCreateWithFlags(&d.waveStream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&d.evJoinWave, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipStreamCreateWithFlags(&d.clearStream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&d.evMatchDone, hipEventDisableTiming) != hipSuccess
            || hipEventCreateWithFlags(&d.evCleared, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipFuncSetAttribute((const void*)zj_enc_match_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ZWLds)) != hipSuccess) return nullptr;
        {   int w = 3; if (const char* ov = zj_tune("ZJNI_WAVE_PER_CU")) { int const v = atoi(ov); if (v >= 1 && v <= 3) w = v; }
            d.waveGrid = d.numCU * w; }
        for (int p = 0; p < 2; p++) if (hipEventCreateWithFlags(&d.cdMatchDone[p], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&d.cdEncDone[p], hipEventDisableTiming) != hipSuccess) return nullptr;
        {   void* hp = nullptr; if (hipHostMalloc(&hp, 64, hipHostMallocDefault) == hipSuccess) { memset(hp, 0, 64); d.decStat = (volatile u32*)hp; } }
        {   void* hp = nullptr; if (hipHostMalloc(&hp, 64, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&d.evLists, hipEventDisableTiming) == hipSuccess) { memset(hp, 0, 64); d.encStat = (volatile u32*)hp; } }
        if (hipMalloc(&d.decScratch, (size_t)(d.decGrid > d.dexecGrid ? d.decGrid : d.dexecGrid) * ZD_LIT_SCRATCH) != hipSuccess) return nullptr;
        if (hipMalloc(&d.encScratch, (size_t)d.encGrid * ZE_SCRATCH_BYTES) != hipSuccess) return nullptr;
        if (zj_tune("ZJNI_PROFILE")) { if (hipMalloc(&d.prof, 32 * 8) != hipSuccess || hipMemset(d.prof, 0, 32 * 8) != hipSuccess) return nullptr; }
        d.ordinal = ordinal;
    }
    return &d;
}

DevState* cur_state() {
    if (t_dev < 0) {
        int cur = 0;
        if (dev_count() <= 0) return nullptr;
        if (hipGetDevice(&cur) != hipSuccess) return nullptr;
        t_dev = cur;
    }
    return get_state(t_dev);
}

// ---- scratch budget (zjni_set_scratch_limit) ----
// The pipelines keep per-frame scratch in HBM (hash tables, sequence records, decode cells): four buffers per device, one per
// pipeline, allocated on first use and kept.  With a limit set, a batch is cut into slices whose scratch fits and a buffer that
// has to grow first evicts the other pipelines' buffers (after the device has drained).  0 = no limit (sized for 288 GB).
size_t g_scratch_limit = 0;
#define ZJ_SCRATCH_LIMIT_MIN ((size_t)4 << 30)
size_t scratch_total(const DevState* d) { return d->splitBufCap + d->wideBufCap + d->cdBufCap + d->dsplitBufCap + d->dlitBufCap + d->dmbBufCap; }
void scratch_free_all(DevState* d) {
    d->clearedValid = false;                                    // (hipFree drains the device, the clear stream included)
    if (d->splitBuf) (void)hipFree(d->splitBuf); d->splitBuf = nullptr; d->splitBufCap = 0;
    if (d->wideBuf) (void)hipFree(d->wideBuf); d->wideBuf = nullptr; d->wideBufCap = 0;
    if (d->cdBuf) (void)hipFree(d->cdBuf); d->cdBuf = nullptr; d->cdBufCap = 0; d->cdSliceCap = 0;
    if (d->dsplitBuf) (void)hipFree(d->dsplitBuf); d->dsplitBuf = nullptr; d->dsplitBufCap = 0;
    if (d->dmbBuf) (void)hipFree(d->dmbBuf); d->dmbBuf = nullptr; d->dmbBufCap = 0;
    if (d->dlitBuf) (void)hipFree(d->dlitBuf); d->dlitBuf = nullptr; d->dlitBufCap = 0;
}
// before a buffer holding `have` bytes is replaced by one of `need` bytes: false when even alone it would exceed the limit
bool scratch_make_room(DevState* d, size_t have, size_t need) {
    if (!g_scratch_limit) return true;
    if (need > g_scratch_limit) return false;
    if (scratch_total(d) - have + need <= g_scratch_limit) return true;
    if (hipDeviceSynchronize() != hipSuccess) return false;      // other pipelines' kernels may still be using what is about to go
    scratch_free_all(d);
    return true;
}
// frames per slice such that `perFrame` bytes of scratch each stay inside `share` of the limit (at least 4 096: below that the
// fused kernels run, which keep their scratch per workgroup)
size_t scratch_slice(size_t perFrame, size_t dflt, size_t shareDiv) {
    if (!g_scratch_limit) return dflt;
    size_t const f = (g_scratch_limit / shareDiv) / (perFrame ? perFrame : 1);
    return f < 4096 ? 4096 : (f < dflt ? f : dflt);
}

bool ensure_staging(StageSlot* d, size_t bytes) {
    if (!d->hostK) {
        if (hipStreamCreateWithFlags(&d->hostIn, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&d->hostK, hipStreamNonBlocking) != hipSuccess
            || hipStreamCreateWithFlags(&d->hostOut, hipStreamNonBlocking) != hipSuccess) { d->hostIn = d->hostK = d->hostOut = nullptr; return false; }
    }
    if (d->hPinnedCap < bytes) {
        if (d->hPinned) (void)hipHostFree(d->hPinned);
        if (d->dStage) (void)hipFree(d->dStage);
        d->hPinned = nullptr; d->dStage = nullptr; d->hPinnedCap = d->dStageCap = 0;
        size_t const cap = bytes + (bytes >> 2) + (1u << 20);
        if (hipHostMalloc(&d->hPinned, cap, hipHostMallocDefault) != hipSuccess) return false;
        if (hipMalloc(&d->dStage, cap) != hipSuccess) return false;
        d->hPinnedCap = d->dStageCap = cap;
    }
    return true;
}
// a free slot of the device, or the next one in turn when both are taken (held until the guard goes)
struct SlotLock {
    StageSlot* s = nullptr;
    explicit SlotLock(DevState* d) {
        for (int k = 0; k < ZJ_STAGE_SLOTS && !s; k++) if (d->slot[k]->mu.try_lock()) s = d->slot[k];
        if (!s) { s = d->slot[d->slotTicket->fetch_add(1u) % ZJ_STAGE_SLOTS]; s->mu.lock(); }
    }
    ~SlotLock() { if (s) s->mu.unlock(); }
    SlotLock(const SlotLock&) = delete; SlotLock& operator=(const SlotLock&) = delete;
};
// Host-side copies of the host-pointer entries (caller's buffers <-> pinned staging) run on a few threads: one memcpy stream moves
// ~10 GB/s, the PCIe link 57 each way — every byte of a batch is copied once by the host on each side, so these copies, not the link,
// bound the entries unless enough threads share them.  Default: the CPUs this process may really use (scheduler affinity and the cgroup
// quota, not the processors the box shows), at most 16; ZJNI_HOST_THREADS overrides (1 = the calling thread only).
#define ZJ_HOST_SLICE ((u64)256 << 20)
static int host_threads() {
    static int const t = []() {
        if (const char* ov = zj_env("ZJNI_HOST_THREADS")) { int const v = atoi(ov); return v < 1 ? 1 : (v > 64 ? 64 : v); }
        long v = 16;
        cpu_set_t set; CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { long const a = CPU_COUNT(&set); if (a >= 1 && a < v) v = a; }
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                    // cgroup v2: "<quota> <period>" or "max <period>"
            char q[32]; long per = 0;
            if (fscanf(f, "%31s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) { long const c = (atol(q) + per - 1) / per; if (c >= 1 && c < v) v = c; }
            fclose(f);
        }
        return (int)(v < 1 ? 1 : v);
    }();
    return t;
}
// fn(lo, hi) over index ranges of about equal bytes; off[0..n] = byte offsets of the items; `threads` = 0: host_threads()
template <class F>
static void par_ranges(const u64* off, size_t n, F fn, int threads = 0) {
    u64 const total = off[n] - off[0];
    int T = total < ((u64)8 << 20) ? 1 : (threads > 0 ? threads : host_threads());
    if ((size_t)T > n) T = (int)(n ? n : 1);
    if (T <= 1) { fn((size_t)0, n); return; }
    std::vector<size_t> cut((size_t)T + 1, n); cut[0] = 0;
    for (int t = 1; t < T; t++) cut[(size_t)t] = (size_t)(std::lower_bound(off, off + n + 1, off[0] + total / (u64)T * (u64)t) - off);
    for (int t = 1; t <= T; t++) if (cut[(size_t)t] < cut[(size_t)t - 1]) cut[(size_t)t] = cut[(size_t)t - 1];
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back([&, t]() { fn(cut[(size_t)t], cut[(size_t)t + 1]); });
    fn(cut[0], cut[1]);
    for (auto& x : th) x.join();
}

}  // namespace

// ============================================================================ C-ABI ============
extern "C" {

const char* zjni_version(void) { return ZJNI_VERSION_STRING; }
int zjni_device_count(void) { return dev_count(); }

int zjni_init(int ordinal) {
    if (dev_count() <= 0) return -(int)ZJNI_ERROR_no_device;
    DevState* d = get_state(ordinal);
    if (!d) return -(int)ZJNI_ERROR_no_device;
    t_dev = ordinal;
    return 0;
}

void zjni_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& d : g_dev) {
        if (d.ordinal < 0) continue;
        (void)hipSetDevice(d.ordinal);
        (void)hipFree(d.counters); if (d.decStat) { (void)hipHostFree((void*)d.decStat); d.decStat = nullptr; } if (d.encStat) { (void)hipHostFree((void*)d.encStat); d.encStat = nullptr; (void)hipEventDestroy(d.evLists); } (void)hipFree(d.decScratch); (void)hipFree(d.encScratch);
        if (d.encList) (void)hipFree(d.encList);
        if (d.splitBuf) (void)hipFree(d.splitBuf);
        if (d.dsplitBuf) (void)hipFree(d.dsplitBuf);
        if (d.dmbBuf) (void)hipFree(d.dmbBuf);
        if (d.dlitBuf) (void)hipFree(d.dlitBuf);
        if (d.wideBuf) (void)hipFree(d.wideBuf);
        if (d.multiTables) (void)hipFree(d.multiTables);
        if (d.cdBuf) (void)hipFree(d.cdBuf);
        if (d.cdList) (void)hipFree(d.cdList);
        for (int p = 0; p < 2; p++) { if (d.cdMatchDone[p]) (void)hipEventDestroy(d.cdMatchDone[p]); if (d.cdEncDone[p]) (void)hipEventDestroy(d.cdEncDone[p]); }
        if (d.sideStream) { (void)hipStreamDestroy(d.sideStream); (void)hipEventDestroy(d.evFork); (void)hipEventDestroy(d.evJoin); }
        if (d.clearStream) { (void)hipStreamDestroy(d.clearStream); (void)hipEventDestroy(d.evMatchDone); (void)hipEventDestroy(d.evCleared); }
        for (int k = 0; k < ZJ_STAGE_SLOTS; k++) if (StageSlot* const sl = d.slot[k]) {
            if (sl->hostIn) { (void)hipStreamDestroy(sl->hostIn); (void)hipStreamDestroy(sl->hostK); (void)hipStreamDestroy(sl->hostOut); }
            for (hipEvent_t e : sl->pipeEv) if (e) (void)hipEventDestroy(e);
            for (hipEvent_t e : sl->stageEv) if (e) (void)hipEventDestroy(e);
            if (sl->hPinned) (void)hipHostFree(sl->hPinned);
            if (sl->dStage) (void)hipFree(sl->dStage);
            delete sl;
        }
        if (d.waveStream) { (void)hipStreamDestroy(d.waveStream); (void)hipEventDestroy(d.evJoinWave); }
        d = DevState();
    }
}

unsigned zjni_isError(size_t r) { return r > ZJNI_ERR(256) ? 1u : 0u; }
int zjni_getErrorCode(size_t r) { return zjni_isError(r) ? (int)(0 - r) : 0; }
const char* zjni_getErrorName(size_t r) {
    // strings are libzstd's (N/common/error_private.c:15-65) so Java-side message checks keep passing
    switch (zjni_getErrorCode(r)) {
    case 0: return "No error detected";
    case 1: return "Error (generic)";
    case 10: return "Unknown frame descriptor";
    case 14: return "Unsupported frame parameter";
    case 16: return "Frame requires too much memory for decoding";
    case 20: return "Data corruption detected";
    case 22: return "Restored data doesn't match checksum";
    case 24: return "Header of Literals' block doesn't respect format specification";
    case 30: return "Dictionary is corrupted";
    case 32: return "Dictionary mismatch";
    case 40: return "Unsupported parameter";
    case 42: return "Parameter is out of bound";
    case 44: return "tableLog requires too much memory : unsupported";
    case 64: return "Allocation error : not enough memory";
    case 70: return "Destination buffer is too small";
    case 72: return "Src size is incorrect";
    case 200: return "zjni: no gfx950 device available";
    case 201: return "zjni: input outside the GPU path (use the CPU path)";
    default: return "Unspecified error code";
    }
}

size_t zjni_compressBound(size_t s) {
    return s + (s >> 8) + (s < (128u << 10) ? (((128u << 10) - s) >> 11) : 0);
}

unsigned long long zjni_getFrameContentSize(const void* srcv, size_t srcSize) {
    // N/decompress/zstd_decompress.c:447-557 / :588-603
    const u8* p = (const u8*)srcv;
    if (srcSize < 5) return (unsigned long long)-2;
    u32 const magic = ld32(p);
    if (magic != 0xFD2FB528u)       // a skippable frame needs its 8-byte header present (:476-478), then counts as size 0 (:596)
        return ((magic & 0xFFFFFFF0u) == 0x184D2A50u && srcSize >= 8) ? 0ull : (unsigned long long)-2;
    u32 const fhd = p[4], didc = fhd & 3, single = (fhd >> 5) & 1, fcsid = fhd >> 6;
    u32 const didSz = didc == 3 ? 4 : didc, fcsSz = fcsid == 0 ? single : (1u << fcsid);
    size_t pos = 5 + !single + didSz;
    if (srcSize < pos + fcsSz || (fhd & 8)) return (unsigned long long)-2;
    if (!single && (p[5] >> 3) + 10u > 31u) return (unsigned long long)-2;     // windowLog > ZSTD_WINDOWLOG_MAX (:517)
    if (fcsid == 0) return single ? p[pos] : (unsigned long long)-1;
    if (fcsid == 1) return ld16(p + pos) + 256;
    if (fcsid == 2) return ld32(p + pos);
    return ld64(p + pos);
}

/* debugging/profiling aid (not in the public header): copies the 32 phase-cycle counters and clears them */
int zjni_debug_read_profile(unsigned long long* out32) {
    DevState* d = cur_state();
    if (!d || !d->prof) return -1;
    if (hipMemcpy(out32, d->prof, 32 * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return hipMemset(d->prof, 0, 32 * 8) == hipSuccess ? 0 : -1;
}

/* Stage durations (ms) of the last large-batch device calls on this device, from HIP events recorded on the
 * caller's stream around the kernels: out[0] match-finder kernel (compress); out[1..4] decode stages prep /
 * sequence decode / execute / fused leftovers.  Blocks until those events have completed; entries whose stage
 * did not run are -1.  A profiling aid for bench.py's roofline line, not part of the data path. */
int zjni_last_timing(float* out5) {
    DevState* d = cur_state();
    if (!d) return -(int)ZJNI_ERROR_no_device;
    for (int i = 0; i < 5; i++) out5[i] = -1.0f;
    if (d->tevCompress) { if (hipEventSynchronize(d->tev[1]) == hipSuccess) (void)hipEventElapsedTime(&out5[0], d->tev[0], d->tev[1]); }
    if (d->tevDecompress && hipEventSynchronize(d->tev[6]) == hipSuccess) {
        for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&out5[1 + i], d->tev[2 + i], d->tev[3 + i]);
    }
    return 0;
}

/* the same five durations plus out8[5] = the wide match-finder kernel (frames > 64 KiB; last slice of the call); out8[6..7] reserved (-1) */
int zjni_last_timing2(float* out8) {
    DevState* d = cur_state();
    int const r = zjni_last_timing(out8);
    if (r) return r;
    out8[5] = out8[6] = out8[7] = -1.0f;
    if (d->tevWide && hipEventSynchronize(d->tev[9]) == hipSuccess) (void)hipEventElapsedTime(&out8[5], d->tev[8], d->tev[9]);
    return 0;
}

int zjni_last_route(void) { DevState* d = cur_state(); return d ? d->lastRoute : -(int)ZJNI_ERROR_no_device; }
// The last large compress call's three lists as the classification kernel filled them — out3[0] frames of the common launch (list A: what zjni_last_route names),
// out3[1] frames of the wide launch (list B: frames above 64 KiB, zj_enc_match_wide_kernel = ZJNI_ROUTE_WIDE), out3[2] frames of the multi-block / wave-per-frame
// kernel (list C).  Waits for the device: a diagnostic for benches and tests, not for the data path.
int zjni_last_lists(unsigned* out3) {
    DevState* d = cur_state();
    if (!d || !out3) return -(int)ZJNI_ERROR_no_device;
    u32 h[8];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, d->counters + 16, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return -(int)ZJNI_ERROR_no_device;
    out3[0] = h[0]; out3[1] = h[1]; out3[2] = h[4];
    if (d->lastRoute == ZJ_ROUTE_WAVE_HBM && d->lastPipeMax && h[4] && h[4] <= d->lastPipeMax && 2u * h[6] >= h[4]) d->lastRoute = ZJNI_ROUTE_PIPE;      // zj_pipe_route, restated for the diagnostics
    return 0;
}
// The last large decompress call: out4[0] frames of the single-block pipeline (list A), out4[1] frames the fused kernel decoded (list B: what no pipeline took, or
// handed over), out4[2] frames of the multi-block stages (list M), out4[3] their blocks.  Waits for the device (diagnostics: benches, tests).
int zjni_last_decode_lists(unsigned* out4) {
    DevState* d = cur_state();
    if (!d || !out4) return -(int)ZJNI_ERROR_no_device;
    u32 a[12], m[8];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(a, d->counters + 32, sizeof a, hipMemcpyDeviceToHost) != hipSuccess
        || hipMemcpy(m, d->counters + 232, sizeof m, hipMemcpyDeviceToHost) != hipSuccess) return -(int)ZJNI_ERROR_no_device;
    out4[0] = a[8]; out4[1] = a[1]; out4[2] = m[2]; out4[3] = m[0];
    return 0;
}
// ... and out5[4]: frames of one stored (raw / RLE) block that stage 1 copied itself (round 6; they are on none of the three lists)
int zjni_last_decode_lists2(unsigned* out5) {
    DevState* d = cur_state();
    if (!d || !out5) return -(int)ZJNI_ERROR_no_device;
    int const r = zjni_last_decode_lists(out5);
    if (r) return r;
    u32 s = 0;
    if (hipMemcpy(&s, d->counters + 32 + 11, sizeof s, hipMemcpyDeviceToHost) != hipSuccess) return -(int)ZJNI_ERROR_no_device;
    out5[4] = s;
    return 0;
}
const char* zjni_route_kernel(int route) {
    switch (route) {
    case ZJNI_ROUTE_WIDE: return "zj_enc_match_wide_kernel";
    case ZJNI_ROUTE_PIPE: return "zj_encode_pipe_kernel";
    case ZJNI_ROUTE_FUSED: return "zj_encode_kernel";
    case ZJNI_ROUTE_WAVE: return "zj_enc_match_wave_kernel";
    case ZJNI_ROUTE_LANE: case ZJNI_ROUTE_HYBRID: return "zj_enc_match_kernel";
    case ZJNI_ROUTE_LANE_GATED: return "zj_enc_match_gated_kernel";
    case ZJNI_ROUTE_RUN: case ZJNI_ROUTE_RUN_FLAGS: return "zj_enc_match_run_kernel";
    case ZJNI_ROUTE_WAVE_HBM: return "zj_encode_multi_kernel";
    default: return "";
    }
}
#ifndef ZJNI_BUILD_STAMP
#define ZJNI_BUILD_STAMP "unknown"
#endif
#ifdef ZJ_TUNING_KERNELS
const char* zjni_build_stamp(void) { return ZJNI_BUILD_STAMP "+tuning"; }
#else
const char* zjni_build_stamp(void) { return ZJNI_BUILD_STAMP; }
#endif

/* ---- resource policy ---- */
size_t zjni_set_scratch_limit(size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_scratch_limit = bytes == 0 ? 0 : (bytes < ZJ_SCRATCH_LIMIT_MIN ? ZJ_SCRATCH_LIMIT_MIN : bytes);
    return g_scratch_limit;
}
size_t zjni_scratch_bytes(void) {
    DevState* d = cur_state();
    if (!d) return 0;
    std::lock_guard<std::mutex> lk(*d->enqueueMu);
    return scratch_total(d);
}
size_t zjni_release_scratch(void) {
    DevState* d = cur_state();
    if (!d) return ZJNI_ERR(ZJNI_ERROR_no_device);
    std::lock_guard<std::mutex> lk(*d->enqueueMu);              // no batch call is being enqueued ...
    if (hipDeviceSynchronize() != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);   // ... and none is still running
    scratch_free_all(d);
    d->lastValid = false;
    return 0;
}

int zjni_kernel_info(int* decodeGrid, int* decodeLds, int* encodeGrid, int* encodeLds) {
    DevState* d = cur_state();
    if (decodeLds) *decodeLds = (int)sizeof(ZDecShared);
    if (encodeLds) *encodeLds = (int)(enc_lds_pass0(3) + sizeof(ZEncShared));
    if (!d) return -(int)ZJNI_ERROR_no_device;
    if (decodeGrid) *decodeGrid = d->decGrid;
    if (encodeGrid) *encodeGrid = d->encGrid;
    return 0;
}

struct zjni_cdict { int ordinal; u8* buf; unsigned dictID; int level; u32 strategy; };   // buf = [ZECDictDev][tagged tables][raw dictionary bytes]
struct zjni_ddict { int ordinal; u8* buf; size_t rawSize; unsigned dictID; };   // buf = [ZDDictDev][raw dictionary bytes]

// Scratch of the multi-block stages (zj_decode_split.h): tables and records per BLOCK, claimed on the device — ZD_MB_BLOCKS blocks (4 GiB of input at 128 KiB a block)
// and a pool of ZD_MB_SEQS records (a frame whose blocks or sequences find no room goes to the fused kernel).  ZJNI_DEC_MB=0 switches the stages off (A/B runs);
// none under a scratch limit, none with a dictionary.
#define ZD_MB_BLOCKS 65536u
#define ZD_MB_SEQS ((size_t)192 << 20)
#define ZD_MB_LIT_BYTES ((size_t)2 << 30)
struct ZDMbHost { ZDMbArgs a; u64* pool; u8* litPool; u32* procFlag; bool on; };
static ZDMbHost decode_mb_scratch(DevState* d, size_t n, hipStream_t st, bool haveDict) {
    ZDMbHost h; memset(&h, 0, sizeof h);
    int const mbEnv = (zj_env("ZJNI_DEC_MB") && atoi(zj_env("ZJNI_DEC_MB")) == 0) ? 0 : 1;
    if (hipMemsetAsync(d->counters + 232, 0, 64, st) != hipSuccess) return h;      // (also when the stages stay off: zjni_last_decode_lists reads them)
    if (!mbEnv || haveDict || g_scratch_limit) return h;
    size_t const tabB = (size_t)ZD_MB_BLOCKS * ZD_SPLIT_TAB_BYTES, blkB = (size_t)ZD_MB_BLOCKS * sizeof(ZDBlk), frB = n * sizeof(ZDFrameMB), listB2 = (size_t)ZD_MB_BLOCKS * 4, lmB = n * 4;
    size_t const poolOff = (tabB + blkB + frB + 2 * listB2 + 2 * lmB + 255) & ~(size_t)255;          // (lmB x 2: list M, stage 3's "done beside stage 2" flags)
    size_t const litOff = poolOff + ZD_MB_SEQS * 8;
    size_t const need = litOff + ZD_MB_LIT_BYTES + 256;
    if (d->dmbBufCap < need) {
        if (!scratch_make_room(d, d->dmbBufCap, need)) return h;
        if (d->dmbBuf) { if (hipStreamSynchronize(st) != hipSuccess) return h; (void)hipFree(d->dmbBuf); d->dmbBuf = nullptr; d->dmbBufCap = 0; }
        if (hipMalloc(&d->dmbBuf, need) != hipSuccess) { (void)hipGetLastError(); return h; }        // no room: the fused kernel serves these frames as before
        d->dmbBufCap = need;
    }
    u32* const ctr = d->counters + 232;               // [0] blocks, [1] seq list, [2] |M|, [3] work seq, [4..5] records, [6] work exec, [8] lit list, [10..11] literal bytes, [12] work lit
    if (hipMemsetAsync(ctr, 0, 64, st) != hipSuccess) return h;
    h.a.tabs = (u16*)d->dmbBuf; h.a.blks = (ZDBlk*)(d->dmbBuf + tabB); h.a.frames = (ZDFrameMB*)(d->dmbBuf + tabB + blkB);
    h.a.seqList = (u32*)(d->dmbBuf + tabB + blkB + frB); h.a.litList = h.a.seqList + ZD_MB_BLOCKS; h.a.listM = h.a.litList + ZD_MB_BLOCKS;
    h.a.ctr = ctr; h.a.blkCap = ZD_MB_BLOCKS; h.a.seqCap = ZD_MB_SEQS; h.a.minBlocks = 1;
    // stage 3 beside stage 2 (ZJNI_DEC_MB_OVERLAP=0: behind it, as before): a flag per frame says which ones it finished there
    h.procFlag = nullptr;
    if (!(zj_tune("ZJNI_DEC_MB_OVERLAP") && atoi(zj_tune("ZJNI_DEC_MB_OVERLAP")) == 0)) {
        u32* const q = h.a.listM + n;
        if (hipMemsetAsync(q, 0, lmB, st) == hipSuccess) h.procFlag = q;
    }
    h.pool = (u64*)(d->dmbBuf + poolOff);
    int const litEnv = (zj_tune("ZJNI_DEC_MB_LIT") && atoi(zj_tune("ZJNI_DEC_MB_LIT")) == 0) ? 0 : 1;       // stage 2b of these frames off (A/B runs)
    h.a.litCap = litEnv ? ZD_MB_LIT_BYTES : 0; if (!litEnv) h.a.litList = nullptr;
    h.litPool = d->dmbBuf + litOff;
    h.on = true;
    return h;
}
// stages 2 and 3 of the multi-block frames stage 1 put on list M; frames stage 3 hands over are appended to list B (count at listBCount) for the fused kernel behind
static void decode_mb_launch(DevState* d, const ZDMbHost& h, hipStream_t st, const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off, uint64_t* d_result,
                             u32* listB, u32* listBCount) {
    if (!h.on) return;
    // Stage 2b (literals) and — round 4 — stage 3 on the side stream BESIDE stage 2: the lane-per-block decode leaves most of every SIMD idle, and a frame's blocks
    // finish at different times (they differ 8 x in their number of sequences).  Stage 3 walks a frame's blocks in order as before, but starts each as soon as stage 2
    // has set its seqReady — a 1 MiB frame is then done about one block's execution behind its slowest block instead of eight behind the whole stage; a sweep pass
    // behind both takes the frames it gave up on (bounded waits).
    bool const lit = h.a.litList != nullptr, beside = h.procFlag != nullptr;
    bool const forked = (lit || beside) && hipEventRecord(d->evFork, st) == hipSuccess && hipStreamWaitEvent(d->sideStream, d->evFork, 0) == hipSuccess;
    hipLaunchKernelGGL(zj_dec_seq_mb_kernel, dim3((u32)d->dseqGrid), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (const u32*)h.a.seqList, (const u32*)(h.a.ctr + 1), h.a.ctr + 3,
                       (const u16*)h.a.tabs, h.pool, h.a.blks, (forked && beside) ? 1u : 0u);
    if (forked) {
        if (lit) hipLaunchKernelGGL(zj_dec_lit_mb_kernel, dim3((u32)d->dlitGrid), dim3(64), ZD_EXEC_LDS, d->sideStream, (const u8*)d_src, (const u64*)d_src_off, (const u32*)h.a.litList, (const u32*)(h.a.ctr + 8), h.a.ctr + 12,
                                    h.a.blks, h.litPool);
        if (beside)
            hipLaunchKernelGGL(zj_dec_exec_mb_kernel, dim3((u32)d->decGrid), dim3(64), 0, d->sideStream, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst, (const u64*)d_dst_off, (u64*)d_result,
                               (const u32*)h.a.listM, (const u32*)(h.a.ctr + 2), h.a.ctr + 6, (const ZDFrameMB*)h.a.frames, (const ZDBlk*)h.a.blks, (const u64*)h.pool, d->decScratch, listB, listBCount,
                               (const u8*)(lit ? h.litPool : nullptr), 1u, h.procFlag);
        if (hipEventRecord(d->evJoin, d->sideStream) != hipSuccess || hipStreamWaitEvent(st, d->evJoin, 0) != hipSuccess) { (void)hipStreamSynchronize(d->sideStream); }
    } else if (lit) {                                   // no fork: stage 2b on the main stream
        hipLaunchKernelGGL(zj_dec_lit_mb_kernel, dim3((u32)d->dlitGrid), dim3(64), ZD_EXEC_LDS, st, (const u8*)d_src, (const u64*)d_src_off, (const u32*)h.a.litList, (const u32*)(h.a.ctr + 8), h.a.ctr + 12,
                           h.a.blks, h.litPool);
    }
    bool const swept = forked && beside;
    hipLaunchKernelGGL(zj_dec_exec_mb_kernel, dim3((u32)d->decGrid), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst, (const u64*)d_dst_off, (u64*)d_result,
                       (const u32*)h.a.listM, (const u32*)(h.a.ctr + 2), swept ? h.a.ctr + 14 : h.a.ctr + 6, (const ZDFrameMB*)h.a.frames, (const ZDBlk*)h.a.blks, (const u64*)h.pool, d->decScratch, listB, listBCount,
                       (const u8*)(lit ? h.litPool : nullptr), swept ? 2u : 0u, swept ? h.procFlag : (u32*)nullptr);
}
static size_t decompress_batch_device_impl(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off,
                                           uint64_t* d_result, size_t n, const zjni_ddict* ddict, void* stream) {
    DevState* d = cur_state();
    const ZDDictDev* const ddDev = ddict ? (const ZDDictDev*)ddict->buf : nullptr;
    const u8* const ddRaw = ddict ? ddict->buf + sizeof(ZDDictDev) : nullptr;
    if (!d) return ZJNI_ERR(ZJNI_ERROR_no_device);
    if (n == 0) return 0;
    if (n > 0xFFFFFFFFull) return ZJNI_ERR(72);
    hipStream_t st = (hipStream_t)stream;
    u32 const grid = (u32)(n < (size_t)d->decGrid ? n : (size_t)d->decGrid);
    // Large batches: the three-stage pipeline (tANS decode lane-per-frame); whatever is not a simple frame, and
    // anything that fails on the way, ends on list B and goes through the fused kernel.  Small batches: fused only.
    size_t splitMin = 4096;
    if (const char* ov = zj_env("ZJNI_DSPLIT_MIN")) splitMin = (size_t)atoll(ov);
    if (n >= splitMin) {
        size_t const tabBytes = n * (size_t)ZD_SPLIT_TAB_BYTES, seqBytes = n * (size_t)ZD_SPLIT_SEQ_BYTES, metaBytes = n * sizeof(ZDMeta), listBytes = n * 4;
        size_t const need = tabBytes + seqBytes + metaBytes + 4 * listBytes + 256;
        if (d->dsplitBufCap < need) {
            if (!scratch_make_room(d, d->dsplitBufCap, need)) return ZJNI_ERR(64);
            if (d->dsplitBuf) { if (hipStreamSynchronize(st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device); (void)hipFree(d->dsplitBuf); d->dsplitBuf = nullptr; d->dsplitBufCap = 0; }
            if (hipMalloc(&d->dsplitBuf, need) != hipSuccess) return ZJNI_ERR(64);
            d->dsplitBufCap = need;
        }
        u16* const tabs = (u16*)d->dsplitBuf; u64* const seqs = (u64*)(d->dsplitBuf + tabBytes);
        ZDMeta* const metas = (ZDMeta*)(d->dsplitBuf + tabBytes + seqBytes);
        u32* const listA = (u32*)(d->dsplitBuf + tabBytes + seqBytes + metaBytes); u32* const listB = listA + n;
        u32* const doneList = listB + n; u32* const procFlag = doneList + n;       // completion queue of the sequence decode, frames the side pass executed
        u32* const c = d->counters + 32;          // [0] frames the pass beside the sequence decode executed, [1] |B|, [2] work prep, [3] work seq, [4] work exec, [5] work fused, [6] queue length, [7] work of the sweep pass, [8] |A|, [9] sequences in A, [10] work of the literal pass, [11] frames of one stored block copied by stage 1
        // Stage 2b (zd_lit_frame): one literal slot per frame, as large as the budget allows (at most a block); frames whose literals do
        // not fit a slot stay with the execution kernel.  Without a dictionary only (treeless literals need the dictionary's table).
        // ZJNI_DEC_LIT=0 switches the pass off (A/B runs); ZJNI_DEC_LIT_BYTES sets the budget (default 4 GiB, nothing under a scratch limit).
        u8* litSlots = nullptr; u32 litSlot = 0;
        {   int const litEnv = (zj_env("ZJNI_DEC_LIT") && atoi(zj_env("ZJNI_DEC_LIT")) == 0) ? 0 : 1;
            if (litEnv && !ddict && !g_scratch_limit) {
                size_t budget = (size_t)4 << 30; if (const char* ov = zj_tune("ZJNI_DEC_LIT_BYTES")) { long long const v = atoll(ov); budget = v <= 0 ? 0 : ((unsigned long long)v > ((unsigned long long)64 << 30) ? (size_t)64 << 30 : (size_t)v); }
                size_t slot = budget / n; if (slot > ZD_BLOCK_MAX) slot = ZD_BLOCK_MAX; slot &= ~(size_t)4095;
                if (slot >= 16384) {
                    size_t const needL = n * slot + 64;
                    if (d->dlitBufCap < needL) {
                        if (d->dlitBuf) { if (hipStreamSynchronize(st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device); (void)hipFree(d->dlitBuf); d->dlitBuf = nullptr; d->dlitBufCap = 0; }
                        if (hipMalloc(&d->dlitBuf, needL) == hipSuccess) d->dlitBufCap = needL; else { d->dlitBuf = nullptr; (void)hipGetLastError(); }
                    }
                    if (d->dlitBuf) { litSlots = d->dlitBuf; litSlot = (u32)slot; }
                }
            }
        }
        static int const overlapEnv = zj_tune("ZJNI_DEC_NO_OVERLAP") ? 0 : 1;
        // Running the execution kernel beside the sequence decode costs two cross-stream dependencies and an extra launch per slice
        // (~0.3 ms) — worth it only for frames with many sequences (zj_dec_heavy, decided on the device).  The host skips the set-up
        // when the last slice it has statistics for was light: the statistics arrive asynchronously and are never waited for, so a
        // change of workload is followed one call late, which costs time, never correctness.
        u32 const statA = d->decStat ? d->decStat[0] : 0u, statS = d->decStat ? d->decStat[1] : 0u;
        int const overlap = overlapEnv && !(statA > 0u && statS < 512u * statA);
        if (hipMemsetAsync(c, 0, 48, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
        (void)hipEventRecord(d->tev[2], st);
        ZDMbHost const mb = decode_mb_scratch(d, n, st, ddict != nullptr);       // frames that are not simple: multi-block, no content size (the stream classes')
        if (ddict) hipLaunchKernelGGL(zj_dec_prep_kernel_t<true>, dim3(grid), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (const u64*)d_dst_off,
                                      (u32)n, c + 2, tabs, metas, listA, listB, c, ddDev, overlap ? doneList : (u32*)nullptr, procFlag, mb.a, 0u, (u8*)d_dst, (u64*)d_result);
        else hipLaunchKernelGGL(zj_dec_prep_kernel, dim3(grid), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (const u64*)d_dst_off,
                                (u32)n, c + 2, tabs, metas, listA, listB, c, ddDev, overlap ? doneList : (u32*)nullptr, procFlag, mb.a, 0u, (u8*)d_dst, (u64*)d_result);
        (void)hipEventRecord(d->tev[3], st);
        u32 const waves = (u32)((n + 63) / 64);
        u32 gridX = (u32)(n < (size_t)d->dexecGrid ? n : (size_t)d->dexecGrid);
        if (const char* ov = zj_tune("ZJNI_DEXEC_PER_CU")) { int const v = atoi(ov); if (v >= 1 && (u32)(v * d->numCU) < gridX) gridX = (u32)(v * d->numCU); }      // (tuning builds: fewer execution / literal workgroups beside the sequence decode)
        // The execution kernel (literals + LZ77 copy, wave per frame) runs on a side stream BESIDE the sequence decode and takes
        // frames in the order they finish there: the lane-per-frame decode is a dependent chain per frame (one wave per SIMD, mostly
        // waiting), so the two share the CUs instead of following each other.  A sweep pass afterwards takes what the side kernel
        // did not get to (its waits are bounded): completion never depends on the two kernels being co-scheduled.
        bool const fork = overlap || litSlots;
        if (fork && (hipEventRecord(d->evFork, st) != hipSuccess || hipStreamWaitEvent(d->sideStream, d->evFork, 0) != hipSuccess)) return ZJNI_ERR(ZJNI_ERROR_no_device);
        if (litSlots) {                                 // beside the sequence decode, ahead of the mode-1 execution pass on the same side stream
            u32 gridL = (u32)(n < (size_t)d->dlitGrid ? n : (size_t)d->dlitGrid);
            if (const char* ov = zj_tune("ZJNI_DLIT_PER_CU")) { int const v = atoi(ov); if (v >= 1 && (u32)(v * d->numCU) < gridL) gridL = (u32)(v * d->numCU); }
            hipLaunchKernelGGL(zj_dec_lit_kernel, dim3(gridL), dim3(64), ZD_EXEC_LDS, d->sideStream, (const u8*)d_src, (const u64*)d_src_off, (const u32*)listA,
                               (const u32*)(c + 8), c + 10, metas, litSlots, litSlot, d->prof);
        }
        hipLaunchKernelGGL(zj_dec_seq_kernel, dim3(waves < (u32)d->dseqGrid ? waves : (u32)d->dseqGrid), dim3(64), 0, st, (const u8*)d_src,
                           (const u64*)d_src_off, (const u32*)listA, (const u32*)(c + 8), c + 3, (const u16*)tabs, seqs, metas, ddDev,
                           overlap ? doneList : (u32*)nullptr, c + 6, (u32)d->dseqHeavy);
        (void)hipEventRecord(d->tev[4], st);
        if (litSlots && !overlap && (hipEventRecord(d->evJoin, d->sideStream) != hipSuccess || hipStreamWaitEvent(st, d->evJoin, 0) != hipSuccess)) {
            (void)hipStreamSynchronize(d->sideStream); (void)hipStreamSynchronize(st);          // the side kernel must not outlive this call's claim on the scratch
            return ZJNI_ERR(ZJNI_ERROR_no_device);
        }
        for (int pass = overlap ? 1 : 0; pass <= (overlap ? 2 : 0); pass++) {
            hipStream_t const es = pass == 1 ? d->sideStream : st;
            u32* const work = pass == 2 ? c + 7 : c + 4;
            if (ddict) hipLaunchKernelGGL(zj_dec_exec_kernel_t<true>, dim3(gridX), dim3(64), ZD_SHARED_NO_FSE, es,
                               (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst, (const u64*)d_dst_off, (u64*)d_result, (const u32*)listA,
                               (const u32*)(c + 8), work, metas, (const u64*)seqs, d->decScratch, listB, c + 1, d->prof, ddDev, ddRaw, (u32)pass, (const u32*)doneList, procFlag, litSlots, litSlot, c);
            else hipLaunchKernelGGL(zj_dec_exec_kernel, dim3(gridX), dim3(64), ZD_SHARED_NO_FSE, es,
                               (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst, (const u64*)d_dst_off, (u64*)d_result, (const u32*)listA,
                               (const u32*)(c + 8), work, metas, (const u64*)seqs, d->decScratch, listB, c + 1, d->prof, ddDev, ddRaw, (u32)pass, (const u32*)doneList, procFlag, litSlots, litSlot, c);
            if (pass == 1 && (hipEventRecord(d->evJoin, d->sideStream) != hipSuccess || hipStreamWaitEvent(st, d->evJoin, 0) != hipSuccess)) {
                (void)hipStreamSynchronize(d->sideStream); (void)hipStreamSynchronize(st);          // the side kernel must not outlive this call's claim on the scratch
                return ZJNI_ERR(ZJNI_ERROR_no_device);
            }
        }
        if (d->decStat) (void)hipMemcpyAsync((void*)d->decStat, c + 8, 8, hipMemcpyDeviceToHost, st);
        decode_mb_launch(d, mb, st, d_src, d_src_off, d_dst, d_dst_off, d_result, listB, c + 1);
        (void)hipEventRecord(d->tev[5], st);
        if (ddict) hipLaunchKernelGGL(zj_decode_dict_kernel, dim3(grid < (u32)d->decDictGrid ? grid : (u32)d->decDictGrid), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                           (const u64*)d_dst_off, (u64*)d_result, (u32)n, c + 5, d->decScratch, d->prof, (const u32*)listB, (const u32*)(c + 1), ddDev, ddRaw);
        else hipLaunchKernelGGL(zj_decode_kernel, dim3(grid), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                           (const u64*)d_dst_off, (u64*)d_result, (u32)n, c + 5, d->decScratch, d->prof, (const u32*)listB, (const u32*)(c + 1), ddDev, ddRaw);
        (void)hipEventRecord(d->tev[6], st); d->tevDecompress = true;
        return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
    }
    if (hipMemsetAsync(d->counters, 0, 4, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
    // Small batches: frames of two blocks and more take the block stages (stage 1 in its multi-block-only mode; a frame is a chain of ~40 000 dependent steps per MiB on
    // the fused kernel, its blocks decode side by side here); everything else, and whatever the stages hand over, is the fused kernel's as before.
    if (!ddict) {
        ZDMbHost mb = decode_mb_scratch(d, n, st, false);
        size_t const lbBytes = n * 4;
        if (mb.on && d->dsplitBufCap < lbBytes + 256) {                                   // list B of this path lives at the start of the split pipeline's buffer
            if (d->dsplitBuf) { if (hipStreamSynchronize(st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device); (void)hipFree(d->dsplitBuf); d->dsplitBuf = nullptr; d->dsplitBufCap = 0; }
            if (hipMalloc(&d->dsplitBuf, lbBytes + 256) != hipSuccess) { (void)hipGetLastError(); mb.on = false; } else d->dsplitBufCap = lbBytes + 256;
        }
        if (mb.on) {
            u32* const c = d->counters + 32; u32* const listB = (u32*)d->dsplitBuf;
            if (hipMemsetAsync(c, 0, 48, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
            mb.a.minBlocks = 2;
            (void)hipEventRecord(d->tev[2], st);
            hipLaunchKernelGGL(zj_dec_prep_kernel, dim3(grid), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (const u64*)d_dst_off,
                               (u32)n, c + 2, (u16*)nullptr, (ZDMeta*)nullptr, (u32*)nullptr, listB, c, (const ZDDictDev*)nullptr, (u32*)nullptr, (u32*)nullptr, mb.a, 1u, (u8*)d_dst, (u64*)d_result);
            (void)hipEventRecord(d->tev[3], st); (void)hipEventRecord(d->tev[4], st);
            decode_mb_launch(d, mb, st, d_src, d_src_off, d_dst, d_dst_off, d_result, listB, c + 1);
            (void)hipEventRecord(d->tev[5], st);
            hipLaunchKernelGGL(zj_decode_kernel, dim3(grid), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                               (const u64*)d_dst_off, (u64*)d_result, (u32)n, c + 5, d->decScratch, d->prof, (const u32*)listB, (const u32*)(c + 1), (const ZDDictDev*)nullptr, (const u8*)nullptr);
            (void)hipEventRecord(d->tev[6], st); d->tevDecompress = true;
            return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
        }
    }
    if (ddict) {
        u32 const gridD = (u32)(n < (size_t)d->decDictGrid ? n : (size_t)d->decDictGrid);
        hipLaunchKernelGGL(zj_decode_dict_kernel, dim3(gridD), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                           (const u64*)d_dst_off, (u64*)d_result, (u32)n, d->counters, d->decScratch, d->prof, (const u32*)nullptr, (const u32*)nullptr, ddDev, ddRaw);
    } else
    hipLaunchKernelGGL(zj_decode_kernel, dim3(grid), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                       (const u64*)d_dst_off, (u64*)d_result, (u32)n, d->counters, d->decScratch, d->prof, (const u32*)nullptr, (const u32*)nullptr, ddDev, ddRaw);
    return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
}
// The per-frame scratch of the pipelines (176 KiB decode, 416 KiB compress) is sized per call; very large batches go
// through it in slices of ZJ_CHUNK_FRAMES on the same stream (stream order makes the reuse safe), so a batch of a
// million buffers needs no more scratch than one of 65 536.
#define ZJ_CHUNK_FRAMES 65536u
// one batch call at a time per device on the host side, and in order on the GPU side
struct BatchOrder {
    DevState* d; hipStream_t st;
    BatchOrder(DevState* dev, void* stream) : d(dev), st((hipStream_t)stream) {
        if (!d) return;
        d->enqueueMu->lock();
        if (d->lastValid) (void)hipStreamWaitEvent(st, d->lastDone, 0);
    }
    ~BatchOrder() { if (!d) return; d->lastValid = (hipEventRecord(d->lastDone, st) == hipSuccess); d->enqueueMu->unlock(); }
};
static size_t decompress_chunked(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off,
                                 uint64_t* d_result, size_t n, const zjni_ddict* ddict, void* stream) {
    BatchOrder order(cur_state(), stream);
    size_t const chunk = scratch_slice((size_t)ZD_SPLIT_TAB_BYTES + ZD_SPLIT_SEQ_BYTES + sizeof(ZDMeta) + 8, ZJ_CHUNK_FRAMES, 1);
    for (size_t at = 0; at < n || at == 0; at += chunk) {
        size_t const m = n - at < chunk ? n - at : chunk;
        size_t const r = decompress_batch_device_impl(d_src, d_src_off + at, d_dst, d_dst_off + at, d_result + at, m, ddict, stream);
        if (r != 0 || n == 0) return r;
    }
    return 0;
}
size_t zjni_decompress_batch_device(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off,
                                    uint64_t* d_result, size_t n, void* stream) {
    return decompress_chunked(d_src, d_src_off, d_dst, d_dst_off, d_result, n, nullptr, stream);
}
size_t zjni_decompress_batch_device_usingDDict(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off,
                                               uint64_t* d_result, size_t n, const zjni_ddict* ddict, void* stream) {
    if (ddict && ddict->ordinal != t_dev && t_dev >= 0) return ZJNI_ERR(32);      // digested on another device
    return decompress_chunked(d_src, d_src_off, d_dst, d_dst_off, d_result, n, ddict, stream);
}

// ZSTD_createDDict (N/decompress/zstd_ddict.c:36-130; ZstdDictDecompress.init, N/jni_fast_zstd.c:56-75): the raw
// dictionary goes to HBM once and one workgroup digests it there.  NULL when the device or the dictionary is bad.
zjni_ddict* zjni_createDDict(const void* dict, size_t dictSize) {
    DevState* d = cur_state();
    if (!d || !dict || dictSize == 0 || dictSize > 0x7FFFFFFFull) return nullptr;
    zjni_ddict* dd = new zjni_ddict{t_dev, nullptr, dictSize, 0};
    if (hipMalloc(&dd->buf, sizeof(ZDDictDev) + dictSize + 16) != hipSuccess) { delete dd; return nullptr; }
    ZDDictDev head;
    bool ok = hipMemcpy(dd->buf + sizeof(ZDDictDev), dict, dictSize, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(zj_ddict_digest_kernel, dim3(1), dim3(64), 0, 0, (const u8*)(dd->buf + sizeof(ZDDictDev)), (u32)dictSize, (ZDDictDev*)dd->buf);
        ok = hipMemcpy(&head, dd->buf, 64, hipMemcpyDeviceToHost) == hipSuccess && head.status == 0;
    }
    if (!ok) { (void)hipFree(dd->buf); delete dd; return nullptr; }
    dd->dictID = head.dictID;
    return dd;
}
size_t zjni_freeDDict(zjni_ddict* dd) {
    if (!dd) return 0;
    (void)hipSetDevice(dd->ordinal);
    (void)hipFree(dd->buf);
    delete dd;
    return 0;
}
unsigned zjni_getDictID_fromDDict(const zjni_ddict* dd) { return dd ? dd->dictID : 0u; }

// The `checksum` argument of the advanced / dictionary entries is a flag word (include/zjni_amd.h ZJNI_FRAME_*): 1 alone is what it
// always meant; the boolean entries (zjni_compress*2) normalise their argument before they get here.
static inline void zj_dbg_sync(const char* what) {        // ZJNI_DEBUG_SYNC=1: drain the device after a launch and say so (finding the kernel that does not return)
    static int const on = zj_env("ZJNI_DEBUG_SYNC") ? 1 : 0;
    if (!on) return;
    fprintf(stderr, "[zjni] waiting for %s ...", what); fflush(stderr);
    hipError_t const e = hipDeviceSynchronize();
    fprintf(stderr, " %s\n", e == hipSuccess ? "done" : hipGetErrorString(e)); fflush(stderr);
}
// Frame-wide match-finder tables of the wave-per-frame kernels (zj_encode_multi_kernel, zj_encode_cdict_copy_kernel): 1 MiB per resident
// workgroup, allocated on first use.  Two workgroups per SIMD (8 per CU): the wave matcher of multi-block frames is a chain of round trips
// and LDS steps per frame, a second wave beside it fills the gaps (2 048 x 1 MiB frames: 656 -> 412 ms); the LDS of the entropy stage
// (18 KiB per workgroup) does not admit a third.  ZJNI_MULTI_PER_CU overrides.
static bool ensure_multi_tables(DevState* d) {
    if (d->multiTables) return true;
    int perCU = 8; if (const char* ov = zj_tune("ZJNI_MULTI_PER_CU")) { int const v = atoi(ov); if (v >= 1 && v <= 16) perCU = v; }
    int fit = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, zj_encode_multi_kernel, 64, sizeof(ZEEntropy)) == hipSuccess && fit >= 1 && fit < perCU) perCU = fit;
    d->multiGrid = d->numCU * perCU; if (d->multiGrid > d->encGrid) d->multiGrid = d->encGrid;     // encScratch has one slot per resident entropy workgroup
    if (hipMalloc(&d->multiTables, (size_t)d->multiGrid * ZE_MULTI_TABLE_BYTES) != hipSuccess) { d->multiTables = nullptr; (void)hipGetLastError(); return false; }
    {   int pfit = 0; size_t const lds = sizeof(ZEEntropy) + ZJ_PIPE_LDS_P;
        d->pipeGrid = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pfit, zj_encode_pipe_kernel, 128, lds + 16u) == hipSuccess && pfit >= 1) {
            int gp = d->numCU * pfit;
            if (gp > d->multiGrid) gp = d->multiGrid;             // a table set per workgroup
            if (gp > d->encGrid / 2) gp = d->encGrid / 2;         // two scratch slots per workgroup
            d->pipeGrid = gp;
        } else (void)hipGetLastError(); }
    return true;
}
static inline u32 zj_frame_flags(int word) { return (u32)word & ZE_FLAG_MASK; }
static size_t compress_batch_device_impl(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off,
                                         uint64_t* d_result, size_t n, int levelWord, u32 flags, void* stream) {
    DevState* d = cur_state();
    if (!d) return ZJNI_ERR(ZJNI_ERROR_no_device);
    int const level = (int)ZE_LW_LEVEL((u32)levelWord);       // kernels take the level word (level | hashLog << 8 | chainLog << 16)
    bool const tuned = ZE_LW_TUNED((u32)levelWord);
    if (level < 1 || level > ZJ_LEVEL_MAX) return ZJNI_ERR(42);
    if (n == 0) return 0;
    if (n > 0xFFFFFFFFull) return ZJNI_ERR(72);
    hipStream_t st = (hipStream_t)stream;
    d->lastRoute = ZJ_ROUTE_OTHER;
    bool wasCleared = d->clearedValid; d->clearedValid = false;      // whatever this call does with the scratch, the promise of the previous one ends here
    if (d->encListCap < n) {                      // grows rarely; the only synchronous step of this entry
        if (d->encList) { if (hipStreamSynchronize(st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device); (void)hipFree(d->encList); d->encList = nullptr; d->encListCap = 0; }
        size_t const cap = n + (n >> 2) + 1024;
        if (hipMalloc(&d->encList, 4 * cap * sizeof(u32)) != hipSuccess) return ZJNI_ERR(64);
        d->encListCap = cap;
    }
    u32* const ctr = d->counters + 16;            // [0] |A|, [1] |B|, [2] work A, [3] work B
    u32* const listA = d->encList; u32* const listB = d->encList + d->encListCap; u32* const listS = d->encList + 2 * d->encListCap; u32* const listC = d->encList + 3 * d->encListCap;
    if (hipMemsetAsync(ctr, 0, 32, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);     // [0] |A|, [1] |B|, [2] work A, [3] work B, [4] |C|, [5] work C
    if (level > 3 && tuned) return ZJNI_ERR(42);
    // Level 3, batches that cannot fill the lane pipeline: a lane of that pipeline takes ~46 000 rounds for a 64 KiB text frame whatever the batch size (a
    // floor of ~96 ms per call), while a wave of the multi-block kernel parses the same frame in ~6 000 windows (zj_match_wavex.h: ~17 ms) —
    // so below ZJNI_L3_WAVE_MAX frames (default 8 192: four rounds of the 2 048 resident waves) every frame goes there, with the level's own tables or the caller's.
    // (ZJNI_SPLIT_MIN — "the lane pipelines from this batch size on" — is honoured: with it set the wave route ends there.)
    size_t l3WaveMax = 8192; if (const char* ov = zj_env("ZJNI_L3_WAVE_MAX")) l3WaveMax = (size_t)atoll(ov);
    if (const char* ov = zj_env("ZJNI_SPLIT_MIN")) { size_t const v = (size_t)atoll(ov); if (v < l3WaveMax) l3WaveMax = v; }
    bool const l3wave = level == 3 && tuned && n < l3WaveMax && !g_scratch_limit;      // (tuned: table sizes beyond the LDS — the level's own 16 / 15 included; explicit 14 / 13 keeps the LDS matcher of small batches)
    if (l3wave) levelWord |= (int)ZE_LW_WAVE_ROUTE;
    u32 const ldsA = level > 3 ? 0u : (u32)enc_lds_pass0(level);
    hipLaunchKernelGGL(zj_enc_classify_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, (const u64*)d_src_off, (u64*)d_result,
                       (u32)n, (u32)levelWord, ldsA, ctr, listA, listB, listC);
    // The wide slice (list B: frames of 64 KiB + 1 .. 128 KiB, ~1.1 MiB of tables and records per frame of a 65 536-frame slice) is allocated when a call HAS such
    // frames (rounds 1-5: by any large level 1-3 call, 72 GiB).  While it does not exist the list counts come back to pinned memory behind an event, and the call waits
    // for them where it would allocate — by then list A's kernels are queued, so the device is not kept waiting; once it exists nothing is asked.
    bool listsAsked = false;
    if (level <= 3 && d->encStat && d->wideBufCap == 0 && !l3wave)
        listsAsked = hipMemcpyAsync((void*)d->encStat, ctr, 24, hipMemcpyDeviceToHost, st) == hipSuccess && hipEventRecord(d->evLists, st) == hipSuccess;
    // levels 4-8, frames of 16-128 KiB: one lane per frame when the batch is large (below) and the scratch budget has room for a table set
    // per lane slot, else one wave per frame (here)
    // (level 4 is double-fast: its frames above 16 KiB stay on list C, where the wave matcher of zj_match_wavex.h parses them — two waves per
    //  SIMD each on its own frame beat 256 waves of one-lane parses; ZJNI_L4_LANES=1 keeps the lane-slot route selectable for A/B runs)
    bool const l4wave = level == 4 && !(zj_tune("ZJNI_L4_LANES") && atoi(zj_tune("ZJNI_L4_LANES")) == 1);
    bool const bigLanes = level > 3 && !l4wave && n >= 4096 && (!g_scratch_limit || g_scratch_limit >= ((size_t)48 << 30));
    if (!bigLanes)
    {   // list C: multi-block frames (levels 1-3) and the single-block frames of levels 4-8 above 16 KiB.  The launch is unconditional (an empty list costs an empty kernel); its tables are a fixed 1 MiB per resident workgroup.
        if (!ensure_multi_tables(d)) return ZJNI_ERR(64);
        u32 const gc = (u32)(n < (size_t)d->multiGrid ? n : (size_t)d->multiGrid);
        // blocks of multi-block frames: the wave matchers (zj_match_wavex.h: double-fast at level 3, fast at levels 1-2); ZJNI_MULTI_WAVE=0 keeps the one-lane parse selectable for A/B runs, =2 the wave matcher without staged spans
        u32 multiSerial = 0; if (const char* ov = zj_tune("ZJNI_MULTI_WAVE")) { int const v = atoi(ov); multiSerial = v == 0 ? ZE_FLAG_MULTI_SERIAL : (v == 2 ? ZE_FLAG_MULTI_NOCARRY : 0u); }
        // levels 1-2 (fast strategy): the one-lane parse unless ZJNI_MULTI_WAVE_FAST=1 — the wave version (ZWaveF) is exact but measured slower there
        // (2 048 x 512 KiB at level 1: 156 ms against 138; one 64 KiB table per frame stays in the L2 / Infinity Cache and an iteration of the one-lane loop is one short trip)
        {   const char* const ov = zj_tune("ZJNI_MULTI_WAVE_FAST"); if (!(ov && atoi(ov) == 1)) multiSerial |= ZE_FLAG_MULTI_FAST_SERIAL; }
        // Batches that leave wave slots empty (every frame resident at once and room to spare) take the pipelined kernel: a parse wave a block ahead of an entropy wave per
        // frame — 1 024 x 1 MiB: the frame's chain is the parse alone.  With the slots full the one-wave kernel does the same work in fewer wave-milliseconds.
        // ZJNI_PIPE_MAX (tuning builds): the largest batch that takes it (0: never).
        u32 pipeMax = level <= 3 ? (u32)d->pipeGrid : 0u; if (const char* ov = zj_tune("ZJNI_PIPE_MAX")) { long long const v = atoll(ov); pipeMax = v <= 0 ? 0u : ((size_t)v < (size_t)d->pipeGrid ? (u32)v : (u32)d->pipeGrid); }
        d->lastPipeMax = pipeMax;
        if (pipeMax) {
            u32 const gp = (u32)(n < (size_t)pipeMax ? n : (size_t)pipeMax);
            hipLaunchKernelGGL(zj_encode_pipe_kernel, dim3(gp), dim3(128), (u32)(((sizeof(ZEEntropy) + 15u) & ~(size_t)15) + ZJ_PIPE_LDS_P), st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst, (const u64*)d_dst_off,
                               (u64*)d_result, (u32)levelWord, (const u32*)listC, (const u32*)(ctr + 4), ctr + 5, d->encScratch, d->multiTables, flags | multiSerial, (u32)sizeof(ZEEntropy), pipeMax);
        }
        hipLaunchKernelGGL(zj_encode_multi_kernel, dim3(gc), dim3(64), (u32)sizeof(ZEEntropy), st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst, (const u64*)d_dst_off,
                           (u64*)d_result, (u32)levelWord, (const u32*)listC, (const u32*)(ctr + 4), ctr + 5, d->encScratch, d->multiTables, flags | multiSerial, (u32)sizeof(ZEEntropy), pipeMax);
    }
    if (l3wave) {                                  // the whole batch was list C's
        d->lastRoute = ZJ_ROUTE_WAVE_HBM; d->tevCompress = false;
        d->clearedValid = wasCleared;              // the lane pipeline's scratch was not touched: what the previous call promised about its tables still holds
        return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
    }
    // Everything below places something in splitBuf.  A clear the previous call queued on clearStream (24 GiB: 5-6 ms) may still be running — BatchOrder only
    // chains `st` to `st` — so EVERY user of the buffer first orders itself behind it, not only the branch that skips its own memset (levels 4-8, the LDS matcher
    // of small batches and a level 1-3 call with larger tables used to start under the running clear: records and table entries zeroed mid-parse).
    if (wasCleared && hipStreamWaitEvent(st, d->evCleared, 0) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
    if (level > 3) {
        // list A of levels 4-8 (frames <= 16 KiB): chain parsers lane-per-frame, then the entropy kernel on their records
        u32 const maxSrcC = ZE_CHAIN_MAX_SRC;
        size_t const tablesBytes = n * (size_t)ZE_CHAIN_TABLE_BYTES, fsBytes = n * (size_t)ZE_FRAME_STRIDE(maxSrcC), metaBytes = n * 12;
        size_t const need = tablesBytes + fsBytes + metaBytes + 256;
        if (d->splitBufCap < need) {
            if (!scratch_make_room(d, d->splitBufCap, need)) return ZJNI_ERR(64);
            if (d->splitBuf) { if (hipStreamSynchronize(st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device); (void)hipFree(d->splitBuf); d->splitBuf = nullptr; d->splitBufCap = 0; wasCleared = false; }
            if (hipMalloc(&d->splitBuf, need) != hipSuccess) return ZJNI_ERR(64);
            d->splitBufCap = need;
        }
        u8* const tables = d->splitBuf; u8* const fs = d->splitBuf + tablesBytes; u32* const mt = (u32*)(fs + fsBytes);
        u32* const mctr = d->counters + 24;
        if (hipMemsetAsync(mctr, 0, 16, st) != hipSuccess || hipMemsetAsync(tables, 0, tablesBytes, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
        u32 const waves = (u32)((n + 63) / 64);
        hipLaunchKernelGGL(zj_enc_match_chain_kernel, dim3(waves < (u32)d->matchGrid * 2u ? waves : (u32)d->matchGrid * 2u), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off,
                           (u32)levelWord, (const u32*)listA, (const u32*)ctr, mctr, tables, fs, mt);
        u32 const ldsRun = (u32)sizeof(ZEEntropy);
        u32 const gridA = (u32)(n < (size_t)d->encGridLvl[1] ? n : (size_t)d->encGridLvl[1]);
        hipLaunchKernelGGL(zj_encode_kernel, dim3(gridA), dim3(64), ldsRun, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                           (const u64*)d_dst_off, (u64*)d_result, (u32)levelWord, (const u32*)listA, (const u32*)ctr, ctr + 2, d->encScratch, (unsigned long long*)nullptr,
                           fs, maxSrcC, (const u32*)mt, 0u, (const u32*)nullptr, (u32*)nullptr, flags, (const ZECDictDev*)nullptr, (u32)(ldsRun), 0u, 0xFFFFFFFFu);
        if (bigLanes) {
            // list C in slices of ZJ_BIG_SLICE frames: [table set per lane slot][records per frame of the slice][meta]
            size_t bigWaves = (size_t)d->numCU;           // measured at level 5, 65 536 x 64 KiB: 128 / 256 / 512 / 1 024 waves -> 3.69 / 1.95 / 2.44 / 2.74 s (1 MiB of table per lane slot: beyond one wave per CU the rows' random requests take over)
            if (const char* ov = zj_tune("ZJNI_BIG_WAVES")) { long const v = atol(ov); if (v >= 1 && v <= 4096) bigWaves = (size_t)v; }
            size_t const slots = bigWaves * 64, tablesB = slots * ZE_MULTI_TABLE_BYTES;
            size_t const fsB = (size_t)ZJ_BIG_SLICE * ZE_FRAME_STRIDE(ZE_BLOCK_MAX), needB = tablesB + fsB + (size_t)ZJ_BIG_SLICE * 12 + 256;
            if (d->wideBufCap < needB) {
                if (!scratch_make_room(d, d->wideBufCap, needB)) return ZJNI_ERR(64);
                if (d->wideBuf) { if (hipStreamSynchronize(st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device); (void)hipFree(d->wideBuf); d->wideBuf = nullptr; d->wideBufCap = 0; }
                if (hipMalloc(&d->wideBuf, needB) != hipSuccess) return ZJNI_ERR(64);
                d->wideBufCap = needB;
            }
            u32* const tb = (u32*)d->wideBuf; u8* const fsb = d->wideBuf + tablesB; u32* const mtb = (u32*)(fsb + fsB);
            u32* const wctr = d->counters + 48;       // [2s] match work, [2s + 1] entropy work of slice s
            size_t const passes = (n + ZJ_BIG_SLICE - 1) / ZJ_BIG_SLICE;
            if (hipMemsetAsync(wctr, 0, 8 * passes, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
            for (size_t sN = 0; sN < passes; sN++) {
                u32 const base = (u32)(sN * ZJ_BIG_SLICE);
                hipLaunchKernelGGL(zj_enc_match_big_kernel, dim3((u32)(slots / 64)), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u32)levelWord,
                                   (const u32*)listC, (const u32*)(ctr + 4), wctr + 2 * sN, tb, fsb, mtb, base, (u32)ZJ_BIG_SLICE);
                u32 const gridB = (u32)(ZJ_BIG_SLICE < (size_t)d->encGridLvl[1] ? ZJ_BIG_SLICE : (size_t)d->encGridLvl[1]);
                hipLaunchKernelGGL(zj_encode_kernel, dim3(gridB), dim3(64), ldsRun, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                                   (const u64*)d_dst_off, (u64*)d_result, (u32)levelWord, (const u32*)listC, (const u32*)(ctr + 4), wctr + 2 * sN + 1, d->encScratch, (unsigned long long*)nullptr,
                                   fsb, (u32)ZE_BLOCK_MAX, (const u32*)mtb, 0u, (const u32*)nullptr, (u32*)nullptr, flags, (const ZECDictDev*)nullptr, (u32)(ldsRun), base, (u32)ZJ_BIG_SLICE);
            }
        }
        return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
    }
    // Large batches: match finding goes lane-per-frame (64 frames per wave) ahead of the wave-per-frame
    // entropy stage; small batches keep the fused wave-per-frame kernel (lower latency, tables in LDS).
    size_t splitMin = 4096;
    if (const char* ov = zj_env("ZJNI_SPLIT_MIN")) splitMin = (size_t)atoll(ov);
    if (tuned) splitMin = 1;                      // explicit table sizes exist only on the lane-per-frame path (tables in HBM)
    u8* fscratch = nullptr; u32* meta = nullptr; u32 const maxSrc = 65536u;
    // Small level-3 batches: list A goes through the wave-per-frame matcher (tables in LDS, 64 positions per step) and the
    // entropy kernel instead of the fused kernel, whose match finder is one lane walking the frame (3-10x the latency).
    bool const smallWave = n < splitMin && level == 3 && !tuned && zj_tune("ZJNI_NO_OVERLAP") == nullptr && zj_tune("ZJNI_NO_WAVE") == nullptr;
    if (n >= splitMin || smallWave) {
        u32 const tableStride = ze_lane_table_stride((u32)levelWord, false);   // fast: u16 entries; dfast: 4-byte tagged entries
        size_t const tablesBytes = smallWave ? 0 : n * (size_t)tableStride, fsBytes = n * (size_t)ZE_FRAME_STRIDE(maxSrc), metaBytes = n * 12, qBytes = n * 4;
        // Level 3, large batches: the run machine (zj_match_run.h), with need flags (zj_need.h: a flag byte per position, computed BESIDE the match
        // kernel) for the frames they pay on.  ZJNI_NEED: 2 (default) = the frames zn_worth() picks — search-dense frames over a small alphabet —
        // 1 = every frame, 0 = none.  64 KiB of flags per frame slot: left out under a scratch budget (zjni_set_scratch_limit) and when the device
        // cannot spare them; the machine then runs with every flag set.  ZJNI_LANE_MACHINE=0: the previous machine (ZLaneD), for A/B runs.
        u32 needMode = 2;
        if (const char* ov = zj_env("ZJNI_NEED")) needMode = (u32)atoi(ov);
        bool const overlap = zj_tune("ZJNI_NO_OVERLAP") == nullptr;
        bool const runMachine = level == 3 && !smallWave && zj_tune("ZJNI_HYBRID") == nullptr && !(zj_tune("ZJNI_LANE_MACHINE") && atoi(zj_tune("ZJNI_LANE_MACHINE")) == 0);
        u32 const hlN = ZE_LW_HL(levelWord) ? ZE_LW_HL(levelWord) : (tuned ? 16u : (u32)ZE_L3_HASHLOG), clN = ZE_LW_CL(levelWord) ? ZE_LW_CL(levelWord) : (tuned ? 15u : (u32)ZE_L3_CHAINLOG);
        bool needGate = (needMode == 1 || needMode == 2) && level == 3 && !smallWave && !g_scratch_limit && zj_tune("ZJNI_HYBRID") == nullptr
                        && overlap && hlN <= ZN_MAX_LOG_L && clN <= ZN_MAX_LOG_S;
        if (needGate && !d->needLdsSet) {                    // more than 64 KiB of dynamic LDS has to be asked for, once per device; refused: no flags
            if (hipFuncSetAttribute((const void*)zj_enc_need_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ZNLds)) == hipSuccess) d->needLdsSet = true;
            else { (void)hipGetLastError(); needGate = false; }
        }
        u32 const needSelective = needMode >= 2 ? 1u : 0u;
        size_t needFlagBytes = needGate ? n * (size_t)ZN_FLAG_STRIDE + 9 * n + 384 : 0;     // flags, gate bytes, ready words, pick list
        size_t need = tablesBytes + fsBytes + metaBytes + 2 * qBytes + n + 256 + needFlagBytes;
        if (d->splitBufCap < need) {
            if (!scratch_make_room(d, d->splitBufCap, need)) return ZJNI_ERR(64);
            if (d->splitBuf) { if (hipStreamSynchronize(st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device); (void)hipFree(d->splitBuf); d->splitBuf = nullptr; d->splitBufCap = 0; wasCleared = false; }
            if (hipMalloc(&d->splitBuf, need) != hipSuccess) {
                if (!needGate) return ZJNI_ERR(64);
                (void)hipGetLastError(); needGate = false; need -= needFlagBytes; needFlagBytes = 0;          // no room for the flags
                if (hipMalloc(&d->splitBuf, need) != hipSuccess) return ZJNI_ERR(64);
            }
            d->splitBufCap = need;
        }
        u8* const tables = d->splitBuf; fscratch = d->splitBuf + tablesBytes; meta = (u32*)(fscratch + fsBytes);
        u32* const doneList = (u32*)((u8*)meta + metaBytes); u32* const procFlag = doneList + n; u8* const score = (u8*)(procFlag + n);
        u8* const needFlags = needGate ? (u8*)(((uintptr_t)(score + n) + 63) & ~(uintptr_t)63) : nullptr;
        u32* const needReady = needGate ? (u32*)(needFlags + n * (size_t)ZN_FLAG_STRIDE) : nullptr;       // (ZN_FLAG_STRIDE is a multiple of 16)
        u32* const needPick = needGate ? needReady + n : nullptr;
        u8* const needGateMap = needGate ? (u8*)(needPick + n) : nullptr;
        u32* const needCtr = d->counters + 208;             // [0] flag kernel's work queue, [1] picked frames

        u32* const mctr = d->counters + 24;       // [0] match work, [1] completion-queue length, [2] work of the sweep pass
        // Experiment (ZJNI_HYBRID=1, off by default): at level 3 with the LDS-sized tables the two match
        // finders share a large batch — the list is partitioned by search density, the lane-per-frame kernel (tables in HBM,
        // bound by their random requests) takes the match-dense part, the wave-per-frame kernel (tables in LDS) the rest.
        // Measured on the metric configuration it does not pay: the wave kernel needs the LDS the entropy kernel runs in,
        // and under the lane kernel's traffic it falls to half its stand-alone rate.
        bool const hybrid = smallWave || (overlap && level == 3 && !tuned && zj_tune("ZJNI_HYBRID") != nullptr && zj_tune("ZJNI_NO_WAVE") == nullptr);
        bool const waveOnly = smallWave || (hybrid && zj_tune("ZJNI_WAVE_ONLY") != nullptr);
        unsigned long long* const work2 = (unsigned long long*)(d->counters + 192);
        const u32* listM = listA;
        if (hybrid && hipMemsetAsync(work2, 0, 32, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);   // split = 0: the whole list is the wave kernel's
#ifdef ZJ_TUNING_KERNELS
        if (hybrid && !waveOnly) {
            u32 const gs = (u32)(n < (size_t)d->numCU * 16 ? n : (size_t)d->numCU * 16);
            u32 threshold = 40;                       // text / JSON-like frames score 15-25, frames searched position by position 50+
            if (const char* ov = zj_tune("ZJNI_WAVE_SCORE")) threshold = (u32)atoi(ov);
            u32 share = 150;                          // permille of the batch the wave kernel takes at most
            if (const char* ov = zj_tune("ZJNI_WAVE_SHARE")) share = (u32)atoi(ov);
            hipLaunchKernelGGL(zj_enc_score_kernel, dim3(gs), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (const u32*)listA, (const u32*)ctr, score);
            hipLaunchKernelGGL(zj_enc_partition_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, (const u32*)listA, (const u32*)ctr, (const u8*)score, threshold, share, work2, listS);
            hipLaunchKernelGGL(zj_enc_partition_done_kernel, dim3(1), dim3(1), 0, st, (const u32*)ctr, share, work2);
            listM = listS;
        }
#endif
        int const preclearEnv = (zj_tune("ZJNI_PRECLEAR") && atoi(zj_tune("ZJNI_PRECLEAR")) == 0) ? 0 : 1;
        if (tablesBytes) {
            if (wasCleared && d->clearedPtr == tables && d->clearedBytes >= tablesBytes) {          // the previous call left them zero: `st` already waits for evCleared (above)
            } else if (hipMemsetAsync(tables, 0, tablesBytes, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
        }
        // after this call's match kernels (recorded below, at tev[1]): the same range zeroed again for the next call
        auto preclear = [&]() {
            if (!preclearEnv || !tablesBytes || g_scratch_limit) return;
            if (hipEventRecord(d->evMatchDone, st) != hipSuccess || hipStreamWaitEvent(d->clearStream, d->evMatchDone, 0) != hipSuccess) return;
            if (hipMemsetAsync(tables, 0, tablesBytes, d->clearStream) != hipSuccess || hipEventRecord(d->evCleared, d->clearStream) != hipSuccess) { (void)hipStreamSynchronize(d->clearStream); return; }
            d->clearedPtr = tables; d->clearedBytes = tablesBytes; d->clearedValid = true;
        };
        if (hipMemsetAsync(mctr, 0, 12, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
        u32 const waves = (u32)((n + 63) / 64);
        u32 gridM = waves < (u32)d->matchGrid ? waves : (u32)d->matchGrid;
        if (const char* ov = zj_tune("ZJNI_MATCH_GRID")) { u32 const v = (u32)atoi(ov); if (v >= 1 && v < gridM) gridM = v; }   // experiments: fewer resident waves, more frames per lane
        // with the sequences already found the entropy kernel only needs the entropy-stage LDS (more workgroups per CU)
        u32 const ldsRun = (u32)sizeof(ZEEntropy);
        u32 const gridA = (u32)(n < (size_t)d->encGridLvl[1] ? n : (size_t)d->encGridLvl[1]);
        unsigned long long* const eprof = d->prof ? d->prof + 16 : nullptr;
        if (needGate) {      // which frames get flags: decided ahead of the fork, a few microseconds per thousand frames
            if (hipMemsetAsync(needCtr, 0, 8, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
            u32 const gw = (u32)(n < (size_t)d->numCU * 8 ? n : (size_t)d->numCU * 8);
            hipLaunchKernelGGL(zj_enc_worth_kernel, dim3(gw), dim3(256), 0, st, (const u8*)d_src, (const u64*)d_src_off, (const u32*)listA, (const u32*)ctr,
                               needGateMap, needReady, needSelective, needPick, needCtr + 1);
            zj_dbg_sync("zj_enc_worth_kernel");
        }
        // an error return after work has been queued on the side streams must not leave it running over scratch the next call reuses
        auto bail = [&](size_t code) { (void)hipStreamSynchronize(d->sideStream); (void)hipStreamSynchronize(d->waveStream); (void)hipStreamSynchronize(st); return code; };
        if (overlap) {
            // The entropy kernel runs on a side stream BESIDE the match kernel and consumes its completion queue:
            // frames that parse quickly are entropy-coded while the slow ones still occupy their lanes (the match
            // kernel leaves most of every CU idle in its tail).  A sweep pass afterwards takes whatever the side
            // kernel did not get to (its waits are bounded), so completion never depends on the two kernels
            // actually being co-scheduled.
            if (hipMemsetAsync(doneList, 0xFF, qBytes, st) != hipSuccess || hipMemsetAsync(procFlag, 0, qBytes, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
            if (hipEventRecord(d->evFork, st) != hipSuccess || hipStreamWaitEvent(d->sideStream, d->evFork, 0) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
            if ((hybrid || needGate) && hipStreamWaitEvent(d->waveStream, d->evFork, 0) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
            if (needGate) {      // the flag kernel on its own stream, enqueued before the entropy kernel: its workgroups take 140 KiB of LDS each (sizeof(ZNLds)), which a CU full of waiting entropy workgroups does not have
                u32 const gn = (u32)(n < (size_t)d->numCU ? n : (size_t)d->numCU);
                hipLaunchKernelGGL(zj_enc_need_kernel, dim3(gn), dim3(zj_need_threads()), sizeof(ZNLds), zj_tune("ZJNI_NEED_INLINE") ? st : d->waveStream, (const u8*)d_src, (const u64*)d_src_off, (u32)levelWord,
                                   (const u32*)listA, (const u32*)needPick, (const u32*)(needCtr + 1), needFlags, needReady, needCtr, (u32)ZN_FLAG_STRIDE);
                if (hipEventRecord(d->evJoinWave, d->waveStream) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
                zj_dbg_sync("zj_enc_need_kernel");
            }
            (void)hipEventRecord(d->tev[0], st);
            if (hybrid) {
                u32 const gw = (u32)(n < (size_t)d->waveGrid ? n : (size_t)d->waveGrid);
                hipLaunchKernelGGL(zj_enc_match_wave_kernel, dim3(gw), dim3(64), sizeof(ZWLds), d->waveStream, (const u8*)d_src, (const u64*)d_src_off, (u32)levelWord,
                                   listM, (const u32*)ctr, work2, fscratch, maxSrc, meta, doneList, mctr + 1);
                if (hipEventRecord(d->evJoinWave, d->waveStream) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
            }
            u32 lanePeriod = 0;                                    // rotation period of the double-fast lane machines (0 = the machine's own)
            if (const char* ov = zj_tune("ZJNI_LANE_PERIOD")) { lanePeriod = (u32)atoi(ov) & 0xFu; if (lanePeriod && lanePeriod < 3u) lanePeriod = 3u; }     // (three non-search states take turns: a shorter rotation would never run one of them)
            if (runMachine) {
                void (*kern)(const u8*, const u64*, u32, const u32*, const u32*, u32*, u8*, u32, u8*, u32, u32*, u32*, u32*, u32, u32, const u8*, const u8*, const u32*) = zj_enc_match_run_kernel;
#ifdef ZJ_TUNING_KERNELS
                if (const char* ov = zj_tune("ZJNI_RUN_JMAX")) { int const j = atoi(ov); kern = j == 3 ? zj_enc_match_run3_kernel : j == 4 ? zj_enc_match_run4_kernel : j == 6 ? zj_enc_match_run6_kernel : j == 7 ? zj_enc_match_run7_kernel : zj_enc_match_run_kernel; }
#endif
                hipLaunchKernelGGL(kern, dim3(gridM), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u32)levelWord | (lanePeriod << 24),
                                   listM, (const u32*)ctr, mctr, tables, tableStride, fscratch, maxSrc, meta, doneList, mctr + 1, 0u, 0xFFFFFFFFu, (const u8*)needFlags, (const u8*)needGateMap, (const u32*)needReady);
                d->lastRoute = needGate ? ZJ_ROUTE_RUN_FLAGS : ZJ_ROUTE_RUN;
            }
#ifdef ZJ_TUNING_KERNELS
            else if (needGate) {
                hipLaunchKernelGGL(zj_enc_match_gated_kernel, dim3(gridM), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u32)levelWord | (lanePeriod << 24),
                                   listM, (const u32*)ctr, mctr, tables, tableStride, fscratch, maxSrc, meta, doneList, mctr + 1, 0u, 0xFFFFFFFFu, (const u8*)needFlags, (const u8*)needGateMap, (const u32*)needReady);
                d->lastRoute = ZJ_ROUTE_LANE_GATED;
            }
#endif
            else if (!waveOnly) {
            hipLaunchKernelGGL(zj_enc_match_kernel, dim3(gridM), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u32)levelWord | (lanePeriod << 24),
                               listM, (const u32*)ctr, mctr, tables, tableStride, fscratch, maxSrc, meta, doneList, mctr + 1, 0u, 0xFFFFFFFFu,
                               hybrid ? work2 : (unsigned long long*)nullptr);
            d->lastRoute = hybrid ? ZJ_ROUTE_HYBRID : ZJ_ROUTE_LANE;
            } else d->lastRoute = ZJ_ROUTE_WAVE;
            if (needGate && hipStreamWaitEvent(st, d->evJoinWave, 0) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
            zj_dbg_sync("match kernel");
            if (hybrid && hipStreamWaitEvent(st, d->evJoinWave, 0) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
            (void)hipEventRecord(d->tev[1], st); d->tevCompress = true;
            preclear();
            // the entropy kernel's persistent workgroups fill the LDS of every CU; the flag kernel's need 140 KiB each: the entropy kernel starts when the flags are done
            // (measured without this: whichever kernel the dispatcher places first wins, and every second call the picked frames' lanes wait out their 50 ms)
            if (needGate && !zj_tune("ZJNI_NEED_INLINE") && hipStreamWaitEvent(d->sideStream, d->evJoinWave, 0) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
            u32 gridE = gridA;                                  // (ZJNI_ENT_GRID: fewer entropy workgroups beside the match kernel — with <= 1 024 every match wave is resident from the start, profiles/r05/b_; the call's time does not move)
            if (const char* ov = zj_tune("ZJNI_ENT_GRID")) { u32 const v = (u32)atoi(ov); if (v >= 1 && v < gridE) gridE = v; }
            hipLaunchKernelGGL(zj_encode_kernel, dim3(gridE), dim3(64), ldsRun, d->sideStream, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                               (const u64*)d_dst_off, (u64*)d_result, (u32)levelWord, listM, (const u32*)ctr, ctr + 2, d->encScratch, eprof,
                               fscratch, maxSrc, (const u32*)meta, 1u, (const u32*)doneList, procFlag, flags, (const ZECDictDev*)nullptr, (u32)(ldsRun), 0u, 0xFFFFFFFFu);
            if (hipEventRecord(d->evJoin, d->sideStream) != hipSuccess || hipStreamWaitEvent(st, d->evJoin, 0) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
            hipLaunchKernelGGL(zj_encode_kernel, dim3(gridA), dim3(64), ldsRun, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                               (const u64*)d_dst_off, (u64*)d_result, (u32)levelWord, listM, (const u32*)ctr, mctr + 2, d->encScratch, eprof,
                               fscratch, maxSrc, (const u32*)meta, 2u, (const u32*)doneList, procFlag, flags, (const ZECDictDev*)nullptr, (u32)(ldsRun), 0u, 0xFFFFFFFFu);
        } else {
            (void)hipEventRecord(d->tev[0], st);
            d->lastRoute = runMachine ? ZJ_ROUTE_RUN : ZJ_ROUTE_LANE;
            if (runMachine)
                hipLaunchKernelGGL(zj_enc_match_run_kernel, dim3(gridM), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u32)levelWord,
                                   (const u32*)listA, (const u32*)ctr, mctr, tables, tableStride, fscratch, maxSrc, meta, (u32*)nullptr, (u32*)nullptr, 0u, 0xFFFFFFFFu, (const u8*)nullptr, (const u8*)nullptr, (const u32*)nullptr);
            else
            hipLaunchKernelGGL(zj_enc_match_kernel, dim3(gridM), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u32)levelWord,
                               (const u32*)listA, (const u32*)ctr, mctr, tables, tableStride, fscratch, maxSrc, meta, (u32*)nullptr, (u32*)nullptr, 0u, 0xFFFFFFFFu, (unsigned long long*)nullptr);
            (void)hipEventRecord(d->tev[1], st); d->tevCompress = true;
            preclear();
            hipLaunchKernelGGL(zj_encode_kernel, dim3(gridA), dim3(64), ldsRun, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                               (const u64*)d_dst_off, (u64*)d_result, (u32)levelWord, (const u32*)listA, (const u32*)ctr, ctr + 2, d->encScratch, eprof,
                               fscratch, maxSrc, (const u32*)meta, 0u, (const u32*)nullptr, (u32*)nullptr, flags, (const ZECDictDev*)nullptr, (u32)(ldsRun), 0u, 0xFFFFFFFFu);
        }
    } else {
        // small batches: the fused wave-per-frame kernel (match finding on lane 0 with the tables in LDS)
        d->lastRoute = ZJ_ROUTE_FUSED;
        u32 const gridA = (u32)(n < (size_t)d->encGridLvl[level] ? n : (size_t)d->encGridLvl[level]);
        hipLaunchKernelGGL(zj_encode_kernel, dim3(gridA), dim3(64), ldsA, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                           (const u64*)d_dst_off, (u64*)d_result, (u32)levelWord, (const u32*)listA, (const u32*)ctr, ctr + 2, d->encScratch, d->prof ? d->prof + 16 : nullptr,
                           (u8*)nullptr, maxSrc, (const u32*)nullptr, 0u, (const u32*)nullptr, (u32*)nullptr, flags, (const ZECDictDev*)nullptr, (u32)(ldsA), 0u, 0xFFFFFFFFu);
    }
    if (n >= splitMin) {
        // List B (frames > 64 KiB, fast-strategy frames with larger tables) through the same two stages, in slices that
        // share one scratch area sized for 4-byte positions and 128 KiB frames; a slice past the end of the list is empty.
        auto bailB = [&](size_t code) { (void)hipStreamSynchronize(d->sideStream); (void)hipStreamSynchronize(d->waveStream); (void)hipStreamSynchronize(st); return code; };
        size_t sliceB = 65536;                       // as many lanes as the common path runs: 32 768 leaves half the wave slots empty (13.8 vs 18.5 GiB/s on 128 KiB frames)
        if (const char* ov = zj_env("ZJNI_WIDE_SLICE")) sliceB = (size_t)atoll(ov);
        sliceB = scratch_slice((size_t)ze_lane_table_stride((u32)levelWord, true) + ZE_FRAME_STRIDE(ZE_WIDE_MAX_SRC) + 12, sliceB, 2);   // half the budget: the common-case buffer of this call has the other half
        if (sliceB > n) sliceB = n;
        if (sliceB < 64) sliceB = 64;
        u32 const strideB = ze_lane_table_stride((u32)levelWord, true);
        size_t const tablesB = sliceB * (size_t)strideB, fsB = sliceB * (size_t)ZE_FRAME_STRIDE(ZE_WIDE_MAX_SRC), metaB = sliceB * 12, qB = sliceB * 4;
        // The entropy kernel BESIDE the match kernel, as on the common path: it takes frames off the completion queue while the slow frames still occupy
        // their lanes; a sweep pass afterwards takes what it did not get to.  ZJNI_NO_OVERLAP: one after the other (A/B runs).
        bool const besideB = zj_tune("ZJNI_NO_OVERLAP") == nullptr;
        // Level 3: need flags for the frames zj_enc_worth_kernel picks (zn_flags_frame_wide: 64 KiB + 1 .. 128 KiB) and the run machine for the slice — what the common
        // path does for frames to 64 KiB.  Only when one slice holds the whole list (a call's frames normally do), ZJNI_NEED / ZJNI_LANE_MACHINE as there;
        // ZJNI_NEED_WIDE=0 keeps the wide launch on ZLaneD without flags (A/B runs).
        u32 needModeB = 2; if (const char* ov = zj_env("ZJNI_NEED")) needModeB = (u32)atoi(ov);
        u32 const hlB = ZE_LW_HL(levelWord) ? ZE_LW_HL(levelWord) : (tuned ? 16u : (u32)ZE_L3_HASHLOG), clB = ZE_LW_CL(levelWord) ? ZE_LW_CL(levelWord) : (tuned ? 15u : (u32)ZE_L3_CHAINLOG);
        bool needW = level == 3 && (needModeB == 1 || needModeB == 2) && besideB && !g_scratch_limit && sliceB >= n && hlB <= ZN_MAX_LOG_L && clB <= ZN_MAX_LOG_S
                     && !(zj_tune("ZJNI_LANE_MACHINE") && atoi(zj_tune("ZJNI_LANE_MACHINE")) == 0) && !(zj_tune("ZJNI_NEED_WIDE") && atoi(zj_tune("ZJNI_NEED_WIDE")) == 0);
        if (needW && !d->needLdsSet) {
            if (hipFuncSetAttribute((const void*)zj_enc_need_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ZNLds)) == hipSuccess) d->needLdsSet = true;
            else { (void)hipGetLastError(); needW = false; }
        }
        size_t flagB = needW ? sliceB * (size_t)ZN_FLAG_STRIDE_WIDE + 9 * sliceB + 384 : 0;      // flags, ready words, pick list, gate bytes
        size_t needB = tablesB + fsB + metaB + 2 * qB + 256 + flagB;
        if (d->wideBufCap < needB && listsAsked) {                 // no wide slice yet: is there a list B at all?
            if (hipEventSynchronize(d->evLists) != hipSuccess) return bailB(ZJNI_ERR(ZJNI_ERROR_no_device));
            if (d->encStat[1] == 0u) return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
        }
        if (d->wideBufCap < needB) {
            if (!scratch_make_room(d, d->wideBufCap, needB)) return ZJNI_ERR(64);
            if (d->wideBuf) { if (hipStreamSynchronize(st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device); (void)hipFree(d->wideBuf); d->wideBuf = nullptr; d->wideBufCap = 0; }
            if (hipMalloc(&d->wideBuf, needB) != hipSuccess) {
                if (!needW) return ZJNI_ERR(64);
                (void)hipGetLastError(); needW = false; needB -= flagB; flagB = 0;                    // no room for the flags
                if (hipMalloc(&d->wideBuf, needB) != hipSuccess) return ZJNI_ERR(64);
            }
            d->wideBufCap = needB;
        }
        u8* const tb = d->wideBuf; u8* const fs = d->wideBuf + tablesB; u32* const mt = (u32*)(fs + fsB);
        u32* const doneB = (u32*)((u8*)mt + metaB); u32* const procB = doneB + sliceB;      // completion queue of a slice's match kernel, frames the side pass encoded
        u8* const flagsW = needW ? (u8*)(((uintptr_t)(procB + sliceB) + 63) & ~(uintptr_t)63) : nullptr;
        u32* const readyW = needW ? (u32*)(flagsW + sliceB * (size_t)ZN_FLAG_STRIDE_WIDE) : nullptr;
        u32* const pickW = needW ? readyW + sliceB : nullptr;
        u8* const gateW = needW ? (u8*)(pickW + sliceB) : nullptr;
        u32* const needCtrW = d->counters + 212;       // [0] flag kernel's work queue, [1] picked frames
        u32* const wctr = d->counters + 48;           // [0] match work, [1] entropy work (queue order), [2] queue length, [3] work of the sweep pass
        u32 const wavesB = (u32)((sliceB + 63) / 64);
        u32 const gridMB = wavesB < (u32)d->matchGrid ? wavesB : (u32)d->matchGrid;
        u32 const gridEB = (u32)(sliceB < (size_t)d->encGridLvl[1] ? sliceB : (size_t)d->encGridLvl[1]);
        for (size_t base = 0; base < n; base += sliceB) {
            if (hipMemsetAsync(wctr, 0, 16, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
            hipLaunchKernelGGL(zj_zero_slots_kernel, dim3((u32)d->numCU * 8), dim3(256), 0, st, tb, strideB, (const u32*)(ctr + 1), (u32)base, (u32)sliceB);
            bool forked = false, flagged = false;
            if (besideB) {
                if (hipMemsetAsync(doneB, 0xFF, qB, st) != hipSuccess || hipMemsetAsync(procB, 0, qB, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
                if (needW) {                               // which frames get flags: decided ahead of the fork
                    if (hipMemsetAsync(needCtrW, 0, 8, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
                    u32 const gw = (u32)(n < (size_t)d->numCU * 8 ? n : (size_t)d->numCU * 8);
                    hipLaunchKernelGGL(zj_enc_worth_kernel, dim3(gw), dim3(256), 0, st, (const u8*)d_src, (const u64*)d_src_off, (const u32*)listB, (const u32*)(ctr + 1),
                                       gateW, readyW, needModeB >= 2 ? 1u : 0u, pickW, needCtrW + 1);
                }
                forked = hipEventRecord(d->evFork, st) == hipSuccess && hipStreamWaitEvent(d->sideStream, d->evFork, 0) == hipSuccess;
                if (forked && needW && hipStreamWaitEvent(d->waveStream, d->evFork, 0) == hipSuccess) {    // the flag kernel on its own stream, enqueued before the entropy kernel (its workgroups need 140 KiB of LDS each)
                    u32 const gn = (u32)(n < (size_t)d->numCU ? n : (size_t)d->numCU);
                    hipLaunchKernelGGL(zj_enc_need_kernel, dim3(gn), dim3(zj_need_threads()), sizeof(ZNLds), d->waveStream, (const u8*)d_src, (const u64*)d_src_off, (u32)levelWord,
                                       (const u32*)listB, (const u32*)pickW, (const u32*)(needCtrW + 1), flagsW, readyW, needCtrW, (u32)ZN_FLAG_STRIDE_WIDE);
                    flagged = hipEventRecord(d->evJoinWave, d->waveStream) == hipSuccess;
                    if (!flagged) { (void)hipStreamSynchronize(d->waveStream); (void)hipStreamSynchronize(d->sideStream); (void)hipStreamSynchronize(st); return ZJNI_ERR(ZJNI_ERROR_no_device); }
                }
            }
            (void)hipEventRecord(d->tev[8], st);
            hipLaunchKernelGGL(zj_enc_match_wide_kernel, dim3(gridMB), dim3(64), 0, st, (const u8*)d_src, (const u64*)d_src_off, (u32)levelWord,
                               (const u32*)listB, (const u32*)(ctr + 1), wctr, tb, strideB, fs, (u32)ZE_WIDE_MAX_SRC, mt, (u32)base, (u32)sliceB, forked ? doneB : (u32*)nullptr, forked ? wctr + 2 : (u32*)nullptr,
                               (const u8*)(flagged ? flagsW : nullptr), (const u8*)(flagged ? gateW : nullptr), (const u32*)(flagged ? readyW : nullptr));
            if (flagged && hipStreamWaitEvent(st, d->evJoinWave, 0) != hipSuccess) { (void)hipStreamSynchronize(d->waveStream); (void)hipStreamSynchronize(d->sideStream); (void)hipStreamSynchronize(st); return ZJNI_ERR(ZJNI_ERROR_no_device); }
            (void)hipEventRecord(d->tev[9], st); d->tevWide = true;
            if (flagged && hipStreamWaitEvent(d->sideStream, d->evJoinWave, 0) != hipSuccess) { (void)hipStreamSynchronize(d->waveStream); (void)hipStreamSynchronize(d->sideStream); (void)hipStreamSynchronize(st); return ZJNI_ERR(ZJNI_ERROR_no_device); }     // the entropy kernel's workgroups fill every CU's LDS: it starts when the flags are done
            if (forked) {
                hipLaunchKernelGGL(zj_encode_kernel, dim3(gridEB), dim3(64), (u32)sizeof(ZEEntropy), d->sideStream, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                                   (const u64*)d_dst_off, (u64*)d_result, (u32)levelWord, (const u32*)listB, (const u32*)(ctr + 1), wctr + 1, d->encScratch, d->prof ? d->prof + 16 : nullptr,
                                   fs, (u32)ZE_WIDE_MAX_SRC, (const u32*)mt, 1u, (const u32*)doneB, procB, flags, (const ZECDictDev*)nullptr, (u32)sizeof(ZEEntropy), (u32)base, (u32)sliceB);
                if (hipEventRecord(d->evJoin, d->sideStream) != hipSuccess || hipStreamWaitEvent(st, d->evJoin, 0) != hipSuccess) { (void)hipStreamSynchronize(d->sideStream); (void)hipStreamSynchronize(st); return ZJNI_ERR(ZJNI_ERROR_no_device); }
            }
            hipLaunchKernelGGL(zj_encode_kernel, dim3(gridEB), dim3(64), (u32)sizeof(ZEEntropy), st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                               (const u64*)d_dst_off, (u64*)d_result, (u32)levelWord, (const u32*)listB, (const u32*)(ctr + 1), forked ? wctr + 3 : wctr + 1, d->encScratch, d->prof ? d->prof + 16 : nullptr,
                               fs, (u32)ZE_WIDE_MAX_SRC, (const u32*)mt, forked ? 2u : 0u, (const u32*)(forked ? doneB : nullptr), forked ? procB : (u32*)nullptr, flags, (const ZECDictDev*)nullptr, (u32)sizeof(ZEEntropy), (u32)base, (u32)sliceB);
        }
        return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
    }
    u32 const gridB = (u32)(n < (size_t)d->encGridBig ? n : (size_t)d->encGridBig);
    hipLaunchKernelGGL(zj_encode_kernel, dim3(gridB), dim3(64), ZJ_ENC_LDS_BIG, st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst,
                       (const u64*)d_dst_off, (u64*)d_result, (u32)levelWord, (const u32*)listB, (const u32*)(ctr + 1), ctr + 3, d->encScratch, d->prof ? d->prof + 16 : nullptr,
                       (u8*)nullptr, maxSrc, (const u32*)nullptr, 0u, (const u32*)nullptr, (u32*)nullptr, flags, (const ZECDictDev*)nullptr, (u32)(ZJ_ENC_LDS_BIG), 0u, 0xFFFFFFFFu);
    return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
}
// Level 3 with nothing set runs the reference's own tables (hashLog 16 / chainLog 15, N/compress/clevels.h:29-31 after ZSTD_adjustCParams): every frame is
// Zstd.compress(x, 3)'s, byte for byte, whatever the batch size — measured at the full batch the 4x larger tables cost nothing (153.6 vs 153.7 ms, the
// kernel is bound by request count, not footprint).  ZstdCompressCtx.setHashLog(14).setChainLog(13) selects the LDS-sized tables: the wave-per-frame
// matcher and the fused kernel of small batches (lower latency per call, 0.2 % smaller frames on the bench set, not the reference's default bytes).
// ZJNI_L3_TABLES=lds restores that as the default (A/B runs against round 2).
static int zj_level3_word(int lw) {
    u32 const w = (u32)lw;
    if (ZE_LW_LEVEL(w) != 3u || (ZE_LW_HL(w) | ZE_LW_CL(w))) return lw;
    static int const lds = (zj_tune("ZJNI_L3_TABLES") && !strcmp(zj_tune("ZJNI_L3_TABLES"), "lds")) ? 1 : 0;
    return lds ? lw : (int)(ZE_LW(3u, 16u, 15u) | ZE_LW_IMPLICIT | (w & ~0xFFFFFFu));
}
static size_t compress_chunked(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off,
                               uint64_t* d_result, size_t n, int level, u32 flags, void* stream) {
    level = zj_level3_word(level);
    BatchOrder order(cur_state(), stream);
    size_t const perFrame = ZE_LW_LEVEL((u32)level) > 3u ? (size_t)ZE_CHAIN_TABLE_BYTES + ZE_FRAME_STRIDE(ZE_CHAIN_MAX_SRC) + 21
                                                          : (size_t)ze_lane_table_stride((u32)level, false) + ZE_FRAME_STRIDE(65536u) + 21;
    size_t const chunk = scratch_slice(perFrame, ZJ_CHUNK_FRAMES, 2);
    for (size_t at = 0; at < n || at == 0; at += chunk) {
        size_t const m = n - at < chunk ? n - at : chunk;
        size_t const r = compress_batch_device_impl(d_src, d_src_off + at, d_dst, d_dst_off + at, d_result + at, m, level, flags, stream);
        if (r != 0 || n == 0) return r;
    }
    return 0;
}
size_t zjni_compress_batch_device(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off,
                                  uint64_t* d_result, size_t n, int level, void* stream) {
    if (level == 0) level = 3;                    // ZSTD_CLEVEL_DEFAULT, as ZSTD_c_compressionLevel = 0 means
    if (level < 1 || level > ZJ_LEVEL_MAX) return ZJNI_ERR(42);
    return compress_chunked(d_src, d_src_off, d_dst, d_dst_off, d_result, n, level, 0u, stream);
}
size_t zjni_compress_batch_device2(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off,
                                   uint64_t* d_result, size_t n, int level, int checksum, void* stream) {
    if (level == 0) level = 3;                    // ZSTD_CLEVEL_DEFAULT, as ZSTD_c_compressionLevel = 0 means
    if (level < 1 || level > ZJ_LEVEL_MAX) return ZJNI_ERR(42);
    return compress_chunked(d_src, d_src_off, d_dst, d_dst_off, d_result, n, level, checksum ? ZE_FLAG_CHECKSUM : 0u, stream);
}
// Stream frames (include/zjni_amd.h): n streams, each buffered whole, through zj_encode_stream_kernel.
size_t zjni_compress_stream_batch_device(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off, uint64_t* d_result, size_t n, int level, int checksum,
                                         const uint32_t* d_flush_at, const uint64_t* d_flush_off, const uint32_t* d_mode, void* stream) {
    if (level == 0) level = 3;
    if (level < 1 || level > 3) return ZJNI_ERR(42);                 // the levels whose multi-block frames the kernels make (above: the bundled library's)
    DevState* d = cur_state();
    if (!d) return ZJNI_ERR(ZJNI_ERROR_no_device);
    if (n == 0) return 0;
    if (n > 0xFFFFFFFFull) return ZJNI_ERR(72);
    BatchOrder order(d, stream);
    hipStream_t st = (hipStream_t)stream;
    if (!ensure_multi_tables(d)) return ZJNI_ERR(64);
    u32* const ctr = d->counters + 224;
    if (hipMemsetAsync(ctr, 0, 4, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
    u32 const gc = (u32)(n < (size_t)d->multiGrid ? n : (size_t)d->multiGrid);
    u32 flags = checksum ? ZE_FLAG_CHECKSUM : 0u;
    if (const char* ov = zj_tune("ZJNI_MULTI_WAVE")) { int const v = atoi(ov); flags |= v == 0 ? ZE_FLAG_MULTI_SERIAL : (v == 2 ? ZE_FLAG_MULTI_NOCARRY : 0u); }
    {   const char* const ov = zj_tune("ZJNI_MULTI_WAVE_FAST"); if (!(ov && atoi(ov) == 1)) flags |= ZE_FLAG_MULTI_FAST_SERIAL; }      // levels 1-2: the one-lane parse (zj_encode_multi_kernel's default)
    hipLaunchKernelGGL(zj_encode_stream_kernel, dim3(gc), dim3(64), (u32)sizeof(ZEEntropy), st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst, (const u64*)d_dst_off,
                       (u64*)d_result, (u32)level, (u32)n, ctr, d->encScratch, d->multiTables, flags, (u32)sizeof(ZEEntropy), (const u32*)d_flush_at, (const u64*)d_flush_off, (const u32*)d_mode);
    return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
}
// One stream from host pointers: what was written so far, where the caller flushed, closing or not.  Returns the bytes the frame has so far (final = 0: up to the
// last flush), or an error; 201 when the total exceeds the level's unknown-size window (the bundled library's stream takes over).
size_t zjni_compress_stream(void* dst, size_t dstCap, const void* src, size_t srcSize, int level, int checksum, const uint32_t* flushAt, size_t nFlush, int final_, int knownEmpty) {
    if (level == 0) level = 3;
    if (level < 1 || level > 3) return ZJNI_ERR(42);
    if (srcSize > ((size_t)1 << ze_stream_window_log((u32)level)) || srcSize > ZE_MULTI_MAX) return ZJNI_ERR(201);
    if ((srcSize && !src) || (dstCap && !dst) || (nFlush && !flushAt) || nFlush > (1u << 20)) return ZJNI_ERR(72);
    DevState* d = cur_state();
    if (!d) return ZJNI_ERR(ZJNI_ERROR_no_device);
    size_t const cap = dstCap > ((size_t)1 << 31) ? ((size_t)1 << 31) : dstCap;
    // staging layout: [srcOff 2][dstOff 2][result 1][flushOff 2][mode (4 bytes, padded to 8)][flush positions][src][dst]
    size_t const oSrcOff = 0, oDstOff = 16, oRes = 32, oFOff = 40, oMode = 56, oFlush = 64, oSrc = (oFlush + 4 * nFlush + 15) & ~(size_t)15, oDst = (oSrc + srcSize + 15) & ~(size_t)15;
    size_t const total = oDst + cap + 16;
    SlotLock slotLock(d); StageSlot* const sl = slotLock.s;
    if (!ensure_staging(sl, total)) return ZJNI_ERR(ZJNI_ERROR_unsupported);
    hipStream_t const hst = sl->hostK;
    struct Drain { hipStream_t st; ~Drain() { (void)hipStreamSynchronize(st); } } drainOnExit{hst};
    u64* const h = (u64*)sl->hPinned;
    h[0] = 0; h[1] = srcSize; h[2] = 0; h[3] = cap; h[4] = 0; h[5] = 0; h[6] = nFlush; ((u32*)(sl->hPinned + oMode))[0] = (final_ ? 1u : 0u) | (knownEmpty ? 2u : 0u); ((u32*)(sl->hPinned + oMode))[1] = 0;
    if (nFlush) memcpy(sl->hPinned + oFlush, flushAt, 4 * nFlush);
    if (srcSize) memcpy(sl->hPinned + oSrc, src, srcSize);
    if (hipMemcpyAsync(sl->dStage, sl->hPinned, oSrc + srcSize, hipMemcpyHostToDevice, hst) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
    size_t const r = zjni_compress_stream_batch_device(sl->dStage + oSrc, (const u64*)(sl->dStage + oSrcOff), sl->dStage + oDst, (const u64*)(sl->dStage + oDstOff), (u64*)(sl->dStage + oRes), 1, level, checksum,
                                                       (const u32*)(sl->dStage + oFlush), (const u64*)(sl->dStage + oFOff), (const u32*)(sl->dStage + oMode), hst);
    if (zjni_isError(r)) return r;
    if (hipMemcpyAsync(sl->hPinned + oRes, sl->dStage + oRes, 8, hipMemcpyDeviceToHost, hst) != hipSuccess || hipStreamSynchronize(hst) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
    size_t const out = (size_t)h[4];
    if (zjni_isError(out) || out == 0) return out;
    if (out > cap) return ZJNI_ERR(70);
    if (hipMemcpyAsync(sl->hPinned + oDst, sl->dStage + oDst, out, hipMemcpyDeviceToHost, hst) != hipSuccess || hipStreamSynchronize(hst) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
    memcpy(dst, sl->hPinned + oDst, out);
    return out;
}
// ZstdCompressCtx.setHashLog / setChainLog (ZSTD_c_hashLog / ZSTD_c_chainLog; 0 = the library's choice) on top of level + checksum.
// Honoured for level 3 (double-fast), hashLog 6..17, chainLog 6..16: with 16 / 15 the frames are the reference's plain level 3.
static size_t level_word(int level, int hashLog, int chainLog, int* out) {
    *out = level;
    if (!hashLog && !chainLog) return 0;
    if (level != 3) return ZJNI_ERR(40);
    if ((hashLog && (hashLog < 6 || hashLog > (int)ZE_HASHLOG_CAP)) || (chainLog && (chainLog < 6 || chainLog > (int)ZE_CHAINLOG_CAP))) return ZJNI_ERR(42);
    *out = (int)ZE_LW(level, hashLog, chainLog);
    return 0;
}
size_t zjni_compress_batch_device_advanced(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off,
                                           uint64_t* d_result, size_t n, int level, int checksum, int hashLog, int chainLog, void* stream) {
    if (level == 0) level = 3;
    int lw; size_t const e = level_word(level, hashLog, chainLog, &lw);
    if (e) return e;
    return compress_chunked(d_src, d_src_off, d_dst, d_dst_off, d_result, n, lw, zj_frame_flags(checksum), stream);
}

// ZSTD_createCDict (N/compress/zstd_compress.c:5710-5719; ZstdDictCompress.init, N/jni_fast_zstd.c:18-52): the raw dictionary
// goes to HBM once, behind its (zeroed) tagged tables, and one workgroup digests it there.  NULL when the device, the level
// or the dictionary is bad.
zjni_cdict* zjni_createCDict(const void* dict, size_t dictSize, int level) {
    DevState* d = cur_state();
    if (level == 0) level = 3;                        // ZSTD_createCDict: 0 = ZSTD_CLEVEL_DEFAULT
    if (!d || !dict || dictSize < 8 || dictSize > 0x3FFFFFFFull || level < 1 || level > 3) return nullptr;
    ZEParams const cp = ze_cdict_params((u32)level, (u32)dictSize);
    size_t const head = (sizeof(ZECDictDev) + 15) & ~(size_t)15, tablesBytes = (size_t)ze_cdict_table_entries(cp) * 4u;
    zjni_cdict* cd = new zjni_cdict{t_dev, nullptr, 0, level, cp.strategy};
    if (hipMalloc(&cd->buf, head + tablesBytes + dictSize + 16) != hipSuccess) { delete cd; return nullptr; }
    ZECDictDev hd; memset(&hd, 0, sizeof(hd));
    hd.tablesOff = (u32)head; hd.rawOff = (u32)(head + tablesBytes);
    bool ok = hipMemset(cd->buf, 0, head + tablesBytes) == hipSuccess
           && hipMemcpy(cd->buf, &hd, sizeof(hd), hipMemcpyHostToDevice) == hipSuccess
           && hipMemcpy(cd->buf + hd.rawOff, dict, dictSize, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(zj_cdict_digest_kernel, dim3(1), dim3(64), 0, 0, (u32)dictSize, (u32)level, (ZECDictDev*)cd->buf);
        ok = hipMemcpy(&hd, cd->buf, 64, hipMemcpyDeviceToHost) == hipSuccess && hd.status == 0;
    }
    if (!ok) { (void)hipFree(cd->buf); delete cd; return nullptr; }
    cd->dictID = hd.dictID;
    return cd;
}
size_t zjni_freeCDict(zjni_cdict* cd) {
    if (!cd) return 0;
    (void)hipSetDevice(cd->ordinal);
    (void)hipFree(cd->buf);
    delete cd;
    return 0;
}
unsigned zjni_getDictID_fromCDict(const zjni_cdict* cd) { return cd ? cd->dictID : 0u; }

// ZstdCompressCtx.loadDict(ZstdDictCompress) + compress, batched: lane-per-frame attach-mode search, then the
// wave-per-frame entropy stage starting from the dictionary's tables.
// Slices of 2 x ZJ_CHUNK_FRAMES: memset/classify/clear/match of slice s on the caller's stream, its entropy kernel on the side
// stream — beside slice s+1's match kernel (which leaves most of every CU idle: one wave per SIMD, waiting on memory).
size_t zjni_compress_batch_device_usingCDict(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off,
                                             uint64_t* d_result, size_t n, const zjni_cdict* cdict, int checksum, void* stream) {
    if (!cdict) return ZJNI_ERR(32);
    if (cdict->ordinal != t_dev && t_dev >= 0) return ZJNI_ERR(32);                       // digested on another device
    DevState* d = cur_state();
    if (!d) return ZJNI_ERR(ZJNI_ERROR_no_device);
    if (n == 0) return 0;
    if (n > 0xFFFFFFFFull) return ZJNI_ERR(72);
    BatchOrder order(d, stream);
    hipStream_t st = (hipStream_t)stream;
    u32 const flags = zj_frame_flags(checksum);
    size_t chunk = 2 * ZJ_CHUNK_FRAMES;            // 2 048 match waves = every SIMD's second wave slot as well: more table requests in flight (measured: +15 % over 65 536)
    chunk = scratch_slice((size_t)ZC_TABLE_STRIDE + 2 * ((size_t)ZE_FRAME_STRIDE(ZC_MAX_SRC) + 12), chunk, 1);
    if (const char* ov = zj_tune("ZJNI_CD_SLICE")) { size_t const v = (size_t)atoll(ov); if (v >= 64) chunk = v; }
    size_t const slice = n < chunk ? n : chunk;
    size_t const fsBytes = slice * (size_t)ZE_FRAME_STRIDE(ZC_MAX_SRC), metaBytes = (slice * 12 + 255) & ~(size_t)255, tablesBytes = slice * (size_t)ZC_TABLE_STRIDE;
    size_t const need = tablesBytes + 2 * (fsBytes + metaBytes) + 256;
    if (d->cdBufCap < need || d->cdSliceCap < slice) {
        if (!scratch_make_room(d, d->cdBufCap, need)) return ZJNI_ERR(64);
        if (d->cdBuf) { if (hipDeviceSynchronize() != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device); (void)hipFree(d->cdBuf); d->cdBuf = nullptr; d->cdBufCap = 0; }
        if (hipMalloc(&d->cdBuf, need) != hipSuccess) return ZJNI_ERR(64);
        d->cdBufCap = need; d->cdSliceCap = slice;
    }
    if (d->cdListCap < slice) {
        if (d->cdList) { if (hipDeviceSynchronize() != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device); (void)hipFree(d->cdList); d->cdList = nullptr; d->cdListCap = 0; }
        if (hipMalloc(&d->cdList, 2 * (slice + 1024) * sizeof(u32)) != hipSuccess) return ZJNI_ERR(64);
        d->cdListCap = slice;
    }
    size_t const sliceCap = d->cdSliceCap;
    size_t const fsB = sliceCap * (size_t)ZE_FRAME_STRIDE(ZC_MAX_SRC), metaB = (sliceCap * 12 + 255) & ~(size_t)255;
    u8* const tables = d->cdBuf;
    const ZECDictDev* const cd = (const ZECDictDev*)cdict->buf;
    unsigned long long* const eprof = d->prof ? d->prof + 16 : nullptr;
    int pending[2] = {0, 0};
    auto bail = [&](size_t code) {                       // an error mid-way still orders the caller's stream after the entropy kernels already posted
        for (int p = 0; p < 2; p++) if (pending[p]) (void)hipStreamWaitEvent(st, d->cdEncDone[p], 0);
        return code;
    };
    size_t s = 0;
    for (size_t at = 0; at < n; at += chunk, s++) {
        size_t const m = n - at < chunk ? n - at : chunk;
        int const par = (int)(s & 1);
        u8* const fscratch = d->cdBuf + sliceCap * (size_t)ZC_TABLE_STRIDE + (size_t)par * (fsB + metaB); u32* const meta = (u32*)(fscratch + fsB);
        u32* const list = d->cdList + (size_t)par * (d->cdListCap + 1024);
        u32* const ctr = d->counters + 32 + 8 * par;              // [0] |list|, [1] unused, [2] entropy work, [4] match work
        const u64* const so = (const u64*)d_src_off + at; const u64* const dofs = (const u64*)d_dst_off + at; u64* const res = (u64*)d_result + at;
        if (pending[par]) { if (hipStreamWaitEvent(st, d->cdEncDone[par], 0) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device)); pending[par] = 0; }   // slice s-2 is done with this set
        if (hipMemsetAsync(ctr, 0, 32, st) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
        hipLaunchKernelGGL(zj_enc_classify_kernel, dim3((u32)((m + 255) / 256)), dim3(256), 0, st, so, res, (u32)m, 1u, 0xFFFFFFFFu, ctr, list, list, (u32*)nullptr);   // every frame <= 128 KiB is listed
        hipLaunchKernelGGL(zj_cdict_zero_tables_kernel, dim3((u32)(m < 16384 ? m : 16384)), dim3(256), 0, st, so, cd, (const u32*)list, (const u32*)ctr, tables);
        u32 const waves = (u32)((m + 63) / 64);
        u32 const gridM = waves < (u32)d->matchGrid ? waves : (u32)d->matchGrid;
        (void)hipEventRecord(d->tev[0], st);
        hipLaunchKernelGGL(zj_enc_match_dict_kernel, dim3(gridM), dim3(64), 0, st, (const u8*)d_src, so, cd, (const u32*)list, (const u32*)ctr, ctr + 4, tables, fscratch, meta);
        (void)hipEventRecord(d->tev[1], st); d->tevCompress = true;
        if (hipEventRecord(d->cdMatchDone[par], st) != hipSuccess || hipStreamWaitEvent(d->sideStream, d->cdMatchDone[par], 0) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
        u32 const gridA = (u32)(m < (size_t)d->encGridSmall ? m : (size_t)d->encGridSmall);
        hipLaunchKernelGGL(zj_encode_kernel, dim3(gridA), dim3(64), ZE_SMALL_LDS_BYTES, d->sideStream, (const u8*)d_src, so, (u8*)d_dst, dofs, res, (u32)cdict->level,
                           (const u32*)list, (const u32*)ctr, ctr + 2, d->encScratch, eprof, fscratch, ZC_MAX_SRC, (const u32*)meta, 0u, (const u32*)nullptr, (u32*)nullptr, flags, cd,
                           (u32)ZE_SMALL_LDS_BYTES, 0u, 0xFFFFFFFFu);
        if (hipEventRecord(d->cdEncDone[par], d->sideStream) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
        pending[par] = 1;
    }
    for (int par = 0; par < 2; par++) if (pending[par] && hipStreamWaitEvent(st, d->cdEncDone[par], 0) != hipSuccess) return bail(ZJNI_ERR(ZJNI_ERROR_no_device));
    {   // sources beyond the attach range (they got parameter_unsupported above): copy mode, if the call has any
        if (!ensure_multi_tables(d)) return ZJNI_ERR(64);
        u32* const cc = d->counters + 56;            // [0] copy-mode frames of the call, [1] work
        if (hipMemsetAsync(cc, 0, 8, st) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
        hipLaunchKernelGGL(zj_cdict_count_copy_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, (const u64*)d_src_off, (u32)n, cd, cc);
        u32 const gc = (u32)(n < (size_t)d->multiGrid ? n : (size_t)d->multiGrid);
        hipLaunchKernelGGL(zj_encode_cdict_copy_kernel, dim3(gc), dim3(64), (u32)sizeof(ZEEntropy), st, (const u8*)d_src, (const u64*)d_src_off, (u8*)d_dst, (const u64*)d_dst_off,
                           (u64*)d_result, (u32)n, cd, (const u32*)cc, cc + 1, d->encScratch, d->multiTables, flags, (u32)sizeof(ZEEntropy));
    }
    return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
}

// The first frame of a buffer, read on the host: its compressed size (0 = not a complete plain frame) and its content size (~0 = not in the header).
// Only used to bound the destination staging of the host-pointer decompress entry: a buffer that IS one frame with a known content size needs that
// many bytes, not the caller's whole capacity (a 4 KiB frame decoded into a 64 MiB buffer would otherwise pin and move 64 MiB).  Format: zstd
// compression format 1.5.7, frame header (N/decompress/zstd_decompress.c:447-557) and block headers (ZSTD_getcBlockSize, :1-3 bytes).
static size_t host_frame_extent(const u8* p, size_t n, u64* content) {
    *content = ~(u64)0;
    if (n < 9 || ld32(p) != 0xFD2FB528u) return 0;
    u32 const fhd = p[4], didc = fhd & 3, single = (fhd >> 5) & 1, fcsid = fhd >> 6, cks = (fhd >> 2) & 1;
    if (fhd & 8) return 0;
    u32 const didSz = didc == 3 ? 4 : didc, fcsSz = fcsid == 0 ? single : (1u << fcsid);
    size_t pos = 5 + !single + didSz;
    if (n < pos + fcsSz + 3) return 0;
    if (fcsid == 0) { if (single) *content = p[pos]; } else if (fcsid == 1) *content = (u64)ld16(p + pos) + 256; else if (fcsid == 2) *content = ld32(p + pos); else *content = ld64(p + pos);
    pos += fcsSz;
    for (;;) {
        if (pos + 3 > n) return 0;
        u32 const bh = (u32)p[pos] | ((u32)p[pos + 1] << 8) | ((u32)p[pos + 2] << 16), type = (bh >> 1) & 3, bs = bh >> 3;
        if (type == 3) return 0;
        pos += 3 + (type == 1 ? 1 : bs);
        if (pos > n) return 0;
        if (bh & 1) break;
    }
    pos += cks ? 4 : 0;
    return pos <= n ? pos : 0;
}

// ZSTD_findFrameCompressedSize + ZSTD_decompressBound for ONE complete zstd frame at src (N/decompress/zstd_decompress.c:739-850): the bytes the frame occupies,
// its content size from the header (~0: not recorded) and an upper bound of what it decodes to (raw / RLE blocks exactly, compressed blocks at the block maximum).
// 0: not a complete, well-formed zstd frame (skippable frames, truncated input, reserved bits): the caller keeps the bundled library's stream.
size_t zjni_frame_extent(const void* srcv, size_t srcSize, unsigned long long* content, unsigned long long* bound) {
    const u8* const p = (const u8*)srcv; u64 c = ~(u64)0;
    size_t const ext = srcv ? host_frame_extent(p, srcSize, &c) : 0;
    if (content) *content = c;
    if (bound) *bound = 0;
    if (!ext) return 0;
    u32 const fhd = p[4], didc = fhd & 3, single = (fhd >> 5) & 1, fcsid = fhd >> 6;
    u32 const didSz = didc == 3 ? 4 : didc, fcsSz = fcsid == 0 ? single : (1u << fcsid);
    u64 blockMax = 128u << 10;
    if (!single) { u32 const wl = (p[5] >> 3) + 10u; if (wl > 31u) return 0; u64 const w = ((u64)1 << wl) + (((u64)1 << wl) >> 3) * (p[5] & 7u); if (w < blockMax) blockMax = w; }
    else if (c < blockMax) blockMax = c;
    size_t pos = 5 + !single + didSz + fcsSz; u64 b = 0;
    for (;;) {
        u32 const bh = (u32)p[pos] | ((u32)p[pos + 1] << 8) | ((u32)p[pos + 2] << 16), type = (bh >> 1) & 3, bs = bh >> 3;
        b += type == 2 ? blockMax : bs;
        pos += 3 + (type == 1 ? 1 : bs);
        if (bh & 1) break;
    }
    if (bound) *bound = (c != ~(u64)0) ? c : b;
    return ext;
}

// Host-pointer decompress as a three-stage pipeline over slices of the batch: while slice k's frames are decoded, slice k + 1's sources cross the
// link one way and slice k - 1's output the other (PCIe: 57 GB/s each way alone, 2 x 49 at once on this box: tools/micro/pcie.py), and the host
// threads gather / scatter the slices on either side.  The first slice is small (its transfer and kernels are the pipeline's lead-in), the others
// large enough for the lane-per-frame sequence decode, whose time is its longest frame's chain whatever the slice size.
static size_t host_decompress_locked(DevState* d, StageSlot* sl, const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n,
                                     const zjni_ddict* ddict, bool exactCaps);
static size_t host_decompress(DevState* d, const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n,
                              const zjni_ddict* ddict) {
    // One host-pointer DECOMPRESS at a time per device: the call is a three-stream pipeline over its own slices already, at the link's and the host threads' limit; a second one
    // beside it only shares both (measured: 41 GiB/s one after the other, 27 with two in flight, profiles/r05/e_).  It still runs beside a compress call's kernels.
    std::lock_guard<std::mutex> oneDecompress(*d->hostDecompMu);
    SlotLock slotLock(d);                         // one of the device's staging slots, for both passes
    size_t const r = host_decompress_locked(d, slotLock.s, src, srcSize, dst, dstCap, result, n, ddict, false);
    if (zjni_isError(r)) return r;
    // A frame that was given its header's content size as capacity and overran it is damaged; the reference, decoding into the caller's larger
    // buffer, goes on to the frame's end and answers from there (usually corruption_detected, not dstSize_tooSmall): those few again, with the caller's capacity
    std::vector<size_t> again;
    for (size_t i = 0; i < n; i++) if (result[i] == ZJNI_ERR(70)) { u64 c; if (srcSize[i] && host_frame_extent((const u8*)src[i], srcSize[i], &c) == srcSize[i] && c != ~(u64)0 && c < dstCap[i]) again.push_back(i); }
    if (again.empty()) return 0;
    size_t const m = again.size();
    std::vector<const void*> s2(m); std::vector<size_t> z2(m), c2(m), r2(m); std::vector<void*> d2(m);
    for (size_t j = 0; j < m; j++) { size_t const i = again[j]; s2[j] = src[i]; z2[j] = srcSize[i]; d2[j] = dst[i]; c2[j] = dstCap[i]; }
    size_t const rr = host_decompress_locked(d, slotLock.s, s2.data(), z2.data(), d2.data(), c2.data(), r2.data(), m, ddict, true);
    if (zjni_isError(rr)) return rr;
    for (size_t j = 0; j < m; j++) result[again[j]] = r2[j];
    return 0;
}
static size_t host_decompress_locked(DevState* d, StageSlot* sl, const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n,
                                     const zjni_ddict* ddict, bool exactCaps) {
    size_t const offBytes = (n + 1) * 8;
    // what each buffer needs on the device: the frame's own content size when the buffer is exactly one frame that says it, else the caller's capacity
    std::vector<u64> need(n + 1, 0), srcAt(n + 1, 0);
    {   std::vector<u64> idx(n + 1); for (size_t i = 0; i <= n; i++) idx[i] = i;
        par_ranges(idx.data(), n, [&](size_t a0, size_t a1) {
            for (size_t i = a0; i < a1; i++) {
                u64 c; size_t const ext = srcSize[i] ? host_frame_extent((const u8*)src[i], srcSize[i], &c) : 0;
                need[i] = (!exactCaps && ext == srcSize[i] && c != ~(u64)0 && c <= dstCap[i]) ? c : dstCap[i];
            }
        }); }
    u64 srcTotal = 0, dstTotal = 0;
    for (size_t i = 0; i < n; i++) { u64 const c = need[i]; need[i] = dstTotal; srcAt[i] = srcTotal; dstTotal += c; srcTotal += srcSize[i]; }
    need[n] = dstTotal; srcAt[n] = srcTotal;                       // (need[] now holds destination offsets)
    // staging layout: [srcOff][dstOff][result][src blob][dst blob]
    size_t const oSrcOff = 0, oDstOff = offBytes, oRes = 2 * offBytes, oSrc = 3 * offBytes, oDst = (oSrc + (size_t)srcTotal + 15) & ~(size_t)15;
    size_t const total = oDst + (size_t)dstTotal + 16;
    if (!ensure_staging(sl, total)) return ZJNI_ERR(ZJNI_ERROR_unsupported);      // no room to stage this batch: the caller's CPU path takes it
    u64* const hs = (u64*)(sl->hPinned + oSrcOff); u64* const hd = (u64*)(sl->hPinned + oDstOff); const u64* const hr = (const u64*)(sl->hPinned + oRes);
    memcpy(hs, srcAt.data(), offBytes); memcpy(hd, need.data(), offBytes);
    u8* const hSrc = sl->hPinned + oSrc; u8* const hDst = sl->hPinned + oDst;
    // slices by destination bytes: a small first one, then up to eight of at least ZJ_HOST_SLICE
    std::vector<size_t> cuts; cuts.push_back(0);
    {   u64 const big = dstTotal / 8 > ZJ_HOST_SLICE ? dstTotal / 8 : ZJ_HOST_SLICE; u64 target = big / 4;
        for (size_t lo = 0; lo < n;) { size_t hi = lo + 1; while (hi < n && hd[hi] - hd[lo] < target) hi++; cuts.push_back(hi); lo = hi; target = big; } }
    size_t const nSlices = cuts.size() - 1;
    if (sl->pipeEv.size() < 3 * nSlices) {
        size_t const have = sl->pipeEv.size(); sl->pipeEv.resize(3 * nSlices, nullptr);
        for (size_t k = have; k < 3 * nSlices; k++) if (hipEventCreateWithFlags(&sl->pipeEv[k], hipEventDisableTiming) != hipSuccess) { sl->pipeEv.resize(k); return ZJNI_ERR(ZJNI_ERROR_no_device); }
    }
    auto drain = [&](size_t code) { (void)hipStreamSynchronize(sl->hostIn); (void)hipStreamSynchronize(sl->hostK); (void)hipStreamSynchronize(sl->hostOut); return code; };   // nothing of ours in flight when the staging lock drops
    int const T = host_threads();
    std::atomic<bool> gathering(true);               // while the sources are still being gathered the two sides share the host's threads
    auto scatter = [&](size_t lo, size_t hi) {
        par_ranges(hd + lo, hi - lo, [&](size_t a0, size_t a1) {
            for (size_t i = lo + a0; i < lo + a1; i++) {
                result[i] = (size_t)hr[i];
                if (!zjni_isError(result[i]) && result[i]) memcpy(dst[i], hDst + hd[i], result[i]);
            }
        }, gathering.load() ? (T + 1) / 2 : T);
    };
    // the returning side runs on a thread of its own: slice k is handed to the caller as soon as it is back, whatever the enqueueing side is doing.
    // With one usable CPU (or when the thread cannot be created) the calling thread returns the slices itself after the last one is enqueued.
    std::mutex qm; std::condition_variable qcv; size_t enqueued = 0; std::atomic<bool> failed(false);
    auto return_slices = [&]() {
        (void)hipSetDevice(d->ordinal);
        for (size_t k = 0; k < nSlices; k++) {
            {   std::unique_lock<std::mutex> lk(qm); qcv.wait(lk, [&] { return enqueued > k || failed.load(); }); if (enqueued <= k) return; }
            if (hipEventSynchronize(sl->pipeEv[3 * k + 2]) != hipSuccess) { failed.store(true); return; }
            scatter(cuts[k], cuts[k + 1]);
        }
    };
    std::thread returner;
    if (T > 1) { try { returner = std::thread(return_slices); } catch (const std::system_error&) { /* no thread to be had: the serial path below */ } }
    bool const threaded = returner.joinable();
    struct Joiner { std::thread& t; std::atomic<bool>& f; std::mutex& m; std::condition_variable& cv; bool ok = false;
                    ~Joiner() { if (!ok) { std::lock_guard<std::mutex> g(m); f.store(true); } cv.notify_all(); if (t.joinable()) t.join(); } } joiner{returner, failed, qm, qcv};
    if (hipMemcpyAsync(sl->dStage, sl->hPinned, 2 * offBytes, hipMemcpyHostToDevice, sl->hostIn) != hipSuccess) return drain(ZJNI_ERR(ZJNI_ERROR_no_device));
    for (size_t k = 0; k < nSlices; k++) {
        size_t const lo = cuts[k], hi = cuts[k + 1];
        hipEvent_t const evIn = sl->pipeEv[3 * k], evK = sl->pipeEv[3 * k + 1], evOut = sl->pipeEv[3 * k + 2];
        par_ranges(hs + lo, hi - lo, [&](size_t a0, size_t a1) { for (size_t i = lo + a0; i < lo + a1; i++) if (srcSize[i]) memcpy(hSrc + hs[i], src[i], srcSize[i]); }, k == 0 ? T : (T + 1) / 2);
        if (hs[hi] > hs[lo] && hipMemcpyAsync(sl->dStage + oSrc + hs[lo], hSrc + hs[lo], (size_t)(hs[hi] - hs[lo]), hipMemcpyHostToDevice, sl->hostIn) != hipSuccess) return drain(ZJNI_ERR(ZJNI_ERROR_no_device));
        if (hipEventRecord(evIn, sl->hostIn) != hipSuccess || hipStreamWaitEvent(sl->hostK, evIn, 0) != hipSuccess) return drain(ZJNI_ERR(ZJNI_ERROR_no_device));
        size_t const r = zjni_decompress_batch_device_usingDDict(sl->dStage + oSrc, (const u64*)(sl->dStage + oSrcOff) + lo, sl->dStage + oDst, (const u64*)(sl->dStage + oDstOff) + lo,
                                                                 (u64*)(sl->dStage + oRes) + lo, hi - lo, ddict, sl->hostK);
        if (zjni_isError(r)) return drain(r);
        if (hipEventRecord(evK, sl->hostK) != hipSuccess || hipStreamWaitEvent(sl->hostOut, evK, 0) != hipSuccess) return drain(ZJNI_ERR(ZJNI_ERROR_no_device));
        if (hipMemcpyAsync(sl->hPinned + oRes + lo * 8, sl->dStage + oRes + lo * 8, (hi - lo) * 8, hipMemcpyDeviceToHost, sl->hostOut) != hipSuccess) return drain(ZJNI_ERR(ZJNI_ERROR_no_device));
        if (hd[hi] > hd[lo] && hipMemcpyAsync(hDst + hd[lo], sl->dStage + oDst + hd[lo], (size_t)(hd[hi] - hd[lo]), hipMemcpyDeviceToHost, sl->hostOut) != hipSuccess) return drain(ZJNI_ERR(ZJNI_ERROR_no_device));
        if (hipEventRecord(evOut, sl->hostOut) != hipSuccess) return drain(ZJNI_ERR(ZJNI_ERROR_no_device));
        {   std::lock_guard<std::mutex> g(qm); enqueued = k + 1; }
        qcv.notify_all();
    }
    gathering.store(false);
    joiner.ok = true;
    if (threaded) returner.join(); else return_slices();
    if (failed.load()) return drain(ZJNI_ERR(ZJNI_ERROR_no_device));
    return 0;
}

// ---- host-pointer batches: pack -> H2D -> kernel -> D2H -> scatter ------------------------------
// ZJNI_HOST_TRACE=1: the phases of a host-pointer compress call on stderr, milliseconds since the process's first one (two calls in flight: which phase waits for what)
static void host_trace(const void* job, const char* what) {
    static bool const on = zj_env("ZJNI_HOST_TRACE") != nullptr;
    if (!on) return;
    static auto const t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "[zjni host %p] %8.1f ms  %s\n", job, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
}
static size_t host_batch(bool compress, const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap,
                         size_t* result, size_t n, int level, int checksum = 0, const zjni_ddict* ddict = nullptr, const zjni_cdict* cdict = nullptr) {
    DevState* d = cur_state();
    if (!d) return ZJNI_ERR(ZJNI_ERROR_no_device);
    if (n == 0) return 0;
    size_t srcTotal = 0, dstTotal = 0;
    for (size_t i = 0; i < n; i++) {                   // one staging blob each way: its size must be representable (and allocatable)
        if (srcSize[i] > ((size_t)1 << 46) || dstCap[i] > ((size_t)1 << 46)) return ZJNI_ERR(64);
        srcTotal += srcSize[i]; dstTotal += dstCap[i];
        if (srcTotal > ((size_t)1 << 46) || dstTotal > ((size_t)1 << 46)) return ZJNI_ERR(64);
        if ((srcSize[i] && !src[i]) || (dstCap[i] && !dst[i])) return ZJNI_ERR(compress ? 72 : 72);
    }
    if (!compress) return host_decompress(d, src, srcSize, dst, dstCap, result, n, ddict);
    size_t const offBytes = (n + 1) * 8;
    // staging layout: [srcOff][dstOff][result][packedOff][src blob][dst blob][device only, compress: packed frames]
    size_t const oSrcOff = 0, oDstOff = offBytes, oRes = 2 * offBytes, oPOff = 3 * offBytes, oSrc = 4 * offBytes, oDst = (oSrc + srcTotal + 15) & ~(size_t)15;
    size_t const oPack = (oDst + dstTotal + 15) & ~(size_t)15;
    size_t const total = oPack + (compress ? dstTotal : 0) + 16;
    host_trace(result, "call");
    SlotLock slotLock(d); StageSlot* const sl = slotLock.s;      // one of the device's staging slots: a second caller stages and copies while this one's kernels run
    if (!ensure_staging(sl, total)) return ZJNI_ERR(ZJNI_ERROR_unsupported);     // no room to stage this batch: the caller's CPU path takes it
    hipStream_t const hst = sl->hostK;            // everything of this call in order on the slot's own stream
    struct Drain { hipStream_t st; ~Drain() { (void)hipStreamSynchronize(st); } } drainOnExit{hst};     // error returns below leave copies in flight on the staging area: none when the slot is given back
    u64* hs = (u64*)(sl->hPinned + oSrcOff); u64* hd = (u64*)(sl->hPinned + oDstOff); u64* hp = (u64*)(sl->hPinned + oPOff);
    size_t a = 0, b = 0;
    for (size_t i = 0; i < n; i++) { hs[i] = a; hd[i] = b; a += srcSize[i]; b += dstCap[i]; }
    hs[n] = a; hd[n] = b;
    u8* const hSrc = sl->hPinned + oSrc;
    // in slices of ~256 MiB: while the link carries slice k the host threads gather slice k + 1 into the pinned area
    if (hipMemcpyAsync(sl->dStage, sl->hPinned, oSrc, hipMemcpyHostToDevice, hst) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
    for (size_t lo = 0; lo < n;) {
        size_t hi = lo + 1; while (hi < n && hs[hi] - hs[lo] < ZJ_HOST_SLICE) hi++;
        par_ranges(hs + lo, hi - lo, [&](size_t a0, size_t a1) { for (size_t i = lo + a0; i < lo + a1; i++) if (srcSize[i]) memcpy(hSrc + hs[i], src[i], srcSize[i]); });
        if (hs[hi] > hs[lo] && hipMemcpyAsync(sl->dStage + oSrc + hs[lo], hSrc + hs[lo], (size_t)(hs[hi] - hs[lo]), hipMemcpyHostToDevice, hst) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
        lo = hi;
    }
    host_trace(result, "sources gathered, H2D enqueued");
    size_t r;
    if (compress && cdict)
        r = zjni_compress_batch_device_usingCDict(sl->dStage + oSrc, (const u64*)(sl->dStage + oSrcOff), sl->dStage + oDst, (const u64*)(sl->dStage + oDstOff),
                                                  (u64*)(sl->dStage + oRes), n, cdict, checksum, hst);
    else if (compress)                             // `level` may be a level word (zjni_compress_batch_advanced)
        r = compress_chunked(sl->dStage + oSrc, (const u64*)(sl->dStage + oSrcOff), sl->dStage + oDst, (const u64*)(sl->dStage + oDstOff),
                             (u64*)(sl->dStage + oRes), n, level, zj_frame_flags(checksum), hst);
    else
        r = zjni_decompress_batch_device_usingDDict(sl->dStage + oSrc, (const u64*)(sl->dStage + oSrcOff), sl->dStage + oDst, (const u64*)(sl->dStage + oDstOff),
                                                    (u64*)(sl->dStage + oRes), n, ddict, hst);
    if (zjni_isError(r)) return r;
    host_trace(result, "kernels enqueued");
    if (hipMemcpyAsync(sl->hPinned + oRes, sl->dStage + oRes, n * 8, hipMemcpyDeviceToHost, hst) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
    const u64* hr = (const u64*)(sl->hPinned + oRes);
    const u64* from = hd;                              // where frame i's bytes start inside the returned blob
    if (compress) {
        // the destinations are compressBound-sized: pack the frames on the device and bring back only their bytes
        if (hipStreamSynchronize(hst) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
        host_trace(result, "kernels done");
        u64 acc = 0;
        for (size_t i = 0; i < n; i++) { hp[i] = acc; if (!zjni_isError((size_t)hr[i])) acc += hr[i]; }
        hp[n] = acc;
        if (hipMemcpyAsync(sl->dStage + oPOff, hp, offBytes, hipMemcpyHostToDevice, hst) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
        size_t const pr = zjni_pack_batch_device(sl->dStage + oDst, (const u64*)(sl->dStage + oDstOff), (const u64*)(sl->dStage + oRes),
                                                 sl->dStage + oPack, (const u64*)(sl->dStage + oPOff), n, hst);
        if (zjni_isError(pr)) return pr;
        from = hp;
    }
    // the way back in slices too: the host threads scatter slice k into the caller's buffers while the link carries slice k + 1
    u8* const dOut = sl->dStage + (compress ? oPack : oDst);
    u8* const hDst = sl->hPinned + oDst;
    std::vector<size_t> cuts; cuts.push_back(0);
    for (size_t lo = 0; lo < n;) { size_t hi = lo + 1; while (hi < n && from[hi] - from[lo] < ZJ_HOST_SLICE) hi++; cuts.push_back(hi); lo = hi; }
    size_t const nSlices = cuts.size() - 1;
    if (sl->stageEv.size() < nSlices) {
        size_t const have = sl->stageEv.size(); sl->stageEv.resize(nSlices, nullptr);
        for (size_t k = have; k < nSlices; k++) if (hipEventCreateWithFlags(&sl->stageEv[k], hipEventDisableTiming) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
    }
    for (size_t k = 0; k < nSlices; k++) {
        size_t const lo = cuts[k], hi = cuts[k + 1];
        if (from[hi] > from[lo] && hipMemcpyAsync(hDst + from[lo], dOut + from[lo], (size_t)(from[hi] - from[lo]), hipMemcpyDeviceToHost, hst) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
        if (hipEventRecord(sl->stageEv[k], hst) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
    }
    host_trace(result, "D2H enqueued");
    for (size_t k = 0; k < nSlices; k++) {
        size_t const lo = cuts[k], hi = cuts[k + 1];
        if (hipEventSynchronize(sl->stageEv[k]) != hipSuccess) return ZJNI_ERR(ZJNI_ERROR_no_device);
        if (k + 1 == nSlices) host_trace(result, "last slice back");
        par_ranges(from + lo, hi - lo, [&](size_t a0, size_t a1) {
            for (size_t i = lo + a0; i < lo + a1; i++) {
                result[i] = (size_t)hr[i];
                if (!zjni_isError(result[i]) && result[i]) memcpy(dst[i], hDst + from[i], result[i]);
            }
        });
    }
    host_trace(result, "scattered: done");
    return 0;
}

size_t zjni_decompress_batch(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n) {
    return host_batch(false, src, srcSize, dst, dstCap, result, n, 0);
}

// ---- asynchronous host batches (include/zjni_amd.h: two in flight through the device's two staging slots) ----
struct zjni_batch_job { std::thread worker; size_t code = 0; };
static zjni_batch_job* batch_begin(bool compress, const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap,
                                   size_t* result, size_t n, int level, int checksum) {
    DevState* d = cur_state();
    if (!d) return nullptr;
    int const ordinal = d->ordinal;
    zjni_batch_job* job = new (std::nothrow) zjni_batch_job();
    if (!job) return nullptr;
    try {
        job->worker = std::thread([=]() {
            if (zjni_init(ordinal) != 0) { job->code = ZJNI_ERR(ZJNI_ERROR_no_device); return; }      // binds the worker to the caller's device
            job->code = compress ? zjni_compress_batch2(src, srcSize, dst, dstCap, result, n, level, checksum) : zjni_decompress_batch(src, srcSize, dst, dstCap, result, n);
        });
    } catch (const std::system_error&) { delete job; return nullptr; }
    return job;
}
zjni_batch_job* zjni_compress_batch_begin(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n, int level, int checksum) {
    return batch_begin(true, src, srcSize, dst, dstCap, result, n, level, checksum);
}
zjni_batch_job* zjni_decompress_batch_begin(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n) {
    return batch_begin(false, src, srcSize, dst, dstCap, result, n, 0, 0);
}
size_t zjni_batch_finish(zjni_batch_job* job) {
    if (!job) return ZJNI_ERR(ZJNI_ERROR_no_device);
    if (job->worker.joinable()) job->worker.join();
    size_t const code = job->code;
    delete job;
    return code;
}

// ---- one host batch over several devices of this process (SURVEY.md section 8e at the boundary a JVM has: one process) ----
// The batch is cut into contiguous index ranges of about equal source bytes, one per listed device; a thread per device binds it
// (zjni_init) and runs its range through the host-pointer path.  mode 0: every device returns its frames to the host over its
// own PCIe link (no inter-GPU traffic — the right choice when the consumer is host memory, as with JVM buffers).  mode 1
// (compress only): the devices pack their frames and send them to devices[0] over xGMI (peer copies, the gather of section 8e),
// which returns the whole batch to the host in one transfer.
static void multi_ranges(const size_t* srcSize, size_t n, int nd, std::vector<size_t>& cut) {
    size_t total = 0; for (size_t i = 0; i < n; i++) total += srcSize[i];
    cut.assign((size_t)nd + 1, n); cut[0] = 0;
    size_t acc = 0; int k = 1;
    for (size_t i = 0; i < n && k < nd; i++) {
        acc += srcSize[i];
        while (k < nd && acc * (size_t)nd >= total * (size_t)k && acc > 0) cut[(size_t)k++] = i + 1;
    }
    for (; k < nd; k++) cut[(size_t)k] = n;
    for (int j = 1; j <= nd; j++) if (cut[(size_t)j] < cut[(size_t)j - 1]) cut[(size_t)j] = cut[(size_t)j - 1];
}
static size_t multi_run(bool compress, const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n,
                        int level, int checksum, const int* devices, int nDevices) {
    if (n == 0) return 0;
    if (!devices || nDevices < 1 || nDevices > 64) return ZJNI_ERR(42);
    for (int j = 0; j < nDevices; j++) if (devices[j] < 0 || devices[j] >= dev_count()) return ZJNI_ERR(ZJNI_ERROR_no_device);
    std::vector<size_t> cut; multi_ranges(srcSize, n, nDevices, cut);
    std::vector<size_t> rc((size_t)nDevices, 0);
    std::vector<std::thread> th;
    for (int j = 0; j < nDevices; j++) th.emplace_back([&, j]() {
        size_t const lo = cut[(size_t)j], cnt = cut[(size_t)j + 1] - lo;
        if (!cnt) return;
        if (zjni_init(devices[j]) != 0) { rc[(size_t)j] = ZJNI_ERR(ZJNI_ERROR_no_device); return; }
        rc[(size_t)j] = host_batch(compress, src + lo, srcSize + lo, dst + lo, dstCap + lo, result + lo, cnt, level, checksum);
    });
    for (auto& t : th) t.join();
    for (int j = 0; j < nDevices; j++) if (zjni_isError(rc[(size_t)j])) return rc[(size_t)j];
    return 0;
}
// mode 1: compress on every device, pack there, peer-copy the packed frames to devices[0], one D2H from there
static size_t multi_compress_gather(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n,
                                    int level, int checksum, const int* devices, int nDevices) {
    if (n == 0) return 0;
    if (!devices || nDevices < 1 || nDevices > 64) return ZJNI_ERR(42);
    for (int j = 0; j < nDevices; j++) if (devices[j] < 0 || devices[j] >= dev_count()) return ZJNI_ERR(ZJNI_ERROR_no_device);
    std::vector<size_t> cut; multi_ranges(srcSize, n, nDevices, cut);
    struct Part { u8* dSrc = nullptr; u8* dComp = nullptr; u8* dPacked = nullptr; u64* dOff = nullptr; u64* dCOff = nullptr; u64* dRes = nullptr; u64* dPOff = nullptr;
                  std::vector<u64> res, poff; size_t packed = 0, rc = 0; };
    std::vector<Part> parts((size_t)nDevices);
    auto phase1 = [&](int j) {
        Part& p = parts[(size_t)j]; size_t const lo = cut[(size_t)j], cnt = cut[(size_t)j + 1] - lo;
        if (!cnt) return;
        if (zjni_init(devices[j]) != 0) { p.rc = ZJNI_ERR(ZJNI_ERROR_no_device); return; }
        size_t st = 0, ct = 0; std::vector<u64> so(cnt + 1), co(cnt + 1);
        for (size_t i = 0; i < cnt; i++) { so[i] = st; co[i] = ct; st += srcSize[lo + i]; ct += dstCap[lo + i]; }
        so[cnt] = st; co[cnt] = ct;
        std::vector<u8> stage(st + 16);
        for (size_t i = 0; i < cnt; i++) if (srcSize[lo + i]) memcpy(stage.data() + so[i], src[lo + i], srcSize[lo + i]);
        bool ok = hipMalloc(&p.dSrc, st + 16) == hipSuccess && hipMalloc(&p.dComp, ct + 16) == hipSuccess && hipMalloc(&p.dOff, (cnt + 1) * 8) == hipSuccess
               && hipMalloc(&p.dCOff, (cnt + 1) * 8) == hipSuccess && hipMalloc(&p.dRes, cnt * 8) == hipSuccess && hipMalloc(&p.dPOff, (cnt + 1) * 8) == hipSuccess;
        ok = ok && hipMemcpy(p.dSrc, stage.data(), st, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(p.dOff, so.data(), (cnt + 1) * 8, hipMemcpyHostToDevice) == hipSuccess
                && hipMemcpy(p.dCOff, co.data(), (cnt + 1) * 8, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) { p.rc = ZJNI_ERR(64); return; }
        size_t const r = zjni_compress_batch_device2(p.dSrc, p.dOff, p.dComp, p.dCOff, p.dRes, cnt, level, checksum, nullptr);
        if (zjni_isError(r)) { p.rc = r; return; }
        p.res.resize(cnt); p.poff.resize(cnt + 1);
        if (hipMemcpy(p.res.data(), p.dRes, cnt * 8, hipMemcpyDeviceToHost) != hipSuccess) { p.rc = ZJNI_ERR(ZJNI_ERROR_no_device); return; }
        size_t a = 0; for (size_t i = 0; i < cnt; i++) { p.poff[i] = a; if (!zjni_isError((size_t)p.res[i])) a += p.res[i]; }
        p.poff[cnt] = a; p.packed = a;
        ok = hipMalloc(&p.dPacked, a + 16) == hipSuccess && hipMemcpy(p.dPOff, p.poff.data(), (cnt + 1) * 8, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) { p.rc = ZJNI_ERR(64); return; }
        size_t const r2 = zjni_pack_batch_device(p.dComp, p.dCOff, p.dRes, p.dPacked, p.dPOff, cnt, nullptr);
        if (zjni_isError(r2) || hipDeviceSynchronize() != hipSuccess) p.rc = ZJNI_ERR(ZJNI_ERROR_no_device);
    };
    {   std::vector<std::thread> th; for (int j = 0; j < nDevices; j++) th.emplace_back(phase1, j); for (auto& t : th) t.join(); }
    size_t rc = 0, total = 0;
    for (auto& p : parts) { if (zjni_isError(p.rc) && !rc) rc = p.rc; total += p.packed; }
    u8* gather = nullptr; std::vector<u8> host;
    if (!rc) {
        if (hipSetDevice(devices[0]) != hipSuccess || hipMalloc(&gather, total + 16) != hipSuccess) rc = ZJNI_ERR(64);
        size_t at = 0;
        for (int j = 0; j < nDevices && !rc; j++) {            // the xGMI step: every device's packed frames land behind each other on devices[0]
            Part& p = parts[(size_t)j];
            if (p.packed && hipMemcpyPeer(gather + at, devices[0], p.dPacked, devices[j], p.packed) != hipSuccess) rc = ZJNI_ERR(ZJNI_ERROR_no_device);
            at += p.packed;
        }
        host.resize(total + 16);
        if (!rc && total && hipMemcpy(host.data(), gather, total, hipMemcpyDeviceToHost) != hipSuccess) rc = ZJNI_ERR(ZJNI_ERROR_no_device);
        size_t base = 0;
        for (int j = 0; j < nDevices && !rc; j++) {
            Part& p = parts[(size_t)j]; size_t const lo = cut[(size_t)j], cnt = cut[(size_t)j + 1] - lo;
            for (size_t i = 0; i < cnt; i++) {
                result[lo + i] = (size_t)p.res[i];
                if (!zjni_isError(result[lo + i]) && result[lo + i]) memcpy(dst[lo + i], host.data() + base + p.poff[i], result[lo + i]);
            }
            base += p.packed;
        }
    }
    if (gather) { (void)hipSetDevice(devices[0]); (void)hipFree(gather); }
    for (int j = 0; j < nDevices; j++) {
        Part& p = parts[(size_t)j];
        if (!p.dSrc && !p.dComp) continue;
        (void)hipSetDevice(devices[j]);
        (void)hipFree(p.dSrc); (void)hipFree(p.dComp); (void)hipFree(p.dPacked); (void)hipFree(p.dOff); (void)hipFree(p.dCOff); (void)hipFree(p.dRes); (void)hipFree(p.dPOff);
    }
    return rc;
}
size_t zjni_compress_batch_multi(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n,
                                 int level, int checksum, const int* devices, int nDevices, int mode) {
    if (level == 0) level = 3;
    if (level < 1 || level > ZJ_LEVEL_MAX) return ZJNI_ERR(42);
    if (mode == 1) return multi_compress_gather(src, srcSize, dst, dstCap, result, n, level, checksum ? 1 : 0, devices, nDevices);
    return multi_run(true, src, srcSize, dst, dstCap, result, n, level, checksum ? 1 : 0, devices, nDevices);
}
size_t zjni_decompress_batch_multi(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n,
                                   const int* devices, int nDevices) {
    return multi_run(false, src, srcSize, dst, dstCap, result, n, 0, 0, devices, nDevices);
}
size_t zjni_compress_batch(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n, int level) {
    if (level == 0) level = 3;                    // ZSTD_CLEVEL_DEFAULT, as ZSTD_c_compressionLevel = 0 means
    if (level < 1 || level > ZJ_LEVEL_MAX) return ZJNI_ERR(42);
    return host_batch(true, src, srcSize, dst, dstCap, result, n, level);
}

size_t zjni_compress_batch2(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n, int level, int checksum) {
    if (level == 0) level = 3;                    // ZSTD_CLEVEL_DEFAULT, as ZSTD_c_compressionLevel = 0 means
    if (level < 1 || level > ZJ_LEVEL_MAX) return ZJNI_ERR(42);
    return host_batch(true, src, srcSize, dst, dstCap, result, n, level, checksum ? 1 : 0);
}

size_t zjni_compress_batch_advanced(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n,
                                    int level, int checksum, int hashLog, int chainLog) {
    if (level == 0) level = 3;                    // ZSTD_CLEVEL_DEFAULT, as ZSTD_c_compressionLevel = 0 means
    int lw; size_t const e = level_word(level, hashLog, chainLog, &lw);
    if (e) return e;
    if (level < 1 || level > ZJ_LEVEL_MAX || (level > 3 && (hashLog | chainLog))) return ZJNI_ERR(42);
    return host_batch(true, src, srcSize, dst, dstCap, result, n, lw, checksum);
}
// ---- cross-thread aggregation of per-buffer calls (second half) ----
// zstd-jni's per-buffer natives are called from many threads, one context each, one buffer per call; a GPU wants thousands of
// buffers per launch.  An aggregator turns concurrent blocking per-buffer calls into batches: the first caller of a kind (compress
// at a level + checksum flag, or decompress) opens a batch and waits up to maxWaitMicros for company, callers that arrive meanwhile
// join it, the opener runs the batch entry once and everybody returns with its own result.  Results are those of the batch entries
// (byte-identical frames); what changes is latency (up to the wait) against launches shared.  One device per aggregator.
struct ZjAggReq { const void* src; size_t srcSize; void* dst; size_t dstCap; size_t result; };
struct ZjAggBatch { std::vector<ZjAggReq*> reqs; bool closed = false, done = false; size_t bytes = 0; };
#define ZJ_AGG_MAX_BYTES ((size_t)2 << 30)          /* staging a batch may ask for (sources + destination capacities): a member that would push it past this opens the next batch */
struct zjni_aggregator {
    int device; size_t maxBatch; unsigned maxWaitMicros;
    std::mutex mu; std::condition_variable cv;
    std::map<int, std::shared_ptr<ZjAggBatch> > open;           // kind -> the batch that is still taking members
    unsigned long long calls = 0, batches = 0;
};
static size_t agg_run(int kind, const void* const* sp, const size_t* ss, void* const* dp, const size_t* dc, size_t* res, size_t n, int level, int checksum) {
    return kind < 0 ? zjni_decompress_batch(sp, ss, dp, dc, res, n) : zjni_compress_batch2(sp, ss, dp, dc, res, n, level, checksum);
}
static size_t agg_submit(zjni_aggregator* a, int kind, ZjAggReq& rq, int level, int checksum) {
    if (!a) return ZJNI_ERR(ZJNI_ERROR_unsupported);
    // a member's own faults are its own: checked before it joins, so that no one else's call fails for them
    if (rq.srcSize > ((size_t)1 << 46) || rq.dstCap > ((size_t)1 << 46)) return ZJNI_ERR(64);
    if ((rq.srcSize && !rq.src) || (rq.dstCap && !rq.dst)) return ZJNI_ERR(72);
    size_t const mine = rq.srcSize + rq.dstCap;
    std::unique_lock<std::mutex> lk(a->mu);
    a->calls++;
    std::shared_ptr<ZjAggBatch> b;
    auto it = a->open.find(kind);
    bool leader = false;
    if (it != a->open.end() && !it->second->closed && !it->second->reqs.empty() && it->second->bytes + mine > ZJ_AGG_MAX_BYTES) {   // would outgrow the staging: that batch goes now
        it->second->closed = true; a->cv.notify_all();
    }
    if (it == a->open.end() || it->second->closed) { b = std::make_shared<ZjAggBatch>(); a->open[kind] = b; leader = true; a->batches++; }
    else b = it->second;
    b->reqs.push_back(&rq); b->bytes += mine;
    if (b->reqs.size() >= a->maxBatch) { b->closed = true; a->cv.notify_all(); }
    if (!leader) {
        a->cv.wait(lk, [&]() { return b->done; });
        return rq.result;
    }
    // the opener: wait for company, close the batch, run it outside the lock
    a->cv.wait_for(lk, std::chrono::microseconds(a->maxWaitMicros), [&]() { return b->closed; });
    b->closed = true;
    if (a->open[kind] == b) a->open.erase(kind);
    std::vector<ZjAggReq*> reqs = b->reqs;                       // (members only join while !closed: the list is final)
    lk.unlock();
    size_t const n = reqs.size();
    std::vector<const void*> sp(n); std::vector<size_t> ss(n), dc(n), res(n); std::vector<void*> dp(n);
    for (size_t i = 0; i < n; i++) { sp[i] = reqs[i]->src; ss[i] = reqs[i]->srcSize; dp[i] = reqs[i]->dst; dc[i] = reqs[i]->dstCap; }
    size_t rc = 0;
    if (zjni_init(a->device) != 0) rc = ZJNI_ERR(ZJNI_ERROR_no_device);
    else rc = agg_run(kind, sp.data(), ss.data(), dp.data(), dc.data(), res.data(), n, level, checksum);
    if (zjni_isError(rc)) {
        // the shared launch failed as a whole (staging, a device error): every member gets its own verdict from a call of its own — what it
        // would have got without the aggregator — instead of a stranger's error
        for (size_t i = 0; i < n; i++) {
            size_t r1 = 0; size_t const rr = n == 1 ? rc : agg_run(kind, &sp[i], &ss[i], &dp[i], &dc[i], &r1, 1, level, checksum);
            res[i] = zjni_isError(rr) ? rr : r1;
        }
    }
    lk.lock();
    for (size_t i = 0; i < n; i++) reqs[i]->result = res[i];
    b->done = true;
    a->cv.notify_all();
    return rq.result;
}
zjni_aggregator* zjni_createAggregator(int device, size_t maxBatch, unsigned maxWaitMicros) {
    if (device < 0 || device >= dev_count() || maxBatch < 1) return nullptr;
    zjni_aggregator* a = new zjni_aggregator();
    a->device = device; a->maxBatch = maxBatch > 65536 ? 65536 : maxBatch; a->maxWaitMicros = maxWaitMicros;
    return a;
}
void zjni_freeAggregator(zjni_aggregator* a) { delete a; }        // no call may be in flight
size_t zjni_aggregator_compress(zjni_aggregator* a, void* dst, size_t dstCap, const void* src, size_t srcSize, int level, int checksum) {
    if (level == 0) level = 3;
    if (level < 1 || level > ZJ_LEVEL_MAX) return ZJNI_ERR(42);
    ZjAggReq rq = { src, srcSize, dst, dstCap, 0 };
    return agg_submit(a, level * 2 + (checksum ? 1 : 0), rq, level, checksum ? 1 : 0);
}
size_t zjni_aggregator_decompress(zjni_aggregator* a, void* dst, size_t dstCap, const void* src, size_t srcSize) {
    ZjAggReq rq = { src, srcSize, dst, dstCap, 0 };
    return agg_submit(a, -1, rq, 0, 0);
}
void zjni_aggregator_stats(zjni_aggregator* a, unsigned long long* calls, unsigned long long* batches) {
    std::lock_guard<std::mutex> lk(a->mu);
    if (calls) *calls = a->calls;
    if (batches) *batches = a->batches;
}

size_t zjni_compress(void* dst, size_t dstCap, const void* src, size_t srcSize, int level) {
    size_t res = 0; const void* s = src; void* dd = dst;
    size_t const r = zjni_compress_batch(&s, &srcSize, &dd, &dstCap, &res, 1, level);
    return zjni_isError(r) ? r : res;
}
size_t zjni_compress2(void* dst, size_t dstCap, const void* src, size_t srcSize, int level, int checksum) {
    size_t res = 0; const void* s = src; void* dd = dst;
    size_t const r = zjni_compress_batch2(&s, &srcSize, &dd, &dstCap, &res, 1, level, checksum);
    return zjni_isError(r) ? r : res;
}
size_t zjni_compress_batch_usingCDict(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n,
                                      const zjni_cdict* cdict, int checksum) {
    if (!cdict) return ZJNI_ERR(32);
    return host_batch(true, src, srcSize, dst, dstCap, result, n, cdict->level, checksum, nullptr, cdict);
}
size_t zjni_compress_usingCDict(void* dst, size_t dstCap, const void* src, size_t srcSize, const zjni_cdict* cdict) {
    size_t res = 0; const void* s = src; void* dd = dst;
    size_t const r = zjni_compress_batch_usingCDict(&s, &srcSize, &dd, &dstCap, &res, 1, cdict, 0);
    return zjni_isError(r) ? r : res;
}
size_t zjni_decompress_batch_usingDDict(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCap, size_t* result, size_t n,
                                        const zjni_ddict* ddict) {
    return host_batch(false, src, srcSize, dst, dstCap, result, n, 0, 0, ddict);
}
size_t zjni_decompress_usingDDict(void* dst, size_t dstCap, const void* src, size_t srcSize, const zjni_ddict* ddict) {
    size_t res = 0; const void* s = src; void* dd = dst;
    size_t const r = zjni_decompress_batch_usingDDict(&s, &srcSize, &dd, &dstCap, &res, 1, ddict);
    return zjni_isError(r) ? r : res;
}
size_t zjni_decompress(void* dst, size_t dstCap, const void* src, size_t srcSize) {
    size_t res = 0; const void* s = src; void* dd = dst;
    size_t const r = zjni_decompress_batch(&s, &srcSize, &dd, &dstCap, &res, 1);
    return zjni_isError(r) ? r : res;
}

void zjni_synth_fill_host(void* dst, size_t bufSize, uint64_t firstIndex, size_t nBuffers) {
    for (size_t i = 0; i < nBuffers; i++) zs_fill((u8*)dst + i * bufSize, (u32)bufSize, firstIndex + i);
}
size_t zjni_synth_fill_device(void* d_dst, size_t bufSize, uint64_t firstIndex, size_t nBuffers, void* stream) {
    if (!cur_state()) return ZJNI_ERR(ZJNI_ERROR_no_device);
    if (nBuffers == 0) return 0;
    hipLaunchKernelGGL(zj_synth_kernel, dim3((u32)((nBuffers + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (u8*)d_dst, (u32)bufSize, firstIndex, (u32)nBuffers);
    return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
}

// exclusive prefix sums of a batch's result sizes (error results count as 0) -> off[0 .. n]: one workgroup, a contiguous run per lane, one scan over the lanes' sums
__global__ __launch_bounds__(1024) void zj_pack_offsets_kernel(const u64* __restrict__ sizes, u64* __restrict__ off, u32 n) {
    __shared__ u64 part[1024];
    u32 const t = threadIdx.x;
    u64 const per = ((u64)n + 1023u) / 1024u, lo0 = (u64)t * per, lo = lo0 < n ? lo0 : n, hi = lo + per < n ? lo + per : n;      // (64-bit: n up to 2^32 - 1 — ADVICE r05)
    u64 sum = 0;
    for (u64 i = lo; i < hi; i++) { u64 const z = sizes[i]; sum += z > ((u64)1 << 40) ? 0 : z; }
    part[t] = sum;
    __syncthreads();
    for (u32 d = 1; d < 1024u; d <<= 1) {                // (Hillis-Steele over 1 024 sums)
        u64 const v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u64 run = t ? part[t - 1] : 0;
    for (u64 i = lo; i < hi; i++) { off[i] = run; u64 const z = sizes[i]; run += z > ((u64)1 << 40) ? 0 : z; }
    if (t == 1023u) off[n] = part[1023];
}
size_t zjni_pack_batch_device2(const void* d_src, const uint64_t* d_src_off, const uint64_t* d_sizes,
                               void* d_dst, uint64_t* d_packed_off, size_t n, void* stream) {
    DevState* d = cur_state();
    if (!d) return ZJNI_ERR(ZJNI_ERROR_no_device);
    if (n > 0xFFFFFFFFull) return ZJNI_ERR(64);
    if (n == 0) return hipMemsetAsync(d_packed_off, 0, 8, (hipStream_t)stream) == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
    hipLaunchKernelGGL(zj_pack_offsets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const u64*)d_sizes, (u64*)d_packed_off, (u32)n);
    return zjni_pack_batch_device(d_src, d_src_off, d_sizes, d_dst, d_packed_off, n, stream);
}
size_t zjni_pack_batch_device(const void* d_src, const uint64_t* d_src_off, const uint64_t* d_sizes,
                              void* d_dst, const uint64_t* d_dst_off, size_t n, void* stream) {
    DevState* d = cur_state();
    if (!d) return ZJNI_ERR(ZJNI_ERROR_no_device);
    if (n == 0) return 0;
    u32 const grid = (u32)(n < (size_t)d->numCU * 8 ? n : (size_t)d->numCU * 8);
    hipLaunchKernelGGL(zj_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u8*)d_src, (const u64*)d_src_off,
                       (const u64*)d_sizes, (u8*)d_dst, (const u64*)d_dst_off, (u32)n);
    return hipGetLastError() == hipSuccess ? 0 : ZJNI_ERR(ZJNI_ERROR_no_device);
}

}  // extern "C"
