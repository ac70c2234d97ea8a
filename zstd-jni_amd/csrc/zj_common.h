// zj_common.h — shared primitives for the gfx950 zstd kernels.
//
// Execution model: one zstd frame is owned by one *group* of W lanes.  On the GPU W = 64 = one
// CDNA4 wavefront = one workgroup, so "group sync" is an LDS-ordering s_barrier of a single wave
// (free) and all cross-lane traffic goes through LDS or DPP.  Kernel bodies are written as
//     GRP_SERIAL(g) { ... }      one lane (lane 0) runs an inherently sequential format step
//     GRP_FOR(g, i, n) { ... }   all lanes stride over i in [0, n)
// with every value that crosses lanes living in the group's LDS block.
//
// The same bodies also instantiate with W = 1 under a plain C++ compiler ("lane-serial build",
// tests/emu/): GRP_SERIAL is always taken and GRP_FOR visits i = 0..n-1 in order.  That build exists
// ONLY so the format logic can be unit-tested in the CPU-only dev container before spending GPU time;
// it is never linked into the product library and the C-ABI never dispatches to it.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZJ_DEV __device__ __forceinline__
#define ZJ_DEVM static __device__ __forceinline__   /* static member functions */
#define ZJ_DEV_MEMBER __device__ __forceinline__
#define ZJ_DEV_NOINLINE __device__ __noinline__
#define ZJ_NO_UNROLL _Pragma("unroll 1")
#define ZJ_HD __host__ __device__ __forceinline__
#define ZJ_ON_GPU 1
#else
#define ZJ_DEV static inline
#define ZJ_DEVM static inline
#define ZJ_DEV_MEMBER inline
#define ZJ_DEV_NOINLINE static
#define ZJ_NO_UNROLL
#define ZJ_HD static inline
#define ZJ_ON_GPU 0
#endif

#if !ZJ_ON_GPU
static inline u32 atomicAdd(u32* p, u32 v) { u32 const o = *p; *p = o + v; return o; }   // lane-serial build (tests/emu)
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long const o = *p; *p = o + v; return o; }
static inline u32 atomicMax(u32* p, u32 v) { u32 const o = *p; if (v > o) *p = v; return o; }
static inline u32 atomicOr(u32* p, u32 v) { u32 const o = *p; *p = o | v; return o; }
static inline u32 atomicCAS(u32* p, u32 cmp, u32 v) { u32 const o = *p; if (o == cmp) *p = v; return o; }
#endif

// ---- error codes: numerically the reference's ZSTD_ErrorCode (src/main/native/zstd_errors.h:60-98)
enum : u32 {
    ZJ_OK = 0,
    ZJ_E_GENERIC = 1,
    ZJ_E_PREFIX_UNKNOWN = 10,
    ZJ_E_FRAMEPARAM_UNSUPPORTED = 14,
    ZJ_E_WINDOW_TOO_LARGE = 16,
    ZJ_E_CORRUPTION = 20,
    ZJ_E_CHECKSUM_WRONG = 22,
    ZJ_E_LITERALS_HEADER = 24,
    ZJ_E_DICT_CORRUPTED = 30,
    ZJ_E_DICT_WRONG = 32,
    ZJ_E_PARAM_UNSUPPORTED = 40,
    ZJ_E_TABLELOG_TOO_LARGE = 44,
    ZJ_E_DSTSIZE_TOO_SMALL = 70,
    ZJ_E_SRCSIZE_WRONG = 72,
};
#define ZJ_ERR64(code) ((u64)0 - (u64)(code))

// ---- group abstraction ------------------------------------------------------------------
template <int W_>
struct Grp {
    static constexpr int W = W_;
#if ZJ_ON_GPU
    ZJ_DEV u32 lane() const { return threadIdx.x & (W - 1); }
    ZJ_DEV void sync() const { __syncthreads(); }   // 1 wave/WG: s_waitcnt lgkmcnt(0) + s_barrier
#else
    u32 lane() const { return 0; }
    void sync() const {}
#endif
};

// One wave of a workgroup that holds SEVERAL, each at work of its own (zj_encode_pipe_kernel: a parse wave and an entropy wave per frame): the group is the
// wave, so its sync() must not be the workgroup's barrier — the waves run different code and would meet different numbers of them.  A wave's LDS and vector
// memory instructions are issued in order; what a sync has to do is make the compiler keep that order and wait for what is outstanding.
struct GrpWave {
    static constexpr int W = 64;
#if ZJ_ON_GPU
    ZJ_DEV u32 lane() const { return threadIdx.x & 63u; }
    ZJ_DEV void sync() const { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#else
    u32 lane() const { return 0; }
    void sync() const {}
#endif
};

// Values that lane 0 publishes through LDS are the same in every lane, but the compiler cannot know
// that and would build exec-masked ("divergent") control flow around them.  With single-wave
// workgroups hipcc also elides s_barrier, so a divergent loop whose exit depends on such a value can
// leave lanes spinning on LDS while the producing lane is masked off (observed: persistent-loop hang).
// ZJ_UNI() re-reads the value through v_readfirstlane so it lives in an SGPR and every branch on it is
// a scalar branch taken by the whole wave.
#if ZJ_ON_GPU
#define ZJ_UNI(x) ((u32)__builtin_amdgcn_readfirstlane((int)(x)))
ZJ_DEV u64 zj_uni64(u64 v) { return ((u64)ZJ_UNI((u32)(v >> 32)) << 32) | ZJ_UNI((u32)v); }
#else
#define ZJ_UNI(x) ((u32)(x))
ZJ_DEV u64 zj_uni64(u64 v) { return v; }
#endif

// Optional per-phase cycle accounting (kernel argument `prof`, nullptr in production): lane 0 adds the
// shader-clock cycles spent since the previous mark to prof[idx].
struct ZjProf {
#if ZJ_ON_GPU
    unsigned long long* acc; unsigned long long last;
    ZJ_DEV void start(unsigned long long* p) { acc = p; last = p ? __builtin_readcyclecounter() : 0; }
    ZJ_DEV void mark(u32 idx) {
        if (acc) { unsigned long long const t = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&acc[idx], t - last); last = t; }
    }
#else
    void start(unsigned long long*) {}
    void mark(u32) {}
#endif
};

#define GRP_SERIAL(g) if ((g).lane() == 0)
#define GRP_FOR(g, i, n) for (u32 i = (g).lane(); i < (u32)(n); i += (u32)(g).W)

// ---- memory helpers (gfx950 supports unaligned global and LDS dword/qword access; hipcc emits a
//      single global_load_dwordx2 / ds_read_b64 for these memcpy's) -------------------------------
ZJ_HD u32 ld16(const u8* p) { u16 v; __builtin_memcpy(&v, p, 2); return v; }
ZJ_HD u32 ld24(const u8* p) { return ld16(p) | ((u32)p[2] << 16); }
ZJ_HD u32 ld32(const u8* p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
ZJ_HD u64 ld64(const u8* p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }
ZJ_HD void st16(u8* p, u32 v) { u16 w = (u16)v; __builtin_memcpy(p, &w, 2); }
ZJ_HD void st32(u8* p, u32 v) { __builtin_memcpy(p, &v, 4); }
ZJ_HD void st64(u8* p, u64 v) { __builtin_memcpy(p, &v, 8); }

ZJ_HD u32 zj_hibit(u32 v) { return 31u - (u32)__builtin_clz(v); }   // v != 0
ZJ_HD u32 zj_ctz32(u32 v) { return (u32)__builtin_ctz(v); }          // v != 0
ZJ_HD u32 zj_min(u32 a, u32 b) { return a < b ? a : b; }
ZJ_HD u32 zj_max(u32 a, u32 b) { return a > b ? a : b; }

// Compiler-level ordering point for global memory traffic that crosses lanes of the same wave
// (LZ77 execution reads bytes other lanes stored a moment ago).  At workgroup scope on gfx950 this
// emits no instruction: one CU's vector memory operations are performed in order by its L1, so only
// the compiler has to be stopped from reordering.
// One-sided fences between kernels that run beside each other (producer: zj_release, then relaxed atomics; consumer: relaxed poll, then zj_acquire).  On gfx950 an agent-scope
// release is `buffer_wbl2 sc1` — the XCD's L2 written back — and an acquire `buffer_inv sc1` — its L2 invalidated; `__threadfence()` is both, and a release STORE behind it
// writes the L2 back a second time (round 5: the hand-overs paid two write-backs and an invalidate where one write-back is what the protocol needs, profiles/r05/g_).
ZJ_DEV void zj_release() {
#if ZJ_ON_GPU
    // the producers' stores include predicated ones issued through inline asm (zr_st32_m, zj_match_run.h), which the compiler's own wait counting does not see: the wait
    // for them is spelled out instead of left to what the fence happens to emit
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
}
ZJ_DEV void zj_acquire() {
#if ZJ_ON_GPU
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
ZJ_DEV void zj_mem_order() {
#if ZJ_ON_GPU
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}

// wave-wide vote: bit l set iff lane l's predicate holds (lane-serial build: bit 0)
template <class G>
ZJ_DEV u64 grp_ballot(const G& g, bool pred) {
#if ZJ_ON_GPU
    (void)g; return __ballot(pred);
#else
    (void)g; return pred ? 1ull : 0ull;
#endif
}

// lane k's value in every lane (k wave-uniform; lane-serial build: the value itself)
template <class G>
ZJ_DEV u32 grp_bcast(const G& g, u32 v, u32 k) {
#if ZJ_ON_GPU
    (void)g; return (u32)__builtin_amdgcn_readlane((int)v, (int)k);
#else
    (void)g; (void)k; return v;
#endif
}

// inclusive scan of a[0..n) (n <= W, array in the group's LDS), result in place
template <class G>
ZJ_DEV void grp_scan_incl(const G& g, u32* a, u32 n) {
#if ZJ_ON_GPU
    u32 const l = g.lane();
    u32 v = l < n ? a[l] : 0;
#pragma unroll
    for (int d = 1; d < G::W; d <<= 1) {
        u32 t = __shfl_up(v, d, G::W);
        if ((int)l >= d) v += t;
    }
    if (l < n) a[l] = v;
    g.sync();
#else
    (void)g;
    for (u32 i = 1; i < n; i++) a[i] += a[i - 1];
#endif
}

// cooperative copy global->global, 16 B per lane per step, no overlap between src and dst
template <class G>
ZJ_DEV void grp_copy_wide(const G& g, u8* dst, const u8* src, u32 n) {
    u32 const n16 = n >> 4;
    GRP_FOR(g, i, n16) {
        u64 a = ld64(src + 16 * i), b = ld64(src + 16 * i + 8);
        st64(dst + 16 * i, a); st64(dst + 16 * i + 8, b);
    }
    GRP_FOR(g, i, n & 15u) dst[(n16 << 4) + i] = src[(n16 << 4) + i];
}

// ------------------------------------------------------------------ XXH64 (frame checksum) ----
// N/common/xxhash.h (XXH64, seed 0); zstd stores its low 32 bits after the last block (N/compress/zstd_compress.c:
// ZSTD_writeEpilogue, N/decompress/zstd_decompress.c:1050-1060).  The four accumulators are four dependent chains:
// lanes 0..3 each run one over the 32-byte stripes, lane 0 merges and finishes.  Returns a wave-uniform value.
#define ZJ_XXP1 0x9E3779B185EBCA87ULL
#define ZJ_XXP2 0xC2B2AE3D27D4EB4FULL
#define ZJ_XXP3 0x165667B19E3779F9ULL
#define ZJ_XXP4 0x85EBCA77C2B2AE63ULL
#define ZJ_XXP5 0x27D4EB2F165667C5ULL
ZJ_HD u64 zj_rotl64(u64 x, u32 r) { return (x << r) | (x >> (64 - r)); }
ZJ_HD u64 zj_xx_round(u64 acc, u64 in) { return zj_rotl64(acc + in * ZJ_XXP2, 31) * ZJ_XXP1; }
ZJ_HD u64 zj_xx_merge(u64 h, u64 v) { return (h ^ zj_xx_round(0, v)) * ZJ_XXP1 + ZJ_XXP4; }
ZJ_HD u64 zj_xx_finish(u64 h, const u8* p, u32 rem) {          // rem < 32 trailing bytes, then the avalanche
    while (rem >= 8) { h ^= zj_xx_round(0, ld64(p)); h = zj_rotl64(h, 27) * ZJ_XXP1 + ZJ_XXP4; p += 8; rem -= 8; }
    if (rem >= 4) { h ^= (u64)ld32(p) * ZJ_XXP1; h = zj_rotl64(h, 23) * ZJ_XXP2 + ZJ_XXP3; p += 4; rem -= 4; }
    while (rem) { h ^= (u64)(*p) * ZJ_XXP5; h = zj_rotl64(h, 11) * ZJ_XXP1; p++; rem--; }
    h ^= h >> 33; h *= ZJ_XXP2; h ^= h >> 29; h *= ZJ_XXP3; h ^= h >> 32;
    return h;
}
template <class G>
ZJ_DEV u64 zj_xxh64(const G& g, const u8* p, u32 len) {
    u64 h;
    u32 const stripes = len >> 5;
#if ZJ_ON_GPU
    u32 const k = g.lane() & 3u;
    u64 v = k == 0 ? ZJ_XXP1 + ZJ_XXP2 : (k == 1 ? ZJ_XXP2 : (k == 2 ? 0 : 0 - ZJ_XXP1));
    if (g.lane() < 4u) {
        const u8* q = p + 8u * k;
        u32 s = 0;
        for (; s + 4 <= stripes; s += 4) {                 // four loads in flight per round trip
            u64 const a = ld64(q), b = ld64(q + 32), c = ld64(q + 64), d = ld64(q + 96);
            v = zj_xx_round(v, a); v = zj_xx_round(v, b); v = zj_xx_round(v, c); v = zj_xx_round(v, d);
            q += 128;
        }
        for (; s < stripes; s++) { v = zj_xx_round(v, ld64(q)); q += 32; }
    }
    u64 const v1 = __shfl(v, 0, 64), v2 = __shfl(v, 1, 64), v3 = __shfl(v, 2, 64), v4 = __shfl(v, 3, 64);
#else
    u64 v1 = ZJ_XXP1 + ZJ_XXP2, v2 = ZJ_XXP2, v3 = 0, v4 = 0 - ZJ_XXP1;
    for (u32 s = 0; s < stripes; s++) {
        const u8* q = p + 32u * s;
        v1 = zj_xx_round(v1, ld64(q)); v2 = zj_xx_round(v2, ld64(q + 8)); v3 = zj_xx_round(v3, ld64(q + 16)); v4 = zj_xx_round(v4, ld64(q + 24));
    }
    (void)g;
#endif
    if (len >= 32) {
        h = zj_rotl64(v1, 1) + zj_rotl64(v2, 7) + zj_rotl64(v3, 12) + zj_rotl64(v4, 18);
        h = zj_xx_merge(h, v1); h = zj_xx_merge(h, v2); h = zj_xx_merge(h, v3); h = zj_xx_merge(h, v4);
    } else h = ZJ_XXP5;
    h += len;
    h = zj_xx_finish(h, p + 32u * stripes, len & 31u);     // every lane computes the same short tail
    return zj_uni64(h);
}
