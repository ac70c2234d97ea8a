// zj_simt.h — explicit-SIMT notation shared by the wave-per-frame matchers (zj_match_wave.h: tables in LDS; zj_match_wavex.h: blocks of
// multi-block frames over HBM tables).  Per-lane variables are ZWV<T>, per-lane code sits in ZW_LANES blocks, lanes talk to each other only
// BETWEEN blocks (ballot / shuffle / LDS), so the same source compiles for the GPU (one thread per lane) and, with a 64-iteration loop per
// block, under g++ for the CPU-side parity tests (tests/emu; test infrastructure only).
#pragma once

#if ZJ_ON_GPU
template <class T> struct ZWV { T v; ZJ_DEV_MEMBER T& operator[](u32) { return v; } ZJ_DEV_MEMBER const T& operator[](u32) const { return v; } };
#define ZW_LANES(l) for (u32 l = threadIdx.x & 63u, zw_once_ = 1; zw_once_; zw_once_ = 0)
// Between two lane blocks that talk through LDS.  One wave per workgroup and the LDS executes a wave's instructions in
// order, so only the COMPILER has to keep the order; __syncthreads() would also wait for every outstanding global store
// (the sequence records) — a full memory round trip per sequence.
#define ZW_SYNC() __asm__ volatile("" ::: "memory")
ZJ_DEV u64 zw_ballot(const ZWV<bool>& b) { return __ballot(b.v); }
ZJ_DEV u32 zw_get(const ZWV<u32>& x, u32 k) { return (u32)__builtin_amdgcn_readlane((int)x.v, (int)k); }
ZJ_DEV u64 zw_get64(const ZWV<u64>& x, u32 k) {
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(x.v >> 32), (int)k) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)x.v, (int)k);
}
ZJ_DEV bool zw_getb(const ZWV<bool>& x, u32 k) { return (__ballot(x.v) >> k) & 1ull; }
ZJ_DEV void zw_shfl(ZWV<u32>& out, const ZWV<u32>& x, const ZWV<u32>& idx) { out.v = (u32)__shfl((int)x.v, (int)idx.v, 64); }
ZJ_DEV void zw_or64(u64* p, u64 v) { atomicOr((unsigned long long*)p, (unsigned long long)v); }
#else
template <class T> struct ZWV { T v[64]; T& operator[](u32 l) { return v[l]; } const T& operator[](u32 l) const { return v[l]; } };
#ifdef ZW_EMU_REVERSE
#define ZW_LANES(l) for (u32 l = 63u; l < 64u; l--)
#else
#define ZW_LANES(l) for (u32 l = 0; l < 64u; l++)
#endif
#define ZW_SYNC() ((void)0)
ZJ_DEV u64 zw_ballot(const ZWV<bool>& b) { u64 m = 0; for (u32 l = 0; l < 64u; l++) if (b.v[l]) m |= 1ull << l; return m; }
ZJ_DEV u32 zw_get(const ZWV<u32>& x, u32 k) { return x.v[k]; }
ZJ_DEV u64 zw_get64(const ZWV<u64>& x, u32 k) { return x.v[k]; }
ZJ_DEV bool zw_getb(const ZWV<bool>& x, u32 k) { return x.v[k]; }
ZJ_DEV void zw_shfl(ZWV<u32>& out, const ZWV<u32>& x, const ZWV<u32>& idx) { ZWV<u32> t; for (u32 l = 0; l < 64u; l++) t.v[l] = x.v[idx.v[l] & 63u]; out = t; }
ZJ_DEV void zw_or64(u64* p, u64 v) { *p |= v; }
#endif

// All loads of a step are consumed at one program point: without it the compiler sinks each load next to its use, behind a
// wait of its own, and a step costs one LDS round trip per load instead of one in total.
#if ZJ_ON_GPU
#define ZW_FENCE2(a, b) asm volatile("" :: "v"(a), "v"(b))
#define ZW_FENCE4(a, b, c, d) asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d))
#else
#define ZW_FENCE2(a, b) ((void)0)
#define ZW_FENCE4(a, b, c, d) ((void)0)
#endif
