// Microbenchmarks behind DESIGN.md section 4 (round 2): where does the random-request ceiling of the lane-per-frame
// match finder come from, and what does one step of a wave-per-frame match finder (tables in LDS) cost?
//   footprint : 65 536 dependent chains of random 4-byte reads (and read+write) over 4 MiB .. 6 GiB in total
//   latency   : one wave, dependent loads: LDS, global at L2 / Infinity-Cache / HBM footprints; dependent ALU issue
//   wavestep  : W single-wave workgroups, each step = LDS probe -> 64 scattered 8-byte reads inside the wave's own
//               64 KiB "frame" (addresses depend on the previous step) -> ballot; the pass cost of the wave kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
typedef uint32_t u32; typedef uint64_t u64;

__global__ void chase_fp(u32* mem, u64 totalEntries, u32 steps, int doWrite, u32* sink) {
    u64 const chain = (u64)blockIdx.x * 64 + (threadIdx.x & 63);
    u64 const chains = (u64)gridDim.x * 64;
    u64 const per = totalEntries / chains;                     // lane-private region, like the per-frame tables
    u32* const base = mem + chain * per;
    u32 x = (u32)(chain * 2654435761u) | 1u, acc = 0;
    for (u32 s = 0; s < steps; s++) {
        u32 const e = (u32)(((u64)(x * 2654435761u) * per) >> 32);
        u32 const v = base[e];
        if (doWrite) base[e] = v + 1;
        acc += v; x = x * 1664525u + 1013904223u + v;
    }
    sink[chain] = acc;
}

__global__ void lat_global(const u32* mem, u32 entriesMask, u32 steps, u32* sink, u64* cyc) {
    u32 idx = threadIdx.x * 977u;
    u64 const t0 = __builtin_readcyclecounter();
    for (u32 s = 0; s < steps; s++) idx = mem[(idx * 2654435761u + threadIdx.x) & entriesMask];
    u64 const t1 = __builtin_readcyclecounter();
    sink[threadIdx.x] = idx;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
__global__ void lat_lds(u32 steps, u32* sink, u64* cyc) {
    __shared__ u32 t[8192];
    for (u32 i = threadIdx.x; i < 8192; i += 64) t[i] = (i * 2654435761u) >> 19;
    __syncthreads();
    u32 idx = threadIdx.x;
    u64 const t0 = __builtin_readcyclecounter();
    for (u32 s = 0; s < steps; s++) idx = t[(idx + threadIdx.x) & 8191u];
    u64 const t1 = __builtin_readcyclecounter();
    sink[threadIdx.x] = idx;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
__global__ void lat_alu(u32 steps, u32* sink, u64* cyc) {
    u32 x = threadIdx.x | 1u;
    u64 const t0 = __builtin_readcyclecounter();
    for (u32 s = 0; s < steps; s++) { x = x * 2654435761u + 12345u; x ^= x >> 7; x = x * 40503u + 1u; x ^= x >> 11; }   // 8 dependent VALU ops
    u64 const t1 = __builtin_readcyclecounter();
    sink[threadIdx.x] = x;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

// one step of a wave-per-frame match finder: table probe in LDS, candidate bytes from the frame, vote
__global__ __launch_bounds__(64) void wavestep(const uint8_t* frames, u32 steps, u32 ldsEntries, u32* sink, u64* cyc) {
    extern __shared__ unsigned short tab[];
    for (u32 i = threadIdx.x; i < ldsEntries; i += 64) tab[i] = (unsigned short)((i * 2654435761u) >> 16);
    __syncthreads();
    const uint8_t* const f = (const uint8_t*)frames + (size_t)blockIdx.x * 65536u;
    u32 pos = threadIdx.x; u32 acc = 0;
    u64 const t0 = __builtin_readcyclecounter();
    for (u32 s = 0; s < steps; s++) {
        u32 const h = ((pos + threadIdx.x) * 2654435761u) >> 8;
        u32 const c0 = tab[h % ldsEntries], c1 = tab[(h >> 3) % ldsEntries];
        u64 a, b;
        memcpy(&a, f + (c0 & 0xFFF8u), 8); memcpy(&b, f + (c1 & 0xFFF8u), 8);
        bool const hit = ((u32)a ^ (u32)(b >> 7)) & 1u;
        u64 const m = __ballot(hit);
        u32 const w = m ? (u32)__builtin_ctzll(m) : 0u;
        pos = (pos + w + 1u + (u32)(a >> 60)) & 0xFFFFu; acc += (u32)b;
        tab[h % ldsEntries] = (unsigned short)pos;
    }
    u64 const t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 64 + threadIdx.x] = acc + pos;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    u32* sink; hipMalloc(&sink, 4096 * 64 * 4);
    u64* cyc; hipMalloc(&cyc, 4096 * 8);
    if (!strcmp(what, "footprint") || !strcmp(what, "all")) {
        size_t const maxBytes = (size_t)6 << 30;
        u32* mem; if (hipMalloc(&mem, maxBytes) != hipSuccess) { printf("no memory\n"); return 1; }
        hipMemset(mem, 0, maxBytes);
        size_t const fps[] = {(size_t)4 << 20, (size_t)32 << 20, (size_t)192 << 20, (size_t)1 << 30, (size_t)6 << 30};
        for (u32 waves : {1024u, 2048u}) for (int wr = 0; wr < 2; wr++) for (size_t fp : fps) {
            u32 const steps = 2000;
            chase_fp<<<waves, 64>>>(mem, fp / 4, 50, wr, sink);
            hipEventRecord(a); chase_fp<<<waves, 64>>>(mem, fp / 4, steps, wr, sink); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double const req = (double)waves * 64 * steps * (wr ? 2 : 1);
            printf("footprint %6zu MiB  waves %4u  write %d : %7.1f ns/step  %6.1f G requests/s\n", fp >> 20, waves, wr, ms * 1e6 / steps, req / (ms * 1e6));
        }
        hipFree(mem);
    }
    if (!strcmp(what, "latency") || !strcmp(what, "all")) {
        u32 const steps = 20000; u64 h;
        size_t const fps[] = {(size_t)256 << 10, (size_t)2 << 20, (size_t)64 << 20, (size_t)2 << 30};
        u32* mem; hipMalloc(&mem, (size_t)2 << 30);
        {   // fill with pseudo-random indices
            size_t const n = ((size_t)2 << 30) / 4; u32* hbuf = (u32*)malloc(n * 4);
            u32 x = 12345; for (size_t i = 0; i < n; i++) { x = x * 1664525u + 1013904223u; hbuf[i] = x; }
            hipMemcpy(mem, hbuf, n * 4, hipMemcpyHostToDevice); free(hbuf);
        }
        for (size_t fp : fps) {
            lat_global<<<1, 64>>>(mem, (u32)(fp / 4 - 1), 200, sink, cyc);
            lat_global<<<1, 64>>>(mem, (u32)(fp / 4 - 1), steps, sink, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("latency global scattered-64, footprint %7zu KiB : %6.1f cycles/step\n", fp >> 10, (double)h / steps);
        }
        lat_lds<<<1, 64>>>(steps, sink, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("latency LDS dependent read          : %6.1f cycles/step\n", (double)h / steps);
        lat_alu<<<1, 64>>>(steps, sink, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("latency 8 dependent VALU ops        : %6.1f cycles/step (%.1f per op)\n", (double)h / steps, (double)h / steps / 8);
        hipFree(mem);
    }
    if (!strcmp(what, "wavestep") || !strcmp(what, "all")) {
        uint8_t* frames; hipMalloc(&frames, (size_t)4096 * 65536); hipMemset(frames, 0x5A, (size_t)4096 * 65536);
        u64* hc = (u64*)malloc(4096 * 8);
        hipFuncSetAttribute((const void*)wavestep, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        for (u32 ldsKiB : {48u, 36u}) for (u32 waves : {256u, 512u, 768u, 1024u, 2048u}) {
            u32 const steps = 4000;
            wavestep<<<waves, 64, ldsKiB * 1024>>>(frames, 50, ldsKiB * 512, sink, cyc);
            hipEventRecord(a); wavestep<<<waves, 64, ldsKiB * 1024>>>(frames, steps, ldsKiB * 512, sink, cyc); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(hc, cyc, waves * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (u32 i = 0; i < waves; i++) avg += (double)hc[i]; avg /= waves;
            printf("wavestep lds %2u KiB  waves %4u : kernel %7.2f ms, %7.1f ns/step wall, %7.1f cycles/step per wave, %6.2f G steps/s\n",
                   ldsKiB, waves, ms, ms * 1e6 / steps, avg / steps, (double)waves * steps / (ms * 1e6));
        }
    }
    return 0;
}
