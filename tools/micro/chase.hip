// Microbenchmark: dependent random 4-byte read(+write) chains, one chain per lane, in different layouts.
// Answers "what does one memory round trip of the lane-per-frame match finder cost, and why".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;
// mode 0: lane-private region (stride regionBytes)   mode 1: entry-major interleave across the 64 lanes of a wave
// mode 2: all lanes share region 0                    activeLanes: lanes >= activeLanes idle
__global__ void chase(u32* mem, u32 entries, u32 steps, int mode, int activeLanes, int doWrite, u32* sink) {
    u32 const lane = threadIdx.x & 63, wave = blockIdx.x;
    if ((int)lane >= activeLanes) return;
    u64 const chain = (u64)wave * 64 + lane;
    u32 x = (u32)(chain * 2654435761u) | 1u, acc = 0;
    for (u32 s = 0; s < steps; s++) {
        u32 const e = (x * 2654435761u) >> (32 - 15);            // 15-bit entry index (< entries = 24576? clamp below)
        u32 const ee = e % entries;
        u64 idx;
        if (mode == 0) idx = chain * entries + ee;
        else if (mode == 1) idx = ((u64)wave * entries + ee) * 64 + lane;
        else idx = ee;
        u32 const v = mem[idx];
        if (doWrite) mem[idx] = v + 1;
        acc += v; x = x * 1664525u + 1013904223u + v;             // next address depends on the loaded value
    }
    sink[chain] = acc;
}
int main(int argc, char** argv) {
    u32 const entries = 24576;                                     // 96 KiB of u32 per chain (L3 tables)
    u32 const maxWaves = 1024; u32 const steps = 2000;
    u32* mem; u32* sink;
    size_t const bytes = (size_t)maxWaves * 64 * entries * 4;
    hipMalloc(&mem, bytes); hipMemset(mem, 0, bytes); hipMalloc(&sink, maxWaves * 64 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    if (argc > 1 && argv[1][0] == 's') {   // "sweep": request rate against chains in flight (1, 2, 4 waves per SIMD), private 96 KiB regions
        u32 const big = 4096; u32* m2; u32* s2;
        size_t const b2 = (size_t)big * 64 * entries * 4;
        if (hipMalloc(&m2, b2) != hipSuccess) { printf("no memory for the sweep\n"); return 1; }
        hipMemset(m2, 0, b2); hipMalloc(&s2, big * 64 * 4);
        for (int wr = 0; wr < 2; wr++) for (u32 w : {512u, 1024u, 2048u, 4096u}) {
            chase<<<w, 64>>>(m2, entries, 10, 0, 64, wr, s2);
            hipEventRecord(a); chase<<<w, 64>>>(m2, entries, steps, 0, 64, wr, s2); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double const acc = (double)w * 64 * steps * (wr ? 2 : 1);
            printf("sweep write %d waves %4u: %7.1f ns/step, %6.1f G requests/s\n", wr, w, ms * 1e6 / steps, acc / (ms * 1e6));
        }
        return 0;
    }
    if (argc > 1) {      // calibration run for rocprofv3 --pmc: one known access count per launch (1024 waves x 64 lanes x steps)
        for (int wr = 0; wr < 2; wr++) { chase<<<1024, 64>>>(mem, entries, steps, 0, 64, wr, sink); hipDeviceSynchronize(); }
        printf("calibration: %llu random 4-byte accesses per launch (launch 1 read-only, launch 2 read+write)\n", (unsigned long long)1024 * 64 * steps);
        return 0;
    }
    int const waveCounts[] = {64, 256, 1024};
    for (int mode = 0; mode < 3; mode++) for (int wr = 0; wr < 2; wr++) for (int al : {1, 16, 64}) for (int w : waveCounts) {
        chase<<<w, 64>>>(mem, entries, 10, mode, al, wr, sink);
        hipEventRecord(a); chase<<<w, 64>>>(mem, entries, steps, mode, al, wr, sink); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("mode %d write %d lanes %2d waves %4d: %7.1f ns/step\n", mode, wr, al, w, ms * 1e6 / steps);
    }
    return 0;
}
