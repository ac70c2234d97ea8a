// Standalone timing harness for the wave-per-frame matcher (zj_match_wave.h): G single-wave workgroups run it over F frames
// of one synthetic class; with -DZW_PROFILE workgroup 0 reports the cycles of every phase of a pass.
//   hipcc -O3 --offload-arch=gfx950 [-DZW_PROFILE] -o wavebench wavebench.hip ;  ./wavebench [frames] [grid]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../zstd-jni_amd/csrc/zj_encode.h"
#include "../../zstd-jni_amd/csrc/zj_match_wave.h"
#include "../../zstd-jni_amd/csrc/zj_synth.h"

extern __shared__ __attribute__((aligned(16))) u8 dyn_lds[];
__global__ void fill(u8* dst, u32 size, u32 cls, u32 n) {
    u32 const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) zs_fill(dst + (size_t)i * size, size, (u64)cls + 4ull * (i + 1));
}
__global__ __launch_bounds__(64) void bench(const u8* src, u32 size, u32 n, u8* fscratch, u32* meta, unsigned long long* prof) {
    ZWLds& lds = *(ZWLds*)dyn_lds;
    for (u32 k = blockIdx.x; k < n; k += gridDim.x) {
        ZWaveD m;
        u32 const lastLL = m.run(lds, src + (size_t)k * size, size, ze_params_of(3, size), fscratch + (size_t)blockIdx.x * ZE_FRAME_STRIDE(65536u), 65536u);
        if (threadIdx.x == 0) { meta[3 * k] = m.o.n; meta[3 * k + 1] = m.o.lit + lastLL; meta[3 * k + 2] = lastLL; }
#ifdef ZW_PROFILE
        if (blockIdx.x == 0 && threadIdx.x == 0) for (int j = 0; j < 16; j++) prof[j] += m.pf[j];
#endif
        __syncthreads();
    }
}
int main(int argc, char** argv) {
    u32 const frames = argc > 1 ? (u32)atoi(argv[1]) : 2048, grid = argc > 2 ? (u32)atoi(argv[2]) : 256, size = 65536;
    u8* src; u8* fs; u32* meta; unsigned long long* prof;
    hipMalloc(&src, (size_t)frames * size); hipMalloc(&fs, (size_t)grid * ZE_FRAME_STRIDE(65536u)); hipMalloc(&meta, frames * 12); hipMalloc(&prof, 128);
    hipFuncSetAttribute((const void*)bench, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ZWLds));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    u32* hm = (u32*)malloc(frames * 12);
    printf("ZWLds %zu bytes, %u frames of 64 KiB, %u workgroups\n", sizeof(ZWLds), frames, grid);
    for (u32 cls = 0; cls < 4; cls++) {
        fill<<<(frames + 255) / 256, 256>>>(src, size, cls, frames);
        hipMemset(prof, 0, 128);
        bench<<<grid, 64, sizeof(ZWLds)>>>(src, size, frames < grid ? frames : grid, fs, meta, prof);   // warm-up
        hipMemset(prof, 0, 128);
        hipEventRecord(a); bench<<<grid, 64, sizeof(ZWLds)>>>(src, size, frames, fs, meta, prof); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(hm, meta, frames * 12, hipMemcpyDeviceToHost);
        unsigned long long seqs = 0, lits = 0; for (u32 i = 0; i < frames; i++) { seqs += hm[3 * i]; lits += hm[3 * i + 1]; }
        printf("class %u: %8.2f ms  (%.2f us/frame amortised, %.1f seqs/frame, %.0f literals/frame)\n", cls, ms, ms * 1e3 / frames, (double)seqs / frames, (double)lits / frames);
#ifdef ZW_PROFILE
        unsigned long long hp[16]; hipMemcpy(hp, prof, 128, hipMemcpyDeviceToHost);
        printf("   wg0 cycles: total %llu | passes %llu hits %llu slow-path %llu repIters %llu | window+insert %llu candidates %llu undo/resolve %llu extend %llu store %llu post+rep %llu stage %llu\n",
               hp[15], hp[12], hp[13], hp[11], hp[14], hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[9]);
#endif
    }
    return 0;
}
