#!/usr/bin/env python3
"""Cut a small translation unit holding only zj_enc_match_run_kernel out of zj_kernels.hip (for quick ISA census builds: ~15 s instead of ~60 s).
usage: make_run_tu.py out.hip"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "zstd-jni_amd", "csrc", "zj_kernels.hip")).read()
def cut(a, b):
    i = src.index(a); j = src.index(b, i); return src[i:j]
csrc = os.path.join(ROOT, "zstd-jni_amd", "csrc")
out = ['#include <hip/hip_runtime.h>', '#include <string.h>', '#include <stdio.h>', '#include <stdlib.h>',
       '#include "%s/../../include/zjni_amd.h"' % csrc, '#include "%s/zj_decode.h"' % csrc, '#include "%s/zj_encode.h"' % csrc]
out.append(cut("__device__ __forceinline__ void zj_publish_done", "__global__ __launch_bounds__(64) void zj_dec_seq_kernel("))
out.append(cut("__device__ __forceinline__ bool zj_claim_front", "__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void zj_enc_match_kernel("))
out.append(cut("template <u32 JMAX>\n__device__ __forceinline__ void zj_enc_match_run_body", "#ifdef ZJ_TUNING_KERNELS"))
open(sys.argv[1], "w").write("\n".join(out) + "\n")
