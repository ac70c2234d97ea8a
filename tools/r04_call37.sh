#!/bin/bash
# quick pass of the JNI harness' GPU-only leg on the GPU (long streams left out: HARNESS_TOTAL_MAX), with both kinds of random scripts
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
D=$PWD/tests/jni/_build/trained.dict
env -u ZSTD_JNI_CPU_LIB HARNESS_TOTAL_MAX=300000 HARNESS_FUZZ=41,40 HARNESS_DICT_FILE=$D HARNESS_PLAIN_MAX_LEVEL=3 HARNESS_MAX_LEVEL=3 HARNESS_EXPECT=gpu HARNESS_STREAM_MAX=0 timeout 75 tests/jni/_build/harness $PWD/oracle/_ref/libzstd-jni-ref.so $PWD/zstd-jni_amd/lib/libzstd-jni-amd.so 2>&1 | grep -v INSPECTION | tail -8 > gpurun_out/r04_jni_gpu_quick.txt
cat gpurun_out/r04_jni_gpu_quick.txt
