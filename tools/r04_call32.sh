#!/bin/bash
# JNI legs on the GPU with the context streams: (a) no bundled library, (b) bundled library + ZSTD_JNI_GPU_STREAMS=1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
D=$(python - <<'PY'
import sys; sys.path.insert(0, 'tests')
import test_jni_shim as t; print(t._dict_file())
PY
)
REF=$PWD/oracle/_ref/libzstd-jni-ref.so; SHIM=$PWD/zstd-jni_amd/lib/libzstd-jni-amd.so; H=tests/jni/_build/harness
{
echo "== leg a: GPU only (no bundled library)"
env -u ZSTD_JNI_CPU_LIB HARNESS_VERBOSE=1 HARNESS_DICT_FILE=$D HARNESS_PLAIN_MAX_LEVEL=8 HARNESS_EXPECT=gpu HARNESS_STREAM_MAX=0 HARNESS_FUZZ=12,60 timeout 500 $H $REF $SHIM 2>&1 | tail -40
echo "== leg b: bundled library behind, ZSTD_JNI_GPU_STREAMS=1"
ZSTD_JNI_CPU_LIB=$REF ZSTD_JNI_GPU_STREAMS=1 HARNESS_VERBOSE=1 HARNESS_SKIP_BATCH=1 HARNESS_MAX_LEVEL=2 HARNESS_FUZZ=13,60 HARNESS_DICT_FILE=$D timeout 500 $H $REF $SHIM 2>&1 | tail -40
} > gpurun_out/r04_jni_gpu_legs.txt 2>&1
cat gpurun_out/r04_jni_gpu_legs.txt
