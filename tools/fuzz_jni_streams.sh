#!/bin/bash
# Long random-script runs of the JNI library's context streams against the reference's JNI library, on the CPU over the C-ABI double (tests/jni/emu_abi.cpp):
#   tools/fuzz_jni_streams.sh [first-seed] [seeds] [scripts-per-seed]     (default 100, 4, 600; even seeds: GPU route only, odd seeds: bundled library behind)
# Each script (context streams, then the heap-array stream classes): one to three frames on one context / stream object, writes of random sizes, flushes, single-directive frames, pledged sizes, any heap / direct combination, a
# target of a few bytes or a roomy one; the bytes handed out must equal the reference's, and the frames must decode through both decompress natives.
cd "$(dirname "$0")/.." || exit 1
make -s -C tests/jni emu || exit 1
python - <<'PY' > /tmp/fuzz_jni_dict.txt
import sys; sys.path.insert(0, 'tests')
import test_jni_shim as t; print(t._dict_file())
PY
D=$(cat /tmp/fuzz_jni_dict.txt); REF=$PWD/oracle/_ref/libzstd-jni-ref.so; SHIM=$PWD/tests/jni/_build/emu/libzstd-jni-amd.so
first=${1:-100}; seeds=${2:-4}; per=${3:-600}; bad=0
for ((s = first; s < first + seeds; s++)); do
  if ((s % 2 == 0)); then
    out=$(env -u ZSTD_JNI_CPU_LIB HARNESS_FUZZ=$s,$per HARNESS_SKIP_BATCH=1 HARNESS_DICT_FILE=$D HARNESS_PLAIN_MAX_LEVEL=1 HARNESS_MAX_LEVEL=1 HARNESS_STREAM_MAX=0 tests/jni/_build/harness $REF $SHIM 2>&1 | grep "MISMATCH\|JNI-HARNESS FUZZ\|JNI-HARNESS OK\|FAILED")
  else
    out=$(ZSTD_JNI_CPU_LIB=$REF ZSTD_JNI_GPU_STREAMS=1 HARNESS_FUZZ=$s,$per HARNESS_SKIP_BATCH=1 HARNESS_MAX_LEVEL=1 HARNESS_DICT_FILE=$D tests/jni/_build/harness $REF $SHIM 2>&1 | grep "MISMATCH\|JNI-HARNESS FUZZ\|JNI-HARNESS OK\|FAILED")
  fi
  echo "$out" | cut -c1-240
  echo "$out" | grep -q "JNI-HARNESS OK" || bad=1
done
[ $bad = 0 ] && echo "JNI-STREAM-FUZZ OK" || echo "JNI-STREAM-FUZZ FAILED"
