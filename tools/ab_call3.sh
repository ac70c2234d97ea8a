# Round 3, multi-block wave matcher: parity on the GPU, then product library (tags + staged spans + unserialised loads) against the
# variant without tags, and the phase profile of the product's flow.   gpurun -- 'bash tools/ab_call3.sh'   -> gpurun_out/call3.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
echo "== gpu multiblock tests"; timeout 300 python -m pytest tests/test_gpu_multiblock.py -x -q 2>&1 | tail -3
cat > $OUT/ab4.txt <<X
serial ZJNI_MULTI_WAVE=0
notags ZJNI_LIB=$R/zstd-jni_amd/lib/libzjni_amd_notags.so
tags ZJNI_MULTI_WAVE=1
tags_nocarry ZJNI_MULTI_WAVE=2
X
echo "== 1024 x 1 MiB level 3"; STEPS=2 bash tools/ab.sh $OUT/ab4.txt 1024 1048576 3
cat > $OUT/ab5.txt <<X
tags_2048_p8 ZJNI_MULTI_WAVE=1 ZJNI_MULTI_PER_CU=8
X
echo "== 2048 x 1 MiB level 3"; STEPS=2 bash tools/ab.sh $OUT/ab5.txt 2048 1048576 3
echo "== phase profile"
ZJNI_LIB=$R/zstd-jni_amd/lib/libzjni_amd_zxprof.so AB_TAG=zxprof timeout 120 python tools/prof_driver.py 1024 1048576 3 1 2>&1 | grep "^zx wg" | sort | awk 'NR % 8 == 1' | head -16
} > $OUT/call3.txt 2>&1
cat $OUT/call3.txt
