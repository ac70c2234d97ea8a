# round 4, call 2: the flat round (zj_match_run.h rewritten: predicated single-instruction accesses, selects instead of per-state regions) against round 3's,
# as one body (flatA) and as four rotation-slot copies (flatB); SQ counters of flatA     -> gpurun_out/r04_call2.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
L=$R/zstd-jni_amd/lib
{
cat > $OUT/ab2.txt <<X
r3base ZJNI_LIB=$L/libzjni_amd_r3base.so
flatA ZJNI_LIB=$L/libzjni_amd_flatA.so
flatB ZJNI_LIB=$L/libzjni_amd_flatB.so
r3base2 ZJNI_LIB=$L/libzjni_amd_r3base.so
flatA2 ZJNI_LIB=$L/libzjni_amd_flatA.so
X
echo "== metric 65536 x 64 KiB L3"; STEPS=3 bash tools/ab.sh $OUT/ab2.txt
echo "== flags for every frame (ZJNI_NEED=1), 16384 frames: exactness of the flagged paths"; 
cat > $OUT/ab2b.txt <<X
r3base_need1 ZJNI_LIB=$L/libzjni_amd_r3base.so ZJNI_NEED=1
flatA_need1 ZJNI_LIB=$L/libzjni_amd_flatA.so ZJNI_NEED=1
flatA_need0 ZJNI_LIB=$L/libzjni_amd_flatA.so ZJNI_NEED=0
X
STEPS=1 bash tools/ab.sh $OUT/ab2b.txt 16384 65536 3
echo "== mixed sizes through the test suite's encode tests on flatA"
cp $L/libzjni_amd.so $OUT/lib_backup.so; cp $L/libzjni_amd_flatA.so $L/libzjni_amd.so
timeout 900 python -m pytest tests/test_gpu_encode.py -x -q -m gpu 2>&1 | tail -4
cp $OUT/lib_backup.so $L/libzjni_amd.so; rm -f $OUT/lib_backup.so
echo "== SQ counters flatA"; bash tools/sq_counters.sh flatA ZJNI_LIB=$L/libzjni_amd_flatA.so
} > $OUT/r04_call2.txt 2>&1
cat $OUT/r04_call2.txt
