# entropy-stage phases of the multi-block kernel on xml slices + timing of the product library   -> gpurun_out/call8.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
cat > $OUT/ab9.txt <<X
wave ZJNI_MULTI_WAVE=1
X
echo "== 1024 x 1 MiB synthetic"; STEPS=2 bash tools/ab.sh $OUT/ab9.txt 1024 1048576 3
echo "== 4096 x 1 MiB xml"; PROF_DATA=xml STEPS=2 bash tools/ab.sh $OUT/ab9.txt 4096 1048576 3
PROF_DATA=xml ZJNI_LIB=$R/zstd-jni_amd/lib/libzjni_amd_zxprof.so AB_TAG=zxprof timeout 120 python tools/prof_driver.py 1024 1048576 3 1 2>&1 | grep "^zx phases\|^zx frame wg 0" | cut -c1-330 | head -6
} > $OUT/call8.txt 2>&1
cat $OUT/call8.txt
