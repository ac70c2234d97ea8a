# product library on 1 MiB frames (synthetic classes and xml slices) + the window-level profile on xml   -> gpurun_out/call6.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
cat > $OUT/ab7.txt <<X
wave ZJNI_MULTI_WAVE=1
X
echo "== 1024 x 1 MiB synthetic"; STEPS=2 bash tools/ab.sh $OUT/ab7.txt 1024 1048576 3
echo "== 1024 x 1 MiB xml"; PROF_DATA=xml STEPS=2 bash tools/ab.sh $OUT/ab7.txt 1024 1048576 3
echo "== 4096 x 1 MiB xml"; PROF_DATA=xml STEPS=2 bash tools/ab.sh $OUT/ab7.txt 4096 1048576 3
PROF_DATA=xml ZJNI_LIB=$R/zstd-jni_amd/lib/libzjni_amd_zxprof.so AB_TAG=zxprof timeout 120 python tools/prof_driver.py 1024 1048576 3 1 2>&1 | grep "^zx " | sort | awk 'NR % 9 == 1' | cut -c1-330 | head -12
} > $OUT/call6.txt 2>&1
cat $OUT/call6.txt
