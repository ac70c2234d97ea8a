# round 4, call 3: slopes of the level-3 match kernel's time — over the request count (1 / 3 extra stream-like loads per search round) and over the
# instruction count (200 extra dependent vector instructions per round)     -> gpurun_out/r04_call3.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
L=$R/zstd-jni_amd/lib
{
cat > $OUT/ab3.txt <<X
flatB ZJNI_LIB=$L/libzjni_amd_flatB.so
x1 ZJNI_LIB=$L/libzjni_amd_x1.so
x3 ZJNI_LIB=$L/libzjni_amd_x3.so
v200 ZJNI_LIB=$L/libzjni_amd_v200.so
flatB2 ZJNI_LIB=$L/libzjni_amd_flatB.so
X
echo "== metric 65536 x 64 KiB L3"; STEPS=2 bash tools/ab.sh $OUT/ab3.txt
echo "== 16384 x 64 KiB L3 (a quarter of the lanes)"; STEPS=2 bash tools/ab.sh $OUT/ab3.txt 16384 65536 3
} > $OUT/r04_call3.txt 2>&1
cat $OUT/r04_call3.txt
