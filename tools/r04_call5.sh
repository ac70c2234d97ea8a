# round 4, call 5: lazy refill of the stream windows (one request per 8 bytes of progress instead of one per round)   -> gpurun_out/r04_call5.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
L=$R/zstd-jni_amd/lib
{
cat > $OUT/ab5.txt <<X
r3base ZJNI_LIB=$L/libzjni_amd_r3base.so
flatD ZJNI_LIB=$L/libzjni_amd_flatD.so
lazyC ZJNI_LIB=$L/libzjni_amd_lazyC.so
lazyD ZJNI_LIB=$L/libzjni_amd_lazyD.so
r3base2 ZJNI_LIB=$L/libzjni_amd_r3base.so
lazyC2 ZJNI_LIB=$L/libzjni_amd_lazyC.so
lazyD2 ZJNI_LIB=$L/libzjni_amd_lazyD.so
X
echo "== metric 65536 x 64 KiB L3"; STEPS=3 bash tools/ab.sh $OUT/ab5.txt
cat > $OUT/ab5b.txt <<X
r3base ZJNI_LIB=$L/libzjni_amd_r3base.so
lazyC ZJNI_LIB=$L/libzjni_amd_lazyC.so
lazyD ZJNI_LIB=$L/libzjni_amd_lazyD.so
X
echo "== 65536 x 128 KiB L3"; STEPS=2 bash tools/ab.sh $OUT/ab5b.txt 65536 131072 3
echo "== 16384 x 64 KiB L3, flags for every frame / none"
cat > $OUT/ab5c.txt <<X
lazyD_need1 ZJNI_LIB=$L/libzjni_amd_lazyD.so ZJNI_NEED=1
lazyD_need0 ZJNI_LIB=$L/libzjni_amd_lazyD.so ZJNI_NEED=0
X
STEPS=1 bash tools/ab.sh $OUT/ab5c.txt 16384 65536 3
} > $OUT/r04_call5.txt 2>&1
cat $OUT/r04_call5.txt
