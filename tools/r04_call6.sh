# round 4, call 6: the same variants with the flags in place from the first round (ZJNI_NEED_INLINE=1: deterministic), 5 steps, alternating; HBM-side traffic of r3base / lazyD
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
L=$R/zstd-jni_amd/lib
{
cat > $OUT/ab6.txt <<X
r3base ZJNI_LIB=$L/libzjni_amd_r3base.so ZJNI_NEED_INLINE=1
flatD ZJNI_LIB=$L/libzjni_amd_flatD.so ZJNI_NEED_INLINE=1
lazyD ZJNI_LIB=$L/libzjni_amd_lazyD.so ZJNI_NEED_INLINE=1
lazyC ZJNI_LIB=$L/libzjni_amd_lazyC.so ZJNI_NEED_INLINE=1
r3base2 ZJNI_LIB=$L/libzjni_amd_r3base.so ZJNI_NEED_INLINE=1
flatD2 ZJNI_LIB=$L/libzjni_amd_flatD.so ZJNI_NEED_INLINE=1
lazyD2 ZJNI_LIB=$L/libzjni_amd_lazyD.so ZJNI_NEED_INLINE=1
lazyC2 ZJNI_LIB=$L/libzjni_amd_lazyC.so ZJNI_NEED_INLINE=1
r3base3 ZJNI_LIB=$L/libzjni_amd_r3base.so ZJNI_NEED=0
lazyD3 ZJNI_LIB=$L/libzjni_amd_lazyD.so ZJNI_NEED=0
X
echo "== metric 65536 x 64 KiB L3, flags first"; STEPS=5 bash tools/ab.sh $OUT/ab6.txt
echo "== traffic r3base"; bash tools/pmc_one.sh r3base ZJNI_LIB=$L/libzjni_amd_r3base.so ZJNI_NEED_INLINE=1 | grep match_run
echo "== traffic lazyD"; bash tools/pmc_one.sh lazyD ZJNI_LIB=$L/libzjni_amd_lazyD.so ZJNI_NEED_INLINE=1 | grep match_run
} > $OUT/r04_call6.txt 2>&1
cat $OUT/r04_call6.txt
