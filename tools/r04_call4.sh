# round 4, call 4: flat round with 0 / 1 flags in vector registers (flatC one body, flatD four copies) against round 3's, 64 KiB and 128 KiB frames (the wide launch on the
# run machine), SQ counters of both     -> gpurun_out/r04_call4.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
L=$R/zstd-jni_amd/lib
{
cat > $OUT/ab4.txt <<X
r3base ZJNI_LIB=$L/libzjni_amd_r3base.so
flatC ZJNI_LIB=$L/libzjni_amd_flatC.so
flatD ZJNI_LIB=$L/libzjni_amd_flatD.so
r3base2 ZJNI_LIB=$L/libzjni_amd_r3base.so
flatC2 ZJNI_LIB=$L/libzjni_amd_flatC.so
flatD2 ZJNI_LIB=$L/libzjni_amd_flatD.so
X
echo "== metric 65536 x 64 KiB L3"; STEPS=3 bash tools/ab.sh $OUT/ab4.txt
cat > $OUT/ab4b.txt <<X
r3base ZJNI_LIB=$L/libzjni_amd_r3base.so
flatC ZJNI_LIB=$L/libzjni_amd_flatC.so
flatD ZJNI_LIB=$L/libzjni_amd_flatD.so
X
echo "== 65536 x 128 KiB L3 (wide launch: ZLaneD in r3base, the run machine without flags in flat*)"; STEPS=2 bash tools/ab.sh $OUT/ab4b.txt 65536 131072 3
echo "== SQ counters flatC"; bash tools/sq_counters.sh flatC ZJNI_LIB=$L/libzjni_amd_flatC.so | grep match_run
echo "== SQ counters flatD"; bash tools/sq_counters.sh flatD ZJNI_LIB=$L/libzjni_amd_flatD.so | grep match_run
} > $OUT/r04_call4.txt 2>&1
cat $OUT/r04_call4.txt
