# Round 3, multi-block frames: GPU parity of the wave matcher, then a same-box A/B of the level-3 block parse on 1 MiB frames
# (one-lane parse / wave matcher without staged spans / wave matcher), at one and two frames per SIMD.
#   gpurun -- 'bash tools/ab_multi.sh'      -> gpurun_out/ab_multi.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
( timeout 300 python -m pytest tests/test_gpu_multiblock.py -x -q 2>&1 | tail -5 ) > $OUT/ab_multi_pytest.txt
cat > $OUT/ab1.txt <<X
serial ZJNI_MULTI_WAVE=0
wave_nocarry ZJNI_MULTI_WAVE=2
wave ZJNI_MULTI_WAVE=1
X
cat > $OUT/ab2.txt <<X
wave_2048 ZJNI_MULTI_WAVE=1
wave_2048_p8 ZJNI_MULTI_WAVE=1 ZJNI_MULTI_PER_CU=8
serial_2048_p8 ZJNI_MULTI_WAVE=0 ZJNI_MULTI_PER_CU=8
X
{ echo "== 1024 x 1 MiB, level 3"; STEPS=2 bash tools/ab.sh $OUT/ab1.txt 1024 1048576 3; echo "== 2048 x 1 MiB, level 3"; STEPS=2 bash tools/ab.sh $OUT/ab2.txt 2048 1048576 3; } > $OUT/ab_multi.txt 2>&1
cat $OUT/ab_multi_pytest.txt $OUT/ab_multi.txt
