# round 4, call 20: the run machine against ZLaneD per class of the synthetic set (text = classes 0 + 1, low-entropy = class 2), 64 KiB and 128 KiB frames   -> gpurun_out/r04_call20.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
echo "== 65 536 x 64 KiB"; STEPS=3 bash tools/ab.sh tools/ab/classes.txt 65536 65536 3
echo "== 65 536 x 128 KiB"; STEPS=2 bash tools/ab.sh tools/ab/classes_wide.txt 65536 131072 3
} > $OUT/r04_call20.txt 2>&1
cat $OUT/r04_call20.txt
