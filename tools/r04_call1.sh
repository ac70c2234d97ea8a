# round 4, call 1: where the level-3 match kernel stands on this box — bench-step times of the committed library, its per-phase cycles (ZL_PROFILE=2 build),
# and the SQ instruction counters of one launch (dynamic wave-instructions per round = SQ_INSTS_* / (waves x rounds))   -> gpurun_out/r04_call1.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
echo "== preclear ordering test"; timeout 600 python -m pytest tests/test_gpu_encode.py -x -q -m gpu -k "pending_table_clear or tables_cleared_ahead" 2>&1 | tail -5
cat > $OUT/ab1.txt <<X
base
X
echo "== metric 65536 x 64 KiB L3"; STEPS=3 bash tools/ab.sh $OUT/ab1.txt
echo "== 65536 x 128 KiB L3 (config 5 shape)"; STEPS=2 bash tools/ab.sh $OUT/ab1.txt 65536 131072 3
echo "== ZL_PROFILE=2 build"; ZJNI_LIB=$R/zstd-jni_amd/lib/libzjni_amd_zlprof.so AB_TAG=zlprof timeout 120 python tools/prof_driver.py 65536 65536 3 1 2>&1 | grep "match lane profile" | head -8
python - <<PY
import numpy as np
try:
    wp = np.load("$OUT/waveprof_zlprof.npy").reshape(2048, 3)
    w = wp[wp[:, 0] > 0]
    print("waves", len(w), "cycles M: min %.1f med %.1f max %.1f; rounds: min %d med %d max %d; cycles/round med %.0f" % (w[:,0].min()/1e6, np.median(w[:,0])/1e6, w[:,0].max()/1e6, w[:,2].min(), np.median(w[:,2]), w[:,2].max(), np.median(w[:,0]/np.maximum(w[:,2],1))))
    print("sum rounds over waves", int(w[:,2].sum()))
except Exception as e: print("waveprof", e)
PY
echo "== SQ counters (one pass)"
cd /tmp; export TMPDIR=/tmp
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"; do
  rm -rf $OUT/sq; ZJNI_NEED_INLINE=1 timeout 200 rocprofv3 --pmc $SET --output-format csv -d $OUT/sq -o p -- python $R/tools/prof_driver.py 65536 65536 3 1 > /dev/null 2> $OUT/sq.err
  f=$(find $OUT/sq -name '*counter_collection.csv' | head -1)
  python - <<PY
import csv, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
try:
    for r in csv.DictReader(open("$f")):
        k = r.get("Kernel_Name", "?").split("(")[0]
        if "match_run" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in tot:
        for c, v in sorted(tot[k].items()): print(k[:40], c, "launches", cnt[(k, c)], "per launch %.4e" % (v / cnt[(k, c)]))
except Exception as e: print("sq failed", e); print(open("$OUT/sq.err").read()[-800:])
PY
done
rm -rf $OUT/sq
} > $OUT/r04_call1.txt 2>&1
cat $OUT/r04_call1.txt
