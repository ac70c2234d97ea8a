# SQ instruction / cycle counters of the level-3 match kernel on the metric workload, one rocprofv3 --pmc pass per counter set (no trace domains):
#   bash tools/sq_counters.sh <tag> [ENV=.. ...]      (e.g. ZJNI_LIB=.../libzjni_amd_x.so)   -> lines on stdout
# dynamic wave-instructions per round = SQ_INSTS_* / (SQ_WAVES x rounds per wave); rounds per wave come from a -DZL_PROFILE build (tools/r04_call*.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"; do
  rm -rf $OUT/sq_$TAG; env ZJNI_NEED_INLINE=1 "$@" timeout 200 rocprofv3 --pmc $SET --output-format csv -d $OUT/sq_$TAG -o p -- python $R/tools/prof_driver.py 65536 65536 3 1 > /dev/null 2> $OUT/sq_$TAG.err
  f=$(find $OUT/sq_$TAG -name '*counter_collection.csv' | head -1)
  python - <<PY
import csv, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
try:
    for r in csv.DictReader(open("$f")):
        k = r.get("Kernel_Name", "?").split("(")[0]
        if "match_run" not in k and "match_wide" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in tot:
        for c, v in sorted(tot[k].items()): print("$TAG", k[:40], c, "launches", cnt[(k, c)], "per launch %.4e" % (v / cnt[(k, c)]))
except Exception as e: print("$TAG sq failed", e); print(open("$OUT/sq_$TAG.err").read()[-600:])
PY
done
rm -rf $OUT/sq_$TAG
