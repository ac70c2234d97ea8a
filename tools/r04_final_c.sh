#!/bin/bash
# Round-4 evidence, third call: rocprofv3 --kernel-trace --stats of the bench lines of configs 1, 4 and 5shape (the metric's is in r04_final_a.sh)   -> gpurun_out/r04final/stats_config*.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04final; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in 1 4 5shape; do
  rm -rf $OUT/st_$C; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$C -o s -- python $R/bench.py --config $C --steps 3 --warmup 1 --skip-cpu > $OUT/bench_config${C}_with_stats.json 2> $OUT/st_$C.err
  f=$(find $OUT/st_$C -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/stats_config$C.csv; rm -rf $OUT/st_$C
  python3 - <<PY
import csv
print("== config $C")
try:
    for r in list(csv.DictReader(open("$OUT/stats_config$C.csv")))[:7]: print("  %-36s calls %4s avg %9.3f ms  %5s %%" % (r["Name"].split("(")[0][:36], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
except Exception as e: print("  failed", e)
PY
done
