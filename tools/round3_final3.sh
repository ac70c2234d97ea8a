#!/bin/bash
# bench.py lines of configs 1, 2, 3 on the committed library   -> gpurun_out/r03final4/
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03final4; mkdir -p $OUT; cd $R
for C in ${CONFIGS:-1 2 3}; do
  timeout 120 python bench.py --config $C --steps 3 --warmup 1 > $OUT/bench_config$C.json 2> $OUT/bench_config$C.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_config$C.json").read().strip().splitlines()[-1])
    print("$C", "value %.2f compress %s decompress %.1f | %s" % (d["value"], d["compress_GiBps_per_gpu"] and round(d["compress_GiBps_per_gpu"], 2), d["decompress_GiBps_per_gpu"], d["library"]["build_stamp"]))
except Exception as ex: print("$C FAILED", ex)
PY
done
