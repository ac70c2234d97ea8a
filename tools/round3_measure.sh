#!/bin/bash
# Round-3 evidence on the GPU box:  gpurun --timeout 1200 -- 'bash tools/round3_measure.sh'
#   every BASELINE config as a bench.py line (5 steps), then tools/pmc_traffic.sh (rocprofv3 --kernel-trace --stats + the two --pmc passes of the SAME library
#   build) for the metric workload and level 1.  Everything under gpurun_out/r03final/; afterwards, where git is:
#   python tools/pmc_summary.py gpurun_out/r03final r03 && cp gpurun_out/r03final/bench_config*.json profiles/ (renamed r03_...)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03final; mkdir -p $OUT
cd $R
for C in metric 3 2 4 1 5shape; do
  timeout 240 python bench.py --config $C --steps 5 --warmup 2 > $OUT/bench_config$C.json 2> $OUT/bench_config$C.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_config$C.json").read().strip().splitlines()[-1])
    e = d.get("end_to_end") or {}
    print("$C", "value %.2f compress %s decompress %.1f | e2e %s / %s | parity %s" % (d["value"], d["compress_GiBps_per_gpu"] and round(d["compress_GiBps_per_gpu"], 2), d["decompress_GiBps_per_gpu"], e.get("compress_GiBps") and round(e["compress_GiBps"], 1), e.get("decompress_GiBps") and round(e["decompress_GiBps"], 1), d["parity"]))
except Exception as ex: print("$C FAILED", ex)
PY
done
timeout 300 python bench.py --config 5 --buffers 131072 --steps 2 --warmup 1 > $OUT/bench_config5_two_chunks.json 2> $OUT/bench_config5.err; tail -c 300 $OUT/bench_config5_two_chunks.json | head -c 300; echo
PMC_LIST=$'metric 3 65536 65536 ZJNI_NEED_INLINE=1\nmetricnoflags 3 65536 65536 ZJNI_NEED=0\n3 1 65536 65536' bash tools/pmc_traffic.sh r03final 2>&1 | tail -8
