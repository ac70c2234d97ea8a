#!/bin/bash
# A round's evidence, second call (after tools/final_a.sh + tools/pmc_summary.py):  gpurun --timeout 1700 -- 'bash tools/final_b.sh r05'
#   every BASELINE config as a bench.py line under gpurun_out/<round>final/ — cp them to profiles/<round>_bench_config*.json
RND=${1:-r05}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/${RND}final; mkdir -p $OUT
cd $R
for C in metric 3 2 4 1 5shape; do
  timeout 300 python bench.py --config $C --steps 4 --warmup 1 > $OUT/bench_config$C.json 2> $OUT/bench_config$C.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_config$C.json").read().strip().splitlines()[-1])
    e = d.get("end_to_end") or {}
    print("$C", "value %.2f compress %s decompress %.1f | e2e %s / %s | parity %s | %s" % (d["value"], d["compress_GiBps_per_gpu"] and round(d["compress_GiBps_per_gpu"], 2), d["decompress_GiBps_per_gpu"], e.get("compress_GiBps") and round(e["compress_GiBps"], 1), e.get("decompress_GiBps") and round(e["decompress_GiBps"], 1), d["parity"], d["library"]))
except Exception as ex: print("$C FAILED", ex)
PY
done
timeout 300 python bench.py --config 1 --buffers 4096 --steps 3 --warmup 1 > $OUT/bench_config1_4096.json 2> $OUT/bench_config1_4096.err; python -c "
import json; d=json.loads(open('$OUT/bench_config1_4096.json').read().strip().splitlines()[-1]); print('1_4096 value %.2f compress %.2f decompress %.1f' % (d['value'], d['compress_GiBps_per_gpu'], d['decompress_GiBps_per_gpu']))"
timeout 400 python bench.py --config 5 --buffers 131072 --steps 2 --warmup 1 > $OUT/bench_config5_two_chunks.json 2> $OUT/bench_config5.err; tail -c 400 $OUT/bench_config5_two_chunks.json | head -c 400; echo
