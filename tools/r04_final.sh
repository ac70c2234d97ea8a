#!/bin/bash
# Round-4 evidence on the GPU box:  gpurun --timeout 1700 -- 'bash tools/r04_final.sh'
#   the -m gpu suite, every BASELINE config as a bench.py line, the driver's own command under the kernel tracer, tools/pmc_traffic.sh (kernel stats + the two
#   --pmc passes of the SAME library build) for the metric workload, SQ instruction counters + rounds per wave of the level-3 match kernel.
#   Everything under gpurun_out/r04final/; afterwards, where git is:  python tools/pmc_summary.py gpurun_out/r04final r04 && cp the lines to profiles/r04_*
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04final; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $OUT/gpu_suite.txt; cat $OUT/gpu_suite.txt
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
# the driver's own command under the kernel tracer
( cd /tmp; export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/bench.py --steps 5 --warmup 2 > $OUT/bench_metric_with_stats.json 2> $OUT/bench_stats.err
  f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_metric_kernel_stats.csv; rm -rf $OUT/stats )
PMC_LIST=$'metric 3 65536 65536 ZJNI_NEED_INLINE=1\nmetricnoflags 3 65536 65536 ZJNI_NEED=0' bash tools/pmc_traffic.sh r04final 2>&1 | tail -6
echo "== SQ counters of the match kernel (final build)"; bash tools/sq_counters.sh final | grep match_run > $OUT/sq_counters.txt; cat $OUT/sq_counters.txt
echo "== rounds per wave (ZL_PROFILE build of the same sources)"; ZJNI_NEED_INLINE=1 ZJNI_LIB=$R/zstd-jni_amd/lib/libzjni_amd_zlprof.so AB_TAG=zlprof timeout 120 python tools/prof_driver.py 65536 65536 3 1 2>&1 | grep "match lane profile" | head -4 > $OUT/zlprof.txt
python - <<PY >> $OUT/zlprof.txt
import numpy as np
try:
    wp = np.load("$R/gpurun_out/waveprof_zlprof.npy").reshape(2048, 3); w = wp[wp[:, 0] > 0]
    print("waves", len(w), "rounds per wave: min %d med %d max %d; sum of rounds over the waves %d; cycles per round med %.0f" % (w[:,2].min(), np.median(w[:,2]), w[:,2].max(), int(w[:,2].sum()), np.median(w[:,0]/np.maximum(w[:,2],1))))
except Exception as e: print("waveprof", e)
PY
cat $OUT/zlprof.txt
ls $OUT
