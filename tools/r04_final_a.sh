#!/bin/bash
# Round-4 evidence on the GPU box, first call:  gpurun --timeout 1700 -- 'bash tools/r04_final_a.sh'
#   the -m gpu suite, the driver's own command under the kernel tracer, tools/pmc_traffic.sh (kernel stats + the two --pmc passes of the SAME library build) for the
#   metric workload and for the dominant kernels of configs 2, 3 and 5shape, SQ instruction counters + rounds per wave of the level-3 match kernel.
#   Everything under gpurun_out/r04final/; afterwards, where git is:  python tools/pmc_summary.py gpurun_out/r04final r04   — then tools/r04_final_b.sh (the bench
#   lines, which quote roofline.traffic from that summary when it carries this build's stamp).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04final; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $OUT/gpu_suite.txt; cat $OUT/gpu_suite.txt
# the driver's own command under the kernel tracer
( cd /tmp; export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/bench.py --steps 5 --warmup 2 > $OUT/bench_metric_with_stats.json 2> $OUT/bench_stats.err
  f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_metric_kernel_stats.csv; rm -rf $OUT/stats )
PMC_LIST=$'metric 3 65536 65536 ZJNI_NEED_INLINE=1\nmetricnoflags 3 65536 65536 ZJNI_NEED=0\n5shape 3 65536 131072\n5shapenoflags 3 65536 131072 ZJNI_NEED_WIDE=0\n3 1 65536 65536\n2 3 65536 65536 ZJNI_NEED_INLINE=1' bash tools/pmc_traffic.sh r04final 2>&1 | tail -6
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
