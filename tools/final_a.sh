#!/bin/bash
# A round's evidence on the GPU box, first call:  gpurun --timeout 1700 -- 'bash tools/final_a.sh r05'
#   the -m gpu suite, the driver's own command under the kernel tracer, tools/pmc_traffic.sh (kernel stats + the two --pmc passes of the SAME library build) for the
#   metric workload and for the dominant kernels of configs 2, 3 and 5shape, SQ instruction counters + rounds per wave of the level-3 match kernel.
#   Everything under gpurun_out/<round>final/; afterwards, where git is:  python tools/pmc_summary.py gpurun_out/<round>final <round>   — then tools/final_b.sh <round> (the bench
#   lines, which quote roofline.traffic from that summary when it carries this build's stamp).
RND=${1:-r05}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/${RND}final; mkdir -p $OUT
cd $R
[ -z "$SKIP_SUITE" ] && { timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $OUT/gpu_suite.txt; cat $OUT/gpu_suite.txt; }
# the driver's own command under the kernel tracer
( cd /tmp; export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/bench.py --steps 5 --warmup 2 > $OUT/bench_metric_with_stats.json 2> $OUT/bench_stats.err
  f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_metric_kernel_stats.csv; rm -rf $OUT/stats )
PMC_LIST=$'metric 3 65536 65536 ZJNI_NEED_INLINE=1\nmetricnoflags 3 65536 65536 ZJNI_NEED=0\n5shape 3 65536 131072\n3 1 65536 65536\n2 3 65536 65536 ZJNI_NEED_INLINE=1\n1 3 1024 1048576 PROF_DATA=xml\n4 3 1048576 4096' bash tools/pmc_traffic.sh ${RND}final 2>&1 | tail -6
echo "== SQ counters of the match kernel (final build)"; bash tools/sq_counters.sh final | grep match_run > $OUT/sq_counters.txt; cat $OUT/sq_counters.txt
ls $OUT
