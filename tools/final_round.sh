#!/bin/bash
# Round-end evidence on the GPU box (gpurun -- 'bash tools/final_round.sh'): the level-3 tests, the metric line, rocprofv3 kernel stats of the same workload.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r02zz; mkdir -p $OUT
cd $R
timeout 80 python -m pytest tests/test_gpu_encode.py tests/test_gpu_multi_and_scratch.py tests/test_gpu_aggregator.py tests/test_gpu_level4.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 70 python bench.py --steps 5 --warmup 2 --e2e-sample 0 --cpu-seconds 0.5 > $OUT/bench_metric.json 2> $OUT/bench_metric.err; tail -c 600 $OUT/bench_metric.json
cd /tmp; export TMPDIR=/tmp
timeout 70 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/tools/prof_driver.py 65536 65536 3 3 > $OUT/driver.json 2> $OUT/stats.err
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/metric_L3_65536x65536_kernel_stats.csv && head -8 $f
rm -rf $OUT/stats
