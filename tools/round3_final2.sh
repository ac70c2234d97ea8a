#!/bin/bash
# Round 3, the library as committed (levels 1-2 of multi-block frames back on the one-lane parse): multi-block tests, the metric line, rocprofv3 stats and the two
# --pmc passes of the same build, config 1 at 4 096 buffers, and where the level-3 wave route meets the lane pipeline.   -> gpurun_out/r03final3/
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03final3; mkdir -p $OUT/pmc
cd $R
timeout 100 python -m pytest tests/test_gpu_multiblock.py tests/test_gpu_encode.py -m gpu -q -k "multiblock or wave_route or cleared_ahead" > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 200 python bench.py --config metric --steps 4 --warmup 1 > $OUT/bench_configmetric.json 2> $OUT/bench_configmetric.err
timeout 200 python bench.py --config 1 --buffers 4096 --steps 3 --warmup 1 > $OUT/bench_config1_4096.json 2> $OUT/bench_config1_4096.err
python - <<PY
import json
for C in ("metric", "1_4096"):
    try:
        d = json.loads(open("$OUT/bench_config%s.json" % C).read().strip().splitlines()[-1])
        print(C, "value %.2f compress %s decompress %.1f | cpu %s | %s" % (d["value"], d["compress_GiBps_per_gpu"] and round(d["compress_GiBps_per_gpu"], 2), d["decompress_GiBps_per_gpu"], d["cpu_baseline"].get("compress_GiBps") and round(d["cpu_baseline"]["compress_GiBps"], 1), d["library"]))
    except Exception as ex: print(C, "FAILED", ex)
PY
cat > $OUT/abR.txt <<X
l3_wave_route ZJNI_L3_WAVE_MAX=100000
l3_lane_pipeline ZJNI_L3_WAVE_MAX=0
X
for N in 8192 12288; do echo "== level 3, $N x 64 KiB"; STEPS=2 bash tools/ab.sh $OUT/abR.txt $N 65536 3 | tee $OUT/level3_batch_${N}_ab.txt; done
cd /tmp; export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_m -o s -- python $R/tools/prof_driver.py 65536 65536 3 3 > $OUT/metric_L3_65536x65536_driver.json 2> $OUT/stats_m.err
f=$(find $OUT/stats_m -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/metric_L3_65536x65536_kernel_stats.csv && head -4 $f | cut -c1-120; rm -rf $OUT/stats_m
KEY=metric_L3_65536x65536
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 150 env ZJNI_NEED_INLINE=1 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc/${KEY}_$C -o p -- python $R/tools/prof_driver.py 65536 65536 3 1 > $OUT/pmc/${KEY}_${C}_driver.json 2> $OUT/pmc/${KEY}_$C.err
  f=$(find $OUT/pmc/${KEY}_$C -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/pmc/${KEY}_$C.csv
  rm -rf $OUT/pmc/${KEY}_$C
done
ls $OUT/pmc | head
