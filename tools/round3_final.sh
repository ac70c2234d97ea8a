#!/bin/bash
# Round 3, last GPU action:  gpurun --timeout 900 -- 'bash tools/round3_final.sh'   -> gpurun_out/r03final2/
#   (1) the whole -m gpu suite on the final library, (2) bench.py lines of the configs this half of the round touched (metric, 1), (3) rocprofv3 --kernel-trace --stats of the metric workload, (4) the two --pmc passes of the same (tools/pmc_traffic.sh's recipe, one key),
#   (5) last: the multi-block tests with levels 1-2 on the wave matcher, its A/B against the one-lane parse, and small level-3 batches on the wave route against the lane pipeline.  Afterwards, where git is:
#   python tools/pmc_summary.py gpurun_out/r03final2 r03 ; copy the lines into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03final2; mkdir -p $OUT/pmc
cd $R
# the suite first with levels 1-2 of multi-block frames on the one-lane parse (the newest kernel code — ZWaveF — gets its own run at the very end,
# so that whatever it does, the measurements above it are in)
ZJNI_MULTI_WAVE_FAST=0 timeout 300 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for C in metric 1; do
  timeout 200 python bench.py --config $C --steps 3 --warmup 1 > $OUT/bench_config$C.json 2> $OUT/bench_config$C.err
done
python - <<PY
import json
for C in ("metric", "1"):
    try:
        d = json.loads(open("$OUT/bench_config%s.json" % C).read().strip().splitlines()[-1])
        e = d.get("end_to_end") or {}
        print(C, "value %.2f compress %s decompress %.1f | e2e %s / %s | cpu %s | parity %s" % (d["value"], d["compress_GiBps_per_gpu"] and round(d["compress_GiBps_per_gpu"], 2), d["decompress_GiBps_per_gpu"], e.get("compress_GiBps") and round(e["compress_GiBps"], 1), e.get("decompress_GiBps") and round(e["decompress_GiBps"], 1), d["cpu_baseline"].get("compress_GiBps") and round(d["cpu_baseline"]["compress_GiBps"], 1), d["parity"]))
    except Exception as ex: print(C, "FAILED", ex)
PY
cd /tmp; export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_m -o s -- python $R/tools/prof_driver.py 65536 65536 3 3 > $OUT/metric_L3_65536x65536_driver.json 2> $OUT/stats_m.err
f=$(find $OUT/stats_m -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/metric_L3_65536x65536_kernel_stats.csv && head -7 $f | cut -c1-160; rm -rf $OUT/stats_m
KEY=metric_L3_65536x65536
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 150 env ZJNI_NEED_INLINE=1 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc/${KEY}_$C -o p -- python $R/tools/prof_driver.py 65536 65536 3 1 > $OUT/pmc/${KEY}_${C}_driver.json 2> $OUT/pmc/${KEY}_$C.err
  f=$(find $OUT/pmc/${KEY}_$C -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/pmc/${KEY}_$C.csv
  rm -rf $OUT/pmc/${KEY}_$C
done
cd $R
echo "== levels 1-2 of multi-block frames on the wave matcher (library default)"
timeout 150 python -m pytest tests/test_gpu_multiblock.py -m gpu -q > $OUT/tests_fast_wave.log 2>&1; tail -3 $OUT/tests_fast_wave.log
cat > $OUT/abL1.txt <<X
l1_wave ZJNI_MULTI_WAVE_FAST=1
l1_one_lane ZJNI_MULTI_WAVE_FAST=0
X
echo "== level 1, 2048 x 512 KiB"; STEPS=2 bash tools/ab.sh $OUT/abL1.txt 2048 524288 1 | tee $OUT/level1_multiblock_ab.txt
cat > $OUT/abR.txt <<X
l3_wave_route ZJNI_L3_WAVE_MAX=8192
l3_lane_pipeline ZJNI_L3_WAVE_MAX=0
X
echo "== level 3, 1024 x 64 KiB: wave route against the lane pipeline"; STEPS=2 bash tools/ab.sh $OUT/abR.txt 1024 65536 3 | tee $OUT/level3_small_batch_1024_ab.txt
echo "== level 3, 4096 x 64 KiB"; STEPS=2 bash tools/ab.sh $OUT/abR.txt 4096 65536 3 | tee $OUT/level3_small_batch_4096_ab.txt
ls $OUT $OUT/pmc | head -40
