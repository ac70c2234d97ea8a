#!/bin/bash
# The round's last GPU action: the driver's own command under the kernel tracer, so that the line's live HIP-event time and the profiler's average
# come from the very same process:   gpurun -- 'bash tools/final_bench_stats.sh'   ->  gpurun_out/final_bench/{bench.json,kernel_stats.csv}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/final_bench; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
rm -rf $OUT/stats
python - <<PY
import json, csv
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.2f compress %.2f decompress %.1f | %s %.1f ms frac %.5f traffic %s (x%.1f algorithmic) | stamp %s" % (d["value"], d["compress_GiBps_per_gpu"], d["decompress_GiBps_per_gpu"], r["kernel"], r["kernel_ms"], r["frac"], r["traffic"], (r["traffic"] or 0) / r["algorithmic_bytes_per_launch"], d["library"]["build_stamp"]))
for row in csv.DictReader(open("$OUT/kernel_stats.csv")):
    if float(row["TotalDurationNs"]) > 5e6: print("  %-44s calls %3s avg %8.2f ms" % (row["Name"].split("(")[0][:44], row["Calls"], float(row["AverageNs"]) / 1e6))
PY
