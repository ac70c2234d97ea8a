# round 4, call 10: multi-block decode: tests, kernel stats of config 1 (1024 and 4096 buffers)     -> gpurun_out/r04_call10.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
timeout 900 python -m pytest tests/test_gpu_decode_multiblock.py tests/test_gpu_decode.py tests/test_gpu_zz_corrupt.py tests/test_gpu_zz_fuzz_decode.py -m gpu -x -q 2>&1 | tail -6
for NB in 1024 4096; do
echo "== kernel stats, config 1, $NB buffers"; cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/st9; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st9 -o s -- python $R/bench.py --config 1 --buffers $NB --steps 3 --skip-cpu > $OUT/b10_$NB.json 2>/dev/null
f=$(find $OUT/st9 -name '*kernel_stats.csv' | head -1)
python3 - <<PY
import csv, json
try:
    for r in csv.DictReader(open("$f")):
        if r["Name"].startswith(("zj_", "void zj_")): print("  %-34s calls %4s avg %9.3f ms" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e6))
except Exception as e: print("stats failed", e)
try:
    d = json.loads(open("$OUT/b10_$NB.json").read().strip().split("\n")[-1]); print("  line:", {k: d.get(k) for k in ("compress_GiBps_per_gpu", "decompress_GiBps_per_gpu")}, d["kernel_ms"].get("decompress_call"))
except Exception as e: print("line failed", e)
PY
rm -rf $OUT/st9; cd $R
done
} > $OUT/r04_call10.txt 2>&1
cat $OUT/r04_call10.txt
