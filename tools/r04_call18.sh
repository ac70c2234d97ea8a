# round 4, call 18: wide launch, flags for every frame (ZJNI_NEED=1) against the picked frames (2) and none    -> gpurun_out/r04_call18.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
for V in "ZJNI_NEED=1" "ZJNI_NEED=2" "ZJNI_NEED=1" "ZJNI_NEED=2"; do
echo "== 5shape $V"; env $V timeout 400 python bench.py --config 5shape --steps 3 --skip-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','compress_GiBps_per_gpu')}, {k: round(v, 1) for k, v in d['kernel_ms'].items() if isinstance(v, (int, float)) and 'dec' not in k})"
done
cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/st18; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st18 -o s -- python $R/bench.py --config 5shape --steps 3 --skip-cpu > /dev/null 2>&1
f=$(find $OUT/st18 -name '*kernel_stats.csv' | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:8]: print("  %-34s calls %4s avg %9.3f ms" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
rm -rf $OUT/st18
} > $OUT/r04_call18.txt 2>&1
cat $OUT/r04_call18.txt
