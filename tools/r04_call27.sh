# round 4, call 27: stage 3 of multi-block frames beside stage 2 (completion queue per frame): tests, config 1 at 1 024 / 4 096 buffers with and without, kernel stats   -> gpurun_out/r04_call27.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
timeout 900 python -m pytest tests/test_gpu_decode_multiblock.py tests/test_gpu_decode.py tests/test_gpu_stream.py tests/test_gpu_zz_corrupt.py tests/test_gpu_zz_fuzz_decode.py -m gpu -x -q 2>&1 | tail -6
for NB in 1024 4096; do for V in "ZJNI_DEC_MB_OVERLAP=1" "ZJNI_DEC_MB_OVERLAP=0" "ZJNI_DEC_MB_OVERLAP=1" "ZJNI_DEC_MB_OVERLAP=0"; do
echo "== config 1, $NB buffers, $V"; env $V timeout 300 python bench.py --config 1 --buffers $NB --steps 3 --skip-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('decompress_GiBps_per_gpu',)}, d['kernel_ms'].get('decompress_call'), d.get('parity'))"
done; done
cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/st27; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st27 -o s -- python $R/bench.py --config 1 --steps 3 --skip-cpu > /dev/null 2>&1
f=$(find $OUT/st27 -name '*kernel_stats.csv' | head -1)
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if r["Name"].startswith(("zj_dec", "void zj_dec")): print("  %-34s calls %4s avg %9.3f ms" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
rm -rf $OUT/st27
} > $OUT/r04_call27.txt 2>&1
cat $OUT/r04_call27.txt
