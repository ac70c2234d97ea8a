# round 4, call 11: stage 2 of multi-block frames with cells and bitstream staged in LDS: tests, config 1 decompress per lane count, kernel stats   -> gpurun_out/r04_call11.txt (the library of profiles/r04/patches/i_seq_mb_lds.diff; LANES=0 = the shipped kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
timeout 900 python -m pytest tests/test_gpu_decode_multiblock.py tests/test_gpu_decode.py tests/test_gpu_zz_corrupt.py tests/test_gpu_zz_fuzz_decode.py -m gpu -x -q 2>&1 | tail -6
for NB in 1024 4096; do for LN in 0 24 40 56; do
echo "== config 1, $NB buffers, ZJNI_DEC_MB_LANES=$LN"; ZJNI_DEC_MB_LANES=$LN timeout 300 python bench.py --config 1 --buffers $NB --steps 3 --skip-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('compress_GiBps_per_gpu','decompress_GiBps_per_gpu')}, d['kernel_ms'].get('decompress_call'))"
done; done
for NB in 1024 4096; do
echo "== kernel stats, config 1, $NB buffers"; cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/st9; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st9 -o s -- python $R/bench.py --config 1 --buffers $NB --steps 3 --skip-cpu > $OUT/b11_$NB.json 2>/dev/null
f=$(find $OUT/st9 -name '*kernel_stats.csv' | head -1)
python3 - <<PY
import csv, json
try:
    for r in csv.DictReader(open("$f")):
        if r["Name"].startswith(("zj_dec", "void zj_dec")): print("  %-34s calls %4s avg %9.3f ms" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e6))
except Exception as e: print("stats failed", e)
PY
rm -rf $OUT/st9; cd $R
done
} > $OUT/r04_call11.txt 2>&1
cat $OUT/r04_call11.txt
