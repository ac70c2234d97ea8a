# round 4, call 17: need flags on the wide launch (frames of 64 KiB + 1 .. 128 KiB): tests, then 65 536 x 128 KiB with and without   -> gpurun_out/r04_call17.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
timeout 900 python -m pytest tests/test_gpu_encode.py -m gpu -x -q 2>&1 | tail -4
for V in "ZJNI_NEED_WIDE=1" "ZJNI_NEED_WIDE=0" "ZJNI_NEED_WIDE=1" "ZJNI_NEED_WIDE=0"; do
echo "== 5shape $V"; env $V timeout 400 python bench.py --config 5shape --steps 3 --skip-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','compress_GiBps_per_gpu','decompress_GiBps_per_gpu')}, d.get('parity'), {k: round(v, 1) for k, v in d['kernel_ms'].items() if isinstance(v, (int, float))})"
done
} > $OUT/r04_call17.txt 2>&1
cat $OUT/r04_call17.txt
