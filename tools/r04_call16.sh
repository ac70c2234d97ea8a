# round 4, call 16: the wide launch (frames above 64 KiB) with its entropy kernel beside the match kernel: tests, then 65 536 x 128 KiB both ways   -> gpurun_out/r04_call16.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
timeout 900 python -m pytest tests/test_gpu_encode.py -m gpu -x -q 2>&1 | tail -4
for V in "" "ZJNI_NO_OVERLAP=1" "" "ZJNI_NO_OVERLAP=1"; do
echo "== 5shape $V"; env $V timeout 400 python bench.py --config 5shape --steps 3 --skip-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','compress_GiBps_per_gpu','decompress_GiBps_per_gpu')}, d.get('parity'), {k: round(v, 1) for k, v in d['kernel_ms'].items() if isinstance(v, (int, float))})"
done
} > $OUT/r04_call16.txt 2>&1
cat $OUT/r04_call16.txt
