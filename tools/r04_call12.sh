# round 4, call 12: tANS tables built by the wave (ze_tans_*): encode-side GPU tests, then the lines the entropy stage shows in (config 4, metric, configs 2 / 3 / 5shape)   -> gpurun_out/r04_call12.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
timeout 1200 python -m pytest tests/test_gpu_encode.py tests/test_gpu_cdict.py tests/test_gpu_level4.py tests/test_gpu_multiblock.py tests/test_gpu_stream.py tests/test_gpu_decode_multiblock.py tests/test_gpu_zz_fuzz.py -m gpu -x -q 2>&1 | tail -6
for CFG in 4 metric 2 3 5shape; do
echo "== config $CFG"; timeout 400 python bench.py --config $CFG --steps 5 --skip-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','compress_GiBps_per_gpu','decompress_GiBps_per_gpu')}, {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, (int, float))})"
done
} > $OUT/r04_call12.txt 2>&1
cat $OUT/r04_call12.txt
