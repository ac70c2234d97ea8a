# round 4, call 9: multi-block decode stages on the GPU: tests, then bench config 1 both ways with the stages on / off     -> gpurun_out/r04_call9.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
timeout 900 python -m pytest tests/test_gpu_decode_multiblock.py tests/test_gpu_decode.py tests/test_gpu_zz_corrupt.py tests/test_gpu_zz_fuzz_decode.py -m gpu -x -q 2>&1 | tail -12
echo "== config 1 (1024 x 1 MiB xml), block stages on"; timeout 300 python bench.py --config 1 --steps 3 --skip-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','compress_GiBps_per_gpu','decompress_GiBps_per_gpu','kernel_ms')})"
echo "== config 1, block stages off"; ZJNI_DEC_MB=0 timeout 300 python bench.py --config 1 --steps 3 --skip-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','compress_GiBps_per_gpu','decompress_GiBps_per_gpu','kernel_ms')})"

} > $OUT/r04_call9.txt 2>&1

{
echo "== config 1, literal pass off"; ZJNI_DEC_MB_LIT=0 timeout 300 python bench.py --config 1 --steps 3 --skip-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('decompress_GiBps_per_gpu','kernel_ms')})"
echo "== kernel stats, config 1"; cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st9 -o s -- python $R/bench.py --config 1 --steps 3 --skip-cpu > /dev/null 2>&1
f=$(find $OUT/st9 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 $f | cut -c1-150; rm -rf $OUT/st9
} >> $OUT/r04_call9.txt 2>&1

cat $OUT/r04_call9.txt
