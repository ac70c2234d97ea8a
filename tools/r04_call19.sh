# round 4, call 19: the flag kernel with 512 / 1 024 lanes per frame: metric configuration (tools/ab.sh), then 65 536 x 128 KiB    -> gpurun_out/r04_call19.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
STEPS=4 bash tools/ab.sh tools/ab/need_threads.txt
for V in "ZJNI_NEED_THREADS=512" "ZJNI_NEED_THREADS=1024" "ZJNI_NEED_THREADS=512" "ZJNI_NEED_THREADS=1024"; do
echo "== 5shape $V"; env $V timeout 400 python bench.py --config 5shape --steps 3 --skip-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','compress_GiBps_per_gpu')}, {k: round(v, 1) for k, v in d['kernel_ms'].items() if isinstance(v, (int, float)) and 'dec' not in k})"
done
} > $OUT/r04_call19.txt 2>&1
cat $OUT/r04_call19.txt
