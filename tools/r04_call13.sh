# round 4, call 13: the whole GPU suite on the build with the wave's tANS tables, then the parity gates of the bench lines   -> gpurun_out/r04_call13.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for CFG in metric 4 1 5shape; do
echo "== config $CFG"; timeout 400 python bench.py --config $CFG --steps 3 --skip-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','compress_GiBps_per_gpu','decompress_GiBps_per_gpu')}, d.get('parity'), d.get('gates'))"
done
} > $OUT/r04_call13.txt 2>&1
cat $OUT/r04_call13.txt
