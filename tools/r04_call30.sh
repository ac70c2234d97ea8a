# round 4, call 30: the dictionary pipeline's slice (records per match-kernel launch) with the launch sized by the kernel's own occupancy: config 4 at 131 072 / 262 144 / 524 288   -> gpurun_out/r04_call30.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
for V in "ZJNI_CD_SLICE=131072" "ZJNI_CD_SLICE=262144" "ZJNI_CD_SLICE=524288" "ZJNI_CD_SLICE=131072" "ZJNI_CD_SLICE=262144"; do
echo "== config 4 $V"; env $V timeout 300 python bench.py --config 4 --steps 4 --warmup 1 --skip-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: round(d.get(k),2) for k in ('value','compress_GiBps_per_gpu','decompress_GiBps_per_gpu')}, d.get('parity'), {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, (int, float)) and 'dec' not in k})"
done
} > $OUT/r04_call30.txt 2>&1
cat $OUT/r04_call30.txt
