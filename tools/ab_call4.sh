# Round 3: GPU tests of the routes this session added (level 4 on the wave matcher, decode literal pass switches), level-4 A/B on the
# metric configuration's buffers, and BASELINE config 1 (1 MiB buffers) at 1 024 and 4 096 buffers.   -> gpurun_out/call4.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
echo "== gpu tests"; timeout 400 python -m pytest tests/test_gpu_level4.py tests/test_gpu_decode.py tests/test_gpu_multiblock.py -x -q 2>&1 | tail -4
cat > $OUT/ab6.txt <<X
l4_wave ZJNI_L4_LANES=0
l4_lanes ZJNI_L4_LANES=1
X
echo "== level 4, 16384 x 64 KiB"; STEPS=1 bash tools/ab.sh $OUT/ab6.txt 16384 65536 4
echo "== bench config 1 (1024 x 1 MiB xml slices)"; timeout 300 python bench.py --config 1 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_c1_1024.json; python - <<PY
import json; d = json.load(open("$OUT/bench_c1_1024.json")); print({k: d[k] for k in ("value", "compress_GiBps_per_gpu", "decompress_GiBps_per_gpu", "ms_per_step")}, d["cpu_baseline"].get("compress_GiBps"), d["parity"])
PY
echo "== bench config 1 with 4096 buffers"; timeout 300 python bench.py --config 1 --buffers 4096 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_c1_4096.json; python - <<PY
import json; d = json.load(open("$OUT/bench_c1_4096.json")); print({k: d[k] for k in ("value", "compress_GiBps_per_gpu", "decompress_GiBps_per_gpu", "ms_per_step")}, d["cpu_baseline"].get("compress_GiBps"), d["parity"])
PY
} > $OUT/call4.txt 2>&1
cat $OUT/call4.txt
