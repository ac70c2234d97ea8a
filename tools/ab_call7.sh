# tables cleared ahead for the next call: GPU test, then A/B on the metric configuration   -> gpurun_out/call7.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
timeout 200 python -m pytest tests/test_gpu_encode.py -x -q -k "cleared_ahead or need_gated or concurrent" 2>&1 | tail -3
cat > $OUT/ab8.txt <<X
preclear_off ZJNI_PRECLEAR=0
preclear_on ZJNI_PRECLEAR=1
preclear_off2 ZJNI_PRECLEAR=0
preclear_on2 ZJNI_PRECLEAR=1
X
STEPS=4 bash tools/ab.sh $OUT/ab8.txt 65536 65536 3
python - <<PY
import json
for line in open("$OUT/ab.jsonl"):
    d = json.loads(line); print(d.get("tag"), "step %.1f ms" % (d["compress_ms"] + d["pack_ms"] + d["decode_ms"]), d.get("stages_ms"))
PY
} > $OUT/call7.txt 2>&1
cat $OUT/call7.txt
