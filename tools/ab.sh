# A/B of library switches on the metric configuration with the torch-free driver: gpurun -- 'bash tools/ab.sh <file with one "tag ENV=.. ENV=.." per line> [n size level]'
# One line per variant in gpurun_out/ab.txt: compress call, match kernel, fingerprint of the compressed sizes (must agree between variants).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; : > $OUT/ab.jsonl
N=${2:-65536}; S=${3:-65536}; L=${4:-3}
while read tag envs; do
  [ -z "$tag" ] && continue
  env AB_TAG=$tag $envs timeout 45 python $R/tools/prof_driver.py $N $S $L ${STEPS:-3} >> $OUT/ab.jsonl 2> $OUT/ab_$tag.err || echo "{\"tag\": \"$tag\", \"failed\": true}" >> $OUT/ab.jsonl
done < $1
python - <<PY
import json
for line in open("$OUT/ab.jsonl"):
    try:
        d = json.loads(line)
        if d.get("failed"): print(d["tag"], "FAILED"); continue
        print("%-14s compress %7.1f ms  match %7.1f ms  decode %6.1f ms  bytes %d  sizes %s  decoded %s" % (d["tag"], d["compress_ms"], d["stages_ms"]["match"], d["decode_ms"], d["compressed_bytes"], d["csz_sha"], d["all_decoded"]))
    except Exception as e: print("??", line[:100], e)
PY
