# Round 3: (1) GPU decode tests with the literal pass (stage 2b) in place, (2) same-box A/B of the pass on the metric configuration,
# (3) per-phase cycle profile of the multi-block wave matcher (library variant built by tools/build_variant.sh zxprof -DZX_PROFILE=1).
#   gpurun -- 'bash tools/ab_call2.sh'   -> gpurun_out/call2.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
echo "== gpu decode tests"; timeout 300 python -m pytest tests/test_gpu_decode.py tests/test_gpu_zz_corrupt.py tests/test_gpu_zz_fuzz_decode.py -x -q 2>&1 | tail -4
cat > $OUT/ab3.txt <<X
declit_off ZJNI_DEC_LIT=0
declit_on ZJNI_DEC_LIT=1
declit_off2 ZJNI_DEC_LIT=0
declit_on2 ZJNI_DEC_LIT=1
X
echo "== literal pass A/B, 65536 x 64 KiB level 3"; STEPS=3 bash tools/ab.sh $OUT/ab3.txt 65536 65536 3
python - <<PY
import json
for line in open("$OUT/ab.jsonl"):
    d = json.loads(line); print(d.get("tag"), d.get("stages_ms"))
PY
echo "== wave matcher phase profile, 1024 x 1 MiB level 3 (synthetic classes: frame index & 3)"
ZJNI_LIB=$R/zstd-jni_amd/lib/libzjni_amd_zxprof.so AB_TAG=zxprof timeout 120 python tools/prof_driver.py 1024 1048576 3 1 2>&1 | grep "^zx wg" | sort | awk 'NR % 6 == 1' | head -40
} > $OUT/call2.txt 2>&1
cat $OUT/call2.txt
