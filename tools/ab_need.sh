# A/B of the need-gated level-3 machine (zj_need.h) on the metric configuration: tools/ab_need.sh, run on a GPU box from the repo root.
run() { name=$1; shift; env "$@" timeout 90 python bench.py --steps 3 --warmup 1 --skip-cpu > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; }
rm -f gpurun_out/ab_*.json
ZJNI_NEED=2 ZJNI_SPLIT_MIN=1 timeout 100 python -m pytest tests/test_gpu_encode.py -m gpu -x -q -k "edge or mixed or wide or explicit or checksum or tight or need" > gpurun_out/need_tests.log 2>&1; tail -2 gpurun_out/need_tests.log
ZJNI_NEED=2 timeout 100 python bench.py --steps 3 --warmup 1 --e2e-sample 0 --cpu-sample 512 --cpu-seconds 0.2 > gpurun_out/ab_need2_verified.json 2> gpurun_out/ab_need2_verified.err
run need2 ZJNI_NEED=2
run need3 ZJNI_NEED=3
run need4 ZJNI_NEED=4
run need1 ZJNI_NEED=1
run base ZJNI_NEED=0
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernel_ms"]
        print(f.split("ab_")[1], "value %.2f compress %.2f | call %.1f match %.1f rest %.1f | parity %s" % (d["value"], d["compress_GiBps_per_gpu"], k["compress_call"], [v for n, v in k.items() if n.startswith("zj_enc_match")][0], [v for n, v in k.items() if n.startswith("compress_rest")][0], d.get("parity",{}).get("frames_byte_identical_to_reference")))
    except Exception as e: print(f, "FAILED", e)
PY
