run() { # name, env...
  name=$1; shift
  env "$@" timeout 90 python bench.py --steps 3 --warmup 1 --skip-cpu > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
}
export ZJNI_NEED=1
ZJNI_SPLIT_MIN=1 timeout 100 python -m pytest tests/test_gpu_encode.py -m gpu -x -q -k "edge or mixed or wide or explicit or checksum or tight" > gpurun_out/need_tests.log 2>&1; tail -2 gpurun_out/need_tests.log
timeout 100 python bench.py --steps 3 --warmup 1 --e2e-sample 0 --cpu-sample 512 --cpu-seconds 0.2 > gpurun_out/ab_need_p8.json 2> gpurun_out/ab_need_p8.err
run need_p6 ZJNI_LANE_PERIOD=6
run need_p5 ZJNI_LANE_PERIOD=5
run need_p4 ZJNI_LANE_PERIOD=4
run need_p3 ZJNI_LANE_PERIOD=3
unset ZJNI_NEED
run base_p8 X=1
run base_p5 ZJNI_LANE_PERIOD=5
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernel_ms"]
        print(f.split("ab_")[1], "value %.2f compress %.2f | call %.1f match %.1f rest %.1f | parity %s" % (d["value"], d["compress_GiBps_per_gpu"], k["compress_call"], k["zj_enc_match_kernel"], k["compress_rest(entropy beside match, memset, sweep)"], d.get("parity",{}).get("frames_byte_identical_to_reference")))
    except Exception as e: print(f, "FAILED", e)
PY
