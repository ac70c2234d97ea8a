"""Open-ended run of tests/gpu_fuzz_decode.py: valid and damaged frames through the C-ABI on the GPU, both decode pipelines, with and
without dictionary, against the reference's portable decoder loops.  usage: fuzz_gpu_decode.py <seed> <cases-per-round> <rounds>
TEST INFRASTRUCTURE."""
import os, sys, time
os.environ.setdefault("ZJNI_DEBUG_LIVE_SWITCHES", "1")    # the library caches its ZJNI_* switches per process (zj_env); this tool flips them between calls
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as e
import gpu_fuzz_decode as F
import util
from oracle import ref
zj = e.load_package(); zj.batch.init(0)
seed = int(sys.argv[1]); per = int(sys.argv[2]); rounds = int(sys.argv[3])
recs = util.json_records(20000, seed=seed)
dic = ref.train_dict([b",".join(recs[i * 13:i * 13 + 200])[:4096] for i in range(1000)], 60000)
total = bad_total = 0; t0 = time.time()
for r in range(rounds):
    for with_dict in (False, True):
        cases = F.make_cases(zj, ref, seed * 1000 + r * 2 + with_dict, per, dic if with_dict else None)
        dd = zj.ZstdDictDecompress(dic) if with_dict else None
        for split_min in (1, 1000000000):
            bad = F.run_cases(zj, cases, dd, split_min)
            total += len(cases); bad_total += len(bad)
            for i, why in bad[:5]:
                open(f"/tmp/fuzz_gpu_dec_{seed}_{r}_{i}.zst", "wb").write(cases[i][0]); print("DIFF", "dict" if with_dict else "plain", "split" if split_min == 1 else "fused", i, why, flush=True)
        if dd: dd.close()
    print(f"round {r}: {total} decodes, {bad_total} differences, {time.time() - t0:.0f} s", flush=True)
print("GPU-DECODE-FUZZ", "OK" if bad_total == 0 else "FAILED", "decodes", total, "differences", bad_total)
