import os, sys, random
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as e
zj = e.load_package(); zj.batch.init(0)
from oracle import ref
import test_gpu_multiblock as T
import test_gpu_level4 as L4
datas = {lvl: T.inputs(zj, ref, 100 + lvl, 60) for lvl in (1, 2, 3)}
want = {}
for lvl in (1, 2, 3):
    for ck in (False, True):
        want[lvl, ck] = [None if len(d) > T.WINDOW[lvl] else ref.compress(d, lvl, ck) for d in datas[lvl]]
l4 = L4._inputs(zj, 5, int(os.environ.get("L4N", "4500")))
bad = 0
for it in range(int(sys.argv[1])):
    if os.environ.get("WITH_L4", "1") == "1": zj.compress_batch(l4, 4)
    for lvl in (1, 2, 3):
        for seqno, ck in enumerate((False, False, True, True, False)):
            outs = zj.compress_batch(datas[lvl], lvl, checksum=ck)
            for i, (z, w) in enumerate(zip(outs, want[lvl, ck])):
                if w is None: continue
                if z != w:
                    bad += 1
                    k = next((j for j in range(min(len(z), len(w))) if z[j] != w[j]), -1) if not isinstance(z, Exception) else -2
                    os.makedirs(os.path.join(ROOT, "gpurun_out", "mbbad"), exist_ok=True)
                    tag = f"l{lvl}_f{i}_{int(ck)}"
                    if not isinstance(z, Exception) and bad <= 12:
                        open(os.path.join(ROOT, "gpurun_out", "mbbad", tag + f"_got{bad}.zst"), "wb").write(z); open(os.path.join(ROOT, "gpurun_out", "mbbad", tag + "_want.zst"), "wb").write(w)
                        open(os.path.join(ROOT, "gpurun_out", "mbbad", tag + "_src.bin"), "wb").write(datas[lvl][i])
                    print("MISMATCH it", it, "seq", seqno, "lvl", lvl, "ck", ck, "frame", i, "size", len(datas[lvl][i]), "first diff", k, "lens", (len(z) if not isinstance(z, Exception) else z), len(w), flush=True)
print("iterations", sys.argv[1], "bad", bad)
