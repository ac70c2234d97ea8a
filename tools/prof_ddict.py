"""Experiment: BASELINE config 4's decode side — n x 4 KiB JSON-like records compressed by the reference (level 3) with
and without a trained dictionary, decompressed on the GPU (dictionary frames: fused kernel; plain frames: split pipeline)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as e
from util import json_records
zj = e.load_package(); B = zj.batch
from oracle import ref
B.init(0)
U = 4096; rep = int(sys.argv[1]) if len(sys.argv) > 1 else 64
recs = []
for i in range(U):
    r = b"".join(json_records(40, seed=i, first=i * 40))[:4096]
    recs.append(r + b" " * (4096 - len(r)))
dic = ref.train_dict([x for i in range(0, U, 4) for x in json_records(40, seed=i, first=i * 40)], 112640)
for name, frames, dd in (("plain", [ref.compress(r, 3) for r in recs], None), ("dict", [ref.compress_using_dict(r, dic, 3) for r in recs], zj.ZstdDictDecompress(dic))):
    sizes = np.array([len(f) for f in frames], dtype=np.int64)
    blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
    n = U * rep
    d_blob = torch.from_numpy(np.tile(blob, rep)).cuda()
    off = np.zeros(n + 1, dtype=np.int64); off[1:] = np.cumsum(np.tile(sizes, rep)); d_off = torch.from_numpy(off).cuda()
    out = torch.empty(n * 4096, dtype=torch.uint8, device="cuda"); ooff = B.uniform_offsets(n, 4096, "cuda")
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); res = B.decompress(d_blob, d_off, out, ooff, dictionary=dd); e1.record(); torch.cuda.synchronize()
    ok = bool((res == 4096).all()) and bytes(out[:4096 * 8].cpu().numpy().tobytes()) == b"".join(recs[:8])
    ms = e0.elapsed_time(e1)
    print(f"{name}: n={n} ratio {4096*U/sizes.sum():.2f} decode {ms:.2f} ms = {n*4096/2**30/(ms/1e3):.1f} GiB/s ok={ok}", flush=True)
