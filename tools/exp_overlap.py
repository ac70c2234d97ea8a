"""Experiment: compress latency vs batch size with the entropy kernel beside / after the match kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as e
zj = e.load_package(); B = zj.batch
B.init(0)
size = 65536
os.environ["ZJNI_SPLIT_MIN"] = "1"
for n in (1, 8, 64, 512, 4096):
    src = B.synth(n, size, 0); soff = B.uniform_offsets(n, size, "cuda"); bound = zj.Zstd.compressBound(size)
    comp = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); coff = B.uniform_offsets(n, bound, "cuda")
    for ov in (True, False):
        if ov: os.environ.pop("ZJNI_NO_OVERLAP", None)
        else: os.environ["ZJNI_NO_OVERLAP"] = "1"
        for level in (1, 3):
            B.compress(src, soff, comp, coff, level); torch.cuda.synchronize()
            t = time.time(); B.compress(src, soff, comp, coff, level); torch.cuda.synchronize(); dt = (time.time() - t) * 1e3
            print(f"n={n:5d} L{level} overlap={ov}: {dt:8.2f} ms", flush=True)
