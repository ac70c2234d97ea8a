"""Experiment: one launch of the level-L match path on n frames of one class (see exp_match.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as e
zj = e.load_package(); B = zj.batch
B.init(0)
size = 65536
level = int(sys.argv[1]); n = int(sys.argv[2]); cls = None if sys.argv[3] == "mix" else int(sys.argv[3])
if cls is None: src = B.synth(n, size, 0)
else:
    parts = [B.synth(1, size, 4 * k + cls) for k in range(256)]
    src = torch.cat(parts * max(1, n // 256))[: n * size].contiguous()
soff = B.uniform_offsets(n, size, "cuda"); bound = zj.Zstd.compressBound(size)
comp = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); coff = B.uniform_offsets(n, bound, "cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); B.compress(src, soff, comp, coff, level); e1.record(); torch.cuda.synchronize()
print(f"L{level} n={n} class={cls}: {e0.elapsed_time(e1):8.2f} ms", flush=True)
