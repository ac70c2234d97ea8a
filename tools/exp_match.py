"""Experiment: level-3 compress time vs number of frames of one generator class (latency vs congestion)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as e
zj = e.load_package(); B = zj.batch
B.init(0)
size = 65536
level = int(os.environ.get("EXP_LEVEL", "3"))
def run(n, cls):
    if cls is None: src = B.synth(n, size, 0)
    else:
        parts = [B.synth(1, size, 4 * k + cls) for k in range(256)]
        src = torch.cat(parts * max(1, n // 256))[: n * size].contiguous()
    soff = B.uniform_offsets(n, size, "cuda"); bound = zj.Zstd.compressBound(size)
    comp = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); coff = B.uniform_offsets(n, bound, "cuda")
    B.compress(src, soff, comp, coff, level); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); B.compress(src, soff, comp, coff, level); e1.record(); torch.cuda.synchronize()
    print(f"L{level} n={n:6d} class={cls}: {e0.elapsed_time(e1):8.2f} ms", flush=True)
for n, cls in [(4096, 2), (16384, 2), (65536, 2), (16384, 0), (65536, 0), (65536, None)]:
    run(n, cls)
