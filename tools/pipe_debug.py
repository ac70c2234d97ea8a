"""one compress call of 1 024 x 1 MiB xml slices at level 3 on a -DZE_PIPE_DEBUG build (ZJNI_LIB): the pipelined kernel's per-role times of the first workgroups"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as e
from oracle import ref
zj = e.load_package(); B = zj.batch; B.init(0)
n, size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 1 << 20
xml = np.frombuffer(ref.decompress(open(os.path.join(ROOT, "tests", "golden", "xml-1.zst"), "rb").read(), 6_000_000), dtype=np.uint8)
hx = torch.from_numpy(xml.copy()).cuda(); span = xml.size - size
src = torch.empty(n * size, dtype=torch.uint8, device="cuda")
for i in range(n): o = (i * 4099) % span; src[i * size:(i + 1) * size] = hx[o:o + size]
off = B.uniform_offsets(n, size, "cuda"); bound = zj.Zstd.compressBound(size)
comp = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); coff = B.uniform_offsets(n, bound, "cuda")
for it in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); csz = B.compress(src, off, comp, coff, 3); e1.record(); torch.cuda.synchronize()
    print("call", it, "ms", round(e0.elapsed_time(e1), 2), "errors", int((csz <= 0).sum()), flush=True)
k = 8
want = [ref.compress(src[i * size:(i + 1) * size].cpu().numpy().tobytes(), 3) for i in range(k)]
got = [comp[i * bound:i * bound + int(csz[i])].cpu().numpy().tobytes() for i in range(k)]
print("first frames identical:", [a == b for a, b in zip(want, got)], [int(c) for c in csz[:k]], [len(w) for w in want])
