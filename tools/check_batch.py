"""GPU debugging helper: device-batch compress + decompress of N synthetic frames, report mismatching indices."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as e
zj = e.load_package()
from oracle import port, ref
zj.batch.init(0)
B = zj.batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
size = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
for level in (1, 3):
    src = B.synth(n, size, 0); soff = B.uniform_offsets(n, size, "cuda")
    bound = zj.Zstd.compressBound(size)
    comp = torch.zeros(n * bound, dtype=torch.uint8, device="cuda"); coff = B.uniform_offsets(n, bound, "cuda")
    csz = B.compress(src, soff, comp, coff, level); torch.cuda.synchronize()
    raw = zj.synth_host(size, 0, n)
    hc = comp.cpu().numpy(); hs = csz.cpu().numpy()
    bad = []
    for i in range(n):
        d = raw[i * size:(i + 1) * size]
        exp = ref.compress(d, level)
        got = hc[i * bound:i * bound + max(int(hs[i]), 0)].tobytes()
        if got != exp: bad.append((i, int(hs[i]), len(exp)))
    print(f"L{level}: {len(bad)} bad of {n}; first: {bad[:12]}", flush=True)
    # decode the CPU-made frames on the GPU
    frames = port.compress_many(raw, size, level, 32)
    blob = torch.from_numpy(np.frombuffer(b"".join(frames), dtype=np.uint8).copy()).cuda()
    off = np.zeros(n + 1, dtype=np.int64); off[1:] = np.cumsum([len(f) for f in frames]); doff = torch.from_numpy(off).cuda()
    out = torch.zeros(n * size, dtype=torch.uint8, device="cuda")
    res = B.decompress(blob, doff, out, soff); torch.cuda.synchronize()
    ok = (res == size)
    eq = (out.view(n, size) == src.view(n, size)).all(dim=1)
    print(f"   decode: sizes ok {int(ok.sum())}/{n}, bytes equal {int(eq.sum())}/{n}; first bad {torch.nonzero(~eq)[:8].flatten().tolist()} res {res[~eq][:4].tolist()}", flush=True)
