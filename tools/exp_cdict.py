"""Experiment: BASELINE config 4 — n x 4 KiB JSON-like records, level 3 (and 1), compressed on the GPU with a shared
ZstdDictCompress dictionary, checked against the reference on a sample, then decompressed on the GPU with the same
dictionary.  usage: exp_cdict.py [rep]   (n = 4096 * rep records)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as e
from util import json_records
zj = e.load_package(); B = zj.batch
from oracle import ref
B.init(0)
import ctypes as C
PROF = os.environ.get("ZJNI_PROFILE") is not None
if PROF:
    zj.lib().zjni_debug_read_profile.argtypes = [C.c_void_p]
def read_prof():
    a = (C.c_ulonglong * 32)(); assert zj.lib().zjni_debug_read_profile(a) == 0; return list(a)
ENC = ["params", "match(l0)", "lit gather+codes", "hist+huf decide", "huf encode", "seq tables", "seq encode", "block place"]
U = 4096; rep = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = U * rep
if os.environ.get("CD_TILED"):
    # 4096 distinct records tiled (cache-friendly; the first measurements used this)
    recs = []
    for i in range(U):
        r = b"".join(json_records(40, seed=i, first=i * 40))[:4096]
        recs.append(r + b" " * (4096 - len(r)))
    dic = ref.train_dict([x for i in range(0, U, 4) for x in json_records(40, seed=i, first=i * 40)], 112640)
    src = torch.from_numpy(np.tile(np.frombuffer(b"".join(recs), dtype=np.uint8), rep)).cuda()
else:
    # n distinct records: the JSON class of the SURVEY 8(d) generator (index & 3 == 1) at 4 KiB
    src = B.synth(4 * n, 4096, 0).view(4 * n, 4096)[1::4].contiguous().view(-1)
    host = zj.synth_host(4096, 0, 4 * 3000)
    samples = [host[(4 * i + 1) * 4096:(4 * i + 2) * 4096] for i in range(3000)]
    dic = ref.train_dict(samples, 112640)
    recs = [bytes(src[i * 4096:(i + 1) * 4096].cpu().numpy().tobytes()) for i in range(64)]
off = B.uniform_offsets(n, 4096, "cuda")
bound = zj.Zstd.compressBound(4096)
dst = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); doff = B.uniform_offsets(n, bound, "cuda")
back = torch.empty(n * 4096, dtype=torch.uint8, device="cuda")
for level in (3, 1):
    cd = zj.ZstdDictCompress(dic, level); dd = zj.ZstdDictDecompress(dic); rcd = ref.CDict(dic, level)
    for name, dictionary in (("dict", cd), ("plain", None)):
        for it in range(3):
            if PROF: read_prof()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); res = B.compress(src, off, dst, doff, level=level, dictionary=dictionary); e1.record(); torch.cuda.synchronize()
        if PROF:
            pe = read_prof(); te = sum(pe[16:24])
            print("   entropy kernel cycles/frame %.0f: " % (te / n) + ", ".join(f"{ENC[i]} {100*pe[16+i]/max(te,1):.0f}%" for i in range(8)))
        ms = e0.elapsed_time(e1); tm = B.last_timing()
        sizes = res.cpu().numpy()
        ok = bool((sizes > 0).all())
        if dictionary is not None:
            outb = dst[:bound * 64].cpu().numpy().tobytes()
            ok = ok and all(outb[i * bound:i * bound + sizes[i]] == rcd.compress(recs[i]) for i in range(64))
            packed, poff = B.pack(res, dst, doff)
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d0.record(); r2 = B.decompress(packed, poff, back, off, dictionary=dd); d1.record(); torch.cuda.synchronize()
            ok = ok and bool((r2 == 4096).all()) and torch.equal(back, src)
            dms = d0.elapsed_time(d1)
        else:
            dms = float("nan")
        print(f"L{level} {name}: n={n} ratio {4096.0*n/sizes.sum():.2f} compress {ms:.2f} ms = {n*4096/2**30/(ms/1e3):.1f} GiB/s (match {tm['match']:.2f} ms) decompress {dms:.2f} ms ok={ok}", flush=True)
    cd.close(); dd.close()
