"""Per-phase cycle breakdown of the encode/decode kernels (ZJNI_PROFILE=1), per frame class."""
import os, sys, ctypes as C
os.environ["ZJNI_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as e
zj = e.load_package(); L = zj.lib(); B = zj.batch
B.init(0)
L.zjni_debug_read_profile.argtypes = [C.c_void_p]
def read():
    a = (C.c_ulonglong * 32)(); assert L.zjni_debug_read_profile(a) == 0; return list(a)
n, size = 4096, 65536
DEC = ["lit hdr", "huf table", "huf decode", "seq hdr/tables", "seq decode(l0)", "execute", "last literals", "raw copy", "frame tail"]
ENC = ["params+zero", "match find(l0)", "lit gather+codes", "hist+huf build", "huf encode", "seq tables", "seq encode(l0)", "block place"]
for level in (int(os.environ.get('PP_LEVEL', '1')),):
    for cls in (None, 0, 1, 2, 3):
        # buffers of one class: indices cls, cls+4, ...  (generator class = index & 3)
        if cls is None: src = B.synth(n, size, 0); name = "mixed"
        else:
            src = torch.empty(n * size, dtype=torch.uint8, device="cuda")
            for k in range(n): pass
            parts = [B.synth(1, size, 4 * k + cls) for k in range(256)]
            src = torch.cat(parts * (n // 256)); name = ["text", "json", "lowent", "random"][cls]
        soff = B.uniform_offsets(n, size, "cuda"); bound = zj.Zstd.compressBound(size)
        comp = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); coff = B.uniform_offsets(n, bound, "cuda")
        back = torch.empty(n * size, dtype=torch.uint8, device="cuda")
        B.compress(src, soff, comp, coff, level); torch.cuda.synchronize(); read()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); csz = B.compress(src, soff, comp, coff, level); e1.record(); torch.cuda.synchronize()
        pe = read()
        packed, poff = B.pack(csz, comp, coff); torch.cuda.synchronize()
        e1.record(); B.decompress(packed, poff, back, soff); e2.record(); torch.cuda.synchronize()
        pd = read()
        te, td = sum(pe[16:]), sum(pd[:16])
        print(f"wg/cu={os.environ.get('ZJNI_DEBUG_WG_PER_CU')} L{level} {name:7s} enc {e0.elapsed_time(e1) if False else 0:.0f} ratio {n*size/int(csz.sum()):.2f} | ENC cyc/frame {te/n/1e3:.0f}k: " + ", ".join(f"{ENC[i]} {100*pe[16+i]/max(te,1):.0f}%" for i in range(8)))
        print(f"            | DEC cyc/frame {td/n/1e3:.0f}k: " + ", ".join(f"{DEC[i]} {100*pd[i]/max(td,1):.0f}%" for i in range(9)), flush=True)
