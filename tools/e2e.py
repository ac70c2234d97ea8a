"""End-to-end (host pointers) timing of zjni_compress_batch2 / zjni_decompress_batch on the metric batch, torch-free: python tools/e2e.py [n] [size] [reps]"""
import ctypes as C, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, __graft_entry__ as e
zj = e.load_package(); L = zj.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536; size = int(sys.argv[2]) if len(sys.argv) > 2 else 65536; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
assert L.zjni_init(0) == 0
host = np.frombuffer(zj.synth_host(size, 0, n), dtype=np.uint8)
bound = zj.Zstd.compressBound(size)
comp = np.empty(n * bound, dtype=np.uint8); back = np.empty(n * size, dtype=np.uint8)
vp = lambda base, stride: (C.c_void_p * n)(*[base + i * stride for i in range(n)])
sp, cp, bp = vp(host.ctypes.data, size), vp(comp.ctypes.data, bound), vp(back.ctypes.data, size)
ss = (C.c_size_t * n)(*([size] * n)); cc = (C.c_size_t * n)(*([bound] * n)); res = (C.c_size_t * n)(); res2 = (C.c_size_t * n)()
bc = bd = 1e9
for it in range(reps + 1):
    t0 = time.perf_counter(); r = L.zjni_compress_batch2(sp, ss, cp, cc, res, n, 3, 0); t1 = time.perf_counter(); assert not L.zjni_isError(r), r
    cs = (C.c_size_t * n)(*[res[i] for i in range(n)])
    t2 = time.perf_counter(); r = L.zjni_decompress_batch(cp, cs, bp, ss, res2, n); t3 = time.perf_counter(); assert not L.zjni_isError(r), r
    if it: bc, bd = min(bc, t1 - t0), min(bd, t3 - t2)
ok = all(res2[i] == size for i in range(n)) and bool((back == host).all())
print(json.dumps({"threads": os.environ.get("ZJNI_HOST_THREADS", "default"), "compress_GiBps": n * size / 2**30 / bc, "decompress_GiBps": n * size / 2**30 / bd, "compress_ms": bc * 1e3, "decompress_ms": bd * 1e3, "roundtrip_exact": ok}))
# two batches in flight (zjni_*_batch_begin / zjni_batch_finish, round 5): K batches, never more than two begun and unfinished
if hasattr(L, "zjni_compress_batch_begin") and os.environ.get("E2E_PIPE", "1") != "0":
    comp2 = np.empty(n * bound, dtype=np.uint8); back2 = np.empty(n * size, dtype=np.uint8)
    cp2, bp2 = vp(comp2.ctypes.data, bound), vp(back2.ctypes.data, size)
    rA, rB = (C.c_size_t * n)(), (C.c_size_t * n)(); sets = [(cp, rA), (cp2, rB)]
    def pipe(begin, K):
        jobs = []; t0 = time.perf_counter()
        for k in range(K):
            if len(jobs) == 2: r = L.zjni_batch_finish(jobs.pop(0)); assert not L.zjni_isError(r), r
            j = begin(k & 1); assert j; jobs.append(j)
        for j in jobs: r = L.zjni_batch_finish(j); assert not L.zjni_isError(r), r
        return time.perf_counter() - t0
    K = int(os.environ.get("E2E_BATCHES", "6"))
    cb = lambda w: L.zjni_compress_batch_begin(sp, ss, sets[w][0], cc, sets[w][1], n, 3, 0)
    pipe(cb, 2); tc = pipe(cb, K)
    same = all(rA[i] == res[i] and rB[i] == res[i] for i in range(n)) and bool((comp2 == comp).all())
    rd = [(C.c_size_t * n)(), (C.c_size_t * n)()]; outs = [bp, bp2]
    db = lambda w: L.zjni_decompress_batch_begin(sets[w][0], cs, outs[w], ss, rd[w], n)
    pipe(db, 2); td = pipe(db, K)
    okp = all(rd[0][i] == size and rd[1][i] == size for i in range(n)) and bool((back2 == host).all())
    print(json.dumps({"two_batches_in_flight": {"batches": K, "compress_GiBps": K * n * size / 2**30 / tc, "decompress_GiBps": K * n * size / 2**30 / td,
                                                "compress_ms_per_batch": tc / K * 1e3, "decompress_ms_per_batch": td / K * 1e3, "same_frames": same, "roundtrip_exact": okp}}))
