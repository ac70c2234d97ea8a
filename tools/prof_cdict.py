"""Profiling driver without torch for the dictionary path (BASELINE config 4 shape): n x 4 KiB mixed-entropy buffers,
ZstdDictCompress level `level`, compress -> pack -> decompress with the dictionary through the C-ABI only.
usage: prof_cdict.py [n] [level] [steps] [cls] [bench]   cls = 0..3: only that class of the generator (1 = JSON), default mixed; "bench": the dictionary bench.py --config 4 trains"""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as e
zj = e.load_package()
if os.environ.get('ZJNI_LIB'): zj.LIB_PATH = os.environ['ZJNI_LIB']      # an experimental build of the library
L = zj.lib()
from oracle import ref
hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p
def chk(r): assert r == 0, r
def dmalloc(n):
    p = vp(); chk(hip.hipMalloc(C.byref(p), C.c_size_t(n))); return p
def upload(arr):
    p = dmalloc(arr.nbytes); chk(hip.hipMemcpy(p, arr.ctypes.data_as(vp), C.c_size_t(arr.nbytes), 1)); return p
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cls = int(sys.argv[4]) if len(sys.argv) > 4 else -1
size = 4096
assert L.zjni_init(0) == 0
if len(sys.argv) > 5 and sys.argv[5] == "bench":      # bench.py --config 4's own dictionary: 110 KiB trained on 10 000 JSON-like records (its data: cls = 1)
    host = zj.synth_host(size, 1 << 24, 40000)
    dic = ref.train_dict([host[i * size:(i + 1) * size] for i in range(1, 40000, 4)], 112640)
else:
    host = zj.synth_host(size, 1 << 20, 16000)
    dic = ref.train_dict([host[i * size:(i + 1) * size] for i in range(16000) if cls < 0 or (i & 3) == cls][:4000], 112640)
cd = L.zjni_createCDict(dic, len(dic), level); dd = L.zjni_createDDict(dic, len(dic))
assert cd and dd
bound = L.zjni_compressBound(size)
src = dmalloc(n * size); comp = dmalloc(n * bound); packed = dmalloc(n * bound); back = dmalloc(n * size)
soff = upload(np.arange(n + 1, dtype=np.uint64) * size); coff = upload(np.arange(n + 1, dtype=np.uint64) * bound)
csz = dmalloc(n * 8); dsz = dmalloc(n * 8); poff = dmalloc((n + 1) * 8)
if cls < 0:
    chk(L.zjni_synth_fill_device(src, size, 0, n, None))
else:                                                       # one class: generate 4n buffers, keep every fourth
    big = dmalloc(4 * n * size); chk(L.zjni_synth_fill_device(big, size, 0, 4 * n, None))
    pick = upload((np.arange(n + 1, dtype=np.uint64) * 4 + cls) * size); sizes = upload(np.full(n, size, dtype=np.uint64))
    chk(L.zjni_pack_batch_device(big, pick, sizes, src, soff, n, None))
chk(hip.hipDeviceSynchronize())
ev = [vp() for _ in range(4)]
for x in ev: chk(hip.hipEventCreate(C.byref(x)))
tc = td = 0.0
h_csz = np.zeros(n, dtype=np.uint64)
PROF = bool(os.environ.get("ZJNI_PROFILE")) and hasattr(L, "zjni_debug_read_profile")
enc_phase = None
def read_prof():
    a = (C.c_ulonglong * 32)(); L.zjni_debug_read_profile.argtypes = [vp]; assert L.zjni_debug_read_profile(a) == 0; return list(a)
for it in range(steps + 1):
    if PROF and it == steps: read_prof()
    chk(hip.hipEventRecord(ev[0], None)); chk(L.zjni_compress_batch_device_usingCDict(src, soff, comp, coff, csz, n, cd, 0, None)); chk(hip.hipEventRecord(ev[1], None))
    chk(hip.hipDeviceSynchronize())
    if PROF and it == steps:
        pe = read_prof(); names = ["params+zero", "match find(l0)", "lit gather+codes", "hist+huf build", "huf encode", "seq tables", "seq encode(l0)", "block place"]
        enc_phase = {names[i]: round(pe[16 + i] / n / 1e3, 1) for i in range(8)}
    chk(hip.hipMemcpy(h_csz.ctypes.data_as(vp), csz, C.c_size_t(n * 8), 2))
    h_poff = np.zeros(n + 1, dtype=np.uint64); h_poff[1:] = np.cumsum(h_csz)
    chk(hip.hipMemcpy(poff, h_poff.ctypes.data_as(vp), C.c_size_t((n + 1) * 8), 1))
    chk(L.zjni_pack_batch_device(comp, coff, csz, packed, poff, n, None)); chk(hip.hipEventRecord(ev[2], None))
    chk(L.zjni_decompress_batch_device_usingDDict(packed, poff, back, soff, dsz, n, dd, None)); chk(hip.hipEventRecord(ev[3], None))
    chk(hip.hipDeviceSynchronize())
    ms = C.c_float()
    chk(hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1])); c_ms = ms.value
    chk(hip.hipEventElapsedTime(C.byref(ms), ev[2], ev[3])); d_ms = ms.value
    if it > 0: tc += c_ms; td += d_ms
h_dsz = np.zeros(n, dtype=np.uint64); chk(hip.hipMemcpy(h_dsz.ctypes.data_as(vp), dsz, C.c_size_t(n * 8), 2))
hb = np.zeros(n * size, dtype=np.uint8); chk(hip.hipMemcpy(hb.ctypes.data_as(vp), back, C.c_size_t(n * size), 2))
hs = np.zeros(n * size, dtype=np.uint8); chk(hip.hipMemcpy(hs.ctypes.data_as(vp), src, C.c_size_t(n * size), 2))
GiB = n * size / 2.0**30
L.zjni_build_stamp.restype = C.c_char_p
print(json.dumps({"tag": os.environ.get("AB_TAG", ""), "build_stamp": L.zjni_build_stamp().decode(), "entropy_kcycles_per_frame": enc_phase, "n": n, "size": size, "level": level, "steps": steps, "compress_ms": tc / steps, "decompress_ms": td / steps,
                  "compress_GiBps": GiB / (tc / steps / 1e3), "decompress_GiBps": GiB / (td / steps / 1e3),
                  "ratio": n * size / float(h_csz.sum()), "round_trip": bool((h_dsz == size).all() and (hb == hs).all())}))
