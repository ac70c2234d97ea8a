"""Profiling driver without torch: the bench step (compress -> pack -> decompress) through the C-ABI
only, HBM buffers from the HIP runtime via ctypes.  Used for rocprofv3 runs (rocprofv3 + torch
segfaults intermittently in this image); prints HIP-event kernel times for cross-checking."""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
zj = e.load_package()
if os.environ.get('ZJNI_LIB'): zj.LIB_PATH = os.environ['ZJNI_LIB']      # an experimental build of the library
L = zj.lib()
hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p
def chk(r): assert r == 0, r
def dmalloc(n):
    p = vp(); chk(hip.hipMalloc(C.byref(p), C.c_size_t(n))); return p
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
size = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
hl = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # ZSTD_c_hashLog / ZSTD_c_chainLog (level 3), 0 = the library's choice
cl = int(sys.argv[6]) if len(sys.argv) > 6 else 0
assert L.zjni_init(0) == 0
bound = L.zjni_compressBound(size)
src = dmalloc(n * size); comp = dmalloc(n * bound); packed = dmalloc(n * bound); back = dmalloc(n * size)
import numpy as np
def upload(arr):
    p = dmalloc(arr.nbytes); chk(hip.hipMemcpy(p, arr.ctypes.data_as(vp), C.c_size_t(arr.nbytes), 1)); return p
soff = upload(np.arange(n + 1, dtype=np.uint64) * size); coff = upload(np.arange(n + 1, dtype=np.uint64) * bound)
csz = dmalloc(n * 8); dsz = dmalloc(n * 8); poff = dmalloc((n + 1) * 8)
if os.environ.get("PROF_DATA") == "xml":          # bench.py --config 1's buffers: overlapping slices of the reference's xml fixture
    from oracle import ref
    xml = np.frombuffer(ref.decompress(open(os.path.join(ROOT, "tests", "golden", "xml-1.zst"), "rb").read(), 6_000_000), dtype=np.uint8)
    host = np.empty(n * size, dtype=np.uint8); span = xml.size - size
    for i in range(n):
        o = (i * 4099) % span; host[i * size:(i + 1) * size] = xml[o:o + size]
    chk(hip.hipMemcpy(src, host.ctypes.data_as(vp), C.c_size_t(host.nbytes), 1)); del host
elif os.environ.get("PROF_CLASSES"):               # only some classes of the synthetic set (zj_synth.h: index & 3 = 0 text, 1 JSON-like, 2 low-entropy, 3 random): "01", "2", ...
    cls = os.environ["PROF_CLASSES"]; first, width = int(cls[0]), len(cls)          # consecutive classes first .. first + width - 1
    assert n % width == 0
    tmp = dmalloc(4 * (n // width) * size); chk(L.zjni_synth_fill_device(tmp, size, 0, 4 * (n // width), None)); chk(hip.hipDeviceSynchronize())
    chk(hip.hipMemcpy2D(src, C.c_size_t(width * size), vp(tmp.value + first * size), C.c_size_t(4 * size), C.c_size_t(width * size), C.c_size_t(n // width), 3))
    chk(hip.hipDeviceSynchronize()); chk(hip.hipFree(tmp))
else:
    chk(L.zjni_synth_fill_device(src, size, 0, n, None))
chk(hip.hipDeviceSynchronize())
ev = [vp() for _ in range(5)]
for x in ev: chk(hip.hipEventCreate(C.byref(x)))
tc = td = tp = 0.0
h_csz = np.zeros(n, dtype=np.uint64)
PROF = bool(os.environ.get("ZJNI_PROFILE")) and hasattr(L, "zjni_debug_read_profile")
enc_phase = None
def read_prof():
    a = (C.c_ulonglong * 32)(); L.zjni_debug_read_profile.argtypes = [vp]; assert L.zjni_debug_read_profile(a) == 0; return list(a)
for it in range(steps + 1):
    if PROF and it == steps: read_prof()                     # (clears the counters: the last step's alone)
    chk(hip.hipEventRecord(ev[0], None)); chk(L.zjni_compress_batch_device_advanced(src, soff, comp, coff, csz, n, level, 0, hl, cl, None)); chk(hip.hipEventRecord(ev[4], None))
    chk(hip.hipDeviceSynchronize())
    ms = C.c_float(); chk(hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[4]))
    if it > 0: tc += ms.value
    if PROF and it == steps:
        pe = read_prof(); names = ["params+zero", "match find(l0)", "lit gather+codes", "hist+huf build", "huf encode", "seq tables", "seq encode(l0)", "block place"]
        enc_phase = {names[i]: round(pe[16 + i] / n / 1e3, 1) for i in range(8)}          # kilo-cycles per frame, entropy kernel's lane 0
    chk(hip.hipMemcpy(h_csz.ctypes.data_as(vp), csz, C.c_size_t(n * 8), 2))
    h_poff = np.zeros(n + 1, dtype=np.uint64); h_poff[1:] = np.cumsum(h_csz)
    chk(hip.hipMemcpy(poff, h_poff.ctypes.data_as(vp), C.c_size_t((n + 1) * 8), 1))
    chk(hip.hipEventRecord(ev[1], None)); chk(L.zjni_pack_batch_device(comp, coff, csz, packed, poff, n, None)); chk(hip.hipEventRecord(ev[2], None))
    chk(L.zjni_decompress_batch_device(packed, poff, back, soff, dsz, n, None)); chk(hip.hipEventRecord(ev[3], None))
    chk(hip.hipDeviceSynchronize())
    ms = C.c_float()
    if it > 0:
        # compress time measured separately above (ev0..first ev1 overwritten): re-measure via events 
        pass
    chk(hip.hipEventElapsedTime(C.byref(ms), ev[1], ev[2])); p_ms = ms.value
    chk(hip.hipEventElapsedTime(C.byref(ms), ev[2], ev[3])); d_ms = ms.value
    if it > 0: tp += p_ms; td += d_ms
h_dsz = np.zeros(n, dtype=np.uint64); chk(hip.hipMemcpy(h_dsz.ctypes.data_as(vp), dsz, C.c_size_t(n * 8), 2))
t8 = (C.c_float * 8)(); L.zjni_last_timing2(t8)
stages = dict(zip(("match", "dec_prep", "dec_seq", "dec_exec", "dec_fused", "match_wide"), [round(float(x), 3) for x in t8][:6]))
if hasattr(L, "zjni_debug_wave_profile"):
    wp = np.zeros(3 * 2048, dtype=np.uint64); L.zjni_debug_wave_profile(wp.ctypes.data_as(vp))
    np.save(os.path.join(ROOT, "gpurun_out", "waveprof_%s.npy" % os.environ.get("AB_TAG", "x")), wp)
import hashlib
fp = hashlib.sha1(h_csz.tobytes()).hexdigest()[:12]
L.zjni_build_stamp.restype = C.c_char_p; L.zjni_route_kernel.restype = C.c_char_p; L.zjni_route_kernel.argtypes = [C.c_int]
route = int(L.zjni_last_route())
print(json.dumps({"tag": os.environ.get("AB_TAG", ""), "csz_sha": fp, "build_stamp": L.zjni_build_stamp().decode(), "route": route, "match_kernel": L.zjni_route_kernel(route).decode(), "entropy_kcycles_per_frame": enc_phase, "stages_ms": stages, "n": n, "size": size, "level": level, "hashLog": hl, "chainLog": cl, "steps": steps, "compress_ms": tc / steps, "pack_ms": tp / steps, "decode_ms": td / steps,
                  "compressed_bytes": int(h_csz.sum()), "all_decoded": bool((h_dsz == size).all())}))
