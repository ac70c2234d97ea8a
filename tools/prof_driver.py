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
    wp = np.zeros(4 * 2048, dtype=np.uint64); L.zjni_debug_wave_profile(wp.ctypes.data_as(vp))
    np.save(os.path.join(ROOT, "gpurun_out", "waveprof_%s.npy" % os.environ.get("AB_TAG", "x")), wp)
    w = wp.reshape(-1, 4)[:min(1024, (n + 63) // 64)]; t0 = w[:, 3].astype(np.int64); late = (t0 - t0.min()) / 100.0        # microseconds after the first wave
    sys.stderr.write("match waves: %d; started > 100 us after the first: %d, > 1 ms: %d, > 10 ms: %d (latest %.1f ms); lane 0 never got a frame in %d waves\n"
                     % (len(w), int((late > 100).sum()), int((late > 1000).sum()), int((late > 10000).sum()), late.max() / 1000.0, int((w[:, 2] <= 1).sum())))
    if (late > 1000).any():
        import collections
        xcc = ((w[:, 1] >> np.uint64(32)) & np.uint64(0xF)).astype(int); lt = late > 1000
        sys.stderr.write("  late waves by XCC %s; all waves by XCC %s; late start times (ms) 10/50/90 %% = %s; blockIdx of late waves %% 8 = %s\n" % (sorted(collections.Counter(xcc[lt].tolist()).items()), sorted(collections.Counter(xcc.tolist()).items()),
                         [round(float(np.quantile(late[lt], q)) / 1000.0, 1) for q in (0.1, 0.5, 0.9)], sorted(collections.Counter((np.nonzero(lt)[0] % 8).tolist()).items())))
if hasattr(L, "zjni_debug_frame_rounds") and n <= 131072:      # -DZL_PROFILE builds: the round in which each list entry's lane finished -> lanes still alive over the launch, by class
    fr = np.zeros(131072, dtype=np.uint32); L.zjni_debug_frame_rounds(fr.ctypes.data_as(vp))
    fr = fr[:n]; total = int(fr.max()) or 1
    sys.stderr.write("frame finish rounds: launch %d rounds; lanes alive after 0/10/25/50/60/70/80/90/95 %% of the launch = %s\n" % (total, [int((fr > total * f).sum()) for f in (0.0, 0.1, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95)]))
    for c in range(4):            # zj_synth.h: index & 3 = 0 text, 1 JSON-like, 2 low-entropy, 3 random
        x = fr[c::4]; sys.stderr.write("  class %d: finish round min / 10 / 50 / 90 / 99 %% / max = %s\n" % (c, [int(x.min())] + [int(np.quantile(x, q)) for q in (0.1, 0.5, 0.9, 0.99)] + [int(x.max())]))
    np.save(os.path.join(ROOT, "gpurun_out", "framerounds_%s.npy" % os.environ.get("AB_TAG", "x")), fr)
import hashlib
fp = hashlib.sha1(h_csz.tobytes()).hexdigest()[:12]
L.zjni_build_stamp.restype = C.c_char_p; L.zjni_route_kernel.restype = C.c_char_p; L.zjni_route_kernel.argtypes = [C.c_int]
route = int(L.zjni_last_route())
print(json.dumps({"tag": os.environ.get("AB_TAG", ""), "csz_sha": fp, "build_stamp": L.zjni_build_stamp().decode(), "route": route, "match_kernel": L.zjni_route_kernel(route).decode(), "entropy_kcycles_per_frame": enc_phase, "stages_ms": stages, "n": n, "size": size, "level": level, "hashLog": hl, "chainLog": cl, "steps": steps, "compress_ms": tc / steps, "pack_ms": tp / steps, "decode_ms": td / steps,
                  "compressed_bytes": int(h_csz.sum()), "all_decoded": bool((h_dsz == size).all())}))
