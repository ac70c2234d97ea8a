"""Torch-free GPU check of the corrupted-frame contract (tests/test_corrupt_frames.py) through the C-ABI with host buffers:
both decode pipelines, plus a batch of valid frames beside them.  TEST INFRASTRUCTURE (uses tests/golden only).
usage: python tools/check_corrupt_gpu.py"""
import hashlib, json, os, sys
os.environ.setdefault("ZJNI_DEBUG_LIVE_SWITCHES", "1")    # the library caches its ZJNI_* switches per process (zj_env); this tool flips ZJNI_DSPLIT_MIN between calls
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
L = C.CDLL(os.path.join(ROOT, "zstd-jni_amd", "lib", "libzjni_amd.so"))     # the product library itself; no torch, no package import
sz, vp = C.c_size_t, C.c_void_p
L.zjni_init.argtypes = [C.c_int]
L.zjni_isError.restype = C.c_uint; L.zjni_isError.argtypes = [sz]
L.zjni_getErrorCode.restype = C.c_int; L.zjni_getErrorCode.argtypes = [sz]
L.zjni_decompress_batch.restype = sz
L.zjni_decompress_batch.argtypes = [C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), sz]
assert L.zjni_init(0) == 0


class Refused(Exception):
    def __init__(self, code): self.code = code
    def getErrorCode(self): return self.code


def decompress_batch(frames, caps):
    n = len(frames)
    keep = [C.create_string_buffer(f, len(f)) for f in frames]; outs = [C.create_string_buffer(max(c, 1)) for c in caps]
    sp = (vp * n)(*[C.addressof(k) for k in keep]); dp = (vp * n)(*[C.addressof(o) for o in outs])
    ss = (sz * n)(*[len(f) for f in frames]); dc = (sz * n)(*caps); res = (sz * n)()
    r = L.zjni_decompress_batch(sp, ss, dp, dc, res, n)
    assert not L.zjni_isError(r), r
    return [Refused(L.zjni_getErrorCode(res[i])) if L.zjni_isError(res[i]) else outs[i].raw[:res[i]] for i in range(n)]


d = os.path.join(ROOT, "tests", "golden", "corrupt")
man = json.load(open(os.path.join(d, "manifest.json")))
CODE = {"Data corruption detected": 20, "Src size is incorrect": 72, "Destination buffer is too small": 70}
names = sorted(man)
frames = [open(os.path.join(d, n), "rb").read() for n in names]
good = open(os.path.join(ROOT, "tests", "golden", "xmlsmall-sized.zst"), "rb").read()
bad = 0
for split_min in ("1", "1000000000"):
    os.environ["ZJNI_DSPLIT_MIN"] = split_min
    outs = decompress_batch(frames * 4 + [good], [man[n]["capacity"] for n in names] * 4 + [102])
    assert outs[-1] == open(os.path.join(ROOT, "tests", "golden", "xmlsmall"), "rb").read()
    for n, o in zip(names * 4, outs):
        want = man[n]["portable"]
        if "error" in want:
            ok = isinstance(o, Exception) and abs(o.getErrorCode()) == CODE[want["error"]]
        else:
            ok = (not isinstance(o, Exception)) and hashlib.sha256(o).hexdigest() == want["sha256"]
        if not ok:
            bad += 1; print("MISMATCH", split_min, n, o if isinstance(o, Exception) else len(o), want)
print("corrupt-frame contract on the GPU:", "OK" if not bad else f"{bad} mismatches")
sys.exit(1 if bad else 0)
