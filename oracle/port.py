"""ctypes view of oracle/libzstd_oracle.so = our plain-C restatement (oracle/zstd_oracle_*.c).
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "libzstd_oracle.so")
_lib = None
ERR_MAX = 120


class ZstdOracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error code {code}")
        self.code = code


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "port"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            build()
        L = C.CDLL(PATH)
        for name in ("zso_decompress",):
            f = getattr(L, name)
            f.restype = C.c_size_t
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.zso_compress.restype = C.c_size_t
        L.zso_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        L.zso_compress_ex.restype = C.c_size_t
        L.zso_compress_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.zso_compress_bound.restype = C.c_size_t
        L.zso_compress_bound.argtypes = [C.c_size_t]
        L.zso_frame_content_size.restype = C.c_ulonglong
        L.zso_frame_content_size.argtypes = [C.c_void_p, C.c_size_t]
        L.zso_find_frame_compressed_size.restype = C.c_size_t
        L.zso_find_frame_compressed_size.argtypes = [C.c_void_p, C.c_size_t]
        L.zso_xxh64.restype = C.c_uint64
        L.zso_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        _lib = L
    return _lib


def _check(r):
    neg = (1 << 64) - r
    if 0 < neg <= ERR_MAX:
        raise ZstdOracleError(neg)
    return r


def decompress(frame: bytes, cap: int) -> bytes:
    dst = C.create_string_buffer(max(cap, 1))
    r = _check(lib().zso_decompress(dst, cap, frame, len(frame)))
    return dst.raw[:r]


def compress(data: bytes, level: int = 3, checksum: bool = False, hash_log: int = 0, chain_log: int = 0) -> bytes:
    L = lib()
    cap = L.zso_compress_bound(len(data))
    dst = C.create_string_buffer(max(cap, 1))
    r = _check(L.zso_compress_ex(dst, cap, data, len(data), level, int(checksum), hash_log, chain_log))
    return dst.raw[:r]


def frame_content_size(frame: bytes) -> int:
    return lib().zso_frame_content_size(frame, len(frame))


def find_frame_compressed_size(frame: bytes) -> int:
    return _check(lib().zso_find_frame_compressed_size(frame, len(frame)))


def xxh64(data: bytes, seed: int = 0) -> int:
    return lib().zso_xxh64(data, len(data), seed)


# ---- multi-threaded batch helpers (oracle/cpu_baseline.c) — tests / bench input preparation ----
def _batch_fn():
    L = lib()
    L.zso_batch.restype = C.c_int
    L.zso_batch.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.zso_cpu_baseline.restype = C.c_int
    L.zso_cpu_baseline.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    return L


def compress_many(data: bytes, buf_size: int, level: int, threads: int, use_ref=True):
    """Compress len(data)//buf_size equal-size buffers with `threads` CPU threads.
    Returns (packed_bytes, sizes list).  use_ref=True -> oracle/_ref (reference libzstd)."""
    import numpy as np
    from . import ref
    L = _batch_fn()
    n = len(data) // buf_size
    bound = L.zso_compress_bound(buf_size)
    src_off = np.arange(n + 1, dtype=np.uint64) * buf_size
    dst_off = np.arange(n + 1, dtype=np.uint64) * bound
    out = np.empty(max(n * bound, 1), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint64)
    path = ref.PATH.encode() if (use_ref and ref.available()) else None
    rc = L.zso_batch(path, 0, level, data, src_off.ctypes.data, out.ctypes.data, dst_off.ctypes.data, sizes.ctypes.data, n, threads)
    if rc != 0:
        raise RuntimeError("zso_batch failed")
    frames = [out[i * bound:i * bound + int(sizes[i])].tobytes() for i in range(n)]
    return frames


def cpu_baseline(data: bytes, buf_size: int, level: int, threads: int, reps: int = 2, use_ref=True):
    """dict(compress_s, decompress_s, compressed_bytes, exact) — see oracle/cpu_baseline.c."""
    from . import ref
    L = _batch_fn()
    n = len(data) // buf_size
    out = (C.c_double * 4)()
    path = ref.PATH.encode() if (use_ref and ref.available()) else None
    rc = L.zso_cpu_baseline(path, data, buf_size, n, level, threads, reps, out)
    if rc != 0:
        raise RuntimeError("zso_cpu_baseline failed")
    return dict(compress_s=out[0], decompress_s=out[1], compressed_bytes=int(out[2]), exact=bool(out[3]),
                kind="reference" if path else "port")


def cpu_baseline2(data, buf_size: int, n: int, level: int, threads: int, min_seconds: float = 1.0, dictionary: bytes = None,
                  hash_log: int = 0, chain_log: int = 0, offsets=None, keep_frames: bool = False):
    """The reference's libzstd on `threads` host threads that exist (with their reused contexts) before the clock starts, released
    by a barrier, repeated until `min_seconds` of timed work are done — oracle/cpu_baseline.c zso_cpu_baseline2.
    `data`: bytes or a numpy uint8 array (n buffers of buf_size bytes, or `offsets` = n+1 byte offsets).  Returns
    dict(compress_s, decompress_s, compressed_bytes, exact, passes, mean_compress_s, mean_decompress_s) — with keep_frames also
    frames (numpy uint8: the reference's frames back to back) and sizes (numpy uint64[n])."""
    import numpy as np
    from . import ref
    L = _batch_fn()
    L.zso_cpu_baseline3.restype = C.c_int
    L.zso_cpu_baseline3.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                    C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.c_void_p, C.c_size_t, C.c_void_p]
    if not ref.available():
        raise RuntimeError("oracle/_ref/libzstd_ref.so is not built")
    arr = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
    out = (C.c_double * 8)()
    keep = sizes = None
    if keep_frames:                                        # room for every frame at its bound (never more than that)
        total_src = int(off[-1]) if off is not None else n * buf_size
        keep = np.empty(total_src + (total_src >> 7) + 1024 * n + 64, dtype=np.uint8); sizes = np.zeros(n, dtype=np.uint64)
    rc = L.zso_cpu_baseline3(ref.PATH.encode(), arr.ctypes.data, None if off is None else off.ctypes.data, buf_size, n, level, hash_log, chain_log,
                             threads, min_seconds, dictionary, len(dictionary) if dictionary else 0, out,
                             keep.ctypes.data if keep is not None else None, keep.size if keep is not None else 0, sizes.ctypes.data if sizes is not None else None)
    if rc != 0:
        raise RuntimeError("zso_cpu_baseline3 failed")
    r = dict(compress_s=out[0], decompress_s=out[1], compressed_bytes=int(out[2]), exact=bool(out[3]), passes=(int(out[4]), int(out[5])),
             mean_compress_s=out[6], mean_decompress_s=out[7], kind="reference")
    if keep_frames:
        r["frames"] = keep[:int(out[2])] if int(out[2]) <= keep.size else None
        r["sizes"] = sizes
    return r


def compress_many_packed(data, buf_size: int, level: int, threads: int):
    """The reference's ZSTD_compress2(level) over len(data)//buf_size equal-size buffers with `threads` host threads; `data` is bytes or a
    numpy uint8 array.  Returns (packed uint8 array: the frames back to back, sizes uint64[n])."""
    import numpy as np
    from . import ref
    L = _batch_fn()
    arr = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else np.ascontiguousarray(data)
    n = arr.size // buf_size
    bound = L.zso_compress_bound(buf_size)
    src_off = np.arange(n + 1, dtype=np.uint64) * buf_size
    dst_off = np.arange(n + 1, dtype=np.uint64) * bound
    out = np.empty(max(n * bound, 1), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint64)
    if not ref.available():
        raise RuntimeError("oracle/_ref/libzstd_ref.so is not built")
    rc = L.zso_batch(ref.PATH.encode(), 0, level, arr.ctypes.data, src_off.ctypes.data, out.ctypes.data, dst_off.ctypes.data, sizes.ctypes.data, n, threads)
    if rc != 0:
        raise RuntimeError("zso_batch failed")
    packed = np.empty(int(sizes.sum()) + 16, dtype=np.uint8)
    pos = 0
    for i in range(n):
        k = int(sizes[i]); packed[pos:pos + k] = out[i * bound:i * bound + k]; pos += k
    return packed, sizes
