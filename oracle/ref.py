"""ctypes view of oracle/_ref/libzstd_ref.so = the reference's vendored libzstd 1.5.7
(/root/reference/src/main/native/zstd.h).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libzstd_ref.so")
_lib = None


def available():
    return os.path.exists(PATH)


def _dl_mode():
    """RTLD_DEEPBIND so that the library binds its own ZSTD_* whatever libzstd the process already holds — except under a
    sanitizer runtime (tests/test_emu_sanitizer.py preloads libasan), which refuses to dlopen with that flag."""
    if "asan" in os.environ.get("LD_PRELOAD", ""):
        return os.RTLD_NOW
    return os.RTLD_NOW | getattr(os, "RTLD_DEEPBIND", 0)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"{PATH} missing: run `make -C oracle ref` where /root/reference exists")
        L = C.CDLL(PATH, mode=_dl_mode())     # its own ZSTD_* first: a profiler (rocprofv3) may have a system libzstd loaded globally
        L.ZSTD_compressBound.restype = C.c_size_t
        L.ZSTD_compressBound.argtypes = [C.c_size_t]
        L.ZSTD_isError.restype = C.c_uint
        L.ZSTD_isError.argtypes = [C.c_size_t]
        L.ZSTD_getErrorName.restype = C.c_char_p
        L.ZSTD_getErrorName.argtypes = [C.c_size_t]
        L.ZSTD_createCCtx.restype = C.c_void_p
        L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        L.ZSTD_CCtx_setParameter.restype = C.c_size_t
        L.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ZSTD_compress2.restype = C.c_size_t
        L.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_createDCtx.restype = C.c_void_p
        L.ZSTD_freeDCtx.argtypes = [C.c_void_p]
        L.ZSTD_decompressDCtx.restype = C.c_size_t
        L.ZSTD_decompressDCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_getFrameContentSize.restype = C.c_ulonglong
        L.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_findFrameCompressedSize.restype = C.c_size_t
        L.ZSTD_findFrameCompressedSize.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_versionString.restype = C.c_char_p
        # dictionary entry points (reference N/jni_fast_zstd.c:133-244 -> ZSTD_compress_usingDict / ZSTD_decompress_usingDict,
        # N/jni_zdict.c -> ZDICT_trainFromBuffer)
        L.ZSTD_compress_usingDict.restype = C.c_size_t
        L.ZSTD_compress_usingDict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.ZSTD_decompress_usingDict.restype = C.c_size_t
        L.ZSTD_decompress_usingDict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        # ZstdDictCompress (reference N/jni_fast_zstd.c:26,47-49) and its two users: ZstdCompressCtx.loadDict ->
        # ZSTD_CCtx_refCDict (:335) + ZSTD_compress2, and Zstd.compress(dst, src, ZstdDictCompress) -> ZSTD_compress_usingCDict (:191,:216)
        L.ZSTD_createCDict.restype = C.c_void_p
        L.ZSTD_createCDict.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        L.ZSTD_freeCDict.argtypes = [C.c_void_p]
        L.ZSTD_CCtx_refCDict.restype = C.c_size_t
        L.ZSTD_CCtx_refCDict.argtypes = [C.c_void_p, C.c_void_p]
        L.ZSTD_compress_usingCDict.restype = C.c_size_t
        L.ZSTD_compress_usingCDict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ZDICT_trainFromBuffer.restype = C.c_size_t
        L.ZDICT_trainFromBuffer.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_uint]
        L.ZDICT_isError.restype = C.c_uint
        L.ZDICT_isError.argtypes = [C.c_size_t]
        L.ZDICT_getDictID.restype = C.c_uint
        L.ZDICT_getDictID.argtypes = [C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


ZSTD_c_compressionLevel = 100
ZSTD_c_hashLog = 102
ZSTD_c_chainLog = 103
ZSTD_c_checksumFlag = 201
ZSTD_c_contentSizeFlag = 200


class ZstdRefError(RuntimeError):
    """message = ZSTD_getErrorName, .code = ZSTD_getErrorCode (N/zstd_errors.h)"""
    def __init__(self, msg, code=0):
        super().__init__(msg)
        self.code = code


def _check(r):
    L = lib()
    if L.ZSTD_isError(r):
        raise ZstdRefError(L.ZSTD_getErrorName(r).decode(), (1 << 64) - r)
    return r


def compress(data: bytes, level: int = 3, checksum: bool = False, hash_log: int = 0, chain_log: int = 0, content_size: bool = True, cap: int = None) -> bytes:
    """ZSTD_compress2 with the parameters zstd-jni's ZstdCompressCtx sets
    (reference src/main/native/jni_fast_zstd.c:606-607)."""
    L = lib()
    cctx = L.ZSTD_createCCtx()
    try:
        _check(L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, level))
        _check(L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_checksumFlag, int(checksum)))
        if not content_size:
            _check(L.ZSTD_CCtx_setParameter(cctx, 200, 0))                        # ZSTD_c_contentSizeFlag: ZstdCompressCtx.setContentSize(false)
        if hash_log:
            _check(L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_hashLog, hash_log))      # ZstdCompressCtx.setHashLog
        if chain_log:
            _check(L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_chainLog, chain_log))    # ZstdCompressCtx.setChainLog
        if cap is None:
            cap = L.ZSTD_compressBound(len(data))                                 # (a smaller `cap`: the destination the caller's buffer offers)
        dst = C.create_string_buffer(max(cap, 1))
        r = _check(L.ZSTD_compress2(cctx, dst, cap, data, len(data)))
        return dst.raw[:r]
    finally:
        L.ZSTD_freeCCtx(cctx)


def decompress(frame: bytes, cap: int) -> bytes:
    L = lib()
    dctx = L.ZSTD_createDCtx()
    try:
        dst = C.create_string_buffer(max(cap, 1))
        r = _check(L.ZSTD_decompressDCtx(dctx, dst, cap, frame, len(frame)))
        return dst.raw[:r]
    finally:
        L.ZSTD_freeDCtx(dctx)


class _Buf(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


def compress_stream(data: bytes, level: int = 3, checksum: bool = False, chunk: int = 50000, flush_every: int = 0) -> bytes:
    """ZSTD_compressStream2 without a pledged size — what ZstdOutputStream / ZstdDirectBufferCompressingStream produce
    (reference N/jni_outputstream_zstd.c): the frame header carries NO content size; ZSTD_e_flush every `flush_every` chunks
    closes blocks early (many small blocks)."""
    L = lib()
    L.ZSTD_compressStream2.restype = C.c_size_t
    L.ZSTD_compressStream2.argtypes = [C.c_void_p, C.POINTER(_Buf), C.POINTER(_Buf), C.c_int]
    cctx = L.ZSTD_createCCtx()
    try:
        _check(L.ZSTD_CCtx_setParameter(cctx, 100, level))              # ZSTD_c_compressionLevel
        _check(L.ZSTD_CCtx_setParameter(cctx, 201, 1 if checksum else 0))  # ZSTD_c_checksumFlag
        cap = L.ZSTD_compressBound(len(data)) + 1024 + 16 * (len(data) // max(chunk, 1) + 1)
        dst = C.create_string_buffer(cap)
        src = C.create_string_buffer(data, max(len(data), 1))
        ob = _Buf(C.addressof(dst), cap, 0)
        pos, k = 0, 0
        while pos < len(data):
            n = min(chunk, len(data) - pos)
            ib = _Buf(C.addressof(src) + pos, n, 0)
            k += 1
            mode = 1 if (flush_every and k % flush_every == 0) else 0      # ZSTD_e_flush / ZSTD_e_continue
            while True:
                r = _check(L.ZSTD_compressStream2(cctx, C.byref(ob), C.byref(ib), mode))
                if ib.pos == ib.size and (mode == 0 or r == 0):
                    break
            pos += n
        ib = _Buf(C.addressof(src), 0, 0)
        while _check(L.ZSTD_compressStream2(cctx, C.byref(ob), C.byref(ib), 2)) != 0:   # ZSTD_e_end
            pass
        return dst.raw[:ob.pos]
    finally:
        L.ZSTD_freeCCtx(cctx)


PORTABLE_PATH = os.path.join(_HERE, "_ref", "libzstd_ref_portable.so")
_portable = None


def decompress_portable(frame: bytes, cap: int, dictionary: bytes = None) -> bytes:
    """ZSTD_decompress[_usingDict] of the reference built with its own HUF_DISABLE_FAST_DECODE switch (`make -C oracle refportable`,
    N/decompress/huf_decompress.c:37): the decoder every platform without the 64-bit fast Huffman loops runs.  It differs from
    decompress() only on corrupted frames (the fast loops skip the end-of-stream check, huf_decompress.c:873-888 vs :697)."""
    global _portable
    if _portable is None:
        if not os.path.exists(PORTABLE_PATH):
            raise RuntimeError(f"{PORTABLE_PATH} missing: run `make -C oracle refportable` where /root/reference exists")
        P = C.CDLL(PORTABLE_PATH, mode=_dl_mode())
        P.ZSTD_isError.restype = C.c_uint
        P.ZSTD_isError.argtypes = [C.c_size_t]
        P.ZSTD_getErrorName.restype = C.c_char_p
        P.ZSTD_getErrorName.argtypes = [C.c_size_t]
        P.ZSTD_createDCtx.restype = C.c_void_p
        P.ZSTD_freeDCtx.argtypes = [C.c_void_p]
        P.ZSTD_decompress_usingDict.restype = C.c_size_t
        P.ZSTD_decompress_usingDict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        _portable = P
    P = _portable
    dctx = P.ZSTD_createDCtx()
    try:
        dst = C.create_string_buffer(max(cap, 1))
        r = P.ZSTD_decompress_usingDict(dctx, dst, cap, frame, len(frame), dictionary, len(dictionary) if dictionary else 0)
        if P.ZSTD_isError(r):
            raise ZstdRefError(P.ZSTD_getErrorName(r).decode(), (1 << 64) - r)
        return dst.raw[:r]
    finally:
        P.ZSTD_freeDCtx(dctx)


def compress_using_dict(data: bytes, dictionary: bytes, level: int = 3) -> bytes:
    """ZSTD_compress_usingDict — what Zstd.compressUsingDict / compressFastDict reach (reference N/jni_zstd.c, jni_fast_zstd.c)."""
    L = lib()
    cctx = L.ZSTD_createCCtx()
    try:
        cap = L.ZSTD_compressBound(len(data))
        dst = C.create_string_buffer(max(cap, 1))
        r = _check(L.ZSTD_compress_usingDict(cctx, dst, cap, data, len(data), dictionary, len(dictionary), level))
        return dst.raw[:r]
    finally:
        L.ZSTD_freeCCtx(cctx)


class CDict:
    """ZSTD_createCDict(dict, level) — ZstdDictCompress (reference N/jni_fast_zstd.c:26)."""

    def __init__(self, dictionary: bytes, level: int = 3):
        self._buf = bytes(dictionary)
        self.ptr = lib().ZSTD_createCDict(self._buf, len(self._buf), level)
        if not self.ptr:
            raise ZstdRefError("ZSTD_createCDict failed")

    def close(self):
        if self.ptr and lib is not None:
            lib().ZSTD_freeCDict(self.ptr)
            self.ptr = None

    def __del__(self):
        self.close()

    def compress(self, data: bytes, checksum: bool = False, dict_id: bool = True, cap: int = None) -> bytes:
        """ZstdCompressCtx.loadDict(ZstdDictCompress) + compress: ZSTD_CCtx_refCDict then ZSTD_compress2."""
        L = lib()
        cctx = L.ZSTD_createCCtx()
        try:
            _check(L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_checksumFlag, int(checksum)))
            if not dict_id:
                _check(L.ZSTD_CCtx_setParameter(cctx, 202, 0))                    # ZSTD_c_dictIDFlag: ZstdCompressCtx.setDictID(false)
            _check(L.ZSTD_CCtx_refCDict(cctx, self.ptr))
            if cap is None:
                cap = L.ZSTD_compressBound(len(data))
            dst = C.create_string_buffer(max(cap, 1))
            r = _check(L.ZSTD_compress2(cctx, dst, cap, data, len(data)))
            return dst.raw[:r]
        finally:
            L.ZSTD_freeCCtx(cctx)

    def compress_using(self, data: bytes, cap: int = None) -> bytes:
        """Zstd.compress(dst, src, ZstdDictCompress): ZSTD_compress_usingCDict."""
        L = lib()
        cctx = L.ZSTD_createCCtx()
        try:
            if cap is None:
                cap = L.ZSTD_compressBound(len(data))
            dst = C.create_string_buffer(max(cap, 1))
            r = _check(L.ZSTD_compress_usingCDict(cctx, dst, cap, data, len(data), self.ptr))
            return dst.raw[:r]
        finally:
            L.ZSTD_freeCCtx(cctx)


def decompress_using_dict(frame: bytes, dictionary: bytes, cap: int) -> bytes:
    L = lib()
    dctx = L.ZSTD_createDCtx()
    try:
        dst = C.create_string_buffer(max(cap, 1))
        r = _check(L.ZSTD_decompress_usingDict(dctx, dst, cap, frame, len(frame), dictionary, len(dictionary)))
        return dst.raw[:r]
    finally:
        L.ZSTD_freeDCtx(dctx)


def train_dict(samples, dict_size: int) -> bytes:
    """ZDICT_trainFromBuffer (Zstd.trainFromBuffer, reference N/jni_zdict.c)."""
    L = lib()
    blob = b"".join(samples)
    sizes = (C.c_size_t * len(samples))(*[len(x) for x in samples])
    dst = C.create_string_buffer(dict_size)
    r = L.ZDICT_trainFromBuffer(dst, dict_size, blob, sizes, len(samples))
    if L.ZDICT_isError(r):
        raise ZstdRefError("ZDICT_trainFromBuffer failed")
    return dst.raw[:r]


def dict_id(dictionary: bytes) -> int:
    return lib().ZDICT_getDictID(dictionary, len(dictionary))


def frame_content_size(frame: bytes) -> int:
    return lib().ZSTD_getFrameContentSize(frame, len(frame))


def version() -> str:
    return lib().ZSTD_versionString().decode()
