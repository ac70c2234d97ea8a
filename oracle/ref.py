"""ctypes view of oracle/_ref/libzstd_ref.so = the reference's vendored libzstd 1.5.7
(/root/reference/src/main/native/zstd.h).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libzstd_ref.so")
_lib = None


def available():
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"{PATH} missing: run `make -C oracle ref` where /root/reference exists")
        L = C.CDLL(PATH)
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
        _lib = L
    return _lib


ZSTD_c_compressionLevel = 100
ZSTD_c_hashLog = 102
ZSTD_c_chainLog = 103
ZSTD_c_checksumFlag = 201
ZSTD_c_contentSizeFlag = 200


class ZstdRefError(RuntimeError):
    pass


def _check(r):
    L = lib()
    if L.ZSTD_isError(r):
        raise ZstdRefError(L.ZSTD_getErrorName(r).decode())
    return r


def compress(data: bytes, level: int = 3, checksum: bool = False, hash_log: int = 0, chain_log: int = 0) -> bytes:
    """ZSTD_compress2 with the parameters zstd-jni's ZstdCompressCtx sets
    (reference src/main/native/jni_fast_zstd.c:606-607)."""
    L = lib()
    cctx = L.ZSTD_createCCtx()
    try:
        _check(L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, level))
        _check(L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_checksumFlag, int(checksum)))
        if hash_log:
            _check(L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_hashLog, hash_log))      # ZstdCompressCtx.setHashLog
        if chain_log:
            _check(L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_chainLog, chain_log))    # ZstdCompressCtx.setChainLog
        cap = L.ZSTD_compressBound(len(data))
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


def frame_content_size(frame: bytes) -> int:
    return lib().ZSTD_getFrameContentSize(frame, len(frame))


def version() -> str:
    return lib().ZSTD_versionString().decode()
