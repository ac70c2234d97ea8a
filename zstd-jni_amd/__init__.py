"""zstd-jni_amd — MI355X-native batched zstd path behind the com.github.luben.zstd API.

Layout (only what the hot path needs):
  csrc/      HIP kernels (gfx950) + the C-ABI of include/zjni_amd.h  -> lib/libzjni_amd.so
  __init__   loader + host-side mirror of the reference's Java classes for this path
             (Zstd, ZstdCompressCtx, ZstdDecompressCtx, ZstdException: same method names, argument
             meaning and error behaviour as /root/reference/src/main/java/com/github/luben/zstd/*.java),
             written in Python because no JVM/javac exists in this image.
  batch      device-resident batch helpers (torch is used for HBM buffers and streams only)

There is NO CPU fallback: every compute entry raises ZstdException(code 200) when no gfx950 device
or no built library is present.  The CPU oracle lives in /oracle and is never imported from here.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "lib", "libzjni_amd.so")
if os.environ.get("ZJNI_LIB"):          # another build of the same library (tools/build_variant.sh: A/B variants, the tuning build with its experiment knobs) — never another implementation
    LIB_PATH = os.environ["ZJNI_LIB"]
# the kernel file first (the one hipcc is given), then every header beside it: the build stamp and the rebuild test cover them all
SOURCES = [os.path.join(_HERE, "csrc", "zj_kernels.hip")] + sorted(
    os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc")) if f.endswith(".h"))
BLOCKSIZE_MAX = 1 << 17
ERR_NO_DEVICE = 200
ERR_UNSUPPORTED = 201

_lib = None


def build_stamp():
    """What the library is built from: a hash of csrc/ and the header (the GPU box has no .git, and a revision says nothing about
    uncommitted edits): profiles/ and roofline.traffic are stamped with it (zjni_build_stamp)."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(SOURCES + [os.path.join(ROOT, "include", "zjni_amd.h")]):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return "src" + h.hexdigest()[:12]


def build(force=False, verbose=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU) into lib/libzjni_amd.so."""
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(s) for s in SOURCES + [os.path.join(ROOT, "include", "zjni_amd.h")])
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-DZJNI_BUILD_STAMP=\"%s\"" % build_stamp(), "-o", LIB_PATH, SOURCES[0]]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")

    def _big_stack():
        # the fully inlined kernels make clang's inliner/optimizer recurse deeply; with the default 8 MiB stack
        # the device compilation can die with SIGSEGV, so the compiler runs with the hard limit as soft limit
        import resource
        soft, hard = resource.getrlimit(resource.RLIMIT_STACK)
        try:
            resource.setrlimit(resource.RLIMIT_STACK, (hard, hard))
        except (ValueError, OSError):
            pass
    subprocess.check_call(cmd, preexec_fn=_big_stack)
    return LIB_PATH


def lib():
    """ctypes handle on libzjni_amd.so.  Fails loudly when the HIP library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not built: run __graft_entry__.build() (needs hipcc); "
                           "there is no CPU fallback for the zstd hot path")
    L = C.CDLL(LIB_PATH)
    sz, vp, u64p = C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64)
    L.zjni_version.restype = C.c_char_p
    L.zjni_device_count.restype = C.c_int
    L.zjni_init.restype = C.c_int
    L.zjni_init.argtypes = [C.c_int]
    L.zjni_isError.restype = C.c_uint
    L.zjni_isError.argtypes = [sz]
    L.zjni_getErrorCode.restype = C.c_int
    L.zjni_getErrorCode.argtypes = [sz]
    L.zjni_getErrorName.restype = C.c_char_p
    L.zjni_getErrorName.argtypes = [sz]
    L.zjni_compressBound.restype = sz
    L.zjni_compressBound.argtypes = [sz]
    L.zjni_getFrameContentSize.restype = C.c_ulonglong
    L.zjni_getFrameContentSize.argtypes = [vp, sz]
    for name in ("zjni_decompress_batch_device",):
        f = getattr(L, name)
        f.restype = sz
        f.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    L.zjni_compress_batch_device.restype = sz
    L.zjni_compress_batch_device.argtypes = [vp, vp, vp, vp, vp, sz, C.c_int, vp]
    L.zjni_compress_batch_device2.restype = sz
    L.zjni_compress_batch_device2.argtypes = [vp, vp, vp, vp, vp, sz, C.c_int, C.c_int, vp]
    L.zjni_compress_batch2.restype = sz
    L.zjni_compress_batch2.argtypes = [C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), sz, C.c_int, C.c_int]
    L.zjni_compress2.restype = sz
    L.zjni_compress2.argtypes = [vp, sz, vp, sz, C.c_int, C.c_int]
    L.zjni_compress_batch_begin.restype = vp         # two host batches in flight (include/zjni_amd.h): begin returns a job, finish waits for it
    L.zjni_compress_batch_begin.argtypes = [C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), sz, C.c_int, C.c_int]
    L.zjni_decompress_batch_begin.restype = vp
    L.zjni_decompress_batch_begin.argtypes = [C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), sz]
    L.zjni_batch_finish.restype = sz
    L.zjni_batch_finish.argtypes = [vp]
    L.zjni_createDDict.restype = vp
    L.zjni_createDDict.argtypes = [vp, sz]
    L.zjni_freeDDict.restype = sz
    L.zjni_freeDDict.argtypes = [vp]
    L.zjni_getDictID_fromDDict.restype = C.c_uint
    L.zjni_getDictID_fromDDict.argtypes = [vp]
    L.zjni_decompress_batch_device_usingDDict.restype = sz
    L.zjni_decompress_batch_device_usingDDict.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp]
    L.zjni_decompress_batch_usingDDict.restype = sz
    L.zjni_decompress_batch_usingDDict.argtypes = [C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), sz, vp]
    L.zjni_decompress_usingDDict.restype = sz
    L.zjni_decompress_usingDDict.argtypes = [vp, sz, vp, sz, vp]
    L.zjni_compress_batch_device_advanced.restype = sz
    L.zjni_compress_batch_device_advanced.argtypes = [vp, vp, vp, vp, vp, sz, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.zjni_compress_batch_advanced.restype = sz
    L.zjni_compress_batch_advanced.argtypes = [C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), sz, C.c_int, C.c_int, C.c_int, C.c_int]
    L.zjni_createCDict.restype = vp
    L.zjni_createCDict.argtypes = [vp, sz, C.c_int]
    L.zjni_freeCDict.restype = sz
    L.zjni_freeCDict.argtypes = [vp]
    L.zjni_getDictID_fromCDict.restype = C.c_uint
    L.zjni_getDictID_fromCDict.argtypes = [vp]
    L.zjni_compress_batch_device_usingCDict.restype = sz
    L.zjni_compress_batch_device_usingCDict.argtypes = [vp, vp, vp, vp, vp, sz, vp, C.c_int, vp]
    L.zjni_compress_batch_usingCDict.restype = sz
    L.zjni_compress_batch_usingCDict.argtypes = [C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), sz, vp, C.c_int]
    L.zjni_compress_usingCDict.restype = sz
    L.zjni_compress_usingCDict.argtypes = [vp, sz, vp, sz, vp]
    L.zjni_decompress_batch.restype = sz
    L.zjni_decompress_batch.argtypes = [C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), sz]
    L.zjni_compress_batch.restype = sz
    L.zjni_compress_batch.argtypes = [C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), sz, C.c_int]
    L.zjni_compress.restype = sz
    L.zjni_compress.argtypes = [vp, sz, vp, sz, C.c_int]
    L.zjni_decompress.restype = sz
    L.zjni_decompress.argtypes = [vp, sz, vp, sz]
    L.zjni_synth_fill_host.restype = None
    L.zjni_synth_fill_host.argtypes = [vp, sz, C.c_uint64, sz]
    L.zjni_synth_fill_device.restype = sz
    L.zjni_synth_fill_device.argtypes = [vp, sz, C.c_uint64, sz, vp]
    L.zjni_pack_batch_device.restype = sz
    L.zjni_pack_batch_device.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    L.zjni_pack_batch_device2.restype = sz
    L.zjni_pack_batch_device2.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    L.zjni_last_timing.restype = C.c_int
    L.zjni_last_timing.argtypes = [C.POINTER(C.c_float)]
    L.zjni_last_timing2.restype = C.c_int
    L.zjni_last_timing2.argtypes = [C.POINTER(C.c_float)]
    L.zjni_compress_batch_multi.restype = sz
    L.zjni_compress_batch_multi.argtypes = [vp, vp, vp, vp, vp, sz, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    L.zjni_decompress_batch_multi.restype = sz
    L.zjni_decompress_batch_multi.argtypes = [vp, vp, vp, vp, vp, sz, vp, C.c_int]
    for fn in ("zjni_set_scratch_limit", "zjni_scratch_bytes", "zjni_release_scratch"):
        getattr(L, fn).restype = sz
    L.zjni_set_scratch_limit.argtypes = [sz]
    L.zjni_createAggregator.restype = vp
    L.zjni_createAggregator.argtypes = [C.c_int, sz, C.c_uint]
    L.zjni_freeAggregator.restype = None
    L.zjni_freeAggregator.argtypes = [vp]
    L.zjni_aggregator_compress.restype = sz
    L.zjni_aggregator_compress.argtypes = [vp, vp, sz, vp, sz, C.c_int, C.c_int]
    L.zjni_aggregator_decompress.restype = sz
    L.zjni_aggregator_decompress.argtypes = [vp, vp, sz, vp, sz]
    L.zjni_aggregator_stats.restype = None
    L.zjni_aggregator_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.zjni_kernel_info.restype = C.c_int
    L.zjni_kernel_info.argtypes = [C.POINTER(C.c_int)] * 4
    L.zjni_shutdown.restype = None
    L.zjni_last_route.restype = C.c_int
    L.zjni_route_kernel.restype = C.c_char_p
    L.zjni_route_kernel.argtypes = [C.c_int]
    L.zjni_build_stamp.restype = C.c_char_p
    L.zjni_last_decode_lists.restype = C.c_int
    L.zjni_last_decode_lists.argtypes = [C.POINTER(C.c_uint)]
    if hasattr(L, "zjni_last_decode_lists2"):            # (absent from older variant libraries loaded through ZJNI_LIB for A/B runs; tests/test_abi.py checks the product library's exports)
        L.zjni_last_decode_lists2.restype = C.c_int
        L.zjni_last_decode_lists2.argtypes = [C.POINTER(C.c_uint)]
    L.zjni_last_lists.restype = C.c_int
    L.zjni_last_lists.argtypes = [C.POINTER(C.c_uint)]
    L.zjni_frame_extent.restype = sz
    L.zjni_frame_extent.argtypes = [vp, sz, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.zjni_compress_stream.restype = sz
    L.zjni_compress_stream.argtypes = [vp, sz, vp, sz, C.c_int, C.c_int, vp, sz, C.c_int, C.c_int]
    L.zjni_compress_stream_batch_device.restype = sz
    L.zjni_compress_stream_batch_device.argtypes = [vp, vp, vp, vp, vp, sz, C.c_int, C.c_int, vp, vp, vp, vp]
    _lib = L
    return L


EXPORTS = ("zjni_version", "zjni_device_count", "zjni_init", "zjni_shutdown", "zjni_isError",
           "zjni_getErrorCode", "zjni_getErrorName", "zjni_compressBound", "zjni_getFrameContentSize",
           "zjni_decompress_batch_device", "zjni_compress_batch_device", "zjni_decompress_batch",
           "zjni_compress_batch", "zjni_compress", "zjni_decompress", "zjni_synth_fill_host",
           "zjni_synth_fill_device", "zjni_kernel_info", "zjni_pack_batch_device", "zjni_last_timing", "zjni_last_timing2",
           "zjni_set_scratch_limit", "zjni_scratch_bytes", "zjni_release_scratch", "zjni_compress_batch_multi", "zjni_decompress_batch_multi",
           "zjni_compress_batch_device2", "zjni_compress_batch2", "zjni_compress2",
           "zjni_createDDict", "zjni_freeDDict", "zjni_getDictID_fromDDict", "zjni_decompress_batch_device_usingDDict",
           "zjni_decompress_batch_usingDDict", "zjni_decompress_usingDDict",
           "zjni_createCDict", "zjni_freeCDict", "zjni_getDictID_fromCDict", "zjni_compress_batch_device_usingCDict",
           "zjni_compress_batch_usingCDict", "zjni_compress_usingCDict",
           "zjni_compress_batch_device_advanced", "zjni_compress_batch_advanced",
           "zjni_createAggregator", "zjni_freeAggregator", "zjni_aggregator_compress", "zjni_aggregator_decompress", "zjni_aggregator_stats",
           "zjni_last_route", "zjni_route_kernel", "zjni_build_stamp", "zjni_compress_stream", "zjni_compress_stream_batch_device", "zjni_frame_extent", "zjni_last_lists", "zjni_last_decode_lists", "zjni_last_decode_lists2",
           "zjni_compress_batch_begin", "zjni_decompress_batch_begin", "zjni_batch_finish", "zjni_pack_batch_device2")


# --------------------------------------------------------------------------- Java API mirror --
class ZstdException(RuntimeError):
    """com.github.luben.zstd.ZstdException (J/ZstdException.java:16-18): code + libzstd's message."""

    def __init__(self, result_or_code, message=None):
        if message is None:
            code = Zstd.getErrorCode(result_or_code)
            message = Zstd.getErrorName(result_or_code)
        else:
            code = result_or_code
        super().__init__(message)
        self.code = code

    def getErrorCode(self):
        return self.code


def _addr(buf, offset=0):
    """(address, keepalive) of a bytes / bytearray / memoryview / ctypes buffer."""
    if isinstance(buf, (bytes,)):
        keep = C.create_string_buffer(buf, len(buf))
        return C.addressof(keep) + offset, keep
    mv = memoryview(buf)
    keep = (C.c_char * mv.nbytes).from_buffer(buf) if not mv.readonly else C.create_string_buffer(bytes(mv), mv.nbytes)
    return C.addressof(keep) + offset, keep


class Zstd:
    """Static helpers of J/Zstd.java for this path (one-shot, no dictionary)."""

    @staticmethod
    def compressBound(srcSize):                                    # J/Zstd.java:914
        return lib().zjni_compressBound(srcSize)

    @staticmethod
    def isError(code):                                             # J/Zstd.java:923
        return bool(lib().zjni_isError(code & 0xFFFFFFFFFFFFFFFF))

    @staticmethod
    def getErrorName(code):                                        # J/Zstd.java:924
        return lib().zjni_getErrorName(code & 0xFFFFFFFFFFFFFFFF).decode()

    @staticmethod
    def getErrorCode(code):                                        # J/Zstd.java:925
        return lib().zjni_getErrorCode(code & 0xFFFFFFFFFFFFFFFF)

    @staticmethod
    def errDstSizeTooSmall():                                      # N/jni_zstd.c:638-667
        return 70

    @staticmethod
    def errSrcSizeWrong():
        return 72

    @staticmethod
    def errCorruptionDetected():
        return 20

    @staticmethod
    def defaultCompressionLevel():                                 # J/Zstd.java:1111
        return 3

    @staticmethod
    def getFrameContentSize(src, srcPosition=0, srcSize=None):     # J/Zstd.java:776
        if srcSize is None:
            srcSize = len(src) - srcPosition
        a, keep = _addr(src, srcPosition)
        r = lib().zjni_getFrameContentSize(a, srcSize)
        return -1 if r == (1 << 64) - 1 else (-2 if r == (1 << 64) - 2 else r)

    decompressedSize = getFrameContentSize                         # J/Zstd.java:792

    @staticmethod
    def compressByteArray(dst, dstOffset, dstSize, src, srcOffset, srcSize, level):   # J/Zstd.java:151
        with ZstdCompressCtx() as ctx:
            ctx.setLevel(level)
            return ctx._raw(dst, dstOffset, dstSize, src, srcOffset, srcSize)

    @staticmethod
    def compress(src, level=3, checksumFlag=False):                # J/Zstd.java:1137 (byte[] -> byte[]), :60-78 (checksumFlag), :1256 (byte[], ZstdDictCompress)
        with ZstdCompressCtx() as ctx:
            if isinstance(level, ZstdDictCompress):
                ctx.loadDict(level)
                return ctx.compress(src)
            ctx.setLevel(level)
            ctx.setChecksum(checksumFlag)
            return ctx.compress(src)

    @staticmethod
    def decompressByteArray(dst, dstOffset, dstSize, src, srcOffset, srcSize):        # J/Zstd.java:463
        with ZstdDecompressCtx() as ctx:
            return ctx._raw(dst, dstOffset, dstSize, src, srcOffset, srcSize)

    @staticmethod
    def decompress(src, originalSize_or_dict, originalSize=None):   # J/Zstd.java:1435 (byte[], int), :1470-1500 (byte[], dict, int)
        with ZstdDecompressCtx() as ctx:
            if originalSize is None:
                return ctx.decompress(src, originalSize_or_dict)
            ctx.loadDict(originalSize_or_dict)
            return ctx.decompress(src, originalSize)


class _AutoClose:
    """J/AutoCloseBase.java: use-after-close raises."""

    def __init__(self):
        self._closed = False

    def close(self):
        self._closed = True

    def _ensure_open(self):
        if self._closed:
            raise RuntimeError("Closed")          # IllegalStateException("Closed") in Java

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class ZstdCompressCtx(_AutoClose):
    """J/ZstdCompressCtx.java — one-shot methods of the hot path."""

    def __init__(self):
        super().__init__()
        self.level = 3                                             # ZSTD_CLEVEL_DEFAULT
        self.checksum = False                                      # ZSTD_c_checksumFlag default
        self.hashLog = 0                                           # ZSTD_c_hashLog / ZSTD_c_chainLog, 0 = not set
        self.chainLog = 0
        self._cdict = None                                         # ZstdDictCompress in use (ZSTD_CCtx_refCDict)
        self._raw_dict = None                                      # byte[] dictionary (ZSTD_CCtx_loadDictionary): digested at the ctx's level
        self._own = None

    def loadDict(self, dictionary):                                # J/ZstdCompressCtx.java:424-470 (ZstdDictCompress, byte[] or null)
        self._ensure_open()
        if self._own is not None:
            self._own[1].close()
            self._own = None
        self._cdict = dictionary if isinstance(dictionary, ZstdDictCompress) else None
        self._raw_dict = None if (dictionary is None or self._cdict is not None) else bytes(dictionary)
        return self

    def close(self):
        if self._own is not None:
            self._own[1].close()
            self._own = None
        super().close()

    def _active_dict(self):
        if self._cdict is not None:
            self._cdict._ensure_open()
            return self._cdict
        if self._raw_dict is not None:
            if self._own is None:      # ZSTD_initLocalDict (zstd_compress.c:1246-1290): a CDict made at the first compress call, at the level set by
                                       # then, and kept until the dictionary is replaced — a later setLevel does not rebuild it (checked against the reference)
                self._own = (self.level, ZstdDictCompress(self._raw_dict, self.level))
            return self._own[1]
        return None

    def setChecksum(self, checksumFlag):                           # J/ZstdCompressCtx.java:105 -> setChecksum0 (N/jni_fast_zstd.c:276-282)
        self._ensure_open()
        self.checksum = bool(checksumFlag)
        return self

    def setHashLog(self, hashLog):                                 # J/ZstdCompressCtx.java (setHashLog0 -> ZSTD_c_hashLog)
        self._ensure_open()
        self.hashLog = hashLog
        return self

    def setChainLog(self, chainLog):                               # J/ZstdCompressCtx.java (setChainLog0 -> ZSTD_c_chainLog)
        self._ensure_open()
        self.chainLog = chainLog
        return self

    def setLevel(self, level):                                     # J/ZstdCompressCtx.java:69
        self._ensure_open()
        self.level = level
        return self

    def _raw(self, dst, dstOffset, dstSize, src, srcOffset, srcSize):
        """compressByteArray0 / compressDirectByteBuffer0 (N/jni_fast_zstd.c:586-640): argument checks
        in the same order with the same error codes, then the GPU path instead of ZSTD_compress2."""
        if dst is None:
            return -70
        if src is None:
            return -72
        if dstOffset < 0:
            return -70
        if srcOffset < 0 or srcSize < 0:
            return -72
        if srcOffset + srcSize > len(src):
            return -72
        if dstOffset + dstSize > len(dst):
            return -70
        sa, k1 = _addr(src, srcOffset)
        da, k2 = _addr(dst, dstOffset)
        cd = self._active_dict()
        if cd is not None:                                         # ZSTD_CCtx_refCDict + ZSTD_compress2
            sp, dp = (C.c_void_p * 1)(sa), (C.c_void_p * 1)(da)
            ss, dc, res = (C.c_size_t * 1)(srcSize), (C.c_size_t * 1)(dstSize), (C.c_size_t * 1)()
            r = lib().zjni_compress_batch_usingCDict(sp, ss, dp, dc, res, 1, cd._ptr, 1 if self.checksum else 0)
            if not lib().zjni_isError(r):
                r = res[0]
        elif self.hashLog or self.chainLog:
            sp, dp = (C.c_void_p * 1)(sa), (C.c_void_p * 1)(da)
            ss, dc, res = (C.c_size_t * 1)(srcSize), (C.c_size_t * 1)(dstSize), (C.c_size_t * 1)()
            r = lib().zjni_compress_batch_advanced(sp, ss, dp, dc, res, 1, self.level, 1 if self.checksum else 0, self.hashLog, self.chainLog)
            if not lib().zjni_isError(r):
                r = res[0]
        else:
            r = lib().zjni_compress2(da, dstSize, sa, srcSize, self.level, 1 if self.checksum else 0)
        return r - (1 << 64) if r >= (1 << 63) else r

    def compressByteArray(self, dstBuff, dstOffset, dstSize, srcBuff, srcOffset, srcSize):   # J/ZstdCompressCtx.java:691
        self._ensure_open()
        size = self._raw(dstBuff, dstOffset, dstSize, srcBuff, srcOffset, srcSize)
        if Zstd.isError(size):
            raise ZstdException(size)
        if size > 0x7FFFFFFF:
            raise ZstdException(1, "Output size is greater than MAX_INT")
        return size

    compressDirectByteBuffer = compressByteArray                   # J/ZstdCompressCtx.java:647 (same contract)

    def compress(self, src, dst=None):                             # J/ZstdCompressCtx.java:780-797
        self._ensure_open()
        if dst is not None:
            return self.compressByteArray(dst, 0, len(dst), src, 0, len(src))
        bound = Zstd.compressBound(len(src))
        if bound > 0x7FFFFFFF:
            raise ZstdException(1, "Max output size is greater than MAX_INT")
        out = bytearray(bound)
        n = self.compressByteArray(out, 0, bound, src, 0, len(src))
        return bytes(out[:n])


class ZstdDictCompress(_AutoClose):
    """J/ZstdDictCompress.java: a dictionary digested once for one compression level (ZSTD_createCDict,
    N/jni_fast_zstd.c:18-52) and shared read-only by any number of compress calls; close() frees the device copy.
    Constructors: (dict, level) and (dict, offset, length, level)."""

    def __init__(self, dictionary, *args):
        super().__init__()
        if len(args) == 1:
            offset, length, level = 0, None, args[0]
        elif len(args) == 3:
            offset, length, level = args
        else:
            raise TypeError("ZstdDictCompress(dict, level) or ZstdDictCompress(dict, offset, length, level)")
        data = bytes(dictionary)[offset:(None if length is None else offset + length)]
        self._level = level
        self._ptr = lib().zjni_createCDict(data, len(data), level)
        if not self._ptr:
            raise ZstdException(30 if 0 <= level <= 3 else 42)   # the Java class throws IllegalStateException("ZSTD_createCDict failed")

    def level(self):                                             # J/ZstdDictCompress.java:110
        return self._level

    def getDictID(self):
        self._ensure_open()
        return lib().zjni_getDictID_fromCDict(self._ptr)

    def close(self):
        if not self._closed and self._ptr:
            lib().zjni_freeCDict(self._ptr)
            self._ptr = None
        super().close()


class ZstdDictDecompress(_AutoClose):
    """J/ZstdDictDecompress.java: a dictionary digested once (ZSTD_createDDict, N/jni_fast_zstd.c:56-75) and shared
    read-only by any number of decompress calls; close() frees the device copy."""

    def __init__(self, dictionary, offset=0, length=None):
        super().__init__()
        data = bytes(dictionary)[offset:(None if length is None else offset + length)]
        self._ptr = lib().zjni_createDDict(data, len(data))
        if not self._ptr:
            raise ZstdException(30)                              # dictionary_corrupted (the Java class throws IllegalStateException)

    def close(self):
        if not self._closed and self._ptr:
            lib().zjni_freeDDict(self._ptr)
            self._ptr = None
        super().close()

    def getDictID(self):                                         # Zstd.getDictIdFromDict
        self._ensure_open()
        return lib().zjni_getDictID_fromDDict(self._ptr)


class ZstdDecompressCtx(_AutoClose):
    """J/ZstdDecompressCtx.java — one-shot methods of the hot path."""

    def __init__(self):
        super().__init__()
        self._ddict = None
        self._owned = None

    def loadDict(self, dictionary):                              # J/ZstdDecompressCtx.java:88-121 (ZstdDictDecompress or byte[])
        self._ensure_open()
        if self._owned is not None:
            self._owned.close()
            self._owned = None
        if dictionary is None:
            self._ddict = None
        elif isinstance(dictionary, ZstdDictDecompress):
            dictionary._ensure_open()
            self._ddict = dictionary
        else:
            self._owned = ZstdDictDecompress(dictionary)
            self._ddict = self._owned
        return self

    def close(self):
        if self._owned is not None:
            self._owned.close()
            self._owned = None
        super().close()

    def _raw(self, dst, dstOffset, dstSize, src, srcOffset, srcSize):
        """decompressByteArray0 / decompressDirectByteBuffer0 (N/jni_fast_zstd.c:777-836)."""
        if dst is None:
            return -70
        if src is None:
            return -72
        if dstOffset < 0:
            return -70
        if srcOffset < 0 or srcSize < 0:
            return -72
        if srcOffset + srcSize > len(src):
            return -72
        if dstOffset + dstSize > len(dst):
            return -70
        sa, k1 = _addr(src, srcOffset)
        da, k2 = _addr(dst, dstOffset)
        dd = self._ddict._ptr if self._ddict is not None else None
        r = lib().zjni_decompress_usingDDict(da, dstSize, sa, srcSize, dd)
        return r - (1 << 64) if r >= (1 << 63) else r

    def decompressByteArray(self, dstBuff, dstOffset, dstSize, srcBuff, srcOffset, srcSize):   # J/ZstdDecompressCtx.java:239
        self._ensure_open()
        size = self._raw(dstBuff, dstOffset, dstSize, srcBuff, srcOffset, srcSize)
        if Zstd.isError(size):
            raise ZstdException(size)
        if size > 0x7FFFFFFF:
            raise ZstdException(1, "Output size is greater than MAX_INT")
        return size

    decompressDirectByteBuffer = decompressByteArray               # J/ZstdDecompressCtx.java:197

    def decompress(self, src, originalSize=None, dst=None):        # J/ZstdDecompressCtx.java:381-420
        self._ensure_open()
        if dst is not None:
            return self.decompressByteArray(dst, 0, len(dst), src, 0, len(src))
        if originalSize is None or originalSize < 0:
            raise ZstdException(72, "Original size should not be negative")
        out = bytearray(originalSize)
        n = self.decompressByteArray(out, 0, originalSize, src, 0, len(src))
        return bytes(out[:n])


# --------------------------------------------------------------------------- batch entries ----
def _check_launch(r):
    if lib().zjni_isError(r):
        raise ZstdException(r)


def compress_batch(buffers, level=3, checksum=False, dictionary=None, hash_log=0, chain_log=0, capacities=None):
    """n independent buffers -> n zstd frames through zjni_compress_batch2 / _usingCDict / _advanced (host pointers);
    `dictionary` is a ZstdDictCompress (its level applies); hash_log / chain_log = ZstdCompressCtx.setHashLog / setChainLog;
    `capacities` = the destination sizes (default Zstd.compressBound of each buffer, what Zstd.compress(src) allocates)."""
    caps = [Zstd.compressBound(len(b)) for b in buffers] if capacities is None else list(capacities)
    return _host_batch(buffers, caps, True, level, checksum, dictionary, hash_log, chain_log)


def compress_stream(data, level=3, checksum=False, flush_at=(), final=True, known_empty=None):
    """The frame com.github.luben.zstd.ZstdDirectBufferCompressingStream / ZstdOutputStream produce for `data` written without a pledged size
    (ZSTD_compressStream2; N/jni_directbuffercompress_zstd.c:97-161), through zjni_compress_stream: `flush_at` = byte counts after which flush() was
    called; final=False: flushed but not closed (the frame's beginning up to the last flush); known_empty: closed before any other call (default:
    nothing was written and nothing flushed)."""
    L = lib()
    data = bytes(data)
    if known_empty is None:
        known_empty = final and not data and not flush_at
    fl = (C.c_uint32 * max(len(flush_at), 1))(*flush_at)
    cap = len(data) + (len(data) >> 8) + 4096 + 64 * (len(flush_at) + 2)
    dst = C.create_string_buffer(cap)
    src = C.create_string_buffer(data, max(len(data), 1))
    r = L.zjni_compress_stream(dst, cap, src, len(data), level, 1 if checksum else 0, fl, len(flush_at), 1 if final else 0, 1 if known_empty else 0)
    if L.zjni_isError(r):
        raise ZstdException(r)
    return dst.raw[:r]


def decompress_batch(frames, capacities, dictionary=None):
    """n independent frames -> n buffers through zjni_decompress_batch[_usingDDict] (host pointers)."""
    return _host_batch(frames, list(capacities), False, 0, False, dictionary)


def _host_batch(srcs, caps, is_compress, level, checksum=False, dictionary=None, hash_log=0, chain_log=0):
    L = lib()
    n = len(srcs)
    if n == 0:
        return []
    keep = [C.create_string_buffer(bytes(s), max(len(s), 1)) for s in srcs]
    outs = [C.create_string_buffer(max(c, 1)) for c in caps]
    sp = (C.c_void_p * n)(*[C.addressof(k) for k in keep])
    dp = (C.c_void_p * n)(*[C.addressof(o) for o in outs])
    ss = (C.c_size_t * n)(*[len(s) for s in srcs])
    dc = (C.c_size_t * n)(*caps)
    res = (C.c_size_t * n)()
    if is_compress and dictionary is not None:
        r = L.zjni_compress_batch_usingCDict(sp, ss, dp, dc, res, n, dictionary._ptr, 1 if checksum else 0)
    elif is_compress and (hash_log or chain_log):
        r = L.zjni_compress_batch_advanced(sp, ss, dp, dc, res, n, level, 1 if checksum else 0, hash_log, chain_log)
    elif is_compress:
        r = L.zjni_compress_batch2(sp, ss, dp, dc, res, n, level, 1 if checksum else 0)
    else:
        r = L.zjni_decompress_batch_usingDDict(sp, ss, dp, dc, res, n, dictionary._ptr if dictionary is not None else None)
    _check_launch(r)
    out = []
    for i in range(n):
        if L.zjni_isError(res[i]):
            out.append(ZstdException(res[i]))
        else:
            out.append(outs[i].raw[:res[i]])
    return out


def synth_host(buf_size, first_index, n):
    """The SURVEY §8(d) mixed-entropy generator on the host (same bytes as the device generator)."""
    out = C.create_string_buffer(buf_size * n)
    lib().zjni_synth_fill_host(out, buf_size, first_index, n)
    return out.raw


from . import batch, shard  # noqa: E402,F401
