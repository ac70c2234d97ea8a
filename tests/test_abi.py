"""CPU (-m "not gpu"): the C-ABI library loads, exports every symbol include/zjni_amd.h declares,
its GPU-free helpers agree with the reference, and compute entries fail loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, golden


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zjni_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zjni_[A-Za-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(zj):
    L = zj.lib()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/zjni_amd.h but not exported"
    assert set(names) == set(zj.EXPORTS)


def test_error_helpers_match_libzstd(zj, oracle_ref):
    L, R = zj.lib(), oracle_ref.lib()
    for code in (1, 10, 14, 16, 20, 22, 24, 30, 32, 40, 42, 44, 64, 70, 72):
        r = (1 << 64) - code
        assert L.zjni_isError(r) == 1
        assert L.zjni_getErrorCode(r) == code
        assert L.zjni_getErrorName(r) == R.ZSTD_getErrorName(r), code
    assert L.zjni_isError(123456) == 0
    assert b"Destination buffer is too small" == L.zjni_getErrorName((1 << 64) - 70)   # T/scala/Zstd.scala:199


def test_compress_bound_matches_reference(zj, oracle_ref):
    L, R = zj.lib(), oracle_ref.lib()
    for s in (0, 1, 100, 4096, 65536, 131071, 131072, 131073, 1 << 20, 10_000_000):
        assert L.zjni_compressBound(s) == R.ZSTD_compressBound(s)


def test_frame_content_size(zj, oracle_ref):
    for data in (b"", b"a", b"abc" * 100, b"x" * 70000):
        z = oracle_ref.compress(data, 3)
        assert zj.Zstd.getFrameContentSize(z) == len(data)
    assert zj.Zstd.getFrameContentSize(golden("xmlsmall-sized.zst")) == 102
    assert zj.Zstd.getFrameContentSize(golden("xml-1.zst")) == -1          # streaming CLI frame: unknown size
    assert zj.Zstd.getFrameContentSize(b"\x00" * 8) == -2


def test_frame_content_size_on_damaged_headers(zj, oracle_ref):
    """ZSTD_getFrameContentSize on truncated / overwritten headers (zstd frames with every header shape, a streamed frame
    without content size, a skippable frame): same value or the same ZSTD_CONTENTSIZE_ERROR as the reference"""
    import ctypes as C
    import random
    import struct
    L, R = zj.lib(), oracle_ref.lib()
    R.ZSTD_getFrameContentSize.argtypes = [C.c_char_p, C.c_size_t]
    rnd = random.Random(3)
    base = [oracle_ref.compress(bytes(rnd.randrange(5) for _ in range(k)), 3, checksum=bool(k & 1)) for k in (0, 1, 100, 255, 256, 300, 70000)]
    base += [struct.pack("<II", 0x184D2A53, 5) + b"hello", oracle_ref.compress_stream(b"abc" * 1000, 3)]
    for _ in range(20000):
        z = bytearray(rnd.choice(base))
        for _ in range(rnd.randrange(0, 3)):
            z[rnd.randrange(0, min(len(z), 14))] = rnd.getrandbits(8)
        z = bytes(z[:rnd.randrange(0, min(len(z), 20) + 1)]) if rnd.random() < 0.5 else bytes(z)
        buf = C.create_string_buffer(z, max(len(z), 1))
        assert L.zjni_getFrameContentSize(buf, len(z)) == R.ZSTD_getFrameContentSize(z, len(z)), z[:16].hex()


def test_synth_is_deterministic_and_classed(zj):
    a = zj.synth_host(4096, 0, 8)
    b = zj.synth_host(4096, 4, 4)
    assert a[4 * 4096:] == b
    assert a[:4096].count(b" ") > 300                       # class 0: words
    assert a[4096:8192].startswith(b'{"id":')               # class 1: JSON-like
    assert max(a[2 * 4096:3 * 4096]) < 16                   # class 2: 4-bit values
    assert len(set(a[3 * 4096:4 * 4096])) > 200             # class 3: random bytes


def test_compute_fails_loudly_without_gpu(zj):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = zj.lib()
    assert L.zjni_device_count() == 0
    assert L.zjni_init(0) == -200
    dst = C.create_string_buffer(100)
    r = L.zjni_compress(dst, 100, b"abc", 3, 3)
    assert L.zjni_isError(r) and L.zjni_getErrorCode(r) == 200
    r = L.zjni_decompress(dst, 100, golden("xmlsmall-sized.zst"), 103)
    assert L.zjni_isError(r) and L.zjni_getErrorCode(r) == 200
    with pytest.raises(zj.ZstdException) as e:
        zj.Zstd.compress(b"abc", 3)
    assert e.value.getErrorCode() == 200


def test_java_mirror_argument_checks(zj):
    # N/jni_fast_zstd.c:586-600, :615-623: same checks, same codes, before any device work
    ctx = zj.ZstdCompressCtx()
    assert ctx._raw(bytearray(10), -1, 10, b"abc", 0, 3) == -70
    assert ctx._raw(bytearray(10), 0, 10, b"abc", -1, 3) == -72
    assert ctx._raw(bytearray(10), 0, 10, b"abc", 0, 4) == -72
    assert ctx._raw(bytearray(10), 5, 10, b"abc", 0, 3) == -70
    dctx = zj.ZstdDecompressCtx()
    assert dctx._raw(bytearray(10), 0, 11, b"abc", 0, 3) == -70
    assert dctx._raw(bytearray(10), 0, 10, b"abc", 2, 3) == -72
    ctx.close()
    with pytest.raises(RuntimeError):
        ctx.setLevel(1)                                     # T/scala/Zstd.scala:1000-1020 use-after-close


def test_route_names_and_build_stamp(zj):
    """zjni_route_kernel names the kernel a route's match-finder time belongs to (bench.py's roofline takes the name from here, include/zjni_amd.h ZJNI_ROUTE_*);
    zjni_build_stamp is the hash of csrc/ + the header the library was compiled from — the one profiles/r03_pmc_traffic.json is keyed by"""
    import ctypes as C
    import re
    L = zj.lib()
    L.zjni_route_kernel.restype = C.c_char_p; L.zjni_route_kernel.argtypes = [C.c_int]
    L.zjni_build_stamp.restype = C.c_char_p
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "zjni_amd.h")).read()
    routes = {name: int(val) for name, val in re.findall(r"#define (ZJNI_ROUTE_[A-Z_]+) (\d+)", header)}
    assert routes["ZJNI_ROUTE_WAVE_HBM"] == 9 and len(set(routes.values())) == len(routes)
    want = {"ZJNI_ROUTE_FUSED": b"zj_encode_kernel", "ZJNI_ROUTE_WAVE": b"zj_enc_match_wave_kernel", "ZJNI_ROUTE_LANE": b"zj_enc_match_kernel",
            "ZJNI_ROUTE_LANE_GATED": b"zj_enc_match_gated_kernel", "ZJNI_ROUTE_RUN": b"zj_enc_match_run_kernel", "ZJNI_ROUTE_RUN_FLAGS": b"zj_enc_match_run_kernel",
            "ZJNI_ROUTE_HYBRID": b"zj_enc_match_kernel", "ZJNI_ROUTE_WAVE_HBM": b"zj_encode_multi_kernel", "ZJNI_ROUTE_OTHER": b"", "ZJNI_ROUTE_NONE": b"",
            "ZJNI_ROUTE_WIDE": b"zj_enc_match_wide_kernel", "ZJNI_ROUTE_PIPE": b"zj_encode_pipe_kernel"}
    for name, val in routes.items():
        assert L.zjni_route_kernel(val) == want[name], name
    assert L.zjni_build_stamp().decode() == zj.build_stamp()          # the library in the tree is the one these sources give


def test_counter_passes_are_of_these_sources(zj):
    """bench.py quotes roofline.traffic only from a PMC pass stamped with the library's own build (bench.py: the newest profiles/r*_pmc_traffic.json first): the committed
    summary of the newest round is of the committed sources, kernel by kernel — an edit under csrc/ after the last counter pass shows up here, not as a silent `traffic: null`"""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    newest = sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_traffic.json")))[-1]
    summary = json.load(open(newest))
    stamps = {rec["build_stamp"] for workload in summary.values() if isinstance(workload, dict) for rec in workload.values() if isinstance(rec, dict) and "build_stamp" in rec}
    assert stamps == {zj.build_stamp()}, (newest, stamps)
    metric = summary["metric_L3_65536x65536"]["zj_enc_match_run_kernel"]
    assert metric["hbm_bytes_per_launch"] == metric["fetch_bytes_per_launch"] + metric["write_bytes_per_launch"] > 0


def test_asynchronous_entries_fail_loudly_without_gpu(zj):
    """zjni_compress_batch_begin / zjni_decompress_batch_begin (round 5): no device, no job — and zjni_batch_finish of no job says so; there is no CPU path behind them either"""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = zj.lib()
    buf = C.create_string_buffer(b"x" * 100, 100); dst = C.create_string_buffer(200)
    sp = (C.c_void_p * 1)(C.addressof(buf)); dp = (C.c_void_p * 1)(C.addressof(dst)); ss = (C.c_size_t * 1)(100); dc = (C.c_size_t * 1)(200); res = (C.c_size_t * 1)()
    assert not L.zjni_compress_batch_begin(sp, ss, dp, dc, res, 1, 3, 0)
    assert not L.zjni_decompress_batch_begin(sp, ss, dp, dc, res, 1)
    r = L.zjni_batch_finish(None)
    assert L.zjni_isError(r) and L.zjni_getErrorCode(r) == 200
