"""The JNI side of the drop-in (zstd-jni_amd/jni/zjni_shim.c): the hot-path Java_com_github_luben_zstd_* natives, driven
through a hand-built JNIEnv (tests/jni/harness.c) next to the reference's own JNI library built from its sources
(oracle/_ref/libzstd-jni-ref.so).  CPU: symbol surface + the forwarding path (no GPU: every call goes to the bundled
library's native of the same name).  GPU: return values and bytes of every native equal the reference's."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "zstd-jni_amd", "lib", "libzstd-jni-amd.so")
REFJNI = os.path.join(ROOT, "oracle", "_ref", "libzstd-jni-ref.so")
HARNESS = os.path.join(ROOT, "tests", "jni", "_build", "harness")

HOT = ["ZstdCompressCtx_init", "ZstdCompressCtx_free", "ZstdCompressCtx_setLevel0", "ZstdCompressCtx_setChecksum0",
       "ZstdCompressCtx_compressDirectByteBuffer0", "ZstdCompressCtx_compressByteArray0",
       "ZstdDecompressCtx_init", "ZstdDecompressCtx_free", "ZstdDecompressCtx_decompressDirectByteBuffer0",
       "ZstdDecompressCtx_decompressByteArray0", "Zstd_compressBound", "Zstd_isError", "Zstd_getErrorName",
       "Zstd_getErrorCode", "Zstd_compressUnsafe", "Zstd_decompressUnsafe", "Zstd_compressBatch0", "Zstd_decompressBatch0",
       "ZstdDictCompress_init", "ZstdDictCompress_initDirect", "ZstdDictCompress_free", "ZstdCompressCtx_loadCDictFast0", "Zstd_compressBatchDict0",
       "ZstdDictDecompress_init", "ZstdDictDecompress_initDirect", "ZstdDictDecompress_free", "ZstdDecompressCtx_loadDDictFast0",
       "Zstd_setCompressionHashLog", "Zstd_setCompressionChainLog"]


def _built():
    import __graft_entry__ as e
    e.build_jni()
    return all(os.path.exists(p) for p in (SHIM, REFJNI, HARNESS))


def test_shim_exports_the_hot_path_natives():
    if not _built():
        pytest.skip("no <jni.h> in this environment and no prebuilt shim")
    syms = subprocess.check_output(["nm", "-D", "--defined-only", SHIM], text=True)
    ref = subprocess.check_output(["nm", "-D", "--defined-only", REFJNI], text=True)
    for name in HOT:
        assert f"Java_com_github_luben_zstd_{name}" in syms, name
        if "Batch" not in name:                      # every replaced native exists under the same name in the reference's library
            assert f"Java_com_github_luben_zstd_{name}" in ref, name


def test_shim_forwards_to_the_bundled_library_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    if not _built():
        pytest.skip("no <jni.h> in this environment and no prebuilt shim")
    env = dict(os.environ, ZSTD_JNI_CPU_LIB=REFJNI, HARNESS_SKIP_BATCH="1", HARNESS_MAX_LEVEL="2")
    out = subprocess.run([HARNESS, REFJNI, SHIM], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "JNI-HARNESS OK" in out.stdout, out.stdout[-2000:] + out.stderr[-500:]


@pytest.mark.gpu
def test_shim_equals_reference_jni_on_the_gpu():
    assert all(os.path.exists(p) for p in (SHIM, REFJNI, HARNESS)), "prebuilt JNI shim / reference JNI library / harness missing"
    env = dict(os.environ)
    env.pop("ZSTD_JNI_CPU_LIB", None)                  # nothing to forward to: every result must come from the GPU library
    out = subprocess.run([HARNESS, REFJNI, SHIM], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "JNI-HARNESS OK" in out.stdout, out.stdout[-3000:] + out.stderr[-500:]
