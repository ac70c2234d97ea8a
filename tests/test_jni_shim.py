"""The JNI side of the drop-in (zstd-jni_amd/jni/zjni_shim.c): the hot-path Java_com_github_luben_zstd_* natives, driven
through a hand-built JNIEnv (tests/jni/harness.c) next to the reference's own JNI library built from its sources
(oracle/_ref/libzstd-jni-ref.so).  CPU: symbol surface + the forwarding path (no GPU: every call goes to the bundled
library's native of the same name).  GPU: return values and bytes of every native equal the reference's.  With the bundled library behind the shim every native
of the two context classes (parameters, reset, pledged size, frame progression, the stream natives) is driven through the
shim's export: nativePtr is the bundled library's own handle, whoever defines the native."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "zstd-jni_amd", "lib", "libzstd-jni-amd.so")
REFJNI = os.path.join(ROOT, "oracle", "_ref", "libzstd-jni-ref.so")
HARNESS = os.path.join(ROOT, "tests", "jni", "_build", "harness")

def _dict_file(tmp_path_factory=None):
    """a zstd-format dictionary (with a dictID) for the harness: trained by the reference's own trainer"""
    import sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import ref
    from util import json_records
    recs = json_records(20000, seed=3)
    samples = [b",".join(recs[i * 13:i * 13 + 200])[:4096] for i in range(1000)]
    d = ref.train_dict(samples, 30000)
    assert ref.dict_id(d) > 0
    path = os.path.join(ROOT, "tests", "jni", "_build", "trained.dict")
    with open(path, "wb") as f:
        f.write(d)
    return path


def _built():
    import __graft_entry__ as e
    e.build_jni()
    return all(os.path.exists(p) for p in (SHIM, REFJNI, HARNESS))


def _exports(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {l.split()[-1] for l in out.splitlines() if " T Java_com_github_luben_zstd_" in l}


def test_shim_exports_every_native_of_the_reference_library():
    """zstd-jni loads ONE library (J/util/Native.java:90-180): the shim's dynamic symbol table is a superset of the reference
    library's 149 Java_* exports, plus the three batch natives"""
    if not _built():
        pytest.skip("no <jni.h> in this environment and no prebuilt shim")
    ref, shim = _exports(REFJNI), _exports(SHIM)
    assert len(ref) == 149
    assert not (ref - shim), sorted(ref - shim)[:10]
    assert shim - ref == {"Java_com_github_luben_zstd_Zstd_" + n for n in ("compressBatch0", "decompressBatch0", "compressBatchDict0", "compressBatchBegin0", "decompressBatchBegin0", "batchFinish0")}
    listed = open(os.path.join(ROOT, "zstd-jni_amd", "jni", "jni_symbols.txt")).read().split()
    assert {"Java_com_github_luben_zstd_" + n for n in listed} == ref           # the committed list the build falls back on


def test_shim_forwards_to_the_bundled_library_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    if not _built():
        pytest.skip("no <jni.h> in this environment and no prebuilt shim")
    env = dict(os.environ, ZSTD_JNI_CPU_LIB=REFJNI, HARNESS_SKIP_BATCH="1", HARNESS_MAX_LEVEL="2", HARNESS_DICT_FILE=_dict_file(), HARNESS_EXPECT="cpu")
    out = subprocess.run([HARNESS, REFJNI, SHIM], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "JNI-HARNESS OK" in out.stdout, out.stdout[-2000:] + out.stderr[-500:]


@pytest.mark.parametrize("bundled", [False, True])
def test_shim_answers_frame_inspection_and_constants_itself(bundled):
    """Zstd.decompressedSize / getFrameContentSize / findFrameCompressedSize / getDictIdFromFrame / getDictIdFromDict (byte[] and direct forms) and the
    36 constants are host-side natives of the shim (zjni_shim.c "frame inspection and constants": no GPU, no bundled library): equal to the reference's
    natives on ~4 900 inputs — real frames, truncations, every header bit flipped, hand-made headers, skippable / pre-1.0 magics, magicless, noise, dictionaries,
    bad direct ranges.  Loading the bundled library changes nothing (only pre-1.0 frames are passed to it)."""
    if not _built():
        pytest.skip("no <jni.h> in this environment and no prebuilt shim")
    env = dict(os.environ, HARNESS_ONLY_HELPERS="1")
    env.pop("ZSTD_JNI_CPU_LIB", None)
    if bundled: env["ZSTD_JNI_CPU_LIB"] = REFJNI
    out = subprocess.run([HARNESS, REFJNI, SHIM], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "JNI-HARNESS OK" in out.stdout and "INSPECTION cases=4" in out.stdout, out.stdout[-2000:] + out.stderr[-500:]
    trampolines = [l for l in open(os.path.join(ROOT, "zstd-jni_amd", "jni", "forward_list.h")) if l.startswith("FWD(")]
    assert sorted(trampolines) == ["FWD(Zstd_trainFromBuffer0)\n", "FWD(Zstd_trainFromBufferDirect0)\n"], trampolines      # dictionary training: the only natives left to the bundled library alone


EMU_SHIM = os.path.join(ROOT, "tests", "jni", "_build", "emu", "libzstd-jni-amd.so")


def _built_emu():
    """the JNI library over the TEST DOUBLE of libzjni_amd.so (tests/jni/emu_abi.cpp: the 22 C-ABI entries the JNI library imports, over the kernel bodies compiled
    lane-serial under g++) — its whole GPU route on a machine without a GPU"""
    if not _built():
        return False
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "jni"), "emu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.exists(EMU_SHIM)


@pytest.fixture(scope="module")
def emu_legs():
    """both legs started side by side (a minute of lane-serial emulation each), collected by the tests below"""
    if not _built_emu():
        pytest.skip("no <jni.h> in this environment and no prebuilt shim")
    procs = {}
    for leg in ("gpu-only", "bundled-library-behind"):
        env = dict(os.environ, HARNESS_DICT_FILE=_dict_file(), HARNESS_FUZZ="11,80")      # + 80 random scripts of directives per kind (tools/fuzz_jni_streams.sh runs thousands)
        for k in ("ZSTD_JNI_CPU_LIB", "ZSTD_JNI_GPU_STREAMS", "ZSTD_JNI_GPU_PER_BUFFER", "ZSTD_JNI_GPU_AGGREGATE"):
            env.pop(k, None)
        if leg == "gpu-only":
            env.update(HARNESS_PLAIN_MAX_LEVEL="8", HARNESS_EXPECT="gpu", HARNESS_STREAM_MAX="0", HARNESS_PIECES="1")      # PIECES: frames fed to the decompress streams in pieces / into small targets — collected by the shim when no bundled stream exists
        else:
            env.update(ZSTD_JNI_CPU_LIB=REFJNI, ZSTD_JNI_GPU_STREAMS="1", HARNESS_SKIP_BATCH="1", HARNESS_MAX_LEVEL="2")
        procs[leg] = subprocess.Popen([HARNESS, REFJNI, EMU_SHIM], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out = {}
    for leg, p in procs.items():
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill(); o, e = p.communicate()
        out[leg] = (p.returncode, o, e)
    return out


@pytest.mark.parametrize("leg", ["gpu-only", "bundled-library-behind"])
def test_shim_gpu_route_over_the_emulated_kernels(emu_legs, leg):
    """The two GPU legs below, on the CPU: the JNI library linked against the emulation of the C-ABI.  gpu-only: no bundled library, every native must be answered by
    the (emulated) GPU path — one-shot natives at levels 1-8, dictionaries, frame parameters, the five stream classes, the context streams, frames in pieces — and equal
    the reference's JNI library (78 000 checks, none forwarded).  bundled-library-behind: ZSTD_JNI_GPU_STREAMS=1 with the reference behind it — streams that outgrow the
    window are replayed into the bundled library mid-frame."""
    rc, stdout, stderr = emu_legs[leg]
    assert rc == 0 and "JNI-HARNESS OK" in stdout, stdout[-3000:] + stderr[-500:]
    stats = [l for l in stdout.splitlines() if l.startswith("JNI-HARNESS STATS")][0]
    served = int(stats.split("served_by_gpu=")[1].split()[0]); declined = int(stats.split("forwarded_after_gpu_declined=")[1].split()[0])
    assert served > (3000 if leg == "gpu-only" else 300), stats
    assert (declined == 0) if leg == "gpu-only" else (declined > 0), stats          # the second leg must have replayed streams into the bundled library


@pytest.mark.gpu
def test_shim_equals_reference_jni_on_the_gpu():
    assert all(os.path.exists(p) for p in (SHIM, REFJNI, HARNESS)), "prebuilt JNI shim / reference JNI library / harness missing"
    env = dict(os.environ, HARNESS_DICT_FILE=_dict_file(), HARNESS_PLAIN_MAX_LEVEL="8", HARNESS_EXPECT="gpu", HARNESS_STREAM_MAX="0", HARNESS_FUZZ="12,25", HARNESS_FUZZ_SKIP_HEAP="1")      # (the heap-array classes' random scripts run in the CPU legs: same shim code, same C-ABI calls)      # (streams: nothing to outgrow into, totals stay within the level's window)      # zjni_shim_stats: served > 0, forwarded == 0   # levels 4-8 (<= 128 KiB) through the one-shot natives as well
    env.pop("ZSTD_JNI_CPU_LIB", None)                  # nothing to forward to: every result must come from the GPU library
    out = subprocess.run([HARNESS, REFJNI, SHIM], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "JNI-HARNESS OK" in out.stdout, out.stdout[-3000:] + out.stderr[-500:]


@pytest.mark.gpu
def test_shim_streams_on_the_gpu_with_the_bundled_library_behind():
    """the stream natives with BOTH a GPU and the bundled library (ZSTD_JNI_GPU_STREAMS=1): streams within the level's window come from the GPU route, a stream that
    outgrows it is replayed into the bundled library's stream and continues there — the same bytes as the reference either way, whatever room the target buffer has"""
    assert all(os.path.exists(p) for p in (SHIM, REFJNI, HARNESS)), "prebuilt JNI shim / reference JNI library / harness missing"
    env = dict(os.environ, ZSTD_JNI_CPU_LIB=REFJNI, ZSTD_JNI_GPU_STREAMS="1", HARNESS_SKIP_BATCH="1", HARNESS_MAX_LEVEL="2", HARNESS_DICT_FILE=_dict_file(), HARNESS_FUZZ="13,25", HARNESS_FUZZ_SKIP_HEAP="1")
    out = subprocess.run([HARNESS, REFJNI, SHIM], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "JNI-HARNESS OK" in out.stdout, out.stdout[-3000:] + out.stderr[-500:]
