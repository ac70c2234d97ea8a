"""GPU (-m gpu): level 4 through the C-ABI — greedy on the hash chain (inputs <= 16 KiB) and double-fast with 2^17-entry tables
(<= 128 KiB), both with their tables in HBM on the wave-per-frame kernel — byte-identical to the reference's ZSTD_compress2 at level 4
(N/compress/clevels.h:84,110; N/compress/zstd_lazy.c:667-723,1516-1784), small and large batches, mixed with the sizes it refuses."""
import os
import random

import pytest

from util import json_records, needs_tuning_build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(zj):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    zj.batch.init(0)
    return zj


def _inputs(gpu, seed, count):
    rnd = random.Random(seed)
    text = b",".join(json_records(6000, seed=seed))
    out = [b"", b"a", b"abcdefg", bytes(8), text[:16384], text[:16385], text[:131072], b"x" * 70000]
    while len(out) < count:
        n = rnd.choice([rnd.randrange(1, 300), rnd.randrange(300, 4097), rnd.randrange(4097, 16385), 4096, 16384, rnd.randrange(16385, 70000), 65536, rnd.randrange(70000, 131073)])
        k = rnd.randrange(4)
        if k == 0: o = rnd.randrange(0, len(text) - n); d = text[o:o + n]
        elif k == 1: d = gpu.synth_host(n, rnd.randrange(1 << 20), 1)
        elif k == 2: per = os.urandom(rnd.choice([1, 3, 17, 300])); d = (per * (n // len(per) + 1))[:n]
        else: h = n // 2; d = text[:h] + os.urandom(n - h)
        out.append(d)
    return out


@pytest.mark.parametrize("route", ["wave", "lanes", "one-lane"])
@pytest.mark.parametrize("count", [40, 4500])
def test_gpu_level4_frames_are_the_references(gpu, oracle_ref, monkeypatch, count, route):
    """route (frames above 16 KiB, double-fast): the wave matcher of zj_match_wavex.h on the wave-per-frame kernel (the default, any
    batch size), the lane-slot kernel of large batches (ZJNI_L4_LANES=1), the one-lane parse (ZJNI_MULTI_WAVE=0)"""
    if route != "wave": needs_tuning_build(gpu)
    if route == "lanes": monkeypatch.setenv("ZJNI_L4_LANES", "1")
    if route == "one-lane": monkeypatch.setenv("ZJNI_MULTI_WAVE", "0")
    datas = _inputs(gpu, 7 + count, count) + [gpu.synth_host(131073, 1, 1), gpu.synth_host(300000, 2, 1)]
    for checksum in (False, True):
        outs = gpu.compress_batch(datas, 4, checksum=checksum)
        good = []
        for d, z in zip(datas, outs):
            if len(d) > 131072:
                assert isinstance(z, Exception) and z.getErrorCode() == 201, len(d)       # the reference's row-based finder: left to the CPU path
                continue
            assert not isinstance(z, Exception), (len(d), z)
            assert z == oracle_ref.compress(d, 4, checksum), (len(d), checksum)
            good.append((d, z))
        back = gpu.decompress_batch([z for _, z in good], [len(d) for d, _ in good])
        for (d, _), b in zip(good, back):
            assert b == d


def test_gpu_level4_context_classes(gpu, oracle_ref):
    d = b",".join(json_records(400, seed=3))[:12000]
    assert gpu.Zstd.compress(d, 4) == oracle_ref.compress(d, 4)
    ctx = gpu.ZstdCompressCtx(); ctx.setLevel(4); ctx.setChecksum(True)
    assert ctx.compress(d) == oracle_ref.compress(d, 4, True)
    for level in (5, 6, 7, 8):                               # lazy / lazy2 on the hash chain, inputs <= 16 KiB
        assert gpu.Zstd.compress(d, level) == oracle_ref.compress(d, level)
    with pytest.raises(gpu.ZstdException) as ex:
        gpu.Zstd.compress(d, 9)                               # btlazy2: not served
    assert ex.value.getErrorCode() == 42
    big = (d * 12)[:131072]
    for level in (5, 6, 7, 8):                               # above 16 KiB: the row-based finder
        assert gpu.Zstd.compress(big, level) == oracle_ref.compress(big, level)
    with pytest.raises(gpu.ZstdException) as ex:
        gpu.Zstd.compress(big + b"x", 5)                      # multi-block frames of these strategies: not served
    assert ex.value.getErrorCode() == 201


@pytest.mark.parametrize("level", [5, 6, 7, 8])
def test_gpu_lazy_levels(gpu, oracle_ref, level):
    datas = _inputs(gpu, 30 + level, 1500) + [gpu.synth_host(16385, 3, 1), gpu.synth_host(131073, 3, 1)]
    outs = gpu.compress_batch(datas, level, checksum=(level % 2 == 0))
    good = []
    for d, z in zip(datas, outs):
        if len(d) > 131072:
            assert isinstance(z, Exception) and z.getErrorCode() == 201
            continue
        assert not isinstance(z, Exception), (len(d), z)
        assert z == oracle_ref.compress(d, level, level % 2 == 0), (level, len(d))
        good.append((d, z))
    back = gpu.decompress_batch([z for _, z in good], [len(d) for d, _ in good])
    for (d, _), b in zip(good, back):
        assert b == d
