"""GPU (-m gpu): dictionary compression through the C-ABI (zjni_createCDict, zjni_compress_batch*_usingCDict) is
byte-identical to the reference's ZstdDictCompress path (ZSTD_createCDict + ZSTD_CCtx_refCDict + ZSTD_compress2 =
ZSTD_compress_usingCDict; T/scala/ZstdDict.scala:58-216 exercises it through the Java API), and the frames decode
bit-exactly through the GPU dictionary decoder."""
import random

import pytest

import dictutil as du
from util import json_records

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(zj):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    zj.batch.init(0)
    return zj


def mixed_sources(gpu, recs, rnd, n, cutoff):
    out = []
    for i in range(n):
        k = rnd.randrange(0, len(recs) - 300)
        size = rnd.choice([0, 1, 7, 50, 130, 400, 1000, 1024, 4096, 4096, 4096, cutoff, rnd.randrange(0, cutoff + 1)])
        kind = rnd.random()
        if kind < 0.75:
            out.append(b",".join(recs[k:k + 150])[:size])
        elif kind < 0.9:
            out.append(gpu.synth_host(max(size, 1), rnd.randrange(100000), 1)[:size])
        else:
            out.append(bytes(rnd.getrandbits(8) for _ in range(min(size, 600))))
    return out


@pytest.mark.parametrize("level", [1, 2, 3])
def test_gpu_cdict_batches_byte_identical(gpu, oracle_ref, level):
    rnd = random.Random(100 + level)
    recs = json_records(30000, seed=3)
    samples = [b",".join(recs[i * 13:i * 13 + 200])[:4096] for i in range(1500)]
    for dict_size in (16384, 112640):
        dbytes = oracle_ref.train_dict(samples, dict_size)
        ref_cd = oracle_ref.CDict(dbytes, level)
        with gpu.ZstdDictCompress(dbytes, level) as cd, gpu.ZstdDictDecompress(dbytes) as dd:
            assert cd.getDictID() == oracle_ref.dict_id(dbytes) and cd.level() == level
            cutoff = 8192 if (level < 3 and not (level == 2 and 131072 < dict_size + 499 <= 262144)) else 16384
            for n in (1, 70, 700):
                srcs = mixed_sources(gpu, recs, rnd, n, cutoff)
                frames = gpu.compress_batch(srcs, dictionary=cd)
                for k, (s, f) in enumerate(zip(srcs, frames)):
                    assert not isinstance(f, Exception), (k, len(s), f)
                    assert f == ref_cd.compress(s), (dict_size, n, k, len(s))
                outs = gpu.decompress_batch(frames, [len(s) for s in srcs], dd)
                for k, (s, o) in enumerate(zip(srcs, outs)):
                    assert o == s, k
            # checksum flag + buffers beyond the attach range in the same batch: copy mode up to one block (the dictionary's tables copied, the
            # dictionary searched as an external segment: ZSTD_resetCCtx_byCopyingCDict + ZSTD_compressBlock_*_extDict), refused beyond
            big = [b",".join(recs[300:300 + 4000])[:sz] for sz in (cutoff + 1, 20000, 65536, 131071)] + [bytes(cutoff + 1), gpu.synth_host(50000, 3, 1)]
            srcs = [b",".join(recs[5:40])] + big + [b",".join(recs[100:140]), bytes(131073)]
            frames = gpu.compress_batch(srcs, checksum=True, dictionary=cd)
            for k, (x, z) in enumerate(zip(srcs[:-1], frames[:-1])):
                assert not isinstance(z, Exception), (k, len(x), z)
                assert z == ref_cd.compress(x, checksum=True), (k, len(x), level, dict_size)
            assert isinstance(frames[-1], gpu.ZstdException) and frames[-1].getErrorCode() == 201      # more than one block with a dictionary: the CPU path's
            outs = gpu.decompress_batch(frames[:-1], [len(x) for x in srcs[:-1]], dd)
            assert all(o == x for o, x in zip(outs, srcs[:-1]))


def test_gpu_cdict_api_mirror(gpu, oracle_ref):
    """ZstdCompressCtx.loadDict(ZstdDictCompress | byte[]) + compress, Zstd.compress(src, ZstdDictCompress)
    (J/ZstdCompressCtx.java:424-470, J/Zstd.java:1256); raw-content and hand-assembled dictionaries"""
    rnd = random.Random(8)
    recs = json_records(4000, seed=5)
    content = b",".join(recs[:150])
    hist = [0] * 256
    for b in content:
        hist[b] += 1
    check_dict = du.build(content, 300, hist, du.normalise([1, 0, 1, 1, 0] + [2] * 12, 7), 7, du.normalise([3 if i < 20 else 1 for i in range(40)], 8), 8,
                          du.normalise([4 if i < 10 else 1 for i in range(36)], 8), 8)
    for dbytes in (content, check_dict):
        for level in (1, 3):
            ref_cd = oracle_ref.CDict(dbytes, level)
            srcs = [b",".join(recs[200 + i * 9:200 + i * 9 + k]) for i, k in enumerate([1, 2, 3, 5, 8, 13, 20, 30, 40])] + [bytes(rnd.choice(content) for _ in range(700)), b""]
            with gpu.ZstdDictCompress(dbytes, level) as cd:
                with gpu.ZstdCompressCtx() as ctx:
                    ctx.loadDict(cd)
                    for s in srcs:
                        assert ctx.compress(s) == ref_cd.compress(s)
                    ctx.setChecksum(True)
                    assert ctx.compress(srcs[3]) == ref_cd.compress(srcs[3], checksum=True)
                    ctx.setChecksum(False).loadDict(None)
                    assert ctx.compress(srcs[3]) == oracle_ref.compress(srcs[3], 3)
                    ctx.setLevel(level).loadDict(dbytes)                          # byte[] dictionary: digested at the ctx's level
                    assert ctx.compress(srcs[4]) == ref_cd.compress(srcs[4])
                assert gpu.Zstd.compress(srcs[5], cd) == ref_cd.compress_using(srcs[5])
                with gpu.ZstdCompressCtx() as ctx:                                # destination too small: the reference's code and message
                    ctx.loadDict(cd)
                    with pytest.raises(gpu.ZstdException) as e:
                        ctx.compress(srcs[6], bytearray(12))
                    assert e.value.getErrorCode() == 70 and "Destination buffer is too small" in str(e.value)
                with pytest.raises(gpu.ZstdException) as e:
                    gpu.Zstd.compress(bytes(140000), cd)                          # more than one block with a dictionary: not served
                assert e.value.getErrorCode() == 201
    with pytest.raises(gpu.ZstdException):
        gpu.ZstdDictCompress(b"\x37\xa4\x30\xec" + bytes(40), 3)                    # magic + garbage
    with pytest.raises(gpu.ZstdException):
        gpu.ZstdDictCompress(content, 7)                                          # level outside 1..3


def test_gpu_cdict_device_batch(gpu, oracle_ref):
    """device-resident entry: 20 000 x 4 KiB JSON-like records (the shape of BASELINE config 4), sampled against the reference"""
    import torch
    recs = json_records(30000, seed=3)
    samples = [b",".join(recs[i * 13:i * 13 + 200])[:4096] for i in range(1500)]
    dbytes = oracle_ref.train_dict(samples, 112640)
    ref_cd = oracle_ref.CDict(dbytes, 3)
    n, size = 20000, 4096
    host = b"".join(b",".join(recs[(i * 7) % 29000:(i * 7) % 29000 + 60])[:size].ljust(size, b" ") for i in range(n))
    src = torch.frombuffer(bytearray(host), dtype=torch.uint8).cuda()
    off = gpu.batch.uniform_offsets(n, size, "cuda")
    bound = gpu.Zstd.compressBound(size)
    dst = torch.empty(n * bound, dtype=torch.uint8, device="cuda")
    doff = gpu.batch.uniform_offsets(n, bound, "cuda")
    with gpu.ZstdDictCompress(dbytes, 3) as cd, gpu.ZstdDictDecompress(dbytes) as dd:
        res = gpu.batch.compress(src, off, dst, doff, dictionary=cd)
        torch.cuda.synchronize()
        sizes = res.cpu().tolist()
        assert min(sizes) > 0
        out = dst.cpu().numpy().tobytes()
        for i in list(range(0, n, 397)) + [n - 1]:
            assert out[i * bound:i * bound + sizes[i]] == ref_cd.compress(host[i * size:(i + 1) * size]), i
        back = torch.empty(n * size, dtype=torch.uint8, device="cuda")
        packed, poff = gpu.batch.pack(res, dst, doff)
        r2 = gpu.batch.decompress(packed, poff, back, off, dictionary=dd)
        torch.cuda.synchronize()
        assert bool((r2 == size).all()) and torch.equal(back, src)
