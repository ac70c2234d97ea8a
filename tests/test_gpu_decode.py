"""GPU (-m gpu): batched decompress through the C-ABI (libzjni_amd.so, wave64 kernels) is bit-exact
against the reference's golden frames and against frames produced by the reference's libzstd."""
import hashlib
import random
import os

import pytest

from conftest import golden, XML_SHA256_PREFIX
from util import edge_inputs, needs_tuning_build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(zj):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    zj.batch.init(0)
    return zj


def test_native_library_is_the_one_running(gpu):
    L = gpu.lib()
    assert L.zjni_device_count() >= 1
    import ctypes as C
    a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert L.zjni_kernel_info(a, b, c, d) == 0
    assert a.value >= 256 and 8000 < b.value < 65536


def test_gpu_golden_frames_one_batch(gpu):
    # T/scala/Zstd.scala:427-638: CLI frames at levels 1/3/6/9/ultra, with and without content size, multi-block, multi-frame (every frame
    # /root/reference/src/test/resources holds: SURVEY.md section 8c)
    names = ["xml-1.zst", "xml-3.zst", "xml-6.zst", "xml-9.zst", "xml-advanced.zst", "xml-1-sized.zst", "xml-sized-combined.zst", "xmlsmall-sized.zst", "xml-1x2.zst", "xml-1-sizedx2.zst"]
    frames = [golden(n) for n in names]
    outs = gpu.decompress_batch(frames, [5_345_280] * 6 + [5_345_382, 102] + [2 * 5_345_280] * 2)
    for n, o in zip(names, outs):
        assert not isinstance(o, Exception), (n, o)
    for o in outs[:6]:
        assert len(o) == 5_345_280 and hashlib.sha256(o).hexdigest().startswith(XML_SHA256_PREFIX)
    assert outs[6][:102] == golden("xmlsmall") and hashlib.sha256(outs[6][102:]).hexdigest().startswith(XML_SHA256_PREFIX)
    assert outs[7] == golden("xmlsmall")
    for o in outs[8:]:
        assert len(o) == 2 * 5_345_280 and o[:5_345_280] == o[5_345_280:] and hashlib.sha256(o[:5_345_280]).hexdigest().startswith(XML_SHA256_PREFIX)


@pytest.mark.parametrize("level", [1, 3])
def test_gpu_edge_inputs(gpu, oracle_ref, level):
    items = edge_inputs()
    frames = [oracle_ref.compress(d, level) for _, d in items]
    outs = gpu.decompress_batch(frames, [len(d) for _, d in items])
    for (name, data), o in zip(items, outs):
        assert not isinstance(o, Exception), (name, o)
        assert o == data, name


def test_gpu_per_buffer_api_and_errors(gpu, oracle_ref):
    data = b"hello hello hello hello " * 100
    z = oracle_ref.compress(data, 3)
    assert gpu.Zstd.decompress(z, len(data)) == data
    dctx = gpu.ZstdDecompressCtx()
    with pytest.raises(gpu.ZstdException) as e:                    # T/scala/Zstd.scala:186-221
        dctx.decompress(z, len(data) - 1)
    assert e.value.getErrorCode() == gpu.Zstd.errDstSizeTooSmall()
    assert "Destination buffer is too small" in str(e.value)
    # short garbage is srcSize_wrong, long garbage prefix_unknown, garbage after a frame srcSize_wrong (zstd_decompress.c:966-979, :1136)
    for junk, code, name in ((b"\x00\x01\x02\x03\x04\x05\x06\x07", 72, "Src size is incorrect"), (bytes(range(20)), 10, "Unknown frame descriptor"),
                             (z + bytes(range(20)), 72, "Src size is incorrect")):
        with pytest.raises(gpu.ZstdException) as e:
            dctx.decompress(junk, len(data))
        assert e.value.getErrorCode() == code
        with pytest.raises(oracle_ref.ZstdRefError, match=name):
            oracle_ref.decompress(junk, len(data))
    with pytest.raises(gpu.ZstdException):
        dctx.decompress(z[:-3], len(data))
    # offsets into larger arrays: J/ZstdDecompressCtx.java:239 decompressByteArray
    dst = bytearray(len(data) + 20)
    src = b"\xAA" * 7 + z + b"\xBB" * 5
    n = dctx.decompressByteArray(dst, 10, len(data), src, 7, len(z))
    assert n == len(data) and bytes(dst[10:10 + n]) == data and dst[:10] == bytes(10)


def test_gpu_host_batch_capacities_and_pipeline(gpu, oracle_ref):
    """zjni_decompress_batch from host pointers: the three-stage pipeline over slices (more than one slice here), destinations larger than the
    content (staging is sized by the frames' own content sizes where a buffer is one frame that states it), concatenated frames, an empty frame and a damaged frame whose header understates its content — answered like the reference decoding into the caller's capacity."""
    rnd = random.Random(12)
    datas = [gpu.synth_host(65536, k, 1) for k in range(6000)]                      # ~375 MiB of output: several 256 MiB-class slices
    frames = [oracle_ref.compress(d, 1 + k % 3) for k, d in enumerate(datas)]
    caps = [len(d) * (1 + k % 4) + (k % 7) for k, d in enumerate(datas)]            # up to 4x the content and odd
    extra_d, extra_f, extra_c, want = [], [], [], []
    a, b = gpu.synth_host(30000, 77, 1), gpu.synth_host(20000, 78, 1)
    extra_f.append(oracle_ref.compress(a, 3) + oracle_ref.compress(b, 1)); extra_c.append(60000); want.append(a + b)               # two frames in one buffer
    extra_f.append(oracle_ref.compress(b"", 3)); extra_c.append(16); want.append(b"")
    z = bytearray(oracle_ref.compress(a, 3)); assert z[4] & 0xC0 == 0x40 and z[4] & 0x20                                           # single segment, 2-byte content size
    z[5:7] = (int.from_bytes(z[5:7], "little") - 300).to_bytes(2, "little")                                                       # header now says 300 bytes fewer than the blocks hold
    try:
        oracle_ref.decompress(bytes(z), 200000); code = None
    except oracle_ref.ZstdRefError as e:
        code = e.code
    assert code is not None
    extra_f.append(bytes(z)); extra_c.append(200000); want.append(-code)
    outs = gpu.decompress_batch(frames + extra_f, caps + extra_c)
    for k, d in enumerate(datas):
        assert outs[k] == d, k
    for o, w in zip(outs[len(datas):], want):
        if isinstance(w, int): assert isinstance(o, Exception) and o.getErrorCode() == -w, (o, w)
        else: assert o == w


@pytest.mark.parametrize("size,count", [(4096, 2048), (65536, 1024), (131072, 256)])
def test_gpu_device_batch_synthetic(gpu, oracle_port, size, count):
    """BASELINE configs' buffer shapes (4 KiB / 64 KiB / 128 KiB): reference-compressed L3 frames of
    the §8(d) mixed-entropy set, decoded on the GPU from HBM-resident blobs, all buffers compared."""
    import numpy as np
    import torch
    raw = gpu.synth_host(size, 0, count)
    frames = oracle_port.compress_many(raw, size, 3, os.cpu_count() or 4)
    blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
    off = np.zeros(count + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(f) for f in frames])
    d_src = torch.from_numpy(blob.copy()).cuda()
    d_soff = torch.from_numpy(off).cuda()
    d_dst = torch.zeros(count * size, dtype=torch.uint8, device="cuda")
    d_doff = gpu.batch.uniform_offsets(count, size, "cuda")
    res = gpu.batch.decompress(d_src, d_soff, d_dst, d_doff)
    torch.cuda.synchronize()
    assert bool((res == size).all()), res[res != size][:8]
    want = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
    assert torch.equal(d_dst, want)
    # device-side generator produces the same bytes as the host-side one
    gen = gpu.batch.synth(count, size, 0)
    torch.cuda.synchronize()
    assert torch.equal(gen, want)


@pytest.fixture
def force_split(monkeypatch):
    monkeypatch.setenv("ZJNI_DSPLIT_MIN", "1")        # every batch through prep -> lane sequence decode -> execute


@pytest.mark.parametrize("lit_pass", ["on", "off", "slots16k"])
def test_gpu_split_pipeline_mixed_batch(gpu, oracle_ref, oracle_port, force_split, monkeypatch, lit_pass):
    """three-stage decoder: simple frames (one block, <= 64 KiB) next to frames it must hand to the fused kernel
    (multi-block, multi-frame, raw blocks, truncated, corrupted) in ONE batch; results == fused path == reference.
    lit_pass: stage 2b (zd_lit_frame: Huffman literals regenerated beside the sequence decode) on (the default), off
    (ZJNI_DEC_LIT=0), and with 16 KiB slots so that the larger frames' literals stay with the execution kernel."""
    import random
    if lit_pass == "off": monkeypatch.setenv("ZJNI_DEC_LIT", "0")
    rnd = random.Random(17)
    items, caps = [], []
    for name, data in edge_inputs():
        for level in (1, 3):
            items.append(oracle_ref.compress(data, level)); caps.append(len(data))
    raw = gpu.synth_host(65536, 0, 64)
    for i in range(64):
        d = raw[i * 65536:(i + 1) * 65536][: rnd.choice([65536, 65536, 40000, 5000, 300, 17])]
        items.append(oracle_ref.compress(d, rnd.choice([1, 3]))); caps.append(len(d))
    items.append(golden("xml-sized-combined.zst")); caps.append(5_345_382)
    good = oracle_ref.compress(raw[:65536], 3)
    for cut in (1, 5):
        items.append(good[:-cut]); caps.append(65536)
    for pos in (7, 20, len(good) // 2, len(good) - 2):
        bad = bytearray(good); bad[pos] ^= 0x5A
        items.append(bytes(bad)); caps.append(65536)
    items.append(good); caps.append(65535)               # destination one byte short
    if lit_pass == "slots16k": needs_tuning_build(gpu)
    if lit_pass == "slots16k": monkeypatch.setenv("ZJNI_DEC_LIT_BYTES", str(16384 * len(items) + 4096))
    split = gpu.decompress_batch(items, caps)
    import os
    os.environ["ZJNI_DSPLIT_MIN"] = "1000000000"
    fused = gpu.decompress_batch(items, caps)
    for k, (a, b) in enumerate(zip(split, fused)):
        if isinstance(b, Exception):
            assert isinstance(a, Exception) and a.getErrorCode() == b.getErrorCode(), (k, a, b)
        else:
            assert a == b, k
    for k, (z, cap) in enumerate(zip(items[: len(items) - 8], caps)):
        assert split[k] == oracle_ref.decompress(z, cap), k


def test_gpu_split_pipeline_device_batch(gpu, oracle_port, force_split):
    import numpy as np
    import torch
    size, count = 65536, 2048
    raw = gpu.synth_host(size, 0, count)
    frames = oracle_port.compress_many(raw, size, 3, os.cpu_count() or 4)
    blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
    off = np.zeros(count + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(f) for f in frames])
    d_src = torch.from_numpy(blob.copy()).cuda(); d_soff = torch.from_numpy(off).cuda()
    d_dst = torch.zeros(count * size, dtype=torch.uint8, device="cuda")
    res = gpu.batch.decompress(d_src, d_soff, d_dst, gpu.batch.uniform_offsets(count, size, "cuda"))
    torch.cuda.synchronize()
    assert bool((res == size).all()), res[res != size][:8]
    assert torch.equal(d_dst, torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda())


@pytest.mark.parametrize("split_min", ["1", "1000000000"])
def test_gpu_dictionary_decode(gpu, oracle_ref, monkeypatch, split_min):
    """ZstdDictDecompress + ZstdDecompressCtx.loadDict / Zstd.decompress(src, dict, size) (T/scala/ZstdDict.scala:58-216):
    reference-made dictionary frames (trained dictionary and raw-content dictionary) decode bit-exactly in one batch;
    missing / wrong dictionaries give the reference's error codes"""
    import random
    from util import json_records
    monkeypatch.setenv("ZJNI_DSPLIT_MIN", split_min)          # three-stage pipeline / fused kernel
    rnd = random.Random(5)
    trained = oracle_ref.train_dict(json_records(2000), 16384)
    raw = b"".join(json_records(40, seed=9, first=7000))
    for dbytes in (trained, raw):
        with gpu.ZstdDictDecompress(dbytes) as dd:
            if dbytes is trained:
                assert dd.getDictID() == oracle_ref.dict_id(trained)
            datas, frames = [], []
            for _ in range(200):
                k = rnd.choice([1, 1, 2, 5, 30, 300])
                data = b"".join(json_records(k, seed=rnd.randrange(1000), first=rnd.randrange(100000)))
                if rnd.random() < 0.15:
                    data = gpu.synth_host(rnd.randrange(1, 70000), rnd.randrange(1000), 1)
                datas.append(data); frames.append(oracle_ref.compress_using_dict(data, dbytes, rnd.choice([1, 3, 5, 9])))
            a, b = datas[0], datas[1]
            datas.append(a + b); frames.append(frames[0] + frames[1])        # two frames in one buffer
            outs = gpu.decompress_batch(frames, [len(d) for d in datas], dd)
            for k, (d, o) in enumerate(zip(datas, outs)):
                assert not isinstance(o, Exception), (k, o)
                assert o == d, k
            # per-buffer API
            with gpu.ZstdDecompressCtx() as ctx:
                ctx.loadDict(dd)
                assert ctx.decompress(frames[3], len(datas[3])) == datas[3]
                ctx.loadDict(dbytes)
                assert ctx.decompress(frames[4], len(datas[4])) == datas[4]
            assert gpu.Zstd.decompress(frames[5], dbytes, len(datas[5])) == datas[5]
    data = b"".join(json_records(5, first=3))
    z = oracle_ref.compress_using_dict(data, trained, 3)
    with pytest.raises(gpu.ZstdException) as e:
        gpu.Zstd.decompress(z, len(data))                                # frame names a dictionary, none loaded
    assert e.value.getErrorCode() == 32
    other = oracle_ref.train_dict(json_records(2000, seed=4, first=50000), 8192)
    if oracle_ref.dict_id(other) != oracle_ref.dict_id(trained):
        with pytest.raises(gpu.ZstdException) as e:
            gpu.Zstd.decompress(z, other, len(data))
        assert e.value.getErrorCode() == 32
    with gpu.ZstdDictDecompress(trained) as dd:                           # dictionary loaded, plain frames still fine
        plain = [oracle_ref.compress(d, 3) for d in (data, data * 7)]
        assert gpu.decompress_batch(plain, [len(data), 7 * len(data)], dd) == [data, data * 7]


def test_gpu_streamed_frames_without_content_size(gpu, oracle_ref):
    """ZstdOutputStream's output (no content size in the header, flushed blocks, optional checksum) through both pipelines"""
    from util import json_records
    data = b",".join(json_records(12000, seed=21))[:400_000]
    frames = [oracle_ref.compress_stream(data, level, checksum, chunk=30000, flush_every=fe)
              for level, checksum, fe in ((1, False, 0), (3, True, 1), (3, False, 3), (9, True, 0))]
    for split_min in ("1", "1000000000"):
        os.environ["ZJNI_DSPLIT_MIN"] = split_min
        try:
            outs = gpu.decompress_batch(frames * 2, [len(data)] * 4 + [len(data) + 1000] * 4)
            assert all(o == data for o in outs)
            short = gpu.decompress_batch(frames, [len(data) - 1] * 4)
            assert all(isinstance(o, Exception) and o.getErrorCode() == 70 for o in short)
        finally:
            os.environ.pop("ZJNI_DSPLIT_MIN", None)


def test_gpu_four_huffman_streams_by_the_whole_wave(gpu, oracle_ref, monkeypatch):
    """the GPU twin of tests/test_emu_decode.py::test_four_huffman_streams_by_the_whole_wave: zd_huf_streams_wave on real lanes (zj_dec_lit_kernel beside the sequence
    decode; zj_dec_lit_mb_kernel for the block stages) — code tables of one length, geometric, near-uniform with a rare long code; damaged streams answer what the
    reference's portable decoder answers (bytes or refusal), whichever kernel ends up deciding"""
    from util import skewed_literal_inputs
    monkeypatch.setenv("ZJNI_DSPLIT_MIN", "1")              # the three-stage pipeline whatever the batch size
    rnd = random.Random(12)
    frames, caps, want = [], [], []
    for d in skewed_literal_inputs():
        for level in (1, 3):
            z = oracle_ref.compress(d, level, level == 1)
            frames.append(z); caps.append(len(d)); want.append(d)
            for _ in range(4):
                zb = bytearray(z); p = rnd.randrange(12, len(zb)); zb[p] ^= 1 << rnd.randrange(8); zb = bytes(zb)
                try: w = oracle_ref.decompress_portable(zb, len(d))
                except oracle_ref.ZstdRefError as ex: w = -ex.code
                frames.append(zb); caps.append(len(d)); want.append(w)
    order = list(range(len(frames))); rnd.shuffle(order)
    outs = gpu.decompress_batch([frames[i] for i in order], [caps[i] for i in order])
    for j, i in enumerate(order):
        got = -abs(outs[j].getErrorCode()) if isinstance(outs[j], Exception) else outs[j]
        assert got == want[i], (i, caps[i], want[i] if isinstance(want[i], int) else "bytes", got if isinstance(got, int) else "bytes differ")
