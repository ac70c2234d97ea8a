"""GPU (-m gpu): the stream route — zjni_compress_stream / zjni_compress_stream_batch_device (ze_compress_stream on zj_encode_stream_kernel) against
ZSTD_compressStream2 without a pledged size (oracle/ref.py compress_stream = what ZstdDirectBufferCompressingStream / ZstdOutputStream write;
N/jni_directbuffercompress_zstd.c:97-161), byte for byte: the cases of tests/test_emu_stream.py on the device."""
import random

import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(zj):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    zj.batch.init(0)
    return zj


def test_gpu_stream_frames_any_total(gpu, oracle_ref):
    rnd = random.Random(3)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    noise = bytes(rnd.getrandbits(8) for _ in range(70000))
    n = 0
    for size in (0, 1, 6, 7, 8, 100, 4096, 65536, 100000, 131071, 131072, 131073, 200000, 262144, 262145, 393216, 393217, 600000, 1048576, 2000000, 2097152):
        o = rnd.randrange(0, len(xml) - size - 1)
        inputs = [xml[o:o + size], b"".join(gpu.synth_host(65536, i, 1) for i in range(size // 65536 + 1))[:size], (noise * 40)[:size]]
        for d in inputs:
            for level in (3, 1, 2):
                ck = bool(n & 1); n += 1
                if size > (1 << (18 + level)):
                    with pytest.raises(gpu.ZstdException) as e:
                        gpu.compress_stream(d, level, ck)
                    assert e.value.getErrorCode() == 201
                    continue
                got = gpu.compress_stream(d, level, ck)
                assert got == oracle_ref.compress_stream(d, level, ck), (size, level, ck)
                assert oracle_ref.decompress(got, max(size, 1)) == d


def test_gpu_stream_frames_with_flushes_and_prefixes(gpu, oracle_ref):
    rnd = random.Random(5)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    n = 0
    for size in (1000, 50000, 131072, 200000, 262144, 300000, 524288):
        o = rnd.randrange(0, len(xml) - size - 1)
        for d in (xml[o:o + size], b"".join(gpu.synth_host(65536, i, 1) for i in range(size // 65536 + 1))[:size]):
            for level in (3, 1):
                if size > (1 << (18 + level)): continue
                for chunk, k in ((50000, 1), (50000, 2), (10000, 3), (131072, 1), (65536, 2), (1000, 7)):
                    calls = (size + chunk - 1) // chunk
                    flushes = [min(j * chunk, size) for j in range(1, calls + 1) if j % k == 0]
                    ck = bool(n & 1); n += 1
                    full = gpu.compress_stream(d, level, ck, flush_at=flushes)
                    assert full == oracle_ref.compress_stream(d, level, ck, chunk=chunk, flush_every=k), (size, level, chunk, k)
                    if flushes and n % 3 == 0:        # flushed, not closed: the frame's beginning, reproduced when the rest arrives
                        part = gpu.compress_stream(d[:flushes[-1]], level, ck, flush_at=flushes, final=False)
                        assert part and full.startswith(part), (size, level, chunk, k)
    assert gpu.compress_stream(b"", 3, False, final=False, known_empty=False) == b""


def test_gpu_stream_batch_on_the_device(gpu, oracle_ref):
    """n streams in one launch (zjni_compress_stream_batch_device): no flushes, all final"""
    import torch
    rnd = random.Random(9)
    sizes = [rnd.choice([0, 5000, 70000, 131072, 140000, 300000, 400000]) for _ in range(24)]
    datas = [gpu.synth_host(65536, 40 + i, 7)[:s] for i, s in enumerate(sizes)]
    dev = "cuda"
    blob = torch.tensor(list(b"".join(datas)) or [0], dtype=torch.uint8, device=dev) if sum(sizes) < (1 << 22) else torch.frombuffer(bytearray(b"".join(datas)), dtype=torch.uint8).to(dev)
    off = torch.tensor([0] + list(__import__("itertools").accumulate(sizes)), dtype=torch.int64, device=dev)
    caps = [s + (s >> 8) + 4096 for s in sizes]
    doff = torch.tensor([0] + list(__import__("itertools").accumulate(caps)), dtype=torch.int64, device=dev)
    dst = torch.zeros(sum(caps), dtype=torch.uint8, device=dev)
    res = torch.zeros(len(sizes), dtype=torch.int64, device=dev)
    mode = torch.tensor([1 | (2 if s == 0 else 0) for s in sizes], dtype=torch.int32, device=dev)
    r = gpu.lib().zjni_compress_stream_batch_device(blob.data_ptr(), off.data_ptr(), dst.data_ptr(), doff.data_ptr(), res.data_ptr(), len(sizes), 3, 1, None, None, mode.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream)
    assert r == 0
    torch.cuda.synchronize()
    out = dst.cpu().numpy().tobytes(); rs = res.cpu().tolist(); dl = doff.cpu().tolist()
    for i, d in enumerate(datas):
        assert rs[i] > 0
        assert out[dl[i]:dl[i] + rs[i]] == oracle_ref.compress_stream(d, 3, True), (i, sizes[i])
