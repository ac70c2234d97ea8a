"""GPU (-m gpu): batched compress through the C-ABI.  Gates:
  (ii)  CPU libzstd decodes every GPU-produced frame to the original,
  (iii) sum(csize_gpu) <= 1.01 * sum(csize_cpu) at the same level,
and the stronger statement this implementation makes: GPU frames are BYTE-IDENTICAL to the
reference's ZSTD_compress2 at levels 1-3 with nothing else set; ZstdCompressCtx.setHashLog(14).setChainLog(13)
selects the LDS-sized level-3 tables (wave-per-frame matcher, fused kernel) = the reference given the same two parameters."""
import os
import random

import pytest

from conftest import golden
from util import needs_tuning_build, edge_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(zj):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    zj.batch.init(0)
    return zj


def ref_expected(ref, data, level, lds=False):
    return ref.compress(data, 3, False, 14, 13) if (level == 3 and lds) else ref.compress(data, level)


LDS = dict(hash_log=14, chain_log=13)          # level 3 on the LDS-sized tables: the small-batch finders (fused kernel, wave-per-frame matcher)


@pytest.mark.parametrize("level", [1, 2, 3, "3lds"])
def test_gpu_frames_byte_identical_on_edge_inputs(gpu, oracle_ref, level):
    lds = level == "3lds"; level = 3 if lds else level
    items = [(n, d) for n, d in edge_inputs()]
    outs = gpu.compress_batch([d for _, d in items], level, **(LDS if lds else {}))
    for (name, data), z in zip(items, outs):
        assert not isinstance(z, Exception), (name, z)
        assert oracle_ref.decompress(z, len(data)) == data, name
        assert z == ref_expected(oracle_ref, data, level, lds), name


def test_gpu_small_inputs_identical_to_default_level3(gpu, oracle_ref):
    rnd = random.Random(5)
    datas = []
    for _ in range(200):
        size = rnd.randrange(0, 8193)
        datas.append(gpu.synth_host(size, rnd.randrange(0, 100000), 1) if size else b"")
    outs = gpu.compress_batch(datas, 3)
    for d, z in zip(datas, outs):
        assert z == oracle_ref.compress(d, 3), len(d)
    assert outs[datas.index(b"")] == bytes.fromhex("28b52ffd2000010000") if b"" in datas else True
    assert gpu.Zstd.compress(golden("xmlsmall"), 3) == golden("xmlsmall-sized.zst")   # reference encode KAT


def test_gpu_mixed_sizes_use_both_lds_passes(gpu, oracle_ref):
    # level-1 inputs of 8-16 KiB need 2^15-entry tables -> deferred to the large-LDS pass
    rnd = random.Random(9)
    datas = [gpu.synth_host(s, rnd.randrange(0, 1000), 1) for s in (100, 4096, 9000, 12000, 16384, 16385, 40000, 65536, 65537, 100000, 131072)]
    for level, lds in ((1, False), (2, False), (3, False), (3, True)):
        outs = gpu.compress_batch(datas, level, **(LDS if lds else {}))
        for d, z in zip(datas, outs):
            assert not isinstance(z, Exception), (level, len(d), z)
            assert z == ref_expected(oracle_ref, d, level, lds), (level, lds, len(d))


@pytest.mark.parametrize("wide_slice", ["32768", "100"])
def test_gpu_wide_frames_through_the_lane_pipeline(gpu, oracle_ref, monkeypatch, wide_slice):
    """frames > 64 KiB (config 5's 128 KiB buffers) and the level-1/2 frames with larger tables take the lane-per-frame
    pipeline too (4-byte positions), in slices; every frame byte-identical to the reference"""
    monkeypatch.setenv("ZJNI_SPLIT_MIN", "1")
    monkeypatch.setenv("ZJNI_WIDE_SLICE", wide_slice)
    rnd = random.Random(31)
    sizes = [131072, 100000, 65537, 70000, 12000, 16384, 9000, 4096, 65536, 50, 0] * 28
    datas = [gpu.synth_host(s, rnd.randrange(0, 100000), 1) if s else b"" for s in sizes]
    for level, lds in ((1, False), (2, False), (3, False), (3, True)):
        outs = gpu.compress_batch(datas, level, **(LDS if lds else {}))
        for k, (d, z) in enumerate(zip(datas, outs)):
            assert not isinstance(z, Exception), (level, k, len(d), z)
            assert z == ref_expected(oracle_ref, d, level, lds), (level, lds, k, len(d))
        assert gpu.decompress_batch(outs, [len(d) for d in datas]) == datas
        if level == 3 and not lds:          # the library says how it split the batch (zjni_last_lists): frames above 64 KiB are the wide launch's (ZJNI_ROUTE_WIDE)
            import ctypes as C
            l3 = (C.c_uint * 3)()
            assert gpu.lib().zjni_last_lists(l3) == 0
            assert l3[1] == sum(1 for s_ in sizes if s_ > 65536) and l3[0] + l3[1] + l3[2] == len(sizes), list(l3)
            assert gpu.lib().zjni_route_kernel(10) == b"zj_enc_match_wide_kernel"


@pytest.mark.parametrize("need", ["1", "2", "0"])
def test_gpu_wide_frames_with_need_flags(gpu, oracle_ref, monkeypatch, need):
    """the wide launch (frames of 64 KiB + 1 .. 128 KiB) with need flags (zn_flags_frame_wide beside the run machine): flags for every frame (ZJNI_NEED=1), for
    the frames zj_enc_worth_kernel picks (2, the default), for none (0: ZLaneD as before) — every frame the reference's, whichever frames got flags and whenever"""
    monkeypatch.setenv("ZJNI_SPLIT_MIN", "1")
    monkeypatch.setenv("ZJNI_NEED", need)
    rnd = random.Random(37)
    sizes = [131072, 100000, 65537, 131071, 90000, 70000] * 40
    datas = [gpu.synth_host(s, rnd.randrange(0, 100000), 1) for s in sizes]
    datas += [bytes(rnd.randrange(16) for _ in range(131072)), bytes(rnd.randrange(4) for _ in range(100000)), bytes([5]) * 131072, bytes(rnd.getrandbits(8) for _ in range(80000))]
    for rep in range(2):                     # (the second call meets tables and flag buffers the first one left)
        outs = gpu.compress_batch(datas, 3)
        for k, (d, z) in enumerate(zip(datas, outs)):
            assert not isinstance(z, Exception), (k, len(d), z)
            assert z == oracle_ref.compress(d, 3), (need, rep, k, len(d))
    assert gpu.decompress_batch(outs, [len(d) for d in datas]) == datas


def test_gpu_explicit_table_sizes(gpu, oracle_ref):
    """ZstdCompressCtx.setHashLog / setChainLog (level 3): byte-identical to the reference given the same two parameters;
    16 / 15 = the reference's plain level 3.  Batches and the per-buffer API; other levels refuse the parameters."""
    rnd = random.Random(41)
    sizes = [131072, 100000, 65536, 65536, 40000, 12000, 4096, 700, 64, 10, 0] * 6
    datas = [gpu.synth_host(s, rnd.randrange(0, 100000), 1) if s else b"" for s in sizes]
    for hl, cl in ((16, 15), (17, 16), (12, 12), (0, 15)):
        outs = gpu.compress_batch(datas, 3, hash_log=hl, chain_log=cl)
        for k, (d, z) in enumerate(zip(datas, outs)):
            assert not isinstance(z, Exception), (hl, cl, k, len(d), z)
            assert z == oracle_ref.compress(d, 3, False, hl, cl), (hl, cl, k, len(d))
            if (hl, cl) == (16, 15):
                assert z == oracle_ref.compress(d, 3)
        assert gpu.decompress_batch(outs, [len(d) for d in datas]) == datas
    with gpu.ZstdCompressCtx() as ctx:
        ctx.setLevel(3).setHashLog(16).setChainLog(15).setChecksum(True)
        assert ctx.compress(datas[2]) == oracle_ref.compress(datas[2], 3, True)
        ctx.setLevel(1)
        with pytest.raises(gpu.ZstdException) as e:
            ctx.compress(datas[2])
        assert e.value.getErrorCode() == 40
        ctx.setLevel(3).setHashLog(25)
        with pytest.raises(gpu.ZstdException) as e:
            ctx.compress(datas[2])
        assert e.value.getErrorCode() == 42


def test_gpu_huf_sort_count_164(gpu, oracle_ref):
    """the input of tests/test_emu_encode.py::test_huf_sort_visits_the_count_164_slot through the C-ABI"""
    from conftest import golden
    d = golden("huf_sort_count164.bin")
    for level in (1, 2, 3):
        assert gpu.compress_batch([d], level)[0] == ref_expected(oracle_ref, d, level), level


def test_gpu_equal_literal_counts(gpu, oracle_ref):
    """tests/test_emu_encode.py::test_equal_literal_counts through the C-ABI (fused kernel and lane pipeline)"""
    from util import equal_count_inputs
    items = [d for _, d in equal_count_inputs()]
    for level in (1, 3):
        for batch in (items, items * 500):                # 9 frames: fused kernel; 4 500: lane-per-frame pipeline
            outs = gpu.compress_batch(batch, level)
            for k, (d, z) in enumerate(zip(batch[:9], outs[:9])):
                assert z == ref_expected(oracle_ref, d, level), (level, len(batch), k)
            assert all(outs[k] == outs[k % 9] for k in range(len(outs)))


def test_gpu_level_zero_is_the_default_level(gpu, oracle_ref):
    """ZSTD_c_compressionLevel = 0 means ZSTD_CLEVEL_DEFAULT (3): ZstdCompressCtx.setLevel(0), ZstdDictCompress(dict, 0)"""
    d = gpu.synth_host(30000, 5, 1)
    with gpu.ZstdCompressCtx() as ctx:
        assert ctx.setLevel(0).compress(d) == ref_expected(oracle_ref, d, 3)
    dbytes = gpu.synth_host(20000, 9, 1)
    with gpu.ZstdDictCompress(dbytes, 0) as cd:
        assert gpu.compress_batch([d[:4000]], dictionary=cd)[0] == oracle_ref.CDict(dbytes, 0).compress(d[:4000])


def test_gpu_per_buffer_api_and_errors(gpu, oracle_ref):
    data = gpu.synth_host(30000, 1, 1)
    ctx = gpu.ZstdCompressCtx().setLevel(3)
    z = ctx.compress(data)
    assert oracle_ref.decompress(z, len(data)) == data
    assert gpu.Zstd.getFrameContentSize(z) == len(data)             # T/scala/Zstd.scala:26-36
    assert gpu.Zstd.decompress(z, len(data)) == data
    small = bytearray(len(z) - 1)
    with pytest.raises(gpu.ZstdException) as e:                       # T/scala/Zstd.scala:186-201
        ctx.compress(data, small)
    assert e.value.getErrorCode() == gpu.Zstd.errDstSizeTooSmall()
    assert gpu.Zstd.compress(b"x" * 131073, 3) == oracle_ref.compress(b"x" * 131073, 3)      # a multi-block frame (tests/test_gpu_multiblock.py)
    with pytest.raises(gpu.ZstdException) as e:
        gpu.Zstd.compress(b"x" * ((2 << 20) + 1), 3)
    assert e.value.getErrorCode() == 201                              # beyond ZJNI_FRAME_MAX: the CPU path's
    # offsets: J/ZstdCompressCtx.java:691 compressByteArray
    dst = bytearray(40000)
    n = ctx.compressByteArray(dst, 100, 39000, b"\x01" * 5 + data + b"\x02" * 3, 5, len(data))
    assert bytes(dst[100:100 + n]) == z and dst[:100] == bytes(100)


@pytest.mark.parametrize("machine", ["run", "lane"])
@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_gpu_need_gated_double_fast(gpu, oracle_ref, monkeypatch, mode, machine):
    """zj_need.h + zj_match_run.h: the flag kernels beside the match kernel give the same frames — flags for every frame (ZJNI_NEED=1), for the
    frames the worth kernel picks (2, the default), for none (0) — on the run machine (the product's route, and what zjni_last_route reports)
    and on the previous lane machine (ZJNI_LANE_MACHINE=0)"""
    if machine == "lane": needs_tuning_build(gpu)
    monkeypatch.setenv("ZJNI_NEED", mode)
    monkeypatch.setenv("ZJNI_SPLIT_MIN", "1")
    if machine == "lane": monkeypatch.setenv("ZJNI_LANE_MACHINE", "0")
    rnd = random.Random(43)
    datas = [gpu.synth_host(65536, k, 1) for k in range(96)] + [gpu.synth_host(s, 100 + s, 1) for s in (64, 65, 1000, 8192, 8193, 30000, 65535, 63, 0)]
    datas += [bytes([7]) * 40000, bytes(rnd.getrandbits(8) for _ in range(20000)), (b"abcdefgh" * 5000)[:33333], golden("xmlsmall")[:60000]]
    outs = gpu.compress_batch(datas, 3)
    for d, z in zip(datas, outs):
        assert z == oracle_ref.compress(d, 3), len(d)
    route = gpu.lib().zjni_last_route()
    assert route == {("run", "0"): 5, ("run", "1"): 6, ("run", "2"): 6, ("lane", "0"): 3, ("lane", "1"): 4, ("lane", "2"): 4}[(machine, mode)], route
    for hl, cl in ((15, 15), (14, 13)):
        outs = gpu.compress_batch(datas, 3, hash_log=hl, chain_log=cl)
        for d, z in zip(datas, outs):
            assert z == oracle_ref.compress(d, 3, False, hl, cl), (len(d), hl, cl)


@pytest.mark.parametrize("n_min", [1, 5000])
def test_gpu_tight_destinations(gpu, oracle_ref, monkeypatch, n_min):
    """destinations between a frame's size and Zstd.compressBound: the reference wants working room (8 bytes of slack behind every bit
    stream, two-byte stores of table descriptions, 18 bytes for any header) and answers dstSize_tooSmall although the frame would fit, or
    emits the block raw when that fits — the same answer, code or bytes, for every capacity, on the small-batch (fused) kernels and on
    the large-batch pipeline (tests/test_emu_encode.py::test_emu_tight_destinations on the lane-serial build; T/scala/Zstd.scala:186-201)"""
    if n_min > 1: monkeypatch.setenv("ZJNI_SPLIT_MIN", "1")          # the lane match finder + entropy kernel route regardless of batch size
    rnd = random.Random(78)
    def barely(seed):
        r = random.Random(seed)
        n = r.randrange(20, 400); a = r.choice([3, 6, 12, 24, 48, 100]); d = bytearray(r.randrange(a) for _ in range(n))
        for _ in range(r.choice([0, 1, 2, 4])):
            ln = r.randrange(4, 12); at = r.randrange(0, max(1, n - 2 * ln)); to = r.randrange(at + ln, max(at + ln + 1, n - ln + 1))
            d[to:to + ln] = d[at:at + ln]
        return bytes(d[:n])
    inputs = [barely(k) for k in (2, 32, 49, 57)] + [barely(1000 + k) for k in range(12)] + [b"", b"a", b"abcdefg" * 3, bytes(rnd.getrandbits(8) for _ in range(300)),
              golden("xmlsmall")[:3000], gpu.synth_host(9000, 5, 1), gpu.synth_host(65536, 1, 1), gpu.synth_host(65536, 7, 1), b"\x07" * 5000, gpu.synth_host(140000, 3, 1)]
    refused = raw = 0
    for level in (1, 3, 5):
        for ck in (False, True):
            datas, caps, wants = [], [], []
            for data in inputs:
                if level >= 5 and len(data) > 16384: continue
                hl, cl = 0, 0
                full = oracle_ref.compress(data, level, ck, hl, cl)
                cs = list(range(max(0, len(full) - 2), len(full) + 24)) + [0, 8, 17, 18, len(data), len(data) + 3, len(data) + 9, len(data) + 12, len(data) + 20]
                for cap in (cs if len(data) < 20000 else cs[::4]):
                    try: want = oracle_ref.compress(data, level, ck, hl, cl, cap=cap)
                    except oracle_ref.ZstdRefError as e: want = -e.code
                    datas.append(data); caps.append(cap); wants.append(want)
                    refused += isinstance(want, int) and cap >= len(full)
                    raw += (not isinstance(want, int)) and want != full
            outs = gpu.compress_batch(datas, level, ck, capacities=caps)
            for d, cap, want, z in zip(datas, caps, wants, outs):
                got = -z.getErrorCode() if isinstance(z, Exception) else z
                assert got == want, (level, ck, len(d), cap, want if isinstance(want, int) else len(want), got if isinstance(got, int) else len(got))
    assert refused > 500 and raw > 0, (refused, raw)


@pytest.mark.parametrize("level", [1, 3])
@pytest.mark.parametrize("size,count", [(4096, 2048), (65536, 1024), (131072, 256)])
def test_gpu_device_batch_roundtrip_and_ratio(gpu, oracle_port, oracle_ref, level, size, count):
    """BASELINE config shapes: compress on the GPU from HBM-resident blobs, decompress on the GPU,
    compare every byte; CPU reference decodes a sample; ratio gate against the CPU at the SAME level."""
    import numpy as np
    import torch
    B = gpu.batch
    src = B.synth(count, size, 0)
    soff = B.uniform_offsets(count, size, "cuda")
    bound = gpu.Zstd.compressBound(size)
    comp = torch.zeros(count * bound, dtype=torch.uint8, device="cuda")
    coff = B.uniform_offsets(count, bound, "cuda")
    csz = B.compress(src, soff, comp, coff, level)
    packed, poff = B.pack(csz, comp, coff)
    back = torch.zeros(count * size, dtype=torch.uint8, device="cuda")
    dsz = B.decompress(packed, poff, back, soff)
    torch.cuda.synchronize()
    assert bool((csz > 0).all()) and bool((dsz == size).all())
    assert torch.equal(back, src)                                     # gate (i)
    raw = gpu.synth_host(size, 0, count)
    hp, ho = packed.cpu().numpy(), poff.cpu().numpy()
    for i in range(0, count, max(1, count // 64)):                    # gate (ii) on a sample
        f = hp[ho[i]:ho[i + 1]].tobytes()
        assert oracle_ref.decompress(f, size) == raw[i * size:(i + 1) * size]
        assert f == ref_expected(oracle_ref, raw[i * size:(i + 1) * size], level)
    cpu_frames = oracle_port.compress_many(raw, size, level, os.cpu_count() or 4)     # default CPU level
    cpu_total = sum(len(f) for f in cpu_frames)
    assert int(csz.sum().item()) <= 1.01 * cpu_total, (int(csz.sum().item()), cpu_total)   # gate (iii)


@pytest.mark.parametrize("split_min", ["1", "1000000000"])
def test_gpu_checksum_flag(gpu, oracle_ref, monkeypatch, split_min):
    """ZstdCompressCtx.setChecksum(true) / Zstd.compress(src, level, checksumFlag): frames byte-identical to the
    reference's with ZSTD_c_checksumFlag, the GPU decoders verify the checksum (T/scala/Zstd.scala checksum cases)"""
    monkeypatch.setenv("ZJNI_SPLIT_MIN", split_min)
    monkeypatch.setenv("ZJNI_DSPLIT_MIN", split_min)
    items = [d for _, d in edge_inputs()] + [gpu.synth_host(65536, k, 1) for k in range(8)] + [gpu.synth_host(4097, 9, 1)]
    for level in (1, 3):
        outs = gpu.compress_batch(items, level, checksum=True)
        for d, z in zip(items, outs):
            assert not isinstance(z, Exception), (len(d), z)
            want = oracle_ref.compress(d, level, True)
            assert z == want, (len(d), level)
        back = gpu.decompress_batch(outs, [len(d) for d in items])
        assert back == items
        broken = [bytes(z[:-1]) + bytes([z[-1] ^ 0x80]) for z in outs]
        for d, r in zip(items, gpu.decompress_batch(broken, [len(d) for d in items])):
            assert isinstance(r, Exception) and r.getErrorCode() == 22, (len(d), r)
    with gpu.ZstdCompressCtx() as ctx:
        z = ctx.setLevel(1).setChecksum(True).compress(items[-1])
        assert z == oracle_ref.compress(items[-1], 1, True)
    assert gpu.Zstd.compress(items[-2], 3, True) == oracle_ref.compress(items[-2], 3, True)


def test_gpu_batch_larger_than_one_scratch_slice(gpu, oracle_ref):
    """batches beyond 65 536 buffers run through the per-frame scratch in slices (config 4 has 2^20 records)"""
    import torch
    n, size = 70001, 512
    src = gpu.batch.synth(n, size, 0)
    soff = gpu.batch.uniform_offsets(n, size, "cuda")
    bound = gpu.Zstd.compressBound(size)
    comp = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); coff = gpu.batch.uniform_offsets(n, bound, "cuda")
    csz = gpu.batch.compress(src, soff, comp, coff, 3)
    packed, poff = gpu.batch.pack(csz, comp, coff)
    back = torch.empty_like(src)
    dsz = gpu.batch.decompress(packed, poff, back, soff)
    torch.cuda.synchronize()
    assert bool((csz > 0).all()) and bool((dsz == size).all()) and torch.equal(back, src)
    for i in (0, 65535, 65536, 70000):
        f = packed[int(poff[i]):int(poff[i + 1])].cpu().numpy().tobytes()
        assert f == oracle_ref.compress(src[i * size:(i + 1) * size].cpu().numpy().tobytes(), 3), i


def test_gpu_concurrent_callers(gpu, oracle_ref):
    """"One context per thread, many threads" (J/ZstdCompressCtx.java:32-34): host threads on their own streams share the
    device; results must not depend on the interleaving"""
    import threading
    import torch
    size = 16384
    want, errors = {}, []

    def worker(t):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for rep in range(3):
                    n = 300 + 50 * t
                    src = gpu.batch.synth(n, size, 1000 * t)
                    soff = gpu.batch.uniform_offsets(n, size, "cuda")
                    bound = gpu.Zstd.compressBound(size)
                    comp = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); coff = gpu.batch.uniform_offsets(n, bound, "cuda")
                    csz = gpu.batch.compress(src, soff, comp, coff, 1 + (t % 3))
                    packed, poff = gpu.batch.pack(csz, comp, coff)
                    back = torch.empty_like(src)
                    dsz = gpu.batch.decompress(packed, poff, back, soff)
                    st.synchronize()
                    assert bool((dsz == size).all()) and torch.equal(back, src)
                    digest = (int(csz.sum()), int(packed[: int(poff[-1])].to(torch.int64).sum()))
                    assert want.setdefault(t, digest) == digest
                    data = gpu.synth_host(5000, t, 1)
                    assert gpu.Zstd.decompress(gpu.Zstd.compress(data, 2), len(data)) == data
        except Exception as ex:          # noqa: BLE001
            errors.append((t, repr(ex)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in threads: th.start()
    for th in threads: th.join()
    assert not errors, errors


@pytest.mark.parametrize("mode", ["small-batch", "wave-only", "shared"])
def test_gpu_wave_matcher(gpu, oracle_ref, monkeypatch, mode):
    """level 3, frames <= 64 KiB: the wave-per-frame matcher (tables in LDS, zj_match_wave.h) gives the reference's bytes — as
    the small-batch path (the default below 4 096 buffers), as the only matcher of the large-batch pipeline, and sharing a
    batch with the lane-per-frame matcher through the partitioned queue (ZJNI_HYBRID, an experiment that stays off)"""
    if mode != "small-batch":
        needs_tuning_build(gpu)
        monkeypatch.setenv("ZJNI_SPLIT_MIN", "1")
        monkeypatch.setenv("ZJNI_HYBRID", "1")
    if mode == "wave-only":
        monkeypatch.setenv("ZJNI_WAVE_ONLY", "1")
    rnd = random.Random(41)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    noise = bytes(rnd.getrandbits(8) for _ in range(70000))
    datas = [d for _, d in edge_inputs() if len(d) <= 65536]
    datas += [noise[:3000] + noise[100:2500] + noise[3000:20000] + noise[5000:9000] + noise[20000:60000],
              noise[:30000] + b"\x00" * 2000 + noise[:30000],
              bytes(rnd.choice(b"ab") for _ in range(65000)),
              b"".join(bytes([i & 255]) * rnd.randrange(1, 40) for i in range(4000))[:65536],
              (noise[:37] * 2000)[:65536], (noise[:64] * 1100)[:65536], (noise[:700] * 100)[:65536]]
    for _ in range(40):
        size = rnd.choice([rnd.randrange(1, 300), rnd.randrange(64, 5000), rnd.randrange(64, 65537), 65536])
        off = rnd.randrange(0, len(xml) - size)
        datas.append(xml[off:off + size])
    for _ in range(400):
        size = rnd.choice([rnd.randrange(0, 300), rnd.randrange(64, 5000), rnd.randrange(64, 65537), 65536, 4096])
        datas.append(gpu.synth_host(size, rnd.randrange(0, 100000), 1) if size else b"")
    for checksum in (False, True):
        outs = gpu.compress_batch(datas, 3, checksum=checksum) if checksum else gpu.compress_batch(datas, 3)
        for k, (d, z) in enumerate(zip(datas, outs)):
            assert not isinstance(z, Exception), (k, len(d), z)
            assert z == oracle_ref.compress(d, 3, checksum), (k, len(d), checksum)


def test_gpu_tables_cleared_ahead_between_calls(gpu, oracle_ref, monkeypatch):
    """the lane pipeline's level-3 tables are zeroed for the NEXT call as soon as a call's match kernel is done (clear stream, zj_kernels.hip):
    calls of equal, smaller and larger batches, other levels and a decompress call in between all find what they need — the reference's
    bytes every time (the unordered cases and ZJNI_PRECLEAR=0: the next test)"""
    monkeypatch.setenv("ZJNI_SPLIT_MIN", "1")
    rnd = random.Random(99)
    def batch(k, seed):
        return [gpu.synth_host(rnd.choice([65536, 65536, 40000, 9000, 300]), seed + i, 1) for i in range(k)]
    want = {}
    for rep, (k, level) in enumerate([(96, 3), (96, 3), (40, 3), (160, 3), (160, 1), (160, 3), (96, 3), (200, 3), (200, 3)]):
        datas = batch(k, 1000 * rep)
        outs = gpu.compress_batch(datas, level)
        for d, z in zip(datas, outs):
            assert z == oracle_ref.compress(d, level), (rep, k, level, len(d))
        if rep % 3 == 1:
            back = gpu.decompress_batch(outs, [len(d) for d in datas])
            assert back == datas


@pytest.mark.parametrize("preclear", ["1", "0"])
def test_gpu_pending_table_clear_is_ordered_before_every_scratch_user(gpu, oracle_ref, monkeypatch, preclear):
    """the clear a level-3 lane call queues for the NEXT call runs on a stream of its own, and every later user of the same
    scratch — levels 4-8, the LDS matcher of small explicit 14 / 13 batches, a level 1-3 call with a larger table range — must order itself
    behind it, not only the call that skips its own memset.  Device-resident calls enqueued back to back (nothing synchronises in between) after
    a batch whose tables take milliseconds to clear (12 288 x 64 KiB: 4.5 GiB); the followers' frames must be the reference's.  Both
    settings of ZJNI_PRECLEAR."""
    import torch
    if preclear == "0": needs_tuning_build(gpu)
    monkeypatch.setenv("ZJNI_PRECLEAR", preclear)
    monkeypatch.setenv("ZJNI_L3_WAVE_MAX", "0")
    monkeypatch.setenv("ZJNI_SPLIT_MIN", "1")
    dev = "cuda"
    nBig, size = 12288, 65536
    big = gpu.batch.synth(nBig, size)
    bigOff = gpu.batch.uniform_offsets(nBig, size, dev)
    bound = int(gpu.lib().zjni_compressBound(size))
    bigDst = torch.empty(nBig * bound, dtype=torch.uint8, device=dev); bigDstOff = gpu.batch.uniform_offsets(nBig, bound, dev)
    followers = [dict(level=5, n=48, size=12000), dict(level=3, n=64, size=65536, hash_log=14, chain_log=13), dict(level=3, n=160, size=65536, hash_log=17, chain_log=16),
                 dict(level=1, n=96, size=40000), dict(level=4, n=40, size=65536)]
    for rep in range(3):
        for f in followers:
            n, fs = f["n"], f["size"]
            src = gpu.batch.synth(n, fs, first_index=1000 * rep + 7)
            srcOff = gpu.batch.uniform_offsets(n, fs, dev)
            fb = int(gpu.lib().zjni_compressBound(fs))
            dst = torch.zeros(n * fb, dtype=torch.uint8, device=dev); dstOff = gpu.batch.uniform_offsets(n, fb, dev)
            r0 = gpu.batch.compress(big, bigOff, bigDst, bigDstOff, 3)                                     # queues the clear behind its match kernel
            res = gpu.batch.compress(src, srcOff, dst, dstOff, f["level"], hash_log=f.get("hash_log", 0), chain_log=f.get("chain_log", 0))   # ... and this call starts under it
            torch.cuda.synchronize()
            assert int((r0 < 0).sum()) == 0
            host = src.cpu().numpy().tobytes(); out = dst.cpu().numpy().tobytes(); sizes = res.cpu().tolist()
            for i in range(n):
                d = host[i * fs:(i + 1) * fs]
                assert sizes[i] > 0, (rep, f, i, sizes[i])
                want = oracle_ref.compress(d, f["level"], False, f.get("hash_log", 0), f.get("chain_log", 0))
                assert out[i * fb:i * fb + sizes[i]] == want, (rep, f, i)


def test_gpu_small_level3_batches_take_the_wave_route(gpu, oracle_ref, monkeypatch):
    """level 3, batches below ZJNI_L3_WAVE_MAX (8 192): every frame its own wave on zj_encode_multi_kernel (ZJNI_ROUTE_WAVE_HBM = 9, zj_match_wavex.h over HBM
    tables) — the reference's plain level-3 bytes for every size up to a block, with checksum, with explicit table sizes, mixed with multi-block frames;
    ZJNI_L3_WAVE_MAX=0 sends the same batch down the lane pipeline (route 5 / 6)"""
    rnd = random.Random(7)
    datas = [d for _, d in edge_inputs() if len(d) <= 131072]
    datas += [gpu.synth_host(rnd.choice([65536, 65536, 131072, 100000, 40000, 9000, 300, 64, 63, 7, 6, 0]), 900 + k, 1) for k in range(300)]
    datas += [golden("xmlsmall")[:60000], gpu.synth_host(300000, 5, 1)]
    for ck in (False, True):
        outs = gpu.compress_batch(datas, 3, checksum=ck)
        assert gpu.lib().zjni_last_route() == 9
        for d, z in zip(datas, outs):
            assert z == oracle_ref.compress(d, 3, ck), (len(d), ck)
    for hl, cl in ((17, 16), (14, 13), (12, 15)):
        outs = gpu.compress_batch(datas[:120], 3, hash_log=hl, chain_log=cl)
        for d, z in zip(datas[:120], outs):
            assert z == oracle_ref.compress(d, 3, False, hl, cl), (len(d), hl, cl)
    back = gpu.decompress_batch(gpu.compress_batch(datas, 3), [len(d) for d in datas])
    assert back == datas
    monkeypatch.setenv("ZJNI_L3_WAVE_MAX", "0")
    outs = gpu.compress_batch(datas[:200], 3)
    assert gpu.lib().zjni_last_route() in (5, 6)
    for d, z in zip(datas[:200], outs):
        assert z == oracle_ref.compress(d, 3), len(d)


@pytest.mark.parametrize("need", ["2", "0"])
def test_gpu_lane_pipeline_hand_overs_at_small_and_odd_counts(gpu, oracle_ref, monkeypatch, need):
    """the match waves hand their finished frames to the entropy kernel in batches (one release per wave and hand-over, a wave-wide ballot to leave) — code only
    the GPU runs.  Batches smaller than a wave, one frame over a wave, not a multiple of anything, with and without need flags, three calls each (the queue and the
    flags of the previous call are there): every frame comes out, once, as the reference's — a frame that missed the queue is the sweep pass's, one that was queued twice
    or early would differ or fail to decode."""
    monkeypatch.setenv("ZJNI_SPLIT_MIN", "1"); monkeypatch.setenv("ZJNI_L3_WAVE_MAX", "0"); monkeypatch.setenv("ZJNI_NEED", need)
    rnd = random.Random(211)
    for count in (1, 2, 63, 64, 65, 127, 129, 191, 257, 1023):
        datas = [gpu.synth_host(rnd.choice([65536, 65536, 30000, 4096, 700, 64]), rnd.randrange(1 << 20), 1) for _ in range(count)]
        want = [oracle_ref.compress(d, 3) for d in datas]
        for rep in range(3):
            outs = gpu.compress_batch(datas, 3)
            assert gpu.lib().zjni_last_route() == (6 if need == "2" else 5)
            for k, (z, w) in enumerate(zip(outs, want)):
                assert z == w, (count, rep, k, len(datas[k]), z if isinstance(z, Exception) else "bytes differ")
