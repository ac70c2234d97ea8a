"""GPU (-m gpu): multi-block frames (128 KiB < input <= 2 MiB) through the C-ABI are byte-identical to the reference's ZSTD_compress2
at levels 1-3 (the level's own parameters), decode back on the GPU, and mix freely with single-block buffers in one batch."""
import random

import pytest

from conftest import golden

from util import needs_tuning_build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(zj):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    zj.batch.init(0)
    return zj


WINDOW = {1: 1 << 19, 2: 1 << 20, 3: 1 << 21}


def inputs(gpu, ref, seed, count):
    rnd = random.Random(seed)
    xml = ref.decompress(golden("xml-1.zst"), 6_000_000)
    noise = bytes(rnd.getrandbits(8) for _ in range(200000))
    out = []
    for _ in range(count):
        size = rnd.choice([131073, rnd.randrange(131073, 300000), rnd.randrange(131073, 1100000), 262144, 524288, 1048576, rnd.randrange(1100000, 2097153), 65536, 4096, 100000])
        parts = []
        while sum(map(len, parts)) < size:
            kind = rnd.randrange(6); n = rnd.choice([1000, 8192, 40000, 131072, 200000])
            if kind == 0: parts.append(noise[:min(n, 60000)])
            elif kind == 1: o = rnd.randrange(0, len(xml) - n); parts.append(xml[o:o + n])
            elif kind == 2: parts.append(gpu.synth_host(min(n, 65536), rnd.randrange(1 << 20), 1))
            elif kind == 3: parts.append(bytes([rnd.getrandbits(8)]) * n)
            elif kind == 4: parts.append(b"".join(bytes([rnd.getrandbits(8)]) * rnd.randrange(200, 9000) + noise[:rnd.randrange(0, 12)] for _ in range(8)))
            else: parts.append(parts[rnd.randrange(len(parts))] if parts else b"seed" * 100)
        out.append(b"".join(parts)[:size])
    return out


@pytest.mark.parametrize("level", [1, 2, 3])
def test_gpu_multiblock_frames_are_the_references(gpu, oracle_ref, level):
    datas = inputs(gpu, oracle_ref, 100 + level, 60)
    for checksum in (False, True):
        outs = gpu.compress_batch(datas, level, checksum=checksum)
        good = []
        for d, z in zip(datas, outs):
            if len(d) > WINDOW[level]:
                assert isinstance(z, Exception) and z.getErrorCode() == 201, len(d)
                continue
            want = oracle_ref.compress(d, level, checksum)
            assert not isinstance(z, Exception), (len(d), z)
            assert z == want, (len(d), level, checksum)
            good.append((d, z))
        back = gpu.decompress_batch([z for _, z in good], [len(d) for d, _ in good])
        for (d, _), b in zip(good, back):
            assert b == d


@pytest.mark.parametrize("switch", [("ZJNI_MULTI_WAVE", "0"), ("ZJNI_MULTI_WAVE", "2"), ("ZJNI_MULTI_WAVE_FAST", "1")])
def test_gpu_multiblock_parse_switches(gpu, oracle_ref, monkeypatch, switch):
    """level-3 blocks of multi-block frames run the wave matcher (zj_match_wavex.h) by default, levels 1-2 the one-lane parse;
    ZJNI_MULTI_WAVE=0 selects the one-lane parses of rounds 1-3, =2 the wave matchers without staged spans, ZJNI_MULTI_WAVE_FAST=1 puts levels 1-2
    on their wave matcher too (exact, measured slower than the one-lane parse there: off by default) — the same frames either way"""
    needs_tuning_build(gpu)
    monkeypatch.setenv(*switch)
    for level in (3, 1, 2):
        datas = [d for d in inputs(gpu, oracle_ref, 100 + level, 24) if len(d) <= WINDOW[level]]
        for d, z in zip(datas, gpu.compress_batch(datas, level)):
            assert z == oracle_ref.compress(d, level), (level, len(d))


def test_gpu_one_mebibyte_buffer_like_baseline_config_1(gpu, oracle_ref):
    """BASELINE config 1's shape: Zstd.compress / decompress of a 1 MiB buffer at level 3 — on the GPU now, same bytes as the CPU path"""
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)[: 1 << 20]
    z = gpu.Zstd.compress(xml, 3)
    assert z == oracle_ref.compress(xml, 3)
    assert gpu.Zstd.decompress(z, len(xml)) == xml
    with pytest.raises(gpu.ZstdException) as ex:         # level 1: the frame (1 MiB) exceeds the window (512 KiB): left to the CPU path
        gpu.Zstd.compress(xml, 1)
    assert ex.value.getErrorCode() == 201


def test_gpu_multiblock_repeated_calls_leave_no_state_behind(gpu, oracle_ref):
    """The kernels are persistent: a workgroup's shared state outlives a frame and a call.  Repeating one batch several times — so that
    a workgroup meets a frame right after another frame's (or its own) leftovers — must give the reference's bytes every time
    (a stale "this block made a Huffman table" flag once let a later block go treeless against a table that did not exist;
    tools/stress_gpu_multiblock.py is the long version)."""
    for level in (1, 3):
        datas = inputs(gpu, oracle_ref, 100 + level, 60)
        want = [None if len(d) > WINDOW[level] else oracle_ref.compress(d, level) for d in datas]
        for rep in range(6):
            outs = gpu.compress_batch(datas, level)
            for i, (z, w) in enumerate(zip(outs, want)):
                if w is not None:
                    assert z == w, (level, rep, i, len(datas[i]))


@pytest.mark.parametrize("route", ["pipelined", "one-wave"])
def test_gpu_multiblock_pipelined_pair_of_waves(gpu, oracle_ref, monkeypatch, route):
    """zj_encode_pipe_kernel (round 6): a batch of multi-block frames that leaves wave slots empty is parsed a block AHEAD of its entropy stage, two waves per frame — the
    parse wave runs on an upper bound of the previous block's compressed size and waits for the real answer where the bound decides nothing (noise, one repeated byte,
    tight destinations); the entropy wave checks every assumption (a wrong one would come back as error 1).  Frames and refusals are the reference's at every capacity;
    the library names the kernel (ZJNI_ROUTE_PIPE after zjni_last_lists).  one-wave: the same batch on zj_encode_multi_kernel (ZJNI_PIPE_MAX=0, tuning builds)."""
    import ctypes as C
    if route == "one-wave":
        needs_tuning_build(gpu)
        monkeypatch.setenv("ZJNI_PIPE_MAX", "0")
    rnd = random.Random(77)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    noise = bytes(rnd.getrandbits(8) for _ in range(300000))
    datas = [xml[1000:1000 + 300000], noise[:140000] + xml[:150000], xml[:131072] + bytes([7]) * 140000 + xml[5000:9000], bytes([9]) * 131072 + noise[:20000] + bytes([9]) * 131072,
             gpu.synth_host(65536, 5, 1) * 4 + noise[:3], (noise[:5000] * 60)[:270000], xml[:131072] + noise[:131072] + xml[:6], xml[:1 << 20], xml[777:777 + (2 << 20)]]
    datas += [d for d in inputs(gpu, oracle_ref, 7, 30) if len(d) > 131072]
    for level in (3, 1, 2):
        ds = [d for d in datas if len(d) <= WINDOW[level]]
        want = [oracle_ref.compress(d, level) for d in ds]
        outs = gpu.compress_batch(ds, level)
        for k, (z, w) in enumerate(zip(outs, want)):
            assert z == w, (level, k, len(ds[k]), z if isinstance(z, Exception) else "bytes differ")
        assert gpu.decompress_batch(outs, [len(d) for d in ds]) == ds
        # destinations around the frame's size: the reference's bytes or its refusal, frame by frame
        caps, exp = [], []
        for d, w in zip(ds, want):
            fs = len(w); cap = rnd.choice([fs - 1, fs, fs + 1, fs + 7, fs + 8, fs + 9, fs + 64, fs + 1030, fs // 2, 17, 18, len(d), len(d) + 20])
            caps.append(cap)
            try:
                exp.append(oracle_ref.compress(d, level, cap=cap))
            except oracle_ref.ZstdRefError as ex:
                exp.append(-ex.code)
        outs = gpu.compress_batch(ds, level, capacities=caps)
        for k, (z, w) in enumerate(zip(outs, exp)):
            assert (-z.getErrorCode() if isinstance(z, Exception) else z) == w, (level, k, caps[k], len(want[k]))
    # which kernel it was: a device-resident batch of equal 300 000-byte frames at level 3, asked of the library
    import torch
    B = gpu.batch
    n, size = 24, 300000
    src = torch.frombuffer(bytearray(b"".join(xml[7 * i:7 * i + size] for i in range(n))), dtype=torch.uint8).cuda()
    bound = gpu.Zstd.compressBound(size)
    comp = torch.empty(n * bound, dtype=torch.uint8, device="cuda")
    csz = B.compress(src, B.uniform_offsets(n, size, "cuda"), comp, B.uniform_offsets(n, bound, "cuda"), 3); torch.cuda.synchronize()
    l3 = (C.c_uint * 3)()
    assert gpu.lib().zjni_last_lists(l3) == 0 and l3[2] == n, list(l3)
    assert gpu.lib().zjni_last_route() == (11 if route == "pipelined" else 9), gpu.lib().zjni_last_route()
    hc = comp.cpu().numpy().tobytes()
    for i in range(n):
        assert hc[i * bound:i * bound + int(csz[i])] == oracle_ref.compress(xml[7 * i:7 * i + size], 3), i
