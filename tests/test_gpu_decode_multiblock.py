"""GPU (-m gpu): multi-block frames and frames without a content size on the split decode pipeline's block stages (zj_decode_split.h: stage 1 per frame, a LANE
per block with the repcode history carried symbolically, stage 3 per frame) — bit-exact with the reference in small batches (two blocks and more) and inside large
ones next to single-block frames; ZJNI_DEC_MB=0 (the fused kernel for all of them) gives the same bytes; damaged frames are answered as before."""
import ctypes as C
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


def lists(gpu):
    a = (C.c_uint * 5)()
    assert gpu.lib().zjni_last_decode_lists2(a) == 0
    return list(a)                 # [0] single-block pipeline, [1] fused kernel, [2] block stages, [3] their blocks, [4] frames of one stored block copied by stage 1


def test_gpu_golden_and_stream_frames_take_the_block_stages(gpu, oracle_ref):
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    frames = [golden(n) for n in ("xml-1.zst", "xml-3.zst", "xml-6.zst", "xml-9.zst", "xml-advanced.zst", "xml-1-sized.zst")]
    outs = gpu.decompress_batch(frames, [len(xml)] * len(frames))
    assert all(o == xml for o in outs)
    l = lists(gpu)
    assert l[2] == len(frames) and l[1] == 0 and l[3] > 6 * 40, l          # every frame through the block stages, none handed over


@pytest.mark.parametrize("mb", ["1", "0", "behind"])
def test_gpu_multiblock_frames_small_and_large_batches(gpu, oracle_ref, monkeypatch, mb):
    monkeypatch.setenv("ZJNI_DEC_MB", "0" if mb == "0" else "1")
    if mb == "behind": needs_tuning_build(gpu)
    if mb == "behind": monkeypatch.setenv("ZJNI_DEC_MB_OVERLAP", "0")      # stage 3 behind stage 2 (the default runs it beside, block by block as stage 2 sets seqReady)
    rnd = random.Random(29)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    noise = bytes(rnd.getrandbits(8) for _ in range(50000))
    datas = []
    for size in (131073, 200000, 262144, 400000, 1048576):
        o = rnd.randrange(0, len(xml) - size)
        datas += [xml[o:o + size], b"".join(gpu.synth_host(65536, 7 * i + size, 1) for i in range(size // 65536 + 1))[:size], (noise * 30)[:size],
                  (b"\0" * 150000 + xml[o:o + 70000] + bytes([7]) * 300000 + noise)[:size]]
    frames, want = [], []
    for k, d in enumerate(datas):
        level = (1, 3, 5, 9)[k & 3]
        frames.append(oracle_ref.compress(d, level, bool(k & 4))); want.append(d)
        frames.append(oracle_ref.compress_stream(d, (3, 1)[k & 1], bool(k & 2), chunk=(50000, 7000, 131072)[k % 3], flush_every=(0, 1, 3)[k % 3])); want.append(d)
    outs = gpu.decompress_batch(frames, [len(d) + 64 for d in want])
    for k, (o, d) in enumerate(zip(outs, want)):
        assert o == d, (k, len(d))
    l = lists(gpu)
    if mb != "0": assert l[2] >= len(frames) - 2 and l[1] <= 2, l
    else: assert l[2] == 0, l
    # the same frames inside a large batch of single-block frames (the three-stage pipeline beside the block stages)
    monkeypatch.setenv("ZJNI_DSPLIT_MIN", "1")
    small = [gpu.synth_host(rnd.choice([65536, 30000, 4096]), 500 + i, 1) for i in range(300)]
    frames2 = frames + [oracle_ref.compress(d, 3) for d in small]; want2 = want + small
    order = list(range(len(frames2))); rnd.shuffle(order)
    outs = gpu.decompress_batch([frames2[i] for i in order], [len(want2[i]) + (64 if i < len(want) else 0) for i in order])
    for j, i in enumerate(order):
        assert outs[j] == want2[i], (i, len(want2[i]))
    l = lists(gpu)
    if mb != "0": assert l[0] >= 200 and l[4] > 0 and l[0] + l[2] + l[4] == len(frames2) and l[1] == 0, l      # single-block frames that are not "simple": one stored block (the random class) is copied by stage 1, the rest take the block stages too
    else: assert l[0] >= 200 and l[2] == 0 and l[0] + l[1] + l[4] == len(frames2), l


def test_gpu_damaged_multiblock_frames(gpu, oracle_ref):
    rnd = random.Random(31)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    d = xml[100000:100000 + 500000]
    z = oracle_ref.compress(d, 3, True)
    frames = []
    for _ in range(200):
        b = bytearray(z); i = rnd.randrange(4, len(b)); b[i] ^= 1 << rnd.randrange(8); frames.append(bytes(b))
    outs = gpu.decompress_batch(frames, [len(d)] * len(frames))
    for f, o in zip(frames, outs):
        try:
            want = oracle_ref.decompress_portable(f, len(d))
        except oracle_ref.ZstdRefError as e:
            assert isinstance(o, Exception) and o.getErrorCode() == e.code, (o, e.code)
            continue
        assert o == want


def test_gpu_concrete_offsets_that_look_symbolic_are_refused(gpu, oracle_ref):
    """ADVICE r04 (see the emulation twin): crafted frames without a checksum whose one sequence carries offset codes up to 31."""
    from util import crafted_far_offset_frame
    cases = [(2, 1), (5, 0), (6, 3), (26, 5), (27, 0), (27, 12345), (28, 0), (29, 7), (30, 1 << 29), (31, 0), (31, 3), (31, 5), (31, (1 << 31) - 1), (31, 1 << 30)]
    frames, totals = zip(*[crafted_far_offset_frame(c, x) for c, x in cases])
    outs = gpu.decompress_batch(list(frames), list(totals))
    for (c, x), f, t, o in zip(cases, frames, totals, outs):
        try:
            want = oracle_ref.decompress_portable(f, t)
        except oracle_ref.ZstdRefError as e:
            assert isinstance(o, Exception) and o.getErrorCode() == e.code, (c, x, o, e.code)
            continue
        assert o == want, (c, x)
