"""GPU (-m gpu): the two decoder classes closed in round 6, through the C-ABI on both batch regimes (three-stage / block-stage pipelines, fused kernel):
compressed blocks of exactly 128 KiB (N/decompress/zstd_decompress_block.c:2073-2081) and literals coded with Huffman tables 12 bits deep (HUF_TABLELOG_MAX,
N/common/huf.h:37; N/decompress/huf_decompress.c:385-518) — valid frames bit-exact, damaged ones answered as the reference's portable build answers them.
The CPU twins (lane-serial bodies) are tests/test_emu_decode.py::test_compressed_block_of_exactly_128KiB / ::test_huffman_tables_12_bits_deep and
tests/test_emu_cdict.py::test_dictionary_with_a_huffman_table_12_bits_deep."""
import random

import pytest

import dictutil as du
from util import deep_huffman_literals, literals_only_block, frame_of_blocks, hand_huffman_section

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(zj):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    zj.batch.init(0)
    return zj


def answer(o):
    return -abs(o.getErrorCode()) if isinstance(o, Exception) else o


def portable(ref, z, cap, d=None):
    try:
        return ref.decompress_portable(z, cap, d)
    except ref.ZstdRefError as ex:
        return -ex.code


def tree_len(sec):
    return sec[0] + 1 if sec[0] < 128 else 1 + (sec[0] - 127 + 1) // 2


@pytest.mark.parametrize("split_min", ["1", "1000000000"])
def test_gpu_compressed_block_of_exactly_128KiB(gpu, oracle_ref, monkeypatch, split_min):
    monkeypatch.setenv("ZJNI_DSPLIT_MIN", split_min)
    rnd = random.Random(5)
    frames, caps = [], []
    body = bytearray(rnd.getrandbits(8) for _ in range(131072))
    for lit_type in range(4):                                               # a raw 128 KiB block whose type bit was flipped
        body[0] = (body[0] & 0xFC) | lit_type
        if lit_type < 2: body[0] |= 0x0C
        frames.append(b"\x28\xb5\x2f\xfd" + b"\x00\x58" + (131072 << 3 | 2 << 1).to_bytes(3, "little") + bytes(body) + b"\x01\x00\x00"); caps.append(1 << 18)
    n_lit = 131072 - 3 - 1
    lits = bytes(rnd.getrandbits(8) for _ in range(n_lit))
    for last, extra in ((1, b""), (0, b"\x01\x00\x00")):                    # the VALID frame of this kind: raw literals filling the block, no sequences
        for sized in (False, True):
            head = b"\x28\xb5\x2f\xfd" + (b"\x80\x58" + n_lit.to_bytes(4, "little") if sized else b"\x00\x58")
            valid = head + (131072 << 3 | 2 << 1 | last).to_bytes(3, "little") + bytes([0 | 3 << 2 | (n_lit & 0xF) << 4, (n_lit >> 4) & 0xFF, n_lit >> 12]) + lits + b"\x00" + extra
            assert oracle_ref.decompress(valid, 1 << 18) == lits
            frames += [valid, valid]; caps += [1 << 18, n_lit - 1]
            for _ in range(6):
                zb = bytearray(valid); zb[rnd.randrange(5, len(zb))] ^= 1 << rnd.randrange(8); frames.append(bytes(zb)); caps.append(1 << 18)
    outs = gpu.decompress_batch(frames * 3, caps * 3)                       # (three of each: several waves at work at once)
    for k, o in enumerate(outs):
        assert answer(o) == portable(oracle_ref, frames[k % len(frames)], caps[k % len(frames)]), k
    assert outs[4] == lits


@pytest.mark.parametrize("split_min", ["1", "1000000000"])
def test_gpu_huffman_tables_12_bits_deep(gpu, oracle_ref, monkeypatch, split_min):
    monkeypatch.setenv("ZJNI_DSPLIT_MIN", split_min)
    rnd = random.Random(12)
    frames, caps, wants = [], [], []
    deep = 0
    for seed in range(10):
        n = rnd.choice([20000, 70000, 131000])
        lits, sec, depth = deep_huffman_literals(oracle_ref, n, seed)
        deep += depth == 12
        blk = literals_only_block(n, sec[4], 4)
        two = frame_of_blocks([literals_only_block(n, sec[4], 4, last=False), literals_only_block(n, sec[4][tree_len(sec[4]):], 4, treeless=True)])
        for f, w in ((frame_of_blocks([blk]), lits), (frame_of_blocks([blk], content_size=n), lits), (two, lits * 2)):
            frames.append(f); caps.append(len(w)); wants.append(w)
            for _ in range(8):
                zb = bytearray(f)
                for _ in range(rnd.randrange(1, 3)):
                    zb[rnd.randrange(6, min(len(zb), 400)) if rnd.random() < 0.6 else rnd.randrange(6, len(zb))] ^= 1 << rnd.randrange(8)
                frames.append(bytes(zb)); caps.append(len(w)); wants.append(None)
    assert deep >= 7
    for w1 in (2, 4, 6, 30, 64, 116):
        for n, streams in ((rnd.randrange(200, 1000), 1), (rnd.randrange(3000, 9000), 4)):
            lits, section = hand_huffman_section(rnd, n, w1, streams)
            f = frame_of_blocks([literals_only_block(n, section, streams)], content_size=n)
            frames.append(f); caps.append(n); wants.append(lits)
            for _ in range(10):
                zb = bytearray(f); zb[rnd.randrange(14, len(zb))] ^= 1 << rnd.randrange(8); frames.append(bytes(zb)); caps.append(n); wants.append(None)
    outs = gpu.decompress_batch(frames, caps)
    for k, (o, w) in enumerate(zip(outs, wants)):
        if w is not None:
            assert o == w, (k, answer(o) if isinstance(o, Exception) else "bytes differ")
        assert answer(o) == portable(oracle_ref, frames[k], caps[k]), k


@pytest.mark.parametrize("split_min", ["1", "1000000000"])
def test_gpu_dictionary_with_a_huffman_table_12_bits_deep(gpu, oracle_ref, monkeypatch, split_min):
    """ZstdDictCompress / ZstdDictDecompress over a dictionary whose literals table is 12 bits deep (refused at load in rounds 1-5): the reference's frames, treeless
    literals coded with the dictionary's 12-bit codes among them, both ways"""
    monkeypatch.setenv("ZJNI_DSPLIT_MIN", split_min)
    r = random.Random(12)
    alphabet = b"etaoinshrdlu."
    hist = [0] * 256
    for k, ch in enumerate(reversed(alphabet)):
        hist[ch] = 1 if k == 0 else 1 << (k - 1)
    pool = bytes(ch for ch in alphabet for _ in range(hist[ch]))
    draw = lambda n: bytes(r.choice(pool) for _ in range(n))
    content = draw(3000) + alphabet * 3
    d = du.build(content, 4242, hist, du.normalise([1] * 20, 7), 7, du.normalise([3 if i < 20 else 1 for i in range(53)], 8), 8, du.normalise([4 if i < 10 else 1 for i in range(36)], 8), 8,
                 huf_max_bits=12)
    datas = []
    for n in (40, 90, 200, 500, 1200, 3000, 7000) * 6:
        datas += [draw(n), draw(n // 2) + content[100:100 + n // 2]]
    for level in (1, 3):
        rc = oracle_ref.CDict(d, level)
        want = [rc.compress(x) for x in datas]
        rc.close()
        treeless = sum((w[4 + 1 + 2 + (1 if len(x) < 256 else 2)] >> 1) & 3 == 2 and w[4 + 1 + 2 + (1 if len(x) < 256 else 2) + 3] & 3 == 3 for w, x in zip(want, datas))
        assert treeless >= 20, treeless
        with gpu.ZstdDictCompress(d, level) as cd:
            got = gpu.compress_batch(datas, level, dictionary=cd)
        for k, (g, w) in enumerate(zip(got, want)):
            assert g == w, (level, k, answer(g) if isinstance(g, Exception) else "bytes differ")
        with gpu.ZstdDictDecompress(d) as dd:
            frames, caps = list(want), [len(x) for x in datas]
            for w, x in zip(want, datas):
                zb = bytearray(w); zb[r.randrange(8, len(zb))] ^= 1 << r.randrange(8); frames.append(bytes(zb)); caps.append(len(x))
            outs = gpu.decompress_batch(frames, caps, dd)
            for k, o in enumerate(outs):
                assert answer(o) == portable(oracle_ref, frames[k], caps[k], d), (level, k)
            assert outs[:len(datas)] == datas
