"""CPU (-m "not gpu"): multi-block frames (128 KiB < input <= 2 MiB; ze_compress_multi in zstd-jni_amd/csrc/zj_encode.h + the pre-split
heuristic of zj_presplit.h), built lane-serial (tests/emu), are byte-identical to the reference's ZSTD_compress2 at levels 1-3 with
the level's own parameters: block sizes (ZSTD_optimalBlockSize / ZSTD_splitBlock), repcodes and the Huffman table carried from
block to block, raw / RLE / compressed blocks, checksum, frames without content size (N/compress/zstd_compress.c:4552-4692)."""
import random

import pytest

from conftest import golden
import ctypes as C
import os

from util import ROOT, emu_lib, emu_compress_multi, emu_decompress


@pytest.fixture(scope="module")
def emu():
    return emu_lib()


WINDOW = {1: 1 << 19, 2: 1 << 20, 3: 1 << 21}           # the frame has to fit the level's window (clevels.h:27-30)


def check(emu, ref, d, level, checksum=False, content_size=True, tag=None):
    got = emu_compress_multi(emu, d, level, checksum, content_size)
    if len(d) > WINDOW[level]:
        assert got == -201, (tag, len(d), level)
        return None
    want = ref.compress(d, level, checksum, content_size=content_size)
    assert got == want, (tag, len(d), level, checksum, content_size, len(want), got if isinstance(got, int) else len(got))
    piped = emu_compress_multi(emu, d, level, checksum, content_size, pipelined=True)          # zj_encode_pipe_kernel's two roles (round 6): a parse wave ahead of an entropy wave
    assert piped == want, ("pipelined", tag, len(d), level, checksum, content_size, len(want), piped if isinstance(piped, int) else len(piped))
    return got


def test_multiblock_synthetic_classes(emu, oracle_ref, zj):
    for size in (131073, 131080, 200000, 262144, 262145, 300000, 524288, 524289, 1048576):
        for cls in range(5):
            if cls < 4:
                parts, i = [], cls
                while sum(map(len, parts)) < size:
                    parts.append(zj.synth_host(65536, i, 1)); i += 4
                d = b"".join(parts)[:size]
            else:
                d = zj.synth_host(65536, 7, (size + 65535) // 65536)[:size]      # classes change every 64 KiB: the pre-split heuristic cuts
            for level in (1, 2, 3):
                check(emu, oracle_ref, d, level, tag=("synth", cls))


def test_multiblock_xml_and_flags(emu, oracle_ref):
    rnd = random.Random(9)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    for size in (140000, 262144, 1 << 20, 1500000, 2 << 20):
        off = rnd.randrange(0, len(xml) - size)
        d = xml[off:off + size]
        for level in (1, 2, 3):
            for ck, cs in ((False, True), (True, True), (False, False), (True, False)):
                z = check(emu, oracle_ref, d, level, ck, cs, tag=("xml", off))
                if z is not None and ck and cs:
                    assert emu_decompress(emu, z, len(d)) == d
    assert emu_compress_multi(emu, xml[:(2 << 20) + 1], 3) == -201              # beyond ZE_MULTI_MAX
    # (<= 128 KiB this entry is the level-4 route of the same kernel: tests/test_emu_level4.py)


def test_multiblock_block_types(emu, oracle_ref):
    """RLE blocks (never the first one), raw blocks between compressible ones (their repcodes and Huffman table are not confirmed),
    literal sections small enough to repeat the previous block's Huffman table, long matches across block borders"""
    rnd = random.Random(4)
    noise = bytes(rnd.getrandbits(8) for _ in range(300000))
    words = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta", b"eta", b"theta"]
    text = b" ".join(rnd.choice(words) for _ in range(120000))
    lowent = bytes(rnd.choice(b"abcdefgh") for _ in range(400000))
    cases = {
        "zeros": b"\x00" * 700000,
        "zeros_then_text": b"\x00" * 300000 + text[:200000],
        "text_zeros_text": text[:131072] + b"z" * 262144 + text[:150000],
        "text": text[:600000],
        "noise_text_noise": noise[:140000] + text[:200000] + noise[:131072] + text[200000:330000],
        "noise": noise,
        "lowent": lowent,
        "long_repeat": (noise[:70000] * 8)[:500000],
        "periodic": (b"0123456789abcdef" * 40000)[:520000],
        "text_small_tail": text[:131072 * 3 + 5],
        "tail_6": text[:131072 * 2 + 6],
        "tail_7": text[:131072 * 2 + 7],
        "rle_tail": text[:131072] + b"q" * (131072 + 40),
        "sparse_literals": (b"A" * 5000 + noise[:300] + b"B" * 7000 + noise[300:500]) * 20,
    }
    for name, d in cases.items():
        for level in (1, 2, 3):
            check(emu, oracle_ref, d, level, tag=name)
            check(emu, oracle_ref, d, level, True, tag=name)


def test_multiblock_random_shapes(emu, oracle_ref, zj):
    rnd = random.Random(77)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    for k in range(40):
        size = rnd.choice([rnd.randrange(131073, 300000), rnd.randrange(131073, 1100000), 131073, 393216, 524288])
        parts = []
        while sum(map(len, parts)) < size:
            kind = rnd.randrange(5); n = rnd.choice([1000, 8192, 40000, 131072, 200000])
            if kind == 0: parts.append(bytes(rnd.getrandbits(8) for _ in range(min(n, 50000))))
            elif kind == 1: o = rnd.randrange(0, len(xml) - n); parts.append(xml[o:o + n])
            elif kind == 2: parts.append(zj.synth_host(min(n, 65536), rnd.randrange(1 << 20), 1))
            elif kind == 3: parts.append(bytes([rnd.getrandbits(8)]) * n)
            else: parts.append(parts[rnd.randrange(len(parts))] if parts else b"seed" * 100)
        d = b"".join(parts)[:size]
        level = rnd.choice([1, 2, 3])
        check(emu, oracle_ref, d, level, rnd.random() < 0.3, rnd.random() < 0.8, tag=("shape", k))


def test_multiblock_wave_matcher_orders_and_switches(emu, oracle_ref, zj):
    """level-3 blocks run the wave matcher (zstd-jni_amd/csrc/zj_match_wavex.h) in its explicit-SIMT build: 64 emulated lanes visited in
    ascending order (libzjni_emu.so) and in descending order (libzjni_emu_rev.so — where lanes of one step store to one address the GPU
    promises no winner, so the frames must not depend on it), with and without the staged spans, and the one-lane parse behind
    ZE_FLAG_MULTI_SERIAL: every variant gives the reference's frame.  Inputs: what drives windows wider than a hit (noise, long
    literal runs), equal hashes inside a window (two-letter alphabets, byte runs), matches longer than the staged span, matches that
    reach back over block borders, steps above 1."""
    rev = C.CDLL(os.path.join(ROOT, "tests", "emu", "libzjni_emu_rev.so"))
    rev.emu_compress_multi.restype = C.c_ulonglong
    rev.emu_compress_multi.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_uint]
    rnd = random.Random(12)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    noise = bytes(rnd.getrandbits(8) for _ in range(200000))
    cases = [xml[:400000], xml[1000000:1000000 + 1048576],
             bytes(rnd.choice(b"ab") for _ in range(300000)),
             b"".join(bytes([i & 255]) * rnd.randrange(1, 40) for i in range(20000))[:330000],
             noise[:150000] + noise[:150000] + xml[:50000],                       # a 150 000-byte match across a block border
             (noise[:37] * 9000)[:300000], (noise[:700] * 500)[:300000],
             noise[:140000] + xml[:140000] + noise[:3000] + xml[100:130000],
             b"".join(zj.synth_host(65536, 2 + 4 * i, 1) for i in range(5)),      # low-entropy class: dense short matches, repcodes
             noise[:5000] + noise[4000:4990] + b"#" + noise[4000:5000] + xml[:200000] + noise[100:400] + b"!" + noise[99:5000]]
    for k, d in enumerate(cases):
        for level in (3, 1, 2):                      # level 3: double-fast (ZWaveX); levels 1-2: fast (ZWaveF; level 2 is double-fast up to 256 KiB)
            if len(d) > WINDOW[level]: continue
            want = oracle_ref.compress(d, level)
            for lib, name in ((emu, "ascending"), (rev, "descending")):
                for serial in (False, 2, True, 4):
                    if lib is rev and serial in (True, 4): continue
                    assert emu_compress_multi(lib, d, level, serial=serial) == want, (k, level, name, serial)


def test_level3_single_block_frames_on_the_wave_route(emu, oracle_ref, zj):
    """small level-3 batches (ZJNI_ROUTE_WAVE_HBM): every frame — any size up to a block — goes to the multi-block kernel, where the wave matcher parses it as
    one block over HBM tables with the level's own table sizes (16 / 15: the reference's plain level 3) or the caller's; frames below 64 bytes take
    the one-lane parse there"""
    from util import edge_inputs
    rev = C.CDLL(os.path.join(ROOT, "tests", "emu", "libzjni_emu_rev.so"))
    rev.emu_compress_multi.restype = C.c_ulonglong
    rev.emu_compress_multi.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_uint]
    rnd = random.Random(21)
    datas = [d for _, d in edge_inputs() if len(d) <= 131072]
    datas += [zj.synth_host(rnd.choice([65536, 65536, 131072, 40000, 9000, 300, 64, 63, 7, 6]), 500 + k, 1) for k in range(60)]
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    datas += [xml[o:o + n] for o, n in ((0, 131072), (777777, 65536), (2000000, 100000), (5, 16384), (9, 16385))]
    for k, d in enumerate(datas):
        for ck in (False, True):
            want = oracle_ref.compress(d, 3, ck)
            assert emu_compress_multi(emu, d, 3, ck, hash_log=16, chain_log=15) == want, (k, len(d), ck)
        assert emu_compress_multi(rev, d, 3, hash_log=16, chain_log=15) == oracle_ref.compress(d, 3), (k, len(d), "descending")
        assert emu_compress_multi(emu, d, 3, hash_log=15, chain_log=16) == oracle_ref.compress(d, 3, False, 15, 16), (k, len(d), "15/16")


def test_presplit_on_the_group_equals_the_one_lane_walk(emu, zj):
    """zj_presplit.h: ZSTD_splitBlock's chunk fingerprints (N/compress/zstd_preSplit.c:152-238) sampled and compared by the whole group give the cut the one-lane
    walk gives — on blocks whose statistics change at every chunk border, at none, and at random places"""
    emu.emu_presplit_chunks.restype = C.c_uint
    emu.emu_presplit_chunks.argtypes = [C.c_char_p, C.c_int]
    rnd = random.Random(5)
    kinds = [lambda n: bytes(rnd.getrandbits(8) for _ in range(n)), lambda n: bytes(rnd.choice(b"abcdefgh ") for _ in range(n)),
             lambda n: bytes([rnd.getrandbits(8)]) * n, lambda n: zj.synth_host(n, rnd.randrange(1 << 20), 1)[:n],
             lambda n: bytes(rnd.getrandbits(8) & 0x0F for _ in range(n))]
    cuts = set()
    for k in range(120):
        parts = []
        while sum(map(len, parts)) < 131072:
            parts.append(rnd.choice(kinds)(rnd.choice([8192, 8192, 16384, 40000, 3000, 131072, 65536])))
        d = b"".join(parts)[:131072]
        a, b = emu.emu_presplit_chunks(d, 0), emu.emu_presplit_chunks(d, 1)
        assert a == b, (k, a, b)
        cuts.add(a)
    assert len(cuts) >= 6 and 131072 in cuts, cuts          # the inputs did exercise several cut positions and "no cut"


def test_pipelined_roles_in_tight_destinations(emu, oracle_ref, zj):
    """zj_encode_pipe_kernel (round 6): the parse role runs a block ahead of the entropy role on an UPPER BOUND of the previous block's compressed size and waits for the
    real answer where the bound decides nothing — data that does not compress, blocks of one repeated byte (RLE blocks), tiny last blocks, destinations with little room
    (the reference's "tight" detours: raw blocks, dstSize_tooSmall).  Every capacity around the frame's size gives the reference's answer, bytes or code; a wrong
    assumption would surface as error 1 (the entropy role checks each one against the block's real result)."""
    rnd = random.Random(66)
    noise = bytes(rnd.getrandbits(8) for _ in range(300000))
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    datas = [xml[1000:1000 + 300000], noise[:140000] + xml[:150000], xml[:131072] + bytes([7]) * 140000 + xml[5000:9000], bytes([9]) * 131072 + noise[:20000] + bytes([9]) * 131072,
             zj.synth_host(65536, 5, 1) * 4 + noise[:3], (noise[:5000] * 60)[:270000], xml[:131072] + noise[:131072] + xml[:6]]
    for k, d in enumerate(datas):
        for level in (3, 1):
            if len(d) > WINDOW[level]: continue
            full = oracle_ref.compress(d, level)
            assert emu_compress_multi(emu, d, level, pipelined=True) == full, (k, level)
            fs = len(full)
            for cap in sorted({fs - 1, fs, fs + 1, fs + 7, fs + 8, fs + 9, fs + 64, fs + 1024, fs + 1030, fs // 2, 17, 18, len(d), len(d) + 20, fs + 131072 + 1100}):
                try:
                    want = oracle_ref.compress(d, level, cap=cap)
                except oracle_ref.ZstdRefError as ex:
                    want = -ex.code
                got = emu_compress_multi(emu, d, level, pipelined=True, cap=cap)
                assert got == want, (k, level, cap, fs, got if isinstance(got, int) else len(got), want if isinstance(want, int) else len(want))
