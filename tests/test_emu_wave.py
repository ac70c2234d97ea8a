"""CPU (-m "not gpu"): the wave-per-frame double-fast matcher (zstd-jni_amd/csrc/zj_match_wave.h) in its explicit-SIMT
build — 64 emulated lanes, cross-lane traffic only through ballot / shuffle / LDS between lane blocks, exactly the points
where the GPU build exchanges data — followed by the entropy stage is byte-identical to the reference's ZSTD_compress2
(level 3, hashLog 14 / chainLog 13).  Two builds: when lanes store to the same LDS address in one step, the lowest resp.
the highest lane wins (the GPU promises neither).  The -m gpu tests repeat this through the C-ABI on the wave64 build."""
import random

import pytest

from conftest import golden
from util import edge_inputs, emu_wave_libs, emu_compress_wave, equal_count_inputs


@pytest.fixture(scope="module")
def waves():
    return emu_wave_libs()


def want(ref, data, checksum=False):
    return ref.compress(data, 3, checksum, 14, 13)


def check(waves, ref, data, tag, checksum=False):
    w = want(ref, data, checksum)
    for k, L in enumerate(waves):
        assert emu_compress_wave(L, data, checksum=checksum) == w, (tag, "descending" if k else "ascending")


def test_wave_matcher_edge_inputs(waves, oracle_ref):
    took = 0
    for name, data in edge_inputs():
        if 64 <= len(data) <= 65536:
            check(waves, oracle_ref, data, name)
            took += 1
        else:
            assert emu_compress_wave(waves[0], data) is None, name
    assert took >= 8


def test_wave_matcher_synthetic_and_xml(waves, oracle_ref, zj):
    rnd = random.Random(31)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    for size in (64, 65, 100, 4096, 12000, 16384, 16385, 40000, 65535, 65536):
        off = rnd.randrange(0, len(xml) - size)
        check(waves, oracle_ref, xml[off:off + size], (size, off))
    for k in range(160):
        size = rnd.choice([rnd.randrange(64, 300), rnd.randrange(64, 5000), rnd.randrange(64, 65537), 65536, 4096])
        check(waves, oracle_ref, zj.synth_host(size, rnd.randrange(0, 100000), 1), (size, k), bool(k & 1))


def hard_cases(rnd):
    """inputs that drive the parts a text frame never reaches: step > 1 windows (long stretches without a match), matches
    longer than one counting step, backward extension over more than 8 / 128 bytes, equal hashes inside one window"""
    noise = bytes(rnd.getrandbits(8) for _ in range(70000))
    cases = [noise[:3000] + noise[100:2500] + noise[3000:20000] + noise[5000:9000] + noise[20000:60000],          # long matches between noise
             noise[:30000] + b"\x00" * 2000 + noise[:30000],                                                       # 30 000-byte match, zero run
             bytes([rnd.choice(b"ab")]) * 10 + bytes(rnd.choice(b"ab") for _ in range(65000)),                     # two-letter alphabet: windows full of equal hashes
             b"".join(bytes([i & 255]) * rnd.randrange(1, 40) for i in range(4000))[:65536],                       # byte runs: overlapping matches at offset 1
             noise[:1000] + noise[8:1000] + noise[:600] + noise[300:1000] * 20,
             (noise[:37] * 2000)[:65536], (noise[:64] * 1100)[:65536], (noise[:700] * 100)[:65536],                # periods below / at / above the window
             noise[:5000] + noise[4000:4990] + b"#" + noise[4000:5000] + noise[100:400] + b"!" + noise[99:5000],   # backward extensions of 10 .. 300 bytes
             bytes(rnd.choice(b"0123456789abcdef") for _ in range(65536)),                                          # 16 symbols: short chance matches everywhere
             bytes(rnd.getrandbits(8) & 0x0F if rnd.random() < 0.125 else 0 for _ in range(65536))]
    return cases


def test_wave_matcher_long_strides_and_repeats(waves, oracle_ref):
    rnd = random.Random(5)
    for k, d in enumerate(hard_cases(rnd)):
        for cut in (len(d), 65536, 5000, 777):
            check(waves, oracle_ref, d[:min(cut, 65536)], (k, cut))


def test_wave_matcher_equal_counts(waves, oracle_ref):
    for name, d in equal_count_inputs():
        if 64 <= len(d) <= 65536:
            check(waves, oracle_ref, d, name)
