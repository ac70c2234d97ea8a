"""Level 4 on the CPU: the kernel body of zj_encode_multi_kernel's single-block route (ze_compress_t with its match-finder tables in
HBM) lane-serial against the reference's ZSTD_compress2 at level 4 — greedy on the hash chain up to 16 KiB (ZSTD_compressBlock_greedy /
ZSTD_HcFindBestMatch, N/compress/zstd_lazy.c:667-723,1516-1780), double-fast with 2^17-entry tables from there to 128 KiB
(N/compress/clevels.h:84,110).  Byte identity, every size class, flags."""
import os
import random

import pytest

from util import emu_lib, emu_compress_multi, emu_compress_chain, json_records


@pytest.fixture(scope="module")
def L():
    return emu_lib()


def _cases(zj):
    rnd = random.Random(44)
    recs = json_records(6000, seed=9)
    text = b",".join(recs)
    out = [b"", b"a", b"ab" * 3, b"abcdefg", bytes(8), os.urandom(9), b"x" * 70000, os.urandom(5000), text[:16384], text[:16385], text[:4096], text[:1000],
           text[:131072], text[:131071], text[:65536], text[:65537], zj.synth_host(16384, 2, 1), zj.synth_host(16384, 6, 1), zj.synth_host(65536, 2, 1), zj.synth_host(131072, 5, 1)]
    for _ in range(160):
        n = rnd.choice([rnd.randrange(1, 300), rnd.randrange(300, 4097), rnd.randrange(4097, 16385), rnd.randrange(16385, 70000), rnd.randrange(70000, 131073)])
        k = rnd.randrange(5)
        if k == 0: d = text[rnd.randrange(0, len(text) - n):][:n]
        elif k == 1: d = zj.synth_host(n, rnd.randrange(1 << 20), 1)
        elif k == 2: per = os.urandom(rnd.choice([1, 2, 3, 5, 17, 64, 300])); d = (per * (n // len(per) + 1))[:n]
        elif k == 3: a = rnd.choice([2, 3, 5, 16, 64]); d = bytes(rnd.randrange(a) + 65 for _ in range(n))
        else: h = n // 2; d = text[:h] + os.urandom(n - h)
        out.append(d)
    return out


def test_emu_level4_byte_identical(L, zj, oracle_ref):
    strategies = set()
    for d in _cases(zj):
        for ck, cs in ((False, True), (True, True), (False, False)):
            got = emu_compress_multi(L, d, 4, ck, cs)
            want = oracle_ref.compress(d, 4, ck, content_size=cs)
            assert got == want, (len(d), ck, cs, got if isinstance(got, int) else len(got), len(want))
            if len(d) <= 16384:
                assert emu_compress_chain(L, d, 4, ck, cs) == want, ("chain route", len(d), ck, cs)
        strategies.add("greedy" if len(d) <= 16384 else "dfast")
    assert strategies == {"greedy", "dfast"}


def test_emu_level4_beyond_128k_is_refused(L, zj):
    assert emu_compress_multi(L, zj.synth_host(131073, 1, 1), 4) == -201


@pytest.mark.parametrize("level", [5, 6, 7, 8])
def test_emu_lazy_levels_small_inputs_byte_identical(L, zj, oracle_ref, level):
    """levels 5-8 up to 16 KiB: lazy / lazy2 on the hash chain (ZSTD_compressBlock_lazy_generic depth 1 / 2, N/compress/zstd_lazy.c:1624-1700)
    and the cost-based choice between the predefined and a new tANS table (strategy >= lazy, N/compress/zstd_compress_sequences.c:196-222)"""
    n = rows = 0
    for d in _cases(zj):
        for ck, cs in ((False, True), (True, False)):
            got = emu_compress_multi(L, d, level, ck, cs)
            want = oracle_ref.compress(d, level, ck, content_size=cs)
            assert got == want, (level, len(d), ck, cs, got if isinstance(got, int) else len(got), len(want))
            if len(d) <= 16384:
                assert emu_compress_chain(L, d, level, ck, cs) == want, ("chain route", level, len(d), ck, cs)  # the large-batch route: parser per lane, entropy stage on its records
        n += 1; rows += len(d) > 16384                          # above 16 KiB: the row-based finder (ZSTD_RowFindBestMatch)
    assert n > 60 and rows > 20
    assert emu_compress_multi(L, zj.synth_host(131073, 1, 1), level) == -201
