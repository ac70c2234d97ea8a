"""CPU (-m "not gpu"): the wave-parallel tANS table construction of the entropy stage (zj_encode.h: ze_tans_shares / ze_tans_describe / ze_tans_table — a symbol per
lane, prefix sums and reductions instead of the reference's running state) in the lane-serial emulation against the reference's own FSE_normalizeCount /
FSE_writeNCount / FSE_buildCTable_wksp (oracle/_ref/libzstd_ref.so exports them) on random and adversarial histograms: the shares, the description bytes, the
destination size the reference's writer asks for, and every cell of the encoding table.  The uncommon branch of the share computation (the largest symbol cannot absorb
the rounding surplus) is counted: the histogram families below reach it thousands of times."""
import ctypes as C
import random

import pytest

from util import emu_lib

RTB = [0, 473195, 504333, 520860, 550000, 700000, 750000, 830000]


def takes_uncommon_branch(count, total, table_log, low):
    """the first pass of FSE_normalizeCount (N/compress/fse_compress.c:465-523) in python integers: does it fall through to its second method?"""
    scale = 62 - table_log; step = (1 << 62) // total; vstep = 1 << (scale - 20)
    still = 1 << table_log; largest = 0; largest_p = 0; norm = [0] * len(count)
    for s, c in enumerate(count):
        if c == 0: continue
        if c <= (total >> table_log): norm[s] = -1 if low else 1; still -= 1
        else:
            p = (c * step) >> scale
            if p < 8 and (c * step) - (p << scale) > vstep * RTB[p]: p += 1
            if p > largest_p: largest_p = p; largest = s
            norm[s] = p; still -= p
    return -still >= (norm[largest] >> 1)


@pytest.fixture(scope="module")
def libs(oracle_ref):
    R = oracle_ref.lib()
    R.FSE_normalizeCount.restype = C.c_size_t
    R.FSE_normalizeCount.argtypes = [C.POINTER(C.c_short), C.c_uint, C.POINTER(C.c_uint), C.c_size_t, C.c_uint, C.c_uint]
    R.FSE_writeNCount.restype = C.c_size_t
    R.FSE_writeNCount.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_short), C.c_uint, C.c_uint]
    R.FSE_buildCTable_wksp.restype = C.c_size_t
    R.FSE_buildCTable_wksp.argtypes = [C.c_void_p, C.POINTER(C.c_short), C.c_uint, C.c_uint, C.c_void_p, C.c_size_t]
    R.FSE_optimalTableLog.restype = C.c_uint
    R.FSE_optimalTableLog.argtypes = [C.c_uint, C.c_size_t, C.c_uint]
    L = emu_lib()
    L.emu_tans.restype = C.c_int
    L.emu_tans.argtypes = [C.POINTER(C.c_uint), C.c_uint, C.c_uint, C.c_uint, C.c_int, C.POINTER(C.c_short), C.c_char_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                           C.POINTER(C.c_ushort), C.POINTER(C.c_int), C.POINTER(C.c_uint)]
    return R, L


def histogram(rnd, n):
    """n symbols, the last one present; families: flat, geometric, one dominant, many singletons, near the rounding bars, sparse"""
    k = rnd.randrange(7)
    if k == 0: c = [rnd.randrange(1, 50) for _ in range(n)]
    elif k == 1: c = [max(0, int(rnd.choice([3000, 300, 30]) * (rnd.uniform(0.5, 0.95) ** i)) + rnd.randrange(0, 2)) for i in range(n)]
    elif k == 2: c = [rnd.randrange(0, 3) for _ in range(n)]; c[rnd.randrange(n)] = rnd.choice([50, 1000, 60000])
    elif k == 3: c = [1] * n; c[rnd.randrange(n)] += rnd.randrange(0, 2000)
    elif k == 4:                                                   # counts around 1.5 shares of a 2^tl table: everyone rounds up
        base = rnd.choice([3, 7, 11, 23, 47, 95]); c = [base + rnd.randrange(0, 2) for _ in range(n)]
    elif k == 5: c = [rnd.choice([0, 0, 0, 1, 2, 40]) for _ in range(n)]
    else: c = [rnd.randrange(0, 4) * rnd.randrange(0, 4) * rnd.randrange(1, 200) for _ in range(n)]
    if c[-1] == 0: c[-1] = rnd.randrange(1, 5)
    if sum(1 for x in c if x) < 2: c[0] += 1 + rnd.randrange(3)
    return c


def test_tables_equal_the_references(libs):
    R, L = libs
    rnd = random.Random(31)
    uncommon = lows = failed = 0
    for case in range(60000):
        n = rnd.choice([2, 3, 5, 13, 29, 32, 36, 53, rnd.randrange(2, 54)])
        count = histogram(rnd, n); total = sum(count); max_sv = n - 1
        if max(count) == total: continue
        max_log = rnd.choice([6, 8, 9, 9])
        tl = R.FSE_optimalTableLog(max_log, total, max_sv) if rnd.random() < 0.8 else rnd.randrange(5, 10)
        if tl < min(total.bit_length(), max_sv.bit_length() + 1): continue      # below FSE_minTableLog: an error there up front, and no caller asks (optimal log >= it)
        low = total >= 2048 if rnd.random() < 0.7 else rnd.random() < 0.5
        carr = (C.c_uint * 64)(*count)
        want_norm = (C.c_short * 64)()
        r = R.FSE_normalizeCount(want_norm, tl, carr, total, max_sv, int(low))
        got_norm = (C.c_short * 64)(); desc = C.create_string_buffer(128); dsz = C.c_uint(0); need = C.c_uint(0)
        state = (C.c_ushort * 512)(); dfind = (C.c_int * 64)(); dnb = (C.c_uint * 64)()
        ok = L.emu_tans(carr, max_sv, total, tl, int(low), got_norm, desc, C.byref(dsz), C.byref(need), state, dfind, dnb)
        uncommon += takes_uncommon_branch(count, total, tl, low); lows += any(want_norm[s] == -1 for s in range(n))
        if r > (1 << 62):                                           # an error there (a share below one cell in the second method)
            assert not ok, (case, count, tl); failed += 1; continue
        assert ok, (case, count, tl)
        assert list(got_norm)[:n] == list(want_norm)[:n], (case, count, tl, low)
        buf = C.create_string_buffer(512)
        h = R.FSE_writeNCount(buf, 512, want_norm, max_sv, tl)
        assert h < 512 and dsz.value == h and desc.raw[:h] == buf.raw[:h], (case, count, tl, low, h, dsz.value)
        # the smallest destination the reference's writer accepts = `need`
        assert R.FSE_writeNCount(buf, need.value, want_norm, max_sv, tl) == h, (case, need.value, h)
        assert R.FSE_writeNCount(buf, need.value - 1, want_norm, max_sv, tl) > (1 << 62), (case, need.value, h)
        size = 1 << tl
        ct = (C.c_uint * (1 + size // 2 + 2 * 64 + 8))(); wk = (C.c_uint * 4096)()
        assert R.FSE_buildCTable_wksp(ct, want_norm, max_sv, tl, wk, C.sizeof(wk)) == 0
        raw = bytes(ct)
        want_state = [int.from_bytes(raw[4 + 2 * u: 6 + 2 * u], "little") for u in range(size)]
        assert list(state)[:size] == want_state, (case, count, tl, low)
        tt = 4 * (1 + size // 2)
        for s in range(n):
            find = int.from_bytes(raw[tt + 8 * s: tt + 8 * s + 4], "little", signed=True); nb = int.from_bytes(raw[tt + 8 * s + 4: tt + 8 * s + 8], "little")
            assert dnb[s] == nb, (case, s, count, tl)
            if want_norm[s] != 0: assert dfind[s] == find, (case, s, count, tl)
    print("uncommon branch", uncommon, "tables with low-probability symbols", lows, "refused", failed)
    assert uncommon > 1000 and lows > 5000, (uncommon, lows, failed)
