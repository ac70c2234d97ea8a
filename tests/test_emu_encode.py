"""CPU (-m "not gpu"): the encoder kernel BODY (zstd-jni_amd/csrc/zj_encode.h), built lane-serial
(W = 1, tests/emu), is byte-identical to the reference's ZSTD_compress2 (level 3 with the LDS-sized
tables = hashLog 14 / chainLog 13).  The -m gpu tests repeat this through the C-ABI on the wave64 build."""
import random

import pytest

from conftest import golden
from util import edge_inputs, emu_lib, emu_compress, equal_count_inputs


@pytest.fixture(scope="module")
def emu():
    return emu_lib()


def expected(ref, data, level):
    return ref.compress(data, 3, False, 14, 13) if level == 3 else ref.compress(data, level)


@pytest.mark.parametrize("level", [1, 2, 3])
def test_emu_encoder_edge_inputs(emu, oracle_ref, level):
    for name, data in edge_inputs():
        assert emu_compress(emu, data, level) == expected(oracle_ref, data, level), name


def test_emu_encoder_xml_and_synthetic(emu, oracle_ref, zj):
    rnd = random.Random(11)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    for size in (4096, 12000, 16384, 65536, 100000, 131072):
        off = rnd.randrange(0, len(xml) - size)
        d = xml[off:off + size]
        for level in (1, 2, 3):
            assert emu_compress(emu, d, level) == expected(oracle_ref, d, level), (size, off, level)
    for _ in range(150):
        size = rnd.choice([rnd.randrange(0, 300), rnd.randrange(0, 5000), rnd.randrange(0, 70000), rnd.randrange(0, 131073), 65536, 4096, 131072])
        d = zj.synth_host(size, rnd.randrange(0, 100000), 1) if size else b""
        for level in (1, 2, 3):
            assert emu_compress(emu, d, level) == expected(oracle_ref, d, level), (size, level)


def test_emu_split_pipeline_matches(emu, oracle_ref, zj):
    """lane-per-frame match finding (records in HBM scratch) + entropy stage == fused path == reference"""
    rnd = random.Random(21)
    for name, data in edge_inputs():
        for level in (1, 2, 3):
            assert emu_compress(emu, data, level, split=True) == expected(oracle_ref, data, level), (name, level)
    for _ in range(100):
        size = rnd.choice([rnd.randrange(0, 300), rnd.randrange(0, 5000), rnd.randrange(0, 65537), 65536, 4096])
        d = zj.synth_host(size, rnd.randrange(0, 100000), 1) if size else b""
        for level in (1, 3):
            assert emu_compress(emu, d, level, split=True) == expected(oracle_ref, d, level), (size, level)
    # the wide launch: frames > 64 KiB (4-byte positions) and level-1/2 frames of 8-16 KiB (hashLog 15)
    for _ in range(40):
        size = rnd.choice([rnd.randrange(65537, 131073), 131072, 100000, rnd.randrange(8193, 16385), 12000])
        d = zj.synth_host(size, rnd.randrange(0, 100000), 1)
        for level in (1, 2, 3):
            assert emu_compress(emu, d, level, split=True) == expected(oracle_ref, d, level), (size, level)


def test_huf_sort_visits_the_count_164_slot(emu, oracle_ref, oracle_port):
    """HUF_sort's quicksort loop starts at RANK_POSITION_DISTINCT_COUNT_CUTOFF = 158 + highbit32(158) = 165 (the reference's
    comment says 166), i.e. at the slot of count == 164: nine or more literals with exactly that count get permuted by the
    unstable quicksort, which changes which of them receive the longer codes.  tests/golden/huf_sort_count164.bin is the input
    on which tools/fuzz_emu_encode.py (seed 82) found the restatements starting at 166."""
    d = golden("huf_sort_count164.bin")
    for level in (1, 2, 3):
        want = expected(oracle_ref, d, level)
        assert emu_compress(emu, d, level) == want and emu_compress(emu, d, level, split=True) == want, level
        assert oracle_port.compress(d, level, False, 14 if level == 3 else 0, 13 if level == 3 else 0) == want, level


def test_equal_literal_counts(emu, oracle_ref, oracle_port):
    """many literals with exactly equal counts: HUF_simpleQuickSort's behaviour on equal keys (every partition peels one
    element; the iterative side never falls back to insertion sort) reproduced exactly, without overflowing the sort stack"""
    for name, d in equal_count_inputs():
        for level in (1, 3):
            want = expected(oracle_ref, d, level)
            assert emu_compress(emu, d, level) == want and emu_compress(emu, d, level, split=True) == want, (name, level)
            assert oracle_port.compress(d, level, False, 14 if level == 3 else 0, 13 if level == 3 else 0) == want, (name, level)


def test_code_tables_closed_form(emu):
    """ZSTD_LLcode / ZSTD_MLcode / LL_bits / ML_bits as arithmetic == the format's tables (N/common/zstd_internal.h:114-140,
    N/compress/zstd_compress_internal.h:584-616), every input"""
    emu.emu_check_code_tables.restype = __import__("ctypes").c_uint
    assert emu.emu_check_code_tables() == 0


def test_emu_explicit_table_sizes(emu, oracle_ref, zj):
    """ZstdCompressCtx.setHashLog / setChainLog on the lane-per-frame path: byte-identical to the reference given the same
    ZSTD_c_hashLog / ZSTD_c_chainLog; 16 / 15 is the reference's plain level 3"""
    rnd = random.Random(77)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    for _ in range(36):
        size = rnd.choice([rnd.randrange(64, 5000), rnd.randrange(1, 65537), 65536, 65536, rnd.randrange(65537, 131073), 131072])
        d = xml[:size] if rnd.random() < 0.3 else zj.synth_host(size, rnd.randrange(0, 100000), 1)
        for hl, cl in ((16, 15), (17, 16), (15, 14), (12, 12), (0, 15), (16, 0), (6, 6)):
            want = oracle_ref.compress(d, 3, False, hl, cl)
            assert emu_compress(emu, d, 3, split=True, hash_log=hl, chain_log=cl) == want, (size, hl, cl)
            if (hl, cl) == (16, 15):
                assert want == oracle_ref.compress(d, 3)                  # the level's own table sizes


def test_emu_encoder_small_inputs_match_default_level3(emu, oracle_ref, zj):
    rnd = random.Random(3)
    for _ in range(100):
        size = rnd.randrange(0, 8193)
        d = zj.synth_host(size, rnd.randrange(0, 100000), 1) if size else b""
        assert emu_compress(emu, d, 3) == oracle_ref.compress(d, 3), size
    assert emu_compress(emu, golden("xmlsmall"), 3) == golden("xmlsmall-sized.zst")


def test_emu_encoder_dst_too_small(emu, zj):
    import ctypes as C
    d = zj.synth_host(65536, 0, 1)
    for cap in (5, 12, 1000):
        dst = C.create_string_buffer(cap)
        r = emu.emu_compress(d, len(d), dst, cap, 3)
        assert (1 << 64) - r == 70


def test_emu_tight_destinations(emu, oracle_ref, zj):
    """a destination between the frame's size and Zstd.compressBound: the reference compresses straight into it and its writers want
    working room (8 bytes of slack behind every bit stream, two-byte stores of table descriptions, 18 bytes for any frame header), so
    it answers dstSize_tooSmall although the frame would fit, or — when the block fits raw — emits the raw block
    (N/compress/zstd_compress.c:3024-3030).  Same answer here for every capacity (T/scala/Zstd.scala:186-201 is the reference's own
    near-exact-destination test); tools/fuzz_emu_tight.py is the randomised version (levels 1-8, multi-block, dictionaries)."""
    import ctypes as C
    import random
    rnd = random.Random(77)
    def small(n, a):
        d = bytearray(rnd.randrange(a) for _ in range(n))
        if n > 40: d[n - 12:n - 4] = d[3:11]
        return bytes(d)
    def barely(seed):                                  # small, barely compressible: the window where the reference emits the block raw instead
        r = random.Random(seed)
        n = r.randrange(20, 400); a = r.choice([3, 6, 12, 24, 48, 100]); d = bytearray(r.randrange(a) for _ in range(n))
        for _ in range(r.choice([0, 1, 2, 4])):
            ln = r.randrange(4, 12); at = r.randrange(0, max(1, n - 2 * ln)); to = r.randrange(at + ln, max(at + ln + 1, n - ln + 1))
            d[to:to + ln] = d[at:at + ln]
        return bytes(d[:n])
    inputs = [barely(2), barely(32), barely(49), barely(57), b"", b"a", b"abcdefg" * 3, small(70, 4), small(120, 12), small(200, 40), small(380, 100), bytes(rnd.getrandbits(8) for _ in range(300)),
              golden("xmlsmall")[:3000], zj.synth_host(9000, 5, 1), zj.synth_host(65536, 1, 1), b"\x07" * 5000, zj.synth_host(140000, 3, 1)]
    seen = {"refused although it fits": 0, "raw instead": 0}
    for data in inputs:
        for level in ((1, 3) if len(data) > 20000 else (1, 2, 3, 5)):
            hl, cl = (14, 13) if (level == 3 and 8192 < len(data) <= 131072) else (0, 0)
            for ck in (False, True):
                full = oracle_ref.compress(data, level, ck, hl, cl)
                fn = emu.emu_compress_multi if (len(data) > 131072 or level >= 4) else emu.emu_compress
                caps = list(range(max(0, len(full) - 2), len(full) + 24)) + [0, 8, 17, 18, len(data), len(data) + 3, len(data) + 9, len(data) + 12, len(data) + 20]
                for cap in (caps if len(data) < 20000 else caps[::3]):
                    try: want = oracle_ref.compress(data, level, ck, hl, cl, cap=cap)
                    except oracle_ref.ZstdRefError as e: want = -e.code
                    dst = C.create_string_buffer(cap + 8)
                    r = fn(data, len(data), dst, cap, level | (int(ck) << 8))
                    got = -((1 << 64) - r) if r >= (1 << 63) else dst.raw[:r]
                    assert got == want, (len(data), level, ck, cap, len(full))
                    if isinstance(want, int) and cap >= len(full): seen["refused although it fits"] += 1
                    if not isinstance(want, int) and want != full: seen["raw instead"] += 1
    assert seen["refused although it fits"] > 500 and seen["raw instead"] > 0, seen


@pytest.mark.parametrize("mode", ["1", "2", "5", "6", "7", "8", "5j3", "6j7"])
def test_emu_need_gated_double_fast(emu, oracle_ref, zj, monkeypatch, mode):
    """the need-gated double-fast machine (zj_need.h, ZJNI_NEED=1 in the library; ZJNI_EMU_NEED=1 here): table probes are made only where some
    other position of the frame carries the probe's key, table writes only into buckets such a probe reads — decided per position by Bloom
    filters ahead of the parse.  Same frames as the ungated machine = the reference's (tools/fuzz_emu_need.py: 1.8 million frames in both modes, 0 differences)."""
    monkeypatch.setenv("ZJNI_EMU_NEED", mode[0])              # ZLaneD: 1 flags for every frame, 2 only for the frames zn_worth() picks (the gated machine without flags for the rest);
    if len(mode) > 1:                                         # the run machine (zj_match_run.h, the product's large-batch level-3 machine): 5 flags for every frame, 6 for the
        monkeypatch.setenv("ZJNI_EMU_JMAX", mode[2])          # picked ones, 7 for none, 8 for every frame but taken over LATE (mid-frame, as the match kernel does while the flag kernel is still at work); jN: runs of up to N quiet positions per round
    rnd = random.Random(41)
    datas = [zj.synth_host(65536, k, 1) for k in range(8)] + [zj.synth_host(s, 100 + s, 1) for s in (64, 65, 1000, 8192, 8193, 30000, 65535)]
    datas += [bytes([7]) * 40000, bytes(rnd.getrandbits(8) for _ in range(20000)), (b"abcdefgh" * 5000)[:33333], golden("xmlsmall")[:60000]]
    for d in datas:
        assert emu_compress(emu, d, 3, split=True) == expected(oracle_ref, d, 3), len(d)
        for hl, cl in ((15, 15), (12, 9)):
            assert emu_compress(emu, d, 3, split=True, hash_log=hl, chain_log=cl) == oracle_ref.compress(d, 3, False, hl, cl), (len(d), hl, cl)


def test_emu_wide_launch_on_the_run_machine(emu, oracle_ref, zj, monkeypatch):
    """frames of 64-128 KiB (the wide launch, zj_enc_match_wide_kernel) run ZLaneR without flags since round 4 — register windows, no quiet runs:
    the reference's bytes for sizes around both ends of the range, with the level's own and explicit table sizes"""
    monkeypatch.setenv("ZJNI_EMU_NEED", "7")
    rnd = random.Random(43)
    datas = [zj.synth_host(131072, k, 1) for k in range(4)] + [zj.synth_host(s, 200 + s, 1) for s in (65537, 65544, 70000, 100000, 131071)]
    datas += [bytes([9]) * 90000, bytes(rnd.getrandbits(8) for _ in range(70000)), (b"abcdefghij" * 13200)[:131000], (golden("xmlsmall") * 1300)[:131072]]
    for d in datas:
        assert emu_compress(emu, d, 3, split=True) == expected(oracle_ref, d, 3), len(d)
        assert emu_compress(emu, d, 3, split=True, hash_log=15, chain_log=16) == oracle_ref.compress(d, 3, False, 15, 16), len(d)


@pytest.mark.parametrize("mode", ["5", "6", "8"])
def test_emu_wide_launch_with_need_flags(emu, oracle_ref, zj, monkeypatch, mode):
    """frames of 64-128 KiB with need flags (zn_flags_frame_wide: a table at a time over filters twice the size) on the run machine — flags for every frame (5),
    for the frames zn_worth() picks (6), taken over mid-frame (8): the reference's bytes"""
    monkeypatch.setenv("ZJNI_EMU_NEED", mode)
    rnd = random.Random(47)
    datas = [zj.synth_host(131072, k, 1) for k in range(8)] + [zj.synth_host(s, 300 + s, 1) for s in (65537, 65544, 70000, 100000, 131071)]
    datas += [bytes([9]) * 90000, bytes(rnd.getrandbits(8) for _ in range(70000)), bytes(rnd.randrange(16) for _ in range(131072)), bytes(rnd.randrange(4) for _ in range(100001)),
              (b"abcdefghij" * 13200)[:131000], (golden("xmlsmall") * 1300)[:131072]]
    for d in datas:
        assert emu_compress(emu, d, 3, split=True) == expected(oracle_ref, d, 3), len(d)
        assert emu_compress(emu, d, 3, split=True, hash_log=15, chain_log=15) == oracle_ref.compress(d, 3, False, 15, 15), len(d)


def test_need_flags_cover_the_exact_answer(emu, zj):
    """zj_need.h's contract, checked without the parse: a position whose key (long: its 8 bytes; short: its bucket and its first 4 bytes) some
    OTHER position of the frame shares must carry the NEED flag, and every position in the bucket of a NEED-flagged position must carry the INS
    flag — Bloom filters may add flags (they only cost requests), never drop one.  Also: how many they add stays small."""
    import ctypes as C
    emu.emu_need_flags.restype = C.c_uint
    rnd = random.Random(9)
    frames = [zj.synth_host(65536, k, 1) for k in range(4)] + [zj.synth_host(20000, 11, 1), bytes(rnd.randrange(16) for _ in range(30000)),
              (b"0123456789abcdef" * 4096)[:65536], bytes([3]) * 5000, bytes(rnd.getrandbits(8) for _ in range(4096))]
    frames += [zj.synth_host(131072, 20 + k, 1) for k in range(4)] + [zj.synth_host(65537, 5, 1), bytes(rnd.randrange(16) for _ in range(131072)), bytes(rnd.getrandbits(8) for _ in range(100000))]      # the wide launch's sizes: zn_flags_frame_wide
    for d in frames:
        n = len(d); flags = C.create_string_buffer(n + 16); prm = (C.c_uint * 3)()
        assert emu.emu_need_flags(d, n, flags, prm) == 1
        f = flags.raw[:n]; npos = n - 7
        bk = (C.c_uint * 2)(); bl, bs = [0] * npos, [0] * npos
        for p in range(npos):
            emu.emu_need_buckets(d, n, p, bk); bl[p], bs[p] = bk[0], bk[1]
        seen_l, seen_s = {}, {}
        for p in range(npos):
            seen_l.setdefault(d[p:p + 8], []).append(p); seen_s.setdefault((bs[p], d[p:p + 4]), []).append(p)
        need_l = {p for ps in seen_l.values() if len(ps) > 1 for p in ps}; need_s = {p for ps in seen_s.values() if len(ps) > 1 for p in ps}
        assert all(f[p] & 1 for p in need_l) and all(f[p] & 2 for p in need_s), n
        flagged_l = {bl[p] for p in range(npos) if f[p] & 1}; flagged_s = {bs[p] for p in range(npos) if f[p] & 2}
        assert all(f[p] & 4 for p in range(npos) if bl[p] in flagged_l) and all(f[p] & 8 for p in range(npos) if bs[p] in flagged_s), n
        assert all(f[p] == 0 for p in range(npos, n))
        extra_l = sum(1 for p in range(npos) if (f[p] & 1) and p not in need_l); extra_s = sum(1 for p in range(npos) if (f[p] & 2) and p not in need_s)
        assert extra_l <= 0.08 * npos and extra_s <= 0.08 * npos, (n, extra_l, extra_s)          # false positives of the filters: ~5 % with 65 536 keys in


def test_emu_encoder_tiny_text_frames(emu, oracle_ref):
    """short natural-text frames sit right at the compressed-vs-raw block decision (ZSTD_minGain): both
    pipelines must take the reference's side of it (regression: 70-byte frame with 0 sequences)"""
    rnd = random.Random(5)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    for _ in range(1500):
        size = rnd.randrange(1, 400)
        off = rnd.randrange(0, len(xml) - size)
        d = xml[off:off + size]
        for level in (1, 3):
            want = expected(oracle_ref, d, level)
            assert emu_compress(emu, d, level) == want, (size, off, level)
            assert emu_compress(emu, d, level, split=True) == want, (size, off, level, "split")


def test_emu_checksum_frames(emu, oracle_ref, zj):
    """ZSTD_c_checksumFlag: frames carry XXH64(content) & 0xFFFFFFFF and stay byte-identical to the reference's
    (ZstdCompressCtx.setChecksum(true), T/scala/Zstd.scala checksum tests); both decoders verify it"""
    from util import emu_decompress, emu_decompress_split
    rnd = random.Random(77)
    cases = [d for _, d in edge_inputs() if len(d) <= 131072]
    for _ in range(60):
        size = rnd.choice([rnd.randrange(0, 300), rnd.randrange(0, 5000), rnd.randrange(0, 65537), 65536, 31, 32, 33, 63, 64])
        cases.append(zj.synth_host(size, rnd.randrange(0, 100000), 1) if size else b"")
    for d in cases:
        for level in (1, 3):
            want = oracle_ref.compress(d, 3, True, 14, 13) if level == 3 else oracle_ref.compress(d, level, True)
            got = emu_compress(emu, d, level, checksum=True)
            assert got == want, (len(d), level)
            if len(d) <= 65536:
                assert emu_compress(emu, d, level, split=True, checksum=True) == want, (len(d), level, "split")
            assert emu_decompress(emu, got, len(d)) == d
            assert emu_decompress_split(emu, got, len(d))[0] == d
            if len(got) > 12:
                bad = bytearray(got); bad[-1] ^= 0x01                  # wrong checksum byte
                assert emu_decompress(emu, bytes(bad), len(d)) == -22
                assert emu_decompress_split(emu, bytes(bad), len(d))[0] == -22


def test_emu_frame_header_flags(emu, oracle_ref, zj):
    """ZstdCompressCtx.setContentSize(false) (ZSTD_c_contentSizeFlag = 0: window descriptor instead of the content size,
    N/compress/zstd_compress.c:4695-4745), with and without checksum, both pipelines; every decoder still takes the frames"""
    from util import emu_decompress, emu_decompress_split
    rnd = random.Random(123)
    cases = [d for _, d in edge_inputs() if len(d) <= 131072]
    for _ in range(40):
        size = rnd.choice([rnd.randrange(0, 300), rnd.randrange(0, 5000), rnd.randrange(0, 131073), 65536, 255, 256, 1023, 1024, 1025])
        cases.append(zj.synth_host(size, rnd.randrange(0, 100000), 1) if size else b"")
    for d in cases:
        for level in (1, 2, 3):
            for ck in (False, True):
                want = oracle_ref.compress(d, level, ck, 14 if level == 3 else 0, 13 if level == 3 else 0, content_size=False)
                got = emu_compress(emu, d, level, checksum=ck, content_size=False)
                assert got == want, (len(d), level, ck)
                assert emu_compress(emu, d, level, split=True, checksum=ck, content_size=False) == want, (len(d), level, ck, "split")
        assert emu_decompress(emu, got, len(d)) == d and emu_decompress_split(emu, got, len(d))[0] == d
