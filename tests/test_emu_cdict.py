"""CPU (-m "not gpu"): dictionary compression bodies (zstd-jni_amd/csrc/zj_cdict.h + the dictionary branches of
zj_encode.h), built lane-serial (tests/emu), are byte-identical to the reference's ZstdDictCompress path:
ZSTD_createCDict(dict, level) + ZSTD_CCtx_refCDict + ZSTD_compress2 (N/jni_fast_zstd.c:26,325-336,586-640), which
equals ZSTD_compress_usingCDict (N/jni_fast_zstd.c:171-216) on these inputs.  Covered: trained dictionaries of the
four parameter classes (<= 16 KiB, <= 128 KiB, <= 256 KiB, larger), raw-content dictionaries, hand-assembled
dictionaries whose tables do not cover every symbol ("check" repeat modes) and 1/2/4-byte dictIDs, sources from 0
bytes to the attach cutoff, and the refusal above it."""
import random

import pytest

import dictutil as du
from util import EmuCDict, emu_lib, json_records

WORDS = [b"the", b"quick", b"brown", b"fox", b"jumps", b"over", b"lazy", b"dog", b"lorem", b"ipsum", b"dolor", b"sit", b"amet", b"zstd", b"frame", b"block"]


@pytest.fixture(scope="module")
def emu():
    return emu_lib()


def text(n, r):
    out = bytearray()
    while len(out) < n:
        out += r.choice(WORDS) + b" "
    return bytes(out[:n])


def lowent(n, r):
    out = bytearray()
    for i in range(n):
        out.append(out[i - r.randrange(1, min(i, 64) + 1)] if i and r.random() < 7 / 8 else r.randrange(16))
    return bytes(out)


def sources(recs, r, cutoff):
    out = []
    for size in [0, 1, 6, 7, 8, 9, 63, 64, 100, 255, 256, 1000, 1023, 1024, 1025, 4096, 8191, 8192, 12000, 16384]:
        if size > cutoff:
            continue
        k = r.randrange(0, len(recs) - 300)
        out += [b",".join(recs[k:k + 200])[:size], text(size, r), lowent(size, r), bytes(r.getrandbits(8) for _ in range(size))]
    return out


def check(emu, ref, dictionary, level, srcs_of):
    cd = ref.CDict(dictionary, level)
    ecd = EmuCDict(emu, dictionary, level)
    info = ecd.info()
    cutoff = 16384 if info["strategy"] == 2 else 8192
    for x in srcs_of(cutoff):
        want = cd.compress(x)
        assert ecd.compress(x) == want, (len(dictionary), level, len(x))
        assert want == cd.compress_using(x)
        assert ref.decompress_using_dict(want, dictionary, len(x)) == x
    # beyond the attach range the reference copies the dictionary's tables and searches it as an external segment (copy mode,
    # ZSTD_resetCCtx_byCopyingCDict + ZSTD_compressBlock_{fast,doubleFast}_extDict): one block, with the dictionary's own parameters
    r2 = random.Random(len(dictionary) * 7 + level)
    recs2 = json_records(4000, seed=11)
    content = max(1, info.get("contentSize", len(dictionary)))
    for size in (cutoff + 1, cutoff + 100, 20000, 32768, 50000, 65536, 100000, 131071, 131072):
        if size <= cutoff:
            continue
        k = r2.randrange(0, len(recs2) - 3000)
        for x in (b",".join(recs2[k:k + 3000])[:size], text(size, r2), dictionary[-min(len(dictionary), size // 2):] + lowent(size - min(len(dictionary), size // 2), r2),
                  bytes(r2.getrandbits(8) for _ in range(size // 3)) + text(size - size // 3, r2)):
            assert len(x) == size
            want = cd.compress(x)
            got = ecd.compress(x)
            if size == 131072 and size >= 6 * content:             # the reference reloads the dictionary with the source's parameters there: not served
                assert got == -40, (len(dictionary), level, size)
                continue
            assert got == want, ("copy mode", len(dictionary), level, size, got if isinstance(got, int) else len(got), len(want))
    assert ecd.compress(bytes(131073)) == -40                # multi-block frames with a dictionary: parameter_unsupported
    return info


@pytest.mark.parametrize("dict_size", [2048, 16384, 112640, 200000, 300000])
def test_emu_cdict_trained(emu, oracle_ref, dict_size):
    r = random.Random(dict_size)
    recs = json_records(30000, seed=3)
    samples = [b",".join(recs[i * 13:i * 13 + 200])[:4096] for i in range(1500)] + [text(4096, r) for _ in range(200)]
    d = oracle_ref.train_dict(samples, dict_size)
    for level in (1, 2, 3):
        info = check(emu, oracle_ref, d, level, lambda cut: sources(recs, r, cut))
        assert (info["hufRepeat"], info["llRepeat"], info["ofRepeat"], info["mlRepeat"]) == (2, 2, 2, 2)
        assert info["dictID"] == oracle_ref.dict_id(d)


def test_emu_cdict_raw_content(emu, oracle_ref):
    r = random.Random(5)
    recs = json_records(3000, seed=9)
    for d in (text(50000, r), b",".join(recs[:300]), b"0123456789abcdef" * 2, bytes(r.getrandbits(8) for _ in range(5000))):
        for level in (1, 2, 3):
            info = check(emu, oracle_ref, d, level, lambda cut: sources(recs, r, min(cut, 4096)))
            assert info["dictID"] == 0 and info["hufRepeat"] == 0


def test_emu_cdict_check_modes_and_dict_ids(emu, oracle_ref):
    r = random.Random(11)
    recs = json_records(5000, seed=5)
    content = b",".join(recs[:150])
    hist = [0] * 256
    for b in content:
        hist[b] += 1
    seen = set()
    for of_zero in (0, 1):
        for ml_full in (0, 1):
            for ll_full in (0, 1):
                did = (7, 300, 70000)[(of_zero + ml_full + ll_full) % 3]
                ofw = [1] * 20 if not of_zero else [1, 0, 1, 1, 0] + [2] * 12
                mlw = [3 if i < 20 else 1 for i in range(53 if ml_full else 40)]
                llw = [4 if i < 10 else 1 for i in range(36 if ll_full else 30)]
                d = du.build(content, did, hist, du.normalise(ofw, 7), 7, du.normalise(mlw, 8), 8, du.normalise(llw, 8), 8)

                def srcs(cut):
                    out = [b",".join(recs[200 + i * 9:200 + i * 9 + k]) for i, k in enumerate([1, 2, 3, 5, 8, 13, 20, 30, 40, 60])]
                    out += [bytes(r.choice(content) for _ in range(n)) for n in (50, 200, 800, 1500, 5000)] + [content[100:3000], content[5:7000]]
                    return [x for x in out if len(x) <= cut]

                for level in (1, 3):
                    info = check(emu, oracle_ref, d, level, srcs)
                    assert info["dictID"] == did and info["hufRepeat"] == 1
                    seen.add((info["llRepeat"], info["ofRepeat"], info["mlRepeat"]))
    assert (1, 1, 1) in seen and (2, 2, 2) in seen


def test_emu_cdict_checksum_and_rejects(emu, oracle_ref):
    recs = json_records(400, seed=2)
    d = b",".join(recs[:100])
    cd = oracle_ref.CDict(d, 3)
    ecd = EmuCDict(emu, d, 3)
    for x in (b"", recs[200], b",".join(recs[150:180])):
        assert ecd.compress(x, checksum=True) == cd.compress(x, checksum=True)
    with pytest.raises(ValueError):
        EmuCDict(emu, b"\x37\xa4\x30\xec" + bytes(40), 3)        # magic + garbage: dictionary_corrupted, like ZSTD_createCDict -> NULL
    with pytest.raises(oracle_ref.ZstdRefError):
        oracle_ref.CDict(b"\x37\xa4\x30\xec" + bytes(40), 3)


def test_emu_cdict_without_dict_id(emu, oracle_ref):
    """ZstdCompressCtx.setDictID(false) (ZSTD_c_dictIDFlag = 0): the dictionary's ID stays out of the frame header, the rest of the
    frame is unchanged; 1-, 2- and 4-byte IDs"""
    r = random.Random(77)
    recs = json_records(20000, seed=3)
    samples = [b",".join(recs[i * 13:i * 13 + 200])[:4096] for i in range(1000)]
    d = oracle_ref.train_dict(samples, 30000)
    assert oracle_ref.dict_id(d) > 0
    for level in (1, 3):
        rc = oracle_ref.CDict(d, level); ec = EmuCDict(emu, d, level)
        for src in sources(recs, r, 8192)[:40]:
            for ck in (False, True):
                want = rc.compress(src, ck, dict_id=False)
                assert ec.compress(src, ck, dict_id=False) == want, (level, len(src), ck)
                if src:
                    assert want != rc.compress(src, ck)
        rc.close(); ec.close()


def test_dictionary_with_a_huffman_table_12_bits_deep(emu, oracle_ref):
    """A dictionary whose literals table is 12 bits deep (HUF_TABLELOG_MAX: HUF_readCTable, N/compress/huf_compress.c:292-345, and ZSTD_loadDEntropy's
    HUF_readDTableX2_wksp, N/decompress/zstd_decompress.c:1473, both take it; the trainer never writes one) was refused at load in rounds 1-5.  Compress side: the
    same frames as ZSTD_CCtx_refCDict + ZSTD_compress2, treeless literals coded with the dictionary's 12-bit codes included.  Decompress side: those frames on both
    pipelines, and damaged ones answered as the reference's portable build answers them."""
    from util import emu_decompress_dict
    r = random.Random(12)
    alphabet = b"etaoinshrdlu."                                     # 13 symbols, counts 1, 1, 2, 4 ... 2048: a code 12 bits deep
    hist = [0] * 256
    for k, ch in enumerate(reversed(alphabet)):
        hist[ch] = 1 if k == 0 else 1 << (k - 1)
    pool = bytes(ch for ch in alphabet for _ in range(hist[ch]))

    def draw(n):
        return bytes(r.choice(pool) for _ in range(n))
    content = draw(3000) + alphabet * 3
    d = du.build(content, 4242, hist, du.normalise([1] * 20, 7), 7, du.normalise([3 if i < 20 else 1 for i in range(53)], 8), 8, du.normalise([4 if i < 10 else 1 for i in range(36)], 8), 8,
                 huf_max_bits=12)
    import ctypes as C
    R = oracle_ref.lib()
    R.HUF_readStats.restype = C.c_size_t
    R.HUF_readStats.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_char_p, C.c_size_t]
    w = C.create_string_buffer(256); rank = (C.c_uint * 16)(); nb = C.c_uint(0); tl = C.c_uint(0)
    assert not R.ZSTD_isError(R.HUF_readStats(w, 256, rank, C.byref(nb), C.byref(tl), d[8:], len(d) - 8)) and tl.value == 12
    treeless = 0
    for level in (1, 3):
        cd = oracle_ref.CDict(d, level); ecd = EmuCDict(emu, d, level)
        assert ecd.info()["hufRepeat"] == 1
        for n in (40, 90, 200, 500, 1200, 3000, 7000):
            for x in (draw(n), draw(n // 2) + content[100:100 + n // 2]):
                want = cd.compress(x)
                assert ecd.compress(x) == want, (level, n)
                hdr = 4 + 1 + 2 + (1 if len(x) < 256 else 2)        # magic, descriptor, dictID 4242 in 2 bytes, content size
                treeless += (want[hdr] >> 1) & 3 == 2 and want[hdr + 3] & 3 == 3
                for split in (False, True):
                    assert emu_decompress_dict(emu, want, len(x), d, split=split) == x, (level, n, split)
                for _ in range(25):
                    zb = bytearray(want); zb[r.randrange(hdr, len(zb))] ^= 1 << r.randrange(8); zb = bytes(zb)
                    try: p = oracle_ref.decompress_portable(zb, len(x), d)
                    except oracle_ref.ZstdRefError as e: p = -e.code
                    for split in (False, True):
                        assert emu_decompress_dict(emu, zb, len(x), d, split=split) == p, (level, n, split, zb.hex())
        cd.close(); ecd.close()
    assert treeless >= 8, treeless                                  # literals coded with the dictionary's own 12-bit table
