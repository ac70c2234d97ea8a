"""CPU (-m "not gpu"): the decoder kernel BODY (zstd-jni_amd/csrc/zj_decode.h), built lane-serial
(W = 1, tests/emu), against the reference's golden frames and reference-compressed edge cases.
This checks the format logic of the HIP code before it reaches a GPU; the -m gpu tests run the same
checks through the C-ABI on the wave64 build."""
import hashlib

import pytest

from conftest import golden, XML_SHA256_PREFIX
from util import edge_inputs, emu_lib, emu_decompress, emu_decompress_split, emu_decompress_dict, json_records


@pytest.fixture(scope="module")
def emu():
    return emu_lib()


@pytest.mark.parametrize("name", ["xml-1.zst", "xml-3.zst", "xml-6.zst", "xml-9.zst", "xml-advanced.zst", "xml-1-sized.zst"])
def test_emu_golden_xml(emu, name):
    out = emu_decompress(emu, golden(name), 6_000_000)
    assert not isinstance(out, int), out
    assert len(out) == 5_345_280 and hashlib.sha256(out).hexdigest().startswith(XML_SHA256_PREFIX)


def test_emu_multiframe(emu):
    out = emu_decompress(emu, golden("xml-sized-combined.zst"), 6_000_000)
    assert out[:102] == golden("xmlsmall")
    assert hashlib.sha256(out[102:]).hexdigest().startswith(XML_SHA256_PREFIX)
    # T/scala/Zstd.scala's doubled streams: xml-1x2.zst / xml-1-sizedx2.zst are the frame twice (conftest.golden puts them together)
    for name in ("xml-1x2.zst", "xml-1-sizedx2.zst"):
        out = emu_decompress(emu, golden(name), 11_000_000)
        assert len(out) == 2 * 5_345_280 and out[:5_345_280] == out[5_345_280:] and hashlib.sha256(out[:5_345_280]).hexdigest().startswith(XML_SHA256_PREFIX), name


@pytest.mark.parametrize("level", [1, 3])
def test_emu_edge_inputs(emu, oracle_ref, level):
    for name, data in edge_inputs():
        z = oracle_ref.compress(data, level)
        out = emu_decompress(emu, z, len(data))
        assert out == data, (name, out if isinstance(out, int) else "bytes differ")


def test_emu_synthetic_classes(emu, oracle_ref, zj):
    for size in (4096, 65536, 131072):
        raw = zj.synth_host(size, 0, 8)
        for i in range(8):
            data = raw[i * size:(i + 1) * size]
            for level in (1, 3):
                assert emu_decompress(emu, oracle_ref.compress(data, level), size) == data, (size, i, level)


def test_emu_errors(emu, oracle_ref):
    data = b"hello hello hello hello " * 100
    z = oracle_ref.compress(data, 3)
    assert emu_decompress(emu, z, len(data) - 1) == -70
    # a header (sized from its descriptor byte) plus one block header must be present before the magic number is examined
    # (zstd_decompress.c:966-979): short garbage is srcSize_wrong, long garbage prefix_unknown, garbage after a frame srcSize_wrong
    for junk, code in ((b"\x00\x01\x02\x03\x04\x05", -72), (bytes(range(20)), -10), (z + bytes(range(20)), -72)):
        assert emu_decompress(emu, junk, len(data)) == code
        with pytest.raises(oracle_ref.ZstdRefError, match="Src size is incorrect" if code == -72 else "Unknown frame descriptor"):
            oracle_ref.decompress(junk, len(data))
    assert isinstance(emu_decompress(emu, z[:-3], len(data)), int)
    bad = bytearray(z); bad[len(z) // 2] ^= 0x55
    out = emu_decompress(emu, bytes(bad), len(data))
    assert isinstance(out, int) or out != data or True      # must not crash; corruption may go unnoticed without checksum


def _mb(emu, frame, cap):
    import ctypes as C
    emu.emu_decompress_mb.restype = C.c_ulonglong
    emu.emu_decompress_mb.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_ulonglong, C.POINTER(C.c_int)]
    dst = C.create_string_buffer(max(cap, 1)); used = C.c_int(0)
    r = emu.emu_decompress_mb(frame, len(frame), dst, cap, C.byref(used))
    return dst.raw[:r] if r < (1 << 63) else -((1 << 64) - r)


def _ref_answers(oracle_ref, frame, cap):
    out = []
    for f in (oracle_ref.decompress, oracle_ref.decompress_portable):
        try:
            out.append(f(frame, cap))
        except oracle_ref.ZstdRefError as ex:
            out.append(-ex.code)
    return out


def test_compressed_block_of_exactly_128KiB(emu, oracle_ref):
    """A block header saying *compressed, 131 072 bytes* is entered, as the reference's 1.5.7 enters it (N/decompress/zstd_decompress_block.c:2073-2081: the
    specification allows a compressed block of exactly blockSizeMax; libzstd's encoder never writes one).  Rounds 1-5 refused it in the block loop (the rule of the
    decoders before 1.5.4) — found by tools/fuzz_emu_decode_mb.py seed 82.  Damaged forms (a raw 128 KiB block whose type bit was flipped, all four literal types:
    the reference answers dictionary_corrupted when the first byte claims treeless literals) and the VALID frame of this kind (raw literals filling the block, no
    sequences) on all three pipelines against both reference builds."""
    import random
    rnd = random.Random(5)
    body = bytearray(rnd.getrandbits(8) for _ in range(131072))
    tail = b"\x01\x00\x00"                                                  # an empty raw last block
    for lit_type in range(4):
        body[0] = (body[0] & 0xFC) | lit_type
        if lit_type < 2: body[0] |= 0x0C                                     # raw / RLE literals: a 3-byte size field, so that the section's size is the random 20 bits that follow
        frame = b"\x28\xb5\x2f\xfd" + b"\x00\x58" + (131072 << 3 | 2 << 1).to_bytes(3, "little") + bytes(body) + tail      # window descriptor 0x58: 2 MiB, no content size
        theirs = _ref_answers(oracle_ref, frame, 1 << 18)
        assert theirs[0] == theirs[1] == (-30 if lit_type == 3 else -20), (lit_type, theirs)
        assert emu_decompress(emu, frame, 1 << 18) == theirs[1], lit_type
        assert emu_decompress_split(emu, frame, 1 << 18)[0] == theirs[1], lit_type
        assert _mb(emu, frame, 1 << 18) == theirs[1], lit_type
    n_lit = 131072 - 3 - 1
    lits = bytes(rnd.getrandbits(8) for _ in range(n_lit))
    from oracle import port
    for last, extra in ((1, b""), (0, tail)):                                # as the frame's only block, and followed by another one
        for sized in (False, True):                                          # without / with a content size (the latter: the split pipeline's "simple" frame when it is the only block)
            head = b"\x28\xb5\x2f\xfd" + (b"\x80\x58" + n_lit.to_bytes(4, "little") if sized else b"\x00\x58")
            valid = head + (131072 << 3 | 2 << 1 | last).to_bytes(3, "little") + bytes([0 | 3 << 2 | (n_lit & 0xF) << 4, (n_lit >> 4) & 0xFF, n_lit >> 12]) + lits + b"\x00" + extra
            assert oracle_ref.decompress(valid, 1 << 18) == lits and oracle_ref.decompress_portable(valid, 1 << 18) == lits and port.decompress(valid, 1 << 18) == lits
            assert emu_decompress(emu, valid, 1 << 18) == lits and emu_decompress_split(emu, valid, 1 << 18)[0] == lits and _mb(emu, valid, 1 << 18) == lits, (last, sized)
            assert emu_decompress(emu, valid, n_lit - 1) == _ref_answers(oracle_ref, valid, n_lit - 1)[1]


def test_huffman_tables_12_bits_deep(emu, oracle_ref):
    """Literals coded with a Huffman table 12 bits deep — HUF_TABLELOG_MAX (N/common/huf.h:37), what the reference's decoder takes (ZSTD_HUFFDTABLE_CAPACITY_LOG,
    N/decompress/zstd_decompress_internal.h:78; HUF_readDTableX1_wksp / X2, N/decompress/huf_decompress.c:385-518, :1179-1263) although its own encoder stops at 11
    (LitHufLog) — were refused in rounds 1-5.  Sections made by the reference's own HUF_compress{1,4}X_repeat at tableLog 12 and hand-built ones (one stream, every
    count of weight-1 symbols: the slots the 2 048-cell form of the table keeps two to a cell, zd_huf_fill), as a frame's only block (with and without a content size:
    the three-stage pipeline and the block stages), followed by a treeless block that reuses the table, and damaged — every pipeline against both reference builds."""
    import random
    from util import deep_huffman_literals, literals_only_block, frame_of_blocks, hand_huffman_section
    rnd = random.Random(12)
    deep = 0
    for seed in range(8):
        n = rnd.choice([20000, 70000, 131000])
        lits, sec, depth = deep_huffman_literals(oracle_ref, n, seed)
        deep += depth == 12
        blk = literals_only_block(n, sec[4], 4)
        for frame in (frame_of_blocks([blk]), frame_of_blocks([blk], content_size=n)):
            assert _ref_answers(oracle_ref, frame, n) == [lits, lits]
            assert emu_decompress(emu, frame, n) == lits, (seed, depth)
            assert emu_decompress_split(emu, frame, n)[0] == lits, (seed, depth)
            assert _mb(emu, frame, n) == lits, (seed, depth)
        # ... a second block whose treeless literals use the first block's table: the same literals' streams without the tree description
        tree = sec[4][0] + 1 if sec[4][0] < 128 else 1 + (sec[4][0] - 127 + 1) // 2
        two = frame_of_blocks([literals_only_block(n, sec[4], 4, last=False), literals_only_block(n, sec[4][tree:], 4, treeless=True)])
        assert _ref_answers(oracle_ref, two, 2 * n) == [lits * 2, lits * 2]
        assert emu_decompress(emu, two, 2 * n) == lits * 2 and _mb(emu, two, 2 * n) == lits * 2, seed
        for _ in range(60):                                              # damaged: the same answer as the portable build — bytes or code
            zb = bytearray(frame_of_blocks([blk], content_size=n) if rnd.random() < 0.5 else two)
            for _ in range(rnd.randrange(1, 3)):
                at = rnd.randrange(6, min(len(zb), 400)) if rnd.random() < 0.6 else rnd.randrange(6, len(zb))
                zb[at] ^= 1 << rnd.randrange(8)
            zb = bytes(zb)
            want = _ref_answers(oracle_ref, zb, 2 * n)[1]
            assert emu_decompress(emu, zb, 2 * n) == want and emu_decompress_split(emu, zb, 2 * n)[0] == want and _mb(emu, zb, 2 * n) == want, (seed, zb[:40].hex())
    assert deep >= 6
    for w1 in (2, 4, 6, 30, 64, 116):                                    # hand-built: w1 symbols of 12 bits, one stream and four
        for n, streams in ((rnd.randrange(200, 1000), 1), (rnd.randrange(3000, 9000), 4)):
            lits, section = hand_huffman_section(rnd, n, w1, streams)
            frame = frame_of_blocks([literals_only_block(n, section, streams)], content_size=n)
            assert _ref_answers(oracle_ref, frame, n) == [lits, lits], (w1, streams)
            assert emu_decompress(emu, frame, n) == lits and emu_decompress_split(emu, frame, n)[0] == lits and _mb(emu, frame, n) == lits, (w1, streams)
            for _ in range(40):
                zb = bytearray(frame); zb[rnd.randrange(14, len(zb))] ^= 1 << rnd.randrange(8); zb = bytes(zb)
                want = _ref_answers(oracle_ref, zb, n)[1]
                assert emu_decompress(emu, zb, n) == want and emu_decompress_split(emu, zb, n)[0] == want, (w1, streams, zb.hex())


def test_emu_split_pipeline(emu, oracle_ref, zj):
    """prep -> lane-per-frame tANS decode -> execute == fused decoder == reference, and the frames the batched
    compressor emits (one block, content <= 64 KiB) really take the three-stage path"""
    import random
    rnd = random.Random(31)
    took = 0
    for name, data in edge_inputs():
        for level in (1, 3):
            z = oracle_ref.compress(data, level)
            out, used = emu_decompress_split(emu, z, len(data))
            assert out == data, (name, level, out if isinstance(out, int) else "bytes differ")
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    for _ in range(120):
        size = rnd.choice([rnd.randrange(1, 400), rnd.randrange(1, 5000), rnd.randrange(1, 65537), 65536, 4096, rnd.randrange(65537, 140000)])
        if rnd.random() < 0.5:
            off = rnd.randrange(0, len(xml) - size); data = xml[off:off + size]
        else:
            data = zj.synth_host(size, rnd.randrange(0, 100000), 1)
        for level in (1, 3):
            z = oracle_ref.compress(data, level)
            out, used = emu_decompress_split(emu, z, len(data))
            assert out == data, (size, level, out if isinstance(out, int) else "bytes differ")
            took += used in (1, 3)
            if 65536 < size <= 131072 and level == 1 and len(z) < size:
                assert used, size                                         # single-block frames up to 128 KiB take the three stages
            out, used = emu_decompress_split(emu, z, len(data) + 77)      # roomy destination
            assert out == data
    assert took > 100
    # errors come out of the fused path, so codes match it
    data = b"hello hello hello hello " * 100
    z = oracle_ref.compress(data, 3)
    assert emu_decompress_split(emu, z, len(data) - 1)[0] == -70
    for cut in (1, 3, 7):
        assert emu_decompress_split(emu, z[:-cut], len(data))[0] == emu_decompress(emu, z[:-cut], len(data))
    for pos in range(6, len(z)):
        bad = bytearray(z); bad[pos] ^= 0x41
        a = emu_decompress_split(emu, bytes(bad), len(data))[0]; b = emu_decompress(emu, bytes(bad), len(data))
        assert a == b, pos


def test_emu_dictionary_decode(emu, oracle_ref, zj):
    """ZstdDictDecompress / ZSTD_decompress_usingDDict (T/scala/ZstdDict.scala:58-216): frames the reference compressed
    with a trained dictionary and with a raw-content dictionary decode to the original; wrong / missing / corrupted
    dictionaries give the reference's error codes"""
    import random
    rnd = random.Random(5)
    trained = oracle_ref.train_dict(json_records(2000), 16384)
    assert trained[:4] == bytes.fromhex("37a430ec")
    raw = b"".join(json_records(40, seed=9, first=7000))                    # content-only dictionary (no magic)
    for d in (trained, raw):
        for _ in range(40):
            k = rnd.choice([1, 1, 2, 5, 30, 300])
            data = b"".join(json_records(k, seed=rnd.randrange(1000), first=rnd.randrange(100000)))
            if rnd.random() < 0.2:
                data = zj.synth_host(rnd.randrange(1, 70000), rnd.randrange(1000), 1)
            for level in (1, 3, 5, 9):
                z = oracle_ref.compress_using_dict(data, d, level)
                assert oracle_ref.decompress_using_dict(z, d, len(data)) == data
                out = emu_decompress_dict(emu, z, len(data), d)
                assert out == data, (len(data), level, out if isinstance(out, int) else "bytes differ")
                out = emu_decompress_dict(emu, z, len(data), d, split=True)       # three-stage pipeline with the dictionary
                assert out == data, (len(data), level, "split", out if isinstance(out, int) else "bytes differ")
        # a buffer of two frames, both using the dictionary
        a, b = b"".join(json_records(3, first=11)), b"".join(json_records(4, first=99))
        z = oracle_ref.compress_using_dict(a, d, 3) + oracle_ref.compress_using_dict(b, d, 3)
        assert emu_decompress_dict(emu, z, len(a) + len(b), d) == a + b
    data = b"".join(json_records(5, first=3))
    z = oracle_ref.compress_using_dict(data, trained, 3)
    assert emu_decompress(emu, z, len(data)) == -32                          # frame names a dictionary, none given
    other = oracle_ref.train_dict(json_records(2000, seed=4, first=50000), 8192)
    if oracle_ref.dict_id(other) != oracle_ref.dict_id(trained):
        assert emu_decompress_dict(emu, z, len(data), other) == -32          # dictionary_wrong
        assert emu_decompress_dict(emu, z, len(data), other, split=True) == -32
    broken = bytearray(trained); broken[9] ^= 0xFF; broken[10] ^= 0xFF; broken[12] ^= 0xFF
    r = emu_decompress_dict(emu, z, len(data), bytes(broken))
    try:
        oracle_ref.decompress_using_dict(z, bytes(broken), len(data)); ref_ok = True
    except Exception:                                                        # noqa: BLE001
        ref_ok = False
    assert ref_ok or r == -30
    # frames made without a dictionary still decode when one is loaded
    plain = oracle_ref.compress(data, 3)
    assert emu_decompress_dict(emu, plain, len(data), trained) == data


def test_emu_streamed_frames_without_content_size(emu, oracle_ref, oracle_port):
    """what ZstdOutputStream writes (reference N/jni_outputstream_zstd.c: ZSTD_compressStream2, no pledged size): no content
    size in the header, blocks closed early by flushes, optional checksum; decoded with the caller's size as capacity, with a
    larger one, and refused with dstSize_tooSmall when one byte short — by both pipelines and the C restatement"""
    data = b",".join(json_records(12000, seed=21))[:400_000]
    for level, checksum, flush_every in ((1, False, 0), (3, True, 1), (3, False, 3), (9, True, 0)):
        z = oracle_ref.compress_stream(data, level, checksum, chunk=30000, flush_every=flush_every)
        assert oracle_ref.lib().ZSTD_getFrameContentSize(z, len(z)) == (1 << 64) - 1           # ZSTD_CONTENTSIZE_UNKNOWN
        assert emu_decompress(emu, z, len(data)) == data
        assert emu_decompress_split(emu, z, len(data))[0] == data
        assert emu_decompress(emu, z, len(data) + 1000) == data
        assert emu_decompress(emu, z, len(data) - 1) == -70
        assert oracle_port.decompress(z, len(data)) == data


def test_ncount_reader_matches_reference_on_random_buffers(emu, oracle_ref, oracle_port):
    """FSE_readNCount (N/common/entropy_common.c:42-188) against the kernels' reader and the C restatement on random and
    truncated table descriptions — including descriptions that run past their buffer, where the reference wraps around on the
    last four bytes instead of failing (:146-153) and the answer (size, counts, or refusal) has to be the same"""
    import ctypes as C
    import random
    R, P = oracle_ref.lib(), oracle_port.lib()
    R.FSE_readNCount.restype = C.c_size_t
    R.FSE_readNCount.argtypes = [C.POINTER(C.c_short), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_char_p, C.c_size_t]
    P.zso_read_ncount.restype = C.c_size_t
    P.zso_read_ncount.argtypes = [C.POINTER(C.c_short), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_char_p, C.c_size_t]
    emu.emu_read_ncount.restype = C.c_uint
    emu.emu_read_ncount.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.POINTER(C.c_short), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    rnd = random.Random(11)
    # real descriptions to mutate: sequence sections of reference-compressed frames hold them; plain random bytes parse too
    seeds = [oracle_ref.compress(bytes(min(255, int(rnd.expovariate(0.03))) for _ in range(4000)), 3)[-400:] for _ in range(8)]
    ok = 0
    for it in range(60000):
        if rnd.random() < 0.5:
            s = rnd.choice(seeds); a = rnd.randrange(0, len(s) - 2); src = bytearray(s[a:a + rnd.randrange(1, 40)])
        else:
            src = bytearray(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 24)))
        if rnd.random() < 0.5:
            src[0] = (src[0] & 0xF0) | rnd.randrange(0, 5)                      # small table logs parse more often
        if rnd.random() < 0.3 and len(src) > 4:
            for k in range(rnd.randrange(1, 5)): src[-1 - k] = 0xFF               # long zero-runs near the end
        src = bytes(src); maxSV = rnd.choice([35, 31, 52, 255])
        n1 = (C.c_short * 256)(); m1 = C.c_uint(maxSV); t1 = C.c_uint(0)
        r = R.FSE_readNCount(n1, C.byref(m1), C.byref(t1), src, len(src))
        n2 = (C.c_short * 256)(); m2 = C.c_uint(maxSV); t2 = C.c_uint(0)
        p = P.zso_read_ncount(n2, C.byref(m2), C.byref(t2), src, len(src))
        n3 = (C.c_short * 256)(); m3 = C.c_uint(0); t3 = C.c_uint(0)
        e = emu.emu_read_ncount(src, len(src), maxSV, n3, C.byref(m3), C.byref(t3))
        if R.ZSTD_isError(r):
            assert p == r and e == 0, (src.hex(), maxSV, hex(r), hex(p), e)
        else:
            ok += 1
            assert p == r and e == r, (src.hex(), maxSV, r, p, e)
            assert m1.value == m2.value == m3.value and t1.value == t2.value == t3.value, src.hex()
            assert list(n1[:m1.value + 1]) == list(n2[:m1.value + 1]) == list(n3[:m1.value + 1]), src.hex()
    assert ok > 3000


def test_huffman_tree_description_reader_matches_reference(emu, oracle_ref):
    """HUF_readStats (N/common/entropy_common.c:243-305) against the kernels' reader on mutated and random tree descriptions:
    direct 4-bit weights and FSE-compressed ones — whose table description may name up to 92 symbols at tableLog 5 before the
    reference's workspace test refuses it (fse_decompress.c:273).  Same verdict, symbol count, depth and weights."""
    import ctypes as C
    import random
    from util import hand_huffman_section
    R = oracle_ref.lib()
    R.HUF_readStats.restype = C.c_size_t
    R.HUF_readStats.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_char_p, C.c_size_t]
    emu.emu_huf_weights.restype = C.c_uint
    emu.emu_huf_weights.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.POINTER(C.c_uint)]
    rnd = random.Random(5)
    seeds = []
    for _ in range(20):
        a = rnd.choice([4, 17, 60, 200])
        seeds.append(oracle_ref.compress(bytes(min(255, int(rnd.expovariate(1.0 / a))) for _ in range(3000)), 3)[9:149])
    ok = 0; deep = 0
    for it in range(40000):
        k = rnd.random()
        if k < 0.05:                                    # a description 12 bits deep (hand_huffman_section), whole or with a flipped bit
            sec = hand_huffman_section(rnd, 10, rnd.choice([2, 4, 10, 40, 116]), 1)[1]
            s = bytearray(sec[:1 + (sec[0] - 127 + 1) // 2])
            if rnd.random() < 0.5: s[rnd.randrange(len(s))] ^= 1 << rnd.randrange(8)
        elif k < 0.6:
            s = bytearray(rnd.choice(seeds)); a = rnd.randrange(0, 8); s = s[a:a + rnd.randrange(1, 140)]
            for _ in range(rnd.randrange(0, 3)):
                if s: s[rnd.randrange(len(s))] ^= 1 << rnd.randrange(8)
        elif k < 0.8:
            n = rnd.randrange(1, 128)
            s = bytearray([127 + n]) + bytearray((rnd.randrange(0, 4) << 4) | rnd.randrange(0, 4) for _ in range(rnd.randrange(0, (n + 1) // 2 + 3)))
        else:
            s = bytearray(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 40))); s[0] = rnd.randrange(1, min(len(s) + 3, 127))
        s = bytes(s)
        w = C.create_string_buffer(256); rank = (C.c_uint * 16)(); nb = C.c_uint(0); tl = C.c_uint(0)
        r = R.HUF_readStats(w, 256, rank, C.byref(nb), C.byref(tl), s, len(s))
        w2 = C.create_string_buffer(256); tl2 = C.c_uint(0)
        e = emu.emu_huf_weights(s, len(s), w2, C.byref(tl2))
        if R.ZSTD_isError(r):
            assert e == 0, s.hex()
        else:
            deep += tl.value == 12
            ok += 1
            assert e == nb.value and tl2.value == tl.value and w.raw[:nb.value] == w2.raw[:nb.value], s.hex()
    assert ok > 800 and deep > 20, (ok, deep)          # (tables 12 bits deep included: HUF_TABLELOG_MAX)


def test_emu_fse_table_by_the_wave():
    """zd_fse_spread + zd_fse_finish_wave (round 5: the decode tables' second pass on the whole wave, a slot's state from its rank among its symbol's slots) gives
    zd_build_fse's cells — random normalised distributions with low-probability (-1) symbols, every table log and kind, and the format's three default distributions"""
    import ctypes as C
    import random
    from util import emu_lib
    L = emu_lib()
    rnd = random.Random(41)
    cases = 0
    for _ in range(4000):
        kind = rnd.randrange(3); maxsv = rnd.randrange(1, (36, 32, 53)[kind]); log = rnd.randrange(5, (10, 9, 10)[kind]); size = 1 << log
        nlow = rnd.randrange(0, min(maxsv + 1, size // 2, 30) + 1) if rnd.random() < 0.7 else 0
        syms = list(range(maxsv + 1)); rnd.shuffle(syms)
        low = set(syms[:nlow]); rest = [s for s in syms if s not in low]
        norm = [0] * 64
        for s in low: norm[s] = -1
        left = size - nlow
        if not rest:
            continue
        k = rnd.randrange(1, len(rest) + 1); chosen = rest[:k]
        if left < k:
            continue
        cuts = sorted(rnd.sample(range(1, left), k - 1)) if k > 1 else []
        parts = [b - a for a, b in zip([0] + cuts, cuts + [left])]
        for s, c in zip(chosen, parts): norm[s] = c
        arr = (C.c_short * 64)(*norm); a = (C.c_uint * 512)(); b = (C.c_uint * 512)()
        r = L.emu_fse_dtable(arr, maxsv, log, kind, a, b)
        assert r == 3, (r, norm[:maxsv + 1], log)
        assert list(a)[:size] == list(b)[:size], (kind, log, norm[:maxsv + 1])
        cases += 1
    assert cases > 3000


def test_frames_of_one_stored_block_are_copied_by_stage_1(emu, oracle_ref):
    """zd_prep_frame_stored (round 6): a frame that is a header and ONE raw or RLE block — what the compressors write for data that does not compress — is copied by stage 1
    of the batch pipelines itself instead of going through the block stages; with and without checksum, hand-made RLE frames, empty content; every damaged or unusual form
    (wrong content size, trailing bytes, a block larger than the window, a dictionary ID, a bad checksum, short destinations) is left to the paths that answer as the
    reference's portable build does."""
    import random
    rnd = random.Random(9)
    taken = 0
    for n in (0, 1, 7, 300, 5000, 65536, 131072):
        data = bytes(rnd.getrandbits(8) for _ in range(n))
        for ck in (False, True):
            z = oracle_ref.compress(data, 3, ck)
            for cap in (n, n + 5, max(n - 1, 0)):
                want = _ref_answers(oracle_ref, z, cap)[1]
                out, used = emu_decompress_split(emu, z, cap)
                assert out == want and _mb(emu, z, cap) == want, (n, ck, cap)
                if cap >= n and n > 0 and (z[4 + 1 + (1 if n < 256 else 2 if n < 65792 else 4)] >> 1) & 3 == 0: taken += used == 4
            for _ in range(30):
                zb = bytearray(z)
                if rnd.random() < 0.5 and len(zb) > 5: zb[rnd.randrange(4, min(len(zb), 12))] ^= 1 << rnd.randrange(8)
                elif rnd.random() < 0.5: zb += bytes(rnd.randrange(1, 6))
                else: zb = zb[:max(5, len(zb) - rnd.randrange(1, 4))]
                zb = bytes(zb); want = _ref_answers(oracle_ref, zb, n + 8)[1]
                assert emu_decompress_split(emu, zb, n + 8)[0] == want and _mb(emu, zb, n + 8) == want, (n, ck, zb[:16].hex())
    assert taken >= 8, taken
    for n in (1, 200, 70000, 131072):                                          # RLE block as the only block: hand-made (libzstd never makes the first block RLE)
        for hdr in (b"\x20" + bytes([n]) if n < 256 else (b"\x60" + (n - 256).to_bytes(2, "little") if n < 65792 else b"\xa0" + n.to_bytes(4, "little")),):
            z = b"\x28\xb5\x2f\xfd" + hdr + (n << 3 | 1 << 1 | 1).to_bytes(3, "little") + b"\x5a"
            want = _ref_answers(oracle_ref, z, n)
            assert want[0] == want[1] == b"\x5a" * n
            out, used = emu_decompress_split(emu, z, n)
            assert out == want[1] and used == 4 and _mb(emu, z, n) == want[1], n
            assert emu_decompress_split(emu, z, n - 1)[0] == _ref_answers(oracle_ref, z, n - 1)[1]


def test_four_huffman_streams_by_the_whole_wave(emu, oracle_ref):
    """zd_huf_streams_wave (round 6): a literals section's four streams cut into 16 spans each, every span decoded by a lane from a GUESSED first code, the guesses checked
    by a counting pass (a lane whose predecessor ended elsewhere decodes again), then written.  Code tables of every temper — one code length (never falls into step:
    the guess sits on the lengths' lattice), geometric (at once), near-uniform with a rare long code (slowly: repeated passes, or the section is left to the one-lane
    rounds) — decode to the reference's bytes on both batch pipelines; the counters say the wave took sections, repeated passes, and gave some up.  Damaged streams
    (bits flipped inside the section) answer what the reference's portable decoder answers: the wave decides nothing about a stream that does not check out."""
    import ctypes as C
    import random
    from util import skewed_literal_inputs
    rnd = random.Random(11)
    emu.emu_hp_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    st = (C.c_ulonglong * 5)()
    emu.emu_hp_stats(st, 1)
    frames = 0
    for d in skewed_literal_inputs():
        for level in (1, 3):
            z = oracle_ref.compress(d, level, level == 1)
            assert emu_decompress_split(emu, z, len(d))[0] == d and _mb(emu, z, len(d)) == d, (len(d), level, len(set(d)))
            frames += 1
            if len(d) <= 20000:
                for _ in range(6):                                   # damage somewhere behind the headers: mostly inside the Huffman streams
                    zb = bytearray(z); p = rnd.randrange(12, len(zb)); zb[p] ^= 1 << rnd.randrange(8); zb = bytes(zb)
                    want = _ref_answers(oracle_ref, zb, len(d))[1]
                    assert emu_decompress_split(emu, zb, len(d))[0] == want and _mb(emu, zb, len(d)) == want, (len(d), level, p)
    emu.emu_hp_stats(st, 0)
    taken, at_check, after_tries, repeats, lanes_again = list(st)
    assert taken > frames and repeats > 0 and at_check > 0, list(st)       # (sections taken on both pipelines; some guesses were wrong; damaged streams were left to the rounds)
