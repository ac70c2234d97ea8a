"""Shared test helpers: seeded inputs covering the block/literal/sequence modes of the format."""
import ctypes as C
import os
import random
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def emu_lib():
    """tests/emu/libzjni_emu.so — lane-serial build of the kernel bodies (test infrastructure)."""
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", d])
    L = C.CDLL(os.environ.get("ZJNI_EMU_LIB") or os.path.join(d, "libzjni_emu.so"))
    L.emu_decompress.restype = C.c_ulonglong
    L.emu_decompress.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint]
    L.emu_decompress_split.restype = C.c_ulonglong
    L.emu_decompress_split.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.POINTER(C.c_int)]
    L.emu_decompress_split_dict.restype = C.c_ulonglong
    L.emu_decompress_split_dict.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.POINTER(C.c_int)]
    L.emu_decompress_dict.restype = C.c_ulonglong
    L.emu_decompress_dict.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_char_p, C.c_uint]
    for fn in ("emu_compress", "emu_compress_split", "emu_compress_multi", "emu_compress_chain"):
        if hasattr(L, fn):
            getattr(L, fn).restype = C.c_ulonglong
            getattr(L, fn).argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_uint]
    L.emu_cdict_create.restype = C.c_void_p
    L.emu_cdict_create.argtypes = [C.c_char_p, C.c_uint, C.c_uint]
    L.emu_cdict_free.argtypes = [C.c_void_p]
    L.emu_cdict_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    L.emu_compress_cdict.restype = C.c_ulonglong
    L.emu_compress_cdict.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_uint]
    return L


class EmuCDict:
    """lane-serial build of the dictionary digest + attach-mode compress (ZstdDictCompress + ZstdCompressCtx.loadDict)"""

    def __init__(self, L, dictionary, level):
        self.L = L
        self.ptr = L.emu_cdict_create(dictionary, len(dictionary), level)
        if not self.ptr:
            raise ValueError("dictionary rejected")

    def info(self):
        out = (C.c_uint * 12)()
        self.L.emu_cdict_info(self.ptr, out)
        keys = ["dictID", "contentSize", "windowLog", "chainLog", "hashLog", "minMatch", "strategy", "hufRepeat", "llRepeat", "ofRepeat", "mlRepeat", "fillStart"]
        return dict(zip(keys, list(out)))

    def compress(self, data, checksum=False, dict_id=True, cap=None):
        if cap is None:
            cap = len(data) + (len(data) >> 8) + 64 + 128
        dst = C.create_string_buffer(max(cap, 1))
        r = self.L.emu_compress_cdict(self.ptr, data, len(data), dst, cap, int(checksum) | (0 if dict_id else 4))
        if r >= (1 << 63):
            return -((1 << 64) - r)
        return dst.raw[:r]

    def close(self):
        if self.ptr:
            self.L.emu_cdict_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.close()


def emu_decompress(L, frame, cap):
    dst = C.create_string_buffer(max(cap, 1))
    r = L.emu_decompress(frame, len(frame), dst, cap)
    if r >= (1 << 63):
        return -((1 << 64) - r)
    return dst.raw[:r]


def emu_decompress_dict(L, frame, cap, dictionary, split=False):
    dst = C.create_string_buffer(max(cap, 1))
    if split:
        used = C.c_int(0)
        r = L.emu_decompress_split_dict(frame, len(frame), dst, cap, dictionary, len(dictionary), C.byref(used))
    else:
        r = L.emu_decompress_dict(frame, len(frame), dst, cap, dictionary, len(dictionary))
    if r >= (1 << 63):
        return -((1 << 64) - r)
    return dst.raw[:r]


def json_records(n, seed=1, first=0):
    """small JSON-like records (the shape of BASELINE config 4) for dictionary tests"""
    rnd = random.Random(seed)
    names = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel"]
    out = []
    for i in range(first, first + n):
        out.append(('{"id":%d,"name":"%s","tags":["%s","%s"],"score":%d,"active":%s,"note":"record number %d of the set"}'
                    % (i, names[i % 8], names[(i * 3) % 8], names[(i * 5) % 8], rnd.randrange(1000), "true" if i % 2 else "false", i)).encode())
    return out


def emu_decompress_split(L, frame, cap):
    """three-stage decode pipeline (prep -> lane sequence decode -> execute); returns (bytes | -code, which path answered)"""
    dst = C.create_string_buffer(max(cap, 1))
    used = C.c_int(0)
    r = L.emu_decompress_split(frame, len(frame), dst, cap, C.byref(used))
    if r >= (1 << 63):
        return -((1 << 64) - r), used.value
    return dst.raw[:r], used.value                      # 0: the fused decoder answered; 1 or 3: the three stages (3: literals from stage 2b); 4: stage 1 copied a frame of one stored block


def emu_compress_multi(L, data, level, checksum=False, content_size=True, serial=False, hash_log=0, chain_log=0, pipelined=False, cap=None):
    """multi-block frames (ze_compress_multi, lane-serial): frame bytes or -code.  serial: level-3 blocks on the one-lane parse
    (ZE_FLAG_MULTI_SERIAL) instead of the wave matcher (zj_match_wavex.h, 64 emulated lanes); serial=2: the wave matchers without staged spans; serial=4: levels 1-2 on the one-lane parse, level 3 on the wave matcher"""
    if cap is None: cap = len(data) + (len(data) >> 8) + 64 + 128
    dst = C.create_string_buffer(max(cap, 1))
    r = L.emu_compress_multi(data, len(data), dst, cap, level | (0x100 if checksum else 0) | (0 if content_size else 0x200) | (0x800 if serial is True else 0x1000 if serial == 2 else 0x2000 if serial == 4 else 0) | (0x8000 if pipelined else 0) | (hash_log << 16) | (chain_log << 24))
    if r >= (1 << 63):
        return -((1 << 64) - r)
    return dst.raw[:r]


def emu_compress_chain(L, data, level, checksum=False, content_size=True):
    """levels 4-8, frames <= 16 KiB, the large-batch route (chain parser per frame, then the entropy stage on its records)"""
    cap = len(data) + (len(data) >> 8) + 64 + 128
    dst = C.create_string_buffer(cap)
    r = L.emu_compress_chain(data, len(data), dst, cap, level | (0x100 if checksum else 0) | (0 if content_size else 0x200))
    if r >= (1 << 63):
        return -((1 << 64) - r)
    return dst.raw[:r]


def emu_compress(L, data, level, split=False, checksum=False, hash_log=0, chain_log=0, content_size=True):
    cap = len(data) + (len(data) >> 8) + 64 + 128
    dst = C.create_string_buffer(cap)
    assert split or not (hash_log or chain_log)           # explicit table sizes exist on the lane-per-frame path only
    r = (L.emu_compress_split if split else L.emu_compress)(data, len(data), dst, cap, level | (0x100 if checksum else 0) | (0 if content_size else 0x200) | (hash_log << 16) | (chain_log << 24))
    if r >= (1 << 63):
        return -((1 << 64) - r)
    return dst.raw[:r]


def emu_wave_libs():
    """the explicit-SIMT builds of the wave-per-frame matcher: lanes visited in ascending and in descending order"""
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", d])
    out = []
    names = ("libzjni_emu_wave.so", "libzjni_emu_wave_rev.so")
    if os.environ.get("ZJNI_EMU_WAVE_LIB"):          # the sanitizer build (tests/test_emu_sanitizer.py)
        names = (os.environ["ZJNI_EMU_WAVE_LIB"],)
    for name in names:
        L = C.CDLL(os.path.join(d, name))
        L.emu_compress_wave.restype = C.c_ulonglong
        L.emu_compress_wave.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_uint]
        out.append(L)
    return out


def emu_compress_wave(L, data, level=3, checksum=False):
    """wave-per-frame matcher (zj_match_wave.h, 64 emulated lanes) + entropy stage; frame bytes, -code, or None when the
    matcher does not take the frame (other level, > 64 KiB, < 64 B)"""
    cap = len(data) + (len(data) >> 8) + 64 + 128
    dst = C.create_string_buffer(cap)
    r = L.emu_compress_wave(data, len(data), dst, cap, level | (0x100 if checksum else 0))
    if r == (1 << 64) - 1:
        return None
    if r >= (1 << 63):
        return -((1 << 64) - r)
    return dst.raw[:r]


def equal_count_inputs(seed=7):
    """shuffled multisets: many literal values with exactly the same count (the Huffman sort's quicksort then peels one
    element per partition — the case that overflowed the kernels' explicit sort stack), incl. the counts around the
    sort's bucket boundary (164, 165, 166)"""
    rnd = random.Random(seed)
    out = []
    for a, c in ((98, 166), (99, 165), (40, 164), (12, 164), (130, 20), (256, 255), (256, 64), (9, 1000), (200, 300)):
        v = [x for x in range(a) for _ in range(c)]
        rnd.shuffle(v)
        out.append((f"{a}x{c}", bytes(v)))
    return out


def edge_inputs(seed=1234):
    """(name, bytes) inputs exercising: empty, tiny, RLE block, raw block, 1-stream and 4-stream
    literals, predefined / RLE / compressed FSE modes, long matches, long literal runs, max block."""
    rnd = random.Random(seed)
    words = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta", b"eta", b"theta"]
    out = [
        ("empty", b""),
        ("one", b"x"),
        ("two", b"ab"),
        ("rle_small", b"a" * 100),
        ("rle_block", b"\x00" * 65536),
        ("rle_max", b"z" * 131072),
        ("random_64k", bytes(rnd.getrandbits(8) for _ in range(65536))),
        ("random_300", bytes(rnd.getrandbits(8) for _ in range(300))),
        ("text_small", b" ".join(rnd.choice(words) for _ in range(40))),
        ("text_4k", b" ".join(rnd.choice(words) for _ in range(700))[:4096]),
        ("text_64k", b" ".join(rnd.choice(words) for _ in range(12000))[:65536]),
        ("text_128k", b" ".join(rnd.choice(words) for _ in range(24000))[:131072]),
        ("lowent_64k", bytes(rnd.choice(b"abcdefgh") for _ in range(65536))),
        ("lowent_1k", bytes(rnd.choice(b"abc") for _ in range(1000))),
        ("skewed_64k", bytes(min(255, int(rnd.expovariate(0.08))) for _ in range(65536))),
        ("long_match", (b"0123456789abcdef" * 64) + bytes(rnd.getrandbits(8) for _ in range(500)) + (b"0123456789abcdef" * 4000)),
        ("periodic_3", b"abc" * 20000),
        ("lit_run_then_match", bytes(rnd.getrandbits(8) for _ in range(40000)) + b"Q" * 30 + bytes(rnd.getrandbits(8) for _ in range(20000))),
        ("mixed_100k", (b" ".join(rnd.choice(words) for _ in range(5000)) + bytes(rnd.getrandbits(8) for _ in range(30000)) + bytes(rnd.choice(b"01") for _ in range(40000)))[:100000]),
        ("binary_struct", b"".join((i % 251).to_bytes(4, "little") + b"\x00\x00\x01\x00" + bytes([rnd.getrandbits(8) & 0x0F]) for i in range(7000))),
    ]
    return out


def crafted_far_offset_frame(of_code, extra, checksum=False, first_raw=100):
    """A hand-built frame WITHOUT checksum: a raw block, then a compressed block of 4 raw literals and ONE sequence whose three tables are in RLE mode — literal
    length code 3, match length code 0, offset code `of_code` with `extra` as its extra bits (N/decompress/zstd_decompress_block.c:1229-1300: offset =
    (1 << code) - 3 + extra for code >= 2).  Offset codes 28 .. 31 give offsets far beyond any output position (and from code 31 on with bit 31 set): the
    reference answers corruption_detected; ADVICE r04 found the lane-per-block decode taking such an offset for a symbolic repcode."""
    assert 2 <= of_code <= 31 and 0 <= extra < (1 << of_code)
    raw = bytes((i * 7 + 1) & 0xFF for i in range(first_raw))
    bits = extra | (1 << of_code)                        # the extra bits, the end mark above them
    stream = bits.to_bytes((of_code + 8) // 8, "little")
    body = bytes([4 << 3]) + b"wxyz" + bytes([1, 0x54, 3, of_code, 0]) + stream
    total = first_raw + 4 + 3                            # (the fourth literal follows the match)
    hdr = b"\x28\xb5\x2f\xfd" + bytes([0x20 | (4 if checksum else 0), total])
    blk1 = ((first_raw << 3) | 0).to_bytes(3, "little") + raw
    blk2 = ((len(body) << 3) | (2 << 1) | 1).to_bytes(3, "little") + body
    return hdr + blk1 + blk2, total


def deep_huffman_literals(ref, n, seed, depth=12):
    """`n` literal bytes whose Huffman code the reference's coder builds `depth` bits deep (a geometric distribution over ~40 symbols), the reference's
    HUF_compress{1,4}X_repeat output for them at tableLog = depth (tree description + streams) — what an encoder other than libzstd's (LitHufLog 11) may put into a
    literals section — and the table depth HUF_readStats reports for it.  Returns (literals, {1: section1, 4: section4}, depth_read)."""
    R = ref.lib()
    rnd = random.Random(seed)
    syms = list(range(256)); rnd.shuffle(syms)
    k = rnd.randrange(24, 60)
    lits = bytearray()
    while len(lits) < n:
        j = 0
        while j < k - 1 and rnd.random() < 0.5: j += 1
        lits.append(syms[j])
    lits = bytes(lits[:n])
    out = {}
    for streams, fn in ((1, R.HUF_compress1X_repeat), (4, R.HUF_compress4X_repeat)):
        fn.restype = C.c_size_t
        fn.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint, C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int), C.c_int]
        dst = C.create_string_buffer(n + 1024); wk = (C.c_ulonglong * 2048)(); ct = (C.c_size_t * 300)(); rep = C.c_int(0)
        r = fn(dst, len(dst), lits, n, 255, depth, wk, C.sizeof(wk), ct, C.byref(rep), 0)
        assert not R.ZSTD_isError(r) and r > 1, R.ZSTD_getErrorName(r)
        out[streams] = dst.raw[:r]
    R.HUF_readStats.restype = C.c_size_t
    R.HUF_readStats.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_char_p, C.c_size_t]
    w = C.create_string_buffer(256); rank = (C.c_uint * 16)(); nb = C.c_uint(0); tl = C.c_uint(0)
    assert not R.ZSTD_isError(R.HUF_readStats(w, 256, rank, C.byref(nb), C.byref(tl), out[4], len(out[4])))
    return lits, out, tl.value


def literals_only_block(lits_len, section, streams, last=True, treeless=False):
    """a compressed block holding one Huffman-coded literals section (5-byte header form for 4 streams, 3-byte for one) and no sequences"""
    c = len(section)
    if streams == 1:
        assert lits_len < 1024 and c < 1024
        hdr = ((3 if treeless else 2) | 0 << 2 | lits_len << 4 | c << 14).to_bytes(3, "little")
    else:
        hdr = ((3 if treeless else 2) | 3 << 2 | lits_len << 4 | c << 22).to_bytes(5, "little")
    body = hdr + section + b"\x00"
    return (len(body) << 3 | 2 << 1 | (1 if last else 0)).to_bytes(3, "little") + body


def frame_of_blocks(blocks, content_size=None, window_descriptor=0x58):
    """magic + a header with a window descriptor (2 MiB by default) and optionally an 8-byte content size + the blocks"""
    if content_size is None:
        return b"\x28\xb5\x2f\xfd" + bytes([0x00, window_descriptor]) + b"".join(blocks)
    return b"\x28\xb5\x2f\xfd" + bytes([0xC0, window_descriptor]) + content_size.to_bytes(8, "little") + b"".join(blocks)


def hand_huffman_section(rnd, n, w1, streams):
    """a Huffman-coded literals section written by hand (N/compress/huf_compress.c:248-290 direct weights, :991-1118 the streams): a complete prefix code 12 bits
    deep with exactly `w1` symbols of 12 bits (w1 even), symbols < 128 so that the 4-bit weight list describes it.  Returns (literals, section bytes)."""
    assert w1 % 2 == 0 and 2 <= w1 <= 116
    # code lengths: w1 leaves at depth 12; the w1/2 internal nodes above them are completed into a full tree by one leaf at each depth where the count is odd
    lengths = [12] * w1
    need = w1 // 2                                  # nodes at depth 11 that are parents of the 12-bit leaves
    depth = 11
    while depth >= 1:
        if need % 2 == 1 or (depth == 1 and need == 1):
            lengths.append(depth); need += 1
        need //= 2; depth -= 1
    assert need == 1 and len(lengths) <= 128 and sum(2.0 ** -l for l in lengths) == 1.0
    syms = sorted(rnd.sample(range(128), len(lengths)))
    rnd.shuffle(lengths)
    nb = dict(zip(syms, lengths))
    last = syms[-1]
    weights = [13 - nb[s] if s in nb else 0 for s in range(last)]                     # the last symbol's weight is implied
    hdr = bytes([127 + len(weights)]) + bytes(((weights[i] << 4) | (weights[i + 1] if i + 1 < len(weights) else 0)) for i in range(0, len(weights), 2))
    # canonical codes: slots by ascending weight, ascending symbol inside a weight; code = first slot >> (weight - 1)
    code = {}; slot = 0
    for w in range(1, 13):
        for s in syms:
            if 13 - nb[s] == w: code[s] = slot >> (w - 1); slot += 1 << (w - 1)
    assert slot == 4096
    pool = [s for s in syms for _ in range(max(1, 4096 >> nb[s] >> 4))]
    lits = bytes(rnd.choice(pool) if rnd.random() < 0.8 else rnd.choice(syms) for _ in range(n))

    def stream(part):
        acc = 0; pos = 0
        for b in reversed(part):
            acc |= code[b] << pos; pos += nb[b]
        acc |= 1 << pos
        return acc.to_bytes(pos // 8 + 1, "little")
    if streams == 1:
        return lits, hdr + stream(lits)
    seg = (n + 3) // 4
    parts = [stream(lits[i * seg:(i + 1) * seg] if i < 3 else lits[3 * seg:]) for i in range(4)]
    return lits, hdr + b"".join(len(p).to_bytes(2, "little") for p in parts[:3]) + b"".join(parts)


def needs_tuning_build(zj):
    """skip unless libzjni_amd.so is a TUNING build (tools/build_variant.sh ... -DZJ_TUNING_KERNELS: its stamp ends in "+tuning"): the experiment knobs and the losing
    routes they select (ZJNI_HYBRID, ZJNI_LANE_MACHINE, ZJNI_MULTI_WAVE, ZJNI_L4_LANES, ZJNI_PRECLEAR, ...) read as unset in the product library (zj_tune, zj_kernels.hip)"""
    import pytest
    if b"+tuning" not in zj.lib().zjni_build_stamp():
        pytest.skip("a switch of tuning builds only (-DZJ_TUNING_KERNELS)")


def skewed_literal_inputs(seed=7):
    """inputs whose frames carry LARGE Huffman-coded literal sections with code tables of every temper (tests of zd_huf_streams_wave, zj_decode.h): few equally likely byte
    values (codes of one length: a decoder never falls into step from the wrong phase), two lengths, geometric distributions (fast to fall into step), a near-uniform one
    with a rare long code (slow), alphabets of 2 to 200 values, runs of repeats in between so that the frames have sequences as well; sizes to 128 KiB"""
    import random
    rnd = random.Random(seed)
    out = []
    def draw(n, values, weights):
        return bytes(rnd.choices(values, weights=weights, k=n))
    for n in (2048, 5000, 20000, 65536, 131072):
        for nsym, shape in ((2, "flat"), (4, "flat"), (16, "flat"), (64, "flat"), (3, "geo"), (12, "geo"), (40, "geo"), (200, "geo"), (17, "near"), (33, "near"), (6, "two")):
            values = rnd.sample(range(256), nsym)
            if shape == "flat": w = [1.0] * nsym
            elif shape == "geo": w = [0.7 ** i + 1e-4 for i in range(nsym)]
            elif shape == "near": w = [1.0] * (nsym - 1) + [0.02]
            else: w = [4.0] * 2 + [1.0] * (nsym - 2)
            d = bytearray(draw(n, values, w))
            for _ in range(n // 4000):                                  # some matches: the frames get sequences (the batch pipeline's "simple" class)
                a = rnd.randrange(0, max(1, n - 300)); ln = rnd.randrange(8, 200); b = rnd.randrange(0, max(1, n - ln))
                d[b:b + ln] = d[a:a + ln][:len(d[b:b + ln])]
            out.append(bytes(d[:n]))
    return out
