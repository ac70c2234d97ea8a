"""CPU (-m "not gpu"): the frames the reference's *stream* classes produce (ZSTD_compressStream2 without a pledged size: ZstdDirectBufferCompressingStream /
ZstdOutputStream, N/jni_directbuffercompress_zstd.c:97-161) from the product's ze_compress_stream (zj_encode.h) in the lane-serial emulation
(tests/emu/emu.cpp emu_compress_stream): unknown-size parameters, header without content size, input taken in 128 KiB pieces with the savings counted as the
stream counts them, flush(), the empty raw last block, the known-empty stream — any total up to the level's window, byte-identical to oracle/ref.py compress_stream.
The GPU twin: tests/test_gpu_stream.py."""
import ctypes as C
import random

import pytest

from conftest import golden
from util import emu_lib


@pytest.fixture(scope="module")
def emu():
    L = emu_lib()
    L.emu_compress_stream.restype = C.c_ulonglong
    L.emu_compress_stream.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_uint]
    L.emu_compress_stream_flush.restype = C.c_ulonglong
    L.emu_compress_stream_flush.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint), C.c_uint]
    return L


def stream(L, d, level, checksum=False, serial=False):
    cap = len(d) + (len(d) >> 8) + 4096
    dst = C.create_string_buffer(cap)
    # an empty stream that is only ever closed: the first call is ZSTD_e_end and the size (0) is known (0x20000); 0x40000: the one-lane block parses
    r = L.emu_compress_stream(d, len(d), dst, cap, level | (0x100 if checksum else 0) | (0x20000 if not d else 0) | (0x40000 if serial else 0))
    return dst.raw[:r] if r < (1 << 63) else -((1 << 64) - r)


def test_stream_frames_rebuilt_from_the_multiblock_pieces(emu, oracle_ref, zj):
    rnd = random.Random(3)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    noise = bytes(rnd.getrandbits(8) for _ in range(70000))
    n = 0
    for size in (262145, 393216, 393217, 600000, 1048576, 2000000, 2097152):
        o = rnd.randrange(0, len(xml) - size)
        inputs = [xml[o:o + size],
                  b"".join(zj.synth_host(65536, i, 1) for i in range(size // 65536 + 1))[:size],       # the class changes every 64 KiB: pre-splits
                  (noise * 40)[:size]]                                                                 # raw blocks, then 70 000-byte matches
        for d in inputs:
            for level in (3, 1, 2):
                if size > (1 << (18 + level)):
                    assert stream(emu, d, level) == -201
                    continue
                ck = bool(n & 1); n += 1
                for chunk in (50000, 131072):                  # how the caller slices its writes does not matter: the stream buffers 128 KiB
                    assert stream(emu, d, level, ck) == oracle_ref.compress_stream(d, level, ck, chunk=chunk), (size, level, ck, chunk)
    # up to 256 KiB the stream still runs the level's default row (window 21 / 20 / 19), which the one-shot parameters of such a size are not: the experiment
    # names a parameter size apart from the frame size there.  An empty stream: the first call is ZSTD_e_end, so its size IS known (single segment, content size 0).
    for size in (0, 1, 6, 7, 8, 100, 4096, 65536, 100000, 131071, 131072, 131073, 200000, 262144):
        o = rnd.randrange(0, len(xml) - size - 1)
        for d in (xml[o:o + size], zj.synth_host(65536, 2, 5)[:size], (noise * 4)[:size]):
            for level in (3, 1, 2):
                ck = bool(n & 1); n += 1
                assert stream(emu, d, level, ck) == oracle_ref.compress_stream(d, level, ck), (size, level, ck)
                if n % 5 == 0: assert stream(emu, d, level, ck, serial=True) == oracle_ref.compress_stream(d, level, ck), (size, level, ck, "one-lane parses")


def test_stream_frames_with_flushes(emu, oracle_ref, zj):
    """flush() (ZSTD_e_flush) ends the block where the caller stands — the buffered bytes become a block of their own, the 128 KiB chunking starts again behind
    them, a flush with nothing buffered writes nothing, and close() after a flush writes the empty last block"""
    rnd = random.Random(5)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    n = 0
    for size in (0, 1000, 50000, 131072, 200000, 262144, 300000, 524288):
        o = rnd.randrange(0, len(xml) - size - 1)
        for d in (xml[o:o + size], b"".join(zj.synth_host(65536, i, 1) for i in range(size // 65536 + 1))[:size]):
            for level in (3, 1):
                if size > (1 << (18 + level)): continue
                for chunk, k in ((50000, 1), (50000, 2), (10000, 3), (131072, 1), (65536, 2), (200000, 1), (1000, 7)):
                    calls = (size + chunk - 1) // chunk
                    flushes = [min(j * chunk, size) for j in range(1, calls + 1) if j % k == 0]       # oracle/ref.py compress_stream flushes with every k-th write
                    cap = len(d) + (len(d) >> 8) + 4096 + 64 * (len(flushes) + 2)
                    dst = C.create_string_buffer(cap)
                    ck = bool(n & 1); n += 1
                    r = emu.emu_compress_stream_flush(d, len(d), dst, cap, level | (0x100 if ck else 0) | (0x20000 if not d else 0), (C.c_uint * max(len(flushes), 1))(*flushes), len(flushes))
                    assert r < (1 << 63) and dst.raw[:r] == oracle_ref.compress_stream(d, level, ck, chunk=chunk, flush_every=k), (size, level, chunk, k)


def test_stream_flushed_but_not_closed_is_the_frames_beginning(emu, oracle_ref, zj):
    """final = 0 (the caller flushed and goes on writing): the output is the frame up to the last flush — exactly the first bytes of what close() produces later,
    whatever follows; a flush with nothing written yet produces nothing"""
    rnd = random.Random(11)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    for size, flushes in ((300000, [100000]), (300000, [131072, 262144]), (70000, [70000]), (500000, [1000, 2000, 400000]), (200000, [0])):
        o = rnd.randrange(0, len(xml) - size - 1); d = xml[o:o + size]
        for level in (3, 1):
            if size > (1 << (18 + level)): continue
            fl = [f for f in flushes if f > 0]
            cap = len(d) + (len(d) >> 8) + 4096 + 64 * (len(fl) + 2)
            part = C.create_string_buffer(cap); full = C.create_string_buffer(cap)
            arr = (C.c_uint * max(len(fl), 1))(*fl)
            rp = emu.emu_compress_stream_flush(d, len(d), part, cap, level | 0x10000, arr, len(fl))
            rf = emu.emu_compress_stream_flush(d, len(d), full, cap, level, arr, len(fl))
            assert rp < (1 << 63) and rf < (1 << 63)
            if not fl: assert rp == 0
            assert full.raw[:rp] == part.raw[:rp] and rp <= rf
            # the same prefix when only the flushed part had been written at the time
            if fl:
                cut = d[:fl[-1]]
                p2 = C.create_string_buffer(cap)
                r2 = emu.emu_compress_stream_flush(cut, len(cut), p2, cap, level | 0x10000, arr, len(fl))
                assert r2 == rp and p2.raw[:r2] == part.raw[:rp]
                # and it is what the reference has written after that flush: its output for the cut stream minus the epilogue (an empty raw last block)
                if all(f == (k + 1) * fl[0] for k, f in enumerate(fl)):           # (expressible as "flush with every write of fl[0] bytes")
                        want = oracle_ref.compress_stream(cut, level, False, chunk=fl[0], flush_every=1)
                        assert want[:rp] == part.raw[:rp] and len(want) == rp + 3
