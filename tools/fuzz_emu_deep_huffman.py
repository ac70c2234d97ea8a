"""Randomised stress of the two decoder classes closed in round 6, kernel bodies lane-serial, against the reference's portable build on all three pipelines (fused,
three-stage, block stages): (a) literals coded with Huffman tables 12 bits deep (HUF_TABLELOG_MAX) — sections from the reference's own HUF_compress{1,4}X_repeat
at tableLog 12 and hand-built codes with every count of 12-bit symbols, as only block, followed by treeless blocks that reuse the table, valid and damaged;
(b) compressed blocks of exactly 128 KiB (N/decompress/zstd_decompress_block.c:2073-2081), valid and damaged.  Same bytes or the same refusal code.
usage: fuzz_emu_deep_huffman.py <seed> <seconds>   TEST INFRASTRUCTURE."""
import ctypes as C, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref
import util
L = util.emu_lib()
L.emu_decompress_mb.restype = C.c_ulonglong
L.emu_decompress_mb.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_ulonglong, C.POINTER(C.c_int)]
seed = int(sys.argv[1]); budget = float(sys.argv[2]); rnd = random.Random(seed)
def mb(frame, cap):
    dst = C.create_string_buffer(max(cap, 1)); used = C.c_int(0)
    r = L.emu_decompress_mb(frame, len(frame), dst, cap, C.byref(used))
    return dst.raw[:r] if r < (1 << 63) else -((1 << 64) - r)
def portable(frame, cap):
    try: return ref.decompress_portable(frame, cap)
    except ref.ZstdRefError as ex: return -ex.code
def ours(frame, cap):
    return (util.emu_decompress(L, frame, cap), util.emu_decompress_split(L, frame, cap)[0], mb(frame, cap))
def tree_len(sec): return sec[0] + 1 if sec[0] < 128 else 1 + (sec[0] - 127 + 1) // 2
t0 = time.time(); cases = 0; bad = 0; damaged = 0; deep = 0
while time.time() - t0 < budget:
    k = rnd.random(); blocks = []; want = b""
    if k < 0.45:
        n = rnd.choice([rnd.randrange(16500, 40000), rnd.randrange(16500, 131072)])
        lits, sec, depth = util.deep_huffman_literals(ref, n, rnd.randrange(1 << 30)); deep += depth == 12
        nb = rnd.choice([1, 1, 2, 3])
        for b in range(nb):
            tl = b > 0 and rnd.random() < 0.7
            blocks.append(util.literals_only_block(n, sec[4][tree_len(sec[4]):] if tl else sec[4], 4, last=b == nb - 1, treeless=tl))
        want = lits * nb
    elif k < 0.9:
        streams = rnd.choice([1, 4]); n = rnd.randrange(6, 1000) if streams == 1 else rnd.randrange(6, 20000)
        lits, sec = util.hand_huffman_section(rnd, n, rnd.choice([2, 2, 4, 6, 8, 16, 30, 62, 64, 100, 116]), streams); deep += 1
        if streams == 1 and len(sec) >= 1024: continue
        nb = rnd.choice([1, 1, 2])
        for b in range(nb):
            tl = b > 0 and rnd.random() < 0.7
            blocks.append(util.literals_only_block(n, sec[tree_len(sec):] if tl else sec, streams, last=b == nb - 1, treeless=tl))
        want = lits * nb
    else:
        n_lit = 131072 - 3 - 1; lits = bytes(rnd.getrandbits(8) for _ in range(n_lit))
        body = bytes([0 | 3 << 2 | (n_lit & 0xF) << 4, (n_lit >> 4) & 0xFF, n_lit >> 12]) + lits + b"\x00"
        last = rnd.random() < 0.5
        blocks = [(131072 << 3 | 2 << 1 | (1 if last else 0)).to_bytes(3, "little") + body] + ([] if last else [b"\x01\x00\x00"])
        want = lits
    frame = util.frame_of_blocks(blocks, content_size=len(want) if rnd.random() < 0.5 else None)
    cap = len(want) + rnd.choice([0, 0, 7])
    p = portable(frame, cap); o = ours(frame, cap); cases += 1
    if p != want or any(x != p for x in o):
        bad += 1; open(f'/tmp/fuzz_deep_bad_{seed}_{cases}.zst', 'wb').write(frame); print('MISMATCH valid', len(want), p if isinstance(p, int) else len(p), [x if isinstance(x, int) else len(x) for x in o], flush=True)
    for _ in range(6):
        zb = bytearray(frame); m = rnd.randrange(6)
        if m <= 1: zb[rnd.randrange(4, len(zb))] ^= 1 << rnd.randrange(8)
        elif m == 2: zb[rnd.randrange(4, min(len(zb), 200))] ^= 1 << rnd.randrange(8)          # headers and the tree description
        elif m == 3: zb[rnd.randrange(4, len(zb))] = rnd.getrandbits(8)
        elif m == 4:
            for _ in range(3): zb[rnd.randrange(4, len(zb))] ^= 1 << rnd.randrange(8)
        else: zb = zb[:rnd.randrange(5, len(zb))]
        zb = bytes(zb); damaged += 1
        c2 = cap if rnd.random() < 0.85 else rnd.randrange(0, cap + 1)
        p = portable(zb, c2); o = ours(zb, c2)
        if any(x != p for x in o):
            bad += 1; open(f'/tmp/fuzz_deep_bad_{seed}_{cases}_{damaged}.zst', 'wb').write(zb)
            print('REFDIFF cap', c2, 'portable', p if isinstance(p, int) else len(p), 'ours', [x if isinstance(x, int) else len(x) for x in o], flush=True)
print('seed', seed, 'cases', cases, 'tables_12_bits_deep', deep, 'damaged', damaged, 'bad', bad, flush=True)
