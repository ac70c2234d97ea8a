"""Randomised stress of the multi-block stages of the split decode pipeline (zd_prep_frame_multi -> ZDSeqLaneT<true> per block ->
zd_lit_block -> zd_exec_frame_multi; lane-serial build) against the reference: one-shot frames of several blocks (levels 1-19,
checksum on/off), stream frames with flushes, and damaged copies of both (bit flips, byte stores, truncation, short destinations),
which have to be answered like the reference's portable decoder loops answer them (bytes, or the same error code).
(Rounds 1-5 counted one class apart: a compressed block of exactly 128 KiB, refused in the block loop; entered like the reference enters it since round 6.)
usage: fuzz_emu_decode_mb.py <seed> <seconds> [EMU_MB_LIT]   TEST INFRASTRUCTURE."""
import sys, time, random, os
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
seed = int(sys.argv[1]); budget = float(sys.argv[2])
os.environ["EMU_MB_LIT"] = sys.argv[3] if len(sys.argv) > 3 else "2"
from oracle import ref
import util
import __graft_entry__ as e
zj = e.load_package()
L = util.emu_lib()
L.emu_decompress_mb.restype = C.c_ulonglong
L.emu_decompress_mb.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_ulonglong, C.POINTER(C.c_int)]
def mb(frame, cap):
    dst = C.create_string_buffer(max(cap, 1)); used = C.c_int(0)
    r = L.emu_decompress_mb(frame, len(frame), dst, cap, C.byref(used))
    return (dst.raw[:r] if r < (1 << 63) else -((1 << 64) - r)), used.value
rnd = random.Random(seed)
recs = util.json_records(20000, seed=seed)
def gen(n):
    k = rnd.randrange(6)
    if k == 0: return bytes(rnd.getrandbits(8) for _ in range(min(n, 40000))) * (n // 40000 + 1)
    if k == 1:
        i = rnd.randrange(0, len(recs) - 3000); return (b",".join(recs[i:i + 3000]) * (n // 100000 + 1))
    if k == 2: return b"".join(zj.synth_host(65536, rnd.randrange(1 << 20), 1) for _ in range(n // 65536 + 1))
    if k == 3:
        b = rnd.randrange(256); p = rnd.choice([0.0, 0.0001, 0.001, 0.05])
        return bytes(b if rnd.random() >= p else rnd.randrange(256) for _ in range(n))
    if k == 4:
        per = bytes(rnd.getrandbits(8) for _ in range(rnd.choice([1, 3, 7, 64, 300, 5000, 131072, 140000])))
        return per * (n // len(per) + 1)
    parts = []; left = n
    while left > 0:
        m = min(left, rnd.choice([1000, 30000, 131072, 200000])); parts.append(gen(m)[:m]); left -= m
    return b"".join(parts)
t0 = time.time(); cases = 0; bad = 0; corrupted = 0; served = 0
while time.time() - t0 < budget:
    n = rnd.choice([rnd.randrange(131073, 400000), rnd.randrange(131073, 1200000), 262144, 262145, 393216, rnd.randrange(0, 131073)])
    d = gen(n)[:n]; lvl = rnd.choice([1, 2, 3, 4, 5, 7, 9, 12, 16, 19] if n < 600000 else [1, 3, 5, 9])
    if rnd.random() < 0.4:
        z = ref.compress_stream(d, lvl, rnd.random() < 0.4, chunk=rnd.choice([300, 1000, 7000, 30000, 131072, 200000]), flush_every=rnd.choice([0, 0, 1, 3]))
        kind = 'stream'
    else:
        z = ref.compress(d, lvl, checksum=rnd.random() < 0.4); kind = 'oneshot'
    cap = len(d) + rnd.choice([0, 0, 1, 100])
    out, used = mb(z, cap); cases += 1; served += used
    if out != d:
        bad += 1; open(f'/tmp/fuzz_mb_bad_{seed}_{cases}.zst', 'wb').write(z)
        print('MISMATCH', kind, n, lvl, out if isinstance(out, int) else len(out), flush=True)
    if len(z) > 12 and rnd.random() < 0.7:
        for _ in range(4):
            zb = bytearray(z); m = rnd.randrange(7)
            if m <= 2: zb[rnd.randrange(4, len(zb))] ^= 1 << rnd.randrange(8)
            elif m == 3: zb[rnd.randrange(4, len(zb))] = rnd.getrandbits(8)
            elif m == 4:
                for _ in range(3): zb[rnd.randrange(4, len(zb))] ^= 1 << rnd.randrange(8)
            elif m == 5: zb = zb[:rnd.randrange(5, len(zb))]
            else: zb[rnd.randrange(4, min(len(zb), 40))] ^= 1 << rnd.randrange(8)               # headers of the frame and its first block
            zb = bytes(zb); corrupted += 1
            c2 = len(d) if rnd.random() < 0.8 else rnd.randrange(0, len(d) + 1)
            try: want = ref.decompress_portable(zb, c2)
            except ref.ZstdRefError as ex: want = -ex.code
            got, _ = mb(zb, c2)
            if got != want:
                bad += 1; open(f'/tmp/fuzz_mb_bad_{seed}_{cases}_{corrupted}.zst', 'wb').write(zb)
                print('REFDIFF', kind, n, lvl, 'cap', c2, 'portable', want if isinstance(want, int) else len(want), 'ours', got if isinstance(got, int) else len(got), flush=True)
print('seed', seed, 'cases', cases, 'served_by_block_stages', served, 'corrupted', corrupted, 'bad', bad, flush=True)
