"""Randomised byte-identity stress of multi-block frames (ze_compress_multi, lane-serial build of tests/emu) against the reference's
ZSTD_compress2, 128 KiB < input <= 2 MiB, levels 1-3, with a census of the block / literals types the inputs produced.
Level-3 blocks run the wave matcher (zj_match_wavex.h, 64 emulated lanes; ZJNI_EMU_LIB=tests/emu/libzjni_emu_rev.so visits them in
descending order), FUZZ_SERIAL=1 the one-lane parse, FUZZ_SERIAL=2 the wave matcher without staged spans; FUZZ_LEVELS=3 restricts the levels.
usage: fuzz_emu_multiblock.py <seed> <seconds>   TEST INFRASTRUCTURE."""
import os, sys, time, random, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref
import util
import __graft_entry__ as e
zj = e.load_package(); L = util.emu_lib()
seed = int(sys.argv[1]); budget = float(sys.argv[2])
LEVELS = [int(x) for x in os.environ.get("FUZZ_LEVELS", "1,2,3").split(",")]; SERIAL = {"1": True, "2": 2}.get(os.environ.get("FUZZ_SERIAL"), False)
rnd = random.Random(seed)
recs = util.json_records(20000, seed=seed)
census = collections.Counter()
def blocks(z):
    fhd = z[4]; single = (fhd >> 5) & 1; fcs = fhd >> 6; pos = 5 + (0 if single else 1) + (0, 1, 2, 4)[fhd & 3] + ((1 if single else 0) if fcs == 0 else (2, 4, 8)[fcs - 1])
    while True:
        bh = z[pos] | (z[pos + 1] << 8) | (z[pos + 2] << 16); last, typ, size = bh & 1, (bh >> 1) & 3, bh >> 3
        census["block_" + ("raw", "rle", "compressed")[typ]] += 1
        if typ == 2: census["literals_" + ("raw", "rle", "huffman", "treeless")[z[pos + 3] & 3]] += 1
        pos += 3 + (1 if typ == 1 else size)
        if last: break
def piece(n):
    k = rnd.randrange(8)
    if k == 0: return os.urandom(n)
    if k == 1: i = rnd.randrange(0, len(recs) - 3000); return b",".join(recs[i:i + 3000])[:n]
    if k == 2: return zj.synth_host(max(n, 1), rnd.randrange(1 << 20), 1)[:n]
    if k == 3: return bytes([rnd.getrandbits(8)]) * n
    if k == 4:
        per = os.urandom(rnd.choice([1, 2, 3, 5, 8, 16, 63, 64, 65, 300, 5000])); return (per * (n // len(per) + 1))[:n]
    if k == 5:
        a = rnd.choice([2, 3, 5, 16, 64, 200]); base = rnd.randrange(0, 257 - a); return bytes(base + rnd.randrange(a) for _ in range(n))
    if k == 6:                                      # long runs with sprinkles: few literals per block, treeless candidates
        out = bytearray()
        while len(out) < n: out += bytes([rnd.getrandbits(8)]) * rnd.randrange(200, 9000) + os.urandom(rnd.randrange(0, 12))
        return bytes(out[:n])
    a = piece(n // 2); return (a + piece(n - len(a)))[:n]
PIPE = os.environ.get("FUZZ_PIPE") == "1"      # zj_encode_pipe_kernel's two roles (round 6: the parse role a block ahead of the entropy role) instead of the one-wave loop
t0 = time.time(); cases = bad = 0
while time.time() - t0 < budget:
    size = rnd.choice([rnd.randrange(131073, 300000), rnd.randrange(131073, 600000), rnd.randrange(131073, 2097153), 131073, 262144, 262145, 524288, 1048576])
    parts = []
    while sum(map(len, parts)) < size:
        parts.append(piece(rnd.choice([500, 8192, 40000, 131072, 131072, 300000])))
        if rnd.random() < 0.2 and parts: parts.append(parts[rnd.randrange(len(parts))])
    d = b"".join(parts)[:size]
    lvl = rnd.choice(LEVELS); ck = rnd.random() < 0.2; cs = rnd.random() < 0.85
    got = util.emu_compress_multi(L, d, lvl, ck, cs, SERIAL, pipelined=PIPE)
    if len(d) > (1 << (18 + lvl)):
        ok = got == -201
    else:
        want = ref.compress(d, lvl, ck, content_size=cs); ok = got == want
        if ok: blocks(want)
    cases += 1
    if not ok:
        bad += 1; open(f"/tmp/fuzz_multi_bad_{seed}_{cases}.bin", "wb").write(d); print("MISMATCH", size, lvl, ck, cs, flush=True)
print("seed", seed, "cases", cases, "bad", bad, dict(census), flush=True)
