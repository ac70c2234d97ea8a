"""Randomised byte-identity stress of level 4 (greedy on the hash chain <= 16 KiB, double-fast with 2^17-entry tables <= 128 KiB; the
single-block route of zj_encode_multi_kernel, lane-serial build of tests/emu) against the reference's ZSTD_compress2.
usage: fuzz_emu_level4.py <seed> <seconds>   TEST INFRASTRUCTURE."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref
import util
import __graft_entry__ as e
zj = e.load_package(); L = util.emu_lib()
seed = int(sys.argv[1]); budget = float(sys.argv[2])
rnd = random.Random(seed)
recs = util.json_records(20000, seed=seed)
def piece(n):
    k = rnd.randrange(8)
    if k == 0: return os.urandom(n)
    if k == 1: i = rnd.randrange(0, len(recs) - 3000); return b",".join(recs[i:i + 3000])[:n]
    if k == 2: return zj.synth_host(max(n, 1), rnd.randrange(1 << 20), 1)[:n]
    if k == 3: return bytes([rnd.getrandbits(8)]) * n
    if k == 4:
        per = os.urandom(rnd.choice([1, 2, 3, 4, 5, 8, 16, 63, 64, 65, 300, 5000])); return (per * (n // len(per) + 1))[:n]
    if k == 5:
        a = rnd.choice([2, 3, 5, 16, 64, 200]); base = rnd.randrange(0, 257 - a); return bytes(base + rnd.randrange(a) for _ in range(n))
    if k == 6:
        out = bytearray()
        while len(out) < n: out += bytes([rnd.getrandbits(8)]) * rnd.randrange(1, 900) + os.urandom(rnd.randrange(0, 12))
        return bytes(out[:n])
    a = piece(n // 2); return (a + piece(n - len(a)))[:n]
t0 = time.time(); cases = bad = greedy = 0
while time.time() - t0 < budget:
    size = rnd.choice([rnd.randrange(0, 64), rnd.randrange(0, 1200), rnd.randrange(0, 16385), rnd.randrange(0, 16385), rnd.randrange(16385, 131073), 16384, 16385, 131072])
    parts = []
    while sum(map(len, parts)) < size:
        parts.append(piece(rnd.choice([7, 100, 500, 4096, 8192, 40000])))
        if rnd.random() < 0.3 and parts: parts.append(parts[rnd.randrange(len(parts))])
    d = b"".join(parts)[:size]
    ck = rnd.random() < 0.2; cs = rnd.random() < 0.85
    lvl = rnd.choice([4, 5, 6, 7, 8])                                   # levels 5-8: hash chain up to 16 KiB, the row-based finder above
    got = util.emu_compress_chain(L, d, lvl, ck, cs) if (len(d) <= 16384 and rnd.random() < 0.5) else util.emu_compress_multi(L, d, lvl, ck, cs)     # both routes of the kernels
    want = ref.compress(d, lvl, ck, content_size=cs)
    cases += 1; greedy += len(d) <= 16384
    if got != want:
        bad += 1; open(f"/tmp/fuzz_l4_bad_{seed}_{cases}.bin", "wb").write(d); print("MISMATCH", size, lvl, ck, cs, flush=True)
print("seed", seed, "cases", cases, "greedy", greedy, "bad", bad, flush=True)
