"""Randomised byte-identity stress of level-3 single-block frames on the wave route (ZJNI_ROUTE_WAVE_HBM: the multi-block kernel's single-block entry,
wave matcher zj_match_wavex.h over HBM tables with the level's own 16 / 15 table sizes or explicit ones; explicit-SIMT build of tests/emu, both lane orders
via ZJNI_EMU_LIB) against the reference's ZSTD_compress2.   usage: fuzz_emu_l3wave.py <seed> <seconds>   TEST INFRASTRUCTURE."""
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
t0 = time.time(); cases = bad = 0
while time.time() - t0 < budget:
    n = rnd.choice([rnd.randrange(0, 200), rnd.randrange(0, 5000), rnd.randrange(0, 70000), rnd.randrange(0, 131073), 65536, 131072, 16384, 16385])
    d = piece(n) if n else b""
    hl, cl = rnd.choice([(16, 15), (16, 15), (16, 15), (17, 16), (14, 13), (12, 15), (rnd.randrange(6, 18), rnd.randrange(6, 17))])
    ck = rnd.random() < 0.2
    got = util.emu_compress_multi(L, d, 3, ck, True, rnd.choice([False, False, 2]), hl, cl)
    want = ref.compress(d, 3, ck) if (hl, cl) == (16, 15) else ref.compress(d, 3, ck, hl, cl)
    cases += 1
    if got != want:
        bad += 1; open(f"/tmp/fuzz_l3wave_bad_{seed}_{cases}.bin", "wb").write(d); print("MISMATCH", n, hl, cl, ck, flush=True)
print("seed", seed, "cases", cases, "bad", bad, flush=True)
