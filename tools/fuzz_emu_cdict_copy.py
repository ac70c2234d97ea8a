"""Randomised byte-identity stress of dictionary compression beyond the attach range (copy mode: ZSTD_resetCCtx_byCopyingCDict +
ZSTD_compressBlock_{fast,doubleFast}_extDict; zj_cdict.h ze_block_*_ext, lane-serial build of tests/emu) against the reference's
ZSTD_CCtx_refCDict + ZSTD_compress2.  usage: fuzz_emu_cdict_copy.py <seed> <seconds>   TEST INFRASTRUCTURE."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref
import util
import __graft_entry__ as e
zj = e.load_package(); L = util.emu_lib()
seed = int(sys.argv[1]); budget = float(sys.argv[2])
rnd = random.Random(seed)
recs = util.json_records(30000, seed=seed)
samples = [b",".join(recs[i * 13:i * 13 + 200])[:4096] for i in range(1500)]
dicts = [ref.train_dict(samples, sz) for sz in (4096, 16384, 112640, 250000)] + [b",".join(recs[:400]), b",".join(recs[100:130])]
pairs = [(d, lvl, ref.CDict(d, lvl), util.EmuCDict(L, d, lvl)) for d in dicts for lvl in (1, 2, 3)]
def piece(n, d):
    k = rnd.randrange(7)
    if k == 0: return os.urandom(n)
    if k == 1: i = rnd.randrange(0, len(recs) - 3000); return b",".join(recs[i:i + 3000])[:n]
    if k == 2: return zj.synth_host(max(n, 1), rnd.randrange(1 << 20), 1)[:n]
    if k == 3: o = rnd.randrange(0, max(1, len(d) - 1)); return (d[o:o + n] * (n // max(1, len(d[o:o + n])) + 1))[:n]      # straight out of the dictionary
    if k == 4: per = os.urandom(rnd.choice([1, 3, 8, 64, 300])); return (per * (n // len(per) + 1))[:n]
    if k == 5: return bytes([rnd.getrandbits(8)]) * n
    a = piece(n // 2, d); return (a + piece(n - len(a), d))[:n]
t0 = time.time(); cases = bad = refused = 0
while time.time() - t0 < budget:
    d, lvl, rc, ec = pairs[rnd.randrange(len(pairs))]
    cutoff = 16384 if ec.info()["strategy"] == 2 else 8192
    size = rnd.choice([cutoff + 1, rnd.randrange(cutoff + 1, 40000), rnd.randrange(cutoff + 1, 131073), 65536, 131071, 131072])
    parts = []
    while sum(map(len, parts)) < size: parts.append(piece(rnd.choice([50, 700, 4096, 20000]), d))
    x = b"".join(parts)[:size]
    ck = rnd.random() < 0.2
    got = ec.compress(x, checksum=ck)
    if size == 131072 and size >= 6 * ec.info()["contentSize"]:
        ok = got == -40; refused += 1
    else:
        ok = got == rc.compress(x, checksum=ck)
    cases += 1
    if not ok:
        bad += 1; print("MISMATCH", len(d), lvl, size, ck, flush=True)
print("seed", seed, "cases", cases, "refused", refused, "bad", bad, flush=True)
