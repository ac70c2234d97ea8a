"""Randomised byte-identity stress of the wave-per-frame matcher (zj_match_wave.h, explicit-SIMT builds of tests/emu: ascending and
descending lane order) against the reference's level 3 with hashLog 14 / chainLog 13, frames of 64 B .. 64 KiB.
usage: fuzz_emu_wave.py <seed> <seconds>   TEST INFRASTRUCTURE."""
import sys, time, random
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref
import util
import __graft_entry__ as e
zj = e.load_package()
WL = util.emu_wave_libs()
seed = int(sys.argv[1]); budget = float(sys.argv[2])
rnd = random.Random(seed)
recs = util.json_records(20000, seed=seed)
words = [b"the",b"quick",b"brown",b"fox",b"jumps",b"over",b"lazy",b"dog",b"lorem",b"ipsum",b"dolor",b"sit",b"amet",b"zstd",b"frame",b"block"]
def text(n):
    out=bytearray()
    while len(out)<n: out+=rnd.choice(words)+b" "
    return bytes(out[:n])
def lowent(n):
    out=bytearray()
    for i in range(n):
        if i and rnd.random()<7/8: out.append(out[i-rnd.randrange(1,min(i,64)+1)])
        else: out.append(rnd.randrange(16))
    return bytes(out)
def gen_shaped(n):
    """a stream of (fresh literals, copy of earlier bytes) pairs whose lengths and distances come from very skewed
    distributions: the LL / ML / OF code histograms get one dominant code plus a tail of rare ones, which is where
    FSE_normalizeCount's low-probability and secondary paths live"""
    out = bytearray(bytes(rnd.getrandbits(8) for _ in range(min(n, 40))))
    dom = (rnd.choice([0, 1, 2, 3, 8]), rnd.choice([4, 5, 6, 7, 8, 12, 35, 67]), rnd.choice([1, 2, 3, 8, 16, 37, 256, 1000]))
    pd = rnd.choice([0.5, 0.9, 0.97, 0.995])
    while len(out) < n:
        if rnd.random() < pd: ll, ml, off = dom
        else: ll, ml, off = rnd.choice([0, 1, 2, 5, 17, 40, 100, 300]), rnd.choice([4, 5, 6, 9, 20, 50, 130, 600]), rnd.randrange(1, len(out) + 1)
        out += bytes(rnd.getrandbits(8) for _ in range(ll))
        off = min(off, len(out))
        for _ in range(ml): out.append(out[-off])
    return bytes(out[:n])
def gen_edge(n):
    """inputs aimed at thresholds of the entropy stage: near-uniform small alphabets, one dominant byte, short periods,
    literal / sequence counts around the format's size classes"""
    k = rnd.randrange(8)
    if k >= 6: return gen_shaped(n)
    if k == 0:                                      # uniform over an alphabet of a few to 256 values: many equal counts
        a = rnd.choice([2, 3, 5, 16, 17, 64, 100, 200, 256]); base = rnd.randrange(0, 257 - a)
        return bytes(base + rnd.randrange(a) for _ in range(n))
    if k == 1:                                      # one dominant byte with sprinkles (rle / near-rle literals)
        b = rnd.randrange(256); p = rnd.choice([0.0, 0.001, 0.01, 0.1])
        return bytes(b if rnd.random() >= p else rnd.randrange(256) for _ in range(n))
    if k == 2:                                      # short period with mutations: long matches, repcodes
        per = bytes(rnd.getrandbits(8) for _ in range(rnd.choice([1, 2, 3, 4, 5, 7, 8, 16, 63, 64, 65, 300])))
        out = bytearray((per * (n // len(per) + 1))[:n])
        for _ in range(rnd.choice([0, 1, 5, 50])):
            if n: out[rnd.randrange(n)] = rnd.getrandbits(8)
        return bytes(out)
    if k == 3:                                      # exactly-equal counts: a shuffled multiset
        a = rnd.choice([9, 12, 40, 130, 256]); c = rnd.choice([1, 2, 20, 163, 164, 165, 166, 255, 256])
        v = [x for x in range(a) for _ in range(c)][:max(n, 1)]
        rnd.shuffle(v)
        return bytes(v[:n])
    if k == 4:                                      # incompressible head + compressible tail (and the reverse)
        h = bytes(rnd.getrandbits(8) for _ in range(n // 2)); t = text(n - len(h))
        return (h + t) if rnd.random() < 0.5 else (t + h)
    a = gen(n // 3); b = gen_edge(n // 3)
    return (a + b + gen(n - len(a) - len(b)))[:n]
EDGE_SIZES = [0, 1, 5, 6, 7, 8, 9, 12, 13, 62, 63, 64, 65, 255, 256, 257, 1022, 1023, 1024, 1025, 4095, 4096, 4097, 16383, 16384, 16385, 65535, 65536, 65537, 131071, 131072]
def gen(n):
    k = rnd.randrange(6)
    if k==0: return text(n)
    if k==1: return lowent(n)
    if k==2: return bytes(rnd.getrandbits(8) for _ in range(n))
    if k==3:
        i=rnd.randrange(0,len(recs)-2000); return b",".join(recs[i:i+2000])[:n]
    if k==4: return zj.synth_host(max(n,1), rnd.randrange(1<<20), 1)[:n]
    a=gen(n//2); return (a+gen(n-len(a)))[:n]
t0=time.time(); cases=0; bad=0
while time.time()-t0 < budget:
    n = rnd.choice([rnd.randrange(64,300), rnd.randrange(64,5000), rnd.randrange(64,65537), 65536, 65535, 64, 65, 4096])
    d = (gen_edge if rnd.random() < 0.5 else gen)(n)
    if len(d) < 64: continue
    ck = rnd.random() < 0.2
    want = ref.compress(d,3,ck,14,13)
    for k, L in enumerate(WL):
        got = util.emu_compress_wave(L, d, 3, ck)
        cases+=1
        if want!=got:
            bad+=1
            open(f'/tmp/fuzz_wave_bad_{seed}_{cases}.bin','wb').write(d)
            print('MISMATCH', k, n, len(want), len(got) if isinstance(got,bytes) else got, flush=True)
print('seed',seed,'cases',cases,'bad',bad,flush=True)
