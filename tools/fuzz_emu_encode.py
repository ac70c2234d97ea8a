"""Randomised byte-identity stress of the encoder bodies (lane-serial build, tests/emu) against the reference: the plain
lane pipeline incl. the wide launch, explicit table sizes, dictionary compression.  usage: fuzz_emu_encode.py <seed> <seconds>
(round 1: 3.7 M cases; one finding, the HUF_sort slot, fixed; 0 mismatches in the 1.7 M cases since.)  TEST INFRASTRUCTURE."""
import sys, time, random
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref
import util
import __graft_entry__ as e
zj = e.load_package()
L = util.emu_lib()
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
dicts=[]
samples=[b",".join(recs[i*13:i*13+200])[:4096] for i in range(1000)]
for dsz in (4096, 30000, 112640):
    dicts.append(ref.train_dict(samples+[text(4096) for _ in range(100)], dsz))
dicts.append(text(20000)); dicts.append(b",".join(recs[:200]))
dicts.append(bytes(rnd.getrandbits(8) for _ in range(9)))          # tiny raw dictionaries: every match into them is near the seam
dicts.append(text(64)); dicts.append(bytes(rnd.getrandbits(8) for _ in range(3000)))
cds={}
while time.time()-t0 < budget:
    mode = rnd.randrange(3)
    if mode==0:      # plain split path incl. wide, levels 1-3
        n = rnd.choice([rnd.randrange(0,300), rnd.randrange(0,5000), rnd.randrange(0,70000), rnd.randrange(60000,131073), 131072, 65536, 65537])
        if rnd.random() < 0.3: n = rnd.choice(EDGE_SIZES)
        d = (gen_edge if rnd.random() < 0.5 else gen)(n); lvl = rnd.choice([1,2,3])
        want = ref.compress(d,3,False,14,13) if lvl==3 else ref.compress(d,lvl)
        got = util.emu_compress(L,d,lvl,split=True)
    elif mode==1:    # tuned tables
        n = rnd.choice([rnd.randrange(0,5000), rnd.randrange(0,70000), rnd.randrange(60000,131073)])
        if rnd.random() < 0.3: n = rnd.choice(EDGE_SIZES)
        d = (gen_edge if rnd.random() < 0.5 else gen)(n); hl=rnd.choice([0,6,9,12,14,15,16,17]); cl=rnd.choice([0,6,9,12,13,15,16])
        if not (hl or cl): hl=16
        want = ref.compress(d,3,False,hl,cl); got = util.emu_compress(L,d,3,split=True,hash_log=hl,chain_log=cl)
    else:
        di = rnd.randrange(len(dicts)); lvl=rnd.choice([1,2,3])
        key=(di,lvl)
        if key not in cds: cds[key]=(ref.CDict(dicts[di],lvl), util.EmuCDict(L,dicts[di],lvl))
        rc,ec = cds[key]
        cut = 16384 if ec.info()['strategy']==2 else 8192
        n = rnd.choice([rnd.randrange(0,300), rnd.randrange(0,5000), rnd.randrange(0,cut+1), cut])
        if rnd.random() < 0.3: n = min(cut, rnd.choice(EDGE_SIZES))
        d = (gen_edge if rnd.random() < 0.5 else gen)(n)
        if rnd.random() < 0.3 and n > 16:                          # splice dictionary content in: matches that start in the dictionary,
            dc = dicts[di][-min(len(dicts[di]), 6000):]            # run over its end into the source, repcodes across the seam
            out = bytearray(d)
            for _ in range(rnd.choice([1, 2, 5])):
                ln = rnd.randrange(4, min(len(dc), n, 300) + 1); a = rnd.randrange(0, len(dc) - ln + 1); b = rnd.randrange(0, n - ln + 1)
                out[b:b + ln] = dc[a:a + ln]
            if rnd.random() < 0.3: out[:min(n, 40)] = dc[-min(n, 40):][:min(n, 40)]
            d = bytes(out)
        want = rc.compress(d); got = ec.compress(d)
    cases+=1
    if want!=got:
        bad+=1
        open(f'/tmp/fuzz_bad_{seed}_{cases}.bin','wb').write(d)
        print('MISMATCH', mode, n, len(want), len(got) if isinstance(got,bytes) else got, flush=True)
print('seed',seed,'cases',cases,'bad',bad,flush=True)
