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
cds={}
while time.time()-t0 < budget:
    mode = rnd.randrange(3)
    if mode==0:      # plain split path incl. wide, levels 1-3
        n = rnd.choice([rnd.randrange(0,300), rnd.randrange(0,5000), rnd.randrange(0,70000), rnd.randrange(60000,131073), 131072, 65536, 65537])
        d = gen(n); lvl = rnd.choice([1,2,3])
        want = ref.compress(d,3,False,14,13) if lvl==3 else ref.compress(d,lvl)
        got = util.emu_compress(L,d,lvl,split=True)
    elif mode==1:    # tuned tables
        n = rnd.choice([rnd.randrange(0,5000), rnd.randrange(0,70000), rnd.randrange(60000,131073)])
        d = gen(n); hl=rnd.choice([0,6,9,12,14,15,16,17]); cl=rnd.choice([0,6,9,12,13,15,16])
        if not (hl or cl): hl=16
        want = ref.compress(d,3,False,hl,cl); got = util.emu_compress(L,d,3,split=True,hash_log=hl,chain_log=cl)
    else:
        di = rnd.randrange(len(dicts)); lvl=rnd.choice([1,2,3])
        key=(di,lvl)
        if key not in cds: cds[key]=(ref.CDict(dicts[di],lvl), util.EmuCDict(L,dicts[di],lvl))
        rc,ec = cds[key]
        cut = 16384 if ec.info()['strategy']==2 else 8192
        n = rnd.choice([rnd.randrange(0,300), rnd.randrange(0,5000), rnd.randrange(0,cut+1), cut])
        d = gen(n); want = rc.compress(d); got = ec.compress(d)
    cases+=1
    if want!=got:
        bad+=1
        open(f'/tmp/fuzz_bad_{seed}_{cases}.bin','wb').write(d)
        print('MISMATCH', mode, n, len(want), len(got) if isinstance(got,bytes) else got, flush=True)
print('seed',seed,'cases',cases,'bad',bad,flush=True)
