"""Randomised byte-identity stress of the need-gated double-fast machine (zj_need.h: flag kernel body + ZLaneD<E, true>) on the lane-serial
build against the reference: level 3, frames of 64 B ... 64 KiB, the library's and explicit table sizes.  NEEDMODE=1 (flags for every frame),
2 (flags for the frames zn_worth() picks, the gated machine without flags for the rest); 5 / 6 / 7: the run machine (zj_match_run.h — the product's
machine for large level-3 batches) with flags for every frame / the picked frames / none, ZJNI_EMU_JMAX=3|7 for other run lengths (372 000 frames
on its first day, 0 differences); WIDE=1: frames of 64 KiB + 1 .. 128 KiB (the wide launch: zn_flags_frame_wide, modes 5 / 6 / 8).  usage: [NEEDMODE=6] [WIDE=1] fuzz_emu_need.py <seed> <seconds>
TEST INFRASTRUCTURE."""
import sys, os, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['ZJNI_EMU_NEED']=os.environ.get('NEEDMODE','1')
import util, __graft_entry__ as e
from oracle import ref
zj=e.load_package(); L=util.emu_lib()
seed=int(sys.argv[1]); budget=float(sys.argv[2]); rnd=random.Random(seed)
words=[b"the",b"quick",b"brown",b"fox",b"jumps",b"over",b"lazy",b"dog",b"lorem",b"ipsum",b"dolor",b"sit",b"amet",b"zstd",b"frame",b"block"]
def text(n):
    out=bytearray()
    while len(out)<n: out+=rnd.choice(words)+b" "
    return bytes(out[:n])
def lowent(n,a):
    out=bytearray()
    for i in range(n):
        if i and rnd.random()<7/8: out.append(out[i-rnd.randrange(1,min(i,64)+1)])
        else: out.append(rnd.randrange(a))
    return bytes(out)
def gen(n):
    k=rnd.randrange(8)
    if k==0: return text(n)
    if k==1: return lowent(n, rnd.choice([2,16,200]))
    if k==2: return bytes(rnd.getrandbits(8) for _ in range(n))
    if k==3: return zj.synth_host(max(n,1), rnd.randrange(1<<20), 1)[:n]
    if k==4:
        per=bytes(rnd.getrandbits(8) for _ in range(rnd.choice([1,2,3,7,64,300,5000])))
        out=bytearray((per*(n//len(per)+1))[:n])
        for _ in range(rnd.choice([0,1,5,50,500])):
            if n: out[rnd.randrange(n)]=rnd.getrandbits(8)
        return bytes(out)
    if k==5: return bytes([rnd.randrange(256)])*n
    if k==6:
        a=rnd.choice([2,3,5,17,64]); return bytes(rnd.randrange(a) for _ in range(n))
    a=gen(n//2); return (a+gen(n-len(a)))[:n]
t0=time.time(); cases=0; bad=0
while time.time()-t0<budget:
    n = rnd.choice([64,65,100,1000,8191,8192,8193,16384,20000,65535,65536]) if rnd.random()<0.4 else rnd.randrange(64,65537)
    if os.environ.get('WIDE'): n = rnd.choice([65537,65544,98304,131071,131072]) if rnd.random()<0.3 else rnd.randrange(65537,131073)
    d=gen(n)
    if rnd.random()<0.3:
        hl=rnd.choice([6,10,12,14,15]); cl=rnd.choice([6,9,13,15])
    else: hl=cl=0
    ck=rnd.random()<0.2
    got=util.emu_compress(L,d,3,split=True,checksum=ck,hash_log=hl,chain_log=cl)
    if hl or cl: want=ref.compress(d,3,ck,hl or 16,cl or 15)
    elif n<=8192: want=ref.compress(d,3,ck)
    else: want=ref.compress(d,3,ck,14,13)
    cases+=1
    if got!=want:
        bad+=1
        if bad<=5:
            print("MISMATCH n=%d hl=%d cl=%d"%(n,hl,cl),flush=True); open('/tmp/need_case_%d.bin'%bad,'wb').write(d)
print("cases=%d mismatches=%d seed=%d"%(cases,bad,seed))
