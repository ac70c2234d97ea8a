"""Randomised stress of the decoder's REFUSALS on small hand-assembled inputs: one to three parts (reference-compressed frames with and
without checksum, skippable frames, short garbage) glued together, up to two bytes of the first 16 overwritten, sometimes truncated,
decoded into full-size / exact / one-byte-short destinations — the kernel bodies (lane-serial) and the C restatement against the
reference's portable build: same bytes, or the same error code.  ~5 000 cases per second.
usage: fuzz_emu_headers.py <seed> <seconds>   (round 1: 1.2 M cases with the final code, 0 differences.)  TEST INFRASTRUCTURE."""
import sys, random, time, collections, struct
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref, port
import util
L=util.emu_lib()
NAMES={  # ZSTD_getErrorName -> ZSTD_ErrorCode (N/zstd_errors.h)
    "Data corruption detected":20,"Src size is incorrect":72,"Destination buffer is too small":70,"Unknown frame descriptor":10,"Unsupported frame parameter":14,"Frame requires too much memory for decoding":16,"Dictionary mismatch":32,"Dictionary is corrupted":30,"Restored data doesn't match checksum":22,"Header of Literals' block doesn't respect format specification":24}
seed=int(sys.argv[1]); budget=float(sys.argv[2]); rnd=random.Random(seed)
t0=time.time(); diff=collections.Counter(); same=0; cases=0
while time.time()-t0<budget:
    n=rnd.randrange(0,400); d=bytes(rnd.randrange(4) for _ in range(n))
    parts=[]
    for _ in range(rnd.choice([1,1,2,3])):
        k=rnd.randrange(5)
        if k<=2: parts.append(ref.compress(d, rnd.choice([1,3]), checksum=rnd.random()<0.3))
        elif k==3: parts.append(struct.pack('<II',0x184D2A50+rnd.randrange(16), 5)+b'hello')
        else: parts.append(bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(0,12))))
    z=bytearray(b''.join(parts))
    for _ in range(rnd.choice([0,1,1,2])):
        if len(z): z[min(len(z)-1,rnd.randrange(0,16))]=rnd.getrandbits(8)
    if rnd.random()<0.2 and len(z): z=z[:rnd.randrange(0,len(z))]
    z=bytes(z); cap=rnd.choice([3*n+10, n, max(0,n-1)])
    try: p=ref.decompress_portable(z,cap); pc=0
    except ref.ZstdRefError as e: p=None; pc=NAMES.get(str(e), str(e))
    o=util.emu_decompress(L,z,cap)
    try: po=port.decompress(z,cap)
    except port.ZstdOracleError as e: po=-e.code
    cases+=1
    for nm,x in (('emu',o),('port',po)):
        xc=-x if isinstance(x,int) else 0
        if xc!=pc or (pc==0 and x!=p):
            diff[(nm,pc,xc)]+=1
            if diff[(nm,pc,xc)]==1: open(f'/tmp/hdr_bad_{nm}_{pc}_{xc}.zst','wb').write(z); open(f'/tmp/hdr_bad_{nm}_{pc}_{xc}.cap','w').write(str(cap))
        else: same+=1
print('cases',cases,'same',same,'diffs',dict(diff))
