"""Randomised stress with DAMAGED DICTIONARIES (flipped bits / stored bytes in the header and entropy tables, truncation) against the
reference, kernel bodies lane-serial:  decode <seed> <seconds>  — frames compressed with the good dictionary, decoded with the
damaged one on both pipelines: same bytes or the same refusal code as the reference's portable build;  encode <seed> <seconds> —
ZSTD_createCDict accepts / refuses the same dictionaries, and where it accepts the frames are byte-identical (dictionaries under 8 bytes excepted: the reference
ignores them, the device digest refuses them and the shim leaves them to the bundled library).
Round 1: 25 M decode and 1.2 M encode cases; the only differences then: dictionaries whose damaged Huffman table is 12 bits deep — refused at load in rounds 1-5,
loaded like the reference loads them since round 6 (zd_huf_fill's 2 048-cell form of a 12-bit table), so no class is counted apart any more.  TEST INFRASTRUCTURE."""
import sys
mode = sys.argv.pop(1)

def run_decode():
    import sys, random, time, collections
    import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import ref
    import util
    L=util.emu_lib()
    seed=int(sys.argv[1]); budget=float(sys.argv[2]); rnd=random.Random(seed)
    recs=util.json_records(20000, seed=seed)
    samples=[b",".join(recs[i*13:i*13+200])[:4096] for i in range(1000)]
    dic=ref.train_dict(samples, 20000)
    datas=[b",".join(recs[i:i+40])[:rnd.randrange(100,6000)] for i in range(0,4000,100)]
    frames=[(d,ref.compress_using_dict(d,dic,rnd.choice([1,3]))) for d in datas]
    t0=time.time(); cases=0; diff=collections.Counter(); same=0; acc=0
    while time.time()-t0<budget:
        bd=bytearray(dic); m=rnd.randrange(5)
        if m==0: bd[rnd.randrange(8,400)]^=1<<rnd.randrange(8)
        elif m==1: bd[rnd.randrange(8,400)]=rnd.getrandbits(8)
        elif m==2: bd=bd[:rnd.randrange(0,600)]
        elif m==3:
            for _ in range(3): bd[rnd.randrange(0,300)]^=1<<rnd.randrange(8)
        else: bd[rnd.randrange(0,len(bd))]^=1<<rnd.randrange(8)
        bd=bytes(bd)
        d,z=rnd.choice(frames)
        try: p=ref.decompress_portable(z,len(d),bd)
        except ref.ZstdRefError as e: p=-e.code
        o=util.emu_decompress_dict(L,z,len(d),bd); o2=util.emu_decompress_dict(L,z,len(d),bd,split=True)
        cases+=1
        if not isinstance(p,int): acc+=1
        for x in (o,o2):
            if x!=p:
                key=(p if isinstance(p,int) else 'ok', x if isinstance(x,int) else 'ok/bytes differ')
                diff[key]+=1
                if diff[key]==1: open(f'/tmp/dict_bad_{seed}_{cases}.dict','wb').write(bd); open(f'/tmp/dict_bad_{seed}_{cases}.zst','wb').write(z); print('first',key,'case',cases,'mode',m,flush=True)
            else: same+=1
    print('cases',cases,'accepted',acc,'same',same,'diffs',dict(diff))

def run_encode():
    import sys, random, time, collections
    import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import ref
    import util
    L=util.emu_lib()
    seed=int(sys.argv[1]); budget=float(sys.argv[2]); rnd=random.Random(seed)
    recs=util.json_records(20000, seed=seed)
    samples=[b",".join(recs[i*13:i*13+200])[:4096] for i in range(1000)]
    dic=ref.train_dict(samples, 20000)
    datas=[b",".join(recs[i:i+40])[:rnd.randrange(100,6000)] for i in range(0,2000,100)]
    t0=time.time(); cases=0; diff=collections.Counter(); both_ok=0; both_rej=0
    while time.time()-t0<budget:
        bd=bytearray(dic); m=rnd.randrange(5)
        if m==0: bd[rnd.randrange(8,400)]^=1<<rnd.randrange(8)
        elif m==1: bd[rnd.randrange(8,400)]=rnd.getrandbits(8)
        elif m==2: bd=bd[:rnd.randrange(8,600)]
        elif m==3:
            for _ in range(3): bd[rnd.randrange(0,300)]^=1<<rnd.randrange(8)
        else: bd[rnd.randrange(0,len(bd))]^=1<<rnd.randrange(8)
        bd=bytes(bd); lvl=rnd.choice([1,2,3])
        try: rc=ref.CDict(bd,lvl)
        except ref.ZstdRefError: rc=None
        try: ec=util.EmuCDict(L,bd,lvl)
        except ValueError: ec=None
        cases+=1
        if (rc is None)!=(ec is None):
            key=('ref rejects' if rc is None else 'ref accepts', 'ours rejects' if ec is None else 'ours accepts'); diff[key]+=1
            if diff[key]==1: open(f'/tmp/cdict_bad_{seed}_{cases}.dict','wb').write(bd); print('first',key,'mode',m,'len',len(bd),flush=True)
        elif rc is None: both_rej+=1
        else:
            both_ok+=1
            d=rnd.choice(datas)
            a=rc.compress(d); b=ec.compress(d)
            if a!=b:
                diff['bytes']+=1
                if diff['bytes']==1: open(f'/tmp/cdict_bad_{seed}_{cases}.dict','wb').write(bd); open(f'/tmp/cdict_bad_{seed}_{cases}.bin','wb').write(d); print('first bytes diff mode',m,lvl,len(a), b if isinstance(b,int) else len(b),flush=True)
        if rc: rc.close()
        if ec: ec.close()
    print('cases',cases,'both ok',both_ok,'both reject',both_rej,'diffs',dict(diff))

(run_decode if mode == 'decode' else run_encode)()
