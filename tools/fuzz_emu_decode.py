"""Randomised stress of the three-stage decoder bodies (lane-serial build) against the reference's frames (levels 1-9, with and
without dictionary / checksum, content up to 300 KB) and, for corrupted frames (bit flips, byte stores, truncation), against
the answer of the reference's portable decoder loops (oracle/_ref/libzstd_ref_portable.so) on both pipelines: the same bytes,
or a refusal with the same error code.
usage: fuzz_emu_decode.py <seed> <seconds>   TEST INFRASTRUCTURE."""
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
def gen(n):
    k = rnd.randrange(4)
    if k==0: return bytes(rnd.getrandbits(8) for _ in range(n))
    if k==1:
        i=rnd.randrange(0,len(recs)-3000); return b",".join(recs[i:i+3000])[:n]
    if k==2: return zj.synth_host(max(n,1), rnd.randrange(1<<20), 1)[:n]
    a=gen(n//2); return (a+gen(n-len(a)))[:n]
def gen_edge(n):
    """frames with the block / literal / table modes ordinary data rarely produces: rle and raw blocks, rle literals, predefined
    and rle tANS tables, overlapping matches with tiny offsets, very long matches"""
    k = rnd.randrange(5)
    if k == 0:
        a = rnd.choice([2, 3, 5, 16, 17, 64, 100, 200, 256]); base = rnd.randrange(0, 257 - a)
        return bytes(base + rnd.randrange(a) for _ in range(n))
    if k == 1:
        b = rnd.randrange(256); p = rnd.choice([0.0, 0.001, 0.01, 0.1])
        return bytes(b if rnd.random() >= p else rnd.randrange(256) for _ in range(n))
    if k == 2:
        per = bytes(rnd.getrandbits(8) for _ in range(rnd.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 63, 64, 65, 300])))
        out = bytearray((per * (n // len(per) + 1))[:n])
        for _ in range(rnd.choice([0, 1, 5, 50])):
            if n: out[rnd.randrange(n)] = rnd.getrandbits(8)
        return bytes(out)
    if k == 3:
        a = rnd.choice([9, 12, 40, 130, 256]); c = rnd.choice([1, 2, 20, 163, 164, 165, 166, 255, 256])
        v = [x for x in range(a) for _ in range(c)][:max(n, 1)]
        rnd.shuffle(v)
        return bytes(v[:n])
    a = gen(n // 3); b = gen_edge(n // 3)
    return (a + b + gen(n - len(a) - len(b)))[:n]
EDGE_SIZES = [0, 1, 5, 6, 7, 8, 9, 12, 13, 62, 63, 64, 65, 255, 256, 257, 1022, 1023, 1024, 1025, 4095, 4096, 4097, 16383, 16384, 16385, 65535, 65536, 65537, 131071, 131072, 131073, 262144]
samples=[b",".join(recs[i*13:i*13+200])[:4096] for i in range(1000)]
dic = ref.train_dict(samples, 60000)
t0=time.time(); cases=0; bad=0; corrupted=0; lax=0
while time.time()-t0 < budget:
    n = rnd.choice([rnd.randrange(0,5000), rnd.randrange(0,70000), rnd.randrange(60000,131073), 131072, rnd.randrange(131073, 300000)])
    if rnd.random() < 0.3: n = rnd.choice(EDGE_SIZES)
    d = (gen_edge if rnd.random() < 0.5 else gen)(n); lvl = rnd.choice([1,3,5,9,19])
    usedDict = rnd.random() < 0.3
    if usedDict:
        z = ref.compress_using_dict(d, dic, lvl); out = util.emu_decompress_dict(L, z, len(d), dic, split=True)
    elif rnd.random() < 0.25:      # streamed frame: no content size in the header, blocks closed by flushes (ZstdOutputStream's output)
        z = ref.compress_stream(d, lvl, rnd.random()<0.3, chunk=rnd.choice([1000, 30000, 200000]), flush_every=rnd.choice([0, 1, 3]))
        out, used = util.emu_decompress_split(L, z, len(d))
    else:
        z = ref.compress(d, lvl, checksum=rnd.random()<0.3); out, used = util.emu_decompress_split(L, z, len(d))
    cases+=1
    if out != d:
        bad+=1; print('MISMATCH', n, lvl, out if isinstance(out,int) else 'bytes', flush=True)
    # corrupted (bit flips, byte stores, truncation): both pipelines answer what the reference's portable decoder loops answer
    # (oracle.ref.decompress_portable — bytes or refusal); `lax` counts the frames its stock x86-64 build answers differently
    if len(z) > 12 and rnd.random() < 0.6:
        for _ in range(4):
            zb = bytearray(z); m = rnd.randrange(6)
            if m <= 2: zb[rnd.randrange(4, len(zb))] ^= 1 << rnd.randrange(8)
            elif m == 3: zb[rnd.randrange(4, len(zb))] = rnd.getrandbits(8)
            elif m == 4:
                for _ in range(3): zb[rnd.randrange(4, len(zb))] ^= 1 << rnd.randrange(8)
            else: zb = zb[:rnd.randrange(5, len(zb))]
            zb = bytes(zb); corrupted += 1
            cap = len(d) if rnd.random() < 0.85 else rnd.randrange(0, len(d) + 1)     # sometimes an undersized destination
            try: want = ref.decompress_portable(zb, cap, dic if usedDict else None)
            except ref.ZstdRefError as e: want = -e.code                                # refusals must carry the same error code
            try: stock = ref.decompress_using_dict(zb, dic, cap) if usedDict else ref.decompress(zb, cap)
            except ref.ZstdRefError as e: stock = -e.code
            lax += (stock != want)
            if usedDict: a = util.emu_decompress_dict(L, zb, cap, dic, split=True); b = util.emu_decompress_dict(L, zb, cap, dic)
            else: a = util.emu_decompress_split(L, zb, cap)[0]; b = util.emu_decompress(L, zb, cap)
            for nm, x in (('split', a), ('fused', b)):
                if x != want:
                    bad += 1; open(f'/tmp/fuzz_dec_bad_{seed}_{cases}.zst', 'wb').write(zb)
                    print('REFDIFF', nm, n, lvl, 'dict' if usedDict else '', 'portable', want if isinstance(want, int) else len(want), 'ours', x if isinstance(x, int) else len(x), flush=True)
print('seed',seed,'cases',cases,'corrupted',corrupted,'stock_build_differs',lax,'bad',bad,flush=True)
