"""Randomised byte-identity stress on the GPU (wave64 build, C-ABI): one batch of n random buffers per level through the lane
pipelines (plain incl. wide, explicit table sizes, dictionary, the need-gated level-3 machines in every ZJNI_NEED mode, tight
destinations), every frame compared with the reference's and decoded back on the GPU.  usage: fuzz_gpu.py <seed> <n>   TEST INFRASTRUCTURE."""
import os, sys, random, time
os.environ.setdefault("ZJNI_DEBUG_LIVE_SWITCHES", "1")    # the library caches its ZJNI_* switches per process (zj_env); this tool flips them between calls
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as e
import util
from oracle import ref
zj = e.load_package(); zj.batch.init(0)
seed = int(sys.argv[1]); n = int(sys.argv[2])
rnd = random.Random(seed)
recs = util.json_records(20000, seed=seed)
def gen(size):
    k = rnd.randrange(7)
    if k == 4:                                      # uniform small alphabets / exactly equal counts (Huffman sort edge cases)
        a = rnd.choice([2, 5, 16, 64, 98, 100, 200, 256]); base = rnd.randrange(0, 257 - a)
        return bytes(base + rnd.randrange(a) for _ in range(size))
    if k == 5:
        a = rnd.choice([9, 12, 40, 98, 130, 256]); c = rnd.choice([2, 20, 163, 164, 165, 166, 255, 256])
        v = [x for x in range(a) for _ in range(c)][:max(size, 1)]
        rnd.shuffle(v)
        return bytes(v[:size])
    if k == 6:                                      # short period with mutations
        per = os.urandom(rnd.choice([1, 2, 3, 4, 5, 7, 8, 16, 63, 64, 65, 300]))
        out = bytearray((per * (size // len(per) + 1))[:size])
        for _ in range(rnd.choice([0, 1, 5, 50])):
            if size: out[rnd.randrange(size)] = rnd.getrandbits(8)
        return bytes(out)
    if k == 0: return os.urandom(size)
    if k == 1:
        i = rnd.randrange(0, len(recs) - 2000); return b",".join(recs[i:i + 2000])[:size]
    if k == 2: return zj.synth_host(max(size, 1), rnd.randrange(1 << 20), 1)[:size]
    a = gen(size // 2); return (a + gen(size - len(a)))[:size]
def sizes(cap):
    return [min(cap, rnd.choice([rnd.randrange(0, 300), rnd.randrange(0, 5000), rnd.randrange(0, 70000), rnd.randrange(60000, 131073), 65536, 4096, 131072])) for _ in range(n)]
bad = 0; t0 = time.time()
for level in (1, 2, 3):
    datas = [gen(s) for s in sizes(131072)]
    outs = zj.compress_batch(datas, level)
    for k, (d, z) in enumerate(zip(datas, outs)):
        want = ref.compress(d, level)
        if isinstance(z, Exception) or z != want:
            bad += 1; print("MISMATCH plain", level, k, len(d), z if isinstance(z, Exception) else len(z), flush=True)
    back = zj.decompress_batch([z for z in outs if not isinstance(z, Exception)], [len(d) for d, z in zip(datas, outs) if not isinstance(z, Exception)])
    bad += sum(1 for b, d in zip(back, datas) if b != d)
    print(f"level {level}: {n} frames done, bad so far {bad}, {time.time() - t0:.0f} s", flush=True)
datas = [gen(s) for s in sizes(131072)]
for hl, cl in ((16, 15), (17, 16), (9, 12)):
    outs = zj.compress_batch(datas, 3, hash_log=hl, chain_log=cl)
    for k, (d, z) in enumerate(zip(datas, outs)):
        if isinstance(z, Exception) or z != ref.compress(d, 3, False, hl, cl):
            bad += 1; print("MISMATCH tuned", hl, cl, k, len(d), flush=True)
    print(f"tables {hl}/{cl}: done, bad so far {bad}, {time.time() - t0:.0f} s", flush=True)
# round 2: multi-block frames mixed with single-block ones (twice: a persistent workgroup meets leftovers of the same data), and
# levels 4-8 (hash-chain finders <= 16 KiB, level 4's double-fast <= 128 KiB)
WINDOW = {1: 1 << 19, 2: 1 << 20, 3: 1 << 21}
for level in (1, 2, 3):
    m = max(40, n // 40)
    datas = [gen(rnd.choice([rnd.randrange(131073, 400000), rnd.randrange(131073, 2097153), 262144, 524288, rnd.randrange(0, 131073), 65536])) for _ in range(m)]
    want = [None if len(d) > WINDOW[level] else ref.compress(d, level) for d in datas]
    for rep in range(3):
        outs = zj.compress_batch(datas, level)
        for k, (d, z, w) in enumerate(zip(datas, outs, want)):
            if w is None:
                if not (isinstance(z, Exception) and z.getErrorCode() == 201): bad += 1; print("MISMATCH multi refusal", level, k, len(d), flush=True)
            elif isinstance(z, Exception) or z != w:
                bad += 1; print("MISMATCH multi", level, rep, k, len(d), z if isinstance(z, Exception) else len(z), flush=True)
    good = [(d, z) for d, z in zip(datas, outs) if not isinstance(z, Exception)]
    back = zj.decompress_batch([z for _, z in good], [len(d) for d, _ in good])
    bad += sum(1 for b, (d, _) in zip(back, good) if b != d)
    print(f"multi-block level {level}: {m} frames x 3, bad so far {bad}, {time.time() - t0:.0f} s", flush=True)
for level in (4, 5, 6, 7, 8):
    cap = 131072
    for m in (n, 64):                                   # a large and a small batch
        datas = [gen(s) for s in sizes(cap)][:m]
        for rep in range(2):
            outs = zj.compress_batch(datas, level, checksum=bool(rep))
            for k, (d, z) in enumerate(zip(datas, outs)):
                if isinstance(z, Exception) or z != ref.compress(d, level, bool(rep)):
                    bad += 1; print("MISMATCH level", level, rep, k, len(d), z if isinstance(z, Exception) else len(z), flush=True)
        back = zj.decompress_batch(outs, [len(d) for d in datas])
        bad += sum(1 for b, d in zip(back, datas) if b != d)
    print(f"level {level}: done, bad so far {bad}, {time.time() - t0:.0f} s", flush=True)
samples = [b",".join(recs[i * 13:i * 13 + 200])[:4096] for i in range(1000)]
for dbytes in (ref.train_dict(samples, 112640), b",".join(recs[:300])):
    for level in (1, 3):
        rcd = ref.CDict(dbytes, level)
        with zj.ZstdDictCompress(dbytes, level) as cd, zj.ZstdDictDecompress(dbytes) as dd:
            cut = 16384 if level == 3 else 8192
            datas = [gen(s) for s in sizes(cut)] + [gen(s) for s in sizes(131071)[:max(20, n // 20)]]      # attach range + copy mode (beyond it, one block)
            outs = zj.compress_batch(datas, dictionary=cd)
            for k, (d, z) in enumerate(zip(datas, outs)):
                if isinstance(z, Exception) or z != rcd.compress(d):
                    bad += 1; print("MISMATCH dict", level, k, len(d), flush=True)
            back = zj.decompress_batch(outs, [len(d) for d in datas], dd)
            bad += sum(1 for b, d in zip(back, datas) if b != d)
        print(f"dict level {level}: done, bad so far {bad}, {time.time() - t0:.0f} s", flush=True)
# level 3 on the need-gated machines (zj_need.h): every mode of ZJNI_NEED on the lane pipeline regardless of batch size
os.environ["ZJNI_SPLIT_MIN"] = "1"
for machine, mode in [(a, b) for a in ("run", "lane") for b in ("0", "1", "2")]:      # the run machine (flags taken over mid-frame, beside the match kernel) and the older gated lane machine
    os.environ["ZJNI_NEED"] = mode
    if machine == "lane": os.environ["ZJNI_LANE_MACHINE"] = "0"
    else: os.environ.pop("ZJNI_LANE_MACHINE", None)
    datas = [gen(s) for s in sizes(65536)] + [bytes(rnd.randrange(16) for _ in range(rnd.randrange(4096, 65537))) for _ in range(max(8, n // 10))]
    if machine == "run":                             # ... and the wide launch's sizes with their flags (zn_flags_frame_wide)
        datas += [gen(rnd.randrange(65537, 131073)) for _ in range(max(20, n // 4))] + [bytes(rnd.randrange(16) for _ in range(rnd.randrange(65537, 131073))) for _ in range(max(8, n // 10))]
    outs = zj.compress_batch(datas, 3)
    for k, (d, z) in enumerate(zip(datas, outs)):
        want = ref.compress(d, 3)
        if isinstance(z, Exception) or z != want:
            bad += 1; print("MISMATCH need mode", mode, k, len(d), flush=True)
    print(f"ZJNI_NEED={mode} on the {machine} machine (route {zj.lib().zjni_last_route()}): done, bad so far {bad}, {time.time() - t0:.0f} s", flush=True)
del os.environ["ZJNI_NEED"]; del os.environ["ZJNI_SPLIT_MIN"]; os.environ.pop("ZJNI_LANE_MACHINE", None)
# tight destinations: capacities around each frame's size, answers (size, bytes or code) against ZSTD_compress2's for the same capacity
for level in (1, 3, 5):
    datas, caps, wants = [], [], []
    for s in sizes(16384 if level >= 5 else 131072)[:max(50, n // 4)]:
        d = gen(s) if rnd.random() < 0.7 else bytes(rnd.randrange(rnd.choice([3, 12, 48])) for _ in range(rnd.randrange(20, 400)))
        hl, cl = 0, 0
        fs = len(ref.compress(d, level, False, hl, cl))
        for cap in (fs + rnd.randrange(-2, 30), len(d) + rnd.randrange(0, 24), rnd.choice([0, 8, 17, 18, fs, fs + 8, fs + 9])):
            cap = max(cap, 0)
            try: want = ref.compress(d, level, False, hl, cl, cap=cap)
            except ref.ZstdRefError as ex: want = -ex.code
            datas.append(d); caps.append(cap); wants.append(want)
    outs = zj.compress_batch(datas, level, capacities=caps)
    for k, (want, z) in enumerate(zip(wants, outs)):
        got = -z.getErrorCode() if isinstance(z, Exception) else z
        if got != want:
            bad += 1; print("MISMATCH tight level", level, k, len(datas[k]), caps[k], flush=True)
    print(f"tight destinations level {level}: done, bad so far {bad}, {time.time() - t0:.0f} s", flush=True)
print("GPU-FUZZ", "OK" if bad == 0 else "FAILED", "bad", bad)
