"""Randomised stress of the encoder bodies (lane-serial build, tests/emu) with TIGHT destinations against the reference:
for a random input the frame is compressed into every capacity around the frame's size (and a few far below / at the
format's thresholds); sizes, error codes and bytes have to be the reference's (ZSTD_compress2 with the same capacity).
Single-block and multi-block frames, levels 1-8, checksum / no-content-size flags, dictionaries.
usage: fuzz_emu_tight.py <seed> <seconds>     TEST INFRASTRUCTURE."""
import ctypes as C
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref
import util
L = util.emu_lib()
seed = int(sys.argv[1]); budget = float(sys.argv[2])
rnd = random.Random(seed)
words = [b"the", b"quick", b"brown", b"fox", b"jumps", b"over", b"lazy", b"dog", b"lorem", b"ipsum", b"dolor", b"sit", b"amet", b"zstd", b"frame", b"block"]
def text(n):
    out = bytearray()
    while len(out) < n: out += rnd.choice(words) + b" "
    return bytes(out[:n])
def lowent(n, a=16):
    out = bytearray()
    for i in range(n):
        if i and rnd.random() < 7 / 8: out.append(out[i - rnd.randrange(1, min(i, 64) + 1)])
        else: out.append(rnd.randrange(a))
    return bytes(out)
def skew(n):
    a = rnd.choice([2, 3, 5, 16, 40, 100, 200, 256]); p = rnd.choice([0.3, 0.6, 0.9])
    return bytes((0 if rnd.random() < p else rnd.randrange(a)) for _ in range(n))
def gen(n):
    k = rnd.randrange(7)
    if k == 0: return text(n)
    if k == 1: return lowent(n, rnd.choice([2, 16, 200]))
    if k == 2: return bytes(rnd.getrandbits(8) for _ in range(n))
    if k == 3: return skew(n)
    if k == 4:
        per = bytes(rnd.getrandbits(8) for _ in range(rnd.choice([1, 2, 3, 7, 64, 300])))
        out = bytearray((per * (n // len(per) + 1))[:n])
        for _ in range(rnd.choice([0, 1, 5, 50])):
            if n: out[rnd.randrange(n)] = rnd.getrandbits(8)
        return bytes(out)
    if k == 5: return bytes([rnd.randrange(256)]) * n
    a = gen(n // 2); return (a + gen(n - len(a)))[:n]
def emu(fn, data, cap, level_word):
    dst = C.create_string_buffer(max(cap, 1) + 64)
    r = fn(data, len(data), dst, cap, level_word)
    if r >= (1 << 63): return -((1 << 64) - r)
    return dst.raw[:r]
def reference(data, cap, level, checksum, content_size, hl, cl):
    try: return ref.compress(data, level, checksum=checksum, hash_log=hl, chain_log=cl, content_size=content_size, cap=cap)
    except ref.ZstdRefError as ex: return -ex.code
# dictionaries: a trained one (tables + content) and a raw-content one, digested at levels 1-3 on both sides
recs = util.json_records(20000, seed=seed)
samples = [b",".join(recs[i * 13:i * 13 + 200])[:4096] for i in range(1000)]
DICTS = []
for dbytes in (ref.train_dict(samples, 20000), text(3000) + lowent(5000)):
    for lvl in (1, 2, 3):
        DICTS.append((ref.CDict(dbytes, lvl), util.EmuCDict(L, dbytes, lvl)))
def dict_case():
    rc, ec = rnd.choice(DICTS)
    n = rnd.choice([0, 1, 7, 100, 1000, 4000, 8000, 9000, 17000, 40000]) if rnd.random() < 0.5 else rnd.randrange(0, rnd.choice([300, 5000, 40000]) + 1)
    k = rnd.randrange(4)
    if k == 0:
        i = rnd.randrange(0, len(recs) - 3000); data = b",".join(recs[i:i + 3000])[:n]
    else: data = gen(n)
    checksum = rnd.random() < 0.3; did = rnd.random() >= 0.2
    full = rc.compress(data, checksum, did); fs = len(full)
    caps = set(range(max(0, fs - 3), fs + 40)) | {0, 5, 9, 17, 18, 19, fs + 64, n, n + 3, n + 9, n + 14, n + 20}
    out = []
    for cap in sorted(caps):
        try: want = rc.compress(data, checksum, did, cap=cap)
        except ref.ZstdRefError as ex: want = -ex.code
        out.append((cap, want, ec.compress(data, checksum, did, cap=cap), full))
    return n, out
SIZES = [0, 1, 6, 7, 8, 20, 63, 64, 65, 100, 255, 256, 300, 1000, 1023, 1024, 1025, 4096, 9000, 16384, 20000, 70000]
t0 = time.time(); cases = 0; bad = 0; kinds = {"ok": 0, "raw_or_other": 0, "err": 0, "err_although_the_frame_fits": 0}
while time.time() - t0 < budget:
    if rnd.random() < 0.2:
        n, out = dict_case()
        for cap, want, got, full in out:
            cases += 1
            kinds["err" if isinstance(want, int) else ("ok" if want == full else "raw_or_other")] += 1
            if isinstance(want, int) and cap >= len(full): kinds["err_although_the_frame_fits"] += 1
            if got != want:
                bad += 1
                if bad <= 15:
                    d = lambda x: x if isinstance(x, int) else "%d bytes" % len(x)
                    print("MISMATCH (dictionary) n=%d cap=%d (frame %d): ref %s, here %s" % (n, cap, len(full), d(want), d(got)), flush=True)
        continue
    multi = rnd.random() < 0.15
    if multi:
        n = rnd.choice([131073, 140000, 200000, 300000]); level = rnd.choice([1, 2, 3])
    else:
        n = rnd.choice(SIZES) if rnd.random() < 0.5 else rnd.randrange(0, rnd.choice([200, 2000, 20000, 131072]) + 1)
        level = rnd.randrange(1, 9)
        if level >= 5 and n > 30000 and rnd.random() < 0.8: level = rnd.randrange(1, 5)          # (one lane parses the row finder: slow here)
    data = gen(n)
    if not multi and rnd.random() < 0.2:
        # barely compressible small inputs: only below 384 bytes can a block be both compressible (ZSTD_minGain) and, in a destination that
        # holds it raw, out of room compressed — the reference then emits the raw block
        n = rnd.randrange(20, 400); a = rnd.choice([3, 6, 12, 24, 48, 100]); data = bytearray(rnd.randrange(a) for _ in range(n))
        for _ in range(rnd.choice([0, 1, 2, 4])):
            ln = rnd.randrange(4, 12); at = rnd.randrange(0, max(1, n - 2 * ln)); to = rnd.randrange(at + ln, max(at + ln + 1, n - ln + 1))
            data[to:to + ln] = data[at:at + ln]
        data = bytes(data[:n])
    checksum = rnd.random() < 0.3; content_size = rnd.random() >= 0.2
    hl = cl = 0
    if level == 3 and 8192 < n <= 131072: hl, cl = 14, 13                                       # the library's level-3 tables for these sizes
    word = level | (int(checksum) << 8) | ((0 if content_size else 1) << 9)
    if n > 131072 or (level >= 4 and (n > 16384 or rnd.random() < 0.5)):
        fn = L.emu_compress_multi      # zj_encode_multi_kernel's routes
        if n > 131072 and rnd.random() < 0.5: word |= 0x8000                                           # ... and zj_encode_pipe_kernel's pair of roles (round 6)
    elif level >= 4: fn = L.emu_compress_chain                                                      # chain parser per lane + entropy stage on its records
    else: fn = L.emu_compress if rnd.random() < 0.6 else L.emu_compress_split                        # fused / lane match finder + entropy stage
    full = reference(data, None, level, checksum, content_size, hl, cl)
    fs = len(full)
    caps = set(range(max(0, fs - 3), fs + 40)) | {0, 1, 5, 8, 9, 17, 18, 19, 24, fs // 2, fs + 64, fs + 100, n, n + 3, n + 9, n + 12, n + 18, n + 25}
    if multi: caps = set(rnd.sample(sorted(caps), 12)) | {fs, fs + 7, fs + 8, fs + 9}
    for cap in sorted(c for c in caps if c >= 0):
        want = reference(data, cap, level, checksum, content_size, hl, cl)
        got = emu(fn, data, cap, word)
        cases += 1
        if isinstance(want, int):
            kinds["err"] += 1
            if cap >= fs: kinds["err_although_the_frame_fits"] += 1
        elif want == full: kinds["ok"] += 1
        else: kinds["raw_or_other"] += 1
        if got != want:
            bad += 1
            if bad <= 15:
                d = lambda x: x if isinstance(x, int) else "%d bytes" % len(x)
                print("MISMATCH n=%d level=%d ck=%d cs=%d cap=%d (frame %d): ref %s, here %s  fn=%s" % (n, level, checksum, content_size, cap, fs, d(want), d(got), fn.__name__), flush=True)
                if bad == 1:
                    open("/tmp/tight_case.bin", "wb").write(data)
print("cases=%d mismatches=%d (reference outcomes: %s) seed=%d" % (cases, bad, kinds, seed))
sys.exit(1 if bad else 0)
