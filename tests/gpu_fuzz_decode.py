"""Differential decode fuzz through the C-ABI on the GPU (wave64 build): valid frames of the reference (levels 1-19, multi-block,
multi-frame, streamed, with / without dictionary, checksum) and DAMAGED ones (bit flips, byte stores, truncation, undersized
destinations) must give what the reference's portable decoder loops give — the same bytes, or a refusal with the same code —
on both decode pipelines.  Destination slots of a batch sit back to back in one HBM blob, so every real slot is followed by a
64-byte guard slot that must come back untouched.  Shared by tests/test_gpu_zz_fuzz_decode.py (bounded) and
tools/fuzz_gpu_decode.py (open-ended).  TEST INFRASTRUCTURE."""
import os
import random

GUARD = 64


def make_cases(zj, ref, seed, count, dictionary=None):
    """[(frame bytes, destination capacity, expected bytes | -code)]"""
    import util
    rnd = random.Random(seed)
    recs = util.json_records(6000, seed=seed)

    def gen(n):
        k = rnd.randrange(6)
        if k == 0: return os.urandom(n)
        if k == 1:
            i = rnd.randrange(0, len(recs) - 3000); return b",".join(recs[i:i + 3000])[:n]
        if k == 2: return zj.synth_host(max(n, 1), rnd.randrange(1 << 20), 1)[:n]
        if k == 3:
            per = os.urandom(rnd.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 63, 64, 65, 300]))
            out = bytearray((per * (n // len(per) + 1))[:n])
            for _ in range(rnd.choice([0, 1, 5, 50])):
                if n: out[rnd.randrange(n)] = rnd.getrandbits(8)
            return bytes(out)
        if k == 4:
            a = rnd.choice([2, 3, 5, 16, 64, 200, 256]); base = rnd.randrange(0, 257 - a)
            return bytes(base + rnd.randrange(a) for _ in range(n))
        a = gen(n // 2); return (a + gen(n - len(a)))[:n]

    def answer(z, cap):
        try:
            return ref.decompress_portable(z, cap, dictionary)
        except ref.ZstdRefError as e:
            return -e.code

    cases = []
    while len(cases) < count:
        n = rnd.choice([rnd.randrange(0, 300), rnd.randrange(0, 5000), rnd.randrange(0, 70000), rnd.randrange(60000, 131073), 131072, 65536,
                        rnd.randrange(131073, 300000)])
        lvl = rnd.choice([1, 1, 3, 3, 5, 9]) if n > 20000 else rnd.choice([1, 3, 5, 9, 19])
        d = gen(n)
        shape = rnd.randrange(5)
        if dictionary is not None: z = ref.compress_using_dict(d, dictionary, lvl)
        elif shape == 0: z = ref.compress_stream(d, lvl, rnd.random() < 0.3, chunk=rnd.choice([1000, 30000, 200000]), flush_every=rnd.choice([0, 1, 3]))
        elif shape == 1:                                   # two frames back to back (ZSTD_decompress takes any number)
            cut = rnd.randrange(0, n + 1); z = ref.compress(d[:cut], lvl, rnd.random() < 0.3) + ref.compress(d[cut:], rnd.choice([1, 3]))
        else: z = ref.compress(d, lvl, checksum=rnd.random() < 0.3)
        cases.append((z, len(d), d))
        if len(z) > 12:
            for _ in range(rnd.choice([1, 2, 3])):
                zb = bytearray(z); m = rnd.randrange(6)
                if m <= 2: zb[rnd.randrange(4, len(zb))] ^= 1 << rnd.randrange(8)
                elif m == 3: zb[rnd.randrange(4, len(zb))] = rnd.getrandbits(8)
                elif m == 4:
                    for _ in range(3): zb[rnd.randrange(4, len(zb))] ^= 1 << rnd.randrange(8)
                else: zb = zb[:rnd.randrange(5, len(zb))]
                zb = bytes(zb)
                cap = len(d) if rnd.random() < 0.8 else rnd.randrange(0, len(d) + 1)     # sometimes an undersized destination
                cases.append((zb, cap, answer(zb, cap)))
    return cases[:count]


def run_cases(zj, cases, dictionary_obj=None, split_min=None):
    """one device batch with guard slots; returns the list of (index, description) that differ"""
    import numpy as np
    import torch
    if split_min is not None:
        os.environ["ZJNI_DSPLIT_MIN"] = str(split_min)
    B = zj.batch
    dev = torch.device("cuda")
    m = len(cases)
    src_sizes = np.zeros(2 * m, dtype=np.int64); dst_sizes = np.zeros(2 * m, dtype=np.int64)
    for i, (z, cap, _) in enumerate(cases):
        src_sizes[2 * i] = len(z); dst_sizes[2 * i] = cap; dst_sizes[2 * i + 1] = GUARD          # slot 2i+1: empty source, guard destination
    soff = np.zeros(2 * m + 1, dtype=np.int64); soff[1:] = np.cumsum(src_sizes)
    doff = np.zeros(2 * m + 1, dtype=np.int64); doff[1:] = np.cumsum(dst_sizes)
    sblob = np.frombuffer(b"".join(z for z, _, _ in cases) + b"\0" * 16, dtype=np.uint8).copy()
    d_src = torch.from_numpy(sblob).to(dev)
    d_dst = torch.full((int(doff[-1]) + 16,), 0xA5, dtype=torch.uint8, device=dev)
    res = B.decompress(d_src, torch.from_numpy(soff).to(dev), d_dst, torch.from_numpy(doff).to(dev), dictionary=dictionary_obj)
    torch.cuda.synchronize()
    r = res.cpu().numpy(); out = d_dst.cpu().numpy()
    bad = []
    for i, (z, cap, want) in enumerate(cases):
        got = int(r[2 * i]); o = int(doff[2 * i])
        if isinstance(want, int):
            if got != want: bad.append((i, f"want code {want}, got {got}"))
        elif got != len(want) or out[o:o + got].tobytes() != want:
            bad.append((i, f"want {len(want)} bytes, got {got}"))
        g = int(doff[2 * i + 1])
        if not (out[g:g + GUARD] == 0xA5).all(): bad.append((i, "guard bytes behind the destination slot overwritten"))
        if int(r[2 * i + 1]) != 0: bad.append((i, f"guard slot result {int(r[2 * i + 1])}"))
    return bad
