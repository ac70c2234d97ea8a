"""GPU (-m gpu): the multi-device host entries (zjni_*_batch_multi: one process, a thread per device) and the
scratch budget (zjni_set_scratch_limit / zjni_release_scratch).  With one GPU on the box the multi entries are driven with the
same ordinal twice — the sharding, the per-device threads and the peer gather are the code that runs with eight; the test with two
distinct devices is skipped where there is one."""
import ctypes as C
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(zj):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    zj.batch.init(0)
    return zj


def _arrays(bufs, caps):
    n = len(bufs)
    keep = [C.create_string_buffer(bytes(b), max(len(b), 1)) for b in bufs]
    outs = [C.create_string_buffer(max(c, 1)) for c in caps]
    return (keep, outs, (C.c_void_p * n)(*[C.addressof(k) for k in keep]), (C.c_void_p * n)(*[C.addressof(o) for o in outs]),
            (C.c_size_t * n)(*[len(b) for b in bufs]), (C.c_size_t * n)(*caps), (C.c_size_t * n)())


def _multi(gpu, bufs, level, devices, mode, checksum=False):
    L = gpu.lib()
    caps = [gpu.Zstd.compressBound(len(b)) for b in bufs]
    keep, outs, sp, dp, ss, dc, res = _arrays(bufs, caps)
    dv = (C.c_int * len(devices))(*devices)
    r = L.zjni_compress_batch_multi(sp, ss, dp, dc, res, len(bufs), level, int(checksum), dv, len(devices), mode)
    assert not L.zjni_isError(r), L.zjni_getErrorCode(r)
    frames = [outs[i].raw[:res[i]] for i in range(len(bufs))]
    keep2, outs2, sp2, dp2, ss2, dc2, res2 = _arrays(frames, [len(b) for b in bufs])
    r = L.zjni_decompress_batch_multi(sp2, ss2, dp2, dc2, res2, len(bufs), dv, len(devices))
    assert not L.zjni_isError(r)
    back = [outs2[i].raw[:res2[i]] for i in range(len(bufs))]
    return frames, back


@pytest.mark.parametrize("mode", [0, 1])
def test_multi_entries_shard_and_reassemble(gpu, oracle_ref, mode):
    rnd = random.Random(3 + mode)
    bufs = [gpu.synth_host(rnd.choice([0, 100, 4096, 20000, 65536, 65536, 100000]) or 1, rnd.randrange(0, 100000), 1) for _ in range(600)]
    bufs[7] = b""
    for devices in ([0], [0, 0], [0, 0, 0]):
        frames, back = _multi(gpu, bufs, 1, devices, mode, checksum=bool(mode))
        for i, (b, f, o) in enumerate(zip(bufs, frames, back)):
            assert f == oracle_ref.compress(b, 1, bool(mode)), (devices, i, len(b))
            assert o == b, (devices, i)
    L = gpu.lib()
    bad = (C.c_int * 1)(99)
    keep, outs, sp, dp, ss, dc, res = _arrays(bufs[:2], [1000, 1000])
    assert L.zjni_getErrorCode(L.zjni_compress_batch_multi(sp, ss, dp, dc, res, 2, 1, 0, bad, 1, 0)) == 200      # no such device


def test_two_devices_do_not_share_state(gpu, oracle_ref):
    L = gpu.lib()
    if L.zjni_device_count() < 2:
        pytest.skip("one GPU on this box")
    bufs = [gpu.synth_host(65536, i, 1) for i in range(5000)]
    frames, back = _multi(gpu, bufs, 3, [0, 1], 0)
    f1, b1 = _multi(gpu, bufs, 3, [1, 0], 1)
    for i, b in enumerate(bufs):
        assert frames[i] == f1[i] == oracle_ref.compress(b, 3) and back[i] == b1[i] == b, i


def test_scratch_limit_bounds_the_library(gpu, oracle_ref):
    import torch
    L, B = gpu.lib(), gpu.batch
    n, size = 20000, 65536
    src = B.synth(n, size, 0, "cuda"); off = B.uniform_offsets(n, size, "cuda")
    bound = gpu.Zstd.compressBound(size)
    comp = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); coff = B.uniform_offsets(n, bound, "cuda")
    assert L.zjni_release_scratch() == 0 and L.zjni_scratch_bytes() == 0
    free = B.compress(src, off, comp, coff, 3).clone(); torch.cuda.synchronize()
    big = L.zjni_scratch_bytes()
    assert big > (6 << 30)                                     # sized for 288 GB: n x (tables + records + flags)
    # ... and NOT the wide slice (list B: frames of 64 KiB + 1 .. 128 KiB; rounds 1-5 allocated its 20 000 x 1.1 MiB here whatever the batch held): round 6 allocates it
    # when a call has such frames.  n x (384 KiB tables + 320 KiB records + 64 KiB flags) and small change:
    assert big < n * (800 << 10), big
    n2 = 8200                                                  # (>= ZJNI_L3_WAVE_MAX: the lane pipeline) a batch WITH wide frames: the slice appears, sized for it, and the frames are the reference's
    src2 = B.synth(n2, 131072, 7, "cuda"); off2 = B.uniform_offsets(n2, 131072, "cuda"); bound2 = gpu.Zstd.compressBound(131072)
    compw = torch.empty(n2 * bound2, dtype=torch.uint8, device="cuda"); coffw = B.uniform_offsets(n2, bound2, "cuda")
    szw = B.compress(src2, off2, compw, coffw, 3); torch.cuda.synchronize()
    assert L.zjni_scratch_bytes() > big + n2 * (1 << 20)
    hs = src2[:8 * 131072].cpu().numpy().tobytes(); hc = compw[:8 * bound2].cpu().numpy().tobytes()
    for i in range(8):
        assert hc[i * bound2:i * bound2 + int(szw[i])] == oracle_ref.compress(hs[i * 131072:(i + 1) * 131072], 3), i
    del src2, compw
    again = B.compress(src, off, comp, coff, 3); torch.cuda.synchronize()          # with the slice in place a batch without wide frames goes through as before
    assert torch.equal(again, free)
    assert L.zjni_release_scratch() == 0 and L.zjni_scratch_bytes() == 0
    try:
        assert L.zjni_set_scratch_limit(1) == (4 << 30)        # raised to the minimum
        assert L.zjni_set_scratch_limit(6 << 30) == (6 << 30)
        comp2 = torch.empty_like(comp)
        lim = B.compress(src, off, comp2, coff, 3); torch.cuda.synchronize()
        assert L.zjni_scratch_bytes() <= (6 << 30)
        assert torch.equal(lim, free) and torch.equal(comp2, comp)          # slicing changes nothing but the time
        packed, poff = B.pack(lim, comp2, coff)
        back = torch.empty(n * size, dtype=torch.uint8, device="cuda")
        dsz = B.decompress(packed, poff, back, off); torch.cuda.synchronize()
        assert bool((dsz == size).all()) and torch.equal(back, src)
        assert L.zjni_scratch_bytes() <= (6 << 30)
    finally:
        L.zjni_set_scratch_limit(0)
        assert L.zjni_release_scratch() == 0


def test_gpu_two_host_batches_in_flight(gpu, oracle_ref):
    """zjni_compress_batch_begin / zjni_decompress_batch_begin (round 5): jobs through the device's two staging slots at once, a third waiting for a slot;
    frames byte-identical to the reference's, every batch back to its input."""
    import ctypes as C
    L = gpu.lib()
    sets = []
    for j in range(3):
        bufs = [gpu.synth_host(sz, 100 * j + k, 1) for k, sz in enumerate([65536, 30000, 4096, 100, 65536, 0, 12345] * 40)]
        n = len(bufs)
        caps = [gpu.Zstd.compressBound(len(b)) for b in bufs]
        srcs = [C.create_string_buffer(b, max(len(b), 1)) for b in bufs]; dsts = [C.create_string_buffer(c) for c in caps]
        sp = (C.c_void_p * n)(*[C.addressof(x) for x in srcs]); dp = (C.c_void_p * n)(*[C.addressof(x) for x in dsts])
        ss = (C.c_size_t * n)(*[len(b) for b in bufs]); dc = (C.c_size_t * n)(*caps); res = (C.c_size_t * n)()
        sets.append((bufs, srcs, dsts, sp, dp, ss, dc, res, n))
    jobs = [L.zjni_compress_batch_begin(s[3], s[5], s[4], s[6], s[7], s[8], 3, 1) for s in sets]          # three at once: two slots
    assert all(jobs)
    for j in jobs:
        r = L.zjni_batch_finish(j); assert not L.zjni_isError(r), r
    backs = []
    for bufs, srcs, dsts, sp, dp, ss, dc, res, n in sets:
        for k in range(n):
            assert not L.zjni_isError(res[k]), (k, res[k])
            assert dsts[k].raw[:res[k]] == oracle_ref.compress(bufs[k], 3, True), k
        outs = [C.create_string_buffer(max(len(b), 1)) for b in bufs]
        op = (C.c_void_p * n)(*[C.addressof(x) for x in outs]); oc = (C.c_size_t * n)(*[len(b) for b in bufs]); cs = (C.c_size_t * n)(*[res[k] for k in range(n)]); r2 = (C.c_size_t * n)()
        backs.append((outs, op, oc, cs, r2))
    jobs = [L.zjni_decompress_batch_begin(s[4], b[3], b[1], b[2], b[4], s[8]) for s, b in zip(sets, backs)]
    assert all(jobs)
    for j in jobs:
        r = L.zjni_batch_finish(j); assert not L.zjni_isError(r), r
    for (bufs, *_), (outs, op, oc, cs, r2) in zip(sets, backs):
        for k, b in enumerate(bufs):
            assert r2[k] == len(b) and outs[k].raw[:len(b)] == b, k


def test_gpu_pack_makes_its_own_offsets(gpu):
    """zjni_pack_batch_device2 (round 5): the exclusive scan of the sizes on the device (error results count as 0), then the frames back to back — equal to torch's scan + the plain pack"""
    import torch
    B = gpu.batch
    for n, size in ((1, 100), (1000, 5000), (4097, 3000), (70001, 300)):
        src = B.synth(n, size, 3, "cuda"); src_off = B.uniform_offsets(n, size, "cuda")
        bound = gpu.Zstd.compressBound(size)
        comp = torch.empty(n * bound, dtype=torch.uint8, device="cuda"); comp_off = B.uniform_offsets(n, bound, "cuda")
        csz = B.compress(src, src_off, comp, comp_off, 1)
        csz[n // 3] = -70                                             # an error result in the middle: skipped, counts as 0
        want_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda"); want_off[1:] = torch.cumsum(csz.clamp(min=0), 0)
        a, a_off = B.pack(csz, comp, comp_off)                         # torch scan + zjni_pack_batch_device
        out = torch.zeros(int(want_off[-1].item()) + 64, dtype=torch.uint8, device="cuda"); off = torch.full((n + 1,), -1, dtype=torch.int64, device="cuda")
        b, b_off = B.pack(csz, comp, comp_off, out=out, out_off=off)   # zjni_pack_batch_device2
        torch.cuda.synchronize()
        assert torch.equal(b_off, want_off) and torch.equal(a_off, want_off), (n, size)
        assert torch.equal(b[:a.numel()], a), (n, size)
