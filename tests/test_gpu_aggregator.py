"""GPU (-m gpu): cross-thread aggregation of per-buffer calls (zjni_aggregator_*) — many threads, one buffer per
call as zstd-jni's per-buffer natives are used, come back with the reference's bytes while the library runs a handful of batches."""
import ctypes as C
import threading

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(zj):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    zj.batch.init(0)
    return zj


def test_aggregator_batches_concurrent_callers(gpu, oracle_ref):
    L = gpu.lib()
    agg = L.zjni_createAggregator(0, 512, 20000)           # up to 512 callers per batch, the opener waits at most 20 ms
    assert agg
    nthreads, per = 48, 6
    bufs = {(t, j): gpu.synth_host([700, 4096, 20000, 65536][(t + j) % 4], 17 * t + j, 1) for t in range(nthreads) for j in range(per)}
    out, errs = {}, []

    def worker(t):
        try:
            for j in range(per):
                d = bufs[t, j]; level = 1 + (j % 3); ck = j % 2
                cap = gpu.Zstd.compressBound(len(d))
                dst = C.create_string_buffer(cap)
                r = L.zjni_aggregator_compress(agg, dst, cap, d, len(d), level, ck)
                assert not L.zjni_isError(r), L.zjni_getErrorCode(r)
                z = dst.raw[:r]
                back = C.create_string_buffer(len(d))
                r2 = L.zjni_aggregator_decompress(agg, back, len(d), z, len(z))
                assert r2 == len(d) and back.raw == d
                out[t, j] = (level, ck, z)
        except Exception as ex:                             # noqa: BLE001
            errs.append(repr(ex))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for x in th: x.start()
    for x in th: x.join()
    assert not errs, errs[:3]
    for (t, j), (level, ck, z) in out.items():
        d = bufs[t, j]
        want = oracle_ref.compress(d, level, bool(ck))
        assert z == want, (t, j, level, ck, len(d))
    calls, batches = C.c_ulonglong(), C.c_ulonglong()
    L.zjni_aggregator_stats(agg, C.byref(calls), C.byref(batches))
    assert calls.value == 2 * nthreads * per
    assert batches.value < calls.value // 3, (calls.value, batches.value)      # callers really shared launches
    # error results stay per caller: a destination that is too small fails alone
    small = C.create_string_buffer(8)
    assert L.zjni_getErrorCode(L.zjni_aggregator_compress(agg, small, 8, bufs[0, 1], len(bufs[0, 1]), 1, 0)) == 70
    assert L.zjni_getErrorCode(L.zjni_aggregator_compress(agg, small, 8, bufs[0, 1], len(bufs[0, 1]), 99, 0)) == 42
    L.zjni_freeAggregator(agg)
