#!/usr/bin/env python3
"""Throughput of the device form of the stream route: n streams closing together, one launch
(zjni_compress_stream_batch_device, include/zjni_amd.h).

Streams are runs of the benchmark generator's 64 KiB mixed-entropy buffers, resident in HBM before the timed region; every stream is
closed (final), some configurations carry flush positions.  Timed with events on the launch stream, best of `reps`; a sample of the
streams is compared byte for byte with the reference's ZSTD_compressStream2 over the same writes and flushes (oracle/_ref — checker
only), and the same sample is timed on one host thread as the CPU figure beside it.

usage: bench_stream_batch.py [out.json]"""
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import importlib
    import torch
    zj = importlib.import_module("zstd-jni_amd")
    from oracle import ref
    zj.batch.init(0)
    L = zj.lib()
    dev = "cuda"
    configs = [  # (streams, bytes per stream, level, flush every this many bytes or 0)
        (4096, 256 << 10, 3, 0), (1024, 1 << 20, 3, 0), (1024, 1 << 20, 3, 256 << 10), (8192, 128 << 10, 1, 0), (16384, 64 << 10, 3, 0), (256, 2 << 20, 3, 0)]
    reps = 3
    lines = []
    for n, size, level, fevery in configs:
        blob = zj.batch.synth(n * size // 65536, 65536, 1000)
        off = zj.batch.uniform_offsets(n, size, dev)
        cap = size + (size >> 8) + 4096 + 64 * ((size // fevery if fevery else 0) + 4)
        doff = zj.batch.uniform_offsets(n, cap, dev)
        dst = torch.empty(n * cap, dtype=torch.uint8, device=dev)
        res = torch.zeros(n, dtype=torch.int64, device=dev)
        flushes = list(range(fevery, size + 1, fevery)) if fevery else []       # flush() after every write of `fevery` bytes, the last one included
        if flushes:
            fat = torch.tensor(flushes * n, dtype=torch.int32, device=dev)
            foff = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * len(flushes)
        best = None
        for _ in range(reps + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = L.zjni_compress_stream_batch_device(blob.data_ptr(), off.data_ptr(), dst.data_ptr(), doff.data_ptr(), res.data_ptr(), n, level, 0,
                                                    fat.data_ptr() if flushes else None, foff.data_ptr() if flushes else None, None, torch.cuda.current_stream().cuda_stream)
            b.record(); torch.cuda.synchronize()
            assert r == 0, r
            ms = a.elapsed_time(b)
            best = ms if best is None or ms < best else best       # (the first pass allocates scratch)
        sizes = res.cpu().tolist()
        if not all(0 < s < cap for s in sizes):
            print(json.dumps({"streams": n, "stream_bytes": size, "level": level, "declined": [s for s in sizes if not 0 < s < cap][:4]}), flush=True)
            continue
        # parity on a sample, and the same sample on one host thread
        sample = sorted(set([0, 1, n // 2, n - 1] + [(i * 2654435761) % n for i in range(4)]))
        host = blob.view(n, size)[sample].cpu().numpy()
        out = dst.view(n, cap)[sample].cpu().numpy()
        cpu_s = 0.0
        for j, i in enumerate(sample):
            d = host[j].tobytes()
            t0 = time.perf_counter()
            want = ref.compress_stream(d, level, False, chunk=fevery if fevery else size, flush_every=1 if fevery else 0)
            cpu_s += time.perf_counter() - t0
            got = out[j][:sizes[i]].tobytes()
            assert got == want, ("stream", i, len(got), len(want))
        gib = n * size / 2**30
        line = {"streams": n, "stream_bytes": size, "level": level, "flush_every": fevery, "ms": round(best, 2), "GiBps": round(gib / (best / 1e3), 2),
                "ratio": round(n * size / sum(sizes), 3), "parity_sample": len(sample), "cpu_one_thread_GiBps": round(len(sample) * size / 2**30 / cpu_s, 3),
                "build_stamp": zj.build_stamp()}
        print(json.dumps(line), flush=True)
        lines.append(line)
        del blob, dst, res
        torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            for l in lines:
                f.write(json.dumps(l) + "\n")


if __name__ == "__main__":
    main()
