#!/usr/bin/env python3
"""bench.py — batched zstd compress + decompress of the BASELINE.json metric config on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by the driver through torch.distributed.run, one rank per GPU)

Workload (config.workload): 65,536 x 64 KiB mixed-entropy synthetic buffers,
level 3, generated in HBM.  One step = one GPU compress pass over the batch + one GPU decompress
pass over the frames it produced (inputs resident in HBM when the timed region starts).  At N>1
every rank runs the same per-GPU batch on different buffer indices (weak scaling) and the packed
compressed output is gathered to rank 0 over RCCL inside the step.

value = uncompressed bytes through a full compress->decompress pass per second, whole job.
roofline = dominant kernel's algorithmic bytes (S + C per buffer) / its HIP-event time.
cpu_baseline = the reference's own libzstd (oracle/_ref) on this box's host cores, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

GIB = float(1 << 30)
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--buffers", type=int, default=65536)
    ap.add_argument("--size", type=int, default=65536)
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=4096, help="buffers in the CPU baseline sample")
    ap.add_argument("--verify-sample", type=int, default=512, help="GPU frames re-decoded by the CPU reference")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL gather of compressed output")
    ap.add_argument("--skip-cpu", action="store_true", help="skip the CPU reference legs (verification sample + cpu_baseline), e.g. under a profiler")
    ap.add_argument("--decode-only", action="store_true",
                    help="diagnostic: time only the decode kernel on reference-compressed frames (unique set of --unique buffers, replicated)")
    ap.add_argument("--unique", type=int, default=4096)
    return ap.parse_args()


def decode_only(a, zj, dev, n, size, level):
    """Diagnostic leg used while bringing kernels up: reference-made frames -> GPU decode."""
    import numpy as np
    from oracle import port
    B = zj.batch
    u = min(a.unique, n)
    raw = zj.synth_host(size, 0, u)
    frames = port.compress_many(raw, size, level, os.cpu_count() or 1)
    fs = np.array([len(f) for f in frames], dtype=np.int64)
    rep = (n + u - 1) // u
    blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
    d_blob = torch.from_numpy(np.tile(blob, rep)).to(dev)
    sizes = np.tile(fs, rep)[:n]
    off = np.zeros(n + 1, dtype=np.int64); off[1:] = np.cumsum(sizes)
    d_off = torch.from_numpy(off).to(dev)
    out = torch.empty(n * size, dtype=torch.uint8, device=dev)
    ooff = B.uniform_offsets(n, size, dev)
    res = torch.empty(n, dtype=torch.int64, device=dev)
    times = []
    for it in range(a.warmup + a.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); B.decompress(d_blob, d_off, out, ooff, res); e1.record()
        torch.cuda.synchronize()
        if it >= a.warmup:
            times.append(e0.elapsed_time(e1))
    want = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(dev)
    exact = bool((res == size).all()) and all(torch.equal(out[k * u * size:(k + 1) * u * size][: want.numel()], want[: min(want.numel(), (n - k * u) * size)]) for k in range(rep))
    ms = sum(times) / len(times)
    alg = n * size + int(off[-1])
    print(json.dumps({"diagnostic": "decode-only", "n": n, "size": size, "level": level, "ms": ms, "exact": exact,
                      "decompress_GiBps": n * size / GIB / (ms / 1e3), "ratio": n * size / int(off[-1]),
                      "roofline": {"achieved": alg / 1e9 / (ms / 1e3), "frac": alg / 1e9 / (ms / 1e3) / HBM_PEAK_GBPS}}))


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    zj = entry.load_package()
    if not os.path.exists(zj.LIB_PATH):
        zj.build()
    zj.batch.init(local)
    dev = torch.device("cuda", local)
    B = zj.batch
    n, size, level = a.buffers, a.size, a.level
    first = rank * n                              # disjoint buffer indices per rank (weak scaling)
    from zstd_jni_amd import shard

    # ---- inputs resident in HBM ----
    src = B.synth(n, size, first, dev)
    src_off = B.uniform_offsets(n, size, dev)
    bound = zj.Zstd.compressBound(size)
    comp = torch.empty(n * bound, dtype=torch.uint8, device=dev)
    comp_off = B.uniform_offsets(n, bound, dev)
    packed = torch.empty(n * bound, dtype=torch.uint8, device=dev)      # frames back to back (what a consumer / the gather sees)
    packed_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    back = torch.empty(n * size, dtype=torch.uint8, device=dev)
    csz = torch.empty(n, dtype=torch.int64, device=dev)
    dsz = torch.empty(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    if a.decode_only:
        decode_only(a, zj, dev, n, size, level)
        return

    ev = lambda: torch.cuda.Event(enable_timing=True)   # recorded on the stream the kernels run on
    t_c, t_p, t_d, t_g = [], [], [], []
    t_stage = []                                        # per-kernel HIP-event times from inside the library (same stream)

    def step(timed):
        e0, e1, e2, e3, e4 = ev(), ev(), ev(), ev(), ev()
        e0.record(); B.compress(src, src_off, comp, comp_off, level, csz); e1.record()
        B.pack(csz, comp, comp_off, out=packed, out_off=packed_off); e2.record()      # frames back to back
        handle = None
        if world > 1 and not a.no_gather:                   # posted before the local decompress: xGMI transfers run beside it
            total = int(packed_off[-1].item())
            handle = shard.gather_packed_start(packed[:total], csz, dst=0)
        B.decompress(packed, packed_off, back, src_off, dsz); e3.record()
        if handle is not None:
            shard.gather_packed_finish(handle)
        e4.record()
        if timed:
            torch.cuda.synchronize()
            t_c.append(e0.elapsed_time(e1)); t_p.append(e1.elapsed_time(e2)); t_d.append(e2.elapsed_time(e3)); t_g.append(e3.elapsed_time(e4))
            t_stage.append(B.last_timing())

    if a.warmup == 0:
        step(False)                                      # one-time scratch allocation (tens of GiB, kept by the library) is set-up, not a step
    for _ in range(a.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        w = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())

    # ---- parity gates (outside the timed region) ----
    ok_sizes = bool((csz > 0).all()) and bool((dsz == size).all())
    roundtrip = ok_sizes and torch.equal(back, src)
    csum = int(csz.clamp(min=0).sum().item())
    cpu = None
    gates = {"gpu_roundtrip_exact": bool(roundtrip)}
    if rank == 0 and not a.skip_cpu:
        from oracle import port, ref
        k = min(a.verify_sample, n)
        sizes = csz[:k].cpu().tolist()
        blob = comp[:k * bound].cpu().numpy()
        host_off = None
        host_src = src[:k * size].cpu().numpy().tobytes()
        checker = ref if ref.available() else port
        cpu_ok, first_bad = True, None
        for i in range(k):
            f = blob[i * bound:i * bound + max(sizes[i], 0)].tobytes()
            try:
                good = sizes[i] > 0 and checker.decompress(f, size) == host_src[i * size:(i + 1) * size]
            except Exception as ex:          # noqa: BLE001 - report, do not crash the bench line
                good = False
                first_bad = first_bad or f"{i}: size {sizes[i]} {ex} head {f[:12].hex()}"
            if not good:
                cpu_ok = False
                first_bad = first_bad or f"{i}: size {sizes[i]} head {f[:12].hex()}"
        gates["cpu_decodes_gpu_frames"] = cpu_ok
        if first_bad:
            gates["first_bad_frame"] = first_bad
        if ref.available():
            # byte identity, outside the timed region: the frames above against the reference given the library's level-3
            # table sizes, and a second GPU pass with the level's own sizes (setHashLog(16).setChainLog(15)) against the
            # reference's plain call
            want = [ref.compress(host_src[i * size:(i + 1) * size], 3, False, 14, 13) if level == 3 else ref.compress(host_src[i * size:(i + 1) * size], level) for i in range(k)]
            gates["frames_byte_identical_to_reference"] = all(blob[i * bound:i * bound + max(sizes[i], 0)].tobytes() == want[i] for i in range(k))
            if level == 3:
                c2 = torch.empty(k * bound, dtype=torch.uint8, device="cuda")
                s2 = B.compress(src[:k * size], B.uniform_offsets(k, size, "cuda"), c2, B.uniform_offsets(k, bound, "cuda"), 3, hash_log=16, chain_log=15)
                torch.cuda.synchronize()
                z2, b2 = s2.cpu().tolist(), c2.cpu().numpy()
                gates["plain_level3_byte_identical_with_hashLog16_chainLog15"] = all(
                    b2[i * bound:i * bound + max(z2[i], 0)].tobytes() == ref.compress(host_src[i * size:(i + 1) * size], 3) for i in range(k))
        neg = int((csz <= 0).sum().item())
        if neg:
            gates["frames_with_error_result"] = neg
            gates["error_results"] = sorted(set(csz[csz <= 0].cpu().tolist()))[:4]
        # CPU baseline on a bounded sample of the same workload (same generator, same indices)
        m = min(a.cpu_sample, n)
        sample = zj.synth_host(size, first, m)
        threads = os.cpu_count() or 1
        r = port.cpu_baseline(sample, size, level, threads, reps=2, use_ref=True)
        gpu_c_sample = int(csz[:m].sum().item())
        gates["ratio_gpu_over_cpu_size"] = gpu_c_sample / max(r["compressed_bytes"], 1)
        gates["ratio_within_1pct"] = gpu_c_sample <= 1.01 * r["compressed_bytes"]
        tot = m * size
        cpu = {"value": tot / GIB / (r["compress_s"] + r["decompress_s"]), "unit": "GiB/s", "cores": threads, "kind": r["kind"],
               "sample": f"{m} x {size} B of the same generator, level {level}, best of 2 after warm-up, one reused ctx per thread",
               "compress_GiBps": tot / GIB / r["compress_s"], "decompress_GiBps": tot / GIB / r["decompress_s"],
               "ratio": tot / max(r["compressed_bytes"], 1), "roundtrip_exact": r["exact"]}

    if rank == 0:
        ms = wall * 1000.0 / a.steps
        total_unc = world * n * size
        mc, mp, md, mg = (sum(x) / len(x) for x in (t_c, t_p, t_d, t_g))
        alg = n * size + csum                                  # S + C per launch
        stage = {k: sum(t[k] for t in t_stage) / len(t_stage) for k in t_stage[0]} if t_stage else {}
        # kernels of the two paths with their own HIP-event durations (ms); "compress_rest" = classify + table memset +
        # entropy kernel (it runs beside the match kernel on a side stream) + sweep, i.e. compress call minus match kernel
        kernels = {"zj_enc_match_kernel": stage.get("match", -1.0), "zj_dec_prep_kernel": stage.get("dec_prep", -1.0),
                   "zj_dec_seq_kernel": stage.get("dec_seq", -1.0), "zj_dec_exec_kernel": stage.get("dec_exec", -1.0),
                   "zj_decode_kernel(leftovers)": stage.get("dec_fused", -1.0), "zj_pack_kernel": mp}
        if kernels["zj_enc_match_kernel"] > 0:
            kernels["compress_rest(entropy beside match, memset, sweep)"] = mc - kernels["zj_enc_match_kernel"]
        dom = max((k for k in kernels if kernels[k] > 0 and not k.startswith("compress_rest")), key=lambda k: kernels[k], default=None)
        if dom is None:                                        # small batches: fused kernels only
            dom, dom_ms = ("zj_encode_kernel", mc) if mc >= md else ("zj_decode_kernel", md)
        else:
            dom_ms = kernels[dom]
        achieved = alg / 1e9 / (dom_ms / 1e3)
        traffic, traffic_note, request_roof = None, None, None
        try:                                                   # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/)
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)
            rec = pmc.get(f"L{level}_{n}x{size}", {}).get(dom.split("(")[0])
            if rec:
                traffic, traffic_note = rec["hbm_bytes_per_launch"], pmc.get("note")
                # scattered table accesses: what bounds this kernel is HBM *requests* (64-B reads, 32-B writes), not bytes —
                # the ceiling is tools/micro/chase's measurement on this part
                reqs = rec["fetch_bytes_per_launch"] / 64.0 + rec["write_bytes_per_launch"] / 32.0
                request_roof = {"hbm_requests_per_launch": reqs, "achieved_G_per_s": reqs / 1e9 / (dom_ms / 1e3),
                                "measured_ceiling_G_per_s": 52.0, "frac": reqs / 1e9 / (dom_ms / 1e3) / 52.0,
                                "note": "random-access request ceiling measured with tools/micro/chase (65 536 dependent chains over 6 GiB)"}
        except OSError:
            pass
        out = {
            "metric": "GiB/s compress+decompress (L3, 64Ki x 64KiB)", "value": total_unc / GIB / (ms / 1e3), "unit": "GiB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{n} x {size} B mixed-entropy buffers per GPU, zstd level {level}, one frame per buffer",
                       "level": level, "buffers_per_gpu": n, "buffer_bytes": size, "parallelism": f"batch-sharded x{world}",
                       "gather": bool(world > 1 and not a.no_gather)},
            "compress_GiBps_per_gpu": n * size / GIB / (mc / 1e3), "decompress_GiBps_per_gpu": n * size / GIB / (md / 1e3),
            "kernel_ms": {"compress_call": mc, "decompress_call": md, **kernels, "rccl_gather": mg},
            "ratio": n * size / max(csum, 1),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel_ms": dom_ms, "algorithmic_bytes_per_launch": alg, "request_roof": request_roof,
                         "decompress_path": {"achieved": alg / 1e9 / (md / 1e3), "frac": alg / 1e9 / (md / 1e3) / HBM_PEAK_GBPS},
                         "compress_path": {"achieved": alg / 1e9 / (mc / 1e3), "frac": alg / 1e9 / (mc / 1e3) / HBM_PEAK_GBPS}},
            "cpu_baseline": cpu, "parity": gates,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
