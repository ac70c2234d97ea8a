#!/usr/bin/env python3
"""bench.py — batched zstd compress + decompress on MI355X, BASELINE.json's metric and its named configs.

  python bench.py --gpus N --steps K --warmup W [--config metric|1|2|3|4|5shape]
  (N>1: launched by the driver through torch.distributed.run, one rank per GPU)

config (config.workload names it in the JSON line):
  metric  65,536 x 64 KiB mixed-entropy buffers, level 3, compress + decompress      value = both ways      (the default)
  1       1,024 x 1 MiB slices of Silesia xml (tests/golden/xml-1.zst), level 3, multi-block frames                value = both ways
  2       65,536 frames made by the REFERENCE at its plain level 3 (64 KiB each), decompress, bit-exact   value = decompress
  3       65,536 x 64 KiB, level 1 (ZSTD_fast) compress, frames checked by the reference                   value = compress
  4       2^20 x 4 KiB JSON-like records, one trained ZstdDictCompress, level 3                              value = compress
  5shape  65,536 x 128 KiB, level 3, compress + decompress (config 5's buffer shape on one GPU)              value = both ways
  5       BASELINE config 5 itself: 2^20 x 128 KiB buffers over the job, 2^20 / N per GPU (131,072 at N = 8), level 3, both ways, in chunks of
          65,536 buffers that reuse the device buffers (a chunk's input is generated in HBM before its timed region)         value = both ways
One step = one GPU pass of the config's direction(s) over the whole batch, inputs resident in HBM when the timed region
starts.  At N>1 every rank runs the same per-GPU batch on different buffer indices (weak scaling) and the packed compressed
output is gathered to rank 0 over RCCL inside the step.

value        = uncompressed bytes through the config's pass per second, whole job.
roofline     = dominant kernel's algorithmic bytes (S + C per buffer) / its HIP-event time, measured inside the library on the
               launch stream; traffic = HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/, stamped with the
               commit they were measured on).
cpu_baseline = the reference's own libzstd (oracle/_ref) on this box's host threads: threads and reused contexts exist before
               the clock starts, passes are released by a barrier and repeated for >= 1 s (oracle/cpu_baseline.c), on the full
               batch when host memory allows; all-core and single-core figures, CPU model string.
end_to_end   = the host-pointer entries a JNI batch native binds (zjni_*_batch: pack -> H2D -> kernels -> D2H -> scatter) on a
               bounded sample — never `value`; one blocking call, and (two_batches_in_flight) 8 batches through
               zjni_compress_batch_begin / zjni_batch_finish with two in flight, run by tools/e2e.py in a process of its own.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# HIP's default number of hardware queues (4), said out loud and recorded in the line (`hw_queues`): it is what a process gets whose launcher sets nothing — the library
# does not touch the environment.  Two host batches in flight want 16 (profiles/r05/e_): that figure comes from a child process started with 16 and says so.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

GIB = float(1 << 30)
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    "metric": dict(level=3, n=65536, size=65536, mode="both", headline="both", metric="GiB/s compress+decompress (L3, 64Ki x 64KiB)"),
    "1": dict(level=3, n=1024, size=1 << 20, mode="xml", headline="both",
              metric="GiB/s compress+decompress of 1 MiB buffers (Silesia xml slices), level 3, multi-block frames (BASELINE config 1's buffer on the GPU)"),
    "2": dict(level=3, n=65536, size=65536, mode="decode_ref", headline="decompress",
              metric="GiB/s batched decompress, 65 536 reference-made level-3 frames of 64 KiB (BASELINE config 2)"),
    "3": dict(level=1, n=65536, size=65536, mode="both", headline="compress", metric="GiB/s batched compress level 1, 65 536 x 64 KiB (BASELINE config 3)"),
    "4": dict(level=3, n=1 << 20, size=4096, mode="dict", headline="compress",
              metric="GiB/s level-3 compress with a shared ZstdDictCompress, 2^20 x 4 KiB JSON-like records (BASELINE config 4)"),
    "5": dict(level=3, n=1 << 20, size=131072, mode="both", headline="both",
              metric="GiB/s compress+decompress, 2^20 x 128 KiB buffers sharded over the GPUs of one node, level 3 (BASELINE config 5)"),
    "5shape": dict(level=3, n=65536, size=131072, mode="both", headline="both",
                   metric="GiB/s compress+decompress (L3, 64Ki x 128KiB: BASELINE config 5's buffers on one GPU)"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="metric", choices=sorted(CONFIGS))
    ap.add_argument("--buffers", type=int, default=0, help="override the config's buffer count (diagnostics)")
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--level", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=0, help="buffers in the CPU baseline (0 = the whole batch when host memory allows)")
    ap.add_argument("--cpu-seconds", type=float, default=1.0, help="timed CPU work per direction")
    ap.add_argument("--verify-sample", type=int, default=512, help="GPU frames re-decoded / re-made by the CPU reference")
    ap.add_argument("--e2e-sample", type=int, default=65536, help="buffers in the end-to-end (host-pointer) leg (at most 4 GiB of them), 0 = skip")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL gather of compressed output")
    ap.add_argument("--skip-lds3", action="store_true", help="skip the extra level-3 pass with the LDS-sized tables (hashLog 14 / chainLog 13)")
    ap.add_argument("--multi", default="", choices=["", "inprocess"], help="inprocess: also time zjni_compress_batch_multi / zjni_decompress_batch_multi (one process, a thread per visible GPU, host pointers) on a sample")
    ap.add_argument("--skip-cpu", action="store_true", help="skip the CPU reference legs (verification + cpu_baseline + end_to_end), e.g. under a profiler")
    return ap.parse_args(argv)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cpu_budget():
    """(threads worth running, description): the CPUs this process may actually use for a sustained run — the scheduler affinity and
    the cgroup CPU quota (cpu.max), not the number of processors /proc/cpuinfo shows.  On a box whose cgroup grants 16 CPUs of 256,
    a run of a few milliseconds bursts over all of them and a run of a second is throttled to 16: the former is what round 1 measured."""
    visible = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = visible
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                     # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    threads = min(visible, affinity)
    if quota:
        threads = max(1, min(threads, int(quota + 0.999)))
    return threads, {"processors_visible": visible, "affinity": affinity, "cgroup_cpu_quota": quota}


def host_mem_available():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:          # noqa: BLE001 - the GPU box has no .git
        return None


def cpu_baseline_leg(a, host, size, n, level, dictionary, offsets=None, keep=None):
    """all-core + single-core timing of the reference on the host (oracle/cpu_baseline.c zso_cpu_baseline3).  keep (a dict): receives the reference's frames of
    the whole sample ("frames": back to back, "sizes") for the byte-identity gate over every frame of the batch."""
    from oracle import port
    threads, budget = host_cpu_budget()
    total = int(offsets[-1]) if offsets is not None else n * size
    r = port.cpu_baseline2(host, size, n, level, threads, a.cpu_seconds, dictionary, offsets=offsets, keep_frames=keep is not None)
    if keep is not None:
        keep["frames"] = r.pop("frames"); keep["sizes"] = r.pop("sizes")
    k = max(1, min(n, (64 << 20) // max(size, 1)))                     # single core: 64 MiB of the same buffers
    r1 = port.cpu_baseline2(host, size, k, level, 1, min(a.cpu_seconds, 1.0), dictionary, offsets=None if offsets is None else offsets[:k + 1])
    tot1 = int(offsets[k]) if offsets is not None else k * size
    return {"value": total / GIB / (r["compress_s"] + r["decompress_s"]), "unit": "GiB/s", "cores": threads, "kind": "reference",
            "cpu_model": cpu_model(), "host": budget,
            "sample": f"{n} x {size} B of the same generator (the {'whole batch' if n >= a._n else 'first buffers of the batch'}), level {level}"
                      + (", shared CDict/DDict" if dictionary else "") + f"; {threads} threads with reused contexts created before the clock starts, "
                      f"barrier start, {r['passes'][0]} + {r['passes'][1]} timed passes (>= {a.cpu_seconds:g} s each way), best pass",
            "compress_GiBps": total / GIB / r["compress_s"], "decompress_GiBps": total / GIB / r["decompress_s"],
            "compress_GiBps_mean_pass": total / GIB / r["mean_compress_s"], "decompress_GiBps_mean_pass": total / GIB / r["mean_decompress_s"],
            "per_core": {"compress_GiBps": tot1 / GIB / r1["compress_s"], "decompress_GiBps": tot1 / GIB / r1["decompress_s"], "buffers": k},
            "ratio": total / max(r["compressed_bytes"], 1), "roundtrip_exact": r["exact"], "compressed_bytes": r["compressed_bytes"]}


def end_to_end_leg(zj, host, size, m, level, cd, dd):
    """zjni_compress_batch* / zjni_decompress_batch* from host pointers: what a JNI batch native pays (PCIe both ways included)"""
    L = zj.lib()
    bound = zj.Zstd.compressBound(size)
    src = np.ascontiguousarray(host[:m * size])
    comp = np.empty(m * bound, dtype=np.uint8); back = np.empty(m * size, dtype=np.uint8)
    vp = lambda base, stride: (C.c_void_p * m)(*[base + i * stride for i in range(m)])
    sp, cp, bp = vp(src.ctypes.data, size), vp(comp.ctypes.data, bound), vp(back.ctypes.data, size)
    ss = (C.c_size_t * m)(*([size] * m)); cc = (C.c_size_t * m)(*([bound] * m)); res = (C.c_size_t * m)(); res2 = (C.c_size_t * m)()
    best_c = best_d = 1e30
    for it in range(3):
        t0 = time.perf_counter()
        r = (L.zjni_compress_batch_usingCDict(sp, ss, cp, cc, res, m, cd, 0) if cd else L.zjni_compress_batch2(sp, ss, cp, cc, res, m, level, 0))
        t1 = time.perf_counter()
        assert not L.zjni_isError(r), r
        cs = (C.c_size_t * m)(*[res[i] for i in range(m)])
        t2 = time.perf_counter()
        r = L.zjni_decompress_batch_usingDDict(cp, cs, bp, ss, res2, m, dd)
        t3 = time.perf_counter()
        assert not L.zjni_isError(r), r
        if it:
            best_c, best_d = min(best_c, t1 - t0), min(best_d, t3 - t2)
    ok = all(res2[i] == size for i in range(m)) and bool((back == src).all())
    tot = m * size
    csum = int(sum(res[i] for i in range(m)))
    # Two batches in flight (zjni_*_batch_begin / zjni_batch_finish: the device's two staging slots): tools/e2e.py in a process of its own, STARTED with GPU_MAX_HW_QUEUES=16
    # (what INTEGRATION.md section 2 tells a deployment to put into the JVM's environment; this process runs with HIP's default 4, recorded as `hw_queues` in both places).
    # 8 batches of the same shape, never more than two begun and not finished; rate = batches / wall time.
    piped = None
    if not cd and level == 3:
        try:
            L.zjni_release_scratch()                              # the child allocates its own pipelines' scratch (~70 GiB) beside this process's buffers
            env = dict(os.environ); env["GPU_MAX_HW_QUEUES"] = "16"; env["E2E_BATCHES"] = "8"
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e.py"), str(m), str(size), "1"], env=env, capture_output=True, text=True, timeout=600)
            for line in out.stdout.splitlines():
                if line.startswith("{") and "two_batches_in_flight" in line:
                    piped = json.loads(line)["two_batches_in_flight"]
            if piped is None:
                piped = {"error": (out.stderr or out.stdout)[-200:]}
            else:
                piped["hw_queues"] = 16
                piped["note"] = "tools/e2e.py as a child process started with GPU_MAX_HW_QUEUES=16 (the launcher's setting, not the library's): zjni_compress_batch_begin / zjni_decompress_batch_begin, two jobs in flight, zjni_batch_finish in order"
        except Exception as ex:                                  # noqa: BLE001 - a reported extra
            piped = {"error": f"{type(ex).__name__}: {str(ex)[:160]}"}
    # what the host link gives: pinned copies of 1 GiB each way (best of 3), and the time the calls' own bytes need at those rates with both
    # directions running at once (compress: S in, C out; decompress: C in, S out) — the floor of a pipeline that hides everything but the link
    link = {}
    try:
        import torch
        nb = min(tot, 1 << 30)
        hp = torch.empty(nb, dtype=torch.uint8).pin_memory(); dv = torch.empty(nb, dtype=torch.uint8, device="cuda")
        h2d = d2h = 1e30
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); dv.copy_(hp, non_blocking=True); torch.cuda.synchronize(); h2d = min(h2d, time.perf_counter() - t0)
            torch.cuda.synchronize(); t0 = time.perf_counter(); hp.copy_(dv, non_blocking=True); torch.cuda.synchronize(); d2h = min(d2h, time.perf_counter() - t0)
        rh, rd = nb / h2d, nb / d2h
        floor_c, floor_d = max(tot / rh, csum / rd), max(csum / rh, tot / rd)
        link = {"h2d_GBps": rh / 1e9, "d2h_GBps": rd / 1e9, "pinned_copy_bytes": nb,
                "compress_fraction_of_link_floor": floor_c / best_c, "decompress_fraction_of_link_floor": floor_d / best_d,
                "note": "floor = max(bytes in / h2d, bytes out / d2h) for the call's own bytes; the calls also gather / scatter through pinned staging on host threads"}
        if piped and "compress_GiBps" in piped:
            link["compress_fraction_of_link_floor_two_in_flight"] = floor_c / (tot / GIB / piped["compress_GiBps"])
            link["decompress_fraction_of_link_floor_two_in_flight"] = floor_d / (tot / GIB / piped["decompress_GiBps"])
        del hp, dv
    except Exception as ex:                                      # noqa: BLE001 - the line is still worth printing
        link = {"error": str(ex)[:120]}
    return {"compress_GiBps": tot / GIB / best_c, "decompress_GiBps": tot / GIB / best_d, "both_GiBps": tot / GIB / (best_c + best_d), "two_batches_in_flight": piped, "link": link,
            "sample": f"{m} x {size} B through zjni_compress_batch{'_usingCDict' if cd else '2'} / zjni_decompress_batch_usingDDict (host pointers: gather into pinned staging on 8 threads, H2D in slices, kernels, device-side packing of the frames, D2H in slices, scatter), best of 2 after warm-up",
            "roundtrip_exact": ok}


def config5(a, zj, dev, rank, world, cfg):
    """BASELINE config 5: 2^20 x 128 KiB buffers over the whole job (strong scaling: 2^20 / N per GPU), in chunks of 65 536 buffers per GPU that
    reuse one set of device buffers.  Timed: compress -> pack -> (gather to rank 0, posted beside) decompress of every chunk, HIP events around each
    chunk on the launch stream; a chunk's input is generated in HBM between the timed regions (a 128 GiB input does not stay resident beside the
    work buffers).  value = job bytes / (max over ranks of the summed chunk times)."""
    from zstd_jni_amd import shard
    B = zj.batch; L = zj.lib()
    size, level = cfg["size"], cfg["level"]
    per_gpu = (cfg["n"] if not a.buffers else a.buffers) // world
    chunk = min(per_gpu, 65536)
    chunks = (per_gpu + chunk - 1) // chunk
    bound = zj.Zstd.compressBound(size)
    src = torch.empty(chunk * size, dtype=torch.uint8, device=dev); back = torch.empty_like(src)
    comp = torch.empty(chunk * bound, dtype=torch.uint8, device=dev); packed = torch.empty_like(comp)
    src_off = B.uniform_offsets(chunk, size, dev); comp_off = B.uniform_offsets(chunk, bound, dev)
    packed_off = torch.zeros(chunk + 1, dtype=torch.int64, device=dev)
    csz = torch.empty(chunk, dtype=torch.int64, device=dev); dsz = torch.empty(chunk, dtype=torch.int64, device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    tot_c = tot_d = tot_all = 0.0; csum = 0; exact = True; stage = []
    def one_pass(timed):
        nonlocal tot_c, tot_d, tot_all, csum, exact
        for c in range(chunks):
            m = min(chunk, per_gpu - c * chunk)
            s = B.synth(m, size, rank * per_gpu + c * chunk, dev); src[:m * size].copy_(s); del s
            torch.cuda.synchronize()
            e0, e1, e2, e3 = ev(), ev(), ev(), ev()
            e0.record(); B.compress(src[:m * size], src_off[:m + 1], comp, comp_off[:m + 1], level, csz[:m]); e1.record()
            B.pack(csz[:m], comp, comp_off[:m + 1], out=packed, out_off=packed_off[:m + 1]); e2.record()
            handle = None
            if world > 1 and not a.no_gather:
                handle = shard.gather_packed_start(packed[:int(packed_off[m].item())], csz[:m], dst=0)
            B.decompress(packed, packed_off[:m + 1], back, src_off[:m + 1], dsz[:m])
            if handle is not None:
                shard.gather_packed_finish(handle)
            e3.record(); torch.cuda.synchronize()
            if timed:
                tot_c += e0.elapsed_time(e1); tot_d += e2.elapsed_time(e3); tot_all += e0.elapsed_time(e3)
                csum += int(csz[:m].clamp(min=0).sum().item()); stage.append(B.last_timing())
                exact = exact and bool((csz[:m] > 0).all()) and bool((dsz[:m] == size).all()) and torch.equal(back[:m * size], src[:m * size])
    for _ in range(max(a.warmup, 0) and 1):
        one_pass(False)                               # (one warm-up pass allocates the library's scratch: a pass is 16 chunks at N = 1)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_pass(True)
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    t = torch.tensor([tot_all, tot_c, tot_d], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tot_all, tot_c, tot_d = (float(x) for x in t.tolist())
    if rank == 0:
        job = world * per_gpu * size
        ms = tot_all / a.steps
        wide = sum(x.get("match_wide", -1.0) for x in stage) / max(len(stage), 1)
        alg = (per_gpu * size + csum / a.steps) / chunks           # S + C of one chunk = one launch of the wide match kernel
        out = {"metric": cfg["metric"], "value": job / GIB / (ms / 1e3), "unit": "GiB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": f"2^20 x {size} B mixed-entropy buffers over the job = {per_gpu} per GPU in {chunks} chunks of {chunk}, zstd level 3, one frame per buffer",
                          "name": "5", "level": level, "buffers_per_gpu": per_gpu, "buffer_bytes": size, "parallelism": f"batch-sharded x{world}",
                          "gather": bool(world > 1 and not a.no_gather), "value_is": "both", "timed": "HIP events around every chunk's compress -> pack -> decompress; generation of the next chunk's input lies between"},
               "compress_GiBps_per_gpu": per_gpu * size / GIB / (tot_c / a.steps / 1e3), "decompress_GiBps_per_gpu": per_gpu * size / GIB / (tot_d / a.steps / 1e3),
               "wall_ms_per_step_including_generation": wall * 1e3 / a.steps, "ratio": per_gpu * size * a.steps / max(csum, 1),
               "roofline": {"bound": "hbm", "kernel": "zj_enc_match_wide_kernel", "kernel_ms": wide, "algorithmic_bytes_per_launch": alg,
                            "achieved": alg / 1e9 / (wide / 1e3) if wide > 0 else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": (alg / 1e9 / (wide / 1e3) / HBM_PEAK_GBPS) if wide > 0 else None, "traffic": None},
               "cpu_baseline": None, "parity": {"gpu_roundtrip_exact": bool(exact)},
               "library": {"build_stamp": L.zjni_build_stamp().decode()}}
        if chunk == 65536 and size == 131072:        # a chunk IS config 5shape's launch (65 536 x 128 KiB of the same generator): its counter pass, when it is of this build
            try:
                import glob
                for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
                    with open(path) as f:
                        rec = json.load(f).get(f"5shape_L{level}_{chunk}x{size}", {}).get("zj_enc_match_wide_kernel")
                    if rec and rec.get("build_stamp") == out["library"]["build_stamp"]:
                        out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                        out["roofline"]["traffic_note"] = f"the 5shape launch's PMC passes ({os.path.basename(path)}), build {rec['build_stamp']} = this library's"
                        break
            except OSError:
                pass
        if not a.skip_cpu and world == 1:
            from oracle import ref
            k = 256
            hk = src[:k * size].cpu().numpy(); zs = csz[:k].cpu().tolist(); blob = comp[:k * bound].cpu().numpy()
            out["parity"]["frames_byte_identical_to_reference"] = all(blob[i * bound:i * bound + zs[i]].tobytes() == ref.compress(hk[i * size:(i + 1) * size].tobytes(), level) for i in range(k))
            out["cpu_baseline"] = cpu_baseline_leg(a, src[:4096 * size].cpu().numpy(), size, 4096, level, None)
            out["cpu_baseline"]["sample"] = "4096 x 131072 B of the last chunk; " + out["cpu_baseline"]["sample"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def multi_inprocess_leg(zj, host, size, m, level):
    """one process, a host thread per visible GPU (zjni_compress_batch_multi / zjni_decompress_batch_multi, the JVM's case): mode 0 = every device returns
    its frames over its own PCIe link; mode 1 = frames packed and peer-copied to devices[0] first (the xGMI gather)"""
    L = zj.lib()
    nd = torch.cuda.device_count()
    devs = (C.c_int * nd)(*range(nd))
    bound = zj.Zstd.compressBound(size)
    src = np.ascontiguousarray(host[:m * size]); comp = np.empty(m * bound, dtype=np.uint8); back = np.empty(m * size, dtype=np.uint8)
    vp = lambda base, stride: (C.c_void_p * m)(*[base + i * stride for i in range(m)])
    sp, cp, bp = vp(src.ctypes.data, size), vp(comp.ctypes.data, bound), vp(back.ctypes.data, size)
    ss = (C.c_size_t * m)(*([size] * m)); cc = (C.c_size_t * m)(*([bound] * m)); res = (C.c_size_t * m)(); res2 = (C.c_size_t * m)()
    L.zjni_compress_batch_multi.argtypes = [C.c_void_p] * 5 + [C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.zjni_decompress_batch_multi.argtypes = [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p, C.c_int]
    out = {"devices": nd, "sample": f"{m} x {size} B from host pointers"}
    for mode in (0, 1):
        best = 1e30
        for it in range(3):
            t0 = time.perf_counter(); r = L.zjni_compress_batch_multi(sp, ss, cp, cc, res, m, level, 0, devs, nd, mode); t1 = time.perf_counter()
            assert not L.zjni_isError(r), r
            if it: best = min(best, t1 - t0)
        out[f"compress_mode{mode}_GiBps"] = m * size / GIB / best
    cs = (C.c_size_t * m)(*[res[i] for i in range(m)])
    best = 1e30
    for it in range(3):
        t0 = time.perf_counter(); r = L.zjni_decompress_batch_multi(cp, cs, bp, ss, res2, m, devs, nd); t1 = time.perf_counter()
        assert not L.zjni_isError(r), r
        if it: best = min(best, t1 - t0)
    out["decompress_GiBps"] = m * size / GIB / best
    out["roundtrip_exact"] = all(res2[i] == size for i in range(m)) and bool((back == src).all())
    return out


class GpuPlatform:
    """Where a step runs: the device of this rank, its stream events, the C-ABI's batch entries (zstd-jni_amd/batch.py) and RCCL.  main() goes through this object for
    everything that needs a GPU, so that tests/test_bench_gloo.py can run the SAME control flow — rank -> first buffer index, the step loop, the posted gather, the
    barriers, the MAX over ranks, rank 0's line — at world size 2 over gloo with a stand-in that decodes and encodes on the CPU (test infrastructure; never a bench mode)."""
    native = True

    def __init__(self, zj, local):
        self.zj, self.local, self.B = zj, local, zj.batch
        self.dev = torch.device("cuda", local)

    def init_dist(self):
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=self.dev)

    def init_device(self):
        if not os.path.exists(self.zj.LIB_PATH):
            self.zj.build()
        self.B.init(self.local)

    def sync(self):
        torch.cuda.synchronize()

    def event(self):
        return torch.cuda.Event(enable_timing=True)       # recorded on the stream the kernels run on


def main(argv=None, platform_factory=GpuPlatform):
    a = parse(argv)
    cfg = dict(CONFIGS[a.config])
    if a.buffers: cfg["n"] = a.buffers
    if a.size: cfg["size"] = a.size
    if a.level: cfg["level"] = a.level
    n, size, level, mode = cfg["n"], cfg["size"], cfg["level"], cfg["mode"]
    a._n = n
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: re-launch under torch.distributed.run, one rank per GPU (the form the driver uses for N > 1).
        # Never print an n_gpus 1 line for a --gpus N request: without N devices this fails loudly instead.
        have = torch.cuda.device_count()
        if have < a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but {have} device(s) visible")
        import socket, subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    zj = entry.load_package()
    P = platform_factory(zj, local)
    if world > 1:
        P.init_dist()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    P.init_device()
    dev = P.dev
    B = P.B
    if a.config == "5":
        return config5(a, zj, dev, rank, world, cfg)
    first = rank * n                              # disjoint buffer indices per rank (weak scaling)
    from zstd_jni_amd import shard

    # ---- inputs resident in HBM ----
    cdict = ddict = None; dict_bytes = None
    if mode == "dict":                            # JSON-like records only (class 1 of the generator: every fourth buffer), one trained dictionary
        from oracle import ref
        big = B.synth(4 * n, size, 4 * first, dev)
        src = big.view(n, 4, size)[:, 1, :].contiguous().view(-1)
        del big
        train = zj.synth_host(size, (1 << 24), 40000)
        dict_bytes = ref.train_dict([train[i * size:(i + 1) * size] for i in range(1, 40000, 4)], 112640)   # 110 KiB from 10 000 records
        cdict = zj.ZstdDictCompress(dict_bytes, level); ddict = zj.ZstdDictDecompress(dict_bytes)
    elif mode == "xml":                           # BASELINE config 1's buffer: 1 MiB of text-like data ('dickens' is not in the reference; its xml fixture is)
        from oracle import ref
        with open(os.path.join(ROOT, "tests", "golden", "xml-1.zst"), "rb") as f:
            xml = np.frombuffer(ref.decompress(f.read(), 6_000_000), dtype=np.uint8)
        span = xml.size - size
        src = torch.empty(n * size, dtype=torch.uint8, device=dev)
        hx = torch.from_numpy(xml.copy()).to(dev)
        for i in range(n):
            o = ((first + i) * 4099) % span
            src[i * size:(i + 1) * size] = hx[o:o + size]
        del hx
    else:
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
    P.sync()
    host_src = None
    if mode == "decode_ref":                      # the frames are the reference's: plain ZSTD_compress2(level), all host threads, outside the timed region
        from oracle import port
        host_src = src.cpu().numpy()
        frames, fsz = port.compress_many_packed(host_src, size, level, host_cpu_budget()[0])
        total = int(fsz.sum())
        packed[:total].copy_(torch.from_numpy(frames[:total]))
        csz.copy_(torch.from_numpy(fsz.astype(np.int64)))
        packed_off[1:] = torch.cumsum(csz, 0)
        del frames
        P.sync()

    ev = P.event
    t_c, t_p, t_d, t_g = [], [], [], []
    t_stage = []                                        # per-kernel HIP-event times from inside the library (same stream)

    def step(timed):
        e0, e1, e2, e3, e4 = ev(), ev(), ev(), ev(), ev()
        e0.record()
        if mode != "decode_ref":
            B.compress(src, src_off, comp, comp_off, level, csz, dictionary=cdict)
        e1.record()
        if mode != "decode_ref":
            B.pack(csz, comp, comp_off, out=packed, out_off=packed_off)               # frames back to back
        e2.record()
        handle = None
        if world > 1 and not a.no_gather and mode != "decode_ref":    # posted before the local decompress: xGMI transfers run beside it
            total = int(packed_off[-1].item())
            handle = shard.gather_packed_start(packed[:total], csz, dst=0)
        B.decompress(packed, packed_off, back, src_off, dsz, dictionary=ddict); e3.record()
        if handle is not None:
            shard.gather_packed_finish(handle)
        e4.record()
        if timed:
            P.sync()
            t_c.append(e0.elapsed_time(e1)); t_p.append(e1.elapsed_time(e2)); t_d.append(e2.elapsed_time(e3)); t_g.append(e3.elapsed_time(e4))
            t_stage.append(B.last_timing())

    if a.warmup == 0:
        step(False)                                      # one-time scratch allocation (tens of GiB, kept by the library) is set-up, not a step
    for _ in range(a.warmup):
        step(False)
    P.sync()
    if world > 1:
        dist.barrier()
    P.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    P.sync()
    if world > 1:
        dist.barrier()
    P.sync()
    wall = time.perf_counter() - t0
    if world > 1:
        w = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())

    # which match finder served the timed steps: asked of the library, not inferred from the environment
    L = zj.lib() if P.native else None
    scratch_bytes = 0; route = 0; lists = None
    if P.native:
        L.zjni_scratch_bytes.restype = C.c_size_t; L.zjni_scratch_bytes.argtypes = []
        scratch_bytes = int(L.zjni_scratch_bytes())       # what the timed steps made the library allocate and keep on this device (before the extra legs below)
        route = int(L.zjni_last_route()) if mode != "decode_ref" else 0
    if P.native and mode != "decode_ref" and hasattr(L, "zjni_last_lists"):
        # the library says how the batch was split: a batch whose frames went through the wide launch (128 KiB buffers) names THAT kernel, not list A's
        l3 = (C.c_uint * 3)()
        if L.zjni_last_lists(l3) == 0:
            lists = {"common_launch": int(l3[0]), "wide_launch": int(l3[1]), "multi_block_or_wave_per_frame": int(l3[2])}
            route = int(L.zjni_last_route())                                      # (again: whether list C went to the pipelined pair of waves is decided on the device — ZJNI_ROUTE_PIPE is known once the lists are)
            if l3[1] > l3[0] and l3[1] >= l3[2]: route = 10                      # ZJNI_ROUTE_WIDE
    route_kernel = L.zjni_route_kernel(route).decode() if route > 0 else ""
    stamp = L.zjni_build_stamp().decode() if P.native else "stand-in"

    # ---- the LDS-sized level-3 tables (hashLog 14 / chainLog 13: round 2's default, now behind setHashLog / setChainLog), on the record every run:
    # one warm-up + a timed pass, outside `value`.  The headline itself runs the reference's own sizes (16 / 15).
    lds3 = None
    if mode == "both" and level == 3 and size <= 65536 and not a.skip_lds3:
        csz2 = torch.empty(n, dtype=torch.int64, device=dev)
        for it in range(2):
            p0, p1 = ev(), ev()
            p0.record(); B.compress(src, src_off, comp, comp_off, 3, csz2, hash_log=14, chain_log=13); p1.record()
            P.sync()
        pms = p0.elapsed_time(p1); pst = B.last_timing(); proute = int(L.zjni_last_route())
        lds3 = {"hashLog": 14, "chainLog": 13, "compress_ms": pms, "compress_GiBps_per_gpu": n * size / GIB / (pms / 1e3),
                "match_kernel": L.zjni_route_kernel(proute).decode(), "match_kernel_ms": pst.get("match", -1.0), "route": proute,
                "compressed_bytes": int(csz2.clamp(min=0).sum().item()),
                "note": "ZstdCompressCtx.setHashLog(14).setChainLog(13) through zjni_compress_batch_device_advanced: the table sizes the LDS-resident finders of small batches use; byte identity with the reference given the same two parameters: parity.lds_tables_byte_identical_with_hashLog14_chainLog13"}
        B.compress(src, src_off, comp, comp_off, level, csz, dictionary=cdict)          # leave the headline's frames in `comp` for the gates below
        P.sync()

    # ---- what ONE call costs when the batch is small (a JVM thread's ZstdCompressCtx.compress, the aggregator's few hundred buffers): level 3, the first 1 024
    # buffers of the same batch, one warm-up + a timed call, outside `value`.  Below ZJNI_L3_WAVE_MAX frames the library sends every frame to a wave of its
    # own (zj_match_wavex.h on zj_encode_multi_kernel, route 9) instead of the lane pipeline, whose call costs ~95 ms whatever the batch size.
    small = None
    if mode == "both" and level == 3 and size <= 131072 and n >= 1024 and not a.skip_lds3:
        ns = 1024
        csz3 = torch.empty(ns, dtype=torch.int64, device=dev)
        for it in range(2):                              # frames go into `packed` (free after the timed steps): `comp` keeps the headline's frames for the gates
            p0, p1 = ev(), ev()
            p0.record(); B.compress(src[:ns * size], src_off[:ns + 1], packed, comp_off[:ns + 1], 3, csz3); p1.record()
            P.sync()
        sroute = int(L.zjni_last_route())
        small = {"frames": ns, "frame_bytes": size, "compress_call_ms": p0.elapsed_time(p1), "route": sroute, "kernel": L.zjni_route_kernel(sroute).decode(),
                 "same_sizes_as_the_full_batch": bool(torch.equal(csz3, csz[:ns])),
                 "note": "one level-3 call over the first 1 024 buffers of the batch (HIP events around the call, inputs in HBM); route 9 = wave per frame over HBM tables (ZJNI_ROUTE_WAVE_HBM)"}

    # ---- parity gates (outside the timed region) ----
    ok_sizes = bool((csz > 0).all()) and bool((dsz == size).all())
    roundtrip = ok_sizes and torch.equal(back, src)
    csum = int(csz.clamp(min=0).sum().item())
    cpu = None; e2e = None; multi = None
    gates = {"gpu_roundtrip_exact" if mode != "decode_ref" else "gpu_decodes_reference_frames_bit_exact": bool(roundtrip)}
    if rank == 0 and not a.skip_cpu:
        from oracle import port, ref
        assert ref.available(), "oracle/_ref/libzstd_ref.so missing: run __graft_entry__.build() where /root/reference exists"
        k = min(a.verify_sample, n, max(8, (64 << 20) // size))
        sizes = csz[:k].cpu().tolist()
        host_k = src[:k * size].cpu().numpy().tobytes()
        if mode != "decode_ref":
            blob = comp[:k * bound].cpu().numpy()
            cpu_ok, first_bad = True, None
            for i in range(k):
                f = blob[i * bound:i * bound + max(sizes[i], 0)].tobytes()
                try:
                    good = sizes[i] > 0 and (ref.decompress_using_dict(f, dict_bytes, size) if dict_bytes else ref.decompress(f, size)) == host_k[i * size:(i + 1) * size]
                except Exception as ex:          # noqa: BLE001 - report, do not crash the bench line
                    good = False
                    first_bad = first_bad or f"{i}: size {sizes[i]} {ex} head {f[:12].hex()}"
                if not good:
                    cpu_ok = False
                    first_bad = first_bad or f"{i}: size {sizes[i]} head {f[:12].hex()}"
            gates["cpu_decodes_gpu_frames"] = cpu_ok
            if first_bad:
                gates["first_bad_frame"] = first_bad
            # byte identity, outside the timed region: the frames above against the reference (given the library's level-3 table
            # sizes), and a second GPU pass with the level's own sizes (setHashLog(16).setChainLog(15)) against the reference's plain call
            if dict_bytes:
                rc = ref.CDict(dict_bytes, level)
                want = [rc.compress(host_k[i * size:(i + 1) * size]) for i in range(k)]
                rc.close()
            else:
                want = [ref.compress(host_k[i * size:(i + 1) * size], level) for i in range(k)]      # the reference's plain call: nothing but the level set
            gates["frames_byte_identical_to_reference"] = all(blob[i * bound:i * bound + max(sizes[i], 0)].tobytes() == want[i] for i in range(k))
            if level == 3 and not dict_bytes and size <= 131072:
                c2 = torch.empty(k * bound, dtype=torch.uint8, device=dev)
                s2 = B.compress(src[:k * size], B.uniform_offsets(k, size, dev), c2, B.uniform_offsets(k, bound, dev), 3, hash_log=14, chain_log=13)
                P.sync()
                z2, b2 = s2.cpu().tolist(), c2.cpu().numpy()
                gates["lds_tables_byte_identical_with_hashLog14_chainLog13"] = all(
                    b2[i * bound:i * bound + max(z2[i], 0)].tobytes() == ref.compress(host_k[i * size:(i + 1) * size], 3, False, 14, 13) for i in range(k))
            neg = int((csz <= 0).sum().item())
            if neg:
                gates["frames_with_error_result"] = neg
                gates["error_results"] = sorted(set(csz[csz <= 0].cpu().tolist()))[:4]
        if world == 1:                                     # the CPU legs belong to the N = 1 line only: at N > 1 seven ranks would wait for rank 0's host work
            # CPU baseline: the whole batch when the host has the memory (source + bound-sized frames + packed frames + decoded copy), else its head
            m = a.cpu_sample if a.cpu_sample else n
            need = m * (2 * size + 2 * bound)
            avail = host_mem_available()
            if avail and need > 0.6 * avail:
                m = max(1024, int(0.6 * avail // (2 * size + 2 * bound)))
            m = min(m, n)
            if host_src is None or host_src.size < m * size:
                host_src = src[:m * size].cpu().numpy()
            kept = {} if mode != "decode_ref" else None
            cpu = cpu_baseline_leg(a, host_src, size, m, level, dict_bytes, keep=kept)
            if kept and kept.get("frames") is not None:
                # EVERY frame of the batch the CPU leg covered against the reference's frame of the same buffer (the gate above looks at the first k):
                # the headline's frames (still in `comp`) packed back to back on the device, the reference's uploaded, sizes and bytes compared there
                B.pack(csz[:m], comp, comp_off[:m + 1], out=packed, out_off=packed_off[:m + 1])
                P.sync()
                rs = torch.from_numpy(kept["sizes"].astype(np.int64)).to(dev)
                same_sizes = bool(torch.equal(rs, csz[:m]))
                tot_ref = int(kept["sizes"].sum())
                same = same_sizes and bool(torch.equal(packed[:tot_ref], torch.from_numpy(kept["frames"][:tot_ref]).to(dev)))
                gates["all_frames_byte_identical_to_reference"] = {"frames_compared": m, "of": n, "identical": bool(same), "sizes_equal": same_sizes}
                if not same_sizes:
                    gates["all_frames_byte_identical_to_reference"]["first_size_difference"] = int((rs != csz[:m]).nonzero()[0].item())
                del rs
            kept = None
            gpu_c_sample = int(csz[:m].sum().item())
            gates["ratio_gpu_over_cpu_size"] = gpu_c_sample / max(cpu["compressed_bytes"], 1)
            gates["ratio_within_1pct"] = gpu_c_sample <= 1.01 * cpu["compressed_bytes"]
            if a.multi == "inprocess":
                try:
                    multi = multi_inprocess_leg(zj, host_src, size, max(1, min(m, 32768, (2 << 30) // size)), level)
                except Exception as ex:
                    multi = {"error": f"{type(ex).__name__}: {ex}"}
            if a.e2e_sample:
                me = max(1, min(a.e2e_sample, n, m, (4 << 30) // size))         # (m: what the host has room for, see above)
                try:
                    e2e = end_to_end_leg(zj, host_src, size, me, level, cdict._ptr if cdict else None, ddict._ptr if ddict else None)
                except Exception as ex:                          # a reported extra, never a reason to lose the line (e.g. pinned staging refused on a small host)
                    e2e = {"error": f"{type(ex).__name__}: {ex}"}

    if rank == 0:
        ms = wall * 1000.0 / a.steps
        total_unc = world * n * size
        mc, mp, md, mg = (sum(x) / len(x) for x in (t_c, t_p, t_d, t_g))
        alg = n * size + csum                                  # S + C per launch
        stage = {k: sum(t[k] for t in t_stage) / len(t_stage) for k in t_stage[0]} if t_stage else {}
        # kernels of the two paths with their own HIP-event durations (ms); "compress_rest" = classify + table memset +
        # entropy kernel (it runs beside the match kernel on a side stream) + sweep, i.e. compress call minus match kernel
        match_name = "zj_enc_match_dict_kernel(last slice)" if mode == "dict" else ("zj_enc_match_wide_kernel" if size > 65536 else (route_kernel or "zj_enc_match_kernel"))
        if size > 131072:
            match_name = route_kernel if route == 11 else "zj_encode_multi_kernel"       # (11 = ZJNI_ROUTE_PIPE: the pipelined pair of waves, decided on the device)
        gated = route == 6 or route == 4                       # ZJNI_ROUTE_RUN_FLAGS / ZJNI_ROUTE_LANE_GATED: flag kernels beside the match kernel
        kernels = {"zj_dec_prep_kernel": stage.get("dec_prep", -1.0), "zj_dec_seq_kernel": stage.get("dec_seq", -1.0),
                   "zj_dec_exec_kernel": stage.get("dec_exec", -1.0), "zj_decode_kernel(leftovers)": stage.get("dec_fused", -1.0)}
        if mode != "decode_ref":
            kernels[match_name] = mc if size > 131072 else stage.get("match_wide" if size > 65536 else "match", -1.0); kernels["zj_pack_kernel"] = mp
            if kernels[match_name] > 0 and mode != "dict" and size <= 65536:
                kernels["compress_rest(%sentropy beside match, memset, sweep)" % ("zj_enc_worth_kernel, " if gated else "")] = mc - kernels[match_name]
        slices = 1
        if mode == "dict":                                     # the dictionary pipeline runs in slices of 131 072 records: one launch = one slice
            slices = max(1, (n + 131071) // 131072)
        cand = [k for k in kernels if kernels[k] > 0 and not k.startswith("compress_rest") and (cfg["headline"] != "compress" or not k.startswith("zj_dec"))
                and (cfg["headline"] != "decompress" or k.startswith("zj_dec"))]
        dom = max(cand, key=lambda k: kernels[k] * (slices if "dict" in k else 1), default=None)
        if dom is None:                                        # small batches: fused kernels only
            dom, dom_ms = ("zj_encode_kernel", mc) if mc >= md else ("zj_decode_kernel", md)
        else:
            dom_ms = kernels[dom]
        alg_launch = alg / slices if (dom and "dict" in dom) else alg
        achieved = alg_launch / 1e9 / (dom_ms / 1e3)
        traffic, traffic_note, request_roof = None, None, None
        try:                                                   # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_traffic.sh -> profiles/r0N_pmc_traffic.json)
            import glob
            pmc, rec, pmc_name = {}, None, ""
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):      # the newest round's file first; a record of THIS build wins
                with open(path) as f:
                    cand_pmc = json.load(f)
                cand_rec = cand_pmc.get(f"{a.config}_L{level}_{n}x{size}", {}).get(dom.split("(")[0])
                if cand_rec and (rec is None or cand_rec.get("build_stamp") == stamp):
                    take = rec is None or rec.get("build_stamp") != stamp
                    if take: pmc, rec, pmc_name = cand_pmc, cand_rec, os.path.basename(path)
            if rec and rec.get("build_stamp") == stamp:        # a figure from another build of the library is not this run's: left out, and said so
                traffic = rec["hbm_bytes_per_launch"]
                traffic_note = f"{pmc.get('note', '')} Measured on build {rec['build_stamp']} = this library's zjni_build_stamp()."
                # scattered table accesses: what bounds this kernel is HBM *requests* (64-B reads, 32-B writes), not bytes —
                # the ceiling is tools/micro/probe's footprint sweep on this part
                reqs = rec["fetch_bytes_per_launch"] / 64.0 + rec["write_bytes_per_launch"] / 32.0
                request_roof = {"hbm_requests_per_launch": reqs, "achieved_G_per_s": reqs / 1e9 / (dom_ms / 1e3),
                                "measured_ceiling_G_per_s": 50.3, "frac": reqs / 1e9 / (dom_ms / 1e3) / 50.3,
                                "note": "random-access request ceiling at a 6 GiB footprint, read-only (40.8 read+write): tools/micro/probe footprint, profiles/r02a_probe_*"}
            elif rec:
                traffic_note = f"profiles/{pmc_name} holds a PMC pass of build {rec.get('build_stamp')}; this library is {stamp}: not quoted (re-run tools/pmc_traffic.sh)."
        except OSError:
            pass
        per_gpu = n * size / GIB
        headline_ms = {"both": ms, "compress": mc, "decompress": md}[cfg["headline"]]
        value = world * per_gpu / (headline_ms / 1e3)
        out = {
            "metric": cfg["metric"], "value": value, "unit": "GiB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic" if mode != "xml" else "Silesia xml (the reference's test fixture), overlapping 1 MiB slices",
            "config": {"workload": f"{n} x {size} B {'JSON-like records' if mode == 'dict' else ('slices of Silesia xml' if mode == 'xml' else 'mixed-entropy buffers')} per GPU, zstd level {level}, one frame per buffer"
                                   + (", one shared 110 KiB trained dictionary (ZstdDictCompress / ZstdDictDecompress)" if mode == "dict" else "")
                                   + (", frames made by the reference at its plain level (hashLog 16 / chainLog 15), GPU decompress only" if mode == "decode_ref" else ""),
                       "name": a.config, "level": level, "buffers_per_gpu": n, "buffer_bytes": size, "parallelism": f"batch-sharded x{world}",
                       "gather": bool(world > 1 and not a.no_gather), "value_is": cfg["headline"],
                       **({"hashLog": 16, "chainLog": 15, "table_sizes": "the reference's own for this level and size (N/compress/clevels.h + ZSTD_adjustCParams): nothing but the level is set; the LDS-sized 14 / 13 behind setHashLog / setChainLog: see lds_tables_level3"} if (level == 3 and 32768 < size <= 131072 and mode in ("both",)) else {})},
            "library": {"build_stamp": stamp, "match_route": route, "match_kernel": route_kernel, "frames_per_launch": lists,
                        "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "scratch_bytes": scratch_bytes, "scratch_GiB": scratch_bytes / GIB},
            "lds_tables_level3": lds3,
            "small_batch_level3": small,
            "compress_GiBps_per_gpu": (per_gpu / (mc / 1e3)) if mode != "decode_ref" else None, "decompress_GiBps_per_gpu": per_gpu / (md / 1e3),
            "kernel_ms": {"compress_call": mc, "decompress_call": md, **kernels, "rccl_gather": mg},
            "ratio": n * size / max(csum, 1),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel_ms": dom_ms, "algorithmic_bytes_per_launch": alg_launch, "request_roof": request_roof,
                         "decompress_path": {"achieved": alg / 1e9 / (md / 1e3), "frac": alg / 1e9 / (md / 1e3) / HBM_PEAK_GBPS},
                         "compress_path": ({"achieved": alg / 1e9 / (mc / 1e3), "frac": alg / 1e9 / (mc / 1e3) / HBM_PEAK_GBPS} if mode != "decode_ref" else None)},
            "cpu_baseline": cpu, "end_to_end": e2e, "multi_inprocess": multi, "parity": gates,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
