"""CPU (-m "not gpu"): bench.py's N = 2 control flow, end to end, over gloo — the path no box of rounds 1-6 could run (every box had one GPU): WORLD_SIZE / RANK from
the environment, rank -> first buffer index (disjoint generator indices per rank, weak scaling), the step loop with the packed-frame gather POSTED before the local
decompress and collected after it (shard.gather_packed_start / _finish), the barriers, the wall-time MAX over ranks, the CPU legs on rank 0 only and skipped at N > 1,
one JSON line from rank 0 with n_gpus = 2 and value = the bytes of BOTH ranks over that time.  bench.main() itself runs; only its platform object (bench.GpuPlatform:
device, events, the C-ABI's batch entries, RCCL) is replaced by a stand-in whose batch entries are the kernel bodies compiled lane-serial (tests/emu) on CPU tensors.
Test infrastructure: bench.py has no such mode."""
import io
import json
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _Event:
    def __init__(self): self.t = 0.0
    def record(self): self.t = time.perf_counter()
    def elapsed_time(self, other): return (other.t - self.t) * 1e3


class _EmuBatch:
    """zstd-jni_amd/batch.py's entries over CPU tensors, a frame at a time through tests/emu/libzjni_emu.so"""

    def __init__(self, zj):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import util
        self.zj, self.util, self.L = zj, util, util.emu_lib()
        self.calls = {"compress": 0, "pack": 0, "decompress": 0}

    def synth(self, n, size, first_index=0, device="cpu"):
        return torch.frombuffer(bytearray(self.zj.synth_host(size, first_index, n)), dtype=torch.uint8)

    def uniform_offsets(self, n, stride, device="cpu"):
        return torch.arange(n + 1, dtype=torch.int64) * stride

    def compress(self, src, src_off, dst, dst_off, level, out=None, dictionary=None, **kw):
        n = src_off.numel() - 1
        res = out if out is not None else torch.empty(n, dtype=torch.int64)
        raw = src.numpy().tobytes()
        for i in range(n):
            z = self.util.emu_compress(self.L, raw[int(src_off[i]):int(src_off[i + 1])], level)
            assert not isinstance(z, int), z
            dst[int(dst_off[i]):int(dst_off[i]) + len(z)] = torch.frombuffer(bytearray(z), dtype=torch.uint8)
            res[i] = len(z)
        self.calls["compress"] += 1
        return res

    def pack(self, results, blob, off, out=None, out_off=None):
        n = results.numel()
        out_off[0] = 0; torch.cumsum(results.clamp(min=0), 0, out=out_off[1:n + 1])
        for i in range(n):
            out[int(out_off[i]):int(out_off[i + 1])] = blob[int(off[i]):int(off[i]) + int(results[i])]
        self.calls["pack"] += 1
        return out, out_off

    def decompress(self, src, src_off, dst, dst_off, out=None, dictionary=None):
        n = src_off.numel() - 1
        res = out if out is not None else torch.empty(n, dtype=torch.int64)
        raw = src.numpy().tobytes()
        for i in range(n):
            cap = int(dst_off[i + 1]) - int(dst_off[i])
            d = self.util.emu_decompress(self.L, raw[int(src_off[i]):int(src_off[i + 1])], cap)
            assert not isinstance(d, int), d
            dst[int(dst_off[i]):int(dst_off[i]) + len(d)] = torch.frombuffer(bytearray(d), dtype=torch.uint8)
            res[i] = len(d)
        self.calls["decompress"] += 1
        return res

    def last_timing(self):
        return {}


class _GlooPlatform:
    native = False

    def __init__(self, zj, local):
        self.zj, self.local, self.dev, self.B = zj, local, torch.device("cpu"), _EmuBatch(zj)

    def init_dist(self): dist.init_process_group("gloo")
    def init_device(self): pass
    def sync(self): pass
    def event(self): return _Event()


def _worker(rank, world, port, outdir, n, size):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import bench
    import __graft_entry__ as entry
    shard = entry.load_package().shard                  # the module object bench.main() imports as zstd_jni_amd.shard
    posted, finished = [], []
    start, finish = shard.gather_packed_start, shard.gather_packed_finish

    def spy_start(packed, sizes, dst=0):
        posted.append((int(packed.numel()), int(sizes.numel()))); return start(packed, sizes, dst)

    def spy_finish(handle):
        blob, off = finish(handle); finished.append(None if blob is None else (int(blob.numel()), int(off.numel()))); return blob, off
    shard.gather_packed_start, shard.gather_packed_finish = spy_start, spy_finish
    plats = []

    def factory(zj, local):
        plats.append(_GlooPlatform(zj, local)); return plats[-1]
    buf = io.StringIO(); real = sys.stdout; sys.stdout = buf
    try:
        bench.main(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--buffers", str(n), "--size", str(size), "--level", "1", "--skip-cpu", "--skip-lds3", "--e2e-sample", "0"], platform_factory=factory)
    finally:
        sys.stdout = real
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump({"stdout": buf.getvalue(), "posted": posted, "finished": finished, "calls": plats[0].B.calls}, f)


def test_bench_control_flow_at_world_size_2(tmp_path):
    n, size, world = 6, 4096, 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), n, size)) for r in range(world)]
    for p in procs: p.start()
    for p in procs:
        p.join(300); assert p.exitcode == 0
    out = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    lines = [l for l in out[0]["stdout"].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in out[1]["stdout"].splitlines() if l.startswith("{")]      # ONE line, rank 0's
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak" and j["unit"] == "GiB/s" and j["config"]["gather"] is True
    assert j["config"]["buffers_per_gpu"] == n and j["config"]["parallelism"] == "batch-sharded x2"
    assert abs(j["value"] - world * n * size / (1 << 30) / (j["ms_per_step"] / 1e3)) < 1e-9 * max(1.0, j["value"])      # the whole job's bytes over the slowest rank's time
    assert j["cpu_baseline"] is None and j["end_to_end"] is None                                              # the CPU legs belong to the N = 1 line
    assert j["parity"]["gpu_roundtrip_exact"] is True
    for r in range(world):
        assert out[r]["calls"] == {"compress": 3, "pack": 3, "decompress": 3}                                # warm-up + 2 timed steps, nothing else
        assert len(out[r]["posted"]) == 3 and all(p[1] == n for p in out[r]["posted"])                      # the gather is posted every step, sizes of this rank's n frames
    assert all(f is None for f in out[1]["finished"]) and len(out[1]["finished"]) == 3                       # rank 1 only sends
    got = out[0]["finished"]
    assert len(got) == 3 and all(g[1] == world * n + 1 for g in got)                                          # rank 0 ends up with every rank's frames, offsets for 2n frames
    assert all(g[0] == out[0]["posted"][k][0] + out[1]["posted"][k][0] for k, g in enumerate(got))          # ... exactly the bytes both ranks posted
