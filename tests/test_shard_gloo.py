"""CPU (-m "not gpu"): the N>1 path of bench.py — contiguous sharding of a batch over ranks and the
output-assembly exchange (size all-gather + packed-payload gather, zstd-jni_amd/shard.py) — on
world_size 2 with the gloo backend.  Payloads are real frames (made by the oracle), so the gathered
blob is checked by decoding every frame back to the rank-order concatenation of the inputs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as entry


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, size, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    zj = entry.load_package()
    from oracle import port as oracle_port
    lo, hi = zj.shard.shard_range(n_total, rank, world)
    raw = zj.synth_host(size, lo, hi - lo)                       # this rank's buffers (generator index = global index)
    frames = [oracle_port.compress(raw[i * size:(i + 1) * size], 1) for i in range(hi - lo)]
    sizes = torch.tensor([len(f) for f in frames], dtype=torch.int64)
    packed = torch.frombuffer(bytearray(b"".join(frames)), dtype=torch.uint8)
    if n_total % 2:
        blob, off = zj.shard.gather_packed(packed, sizes, dst=0)
    else:                                                        # the way bench.py uses it: post, do local work, then collect
        handle = zj.shard.gather_packed_start(packed, sizes, dst=0)
        local = [oracle_port.decompress(f, size) for f in frames]            # stands in for the local GPU decompress
        assert b"".join(local) == raw
        blob, off = zj.shard.gather_packed_finish(handle)
    if rank == 0:
        assert off.numel() == n_total + 1
        data = blob.numpy().tobytes()
        whole = zj.synth_host(size, 0, n_total)
        for i in range(n_total):
            f = data[int(off[i]):int(off[i + 1])]
            assert oracle_port.decompress(f, size) == whole[i * size:(i + 1) * size], i
        out_q.put(("ok", n_total, int(off[-1])))
    else:
        assert blob is None and off is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 64])
def test_gather_packed_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, 4096, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    tag, n, total = q.get(timeout=5)
    assert tag == "ok" and n == n_total and total > 0


def test_shard_range_partitions_exactly():
    zj = entry.load_package()
    for n in (0, 1, 7, 64, 65536, 1000003):
        for world in (1, 2, 4, 8):
            ranges = [zj.shard.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in ranges) - min(h - l for l, h in ranges) <= 1
