"""Multi-GPU sharding of a batch: buffers are independent, so the batch is cut into
contiguous index ranges, one per rank, with NO collective on the compute path.  The only exchange
is output assembly: a size-vector all-gather followed by a payload gather of the tightly packed
compressed bytes (variable length per rank).

torch.distributed backend "nccl" IS RCCL on ROCm; the same code runs on "gloo" for the CPU
world_size-2 tests.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a gather-to-root fans
in over 7 distinct links, so the payload uses direct send/recv (batched isend/irecv = one
ncclGroupStart/End) instead of a ring all-gather, which would be single-link bound.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous index range [lo, hi) of rank `rank` when n buffers are split over `world` ranks."""
    return n * rank // world, n * (rank + 1) // world


def gather_sizes(sizes):
    """All-gather of the per-buffer compressed sizes (int64[n_local]); ranks may hold different n.
    Returns a list of int64 tensors, one per rank."""
    world = dist.get_world_size()
    n_local = torch.tensor([sizes.numel()], dtype=torch.int64, device=sizes.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    nmax = int(max(int(c.item()) for c in counts))
    pad = torch.zeros(nmax, dtype=torch.int64, device=sizes.device)
    pad[: sizes.numel()] = sizes
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return [o[: int(c.item())] for o, c in zip(outs, counts)]


def gather_packed_start(packed, sizes, dst=0):
    """Post the gather of every rank's packed compressed bytes on `dst` and return a handle; the transfers run on the
    backend's own stream, so work enqueued afterwards on the caller's stream (the local decompress) overlaps them.
    `packed` must stay untouched until gather_packed_finish(handle).

    packed: uint8[sum(sizes)] (this rank's frames back to back), sizes: int64[n_local]."""
    world, rank = dist.get_world_size(), dist.get_rank()
    all_sizes = gather_sizes(sizes.clamp(min=0))
    totals = [int(s.sum().item()) for s in all_sizes]
    if rank == dst:
        blob = torch.empty(sum(totals), dtype=torch.uint8, device=packed.device)
        ops, pos = [], 0
        for r in range(world):
            view = blob[pos:pos + totals[r]]
            if r == rank:
                view.copy_(packed[: totals[r]])
            elif totals[r]:
                ops.append(dist.P2POp(dist.irecv, view, r))
            pos += totals[r]
        works = dist.batch_isend_irecv(ops) if ops else []
        return {"works": works, "blob": blob, "sizes": all_sizes, "keep": None}
    works, keep = [], None
    if totals[rank]:
        keep = packed[: totals[rank]].contiguous()
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend, keep, dst)])
    return {"works": works, "blob": None, "sizes": all_sizes, "keep": keep}


def gather_packed_finish(handle):
    """Wait for the transfers of gather_packed_start.  On the destination rank returns (blob uint8[total], offsets
    int64[n_total+1]) with ranks concatenated in rank order — i.e. buffer order of the original batch; elsewhere (None, None)."""
    for w in handle["works"]:
        w.wait()
    if handle["blob"] is None:
        return None, None
    cat = torch.cat(handle["sizes"])
    off = torch.zeros(cat.numel() + 1, dtype=torch.int64, device=handle["blob"].device)
    off[1:] = torch.cumsum(cat, 0)
    return handle["blob"], off


def gather_packed(packed, sizes, dst=0):
    """gather_packed_start + gather_packed_finish."""
    return gather_packed_finish(gather_packed_start(packed, sizes, dst))
