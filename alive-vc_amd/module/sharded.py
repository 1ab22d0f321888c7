"""Multi-GPU forms of the path (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

1. Windows are independent units -> `partition_windows`: pure data parallelism, no collective.
2. Library-sharded kNN (north-star option; BASELINE config 4): the bank is cut into contiguous row
   slabs, every rank scores ALL frames against its slab and rescoring happens on the owning rank, so
   the exchanged lists are already exact fp32 cosines: one all-gather of [Tt, k] (val fp32, idx int32
   global) per rank -- 8*k bytes per frame per rank, latency-bound on xGMI, not link-bound -- then
   every rank merges the S*k candidates and gathers from its replicated fp32 row table.

The search / merge callables are injected so the protocol itself (bounds, index bases, gather order,
merge inputs) is exercised by the world_size-2 gloo tests on CPU; on the GPU they are the HIP kernels.
"""
import time

import torch
import torch.distributed as dist


def shard_bounds(M, world):
    """contiguous row slabs, sizes differing by at most one: [(begin, end)] * world"""
    base, rem = divmod(M, world)
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < rem else 0)
        out.append((b, e))
        b = e
    return out


def partition_windows(n_windows, world, rank):
    """contiguous block of windows owned by `rank` (weak/strong DP over independent windows)"""
    b, e = shard_bounds(n_windows, world)[rank]
    return slice(b, e)


def allgather_candidates(val, idx, group=None):
    """[Tt, k] per rank -> ([S, Tt, k], [S, Tt, k]) in rank order (the layout alive_knn_merge_gather reads)."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo" and val.is_cuda:      # test-only path (1-GPU box): gloo moves host memory
        gv, gi = allgather_candidates(val.cpu(), idx.cpu(), group)
        return gv.to(val.device), gi.to(val.device)
    tt, k = val.shape
    gv = torch.empty((world * tt, k), dtype=val.dtype, device=val.device)
    gi = torch.empty((world * tt, k), dtype=idx.dtype, device=idx.device)
    dist.all_gather_into_tensor(gv, val.contiguous(), group=group)
    dist.all_gather_into_tensor(gi, idx.contiguous(), group=group)
    return gv.view(world, tt, k), gi.view(world, tt, k)


class ShardedLibrary:
    """`search(source, k) -> (val, idx_global)` over this rank's slab, `merge(gv, gi, S, k, alpha, source)`
    over the gathered lists."""

    def __init__(self, search, merge, group=None):
        self.search, self.merge, self.group = search, merge, group

    def match(self, source, k=4, alpha=0.0):
        val, idx = self.search(source, k)
        gv, gi = allgather_candidates(val, idx, self.group)
        return self.merge(gv, gi, dist.get_world_size(self.group), k, alpha, source)


def make_hip_sharded_library(tokens_DxM, rank, world, group=None):
    """HIP instantiation: this rank packs only its slab for scoring and keeps the full fp32 row table
    (3 GB at 1 M vectors: trivial in 288 GB) for the final gather."""
    from .common import PackedLibrary, merge_gather
    M = tokens_DxM.shape[1]
    b, e = shard_bounds(M, world)[rank]
    shard = PackedLibrary(tokens_DxM[:, b:e].contiguous(), idx_base=b)
    rows_full = tokens_DxM.t().contiguous()

    def merge(gv, gi, S, k, alpha, source):
        return merge_gather(gv, gi, S, k, alpha, rows_full, source)
    return ShardedLibrary(shard.search, merge, group)


def bench_sharded_knn(conv, windows, M, k, world, rank, dev, steps=3):
    """times content features -> sharded match (search + all-gather + merge) for one window batch."""
    g = torch.Generator(device=dev).manual_seed(1234)
    tokens = torch.randn(768, M, device=dev, generator=g)
    lib = make_hip_sharded_library(tokens, rank, world)
    del tokens
    feat, _ = conv.features(windows)
    lib.match(feat, k)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        lib.match(feat, k)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tt = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    frames = feat.shape[0] * feat.shape[2]
    return {"frames": frames, "ms": round(tt.item() * 1e3, 3), "frames_per_s": round(frames / tt.item(), 1),
            "shards": world, "rows_per_shard": M // world, "exchange": "all_gather [S,Tt,k] fp32+int32"}
