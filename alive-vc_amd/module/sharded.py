"""Multi-GPU forms of the path (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

1. Windows are independent units -> `partition_windows`: pure data parallelism, no collective.
2. Library-sharded kNN (north-star option; BASELINE config 4): the bank is cut into contiguous row
   slabs and every rank scores ALL frames against its slab.  Two exchange steps:
     a. the frames: encoders run data-parallel, so a rank only holds the content features of its own
        windows -> one all-gather of [n, 768, T] fp32 (3 KB per frame; rescoring on the owning shard is
        exact fp32, so the frames travel as fp32);
     b. the lists: rescoring happens on the rank that owns the rows, so the exchanged lists are already
        exact fp32 cosines -> one all-to-all that routes every shard's lists to the rank that owns the frames
        ([S, own frames, k], val fp32 + idx int32 global: 8*k bytes per frame per shard -- latency-bound on
        xGMI, not link-bound; an all-gather would deliver `world` times as much to every rank).
   Every rank then merges the S*k candidates of ITS OWN frames and gathers from its replicated fp32 row
   table; the decoder runs data-parallel again.  `ShardedLibrary.match` is the form for frames that are
   already replicated (step b only), `match_distributed` the end-to-end form (a + b).

The search / merge callables are injected so the protocol itself (bounds, index bases, gather order,
padding of uneven window counts, merge inputs) is exercised by the world_size-2 gloo tests on CPU; on
the GPU they are the HIP kernels.
"""
import time

import torch
import torch.distributed as dist

from .pipeline import Converter


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


def allgather_rows(x, group=None):
    """[r, ...] per rank (same shape on every rank) -> [world * r, ...] in rank order."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo" and x.is_cuda:         # test-only path (1-GPU box): gloo moves host memory
        return allgather_rows(x.cpu(), group).to(x.device)
    x = x.contiguous()
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out


def allgather_candidates(val, idx, group=None):
    """[Tt, k] per rank -> ([S, Tt, k], [S, Tt, k]) in rank order (the layout alive_knn_merge_gather reads)."""
    world = dist.get_world_size(group)
    tt, k = val.shape
    return allgather_rows(val, group).view(world, tt, k), allgather_rows(idx, group).view(world, tt, k)


def alltoall_candidates(val, idx, group=None):
    """Routes the lists to the ranks that own the frames.  val / idx [world * r, k] on every rank: this shard's exact lists
    of ALL frames, frames in owner-rank order, r frames per owner -> ([S, r, k], [S, r, k]): the lists of every shard for
    THIS rank's r frames.  One all_to_all_single of val and idx packed side by side ([.., 2k] fp32 words): a rank receives
    (world - 1) * r * k * 8 bytes, where an all-gather of everything would deliver world times that."""
    world = dist.get_world_size(group)
    tt, k = val.shape
    r = tt // world
    if dist.get_backend(group) == "gloo" and val.is_cuda:       # test-only path (1-GPU box): gloo moves host memory
        gv, gi = alltoall_candidates(val.cpu(), idx.cpu(), group)
        return gv.to(val.device), gi.to(val.device)
    send = torch.cat([val.contiguous(), idx.contiguous().view(torch.float32)], dim=1).contiguous()      # [world * r, 2k]
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    recv = recv.view(world, r, 2 * k)
    return recv[:, :, :k].contiguous(), recv[:, :, k:].contiguous().view(torch.int32)


class ShardedLibrary:
    """`search(source, k) -> (val, idx_global)` over this rank's slab, `merge(gv, gi, S, k, alpha, source)`
    over the gathered lists (optionally `return_indices=True` -> (out, idx[Tt, k]))."""

    def __init__(self, search, merge, group=None):
        self.search, self.merge, self.group = search, merge, group
        self.last_exchange_bytes = None

    def match(self, source, k=4, alpha=0.0, **kw):
        """frames replicated on every rank: search the slab, all-gather the lists, merge."""
        val, idx = self.search(source, k)
        gv, gi = allgather_candidates(val, idx, self.group)
        return self.merge(gv, gi, dist.get_world_size(self.group), k, alpha, source, **kw)

    def match_distributed(self, source_local, k=4, alpha=0.0, counts=None, **kw):
        """frames data-parallel: `source_local` [n_r, 768, T] are this rank's windows.  counts = windows per rank in rank
        order (None: the same on every rank); uneven counts are padded with copies of the rank's first window so that
        both exchanges are plain equal-size all-gathers.  Returns the matched features of this rank's windows."""
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        n, d, t = source_local.shape
        counts = [n] * world if counts is None else list(counts)
        if counts[rank] != n:
            raise ValueError(f"rank {rank} holds {n} windows, counts say {counts[rank]}")
        n_max = max(counts)
        if n_max == 0:
            return (source_local, torch.empty(0, k, dtype=torch.int32, device=source_local.device)) if kw.get("return_indices") else source_local
        if min(counts) == 0:
            raise ValueError("every rank needs at least one window (pad the batch or use fewer ranks)")
        mine = source_local.contiguous()
        if n < n_max:
            mine = torch.cat([mine, mine[:1].expand(n_max - n, -1, -1)], 0)
        frames = allgather_rows(mine, self.group)                           # (a) [world * n_max, 768, T]
        val, idx = self.search(frames, k)                                   # this slab's exact top-k of EVERY frame
        gv, gi = alltoall_candidates(val, idx, self.group)                  # (b) [S, n_max * T, k]: only this rank's frames
        gv, gi = gv[:, :n * t].contiguous(), gi[:, :n * t].contiguous()
        self.last_exchange_bytes = {"frames_allgather_received": int((world - 1) * n_max * d * t * 4),
                                    "lists_alltoall_received": int((world - 1) * n_max * t * k * 8)}
        return self.merge(gv, gi, world, k, alpha, source_local, **kw)


def make_hip_sharded_library(tokens_DxM, rank, world, group=None, prefilter=None, rows_full=None):
    """HIP instantiation: this rank packs only its slab for scoring and keeps the full fp32 row table
    (3 GB at 1 M vectors: trivial in 288 GB) for the final gather.  rows_full: an existing [M, 768] fp32 row table of the
    same library (a replicated PackedLibrary's `.rows`) to reuse instead of a second transposed copy."""
    from .common import PackedLibrary, merge_gather
    M = tokens_DxM.shape[1]
    b, e = shard_bounds(M, world)[rank]
    shard = PackedLibrary(tokens_DxM[:, b:e].contiguous(), idx_base=b, prefilter=prefilter)
    if rows_full is None:
        rows_full = tokens_DxM.t().contiguous()
    elif tuple(rows_full.shape) != (M, tokens_DxM.shape[0]):
        raise ValueError(f"rows_full is {tuple(rows_full.shape)}, expected {(M, tokens_DxM.shape[0])}")

    def merge(gv, gi, S, k, alpha, source, return_indices=False):
        return merge_gather(gv, gi, S, k, alpha, rows_full, source, return_indices)
    sl = ShardedLibrary(shard.search, merge, group)
    sl.shard, sl.rows_full = shard, rows_full
    return sl


class ShardedConverter(Converter):
    """BASELINE config 4 end to end: encoders data-parallel over this rank's windows -> feature all-gather -> every rank
    scores all frames against its library slab -> list all-gather -> merge + gather for its own frames -> decoder
    data-parallel.  Results equal the replicated-library `Converter` on the same windows."""

    def set_sharded_library(self, sharded: ShardedLibrary, counts=None):
        self.sharded, self.counts = sharded, counts
        return self

    def match(self, feat, k=4, alpha=0.0):
        return self.sharded.match_distributed(feat, k, alpha, self.counts)

    def convert_windows(self, windows, *a, share_overlap=None, **kw):
        # overlap sharing flattens a rank's distinct frames into ONE [1, 768, F] source whose F differs from rank to rank;
        # the exchange steps are equal-size collectives over whole windows, so the mode is refused here rather than padded
        if share_overlap:
            raise ValueError("share_overlap is not available with a sharded library: use the replicated Converter, or "
                             "share_overlap=None")
        return super().convert_windows(windows, *a, share_overlap=None, **kw)

    def _agree_on_saturations(self):
        # the fp16 range guard (ops.Fp16Guard) repeats a saturated batch -- through the exchange collectives: all ranks or none
        def agree(n):
            grp = self.sharded.group
            t = torch.tensor([float(n)], dtype=torch.float64, device=self.device if dist.get_backend(grp) == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
            return int(t.item())
        return agree


def _fence(dev):
    torch.cuda.synchronize(dev)
    dist.barrier()
    torch.cuda.synchronize(dev)


def _max_over_ranks(x, dev):
    t = torch.tensor([x], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def _all_true(flag, dev):
    t = torch.tensor([1.0 if flag else 0.0], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def bench_sharded(conv, tokens_DxM, windows_global, k, world, rank, dev, window_batch, steps=2):
    """BASELINE config 4 on the SAME global window batch on every rank (strong scaling of one fixed batch): rank r owns a
    contiguous block of the windows and 1/world of the library rows.  Checks, on the device and on every rank, that (i) the
    merged top-k lists of the rank's frames equal the unsharded search of the whole library bitwise and (ii) the waveforms
    of the end-to-end sharded conversion equal those of the replicated-library conversion bitwise; then times the
    end-to-end sharded step (max over ranks) and the match stage alone."""
    from .common import merge_gather
    n_total = windows_global.shape[0]
    counts = [e - b for b, e in shard_bounds(n_total, world)]
    own = windows_global[partition_windows(n_total, world, rank)].contiguous()
    # Everything that can fail on ONE rank alone (allocations: the slab's packed forms) happens before the first collective
    # of the leg, and the ranks agree on the outcome first: a rank that raised later, inside the exchange, would leave its
    # peers blocked in an all-gather for good (and the job's one JSON line unprinted).
    sl, err = None, None
    try:
        sl = make_hip_sharded_library(tokens_DxM, rank, world, rows_full=conv.library.rows)
    except Exception as e:                                    # noqa: BLE001
        err = f"{type(e).__name__}: {e}"[:200]
    if not _all_true(sl is not None, dev):
        return {"error": f"library shard set-up failed on some rank (this rank: {err})", "shards": world}
    sconv = ShardedConverter(conv.ce, conv.pe, conv.dec, dev).set_sharded_library(sl, counts)

    # (i) lists: sharded merge vs unsharded search, own frames (first window batch is enough to cover every code path once
    # per rank; every rank checks different frames)
    feat, _ = conv.features(own[:window_batch])
    cnt_b = [min(c, window_batch) for c in counts]
    out_s, idx_s = sl.match_distributed(feat, k, 0.0, cnt_b, return_indices=True)
    v1, i1 = conv.library.search(feat, k)
    out_u, idx_u = merge_gather(v1, i1, 1, k, 0.0, conv.library.rows, feat, return_indices=True)
    lists_equal = _all_true(torch.equal(idx_s, idx_u) and torch.equal(out_s, out_u), dev)
    del feat, out_s, idx_s, v1, i1, out_u, idx_u

    # (ii) end to end
    ref = conv.convert_windows(own, k=k, window_batch=window_batch)
    got = sconv.convert_windows(own, k=k, window_batch=window_batch)      # also the warm-up of the timed loop
    waves_equal = _all_true(torch.equal(ref, got), dev)
    del ref, got

    _fence(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        sconv.convert_windows(own, k=k, window_batch=window_batch)
    _fence(dev)
    dt = _max_over_ranks((time.perf_counter() - t0) / steps, dev)

    feat, _ = conv.features(own)
    sl.match_distributed(feat, k, 0.0, counts)
    _fence(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        sl.match_distributed(feat, k, 0.0, counts)
    _fence(dev)
    dm = _max_over_ranks((time.perf_counter() - t0) / steps, dev)

    lf = windows_global.shape[1] // 320
    M = tokens_DxM.shape[1]
    return {"equals_unsharded": bool(lists_equal and waves_equal), "lists_equal_unsharded": lists_equal,
            "waveforms_equal_replicated": waves_equal,
            "ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(n_total * lf / dt, 1),
            "match_ms": round(dm * 1e3, 3), "match_scoring_tflops_aggregate": round(2.0 * 768 * M * n_total * lf / dm / 1e12, 1),
            "global_windows": n_total, "global_frames": n_total * lf, "windows_per_rank": counts,
            "shards": world, "rows_per_shard": [e - b for b, e in shard_bounds(M, world)],
            "scaling": "strong (one fixed global batch; never part of `value`)",
            "exchange": "all_gather frames [n,768,T] fp32 + all_to_all lists [own frames,k] fp32+int32",
            "exchange_bytes_received_per_rank": sl.last_exchange_bytes, "backend": dist.get_backend()}
