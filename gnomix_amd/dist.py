"""Multi-GPU: haplotype-pair sharding + gather-only collectives.

Every haplotype is independent in base+smoother (src/Base/base.py:174, src/Smooth/utils.py:21) and Gnofix
couples only the two haplotypes of one individual (src/model.py:205-210), so the path shards by individual
with NO data-path collective; the model is replicated per GPU.  The only communication is the output
gather (labels / probabilities), done once per batch with torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests):

  * `dst=None`  every rank gets the full array: ONE all_gather of max-shard-sized blocks;
  * `dst=r`     only rank r receives: a true gather (`dist.gather`: ncclSend/ncclRecv pairs under RCCL) — the other
                ranks send their shard and receive nothing (whole genome, 100 k haplotypes = 49.6 GB of outputs: it
                must not land on all 8 GPUs to serve rank 0);
  * `gather=False` in `infer_sharded`: no collective at all — every rank keeps (and writes) its own row block,
                   which is what the writers want when outputs go straight to disk.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_haplotypes: int, world: int, rank: int):
    """Contiguous block of whole individuals for `rank`: returns (lo, hi) haplotype indices.
    Individuals (pairs) are split as evenly as possible; the first `n_ind % world` ranks get one more."""
    if n_haplotypes % 2:
        raise ValueError("haplotypes come in pairs (rows 2i, 2i+1 = individual i, src/utils.py:121-123)")
    n_ind = n_haplotypes // 2
    q, r = divmod(n_ind, world)
    lo = rank * q + min(rank, r)
    hi = lo + q + (1 if rank < r else 0)
    return 2 * lo, 2 * hi


def gather_rows(local, n_total, group=None, dst=None):
    """Gather per-rank row blocks (axis 0, sizes given by shard_bounds) into the full array.
    `local` is a torch tensor (CUDA under nccl, CPU under gloo).  dst=None -> every rank gets the result
    (all_gather); dst=r -> only rank r (a gather: the others send and get None).  Shards are padded to the
    largest shard so that one fixed-size collective is used (RCCL wants equal counts)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(n_total, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    lo, hi = sizes[rank]
    assert local.shape[0] == hi - lo, (local.shape, lo, hi)
    if hi - lo == mx:
        pad = local.contiguous()
    else:
        pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:hi - lo] = local
    if dst is None:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
    else:
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, gather_list=bufs, dst=dst if group is None else dist.get_global_rank(group, dst), group=group)
        if rank != dst:
            return None
    return torch.cat([b[:h - l] for b, (l, h) in zip(bufs, sizes)], dim=0)


def infer_sharded(fn, X, group=None, dst=0, gather=True):
    """Run `fn(X_shard) -> array/tensor or tuple of them (leading dim = shard haplotypes; None entries allowed)` on this
    rank's shard of X (numpy (N, C) or torch tensor, the FULL matrix or anything sliceable by rows) and gather the
    outputs.  Always returns a TUPLE (also without torch.distributed: the single-process result, un-gathered);
    None outputs stay None.  gather=False returns this rank's shard outputs and its (lo, hi) bounds last."""
    import torch
    import torch.distributed as dist

    def as_tuple(o):
        return tuple(o) if isinstance(o, (tuple, list)) else (o,)

    if not (dist.is_available() and dist.is_initialized()):
        outs = as_tuple(fn(X))
        return outs if gather else outs + ((0, X.shape[0]),)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    N = X.shape[0]
    lo, hi = shard_bounds(N, world, rank)
    outs = as_tuple(fn(X[lo:hi]))
    if not gather:
        return outs + ((lo, hi),)
    res = []
    for o in outs:
        if o is None:
            res.append(None)
            continue
        t = o if isinstance(o, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(o))
        if dist.get_backend(group) == "nccl" and not t.is_cuda:
            t = t.cuda()
        g = gather_rows(t, N, group=group, dst=dst)
        if g is not None and not isinstance(o, torch.Tensor):
            g = g.cpu().numpy()
        res.append(g)
    return tuple(res)
