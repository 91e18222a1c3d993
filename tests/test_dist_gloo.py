"""world_size-2 gloo tests (CPU) of the N>1 path: sharding by individual + gather-only collective.
The per-shard compute here is the oracle (no GPU in this container); on the GPU box the same functions
wrap DeviceModel.infer_device under backend nccl (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_whole_individuals():
    from gnomix_amd.dist import shard_bounds
    for n_ind in (1, 2, 3, 7, 8, 100, 101):
        for world in (1, 2, 3, 4, 8):
            b = [shard_bounds(2 * n_ind, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == 2 * n_ind
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert all(lo % 2 == 0 and hi % 2 == 0 for lo, hi in b)
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 2
    with pytest.raises(ValueError):
        shard_bounds(7, 2, 0)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from gnomix_amd import synth
    from gnomix_amd.dist import infer_sharded, shard_bounds
    from oracle import gnx_oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = synth.synthetic_model(C=3037, M=100, A=4, S=11, n_rounds=4, seed=2)
    X = synth.synthetic_X(14, d.C, seed=9)  # 7 individuals -> uneven shards (4 + 3)
    T = O.Trees(d.tree_off, d.left, d.right, d.feat, d.cond, d.tree_class, d.A, d.base_score)
    calls = []

    def fn(xs):
        calls.append(xs.shape[0])
        B = O.base_lr(xs, d.M, d.context, d.lr_coef, d.lr_intercept)
        return O.smooth_xgb(T, B, d.S)

    p_all, l_all = infer_sharded(fn, X, dst=None)
    p_dst, l_dst = infer_sharded(fn, X, dst=0)
    p_ref, l_ref = fn(X)
    lo, hi = shard_bounds(14, world, rank)
    ok = (np.array_equal(p_all, p_ref) and np.array_equal(l_all, l_ref) and calls[0] == hi - lo and
          ((rank == 0 and np.array_equal(p_dst, p_ref)) or (rank != 0 and p_dst is None)))
    q.put((rank, bool(ok), calls[0]))
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, True, 8), (1, True, 6)], res
