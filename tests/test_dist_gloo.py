"""world_size-2 gloo tests of the N>1 path: sharding by individual + gather-only collective.
Without a GPU (this container) the per-shard compute is the oracle, so the test pins shard_bounds / gather_rows /
infer_sharded; WITH a GPU (the -m gpu tier runs this file too) each rank opens its own gnx_ctx on the device and the
per-shard compute is DeviceModel.infer — one context per process, model replicated, the product path — checked against
the oracle on the full batch.  The nccl (RCCL) flavour of the same path is tests/test_dist_nccl.py."""
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


def _worker(rank, world, port, q, use_gpu):
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

    def fn_oracle(xs):
        B = O.base_lr(xs, d.M, d.context, d.lr_coef, d.lr_intercept)
        return O.smooth_xgb(T, B, d.S)

    if use_gpu:  # every rank its own context + model replica on the (one) device: the product path
        import gnomix_amd
        dev = gnomix_amd.DeviceModel(d, ctx=gnomix_amd._lib.Context(0))

    def fn(xs):
        calls.append(xs.shape[0])
        return dev.infer(xs) if use_gpu else fn_oracle(xs)

    p_all, l_all = infer_sharded(fn, X, dst=None)
    p_dst, l_dst = infer_sharded(fn, X, dst=0)
    p_one, none_out = infer_sharded(lambda xs: (fn(xs)[0], None), X, dst=1)   # None outputs pass through un-gathered
    p_loc, l_loc, bounds = infer_sharded(fn, X, gather=False)                 # no collective: rank-local row block
    p_ref, l_ref = fn_oracle(X)
    lo, hi = shard_bounds(14, world, rank)
    close = (lambda a, b: np.array_equal(a, b)) if not use_gpu else (lambda a, b: np.max(np.abs(a - b)) <= 2.4e-7)
    ok = (close(p_all, p_ref) and np.array_equal(l_all, l_ref) and calls[0] == hi - lo and
          ((rank == 0 and close(p_dst, p_ref) and np.array_equal(l_dst, l_ref)) or (rank != 0 and p_dst is None and l_dst is None)) and
          none_out is None and ((rank == 1 and close(p_one, p_ref)) or (rank != 1 and p_one is None)) and
          bounds == (lo, hi) and close(p_loc, p_ref[lo:hi]) and np.array_equal(l_loc, l_ref[lo:hi]))
    q.put((rank, bool(ok), calls[0]))
    dist.destroy_process_group()


def _gpu_usable():
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def _run_two_ranks(use_gpu):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, use_gpu)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, True, 8), (1, True, 6)], res


def test_two_rank_gather_matches_single_process():
    _run_two_ranks(use_gpu=False)


@pytest.mark.gpu
def test_two_rank_gather_device_model_per_rank():
    """same harness, per-shard compute = DeviceModel on the GPU (one gnx_ctx per process, model replicated)"""
    if not _gpu_usable():
        pytest.skip("no GPU")
    _run_two_ranks(use_gpu=True)


def test_infer_sharded_without_process_group_returns_tuple():
    from gnomix_amd.dist import infer_sharded
    X = np.zeros((4, 3), np.int8)
    assert infer_sharded(lambda xs: xs.sum(1), X)[0].shape == (4,)           # bare array -> 1-tuple
    out = infer_sharded(lambda xs: (xs.sum(1), None), X)
    assert isinstance(out, tuple) and out[1] is None
    assert infer_sharded(lambda xs: (xs,), X, gather=False)[-1] == (0, 4)
