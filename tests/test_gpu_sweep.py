"""-m gpu: launch-geometry sweep.  For every (ancestries, window count, smoother) the large-N launch configuration of the
whole path must agree BIT-exactly with the small-N one on the same haplotypes (integer logits and in-order float32 tree sums
are tiling-independent by construction), and with the oracle on a few rows."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("A,W,smooth", list(itertools.product([2, 7, 12, 20, 32], [370, 1431], ["xgb", "crf"])))
def test_large_vs_small_launch_geometry(oracle, A, W, smooth):
    import gnomix_amd
    from gnomix_amd import synth
    gnomix_amd.load_library()
    M, ctx, S, N = 64, 32, 75, 700
    C = W * M + 37
    d = synth.synthetic_model(C=C, M=M, A=A, S=S, context=ctx, seed=A + W, smooth=smooth, n_rounds=8)
    X = synth.synthetic_X(N, C, seed=3, miss=0.02)
    dev = gnomix_amd.DeviceModel(d)
    p_big, l_big = dev.infer(X)            # 256-haplotype base tiles, 8-haplotype smoother blocks
    p_small, l_small = dev.infer(X[:40])   # 64-haplotype base tiles
    assert np.array_equal(l_big[:40], l_small)
    assert np.array_equal(p_big[:40], p_small)
    Bo = oracle.base_lr(X[:3], M, ctx, d.lr_coef, d.lr_intercept)
    if smooth == "xgb":
        T = oracle.Trees(d.tree_off, d.left, d.right, d.feat, d.cond, d.tree_class, d.A, d.base_score)
        po, lo = oracle.smooth_xgb(T, Bo, S)
    else:
        po, lo = oracle.smooth_crf(Bo, d.crf_state, d.crf_trans)
    assert np.array_equal(l_big[:3], lo)
    assert np.max(np.abs(p_big[:3] - po)) < 1e-5   # the north star's probability bar
    dev.close()
