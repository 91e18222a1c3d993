"""-m gpu: the 2-bit packed input path (gnx_pack_x / gnx_infer_packed / gnx_unpack_x_dev) and the overlapped host-pointer
pipeline.  Bar: outputs BIT-identical to the int8 entry point gnx_infer, which keeps the reference's contract
(src/utils.py:153: int8 {0,1,2}, one byte per SNP)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ga():
    import gnomix_amd
    gnomix_amd.load_library()
    return gnomix_amd


@pytest.mark.parametrize("C,M,A,S,N,smooth", [
    (6037, 100, 7, 21, 70, "xgb"),       # C % 16 = 5: the last 16-SNP group is partial
    (4112, 100, 4, 11, 33, "xgb"),       # C % 16 = 0, odd N
    (3001, 50, 12, 75, 8, "crf"),        # float64 smoother outputs
    (2049, 64, 3, 9, 129, "cnn"),
])
def test_infer_packed_is_bit_identical_to_int8(ga, C, M, A, S, N, smooth):
    from gnomix_amd import synth
    d = synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=6, seed=C, smooth=smooth)
    X = synth.synthetic_X(N, C, seed=N, miss=0.05)
    dev = ga.DeviceModel(d)
    p_ref, l_ref = dev.infer(X)
    P = dev.pack_x(X)
    assert P.dtype == np.uint8 and P.shape == (N, ((C + 15) // 16) * 4)
    p, l = dev.infer_packed(P)
    assert p.dtype == p_ref.dtype and np.array_equal(p, p_ref) and np.array_equal(l, l_ref)
    # a pageable copy with a larger row stride works as well
    P2 = np.zeros((N, P.shape[1] + 8), np.uint8)
    P2[:, :P.shape[1]] = P
    p2, l2 = dev.infer_packed(P2)
    assert np.array_equal(p2, p_ref) and np.array_equal(l2, l_ref)


def test_unpack_on_device_matches_host(ga):
    import ctypes
    import torch
    from gnomix_amd import synth, _lib
    d = synth.synthetic_model(C=1037, M=100, A=3, S=5, smooth=None)
    dev = ga.DeviceModel(d)
    for C, ldx_extra, off in ((1037, 0, 0), (64, 16, 0), (999, 3, 1)):   # aligned fast path, padded rows, misaligned buffers
        N = 19
        X = synth.synthetic_X(N, C, seed=C, miss=0.2)
        ldp = int(dev.lib.gnx_packed_row_bytes(C))
        P = np.zeros((N, ldp), np.uint8)
        assert dev.lib.gnx_pack_x(X.ctypes.data, N, C, C, P.ctypes.data, ldp, 1) == 0
        Pd = torch.zeros(N * ldp + 8, dtype=torch.uint8, device="cuda")
        Pd[off:off + N * ldp] = torch.from_numpy(P.reshape(-1)).cuda()
        ldx = C + ldx_extra
        Xd = torch.full((N * ldx + 32,), 9, dtype=torch.int8, device="cuda")
        dev._bind_torch_stream()
        rc = dev.lib.gnx_unpack_x_dev(dev.ctx.h, Pd.data_ptr() + off, N, ldp, C, Xd.data_ptr() + off, ldx)
        dev.ctx.check(rc)
        torch.cuda.synchronize()
        got = Xd[off:off + N * ldx].cpu().numpy().reshape(N, ldx)
        assert np.array_equal(got[:, :C], X)
        assert (got[:, C:] == 9).all() and (Xd[off + N * ldx:].cpu().numpy() == 9).all()   # nothing written past C


def test_host_pipeline_overlapped_equals_serial(ga, monkeypatch):
    """>= 4 batches through the three-stream pipeline (both staging halves reused) == one serial batch == packed"""
    from gnomix_amd import synth, _lib
    C, M, A, S, N = 6037, 100, 5, 21, 90
    d = synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=8, seed=8)
    X = synth.synthetic_X(N, C, seed=4, miss=0.02)
    p_ref, l_ref = ga.DeviceModel(d).infer(X)
    monkeypatch.setenv("GNX_HOST_BATCH", "14")            # 7 batches, the last one partial (6 haplotypes)
    ctx = _lib.Context(0)
    dev = ga.DeviceModel(d, ctx=ctx)
    Xp = ctx.pinned_empty(X.shape, np.int8)
    Xp[...] = X
    for src in (X, Xp):
        p, l = dev.infer(src)
        assert np.array_equal(p, p_ref) and np.array_equal(l, l_ref)
    p, l = dev.infer_packed(dev.pack_x(X))
    assert np.array_equal(p, p_ref) and np.array_equal(l, l_ref)
    p64, _ = dev.infer(X, proba_dtype=np.float64)
    assert np.array_equal(p64, p_ref.astype(np.float64))
    monkeypatch.setenv("GNX_H2D_OVERLAP", "0")
    dev2 = ga.DeviceModel(d, ctx=_lib.Context(0))
    p, l = dev2.infer(X)
    assert np.array_equal(p, p_ref) and np.array_equal(l, l_ref)


def test_gnofix_host_pipeline_equals_one_batch(ga, monkeypatch):
    """gnx_gnofix over several overlapped batches (both staging halves reused, pinned and pageable arrays, in place and on a
    copy) == one batch"""
    from gnomix_amd import synth, _lib
    W, A, S, M = 70, 5, 11, 16
    C = W * M + 5
    d = ga.GnxModelData(C=C, M=M, A=A, S=S, context=0, smooth_kind="xgb")
    for k, v in synth.synthetic_trees(3, A, S * A, seed=5, thr_lo=0.0, thr_hi=0.5, leaf_scale=1.0).items():
        setattr(d, k, v)
    rng = np.random.RandomState(12)
    n_ind = 23
    X = rng.randint(0, 2, size=(2 * n_ind, C)).astype(np.int8)
    B = rng.dirichlet(np.ones(A) * 0.3, size=(2 * n_ind, W))
    X_ref, Y_ref, ns_ref = ga.DeviceModel(d).gnofix(X, B, max_it=6)
    assert int(ns_ref.sum()) > 0 and not np.array_equal(X_ref, X)
    monkeypatch.setenv("GNX_HOST_BATCH", "10")            # 5 individuals per batch: 5 batches, the last one partial
    ctx = _lib.Context(0)
    dev = ga.DeviceModel(d, ctx=ctx)
    Xo, Y, ns = dev.gnofix(X, B, max_it=6)
    assert np.array_equal(Xo, X_ref) and np.array_equal(Y, Y_ref) and np.array_equal(ns, ns_ref)
    Xp = ctx.pinned_empty(X.shape, np.int8); Xp[...] = X
    Bp = ctx.pinned_empty(B.shape, np.float64); Bp[...] = B
    Yp = ctx.pinned_empty(Y_ref.shape, np.int32); nsp = ctx.pinned_empty(ns_ref.shape, np.int32)
    Xo, Y, ns = dev.gnofix(Xp, Bp, max_it=6, inplace=True, out=(Yp, nsp))
    assert Xo is Xp and Y is Yp and ns is nsp
    assert np.array_equal(Xp, X_ref) and np.array_equal(Yp, Y_ref) and np.array_equal(nsp, ns_ref)
    with pytest.raises(ValueError):
        dev.gnofix(X[:, ::-1], B, inplace=True)
    with pytest.raises(ValueError):
        dev.gnofix(X, B, out=(Yp[:-2], nsp))
    monkeypatch.setenv("GNX_H2D_OVERLAP", "0")
    Xo, Y, ns = ga.DeviceModel(d, ctx=_lib.Context(0)).gnofix(X, B, max_it=6)
    assert np.array_equal(Xo, X_ref) and np.array_equal(Y, Y_ref) and np.array_equal(ns, ns_ref)


def test_infer_packed_rejects_bad_arguments(ga):
    from gnomix_amd import synth
    d = synth.synthetic_model(C=1037, M=100, A=3, S=5, n_rounds=2)
    dev = ga.DeviceModel(d)
    with pytest.raises(ValueError):
        dev.infer_packed(np.zeros((4, 100), np.uint8))         # fewer than ceil(C/4) bytes per row
    X = synth.synthetic_X(4, 1037, seed=1)
    X[0, 0] = 5
    with pytest.raises(ga.GnxError):
        dev.pack_x(X)
    p, l = dev.infer_packed(np.zeros((0, 260), np.uint8))
    assert p.shape == (0, 10, 3) and l.shape == (0, 10)
