"""-m gpu: k_base_logistic_p2, the logistic base pass that reads 2-bit rows (gnx_base_predict_packed_dev / gnx_infer_packed*).

Bar: B BIT-identical to the int8 kernels' (both compute the logits exactly in integers; the epilogue's float64 arithmetic is the
same expression per element), hence also <= 1e-12 from the reference's own output (G1, G15) and the oracle.
Reference contract: src/Base/base.py:146-180, src/Base/models.py:12-21; X values src/utils.py:153."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ga():
    import gnomix_amd
    gnomix_amd.load_library()
    return gnomix_amd


@pytest.fixture
def p2ctx(monkeypatch):
    """GNX_LR_P2=2: build the 2-bit planes whatever the run padding costs (the small windows of the test geometries would otherwise
    keep the int8 kernels); the variable is read at model load and at context creation"""
    from gnomix_amd import _lib
    monkeypatch.setenv("GNX_LR_P2", "2")
    return _lib.Context(0)


def _both(dev, X, f64=True):
    import torch
    Xt = torch.from_numpy(np.ascontiguousarray(X)).cuda()
    P = torch.from_numpy(np.asarray(dev.pack_x(X))).cuda()
    b_i8 = dev.base_predict_device(Xt, f64=f64)
    b_p2 = dev.base_predict_packed_device(P, f64=f64)
    torch.cuda.synchronize()
    return b_i8.cpu().numpy(), b_p2.cpu().numpy()


GEOMS = [
    (4037, 100, 7, 50, 24),      # default context ratio 0.5
    (4037, 100, 7, 0, 5),        # no context
    (2531, 100, 3, 30, 70),      # ratio 0.3: windows end mid-piece
    (1999, 64, 2, 32, 130),      # M multiple of 64
    (3001, 100, 12, 50, 33),     # A = 12: 24 class columns -> one tile per slot, two passes
    (2201, 100, 9, 50, 600),     # A = 9, more rows than one block
    (1801, 60, 7, 45, 50),       # ratio 0.75: R = 3 slots, 21 columns -> three passes
    (1503, 100, 16, 50, 9),      # A = 16: a full tile per slot
    (2777, 100, 5, 120, 40),     # ratio 1.2: 4 slots, reflections reach window 1
    (1237, 50, 7, 25, 600),      # many rows, short runs
    (1237, 50, 7, 25, 1),        # a single haplotype
    (937, 300, 4, 150, 66),      # W = 3 windows only
    (20500, 1000, 7, 500, 700),  # the chr22 window shape (two 256-SNP runs per piece), three row tiles, epilogue waves
    (30500, 1000, 12, 500, 1100),  # the same with 24 class columns: flat column tiles (k_base_logistic_p2f) at 256-SNP runs, two passes at 512
    (2401, 100, 8, 100, 70),     # ratio 1.0: R = 3 slots x A = 8 = 24 columns: the flat kernel with three slots of eight columns
    (3001, 100, 12, 50, 700),    # flat tiles, three row tiles of 256
    (3107, 100, 12, 50, 1),      # flat tiles, one haplotype
]


@pytest.mark.parametrize("run", ["256", "512"])   # run length of the packed rows' walk: 64 or 128 bytes per row visit (GNX_LR_P2_RUN forces)
@pytest.mark.parametrize("C,M,A,ctx,N", GEOMS)
def test_p2_is_bit_identical_to_int8(ga, oracle, p2ctx, monkeypatch, C, M, A, ctx, N, run):
    from gnomix_amd import synth
    monkeypatch.setenv("GNX_LR_P2_RUN", run)
    d = synth.synthetic_model(C=C, M=M, A=A, S=5, context=ctx, seed=C + A, smooth=None)
    X = synth.synthetic_X(N, C, seed=N, miss=0.03)
    dev = ga.DeviceModel(d, ctx=p2ctx)
    for f64 in (True, False):
        b_i8, b_p2 = _both(dev, X, f64)
        assert np.array_equal(b_i8, b_p2), ("f64" if f64 else "f32")
    ref = oracle.base_lr(X[: min(N, 64)], M, ctx, d.lr_coef, d.lr_intercept)
    assert np.max(np.abs(_both(dev, X[: min(N, 64)])[1] - ref)) < 1e-12


@pytest.mark.parametrize("tune", ["2,8,0,3,4", "2,8,4,2,3", "2,8,4,1,4", "1,8,4,4,3", "1,8,2,4,3", "1,4,0,2,3", "2,4,2,2,3"])
def test_p2_every_shape(ga, monkeypatch, tune):
    """the instantiations the dispatcher can fall back to (no epilogue waves, shallower rings, 16-row tiles), forced"""
    from gnomix_amd import synth, _lib
    monkeypatch.setenv("GNX_LR_P2", "2")
    monkeypatch.setenv("GNX_P2_TUNE", tune)
    ctx = _lib.Context(0)
    for run in ("256", "512"):
        monkeypatch.setenv("GNX_LR_P2_RUN", run)
        for (C, M, A, cx, N) in ((20500, 1000, 7, 500, 700), (6100, 200, 12, 100, 300), (1801, 60, 7, 45, 50)):
            d = synth.synthetic_model(C=C, M=M, A=A, S=5, context=cx, seed=C, smooth=None)
            X = synth.synthetic_X(N, C, seed=N, miss=0.03)
            b_i8, b_p2 = _both(ga.DeviceModel(d, ctx=ctx), X)
            assert np.array_equal(b_i8, b_p2), (tune, run, C, A)


def test_p2_goldens_of_the_reference(ga, p2ctx):
    """G1 / G15: the REFERENCE's own Base.predict_proba output, through the 2-bit pass"""
    import torch
    from conftest import load_golden
    for name, A in (("G1_lr.npz", None), ("G15_lr_binary.npz", 2)):
        g = load_golden(name)
        d = ga.GnxModelData(C=int(g["C"]), M=int(g["M"]), A=int(g["A"]) if A is None else A, S=5, context=int(g["ctx"]),
                            base_kind="logistic", lr_coef=g["coef"], lr_intercept=g["intercept"])
        dev = ga.DeviceModel(d, ctx=p2ctx)
        b_i8, b_p2 = _both(dev, g["X"])
        assert np.array_equal(b_i8, b_p2)
        assert np.max(np.abs(b_p2 - g["B"])) < 1e-12
        assert np.array_equal(np.argmax(b_p2, -1), np.argmax(g["B"], -1))


def test_p2_row_strides_offsets_and_the_value_3(ga, p2ctx):
    """packed rows at an odd stride from an odd base address (loads split by the addresser: slower, same bytes), and fields that hold
    3 (not an int8 code the reference produces, but representable): same numbers as widening to int8 first"""
    import torch
    from gnomix_amd import synth
    C, M, A, cx, N = 5037, 100, 7, 50, 77
    d = synth.synthetic_model(C=C, M=M, A=A, S=5, context=cx, seed=3, smooth=None)
    dev = ga.DeviceModel(d, ctx=p2ctx)
    X = synth.synthetic_X(N, C, seed=9, miss=0.05)
    X[::3, ::7] = 3
    P = np.asarray(dev.pack_x(X))
    ref = dev.base_predict_device(torch.from_numpy(X).cuda(), f64=True).cpu().numpy()
    for stride_extra, off in ((0, 0), (3, 1), (61, 2), (128, 0)):
        ldp = P.shape[1] + stride_extra
        buf = torch.full((N * ldp + 64,), 0xFF, dtype=torch.uint8, device="cuda")
        view = buf[off:off + N * ldp].view(N, ldp)
        view[:, :P.shape[1]] = torch.from_numpy(P).cuda()
        got = dev.base_predict_packed_device(view, f64=True).cpu().numpy()
        assert np.array_equal(got, ref), (stride_extra, off)


@pytest.mark.parametrize("smooth,A", [("xgb", 7), ("crf", 12), ("cnn", 3)])
def test_infer_packed_through_p2(ga, monkeypatch, smooth, A):
    """gnx_infer_packed (host batches on three streams, several batches forced) and gnx_infer_packed_dev with the 2-bit pass ==
    gnx_infer on int8"""
    import torch
    from gnomix_amd import synth, _lib
    monkeypatch.setenv("GNX_LR_P2", "2")
    monkeypatch.setenv("GNX_HOST_BATCH", "64")
    C, M, S, N = 6037, 100, 21, 333
    d = synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=6, seed=C, smooth=smooth)
    X = synth.synthetic_X(N, C, seed=N, miss=0.05)
    dev = ga.DeviceModel(d, ctx=_lib.Context(0))
    p_ref, l_ref = dev.infer(X)
    P = dev.pack_x(X)
    p, l = dev.infer_packed(P)
    assert p.dtype == p_ref.dtype and np.array_equal(p, p_ref) and np.array_equal(l, l_ref)
    if smooth == "xgb":
        pt, lt = dev.infer_packed_device(torch.from_numpy(np.asarray(P)).cuda())
        assert np.array_equal(pt.cpu().numpy(), p_ref) and np.array_equal(lt.cpu().numpy(), l_ref)


def test_small_windows_keep_the_int8_kernels(ga):
    """without GNX_LR_P2=2 a model whose pieces would be mostly run padding gets no 2-bit planes: packed input is widened on the
    device and runs through the int8 kernels, bit-identical as before"""
    from gnomix_amd import synth, _lib
    d = synth.synthetic_model(C=6037, M=100, A=7, S=21, n_rounds=6, seed=1)
    X = synth.synthetic_X(70, d.C, seed=2)
    dev = ga.DeviceModel(d, ctx=_lib.Context(0))
    p_ref, l_ref = dev.infer(X)
    p, l = dev.infer_packed(dev.pack_x(X))
    assert np.array_equal(p, p_ref) and np.array_equal(l, l_ref)


@pytest.mark.parametrize("A,N", [(7, 10000), (12, 4096)])
def test_p2_full_size_chr22(ga, A, N):
    """BASELINE configs[1] geometry (C = 370 500, W = 370), X generated in HBM: the 2-bit pass == the int8 pass on every one of
    the N x 370 x A probabilities, default dispatch (no environment knobs)"""
    import torch
    from gnomix_amd import synth, _lib
    d = synth.synthetic_model(C=370500, M=1000, A=A, S=75, context=500, seed=0, smooth=None)
    dev = ga.DeviceModel(d, ctx=_lib.Context(0))
    g = torch.Generator(device="cuda").manual_seed(A)
    Xt = (torch.rand((N, d.C), device="cuda", generator=g) < 0.4).to(torch.int8)
    Xt[torch.rand((N, d.C), device="cuda", generator=g) < 0.01] = 2
    Pt = dev.pack_device(Xt)
    assert torch.equal(dev.base_predict_device(Xt), dev.base_predict_packed_device(Pt))
    assert torch.equal(dev.base_predict_device(Xt[:777], f64=True), dev.base_predict_packed_device(Pt[:777], f64=True))
    # permutation of the rows permutes the outputs (row tiles, loader lanes and epilogue waves see different rows)
    perm = torch.randperm(N, device="cuda", generator=g)
    assert torch.equal(dev.base_predict_packed_device(Pt[perm].contiguous()), dev.base_predict_packed_device(Pt)[perm])


# ---------------------------------------------------------------- Gnofix on 2-bit rows ------------------
def _unpack(P, C):
    P = np.asarray(P)
    return np.stack([(P[:, c // 4] >> (2 * (c % 4))) & 3 for c in range(C)], axis=1).astype(np.int8)


def _gnofix_both(dev, X, B, max_it):
    """(X', Y, n_switches) of the int8 host entry point and of gnx_gnofix_packed_dev on the packed copy of the same rows"""
    import torch
    Xo, Y, nsw = dev.gnofix(X.copy(), B, max_it=max_it)
    Pt = torch.from_numpy(np.asarray(dev.pack_x(X))).cuda()
    Yt, nt = dev.gnofix_packed_device(Pt, torch.from_numpy(np.ascontiguousarray(B, dtype=np.float64)).cuda(), max_it=max_it)
    torch.cuda.synchronize()
    return (Xo, Y, nsw), (_unpack(Pt.cpu().numpy(), X.shape[1]), Yt.cpu().numpy(), nt.cpu().numpy())


@pytest.mark.parametrize("W,A,S,M,extra,n_ind,seed", [
    (170, 5, 75, 7, 5, 12, 0),       # windows of 7 SNPs: several windows inside one 32-bit word of the packed row
    (96, 5, 11, 208, 16, 3, 56),     # wide windows, blocks equal between the haplotypes or differing in their last SNP only
    (96, 5, 11, 208, 13, 3, 53),     # ... C not a multiple of 4: the last byte of a row is partial
    (40, 3, 9, 1000, 500, 5, 7),     # the chr22 window shape
])
def test_gnofix_packed_equals_int8(ga, W, A, S, M, extra, n_ind, seed):
    """Gnomix.phase's loop (src/model.py:188-214, src/Gnofix/gnofix.py:58-208) with X as 2-bit rows: the same labels, switch counts
    and re-phased SNPs as the int8 route (itself pinned to the reference's gnofix() by G5 and to the oracle)"""
    from gnomix_amd import synth, _lib
    C = W * M + extra
    d = ga.GnxModelData(C=C, M=M, A=A, S=S, context=0, smooth_kind="xgb")
    for k, v in synth.synthetic_trees(4, A, S * A, seed=3 + seed, thr_lo=0.0, thr_hi=0.6, leaf_scale=1.0).items():
        setattr(d, k, v)
    dev = ga.DeviceModel(d, ctx=_lib.Context(0))
    rng = np.random.RandomState(seed)
    X = rng.randint(0, 3, size=(2 * n_ind, C)).astype(np.int8)
    for i in range(n_ind):
        for u in range(W):
            kind = rng.randint(4)
            lo, hi = u * M, (u + 1) * M if u < W - 1 else C
            if kind == 0:
                X[2 * i + 1, lo:hi] = X[2 * i, lo:hi]
            elif kind == 1:
                X[2 * i + 1, lo:hi] = X[2 * i, lo:hi]
                X[2 * i + 1, hi - 1] ^= 1
    B = rng.dirichlet(np.ones(A) * 0.3, size=(2 * n_ind, W))
    (Xo, Y, nsw), (Xq, Yq, nq) = _gnofix_both(dev, X, B, 6)
    assert np.array_equal(Yq, Y) and np.array_equal(nq, nsw)
    assert np.array_equal(Xq, Xo)
    assert int(nsw.sum()) > 0


def test_gnofix_packed_goldens_G5(ga):
    """the REFERENCE's gnofix() outputs (G5) through the 2-bit route"""
    from conftest import load_golden
    from gnomix_amd import _lib
    g = load_golden("G5_gnofix.npz")
    ctx = _lib.Context(0)
    for name in ("none", "one", "two", "edges", "many", "rand"):
        prefix = "r_" if name == "rand" else "t_"
        W, A, S, C = int(g["W"]), int(g["A"]), int(g["S"]), int(g["C"])
        d = ga.GnxModelData(C=C, M=C // W, A=A, S=S, context=0, smooth_kind="xgb", tree_off=g[prefix + "tree_off"], left=g[prefix + "left"],
                            right=g[prefix + "right"], feat=g[prefix + "feat"], cond=g[prefix + "cond"], tree_class=g[prefix + "tree_class"],
                            base_score=float(g[prefix + "base_score"]))
        dev = ga.DeviceModel(d, ctx=ctx)
        X = np.stack([g[name + "_Xm"], g[name + "_Xp"]]).astype(np.int8)
        _, (Xq, Yq, nq) = _gnofix_both(dev, X, g[name + "_B"], 4 if name == "rand" else 50)
        assert np.array_equal(Xq[0], g[name + "_oXm"]) and np.array_equal(Xq[1], g[name + "_oXp"]), name
        assert np.array_equal(Yq[0], g[name + "_oYm"]) and np.array_equal(Yq[1], g[name + "_oYp"]), name
        assert int(nq[0]) == int(g[name + "_nhist"]) - 2


@pytest.mark.parametrize("C,M,A,ctx,N", [(5003, 100, 12, 50, 300), (20500, 1000, 12, 500, 520), (2401, 100, 8, 100, 33)])
def test_p2_flat_tiles_every_variant(ga, monkeypatch, C, M, A, ctx, N):
    """24 class columns: the flat-tile kernel's block shapes (4 / 2 / no epilogue waves, 3- and 2-step plane rings; GNX_P2_TUNE) and the
    slot-tile two-pass kernel it replaces (GNX_LR_P2_FLAT=0) all give the int8 kernels' B bit for bit — also with an output base that
    is not 16-byte aligned (the 1 KB stores fall back to 512-byte ones)"""
    import torch
    from gnomix_amd import synth, _lib
    monkeypatch.setenv("GNX_LR_P2", "2")
    d = synth.synthetic_model(C=C, M=M, A=A, S=5, context=ctx, seed=C, smooth=None)
    X = synth.synthetic_X(N, C, seed=N, miss=0.03)
    ref = None
    # GNX_LR_FLAGS (development switches that keep the output): bit 25 = the flat kernel declines the model as it would one with windows
    # too wide for its int32 limb pairs (the rows are widened and the int8 kernels run), 2048 = no 16-byte float32 stores, bits 16-18 / 20-22 = classes per sigmoid unit / store parts of the epilogue waves
    variants = [("1", None, 0), ("1", "2,8,2,2,3", 0), ("1", "2,8,0,2,3", 0), ("1", "2,8,4,2,2", 0), ("1", "2,8,0,2,2", 0), ("0", None, 0),
                ("1", None, 1 << 25), ("1", None, 2048), ("1", None, (1 << 16) | (4 << 20)), ("1", None, 6 << 16), ("1", None, (3 << 16) | (1 << 20)),
                ("1", "2,8,0,2,3", (1 << 25) | 2048)]
    for flat, tune, flags in variants:
        monkeypatch.setenv("GNX_LR_P2_FLAT", flat)
        monkeypatch.setenv("GNX_LR_FLAGS", str(flags))
        if tune:
            monkeypatch.setenv("GNX_P2_TUNE", tune)
        else:
            monkeypatch.delenv("GNX_P2_TUNE", raising=False)
        ctx_ = _lib.Context(0)
        dev = ga.DeviceModel(d, ctx=ctx_)
        for f64 in (True, False):
            b_i8, b_p2 = _both(dev, X, f64)
            assert np.array_equal(b_i8, b_p2), (flat, tune, flags, f64)
        if ref is None:
            ref = b_p2
        if flat == "1" and tune is None and flags == 0:      # unaligned outputs: B at odd element offsets of a larger tensor
            P = torch.from_numpy(np.asarray(dev.pack_x(X))).cuda()
            W = C // M
            big = torch.zeros(N * W * A + 1, dtype=torch.float64, device="cuda")
            dev._bind_torch_stream()
            dev.ctx.check(dev.lib.gnx_base_predict_packed_dev(dev.h, P.data_ptr(), N, P.stride(0), None, big.data_ptr() + 8))
            torch.cuda.synchronize()
            b64 = _both(dev, X, True)[1]
            assert np.array_equal(big[1:].cpu().numpy().reshape(N, W, A), b64)
            b32 = _both(dev, X, False)[1]
            for off in (1, 2, 3):           # float32: 4 / 8 / 12 bytes past a 16-byte boundary (512-byte, 1 KB pair, 512-byte stores)
                big32 = torch.zeros(N * W * A + 4, dtype=torch.float32, device="cuda")
                dev.ctx.check(dev.lib.gnx_base_predict_packed_dev(dev.h, P.data_ptr(), N, P.stride(0), big32.data_ptr() + 4 * off, None))
                torch.cuda.synchronize()
                assert np.array_equal(big32[off:off + N * W * A].cpu().numpy().reshape(N, W, A), b32), off
                assert float(big32[:off].abs().sum()) == 0.0 and float(big32[off + N * W * A:].abs().sum()) == 0.0
        dev.close()
        ctx_.close()


@pytest.mark.parametrize("C,M,A,ctx,N", [(9001, 300, 6, 450, 200), (9001, 300, 4, 750, 150), (9001, 300, 3, 1050, 100), (12001, 300, 2, 1650, 90),
                                         (3001, 100, 3, 350, 70), (3001, 100, 6, 150, 70), (2801, 128, 4, 320, 40)])
def test_p2_flat_tiles_other_class_counts(ga, monkeypatch, C, M, A, ctx, N):
    """R * A == 24 with other splits than 2 x 12 / 3 x 8: four, six, eight and twelve windows in flight (contexts of 1.5 to 5.5 windows),
    an ODD class count (parked rows A + 2 float64 apart), windows shorter than a run (several end at once: the flat kernel's rare path)
    — the flat-tile kernel, its block shapes and the slot-tile kernel against the int8 kernels, bit for bit"""
    from gnomix_amd import synth, _lib
    monkeypatch.setenv("GNX_LR_P2", "2")
    d = synth.synthetic_model(C=C, M=M, A=A, S=5, context=ctx, seed=C + A, smooth=None)
    X = synth.synthetic_X(N, C, seed=N, miss=0.03)
    for flat, tune in (("1", None), ("1", "2,8,2,2,3"), ("1", "2,8,0,2,3"), ("0", None)):
        monkeypatch.setenv("GNX_LR_P2_FLAT", flat)
        if tune:
            monkeypatch.setenv("GNX_P2_TUNE", tune)
        else:
            monkeypatch.delenv("GNX_P2_TUNE", raising=False)
        ctx_ = _lib.Context(0)
        dev = ga.DeviceModel(d, ctx=ctx_)
        for f64 in (True, False):
            b_i8, b_p2 = _both(dev, X, f64)
            assert np.array_equal(b_i8, b_p2), (flat, tune, f64)
        dev.close()
        ctx_.close()
