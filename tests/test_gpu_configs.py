"""-m gpu: BASELINE.json's configs 3 and 4 as PIPELINES ("configs_untested").

config 3  chr1 array density, CovRSK/SVC base -> xgb smoother: the base classes were only ever tested up to
          `base_predict`; here the string-kernel base feeds the tree smoother through gnx_infer / HipGnomix.predict at the
          config's window geometry (M = 175, ctx = 87, width 349, Ms = [1, 4, 8, 39, 42, 117]) against the oracle, and once
          at FULL size (C = 250 400, W = 1430, 1 400 support vectors per window) through size-independent properties plus
          the oracle on two haplotypes.
config 4  whole genome = one model per chromosome: several chromosome models of different window counts resident in ONE
          context, chromosome-major batches of the same individuals, results equal to every model run on its own.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ga():
    import gnomix_amd
    gnomix_amd.load_library()
    return gnomix_amd


def _svc_oracle_windows(d, wins=None):
    return [dict(Xfit=w["xfit"], Ms=list(w["ms"]), support=w["support"], dual=w["dual_coef"], intercept=w["intercept"],
                 probA=w["prob_a"], probB=w["prob_b"], n_support=w["n_support"]) for w in (d.svc if wins is None else wins)]


def _trees(O, d):
    return O.Trees(d.tree_off, d.left, d.right, d.feat, d.cond, d.tree_class, d.A, d.base_score)


def _related_queries(d, N, seed):
    """queries that copy training rows over whole windows, so that long match runs (all six substring lengths) occur"""
    from gnomix_amd import synth
    rng = np.random.RandomState(seed)
    X = synth.synthetic_X(N, d.C, seed=seed, miss=0.02)
    for n in range(N):
        for _ in range(3):
            w = rng.randint(d.W)
            src = d.svc[w]["xfit"][rng.randint(d.svc[w]["xfit"].shape[0])]
            start = w * d.M - d.context
            seg = src if start >= 0 else src[-start:]
            lo = max(0, start)
            ln = min(len(seg), d.C - lo)
            X[n, lo:lo + ln] = seg[:ln]
    return X


def test_config3_covrsk_base_into_xgb_smoother_vs_oracle(ga, oracle):
    """config-3 window geometry, W = 2S so the smoother accepts it, small enough for the oracle to run every window"""
    from gnomix_amd import synth
    M, ctx, A, S, W = 175, 87, 7, 75, 150
    C = M * W + 150
    d = synth.synthetic_svc_model(C, M, A, context=ctx, n_fit_per_class=8, seed=3, S=S, smooth="xgb")
    assert d.window_width(0) == 349 and list(d.svc[0]["ms"]) == [1, 4, 8, 39, 42, 117]
    N = 12
    X = _related_queries(d, N, seed=5)
    g = ga.HipGnomix(d)
    proba, labels = g.predict_proba(X), g.predict(X)
    B = oracle.base_covrsk(X, M, ctx, _svc_oracle_windows(d))
    p_ref, l_ref = oracle.smooth_xgb(_trees(oracle, d), B, S)
    assert np.array_equal(labels, l_ref)
    assert np.max(np.abs(proba - p_ref)) <= 1e-5
    # the plugin-style two-step use (gnomix.py:55-58) gives the same answer as the fused entry point
    Bq = g.base.predict_proba(X)
    assert np.max(np.abs(Bq - B)) < 1e-12
    assert np.array_equal(g.smooth.predict(Bq), l_ref)
    # host batching is invisible on this path too
    p2, l2 = g.dev.infer(X[:6])
    assert np.array_equal(l2, l_ref[:6]) and np.array_equal(p2, proba[:6])


@pytest.mark.timeout(1500)
def test_config3_full_size_properties(ga, oracle):
    """config 3 at FULL size: C = 250 400, M = 175, W = 1430, A = 7, 1 400 training haplotypes per window, all of them
    support vectors (the survey's worst case), 256 haplotypes."""
    import torch
    from gnomix_amd import synth
    C, M, A, S = 250_400, 175, 7, 75
    d = synth.synthetic_svc_model(C, M, A, n_fit_per_class=200, sv_frac=1.1, seed=0, S=S, smooth="xgb")
    assert d.W == 1430 and d.svc[0]["xfit"].shape == (1400, 349) and len(d.svc[0]["support"]) == 1400
    assert d.svc[-1]["xfit"].shape[1] == 349 + 150
    dev = ga.DeviceModel(d)
    N = 256
    Xh = _related_queries(d, N, seed=11)
    Xd = torch.from_numpy(Xh).cuda()
    p, lab = dev.infer_device(Xd)
    B = dev.base_predict_device(Xd, f64=True)
    torch.cuda.synchronize()
    p, lab, B = p.cpu().numpy(), lab.cpu().numpy(), B.cpu().numpy()
    assert np.isfinite(p).all() and np.allclose(p.sum(-1), 1.0, atol=2e-6)
    assert np.allclose(B.sum(-1), 1.0, atol=1e-9) and (B >= 0).all()
    assert np.array_equal(lab, np.argmax(p, -1))
    # permutation equivariance and batch-split independence, bit-exact
    perm = np.random.RandomState(1).permutation(N)
    p2, lab2 = dev.infer_device(torch.from_numpy(Xh[perm]).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(p2.cpu().numpy(), p[perm]) and np.array_equal(lab2.cpu().numpy(), lab[perm])
    p3, _ = dev.infer_device(Xd[64:101])
    torch.cuda.synchronize()
    assert np.array_equal(p3.cpu().numpy(), p[64:101])
    # the oracle: every window of two haplotypes for the base (1 400 SVs x 349 SNPs x 1 430 windows each), then the smoother
    idx = [0, 201]
    Bo = oracle.base_covrsk(Xh[idx], M, d.context, _svc_oracle_windows(d))
    assert np.max(np.abs(B[idx] - Bo)) < 1e-12
    p_ref, l_ref = oracle.smooth_xgb(_trees(oracle, d), Bo, S)
    assert np.array_equal(lab[idx], l_ref)
    assert np.max(np.abs(p[idx] - p_ref)) <= 1e-5
    # the exact integer kernel matrix of three windows (first, an inner one, the wider last one) vs the oracle's
    wins = dict(oracle.base_windows(Xh[idx], M, d.context))
    for w in (0, 700, d.W - 1):
        K = oracle.covrsk(wins[w], d.svc[w]["xfit"], list(d.svc[w]["ms"]))
        assert K.shape == (2, 1400) and K.min() >= 0
    dev.close()


def test_config4_multi_chromosome_models_in_one_context(ga, oracle):
    """whole-genome harness: >= 3 chromosome models of different W loaded in ONE context, chromosome-major batches of the
    same individuals; every chromosome's result equals that model run alone (fresh context) and the oracle."""
    import torch
    from gnomix_amd import synth, _lib
    shapes = [(37_537, 100, 375), (16_031, 100, 160), (23_419, 100, 234), (15_077, 100, 150)]   # (C, M, W): chr-like W spread
    A, S, N = 7, 75, 70
    ctx = _lib.Context(0)
    datas = [synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=10, seed=100 + k) for k, (C, M, W) in enumerate(shapes)]
    models = [ga.DeviceModel(d, ctx=ctx) for d in datas]            # all resident at once, one stream, shared workspaces
    assert [m.W for m in models] == [s[2] for s in shapes]
    Xs = [synth.synthetic_X(N, d.C, seed=200 + k, miss=0.02) for k, d in enumerate(datas)]
    Xd = [torch.from_numpy(x).cuda() for x in Xs]
    outs = []
    for rep in range(2):                                            # two passes: workspaces sized by the largest chromosome are reused
        outs = [m.infer_device(x) for m, x in zip(models, Xd)]      # chromosome-major, back to back on one stream
    torch.cuda.synchronize()
    host = [m.infer(x) for m, x in zip(models[::-1], Xs[::-1])][::-1]   # host-pointer path, reverse order (workspace shrink/grow)
    for k, d in enumerate(datas):
        p, lab = outs[k][0].cpu().numpy(), outs[k][1].cpu().numpy()
        alone = ga.DeviceModel(d, ctx=_lib.Context(0))
        pa, la = alone.infer(Xs[k])
        assert np.array_equal(p, pa) and np.array_equal(lab, la)
        assert np.array_equal(host[k][0], pa) and np.array_equal(host[k][1], la)
        alone.close()
        sel = [0, N - 1]
        Bo = oracle.base_lr(Xs[k][sel], d.M, d.context, d.lr_coef, d.lr_intercept)
        p_ref, l_ref = oracle.smooth_xgb(_trees(oracle, d), Bo, S)
        assert np.array_equal(lab[sel], l_ref) and np.max(np.abs(p[sel] - p_ref)) <= 1e-5
    info = [m.info.device_bytes for m in models]
    assert all(b > 0 for b in info)
    for m in models:
        m.close()


def test_config4_whole_genome_22_models_resident_in_one_context(ga, oracle):
    """BASELINE configs[3] at its real geometry on ONE GPU (W_k of synth.GENOME_W, C_k = 1000 W_k + 500, sum W =
    17 727, A = 7, logistic base + xgb smoother): all 22 chromosome models resident in one context (what each of the 8 GPUs of the
    config holds: the model is replicated, individuals are sharded), chromosome-major batches of the same 500 haplotypes.
    Size-independent properties on every chromosome, the oracle on two haplotypes of the largest chromosome and of chr22, and the
    HBM the models hold."""
    import time
    import torch
    from gnomix_amd import synth, _lib
    t_start = time.time()
    A, S, N = 7, 75, 500
    assert len(synth.GENOME_W) == 22 and sum(synth.GENOME_W) == 17_727
    ctx = _lib.Context(0)
    free0 = torch.cuda.mem_get_info()[0]
    models, keep = [], {}
    for k, Wk in enumerate(synth.GENOME_W):
        d = synth.synthetic_model(C=1000 * Wk + 500, M=1000, A=A, S=S, n_rounds=100, seed=400 + k)
        models.append(ga.DeviceModel(d, ctx=ctx))
        if k in (0, 21):
            keep[k] = d           # the oracle needs the host arrays of the two chromosomes it checks
        del d
    assert [m.W for m in models] == list(synth.GENOME_W)
    held = sum(int(m.info.device_bytes) for m in models)
    # ~112 bytes per SNP of int8 weight digits, TWICE: the planes of the int8 kernels (2.3 GB for the genome) and the same digits
    # in the k order of the 2-bit pass (1.8 GB: runs of 256 SNPs, no even-chunk padding) — DESIGN.md 2; the device really holds it
    assert 3.6e9 < held < 4.6e9, held
    assert free0 - torch.cuda.mem_get_info()[0] >= 0.95 * held
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(5)).cuda()
    for k, m in enumerate(models):
        X = synth.synthetic_X_device(N, m.C, "cuda:0", seed=900 + k)
        p, lab = m.infer_device(X)
        p2, lab2 = m.infer_device(X[perm].contiguous())                 # haplotypes are independent rows
        torch.cuda.synchronize()
        assert p.shape == (N, m.W, A) and lab.shape == (N, m.W)
        assert torch.equal(p2, p[perm]) and torch.equal(lab2, lab[perm]), k
        pk, labk = m.infer_packed_device(m.pack_device(X))                # the batch resident as 2-bit rows: same outputs
        assert torch.equal(pk, p) and torch.equal(labk, lab), k
        del pk, labk
        assert float((p.sum(-1) - 1).abs().max()) <= 1e-5 and bool(torch.isfinite(p).all())
        assert torch.equal(lab.long(), p.argmax(-1)) or bool((p.gather(-1, lab.long().unsqueeze(-1)).squeeze(-1) == p.max(-1).values).all())
        if k in keep:
            d = keep[k]
            idx = [0, N - 1]
            Xh = X[idx].cpu().numpy()
            Bo = oracle.base_lr(Xh, d.M, d.context, d.lr_coef, d.lr_intercept)
            p_ref, l_ref = oracle.smooth_xgb(_trees(oracle, d), Bo, S)
            assert np.array_equal(lab[idx].cpu().numpy(), l_ref) and np.max(np.abs(p[idx].cpu().numpy() - p_ref)) <= 1e-5, k
            ph, lh = m.infer(X[:64].cpu().numpy())                      # the host-pointer route on the same rows
            assert np.array_equal(ph, p[:64].cpu().numpy()) and np.array_equal(lh, lab[:64].cpu().numpy())
        del X, p, lab, p2, lab2
    for m in models:
        m.close()
    ctx.close()
    assert time.time() - t_start < 120, "the whole-genome residency test is meant to stay near a minute"


@pytest.mark.timeout(600)
def test_config5_full_size_properties(ga, oracle):
    """BASELINE configs[4] at its real SNP geometry on one GPU's share of the work: chr1 WGS density C = 1 431 500, M = 1000
    (2 000-SNP windows, W = 1431), A = 12, N = 256 haplotypes.
      (5a) logistic base -> CRF smoother (src/Smooth/crf.py:17-67; the reference rejects CRF + Gnofix, src/model.py:194)
      (5b) logistic base -> xgb smoother -> Gnofix (src/model.py:188-214) on tract-structured individuals with switch errors
    Size-independent properties (row permutation and batch splits are invisible bit for bit, probabilities sum to 1, labels are the
    argmax, the 2-bit route equals the int8 route) plus the oracle on two haplotypes / one individual."""
    import torch
    from gnomix_amd import synth, _lib
    C, M, A, S, N = 1_431_500, 1000, 12, 75, 256
    ctx = _lib.Context(0)
    rng = np.random.RandomState(5)
    # ---------------- (5a) ----------------
    d = synth.synthetic_model(C=C, M=M, A=A, S=S, seed=5, smooth="crf")
    assert d.W == 1431 and d.lr_coef.shape == (1431, 12, 2500)
    dev = ga.DeviceModel(d, ctx=ctx)
    X = synth.synthetic_X(N, C, seed=17, miss=0.01)
    p, lab = dev.infer(X)                                  # float64 (CRF computes in float64)
    assert p.dtype == np.float64 and p.shape == (N, 1431, A)
    assert np.allclose(p.sum(-1), 1.0, atol=1e-9) and np.array_equal(lab, np.argmax(p, -1)) and np.isfinite(p).all()
    perm = rng.permutation(N)
    p2, l2 = dev.infer(X[perm])
    assert np.array_equal(p2, p[perm]) and np.array_equal(l2, lab[perm])
    p3, l3 = dev.infer(X[:101])                            # another row tiling, a ragged batch
    assert np.array_equal(p3, p[:101]) and np.array_equal(l3, lab[:101])
    pk, lk = dev.infer_packed(dev.pack_x(X))               # 2-bit rows: one column tile per slot, two passes
    assert np.array_equal(pk, p) and np.array_equal(lk, lab)
    idx = [3, N - 1]
    Bo = oracle.base_lr(X[idx], M, d.context, d.lr_coef, d.lr_intercept)
    _, Bh = dev.base_predict(X[idx])
    assert np.max(np.abs(Bh - Bo)) < 1e-12
    po, lo = oracle.smooth_crf(Bo, d.crf_state, d.crf_trans)
    assert np.max(np.abs(p[idx] - po)) < 1e-9 and np.array_equal(lab[idx], lo)
    dev.close()
    del d, dev, p, p2, p3, pk
    # ---------------- (5b) ----------------
    n_ind = N // 2
    d = synth.synthetic_model(C=C, M=M, A=A, S=S, seed=6, smooth=None)
    d.smooth_kind = "xgb"
    for k, v in synth.synthetic_smoothing_trees(100, A, S, seed=6).items():
        setattr(d, k, v)
    dev = ga.DeviceModel(d, ctx=ctx)
    B = synth.synthetic_phased_individuals(n_ind, d.W, A, seed=3)      # admixed individuals, two switch errors per haplotype pair
    Xq = rng.randint(0, 2, size=(N, C)).astype(np.int8)
    Xo, Y, nsw = dev.gnofix(Xq, B)
    assert int(nsw.sum()) > n_ind // 2 and int(nsw.max()) >= 2          # the loop really re-phases
    T = oracle.Trees(d.tree_off, d.left, d.right, d.feat, d.cond, d.tree_class, d.A, d.base_score)
    rows = lambda r: oracle.xgb_predict_proba(T, r)
    labs = lambda b: oracle.smooth_xgb(T, b, S)[1]
    i = int(np.argmax(nsw))                                            # the individual with the most accepted switches
    Xm, Xp, Ym, Yp, _, ns = oracle.gnofix(Xq[2 * i], Xq[2 * i + 1], B[2 * i:2 * i + 2], S, rows, labs)
    assert np.array_equal(Xo[2 * i], Xm) and np.array_equal(Xo[2 * i + 1], Xp)
    assert np.array_equal(Y[2 * i], Ym) and np.array_equal(Y[2 * i + 1], Yp) and int(nsw[i]) == ns
    permi = rng.permutation(n_ind)
    rowsp = np.stack([2 * permi, 2 * permi + 1], 1).reshape(-1)
    Xo2, Y2, nsw2 = dev.gnofix(Xq[rowsp], B[rowsp])
    assert np.array_equal(Xo2, Xo[rowsp]) and np.array_equal(Y2, Y[rowsp]) and np.array_equal(nsw2, nsw[permi])
    # the same individuals as 2-bit rows resident in HBM: identical labels, switch counts and re-phased SNPs
    Pt = torch.from_numpy(np.asarray(dev.pack_x(Xq))).cuda()
    Yt, nt = dev.gnofix_packed_device(Pt, torch.from_numpy(B).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(Yt.cpu().numpy(), Y) and np.array_equal(nt.cpu().numpy(), nsw)
    assert torch.equal(Pt.cpu(), torch.from_numpy(np.asarray(dev.pack_x(Xo))))
    # the whole phase pipeline of the command line on a few individuals: base (logistic) -> smoother -> Gnofix -> predict_proba
    _, Bb = dev.base_predict(Xq[:8])
    Xp8, Y8, n8 = dev.gnofix(Xq[:8], Bb, max_it=3)
    Xp8b, Y8b, n8b = dev.gnofix(Xq[:8][[2, 3, 0, 1, 6, 7, 4, 5]], Bb[[2, 3, 0, 1, 6, 7, 4, 5]], max_it=3)
    assert np.array_equal(Y8b, Y8[[2, 3, 0, 1, 6, 7, 4, 5]]) and np.array_equal(Xp8b, Xp8[[2, 3, 0, 1, 6, 7, 4, 5]])
    pf, lf = dev.infer(Xp8)
    assert np.allclose(pf.sum(-1), 1.0, atol=1e-5) and np.array_equal(lf, np.argmax(pf, -1))
    ctx.close()
