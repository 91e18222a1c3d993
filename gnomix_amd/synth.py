"""Synthetic models and query data of the shapes BASELINE.json names: there is no
network, so pre-trained model.pkl files and the demo VCF are unavailable; weights are random-init
of the reference's architecture (per-window liblinear-style logistic models, multi:softprob trees of
depth <= 4, tree t of round r belongs to class t % A)."""
from __future__ import annotations

import numpy as np

from .model import GnxModelData

# chr22-like shape of config 2 (W = 370 is pinned by demo.ipynb cell 13; M = 1000 is our choice)
CHR22 = dict(C=370_500, M=1000, A=7, S=75, context=500)
# synthetic per-chromosome window counts for the whole-genome config
GENOME_W = [1431, 1344, 1117, 1073, 1020, 960, 936, 840, 832, 906, 791, 873, 629, 601, 709, 670, 642, 589, 539, 541,
            314, 370]


def synthetic_trees(n_rounds, A, n_feat, depth=4, seed=0, thr_lo=0.02, thr_hi=0.5, leaf_scale=0.3,
                    p_early_leaf=0.1):
    rng = np.random.RandomState(seed)
    off, L, R, F, Cd, cls = [0], [], [], [], [], []
    for t in range(n_rounds * A):
        nodes = []

        def grow(d):
            idx = len(nodes)
            nodes.append(None)
            if d == depth or (d > 0 and rng.rand() < p_early_leaf):
                nodes[idx] = (-1, -1, 0, np.float32(rng.randn() * leaf_scale))
            else:
                f = rng.randint(n_feat)
                thr = np.float32(rng.uniform(thr_lo, thr_hi))
                l = grow(d + 1)
                r = grow(d + 1)
                nodes[idx] = (l, r, f, thr)
            return idx

        grow(0)
        for (l, r, f, c) in nodes:
            L.append(l); R.append(r); F.append(f); Cd.append(c)
        off.append(len(L))
        cls.append(t % A)
    return dict(tree_off=np.array(off, np.int32), left=np.array(L, np.int32), right=np.array(R, np.int32),
                feat=np.array(F, np.int32), cond=np.array(Cd, np.float32), tree_class=np.array(cls, np.int32))


def synthetic_model(C, M, A, S=75, context=None, n_rounds=100, depth=4, seed=0, base="logistic", smooth="xgb",
                    coef_sd=0.05, icpt_sd=0.5):
    """Random-init model of the reference architecture (LogisticRegressionBase + XGB_Smoother by default)."""
    rng = np.random.RandomState(seed)
    context = int(M * 0.5) if context is None else int(context)
    W = C // M
    rem = C - M * W
    assert rem > 0, "the reference requires C % M != 0 (gnomix.py:124-125)"
    m = GnxModelData(C=C, M=M, A=A, S=S, context=context)
    if base == "logistic":
        ldc = M + 2 * context + rem
        m.base_kind = "logistic"
        m.lr_coef = (rng.standard_normal((W, A, ldc)) * coef_sd)
        m.lr_intercept = rng.standard_normal((W, A)) * icpt_sd
    if smooth == "xgb":
        m.smooth_kind = "xgb"
        for k, v in synthetic_trees(n_rounds, A, S * A, depth=depth, seed=seed + 1).items():
            setattr(m, k, v)
    elif smooth == "cnn":
        m.smooth_kind = "cnn"
        m.cnn_weight = (rng.standard_normal((A, A, S)) * (1.0 / np.sqrt(A * S)) * 3.0).astype(np.float32)
        m.cnn_bias = (rng.standard_normal(A) * 0.2).astype(np.float32)
    elif smooth == "crf":
        m.smooth_kind = "crf"
        m.crf_state = rng.standard_normal((A, A)) * 2.0 + 4.0 * np.eye(A)
        m.crf_trans = rng.standard_normal((A, A)) * 0.5 + 3.0 * np.eye(A)
    m.population_order = ["POP%d" % a for a in range(A)]
    return m


def synthetic_X(N, C, seed=94305, miss=0.01):
    """Phased haplotypes: per-SNP allele frequency ~ U(0.05, 0.95), `miss` of the entries set to 2."""
    rng = np.random.RandomState(seed)
    p = rng.uniform(0.05, 0.95, size=C).astype(np.float32)
    X = np.empty((N, C), dtype=np.int8)
    step = max(1, (1 << 24) // C)
    for n0 in range(0, N, step):
        n1 = min(N, n0 + step)
        u = rng.random_sample((n1 - n0, C)).astype(np.float32)
        X[n0:n1] = u < p
        X[n0:n1][rng.random_sample((n1 - n0, C)) < miss] = 2
    return X


def synthetic_X_device(N, C, device, seed=94305, miss=0.01):
    """The same distribution generated directly in HBM (torch is only the allocator / RNG here)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    p = torch.empty(C, device=device, dtype=torch.float32).uniform_(0.05, 0.95, generator=g)
    X = torch.empty((N, C), device=device, dtype=torch.int8)
    step = max(1, (1 << 26) // C)
    for n0 in range(0, N, step):
        n1 = min(N, n0 + step)
        u = torch.rand((n1 - n0, C), device=device, generator=g)
        x = (u < p).to(torch.int8)
        x[torch.rand((n1 - n0, C), device=device, generator=g) < miss] = 2
        X[n0:n1] = x
    return X


def synthetic_svc_model(C, M, A, context=None, n_fit_per_class=20, sv_frac=0.7, seed=0, S=75, smooth=None):
    """Random-parameter CovRSK/SVC base of the reference architecture (src/Base/models.py:195-215):
    per window an SVC over `A*n_fit_per_class` training haplotypes with a random subset as support vectors."""
    from .convert import cov_sample
    rng = np.random.RandomState(seed)
    context = int(M * 0.5) if context is None else int(context)
    m = synthetic_model(C, M, A, S=S, context=context, seed=seed, base=None, smooth=smooth)
    m.base_kind = "covrsk"
    W, P = m.W, A * (A - 1) // 2
    m.svc = []
    n_fit = A * n_fit_per_class
    for i in range(W):
        width = m.window_width(i)
        xfit = (rng.random_sample((n_fit, width)) < rng.uniform(0.1, 0.9, size=width)).astype(np.int8)
        xfit[rng.random_sample(xfit.shape) < 0.01] = 2
        cls = np.repeat(np.arange(A), n_fit_per_class)
        sup = np.sort(np.concatenate([np.where(cls == c)[0][rng.random_sample(n_fit_per_class) < sv_frac] for c in range(A)]))
        for c in range(A):  # every class keeps at least one SV
            if not (cls[sup] == c).any():
                sup = np.sort(np.append(sup, c * n_fit_per_class))
        n_support = np.array([(cls[sup] == c).sum() for c in range(A)], np.int32)
        m.svc.append(dict(xfit=xfit, support=sup.astype(np.int32),
                          dual_coef=rng.standard_normal((A - 1, len(sup))) * 0.01,
                          intercept=rng.standard_normal(P) * 0.3, prob_a=-np.abs(rng.standard_normal(P)) - 0.5,
                          prob_b=rng.standard_normal(P) * 0.2, n_support=n_support, ms=cov_sample(width)))
    return m


def synthetic_forest_model(C, M, A, context=None, n_rounds=20, depth=4, seed=0, S=75, smooth=None, p_early_leaf=0.1,
                           missing=2):
    """Random tree-ensemble base of the reference architecture (XGBBase, src/Base/models.py:24-35: per window
    XGBClassifier(n_estimators=20, max_depth=4)): `n_rounds` rounds per window, A trees per round (one when A == 2,
    binary:logistic), split features = SNPs of the window's padded slice, thresholds 0.5 / 1.5 (what splits on
    {0, 1} data with code-2 cells produce), random default directions for the missing code."""
    rng = np.random.RandomState(seed)
    context = int(M * 0.5) if context is None else int(context)
    m = synthetic_model(C, M, A, S=S, context=context, seed=seed, base=None, smooth=smooth)
    m.base_kind = "forest"
    per_round = 1 if A == 2 else A
    off, L, R, F, Cd, Dl, cls, wt0 = [0], [], [], [], [], [], [], [0]
    for i in range(m.W):
        width = m.window_width(i)
        for t in range(n_rounds * per_round):
            nodes = []

            def grow(d):
                idx = len(nodes)
                nodes.append(None)
                if d == depth or (d > 0 and rng.rand() < p_early_leaf):
                    nodes[idx] = (-1, -1, 0, np.float32(rng.randn() * 0.2), 0)
                else:
                    f = rng.randint(width)
                    thr = np.float32(0.5 if rng.rand() < 0.85 else 1.5)
                    dl = int(rng.rand() < 0.5)
                    l = grow(d + 1)
                    r = grow(d + 1)
                    nodes[idx] = (l, r, f, thr, dl)
                return idx

            grow(0)
            for (l, r, f, c, dl) in nodes:
                L.append(l); R.append(r); F.append(f); Cd.append(c); Dl.append(dl)
            off.append(len(L))
            cls.append(t % per_round)
        wt0.append(len(off) - 1)
    m.fb_win_tree0 = np.array(wt0, np.int32)
    m.fb_tree_off = np.array(off, np.int32)
    m.fb_left, m.fb_right, m.fb_feat = np.array(L, np.int32), np.array(R, np.int32), np.array(F, np.int32)
    m.fb_cond, m.fb_default_left = np.array(Cd, np.float32), np.array(Dl, np.uint8)
    m.fb_tree_class = np.array(cls, np.int32)
    m.fb_base_score, m.fb_missing = 0.5, int(missing)
    return m


def synthetic_rforest_model(C, M, A, context=None, n_trees=20, depth=4, seed=0, S=75, smooth=None, p_early_leaf=0.1):
    """Random forest base of the reference architecture (RFBase, src/Base/models.py:54-66: per window
    RandomForestClassifier(n_estimators=20, max_depth=4)) in sklearn's tree_ arrays: thresholds 0.5 / 1.5, leaf rows =
    random class distributions (what predict_proba returns at a leaf)."""
    rng = np.random.RandomState(seed)
    context = int(M * 0.5) if context is None else int(context)
    m = synthetic_model(C, M, A, S=S, context=context, seed=seed, base=None, smooth=smooth)
    m.base_kind = "rforest"
    off, L, R, F, T, V, wt0 = [0], [], [], [], [], [], [0]
    for i in range(m.W):
        width = m.window_width(i)
        for t in range(n_trees):
            nodes = []

            def grow(d):
                idx = len(nodes)
                nodes.append(None)
                if d == depth or (d > 0 and rng.rand() < p_early_leaf):
                    nodes[idx] = (-1, -1, 0, -2.0, rng.dirichlet(np.ones(A) * 0.5))
                else:
                    f = rng.randint(width)
                    thr = 0.5 if rng.rand() < 0.85 else 1.5
                    l = grow(d + 1)
                    r = grow(d + 1)
                    nodes[idx] = (l, r, f, thr, rng.dirichlet(np.ones(A)))
                return idx

            grow(0)
            for (l, r, f, thr, v) in nodes:
                L.append(l); R.append(r); F.append(f); T.append(thr); V.append(v)
            off.append(len(L))
        wt0.append(len(off) - 1)
    m.rf_win_tree0, m.rf_tree_off = np.array(wt0, np.int32), np.array(off, np.int32)
    m.rf_left, m.rf_right, m.rf_feat = np.array(L, np.int32), np.array(R, np.int32), np.array(F, np.int32)
    m.rf_thr, m.rf_value = np.array(T, np.float64), np.array(V, np.float64)
    return m


def synthetic_smoothing_trees(n_rounds, A, S, depth=4, seed=0, reach=8, noise_leaf=0.01):
    """An ensemble that behaves like a TRAINED smoother (labels piecewise constant along the chromosome) while keeping
    the cost profile of the reference's 100-round model: the first 2*reach+1 rounds are signal trees (class c votes by
    thresholding the class-c base probability at windows w-reach..w+reach around the centre of the sliding window),
    the remaining rounds are full-depth random trees with tiny leaves."""
    rng = np.random.RandomState(seed)
    pad = (S + 1) // 2
    off, L, R, F, Cd, cls = [0], [], [], [], [], []
    n_sig = min(n_rounds, 2 * reach + 1)
    for r in range(n_sig):
        k = r - reach
        for c in range(A):
            f = (pad + k) * A + c            # slide_window is centred on w-1: feature s=pad is window w
            wgt = np.float32(0.6 / (1 + abs(k)))
            nodes = [(1, 2, f, np.float32(0.5)), (-1, -1, 0, -wgt), (3, 4, f, np.float32(0.8)), (-1, -1, 0, wgt),
                     (-1, -1, 0, np.float32(1.5) * wgt)]
            for (l, rr, ff, cc) in nodes:
                L.append(l); R.append(rr); F.append(ff); Cd.append(cc)
            off.append(len(L)); cls.append(c)
    rest = synthetic_trees(n_rounds - n_sig, A, S * A, depth=depth, seed=seed + 1, leaf_scale=noise_leaf, p_early_leaf=0.0) \
        if n_rounds > n_sig else None
    out = dict(tree_off=np.array(off, np.int32), left=np.array(L, np.int32), right=np.array(R, np.int32),
               feat=np.array(F, np.int32), cond=np.array(Cd, np.float32), tree_class=np.array(cls, np.int32))
    if rest is not None:
        base = out["tree_off"][-1]
        out = dict(tree_off=np.concatenate([out["tree_off"], rest["tree_off"][1:] + base]).astype(np.int32),
                   left=np.concatenate([out["left"], rest["left"]]), right=np.concatenate([out["right"], rest["right"]]),
                   feat=np.concatenate([out["feat"], rest["feat"]]), cond=np.concatenate([out["cond"], rest["cond"]]),
                   tree_class=np.concatenate([out["tree_class"], rest["tree_class"]]))
    return out


def synthetic_phased_individuals(n_ind, W, A, seed=0, mean_segments=3, phase_errors=2, noise=0.15):
    """Base probabilities (2*n_ind, W, A) float64 of admixed individuals whose two haplotypes carry `phase_errors`
    switch errors each (the situation Gnofix repairs): piecewise-constant true ancestry, noisy one-hot B."""
    rng = np.random.RandomState(seed)
    B = np.empty((2 * n_ind, W, A))
    for i in range(n_ind):
        ys = []
        for h in range(2):
            cuts = np.sort(rng.choice(np.arange(40, W - 40), size=rng.poisson(mean_segments - 1), replace=False)) if W > 100 else []
            y = np.empty(W, dtype=int)
            a = rng.randint(A)
            prev = 0
            for c in list(cuts) + [W]:
                y[prev:c] = a
                a = (a + 1 + rng.randint(A - 1)) % A
                prev = c
            ys.append(y)
        b = []
        for y in ys:
            p = np.full((W, A), noise / (A - 1))
            p[np.arange(W), y] = 1 - noise
            p *= rng.uniform(0.8, 1.2, size=p.shape)
            b.append(p / p.sum(1, keepdims=True))
        bm, bp = b
        for s in np.sort(rng.choice(np.arange(10, W - 10), size=phase_errors, replace=False)):
            bm, bp = np.concatenate([bm[:s], bp[s:]]), np.concatenate([bp[:s], bm[s:]])
        B[2 * i], B[2 * i + 1] = bm, bp
    return B


def write_vcf_gt2(path, G, n_samples, pos, ref, alt, chrom="22", samples=None, missing_as_dot=True, n_threads=0):
    """A synthetic phased query VCF ("throughput on synthetic phased VCFs", BASELINE.json): G (n_variants, ldg) variant-major
    2-bit rows (include/gnomix_io.h; vcfio.pack_gt2 makes them from an (N, V) matrix, DeviceModel / gnx_x_to_gt2_dev from X in
    HBM), one record per variant `chrom pos . ref alt . PASS . GT a|b ...`; the text is produced by the library's VCF writer."""
    import ctypes as C
    from . import _lib
    G = np.ascontiguousarray(G, np.uint8)
    V = G.shape[0]
    names = samples if samples is not None else ["S%d" % i for i in range(n_samples)]
    head = ("##fileformat=VCFv4.2\n##source=gnomix_amd.synth\n##contig=<ID=%s>\n" % chrom +
            '##FORMAT=<ID=GT,Number=1,Type=String,Description="Phased Genotype">\n' +
            "#" + "\t".join(["CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] + [str(s) for s in names]) + "\n").encode()
    pre = [("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT" % (chrom, p, r, a)).encode() for p, r, a in zip(np.asarray(pos).tolist(), ref, alt)]
    off = np.zeros(V + 1, np.int64)
    if V:
        np.cumsum([len(e) for e in pre], out=off[1:])
    blob = b"".join(pre)
    _lib.io_check(_lib.load().gnx_write_vcf_gt2(str(path).encode(), head, len(head), blob, off.ctypes.data, G.ctypes.data, V, G.shape[1],
                                                int(n_samples), int(bool(missing_as_dot)), int(n_threads)))
    return str(path)


def synthetic_admixed_device(n_ind, C, M, A, device, seed=0, mean_segments=3, phase_errors=0, miss=0.01, chunk=1024, freqs=None):
    """Admixed individuals at SNP level, generated in HBM (torch): per-ancestry allele frequencies f[a, j] ~ U(0.05, 0.95), every
    haplotype a mosaic of ancestry tracts (window granularity, `mean_segments` tracts on average), alleles ~ Bernoulli(f[tract
    ancestry]), `miss` of the calls set to 2 (missing); `phase_errors` switch errors per individual exchange the two haplotypes
    from a random SNP on (what Gnofix repairs).  -> X (2 * n_ind, C) int8 on `device`, y (2 * n_ind, W) int32 window labels of the
    haplotypes BEFORE the switch errors (numpy), freqs (A, C) float32 on the device (pass it back in to draw more individuals of
    the same populations)."""
    import torch
    W = C // M
    n = 2 * n_ind
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    if freqs is None:
        freqs = torch.rand((A, C), generator=g, device=device) * 0.9 + 0.05
    rng = np.random.RandomState(seed)
    y = np.empty((n, W), np.int32)
    for h in range(n):
        k = min(rng.poisson(max(mean_segments - 1, 0)), max(W - 1, 0))
        cuts = np.sort(rng.choice(np.arange(1, W), size=k, replace=False)) if k else np.empty(0, int)
        a = rng.randint(A)
        prev = 0
        for c in list(cuts) + [W]:
            y[h, prev:c] = a
            a = (a + 1 + rng.randint(max(A - 1, 1))) % A
            prev = c
    X = torch.empty((n, C), dtype=torch.int8, device=device)
    win = torch.clamp(torch.arange(C, device=device) // M, max=W - 1)          # SNP -> window (the last window takes the remainder)
    for i0 in range(0, n, chunk):
        i1 = min(n, i0 + chunk)
        yc = torch.as_tensor(y[i0:i1], device=device).long()[:, win]             # (chunk, C) ancestry of every SNP
        u = torch.rand((i1 - i0, C), generator=g, device=device)
        p = torch.zeros_like(u)
        for a in range(A):
            p = torch.where(yc == a, freqs[a].unsqueeze(0), p)
        x = (u < p).to(torch.int8)
        if miss > 0:
            x = torch.where(torch.rand((i1 - i0, C), generator=g, device=device) < miss, torch.full_like(x, 2), x)
        X[i0:i1] = x
        del yc, u, p, x
    for i in range(n_ind if phase_errors else 0):
        for s in np.sort(rng.choice(np.arange(M, C - M), size=phase_errors, replace=False)):
            t = X[2 * i, s:].clone()
            X[2 * i, s:] = X[2 * i + 1, s:]
            X[2 * i + 1, s:] = t
    return X, y, freqs


def bgzf_compress_file(src, dst, n_threads=8, level=1, piece=32 << 20):
    """`src` -> `dst` in BGZF (bgzip's container: gzip members of <= 64 KB of text with a BC extra field, an empty member at the
    end), the way the reference's demo query ships (.vcf.gz, src/utils.py:64-66); zlib on `n_threads` threads (it releases the GIL)."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    def member(data):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = co.compress(data) + co.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25) + raw +
                struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))

    def work(buf):
        return b"".join(member(buf[o:o + 65280]) for o in range(0, len(buf), 65280))
    with open(src, "rb") as fi, open(dst, "wb") as fo, ThreadPoolExecutor(max_workers=max(1, n_threads)) as pool:
        while True:
            bufs = [b for b in (fi.read(piece) for _ in range(max(1, n_threads))) if b]
            if not bufs:
                break
            for out in pool.map(work, bufs):
                fo.write(out)
        fo.write(member(b""))
    return dst
