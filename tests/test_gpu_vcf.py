"""-m gpu: the file path on the device (include/gnomix_io.h): gt2 <-> X kernels against their numpy statement, and the
pipelines  parsed VCF -> outputs  against the reference's own sequence vcf_to_npy -> predict_proba / phase
(gnomix.py:48-72) run through the int8 entry points.  Bar: bit-identical (integers and the same kernels behind both)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ga():
    import gnomix_amd
    gnomix_amd.load_library()
    return gnomix_amd


def _rows(rng, V, N, p3=0.02):
    """random variant-major 2-bit rows with a canonical stride; tail fields zero"""
    code = rng.choice(4, size=(V, N), p=[0.45, 0.45, 0.1 - p3, p3]).astype(np.uint8)
    ldg = (N + 15) // 16 * 4
    pad = np.zeros((V, ldg * 4), np.uint8)
    pad[:, :N] = code
    q = pad.reshape(V, ldg, 4)
    return (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8), code


def _x_from(code, src):
    """numpy statement of k_gt2_to_x (= vcf_to_npy's fill / flip / missing rule, src/utils.py:125-151)"""
    N = code.shape[1]
    X = np.full((N, len(src)), 2, np.int8)
    have = src >= 0
    v, flip = src[have] & 0x3FFFFFFF, (src[have] >> 30) & 1
    c = code[v].T.astype(np.int8)
    X[:, have] = np.where(c >= 2, 2, np.where(flip[None, :] == 1, 1 - c, c))
    return X


@pytest.mark.parametrize("V,N,C,ldx_extra,n0", [(300, 70, 257, 0, 0), (64, 1030, 64, 0, 0), (500, 2100, 1037, 27, 0), (200, 48, 130, 62, 16),
                                                (200, 52, 131, 5, 4), (90, 6, 1, 0, 0), (1500, 4096 + 20, 300, 212, 1024)])
def test_gt2_to_x_and_back(ga, V, N, C, ldx_extra, n0):
    import torch
    from gnomix_amd import _lib
    rng = np.random.default_rng(V * N + C)
    G, code = _rows(rng, V, N)
    src = rng.integers(0, V, C).astype(np.int32)
    src |= (rng.random(C) < 0.2).astype(np.int32) << 30
    src[rng.random(C) < 0.15] = -1
    ctx = _lib.default_context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ldg = G.shape[1]
    n = N - n0
    ldx = (C + ldx_extra) if ldx_extra else (C + 63) // 64 * 64
    Gd = torch.from_numpy(G).cuda()
    sd = torch.from_numpy(src).cuda()
    Xd = torch.full((n * ldx + 64,), 9, dtype=torch.int8, device="cuda")
    base = Xd.data_ptr() + (0 if ldx % 16 == 0 else 3)      # a misaligned buffer takes the byte-store variant
    ctx.check(ctx.lib.gnx_gt2_to_x_dev(ctx.h, Gd.data_ptr(), V, ldg, n0, n, sd.data_ptr(), C, base, ldx))
    torch.cuda.synchronize()
    off = base - Xd.data_ptr()
    got = Xd[off:off + n * ldx].cpu().numpy().reshape(n, ldx)
    want = _x_from(code, src)[n0:]
    assert np.array_equal(got[:, :C], want)
    assert (got[:, C:] == 9).all() and (Xd[off + n * ldx:].cpu().numpy() == 9).all()      # nothing outside [0, C) of a row
    # the way back: columns `cols` of X as 2-bit rows
    cols = np.sort(rng.choice(C, max(1, C // 2), replace=False)).astype(np.int32)
    cd = torch.from_numpy(cols).cuda()
    Go = torch.zeros((len(cols), ldg), dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.gnx_x_to_gt2_dev(ctx.h, base, n, ldx, n0, cd.data_ptr(), len(cols), Go.data_ptr(), ldg))
    torch.cuda.synchronize()
    Gh = Go.cpu().numpy()
    back = np.stack([(Gh[:, h // 4] >> (2 * (h % 4))) & 3 for h in range(N)], axis=0).astype(np.int8)     # (N, len(cols))
    assert np.array_equal(back[n0:], want[:, cols]) and not back[:n0].any()
    # argument checks
    assert ctx.lib.gnx_gt2_to_x_dev(ctx.h, Gd.data_ptr(), V, ldg, 2, n, sd.data_ptr(), C, base, ldx) == _lib.GNX_EINVAL
    assert ctx.lib.gnx_gt2_to_x_dev(ctx.h, Gd.data_ptr(), V, ldg, 0, 4 * ldg + 1, sd.data_ptr(), C, base, ldx) == _lib.GNX_EINVAL
    ctx.reset_stream()


@pytest.mark.parametrize("V,N,C,ldp_extra,n0", [(300, 70, 257, 0, 0), (64, 1030, 64, 0, 0), (500, 2100, 1037, 27, 0), (200, 48, 130, 62, 16),
                                                (200, 52, 131, 5, 4), (90, 6, 1, 0, 0), (1500, 4096 + 20, 300, 212, 1024)])
def test_gt2_to_p2(ga, V, N, C, ldp_extra, n0):
    """k_gt2_to_p2: the haplotype-major matrix as 2-bit rows == gnx_pack_x of the int8 matrix k_gt2_to_x builds (numpy statement)"""
    import torch
    from gnomix_amd import _lib
    rng = np.random.default_rng(V * N + C + 1)
    G, code = _rows(rng, V, N)
    src = rng.integers(0, V, C).astype(np.int32)
    src |= (rng.random(C) < 0.2).astype(np.int32) << 30
    src[rng.random(C) < 0.15] = -1
    ctx = _lib.default_context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ldg = G.shape[1]
    n = N - n0
    canon = (C + 15) // 16 * 4
    ldp = canon + ldp_extra
    Gd = torch.from_numpy(G).cuda()
    sd = torch.from_numpy(src).cuda()
    Pd = torch.full((n * ldp + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    base = Pd.data_ptr() + (0 if ldp % 4 == 0 else 1)        # a misaligned buffer takes the byte-store variant
    ctx.check(ctx.lib.gnx_gt2_to_p2_dev(ctx.h, Gd.data_ptr(), V, ldg, n0, n, sd.data_ptr(), C, base, ldp))
    torch.cuda.synchronize()
    off = base - Pd.data_ptr()
    got = Pd[off:off + n * ldp].cpu().numpy().reshape(n, ldp)
    want = _x_from(code, src)[n0:]
    fields = np.stack([(got[:, c // 4] >> (2 * (c % 4))) & 3 for c in range(C)], axis=1).astype(np.int8)
    assert np.array_equal(fields, want)
    nb = (C + 3) // 4
    if C % 4:
        assert not (got[:, nb - 1] >> (2 * (C % 4))).any()              # the fields past C of the last byte are zero
    assert (got[:, canon:] == 0xEE).all() and (Pd[off + n * ldp:].cpu().numpy() == 0xEE).all()   # nothing outside the canonical row
    assert ctx.lib.gnx_gt2_to_p2_dev(ctx.h, Gd.data_ptr(), V, ldg, 2, n, sd.data_ptr(), C, base, ldp) == _lib.GNX_EINVAL
    assert ctx.lib.gnx_gt2_to_p2_dev(ctx.h, Gd.data_ptr(), V, ldg, 0, n, sd.data_ptr(), C, base, (C + 3) // 4 - 1) == _lib.GNX_EINVAL
    ctx.reset_stream()


def _query(tmp_path, d, n_ind, rng, drop=200, flip_frac=0.05, miss=0.02):
    """a model with SNP metadata + a query VCF that lacks `drop` model SNPs, has extra SNPs of its own, REF mismatches and
    missing calls; returns (vcf path, the matrix the reference's vcf_to_npy builds from it)"""
    from gnomix_amd import synth, vcfio
    d.snp_pos = np.sort(rng.choice(np.arange(10_000, 9_000_000), size=d.C, replace=False))
    d.snp_ref = rng.choice(list("ACGT"), size=d.C)
    d.snp_alt = rng.choice(list("ACGT"), size=d.C)
    keep = np.sort(rng.choice(d.C, size=d.C - drop, replace=False))
    extra = np.setdiff1d(rng.choice(np.arange(10_000, 9_000_000), size=300), d.snp_pos)
    qpos = np.concatenate([d.snp_pos[keep], extra])
    qref = np.concatenate([d.snp_ref[keep], rng.choice(list("ACGT"), len(extra))])
    fl = rng.random(len(keep)) < flip_frac
    qref[:len(keep)][fl] = np.where(qref[:len(keep)][fl] == "A", "C", "A")
    order = np.argsort(qpos)
    Xq = synth.synthetic_X(2 * n_ind, len(qpos), seed=int(rng.integers(1 << 30)), miss=miss)
    p = synth.write_vcf_gt2(str(tmp_path / "q.vcf"), vcfio.pack_gt2(Xq[:, order]), n_ind, qpos[order], qref[order], ["N"] * len(qpos), chrom="22")
    return p


@pytest.mark.parametrize("p2", ["1", "2"])   # "2": the logistic pass reads the 2-bit rows k_gt2_to_p2 builds (GNX_LR_P2, forced for small windows)
@pytest.mark.parametrize("C,M,A,S,n_ind,smooth", [(6037, 100, 7, 21, 35, "xgb"), (4112, 100, 4, 11, 17, "xgb"), (3001, 50, 12, 75, 4, "crf"),
                                                  (2049, 64, 3, 9, 64, "cnn")])
def test_infer_gt2_equals_vcf_to_npy_then_infer(ga, tmp_path, monkeypatch, C, M, A, S, n_ind, smooth, p2):
    from gnomix_amd import synth, vcfio, _lib
    monkeypatch.setenv("GNX_LR_P2", p2)
    rng = np.random.default_rng(C)
    d = synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=6, seed=C, smooth=smooth)
    p = _query(tmp_path, d, n_ind, rng)
    dev = ga.DeviceModel(d)
    vcf = vcfio.read_vcf(p, chm="22", ctx=dev.ctx)
    assert vcf.info.gt2_pinned == 1 and vcf.info.n_fast_lines == vcf.n_variants
    X, vi, fi = vcfio.vcf_to_npy(vcf, d.snp_pos, d.snp_ref, return_idx=True, verbose=False)      # the reference's route
    assert (X == 2).any() and X.shape == (2 * n_ind, C)
    p_ref, l_ref = dev.infer(X)
    src, vi2, fi2 = vcfio.column_map(vcf, d.snp_pos, d.snp_ref, verbose=False)
    assert np.array_equal(vi, vi2) and np.array_equal(fi, fi2) and (src >> 30 == 1).any()
    pr, lb = dev.infer_gt2(vcf.gt2, 2 * n_ind, src)
    assert pr.dtype == p_ref.dtype and np.array_equal(pr, p_ref) and np.array_equal(lb, l_ref)
    # several haplotype batches through both output halves, pageable genotype rows, labels only
    monkeypatch.setenv("GNX_HOST_BATCH", "12")
    ctx = _lib.Context(0)
    dev2 = ga.DeviceModel(d, ctx=ctx)
    pr2, lb2 = dev2.infer_gt2(np.array(vcf.gt2), 2 * n_ind, src)
    assert np.array_equal(pr2, p_ref) and np.array_equal(lb2, l_ref)
    none, lb3 = dev2.infer_gt2(vcf.gt2, 2 * n_ind, src, want_proba=False)
    assert none is None and np.array_equal(lb3, l_ref)
    bad = src.copy()
    bad[0] = vcf.n_variants
    with pytest.raises(_lib.GnxError):
        dev.infer_gt2(vcf.gt2, 2 * n_ind, bad)
    ctx.close()


@pytest.mark.parametrize("p2", ["1", "2"])   # "2": the rows stay 2-bit through base, Gnofix and the way back (GNX_LR_P2)
@pytest.mark.parametrize("C,M,A,n_ind,batch", [(16037, 100, 4, 9, 0), (9037, 60, 5, 14, 8)])
def test_phase_gt2_equals_the_int8_route(ga, tmp_path, monkeypatch, C, M, A, n_ind, batch, p2):
    """gnomix.py:60-72: B = base.predict_proba(X); X_phased, labels = model.phase(X, B); proba = model.predict_proba(X_phased)"""
    from gnomix_amd import synth, vcfio, _lib
    monkeypatch.setenv("GNX_LR_P2", p2)
    rng = np.random.default_rng(C)
    d = synth.synthetic_model(C=C, M=M, A=A, S=75, n_rounds=8, seed=11)
    p = _query(tmp_path, d, n_ind, rng)
    if batch:
        monkeypatch.setenv("GNX_HOST_BATCH", str(batch))
    ctx = _lib.Context(0)
    dev = ga.DeviceModel(d, ctx=ctx)
    vcf = vcfio.read_vcf(p, chm="22", ctx=ctx)
    X, vi, fi = vcfio.vcf_to_npy(vcf, d.snp_pos, d.snp_ref, return_idx=True, verbose=False)
    _, B = dev.base_predict(X)
    Xp, Y, nsw = dev.gnofix(X, B)
    p_ref, _ = dev.infer(Xp)
    src, _, _ = vcfio.column_map(vcf, d.snp_pos, d.snp_ref, verbose=False)
    Go, pr, lab, ns2 = dev.phase_gt2(vcf.gt2, 2 * n_ind, src, out_cols=fi)
    assert np.array_equal(lab, Y) and np.array_equal(ns2, nsw) and np.array_equal(pr, p_ref)
    back = np.stack([(Go[:, h // 4] >> (2 * (h % 4))) & 3 for h in range(2 * n_ind)], axis=0).astype(np.int8)
    assert np.array_equal(back, Xp[:, fi])
    assert nsw.sum() >= 0 and Go.shape == (len(fi), vcf.gt2.shape[1])
    # a smoother that cannot re-phase is refused like src/model.py:194
    dc = synth.synthetic_model(C=2049, M=64, A=3, S=9, n_rounds=2, seed=1, smooth="crf")
    devc = ga.DeviceModel(dc, ctx=ctx)
    with pytest.raises(_lib.GnxError):
        devc.phase_gt2(np.zeros((10, 4), np.uint8), 4, np.full(2049, -1, np.int32))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,W,A", [(2, 3, 2), (14, 41, 7), (500, 9, 3), (1030, 171, 6), (3, 1, 33)])
def test_write_fb_on_the_gpu_is_byte_identical_to_the_host_writer(tmp_path, N, W, A):
    """gnx_write_fb_dev (number text produced by k_fb_text.hip: Schubfach digits, numpy's layout rules) against gnx_write_fb, which
    tests/test_io_native.py holds to numpy's own text and to golden G6: every exponent, powers of two, subnormals, 0, -0, 1, the
    1e-4 and 1e16 switches between positional and scientific notation, inf, and NaN (an empty field)"""
    import gnomix_amd as ga
    from gnomix_amd import postprocess as pp, _lib
    rng = np.random.default_rng(N * 1000 + W)
    proba = rng.dirichlet(np.ones(A) * 0.3, size=(N, W)).astype(np.float32)
    flat = proba.reshape(-1)
    special = np.array([0.0, -0.0, 1.0, 1e-4, 9.9999e-5, 9.999999e-5, 1e16, 9.9999998e15, 1e-45, 3.4028235e38, np.inf, -np.inf, 0.1, 0.14285715,
                        1 / 3, 16777216, 1e-5, 123456.79, 0.001, 1e15, np.nan, -2.5e-7, 5e-39], np.float32)
    k = min(len(special), flat.size)
    flat[rng.choice(flat.size, k, replace=False)] = special[:k]
    if flat.size > 4000:   # raw bit patterns: every exponent; powers of two and their neighbours
        bits = rng.integers(0, 2 ** 32, 3000, dtype=np.uint64).astype(np.uint32)
        pw = ((np.arange(1, 255, dtype=np.uint32) << np.uint32(23))[:, None] + np.array([0, 1, 0xFFFFFFFF], np.uint32)[None, :]).astype(np.uint32).ravel()
        vals = np.concatenate([bits, pw]).view(np.float32)
        flat[rng.choice(flat.size, len(vals), replace=False)] = vals
    meta = {"chm": ["22"] * W, "spos": np.arange(W) * 1000 + 16_000_000, "epos": np.arange(W) * 1000 + 16_000_999, "sgpos": np.arange(W) * 0.21,
            "egpos": np.arange(W) * 0.21 + 0.2}
    samples, pops = ["I%d" % i for i in range((N + 1) // 2)], ["P%d" % a for a in range(A)]
    if N % 2:
        proba = proba[:N - 1]
        samples = samples[:(N - 1) // 2]
    ctx = _lib.default_context(0)
    pp.write_fb(str(tmp_path / "host"), meta, proba, pops, samples)
    pp.write_fb(str(tmp_path / "dev"), meta, proba, pops, samples, ctx=ctx)
    a, b = open(str(tmp_path / "host.fb"), "rb").read(), open(str(tmp_path / "dev.fb"), "rb").read()
    assert len(a) == len(b) and a == b


@pytest.mark.gpu
@pytest.mark.parametrize("phase", [False, True])
@pytest.mark.parametrize("n_ind,k", [(11, 3), (5, 2), (3, 3)])
def test_one_process_several_contexts_writes_the_same_files(ga, tmp_path, phase, n_ind, k):
    """SURVEY 8e on the file path (gnomix_amd/multi.py): the individuals of ONE parsed query cut over k contexts (here k contexts on
    the one GPU of the box; a node gives each its own device), ragged blocks, every context uploading only its columns of the 2-bit
    rows and writing its row block of the shared outputs — .msp / .fb (and the phased VCF) byte-identical to the one-context run"""
    from gnomix_amd import synth, vcfio, cli, multi
    rng = np.random.default_rng(100 * n_ind + k)
    d = synth.synthetic_model(C=7037, M=100, A=5, S=21, n_rounds=5, seed=7)
    q = _query(tmp_path, d, n_ind, rng)
    d.gen_map_pos = np.array([1, int(d.snp_pos[len(d.snp_pos) // 3]), int(d.snp_pos[-1]) + 1000])
    d.gen_map_cm = np.array([0.0, 1.5, 9.25])
    d.population_order = np.array(["P%d" % a for a in range(d.A)])
    blocks = multi.shard_individuals(n_ind, k)
    assert sum(n for _, n in blocks) == n_ind and all(i0 % 2 == 0 for i0, _ in blocks) and all(n > 0 for _, n in blocks)
    if n_ind >= 2 * k:
        assert len(blocks) == k and len({n for _, n in blocks}) > 1 or n_ind % k == 0      # ragged
    g = ga.HipGnomix(d, device=0)
    outs = []
    for name, devices in (("one", None), ("many", [0] * k)):
        od = tmp_path / name
        od.mkdir()
        cli.run_inference({"query_file": q, "chm": "22", "output_basename": str(od), "phase": phase}, g, devices=devices)
        outs.append(od)
    files = ["query_results.msp", "query_results.fb"] + (["query_file_phased.vcf"] if phase else [])
    for f in files:
        a, b = (outs[0] / f).read_bytes(), (outs[1] / f).read_bytes()
        assert len(a) > 100 and a == b, f
    # the group API directly, labels only through float64 outputs as well
    vcf = vcfio.read_vcf(q, chm="22", ctx=g.dev.ctx)
    src, _, fi = vcfio.column_map(vcf, d.snp_pos, d.snp_ref, verbose=False)
    grp = multi.DeviceGroup(d, [0] * k, first=g.dev)
    p1, l1 = g.dev.infer_gt2(vcf.gt2, 2 * n_ind, src, proba_dtype=np.float64)
    p2, l2 = grp.infer_gt2(vcf.gt2, 2 * n_ind, src, proba_dtype=np.float64)
    assert np.array_equal(p1, p2) and np.array_equal(l1, l2)
    if phase:
        G1, q1, y1, s1 = g.dev.phase_gt2(vcf.gt2, 2 * n_ind, src, out_cols=fi)
        G2, q2, y2, s2 = grp.phase_gt2(vcf.gt2, 2 * n_ind, src, out_cols=fi)
        assert np.array_equal(G1, G2) and np.array_equal(q1, q2) and np.array_equal(y1, y2) and np.array_equal(s1, s2)
    for m in grp.models[1:]:
        m.close()


# ---------------------------------------------------------------- multi-GPU first-contact hardening ------------------
def test_page_locked_buffers_are_portable(ga):
    """every buffer multi.py shares between contexts (the parsed gt2 rows, the outputs) comes from gnx_host_alloc: page-locked for
    EVERY device of the process (hipHostMallocPortable), not only for the device that was current when it was made"""
    import ctypes
    from gnomix_amd import _lib
    ctx = _lib.default_context(0)
    arr = ctx.pinned_empty((1 << 20,), np.uint8)
    fl = ctypes.c_uint(0)
    assert ctx.lib.gnx_host_flags(arr.ctypes.data, ctypes.byref(fl)) == _lib.GNX_OK
    assert fl.value & 1, hex(fl.value)      # GNX_HOST_PORTABLE
    assert ctx.lib.gnx_host_flags(None, ctypes.byref(fl)) == _lib.GNX_EINVAL


def test_gnx_devices_is_validated(ga, monkeypatch):
    from gnomix_amd import multi, _lib
    n = _lib.load().gnx_device_count()
    monkeypatch.setenv("GNX_DEVICES", "0,%d" % n)
    with pytest.raises(_lib.GnxError, match="outside"):
        multi.visible_devices()
    monkeypatch.setenv("GNX_DEVICES", ",")
    with pytest.raises(_lib.GnxError, match="names no device"):
        multi.visible_devices()
    monkeypatch.setenv("GNX_DEVICES", "zero")
    with pytest.raises(_lib.GnxError):
        multi.visible_devices()
    monkeypatch.setenv("GNX_DEVICES", "0, 0")
    assert multi.visible_devices() == [0, 0]
    monkeypatch.delenv("GNX_DEVICES")
    assert multi.visible_devices() == list(range(n))
    from gnomix_amd import synth
    d = synth.synthetic_model(C=1037, M=100, A=3, S=5, n_rounds=2, seed=1)
    with pytest.raises(_lib.GnxError, match="outside"):
        multi.DeviceGroup(d, [0, n + 3])


def test_device_group_reports_every_failed_device_and_closes(ga, tmp_path):
    """k = 3 contexts, two of them broken (their model handles closed under the group): ONE error names both, the group's own
    contexts are released, the caller's first model keeps working"""
    from gnomix_amd import synth, vcfio, multi, _lib
    rng = np.random.default_rng(5)
    d = synth.synthetic_model(C=4037, M=100, A=4, S=11, n_rounds=4, seed=2)
    n_ind = 12
    q = _query(tmp_path, d, n_ind, rng)
    first = ga.DeviceModel(d, ctx=_lib.Context(0))
    vcf = vcfio.read_vcf(q, chm="22", ctx=first.ctx)
    src, _, _ = vcfio.column_map(vcf, d.snp_pos, d.snp_ref, verbose=False)
    grp = multi.DeviceGroup(d, [0, 0, 0], first=first)
    assert len(grp.models) == 3 and len(grp._own) == 2
    p_ok, l_ok = grp.infer_gt2(vcf.gt2, 2 * n_ind, src)
    grp.models[1].close()
    grp.models[2].close()
    with pytest.raises(multi.DeviceGroupError) as ei:
        grp.infer_gt2(vcf.gt2, 2 * n_ind, src)
    assert [f[0] for f in ei.value.failures] == [1, 2] and "context 1" in str(ei.value) and "context 2" in str(ei.value)
    assert grp._own == []
    p1, l1 = first.infer_gt2(vcf.gt2, 2 * n_ind, src)       # the caller's model and context are untouched
    assert np.array_equal(l1, l_ok) and np.array_equal(p1, p_ok)
    # a group never makes more replicas than there are shards of whole individuals
    g2 = multi.DeviceGroup(d, [0] * 8, first=first, n_ind=3)
    assert len(g2.models) <= 3
    g2.close()


def test_device_group_follows_the_first_models_calibrate_switch(ga, tmp_path):
    """gnomix.py:365-370 pokes model.calibrate after loading; the replicas must calibrate (or not) like the first model, or the
    shards of one output array would differ"""
    from gnomix_amd import synth, vcfio, multi, _lib
    rng = np.random.default_rng(6)
    d = synth.synthetic_model(C=4037, M=100, A=4, S=11, n_rounds=4, seed=3)
    xs = [np.sort(rng.random(9)) for _ in range(d.A)]
    d.calib_off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.int32)
    d.calib_x = np.concatenate(xs)
    d.calib_y = np.concatenate([np.sort(rng.random(len(x))) for x in xs])
    n_ind = 10
    q = _query(tmp_path, d, n_ind, rng)
    first = ga.DeviceModel(d, ctx=_lib.Context(0))
    vcf = vcfio.read_vcf(q, chm="22", ctx=first.ctx)
    src, _, _ = vcfio.column_map(vcf, d.snp_pos, d.snp_ref, verbose=False)
    with multi.DeviceGroup(d, [0, 0, 0], first=first) as grp:
        for on in (True, False, True):
            first.set_calibrate(on)
            p1, l1 = first.infer_gt2(vcf.gt2, 2 * n_ind, src)
            p2, l2 = grp.infer_gt2(vcf.gt2, 2 * n_ind, src)
            assert p1.dtype == p2.dtype and np.array_equal(p1, p2) and np.array_equal(l1, l2), on
        raw = first.infer_gt2(vcf.gt2, 2 * n_ind, src)[0]
        first.set_calibrate(False)
        assert not np.array_equal(raw.astype(np.float64), first.infer_gt2(vcf.gt2, 2 * n_ind, src)[0].astype(np.float64))
