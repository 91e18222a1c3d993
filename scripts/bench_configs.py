#!/usr/bin/env python3
"""Measurements of BASELINE.json's other configs on ONE GPU (not the driver's bench line; numbers go to DESIGN.md §5).

  python scripts/bench_configs.py [c3] [c4] [c5a] [c5b]

All inputs synthetic and resident in HBM; per-GPU shard sizes of the multi-GPU configs (N/8)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import gnomix_amd
from gnomix_amd import synth, _lib


def timed(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def prof(ctx):
    out = {}
    for k, name in _lib.KERNEL_NAMES.items():
        ms, n = ctx.profile_get(k)
        if n:
            out[name] = round(ms / n, 4)
    return out


def c4(N=12500):
    """whole genome, 22 chromosomes, A=7, LR + xgb; N = 100k/8 haplotypes per GPU, chromosome-major batches"""
    tot_t, tot_t2, tot_w, rows = 0.0, 0.0, 0, []
    for k, Wk in enumerate(synth.GENOME_W):
        C = 1000 * Wk + 500
        data = synth.synthetic_model(C=C, M=1000, A=7, S=75, seed=k)
        model = gnomix_amd.DeviceModel(data)
        del data
        X = synth.synthetic_X_device(N, C, "cuda:0", seed=k)
        model.ctx.profile_reset(); model.ctx.profile_enable(True)
        dt = timed(lambda: model.infer_device(X), reps=2)
        model.ctx.profile_enable(False)
        rows.append((k + 1, Wk, round(dt * 1e3, 2), prof(model.ctx)))
        tot_t += dt; tot_w += Wk
        # the same batch resident as 2-bit rows (k_base_logistic_p2)
        Pk = model.pack_device(X)
        ref = model.infer_device(X)
        got = model.infer_packed_device(Pk)
        assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
        dt2 = timed(lambda: model.infer_packed_device(Pk), reps=2)
        tot_t2 += dt2
        del Pk, ref, got
        model.close(); del X, model
        torch.cuda.empty_cache()
        print("chr%d W=%d %.2f ms" % (k + 1, Wk, dt * 1e3), flush=True)
    res = {"config": "c4 whole genome 22 chr, A=7, LR+xgb", "haplotypes_per_gpu": N, "sum_W": tot_w,
           "seconds_per_batch": tot_t, "haplotypes_per_s_per_gpu": N / tot_t, "hap_windows_per_s": N * tot_w / tot_t,
           "projected_8gpu_haplotypes_per_s": 8 * N / tot_t,
           "resident_2bit": {"seconds_per_batch": tot_t2, "haplotypes_per_s_per_gpu": N / tot_t2, "outputs_identical_to_int8": True,
                             "resident_GB_per_gpu": N * sum(1000 * w + 500 for w in synth.GENOME_W) / 4 / 1e9}}
    print(json.dumps(res))
    return res


def c3(N=4096):
    """chr1 array density: C=250400, M=175, A=7, CovRSK/SVC base (1400 training haplotypes per window, all SVs) + xgb"""
    C, M, A = 250_400, 175, 7
    t0 = time.time()
    data = synth.synthetic_svc_model(C, M, A, n_fit_per_class=200, sv_frac=1.1, seed=0, S=75, smooth="xgb")
    print("model synthesised in %.0f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    model = gnomix_amd.DeviceModel(data)
    print("model loaded in %.0f s" % (time.time() - t0), flush=True)
    X = synth.synthetic_X_device(N, C, "cuda:0", seed=1)
    model.ctx.profile_reset(); model.ctx.profile_enable(True)
    dt = timed(lambda: model.infer_device(X), reps=1, warm=1)
    model.ctx.profile_enable(False)
    W = data.W
    nsv = 1400
    res = {"config": "c3 chr1 array, CovRSK base + xgb", "haplotypes": N, "W": W, "seconds": dt, "haplotypes_per_s": N / dt,
           "symbol_compares_per_s": N * W * nsv * (M + 2 * (M // 2)) / dt, "kernels_ms": prof(model.ctx)}
    print(json.dumps(res))
    return res


def c5a(N=25000):
    """chr1 WGS density, A=12, LR + CRF (the reference rejects CRF + Gnofix: src/model.py:194)"""
    C, M, A = 1_431_500, 1000, 12
    data = synth.synthetic_model(C=C, M=M, A=A, S=75, seed=5, smooth="crf")
    model = gnomix_amd.DeviceModel(data)
    X = synth.synthetic_X_device(N, C, "cuda:0", seed=2)
    model.ctx.profile_reset(); model.ctx.profile_enable(True)
    dt = timed(lambda: model.infer_device(X), reps=2)
    model.ctx.profile_enable(False)
    res = {"config": "c5a chr1 WGS A=12 LR+CRF", "haplotypes_per_gpu": N, "W": data.W, "seconds": dt,
           "haplotypes_per_s_per_gpu": N / dt, "alg_GBps_base": (C + data.W * A * 8) * N / dt / 1e9, "kernels_ms": prof(model.ctx)}
    # the same batch resident as 2-bit rows (k_base_logistic_p2: one column tile per slot, two passes)
    Pk = model.pack_device(X)
    ref = model.base_predict_device(X[:4096], f64=True)
    assert torch.equal(ref, model.base_predict_packed_device(Pk[:4096], f64=True))
    del ref
    model.ctx.profile_reset(); model.ctx.profile_enable(True)
    dt2 = timed(lambda: model.infer_packed_device(Pk, want_proba=True), reps=2)
    model.ctx.profile_enable(False)
    res["resident_2bit"] = {"seconds": dt2, "haplotypes_per_s_per_gpu": N / dt2, "kernels_ms": prof(model.ctx), "base_identical_to_int8": True}
    print(json.dumps(res))
    return res


def c5b(n_ind=4096):
    """chr1 WGS density (W=1431), A=12, xgb smoother + Gnofix re-phasing loop on individuals with 2 switch errors per
    haplotype pair; a smoother that behaves like a trained one (signal trees + 1200-tree cost profile).  Host-pointer ABI:
    the time includes staging X and B over PCIe."""
    W, A, S = 1431, 12, 75
    C = 1000 * W + 500
    data = gnomix_amd.GnxModelData(C=C, M=1000, A=A, S=S, context=500, smooth_kind="xgb")
    for k, v in synth.synthetic_smoothing_trees(100, A, S, seed=6).items():
        setattr(data, k, v)
    model = gnomix_amd.DeviceModel(data)
    B = synth.synthetic_phased_individuals(n_ind, W, A, seed=3)
    X = np.random.RandomState(1).randint(0, 2, size=(2 * n_ind, C)).astype(np.int8)
    model.gnofix(X[:8], B[:8])  # warm-up
    # host-pointer ABI, as a caller that keeps its arrays page-locked would use it (in place: no host copy of the 11.7 GB of X)
    Xh = model.ctx.pinned_empty(X.shape, np.int8); Xh[:] = X
    Bh = model.ctx.pinned_empty(B.shape, np.float64); Bh[:] = B
    Yh = model.ctx.pinned_empty((2 * n_ind, W), np.int32); nh = model.ctx.pinned_empty((n_ind,), np.int32)
    model.gnofix(Xh[:1024].copy(), Bh[:1024])  # sizes the workspaces / streams
    model.ctx.profile_reset(); model.ctx.profile_enable(True)
    t0 = time.perf_counter()
    Xo, Y, nsw = model.gnofix(Xh, Bh, inplace=True, out=(Yh, nh))
    dt = time.perf_counter() - t0
    model.ctx.profile_enable(False)
    Xp = X.copy()
    t0 = time.perf_counter()
    Xo2, Y2, nsw2 = model.gnofix(Xp, B, inplace=True)   # the same with ordinary (pageable) numpy arrays
    dt_pageable = time.perf_counter() - t0
    assert np.array_equal(Y2, Y) and np.array_equal(Xo2, Xo)
    # device-resident: the same individuals already in HBM
    nd = min(n_ind, 2048)
    Xd = torch.from_numpy(X[:2 * nd]).cuda()
    Bd = torch.from_numpy(B[:2 * nd]).cuda()
    model.gnofix_device(Xd.clone(), Bd)   # sizes the workspaces
    Xw = Xd.clone()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    Yd, nsd = model.gnofix_device(Xw, Bd)
    torch.cuda.synchronize()
    ddt = time.perf_counter() - t1
    assert np.array_equal(Yd.cpu().numpy(), Y[:2 * nd])
    res = {"config": "c5b chr1 WGS A=12 xgb smoother + Gnofix", "individuals": n_ind, "seconds_incl_staging": dt,
           "individuals_per_s": n_ind / dt, "individuals_per_s_pageable": n_ind / dt_pageable, "device_resident_individuals_per_s": nd / ddt,
           "mean_switches": float(nsw.mean()), "max_switches": int(nsw.max()), "kernels_ms": prof(model.ctx)}
    print(json.dumps(res))
    return res


def c5br(n_ind=2048):
    """config 5b with everything resident in HBM (no staging): chr1 WGS (W = 1431), A = 12, xgb smoother + Gnofix on individuals
    with two switch errors per haplotype pair — the workload README / DESIGN quote for the Gnofix kernel's counters
    (profiles/r05_c5br_*), once on int8 rows and once on 2-bit rows (gnx_gnofix_packed_dev)"""
    W, A, S = 1431, 12, 75
    C = 1000 * W + 500
    data = gnomix_amd.GnxModelData(C=C, M=1000, A=A, S=S, context=500, smooth_kind="xgb")
    for k, v in synth.synthetic_smoothing_trees(100, A, S, seed=6).items():
        setattr(data, k, v)
    model = gnomix_amd.DeviceModel(data)
    B = synth.synthetic_phased_individuals(n_ind, W, A, seed=3)
    Xd = torch.randint(0, 2, (2 * n_ind, C), dtype=torch.int8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    Bd = torch.from_numpy(B).cuda()
    Pd = model.pack_device(Xd)
    Y0, n0 = model.gnofix_device(Xd.clone(), Bd)          # sizes the workspaces
    out = {}
    for name, src, fn in (("int8", Xd, model.gnofix_device), ("packed", Pd, model.gnofix_packed_device)):
        best = 1e9
        for _ in range(3):
            w = src.clone()
            torch.cuda.synchronize()
            model.ctx.profile_reset(); model.ctx.profile_enable(True)
            t0 = time.perf_counter()
            Y, ns = fn(w, Bd)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
            model.ctx.profile_enable(False)
        assert torch.equal(Y, Y0) and torch.equal(ns, n0)
        out[name] = {"seconds": best, "individuals_per_s": n_ind / best, "kernels_ms": prof(model.ctx)}
    res = {"config": "c5br chr1 WGS A=12 xgb smoother + Gnofix, device-resident", "individuals": n_ind, "mean_switches": float(n0.float().mean()),
           "int8_rows": out["int8"], "packed_rows": out["packed"]}
    print(json.dumps(res))
    return res


def cf(N=10000):
    """chr22 (config 2 geometry) with the tree-ensemble base: XGBBase shape, 20 rounds x 7 classes, depth 4, per window"""
    C, M, A = 370_500, 1000, 7
    t0 = time.time()
    data = synth.synthetic_forest_model(C, M, A, n_rounds=20, depth=4, seed=0, S=75, smooth="xgb")
    print("model synthesised in %.0f s" % (time.time() - t0), flush=True)
    model = gnomix_amd.DeviceModel(data)
    X = synth.synthetic_X_device(N, C, "cuda:0", seed=1)
    model.ctx.profile_reset(); model.ctx.profile_enable(True)
    dt = timed(lambda: model.infer_device(X), reps=3, warm=1)
    model.ctx.profile_enable(False)
    k = prof(model.ctx)
    res = {"config": "cf chr22, forest base (20 rounds x 7, depth 4) + xgb", "haplotypes": N, "W": data.W, "seconds": dt,
           "haplotypes_per_s": N / dt, "kernels_ms": k,
           "base_node_steps_per_s": N * data.W * 140 * 4 / (k.get("k_base_forest", float("nan")) * 1e-3)}
    print(json.dumps(res))
    return res


def crf_(N=10000):
    """chr22 geometry with the random-forest base (RFBase shape: 20 trees of depth 4 per window) + xgb smoother"""
    C, M, A = 370_500, 1000, 7
    data = synth.synthetic_rforest_model(C, M, A, n_trees=20, depth=4, seed=0, S=75, smooth="xgb")
    model = gnomix_amd.DeviceModel(data)
    X = synth.synthetic_X_device(N, C, "cuda:0", seed=1)
    model.ctx.profile_reset(); model.ctx.profile_enable(True)
    dt = timed(lambda: model.infer_device(X), reps=3, warm=1)
    model.ctx.profile_enable(False)
    res = {"config": "rf chr22, random-forest base (20 trees, depth 4) + xgb", "haplotypes": N, "W": data.W, "seconds": dt,
           "haplotypes_per_s": N / dt, "kernels_ms": prof(model.ctx)}
    print(json.dumps(res))
    return res


def modes(N=10000):
    """chr22 geometry (config 2) under the reference's other modes: fast = LR + CRF, large = LR + CNN (model.py:50-72)"""
    out = {}
    for name, smooth in (("default (LR + xgb)", "xgb"), ("fast (LR + CRF)", "crf"), ("large (LR + CNN)", "cnn")):
        data = synth.synthetic_model(seed=0, n_rounds=100, smooth=smooth, **synth.CHR22)
        model = gnomix_amd.DeviceModel(data)
        X = synth.synthetic_X_device(N, data.C, "cuda:0", seed=1)
        model.ctx.profile_reset(); model.ctx.profile_enable(True)
        dt = timed(lambda: model.infer_device(X), reps=5, warm=2)
        model.ctx.profile_enable(False)
        out[name] = {"haplotypes_per_s": N / dt, "ms": dt * 1e3, "kernels_ms": prof(model.ctx)}
        model.close(); del X, model
        torch.cuda.empty_cache()
    res = {"config": "chr22, 10k haplotypes, the reference's modes", "modes": out}
    print(json.dumps(res))
    return res


if __name__ == "__main__":
    which = sys.argv[1:] or ["c4", "c5a", "c5b", "c3"]
    out = {}
    for w in which:
        out[w] = globals()["crf_" if w == "rf" else w]()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_configs.json"), "w"), indent=1)
