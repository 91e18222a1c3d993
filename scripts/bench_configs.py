#!/usr/bin/env python3
"""BASELINE.json's configs 3, 4, 5 on ONE GPU: one GPU's shard of the multi-GPU configs (N / 8), all inputs synthetic and resident
in HBM.  Two users:

  python scripts/bench_configs.py [c3] [c4] [c5a] [c5b] [c5br] ...   the full workloads with their JSON lines (profiling: collect_profiles.sh)
  bench.py -> bench_legs()                                            the same workloads as the `configs` object of the driver-run line:
                                                                      c3, c4_2bit, c5a_2bit, c5b_resident, 3 timed repetitions each

Ceilings the legs name (MI355X_MICROARCH.md): HBM 8 TB/s; int8 MFMA 16x16x64 3944 TOPS (measured ceiling); VALU 256 CU x 4 SIMD x 16
lanes x 2.4 GHz = 3.93e13 lane-operations/s."""
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


HBM_PEAK = 8.0e12
I8_MFMA_PEAK = 3944e12
VALU_LANE_OPS = 256 * 4 * 16 * 2.4e9


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


def c4(N=12500, reps=2, reps_int8=None, verbose=True, ctx=None):
    """whole genome, 22 chromosomes, A=7, LR + xgb; N = 100k/8 haplotypes per GPU, chromosome-major batches.  Every chromosome's
    batch is run int8-resident (reps_int8 timed passes, default = reps; 1 = a single untimed-quality pass, just the reference outputs)
    and 2-bit-resident (reps timed passes), outputs compared bit for bit."""
    reps_int8 = reps if reps_int8 is None else reps_int8
    tot_t, tot_t2, tot_w, rows, same = 0.0, 0.0, 0, [], True
    t_synth = t_load = 0.0
    kms = {}
    for k, Wk in enumerate(synth.GENOME_W):
        C = 1000 * Wk + 500
        t0 = time.perf_counter()
        data = synth.synthetic_model(C=C, M=1000, A=7, S=75, seed=k)
        t1 = time.perf_counter()
        model = gnomix_amd.DeviceModel(data, ctx=ctx) if ctx is not None else gnomix_amd.DeviceModel(data)
        t_synth += t1 - t0; t_load += time.perf_counter() - t1
        del data
        X = synth.synthetic_X_device(N, C, "cuda:0", seed=k)
        model.ctx.profile_reset(); model.ctx.profile_enable(True)
        dt = timed(lambda: model.infer_device(X), reps=reps_int8, warm=1 if reps_int8 > 1 else 0)
        model.ctx.profile_enable(False)
        rows.append((k + 1, Wk, round(dt * 1e3, 2), prof(model.ctx)))
        tot_t += dt; tot_w += Wk
        # the same batch resident as 2-bit rows (k_base_logistic_p2)
        Pk = model.pack_device(X)
        ref = model.infer_device(X)
        del X
        got = model.infer_packed_device(Pk)
        same = same and bool(torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]))
        model.ctx.profile_reset(); model.ctx.profile_enable(True)
        dt2 = timed(lambda: model.infer_packed_device(Pk), reps=reps, warm=0)
        model.ctx.profile_enable(False)
        for name, ms in prof(model.ctx).items():
            kms[name] = kms.get(name, 0.0) + ms
        tot_t2 += dt2
        del Pk, ref, got
        model.close(); del model
        torch.cuda.empty_cache()
        if verbose:
            print("chr%d W=%d %.2f ms" % (k + 1, Wk, dt * 1e3), flush=True)
    sumC = sum(1000 * w + 500 for w in synth.GENOME_W)
    res = {"config": "c4 whole genome 22 chr, A=7, LR+xgb", "haplotypes_per_gpu": N, "sum_W": tot_w,
           "seconds_per_batch": tot_t, "haplotypes_per_s_per_gpu": N / tot_t, "hap_windows_per_s": N * tot_w / tot_t,
           "projected_8gpu_haplotypes_per_s": 8 * N / tot_t, "int8_reps": reps_int8,
           "model_synthesis_s": round(t_synth, 2), "model_load_s": round(t_load, 2),
           "resident_2bit": {"seconds_per_batch": tot_t2, "haplotypes_per_s_per_gpu": N / tot_t2, "outputs_identical_to_int8": same,
                             "kernels_ms_sum_over_chromosomes": {k: round(v, 3) for k, v in kms.items()},
                             "alg_bytes_per_haplotype": sumC // 4 + 3 * tot_w * 7 * 4 + tot_w,
                             "resident_GB_per_gpu": N * sumC / 4 / 1e9}}
    if verbose:
        print(json.dumps(res))
    return res


def c3(N=4096, reps=1, verbose=True, ctx=None):
    """chr1 array density: C=250400, M=175, A=7, CovRSK/SVC base (1400 training haplotypes per window, all SVs) + xgb"""
    C, M, A = 250_400, 175, 7
    t0 = time.time()
    data = synth.synthetic_svc_model(C, M, A, n_fit_per_class=200, sv_frac=1.1, seed=0, S=75, smooth="xgb")
    t_synth = time.time() - t0
    t0 = time.time()
    model = gnomix_amd.DeviceModel(data, ctx=ctx) if ctx is not None else gnomix_amd.DeviceModel(data)
    t_load = time.time() - t0
    if verbose:
        print("model synthesised in %.0f s, loaded in %.0f s" % (t_synth, t_load), flush=True)
    X = synth.synthetic_X_device(N, C, "cuda:0", seed=1)
    model.ctx.profile_reset(); model.ctx.profile_enable(True)
    dt = timed(lambda: model.infer_device(X), reps=reps, warm=1)
    model.ctx.profile_enable(False)
    W = data.W
    nsv = 1400
    width = M + 2 * (M // 2)
    cmp_s = N * W * nsv * width / dt
    res = {"config": "c3 chr1 array, CovRSK base + xgb", "haplotypes": N, "W": W, "seconds": dt, "haplotypes_per_s": N / dt,
           "symbol_compares_per_s": cmp_s, "kernels_ms": prof(model.ctx),
           # symbol equality on two bit planes is 3 VALU operations per 32 SNPs of one (query, support vector) pair: the share of the
           # chip's VALU lane-operations that the comparisons ALONE need (run peeling, the g() look-ups and libsvm's sums come on top)
           "valu_compare_floor_frac": cmp_s / 32 * 3 / VALU_LANE_OPS,
           "model_synthesis_s": round(t_synth, 2), "model_load_s": round(t_load, 2)}
    model.close()
    if verbose:
        print(json.dumps(res))
    return res


def c5a(N=25000, reps=2, reps_int8=None, verbose=True, ctx=None, only=None):
    """chr1 WGS density, A=12, LR + CRF (the reference rejects CRF + Gnofix: src/model.py:194).  only="int8" / "p2": ONE launch
    size and ONE logistic kernel in the process, so that a rocprofv3 kernel-stats average of the run means something."""
    reps_int8 = reps if reps_int8 is None else reps_int8
    C, M, A = 1_431_500, 1000, 12
    t0 = time.perf_counter()
    data = synth.synthetic_model(C=C, M=M, A=A, S=75, seed=5, smooth="crf")
    t1 = time.perf_counter()
    model = gnomix_amd.DeviceModel(data, ctx=ctx) if ctx is not None else gnomix_amd.DeviceModel(data)
    t_load = time.perf_counter() - t1
    X = synth.synthetic_X_device(N, C, "cuda:0", seed=2)
    W = data.W
    res = {"config": "c5a chr1 WGS A=12 LR+CRF", "haplotypes_per_gpu": N, "W": W,
           "model_synthesis_s": round(t1 - t0, 2), "model_load_s": round(t_load, 2)}
    if only != "p2":
        model.ctx.profile_reset(); model.ctx.profile_enable(True)
        dt = timed(lambda: model.infer_device(X), reps=reps_int8, warm=1 if reps_int8 > 1 else 0)
        model.ctx.profile_enable(False)
        res.update({"seconds": dt, "haplotypes_per_s_per_gpu": N / dt, "alg_GBps_base": (C + W * A * 8) * N / dt / 1e9, "kernels_ms": prof(model.ctx)})
    if only != "int8":
        # the same batch resident as 2-bit rows (k_base_logistic_p2)
        Pk = model.pack_device(X)
        same = None
        if only is None:
            ref = model.infer_device(X)
            got = model.infer_packed_device(Pk, want_proba=True)
            same = bool(torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]))
            del ref, got
        del X
        model.ctx.profile_reset(); model.ctx.profile_enable(True)
        dt2 = timed(lambda: model.infer_packed_device(Pk, want_proba=True), reps=reps)
        model.ctx.profile_enable(False)
        k2 = prof(model.ctx)
        base_s = k2.get("k_base_logistic", float("nan")) * 1e-3
        alg_bytes = (C + 3) // 4 + W * A * 8                      # 2-bit X once + B float64 (the CRF's input) once
        # R = 2 windows per SNP at context = M / 2: 24 class columns x 7 limbs = 168 flat columns = 11 MFMA tiles per 64 SNPs and 16 rows
        # (k_base_logistic_p2f; one 16-column tile per slot would be 2 x 7 = 14)
        i8_ops = (C / 64.0) * 11 * (2 * 16 * 16 * 64) / 16.0       # per haplotype
        res["resident_2bit"] = {"seconds": dt2, "haplotypes_per_s_per_gpu": N / dt2, "kernels_ms": k2, "outputs_identical_to_int8": same,
                                "base_alg_bytes": alg_bytes * N, "base_hbm_frac": alg_bytes * N / base_s / HBM_PEAK,
                                "base_i8_ops": i8_ops * N, "base_i8_mfma_frac": i8_ops * N / base_s / I8_MFMA_PEAK}
    model.close()
    if verbose:
        print(json.dumps(res))
    return res


def c5a_int8():
    return c5a(only="int8")


def c5a_p2():
    return c5a(only="p2")


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


def c5br(n_ind=2048, verbose=True, ctx=None):
    """config 5b with everything resident in HBM (no staging): chr1 WGS (W = 1431), A = 12, xgb smoother + Gnofix on individuals
    with two switch errors per haplotype pair — the workload README / DESIGN quote for the Gnofix kernel's counters
    (profiles/r05_c5br_*), once on int8 rows and once on 2-bit rows (gnx_gnofix_packed_dev)"""
    W, A, S = 1431, 12, 75
    C = 1000 * W + 500
    data = gnomix_amd.GnxModelData(C=C, M=1000, A=A, S=S, context=500, smooth_kind="xgb")
    for k, v in synth.synthetic_smoothing_trees(100, A, S, seed=6).items():
        setattr(data, k, v)
    model = gnomix_amd.DeviceModel(data, ctx=ctx) if ctx is not None else gnomix_amd.DeviceModel(data)
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
        out[name] = {"seconds": best, "individuals_per_s": n_ind / best, "kernels_ms": prof(model.ctx),
                     "outputs_identical_to_int8": bool(torch.equal(Y, Y0) and torch.equal(ns, n0))}
    res = {"config": "c5br chr1 WGS A=12 xgb smoother + Gnofix, device-resident", "individuals": n_ind, "mean_switches": float(n0.float().mean()),
           "int8_rows": out["int8"], "packed_rows": out["packed"]}
    model.close()
    if verbose:
        print(json.dumps(res))
    return res


def _r(x, nd=4):
    """numbers of a leg rounded to `nd` significant digits (the driver keeps ~8 KB of bench.py's line)"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def bench_legs(ctx=None, which=("c3", "c4_2bit", "c5a_2bit", "c5b_resident"), reps=3, log=None):
    """bench.py's `configs` object: BASELINE.json's configs 3, 4, 5 at their documented one-GPU sizes, `reps` timed
    repetitions each, inputs resident in HBM, never `value`.  Every leg: throughput, per-kernel hipEvent averages, the algorithmic
    bytes or operations of its dominant kernel with the fraction of the named ceiling, outputs_identical_to_int8 where a 2-bit route
    ran, wall seconds of the whole leg (model synthesis and load included)."""
    legs = {}
    for name in which:
        t0 = time.perf_counter()
        try:
            if name == "c3":
                r = c3(reps=reps, verbose=False, ctx=ctx)
                leg = {"workload": "configs[2] one-GPU shard: chr1 array C=250400 M=175 W=1430 A=7, CovRSK/SVC base (1400 SVs per window) + xgb, 4096 haplotypes",
                       "haplotypes_per_s": r["haplotypes_per_s"], "kernels_ms": r["kernels_ms"], "alg_symbol_compares": r["symbol_compares_per_s"] * r["seconds"],
                       "ceiling": "VALU lane-ops (3 per 32 compares)", "frac": r["valu_compare_floor_frac"]}
            elif name == "c4_2bit":
                r = c4(reps=reps, reps_int8=1, verbose=False, ctx=ctx)
                q = r["resident_2bit"]
                leg = {"workload": "configs[3] one-GPU shard: 22 chromosome models (sum W = 17727, sum C = 17.7 M), A=7, LR + xgb, 12500 haplotypes as 2-bit rows",
                       "haplotypes_per_s": q["haplotypes_per_s_per_gpu"], "kernels_ms": q["kernels_ms_sum_over_chromosomes"],
                       "alg_bytes": q["alg_bytes_per_haplotype"] * r["haplotypes_per_gpu"], "ceiling": "HBM 8 TB/s (whole pipeline, SURVEY 8d)",
                       "frac": q["alg_bytes_per_haplotype"] * q["haplotypes_per_s_per_gpu"] / HBM_PEAK,
                       "outputs_identical_to_int8": q["outputs_identical_to_int8"], "int8_haplotypes_per_s_one_pass": r["haplotypes_per_s_per_gpu"],
                       "model_synthesis_s": r["model_synthesis_s"], "model_load_s": r["model_load_s"]}
            elif name == "c5a_2bit":
                r = c5a(reps=reps, reps_int8=1, verbose=False, ctx=ctx)
                q = r["resident_2bit"]
                leg = {"workload": "configs[4] inference half, one-GPU shard: chr1 WGS C=1431500 W=1431 A=12, LR + CRF, 25000 haplotypes as 2-bit rows",
                       "haplotypes_per_s": q["haplotypes_per_s_per_gpu"], "kernels_ms": q["kernels_ms"], "alg_i8_ops": q["base_i8_ops"],
                       "alg_bytes": q["base_alg_bytes"], "ceiling": "int8 MFMA 3944 TOPS (base pass)", "frac": q["base_i8_mfma_frac"],
                       "base_hbm_frac": q["base_hbm_frac"], "outputs_identical_to_int8": q["outputs_identical_to_int8"],
                       "int8_haplotypes_per_s_one_pass": r["haplotypes_per_s_per_gpu"]}
            elif name == "c5b_resident":
                r = c5br(verbose=False, ctx=ctx)
                q = r["packed_rows"]
                # the initial smoother pass is the leg's dominant kernel: W x 100 rounds x A trees x depth 4 node-steps per haplotype
                steps = 2 * r["individuals"] * 1431 * 1200 * 4
                sm = q["kernels_ms"].get("k_smooth_xgb", float("nan")) * 1e-3
                leg = {"workload": "configs[4] Gnofix half: chr1 WGS W=1431 A=12, xgb smoother + Gnofix, 2048 individuals (2 switch errors each) as 2-bit rows",
                       "individuals_per_s": q["individuals_per_s"], "kernels_ms": q["kernels_ms"], "alg_node_steps": steps,
                       "ceiling": "LDS gathers 256 CU x 2.4 GHz x 32 lanes (initial smoother pass)", "frac": steps / sm / (256 * 2.4e9 * 32),
                       "outputs_identical_to_int8": q["outputs_identical_to_int8"], "int8_individuals_per_s": r["int8_rows"]["individuals_per_s"],
                       "mean_switches": r["mean_switches"]}
            else:
                raise KeyError(name)
        except Exception as e:  # a leg must never take the headline line down
            leg = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
        leg["leg_wall_s"] = round(time.perf_counter() - t0, 2)
        legs[name] = _r(leg)
        if log:
            log("%s: %s" % (name, json.dumps(legs[name])[:400]))
    return legs


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
