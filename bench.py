#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config 2:
  chr22-like, 7 ancestries, logistic base + xgb smoother, 10k synthetic haplotypes per GPU.

One "step" = one pass of the hot path (X int8 resident in HBM -> B -> proba f32 + labels) over the
batch.  Haplotypes shard across ranks with no data-path collective (weak scaling: per-GPU batch fixed).

  python bench.py [--gpus N --steps K --warmup W]

N > 1 without a torch.distributed environment: bench.py launches its own N ranks (re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`), one process per GPU,
backend nccl (= RCCL); it refuses (non-zero exit, no JSON) when the box has fewer than N GPUs or when the
WORLD_SIZE it finds differs from --gpus.  GNX_BENCH_FORCE_DIST=1 runs the N=1 case through the same
process-group / RCCL code (what a 1-GPU box can exercise of it).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (dominant kernel, measured
with hipEvents on the launch stream inside libgnomix_hip), `e2e` (host pointers in, labels out, PCIe included —
never `value`) and, at N=1, `cpu_baseline` (the oracle's C restatement timed on ALL of this box's host cores on a
bounded sample, base and smoother timed separately) and `configs` (BASELINE.json's configs 3, 4, 5 at their one-GPU
shard sizes: scripts/bench_configs.py bench_legs(); never `value`).  The line is kept compact (the driver's record
holds its last ~8 KB: `configs`, `roofline` and `cpu_baseline` are printed last); what every field means is DESIGN.md 5;
the same object, indented, is also written to gpurun_out/bench_last.json.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
I8_MFMA_PEAK_TOPS = 3944.0 # MI355X_MICROARCH.md: int8 MFMA 16x16x64 measured ceiling (~2x bf16 dense)
F64_MFMA_PEAK_TF = 78.6    # AMD public spec for MI355X FP64 matrix (only used when GNX_BASE_LR_IMPL=f64)
N_CU, CLK_GHZ = 256, 2.4   # MI355X_MICROARCH.md: 256 CUs, 2.4 GHz peak engine clock
# LDS pipe: 128 B/clk/CU (MI355X_MICROARCH.md) = 32 four-byte lanes per clock.  A tree node-step needs at least ONE
# data-dependent LDS gather per lane (the feature the node asks for), so the LDS-bound ceiling of the tree pass is
# 256 CU x 2.4 GHz x 32 lanes = 1.966e13 node-steps/s.
LDS_PEAK_NODE_STEPS = N_CU * CLK_GHZ * 1e9 * 32


def kernel_src_sha16():
    """identity of the kernel sources a counter file was collected from / this run executes (there is no .git on the GPU box)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gnomix_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def checked_traffic(counters, kernel, alg_bytes):
    """HBM bytes per launch of `kernel` from the committed counter summary -> (bytes or None, note or None).  A kernel cannot move
    fewer bytes than its algorithmic ones: a counter below 0.95 x of them belongs to another kernel (round 5: the parked
    k_smooth_xgb_bs was filed under the rank walk's key) and is never printed."""
    t = counters.get(kernel) if isinstance(counters, dict) else None
    if not isinstance(t, (int, float)) or isinstance(t, bool):
        return None, None
    if t < 0.95 * alg_bytes:
        return None, "profiles/traffic_latest.json holds %.4g B for %s, below its algorithmic %.4g B: not this kernel's counter" % (t, kernel, alg_bytes)
    return float(t), None


def usable_cpus():
    """(logical CPUs this process may be scheduled on, CPUs' worth of time its cgroup allows — None when unlimited).  The GPU boxes of
    this project show 256 CPUs and a quota of 16 (cpu.max "1600000 100000"): threads beyond the quota are throttled, not run."""
    try:
        shown = len(os.sched_getaffinity(0))
    except AttributeError:
        shown = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = -(-int(q) // int(per))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = -(-q // per)
        except Exception:
            pass
    return shown, quota


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _self_launch(args):
    """--gpus N > 1 outside torchrun: spawn the N ranks ourselves (one process per GPU over RCCL)."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but this box exposes %d GPU(s): refusing to run (a smaller world would "
                         "silently report a different experiment)\n" % (args.gpus, have))
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--haps", type=int, default=10000, help="haplotypes per GPU per pass")
    ap.add_argument("--configs", type=int, default=1, help="0: skip the `configs` legs (BASELINE configs 3, 4, 5 on one GPU: c3, c4_2bit, c5a_2bit, c5b_resident)")
    ap.add_argument("--config-reps", type=int, default=3, help="timed repetitions per `configs` leg")
    ap.add_argument("--resident-2bit", type=int, default=1, help="0: skip the 2-bit-resident leg (resident_2bit, kernels.k_base_logistic_p2)")
    ap.add_argument("--passes", type=int, default=20, help="passes over the resident batch per step: a step is `passes` x `haps` haplotypes, so "
                    "that the 20 steps the driver asks for time > 1 s of device work instead of 57 ms")
    ap.add_argument("--trained", type=int, default=1, help="also time the smoother on a gnx_train_gbt-trained ensemble (N=1, rank 0)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU-baseline core-seconds budget scale (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the CPU baseline (0 = every core this process may use)")
    ap.add_argument("--e2e-steps", type=int, default=3, help="host-pointer (PCIe-inclusive) passes at N=1 (0 = skip)")
    ap.add_argument("--bgzf-level", type=int, default=6, help="deflate level of the BGZF leg's input (6 = bgzip's default; earlier rounds used 1)")
    ap.add_argument("--vcf-reps", type=int, default=3, help="file-to-file passes (VCF text in, .msp/.fb out) at N=1 (0 = skip)")
    ap.add_argument("--phase-leg", type=int, default=1, help="file-to-file with phase=True on admixed individuals through a device-trained model (N=1; needs --vcf-reps > 0)")
    ap.add_argument("--vcf-dir", default="", help="where the synthetic VCF and the outputs go (default: /dev/shm when it has room, else a temp dir)")
    ap.add_argument("--seed", type=int, default=94305)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    in_dist_env = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not in_dist_env and args.gpus > 1:
        sys.exit(_self_launch(args))
    rank = int(os.environ.get("RANK", "0")) if in_dist_env else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if in_dist_env else 1
    local = int(os.environ.get("LOCAL_RANK", "0")) if in_dist_env else 0
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to run" % (args.gpus, world))

    import numpy as np
    import torch
    import gnomix_amd
    from gnomix_amd import synth, _lib
    from gnomix_amd.dist import infer_sharded

    if not torch.cuda.is_available() or local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU %d" % (rank, local))
    dist = None
    backend = None
    if world > 1 or os.environ.get("GNX_BENCH_FORCE_DIST"):  # the env knob lets a 1-GPU box exercise the RCCL code path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        backend = dist.get_backend()
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if rank != 0:
        os.dup2(2, 1)   # only rank 0 owns stdout (see the end of main)

    cfg = dict(synth.CHR22)
    data = synth.synthetic_model(seed=0, n_rounds=100, **cfg)
    model = gnomix_amd.DeviceModel(data, ctx=_lib.Context(local))   # one gnx_ctx per process, bound to this rank's GPU
    ctx = model.ctx
    N = args.haps
    X = synth.synthetic_X_device(N, data.C, dev, seed=args.seed + rank)   # resident in HBM before timing
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    P = max(1, args.passes)
    for _ in range(args.warmup):
        for _ in range(P):
            out = model.infer_device(X)
    barrier()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for _ in range(P):
            out = model.infer_device(X)
    barrier()
    t1 = time.perf_counter()
    ctx.profile_enable(False)
    dt = t1 - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # gather-only epilogue OUTSIDE the timed region: the product's own gather (gnomix_amd.dist, dst = rank 0) moves every
    # rank's labels of the last step to rank 0, which checksums them
    lab_sum = None
    if dist is not None:
        # infer_sharded slices rows [lo, hi) of a "full" matrix; here every rank already holds its own shard, so the
        # shard function ignores the slice and returns this rank's outputs of the last step
        class _Rows:
            shape = (N * world, data.C)

            def __getitem__(self, sl):
                return X

        (lab_all,) = infer_sharded(lambda xs: (out[1],), _Rows(), dst=0)
        if rank == 0:
            assert lab_all.shape == (N * world, model.W)
            lab_sum = int(lab_all.to(torch.int64).sum().item())
    else:
        lab_sum = int(out[1].to(torch.int64).sum().item())

    W, A, C = model.W, model.A, model.C
    ms_base, n_base = ctx.profile_get(_lib.K_BASE_LOGISTIC)
    ms_sm, n_sm = ctx.profile_get(_lib.K_SMOOTH_XGB)
    avg_base = ms_base / max(n_base, 1) * 1e-3
    avg_sm = ms_sm / max(n_sm, 1) * 1e-3
    # algorithmic bytes per haplotype
    bytes_base = C + W * A * 4
    bytes_sm = 2 * W * A * 4 + W
    flops_base = 2.0 * A * (data.M + 2 * data.context) * W          # useful multiply-adds of the logits, as flops
    lr_impl = os.environ.get("GNX_BASE_LR_IMPL", "i8")
    # matrix-pipe work actually issued per haplotype by the exact int8 path: ~C/64 chunks x 7 digit planes x one
    # 16x16x64 MFMA per 16 haplotypes
    i8_ops = (C / 64.0) * 7 * (2 * 16 * 16 * 64) / 16.0
    node_steps = W * data.n_trees * 4
    kernels = {
        "k_base_logistic": {"avg_ms": avg_base * 1e3, "launches": n_base, "alg_GBps": bytes_base * N / avg_base / 1e9 if avg_base else None,
                            "hbm_frac": bytes_base * N / avg_base / 1e9 / HBM_PEAK_GBS if avg_base else None,
                            "impl": lr_impl, "logit_TFLOPs_equiv": flops_base * N / avg_base / 1e12 if avg_base else None,
                            "mfma_frac": (i8_ops * N / avg_base / 1e12 / I8_MFMA_PEAK_TOPS if lr_impl == "i8" else
                                          flops_base * N / avg_base / 1e12 / F64_MFMA_PEAK_TF) if avg_base else None},
        "k_smooth_xgb": {"avg_ms": avg_sm * 1e3, "launches": n_sm, "alg_GBps": bytes_sm * N / avg_sm / 1e9 if avg_sm else None,
                         "hbm_frac": bytes_sm * N / avg_sm / 1e9 / HBM_PEAK_GBS if avg_sm else None,
                         "node_steps_per_s": node_steps * N / avg_sm if avg_sm else None,
                         "lds_frac": node_steps * N / avg_sm / LDS_PEAK_NODE_STEPS if avg_sm else None},
    }
    # counters are NOT collected in this run (rocprofv3 --pmc needs its own passes: scripts/collect_profiles.sh);
    # what is printed under "counters" is the committed summary of the same command, named, never mixed with live times
    counters = {}
    try:
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp):
            counters = json.load(open(tp))
    except Exception:
        counters = {}
    dom = "k_smooth_xgb" if avg_sm >= avg_base else "k_base_logistic"
    alg_dom = (bytes_sm if dom == "k_smooth_xgb" else bytes_base) * N
    traffic, traffic_note = checked_traffic(counters, dom, alg_dom)
    if dom == "k_smooth_xgb":
        ach = kernels[dom]["node_steps_per_s"] or 0.0
        roofline = {"kernel": dom, "bound": "lds", "achieved": ach / 1e9, "peak": LDS_PEAK_NODE_STEPS / 1e9, "unit": "Gnode-steps/s",
                    "frac": ach / LDS_PEAK_NODE_STEPS, "traffic": traffic, "alg_bytes": alg_dom,
                    "hbm_view": {"achieved": kernels[dom]["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kernels[dom]["hbm_frac"]},
                    "note": "tree pass: %d B and %d node-steps per haplotype: LDS-gather bound (peak = 256 CU x 2.4 GHz x 32 lanes, one gather per "
                            "node-step), hbm_view = the HBM reading of the same launch; the HBM-bound kernel of the path is kernels.k_base_logistic "
                            "(%.1f%% of 8 TB/s)" % (bytes_sm, node_steps, 100 * (kernels["k_base_logistic"]["hbm_frac"] or 0))}
    else:
        ach = kernels[dom]["alg_GBps"] or 0.0
        roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic, "alg_bytes": alg_dom,
                    "note": "algorithmic bytes/launch = %d B/haplotype x %d haplotypes (SURVEY.md 8d)" % (bytes_base, N)}

    if traffic_note:
        roofline["traffic_note"] = traffic_note
    hps = world * N * P * args.steps / dt
    res = {
        "metric": "haplotypes/sec (+ windows/sec) chr22 7-ancestry inference",
        "value": hps, "unit": "haplotypes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "passes_per_step": P, "ms_per_pass": dt / args.steps / P * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int8 x 7-digit fixed-point weights -> exact i32/i64 logits, f64 sigmoid (logistic base); f32 (tree smoother)",
        "data": "synthetic",
        "config": {"workload": "configs[1]: chr22-like C=370500 M=1000 ctx=500 W=370 A=7 S=75, logistic base + xgb smoother (100 rounds x 7 "
                               "uniform-random depth-4 trees), %d synthetic haplotypes per GPU resident in HBM as int8, a step = %d passes" % (N, P),
                   "haplotypes_per_gpu": N, "haplotypes_per_step_per_gpu": N * P, "sharding": "haplotypes across ranks, no data-path collective",
                   "dist_backend": backend},
        "windows_per_s": hps * W, "label_checksum": lab_sum,
    }
    if counters:
        sha = kernel_src_sha16()
        res["counters"] = {"source": counters.get("source", "profiles/traffic_latest.json"), "hbm_bytes_per_launch": {
            k: v for k, v in counters.items() if isinstance(v, (int, float))}, "kernel_of_key": counters.get("kernel_of_key"),
            "collected_at_kernel_src": counters.get("kernel_src_sha16"), "this_run_kernel_src": sha,
            "stale": counters.get("kernel_src_sha16") != sha}
        if res["counters"]["stale"]:
            roofline["traffic"] = None        # never print a counter next to times of other kernels
            roofline["traffic_note"] = "profiles/traffic_latest.json was collected from other kernel sources: rerun scripts/collect_profiles.sh"

    # ---- the same batch resident in HBM as 2-bit rows (gnx_pack_x layout): the logistic pass reads a quarter of the X bytes
    # (k_base_logistic_p2); never `value` (SURVEY.md 8d quotes the metric on int8-resident X), printed beside it -------------
    if rank == 0 and world == 1 and args.resident_2bit:
        try:
            Pk = model.pack_device(X)
            for _ in range(max(1, args.warmup)):
                o2 = model.infer_packed_device(Pk)
            torch.cuda.synchronize()
            ctx.profile_reset()
            ctx.profile_enable(True)
            t0 = time.perf_counter()
            n2 = max(1, args.steps) * P
            for _ in range(n2):
                o2 = model.infer_packed_device(Pk)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            ctx.profile_enable(False)
            ms_b2, n_b2 = ctx.profile_get(_lib.K_BASE_LOGISTIC)
            ms_s2, n_s2 = ctx.profile_get(_lib.K_SMOOTH_XGB)
            avg_b2 = ms_b2 / max(n_b2, 1) * 1e-3
            bytes_p2 = (C + 3) // 4 + W * A * 4
            res["resident_2bit"] = {
                "ms_per_pass": dt2 / n2 * 1e3, "haplotypes_per_s": N * n2 / dt2, "base_ms": avg_b2 * 1e3, "smoother_ms": ms_s2 / max(n_s2, 1),
                "outputs_identical_to_int8": bool(torch.equal(o2[0], out[0]) and torch.equal(o2[1], out[1])),
                "resident_bytes_per_haplotype": int(Pk.shape[1])}
            kernels["k_base_logistic_p2"] = {
                "avg_ms": avg_b2 * 1e3, "launches": n_b2, "alg_GBps": bytes_p2 * N / avg_b2 / 1e9 if avg_b2 else None,
                "hbm_frac": bytes_p2 * N / avg_b2 / 1e9 / HBM_PEAK_GBS if avg_b2 else None,
                "alg_bytes_per_haplotype": bytes_p2, "int8_equiv_GBps": bytes_base * N / avg_b2 / 1e9 if avg_b2 else None,
                "mfma_frac": i8_ops * N / avg_b2 / 1e12 / I8_MFMA_PEAK_TOPS if avg_b2 else None}
            del Pk, o2
        except Exception as e:
            res["resident_2bit"] = {"error": repr(e)}

    # ---- PCIe-inclusive rate: host pointers in, labels + probabilities out (never `value`) ----------------------------
    if rank == 0 and world == 1 and args.e2e_steps > 0:
        try:
            res["e2e"] = _e2e(model, X, args.e2e_steps, out)
        except Exception as e:  # the headline line must still print
            res["e2e"] = {"error": repr(e)}

    if rank == 0 and world == 1 and args.trained:
        try:
            te = _trained_ensemble(ctx, data, avg_sm * 1e3)
            res["trained_ensemble"] = te
            # the same kernel on what it is for (a trained ensemble, ancestry tracts) next to the random-tree worst case of `value`
            ms = te["trained_trees_tract_inputs_ms"]
            ti = {"ms": ms, "frac": te["trained_node_steps"] * te["haplotypes"] / (ms * 1e-3) / LDS_PEAK_NODE_STEPS, "lds_conflict_frac": None}
            try:
                tc = json.load(open(os.path.join(ROOT, "profiles", "trained_inputs_latest.json")))
                ti["lds_conflict_frac"] = tc.get("lds_conflict_frac_of_active")
                ti["lds_conflict_source"] = tc.get("source")
                ti["lds_conflict_stale"] = tc.get("kernel_src_sha16") != kernel_src_sha16()   # (a committed counter, not a live one)
            except Exception:
                pass
            if dom == "k_smooth_xgb":
                roofline["trained_inputs"] = ti
        except Exception as e:
            res["trained_ensemble"] = {"error": repr(e)}

    # ---- file to file: synthetic phased VCF -> .msp / .fb through the command line's own run_inference (never `value`) ----
    # at N > 1 it is ONE process (rank 0) driving all N devices (gnomix_amd/multi.py: one parse, one context and host thread per
    # GPU, outputs written once) while the other ranks wait — the file path has no collective either
    if rank == 0 and args.vcf_reps > 0:
        try:
            # GNX_BENCH_VCF_DEVICES="0,0": the one-process multi-context file path on a box with fewer GPUs (tests)
            vdev = [int(t) for t in os.environ["GNX_BENCH_VCF_DEVICES"].split(",")] if os.environ.get("GNX_BENCH_VCF_DEVICES") else None
            res["e2e_vcf"] = _e2e_vcf(args, model, data, X, out, devices=vdev if vdev else (list(range(world)) if world > 1 else None))
        except Exception as e:
            res["e2e_vcf"] = {"error": repr(e)}
    if dist is not None:
        # the other ranks wait for rank 0's file leg (a minute of work that drives THEIR GPUs too) on the rendezvous store, on the CPU:
        # an RCCL barrier would sit on every GPU as a spinning kernel for that long, beside the leg's own kernels
        try:
            from datetime import timedelta
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("gnx_bench_rank0_legs_done", "1")
            else:
                store.wait(["gnx_bench_rank0_legs_done"], timedelta(minutes=30))
        except Exception:
            pass
        dist.barrier()

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        res["cpu_baseline"] = _cpu_baseline(args, data, X, out)
    res["kernels"] = kernels
    res["roofline"] = roofline
    # ---- BASELINE.json's configs 3, 4, 5 at their one-GPU shard sizes (scripts/bench_configs.py; never `value`) -------------
    if rank == 0 and world == 1 and args.configs:
        try:
            del X, out
            torch.cuda.empty_cache()
            import importlib.util
            spec = importlib.util.spec_from_file_location("gnx_bench_configs", os.path.join(ROOT, "scripts", "bench_configs.py"))
            bc = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(bc)
            res["configs"] = bc.bench_legs(ctx=ctx, reps=max(1, args.config_reps), log=lambda m: sys.stderr.write("bench.py configs " + m + "\n"))
        except Exception as e:
            res["configs"] = {"error": repr(e)[:300]}
    if dist is not None:
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        try:   # the whole object, indented, where gpurun merges it back (the driver's record keeps the tail of stdout only)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            json.dump(res, open(os.path.join(ROOT, "gpurun_out", "bench_last.json"), "w"), indent=1)
        except Exception:
            pass
        print(json.dumps(_compact(res)), flush=True)
    # RCCL prints a version banner to STDOUT from a library destructor at process exit (after anything Python can
    # print): point fd 1 at stderr from here on, on every rank, so the JSON line stays the last line of stdout
    os.dup2(2, 1)


def _compact(x, nd=5, notes=False):
    """the printed line: floats at `nd` significant digits, explanatory "note" strings dropped except under `roofline` (full precision
    and every note stay in gpurun_out/bench_last.json; DESIGN.md 5 says what each field is)"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _compact(v, nd, notes or k == "roofline") for k, v in x.items() if notes or k != "note"}
    if isinstance(x, (list, tuple)):
        return [_compact(v, nd, notes) for v in x]
    return x


def _e2e(model, X, steps, out_dev):
    """X in page-locked host memory -> gnx_infer* (H2D, kernels, D2H of labels + probabilities) -> host arrays"""
    import numpy as np
    import torch
    ctx = model.ctx
    N, C = X.shape
    Xh = ctx.pinned_empty((N, C), np.int8)
    torch.from_numpy(Xh).copy_(X.cpu())
    res = {}
    outs = (ctx.pinned_empty((N, model.W, model.A), np.float32), ctx.pinned_empty((N, model.W), np.int32))  # page-locked, reused
    p, lab = model.infer(Xh, out=outs)     # warm-up (workspaces)
    t0 = time.perf_counter()
    for _ in range(steps):
        p, lab = model.infer(Xh, out=outs)
    dt = (time.perf_counter() - t0) / steps
    p, lab = p.copy(), lab.copy()
    same = bool((torch.from_numpy(lab) == out_dev[1].cpu()).all())
    res["int8"] = {"haplotypes_per_s": N / dt, "ms": dt * 1e3, "x_GBps": N * C / dt / 1e9, "labels_equal_device_path": same}
    if hasattr(model, "infer_packed"):
        Xp = model.pack_x(Xh)
        tp = time.perf_counter()
        model.pack_x(Xh, out=Xp)
        tp = time.perf_counter() - tp
        res["host_pack_int8_to_2bit"] = {"ms": tp * 1e3, "GBps_of_int8": N * C / tp / 1e9, "note": "gnx_pack_x on the host's cores; a VCF "
                                         "reader that emits 2-bit fields directly skips this pass (it is not part of packed2bit.ms)"}
        p2, lab2 = model.infer_packed(Xp, N, out=outs)
        t0 = time.perf_counter()
        for _ in range(steps):
            p2, lab2 = model.infer_packed(Xp, N, out=outs)
        dt2 = (time.perf_counter() - t0) / steps
        res["packed2bit"] = {"haplotypes_per_s": N / dt2, "ms": dt2 * 1e3, "x_GBps": Xp.nbytes / dt2 / 1e9,
                             "bit_identical_to_int8_path": bool(np.array_equal(lab2, lab) and np.array_equal(p2, p))}
    res["e2e_haplotypes_per_s"] = max(v["haplotypes_per_s"] for v in res.values() if isinstance(v, dict) and "haplotypes_per_s" in v)
    res["note"] = "host pointers in and out (page-locked arrays), probabilities f32 + labels i32 out, PCIe both ways included; never `value`"
    return res


def _trained_ensemble(ctx, data, bench_ms):
    """The tree smoother alone on an ensemble TRAINED by gnx_train_gbt (same size: 100 rounds x A trees, depth 4) and on inputs
    with ancestry tracts, next to the bench's uniform-random trees on the same inputs.  Random trees and unstructured inputs send
    neighbouring lanes to unrelated nodes (worst case for divergence and LDS bank conflicts); the headline `value` keeps them."""
    import numpy as np
    import torch
    import gnomix_amd
    from gnomix_amd import synth, train, _lib
    W, A, S = data.C // data.M, data.A, data.S
    N = 10000
    rng = np.random.RandomState(3)

    def noisy(Bc, sd):
        B = np.clip(Bc + rng.normal(0, sd, Bc.shape), 1e-4, None)
        return B / B.sum(-1, keepdims=True)
    Bt = synth.synthetic_phased_individuals(500, W, A, seed=5, phase_errors=0, noise=0.02)
    yt = np.argmax(Bt, -1).astype(np.int32)
    trees, _ = train.train_gbt_arrays(noisy(Bt, 0.45), yt, S, ctx=ctx)
    Bq = noisy(synth.synthetic_phased_individuals(N // 2, W, A, seed=9, phase_errors=0, noise=0.02), 0.45).astype(np.float32)
    Bd = torch.from_numpy(Bq).cuda()
    d_tr = synth.synthetic_model(C=data.C, M=data.M, A=A, S=S, n_rounds=1, seed=1)
    for k, v in trees.items():
        setattr(d_tr, k, v)
    out = {}
    for name, d in (("random_trees_tract_inputs_ms", data), ("trained_trees_tract_inputs_ms", d_tr)):
        m = gnomix_amd.DeviceModel(d, ctx=ctx)
        m.smooth_predict_device(Bd)
        torch.cuda.synchronize()
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(10):
            m.smooth_predict_device(Bd)
        torch.cuda.synchronize()
        ctx.profile_enable(False)
        ms, n = ctx.profile_get(_lib.K_SMOOTH_XGB)
        out[name] = ms / max(n, 1)
        if d is not data:
            out["trained_nodes"] = int(len(d.left))
            out["trained_node_steps"] = int(W * (len(d.tree_off) - 1) * 4)
            out["haplotypes"] = N
            m.close()
    out["random_trees_bench_inputs_ms"] = bench_ms
    out["random_nodes"] = int(len(data.left))
    out["note"] = "k_smooth_xgb average launch, %d haplotypes x %d windows; tract inputs = synthetic admixed individuals + noise" % (N, W)
    return out


GENOME_SNPS = 17_738_000   # sum of C over the 22 chromosome models (chr22: 370 500)


def _vcf_legs(run, reps):
    """`reps` timed passes of run(T) (T receives the stage seconds) -> best / median of the WARM passes (the first creates page-locked
    buffers and the worker pool: reported as first_pass_s), stages of the best one"""
    rows = []
    for _ in range(reps):
        T = {}
        t0 = time.perf_counter()
        run(T)
        T["total"] = time.perf_counter() - t0
        rows.append(T)
    warm = rows[1:] if len(rows) > 1 else rows
    tot = sorted(t["total"] for t in warm)
    best = min(warm, key=lambda t: t["total"])
    return {"seconds": best["total"], "median_s": tot[len(tot) // 2], "first_pass_s": round(rows[0]["total"], 4), "passes": len(rows),
            "stages_s": {k: round(v, 4) for k, v in best.items()}}


def _e2e_vcf(args, model, data, X, out_dev, devices=None):
    """`north_star`: "throughput on synthetic phased VCFs".  The batch of the headline run is written as a phased VCF (GT-only
    records, '.' for missing calls; outside the timed region), then gnomix_amd.cli.run_inference — the command line's own
    function — takes it from TEXT to query_results.msp / .fb: native parse on every host core -> 2-bit rows over PCIe -> X built
    in HBM -> base + smoother -> native formatting.  Legs: the plain text; the same file as BGZF (the reference's demo query is a
    .vcf.gz: src/utils.py:64-66); phase=True on tract-structured admixed individuals with switch errors through a model TRAINED on
    the device (gnx_train_logistic + gnx_train_gbt); the whole command line as a fresh process.  `devices`: GPU ordinals of a
    one-process multi-GPU run (gnomix_amd/multi.py)."""
    import shutil
    import tempfile
    import numpy as np
    import torch
    from gnomix_amd import HipGnomix, cli, synth, vcfio
    N, C = X.shape
    ns = N // 2
    rng = np.random.RandomState(7)
    data.snp_pos = (16_050_000 + np.cumsum(rng.randint(1, 180, size=C))).astype(np.int64)
    data.snp_ref = rng.choice(list("ACGT"), size=C)
    data.snp_alt = rng.choice(list("ACGT"), size=C)
    data.gen_map_pos = np.array([16_000_000, 30_000_000, 52_000_000])
    data.gen_map_cm = np.array([0.0, 31.5, 74.1])
    need = 4 * ns * C + (64 << 20) + N * model.W * model.A * 14
    root = args.vcf_dir
    if not root:
        root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 3 * need else tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="gnx_e2e_", dir=root)
    cpus = usable_cpus()
    io_threads = min(cpus[0], cpus[1] or cpus[0])
    try:
        ctx = model.ctx

        def write_query(Xd, path):   # X (HBM) -> variant-major 2-bit rows -> text
            n = Xd.shape[0]
            ldg = (n + 15) // 16 * 4
            cols = torch.arange(C, dtype=torch.int32, device=Xd.device)
            Gd = torch.zeros((C, ldg), dtype=torch.uint8, device=Xd.device)
            model._bind_torch_stream()
            ctx.check(ctx.lib.gnx_x_to_gt2_dev(ctx.h, Xd.data_ptr(), n, Xd.stride(0), 0, cols.data_ptr(), C, Gd.data_ptr(), ldg))
            torch.cuda.synchronize()
            G = Gd.cpu().numpy()
            del Gd
            t0 = time.perf_counter()
            synth.write_vcf_gt2(path, G, n // 2, data.snp_pos, data.snp_ref, data.snp_alt, chrom="22")
            return time.perf_counter() - t0
        vcf_path = os.path.join(work, "query.vcf")
        t_gen = write_query(X, vcf_path)
        vcf_bytes = os.path.getsize(vcf_path)
        gm = HipGnomix(data, ctx=ctx)
        group = None
        if devices is not None and len(devices) > 1:
            from gnomix_amd.multi import DeviceGroup
            group = DeviceGroup(data, devices, first=gm.dev)
        base_args = {"query_file": vcf_path, "chm": "22", "output_basename": work, "phase": False}
        leg = _vcf_legs(lambda T: cli.run_inference(base_args, gm, verbose=False, timings=T, devices=group), args.vcf_reps)
        best = leg["stages_s"]
        msp = open(os.path.join(work, "query_results.msp")).read().split("\n")[2:2 + model.W]
        lab_file = np.stack([np.array(ln.split("\t")[6:], dtype=np.int32) for ln in msp], axis=1)
        same = bool(np.array_equal(lab_file, out_dev[1].cpu().numpy()))
        fb_bytes = os.path.getsize(os.path.join(work, "query_results.fb"))
        msp_bytes = os.path.getsize(os.path.join(work, "query_results.msp"))
        msp_text = open(os.path.join(work, "query_results.msp")).read()
        text_per_hap_genome = vcf_bytes / N * (GENOME_SNPS / C)
        parse_rate = vcf_bytes / best["read_vcf"]
        res = {"haplotypes_per_s": N / leg["seconds"], "haplotypes_per_s_median": N / leg["median_s"], "seconds": leg["seconds"],
               "median_s": leg["median_s"], "stages_s": best, "first_pass_s": leg["first_pass_s"], "vcf_GB": vcf_bytes / 1e9,
               "parse_GBps": parse_rate / 1e9, "write_MBps": (fb_bytes + msp_bytes) / (best["write_fb"] + best["write_msp"]) / 1e6,
               "fb_MB": fb_bytes / 1e6, "host_cpus_shown": cpus[0], "host_cpu_quota": cpus[1], "vcf_written_s": round(t_gen, 3), "dir": root,
               "devices": list(devices) if devices is not None else [ctx.device], "msp_labels_equal_device_path": same,
               "whole_genome_from_text": {"text_MB_per_haplotype": text_per_hap_genome / 1e6,
                                          "parse_bound_haplotypes_per_s": parse_rate / text_per_hap_genome,
                                          "note": "22 chromosomes = %.1f x this file's SNPs: at the parse rate measured here ONE host reads whole-genome "
                                                  "text for this many haplotypes per second whatever the number of GPUs — the north star's 50 k "
                                                  "haplotypes/s is a device-resident / packed-input figure (DESIGN.md 5.1), not a from-text one" % (GENOME_SNPS / C)},
               "note": "chr22 x %d samples as VCF TEXT in, query_results.msp + .fb out, through gnomix_amd.cli.run_inference; best and median of the "
                       "%d warm passes in this process (first_pass_s includes page-locked allocations and the worker pool's start); never `value`"
                       % (ns, max(args.vcf_reps - 1, 1))}
        # ---- the same query as BGZF ----
        try:
            t0 = time.perf_counter()
            gz_path = synth.bgzf_compress_file(vcf_path, os.path.join(work, "query.vcf.gz"), n_threads=io_threads, level=args.bgzf_level)
            t_gz = time.perf_counter() - t0
            gz_args = dict(base_args, query_file=gz_path)
            lg = _vcf_legs(lambda T: cli.run_inference(gz_args, gm, verbose=False, timings=T, devices=group), min(args.vcf_reps, 3))
            res["bgzf"] = {"haplotypes_per_s": N / lg["seconds"], "haplotypes_per_s_median": N / lg["median_s"], "seconds": lg["seconds"],
                           "stages_s": lg["stages_s"], "file_GB": os.path.getsize(gz_path) / 1e9, "text_GBps": vcf_bytes / lg["stages_s"]["read_vcf"] / 1e9,
                           "compressed_in_s": round(t_gz, 2), "deflate_level": args.bgzf_level,
                           "msp_identical": open(os.path.join(work, "query_results.msp")).read() == msp_text,
                           "note": "deflate level 6 = bgzip's (zlib's) default, what a .vcf.gz in the wild is written with; level 1 streams (the "
                                   "figure of earlier rounds: shorter matches, more literals) inflate 1.7x slower per byte of text "
                                   "(scripts/dev/inflate_bench.py)"}
            os.remove(gz_path)
        except Exception as e:
            res["bgzf"] = {"error": repr(e)}
        # the whole command line as a fresh process: interpreter + library + gnx_init + model load + the above
        try:
            mp = os.path.join(work, "model.gnx")
            data.save(mp)
            outdir = os.path.join(work, "cli")
            t0 = time.perf_counter()
            env = dict(os.environ, GNX_CLI_TIMING="1")
            env.pop("GNX_NO_TORCH", None)
            if devices is not None:
                env["GNX_DEVICES"] = ",".join(str(d) for d in devices)
            # the first start of a model writes the cache of its prepared planes beside it (<model>.gnx.planes); every later start —
            # the timed one — reads it
            subprocess.run([sys.executable, os.path.join(ROOT, "gnomix.py"), vcf_path, outdir, "22", "False", mp],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=work, env=env)
            first_start = time.perf_counter() - t0
            shutil.rmtree(outdir, ignore_errors=True)
            t0 = time.perf_counter()
            pr = subprocess.run([sys.executable, os.path.join(ROOT, "gnomix.py"), vcf_path, outdir, "22", "False", mp],
                                stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd=work, env=env, text=True)
            rc = pr.returncode
            dt = time.perf_counter() - t0
            stages = [ln for ln in pr.stderr.splitlines() if ln.startswith("gnomix_amd timings")]
            same_cli = rc == 0 and open(os.path.join(outdir, "query_results.msp")).read() == msp_text
            res["cli_process"] = {"wall_s": round(dt, 3), "first_start_wall_s": round(first_start, 3), "haplotypes_per_s": N / dt, "rc": rc, "msp_identical": bool(same_cli),
                                  "stages": stages[-1] if stages else None}
            shutil.rmtree(outdir, ignore_errors=True)
        except Exception as e:
            res["cli_process"] = {"error": repr(e)}
        os.remove(vcf_path)
        # ---- phase=True: a model trained on the device, admixed individuals with two switch errors each ----
        if args.phase_leg:
            try:
                res["phase"] = _e2e_phase(args, ctx, data, work, write_query, devices)
            except Exception as e:
                res["phase"] = {"error": repr(e)}
        if group is not None:
            for m in group.models[1:]:
                m.close()
        return res
    finally:
        shutil.rmtree(work, ignore_errors=True)


def _trained_model(ctx, data, dev, n1=2000, n2=600, seed=11):
    """a chr22-shaped model TRAINED on the device from synthetic admixed haplotypes (Gnomix.train: logistic base on train1, tree
    smoother on the base's probabilities of train2) -> (HipGnomix, allele frequencies, seconds)"""
    import numpy as np
    import gnomix_amd
    from gnomix_amd import synth
    d = synth.synthetic_model(C=data.C, M=data.M, A=data.A, S=data.S, n_rounds=1, seed=1)
    for k in ("snp_pos", "snp_ref", "snp_alt", "gen_map_pos", "gen_map_cm", "population_order"):
        setattr(d, k, getattr(data, k, None))
    X1, y1, f = synth.synthetic_admixed_device(n1 // 2, data.C, data.M, data.A, dev, seed=seed)
    X2, y2, _ = synth.synthetic_admixed_device(n2 // 2, data.C, data.M, data.A, dev, seed=seed + 1, freqs=f)
    gm = gnomix_amd.HipGnomix(d, ctx=ctx)
    t0 = time.perf_counter()
    gm.train(((X1.cpu().numpy(), y1), (X2.cpu().numpy(), y2), (None, None)), retrain_base=False, evaluate=False)
    return gm, f, time.perf_counter() - t0


def _e2e_phase(args, ctx, data, work, write_query, devices):
    import numpy as np
    import torch
    from gnomix_amd import cli, synth
    dev = torch.device("cuda", ctx.device)
    gm, f, t_train = _trained_model(ctx, data, dev)
    n_ind = args.haps // 2
    Xq, _, _ = synth.synthetic_admixed_device(n_ind, data.C, data.M, data.A, dev, seed=23, freqs=f, phase_errors=2)
    path = os.path.join(work, "admixed.vcf")
    write_query(Xq, path)
    vb = os.path.getsize(path)
    group = None
    if devices is not None and len(devices) > 1:
        from gnomix_amd.multi import DeviceGroup
        group = DeviceGroup(gm.data, devices, first=gm.dev)
    out = {}
    for name, ph in (("phase_true", True), ("phase_false", False)):
        a = {"query_file": path, "chm": "22", "output_basename": work, "phase": ph}
        lg = _vcf_legs(lambda T: cli.run_inference(a, gm, verbose=False, timings=T, devices=group), 3 if ph else 2)
        out[name] = {"haplotypes_per_s": 2 * n_ind / lg["seconds"], "seconds": lg["seconds"], "median_s": lg["median_s"], "stages_s": lg["stages_s"]}
    _, _, _, nsw = (group or gm.dev).phase_gt2(*_gt2_of(path, gm, ctx))
    out["mean_switches_per_individual"] = float(np.mean(nsw))
    out["vcf_GB"] = vb / 1e9
    out["model"] = "gnx_train_logistic (2000 admixed haplotypes) + gnx_train_gbt (600), %.1f s outside the timed region" % t_train
    out["note"] = ("chr22 x %d admixed individuals (ancestry tracts, 2 switch errors each) as VCF text -> Gnofix -> query_results.msp / .fb + "
                   "query_file_phased.vcf; the random-tree / unstructured-haplotype file of the other legs is the WORST case for Gnofix "
                   "(a label change at almost every window: DESIGN.md 4.11) and is not what phase=True is for" % n_ind)
    if group is not None:
        for m in group.models[1:]:
            m.close()
    os.remove(path)
    return out


def _gt2_of(path, gm, ctx):
    from gnomix_amd import vcfio
    vcf = vcfio.read_vcf(path, chm="22", ctx=ctx)
    src, _, _ = vcfio.column_map(vcf, gm.snp_pos, gm.snp_ref, verbose=False)
    return vcf.gt2, 2 * vcf.n_samples, src


def _cpu_baseline(args, data, X, out_dev):
    """The oracle's C restatement on every host core, with the work cut the way a CPU wants it: the logistic base by WINDOWS (a
    core keeps a window's A x M_ float64 weights in its cache and runs the whole sample through them — cutting by haplotypes
    makes every core stream all 41 MB of weights from DRAM, which is what round 2's 1 288 haplotypes/s on 256 threads measured),
    the tree smoother by haplotypes (the 90 KB of trees stay in every core's L2).  Beside it the reference's own arithmetic for
    the base: one BLAS product per window (sklearn's predict_proba), numpy's BLAS threads."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import gnx_oracle as O
    O.build()
    T = O.Trees(data.tree_off, data.left, data.right, data.feat, data.cond, data.tree_class, data.A, data.base_score)
    avail, quota = usable_cpus()
    cores = max(1, args.cpu_threads or min(avail, quota or avail))   # one thread per CPU the cgroup lets run
    Xh = X.cpu().numpy()
    W = data.C // data.M
    # one core alone first (also sizes the sample)
    c0 = time.perf_counter()
    B4 = O.base_lr(Xh[:4], data.M, data.context, data.lr_coef, data.lr_intercept)
    c1 = time.perf_counter()
    O.smooth_xgb(T, B4, data.S)
    c2 = time.perf_counter()
    per = (c2 - c0) / 4
    # bounded sample: ~cpu_seconds of wall time with every core busy, never more than the batch
    n_s = int(min(Xh.shape[0], max(cores, cores * args.cpu_seconds / max(per, 1e-6))))
    n_s = max(cores, n_s - n_s % cores) if n_s >= cores else n_s
    Xs = np.ascontiguousarray(Xh[:n_s])
    B = np.empty((n_s, W, data.A), np.float64)
    chunk = max(1, n_s // cores)
    sl = [slice(i, min(n_s, i + chunk)) for i in range(0, n_s, chunk)]
    with ThreadPoolExecutor(max_workers=cores) as pool:
        c0 = time.perf_counter()
        list(pool.map(lambda w: O.base_lr_windows(Xs, data.M, data.context, data.lr_coef, data.lr_intercept, w, w + 1, B), range(W)))
        c1 = time.perf_counter()
        parts = list(pool.map(lambda s_: O.smooth_xgb(T, B[s_], data.S), sl))
        c2 = time.perf_counter()
    l_ref = np.concatenate([p[1] for p in parts])
    same = bool((out_dev[1][:n_s].cpu().numpy() == l_ref).all())
    t_base, t_sm = c1 - c0, c2 - c1
    # the reference's arithmetic for the base leg: one BLAS product per window (what sklearn's predict_proba does), cut by windows
    # over the same thread pool, every BLAS call single-threaded (numpy's pool would otherwise start one thread per CPU the host
    # SHOWS — 256 on the GPU boxes — inside a 16-CPU quota: round 3 measured 173 haplotypes/s that way)
    n_b = int(min(n_s, 2000))
    blas_threads = None
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1, user_api="blas")
        blas_threads = 1
    except Exception:
        limit = None
    Xb = Xs[:n_b]
    Bb = np.empty((n_b, W, data.A), np.float64)
    Xpad = O.base_lr_pad(Xb, data.context)
    wr = [(w, min(W, w + max(1, W // (4 * cores)))) for w in range(0, W, max(1, W // (4 * cores)))]
    with ThreadPoolExecutor(max_workers=cores) as pool:
        c0 = time.perf_counter()
        list(pool.map(lambda r: O.base_lr_blas(Xb, data.M, data.context, data.lr_coef, data.lr_intercept, r[0], r[1], Bb, Xpad), wr))
        t_blas = time.perf_counter() - c0
    if limit is not None:
        limit.restore_original_limits() if hasattr(limit, "restore_original_limits") else limit.unregister()
    blas_err = float(np.abs(Bb - B[:n_b]).max())
    # the stated baseline: the FASTER CPU arithmetic of each leg — the reference's own base arithmetic (one BLAS product per window,
    # float64) and the port's tree smoother (xgboost itself is absent from the image) — each scaled to one haplotype and added
    t_base_port = t_base / n_s
    t_base_blas = t_blas / n_b
    t_base_best = min(t_base_port, t_base_blas)
    composite = 1.0 / (t_base_best + t_sm / n_s)
    return {"value": composite, "unit": "haplotypes/s", "cores": cores, "kind": "blas+port",
            "components": {"base": "base_lr_blas (numpy / BLAS, the reference's arithmetic)" if t_base_blas <= t_base_port else "oracle port (scalar C)",
                           "base_haplotypes_per_s": 1.0 / t_base_best, "base_threads": cores, "blas_threads_per_call": blas_threads,
                           "smoother": "oracle port (scalar C walker, float32 sums in tree order)", "smoother_haplotypes_per_s": n_s / t_sm,
                           "smoother_threads": cores},
            "port_only_haplotypes_per_s": n_s / (t_base + t_sm),
            "base_lr_haplotypes_per_s": n_s / t_base, "smooth_xgb_haplotypes_per_s": n_s / t_sm,
            "one_core_haplotypes_per_s": 1.0 / per, "parallel_efficiency": (n_s / (t_base + t_sm)) / (cores / per),
            "base_lr_blas": {"haplotypes_per_s": n_b / t_blas, "sample": n_b, "max_abs_diff_vs_port": blas_err,
                             "threads": cores, "blas_threads_per_call": blas_threads,
                             "note": "numpy: Xw.astype(float64) @ coef.T per window (what sklearn's predict_proba does), windows cut over the "
                                     "same %d threads as the port, one BLAS thread per call (threadpoolctl)" % cores},
            "labels_identical_to_gpu_on_sample": same,
            "sample": "%d haplotypes of the same workload on %d threads (host shows %d CPUs, cgroup quota %s): base = numpy/BLAS product per window "
                      "(the reference's arithmetic) or the oracle's scalar C port, whichever is faster; tree smoother = the oracle's C port "
                      "(xgboost is absent from the image); base by windows %.2f s, smoother by haplotypes %.2f s wall" %
                      (n_s, cores, os.cpu_count() or 0, quota, t_base, t_sm)}


if __name__ == "__main__":
    main()
