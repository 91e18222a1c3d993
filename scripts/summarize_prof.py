#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof/<workload>/<pass>/...) into the small summaries kept under profiles/.

  python scripts/summarize_prof.py gpurun_out/prof profiles r02

writes per workload   profiles/<tag>_<workload>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, names shortened)
                      profiles/<tag>_<workload>_pmc.json           (per-kernel averages of every counter, one --pmc pass
                                                                    per group; `derived` adds busy fractions and HBM bytes)
                      profiles/<tag>_<workload>_result.jsonl       (the workload's own JSON line(s) from the stats pass)
and profiles/traffic_latest.json (HBM bytes per launch of the two bench.py kernels: FETCH_SIZE doubled as
MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads on gfx950; KB -> bytes).
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

N_CU, SIMD_PER_CU = 256, 4


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or ""))[:60] if m else name[:60]


# bench.py's kernel keys <- EXACT kernel names (the part before the template arguments).  A substring match once filed the parked
# k_smooth_xgb_bs under the dominant kernel's key and the driver-run line carried its 197 MB as the rank walk's traffic (591 MB).
TRAFFIC_KEYS = {
    "k_base_logistic_i8": "k_base_logistic", "k_base_logistic_i8_dl": "k_base_logistic",   # the int8-resident base pass (one of the two runs)
    "k_base_logistic_p2": "k_base_logistic_p2",
    "k_smooth_xgb_rk": "k_smooth_xgb",                                                       # the default tree smoother
    "k_smooth_xgb": "k_smooth_xgb_f32", "k_smooth_xgb_bs": "k_smooth_xgb_bs", "k_bs_ranks": "k_bs_ranks",
}


def traffic_table(pmc):
    """{bench.py kernel key: HBM bytes per launch} from one workload's per-kernel counters.  Two instantiations of ONE kernel (same
    name, other template arguments) share a key: the one with more launches is the workload's, the other is listed under
    "also_seen"; two different kernels never share a key."""
    out, owner, launches, also = {}, {}, {}, []
    for k, v in sorted(pmc.items()):
        b = v.get("derived", {}).get("hbm_bytes_per_launch")
        if b is None:
            continue
        base = k.split("<")[0]
        key = TRAFFIC_KEYS.get(base, base)
        n = v.get("FETCH_SIZE", {}).get("launches", 0)
        if key in out and launches[key] >= n:
            also.append(k)
            continue
        if key in out:
            also.append(owner[key])
        out[key], owner[key], launches[key] = b, k, n
    out["kernel_of_key"] = owner
    if also:
        out["also_seen"] = also
    return out


def one_workload(src, dst, tag, cfg):
    ks = glob.glob(os.path.join(src, cfg, "stats", "**", "*kernel_stats.csv"), recursive=True)
    avg_ns = {}
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        with open(os.path.join(dst, f"{tag}_{cfg}_kernel_stats.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
                avg_ns[short(r["Name"])] = float(r["AverageNs"])
    pmc = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(src, cfg, "*"))):
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not cc:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(cc[0])):
            n = r["Kernel_Name"]
            if "k_" not in n:
                continue
            agg[short(n)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[short(n)]["_meta"] = [(r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"],
                                       r["Grid_Size"], r["Workgroup_Size"])]
        for k, v in agg.items():
            for c, vals in v.items():
                if c == "_meta":
                    pmc[k]["vgpr,agpr,sgpr,lds,grid,wg"] = ",".join(vals[0])
                else:
                    pmc[k][c] = {"avg_per_launch": sum(vals) / len(vals), "launches": len(vals), "pass": os.path.basename(d)}
    for k, v in pmc.items():
        der = {}
        t = avg_ns.get(k)
        if t:
            der["avg_ns_kernel_trace"] = t
            cyc = t * 1e-9 * 2.4e9  # 2.4 GHz peak engine clock (MI355X_MICROARCH.md); counters summed over the chip
            g = lambda c: v[c]["avg_per_launch"] if c in v else None
            if g("SQ_LDS_IDX_ACTIVE") is not None:
                der["lds_pipe_busy_frac"] = g("SQ_LDS_IDX_ACTIVE") / (N_CU * cyc)
                if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
                    der["lds_conflict_frac_of_active"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
            if g("SQ_ACTIVE_INST_VALU") is not None:      # quad-cycles (MI355X_MICROARCH.md §counters): x4 = cycles, per SIMD
                der["valu_busy_frac"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / (N_CU * SIMD_PER_CU * cyc)
            if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                der["mfma_busy_frac"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (N_CU * SIMD_PER_CU * cyc)
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            der["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"]["avg_per_launch"] + v["WRITE_SIZE"]["avg_per_launch"]) * 1024.0
            if t:
                der["hbm_GBps"] = der["hbm_bytes_per_launch"] / (t * 1e-9) / 1e9
        v["derived"] = der
    if pmc:
        json.dump(pmc, open(os.path.join(dst, f"{tag}_{cfg}_pmc.json"), "w"), indent=1, sort_keys=True)
    res = os.path.join(src, cfg, "result.jsonl")
    if os.path.exists(res) and os.path.getsize(res):
        shutil.copy(res, os.path.join(dst, f"{tag}_{cfg}_result.jsonl"))
    return pmc


def main(src, dst, tag):
    os.makedirs(dst, exist_ok=True)
    for cfg in sorted(os.listdir(src)):
        if not os.path.isdir(os.path.join(src, cfg)):
            if cfg.endswith(".json"):
                shutil.copy(os.path.join(src, cfg), os.path.join(dst, f"{tag}_{cfg}"))
            continue
        pmc = one_workload(src, dst, tag, cfg)
        print(cfg, {k: {c: round(x, 3) for c, x in v.get("derived", {}).items()} for k, v in pmc.items()})
        if cfg == "bench":
            traffic = {"source": f"profiles/{tag}_bench_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)"}
            traffic.update(traffic_table(pmc))
            # stamp: the kernel sources these counters were collected from (bench.py prints counters.stale when its own differ)
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from bench import kernel_src_sha16
            traffic["kernel_src_sha16"] = kernel_src_sha16()
            json.dump(traffic, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
