#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the small summaries kept under profiles/.

  python scripts/summarize_prof.py gpurun_out profiles r01

writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, kernel names truncated),
profiles/<tag>_pmc.json (per-kernel averages of every collected counter, one --pmc pass per group)
and profiles/traffic_latest.json (HBM bytes per launch per kernel, FETCH_SIZE doubled as
MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads on gfx950; KB -> bytes).
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    for k in ("k_base_logistic", "k_smooth_xgb", "k_smooth_rows", "k_base_covrsk", "k_smooth_crf", "k_gnofix"):
        if k in name:
            return name[name.index(k):][:40]
    return name[:60]


def main(src, dst, tag):
    os.makedirs(dst, exist_ok=True)
    ks = glob.glob(os.path.join(src, "prof_stats", "*kernel_stats.csv"))
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
    pmc = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(src, "prof_*"))):
        cc = glob.glob(os.path.join(d, "*counter_collection.csv"))
        if not cc:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(cc[0])):
            n = r["Kernel_Name"]
            if "k_" not in n or "anonymous" not in n:
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
    json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    traffic = {}
    for k, v in pmc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            name = "k_base_logistic" if "k_base_logistic" in k else "k_smooth_xgb" if "k_smooth_xgb" in k else k
            traffic[name] = (2.0 * v["FETCH_SIZE"]["avg_per_launch"] + v["WRITE_SIZE"]["avg_per_launch"]) * 1024.0
    json.dump(traffic, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
    print(json.dumps(traffic))


if __name__ == "__main__":
    main(*sys.argv[1:4])
