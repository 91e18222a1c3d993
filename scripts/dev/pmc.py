#!/usr/bin/env python3
"""Counter passes over one command (development aid; run on the GPU box through gpurun):

  python scripts/dev/pmc.py 'k_smooth_crf|k_crf_psi' -- python scripts/bench_configs.py c5a

One rocprofv3 --pmc pass per counter set (counters only, never combined with trace domains); prints, per matching kernel, the
average per launch of every counter plus a few derived ratios.  GNX_PMC_SETS selects sets by name (default: sq,sq2,mem)."""
import collections
import csv
import glob
import os
import re
import shutil
import subprocess
import sys

SETS = {
    "sq": "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES",
    "sq2": "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU",
    "mem": "FETCH_SIZE",
    "memw": "WRITE_SIZE",
    "tcp": "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum",
}


def main():
    pat = re.compile(sys.argv[1])
    cmd = sys.argv[sys.argv.index("--") + 1:]
    which = os.environ.get("GNX_PMC_SETS", "sq,sq2,mem,memw").split(",")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for name in which:
        d = "/tmp/pmc_%s" % name
        shutil.rmtree(d, ignore_errors=True)
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            subprocess.run(["rocprofv3", "--pmc"] + SETS[name].split() + ["--output-format", "csv", "-d", d, "-o", "c", "--"] + cmd,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, timeout=600)
        except subprocess.TimeoutExpired:
            print("set %s: timeout" % name)
            continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if pat.search(r["Kernel_Name"]):
                    k = re.sub(r"\(Base.*|\(Smooth.*|\(Gnofix.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", ""))
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
    for k, v in agg.items():
        a = {c: sum(x) / len(x) for c, x in v.items()}
        n = max(len(x) for x in v.values())
        print("%s  (%d launches)" % (k[:100], n))
        print("   " + "  ".join("%s=%.4g" % (c, a[c]) for c in sorted(a)))
        der = {}
        if "SQ_BUSY_CYCLES" in a and a["SQ_BUSY_CYCLES"]:
            cu_cyc = a["SQ_BUSY_CYCLES"] / 8 * 256 / 32  # per-XCD busy cycles summed over 8 XCD x (32 CU) -> rough CU-cycles
            if "SQ_ACTIVE_INST_VALU" in a: der["valu_busy"] = a["SQ_ACTIVE_INST_VALU"] / (a["SQ_BUSY_CYCLES"] * 4)
            if "SQ_LDS_IDX_ACTIVE" in a: der["lds_busy"] = a["SQ_LDS_IDX_ACTIVE"] / a["SQ_BUSY_CYCLES"]
        if "SQ_WAVE_CYCLES" in a and a.get("SQ_WAIT_INST_ANY"): der["wait_inst_frac"] = a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"]
        if "SQ_WAVE_CYCLES" in a and a.get("SQ_WAIT_ANY"): der["wait_any_frac"] = a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"]
        # KB units; FETCH_SIZE doubled for wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section; scripts/summarize_prof.py)
        if "FETCH_SIZE" in a: der["fetch_GB(x2)"] = 2.0 * a["FETCH_SIZE"] * 1024 / 1e9
        if "WRITE_SIZE" in a: der["write_GB"] = a["WRITE_SIZE"] * 1024 / 1e9
        print("   derived: " + "  ".join("%s=%.3g" % kv for kv in der.items()))


if __name__ == "__main__":
    main()
