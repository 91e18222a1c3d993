# LDS counters of the tree smoother on what it is for — an ensemble trained by gnx_train_gbt, inputs with ancestry tracts — for
# bench.py's roofline.trained_inputs.lds_conflict_frac (profiles/trained_inputs_latest.json).  Run on the GPU box:
#   python scripts/dev/trained_inputs_counters.py         (drives rocprofv3 --pmc over itself with GNX_TI_CHILD=1)
import os, sys, json, subprocess, glob, csv, shutil
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
if os.environ.get("GNX_TI_CHILD"):
    import numpy as np, torch
    import gnomix_amd
    from gnomix_amd import synth, train
    W, A, S, N = 370, 7, 75, 10000
    rng = np.random.RandomState(3)
    def noisy(Bc, sd):
        B = np.clip(Bc + rng.normal(0, sd, Bc.shape), 1e-4, None)
        return B / B.sum(-1, keepdims=True)
    Bt = synth.synthetic_phased_individuals(500, W, A, seed=5, phase_errors=0, noise=0.02)
    trees, _ = train.train_gbt_arrays(noisy(Bt, 0.45), np.argmax(Bt, -1).astype(np.int32), S)
    Bd = torch.from_numpy(noisy(synth.synthetic_phased_individuals(N // 2, W, A, seed=9, phase_errors=0, noise=0.02), 0.45).astype(np.float32)).cuda()
    d = synth.synthetic_model(C=W * 1000 + 500, M=1000, A=A, S=S, n_rounds=1, seed=1)
    for k, v in trees.items():
        setattr(d, k, v)
    m = gnomix_amd.DeviceModel(d)
    for _ in range(5):
        m.smooth_predict_device(Bd)
    torch.cuda.synchronize()
    sys.exit(0)
out = "/tmp/ti_pmc"
shutil.rmtree(out, ignore_errors=True)
subprocess.run(["rocprofv3", "--pmc", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "--output-format", "csv", "-d", out, "-o", "c", "--",
                sys.executable, os.path.abspath(__file__)], env=dict(os.environ, GNX_TI_CHILD="1", TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
agg = {}
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_smooth_xgb_rk" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
a = {k: sum(v) / len(v) for k, v in agg.items()}
from bench import kernel_src_sha16
res = {"source": "scripts/dev/trained_inputs_counters.py (rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT; k_smooth_xgb_rk on a gnx_train_gbt ensemble, tract inputs, 10 000 haplotypes x 370 windows)",
       "SQ_LDS_IDX_ACTIVE": a.get("SQ_LDS_IDX_ACTIVE"), "SQ_LDS_BANK_CONFLICT": a.get("SQ_LDS_BANK_CONFLICT"),
       "lds_conflict_frac_of_active": a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"] if a.get("SQ_LDS_IDX_ACTIVE") else None,
       "launches": len(agg.get("SQ_LDS_IDX_ACTIVE", [])), "kernel_src_sha16": kernel_src_sha16()}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/trained_inputs_latest.json", "w"), indent=1)
print(json.dumps(res))
shutil.rmtree(out, ignore_errors=True)
