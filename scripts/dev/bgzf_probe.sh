# usage (GPU box): bash scripts/dev/bgzf_probe.sh — read_vcf of the bench's BGZF query under chunk sizes / thread counts
python - <<'PY'
import os, sys, time, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gnomix_amd
from gnomix_amd import synth, vcfio, _lib
cfg = dict(synth.CHR22)
data = synth.synthetic_model(seed=0, n_rounds=2, **cfg)
N = 10000
X = synth.synthetic_X_device(N, data.C, "cuda:0", seed=1)
data.snp_pos = 1000 + 37 * np.arange(data.C); data.snp_ref = np.array(["A"] * data.C)
G = np.zeros((data.C, (N + 15) // 16 * 4), np.uint8)
ctx = _lib.default_context(0)
import ctypes
Xh = X.cpu().numpy()
path = "/dev/shm/q.vcf"
synth.write_vcf_gt2(path, vcfio.pack_gt2(Xh), N // 2, data.snp_pos, data.snp_ref, ["C"] * data.C)
subprocess.check_call("python - <<'Q'\nimport sys\nsys.path.insert(0,'.')\nfrom gnomix_amd import synth\nsynth.bgzip_file('/dev/shm/q.vcf','/dev/shm/q.vcf.gz')\nQ", shell=True) if hasattr(synth, "bgzip_file") else None
print("files", os.path.getsize(path) / 1e9, os.path.exists(path + ".gz"))
PY
