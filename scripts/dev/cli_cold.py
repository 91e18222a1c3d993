# cold command-line run on a chr22 x NS-sample synthetic VCF in /dev/shm: where does the wall time go?
import os, sys, time, subprocess, shutil, numpy as np
sys.path.insert(0, os.getcwd())
import torch, gnomix_amd
from gnomix_amd import synth, vcfio
NS = int(os.environ.get("NS", 5000))
work = "/dev/shm/cli_cold"; shutil.rmtree(work, ignore_errors=True); os.makedirs(work)
data = synth.synthetic_model(seed=0, n_rounds=100, **synth.CHR22)
C = data.C; N = 2 * NS
rng = np.random.RandomState(7)
data.snp_pos = (16_050_000 + np.cumsum(rng.randint(1, 180, size=C))).astype(np.int64)
data.snp_ref = rng.choice(list("ACGT"), size=C); data.snp_alt = rng.choice(list("ACGT"), size=C)
data.gen_map_pos = np.array([16_000_000, 30_000_000, 52_000_000]); data.gen_map_cm = np.array([0.0, 31.5, 74.1])
X = synth.synthetic_X(N, C, seed=5, miss=0.0)
G = vcfio.pack_gt2(X)
vcf = os.path.join(work, "q.vcf")
t0 = time.perf_counter(); synth.write_vcf_gt2(vcf, G, NS, data.snp_pos, data.snp_ref, data.snp_alt, chrom="22"); print("vcf written %.2f s, %.2f GB" % (time.perf_counter() - t0, os.path.getsize(vcf) / 1e9), flush=True)
mp = os.path.join(work, "m.gnx"); data.save(mp)
env = dict(os.environ, GNX_CLI_TIMING="1"); env.pop("GNX_NO_TORCH", None)
for rep in range(3):
    t0 = time.perf_counter()
    pr = subprocess.run([sys.executable, "gnomix.py", vcf, os.path.join(work, "out"), "22", "False", mp], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, text=True)
    dt = time.perf_counter() - t0
    print("wall %.3f s  rc %d  %s" % (dt, pr.returncode, [l for l in pr.stderr.splitlines() if "timings" in l][-1:]), flush=True)
shutil.rmtree(work, ignore_errors=True)
