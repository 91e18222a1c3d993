# The worst case of Gnofix as a FILE: the bench's chr22 x 5 000 samples (unstructured haplotypes, uniform-random trees: a label change
# at almost every window, 50 sweeps per individual) through run_inference with phase=True (16.6 s, of it Gnofix 14.5 s)
import os, sys, time, shutil, tempfile, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth, cli, HipGnomix
ns = int(os.environ.get("NS", 5000)); N = 2 * ns
data = synth.synthetic_model(seed=0, n_rounds=100, **synth.CHR22)
C = data.C
rng = np.random.RandomState(7)
data.snp_pos = (16_050_000 + np.cumsum(rng.randint(1, 180, size=C))).astype(np.int64)
data.snp_ref = rng.choice(list("ACGT"), size=C); data.snp_alt = rng.choice(list("ACGT"), size=C)
data.gen_map_pos = np.array([16_000_000, 30_000_000, 52_000_000]); data.gen_map_cm = np.array([0.0, 31.5, 74.1])
gm = HipGnomix(data)
ctx = gm.dev.ctx
X = synth.synthetic_X_device(N, C, torch.device("cuda", 0), seed=94305)
ldg = (N + 15) // 16 * 4
cols = torch.arange(C, dtype=torch.int32, device="cuda"); Gd = torch.zeros((C, ldg), dtype=torch.uint8, device="cuda")
gm.dev._bind_torch_stream()
ctx.check(ctx.lib.gnx_x_to_gt2_dev(ctx.h, X.data_ptr(), N, X.stride(0), 0, cols.data_ptr(), C, Gd.data_ptr(), ldg)); torch.cuda.synchronize()
G = Gd.cpu().numpy(); del Gd, X
work = tempfile.mkdtemp(prefix="gnx_e2e_", dir="/dev/shm")
plain = os.path.join(work, "q.vcf")
synth.write_vcf_gt2(plain, G, ns, data.snp_pos, data.snp_ref, data.snp_alt, chrom="22")
for rep in range(2):
    T = {}; t0 = time.perf_counter()
    cli.run_inference({"query_file": plain, "chm": "22", "output_basename": work, "phase": True}, gm, timings=T)
    T["total"] = time.perf_counter() - t0
    print("phase=True, random chr22 x %d samples:" % ns, {k: round(v, 3) for k, v in T.items()}, flush=True)
shutil.rmtree(work, ignore_errors=True)
