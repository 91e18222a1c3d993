# The command line's run_inference on chr22 x 5 000 samples: plain VCF / BGZF / gzip input, phase=False / True (file sizes, stage times)
import os, sys, time, gzip, struct, zlib, shutil, tempfile, subprocess, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth, cli, vcfio, HipGnomix, _lib
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
print("plain VCF %.2f GB" % (os.path.getsize(plain) / 1e9), flush=True)
def run(path, phase, tag):
    T = {}
    for rep in range(2):
        T = {}; t0 = time.perf_counter()
        cli.run_inference({"query_file": path, "chm": "22", "output_basename": work, "phase": phase}, gm, timings=T)
        T["total"] = time.perf_counter() - t0
    print(tag, "%.2f GB file" % (os.path.getsize(path) / 1e9), {k: round(v, 3) for k, v in T.items()}, "-> %.0f haplotypes/s" % (N / T["total"]), flush=True)
run(plain, False, "plain, phase=False:")
run(plain, True, "plain, phase=True :")
print("phased VCF out: %.2f GB" % (os.path.getsize(os.path.join(work, "query_file_phased.vcf")) / 1e9))
# BGZF (what bgzip writes) and plain gzip of the first 1/8 of the records (zlib level 1 to keep the preparation short)
txt = open(plain, "rb").read(os.path.getsize(plain) // 8)
txt = txt[:txt.rfind(b"\n") + 1]
def bgzf(data, block=65280):
    out = bytearray()
    for o in list(range(0, len(data), block)) + [None]:
        chunk = b"" if o is None else data[o:o + block]
        c = zlib.compressobj(1, zlib.DEFLATED, -15); payload = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(payload) + 8 - 1)
        out += payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)
t = time.time(); open(plain + ".bgz.gz", "wb").write(bgzf(txt)); print("bgzf prepared in %.1f s" % (time.time() - t), flush=True)
t = time.time(); open(plain + ".gz", "wb").write(gzip.compress(txt, 1)); print("gzip prepared in %.1f s" % (time.time() - t), flush=True)
for p, tag in ((plain + ".bgz.gz", "BGZF"), (plain + ".gz", "gzip")):
    for rep in range(2):
        t = time.time(); d = vcfio.read_vcf(p, chm="22", ctx=ctx); dt = time.time() - t
    i = d.info
    print("%s: %.2f GB -> %.2f GB of text, read_vcf %.3f s (load/inflate %.3f, parse %.3f) = %.2f GB/s of text, compression %d" %
          (tag, i.file_bytes / 1e9, i.text_bytes / 1e9, dt, i.seconds_load, i.seconds_parse, i.text_bytes / dt / 1e9, i.compression), flush=True)
shutil.rmtree(work, ignore_errors=True)
