"""Pure-Python restatement of the reference's VCF reading (src/utils.py:55-81: allel.read_vcf behind gzip.open), TEST
INFRASTRUCTURE: the checker for the native reader (gnomix_amd/csrc/gnx_io.cpp, gnomix_amd.vcfio.read_vcf).  scikit-allel is
absent from this image, so the keys / dtypes / missing conventions follow its documented output for the fields the
reference touches (calldata/GT int8 with -1 = missing, variants/ALT padded to 3 alternates, QUAL float32 NaN for '.');
`vcf_to_npy` downstream of it is pinned against the reference's own function (tests/golden/G7_vcf.npz).
Only tests/ may import this module."""
from __future__ import annotations

import gzip

import numpy as np


def _open(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "r")


def read_vcf(vcf_file, chm=None, fields=None, verbose=False):
    """-> dict with the scikit-allel keys used downstream: calldata/GT (n_var, n_samples, 2) int8 (-1 = missing),
    variants/CHROM, POS, ID, REF, ALT (n_var, 3), QUAL, samples.  `chm` filters on the CHROM column; when the
    region holds no record the whole file is used instead, as src/utils.py:72-78 does."""
    chrom, pos, vid, ref, alt, qual, gts = [], [], [], [], [], [], []
    samples = []
    with _open(vcf_file) as f:
        for line in f:
            if line.startswith("##"):
                continue
            if line.startswith("#"):
                samples = line.rstrip("\n").split("\t")[9:]
                continue
            t = line.rstrip("\n").split("\t")
            if len(t) < 10:
                continue
            chrom.append(t[0]); pos.append(int(t[1])); vid.append(t[2]); ref.append(t[3])
            a = t[4].split(",")
            alt.append((a + ["", "", ""])[:3])
            qual.append(np.nan if t[5] in (".", "") else float(t[5]))
            fmt = t[8].split(":")
            gi = fmt.index("GT") if "GT" in fmt else None   # no GT key: every call missing
            row = np.full((len(t) - 9, 2), -1, dtype=np.int8)
            for s, field in enumerate(t[9:]):
                parts = field.split(":")
                if gi is None or gi >= len(parts):
                    continue
                g = parts[gi]
                sep = "|" if "|" in g else "/"
                al = g.split(sep)
                for h in range(min(2, len(al))):
                    if al[h] not in (".", ""):
                        row[s, h] = min(int(al[h]), 127)
            gts.append(row)
    if not pos:
        print("No data found in vcf file {}".format(vcf_file))
        return None
    data = {"calldata/GT": np.stack(gts), "variants/CHROM": np.array(chrom, dtype=object), "variants/POS": np.array(pos),
            "variants/ID": np.array(vid, dtype=object), "variants/REF": np.array(ref, dtype=object),
            "variants/ALT": np.array(alt, dtype=object), "variants/QUAL": np.array(qual, dtype=np.float32),
            "samples": np.array(samples, dtype=object)}
    if chm is not None:
        keep = data["variants/CHROM"] == str(chm)
        if not keep.any():
            print('Found no data in vcf file {} in region labeled "{}". Using all data from vcf instead...'.format(vcf_file, chm))
        else:
            data = {k: (v if k == 'samples' else v[keep]) for k, v in data.items()}
    if verbose:
        n_var, n, _ = data["calldata/GT"].shape
        print("File read:", n_var, "SNPs for", n, "individuals")
    return data


