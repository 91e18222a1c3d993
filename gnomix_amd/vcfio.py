"""Input adapter on the front of the hot path: phased VCF -> the (N, C) int8 matrix the model consumes.

Mirrors the reference's src/utils.py:55-159 (read_vcf via scikit-allel, snp_intersection, vcf_to_npy) and
src/utils.py:222-329 (update_vcf, npy_to_vcf, read_headers) without scikit-allel: a small pure-Python
parser for the fields this path touches.  vcf_to_npy is pinned against the reference's own function
(tests/golden/G7_vcf.npz)."""
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
            gi = fmt.index("GT") if "GT" in fmt else 0
            row = np.full((len(t) - 9, 2), -1, dtype=np.int8)
            for s, field in enumerate(t[9:]):
                g = field.split(":")[gi]
                sep = "|" if "|" in g else "/"
                al = g.split(sep)
                for h in range(min(2, len(al))):
                    if al[h] not in (".", ""):
                        row[s, h] = int(al[h])
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
            data = update_vcf(data, mask=keep)
    if verbose:
        n_var, n, _ = data["calldata/GT"].shape
        print("File read:", n_var, "SNPs for", n, "individuals")
    return data


def snp_intersection(pos1, pos2, verbose=False):
    """indices of the common positions in both arrays (src/utils.py:83-102)"""
    assert len(pos2) != 0, "No SNPs of specified chromosome found in query file."
    inter, idx1, idx2 = np.intersect1d(pos1, pos2, return_indices=True)
    if verbose:
        print("- Number of SNPs from model:", len(pos1))
        print("- Number of SNPs from file:", len(pos2))
        print("- Number of intersecting SNPs:", len(inter))
        print("- Percentage of model SNPs covered by query file: ", round(len(inter) / len(pos1), 4) * 100, "%", sep="")
    return idx1, idx2


def vcf_to_npy(vcf_data, snp_pos_fmt=None, snp_ref_fmt=None, miss_fill=2, return_idx=False, verbose=True):
    """(n_var, n_ind, 2) genotypes -> (2*n_ind, C_model) int8 in the model's SNP order: model SNPs absent from the
    query are `miss_fill`, REF mismatches are flipped 0<->1, anything not 0/1 becomes `miss_fill`
    (src/utils.py:104-159)."""
    data = vcf_data["calldata/GT"]
    n_var, n_ind, _ = data.shape
    data = data.reshape(n_var, n_ind * 2).T
    mat = data
    vcf_idx, fmt_idx = np.arange(n_ind * 2), np.arange(n_ind * 2)
    if snp_pos_fmt is not None:
        fmt_idx, vcf_idx = snp_intersection(snp_pos_fmt, vcf_data["variants/POS"], verbose=verbose)
        fill = np.full((n_ind * 2, len(snp_pos_fmt)), miss_fill)
        fill[:, fmt_idx] = data[:, vcf_idx]
        mat = fill
    if snp_ref_fmt is not None:
        swap = vcf_data["variants/REF"][vcf_idx] != np.asarray(snp_ref_fmt)[fmt_idx]
        if swap.any() and verbose:
            print("- Found ", int(swap.sum()), " (", round(float(np.mean(swap)) * 100, 4), "%) different reference variants. Adjusting...", sep="")
        fs = np.array(fmt_idx)[swap]
        mat[:, fs] = (mat[:, fs] - 1) * (-1)
    mat[np.logical_and(mat != 0, mat != 1)] = miss_fill
    mat = mat.astype(np.int8)
    if return_idx:
        return mat, vcf_idx, fmt_idx
    return mat


def update_vcf(vcf_data, mask=None, Updates=None):
    out = dict(vcf_data)
    if mask is not None:
        for k in vcf_data:
            if k != "samples":
                out[k] = vcf_data[k][mask]
    if Updates is not None:
        for k in Updates:
            if k != "samples":
                out[k] = Updates[k]
    return out


def read_headers(vcf_file):
    header = ""
    with _open(vcf_file) as f:
        for line in f:
            if line[0:2] == "##":
                header += line
    return header


def npy_to_vcf(reference, npy, results_file, headers=""):
    """phased haplotypes (2*n, n_var) + the query's variant metadata -> VCF text (src/utils.py:247-329)"""
    if not results_file.endswith(".vcf"):
        results_file += ".vcf"
    data = reference
    npy = np.asarray(npy).astype(int)
    n_var = data["calldata/GT"].shape[0]
    h, c = npy.shape
    assert n_var == c, "reference (" + str(n_var) + ") and numpy matrix (" + str(c) + ") not compatible"
    n = h // 2
    names = list(data["samples"]) if len(data.get("samples", [])) == n else ["sample%d" % i for i in range(n)]
    alt = data["variants/ALT"]
    alt0 = alt[:, 0] if np.ndim(alt) == 2 else alt
    with open(results_file, "w") as f:
        f.write(headers)
        f.write("##fileformat=VCFv4.1\n")
        f.write("##source=gnomix.py\n")
        f.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="Phased Genotype">\n')
        f.write("#" + "\t".join(["CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] + [str(s) for s in names]) + "\n")
        for v in range(n_var):
            q = data["variants/QUAL"][v]
            qs = "" if (isinstance(q, float) and np.isnan(q)) or (hasattr(q, "dtype") and np.isnan(q)) else str(q)
            row = [str(data["variants/CHROM"][v]), str(data["variants/POS"][v]), str(data["variants/ID"][v]),
                   str(data["variants/REF"][v]), str(alt0[v]), qs, "PASS", ".", "GT"]
            row += [str(npy[2 * i, v]) + "|" + str(npy[2 * i + 1, v]) for i in range(n)]
            f.write("\t".join(row) + "\n")
    return results_file
