"""Writers on the output side of the hot path: window metadata, .msp, .fb (+ .lai, .bed).

Mirrors the reference's src/postprocess.py:25-210 byte for byte (pinned by tests/golden/G6_writers, produced by
running the reference's own get_meta_data / write_msp / write_fb).  The two big tables (.msp labels, .fb probabilities) are
formatted and written by the library (include/gnomix_io.h: gnx_write_msp / gnx_write_fb, all host cores); the window
metadata (W rows) is plain numpy.  No pandas needed."""
from __future__ import annotations

import os

import numpy as np

from . import _lib


def _fmt(v):
    """str() of a numpy scalar the way `np.array([...mixed...]).astype(str)` / pandas print it: shortest repr"""
    if isinstance(v, (np.floating, float)):
        return str(np.float64(v)) if not isinstance(v, np.float32) else str(v)
    return str(v)


def get_meta_data(chm, model_pos, query_pos, n_wind, wind_size, gen_map_pos, gen_map_cm):
    """Window table of the .msp/.fb files (postprocess.py:25-67): columns chm, spos, epos, sgpos, egpos, "n snps".
    Returns a dict of equal-length lists/arrays (a stand-in for the reference's DataFrame)."""
    model_pos = np.asarray(model_pos)
    query_pos = np.asarray(query_pos)
    C = len(model_pos)
    idx = np.arange(0, C, wind_size)
    spos_idx = idx[:-1]
    epos_idx = np.concatenate([idx[1:-1], np.array([C])]) - 1
    spos, epos = model_pos[spos_idx], model_pos[epos_idx]
    gpos = np.asarray(gen_map_pos, dtype=float)
    gcm = np.asarray(gen_map_cm, dtype=float)

    def interp(x):  # linear interpolation, ends clamped to the first/last cM (interp1d(fill_value=end_pts))
        x = np.asarray(x, dtype=float)
        k = np.clip(np.searchsorted(gpos, x, side="left"), 1, len(gpos) - 1)  # interp1d: x_new in (x[k-1], x[k]]
        lo, hi = k - 1, k
        slope = (gcm[hi] - gcm[lo]) / (gpos[hi] - gpos[lo])
        y = slope * (x - gpos[lo]) + gcm[lo]
        y = np.where(x < gpos[0], gcm[0], y)
        y = np.where(x > gpos[-1], gcm[-1], y)
        return y

    sgpos = np.round(interp(spos), 5)
    egpos = np.round(interp(epos), 5)
    n_snps = np.zeros_like(epos)
    if len(query_pos) < 2 or bool(np.all(query_pos[1:] >= query_pos[:-1])):
        # the reference's running pointer (postprocess.py:56-62) on sorted positions = counts between successive window ends
        # (the pointer never moves backwards: a window end below its predecessor adds nothing)
        cut = np.maximum.accumulate(np.searchsorted(query_pos, epos[:n_wind - 1], side="right")) if n_wind > 1 else np.zeros(0, int)
        n_snps[:n_wind - 1] = np.diff(np.concatenate([[0], cut]))
        q = int(cut[-1]) if n_wind > 1 else 0
    else:
        q = 0
        for w in range(n_wind - 1):
            while q < len(query_pos) and query_pos[q] <= epos[w]:
                n_snps[w] += 1
                q += 1
    n_snps[n_wind - 1] = len(query_pos) - q
    return {"chm": [chm] * n_wind, "spos": spos, "epos": epos, "sgpos": sgpos, "egpos": egpos, "n snps": n_snps}


META_COLUMNS = ["chm", "spos", "epos", "sgpos", "egpos", "n snps"]


def _meta_strings(meta):
    n = len(meta["spos"])
    return [[_fmt(meta[c][i]) for c in META_COLUMNS] for i in range(n)]


def _blob(rows):
    enc = [r.encode() for r in rows]
    off = np.zeros(len(enc) + 1, np.int64)
    if enc:
        np.cumsum([len(e) for e in enc], out=off[1:])
    return b"".join(enc), off


def write_msp(msp_prefix, meta_data, pred_labels, populations, query_samples, n_threads=0):
    """<prefix>.msp (postprocess.py:84-98): pred_labels (N, W) ints, haplotype columns sample.0 / sample.1.  The header and
    the W metadata prefixes are built here; the N x W label text is produced by the library on all host cores."""
    rows = ["\t".join(r) for r in _meta_strings(meta_data)]
    lab = np.ascontiguousarray(pred_labels, dtype=np.int32)
    head = "#Subpopulation order/codes: " + "\t".join([str(pop) + "=" + str(i) for i, pop in enumerate(populations)]) + "\n"
    head += "#" + "\t".join(META_COLUMNS) + "\t"
    head += "\t".join([str(s) for q in query_samples for s in (str(q) + ".0", str(q) + ".1")]) + "\n"
    head = head.encode()
    pb, po = _blob(rows)
    if lab.ndim != 2 or lab.shape[1] != len(rows):
        raise ValueError(f"pred_labels must be (N, W={len(rows)}), got {lab.shape}")
    _lib.io_check(_lib.load().gnx_write_msp((msp_prefix + ".msp").encode(), head, len(head), pb, po.ctypes.data, lab.ctypes.data,
                                            lab.shape[0], lab.shape[1], lab.shape[1], int(n_threads)))


def write_fb(fb_prefix, meta_data, proba, ancestry, query_samples, n_threads=0, ctx=None):
    """<prefix>.fb (postprocess.py:100-126): proba (N, W, A); every value is printed as pandas' to_csv prints a float column
    (numpy's shortest round-trip text of the array's dtype) — by the library, W rows in parallel on the host's cores, or, when a
    context is given and the probabilities are float32, as text produced on the GPU (gnx_write_fb_dev: same bytes, one write())."""
    proba = np.asarray(proba)
    if proba.dtype not in (np.float32, np.float64):
        proba = proba.astype(np.float64)
    proba = np.ascontiguousarray(proba)
    n_rows = len(meta_data["spos"])
    if proba.ndim != 3 or proba.shape[1] != n_rows:
        raise ValueError(f"proba must be (N, W={n_rows}, A), got {proba.shape}")
    se = np.stack([np.asarray(meta_data["spos"]).astype(int), np.asarray(meta_data["epos"]).astype(int)], axis=1)
    pp = np.round(np.mean(se, axis=1)).astype(int)
    gp = np.mean(np.stack([np.asarray(meta_data["sgpos"], dtype=float), np.asarray(meta_data["egpos"], dtype=float)], 1), axis=1)
    header = ["chromosome", "physical position", "genetic_position", "genetic_marker_index"]
    header += [":::".join([str(q), h, str(a)]) for q in query_samples for h in ["hap1", "hap2"] for a in ancestry]
    head = ("#reference_panel_population:\t" + "\t".join([str(a) for a in ancestry]) + "\n" + "\t".join(header) + "\n").encode()
    rows = ["\t".join([str(meta_data["chm"][r]), str(pp[r]), _fmt(np.float64(gp[r])), "."]) for r in range(n_rows)]
    pb, po = _blob(rows)
    # (measured, chr22 x 10 000 haplotypes into tmpfs: 0.078-0.082 s either way — one thread putting 305 MB into a FRESH file costs
    #  0.052-0.056 s in the kernel's page cache whoever produced the text: scripts/dev/tmpfs_write_probe.py.  The GPU route is what a
    #  caller asks for with ctx=; the command line keeps the host writer unless GNX_FB_DEV=1)
    if ctx is not None and proba.dtype == np.float32 and proba.size > 0:
        ctx.check(ctx.lib.gnx_write_fb_dev(ctx.h, (fb_prefix + ".fb").encode(), head, len(head), pb, po.ctypes.data, proba.ctypes.data,
                                           proba.shape[0], n_rows, proba.shape[2]))
        return
    _lib.io_check(_lib.load().gnx_write_fb((fb_prefix + ".fb").encode(), head, len(head), pb, po.ctypes.data, proba.ctypes.data,
                                           int(proba.dtype == np.float64), proba.shape[0], n_rows, proba.shape[2], int(n_threads)))


def msp_to_lai(msp_file, positions, lai_file=None):
    """SNP-level labels (postprocess.py:128-160): every window row repeated `n snps` times."""
    with open(msp_file) as f:
        first, second = f.readline(), f.readline()
        rows = [ln.rstrip("\n").split("\t") for ln in f if ln.strip()]
    samples = second[:-1].split("\t")[6:]
    n_reps = np.array([int(r[5]) for r in rows])
    assert n_reps.sum() == len(positions)
    data = np.array([[int(v) for v in r[6:]] for r in rows])
    snp = np.repeat(data, n_reps, axis=0)
    if lai_file is not None:
        with open(lai_file, "w") as f:
            f.write(first)
            f.write("position\t" + "\t".join(samples) + "\n")
            for p, r in zip(positions, snp):
                f.write(str(p) + "\t" + "\t".join(str(v) for v in r) + "\n")
    return samples, snp


def msp_to_bed(msp_file, root, pop_order=None):
    """one run-length encoded .bed per haplotype column (postprocess.py:162-210)"""
    with open(msp_file) as f:
        f.readline()
        header = f.readline().rstrip("\n").split("\t")
        rows = [ln.rstrip("\n").split("\t") for ln in f if ln.strip()]
    os.makedirs(root, exist_ok=True)
    lab = (lambda a: a) if pop_order is None else (lambda a: pop_order[int(a)])
    for ci, sample in enumerate(header[6:]):
        out = []
        start = 0
        for i in range(1, len(rows) + 1):
            if i == len(rows) or rows[i][6 + ci] != rows[start][6 + ci]:
                out.append((rows[start][0], rows[start][1], rows[i - 1][2], lab(rows[start][6 + ci]), rows[start][3], rows[i - 1][4]))
                start = i
        with open(os.path.join(root, sample.replace(".", "_") + ".bed"), "w") as f:
            f.write("chm\tspos\tepos\tancestry\tsgpos\tegpos\n")
            for r in out:
                f.write("\t".join(str(v) for v in r) + "\n")
