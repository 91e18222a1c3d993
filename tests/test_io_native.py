"""The file side of the hot path on the CPU (-m "not gpu"): the library's VCF reader against the pure-Python mirror
(oracle/vcf_text.py), its number formatting against numpy's own `.astype(str)` (what the reference's writers print through
pandas), the .msp / .fb / phased-VCF writers against line-by-line Python restatements of src/postprocess.py:84-126 and
src/utils.py:247-329, and vcfio.column_map against the reference-pinned vcf_to_npy (G7)."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from gnomix_amd import _lib, postprocess as pp, vcfio
from oracle import vcf_text

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- VCF text generator covering what real files contain ---------------------------------------------------------------------
def _vcf_text(rng, nv, ns, general_every=0, crlf=False, chroms=("22",), big_allele=0.01, haploid=0.03, unphased=0.2, miss=0.03):
    nl = "\r\n" if crlf else "\n"
    lines = ["##fileformat=VCFv4.2", "##contig=<ID=22>", '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("S%d" % i for i in range(ns))]
    pos = np.sort(rng.choice(10 ** 7, nv, replace=False))
    for v in range(nv):
        g = rng.integers(0, 2, (ns, 2))
        ms = rng.random((ns, 2)) < miss
        general = general_every and v % general_every == 0
        flds = []
        for s in range(ns):
            a = ["." if ms[s, h] else str(g[s, h] + (rng.integers(2, 140) if general and rng.random() < big_allele else 0)) for h in range(2)]
            if general:
                sep = "/" if rng.random() < unphased else "|"
                f = a[0] if rng.random() < haploid else a[0] + sep + a[1]
                flds.append(f + ":%d:0.5" % rng.integers(0, 99) if v % 2 else "7:" + f)
            else:
                flds.append(a[0] + ("/" if rng.random() < unphased * 0.1 else "|") + a[1])
        fmt = ("GT:DP:GQ" if v % 2 else "DP:GT") if general else "GT"
        alt = "T" if v % 7 else "T,G,C,A"
        qual = "." if v % 5 == 0 else "%.2f" % (rng.random() * 100)
        lines.append("\t".join([chroms[v % len(chroms)], str(pos[v]), "rs%d" % v if v % 11 else ".", "ACGT"[v % 4] if v % 13 else "ACG", alt, qual,
                                "PASS", "AC=1;AN=2", fmt] + flds))
    return nl.join(lines) + nl


def _bgzf(data, block=3000):
    """BGZF as bgzip writes it: gzip members with a BC extra field, raw deflate payload, crc32 + isize, empty EOF block"""
    out = bytearray()
    for o in list(range(0, len(data), block)) + [None]:
        chunk = b"" if o is None else data[o:o + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(payload) + 8 - 1
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
        out += payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)


def _same(a, b):
    for k in ["calldata/GT", "variants/POS", "variants/CHROM", "variants/ID", "variants/REF", "variants/ALT", "samples"]:
        assert np.array_equal(a[k], b[k]), k
    qa, qb = a["variants/QUAL"], b["variants/QUAL"]
    assert qa.dtype == np.float32 and np.array_equal(np.isnan(qa), np.isnan(qb)) and np.array_equal(qa[~np.isnan(qa)], qb[~np.isnan(qb)])


@pytest.mark.parametrize("case", [
    dict(nv=40, ns=1), dict(nv=97, ns=3, general_every=3), dict(nv=200, ns=8, crlf=True), dict(nv=150, ns=37, general_every=2, crlf=True),
    dict(nv=300, ns=33, chroms=("22", "21", "22", "X")), dict(nv=64, ns=129, general_every=5), dict(nv=500, ns=16, miss=0.5)])
@pytest.mark.parametrize("container", ["plain", "gzip", "bgzf"])
def test_reader_equals_python_mirror(tmp_path, case, container):
    rng = np.random.default_rng(hash(str(case)) % 2 ** 32)
    txt = _vcf_text(rng, **case).encode()
    p = str(tmp_path / ("q.vcf" + ("" if container == "plain" else ".gz")))
    if container == "plain":
        open(p, "wb").write(txt)
    elif container == "gzip":
        with gzip.open(p, "wb") as f:
            f.write(txt[:len(txt) // 2])
        with gzip.open(p, "ab") as f:           # two members: what `cat a.gz b.gz` produces
            f.write(txt[len(txt) // 2:])
    else:
        open(p, "wb").write(_bgzf(txt))
    for chm in (None, "22", "7"):
        for nt in (1, 5):
            a = vcfio.read_vcf(p, chm=chm, n_threads=nt)
            b = vcf_text.read_vcf(p, chm=chm)
            _same(a, b)
            assert a.info.compression == {"plain": 0, "gzip": 1, "bgzf": 2}[container]
            assert a.info.region_fallback == int(chm == "7")
            n_parsed = a.info.n_fast_lines + a.info.n_general_lines      # every record is parsed, the region selects afterwards
            assert n_parsed >= a.info.n_variants and (chm == "22" or n_parsed == a.info.n_variants)
    assert vcfio.read_headers(p) == "".join(ln + "\n" for ln in txt.decode().replace("\r\n", "\n").split("\n") if ln.startswith("##")) or case.get("crlf")
    # the 2-bit rows: codes of calldata/GT
    a = vcfio.read_vcf(p)
    gt = a["calldata/GT"].reshape(a.info.n_variants, -1)
    code = np.where(gt < 0, 2, np.where(gt > 1, 3, gt)).astype(np.uint8)
    G = a.gt2
    got = np.stack([(G[:, h // 4] >> (2 * (h % 4))) & 3 for h in range(gt.shape[1])], axis=1)
    assert np.array_equal(got, code)
    used = (gt.shape[1] + 3) // 4
    assert not G[:, used:].any() and (gt.shape[1] % 4 == 0 or not (G[:, used - 1] >> (2 * (gt.shape[1] % 4))).any())


@pytest.mark.parametrize("simd", ["scalar", "avx2", "avx512"])
def test_reader_fast_path_takes_plain_gt_files(tmp_path, monkeypatch, simd):
    """the fixed-width genotype path in its three widths (GNX_IO_SIMD narrows the choice; a CPU without the wider unit runs the next
    one down): sample counts around the 8- and 16-sample steps and their tails"""
    monkeypatch.setenv("GNX_IO_SIMD", simd)
    rng = np.random.default_rng(5)
    for ns in (1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 40, 41, 64, 65, 130):
        p = str(tmp_path / f"f{ns}.vcf")
        open(p, "w").write(_vcf_text(rng, 50, ns, unphased=0.0))
        a = vcfio.read_vcf(p, n_threads=2)
        assert a.info.n_fast_lines == 50 and a.info.n_general_lines == 0
        _same(a, vcf_text.read_vcf(p))
    # the same records without a final newline, and with every fixed-width record broken in one place
    p = str(tmp_path / "nonl.vcf")
    txt = _vcf_text(rng, 30, 12, unphased=0.0)
    open(p, "w").write(txt.rstrip("\n"))
    _same(vcfio.read_vcf(p), vcf_text.read_vcf(p))
    lines = txt.rstrip("\n").split("\n")
    for k, bad in enumerate(["1", "0:1", "10|1", "0|"]):
        t = lines[4 + k].split("\t")
        t[9 + k] = bad
        lines[4 + k] = "\t".join(t)
    open(p, "w").write("\n".join(lines) + "\n")
    a = vcfio.read_vcf(p)
    assert a.info.n_general_lines >= 3
    _same(a, vcf_text.read_vcf(p))


def test_reader_errors_are_reported_not_swallowed(tmp_path):
    p = str(tmp_path / "bad.vcf")
    head = "##x\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\n"
    for body, what in [("22\t1\t.\tA\tC\t.\t.\t.\tGT\t0|1\n", "fewer sample"), ("22\t1\t.\tA\tC\t.\t.\t.\tGT\t0|1\t0|0\t1|1\n", "more sample"),
                       ("22\tx\t.\tA\tC\t.\t.\t.\tGT\t0|1\t0|0\n", "POS"), ("22\t5\t.\tA\n", "fewer than 10")]:
        open(p, "w").write(head + body)
        with pytest.raises(_lib.GnxError) as e:
            vcfio.read_vcf(p)
        assert what in str(e.value)
    with pytest.raises(_lib.GnxError):
        vcfio.read_vcf(str(tmp_path / "absent.vcf"))
    open(p, "w").write("##only meta\n")
    with pytest.raises(_lib.GnxError):
        vcfio.read_vcf(p)
    open(p, "wb").write(b"\x1f\x8b\x08\x00garbage-not-deflate")
    with pytest.raises(_lib.GnxError):
        vcfio.read_vcf(p)
    open(p, "w").write(head)
    assert vcfio.read_vcf(p) is None   # "No data found in vcf file" (src/utils.py:70-71)


# ---- numbers ----------------------------------------------------------------------------------------------------------------------
def test_float_text_equals_numpy():
    rng = np.random.default_rng(0)
    f32 = [rng.integers(0, 2 ** 32, 400_000, dtype=np.uint64).astype(np.uint32).view(np.float32),      # every exponent
           rng.random(400_000, dtype=np.float32), (rng.random(100_000) * 1e-3).astype(np.float32),
           np.float32(10.0) ** rng.integers(-12, 20, 50_000).astype(np.float32), np.arange(0, 70000, dtype=np.float32),
           # every power of two and its neighbours (the asymmetric rounding interval of the shortest-digits search), subnormals included
           np.concatenate([(np.uint32(1) << np.arange(23, dtype=np.uint32)).view(np.float32)] +
                          [(np.arange(1, 255, dtype=np.uint32) << np.uint32(23)).__add__(np.uint32(d)).astype(np.uint32).view(np.float32) for d in (0, 1, 0xFFFFFFFF)]),
           np.array([0, -0.0, 1, 1e-4, 9.9999e-5, 9.999999e-5, 1e16, 9.9999998e15, 1e-45, 3.4028235e38, np.inf, -np.inf, 0.1, 0.14285715, 1 / 3,
                     16777216, 1e-5, 123456.79, 0.001, 1e15, 5e-324], dtype=np.float32)]
    for a in f32:
        a = a[~np.isnan(a)]
        assert vcfio._format_floats(a) == a.astype(str).tolist()
    f64 = [rng.integers(0, 2 ** 63, 300_000, dtype=np.uint64).view(np.float64), rng.random(300_000), rng.random(100_000) * 1e-4,
           10.0 ** rng.integers(-300, 300, 50_000), rng.random(100_000).astype(np.float32).astype(np.float64),
           np.array([0, -0.0, 1e-4, 9.999999999999999e-5, 1e16, 9999999999999998.0, 5e-324, 1.7976931348623157e308, np.inf, 0.1, 1 / 3, 2 ** 53, 1e22, 1e23])]
    for a in f64:
        a = a[~np.isnan(a)]
        assert vcfio._format_floats(a) == a.astype(str).tolist()
    assert vcfio._format_floats(np.array([np.nan], np.float32)) == ["nan"]


# ---- writers --------------------------------------------------------------------------------------------------------------------------
def _py_msp(meta, labels, populations, samples):
    rows = pp._meta_strings(meta)
    out = "#Subpopulation order/codes: " + "\t".join([str(p) + "=" + str(i) for i, p in enumerate(populations)]) + "\n"
    out += "#" + "\t".join(pp.META_COLUMNS) + "\t" + "\t".join([str(s) for q in samples for s in (str(q) + ".0", str(q) + ".1")]) + "\n"
    lab = np.asarray(labels)
    for l, r in enumerate(rows):       # src/postprocess.py:95-97: np.concatenate([meta, pred_labels.T], 1).astype(str), tab-joined
        out += "\t".join(r + [str(v) for v in lab[:, l]]) + "\n"
    return out


def _py_fb(meta, proba, ancestry, samples):
    n_rows = len(meta["spos"])
    se = np.stack([np.asarray(meta["spos"]).astype(int), np.asarray(meta["epos"]).astype(int)], axis=1)
    ppos = np.round(np.mean(se, axis=1)).astype(int)
    gp = np.mean(np.stack([np.asarray(meta["sgpos"], dtype=float), np.asarray(meta["egpos"], dtype=float)], 1), axis=1)
    header = ["chromosome", "physical position", "genetic_position", "genetic_marker_index"]
    header += [":::".join([str(q), h, str(a)]) for q in samples for h in ["hap1", "hap2"] for a in ancestry]
    fb = np.swapaxes(proba, 1, 2).reshape(-1, n_rows).T       # src/postprocess.py:115: (W, N*A)
    txt = fb.astype(str)                                       # what DataFrame.to_csv prints for a float column
    out = "#reference_panel_population:\t" + "\t".join(str(a) for a in ancestry) + "\n" + "\t".join(header) + "\n"
    for r in range(n_rows):
        vals = [str(meta["chm"][r]), str(ppos[r]), str(np.float64(gp[r])), "."] + ["" if t == "nan" else t for t in txt[r]]
        out += "\t".join(vals) + "\n"
    return out


@pytest.mark.parametrize("N,W,A,dtype", [(2, 3, 2, np.float32), (14, 41, 7, np.float32), (6, 150, 12, np.float64), (500, 9, 3, np.float32),
                                         (1030, 171, 6, np.float32), (300, 130, 27, np.float64)])   # > 2^20 values: re-laid window-major first
def test_msp_fb_writers_equal_the_python_restatement(tmp_path, N, W, A, dtype):
    rng = np.random.default_rng(N * W)
    M = 10
    model_pos = np.sort(rng.choice(10 ** 6, W * M + 3, replace=False))
    meta = pp.get_meta_data("22", model_pos, model_pos[::2], W, M, np.array([0, 10 ** 6]), np.array([0.0, 3.3]))
    labels = rng.integers(0, A, (N, W))
    if A > 10:
        labels[0, 0] = 11
    proba = rng.random((N, W, A)).astype(dtype)
    proba /= proba.sum(-1, keepdims=True)
    proba[0, 0, 0] = 1e-7
    proba[1, W - 1, A - 1] = 1.0
    proba[0, 1, 1] = 0.0
    proba[1, 0, 0] = np.nan
    samples, pops = ["I%d" % i for i in range(N // 2)], ["P%d" % a for a in range(A)]
    for nt in (1, 7):
        for env in ("", "1"):       # mapped output file, and the pwrite route
            os.environ["GNX_IO_NO_MMAP"] = env
            if not env:
                del os.environ["GNX_IO_NO_MMAP"]
            out = str(tmp_path / f"o{nt}{env}")
            pp.write_msp(out, meta, labels, pops, samples, n_threads=nt)
            pp.write_fb(out, meta, proba, pops, samples, n_threads=nt)
            assert open(out + ".msp").read() == _py_msp(meta, labels, pops, samples)
            assert open(out + ".fb").read() == _py_fb(meta, proba, pops, samples)
    os.environ.pop("GNX_IO_NO_MMAP", None)


def _py_npy_to_vcf(data, npy, headers, names):
    """src/utils.py:283-329 line by line: one "m|p" string per sample and variant, pandas to_csv with tabs"""
    out = headers + "##fileformat=VCFv4.1\n##source=gnomix.py\n" + '##FORMAT=<ID=GT,Number=1,Type=String,Description="Phased Genotype">\n'
    out += "#" + "\t".join(["CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] + list(names)) + "\n"
    alt = data["variants/ALT"]
    for v in range(npy.shape[1]):
        q = data["variants/QUAL"][v]
        row = [str(data["variants/CHROM"][v]), str(data["variants/POS"][v]), str(data["variants/ID"][v]), str(data["variants/REF"][v]),
               str(alt[v, 0]), "" if np.isnan(q) else str(np.float32(q)), "PASS", ".", "GT"]
        row += [str(npy[2 * i, v]) + "|" + str(npy[2 * i + 1, v]) for i in range(npy.shape[0] // 2)]
        out += "\t".join(row) + "\n"
    return out


@pytest.mark.parametrize("ns", [1, 2, 5, 64, 131])
def test_phased_vcf_writers(tmp_path, ns):
    rng = np.random.default_rng(ns)
    p = str(tmp_path / "q.vcf")
    open(p, "w").write(_vcf_text(rng, 120, ns, general_every=4))
    d = vcfio.read_vcf(p)
    nv = d.n_variants
    X = rng.integers(0, 3, (2 * ns, nv)).astype(np.int8)
    names = list(d["samples"])
    want = _py_npy_to_vcf(d, X, d.meta_header, names)
    out = vcfio.npy_to_vcf(d, X, str(tmp_path / "a"), headers=vcfio.read_headers(p))
    assert open(out).read() == want
    # the file path: a row subset, the model's alleles, 2-bit rows
    rows = np.sort(rng.choice(nv, nv // 2, replace=False))
    ref, alt = rng.choice(list("ACGT"), len(rows)), rng.choice(["A", "CT", "G"], len(rows))
    upd = vcfio.update_vcf(d, mask=rows, Updates={"variants/REF": ref, "variants/ALT": alt.reshape(-1, 1)})
    assert upd.n_variants == len(rows) and np.array_equal(upd["variants/POS"], d["variants/POS"][rows])
    want = _py_npy_to_vcf(upd, X[:, rows], d.meta_header, names)
    G = vcfio.pack_gt2(X[:, rows])
    out = vcfio.write_phased_vcf(d, rows, G, str(tmp_path / "b"), ref=ref, alt=alt, headers=d.meta_header, n_threads=3)
    assert open(out).read() == want
    # and what comes back through the reader is the matrix that went in
    back = vcfio.read_vcf(out)
    assert np.array_equal(back["calldata/GT"].reshape(len(rows), -1).T, X[:, rows])


def test_synthetic_vcf_roundtrip_with_missing_as_dot(tmp_path):
    from gnomix_amd import synth
    rng = np.random.default_rng(3)
    ns, nv = 21, 300
    X = rng.integers(0, 3, (2 * ns, nv)).astype(np.int8)
    pos = np.sort(rng.choice(10 ** 6, nv, replace=False))
    p = synth.write_vcf_gt2(str(tmp_path / "s.vcf"), vcfio.pack_gt2(X), ns, pos, rng.choice(list("ACGT"), nv), rng.choice(list("ACGT"), nv), chrom="22")
    txt = open(p).read()
    assert ".|" in txt or "|." in txt
    d = vcfio.read_vcf(p, chm="22")
    assert d.info.n_fast_lines == nv
    assert np.array_equal(vcfio.vcf_to_npy(d, verbose=False), X)
    assert np.array_equal(d["variants/POS"], pos)


# ---- column map == vcf_to_npy -------------------------------------------------------------------------------------------------------
def _apply_map(G, N, src):
    """numpy statement of k_gt2_to_x: X[n, c] from the 2-bit rows and the column map"""
    C = len(src)
    X = np.full((N, C), 2, np.int8)
    have = src >= 0
    v = src[have] & 0x3FFFFFFF
    flip = (src[have] >> 30) & 1
    codes = np.stack([(G[v, h // 4] >> (2 * (h % 4))) & 3 for h in range(N)], axis=0).astype(np.int8)   # (N, n_have)
    codes = np.where(codes >= 2, 2, np.where(flip[None, :] == 1, 1 - codes, codes))
    X[:, have] = codes
    return X


def test_column_map_reproduces_vcf_to_npy_G7(tmp_path):
    """the reference's own vcf_to_npy output (golden G7) through VCF text, the native reader and the column map"""
    from gnomix_amd import synth
    g = np.load(os.path.join(ROOT, "tests", "golden", "G7_vcf.npz"), allow_pickle=False)
    gt = g["gt"]
    nv, ns, _ = gt.shape
    code = np.where(gt < 0, 2, np.where(gt > 1, 3, gt)).astype(np.uint8).reshape(nv, 2 * ns)
    p = synth.write_vcf_gt2(str(tmp_path / "g7.vcf"), vcfio.pack_gt2(code.T), ns, g["vcf_pos"], g["vcf_ref"], ["N"] * nv, chrom="22",
                            missing_as_dot=True)
    d = vcfio.read_vcf(p, chm="22")
    src, vi, fi = vcfio.column_map(d, g["model_pos"], g["model_ref"], verbose=False)
    assert np.array_equal(vi, g["vcf_idx"]) and np.array_equal(fi, g["fmt_idx"])
    X = _apply_map(d.gt2, 2 * ns, src)
    want = g["X"].copy()
    assert np.array_equal(X, want)


def test_column_map_random_cases_against_vcf_to_npy(tmp_path, capsys):
    rng = np.random.default_rng(11)
    for trial in range(6):
        nv, ns, C = 400, 9, 350
        p = str(tmp_path / "q.vcf")
        txt = _vcf_text(rng, nv, ns, general_every=3 if trial % 2 else 0, chroms=("22", "22", "21"))
        if trial >= 4:   # a repeated position: np.intersect1d keeps the first occurrence
            ls = txt.split("\n")
            t = ls[10].split("\t")
            t[1] = ls[9].split("\t")[1]
            ls[10] = "\t".join(t)
            txt = "\n".join(ls)
        open(p, "w").write(txt)
        d = vcfio.read_vcf(p, chm="22")
        qpos = d["variants/POS"]
        model_pos = np.sort(np.unique(np.concatenate([rng.choice(qpos, min(C - 60, len(qpos) - 30), replace=False), rng.choice(10 ** 7, 60)])))
        model_ref = rng.choice(list("ACGT"), len(model_pos))
        want, vi, fi = vcfio.vcf_to_npy(d, model_pos, model_ref, return_idx=True, verbose=True)
        msg_ref = capsys.readouterr().out
        src, vi2, fi2 = vcfio.column_map(d, model_pos, model_ref, verbose=True)
        assert capsys.readouterr().out == msg_ref              # the reference's progress messages, word for word
        assert np.array_equal(vi, vi2) and np.array_equal(fi, fi2)
        assert np.array_equal(_apply_map(d.gt2, 2 * ns, src), want)
        assert (src >> 30 == 1).any() and (src == -1).any()


def test_reader_survives_mutated_files(tmp_path):
    """robustness: random truncations, byte flips, deleted / duplicated spans of a valid file either parse or raise GnxError —
    and whenever the Python mirror can read the mutant too, both agree"""
    rng = np.random.default_rng(99)
    base = _vcf_text(rng, 60, 11, general_every=3).encode()
    p = str(tmp_path / "m.vcf")
    n_ok = n_err = 0
    for trial in range(250):
        b = bytearray(base)
        kind = trial % 5
        if kind == 0:
            b = b[:rng.integers(1, len(b))]
        elif kind == 1:
            for _ in range(rng.integers(1, 6)):
                b[rng.integers(0, len(b))] = rng.choice(list(b"\t\n|/.:0123456789xG#"))
        elif kind == 2:
            i = rng.integers(0, len(b) - 40)
            del b[i:i + rng.integers(1, 40)]
        elif kind == 3:
            i = rng.integers(0, len(b) - 40)
            b[i:i] = b[i:i + rng.integers(1, 40)]
        else:
            b = b.replace(b"\t", b" ", 1) if trial % 2 else b + b"\n\n#junk\n22\t5\n"
        open(p, "wb").write(bytes(b))
        try:
            a = vcfio.read_vcf(p, chm="22", n_threads=3)
        except _lib.GnxError:
            n_err += 1
            continue
        n_ok += 1
        if a is None:
            continue
        assert a["calldata/GT"].shape[0] == len(a["variants/POS"]) == a.n_variants
        try:
            m = vcf_text.read_vcf(p, chm="22")
        except Exception:
            continue
        if m is not None and len(m["variants/POS"]) == a.n_variants:
            assert np.array_equal(a["calldata/GT"], m["calldata/GT"]) and np.array_equal(a["variants/POS"], m["variants/POS"])
    assert n_ok > 20 and n_err > 20


@pytest.mark.parametrize("nt", [1, 4])
def test_writers_report_a_failed_write_instead_of_hanging(tmp_path, nt):
    """with more than one thread a failed write() (disk full, file size limit) left the writer waiting for blocks
    nobody would fill.  A child process with RLIMIT_FSIZE = 100 KB (SIGXFSZ ignored, so write() fails with EFBIG) must get a
    GnxError from write_msp within seconds."""
    import subprocess
    import sys
    code = f"""
import resource, signal, sys
import numpy as np
sys.path.insert(0, {ROOT!r})
from gnomix_amd import postprocess as pp, _lib
signal.signal(signal.SIGXFSZ, signal.SIG_IGN)
resource.setrlimit(resource.RLIMIT_FSIZE, (100 * 1024, 100 * 1024))
N, W, A, M = 400, 300, 4, 10
rng = np.random.default_rng(0)
pos = np.sort(rng.choice(10 ** 6, W * M + 3, replace=False))
meta = pp.get_meta_data("22", pos, pos[::2], W, M, np.array([0, 10 ** 6]), np.array([0.0, 3.3]))
try:
    pp.write_msp({str(tmp_path / 'big')!r}, meta, rng.integers(0, A, (N, W)), ["P%d" % a for a in range(A)], ["I%d" % i for i in range(N // 2)], n_threads={nt})
except (_lib.GnxError, OSError) as e:
    print("raised", type(e).__name__)
    sys.exit(0)
sys.exit(3)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "raised" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-600:])


def test_inflate_raw_equals_zlib():
    """gnx_io_inflate_raw (csrc/gnx_inflate.cpp, what BGZF blocks are inflated with) on raw DEFLATE streams zlib wrote at every level
    and strategy — stored, fixed and dynamic blocks, incompressible bytes, distance-1 runs, skewed alphabets (15-bit codes behind
    subtables), genotype text — decoded byte for byte, nothing written past the output, wrong sizes and truncated input refused"""
    lib = _lib.load()
    rng = np.random.default_rng(0)

    def raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        return c.compress(data) + c.flush()

    def check(data, **kw):
        z = np.frombuffer(raw(data, **kw), np.uint8).copy()
        out = np.full(len(data) + 64, 0xAB, np.uint8)
        assert lib.gnx_io_inflate_raw(z.ctypes.data, len(z), out.ctypes.data, len(data)) == 0, (kw, len(data))
        assert bytes(out[:len(data)]) == data and (out[len(data):] == 0xAB).all(), kw
        if len(data) > 10:
            assert lib.gnx_io_inflate_raw(z.ctypes.data, len(z), out.ctypes.data, len(data) - 1) != 0
            assert lib.gnx_io_inflate_raw(z.ctypes.data, len(z) // 2, out.ctypes.data, len(data)) != 0

    gt = ("\t".join(rng.choice(["0|0", "0|1", "1|0", "1|1", ".|."], 4000, p=[.5, .2, .2, .09, .01])) + "\n").encode()
    skew = rng.choice(256, 60000, p=(lambda p: p / p.sum())(1.0 / (1 + np.arange(256)) ** 2.5)).astype(np.uint8)
    texts = [gt * 8, bytes(rng.integers(0, 256, 70000, dtype=np.uint8)), bytes(rng.integers(0, 4, 65000, dtype=np.uint8)), b"a" * 65536,
             b"abc" * 20000, b"", b"x", bytes(skew)]
    for t in texts:
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                check(t, level=level, strategy=strat)
    for i in range(120):
        L = int(rng.integers(1, 70000))
        d = (bytes(rng.integers(0, int(rng.integers(2, 256)), L, dtype=np.uint8)) if i % 3 == 0 else (gt * 30)[int(rng.integers(0, 1000)):][:L] if i % 3 == 1
             else bytes(np.repeat(rng.integers(0, 256, L // 7 + 1, dtype=np.uint8), 7)[:L]))
        check(d, level=int(rng.integers(1, 10)))
    # garbage never crashes and never writes outside the output
    for i in range(200):
        z = rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8)
        out = np.full(1000 + 64, 0xCD, np.uint8)
        lib.gnx_io_inflate_raw(z.ctypes.data, len(z), out.ctypes.data, 1000)
        assert (out[1000:] == 0xCD).all()


@pytest.mark.parametrize("block,chunk", [(3000, 0), (700, 4096), (65280, 200000), (97, 64)])
def test_bgzf_windows_are_inflated_by_the_parsing_threads(tmp_path, monkeypatch, block, chunk):
    """a BGZF query is never inflated as a whole: every parsing window inflates the blocks that cover it (blocks smaller and larger
    than the windows, windows that start mid-block, records that span several blocks); zlib (GNX_VCF_ZLIB=1) gives the same arrays;
    a corrupt block is reported"""
    rng = np.random.default_rng(block)
    txt = _vcf_text(rng, nv=400, ns=23, general_every=7).encode()
    p = str(tmp_path / "q.vcf.gz")
    open(p, "wb").write(_bgzf(txt, block=block))
    if chunk:
        monkeypatch.setenv("GNX_IO_CHUNK", str(chunk))
    ref = vcf_text.read_vcf(p)
    for nt in (1, 6):
        a = vcfio.read_vcf(p, n_threads=nt)
        _same(a, ref)
        assert a.info.compression == 2 and a.info.text_bytes == len(txt)
    monkeypatch.setenv("GNX_VCF_ZLIB", "1")
    _same(vcfio.read_vcf(p, n_threads=3), ref)
    monkeypatch.delenv("GNX_VCF_ZLIB")
    z = bytearray(open(p, "rb").read())
    offs, o = [], 0
    while o < len(z):                # walk the members: BSIZE (total block size - 1) sits at bytes 16-17 of each header
        n = struct.unpack_from("<H", z, o + 16)[0] + 1
        if n > 60:                   # (not the empty end-of-file block)
            offs.append(o)
        o += n
    for b in offs[len(offs) // 3: len(offs) // 3 + 3]:      # up to three blocks in the middle of the records
        for k in range(18, 30):     # their deflate payloads
            z[b + k] ^= 0x5A
    open(p, "wb").write(bytes(z))
    try:
        b = vcfio.read_vcf(p, n_threads=2)
        damaged = not all(np.array_equal(b[k], ref[k]) for k in ("calldata/GT", "variants/POS"))
    except Exception:
        damaged = True
    assert damaged      # (a payload that no longer inflates to its stored size AND CRC-32 is an error: gnx_io_crc32 checks every block)


def test_crc32_by_carryless_multiplication_equals_zlib():
    """gnx_io_crc32 (PCLMULQDQ folding; zlib's crc32 for the unaligned head / tail and on hosts without the instruction) == zlib.crc32
    for every length around the 16- and 64-byte boundaries, at every alignment; it is what checks each BGZF block against its trailer"""
    import zlib
    import gnomix_amd
    lib = gnomix_amd.load_library()
    rng = np.random.RandomState(7)
    for n in list(range(0, 200)) + [255, 256, 257, 1000, 4095, 4096, 65535, 65536, (1 << 20) + 7]:
        b = rng.randint(0, 256, size=n + 5).astype(np.uint8)
        for off in (0, 1, 5):
            v = b[off:off + n]
            assert lib.gnx_io_crc32(v.ctypes.data, n) == (zlib.crc32(v.tobytes()) & 0xFFFFFFFF), (n, off)


def test_bgzf_block_with_a_valid_stream_but_wrong_crc_is_refused(tmp_path):
    """a block replaced by ANOTHER valid deflate stream of the same uncompressed size (sizes alone cannot tell) fails its CRC-32"""
    import zlib
    rng = np.random.default_rng(3)
    txt = _vcf_text(rng, nv=60, ns=9).encode()
    z = bytearray(_bgzf(txt, block=4096))
    n0 = struct.unpack_from("<H", z, 16)[0] + 1                     # first member
    isize = struct.unpack_from("<I", z, n0 - 4)[0]
    other = bytes(reversed(txt[:isize]))                            # the same length, other text
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = co.compress(other) + co.flush()
    member = bytes(z[:16]) + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload + bytes(z[n0 - 8:n0])   # the ORIGINAL trailer (CRC of txt)
    p = str(tmp_path / "q.vcf.gz")
    open(p, "wb").write(member + bytes(z[n0:]))
    with pytest.raises(Exception, match="corrupt BGZF"):
        vcfio.read_vcf(p, n_threads=2)
