"""Input adapter on the front of the hot path: phased VCF -> the genotypes the models consume, and the phased VCF back out.

Mirrors the reference's src/utils.py:55-159 (read_vcf via scikit-allel, snp_intersection, vcf_to_npy) and
src/utils.py:222-329 (update_vcf, npy_to_vcf, read_headers).  The text is parsed by the library (include/gnomix_io.h:
gnx_vcf_read, every host core, 2-bit genotypes in the order the text arrives); there is no Python parser in the product
(oracle/vcf_text.py is the test mirror).  Two ways to use the result:

* the reference's way — `vcf_to_npy(read_vcf(path), snp_pos, snp_ref)` builds the (N, C) int8 matrix with numpy exactly as
  src/utils.py:104-159 does (pinned against the reference's own function: tests/golden/G7_vcf.npz);
* the file path of the command line — `column_map(...)` reduces vcf_to_npy to ONE int32 per model SNP (source variant, REF
  flip, or "absent") and DeviceModel.infer_gt2 / phase_gt2 hand the parsed 2-bit rows and that map to the GPU, where the
  matrix is built in HBM (k_gt2.hip); the host never holds an (N, C) array.
"""
from __future__ import annotations

import ctypes as C
from collections.abc import Mapping

import numpy as np

from . import _lib

_FIELDS = {"variants/CHROM": 0, "variants/ID": 1, "variants/REF": 2, "variants/ALT": 3, "samples": 6}
_KEYS = ["calldata/GT", "variants/CHROM", "variants/POS", "variants/ID", "variants/REF", "variants/ALT", "variants/QUAL", "samples"]


class _Handle:
    """owner of one gnx_vcf*"""

    def __init__(self, lib, h):
        self.lib, self.h = lib, h

    def __del__(self):
        try:
            if self.h:
                self.lib.gnx_vcf_free(self.h)
                self.h = None
        except Exception:
            pass


class VcfData(Mapping):
    """What read_vcf returns: the scikit-allel dictionary of the reference (same keys, dtypes and conventions), materialised
    key by key on first use, over the parsed file held by the library.  `rows` (set by update_vcf(mask=...)) selects variants;
    `overrides` (update_vcf(Updates=...)) replaces whole columns."""

    def __init__(self, handle, info, rows=None, overrides=None, cache=None):
        self._hd, self.info = handle, info
        self.rows = rows
        self._over = dict(overrides or {})
        self._cache = cache if cache is not None else {}

    # ---- raw views for the file path -----------------------------------------------------------------------------------
    @property
    def n_variants(self):
        return int(self.info.n_variants) if self.rows is None else len(self.rows)

    @property
    def n_samples(self):
        return int(self.info.n_samples)

    @property
    def gt2(self):
        """(n_variants_in_file, ldg) uint8 view of the library's 2-bit genotype matrix (all variants: `rows` is not applied)"""
        lib, h = self._hd.lib, self._hd.h
        n = int(self.info.n_variants) * int(self.info.ldg)
        buf = (C.c_uint8 * max(n, 1)).from_address(lib.gnx_vcf_gt2(h))
        # the ctypes object is the base of the array and every view of it: hanging the handle on it keeps the library's matrix alive
        # for as long as any such view exists, also after this VcfData is dropped (`g = read_vcf(p).gt2` dangled)
        buf._gnx_handle = self._hd
        a = np.frombuffer(buf, dtype=np.uint8, count=n).reshape(int(self.info.n_variants), int(self.info.ldg))
        a.flags.writeable = False
        return a

    def _blob(self, field):
        lib, h = self._hd.lib, self._hd.h
        blob, off, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        _lib.io_check(lib.gnx_vcf_strings(h, field, C.byref(blob), C.byref(off), C.byref(n)))
        offs = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_int64)), shape=(n.value + 1,)).copy()
        data = C.string_at(blob, int(offs[-1])) if offs[-1] else b""
        return data, offs

    def strings(self, field):
        """object array of str for one string column of the file (all variants)"""
        data, offs = self._blob(field)
        s = data.decode("utf-8", errors="replace")
        if len(s) == len(data):  # ASCII: offsets are character offsets
            return np.array([s[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)], dtype=object)
        return np.array([data[offs[i]:offs[i + 1]].decode("utf-8", errors="replace") for i in range(len(offs) - 1)], dtype=object)

    def fixed_bytes(self, field):
        """one string column as a numpy 'S<w>' array, built without a Python loop (REF comparison of column_map)"""
        data, offs = self._blob(field)
        n = len(offs) - 1
        ln = np.diff(offs)
        w = int(ln.max()) if n else 1
        out = np.zeros((n, max(w, 1)), np.uint8)
        if n and len(data):
            b = np.frombuffer(data, np.uint8)
            k = np.arange(max(w, 1))
            m = k[None, :] < ln[:, None]
            out[m] = b[(offs[:-1, None] + k[None, :])[m]]
        return out.view("S%d" % max(w, 1)).reshape(n)

    @property
    def meta_header(self):
        """the '##' lines, as read_headers(vcf_file) returns them (src/utils.py:232-245)"""
        data, _ = self._blob(7)
        return data.decode("utf-8", errors="replace")

    # ---- the dictionary ----------------------------------------------------------------------------------------------------
    def _full(self, key):
        c = self._cache
        if key in c:
            return c[key]
        lib, h = self._hd.lib, self._hd.h
        nv, ns = int(self.info.n_variants), int(self.info.n_samples)
        if key == "calldata/GT":
            v = np.empty((nv, ns, 2), np.int8)
            _lib.io_check(lib.gnx_vcf_gt_int8(h, v.ctypes.data, 0))
        elif key == "variants/POS":
            v = np.ctypeslib.as_array(C.cast(lib.gnx_vcf_pos(h), C.POINTER(C.c_int64)), shape=(max(nv, 1),))[:nv].copy()
        elif key == "variants/QUAL":
            v = np.ctypeslib.as_array(C.cast(lib.gnx_vcf_qual(h), C.POINTER(C.c_float)), shape=(max(nv, 1),))[:nv].copy()
        elif key == "variants/ALT":
            v = np.stack([self.strings(3), self.strings(4), self.strings(5)], axis=1) if nv else np.empty((0, 3), object)
        elif key in _FIELDS:
            v = self.strings(_FIELDS[key])
        else:
            raise KeyError(key)
        c[key] = v
        return v

    def __getitem__(self, key):
        if key in self._over:
            return self._over[key]
        v = self._full(key)
        if self.rows is not None and key != "samples":
            return v[self.rows]
        return v

    def __iter__(self):
        return iter(_KEYS)

    def __len__(self):
        return len(_KEYS)

    def copy(self):
        return VcfData(self._hd, self.info, self.rows, self._over, self._cache)


def read_vcf(vcf_file, chm=None, fields=None, verbose=False, ctx=None, n_threads=0):
    """-> the scikit-allel style mapping of src/utils.py:55-81: calldata/GT (n_var, n_samples, 2) int8 (-1 = missing),
    variants/CHROM, POS, ID, REF, ALT (n_var, 3), QUAL, samples.  `chm` keeps the records of that CHROM; when the region
    holds no record the whole file is used instead, with the reference's message (src/utils.py:72-78).  `ctx` (a
    gnomix_amd.Context) makes the genotype matrix page-locked for the device path; parsing itself needs no GPU."""
    lib = _lib.load()
    h = C.c_void_p()
    region = None if chm is None else str(chm).encode()
    rc = lib.gnx_vcf_read(ctx.h if ctx is not None else None, str(vcf_file).encode(), region, int(n_threads), C.byref(h))
    if rc != _lib.GNX_OK:
        raise _lib.GnxError(rc, lib.gnx_io_last_error().decode(errors="replace"))
    hd = _Handle(lib, h)
    info = _lib.VcfInfo()
    _lib.io_check(lib.gnx_vcf_get_info(h, C.byref(info)))
    if info.n_variants == 0:
        print("No data found in vcf file {}".format(vcf_file))
        return None
    if info.region_fallback:
        print('Found no data in vcf file {} in region labeled "{}". Using all data from vcf instead...'.format(vcf_file, chm))
    data = VcfData(hd, info)
    if verbose:
        print("File read:", int(info.n_variants), "SNPs for", int(info.n_samples), "individuals")
    return data


def snp_intersection(pos1, pos2, verbose=False):
    """indices of the common positions in both arrays (src/utils.py:83-102)"""
    assert len(pos2) != 0, "No SNPs of specified chromosome found in query file."
    pos1, pos2 = np.asarray(pos1), np.asarray(pos2)
    if (pos1.ndim == pos2.ndim == 1 and pos1.dtype.kind in "iu" and pos2.dtype.kind in "iu" and len(pos1) > 0
            and bool(np.all(pos1[1:] > pos1[:-1])) and bool(np.all(pos2[1:] > pos2[:-1]))):
        # both strictly increasing (a model's SNPs and a sorted VCF without repeated positions): np.intersect1d's answer
        # without its two sorts — a binary search of one list in the other
        k = np.minimum(np.searchsorted(pos2, pos1), len(pos2) - 1)
        hit = pos2[k] == pos1
        idx1, idx2 = np.nonzero(hit)[0], k[hit]
        inter = pos1[idx1]
    else:
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


def column_map(vcf_data, snp_pos_fmt, snp_ref_fmt=None, verbose=True):
    """vcf_to_npy reduced to its column bookkeeping (src/utils.py:125-147): -> (src, vcf_idx, fmt_idx) where
    src (C,) int32 is what gnx_infer_gt2 / gnx_phase_gt2 take: src[c] = variant row | flip << 30, or -1 when model SNP c is
    absent from the query.  Same intersection (first occurrence of a repeated position), same REF rule, same messages."""
    pos = vcf_data["variants/POS"]
    fmt_idx, vcf_idx = snp_intersection(np.asarray(snp_pos_fmt), pos, verbose=verbose)
    rows = np.asarray(vcf_idx, np.int64)
    if isinstance(vcf_data, VcfData) and vcf_data.rows is not None:
        rows = np.asarray(vcf_data.rows)[rows]   # index into the file's variant rows
    src = np.full(len(snp_pos_fmt), -1, np.int32)
    val = rows.astype(np.int32)
    if snp_ref_fmt is not None:
        if isinstance(vcf_data, VcfData) and "variants/REF" not in vcf_data._over:
            qref = vcf_data.fixed_bytes(2)[rows]
            mref = np.asarray(snp_ref_fmt)[fmt_idx]
            if mref.dtype.kind == "U" and mref.dtype.itemsize:
                w = mref.dtype.itemsize // 4
                u = np.ascontiguousarray(mref).view(np.uint32).reshape(len(mref), w)      # UCS-4 code points
                mref = u.astype(np.uint8).view("S%d" % w).reshape(len(mref)) if (u < 128).all() else np.char.encode(mref, "utf-8")
            elif mref.dtype.kind != "S":
                mref = np.char.encode(mref.astype(str), "utf-8")
            swap = qref != mref
        else:
            swap = vcf_data["variants/REF"][vcf_idx] != np.asarray(snp_ref_fmt)[fmt_idx]
        if swap.any() and verbose:
            print("- Found ", int(swap.sum()), " (", round(float(np.mean(swap)) * 100, 4), "%) different reference variants. Adjusting...", sep="")
        val = val | (swap.astype(np.int32) << 30)
    src[fmt_idx] = val
    return src, vcf_idx, fmt_idx


def update_vcf(vcf_data, mask=None, Updates=None):
    """src/utils.py:222-237"""
    if isinstance(vcf_data, VcfData):
        out = vcf_data.copy()
        if mask is not None:
            base = np.arange(int(vcf_data.info.n_variants)) if vcf_data.rows is None else np.asarray(vcf_data.rows)
            out.rows = base[mask]
            out._over = {k: (v if k == "samples" else np.asarray(v)[mask]) for k, v in out._over.items()}
        if Updates is not None:
            for k in Updates:
                if k != "samples":
                    out._over[k] = Updates[k]
        return out
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
    """the '##' lines of a VCF (src/utils.py:332-348), read without parsing a record: plain text directly, gzip / BGZF through
    zlib's streaming reader, stopping at the first line that is not a meta line.  (The reference scans every line of the file; the
    format puts all meta lines before #CHROM, so the result is the same for any VCF — and a multi-GB query is not read for it.)"""
    import gzip
    with open(vcf_file, "rb") as f:
        magic = f.read(2)
    opener = gzip.open if magic == b"\x1f\x8b" else open
    out = []
    with opener(vcf_file, "rb") as f:
        for ln in f:
            if not ln.startswith(b"##"):
                break
            out.append(ln.decode("utf-8", errors="replace").replace("\r\n", "\n"))
    return "".join(out)


# ---- phased VCF out ------------------------------------------------------------------------------------------------------------
def _blob_of(strings):
    enc = [str(s).encode() for s in strings]
    off = np.zeros(len(enc) + 1, np.int64)
    np.cumsum([len(e) for e in enc], out=off[1:])
    return b"".join(enc), off


def _vcf_head(headers, names):
    head = headers + "##fileformat=VCFv4.1\n" + "##source=gnomix.py\n" + '##FORMAT=<ID=GT,Number=1,Type=String,Description="Phased Genotype">\n'
    head += "#" + "\t".join(["CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] + [str(s) for s in names]) + "\n"
    return head.encode()


def _names(data, n):
    return list(data["samples"]) if len(data.get("samples", [])) == n else ["sample%d" % i for i in range(n)]


def pack_gt2(npy):
    """(2n, V) haplotype-major integers -> (V, ldg) variant-major 2-bit rows (include/gnomix_io.h); host numpy, for callers
    that hold the matrix on the host (the file path gets these rows from the device: DeviceModel.phase_gt2)"""
    npy = np.asarray(npy)
    if npy.size and (npy.min() < 0 or npy.max() > 3):
        raise ValueError("pack_gt2: entries must be 0 .. 3 (two bits per call); got values in [%s, %s]" % (npy.min(), npy.max()))
    a = npy.T.astype(np.uint8)   # (V, 2n)
    V, N = a.shape
    ldg = (N + 15) // 16 * 4
    pad = np.zeros((V, ldg * 4), np.uint8)
    pad[:, :N] = a
    q = pad.reshape(V, ldg, 4)
    return (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8)


def write_phased_vcf(vcf_data, rows, G, results_file, ref=None, alt=None, headers="", n_threads=0):
    """the file path's npy_to_vcf: variant rows `rows` of the parsed query supply CHROM / POS / ID / QUAL, `ref` / `alt` the
    model's alleles (gnomix.py:62-66), G (len(rows), ldg) the re-phased 2-bit rows; text is produced by the library"""
    if not results_file.endswith(".vcf"):
        results_file += ".vcf"
    lib = _lib.load()
    rows = np.ascontiguousarray(rows, np.int64)
    G = np.ascontiguousarray(G, np.uint8)
    ns = vcf_data.n_samples
    head = _vcf_head(headers, _names(vcf_data, ns))
    rb, ro = _blob_of(ref) if ref is not None else (None, None)
    ab, ao = _blob_of(alt) if alt is not None else (None, None)
    rc = lib.gnx_write_phased_vcf(results_file.encode(), head, len(head), vcf_data._hd.h, rows.ctypes.data, len(rows),
                                  rb, ro.ctypes.data if ro is not None else None, ab, ao.ctypes.data if ao is not None else None,
                                  G.ctypes.data, G.shape[1], ns, int(n_threads))
    _lib.io_check(rc)
    return results_file


def npy_to_vcf(reference, npy, results_file, headers=""):
    """phased haplotypes (2*n, n_var) + the query's variant metadata -> VCF text (src/utils.py:247-329)"""
    if not results_file.endswith(".vcf"):
        results_file += ".vcf"
    data = reference
    npy = np.asarray(npy).astype(int)
    n_var = data["calldata/GT"].shape[0] if not isinstance(data, VcfData) else data.n_variants
    h, c = npy.shape
    assert n_var == c, "reference (" + str(n_var) + ") and numpy matrix (" + str(c) + ") not compatible"
    n = h // 2
    alt = data["variants/ALT"]
    alt0 = alt[:, 0] if np.ndim(alt) == 2 else alt
    qual = np.asarray(data["variants/QUAL"], np.float32)
    qs = _format_floats(qual)
    pre = []
    for v in range(n_var):
        pre.append("\t".join([str(data["variants/CHROM"][v]), str(data["variants/POS"][v]), str(data["variants/ID"][v]),
                              str(data["variants/REF"][v]), str(alt0[v]), "" if np.isnan(qual[v]) else qs[v], "PASS", ".", "GT"]))
    pb, po = _blob_of(pre)
    head = _vcf_head(headers, _names(data, n))
    G = pack_gt2(npy)
    lib = _lib.load()
    _lib.io_check(lib.gnx_write_vcf_gt2(results_file.encode(), head, len(head), pb, po.ctypes.data, G.ctypes.data, n_var,
                                        G.shape[1], n, 0, 0))
    return results_file


def _format_floats(a):
    """numpy's text of each float (the library's formatter: what the .fb writer prints)"""
    a = np.ascontiguousarray(a)
    assert a.dtype in (np.float32, np.float64)
    n = a.size
    out = C.create_string_buffer(32 * max(n, 1))
    off = np.zeros(n + 1, np.int64)
    _lib.io_check(_lib.load().gnx_format_floats(a.ctypes.data, int(a.dtype == np.float64), n, out, off.ctypes.data))
    raw = out.raw
    return [raw[off[i]:off[i + 1]].decode() for i in range(n)]
