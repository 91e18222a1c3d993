#!/usr/bin/env python3
"""Register / scratch audit of libgnomix_hip.so: every gfx950 kernel's VGPRs, SGPRs, LDS and private (scratch) bytes, read
from the code objects' metadata (no GPU needed).

  python scripts/scratch_audit.py            table of all kernels with scratch, summary of the rest
  python scripts/scratch_audit.py --all      every kernel
  python scripts/scratch_audit.py --check    exit 1 when a kernel a DEFAULT dispatch reaches carries scratch

How: the library's .hip_fatbin section is a run of clang offload bundles; each bundle's gfx950 entry is an ELF whose
NT_AMDGPU_METADATA note lists, per kernel, .private_segment_fixed_size (bytes of scratch per lane: spills and dynamically
indexed private arrays), .vgpr_count, .sgpr_count, .group_segment_fixed_size.  tests/test_host_cpu.py runs --check.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "gnomix_amd", "libgnomix_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# Kernels the DEFAULT dispatch of a supported configuration reaches: no scratch allowed (a spilled kernel is a correctness-only
# kernel).  Patterns are matched against the demangled name.  Variants outside these (ablation / fallback templates that only
# an environment knob or an unusual model selects) are reported but do not fail the check.
MUST_BE_CLEAN = [
    r"k_base_logistic_i8<2, 1, 8, 2>", r"k_base_logistic_i8_dl<1, 2, 16, 2>", r"k_base_logistic<",
    r"k_smooth_xgb_rk<3, 8, 4, true, ", r"k_smooth_crf_ck<", r"k_smooth_crf_row16<", r"k_crf_psi<", r"k_smooth_cnn<",
    r"k_gnofix<", r"k_gnofix_ranks", r"k_gnofix_dif", r"k_gnofix_swap", r"k_gnofix_pmax", r"k_gnofix_count", r"k_gnofix_scan", r"k_gnofix_scatter", r"k_covrsk_dec_fast<\d+, 7>", r"k_covrsk_dec_fast<\d+, 0>", r"k_covrsk_dec<", r"k_svc_couple", r"k_calibrate",
    r"k_base_forest2<", r"k_base_forest<1, true, 8, 1>", r"k_base_forest<1, true, 16, 1>",
    r"k_unpack2<", r"k_gt2_to_x<", r"k_x_to_gt2", r"k_tr_forward<", r"k_tr_backward<", r"k_gbt_",
]


def code_objects(so=SO):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so, os.devnull])
        d = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        i = d.find(magic)
        n = 0
        while i >= 0:
            num = struct.unpack_from("<Q", d, i + 24)[0]
            o = i + 32
            for _ in range(num):
                off, size, tl = struct.unpack_from("<QQQ", d, o)
                o += 24
                triple = d[o:o + tl].decode()
                o += tl
                if "gfx950" in triple and size > 0:
                    p = os.path.join(td, "co_%d.elf" % n)
                    open(p, "wb").write(d[i + off:i + off + size])
                    n += 1
                    yield p
            i = d.find(magic, i + 1)


def kernels(so=SO):
    """-> list of dicts: name (demangled), vgpr, sgpr, agpr, lds, scratch"""
    out = []
    for p in code_objects(so):
        txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", p], text=True)
        # amdhsa.kernels is a YAML list: an item starts with "  - .<first key>:", its keys follow indented without the dash
        cur = None
        in_kernels = False
        for ln in txt.splitlines():
            if re.match(r"\s*amdhsa\.kernels:", ln):
                in_kernels = True
                continue
            if in_kernels and re.match(r"\s*amdhsa\.\w+:", ln):
                in_kernels = False
            if not in_kernels:
                continue
            item = re.match(r"^\s{2}-\s+\.(\w+):\s+(.*)$", ln)
            m = item or re.match(r"^\s{4}\.(\w+):\s+(.*)$", ln)
            if not m:
                continue
            if item:
                if cur and "name" in cur and "scratch" in cur:
                    out.append(cur)
                cur = {}
            k, v = m.group(1), m.group(2).strip()
            key = {"name": "name", "private_segment_fixed_size": "scratch", "vgpr_count": "vgpr", "sgpr_count": "sgpr", "agpr_count": "agpr",
                   "group_segment_fixed_size": "lds", "vgpr_spill_count": "vspill", "sgpr_spill_count": "sspill"}.get(k)
            if key and cur is not None:
                cur[key] = v if key == "name" else int(v)
        if cur and "name" in cur and "scratch" in cur:
            out.append(cur)
    names = "\n".join(k["name"] for k in out)
    dem = subprocess.run(["c++filt"], input=names, capture_output=True, text=True).stdout.splitlines()
    for k, d in zip(out, dem):
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        k["name"] = re.sub(r"^void ", "", re.sub(r"\(.*\)$", "", d))
    return out


def main(argv):
    ks = kernels()
    dirty = [k for k in ks if k["scratch"] > 0]
    must = [k for k in dirty if any(re.search(p, k["name"]) for p in MUST_BE_CLEAN)]
    show = ks if "--all" in argv else dirty
    print("%-84s %5s %5s %5s %7s %8s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "spills"))
    for k in sorted(show, key=lambda k: (-k["scratch"], k["name"])):
        flag = "  <-- default dispatch" if k in must else ""
        print("%-84s %5d %5d %5d %7d %8d %3d/%-3d%s" % (k["name"][:84], k.get("vgpr", 0), k.get("agpr", 0), k.get("sgpr", 0), k.get("lds", 0),
                                                     k["scratch"], k.get("vspill", 0), k.get("sspill", 0), flag))
    print("%d kernels, %d with scratch, %d of those on a default dispatch" % (len(ks), len(dirty), len(must)))
    if "--check" in argv and must:
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
