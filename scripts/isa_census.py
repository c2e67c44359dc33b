#!/usr/bin/env python3
"""Instruction census of a line range of a gfx950 .s file (hipcc -save-temps): how many wave-instructions of each class a
phase of a kernel issues, and the vector-issue cycles they hold (MI355X_MICROARCH.md 'vector-instruction ISSUE cost':
plain VALU 4, transcendental 8, an MFMA holds the port 8 of its 16 / 32 cycles).

usage: isa_census.py file.s --kernel <mangled-substring> [--range name:first:last ...]   (1-based lines inside the kernel)
       isa_census.py file.s --kernel <substr> --labels           (print labels / branches with line numbers)
"""
import argparse
import collections
import re
import sys

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        if op.startswith(TRANS):
            return "valu_trans"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu_lane"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "valu_cmp"
        if op.startswith(("v_cvt", "v_fma_mix", "v_pack", "v_perm", "v_bfi", "v_and_or", "v_lshl_or", "v_lshl_add")):
            return "valu_cvt_pack"
        if op.startswith(("v_mov", "v_accvgpr")):
            return "valu_mov"
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
        return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store") or op.startswith("ds_add") or op.startswith("ds_"):
        return "lds_write_atomic"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")):
        return "vmem_store"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "s_branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_sleep") or op.startswith("s_barrier") or op.startswith("s_setprio"):
        return "s_sync"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("buffer_") or op.startswith("global_"):
        return "vmem_other"
    return "other"


def kernel_lines(path, needle):
    lines = open(path).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\S*:", l) and needle in l:
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    if start is None:
        sys.exit(f"no kernel matching {needle}")
    return lines[start:end]


def census(lines):
    c = collections.Counter()
    detail = collections.Counter()
    for l in lines:
        s = l.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":") or re.match(r"^\.?[A-Za-z_0-9$]+:", s):
            continue
        op = s.split()[0]
        k = classify(op)
        c[k] += 1
        detail[op] += 1
    return c, detail


def issue_cycles(c, detail):
    plain = sum(v for k, v in c.items() if k.startswith("valu") and k != "valu_trans")
    mf16 = sum(v for k, v in detail.items() if k.startswith("v_mfma") and "16x16" in k)
    mf32 = sum(v for k, v in detail.items() if k.startswith("v_mfma") and "32x32" in k)
    return {"plain_valu": plain, "trans": c["valu_trans"], "mfma_16x16": mf16, "mfma_32x32": mf32,
            "vector_issue_cycles": 4 * plain + 8 * c["valu_trans"] + 8 * (mf16 + mf32),
            "matrix_pipe_cycles": 16 * mf16 + 32 * mf32}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--range", action="append", default=[])
    ap.add_argument("--labels", action="store_true")
    ap.add_argument("--top", type=int, default=0)
    a = ap.parse_args()
    kl = kernel_lines(a.file, a.kernel)
    if a.labels:
        for i, l in enumerate(kl, 1):
            if re.match(r"^\.LBB", l) or "s_cbranch" in l or "s_branch" in l or "s_endpgm" in l:
                print(i, l.strip()[:110])
        return
    ranges = a.range or [f"all:1:{len(kl)}"]
    for r in ranges:
        name, lo, hi = r.rsplit(":", 2)
        c, d = census(kl[int(lo) - 1:int(hi)])
        ic = issue_cycles(c, d)
        print(f"== {name} (lines {lo}-{hi})")
        print("   " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
        print("   " + "  ".join(f"{k}={v}" for k, v in ic.items()))
        if a.top:
            print("   top: " + ", ".join(f"{k} {v}" for k, v in d.most_common(a.top)))


if __name__ == "__main__":
    main()
