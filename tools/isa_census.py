#!/usr/bin/env python3
"""Static instruction census of k_paths by source region (profiles/r05_adv_census.md).

Builds the device code with -DRT_ISA_MARKS (rt_device.h: RT_MARK emits assembler COMMENTS, no instruction), takes the
assembly of one k_paths instantiation and counts vector instructions per region and per category.  Regions follow the
code layout (the compiler keeps the blocks in source order; instructions it moved across a comment are counted where they
landed), so the numbers are a map of where the block's instructions are, not a trace of what a lane executes.

    tools/isa_census.py [mangled-name-prefix]      (default: the 4-wave VERIFY build bench.py times)
"""
import collections
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rtcuda_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -Wno-unused-function -Wno-unused-result".split()

CATS = [
    ("int xor/shift/add/logic (XORWOW, bit tricks, addresses)", r"^v_(xor|lshlrev|lshrrev|ashrrev|add_u32|sub_u32|subrev_u32|add3_u32|lshl_add_u32|xad_u32|and|or|or3|and_or|bfe|bfi|not|mul_u32_u24|mul_lo_u32|mad_u32_u24|mad_u64_u32|mul_hi_u32|add_co|addc_co|sub_co|subb_co|lshl_or|alignbit|perm|min_u32|max_u32|min_i32|max_i32|add_lshl|xor3|mbcnt|bcnt)"),
    ("division / sqrt expansion (div_scale, div_fmas, div_fixup, rcp, rsq, sqrt, ldexp, frexp)", r"^v_(div_scale|div_fmas|div_fixup|rcp|rsq|sqrt|ldexp|frexp)"),
    ("fma (incl. the expansions' Newton steps)", r"^v_(fma_f32|fmac_f32|fmaak|fmamk)"),
    ("mul / add / sub f32", r"^v_(mul_f32|add_f32|sub_f32|subrev_f32|mul_legacy)"),
    ("packed f32", r"^v_pk_"),
    ("min / max / med f32", r"^v_(min|max|med)3?_(f32|num_f32)"),
    ("compare", r"^v_cmp"),
    ("select (cndmask)", r"^v_cndmask"),
    ("move / readlane / writelane", r"^v_(mov|readlane|writelane|readfirstlane|swap|accvgpr)"),
    ("convert / floor / fract / trunc", r"^v_(cvt|floor|fract|trunc|rndne|ceil)"),
]
SKIP = ("VALU", "LDS", "VMEM", "SALU/branch/waitcnt", "s_waitcnt")


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "_Z7k_pathsILb1ELb1ELb1ELi4ELb0ELb0ELb1EE"
    out = "/tmp/rt_isa_census.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-DRT_ISA_MARKS", "--cuda-device-only", "-S", "-o", out,
                                                              os.path.join(CSRC, "rtcuda_amd.hip")], stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith(want) and ":" in ln.split(";")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    region = "prologue"
    per, order = {}, []
    for ln in lines[start:end]:
        t = ln.strip()
        m = re.match(r"; RT_MARK (\S+)", t)
        if m:
            region = m.group(1)
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.split(";")[0].strip().endswith(":"):
            continue
        op = t.split()[0]
        d = per.setdefault(region, collections.Counter())
        if region not in order:
            order.append(region)
        if op.startswith("v_"):
            cat = next((name for name, pat in CATS if re.match(pat, op)), "other vector: " + op)
            d["VALU"] += 1
            d[cat] += 1
        elif op.startswith("ds_"):
            d["LDS"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            d["VMEM"] += 1
        elif op.startswith("s_"):
            d["SALU/branch/waitcnt"] += 1
            if op.startswith("s_waitcnt"):
                d["s_waitcnt"] += 1
    tot = collections.Counter()
    print(f"function {want}: assembly lines {start}-{end}")
    for r in order:
        d = per[r]
        tot.update(d)
        print(f"\n## {r}: VALU {d['VALU']}, LDS {d['LDS']}, VMEM {d['VMEM']}, scalar {d['SALU/branch/waitcnt']} (s_waitcnt {d['s_waitcnt']})")
        for k, v in sorted(d.items(), key=lambda kv: -kv[1]):
            if k not in SKIP:
                print(f"    {v:5d}  {k}")
    print(f"\n## whole kernel: VALU {tot['VALU']}, LDS {tot['LDS']}, VMEM {tot['VMEM']}, scalar {tot['SALU/branch/waitcnt']}")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        if k not in SKIP:
            print(f"    {v:5d}  {k}")


if __name__ == "__main__":
    main()
