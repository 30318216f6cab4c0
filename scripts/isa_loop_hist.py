#!/usr/bin/env python3
"""Instruction histogram of the innermost loops of one kernel in the hipcc ISA dump (make asm).
usage: isa_loop_hist.py file.s mangled_kernel_name [--dump]"""
import collections
import re
import sys

src, kern = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.append((labels[t], i))


def cat(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_sqrt_f64", "v_rcp_f64", "v_div", "v_rsq_f64", "v_max_f64", "v_min_f64", "v_fract", "v_floor_f64", "v_trunc_f64", "v_ldexp_f64", "v_frexp")):
        return "valu_f64"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "valu_cmp"
    if op.startswith("v_cvt"):
        return "valu_cvt"
    if op.startswith(("v_mov", "v_accvgpr", "v_pk_mov")):
        return "valu_mov"
    if op.startswith("v_cndmask"):
        return "valu_select"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "valu_lane"
    if op.startswith("v_"):
        return "valu_int"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


for a, b in sorted(set(loops), key=lambda x: x[1] - x[0]):
    ins = [l.split()[0] for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    h = collections.Counter(cat(o) for o in ins)
    valu = sum(v for k, v in h.items() if k.startswith("valu"))
    print(f"loop lines {a}-{b}: {len(ins)} instr, VALU {valu}: " + ", ".join(f"{k}={v}" for k, v in sorted(h.items())))
    if "--dump" in sys.argv and len(ins) > 200:
        ops = collections.Counter(ins)
        print("   " + ", ".join(f"{k}:{v}" for k, v in ops.most_common(60)))
