#!/usr/bin/env python3
"""Instruction counts of the LOOP bodies of every kernel of tools/ubench_overlap.hip, from the ISA.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -save-temps=obj tools/ubench_overlap.hip -o /tmp/ub/ubench_overlap
  python tools/ubench_overlap_isa.py /tmp/ub/ubench_overlap-hip-amdgcn-amd-amdhsa-gfx950.s

A loop body = a basic block that ends in a branch to its own label.  Printed per kernel (template arguments demangled):
v_mfma, v_min3, v_min, v_med3, v_and_or, other VALU, ds_read, s_nop (with the summed wait states), s_setprio, SALU / other --
so that a row of the timing table can be checked against what the compiler actually emitted (the round-3 version of the
microbenchmark lost half of its MFMAs to dead-code elimination and nobody saw it).
"""
import re
import subprocess
import sys
from collections import Counter


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        return name


KINDS = ["mfma", "valu", "both", "fold", "fold_before", "roles", "grp4"]


def pretty(dm):
    m = re.search(r"k<(\d+), (\d+), (\d+), (\d+), (\d+)>", dm)
    if not m:
        return dm
    kind, nv, op, prio, role = (int(x) for x in m.groups())
    return f"{KINDS[kind]:<11} NV={nv:<2} op={'min3' if op == 0 else 'min '} prio={prio} role_by={role}"


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        return "v_mfma"
    for p in ("v_min3", "v_med3", "v_and_or"):
        if op.startswith(p):
            return p
    if op.startswith("v_min_f32"):
        return "v_min"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "ds"
    if op == "s_nop":
        return "s_nop"
    if op == "s_setprio":
        return "s_setprio"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    return "salu_other"


def main(path):
    text = open(path).read().splitlines()
    kernels, cur, blocks, label = {}, None, None, None
    for ln in text:
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            blocks = kernels.setdefault(cur, [])
            label = "entry"
            blocks.append([label, []])
            continue
        if cur is None:
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            label = m.group(1)
            blocks.append([label, []])
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        blocks[-1][1].append(s.split(";")[0].strip())
    cols = ["v_mfma", "v_min3", "v_min", "v_med3", "v_and_or", "valu_other", "ds", "s_nop", "s_setprio", "s_waitcnt", "salu_other"]
    print(f"{'kernel':<52} {'loop':<10} " + " ".join(f"{c:>10}" for c in cols) + "   nop states")
    for name, blks in kernels.items():
        if "_Z1kI" not in name:
            continue
        title = pretty(demangle(name))
        for lab, ins in blks:
            if not ins:
                continue
            back = [n for n, i in enumerate(ins) if i.startswith("s_cbranch") and i.split()[-1] == lab]
            if not back:
                continue
            ins = ins[:back[-1] + 1]   # (what follows the back edge in the same block is the loop's exit code)
            c = Counter(classify(i) for i in ins)
            states = sum(int(i.split()[1], 0) + 1 for i in ins if i.startswith("s_nop"))
            print(f"{title:<52} {lab:<10} " + " ".join(f"{c.get(k, 0):>10}" for k in cols) + f"   {states}")


if __name__ == "__main__":
    main(sys.argv[1])
