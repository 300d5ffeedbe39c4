"""Inline-asm consumers of MFMA results are invisible to the compiler's hazard recogniser ("a compiler
bump can silently reorder them").  This test compiles the two kernel sources to gfx950 assembly with the Makefile's flags
and checks, for every instruction inside an inline-asm block, that no VGPR it READS was written by a v_mfma fewer than the
required wait states earlier (CDNA3/4 ISA: an XDL write of VGPRs followed by a VALU read needs 5 / 11 / 19 wait states for
2- / 8- / 16-pass MFMAs -- one wait state = one issue slot of four cycles, an MFMA keeps the matrix pipe for `passes` of them; the kernels use the 8-pass v_mfma_f32_32x32x16_f16 and the 16-pass v_mfma_f32_32x32x2_f32).
Instructions the compiler emitted itself are covered by its own hazard recogniser and are not checked.  CPU only (hipcc
cross-compiles); the assembly is cached under /tmp by source hash."""
import hashlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flux3d.jl_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-mllvm",
         "-amdgpu-mfma-vgpr-form", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]
PASSES = {"v_mfma_f32_32x32x16_f16": 8, "v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_32x32x2f32": 16}
NEED = {2: 5, 4: 7, 8: 11, 16: 19}


def _assembly(name):
    src = os.path.join(CSRC, name)
    h = hashlib.sha256()
    for f in (src, os.path.join(CSRC, "fx3d_common.h"), os.path.join(ROOT, "include", "flux3d_hip.h")):
        h.update(open(f, "rb").read())
    out = f"/tmp/fx3d_isa_{name}_{h.hexdigest()[:16]}.s"
    if not os.path.exists(out):
        subprocess.run([HIPCC] + FLAGS + [src, "-o", out], check=True, capture_output=True, cwd=CSRC, timeout=600)
    return open(out).read()


def _regs(tok):
    """VGPR numbers named by an operand token: v7, v[4:7]; anything else -> empty."""
    m = re.fullmatch(r"-?\|?v(\d+)\|?", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _check(asm):
    """(violations, number of inline-asm instructions checked against a recent MFMA result)."""
    viol, checked = [], 0
    recent = []          # (dest registers, wait-state clock at issue, passes, text)
    clock, in_asm, fn, xdl_free = 0, False, "?", 0
    for ln, line in enumerate(asm.splitlines(), 1):
        t = line.strip()
        if not t:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith(";") or t.startswith("."):
            if t.startswith(".LBB") or t.endswith(":"):
                recent = []  # a branch target: what was issued before is unknown, the straight-line window restarts
            continue
        if t.endswith(":"):
            fn = t[:-1]
            recent, clock, xdl_free = [], 0, 0
            continue
        t = t.split(";")[0].strip()
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if in_asm and op.startswith("v_") and not op.startswith("v_mfma"):
            reads = set()
            for o in ops[1:]:
                reads |= _regs(o)
            # (accumulating forms, v_min3 d, d, a, b, name their destination among the sources: ops[1:] has it)
            for dest, at, passes, text in recent:
                if reads & dest:
                    checked += 1
                    if clock - at < NEED[passes]:
                        viol.append(f"{fn} line {ln}: `{t}` reads the result of `{text}` after {clock - at} wait states (< {NEED[passes]})")
        if op in PASSES and ops:
            clock = max(clock, xdl_free)        # the matrix pipe takes one MFMA at a time: back-to-back MFMAs issue `passes` apart
            xdl_free = clock + PASSES[op]
            recent.append((_regs(ops[0]), clock, PASSES[op], t))
            recent = recent[-8:]
        clock += (int(ops[0], 0) + 1) if op == "s_nop" and ops else 1
    return viol, checked


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("name,min_checked", [("knn_d3.hip", 8), ("knn_mfma.hip", 8), ("chamfer.hip", 0)])
def test_inline_asm_consumers_keep_their_distance_from_the_mfma(name, min_checked):
    viol, checked = _check(_assembly(name))
    assert not viol, "\n".join(viol[:10])
    assert checked >= min_checked, f"only {checked} inline-asm reads of MFMA results found in {name}: the check is not looking at the kernels"
