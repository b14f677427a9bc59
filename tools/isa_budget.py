#!/usr/bin/env python3
"""Instruction budget of a persistent integrate kernel's wave-step, by BLOCK of the step: which part of
integrate_body every instruction of the hot loop was compiled from — RHS / stage sums / error norm / reach bound (FAR) or sample-
point scan (NEAR, FULL) / step-size controller / commit + bookkeeping / refill + hand-over + record writes.

How: the translation unit is compiled to a device object with full debug info (-g does not change the code: checked against the
production object's loop, see --check), disassembled with llvm-objdump, and every instruction address of the kernel's hot loop
(the span of its longest backward branch) is resolved with `llvm-symbolizer --inlines` to its inline stack.  The frame INSIDE
integrate_body (rtgr_persistent.hpp) — the outermost one — gives the source line the instruction belongs to, whatever helper
(rfma, frsq, accel_*, fold_distances …) it was inlined from; lines map to blocks through the `// [budget: …]` markers that
bracket the blocks in the source.  No code is changed for counting.

    python tools/isa_budget.py tu_f64_ksref.hip integrate_far4_kernel [-DNAME=V ...] [--json out.json]

Static counts are an upper bound per iteration for the blocks that run in a minority of iterations (refill, hand-over, records);
the stage / RHS / norm / bound / controller blocks run every iteration, so their static count is their executed count, and the
PMC total (tools/prof_summary.py: SQ_INSTS_VALU per wave-step) minus their sum is what the rest really costs per iteration.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracegr.jl_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "--no-gpu-bundle-output"]


def block_ranges(path):
    """[(first line, last line, block)] from the `// [budget: name]` … `// [budget: end]` markers of a source file"""
    out, cur, start = [], None, 0
    for n, line in enumerate(open(path), 1):
        m = re.search(r"//\s*\[budget:\s*([a-z_ ]+?)\s*\]", line)
        if not m:
            continue
        if cur is not None:
            out.append((start, n - 1, cur))
        cur, start = (None, 0) if m.group(1) == "end" else (m.group(1), n)
    return out


def classify_valu(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64")):
        return "fma_f64"
    if op.startswith("v_mul_f64"):
        return "mul_f64"
    if op.startswith("v_add_f64"):
        return "add_f64"
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
        return "trans_f64"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32", op):
        return "trans_f32"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith(("v_mov", "v_cndmask", "v_accvgpr")):
        return "mov_sel"
    if op.startswith("v_cmp"):
        return "cmp"
    return "other_valu"


def main():
    args = sys.argv[1:]
    unit, kernel = args[0], args[1]
    extra = [a for a in args[2:] if a.startswith("-D")]
    jout = args[args.index("--json") + 1] if "--json" in args else None
    src = unit if os.path.isabs(unit) else os.path.join(CSRC, unit)
    with tempfile.TemporaryDirectory() as td:
        obj = os.path.join(td, "unit.o")
        subprocess.check_call([HIPCC] + FLAGS + extra + ["-g", "-c", "-o", obj, src], stderr=subprocess.DEVNULL)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], capture_output=True, text=True, check=True).stdout
        # ---- the kernel's instructions: (address, mnemonic, text)
        insts, inside = [], False
        for line in dis.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                inside = kernel in m.group(1) and m.group(1).startswith("_Z") or m.group(1) == kernel
                continue
            if not inside:
                continue
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
            if m:
                insts.append((int(m.group(3), 16), m.group(1), m.group(2)))
        if not insts:
            raise SystemExit(f"kernel matching {kernel!r} not found")
        addr_index = {a: i for i, (a, _, _) in enumerate(insts)}
        # ---- hot loop: the longest backward branch
        best = (0, 0, 0)
        for i, (a, op, txt) in enumerate(insts):
            if op.startswith(("s_cbranch", "s_branch")):
                m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", line := txt) or re.match(r"(-?\d+)", txt)
                tgt = None
                m2 = re.match(r"(-?\d+)", txt)
                if m2:   # simm16 in dwords relative to the next instruction
                    off = int(m2.group(1))
                    off = off - 65536 if off >= 32768 else off
                    tgt = a + 4 + 4 * off
                if tgt is not None and tgt in addr_index and tgt <= a and a - tgt > best[0]:
                    best = (a - tgt, addr_index[tgt], i)
        loop = insts[best[1]:best[2] + 1]
        # ---- inline stacks of the loop's instructions
        base = None
        for line in dis.split("\n"):
            m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
            if m and kernel in m.group(2):
                base = int(m.group(1), 16)
        q = "\n".join(f"0x{a:x}" for a, _, _ in loop) + "\n"
        sym = subprocess.run([os.path.join(LLVM, "llvm-symbolizer"), f"--obj={obj}", "--inlines", "--output-style=JSON"],
                             input=q, capture_output=True, text=True, check=True).stdout
        stacks = [json.loads(l) for l in sym.strip().split("\n") if l.strip()]
    ranges = {os.path.basename(p): block_ranges(p) for p in (os.path.join(CSRC, "rtgr_persistent.hpp"), os.path.join(CSRC, "rtgr_packed_f32.hpp"))}

    def block_of(stack):
        # frames are innermost first; take the frame of the kernel BODY function (integrate_body / integrate2_body: the __global__
        # wrapper above it is one line), whatever helper the instruction was inlined from
        for fr in reversed(stack.get("Symbol", [])):
            fn = os.path.basename(fr.get("FileName", ""))
            if fn in ranges and "_body" in fr.get("FunctionName", ""):
                for a, b, name in ranges[fn]:
                    if a <= fr.get("Line", 0) <= b:
                        return name
                return "unmarked"
        return "unmarked"

    per = collections.defaultdict(collections.Counter)
    for (a, op, txt), st in zip(loop, stacks):
        blk = block_of(st)
        kind = classify_valu(op) if op.startswith("v_") else ("salu" if op.startswith("s_") else "mem")
        per[blk][kind] += 1
    valu_kinds = ("fma_f64", "mul_f64", "add_f64", "trans_f64", "trans_f32", "cvt", "mov_sel", "cmp", "other_valu")
    total_valu = sum(sum(c[k] for k in valu_kinds) for c in per.values())
    print(f"== {kernel} ({os.path.basename(src)} {' '.join(extra)}): hot loop {len(loop)} instructions, {total_valu} VALU (static)")
    print(f"{'block':26s} {'VALU':>6s} {'share':>6s}   fma  mul  add tr64 tr32  cvt mov/sel cmp other | salu  mem")
    table = {}
    for blk, c in sorted(per.items(), key=lambda kv: -sum(kv[1][k] for k in valu_kinds)):
        v = sum(c[k] for k in valu_kinds)
        table[blk] = {"valu": v, **{k: c[k] for k in valu_kinds}, "salu": c["salu"], "mem": c["mem"]}
        print(f"{blk:26s} {v:6d} {v / max(total_valu, 1):6.1%}  {c['fma_f64']:4d} {c['mul_f64']:4d} {c['add_f64']:4d} {c['trans_f64']:4d} {c['trans_f32']:4d} "
              f"{c['cvt']:4d} {c['mov_sel']:6d} {c['cmp']:4d} {c['other_valu']:5d} | {c['salu']:4d} {c['mem']:4d}")
    if jout:
        json.dump({"kernel": kernel, "unit": os.path.basename(src), "defines": extra, "loop_instructions": len(loop), "loop_valu_static": total_valu,
                   "blocks": table}, open(jout, "w"), indent=1)


if __name__ == "__main__":
    main()
