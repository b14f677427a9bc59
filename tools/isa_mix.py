#!/usr/bin/env python3
"""Static instruction mix of a kernel's gfx950 ISA (hipcc -S --cuda-device-only output): every instruction of the
kernel, and of its HOT LOOP (the span of the longest backward branch), by class.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -S --cuda-device-only -o ksref.s csrc/tu_f64_ksref.hip
    python tools/isa_mix.py ksref.s integrate_far4_kernel [--list CLASS] [--hazards]

The static loop count is an UPPER bound of what a wave executes per iteration (divergent regions inside the loop body
are counted once whether or not a lane enters them; the refill / hand-over / event-record blocks run in a minority of
iterations).  The executed count comes from the PMC pass (tools/prof_summary.py: SQ_INSTS_VALU_* per wave-step); this
script is its cross-check and the tool for finding WHERE the slots go.

--hazards: build-time check for the gfx940-family "trans use" hazard around inline asm (LLVM's hazard recogniser treats
inline asm as opaque): a transcendental's result (v_rcp/rsq/sqrt/exp/log/sin/cos, f32 or f64) consumed by the
very next VALU instruction.  Exit code 1 when one is found.
"""
import collections
import re
import sys

TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f16|f32|f64)")


def classify(op):
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"):
        return "fma_f64"
    if op.startswith("v_mul_f64"):
        return "mul_f64"
    if op.startswith("v_add_f64"):
        return "add_f64"
    if op.startswith(("v_max_f64", "v_min_f64")):
        return "minmax_f64"
    if op.startswith("v_cmp") and "f64" in op:
        return "cmp_f64"
    if TRANS.match(op):
        return "trans_" + op.rsplit("_", 1)[1][:3]
    if op.startswith(("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")):
        return "pk_f32"      # two f32 operations per lane in ONE issue slot (the PMC counters count such an instruction once)
    if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mac_f32")):
        return "fma_f32"
    if op.startswith("v_mul_f32"):
        return "mul_f32"
    if op.startswith(("v_add_f32", "v_sub_f32")):
        return "add_f32"
    if op.startswith(("v_pk_", "v_max_f32", "v_min_f32", "v_med3_f32")):
        return "f32_other"
    if op.startswith("v_cmp"):
        return "cmp_other"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith(("v_mov", "v_accvgpr", "v_readfirstlane", "v_readlane", "v_writelane", "v_swap")):
        return "mov"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def kernel_body(lines, needle):
    start = None
    for i, l in enumerate(lines):
        head = l.split(";")[0].rstrip()
        if start is None and head.endswith(":") and needle in head and (head.startswith("_Z") or head == needle + ":"):   # (extern "C" kernels of the user units are not mangled)
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            return lines[start:i]
    raise SystemExit(f"kernel matching {needle!r} not found")


def main():
    path, needle = sys.argv[1], sys.argv[2]
    want_list = sys.argv[sys.argv.index("--list") + 1] if "--list" in sys.argv else None
    lines = [l.rstrip("\n") for l in open(path)]
    body = kernel_body(lines, needle)
    insts, labels = [], {}
    for l in body:
        s = l.split(";")[0].strip()
        if not s:
            continue
        if s.endswith(":"):
            labels[s[:-1]] = len(insts)
            continue
        op = s.split()[0]
        if op.startswith("."):
            continue
        insts.append((op, s))
    # the longest backward branch = the hot loop
    best = (0, 0, 0)
    for i, (op, s) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i and i - labels[tgt] > best[0]:
                best = (i - labels[tgt], labels[tgt], i)
    span = insts[best[1]:best[2] + 1]

    def table(seq, title):
        c = collections.Counter(classify(op) for op, _ in seq)
        valu = sum(v for k, v in c.items() if k not in ("salu", "vmem", "lds", "other"))
        f64 = c["fma_f64"] + c["mul_f64"] + c["add_f64"]
        flop = 2 * c["fma_f64"] + c["mul_f64"] + c["add_f64"]
        print(f"== {title}: {len(seq)} instructions, {valu} VALU, {f64} f64 FMA/MUL/ADD ({c['fma_f64']}/{c['mul_f64']}/{c['add_f64']}), "
              f"{flop} f64 flop per lane, flop per VALU slot {flop / max(valu, 1):.3f} of 2")
        pk = collections.Counter(op.split("_e")[0] for op, _ in seq if classify(op) == "pk_f32")
        if pk or c["fma_f32"] + c["mul_f32"] + c["add_f32"] > 50:
            f32 = c["fma_f32"] + c["mul_f32"] + c["add_f32"]
            flop32 = 2 * c["fma_f32"] + c["mul_f32"] + c["add_f32"] + 4 * pk["v_pk_fma_f32"] + 2 * (pk["v_pk_mul_f32"] + pk["v_pk_add_f32"])
            print(f"   f32: {f32} scalar FMA/MUL/ADD ({c['fma_f32']}/{c['mul_f32']}/{c['add_f32']}) + {sum(pk.values())} PACKED "
                  f"({pk['v_pk_fma_f32']} pk_fma / {pk['v_pk_mul_f32']} pk_mul / {pk['v_pk_add_f32']} pk_add) = {flop32} f32 flop per lane; "
                  f"{(sum(pk.values())) / max(f32 + sum(pk.values()), 1):.2f} of the arithmetic instructions are packed, "
                  f"flop per VALU slot {flop32 / max(valu, 1):.3f} of 4")
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
            print(f"   {k:12s} {v:6d}")
        return c

    table(insts, f"whole kernel ({needle})")
    table(span, "hot loop (static, upper bound per iteration)")
    if want_list:
        for op, s in span:
            if classify(op) == want_list:
                print("   ", s)
    if "--hazards" in sys.argv:
        bad = 0
        for (op, s), (op2, s2) in zip(insts, insts[1:]):
            if TRANS.match(op) and op2.startswith("v_"):
                dst = s.split()[1].rstrip(",")
                srcs = s2.split(None, 1)[1] if " " in s2 else ""
                srcs = srcs.split(",", 1)[1] if "," in srcs else ""
                regs = set(re.findall(r"v\[?\d+(?::\d+)?\]?", srcs))
                if dst in regs:
                    bad += 1
                    print("trans-use hazard:", s, "->", s2)
        print(f"== trans-use adjacency: {bad} found")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
