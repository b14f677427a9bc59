#!/usr/bin/env python3
"""Static cost of ONE MORE SPHERE in the integrate kernels: the innermost loops of a kernel's listing that walk the
object list — recognised by their scalar loads of an object's fields; a loop without a compare of the kind field is a sphere loop — with their VALU,
SALU and scalar-load counts per iteration.  FAR pass: the reach test.  NEAR pass: the reach test, then the two blocks of the sample-point
scan (5 + 4 points), each in a variant per sign of R.  Needs hipcc, no GPU.

    python tools/object_loop_cost.py [tu_f64_ksref.hip]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracegr.jl_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
KERNELS = (("FAR  (integrate_far4_kernel<double, KS_REF>)", "_ZN4rtgr21integrate_far4_kernelIdLi1EEEvNS_13IntegrateArgsIT_EE"),
           ("NEAR (integrate_kernel<double, KS_REF, a = 0, 10 points, NEAR>)", "_ZN4rtgr16integrate_kernelIdLi1ELb0ELb1ELi2EEEvNS_13IntegrateArgsIT_EE"))


def loops(body):
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    out = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] < 130:
            seg = body[labels[m.group(1)]:i + 1]
            loads = sum(1 for s in seg if re.match(r"\s+s_load", s))
            if not loads:
                continue
            out.append(dict(first=labels[m.group(1)], last=i, valu=sum(1 for s in seg if re.match(r"\s+v_", s)),
                            salu=sum(1 for s in seg if re.match(r"\s+s_", s) and not re.match(r"\s+s_(load|waitcnt|nop)", s)),
                            loads=loads, sphere=not any(re.search(r"s_cmp_(eq|lg|lt|gt)_[iu]32 s\d+, [1-4]\b", s) for s in seg)))   # (no compare of a kind field)
    return out


def main():
    unit = sys.argv[1] if len(sys.argv) > 1 else "tu_f64_ksref.hip"
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "unit.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S", "-o", asm,
                               os.path.join(CSRC, unit)], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    for title, sym in KERNELS:
        start = next((i for i, l in enumerate(lines) if l.startswith(sym + ":")), None)
        if start is None:
            continue
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        print(f"== {title}: {end - start} lines")
        for lo in loops(lines[start:end]):
            print(f"   loop at lines {lo['first']:5d}-{lo['last']:5d}: {lo['valu']:3d} VALU  {lo['salu']:3d} SALU  {lo['loads']} scalar loads"
                  f"   {'<- a SPHERE of the list (no dispatch on the kind)' if lo['sphere'] else '   (another kind: dispatch inside)'}")


if __name__ == "__main__":
    main()
