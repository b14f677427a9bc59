#!/usr/bin/env python3
"""Command line of raytracegr.jl_amd/isa_exec.py: look for (and with --repair rewrite) vector instructions that stand ahead of a
FLOW block's EXEC flip in a gfx950 assembly listing — a code-generation fault of ROCm 7.2's LLVM, described in that module.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -S --cuda-device-only -o unit.s csrc/tu_f64_ksref.hip
    python tools/isa_exec_check.py unit.s [--repair fixed.s]          exit code 1 when the shape is found (and not repaired)
    python tools/isa_exec_check.py --object unit.hsaco | librtgr_hip.so   the library's audit of a code object's disassembly
                                                                          (rtgr_code_object_audit; loads librtgr_hip.so, no GPU needed)
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("rtgr_isa_exec", os.path.join(ROOT, "raytracegr.jl_amd", "isa_exec.py"))
isa_exec = importlib.util.module_from_spec(spec)     # (loaded by path: importing the package would load the HIP library)
spec.loader.exec_module(isa_exec)


def main():
    if sys.argv[1] == "--object":
        sys.path.insert(0, ROOT)
        from __graft_entry__ import load_package
        rt = load_package()
        found, report = sys.modules[rt.__name__ + ".user_metric"].audit(os.path.abspath(sys.argv[2]))
        print(report, end="")
        print(f"== FLOW blocks with vector instructions ahead of the EXEC flip: {found}")
        return 1 if found else 0
    path = sys.argv[1]
    lines = open(path).read().split("\n")
    hits = isa_exec.find(lines)
    for h in hits:
        print(h)
    print(f"== FLOW blocks with vector instructions ahead of the EXEC flip: {len(hits)}")
    if "--repair" in sys.argv and hits:
        fixed, n = isa_exec.repair(lines)
        with open(sys.argv[sys.argv.index("--repair") + 1], "w") as fh:
            fh.write("\n".join(fixed))
        print(f"== repaired {n}")
        return 0
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
