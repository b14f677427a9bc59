"""Build-time checks on the generated gfx950 ISA (CPU only: hipcc cross-compiles without a GPU).

The one-instruction min / max / clamp helpers of the hot loop are raw inline asm (rtgr_physics.hpp rmin / rmax / rmaxabs,
rtgr_persistent.hpp fmax1 / fmin1).  LLVM's hazard recogniser treats inline asm as opaque, so a transcendental result
(v_rcp / v_rsq / v_exp / v_log …) consumed by the very next VALU instruction would get no s_nop for the gfx940-family
"trans use" hazard.  tools/isa_mix.py --hazards scans the final ISA of every integrate / prepare kernel for that
adjacency; this test fails the suite when one appears."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "raytracegr.jl_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.parametrize("unit", ["tu_f64_ksref", "tu_f64_kstrue", "tu_f32_closed"])
def test_no_trans_use_hazard_next_to_inline_asm(unit, tmp_path):
    asm = str(tmp_path / f"{unit}.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only",
                           "-Wno-unused-command-line-argument", "-o", asm, os.path.join(CSRC, unit + ".hip")],
                          stderr=subprocess.DEVNULL)
    kernels = [l.split(":")[0] for l in open(asm) if l.startswith("_ZN4rtgr") and l.rstrip().split(";")[0].rstrip().endswith(":")]
    kernels = [k for k in kernels if "integrate" in k or "prepare_kernel" in k or "resolve_kernel" in k]
    assert len(kernels) >= 8, kernels
    for k in kernels:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), asm, k, "--hazards"],
                           capture_output=True, text=True)
        assert r.returncode == 0, (k, r.stdout[-2000:])
