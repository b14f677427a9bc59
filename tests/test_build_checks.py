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


def test_plain_build_relinks_after_an_experiment_build(tmp_path):
    """build.py records which flag set librtgr_hip.so was linked from (librtgr_hip.so.tag).  Sequence of ADVICE r2: a plain
    build, an experiment build (-DRTGR_ROOT_STATS), a plain build again — the last one finds its own objects fresh AND older
    than the library, and must still relink, or tests and bench would run the experiment binary.  (A stand-in compiler
    that only creates its output file: the logic under test is the driver's, not hipcc's.)"""
    import importlib.util
    fake = tmp_path / "fake_hipcc"
    log = tmp_path / "calls.log"
    fake.write_text("#!/bin/sh\nprev=\nfor a in \"$@\"; do [ \"$prev\" = -o ] && out=\"$a\"; prev=\"$a\"; done\n"
                    f"echo \"$@\" >> {log}\n: > \"$out\"\n")
    fake.chmod(0o755)
    spec = importlib.util.spec_from_file_location("rtgr_build_t", os.path.join(ROOT, "raytracegr.jl_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.HIPCC = str(fake)
    out, obj = str(tmp_path / "lib.so"), str(tmp_path / "obj")

    def links():
        return sum(1 for l in open(log) if " -shared " in l) if log.exists() else 0

    m.build(out=out, obj_dir=obj, verbose=False)
    assert links() == 1 and m.linked_tag(out) == "std"
    m.build(out=out, obj_dir=obj, verbose=False)
    assert links() == 1                                   # nothing to do
    m.build(out=out, obj_dir=obj, verbose=False, extra=("-DRTGR_ROOT_STATS",))
    assert links() == 2 and m.linked_tag(out) not in (None, "std")
    m.build(out=out, obj_dir=obj, verbose=False)          # objects of the plain build are fresh and OLDER than the library
    assert links() == 3 and m.linked_tag(out) == "std"
