"""Builds the macro-gated experiment variants of the library (LDS stage storage, other waves/SIMD) into
raytracegr.jl_amd/build/variants/librtgr_<name>.so; tools/ab_variants.sh times them against the default build (RTGR_LIB).
Results of round 2: DESIGN.md §4.2 "LDS stage storage"."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "raytracegr.jl_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
V = {"ldsk_gen2": ["-DRTGR_LDSK_GENERIC=1"],
     "ldsk_gen3": ["-DRTGR_LDSK_GENERIC=1", "-DRTGR_WAVES_PER_SIMD_GENERIC=3"],
     "gen3": ["-DRTGR_WAVES_PER_SIMD_GENERIC=3"],
     "ldsk_spin4": ["-DRTGR_LDSK_SPIN_FAR=1", "-DRTGR_WAVES_PER_SIMD_SPIN_FAR=4"],
     "spin4": ["-DRTGR_WAVES_PER_SIMD_SPIN_FAR=4"]}
os.makedirs(os.path.join(ROOT, "raytracegr.jl_amd", "build", "variants"), exist_ok=True)
for name, extra in V.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    out = os.path.join(ROOT, "raytracegr.jl_amd", "build", "variants", f"librtgr_{name}.so")
    b.build(extra=extra + ["-Rpass-analysis=kernel-resource-usage"] if False else extra, out=out, verbose=False)
    print("built", out)
