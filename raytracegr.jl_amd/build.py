"""hipcc driver: builds raytracegr.jl_amd/librtgr_hip.so for gfx950 IN-TREE (the .so travels with gpurun snapshots).

    python raytracegr.jl_amd/build.py [--force] [--resource-usage] [--save-temps]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "rtgr_hip.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "rtgr_persistent.hpp"), os.path.join(HERE, "csrc", "rtgr_tsit5_tables.hpp"), os.path.join(HERE, "csrc", "rtgr_physics.hpp"), os.path.join(HERE, "csrc", "rtgr_integrator.hpp"),
        os.path.join(HERE, "..", "include", "rtgr.h")]
OUT = os.path.join(HERE, "librtgr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function"]


def build(force=False, extra=(), verbose=True):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = [HIPCC] + FLAGS + list(extra) + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=HERE)
    return OUT


if __name__ == "__main__":
    extra = []
    if "--resource-usage" in sys.argv:
        extra.append("-Rpass-analysis=kernel-resource-usage")
    if "--save-temps" in sys.argv:
        os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
        extra += ["-save-temps=obj"]
    build(force=("--force" in sys.argv) or bool(extra), extra=extra)
