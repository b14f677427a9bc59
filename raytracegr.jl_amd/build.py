"""hipcc driver: builds raytracegr.jl_amd/librtgr_hip.so for gfx950 IN-TREE (the .so travels with gpurun snapshots).

    python raytracegr.jl_amd/build.py [--force] [--resource-usage] [--save-temps] [-DNAME[=V] ...]

The library is several translation units (csrc/tu_*.hip hold the kernels of one metric-variant group each,
csrc/rtgr_misc.hip the small kernels, csrc/rtgr_api.hip the C ABI and no kernel at all); they are compiled in parallel
into raytracegr.jl_amd/build/obj/ and linked with `hipcc -shared`.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
UNITS = ["tu_f64_ksref.hip", "tu_f64_kstrue.hip", "tu_f64_generic.hip", "tu_f64_mink.hip", "tu_f32_closed.hip",
         "tu_f32_generic.hip", "rtgr_misc.hip", "rtgr_api.hip"]
# the device-side headers: what the KERNELS are made of (bench.py keys its roofline profile on their hash)
KERNEL_HEADERS = ["rtgr_args.hpp", "rtgr_physics.hpp", "rtgr_integrator.hpp", "rtgr_persistent.hpp", "rtgr_packed_f32.hpp",
                  "rtgr_tsit5_tables.hpp"]
HEADERS = KERNEL_HEADERS + ["rtgr_host.hpp", "rtgr_pipeline.hpp", "rtgr_isa_audit.hpp", "rtgr_isa_repair.hpp", "rtgr_unit_build.hpp"]
DEPS = [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(HERE, "..", "include", "rtgr.h")]
OUT = os.path.join(HERE, "librtgr_hip.so")
OBJ = os.path.join(HERE, "build", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def kernel_source_hash():
    """sha256 over the device headers + compile flags: identifies the kernels' source (profiles/*/flops.json records it)."""
    h = hashlib.sha256()
    for f in KERNEL_HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def _stale(target, deps):
    return (not os.path.exists(target)) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)


def build(force=False, extra=(), verbose=True, out=OUT, obj_dir=OBJ):
    os.makedirs(obj_dir, exist_ok=True)
    tag = hashlib.sha256(" ".join(extra).encode()).hexdigest()[:8] if extra else "std"
    jobs = []
    for u in UNITS:
        src = os.path.join(CSRC, u)
        obj = os.path.join(obj_dir, f"{os.path.splitext(u)[0]}.{tag}.o")
        if force or _stale(obj, [src] + DEPS):
            jobs.append((src, obj))
    objs = [os.path.join(obj_dir, f"{os.path.splitext(u)[0]}.{tag}.o") for u in UNITS]

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + list(extra) + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, cwd=obj_dir, capture_output=True, text=True)
        return job, r

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for (src, obj), r in ex.map(compile_one, jobs):
                if r.stderr and verbose:
                    sys.stderr.write(r.stderr)
                if r.returncode != 0:
                    if os.path.exists(obj):
                        os.unlink(obj)
                    raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-6000:]}")
    # Which flag set the existing library was linked from is recorded next to it: after an experiment build
    # (`build.py -DRTGR_ROOT_STATS`, --save-temps, …) a plain build finds its own objects fresh and OLDER than the library,
    # and must still relink — or tests and bench would silently run the experiment binary.
    tag_file = out + ".tag"
    linked_tag = open(tag_file).read().strip() if os.path.exists(tag_file) else None
    if jobs or force or _stale(out, objs) or linked_tag != tag:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", out] + objs + ["-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=HERE)
        with open(tag_file, "w") as fh:
            fh.write(tag + "\n")
    return out


def linked_tag(out=OUT):
    """'std' for the production flag set, a hash for an experiment build, None when unknown."""
    try:
        return open(out + ".tag").read().strip()
    except OSError:
        return None


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    if "--resource-usage" in sys.argv:
        extra.append("-Rpass-analysis=kernel-resource-usage")
    if "--save-temps" in sys.argv:
        extra += ["-save-temps=obj"]
    # (--resource-usage / --save-temps only print or write something when the units are actually compiled)
    build(force=("--force" in sys.argv or "--resource-usage" in sys.argv or "--save-temps" in sys.argv), extra=extra)
