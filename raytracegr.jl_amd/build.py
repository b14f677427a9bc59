"""hipcc driver: builds raytracegr.jl_amd/librtgr_hip.so for gfx950 IN-TREE (the .so travels with gpurun snapshots).

    python raytracegr.jl_amd/build.py [--force] [--resource-usage] [--save-temps] [--via-listing] [-DNAME[=V] ...]

The library is several translation units (csrc/tu_*.hip hold the kernels of one metric-variant group each,
csrc/rtgr_misc.hip the small kernels; the host side is kernel-free: csrc/rtgr_context.hip, rtgr_host_pipeline.hip, rtgr_sharded.hip,
rtgr_hooks.hip, rtgr_units.hip and the extern "C" shims of csrc/rtgr_abi.hip — rtgr_internal.hpp says what is where); they are compiled in parallel
into raytracegr.jl_amd/build/obj/ and linked with `hipcc -shared`.

After linking, the library AUDITS the kernels embedded in it for the EXEC-flip fault of ROCm 7.2's compiler (DESIGN.md §4.6;
rtgr_code_object_audit).  Clean so far — by a rule that sees the fault only where the compiler kept the `s_cbranch_execz` that skips the
`then` side (isa_exec.py): "clean" means "none of THAT shape".  If an edit of the kernels ever brings the fault out, the kernel units are rebuilt through
their assembly LISTING (--via-listing, or RTGR_BUILD_VIA_LISTING=1, forces that route): `hipcc --cuda-device-only -S`, the check /
repair of isa_exec.py, assembler, lld, clang-offload-bundler, then the host half with `-fcuda-include-gpubinary` — the steps hipcc
runs itself, with a look at the listing in between; for a unit that needs no repair the device code is the same, instruction for
instruction (tests/test_build_checks.py).
"""
import concurrent.futures
import hashlib
import importlib.util
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
UNITS = ["tu_f64_ksref.hip", "tu_f64_kstrue.hip", "tu_f64_generic.hip", "tu_f64_mink.hip", "tu_f32_closed.hip",
         "tu_f32_generic.hip", "rtgr_misc.hip", "rtgr_context.hip", "rtgr_host_pipeline.hip", "rtgr_sharded.hip", "rtgr_hooks.hip",
         "rtgr_units.hip", "rtgr_abi.hip"]
# the device-side headers: what the KERNELS are made of (bench.py keys its roofline profile on their hash)
KERNEL_HEADERS = ["rtgr_args.hpp", "rtgr_physics.hpp", "rtgr_integrator.hpp", "rtgr_persistent.hpp", "rtgr_packed_f32.hpp",
                  "rtgr_tsit5_tables.hpp"]
HEADERS = KERNEL_HEADERS + ["rtgr_host.hpp", "rtgr_internal.hpp", "rtgr_pipeline.hpp", "rtgr_isa_audit.hpp", "rtgr_isa_repair.hpp", "rtgr_unit_build.hpp"]
DEPS = [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(HERE, "..", "include", "rtgr.h")]
OUT = os.path.join(HERE, "librtgr_hip.so")
OBJ = os.path.join(HERE, "build", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LLVM_BIN = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin")
# no kernel in them: they never need the listing route
HOST_ONLY_UNITS = ("rtgr_context.hip", "rtgr_host_pipeline.hip", "rtgr_sharded.hip", "rtgr_hooks.hip", "rtgr_units.hip", "rtgr_abi.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def kernel_source_hash(out=None):
    """sha256 over the device headers + compile flags (+ the listing check / repair and whether the linked library went through it:
    a repaired build is other device code than an unrepaired one from the same sources — ADVICE r4): identifies the kernels
    (profiles/*/flops.json records it)."""
    h = hashlib.sha256()
    for f in KERNEL_HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    with open(os.path.join(HERE, "isa_exec.py"), "rb") as fh:
        h.update(fh.read())
    if (linked_tag(out or OUT) or "").endswith("+listing"):
        h.update(b"via-listing")
    return h.hexdigest()[:16]


# the headers a run-time unit is compiled against (user_metric._HEADERS, rtgr_units.hip header_hash_of: same files, same order)
UNIT_HEADERS = ["rtgr_args.hpp", "rtgr_physics.hpp", "rtgr_integrator.hpp", "rtgr_persistent.hpp", "rtgr_tsit5_tables.hpp",
                os.path.join("..", "..", "include", "rtgr.h")]


def header_hash():
    """FNV-1a (64 bit) over the device headers of run-time units: compiled into the library (-DRTGR_HEADER_HASH, rtgr_units.hip) and
    into every unit (rtgr_user_header_hash); a unit built from other headers than the library's kernels is refused at load."""
    h = 1469598103934665603
    for f in UNIT_HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            for b in fh.read():
                h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h or 1


def _stale(target, deps):
    return (not os.path.exists(target)) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)


def _isa_exec():
    spec = importlib.util.spec_from_file_location("rtgr_isa_exec_b", os.path.join(HERE, "isa_exec.py"))
    m = importlib.util.module_from_spec(spec)   # (by path: importing the package would load the library being built)
    spec.loader.exec_module(m)
    return m


def compile_via_listing(src, obj, extra=(), verbose=True, cwd=None):
    """One translation unit through its device LISTING: what `hipcc -c` does in one go, in five steps with the check / repair of
    isa_exec.py after the first.  Returns the number of FLOW blocks rewritten."""
    base = obj[:-2] if obj.endswith(".o") else obj
    asm, dev_o, hsaco, fatbin = base + ".dev.s", base + ".dev.o", base + ".hsaco", base + ".hipfb"
    cuid = "-cuid=" + hashlib.sha256(os.path.basename(src).encode()).hexdigest()[:16]   # the two halves must agree on it

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{cmd[0]} failed on {src}:\n{r.stderr[-6000:]}")
        return r

    run([HIPCC] + FLAGS + list(extra) + [cuid, "--cuda-device-only", "-S", "-o", asm, src])
    with open(asm) as fh:
        lines = fh.read().split("\n")
    lines, repaired = _isa_exec().repair(lines)      # RepairError: a block the rewrite is not proven for — the build stops there
    if repaired:
        with open(asm, "w") as fh:
            fh.write("\n".join(lines))
        if verbose:
            print(f"   {os.path.basename(src)}: {repaired} FLOW block(s) rewritten in the listing", flush=True)
    run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm, "-o", dev_o])
    run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dev_o])
    run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", f"-input={hsaco}", f"-output={fatbin}"])
    run([HIPCC] + FLAGS + list(extra) + [cuid, "--cuda-host-only", "-Wno-unused-command-line-argument", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin,
                                         "-c", "-o", obj, src])
    for f in (dev_o, hsaco, fatbin):
        os.unlink(f)
    return repaired


def audit(lib):
    """(FLOW blocks with the EXEC-flip fault in the kernels embedded in `lib`, report) — by the library itself, in a child process
    (this process may go on to import torch, whose libamdhip64 must be the first one loaded: _abi.load)"""
    code = ("import ctypes, sys; lib = ctypes.CDLL(sys.argv[1]); n = ctypes.c_int32(-1); buf = ctypes.create_string_buffer(8192);"
            "rc = lib.rtgr_code_object_audit(sys.argv[1].encode(), ctypes.byref(n), buf, 8192);"
            "print(rc, n.value); print(buf.value.decode())")
    r = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True)
    head = r.stdout.split("\n", 1)
    try:
        rc, n = (int(v) for v in head[0].split())
    except ValueError:
        return None, r.stderr[-2000:]          # could not be run (no libamdhip64 / libamd_comgr here): the tests audit again
    return (n if rc == 0 else None), (head[1] if len(head) > 1 else "")


def build(force=False, extra=(), verbose=True, out=OUT, obj_dir=OBJ, via_listing=None):
    if via_listing is None:
        via_listing = os.environ.get("RTGR_BUILD_VIA_LISTING") == "1"
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
        if via_listing and os.path.basename(src) not in HOST_ONLY_UNITS:
            try:
                compile_via_listing(src, obj, extra, verbose, cwd=obj_dir)
                return job, subprocess.CompletedProcess([], 0, "", "")
            except RuntimeError as e:
                return job, subprocess.CompletedProcess([], 1, "", str(e))
        cmd = [HIPCC] + FLAGS + list(extra) + ["-c", "-o", obj, src]
        if os.path.basename(src) == "rtgr_units.hip":   # (depends on every header: rebuilt whenever the hash moves)
            cmd.insert(-4, f"-DRTGR_HEADER_HASH={header_hash():#x}ull")
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
    linked = open(tag_file).read().strip() if os.path.exists(tag_file) else None
    full_tag = tag + ("+listing" if via_listing else "")
    if jobs or force or _stale(out, objs) or linked != full_tag:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", out] + objs + ["-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=HERE)
        with open(tag_file, "w") as fh:
            fh.write(full_tag + "\n")
        # the kernels just linked in must be free of the compiler's EXEC-flip fault; if they are not, once more through the listings
        found, report = audit(out)
        if found is None:
            # the audit could not RUN (no libamdhip64 / libamd_comgr for the child, a crash): say so — an unaudited library must not
            # look like a clean one.  __graft_entry__.build() and tests/test_build_checks.py assert a real result.
            msg = f"WARNING: {out}: the post-link EXEC-flip audit could not be run ({report.strip()[-300:] or 'no output'}): kernels NOT audited"
            if os.environ.get("RTGR_REQUIRE_AUDIT") == "1":
                raise RuntimeError(msg)
            print(msg, file=sys.stderr, flush=True)
        if found:
            if via_listing:
                raise RuntimeError(f"{out}: {found} FLOW block(s) with vector instructions ahead of the EXEC flip survive the listing route:\n{report}")
            print(f"{out}: {found} FLOW block(s) with vector instructions ahead of the EXEC flip (DESIGN.md §4.6): rebuilding the kernel units "
                  f"through their listings\n{report}", flush=True)
            return build(force=True, extra=extra, verbose=verbose, out=out, obj_dir=obj_dir, via_listing=True)
    return out


def linked_tag(out=OUT):
    """'std' for the production flag set, a hash for an experiment build ('+listing' appended when the kernel units were built
    through their listings), None when unknown."""
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
    build(force=("--force" in sys.argv or "--resource-usage" in sys.argv or "--save-temps" in sys.argv or "--via-listing" in sys.argv), extra=extra,
          via_listing=True if "--via-listing" in sys.argv else None)
