#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: geodesic RK (Tsit5) step attempts/s and rays/s on the Kerr–Schild
4096² screen (BASELINE.json metric / configs[2]), rows sharded over N GPUs of one node with one RCCL gather.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" of this bench is one full pass of the hot path over the whole screen (every ray integrated to its event and
coloured, rows — RGB and status bytes — gathered on rank 0).  `value` = Tsit5 step attempts (accepted + rejected, each =
6 RHS evaluations) of the whole job per second; rays/s is reported beside it.  Inputs are generated on the
device (make_canvas fused into the set-up kernel), so nothing crosses PCIe inside the timed region (`--entry device`).

`roofline` (dominant kernels: the integrate kernel's FAR + NEAR passes) is an EXECUTED-flop figure:
    achieved = executed f64 flop per step attempt (hardware-counted: 64 x (2 FMA + MUL + ADD) wave-instructions)
               x the step attempts THIS run counted / the integrate kernels' time THIS run measured with HIP events on the launch stream
    frac     = achieved / 78.6 TF/s  (fp64 vector peak)  — always <= 1
    traffic  = HBM bytes per pass of the whole pipeline (2 x FETCH_SIZE + WRITE_SIZE)
The counters are read BY THIS RUN on this box (`roofline.counters: "live"`; default run at N = 1): after the timed region one frame of
the same workload is run three times under `rocprofv3 --pmc` in child processes — arithmetic counters, FETCH_SIZE, WRITE_SIZE, each
set in a run of its own, never with a trace (live_counters()).  The committed profile of the same kernel sources
(profiles/rNN/flops.json, keyed by the sha256 of the device headers + flags, raytracegr.jl_amd/build.py:kernel_source_hash) is
then the cross-check (`live_over_profile`), and the source when the live passes are off (--extras 0 / --live-counters 0, N > 1)
or fail; with neither, achieved / frac / traffic are null and `stale_profile` / `live_counters_skipped` say why.
The reference-formulation figure of SURVEY §8d (5404 flop per attempt, what Julia's dual-number chain would execute for
the same steps) is reported separately as `reference_equivalent_tflops` and may exceed the hardware peak: the closed
Kerr–Schild contraction does not execute that work.

Rank 0 prints ONE JSON line.
"""
import argparse
import glob
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_RHS = 814           # algorithmic flop per RHS, lean generic-metric form
F_STEP = 6 * F_RHS + 520   # = 5404 flop per Tsit5 step attempt: the REFERENCE FORMULATION's work
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (½ of the
#                                157.3 TF fp32 vector figure in MI355X_MICROARCH.md)
# Float32 kernels issue SCALAR v_fma_f32 (one ray per lane), which issues at the f64 rate on gfx950 (measured 0.9 of a
# v_fma_f64 slot, tools/micro/valu_rates.hip); the 157.3 TF/s fp32 vector peak needs v_pk_fma_f32 = two rays per lane.
F32_SCALAR_VALU_PEAK_TFLOPS = 78.6
F32_PACKED_VALU_PEAK_TFLOPS = 157.3   # fp32 vector peak (MI355X_MICROARCH.md): v_pk_fma_f32, two f32 operations per lane per slot


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096, help="screen is size x size")
    ap.add_argument("--variant", default="ks_ref0",
                    choices=["ks_ref0", "ks_ref08", "ks_true0", "ks_true08", "ks_true0998", "ks_true0998_disk", "mink"])
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="CPU baseline renders a sample x sample screen; -1 = auto (~15 s of CPU work), 0 = off")
    ap.add_argument("--rhs", default="closed", choices=["closed", "generic", "user", "user_ks"],
                    help="closed = Kerr-Schild-form contraction (production); generic = reference-style dual-number RHS "
                         "(RTGR_METRIC_GENERIC): its executed flops are the reference formulation's; "
                         "user = the same metric typed as run-time compiled source (api.UserMetric; ks_true* variants); "
                         "user_ks = typed in Kerr-Schild form (f and k only: rtgr_user_ks)")
    ap.add_argument("--ctx-devices", type=int, default=0,
                    help="entries host / pixels / sharded: run on an explicit rtgr_context of this many devices (device k of the "
                         "context = visible GPU k mod #visible, so on a one-GPU box the same GPU is listed N times: a rehearsal "
                         "of the code path, not of the speed).  0 = the default context (host, pixels) / every visible GPU (sharded)")
    ap.add_argument("--entry", default="device", choices=["device", "host", "pixels", "sharded"],
                    help="which C-ABI entry point the timed passes go through: device = rtgr_trace_rows_device (inputs and "
                         "outputs resident in HBM: the headline), host = rtgr_trace (camera on the device, RGB planes to "
                         "host memory), pixels = rtgr_trace_pixels_f64 (the reference's Array{Pixel} in and out over "
                         "PCIe — what a Julia ccall binds; with --ctx-devices N the call deals the rows to N devices), "
                         "sharded = rtgr_trace_sharded_device_f64: ONE process drives every device of the context, rows "
                         "gathered on device 0 by peer copies — the single-process twin of the torchrun form, same "
                         "frame_checksum.  host / pixels / sharded: one process (--gpus 1 under no launcher)")
    ap.add_argument("--extras", type=int, default=1,
                    help="1: after the timed region (rank 0, N = 1, default workload) also time the host and pixels entry "
                         "points and the Kerr a = 0.8 variant, reported in entry_points / variants; 0: skip")
    ap.add_argument("--live-counters", type=int, default=-1,
                    help="-1 (default): as --extras.  1: after the timed region (rank 0, N = 1, device entry) run ONE pass of the same workload three times under "
                         "`rocprofv3 --pmc` in child processes (f64 / f32 FMA-MUL-ADD counters; FETCH_SIZE; WRITE_SIZE — each set in a "
                         "run of its own) and compute the roofline's executed flops and HBM traffic from THIS box's counters; the "
                         "hash-keyed profile of profiles/rNN/flops.json is then the cross-check, and the fallback when a pass fails "
                         "or rocprofv3 is not there.  0: the profile only")
    ap.add_argument("--emit-row-checksums", action="store_true",
                    help="put the per-image-row checksums of the delivered frame on the line (nj integers): what tools/merge_flops.py "
                         "records as the N = 1 reference a mismatch at N > 1 is attributed to rows — and ranks — with")
    ap.add_argument("--checksum-reference", default="",
                    help="a file holding the JSON line of an N = 1 run with --emit-row-checksums of the SAME configuration: the reference the "
                         "delivered frame is checked against instead of the one recorded under profiles/ (rehearsals at sizes that have none)")
    ap.add_argument("--user-sphere", action="store_true",
                    help="example2's small sphere as a USER-DEFINED object of the same geometry (examples/user_objects.py "
                         "SPHERE_AS_USER_OBJECT through a run-time unit): what variants.user_sphere_* times, as a workload of its own")
    ap.add_argument("--objects", type=int, default=3,
                    help="length of the object list: example2's three objects plus N - 3 small spheres on a spiral around the hole (the "
                         "reference's Vector{Object} has no length limit, src/RayTraceGR.jl:433-441; beyond 16 the list travels through "
                         "rtgr_scene.objects) — what variants.objects16 / objects64 time, as a workload of its own")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: one stream, trace and gather strictly in turn (no frames in flight)")
    ap.add_argument("--exchange-at-n1", action="store_true",
                    help="debug: run the N > 1 code path (process group, gather, frames in flight) with a one-rank group")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="frames traced concurrently on alternating streams (0 = auto: 1 at N = 1, 2 at N > 1)")
    ap.add_argument("--layout", default="cyclic", choices=["cyclic", "slab"], help="row distribution over ranks")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N>1: nccl (= RCCL over xGMI, the real thing) or gloo (CPU-staged; lets "
                         "the multi-rank logic be rehearsed with several ranks sharing one GPU)")
    a = ap.parse_args()
    if a.live_counters < 0:
        a.live_counters = 1 if a.extras else 0
    return a


_USER_SPHERE = {}


def spiral_spheres(n):
    """n small spheres (radius 0.2-0.45) on a spiral of radius 3-7.5 around the hole: the extra objects of --objects N"""
    out = []
    for k in range(n):
        t, rad = 0.7 * (k + 2), 3.0 + 4.5 * k / max(n, 1)
        out.append(((0.0, rad * math.cos(t), rad * math.sin(t) + 1.5, 1.2 * math.sin(2.3 * t)), 0.2 + 0.25 * abs(math.sin(1.1 * t))))
    return out


def build_scene(rt, variant, generic=False, user_sphere=False, nobj=3):
    """user_sphere: example2's small sphere typed as a USER object (examples/user_objects.py SPHERE_AS_USER_OBJECT: the same
    distance, colour rule and reach bound through the run-time unit's generic dispatch) — the frame is the same, the difference in
    time is what a user-defined Object costs against a built-in one.  nobj > 3: that many objects (spiral_spheres after example2's three)."""
    _, objs, cam = rt.example2_scene()
    if variant == "mink":
        metric, objs, cam = rt.example1_scene()
    else:
        metric = {"ks_ref0": rt.kerr_schild, "ks_ref08": rt.KerrSchild(1, 0.8, textbook=False),
                  "ks_true0": rt.KerrSchild(1, 0.0), "ks_true08": rt.KerrSchild(1, 0.8),
                  "ks_true0998": rt.KerrSchild(1, 0.998), "ks_true0998_disk": rt.KerrSchild(1, 0.998)}[variant]
        if variant == "ks_true0998_disk":  # BASELINE config 5: thin accretion disk instead of the small sphere
            objs = objs[:2] + [rt.Disk(0.05, 2.0, 4.0)]  # camera (cylindrical radius 4.5) stays outside the disk
    if generic in ("user", "user_ks"):
        assert variant.startswith("ks_true"), "--rhs user: the example source is the textbook Kerr-Schild metric"
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        import user_metrics
        src = user_metrics.KERR_SCHILD if generic == "user" else user_metrics.KERR_SCHILD_KS
        metric = rt.UserMetric(src, M=metric.M, a=metric.a, stationary=True)
        generic = False
    if user_sphere:
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        import user_objects
        fam = _USER_SPHERE.setdefault("fam", rt.UserObjects(user_objects.SPHERE_AS_USER_OBJECT, name="sphere as a user object"))
        objs = objs[:2] + [fam(0, objs[2]._pack())]
        if generic is True and variant != "mink":     # (the unit is built for the scene's metric variant)
            metric = rt.Metric(metric.kind, metric.M, metric.a, name=metric.__name__, generic=True)
            generic = False
    if nobj > 3:
        objs = list(objs) + [rt.Sphere(c, (1, 0, 0, 0), r) for c, r in spiral_spheres(nobj - 3)]
    sc = rt.make_scene(metric, objs)
    if generic and variant != "mink":
        sc.metric |= rt._abi.METRIC_GENERIC
    return sc, rt.make_camera(**cam)


def cpu_baseline(rt, scene, cam, opt, sample):
    """The CPU oracle (a restatement of the reference's algorithm — kind "port", NOT the Julia code) timed on this
    host's cores on a bounded sample of the same workload: the central sample x sample pixels' worth of rays of the
    same camera (a sample² screen of the same scene; step statistics are resolution independent)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    nthreads = int(O.lib().rtgr_oracle_num_threads())
    O.trace(scene, opt, 64, 64, cam=cam, details=False, nthreads=nthreads)  # warm (thread pool, page faults)
    t0 = time.perf_counter()
    O.trace(scene, opt, 256, 256, cam=cam, details=False, nthreads=nthreads)  # pilot: big enough to keep every thread busy
    pilot = time.perf_counter() - t0
    if sample <= 0:  # auto: about 8 s of CPU work (the whole leg, with the pilot, stays near 10 s), bounded
        sample = int(min(1024, max(128, 256 * (8.0 / max(pilot, 1e-3)) ** 0.5)))
    t0 = time.perf_counter()
    r = O.trace(scene, opt, sample, sample, cam=cam, details=False, nthreads=nthreads)
    dt = time.perf_counter() - t0
    c = r["counters"]
    attempts = c["accepted"] + c["rejected"]
    phys, logical = physical_cores()
    return {"value": attempts / dt, "unit": "RK step attempts/s", "cores": nthreads, "threads": nthreads,
            "physical_cores": phys, "logical_cpus": logical, "kind": "port",
            "sample": f"{sample}x{sample} screen of the same scene ({c['rays']} rays, {attempts} step attempts) "
                      f"in {dt:.2f} s on {nthreads} OpenMP threads ({phys} physical cores, {logical} logical CPUs visible); "
                      f"C++/OpenMP restatement of the reference algorithm (Julia absent)",
            "rays_per_s": c["rays"] / dt, "seconds": dt}


def physical_cores():
    """(physical cores, logical CPUs) this process may run on — `cores` of cpu_baseline is the THREAD count the oracle used
    (omp_get_max_threads), which on an SMT host is the logical count, not the cores."""
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        allowed = os.sched_getaffinity(0)
        seen, cur = set(), {}
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if cur:
                    if int(cur.get("processor", -1)) in allowed:
                        seen.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                    cur = {}
                continue
            k, v = line.split(":", 1)
            cur[k.strip()] = v.strip()
        if cur and int(cur.get("processor", -1)) in allowed:
            seen.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
        return (len(seen) or logical), logical
    except Exception:  # noqa: BLE001
        return logical, logical


def kernel_source_hash():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rtgr_build", os.path.join(ROOT, "raytracegr.jl_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.kernel_source_hash()


def load_profile(a):
    """The PMC-derived entry (executed flop per step attempt, HBM bytes per ray, instruction mix) of THIS configuration
    from the newest profiles/rNN/flops.json whose recorded kernel-source hash equals the current sources' — or (None, why).
    PMC counters cannot be collected from inside the bench; a profile of other sources must not be presented as this run's."""
    cur = kernel_source_hash()
    key = f"{a.variant}/{a.dtype}/{getattr(a, 'rhs_key', a.rhs)}"
    seen = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "flops.json")), reverse=True):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        seen.append((os.path.relpath(f, ROOT), t.get("kernel_source_hash")))
        if t.get("kernel_source_hash") == cur and key in t.get("entries", {}):
            e = dict(t["entries"][key])
            e["source"] = os.path.relpath(f, ROOT)
            e["kernel_source_hash"] = cur
            return e, None
    return None, f"no profiles/r*/flops.json entry {key!r} collected from kernel sources {cur} (found: {seen})"


_LIVE_BROKEN = None
_TRACE_BROKEN = False
_KILLED_PASSES = []   # profiler passes of this run that had to be killed (on the line: roofline.profiler_passes_killed)


def run_group(cmd, env, timeout, what):
    """subprocess.run for a profiler pass, in a process GROUP of its own: `rocprofv3 -- python bench.py` is three processes deep, and
    killing only rocprofv3 at the timeout left the profiled python running on the GPU beside the configurations timed next, with no
    note anywhere.  On timeout the whole group is killed and waited for, and the pass is recorded."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        p.communicate()
        _KILLED_PASSES.append(what)
        raise
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def live_counters(a, log=None, only_flop=False):
    """Executed flops and HBM bytes of ONE pass of this workload, hardware-counted here and now: bench.py re-run (one pass, no
    extras, no CPU leg) under `rocprofv3 --pmc` in child processes — the arithmetic counters, FETCH_SIZE and WRITE_SIZE each in a
    run of its own, never combined with a trace (MI355X_MICROARCH.md's recipe).  Returns (dict, None) or (None, why).  Children
    are started only from a process that is not itself profiled, and start nothing themselves (RTGR_NO_COMPILE=1, --live-counters 0)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    global _LIVE_BROKEN
    if _LIVE_BROKEN:      # one failed or hung pass ends the live counting of this run: the other configurations use their profiles
        return None, "skipped after: " + _LIVE_BROKEN
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler itself"
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return None, "rocprofv3 not found"
    if a.size * a.size > (1 << 26):
        return None, "more than one pipeline chunk per pass"
    sfx = "F64" if a.dtype == "f64" else "F32"
    # Float32: the FULL pass is the packed two-rays-per-lane kernel, and the per-kind counters count a v_pk_fma_f32 ONCE — half its
    # flops (calibrated: tools/micro/pk_counter_probe.hip, profiles/r05/pk_counter_probe.log: 4096 v_pk_fma_f32 -> FMA_F32 4096,
    # FLOPS_FP32 16386).  SQ_INSTS_VALU_FLOPS_FP32 counts flops per lane per instruction (2 / FMA, 1 / MUL or ADD, 4 / v_pk_fma_f32,
    # 2 / v_pk_mul or add): 64 x that is the executed flops, packed or not.
    flop_set = ([f"SQ_INSTS_VALU_FMA_{sfx}", f"SQ_INSTS_VALU_MUL_{sfx}", f"SQ_INSTS_VALU_ADD_{sfx}", "SQ_INSTS_VALU"] if a.dtype == "f64"
                else ["SQ_INSTS_VALU_FLOPS_FP32", "SQ_INSTS_VALU"])
    sets = {"flop": flop_set, "fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"]}
    if only_flop:      # (the objects16 / objects64 variants want the instruction counts only: one child instead of four)
        sets = {"flop": flop_set}
    work = tempfile.mkdtemp(prefix="rtgr_pmc_", dir="/tmp")
    tot, attempts, rays = {}, None, None
    t0 = time.time()
    try:
        for name, counters in sets.items():
            out = os.path.join(work, name)
            cmd = [tool, "--pmc", *counters, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--size", str(a.size), "--variant", a.variant, "--dtype", a.dtype, "--rhs", a.rhs, "--steps", "1", "--warmup", "0",
                   "--cpu-sample", "0", "--extras", "0", "--live-counters", "0", "--objects", str(getattr(a, "objects", 3))] + (["--user-sphere"] if getattr(a, "user_sphere", False) else [])
            # (RTGR_UNIT_PROBE=0: the parent process probed the unit when it loaded it; in the counted child the load-time probe's
            #  own rtgr_user_prepare* / integrate dispatches would be counted as passes of the workload — ADVICE r5)
            env = dict(os.environ, RTGR_NO_COMPILE="1", RTGR_UNIT_PROBE="0", TMPDIR="/tmp")
            r = run_group(cmd, env, 150, f"--pmc {' '.join(counters)} ({a.variant} {a.size} {a.dtype} {a.rhs})")
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                _LIVE_BROKEN = f"rocprofv3 --pmc {' '.join(counters)} failed (rc {r.returncode}): {r.stderr[-300:]}"
                return None, _LIVE_BROKEN
            child = json.loads(lines[-1])
            attempts, rays = child["step_attempts_per_pass"], child["rays"]
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"rocprofv3 wrote no counter_collection.csv for {counters}"
            part, passes = {}, 0
            for f in files:
                for row in csv.DictReader(open(f)):
                    kernel = row["Kernel_Name"]
                    # arithmetic: the integrate kernels (FAR + NEAR / FULL, built-in or of a run-time unit); bytes: every kernel of
                    # the library's pipeline (torch's own kernels of the bench harness are not the path's)
                    mine = "integrate" in kernel if name == "flop" else ("rtgr" in kernel)
                    if mine:
                        part[row["Counter_Name"]] = part.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    if "prepare" in kernel and row["Counter_Name"] == counters[0]:
                        passes += 1      # the set-up kernel runs once per pass (one pipeline chunk: checked below)
            if passes < 1:
                return None, f"no set-up kernel among the dispatches rocprofv3 recorded for {counters}"
            # … and the divisor must be what the child says it ran (its timed, warm-up and allocation passes): a set-up kernel of
            # anything else in the process — a probe, a scene check — would deflate every per-pass figure
            if child.get("passes_in_process") not in (None, passes):
                return None, f"{passes} set-up dispatches counted for {counters}, but the child ran {child['passes_in_process']} passes"
            for k, v in part.items():     # (the child runs its one timed pass plus the warm-up the bench always does)
                tot[k] = v / passes
    except subprocess.TimeoutExpired:
        _LIVE_BROKEN = "a rocprofv3 --pmc pass did not finish in 150 s"
        return None, _LIVE_BROKEN
    except Exception as e:  # noqa: BLE001  (a measurement aid must never cost the headline line)
        _LIVE_BROKEN = repr(e)
        return None, _LIVE_BROKEN
    finally:
        shutil.rmtree(work, ignore_errors=True)
    fma, mul, add = (tot.get(f"SQ_INSTS_VALU_{k}_{sfx}", 0.0) for k in ("FMA", "MUL", "ADD"))
    flops_lane = tot.get("SQ_INSTS_VALU_FLOPS_FP32", 0.0) if a.dtype == "f32" else (2 * fma + mul + add)
    if flops_lane <= 0 or not attempts:
        return None, f"counters came back empty: {tot}"
    trace = None if only_flop else kernel_trace_pass(a, tool)     # (best effort: the kernels' average durations as rocprofv3 itself reports them)
    return {"flop_per_step_attempt": 64.0 * flops_lane / attempts,
            "flop_counter": "64 x SQ_INSTS_VALU_FLOPS_FP32 (counts packed instructions in full)" if a.dtype == "f32" else "64 x (2 FMA + MUL + ADD) of SQ_INSTS_VALU_*_F64",
            "valu_per_wave_step": tot.get("SQ_INSTS_VALU", 0.0) / (attempts / 64.0),
            "fma_mul_add_per_wave_step": [fma / (attempts / 64.0), mul / (attempts / 64.0), add / (attempts / 64.0)] if a.dtype == "f64" else None,
            # FETCH_SIZE / WRITE_SIZE count KB; FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes on gfx950
            "hbm_bytes_per_ray": None if only_flop else (2 * tot.get("FETCH_SIZE", 0.0) + tot.get("WRITE_SIZE", 0.0)) * 1024.0 / rays,
            "hbm_note": "2 x FETCH_SIZE + WRITE_SIZE over the library's pipeline kernels (KB x 1024), per pass, per ray",
            "step_attempts_counted": attempts, "seconds": round(time.time() - t0, 1),
            "rocprof_kernel_trace": trace,
            "how": "rocprofv3 --pmc in child processes of this run: arithmetic counters, FETCH_SIZE, WRITE_SIZE each in a run of its own; "
                   "rocprofv3 --kernel-trace --stats in one more (never combined with --pmc)"}, None


def kernel_trace_pass(a, tool):
    """{kernel: {calls, avg_ms}} of the integrate kernels from `rocprofv3 --kernel-trace --stats` over three frames of this workload
    in a child process — the durations the profiler reports, to set beside the HIP-event times of the timed region — or None."""
    import csv
    import shutil
    import subprocess
    import tempfile
    global _TRACE_BROKEN
    if _TRACE_BROKEN:
        return None
    work = tempfile.mkdtemp(prefix="rtgr_trace_", dir="/tmp")
    try:
        cmd = [tool, "--kernel-trace", "--stats", "--output-format", "csv", "-d", work, "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--size", str(a.size), "--variant", a.variant, "--dtype", a.dtype, "--rhs", a.rhs, "--steps", "3", "--warmup", "1",
               "--cpu-sample", "0", "--extras", "0", "--live-counters", "0", "--objects", str(getattr(a, "objects", 3))] + (["--user-sphere"] if getattr(a, "user_sphere", False) else [])
        r = run_group(cmd, dict(os.environ, RTGR_NO_COMPILE="1", RTGR_UNIT_PROBE="0", TMPDIR="/tmp"), 150, f"--kernel-trace ({a.variant} {a.size} {a.dtype} {a.rhs})")
        if r.returncode != 0:
            return None
        out = {}
        for f in glob.glob(os.path.join(work, "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "integrate" in row["Name"]:
                    out[row["Name"].split("(")[0][-60:]] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6}
        return out or None
    except Exception:  # noqa: BLE001   (a hung or failed trace pass is not tried again in this run)
        _TRACE_BROKEN = True
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def expected_checksum(a):
    """(checksum, per-row checksums or None, where from) of the N = 1 frame of this configuration recorded from the CURRENT kernel
    sources, or (None, None, why).  The per-row vector (profiles/rNN/row_checksums.json) is what lets a mismatch name the ROWS — and
    through the row deal the rank or context device — that delivered other bits than the N = 1 frame."""
    cur = kernel_source_hash()
    key = f"{a.variant}/{a.dtype}/{getattr(a, 'rhs_key', a.rhs)}/{a.size}"
    if a.checksum_reference:
        ref = next(json.loads(l) for l in open(a.checksum_reference) if l.startswith("{"))
        c = ref["config"]
        assert (c["variant"], ref["dtype"], c["rhs"], c["size"]) == (a.variant, a.dtype, getattr(a, "rhs_key", a.rhs), a.size), "--checksum-reference: another configuration"
        return int(ref["frame_checksum"]), ref.get("row_checksums"), a.checksum_reference
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "flops.json")), reverse=True):
        try:
            t = json.load(open(f))
        except (OSError, ValueError):
            continue
        if t.get("kernel_source_hash") == cur and key in t.get("frame_checksums", {}):
            rows = None
            try:
                rj = json.load(open(os.path.join(os.path.dirname(f), "row_checksums.json")))
                if rj.get("kernel_source_hash") == cur:
                    rows = rj.get("rows", {}).get(key)
            except (OSError, ValueError):
                pass
            return int(t["frame_checksums"][key]), rows, os.path.relpath(f, ROOT)
    return None, None, f"no frame_checksums[{key!r}] recorded from kernel sources {cur}"


def main():
    a = parse()
    import ctypes
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    rt = load_package()
    from raytracegr_jl_amd import sharded
    abi = rt._abi

    failed, line = False, {}
    a.rhs_key = a.rhs + ("+user_sphere" if a.user_sphere else "") + (f"+objects{a.objects}" if a.objects != 3 else "")   # (profile / checksum key: another workload)
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)  # (gloo rehearsal: several ranks may share one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if a.backend == "nccl" else torch.device("cpu")  # where collective buffers live
    # the N > 1 code path (process group, exchange, frames in flight); --exchange-at-n1 runs it with a one-rank group, which
    # is how the RCCL calls themselves (not just the gloo-staged logic) are exercised on a one-GPU box
    multi = ws > 1 or a.exchange_at_n1
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if ws == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    assert a.entry == "device" or ws == 1, "--entry host / pixels / sharded: ONE process (it drives the devices itself)"
    if a.entry == "device":
        assert ws == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={ws}"
        if multi:   # the collective's own idea of the job must agree with the launcher's
            assert dist.get_world_size() == ws and dist.get_rank() == rank, (dist.get_world_size(), ws, dist.get_rank(), rank)
    lib = abi.load()
    abi.check(lib, lib.rtgr_init(local))
    # host / pixels / sharded on an explicit context (--ctx-devices N; sharded: every visible GPU by default)
    ctx, ctx_ids = None, [local]
    if a.entry != "device" and (a.ctx_devices > 0 or a.entry == "sharded"):
        nctx = a.ctx_devices if a.ctx_devices > 0 else max(ndev, 1)
        ctx_ids = [k % max(ndev, 1) for k in range(nctx)]
        ctx = abi.create_context(lib, ctx_ids)
        assert lib.rtgr_context_devices(ctx) == nctx
    n_physical = len(set(ctx_ids))
    if a.entry != "device":
        assert a.gpus in (1, n_physical), f"--gpus {a.gpus} but the context spans {n_physical} physical GPU(s)"

    if ws > 1:
        os.environ["RTGR_NO_COMPILE"] = "1"   # --rhs user on a cold cache: fail fast instead of N ranks starting hipcc
    npdt = np.float64 if a.dtype == "f64" else np.float32
    scene, cam = build_scene(rt, a.variant, {"closed": False, "generic": True, "user": "user", "user_ks": "user_ks"}[a.rhs], user_sphere=a.user_sphere, nobj=a.objects)
    opt = rt.solver_defaults(npdt)
    ni = nj = a.size
    # rows are dealt cyclically (rank r: rows r, r+N, …): contiguous slabs of a black-hole image are unbalanced
    j0, jstride, nrows = sharded.row_assignment(nj, ws, rank, a.layout)
    ctr = torch.zeros(8, dtype=torch.int64, device=dev)
    host = {}
    # N > 1: frames are delivered one after another, so two are kept in flight — (1) the exchange of pass k runs on a side
    # stream while pass k+1 is traced, and (2) the passes alternate between two streams (each has its own pipeline
    # workspace inside the library), so that the low-occupancy end of one frame's passes (a rank's share is ~10 rays per
    # lane) overlaps the start of the next: measured on one GPU with the N = 8 share, 12.44 -> 11.61 ms per frame, frames
    # bit-identical (tools/two_frames_in_flight.py; 0.4 % at the full frame, so N = 1 keeps one stream and clean kernel
    # timings).  All K traces and all K gathers are inside the timed region, which ends with a device-wide synchronize.
    # --no-overlap: one stream, trace and gather strictly in turn.
    overlap = multi and not a.no_gather and not a.no_overlap
    nflight = 1 if (a.no_overlap or a.entry != "device") else (a.in_flight if a.in_flight > 0 else (2 if multi else 1))
    nbuf = nflight + (1 if overlap else 0)   # one more output buffer than frames in flight: the exchange holds one
    outs = [{} for _ in range(nbuf)]
    traced = [None] * nbuf                 # event on the tracing stream: this buffer's frame is complete
    gathered = [None] * nbuf               # event on the exchange stream: this buffer's rows have left
    comm = torch.cuda.Stream(device=dev) if overlap else None
    lanes = [torch.cuda.Stream(device=dev) for _ in range(nflight)] if nflight > 1 else [None]
    npass = [0]
    corrupt_rank = int(os.environ.get("RTGR_BENCH_CORRUPT_RANK", "-1"))
    exch_events = []   # (start, end) events around every exchange, recorded on the stream the collective is enqueued on

    def timed_gather(rgb_t, status_t, stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        gather(rgb_t, status_t)
        e1.record(stream)
        exch_events.append((e0, e1))

    def device_pass():
        p = npass[0]
        npass[0] += 1
        b = p % nbuf
        out = outs[b]
        st = lanes[p % nflight] or torch.cuda.current_stream(dev)
        for ev in (traced[b], gathered[b]):   # the buffer's previous frame must have been traced and sent
            if ev is not None:
                st.wait_event(ev)
        with torch.cuda.stream(st):
            sharded.trace_rows_torch(scene, opt, cam, ni, nj, j0, jstride, nrows, device=dev, dtype=npdt, counters=ctr,
                                     out=out, status=multi)
            if corrupt_rank == rank and nrows > 1:
                # TEST HOOK (RTGR_BENCH_CORRUPT_RANK, tests/test_bench_gpu.py): this rank delivers one wrong row — its local row 1 —
                # so that the failure path of the line can be rehearsed: the line must still appear, name the row and the rank, rc 2
                out["rgb"][1, ni:2 * ni] = 0.125
            traced[b] = torch.cuda.Event()
            traced[b].record(st)
            if multi and not a.no_gather:
                if not overlap:
                    timed_gather(out["rgb"], out["status"], st)
                    return
                with torch.cuda.stream(comm):
                    comm.wait_event(traced[b])
                    timed_gather(out["rgb"], out["status"], comm)
                    gathered[b] = torch.cuda.Event()
                    gathered[b].record(comm)

    def host_pass():   # rtgr_trace_f64/_f32: camera on the device, RGB planes to (pageable) host memory
        if "rgb" not in host:
            host["rgb"] = np.empty((3, ni * nj), npdt)
        c = abi.rtgr_counters()
        fn = lib.rtgr_trace_f64 if a.dtype == "f64" else lib.rtgr_trace_f32
        abi.check(lib, fn(ctx, ctypes.byref(scene), ctypes.byref(opt), None, ctypes.byref(cam), ni, nj, 0, nj,
                          host["rgb"].ctypes.data, None, ctypes.byref(c)))
        return c

    def sharded_pass():   # rtgr_trace_sharded_device_*: one call, every device of the context, frame assembled on device 0
        if "d_rgb" not in host:
            d0 = torch.device("cuda", ctx_ids[0])
            host["d_rgb"] = torch.empty((3, ni * nj), dtype=torch.float64 if a.dtype == "f64" else torch.float32, device=d0)
            host["d_status"] = torch.empty(ni * nj, dtype=torch.uint8, device=d0)
        c = abi.rtgr_counters()
        o = abi.rtgr_ray_outputs()
        o.status = host["d_status"].data_ptr()
        fn = lib.rtgr_trace_sharded_device_f64 if a.dtype == "f64" else lib.rtgr_trace_sharded_device_f32
        abi.check(lib, fn(ctx, ctypes.byref(scene), ctypes.byref(opt), ctypes.byref(cam), ni, nj, host["d_rgb"].data_ptr(),
                          ctypes.byref(o), ctypes.byref(c)))
        return c

    def pixels_pass():  # rtgr_trace_pixels_f64: the reference's Array{Pixel{Float64},2} in, a new one out
        if "px" not in host:
            st = np.empty((ni * nj, 8))
            abi.check(lib, lib.rtgr_make_canvas_f64(ctx, ctypes.byref(scene), ctypes.byref(cam), ni, nj, 0, nj, st.ctypes.data))
            px = np.zeros(ni * nj, dtype=rt.pixel_dtype())
            px["pos"], px["normal"] = st[:, :4], st[:, 4:]
            host["px"], host["px_out"] = px, np.empty_like(px)
        c = abi.rtgr_counters()
        abi.check(lib, lib.rtgr_trace_pixels_f64(ctx, ctypes.byref(scene), ctypes.byref(opt), host["px"].ctypes.data, ni, nj,
                                                 host["px_out"].ctypes.data, ctypes.byref(c)))
        return c

    nmax = ni * max(sharded.row_assignment(nj, ws, r, a.layout)[2] for r in range(ws))
    parts = {}
    image = {}

    def gather(slab, status):
        """ONE exchange per pass: every rank's rows — RGB planes and status bytes — to rank 0 ("status bytes and
        counters ride the same gather"; the counters are summed after the timed region)."""
        for name, t, shape in (("rgb", slab, (3, nmax)), ("status", status, (nmax,))):
            send = t if t.shape[-1] == nmax else torch.cat([t, t.new_zeros(shape[:-1] + (nmax - t.shape[-1],))], -1)
            send = send.contiguous().to(cdev)
            if rank == 0:
                if name not in parts:
                    parts[name] = [torch.empty(shape, dtype=t.dtype, device=cdev) for _ in range(ws)]
                dist.gather(send, parts[name], dst=0)
            else:
                dist.gather(send, None, dst=0)
        if rank == 0:
            image["rgb"] = sharded.assemble_rows(parts["rgb"], ni, nj, ws, a.layout)  # the gathered frame, rows back in place
            image["status"] = sharded.assemble_rows([p[None] for p in parts["status"]], ni, nj, ws, a.layout)[0]

    def frames_call(K):
        """--entry host | pixels --in-flight 2: K frames through ONE call of rtgr_trace_frames_* — the library keeps two of them in
        flight on its two pipelines (what the device entry gets from two caller streams, for the callers of the blocking entry
        points).  Frames alternate between two output buffers (frames k and k + 2 are never in flight together)."""
        cs = (abi.rtgr_counters * K)()
        if a.entry == "host":
            if "rgb2" not in host:
                host["rgb2"] = [np.empty((3, ni * nj), npdt) for _ in range(2)]
            cams = (abi.rtgr_camera * K)(*[cam] * K)
            ptrs = (ctypes.c_void_p * K)(*[host["rgb2"][k % 2].ctypes.data for k in range(K)])
            fn = lib.rtgr_trace_frames_f64 if a.dtype == "f64" else lib.rtgr_trace_frames_f32
            abi.check(lib, fn(ctx, ctypes.byref(scene), ctypes.byref(opt), K, cams, None, ni, nj, ptrs, None, cs))
            host["rgb"] = host["rgb2"][(K - 1) % 2]
        else:
            if "px" not in host:
                pixels_pass()
            if "px_out2" not in host:
                host["px_out2"] = [np.empty_like(host["px"]) for _ in range(2)]
            pin = (ctypes.c_void_p * K)(*[host["px"].ctypes.data] * K)
            pout = (ctypes.c_void_p * K)(*[host["px_out2"][k % 2].ctypes.data for k in range(K)])
            abi.check(lib, lib.rtgr_trace_frames_pixels_f64(ctx, ctypes.byref(scene), ctypes.byref(opt), K, pin, ni, nj, pout, cs))
            host["px_out"] = host["px_out2"][(K - 1) % 2]
        return list(cs)

    frames_mode = a.entry in ("host", "pixels") and a.in_flight >= 2
    one_pass = {"device": device_pass, "host": host_pass, "pixels": pixels_pass, "sharded": sharded_pass}[a.entry]
    # untimed: W warm-up passes, and at least one pass through every stream / output buffer of the frames in flight, so that
    # no workspace is allocated inside the timed region whatever W is
    extra_warm = max(0, nbuf - a.warmup) if a.entry == "device" else 0
    if frames_mode:
        frames_call(max(2, a.warmup))     # (both pipelines allocate their staging and workspace here)
    else:
        for _ in range(a.warmup + extra_warm):
            one_pass()
    torch.cuda.synchronize()
    ctr.zero_()
    del exch_events[:]   # (the warm-up passes' exchanges are not the timed region's)
    for k in range(len(ctx_ids) if ctx else 1):
        abi.check(lib, lib.rtgr_timing_enable(ctx, k, 1))
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hc = []
    if frames_mode:
        hc = frames_call(a.steps)
    else:
        for k in range(a.steps):
            hc.append(one_pass())
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if a.entry != "device":   # the host entry points return their counters by value
        for c in hc:
            ctr += torch.tensor([c.rays, c.accepted, c.rejected, c.rhs_evals, c.events, c.events_interior,
                                 c.not_finished, 0], dtype=torch.int64, device=dev)

    my_wall_ms = dt * 1e3
    tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
    totals = ctr.clone().to(cdev)
    if multi:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    dt = float(tt[0])
    rays, acc, rej, nrhs = (int(totals[i]) for i in range(4))
    attempts = acc + rej

    # per-kernel HIP-event timing recorded by the library on the launch stream (rtgr_timing_*)
    kms = (ctypes.c_double * 4)()
    kln = (ctypes.c_uint64 * 4)()
    abi.check(lib, lib.rtgr_timing_read(ctx, 0, ctypes.byref(kms), ctypes.byref(kln)))
    abi.check(lib, lib.rtgr_timing_enable(ctx, 0, 0))
    per_device_ms = None
    if ctx:   # a multi-device context: every device has its timers; the roofline below uses the BUSIEST device's
        per_device_ms = [[float(kms[w]) for w in range(4)]]
        for k in range(1, len(ctx_ids)):
            m2, l2 = (ctypes.c_double * 4)(), (ctypes.c_uint64 * 4)()
            abi.check(lib, lib.rtgr_timing_read(ctx, k, ctypes.byref(m2), ctypes.byref(l2)))
            abi.check(lib, lib.rtgr_timing_enable(ctx, k, 0))
            per_device_ms.append([float(m2[w]) for w in range(4)])
        busiest = max(per_device_ms, key=lambda v: v[1] + v[3])
        for w in range(4):
            kms[w] = busiest[w]
    # ---- N > 1: what every rank did, on rank 0's line (the first multi-GPU contact must be diagnosable from its one
    # line).  Kernel times are each rank's own HIP events on its launch stream; `exchange_ms` is the time between the two events
    # that bracket the rank's part in the collectives on the stream they are enqueued on (for RCCL: the collective kernels'
    # duration on that rank, waiting for the slowest peer included); `wall_ms` the rank's own timed region.
    exch_ms = sum(e0.elapsed_time(e1) for e0, e1 in exch_events) if exch_events else 0.0
    my_stats = torch.tensor([float(kms[0]), float(kms[1]), float(kms[2]), float(kms[3]), float(ctr[0]), float(ctr[1] + ctr[2]),
                             exch_ms, my_wall_ms, float(local)], dtype=torch.float64, device=cdev)
    all_stats = [my_stats]
    if multi:
        all_stats = [torch.empty_like(my_stats) for _ in range(ws)]
        dist.all_gather(all_stats, my_stats)
    per_rank = [{"rank": r, "device": int(v[8]), "setup_and_order_ms": float(v[0]) / a.steps, "far_ms": float(v[1]) / a.steps,
                 "resolve_ms": float(v[2]) / a.steps, "near_ms": float(v[3]) / a.steps, "rays": int(v[4]) // a.steps,
                 "step_attempts": int(v[5]) // a.steps, "exchange_ms": float(v[6]) / a.steps, "wall_ms": float(v[7]) / a.steps}
                for r, v in enumerate(t.cpu().tolist() for t in all_stats)]
    if rank == 0:
        # roofline of the dominant kernels — integrate_kernel's FAR pass (~93 % of device time) and NEAR pass, the two
        # launches of the same template that together perform the counted step attempts — over this rank's launches
        my = ctr.cpu().numpy()
        n_launch = max(int(kln[1]), 1)   # (host entry points launch once per pipeline chunk)
        my_attempts, my_rays = int(my[1] + my[2]), int(my[0])
        if ctx and len(ctx_ids) > 1:   # the call returns the counters summed over the context's devices: an even share
            my_attempts, my_rays = my_attempts // len(ctx_ids), my_rays // len(ctx_ids)
        # (… which is only approximately a device's own count — cyclic rows are not an exactly even share — and when the logical
        #  devices of the context time-share ONE GPU their kernel times overlap: no roofline on such lines.  ADVICE r3.)
        approx_roofline = bool(ctx) and len(ctx_ids) > 1
        k_s = (float(kms[1]) + float(kms[3])) * 1e-3   # seconds in the integrate kernels, all launches of this rank
        prof, why = load_profile(a)
        # … and the same two figures counted HERE, on this box, by this run (N = 1, device entry): the profile is then the cross-check
        live, why_not_live = (None, "off")
        if a.live_counters and not multi and a.entry == "device" and not ctx:
            live, why_not_live = live_counters(a)
        elif a.live_counters:
            why_not_live = "N = 1, device entry only"
        peak = FP64_VALU_PEAK_TFLOPS if a.dtype == "f64" else F32_SCALAR_VALU_PEAK_TFLOPS
        roof = {"bound": "valu_f64" if a.dtype == "f64" else "valu_f32_scalar", "achieved": None, "peak": peak,
                "unit": "TFLOP/s", "frac": None, "traffic": None,
                "kernel": "rtgr::integrate_kernel (FAR pass + NEAR pass)", "kernel_ms_per_pass": k_s * 1e3 / a.steps,
                "launches": n_launch, "far_pass_ms_per_pass": float(kms[1]) / a.steps,
                "near_pass_ms_per_pass": float(kms[3]) / a.steps,
                "other_kernels_ms_per_pass": {"setup_and_order": float(kms[0]) / a.steps, "resolve": float(kms[2]) / a.steps},
                "step_attempts_this_rank": my_attempts,
                "reference_equivalent_tflops": (my_attempts * F_STEP + 2 * my_rays * F_RHS) / k_s / 1e12,
                "reference_equivalent_model": f"{F_STEP} flop/step attempt + 2x{F_RHS} per ray (SURVEY 8d: what the "
                                              f"reference's dual-number formulation would execute for the same steps; NOT "
                                              f"a utilisation — the closed Kerr-Schild contraction executes fewer)",
                # <= 64 B in + 24 B out + 1 status byte per ray is what the ALGORITHM moves (round-4 review: this field
                # priced 641 B per ray, the record layout of an earlier round); what the pipeline's kernels really move is
                # hbm_measured_GBps below, from the FETCH_SIZE / WRITE_SIZE counters
                "hbm_algorithmic_GBps": my_rays * 89 / max(sum(float(kms[w]) for w in range(4)) * 1e-3, 1e-12) / 1e9}
        # SURVEY §8(d)'s own fraction, for the record: the reference FORMULATION's flops for the steps taken, over the time the
        # closed contraction needed for them.  It exceeds 1 on the closed-form kernels BECAUSE they do not execute the
        # dual-number work (parity-checked pointwise instead); `frac` is the executed-flop figure and the one to read.
        roof["contract_8d_frac"] = roof["reference_equivalent_tflops"] / peak
        roof["contract_8d_note"] = ("SURVEY 8d formula: (attempts x 5404 + 2 x rays x 814) / kernel time / peak; > 1 on the "
                                    "closed Kerr-Schild contraction because it elides the dual-number chain; not a utilisation")
        ALGORITHMIC_BYTES_PER_RAY = 89    # <= 64 B in + 24 B out + 1 status byte
        if approx_roofline:
            roof["approximate"] = ("multi-device context: counters are summed over the devices (an even share is assumed) and the "
                                   "kernel time is the busiest device's" + ("; the logical devices share one GPU, so achieved / frac "
                                   "are omitted" if n_physical < len(ctx_ids) else ""))
        src = live if live is not None else prof
        roof["counters"] = "live" if live is not None else ("profile" if prof is not None else None)
        if live is None and a.live_counters:
            roof["live_counters_skipped"] = why_not_live
        if src is not None and not (approx_roofline and n_physical < len(ctx_ids)):
            flop = src["flop_per_step_attempt"] * my_attempts
            roof["achieved"] = flop / k_s / 1e12
            roof["frac"] = roof["achieved"] / peak
            roof["executed_flop_per_step_attempt"] = src["flop_per_step_attempt"]
            roof["traffic"] = src["hbm_bytes_per_ray"] * my_rays / a.steps if src.get("hbm_bytes_per_ray") else None
            if src.get("hbm_bytes_per_ray"):   # what the pipeline's hand-over records cost over the algorithm's own bytes
                roof["traffic_over_algorithmic"] = src["hbm_bytes_per_ray"] / ALGORITHMIC_BYTES_PER_RAY
                # measured HBM bytes of ALL pipeline kernels (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction) over the time
                # all of them took: ~2 % of the 8 TB/s peak — the path is VALU-bound, the figure is here for completeness
                roof["hbm_measured_GBps"] = src["hbm_bytes_per_ray"] * my_rays / max(sum(float(kms[w]) for w in range(4)) * 1e-3, 1e-12) / 1e9
                roof["hbm_frac_of_peak"] = roof["hbm_measured_GBps"] / 8000.0
            if live is not None:
                roof["live"] = live
                tr = live.get("rocprof_kernel_trace")
                if tr:   # the profiler's own average durations of the two passes beside this run's HIP-event times (must agree)
                    prof_ms = sum(v["avg_ms"] for v in tr.values())
                    roof["rocprof_integrate_ms_per_pass"] = prof_ms
                    roof["event_over_rocprof"] = (k_s * 1e3 / a.steps) / prof_ms if prof_ms > 0 else None
                if prof is not None:   # the builder-collected profile of the same kernel sources, as the cross-check
                    roof["profile_flop_per_step_attempt"] = prof["flop_per_step_attempt"]
                    roof["live_over_profile"] = live["flop_per_step_attempt"] / prof["flop_per_step_attempt"]
        if prof is not None and not (approx_roofline and n_physical < len(ctx_ids)):
            roof["profile"] = prof
            ic = prof.get("issue_ceiling")
            if ic and ic.get("frac_of_peak_this_mix_can_issue"):   # what the machine can issue for THIS kernel's instruction mix with no operand ever waited for (measured)
                roof["issue_ceiling_frac"] = ic["frac_of_peak_this_mix_can_issue"]
                roof["frac_of_issue_ceiling"] = roof["frac"] / ic["frac_of_peak_this_mix_can_issue"]
        if prof is None:
            roof["stale_profile"] = why
        name = C_name(lib)
        extras = {}
        if a.extras and not multi and a.entry == "device" and a.rhs == "closed" and a.dtype == "f64" and not a.user_sphere and a.objects == 3:
            try:   # (outside the timed region; a failure here must not cost the headline line)
                extras = run_extras(a, rt, host_pass, pixels_pass, dt / a.steps)
            except Exception as e:  # noqa: BLE001
                extras = {"extras_error": repr(e)}
        cpu = cpu_baseline(rt, scene, cam, opt, a.cpu_sample) if (a.cpu_sample != 0 and not multi) else None  # N=1 only
        line = {
            "metric": "geodesic RK step attempts/s (Tsit5, 6 RHS each), Kerr-Schild screen, whole job",
            "value": attempts / dt, "unit": "RK step attempts/s", "n_gpus": ws if a.entry == "device" else n_physical, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"example2 scene (Kerr-Schild {a.variant}, 3 objects) {ni}x{nj} screen, "
                                   f"tol=eps^(3/4), lambda in [0,100]; rows dealt {a.layout} over "
                                   f"{ws if a.entry == 'device' else str(len(ctx_ids)) + ' context device(s) on ' + str(n_physical)} GPU(s)"
                                   f"{'' if ws == 1 or a.no_gather else ' + ' + ('RCCL' if a.backend == 'nccl' else 'gloo') + ' gather of RGB + status to rank 0'}",
                       "size": a.size, "variant": a.variant, "rhs": a.rhs_key, "entry": a.entry,
                       "parallelism": f"rows/{ws if a.entry == 'device' else len(ctx_ids)}"},
            "rays_per_s": rays / dt, "rays": rays // a.steps, "step_attempts_per_pass": attempts // a.steps,
            "passes_in_process": a.warmup + extra_warm + a.steps,
            "accepted": acc // a.steps, "rejected": rej // a.steps, "rhs_evals_per_pass": nrhs // a.steps,
            "device": name, "roofline": roof, "cpu_baseline": cpu,
        }
        line.update(extras)
        if multi and not a.no_gather:
            line["gathered_status_not_event"] = int((image["status"] != 0).sum())
            line["exchange"] = "overlapped with the next pass" if overlap else "in turn"
        if multi:
            line["world_size_checked"] = dist.get_world_size()
            line["per_rank"] = per_rank
            busy = [r["far_ms"] + r["near_ms"] for r in per_rank]
            line["rank_imbalance"] = {"integrate_ms_min": min(busy), "integrate_ms_max": max(busy),
                                      "max_over_mean": max(busy) / (sum(busy) / len(busy)) if sum(busy) else None,
                                      "exchange_ms_max": max(r["exchange_ms"] for r in per_rank),
                                      "note": "per pass; integrate = FAR + NEAR kernel time of the rank (HIP events on its launch stream); "
                                              "exchange = between the events that bracket the rank's gather calls on the stream they run on"}
        if a.entry == "sharded":   # the single-process form: peer-access table and the library's own exchange timers
            table, exch = [], []
            for k in range(len(ctx_ids)):
                why = ctypes.create_string_buffer(256)
                ok = lib.rtgr_peer_access(ctx, k, why, 256)
                table.append({"context_device": k, "hip_device": ctx_ids[k], "peer_access_with_device_0": int(ok),
                              "why_not": why.value.decode() or None})
                m2, l2 = (ctypes.c_double * 2)(), (ctypes.c_uint64 * 2)()
                abi.check(lib, lib.rtgr_timing_read_exchange(ctx, k, ctypes.byref(m2), ctypes.byref(l2)))
                exch.append({"context_device": k, "rows_out_ms": float(m2[0]) / a.steps, "copies": int(l2[0]) // max(a.steps, 1),
                             "place_rows_ms": float(m2[1]) / a.steps})
            line["peer_access"] = table
            line["exchange_per_device"] = exch
            line["exchange_note"] = ("rows_out_ms: the device's rows leaving it for device 0 (hipMemcpyPeerAsync, or the device -> pinned "
                                     "host leg where there is no peer access), HIP events on the source device's stream behind its "
                                     "trace; place_rows_ms (device 0): the kernels that put every rank's rows back into the frame")
        if ctx:
            line["ctx_devices"] = ctx_ids
            line["per_device_kernel_ms"] = [{"setup_and_order": v[0], "far": v[1], "resolve": v[2], "near": v[3]} for v in per_device_ms]
            if n_physical < len(ctx_ids):
                line["ctx_note"] = (f"{len(ctx_ids)} context devices on {n_physical} physical GPU(s): a rehearsal of the "
                                    "multi-device code path — the logical devices share one GPU, so this is not a scaling figure")
        line["frames_in_flight"] = 2 if frames_mode else nflight
        if frames_mode:
            line["frames_call"] = f"rtgr_trace_frames{'_pixels' if a.entry == 'pixels' else ''}_{a.dtype}: the {a.steps} timed frames in ONE blocking call, two in flight inside the library"
        if extra_warm:
            line["allocation_passes"] = extra_warm   # untimed passes beyond `warmup` (one per stream / output buffer)
        if nflight > 1:
            roof["note"] = ("kernel durations overlap between the frames in flight: kernel_ms_per_pass is a sum of stretched "
                            "durations and `achieved` an underestimate; the N = 1 line carries the clean figure")
        # order-independent bit-level checksum of the delivered frame: equal at every N iff the frames are bit-identical
        frame = None
        if a.entry == "device":
            frame = image["rgb"] if (multi and not a.no_gather) else (outs[0]["rgb"] if ws == 1 else None)
        elif a.entry == "sharded":
            frame = host["d_rgb"]
        elif a.entry == "host":
            frame = torch.from_numpy(host["rgb"])
        elif a.entry == "pixels":
            frame = torch.from_numpy(np.ascontiguousarray(host["px_out"]["rgb"].T))   # planes, like the other entries
        if frame is not None:
            bits = frame.contiguous().view(torch.int64 if a.dtype == "f64" else torch.int32).to(torch.int64)
            line["frame_checksum"] = int(bits.sum().item())
            # one checksum per IMAGE ROW (the unit the frame is dealt by): rgb planes are [3, ni * nj] with pixel i + j * ni
            rows = bits.view(3, nj, ni).sum(dim=(0, 2)).cpu().numpy().astype(np.int64)
            line["row_checksums_sha256"] = hashlib.sha256(rows.tobytes()).hexdigest()[:16]
            if a.emit_row_checksums:
                line["row_checksums"] = [int(v) for v in rows]
            # … which must equal the N = 1 device-entry frame's of the same configuration and kernel sources, recorded with the
            # round's profiles (profiles/rNN/flops.json "frame_checksums", row_checksums.json): checked at every N and through every
            # entry point.  A mismatch does NOT stop the line (round-4 review: an assert here killed rank 0 before it printed and left
            # the other ranks in the closing barrier): the line says which rows differ and whose they are, then the run exits 2.
            want, want_rows, src = expected_checksum(a)
            line["frame_checksum_expected"] = want
            line["frame_checksum_source"] = src
            if want is not None:
                line["frame_checksum_ok"] = bool(want == line["frame_checksum"])
                if want_rows is not None and len(want_rows) == nj:
                    bad = np.nonzero(rows != np.asarray(want_rows, dtype=np.int64))[0]
                    line["frame_checksum_ok"] = bool(line["frame_checksum_ok"] and bad.size == 0)
                    if bad.size:
                        nparts = ws if a.entry == "device" else len(ctx_ids)
                        owner = [int(r) for r in sorted({int(sharded.row_owner(nj, nparts, int(j), a.layout if a.entry == "device" else "cyclic")) for j in bad})]
                        line["frame_checksum_bad_rows"] = {"count": int(bad.size), "first": [int(j) for j in bad[:16]],
                                                           ("ranks" if a.entry == "device" else "context_devices"): owner,
                                                           "note": "image rows whose bits differ from the recorded N = 1 frame, and who traced them"}
                elif not line["frame_checksum_ok"]:
                    line["frame_checksum_bad_rows"] = {"note": "no per-row vector recorded for this configuration: the whole-frame checksum differs"}
                failed = not line["frame_checksum_ok"]
        if _KILLED_PASSES:
            roof["profiler_passes_killed"] = list(_KILLED_PASSES)   # (their whole process group was killed at the timeout: nothing left on the GPU)
        print(json.dumps(line), flush=True)
    if ctx:
        abi.check(lib, lib.rtgr_destroy(ctx))
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if failed:   # (rank 0 only: its line has been printed and every rank has passed the closing barrier)
        print(f"bench.py: the delivered frame differs from the N = 1 frame of the same kernel sources: {line.get('frame_checksum_bad_rows')}",
              file=sys.stderr, flush=True)
        sys.exit(2)


_LAST_FRAME = [None]


def time_variant(rt, variant, size, dtype="f64", rhs="closed", reps=3, warm=1, live_on=False, user_sphere=False, keep_frame=False, nobj=3):
    """One BASELINE configuration outside the headline's timed region: `reps` device-resident frames (camera on the device,
    nothing over PCIe), wall time + the library's HIP-event kernel times, and the same executed-flop roofline as the headline's,
    from counters read by this run (live_counters(); Float64 configurations) or THIS configuration's profile entry
    (profiles/rNN/flops.json, used only when its kernel-source hash is the current one)."""
    import ctypes
    import torch
    from raytracegr_jl_amd import sharded
    npdt = np.float64 if dtype == "f64" else np.float32
    sc, cam = build_scene(rt, variant, {"closed": False, "generic": True}[rhs], user_sphere=user_sphere, nobj=nobj)
    opt = rt.solver_defaults(npdt)
    ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    o = {}
    n = size
    lib = rt._abi.load()
    first_call_ms = None
    if user_sphere:
        # what the AUTOMATIC scene check costs: a scene whose user objects bring a reach bound is compared — FULL
        # pass against FAR + NEAR on a coarse sample of the call's rays — the first time it is traced.  One pass with the check
        # switched off (allocates the workspace, registers nothing), then the first checked pass, timed; the steady passes follow.
        with rt._abi.options(lib, scene_check=0):
            sharded.trace_rows_torch(sc, opt, cam, n, n, 0, 1, n, dtype=npdt, counters=ctr, out=o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sharded.trace_rows_torch(sc, opt, cam, n, n, 0, 1, n, dtype=npdt, counters=ctr, out=o)
        torch.cuda.synchronize()
        first_call_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(warm):
        sharded.trace_rows_torch(sc, opt, cam, n, n, 0, 1, n, dtype=npdt, counters=ctr, out=o)
    torch.cuda.synchronize()
    ctr.zero_()
    rt._abi.check(lib, lib.rtgr_timing_enable(None, 0, 1))
    t0 = time.perf_counter()
    for _ in range(reps):
        sharded.trace_rows_torch(sc, opt, cam, n, n, 0, 1, n, dtype=npdt, counters=ctr, out=o)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kms, kln = (ctypes.c_double * 4)(), (ctypes.c_uint64 * 4)()
    rt._abi.check(lib, lib.rtgr_timing_read(None, 0, ctypes.byref(kms), ctypes.byref(kln)))
    rt._abi.check(lib, lib.rtgr_timing_enable(None, 0, 0))
    att = (int(ctr[1]) + int(ctr[2])) / reps
    rays = n * n
    # A render loop delivers frame after frame: with TWO frames in flight — frame k on stream A, frame k + 1 on stream B, each stream
    # with the pipeline workspace the library keeps per stream — the thin end of one frame's passes overlaps the start of the next
    # (what bench.py does at N > 1).  Small frames only (it is worth 0.4 % at 4096²); throughput, not latency.
    two = None
    if n * n <= (1 << 22) and not user_sphere:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs2 = [{}, {}]
        ctr2 = torch.zeros(8, dtype=torch.int64, device="cuda")
        def frame(k):
            with torch.cuda.stream(streams[k % 2]):
                sharded.trace_rows_torch(sc, opt, cam, n, n, 0, 1, n, dtype=npdt, counters=ctr2, out=outs2[k % 2])
        for k in range(4):
            frame(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(2 * reps):
            frame(k)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / (2 * reps)
        two = {"ms_per_pass": dt2 * 1e3, "step_attempts_per_s": att / dt2, "over_single_stream": dt2 / dt,
               "same_frame": bool(torch.equal(outs2[0]["rgb"], o["rgb"]) and torch.equal(outs2[1]["rgb"], o["rgb"]))}
    bits = o["rgb"].contiguous().view(torch.int64 if dtype == "f64" else torch.int32).to(torch.int64)
    _LAST_FRAME[0] = o["rgb"].clone() if user_sphere or keep_frame else None
    v = {"workload": f"{variant}, same camera, {n}x{n}, {dtype}, rhs {rhs}" + (", small sphere as a USER object" if user_sphere else "") +
                     (f", {nobj} objects (example2's three + {nobj - 3} small spheres)" if nobj != 3 else ""),
         "size": n, "dtype": dtype, "rhs": rhs, "frame_checksum": int(bits.sum().item()),
         "ms_per_pass": dt * 1e3, "step_attempts_per_s": att / dt, "rays_per_s": rays / dt,
         "step_attempts_per_ray": att / rays, "rejected_per_pass": int(ctr[2]) // reps}
    if two is not None:
        v["two_frames_in_flight"] = two
    if first_call_ms is not None:
        v["first_checked_call_ms"] = first_call_ms
        v["automatic_scene_check_ms"] = first_call_ms - dt * 1e3
    k_s = (float(kms[1]) + float(kms[3])) * 1e-3 / reps
    class _A:  # noqa: E701
        pass
    _A.variant, _A.dtype, _A.rhs, _A.size, _A.objects = variant, dtype, rhs, size, nobj
    prof, why = load_profile(_A)
    if user_sphere:     # (the unit's kernels are other kernels than the profiled ones: timing only)
        prof, why, live_on = None, "a run-time unit's kernels: no profile entry", False
    if nobj != 3:       # (another workload than the profiled one: live counters or nothing)
        prof, why = None, "another object list than the profiled workload's"
    live, why_not_live = live_counters(_A, only_flop=(nobj != 3)) if live_on else (None, "off")
    # Float32: the packed two-rays-per-lane kernel is priced against the fp32 VECTOR peak (v_pk_fma_f32), as asked
    peak = FP64_VALU_PEAK_TFLOPS if dtype == "f64" else F32_PACKED_VALU_PEAK_TFLOPS
    r = {"bound": "valu_f64" if dtype == "f64" else "valu_f32_packed", "peak": peak, "unit": "TFLOP/s", "kernel_ms_per_pass": k_s * 1e3,
         "far_pass_ms_per_pass": float(kms[1]) / reps, "near_pass_ms_per_pass": float(kms[3]) / reps,
         "near_share_of_frame": (float(kms[3]) / reps) / (dt * 1e3),
         "other_kernels_ms_per_pass": {"setup_and_order": float(kms[0]) / reps, "resolve": float(kms[2]) / reps},
         "achieved": None, "frac": None,
         # SURVEY 8(d)'s own formula: a UTILISATION only where the kernel executes the reference formulation (rhs generic)
         "contract_8d_frac": (att * F_STEP + 2 * rays * F_RHS) / k_s / 1e12 / FP64_VALU_PEAK_TFLOPS if dtype == "f64" else None}
    src = live if live is not None else prof
    r["counters"] = "live" if live is not None else ("profile" if prof is not None else None)
    if live is None and live_on:
        r["live_counters_skipped"] = why_not_live
    if src is not None:
        r["achieved"] = src["flop_per_step_attempt"] * att / k_s / 1e12
        r["frac"] = r["achieved"] / peak
        if dtype == "f32":
            r["frac_of_scalar_issue_peak"] = r["achieved"] / F32_SCALAR_VALU_PEAK_TFLOPS
        r["executed_flop_per_step_attempt"] = src["flop_per_step_attempt"]
        r["valu_per_wave_step"] = live["valu_per_wave_step"] if live is not None else prof.get("per_wave_step", {}).get("valu")
        r["traffic"] = src["hbm_bytes_per_ray"] * rays if src.get("hbm_bytes_per_ray") else None
        if live is not None and prof is not None:
            r["live_over_profile"] = live["flop_per_step_attempt"] / prof["flop_per_step_attempt"]
    if prof is not None:
        r["valu_busy"] = prof.get("valu_busy")
        r["source"] = prof.get("source")
        ic = prof.get("issue_ceiling")
        if ic and ic.get("frac_of_peak_this_mix_can_issue"):
            r["issue_ceiling_frac"] = ic["frac_of_peak_this_mix_can_issue"]
    else:
        r["stale_profile"] = why
    v["roofline"] = r
    return v


def run_extras(a, rt, host_pass, pixels_pass, device_s):
    """N = 1, default workload only, OUTSIDE the timed region: (1) the same frame through the host-pointer entry points
    (rtgr_trace_f64: RGB planes to host; rtgr_trace_pixels_f64: the reference's Array{Pixel} in and out — PCIe-inclusive,
    never `value`); (2) EVERY BASELINE.json configuration on this GPU — C2 (1024², as written and Kerr a = 0.8), C3 as
    BASELINE words it (a = 0.8, 4096²), C4 (2048² Float32), C5 (8192², a = 0.998 + disk) — and the reference FORMULATION of the
    RHS (generic dual-number path) at 4096², the one kernel for which SURVEY 8(d)'s contract fraction is a utilisation.  Each with
    its own executed-flop roofline (live counters for the Float64 ones).  About 12 s of GPU time + the counter passes (the driver's line carries them)."""
    ex = {}
    ep = {"device_ms": device_s * 1e3}
    for name, fn in (("host", host_pass), ("pixels", pixels_pass)):
        fn()  # warm (allocates the pinned staging)
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            fn()
        ep[f"{name}_ms"] = (time.perf_counter() - t0) / reps * 1e3
    ep["pixels_over_device"] = ep["pixels_ms"] / ep["device_ms"]
    ep["note"] = ("host = rtgr_trace_f64 (camera on device, RGB planes -> pageable host memory); pixels = "
                  "rtgr_trace_pixels_f64 (88-byte Pixel array in and out of pageable host memory; 64 B/ray up, 24 B/ray "
                  "down over PCIe, H2D || integrate || D2H pipelined).  Wall time of blocking calls, PCIe-inclusive.")
    ex["entry_points"] = ep
    lib = rt._abi.load()
    rt._abi.check(lib, lib.rtgr_trim(None))   # the pinned staging of the host entries is not needed below; C5 wants the memory
    plan = (("c2_ks_ref0_1024", "ks_ref0", 1024, "f64", "closed", 20, 4),
            ("c2_ks_true08_1024", "ks_true08", 1024, "f64", "closed", 20, 4),
            ("ks_true08", "ks_true08", a.size, "f64", "closed", 3, 1),            # C3 as BASELINE.json words it
            ("c4_f32_ks_true08_2048", "ks_true08", 2048, "f32", "closed", 30, 6),
            ("c5_ks_true0998_disk_8192", "ks_true0998_disk", 8192, "f64", "closed", 2, 1),
            ("generic_ks_ref0_4096", "ks_ref0", 4096, "f64", "generic", 2, 1))
    ex["variants"] = {}
    for key, variant, size, dtype, rhs, reps, warm in plan:
        try:
            ex["variants"][key] = time_variant(rt, variant, size, dtype, rhs, reps, warm, live_on=bool(a.live_counters))
        except Exception as e:  # noqa: BLE001   (one configuration must not cost the others their place on the line)
            ex["variants"][key] = {"error": repr(e)}
        rt._abi.check(lib, lib.rtgr_trim(None))   # (the 14 GB workspace of C5 is not kept for the next configuration)
    # What a USER-DEFINED Object costs against a built-in one (round-4 review item 1): example2 at 2048² with its small sphere as the
    # built-in RTGR_SPHERE and as an RTGR_USER_OBJECT of the same geometry (the unit is built here with hipcc when it is not in the
    # cache — never under RTGR_NO_COMPILE; a failure costs only this entry).  Same frame; the ratio is the generic dispatch.
    try:
        if os.environ.get("RTGR_NO_COMPILE") == "1":
            raise RuntimeError("RTGR_NO_COMPILE=1: the unit is not built inside this run")
        bi = time_variant(rt, "ks_ref0", 2048, "f64", "closed", 10, 2, keep_frame=True)
        f_bi = _LAST_FRAME[0]
        us = time_variant(rt, "ks_ref0", 2048, "f64", "closed", 10, 2, user_sphere=True)
        f_us = _LAST_FRAME[0]
        _LAST_FRAME[0] = None
        ex["variants"]["user_sphere_ks_ref0_2048"] = {
            "workload": us["workload"], "ms_per_pass": us["ms_per_pass"], "builtin_ms_per_pass": bi["ms_per_pass"],
            "user_over_builtin": us["ms_per_pass"] / bi["ms_per_pass"],
            "far_near_ms": [us["roofline"]["far_pass_ms_per_pass"], us["roofline"]["near_pass_ms_per_pass"]],
            "builtin_far_near_ms": [bi["roofline"]["far_pass_ms_per_pass"], bi["roofline"]["near_pass_ms_per_pass"]],
            "same_frame": bool(us["frame_checksum"] == bi["frame_checksum"]),
            "pixels_that_differ": int(((f_us != f_bi).any(dim=0)).sum().item()), "max_rgb_difference": float((f_us - f_bi).abs().max().item()),
            "step_attempts_per_s": us["step_attempts_per_s"],
            "first_checked_call_ms": us.get("first_checked_call_ms"), "automatic_scene_check_ms": us.get("automatic_scene_check_ms"),
            "scene_check_note": "the first trace of a scene whose user objects bring a reach bound runs rtgr_scene_check's comparison on a coarse "
                                "sample of its rays (48 x 48 of the camera's canvas) before it is enqueued: first call = check + frame; every later call = frame",
            "note": "RTGR_USER_OBJECT through a run-time unit built for this scene's metric variant (rtgr_user_unit_compile): distance / objcolor / "
                    "reach bound of the source called from the unit's own set-up, FAR, NEAR and resolve kernels"}
    except Exception as e:  # noqa: BLE001
        ex["variants"]["user_sphere_ks_ref0_2048"] = {"error": repr(e)}
    # What a LONGER OBJECT LIST costs (round-5 review item 1: `objs::Vector{Object{T}}` has no length limit, src/RayTraceGR.jl:433-441):
    # example2 at 2048² with its three objects, with 16 (the kernels' argument block full), with 64 and with 256 (a device table; their
    # spheres in groups of neighbours whose bounding spheres the FAR pass's reach test asks first).  The slope is read
    # off the hardware's instruction counter (SQ_INSTS_VALU per wave-step, live) and off the pass times.
    try:
        base = time_variant(rt, "ks_ref0", 2048, "f64", "closed", 10, 2, live_on=False)
        if a.live_counters:    # (the three-object frame's instruction count at THIS size, one counter pass)
            class _B:  # noqa: E701
                pass
            _B.variant, _B.dtype, _B.rhs, _B.size, _B.objects = "ks_ref0", "f64", "closed", 2048, 3
            lv, _ = live_counters(_B, only_flop=True)
            if lv:
                base["roofline"]["valu_per_wave_step"], base["roofline"]["counters"] = lv["valu_per_wave_step"], "live"
        rows = {3: base}
        for nobj in (16, 64, 256):   # (256: time only — where the groups of a long list and the resolve kernel's selection decide)
            rows[nobj] = time_variant(rt, "ks_ref0", 2048, "f64", "closed", 6 if nobj == 16 else 3, 1,
                                      live_on=bool(a.live_counters) and nobj <= 64, nobj=nobj)
        for nobj in (16, 64, 256):
            r, b = rows[nobj], rows[3]
            e = {"workload": r["workload"], "ms_per_pass": r["ms_per_pass"], "three_objects_ms_per_pass": b["ms_per_pass"],
                 "over_three_objects": r["ms_per_pass"] / b["ms_per_pass"],
                 "far_near_ms": [r["roofline"]["far_pass_ms_per_pass"], r["roofline"]["near_pass_ms_per_pass"]],
                 "three_objects_far_near_ms": [b["roofline"]["far_pass_ms_per_pass"], b["roofline"]["near_pass_ms_per_pass"]],
                 "step_attempts_per_ray": r["step_attempts_per_ray"], "step_attempts_per_s": r["step_attempts_per_s"],
                 "frame_checksum": r["frame_checksum"]}
            vr, vb = r["roofline"].get("valu_per_wave_step"), b["roofline"].get("valu_per_wave_step")
            if vr and vb and r["roofline"].get("counters") == "live" and b["roofline"].get("counters") == "live":
                e["valu_per_wave_step"] = vr
                e["three_objects_valu_per_wave_step"] = vb
                e["valu_per_wave_step_per_additional_sphere"] = (vr - vb) / (nobj - 3)
                e["valu_note"] = ("SQ_INSTS_VALU of the integrate kernels (FAR + NEAR) per wave-step, counted live by this run; the slope is "
                                  "the executed cost of one more sphere per step, both passes together (DESIGN.md §4.7 has the static split)")
            ex["variants"][f"objects{nobj}_ks_ref0_2048"] = e
    except Exception as e:  # noqa: BLE001
        ex["variants"]["objects64_ks_ref0_2048"] = {"error": repr(e)}
    g = ex["variants"].get("generic_ks_ref0_4096", {}).get("roofline")
    if g:
        g["contract_8d_note"] = ("this kernel EXECUTES the reference formulation (4-wide duals through the metric, symmetric inverse, "
                                 "contract-then-raise), so SURVEY 8d's (attempts x 5404 + 2 x rays x 814) / kernel time / peak is a "
                                 "utilisation here; `frac` is the hardware-counted executed-flop figure of the same run")
    return ex


def C_name(lib):
    import ctypes
    buf = ctypes.create_string_buffer(128)
    cu, mhz, wf = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    lib.rtgr_device_info(None, 0, buf, 128, ctypes.byref(cu), ctypes.byref(mhz), ctypes.byref(wf))
    return f"{buf.value.decode()} {cu.value} CU @ {mhz.value} MHz"


if __name__ == "__main__":
    main()
