#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: geodesic RK (Tsit5) step attempts/s and rays/s on the Kerr–Schild
4096² screen (BASELINE.json metric / configs[2]), rows sharded over N GPUs of one node with one RCCL gather.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" of this bench is one full pass of the hot path over the whole screen (every ray integrated to its event and
coloured, slabs gathered on rank 0).  `value` = Tsit5 step attempts (accepted + rejected, each = 6 RHS evaluations,
SURVEY §8d) of the whole job per second; rays/s is reported beside it.  Inputs are generated on the device
(make_canvas fused into the kernel), so nothing crosses PCIe inside the timed region.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_RHS = 814           # algorithmic flop per RHS, lean generic-metric form
F_STEP = 6 * F_RHS + 520   # = 5404 flop per Tsit5 step attempt
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (½ of the
#                                157.3 TF fp32 vector figure in MI355X_MICROARCH.md)


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
    ap.add_argument("--rhs", default="closed", choices=["closed", "generic", "user"],
                    help="closed = Kerr-Schild-form contraction (production); generic = reference-style dual-number RHS "
                         "(RTGR_METRIC_GENERIC): its executed flops equal the algorithmic count of the roofline model; "
                         "user = the same metric typed as run-time compiled source (api.UserMetric; ks_true* variants)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--layout", default="cyclic", choices=["cyclic", "slab"], help="row distribution over ranks")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N>1: nccl (= RCCL over xGMI, the real thing) or gloo (CPU-staged; lets "
                         "the multi-rank logic be rehearsed with several ranks sharing one GPU)")
    return ap.parse_args()


def build_scene(rt, variant, generic=False):
    _, objs, cam = rt.example2_scene()
    if variant == "mink":
        metric, objs, cam = rt.example1_scene()
    else:
        metric = {"ks_ref0": rt.kerr_schild, "ks_ref08": rt.KerrSchild(1, 0.8, textbook=False),
                  "ks_true0": rt.KerrSchild(1, 0.0), "ks_true08": rt.KerrSchild(1, 0.8),
                  "ks_true0998": rt.KerrSchild(1, 0.998), "ks_true0998_disk": rt.KerrSchild(1, 0.998)}[variant]
        if variant == "ks_true0998_disk":  # BASELINE config 5: thin accretion disk instead of the small sphere
            objs = objs[:2] + [rt.Disk(0.05, 2.0, 4.0)]  # camera (cylindrical radius 4.5) stays outside the disk
    if generic == "user":
        assert variant.startswith("ks_true"), "--rhs user: the example source is the textbook Kerr-Schild metric"
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        import user_metrics
        metric = rt.UserMetric(user_metrics.KERR_SCHILD, M=metric.M, a=metric.a)
        generic = False
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
    t0 = time.perf_counter()
    O.trace(scene, opt, 64, 64, cam=cam, details=False, nthreads=nthreads)  # warm + pilot
    pilot = time.perf_counter() - t0
    if sample <= 0:  # auto: about 15 s of CPU work, bounded
        sample = int(min(1536, max(64, 64 * (15.0 / max(pilot, 1e-3)) ** 0.5)))
    t0 = time.perf_counter()
    r = O.trace(scene, opt, sample, sample, cam=cam, details=False, nthreads=nthreads)
    dt = time.perf_counter() - t0
    c = r["counters"]
    attempts = c["accepted"] + c["rejected"]
    return {"value": attempts / dt, "unit": "RK step attempts/s", "cores": nthreads, "kind": "port",
            "sample": f"{sample}x{sample} screen of the same scene ({c['rays']} rays, {attempts} step attempts) "
                      f"in {dt:.2f} s; C++/OpenMP restatement of the reference algorithm (Julia absent)",
            "rays_per_s": c["rays"] / dt, "seconds": dt}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    rt = load_package()
    from raytracegr_jl_amd import sharded

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
    if ws > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    assert ws == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={ws}"
    lib = rt._abi.load()
    rt._abi.check(lib, lib.rtgr_init(local))

    npdt = np.float64 if a.dtype == "f64" else np.float32
    scene, cam = build_scene(rt, a.variant, {"closed": False, "generic": True, "user": "user"}[a.rhs])
    opt = rt.solver_defaults(npdt)
    ni = nj = a.size
    # rows are dealt cyclically (rank r: rows r, r+N, …): contiguous slabs of a black-hole image are unbalanced
    j0, jstride, nrows = sharded.row_assignment(nj, ws, rank, a.layout)
    ctr = torch.zeros(8, dtype=torch.int64, device=dev)
    out = {}

    def one_pass(timed_events=None):
        if timed_events is not None:
            timed_events[0].record()
        sharded.trace_rows_torch(scene, opt, cam, ni, nj, j0, jstride, nrows, device=dev, dtype=npdt, counters=ctr,
                                 out=out)
        if timed_events is not None:
            timed_events[1].record()
        if ws > 1 and not a.no_gather:
            gather(out["rgb"])

    nmax = ni * max(sharded.row_assignment(nj, ws, r, a.layout)[2] for r in range(ws))
    parts = None
    image = None

    def gather(slab):
        nonlocal parts
        send = slab if slab.shape[1] == nmax else torch.cat([slab, slab.new_zeros((3, nmax - slab.shape[1]))], 1)
        send = send.contiguous().to(cdev)
        if rank == 0:
            if parts is None:
                parts = [torch.empty((3, nmax), dtype=slab.dtype, device=cdev) for _ in range(ws)]
            dist.gather(send, parts, dst=0)
            nonlocal image
            image = sharded.assemble_rows(parts, ni, nj, ws, a.layout)  # the gathered frame, rows back in place
        else:
            dist.gather(send, None, dst=0)

    for _ in range(a.warmup):
        one_pass()
    torch.cuda.synchronize()
    ctr.zero_()
    rt._abi.check(lib, lib.rtgr_timing_enable(None, 0, 1))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        one_pass(evs[k])
    torch.cuda.synchronize()
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = [e0.elapsed_time(e1) for e0, e1 in evs]  # whole pipeline (3 kernels) per pass, torch events

    tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
    totals = ctr.clone().to(cdev)
    kmax = torch.tensor([max(kern_ms)], dtype=torch.float64, device=cdev)
    if ws > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    dt = float(tt[0])
    rays, acc, rej, nrhs = (int(totals[i]) for i in range(4))
    attempts = acc + rej

    # per-kernel HIP-event timing recorded by the library on the launch stream (rtgr_timing_*)
    import ctypes
    kms = (ctypes.c_double * 4)()
    kln = (ctypes.c_uint64 * 4)()
    rt._abi.check(lib, lib.rtgr_timing_read(None, 0, ctypes.byref(kms), ctypes.byref(kln)))
    if rank == 0:
        # roofline of the dominant kernel — integrate_kernel's main (FAR) pass, ~93 % of device time — over this rank's
        # launches: algorithmic flop per launch / average launch duration (HIP events around the kernel itself).
        # The few steps per ray that the NEAR pass redoes/finishes are attributed to the NEAR pass's own time below;
        # charging ALL step attempts to the FAR kernel's time alone would overstate it, so the denominator is
        # FAR + NEAR (the two launches of the same kernel template that together perform the counted attempts).
        my = ctr.cpu().numpy()
        n_launch = max(int(kln[1]), 1)
        my_attempts, my_rays = int(my[1] + my[2]) / n_launch, int(my[0]) / n_launch
        k_avg_s = (float(kms[1]) + float(kms[3])) / n_launch * 1e-3
        flop_launch = my_attempts * F_STEP + 2 * my_rays * F_RHS
        achieved = flop_launch / k_avg_s / 1e12
        # fp32 runs (--dtype f32) are priced against the packed-f32 vector peak (v_pk_fma_f32: two FMAs per lane)
        peak = FP64_VALU_PEAK_TFLOPS if a.dtype == "f64" else 2 * FP64_VALU_PEAK_TFLOPS
        roof = {"bound": "valu_f64" if a.dtype == "f64" else "valu_f32", "achieved": achieved, "peak": peak,
                "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                "kernel": "rtgr::integrate_kernel (FAR pass + NEAR pass)", "kernel_ms_avg": k_avg_s * 1e3,
                "launches": n_launch, "far_pass_ms_avg": float(kms[1]) / n_launch,
                "near_pass_ms_avg": float(kms[3]) / max(int(kln[3]), 1),
                "algorithmic_flop_per_launch": flop_launch,
                "flop_model": f"{F_STEP} flop/step attempt + 2x{F_RHS} per ray (SURVEY 8d: reference-formulation "
                              f"work; the kernel's closed Kerr-Schild contraction executes fewer — see DESIGN.md "
                              f"and profiles/ for the hardware-counted f64 flops and VALU utilisation)",
                "other_kernels_ms_avg": {"canvas": float(kms[0]) / max(int(kln[0]), 1),
                                         "resolve": float(kms[2]) / max(int(kln[2]), 1)},
                "hbm_algorithmic_GBps": (my_rays * (132 + 2 * 140 + 204 + 25)) / k_avg_s / 1e9}
        roof["traffic"] = load_traffic(a)
        roof["hardware_counted"] = load_traffic(a, key="hardware_counted")  # ALU-side truth next to the algorithmic figure
        name = C_name(lib)
        cpu = cpu_baseline(rt, scene, cam, opt, a.cpu_sample) if (a.cpu_sample != 0 and ws == 1) else None  # N=1 only
        line = {
            "metric": "geodesic RK step attempts/s (Tsit5, 6 RHS each), Kerr-Schild screen, whole job",
            "value": attempts / dt, "unit": "RK step attempts/s", "n_gpus": ws, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"example2 scene (Kerr-Schild {a.variant}, 3 objects) {ni}x{nj} screen, "
                                   f"tol=eps^(3/4), lambda in [0,100]; rows dealt {a.layout} over {ws} GPU(s)"
                                   f"{'' if ws == 1 or a.no_gather else ' + ' + ('RCCL' if a.backend == 'nccl' else 'gloo') + ' gather to rank 0'}",
                       "size": a.size, "variant": a.variant, "rhs": a.rhs, "parallelism": f"rows/{ws}"},
            "rays_per_s": rays / dt, "rays": rays // a.steps, "step_attempts_per_pass": attempts // a.steps,
            "accepted": acc // a.steps, "rejected": rej // a.steps, "rhs_evals_per_pass": nrhs // a.steps,
            "pipeline_ms_max_over_ranks": float(kmax[0]), "device": name,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if ws > 1:
        dist.barrier()
        dist.destroy_process_group()


def load_traffic(a, key="traffic_bytes_per_launch"):
    """HBM bytes per launch of the dominant kernel (or another recorded PMC-derived entry) from the latest committed
    profile of the SAME configuration (profiles/rNN/traffic.json; PMC counters cannot be collected from inside the
    bench), else None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        c = t.get("config", {})
        if (c.get("size"), c.get("variant"), c.get("dtype")) == (a.size, a.variant, a.dtype) and a.gpus == 1 \
                and a.rhs == "closed":
            return t.get(key)
    return None


def C_name(lib):
    import ctypes
    buf = ctypes.create_string_buffer(128)
    cu, mhz, wf = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    lib.rtgr_device_info(None, 0, buf, 128, ctypes.byref(cu), ctypes.byref(mhz), ctypes.byref(wf))
    return f"{buf.value.decode()} {cu.value} CU @ {mhz.value} MHz"


if __name__ == "__main__":
    main()
