#!/usr/bin/env python3
"""Experiment: successive frames (one rank's share of a 4096² frame split N ways) on ONE stream vs alternating between TWO
streams (each stream has its own pipeline workspace inside the library), i.e. the tail of frame k's FAR pass and its NEAR /
resolve kernels overlapping the start of frame k+1.  Prints ms per frame.

    python tools/two_frames_in_flight.py [N=8] [variant=ks_ref0] [passes=24] [size=4096] [kernel timing on=0]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
from raytracegr_jl_amd import sharded  # noqa: E402
import bench  # noqa: E402

abi = rt._abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variant = sys.argv[2] if len(sys.argv) > 2 else "ks_ref0"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 24
SIZE = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
TIMING = int(sys.argv[5]) if len(sys.argv) > 5 else 0
sc, cam = bench.build_scene(rt, variant)
opt = rt.solver_defaults()
lib = abi.load()
abi.check(lib, lib.rtgr_init(-1))
ni = nj = SIZE
if TIMING:
    abi.check(lib, lib.rtgr_timing_enable(None, 0, 1))
j0, st, nr = sharded.row_assignment(nj, N, 0, "cyclic")
ctr = torch.zeros(8, dtype=torch.int64, device="cuda")


def run(nstreams):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    outs = [{} for _ in range(nstreams)]
    def passes(k):
        for p in range(k):
            s = p % nstreams
            with torch.cuda.stream(streams[s]):
                sharded.trace_rows_torch(sc, opt, cam, ni, nj, j0, st, nr, out=outs[s], counters=ctr)
    passes(2 * nstreams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    passes(K)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3, outs


ref = None
for ns in (1, 2, 3, 1, 2):
    ms, outs = run(ns)
    if ref is None:
        ref = outs[0]["rgb"].clone()
    same = all(torch.equal(o["rgb"], ref) for o in outs)
    print(f"N={N} share of {SIZE}² ({ni * nr / 1e6:.2f} M rays) {variant} timing={TIMING}: {ns} stream(s): {ms:.3f} ms per frame; frames bit-identical: {same}", flush=True)
