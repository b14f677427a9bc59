"""Row-sharded tracing across the GPUs of one node: one process per GPU (torch.distributed, backend "nccl" = RCCL),
image rows j dealt cyclically to the ranks (rank r traces rows r, r+N, …), one gather of the RGB rows to rank 0 over
xGMI.

torch is plumbing here (device memory, streams, the process group); the compute is librtgr_hip.so through
rtgr_trace_device_f64 with raw device pointers.

The reference has no counterpart (threads only, README.md:129-135); rows are the natural unit because a row is
contiguous in the column-major `pixels[i,j]` layout (src/RayTraceGR.jl:463-464).
"""
import ctypes as C

import numpy as np

from . import _abi


def slab_bounds(nj, world_size, rank):
    """Contiguous j-range [j0, j1) of `rank`; the first (nj % world_size) ranks get one extra row."""
    base, rem = divmod(nj, world_size)
    j0 = rank * base + min(rank, rem)
    return j0, j0 + base + (1 if rank < rem else 0)


def _torch_dtype(dtype):
    import torch
    return {np.float64: torch.float64, np.float32: torch.float32}[dtype]


def trace_slab_torch(scene, opt, cam, ni, nj, j0, j1, device="cuda", dtype=np.float64, details=False, counters=None,
                     out=None, state0=None, ctx=None, hit_only=False):
    """Trace rows [j0, j1) on `device` into torch tensors (device-resident in and out). Asynchronous: enqueues on
    torch's current stream.  Returns dict(rgb[3, n] (+ per-ray tensors when details))."""
    import torch
    lib = _abi.load()
    n = ni * (j1 - j0)
    td = _torch_dtype(dtype)
    dev = torch.device(device)
    with torch.cuda.device(dev):
        res = out if out is not None else {}
        if "rgb" not in res:
            res["rgb"] = torch.empty((3, n), dtype=td, device=dev)
        o = _abi.rtgr_ray_outputs()
        # the hit map (omin, :518-526) is one byte per ray; an object list beyond 255 needs the 32-bit output (rtgr_ray_outputs.hit32)
        wide = scene.nobj > 255
        hit_dtype, hit_field = (torch.int32, "hit32") if wide else (torch.uint8, "hit")
        if hit_only:  # big frames: the hit map is one byte per ray, the other per-ray outputs are 85
            if "hit" not in res:
                res["hit"] = torch.empty(n, dtype=hit_dtype, device=dev)
            setattr(o, hit_field, res["hit"].data_ptr())
        if details:
            if "state_end" not in res:
                res["state_end"] = torch.empty((n, 8), dtype=td, device=dev)
                res["lambda_end"] = torch.empty(n, dtype=td, device=dev)
                res["status"] = torch.empty(n, dtype=torch.uint8, device=dev)
                res["hit"] = torch.empty(n, dtype=hit_dtype, device=dev)
                res["n_accept"] = torch.empty(n, dtype=torch.int32, device=dev)
                res["n_reject"] = torch.empty(n, dtype=torch.int32, device=dev)
            for k in ("state_end", "lambda_end", "status", "n_accept", "n_reject"):
                setattr(o, k, res[k].data_ptr())
            setattr(o, hit_field, res["hit"].data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
        fn = lib.rtgr_trace_device_f64 if dtype == np.float64 else lib.rtgr_trace_device_f32
        s0 = None
        if state0 is not None:
            assert state0.is_cuda and state0.is_contiguous() and state0.shape == (n, 8) and state0.dtype == td
            s0 = state0.data_ptr()
        _abi.check(lib, fn(ctx, C.byref(scene), C.byref(opt), s0, C.byref(cam) if cam is not None else None, ni, nj, j0,
                           j1, res["rgb"].data_ptr(), C.byref(o),
                           counters.data_ptr() if counters is not None else None, stream))
    return res


def row_assignment(nj, world_size, rank, layout="cyclic"):
    """(j0, jstride, nrows) of `rank`.  "cyclic": rows rank, rank+N, … — every rank gets a statistically identical
    sample of the image, which matters because rows through the hole cost ~1.8x the edge rows (measured: contiguous
    slabs of example2 at N=8 take 10.9 … 19.2 ms).  "slab": the contiguous range of slab_bounds()."""
    if layout == "cyclic":
        return rank, world_size, len(range(rank, nj, world_size))
    j0, j1 = slab_bounds(nj, world_size, rank)
    return j0, 1, j1 - j0


def row_owner(nj, world_size, j, layout="cyclic"):
    """the rank (or context device) that traces image row j under row_assignment — what a per-row checksum mismatch is
    attributed to (bench.py)"""
    if layout == "cyclic":
        return j % world_size
    for r in range(world_size):
        j0, j1 = slab_bounds(nj, world_size, r)
        if j0 <= j < j1:
            return r
    raise ValueError(f"row {j} outside the canvas of {nj} rows")


def trace_rows_torch(scene, opt, cam, ni, nj, j0, jstride, nrows, device="cuda", dtype=np.float64, counters=None,
                     out=None, ctx=None, status=False):
    """Trace image rows j0, j0+jstride, … (nrows of them) into a device tensor rgb[3, ni*nrows]; asynchronous."""
    import torch
    lib = _abi.load()
    n = ni * nrows
    td = _torch_dtype(dtype)
    dev = torch.device(device)
    with torch.cuda.device(dev):
        res = out if out is not None else {}
        if "rgb" not in res:
            res["rgb"] = torch.empty((3, n), dtype=td, device=dev)
        o = None
        if status:  # per-ray status bytes (they ride the multi-GPU gather next to the RGB rows)
            if "status" not in res:
                res["status"] = torch.empty(n, dtype=torch.uint8, device=dev)
            o = _abi.rtgr_ray_outputs()
            o.status = res["status"].data_ptr()
        stream = torch.cuda.current_stream(dev).cuda_stream
        fn = lib.rtgr_trace_rows_device_f64 if dtype == np.float64 else lib.rtgr_trace_rows_device_f32
        _abi.check(lib, fn(ctx, C.byref(scene), C.byref(opt), C.byref(cam), ni, nj, j0, jstride, nrows,
                           res["rgb"].data_ptr(), C.byref(o) if o is not None else None,
                           counters.data_ptr() if counters is not None else None, stream))
    return res


def assemble_rows(parts, ni, nj, world_size, layout="cyclic"):
    """Rank-0 side of the gather: parts[r] = rgb[3, >= ni*nrows_r] of rank r  ->  full image rgb[3, ni*nj]."""
    planes = parts[0].shape[0]   # 3 for RGB, 1 for a per-ray array such as the status bytes
    full = parts[0].new_empty((planes, nj, ni))
    for r in range(world_size):
        j0, st, nr = row_assignment(nj, world_size, r, layout)
        full[:, j0:j0 + (nr - 1) * st + 1:st, :] = parts[r][:, :ni * nr].reshape(planes, nr, ni)
    return full.reshape(planes, nj * ni)


def trace_sharded(scene, opt, cam, ni, nj, group=None, device=None, dtype=np.float64, trace_rows=None,
                  gather=True, counters=None, layout="cyclic", with_status=False):
    """Every rank traces its rows; rank 0 receives the whole image [3, ni*nj] (None elsewhere).  with_status=True: the
    per-ray status bytes ride the same exchange and the counters are summed over the ranks — rank 0 gets
    dict(rgb[3, n], status[n], counters[8]).

    `trace_rows(scene, opt, cam, ni, nj, j0, jstride, nrows) -> tensor[3, ni*nrows]` (or, with_status,
    dict(rgb=…, status=…, counters=…)) is an injection point for the world_size-2 gloo tests on CPU-only hosts; the
    default is the HIP path (fails loudly without a GPU).
    """
    import torch
    import torch.distributed as dist
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    j0, st, nr = row_assignment(nj, ws, rank, layout)
    if trace_rows is None:
        if with_status and counters is None:
            counters = torch.zeros(8, dtype=torch.int64, device=device or "cuda")
        res = trace_rows_torch(scene, opt, cam, ni, nj, j0, st, nr, device=device or "cuda", dtype=dtype,
                               counters=counters, status=with_status)
        mine = {"rgb": res["rgb"], "status": res.get("status"), "counters": counters}
    else:
        mine = trace_rows(scene, opt, cam, ni, nj, j0, st, nr)
        if not isinstance(mine, dict):
            mine = {"rgb": mine}
    if ws == 1 or not gather:
        return mine if with_status else mine["rgb"]
    # gather of unequal shares: pad to the largest (row counts differ by at most one)
    nmax = ni * max(row_assignment(nj, ws, r, layout)[2] for r in range(ws))
    full = {}
    for name in (("rgb", "status") if with_status else ("rgb",)):
        t = mine[name] if name == "rgb" else mine[name][None]
        send = t if t.shape[1] == nmax else torch.cat([t, t.new_zeros((t.shape[0], nmax - t.shape[1]))], dim=1)
        send = send.contiguous()
        if rank == 0:
            parts = [torch.empty_like(send) for _ in range(ws)]
            dist.gather(send, parts, dst=0, group=group)
            full[name] = assemble_rows(parts, ni, nj, ws, layout)
        else:
            dist.gather(send, None, dst=0, group=group)
    if with_status:
        total = mine["counters"].clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        if rank == 0:
            return {"rgb": full["rgb"], "status": full["status"][0], "counters": total}
        return None
    return full.get("rgb")
