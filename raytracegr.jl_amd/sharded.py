"""Row-sharded tracing across the GPUs of one node: one process per GPU (torch.distributed, backend "nccl" = RCCL),
image rows j split into contiguous slabs, one gather of the RGB slabs to rank 0 over xGMI.

torch is plumbing here (device memory, streams, the process group); the compute is librtgr_hip.so through
rtgr_trace_device_f64 with raw device pointers.

The reference has no counterpart (threads only, README.md:129-135); the slab decomposition follows from the
column-major `pixels[i,j]` layout (src/RayTraceGR.jl:463-464): a j-slab is contiguous.
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
                     out=None, state0=None):
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
        if details:
            if "state_end" not in res:
                res["state_end"] = torch.empty((n, 8), dtype=td, device=dev)
                res["lambda_end"] = torch.empty(n, dtype=td, device=dev)
                res["status"] = torch.empty(n, dtype=torch.uint8, device=dev)
                res["hit"] = torch.empty(n, dtype=torch.uint8, device=dev)
                res["n_accept"] = torch.empty(n, dtype=torch.int32, device=dev)
                res["n_reject"] = torch.empty(n, dtype=torch.int32, device=dev)
            for k in ("state_end", "lambda_end", "status", "hit", "n_accept", "n_reject"):
                setattr(o, k, res[k].data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
        fn = lib.rtgr_trace_device_f64 if dtype == np.float64 else lib.rtgr_trace_device_f32
        s0 = None
        if state0 is not None:
            assert state0.is_cuda and state0.is_contiguous() and state0.shape == (n, 8) and state0.dtype == td
            s0 = state0.data_ptr()
        _abi.check(lib, fn(C.byref(scene), C.byref(opt), s0, C.byref(cam) if cam is not None else None, ni, nj, j0,
                           j1, res["rgb"].data_ptr(), C.byref(o),
                           counters.data_ptr() if counters is not None else None, stream))
    return res


def trace_sharded(scene, opt, cam, ni, nj, group=None, device=None, dtype=np.float64, trace_slab=None,
                  gather=True, counters=None):
    """Every rank traces its slab; rank 0 receives the whole image [3, ni*nj] (None elsewhere).

    `trace_slab(scene, opt, cam, ni, nj, j0, j1) -> tensor[3, n]` is an injection point for the world_size-2 gloo
    tests on CPU-only hosts; the default is the HIP path (fails loudly without a GPU).
    """
    import torch
    import torch.distributed as dist
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    j0, j1 = slab_bounds(nj, ws, rank)
    if trace_slab is None:
        slab = trace_slab_torch(scene, opt, cam, ni, nj, j0, j1, device=device or "cuda", dtype=dtype,
                                counters=counters)["rgb"]
    else:
        slab = trace_slab(scene, opt, cam, ni, nj, j0, j1)
    if ws == 1 or not gather:
        return slab
    # gather of unequal slabs: pad to the largest slab (rows differ by at most one)
    nmax = ni * (slab_bounds(nj, ws, 0)[1] - slab_bounds(nj, ws, 0)[0])
    send = slab if slab.shape[1] == nmax else torch.cat(
        [slab, slab.new_zeros((3, nmax - slab.shape[1]))], dim=1)
    send = send.contiguous()
    if rank == 0:
        parts = [torch.empty_like(send) for _ in range(ws)]
        dist.gather(send, parts, dst=0, group=group)
        full = slab.new_empty((3, ni * nj))
        for r in range(ws):
            a, b = slab_bounds(nj, ws, r)
            full[:, a * ni:b * ni] = parts[r][:, :ni * (b - a)]
        return full
    dist.gather(send, None, dst=0, group=group)
    return None
