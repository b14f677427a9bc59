"""world_size-2 run of the row-sharded driver on the gloo backend (CPU): row assignment (cyclic and contiguous), the
gather of unequal shares and rank-0 assembly.  No GPU here, so the oracle is injected as the slab tracer (test-only injection point of
sharded.trace_sharded); the gathered image must equal the single-process oracle image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, ni, nj, out_path, layout):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    import oracle_lib as O
    from scenes import example, rt
    from raytracegr_jl_amd import sharded
    sc, cam = example(2)
    opt = rt.solver_defaults()

    def oracle_rows(scene, o, camera, ni_, nj_, j0, jstride, nrows):
        rs = [O.trace(scene, o, ni_, nj_, j0=j, j1=j + 1, cam=camera, details=True, nthreads=2)
              for j in range(j0, j0 + nrows * jstride, jstride)]
        ctr = np.zeros(8, np.int64)
        if not rs:   # a rank without rows (nj < world size): empty arrays of the right shape ride the exchange
            return {"rgb": torch.zeros((3, 0), dtype=torch.float64), "status": torch.zeros(0, dtype=torch.uint8), "counters": torch.from_numpy(ctr)}
        for r in rs:
            ctr[:7] += [r["counters"][k] for k in ("rays", "accepted", "rejected", "rhs_evals", "events",
                                                   "events_interior", "not_finished")]
        return {"rgb": torch.from_numpy(np.concatenate([r["rgb"] for r in rs], axis=1)),
                "status": torch.from_numpy(np.concatenate([r["status"] for r in rs])),
                "counters": torch.from_numpy(ctr)}

    # RGB rows, status bytes and counters ride the exchange
    full = sharded.trace_sharded(sc, opt, cam, ni, nj, trace_rows=oracle_rows, layout=layout, with_status=True)
    if rank == 0:
        np.savez(out_path, rgb=full["rgb"].numpy(), status=full["status"].numpy(), counters=full["counters"].numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ws,ni,nj,layout", [(2, 24, 17, "cyclic"), (2, 16, 16, "cyclic"), (2, 24, 17, "slab"),
                                             (3, 12, 8, "cyclic"), (3, 8, 2, "cyclic"), (3, 8, 2, "slab")])   # (nj < ws: a rank without rows)
def test_sharded_gather_world_size_2(tmp_path, ws, ni, nj, layout):
    import oracle_lib as O
    from scenes import example, rt
    out = str(tmp_path / "img.npz")
    mp.spawn(_worker, args=(ws, _free_port(), ni, nj, out, layout), nprocs=ws, join=True)
    got = np.load(out)
    sc, cam = example(2)
    ref = O.trace(sc, rt.solver_defaults(), ni, nj, cam=cam, details=True)
    assert got["rgb"].shape == (3, ni * nj)
    assert np.array_equal(got["rgb"], ref["rgb"])
    assert np.array_equal(got["status"], ref["status"])
    c = ref["counters"]
    assert list(got["counters"][:7]) == [c[k] for k in ("rays", "accepted", "rejected", "rhs_evals", "events",
                                                        "events_interior", "not_finished")]


def test_slab_bounds_tile_the_frame():
    from conftest import load_package
    load_package()
    from raytracegr_jl_amd import sharded
    for nj in (1, 7, 8, 200, 4096, 4099):
        for ws in (1, 2, 3, 8):
            b = [sharded.slab_bounds(nj, ws, r) for r in range(ws)]
            assert b[0][0] == 0 and b[-1][1] == nj
            assert all(b[i][1] == b[i + 1][0] for i in range(ws - 1))
            sizes = [j1 - j0 for j0, j1 in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_row_assignments_cover_every_row_once():
    from conftest import load_package
    load_package()
    from raytracegr_jl_amd import sharded
    for layout in ("cyclic", "slab"):
        for nj in (1, 7, 8, 200, 4099):
            for ws in (1, 2, 3, 8):
                rows = []
                for r in range(ws):
                    j0, st, nr = sharded.row_assignment(nj, ws, r, layout)
                    rows += list(range(j0, j0 + nr * st, st))
                assert sorted(rows) == list(range(nj)), (layout, nj, ws)
