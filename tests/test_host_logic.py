"""Host-side logic of the row split, property-tested on the CPU (no library calls): every row is traced exactly once for
every (rows, ranks, layout); the rank-0 assembly inverts the split for ragged shares; padded shares are ignored."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from conftest import load_package

rt = load_package()
from raytracegr_jl_amd import sharded  # noqa: E402


@settings(max_examples=200, deadline=None)
@given(nj=st.integers(1, 300), ws=st.integers(1, 16), layout=st.sampled_from(["cyclic", "slab"]))
def test_every_row_is_assigned_exactly_once(nj, ws, layout):
    seen = np.zeros(nj, np.int32)
    sizes = []
    for r in range(ws):
        j0, stride, nr = sharded.row_assignment(nj, ws, r, layout)
        rows = j0 + stride * np.arange(nr)
        assert nr == 0 or (rows.min() >= 0 and rows.max() < nj)
        seen[rows] += 1
        sizes.append(nr)
    assert (seen == 1).all()
    assert max(sizes) - min(sizes) <= 1          # shares differ by at most one row
    if layout == "slab":
        b = [sharded.slab_bounds(nj, ws, r) for r in range(ws)]
        assert b[0][0] == 0 and b[-1][1] == nj and all(b[k][1] == b[k + 1][0] for k in range(ws - 1))


@settings(max_examples=60, deadline=None)
@given(ni=st.integers(1, 9), nj=st.integers(1, 40), ws=st.integers(1, 8), layout=st.sampled_from(["cyclic", "slab"]),
       planes=st.sampled_from([1, 3]))
def test_assembly_inverts_the_split_with_padded_shares(ni, nj, ws, layout, planes):
    full = torch.arange(planes * nj * ni, dtype=torch.float64).reshape(planes, nj, ni)
    nmax = ni * max(sharded.row_assignment(nj, ws, r, layout)[2] for r in range(ws))
    parts = []
    for r in range(ws):
        j0, stride, nr = sharded.row_assignment(nj, ws, r, layout)
        mine = full[:, j0:j0 + (nr - 1) * stride + 1:stride, :].reshape(planes, ni * nr) if nr else full.new_zeros((planes, 0))
        pad = torch.full((planes, nmax - mine.shape[1]), -7.0, dtype=torch.float64)   # what a gather of unequal shares carries
        parts.append(torch.cat([mine, pad], dim=1))
    got = sharded.assemble_rows(parts, ni, nj, ws, layout)
    assert torch.equal(got, full.reshape(planes, nj * ni))


@settings(max_examples=100, deadline=None)
@given(nj=st.integers(1, 200), ws=st.integers(1, 16), layout=st.sampled_from(["cyclic", "slab"]))
def test_row_owner_is_the_rank_that_row_assignment_gives_the_row_to(nj, ws, layout):
    """bench.py attributes a row whose checksum differs to the rank (or context device) that traced it: row_owner must be the inverse
    of row_assignment for every row, rank count and layout."""
    for r in range(ws):
        j0, stride, nr = sharded.row_assignment(nj, ws, r, layout)
        for j in (j0 + stride * np.arange(nr)):
            assert sharded.row_owner(nj, ws, int(j), layout) == r


def test_per_row_checksums_sum_to_the_frame_checksum_and_localise_a_wrong_row():
    """the arithmetic of bench.py's frame / row checksums on a synthetic frame: int64 sums of the bit patterns (wrapping), one per
    image row of the [3, ni * nj] planes (pixel i + j * ni); a change in one row moves that row's checksum only."""
    ni, nj = 7, 5
    rng = np.random.default_rng(3)
    frame = torch.from_numpy(rng.normal(size=(3, ni * nj)))
    bits = frame.contiguous().view(torch.int64)
    rows = bits.view(3, nj, ni).sum(dim=(0, 2)).numpy()
    assert (int(rows.sum()) - int(bits.sum().item())) % 2 ** 64 == 0
    other = frame.clone()
    other[1, 3 * ni + 2] += 1e-9            # plane 1, row 3, column 2
    rows2 = other.contiguous().view(torch.int64).view(3, nj, ni).sum(dim=(0, 2)).numpy()
    assert list(np.nonzero(rows != rows2)[0]) == [3]


# ---- the groups of a long list's spheres (rtgr_context.hip: group_spheres / bounding_sphere; DevScene in rtgr_args.hpp) -------------
def _group(spheres, f32=False):
    """spheres: (n, 4) array of centre x, y, z and radius -> (order, nloose, groups[ng, 6]) through the library's test hook"""
    import ctypes as C
    from raytracegr_jl_amd import _abi as abi
    lib = abi.load()
    n = len(spheres)
    objs = (abi.rtgr_object * n)()
    for k, (x, y, z, r) in enumerate(spheres):
        objs[k].kind = abi.SPHERE
        objs[k].p[1], objs[k].p[2], objs[k].p[3], objs[k].p[8] = x, y, z, r
    order = np.zeros(n, np.uint32)
    nloose = C.c_uint32(0)
    groups = np.zeros((n + 1, 6))
    fn = lib.rtgr_testhook_group_spheres
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(abi.rtgr_object), C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    nsuper = C.c_uint32(0)
    ng = fn(objs, n, int(f32), order.ctypes.data, C.byref(nloose), groups.ctypes.data, n + 1, C.byref(nsuper))
    assert ng >= 0
    _group.runs = groups[ng:ng + nsuper.value]              # (the second level of the last call: runs of groups)
    return order, nloose.value, groups[:ng]


@settings(max_examples=40, deadline=None)
@given(st.one_of(st.integers(16, 400), st.integers(3000, 6000)), st.integers(0, 2**31 - 1), st.booleans(), st.sampled_from(["cloud", "line", "same", "wide_radii"]))
def test_groups_of_a_long_list_contain_their_members(n, seed, f32, shape):
    """What the FAR pass's group test rests on: the groups' member ranges tile the spheres behind the loose ones, no sphere is lost or
    listed twice, and every member — as the kernels of the scalar type see it — lies INSIDE its group's bounding sphere
    (|c_i − C| + r_i <= R_g, evaluated in extended precision); groups hold 1..RTGR_GROUP_MAX members; the layout is a function of
    the list alone."""
    rng = np.random.default_rng(seed)
    c = {"cloud": rng.normal(size=(n, 3)) * 5, "line": np.outer(rng.uniform(-8, 8, n), [1.0, 0.5, 0.0]),
         "same": np.zeros((n, 3)) + 1.25, "wide_radii": rng.uniform(-6, 6, (n, 3))}[shape]
    r = np.exp(rng.uniform(-4, 2, n)) if shape == "wide_radii" else rng.uniform(0.05, 0.5, n)
    sph = np.column_stack([c, r])
    order, nloose, groups = _group(sph, f32)
    assert sorted(order.tolist()) == list(range(n))
    if len(groups) == 0:
        assert nloose == 0 and order.tolist() == list(range(n))
        return
    pos = nloose
    ty = np.float32 if f32 else np.float64
    seen = sph[order].astype(ty).astype(np.longdouble)      # the members as the kernels get them
    for cx, cy, cz, rg, first, count in groups:
        assert first == pos and 1 <= count <= 8
        m = seen[int(first):int(first + count)]
        gc = np.array([cx, cy, cz], np.longdouble)
        assert (np.sqrt(((m[:, :3] - gc) ** 2).sum(1)) + np.abs(m[:, 3]) <= np.longdouble(rg)).all()
        assert float(ty(rg)) == rg                          # (the radius is a value of the scalar type)
        pos += int(count)
    assert pos == n
    if shape != "wide_radii":
        assert nloose == 0
    # the second level (lists of 24 groups and more): runs of groups that tile the groups, each with a bounding sphere around ALL the
    # members of its groups
    runs = _group.runs
    assert (len(runs) > 0) == (len(groups) >= 24)
    nxt = 0
    for cx, cy, cz, rr, g0, cnt in runs:
        assert g0 == nxt and 1 <= cnt <= 2 * max(8, int((len(groups) / 3.0) ** 0.5))
        first = int(groups[int(g0)][4])
        last = int(groups[int(g0 + cnt) - 1][4] + groups[int(g0 + cnt) - 1][5])
        m = seen[first:last]
        assert (np.sqrt(((m[:, :3] - np.array([cx, cy, cz], np.longdouble)) ** 2).sum(1)) + np.abs(m[:, 3]) <= np.longdouble(rr)).all()
        nxt += int(cnt)
    assert nxt == (len(groups) if len(runs) else 0)
    again = _group(sph, f32)
    assert np.array_equal(order, again[0]) and np.array_equal(groups, again[2])


def test_groups_leave_a_sky_sphere_loose_and_refuse_non_finite_lists():
    rng = np.random.default_rng(5)
    sph = np.column_stack([rng.uniform(-6, 6, (60, 3)), rng.uniform(0.2, 0.4, 60)])
    sph[17] = [0, 0, 0, 30.0]                               # a sphere around the whole scene: would blow its group's bounding sphere up
    sph[33, 3] = -0.3                                       # an inside-out sphere (sign(R) * (|x − c|² − R²), :415-419): loose too
    order, nloose, groups = _group(sph)
    assert nloose == 2 and sorted(order[:2].tolist()) == [17, 33] and len(groups) >= 8 and groups[:, 3].max() < 8
    # neighbours end up together: the groups' bounding spheres are far smaller than the cloud
    assert np.median(groups[:, 3]) < 4.5
    for bad in (np.nan, np.inf):
        broken = sph.copy()
        broken[3, 1] = bad
        assert len(_group(broken)[2]) == 0
    assert len(_group(sph[:15])[2]) == 0                    # fewer than two full groups: none
    far = sph.copy()
    far[5, 0] = 1e39                                        # finite in Float64, inf in Float32: groups there, none here
    assert len(_group(far)[2]) > 0 and len(_group(far, f32=True)[2]) == 0
