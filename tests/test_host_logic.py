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
