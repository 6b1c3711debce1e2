"""Atom order on the HIP path (csrc/order.hip, System's cell-sorted twin): the four kernels against numpy, and a shuffled
1 000 188-atom system (BASELINE config 1's lattice under one random permutation of its atoms) through ``System`` — rows (ids,
row order, distances, counts), CNA labels and CSP against the CPU oracle on the SAME shuffled input, bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mdapy_amd as mp
from mdapy_amd import _order, _fast_knn
from mdapy_amd.build_lattice import lattice_positions
from mdapy_amd.devarray import HArray, as_numpy
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ORG0, PBC = np.zeros(3), np.array([1, 1, 1], np.int32)


def _lattice(cells, sigma, seed, shuffle=True):
    pos, box = lattice_positions("fcc", 3.615, cells, cells, cells)
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0, sigma, pos.shape)
    if shuffle:
        pos = pos[rng.permutation(len(pos))]
    return np.ascontiguousarray(pos), np.asarray(box, float)


@pytest.mark.parametrize("boundary", [(1, 1, 1), (1, 0, 1)])
def test_order_kernels_against_numpy(boundary):
    pos, box = _lattice(24, 0.05, 1, shuffle=False)
    bd = np.array(boundary, np.int32)
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    assert _order.order_statistic(x, y, z, box, ORG0, bd) < 0.02            # a lattice builder's order
    perm0 = np.random.default_rng(2).permutation(len(x))
    xs_, ys_, zs_ = x[perm0].copy(), y[perm0].copy(), z[perm0].copy()
    assert _order.order_statistic(xs_, ys_, zs_, box, ORG0, bd) > 0.9       # shuffled
    for dev in (False, True):
        args = [HArray.from_numpy(a) for a in (xs_, ys_, zs_)] if dev else [xs_, ys_, zs_]
        xs, ys, zs, perm, n = _order.spatial_sort(*args, box, ORG0, bd)
        perm = as_numpy(perm)
        assert n == len(x) and np.array_equal(np.sort(perm), np.arange(len(x)))
        assert np.array_equal(as_numpy(xs), xs_[perm]) and np.array_equal(as_numpy(ys), ys_[perm]) and np.array_equal(as_numpy(zs), zs_[perm])
        assert _order.order_statistic(as_numpy(xs), as_numpy(ys), as_numpy(zs), box, ORG0, bd) < 0.05  # sorted = spatially ordered
    # deterministic: the same input gives the same permutation
    again = as_numpy(_order.spatial_sort(xs_, ys_, zs_, box, ORG0, bd)[3])
    assert np.array_equal(again, perm)
    # gather / scatter, 4 and 8 bytes
    rng = np.random.default_rng(3)
    for col in (rng.normal(size=len(x)), rng.integers(-5, 5, len(x)).astype(np.int32), np.arange(len(x), dtype=np.int64)):
        g = as_numpy(_order.permute(col, perm.astype(np.int32)))
        assert np.array_equal(g, col[perm])
        back = as_numpy(_order.permute(HArray.from_numpy(g), HArray.from_numpy(perm.astype(np.int32)), scatter=True))
        assert np.array_equal(back, col)
    # rows
    for M in (12, 7):
        rows = rng.integers(-1, len(x), (len(x), M)).astype(np.int32)
        dist = rng.random((len(x), M))
        cnt = rng.integers(0, M + 1, len(x)).astype(np.int32)
        v, d, c = _order.translate_rows(rows, dist, cnt, perm.astype(np.int32))
        V = np.empty_like(rows); V[perm] = np.where(rows >= 0, perm[np.clip(rows, 0, None)], rows)
        D = np.empty_like(dist); D[perm] = dist
        C = np.empty_like(cnt); C[perm] = cnt
        assert np.array_equal(as_numpy(v), V) and np.array_equal(as_numpy(d), D) and np.array_equal(as_numpy(c), C)
        v2, d2, c2 = _order.translate_rows(HArray.from_numpy(rows), None, None, HArray.from_numpy(perm.astype(np.int32)))
        assert d2 is None and c2 is None and np.array_equal(as_numpy(v2), V)


def test_absent_atoms_make_the_sort_stand_down():
    pos, box = _lattice(10, 0.05, 4)
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    x[17] = np.nan
    assert _order.spatial_sort(x, y, z, box, ORG0, PBC)[4] == len(x) - 1


@pytest.mark.parametrize("sigma", [0.05, 0.20])
def test_shuffled_1M_atom_system_goes_through_the_twin_and_equals_the_oracle(sigma):
    pos, box = _lattice(63, sigma, 7)
    assert len(pos) == 1000188
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    s = mp.System(pos=pos, box=box)
    rc, M = 3.615, 24
    s.build_neighbor(rc, max_neigh=M)
    twin = s._spatial()
    assert twin is not None and twin.N == s.N, "a shuffled system of this size is analysed on its cell-sorted twin"
    from mdapy_amd.devarray import LazyHArray
    assert isinstance(s.verlet_list, LazyHArray) and not s.verlet_list.produced  # nothing translated before somebody reads the rows
    v, d, n = (np.asarray(a) for a in (s.verlet_list, s.distance_list, s.neighbor_number))
    V = np.full_like(v, -1); D = np.full_like(d, rc + 1.0); N_ = np.zeros_like(n)
    O.build_neighbor(x, y, z, box, ORG0, PBC, rc, V, D, N_, 64)
    assert np.array_equal(n, N_) and np.array_equal(v, V) and np.array_equal(d, D)
    # exact-width rows + CNA, through the System path
    rcna = 0.854 * 3.615
    s = mp.System(pos=pos, box=box)
    s.cal_common_neighbor_analysis(rc=rcna)  # (builds the exact-width list on the twin, keyed by the original index)
    Vc, Dc, Nc = O.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, rcna, 64)
    P = np.zeros(len(x), np.int32)
    O.fcna(x, y, z, box, ORG0, PBC, Vc, Nc, P, rcna, 64)
    assert np.array_equal(s.data["cna"].to_numpy(), P)
    assert np.array_equal(np.asarray(s.verlet_list), Vc) and np.array_equal(np.asarray(s.distance_list), Dc)
    assert np.array_equal(np.asarray(s.neighbor_number), Nc)
    # CSP on the twin against the oracle fed the k-nearest rows of the plain HIP search on the shuffled input
    I = np.zeros((len(x), 12), np.int32); Dk = np.zeros((len(x), 12))
    _fast_knn.knn(x, y, z, box, ORG0, PBC, 12, I, Dk, 1)
    s.cal_centro_symmetry_parameter(12)
    C = np.zeros(len(x))
    O.get_csp(x, y, z, box, ORG0, PBC, I, 12, C, 64)
    assert np.allclose(s.data["csp"].to_numpy(), C, rtol=1e-6, atol=1e-9)
    # the same system analysed in the order it has (twin switched off): identical columns
    os.environ["MDAPY_SPATIAL_SORT"] = "0"
    try:
        p = mp.System(pos=pos, box=box)
    finally:
        del os.environ["MDAPY_SPATIAL_SORT"]
    assert p._spatial() is None
    p.cal_common_neighbor_analysis(rc=rcna)
    p.cal_centro_symmetry_parameter(12)
    assert np.array_equal(p.data["cna"].to_numpy(), s.data["cna"].to_numpy())
    assert np.array_equal(p.data["csp"].to_numpy(), s.data["csp"].to_numpy())


def test_lattice_ordered_system_gets_no_twin():
    pos, box = _lattice(40, 0.05, 9, shuffle=False)
    s = mp.System(pos=pos, box=box)
    s.cal_common_neighbor_analysis(rc=0.854 * 3.615)
    assert s._spatial() is None and int((s.data["cna"].to_numpy() == 1).sum()) == s.N


@pytest.mark.parametrize("kind,k", [("bcc", 12), ("fcc", 18), ("bcc", 14), ("fcc", 8)])
def test_keyed_knn_breaks_exact_ties_as_the_original_numbering_does(kind, k):
    """a perfect lattice is all ties: WHICH of bcc's six second neighbours are among the twelve nearest, and in which order equal
    distances are listed, is decided by atom number (knn.hip).  The search on a permuted copy with key = original number must
    give the original system's rows, neighbour for neighbour (what the twin's k-nearest analyses rest on)."""
    pos, box = lattice_positions(kind, 3.2, 9, 8, 7)
    x, y, z = (np.ascontiguousarray(pos[:, c]) for c in range(3))
    N = len(x)
    ref_i, ref_d = np.zeros((N, k), np.int32), np.zeros((N, k))
    _fast_knn.knn(x, y, z, np.asarray(box, float), ORG0, PBC, k, ref_i, ref_d, 1)
    perm = np.random.default_rng(5).permutation(N)
    xs, ys, zs = x[perm].copy(), y[perm].copy(), z[perm].copy()
    for dev in (False, True):
        got_i, got_d = np.zeros((N, k), np.int32), np.zeros((N, k))
        args = [HArray.from_numpy(a) for a in (xs, ys, zs)] if dev else [xs, ys, zs]
        key = HArray.from_numpy(perm.astype(np.int64)) if dev else perm.astype(np.int64)
        out_i = HArray.from_numpy(got_i) if dev else got_i
        out_d = HArray.from_numpy(got_d) if dev else got_d
        _fast_knn.knn(*args, np.asarray(box, float), ORG0, PBC, k, out_i, out_d, 1, key=key)
        rows, dist, _ = _order.translate_rows(as_numpy(out_i), as_numpy(out_d), None, perm.astype(np.int32))
        assert np.array_equal(as_numpy(dist), ref_d)
        assert np.array_equal(as_numpy(rows), ref_i)
    # without the key the permuted copy lists other neighbours among the ties (that is what the key is for)
    plain_i, plain_d = np.zeros((N, k), np.int32), np.zeros((N, k))
    _fast_knn.knn(xs, ys, zs, np.asarray(box, float), ORG0, PBC, k, plain_i, plain_d, 1)
    assert not np.array_equal(_order.translate_rows(plain_i, plain_d, None, perm.astype(np.int32))[0], ref_i)


def test_perfect_lattice_through_the_twin_equals_the_plain_path():
    pos, box = lattice_positions("bcc", 3.2, 12, 11, 10)
    pos = pos[np.random.default_rng(9).permutation(len(pos))]
    cols = {}
    for mode in ("0", "1"):
        os.environ["MDAPY_SPATIAL_SORT"] = mode
        try:
            s = mp.System(pos=pos, box=box)
        finally:
            del os.environ["MDAPY_SPATIAL_SORT"]
        s.cal_steinhardt_bond_orientation([4, 6], nnn=12, wl=True)
        s.cal_centro_symmetry_parameter(8)
        s.cal_common_neighbor_analysis()
        s.cal_ackland_jones_analysis()
        s.cal_polyhedral_template_matching(return_rmsd=True)
        cols[mode] = {c: s.data[c].to_numpy() for c in s.data.columns if c not in "xyz"}
        cols[mode]["rows"] = np.asarray(s.verlet_list)
        assert (s._spatial() is not None) == (mode == "1")
    for c in cols["0"]:
        assert np.array_equal(cols["0"][c], cols["1"][c], equal_nan=True), c


def test_next_frame_of_a_trajectory_is_read_through_the_last_permutation():
    """frame 2 of a trajectory — the same numbering, every atom moved by a fraction of a cell — is not sorted again: its positions are
    gathered through frame 1's permutation and found to be in a spatial order (system.py _sorted_as_last_time); lists and labels of
    both frames against the oracle, and a frame in ANOTHER numbering is sorted afresh"""
    from mdapy_amd import system as system_mod

    pos, box = _lattice(40, 0.05, 11)  # 256 000 shuffled atoms
    rng = np.random.default_rng(3)
    rc = 0.854 * 3.615
    perms = []
    for frame in range(2):
        p = pos + rng.normal(0, 0.04, pos.shape) * frame
        s = mp.System(pos=p, box=box)
        s.cal_common_neighbor_analysis(rc=rc)
        twin = s._spatial()
        assert twin is not None
        perms.append(np.asarray(as_numpy(twin._perm)).copy())
        x, y, z = (np.ascontiguousarray(p[:, k]) for k in range(3))
        V, D, Nn = O.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, rc, 16)
        P = np.zeros(len(x), np.int32)
        O.fcna(x, y, z, box, ORG0, PBC, V, Nn, P, rc, 16)
        assert np.array_equal(s.data["cna"].to_numpy(), P) and np.array_equal(np.asarray(s.verlet_list), V)
        assert np.array_equal(np.asarray(s.distance_list), D)
    assert np.array_equal(perms[0], perms[1]), "the second frame went through the first frame's permutation"
    # another numbering of the same atoms: the old permutation leaves no spatial order, the frame is sorted
    again = rng.permutation(len(pos))
    s = mp.System(pos=pos[again], box=box)
    s.cal_common_neighbor_analysis(rc=rc)
    assert s._spatial() is not None and not np.array_equal(np.asarray(as_numpy(s._spatial()._perm)), perms[0])
    assert int((s.data["cna"].to_numpy() == 1).sum()) > 0.9 * len(pos)
