"""GPU suite (-m gpu): the HIP kernels, called through the C ABI (mdapy_amd._<module> shims ->
libmdapy_amd.so), against (1) the CPU oracle on the same seeded inputs, (2) the golden vectors of the
reference's test-suite, (3) size-independent properties at large N.

Bars: bit-exact for integer outputs (neighbor ids, counts, labels, histogram counts) and — where the
arithmetic is IEEE-exact — for distances; 1e-6 relative for floating-point outputs (CSP, q_l, g(r)).
"""
import json
import os

import numpy as np
import pytest

import mdapy_amd as mp
from _golden import GOLDEN, fixtures_with, ids_of, input_path, misc, system_from_fixture
from mdapy_amd import _cna, _csp, _fast_knn, _neighbor, _rdf, _repeat_cell, _sbo, _wcp
from mdapy_amd.build_lattice import lattice_positions
from oracle import oracle as O

pytestmark = pytest.mark.gpu

PBC = np.array([1, 1, 1], np.int32)
ORG0 = np.zeros(3)


def _xyz(pos):
    return tuple(np.ascontiguousarray(pos[:, k]) for k in range(3))


def _fcc(n, sigma=0.0, seed=0, a=3.615):
    pos, box = lattice_positions("fcc", a, n, n, n)
    if sigma > 0:
        pos = pos + np.random.default_rng(seed).normal(0.0, sigma, pos.shape)
    return pos, box


def _cases():
    """(name, pos, box, origin, boundary) — ortho / triclinic, periodic / open, in-box / out-of-box atoms"""
    rng = np.random.default_rng(11)
    out = []
    p, b = _fcc(8, 0.05, 1)
    out.append(("fcc_rattled", p, b, ORG0, PBC))
    p, b = _fcc(7, 0.2, 2)
    out.append(("fcc_hot_shifted_origin", p + np.array([-3.0, 5.0, 1.5]), b, np.array([-3.0, 5.0, 1.5]), PBC))
    p, b = _fcc(6, 0.1, 3)
    p2 = p + rng.integers(-2, 3, p.shape) * np.diag(b)  # unwrapped input: atoms whole box lengths away
    out.append(("fcc_unwrapped", p2, b, ORG0, PBC))
    p3 = p + rng.integers(-13, 14, p.shape) * np.diag(b)  # an unwrapped trajectory after a long run: up to 13 box lengths away
    out.append(("fcc_unwrapped_far", p3 + np.array([4.0, -2.5, 9.0]), b, np.array([4.0, -2.5, 9.0]), PBC))
    out.append(("slab_open_z", p, b, ORG0, np.array([1, 1, 0], np.int32)))
    out.append(("cluster_open", p, b * 1.0, ORG0, np.array([0, 0, 0], np.int32)))
    tri = np.array([[22.0, 0.0, 0.0], [4.0, 21.0, 0.0], [-3.0, 5.0, 20.0]])
    frac = rng.random((3000, 3))
    out.append(("triclinic_random", frac @ tri + np.array([1.0, -2.0, 0.5]), tri, np.array([1.0, -2.0, 0.5]), PBC))
    out.append(("triclinic_open_y", frac @ tri, tri, ORG0, np.array([1, 0, 1], np.int32)))
    out.append(("random_gas", rng.random((4000, 3)) * 30.0, np.eye(3) * 30.0, ORG0, PBC))
    out.append(("thin_box_3cells", rng.random((600, 3)) * np.array([9.1, 30.0, 30.0]), np.diag([9.1, 30.0, 30.0]), ORG0, PBC))
    # a dense blob in a dilute box: the blob's tiles overflow the LDS halo of the tiled kernel (mop-up path)
    blob = np.concatenate([rng.random((2500, 3)) * 9.0 + 20.0, rng.random((1500, 3)) * 60.0])
    out.append(("dense_blob", blob, np.eye(3) * 60.0, ORG0, PBC))
    p, b = _fcc(9, 0.03, 8)
    out.append(("fcc_partial_tiles", p, b, ORG0, PBC))  # 9..10 cells per axis: clipped tiles, periodic seam inside a tile
    # sheared boxes large enough for the tile kernel (>= 7 cells along every vector), atoms handed in one cell vector
    # outside the box, an origin away from zero; and one with a dense blob (tiles over the LDS budget -> mop-up)
    shear = np.array([[34.0, 0.0, 0.0], [3.4, 34.0, 0.0], [-1.7, 3.4, 34.0]])  # ~10.2 cells of 3.3 along every vector
    p, b = _fcc(10, 0.06, 12, a=3.4)
    fr = p / 34.0
    fr = fr + (rng.random(fr.shape) < 0.02) * rng.integers(-1, 2, fr.shape)
    org = np.array([2.0, -7.5, 11.0])
    out.append(("triclinic_fcc_sheared", fr @ shear + org, shear, org, PBC))
    # the same sheared box OPEN along one / two vectors (a slab, a wire): the tile kernel clamps instead of wrapping there; a few
    # atoms handed in outside the box along the open vectors (clamped into the edge cells, box.h:131-156 leaves them where they
    # are), two of them far outside (their tiles go to the thread-per-atom code)
    fo = fr.copy()
    fo[:40] += rng.normal(0.0, 0.08, (40, 3)) * np.array([0, 1, 0]) + np.array([0, 1.0, 0]) * (rng.random((40, 1)) < 0.5)
    fo[40] += np.array([0, 7.5, 0]); fo[41] -= np.array([0, 3.25, 0])
    out.append(("triclinic_fcc_sheared_open_b", fo @ shear + org, shear, org, np.array([1, 0, 1], np.int32)))
    out.append(("triclinic_fcc_sheared_open_ac", fr @ shear + org, shear, org, np.array([0, 1, 0], np.int32)))
    big = np.array([[60.0, 0.0, 0.0], [6.0, 60.0, 0.0], [3.0, -6.0, 60.0]])
    fr = np.concatenate([rng.random((2500, 3)) * 0.41 + 0.3, rng.random((2500, 3))])  # ~6 atoms per cell in the blob: tiles overflow, runs fit
    out.append(("triclinic_dense_blob", fr @ big, big, ORG0, PBC))
    return out


CASES = _cases()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("rc", [3.0, 4.4])
def test_neighbor_bit_exact_vs_oracle(case, rc):
    """ids, order inside a row, counts and distances are identical to the oracle (host-space C ABI)."""
    _, pos, box, org, bnd = case
    x, y, z = _xyz(pos)
    v0, d0, n0 = O.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 4)
    v1, d1, n1 = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 1)
    assert np.array_equal(n1, n0)
    assert v1.shape == v0.shape and np.array_equal(v1, v0)
    assert np.array_equal(d1, d0)  # bitwise: IEEE sqrt / div / floor, no FMA contraction
    # fixed width, reference semantics (caller pads), with overflow: count keeps running past M
    M = max(int(n0.max()) - 2, 1)
    va = np.full((len(x), M), -1, np.int32); da = np.full((len(x), M), rc + 1.0); na = np.zeros(len(x), np.int32)
    vb = va.copy(); db = da.copy(); nb = na.copy()
    O.build_neighbor(x, y, z, box, org, bnd, rc, va, da, na, 4)
    _neighbor.build_neighbor(x, y, z, box, org, bnd, rc, vb, db, nb, 1)
    assert np.array_equal(nb, na) and np.array_equal(vb, va) and np.array_equal(db, da)


def test_unwrapped_input_takes_the_tile_kernel():
    """atoms handed in up to 14 box lengths outside an orthogonal periodic box (an unwrapped trajectory) carry their image number
    in the cell-sorted record and go through the LDS-tile kernel (asserted through the build's device flag); farther out the
    thread-per-atom kernel takes the whole call.  Rows bit for bit the oracle's either way (src/neighbor.cpp:139-177: raw
    x[j] - wrapped x[i], then box.h:120-124)."""
    import ctypes
    from mdapy_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(23)
    pos, box = _fcc(9, 0.08, 4)  # 32.5 A: ten cells of 3.2 A per axis
    org, rc = np.array([-1.0, 2.0, 0.25]), 3.2
    out4 = (ctypes.c_int64 * 4)()
    L.mdh_debug_track_counters(1)
    try:
        for reach, moved in ((1, 0), (14, 0), (15, 1), (40, 1)):
            shift = rng.integers(-reach, reach + 1, pos.shape)
            shift[0] = reach; shift[1] = -reach  # the extremes are there
            p = pos % np.diag(box) + shift * np.diag(box) + org
            x, y, z = _xyz(p)
            v0, d0, n0 = O.build_neighbor_without_max_neigh(x, y, z, box, org, PBC, rc, 4)
            v1, d1, n1 = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, org, PBC, rc, 1)
            L.mdh_debug_counters(out4)
            assert int(out4[2]) == moved, (reach, list(out4))
            assert np.array_equal(n1, n0) and np.array_equal(v1, v0) and np.array_equal(d1, d0), reach
    finally:
        L.mdh_debug_track_counters(0)


def test_neighbor_tile_overflow_and_variants():
    """fixed narrow rows (max_neigh=20, heavy overflow) on the dense-blob case: blob tiles exceed the LDS halo of
    the tiled kernel and are finished by the thread-per-atom kernel; forcing that kernel everywhere gives the same bits."""
    from mdapy_amd import _lib

    _, pos, box, org, bnd = [c for c in CASES if c[0] == "dense_blob"][0]
    x, y, z = _xyz(pos)
    rc, M = 3.0, 20
    va = np.full((len(x), M), -1, np.int32); da = np.full((len(x), M), rc + 1.0); na = np.zeros(len(x), np.int32)
    O.build_neighbor(x, y, z, box, org, bnd, rc, va, da, na, 4)
    assert na.max() > M  # rows overflow: counts keep running
    res = []
    for variant in (0, 1):
        _lib.lib().mdh_debug_set_neighbor_variant(variant)
        try:
            vb = np.full((len(x), M), -1, np.int32); db = np.full((len(x), M), rc + 1.0); nb = np.zeros(len(x), np.int32)
            _neighbor.build_neighbor(x, y, z, box, org, bnd, rc, vb, db, nb, 1)
            vc = np.empty((len(x), M), np.int32); dc = np.empty((len(x), M)); nc = np.empty(len(x), np.int32)
            _neighbor.build_neighbor(x, y, z, box, org, bnd, rc, vc, dc, nc, 1, fill_pads=True)
        finally:
            _lib.lib().mdh_debug_set_neighbor_variant(0)
        assert np.array_equal(nb, na) and np.array_equal(vb, va) and np.array_equal(db, da)
        assert np.array_equal(nc, na) and np.array_equal(vc, va) and np.array_equal(dc, da)
        res.append((vb, db, nb))
    for case in CASES[:4] + [c for c in CASES if c[0].startswith("triclinic")]:
        name, pos, box, org, bnd = case
        x, y, z = _xyz(pos)
        outs = []
        for variant in (0, 1):
            _lib.lib().mdh_debug_set_neighbor_variant(variant)
            try:
                outs.append(_neighbor.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, 3.3, 1))
                if name.startswith("triclinic_fcc_sheared") or name == "triclinic_dense_blob":  # fixed rows (narrow ones overflow inside the blob)
                    M = 20
                    va = np.full((len(x), M), -1, np.int32); da = np.full((len(x), M), 4.3); na = np.zeros(len(x), np.int32)
                    O.build_neighbor(x, y, z, box, org, bnd, 3.3, va, da, na, 4)
                    vb = np.empty((len(x), M), np.int32); db = np.empty((len(x), M)); nb = np.empty(len(x), np.int32)
                    _neighbor.build_neighbor(x, y, z, box, org, bnd, 3.3, vb, db, nb, 1, fill_pads=True)
                    assert np.array_equal(nb, na) and np.array_equal(vb, va) and np.array_equal(db, da)
                    assert (na.max() > M) == (name == "triclinic_dense_blob")  # narrow rows overflow inside the blob
                    if variant == 0:
                        plan = np.zeros(8, np.int32)
                        _lib.lib().mdh_debug_neighbor_plan(plan.ctypes.data)
                        assert plan[0] > 0 and plan[7] == 1, plan  # the tile kernel took the sheared box
            finally:
                _lib.lib().mdh_debug_set_neighbor_variant(0)
        assert all(np.array_equal(a, b) for a, b in zip(*outs))



@pytest.mark.parametrize("kind", ["fcc", "fcc_rattled_shifted", "fcc_narrow_rows", "gas", "triclinic"])
def test_neighbor_dense_cells_take_the_wide_tile_kernel(kind):
    """The reference's own benchmark call, build_neighbor(5.0, max_neigh=50) on fcc Cu (doc/gettingstarted/benchmark.ipynb):
    ~11 atoms per cell, 3-cell runs of 32 and more -> the wide instance of the tile kernel (two hit masks per run, two-byte
    tickets, row buffers sized by the tile's centres).  Rows bit for bit vs the oracle, and the tile kernel did take the call."""
    from mdapy_amd import _lib
    rng = np.random.default_rng(77)
    rc, M = 5.0, 50
    org, bnd = ORG0, PBC
    if kind == "fcc":
        pos, box = _fcc(20)
    elif kind == "fcc_rattled_shifted":
        pos, box = _fcc(18, 0.1, 5)
        org = np.array([4.0, -11.0, 2.5])
        pos = pos + org
        pos[::11] += rng.integers(-1, 2, (len(pos[::11]), 3)) * np.diag(box)  # atoms a box length outside
    elif kind == "fcc_narrow_rows":
        pos, box = _fcc(16, 0.05, 6)
        M = 20  # 42 neighbours per atom: every row overflows, the counts keep running
    elif kind == "gas":
        pos, box = rng.random((50000, 3)) * 80.0, np.eye(3) * 80.0  # ~12 atoms per cell: runs of 25 to 50 (open y: edge cells clamp outside atoms)
        bnd = np.array([1, 0, 1], np.int32)
    else:
        box = np.array([[60.0, 0.0, 0.0], [7.0, 58.0, 0.0], [-4.0, 6.0, 61.0]])
        pos = rng.random((14000, 3)) @ box
    x, y, z = _xyz(pos)
    va = np.full((len(x), M), -1, np.int32); da = np.full((len(x), M), rc + 1.0); na = np.zeros(len(x), np.int32)
    O.build_neighbor(x, y, z, box, org, bnd, rc, va, da, na, 4)
    vb = np.empty((len(x), M), np.int32); db = np.empty((len(x), M)); nb = np.empty(len(x), np.int32)
    for _ in range(2):  # (the second call plans from the first call's run-length statistics)
        _neighbor.build_neighbor(x, y, z, box, org, bnd, rc, vb, db, nb, 1, fill_pads=True)
    plan = np.zeros(8, np.int32)
    _lib.lib().mdh_debug_neighbor_plan(plan.ctypes.data)
    assert plan[0] > 0 and plan[7] == 1 and (plan[4] & 2) == 0, plan  # a tile plan was made, and it is the wide (two-byte) instance
    assert np.array_equal(nb, na) and np.array_equal(vb, va) and np.array_equal(db, da)
    if kind == "fcc":
        assert (na == 42).all()
    if kind == "fcc_narrow_rows":
        assert na.min() > M
    # the exact-width call (counting pass of the same instance) and the reference-semantics call (caller's pads)
    v2, d2, n2 = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 1)
    vo, do, no = O.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 4)
    assert np.array_equal(n2, no) and np.array_equal(v2, vo) and np.array_equal(d2, do)


@pytest.mark.parametrize("kind", ["fcc_rc6", "fcc_rc6_rattled_narrow", "gas_open", "triclinic"])
def test_neighbor_rows_of_65_to_128_slots_take_the_wide_tile_kernel(kind):
    """rc = 5.6 ... 6.5 A on fcc Cu (the structure-entropy and Steinhardt cutoffs): 78 ... 86 neighbours, ~18 atoms per cell,
    3-cell runs of ~55 — rows of up to 128 slots on the wide instance of the tile kernel (three hit masks per run), bit for bit
    vs the oracle, and the tile kernel did take the call."""
    from mdapy_amd import _lib
    rng = np.random.default_rng(78)
    rc, M = 6.0, 96
    org, bnd = ORG0, PBC
    if kind == "fcc_rc6":
        pos, box = _fcc(15)  # (9 cells of 6.025 A: 18.5 atoms per cell; beyond 19.5 the round-1 tiled kernel keeps the call)
    elif kind == "fcc_rc6_rattled_narrow":
        pos, box = _fcc(15, 0.1, 5)  # (9 cells of 6.025 A: 18.5 atoms per cell)
        M = 68  # 78 neighbours: every row overflows
    elif kind == "gas_open":
        pos, box = rng.random((30000, 3)) * 70.0, np.eye(3) * 70.0  # ~17 atoms per cell, ~61 neighbours
        bnd = np.array([1, 0, 1], np.int32)
        rc, M = 5.5, 128
    else:
        box = np.array([[60.0, 0.0, 0.0], [7.0, 58.0, 0.0], [-4.0, 6.0, 61.0]])
        pos = rng.random((11000, 3)) @ box  # (a gas: Poisson runs; denser, more than 5 % of them pass 88 candidates)
        M = 112
    x, y, z = _xyz(pos)
    va = np.full((len(x), M), -1, np.int32); da = np.full((len(x), M), rc + 1.0); na = np.zeros(len(x), np.int32)
    O.build_neighbor(x, y, z, box, org, bnd, rc, va, da, na, 4)
    vb = np.empty((len(x), M), np.int32); db = np.empty((len(x), M)); nb = np.empty(len(x), np.int32)
    for _ in range(2):  # (the second call plans from the first call's run-length statistics)
        _neighbor.build_neighbor(x, y, z, box, org, bnd, rc, vb, db, nb, 1, fill_pads=True)
    plan = np.zeros(8, np.int32)
    _lib.lib().mdh_debug_neighbor_plan(plan.ctypes.data)
    assert plan[0] > 0 and plan[7] == 1 and (plan[4] & 2) == 0, plan  # a tile plan was made, and it is the wide (two-byte) instance
    assert np.array_equal(nb, na) and np.array_equal(vb, va) and np.array_equal(db, da)
    if kind == "fcc_rc6":
        assert (na == 78).all()
    if kind == "fcc_rc6_rattled_narrow":
        assert na.min() > M
    # the fused neighbor + CNA entry on rows this wide: the same lists, and the labels of the two-call path
    pa = np.zeros(len(x), np.int32); pb = np.zeros(len(x), np.int32)
    _cna.fcna(x, y, z, box, org, bnd, vb, nb, pa, rc, 1)
    vf = np.empty((len(x), M), np.int32); df = np.empty((len(x), M)); nf = np.empty(len(x), np.int32)
    _neighbor.build_neighbor_fcna(x, y, z, box, org, bnd, rc, vf, df, nf, pb, 1, fill_pads=True)
    assert np.array_equal(nf, na) and np.array_equal(vf, va) and np.array_equal(df, da) and np.array_equal(pb, pa)
    v2, d2, n2 = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 1)
    vo, do, no = O.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 4)
    assert v2.shape[1] > 64
    assert np.array_equal(n2, no) and np.array_equal(v2, vo) and np.array_equal(d2, do)


def test_fused_neighbor_fcna_equals_the_two_calls():
    """mdh_build_neighbor_fcna == mdh_build_neighbor then mdh_fcna, bit for bit (lists AND labels), on every kind of tile the
    kernel meets: interior, periodic seam, open faces, atoms handed in outside the box / unwrapped (the stand-by kernel
    takes the call), vacuum (walked tile list), tiles over the LDS budget (mop-up + to-do list), triclinic, rows exactly
    12 / 14 wide, rows too narrow for a label; and against the oracle on the crystalline cases"""
    from mdapy_amd import _cna

    rng = np.random.default_rng(5)
    cases = [c for c in CASES if c[0] in ("fcc_rattled", "fcc_hot_shifted_origin", "fcc_unwrapped", "slab_open_z", "cluster_open",
                                          "dense_blob", "fcc_partial_tiles", "triclinic_fcc_sheared", "triclinic_dense_blob",
                                          "triclinic_random")]
    assert len(cases) == 10
    # a bcc crystal (14 neighbours inside 1.2 a) and an fcc/hcp bicrystal-like stacking through random rattling
    pb, bb = lattice_positions("bcc", 2.87, 12, 12, 12)
    cases.append(("bcc_rattled", pb + rng.normal(0, 0.03, pb.shape), bb, ORG0, PBC))
    ph, bh = lattice_positions("hcp", 2.95, 12, 14, 8)
    cases.append(("hcp_rattled", ph + rng.normal(0, 0.03, ph.shape), bh, ORG0, PBC))
    labelled = 0
    for name, pos, box, org, bnd in cases:
        x, y, z = _xyz(pos)
        n = len(x)
        rcs = {"bcc_rattled": 1.2 * 2.87, "hcp_rattled": 0.5 * (1 + 2 ** 0.5) * 2.95}.get(name, 0.854 * 3.615 if "fcc" in name or "slab" in name or "cluster" in name else 3.3)
        if name.startswith("triclinic_fcc"):
            rcs = 0.854 * 3.4
        for M in ((14, 12, 20, 8) if name in ("fcc_rattled", "bcc_rattled") else (16,)):
            va = np.full((n, M), -1, np.int32); da = np.full((n, M), rcs + 1.0); na = np.zeros(n, np.int32)
            _neighbor.build_neighbor(x, y, z, box, org, bnd, rcs, va, da, na, 1)
            pa = np.zeros(n, np.int32)
            _cna.fcna(x, y, z, box, org, bnd, va, na, pa, rcs, 1)
            for fill in (False, True):
                vb = np.full((n, M), -1, np.int32) if not fill else np.empty((n, M), np.int32)
                db = np.full((n, M), rcs + 1.0) if not fill else np.empty((n, M))
                nb = np.zeros(n, np.int32); pb_ = np.zeros(n, np.int32)
                _neighbor.build_neighbor_fcna(x, y, z, box, org, bnd, rcs, vb, db, nb, pb_, 1, fill_pads=fill)
                assert np.array_equal(nb, na) and np.array_equal(vb, va) and np.array_equal(db, da), (name, M, fill)
                assert np.array_equal(pb_, pa), (name, M, fill, int((pb_ != pa).sum()))
            if M == 8:
                assert not pa.any()  # rows too narrow: nothing is labelled (cna.cpp:456)
            labelled += int((pa > 0).sum())
            if name in ("fcc_rattled", "bcc_rattled", "hcp_rattled", "slab_open_z") and M >= 14:
                po = np.zeros(n, np.int32)
                O.fcna(x, y, z, box, org, bnd, va, na, po, rcs, 4)
                assert np.array_equal(pa, po) and (po > 0).any(), name
    assert labelled > 10000
    # caller-initialised labels survive where the analysis does not speak
    _, pos, box, org, bnd = [c for c in CASES if c[0] == "random_gas"][0]
    x, y, z = _xyz(pos)
    n = len(x)
    v = np.empty((n, 16), np.int32); d = np.empty((n, 16)); c_ = np.empty(n, np.int32); p7 = np.full(n, 7, np.int32)
    _neighbor.build_neighbor_fcna(x, y, z, box, org, bnd, 3.0, v, d, c_, p7, 1, fill_pads=True)
    p7b = np.full(n, 7, np.int32)
    _cna.fcna(x, y, z, box, org, bnd, v, c_, p7b, 3.0, 1)
    assert np.array_equal(p7, p7b) and (p7 == 7).any()



def test_neighbor_live_tile_list_longer_than_expected():
    """the tile kernel sizes its first launch from the occupancy statistics of the LAST call with the same (N, grid); when the
    atoms have spread since (stale statistics: they are recounted only every 8th call), the list of live tiles is longer
    than the launch and the stand-by launch walks the rest.  Same N and box, first a compact crystal in a corner of a
    large open box, then the same number of atoms as a thin sheet across it: rows equal the oracle's both times."""
    from mdapy_amd import _lib

    a = 3.615
    blob, _ = lattice_positions("fcc", a, 16, 16, 16)            # 16 384 atoms in a 58 A cube
    sheet, _ = lattice_positions("fcc", a, 64, 64, 1)            # 16 384 atoms as a 231 x 231 x 3.6 A sheet
    assert len(blob) == len(sheet)
    rng = np.random.default_rng(3)
    box = np.diag([260.0, 260.0, 260.0])
    bnd = np.array([0, 0, 0], np.int32)
    rc, M = 0.854 * a, 16
    plan = np.zeros(8, np.int32)
    for k, pos in enumerate((blob + 5.0, sheet + np.array([5.0, 5.0, 120.0]), sheet + np.array([9.0, 7.0, 60.0]))):
        pos = pos + rng.normal(0, 0.03, pos.shape)
        x, y, z = _xyz(pos)
        n = len(x)
        va = np.full((n, M), -1, np.int32); da = np.full((n, M), rc + 1.0); na = np.zeros(n, np.int32)
        O.build_neighbor(x, y, z, box, ORG0, bnd, rc, va, da, na, 4)
        vb = np.empty((n, M), np.int32); db = np.empty((n, M)); nb = np.empty(n, np.int32)
        _neighbor.build_neighbor(x, y, z, box, ORG0, bnd, rc, vb, db, nb, 1, fill_pads=True)
        _lib.lib().mdh_debug_neighbor_plan(plan.ctypes.data)
        assert plan[0] > 0 and (plan[4] & 1) == 0, plan  # the tile kernel ran, on a list of live tiles (vacuum around the atoms)
        assert np.array_equal(nb, na) and np.array_equal(vb, va) and np.array_equal(db, da), k


def test_neighbor_second_pass_grid_from_a_stale_count():
    """the tile kernel's second pass (one-cell slices of the tiles whose halo outgrew LDS) gets a small stand-by grid when the
    previous build with the same (N, grid) listed nothing — the count travels through pinned memory.  Same N and box: first an
    even crystal (nothing listed), then the same atoms with half of them squeezed into a slab three times as dense (most of
    its tiles are listed): the stand-by grid walks the whole list, rows equal the oracle's, and again with the count updated."""
    a = 3.615
    pos0, box = lattice_positions("fcc", a, 24, 24, 24)
    box = np.asarray(box, float)
    rng = np.random.default_rng(17)
    pos0 = pos0 + rng.normal(0, 0.04, pos0.shape)
    L = box[0][0]
    dense = pos0.copy()
    lower = dense[:, 2] < 0.5 * L
    dense[lower, 2] = dense[lower, 2] / 3.0  # the lower half of the box into its lowest sixth: ~7.5 atoms per cell there
    rc, M = 0.854 * a, 40
    for k, pos in enumerate((pos0, dense, dense)):
        x, y, z = _xyz(pos)
        n = len(x)
        va = np.full((n, M), -1, np.int32); da = np.full((n, M), rc + 1.0); na = np.zeros(n, np.int32)
        O.build_neighbor(x, y, z, box, ORG0, PBC, rc, va, da, na, 4)
        vb = np.empty((n, M), np.int32); db = np.empty((n, M)); nb = np.empty(n, np.int32)
        _neighbor.build_neighbor(x, y, z, box, ORG0, PBC, rc, vb, db, nb, 1, fill_pads=True)
        assert np.array_equal(nb, na) and np.array_equal(vb, va) and np.array_equal(db, da), k
        assert k == 0 or na.max() > 16


def test_neighbor_device_space_and_pads():
    """HBM-resident path (torch tensors in, HArray out) == host-space path; kernel-written pads == -1 / rc+1."""
    import torch

    pos, box = _fcc(10, 0.05, 4)
    x, y, z = _xyz(pos)
    rc = 3.615
    v0, d0, n0 = O.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, rc, 4)
    tx, ty, tz = (torch.from_numpy(a).cuda() for a in (x, y, z))
    v1, d1, n1 = _neighbor.build_neighbor_without_max_neigh(tx, ty, tz, box, ORG0, PBC, rc, 1)
    assert type(v1).__name__ == "HArray"
    assert np.array_equal(np.asarray(v1), v0) and np.array_equal(np.asarray(d1), d0) and np.array_equal(np.asarray(n1), n0)
    assert (np.asarray(v1) == -1).sum() == (v0 == -1).sum() and np.all(np.asarray(d1)[v0 == -1] == rc + 1.0)


def test_neighbor_knife_edge_counts():
    ref = json.load(open(GOLDEN / "knife_edge_counts.json"))["counts"]
    s = mp.build_crystal("Cu", "fcc", 3.615, nx=10, ny=10, nz=10)
    s.build_neighbor(3.615)
    got = np.bincount(np.asarray(s.neighbor_number))
    assert {str(k): int(v) for k, v in enumerate(got) if v} == ref
    assert s.verlet_list.shape == (4000, 18)


def test_neighbor_determinism_and_large_properties():
    """1M atoms: run twice (bitwise identical), symmetric pairs, sorted-ness of nothing assumed; counts vs closed form."""
    pos, box = _fcc(63)  # 1 000 188 atoms
    x, y, z = _xyz(pos)
    rc = 0.854 * 3.615
    v1, d1, n1 = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, rc, 1)
    v2, d2, n2 = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, rc, 1)
    assert np.array_equal(v1, v2) and np.array_equal(d1, d2) and np.array_equal(n1, n2)
    assert np.all(n1 == 12) and v1.shape[1] == 12
    assert np.allclose(d1, 3.615 / np.sqrt(2), rtol=1e-12)
    # symmetry: j in row(i)  <=>  i in row(j)   (checked through a hash of unordered pairs)
    i = np.repeat(np.arange(len(x), dtype=np.int64), 12)
    j = v1.reshape(-1).astype(np.int64)
    assert np.array_equal(np.sort(i * len(x) + j), np.sort(j * len(x) + i))


@pytest.mark.parametrize("case", CASES[:7], ids=[c[0] for c in CASES[:7]])
def test_sort_and_cna_vs_oracle(case):
    _, pos, box, org, bnd = case
    x, y, z = _xyz(pos)
    rc = 3.2
    v, d, n = O.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 4)
    p0 = np.zeros(len(x), np.int32); p1 = np.zeros(len(x), np.int32)
    O.fcna(x, y, z, box, org, bnd, v, n, p0, rc, 4)
    _cna.fcna(x, y, z, box, org, bnd, v, n, p1, rc, 1)
    assert np.array_equal(p1, p0)
    # partial selection sort (ties: first minimum wins)
    k = min(6, v.shape[1])
    va, da = v.copy(), d.copy(); vb, db = v.copy(), d.copy()
    O.sort_verlet_by_distance(va, da, k, 4)
    _neighbor.sort_verlet_by_distance(vb, db, k, 1)
    assert np.array_equal(vb, va) and np.array_equal(db, da)


def test_sort_wide_rows_keeps_the_reference_order_among_equal_distances():
    """rows of more than 160 slots (sorted in HBM) on a PERFECT lattice, whole rows and the first 20: the reference's selection
    sort decides the order of the equidistant neighbours, and so must this one"""
    pos, box = _fcc(6)
    x, y, z = _xyz(pos)
    v, d, n = O.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, 8.2, 4)
    assert v.shape[1] > 160
    for k in (v.shape[1], 20):
        va, da = v.copy(), d.copy(); vb, db = v.copy(), d.copy()
        O.sort_verlet_by_distance(va, da, k, 4)
        _neighbor.sort_verlet_by_distance(vb, db, k, 1)
        assert np.array_equal(vb, va) and np.array_equal(db, da)


@pytest.mark.parametrize("M", [1, 2, 3, 13, 14, 16, 27, 50, 53, 100, 107, 171, 300, 681, 1024, 1025, 1100])
def test_sort_every_row_width_vs_oracle(M):
    """the selection kernel gives a row 1, 2, 4, 8 or 16 lanes by its width (and rows past 1024 entries are sorted in HBM):
    rows full of EQUAL distances (a handful of distinct values), row counts that end inside a workgroup, sorted rows that must
    be left alone — entry for entry the reference's selection (neighbor.cpp:745-775)"""
    rng = np.random.default_rng(100 + M)
    for N in (1, 67, 1000 if M < 200 else 130):
        d = rng.integers(0, 5, (N, M)).astype(np.float64) * 0.25 + 2.0
        d[N // 2:] += rng.random((N - N // 2, M)) * 1e-3  # half the rows without ties
        d[::5] = np.sort(d[::5], axis=1)  # every fifth row arrives sorted
        v = rng.integers(0, 1 << 30, (N, M)).astype(np.int32)
        for k in sorted({1, min(12, M), min(14, M), M if M <= 171 else 20}):
            va, da = v.copy(), d.copy(); vb, db = v.copy(), d.copy()
            O.sort_verlet_by_distance(va, da, k, 4)
            _neighbor.sort_verlet_by_distance(vb, db, k, 1)
            assert np.array_equal(vb, va) and np.array_equal(db, da), (M, N, k)


def _fcna_cases():
    """boxes of >= 10 cutoffs per periodic edge: the single-precision pair tests of k_fcna_f32 apply"""
    rng = np.random.default_rng(5)
    out = []
    p, b = _fcc(11, 0.06, 21)
    out.append(("fcc_rattled_seams", p, b, ORG0, PBC, 3.2))
    org = np.array([-7.0, 2.5, 40.0])
    p, b = _fcc(10, 0.15, 22)
    p = p + org
    p[::7] += rng.integers(-1, 2, (len(p[::7]), 3)) * np.diag(b)  # some atoms handed in a box length away
    out.append(("fcc_hot_shifted_unwrapped", p, b, org, PBC, 3.2))
    pb, bb = lattice_positions("bcc", 2.87, 13, 13, 13)
    out.append(("bcc_rattled", pb + rng.normal(0, 0.04, pb.shape), bb, ORG0, PBC, 3.2))  # 14 listed neighbours: (6,6,6) signatures walk the clusters
    ph, bh = lattice_positions("hcp", 2.95, 12, 7, 7)
    out.append(("hcp_open_z", ph + rng.normal(0, 0.03, ph.shape), bh, ORG0, np.array([1, 1, 0], np.int32), 3.3))
    out.append(("random_gas", rng.random((30000, 3)) * 40.0, np.eye(3) * 40.0, ORG0, PBC, 3.6))
    # knife edge: the second shell of perfect fcc 1e-7 (relative) beyond the cutoff — every atom has a pair inside the band
    p, b = _fcc(10)
    out.append(("fcc_second_shell_in_band", p, b, ORG0, PBC, 3.615 * (1.0 - 1e-7)))
    return out


@pytest.mark.parametrize("case", _fcna_cases(), ids=lambda c: c[0])
def test_fcna_single_precision_pair_tests_vs_oracle(case):
    """mdh_fcna on boxes where the single-precision kernel runs: labels == oracle == the double-precision kernel"""
    from mdapy_amd import _lib
    _, pos, box, org, bnd, rc = case
    x, y, z = _xyz(pos)
    v, d, n = O.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 4)
    want = np.zeros(len(x), np.int32)
    O.fcna(x, y, z, box, org, bnd, v, n, want, rc, 4)
    got = np.zeros(len(x), np.int32)
    _cna.fcna(x, y, z, box, org, bnd, v, n, got, rc, 1)
    assert np.array_equal(got, want)
    try:
        _lib.lib().mdh_debug_set_fcna_variant(1)
        f64 = np.zeros(len(x), np.int32)
        _cna.fcna(x, y, z, box, org, bnd, v, n, f64, rc, 1)
    finally:
        _lib.lib().mdh_debug_set_fcna_variant(0)
    assert np.array_equal(f64, want)
    if case[0].startswith("fcc") or case[0].startswith("bcc") or case[0].startswith("hcp"):
        assert (want > 0).mean() > 0.5  # the case does exercise labelled atoms


@pytest.mark.parametrize("case", _fcna_cases(), ids=lambda c: c[0])
def test_fused_labels_single_precision_pair_tests_vs_oracle(case):
    """the labels made INSIDE the tile kernel (single-precision pair tests on the tile's staged coordinates, the scan's decision band,
    pairs inside it finished in double precision from the to-do list) == the oracle's fcna on the oracle's lists: through
    mdh_build_neighbor_fcna (fixed width) and mdh_build_neighbor_exact_fcna (max_neigh=None: host arrays, then HBM-resident twice — the
    second call builds at the remembered width), lists included; the knife-edge case puts a pair of EVERY atom inside the band"""
    import torch

    name, pos, box, org, bnd, rc = case
    x, y, z = _xyz(pos)
    n = len(x)
    v, d, c = O.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 4)
    want = np.zeros(n, np.int32)
    O.fcna(x, y, z, box, org, bnd, v, c, want, rc, 4)
    M = int(v.shape[1])
    # fixed width = the exact one, and a wider one
    for width in (M, M + 3):
        vf = np.empty((n, width), np.int32); df = np.empty((n, width)); nf = np.empty(n, np.int32); pf = np.zeros(n, np.int32)
        _neighbor.build_neighbor_fcna(x, y, z, box, org, bnd, rc, vf, df, nf, pf, 1, fill_pads=True)
        assert np.array_equal(nf, c) and np.array_equal(vf[:, :M], v) and np.array_equal(df[:, :M], d), (name, width)
        assert np.array_equal(pf, want), (name, width, int((pf != want).sum()))
    # exact width, host arrays
    pe = np.zeros(n, np.int32)
    ve, de, ne = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 1, pattern=pe)
    assert np.array_equal(ne, c) and np.array_equal(ve, v) and np.array_equal(de, d) and np.array_equal(pe, want), name
    # exact width, HBM-resident: twice (counting pass, then the remembered width), and once more after ANOTHER system of the same
    # size and grid left a different width behind (a wrong hint: one wasted build, the labels of the second)
    dev = torch.device("cuda", 0)
    tx, ty, tz = (torch.from_numpy(a).to(dev) for a in (x, y, z))
    for _ in range(2):
        pt = torch.zeros(n, dtype=torch.int32, device=dev)
        vt, dt, nt = _neighbor.build_neighbor_without_max_neigh(tx, ty, tz, box, org, bnd, rc, 1, pattern=pt)
        assert np.array_equal(np.asarray(nt), c) and np.array_equal(np.asarray(vt), v) and np.array_equal(np.asarray(dt), d), name
        assert np.array_equal(pt.cpu().numpy(), want), name
    if name.startswith("fcc_rattled"):
        sq = torch.from_numpy(np.ascontiguousarray(pos * np.array([1.0, 1.0, 0.97]))).to(dev)  # the same grid (cells absorb 3 %), denser along z
        _neighbor.build_neighbor_without_max_neigh(sq[:, 0].contiguous(), sq[:, 1].contiguous(), sq[:, 2].contiguous(), box, org, bnd, rc, 1)
        pt = torch.zeros(n, dtype=torch.int32, device=dev)
        vt, dt, nt = _neighbor.build_neighbor_without_max_neigh(tx, ty, tz, box, org, bnd, rc, 1, pattern=pt)
        assert np.array_equal(np.asarray(vt), v) and np.array_equal(pt.cpu().numpy(), want), name
    if name == "fcc_second_shell_in_band":
        assert (c == 12).all() and (want == 1).all()
    if name[:3] in ("fcc", "bcc", "hcp"):
        assert (want > 0).mean() > 0.5


def _fcna_wide_and_sheared_cases():
    """the instances of the fused tile kernel that `_fcna_cases` (orthogonal boxes, rows of <= 16 slots: TRI=0, TK8=1) does not reach:
    a sheared box periodic along all three vectors and one open along b (TRI=1), and an orthogonal box whose dense region forces
    the wide (two-byte ticket) instance on a call that also holds 12- and 14-neighbour atoms (TK8=0, FCNA=1)"""
    rng = np.random.default_rng(61)
    out = []
    shear = np.array([[34.0, 0.0, 0.0], [3.4, 34.0, 0.0], [-1.7, 3.4, 34.0]])
    p, _ = _fcc(10, 0.12, 31, a=3.4)
    fr = p / 34.0
    fr = fr + (rng.random(fr.shape) < 0.02) * rng.integers(-1, 2, fr.shape)  # some atoms a cell vector outside the box
    org = np.array([2.0, -7.5, 11.0])
    out.append(("sheared_periodic", fr @ shear + org, shear, org, PBC, 0.854 * 3.4, "tri"))
    fo = p / 34.0
    fo[:40] += rng.normal(0.0, 0.05, (40, 3)) * np.array([0, 1, 0])
    out.append(("sheared_open_b", fo @ shear + org, shear, org, np.array([1, 0, 1], np.int32), 0.854 * 3.4, "tri"))
    # fcc crystal with a ball of it replaced by a dense gas (~11 atoms per cell: 3-cell runs of 32 and more -> the wide instance for
    # the whole call); the crystal around it keeps its 12 neighbours
    a = 3.615
    p, b = _fcc(14, 0.04, 32)
    c = np.diag(b) / 2
    keep = np.linalg.norm(p - c, axis=1) > 9.5
    ball = rng.normal(size=(800, 3))
    ball = ball / np.linalg.norm(ball, axis=1)[:, None] * (rng.random((800, 1)) ** (1 / 3)) * 8.0 + c
    out.append(("fcc_with_dense_ball", np.concatenate([p[keep], ball]), b, ORG0, PBC, 0.854 * a, "wide"))
    pb, bb = lattice_positions("bcc", 2.87, 16, 16, 16)
    pb = pb + rng.normal(0, 0.03, pb.shape)
    cb = np.diag(bb) / 2
    keep = np.linalg.norm(pb - cb, axis=1) > 9.5
    ballb = rng.normal(size=(580, 3))
    ballb = ballb / np.linalg.norm(ballb, axis=1)[:, None] * (rng.random((580, 1)) ** (1 / 3)) * 8.0 + cb
    out.append(("bcc_with_dense_ball", np.concatenate([pb[keep], ballb]), bb, ORG0, PBC, 1.2 * 2.87, "wide"))
    return out


@pytest.mark.parametrize("case", _fcna_wide_and_sheared_cases(), ids=lambda c: c[0])
def test_fused_labels_sheared_and_wide_instances_vs_oracle(case):
    """k_neighbor_lane<TRI=1, FCNA=1> and k_neighbor_lane<TK8=0, FCNA=1> DIRECTLY against the oracle: the labels and lists of
    mdh_build_neighbor_fcna / mdh_build_neighbor_exact_fcna == O.fcna on O.build_neighbor_without_max_neigh (src/cna.cpp:429-506,
    src/neighbor.cpp:189-388), with the instance that ran asserted through mdh_debug_neighbor_plan"""
    from mdapy_amd import _lib

    name, pos, box, org, bnd, rc, inst = case
    x, y, z = _xyz(pos)
    n = len(x)
    v, d, c = O.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 4)
    want = np.zeros(n, np.int32)
    O.fcna(x, y, z, box, org, bnd, v, c, want, rc, 4)
    M = int(v.shape[1])
    assert ((c == 12) | (c == 14)).sum() > 1000 and (want > 0).sum() > 1000, name  # labelled atoms are there
    plan = np.zeros(8, np.int32)
    for width in (M, M + 5):
        vf = np.empty((n, width), np.int32); df = np.empty((n, width)); nf = np.empty(n, np.int32); pf = np.zeros(n, np.int32)
        for _ in range(2):  # (the second call plans from the first call's run-length statistics)
            pf[:] = 0
            _neighbor.build_neighbor_fcna(x, y, z, box, org, bnd, rc, vf, df, nf, pf, 1, fill_pads=True)
        _lib.lib().mdh_debug_neighbor_plan(plan.ctypes.data)
        assert plan[0] > 0 and plan[7] == 1, (name, plan)  # the tile kernel took the call
        if inst == "wide":
            assert (plan[4] & 2) == 0 and M > 16, (name, plan)  # ... as its wide (two-byte ticket) instance
        else:  # sheared: the one-byte instance for rows of <= 16 slots (TRI=1, TK8=1), the wide one beyond (TRI=1, TK8=0) — both labelled here
            assert ((plan[4] & 2) != 0) == (width <= 16), (name, width, plan)
        assert np.array_equal(nf, c) and np.array_equal(vf[:, :M], v) and np.array_equal(df[:, :M], d), (name, width)
        assert np.array_equal(pf, want), (name, width, int((pf != want).sum()))
    pe = np.zeros(n, np.int32)
    ve, de, ne = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 1, pattern=pe)
    assert np.array_equal(ne, c) and np.array_equal(ve, v) and np.array_equal(de, d) and np.array_equal(pe, want), name


def test_neighbor_builds_from_the_cell_sorted_records_vs_oracle():
    """A neighbor build of input that comes in some spatial order keeps no cell-sorted copy of the atoms (CellGrid::ix: its kernels
    read them through the cell-sorted id list) — the default, asserted here, and so what every other test of this file runs.  With
    mdh_debug_set_indirect(0) the same calls go through the 32-byte records of k_gather, the path unordered input takes: the rows
    (and fused labels) of both are the oracle's bit for bit — periodic and open, orthogonal and sheared boxes, atoms handed in
    outside the box (image codes), unwrapped trajectories, dense cells (the wide instance), the mop-up kernels
    (src/neighbor.cpp:64-187)."""
    from mdapy_amd import _lib

    L = _lib.lib()
    prev = L.mdh_debug_set_indirect(0)
    try:
        assert prev == 1 or os.environ.get("MDH_INDIRECT") == "0"
        for case in CASES:
            for rc in (3.0, 4.4):
                test_neighbor_bit_exact_vs_oracle(case, rc)
        test_unwrapped_input_takes_the_tile_kernel()
        test_neighbor_tile_overflow_and_variants()
        for kind in ("fcc_rattled_shifted", "gas", "triclinic"):
            test_neighbor_dense_cells_take_the_wide_tile_kernel(kind)
        test_neighbor_rows_of_65_to_128_slots_take_the_wide_tile_kernel("fcc_rc6_rattled_narrow")
        test_fused_neighbor_fcna_equals_the_two_calls()
        for case in _fcna_cases():
            test_fused_labels_single_precision_pair_tests_vs_oracle(case)
        for case in _fcna_wide_and_sheared_cases():
            test_fused_labels_sheared_and_wide_instances_vs_oracle(case)
        test_neighbor_cell_window_hint_same_rows_and_broken_promise_is_reported()
    finally:
        L.mdh_debug_set_indirect(prev)
    # and one system large enough for the build's own order sample (>= 2^18 atoms): in lattice order (indirect) and shuffled
    # (the sample sends the NEXT build of this (N, grid) to the records), both against the oracle
    pos, box = _fcc(42, 0.05, 11)  # 296 352 atoms
    rc = 3.3
    for shuffled in (False, True):
        p = pos[np.random.default_rng(5).permutation(len(pos))] if shuffled else pos
        x, y, z = _xyz(p)
        v0, d0, n0 = O.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, rc, 8)
        for _ in range(3):
            v1, d1, n1 = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, rc, 1)
            assert np.array_equal(n1, n0) and np.array_equal(v1, v0) and np.array_equal(d1, d0), shuffled


def test_fused_labels_where_the_tile_kernel_does_not_apply():
    """boxes of fewer than seven cells per periodic axis, a thin slab that is replicated first, an empty and a one-atom system: the
    one-call entries label from the finished rows (thread-per-atom build, then the analysis inside the call) — lists and labels as
    the two reference calls leave them; System.cal_common_neighbor_analysis(rc) on the same inputs against the oracle-backed class"""
    rng = np.random.default_rng(17)
    a = 3.615
    for n, sig, bnd in (((5, 5, 5), 0.05, PBC), ((6, 4, 3), 0.1, np.array([1, 0, 1], np.int32)), ((3, 3, 3), 0.02, PBC)):
        pos, box = lattice_positions("fcc", a, *n)
        pos = pos + rng.normal(0, sig, pos.shape)
        x, y, z = _xyz(pos)
        rc = 0.854 * a
        if min(np.diag(box)[np.asarray(bnd) == 1]) < 2 * rc:
            continue
        v, d, c = O.build_neighbor_without_max_neigh(x, y, z, box, ORG0, bnd, rc, 4)
        want = np.zeros(len(x), np.int32)
        O.fcna(x, y, z, box, ORG0, bnd, v, c, want, rc, 4)
        pe = np.zeros(len(x), np.int32)
        ve, de, ne = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, ORG0, bnd, rc, 1, pattern=pe)
        assert np.array_equal(ve, v) and np.array_equal(de, d) and np.array_equal(ne, c) and np.array_equal(pe, want), n
        M = int(v.shape[1]) + 2
        vf = np.empty((len(x), M), np.int32); df = np.empty((len(x), M)); nf = np.empty(len(x), np.int32); pf = np.zeros(len(x), np.int32)
        _neighbor.build_neighbor_fcna(x, y, z, box, ORG0, bnd, rc, vf, df, nf, pf, 1, fill_pads=True)
        assert np.array_equal(nf, c) and np.array_equal(vf[:, :M - 2], v) and np.array_equal(pf, want), n
    # through System: a thin periodic slab (replicated: the labels come from the class, not from the one-call entry), and the 5 x 5 x 5 box
    import mdapy_amd as mp
    for n in ((5, 5, 5), (8, 8, 1)):
        pos, box = lattice_positions("fcc", a, *n)
        pos = pos + rng.normal(0, 0.04, pos.shape)
        s = mp.System(pos=pos, box=box)
        s.cal_common_neighbor_analysis(rc=0.854 * a)
        got = np.asarray(s.data["cna"].to_numpy())
        x, y, z = _xyz(pos)
        # the oracle on a replica wide enough for the reference's own rule (15 A): labels of the first copy
        reps = [int(np.ceil(15.0 / L)) if L < 15.0 else 1 for L in np.diag(box)]
        big = np.concatenate([pos + np.array([i, j, k]) * np.diag(box) for i in range(reps[0]) for j in range(reps[1]) for k in range(reps[2])])
        bb = box * np.array(reps)[:, None]
        bx, by, bz = _xyz(big)
        v, d, c = O.build_neighbor_without_max_neigh(bx, by, bz, bb, ORG0, PBC, 0.854 * a, 4)
        want = np.zeros(len(bx), np.int32)
        O.fcna(bx, by, bz, bb, ORG0, PBC, v, c, want, 0.854 * a, 4)
        assert np.array_equal(got, want[:len(pos)]), n


@pytest.mark.parametrize("case", [c for c in _fcna_cases() if c[0] != "fcc_second_shell_in_band"], ids=lambda c: c[0])
def test_acna_single_precision_pair_tests_vs_oracle(case):
    """adaptive CNA on boxes large enough for the single-precision kernel (edges > 8 local cutoffs): labels == oracle == the
    double-precision kernel; lattices, unwrapped atoms, an open axis, a gas (every atom unlabelled, many through the to-do list)"""
    from mdapy_amd import _lib
    name, pos, box, org, bnd, _ = case
    x, y, z = _xyz(pos)
    N, k = len(x), 14
    idx = np.zeros((N, k), np.int32); dk = np.zeros((N, k))
    _fast_knn.knn(x, y, z, box, org, bnd, k, idx, dk, 1)
    want = np.zeros(N, np.int32)
    O.acna(x, y, z, box, org, bnd, idx, want, 4)
    got = np.zeros(N, np.int32)
    _cna.acna(x, y, z, box, org, bnd, idx, got, 1)
    assert np.array_equal(got, want)
    try:
        _lib.lib().mdh_debug_set_fcna_variant(1)
        f64 = np.zeros(N, np.int32)
        _cna.acna(x, y, z, box, org, bnd, idx, f64, 1)
    finally:
        _lib.lib().mdh_debug_set_fcna_variant(0)
    assert np.array_equal(f64, want)
    if name != "random_gas":
        assert (want > 0).mean() > 0.5


@pytest.mark.parametrize("name", ["fcc", "bcc", "hcp", "diamond"])
@pytest.mark.parametrize("sigma", [0.0, 0.08])
def test_knn_acna_csp_ids_vs_oracle(name, sigma):
    a = {"fcc": 3.615, "bcc": 2.86, "hcp": 3.21, "diamond": 3.57}[name]
    n = {"fcc": 6, "bcc": 8, "hcp": 8, "diamond": 5}[name]
    pos, box = lattice_positions(name, a, n, n, n)
    if sigma:
        pos = pos + np.random.default_rng(5).normal(0, sigma, pos.shape)
    x, y, z = _xyz(pos)
    N = len(x)
    k = 14
    i0 = np.zeros((N, k), np.int32); q0 = np.zeros((N, k)); i1 = np.zeros((N, k), np.int32); q1 = np.zeros((N, k))
    O.knn(x, y, z, box, ORG0, PBC, k, i0, q0, 4)
    _fast_knn.knn(x, y, z, box, ORG0, PBC, k, i1, q1, 1)
    assert np.array_equal(q1, q0)  # sorted distances are bit-identical; ids may permute inside exact ties
    if sigma:  # no exact ties: ids must agree too (on perfect lattices ids may permute inside a tie group)
        assert np.array_equal(i1, i0)
    p0 = np.zeros(N, np.int32); p1 = np.zeros(N, np.int32)
    O.acna(x, y, z, box, ORG0, PBC, i1, p0, 4)
    _cna.acna(x, y, z, box, ORG0, PBC, i1, p1, 1)
    assert np.array_equal(p1, p0)
    c0 = np.zeros(N); c1 = np.zeros(N)
    O.get_csp(x, y, z, box, ORG0, PBC, i1, 12, c0, 4)
    _csp.get_csp(x, y, z, box, ORG0, PBC, i1, 12, c1, 1)
    assert np.allclose(c1, c0, rtol=1e-6, atol=1e-12)
    O.get_csp(x, y, z, box, ORG0, PBC, i1, 6, c0, 4)  # generic-K path
    _csp.get_csp(x, y, z, box, ORG0, PBC, i1, 6, c1, 1)
    assert np.allclose(c1, c0, rtol=1e-6, atol=1e-12)
    s0 = np.zeros((N, 12), np.int32); s1 = np.zeros((N, 12), np.int32); p0 = np.zeros(N, np.int32); p1 = np.zeros(N, np.int32)
    O.ids(x, y, z, box, ORG0, PBC, i1, s0, p0, 4)
    _cna.ids(x, y, z, box, ORG0, PBC, i1, s1, p1, 1)
    assert np.array_equal(s1, s0) and np.array_equal(p1, p0)


@pytest.mark.parametrize("case", [CASES[0], CASES[5], CASES[7]], ids=[CASES[0][0], CASES[5][0], CASES[7][0]])
def test_knn_general_vs_oracle(case):
    _, pos, box, org, bnd = case
    x, y, z = _xyz(pos[:1500])
    N = len(x)
    for k in (1, 12, 18):
        i0 = np.zeros((N, k), np.int32); q0 = np.zeros((N, k)); i1 = np.zeros((N, k), np.int32); q1 = np.zeros((N, k))
        O.knn(x, y, z, box, org, bnd, k, i0, q0, 4)
        _fast_knn.knn(x, y, z, box, org, bnd, k, i1, q1, 1)
        assert np.array_equal(q1, q0)
        assert np.array_equal(i1, i0)


def test_knn_near_kernel_equals_general_kernel():
    """k <= 24: the near kernel (sorted list in registers, the 27 cells around the query) + the general kernel on what it
    lists == the general kernel alone (rings until proven, list in LDS), bit for bit, for every list width the templates
    serve (12 / 14 / 18 / 24 slots), systems where most queries finish near and systems where none does (gas with k above
    the cell population, tiny boxes with periodic twins, open clusters), and against the brute-force oracle"""
    from mdapy_amd import _lib

    rng = np.random.default_rng(17)
    systems = [c for c in CASES if c[0] in ("fcc_rattled", "fcc_unwrapped", "slab_open_z", "cluster_open", "triclinic_random",
                                            "random_gas", "thin_box_3cells", "dense_blob")]
    tiny = rng.random((40, 3)) * 7.0
    systems.append(("tiny_periodic", tiny, np.eye(3) * 7.0, ORG0, PBC))
    for name, pos, box, org, bnd in systems:
        x, y, z = _xyz(pos[:2500])
        n = len(x)
        for k in (1, 5, 12, 13, 14, 17, 18, 19, 24, 25):
            out = []
            for variant in (0, 1):
                _lib.lib().mdh_debug_set_knn_variant(variant)
                try:
                    idx = np.zeros((n, k), np.int32); dist = np.zeros((n, k))
                    _fast_knn.knn(x, y, z, box, org, bnd, k, idx, dist, 1)
                finally:
                    _lib.lib().mdh_debug_set_knn_variant(0)
                out.append((idx, dist))
            assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0]), (name, k)
            if k in (12, 18) and n <= 2500:
                i0 = np.zeros((n, k), np.int32); q0 = np.zeros((n, k))
                O.knn(x, y, z, box, org, bnd, k, i0, q0, 4)
                assert np.array_equal(out[0][1], q0), (name, k)


def test_knn_from_cutoff_rows_vs_oracle(monkeypatch):
    """The k nearest through the rows of a cutoff build (knn.hip k_knn_rows behind the tile kernel; large systems only, here let in
    through MDH_KNN_ROWS_MIN): distances bit for bit and ids equal to the brute-force oracle's (src/fast_knn.cpp:846-916) on rattled
    fcc / bcc, a sheared box, a slab open along z, atoms handed in box lengths away, a gas (most queries take the cell walk behind
    it) and a perfect lattice (exact ties: distances equal, ids as the cell walk lists them); the event ranges say which path ran"""
    import ctypes
    from mdapy_amd import _lib

    monkeypatch.setenv("MDH_KNN_ROWS_MIN", "1000")
    L = _lib.lib()
    rng = np.random.default_rng(41)
    systems = []
    p, b = _fcc(14, 0.06, 3)
    systems.append(("fcc_rattled", p, b, ORG0, PBC, True))
    pb, bb = lattice_positions("bcc", 2.87, 17, 17, 17)
    systems.append(("bcc_rattled", pb + rng.normal(0, 0.05, pb.shape), bb, ORG0, PBC, True))
    shear = np.array([[40.8, 0.0, 0.0], [4.1, 40.8, 0.0], [-2.0, 4.1, 40.8]])
    pt, _ = _fcc(12, 0.06, 5, a=3.4)
    org = np.array([2.0, -7.5, 11.0])
    systems.append(("sheared", (pt / 40.8) @ shear + org, shear, org, PBC, True))
    systems.append(("slab_open_z", p, b, ORG0, np.array([1, 1, 0], np.int32), True))
    pu = p + rng.integers(-2, 3, p.shape) * np.diag(b)
    systems.append(("fcc_unwrapped", pu, b, ORG0, PBC, True))
    systems.append(("gas", rng.random((12000, 3)) * 52.0, np.eye(3) * 52.0, ORG0, PBC, True))
    pp, bp = _fcc(13)
    systems.append(("fcc_perfect", pp, bp, ORG0, PBC, False))
    for name, pos, box, org, bnd, ids_too in systems:
        x, y, z = _xyz(pos)
        n = len(x)
        for k in (5, 12, 14, 18):
            i0 = np.zeros((n, k), np.int32); q0 = np.zeros((n, k)); i1 = np.zeros((n, k), np.int32); q1 = np.zeros((n, k))
            O.knn(x, y, z, box, org, bnd, k, i0, q0, 8)
            L.mdh_prof_reset(); L.mdh_prof_enable(1)
            try:
                _fast_knn.knn(x, y, z, box, org, bnd, k, i1, q1, 1)
            finally:
                L.mdh_prof_enable(0)
            buf = ctypes.create_string_buffer(4096)
            L.mdh_prof_report(buf, 4096)
            assert b"knn_rows_build" in buf.value, (name, k, buf.value)  # the rows path did take the call
            assert np.array_equal(q1, q0), (name, k)
            if ids_too:
                assert np.array_equal(i1, i0), (name, k)
            else:  # exact ties: the same rows as the cell walk alone gives
                monkeypatch.setenv("MDH_KNN_ROWS_MIN", "1000000000")
                i2 = np.zeros((n, k), np.int32); q2 = np.zeros((n, k))
                _fast_knn.knn(x, y, z, box, org, bnd, k, i2, q2, 1)
                monkeypatch.setenv("MDH_KNN_ROWS_MIN", "1000")
                assert np.array_equal(i1, i2) and np.array_equal(q1, q2), (name, k)
    # keyed (the twin's searches): a perfect bcc lattice, permuted, key = the original number -> the original system's rows
    from mdapy_amd import _order
    pos, box = lattice_positions("bcc", 3.2, 14, 13, 12)
    x, y, z = _xyz(pos)
    n = len(x)
    for k in (12, 14):
        ref_i, ref_d = np.zeros((n, k), np.int32), np.zeros((n, k))
        _fast_knn.knn(x, y, z, np.asarray(box, float), ORG0, PBC, k, ref_i, ref_d, 1)
        perm = rng.permutation(n)
        got_i, got_d = np.zeros((n, k), np.int32), np.zeros((n, k))
        _fast_knn.knn(x[perm].copy(), y[perm].copy(), z[perm].copy(), np.asarray(box, float), ORG0, PBC, k, got_i, got_d, 1, key=perm.astype(np.int64))
        rows, dist, _ = _order.translate_rows(got_i, got_d, None, perm.astype(np.int32))
        assert np.array_equal(np.asarray(dist), ref_d) and np.array_equal(np.asarray(rows), ref_i), k


def test_knn_candidate_rows_are_shared_between_the_searches_of_a_system(monkeypatch):
    """The searches of one System over the same positions — centro-symmetry (12 nearest), adaptive CNA (14), a k = 18 list — share
    the candidate rows of the first search's cutoff build where they reach (knn.py keeps them with the position columns;
    mdh_knn_keyed_rows): every result equals the one a fresh System computes alone, and the build ran for the first search and for
    the one whose radius the kept rows do not reach, not for the others."""
    import ctypes
    from mdapy_amd import _lib
    from mdapy_amd.devarray import as_numpy

    monkeypatch.setenv("MDH_KNN_ROWS_MIN", "1000")
    L = _lib.lib()
    pos, box = _fcc(14, 0.06, 3)

    def builds(fn):
        L.mdh_prof_reset(); L.mdh_prof_enable(1)
        try:
            fn()
        finally:
            L.mdh_prof_enable(0)
        buf = ctypes.create_string_buffer(4096)
        L.mdh_prof_report(buf, 4096)
        rec = {ln.split()[0]: int(ln.split()[1]) for ln in buf.value.decode().strip().splitlines() if ln}
        return rec.get("knn_rows_build", 0), rec.get("knn_rows_select", 0)

    alone = {}
    for name, fn in (("csp", lambda q: q.cal_centro_symmetry_parameter(12)), ("cna", lambda q: q.cal_common_neighbor_analysis()),
                     ("knn", lambda q: q.build_nearest_neighbor(18))):
        q = mp.System(pos=pos, box=box)
        fn(q)
        alone[name] = (q.data[name].to_numpy().copy() if name != "knn" else (as_numpy(q.verlet_list).copy(), as_numpy(q.distance_list).copy()))
    s = mp.System(pos=pos, box=box)
    assert builds(lambda: s.cal_centro_symmetry_parameter(12)) == (1, 1)
    assert builds(lambda: s.cal_common_neighbor_analysis()) == (0, 1)   # the 14 nearest from the rows of the 12-nearest search
    assert builds(lambda: s.build_nearest_neighbor(18)) == (1, 1)        # ... which do not reach the 18th neighbour: rows of its own
    v18, d18 = as_numpy(s.verlet_list).copy(), as_numpy(s.distance_list).copy()
    assert builds(lambda: s.cal_centro_symmetry_parameter(12)) == (0, 1)  # (the list is a k-nearest list: searched again, from the kept rows)
    assert np.array_equal(s.data["csp"].to_numpy(), alone["csp"]) and np.array_equal(s.data["cna"].to_numpy(), alone["cna"])
    assert np.array_equal(v18, alone["knn"][0]) and np.array_equal(d18, alone["knn"][1])
    # new positions: new columns, nothing is borrowed
    s2 = mp.System(pos=pos + 0.01, box=box)
    assert builds(lambda: s2.cal_common_neighbor_analysis()) == (1, 1)


@pytest.mark.parametrize("case", [CASES[0], CASES[5]], ids=[CASES[0][0], CASES[5][0]])
@pytest.mark.parametrize("mode", ["rc", "nnn"])
def test_steinhardt_vs_oracle(case, mode):
    _, pos, box, org, bnd = case
    x, y, z = _xyz(pos)
    N = len(x)
    if mode == "rc":
        rc = 3.4
        v, d, n = O.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 4)
        nnn = 0
    else:
        nnn = 12
        v = np.zeros((N, nnn), np.int32); d = np.zeros((N, nnn))
        O.knn(x, y, z, box, org, bnd, nnn, v, d, 4)
        n = np.full(N, nnn, np.int32)
        rc = 1e9
    ll = np.array([4, 6, 8], np.int32)
    lmax = 8
    for avg in (False, True):
        outs = []
        for be in (O, _sbo):
            qr = np.zeros((N, 3, 2 * lmax + 1)); qi = np.zeros_like(qr); qn = np.zeros((N, 9))
            be.get_sq(x, y, z, box, org, bnd, v, d, n, np.zeros((2, 2)), ll, nnn, lmax, True, True, avg, False, rc,
                      False, qr, qi, qn, 4)
            outs.append((qr, qi, qn))
        ok = np.isfinite(outs[0][2]).all(axis=1)
        assert np.array_equal(ok, np.isfinite(outs[1][2]).all(axis=1))
        for a, b in zip(outs[0], outs[1]):
            assert np.allclose(b[ok], a[ok], rtol=1e-6, atol=1e-12)
    # q_l alone (no w_l, no averaging): the stage-1 kernels write it themselves
    plain = []
    for be in (O, _sbo):
        qr = np.zeros((N, 3, 2 * lmax + 1)); qi = np.zeros_like(qr); qn = np.zeros((N, 3))
        be.get_sq(x, y, z, box, org, bnd, v, d, n, np.zeros((2, 2)), ll, nnn, lmax, False, False, False, False, rc, False, qr, qi, qn, 4)
        plain.append(qn)
    ok = np.isfinite(plain[0]).all(axis=1)
    assert np.array_equal(ok, np.isfinite(plain[1]).all(axis=1))
    assert np.allclose(plain[1][ok], plain[0][ok], rtol=1e-6, atol=1e-12)
    # solid / liquid bond counting on the averaged-off q6m
    qr, qi, qn = outs[1]
    Q6 = np.ascontiguousarray(qn[:, 1])
    res = []
    for be in (O, _sbo):
        sl = np.zeros(N, np.int32); nb = np.zeros(N, np.int32)
        be.identifySolidLiquid(1, Q6, v, d, n, qr, qi, 0.7, 7, sl, nb, False, nnn, rc, 4)
        res.append((sl, nb))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0], res[1][0])


def test_steinhardt_per_degree_kernels_equal_the_generic_one():
    """stage 1 of the Steinhardt parameters has register-resident instantiations for l = 2..8, 10, 12 (one launch per entry of
    llist); sums and their order are the generic kernel's, so q_lm, q_l, w_l agree bit for bit — cutoff and nnn lists,
    weights, a degree list that mixes compiled and uncompiled degrees (falls back as a whole)"""
    from mdapy_amd import _lib

    _, pos, box, org, bnd = CASES[1]
    x, y, z = _xyz(pos)
    N = len(x)
    v, d, n = O.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, 3.4, 4)
    w = np.random.default_rng(3).random(v.shape)
    # ([4, 6]: both degrees in ONE launch, k_sq_stage1_pair — the recurrence to l = 6 passes through l = 4)
    for ls, nnn, use_w in (([4, 6], 0, False), ([4, 6], 10, True), ([6, 4], 0, False), ([2, 3, 5, 7, 8, 10, 12], 0, True), ([6], 10, False), ([4, 9], 0, False)):
        ll = np.array(ls, np.int32)
        lmax = int(ll.max())
        # (w_l, w-hat_l, averaging) all on; all off: q_l then leaves the stage-1 kernels themselves and stage 3 is not launched
        for full in (True, False):
            outs = []
            for variant in (0, 1):
                _lib.lib().mdh_debug_set_sq_variant(variant)
                try:
                    qr = np.zeros((N, len(ls), 2 * lmax + 1)); qi = np.zeros_like(qr); qn = np.zeros((N, (3 if full else 1) * len(ls)))
                    _sbo.get_sq(x, y, z, box, org, bnd, v, d, n, w if use_w else np.zeros((2, 2)), ll, nnn, lmax, full, full, full, False,
                                3.4 if nnn == 0 else 1e9, use_w, qr, qi, qn, 1)
                    outs.append((qr, qi, qn))
                finally:
                    _lib.lib().mdh_debug_set_sq_variant(0)
            for a, b_ in zip(*outs):
                assert a.tobytes() == b_.tobytes(), (ls, full)
            assert np.isfinite(outs[0][2]).any()


def _solid_liquid_serial(v, d, n, q, thr, n_bond, rc):
    """the reference's identifySolidLiquid run with ONE thread (src/steinhardt_bond_orientation.cpp:605-674): bonds counted per
    atom, then the in-place sweep in index order that turns a solid atom without a solid neighbour into a liquid one — an
    atom visited later sees the labels the sweep has already changed.  q: (N, 13) real q6m vectors with |q| = 1 and
    Q6 = sqrt(4 pi / 13), so that s_ij is the plain dot product."""
    N = len(n)
    nb = np.zeros(N, np.int32)
    for i in range(N):
        for jj in range(n[i]):
            j = v[i, jj]
            if j < 0 or d[i, jj] > rc:
                continue
            if float(q[i] @ q[j]) > thr:
                nb[i] += 1
    sl = (nb >= n_bond).astype(np.int32)
    for i in range(N):
        if sl[i] == 1 and not any(v[i, jj] >= 0 and sl[v[i, jj]] == 1 for jj in range(n[i])):
            sl[i] = 0
    return sl, nb


def _run_solid_liquid(v, d, n, q, thr, n_bond, rc):
    N = len(n)
    qr = np.zeros((N, 1, 13)); qr[:, 0, :] = q
    qi = np.zeros_like(qr)
    Q6 = np.full(N, np.sqrt(4 * np.pi / 13))
    out = []
    for be in (_sbo, O):
        sl, nb = np.zeros(N, np.int32), np.zeros(N, np.int32)
        be.identifySolidLiquid(0, Q6, v, d, n, qr, qi, thr, n_bond, sl, nb, False, 0, rc, 1)
        out.append((sl, nb))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    return out[0]


def test_solid_liquid_second_pass_equals_the_serial_sweep():
    """a16, pass 2.  The kernel judges every atom against the labels of pass 1 (no race, no order dependence); the
    reference's serial sweep reads labels it is changing.  For symmetric lists (cutoff lists, Voronoi lists) the two agree
    — an atom's solid neighbour cannot have been cleared before it, because the atom itself was that neighbour's solid
    neighbour.  Checked on a hand-built graph with two adjacent isolated solids and a lone one, and on random symmetric
    graphs; the one-sided lists of the nnn mode are where they can part, shown on the smallest example."""
    e0 = np.zeros(13); e0[0] = 1.0          # "crystalline" orientation
    def other(k):                           # mutually orthogonal, orthogonal to e0
        u = np.zeros(13); u[1 + k % 12] = 1.0
        return u
    # A(0) - B(1) adjacent, each with two aligned leaves; C(6) alone with three aligned leaves; n_bond = 3 aligned neighbours
    edges = [(0, 1), (0, 2), (0, 3), (1, 4), (1, 5), (6, 7), (6, 8), (6, 9)]
    N = 10
    rows = [[] for _ in range(N)]
    for a_, b_ in edges:
        rows[a_].append(b_); rows[b_].append(a_)
    M = max(len(r) for r in rows)
    v = np.full((N, M), -1, np.int32); d = np.full((N, M), 9.0)
    n = np.array([len(r) for r in rows], np.int32)
    for i, r in enumerate(rows):
        v[i, :len(r)] = r; d[i, :len(r)] = 1.0
    q = np.tile(e0, (N, 1))
    sl, nb = _run_solid_liquid(v, d, n, q, 0.7, 3, 2.0)
    ref_sl, ref_nb = _solid_liquid_serial(v, d, n, q, 0.7, 3, 2.0)
    assert np.array_equal(nb, ref_nb) and np.array_equal(sl, ref_sl)
    assert list(sl) == [1, 1, 0, 0, 0, 0, 0, 0, 0, 0] and list(nb[[0, 1, 6]]) == [3, 3, 3]  # the pair survives, the lone one does not
    # random symmetric graphs, random orientations from a small set, both orders of the atoms
    rng = np.random.default_rng(8)
    for trial in range(20):
        N = 400
        pairs = rng.integers(0, N, (900, 2))
        rows = [set() for _ in range(N)]
        for a_, b_ in pairs:
            if a_ != b_:
                rows[a_].add(int(b_)); rows[b_].add(int(a_))
        M = max(1, max(len(r) for r in rows))
        v = np.full((N, M), -1, np.int32); d = np.full((N, M), 9.0); n = np.zeros(N, np.int32)
        for i, r in enumerate(rows):
            r = sorted(r); rng.shuffle(r)
            v[i, :len(r)] = r; d[i, :len(r)] = rng.uniform(0.5, 2.5, len(r)); n[i] = len(r)
        # distances must be symmetric too, or the cutoff makes the graph one-sided
        for i in range(N):
            for jj in range(n[i]):
                j = v[i, jj]
                d[i, jj] = 0.5 + ((i * 7919 + j * 7919) % 2000) / 1000.0
        kinds = rng.integers(0, 3, N)
        q = np.array([e0 if k == 0 else other(k + i % 5) for i, k in enumerate(kinds)])
        for n_bond in (1, 2, 3):
            sl, nb = _run_solid_liquid(v, d, n, q, 0.7, n_bond, 2.0)
            ref_sl, ref_nb = _solid_liquid_serial(v, d, n, q, 0.7, n_bond, 2.0)
            assert np.array_equal(nb, ref_nb) and np.array_equal(sl, ref_sl)
    # one-sided lists (nnn mode): atom 0 lists nobody solid, atom 1 lists only atom 0.  The serial sweep clears 0 first and
    # then 1; against the pass-1 labels atom 1 keeps its solid neighbour.  The kernel (and the oracle) do the latter.
    v = np.array([[2, 3], [0, 0], [0, -1], [0, -1]], np.int32)
    d = np.ones((4, 2)); n = np.array([2, 2, 1, 1], np.int32)
    q = np.tile(e0, (4, 1))
    sl, nb = _run_solid_liquid(v, d, n, q, 0.7, 2, 2.0)
    ser, _ = _solid_liquid_serial(v, d, n, q, 0.7, 2, 2.0)
    assert list(nb) == [2, 2, 1, 1] and list(sl) == [0, 1, 0, 0] and list(ser) == [0, 0, 0, 0]


def test_streaming_rdf_tile_kernel_edge_cases():
    """the LDS-tile kernel of the streaming RDF (half shell, single-precision sorting, exact re-binning near shell
    boundaries) against the CPU oracle and against the thread-per-atom kernel: open axes, three cells on a periodic axis,
    atoms handed in outside the box, three species, cells that hold more centre atoms than one batch, many bins"""
    from mdapy_amd import _lib

    rng = np.random.default_rng(31)
    jobs = []
    pos, box = lattice_positions("fcc", 4.0, 9, 9, 9)
    pos = pos + rng.normal(0, 0.3, pos.shape)
    ty3 = rng.integers(0, 3, len(pos)).astype(np.int32)
    jobs.append(("pbc_3types", pos, box, ORG0, PBC, ty3, 6.0, 150))
    jobs.append(("open_z", pos, box, ORG0, np.array([1, 1, 0], np.int32), ty3, 6.0, 77))
    jobs.append(("cluster", pos, box, ORG0, np.array([0, 0, 0], np.int32), ty3, 11.9, 40))   # three cells across
    far = pos + rng.integers(-1, 2, pos.shape) * 36.0                                          # whole box lengths away
    jobs.append(("unwrapped_shifted", far + 5.0, box, np.array([5.0, 5.0, 5.0]), PBC, ty3, 7.5, 200))
    dense = rng.random((9000, 3)) * np.array([30.0, 30.0, 30.0])
    jobs.append(("dense_gas", dense, np.eye(3) * 30.0, ORG0, PBC, rng.integers(0, 2, 9000).astype(np.int32), 9.9, 500))  # ~330 atoms per cell
    edge = np.concatenate([np.arange(0, 200)[:, None] * np.array([[0.04, 0, 0]]) + 10.0, [[10.0, 10.0, 10.0]]])      # distances exactly on shell boundaries
    jobs.append(("on_the_boundaries", edge, np.eye(3) * 40.0, ORG0, PBC, np.zeros(len(edge), np.int32), 8.0, 200))
    # triclinic boxes periodic along all three vectors take the tile kernel too (cell edge vectors instead of widths, a band
    # widened with the shear); an open triclinic box keeps the thread-per-atom kernel — same counts either way
    for tag, shear in (("tri_mild", 0.15), ("tri_strong", 0.45)):
        H = np.array([[36.0, 0, 0], [shear * 36.0, 36.0, 0], [-0.5 * shear * 36.0, shear * 36.0, 36.0]])
        frac = pos / 36.0
        ptri = frac @ H + rng.integers(-1, 2, pos.shape) @ H * (tag == "tri_strong")  # (the strong one also handed in unwrapped)
        jobs.append((tag, ptri + 2.0, H, np.array([2.0, 2.0, 2.0]), PBC, ty3, 6.5, 120))
    jobs.append(("tri_open_x", (pos / 36.0) @ np.array([[36.0, 0, 0], [7.0, 36.0, 0], [0, 5.0, 36.0]]), np.array([[36.0, 0, 0], [7.0, 36.0, 0], [0, 5.0, 36.0]]),
                 ORG0, np.array([0, 1, 1], np.int32), ty3, 6.0, 60))
    for name, p, bx, org, bd, ty, rc, nbin in jobs:
        x, y, z = _xyz(p)
        nt = int(ty.max()) + 1
        g0 = np.zeros((nt, nt, nbin))
        O._rdf_streaming(x, y, z, ty, bx, org, bd, g0, rc, nbin, 8)
        got = []
        for variant in (0, 1):
            _lib.lib().mdh_debug_set_rdf_variant(variant)
            try:
                g1 = np.zeros((nt, nt, nbin))
                _rdf._rdf_streaming(x, y, z, ty, bx, org, bd, g1, rc, nbin, 1)
                got.append(g1)
            finally:
                _lib.lib().mdh_debug_set_rdf_variant(0)
        assert np.array_equal(got[0], g0), name
        assert np.array_equal(got[1], g0), name
        assert g0.sum() > 0


def test_rdf_and_wcp_vs_oracle():
    rng = np.random.default_rng(21)
    pos, box = lattice_positions("fcc", 4.0, 9, 9, 9)
    pos = pos + rng.normal(0, 0.35, pos.shape)
    typ = (rng.random(len(pos)) < 0.36).astype(np.int32)
    x, y, z = _xyz(pos)
    for bnd in (PBC, np.array([1, 0, 1], np.int32)):
        for rc, nbin in ((8.0, 200), (13.0, 50)):  # 13 A > L/3: all-pairs branch
            g0 = np.zeros((2, 2, nbin)); g1 = np.zeros((2, 2, nbin))
            O._rdf_streaming(x, y, z, typ, box, ORG0, bnd, g0, rc, nbin, 4)
            _rdf._rdf_streaming(x, y, z, typ, box, ORG0, bnd, g1, rc, nbin, 1)
            assert np.array_equal(g1, g0)  # integer counts
    v, d, n = O.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, 5.0, 4)
    g0 = np.zeros((2, 2, 60)); g1 = np.zeros((2, 2, 60))
    O._rdf(v, d, n, typ, g0, 5.0, 60); _rdf._rdf(v, d, n, typ, g1, 5.0, 60)
    assert np.array_equal(g1, g0)
    h0 = np.zeros(60); h1 = np.zeros(60)
    O._rdf_single_species(v, d, n, h0, 5.0, 60); _rdf._rdf_single_species(v, d, n, h1, 5.0, 60)
    assert np.array_equal(h1, h0)
    w0 = np.zeros((2, 2)); w1 = np.zeros((2, 2))
    O.get_wcp(v, n, typ, 2, w0); _wcp.get_wcp(v, n, typ, 2, w1, 1)
    assert np.array_equal(w1, w0)


def test_small_helpers_vs_oracle():
    rng = np.random.default_rng(3)
    tri = np.array([[12.0, 0.0, 0.0], [3.0, 11.0, 0.0], [-2.0, 1.0, 10.0]])
    for box in (np.diag([9.0, 10.0, 11.0]), tri):
        p = rng.normal(0, 25, (2000, 3))
        a = [np.ascontiguousarray(p[:, k]) for k in range(3)]
        b = [c.copy() for c in a]
        O.wrap_positions(*a, box, np.array([1.0, 2.0, 3.0]), np.array([1, 0, 1], np.int32), 4)
        _neighbor.wrap_positions(*b, box, np.array([1.0, 2.0, 3.0]), np.array([1, 0, 1], np.int32), 1)
        assert all(np.array_equal(u, w) for u, w in zip(a, b))
        old = rng.random((7, 3)) @ box
        n0 = np.zeros(7 * 24 * 3); n1 = np.zeros(7 * 24 * 3)
        O.repeat_cell(n0, box, old, 2, 3, 4); _repeat_cell.repeat_cell(n1, box, old, 2, 3, 4, 1)
        assert np.array_equal(n1, n0)
    pos, bx = _fcc(5, 0.1, 9)
    x, y, z = _xyz(pos)
    v, d, n = O.build_neighbor_without_max_neigh(x, y, z, bx, ORG0, PBC, 4.0, 4)
    o0 = np.zeros(len(x)); o1 = np.zeros(len(x))
    O.average_by_neighbor(3.0, v, d, n, x, o0, True, 4)
    _neighbor.average_by_neighbor(3.0, v, d, n, x, o1, True, 1)
    assert np.array_equal(o1, o0)


# ------------------------------------------------------------------ golden vectors, full System flow on the GPU
CNA_PATHS, CSP_PATHS, QL_PATHS, IDS_PATHS = (fixtures_with(k) for k in ("cna", "csp", "q4", "ids"))


@pytest.mark.parametrize("path", CNA_PATHS, ids=ids_of(CNA_PATHS))
def test_golden_cna(path):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_common_neighbor_analysis(rc=float(d["cna_cutoff"]))
    assert np.array_equal(s.data["cna"].to_numpy(), d["cna"])


@pytest.mark.parametrize("path", CSP_PATHS, ids=ids_of(CSP_PATHS))
def test_golden_csp(path):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_centro_symmetry_parameter(int(d["csp_num_neighbors"]))
    assert np.allclose(s.data["csp"].to_numpy(), d["csp"], atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("path", QL_PATHS, ids=ids_of(QL_PATHS))
def test_golden_ql(path):
    d = np.load(path)
    s = system_from_fixture(d)
    rc = float(d["ql_cutoff"])
    s.cal_steinhardt_bond_orientation([4, 6], rc=rc)
    for l in (4, 6):
        assert np.allclose(s.data[f"ql{l}"].to_numpy(), d[f"q{l}"], atol=1e-6, rtol=1e-6)
    s.cal_steinhardt_bond_orientation([4, 6], rc=rc, average=True)
    for l in (4, 6):
        assert np.allclose(s.data[f"ql{l}"].to_numpy(), d[f"q{l}_avg"], atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("path", IDS_PATHS, ids=ids_of(IDS_PATHS))
def test_golden_ids(path):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_identify_diamond_structure()
    assert np.array_equal(s.data["ids"].to_numpy(), d["ids"])


def test_golden_rdf_wcp_average():
    d = misc("rdf")
    s = mp.System(input_path("AlCrNi.xyz"))
    rdf = s.cal_radial_distribution_function(float(d["cutoff"]), int(d["nbins"]))
    el = list(d["elements"])
    for i in range(len(el)):
        for j in range(i, len(el)):
            assert np.allclose(rdf.g_partial[(el[i], el[j])], d["g"][i, j], atol=1e-6)
    s = mp.System(input_path("CoCuFeNiPd-4M.dump"))
    wcp = s.cal_warren_cowley_parameter(rc=3.0)
    ref = np.array([[-1.39, 0.64, 0.39, -0.3, 0.66], [0.64, -1.94, 0.58, 0.51, 0.2], [0.39, 0.58, -0.56, 0.63, -1.04],
                    [-0.3, 0.51, 0.63, -1.69, 0.85], [0.66, 0.2, -1.04, 0.85, -0.67]])
    assert np.allclose(wcp.WCP.round(2), ref)
    d = misc("average_neighbor")
    for name in ("rec_box_big", "tri_box_big"):
        s = mp.System(input_path(f"{name}.xyz"))
        s.average_by_neighbor(float(d[f"{name}__cutoff"]), "x", include_self=True)
        assert np.allclose(s.data["x_ave"].to_numpy(), d[f"{name}__x_ave"], atol=1e-6)


def test_system_flow_perfect_crystals_and_errors():
    a = 4.05
    fcc = mp.build_crystal("Al", "fcc", a, nx=4, ny=4, nz=4)
    fcc.cal_common_neighbor_analysis(rc=0.854 * a)
    assert np.all(fcc.data["cna"].to_numpy() == 1)
    fcc.cal_common_neighbor_analysis()
    assert np.all(fcc.data["cna"].to_numpy() == 1)
    fcc.cal_centro_symmetry_parameter(12)
    assert np.allclose(fcc.data["csp"].to_numpy(), 0.0, atol=1e-10)
    fcc.cal_steinhardt_bond_orientation([4, 6], rc=0.95 * a)
    assert np.allclose(fcc.data["ql4"].to_numpy(), 0.190941, atol=1e-5)
    assert np.allclose(fcc.data["ql6"].to_numpy(), 0.574524, atol=1e-5)
    bcc = mp.build_crystal("Fe", "bcc", 2.86, nx=6, ny=6, nz=6)
    bcc.cal_common_neighbor_analysis(rc=1.21 * 2.86)
    assert np.all(bcc.data["cna"].to_numpy() == 3)
    with pytest.raises(ValueError, match="max_neigh=5 is too small"):
        bcc.build_neighbor(3.0, max_neigh=5)
    with pytest.raises(RuntimeError, match="volume of the box is zero"):
        x = np.zeros(4); v = np.full((4, 2), -1, np.int32); dd = np.zeros((4, 2)); nn = np.zeros(4, np.int32)
        _neighbor.build_neighbor(x, x, x, np.array([[1.0, 1, 0], [2, 2, 0], [0, 0, 1]]), ORG0, PBC, 1.0, v, dd, nn, 1)


# ------------------------------------------------------------------ PTM (a13): HIP kernel vs oracle/_ref, the reference's own library
def test_neighbor_cell_window_hint_same_rows_and_broken_promise_is_reported():
    """mdh_hint_cell_window (a rank's slab of a decomposed system): the build's passes over the cells of the global grid and
    the tile kernel's range cover the promised planes only — rows, counts, distances bit-identical to the build without the
    hint, with and without an ordering key; an atom outside the promised window is counted where the atoms are binned and the
    NEXT build of the thread refuses loudly."""
    import torch

    pos, box = lattice_positions("fcc", 3.615, 40, 6, 6)
    pos = pos + np.random.default_rng(8).normal(0, 0.05, pos.shape)
    L = box[0][0] if np.ndim(box) == 2 else box[0]
    f = (pos[:, 0] / L) % 1.0
    pos = pos[(f >= 0.30) & (f < 0.52)]
    N = len(pos)
    rc, M = 0.854 * 3.615, 18
    dev = torch.device("cuda", 0)
    x, y, z = (torch.from_numpy(np.ascontiguousarray(pos[:, k])).to(dev) for k in range(3))
    key = torch.from_numpy(np.random.default_rng(3).permutation(N).astype(np.int64) + 7).to(dev)

    def build(hint, k=None):
        v = torch.empty((N, M), dtype=torch.int32, device=dev)
        d = torch.empty((N, M), dtype=torch.float64, device=dev)
        n = torch.empty((N,), dtype=torch.int32, device=dev)
        if hint is not None:
            _neighbor.hint_cell_window(0, *hint)
        _neighbor.build_neighbor(x, y, z, box, ORG0, PBC, rc, v, d, n, 1, fill_pads=True, key=k)
        torch.cuda.synchronize()
        return v.cpu().numpy(), d.cpu().numpy(), n.cpu().numpy()

    for k in (None, key):
        ref = build(None, k)
        assert ref[2].max() >= 12
        for _ in range(2):
            got = build((0.28, 0.54), k)
            assert all(np.array_equal(a, b_) for a, b_ in zip(got, ref))
    build((0.40, 0.54))  # a third of the atoms lie below the promised window: counted on the device ...
    with pytest.raises(ValueError, match="outside the cell window"):
        build(None)      # ... and reported by the next build
    got = build((0.28, 0.54))  # the thread is usable again
    assert all(np.array_equal(a, b_) for a, b_ in zip(got, build(None)))
    # a broken promise on the paths where the thread-per-atom kernel takes the whole call (rows wider than the tile kernel's 128
    # slots): the records behind the atoms binned are never written, so no cell may span them and no centre may be read from
    # them — every id written must be an atom's, and the atoms inside the window keep their full rows
    Mw = 130
    vw = torch.full((N, Mw), -7, dtype=torch.int32, device=dev); dw = torch.zeros((N, Mw), dtype=torch.float64, device=dev)
    nw = torch.full((N,), -7, dtype=torch.int32, device=dev)
    _neighbor.hint_cell_window(0, 0.40, 0.54)
    _neighbor.build_neighbor(x, y, z, box, ORG0, PBC, rc, vw, dw, nw, 1, fill_pads=True)
    torch.cuda.synchronize()
    vw, nw = vw.cpu().numpy(), nw.cpu().numpy()
    with pytest.raises(ValueError, match="outside the cell window"):
        build(None)
    ref_v, _, ref_n = build(None)
    f = (pos[:, 0] / L) % 1.0
    deep = (f >= 0.44) & (f < 0.50)  # atoms whose whole neighbourhood lies inside the promised planes
    assert deep.sum() > 100 and np.array_equal(nw[deep], ref_n[deep])
    assert np.array_equal(vw[deep][:, :M], ref_v[deep]) and (vw[deep][:, M:] == -1).all()
    assert ((vw >= -1) | (vw == -7)).all() and (vw < N).all() and ((nw == -7) | ((nw >= 0) & (nw <= Mw))).all()


from _ptm_cases import compare_ptm, ptm_cases
from mdapy_amd import _ptm

PTM_CASES = ptm_cases()
needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libptm_ref.so missing")


@needs_ref
@pytest.mark.parametrize("case", PTM_CASES, ids=[c[0] for c in PTM_CASES])
def test_ptm_vs_reference_library(case):
    name, pos, box, boundary, structure, types, thr = case
    N = len(pos)
    x, y, z = _xyz(pos)
    bd = np.array(boundary, np.int32)
    k = min(18, N - 1)
    idx, dist = np.zeros((N, k), np.int32), np.zeros((N, k))
    O.knn(x, y, z, box, ORG0, bd, k, idx, dist, 4)
    out_r, ind_r = np.zeros((N, 8)), np.zeros((N, 18), np.int32)
    O.get_ptm(structure, x, y, z, box, ORG0, bd, idx, types, thr, out_r, ind_r)
    out_g, ind_g = np.full((N, 8), 7.0), np.full((N, 18), 7, np.int32)
    _ptm.get_ptm(structure, x, y, z, box, ORG0, bd, idx, types, thr, out_g, ind_g)
    compare_ptm(out_g, ind_g, out_r, ind_r)


def test_ptm_ordering_second_pass_gives_the_same_rows():
    """the ordering kernel's second pass (28-vertex polygons, for faces that outgrow the first pass's 15): with the first
    pass shrunk to 5 vertices most atoms take it, and every output must stay what it was"""
    from mdapy_amd import _lib

    pos, box = _fcc(8, 0.08, 3)
    x, y, z = _xyz(pos)
    N = len(x)
    idx, dist = np.zeros((N, 18), np.int32), np.zeros((N, 18))
    _fast_knn.knn(x, y, z, box, ORG0, PBC, 18, idx, dist, 1)
    outs = []
    for cap in (15, 10, 5, -15, -10, -5): # negative: the polygons in space instead of in their plane's coordinates (A/B form)
        _lib.lib().mdh_debug_set_ptm_order_cap(cap)
        try:
            o, i = np.zeros((N, 8)), np.zeros((N, 18), np.int32)
            _ptm.get_ptm("all", x, y, z, box, ORG0, PBC, idx, None, 0.1, o, i)
            outs.append((o, i))
        finally:
            _lib.lib().mdh_debug_set_ptm_order_cap(0)  # back to the automatic choice
    for o, i in outs[1:3]:
        assert np.array_equal(outs[0][0], o) and np.array_equal(outs[0][1], i)
    for o, i in outs[4:]:
        assert np.array_equal(outs[3][0], o) and np.array_equal(outs[3][1], i)
    # 2-D against 3-D polygons: the same faces up to rounding; on a rattled lattice no two areas are that close, so the order
    # of the rows — and with it every output — is the same
    assert np.array_equal(outs[0][1], outs[3][1]) and np.array_equal(outs[0][0], outs[3][0])
    assert (outs[0][0][:, 0] == 1).mean() > 0.9


@pytest.mark.parametrize("kind", ["gas", "fcc_hot", "bcc_rattled", "hcp_perfect", "fcc_perfect", "liquid_like"])
def test_ptm_plane_polygons_against_space_polygons_on_random_systems(kind):
    """the ordering kernel keeps a face's polygon in the 2-D coordinates of the face's plane; the 3-D form of rounds 1-2 stays
    selectable (negative cap).  Same cell, same faces, different rounding: structure types must agree everywhere and the
    rows everywhere but at exact-tie neighbourhoods (perfect lattices: equal areas, whose order is decided by the last bit —
    the same class of difference the reference's voro++ has against either form)"""
    from mdapy_amd import _lib

    rng = np.random.default_rng(97)
    if kind == "gas":
        box = np.diag([40.0, 38.0, 42.0]); pos = rng.random((6000, 3)) * np.diag(box)
    elif kind == "liquid_like": # jittered grid: no close pairs, wide spread of face sizes
        g = np.stack(np.meshgrid(*[np.arange(16)] * 3, indexing="ij"), -1).reshape(-1, 3) * 2.6
        pos = g + rng.uniform(-1.0, 1.0, g.shape); box = np.diag([16 * 2.6] * 3)
    elif kind == "fcc_hot":
        pos, box = _fcc(9, 0.25, 5)
    elif kind == "bcc_rattled":
        pos, box = lattice_positions("bcc", 2.87, 11, 11, 11); pos = pos + rng.normal(0, 0.07, pos.shape)
    elif kind == "hcp_perfect":
        pos, box = lattice_positions("hcp", 3.2, 8, 8, 8)
    else:
        pos, box = _fcc(8)
    x, y, z = _xyz(pos)
    N = len(x)
    idx, dist = np.zeros((N, 18), np.int32), np.zeros((N, 18))
    _fast_knn.knn(x, y, z, box, ORG0, PBC, 18, idx, dist, 1)
    res = {}
    for cap in (10, -10):
        _lib.lib().mdh_debug_set_ptm_order_cap(cap)
        try:
            o, i = np.zeros((N, 8)), np.zeros((N, 18), np.int32)
            _ptm.get_ptm("all", x, y, z, box, ORG0, PBC, idx, None, 0.1, o, i)
            res[cap] = (o, i)
        finally:
            _lib.lib().mdh_debug_set_ptm_order_cap(0)  # back to the automatic choice
    (o2, i2), (o3, i3) = res[10], res[-10]
    assert np.array_equal(o2[:, 0], o3[:, 0])  # structure type
    same = (i2 == i3).all(axis=1)
    if "perfect" not in kind:
        assert same.mean() > 0.999, same.mean()
    assert np.allclose(o2[same], o3[same], rtol=0, atol=1e-6)  # (the contract of the PTM floats; a perfect lattice has rmsd = sqrt(rounding noise) ~ 1e-8)
    assert np.allclose(o2[:, 2], o3[:, 2], rtol=0, atol=1e-6)  # rmsd, whatever the labelling


def test_ptm_first_ordering_pass_chosen_from_a_stale_count():
    """the first ordering pass uses eight-vertex polygons when the previous call with the same number of atoms counted few
    atoms with larger faces (a crystal), ten otherwise; the count travels through pinned memory.  A crystal twice (the second
    call runs with eight), then a gas of the same size (still eight, by the stale count: half its atoms take the second
    pass), then the gas again (ten): every result equals the one with the pass size forced."""
    from mdapy_amd import _lib

    rng = np.random.default_rng(23)
    cry, box = _fcc(9, 0.05, 6)
    gas = rng.random(cry.shape) * np.diag(np.asarray(box, float) if np.ndim(box) == 2 else np.diag(box))
    L = _lib.lib()

    def run(pos, cap):
        L.mdh_debug_set_ptm_order_cap(cap)
        x, y, z = _xyz(pos)
        N = len(x)
        idx, dist = np.zeros((N, 18), np.int32), np.zeros((N, 18))
        _fast_knn.knn(x, y, z, box, ORG0, PBC, 18, idx, dist, 1)
        o, i = np.zeros((N, 8)), np.zeros((N, 18), np.int32)
        _ptm.get_ptm("fcc-hcp-bcc", x, y, z, box, ORG0, PBC, idx, None, 0.1, o, i)
        return o, i

    try:
        ref_c, ref_g = run(cry, 10), run(gas, 10)
        L.mdh_debug_set_ptm_order_cap(0)
        for pos, ref in ((cry, ref_c), (cry, ref_c), (gas, ref_g), (gas, ref_g), (cry, ref_c)):
            o, i = run(pos, 0)
            assert np.array_equal(o, ref[0]) and np.array_equal(i, ref[1])
    finally:
        L.mdh_debug_set_ptm_order_cap(0)


PTM_PATHS = fixtures_with("ptm")


# reference: tests/test_polyhedral_template_matching.py:21-31
@pytest.mark.parametrize("path", PTM_PATHS, ids=ids_of(PTM_PATHS))
def test_golden_ptm(path):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_polyhedral_template_matching()
    assert np.array_equal(s.data["ptm"].to_numpy(), d["ptm"])


def test_ptm_system_flow_and_errors():
    fcc = mp.build_crystal("Al", "fcc", 4.05, nx=4, ny=4, nz=4)
    fcc.cal_polyhedral_template_matching(return_rmsd=True, return_atomic_distance=True, return_orientation=True)
    assert np.all(fcc.data["ptm"].to_numpy() == 1)
    assert np.allclose(fcc.data["interatomic_distance"].to_numpy(), 4.05 / 2 ** 0.5, rtol=1e-9)
    assert np.allclose(np.abs(fcc.data["qw"].to_numpy()), 1.0, atol=1e-6)
    bcc = mp.build_crystal("Fe", "bcc", 2.86, nx=4, ny=4, nz=4)
    bcc.cal_polyhedral_template_matching()
    assert np.all(bcc.data["ptm"].to_numpy() == 3)
    hcp = mp.build_crystal("Mg", "hcp", 3.21, nx=4, ny=4, nz=3)
    hcp.cal_polyhedral_template_matching()
    assert np.all(hcp.data["ptm"].to_numpy() == 2)
    dia = mp.build_crystal("C", "diamond", 3.5, nx=3, ny=3, nz=3)  # reference: tests/test_polyhedral_template_matching.py:53-58
    dia.cal_polyhedral_template_matching(structure="all")
    assert np.all(dia.data["ptm"].to_numpy() == 6)
    # device-resident call: same answer with HBM-resident inputs
    import torch
    pos, box = _fcc(6, 0.05, 4)
    x, y, z = _xyz(pos)
    idx, dist = np.zeros((len(pos), 18), np.int32), np.zeros((len(pos), 18))
    O.knn(x, y, z, box, ORG0, PBC, 18, idx, dist, 4)
    out_h, ind_h = np.zeros((len(pos), 8)), np.zeros((len(pos), 18), np.int32)
    _ptm.get_ptm("default", x, y, z, box, ORG0, PBC, idx, None, 0.1, out_h, ind_h)
    dev = [torch.from_numpy(a).cuda() for a in (x, y, z, idx)]
    out_d = torch.zeros((len(pos), 8), dtype=torch.float64, device="cuda")
    ind_d = torch.zeros((len(pos), 18), dtype=torch.int32, device="cuda")
    _ptm.get_ptm("default", dev[0], dev[1], dev[2], box, ORG0, PBC, dev[3], None, 0.1, out_d, ind_d)
    assert np.array_equal(out_d.cpu().numpy(), out_h) and np.array_equal(ind_d.cpu().numpy(), ind_h)


def test_wcp_counts_extension():
    """mdh_wcp_counts (multi-GPU building block): integer reductions over a row mask vs numpy, and consistency with get_wcp"""
    rng = np.random.default_rng(3)
    pos, box = _fcc(7, 0.05, 6)
    x, y, z = _xyz(pos)
    N, M, rc, T = len(pos), 16, 0.854 * 3.615, 3
    v, d, nn = np.full((N, M), -1, np.int32), np.full((N, M), rc + 1.0), np.zeros(N, np.int32)
    _neighbor.build_neighbor(x, y, z, box, ORG0, PBC, rc, v, d, nn, 1)
    ty = rng.integers(0, T, N).astype(np.int32)
    rows = (rng.random(N) < 0.6).astype(np.uint8)
    for mask in (None, rows):
        got = np.zeros(T * T + 2 * T, np.int64)
        _wcp.get_wcp_counts(v, nn, ty, T, got, rows=mask)
        exp = np.zeros_like(got)
        for i in range(N):
            if mask is not None and not mask[i]:
                continue
            exp[T * T + T + ty[i]] += 1
            exp[T * T + ty[i]] += nn[i]
            np.add.at(exp, ty[i] * T + ty[v[i, : nn[i]]], 1)
        assert np.array_equal(got, exp)
    w = np.zeros((T, T))
    _wcp.get_wcp(v, nn, ty, T, w, 1)
    full = np.zeros(T * T + 2 * T, np.int64)
    _wcp.get_wcp_counts(v, nn, ty, T, full)
    zmn, zm, cnt = full[: T * T].reshape(T, T), full[T * T: T * T + T], full[T * T + T:]
    w2 = 1.0 - zmn / ((cnt / N)[None, :] * zm[:, None])
    assert np.allclose(w, w2, rtol=0, atol=1e-15)


# ------------------------------------------------------------------ list consumers: AJA, CNP, structure entropy
from mdapy_amd import _aja, _cnp, _structure_entropy


@pytest.mark.parametrize("case", ["fcc_rattled", "fcc_hot_shifted_origin", "triclinic_random", "random_gas", "slab_open_z"])
def test_aja_cnp_entropy_vs_oracle(case):
    name, pos, box, origin, bd = next(c for c in _cases() if c[0] == case)
    x, y, z = _xyz(pos)
    N = len(pos)
    # AJA on 14 nearest neighbours
    idx, dk = np.zeros((N, 14), np.int32), np.zeros((N, 14))
    O.knn(x, y, z, box, origin, bd, 14, idx, dk, 4)
    a0, a1 = np.zeros(N, np.int32), np.full(N, 9, np.int32)
    O.compute_aja(x, y, z, box, origin, bd, idx, dk, a0, 4)
    _aja.compute_aja(x, y, z, box, origin, bd, idx, dk, a1, 1)
    assert np.array_equal(a1, a0)
    # CNP + entropy on a cutoff list
    rc = 3.2
    v, d, nn = O.build_neighbor_without_max_neigh(x, y, z, box, origin, bd, rc, 4)
    c0, c1 = np.zeros(N), np.full(N, -1.0)
    O.compute_cnp(x, y, z, box, origin, bd, v, d, nn, c0, rc, 4)
    _cnp.compute_cnp(x, y, z, box, origin, bd, v, d, nn, c1, rc, 1)
    assert np.allclose(c1, c0, rtol=1e-6, atol=1e-9)
    vol = abs(np.linalg.det(np.asarray(box, float) if np.ndim(box) == 2 else np.diag(box)))
    for local in (False, True):
        e0, e1 = np.zeros(N), np.full(N, -1.0)
        O.calculate_structure_entropy(rc, 0.2, local, vol, d, nn, e0, 4)
        _structure_entropy.calculate_structure_entropy(rc, 0.2, local, vol, d, nn, e1, 1)
        ok = np.isfinite(e0)
        assert np.array_equal(np.isfinite(e1), ok)
        assert np.allclose(e1[ok], e0[ok], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("sigma", [0.2, 0.5, 0.14, 0.1])
def test_structure_entropy_ladder_and_direct_kernels_vs_oracle(sigma):
    """rc 5.0 on rattled fcc (42 neighbours; rows of 50 and of the exact width): the ladder kernel (<= 40 bins; 26, 11 and 36 here)
    with 1, 2, 4 and 8 lanes to a row and the direct kernel (sigma 0.1: 51 bins, the only one) against the oracle.  The bar is
    rtol 1e-6; the ladder's rounding (DESIGN: <= 3e-13 on the sums) is checked at 1e-10."""
    from mdapy_amd import _lib
    pos, box = _fcc(9, 0.08, 77)
    x, y, z = _xyz(pos)
    N, rc = len(x), 5.0
    vol = abs(np.linalg.det(np.diag(box) if np.ndim(box) == 1 else np.asarray(box, float)))
    v, d, nn = O.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, rc, 4)
    wide = np.full((N, 50), rc + 1.0); wide[:, :d.shape[1]] = d
    try:
        for dist in (d, wide, d[:67], wide[:1]):
            n = np.ascontiguousarray(nn[:len(dist)])
            for local in (False, True):
                e0 = np.zeros(len(dist)); O.calculate_structure_entropy(rc, sigma, local, vol, dist, n, e0, 4)
                for variant in (0, 1, 2, 3, 4, 5):
                    _lib.check(_lib.lib().mdh_debug_set_entropy_variant(variant))
                    e1 = np.full(len(dist), -1.0)
                    _structure_entropy.calculate_structure_entropy(rc, sigma, local, vol, dist, n, e1, 1)
                    assert np.allclose(e1, e0, rtol=1e-10, atol=1e-12), (sigma, dist.shape, local, variant, np.abs(e1 / e0 - 1).max())
    finally:
        _lib.check(_lib.lib().mdh_debug_set_entropy_variant(0))


@pytest.mark.parametrize("cells,sigma", [(9, 0.08), (2, 0.05), (7, 0.3)])
def test_list_consumers_on_a_borrowed_wide_list_vs_oracle(cells, sigma):
    """the reference's policy hands an analysis any held cutoff list that reaches far enough: here a list built for 5.0 A (42+ columns)
    under analyses that want 3.1-3.3 A — common neighbour parameter (narrowed to 16 columns first; the 2-cell box is narrower than
    twice the list's reach, where an atom could be listed twice in a row, and must take the general form; the hot 7-cell one has rows of more than 16 within rc), cluster analysis, neighbour
    average, atomic temperature, list RDF"""
    from mdapy_amd import _rdf
    pos, box = _fcc(cells, sigma, 5)
    x, y, z = _xyz(pos)
    N = len(x)
    rng = np.random.default_rng(3)
    v, d, nn = O.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, 5.0, 4)
    assert v.shape[1] > 16
    for rc in (3.1, 3.3):
        c0, c1 = np.zeros(N), np.full(N, -1.0)
        O.compute_cnp(x, y, z, box, ORG0, PBC, v, d, nn, c0, rc, 4)
        _cnp.compute_cnp(x, y, z, box, ORG0, PBC, v, d, nn, c1, rc, 1)
        assert np.allclose(c1, c0, rtol=1e-9, atol=1e-12)
        k0, k1 = np.full(N, -1, np.int32), np.full(N, -7, np.int32)
        assert _cluster.get_cluster(v, d, nn, rc - 0.5, k1) == O.get_cluster(v, d, nn, rc - 0.5, k0) and np.array_equal(k1, k0)
        val = rng.random(N)
        a0, a1 = np.zeros(N), np.full(N, -1.0)
        O.average_by_neighbor(rc, v, d, nn, val, a0, True)
        _neighbor.average_by_neighbor(rc, v, d, nn, val, a1, True)
        assert np.array_equal(a1, a0)
    vel = rng.normal(0, 300.0, (N, 3)); mass = rng.choice([26.98, 63.546], N)
    t0, t1 = np.zeros(N), np.full(N, -1.0)
    O.compute_temp(v, d, vel[:, 0].copy(), vel[:, 1].copy(), vel[:, 2].copy(), mass, t0, 4.2, 4)
    _atomtemp.compute_temp(v, d, vel[:, 0].copy(), vel[:, 1].copy(), vel[:, 2].copy(), mass, t1, 4.2, 1)
    assert np.allclose(t1, t0, rtol=1e-12, atol=0)
    ty = rng.integers(0, 2, N).astype(np.int32)
    g0, g1 = np.zeros((2, 2, 50)), np.zeros((2, 2, 50))
    O._rdf(v, d, nn, ty, g0, 4.5, 50)
    _rdf._rdf(v, d, nn, ty, g1, 4.5, 50)
    assert np.array_equal(g1, g0) and g0.sum() > 0


AJA_PATHS, CNP_PATHS = fixtures_with("aja"), fixtures_with("cnp")


@pytest.mark.parametrize("path", AJA_PATHS, ids=ids_of(AJA_PATHS))
def test_golden_aja(path):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_ackland_jones_analysis()
    assert np.array_equal(s.data["aja"].to_numpy(), d["aja"])


@pytest.mark.parametrize("path", CNP_PATHS, ids=ids_of(CNP_PATHS))
def test_golden_cnp(path):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_common_neighbor_parameter(float(d["cnp_cutoff"]))
    assert np.allclose(s.data["cnp"].to_numpy(), d["cnp"], atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("name", ["rec_box_big", "tri_box_small"])
@pytest.mark.parametrize("mode", ["default", "use_local_density", "compute_average"])
def test_golden_structure_entropy(name, mode):
    expected = misc("structure_entropy")[f"{name}__{mode}"]
    s = mp.System(input_path(f"{name}.xyz"))
    if mode == "compute_average":
        s.cal_structure_entropy(5.0, 0.2, False, average_rc=4.0)
        got = s.data["entropy_ave"].to_numpy()
    else:
        s.cal_structure_entropy(5.0, 0.2, mode == "use_local_density")
        got = s.data["entropy"].to_numpy()
    assert np.allclose(got, expected, atol=1e-6)


from mdapy_amd import _atomtemp, _cluster


@pytest.mark.parametrize("case", ["fcc_rattled", "triclinic_random", "random_gas", "dense_blob"])
def test_atomic_temperature_and_cluster_vs_oracle(case):
    name, pos, box, origin, bd = next(c for c in _cases() if c[0] == case)
    x, y, z = _xyz(pos)
    N = len(pos)
    rng = np.random.default_rng(12)
    rc = 3.0
    v, d, nn = O.build_neighbor_without_max_neigh(x, y, z, box, origin, bd, rc, 4)
    vel = rng.normal(0, 300.0, (N, 3))
    mass = rng.choice([26.98, 58.69, 63.546], N)
    t0, t1 = np.zeros(N), np.full(N, -1.0)
    O.compute_temp(v, d, vel[:, 0].copy(), vel[:, 1].copy(), vel[:, 2].copy(), mass, t0, 2.8, 4)
    _atomtemp.compute_temp(v, d, vel[:, 0].copy(), vel[:, 1].copy(), vel[:, 2].copy(), mass, t1, 2.8, 1)
    assert np.allclose(t1, t0, rtol=1e-12, atol=0)
    for cut in (1.2, 2.0, 2.8):  # from many small clusters to a few big ones
        c0, c1 = np.full(N, -1, np.int32), np.full(N, -7, np.int32)
        n0 = O.get_cluster(v, d, nn, cut, c0)
        n1 = _cluster.get_cluster(v, d, nn, cut, c1)
        assert n1 == n0 and np.array_equal(c1, c0)
    ty = rng.integers(1, 3, N).astype(np.int32)
    t1a, t2a, r = np.array([1, 2, 1, 2], np.int32), np.array([1, 2, 2, 1], np.int32), np.array([2.8, 2.8, 1.5, 1.5])
    va, vb = v.copy(), v.copy()
    O.filter_by_type(va, d, nn, ty, t1a, t2a, r)
    _cluster.filter_by_type(vb, d, nn, ty, t1a, t2a, r)
    assert np.array_equal(vb, va) and (va != v).any()
    c0, c1 = np.full(N, -1, np.int32), np.full(N, -7, np.int32)
    assert _cluster.get_cluster_by_bond(vb, nn, c1) == O.get_cluster_by_bond(va, nn, c0)
    assert np.array_equal(c1, c0)


@pytest.mark.parametrize("case", ["random_gas", "fcc_hot_shifted_origin", "dense_blob"])
def test_cluster_directed_lists_follow_the_reference_sweep(case):
    """k-nearest lists (what System.verlet_list holds after cal_centro_symmetry_parameter) and one-sided type filters are
    not symmetric: the reference's sweep follows bonds in one direction only (src/cluster.cpp:14-53)."""
    name, pos, box, origin, bd = next(c for c in _cases() if c[0] == case)
    x, y, z = _xyz(pos)
    N = len(pos)
    k = 6
    idx, dk = np.zeros((N, k), np.int32), np.zeros((N, k))
    O.knn(x, y, z, box, origin, bd, k, idx, dk, 4)
    nn = np.full(N, k, np.int32)
    for cut in (float(np.quantile(dk, 0.3)), float(np.quantile(dk, 0.7)), float(dk.max())):
        c0, c1 = np.full(N, -1, np.int32), np.full(N, -7, np.int32)
        n0 = O.get_cluster(idx, dk, nn, cut, c0)
        n1 = _cluster.get_cluster(idx, dk, nn, cut, c1)
        assert n1 == n0 and np.array_equal(c1, c0)
    ty = np.random.default_rng(5).integers(1, 3, N).astype(np.int32)
    va, vb = idx.copy(), idx.copy()
    one_sided = (np.array([1], np.int32), np.array([2], np.int32), np.array([float(np.quantile(dk, 0.5))]))
    O.filter_by_type(va, dk, nn, ty, *one_sided)
    _cluster.filter_by_type(vb, dk, nn, ty, *one_sided)
    assert np.array_equal(vb, va)
    c0, c1 = np.full(N, -1, np.int32), np.full(N, -7, np.int32)
    assert _cluster.get_cluster_by_bond(vb, nn, c1) == O.get_cluster_by_bond(va, nn, c0)
    assert np.array_equal(c1, c0)


def test_cluster_system_flow_large():
    """1 M atoms: two half-spaces separated by a gap wider than rc -> exactly 2 clusters (periodic in x,y; open in z)"""
    pos, box = _fcc(64, 0.02, 3)
    L = box[2, 2] if np.ndim(box) == 2 else box[2]
    pos = pos[(pos[:, 2] < 0.4 * L) | (pos[:, 2] > 0.6 * L)]
    s = mp.System(pos=pos, box=mp.Box(box, boundary=[1, 1, 0]))
    s.cal_cluster_analysis(3.0)
    cid = s.data["cluster_id"].to_numpy()
    assert s.cluster_number == 2 and set(np.unique(cid)) == {1, 2}
    assert np.array_equal(cid == cid[0], (pos[:, 2] < 0.5 * L) == (pos[0, 2] < 0.5 * L))


# ------------------------------------------------------------------ FCC planar faults (ordered sweep reproduced in parallel rounds)
from mdapy_amd import _fccpft


def test_fcc_planar_faults_fixture_and_oracle():
    expected = misc("fcc_planar_faults")["pft"]
    s = mp.System(input_path("ISF.dump"))
    s.cal_polyhedral_template_matching("all", identify_fcc_planar_faults=True, identify_esf=False)
    got = s.data["pft"].to_numpy()
    assert np.array_equal(got, expected)
    # same PTM output, ESF pass on, and a scrambled atom order (the sweep is order dependent): HIP == serial oracle
    st = s.data["ptm"].to_numpy().astype(np.int32)
    p12 = np.ascontiguousarray(np.asarray(s.ptm_indices)[:, 1:13]).astype(np.int32)
    rng = np.random.default_rng(8)
    for scramble in (False, True):
        if scramble:
            perm = rng.permutation(len(st))
            inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
            st_, p_ = st[perm], np.where(p12[perm] >= 0, inv[np.clip(p12[perm], 0, None)], -1).astype(np.int32)
        else:
            st_, p_ = st, p12
        h = np.where(st_ == 2)[0].astype(np.int32)
        f0, f1 = np.zeros_like(st_), np.zeros_like(st_)
        hn0, hn1 = np.zeros((len(h), 12), np.int32), np.zeros((len(h), 12), np.int32)
        O.identify_sftb_fcc(h, hn0, p_, st_, f0, True)
        _fccpft.identify_sftb_fcc(h, hn1, p_, st_, f1, True)
        assert np.array_equal(hn1, hn0) and np.array_equal(f1, f0)
        assert len(np.unique(f0)) >= 4


@pytest.mark.parametrize("case", ["fcc_hot_shifted_origin", "fcc_unwrapped", "triclinic_random", "random_gas", "thin_box_3cells", "cluster_open"])
def test_filter_overlap_atom_vs_oracle(case):
    name, pos, box, origin, bd = next(c for c in _cases() if c[0] == case)
    x, y, z = _xyz(pos)
    for rc in (1.5, 2.7):
        k0 = O.filter_overlap_atom(x, y, z, box, origin, bd, rc, 4)
        k1 = _neighbor.filter_overlap_atom(x, y, z, box, origin, bd, rc, 1)
        assert np.array_equal(k1, k0)
    assert 0 < k0.sum() <= len(k0)


# ------------------------------------------------------------------ BASELINE.json configurations at FULL size (properties that do
# not need a second implementation: closed forms, invariants, determinism) — full System flow, host arrays in/out
def _fcc_system(cells, sigma=0.0, seed=0, binary=False):
    pos, box = lattice_positions("fcc", 3.615, cells, cells, cells)
    rng = np.random.default_rng(seed)
    if sigma > 0:
        pos = pos + rng.normal(0.0, sigma, pos.shape)
    s = mp.System(pos=pos, box=box)
    if binary:
        s.update_data(s.data.with_columns(type=rng.integers(1, 3, len(pos)).astype(np.int32)))
    return s


@pytest.mark.parametrize("sigma,seed", [(0.05, 0), (0.05, 1), (0.20, 0), (0.20, 1)])
def test_config1_1M_neighbor_cna_csp_bit_exact_vs_cpu(sigma, seed):
    """configs[1] at its specification: 63^3 cells = 1 000 188 atoms of fcc Cu, pos += Normal(0, sigma) with
    default_rng(seed); build_neighbor(rc = a = 3.615, max_neigh = 24) — the knife edge: the second shell sits ON the cutoff,
    counts run from 12 to 18 and more — rows, counts and distances bit for bit against the CPU oracle; fixed-cutoff CNA at
    rc = 0.854 a = 3.08721 from a list of that cutoff, labels bit for bit; CSP(12) <= 1e-6."""
    s = _fcc_system(63, sigma, seed)
    x, y, z = (np.ascontiguousarray(s.data[c].to_numpy()) for c in "xyz")
    assert len(x) == 1000188
    rc, M = 3.615, 24
    s.build_neighbor(rc, max_neigh=M)
    v, d, n = (np.asarray(a) for a in (s.verlet_list, s.distance_list, s.neighbor_number))
    V = np.full_like(v, -1); D = np.full_like(d, rc + 1.0); N_ = np.zeros_like(n)
    O.build_neighbor(x, y, z, s.box.box, s.box.origin, s.box.boundary, rc, V, D, N_, 64)
    assert np.array_equal(n, N_) and np.array_equal(v, V) and np.array_equal(d, D)
    assert n.min() >= 10 and n.max() <= M and len(np.unique(n)) >= 4  # the knife edge spreads the counts; no row overflows
    rcna = 0.854 * 3.615
    s2 = _fcc_system(63, sigma, seed)
    s2.cal_common_neighbor_analysis(rc=rcna)
    Vc, Dc, Nc = O.build_neighbor_without_max_neigh(x, y, z, s.box.box, s.box.origin, s.box.boundary, rcna, 64)
    P = np.zeros(len(x), np.int32)
    O.fcna(x, y, z, s.box.box, s.box.origin, s.box.boundary, Vc, Nc, P, rcna, 64)
    assert np.array_equal(s2.data["cna"].to_numpy(), P)
    assert (len(np.unique(P)) > 1) == (sigma > 0.1)  # sigma 0.05 leaves every atom fcc, 0.20 does not
    # CSP: the 12-neighbour search of the GPU (checked against the brute-force oracle at small sizes above; the oracle's
    # search is O(N^2)) feeds both implementations of the parameter itself
    from mdapy_amd import _fast_knn
    I = np.zeros((len(x), 12), np.int32); Dk = np.zeros((len(x), 12))
    _fast_knn.knn(x, y, z, s.box.box, s.box.origin, s.box.boundary, 12, I, Dk, 1)
    s2.cal_centro_symmetry_parameter(12)
    C = np.zeros(len(x))
    O.get_csp(x, y, z, s.box.box, s.box.origin, s.box.boundary, I, 12, C, 64)
    assert np.allclose(s2.data["csp"].to_numpy(), C, rtol=1e-6, atol=1e-9)
    assert np.all(np.diff(Dk, axis=1) >= 0) and Dk[:, 0].min() > 1.0


def test_config2_10M_ptm_steinhardt_closed_forms():
    """configs[2]: 10 M-atom FCC Cu, PTM + Steinhardt q4/q6.  Perfect lattice: every atom FCC, rmsd ~ 0, d = a/sqrt(2),
    identity orientation; q4 = 0.190941, q6 = 0.574524 (the reference's closed forms, tests/test_steinhardt_bond_orientation.py)."""
    s = _fcc_system(136)  # 10 061 824 atoms
    s.cal_polyhedral_template_matching(return_rmsd=True, return_atomic_distance=True, return_orientation=True)
    assert np.all(s.data["ptm"].to_numpy() == 1)
    assert s.data["rmsd"].to_numpy().max() < 1e-6
    assert np.allclose(s.data["interatomic_distance"].to_numpy(), 3.615 / np.sqrt(2), rtol=1e-9)
    assert np.allclose(np.abs(s.data["qw"].to_numpy()), 1.0, atol=1e-6)
    ind = np.asarray(s.ptm_indices)
    assert np.array_equal(ind[:, 0], np.arange(s.N)) and np.all(ind[:, 13:] == -1) and np.all(ind[:, 1:13] >= 0)
    s.cal_steinhardt_bond_orientation([4, 6], rc=0.854 * 3.615)
    assert np.allclose(s.data["ql4"].to_numpy(), 0.190941, atol=1e-6)
    assert np.allclose(s.data["ql6"].to_numpy(), 0.574524, atol=1e-6)


def _reduced_rows(block, rows, n_total):
    """sub-system for an oracle run on a block of a big system: the block's atoms plus every atom their rows name,
    renumbered in ascending order; rows of the block's atoms translated, rows of the others a copy of the first block
    atom's row (never looked at).  Returns (ids of the sub-system, translated rows, positions of the block in it)."""
    ids = np.unique(np.concatenate([block, rows[block].ravel()]))
    ids = ids[ids >= 0]
    lut = np.full(n_total + 1, -1, np.int64)  # slot n_total serves the -1 pads
    lut[ids] = np.arange(len(ids))
    sub = np.tile(lut[rows[block[0]]], (len(ids), 1)).astype(np.int32)
    where = lut[block]
    sub[where] = lut[rows[block]]
    return ids, sub, where


def test_config2_at_spec_ptm_and_steinhardt_block_vs_reference():
    """configs[2] as SURVEY 8d C3 states it: 136^3 fcc cells = 10 061 824 atoms displaced by N(0, 0.05) seed 0, kNN-18 ->
    PTM("fcc-hcp-bcc", rmsd 0.1) and Steinhardt q4/q6 (nnn = 12 and rc = 0.85 a).  A contiguous block of 120 000 atoms is
    recomputed by the reference's own PTM library (oracle/_ref) / the CPU oracle from the rows of the 10 M-atom lists."""
    if not O.have_ref():
        pytest.skip("oracle/_ref/libptm_ref.so missing")
    a = 3.615
    s = _fcc_system(136, 0.05, 0)
    n = s.N
    assert n == 10_061_824
    x, y, z = (np.ascontiguousarray(s.data[c].to_numpy()) for c in "xyz")
    s.build_nearest_neighbor(18)
    knn = np.asarray(s.verlet_list)
    kd = np.asarray(s.distance_list)
    assert knn.shape == (n, 18) and np.all(np.diff(kd, axis=1) >= 0)
    s.cal_polyhedral_template_matching("fcc-hcp-bcc", rmsd_threshold=0.1, return_rmsd=True, return_atomic_distance=True, return_orientation=True,
                                       return_ordering=True)
    block = np.arange(4_000_000, 4_120_000)
    ids, sub, where = _reduced_rows(block, knn, n)
    out_r, ind_r = np.zeros((len(ids), 8)), np.zeros((len(ids), 18), np.int32)
    O.get_ptm("fcc-hcp-bcc", x[ids], y[ids], z[ids], s.box.box, s.box.origin, s.box.boundary, sub, None, 0.1, out_r, ind_r)
    d = s.data
    out_g = np.column_stack([d["ptm"].to_numpy()[block], d["ordering"].to_numpy()[block], d["rmsd"].to_numpy()[block],
                             d["interatomic_distance"].to_numpy()[block], d["qw"].to_numpy()[block], d["qx"].to_numpy()[block],
                             d["qy"].to_numpy()[block], d["qz"].to_numpy()[block]]).astype(np.float64)
    ind_g = np.asarray(s.ptm_indices)[block]
    back = np.append(ids, -1)  # sub-system index -> atom id (-1 stays -1)
    compare_ptm(out_g, ind_g, out_r[where], back[ind_r[where]].astype(np.int32))
    labels = np.bincount(d["ptm"].to_numpy(), minlength=9)
    assert labels[1] > 0.999 * n and labels.sum() == n
    # Steinhardt, both neighbourhood conventions, against the CPU oracle on the same block
    ll = np.array([4, 6], np.int32)
    s.cal_steinhardt_bond_orientation([4, 6], nnn=12)
    q_nnn = np.column_stack([s.data["ql4"].to_numpy()[block], s.data["ql6"].to_numpy()[block]])
    ids12, sub12, where12 = _reduced_rows(block, knn[:, :12], n)
    dsub = np.tile(kd[block[0], :12], (len(ids12), 1))
    dsub[where12] = kd[block, :12]
    qr = np.zeros((len(ids12), 2, 13)); qi = np.zeros_like(qr); qn = np.zeros((len(ids12), 2))
    O.get_sq(x[ids12], y[ids12], z[ids12], s.box.box, s.box.origin, s.box.boundary, sub12, dsub, np.full(len(ids12), 12, np.int32),
             np.zeros((2, 2)), ll, 12, 6, False, False, False, False, 1e9, False, qr, qi, qn, 64)
    assert np.allclose(q_nnn, qn[where12], rtol=1e-6, atol=1e-12)
    rc = 0.85 * a
    s.build_neighbor(rc, max_neigh=16)
    v, dd, nn = (np.asarray(t) for t in (s.verlet_list, s.distance_list, s.neighbor_number))
    s.cal_steinhardt_bond_orientation([4, 6], rc=rc)
    q_rc = np.column_stack([s.data["ql4"].to_numpy()[block], s.data["ql6"].to_numpy()[block]])
    idsr, subr, wherer = _reduced_rows(block, v, n)
    dr_ = np.tile(dd[block[0]], (len(idsr), 1)); dr_[wherer] = dd[block]
    nr_ = np.full(len(idsr), nn[block[0]], np.int32); nr_[wherer] = nn[block]
    qr = np.zeros((len(idsr), 2, 13)); qi = np.zeros_like(qr); qn = np.zeros((len(idsr), 2))
    O.get_sq(x[idsr], y[idsr], z[idsr], s.box.box, s.box.origin, s.box.boundary, subr, dr_, nr_, np.zeros((2, 2)), ll, 0, 6, False, False,
             False, False, rc, False, qr, qi, qn, 64)
    assert np.allclose(q_rc, qn[wherer], rtol=1e-6, atol=1e-12)
    assert 0.15 < q_rc[:, 0].mean() < 0.20 and 0.45 < q_rc[:, 1].mean() < 0.58  # rattled fcc: a little below 0.1909 / 0.5745


def test_config4_at_spec_glass_rdf_wcp_full_size_vs_oracle():
    """configs[4] as SURVEY 8d C5 states it: 135^3 x 4 = 9 841 500 fcc sites (a = 4.0) displaced by N(0, 0.35) seed 7,
    Cu64Zr36 by a shuffled repeat (seed 42); streaming partial g_ab(r), rc = 8, 200 bins — the 2 x 2 x 200 pair counts of the
    WHOLE system bit for bit against the CPU oracle, and conserved against an independent kernel (the counting pass of the
    neighbour search); Warren-Cowley at rc = 3.6 against the oracle on the 10 M-atom list."""
    pos, box = lattice_positions("fcc", 4.0, 135, 135, 135)
    pos = pos + np.random.default_rng(7).normal(0.0, 0.35, pos.shape)
    n = len(pos)
    assert n == 9_841_500
    ty = np.repeat([0, 1], [int(round(0.64 * n)), n - int(round(0.64 * n))]).astype(np.int32)
    np.random.default_rng(42).shuffle(ty)
    x, y, z = _xyz(pos)
    g_gpu, g_cpu = np.zeros((2, 2, 200)), np.zeros((2, 2, 200))
    _rdf._rdf_streaming(x, y, z, ty, box, ORG0, PBC, g_gpu, 8.0, 200, 1)
    O._rdf_streaming(x, y, z, ty, box, ORG0, PBC, g_cpu, 8.0, 200, 64)
    assert np.array_equal(g_gpu, g_cpu)
    s = mp.System(pos=pos, box=box)
    s.update_data(s.data.with_columns(type=(ty + 1).astype(np.int32)))
    rdf = s.cal_radial_distribution_function(8.0, nbin=200, streaming=True)
    conc = np.array([0.64, 0.36])
    total = sum(conc[a_] * conc[b_] * rdf.g_partial[(a_ + 1, b_ + 1)] * (1.0 if a_ == b_ else 2.0) for a_ in range(2) for b_ in range(a_, 2))
    assert np.allclose(total, rdf.g_total, rtol=1e-9, atol=1e-12)
    assert abs(rdf.g_total[rdf.r > 6.0].mean() - 1.0) < 0.02 and rdf.g_total[rdf.r < 1.0].max() < 0.05
    # Warren-Cowley at 3.6 on the full list, and the list's pair count against the histogram's first bins
    v, d, nn = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, 3.6, 1)
    w_cpu, w_gpu = np.zeros((2, 2)), np.zeros((2, 2))
    O.get_wcp(np.asarray(v), np.asarray(nn), ty, 2, w_cpu, 64)
    _wcp.get_wcp(v, nn, ty, 2, w_gpu, 1)
    assert np.array_equal(w_gpu, w_cpu) and np.abs(w_cpu).max() < 5e-3
    assert int(np.asarray(nn).sum()) == int(g_cpu[:, :, :90].sum())  # 90 bins of 0.04 = 3.6: every ordered pair once in each


def test_config4_10M_binary_rdf_wcp_invariants():
    """configs[4]: 10 M-atom binary system, partial g_ab(r) + Warren-Cowley.  Invariants: the species-weighted partials
    add up to the total g(r); g averages to 1 beyond the first shell and vanishes inside the core; alpha_ab of a random
    occupation vanishes (|alpha| < 2e-3 at 10 M atoms) and sum_b c_b (1 - alpha_ab) = 1 exactly."""
    s = _fcc_system(136, 0.25, 5, binary=True)
    rdf = s.cal_radial_distribution_function(6.0, nbin=120)
    ty = s.data["type"].to_numpy()
    c = np.bincount(ty)[1:] / len(ty)
    gab = rdf.g_partial
    keys = sorted(gab.keys())
    total = sum(c[a_ - 1] * c[b_ - 1] * gab[(a_, b_)] * (1.0 if a_ == b_ else 2.0) for a_, b_ in keys)
    assert np.allclose(total, rdf.g_total, rtol=1e-9, atol=1e-12)
    assert abs(rdf.g_total[rdf.r > 3.0].mean() - 1.0) < 0.1 and rdf.g_total[rdf.r < 1.2].max() < 0.01
    w = s.cal_warren_cowley_parameter(0.854 * 3.615)
    alpha = np.asarray(w.WCP)
    assert np.abs(alpha).max() < 2e-3
    assert np.allclose(((1.0 - alpha) * c[None, :]).sum(axis=1), 1.0, atol=1e-12)


# ------------------------------------------------------------------ Voronoi volume / faces / cavity radius: HIP (half-space clipping,
# one wave per cell) vs oracle/_ref (the reference's voro++) and the OVITO-derived fixtures
from mdapy_amd import _voronoi

VOR_PATHS = fixtures_with("voronoi_volume")
needs_voro = pytest.mark.skipif(not O.have_voro_ref(), reason="oracle/_ref/libvoro_ref.so missing")


@pytest.mark.parametrize("path", VOR_PATHS, ids=ids_of(VOR_PATHS))
def test_golden_voronoi(path):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_voronoi_volume()
    assert np.allclose(d["voronoi_volume"], s.data["volume"].to_numpy(), atol=1e-6)
    assert np.allclose(d["voronoi_cavity_radius"], s.data["cavity_radius"].to_numpy() * 0.5, atol=1e-6)
    assert np.array_equal(d["voronoi_coord"], s.data["neighbor_number"].to_numpy())


@needs_voro
@pytest.mark.parametrize("case", ["fcc_rattled", "fcc_hot_shifted_origin", "random_gas", "slab_open_z", "thin_box_3cells"])
def test_voronoi_vs_reference_library(case):
    name, pos, box, origin, bd = next(c for c in _cases() if c[0] == case)
    if case == "slab_open_z":  # keep every atom inside the container along the open axis
        pos = pos.copy(); pos[:, 2] = np.clip(pos[:, 2], 1e-3, (box[2, 2] if np.ndim(box) == 2 else box[2]) - 1e-3)
    x, y, z = _xyz(pos)
    N = len(pos)
    v0, n0, r0 = np.zeros(N), np.zeros(N, np.int32), np.zeros(N)
    O.get_voronoi_volume_number_radius(x, y, z, box, origin, bd, v0, n0, r0)
    v1, n1, r1 = np.zeros(N), np.zeros(N, np.int32), np.zeros(N)
    _voronoi.get_voronoi_volume_number_radius(x, y, z, box, origin, bd, v1, n1, r1)
    assert np.allclose(v1, v0, rtol=1e-9, atol=1e-9) and np.allclose(r1, r0, rtol=1e-9, atol=1e-9)
    if not np.array_equal(n1, n0):
        # voro++ now and then keeps a sliver its vertex tolerance leaves behind (a "face" of 1e-14 A^2 where four cells meet
        # almost in a line; found by the randomised sweep, seed 54901): the counts must agree once faces below 1e-10 A^2 are
        # left out on both sides
        va, da, fa, ca = O.get_voronoi_neighbor(x, y, z, box, origin, bd, -1.0, -1.0)
        vb, db, fb, cb = _voronoi.get_voronoi_neighbor(x, y, z, box, origin, bd, -1.0, -1.0)
        real = lambda f, c: np.array([(f[r, :c[r]] >= 1e-10).sum() for r in range(len(c))])
        assert np.array_equal(ca, n0) and np.array_equal(cb, n1) and np.array_equal(real(fa, ca), real(fb, cb))
    if all(bd):
        vol = abs(np.linalg.det(np.asarray(box, float))) if np.ndim(box) == 2 else float(np.prod(box))
        assert abs(v1.sum() - vol) < 1e-6 * vol  # the cells tile the periodic box


def test_voronoi_neighbor_rows_in_one_construction_equal_count_then_fill():
    """get_voronoi_neighbor builds the cells once (mdh_voronoi_neighbor_rows: rows 32 columns wide on the device, handed over at the
    width the face counts ask for); a cell with more faces than the first attempt's columns sends it round again — both against
    mdh_voronoi_neighbor_count + mdh_voronoi_neighbor, the two-construction form, bit for bit."""
    import ctypes
    from mdapy_amd import _lib
    rng = np.random.default_rng(93)
    pos, box = rng.random((3000, 3)) * 28.0, np.eye(3) * 28.0  # a gas: 8 ... 30 faces per cell
    x, y, z = _xyz(pos)
    keep, (pb, po, pp) = _lib.host_box(box, ORG0, PBC)
    n = len(x)
    L = _lib.lib()
    for a_thr, r_thr in ((-1.0, -1.0), (0.5, -1.0), (-1.0, 0.01)):
        nn0 = np.zeros(n, np.int32); w = ctypes.c_int(0)
        _lib.check(L.mdh_voronoi_neighbor_count(x.ctypes.data, y.ctypes.data, z.ctypes.data, n, pb, po, pp, nn0.ctypes.data, ctypes.byref(w), _lib.HOST, None))
        W = int(w.value)
        v0 = np.full((n, W), -1, np.int32); d0 = np.full((n, W), 10000.0); f0 = np.zeros((n, W))
        _lib.check(L.mdh_voronoi_neighbor(x.ctypes.data, y.ctypes.data, z.ctypes.data, n, pb, po, pp, a_thr, r_thr, v0.ctypes.data, d0.ctypes.data, f0.ctypes.data, W, _lib.HOST, None))
        assert W > 8
        for guess in (32, 8):
            old = _voronoi._ROW_GUESS
            _voronoi._ROW_GUESS = guess
            try:
                v1, d1, f1, nn1 = _voronoi.get_voronoi_neighbor(x, y, z, box, ORG0, PBC, a_thr, r_thr)
            finally:
                _voronoi._ROW_GUESS = old
            assert v1.shape == (n, W) and np.array_equal(nn1, nn0)
            assert np.array_equal(v1, v0) and np.array_equal(d1, d0) and np.array_equal(f1, f0)


@needs_voro
@pytest.mark.parametrize("kind", ["slab", "cluster", "void"])
def test_voronoi_open_cells_take_the_listed_atom_passes(kind):
    """After the first pass only the atoms whose cell is still open get rows at a wider radius (voronoi.hip k_rows_of_listed):
    a slab, a free cluster and a crystal with a spherical void — volumes, cavity radii and face counts against voro++, the
    neighbour rows too, and the listed-atom passes did run."""
    from mdapy_amd import _lib
    rng = np.random.default_rng(91)
    pos, box = _fcc(10, 0.05, 3)
    bd = {"slab": np.array([1, 1, 0], np.int32), "cluster": np.array([0, 0, 0], np.int32), "void": PBC}[kind]
    bl = np.diag(box) if np.ndim(box) == 2 else np.asarray(box, float)
    if kind == "void":
        pos = pos[np.linalg.norm(pos - 0.5 * bl, axis=1) > 6.0]
    pos = np.clip(pos, 1e-3, bl - 1e-3)  # inside the container along open axes
    x, y, z = _xyz(pos)
    N = len(pos)
    v0, n0, r0 = np.zeros(N), np.zeros(N, np.int32), np.zeros(N)
    O.get_voronoi_volume_number_radius(x, y, z, box, ORG0, bd, v0, n0, r0)
    v1, n1, r1 = np.zeros(N), np.zeros(N, np.int32), np.zeros(N)
    _voronoi.get_voronoi_volume_number_radius(x, y, z, box, ORG0, bd, v1, n1, r1)
    cnt = np.zeros(4, np.int64)
    _lib.lib().mdh_debug_counters(cnt.ctypes.data)
    assert cnt[3] >= 1, cnt
    assert np.allclose(v1, v0, rtol=1e-9, atol=1e-9) and np.allclose(r1, r0, rtol=1e-9, atol=1e-9)
    va, da, fa, ca = O.get_voronoi_neighbor(x, y, z, box, ORG0, bd, -1.0, -1.0)
    vb, db, fb, cb = _voronoi.get_voronoi_neighbor(x, y, z, box, ORG0, bd, -1.0, -1.0)
    real = lambda f, c: np.array([(f[r, :c[r]] >= 1e-10).sum() for r in range(len(c))])
    assert np.array_equal(ca, n0) and np.array_equal(cb, n1) and np.array_equal(real(fa, ca), real(fb, cb))
    for r in rng.choice(N, 200, replace=False):  # the same neighbour sets with the same face areas (rows are in different orders)
        ka = {int(j): float(f) for j, f in zip(va[r, :ca[r]], fa[r, :ca[r]]) if f >= 1e-10}
        kb = {int(j): float(f) for j, f in zip(vb[r, :cb[r]], fb[r, :cb[r]]) if f >= 1e-10}
        assert ka.keys() == kb.keys() and all(abs(ka[j] - kb[j]) <= 1e-7 for j in ka)
    if kind == "void":
        assert abs(v1.sum() - float(np.prod(bl))) < 1e-6 * float(np.prod(bl))


@needs_voro
@pytest.mark.parametrize("case", ["fcc_rattled", "fcc_hot_shifted_origin", "random_gas", "slab_open_z"])
def test_voronoi_neighbors_vs_reference_library(case):
    """neighbour SETS with their face areas and distances (the reference's row order is voro++'s internal face order)"""
    name, pos, box, origin, bd = next(c for c in _cases() if c[0] == case)
    if case == "slab_open_z":
        pos = pos.copy(); pos[:, 2] = np.clip(pos[:, 2], 1e-3, (box[2, 2] if np.ndim(box) == 2 else box[2]) - 1e-3)
    x, y, z = _xyz(pos)
    # voro++ now and then keeps a sliver its vertex tolerance leaves behind (test_voronoi_vs_reference_library): such "faces"
    # (< 1e-10 A^2) are counted per side here and left out of every comparison below; the reference's count is the cell's
    # number of faces whatever the thresholds filter out of the row
    _, _, fa, ca = O.get_voronoi_neighbor(x, y, z, box, origin, bd, -1.0, -1.0)
    _, _, fb, cb = _voronoi.get_voronoi_neighbor(x, y, z, box, origin, bd, -1.0, -1.0)
    slivers = lambda f, c: np.array([(f[r, :c[r]] < 1e-10).sum() for r in range(len(c))])
    sl0, sl1 = slivers(fa, ca), slivers(fb, cb)
    for a_thr, r_thr in ((-1.0, -1.0), (0.3, -1.0), (-1.0, 0.02)):
        v0, d0, f0, n0 = O.get_voronoi_neighbor(x, y, z, box, origin, bd, a_thr, r_thr)
        v1, d1, f1, n1 = _voronoi.get_voronoi_neighbor(x, y, z, box, origin, bd, a_thr, r_thr)
        assert np.array_equal(n1 - sl1, n0 - sl0)
        v1_rows, d1_rows = v1, d1  # (as returned: the order checks at the end look at these)
        if a_thr < 0 and r_thr < 0 and (sl0.any() or sl1.any()):  # unfiltered rows list the slivers: make them look like pads
            drop = lambda v, d, f: tuple(np.where(f >= 1e-10, a, fill) for a, fill in ((v, -1), (d, 10000.0), (f, 0.0)))
            (v0, d0, f0), (v1, d1, f1) = drop(v0, d0, f0), drop(v1, d1, f1)
        # canonical form of a row: entries sorted by (neighbour id, face area)
        def canon(v, d, f):
            # (neighbour id, face area): one neighbour can appear twice (two images, same minimum-image distance)
            o = np.stack([np.lexsort((np.round(f[r], 6), np.where(v[r] >= 0, v[r], np.iinfo(np.int32).max))) for r in range(len(v))])
            return np.take_along_axis(v, o, 1), np.take_along_axis(d, o, 1), np.take_along_axis(f, o, 1)
        w = max(v0.shape[1], v1.shape[1])
        pad = lambda a, fill: np.pad(a, ((0, 0), (0, w - a.shape[1])), constant_values=fill)
        c0 = canon(pad(v0, -1), pad(d0, 10000.0), pad(f0, 0.0))
        c1 = canon(pad(v1, -1), pad(d1, 10000.0), pad(f1, 0.0))
        assert np.array_equal(c1[0], c0[0])
        there = c0[0] >= 0
        assert np.allclose(c1[1][there], c0[1][there], rtol=1e-9, atol=1e-9) and np.allclose(c1[2][there], c0[2][there], rtol=1e-7, atol=1e-9)
        # rows are ordered by the distance of the image that makes the face, pads last; the reported distance is the
        # reference's minimum-image one, so the two orders can only differ where a thin box makes a cell touch a
        # farther image of a neighbour
        v1, d1 = v1_rows, d1_rows
        dd = np.where(v1 >= 0, d1, 1e9)
        L = np.diag(np.asarray(box, float)) if np.ndim(box) == 2 else np.asarray(box, float)
        vs = np.sort(np.where(v1 >= 0, v1, -np.arange(1, v1.shape[1] + 1)[None, :]), axis=1)
        once = (np.diff(vs, axis=1) != 0).all(axis=1)  # rows in which no neighbour appears through two images
        if 4.0 * d1[v1 >= 0].max() < L[np.asarray(bd) != 0].min(initial=np.inf):  # no cell reaches a farther image
            assert np.all(np.diff(dd[once], axis=1) >= 0)
        assert np.all(np.diff((v1 < 0).astype(int), axis=1) >= 0)


def test_voronoi_weighted_steinhardt_system_flow():
    s = mp.build_crystal("Al", "fcc", 4.05, nx=6, ny=6, nz=6)
    s.cal_steinhardt_bond_orientation([4, 6], use_voronoi=True, use_weight=True)
    assert np.allclose(s.data["ql4"].to_numpy(), 0.190941, atol=1e-6) and np.allclose(s.data["ql6"].to_numpy(), 0.574524, atol=1e-6)
    assert np.all(np.asarray(s.voro_neighbor_number) == 12)


# ------------------------------------------------------------------ static structure factor (direct summation kernel + Debye over the
# streaming RDF kernel): reference fixture, and HIP vs the CPU oracle on seeded inputs
from mdapy_amd import _sfc


@pytest.mark.parametrize("mode", ["direct", "debye"])
def test_golden_structure_factor(mode):
    d = misc("structure_factor")
    n, nbins = int(d["N"]), int(d["nbins"])
    s = mp.System(box=d["box"], pos=d["points"])
    s.update_data(s.data.with_columns(type=np.array([1] * (n // 2) + [2] * (n // 2))))
    sf = s.cal_structure_factor(float(d["k_min"]), float(d["k_max"]), nbins, cal_partial=True, mode=mode)
    for key, name in (((1, 1), "11"), ((1, 2), "12"), ((2, 2), "22")):
        assert np.allclose(sf.Sk_partial[key], d[f"{mode}_{name}"], atol=1e-4, equal_nan=True)
    assert np.allclose(sf.Sk, d[f"{mode}_all"], atol=1e-4, equal_nan=True)
    sf2 = s.cal_structure_factor(float(d["k_min"]), float(d["k_max"]), nbins, cal_partial=False, mode=mode)
    assert np.allclose(sf2.Sk, d[f"{mode}_all"], atol=1e-4, equal_nan=True)


def test_sfc_direct_vs_oracle():
    rng = np.random.default_rng(21)
    tri = np.array([[14.0, 0.0, 0.0], [2.0, 13.0, 0.0], [-1.5, 2.5, 15.0]])
    for box in (np.diag([13.0, 15.0, 17.0]), tri):
        pos = rng.random((1500, 3)) @ box
        x, y, z = _xyz(pos)
        ty = rng.integers(0, 3, len(pos)).astype(np.int32)
        s0, s1 = np.zeros(40), np.zeros(40)
        O.compute_sfc_direct(x, y, z, box, ORG0, PBC, s0, 40, 6.0, 0.3, num_t=8)
        _sfc.compute_sfc_direct(x, y, z, box, ORG0, PBC, s1, 40, 6.0, 0.3)
        assert np.array_equal(np.isnan(s1), np.isnan(s0)) and np.allclose(s1, s0, rtol=1e-9, atol=1e-9, equal_nan=True)
        q = pos[ty == 1]
        qx, qy, qz = _xyz(q)
        O.compute_sfc_direct(x, y, z, box, ORG0, PBC, s0, 40, 6.0, 0.3, qx, qy, qz, len(pos), 8)
        _sfc.compute_sfc_direct(x, y, z, box, ORG0, PBC, s1, 40, 6.0, 0.3, qx, qy, qz, len(pos))
        assert np.allclose(s1, s0, rtol=1e-9, atol=1e-9, equal_nan=True)
        p0, p1 = np.zeros((3, 3, 40)), np.zeros((3, 3, 40))
        O.compute_sfc_direct_partial(x, y, z, ty, 3, box, ORG0, PBC, p0, 40, 6.0, 0.3, 8)
        _sfc.compute_sfc_direct_partial(x, y, z, ty, 3, box, ORG0, PBC, p1, 40, 6.0, 0.3)
        assert np.allclose(p1, p0, rtol=1e-9, atol=1e-9, equal_nan=True)
    with pytest.raises(ValueError):
        _sfc.compute_sfc_direct(x, y, z, box, ORG0, PBC, s1, 40, 6.0, 0.3, qx, qy, qz, 0)


# ------------------------------------------------------------------ SURVEY 8d synthetic configurations C4 / C5 at parity scale
def _polycrystal(L=132.0, nseed=8, a=3.615, seed=2024):
    """FCC grains: rotated lattices clipped to the Voronoi cells of random seeds under PBC (the construction of
    create_polycrystal.py:684-848 in miniature), before overlap removal"""
    rng = np.random.default_rng(seed)
    seeds = rng.random((nseed, 3)) * L
    ang = np.deg2rad(rng.uniform(-180, 180, (nseed, 3)))
    out = []
    n = int(np.ceil(L * np.sqrt(3) / a)) + 1
    g = np.arange(-n // 2, n // 2 + 1)
    basis = np.array([[0, 0, 0], [0.5, 0.5, 0], [0, 0.5, 0.5], [0.5, 0, 0.5]])
    cells = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    lat = ((cells[:, None, :] + basis[None]) * a).reshape(-1, 3)
    for s, (al, be, ga) in zip(seeds, ang):
        ca, sa, cb, sb, cg, sg = np.cos(al), np.sin(al), np.cos(be), np.sin(be), np.cos(ga), np.sin(ga)
        R = np.array([[ca * cb, ca * sb * sg - sa * cg, ca * sb * cg + sa * sg], [sa * cb, sa * sb * sg + ca * cg, sa * sb * cg - ca * sg],
                      [-sb, cb * sg, cb * cg]])
        p = lat @ R.T
        p = p[np.all(np.abs(p) < L / 2, axis=1)] + s
        d = p[:, None, :] - seeds[None]
        d -= L * np.round(d / L)
        own = np.argmin((d ** 2).sum(-1), axis=1)
        k = np.nonzero(seeds == s)[0][0]
        out.append(p[own == k] % L)
    return np.concatenate(out), np.eye(3) * L


def test_config3_polycrystal_overlap_filter_neighbor_cna():
    """configs[3] at 1/512 scale: Voronoi-grain polycrystal, overlap removal at 2.0 A, neighbor + CNA — filter
    mask, rows and labels bit-exact against the CPU oracle"""
    pos, box = _polycrystal()
    x, y, z = _xyz(pos)
    k0 = O.filter_overlap_atom(x, y, z, box, ORG0, PBC, 2.0, 64)
    k1 = _neighbor.filter_overlap_atom(x, y, z, box, ORG0, PBC, 2.0, 1)
    assert np.array_equal(k1, k0) and 0 < (~k0).sum() < 0.1 * len(k0)
    pos = pos[k0]
    assert len(pos) > 150000
    x, y, z = _xyz(pos)
    rc, M = 0.854 * 3.615, 20
    v0, d0, n0 = np.full((len(x), M), -1, np.int32), np.full((len(x), M), rc + 1.0), np.zeros(len(x), np.int32)
    O.build_neighbor(x, y, z, box, ORG0, PBC, rc, v0, d0, n0, 64)
    v1, d1, n1 = np.full((len(x), M), -1, np.int32), np.full((len(x), M), rc + 1.0), np.zeros(len(x), np.int32)
    _neighbor.build_neighbor(x, y, z, box, ORG0, PBC, rc, v1, d1, n1, 1)
    assert np.array_equal(n1, n0) and np.array_equal(v1, v0) and np.array_equal(d1, d0)
    p0, p1 = np.zeros(len(x), np.int32), np.zeros(len(x), np.int32)
    O.fcna(x, y, z, box, ORG0, PBC, v0, n0, p0, rc, 64)
    _cna.fcna(x, y, z, box, ORG0, PBC, v1, n1, p1, rc, 1)
    assert np.array_equal(p1, p0)
    frac_fcc = (p0 == 1).mean()
    assert 0.6 < frac_fcc < 0.99  # grain interiors are fcc, boundaries are not


def test_config3_full_size_polycrystal_neighbor_cna_vs_oracle():
    """configs[3] at FULL size on one GPU: 512 Voronoi grains in a 1 057 A box through the package's own
    builder (`CreatePolycrystal`: HIP Voronoi container, rotate + half-space filter per grain, overlap removal at 2.0 A) —
    ~9.8e7 atoms — then neighbor + fixed-cutoff CNA.  Counts, rows, distances and labels of ALL atoms bit for bit against the
    CPU oracle (64 threads).  What one GPU cannot show is the 8-GPU decomposition; the decomposed code path is tested in
    test_gpu_distributed.py."""
    import time

    L, G = 1057.0, 512
    rng = np.random.default_rng(2024)
    seeds = rng.random((G, 3)) * L
    theta = rng.uniform(-180, 180, (G, 3))
    unit = mp.build_crystal("Cu", "fcc", 3.615)
    t0 = time.perf_counter()
    s = mp.CreatePolycrystal(unit, box=L, seed_number=G, seed_position=seeds, theta_list=theta, metal_overlap_dis=2.0).compute()
    t_build = time.perf_counter() - t0
    n = s.N
    assert 9.6e7 < n < 1.0e8
    grains = s.data["grain_id"].to_numpy()
    assert grains.min() == 1 and grains.max() == G and np.bincount(grains)[1:].min() > 1000
    rc, M = 0.854 * 3.615, 14
    t0 = time.perf_counter()
    s.build_neighbor(rc, max_neigh=M)
    s.cal_common_neighbor_analysis(rc=rc)
    t_gpu = time.perf_counter() - t0
    x, y, z = (np.ascontiguousarray(s.data[c].to_numpy()) for c in "xyz")
    v0 = np.full((n, M), -1, np.int32); d0 = np.full((n, M), rc + 1.0); n0 = np.zeros(n, np.int32)
    t0 = time.perf_counter()
    O.build_neighbor(x, y, z, s.box.box, s.box.origin, s.box.boundary, rc, v0, d0, n0, 64)
    p0 = np.zeros(n, np.int32)
    O.fcna(x, y, z, s.box.box, s.box.origin, s.box.boundary, v0, n0, p0, rc, 64)
    t_cpu = time.perf_counter() - t0
    assert np.array_equal(np.asarray(s.neighbor_number), n0) and n0.max() <= M
    assert np.array_equal(np.asarray(s.verlet_list), v0)
    assert np.array_equal(np.asarray(s.distance_list), d0)
    assert np.array_equal(s.data["cna"].to_numpy(), p0)
    frac = np.bincount(p0, minlength=5) / n
    assert 0.85 < frac[1] < 0.95 and frac[2] < 0.01  # grain interiors fcc, boundaries not
    # the ONE-CALL path (the kernel bench.py times): a fresh System has no list to lend, so cal_common_neighbor_analysis(rc) makes
    # the labels inside the tile kernel (mdh_build_neighbor_exact_fcna) — against the oracle's labels, and its exact-width lists
    # against the oracle's rows
    del s
    s2 = mp.System(pos=np.stack([x, y, z], axis=1), box=np.diag([L, L, L]))
    s2.cal_common_neighbor_analysis(rc=rc)
    assert "verlet_list" in s2.__dict__ and s2.rc == rc  # the list the labelling pass built stays the system's list
    assert np.array_equal(s2.data["cna"].to_numpy(), p0)
    w = int(np.asarray(s2.verlet_list).shape[1])
    assert w == int(n0.max()) and np.array_equal(np.asarray(s2.neighbor_number), n0)
    assert np.array_equal(np.asarray(s2.verlet_list), v0[:, :w]) and np.array_equal(np.asarray(s2.distance_list), d0[:, :w])
    print(f"config 3 full size: {n} atoms, builder {t_build:.1f} s, neighbor + CNA through System {t_gpu * 1e3:.0f} ms "
          f"({n / t_gpu / 1e6:.0f} M atoms/s, host arrays in, labels out), oracle on 64 threads {t_cpu:.1f} s ({n / t_cpu / 1e6:.1f} M atoms/s)")


def test_config4_glass_streaming_rdf_wcp_vs_oracle():
    """configs[4] at parity scale: 37^3 x 4 = 202 612 fcc sites (a = 4.0) displaced by N(0, 0.35), Cu64Zr36 by
    shuffled repeat; streaming partial g_ab(r) with rc = 8, 200 bins — pair counts bit-exact; Warren-Cowley at rc = 3.6"""
    pos, box = lattice_positions("fcc", 4.0, 37, 37, 37)
    rng = np.random.default_rng(7)
    pos = pos + rng.normal(0, 0.35, pos.shape)
    n = len(pos)
    ty = np.repeat([0, 1], [int(round(0.64 * n)), n - int(round(0.64 * n))]).astype(np.int32)
    np.random.default_rng(42).shuffle(ty)
    x, y, z = _xyz(pos)
    g0, g1 = np.zeros((2, 2, 200)), np.zeros((2, 2, 200))
    O._rdf_streaming(x, y, z, ty, box, ORG0, PBC, g0, 8.0, 200, 64)
    _rdf._rdf_streaming(x, y, z, ty, box, ORG0, PBC, g1, 8.0, 200, 1)
    assert np.array_equal(g1, g0) and g0.sum() > 1e7
    v, d, nn = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, 3.6, 1)
    w0, w1 = np.zeros((2, 2)), np.zeros((2, 2))
    O.get_wcp(np.asarray(v), np.asarray(nn), ty, 2, w0, 8)
    _wcp.get_wcp(v, nn, ty, 2, w1, 1)
    assert np.array_equal(w1, w0) and np.abs(w0).max() < 0.02


# ------------------------------------------------------------------ polycrystal builder: grain filling
def test_transform_and_filter_bit_exact_vs_oracle():
    from mdapy_amd import _polycrystal

    rng = np.random.default_rng(17)
    for n, nf in ((0, 4), (1, 1), (70_000, 14), (300_000, 40)):
        pos = rng.random((n, 3)) * 60.0
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        centre = pos.mean(0) if n else np.zeros(3)
        target = rng.random(3) * 30.0
        nrm = rng.normal(size=(nf, 3)); nrm /= np.linalg.norm(nrm, axis=1)[:, None]
        planes = np.c_[nrm, -(nrm @ target) - rng.uniform(8.0, 25.0, nf)]  # a random polyhedron around the target
        x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
        a = O.transform_and_filter(x, y, z, q, centre, target, planes)
        b = _polycrystal.transform_and_filter(x, y, z, q, centre, target, planes, 1)
        assert b.shape == a.shape and np.array_equal(b, a)  # same atoms, same order, same bits
        if n > 1000:
            assert 0 < len(a) < n
    with pytest.raises(ValueError):
        _polycrystal.transform_and_filter(x, y, z, q, centre, target, np.zeros((1100, 4)), 1)


def test_create_polycrystal_hip_equals_oracle_build(monkeypatch):
    """the same builder, once through the HIP library and once with its three device calls routed to the oracle: identical
    atoms (positions bit for bit), and the geometric contract of a Voronoi polycrystal"""
    import _oracle_backend as ob
    from mdapy_amd import devarray

    unit = mp.build_crystal("Al", "fcc", 4.05)
    kw = dict(box=70.0, seed_number=9, randomseed=11, metal_overlap_dis=2.0)
    a = mp.CreatePolycrystal(unit, **kw).compute()
    with monkeypatch.context() as m:
        ob.install(m)
        m.setattr(devarray, "_gpu", False)
        pb = mp.CreatePolycrystal(unit, **kw)
        b = pb.compute()
    assert a.N == b.N and a.data.columns == b.data.columns == ["element", "x", "y", "z", "grain_id", "type"]
    for c in ("x", "y", "z", "grain_id", "type"):
        assert np.array_equal(a.data[c].to_numpy(), b.data[c].to_numpy())
    pos = np.c_[a.data["x"].to_numpy(), a.data["y"].to_numpy(), a.data["z"].to_numpy()]
    assert pos.min() >= 0.0 and pos.max() < 70.0
    d = pos[:, None, :] - pb.seed_position[None, :, :]
    d -= 70.0 * np.round(d / 70.0)
    assert np.array_equal(np.argmin((d ** 2).sum(-1), axis=1) + 1, a.data["grain_id"].to_numpy())  # Voronoi assignment
    assert abs(pb.volume.sum() - 70.0 ** 3) < 1e-6 * 70.0 ** 3
    assert 0.90 < a.N / (70.0 ** 3 * 4 / 4.05 ** 3) < 1.0  # bulk density minus the grain-boundary overlaps
    a.build_neighbor(2.0 - 1e-9)
    assert int(np.asarray(a.neighbor_number).max()) == 0  # no pair closer than the overlap distance is left
    with pytest.raises(ValueError, match="Triclinic"):
        mp.CreatePolycrystal(unit, box=np.array([[70.0, 0, 0], [5.0, 70.0, 0], [0, 0, 70.0]]), seed_number=4)


@needs_voro
@pytest.mark.parametrize("case", ["fcc_rattled", "random_gas", "slab_open_z", "thin_box_3cells"])
def test_voronoi_cell_info_vs_reference_library(case):
    """get_cell_info (src/voronoi.cpp:449-540): the same polyhedra as voro++ builds — faces, vertices, areas, volume — up to the
    order in which faces and vertices are listed (internal to either construction)"""
    name, pos, box, origin, bd = next(c for c in _cases() if c[0] == case)
    pos = pos[:400].copy() if case == "random_gas" else pos[:1200].copy()
    if case == "fcc_rattled":  # a sub-block of the crystal in an open box: walls and bulk cells
        bd = np.array([0, 0, 0], np.int32)
        pos = pos[(pos < 14.0).all(1)]
        box = np.eye(3) * 14.5
    if case == "slab_open_z":
        pos[:, 2] = np.clip(pos[:, 2], 1e-3, box[2, 2] - 1e-3)
    x, y, z = _xyz(pos)
    fi0, fp0, v0, r0, a0 = O.get_cell_info(x, y, z, box, origin, bd)
    fi1, fp1, v1, r1, a1 = _voronoi.get_cell_info(x, y, z, box, origin, bd, 1)
    assert np.allclose(v1, v0, rtol=1e-9, atol=1e-9) and np.allclose(r1, r0, rtol=1e-9, atol=1e-9)
    scale = float(np.max(r0))
    for i in range(len(pos)):
        assert len(fi1[i]) == len(fi0[i]) and len(fp1[i]) == len(fp0[i]), i
        if not len(fp0[i]):  # not in the reference's container (outside on an open axis): no cell on either side
            continue
        assert sorted(len(f) for f in fi1[i]) == sorted(len(f) for f in fi0[i])
        assert np.allclose(np.sort(a1[i]), np.sort(a0[i]), rtol=1e-7, atol=1e-9 * scale ** 2)
        p0, p1 = np.array(fp0[i]), np.array(fp1[i])
        k0, k1 = np.lexsort(np.round(p0, 6).T), np.lexsort(np.round(p1, 6).T)
        assert np.allclose(p1[k1], p0[k0], rtol=0, atol=1e-7 * scale)
        n_edges = sum(len(f) for f in fi1[i]) // 2
        assert len(fp1[i]) - n_edges + len(fi1[i]) == 2  # Euler: the merged faces close up into one polyhedron
        for f, ar in zip(fi1[i], a1[i]):  # the index lists really are the polygons: area of the fan equals the reported area
            q = p1[f]
            cr = np.cross(q[1:-1] - q[0], q[2:] - q[0]).sum(0)
            assert abs(0.5 * np.linalg.norm(cr) - ar) <= 1e-9 * scale ** 2 + 1e-9 * ar
    con = mp.voronoi.Container(np.ascontiguousarray(pos - origin), mp.Box(box, boundary=bd))
    assert len(con) == len(pos) and abs(con[3].volume - v0[3]) < 1e-9 * v0[3] and con[3].vertices.shape == (len(fp0[3]), 3)


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[5], CASES[9]], ids=[CASES[0][0], CASES[2][0], CASES[5][0], CASES[9][0]])
def test_neighbor_ordering_key_equals_reordered_system(case):
    """build_neighbor(key=...) (multi-GPU extension): "descending key inside a cell" must give, for every atom, the row the
    reference gives after the atoms have been renumbered in ascending key order"""
    _, pos, box, org, bnd = case
    x, y, z = _xyz(pos)
    N, rc = len(x), 3.3
    key = np.random.default_rng(3).permutation(N).astype(np.int64) * 7 + 5  # arbitrary distinct ids
    perm = np.argsort(key, kind="stable")
    v0, d0, n0 = O.build_neighbor_without_max_neigh(x[perm].copy(), y[perm].copy(), z[perm].copy(), box, org, bnd, rc, 4)
    M = v0.shape[1]
    v1, d1, n1 = np.zeros((N, M), np.int32), np.zeros((N, M)), np.zeros(N, np.int32)
    _neighbor.build_neighbor(x, y, z, box, org, bnd, rc, v1, d1, n1, 1, fill_pads=True, key=key)
    assert np.array_equal(n1[perm], n0) and np.array_equal(d1[perm], d0)
    assert np.array_equal(v1[perm], np.where(v0 >= 0, perm[np.clip(v0, 0, None)], -1))
    v2, d2, n2 = np.zeros((N, M), np.int32), np.zeros((N, M)), np.zeros(N, np.int32)
    _neighbor.build_neighbor(x, y, z, box, org, bnd, rc, v2, d2, n2, 1, fill_pads=True, key=np.arange(N, dtype=np.int64))
    va, da, na = O.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, rc, 4)
    assert np.array_equal(v2[:, : va.shape[1]], va) and np.array_equal(n2, na)  # key = index: the plain build


def test_slab_halo_selection_kernel_equals_its_torch_definition():
    import torch

    from mdapy_amd.distributed import SlabDecomposition

    rng = np.random.default_rng(8)
    tri = np.array([[40.0, 0.0, 0.0], [6.0, 33.0, 0.0], [-4.0, 5.0, 28.0]])
    for box, axis in ((mp.Box(np.diag([50.0, 20.0, 30.0])), 0), (mp.Box(tri, origin=np.array([1.0, -2.0, 3.0])), 0), (mp.Box(tri), 2)):
        pos = (rng.random((200_000, 3)) * 1.4 - 0.2) @ box.box + box.origin  # some atoms outside the box: wrapped ownership
        x, y, z = (torch.from_numpy(np.ascontiguousarray(pos[:, k])).cuda() for k in range(3))
        for rank, world in ((0, 4), (3, 4), (1, 2)):
            dec = SlabDecomposition(box, rank, world, axis=axis)
            h = dec.halo_fraction(3.3)
            lo, hi = rank / world, (rank + 1) / world
            f = dec.frac(x, y, z)
            up0 = (f >= hi - h).nonzero().flatten()
            down0 = (f < lo + h).nonzero().flatten()
            up1, down1 = dec._select_device(x, y, z, hi - h, lo + h)
            assert torch.equal(torch.sort(up1).values, up0) and torch.equal(torch.sort(down1).values, down0)
            assert 0 < len(up0) < len(f)
            gid = torch.arange(len(f), dtype=torch.int64, device=x.device) * 3 + 1
            iu, idn, ru, rd = dec._select_device(x, y, z, hi - h, lo + h, gid)
            for idx, rows in ((iu.long(), ru), (idn.long(), rd)):
                assert torch.equal(rows, torch.stack([x[idx], y[idx], z[idx], gid[idx].double()], dim=1))


@pytest.mark.parametrize("case", ["fcc_hot_shifted_origin", "triclinic_random", "random_gas", "thin_box_3cells", "cluster_open", "dense_blob"])
def test_filter_overlap_atom_with_grain_vs_oracle(case):
    """the order-dependent sweep of src/neighbor.cpp:489-672 (serial order) from priority-ordered parallel rounds"""
    name, pos, box, origin, bd = next(c for c in _cases() if c[0] == case)
    pos = pos[:2500]
    x, y, z = _xyz(pos)
    rng = np.random.default_rng(21)
    ty = rng.choice([1, 2], len(pos), p=[0.7, 0.3]).astype(np.int32)
    gr = rng.integers(1, 6, len(pos)).astype(np.int32)
    for mm, cc, mc in ((2.4, 1.9, 2.9), (3.3, 3.3, 3.3), (1.0, 2.8, 1.6)):
        k0 = O.filter_overlap_atom_with_grain(x, y, z, ty, gr, box, origin, bd, mm, cc, mc)
        k1 = _neighbor.filter_overlap_atom_with_grain(x, y, z, ty, gr, box, origin, bd, mm, cc, mc, 1)
        assert np.array_equal(k1, k0)
    assert 0 < k0.sum() < len(k0)


def test_create_polycrystal_with_graphene_hip_equals_oracle_build(monkeypatch):
    """graphene-decorated grain boundaries.  Generated atoms (before the overlap sweep): the HIP build and the oracle-routed
    build agree atom for atom, up to atoms on the edge of a face polygon (the two constructions of the cells list faces and
    vertices in different orders, and the sheets are cut in single precision).  After the sweep only the rules can be checked:
    which partner of an overlapping pair survives depends on the order in which voro++ lists the faces."""
    import _oracle_backend as ob
    from mdapy_amd import devarray
    from mdapy_amd.voronoi import Container

    unit = mp.build_crystal("Al", "fcc", 4.05)
    kw = dict(box=55.0, seed_number=5, randomseed=4, metal_overlap_dis=2.0, add_graphene=True, face_threshold=5.0, metal_gra_overlap_dis=3.0)

    def generated(pc):
        pc.con = Container(np.ascontiguousarray(pc.seed_position), mp.Box(pc.box.box))
        pos, grain, ty = pc._get_pos()
        q = np.round(pos * 1e5).astype(np.int64)
        return set(map(tuple, np.c_[q, ty, grain].tolist()))

    pa = mp.CreatePolycrystal(unit, **kw)
    ga = generated(pa)
    with monkeypatch.context() as m:
        ob.install(m)
        m.setattr(devarray, "_gpu", False)
        pb = mp.CreatePolycrystal(unit, **kw)
        gb = generated(pb)
        b = pb.compute()
    assert len(ga ^ gb) <= 0.002 * len(ga)
    a = pa.compute()
    assert abs(a.N - b.N) <= 0.02 * a.N
    ty, gr = a.data["type"].to_numpy(), a.data["grain_id"].to_numpy()
    assert set(np.unique(ty).tolist()) == {1, 2} and set(a.data["element"].to_numpy()[ty == 2].tolist()) == {"C"}
    pos = np.c_[a.data["x"].to_numpy(), a.data["y"].to_numpy(), a.data["z"].to_numpy()]
    # every carbon atom lies (within the 0.5 A slab of the sheet) on a bisector plane between its grain's seed and another seed image
    seeds = pb.seed_position
    L = 55.0
    car = np.flatnonzero(ty == 2)[::7]
    shifts = np.array([[i, j, k] for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)]) * L
    others = (seeds[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
    best = np.zeros(len(car))
    for q, c in enumerate(car):
        dist = np.sort(np.linalg.norm(others - pos[c], axis=1))
        best[q] = dist[1] - dist[0]  # the two nearest seed images are (almost) equally far: the atom is on their bisector plane
    assert best.max() < 1.1  # at most 0.5 A off the plane: the two distances differ by less than 2 * 0.5 A
    # no metal closer than 3.0 A to a carbon atom, no two metals closer than 2.0 A
    a.build_neighbor(3.0 - 1e-9)
    v, d, nn = np.asarray(a.verlet_list), np.asarray(a.distance_list), np.asarray(a.neighbor_number)
    valid = np.arange(v.shape[1])[None, :] < nn[:, None]
    tj = ty[np.clip(v, 0, None)]
    assert not (valid & (ty[:, None] != tj)).any()
    assert not (valid & (ty[:, None] == 1) & (tj == 1) & (d <= 2.0 - 1e-9)).any()


@needs_voro
def test_voronoi_few_atoms_in_a_small_periodic_cell():
    """2 ... 8 atoms in a cell of a few Angstrom: every face is shared with a periodic image; an atom that has nobody within
    the first search radius must not come back with an empty cell"""
    for N in (2, 3, 4, 8):
        for seed in range(6):
            rng = np.random.default_rng(100 * N + seed)
            L = rng.uniform(6, 11, 3)
            box, pos = np.diag(L), rng.random((N, 3)) * L
            x, y, z = _xyz(pos)
            v0, n0, r0 = np.zeros(N), np.zeros(N, np.int32), np.zeros(N)
            O.get_voronoi_volume_number_radius(x, y, z, box, ORG0, PBC, v0, n0, r0)
            v1, n1, r1 = np.zeros(N), np.zeros(N, np.int32), np.zeros(N)
            _voronoi.get_voronoi_volume_number_radius(x, y, z, box, ORG0, PBC, v1, n1, r1)
            assert np.allclose(v1, v0, rtol=1e-9) and np.allclose(r1, r0, rtol=1e-9) and np.array_equal(n1, n0)
            assert abs(v1.sum() - L.prod()) < 1e-9 * L.prod()
    v1, n1, r1 = np.zeros(1), np.zeros(1, np.int32), np.zeros(1)
    _voronoi.get_voronoi_volume_number_radius(np.array([1.0]), np.array([2.0]), np.array([3.0]), np.diag([4.0, 5.0, 6.0]), ORG0,
                                              np.zeros(3, np.int32), v1, n1, r1)
    assert abs(v1[0] - 120.0) < 1e-12 and n1[0] == 6  # a lone atom in an open box owns the box


def test_exact_width_build_with_a_stale_width_hint():
    """mdh_build_neighbor_exact remembers the row width per (N, grid) for HBM-resident calls and skips the counting pass when the
    counts of the build confirm it; a system with the same signature but another maximum makes it allocate twice — the result
    must not depend on the hint in either direction (wider, narrower, equal)"""
    import torch

    pos, box = _fcc(12, 0.02, 1)
    rng = np.random.default_rng(5)
    variants = [pos, pos.copy(), pos.copy(), pos]
    variants[1][:40] = pos[40:80] + rng.normal(0, 0.4, (40, 3))   # a clump: larger maximum
    variants[2] = pos + rng.normal(0, 0.01, pos.shape)            # back to the crystal's maximum
    rc = 3.2
    for p in variants:
        x, y, z = _xyz(p)
        v0, d0, n0 = O.build_neighbor_without_max_neigh(x, y, z, box, ORG0, PBC, rc, 4)
        tx, ty, tz = (torch.from_numpy(a).cuda() for a in (x, y, z))
        v1, d1, n1 = _neighbor.build_neighbor_without_max_neigh(tx, ty, tz, box, ORG0, PBC, rc, 1)
        assert np.asarray(v1).shape == v0.shape
        assert np.array_equal(np.asarray(n1), n0) and np.array_equal(np.asarray(v1), v0) and np.array_equal(np.asarray(d1), d0)
