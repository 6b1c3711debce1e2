"""The cell-sorted twin of ``System`` (mdapy_amd/system.py): a system handed in in no spatial order is analysed on a sorted
copy and every result the user reads is translated back.  Checked here on the CPU with the kernels routed to the oracle
(fixture ``oracle_backend``; the sort itself is a numpy stand-in — any permutation serves the host logic): every analysis
of a shuffled system with the twin forced on (MDAPY_SPATIAL_SORT=1) against the same system analysed in the order it has
(MDAPY_SPATIAL_SORT=0) — labels, lists (ids, ROW ORDER, distances, counts) and floating-point columns bit for bit."""
import numpy as np
import pytest

import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions
from mdapy_amd.devarray import as_numpy


def _system(monkeypatch, mode, seed=3, with_types=False, cells=(7, 6, 6)):
    monkeypatch.setenv("MDAPY_SPATIAL_SORT", mode)
    pos, box = lattice_positions("fcc", 3.615, *cells)
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0, 0.12, pos.shape)
    order = rng.permutation(len(pos))
    pos = pos[order]
    data = {"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2]}
    if with_types:
        data["type"] = rng.integers(1, 3, len(pos)).astype(np.int32)
        data["vx"], data["vy"], data["vz"] = (rng.normal(0, 1.0, len(pos)) for _ in range(3))
        data["amass"] = np.where(data["type"] == 1, 63.5, 91.2)
    return mp.System(data=data, box=box)


def _same_columns(a, b, names):
    for n in names:
        va, vb = a.data[n].to_numpy(), b.data[n].to_numpy()
        assert va.dtype == vb.dtype and np.array_equal(va, vb, equal_nan=True), n


def test_twin_is_made_and_lists_translate_bit_for_bit(oracle_backend, monkeypatch):
    plain, twin = _system(monkeypatch, "0"), _system(monkeypatch, "1")
    plain.build_neighbor(3.9, max_neigh=30)
    twin.build_neighbor(3.9, max_neigh=30)
    assert plain._spatial() is None
    t = twin._spatial()
    assert t is not None and sorted(np.asarray(t._perm).tolist()) == list(range(twin.N))
    for name in ("verlet_list", "distance_list", "neighbor_number"):
        assert np.array_equal(as_numpy(getattr(plain, name)), as_numpy(getattr(twin, name))), name
    assert twin.rc == plain.rc == 3.9
    # exact-width rows
    plain.build_neighbor(3.3); twin.build_neighbor(3.3)
    for name in ("verlet_list", "distance_list", "neighbor_number"):
        assert np.array_equal(as_numpy(getattr(plain, name)), as_numpy(getattr(twin, name))), name
    # k nearest
    plain.build_nearest_neighbor(9)
    twin.build_nearest_neighbor(9)
    assert np.array_equal(as_numpy(plain.distance_list), as_numpy(twin.distance_list))
    assert np.array_equal(as_numpy(plain.verlet_list), as_numpy(twin.verlet_list))  # (no exact ties in a rattled crystal)
    assert "rc" in twin.__dict__  # (the reference keeps rc behind a k-nearest list, system.py:1256-1263)


def test_every_twin_analysis_equals_the_plain_one(oracle_backend, monkeypatch):
    out = {}
    for mode in ("0", "1"):
        s = _system(monkeypatch, mode, with_types=True)
        s.cal_common_neighbor_analysis(rc=0.854 * 3.615)
        s.cal_common_neighbor_analysis()
        s.data  # noqa
        cna_adaptive = s.data["cna"].to_numpy().copy()
        s.cal_centro_symmetry_parameter(12)
        s.cal_ackland_jones_analysis()
        s.cal_common_neighbor_parameter(3.6)
        s.cal_structure_entropy(4.0, 0.2, use_local_density=True, average_rc=3.2)
        s.cal_atomic_temperature(4.0)
        s.cal_steinhardt_bond_orientation([4, 6], rc=3.3, wl=True, wlhat=True, average=True)
        s.average_by_neighbor(3.4, "csp")
        g = s.cal_radial_distribution_function(4.5, nbin=40)
        w = s.cal_warren_cowley_parameter(3.4)
        s.cal_polyhedral_template_matching("fcc-hcp-bcc", return_rmsd=True, return_ordering=True, identify_fcc_planar_faults=True)
        s.cal_cluster_analysis(2.7)                 # numbering depends on the atom order: runs on the translated list
        s.cal_identify_diamond_structure()
        s.cal_steinhardt_bond_orientation([6], nnn=12, identify_liquid=True)
        out[mode] = (s, cna_adaptive, g, w)
    (a, ca, ga, wa), (b, cb, gb, wb) = out["0"], out["1"]
    assert a._spatial() is None and b._spatial() is not None
    assert np.array_equal(ca, cb)
    assert set(a.data.columns) == set(b.data.columns)
    _same_columns(a, b, [c for c in a.data.columns])
    assert np.array_equal(ga.g_total, gb.g_total) and set(ga.g_partial) == set(gb.g_partial)
    assert all(np.array_equal(ga.g_partial[k], gb.g_partial[k]) for k in ga.g_partial)
    assert np.array_equal(wa.WCP, wb.WCP)
    assert a.cluster_number == b.cluster_number
    assert np.array_equal(as_numpy(a.ptm_indices), as_numpy(b.ptm_indices))
    assert np.array_equal(as_numpy(a.verlet_list), as_numpy(b.verlet_list))


def test_user_weight_rows_keep_the_steinhardt_call_off_the_twin(oracle_backend, monkeypatch):
    """cal_steinhardt_bond_orientation(use_weight=True, weight=<array>): the weight rows line up with THIS system's list rows,
    which the twin holds permuted — the call must run in the system's own order: twin == plain bit for bit"""
    out = {}
    for mode in ("0", "1"):
        s = _system(monkeypatch, mode)
        s.build_neighbor(3.3, max_neigh=20)
        rows = as_numpy(s.verlet_list)
        w = np.where(rows >= 0, 1.0 + (rows % 7) * 0.25, 0.0)  # a weight that depends on the NEIGHBOUR's number: any row mix-up shows
        assert s._twin_for("cal_steinhardt_bond_orientation", ([4, 6],), dict(rc=3.3, use_weight=True, weight=w)) is None
        s.cal_steinhardt_bond_orientation([4, 6], rc=3.3, use_weight=True, weight=w)
        out[mode] = s
    _same_columns(out["0"], out["1"], ["ql4", "ql6"])
    assert out["1"]._spatial() is not None


def test_twin_follows_data_and_box_changes(oracle_backend, monkeypatch):
    s = _system(monkeypatch, "1", with_types=True)
    s.cal_centro_symmetry_parameter(12)
    first = s._spatial()
    assert first is not None
    # new per-atom columns reach the twin through the permutation
    s.update_data(s.data.with_columns(score=np.arange(s.N, dtype=np.float64)))
    s.average_by_neighbor(3.4, "score")
    assert s._spatial() is first
    ref = _system(monkeypatch, "0", with_types=True)
    ref.update_data(ref.data.with_columns(score=np.arange(ref.N, dtype=np.float64)))
    ref.average_by_neighbor(3.4, "score")
    assert np.array_equal(ref.data["score_ave"].to_numpy(), s.data["score_ave"].to_numpy())
    # forgetting the list forgets the twin's
    s.update_data(s.data, reset_neighbor=True)
    assert "verlet_list" not in s.__dict__ and "verlet_list" not in first.__dict__
    # a new box: a new twin; moved atoms: a new twin
    s.box = mp.Box(s.box.box * 1.0)
    s.build_neighbor(3.3)
    second = s._spatial()
    assert second is not None and second is not first
    s.wrap_pos()
    s.build_neighbor(3.3)
    assert s._spatial() is not second
    # a list the user put there himself is not the twin's: the plain path takes over
    s.verlet_list = as_numpy(s.verlet_list).copy()
    assert s._twin_for("cal_common_neighbor_parameter", (3.2,), {}) is None


def test_perfect_lattice_ties_fall_as_in_the_original_numbering(oracle_backend, monkeypatch):
    """a perfect bcc lattice: the twelve nearest of an atom are its eight first neighbours and FOUR of its six second ones — which
    four, the search decides by atom number; the twin's searches carry the original numbers as their tie-breaking key"""
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MDAPY_SPATIAL_SORT", mode)
        pos, box = lattice_positions("bcc", 3.2, 6, 5, 5)
        pos = pos[np.random.default_rng(2).permutation(len(pos))]
        s = mp.System(pos=pos, box=box)
        s.cal_steinhardt_bond_orientation([4, 6], nnn=12)
        s.cal_centro_symmetry_parameter(8)
        s.build_nearest_neighbor(12)
        out[mode] = (s, as_numpy(s.verlet_list).copy())
    assert out["0"][0]._spatial() is None and out["1"][0]._spatial() is not None
    assert np.array_equal(out["0"][1], out["1"][1])
    _same_columns(out["0"][0], out["1"][0], ["ql4", "ql6", "csp"])
