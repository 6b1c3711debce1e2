"""CPU suite (-m "not gpu"): pins the oracle (oracle/mdapy_oracle.c) — driven through mdapy_amd's own
host-side policy layer — against the golden vectors of the reference's test-suite.

Each test mirrors a reference test (cited), with ``import mdapy as mp`` replaced by
``import mdapy_amd as mp`` and the native backend replaced by the oracle adapters
(fixture ``oracle_backend``).  The same flows run against the HIP kernels in
tests/test_gpu_parity.py.
"""
import json

import numpy as np
import pytest

import mdapy_amd as mp
from _golden import GOLDEN, fixtures_with, ids_of, input_path, misc, system_from_fixture

CNA_PATHS = fixtures_with("cna")
CSP_PATHS = fixtures_with("csp")
QL_PATHS = fixtures_with("q4")
IDS_PATHS = fixtures_with("ids")


# reference: tests/test_common_neighbor_analysis.py:19-29
@pytest.mark.parametrize("path", CNA_PATHS, ids=ids_of(CNA_PATHS))
def test_cna_against_fixture(path, oracle_backend):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_common_neighbor_analysis(rc=float(d["cna_cutoff"]))
    got = s.data["cna"].to_numpy()
    assert int(np.sum(got != d["cna"])) == 0


# reference: tests/test_common_neighbor_analysis.py:32-46
def test_cna_perfect_crystals(oracle_backend):
    a = 4.05
    fcc = mp.build_crystal("Al", "fcc", a, nx=4, ny=4, nz=4)
    fcc.cal_common_neighbor_analysis(rc=0.854 * a)
    assert np.all(fcc.data["cna"].to_numpy() == 1)
    bcc = mp.build_crystal("Fe", "bcc", 2.86, nx=4, ny=4, nz=4)
    bcc.cal_common_neighbor_analysis(rc=1.21 * 2.86)
    assert np.all(bcc.data["cna"].to_numpy() == 3)
    hcp = mp.build_crystal("Mg", "hcp", 3.21, nx=4, ny=4, nz=3)
    hcp.cal_common_neighbor_analysis(rc=1.207 * 3.21)
    assert np.all(hcp.data["cna"].to_numpy() == 2)


def test_adaptive_cna_perfect_crystals(oracle_backend):
    fcc = mp.build_crystal("Al", "fcc", 4.05, nx=4, ny=4, nz=4)
    fcc.cal_common_neighbor_analysis()
    assert np.all(fcc.data["cna"].to_numpy() == 1)
    bcc = mp.build_crystal("Fe", "bcc", 2.86, nx=6, ny=6, nz=6)
    bcc.cal_common_neighbor_analysis()
    assert np.all(bcc.data["cna"].to_numpy() == 3)
    hcp = mp.build_crystal("Mg", "hcp", 3.21, nx=5, ny=5, nz=4)
    hcp.cal_common_neighbor_analysis()
    assert np.all(hcp.data["cna"].to_numpy() == 2)


# reference: tests/test_centro_symmetry_parameter.py:18-27
@pytest.mark.parametrize("path", CSP_PATHS, ids=ids_of(CSP_PATHS))
def test_csp_against_fixture(path, oracle_backend):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_centro_symmetry_parameter(int(d["csp_num_neighbors"]))
    got = s.data["csp"].to_numpy()
    assert np.allclose(got, d["csp"], atol=1e-6, rtol=1e-6), np.abs(got - d["csp"]).max()


# reference: tests/test_steinhardt_bond_orientation.py:25-47
@pytest.mark.parametrize("path", QL_PATHS, ids=ids_of(QL_PATHS))
def test_ql_against_fixture(path, oracle_backend):
    d = np.load(path)
    s = system_from_fixture(d)
    rc = float(d["ql_cutoff"])
    s.cal_steinhardt_bond_orientation([4, 6], rc=rc)
    for l in (4, 6):
        got = s.data[f"ql{l}"].to_numpy()
        assert np.allclose(got, d[f"q{l}"], atol=1e-6, rtol=1e-6), (l, np.abs(got - d[f"q{l}"]).max())
    s.cal_steinhardt_bond_orientation([4, 6], rc=rc, average=True)
    for l in (4, 6):
        got = s.data[f"ql{l}"].to_numpy()
        assert np.allclose(got, d[f"q{l}_avg"], atol=1e-6, rtol=1e-6), (l, np.abs(got - d[f"q{l}_avg"]).max())


# reference: tests/test_steinhardt_bond_orientation.py:50-60
def test_ql_perfect_fcc_known_values(oracle_backend):
    a = 4.05
    s = mp.build_crystal("Al", "fcc", a, nx=4, ny=4, nz=4)
    s.cal_steinhardt_bond_orientation([4, 6], rc=0.95 * a)
    assert np.allclose(s.data["ql4"].to_numpy(), 0.190941, atol=1e-5)
    assert np.allclose(s.data["ql6"].to_numpy(), 0.574524, atol=1e-5)
    s.cal_steinhardt_bond_orientation([4, 6], nnn=12, wl=True, wlhat=True)
    assert np.allclose(s.data["ql6"].to_numpy(), 0.574524, atol=1e-5)
    assert np.allclose(s.data["wlh6"].to_numpy(), -0.013161, atol=1e-5)  # FCC w6-hat (Steinhardt 1983)


# reference: tests/test_identify_diamond.py (fixture-driven labels)
@pytest.mark.parametrize("path", IDS_PATHS, ids=ids_of(IDS_PATHS))
def test_ids_against_fixture(path, oracle_backend):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_identify_diamond_structure()
    assert int(np.sum(s.data["ids"].to_numpy() != d["ids"])) == 0


# reference: tests/test_radial_distribution_function.py:11-25
def test_rdf_fixture(oracle_backend):
    d = misc("rdf")
    s = mp.System(input_path("AlCrNi.xyz"))
    rdf = s.cal_radial_distribution_function(float(d["cutoff"]), int(d["nbins"]))
    el = list(d["elements"])
    for i in range(len(el)):
        for j in range(i, len(el)):
            assert np.allclose(rdf.g_partial[(el[i], el[j])], d["g"][i, j], atol=1e-6), (el[i], el[j])


# reference: tests/test_rdf_streaming.py:15-53 (streaming == verlet path)
def test_rdf_streaming_equals_verlet(oracle_backend):
    s = mp.build_crystal("Cu", "fcc", 3.615, nx=6, ny=6, nz=6)
    a = s.cal_radial_distribution_function(5.0, 80, streaming=False)
    b = s.cal_radial_distribution_function(5.0, 80, streaming=True)
    assert np.allclose(a.g_total, b.g_total, atol=1e-12, rtol=1e-9)
    s2 = mp.System(input_path("AlCrNi.xyz"))
    a = s2.cal_radial_distribution_function(4.0, 60, streaming=False)
    s3 = mp.System(input_path("AlCrNi.xyz"))
    b = s3.cal_radial_distribution_function(4.0, 60, streaming=True)
    for k in a.g_partial:
        assert np.allclose(a.g_partial[k], b.g_partial[k], atol=1e-12, rtol=1e-9), k


# reference: tests/test_warren_cowley_parameter.py:7-22
def test_wcp_known_answer(oracle_backend):
    s = mp.System(input_path("CoCuFeNiPd-4M.dump"))
    wcp = s.cal_warren_cowley_parameter(rc=3.0)
    ref = np.array([[-1.39, 0.64, 0.39, -0.3, 0.66], [0.64, -1.94, 0.58, 0.51, 0.2], [0.39, 0.58, -0.56, 0.63, -1.04],
                    [-0.3, 0.51, 0.63, -1.69, 0.85], [0.66, 0.2, -1.04, 0.85, -0.67]])
    assert np.allclose(wcp.WCP.round(2), ref)


# reference: tests/test_average_neighbor.py
@pytest.mark.parametrize("name", ["rec_box_big", "tri_box_big"])
def test_average_neighbor(name, oracle_backend):
    d = misc("average_neighbor")
    s = mp.System(input_path(f"{name}.xyz"))
    s.average_by_neighbor(float(d[f"{name}__cutoff"]), "x", include_self=True)
    assert np.allclose(s.data["x_ave"].to_numpy(), d[f"{name}__x_ave"], atol=1e-6)


def test_knife_edge_neighbor_counts(oracle_backend):
    """rc == a on FCC Cu: counts are decided by fp64 rounding; the histogram below was recorded from the
    reference C++ and pins the operation order of wrap / pbc / d2."""
    ref = json.load(open(GOLDEN / "knife_edge_counts.json"))["counts"]
    s = mp.build_crystal("Cu", "fcc", 3.615, nx=10, ny=10, nz=10)
    s.build_neighbor(3.615)
    got = np.bincount(np.asarray(s.neighbor_number))
    assert {str(k): int(v) for k, v in enumerate(got) if v} == ref
    assert s.verlet_list.shape == (4000, 18)


# reference: tests/test_neighbor_cutoff.py:23-88 (brute-force minimum-image check; order is not pinned)
def _bf(idx, pos, box, rc):
    rij = pos - pos[idx]
    frac = rij @ box.inverse_box
    frac -= np.round(frac) * np.asarray(box.boundary, dtype=float)
    dist = np.linalg.norm(frac @ box.box, axis=1)
    mask = dist <= rc + 1e-9
    mask[idx] = False
    inds = np.nonzero(mask)[0]
    return inds, dist[inds]


@pytest.mark.parametrize("filename", ["rec_box_big.xyz", "rec_box_small.xyz", "tri_box_big.xyz", "tri_box_small.xyz",
                                      "AlCrNi.xyz", "HexDiamond.xyz"])
@pytest.mark.parametrize("rc", [2.5, 5.0])
@pytest.mark.parametrize("max_neigh", [None, 150])
def test_neighbor_files_bruteforce(filename, rc, max_neigh, oracle_backend):
    s = mp.System(input_path(filename))
    s.build_neighbor(rc, max_neigh)
    box, data = s._get_compute_view()
    pos = data.select("x", "y", "z").to_numpy()
    for i in sorted({0, s.N // 2, s.N - 1}):
        ref_idx, ref_dist = _bf(i, pos, box, rc)
        nn = int(s.neighbor_number[i])
        assert nn == len(ref_idx)
        got_idx = np.asarray(s.verlet_list[i, :nn])
        o1, o2 = np.argsort(got_idx), np.argsort(ref_idx)
        assert np.array_equal(got_idx[o1], ref_idx[o2])
        assert np.allclose(np.asarray(s.distance_list[i, :nn])[o1], ref_dist[o2], atol=1e-6)


# reference: tests/test_neighbor_cutoff.py:214-258
def test_max_neigh_too_small_raises(oracle_backend):
    s = mp.build_crystal("Cu", "fcc", 3.615, nx=5, ny=5, nz=5)
    with pytest.raises(ValueError, match="max_neigh=5 is too small"):
        s.build_neighbor(3.0, max_neigh=5)
    s.build_neighbor(3.0, max_neigh=12)
    assert int(np.asarray(s.neighbor_number).max()) == 12
    with pytest.raises(AssertionError, match="rc must be positive"):
        s.build_neighbor(-1.0)
    with pytest.raises(AssertionError, match="max_neigh must be positive"):
        s.build_neighbor(3.0, max_neigh=0)


def test_system_cache_invalidation(oracle_backend):
    s = mp.build_crystal("Cu", "fcc", 3.615, nx=5, ny=5, nz=5)
    s.build_neighbor(3.0)
    assert hasattr(s, "verlet_list") and hasattr(s, "rc")
    s.box = mp.Box(s.box.box * 1.0)
    assert not hasattr(s, "verlet_list") and not hasattr(s, "rc")
    s.build_neighbor(3.0)
    s.update_data(s.data, reset_neighbor=True)
    assert not hasattr(s, "neighbor_number")


# ---------------------------------------------------------------- PTM: oracle/_ref = the reference's own PTM library
from oracle import oracle as _O

PTM_PATHS = fixtures_with("ptm")
needs_ref = pytest.mark.skipif(not _O.have_ref(), reason="oracle/_ref/libptm_ref.so not built (needs /root/reference)")


# reference: tests/test_polyhedral_template_matching.py:21-31
@needs_ref
@pytest.mark.parametrize("path", PTM_PATHS, ids=ids_of(PTM_PATHS))
def test_ptm_against_fixture(path, oracle_backend):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_polyhedral_template_matching()
    assert int(np.sum(s.data["ptm"].to_numpy() != d["ptm"])) == 0


# reference: tests/test_polyhedral_template_matching.py:34-58
@needs_ref
def test_ptm_perfect_crystals(oracle_backend):
    fcc = mp.build_crystal("Al", "fcc", 4.05, nx=4, ny=4, nz=4)
    fcc.cal_polyhedral_template_matching()
    assert np.all(fcc.data["ptm"].to_numpy() == 1)
    bcc = mp.build_crystal("Fe", "bcc", 2.86, nx=4, ny=4, nz=4)
    bcc.cal_polyhedral_template_matching()
    assert np.all(bcc.data["ptm"].to_numpy() == 3)
    hcp = mp.build_crystal("Mg", "hcp", 3.21, nx=4, ny=4, nz=3)
    hcp.cal_polyhedral_template_matching()
    assert np.all(hcp.data["ptm"].to_numpy() == 2)
    dia = mp.build_crystal("C", "diamond", 3.5, nx=3, ny=3, nz=3)
    dia.cal_polyhedral_template_matching(structure="all")
    assert np.all(dia.data["ptm"].to_numpy() == 6)


# ---------------------------------------------------------------- list consumers: AJA, CNP, structure entropy
AJA_PATHS, CNP_PATHS = fixtures_with("aja"), fixtures_with("cnp")


# reference: tests/test_ackland_jones_analysis.py (fixture-driven, OVITO labels)
@pytest.mark.parametrize("path", AJA_PATHS, ids=ids_of(AJA_PATHS))
def test_aja_against_fixture(path, oracle_backend):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_ackland_jones_analysis()
    assert np.array_equal(s.data["aja"].to_numpy(), d["aja"])


# reference: tests/test_common_neighbor_parameter.py:19-52
@pytest.mark.parametrize("path", CNP_PATHS, ids=ids_of(CNP_PATHS))
def test_cnp_against_fixture(path, oracle_backend):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_common_neighbor_parameter(float(d["cnp_cutoff"]))
    assert np.allclose(s.data["cnp"].to_numpy(), d["cnp"], atol=1e-6, rtol=1e-6)


def test_cnp_perfect_crystals(oracle_backend):
    a = 3.615
    s = mp.build_crystal("Cu", "fcc", a, nx=4, ny=4, nz=4)
    s.cal_common_neighbor_parameter(0.86 * a)
    assert np.allclose(s.data["cnp"].to_numpy().max(), 0.0)
    s = mp.build_crystal("Cu", "bcc", a, nx=4, ny=4, nz=4)
    s.cal_common_neighbor_parameter(1.21 * a)
    assert np.allclose(s.data["cnp"].to_numpy().max(), 0.0)
    s = mp.build_crystal("Cu", "hcp", a, c=a * 1.633)
    s.cal_common_neighbor_parameter(1.21 * a)
    assert np.allclose(s.data["cnp"].to_numpy().max(), 8.71215)


# reference: tests/test_structure_entropy.py:11-33
@pytest.mark.parametrize("name", ["rec_box_big", "rec_box_small", "tri_box_big", "tri_box_small"])
@pytest.mark.parametrize("mode", ["default", "use_local_density", "compute_average"])
def test_structure_entropy_against_fixture(name, mode, oracle_backend):
    expected = misc("structure_entropy")[f"{name}__{mode}"]
    s = mp.System(input_path(f"{name}.xyz"))
    if mode == "compute_average":
        s.cal_structure_entropy(5.0, 0.2, False, average_rc=4.0)
        got = s.data["entropy_ave"].to_numpy()
    else:
        s.cal_structure_entropy(5.0, 0.2, mode == "use_local_density")
        got = s.data["entropy"].to_numpy()
    assert np.allclose(got, expected, atol=1e-6)


# ---------------------------------------------------------------- atomic temperature / cluster analysis (no reference fixtures:
# tests/test_atomic_temperature.py needs generate_velocity + the mass table; closed-form checks instead)
def test_atomic_temperature_closed_forms(oracle_backend):
    s = mp.build_crystal("Cu", "fcc", 3.615, nx=6, ny=6, nz=6)
    rng = np.random.default_rng(2)
    n = s.N
    s.update_data(s.data.with_columns(vx=np.full(n, 1.5), vy=np.full(n, -0.5), vz=np.zeros(n)))
    s.cal_atomic_temperature(5.0)
    assert np.allclose(s.data["atomic_temp"].to_numpy(), 0.0, atol=1e-9)      # rigid translation: no thermal motion
    # Maxwell velocities at T0: the neighbourhood temperature scatters around T0 * (n-1)/n (centre-of-mass removed)
    kb, amu, T0, m = 1.380649e-23, 1.0 / 6.022140857e23 / 1000.0, 300.0, 63.546
    sig = np.sqrt(kb * T0 / (m * amu)) / 1e5                                   # A/fs
    v = rng.normal(0, sig, (n, 3))
    s.update_data(s.data.with_columns(vx=v[:, 0], vy=v[:, 1], vz=v[:, 2]))
    s.cal_atomic_temperature(8.0)
    t = s.data["atomic_temp"].to_numpy()
    assert abs(t.mean() / T0 - 1.0) < 0.03 and t.min() > 150 and t.max() < 450


def test_cluster_analysis_known_components(oracle_backend):
    rng = np.random.default_rng(4)
    blobs = [rng.random((40, 3)) * 4.0 + c for c in ([5, 5, 5], [25, 25, 25], [5, 25, 5])]
    lone = np.array([[40.0, 40.0, 40.0], [45.0, 10.0, 30.0]])
    pos = np.concatenate(blobs + [lone])
    perm = rng.permutation(len(pos))
    pos = pos[perm]
    s = mp.System(pos=pos, box=np.eye(3) * 50.0)
    s.cal_cluster_analysis(3.0)
    cid = s.data["cluster_id"].to_numpy()
    # brute-force components, numbered by smallest member index
    d = np.linalg.norm(pos[:, None] - pos[None], axis=2)
    lab = np.arange(len(pos))
    for _ in range(len(pos)):
        new = np.array([lab[d[i] <= 3.0].min() for i in range(len(pos))])
        if np.array_equal(new, lab):
            break
        lab = new
    roots = np.unique(lab)
    expect = np.searchsorted(roots, lab) + 1
    assert np.array_equal(cid, expect) and s.cluster_number == len(roots)
    # type-pair cutoffs: bonds between unlike types are cut at 1.0
    ty = (rng.random(len(pos)) < 0.5).astype(np.int32) + 1
    s2 = mp.System(pos=pos, box=np.eye(3) * 50.0)
    s2.update_data(s2.data.with_columns(type=ty))
    s2.cal_cluster_analysis({"1-1": 3.0, "2-2": 3.0, "1-2": 1.0})
    bond = (d <= 3.0) & ((ty[:, None] == ty[None]) | (d <= 1.0))
    lab = np.arange(len(pos))
    for _ in range(len(pos)):
        new = np.array([lab[bond[i]].min() for i in range(len(pos))])
        if np.array_equal(new, lab):
            break
        lab = new
    roots = np.unique(lab)
    assert np.array_equal(s2.data["cluster_id"].to_numpy(), np.searchsorted(roots, lab) + 1)


# reference: tests/test_identify_fcc_planar_faults.py:9-19 (103 056-atom Cu dump with an intrinsic stacking fault)
@needs_ref
def test_fcc_planar_faults_against_fixture(oracle_backend):
    expected = misc("fcc_planar_faults")["pft"]
    s = mp.System(input_path("ISF.dump"))
    s.cal_polyhedral_template_matching("all", identify_fcc_planar_faults=True, identify_esf=False)
    got = s.data["pft"].to_numpy()
    assert np.array_equal(got, expected), f"{int(np.sum(got != expected))} mismatches; labels {np.bincount(expected)}"


def test_filter_overlap_atom_oracle_vs_brute_force():
    """src/neighbor.cpp:390-486 — an atom is dropped iff a lower-numbered atom lies within rc (minimum image)"""
    rng = np.random.default_rng(6)
    L = 12.0
    pos = rng.random((600, 3)) * L
    keep = _O.filter_overlap_atom(pos[:, 0].copy(), pos[:, 1].copy(), pos[:, 2].copy(), np.eye(3) * L, np.zeros(3),
                                  np.array([1, 1, 0], np.int32), 1.1, 2)
    d = pos[:, None, :] - pos[None, :, :]
    d[..., :2] -= L * np.round(d[..., :2] / L)
    close = (np.sqrt((d ** 2).sum(-1)) <= 1.1) & (np.arange(600)[None, :] < np.arange(600)[:, None])
    assert np.array_equal(keep, ~close.any(axis=1)) and 0 < keep.sum() < 600


# ---------------------------------------------------------------- Voronoi volume / faces / cavity radius: oracle/_ref = the
# reference's own voro++ behind the restated driver, pinned against the OVITO-derived fixtures
VOR_PATHS = fixtures_with("voronoi_volume")
needs_voro = pytest.mark.skipif(not _O.have_voro_ref(), reason="oracle/_ref/libvoro_ref.so not built (needs /root/reference)")


# reference: tests/test_voronoi.py:11-36
@needs_voro
@pytest.mark.parametrize("path", VOR_PATHS, ids=ids_of(VOR_PATHS))
def test_voronoi_against_fixture(path, oracle_backend):
    d = np.load(path)
    s = system_from_fixture(d)
    s.cal_voronoi_volume()
    assert np.allclose(d["voronoi_volume"], s.data["volume"].to_numpy(), atol=1e-6)
    assert np.allclose(d["voronoi_cavity_radius"], s.data["cavity_radius"].to_numpy() * 0.5, atol=1e-6)  # OVITO convention
    assert np.array_equal(d["voronoi_coord"], s.data["neighbor_number"].to_numpy())


@needs_voro
def test_voronoi_neighbors_and_voronoi_weighted_steinhardt_closed_forms(oracle_backend):
    """perfect fcc: 12 Voronoi neighbours with equal faces (rhombic dodecahedron, face area a^2 sqrt(2) / 8); the
    Voronoi-neighbour q6 equals the cutoff value 0.574524 (reference: tests/test_steinhardt_bond_orientation.py)"""
    a = 4.05
    s = mp.build_crystal("Al", "fcc", a, nx=4, ny=4, nz=4)
    s.build_voronoi_neighbor()
    v, d, f, n = (np.asarray(q) for q in (s.voro_verlet_list, s.voro_distance_list, s.voro_face_area, s.voro_neighbor_number))
    assert np.all(n == 12) and np.all((v >= 0).sum(axis=1) == 12)
    assert np.allclose(d[v >= 0], a / np.sqrt(2)) and np.allclose(f[v >= 0], a * a * np.sqrt(2) / 8)
    s.cal_steinhardt_bond_orientation([6], use_voronoi=True)
    assert np.allclose(s.data["ql6"].to_numpy(), 0.574524, atol=1e-6)
    s.cal_steinhardt_bond_orientation([4, 6], use_voronoi=True, use_weight=True)
    assert np.allclose(s.data["ql4"].to_numpy(), 0.190941, atol=1e-6)


# reference: tests/test_structure_factor.py:12-52 (direct + Debye, partial + total)
@pytest.mark.parametrize("mode", ["direct", "debye"])
def test_structure_factor_against_fixture(mode, oracle_backend):
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


def test_transform_and_filter_oracle_vs_numpy():
    """src/polycrystal.cpp:20-125 restated in C against the formula of its docstring, pos_new = (pos - center) @ R.T + target,
    and the half-space test, on a rotated cube and on an empty / full selection"""
    rng = np.random.default_rng(3)
    pos = rng.random((5000, 3)) * 20.0
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    centre, target = pos.mean(0), np.array([3.0, -2.0, 7.5])
    new = (pos - centre) @ q.T + target
    planes = []
    for a in range(3):
        e = np.zeros(3); e[a] = 1.0
        planes += [np.r_[e, -(target[a] + 4.0)], np.r_[-e, target[a] - 4.0]]
    planes = np.array(planes)
    inside = (new @ planes[:, :3].T + planes[:, 3] < 0).all(1)
    out = _O.transform_and_filter(pos[:, 0].copy(), pos[:, 1].copy(), pos[:, 2].copy(), q, centre, target, planes)
    assert 0 < len(out) < len(pos) and out.shape == (int(inside.sum()), 3)
    assert np.allclose(out, new[inside], rtol=0, atol=1e-12)  # same atoms, same order
    assert len(_O.transform_and_filter(pos[:, 0].copy(), pos[:, 1].copy(), pos[:, 2].copy(), q, centre, target, np.array([[0, 0, 0, 1.0]]))) == 0
    assert len(_O.transform_and_filter(pos[:, 0].copy(), pos[:, 1].copy(), pos[:, 2].copy(), q, centre, target, np.zeros((0, 4)))) == len(pos)


def test_filter_overlap_atom_with_grain_oracle_hand_cases():
    """src/neighbor.cpp:489-672 in serial order, on pairs and chains whose outcome can be read off the rules"""
    box, org, bd = np.eye(3) * 20.0, np.zeros(3), np.ones(3, np.int32)

    def run(pos, types, grains, mm=2.0, cc=1.4, mc=3.0):
        p = np.asarray(pos, float)
        return _O.filter_overlap_atom_with_grain(p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy(), np.asarray(types, np.int32),
                                                 np.asarray(grains, np.int32), box, org, bd, mm, cc, mc).tolist()

    assert run([[1, 1, 1], [2.5, 1, 1]], [1, 1], [1, 2]) == [True, False]            # metal-metal: the higher index goes
    assert run([[1, 1, 1], [2.0, 1, 1]], [2, 2], [3, 1]) == [False, True]            # carbon-carbon, grains differ: larger grain id goes
    assert run([[1, 1, 1], [2.0, 1, 1]], [2, 2], [1, 1]) == [True, False]            # same grain: the higher index goes
    assert run([[1, 1, 1], [3.5, 1, 1]], [2, 1], [1, 1]) == [True, False]            # metal-carbon: the metal goes (here j)
    assert run([[1, 1, 1], [3.5, 1, 1]], [1, 2], [1, 1]) == [False, True]            # ... and here i
    # a chain of metals 1.5 apart: 0 removes 1; 1 no longer acts, so 2 stays; 2 removes 3
    assert run([[1, 1, 1], [2.5, 1, 1], [4.0, 1, 1], [5.5, 1, 1]], [1] * 4, [1] * 4) == [True, False, True, False]
    # across the periodic face
    assert run([[0.3, 1, 1], [19.5, 1, 1]], [1, 1], [1, 2]) == [True, False]
    # a metal that removes itself against a carbon still removes a later metal in the same turn
    assert run([[1, 1, 1], [3.0, 1, 1], [2.0, 2.0, 1]], [1, 2, 1], [1, 1, 2]) == [False, True, False]


def test_dense_labels_equals_the_reference_mapping():
    """tool.dense_labels == sorted(set(labels)) + per-atom dictionary lookup (radial_distribution_function.py:136-142)"""
    from mdapy_amd import tool_function as tool

    rng = np.random.default_rng(1)
    for raw in (np.array(["Cu", "Al", "Cu", "Ni", "Al"]), rng.integers(1, 5, 1000).astype(np.int32), np.full(50, 7, np.int64),
                np.array(["Fe"] * 9), np.array([], np.int32)):
        names, idx = tool.dense_labels(raw)
        ref_names = sorted(set(raw.tolist()))
        lut = {v: i for i, v in enumerate(ref_names)}
        assert names == ref_names and idx.dtype == np.int32
        assert np.array_equal(idx, np.array([lut[v] for v in raw.tolist()], dtype=np.int32))


def test_readers_fast_and_line_by_line_paths_agree(monkeypatch):
    """the pandas tokenizer path of load_save returns the values and dtypes of the line-by-line parser on every input file
    of the reference's test-suite"""
    import glob
    import os

    import mdapy_amd.load_save as LS

    pd = pytest.importorskip("pandas")
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "golden", "input_files", "*")))
    assert files

    def refuse(*a, **k):
        raise RuntimeError("no pandas today")

    for f in files:
        fast = LS.read_file(f)
        with monkeypatch.context() as m:
            m.setattr(pd, "read_csv", refuse)
            slow = LS.read_file(f)
        assert fast[0].columns == slow[0].columns and np.array_equal(fast[1].box, slow[1].box)
        for c in fast[0].columns:
            a, b = fast[0][c].to_numpy(), slow[0][c].to_numpy()
            if a.dtype == object or b.dtype == object:
                assert list(a) == list(b)
            else:
                assert a.dtype == b.dtype and np.array_equal(a, b)


@pytest.mark.skipif(not _O.have_voro_ref(), reason="oracle/_ref/libvoro_ref.so missing")
def test_create_polycrystal_host_logic_on_the_oracle(oracle_backend):
    """CreatePolycrystal with its device calls routed to the oracle: every atom in the Voronoi cell of its grain's seed, bulk
    density minus the boundary overlaps, reproducible from the random seed; graphene variant: carbon on the cell faces"""
    unit = mp.build_crystal("Al", "fcc", 4.05)
    kw = dict(box=48.0, seed_number=5, randomseed=12, metal_overlap_dis=2.0)
    pa = mp.CreatePolycrystal(unit, **kw)
    a = pa.compute()
    b = mp.CreatePolycrystal(unit, **kw).compute()
    assert a.N == b.N and np.array_equal(a.data["x"].to_numpy(), b.data["x"].to_numpy())
    pos = np.c_[a.data["x"].to_numpy(), a.data["y"].to_numpy(), a.data["z"].to_numpy()]
    d = pos[:, None, :] - pa.seed_position[None, :, :]
    d -= 48.0 * np.round(d / 48.0)
    assert np.array_equal(np.argmin((d ** 2).sum(-1), axis=1) + 1, a.data["grain_id"].to_numpy())
    assert abs(pa.volume.sum() - 48.0 ** 3) < 1e-6 * 48.0 ** 3 and 0.88 < a.N / (48.0 ** 3 * 4 / 4.05 ** 3) < 1.0
    assert pos.min() >= 0.0 and pos.max() < 48.0
    g = mp.CreatePolycrystal(unit, add_graphene=True, face_threshold=5.0, **kw).compute()
    ty = g.data["type"].to_numpy()
    assert set(np.unique(ty).tolist()) == {1, 2} and set(g.data["element"].to_numpy()[ty == 2].tolist()) == {"C"}
    gp = np.c_[g.data["x"].to_numpy(), g.data["y"].to_numpy(), g.data["z"].to_numpy()][ty == 2][::5]
    shifts = np.array([[i, j, k] for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)]) * 48.0
    others = (pa.seed_position[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
    dist = np.sort(np.linalg.norm(others[None, :, :] - gp[:, None, :], axis=2), axis=1)
    assert (dist[:, 1] - dist[:, 0]).max() < 1.1  # carbon within 0.5 A of a bisector plane of two seeds
    with pytest.raises(ValueError, match="Triclinic"):
        mp.CreatePolycrystal(unit, box=np.array([[48.0, 0, 0], [5.0, 48.0, 0], [0, 0, 48.0]]), seed_number=4)
    with pytest.raises(ValueError, match="Free boundary"):
        mp.CreatePolycrystal(unit, box=mp.Box(np.eye(3) * 48.0, boundary=[1, 1, 0]), seed_number=4)
