"""Host-side policy layer (mdapy_amd/policy.py, scattering.py, structure_factor.py weighting) — no GPU needed."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mdapy_amd as mp
from mdapy_amd import policy, scattering
from mdapy_amd.structure_factor import StructureFactor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_axis_copies_follow_the_replication_rules():
    box = mp.Box(np.diag([10.0, 40.0, 7.0]), boundary=[1, 1, 0])
    assert policy.axis_copies(box, 15.0).tolist() == [2, 1, 1]          # thin periodic x; open z is never replicated
    assert policy.axis_copies(box, 2 * 5.5).tolist() == [2, 1, 1]       # 2 rc = 11 > 10
    assert policy.axis_copies(box, 10.0).tolist() == [1, 1, 1]          # exactly thick enough: no copy
    assert box.check_small_box(5.5).tolist() == [2, 1, 1] and box.check_small_box(5.5).dtype == np.int32
    tri = mp.Box([[10.0, 0, 0], [5.0, 8.0, 0], [0, 0, 30.0]])
    t = tri.get_thickness()
    assert np.allclose(t, [8.0 * 10 * 30 / np.linalg.norm(np.cross(tri.box[1], tri.box[2])), 8.0, 30.0])
    assert policy.axis_copies(tri, 15.0).tolist() == [int(np.ceil(15 / t[0])), 2, 1]
    assert policy.hopeless(mp.Box(5.0, boundary=[0, 0, 0]), 14, 14) and not policy.hopeless(mp.Box(5.0), 3, 14)


def test_label_codes_and_shell_table():
    names, codes = policy.label_codes(np.array(["Ni", "Al", "Ni", "Cr"], dtype=object))
    assert names == ["Al", "Cr", "Ni"] and codes.tolist() == [2, 0, 2, 1] and codes.dtype == np.int32
    names, codes = policy.label_codes(np.array([3, 3, 3]))
    assert names == [3] and codes.tolist() == [0, 0, 0]
    assert policy.label_codes(np.array([]))[0] == []
    r, shell = policy.shell_table(4.0, 8, 100.0)
    assert np.allclose(r, np.arange(8) * 0.5 + 0.25)
    assert np.isclose(shell.sum() * 100.0, 4.0 / 3.0 * np.pi * 4.0 ** 3)  # the shells fill the sphere


def test_cromer_mann_fits_reproduce_the_atomic_number():
    """f(0) = c + sum a_i of a Cromer-Mann fit equals Z to a few hundredths of an electron: catches a mistyped
    coefficient of the table in mdapy_amd/scattering.py"""
    for name, (z, a, b, c) in scattering.CROMER_MANN.items():
        f0 = float(scattering.xray(name, 0.0))
        assert abs(f0 - z) < 0.06, (name, f0)
        assert all(bi > 0 for bi in b) and len(a) == len(b) == 4
        # monotone decay towards c (+ the short-range Gaussians) and the Mott-Bethe limit stays finite
        k = np.linspace(0.5, 20.0, 40)
        f = scattering.xray(name, k)
        assert np.all(np.diff(f) < 1e-9), name
        assert np.all(np.isfinite(scattering.electron(name, k)))
    assert set(scattering.NEUTRON_LENGTH_FM) == set(scattering.CROMER_MANN)
    with pytest.raises(KeyError, match="Xx"):
        scattering.xray("Xx", 1.0)


def test_weighted_structure_factor_totals():
    """S_w = sum (2 - delta) c_a c_b f_a f_b A_ab / (sum c f)^2 on hand-made partials"""
    sf = StructureFactor.__new__(StructureFactor)
    sf.k = np.linspace(0.5, 8.0, 16)
    A = {("Cu", "Cu"): 1.0 + 0.3 * np.sin(sf.k), ("Cu", "Zr"): 1.0 - 0.2 * np.cos(sf.k), ("Zr", "Zr"): 1.0 + 0.1 * sf.k / 8.0}
    sf.Sk_partial, sf._uniele, sf._concentrations = A, ["Cu", "Zr"], np.array([0.64, 0.36])
    for kind, fn in (("xray", scattering.xray), ("neutron", scattering.neutron), ("electron", scattering.electron)):
        fc, fz = fn("Cu", sf.k), fn("Zr", sf.k)
        want = (0.64 ** 2 * fc * fc * A[("Cu", "Cu")] + 2 * 0.64 * 0.36 * fc * fz * A[("Cu", "Zr")] + 0.36 ** 2 * fz * fz * A[("Zr", "Zr")]) \
            / (0.64 * fc + 0.36 * fz) ** 2
        assert np.allclose(sf._weighted_total(kind), want, rtol=1e-13)
    # one species: every weighting returns the partial itself
    sf.Sk_partial, sf._uniele, sf._concentrations = {("Cu", "Cu"): A[("Cu", "Cu")]}, ["Cu"], np.array([1.0])
    assert np.allclose(sf.get_xray_structure_factor(), A[("Cu", "Cu")]) and np.allclose(sf.get_neutron_structure_factor(), A[("Cu", "Cu")])
    # an absorbing isotope mixture gives a complex length: the modulus is returned
    sf.Sk_partial, sf._uniele = {("Cd", "Cd"): A[("Cu", "Cu")]}, ["Cd"]
    assert np.isrealobj(sf.get_neutron_structure_factor())
    sf.Sk_partial = None
    with pytest.raises(RuntimeError, match="cal_partial=True"):
        sf._weighted_total("xray")


def test_structure_factor_with_form_factors_system_flow(oracle_backend):
    """atomic_form_factors=True promotes cal_partial and fills Sk_xray (debye and direct routes)"""
    rng = np.random.default_rng(3)
    pos = rng.random((600, 3)) * 18.0
    ele = np.where(rng.random(600) < 0.6, "Cu", "Zr").astype(object)
    s = mp.System(data=mp.Frame({"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2], "element": ele}), box=mp.Box(18.0))
    for mode in ("debye", "direct"):
        sf = s.cal_structure_factor(1.0, 6.0, 12, atomic_form_factors=True, mode=mode)
        assert sf.cal_partial and set(sf.Sk_partial) == {("Cu", "Cu"), ("Cu", "Zr"), ("Zr", "Zr")}
        assert sf.Sk_xray.shape == (12,) and np.all(np.isfinite(sf.Sk_xray))
        assert np.allclose(sf.Sk_xray, sf._weighted_total("xray"))
        assert sf.get_neutron_structure_factor().shape == (12,) and sf.get_electron_structure_factor().shape == (12,)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/mdapy"), reason="needs the reference tree (build container only)")
def test_host_layer_is_not_a_transcription_of_the_reference():
    """token overlap (runs of >= 8 tokens, comments and docstrings stripped) of every mdapy_amd/*.py with the same-named
    reference file stays below 0.35 (tools/overlap_check.py)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "overlap_check.py"), "0.35"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout


def test_identity_caches_serve_only_arrays_the_package_froze():
    """devarray.frozen_by_us / policy.label_codes: a caller's own read-only array is never cached by identity (it may be
    thawed and rewritten); a column the package froze is, and a thaw-rewrite-freeze of even that one is noticed."""
    import numpy as np
    from mdapy_amd import devarray, policy
    from mdapy_amd.frame import Frame

    mine = np.arange(300000, dtype=np.int32) % 3
    mine.setflags(write=False)
    assert not devarray.frozen_by_us(mine)
    names, codes = policy.label_codes(mine)
    mine.setflags(write=True); mine[:] = 7; mine.setflags(write=False)
    assert policy.label_codes(mine)[0] == [7]  # recomputed, not served from a cache keyed by the array's identity

    col = Frame({"type": np.arange(300000, dtype=np.int32) % 3})["type"].to_numpy()
    assert devarray.frozen_by_us(col)
    n1, c1 = policy.label_codes(col)
    n2, c2 = policy.label_codes(col)
    assert c2 is c1 and n1 == [0, 1, 2]  # the cached codes
    col.setflags(write=True); col[:] = 5; col.setflags(write=False)  # (reaching into the frame's own array)
    assert not devarray.frozen_by_us(col)
    assert policy.label_codes(col)[0] == [5]


def test_fixed_cutoff_cna_is_lent_a_remembered_list_of_the_same_cutoff(oracle_backend, monkeypatch):
    """cal_common_neighbor_analysis(rc) right after build_neighbor(rc): the reference lends nothing and the analysis builds a
    list of its own with this very cutoff (src/mdapy/system.py:2038-2058) — the same rows; here the remembered one is lent
    (no second build), a list with a LARGER cutoff is not (its rows hold more neighbours), and the labels are those of a
    fresh System either way"""
    from mdapy_amd import system as S
    from mdapy_amd.build_lattice import lattice_positions

    pos, box = lattice_positions("fcc", 3.615, 6, 6, 6)
    pos = pos + np.random.default_rng(5).normal(0, 0.05, pos.shape)
    rc = 0.854 * 3.615
    fresh = mp.System(pos=pos, box=mp.Box(box))
    fresh.cal_common_neighbor_analysis(rc)
    want = fresh.data["cna"].to_numpy()
    lent = []
    real = S.CommonNeighborAnalysis

    def spy(frame, cell, rows, counts, cutoff):
        lent.append(rows is not None)
        return real(frame, cell, rows, counts, cutoff)

    monkeypatch.setattr(S, "CommonNeighborAnalysis", spy)
    s = mp.System(pos=pos, box=mp.Box(box))
    s.build_neighbor(rc, max_neigh=20)
    s.cal_common_neighbor_analysis(rc)
    assert lent == [True] and np.array_equal(s.data["cna"].to_numpy(), want)
    s.build_neighbor(rc + 0.5, max_neigh=40)
    s.cal_common_neighbor_analysis(rc)
    assert lent == [True, False] and np.array_equal(s.data["cna"].to_numpy(), want)


def test_labels_made_in_the_pass_that_builds_the_list(oracle_backend, monkeypatch):
    """a System without a list: cal_common_neighbor_analysis(rc[, max_neigh]) builds the list (it stays the system's, with its
    provenance) and takes the labels from the SAME pass (Neighbor.compute(label=True) -> build_neighbor_fcna / the exact-width entry
    with `pattern`) — no CommonNeighborAnalysis job, labels those of the class on explicit lists; the class without a list does the
    same; a thin box (searched on a replica) falls back to the class"""
    from mdapy_amd import kernels
    from mdapy_amd import system as S
    from mdapy_amd.build_lattice import lattice_positions
    from mdapy_amd.common_neighbor_analysis import CommonNeighborAnalysis
    from mdapy_amd.neighbor import Neighbor

    pos, box = lattice_positions("fcc", 3.615, 6, 6, 6)
    pos = pos + np.random.default_rng(7).normal(0, 0.06, pos.shape)
    rc = 0.854 * 3.615
    calls = {"fixed": 0, "exact": 0, "job": 0}
    real_fixed, real_exact = kernels.neighbor.build_neighbor_fcna, kernels.neighbor.build_neighbor_without_max_neigh

    def fixed(*a, **k):
        calls["fixed"] += 1
        return real_fixed(*a, **k)

    def exact(*a, **k):
        calls["exact"] += "pattern" in k
        return real_exact(*a, **k)

    monkeypatch.setattr(kernels.neighbor, "build_neighbor_fcna", fixed)
    monkeypatch.setattr(kernels.neighbor, "build_neighbor_without_max_neigh", exact)
    real_job = S.CommonNeighborAnalysis

    def job(*a):
        calls["job"] += 1
        return real_job(*a)

    monkeypatch.setattr(S, "CommonNeighborAnalysis", job)
    # the reference's two steps, by hand
    ref = Neighbor(rc, mp.Box(box), mp.System(pos=pos, box=mp.Box(box)).data)
    ref.compute()
    by_hand = CommonNeighborAnalysis(mp.System(pos=pos, box=mp.Box(box)).data, mp.Box(box), ref.verlet_list, ref.neighbor_number, rc)
    by_hand.compute()
    want = np.asarray(by_hand.pattern)
    assert (want == 1).mean() > 0.9
    for max_neigh, key in ((None, "exact"), (20, "fixed")):
        s = mp.System(pos=pos, box=mp.Box(box))
        s.cal_common_neighbor_analysis(rc, max_neigh)
        assert calls[key] == 1 and calls["job"] == 0, (max_neigh, calls)
        assert np.array_equal(s.data["cna"].to_numpy(), want)
        assert s.rc == rc and s._list_cutoff == float(rc) and np.array_equal(np.asarray(s.neighbor_number), np.asarray(ref.neighbor_number))
        width = 20 if max_neigh else int(np.asarray(ref.verlet_list).shape[1])
        assert np.asarray(s.verlet_list).shape == (len(pos), width)
        s.cal_common_neighbor_analysis(rc, max_neigh)  # now the remembered list is lent to the class
        assert calls["job"] == 1 and np.array_equal(s.data["cna"].to_numpy(), want)
        calls.update(fixed=0, exact=0, job=0)
    # the class on its own (no list handed in)
    alone = CommonNeighborAnalysis(mp.System(pos=pos, box=mp.Box(box)).data, mp.Box(box), rc=rc)
    alone.compute()
    assert calls["exact"] == 1 and np.array_equal(np.asarray(alone.pattern), want)
    # Neighbor.compute(label=True) on a box thinner than two cutoffs: searched on a replica, no labels offered
    thin_pos, thin_box = lattice_positions("fcc", 3.615, 6, 6, 1)
    thin = Neighbor(rc, mp.Box(thin_box), mp.System(pos=thin_pos, box=mp.Box(thin_box)).data)
    thin.compute(label=True)
    assert thin.pattern is None and hasattr(thin, "_enlarge_data")


def test_stale_rc_beside_a_k_nearest_list_lends_nothing(oracle_backend, monkeypatch):
    """build_neighbor(rc), build_nearest_neighbor(8), cal_common_neighbor_analysis(rc): `rc` survives the k-nearest search (as in
    the reference, src/mdapy/system.py:1256-1263) but the current list is a k-nearest list — the analysis must build its own
    (the remembered rows were lent and every fcc atom came out 0)."""
    from mdapy_amd import system as S
    from mdapy_amd.build_lattice import lattice_positions

    pos, box = lattice_positions("fcc", 3.615, 6, 6, 6)
    rc = 0.854 * 3.615
    lent = []
    real = S.CommonNeighborAnalysis

    def spy(frame, cell, rows, counts, cutoff):
        lent.append(rows is not None)
        return real(frame, cell, rows, counts, cutoff)

    monkeypatch.setattr(S, "CommonNeighborAnalysis", spy)
    s = mp.System(pos=pos, box=mp.Box(box))
    s.build_neighbor(rc)
    s.build_nearest_neighbor(8)
    assert "rc" in s.__dict__ and "_list_cutoff" not in s.__dict__
    s.cal_common_neighbor_analysis(rc)
    assert lent == [False]
    assert (s.data["cna"].to_numpy() == 1).all()  # 864 fcc atoms, as a fresh System says
    # a box assignment forgets the provenance with the list
    s.build_neighbor(rc)
    assert s._list_cutoff == rc
    s.box = mp.Box(box)
    assert "_list_cutoff" not in s.__dict__ and "rc" not in s.__dict__


class _FakeAtoms:
    """the four getters of ase.Atoms that BuildSystem.from_ase calls (src/mdapy/load_save.py:537-541)"""

    def __init__(self, cell, pbc, pos, symbols):
        self._c, self._p, self._x, self._s = cell, pbc, pos, symbols

    def get_cell(self): return self._c
    def get_pbc(self): return self._p
    def get_positions(self): return self._x
    def get_chemical_symbols(self): return self._s


class _FakeCell:
    def __init__(self, m34, pbc):
        self._m, self.pbc = m34, pbc

    def __getitem__(self, key): return self._m[key]


class _FakeParticles(dict):
    particle_type = None


def test_system_from_ase_and_ovito_like_objects():
    """System(ase_atom=...) / System(ovito_atom=...) (src/mdapy/system.py:181-203), duck-typed: neither package is installed
    here, and the reference only uses the getters / mappings the fakes provide.  Positional order is the reference's:
    filename, data, pos, box, ase_atom, ovito_atom, format, global_info."""
    import inspect

    assert list(inspect.signature(mp.System.__init__).parameters)[1:] == [
        "filename", "data", "pos", "box", "ase_atom", "ovito_atom", "format", "global_info"]
    rng = np.random.default_rng(3)
    cell = np.array([[10.0, 0, 0], [2.0, 9.0, 0], [0, 0, 8.0]])
    pos = rng.random((7, 3)) * 8
    s = mp.System(ase_atom=_FakeAtoms(cell, [True, False, True], pos, ["Cu"] * 4 + ["Zr"] * 3))
    assert s.N == 7 and np.array_equal(s.box.box, cell) and s.box.boundary.tolist() == [1, 0, 1]
    assert np.array_equal(np.column_stack([s.data[c].to_numpy() for c in "xyz"]), pos)
    assert s.data["element"].to_numpy().tolist() == ["Cu"] * 4 + ["Zr"] * 3
    with pytest.raises(TypeError, match="ASE Atoms"):
        mp.System(ase_atom=object())

    m34 = np.zeros((3, 4)); m34[:, :3] = cell.T; m34[:, 3] = [1.0, 2.0, 3.0]  # OVITO: cell vectors as columns + origin column
    class T:  # particle types with names
        def __init__(self, i, n): self.id, self.name = i, n
    class Table:
        types = [T(1, "Cu"), T(2, "Zr")]
    parts = _FakeParticles({"Position": pos, "Particle Type": np.array([1, 1, 2, 2, 1, 2, 1]), "Particle Identifier": np.arange(1, 8),
                            "Velocity": pos * 0.1, "Velocity Magnitude": np.ones(7), "Potential Energy": np.arange(7.0),
                            "Stress Tensor": np.ones((7, 6))})
    parts.particle_type = Table()

    class Coll:
        pass
    c = Coll(); c.cell = _FakeCell(m34, (True, True, False)); c.particles = parts; c.attributes = {"Timestep": 100}
    s = mp.System(ovito_atom=c)
    assert s.N == 7 and np.array_equal(s.box.box, cell) and s.box.boundary.tolist() == [1, 1, 0]
    assert s.box.origin.tolist() == [1.0, 2.0, 3.0]  # OVITO's origin column is part of the box (load_save.py:444)
    assert s.global_info == {"Timestep": 100}
    cols = set(s.data.columns)
    assert {"x", "y", "z", "type", "id", "vx", "vy", "vz", "PotentialEnergy", "StressTensor_0", "StressTensor_5", "element"} <= cols
    assert "VelocityMagnitude" not in cols and "Velocity Magnitude" not in cols
    assert s.data["element"].to_numpy().tolist() == ["Cu", "Cu", "Zr", "Zr", "Cu", "Zr", "Cu"]
    with pytest.raises(RuntimeError, match="ase_atom or ovito_atom"):
        mp.System()


def test_update_data_accepts_the_misspelled_alias_with_a_warning():
    pos = np.random.default_rng(0).random((5, 3)) * 4
    s = mp.System(pos=pos, box=mp.Box(5.0))
    with pytest.warns(DeprecationWarning, match="reset_calcolator"):
        s.update_data(s.data.with_columns(e=np.arange(5.0)), reset_calcolator=True)
    assert "e" in s.data.columns
    import inspect
    from mdapy_amd.create_polycrystal import CreatePolycrystal

    assert list(inspect.signature(mp.System.update_data).parameters)[1:] == ["data", "reset_calculator", "reset_neighbor", "reset_calcolator"]
    assert inspect.signature(CreatePolycrystal.compute).parameters["verbose"].default is True


def test_mdapy_hip_device_is_validated_at_load():
    """MDAPY_HIP_DEVICE: a non-integer is refused when the library loads; on a CPU-only box an index is accepted
    (there is nothing to select) and compute calls still raise on their own"""
    code = "import mdapy_amd._lib as L; L.lib(); print('ok')"
    env = dict(os.environ, MDAPY_HIP_DEVICE="zero")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True)
    assert out.returncode != 0 and "MDAPY_HIP_DEVICE must be a device index" in out.stderr
    env["MDAPY_HIP_DEVICE"] = "0"
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_system_small_helpers_follow_the_reference(oracle_backend):
    """set_element / set_type_by_element / get_positions / get_velocities / update_box / delete_overlap and the positional order
    of tool_function.average_by_neighbor and build_lattice.build_crystal (src/mdapy/system.py:333-500, 786-846, 1414-1490;
    tool_function.py:14-23; build_lattice.py:657-668)"""
    import inspect

    from mdapy_amd import build_lattice, tool_function

    assert list(inspect.signature(tool_function.average_by_neighbor).parameters)[:6] == [
        "average_rc", "data", "property_name", "verlet_list", "distance_list", "neighbor_number"]
    assert list(inspect.signature(build_lattice.build_crystal).parameters) == [
        "name", "structure", "a", "miller1", "miller2", "miller3", "nx", "ny", "nz", "c"]
    rng = np.random.default_rng(3)
    pos = rng.random((400, 3)) * 12.0
    s = mp.System(pos=pos, box=np.diag([12.0, 12.0, 12.0]))
    s.set_element("Cu")
    assert set(s.data["element"].to_numpy()) == {"Cu"}
    names = np.where(np.arange(400) % 3 == 0, "Zr", "Cu")
    s.set_element(list(names))
    with pytest.raises(AssertionError, match="must equal the atom number"):
        s.set_element(["Cu"] * 3)
    s.set_type_by_element(["Zr", "Cu"])
    assert s.data["type"].to_numpy().tolist() == np.where(names == "Zr", 1, 2).tolist() and s.data["type"].to_numpy().dtype == np.int32
    with pytest.raises(AssertionError, match="element_list must include element"):
        s.set_type_by_element(["Zr"])
    assert np.array_equal(s.get_positions().to_numpy(), pos)
    assert np.allclose(s.get_positions(reduced=True).to_numpy(), pos / 12.0) and s.get_positions(reduced=True).columns == ["r_x", "r_y", "r_z"]
    with pytest.raises(AssertionError):
        s.get_velocities()
    s.build_neighbor(3.0)
    s.update_box(np.diag([13.2, 12.0, 12.0]), scale_pos=True)
    assert not hasattr(s, "verlet_list")  # the box setter forgot the list
    assert np.allclose(s.data["x"].to_numpy(), pos[:, 0] * 1.1) and np.array_equal(s.data["y"].to_numpy(), pos[:, 1])
    with pytest.raises(AssertionError, match="all periodic"):
        s.update_box(mp.Box(np.diag([13.2, 12.0, 12.0]), boundary=[1, 1, 0]), scale_pos=True)
    # delete_overlap against the reference's sweep in index order, on a gas with chains of close atoms
    gas = rng.random((600, 3)) * 14.0
    gas[100:160] = gas[40:100] + rng.normal(0, 0.25, (60, 3))
    gas[160:200] = gas[100:140] + rng.normal(0, 0.25, (40, 3))
    t = mp.System(pos=gas, box=np.diag([14.0] * 3))
    rc = 0.6
    t.build_neighbor(rc)
    v, d, nn = (np.asarray(a) for a in (t.verlet_list, t.distance_list, t.neighbor_number))
    gone = np.zeros(600, bool)
    for j in range(600):
        near = v[j, :nn[j]][(v[j, :nn[j]] < j) & (d[j, :nn[j]] < rc)]
        if near.size and not gone[near].all():
            gone[j] = True
    assert gone.sum() > 40
    assert t.delete_overlap(rc) == int(gone.sum())
    assert t.N == 600 - int(gone.sum()) and np.array_equal(t.data.to_numpy()[:, :3], gas[~gone]) and not hasattr(t, "verlet_list")
    assert t.delete_overlap(rc) == 0


def test_structure_factor_back_to_real_space():
    """the nine (r, g / G / R) accessors (src/mdapy/structure_factor.py:560-653) on an exactly known pair:
    g(r) = 1 + A exp(-r^2 / 2 s^2)  <->  S(k) - 1 = rho A (2 pi s^2)^(3/2) exp(-k^2 s^2 / 2)"""
    sf = StructureFactor.__new__(StructureFactor)
    sf.k = np.linspace(0.05, 25.0, 500)
    rho, amp, s = 0.05, -1.0, 0.8
    sf._density = rho
    total = 1 + rho * amp * (2 * np.pi * s * s) ** 1.5 * np.exp(-sf.k ** 2 * s * s / 2)
    for kind in ("xray", "neutron", "electron"):
        setattr(sf, f"get_{kind}_structure_factor", lambda total=total: total)
        grid = np.linspace(0, 6, 61)
        r, g = getattr(sf, f"get_{kind}_pair_distribution_function")(grid)
        assert g[0] == 0.0 and np.abs(g[1:] - (1 + amp * np.exp(-r[1:] ** 2 / (2 * s * s)))).max() < 1e-4
        _, red = getattr(sf, f"get_{kind}_reduced_pair_distribution_function")(grid)
        _, rad = getattr(sf, f"get_{kind}_radial_distribution_function")(grid)
        assert np.allclose(red[1:], 4 * np.pi * r[1:] * rho * (g[1:] - 1)) and np.allclose(rad, 4 * np.pi * r ** 2 * rho * g)
    r, _ = sf.get_xray_pair_distribution_function()  # default grid: 200 points out to pi / dk
    assert len(r) == 200 and np.isclose(r[-1], np.pi / (sf.k[1] - sf.k[0]))
    with pytest.raises(ValueError, match="unknown weighting kind"):
        sf._real_space("muon")
