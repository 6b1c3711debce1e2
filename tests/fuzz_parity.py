#!/usr/bin/env python
"""Randomised parity sweep on the GPU box: the parity checks of tests/test_gpu_parity.py (HIP path vs the CPU
oracle / the oracle/_ref libraries) on freshly drawn systems instead of the fixed cases.

    python tests/fuzz_parity.py [seconds] [first_seed]

Every seed draws one system — orthogonal or triclinic box, random boundary flags, origin, density, lattice or
gas, optionally out-of-box ("unwrapped") atoms — and runs the checks that support that kind of input.  A failure
prints the seed and the check; the exit code is the number of failures.  Test infrastructure, like tests/.
"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import test_gpu_parity as T
from mdapy_amd.build_lattice import lattice_positions


def draw(seed):
    rng = np.random.default_rng(seed)
    sigma = -1.0
    kind = rng.choice(["gas", "fcc", "bcc", "hcp", "blob", "tiny", "big"])
    tri = rng.random() < 0.4
    bnd = np.array(rng.random(3) < 0.75, np.int32)
    origin = rng.normal(0, 5.0, 3) if rng.random() < 0.5 else np.zeros(3)
    if kind == "big":  # 10-16 cells per axis: large enough for the tile kernels (orthogonal and sheared, periodic and open)
        n = [int(rng.integers(10, 17)) for _ in range(3)]
        pos, box = lattice_positions("fcc", 3.615, *n)
        box = np.asarray(box, float)
        sigma = float(rng.choice([0.03, 0.15, 0.4]))
        pos = pos + rng.normal(0, sigma, pos.shape)
        if rng.random() < 0.6:
            bnd = np.array([1, 1, 1], np.int32)
        if tri:
            sh = np.eye(3)
            sh[1, 0], sh[2, 0], sh[2, 1] = rng.uniform(-0.15, 0.15, 3)
            pos, box = pos @ sh, box @ sh
    elif kind in ("fcc", "bcc", "hcp"):
        a = {"fcc": 3.615, "bcc": 2.87, "hcp": 2.95}[kind]
        n = [int(rng.integers(4, 9)) for _ in range(3)]
        pos, box = lattice_positions(kind, a, *n)
        box = np.asarray(box, float)
        sigma = float(rng.choice([0.0, 0.03, 0.15]))
        pos = pos + rng.normal(0, sigma, pos.shape)
        if tri:  # shear the whole crystal (still a periodic crystal of the sheared box)
            sh = np.eye(3)
            sh[1, 0], sh[2, 0], sh[2, 1] = rng.uniform(-0.3, 0.3, 3)
            pos, box = pos @ sh, box @ sh
    elif kind == "tiny":  # a few atoms in a box of a few cutoffs: replication policies, rows with periodic twins
        L = rng.uniform(5.0, 11.0, 3)
        box = np.diag(L)
        if tri:
            box[1, 0], box[2, 0], box[2, 1] = rng.uniform(-0.3, 0.3, 3) * L[0]
        pos = rng.random((int(rng.integers(3, 60)), 3)) @ box
    else:
        L = rng.uniform(12.0, 34.0, 3)
        box = np.diag(L)
        if tri:
            box[1, 0], box[2, 0], box[2, 1] = rng.uniform(-0.35, 0.35, 3) * L[0]
        rho = rng.uniform(0.01, 0.07)
        N = int(min(max(rho * abs(np.linalg.det(box)), 150), 5000))
        frac = rng.random((N, 3))
        if kind == "blob":
            frac[: N // 2] = 0.3 + 0.25 * frac[: N // 2]
        pos = frac @ box
    pos = pos + origin
    if tri and rng.random() < 0.4:  # general orientation: upper-triangular entries, cell vectors off the axes
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        q *= np.sign(np.linalg.det(q))
        pos, box, origin = pos @ q, box @ q, origin @ q
    unwrapped = rng.random() < 0.2
    if unwrapped:
        far = int(rng.choice([2, 2, 13, 20]))  # (up to 14 box lengths the tile kernel keeps the call, beyond the thread-per-atom one takes it)
        pos = pos + (rng.integers(-far, far + 1, pos.shape) * bnd) @ box
    return dict(seed=seed, sigma=sigma, kind=kind, tri=tri, unwrapped=unwrapped, pos=pos, box=box, origin=origin, bnd=bnd)


def rdf_stream_check(s):
    """streaming partial RDF (tile kernel for orthogonal boxes, thread-per-atom otherwise) against the oracle: integer counts"""
    r = np.random.default_rng(s["seed"] + 23)
    pos = s["pos"]
    nt = int(r.integers(1, 4))
    ty = r.integers(0, nt, len(pos)).astype(np.int32)
    thick = np.abs(np.linalg.det(s["box"])) / np.array([np.linalg.norm(np.cross(s["box"][(d + 1) % 3], s["box"][(d + 2) % 3])) for d in range(3)])
    rc = float(r.uniform(2.5, max(3.0, min(9.0, thick.min() * 0.45))))
    nbin = int(r.integers(10, 400))
    x, y, z = T._xyz(pos)
    g0, g1 = np.zeros((nt, nt, nbin)), np.zeros((nt, nt, nbin))
    T.O._rdf_streaming(x, y, z, ty, s["box"], s["origin"], s["bnd"], g0, rc, nbin, 8)
    T._rdf._rdf_streaming(x, y, z, ty, s["box"], s["origin"], s["bnd"], g1, rc, nbin, 1)
    assert np.array_equal(g0, g1)


def fused_check(s):
    """mdh_build_neighbor_fcna against mdh_build_neighbor followed by mdh_fcna: lists and labels bit for bit"""
    r = np.random.default_rng(s["seed"] + 31)
    x, y, z = T._xyz(s["pos"])
    n = len(x)
    rc = float(r.uniform(2.7, 3.9))
    M = int(r.choice([12, 14, 16, 20, 30]))
    va = np.full((n, M), -1, np.int32); da = np.full((n, M), rc + 1.0); na = np.zeros(n, np.int32); pa = np.zeros(n, np.int32)
    T._neighbor.build_neighbor(x, y, z, s["box"], s["origin"], s["bnd"], rc, va, da, na, 1)
    T._cna.fcna(x, y, z, s["box"], s["origin"], s["bnd"], va, na, pa, rc, 1)
    vb = np.empty((n, M), np.int32); db = np.empty((n, M)); nb = np.empty(n, np.int32); pb = np.zeros(n, np.int32)
    T._neighbor.build_neighbor_fcna(x, y, z, s["box"], s["origin"], s["bnd"], rc, vb, db, nb, pb, 1, fill_pads=True)
    assert np.array_equal(nb, na) and np.array_equal(vb, va) and np.array_equal(db, da) and np.array_equal(pb, pa)


def wide_rows_check(s):
    """rows of 65 ... 128 slots (rc 5.4 ... 6.1 A on the big crystals): exact-width and fixed-width rows bit for bit vs the oracle"""
    r = np.random.default_rng(s["seed"] + 41)
    x, y, z = T._xyz(s["pos"])
    n = len(x)
    rc = float(r.uniform(5.4, 6.1))
    v2, d2, n2 = T._neighbor.build_neighbor_without_max_neigh(x, y, z, s["box"], s["origin"], s["bnd"], rc, 1)
    vo, do, no = T.O.build_neighbor_without_max_neigh(x, y, z, s["box"], s["origin"], s["bnd"], rc, 8)
    assert np.array_equal(n2, no) and np.array_equal(v2, vo) and np.array_equal(d2, do)
    M = int(r.choice([66, 80, 97, 128]))
    va = np.full((n, M), -1, np.int32); da = np.full((n, M), rc + 1.0); na = np.zeros(n, np.int32)
    T.O.build_neighbor(x, y, z, s["box"], s["origin"], s["bnd"], rc, va, da, na, 8)
    vb = np.empty((n, M), np.int32); db = np.empty((n, M)); nb = np.empty(n, np.int32)
    T._neighbor.build_neighbor(x, y, z, s["box"], s["origin"], s["bnd"], rc, vb, db, nb, 1, fill_pads=True)
    assert np.array_equal(nb, na) and np.array_equal(vb, va) and np.array_equal(db, da)


def checks(s):
    case = ("fuzz", s["pos"], s["box"], s["origin"], s["bnd"])
    if s["kind"] == "big":  # the tile kernels: neighbour rows bit for bit (fixed and exact width), pair counts
        rc_big = float(np.random.default_rng(s["seed"] + 7).uniform(2.8, 3.7))
        T._cases = lambda: [(n, ) + case[1:] for n in NAMES]
        return [("neighbor", lambda: T.test_neighbor_bit_exact_vs_oracle(case, rc_big)), ("rdf_stream", lambda: rdf_stream_check(s)),
                ("fused_cna", lambda: fused_check(s))] + ([("wide_rows", lambda: wide_rows_check(s))] if s["seed"] % 3 == 0 else [])
    T._cases = lambda: [(n, ) + case[1:] for n in NAMES]
    rc = float(np.random.default_rng(s["seed"] + 7).uniform(2.6, 4.6))
    out = [("neighbor", lambda: T.test_neighbor_bit_exact_vs_oracle(case, rc)),
           ("rdf_stream", lambda: rdf_stream_check(s)),
           ("fused_cna", lambda: fused_check(s)),
           ("sort_cna", lambda: T.test_sort_and_cna_vs_oracle(case)),
           ("overlap", lambda: T.test_filter_overlap_atom_vs_oracle("fuzz"))]
    if len(s["pos"]) >= 300 and s["kind"] in ("fcc", "bcc", "blob"):  # (the check also wants some atoms removed by its last cutoff set)
        out += [("overlap_grain", lambda: T.test_filter_overlap_atom_with_grain_vs_oracle("fuzz"))]
    if not s["unwrapped"] and (len(s["pos"]) >= 20 or all(s["bnd"])):  # open box with fewer atoms than neighbours asked for: the reference indexes x[-1]
        out += [("knn", lambda: T.test_knn_general_vs_oracle(case)),
                ("steinhardt_rc", lambda: T.test_steinhardt_vs_oracle(case, "rc")),
                ("steinhardt_nnn", lambda: T.test_steinhardt_vs_oracle(case, "nnn")),
                ("aja_cnp_entropy", lambda: T.test_aja_cnp_entropy_vs_oracle("fuzz")),
                ]
        if len(s["pos"]) >= 100:  # that check also asserts that its type filter removes something
            out += [("temp_cluster", lambda: T.test_atomic_temperature_and_cluster_vs_oracle("fuzz"))]
        if T.O.have_ref() and s["sigma"] != 0.0:  # perfect lattices: exact ties / degenerate hulls decide by rounding noise
            r2 = np.random.default_rng(s["seed"] + 13)
            structure = str(r2.choice(["default", "all", "fcc-hcp-bcc-ico-sc", "fcc-hcp-bcc", "dcub-dhex", "bcc,sc", "graphene-fcc", "ico"]))
            types = r2.integers(1, 4, len(s["pos"])).astype(np.int32) if r2.random() < 0.4 else None
            pc = ("fuzz", s["pos"] - s["origin"], s["box"], tuple(int(v) for v in s["bnd"]), structure, types, float(r2.choice([0.0, 0.05, 0.1, 0.3])))
            out += [("ptm", lambda: T.test_ptm_vs_reference_library(pc))]
        ortho = not np.any(s["box"] - np.diag(np.diag(s["box"])))  # the orthogonal entry points (hcp cells are hexagonal)
        if T.O.have_voro_ref() and ortho and s["kind"] != "blob" and all(s["bnd"]):
            out += [("voronoi", lambda: T.test_voronoi_vs_reference_library("fuzz")),
                    ("voronoi_nb", lambda: T.test_voronoi_neighbors_vs_reference_library("fuzz"))]
        if T.O.have_voro_ref() and ortho and s["kind"] == "gas":  # (a 400-atom corner of a crystal in its full periodic box is the
            # cluster-in-vacuum case the Voronoi driver refuses)
            out += [("cell_info", lambda: T.test_voronoi_cell_info_vs_reference_library("random_gas"))]
    return out


NAMES = ["fuzz", "random_gas"]  # "random_gas": the name under which the cell-info check takes a 400-atom subset


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    fails, ran = [], 0
    while time.time() - t0 < budget:
        s = draw(seed)
        only = os.environ.get("FUZZ_ONLY")
        for name, fn in checks(s):
            if only and not name.startswith(only):
                continue
            if os.environ.get("FUZZ_TRACE"):
                print("seed", seed, name, flush=True)
            try:
                fn()
                ran += 1
            except AssertionError:
                tb = traceback.extract_tb(sys.exc_info()[2])[-1]
                fails.append((seed, name, f"assert at {os.path.basename(tb.filename)}:{tb.lineno}"))
            except Exception as e:  # loud refusals are findings too: list them
                fails.append((seed, name, f"{type(e).__name__}: {str(e)[:120]}"))
        seed += 1
    print(f"fuzz: {ran} checks passed over seeds up to {seed - 1}; {len(fails)} failures", flush=True)
    by = {}
    for f in fails:
        by[f[1]] = by.get(f[1], 0) + 1
    print("  by check:", by)
    for f in fails[:60]:
        s = draw(f[0])
        print("  FAIL seed=%d check=%s %s  [kind=%s tri=%s unwrapped=%s bnd=%s N=%d]" % (f + (s["kind"], s["tri"], s["unwrapped"], s["bnd"].tolist(), len(s["pos"]))))
    return len(fails)


if __name__ == "__main__":
    sys.exit(min(main(), 100))
