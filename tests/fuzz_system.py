#!/usr/bin/env python
"""Randomised System-level parity sweep on the GPU box: the same random call sequence on a random system, once
through the HIP library and once with the host classes routed to the CPU oracle (tests/_oracle_backend.py), then
every per-atom column and every returned curve compared.

    python tests/fuzz_system.py [seconds] [first_seed]

FUZZ_TWIN=1: the HIP side is analysed on System's cell-sorted twin (forced, whatever the size and order of the system).

This exercises the policy layer above the C ABI as well (small-box replication, list reuse, triclinic alignment of
the Voronoi calls, column naming).  Integer columns must be equal, floating ones agree to 1e-6.  Test infrastructure.
"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _pytest.monkeypatch import MonkeyPatch

import _oracle_backend as ob
import mdapy_amd as mp
from mdapy_amd import devarray
from fuzz_parity import draw
from oracle import oracle as O


def make_system(s):
    rng = np.random.default_rng(s["seed"] + 3)
    box = mp.Box(s["box"], origin=s["origin"], boundary=s["bnd"])
    sysm = mp.System(pos=s["pos"], box=box)
    n = sysm.N
    cols = dict(type=rng.integers(1, 3, n).astype(np.int32), vx=rng.normal(0, 3, n), vy=rng.normal(0, 3, n), vz=rng.normal(0, 3, n),
                element=rng.choice(["Cu", "Al"], n))
    sysm.update_data(sysm.data.with_columns(**cols))
    return sysm


def plan(s):
    """the call sequence of one seed: (label, callable(system) -> object with curves or None)"""
    rng = np.random.default_rng(s["seed"] + 5)
    rc = float(rng.uniform(2.8, 4.8))
    ortho = not np.any(s["box"] - np.diag(np.diag(s["box"])))
    P = dict(csp_n=int(rng.choice([8, 12])), ent_local=bool(rng.integers(0, 2)), ent_avg=float(rng.choice([0.0, rc * 0.8])),
             cl_rc=float(rng.uniform(1.5, 3.2)), st_avg=bool(rng.integers(0, 2)), nbin=int(rng.integers(20, 120)),
             rdf_long=float(rng.uniform(5.0, 9.0)), sf_kmax=float(rng.uniform(3.0, 6.0)), sf_partial=bool(rng.integers(0, 2)),
             sf_rc=float(rng.uniform(6.0, 9.0)), vw=bool(rng.integers(0, 2)),
             rep=[int(v) for v in rng.permutation([2, 1, 1])])
    calls = [
        ("neighbor", lambda y: y.build_neighbor(rc)),
        ("cna_rc", lambda y: y.cal_common_neighbor_analysis(rc=min(rc, 3.6))),
        ("cna_adaptive", lambda y: y.cal_common_neighbor_analysis()),
        ("csp", lambda y: y.cal_centro_symmetry_parameter(P["csp_n"])),
        ("ids", lambda y: y.cal_identify_diamond_structure()),
        ("aja", lambda y: y.cal_ackland_jones_analysis()),
        ("cnp", lambda y: y.cal_common_neighbor_parameter(min(rc, 3.5))),
        ("entropy", lambda y: y.cal_structure_entropy(rc, 0.2, P["ent_local"], P["ent_avg"])),
        ("temperature", lambda y: y.cal_atomic_temperature(rc)),
        ("cluster", lambda y: y.cal_cluster_analysis(P["cl_rc"])),
        ("cluster_by_type", lambda y: y.cal_cluster_analysis({"1-1": 2.9, "1-2": 2.4, "2-2": 2.0})),
        ("steinhardt_nnn", lambda y: y.cal_steinhardt_bond_orientation([4, 6], nnn=12, average=P["st_avg"], wl=True, wlhat=True)),
        ("steinhardt_rc", lambda y: y.cal_steinhardt_bond_orientation([6, 8], rc=rc, identify_liquid=True)),
        ("rdf", lambda y: y.cal_radial_distribution_function(rc, P["nbin"])),
        ("rdf_long", lambda y: y.cal_radial_distribution_function(P["rdf_long"], 60)),
        ("wcp", lambda y: y.cal_warren_cowley_parameter(rc)),
        ("sfc_direct", lambda y: y.cal_structure_factor(0.5, P["sf_kmax"], 40, cal_partial=P["sf_partial"], mode="direct")),
        ("wrap", lambda y: y.wrap_pos()),
        ("replicate", lambda y: y.replicate(*P["rep"])),
        ("average", lambda y: y.average_by_neighbor(rc * 0.8, "vx", include_self=P["st_avg"])),
        ("sfc_debye", lambda y: y.cal_structure_factor(0.5, 8.0, 50, mode="debye", rc=P["sf_rc"])),
    ]
    if not s["unwrapped"]:
        if s["sigma"] != 0.0:
            calls.append(("ptm", lambda y: y.cal_polyhedral_template_matching("default", return_rmsd=True, return_ordering=True, return_atomic_distance=True)))
            calls.append(("ptm_faults", lambda y: y.cal_polyhedral_template_matching("fcc-hcp-bcc", identify_fcc_planar_faults=True)))
        if O.have_voro_ref() and s["kind"] not in ("blob", "tiny") and (ortho or all(s["bnd"])):
            calls.append(("voronoi_volume", lambda y: y.cal_voronoi_volume()))
            if all(s["bnd"]):
                calls.append(("steinhardt_voronoi", lambda y: y.cal_steinhardt_bond_orientation([6], use_voronoi=True, use_weight=P["vw"])))
    order = rng.permutation(len(calls))[: int(rng.integers(4, 9))]
    return [calls[i] for i in order]


def curves(obj):
    out = {}
    if obj is None:
        return out
    for name in ("g_total", "g", "r", "Npair", "WCP", "k", "Sk", "Sk_partial", "coordination"):
        v = getattr(obj, name, None)
        if v is None:
            continue
        if isinstance(v, dict):
            for kk, vv in v.items():
                out[f"{name}[{kk}]"] = np.asarray(vv)
        else:
            try:
                out[name] = np.asarray(v)
            except Exception:
                pass
    return out


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind in "iub" or a.dtype.kind in "USO":
        return bool(np.array_equal(a, b))
    fin = np.isfinite(a)
    return bool(np.array_equal(fin, np.isfinite(b)) and np.allclose(a[fin], b[fin], rtol=1e-6, atol=1e-8))


def run_seed(seed, fails):
    s = draw(seed)
    ran = 0
    try:
        calls = plan(s)
    except Exception as e:
        fails.append((seed, "plan", repr(e)[:160]))
        return 0
    a, b = make_system(s), make_system(s)
    if os.environ.get("FUZZ_TWIN") == "1":  # the HIP side analysed on its cell-sorted twin whatever its size and order, the oracle side as it is
        a._sort_mode, b._sort_mode = "1", "0"
    for label, fn in calls:
        if os.environ.get("FUZZ_TRACE"):
            print("  call", label, flush=True)
        res = []
        for which, sysm in (("hip", a), ("oracle", b)):
            patch = MonkeyPatch()
            try:
                if which == "oracle":  # host classes -> oracle, output buffers -> numpy
                    ob.install(patch)
                    patch.setattr(devarray, "_gpu", False)
                res.append(("ok", fn(sysm)))
            except Exception as e:
                res.append(("err", f"{type(e).__name__}: {str(e)[:140]}"))
            finally:
                patch.undo()
        if res[0][0] != res[1][0]:
            fails.append((seed, label, f"hip={res[0]} oracle={res[1]}"[:300]))
            break
        if res[0][0] == "err":
            if res[0][1].split(":")[0] != res[1][1].split(":")[0]:
                fails.append((seed, label, f"different errors: hip={res[0][1]} oracle={res[1][1]}"[:300]))
            break  # the same refusal on both sides ends the sequence
        bad = [c for c in a.data.columns if c not in b.data.columns or not same(a.data[c].to_numpy(), b.data[c].to_numpy())]
        ca, cb = curves(res[0][1]), curves(res[1][1])
        bad += [f"curve {k}" for k in ca if k not in cb or not same(ca[k], cb[k])]
        if bad:
            fails.append((seed, label, "differs: " + ", ".join(bad)))
            if os.environ.get("FUZZ_DEBUG") and hasattr(a, "verlet_list") and hasattr(b, "verlet_list"):
                va, vb = np.asarray(a.verlet_list), np.asarray(b.verlet_list)
                print("   debug", seed, label, "lists", va.shape, vb.shape, "equal", va.shape == vb.shape and np.array_equal(va, vb),
                      "dist equal", np.array_equal(np.asarray(a.distance_list), np.asarray(b.distance_list)),
                      "nn equal", np.array_equal(np.asarray(a.neighbor_number), np.asarray(b.neighbor_number)),
                      "rc", getattr(a, "rc", None), getattr(b, "rc", None), "ncl", getattr(a, "cluster_number", None), getattr(b, "cluster_number", None), flush=True)
                if "cluster_id" in a.data.columns:
                    ca, cb = a.data["cluster_id"].to_numpy(), b.data["cluster_id"].to_numpy()
                    w = np.flatnonzero(ca != cb)
                    print("   debug ids: N", len(ca), "differ at", len(w), w[:8], "hip", ca[w[:8]], "oracle", cb[w[:8]], "hip min/max", ca.min(), ca.max(), "oracle min/max", cb.min(), cb.max(),
                          "nn of those", np.asarray(a.neighbor_number)[w[:8]], flush=True)
            break
        ran += 1
    return ran


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    fails, ran = [], 0
    while time.time() - t0 < budget:
        if os.environ.get("FUZZ_TRACE"):
            print("seed", seed, flush=True)
        try:
            ran += run_seed(seed, fails)
        except Exception:
            fails.append((seed, "driver", traceback.format_exc()[-300:]))
        seed += 1
    print(f"fuzz_system: {ran} calls agreed over seeds up to {seed - 1}; {len(fails)} failures", flush=True)
    for f in fails[:60]:
        s = draw(f[0])
        print("  FAIL seed=%d call=%s %s  [kind=%s tri=%s unwrapped=%s bnd=%s N=%d]" % (f + (s["kind"], s["tri"], s["unwrapped"], s["bnd"].tolist(), len(s["pos"]))))
    return len(fails)


if __name__ == "__main__":
    sys.exit(min(main(), 100))
