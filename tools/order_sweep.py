#!/usr/bin/env python
"""Every analysis through System on the SAME atoms handed in in lattice order and under one random permutation — the latter
twice: analysed in the order it has (MDAPY_SPATIAL_SORT=0) and on System's cell-sorted twin (the default for a system of this
size).  python tools/order_sweep.py [cells=100]   -> profiles/r05_order_sweep.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pos, box = lattice_positions("fcc", 3.615, cells, cells, cells)
rng = np.random.default_rng(0)
pos = pos + rng.normal(0.0, 0.05, pos.shape)
ty = rng.integers(1, 3, len(pos)).astype(np.int32)
perm = rng.permutation(len(pos))
CALLS = [("build_neighbor(0.854a, 16)", lambda s: s.build_neighbor(0.854 * 3.615, max_neigh=16)),
         ("cna(0.854a)", lambda s: s.cal_common_neighbor_analysis(0.854 * 3.615)),
         ("build_nearest_neighbor(18)", lambda s: s.build_nearest_neighbor(18)),
         ("ptm", lambda s: s.cal_polyhedral_template_matching()),
         ("csp(12)", lambda s: s.cal_centro_symmetry_parameter(12)),
         ("adaptive cna", lambda s: s.cal_common_neighbor_analysis()),
         ("steinhardt [4,6] nnn=12", lambda s: s.cal_steinhardt_bond_orientation([4, 6], nnn=12)),
         ("steinhardt [6] rc", lambda s: s.cal_steinhardt_bond_orientation([6], rc=0.85 * 3.615)),
         ("rdf(8, 200) streaming", lambda s: s.cal_radial_distribution_function(8.0, 200, streaming=True)),
         ("wcp(3.6)", lambda s: s.cal_warren_cowley_parameter(3.6)),
         ("cnp(0.854a)", lambda s: s.cal_common_neighbor_parameter(0.854 * 3.615)),
         ("aja", lambda s: s.cal_ackland_jones_analysis()),
         ("entropy(5, 0.2)", lambda s: s.cal_structure_entropy(5.0, 0.2)),
         ("read verlet_list on the host", lambda s: np.asarray(s.verlet_list).shape)]
res = {}
KINDS = (("lattice order", pos, ty, ""), ("shuffled, as it is", pos[perm], ty[perm], "0"), ("shuffled, twin", pos[perm], ty[perm], ""))
for tag, p, t, mode in KINDS:
    if mode:
        os.environ["MDAPY_SPATIAL_SORT"] = mode
    else:
        os.environ.pop("MDAPY_SPATIAL_SORT", None)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = mp.System(pos=p, box=box)
        s.update_data(s.data.with_columns(type=t))
        torch.cuda.synchronize()
        res[(tag, "System(pos, box)")] = (time.perf_counter() - t0) * 1e3
        for name, fn in CALLS:
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(s); torch.cuda.synchronize()
            res[(tag, name)] = (time.perf_counter() - t0) * 1e3
        out = {c: s.data[c].to_numpy() for c in ("cna", "ptm", "csp", "aja", "ql4", "ql6", "cnp", "entropy") if c in s.data.columns}
    res[(tag, "_out")] = out
    res[(tag, "_twin")] = s._spatial() is not None
print(f"N = {len(pos)}; twin made: " + ", ".join(f"{k[0]}: {res[(k[0], '_twin')]}" for k in KINDS))
print(f"{'call':32s} {'lattice order':>14s} {'shuffled as is':>15s} {'':>7s} {'shuffled twin':>14s}")
for name in ["System(pos, box)"] + [c[0] for c in CALLS]:
    a, b, c = (res[(k[0], name)] for k in KINDS)
    print(f"{name:32s} {a:11.2f} ms {b:12.2f} ms  x{b / a:5.2f} {c:11.2f} ms  x{c / a:5.2f}")
tot = [sum(res[(k[0], c[0])] for c in CALLS) for k in KINDS]
print(f"{'sum of the calls':32s} {tot[0]:11.2f} ms {tot[1]:12.2f} ms  x{tot[1] / tot[0]:5.2f} {tot[2]:11.2f} ms  x{tot[2] / tot[0]:5.2f}")
inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
for c, v in res[("lattice order", "_out")].items():
    for tag in ("shuffled, as it is", "shuffled, twin"):
        w = res[(tag, "_out")][c][inv]
        same = np.array_equal(v, w) if v.dtype.kind in "iu" else np.allclose(v, w, rtol=1e-9, atol=1e-12)
        print(f"column {c:8s} {tag:20s} == lattice order (atom by atom): {same}")
