#!/usr/bin/env python
"""Every analysis on a wrapped frame and on the same frame unwrapped (atoms +-3 whole box lengths away), through System: where an
unwrapped trajectory falls off a fast path.  python tools/unwrapped_sweep.py [cells=100]"""
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
far = pos + rng.integers(-3, 4, pos.shape) * np.diag(box)
ty = rng.integers(1, 3, len(pos)).astype(np.int32)
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
         ("aja", lambda s: s.cal_ackland_jones_analysis())]
res = {}
for tag, p in (("wrapped", pos), ("unwrapped", far)):
    for rep in range(2):
        s = mp.System(pos=p, box=box)
        s.update_data(s.data.with_columns(type=ty))
        for name, fn in CALLS:
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(s); torch.cuda.synchronize()
            res[(tag, name)] = (time.perf_counter() - t0) * 1e3
        out = {c: s.data[c].to_numpy() for c in ("cna", "ptm", "csp", "aja") if c in s.data.columns}
    res[(tag, "_out")] = out
print(f"N = {len(pos)}")
for name, _ in CALLS:
    a, b = res[("wrapped", name)], res[("unwrapped", name)]
    print(f"{name:32s} wrapped {a:8.2f} ms   unwrapped {b:8.2f} ms   x{b / a:5.2f}")
for c, v in res[("wrapped", "_out")].items():
    w = res[("unwrapped", "_out")][c]
    same = np.array_equal(v, w) if v.dtype.kind in "iu" else np.allclose(v, w, rtol=1e-9, atol=1e-9)
    print(f"column {c}: unwrapped == wrapped: {same}")
