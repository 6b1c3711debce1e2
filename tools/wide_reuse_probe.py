#!/usr/bin/env python
"""analyses that borrow a WIDE list: python tools/wide_reuse_probe.py   (4 M atoms of rattled fcc Cu, two species, velocities; every
call once on a System that holds only its own list and once after build_neighbor(5.0, max_neigh=50) — the reference's policy
reuses a held cutoff list whenever it reaches far enough, whatever its width)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions

pos, box = lattice_positions("fcc", 3.615, 100, 100, 100)
rng = np.random.default_rng(0)
pos = pos + rng.normal(0, 0.05, pos.shape)
vel = rng.normal(0, 1.0, pos.shape)
cols = dict(vx=vel[:, 0], vy=vel[:, 1], vz=vel[:, 2], amass=np.full(len(pos), 63.5), type=rng.integers(1, 3, len(pos)).astype(np.int32))
rc = 0.854 * 3.615


def system(wide):
    s = mp.System(pos=pos, box=box)
    s.update_data(s.data.with_columns(**cols))
    if wide:
        s.build_neighbor(5.0, max_neigh=50)
    return s


CALLS = [
    ("common_neighbor_analysis(rc)", lambda s: s.cal_common_neighbor_analysis(rc)),
    ("common_neighbor_parameter(rc)", lambda s: s.cal_common_neighbor_parameter(rc)),
    ("steinhardt [4,6] rc", lambda s: s.cal_steinhardt_bond_orientation([4, 6], rc=rc)),
    ("steinhardt [6] rc identify_liquid", lambda s: s.cal_steinhardt_bond_orientation([6], rc=rc, identify_liquid=True)),
    ("warren_cowley(3.6)", lambda s: s.cal_warren_cowley_parameter(3.6)),
    ("rdf(5.0, 200) from the list", lambda s: s.cal_radial_distribution_function(5.0, 200, streaming=False)),
    ("cluster_analysis(3.0)", lambda s: s.cal_cluster_analysis(3.0)),
    ("atomic_temperature(5.0)", lambda s: s.cal_atomic_temperature(5.0)),
    ("structure_entropy(5.0, 0.2, average 4.0)", lambda s: s.cal_structure_entropy(5.0, 0.2, False, 4.0)),
    ("centro_symmetry(12)", lambda s: s.cal_centro_symmetry_parameter(12)),
    ("ackland_jones", lambda s: s.cal_ackland_jones_analysis()),
]
for name, fn in CALLS:
    out = []
    for wide in (False, True):
        best = 1e9
        for rep in range(3):
            s = system(wide)
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(s); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
            del s
        out.append(best)
    print(f"{name:44s} own list {out[0]:8.2f} ms   after build_neighbor(5.0, 50) {out[1]:8.2f} ms", flush=True)
