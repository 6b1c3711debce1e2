"""Wall times of the list consumers and the other secondary analyses through System on one GPU: 4 000 000-atom fcc Cu (100^3 cells),
N(0, 0.05) rattle, velocities and two species.  python tools/consumer_times.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions
pos, box = lattice_positions("fcc", 3.615, 100, 100, 100)   # 4 M atoms
rng = np.random.default_rng(0)
pos = pos + rng.normal(0, 0.05, pos.shape)
vel = rng.normal(0, 1.0, pos.shape)
s = mp.System(pos=pos, box=box)
s.update_data(s.data.with_columns(vx=vel[:, 0], vy=vel[:, 1], vz=vel[:, 2], amass=np.full(len(pos), 63.5), type=rng.integers(1, 3, len(pos)).astype(np.int32)))
def T(name, fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); print(f"{name:50s} {(time.perf_counter()-t0)/reps*1e3:8.1f} ms", flush=True)
rc = 0.854 * 3.615
T("common_neighbor_parameter(rc)", lambda: s.cal_common_neighbor_parameter(rc))
T("ackland_jones", lambda: s.cal_ackland_jones_analysis())
T("structure_entropy(rc=5, sigma=0.2)", lambda: s.cal_structure_entropy(5.0, 0.2))
T("structure_entropy(local density, average 4)", lambda: s.cal_structure_entropy(5.0, 0.2, True, 4.0))
T("atomic_temperature(rc=5)", lambda: s.cal_atomic_temperature(5.0))
T("cluster_analysis(rc=3)", lambda: s.cal_cluster_analysis(3.0))
T("identify_diamond", lambda: s.cal_identify_diamond_structure())
T("steinhardt [4,6] nnn=12 average", lambda: s.cal_steinhardt_bond_orientation([4, 6], nnn=12, average=True))
T("steinhardt [6] rc=0.85a, identify_liquid", lambda: s.cal_steinhardt_bond_orientation([6], rc=rc, identify_liquid=True))
T("rdf(rc=5, list)", lambda: s.cal_radial_distribution_function(5.0, 100, streaming=False))
T("voronoi_volume", lambda: s.cal_voronoi_volume(), reps=1)
