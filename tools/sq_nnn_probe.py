"""where the time of cal_steinhardt_bond_orientation([4, 6], nnn=12) goes after an 18-NN search (10 M atoms)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions

pos, box = lattice_positions("fcc", 3.615, 136, 136, 136)
pos += np.random.default_rng(0).normal(0.0, 0.05, pos.shape)
s = mp.System(pos=pos, box=box)
s.build_nearest_neighbor(18)

def lap(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{label:40s} {(time.perf_counter() - t0) * 1e3:8.2f} ms", flush=True)
    return r

for rep in range(3):
    lap("_nearest_prefix(12)", lambda: s._nearest_prefix(12))
    lap("_get_compute_view", lambda: s._get_compute_view())
    lap("whole call nnn=12", lambda: s.cal_steinhardt_bond_orientation([4, 6], nnn=12))
    lap("whole call rc", lambda: s.cal_steinhardt_bond_orientation([4, 6], rc=0.85 * 3.615) if "rc" in s.__dict__ else None)
