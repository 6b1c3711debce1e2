"""18-NN search, then CSP(12) and adaptive CNA on 10 M rattled fcc atoms: wall times through System (second pass).  Under
rocprofv3 --kernel-trace --stats the kernels' own times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
pos, box = lattice_positions("fcc", 3.615, cells, cells, cells)
pos += np.random.default_rng(0).normal(0.0, 0.05, pos.shape)
s = mp.System(pos=pos, box=box)
def lap(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(f"  {label:40s} {(time.perf_counter() - t0) * 1e3:8.2f} ms", flush=True)
for rep in range(3):
    print("pass", rep)
    lap("build_nearest_neighbor(18)", lambda: s.build_nearest_neighbor(18))
    lap("cal_centro_symmetry_parameter(12)", lambda: s.cal_centro_symmetry_parameter(12))
    lap("cal_common_neighbor_analysis() adaptive", lambda: s.cal_common_neighbor_analysis())
print("labels", np.bincount(s.data["cna"].to_numpy()), "csp mean", float(s.data["csp"].to_numpy().mean()))
