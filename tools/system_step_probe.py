"""the headline step through the System API on HBM-resident columns: build_neighbor(rc, max_neigh=16) + fixed-cutoff CNA"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions

pos, box = lattice_positions("fcc", 3.615, 136, 136, 136)
s = mp.System(pos=pos, box=box)
rc = 0.854 * 3.615
for name, fn in (("build_neighbor(rc, 16)", lambda: s.build_neighbor(rc, max_neigh=16)),
                 ("cal_common_neighbor_analysis(rc)", lambda: s.cal_common_neighbor_analysis(rc)),
                 ("both", lambda: (s.build_neighbor(rc, max_neigh=16), s.cal_common_neighbor_analysis(rc)))):
    for rep in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) * 100:8.3f} ms per call", flush=True)
print("labels", np.bincount(s.data["cna"].to_numpy(), minlength=5).tolist())
