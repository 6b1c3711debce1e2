"""The reference's own call pattern — numpy positions in, numpy labels out — timed piece by piece on the GPU box."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
pos, box = lattice_positions("fcc", 3.615, cells, cells, cells)
rc = 0.854 * 3.615
def lap(label, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); print(f"   {label}: {(t1 - t0) * 1e3:.1f} ms"); return t1
for it in range(3):
    torch.cuda.synchronize(); t00 = t0 = time.perf_counter()
    s = mp.System(pos=pos, box=box)
    if it == 2: t0 = lap("System(pos=(N,3) numpy, box)", t0)
    s.build_neighbor(rc, max_neigh=16)
    if it == 2: t0 = lap("build_neighbor", t0)
    s.cal_common_neighbor_analysis(rc=rc)
    if it == 2: t0 = lap("cal_common_neighbor_analysis", t0)
    lab = s.data["cna"].to_numpy()
    if it == 2: t0 = lap("labels to numpy", t0)
    dt = time.perf_counter() - t00
print(f"end-to-end host positions -> host CNA labels (lists stay in HBM), {len(pos)} atoms: {dt * 1e3:.1f} ms = {len(pos) / dt / 1e6:.0f} M atoms/s, labels {np.bincount(lab)}")
