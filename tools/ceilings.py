"""Measured ceilings on the GPU box, for DESIGN.md: HBM stream rates (torch copy / fill / reduce of 4 GiB) and the
PCIe-inclusive end-to-end rate of the bench workload (host numpy in -> host labels out)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions

n = 1 << 29  # 4 GiB of f64
a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
a.fill_(1.0); torch.cuda.synchronize()
def t(f, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.time() - t0) / reps
tc = t(lambda: b.copy_(a)); tf = t(lambda: b.fill_(2.0)); tr = t(lambda: a.sum())
print(f"copy  (read+write) {2 * n * 8 / tc / 1e12:.2f} TB/s")
print(f"fill  (write only) {n * 8 / tf / 1e12:.2f} TB/s")
print(f"sum   (read only)  {n * 8 / tr / 1e12:.2f} TB/s")
del a, b
pos, box = lattice_positions("fcc", 3.615, 136, 136, 136)
rc = 0.854 * 3.615
for it in range(3):
    t0 = time.time()
    s = mp.System(pos=pos, box=box)
    s.build_neighbor(rc, max_neigh=16)
    s.cal_common_neighbor_analysis(rc=rc)
    lab = s.data["cna"].to_numpy()
    dt = time.time() - t0
print(f"end-to-end host positions -> host CNA labels (lists stay in HBM): {dt * 1e3:.1f} ms = {len(pos) / dt / 1e6:.0f} M atoms/s, labels {np.bincount(lab)}")
t0 = time.time(); v = np.asarray(s.verlet_list); d = np.asarray(s.distance_list); dt2 = time.time() - t0
print(f"downloading the neighbor lists as numpy ({(v.nbytes + d.nbytes) / 1e9:.2f} GB): {dt2 * 1e3:.0f} ms = {(v.nbytes + d.nbytes) / dt2 / 1e9:.1f} GB/s")
