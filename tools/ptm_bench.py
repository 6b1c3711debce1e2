"""Time mdh_ptm on device-resident inputs: python tools/ptm_bench.py [cells]  (fcc cells per axis, default 64 -> 1.05 M atoms)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdapy_amd as mp
from mdapy_amd import _ptm, _lib
from mdapy_amd.build_lattice import lattice_positions

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lat = os.environ.get("PTM_LATTICE", "fcc")  # fcc / bcc / hcp lattice, or "gas" (uniform random points)
pos, box = lattice_positions("fcc" if lat == "gas" else lat, 3.615 if lat != "bcc" else 2.87, n, n, n)
if lat == "gas":
    pos = np.random.default_rng(1).random(pos.shape) * np.diag(np.asarray(box, float) if np.ndim(box) == 2 else np.diag(box))
pos = pos + np.random.default_rng(0).normal(0, float(os.environ.get("PTM_SIGMA", "0.08")), pos.shape)
s = mp.System(pos=pos, box=box)
t0 = time.time(); s.build_nearest_neighbor(18) if hasattr(s, "build_nearest_neighbor") else None
torch.cuda.synchronize(); print("knn s", time.time() - t0)
N = len(pos)
x, y, z = (torch.from_numpy(np.ascontiguousarray(pos[:, k])).cuda() for k in range(3))
v = s.verlet_list
v = v.dev() if hasattr(v, "dev") else torch.from_numpy(np.asarray(v)).cuda()
out = torch.zeros((N, 8), dtype=torch.float64, device="cuda"); ind = torch.zeros((N, 18), dtype=torch.int32, device="cuda")
b = np.asarray(box, float); b = b if b.shape == (3, 3) else np.diag(b)
import ctypes
L = _lib.lib()
if len(sys.argv) > 3: L.mdh_debug_set_ptm_order_cap(int(sys.argv[3]))
structures = sys.argv[2].split("+") if len(sys.argv) > 2 else ["fcc-hcp-bcc", "all"]
for structure in structures:
    L.mdh_prof_reset(); L.mdh_prof_enable(1)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        _ptm.get_ptm(structure, x, y, z, b, np.zeros(3), np.array([1, 1, 1], np.int32), v, None, 0.1, out, ind)
        torch.cuda.synchronize(); dt = time.time() - t0
    L.mdh_prof_enable(0)
    buf = ctypes.create_string_buffer(4096); L.mdh_prof_report(buf, 4096)
    print(buf.value.decode().strip().replace("\n", " | "))
    print(f"{structure}: N={N} {dt*1e3:.1f} ms  {N/dt/1e6:.2f} M atoms/s  types {np.bincount(out[:,0].cpu().numpy().astype(int), minlength=6)}")
