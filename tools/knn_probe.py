"""k nearest neighbours on 10 M rattled fcc atoms, k = 12, 14, 18: wall time of the search on HBM-resident positions.
python tools/knn_probe.py [cells]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from mdapy_amd import _fast_knn, _lib
from mdapy_amd.build_lattice import lattice_positions
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
pos, box = lattice_positions("fcc", 3.615, cells, cells, cells)
pos += np.random.default_rng(0).normal(0.0, 0.05, pos.shape)
dev = torch.device("cuda", 0)
x, y, z = (torch.from_numpy(np.ascontiguousarray(pos[:, k])).to(dev) for k in range(3))
if os.environ.get('NB_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['NB_LIB'])  # A/B against another build
N = len(pos); L = _lib.lib()
for k in (12, 14, 18, 24):
    idx = torch.empty((N, k), dtype=torch.int32, device=dev); d = torch.empty((N, k), dtype=torch.float64, device=dev)
    for it in range(2): _fast_knn.knn(x, y, z, box, np.zeros(3), np.array([1, 1, 1], np.int32), k, idx, d, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(5): _fast_knn.knn(x, y, z, box, np.zeros(3), np.array([1, 1, 1], np.int32), k, idx, d, 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"k={k}: {dt * 1e3:.2f} ms per search (cell grid + kernels), checksum {float(d[:, -1].sum())!r}")
