"""one neighbor build + CNA at the bench size, for rocprofv3 counter passes: python tools/nb_once.py [cells] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _neighbor, _cna
from bench import slab_positions, A_CU
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
M, rc = 16, 0.854 * A_CU
dev = torch.device("cuda", 0)
x, y, z, gid = slab_positions(torch, dev, cells, 0, 0.0)
n = x.shape[0]
box = mp.Box(np.diag([A_CU * cells] * 3))
verlet = torch.empty((n, M), dtype=torch.int32, device=dev); dist = torch.empty((n, M), dtype=torch.float64, device=dev)
nn = torch.empty((n,), dtype=torch.int32, device=dev); pat = torch.zeros((n,), dtype=torch.int32, device=dev)
for _ in range(reps):
    _neighbor.build_neighbor(x, y, z, box.box, box.origin, box.boundary, rc, verlet, dist, nn, 1, fill_pads=True)
    _cna.fcna(x, y, z, box.box, box.origin, box.boundary, verlet, nn, pat, rc, 1)
torch.cuda.synchronize()
print("done", n)
