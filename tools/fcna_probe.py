#!/usr/bin/env python
"""fixed-cutoff CNA on a rattled lattice, many calls: the command a rocprofv3 --kernel-trace --stats run wraps.
    python tools/fcna_probe.py [cells=136] [sigma=0.20] [reps=20]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _cna, _neighbor
from bench import slab_positions, A_CU, RC
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
x, y, z, _ = slab_positions(torch, dev, cells, 0, sigma)
N = int(x.shape[0]); M = 16
box = mp.Box(np.diag([A_CU * cells] * 3)); bx = (box.box, box.origin, box.boundary)
v = torch.empty((N, M), dtype=torch.int32, device=dev); d = torch.empty((N, M), dtype=torch.float64, device=dev)
nn = torch.empty((N,), dtype=torch.int32, device=dev); pat = torch.zeros((N,), dtype=torch.int32, device=dev)
_neighbor.build_neighbor(x, y, z, *bx, RC, v, d, nn, 1, fill_pads=True)
for _ in range(reps):
    pat.zero_()
    _cna.fcna(x, y, z, *bx, v, nn, pat, RC, 1)
torch.cuda.synchronize()
print("nn histogram", torch.bincount(nn.clamp(0, 20)).tolist(), "labels", torch.bincount(pat, minlength=5).tolist())
