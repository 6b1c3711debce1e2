import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _neighbor
from mdapy_amd.devarray import HArray
from bench import slab_positions, A_CU
dev = torch.device("cuda", 0)
cells = 100
x, y, z, _ = slab_positions(torch, dev, cells, 0, 0.0)
n = int(x.shape[0]); box = mp.Box(np.diag([A_CU*cells]*3)); bx = (box.box, box.origin, box.boundary)
for rc, M in ((5.0, 50), (0.854*A_CU, 16), (3.615*1.01, 24)):
    v = HArray.empty((n, M), np.int32); d = HArray.empty((n, M), np.float64); nn = HArray.empty((n,), np.int32)
    _neighbor.build_neighbor(HArray(x), HArray(y), HArray(z), *bx, rc, v, d, nn, 1, fill_pads=True)
    v0, d0 = v.dev().clone(), d.dev().clone()
    for k in (12, 14):
        ts = []
        for rep in range(3):
            v.dev().copy_(v0); d.dev().copy_(d0); torch.cuda.synchronize()
            t0 = time.perf_counter(); _neighbor.sort_verlet_by_distance(v, d, k, 1); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
        t0 = time.perf_counter(); _neighbor.sort_verlet_by_distance(v, d, k, 1); torch.cuda.synchronize(); again = (time.perf_counter()-t0)*1e3
        print(f"N={n} rc={rc:.3f} M={M} nn={int(nn.dev().min())}..{int(nn.dev().max())} sort front {k}: {min(ts):.3f} ms (already sorted: {again:.3f} ms)", flush=True)
