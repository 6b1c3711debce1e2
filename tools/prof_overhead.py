"""what the HIP-event ranges of bench.py's live kernel timing cost per step: the headline step timed with and without them"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _lib, _neighbor, _cna
from bench import slab_positions, A_CU, RC
dev = torch.device("cuda", 0)
x, y, z, gid = slab_positions(torch, dev, 136, 0, 0.0)
n = x.shape[0]; box = mp.Box(np.diag([A_CU * 136] * 3)); bx = (box.box, box.origin, box.boundary)
v = torch.empty((n, 16), dtype=torch.int32, device=dev); d = torch.empty((n, 16), dtype=torch.float64, device=dev)
nn = torch.empty((n,), dtype=torch.int32, device=dev); pat = torch.empty((n,), dtype=torch.int32, device=dev)
L = _lib.lib()
def step():
    pat.zero_()
    _neighbor.build_neighbor(x, y, z, *bx, RC, v, d, nn, 1, fill_pads=True)
    _cna.fcna(x, y, z, *bx, v, nn, pat, RC, 1)
for on in (1, 0, 1, 0):
    L.mdh_prof_reset(); L.mdh_prof_enable(on)
    for _ in range(20): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    L.mdh_prof_enable(0)
    print(f"event ranges {'on ' if on else 'off'}: {dt * 1e3:.4f} ms per step")
