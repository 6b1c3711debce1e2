"""count-only pass vs build: python tools/nb_count_probe.py [cells] [sigma]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _lib, _neighbor
from bench import slab_positions, A_CU, RC
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device("cuda", 0)
x, y, z, gid = slab_positions(torch, dev, cells, 0, sigma)
n = x.shape[0]
box = mp.Box(np.diag([A_CU * cells] * 3))
L = _lib.lib()
keep, (pb, po, pp) = _lib.host_box(box.box, box.origin, box.boundary)
nn = torch.empty((n,), dtype=torch.int32, device=dev)
mx = ctypes.c_int(0)
st = int(torch.cuda.current_stream().cuda_stream)
plan = (ctypes.c_int * 8)()
def count():
    _lib.check(L.mdh_neighbor_count(x.data_ptr(), y.data_ptr(), z.data_ptr(), n, pb, po, pp, RC, nn.data_ptr(), ctypes.addressof(mx), _lib.DEVICE, st))
for _ in range(2): count()
L.mdh_debug_neighbor_plan(plan); print("count plan", list(plan), "max", mx.value)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): count()
torch.cuda.synchronize(); print("count ms", (time.perf_counter() - t0) / 5 * 1e3)
for _ in range(2): v, d, n2 = _neighbor.build_neighbor_without_max_neigh(x, y, z, box.box, box.origin, box.boundary, RC, 1)
L.mdh_debug_neighbor_plan(plan); print("exact plan (build pass)", list(plan), v.shape)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): v, d, n2 = _neighbor.build_neighbor_without_max_neigh(x, y, z, box.box, box.origin, box.boundary, RC, 1)
torch.cuda.synchronize(); print("exact ms", (time.perf_counter() - t0) / 5 * 1e3)
