#!/usr/bin/env python
"""The tile plan (mdh_debug_neighbor_plan) and the tiles listed for the slice pass, for the plain build and for the build that also labels.
python tools/plan_probe.py [cells] [sigma]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _lib, _neighbor
from bench import slab_positions, A_CU, RC
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device("cuda", 0)
L = _lib.lib()
x, y, z, _ = slab_positions(torch, dev, cells, 0, sigma)
n, M = int(x.shape[0]), 16
b = mp.Box(np.diag([A_CU * cells] * 3)); bx = (b.box, b.origin, b.boundary)
v = torch.empty((n, M), dtype=torch.int32, device=dev); d = torch.empty((n, M), dtype=torch.float64, device=dev)
c = torch.empty((n,), dtype=torch.int32, device=dev); p = torch.zeros((n,), dtype=torch.int32, device=dev)
plan = (ctypes.c_int * 8)(); cnt = (ctypes.c_int64 * 4)()
which = sys.argv[3] if len(sys.argv) > 3 else "both"
for name, fn in (("plain", lambda: _neighbor.build_neighbor(x, y, z, *bx, RC, v, d, c, 1, fill_pads=True)),
                 ("labels", lambda: _neighbor.build_neighbor_fcna(x, y, z, *bx, RC, v, d, c, p, 1, fill_pads=True))):
    if which != "both" and which != name:
        continue
    L.mdh_debug_track_counters(1)
    for it in range(4):
        fn()
        torch.cuda.synchronize()
        L.mdh_debug_neighbor_plan(plan); L.mdh_debug_counters(cnt)
        print("  call", it, name, list(plan), "listed (sink):", int(cnt[1]), flush=True)
    L.mdh_debug_track_counters(0)
    print(name, "plan [txy, tz, cap, lds, flags, ...] =", list(plan), "tiles to the slice pass:", int(cnt[1]), flush=True)
