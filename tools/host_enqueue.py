#!/usr/bin/env python
"""How much of a small step is the host: time inside the Python calls of one neighbor + CNA step (enqueue only, the queue drained
every 20 steps outside the clock) against the step time with the device in the loop.  python tools/host_enqueue.py [cells=10]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _cna, _neighbor
from bench import slab_positions, A_CU, RC
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
x, y, z, _ = slab_positions(torch, dev, cells, 0, 0.0)
N = int(x.shape[0]); M = 16
box = mp.Box(np.diag([A_CU * cells] * 3)); bx = (box.box, box.origin, box.boundary)
v = torch.empty((N, M), dtype=torch.int32, device=dev); d = torch.empty((N, M), dtype=torch.float64, device=dev)
nn = torch.empty((N,), dtype=torch.int32, device=dev); pat = torch.empty((N,), dtype=torch.int32, device=dev)
parts = {"pat.zero_()": lambda: pat.zero_(), "build_neighbor": lambda: _neighbor.build_neighbor(x, y, z, *bx, RC, v, d, nn, 1, fill_pads=True),
         "fcna": lambda: _cna.fcna(x, y, z, *bx, v, nn, pat, RC, 1)}
for f in parts.values(): f()
torch.cuda.synchronize()
acc = {k: 0.0 for k in parts}; reps = 400
for r in range(reps):
    for k, f in parts.items():
        t0 = time.perf_counter(); f(); acc[k] += time.perf_counter() - t0
    if r % 20 == 19: torch.cuda.synchronize()
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(reps):
    for f in parts.values(): f()
torch.cuda.synchronize()
whole = (time.perf_counter() - t0) / reps * 1e6
print(f"N = {N}: step {whole:.1f} us; host time inside the calls (enqueue): " + ", ".join(f"{k} {v / reps * 1e6:.1f}" for k, v in acc.items()) + f" = {sum(acc.values()) / reps * 1e6:.1f} us")
if "--profile" in sys.argv:
    import cProfile, io, pstats
    pr = cProfile.Profile(); pr.enable()
    for r in range(200):
        for f in parts.values(): f()
        if r % 20 == 19: torch.cuda.synchronize()
    pr.disable()
    out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(25)
    print("\n".join(l for l in out.getvalue().splitlines() if l.strip())[:5000])
