#!/usr/bin/env python
"""cal_voronoi_volume and build_voronoi_neighbor on a rattled fcc crystal (periodic, slab, free cluster), a few calls each, for
a kernel trace: python tools/voronoi_probe.py [cells=40]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pos, box = lattice_positions("fcc", 3.615, cells, cells, cells)
pos = pos + np.random.default_rng(0).normal(0.0, 0.05, pos.shape)
for tag, bd in (("periodic", [1, 1, 1]), ("slab", [1, 1, 0]), ("cluster", [0, 0, 0])):
    for name in ("cal_voronoi_volume", "build_voronoi_neighbor"):
        best = 1e9
        for rep in range(4):
            s = mp.System(pos=pos, box=mp.Box(box, boundary=bd))
            torch.cuda.synchronize(); t0 = time.perf_counter(); getattr(s, name)(); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        print(f"N {len(pos)} {tag:9s} {name:24s} {best:8.2f} ms")
