#!/usr/bin/env python
"""Where the host time of the System path goes: a NEW System per frame (HBM-resident columns) + cal_common_neighbor_analysis(rc, 16),
as bench.py's extra.shuffled_ids.system_path runs it.  Wall time per frame, then cProfile's top functions by cumulative time."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.devarray import HArray
from mdapy_amd.frame import Frame
from bench import slab_positions, A_CU, RC

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
dev = torch.device("cuda", 0)
x, y, z, _ = slab_positions(torch, dev, cells, 0, 0.0)
box = mp.Box(np.diag([A_CU * cells] * 3))


def frame():
    s = mp.System(data=Frame({"x": HArray(x), "y": HArray(y), "z": HArray(z)}), box=box)
    s.cal_common_neighbor_analysis(rc=RC, max_neigh=16)
    return s


for _ in range(5):
    s = frame()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30):
    s = frame()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per frame (device busy or not: wall)")
t0 = time.perf_counter()
for _ in range(30):
    s = frame()
host = (time.perf_counter() - t0) / 30 * 1e3
torch.cuda.synchronize()
print(f"{host:.3f} ms of host time per frame until the last call returns (the device may still be working)")
pr = cProfile.Profile(); pr.enable()
for _ in range(30):
    s = frame()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative")
st.print_stats(28)
