#!/usr/bin/env python
"""A new System per trajectory frame, neighbor + CNA (the one-call path): frames in the builder's order, frames in a shuffled numbering
with every frame sorted (MDAPY_REUSE_ORDER=0), and frames read through the previous frame's permutation.  python tools/frame_probe.py [cells=136] [reps=20]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.devarray import HArray
from mdapy_amd.frame import Frame
from bench import slab_positions, A_CU, RC
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
x, y, z, _ = slab_positions(torch, dev, cells, 0, 0.0)
n = int(x.shape[0]); box = mp.Box(np.diag([A_CU * cells] * 3))
g = torch.Generator(device=dev); g.manual_seed(7)
perm = torch.randperm(n, device=dev, generator=g)
def frames_of(cols):
    return [tuple((c + 0.03 * torch.randn(c.shape, generator=g, dtype=torch.float64, device=dev)).contiguous() for c in cols) for _ in range(2)]
def run(frames, label):
    def one(i):
        f = frames[i & 1]
        s = mp.System(data=Frame({"x": HArray(f[0]), "y": HArray(f[1]), "z": HArray(f[2])}), box=box)
        s.cal_common_neighbor_analysis(rc=RC, max_neigh=16)
        return s
    for i in range(3): s = one(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): s = one(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e3
    frac = float((s.data["cna"].device_array().dev() == 1).double().mean().item())
    print(f"{label:62s} {dt:7.3f} ms per frame   fcc {frac:.4f}  twin {s._spatial() is not None}", flush=True)
    return dt
only = os.environ.get("FRAME_PROBE_ONLY", "")
a = run(frames_of((x, y, z)), "frames in the builder's order") if only in ("", "a") else 1.0
shuf = frames_of((x[perm].contiguous(), y[perm].contiguous(), z[perm].contiguous()))
os.environ["MDAPY_REUSE_ORDER"] = "0"
b = run(shuf, "shuffled numbering, every frame sorted") if only in ("", "b") else 1.0
del os.environ["MDAPY_REUSE_ORDER"]
c = run(shuf, "shuffled numbering, frames through the last permutation") if only in ("", "c") else 1.0
print(f"ratios to the ordered frame: sorted {b / a:.3f}, through the last permutation {c / a:.3f}")
