#!/usr/bin/env python
"""Where the time of a shuffled frame's System step goes: python tools/twin_probe.py [cells]  (10 M atoms by default)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mdapy_amd as mp
from mdapy_amd import _order, _neighbor, _cna
from mdapy_amd.devarray import HArray
from mdapy_amd.frame import Frame
from bench import slab_positions, A_CU, RC

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
dev = torch.device("cuda", 0)
x, y, z, _ = slab_positions(torch, dev, cells, 0, 0.0)
n = int(x.shape[0])
gen = torch.Generator(device=dev); gen.manual_seed(11)
perm0 = torch.randperm(n, device=dev, generator=gen)
xs_, ys_, zs_ = (c[perm0].contiguous() for c in (x, y, z))
box = mp.Box(np.diag([A_CU * cells] * 3))
bx = (box.box, box.origin, box.boundary)
M = 16


def lap(label, fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print(f"  {label:58s} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms", flush=True)
    return out


for tag, cols in (("ordered", (x, y, z)), ("shuffled", (xs_, ys_, zs_))):
    print(tag)
    H = [HArray(c) for c in cols]
    lap("order_statistic", lambda: _order.order_statistic(*H, *bx))
    xs, ys, zs, perm, cnt = lap("spatial_sort", lambda: _order.spatial_sort(*H, *bx))
    key = lap("perm -> int64 key", lambda: HArray(perm.dev().long()))
    v = HArray.empty((n, M), np.int32); d = HArray.empty((n, M), np.float64); nn = HArray.empty((n,), np.int32)
    lap("build_neighbor keyed on the sorted copy", lambda: _neighbor.build_neighbor(xs, ys, zs, *bx, RC, v, d, nn, 1, fill_pads=True, key=key))
    lap("build_neighbor plain on the input order", lambda: _neighbor.build_neighbor(*H, *bx, RC, v, d, nn, 1, fill_pads=True))
    lap("build_neighbor plain on the sorted copy", lambda: _neighbor.build_neighbor(xs, ys, zs, *bx, RC, v, d, nn, 1, fill_pads=True))
    lap("nn.max()", lambda: nn.max(initial=0))
    pat = HArray.full((n,), 0, np.int32)
    lap("fcna on the sorted copy", lambda: _cna.fcna(xs, ys, zs, *bx, v, nn, pat, RC, 1))
    lap("labels scattered back", lambda: _order.permute(pat, perm, scatter=True))
    lap("rows translated (ids, distances, counts)", lambda: _order.translate_rows(v, d, nn, perm))

    def system():
        s = mp.System(data=Frame({"x": H[0], "y": H[1], "z": H[2]}), box=box)
        s.cal_common_neighbor_analysis(rc=RC, max_neigh=M)
        return s
    s = lap("System(...).cal_common_neighbor_analysis(rc, 16)", system)
    print("   twin:", s._spatial() is not None, " fcc:", int((s.data["cna"].to_numpy() == 1).sum()) == n)

# the bench's form of the same step: a NEW System per step from torch tensors, a few steps, after the C-ABI step on the same columns
v = torch.empty((n, M), dtype=torch.int32, device=dev); d = torch.empty((n, M), dtype=torch.float64, device=dev)
nn = torch.empty((n,), dtype=torch.int32, device=dev); pat = torch.zeros((n,), dtype=torch.int32, device=dev)
for _ in range(3):
    _neighbor.build_neighbor(xs_, ys_, zs_, *bx, RC, v, d, nn, 1, fill_pads=True); _cna.fcna(xs_, ys_, zs_, *bx, v, nn, pat, RC, 1)
torch.cuda.synchronize()
def bench_like():
    s = mp.System(data=Frame({"x": HArray(xs_), "y": HArray(ys_), "z": HArray(zs_)}), box=box)
    s.cal_common_neighbor_analysis(rc=RC, max_neigh=M)
    return s
for _ in range(2):
    out = bench_like()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = bench_like()
torch.cuda.synchronize()
print(f"a NEW System per step from the shuffled columns (the bench's extra.shuffled_ids.system_path): {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")
