#!/usr/bin/env python
"""Cost of the per-step host/torch work of the multi-GPU step on ONE GPU: mdapy_amd.distributed.exchange_halo with the two
P2P exchanges replaced by a loop-back (the slab receives what its periodic neighbours — identical perfect-FCC slabs — would
send), followed by the neighbor + CNA kernels on owned + ghost atoms.  Usage: python tools/halo_cost.py [cells] [world]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import mdapy_amd as mp
from mdapy_amd import distributed as D
from bench import slab_positions, A_CU, RC

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rank = 1
dev = torch.device("cuda", 0)
x, y, z, gid = slab_positions(torch, dev, cells, rank, 0.0)
n = int(x.shape[0])
box = mp.Box(np.diag([A_CU * cells * world, A_CU * cells, A_CU * cells]))
dec = D.SlabDecomposition(box, rank, world, axis=0)
x, y, z, gid = (dec.with_room(a, 0.05) for a in (x, y, z, gid))  # ghosts are appended behind the owned atoms in place
Lx = A_CU * cells


from tools import _loopback
_loopback.install(dec, Lx, n)  # the two P2P copies of the ring as two device kernels (message + the neighbour slab's offset)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


import ctypes
from mdapy_amd import _lib
L = _lib.lib()


def prof(fn):
    fn(); torch.cuda.synchronize()
    L.mdh_prof_reset(); L.mdh_prof_enable(1)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    L.mdh_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    L.mdh_prof_report(buf, len(buf))
    return {l.split()[0]: round(float(l.split()[2]) / int(l.split()[1]), 3) for l in buf.value.decode().strip().splitlines()}


ms_halo, dom = timed(lambda: dec.exchange_halo(x, y, z, gid, RC))
ms_step, out = timed(lambda: D.neighbor_cna_step(dec, x, y, z, gid, RC, 16))
ms_pipe, out = timed(lambda: D.neighbor_cna_step(dec, x, y, z, gid, RC, 16, next_frame=(x, y, z, gid)), reps=20)
dec._drop_pending(); torch.cuda.synchronize()
dom_p, v_p, d_p, nn_p, pat_p = out
ok_pipe = bool((nn_p[dom_p.owned] == 12).all()) and bool((pat_p[dom_p.owned] == 1).all()) and int(dom_p.x.shape[0]) > n
dom2, v, d, nn, pat = out
ok = bool((nn[dom2.owned] == 12).all()) and bool((pat[dom2.owned] == 1).all())
print("kernels (ms):", prof(lambda: D.neighbor_cna_step(dec, x, y, z, gid, RC, 16)))
print(f"owned {n}, ghosts {int(dom.x.shape[0]) - n}; exchange_halo (loop-back) {ms_halo:.2f} ms; whole step {ms_step:.2f} ms; "
      f"with the next frame's exchange under way on a side stream {ms_pipe:.2f} ms; owned atoms all FCC with 12 neighbours: {ok} / {ok_pipe}")
# the undivided step on the same box, for the ratio
xs, ys, zs, _ = slab_positions(torch, dev, cells, 0, 0.0)
ubox = mp.Box(np.diag([A_CU * cells] * 3))
from mdapy_amd import _neighbor as _nb, _cna as _cn
uv, ud = torch.empty((n, 16), dtype=torch.int32, device=dev), torch.empty((n, 16), dtype=torch.float64, device=dev)
un, up = torch.empty((n,), dtype=torch.int32, device=dev), torch.zeros((n,), dtype=torch.int32, device=dev)
def undivided():  # (the headline step: lists and labels in one call, as the decomposed step makes them)
    up.zero_()
    _nb.build_neighbor_fcna(xs, ys, zs, ubox.box, ubox.origin, ubox.boundary, RC, uv, ud, un, up, 1, fill_pads=True)
ms_und, _ = timed(undivided, reps=20)
print(f"undivided step of {n} atoms on the same box {ms_und:.2f} ms: loop-back slab step = {ms_step / ms_und:.3f}x, pipelined {ms_pipe / ms_und:.3f}x")

import time as _t
def lap(label, fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = _t.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); print(f"   {label}: {(_t.perf_counter()-t0)/reps*1e3:.2f} ms"); return r
h = dec.halo_fraction(RC); lo, hi = rank / world, (rank + 1) / world
lap("select_device", lambda: dec._select_device(x, y, z, hi - h, lo + h))
d0 = lap("exchange_halo sort=False", lambda: dec.exchange_halo(x, y, z, gid, RC, sort=False))
nloc = int(d0.x.shape[0])
def alloc():
    return (torch.empty((nloc, 16), dtype=torch.int32, device=dev), torch.empty((nloc, 16), dtype=torch.float64, device=dev),
            torch.empty((nloc,), dtype=torch.int32, device=dev), torch.zeros((nloc,), dtype=torch.int32, device=dev))
v_, d_, n_, p_ = lap("alloc", alloc)
from mdapy_amd import _neighbor, _cna
b = dec.box
lap("build_neighbor keyed", lambda: _neighbor.build_neighbor(d0.x, d0.y, d0.z, b.box, b.origin, b.boundary, RC, v_, d_, n_, 1, fill_pads=True, key=d0.gid))
lap("build_neighbor plain", lambda: _neighbor.build_neighbor(d0.x, d0.y, d0.z, b.box, b.origin, b.boundary, RC, v_, d_, n_, 1, fill_pads=True))
lap("fcna", lambda: _cna.fcna(d0.x, d0.y, d0.z, b.box, b.origin, b.boundary, v_, n_, p_, RC, 1))
