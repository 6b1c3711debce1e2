#!/usr/bin/env python
"""The slab of rank 1 of 8 of the headline box (bench.py's extra.strong_1of8), exchange in loop-back: K steps, for a kernel trace.
python tools/strong_probe.py [cells=136] [steps=20]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.distributed import SlabDecomposition, neighbor_cna_step
from tools import _loopback
from bench import slab_positions, A_CU, RC

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
w8, M = 8, 16
weak = len(sys.argv) > 3 and sys.argv[3] == "weak"  # a whole cells^3 slab of an (8 cells) x cells x cells box instead of an eighth of the cells^3 box
cx = cells if weak else cells // w8
sx, sy, sz, sg = slab_positions(torch, dev, cells, 1, 0.0, cells_x=cx)
n = int(sx.shape[0])
dec = SlabDecomposition(mp.Box(np.diag([A_CU * cx * w8, A_CU * cells, A_CU * cells])), 1, w8, axis=0)
_loopback.install(dec, A_CU * cx, n)
sx, sy, sz, sg = (dec.with_room(a, 0.05 if weak else 0.25) for a in (sx, sy, sz, sg))
for prefetch in (False, True):
    nf = (sx, sy, sz, sg) if prefetch else None
    for _ in range(3):
        out = neighbor_cna_step(dec, sx, sy, sz, sg, RC, M, next_frame=nf, reuse_buffers=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = neighbor_cna_step(dec, sx, sy, sz, sg, RC, M, next_frame=nf, reuse_buffers=True)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dec._drop_pending(); torch.cuda.synchronize()
    dom, v, d, nn, pat = out
    ok = bool((nn[dom.owned] == 12).all().item()) and bool((pat[dom.owned] == 1).all().item())
    print(f"slab {cx}x{cells}x{cells} cells, {n} owned atoms, local atoms {int(dom.x.shape[0])} (static block: {getattr(dom, 'absent_slots', False)}), "
          f"prefetch={prefetch}: {dt / steps * 1e3:.4f} ms per step (host enqueue {t_host / steps * 1e3:.4f} ms), all fcc: {ok}", flush=True)
