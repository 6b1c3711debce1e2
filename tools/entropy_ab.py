#!/usr/bin/env python
"""structure entropy kernels side by side: python tools/entropy_ab.py [cells]   (4 M atoms, rc 5.0 / sigma 0.2 on a 50-wide list
and rc 3.6 / sigma 0.2 on a 24-wide one; variant 1 = direct kernel, 2..5 = ladder with 1/2/4/8 lanes to a row, 0 = automatic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _neighbor, _lib, _structure_entropy
from mdapy_amd.devarray import HArray
from bench import slab_positions, A_CU
dev = torch.device("cuda", 0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100
x, y, z, _ = slab_positions(torch, dev, cells, 0, 0.0)
n = int(x.shape[0]); box = mp.Box(np.diag([A_CU * cells] * 3)); bx = (box.box, box.origin, box.boundary)
vol = (A_CU * cells) ** 3
L = _lib.lib()
for rc, M, sigma in ((5.0, 50, 0.2), (3.6, 24, 0.2), (5.0, 50, 0.14)):
    v = HArray.empty((n, M), np.int32); d = HArray.empty((n, M), np.float64); nn = HArray.empty((n,), np.int32)
    _neighbor.build_neighbor(HArray(x), HArray(y), HArray(z), *bx, rc, v, d, nn, 1, fill_pads=True)
    ref = None
    for variant in (1, 2, 3, 4, 5, 0):
        _lib.check(L.mdh_debug_set_entropy_variant(variant))
        e = HArray.empty((n,), np.float64)
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _structure_entropy.calculate_structure_entropy(rc, sigma, False, vol, d, nn, e, 1)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        got = e.dev().clone()
        if ref is None: ref = got
        print(f"rc {rc} sigma {sigma} M {M} N {n} variant {variant}: {min(ts):8.3f} ms   max rel diff to the direct kernel {float(((got - ref).abs() / ref.abs().clamp_min(1e-300)).max()):.2e}", flush=True)
_lib.check(L.mdh_debug_set_entropy_variant(0))
