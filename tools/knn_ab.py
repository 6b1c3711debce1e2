#!/usr/bin/env python
"""k nearest neighbours: the counting kernel first (k_knn_select, mdh_debug_set_knn_variant(3), a measuring variant) against the
insertion kernel alone (the product), rows compared bit for bit.  python tools/knn_ab.py [cells=136] [sigma=0.05]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _fast_knn, _lib
from bench import slab_positions, A_CU
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
dev = torch.device("cuda", 0)
L = _lib.lib()
box = np.diag([A_CU * cells] * 3); org, bnd = np.zeros(3), np.array([1, 1, 1], np.int32)
for sigma in (0.05, 0.0, 0.3):
    x, y, z, _ = slab_positions(torch, dev, cells, 0, sigma)
    N = int(x.shape[0])
    for k in (12, 14, 18, 24):
        res = {}
        for variant in (0, 3):
            L.mdh_debug_set_knn_variant(variant)
            idx = torch.empty((N, k), dtype=torch.int32, device=dev); d = torch.empty((N, k), dtype=torch.float64, device=dev)
            for it in range(2): _fast_knn.knn(x, y, z, box, org, bnd, k, idx, d, 1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for it in range(5): _fast_knn.knn(x, y, z, box, org, bnd, k, idx, d, 1)
            torch.cuda.synchronize()
            res[2 if variant == 0 else 0] = ((time.perf_counter() - t0) / 5 * 1e3, idx, d)
        L.mdh_debug_set_knn_variant(0)
        same = bool(torch.equal(res[0][1], res[2][1])) and bool(torch.equal(res[0][2], res[2][2]))
        print(f"sigma {sigma:4.2f} k={k:2d}: insertion kernel {res[2][0]:7.2f} ms   counting kernel first {res[0][0]:7.2f} ms   rows identical: {same}", flush=True)
