#!/usr/bin/env python
"""fixed-cutoff CNA with the neighbours' positions from the caller's three arrays (variant 0) against 32-byte records packed at the
start of the call (variant 2), on the headline lattice in lattice order and under a random permutation.
python tools/fcna_rec_ab.py [cells=136]  -> profiles/r05_fcna_records.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _cna, _neighbor, _lib
from bench import slab_positions, A_CU, RC
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
dev = torch.device("cuda", 0)
L = _lib.lib()
box = mp.Box(np.diag([A_CU * cells] * 3)); bx = (box.box, box.origin, box.boundary)
M = 16
for sigma in (0.0, 0.2):
    x, y, z, _ = slab_positions(torch, dev, cells, 0, sigma)
    N = int(x.shape[0])
    perm = torch.randperm(N, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    for tag, cols in (("lattice order", (x, y, z)), ("shuffled", tuple(c[perm].contiguous() for c in (x, y, z)))):
        v = torch.empty((N, M), dtype=torch.int32, device=dev); d = torch.empty((N, M), dtype=torch.float64, device=dev)
        nn = torch.empty((N,), dtype=torch.int32, device=dev)
        _neighbor.build_neighbor(*cols, *bx, RC, v, d, nn, 1, fill_pads=True)
        res = {}
        for variant in (3, 2, 0):
            L.mdh_debug_set_fcna_variant(variant)
            pat = torch.zeros((N,), dtype=torch.int32, device=dev)
            for _ in range(2):
                pat.zero_(); _cna.fcna(*cols, *bx, v, nn, pat, RC, 1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                pat.zero_(); _cna.fcna(*cols, *bx, v, nn, pat, RC, 1)
            torch.cuda.synchronize()
            res[variant] = ((time.perf_counter() - t0) / 10 * 1e3, pat.clone())
        L.mdh_debug_set_fcna_variant(0)
        print(f"sigma {sigma:4.2f} {tag:14s}: three arrays {res[3][0]:7.3f} ms   32-byte records always (pack + CNA) {res[2][0]:7.3f} ms   decided on the device "
              f"(the product) {res[0][0]:7.3f} ms   labels equal: {bool(torch.equal(res[0][1], res[2][1]) and torch.equal(res[0][1], res[3][1]))}", flush=True)
