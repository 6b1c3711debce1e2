#!/usr/bin/env python
"""k nearest neighbours from the rows of a cutoff build (round 6: k_knn_rows behind the tile kernel) against the cell walk alone
(MDH_KNN_ROWS=0, a process of its own): python tools/knn_rows_ab.py [cells=136]   -> ms per search, and a checksum of the rows
(equal checksums in the two runs = identical rows)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from mdapy_amd import _fast_knn, _lib
from bench import slab_positions, A_CU
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
dev = torch.device("cuda", 0)
box = np.diag([A_CU * cells] * 3); org, bnd = np.zeros(3), np.array([1, 1, 1], np.int32)
for kind, sigma in (("fcc", 0.05), ("fcc", 0.0), ("fcc", 0.3), ("bcc", 0.05), ("gas", 0.0)):
    x, y, z, _ = slab_positions(torch, dev, cells, 0, sigma)
    if kind == "bcc":
        from mdapy_amd.build_lattice import lattice_positions
        pb, bb = lattice_positions("bcc", 2.87, 171, 171, 171)
        pb = pb + np.random.default_rng(1).normal(0, sigma, pb.shape)
        x, y, z = (torch.from_numpy(np.ascontiguousarray(pb[:, c])).to(dev) for c in range(3))
        box = np.asarray(bb, float)
    if kind == "gas":
        box = np.diag([A_CU * cells] * 3)
        g = torch.Generator(device=dev); g.manual_seed(3)
        x, y, z = (torch.rand(x.shape[0], dtype=torch.float64, device=dev, generator=g) * (A_CU * cells) for _ in range(3))
    N = int(x.shape[0])
    for k in (12, 14, 18, 24):
        idx = torch.empty((N, k), dtype=torch.int32, device=dev); d = torch.empty((N, k), dtype=torch.float64, device=dev)
        for it in range(2): _fast_knn.knn(x, y, z, box, org, bnd, k, idx, d, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for it in range(5): _fast_knn.knn(x, y, z, box, org, bnd, k, idx, d, 1)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        chk = int((idx.long() * torch.arange(1, k + 1, device=dev)).sum().item()) ^ int(d.view(torch.int64).sum().item() & 0xffffffffffff)
        print(f"{kind} sigma {sigma:4.2f} k={k:2d}: {ms:7.2f} ms   checksum {chk:x}", flush=True)
