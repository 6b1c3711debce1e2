#!/usr/bin/env python
"""Steinhardt stage 1, lane per atom (the product's k_sq_stage1_l) against sixteen lanes per atom (k_sq_stage1_g16, a measuring
variant): q4 / q6 of BASELINE config 2's 10 M rattled fcc atoms over the 12 nearest neighbours and over a cutoff list.
python tools/sq_wave_ab.py [cells=136]  -> profiles/r05_sq_wave.txt (run under rocprofv3 --kernel-trace --stats for the kernel times)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mdapy_amd as mp
from mdapy_amd import _lib, _sbo, _fast_knn, _neighbor
from mdapy_amd.devarray import HArray
from bench import slab_positions, A_CU

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
dev = torch.device("cuda", 0)
x, y, z, _ = slab_positions(torch, dev, cells, 0, 0.05)
n = int(x.shape[0])
box = mp.Box(np.diag([A_CU * cells] * 3))
bx = (box.box, box.origin, box.boundary)
X, Y, Z = HArray(x), HArray(y), HArray(z)
L = _lib.lib()
ll = np.array([4, 6], np.int32)
idx = HArray.empty((n, 12), np.int32); dk = HArray.empty((n, 12), np.float64)
_fast_knn.knn(X, Y, Z, *bx, 12, idx, dk, 1)
nn12 = HArray.full((n,), 12, np.int32)
rc = 0.85 * A_CU
v = HArray.empty((n, 16), np.int32); d = HArray.empty((n, 16), np.float64); nn = HArray.empty((n,), np.int32)
_neighbor.build_neighbor(X, Y, Z, *bx, rc, v, d, nn, 1, fill_pads=True)


def run(variant, rows, dist, counts, nnn, cutoff):
    L.mdh_debug_set_sq_variant(variant)
    qr = HArray.full((n, 2, 13), 0.0, np.float64); qi = HArray.full((n, 2, 13), 0.0, np.float64); qn = HArray.full((n, 2), 0.0, np.float64)
    def call():
        qr.dev().zero_(); qi.dev().zero_()
        _sbo.get_sq(X, Y, Z, *bx, rows, dist, counts, np.zeros((2, 2)), ll, nnn, 6, False, False, False, False, cutoff, False, qr, qi, qn, 1)
    call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    L.mdh_debug_set_sq_variant(0)
    return (time.perf_counter() - t0) / 5 * 1e3, qn.numpy().copy()


print(f"N = {n}")
for tag, args in (("12 nearest neighbours", (idx, dk, nn12, 12, 0.0 + 1e9)), (f"cutoff list rc = 0.85 a (rows of 16)", (v, d, nn, 0, rc))):
    ta, qa = run(0, *args)
    tb, qb = run(2, *args)
    rel = np.abs(qa - qb).max() / np.abs(qa).max()
    print(f"{tag:40s} lane per atom {ta:7.3f} ms   sixteen lanes per atom {tb:7.3f} ms  (x{tb / ta:4.2f})   max |dq| / max q = {rel:.1e}   "
          f"q4 {qa[:, 0].mean():.6f} q6 {qa[:, 1].mean():.6f}")
print("(whole mdh_get_sq calls incl. two memsets of the q_lm arrays; kernel times: the rocprofv3 table below)")
