#!/usr/bin/env python
"""where a k-nearest search spends its time: the event ranges of the library (mdh_prof) around one search, rattled fcc Cu
python tools/knn_split.py [cells=136] [k=18]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from mdapy_amd import _fast_knn, _lib
from bench import slab_positions, A_CU
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
ks = [int(a) for a in sys.argv[2:]] or [12, 14, 18]
dev = torch.device("cuda", 0); L = _lib.lib()
box = np.diag([A_CU * cells] * 3); org, bnd = np.zeros(3), np.array([1, 1, 1], np.int32)
x, y, z, _ = slab_positions(torch, dev, cells, 0, 0.05)
N = int(x.shape[0])
for k in ks:
    idx = torch.empty((N, k), dtype=torch.int32, device=dev); d = torch.empty((N, k), dtype=torch.float64, device=dev)
    for it in range(2): _fast_knn.knn(x, y, z, box, org, bnd, k, idx, d, 1)
    L.mdh_prof_reset(); L.mdh_prof_enable(1)
    for it in range(4): _fast_knn.knn(x, y, z, box, org, bnd, k, idx, d, 1)
    torch.cuda.synchronize(); L.mdh_prof_enable(0)
    buf = ctypes.create_string_buffer(4096); L.mdh_prof_report(buf, 4096)
    print(f"k={k}:", buf.value.decode().strip().replace("\n", " | "), "(4 calls)")
