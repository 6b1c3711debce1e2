#!/usr/bin/env python
"""Per-kernel A/B timing on the GPU box (HIP events inside the library).  Usage: python tools/kbench.py [cells] [M] [rc_over_a]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mdapy_amd as mp
from mdapy_amd import _lib, _neighbor, _cna

sys.path.insert(0, ROOT)
from bench import slab_positions, A_CU

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rc = (float(sys.argv[3]) if len(sys.argv) > 3 else 0.854) * A_CU
sigma = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
dev = torch.device("cuda", 0)
x, y, z, gid = slab_positions(torch, dev, cells, 0, sigma)
n = x.shape[0]
box = mp.Box(np.diag([A_CU * cells] * 3))
L = _lib.lib()
verlet = torch.empty((n, M), dtype=torch.int32, device=dev)
dist = torch.empty((n, M), dtype=torch.float64, device=dev)
nn = torch.empty((n,), dtype=torch.int32, device=dev)
pat = torch.zeros((n,), dtype=torch.int32, device=dev)


def report(tag):
    buf = ctypes.create_string_buffer(1 << 16)
    L.mdh_prof_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().strip().splitlines():
        name, cnt, tot = line.split()
        out[name] = float(tot) / int(cnt)
    print(tag, {k: round(v, 4) for k, v in out.items()}, flush=True)
    return out


ref = None
for variant in (1, 2, 0):
    L.mdh_debug_set_neighbor_variant(variant)
    for it in range(2):
        _neighbor.build_neighbor(x, y, z, box.box, box.origin, box.boundary, rc, verlet, dist, nn, 1, fill_pads=True)
    torch.cuda.synchronize()
    L.mdh_prof_reset(); L.mdh_prof_enable(1)
    for it in range(5):
        _neighbor.build_neighbor(x, y, z, box.box, box.origin, box.boundary, rc, verlet, dist, nn, 1, fill_pads=True)
        pat.zero_()
        _cna.fcna(x, y, z, box.box, box.origin, box.boundary, verlet, nn, pat, rc, 1)
    torch.cuda.synchronize()
    L.mdh_prof_enable(0)
    r = report(f"variant={variant} N={n} M={M} rc={rc:.4f}")
    alg = (28 + 12 * M) * n
    print(f"   k_neighbor: {alg / r['k_neighbor'] / 1e6:.1f} GB/s algorithmic = {alg / r['k_neighbor'] / 1e6 / 8000 * 100:.2f}% of 8 TB/s; nn[min,max]={int(nn.min())},{int(nn.max())} labels={torch.bincount(pat).tolist()}")
    cur = (verlet.clone(), dist.clone(), nn.clone())
    if ref is None:
        ref = cur
    else:
        print("   identical to variant 1:", all(bool(torch.equal(a, b)) for a, b in zip(ref, cur)))
L.mdh_debug_set_neighbor_variant(0)
