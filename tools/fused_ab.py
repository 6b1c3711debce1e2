#!/usr/bin/env python
"""The headline step as two C-ABI calls (build_neighbor, then fcna) against ONE fused call (mdh_build_neighbor_fcna: the label of a
centre worked out inside the tile kernel).  python tools/fused_ab.py [cells] [sigma] [steps]
NB_LIB=<path>: another build of the library (make -C mdapy_amd/csrc fcna64: the fused label with double-precision pair tests, as
until round 5).  Prints ms per step (wall, stream idle at both ends), the ranges of the library's own HIP events, and whether
lists and labels of the two forms are identical."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mdapy_amd as mp
from mdapy_amd import _lib, _neighbor, _cna
from bench import slab_positions, A_CU, RC

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
kind = sys.argv[4] if len(sys.argv) > 4 else "fcc"   # "bcc": a bcc Fe lattice of about the same size, rc = 1.2 a (14 neighbours: (6,6,6) and (4,4,4) bonds)
M = 16
dev = torch.device("cuda", 0)
if os.environ.get("NB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NB_LIB"])
L = _lib.lib()
if kind == "bcc":
    from mdapy_amd.build_lattice import lattice_positions
    a_fe = 2.87
    nb = int(round(cells * (2.0 ** (1.0 / 3.0))))
    pos, boxm = lattice_positions("bcc", a_fe, nb, nb, nb)
    if sigma:
        pos = pos + np.random.default_rng(3).normal(0.0, sigma, pos.shape)
    x, y, z = (torch.from_numpy(np.ascontiguousarray(pos[:, k])).to(dev) for k in range(3))
    b = mp.Box(boxm)
    RC = 1.2 * a_fe
    del pos
else:
    x, y, z, _ = slab_positions(torch, dev, cells, 0, sigma)
    b = mp.Box(np.diag([A_CU * cells] * 3))
n = int(x.shape[0])
bx = (b.box, b.origin, b.boundary)


def bufs():
    return (torch.empty((n, M), dtype=torch.int32, device=dev), torch.empty((n, M), dtype=torch.float64, device=dev),
            torch.empty((n,), dtype=torch.int32, device=dev), torch.zeros((n,), dtype=torch.int32, device=dev))


v1, d1, c1, p1 = bufs()
v2, d2, c2, p2 = bufs()


def two_calls():
    p1.zero_()
    _neighbor.build_neighbor(x, y, z, *bx, RC, v1, d1, c1, 1, fill_pads=True)
    _cna.fcna(x, y, z, *bx, v1, c1, p1, RC, 1)


def fused():
    p2.zero_()
    _neighbor.build_neighbor_fcna(x, y, z, *bx, RC, v2, d2, c2, p2, 1, fill_pads=True)


def report():
    buf = ctypes.create_string_buffer(1 << 16)
    L.mdh_prof_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().strip().splitlines():
        name, cnt, tot = line.split()
        out[name] = round(float(tot) / int(cnt), 4)
    return out


res = {}
for name, fn in (("two_calls", two_calls), ("fused", fused), ("two_calls", two_calls), ("fused", fused)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    L.mdh_prof_reset(); L.mdh_prof_enable(1)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    L.mdh_prof_enable(0)
    r = report()
    res.setdefault(name, []).append(ms)
    print(f"{name:10s} {kind} N={n} sigma={sigma}: {ms:.4f} ms per step; ranges {r}", flush=True)
same = bool(torch.equal(v1, v2) and torch.equal(d1, d2) and torch.equal(c1, c2) and torch.equal(p1, p2))
print("lists and labels identical:", same, "labels", torch.bincount(p1).tolist())
a, f = min(res["two_calls"]), min(res["fused"])
print(f"fused / two calls = {f / a:.4f}  ({a:.4f} -> {f:.4f} ms)")
