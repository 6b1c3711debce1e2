#!/usr/bin/env python
"""Steinhardt q4 + q6 over 12 nearest neighbours at 10 M atoms through System, with the library's event ranges: python tools/sq_probe.py [cells=136]"""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _lib
from mdapy_amd.devarray import HArray
from mdapy_amd.frame import Frame
from bench import slab_positions, A_CU
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
dev = torch.device("cuda", 0); L = _lib.lib()
x, y, z, _ = slab_positions(torch, dev, cells, 0, 0.05)
s = mp.System(data=Frame({"x": HArray(x), "y": HArray(y), "z": HArray(z)}), box=mp.Box(np.diag([A_CU * cells] * 3)))
s.build_nearest_neighbor(12)
for kw in (dict(nnn=12), dict(nnn=12, wl=True, wlhat=True), dict(nnn=12, average=True)):
    s.cal_steinhardt_bond_orientation([4, 6], **kw)
    torch.cuda.synchronize(); L.mdh_prof_reset(); L.mdh_prof_enable(1); t0 = time.perf_counter()
    for _ in range(3): s.cal_steinhardt_bond_orientation([4, 6], **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3; L.mdh_prof_enable(0)
    buf = ctypes.create_string_buffer(4096); L.mdh_prof_report(buf, 4096)
    print(kw, f"{dt:.2f} ms |", buf.value.decode().strip().replace("\n", " | "), " q6 mean", float(s.data["ql6"].to_numpy().mean()))
