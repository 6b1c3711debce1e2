#!/usr/bin/env python
"""k_rdf_tile on BASELINE config 4 (9.84 M-atom Cu64Zr36 glass, rc 8 A, 200 bins): the whole kernel, and the kernel with every pair
rejected (MDH_RDF_PROBE=1: the candidate walk and the single-precision distance tests alone, nothing pushed or binned) — the part of
the time that cells of rc/2 could shrink (they test ~0.58 of the pairs, bin the same hits).  python tools/rdf_probe.py [cells=135]"""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _lib
from mdapy_amd.build_lattice import lattice_positions
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 135
pos, box = lattice_positions("fcc", 4.0, cells, cells, cells)
pos += np.random.default_rng(7).normal(0.0, 0.35, pos.shape)
n = len(pos)
ty = np.repeat([1, 2], [int(round(0.64 * n)), n - int(round(0.64 * n))]).astype(np.int32)
np.random.default_rng(42).shuffle(ty)
s = mp.System(pos=pos, box=box); s.update_data(s.data.with_columns(type=ty))
L = _lib.lib()
for probe in (os.environ.get("RDF_PROBES", "0,1,0,1").split(",")):
    os.environ["MDH_RDF_PROBE"] = probe
    s.cal_radial_distribution_function(8.0, nbin=200, streaming=True)
    L.mdh_prof_reset(); L.mdh_prof_enable(1)
    for _ in range(3): g = s.cal_radial_distribution_function(8.0, nbin=200, streaming=True)
    torch.cuda.synchronize(); L.mdh_prof_enable(0)
    buf = ctypes.create_string_buffer(4096); L.mdh_prof_report(buf, 4096)
    rec = {ln.split()[0]: float(ln.split()[2]) / int(ln.split()[1]) for ln in buf.value.decode().strip().splitlines()}
    print(f"MDH_RDF_PROBE={probe}: k_rdf_tile {rec.get('k_rdf_tile', float('nan')):.3f} ms   (N = {n})", flush=True)
