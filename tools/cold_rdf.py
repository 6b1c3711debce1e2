#!/usr/bin/env python
"""The first streaming-RDF call of BASELINE config 4 (9.84 M-atom Cu64Zr36 glass, rc 8, 200 bins) taken apart: what of its
time is the upload of the frame, what the first launches, what the kernel (77 ms against 7).
    python tools/cold_rdf.py [cells=135]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 135


def lap(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"  {label:58s} {(time.perf_counter() - t0) * 1e3:9.2f} ms", flush=True)
    return r


pos, box = lattice_positions("fcc", 4.0, cells, cells, cells)
pos += np.random.default_rng(7).normal(0.0, 0.35, pos.shape)
n = len(pos)
ty = np.repeat([1, 2], [int(round(0.64 * n)), n - int(round(0.64 * n))]).astype(np.int32)
np.random.default_rng(42).shuffle(ty)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
print(f"N = {n}")
s = lap("System(pos, box)", lambda: mp.System(pos=pos, box=box))
lap("update_data(with_columns(type))", lambda: s.update_data(s.data.with_columns(type=ty)))
pr = cProfile.Profile(); pr.enable()
lap("rdf(8.0, 200, streaming) first", lambda: s.cal_radial_distribution_function(8.0, nbin=200, streaming=True))
pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(22)
print("\n".join(l for l in out.getvalue().splitlines() if l.strip())[:4000])
lap("rdf(8.0, 200, streaming) second", lambda: s.cal_radial_distribution_function(8.0, nbin=200, streaming=True))
s2 = lap("System(pos, box) again", lambda: mp.System(pos=pos, box=box))
lap("update_data(with_columns(type)) again", lambda: s2.update_data(s2.data.with_columns(type=ty)))
lap("rdf on the second System, first call", lambda: s2.cal_radial_distribution_function(8.0, nbin=200, streaming=True))
