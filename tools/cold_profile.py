#!/usr/bin/env python
"""cProfile of the FIRST build_neighbor / cal_* of a process (after System): which host code the cold calls spend their time in.
    python tools/cold_profile.py [cells=63]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 63
A = 3.615; rc = 0.854 * A
pos0, box0 = lattice_positions("fcc", A, cells, cells, cells)
pos0 = pos0 + np.random.default_rng(11).normal(0.0, 0.05, pos0.shape)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
s = mp.System(pos=pos0, box=box0); torch.cuda.synchronize()
for name, fn in [("build_neighbor", lambda: s.build_neighbor(rc)), ("cna", lambda: s.cal_common_neighbor_analysis(rc)),
                 ("csp", lambda: s.cal_centro_symmetry_parameter(12))]:
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
    ms = (time.perf_counter() - t0) * 1e3
    out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(14)
    print(f"==== first {name}: {ms:.2f} ms"); print("\n".join(l for l in out.getvalue().splitlines() if l.strip())[:2600])
