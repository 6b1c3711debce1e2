#!/usr/bin/env python
"""The analyses of BASELINE.json configs 2 and 4 at their full sizes, through the System API, for rocprofv3:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_an -o an -- python tools/profile_analyses.py c3 c5

c3 = 136^3 fcc Cu rattled by N(0, 0.05) seed 0: kNN-18 -> PTM, Steinhardt q4/q6 (rc = 0.85 a and nnn = 12), CSP-12, adaptive CNA
c5 = 135^3 x 4 sites of fcc a = 4.0 displaced by N(0, 0.35) seed 7, Cu64Zr36: streaming partial g_ab(r) rc 8 / 200 bins, WCP rc 3.6
Prints wall times per call (second pass: scratch cache warm)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions

which = sys.argv[1:] or ["c3", "c5"]
small = "--small" in which


def lap(label, n, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"  {label:44s} {dt * 1e3:9.1f} ms   {n / dt / 1e6:8.1f} M atoms/s", flush=True)
    return r


if "c3" in which:
    cells = 40 if small else 136
    pos, box = lattice_positions("fcc", 3.615, cells, cells, cells)
    pos += np.random.default_rng(0).normal(0.0, 0.05, pos.shape)
    s = mp.System(pos=pos, box=box)
    print(f"c3: N = {s.N}")
    for rep in range(2):
        print(" pass", rep)
        lap("build_nearest_neighbor(18)", s.N, lambda: s.build_nearest_neighbor(18))
        lap("cal_polyhedral_template_matching(fcc-hcp-bcc)", s.N, lambda: s.cal_polyhedral_template_matching("fcc-hcp-bcc"))
        lap("cal_polyhedral_template_matching(all)", s.N, lambda: s.cal_polyhedral_template_matching("all"))
        lap("steinhardt [4,6] nnn=12", s.N, lambda: s.cal_steinhardt_bond_orientation([4, 6], nnn=12))
        lap("cal_centro_symmetry_parameter(12)", s.N, lambda: s.cal_centro_symmetry_parameter(12))
        lap("cal_common_neighbor_analysis() adaptive", s.N, lambda: s.cal_common_neighbor_analysis())
        lap("build_neighbor(0.85a)", s.N, lambda: s.build_neighbor(0.85 * 3.615, max_neigh=16))
        lap("steinhardt [4,6] rc=0.85a", s.N, lambda: s.cal_steinhardt_bond_orientation([4, 6], rc=0.85 * 3.615))
    print("  ptm labels", np.bincount(s.data["ptm"].to_numpy(), minlength=9).tolist())
    del s

if "c5" in which:
    cells = 40 if small else 135
    pos, box = lattice_positions("fcc", 4.0, cells, cells, cells)
    pos += np.random.default_rng(7).normal(0.0, 0.35, pos.shape)
    n = len(pos)
    ty = np.repeat([1, 2], [int(round(0.64 * n)), n - int(round(0.64 * n))]).astype(np.int32)
    np.random.default_rng(42).shuffle(ty)
    s = mp.System(pos=pos, box=box)
    s.update_data(s.data.with_columns(type=ty))
    print(f"c5: N = {s.N}")
    for rep in range(2):
        print(" pass", rep)
        rdf = lap("rdf(8.0, 200) streaming partials", s.N, lambda: s.cal_radial_distribution_function(8.0, nbin=200, streaming=True))
        lap("build_neighbor(3.6)", s.N, lambda: s.build_neighbor(3.6))
        w = lap("warren_cowley(3.6)", s.N, lambda: s.cal_warren_cowley_parameter(3.6))
    print("  g_total[::40]", np.round(rdf.g_total[::40], 4).tolist(), " WCP", np.round(np.asarray(w.WCP), 5).tolist())
