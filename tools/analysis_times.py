#!/usr/bin/env python
"""Wall time of the kNN-based analyses through the System API on one GPU (HBM-resident lists).  Usage: analysis_times.py [cells]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mdapy_amd as mp

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100
s = mp.build_crystal("Cu", "fcc", 3.615, nx=cells, ny=cells, nz=cells)
rng = np.random.default_rng(0)
d = s.data
s.update_data(d.with_columns(x=d["x"].to_numpy() + rng.normal(0, 0.05, s.N), y=d["y"].to_numpy() + rng.normal(0, 0.05, s.N),
                             z=d["z"].to_numpy() + rng.normal(0, 0.05, s.N)))
print(f"N = {s.N}")


def lap(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"  {label:34s} {dt * 1e3:9.1f} ms   {s.N / dt / 1e6:8.1f} M atoms/s", flush=True)


for rep in range(2):
    print("pass", rep)
    lap("build_nearest_neighbor(18)", lambda: s.build_nearest_neighbor(18))
    lap("cal_polyhedral_template_matching", lambda: s.cal_polyhedral_template_matching("default"))
    lap("cal_common_neighbor_analysis()", lambda: s.cal_common_neighbor_analysis())
    lap("cal_centro_symmetry_parameter(12)", lambda: s.cal_centro_symmetry_parameter(12))
    lap("cal_ackland_jones_analysis", lambda: s.cal_ackland_jones_analysis())
    lap("steinhardt q4,q6 (nnn=12)", lambda: s.cal_steinhardt_bond_orientation([4, 6], nnn=12))
    lap("build_neighbor(5.0)", lambda: s.build_neighbor(5.0))
    lap("cal_common_neighbor_analysis(rc)", lambda: s.cal_common_neighbor_analysis(rc=3.087))
    lap("rdf(5.0, 200 bins)", lambda: s.cal_radial_distribution_function(5.0, 200))
    lap("cal_voronoi_volume", lambda: s.cal_voronoi_volume())
