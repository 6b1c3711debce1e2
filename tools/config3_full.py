#!/usr/bin/env python
"""BASELINE config 3 at full size on ONE GPU: 512-grain polycrystalline Cu in a 1057 A box (~1e8 atoms) through the
package's own builder (Voronoi grains, rotation + half-space filter per grain, overlap removal at 2.0 A), then
neighbor + fixed-cutoff CNA.  python tools/config3_full.py [box] [grains]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp

L = float(sys.argv[1]) if len(sys.argv) > 1 else 1057.0
G = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rng = np.random.default_rng(2024)
seeds = rng.random((G, 3)) * L
theta = rng.uniform(-180, 180, (G, 3))
unit = mp.build_crystal("Cu", "fcc", 3.615)
t0 = time.perf_counter()
poly = mp.CreatePolycrystal(unit, box=L, seed_number=G, seed_position=seeds, theta_list=theta, metal_overlap_dis=2.0)
s = poly.compute(verbose=True)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"built {s.N:,} atoms in {t1 - t0:.1f} s", flush=True)
rc = 0.854 * 3.615
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.build_neighbor(rc, max_neigh=20)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s.cal_common_neighbor_analysis(rc=rc)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"pass {rep}: build_neighbor {1e3 * (t1 - t0):.1f} ms, CNA {1e3 * (t2 - t1):.1f} ms = {s.N / (t2 - t0) / 1e6:.1f} M atoms/s end to end", flush=True)
lab = np.bincount(s.data["cna"].to_numpy(), minlength=5)
print("labels other/fcc/hcp/bcc/ico:", lab.tolist(), "fcc fraction %.4f" % (lab[1] / s.N), "max neighbours", int(np.asarray(s.neighbor_number).max()))
print("GPU memory allocated (torch) %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))
