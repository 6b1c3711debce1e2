#!/usr/bin/env python
"""Every analysis on the same crystal in an orthogonal box, in a 10 % sheared periodic box and in the sheared box open along its
second vector, through System: where a triclinic / open box falls off a fast path.  python tools/triclinic_sweep.py [cells=100]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions
cells = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
pos, box = lattice_positions("fcc", 3.615, cells, cells, cells)
rng = np.random.default_rng(0)
pos = pos + rng.normal(0.0, 0.05, pos.shape)
L = 3.615 * cells
sh = 0.1
H = np.array([[L, 0, 0], [sh * L, L, 0], [0.5 * sh * L, sh * L, L]])
tri = pos @ (H / L)
ty = rng.integers(1, 3, len(pos)).astype(np.int32)
CALLS = [("build_neighbor(0.854a)", lambda s: s.build_neighbor(0.854 * 3.615, max_neigh=None if "--disorder" in sys.argv else 16)),
         ("cna(0.854a)", lambda s: s.cal_common_neighbor_analysis(0.854 * 3.615)),
         ("build_nearest_neighbor(18)", lambda s: s.build_nearest_neighbor(18)),
         ("ptm", lambda s: s.cal_polyhedral_template_matching()),
         ("csp(12)", lambda s: s.cal_centro_symmetry_parameter(12)),
         ("adaptive cna", lambda s: s.cal_common_neighbor_analysis()),
         ("steinhardt [4,6] nnn=12", lambda s: s.cal_steinhardt_bond_orientation([4, 6], nnn=12)),
         ("steinhardt [6] rc", lambda s: s.cal_steinhardt_bond_orientation([6], rc=0.85 * 3.615)),
         ("rdf(8, 200) streaming", lambda s: s.cal_radial_distribution_function(8.0, 200, streaming=True)),
         ("wcp(3.6)", lambda s: s.cal_warren_cowley_parameter(3.6)),
         ("aja", lambda s: s.cal_ackland_jones_analysis())]
CASES = [("orthogonal", pos, mp.Box(box)), ("sheared", tri, mp.Box(H)), ("sheared, open b", tri, mp.Box(H, boundary=[1, 0, 1]))]
if "--secondary" in sys.argv:  # the list consumers and Voronoi instead of the main analyses (run it on a smaller system: Voronoi is ~25 ns per atom)
    CALLS = [("cnp(0.854a)", lambda s: s.cal_common_neighbor_parameter(0.854 * 3.615)),
             ("structure entropy(5, 0.2)", lambda s: s.cal_structure_entropy(5.0, 0.2)),
             ("cluster analysis(3.0)", lambda s: s.cal_cluster_analysis(3.0)),
             ("identify diamond", lambda s: s.cal_identify_diamond_structure()),
             ("voronoi volume", lambda s: s.cal_voronoi_volume()),
             ("structure factor(debye, rc 10)", lambda s: s.cal_structure_factor(0.5, 10.0, 100, mode="debye", rc=10.0)),
             ("average_by_neighbor(3.0, x)", lambda s: s.average_by_neighbor(3.0, "x"))]
if "--tertiary" in sys.argv:  # Voronoi neighbour rows and their consumer, the list form of the RDF, averaged Steinhardt, bookkeeping calls
    CALLS = [("build_voronoi_neighbor", lambda s: s.build_voronoi_neighbor()),
             ("steinhardt [6] use_voronoi", lambda s: s.cal_steinhardt_bond_orientation([6], use_voronoi=True)),
             ("steinhardt [4,6] nnn=12 average", lambda s: s.cal_steinhardt_bond_orientation([4, 6], nnn=12, average=True)),
             ("atomic temperature(5.0)", lambda s: s.cal_atomic_temperature(5.0)),
             ("rdf(5, 200) from the list", lambda s: s.cal_radial_distribution_function(5.0, 200, streaming=False)),
             ("structure factor(direct, 50 bins)", lambda s: s.cal_structure_factor(0.5, 5.0, 50, mode="direct")),
             ("ptm(all)", lambda s: s.cal_polyhedral_template_matching("all")),
             ("wrap_pos", lambda s: s.wrap_pos()),
             ("replicate(2,2,1)", lambda s: s.replicate(2, 2, 1)),
             ("delete_overlap(1.0)", lambda s: s.delete_overlap(1.0))]
if "--disorder" in sys.argv:  # the same lattice rattled more and more (sigma 0.05 / 0.20 / 0.50 A): hot crystal, liquid-like
    base, _ = lattice_positions("fcc", 3.615, cells, cells, cells)
    CASES = [(f"sigma {sg}", base + np.random.default_rng(1).normal(0.0, sg, base.shape), mp.Box(box)) for sg in (0.05, 0.20, 0.50)]
if "--sizes" in sys.argv:  # the same crystal at 10^3, 20^3, 40^3 cells: what a call costs when the system is small (host overhead)
    CASES = []
    for n in (10, 20, 40):
        pn, bn = lattice_positions("fcc", 3.615, n, n, n)
        CASES.append((f"{len(pn)} atoms", pn + np.random.default_rng(2).normal(0.0, 0.05, pn.shape), mp.Box(bn)))
if "--open" in sys.argv:  # the orthogonal box periodic, as a slab (open z) and as a cluster (open everywhere)
    CASES = [("orthogonal", pos, mp.Box(box)), ("slab (open z)", pos, mp.Box(box, boundary=[1, 1, 0])), ("cluster (open)", pos, mp.Box(box, boundary=[0, 0, 0]))]
res = {}
for tag, p, bx in CASES:
    for rep in range(2):
        s = mp.System(pos=p, box=bx)
        s.update_data(s.data.with_columns(type=ty[:len(p)]))
        if "--tertiary" in sys.argv:
            vel = np.random.default_rng(3).normal(0.0, 0.01, (len(p), 3))
            s.update_data(s.data.with_columns(vx=vel[:, 0], vy=vel[:, 1], vz=vel[:, 2], amass=np.full(len(p), 63.546)))
        for name, fn in CALLS:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            try:
                fn(s); torch.cuda.synchronize()
                res[(tag, name)] = (time.perf_counter() - t0) * 1e3
            except (ValueError, KeyError, TypeError, NotImplementedError) as e:  # a refusal (printed as nan, with the time it took to refuse)
                res[(tag, name)] = float("nan")
                print(f"# {tag}: {name} refused after {(time.perf_counter() - t0) * 1e3:.0f} ms: {str(e)[:110]}")
print(f"N = {len(pos)}" if "--sizes" not in sys.argv else "sizes")
for name, _ in CALLS:
    a = res[(CASES[0][0], name)]
    print(f"{name:32s} " + "   ".join(f"{tag} {res[(tag, name)]:8.2f} ms (x{res[(tag, name)] / a:5.2f})" for tag, _, _ in CASES))
