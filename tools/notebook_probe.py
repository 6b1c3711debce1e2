#!/usr/bin/env python
"""the published workflow of the reference's benchmark notebook, call by call: python tools/notebook_probe.py [fresh|reuse|both] [reps]
(4 M atoms: csp(12) + ackland_jones + structure_entropy(5.0, 0.2), on a new System or after build_neighbor(5.0, 50));
run it under `rocprofv3 --kernel-trace --stats` (tools/measure_r05.sh notebook) for the kernels behind each call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.devarray import HArray
from mdapy_amd.frame import Frame
from bench import A_CU

which = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)


def system(n):
    a = A_CU
    basis = torch.tensor([[0.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5]], dtype=torch.float64, device=dev) * a
    cols = []
    for k in range(3):
        shape = [1, 1, 1, 1]
        shape[k] = n
        comp = basis[:, k].view(1, 1, 1, 4) + (torch.arange(n, dtype=torch.float64, device=dev) * a).view(shape)
        cols.append(comp.expand(n, n, n, 4).reshape(-1).contiguous())
    return mp.System(data=Frame({"x": HArray(cols[0]), "y": HArray(cols[1]), "z": HArray(cols[2])}), box=mp.Box(np.diag([a * n] * 3)))


def flow(reuse):
    laps = []

    def lap(name, fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        laps.append((name, (time.perf_counter() - t0) * 1e3))

    box = {}
    lap("System", lambda: box.setdefault("s", system(100)))
    s = box["s"]
    if reuse:
        lap("build_neighbor(5.0, 50)", lambda: s.build_neighbor(rc=5.0, max_neigh=50))
    lap("csp(12)", lambda: s.cal_centro_symmetry_parameter(12))
    lap("ackland_jones", lambda: s.cal_ackland_jones_analysis())
    lap("structure_entropy(5.0, 0.2)", lambda: s.cal_structure_entropy(5.0, 0.2))
    return laps


for name in (("fresh", "reuse") if which == "both" else (which,)):
    best = None
    for rep in range(reps + 1):
        laps = flow(name == "reuse")
        if rep and (best is None or sum(t for _, t in laps) < sum(t for _, t in best)):
            best = laps
    print(f"{name}: total {sum(t for _, t in best):.2f} ms  " + "  ".join(f"{n} {t:.2f}" for n, t in best), flush=True)
