#!/usr/bin/env python
"""The calls the reference's users make, timed as they make them: cold and non-repeating.

(a) fresh process -> System(pos) -> build_neighbor, cal_common_neighbor_analysis, cal_centro_symmetry_parameter,
    cal_polyhedral_template_matching, cal_radial_distribution_function, each ONCE ("first"), then the same calls on a second
    System of the same atoms ("warm"; a System keeps one list, so every System builds its own).  Rattled fcc Cu, host numpy in.
(b) an NPT-like trajectory: 20 frames, every frame a new System with box +-0.5 % and N +-0.1 % (atoms removed at random), so
    every (N, grid) signature is new; build_neighbor(max_neigh=None... as cal_* calls it) + CNA per frame, against the same
    calls repeated on ONE frame.

    python tools/cold_path.py [cells=63] [what=apis,npt]
One size per process: "cold" means the process has launched nothing yet."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t_import = time.perf_counter()
import numpy as np
import torch
import mdapy_amd as mp
from mdapy_amd.build_lattice import lattice_positions
t_import = time.perf_counter() - t_import

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 63
what = sys.argv[2] if len(sys.argv) > 2 else "apis,npt"
A = 3.615
rc = 0.854 * A
pos0, box0 = lattice_positions("fcc", A, cells, cells, cells)
rng = np.random.default_rng(11)
pos0 = pos0 + rng.normal(0.0, 0.05, pos0.shape)
N = len(pos0)


def lap(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, r


CALLS = [
    ("System(pos, box)", None),
    ("build_neighbor(rc)", lambda s: s.build_neighbor(rc)),
    ("cal_common_neighbor_analysis(rc)", lambda s: s.cal_common_neighbor_analysis(rc)),
    ("cal_centro_symmetry_parameter(12)", lambda s: s.cal_centro_symmetry_parameter(12)),
    ("cal_polyhedral_template_matching()", lambda s: s.cal_polyhedral_template_matching()),
    ("cal_radial_distribution_function(5.0, 200)", lambda s: s.cal_radial_distribution_function(5.0, 200)),
]

print(f"# {N} atoms ({cells}^3 fcc cells, sigma 0.05), import torch + mdapy_amd {t_import:.2f} s")
if "apis" in what:
    t_ctx, _ = lap(lambda: torch.zeros(1, device="cuda"))
    print(f"# first touch of the device (context, torch allocator): {t_ctx:.1f} ms")
    from mdapy_amd import _lib
    t_lib, _ = lap(_lib.lib)
    print(f"# library load + its code objects on the device (mdh_warm; MDAPY_HIP_WARM={os.environ.get('MDAPY_HIP_WARM', '1')}; with 0 every code object is loaded by the first launch that needs it): {t_lib:.1f} ms")
    rounds = []
    for rnd in range(3):
        row = []
        ms, s = lap(lambda: mp.System(pos=pos0, box=box0))
        row.append(ms)
        for name, fn in CALLS[1:]:
            ms, _ = lap(lambda: fn(s))
            row.append(ms)
        rounds.append(row)
        del s
    print(f"{'call':46s} {'first':>9s} {'second':>9s} {'third':>9s}  first/third")
    for k, (name, _) in enumerate(CALLS):
        a, b, c = rounds[0][k], rounds[1][k], rounds[2][k]
        print(f"{name:46s} {a:9.2f} {b:9.2f} {c:9.2f}  {a / c:6.2f}x")

if "npt" in what:
    frames = []
    for f in range(20):
        scale = 1.0 + 0.005 * np.sin(1.3 * f + 0.4)
        keep = rng.random(N) >= 0.001 * (1.0 + np.cos(0.9 * f))
        frames.append((np.ascontiguousarray(pos0[keep] * scale), np.asarray(box0) * scale))

    def one(p, b):
        s = mp.System(pos=p, box=b)
        s.build_neighbor(rc, max_neigh=16)
        s.cal_common_neighbor_analysis(rc)
        return s.data["cna"]

    # warm: the same frame again and again
    for _ in range(3):
        one(*frames[0])
    warm = [lap(lambda: one(*frames[0]))[0] for _ in range(10)]
    seq = [lap(lambda: one(p, b))[0] for p, b in frames]
    sig = len({(len(p), tuple(np.floor(np.diag(b) / rc).astype(int))) for p, b in frames})
    print(f"NPT-like sequence, {len(frames)} frames, {sig} distinct (N, grid) signatures: System + build_neighbor(M=16) + CNA per frame")
    print(f"   same frame repeated: median {np.median(warm):.2f} ms (min {min(warm):.2f})")
    print(f"   trajectory frames:   median {np.median(seq):.2f} ms (min {min(seq):.2f}, max {max(seq):.2f}, first {seq[0]:.2f})  = {np.median(seq) / np.median(warm):.3f}x warm")
    print("   per frame:", " ".join(f"{v:.1f}" for v in seq))
