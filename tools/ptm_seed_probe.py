#!/usr/bin/env python
"""One fuzz seed's PTM check in detail (tests/fuzz_parity.py draws the system and the structure string): which atoms differ from the
reference's library (oracle/_ref) and how.  python tools/ptm_seed_probe.py <seed> [...]   PROBE_LIB=<path>: another build of the library"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mdapy_amd import _lib
if os.environ.get("PROBE_LIB"):  # (an older build: only the entries it has)
    import ctypes, torch  # noqa: F401
    _lib.LIB_PATH = os.path.abspath(os.environ["PROBE_LIB"])
    _L0 = ctypes.CDLL(_lib.LIB_PATH)
    _lib._SIGNATURES = {k: v for k, v in _lib._SIGNATURES.items() if hasattr(_L0, k)}
import test_gpu_parity as T
from fuzz_parity import draw
for seed in [int(a) for a in sys.argv[1:]]:
    s = draw(seed)
    r2 = np.random.default_rng(s["seed"] + 13)
    structure = str(r2.choice(["default", "all", "fcc-hcp-bcc-ico-sc", "fcc-hcp-bcc", "dcub-dhex", "bcc,sc", "graphene-fcc", "ico"]))
    types = r2.integers(1, 4, len(s["pos"])).astype(np.int32) if r2.random() < 0.4 else None
    thr = float(r2.choice([0.0, 0.05, 0.1, 0.3]))
    pos, box, bd = s["pos"] - s["origin"], s["box"], np.array([int(v) for v in s["bnd"]], np.int32)
    N = len(pos); x, y, z = T._xyz(pos); k = min(18, N - 1)
    idx, dist = np.zeros((N, k), np.int32), np.zeros((N, k))
    T.O.knn(x, y, z, box, T.ORG0, bd, k, idx, dist, 4)
    out_r, ind_r = np.zeros((N, 8)), np.zeros((N, 18), np.int32)
    T.O.get_ptm(structure, x, y, z, box, T.ORG0, bd, idx, types, thr, out_r, ind_r)
    out_g, ind_g = np.full((N, 8), 7.0), np.full((N, 18), 7, np.int32)
    T._ptm.get_ptm(structure, x, y, z, box, T.ORG0, bd, idx, types, thr, out_g, ind_g)
    dq = np.minimum(np.abs(out_g[:, 4:] - out_r[:, 4:]).max(1), np.abs(out_g[:, 4:] + out_r[:, 4:]).max(1))
    bad = np.where((out_g[:, 0] != out_r[:, 0]) | (out_g[:, 1] != out_r[:, 1]) | (ind_g != ind_r).any(1) | (np.abs(out_g[:, 2:4] - out_r[:, 2:4]).max(1) > 1e-6) | (dq > 1e-6))[0]
    print(f"seed {seed}: kind {s['kind']} sigma {s['sigma']} tri {s['tri']} bnd {s['bnd']} N {N} structure {structure} types {types is not None} thr {thr}: {len(bad)} atoms differ", flush=True)
    for a in bad[:6]:
        print("  atom", a, "type g/r", out_g[a, 0], out_r[a, 0], "rmsd", out_g[a, 2], out_r[a, 2], "scale", out_g[a, 3], out_r[a, 3], "dq", dq[a])
        print("    ind g", ind_g[a].tolist()); print("    ind r", ind_r[a].tolist())
        print("    q g", out_g[a, 4:].tolist()); print("    q r", out_r[a, 4:].tolist())
        print("    knn dist", np.round(dist[a], 6).tolist())
