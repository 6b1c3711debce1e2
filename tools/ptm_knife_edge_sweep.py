#!/usr/bin/env python
"""How often the PTM restatement and the reference's library disagree on the systems of tests/fuzz_parity.py, and why: the product's
PTM sources compiled for the HOST (tests/native/ptm_host.cpp, the same ptm_core.hpp the GPU runs; GPU == host is a test of its own)
against oracle/_ref on every seed of a range whose draw takes the fuzz's PTM check.  CPU only; test infrastructure (uses the oracle).
    python tools/ptm_knife_edge_sweep.py <first_seed> <count> [processes]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def one(seed):
    from fuzz_parity import draw
    from oracle import oracle as O
    import test_ptm_host as TH
    host = C.CDLL(TH.SO); host.ptmh_init.restype = C.c_char_p
    if host.ptmh_init() != b"": return (seed, "init", 0, [])
    s = draw(seed)
    if s["unwrapped"] or s["sigma"] == 0.0 or not (len(s["pos"]) >= 20 or all(s["bnd"])): return None
    if np.any(s["box"] - np.triu(s["box"]) * 0 != s["box"]): pass
    r2 = np.random.default_rng(seed + 13)
    structure = str(r2.choice(["default", "all", "fcc-hcp-bcc-ico-sc", "fcc-hcp-bcc", "dcub-dhex", "bcc,sc", "graphene-fcc", "ico"]))
    types = r2.integers(1, 4, len(s["pos"])).astype(np.int32) if r2.random() < 0.4 else None
    thr = float(r2.choice([0.0, 0.05, 0.1, 0.3]))
    pos, box, bd = s["pos"] - s["origin"], s["box"], np.array([int(v) for v in s["bnd"]], np.int32)
    N = len(pos); x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3)); k = min(18, N - 1)
    idx, dist = np.zeros((N, k), np.int32), np.zeros((N, k))
    O.knn(x, y, z, box, np.zeros(3), bd, k, idx, dist, 1)
    out_r, ind_r = np.zeros((N, 8)), np.zeros((N, 18), np.int32); cached = np.zeros(N, np.uint64)
    O.get_ptm(structure, x, y, z, box, np.zeros(3), bd, idx, types, thr, out_r, ind_r, cached=cached)
    out_m, ind_m = np.zeros((N, 8)), np.zeros((N, 18), np.int32); order = np.zeros((N, 18), np.int8)
    b9 = np.ascontiguousarray(box, dtype=np.float64).reshape(9); P = TH._ptr
    if host.ptmh_run(P(x), P(y), P(z), C.c_int64(N), P(b9), P(bd), P(idx), C.c_int64(k), P(types), TH._flags(structure), C.c_double(thr), P(out_m), P(ind_m), P(order)) != 0:
        return (seed, "run", N, [])
    dq = np.minimum(np.abs(out_m[:, 4:] - out_r[:, 4:]).max(1), np.abs(out_m[:, 4:] + out_r[:, 4:]).max(1))
    bad = np.where((out_m[:, 0] != out_r[:, 0]) | (out_m[:, 1] != out_r[:, 1]) | (ind_m != ind_r).any(1) | (np.abs(out_m[:, 2:4] - out_r[:, 2:4]).max(1) > 1e-6) | (dq > 1e-6))[0]
    why = []
    for a in bad[:4]:  # is it the neighbour ORDER (the Voronoi cell's slivers) that differs?
        o19 = np.zeros(19, np.int8); O.ref_lib().ref_ptm_decode_order(C.c_uint64(int(cached[a])), o19.ctypes.data_as(C.c_void_p))
        ref_order = [int(v) - 1 for v in o19[1:k + 1]]
        why.append((int(a), "order" if ref_order != order[a][:k].tolist() else "same order", float(out_r[a, 0]), float(out_m[a, 0])))
    return (seed, f"{s['kind']} {structure}", N, why)


if __name__ == "__main__":
    import multiprocessing as mp
    first, count = int(sys.argv[1]), int(sys.argv[2]); procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    atoms = checked = 0; found = []
    with mp.Pool(procs) as pool:
        for r in pool.imap_unordered(one, range(first, first + count), chunksize=8):
            if r is None: continue
            checked += 1; atoms += r[2]
            if r[3]: found.append(r); print("DIFF", r, flush=True)
    print(f"seeds {first}..{first + count - 1}: {checked} seeds took the PTM check, {atoms} atoms, {len(found)} seeds with a difference ({sum(len(f[3]) for f in found)} atoms listed)")
