#!/usr/bin/env python
"""Step time against system size: where the fixed cost of a neighbor + CNA step sits.

For FCC Cu cubes of n^3 cells (4 n^3 atoms) the undivided step of bench.py (build_neighbor M = 16 + fixed CNA through the C
ABI, HBM-resident positions); for slabs of a box split 8 ways along x the decomposed step of rank 1 in loop-back
(tools/_loopback.py: the product's exchange code, the wire replaced by a device copy).  Prints one row per size, the library's
own per-range kernel times, and a least-squares fit t = t0 + N / rate over the undivided rows.

    python tools/step_vs_n.py                  # the table
    python tools/step_vs_n.py --only 10        # one size, many steps: the command a rocprofv3 --kernel-trace run wraps
"""
import argparse, ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import mdapy_amd as mp
from mdapy_amd import _cna, _lib, _neighbor
from mdapy_amd import distributed as D
from bench import slab_positions, A_CU, RC
import _loopback

p = argparse.ArgumentParser()
p.add_argument("--only", type=int, default=0)
p.add_argument("--sizes", default="10,20,40,64,68,96,136")
p.add_argument("--slabs", default="64,96,136", help="boxes of n^3 cells split 8 ways along x (n divisible by 8)")
p.add_argument("--json", default="")
args = p.parse_args()
L = _lib.lib()
dev = torch.device("cuda", 0)
M = 16


def ranges(fn, reps):
    L.mdh_prof_reset(); L.mdh_prof_enable(1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(); L.mdh_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    L.mdh_prof_report(buf, len(buf))
    return {l.split()[0]: round(float(l.split()[2]) / int(l.split()[1]) * 1e3, 1) for l in buf.value.decode().strip().splitlines()}


def timed(fn, n_atoms):
    reps = int(min(400, max(20, 3e8 / max(n_atoms, 1) / 30)))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6, reps


def undivided(n):
    x, y, z, _ = slab_positions(torch, dev, n, 0, 0.0)
    N = int(x.shape[0])
    box = mp.Box(np.diag([A_CU * n] * 3))
    bx = (box.box, box.origin, box.boundary)
    v = torch.empty((N, M), dtype=torch.int32, device=dev); d = torch.empty((N, M), dtype=torch.float64, device=dev)
    nn = torch.empty((N,), dtype=torch.int32, device=dev); pat = torch.empty((N,), dtype=torch.int32, device=dev)

    def step():
        pat.zero_()
        _neighbor.build_neighbor(x, y, z, *bx, RC, v, d, nn, 1, fill_pads=True)
        _cna.fcna(x, y, z, *bx, v, nn, pat, RC, 1)

    us, reps = timed(step, N)
    plan = (ctypes.c_int * 8)(); L.mdh_debug_neighbor_plan(plan)
    ok = bool((nn == 12).all()) and bool((pat == 1).all())
    return dict(kind="undivided", cells=n, atoms=N, us=round(us, 1), reps=reps, ok=ok, plan=list(plan)[:5], ranges_us=ranges(step, 10)), step


def slab(n, world=8, rank=1):
    cx = n // world
    x, y, z, gid = slab_positions(torch, dev, n, rank, 0.0, cells_x=cx)
    N = int(x.shape[0])
    box = mp.Box(np.diag([A_CU * n] * 3))
    dec = D.SlabDecomposition(box, rank, world, axis=0)
    _loopback.install(dec, A_CU * cx, N)
    x, y, z, gid = (dec.with_room(a, 0.5 if N < 200000 else 0.2) for a in (x, y, z, gid))
    out = {}

    def step():
        out["r"] = D.neighbor_cna_step(dec, x, y, z, gid, RC, M)

    def step_pipe():
        out["r"] = D.neighbor_cna_step(dec, x, y, z, gid, RC, M, next_frame=(x, y, z, gid))

    us, reps = timed(step, N)
    dom, _, _, nn, pat = out["r"]
    ok = bool((nn[dom.owned] == 12).all()) and bool((pat[dom.owned] == 1).all())
    us_p, _ = timed(step_pipe, N)
    dec._drop_pending(); torch.cuda.synchronize()
    return dict(kind=f"slab 1 of {world} (loop-back)", cells=f"{cx}x{n}x{n}", atoms=N, ghosts=int(dom.x.shape[0]) - N, us=round(us, 1),
                us_prefetched=round(us_p, 1), reps=reps, ok=ok, ranges_us=ranges(step, 10)), step


if args.only:
    row, step = undivided(args.only)
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    print(json.dumps(row))
    sys.exit(0)

rows = []
for n in [int(s) for s in args.sizes.split(",") if s]:
    rows.append(undivided(n)[0]); print(json.dumps(rows[-1]), flush=True)
und = [r for r in rows if r["kind"] == "undivided"]
if len(und) >= 3:
    A = np.array([[1.0, r["atoms"]] for r in und]); b = np.array([r["us"] for r in und])
    # weight by 1/t: the small sizes decide t0, the large ones the rate
    w = 1.0 / b
    t0, inv = np.linalg.lstsq(A * w[:, None], b * w, rcond=None)[0]
    fit = dict(t0_us=round(float(t0), 1), atoms_per_us=round(1.0 / float(inv), 1))
    print(json.dumps({"fit t = t0 + N / rate (undivided)": fit}), flush=True)
for n in [int(s) for s in args.slabs.split(",") if s]:
    rows.append(slab(n)[0]); print(json.dumps(rows[-1]), flush=True)
full = {r["cells"]: r for r in und}
for r in rows:
    if r["kind"].startswith("slab"):
        n = int(str(r["cells"]).split("x")[1])
        if n in full:
            r["vs_undivided_over_8"] = round(r["us"] / (full[n]["us"] / 8.0), 3)
            r["prefetched_vs_undivided_over_8"] = round(r["us_prefetched"] / (full[n]["us"] / 8.0), 3)
            print(json.dumps({"slab": r["cells"], "step / (undivided / 8)": r["vs_undivided_over_8"], "prefetched": r["prefetched_vs_undivided_over_8"]}))
if args.json:
    json.dump(rows, open(args.json, "w"), indent=1)
