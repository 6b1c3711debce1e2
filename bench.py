#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json: atoms/sec for neighbor-build + CNA on FCC Cu.

A "step" is one pass of the hot path over the whole (synthetic) system, positions already resident in HBM: cell-list
neighbor build (rc = 0.854 a, max_neigh = 16: the `System.cal_common_neighbor_analysis(rc)` configuration the reference's
own CNA tests and SURVEY.md 6 use) and fixed-cutoff CNA, through the C ABI of libmdapy_amd.so with device pointers — since
round 5 as ONE call (mdh_build_neighbor_fcna: the lists as build_neighbor leaves them AND the labels as fcna leaves them, the
label of a centre worked out inside the tile kernel while its neighbours are still staged in LDS; what
`System.cal_common_neighbor_analysis(rc)` runs when the system has no list yet).  `extra.two_calls` times the same step as the
two calls it was until then (mdh_build_neighbor, then mdh_fcna), with the plain neighbour kernel's own roofline figure; `extra.records_path`
the one call with the atoms gathered into cell-sorted records first (rounds 1-5; from round 6 on a build of spatially ordered input keeps no
sorted copy and the tile kernel reads the atoms through the cell-sorted id list: a faster step, a slower tile kernel).  At N GPUs every rank owns a 136^3-cell slab of a (136 N) x 136 x 136-cell box (weak scaling, the default) — or,
with --scaling strong, a 136/N-cell slab of the one 136^3 box — and exchanges a one-cutoff ghost halo with its two ring
neighbours over RCCL each step.

Prints ONE JSON line (rank 0):
* `roofline` is for the dominant kernel of the step (the tile kernel, `k_neighbor` = everything between the cell grid
  and the CNA's to-do kernel: it writes the lists AND the labels), timed with HIP events recorded inside the library on the
  launch stream; its algorithmic bytes are what it must move — 24 B of position read, 4 B count, 12 M B of rows and the 4 B label
  written per atom (SURVEY.md 8d counts the list a second time for a CNA that reads it back: 60 + 16 M; the fused kernel does not).  `traffic` is the
  HBM byte count of that kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) taken by THIS run when rocprofv3 is
  on PATH (`traffic_source: "live"`), else the committed measurement under profiles/ (`"committed"`); FETCH_SIZE is
  doubled as MI355X_MICROARCH.md prescribes for gfx950 (the raw sum is reported beside it).
* `extra` times, on the same box, other ways the step is reached and other inputs: `max_neigh=None` (exact-width rows:
  counting pass + build on one cell grid with the labels made in the build, mdh_build_neighbor_exact_fcna); thermally rattled lattices (sigma = 0.05 and 0.20 A) and a
  polycrystal of the headline's size (BASELINE config 3's construction), each with the fraction of atoms the CNA finished in
  double precision (`todo_fraction`) and the tiles that went to the neighbour kernel's slice pass; `strong_1of8`: the slab of
  rank 1 of 8 of the SAME box with the exchange in loop-back on this one GPU (the wire is not measured).
* `cpu_baseline` is the CPU oracle (a parity-checked port of the reference's OpenMP C++, oracle/mdapy_oracle.c) timed on
  this box's host cores on the headline's own input; a thread sweep on that same input picks the thread count.
* `--scaling strong` (N > 1): the ONE --cells^3 box cut into N slabs along x instead of one --cells^3 slab per rank.
"""
import argparse
import ctypes
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

A_CU = 3.615
RC = 0.854 * A_CU
METRIC = "atoms/sec for neighbor+CNA on 10M-atom FCC Cu; 1/2/4/8-GPU scaling"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# the OpenMP port against the reference's own C++ on identical input and cores (1 000 188-atom FCC Cu, neighbor M = 16 + fixed
# CNA, 8 threads, build container): port 0.315 s, reference 0.40 s
PORT_OVER_REFERENCE = 0.40 / 0.315


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--cells", type=int, default=136, help="FCC cells per axis per GPU (136 -> 10 061 824 atoms)")
    p.add_argument("--max-neigh", type=int, default=16)
    p.add_argument("--sigma", type=float, default=0.0, help="optional thermal rattle (A) of the lattice")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra", action="store_true", help="skip the max_neigh=None / rattled-lattice timings")
    p.add_argument("--no-pmc", action="store_true", help="do not take the live rocprofv3 PMC passes for roofline.traffic")
    p.add_argument("--cpu-cells", type=int, default=63, help="cells per axis of the CPU-baseline thread sweep (63 -> 1 000 188 atoms)")
    p.add_argument("--cpu-full-cells", type=int, default=0, help="cells per axis of the CPU-baseline run proper (0: --cells, the headline input)")
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                   help="N > 1: weak = a --cells^3 slab per rank (default); strong = the ONE --cells^3 box cut into N slabs along x")
    p.add_argument("--pmc-child", choices=["fetch", "write"], help=argparse.SUPPRESS)
    p.add_argument("--cpu-child", action="store_true", help=argparse.SUPPRESS)
    return p.parse_args()


def slab_positions(torch, dev, cells, rank, sigma, cells_x=None, a=None):
    """FCC Cu slab of rank `rank`: cells ix in [cx*rank, cx*(rank+1)) x cells x cells (cx = cells unless cells_x is given: the
    slab of a box split along x), cell-major order, basis innermost — same expression as build_crystal / repeat_cell
    (basis@cell + (ix*a1 + iy*a2 + iz*a3))."""
    a = A_CU if a is None else a
    cx = cells if cells_x is None else cells_x
    basis = torch.tensor([[0.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5]], dtype=torch.float64, device=dev) * a
    ix = torch.arange(cx * rank, cx * (rank + 1), dtype=torch.float64, device=dev) * a
    iy = torch.arange(cells, dtype=torch.float64, device=dev) * a
    out = []
    for k, s in enumerate((ix, iy, iy)):
        shape = [1, 1, 1, 1]
        shape[k] = cx if k == 0 else cells
        comp = basis[:, k].view(1, 1, 1, 4) + s.view(shape)
        out.append(comp.expand(cx, cells, cells, 4).reshape(-1).contiguous())
    if sigma > 0:
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
        out = [c + torch.randn(c.shape, generator=g, dtype=torch.float64, device=dev) * sigma for c in out]
    n = cx * cells * cells * 4
    gid = torch.arange(n * rank, n * (rank + 1), dtype=torch.int64, device=dev)
    return out[0], out[1], out[2], gid


def analyses_extra(torch, dev, mp, cells):
    from mdapy_amd.devarray import HArray
    from mdapy_amd.frame import Frame

    def forget_candidates():
        # a k-nearest search leaves the candidate rows of its cutoff build with the position columns for the next search of the same
        # System (knn.py); a timed search must be a whole one, not the second half of the warm-up call's
        from mdapy_amd import knn as knn_mod

        knn_mod._bags.clear()

    def lap(n, alg_bytes, fn, fresh=True):
        fn()  # (first call: sizes the scratch cache, loads code objects)
        if fresh:
            forget_candidates()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"ms": dt * 1e3, "atoms_per_s": n / dt, "alg_B_per_atom": alg_bytes, "frac_of_hbm_peak": alg_bytes * n / dt / 1e9 / HBM_PEAK_GBS}, out

    found = {}
    # ---- config 2
    x, y, z, _ = slab_positions(torch, dev, cells, 0, 0.05)
    n = int(x.shape[0])
    s = mp.System(data=Frame({"x": HArray(x), "y": HArray(y), "z": HArray(z)}), box=mp.Box(np.diag([A_CU * cells] * 3)))
    c2 = {"atoms": n, "input": f"{cells}^3-cell FCC Cu rattled by N(0, 0.05 A)"}
    c2["knn18"], _ = lap(n, 24 + 18 * 12, lambda: s.build_nearest_neighbor(18))
    c2["ptm_fcc_hcp_bcc"], _ = lap(n, 24 + 18 * 4 + 8 * 8 + 18 * 4, lambda: s.cal_polyhedral_template_matching("fcc-hcp-bcc", return_rmsd=True))
    c2["steinhardt_q4_q6_nnn12"], _ = lap(n, 24 + 12 * 12 + 16, lambda: s.cal_steinhardt_bond_orientation([4, 6], nnn=12))
    c2["csp12"], _ = lap(n, 24 + 12 * 4 + 8, lambda: s.cal_centro_symmetry_parameter(12))
    # the adaptive CNA's own 14-nearest search borrows the candidate rows the centro-symmetry call just left with the System's
    # position columns (what a user who calls the two on one System gets); `adaptive_cna_alone`: with nothing to borrow
    c2["adaptive_cna"], _ = lap(n, 24 + 14 * 4 + 4, lambda: s.cal_common_neighbor_analysis(), fresh=False)
    c2["adaptive_cna_alone"], _ = lap(n, 24 + 14 * 4 + 4, lambda: s.cal_common_neighbor_analysis())
    c2["knn_note"] = ("knn18, csp12, adaptive_cna_alone: whole searches (the candidate rows a warm-up call left behind are dropped first); "
                      "adaptive_cna: after csp12 on the same System, its search reads csp12's candidate rows (mdh_knn_keyed_rows)")
    lab = np.bincount(s.data["ptm"].to_numpy(), minlength=4).tolist()
    c2["ptm_fcc_fraction"] = lab[1] / n
    c2["total_ms_knn_ptm_steinhardt"] = c2["knn18"]["ms"] + c2["ptm_fcc_hcp_bcc"]["ms"] + c2["steinhardt_q4_q6_nnn12"]["ms"]
    c2["dominant"] = "ptm_fcc_hcp_bcc"
    found["config2"] = c2
    del s, x, y, z
    # ---- config 4
    gc = cells - 1 if cells > 8 else cells
    x, y, z, _ = slab_positions(torch, dev, gc, 0, 0.35, a=4.0)
    n = int(x.shape[0])
    gen = torch.Generator(device=dev); gen.manual_seed(42)
    ty = (torch.rand(n, device=dev, generator=gen) < 0.36).to(torch.int32) + 1  # 1 = Cu (64 %), 2 = Zr (36 %)
    s = mp.System(data=Frame({"x": HArray(x), "y": HArray(y), "z": HArray(z), "type": HArray(ty)}), box=mp.Box(np.diag([4.0 * gc] * 3)))
    c4 = {"atoms": n, "input": f"{gc}^3 x 4 sites of fcc a = 4.0 A displaced by N(0, 0.35 A), Cu64Zr36 at random"}
    c4["rdf_partial_rc8_200bins_streaming"], g = lap(n, 28, lambda: s.cal_radial_distribution_function(8.0, nbin=200, streaming=True))
    c4["build_neighbor_rc3.6"], _ = lap(n, 28 + 12 * int(1), lambda: s.build_neighbor(3.6))
    c4["build_neighbor_rc3.6"]["row_width"] = int(s.verlet_list.shape[1])
    c4["build_neighbor_rc3.6"]["alg_B_per_atom"] = 28 + 12 * int(s.verlet_list.shape[1])
    c4["build_neighbor_rc3.6"]["frac_of_hbm_peak"] = c4["build_neighbor_rc3.6"]["alg_B_per_atom"] * n / (c4["build_neighbor_rc3.6"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    c4["warren_cowley_rc3.6"], w = lap(n, 8 + 4 * int(s.verlet_list.shape[1]), lambda: s.cal_warren_cowley_parameter(3.6))
    c4["g_total_tail"] = float(np.asarray(g.g_total)[-20:].mean())  # -> 1 for an uncorrelated frame
    c4["wcp_max_abs"] = float(np.abs(np.asarray(w.WCP)).max())     # -> 0 for species placed at random
    c4["total_ms_rdf_wcp"] = c4["rdf_partial_rc8_200bins_streaming"]["ms"] + c4["warren_cowley_rc3.6"]["ms"]
    c4["dominant"] = "rdf_partial_rc8_200bins_streaming"
    found["config4"] = c4
    return found


def notebook_calls(torch, dev, mp):
    from mdapy_amd.devarray import HArray
    from mdapy_amd.frame import Frame

    def system(nx, ny, nz):
        a = A_CU
        basis = torch.tensor([[0.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5]], dtype=torch.float64, device=dev) * a
        cols = []
        for k, n in enumerate((nx, ny, nz)):
            shape = [1, 1, 1, 1]
            shape[k] = n
            comp = basis[:, k].view(1, 1, 1, 4) + (torch.arange(n, dtype=torch.float64, device=dev) * a).view(shape)
            cols.append(comp.expand(nx, ny, nz, 4).reshape(-1).contiguous())
        return mp.System(data=Frame({"x": HArray(cols[0]), "y": HArray(cols[1]), "z": HArray(cols[2])}), box=mp.Box(np.diag([a * nx, a * ny, a * nz])))

    def lap(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    out = {}
    s = system(85, 100, 100)  # the largest system of the benchmark notebook: 3.4 M atoms
    n = s.N
    t = lap(lambda: s.build_neighbor(5.0, max_neigh=50))
    out["build_neighbor(5.0, max_neigh=50), 3.4 M atoms"] = {"ms": t, "atoms_per_s": n / (t * 1e-3), "reference_published": "0.4 s per 10^6 atoms (1.36 s at this size)",
                                                             "ratio_to_published": 0.4 * n / 1e6 / (t * 1e-3)}
    t = lap(lambda: s.build_nearest_neighbor(12))
    out["build_nearest_neighbor(12), 3.4 M atoms"] = {"ms": t, "atoms_per_s": n / (t * 1e-3), "reference_published": "0.2 s per 10^6 atoms (0.68 s at this size)",
                                                      "ratio_to_published": 0.2 * n / 1e6 / (t * 1e-3)}
    del s

    def three(sysm):
        sysm.cal_centro_symmetry_parameter(12)
        sysm.cal_ackland_jones_analysis()
        sysm.cal_structure_entropy(5.0, 0.2)

    def fresh():
        three(system(100, 100, 100))

    t = lap(fresh)
    out["4 M atoms: csp(12) + ackland_jones + structure_entropy(5.0, 0.2) on a new System"] = {
        "ms": t, "reference_published": "8.72 s wall", "ratio_to_published": 8.72 / (t * 1e-3)}

    def reuse():
        sysm = system(100, 100, 100)
        sysm.build_neighbor(rc=5.0, max_neigh=50)
        three(sysm)

    t = lap(reuse)
    out["the same after build_neighbor(rc=5.0, max_neigh=50) (list reuse), build included"] = {
        "ms": t, "reference_published": "6.39 s wall (build not included)", "ratio_to_published": 6.39 / (t * 1e-3)}
    return out


def cpu_baseline_child(args):
    """the timed sample itself (a process of its own: OpenMP reads its binding when the runtime starts, and the parent has
    long started one with torch): a thread sweep on the HEADLINE input (--cpu-full-cells, 10 061 824 atoms) picks the thread
    count, the figure is the best of <= 3 runs at that count and its two neighbours"""
    from mdapy_amd.build_lattice import lattice_positions
    from oracle import oracle as O

    M = args.max_neigh
    org, bnd = np.zeros(3), np.array([1, 1, 1], np.int32)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def setup(n):
        pos, box = lattice_positions("fcc", A_CU, n, n, n)
        x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
        N = len(x)
        del pos
        bufs = (np.full((N, M), -1, np.int32), np.full((N, M), RC + 1.0), np.zeros(N, np.int32), np.zeros(N, np.int32))
        return x, y, z, box, N, bufs

    def one(inp, threads):
        x, y, z, box, N, (v, d, nn, pat) = inp
        v[...] = -1; d[...] = RC + 1.0; nn[...] = 0; pat[...] = 0  # what the reference's Python does before the call (neighbor.py:125-129)
        t0 = time.perf_counter()
        O.build_neighbor(x, y, z, box, org, bnd, RC, v, d, nn, threads)
        O.fcna(x, y, z, box, org, bnd, v, nn, pat, RC, threads)
        dt = time.perf_counter() - t0
        assert int((pat == 1).sum()) == N
        return dt

    t_all = time.perf_counter()
    full_cells = args.cpu_full_cells or args.cells
    grid = sorted({t for t in (8, 16, 32, 64, 128, 256, cores) if t <= cores})
    # the sweep runs on the input the figure is quoted on (round 5's sweep on a 1 M-atom lattice mis-predicted the 10 M-atom
    # optimum by 1.6 x: the small system lives in the caches): one run per thread count, from the middle of the grid outwards
    # while the time allows, then two more runs at the best count and its two neighbours
    big = setup(full_cells)
    one(big, min(cores, 32))  # first touch of the pages
    mid = min(range(len(grid)), key=lambda j: abs(grid[j] - 32))
    order = sorted(range(len(grid)), key=lambda j: (abs(j - mid), j))
    sweep = {}
    for j in order:
        sweep[grid[j]] = one(big, grid[j])
        if time.perf_counter() - t_all > 16.0 and len(sweep) >= 3:
            break
    best0 = min(sweep, key=sweep.get)
    k = grid.index(best0)
    full = {}
    for threads in [grid[j] for j in (k, k - 1, k + 1) if 0 <= j < len(grid)]:
        runs = [sweep[threads]] if threads in sweep else []
        for _ in range(2):
            if time.perf_counter() - t_all > 30.0 and runs:
                break
            runs.append(one(big, threads))
        full[threads] = min(runs)
    out = {"N_sweep": big[4], "cores": cores, "sweep": {str(k): v for k, v in sweep.items()}, "N": big[4], "full": {str(k): v for k, v in full.items()}}
    nodes = 0
    try:
        nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        pass
    out["numa_nodes"] = nodes
    print(json.dumps(out))


def cpu_baseline(args):
    """the OpenMP port of the oracle on the host cores: threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores), the thread
    count picked by a sweep on the headline's own input, the figure taken there too (best of <= 3 runs)"""
    import subprocess

    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_DYNAMIC="false")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-child", "--cpu-cells", str(args.cpu_cells), "--max-neigh", str(args.max_neigh),
           "--cells", str(args.cells), "--cpu-full-cells", str(args.cpu_full_cells)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        raise RuntimeError("cpu baseline child failed: " + out.stderr[-2000:])
    got = json.loads(out.stdout.strip().splitlines()[-1])
    N, cores, M = got["N"], got["cores"], args.max_neigh
    full = {int(k): v for k, v in got["full"].items()}
    sweep = {int(k): v for k, v in got["sweep"].items()}
    best_threads = min(full, key=full.get)
    best = full[best_threads]
    return {"value": N / best, "unit": "atoms/s", "cores": cores, "threads": best_threads, "kind": "port",
            "sample": f"{N}-atom FCC Cu (the headline input), neighbor(rc={RC:.5f}, max_neigh={M}) + fixed CNA, OpenMP oracle port "
                      f"(oracle/mdapy_oracle.c), best of <= 3 runs at {sorted(full)} threads (picked by a sweep {sorted(sweep)} on "
                      f"the same {got['N_sweep']} atoms), {cores} host cores, {got.get('numa_nodes', 0)} NUMA node(s)",
            "placement": "OMP_PROC_BIND=close OMP_PLACES=cores OMP_DYNAMIC=false, a process of its own",
            "reference_estimate": {"value": N / best / PORT_OVER_REFERENCE, "unit": "atoms/s",
                                   "how": f"port / {PORT_OVER_REFERENCE:.2f}: on the build container's 8 cores the port runs 0.315 s where the "
                                          "reference's own C++ runs 0.40 s (1 000 188 atoms, BASELINE.md 2)"},
            "threads_full_atoms_per_s": {str(k): N / v for k, v in full.items()},
            "threads_sweep_atoms_per_s": {str(k): got["N_sweep"] / v for k, v in sweep.items()}}


# ---------------------------------------------------------------------------------------------------------------------
# HBM traffic of the neighbor kernel: two rocprofv3 PMC passes over a child process that runs two builds
# ---------------------------------------------------------------------------------------------------------------------
def pmc_child(args):
    import torch

    import mdapy_amd as mp
    from mdapy_amd import _neighbor

    dev = torch.device("cuda", 0)
    x, y, z, _ = slab_positions(torch, dev, args.cells, 0, args.sigma)
    n, M = int(x.shape[0]), args.max_neigh
    box = mp.Box(np.diag([A_CU * args.cells] * 3))
    verlet = torch.empty((n, M), dtype=torch.int32, device=dev); dist = torch.empty((n, M), dtype=torch.float64, device=dev)
    nn = torch.empty((n,), dtype=torch.int32, device=dev)
    pat = torch.zeros((n,), dtype=torch.int32, device=dev)
    for _ in range(2):
        _neighbor.build_neighbor_fcna(x, y, z, box.box, box.origin, box.boundary, RC, verlet, dist, nn, pat, 1, fill_pads=True)
    torch.cuda.synchronize()


def live_traffic(args):
    """(fetch_bytes_raw, write_bytes) per call of the neighbor kernel, or None"""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    import csv

    got = {}
    for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        out = tempfile.mkdtemp(prefix="mdh_pmc_", dir="/tmp")
        cmd = ["timeout", "-s", "KILL", "180", exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", which, "--cells", str(args.cells), "--max-neigh", str(args.max_neigh),
               "--sigma", str(args.sigma)]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
            files = glob.glob(os.path.join(out, "**", "p_counter_collection.csv"), recursive=True)
            if not files:
                return None
            per_dispatch = {}
            for row in csv.DictReader(open(files[0])):
                if "k_neighbor_lane" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    per_dispatch[row["Dispatch_Id"]] = per_dispatch.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per_dispatch:
                return None
            # two launches of the kernel per build (tiles, then the slices of the listed tiles), two builds: KB per build
            got[which] = sum(per_dispatch.values()) / 2.0 * 1024.0
        except Exception:
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return got["fetch"], got["write"]


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    the environment, rendezvous on 127.0.0.1 — what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would set),
    pass rank 0's ONE JSON line through, and when any rank fails stop the others and print a JSON line with "error"."""
    n = args.gpus
    shared = os.environ.get("MDH_BENCH_SHARED_GPU", "") == "1"

    def fail(msg, code=1):
        print(json.dumps({"metric": METRIC, "value": None, "unit": "atoms/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
                          "error": msg}))
        sys.stdout.flush()
        raise SystemExit(code)

    if not shared:
        import torch

        have = torch.cuda.device_count()
        if have < n:
            fail(f"--gpus {n} on a box with {have} visible GPU(s) (MDH_BENCH_SHARED_GPU=1 runs the ranks on one GPU over gloo, for tests)")
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               MDH_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    procs = []
    errs = []
    for r in range(n):
        errs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[r], text=True))
    bad = None
    deadline = time.time() + float(os.environ.get("MDH_BENCH_LAUNCH_TIMEOUT", "1500"))
    live = set(range(n))
    while live and bad is None:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                bad = (r, rc)
                break
        if time.time() > deadline:
            bad = (-1, "timeout")
        if live and bad is None:
            time.sleep(0.05)
    if bad is not None:
        for p in procs:  # exactly the processes started here, by handle
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        r = bad[0]
        tail = ""
        if r >= 0:
            errs[r].seek(0)
            tail = errs[r].read()[-1500:]
        fail(f"rank {r} of {n} ended with {bad[1]}: {tail.strip()}" if r >= 0 else f"no result within the launch timeout ({n} ranks)")
    out = procs[0].stdout.read()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        fail(f"rank 0 printed {len(lines)} JSON lines")
    print(lines[0])
    sys.stdout.flush()


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
    if args.cpu_child:
        return cpu_baseline_child(args)
    import torch
    import torch.distributed as dist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)  # plain `python bench.py --gpus N`: this process starts the N ranks itself
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: the two must agree")
    # MDH_BENCH_SHARED_GPU=1 (tests on a one-GPU box): every rank on cuda:0, gloo instead of RCCL — the same code path
    # above the transport, no claim about its speed
    shared = os.environ.get("MDH_BENCH_SHARED_GPU", "") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if shared else "nccl", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group of {dist.get_world_size()} ranks for --gpus {args.gpus}")

    import mdapy_amd as mp
    from mdapy_amd import _cna, _lib, _neighbor
    from mdapy_amd.distributed import SlabDecomposition, neighbor_cna_step

    L = _lib.lib()
    _lib.check(L.mdh_set_device(local_rank))
    cells, M = args.cells, args.max_neigh
    strong = args.scaling == "strong" and world > 1
    if strong:  # the ONE cells^3 box (BASELINE.json's 10 M atoms) cut into `world` slabs along x
        if cells % world != 0:
            raise SystemExit(f"--scaling strong: {cells} cells do not split into {world} equal slabs")
        x, y, z, gid = slab_positions(torch, dev, cells, rank, args.sigma, cells_x=cells // world)
        box = mp.Box(np.diag([A_CU * cells] * 3))
    else:
        x, y, z, gid = slab_positions(torch, dev, cells, rank, args.sigma)
        box = mp.Box(np.diag([A_CU * cells * world, A_CU * cells, A_CU * cells]))
    n_local = int(x.shape[0])
    dec = SlabDecomposition(box, rank, world, axis=0)
    bx = (box.box, box.origin, box.boundary)
    if world > 1:  # room behind the owned atoms: the ghosts of every step are appended there, no concatenation of the slab's arrays
        x, y, z, gid = (dec.with_room(a, 0.25 if strong else 0.05) for a in (x, y, z, gid))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(step, steps, warmup, ranges=1):
        # ranges: which launch ranges the library brackets with HIP events while the steps run — 1 all of them, 2 only
        # "k_neighbor" (the kernel of the roofline figure): every bracketed range costs ~8 us of stream time per step
        for _ in range(warmup):
            out = step()
        sync()
        L.mdh_prof_reset()
        L.mdh_prof_enable(ranges)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        sync()
        elapsed = time.perf_counter() - t0
        L.mdh_prof_enable(0)
        buf = ctypes.create_string_buffer(1 << 16)
        prof = {}
        if L.mdh_prof_report(buf, len(buf)) > 0:
            for line in buf.value.decode().strip().splitlines():
                name, cnt, tot = line.split()
                prof[name] = (int(cnt), float(tot))
        return elapsed, out, prof

    if world == 1:
        verlet = torch.empty((n_local, M), dtype=torch.int32, device=dev)
        distl = torch.empty((n_local, M), dtype=torch.float64, device=dev)
        nn = torch.empty((n_local,), dtype=torch.int32, device=dev)
        pattern = torch.empty((n_local,), dtype=torch.int32, device=dev)

        def step():
            pattern.zero_()  # the kernels rely on the caller's pre-zeroing (common_neighbor_analysis.py:128)
            _neighbor.build_neighbor_fcna(x, y, z, *bx, RC, verlet, distl, nn, pattern, 1, fill_pads=True)
            return nn, pattern, None

        def step_two_calls():  # the same step as the two reference calls one after the other (the headline until round 5)
            pattern.zero_()
            _neighbor.build_neighbor(x, y, z, *bx, RC, verlet, distl, nn, 1, fill_pads=True)
            _cna.fcna(x, y, z, *bx, verlet, nn, pattern, RC, 1)
            return nn, pattern, None
    else:
        def step():
            # every step: one halo exchange, one build, one CNA.  The exchange of the NEXT step (the next frame of a trajectory
            # does not depend on this one) is started on a side stream before this step's kernels are enqueued.
            # (the prefetch pays where the exchange's kernels are worth hiding: 2.01 -> 1.96 ms for a 10 M-atom slab; for a 1.26 M-atom
            # slab the second stream costs more than it hides, 0.355 -> 0.377 ms — profiles/r05_strong.txt, r05_halo_cost.txt)
            dom, v_, d_, nn_, pat_ = neighbor_cna_step(dec, x, y, z, gid, RC, M, next_frame=(x, y, z, gid) if n_local >= (1 << 22) else None,
                                                       reuse_buffers=True)  # (the outputs allocated once, as the single-GPU step has them)
            return nn_, pat_, dom

    # the timed region: K steps, the neighbour kernel's range timed live by HIP events on its launch stream; the other
    # ranges (cell grid, CNA) are timed by a second, short loop afterwards so that their event pairs do not sit in the K steps
    elapsed, out, prof = timed(step, args.steps, args.warmup, ranges=2)
    if world == 1:
        _, _, prof_all = timed(step, min(args.steps, 10), 0, ranges=1)
        for name, rec in prof_all.items():
            prof.setdefault(name, rec)
    multi = None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # what the exchange alone costs (selection + packing + ring + append, nothing overlapped), after the timed region:
        # max over ranks; and every rank's owned / ghost atom counts
        dom_x = dec.exchange_halo(x, y, z, gid, RC, sort=False)  # (picks up the exchange the last timed step prefetched)
        sync()
        t0 = time.perf_counter()
        for _ in range(5):
            dom_x = dec.exchange_halo(x, y, z, gid, RC, sort=False)
        sync()
        n_ghost = int(dom_x.x.shape[0]) - n_local
        mine = torch.zeros((world, 3), dtype=torch.float64, device="cpu" if shared else dev)
        mine[rank, 0], mine[rank, 1], mine[rank, 2] = n_local, n_ghost, (time.perf_counter() - t0) / 5 * 1e3
        dist.all_reduce(mine)
        mine = mine.cpu()
        multi = {"atoms_per_rank": [int(v) for v in mine[:, 0].tolist()], "ghosts_per_rank": [int(v) for v in mine[:, 1].tolist()],
                 "exchange_ms": float(mine[:, 2].max()),
                 # a ghost is x, y, z and its id, 8 bytes each; every rank receives its ghosts once per step
                 "halo_bytes_per_step": int(mine[:, 1].sum().item()) * 32,
                 "transport": "gloo, host-staged, ranks sharing one GPU (MDH_BENCH_SHARED_GPU=1): says nothing about xGMI" if shared
                              else "RCCL (torch.distributed backend nccl), batch_isend_irecv ring, device buffers"}
        del dom_x

    # N > 1: the SAME run also times the metric's own box — the ONE cells^3 box (10 M atoms) cut into `world` slabs along x (strong
    # scaling) — when the default line is the weak one; reported as config.strong.  (--scaling strong makes it the line itself.)
    strong_extra = None
    if world > 1:
        torch.cuda.synchronize()
        dec.check_halo()  # (a static exchange reports an overflowing message one step late: the last step's here)
        if not strong and cells % world == 0:
            xs_, ys_, zs_, gs_ = slab_positions(torch, dev, cells, rank, args.sigma, cells_x=cells // world)
            box_s = mp.Box(np.diag([A_CU * cells] * 3))
            dec_s = SlabDecomposition(box_s, rank, world, axis=0)
            n_s = int(xs_.shape[0])
            xs_, ys_, zs_, gs_ = (dec_s.with_room(a, 0.25) for a in (xs_, ys_, zs_, gs_))

            def step_strong():
                dom, v_, d_, nn_, pat_ = neighbor_cna_step(dec_s, xs_, ys_, zs_, gs_, RC, M, reuse_buffers=True)
                return nn_, pat_, dom

            e_s, out_s, _ = timed(step_strong, args.steps, args.warmup, ranges=0)
            torch.cuda.synchronize()
            dec_s.check_halo()
            tt = torch.tensor([e_s], dtype=torch.float64, device="cpu" if shared else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e_s = float(tt.item())
            nn_s, pat_s, dom_s = out_s
            ok_s = bool((nn_s[dom_s.owned] == 12).all().item()) and bool((pat_s[dom_s.owned] == 1).all().item()) if args.sigma == 0.0 else True
            okt = torch.tensor([1.0 if ok_s else 0.0], dtype=torch.float64, device="cpu" if shared else dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            strong_extra = {"workload": f"the ONE {cells}^3-cell box ({n_s * world} atoms) cut into {world} slabs along x, {n_s} atoms per GPU, same step",
                            "ms_per_step": e_s / args.steps * 1e3, "value": n_s * world / (e_s / args.steps), "unit": "atoms/s",
                            "atoms_per_gpu": n_s, "steps": args.steps, "result_ok": bool(okt.item() > 0.5),
                            "note": "second timed loop of the same run (barrier + synchronize on both sides, max over ranks)"}
            del xs_, ys_, zs_, gs_, out_s

    # correctness of what was timed: perfect FCC -> every owned atom has 12 neighbours and label 1
    nn_o, pat_o, dom = out
    if dom is not None:
        nn_o, pat_o = nn_o[dom.owned], pat_o[dom.owned]
    ok = int(nn_o.shape[0]) == n_local
    if args.sigma == 0.0:
        ok = ok and bool((nn_o == 12).all().item()) and bool((pat_o == 1).all().item())
    if not ok:
        raise SystemExit(f"rank {rank}: timed result is wrong (expected 12 neighbours / FCC label everywhere)")

    if rank == 0:
        n_total = n_local * world
        ms_per_step = elapsed / args.steps * 1e3
        n_rows = n_local if dom is None else int(dom.x.shape[0])  # rows the kernel actually processed on rank 0
        res = {
            "metric": METRIC,
            "value": n_total / (elapsed / args.steps),
            "unit": "atoms/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64 (CNA pair tests: f32 filter + f64 finish, labels exact)", "data": "synthetic",
            "config": {"workload": (f"FCC Cu a={A_CU}, ONE {cells}^3-cell box cut into {world} slabs along x ({n_local} atoms/GPU, {n_total} total), "
                                    if strong else f"FCC Cu a={A_CU}, {cells}^3 cells per GPU ({n_local} atoms/GPU, {n_total} total), ") +
                                   f"build_neighbor(rc=0.854a={RC:.5f}, max_neigh={M}) + fixed-cutoff CNA as one C-ABI call (mdh_build_neighbor_fcna: lists and labels), positions resident in HBM",
                       "atoms_per_gpu": n_local, "rc": RC, "max_neigh": M, "sigma": args.sigma,
                       "parallelism": f"slab{world}" if world > 1 else "single", "world_size_checked": world,
                       "launched_by": "bench.py itself (one process per GPU)" if os.environ.get("MDH_BENCH_SELF_LAUNCHED") == "1"
                                      else ("an external launcher" if world > 1 else "single process")},
        }
        if multi is not None:
            res["config"].update(multi)
            if strong_extra is not None:
                res["config"]["strong"] = strong_extra
            res["config"]["exchange_note"] = ("exchange_ms: the halo exchange alone, nothing overlapped, max over ranks; inside the timed "
                                              "steps the next step's exchange travels on a side stream under this step's kernels")
        if "k_neighbor" in prof:
            cnt, tot = prof["k_neighbor"]
            avg_ms = tot / cnt
            # read x,y,z (24 B) + write nn (4 B) + rows (12 M B) per atom, and the label this kernel writes as well (4 B)
            alg_bytes = (28 + 12 * M + 4) * n_rows
            ach = alg_bytes / (avg_ms * 1e-3) / 1e9
            traffic = traffic_raw = None
            source = None
            if world == 1 and not args.no_pmc:
                live = live_traffic(args)
                if live is not None:
                    traffic, traffic_raw, source = 2.0 * live[0] + live[1], live[0] + live[1], "live"
            if traffic is None:
                tpath = os.path.join(ROOT, "profiles", "r06_traffic.json")
                if os.path.exists(tpath) and cells == 136 and M == 16 and world == 1 and args.sigma == 0.0:
                    with open(tpath) as fh:
                        t = json.load(fh)
                    traffic, traffic_raw, source = t["traffic_bytes_per_launch_fetch_x2"], t["traffic_bytes_per_launch_raw"], "committed"
            res["roofline"] = {"bound": "hbm", "kernel": "k_neighbor (k_neighbor_lane, the instance that also labels: lists + CNA in one pass; from round 6 it also gathers "
                                                           "the atoms through the cell-sorted id list — k_gather's work, on unchanged algorithmic bytes: extra.records_path has the kernel of rounds 1-5)",
                               "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_raw_fetch_plus_write": traffic_raw,
                               "traffic_source": source, "avg_kernel_ms": avg_ms, "launches": cnt,
                               "algorithmic_bytes_per_launch": alg_bytes}
            res["kernels_ms"] = {k: v[1] / v[0] for k, v in prof.items()}
            res["kernels_ms_note"] = ("k_neighbor: HIP events inside the timed steps; the other ranges: a second loop of "
                                      f"{min(args.steps, 10)} steps after them (an event pair costs ~8 us of stream time per range and step)")
        if world == 1 and not args.no_extra:
            extra = {}
            # (0) the headline step as the two calls it was until round 5: mdh_build_neighbor, then mdh_fcna — with the plain neighbour
            # kernel's own roofline figure ((28 + 12 M) B per atom), the number the earlier rounds' lines carry
            try:
                k = max(5, args.steps // 2)
                e0, o0, p0 = timed(step_two_calls, k, 3)
                ok0 = bool((o0[0] == 12).all().item()) and bool((o0[1] == 1).all().item()) if args.sigma == 0.0 else True
                km = {k_: v_[1] / v_[0] for k_, v_ in p0.items()}
                extra["two_calls"] = {"ms_per_step": e0 / k * 1e3, "atoms_per_s": n_local / (e0 / k), "kernels_ms": km, "result_ok": ok0,
                                      "ratio_one_call_to_two": ms_per_step / (e0 / k * 1e3)}
                if "k_neighbor" in km:
                    a0 = (28 + 12 * M) * n_local / (km["k_neighbor"] * 1e-3) / 1e9
                    extra["two_calls"]["plain_neighbor_kernel_roofline"] = {"achieved": a0, "frac": a0 / HBM_PEAK_GBS, "alg_B_per_atom": 28 + 12 * M}
            except Exception as e:
                extra["two_calls"] = {"error": f"{type(e).__name__}: {e}"}

            # (0') the same one call with the atoms gathered into cell-sorted 32-byte records first (mdh_debug_set_indirect(0): the path
            # of rounds 1-5, and what unordered input still takes).  From round 6 on a build of spatially ordered input keeps no sorted
            # copy: k_gather is gone from the cell grid and the tile kernel reads the atoms through the cell-sorted id list — a faster
            # step, and a slower tile kernel on the same algorithmic bytes (the roofline object above is that kernel's)
            try:
                from mdapy_amd import _lib as _l
                k = max(5, args.steps // 2)
                prev = _l.lib().mdh_debug_set_indirect(0)
                try:
                    e1, o1, p1 = timed(step, k, 3)
                finally:
                    _l.lib().mdh_debug_set_indirect(prev)
                km1 = {k_: v_[1] / v_[0] for k_, v_ in p1.items()}
                extra["records_path"] = {"ms_per_step": e1 / k * 1e3, "kernels_ms": km1, "ratio_default_to_records": ms_per_step / (e1 / k * 1e3),
                                         "what": "mdh_debug_set_indirect(0): cell grid incl. k_gather (32-byte records), tile kernel staging from the records"}
                if "k_neighbor" in km1:
                    a1 = (28 + 12 * M + 4) * n_local / (km1["k_neighbor"] * 1e-3) / 1e9
                    extra["records_path"]["tile_kernel_roofline"] = {"achieved": a1, "frac": a1 / HBM_PEAK_GBS, "alg_B_per_atom": 28 + 12 * M + 4}
            except Exception as e:
                extra["records_path"] = {"error": f"{type(e).__name__}: {e}"}

            # (a) the default API path: max_neigh=None -> exact-width rows, counting pass + build on one cell grid (and the labels
            # in the pass that builds: mdh_build_neighbor_exact_fcna)
            def step_exact():
                p_ = torch.zeros((n_local,), dtype=torch.int32, device=dev)
                v_, d_, n_ = _neighbor.build_neighbor_without_max_neigh(x, y, z, *bx, RC, 1, pattern=p_)
                return n_, p_, v_

            del verlet, distl
            e2, o2, p2 = timed(step_exact, max(3, args.steps // 4), 2)
            extra["max_neigh_none"] = {"ms_per_step": e2 / max(3, args.steps // 4) * 1e3, "row_width": int(o2[2].shape[1]),
                                       "kernels_ms": {k: v[1] / v[0] for k, v in p2.items()}}
            del o2
            # (b) steps on inputs that are not the kernels' best case: thermally rattled lattices and a polycrystal — the CNA's
            # double-precision to-do list and the neighbour kernel's slice pass are not empty there (counts reported)
            v3 = torch.empty((n_local, M), dtype=torch.int32, device=dev); d3 = torch.empty((n_local, M), dtype=torch.float64, device=dev)
            cnt4 = (ctypes.c_int64 * 4)()

            def other_input(xs, ys, zs, bxs, n_at, both=False):
                nn_, pat_ = nn[:n_at], pattern[:n_at]
                v_, d_ = v3[:n_at], d3[:n_at]

                def step_other():  # the one call of the headline step
                    pat_.zero_()
                    _neighbor.build_neighbor_fcna(xs, ys, zs, *bxs, RC, v_, d_, nn_, pat_, 1, fill_pads=True)
                    return nn_, pat_, None

                def step_other_two():  # the two calls it replaces
                    pat_.zero_()
                    _neighbor.build_neighbor(xs, ys, zs, *bxs, RC, v_, d_, nn_, 1, fill_pads=True)
                    _cna.fcna(xs, ys, zs, *bxs, v_, nn_, pat_, RC, 1)
                    return nn_, pat_, None

                k = max(3, args.steps // 4)
                e_, o_, p_ = timed(step_other, k, 2)
                lab = torch.bincount(o_[1], minlength=5).tolist()
                mx = int(o_[0].max().item())
                two = None
                if both:
                    e2_, o2_, p2_ = timed(step_other_two, k, 2)
                    two = {"ms_per_step": e2_ / k * 1e3, "kernels_ms": {k_: v_[1] / v_[0] for k_, v_ in p2_.items()},
                           "labels_equal": torch.bincount(o2_[1], minlength=5).tolist() == lab}
                L.mdh_debug_track_counters(1)  # (the counters of the two-call form: the CNA's double-precision list, the slice pass of the build)
                step_other_two(); step_other_two(); torch.cuda.synchronize()
                L.mdh_debug_counters(cnt4)
                L.mdh_debug_track_counters(0)
                got = {"atoms": n_at, "ms_per_step": e_ / k * 1e3, "atoms_per_s": n_at / (e_ / k), "fcc_fraction": lab[1] / n_at,
                       "labels_other_fcc_hcp_bcc_ico": lab[:5], "max_neighbors": mx,
                       "todo_fraction": max(int(cnt4[0]), 0) / n_at, "cna_todo_atoms": int(cnt4[0]), "tiles_to_slice_pass": int(cnt4[1]),
                       "kernels_ms": {k_: v_[1] / v_[0] for k_, v_ in p_.items()}}
                if two is not None:
                    got["two_calls"] = two
                return got

            for sg in (0.05, 0.20):
                xs, ys, zs, _ = slab_positions(torch, dev, cells, 0, sg)
                extra[f"sigma_{sg:.2f}"] = other_input(xs, ys, zs, bx, n_local, both=True)
                del xs, ys, zs
            # (b0) atom ORDER: the headline lattice under one random permutation of its atoms (the reference's linked-list cell build
            # does not care in which order atoms arrive, neighbor.cpp:64-100; these kernels' gathers and atomics do)
            try:
                gen = torch.Generator(device=dev); gen.manual_seed(11)
                perm = torch.randperm(n_local, device=dev, generator=gen)
                got = other_input(x[perm].contiguous(), y[perm].contiguous(), z[perm].contiguous(), bx, n_local)
                extra["shuffled_ids"] = {k_: got[k_] for k_ in ("atoms", "ms_per_step", "atoms_per_s", "fcc_fraction", "kernels_ms")}
                extra["shuffled_ids"]["ratio_to_ordered"] = got["ms_per_step"] / ms_per_step
                extra["shuffled_ids"]["what"] = ("the two C-ABI calls of the headline step handed the shuffled columns as they are (what a binding "
                                                 "that bypasses System gets)")
                # the same frames through System — a NEW System per step, as a trajectory loop makes them: the shuffled one is
                # analysed on its cell-sorted twin (order statistic + sort + keyed build + CNA + labels scattered back, all inside the step)
                from mdapy_amd.devarray import HArray
                from mdapy_amd.frame import Frame

                def system_step(cols):
                    def run():
                        s_ = mp.System(data=Frame({"x": HArray(cols[0]), "y": HArray(cols[1]), "z": HArray(cols[2])}), box=box)
                        s_.cal_common_neighbor_analysis(rc=RC, max_neigh=M)
                        return s_
                    return run

                k = max(3, args.steps // 4)
                shuf = (x[perm].contiguous(), y[perm].contiguous(), z[perm].contiguous())
                res_sys = {}
                for tag, cols in (("ordered", (x, y, z)), ("shuffled", shuf)):
                    a0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
                    e_, s_, _ = timed(system_step(cols), k, 2, ranges=0)
                    lab_ = s_.data["cna"].device_array().dev()
                    res_sys[tag] = {"ms_per_step": e_ / k * 1e3, "all_fcc": bool((lab_ == 1).all().item()), "twin": s_._spatial() is not None,
                                    "hipMallocs_by_torch_during_the_loop": torch.cuda.memory_stats().get("num_device_alloc", 0) - a0}
                    del s_, lab_
                # frames 2 ... n of a trajectory (the same numbering, atoms moved a little): a new System per frame, read through the
                # previous frame's permutation instead of sorted again (system.py _sorted_as_last_time)
                gen2 = torch.Generator(device=dev); gen2.manual_seed(77)
                frames = [tuple((c + 0.03 * torch.randn(c.shape, generator=gen2, dtype=torch.float64, device=dev)).contiguous() for c in shuf) for _ in range(2)]
                state = {"i": 0}

                def next_frame():
                    state["i"] += 1
                    return system_step(frames[state["i"] & 1])()

                os.environ["MDAPY_REUSE_ORDER"] = "0"
                try:
                    e0_, _, _ = timed(next_frame, k, 2, ranges=0)
                finally:
                    del os.environ["MDAPY_REUSE_ORDER"]
                e1_, s1_, _ = timed(next_frame, k, 2, ranges=0)
                res_sys["shuffled"]["every_frame_sorted_ms"] = e0_ / k * 1e3
                res_sys["shuffled"]["second_frame_ms"] = e1_ / k * 1e3
                res_sys["shuffled"]["second_frame_fcc_fraction"] = float((s1_.data["cna"].device_array().dev() == 1).double().mean().item())
                res_sys["second_frame_ratio"] = res_sys["shuffled"]["second_frame_ms"] / res_sys["ordered"]["ms_per_step"]
                del frames, s1_
                res_sys["ratio"] = res_sys["shuffled"]["ms_per_step"] / res_sys["ordered"]["ms_per_step"]
                res_sys["ratio_to_the_headline_step"] = res_sys["shuffled"]["ms_per_step"] / ms_per_step
                extra["shuffled_ids"]["system_path"] = res_sys
                del perm, shuf
            except Exception as e:
                extra["shuffled_ids"] = {"error": f"{type(e).__name__}: {e}"}
            # (b') the inputs that used to leave the tile kernel (round 4): an unwrapped trajectory frame — every atom a few whole box
            # lengths outside the box — and the same crystal in a sheared box open along its second vector; the build's device flag says
            # whether the thread-per-atom kernel had to take the call
            try:
                xs, ys, zs, _ = slab_positions(torch, dev, cells, 0, 0.05)
                gen = torch.Generator(device=dev); gen.manual_seed(5)
                Lb = A_CU * cells
                far = [c + Lb * torch.randint(-3, 4, (n_local,), device=dev, generator=gen).double() for c in (xs, ys, zs)]
                got = other_input(*far, bx, n_local)
                extra["unwrapped_pm3_boxes"] = {k_: got[k_] for k_ in ("atoms", "ms_per_step", "fcc_fraction", "kernels_ms")}
                extra["unwrapped_pm3_boxes"]["thread_per_atom_kernel_took_the_call"] = int(cnt4[2])
                del far
                sh = 0.1
                tri = np.array([[Lb, 0, 0], [sh * Lb, Lb, 0], [0.5 * sh * Lb, sh * Lb, Lb]])
                tb = mp.Box(tri, boundary=[1, 0, 1])
                got = other_input(xs + sh * ys + 0.5 * sh * zs, ys + sh * zs, zs, (tb.box, tb.origin, tb.boundary), n_local)
                extra["sheared_open_b"] = {k_: got[k_] for k_ in ("atoms", "ms_per_step", "fcc_fraction", "kernels_ms")}
                plan8 = (ctypes.c_int * 8)(); L.mdh_debug_neighbor_plan(plan8)
                extra["sheared_open_b"]["tile_plan"] = list(plan8)[:3]
                del xs, ys, zs
            except Exception as e:  # an extra, not the headline
                extra["unwrapped_pm3_boxes"] = {"error": f"{type(e).__name__}: {e}"}
            try:  # (c) a polycrystal of the headline's size: BASELINE config 3's construction (Voronoi grains, 2.0 A overlap filter) in the same box
                rng = np.random.default_rng(2024)
                grains = max(4, int(round((A_CU * cells) ** 3 / 2.3e6)))
                unit = mp.build_crystal("Cu", "fcc", A_CU)
                poly = mp.CreatePolycrystal(unit, box=A_CU * cells, seed_number=grains, seed_position=rng.random((grains, 3)) * A_CU * cells,
                                            theta_list=rng.uniform(-180, 180, (grains, 3)), metal_overlap_dis=2.0).compute(verbose=False)
                n_poly = min(int(poly.N), n_local)
                cols = [torch.from_numpy(np.ascontiguousarray(poly.data[c].to_numpy()[:n_poly])).to(dev) for c in "xyz"]
                pb = (poly.box.box, poly.box.origin, poly.box.boundary)
                if n_poly < int(poly.N):  # (the buffers are the headline's: a box that came out fuller is cut to its first atoms, open along x)
                    pb = (poly.box.box, poly.box.origin, np.array([0, 1, 1], np.int32))
                extra["polycrystal"] = dict(other_input(*cols, pb, n_poly), grains=grains, atoms_built=int(poly.N),
                                            order="the builder's: grain by grain, each in its rotated lattice's order")
                gen = torch.Generator(device=dev); gen.manual_seed(12)
                perm = torch.randperm(n_poly, device=dev, generator=gen)
                got = other_input(*[c[perm].contiguous() for c in cols], pb, n_poly)
                extra["polycrystal_shuffled"] = {k_: got[k_] for k_ in ("atoms", "ms_per_step", "atoms_per_s", "fcc_fraction", "kernels_ms")}
                extra["polycrystal_shuffled"]["ratio_to_builder_order"] = got["ms_per_step"] / extra["polycrystal"]["ms_per_step"]
                del cols, poly, perm
            except Exception as e:  # an extra, not the headline
                extra["polycrystal"] = {"error": f"{type(e).__name__}: {e}"}
            # (d) strong scaling, one rank's share: the slab of rank 1 of 8 of the SAME box, exchange in loop-back on this GPU
            # (tools/_loopback.py: the product's exchange code with the wire replaced by a device copy; never ran on RCCL)
            try:
                from tools import _loopback

                w8 = 8
                cx = cells // w8
                if cells % w8 == 0 and cx >= 2:
                    sx, sy, sz, sg_ = slab_positions(torch, dev, cells, 1, 0.0, cells_x=cx)
                    n_slab = int(sx.shape[0])
                    dec8 = SlabDecomposition(mp.Box(np.diag([A_CU * cells] * 3)), 1, w8, axis=0)
                    _loopback.install(dec8, A_CU * cx, n_slab)
                    sx, sy, sz, sg_ = (dec8.with_room(a, 0.25) for a in (sx, sy, sz, sg_))
                    k = max(10, args.steps)
                    e_a, o_a, _ = timed(lambda: neighbor_cna_step(dec8, sx, sy, sz, sg_, RC, M, reuse_buffers=True)[3:] + (None,), k, 3, ranges=0)
                    e_b, o_b, _ = timed(lambda: neighbor_cna_step(dec8, sx, sy, sz, sg_, RC, M, next_frame=(sx, sy, sz, sg_))[3:] + (None,), k, 3, ranges=0)
                    dec8._drop_pending(); torch.cuda.synchronize()
                    extra["strong_1of8"] = {"slab_atoms": n_slab, "slab_cells": f"{cx}x{cells}x{cells}", "ms_per_step": e_a / k * 1e3,
                                            "ms_per_step_next_halo_prefetched": e_b / k * 1e3, "undivided_ms_over_8": ms_per_step / 8.0,
                                            "ratio": (e_a / k * 1e3) / (ms_per_step / 8.0), "ratio_prefetched": (e_b / k * 1e3) / (ms_per_step / 8.0),
                                            "transport": "loop-back on one GPU (device copy instead of RCCL): the wire is NOT measured"}
            except Exception as e:
                extra["strong_1of8"] = {"error": f"{type(e).__name__}: {e}"}
            # (e) BASELINE configs 2 and 4 at full size through the System API (wall time of each call, HBM-resident columns, the second
            # of two calls): config 2 = the headline lattice rattled by 0.05 A: 18 nearest neighbours, PTM, Steinhardt q4 / q6;
            # config 4 = a 9.84 M-atom Cu64Zr36 glass-like frame: streaming partial g_ab(r) to 8 A in 200 bins, Warren-Cowley at 3.6 A.
            # alg_B_per_atom: compulsory unique traffic of the call (inputs read once + outputs written once), frac = that / time / 8 TB/s
            try:
                extra.update(analyses_extra(torch, dev, mp, cells))
            except Exception as e:
                extra["config2"] = {"error": f"{type(e).__name__}: {e}"}
            # (f) the reference's OWN published calls (doc/gettingstarted/benchmark.ipynb cells 4-10, use_mdapy_efficiently.ipynb
            # cells 9-11; hardware not stated there) on the same systems, through System: wall time of the second of two calls
            try:
                extra["reference_notebook_calls"] = notebook_calls(torch, dev, mp)
            except Exception as e:
                extra["reference_notebook_calls"] = {"error": f"{type(e).__name__}: {e}"}
            res["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
