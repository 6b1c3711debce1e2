"""-m gpu: the decomposed path on the HIP kernels.

A one-GPU box cannot host two RCCL ranks, but it can host two PROCESSES that share the GPU and talk over gloo: the slab
layer stages its messages through host memory then (mdapy_amd/distributed.py, "Transport") and everything above the wire
— halo selection kernel, keyed neighbor build, CNA, Steinhardt, verified-halo kNN analyses incl. PTM with types
travelling in the halo, RDF / Warren-Cowley reductions — runs exactly as it does over RCCL.  Every rank compares its owned
rows bit for bit with the same kernels run on the undivided system."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _system(seed=5):
    from mdapy_amd.build_lattice import lattice_positions

    a = 3.615
    pos, boxm = lattice_positions("fcc", a, 28, 14, 14)  # 21 952 atoms; 14 cells per rank and axis at world 2
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0, 0.07, pos.shape)             # some atoms leave the box: ownership uses wrapped coordinates
    pos[rng.random(len(pos)) < 0.03] += rng.normal(0, 0.5, 3)
    types = rng.integers(1, 3, len(pos)).astype(np.int32)
    perm = rng.permutation(len(pos))                        # arbitrary input order: global ids are not slab-contiguous
    return pos[perm], types[perm], boxm, a


def _undivided(torch, K, pos, types, boxm, rc, M, ll):
    """the same kernels on the whole system (HBM resident)"""
    dev = torch.device("cuda", 0)
    x, y, z = (torch.from_numpy(np.ascontiguousarray(pos[:, k])).to(dev) for k in range(3))
    org, bnd = np.zeros(3), np.array([1, 1, 1], np.int32)
    N = len(pos)
    where = (x, y, z, boxm, org, bnd)
    v = torch.empty((N, M), dtype=torch.int32, device=dev); d = torch.empty((N, M), dtype=torch.float64, device=dev)
    nn = torch.empty(N, dtype=torch.int32, device=dev)
    K.neighbor.build_neighbor(*where, rc, v, d, nn, 1, fill_pads=True)
    pat = torch.zeros(N, dtype=torch.int32, device=dev)
    K.cna.fcna(*where, v, nn, pat, rc, 1)
    out = {"v": v, "d": d, "nn": nn, "fcna": pat}
    for average in (False, True):
        qr = torch.zeros((N, 2, 13), dtype=torch.float64, device=dev); qi = torch.zeros_like(qr)
        qn = torch.zeros((N, 4), dtype=torch.float64, device=dev)
        K.sbo.get_sq(*where, v, d, nn, np.zeros((2, 2)), ll, 0, 6, True, False, average, False, rc, False, qr, qi, qn, 1)
        out[f"q{int(average)}"] = qn
    idx = torch.empty((N, 18), dtype=torch.int32, device=dev); dk = torch.empty((N, 18), dtype=torch.float64, device=dev)
    K.fast_knn.knn(*where, 18, idx, dk, 1)
    acna = torch.zeros(N, dtype=torch.int32, device=dev); K.cna.acna(*where, idx, acna, 1)
    csp = torch.zeros(N, dtype=torch.float64, device=dev); K.csp.get_csp(*where, idx, 12, csp, 1)
    res = torch.zeros((N, 8), dtype=torch.float64, device=dev); ind = torch.zeros((N, 18), dtype=torch.int32, device=dev)
    K.ptm.get_ptm("fcc-hcp-bcc", *where, idx, torch.from_numpy(types).to(dev), 0.1, res, ind, 1)
    out.update(knn_idx=idx, knn_dist=dk, acna=acna, csp=csp, ptm=res, ptm_indices=ind)
    t0 = torch.from_numpy(types - 1).to(dev)
    g = torch.zeros((2, 2, 40), dtype=torch.float64, device=dev)
    K.rdf._rdf(v, d, nn, t0, g, rc, 40)
    W = np.zeros((2, 2)); K.wcp.get_wcp(v, nn, t0, 2, W, 1)
    gs = torch.zeros((2, 2, 50), dtype=torch.float64, device=dev)
    K.rdf._rdf_streaming(x, y, z, t0, boxm, org, bnd, gs, 2.2 * rc, 50)  # a cutoff no list is built for: the streaming kernel's case
    out.update(rdf=g, wcp=W, rdf_stream=gs)
    return {k: (a.cpu().numpy() if hasattr(a, "cpu") else a) for k, a in out.items()}


def _run_rank(rank, world, torch, dist_ready=True):
    import mdapy_amd as mp
    import mdapy_amd.distributed as D
    import mdapy_amd.kernels as K

    dev = torch.device("cuda", 0)
    pos, types, boxm, a = _system()
    rc, M = 0.854 * a, 20
    ll = np.array([4, 6], np.int32)
    box = mp.Box(boxm)
    ref = _undivided(torch, K, pos, types, boxm, rc, M, ll)
    owned_ids = D.partition_atoms(pos, box, world, axis=0)[rank]
    t = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    own_args = (t(pos[owned_ids, 0]), t(pos[owned_ids, 1]), t(pos[owned_ids, 2]), t(owned_ids))
    dec = D.SlabDecomposition(box, rank, world, axis=0)
    bad = []

    def check(name, ok):
        if not ok:
            bad.append(name)

    # ---- neighbor + fixed-cutoff CNA (the bench's step)
    def check_step(tag, dom, v, d, nn, pat):
        own, gid = dom.owned.cpu().numpy(), dom.gid.cpu().numpy()
        g_own = gid[own]
        check(tag + "owned set", own.sum() == len(owned_ids) and np.array_equal(np.sort(g_own), np.sort(owned_ids)))
        vloc = v.cpu().numpy()[own]
        check(tag + "rows", np.array_equal(np.where(vloc >= 0, gid[np.clip(vloc, 0, None)], -1), np.where(ref["v"][g_own] >= 0, ref["v"][g_own], -1)))
        check(tag + "counts", np.array_equal(nn.cpu().numpy()[own], ref["nn"][g_own]))
        check(tag + "distances", np.array_equal(d.cpu().numpy()[own], ref["d"][g_own]))
        check(tag + "fcna", np.array_equal(pat.cpu().numpy()[own], ref["fcna"][g_own]))
        return own, gid

    dom, v, d, nn, pat = D.neighbor_cna_step(dec, *own_args, rc, M)
    own, gid = check_step("", dom, v, d, nn, pat)
    check("fcna nontrivial", len(np.unique(ref["fcna"])) > 1)
    # the same step with the next frame's halo started before this frame's kernels (start_halo): three times, so that the
    # later calls consume the exchange the one before began (the order of the ghosts inside a message is not fixed)
    for it in range(3):
        check_step(f"prefetched halo {it}: ", *D.neighbor_cna_step(dec, *own_args, rc, M, next_frame=own_args))
        check("an exchange is under way after the step", world < 2 or len(dec._pending) == 1)
    dec._drop_pending()
    # ---- the same step on tensors with room behind them (with_room): the STATIC exchange — the ghost count never leaves the device,
    # the ghost block has a fixed size and its unused slots are absent atoms (x = NaN, id -1) that the build gives no cell
    roomy = tuple(dec.with_room(a, 0.6) for a in own_args)
    for it in range(3):
        out_s = D.neighbor_cna_step(dec, *roomy, rc, M, next_frame=roomy)
        check(f"static exchange {it}: taken", world < 2 or getattr(out_s[0], "absent_slots", False))
        check_step(f"static exchange {it}: ", *out_s)
        if world > 1:
            gx, gg = out_s[0].x.cpu().numpy(), out_s[0].gid.cpu().numpy()
            check(f"static exchange {it}: absent slots", np.array_equal(np.isnan(gx), gg < 0) and np.isnan(gx).sum() > 0
                  and not np.isnan(gx[:len(owned_ids)]).any())
            check(f"static exchange {it}: absent atoms have no neighbours", int(out_s[3].cpu().numpy()[gg < 0].max()) == 0)
    dec._drop_pending()
    dec.check_halo()
    if world > 1:  # a message that does not fit its agreed size is reported once the step has run
        dec2 = D.SlabDecomposition(box, rank, world, axis=0)
        dec2._msg_cap[(float(rc), 4)] = 8
        roomy2 = tuple(dec2.with_room(a, 0.6) for a in own_args)
        D.neighbor_cna_step(dec2, *roomy2, rc, M)
        torch.cuda.synchronize()
        try:
            dec2.check_halo()
            check("overflow of a static halo message is reported", False)
        except RuntimeError:
            pass
    # ---- a prefetched exchange keeps its own message buffers: frame B's halo is started, then ANOTHER exchange
    # with the same (halo, columns) runs for frame A on the main stream, then B's is picked up — and must hold B's ghosts
    yB = pos[:, 1] + 0.25
    frame_b = (own_args[0].clone(), t(yB[owned_ids]), own_args[2].clone(), own_args[3].clone())
    D.neighbor_cna_step(dec, *own_args, rc, M, next_frame=frame_b)
    check("frame B's exchange is under way", world < 2 or len(dec._pending) == 1)
    # ---- Steinhardt over the cutoff list, plain and neighbour-averaged (halo 2 rc); the first one exchanges frame A again
    own_a2 = tuple(a.clone() for a in own_args)  # (other tensors: not the pending key)
    for average in (False, True):
        dq, qloc = D.steinhardt_step(dec, *own_a2, ll, rc, M, average=average, wl=True)
        oq = dq.owned.cpu().numpy()
        check(f"steinhardt average={average}", np.array_equal(qloc.cpu().numpy()[oq], ref[f"q{int(average)}"][dq.gid.cpu().numpy()[oq]]))
    check("frame B's exchange is still waiting", world < 2 or len(dec._pending) == 1)
    dom_b = dec.exchange_halo(*frame_b, rc, sort=False)
    gb, ob = dom_b.gid.cpu().numpy(), dom_b.owned.cpu().numpy()
    check("frame B's ghosts are frame B's", np.array_equal(dom_b.y.cpu().numpy(), yB[gb]) and np.array_equal(dom_b.x.cpu().numpy(), pos[gb, 0])
          and (world < 2 or (~ob).sum() > 0))
    check("no exchange is left over", len(dec._pending) == 0 and len(dec._busy) == 0)
    # ---- verified-halo kNN analyses; the types of the ghosts travel with the halo
    dk, res = D.knn_analysis_step(dec, *own_args, what=("acna", "csp", "ptm"), types=t(types[owned_ids]))
    ok_, gk = dk.owned.cpu().numpy(), dk.gid.cpu().numpy()
    gko = gk[ok_]
    check("knn_dist", np.array_equal(res["knn_dist"].cpu().numpy()[ok_], ref["knn_dist"][gko]))
    check("knn_ids", np.array_equal(gk[res["knn_idx"].cpu().numpy()[ok_]], ref["knn_idx"][gko]))
    check("acna", np.array_equal(res["acna"].cpu().numpy()[ok_], ref["acna"][gko]))
    check("csp", np.array_equal(res["csp"].cpu().numpy()[ok_], ref["csp"][gko]))
    check("ptm", np.array_equal(res["ptm"].cpu().numpy()[ok_], ref["ptm"][gko]))
    pi = res["ptm_indices"].cpu().numpy()[ok_]
    check("ptm_indices", np.array_equal(np.where(pi >= 0, gk[np.clip(pi, 0, None)], -1), ref["ptm_indices"][gko]))
    check("ptm nontrivial", len(np.unique(ref["ptm"][:, 0])) > 1)
    # a deeper search than PTM's 18 (CSP over 24): PTM still gets its 18 columns
    dk2, res2 = D.knn_analysis_step(dec, *own_args, what=("csp", "ptm"), csp_neighbors=24, types=t(types[owned_ids]))
    o2 = dk2.owned.cpu().numpy()
    check("ptm beside csp24", res2["knn_idx"].shape[1] == 24 and np.array_equal(res2["ptm"].cpu().numpy()[o2], ref["ptm"][dk2.gid.cpu().numpy()[o2]]))
    # ---- list reductions
    ty = t((types - 1)[gid])
    g = D.rdf_counts_step(dec, dom, v, d, nn, ty, 2, rc, 40)
    w = D.wcp_step(dec, dom, v, nn, ty, 2)
    check("rdf counts", np.array_equal(g.cpu().numpy(), ref["rdf"]) and ref["rdf"].sum() > 0)
    check("wcp", np.array_equal(w, ref["wcp"]))
    gs = D.rdf_streaming_step(dec, *own_args, t((types - 1)[owned_ids]), 2, 2.2 * rc, 50)
    check("streaming rdf counts", np.array_equal(gs.cpu().numpy(), ref["rdf_stream"]) and ref["rdf_stream"].sum() > 0)
    return bad, int(own.sum()), int((~own).sum())


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _run_rank(rank, world, torch))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, [repr(e) + traceback.format_exc()[-1500:]], 0, 0))
    finally:
        dist.destroy_process_group()


def test_decomposed_steps_world1_equal_the_undivided_system():
    import torch

    bad, n_own, n_ghost = _run_rank(0, 1, torch)
    assert bad == [] and n_ghost == 0 and n_own == 28 * 14 * 14 * 4


@pytest.mark.parametrize("world", [2, 4])
def test_decomposed_steps_on_hip_kernels_equal_the_undivided_system(world):
    import torch.multiprocessing as tmp

    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, bad, n_own, n_ghost in sorted(res):
        assert bad == [], f"rank {rank}: {bad}"
        assert n_own > 0 and n_ghost > 0
    assert sum(r[2] for r in res) == 28 * 14 * 14 * 4


def test_loop_back_slab_step_with_the_halo_on_a_side_stream():
    """tools/halo_cost.py: one slab of an 8-rank decomposition with the two P2P exchanges looped back on the device — the only
    place where the exchange really runs on its own HIP stream beside the kernels on a one-GPU box.  Every owned atom of the
    perfect lattice must come out FCC with 12 neighbours, in the plain and in the pipelined step."""
    import subprocess

    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "halo_cost.py"), "40", "8"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    assert "owned atoms all FCC with 12 neighbours: True / True" in run.stdout, run.stdout[-1500:]


def test_bench_runs_its_multi_rank_path():
    """bench.py --gpus 2 exactly as the driver launches it (torch.distributed.run, one process per rank), with the one
    change a one-GPU box forces: MDH_BENCH_SHARED_GPU=1 puts both ranks on cuda:0 and swaps RCCL for gloo.  Checks the
    JSON contract of the N > 1 line and that a world size that does not match --gpus is refused."""
    import json
    import subprocess

    env = dict(os.environ, MDH_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--cells", "40"]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 prints ONE line
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["world_size_checked"] == 2 and res["config"]["parallelism"] == "slab2"
    assert res["config"]["atoms_per_gpu"] == 4 * 40 ** 3 and res["value"] > 0 and res["higher_is_better"] is True
    assert abs(res["value"] - 2 * res["config"]["atoms_per_gpu"] / (res["ms_per_step"] * 1e-3)) < 1e-6 * res["value"]
    assert res["roofline"]["bound"] == "hbm" and 0 < res["roofline"]["frac"] < 1 and "cpu_baseline" not in res
    assert res["config"]["launched_by"] == "an external launcher" and len(res["config"]["atoms_per_rank"]) == 2
    wrong = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "20"], cwd=ROOT,
                           env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True, text=True, timeout=300)
    assert wrong.returncode != 0 and "must agree" in (wrong.stderr + wrong.stdout)


def test_bench_starts_its_own_ranks():
    """plain `python bench.py --gpus 2` — the driver's command form — with no launcher around it: bench.py starts one process
    per rank itself, rank 0's ONE JSON line comes through, and a rank that fails turns into a JSON line with "error" and a
    non-zero exit code.  (MDH_BENCH_SHARED_GPU=1: both ranks on this box's one GPU, gloo instead of RCCL.)"""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MDH_BENCH_SHARED_GPU"] = "1"
    run = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--cells", "40"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, (run.stdout + run.stderr)[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    cfg = res["config"]
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] > 0 and "error" not in res
    assert cfg["launched_by"].startswith("bench.py itself") and cfg["parallelism"] == "slab2"
    assert cfg["atoms_per_rank"] == [4 * 40 ** 3] * 2 and all(g > 0 for g in cfg["ghosts_per_rank"])
    assert cfg["halo_bytes_per_step"] == 32 * sum(cfg["ghosts_per_rank"]) and cfg["exchange_ms"] > 0
    assert abs(res["value"] - 2 * cfg["atoms_per_gpu"] / (res["ms_per_step"] * 1e-3)) < 1e-6 * res["value"]
    # the same run also times the metric's own box cut into the ranks' slabs (strong scaling), as config.strong
    st = cfg["strong"]
    assert st["result_ok"] and st["atoms_per_gpu"] == 4 * 40 ** 3 // 2 and st["value"] > 0 and st["ms_per_step"] > 0
    assert abs(st["value"] - 4 * 40 ** 3 / (st["ms_per_step"] * 1e-3)) < 1e-6 * st["value"]
    # a failing rank: 20 cells do not split into 3 equal slabs
    bad = subprocess.run([sys.executable, "bench.py", "--gpus", "3", "--scaling", "strong", "--steps", "1", "--warmup", "0", "--cells", "20"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0
    lines = [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "do not split" in json.loads(lines[0])["error"]
    # more ranks than GPUs without the test switch: refused with an error line, no rank started
    env.pop("MDH_BENCH_SHARED_GPU")
    import torch

    many = torch.cuda.device_count() + 1
    over = subprocess.run([sys.executable, "bench.py", "--gpus", str(many), "--steps", "1", "--warmup", "0", "--cells", "20"], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=300)
    assert over.returncode != 0 and "visible GPU" in json.loads([ln for ln in over.stdout.splitlines() if ln.startswith("{")][0])["error"]
