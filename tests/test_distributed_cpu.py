"""world_size-2 gloo test (CPU) of the slab decomposition + ghost-halo exchange in mdapy_amd/distributed.py.

The local kernels are replaced by the CPU oracle (test infrastructure), so what is checked here is the N>1 data
path: ownership, halo selection, the ring exchange, global-id ordering — and the claim that solving the local
problem with the GLOBAL box reproduces the single-process result for every owned atom bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mdapy_amd as mp
        import mdapy_amd.distributed as D
        from mdapy_amd.build_lattice import lattice_positions
        from oracle import oracle as O

        # oracle adapters for the two shim functions the distributed step calls (torch CPU tensors in / out)
        def build_neighbor(x, y, z, box, origin, boundary, rc, v, d, nn, num_t=1, fill_pads=False):
            vn, dn, nnn = v.numpy(), d.numpy(), nn.numpy()
            if fill_pads:
                vn.fill(-1); dn.fill(rc + 1.0); nnn.fill(0)
            O.build_neighbor(x.numpy(), y.numpy(), z.numpy(), box, origin, boundary, rc, vn, dn, nnn, 2)

        def fcna(x, y, z, box, origin, boundary, v, nn, pat, rc, num_t=1):
            O.fcna(x.numpy(), y.numpy(), z.numpy(), box, origin, boundary, v.numpy(), nn.numpy(), pat.numpy(), rc, 2)

        D._neighbor = type("M", (), {"build_neighbor": staticmethod(build_neighbor)})
        D._cna = type("M", (), {"fcna": staticmethod(fcna)})

        a = 3.615
        pos, boxm = lattice_positions("fcc", a, 12, 6, 6)
        rng = np.random.default_rng(5)
        pos = pos + rng.normal(0, 0.08, pos.shape)       # some atoms leave the box: ownership uses wrapped coordinates
        perm = rng.permutation(len(pos))                  # arbitrary input order: global ids are not slab-contiguous
        pos = pos[perm]
        box = mp.Box(boxm)
        rc, M = 0.854 * a, 16
        owned_ids = D.partition_atoms(pos, box, world, axis=0)[rank]
        t = lambda arr: torch.from_numpy(np.ascontiguousarray(arr))
        dec = D.SlabDecomposition(box, rank, world, axis=0)
        dom, v, d, nn, pat = D.neighbor_cna_step(dec, t(pos[owned_ids, 0]), t(pos[owned_ids, 1]), t(pos[owned_ids, 2]),
                                                 t(owned_ids), rc, M)
        own = dom.owned.numpy()
        gid = dom.gid.numpy()
        assert own.sum() == len(owned_ids) and np.array_equal(np.sort(gid[own]), np.sort(owned_ids))
        assert np.all(gid[1:] > gid[:-1])
        # single-process reference on the whole system
        x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
        org, bnd = np.zeros(3), np.array([1, 1, 1], np.int32)
        V = np.full((len(x), M), -1, np.int32); Dd = np.full((len(x), M), rc + 1.0); NN = np.zeros(len(x), np.int32)
        O.build_neighbor(x, y, z, boxm, org, bnd, rc, V, Dd, NN, 2)
        P = np.zeros(len(x), np.int32)
        O.fcna(x, y, z, boxm, org, bnd, V, NN, P, rc, 2)
        g_own = gid[own]
        vloc = v.numpy()[own]
        vglob = np.where(vloc >= 0, gid[np.clip(vloc, 0, None)], -1)   # local indices -> global ids
        ok = (np.array_equal(nn.numpy()[own], NN[g_own]) and np.array_equal(vglob, V[g_own])
              and np.array_equal(d.numpy()[own], Dd[g_own]) and np.array_equal(pat.numpy()[own], P[g_own]))
        q.put((rank, bool(ok), int(own.sum()), int((~own).sum())))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, False, repr(e) + traceback.format_exc()[-800:], 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_halo_exchange_matches_single_process(world):
    import torch.multiprocessing as tmp

    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, n_own, n_ghost in sorted(res):
        assert ok is True, f"rank {rank}: {n_own}"
        assert n_own > 0 and n_ghost > 0
    assert sum(r[2] for r in res) == 12 * 6 * 6 * 4
