"""Multi-GPU: slab domain decomposition with a ghost halo exchanged over RCCL (xGMI).

The reference is a single-process library; this layer is added by the build
(SURVEY.md §8e).  One process per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm;
"gloo" for the CPU tests).  Atoms are owned by the rank whose slab (along one box axis, in
wrapped fractional coordinates) contains them; before a neighbor build every rank receives, from
its two ring neighbours, the atoms lying within `halo` of the shared faces (ncclSend/ncclRecv
pairs — there is no other data-path collective).

Exactness.  The local problem is solved with the GLOBAL box: ghost positions are NOT shifted, the
kernels apply the same minimum-image arithmetic and use the same global cell grid as a
single-GPU run, and the local arrays are ordered by global atom id so that "descending local
index inside a cell" == "descending global id".  The rows of owned atoms (ids, order, counts,
distances) and every label derived from them are therefore bit-identical to the single-GPU
result for the whole system; rows of ghost atoms are incomplete and discarded.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import _cna, _csp, _fast_knn, _neighbor, _ptm, _rdf, _sbo, _wcp
from .box import Box


def _torch():
    import torch

    return torch


@dataclass
class LocalDomain:
    """owned + ghost atoms of one rank, ordered by global id"""
    x: "object"
    y: "object"
    z: "object"
    gid: "object"       # int64 global ids, ascending
    owned: "object"     # bool mask
    n_owned: int


class SlabDecomposition:
    def __init__(self, box: Box, rank: int, world: int, axis: int = 0, group=None):
        assert 0 <= rank < world
        assert box.boundary[axis] == 1 or world == 1, "the decomposed axis must be periodic (ring of slabs)"
        self.box, self.rank, self.world, self.axis, self.group = box, rank, world, axis, group
        self.left = (rank - 1) % world
        self.right = (rank + 1) % world

    # -- geometry -----------------------------------------------------------
    def frac(self, x, y, z):
        """wrapped fractional coordinate along the slab axis in [0, 1)"""
        t = _torch()
        hi = t.as_tensor(self.box.inverse_box[:, self.axis].copy(), dtype=t.float64, device=x.device)
        o = self.box.origin
        f = (x - o[0]) * hi[0] + (y - o[1]) * hi[1] + (z - o[2]) * hi[2]
        f = f - t.floor(f)
        return t.where(f >= 1.0, f - 1.0, f)

    def _select_device(self, x, y, z, up_from: float, down_below: float, gid=None):
        """-> (up, down) index tensors; with gid also the (n_sel, 4) rows (x, y, z, id as f64) of both selections"""
        import ctypes

        from . import _lib

        t = _torch()
        x, y, z = x.contiguous(), y.contiguous(), z.contiguous()
        assert x.dtype == t.float64 and y.dtype == t.float64 and z.dtype == t.float64
        n = int(x.shape[0])
        up = t.empty(n, dtype=t.int32, device=x.device)
        down = t.empty(n, dtype=t.int32, device=x.device)
        o = np.ascontiguousarray(self.box.origin, dtype=np.float64)
        hi3 = np.ascontiguousarray(self.box.inverse_box[:, self.axis], dtype=np.float64)
        cnt = (ctypes.c_int64 * 2)(0, 0)
        pu = pd = g = None
        if gid is not None:
            g = gid.contiguous()
            assert g.dtype == t.int64
            pu = t.empty((n, 4), dtype=t.float64, device=x.device)
            pd = t.empty((n, 4), dtype=t.float64, device=x.device)
        _lib.check(_lib.lib().mdh_slab_halo_select(x.data_ptr(), y.data_ptr(), z.data_ptr(), n, o.ctypes.data, hi3.ctypes.data,
                                                   float(up_from), float(down_below), up.data_ptr(), down.data_ptr(), cnt,
                                                   g.data_ptr() if g is not None else None, pu.data_ptr() if pu is not None else None,
                                                   pd.data_ptr() if pd is not None else None,
                                                   _lib.DEVICE, int(t.cuda.current_stream().cuda_stream)))
        nu, nd = int(cnt[0]), int(cnt[1])
        if gid is None:
            return up[:nu].to(t.int64), down[:nd].to(t.int64)
        return up[:nu], down[:nd], pu[:nu], pd[:nd]

    def owner_of(self, x, y, z):
        t = _torch()
        return t.clamp((self.frac(x, y, z) * self.world).to(t.int64), 0, self.world - 1)

    def halo_fraction(self, halo: float) -> float:
        thick = float(self.box.get_thickness()[self.axis])
        h = (halo * (1.0 + 1e-9) + 1e-9) / thick
        assert h <= 1.0 / self.world + 1e-12 or self.world == 1, (
            f"slab thickness {thick / self.world:.3f} is smaller than the halo {halo}: use fewer ranks")
        return h

    # -- halo exchange --------------------------------------------------------
    def exchange_halo(self, x, y, z, gid, halo: float, sort: bool = True) -> LocalDomain:
        """x,y,z (f64) and gid (i64) of the OWNED atoms (1-D tensors on this rank's device).

        sort=True: the local order is ascending global id (what index-ordered kernels need to reproduce the undivided
        system's rows).  sort=False: owned atoms first, in the caller's order, then the ghosts — for kernels that take the
        ids as an ordering key (``build_neighbor(..., key=dom.gid)``); no pass over the owned atoms beyond the copy."""
        t = _torch()
        import torch.distributed as dist

        n_owned = int(x.shape[0])
        if self.world == 1:
            order = t.argsort(gid) if n_owned and not bool((gid[1:] > gid[:-1]).all()) else None
            if order is not None:
                x, y, z, gid = x[order], y[order], z[order], gid[order]
            return LocalDomain(x, y, z, gid, t.ones(n_owned, dtype=t.bool, device=x.device), n_owned)
        h = self.halo_fraction(halo)
        lo, hi = self.rank / self.world, (self.rank + 1) / self.world
        if x.is_cuda:  # selection and packing in one fused pass (slab.hip); the torch expressions below are its definition
            _, _, rows_r, rows_l = self._select_device(x, y, z, hi - h, lo + h, gid)
            send_r, send_l = rows_r.t().contiguous(), rows_l.t().contiguous()
        else:
            f = self.frac(x, y, z)
            up = (f >= hi - h).nonzero().flatten()    # goes to the right neighbour
            down = (f < lo + h).nonzero().flatten()   # goes to the left neighbour

            def pack(sel):
                return t.stack([x[sel], y[sel], z[sel], gid[sel].to(t.float64)], dim=0).contiguous()  # ids < 2^53: exact

            send_r, send_l = pack(up), pack(down)
        # sizes first (order: to-right then to-left / from-left then from-right, consistent for world == 2)
        cnt_s = [t.tensor([send_r.shape[1]], dtype=t.int64, device=x.device),
                 t.tensor([send_l.shape[1]], dtype=t.int64, device=x.device)]
        cnt_r = [t.zeros(1, dtype=t.int64, device=x.device), t.zeros(1, dtype=t.int64, device=x.device)]
        ops = [dist.P2POp(dist.isend, cnt_s[0], self.right, self.group), dist.P2POp(dist.isend, cnt_s[1], self.left, self.group),
               dist.P2POp(dist.irecv, cnt_r[0], self.left, self.group), dist.P2POp(dist.irecv, cnt_r[1], self.right, self.group)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        n_from_left, n_from_right = (int(v) for v in t.cat(cnt_r).tolist())  # one device-to-host read for both counts
        recv_l = t.empty((4, n_from_left), dtype=t.float64, device=x.device)
        recv_r = t.empty((4, n_from_right), dtype=t.float64, device=x.device)
        ops = [dist.P2POp(dist.isend, send_r, self.right, self.group), dist.P2POp(dist.isend, send_l, self.left, self.group),
               dist.P2POp(dist.irecv, recv_l, self.left, self.group), dist.P2POp(dist.irecv, recv_r, self.right, self.group)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        ghosts = t.cat([recv_l, recv_r], dim=1)
        dev = x.device
        if not sort and self.world > 2:  # two different neighbours: every ghost arrives once, any order will do
            ggid = ghosts[3].to(t.int64)
            n_tot = n_owned + int(ggid.shape[0])
            own = t.arange(n_tot, device=dev) < n_owned
            return LocalDomain(t.cat([x, ghosts[0]]), t.cat([y, ghosts[1]]), t.cat([z, ghosts[2]]), t.cat([gid, ggid]), own, n_owned)
        # ghosts in ascending id order, each once (world == 2: the same atom can arrive through both faces of the one neighbour)
        ggid, first = _unique_first(ghosts[3].to(t.int64))
        ghosts = ghosts[:, first]
        n_ghost = int(ggid.shape[0])
        n_tot = n_owned + n_ghost
        if not sort:
            own = t.arange(n_tot, device=dev) < n_owned
            return LocalDomain(t.cat([x, ghosts[0]]), t.cat([y, ghosts[1]]), t.cat([z, ghosts[2]]), t.cat([gid, ggid]), own, n_owned)
        if n_owned < 2 or bool((gid[1:] > gid[:-1]).all()):
            # Local order = ascending global id (it fixes the order inside the reference's rows).  The owned ids are already
            # ascending, the few ghosts are sorted: MERGE the two runs (two binary searches, three scatters per array) instead
            # of sorting all of them every step.
            gslot = t.searchsorted(gid, ggid) + t.arange(n_ghost, dtype=t.int64, device=dev)
            oslot = t.searchsorted(ggid, gid) + t.arange(n_owned, dtype=t.int64, device=dev)

            def merge(a_owned, a_ghost, dtype):
                out = t.empty(n_tot, dtype=dtype, device=dev)
                out[oslot] = a_owned
                out[gslot] = a_ghost
                return out

            own = t.zeros(n_tot, dtype=t.bool, device=dev)
            own[oslot] = True
            return LocalDomain(merge(x, ghosts[0], t.float64), merge(y, ghosts[1], t.float64), merge(z, ghosts[2], t.float64),
                               merge(gid, ggid, t.int64), own, n_owned)
        ax = t.cat([x, ghosts[0]]); ay = t.cat([y, ghosts[1]]); az = t.cat([z, ghosts[2]])
        ag = t.cat([gid, ggid])
        own = t.cat([t.ones(n_owned, dtype=t.bool, device=dev), t.zeros(n_ghost, dtype=t.bool, device=dev)])
        order = t.argsort(ag)
        return LocalDomain(ax[order].contiguous(), ay[order].contiguous(), az[order].contiguous(), ag[order].contiguous(),
                           own[order].contiguous(), n_owned)


def _unique_first(ids):
    t = _torch()
    s, idx = t.sort(ids, stable=True)
    keep = t.ones_like(s, dtype=t.bool)
    keep[1:] = s[1:] != s[:-1]
    return s[keep], idx[keep]


def partition_atoms(pos: np.ndarray, box: Box, world: int, axis: int = 0):
    """host helper: global ids owned by every rank (list of int64 arrays) for arbitrary input order"""
    f = (pos - box.origin) @ box.inverse_box[:, axis]
    f = f - np.floor(f)
    f[f >= 1.0] -= 1.0
    owner = np.clip((f * world).astype(np.int64), 0, world - 1)
    return [np.nonzero(owner == r)[0].astype(np.int64) for r in range(world)]


def neighbor_cna_step(dec: SlabDecomposition, x, y, z, gid, rc: float, max_neigh: int):
    """One pass of the distributed hot path: halo exchange -> neighbor build -> fixed-cutoff CNA.

    Returns (dom, verlet, dist, nn, pattern): neighbor arrays / labels for ALL local atoms in `dom`
    order (rows of ghost atoms are incomplete; select with ``dom.owned``).  ``verlet`` holds local
    indices; ``dom.gid[verlet]`` maps them to global ids.
    """
    t = _torch()
    dom = dec.exchange_halo(x, y, z, gid, rc, sort=False)
    n = int(dom.x.shape[0])
    b = dec.box
    verlet = t.empty((n, max_neigh), dtype=t.int32, device=dom.x.device)
    dist = t.empty((n, max_neigh), dtype=t.float64, device=dom.x.device)
    nn = t.empty((n,), dtype=t.int32, device=dom.x.device)
    pattern = t.zeros((n,), dtype=t.int32, device=dom.x.device)
    _neighbor.build_neighbor(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, rc, verlet, dist, nn, 1, fill_pads=True,
                             key=dom.gid if dec.world > 1 else None)
    _cna.fcna(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, verlet, nn, pattern, rc, 1)
    return dom, verlet, dist, nn, pattern


# ------------------------------------------------------------------------------------------------------------------
# k-nearest-neighbour analyses (adaptive CNA, CSP, PTM, ...): the halo is not known in advance
# ------------------------------------------------------------------------------------------------------------------
def knn_step(dec: SlabDecomposition, x, y, z, gid, k: int, halo: Optional[float] = None, neighbor_rows: int = 0,
             max_tries: int = 6):
    """k nearest neighbours of every owned atom, bit-identical to a single-GPU search of the whole system.

    A slab knows the atoms within `halo` of its faces.  The k-th neighbour distance r_k(i) of an atom proves its own row
    complete iff r_k(i) is smaller than the distance from i to the edge of the known region; the halo is doubled (all
    ranks together, one 1-element all-reduce per try) until that holds for every owned atom — and, when
    ``neighbor_rows = m > 0``, also for the first m neighbours of every owned atom (analyses that read their
    neighbours' rows: identify-diamond m=4, the two-shell PTM templates m=13).

    Returns (dom, idx, dist, valid): rows in ``dom`` order; ``idx`` holds local indices (``dom.gid[idx]`` = global ids);
    ``valid`` marks the rows proven complete (all owned rows are).
    """
    t = _torch()
    import torch.distributed as dist_

    b = dec.box
    thick = float(b.get_thickness()[dec.axis])
    if halo is None:  # about the k-th neighbour distance of a uniform system of the local density, with head-room
        n_loc = max(int(x.shape[0]), 1)
        vol = float(b.volume) / dec.world
        halo = 1.5 * (3.0 * (k + 1) / (4.0 * np.pi) * vol / n_loc) ** (1.0 / 3.0)
    for _ in range(max_tries):
        if dec.world > 1:
            halo = min(halo, thick / dec.world * (1.0 - 1e-6))
        dom = dec.exchange_halo(x, y, z, gid, halo)
        n = int(dom.x.shape[0])
        idx = t.empty((n, k), dtype=t.int32, device=dom.x.device)
        dst = t.empty((n, k), dtype=t.float64, device=dom.x.device)
        _fast_knn.knn(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, k, idx, dst, 1)
        if dec.world == 1:
            return dom, idx, dst, t.ones(n, dtype=t.bool, device=dom.x.device)
        h = dec.halo_fraction(halo)
        f = dec.frac(dom.x, dom.y, dom.z)
        lo, hi = dec.rank / dec.world, (dec.rank + 1) / dec.world
        # fractional distance to the nearer edge of the known region [lo - h, hi + h), measured around the ring
        centre = 0.5 * (lo + hi)
        off = f - centre
        off = off - t.round(off)
        room = (0.5 * (hi - lo) + h - off.abs()) * thick
        if (hi - lo) + 2 * h >= 1.0 - 1e-12:
            room = t.full_like(room, float("inf"))  # own slab + both halos cover the whole ring
        valid = dst[:, k - 1] < room * (1.0 - 1e-12)
        ok = valid[dom.owned]
        if neighbor_rows > 0:
            m = min(neighbor_rows, k)
            nb = idx[dom.owned][:, :m].long()
            ok = ok & valid[nb.clamp(min=0)].all(dim=1)
        bad = t.tensor([int((~ok).sum().item())], dtype=t.int64, device=dom.x.device)
        dist_.all_reduce(bad, group=dec.group)
        if int(bad.item()) == 0:
            return dom, idx, dst, valid
        if halo >= thick / dec.world * (1.0 - 1e-6):
            raise RuntimeError(f"knn_step: slab of thickness {thick / dec.world:.3f} cannot prove {k} neighbours complete; "
                               "use fewer ranks along this axis")
        halo *= 2.0
    raise RuntimeError("knn_step: halo did not converge")


def knn_analysis_step(dec: SlabDecomposition, x, y, z, gid, what=("acna", "csp", "ptm"), csp_neighbors: int = 12,
                      ptm_structure: str = "fcc-hcp-bcc", ptm_threshold: float = 0.1, types=None):
    """Adaptive CNA / CSP / PTM of the owned atoms from one verified 18-neighbour search (rows sorted by distance, as the
    reference's classes build them: common_neighbor_analysis.py:125-140, centro_symmetry_parameter.py:79-100,
    polyhedral_template_matching.py:110-150).  ``types``: optional int32 per OWNED atom (PTM alloy ordering) — it
    travels with the halo as a fifth packed column.  Returns (dom, results) with results[name] in ``dom`` order;
    only rows with ``dom.owned`` are meaningful."""
    t = _torch()
    flags = ptm_structure.replace("all", "dcub-dhex-graphene")
    two_shell = "ptm" in what and any(s in flags for s in ("dcub", "dhex", "graphene"))
    k = 18 if "ptm" in what else max(14 if "acna" in what else 0, csp_neighbors if "csp" in what else 0)
    tdom = None
    if types is not None:  # ghosts need their types: ship them as an extra coordinate-like payload keyed by gid
        tdom = _gather_by_gid(dec, gid, types)
    dom, idx, dst, valid = knn_step(dec, x, y, z, gid, k, neighbor_rows=13 if two_shell else 0)
    n = int(dom.x.shape[0])
    b = dec.box
    out = {}
    if "acna" in what:
        pat = t.zeros((n,), dtype=t.int32, device=dom.x.device)
        _cna.acna(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, idx, pat, 1)
        out["acna"] = pat
    if "csp" in what:
        csp = t.zeros((n,), dtype=t.float64, device=dom.x.device)
        _csp.get_csp(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, idx, csp_neighbors, csp, 1)
        out["csp"] = csp
    if "ptm" in what:
        res = t.zeros((n, 8), dtype=t.float64, device=dom.x.device)
        ind = t.zeros((n, 18), dtype=t.int32, device=dom.x.device)
        ty = None
        if tdom is not None:
            ty = tdom(dom.gid)
        _ptm.get_ptm(ptm_structure, dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, idx, ty, ptm_threshold, res, ind, 1)
        out["ptm"] = res
        out["ptm_indices"] = ind
    out["knn_idx"], out["knn_dist"], out["valid"] = idx, dst, valid
    return dom, out


def steinhardt_step(dec: SlabDecomposition, x, y, z, gid, llist, rc: float, max_neigh: int, average: bool = False,
                    wl: bool = False, wlhat: bool = False):
    """Steinhardt q_l (w_l, w_l-hat) of the owned atoms over all neighbours within ``rc`` (steinhardt_bond_orientation.py:
    cutoff mode, src/steinhardt_bond_orientation.cpp:288-575).  q_lm of an atom needs its neighbours within rc: halo rc.
    ``average=True`` (Lechner-Dellago) also needs the q_lm of those neighbours, i.e. THEIR neighbourhoods: halo 2 rc, no second
    exchange.  Rows come from the keyed neighbor build, so sums run in the order of the undivided system: identical bits.
    Returns (dom, qnarray) in ``dom`` order; rows with ``dom.owned`` are meaningful."""
    t = _torch()
    dom = dec.exchange_halo(x, y, z, gid, (2.0 if average else 1.0) * rc, sort=False)
    n = int(dom.x.shape[0])
    b = dec.box
    dev = dom.x.device
    verlet = t.empty((n, max_neigh), dtype=t.int32, device=dev)
    dist = t.empty((n, max_neigh), dtype=t.float64, device=dev)
    nn = t.empty((n,), dtype=t.int32, device=dev)
    _neighbor.build_neighbor(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, rc, verlet, dist, nn, 1, fill_pads=True,
                             key=dom.gid if dec.world > 1 else None)
    ll = np.ascontiguousarray(np.asarray(llist), dtype=np.int32)
    nl, lmax = int(ll.shape[0]), int(ll.max())
    qlm_r = t.zeros((n, nl, 2 * lmax + 1), dtype=t.float64, device=dev)
    qlm_i = t.zeros((n, nl, 2 * lmax + 1), dtype=t.float64, device=dev)
    qn = t.zeros((n, nl * (1 + int(bool(wl)) + int(bool(wlhat)))), dtype=t.float64, device=dev)
    _sbo.get_sq(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, verlet, dist, nn, np.zeros((2, 2)), ll, 0, lmax, wl, wlhat,
                average, False, rc, False, qlm_r, qlm_i, qn, 1)
    return dom, qn


def _gather_by_gid(dec: SlabDecomposition, gid, values):
    """all-gather a per-owned-atom int32 column so that any rank can look it up by global id (types are 4 B/atom:
    one all-gather of N x 12 B in total; used for PTM alloy ordering only)"""
    t = _torch()
    import torch.distributed as dist_

    if dec.world == 1:
        g, v = gid, values
    else:
        n = t.tensor([int(gid.shape[0])], dtype=t.int64, device=gid.device)
        sizes = [t.zeros(1, dtype=t.int64, device=gid.device) for _ in range(dec.world)]
        dist_.all_gather(sizes, n, group=dec.group)
        mx = int(max(int(s.item()) for s in sizes))
        pad_g = t.full((mx,), -1, dtype=t.int64, device=gid.device); pad_g[: gid.shape[0]] = gid
        pad_v = t.zeros((mx,), dtype=t.int64, device=gid.device); pad_v[: gid.shape[0]] = values.to(t.int64)
        gs = [t.empty_like(pad_g) for _ in range(dec.world)]
        vs = [t.empty_like(pad_v) for _ in range(dec.world)]
        dist_.all_gather(gs, pad_g, group=dec.group)
        dist_.all_gather(vs, pad_v, group=dec.group)
        g = t.cat([a[: int(s.item())] for a, s in zip(gs, sizes)])
        v = t.cat([a[: int(s.item())] for a, s in zip(vs, sizes)])
    order = t.argsort(g)
    g, v = g[order], v[order]

    def lookup(q):
        pos = t.searchsorted(g, q)
        return v[pos].to(t.int32).contiguous()

    return lookup


# ------------------------------------------------------------------------------------------------------------------
# reductions over the neighbor list: g(r) and the Warren-Cowley matrix (integer counts, one all-reduce each)
# ------------------------------------------------------------------------------------------------------------------
def rdf_counts_step(dec: SlabDecomposition, dom: LocalDomain, verlet, dist, nn, types, ntype: int, rc: float, nbin: int):
    """Pair counts (Nt,Nt,nbin) of the WHOLE system from every rank's owned rows (`_rdf._rdf`,
    radial_distribution_function.cpp:22-54): ghost rows are switched off through their neighbour count, the integer
    counts (exact in f64 below 2^53) are summed over the ranks.  `types` int32 0-based in ``dom`` order."""
    t = _torch()
    import torch.distributed as dist_

    nn_own = t.where(dom.owned, nn, t.zeros_like(nn)).contiguous()
    g = t.zeros((ntype, ntype, nbin), dtype=t.float64, device=dom.x.device)
    _rdf._rdf(verlet, dist, nn_own, types, g, rc, nbin)
    if dec.world > 1:
        dist_.all_reduce(g, group=dec.group)
    return g


def wcp_step(dec: SlabDecomposition, dom: LocalDomain, verlet, nn, types, ntype: int):
    """Warren-Cowley matrix of the whole system: Z_mn / Z_m / atoms-per-type counted over owned rows
    (`mdh_wcp_counts`), one int64 all-reduce, then warren_cowley_parameter.cpp:57-75."""
    t = _torch()
    import torch.distributed as dist_

    counts = t.zeros((ntype * ntype + 2 * ntype,), dtype=t.int64, device=dom.x.device)
    _wcp.get_wcp_counts(verlet, nn, types, ntype, counts, rows=dom.owned.to(t.uint8).contiguous())
    if dec.world > 1:
        dist_.all_reduce(counts, group=dec.group)
    c = counts.cpu().numpy().astype(np.int64)
    T = ntype
    zmn, zm, cnt = c[: T * T].reshape(T, T), c[T * T: T * T + T], c[T * T + T:]
    ntot = float(cnt.sum())
    wcp = np.zeros((T, T))
    for a in range(T):
        for bb in range(T):
            conc = float(cnt[bb]) / ntot
            if conc > 0 and zm[a] > 0:
                wcp[a, bb] = 1.0 - float(zmn[a, bb]) / (conc * float(zm[a]))
    return wcp
