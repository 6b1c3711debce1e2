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

from . import _cna, _neighbor
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
    def exchange_halo(self, x, y, z, gid, halo: float) -> LocalDomain:
        """x,y,z (f64) and gid (i64) of the OWNED atoms (1-D tensors on this rank's device)."""
        t = _torch()
        import torch.distributed as dist

        n_owned = int(x.shape[0])
        if self.world == 1:
            order = t.argsort(gid) if n_owned and not bool((gid[1:] > gid[:-1]).all()) else None
            if order is not None:
                x, y, z, gid = x[order], y[order], z[order], gid[order]
            return LocalDomain(x, y, z, gid, t.ones(n_owned, dtype=t.bool, device=x.device), n_owned)
        h = self.halo_fraction(halo)
        f = self.frac(x, y, z)
        lo, hi = self.rank / self.world, (self.rank + 1) / self.world
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
        recv_l = t.empty((4, int(cnt_r[0].item())), dtype=t.float64, device=x.device)
        recv_r = t.empty((4, int(cnt_r[1].item())), dtype=t.float64, device=x.device)
        ops = [dist.P2POp(dist.isend, send_r, self.right, self.group), dist.P2POp(dist.isend, send_l, self.left, self.group),
               dist.P2POp(dist.irecv, recv_l, self.left, self.group), dist.P2POp(dist.irecv, recv_r, self.right, self.group)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        ghosts = t.cat([recv_l, recv_r], dim=1)
        ggid = ghosts[3].to(t.int64)
        if self.world == 2:  # the same atom can arrive through both faces of the single neighbour
            ggid, first = _unique_first(ggid)
            ghosts = ghosts[:, first]
        ax = t.cat([x, ghosts[0]]); ay = t.cat([y, ghosts[1]]); az = t.cat([z, ghosts[2]])
        ag = t.cat([gid, ggid])
        own = t.cat([t.ones(n_owned, dtype=t.bool, device=x.device), t.zeros(ggid.shape[0], dtype=t.bool, device=x.device)])
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
    dom = dec.exchange_halo(x, y, z, gid, rc)
    n = int(dom.x.shape[0])
    b = dec.box
    verlet = t.empty((n, max_neigh), dtype=t.int32, device=dom.x.device)
    dist = t.empty((n, max_neigh), dtype=t.float64, device=dom.x.device)
    nn = t.empty((n,), dtype=t.int32, device=dom.x.device)
    pattern = t.zeros((n,), dtype=t.int32, device=dom.x.device)
    _neighbor.build_neighbor(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, rc, verlet, dist, nn, 1, fill_pads=True)
    _cna.fcna(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, verlet, nn, pattern, rc, 1)
    return dom, verlet, dist, nn, pattern
