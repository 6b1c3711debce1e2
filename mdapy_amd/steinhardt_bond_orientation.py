"""Steinhardt bond-orientational order — the drop-in for
``mdapy.steinhardt_bond_orientation.SteinhardtBondOrientation`` (src/mdapy/steinhardt_bond_orientation.py:162-302).

``qnarray`` (N, columns): q_l for every l of ``llist``, then w_l, then w_l-hat when asked for; ``qlm_r`` / ``qlm_i`` the
complex q_lm they were built from.  The neighbourhood is one of: a Voronoi list (optionally face-area weighted), the
``nnn`` nearest neighbours, or everything within ``rc``.  ``identify_liquid`` adds ``solidliquid`` / ``nbond`` from the
q6-q6 bond criterion."""
import numpy as np

from . import kernels, policy
from .devarray import as_numpy, zeros
from .parallel import get_num_threads

# the kernels cut the neighbourhood by distance; a k-nearest or Voronoi list is passed through with a cut nothing reaches
NO_CUT_NEAREST = 1000000000.0
NO_CUT_VORONOI = 10000000000.0


class SteinhardtBondOrientation:
    def __init__(self, box, data, llist, nnn, rc, average, use_voronoi, use_weight, weight, verlet_list, distance_list,
                 neighbor_number, wl, wlhat, identify_liquid, threshold, n_bond):
        self.box, self.data = box, data
        self.llist = np.asarray(llist, int)
        self.nnn, self.rc, self.average = nnn, rc, average
        self.use_voronoi, self.use_weight, self.weight = use_voronoi, use_weight, weight
        self.verlet_list, self.distance_list, self.neighbor_number = verlet_list, distance_list, neighbor_number
        self.wl, self.wlhat = wl, wlhat
        self.identify_liquid, self.threshold, self.n_bond = identify_liquid, threshold, n_bond

    def _effective_cut(self):
        if self.use_voronoi:
            return NO_CUT_VORONOI
        if self.nnn > 0:
            return NO_CUT_NEAREST
        assert self.rc > 0
        return self.rc

    def compute(self):
        if self.identify_liquid:
            assert 6 in self.llist
            assert self.threshold > 0
            assert self.n_bond > 0
        atoms, orders = self.data.shape[0], self.llist.shape[0]
        top = int(self.llist.max())
        shape_lm = (atoms, orders, 2 * top + 1)
        self.qlm_r, self.qlm_i = zeros(shape_lm, np.float64), zeros(shape_lm, np.float64)
        blocks = 1 + int(bool(self.wl)) + int(bool(self.wlhat))
        self.qnarray = zeros((atoms, orders * blocks), np.float64)
        self.rc = self._effective_cut()
        if self.use_weight:
            assert tuple(self.weight.shape) == tuple(self.verlet_list.shape)
        else:
            self.weight = np.zeros((2, 2))  # placeholder the unweighted kernel never reads
        lists = (self.verlet_list, self.distance_list, self.neighbor_number)
        mode = (self.wl, self.wlhat, self.average, self.use_voronoi, self.rc, self.use_weight)
        moments = (self.qlm_r, self.qlm_i)
        kernels.sbo.get_sq(*policy.positions(self.data), *policy.box_args(self.box), *lists, self.weight, self.llist,
                           self.nnn, top, *mode, *moments, self.qnarray, get_num_threads())
        if not self.identify_liquid:
            return
        slot = int(np.flatnonzero(self.llist == 6)[0])
        q6 = np.ascontiguousarray(as_numpy(self.qnarray)[:, slot])
        self.solidliquid, self.nbond = zeros(atoms, np.int32), zeros(atoms, np.int32)
        rule = (float(self.threshold), int(self.n_bond))
        kernels.sbo.identifySolidLiquid(slot, q6, *lists, *moments, *rule, self.solidliquid, self.nbond, self.use_voronoi,
                                        self.nnn, self.rc, get_num_threads())
