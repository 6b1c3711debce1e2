"""Steinhardt bond-orientational order.  Mirrors
``mdapy.steinhardt_bond_orientation.SteinhardtBondOrientation``
(src/mdapy/steinhardt_bond_orientation.py:162-302)."""
from __future__ import annotations

import numpy as np

from . import _sbo
from . import tool_function as tool
from .box import Box
from .devarray import as_numpy, zeros
from .frame import Frame
from .parallel import get_num_threads


class SteinhardtBondOrientation:
    def __init__(self, box: Box, data: Frame, llist, nnn: int, rc: float, average: bool, use_voronoi: bool,
                 use_weight: bool, weight, verlet_list, distance_list, neighbor_number, wl: bool, wlhat: bool,
                 identify_liquid: bool, threshold: float, n_bond: int) -> None:
        self.box = box
        self.data = data
        self.llist = np.asarray(llist, int)
        self.nnn = nnn
        self.rc = rc
        self.average = average
        self.use_voronoi = use_voronoi
        self.use_weight = use_weight
        self.weight = weight
        self.verlet_list = verlet_list
        self.distance_list = distance_list
        self.neighbor_number = neighbor_number
        self.wl = wl
        self.wlhat = wlhat
        self.identify_liquid = identify_liquid
        self.threshold = threshold
        self.n_bond = n_bond

    def compute(self) -> None:
        if self.identify_liquid:
            assert 6 in self.llist
            assert self.threshold > 0
            assert self.n_bond > 0
        N = self.data.shape[0]
        nl = self.llist.shape[0]
        lmax = int(self.llist.max())
        self.qlm_r = zeros((N, nl, 2 * lmax + 1), np.float64)
        self.qlm_i = zeros((N, nl, 2 * lmax + 1), np.float64)
        ncol = nl + (nl if self.wl else 0) + (nl if self.wlhat else 0)
        self.qnarray = zeros((N, ncol), np.float64)
        if self.use_voronoi:  # :240-246
            self.rc = 10000000000.0
        elif self.nnn > 0:
            self.rc = 1000000000.0
        else:
            assert self.rc > 0
        if not self.use_weight:
            self.weight = np.zeros((2, 2))
        else:
            assert tuple(self.weight.shape) == tuple(self.verlet_list.shape)
        x, y, z = tool.xyz(self.data)
        _sbo.get_sq(x, y, z, self.box.box, self.box.origin, self.box.boundary, self.verlet_list, self.distance_list,
                    self.neighbor_number, self.weight, self.llist, self.nnn, lmax, self.wl, self.wlhat, self.average,
                    self.use_voronoi, self.rc, self.use_weight, self.qlm_r, self.qlm_i, self.qnarray,
                    get_num_threads())
        if self.identify_liquid:
            Q6index = int(np.where(self.llist == 6)[0][0])
            Q6 = np.ascontiguousarray(as_numpy(self.qnarray)[:, Q6index])
            self.solidliquid = zeros(N, np.int32)
            self.nbond = zeros(N, np.int32)
            _sbo.identifySolidLiquid(Q6index, Q6, self.verlet_list, self.distance_list, self.neighbor_number,
                                     self.qlm_r, self.qlm_i, float(self.threshold), int(self.n_bond),
                                     self.solidliquid, self.nbond, self.use_voronoi, self.nnn, self.rc,
                                     get_num_threads())
