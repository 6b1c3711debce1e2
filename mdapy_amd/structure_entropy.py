"""Pair-entropy fingerprint.  Mirrors ``mdapy.structure_entropy.StructureEntropy``
(src/mdapy/structure_entropy.py:14-130)."""
from __future__ import annotations

import numpy as np

from . import _neighbor, _structure_entropy
from .box import Box
from .devarray import empty
from .parallel import get_num_threads


class StructureEntropy:
    def __init__(self, box: Box, verlet_list, distance_list, neighbor_number, rc: float, sigma: float,
                 use_local_density: bool, average_rc: float = 0.0):
        self.box = box
        self.verlet_list = verlet_list
        self.distance_list = distance_list
        self.neighbor_number = neighbor_number
        self.rc = rc
        self.sigma = sigma
        self.use_local_density = use_local_density
        self.average_rc = average_rc

    def compute(self):
        n = int(self.verlet_list.shape[0])
        self.entropy = empty(n, np.float64)
        _structure_entropy.calculate_structure_entropy(self.rc, self.sigma, self.use_local_density, self.box.volume,
                                                       self.distance_list, self.neighbor_number, self.entropy,
                                                       get_num_threads())
        if self.average_rc > 0:
            assert self.average_rc <= self.rc, "average_rc should be smaller than rc."
            self.entropy_ave = empty(n, np.float64)
            _neighbor.average_by_neighbor(self.average_rc, self.verlet_list, self.distance_list, self.neighbor_number,
                                          self.entropy, self.entropy_ave, True, get_num_threads())
