"""Ackland-Jones bond-angle analysis — the drop-in for ``mdapy.ackland_jones_analysis.AcklandJonesAnalysis``
(src/mdapy/ackland_jones_analysis.py:15-120).  ``aja``: 0 other, 1 fcc, 2 hcp, 3 bcc, 4 ico; needs the 14 nearest
neighbours with their distances, nearest first."""
import numpy as np

from . import kernels, policy
from .devarray import empty
from .parallel import get_num_threads


class AcklandJonesAnalysis:
    def __init__(self, data, box, verlet_list, distance_list):
        self.data, self.box = data, box
        self.verlet_list, self.distance_list = verlet_list, distance_list

    def compute(self):
        self.aja = empty(self.data.shape[0], np.int32)
        kernels.aja.compute_aja(*policy.positions(self.data), *policy.box_args(self.box), self.verlet_list,
                                self.distance_list, self.aja, get_num_threads())
