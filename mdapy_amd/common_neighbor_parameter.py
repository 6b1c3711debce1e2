"""Common neighbour parameter — the drop-in for ``mdapy.common_neighbor_parameter.CommonNeighborParameter``
(src/mdapy/common_neighbor_parameter.py:14-110); ``cnp`` (atoms) f64 over a cutoff list."""
import numpy as np

from . import kernels, policy
from .devarray import empty
from .parallel import get_num_threads


class CommonNeighborParameter:
    def __init__(self, data, box, rc, verlet_list, distance_list, neighbor_number):
        assert rc > 0
        self.rc = rc
        self.data, self.box = data, box
        self.verlet_list, self.distance_list, self.neighbor_number = verlet_list, distance_list, neighbor_number

    def compute(self):
        frame = self.data
        self.cnp = empty(frame.shape[0], np.float64)
        lists = (self.verlet_list, self.distance_list, self.neighbor_number)
        kernels.cnp.compute_cnp(*policy.positions(frame), *policy.box_args(self.box), *lists, self.cnp, self.rc, get_num_threads())
