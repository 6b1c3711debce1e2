"""Pair-entropy fingerprint — the drop-in for ``mdapy.structure_entropy.StructureEntropy``
(src/mdapy/structure_entropy.py:14-130): ``entropy`` per atom from the distances of a cutoff list, optionally its
neighbourhood average ``entropy_ave`` within ``average_rc``."""
import numpy as np

from . import kernels
from .devarray import empty
from .parallel import get_num_threads


class StructureEntropy:
    def __init__(self, box, verlet_list, distance_list, neighbor_number, rc, sigma, use_local_density, average_rc=0.0):
        self.box = box
        self.verlet_list, self.distance_list, self.neighbor_number = verlet_list, distance_list, neighbor_number
        self.rc, self.sigma = rc, sigma
        self.use_local_density, self.average_rc = use_local_density, average_rc

    def compute(self):
        rows, gaps, counts = self.verlet_list, self.distance_list, self.neighbor_number
        atoms = int(rows.shape[0])
        threads = get_num_threads()
        self.entropy = empty(atoms, np.float64)
        kernels.structure_entropy.calculate_structure_entropy(self.rc, self.sigma, self.use_local_density, self.box.volume, gaps,
                                                              counts, self.entropy, threads)
        if not self.average_rc > 0:
            return
        if self.average_rc > self.rc:
            raise AssertionError("average_rc should be smaller than rc.")
        self.entropy_ave = empty(atoms, np.float64)
        kernels.neighbor.average_by_neighbor(self.average_rc, rows, gaps, counts, self.entropy, self.entropy_ave, True, threads)
