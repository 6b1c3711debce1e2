"""Cluster analysis — the drop-in for ``mdapy.cluster_analysis.ClusterAnalysis`` (src/mdapy/cluster_analysis.py:14-150).

Atoms closer than ``rc`` belong to one cluster; ``rc`` is one number, or one number per pair of types
(``{'1-1': 1.5, '1-2': 1.3}``).  ``particleClusters`` holds the cluster id of every atom (ids start at 1 in the order of
each cluster's smallest atom index), ``cluster_number`` how many there are."""
import numpy as np

from . import kernels
from .devarray import HArray, as_numpy, empty
from .parallel import get_num_threads

_NUMBER = (float, int, np.integer, np.floating)


class ClusterAnalysis:
    def __init__(self, rc, verlet_list, distance_list, neighbor_number, type_list=None):
        self.rc = rc
        self.by_pair = isinstance(rc, dict)
        if self.by_pair:
            if type_list is None:
                raise AssertionError("Need type_list for multi cutoff mode.")
            self.max_rc = max(rc.values())
            # bonds that fail their pair's cutoff are cut out of the list: do that on a copy, the caller keeps his
            verlet_list = verlet_list.copy() if hasattr(verlet_list, "copy") else np.array(as_numpy(verlet_list))
        elif isinstance(rc, _NUMBER):
            self.max_rc = rc
        else:
            raise TypeError("rc should be a positive number, or a dict like {'1-1':1.5, '1-2':1.3}")
        self.verlet_list, self.distance_list, self.neighbor_number = verlet_list, distance_list, neighbor_number
        self.type_list = type_list

    def _pair_table(self):
        """(type a, type b, cutoff) rows, both orders of a mixed pair"""
        rows = []
        for pair, cut in self.rc.items():
            a, b = pair.split("-")
            rows.append((a, b, cut))
            if a != b:
                rows.append((b, a, cut))
        first, second, cuts = zip(*rows)
        return np.array(first, np.int32), np.array(second, np.int32), np.array(cuts, float)

    def _filter_verlet(self):
        types = self.type_list if isinstance(self.type_list, HArray) else np.ascontiguousarray(as_numpy(self.type_list), dtype=np.int32)
        kernels.cluster.filter_by_type(self.verlet_list, self.distance_list, self.neighbor_number, types, *self._pair_table(),
                                       get_num_threads())

    def compute(self):
        self.particleClusters = empty(int(self.verlet_list.shape[0]), np.int32)
        if self.by_pair:
            self._filter_verlet()
            self.cluster_number = kernels.cluster.get_cluster_by_bond(self.verlet_list, self.neighbor_number,
                                                                      self.particleClusters)
        else:
            self.cluster_number = kernels.cluster.get_cluster(self.verlet_list, self.distance_list, self.neighbor_number,
                                                              self.max_rc, self.particleClusters)
