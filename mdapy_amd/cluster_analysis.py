"""Cluster analysis.  Mirrors ``mdapy.cluster_analysis.ClusterAnalysis`` (src/mdapy/cluster_analysis.py:14-150):
atoms closer than ``rc`` (one number, or one number per type pair ``{'1-1': 1.5, '1-2': 1.3}``) belong to one cluster;
ids start at 1 and follow the smallest atom index of each cluster."""
from __future__ import annotations

from typing import Dict, Optional, Union

import numpy as np

from . import _cluster
from .devarray import HArray, as_numpy, empty
from .parallel import get_num_threads


class ClusterAnalysis:
    def __init__(self, rc: Union[float, int, Dict[str, float]], verlet_list, distance_list, neighbor_number,
                 type_list: Optional[np.ndarray] = None):
        self.rc = rc
        if isinstance(rc, (float, int, np.integer, np.floating)):
            self.max_rc = self.rc
        elif isinstance(rc, dict):
            assert type_list is not None, "Need type_list for multi cutoff mode."
            self.max_rc = max(self.rc.values())
        else:
            raise TypeError("rc should be a positive number, or a dict like {'1-1':1.5, '1-2':1.3}")
        if isinstance(rc, dict):  # the filter edits the list: work on a copy (cluster_analysis.py:93-96)
            self.verlet_list = verlet_list.copy() if hasattr(verlet_list, "copy") else np.array(as_numpy(verlet_list))
        else:
            self.verlet_list = verlet_list
        self.distance_list = distance_list
        self.neighbor_number = neighbor_number
        self.type_list = type_list

    def _filter_verlet(self):
        type1, type2, r = [], [], []
        for key, value in self.rc.items():
            left, right = key.split("-")
            type1.append(left); type2.append(right); r.append(value)
            if left != right:
                type1.append(right); type2.append(left); r.append(value)
        _cluster.filter_by_type(self.verlet_list, self.distance_list, self.neighbor_number,
                                np.ascontiguousarray(as_numpy(self.type_list), dtype=np.int32) if not isinstance(self.type_list, HArray) else self.type_list,
                                np.array(type1, np.int32), np.array(type2, np.int32), np.array(r, float), get_num_threads())

    def compute(self):
        if isinstance(self.rc, dict):
            self._filter_verlet()
        n = int(self.verlet_list.shape[0])
        self.particleClusters = empty(n, np.int32)
        if isinstance(self.rc, dict):
            self.cluster_number = _cluster.get_cluster_by_bond(self.verlet_list, self.neighbor_number, self.particleClusters)
        else:
            self.cluster_number = _cluster.get_cluster(self.verlet_list, self.distance_list, self.neighbor_number,
                                                       self.max_rc, self.particleClusters)
