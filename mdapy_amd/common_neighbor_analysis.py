"""Common neighbour analysis — the drop-in for ``mdapy.common_neighbor_analysis.CommonNeighborAnalysis``
(src/mdapy/common_neighbor_analysis.py:64-154).  ``pattern``: 0 other, 1 FCC, 2 HCP, 3 BCC, 4 ICO.

``rc`` given: fixed-cutoff variant over a cutoff list (atoms with 12 or 14 neighbours are classified);
``rc=None``: adaptive variant over the 14 nearest neighbours.  A list handed in by the caller is used as it is; without
one the class searches itself, on a replica if a periodic direction is thinner than 15 A."""
import numpy as np

from . import kernels, policy
from .devarray import zeros
from .neighbor import Neighbor
from .parallel import get_num_threads

ADAPTIVE_DEPTH = 14


class CommonNeighborAnalysis:
    def __init__(self, data, box, verlet_list=None, neighbor_number=None, rc=None):
        if rc is not None:
            assert rc > 0
        self.data, self.box, self.rc = data, box, rc
        self.verlet_list, self.neighbor_number = verlet_list, neighbor_number
        self.pattern = None

    def _own_lists(self):
        """(frame, box, rows, counts) from a search of our own"""
        frame, cell, _ = policy.widened(self.data, self.box, policy.NEAREST_SPAN)
        if self.rc is None:
            return frame, cell, policy.nearest_rows(frame, cell, ADAPTIVE_DEPTH), None
        frame, cell, _ = policy.widened(frame, cell, 2.0 * self.rc)
        found = Neighbor(self.rc, cell, frame)
        found.compute(label=True)  # (the labels in the pass that builds the list, where that applies)
        self._labels = found.pattern
        return frame, cell, found.verlet_list, found.neighbor_number

    def compute(self):
        if policy.hopeless(self.box, self.data.shape[0], ADAPTIVE_DEPTH):
            self.pattern = np.zeros(self.data.shape[0], dtype=np.int32)
            return
        self._labels = None
        if self.verlet_list is None:
            frame, cell, rows, counts = self._own_lists()
            if self._labels is not None and self._labels.shape[0] == frame.shape[0]:
                self.pattern = self._labels
                return
        else:
            frame, cell, rows, counts = self.data, self.box, self.verlet_list, self.neighbor_number
            assert counts is not None or self.rc is None
        self.pattern = zeros(frame.shape[0], np.int32)  # the kernels only ever raise a label: start from "other"
        where = (*policy.positions(frame), *policy.box_args(cell), rows)
        if self.rc is None:
            kernels.cna.acna(*where, self.pattern, get_num_threads())
        else:
            kernels.cna.fcna(*where, counts, self.pattern, self.rc, get_num_threads())
