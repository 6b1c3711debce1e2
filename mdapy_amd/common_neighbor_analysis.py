"""Common neighbour analysis.  Mirrors ``mdapy.common_neighbor_analysis.CommonNeighborAnalysis``
(src/mdapy/common_neighbor_analysis.py:64-154): 0 other, 1 FCC, 2 HCP, 3 BCC, 4 ICO."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _cna
from . import tool_function as tool
from .box import Box
from .devarray import zeros
from .frame import Frame
from .knn import NearestNeighbor
from .neighbor import Neighbor
from .parallel import get_num_threads


class CommonNeighborAnalysis:
    def __init__(self, data: Frame, box: Box, verlet_list=None, neighbor_number=None, rc: Optional[float] = None):
        self.data = data
        self.box = box
        self.verlet_list = verlet_list
        self.neighbor_number = neighbor_number
        if rc is not None:
            assert rc > 0
        self.rc = rc
        self.pattern = None

    def compute(self):
        N = self.data.shape[0]
        if sum(self.box.boundary) == 0 and N <= 14:  # :88-91
            self.pattern = np.zeros(N, dtype=np.int32)
            return
        box, data = self.box, self.data
        verlet_list, neighbor_number = self.verlet_list, self.neighbor_number
        wrap_pos_L = 15
        if self.verlet_list is None:
            repeat = np.ceil(wrap_pos_L / self.box.get_thickness()).astype(int)
            for i in range(3):
                if self.box.boundary[i] == 0:
                    repeat[i] = 1
            if sum(repeat) != 3:
                data, box = tool._replicate_pos(data, box, *repeat)
            if self.rc is None:
                knn = NearestNeighbor(data, box, 14)
                knn.compute()
                verlet_list = knn.indices_py
            else:
                repeat = box.check_small_box(self.rc)
                if sum(repeat) != 3:
                    data, box = tool._replicate_pos(data, box, *repeat)
                neigh = Neighbor(self.rc, box, data)
                neigh.compute()
                verlet_list = neigh.verlet_list
                neighbor_number = neigh.neighbor_number
        else:
            assert neighbor_number is not None or self.rc is None
        N = data.shape[0]
        self.pattern = zeros(N, np.int32)  # the kernels rely on the pre-zeroing (:128)
        x, y, z = tool.xyz(data)
        if self.rc is None:
            _cna.acna(x, y, z, box.box, box.origin, box.boundary, verlet_list, self.pattern, get_num_threads())
        else:
            _cna.fcna(x, y, z, box.box, box.origin, box.boundary, verlet_list, neighbor_number, self.pattern,
                      self.rc, get_num_threads())
