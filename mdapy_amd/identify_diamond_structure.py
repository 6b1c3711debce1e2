"""Identify diamond structure.  Mirrors ``mdapy.identify_diamond_structure.IdentifyDiamondStructure``
(src/mdapy/identify_diamond_structure.py:60-125): 0 other, 1 cubic diamond, 2/3 its 1st/2nd neighbours,
4 hexagonal diamond, 5/6 its 1st/2nd neighbours."""
from __future__ import annotations

import numpy as np

from . import _cna
from . import tool_function as tool
from .box import Box
from .devarray import zeros
from .frame import Frame
from .knn import NearestNeighbor
from .parallel import get_num_threads


class IdentifyDiamondStructure:
    def __init__(self, data: Frame, box: Box, verlet_list=None):
        self.data = data
        self.box = box
        self.verlet_list = verlet_list

    def compute(self):
        N = self.data.shape[0]
        if sum(self.box.boundary) == 0 and N <= 4:
            self.pattern = np.zeros(N, dtype=np.int32)
            return
        box, data, verlet_list = self.box, self.data, self.verlet_list
        safe_L = 15
        if self.verlet_list is None:
            repeat = np.ceil(safe_L / self.box.get_thickness()).astype(int)
            for i in range(3):
                if self.box.boundary[i] == 0:
                    repeat[i] = 1
            if sum(repeat) != 3:
                data, box = tool._replicate_pos(data, box, *repeat)
            knn = NearestNeighbor(data, box, 4)
            knn.compute()
            verlet_list = knn.indices_py
        N = data.shape[0]
        self.pattern = zeros(N, np.int32)
        new_verlet_list = zeros((N, 12), np.int32)
        x, y, z = tool.xyz(data)
        _cna.ids(x, y, z, box.box, box.origin, box.boundary, verlet_list, new_verlet_list, self.pattern,
                 get_num_threads())
