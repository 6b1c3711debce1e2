"""Common neighbour parameter.  Mirrors ``mdapy.common_neighbor_parameter.CommonNeighborParameter``
(src/mdapy/common_neighbor_parameter.py:14-110)."""
from __future__ import annotations

import numpy as np

from . import _cnp
from . import tool_function as tool
from .box import Box
from .devarray import empty
from .frame import Frame
from .parallel import get_num_threads


class CommonNeighborParameter:
    def __init__(self, data: Frame, box: Box, rc: float, verlet_list, distance_list, neighbor_number) -> None:
        self.data = data
        self.box = box
        self.rc = rc
        assert rc > 0
        self.verlet_list = verlet_list
        self.distance_list = distance_list
        self.neighbor_number = neighbor_number

    def compute(self) -> None:
        self.cnp = empty(self.data.shape[0], np.float64)
        x, y, z = tool.xyz(self.data)
        _cnp.compute_cnp(x, y, z, self.box.box, self.box.origin, self.box.boundary, self.verlet_list, self.distance_list,
                         self.neighbor_number, self.cnp, self.rc, get_num_threads())
