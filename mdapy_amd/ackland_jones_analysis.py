"""Ackland-Jones analysis.  Mirrors ``mdapy.ackland_jones_analysis.AcklandJonesAnalysis``
(src/mdapy/ackland_jones_analysis.py:15-120): 0 other, 1 fcc, 2 hcp, 3 bcc, 4 ico."""
from __future__ import annotations

import numpy as np

from . import _aja
from . import tool_function as tool
from .box import Box
from .devarray import empty
from .frame import Frame
from .parallel import get_num_threads


class AcklandJonesAnalysis:
    def __init__(self, data: Frame, box: Box, verlet_list, distance_list) -> None:
        self.data = data
        self.box = box
        self.verlet_list = verlet_list
        self.distance_list = distance_list

    def compute(self) -> None:
        self.aja = empty(self.data.shape[0], np.int32)
        x, y, z = tool.xyz(self.data)
        _aja.compute_aja(x, y, z, self.box.box, self.box.origin, self.box.boundary, self.verlet_list, self.distance_list,
                         self.aja, get_num_threads())
