"""Voronoi cell volume / face count / cavity radius.  Mirrors ``mdapy.voronoi.Voronoi.get_volume``
(src/mdapy/voronoi.py:160-215); Voronoi neighbour lists (``get_neighbor``) are not built yet."""
from __future__ import annotations

import numpy as np

from . import _voronoi
from . import tool_function as tool
from .box import Box
from .devarray import as_numpy
from .frame import Frame
from .parallel import get_num_threads


class Voronoi:
    def __init__(self, box: Box, data: Frame):
        self.box = box
        self.data = data

    def get_volume(self):
        n = self.data.shape[0]
        volume = np.zeros(n)
        neighbor_number = np.zeros(n, np.int32)
        cavity_radius = np.zeros(n)
        x, y, z = (np.ascontiguousarray(as_numpy(a), dtype=np.float64) for a in tool.xyz(self.data))
        if self.box.triclinic:
            b = self.box.box
            need_rotation = bool(abs(b[0, 1]) > 1e-10 or abs(b[0, 2]) > 1e-10 or abs(b[1, 2]) > 1e-10 or b[0, 0] < 0
                                 or b[1, 1] < 0 or b[2, 2] < 0)
            box, rotation = self.box.align_to_lammps_box()
            bb = box.box.copy()
            for i in range(3):
                if box.boundary[i] == 0:
                    bb[i] *= 3
            _voronoi.get_voronoi_volume_number_radius_tri(x, y, z, bb, box.origin, box.boundary, rotation, volume,
                                                          neighbor_number, cavity_radius, need_rotation, get_num_threads())
        else:
            _voronoi.get_voronoi_volume_number_radius(x, y, z, self.box.box, self.box.origin, self.box.boundary, volume,
                                                      neighbor_number, cavity_radius, get_num_threads())
        return volume, neighbor_number, cavity_radius
