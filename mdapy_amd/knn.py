"""k nearest neighbours.  Mirrors ``mdapy.knn.NearestNeighbor`` (src/mdapy/knn.py:17-129)."""
from __future__ import annotations

import numpy as np

from . import _fast_knn
from . import tool_function as tool
from .box import Box
from .devarray import empty
from .frame import Frame
from .parallel import get_num_threads

MAX_K = 24  # knn.py:14


class NearestNeighbor:
    def __init__(self, data: Frame, box: Box, k: int):
        for col in ("x", "y", "z"):
            assert col in data.columns, f"data must contain column {col!r}."
        assert data.shape[0] > 0, "data must contain at least one atom."
        k = int(k)
        assert 1 <= k <= MAX_K, f"k must be in [1, {MAX_K}], got {k}."
        self.data = data
        self.box = box
        self.k = k

    def compute(self):
        data, box = self.data, self.box
        repeat = self._check_repeat_nearest()
        if sum(repeat) != 3:
            self._enlarge_data, self._enlarge_box = tool.replicate(data, box, *repeat)
            box, data = self._enlarge_box, self._enlarge_data
        N = data.shape[0]
        self.indices_py = empty((N, self.k), np.int32)
        self.distances_py = empty((N, self.k), np.float64)
        x, y, z = tool.xyz(data)
        _fast_knn.knn(x, y, z, box.box, box.origin, box.boundary, self.k, self.indices_py, self.distances_py,
                      get_num_threads())

    def _check_repeat_nearest(self):
        """replicate (+3 per periodic axis) until the system holds at least k atoms (knn.py:105-129)"""
        repeat = [1, 1, 1]
        N = self.data.shape[0]
        if self.k > N:
            assert sum(self.box.boundary) > 0, (
                f"Need periodic boundary if you want to query {self.k} neighbors " f"in {N}-atom system."
            )
            while np.prod(repeat) * N < self.k:
                for i in range(3):
                    if self.box.boundary[i] == 1:
                        repeat[i] += 3
        return repeat
