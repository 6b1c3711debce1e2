"""Cutoff neighbor list.  Mirrors ``mdapy.neighbor.Neighbor`` (src/mdapy/neighbor.py:13-142):
same constructor checks, small-box replication, the two kernels (exact width when
``max_neigh`` is None, fixed width otherwise) and the ``ValueError`` on overflow."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _neighbor
from . import tool_function as tool
from .box import Box
from .devarray import empty
from .frame import Frame
from .parallel import get_num_threads


class Neighbor:
    def __init__(self, rc: float, box: Box, data: Frame, max_neigh: Optional[int] = None):
        rc = float(rc)
        assert rc > 0, f"rc must be positive, got {rc}."
        if max_neigh is not None:
            max_neigh = int(max_neigh)
            assert max_neigh > 0, f"max_neigh must be positive, got {max_neigh}."
        for col in ("x", "y", "z"):
            assert col in data.columns, f"data must contain column {col!r}."
        self.rc = rc
        self.box = box
        self.data = data
        self.max_neigh = max_neigh
        self.N = self.data.shape[0]
        assert self.N > 0, "data must contain at least one atom."

    def compute(self):
        repeat = self.box.check_small_box(self.rc)  # neighbor.py:94
        if sum(repeat) != 3:
            self._enlarge_data, self._enlarge_box = tool.replicate(self.data, self.box, *repeat)
            data, box = self._enlarge_data, self._enlarge_box
        else:
            data, box = self.data, self.box
        x, y, z = tool.xyz(data)
        N = data.shape[0]

        if self.max_neigh is None:  # neighbor.py:108-117
            self.verlet_list, self.distance_list, self.neighbor_number = _neighbor.build_neighbor_without_max_neigh(
                x, y, z, box.box, box.origin, box.boundary, self.rc, get_num_threads()
            )
            return

        # fixed width (neighbor.py:125-134): the kernel writes the -1 / rc+1 pads itself
        self.verlet_list = empty((N, self.max_neigh), np.int32)
        self.distance_list = empty((N, self.max_neigh), np.float64)
        self.neighbor_number = empty((N,), np.int32)
        self._fill(x, y, z, box)
        real_max = int(self.neighbor_number.max(initial=0))
        if real_max > self.max_neigh:  # neighbor.py:135-142
            raise ValueError(
                f"max_neigh={self.max_neigh} is too small: at least one "
                f"atom has {real_max} neighbors within rc={self.rc}. "
                f"Re-run with max_neigh>={real_max} (or omit max_neigh "
                "to let mdapy size the buffer automatically)."
            )

    def _fill(self, x, y, z, box):
        if isinstance(self.verlet_list, np.ndarray):  # host buffers (backend patched in CPU tests): reference init
            self.verlet_list.fill(-1)
            self.distance_list.fill(self.rc + 1.0)
            self.neighbor_number.fill(0)
            _neighbor.build_neighbor(x, y, z, box.box, box.origin, box.boundary, self.rc, self.verlet_list,
                                     self.distance_list, self.neighbor_number, get_num_threads())
        else:
            _neighbor.build_neighbor(x, y, z, box.box, box.origin, box.boundary, self.rc, self.verlet_list,
                                     self.distance_list, self.neighbor_number, get_num_threads(), fill_pads=True)
