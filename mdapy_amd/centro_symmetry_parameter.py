"""Centro-symmetry parameter.  Mirrors ``mdapy.centro_symmetry_parameter.CentroSymmetryParameter``
(src/mdapy/centro_symmetry_parameter.py:69-103)."""
from __future__ import annotations

import numpy as np

from . import _csp
from . import tool_function as tool
from .box import Box
from .devarray import empty
from .frame import Frame
from .parallel import get_num_threads


class CentroSymmetryParameter:
    def __init__(self, data: Frame, box: Box, N: int, verlet_list) -> None:
        self.data = data
        self.box = box
        assert N % 2 == 0 and N > 0, f"N must be a positive even number: {N}."
        self.N = int(N)
        self.verlet_list = verlet_list

    def compute(self) -> None:
        self.csp = empty(self.data.shape[0], np.float64)
        x, y, z = tool.xyz(self.data)
        _csp.get_csp(x, y, z, self.box.box, self.box.origin, self.box.boundary, self.verlet_list, self.N, self.csp,
                     get_num_threads())
