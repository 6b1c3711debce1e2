"""Warren-Cowley short-range order.  Mirrors
``mdapy.warren_cowley_parameter.WarrenCowleyParameter`` (src/mdapy/warren_cowley_parameter.py:76-112)."""
from __future__ import annotations

import numpy as np

from . import _wcp
from . import tool_function as tool
from .frame import Frame
from .parallel import get_num_threads


class WarrenCowleyParameter:
    def __init__(self, verlet_list, neighbor_number, data: Frame) -> None:
        self.verlet_list = verlet_list
        self.neighbor_number = neighbor_number
        self.data = data
        if "element" in self.data.columns:  # elements -> index in sorted order (:82-89)
            names, self.type_list = tool.dense_labels(self.data["element"].to_numpy())
            self.ele2type = {j: i for i, j in enumerate(names)}
            self.Ntype = len(self.ele2type)
        else:
            assert "type" in self.data.columns
            self.type_list = (self.data["type"].to_numpy() - 1).astype(np.int32)
            self.Ntype = len(np.unique(self.type_list))
            assert self.type_list.max() + 1 == self.Ntype

    def compute(self) -> None:
        self.WCP = np.zeros((self.Ntype, self.Ntype), float)
        _wcp.get_wcp(self.verlet_list, self.neighbor_number, self.type_list, self.Ntype, self.WCP, get_num_threads())
