"""Polyhedral template matching.  Mirrors ``mdapy.polyhedral_template_matching.PolyhedralTemplateMatching``
(src/mdapy/polyhedral_template_matching.py:60-167).  Output columns: structure type, ordering type, rmsd,
interatomic distance, quaternion w,x,y,z; ``ptm_indices`` = template-ordered neighbour atom ids."""
from __future__ import annotations

import numpy as np

from . import _ptm
from . import tool_function as tool
from .box import Box
from .devarray import zeros
from .frame import Frame
from .knn import NearestNeighbor
from .parallel import get_num_threads

_STRUCTURES = ["fcc", "hcp", "bcc", "ico", "sc", "dcub", "dhex", "graphene", "all", "default"]


class PolyhedralTemplateMatching:
    def __init__(self, structure: str, data: Frame, box: Box, rmsd_threshold: float = 0.1, verlet_list=None):
        self.structure = structure
        self.data = data
        self.box = box
        self.rmsd_threshold = rmsd_threshold
        self.verlet_list = verlet_list
        for i in self.structure.split("-"):
            assert i in _STRUCTURES, (
                'Structure should in ["fcc", "hcp", "bcc", "ico", "sc","dcub", "dhex", "graphene", "all", "default"].'
            )

    def compute(self) -> None:
        N = self.data.shape[0]
        if sum(self.box.boundary) == 0 and N <= 18:  # :119-123 (7 columns in this early-out, as in the reference)
            self.output = np.zeros((N, 7))
            self.ptm_indices = np.zeros((N, 18), np.int32)
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
            knn = NearestNeighbor(data, box, 18)
            knn.compute()
            verlet_list = knn.indices_py
        N = data.shape[0]
        self.output = zeros((N, 8), np.float64)
        self.ptm_indices = zeros((N, 18), np.int32)
        if "type" in data.columns:  # :144-152
            type_list = np.ascontiguousarray(data["type"].to_numpy(), dtype=np.int32)
        elif "element" in data.columns:
            _, dense = tool.dense_labels(data["element"].to_numpy())  # sorted element names -> 1, 2, ...
            type_list = dense + 1
        else:
            type_list = np.ones(N, np.int32)
        x, y, z = tool.xyz(data)
        _ptm.get_ptm(self.structure, x, y, z, box.box, box.origin, box.boundary, verlet_list, type_list,
                     self.rmsd_threshold, self.output, self.ptm_indices, get_num_threads())
