"""Stacking faults and twin boundaries in FCC crystals.  Mirrors
``mdapy.identify_fcc_planar_faults.IdentifyFccPlanarFaults`` (src/mdapy/identify_fcc_planar_faults.py:14-90):
0 non-hcp, 1 other hcp, 2 intrinsic stacking fault, 3 twin boundary, 4 multi-layer stacking fault, 5 extrinsic SF."""
from __future__ import annotations

import numpy as np

from . import _fccpft
from .devarray import as_numpy
from .parallel import get_num_threads


class IdentifyFccPlanarFaults:
    def __init__(self, structure_types: np.ndarray, ptm_indices: np.ndarray, cal_esf: bool = True):
        self.structure_types = structure_types
        self.ptm_indices = ptm_indices
        self.cal_esf = cal_esf

    def compute(self):
        st = np.ascontiguousarray(as_numpy(self.structure_types), dtype=np.int32)
        hcp_indices = np.where(st == 2)[0].astype(np.int32)
        hcp_neighbors = np.zeros((hcp_indices.shape[0], 12), dtype=np.int32)
        self.fault_types = np.zeros_like(st)
        _fccpft.identify_sftb_fcc(hcp_indices, hcp_neighbors, np.ascontiguousarray(as_numpy(self.ptm_indices), dtype=np.int32), st,
                                  self.fault_types, self.cal_esf, get_num_threads())
