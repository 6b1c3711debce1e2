"""Stacking faults and twin boundaries in FCC crystals — the drop-in for
``mdapy.identify_fcc_planar_faults.IdentifyFccPlanarFaults`` (src/mdapy/identify_fcc_planar_faults.py:14-90).
``fault_types``: 0 non-hcp, 1 other hcp, 2 intrinsic stacking fault, 3 twin boundary, 4 multi-layer stacking fault,
5 extrinsic stacking fault.  Input: PTM structure types and the 12 template-ordered neighbours of every atom."""
import numpy as np

from . import kernels
from .devarray import as_numpy
from .parallel import get_num_threads

HCP = 2


class IdentifyFccPlanarFaults:
    def __init__(self, structure_types, ptm_indices, cal_esf=True):
        self.structure_types, self.ptm_indices, self.cal_esf = structure_types, ptm_indices, cal_esf

    def compute(self):
        types = np.ascontiguousarray(as_numpy(self.structure_types), dtype=np.int32)
        order = np.ascontiguousarray(as_numpy(self.ptm_indices), dtype=np.int32)
        hcp_atoms = np.flatnonzero(types == HCP).astype(np.int32)
        work = np.zeros((len(hcp_atoms), 12), dtype=np.int32)  # the kernel's table of each hcp atom's hcp neighbours
        self.fault_types = np.zeros_like(types)
        kernels.fccpft.identify_sftb_fcc(hcp_atoms, work, order, types, self.fault_types, self.cal_esf, get_num_threads())
