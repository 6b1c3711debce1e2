"""Drop-in for ``mdapy._fccpft`` (src/identify_fcc_planar_faults.cpp:247-249)."""
import numpy as np

from . import _lib
from .devarray import Call

i32 = np.int32


def identify_sftb_fcc(hcp_indices, hcp_neighbors, ptm_indices, structure_types, fault_types, identify_esf, num_t=1):
    """src/identify_fcc_planar_faults.cpp:54 — fault_types (pre-zeroed) receives 1..5 for the HCP atoms"""
    c = Call(hcp_indices, hcp_neighbors, ptm_indices, structure_types, fault_types)
    rc_ = _lib.lib().mdh_identify_sftb_fcc(c.inp(hcp_indices, i32), int(hcp_indices.shape[0]), c.out(hcp_neighbors, i32, upload=False),
                                           c.inp(ptm_indices, i32), c.inp(structure_types, i32), int(structure_types.shape[0]),
                                           c.out(fault_types, i32), int(bool(identify_esf)), c.space, c.stream)
    c.done(rc_)
