"""Polyhedral template matching — the drop-in for ``mdapy.polyhedral_template_matching.PolyhedralTemplateMatching``
(src/mdapy/polyhedral_template_matching.py:60-167).

``output`` (N, 8): structure type, ordering type, rmsd, interatomic distance, quaternion w, x, y, z;
``ptm_indices`` (N, 18): the neighbour atoms in template order.  Needs the 18 nearest neighbours of every atom, from the
caller or from a search of its own (replica when a periodic direction is thinner than 15 A)."""
import numpy as np

from . import kernels, policy
from .devarray import HArray, full, zeros
from .parallel import get_num_threads

DEPTH = 18
KNOWN = ("fcc", "hcp", "bcc", "ico", "sc", "dcub", "dhex", "graphene", "all", "default")


def _alloy_types(frame):
    """1-based species codes for the alloy orderings: the ``type`` column, else elements in sorted order, else one species"""
    if "type" in frame.columns:
        column = frame["type"]
        held = getattr(column, "_dev", None)
        if isinstance(held, HArray) and held.dtype == np.int32:
            return held  # already in HBM (a file reader's column), as the kernel reads it
        return np.ascontiguousarray(column.to_numpy(), dtype=np.int32)
    if "element" in frame.columns:
        return policy.label_codes(frame["element"].to_numpy())[1] + 1
    return full((frame.shape[0],), 1, np.int32)  # (made where the kernel reads it: a host array is 40 MB over PCIe per 10 M atoms)


class PolyhedralTemplateMatching:
    def __init__(self, structure, data, box, rmsd_threshold=0.1, verlet_list=None):
        if any(part not in KNOWN for part in structure.split("-")):
            raise AssertionError(
                'Structure should in ["fcc", "hcp", "bcc", "ico", "sc","dcub", "dhex", "graphene", "all", "default"].'
            )
        self.structure, self.rmsd_threshold = structure, rmsd_threshold
        self.data, self.box, self.verlet_list = data, box, verlet_list

    def compute(self):
        atoms = self.data.shape[0]
        if policy.hopeless(self.box, atoms, DEPTH):
            # (seven columns here, eight below: the reference's early-out is one short, its callers read columns 0..3 only)
            self.output, self.ptm_indices = np.zeros((atoms, 7)), np.zeros((atoms, DEPTH), np.int32)
            return
        frame, cell, rows = self.data, self.box, self.verlet_list
        if rows is None:
            frame, cell, _ = policy.widened(frame, cell, policy.NEAREST_SPAN)
            rows = policy.nearest_rows(frame, cell, DEPTH)
        atoms = frame.shape[0]
        self.output = zeros((atoms, 8), np.float64)
        self.ptm_indices = zeros((atoms, DEPTH), np.int32)
        kernels.ptm.get_ptm(self.structure, *policy.positions(frame), *policy.box_args(cell), rows, _alloy_types(frame),
                            self.rmsd_threshold, self.output, self.ptm_indices, get_num_threads())
