"""Diamond-structure identification — the drop-in for ``mdapy.identify_diamond_structure.IdentifyDiamondStructure``
(src/mdapy/identify_diamond_structure.py:60-125).  ``pattern``: 0 other, 1 cubic diamond, 2 / 3 its first / second
neighbours, 4 hexagonal diamond, 5 / 6 its first / second neighbours.  Works from the 4 nearest neighbours."""
import numpy as np

from . import kernels, policy
from .devarray import zeros
from .parallel import get_num_threads

DEPTH = 4


class IdentifyDiamondStructure:
    def __init__(self, data, box, verlet_list=None):
        self.data, self.box, self.verlet_list = data, box, verlet_list

    def compute(self):
        if policy.hopeless(self.box, self.data.shape[0], DEPTH):
            self.pattern = np.zeros(self.data.shape[0], dtype=np.int32)
            return
        frame, cell, rows = self.data, self.box, self.verlet_list
        if rows is None:
            frame, cell, _ = policy.widened(frame, cell, policy.NEAREST_SPAN)
            rows = policy.nearest_rows(frame, cell, DEPTH)
        atoms = frame.shape[0]
        self.pattern = zeros(atoms, np.int32)
        second_shell = zeros((atoms, 12), np.int32)  # scratch of the kernel: 3 further neighbours of each of the 4
        kernels.cna.ids(*policy.positions(frame), *policy.box_args(cell), rows, second_shell, self.pattern,
                        get_num_threads())
