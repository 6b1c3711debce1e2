"""Centro-symmetry parameter — the drop-in for ``mdapy.centro_symmetry_parameter.CentroSymmetryParameter``
(src/mdapy/centro_symmetry_parameter.py:69-103): sum of the N/2 smallest |r_j + r_k|^2 over the pairs of an atom's N
nearest neighbours; ``csp`` (atoms) f64.  The list comes from the caller, nearest first."""
import numpy as np

from . import kernels, policy
from .devarray import empty
from .parallel import get_num_threads


class CentroSymmetryParameter:
    def __init__(self, data, box, N, verlet_list):
        if not (N > 0 and N % 2 == 0):
            raise AssertionError(f"N must be a positive even number: {N}.")
        self.N = int(N)
        self.data, self.box, self.verlet_list = data, box, verlet_list

    def compute(self):
        self.csp = empty(self.data.shape[0], np.float64)
        kernels.csp.get_csp(*policy.positions(self.data), *policy.box_args(self.box), self.verlet_list, self.N, self.csp,
                            get_num_threads())
