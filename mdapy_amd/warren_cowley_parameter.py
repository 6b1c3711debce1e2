"""Warren-Cowley short-range order — the drop-in for ``mdapy.warren_cowley_parameter.WarrenCowleyParameter``
(src/mdapy/warren_cowley_parameter.py:76-112): ``WCP[a, b] = 1 - Z_ab / (c_b Z_a)`` over a cutoff list.  Species are the
elements in sorted order (``ele2type`` maps a name to its row) or, without an element column, ``type - 1``."""
import numpy as np

from . import kernels, policy
from .parallel import get_num_threads


class WarrenCowleyParameter:
    def __init__(self, verlet_list, neighbor_number, data):
        self.verlet_list, self.neighbor_number, self.data = verlet_list, neighbor_number, data
        if "element" in data.columns:
            names, self.type_list = policy.label_codes(data["element"].to_numpy())
            self.ele2type = dict(zip(names, range(len(names))))
            self.Ntype = len(names)
            return
        assert "type" in data.columns
        present, codes = policy.label_codes(data["type"].to_numpy(), device_ok=True)  # (cached per immutable column, with its copy in HBM)
        kinds = len(present)
        assert present == list(range(1, kinds + 1))  # types must be 1..Ntype without gaps: then code == type - 1
        self.type_list, self.Ntype = codes, kinds

    def compute(self):
        kinds = self.Ntype
        self.WCP = np.zeros((kinds, kinds), float)
        kernels.wcp.get_wcp(self.verlet_list, self.neighbor_number, self.type_list, kinds, self.WCP, get_num_threads())
