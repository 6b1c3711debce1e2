import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from mdapy_amd import _neighbor, _lib
import test_gpu_parity as T
for name in ("triclinic_fcc_sheared", "triclinic_dense_blob", "fcc_rattled"):
    _, pos, box, org, bnd = [c for c in T.CASES if c[0] == name][0]
    x, y, z = T._xyz(pos)
    v, d, n = _neighbor.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, 3.3, 1)
    plan = np.zeros(8, np.int32); _lib.lib().mdh_debug_neighbor_plan(plan.ctypes.data)
    print(name, plan.tolist(), np.asarray(n).max())
import ctypes
plan2 = (ctypes.c_int * 8)()
_, pos, box, org, bnd = [c for c in T.CASES if c[0] == "fcc_rattled"][0]
x, y, z = T._xyz(pos)
M = 20
v = np.empty((len(x), M), np.int32); d = np.empty((len(x), M)); n = np.empty(len(x), np.int32)
_neighbor.build_neighbor(x, y, z, box, org, bnd, 3.3, v, d, n, 1, fill_pads=True)
_lib.lib().mdh_debug_neighbor_plan(plan2); print("fixed M", list(plan2))
_neighbor.build_neighbor_without_max_neigh(x, y, z, box, org, bnd, 3.3, 1)
_lib.lib().mdh_debug_neighbor_plan(plan2); print("exact", list(plan2))
print(np.asarray(box), org, bnd)
