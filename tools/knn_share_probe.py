import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd.devarray import HArray
from mdapy_amd.frame import Frame
from bench import slab_positions, A_CU
cells=136; dev=torch.device("cuda",0)
x,y,z,_=slab_positions(torch,dev,cells,0,0.05)
n=int(x.shape[0])
def lap(label, fn):
    torch.cuda.synchronize(); t0=time.perf_counter(); fn(); torch.cuda.synchronize(); print(f"{label:40s} {(time.perf_counter()-t0)*1e3:8.2f} ms", flush=True)
for rep in range(3):
    s = mp.System(data=Frame({"x": HArray(x), "y": HArray(y), "z": HArray(z)}), box=mp.Box(np.diag([A_CU*cells]*3)))
    print("pass", rep)
    lap("csp(12)", lambda: s.cal_centro_symmetry_parameter(12))
    lap("adaptive cna", lambda: s.cal_common_neighbor_analysis())
    lap("build_nearest_neighbor(18)", lambda: s.build_nearest_neighbor(18))
    lap("ptm", lambda: s.cal_polyhedral_template_matching("fcc-hcp-bcc"))
    lap("csp(12) again", lambda: s.cal_centro_symmetry_parameter(12))
# parity of shared vs unshared rows
s1 = mp.System(data=Frame({"x": HArray(x), "y": HArray(y), "z": HArray(z)}), box=mp.Box(np.diag([A_CU*cells]*3)))
s1.cal_centro_symmetry_parameter(12); s1.cal_common_neighbor_analysis()
a = s1.data["cna"].to_numpy().copy(); c1 = s1.data["csp"].to_numpy().copy()
os.environ["MDH_KNN_ROWS"] = "0"
