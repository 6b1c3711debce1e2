"""lane-kernel probe: plan + per-range time.  python tools/nb_probe.py [cells] [M] [rc/a] [sigma] [reps] [shear]
NB_LIB=<path>: another build of the library (A/B on one box); with mdapy_amd/csrc/libmdapy_amd_stamps.so (make stamps) the
phase time stamps of the kernel are printed as well."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdapy_amd as mp
from mdapy_amd import _lib, _neighbor
from bench import slab_positions, A_CU
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rc = (float(sys.argv[3]) if len(sys.argv) > 3 else 0.854) * A_CU
sigma = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda", 0)
x, y, z, gid = slab_positions(torch, dev, cells, 0, sigma)
n = x.shape[0]
box = mp.Box(np.diag([A_CU * cells] * 3))
if len(sys.argv) > 6:  # sheared box: the same fractional coordinates in a triclinic cell
    sh = float(sys.argv[6]); Lb = A_CU * cells
    Hm = np.array([[Lb, 0, 0], [sh * Lb, Lb, 0], [0.5 * sh * Lb, sh * Lb, Lb]])
    x, y, z = x + sh * y + 0.5 * sh * z, y + sh * z, z
    box = mp.Box(Hm)
if os.environ.get('NB_PBC'):  # e.g. 101: open along the second box vector
    box = mp.Box(box.box, boundary=[int(ch) for ch in os.environ['NB_PBC']], origin=box.origin)
if os.environ.get('NB_UNWRAP'):  # an unwrapped trajectory: every atom handed in a few whole box lengths away (orthogonal box)
    k = int(os.environ['NB_UNWRAP']); Lb = A_CU * cells
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    x = x + Lb * torch.randint(-k, k + 1, (n,), device=dev, generator=gen).double()
    y = y + Lb * torch.randint(-k, k + 1, (n,), device=dev, generator=gen).double()
    z = z + Lb * torch.randint(-k, k + 1, (n,), device=dev, generator=gen).double()
if os.environ.get('NB_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['NB_LIB'])  # A/B against another build of the library
L = _lib.lib()
if os.environ.get('NB_VARIANT'): L.mdh_debug_set_neighbor_variant(int(os.environ['NB_VARIANT']))
verlet = torch.empty((n, M), dtype=torch.int32, device=dev); dist = torch.empty((n, M), dtype=torch.float64, device=dev)
nn = torch.empty((n,), dtype=torch.int32, device=dev)
for _ in range(2):
    _neighbor.build_neighbor(x, y, z, box.box, box.origin, box.boundary, rc, verlet, dist, nn, 1, fill_pads=True)
torch.cuda.synchronize()
plan = (ctypes.c_int * 8)()
L.mdh_prof_reset(); L.mdh_prof_enable(1)
for _ in range(reps):
    _neighbor.build_neighbor(x, y, z, box.box, box.origin, box.boundary, rc, verlet, dist, nn, 1, fill_pads=True)
torch.cuda.synchronize()
L.mdh_prof_enable(0)
L.mdh_debug_neighbor_plan(plan)
buf = ctypes.create_string_buffer(1 << 16); L.mdh_prof_report(buf, len(buf))
print("plan txy,tz,cap,lds,full|tk8<<1|wgs<<2,pop*1000,occ,fresh:", list(plan))
print(buf.value.decode().strip(), "N", n, "env", {k: v for k, v in os.environ.items() if k.startswith("MDH_")}, flush=True)

if os.environ.get("NB_SUM"):  # a fingerprint of the rows: two builds of the library (NB_LIB) must print the same three numbers
    w = torch.arange(1, M + 1, device=dev, dtype=torch.int64)[None, :]
    print("rows fingerprint:", int((verlet.long() * w).sum()), int(nn.long().sum()), float((dist * w).sum()), flush=True)

if hasattr(L, "mdh_debug_lane_stamps"):  # experiment builds only (-DMDH_STAMPS): phase stamps of the first 51200 tiles
    nt = 51200
    st = np.zeros(nt * 8, dtype=np.uint64)
    L.mdh_debug_lane_stamps(ctypes.c_void_p(st.ctypes.data), nt * 8)
    st = st.reshape(nt, 8).astype(np.int64)
    d = np.diff(st, axis=1)
    good = (st[:, 0] > 0) & (d > 0).all(axis=1) & (d < 10**7).all(axis=1)
    print("tiles stamped", int(good.sum()), "mean ticks per phase [cell ranges arrive, block scan, stage, scan, tickets+barrier, writeout, barrier]:", np.round(d[good].mean(axis=0), 0), "total", round(float((st[good, 7] - st[good, 0]).mean())))
    print("median:", np.median(d[good], axis=0), "span of the launch", int(st[good, 7].max() - st[good, 0].min()))
