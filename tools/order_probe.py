#!/usr/bin/env python
"""The headline step (build_neighbor M = 16 + fixed CNA through the C ABI: ONE call, mdh_build_neighbor_fcna; PROBE_TWO_CALLS=1: the two calls
it replaces) on the SAME atoms handed in in different orders.
Usage: python tools/order_probe.py ORDER [cells] [steps]   ORDER in lattice | shuffled | blocks | poly | poly_shuffled
(run one ORDER per process under `rocprofv3 --kernel-trace --stats` for the per-kernel split; tools/measure_r05.sh section order)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mdapy_amd as mp
from mdapy_amd import _lib, _neighbor, _cna
from bench import slab_positions, A_CU, RC

order = sys.argv[1] if len(sys.argv) > 1 else "lattice"
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 136
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
M = 16
dev = torch.device("cuda", 0)
L = _lib.lib()
if order.startswith("poly"):
    rng = np.random.default_rng(2024)
    grains = max(4, int(round((A_CU * cells) ** 3 / 2.3e6)))
    unit = mp.build_crystal("Cu", "fcc", A_CU)
    poly = mp.CreatePolycrystal(unit, box=A_CU * cells, seed_number=grains, seed_position=rng.random((grains, 3)) * A_CU * cells,
                                theta_list=rng.uniform(-180, 180, (grains, 3)), metal_overlap_dis=2.0).compute(verbose=False)
    x, y, z = (torch.from_numpy(np.ascontiguousarray(poly.data[c].to_numpy())).to(dev) for c in "xyz")
    bx = (poly.box.box, poly.box.origin, poly.box.boundary)
else:
    x, y, z, _ = slab_positions(torch, dev, cells, 0, 0.0)
    b = mp.Box(np.diag([A_CU * cells] * 3))
    bx = (b.box, b.origin, b.boundary)
n = int(x.shape[0])
gen = torch.Generator(device=dev); gen.manual_seed(11)
if order.endswith("shuffled"):
    perm = torch.randperm(n, device=dev, generator=gen)
    x, y, z = (c[perm].contiguous() for c in (x, y, z))
elif order == "blocks":  # an id-sorted dump of a run that started as 4096-atom blocks dealt out at random (locality inside a block only)
    nb = n // 4096
    perm = (torch.randperm(nb, device=dev, generator=gen)[:, None] * 4096 + torch.arange(4096, device=dev)[None, :]).reshape(-1)
    perm = torch.cat([perm, torch.arange(nb * 4096, n, device=dev)])
    x, y, z = (c[perm].contiguous() for c in (x, y, z))
verlet = torch.empty((n, M), dtype=torch.int32, device=dev); dist = torch.empty((n, M), dtype=torch.float64, device=dev)
nn = torch.empty((n,), dtype=torch.int32, device=dev); pat = torch.zeros((n,), dtype=torch.int32, device=dev)


two_calls = os.environ.get("PROBE_TWO_CALLS", "") == "1"  # the step as mdh_build_neighbor + mdh_fcna (the headline until the one-pass form)


def step():
    pat.zero_()
    if two_calls:
        _neighbor.build_neighbor(x, y, z, *bx, RC, verlet, dist, nn, 1, fill_pads=True)
        _cna.fcna(x, y, z, *bx, verlet, nn, pat, RC, 1)
    else:
        _neighbor.build_neighbor_fcna(x, y, z, *bx, RC, verlet, dist, nn, pat, 1, fill_pads=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
L.mdh_prof_reset(); L.mdh_prof_enable(1)
for _ in range(5):
    step()
torch.cuda.synchronize()
L.mdh_prof_enable(0)
buf = ctypes.create_string_buffer(1 << 16); L.mdh_prof_report(buf, len(buf))
k = {ln.split()[0]: round(float(ln.split()[2]) / int(ln.split()[1]), 4) for ln in buf.value.decode().strip().splitlines()}
print(f"order={order} form={'two calls' if two_calls else 'one call'} N={n} ms_per_step={ms:.4f} ranges_ms={k} fcc={int((pat == 1).sum())} nn_max={int(nn.max())}", flush=True)
