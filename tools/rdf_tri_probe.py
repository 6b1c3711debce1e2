"""streaming RDF in a sheared periodic box: tile kernel against the thread-per-atom kernel.  python tools/rdf_tri_probe.py [N] [shear]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdapy_amd import _lib, _rdf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
sh = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
rng = np.random.default_rng(0)
L = (N / 0.055) ** (1 / 3)  # number density of a metallic glass
H = np.array([[L, 0, 0], [sh * L, L, 0], [0.5 * sh * L, sh * L, L]])
frac = rng.random((N, 3))
pos = frac @ H
ty = rng.integers(0, 2, N).astype(np.int32)
dev = torch.device("cuda", 0)
x, y, z = (torch.from_numpy(np.ascontiguousarray(pos[:, k])).to(dev) for k in range(3))
t = torch.from_numpy(ty).to(dev)
out = {}
for variant in (0, 1):
    _lib.lib().mdh_debug_set_rdf_variant(variant)
    for it in range(3):
        g = torch.zeros((2, 2, 200), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(); t0 = time.time()
        _rdf._rdf_streaming(x, y, z, t, H, np.zeros(3), np.array([1, 1, 1], np.int32), g, 8.0, 200, 1)
        torch.cuda.synchronize(); dt = time.time() - t0
    out[variant] = g.cpu().numpy()
    print(f"N={N} shear={sh} variant {variant} ({'tile' if variant == 0 else 'thread-per-atom'}): {dt*1e3:.1f} ms")
_lib.lib().mdh_debug_set_rdf_variant(0)
print("identical counts:", np.array_equal(out[0], out[1]), "pairs", int(out[0].sum()))
