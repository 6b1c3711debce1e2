#!/usr/bin/env python
"""Write an n-atom dump (default 4 M) to /tmp and read it through the HIP tokenizer three times: python tools/reader_probe.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, pandas as pd, torch
import mdapy_amd as mp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
pos = np.random.default_rng(1).random((n, 3)) * 400.0
p = "/tmp/reader_probe.dump"
with open(p, "w") as f:
    f.write(f"ITEM: TIMESTEP\n0\nITEM: NUMBER OF ATOMS\n{n}\nITEM: BOX BOUNDS pp pp pp\n0 400\n0 400\n0 400\nITEM: ATOMS id type x y z\n")
    pd.DataFrame({"id": np.arange(1, n + 1, dtype=np.int32), "type": np.ones(n, np.int32), "x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2]}).to_csv(
        f, sep=" ", header=False, index=False, float_format="%.17g")
size = os.path.getsize(p)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = mp.System(p)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"read {n} atoms, {size / 1e6:.0f} MB in {dt * 1e3:.1f} ms = {size / dt / 1e9:.2f} GB/s, {n / dt / 1e6:.1f} M atoms/s", flush=True)
assert s.data["x"].to_numpy().tobytes() == pos[:, 0].tobytes()
