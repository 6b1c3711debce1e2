"""HBM-resident calls issued from several host threads, each on its own HIP stream: the scratch cache hands a block that
one stream has just released to a call on another stream only after that stream's work on it (runtime.hip, DoneEvent)."""
import threading

import numpy as np
import pytest

from mdapy_amd import _neighbor
from mdapy_amd.build_lattice import lattice_positions
from oracle import oracle as O

pytestmark = pytest.mark.gpu

ORG0 = np.zeros(3)
PBC = np.array([1, 1, 1], dtype=np.int32)


def _rattled(n, seed):
    pos, box = lattice_positions("fcc", 3.615, n, n, n)
    pos = pos + np.random.default_rng(seed).normal(0.0, 0.05, pos.shape)
    return [np.ascontiguousarray(pos[:, k]) for k in range(3)], box


def test_two_streams_share_the_scratch_cache():
    import torch

    rc, M = 3.0, 16
    jobs = [_rattled(12, 1), _rattled(12, 2)]  # equal sizes: every released block fits the other thread's requests
    want = []
    for (x, y, z), box in jobs:
        v = np.empty((len(x), M), np.int32)
        d = np.empty((len(x), M))
        n = np.empty(len(x), np.int32)
        O.build_neighbor(x, y, z, box, ORG0, PBC, rc, v, d, n, 4)
        want.append((v, n))
    bad = []

    def worker(k):
        (x, y, z), box = jobs[k]
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            tx, ty, tz = (torch.from_numpy(a).cuda() for a in (x, y, z))
            for it in range(40):
                v = torch.empty((len(x), M), dtype=torch.int32, device="cuda")
                d = torch.empty((len(x), M), dtype=torch.float64, device="cuda")
                n = torch.empty(len(x), dtype=torch.int32, device="cuda")
                _neighbor.build_neighbor(tx, ty, tz, box, ORG0, PBC, rc, v, d, n, 1)
                stream.synchronize()
                if not (np.array_equal(n.cpu().numpy(), want[k][1])):
                    bad.append((k, it, "count"))
                    return
                got = v.cpu().numpy()
                live = np.arange(M)[None, :] < want[k][1][:, None]
                if not np.array_equal(got[live], want[k][0][live]):
                    bad.append((k, it, "ids"))
                    return

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad, bad
