"""Drop-in for ``mdapy._ptm`` (src/polyhedral_template_matching.cpp:321-338)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def get_ptm(structure, x, y, z, box, origin, boundary, verlet_list, atom_types, rmsd_threshold, output, ptm_indices,
            num_t=1):
    """src/polyhedral_template_matching.cpp:135 — output (N,8) f64, ptm_indices (N,18) i32"""
    _lib.same_rows("get_ptm", len(x), y=y, z=z, verlet_list=verlet_list, output=output, ptm_indices=ptm_indices)
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    has_types = atom_types is not None and len(atom_types) == len(x)
    t = atom_types if has_types else None
    c = Call(x, y, z, verlet_list, t, output, ptm_indices)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_ptm(structure.encode(), c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), N, pb, po, pp,
                             c.inp(verlet_list, i32), M, c.inp(t, i32), float(rmsd_threshold),
                             c.out(output, f64, upload=False), int(output.shape[1]),
                             c.out(ptm_indices, i32, upload=False), int(ptm_indices.shape[1]), c.space, c.stream)
    c.done(rc_)
