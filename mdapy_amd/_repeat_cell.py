"""Drop-in for ``mdapy._repeat_cell`` (src/repeat_cell.cpp:63-66)."""
import numpy as np

from . import _lib
from .devarray import Call

f64 = np.float64


def repeat_cell(new_pos, old_box, old_pos, nx, ny, nz, num_t=1):
    """src/repeat_cell.cpp:19 — new_pos flat (n_old*nx*ny*nz*3)"""
    ob = np.ascontiguousarray(np.asarray(old_box, dtype=f64).reshape(3, 3))
    c = Call(new_pos, old_pos)
    n_old = int(old_pos.shape[0])
    rc_ = _lib.lib().mdh_repeat_cell(c.out(new_pos, f64, upload=False), ob.ctypes.data, c.inp(old_pos, f64), n_old,
                                     int(nx), int(ny), int(nz), c.space, c.stream)
    c.done(rc_)
