"""Drop-in for ``mdapy._atomtemp`` (src/atomic_temperature.cpp:114-117)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def compute_temp(verlet_list, distance_list, vx, vy, vz, mass_list, T, rc, num_t=1):
    """src/atomic_temperature.cpp:9 — velocities in m/s, masses in g/mol"""
    c = Call(verlet_list, distance_list, vx, vy, vz, mass_list, T)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_atomic_temperature(c.inp(verlet_list, i32), c.inp(distance_list, f64), N, M, c.inp(vx, f64),
                                            c.inp(vy, f64), c.inp(vz, f64), c.inp(mass_list, f64),
                                            c.out(T, f64, upload=False), float(rc), c.space, c.stream)
    c.done(rc_)
