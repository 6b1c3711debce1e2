"""Drop-in for ``mdapy._structure_entropy`` (src/structure_entropy.cpp:110-113)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def calculate_structure_entropy(rc, sigma, use_local_density, volume, distance_list, neighbor_number, entropy, num_t=1):
    """src/structure_entropy.cpp:9"""
    c = Call(distance_list, neighbor_number, entropy)
    N, M = int(distance_list.shape[0]), int(distance_list.shape[1])
    rc_ = _lib.lib().mdh_structure_entropy(float(rc), float(sigma), int(bool(use_local_density)), float(volume),
                                           c.inp(distance_list, f64), c.inp(neighbor_number, i32), N, M,
                                           c.out(entropy, f64, upload=False), c.space, c.stream)
    c.done(rc_)
