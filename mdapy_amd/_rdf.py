"""Drop-in for ``mdapy._rdf`` (src/radial_distribution_function.cpp:319-331)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def _rdf(verlet_list, distance_list, neighbor_number, type_list, g, rc, nbin):
    """src/radial_distribution_function.cpp:22 — g (Nt,Nt,nbin) is accumulated into"""
    c = Call(verlet_list, distance_list, neighbor_number, type_list, g)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_rdf(c.inp(verlet_list, i32), c.inp(distance_list, f64), c.inp(neighbor_number, i32),
                             c.inp(type_list, i32), N, M, c.out(g, f64), int(g.shape[0]), float(rc), int(nbin),
                             c.space, c.stream)
    c.done(rc_)


def _rdf_single_species(verlet_list, distance_list, neighbor_number, g, rc, nbin):
    """src/radial_distribution_function.cpp:56"""
    c = Call(verlet_list, distance_list, neighbor_number, g)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_rdf_single_species(c.inp(verlet_list, i32), c.inp(distance_list, f64),
                                            c.inp(neighbor_number, i32), N, M, c.out(g, f64), float(rc), int(nbin),
                                            c.space, c.stream)
    c.done(rc_)


def _rdf_streaming(x, y, z, type_list, box, origin, boundary, g, rc, nbin, num_t=1):
    """src/radial_distribution_function.cpp:143"""
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, type_list, g)
    rc_ = _lib.lib().mdh_rdf_streaming(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), c.inp(type_list, i32),
                                       int(len(x)), pb, po, pp, c.out(g, f64), int(g.shape[0]), float(rc), int(nbin),
                                       c.space, c.stream)
    c.done(rc_)
