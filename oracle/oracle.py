"""ctypes front end of the CPU oracle (oracle/mdapy_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the mdapy_amd product package.

The functions mirror the Python-visible signatures of the reference's nanobind
modules so parity tests read like the reference's own tests:
caller-allocated, caller-initialised numpy outputs and a trailing ``num_t``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("MDAPY_ORACLE_SO") or os.path.join(_HERE, "libmdapy_oracle.so")  # (MDAPY_ORACLE_SO: the sanitizer build, `make -C oracle asan`)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mdapy_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a, dtype):
    """pointer to a C-contiguous array of the exact dtype (writable outputs must not be copied)."""
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.c_contiguous, (a.dtype, dtype)
    return a.ctypes.data_as(C.c_void_p)


def _ro(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _boxargs(box, origin, boundary):
    b = _ro(box, np.float64).reshape(9)
    o = _ro(origin, np.float64)
    p = _ro(boundary, np.int32)
    return b, o, p


def _rows(what, n, **arrays):
    """the reference indexes per-atom arguments with len(x) unchecked (UB on a mismatch); the oracle refuses instead of
    crashing the test process"""
    for name, a in arrays.items():
        if a is not None and int(np.shape(a)[0]) != int(n):
            raise ValueError(f"{what}: {name} has {int(np.shape(a)[0])} rows for {int(n)} atoms")


def _chk(rc):
    if rc == -1:
        raise RuntimeError("The volume of the box is zero.")
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}")


i64 = C.c_int64
dbl = C.c_double
cint = C.c_int


# ---------------------------------------------------------------- _neighbor
def build_neighbor(x, y, z, box, origin, boundary, rc, verlet_list, distance_list, neighbor_number, num_t=1):
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    M = 0 if verlet_list is None else verlet_list.shape[1]
    _chk(lib().orc_build_neighbor(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(x.shape[0]),
                                  _p(b, np.float64), _p(o, np.float64), _p(p, np.int32), dbl(rc),
                                  _p(verlet_list, np.int32), _p(distance_list, np.float64),
                                  _p(neighbor_number, np.int32), i64(M), cint(num_t)))


def build_neighbor_without_max_neigh(x, y, z, box, origin, boundary, rc, num_t=1):
    N = len(x)
    nn = np.zeros(N, np.int32)
    build_neighbor(x, y, z, box, origin, boundary, rc, None, None, nn, num_t)
    M = max(int(nn.max(initial=0)), 1)
    v = np.full((N, M), -1, np.int32)
    d = np.full((N, M), rc + 1.0, np.float64)
    build_neighbor(x, y, z, box, origin, boundary, rc, v, d, nn, num_t)
    return v, d, nn


def sort_verlet_by_distance(verlet_list, distance_list, sortNum, num_t=1):
    N, M = verlet_list.shape
    lib().orc_sort_verlet_by_distance(_p(verlet_list, np.int32), _p(distance_list, np.float64), i64(N), i64(M),
                                      cint(sortNum), cint(num_t))


def wrap_positions(x, y, z, box, origin, boundary, num_t=1):
    b, o, p = _boxargs(box, origin, boundary)
    _chk(lib().orc_wrap_positions(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(x.shape[0]),
                                  _p(b, np.float64), _p(o, np.float64), _p(p, np.int32), cint(num_t)))


def average_by_neighbor(rc, verlet_list, distance_list, neighbor_number, value, value_ave, include_self, num_t=1):
    N, M = verlet_list.shape
    value = _ro(value, np.float64)
    lib().orc_average_by_neighbor(dbl(rc), _p(_ro(verlet_list, np.int32), np.int32),
                                  _p(_ro(distance_list, np.float64), np.float64),
                                  _p(_ro(neighbor_number, np.int32), np.int32), i64(N), i64(M),
                                  _p(value, np.float64), _p(value_ave, np.float64), cint(bool(include_self)),
                                  cint(num_t))


# --------------------------------------------------------------------- _cna
def fcna(x, y, z, box, origin, boundary, verlet_list, neighbor_number, pattern, rc, num_t=1):
    _rows("fcna", len(x), verlet_list=verlet_list, neighbor_number=neighbor_number)
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    v = _ro(verlet_list, np.int32)
    nn = _ro(neighbor_number, np.int32)
    _chk(lib().orc_fcna(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(x.shape[0]),
                        _p(b, np.float64), _p(o, np.float64), _p(p, np.int32), _p(v, np.int32), i64(v.shape[1]),
                        _p(nn, np.int32), _p(pattern, np.int32), dbl(rc), cint(num_t)))


def acna(x, y, z, box, origin, boundary, verlet_list, pattern, num_t=1):
    _rows("acna", len(x), verlet_list=verlet_list)
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    v = _ro(verlet_list, np.int32)
    assert v.shape[1] >= 14
    _chk(lib().orc_acna(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(x.shape[0]),
                        _p(b, np.float64), _p(o, np.float64), _p(p, np.int32), _p(v, np.int32), i64(v.shape[1]),
                        _p(pattern, np.int32), cint(num_t)))


def ids(x, y, z, box, origin, boundary, verlet_list, new_verlet_list, pattern, num_t=1):
    _rows("ids", len(x), verlet_list=verlet_list)
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    v = _ro(verlet_list, np.int32)
    _chk(lib().orc_ids(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(x.shape[0]),
                       _p(b, np.float64), _p(o, np.float64), _p(p, np.int32), _p(v, np.int32), i64(v.shape[1]),
                       _p(new_verlet_list, np.int32), _p(pattern, np.int32), cint(num_t)))


# --------------------------------------------------------------------- _csp
def get_csp(x, y, z, box, origin, boundary, verlet_list, N, csp, num_t=1):
    _rows("get_csp", len(x), verlet_list=verlet_list)
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    v = _ro(verlet_list, np.int32)
    _chk(lib().orc_csp(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(x.shape[0]),
                       _p(b, np.float64), _p(o, np.float64), _p(p, np.int32), _p(v, np.int32), i64(v.shape[1]),
                       cint(N), _p(csp, np.float64), cint(num_t)))


# --------------------------------------------------------------------- _sbo
def get_sq(x, y, z, box, origin, boundary, verlet_list, distance_list, neighbor_number, weight, llist, nnn, lmax,
           wl, wlhat, average, use_voronoi, rc, use_weight, qlm_r, qlm_i, qnarray, num_t=1):
    _rows("get_sq", len(x), verlet_list=verlet_list, neighbor_number=neighbor_number)
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    v = _ro(verlet_list, np.int32)
    d = _ro(distance_list, np.float64)
    nn = _ro(neighbor_number, np.int32)
    w = _ro(weight, np.float64) if use_weight else None
    ll = _ro(llist, np.int32)
    _chk(lib().orc_get_sq(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(x.shape[0]),
                          _p(b, np.float64), _p(o, np.float64), _p(p, np.int32), _p(v, np.int32),
                          _p(d, np.float64), i64(v.shape[1]), _p(nn, np.int32), _p(w, np.float64),
                          _p(ll, np.int32), cint(len(ll)), cint(nnn), cint(lmax), cint(bool(wl)), cint(bool(wlhat)),
                          cint(bool(average)), cint(bool(use_voronoi)), dbl(rc), cint(bool(use_weight)),
                          _p(qlm_r, np.float64), _p(qlm_i, np.float64), _p(qnarray, np.float64), cint(num_t)))


def identifySolidLiquid(Q6index, Q6, verlet_list, distance_list, neighbor_number, qlm_r, qlm_i, threshold, n_bond,
                        solidliquid, nbond, use_voronoi, nnn, rc, num_t=1):
    v = _ro(verlet_list, np.int32)
    d = _ro(distance_list, np.float64)
    nn = _ro(neighbor_number, np.int32)
    q6 = _ro(Q6, np.float64)
    qr, qi = _ro(qlm_r, np.float64), _ro(qlm_i, np.float64)
    lib().orc_identify_solid_liquid(cint(Q6index), _p(q6, np.float64), _p(v, np.int32), _p(d, np.float64),
                                    _p(nn, np.int32), i64(v.shape[0]), i64(v.shape[1]), _p(qr, np.float64),
                                    _p(qi, np.float64), cint(qr.shape[1]), cint(qr.shape[2]), dbl(threshold),
                                    cint(n_bond), _p(solidliquid, np.int32), _p(nbond, np.int32),
                                    cint(bool(use_voronoi)), cint(nnn), dbl(rc), cint(num_t))


# --------------------------------------------------------------------- _rdf
def _rdf(verlet_list, distance_list, neighbor_number, type_list, g, rc, nbin):
    v = _ro(verlet_list, np.int32)
    d = _ro(distance_list, np.float64)
    nn = _ro(neighbor_number, np.int32)
    t = _ro(type_list, np.int32)
    lib().orc_rdf(_p(v, np.int32), _p(d, np.float64), _p(nn, np.int32), _p(t, np.int32), i64(v.shape[0]),
                  i64(v.shape[1]), _p(g, np.float64), cint(g.shape[0]), dbl(rc), cint(nbin))


def _rdf_single_species(verlet_list, distance_list, neighbor_number, g, rc, nbin):
    v = _ro(verlet_list, np.int32)
    d = _ro(distance_list, np.float64)
    nn = _ro(neighbor_number, np.int32)
    lib().orc_rdf_single(_p(v, np.int32), _p(d, np.float64), _p(nn, np.int32), i64(v.shape[0]), i64(v.shape[1]),
                         _p(g, np.float64), dbl(rc), cint(nbin))


def _rdf_streaming(x, y, z, type_list, box, origin, boundary, g, rc, nbin, num_t=1):
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    t = _ro(type_list, np.int32)
    _chk(lib().orc_rdf_streaming(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), _p(t, np.int32),
                                 i64(x.shape[0]), _p(b, np.float64), _p(o, np.float64), _p(p, np.int32),
                                 _p(g, np.float64), cint(g.shape[0]), dbl(rc), cint(nbin), cint(num_t)))


# --------------------------------------------------------------------- _wcp
def get_wcp(verlet_list, neighbor_number, type_list, Ntype, WCP, num_t=1):
    v = _ro(verlet_list, np.int32)
    nn = _ro(neighbor_number, np.int32)
    t = _ro(type_list, np.int32)
    lib().orc_wcp(_p(v, np.int32), _p(nn, np.int32), _p(t, np.int32), i64(v.shape[0]), i64(v.shape[1]), cint(Ntype),
                  _p(WCP, np.float64))


# ---------------------------------------------------------------- _fast_knn
def knn(x, y, z, box, origin, boundary, k, indices, distances, num_t=1):
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    _chk(lib().orc_knn(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(x.shape[0]),
                       _p(b, np.float64), _p(o, np.float64), _p(p, np.int32), cint(k), _p(indices, np.int32),
                       _p(distances, np.float64), cint(num_t)))


# ------------------------------------------------------------- _repeat_cell
def repeat_cell(new_pos, old_box, old_pos, nx, ny, nz, num_t=1):
    ob = _ro(old_box, np.float64).reshape(9)
    op = _ro(old_pos, np.float64)
    lib().orc_repeat_cell(_p(new_pos, np.float64), _p(ob, np.float64), _p(op, np.float64), i64(op.shape[0]),
                          cint(nx), cint(ny), cint(nz))


def num_procs() -> int:
    return int(lib().orc_num_procs())


# --------------------------------------------------------------------- _ptm  (oracle/_ref: the reference's own PTM library)
_REF_SO = os.path.join(_HERE, "_ref", "libptm_ref.so")
_ref = None


def build_ref(force: bool = False) -> str:
    """Build oracle/_ref/libptm_ref.so from /root/reference/extern/ptm (only possible where the reference is present;
    the GPU box uses the prebuilt file that travels with the snapshot)."""
    if (force or not os.path.exists(_REF_SO)) and os.path.isdir("/root/reference/extern/ptm"):
        subprocess.check_call(["make", "-C", _HERE, "-f", "Makefile.ref", "-s", "-j", "8"])
    return _REF_SO


def have_ref() -> bool:
    return os.path.exists(build_ref())


def ref_lib():
    global _ref
    if _ref is None:
        _ref = C.CDLL(build_ref())
    return _ref


def get_ptm(structure, x, y, z, box, origin, boundary, verlet_list, atom_types, rmsd_threshold, output, ptm_indices,
            num_t=1, cached=None):
    """mdapy._ptm.get_ptm (src/polyhedral_template_matching.cpp:135) through the reference PTM library."""
    _rows("get_ptm", len(x), verlet_list=verlet_list)
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    v = _ro(verlet_list, np.int32)
    t = None if atom_types is None or len(atom_types) != len(x) else _ro(atom_types, np.int32)
    rc = ref_lib().ref_get_ptm(structure.encode(), _p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(len(x)),
                               _p(b, np.float64), _p(o, np.float64), _p(p, np.int32), _p(v, np.int32), i64(v.shape[1]),
                               _p(t, np.int32), dbl(rmsd_threshold), _p(output, np.float64), cint(output.shape[1]),
                               _p(ptm_indices, np.int32), cint(ptm_indices.shape[1]),
                               None if cached is None else cached.ctypes.data_as(C.c_void_p))
    _chk(rc)


# --------------------------------------------------------------------- list consumers
def compute_aja(x, y, z, box, origin, boundary, verlet_list, distance_list, aja, num_t=1):
    """mdapy._aja.compute_aja (src/ackland_jones_analysis.cpp:9)"""
    _rows("compute_aja", len(x), verlet_list=verlet_list)
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    v, d = _ro(verlet_list, np.int32), _ro(distance_list, np.float64)
    _chk(lib().orc_aja(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(x.shape[0]), _p(b, np.float64),
                       _p(o, np.float64), _p(p, np.int32), _p(v, np.int32), _p(d, np.float64), i64(v.shape[1]),
                       _p(aja, np.int32), cint(num_t)))


def compute_cnp(x, y, z, box, origin, boundary, verlet_list, distance_list, neighbor_number, cnp, rc, num_t=1):
    """mdapy._cnp.compute_cnp (src/common_neighbor_parameter.cpp:10)"""
    _rows("compute_cnp", len(x), verlet_list=verlet_list, neighbor_number=neighbor_number)
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    v, d, n = _ro(verlet_list, np.int32), _ro(distance_list, np.float64), _ro(neighbor_number, np.int32)
    _chk(lib().orc_cnp(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(x.shape[0]), _p(b, np.float64),
                       _p(o, np.float64), _p(p, np.int32), _p(v, np.int32), _p(d, np.float64), _p(n, np.int32),
                       i64(v.shape[1]), _p(cnp, np.float64), dbl(rc), cint(num_t)))


def calculate_structure_entropy(rc, sigma, use_local_density, volume, distance_list, neighbor_number, entropy, num_t=1):
    """mdapy._structure_entropy.calculate_structure_entropy (src/structure_entropy.cpp:9)"""
    d, n = _ro(distance_list, np.float64), _ro(neighbor_number, np.int32)
    _chk(lib().orc_structure_entropy(dbl(rc), dbl(sigma), cint(bool(use_local_density)), dbl(volume), _p(d, np.float64),
                                     _p(n, np.int32), i64(d.shape[0]), i64(d.shape[1]), _p(entropy, np.float64),
                                     cint(num_t)))


def compute_temp(verlet_list, distance_list, vx, vy, vz, mass, T, rc, num_t=1):
    """mdapy._atomtemp.compute_temp (src/atomic_temperature.cpp:9)"""
    v, d = _ro(verlet_list, np.int32), _ro(distance_list, np.float64)
    a = [_ro(q, np.float64) for q in (vx, vy, vz, mass)]
    _chk(lib().orc_atomic_temperature(_p(v, np.int32), _p(d, np.float64), i64(v.shape[0]), i64(v.shape[1]),
                                      _p(a[0], np.float64), _p(a[1], np.float64), _p(a[2], np.float64), _p(a[3], np.float64),
                                      _p(T, np.float64), dbl(rc), cint(num_t)))


def get_cluster(verlet_list, distance_list, neighbor_number, rc, particleClusters):
    """mdapy._cluster.get_cluster (src/cluster.cpp:9); particleClusters pre-filled with -1; returns the cluster count"""
    v, d, n = _ro(verlet_list, np.int32), _ro(distance_list, np.float64), _ro(neighbor_number, np.int32)
    return int(lib().orc_cluster(_p(v, np.int32), _p(d, np.float64), _p(n, np.int32), i64(v.shape[0]), i64(v.shape[1]),
                                 dbl(rc), cint(0), _p(particleClusters, np.int32)))


def get_cluster_by_bond(verlet_list, neighbor_number, particleClusters):
    """mdapy._cluster.get_cluster_by_bond (src/cluster.cpp:58)"""
    v, n = _ro(verlet_list, np.int32), _ro(neighbor_number, np.int32)
    return int(lib().orc_cluster(_p(v, np.int32), None, _p(n, np.int32), i64(v.shape[0]), i64(v.shape[1]), dbl(0.0),
                                 cint(1), _p(particleClusters, np.int32)))


def filter_by_type(verlet_list, distance_list, neighbor_number, type_list, type1, type2, r, num_t=1):
    """mdapy._cluster.filter_by_type (src/cluster.cpp:108); verlet_list modified in place"""
    _rows("filter_by_type", np.shape(verlet_list)[0], distance_list=distance_list, neighbor_number=neighbor_number, type_list=type_list)
    d, n, t = _ro(distance_list, np.float64), _ro(neighbor_number, np.int32), _ro(type_list, np.int32)
    t1, t2, rr = _ro(type1, np.int32), _ro(type2, np.int32), _ro(r, np.float64)
    _chk(lib().orc_filter_by_type(_p(verlet_list, np.int32), _p(d, np.float64), _p(n, np.int32), _p(t, np.int32),
                                  i64(verlet_list.shape[0]), i64(verlet_list.shape[1]), _p(t1, np.int32), _p(t2, np.int32),
                                  _p(rr, np.float64), cint(len(t1))))


def identify_sftb_fcc(hcp_indices, hcp_neighbors, ptm_indices, structure_types, fault_types, identify_esf, num_t=1):
    """mdapy._fccpft.identify_sftb_fcc (src/identify_fcc_planar_faults.cpp:54)"""
    h, p, s = _ro(hcp_indices, np.int32), _ro(ptm_indices, np.int32), _ro(structure_types, np.int32)
    _chk(lib().orc_identify_sftb_fcc(_p(h, np.int32), i64(h.shape[0]), _p(hcp_neighbors, np.int32), _p(p, np.int32),
                                     _p(s, np.int32), i64(s.shape[0]), _p(fault_types, np.int32), cint(bool(identify_esf))))


def filter_overlap_atom(x, y, z, box, origin, boundary, rc, num_t=1):
    """mdapy._neighbor.filter_overlap_atom (src/neighbor.cpp:390) -> bool (N)"""
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    keep = np.zeros(len(x), np.uint8)
    _chk(lib().orc_filter_overlap_atom(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(len(x)), _p(b, np.float64),
                                       _p(o, np.float64), _p(p, np.int32), dbl(rc), keep.ctypes.data_as(C.c_void_p), cint(num_t)))
    return keep.astype(bool)


def filter_overlap_atom_with_grain(x, y, z, type_list, grain_id, box, origin, boundary, rc_metal_metal, rc_cc, rc_metal_c, num_t=1):
    """mdapy._neighbor.filter_overlap_atom_with_grain (src/neighbor.cpp:489), serial order -> bool (N)"""
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    t, g = _ro(type_list, np.int32), _ro(grain_id, np.int32)
    b, o, p = _boxargs(box, origin, boundary)
    keep = np.zeros(len(x), np.uint8)
    _chk(lib().orc_filter_overlap_atom_with_grain(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), _p(t, np.int32), _p(g, np.int32),
                                                  i64(len(x)), _p(b, np.float64), _p(o, np.float64), _p(p, np.int32), dbl(rc_metal_metal),
                                                  dbl(rc_cc), dbl(rc_metal_c), keep.ctypes.data_as(C.c_void_p)))
    return keep.astype(bool)


def transform_and_filter(x, y, z, rotation_matrix, center, target_center, coeffs, num_t=1):
    """mdapy._polycrystal.transform_and_filter (src/polycrystal.cpp:20) -> (count, 3)"""
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    r = np.ascontiguousarray(rotation_matrix, dtype=np.float64).reshape(9)
    c = np.ascontiguousarray(center, dtype=np.float64).reshape(3)
    t = np.ascontiguousarray(target_center, dtype=np.float64).reshape(3)
    pl = np.ascontiguousarray(coeffs, dtype=np.float64).reshape(-1, 4)
    out = np.empty((len(x), 3))
    f = lib().orc_transform_and_filter
    f.restype = C.c_int64
    cnt = f(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(len(x)), _p(r, np.float64), _p(c, np.float64),
            _p(t, np.float64), _p(pl, np.float64), cint(len(pl)), _p(out, np.float64))
    return out[: int(cnt)].copy()


# --------------------------------------------------------------------- _voronoi  (oracle/_ref: the reference's own voro++)
_VORO_SO = os.path.join(_HERE, "_ref", "libvoro_ref.so")
_voro = None


def have_voro_ref() -> bool:
    build_ref()
    return os.path.exists(_VORO_SO)


def voro_lib():
    global _voro
    if _voro is None:
        build_ref()
        _voro = C.CDLL(_VORO_SO)
    return _voro


def get_voronoi_volume_number_radius(x, y, z, box, origin, boundary, volume, neighbor_number, cavity_radius, num_t=1):
    """mdapy._voronoi.get_voronoi_volume_number_radius (src/voronoi.cpp:16) through the reference's voro++"""
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    _chk(voro_lib().ref_voronoi_volume_number_radius(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(len(x)),
                                                     _p(b, np.float64), _p(o, np.float64), _p(p, np.int32), _p(volume, np.float64),
                                                     _p(neighbor_number, np.int32), _p(cavity_radius, np.float64)))


def get_voronoi_volume_number_radius_tri(x, y, z, box, origin, boundary, rotation, volume, neighbor_number, cavity_radius,
                                         need_rotation, num_t=1):
    """mdapy._voronoi.get_voronoi_volume_number_radius_tri (src/voronoi.cpp:73)"""
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    r = np.ascontiguousarray(rotation, dtype=np.float64).reshape(9)
    _chk(voro_lib().ref_voronoi_volume_number_radius_tri(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(len(x)),
                                                         _p(b, np.float64), _p(o, np.float64), _p(r, np.float64),
                                                         cint(bool(need_rotation)), _p(volume, np.float64),
                                                         _p(neighbor_number, np.int32), _p(cavity_radius, np.float64)))


def voronoi_faces(x, y, z, box, origin, boundary, width=64):
    """per-cell neighbour ids (negative: wall) and face areas from voro++ (orthogonal container), for set comparisons"""
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    n = len(x)
    nbr, area, cnt = np.full((n, width), -1, np.int32), np.zeros((n, width)), np.zeros(n, np.int32)
    voro_lib().ref_voronoi_faces(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(n), _p(b, np.float64), _p(o, np.float64),
                                 _p(p, np.int32), _p(nbr, np.int32), _p(area, np.float64), cint(width), _p(cnt, np.int32))
    return nbr, area, cnt


def get_voronoi_neighbor(x, y, z, box, origin, boundary, a_face_area_threshold, r_face_area_threshold, num_t=1):
    """mdapy._voronoi.get_voronoi_neighbor (src/voronoi.cpp:307-447): voro++ faces, then the thresholds and the -1 holes
    of :389-430, restated with numpy"""
    nbr, area, cnt = voronoi_faces(x, y, z, box, origin, boundary, width=96)
    width = max(int(cnt.max()), 1)
    nbr, area = nbr[:, :width], area[:, :width]
    n = len(cnt)
    amin = np.full(n, a_face_area_threshold if a_face_area_threshold > 0 else 0.0)
    if r_face_area_threshold > 0:
        amin = area.sum(axis=1) * r_face_area_threshold
    amin = np.maximum(amin, a_face_area_threshold)
    valid = (np.arange(width)[None, :] < cnt[:, None]) & (nbr >= 0) & (area > amin[:, None])
    verlet = np.where(valid, nbr, -1).astype(np.int32)
    face = np.where(valid, area, 0.0)
    b = np.asarray(box, float)
    b = b if b.ndim == 2 else np.diag(b)
    pos = np.stack([np.asarray(x, float), np.asarray(y, float), np.asarray(z, float)], axis=1)
    d = pos[np.clip(verlet, 0, None)] - pos[:, None, :]
    for a in range(3):
        if boundary[a]:
            d[..., a] -= b[a, a] * np.floor(d[..., a] / b[a, a] + 0.5)
    dist = np.where(valid, np.sqrt((d ** 2).sum(-1)), 10000.0)
    return verlet, dist, face, cnt.astype(np.int32)


def get_cell_info(x, y, z, box, origin, boundary, num_t=1):
    """mdapy._voronoi.get_cell_info (src/voronoi.cpp:449-540) through the reference's own voro++"""
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    n = len(x)
    nf, nv = np.zeros(n, np.int32), np.zeros(n, np.int32)
    vol, rad = np.zeros(n), np.zeros(n)
    nfv, nvert = C.c_int64(0), C.c_int64(0)
    f = voro_lib().ref_voronoi_cell_info
    args = (_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(n), _p(b, np.float64), _p(o, np.float64), _p(p, np.int32),
            _p(nf, np.int32), _p(nv, np.int32), C.byref(nfv), C.byref(nvert))
    nface = f(*args, None, None, None, None, _p(vol, np.float64), _p(rad, np.float64))
    fs, fid = np.zeros(max(nface, 1), np.int32), np.zeros(max(nfv.value, 1), np.int32)
    vert, fa = np.zeros((max(nvert.value, 1), 3)), np.zeros(max(nface, 1))
    f(*args, _p(fs, np.int32), _p(fid, np.int32), _p(vert, np.float64), _p(fa, np.float64), _p(vol, np.float64), _p(rad, np.float64))
    face_idx, face_pos, areas = [], [], []
    kf = kid = kv = 0
    for i in range(n):
        faces, ar = [], []
        for _ in range(int(nf[i])):
            m = int(fs[kf])
            faces.append(fid[kid:kid + m].tolist())
            ar.append(float(fa[kf]))
            kid += m
            kf += 1
        face_idx.append(faces)
        areas.append(ar)
        face_pos.append(vert[kv:kv + int(nv[i])].tolist())
        kv += int(nv[i])
    return face_idx, face_pos, vol.tolist(), rad.tolist(), areas


def get_voronoi_neighbor_tri(x, y, z, box, origin, boundary, rotation, need_rotation, a_face_area_threshold,
                             r_face_area_threshold, num_t=1):
    """mdapy._voronoi.get_voronoi_neighbor_tri (src/voronoi.cpp:149-305): faces from the triclinic container of the ROTATED
    positions; thresholds as in the orthogonal variant; the reported distance is box.pbc of the UNROTATED difference
    x[j] - x[i] in the aligned box (:277-282), whatever that means when a rotation was needed"""
    xr, yr, zr = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b, o, p = _boxargs(box, origin, boundary)
    r = np.ascontiguousarray(rotation, dtype=np.float64).reshape(9)
    n, width = len(xr), 96
    nbr, area, cnt = np.full((n, width), -1, np.int32), np.zeros((n, width)), np.zeros(n, np.int32)
    voro_lib().ref_voronoi_faces_tri(_p(xr, np.float64), _p(yr, np.float64), _p(zr, np.float64), i64(n), _p(b, np.float64),
                                     _p(o, np.float64), _p(r, np.float64), cint(bool(need_rotation)), _p(nbr, np.int32),
                                     _p(area, np.float64), cint(width), _p(cnt, np.int32))
    width = max(int(cnt.max()), 1)
    nbr, area = nbr[:, :width], area[:, :width]
    amin = np.full(n, a_face_area_threshold if a_face_area_threshold > 0 else 0.0)
    if r_face_area_threshold > 0:
        amin = area.sum(axis=1) * r_face_area_threshold
    amin = np.maximum(amin, a_face_area_threshold)
    valid = (np.arange(width)[None, :] < cnt[:, None]) & (nbr >= 0) & (area > amin[:, None])
    verlet = np.where(valid, nbr, -1).astype(np.int32)
    face = np.where(valid, area, 0.0)
    h = np.asarray(box, float).reshape(3, 3)
    pos = np.stack([xr, yr, zr], axis=1)
    d = pos[np.clip(verlet, 0, None)] - pos[:, None, :]
    tri = bool(np.any(np.abs(h - np.diag(np.diag(h))) > 1e-10) or np.any(np.diag(h) < 0))  # box.h:182-244
    if tri:
        f = d @ np.linalg.inv(h)  # box.h:86-118
        for a in range(3):
            if boundary[a]:
                f[..., a] -= np.floor(f[..., a] + 0.5)
        d = f @ h
    else:
        for a in range(3):
            if boundary[a]:
                d[..., a] -= h[a, a] * np.floor(d[..., a] / h[a, a] + 0.5)
    dist = np.where(valid, np.sqrt((d ** 2).sum(-1)), 10000.0)
    return verlet, dist, face, cnt.astype(np.int32)


# --------------------------------------------------------------------- _sfc (static structure factor, direct summation)
def compute_sfc_direct(x, y, z, box, origin, boundary, structure_factor_py, bins, k_max, k_min, query_x=None, query_y=None,
                       query_z=None, N_total=0, num_t=1):
    """mdapy._sfc.compute_sfc_direct (src/structure_factor.cpp:654)"""
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b = _ro(box, np.float64).reshape(9)
    q = [None, None, None] if query_x is None else [_ro(query_x, np.float64), _ro(query_y, np.float64), _ro(query_z, np.float64)]
    if query_x is not None and N_total == 0:
        raise ValueError("N_total is required when query points are provided.")
    rc = lib().orc_sfc_direct(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), i64(len(x)), _p(b, np.float64),
                              _p(structure_factor_py, np.float64), cint(bins), dbl(k_max), dbl(k_min), _p(q[0], np.float64),
                              _p(q[1], np.float64), _p(q[2], np.float64), i64(0 if q[0] is None else len(q[0])),
                              C.c_uint(int(N_total)), cint(num_t))
    if rc == -2:
        raise RuntimeError("No k-points generated. Check k_min and k_max values.")
    _chk(rc)


def compute_sfc_direct_partial(x, y, z, type_list, Ntype, box, origin, boundary, partial_out, bins, k_max, k_min, num_t=1):
    """mdapy._sfc.compute_sfc_direct_partial (src/structure_factor.cpp:682)"""
    x, y, z = _ro(x, np.float64), _ro(y, np.float64), _ro(z, np.float64)
    b = _ro(box, np.float64).reshape(9)
    t = _ro(type_list, np.int32)
    rc = lib().orc_sfc_direct_partial(_p(x, np.float64), _p(y, np.float64), _p(z, np.float64), _p(t, np.int32), cint(Ntype),
                                      i64(len(x)), _p(b, np.float64), _p(partial_out, np.float64), cint(bins), dbl(k_max),
                                      dbl(k_min), cint(num_t))
    if rc == -2:
        raise RuntimeError("No k-points generated")
    _chk(rc)


def sfc_kpoints(box, k_max, k_min, partial=False):
    b = _ro(box, np.float64).reshape(9)
    f = lib().orc_sfc_kpoints_partial if partial else lib().orc_sfc_kpoints_total
    f.restype = C.c_int64
    n = int(f(_p(b, np.float64), dbl(k_max), dbl(k_min), None))
    kp = np.zeros((max(n, 1), 3))
    f(_p(b, np.float64), dbl(k_max), dbl(k_min), _p(kp, np.float64))
    return kp[:n]
