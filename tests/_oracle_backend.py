"""Adapters that give the CPU oracle the call surface of mdapy_amd's backend shim modules.
TEST INFRASTRUCTURE: used by the CPU ("not gpu") tests to run the host-side policy layer of
mdapy_amd against the golden vectors, and by the GPU tests as the checker."""
import types

import numpy as np

from oracle import oracle as O
from mdapy_amd.devarray import as_numpy

NT = 4


def _np(a):
    return None if a is None else as_numpy(a)


def _mod(**fns):
    return types.SimpleNamespace(**fns)


def _keyed(key, x, y, z):
    """"descending key inside a cell" == the reference's "descending index" after sorting the atoms by key"""
    perm = np.argsort(_np(key), kind="stable")
    return perm, tuple(np.ascontiguousarray(_np(c)[perm]) for c in (x, y, z))


def _unkey(perm, vs, ds, ns, v, d, nn):
    v[perm] = np.where(vs >= 0, perm[np.clip(vs, 0, None)], -1)
    d[perm] = ds
    nn[perm] = ns


def _build_neighbor(x, y, z, box, origin, boundary, rc, v, d, nn, num_t=1, fill_pads=False, key=None):
    if fill_pads:
        v.fill(-1)
        d.fill(rc + 1.0)
    if key is None:
        O.build_neighbor(_np(x), _np(y), _np(z), box, origin, boundary, rc, v, d, nn, NT)
        return
    perm, (xs, ys, zs) = _keyed(key, x, y, z)
    vs, ds, ns = v[perm].copy(), d[perm].copy(), nn[perm].copy()  # (the caller's pads travel with their rows)
    O.build_neighbor(xs, ys, zs, box, origin, boundary, rc, vs, ds, ns, NT)
    _unkey(perm, vs, ds, ns, v, d, nn)


def _build_neighbor_exact(x, y, z, box, origin, boundary, rc, num_t=1, key=None, pattern=None):
    if key is None:
        v, d, nn = O.build_neighbor_without_max_neigh(_np(x), _np(y), _np(z), box, origin, boundary, rc, NT)
    else:
        perm, (xs, ys, zs) = _keyed(key, x, y, z)
        vs, ds, ns = O.build_neighbor_without_max_neigh(xs, ys, zs, box, origin, boundary, rc, NT)
        v, d, nn = np.empty_like(vs), np.empty_like(ds), np.empty_like(ns)
        _unkey(perm, vs, ds, ns, v, d, nn)
    if pattern is not None:  # the fused call of the library = the two reference calls one after the other
        O.fcna(_np(x), _np(y), _np(z), box, origin, boundary, v, nn, pattern, rc, NT)
    return v, d, nn


def _build_neighbor_fcna(x, y, z, box, origin, boundary, rc, v, d, nn, pattern, num_t=1, fill_pads=False, key=None):
    _build_neighbor(x, y, z, box, origin, boundary, rc, v, d, nn, num_t, fill_pads, key)
    O.fcna(_np(x), _np(y), _np(z), box, origin, boundary, v, nn, pattern, rc, NT)


def _spatial_sort(x, y, z, box, origin, boundary):
    """a stand-in for mdh_spatial_sort: ANY permutation serves the host logic that is tested with it — here cells of 3 A"""
    pos = np.column_stack([_np(x), _np(y), _np(z)])
    cells = np.floor((pos - pos.min(axis=0)) / 3.0).astype(np.int64)
    perm = np.lexsort((-np.arange(len(pos)), cells[:, 2], cells[:, 1], cells[:, 0])).astype(np.int32)
    return pos[perm, 0].copy(), pos[perm, 1].copy(), pos[perm, 2].copy(), perm, len(pos)


def _permute(values, perm, scatter=False):
    a, p = _np(values), _np(perm)
    if not scatter:
        return a[p]
    out = np.empty_like(a)
    out[p] = a
    return out


def _translate_rows(rows, dist, counts, perm):
    rows, p = _np(rows), _np(perm)
    v = np.empty_like(rows)
    v[p] = np.where(rows >= 0, p[np.clip(rows, 0, None)], rows)
    d = n = None
    if dist is not None:
        d = np.empty_like(_np(dist)); d[p] = _np(dist)
    if counts is not None:
        n = np.empty_like(_np(counts)); n[p] = _np(counts)
    return v, d, n


order = _mod(order_statistic=lambda x, y, z, box, origin, boundary: 1.0, spatial_sort=_spatial_sort, permute=_permute,
             translate_rows=_translate_rows)
neighbor = _mod(
    build_neighbor=_build_neighbor,
    build_neighbor_fcna=_build_neighbor_fcna,
    build_neighbor_without_max_neigh=_build_neighbor_exact,
    sort_verlet_by_distance=lambda v, d, k, num_t=1: O.sort_verlet_by_distance(v, d, k, NT),
    wrap_positions=lambda x, y, z, box, origin, boundary, num_t=1: O.wrap_positions(x, y, z, box, origin, boundary, NT),
    average_by_neighbor=lambda rc, v, d, nn, value, out, inc, num_t=1:
        O.average_by_neighbor(rc, _np(v), _np(d), _np(nn), _np(value), out, inc, NT),
    filter_overlap_atom=lambda x, y, z, box, origin, boundary, rc, num_t=1:
        O.filter_overlap_atom(_np(x), _np(y), _np(z), box, origin, boundary, rc, NT),
    filter_overlap_atom_with_grain=lambda x, y, z, t, g, box, origin, boundary, mm, cc, mc, num_t=1:
        O.filter_overlap_atom_with_grain(_np(x), _np(y), _np(z), _np(t), _np(g), box, origin, boundary, mm, cc, mc, NT),
)
polycrystal = _mod(
    transform_and_filter=lambda x, y, z, rot, c, t, pl, num_t=1: O.transform_and_filter(_np(x), _np(y), _np(z), rot, c, t, pl, NT),
)
cna = _mod(
    fcna=lambda x, y, z, box, origin, boundary, v, nn, pat, rc, num_t=1:
        O.fcna(_np(x), _np(y), _np(z), box, origin, boundary, _np(v), _np(nn), pat, rc, NT),
    acna=lambda x, y, z, box, origin, boundary, v, pat, num_t=1:
        O.acna(_np(x), _np(y), _np(z), box, origin, boundary, _np(v), pat, NT),
    ids=lambda x, y, z, box, origin, boundary, v, nv, pat, num_t=1:
        O.ids(_np(x), _np(y), _np(z), box, origin, boundary, _np(v), nv, pat, NT),
)
csp = _mod(get_csp=lambda x, y, z, box, origin, boundary, v, N, out, num_t=1:
           O.get_csp(_np(x), _np(y), _np(z), box, origin, boundary, _np(v), N, out, NT))
sbo = _mod(
    get_sq=lambda x, y, z, box, origin, boundary, v, d, nn, w, ll, nnn, lmax, wl, wlhat, avg, uv, rc, uw, qr, qi, qn,
    num_t=1: O.get_sq(_np(x), _np(y), _np(z), box, origin, boundary, _np(v), _np(d), _np(nn), _np(w), ll, nnn, lmax,
                      wl, wlhat, avg, uv, rc, uw, qr, qi, qn, NT),
    identifySolidLiquid=lambda qi_, Q6, v, d, nn, qr, qi, thr, nb, sl, nbond, uv, nnn, rc, num_t=1:
        O.identifySolidLiquid(qi_, _np(Q6), _np(v), _np(d), _np(nn), _np(qr), _np(qi), thr, nb, sl, nbond, uv, nnn,
                              rc, NT),
)
rdf = _mod(
    _rdf=lambda v, d, nn, t, g, rc, nbin: O._rdf(_np(v), _np(d), _np(nn), _np(t), g, rc, nbin),
    _rdf_single_species=lambda v, d, nn, g, rc, nbin: O._rdf_single_species(_np(v), _np(d), _np(nn), g, rc, nbin),
    _rdf_streaming=lambda x, y, z, t, box, origin, boundary, g, rc, nbin, num_t=1:
        O._rdf_streaming(_np(x), _np(y), _np(z), _np(t), box, origin, boundary, g, rc, nbin, NT),
)
wcp = _mod(get_wcp=lambda v, nn, t, Nt, W, num_t=1: O.get_wcp(_np(v), _np(nn), _np(t), Nt, W, NT))
def _knn(x, y, z, box, origin, boundary, k, idx, dist, num_t=1, key=None):
    if key is None:
        O.knn(_np(x), _np(y), _np(z), box, origin, boundary, k, idx, dist, NT)
        return
    # ties by key: the search in the key's numbering, rows and entries mapped back
    perm, (xs, ys, zs) = _keyed(key, x, y, z)
    ik, dk = np.empty_like(idx), np.empty_like(dist)
    O.knn(xs, ys, zs, box, origin, boundary, k, ik, dk, NT)
    idx[perm] = np.where(ik >= 0, perm[np.clip(ik, 0, None)], ik)
    dist[perm] = dk


fast_knn = _mod(knn=_knn)
ptm = _mod(get_ptm=lambda st, x, y, z, box, origin, boundary, v, t, thr, out, ind, num_t=1:
           O.get_ptm(st, _np(x), _np(y), _np(z), box, origin, boundary, _np(v), _np(t), thr, out, ind, NT))
aja = _mod(compute_aja=lambda x, y, z, box, origin, boundary, v, d, out, num_t=1:
           O.compute_aja(_np(x), _np(y), _np(z), box, origin, boundary, _np(v), _np(d), out, NT))
cnp = _mod(compute_cnp=lambda x, y, z, box, origin, boundary, v, d, nn, out, rc, num_t=1:
           O.compute_cnp(_np(x), _np(y), _np(z), box, origin, boundary, _np(v), _np(d), _np(nn), out, rc, NT))
structure_entropy = _mod(calculate_structure_entropy=lambda rc, sigma, loc, vol, d, nn, out, num_t=1:
                         O.calculate_structure_entropy(rc, sigma, loc, vol, _np(d), _np(nn), out, NT))
atomtemp = _mod(compute_temp=lambda v, d, vx, vy, vz, m, T, rc, num_t=1:
                O.compute_temp(_np(v), _np(d), _np(vx), _np(vy), _np(vz), _np(m), T, rc, NT))


def _fill(a):
    a.fill(-1)
    return a


cluster = _mod(
    get_cluster=lambda v, d, nn, rc, out: O.get_cluster(_np(v), _np(d), _np(nn), rc, _fill(out)),
    get_cluster_by_bond=lambda v, nn, out: O.get_cluster_by_bond(_np(v), _np(nn), _fill(out)),
    filter_by_type=lambda v, d, nn, t, t1, t2, r, num_t=1: O.filter_by_type(v, _np(d), _np(nn), _np(t), t1, t2, r, NT),
)
fccpft = _mod(identify_sftb_fcc=lambda h, hn, p, s, f, esf, num_t=1: O.identify_sftb_fcc(h, hn, _np(p), _np(s), f, esf, NT))
voronoi = _mod(
    get_voronoi_volume_number_radius=lambda x, y, z, box, origin, boundary, v, n, r, num_t=1:
        O.get_voronoi_volume_number_radius(_np(x), _np(y), _np(z), box, origin, boundary, v, n, r, NT),
    get_voronoi_volume_number_radius_tri=lambda x, y, z, box, origin, boundary, rot, v, n, r, need, num_t=1:
        O.get_voronoi_volume_number_radius_tri(_np(x), _np(y), _np(z), box, origin, boundary, rot, v, n, r, need, NT),
    get_voronoi_neighbor=lambda x, y, z, box, origin, boundary, a, r, num_t=1:
        O.get_voronoi_neighbor(_np(x), _np(y), _np(z), box, origin, boundary, a, r, NT),
    get_cell_info=lambda x, y, z, box, origin, boundary, num_t=1: O.get_cell_info(_np(x), _np(y), _np(z), box, origin, boundary, NT),
    get_voronoi_neighbor_tri=lambda x, y, z, box, origin, boundary, rot, need, a, r, num_t=1:
        O.get_voronoi_neighbor_tri(_np(x), _np(y), _np(z), box, origin, boundary, rot, need, a, r, NT),
)
sfc = _mod(
    compute_sfc_direct=lambda x, y, z, box, origin, boundary, sf, bins, k_max, k_min, query_x=None, query_y=None, query_z=None,
    N_total=0, num_t=1: O.compute_sfc_direct(_np(x), _np(y), _np(z), box, origin, boundary, sf, bins, k_max, k_min, _np(query_x),
                                             _np(query_y), _np(query_z), N_total, NT),
    compute_sfc_direct_partial=lambda x, y, z, t, nt, box, origin, boundary, out, bins, k_max, k_min, num_t=1:
        O.compute_sfc_direct_partial(_np(x), _np(y), _np(z), _np(t), nt, box, origin, boundary, out, bins, k_max, k_min, NT),
)
repeat_cell = _mod(repeat_cell=lambda new, ob, op, nx, ny, nz, num_t=1: O.repeat_cell(new, ob, _np(op), nx, ny, nz, NT))


def install(monkeypatch):
    """swap the shim modules behind mdapy_amd.kernels (the package's one door to the C ABI) for the adapters above"""
    import mdapy_amd.kernels as K

    table = dict(neighbor=neighbor, polycrystal=polycrystal, repeat_cell=repeat_cell, fast_knn=fast_knn, cna=cna, csp=csp,
                 sbo=sbo, ptm=ptm, rdf=rdf, wcp=wcp, aja=aja, atomtemp=atomtemp, cluster=cluster, sfc=sfc, voronoi=voronoi,
                 fccpft=fccpft, cnp=cnp, structure_entropy=structure_entropy, order=order)
    assert set(table) == set(K.NAMES), "an adapter per shim module"
    for name, adapter in table.items():
        monkeypatch.setattr(K, name, adapter)
