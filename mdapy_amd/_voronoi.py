"""Drop-in for the volume functions of ``mdapy._voronoi`` (src/voronoi.cpp:544-545)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def get_voronoi_volume_number_radius(x, y, z, box, origin, boundary, volume, neighbor_number, cavity_radius, num_t=1):
    """src/voronoi.cpp:16 — per atom: Voronoi cell volume, number of faces (walls of open axes included), largest
    vertex distance"""
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, volume, neighbor_number, cavity_radius)
    rc_ = _lib.lib().mdh_voronoi_volume_number_radius(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), int(len(x)), pb, po, pp,
                                                      c.out(volume, f64, upload=False), c.out(neighbor_number, i32, upload=False),
                                                      c.out(cavity_radius, f64, upload=False), c.space, c.stream)
    c.done(rc_)


def get_voronoi_volume_number_radius_tri(x, y, z, box, origin, boundary, rotation, volume, neighbor_number, cavity_radius,
                                         need_rotation, num_t=1):
    """src/voronoi.cpp:73 — the box is LAMMPS-aligned and treated as fully periodic (container_triclinic); the rotation
    is a rigid motion, so cell volumes, face counts and vertex distances are those of the rotated positions"""
    x = np.asarray(x, f64) - origin[0]
    y = np.asarray(y, f64) - origin[1]
    z = np.asarray(z, f64) - origin[2]
    if need_rotation:
        r = np.asarray(rotation, f64)
        x, y, z = (x * r[0, k] + y * r[1, k] + z * r[2, k] for k in range(3))
    get_voronoi_volume_number_radius(np.ascontiguousarray(x), np.ascontiguousarray(y), np.ascontiguousarray(z), box, np.zeros(3),
                                     np.ones(3, i32), volume, neighbor_number, cavity_radius, num_t)


_ROW_GUESS = 32  # columns of the device-side rows of get_voronoi_neighbor's first attempt (a test lowers it to reach the second)


def get_voronoi_neighbor(x, y, z, box, origin, boundary, a_face_area_threshold, r_face_area_threshold, num_t=1):
    """src/voronoi.cpp:307 -> (verlet (N,W) i32, distance (N,W) f64, face_area (N,W) f64, neighbor_number (N) i32).
    Rows list the faces shared with atoms, nearest first (the reference's rows follow voro++'s internal face order and
    keep -1 holes where a face was filtered; every consumer skips -1 entries)."""
    import ctypes

    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    n = int(len(x))
    nn = np.zeros(n, i32)
    w = ctypes.c_int(0)
    c = Call(x, y, z)
    if c.space != _lib.HOST:
        raise TypeError("get_voronoi_neighbor takes host (numpy) positions")
    L = _lib.lib()
    # one construction of the cells (the reference builds its container once): rows 32 columns wide on the device — cells of
    # crystals and liquids have 12 to ~26 faces — handed over at the width the face counts ask for; a cell with more faces: again
    guess = _ROW_GUESS
    while True:
        verlet, dist, area = np.empty(n * guess, i32), np.empty(n * guess, f64), np.empty(n * guess, f64)
        _lib.check(L.mdh_voronoi_neighbor_rows(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), n, pb, po, pp, float(a_face_area_threshold),
                                               float(r_face_area_threshold), verlet.ctypes.data, dist.ctypes.data, area.ctypes.data,
                                               guess, nn.ctypes.data, ctypes.byref(w), c.space, c.stream))
        width = max(int(w.value), 1)
        if width <= guess:
            break
        guess = width
    cut = lambda a: a[:n * width].reshape(n, width)
    return cut(verlet), cut(dist), cut(area), nn


def get_voronoi_neighbor_tri(x, y, z, box, origin, boundary, rotation, need_rotation, a_face_area_threshold,
                             r_face_area_threshold, num_t=1):
    """src/voronoi.cpp:149 — LAMMPS-aligned box, fully periodic container; ids and areas are invariant under the rotation.
    The distance column is the reference's: box.pbc of the UNROTATED x[j] - x[i] with this call's box and boundary flags
    (:277-282)."""
    x0, y0, z0 = (np.ascontiguousarray(np.asarray(v, f64)) for v in (x, y, z))
    xr, yr, zr = x0 - origin[0], y0 - origin[1], z0 - origin[2]
    if need_rotation:
        r = np.asarray(rotation, f64)
        xr, yr, zr = (xr * r[0, k] + yr * r[1, k] + zr * r[2, k] for k in range(3))
    verlet, dist, area, nn = get_voronoi_neighbor(np.ascontiguousarray(xr), np.ascontiguousarray(yr), np.ascontiguousarray(zr), box,
                                                  np.zeros(3), np.ones(3, i32), a_face_area_threshold, r_face_area_threshold, num_t)
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x0, y0, z0)
    _lib.check(_lib.lib().mdh_voronoi_row_distance(verlet.ctypes.data, int(len(x0)), int(verlet.shape[1]), c.inp(x0, f64),
                                                   c.inp(y0, f64), c.inp(z0, f64), pb, po, pp, dist.ctypes.data, c.space, c.stream))
    return verlet, dist, area, nn


def get_cell_info(x, y, z, box, origin, boundary, num_t=1):
    """src/voronoi.cpp:449 -> (face_vertices_indices, face_vertices_positions, volume, radius, face_areas), lists per atom like
    the reference's: faces as lists of indices into the cell's vertex list, the vertex list (container frame: position minus
    origin, wrapped on periodic axes, plus the vertex offsets), cell volume, cavity radius, face areas.

    The polygons come from the device (``mdh_voronoi_cell_info``: each face clipped on its own); here the copies of a vertex
    that three or more faces share are merged into one entry.  Faces are listed walls first, then nearest neighbour first,
    vertices in order of first appearance — voro++'s own orders are internal to that library.  Atoms the reference's container
    does not hold (outside on an open axis) get empty lists and zeros."""
    import ctypes

    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    n = int(len(x))
    xs, ys, zs = (np.ascontiguousarray(np.asarray(a, f64)) for a in (x, y, z))
    L = _lib.lib()
    nn = np.zeros(n, i32)
    w = ctypes.c_int(0)
    _lib.check(L.mdh_voronoi_neighbor_count(xs.ctypes.data, ys.ctypes.data, zs.ctypes.data, n, pb, po, pp, nn.ctypes.data,
                                            ctypes.byref(w), _lib.HOST, None))
    W, V = max(int(w.value), 1), 12
    while True:
        nf = np.zeros(n, i32)
        fnv = np.zeros((n, W), i32)
        farea = np.zeros((n, W))
        fvert = np.zeros((n, W, V, 3))
        vol, rad = np.zeros(n), np.zeros(n)
        need = ctypes.c_int(0)
        _lib.check(L.mdh_voronoi_cell_info(xs.ctypes.data, ys.ctypes.data, zs.ctypes.data, n, pb, po, pp, W, V, nf.ctypes.data,
                                           fnv.ctypes.data, farea.ctypes.data, fvert.ctypes.data, vol.ctypes.data, rad.ctypes.data,
                                           ctypes.byref(need), _lib.HOST, None))
        if need.value <= V:
            break
        V = int(need.value)
    h = np.asarray(box, f64).reshape(3, 3)
    o = np.asarray(origin, f64).reshape(3)
    bd = np.asarray(boundary).reshape(3)
    p = np.stack([xs, ys, zs], axis=1) - o
    for a in range(3):  # the container holds periodic coordinates folded into [0, L)
        if bd[a]:
            p[:, a] -= h[a, a] * np.floor(p[:, a] / h[a, a])
    def assemble(i, tol):
        verts, faces, ar = [], [], []
        for s_ in range(W):
            m = int(fnv[i, s_])
            if m == 0:
                continue
            ids = []
            for c in range(m):
                v = fvert[i, s_, c]
                hit = -1
                for q, u in enumerate(verts):
                    if abs(u[0] - v[0]) <= tol and abs(u[1] - v[1]) <= tol and abs(u[2] - v[2]) <= tol:
                        hit = q
                        break
                if hit < 0:
                    verts.append(v)
                    hit = len(verts) - 1
                if not ids or (ids[-1] != hit and ids[0] != hit):  # a sliver edge shorter than the tolerance collapses
                    ids.append(hit)
            faces.append(ids)
            ar.append(float(farea[i, s_]))
        return verts, faces, ar

    face_idx, face_pos, areas = [], [], []
    for i in range(n):
        # The copies of a shared vertex agree to rounding, usually ~1e-13 of the cell size, worse where planes meet at a
        # shallow angle.  The merge tolerance is the smallest one for which the faces close up into a polyhedron
        # (Euler: V - E + F = 2 with every edge shared by two faces).
        verts, faces, ar = assemble(i, 1e-9 * max(float(rad[i]), 1e-300))
        for scale in (1e-8, 1e-7, 1e-6, 1e-5):
            n_half_edges = sum(len(f) for f in faces)
            if not faces or (n_half_edges % 2 == 0 and len(verts) - n_half_edges // 2 + len(faces) == 2):
                break
            verts, faces, ar = assemble(i, scale * max(float(rad[i]), 1e-300))
        face_idx.append(faces)
        face_pos.append([(p[i] + np.asarray(v)).tolist() for v in verts])
        areas.append(ar)
    return face_idx, face_pos, vol.tolist(), rad.tolist(), areas
