"""Small rigid-geometry helpers of the polycrystal builder: rotations, face planes of a convex cell, points inside a
planar polygon.  Plain numpy on a handful of faces per grain; the per-atom work of the builder is on the GPU."""
import numpy as np


def rotation_about(axis, degrees):
    """3x3 matrix of the right-handed rotation by ``degrees`` about ``axis`` (Rodrigues):
    R_ij = cos t delta_ij + (1 - cos t) n_i n_j - sin t eps_ijk n_k"""
    n = np.array(axis, dtype=float)
    length = np.linalg.norm(n)
    if length == 0:
        raise ValueError("Rotation axis must be non-zero")
    n = n / length
    t = np.radians(degrees)
    cos_t, sin_t = np.cos(t), np.sin(t)
    rest = 1 - cos_t
    out = np.empty((3, 3), dtype=float)
    for i in range(3):
        for j in range(3):
            if i == j:
                out[i, j] = cos_t + rest * n[i] * n[j]
            else:
                k = 3 - i - j
                sign = 1.0 if (j - i) % 3 == 2 else -1.0  # -eps_ijk
                out[i, j] = rest * n[i] * n[j] + sign * sin_t * n[k]
    return out


def euler_xyz(angles):
    """rotation about x, then the result's product with rotations about y and z: R_x(a) R_y(b) R_z(c)"""
    a, b, c = angles
    return rotation_about((1.0, 0.0, 0.0), a) @ rotation_about((0.0, 1.0, 0.0), b) @ rotation_about((0.0, 0.0, 1.0), c)


def rotation_taking(source, target):
    """rotation that turns the direction ``source`` into the direction ``target``"""
    u, v = source / np.linalg.norm(source), target / np.linalg.norm(target)
    cosine = float(np.dot(u, v))
    if np.isclose(cosine, 1.0, atol=1e-6):
        return np.eye(3)
    if np.isclose(cosine, -1.0, atol=1e-6):  # opposite: half a turn about any axis perpendicular to u
        helper = np.array([1.0, 0.0, 0.0]) if abs(u[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
        axis = np.cross(u, helper)
        return rotation_about(tuple(axis / np.linalg.norm(axis)), 180.0)
    axis = np.cross(u, v)
    angle = float(np.degrees(np.arccos(np.clip(cosine, -1.0, 1.0))))
    return rotation_about(tuple(axis / np.linalg.norm(axis)), angle)


def inward_planes(cell):
    """(faces, 4) rows (a, b, c, d): a x + b y + c z + d < 0 for points on the side of the cell's own atom.  The normal of
    a face comes from the two edges at its first vertex."""
    planes = np.zeros((len(cell.face_vertices), 4))
    for f, corners in enumerate(cell.face_vertices):
        first, second, third = (cell.vertices[corners[i]] for i in range(3))
        normal = np.cross(second - first, third - first)
        size = np.linalg.norm(normal)
        if size < 1e-10:
            raise ValueError(f"Degenerate face vertices at face {f}")
        normal = normal / size
        offset = -np.dot(normal, first)
        if np.dot(normal, cell.pos) + offset > 0:
            normal, offset = -normal, -offset
        planes[f, :3], planes[f, 3] = normal, offset
    return planes


def inside_polygon(polygon, points):
    """which 2-D points lie inside a polygon: even-odd ray casting along +x in single precision; a point on a corner counts
    as inside"""
    poly = np.asarray(polygon, dtype=np.float32)
    pts = np.asarray(points, dtype=np.float32)[:, None, :]
    start = poly[None, :, :]
    end = np.roll(poly, -1, axis=0)[None, :, :]
    at_corner = np.isclose(pts, start, atol=1e-6).all(axis=2).any(axis=1)
    straddles = (start[..., 1] > pts[..., 1]) != (end[..., 1] > pts[..., 1])
    crossing_x = (end[..., 0] - start[..., 0]) * (pts[..., 1] - start[..., 1]) / (end[..., 1] - start[..., 1] + 1e-10) + start[..., 0]
    crossings = np.count_nonzero(straddles & (pts[..., 0] < crossing_x), axis=1)
    return (crossings % 2 == 1) | at_corner


def on_face(points, corners, normal, slab=0.5):
    """the points within ``slab`` of the plane of a convex face whose projection falls inside the face"""
    up = normal / np.linalg.norm(normal)
    middle = corners.mean(axis=0)
    for corner in corners[:2]:  # an in-plane axis from the centre towards a corner (the second if the first is degenerate)
        along = corner - middle
        along = along - np.dot(along, up) * up
        if np.linalg.norm(along) >= 1e-8:
            break
    along = along / np.linalg.norm(along)
    axes = np.array([along, np.cross(up, along), up])
    flat_corners = (corners - middle) @ axes.T
    flat_points = (points - middle) @ axes.T
    chosen = (np.abs(flat_points[:, 2]) < slab) & inside_polygon(flat_corners[:, :2], flat_points[:, :2])
    return points[chosen]
