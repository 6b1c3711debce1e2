// voro_core.hpp — one face of a Voronoi cell by half-space clipping (host + device).
//
// The reference computes cells with voro++ (extern/voro++, used by src/voronoi.cpp:16-147): a vertex/edge graph that is
// cut plane by plane.  On the GPU the cell is assembled face by face instead — one lane per face, no shared mutable
// cell: face f is the part of its own plane that satisfies every other constraint, obtained by clipping a large square
// in that plane (clip_poly, ptm_core.hpp).  Constraints are half-spaces v.n <= o around the atom: bisectors of the
// neighbours (n = r_j, o = |r_j|^2 / 2), walls of open box axes, and a bounding cube that only matters when the
// neighbourhood is incomplete.  With the constraints sorted by their distance from the atom a face stops clipping as
// soon as the next plane lies beyond its farthest vertex.
//   volume        = sum over faces  area_f * h_f / 3          (h_f = distance of the plane from the atom)
//   faces         = faces with a non-degenerate polygon        (voro++: number_of_faces)
//   cavity radius = largest vertex distance                    (voro++: sqrt(max_radius_squared))
#pragma once
#include "ptm_core.hpp"

namespace voroc {

using ptmc::clip_poly;
using ptmc::cross3;
using ptmc::dot3;

struct FaceResult {
    double area;   // 0 when the face does not exist
    double maxr2;  // largest squared vertex distance of this face
    bool overflow; // polygon storage too small
    int nv;        // vertices of the face polygon (left in `poly`, counter-clockwise or clockwise as clipped)
};

// constraint k: v . nrm[k] <= off[k]; dist[k] = off[k] / |nrm[k]| ascending for k >= first_sorted.
// Face f is computed against constraints [0, nc).  `big` bounds the initial square.
template <class P>
PTM_HDN FaceResult voronoi_face(P &poly, int f, int nc, const double (*nrm)[3], const double *off, const double *dist,
                                int first_sorted, double big)
{
    FaceResult r{0.0, 0.0, false, 0};
    const double *p = nrm[f];
    const double pn2 = dot3(p, p);
    if (!(pn2 > 0))
        return r;
    // closest point of the plane to the atom and two in-plane unit vectors
    const double s = off[f] / pn2;
    const double c0[3] = {p[0] * s, p[1] * s, p[2] * s};
    double u[3], v[3], e[3] = {0, 0, 0};
    const double ax = fabs(p[0]), ay = fabs(p[1]), az = fabs(p[2]);
    if (ax <= ay && ax <= az) e[0] = 1; else if (ay <= az) e[1] = 1; else e[2] = 1;
    cross3(p, e, u);
    const double un = sqrt(dot3(u, u));
    u[0] /= un; u[1] /= un; u[2] /= un;
    cross3(p, u, v);
    const double vn = sqrt(dot3(v, v));
    v[0] /= vn; v[1] /= vn; v[2] /= vn;
    const double R = 4 * big;
    const double sg[4][2] = {{1, 1}, {-1, 1}, {-1, -1}, {1, -1}};
    for (int c = 0; c < 4; ++c)
        for (int d = 0; d < 3; ++d)
            poly.set(c, d, c0[d] + R * (sg[c][0] * u[d] + sg[c][1] * v[d]));
    int m = 4;
    double far2 = 3 * (5 * big) * (5 * big); // squared distance of the farthest vertex (upper bound until first cut)
    for (int k = 0; k < nc && m >= 3; ++k) {
        if (k == f)
            continue;
        if (k >= first_sorted && dist[k] * dist[k] > far2)
            break; // this plane and all later ones pass beyond the farthest vertex
        const int before = m;
        m = clip_poly(poly, m, nrm[k], off[k]);
        if (m < 0) {
            r.overflow = true;
            return r;
        }
        if (m != before || true) { // vertices may have moved: refresh the farthest-vertex bound
            double mx = 0;
            for (int c = 0; c < m; ++c) {
                const double x = poly.get(c, 0), y = poly.get(c, 1), z = poly.get(c, 2);
                mx = fmax(mx, x * x + y * y + z * z);
            }
            far2 = mx;
        }
    }
    if (m < 3)
        return r;
    // area of the planar polygon: half the norm of the summed cross products of a fan from vertex 0
    double a0[3] = {poly.get(0, 0), poly.get(0, 1), poly.get(0, 2)};
    double acc[3] = {0, 0, 0}, prev[3] = {0, 0, 0};
    double mx = a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2];
    for (int c = 1; c < m; ++c) {
        const double x = poly.get(c, 0), y = poly.get(c, 1), z = poly.get(c, 2);
        mx = fmax(mx, x * x + y * y + z * z);
        const double cur[3] = {x - a0[0], y - a0[1], z - a0[2]};
        if (c >= 2) {
            double cr[3];
            cross3(prev, cur, cr);
            acc[0] += cr[0]; acc[1] += cr[1]; acc[2] += cr[2];
        }
        prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
    }
    r.area = 0.5 * sqrt(dot3(acc, acc));
    r.maxr2 = mx;
    r.nv = m;
    return r;
}

// The same face with its polygon in the 2-D coordinates of its own plane (as ptm_core.hpp face_solid_angle_2d): a vertex (a, b)
// is the point c0 + a u + b v, c0 the foot of the perpendicular from the atom (so |vertex|^2 = |c0|^2 + a^2 + b^2: the
// farthest-vertex bound costs no reconstruction), constraint k cuts the plane in the half-plane
// a (u.n_k) + b (v.n_k) <= o_k - c0.n_k, and the area is the shoelace sum.  Two thirds of the polygon storage, two
// multiplications per vertex and cut instead of three, no run-time modulo (clip_poly2).  The basis comes back in the
// result for callers that want the vertices in space (vertex()).
struct FaceResult2 {
    double area, maxr2;
    bool overflow;
    int nv;
    double c0[3], u[3], v[3];
    template <class P> PTM_HD void vertex(const P &poly, int c, double *out) const
    {
        const double a = poly.get(c, 0), b = poly.get(c, 1);
        out[0] = c0[0] + a * u[0] + b * v[0]; out[1] = c0[1] + a * u[1] + b * v[1]; out[2] = c0[2] + a * u[2] + b * v[2];
    }
};

template <class P>
PTM_HDN FaceResult2 voronoi_face_2d(P &poly, int f, int nc, const double (*nrm)[3], const double *off, const double *dist,
                                    int first_sorted, double big)
{
    FaceResult2 r;
    r.area = 0.0; r.maxr2 = 0.0; r.overflow = false; r.nv = 0;
    const double *p = nrm[f];
    const double pn2 = dot3(p, p);
    if (!(pn2 > 0))
        return r;
    const double s = off[f] / pn2;
    r.c0[0] = p[0] * s; r.c0[1] = p[1] * s; r.c0[2] = p[2] * s;
    double e[3] = {0, 0, 0};
    const double ax = fabs(p[0]), ay = fabs(p[1]), az = fabs(p[2]);
    if (ax <= ay && ax <= az) e[0] = 1; else if (ay <= az) e[1] = 1; else e[2] = 1;
    cross3(p, e, r.u);
    const double iu = 1.0 / sqrt(dot3(r.u, r.u));
    r.u[0] *= iu; r.u[1] *= iu; r.u[2] *= iu;
    cross3(p, r.u, r.v);
    const double iv = 1.0 / sqrt(dot3(r.v, r.v));
    r.v[0] *= iv; r.v[1] *= iv; r.v[2] *= iv;
    const double R = 4 * big;
    poly.set(0, 0, R); poly.set(0, 1, R);
    poly.set(1, 0, -R); poly.set(1, 1, R);
    poly.set(2, 0, -R); poly.set(2, 1, -R);
    poly.set(3, 0, R); poly.set(3, 1, -R);
    int m = 4;
    const double c02 = dot3(r.c0, r.c0);
    double far2 = 3 * (5 * big) * (5 * big); // squared distance of the farthest vertex (upper bound until first cut)
    for (int k = 0; k < nc && m >= 3; ++k) {
        if (k == f)
            continue;
        if (k >= first_sorted && dist[k] * dist[k] > far2)
            break; // this plane and all later ones pass beyond the farthest vertex
        m = ptmc::clip_poly2(poly, m, dot3(r.u, nrm[k]), dot3(r.v, nrm[k]), off[k] - dot3(r.c0, nrm[k]));
        if (m < 0) {
            r.overflow = true;
            return r;
        }
        double mx = 0;
        for (int c = 0; c < m; ++c) {
            const double a = poly.get(c, 0), b = poly.get(c, 1);
            mx = fmax(mx, a * a + b * b);
        }
        far2 = c02 + mx;
    }
    if (m < 3)
        return r;
    double twice = 0, mx = 0;
    for (int c = 0; c < m; ++c) { // shoelace
        const int d = c + 1 < m ? c + 1 : 0;
        const double a = poly.get(c, 0), b = poly.get(c, 1);
        twice += a * poly.get(d, 1) - poly.get(d, 0) * b;
        mx = fmax(mx, a * a + b * b);
    }
    r.area = 0.5 * fabs(twice);
    r.maxr2 = c02 + mx;
    r.nv = m;
    return r;
}

// Does voro++ have this face?  A plane that only touches the cell — through an edge or a vertex of a perfect lattice's cell — makes
// no face there (its vertex tolerance snaps the contact onto the plane) where exact clipping leaves a needle; a plane that really
// cuts a corner off makes one however small.  Until round 6 the rule was an area (below 1e-14 d^2 = debris), which miscounts the
// faces of 5 % of the atoms of a lattice rattled by 1e-6 A and now and then one of an ordinary system (profiles/r06_fuzz.txt);
// the two kinds differ in WIDTH by five orders of magnitude (ptm_core.hpp SLIVER_WIDTH), not in area.
constexpr double SLIVER_WIDTH = 1e-11; // (needles <= 2.2e-15 A, kept slivers >= 2.8e-10 A: tools/voro_sliver_scan.py)
template <class P> PTM_HD bool face_exists(const FaceResult2 &r, const P &poly, double d)
{
    if (!(r.area > 0.0) || r.nv < 3)
        return false;
    if (r.area >= 1e-8 * d * d)
        return true;
    return ptmc::poly_width2(poly, r.nv) >= SLIVER_WIDTH;
}

} // namespace voroc
