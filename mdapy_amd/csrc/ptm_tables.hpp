// ptm_tables.hpp — host-side GENERATION of every table the PTM kernels need (ptm_core.hpp: Tables).
//
// The reference ships these as literal data (extern/ptm/ptm_graph_data.cpp: 8/16/1/1/218 graph classes with
// their automorphisms; ptm_fundamental_mappings.h: symmetry permutations; ptm_quat.h:15-160: generator
// quaternions).  Here only the conventional template coordinates are stated (the point ORDER is part of the output
// format: ptm_indices and the alloy codes refer to it — extern/ptm/ptm_templates.h:21-104); everything else is
// derived from them at start-up:
//   * hull faces of the ideal template, every triangulation of its quadrilateral faces (2^6, 2^6, 2^12 ...),
//     grouped into classes by canonical code  -> graph tables
//   * traversals reproducing the canonical code                                  -> automorphisms
//   * proper rotations carrying template 0 onto itself (or onto its variants)    -> permutations + quaternions
// tests/test_ptm_host.py checks the generated sets against the reference's literal tables through oracle/_ref.
#pragma once
#include "ptm_core.hpp"
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

namespace ptmc {

struct TemplateDef {
    int type, num_nbrs, max_degree, num_variants;
    double pts[4][MAX_PTS][3];
};

inline void tables_fill_templates(std::vector<TemplateDef> &defs)
{
    defs.clear();
    const double s2 = std::sqrt(2.0) / 2, s3 = std::sqrt(3.0), s6 = std::sqrt(6.0), s5 = std::sqrt(5.0);
    auto put = [](TemplateDef &d, int v, int i, double x, double y, double z) { d.pts[v][i][0] = x; d.pts[v][i][1] = y; d.pts[v][i][2] = z; };
    { // simple cubic: -z +z -y +y -x +x
        TemplateDef d{};
        d.type = T_SC; d.num_nbrs = 6; d.max_degree = 4; d.num_variants = 1;
        const int v[6][3] = {{0, 0, -1}, {0, 0, 1}, {0, -1, 0}, {0, 1, 0}, {-1, 0, 0}, {1, 0, 0}};
        for (int i = 0; i < 6; ++i) put(d, 0, i + 1, v[i][0], v[i][1], v[i][2]);
        defs.push_back(d);
    }
    { // fcc: <110>/sqrt(2); the three cyclic (+,+) first, their negatives, then the mixed-sign ones
        TemplateDef d{};
        d.type = T_FCC; d.num_nbrs = 12; d.max_degree = 6; d.num_variants = 1;
        const int v[12][3] = {{1, 1, 0}, {0, 1, 1}, {1, 0, 1}, {-1, -1, 0}, {0, -1, -1}, {-1, 0, -1},
                              {-1, 1, 0}, {0, -1, 1}, {-1, 0, 1}, {1, -1, 0}, {0, 1, -1}, {1, 0, -1}};
        for (int i = 0; i < 12; ++i) put(d, 0, i + 1, v[i][0] * s2, v[i][1] * s2, v[i][2] * s2);
        defs.push_back(d);
    }
    { // hcp, c along z, unit neighbour distance; variant 1 = the other layer's environment
        TemplateDef d{};
        d.type = T_HCP; d.num_nbrs = 12; d.max_degree = 6; d.num_variants = 2;
        const double h = s3 / 2, t6 = s3 / 6, t3 = s3 / 3, c = s6 / 3;
        const double a[12][3] = {{0.5, -h, 0}, {-1, 0, 0}, {-0.5, t6, -c}, {0.5, t6, -c}, {0, -t3, -c}, {-0.5, h, 0},
                                 {0.5, h, 0}, {1, 0, 0}, {-0.5, -h, 0}, {0, -t3, c}, {0.5, t6, c}, {-0.5, t6, c}};
        const double b[12][3] = {{1, 0, 0}, {-0.5, -h, 0}, {-0.5, -t6, -c}, {0, t3, -c}, {0.5, -t6, -c}, {-1, 0, 0},
                                 {-0.5, h, 0}, {0.5, h, 0}, {0.5, -h, 0}, {0.5, -t6, c}, {0, t3, c}, {-0.5, -t6, c}};
        for (int i = 0; i < 12; ++i) { put(d, 0, i + 1, a[i][0], a[i][1], a[i][2]); put(d, 1, i + 1, b[i][0], b[i][1], b[i][2]); }
        defs.push_back(d);
    }
    { // icosahedron, five-fold axis along z
        TemplateDef d{};
        d.type = T_ICO; d.num_nbrs = 12; d.max_degree = 6; d.num_variants = 1;
        const double am = std::sqrt((5 - s5) / 10), ap = std::sqrt((5 + s5) / 10), bp = (5 + s5) / 10, bm = (5 - s5) / 10, z = s5 / 5;
        const double v[12][3] = {{0, 0, 1}, {0, 0, -1}, {-am, bp, -z}, {am, -bp, z}, {0, -2 * z, -z}, {0, 2 * z, z},
                                 {ap, -bm, -z}, {-ap, bm, z}, {-ap, -bm, -z}, {ap, bm, z}, {am, bp, -z}, {-am, -bp, z}};
        for (int i = 0; i < 12; ++i) put(d, 0, i + 1, v[i][0], v[i][1], v[i][2]);
        defs.push_back(d);
    }
    { // bcc: 8 x <111> then 6 x <200>, scaled so that the mean neighbour distance is 1
        TemplateDef d{};
        d.type = T_BCC; d.num_nbrs = 14; d.max_degree = 8; d.num_variants = 1;
        const double u = 7 * s3 / 3 - 7. / 2, w = 14 * s3 / 3 - 7;
        const int v[8][3] = {{1, 1, 1}, {-1, 1, 1}, {1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {-1, 1, -1}, {-1, -1, -1}, {1, -1, -1}};
        for (int i = 0; i < 8; ++i) put(d, 0, i + 1, v[i][0] * u, v[i][1] * u, v[i][2] * u);
        for (int a = 0; a < 3; ++a) {
            double p[3] = {0, 0, 0}, m[3] = {0, 0, 0};
            p[a] = w; m[a] = -w;
            put(d, 0, 9 + 2 * a, p[0], p[1], p[2]);
            put(d, 0, 10 + 2 * a, m[0], m[1], m[2]);
        }
        defs.push_back(d);
    }
    { // diamond cubic: 4 x <111>*u then, per inner atom, its three further bonds = 12 x <220>*u; variant 1 = the other sublattice
        TemplateDef d{};
        d.type = T_DCUB; d.num_nbrs = 16; d.max_degree = 8; d.num_variants = 2;
        const double u = 4 / (s3 + 6 * std::sqrt(2.0));
        const int in0[4][3] = {{1, 1, 1}, {1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}};
        const int out0[12][3] = {{2, 2, 0}, {0, 2, 2}, {2, 0, 2}, {0, -2, -2}, {2, -2, 0}, {2, 0, -2},
                                 {-2, -2, 0}, {0, -2, 2}, {-2, 0, 2}, {-2, 0, -2}, {-2, 2, 0}, {0, 2, -2}};
        const int in1[4][3] = {{1, -1, 1}, {1, 1, -1}, {-1, -1, -1}, {-1, 1, 1}};
        const int out1[12][3] = {{2, 0, 2}, {0, -2, 2}, {2, -2, 0}, {0, 2, -2}, {2, 0, -2}, {2, 2, 0},
                                 {-2, 0, -2}, {0, -2, -2}, {-2, -2, 0}, {-2, 2, 0}, {-2, 0, 2}, {0, 2, 2}};
        for (int i = 0; i < 4; ++i) { put(d, 0, 1 + i, in0[i][0] * u, in0[i][1] * u, in0[i][2] * u); put(d, 1, 1 + i, in1[i][0] * u, in1[i][1] * u, in1[i][2] * u); }
        for (int i = 0; i < 12; ++i) { put(d, 0, 5 + i, out0[i][0] * u, out0[i][1] * u, out0[i][2] * u); put(d, 1, 5 + i, out1[i][0] * u, out1[i][1] * u, out1[i][2] * u); }
        defs.push_back(d);
    }
    { // diamond hexagonal (lonsdaleite), c along z: bond length L, three bonds at z = -/+ L/3, one along +/- z.
      // Coordinates in units (a, b, c, e, f) = (L sqrt(2/3), L sqrt(2)/3, L/3, L, 4L/3): x multiples of a, y of b, z of c|e|f.
        TemplateDef d{};
        d.type = T_DHEX; d.num_nbrs = 16; d.max_degree = 8; d.num_variants = 4;
        const double L = 16 / (4 + 12 * std::sqrt(8.0 / 3.0));
        const double a = L * std::sqrt(2.0 / 3.0), b = L * std::sqrt(2.0) / 3, c = L / 3, f = 4 * L / 3;
        // each row: x/a, y/b, z-code (0: 0, 1: c, 2: L, 3: f; sign carried by the code's sign)
        static const int tab[4][16][3] = {
            {{-1, 1, -1}, {0, -2, -1}, {1, 1, -1}, {0, 0, 2}, {-2, 0, 0}, {-1, 1, -3}, {-1, 3, 0}, {1, -3, 0}, {0, -2, -3}, {-1, -3, 0},
             {1, 1, -3}, {1, 3, 0}, {2, 0, 0}, {0, -2, 3}, {1, 1, 3}, {-1, 1, 3}},
            {{-1, -1, -1}, {1, -1, -1}, {0, 2, -1}, {0, 0, 2}, {-1, -3, 0}, {-1, -1, -3}, {-2, 0, 0}, {2, 0, 0}, {1, -1, -3}, {1, -3, 0},
             {0, 2, -3}, {-1, 3, 0}, {1, 3, 0}, {1, -1, 3}, {0, 2, 3}, {-1, -1, 3}},
            {{0, -2, 1}, {-1, 1, 1}, {1, 1, 1}, {0, 0, -2}, {-1, -3, 0}, {0, -2, 3}, {1, -3, 0}, {-1, 3, 0}, {-1, 1, 3}, {-2, 0, 0},
             {1, 1, 3}, {2, 0, 0}, {1, 3, 0}, {-1, 1, -3}, {1, 1, -3}, {0, -2, -3}},
            {{1, -1, 1}, {-1, -1, 1}, {0, 2, 1}, {0, 0, -2}, {1, -3, 0}, {1, -1, 3}, {2, 0, 0}, {-2, 0, 0}, {-1, -1, 3}, {-1, -3, 0},
             {0, 2, 3}, {1, 3, 0}, {-1, 3, 0}, {-1, -1, -3}, {0, 2, -3}, {1, -1, -3}}};
        for (int v = 0; v < 4; ++v)
            for (int i = 0; i < 16; ++i) {
                const int zc = tab[v][i][2];
                const double zz = (zc < 0 ? -1 : 1) * (std::abs(zc) == 1 ? c : std::abs(zc) == 2 ? L : std::abs(zc) == 3 ? f : 0.0);
                put(d, v, 1 + i, tab[v][i][0] * a, tab[v][i][1] * b, zz);
            }
        defs.push_back(d);
    }
    { // graphene, in the xy plane: three bonds of length L, then the two further bonds of each
        TemplateDef d{};
        d.type = T_GRAPHENE; d.num_nbrs = 9; d.max_degree = -1; d.num_variants = 2;
        const double L = 3 / (1 + 2 * s3), hx = L * s3 / 2, hy = L / 2;
        // x in units of hx, y in units of hy
        static const int tab[2][9][2] = {{{0, 2}, {1, -1}, {-1, -1}, {-1, 3}, {1, 3}, {2, 0}, {1, -3}, {-1, -3}, {-2, 0}},
                                         {{-1, 1}, {1, 1}, {0, -2}, {-2, 0}, {-1, 3}, {1, 3}, {2, 0}, {1, -3}, {-1, -3}}};
        for (int v = 0; v < 2; ++v)
            for (int i = 0; i < 9; ++i) put(d, v, 1 + i, tab[v][i][0] * hx, tab[v][i][1] * hy, 0.0);
        defs.push_back(d);
    }
}

// ---- hull faces of the ideal template (neighbours only, index = template index - 1) -----------------------
struct Face { std::vector<int> v; }; // vertices in cyclic order

inline void template_faces(int nn, const double (*p)[3], std::vector<Face> &faces)
{
    faces.clear();
    std::set<std::vector<int>> seen;
    for (int a = 0; a < nn; ++a)
        for (int b = a + 1; b < nn; ++b)
            for (int c = b + 1; c < nn; ++c) {
                double u[3] = {p[b][0] - p[a][0], p[b][1] - p[a][1], p[b][2] - p[a][2]};
                double v[3] = {p[c][0] - p[a][0], p[c][1] - p[a][1], p[c][2] - p[a][2]};
                double n[3];
                cross3(u, v, n);
                const double nl = std::sqrt(dot3(n, n));
                if (nl < 1e-9) continue;
                n[0] /= nl; n[1] /= nl; n[2] /= nl;
                int pos = 0, neg = 0;
                std::vector<int> on;
                for (int i = 0; i < nn; ++i) {
                    const double d = n[0] * (p[i][0] - p[a][0]) + n[1] * (p[i][1] - p[a][1]) + n[2] * (p[i][2] - p[a][2]);
                    if (d > 1e-9) ++pos; else if (d < -1e-9) ++neg; else on.push_back(i);
                }
                if (pos && neg) continue;
                if (!seen.insert(on).second) continue;
                // cyclic order around the face centre
                if (pos) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; } // outward
                double cen[3] = {0, 0, 0};
                for (int i : on) { cen[0] += p[i][0]; cen[1] += p[i][1]; cen[2] += p[i][2]; }
                for (double &x : cen) x /= on.size();
                double e1[3] = {p[on[0]][0] - cen[0], p[on[0]][1] - cen[1], p[on[0]][2] - cen[2]}, e2[3];
                cross3(n, e1, e2);
                std::vector<std::pair<double, int>> ang;
                for (int i : on) {
                    const double r[3] = {p[i][0] - cen[0], p[i][1] - cen[1], p[i][2] - cen[2]};
                    ang.push_back({std::atan2(dot3(r, e2), dot3(r, e1)), i});
                }
                std::sort(ang.begin(), ang.end());
                Face f;
                for (auto &q : ang) f.v.push_back(q.second);
                faces.push_back(f);
            }
}

// orientation rule of the hull code (add_facet with the template centre as the inside point)
inline void orient_facet(const double (*p)[3], int a, int b, int c, int8_t *out)
{
    double n[3];
    plane_normal(p, a, b, c, n);
    const double zero[3] = {0, 0, 0};
    if (plane_dist(zero, p[a], n) > 0) { out[0] = (int8_t)b; out[1] = (int8_t)a; out[2] = (int8_t)c; }
    else { out[0] = (int8_t)a; out[1] = (int8_t)b; out[2] = (int8_t)c; }
}

// all traversals reproducing the best code -> automorphisms (as template-index permutations, aut[0] = 0)
inline void graph_automorphisms(int nf, const int8_t (*f)[3], int nn, const int8_t *best_code, const int8_t *canon,
                                const int8_t *colours, std::vector<std::array<int8_t, MAX_PTS>> &auts)
{
    auts.clear();
    const int ne = 3 * nf / 2;
    std::set<std::array<int8_t, MAX_PTS>> uniq;
    for (int i = 0; i < nf; ++i)
        for (int e = 0; e < 3; ++e) {
            Canon C;
            for (int a = 0; a < MAX_NBR; ++a)
                for (int b = 0; b < MAX_NBR; ++b) C.common[a][b] = -1;
            for (int k = 0; k < nf; ++k) {
                const int a = f[k][0], b = f[k][1], c = f[k][2];
                C.common[a][b] = (int8_t)c; C.common[b][c] = (int8_t)a; C.common[c][a] = (int8_t)b;
            }
            for (int k = 0; k < 2 * MAX_EDGES; ++k) C.best[k] = 127;
            weinberg(nn, ne, C, colours, f[i][e], f[i][(e + 1) % 3]);
            if (std::memcmp(C.best, best_code, 2 * ne) != 0) continue;
            // C.label[v] = traversal index of neighbour v; canonical label of template point v+1 = label+1
            // the automorphism sends the point with canonical label L to the point that got L in this traversal
            std::array<int8_t, MAX_PTS> a{};
            a[0] = 0;
            int8_t inv_here[MAX_PTS];
            for (int v = 0; v < nn; ++v) inv_here[(C.label[v] % nn) + 1] = (int8_t)(v + 1);
            for (int v = 1; v <= nn; ++v) a[v] = inv_here[canon[v]];
            if (uniq.insert(a).second) auts.push_back(a);
        }
    std::sort(auts.begin(), auts.end());
}

// ---- proper rotations taking template 0 onto variant t ---------------------------------------------------
inline void matrix_to_quat(const double *m, double *q)
{
    const double tr = m[0] + m[4] + m[8];
    if (tr > 0) {
        const double s = std::sqrt(tr + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (m[7] - m[5]) / s; q[2] = (m[2] - m[6]) / s; q[3] = (m[3] - m[1]) / s;
    } else if (m[0] > m[4] && m[0] > m[8]) {
        const double s = std::sqrt(1.0 + m[0] - m[4] - m[8]) * 2;
        q[0] = (m[7] - m[5]) / s; q[1] = 0.25 * s; q[2] = (m[1] + m[3]) / s; q[3] = (m[2] + m[6]) / s;
    } else if (m[4] > m[8]) {
        const double s = std::sqrt(1.0 + m[4] - m[0] - m[8]) * 2;
        q[0] = (m[2] - m[6]) / s; q[1] = (m[1] + m[3]) / s; q[2] = 0.25 * s; q[3] = (m[5] + m[7]) / s;
    } else {
        const double s = std::sqrt(1.0 + m[8] - m[0] - m[4]) * 2;
        q[0] = (m[3] - m[1]) / s; q[1] = (m[2] + m[6]) / s; q[2] = (m[5] + m[7]) / s; q[3] = 0.25 * s;
    }
    if (q[0] < 0 || (q[0] == 0 && (q[1] < 0 || (q[1] == 0 && (q[2] < 0 || (q[2] == 0 && q[3] < 0))))))
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
    for (int i = 0; i < 4; ++i)
        if (std::fabs(q[i]) < 1e-15) q[i] = 0;
}

struct SymOp {
    double g[4];                       // generator quaternion (q_new = q * g)
    std::array<int8_t, MAX_PTS> perm;  // permute_mapping permutation
};

// rotations R with R * template0 == variant (as sets).  For each: perm[i] = index in the variant of R*p0_i, and the
// generator is the quaternion of R^-1 (see DESIGN.md §PTM for the derivation of the convention).
inline void template_rotations(int np, const double (*p0)[3], const double (*pt)[3], std::vector<SymOp> &ops)
{
    ops.clear();
    int i1 = 1, i2 = -1;
    for (int k = 2; k < np; ++k) {
        double c[3];
        cross3(p0[i1], p0[k], c);
        if (dot3(c, c) > 1e-6) { i2 = k; break; }
    }
    double c0[3];
    cross3(p0[i1], p0[i2], c0);
    const double B[9] = {p0[i1][0], p0[i2][0], c0[0], p0[i1][1], p0[i2][1], c0[1], p0[i1][2], p0[i2][2], c0[2]};
    const double det = B[0] * (B[4] * B[8] - B[5] * B[7]) - B[1] * (B[3] * B[8] - B[5] * B[6]) + B[2] * (B[3] * B[7] - B[4] * B[6]);
    double Bi[9];
    Bi[0] = (B[4] * B[8] - B[5] * B[7]) / det; Bi[1] = -(B[1] * B[8] - B[2] * B[7]) / det; Bi[2] = (B[1] * B[5] - B[2] * B[4]) / det;
    Bi[3] = -(B[3] * B[8] - B[5] * B[6]) / det; Bi[4] = (B[0] * B[8] - B[2] * B[6]) / det; Bi[5] = -(B[0] * B[5] - B[2] * B[3]) / det;
    Bi[6] = (B[3] * B[7] - B[4] * B[6]) / det; Bi[7] = -(B[0] * B[7] - B[1] * B[6]) / det; Bi[8] = (B[0] * B[4] - B[1] * B[3]) / det;
    const double l1 = dot3(p0[i1], p0[i1]), l2 = dot3(p0[i2], p0[i2]), l12 = dot3(p0[i1], p0[i2]);
    for (int a = 1; a < np; ++a)
        for (int b = 1; b < np; ++b) {
            if (a == b) continue;
            if (std::fabs(dot3(pt[a], pt[a]) - l1) > 1e-9 || std::fabs(dot3(pt[b], pt[b]) - l2) > 1e-9 ||
                std::fabs(dot3(pt[a], pt[b]) - l12) > 1e-9)
                continue;
            double cq[3];
            cross3(pt[a], pt[b], cq);
            const double Q[9] = {pt[a][0], pt[b][0], cq[0], pt[a][1], pt[b][1], cq[1], pt[a][2], pt[b][2], cq[2]};
            double R[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) R[r * 3 + c] = Q[r * 3] * Bi[c] + Q[r * 3 + 1] * Bi[3 + c] + Q[r * 3 + 2] * Bi[6 + c];
            SymOp op;
            op.perm.fill(0);
            bool ok = true;
            for (int i = 1; i < np && ok; ++i) {
                const double v[3] = {R[0] * p0[i][0] + R[1] * p0[i][1] + R[2] * p0[i][2], R[3] * p0[i][0] + R[4] * p0[i][1] + R[5] * p0[i][2],
                                     R[6] * p0[i][0] + R[7] * p0[i][1] + R[8] * p0[i][2]};
                int hit = -1;
                for (int j = 1; j < np; ++j) {
                    const double d[3] = {v[0] - pt[j][0], v[1] - pt[j][1], v[2] - pt[j][2]};
                    if (dot3(d, d) < 1e-12) { hit = j; break; }
                }
                if (hit < 0) ok = false; else op.perm[i] = (int8_t)hit;
            }
            if (!ok) continue;
            const double Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
            matrix_to_quat(Rt, op.g);
            ops.push_back(op);
        }
}

inline void sort_ops(std::vector<SymOp> &ops)
{ // identity first, then by decreasing scalar part, then lexicographically: a fixed, reproducible order
    std::stable_sort(ops.begin(), ops.end(), [](const SymOp &a, const SymOp &b) {
        for (int k = 0; k < 4; ++k) {
            if (std::fabs(a.g[k] - b.g[k]) > 1e-12) return a.g[k] > b.g[k];
        }
        return false;
    });
}

// Builds everything.  Returns an empty string on success, otherwise what went wrong.
inline std::string tables_generate(Tables &T)
{
    std::memset(&T, 0, sizeof(T));
    std::vector<TemplateDef> defs;
    tables_fill_templates(defs);
    for (const TemplateDef &d : defs) {
        TypeInfo &ti = T.types[d.type];
        ti.type = d.type;
        ti.num_nbrs = d.num_nbrs;
        ti.max_degree = d.max_degree;
        const int np = d.num_nbrs + 1, nn = d.num_nbrs;
        std::memcpy(ti.points, d.pts[0], sizeof(ti.points));
        // --- symmetry operations -----------------------------------------------------------------------
        std::vector<SymOp> self, conv;
        template_rotations(np, d.pts[0], d.pts[0], self);
        sort_ops(self);
        conv = self;
        for (int v = 1; v < d.num_variants; ++v) {
            std::vector<SymOp> o;
            template_rotations(np, d.pts[0], d.pts[v], o);
            conv.insert(conv.end(), o.begin(), o.end());
        }
        sort_ops(conv);
        const bool shared = d.num_variants == 1; // the remap group is the template's own rotation group
        if (T.num_maps + (int)self.size() + (shared ? 0 : (int)conv.size()) > MAX_MAPS || T.num_gens + (int)conv.size() > MAX_GENS)
            return "ptm tables: symmetry table overflow";
        ti.map_begin = T.num_maps;
        ti.num_maps = (int)self.size();
        for (const SymOp &o : self) std::memcpy(T.maps[T.num_maps++], o.perm.data(), MAX_PTS);
        ti.conv_begin = shared ? ti.map_begin : T.num_maps;
        ti.num_conv = (int)conv.size();
        ti.gen_begin = T.num_gens;
        for (const SymOp &o : conv) {
            if (!shared) std::memcpy(T.maps[T.num_maps++], o.perm.data(), MAX_PTS);
            std::memcpy(T.gens[T.num_gens++], o.g, sizeof(o.g));
        }
        // --- graph classes ------------------------------------------------------------------------------
        if (d.type == T_GRAPHENE) { // matched by direct trial (match_graphene), no graphs
            ti.num_facets = 0;
            ti.graph_begin = T.num_graphs;
            ti.num_graphs = 0;
            continue;
        }
        // diamond types: the graph is the hull of the 12 outer atoms in which the facet spanned by an inner atom's three
        // outer atoms is replaced by the three facets through that inner atom (what match_dcub_dhex builds); inner
        // atoms carry colour 1
        const bool two_shell = d.type == T_DCUB || d.type == T_DHEX;
        const int first = two_shell ? 4 : 0; // neighbour index of the first hull vertex
        int8_t colours[MAX_PTS] = {0};
        if (two_shell) colours[0] = colours[1] = colours[2] = colours[3] = 1;
        std::vector<Face> faces;
        template_faces(nn - first, d.pts[0] + 1 + first, faces);
        for (Face &fc : faces)
            for (int &v : fc.v) v += first;
        std::vector<int> quads;
        int ntri = 0;
        for (size_t i = 0; i < faces.size(); ++i) {
            if (faces[i].v.size() == 4) quads.push_back((int)i);
            else if (faces[i].v.size() != 3) return "ptm tables: template face is neither triangle nor quadrilateral";
            ntri += (int)faces[i].v.size() - 2;
        }
        if (two_shell) ntri += 8; // four facets become twelve
        ti.num_facets = ntri;
        if (ntri > MAX_FACETS || quads.size() > 20) return "ptm tables: template too large";
        ti.graph_begin = T.num_graphs;
        // one class per ORBIT of triangulations under the template's proper rotations (two triangulations can share a
        // canonical code without being related by a rotation; the matcher then tries both representatives)
        std::set<std::vector<std::array<int8_t, 3>>> seen_orbits;
        auto normalised = [](int nf, const int8_t (*f)[3], const int8_t *perm) {
            std::vector<std::array<int8_t, 3>> v((size_t)nf);
            for (int i = 0; i < nf; ++i) {
                int8_t a = f[i][0], b = f[i][1], c = f[i][2];
                if (perm) { a = (int8_t)(perm[a + 1] - 1); b = (int8_t)(perm[b + 1] - 1); c = (int8_t)(perm[c + 1] - 1); }
                while (!(a <= b && a <= c)) { const int8_t t = a; a = b; b = c; c = t; }
                v[(size_t)i] = {a, b, c};
            }
            std::sort(v.begin(), v.end());
            return v;
        };
        for (uint32_t mask = 0; mask < (1u << quads.size()); ++mask) {
            int8_t f[MAX_FACETS][3];
            int nf = 0;
            size_t qi = 0;
            for (size_t i = 0; i < faces.size(); ++i) {
                const std::vector<int> &v = faces[i].v;
                if (v.size() == 3) {
                    int8_t t[3];
                    orient_facet(d.pts[0] + 1, v[0], v[1], v[2], t);
                    const int i0 = (t[0] - 4) / 3, i1 = (t[1] - 4) / 3, i2 = (t[2] - 4) / 3;
                    if (two_shell && i0 == i1 && i0 == i2) {
                        const int8_t in = (int8_t)i0;
                        const int8_t add[3][3] = {{in, t[1], t[2]}, {t[0], in, t[2]}, {t[0], t[1], in}};
                        for (int q = 0; q < 3; ++q) { f[nf][0] = add[q][0]; f[nf][1] = add[q][1]; f[nf][2] = add[q][2]; ++nf; }
                    } else {
                        f[nf][0] = t[0]; f[nf][1] = t[1]; f[nf][2] = t[2];
                        ++nf;
                    }
                } else {
                    const bool alt = (mask >> qi++) & 1;
                    if (!alt) { orient_facet(d.pts[0] + 1, v[0], v[1], v[2], f[nf]); ++nf; orient_facet(d.pts[0] + 1, v[0], v[2], v[3], f[nf]); ++nf; }
                    else      { orient_facet(d.pts[0] + 1, v[1], v[2], v[3], f[nf]); ++nf; orient_facet(d.pts[0] + 1, v[1], v[3], v[0], f[nf]); ++nf; }
                }
            }
            if (seen_orbits.count(normalised(nf, f, nullptr))) continue;
            for (const SymOp &o : self) seen_orbits.insert(normalised(nf, f, o.perm.data()));
            int8_t deg[MAX_NBR];
            if (graph_degree(nf, f, nn, deg) > d.max_degree) continue;
            Canon C;
            uint64_t hash = 0;
            if (canonical_form(nf, f, nn, deg, colours, C, &hash) != 0) return "ptm tables: template triangulation is not a closed surface";
            if (T.num_graphs >= MAX_GRAPHS) return "ptm tables: graph table overflow";
            Graph &g = T.graphs[T.num_graphs++];
            g.hash = hash;
            std::memcpy(g.canon, C.label, MAX_PTS);
            std::vector<std::array<int8_t, MAX_PTS>> auts;
            graph_automorphisms(nf, f, nn, C.best, C.label, colours, auts);
            { // automorphisms that differ by a rotation of the template give the same rmsd: keep one per coset
                std::set<std::array<int8_t, MAX_PTS>> kept;
                std::vector<std::array<int8_t, MAX_PTS>> out;
                for (const auto &a : auts) {
                    bool dup = false;
                    for (const SymOp &o : self) {
                        std::array<int8_t, MAX_PTS> b{};
                        for (int k = 0; k < np; ++k) b[k] = o.perm[a[k]];
                        if (kept.count(b)) { dup = true; break; }
                    }
                    if (!dup) { kept.insert(a); out.push_back(a); }
                }
                auts.swap(out);
            }
            if (T.num_auts + (int)auts.size() > MAX_AUTS) return "ptm tables: automorphism table overflow";
            g.aut_begin = T.num_auts;
            g.num_aut = (int)auts.size();
            for (auto &a : auts) std::memcpy(T.auts[T.num_auts++], a.data(), MAX_PTS);
        }
        ti.num_graphs = T.num_graphs - ti.graph_begin;
    }
    return "";
}

} // namespace ptmc
