// voro_ref_driver.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref).
//
// C-ABI driver around the REFERENCE's own Voronoi library (extern/voro++, plain C++, C. Rycroft), compiled from the
// sources where they lie (unity file extern/voro++/src/voro++.cc, as the reference's CMakeLists.txt:73-74 does) by
// oracle/Makefile.ref into oracle/_ref/libvoro_ref.so.  The reference's driver src/voronoi.cpp includes nanobind and
// cannot be compiled in this image; its logic is restated here:
//   get_voronoi_volume_number_radius       src/voronoi.cpp:16-71
//   get_voronoi_volume_number_radius_tri   src/voronoi.cpp:73-147
//   get_voronoi_neighbor (orthogonal)      src/voronoi.cpp:307-447   (unfiltered: face areas and neighbour ids per cell)
//   get_voronoi_neighbor_tri               src/voronoi.cpp:149-305   (likewise)
#include "voro++.hh"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {

// box9 row-major cell vectors (orthogonal), origin, boundary flags; outputs volume, number of faces, cavity radius
__attribute__((visibility("default"))) int ref_voronoi_volume_number_radius(const double *x, const double *y, const double *z, int64_t N,
                                                                            const double *box9, const double *origin, const int *boundary,
                                                                            double *volume, int *nfaces, double *radius)
{
    const double bx = box9[0], by = box9[4], bz = box9[8];
    const double vol = bx * by * bz, init_mem = 4.6;
    const double ilscale = std::pow(N / (init_mem * vol), 1 / 3.0);
    const int nx = int(bx * ilscale + 1), ny = int(by * ilscale + 1), nz = int(bz * ilscale + 1);
    voro::container_3d con(0., bx, 0., by, 0., bz, nx, ny, nz, bool(boundary[0]), bool(boundary[1]), bool(boundary[2]), init_mem, 1);
    for (int64_t i = 0; i < N; ++i) con.put((int)i, x[i] - origin[0], y[i] - origin[1], z[i] - origin[2]);
    voro::voronoicell_neighbor_3d cell(con);
    const int grid = con.nx * con.ny * con.nz;
    for (int ijk = 0; ijk < grid; ++ijk)
        for (int q = 0; q < con.co[ijk]; ++q)
            if (con.compute_cell(cell, ijk, q)) {
                const int i = con.id[ijk][q];
                volume[i] = cell.volume();
                nfaces[i] = cell.number_of_faces();
                radius[i] = std::sqrt(cell.max_radius_squared());
            }
    return 0;
}

// LAMMPS-aligned triclinic box (bx, bxy, by, bxz, byz, bz from rows of box9), fully periodic (voronoi.py multiplies the
// vectors of open axes by 3 beforehand); rotation (3,3) applied when need_rotation (:104-117)
__attribute__((visibility("default"))) int ref_voronoi_volume_number_radius_tri(const double *x, const double *y, const double *z, int64_t N,
                                                                                const double *box9, const double *origin, const double *rot9,
                                                                                int need_rotation, double *volume, int *nfaces, double *radius)
{
    const double bx = box9[0], bxy = box9[3], by = box9[4], bxz = box9[6], byz = box9[7], bz = box9[8];
    const double vol = std::fabs(bx * by * bz), init_mem = 4.6;
    const double ilscale = std::pow(N / (init_mem * vol), 1 / 3.0);
    auto len = [&](int r) { return std::sqrt(box9[3 * r] * box9[3 * r] + box9[3 * r + 1] * box9[3 * r + 1] + box9[3 * r + 2] * box9[3 * r + 2]); };
    const int nx = int(len(0) * ilscale + 1), ny = int(len(1) * ilscale + 1), nz = int(len(2) * ilscale + 1);
    voro::container_triclinic con(bx, bxy, by, bxz, byz, bz, nx, ny, nz, init_mem, 1);
    for (int64_t i = 0; i < N; ++i) {
        const double v[3] = {x[i] - origin[0], y[i] - origin[1], z[i] - origin[2]};
        if (need_rotation)
            con.put((int)i, v[0] * rot9[0] + v[1] * rot9[3] + v[2] * rot9[6], v[0] * rot9[1] + v[1] * rot9[4] + v[2] * rot9[7],
                    v[0] * rot9[2] + v[1] * rot9[5] + v[2] * rot9[8]);
        else
            con.put((int)i, v[0], v[1], v[2]);
    }
    std::vector<int> co(con.co, con.co + con.oxyz);
    voro::voronoicell_neighbor_3d cell(con);
    for (int ijk = 0; ijk < con.oxyz; ++ijk)
        for (int q = 0; q < co[ijk]; ++q)
            if (con.compute_cell(cell, ijk, q)) {
                const int i = con.id[ijk][q];
                volume[i] = cell.volume();
                nfaces[i] = cell.number_of_faces();
                radius[i] = std::sqrt(cell.max_radius_squared());
            }
    return 0;
}

// per-cell neighbour ids (negative = wall) and face areas, row width `width` (-1 / 0 padded); returns the largest count
__attribute__((visibility("default"))) int ref_voronoi_faces(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                                             const double *origin, const int *boundary, int *nbr, double *area, int width,
                                                             int *count)
{
    const double bx = box9[0], by = box9[4], bz = box9[8];
    const double vol = bx * by * bz, init_mem = 4.6;
    const double ilscale = std::pow(N / (init_mem * vol), 1 / 3.0);
    const int nx = int(bx * ilscale + 1), ny = int(by * ilscale + 1), nz = int(bz * ilscale + 1);
    voro::container_3d con(0., bx, 0., by, 0., bz, nx, ny, nz, bool(boundary[0]), bool(boundary[1]), bool(boundary[2]), init_mem, 1);
    for (int64_t i = 0; i < N; ++i) con.put((int)i, x[i] - origin[0], y[i] - origin[1], z[i] - origin[2]);
    voro::voronoicell_neighbor_3d cell(con);
    std::vector<int> ids;
    std::vector<double> ar;
    int mx = 0;
    for (int ijk = 0; ijk < con.nx * con.ny * con.nz; ++ijk)
        for (int q = 0; q < con.co[ijk]; ++q)
            if (con.compute_cell(cell, ijk, q)) {
                const int i = con.id[ijk][q];
                cell.neighbors(ids);
                cell.face_areas(ar);
                count[i] = (int)ids.size();
                mx = count[i] > mx ? count[i] : mx;
                for (int k = 0; k < width; ++k) {
                    nbr[(int64_t)i * width + k] = k < (int)ids.size() ? ids[k] : -1;
                    area[(int64_t)i * width + k] = k < (int)ids.size() ? ar[k] : 0.0;
                }
            }
    return mx;
}
// the same for the LAMMPS-aligned triclinic container of get_voronoi_neighbor_tri (src/voronoi.cpp:149-232)
__attribute__((visibility("default"))) int ref_voronoi_faces_tri(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                                                 const double *origin, const double *rot9, int need_rotation, int *nbr,
                                                                 double *area, int width, int *count)
{
    const double bx = box9[0], bxy = box9[3], by = box9[4], bxz = box9[6], byz = box9[7], bz = box9[8];
    const double vol = std::fabs(bx * by * bz), init_mem = 4.6;
    const double ilscale = std::pow(N / (init_mem * vol), 1 / 3.0);
    auto len = [&](int r) { return std::sqrt(box9[3 * r] * box9[3 * r] + box9[3 * r + 1] * box9[3 * r + 1] + box9[3 * r + 2] * box9[3 * r + 2]); };
    const int nx = int(len(0) * ilscale + 1), ny = int(len(1) * ilscale + 1), nz = int(len(2) * ilscale + 1);
    voro::container_triclinic con(bx, bxy, by, bxz, byz, bz, nx, ny, nz, init_mem, 1);
    for (int64_t i = 0; i < N; ++i) {
        const double v[3] = {x[i] - origin[0], y[i] - origin[1], z[i] - origin[2]};
        if (need_rotation)
            con.put((int)i, v[0] * rot9[0] + v[1] * rot9[3] + v[2] * rot9[6], v[0] * rot9[1] + v[1] * rot9[4] + v[2] * rot9[7],
                    v[0] * rot9[2] + v[1] * rot9[5] + v[2] * rot9[8]);
        else
            con.put((int)i, v[0], v[1], v[2]);
    }
    std::vector<int> co(con.co, con.co + con.oxyz);
    voro::voronoicell_neighbor_3d cell(con);
    std::vector<int> ids;
    std::vector<double> ar;
    int mx = 0;
    for (int ijk = 0; ijk < con.oxyz; ++ijk)
        for (int q = 0; q < co[ijk]; ++q)
            if (con.compute_cell(cell, ijk, q)) {
                const int i = con.id[ijk][q];
                cell.neighbors(ids);
                cell.face_areas(ar);
                count[i] = (int)ids.size();
                mx = count[i] > mx ? count[i] : mx;
                for (int k = 0; k < width; ++k) {
                    nbr[(int64_t)i * width + k] = k < (int)ids.size() ? ids[k] : -1;
                    area[(int64_t)i * width + k] = k < (int)ids.size() ? ar[k] : 0.0;
                }
            }
    return mx;
}
// get_cell_info (src/voronoi.cpp:449-540) flattened: per cell the number of faces and vertices, then CSR-like tables.
// Call once with the tables NULL to get the sizes (total face-vertex entries in *n_fv, total vertices in *n_vert).
__attribute__((visibility("default"))) int ref_voronoi_cell_info(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                                                 const double *origin, const int *boundary, int *nfaces, int *nverts,
                                                                 int64_t *n_fv, int64_t *n_vert, int *face_sizes, int *face_vertex_ids,
                                                                 double *vertices, double *face_area, double *volume, double *radius)
{
    const double bx = box9[0], by = box9[4], bz = box9[8];
    const double vol = bx * by * bz, init_mem = 4.6;
    const double ilscale = std::pow(N / (init_mem * vol), 1 / 3.0);
    const int nx = int(bx * ilscale + 1), ny = int(by * ilscale + 1), nz = int(bz * ilscale + 1);
    voro::container_3d con(0., bx, 0., by, 0., bz, nx, ny, nz, bool(boundary[0]), bool(boundary[1]), bool(boundary[2]), init_mem, 1);
    for (int64_t i = 0; i < N; ++i) con.put((int)i, x[i] - origin[0], y[i] - origin[1], z[i] - origin[2]);
    voro::voronoicell_neighbor_3d cell(con);
    // per-atom results first (cells come in container order), then flattened in atom order
    std::vector<std::vector<int>> fv(N);
    std::vector<std::vector<double>> vx(N), fa(N);
    for (int64_t i = 0; i < N; ++i) { nfaces[i] = 0; nverts[i] = 0; volume[i] = 0; radius[i] = 0; }
    for (int ijk = 0; ijk < con.nx * con.ny * con.nz; ++ijk)
        for (int q = 0; q < con.co[ijk]; ++q)
            if (con.compute_cell(cell, ijk, q)) {
                const int i = con.id[ijk][q];
                volume[i] = cell.volume();
                radius[i] = std::sqrt(cell.max_radius_squared());
                cell.face_vertices(fv[i]);
                double *pp = con.p[ijk] + con.ps * q;
                cell.vertices(pp[0], pp[1], pp[2], vx[i]);
                cell.face_areas(fa[i]);
                nfaces[i] = (int)fa[i].size();
                nverts[i] = cell.p;
            }
    int64_t tf = 0, tv = 0, tface = 0;
    for (int64_t i = 0; i < N; ++i) {
        size_t j = 0;
        int f = 0;
        while (j < fv[i].size()) {
            const int m = fv[i][j];
            if (face_sizes) face_sizes[tface] = m;
            for (int k = 0; k < m; ++k) {
                if (face_vertex_ids) face_vertex_ids[tf] = fv[i][j + 1 + k];
                ++tf;
            }
            if (face_area) face_area[tface] = fa[i][f];
            ++tface; ++f;
            j += m + 1;
        }
        for (int k = 0; k < nverts[i] * 3; ++k) {
            if (vertices) vertices[tv * 3 + k] = vx[i][k];
        }
        tv += nverts[i];
    }
    *n_fv = tf;
    *n_vert = tv;
    return (int)tface;
}
}
