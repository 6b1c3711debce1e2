// ptm_ref_driver.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref).
//
// C-ABI driver around the REFERENCE's own PTM library (extern/ptm, plain C++, MIT, P. M. Larsen), compiled from
// the sources where they lie under /root/reference by oracle/Makefile.ref into oracle/_ref/libptm_ref.so.
// The reference's driver (src/polyhedral_template_matching.cpp) includes nanobind and cannot be compiled in this
// image, so its logic is restated here, function by function:
//   get_neighbours callback        src/polyhedral_template_matching.cpp:32-133
//   structure-string parsing       :168-206
//   pass 1 (neighbour pre-ordering) :215-255   -> ptm_preorder_neighbours (extern/ptm/ptm_neighbour_ordering.cpp:174)
//   pass 2 (ptm_index + outputs)    :258-318   -> ptm_index (extern/ptm/ptm_index.cpp:114)
#include <ptm_constants.h>
#include <ptm_functions.h>
#include <ptm_initialize_data.h>
#include <ptm_quat.h>
#include <ptm_solid_angles.h>
#include <ptm_voronoi_cell.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct RBox { // src/box.h:8-126 (orthogonal + triclinic minimum image)
    double h[9], hi[9];
    int pbc[3];
    bool tri;
    void fold(double &x, double &y, double &z) const
    {
        if (tri) {
            double fx = x * hi[0] + y * hi[3] + z * hi[6], fy = x * hi[1] + y * hi[4] + z * hi[7], fz = x * hi[2] + y * hi[5] + z * hi[8];
            if (pbc[0]) fx -= std::floor(fx + 0.5);
            if (pbc[1]) fy -= std::floor(fy + 0.5);
            if (pbc[2]) fz -= std::floor(fz + 0.5);
            x = fx * h[0] + fy * h[3] + fz * h[6];
            y = fx * h[1] + fy * h[4] + fz * h[7];
            z = fx * h[2] + fy * h[5] + fz * h[8];
        } else {
            if (pbc[0]) x -= h[0] * std::floor(x / h[0] + 0.5);
            if (pbc[1]) y -= h[4] * std::floor(y / h[4] + 0.5);
            if (pbc[2]) z -= h[8] * std::floor(z / h[8] + 0.5);
        }
    }
};

bool make_box(RBox &b, const double *box9, const int *boundary) // box.h:182-244
{
    b.tri = false;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            b.h[i * 3 + j] = box9[i * 3 + j];
            if (i != j && std::fabs(box9[i * 3 + j]) > 1e-10) b.tri = true;
        }
    if (b.h[0] < 0 || b.h[4] < 0 || b.h[8] < 0) b.tri = true;
    std::memset(b.hi, 0, sizeof(b.hi));
    if (b.tri) {
        const double *m = b.h;
        const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        if (std::fabs(det) < 1e-12) return false;
        const double id = 1.0 / det;
        b.hi[0] = (m[4] * m[8] - m[5] * m[7]) * id;  b.hi[1] = -(m[1] * m[8] - m[2] * m[7]) * id; b.hi[2] = (m[1] * m[5] - m[2] * m[4]) * id;
        b.hi[3] = -(m[3] * m[8] - m[5] * m[6]) * id; b.hi[4] = (m[0] * m[8] - m[2] * m[6]) * id;  b.hi[5] = -(m[0] * m[5] - m[2] * m[3]) * id;
        b.hi[6] = (m[3] * m[7] - m[4] * m[6]) * id;  b.hi[7] = -(m[0] * m[7] - m[1] * m[6]) * id; b.hi[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    }
    for (int i = 0; i < 3; ++i) b.pbc[i] = boundary[i];
    return true;
}

struct NbrData {
    const double *x, *y, *z;
    const int *verlet;
    int stride, n;
    RBox box;
    const int *types;
    const uint64_t *cached;
};

int get_neighbours(void *vdata, size_t, size_t atom, int num_requested, ptm_atomicenv_t *env) // :32-133
{
    NbrData *d = static_cast<NbrData *>(vdata);
    if (atom >= (size_t)d->n) return -1;
    const int *row = d->verlet + atom * (size_t)d->stride;
    int max_nbrs = std::min(num_requested - 1, PTM_MAX_INPUT_POINTS - 1);
    max_nbrs = std::min(max_nbrs, d->stride);
    int cnt = 0, ids[PTM_MAX_INPUT_POINTS - 1];
    double pts[PTM_MAX_INPUT_POINTS - 1][3];
    for (int i = 0; i < max_nbrs; ++i) {
        const int j = row[i];
        if (j < 0 || j >= d->n) break;
        if (j == (int)atom) continue;
        double dx = d->x[j] - d->x[atom], dy = d->y[j] - d->y[atom], dz = d->z[j] - d->z[atom];
        d->box.fold(dx, dy, dz);
        pts[cnt][0] = dx; pts[cnt][1] = dy; pts[cnt][2] = dz;
        ids[cnt++] = j;
    }
    int dummy = 0;
    ptm_decode_correspondences(PTM_MATCH_FCC, d->cached[atom], env->correspondences, &dummy);
    env->atom_indices[0] = atom;
    env->points[0][0] = env->points[0][1] = env->points[0][2] = 0;
    for (int i = 0; i < cnt; ++i) {
        const int p = env->correspondences[i + 1] - 1;
        if (p >= 0 && p < cnt) {
            env->atom_indices[i + 1] = ids[p];
            std::memcpy(env->points[i + 1], pts[p], 3 * sizeof(double));
        }
    }
    if (d->types) {
        env->numbers[0] = d->types[atom];
        for (int i = 0; i < cnt; ++i) {
            const int p = env->correspondences[i + 1] - 1;
            if (p >= 0 && p < cnt) env->numbers[i + 1] = d->types[ids[p]];
        }
    } else {
        for (int i = 0; i < cnt + 1; ++i) env->numbers[i] = 0;
    }
    env->num = cnt + 1;
    return cnt + 1;
}

int parse_flags(const char *s) // :168-206
{
    static const char *names[] = {"fcc", "hcp", "bcc", "ico", "sc", "dcub", "dhex", "graphene", "all", "default"};
    static const int32_t flags[] = {PTM_CHECK_FCC, PTM_CHECK_HCP, PTM_CHECK_BCC, PTM_CHECK_ICO, PTM_CHECK_SC, PTM_CHECK_DCUB, PTM_CHECK_DHEX,
                                    PTM_CHECK_GRAPHENE, PTM_CHECK_ALL, PTM_CHECK_FCC | PTM_CHECK_HCP | PTM_CHECK_BCC | PTM_CHECK_ICO};
    auto sep = [](char c) { return c == ' ' || c == ',' || c == '-' || c == '_' || c == '|'; };
    int out = 0;
    while (*s) {
        if (sep(*s)) { ++s; continue; }
        bool found = false;
        for (int i = 0; i < 10; ++i) {
            const size_t len = std::strlen(names[i]);
            if (std::strncmp(s, names[i], len) == 0 && (s[len] == 0 || sep(s[len]))) { out |= flags[i]; s += len; found = true; break; }
        }
        if (!found) ++s;
    }
    return out ? out : (PTM_CHECK_FCC | PTM_CHECK_HCP | PTM_CHECK_BCC | PTM_CHECK_ICO);
}

} // namespace

extern "C" {

__attribute__((visibility("default"))) int ref_ptm_parse_flags(const char *structure) { return parse_flags(structure); }

// output (N, ncol_out) f64, ptm_indices (N, nind) i32; types may be NULL.  Also returns the pass-1 permutation codes.
__attribute__((visibility("default"))) int ref_get_ptm(const char *structure, const double *x, const double *y, const double *z, int64_t N,
                                                       const double *box9, const double *origin, const int *boundary, const int *verlet,
                                                       int64_t M, const int *types, double rmsd_threshold, double *output, int ncol_out,
                                                       int *ptm_indices, int nind, uint64_t *cached_out)
{
    (void)origin;
    NbrData d;
    d.x = x; d.y = y; d.z = z; d.verlet = verlet; d.stride = (int)M; d.n = (int)N; d.types = types;
    if (!make_box(d.box, box9, boundary)) return -1;
    const int flags = parse_flags(structure);
    ptm_initialize_global();
    std::vector<uint64_t> cached((size_t)N);
    d.cached = cached.data();
    ptm_local_handle_t lh = ptm_initialize_local();
    for (int64_t i = 0; i < N; ++i) { // pass 1, serial in the reference (:215-255)
        const int *row = verlet + i * M;
        int cnt = 0;
        double pts[PTM_MAX_INPUT_POINTS - 1][3];
        for (int j = 0; j < (int)M; ++j) {
            const int k = row[j];
            if (k < 0 || k >= N) break;
            if (k == i) continue;
            double dx = x[k] - x[i], dy = y[k] - y[i], dz = z[k] - z[i];
            d.box.fold(dx, dy, dz);
            pts[cnt][0] = dx; pts[cnt][1] = dy; pts[cnt][2] = dz;
            if (++cnt >= PTM_MAX_INPUT_POINTS - 1) break;
        }
        ptm_preorder_neighbours(lh, cnt, pts, &cached[(size_t)i]);
    }
    for (int64_t i = 0; i < N; ++i) { // pass 2 (:258-318)
        for (int k = 0; k < ncol_out; ++k) output[i * ncol_out + k] = 0.0;
        ptm_result_t res;
        std::memset(&res, 0, sizeof(res));
        ptm_atomicenv_t oenv;
        std::memset(&oenv, 0, sizeof(oenv));
        const int ret = ptm_index(lh, (size_t)i, get_neighbours, &d, flags, false, &res, &oenv);
        if (ret != 0 || res.rmsd > rmsd_threshold || res.structure_type == PTM_MATCH_NONE) {
            res.structure_type = PTM_MATCH_NONE;
            res.ordering_type = PTM_ALLOY_NONE;
        }
        const int nsave = std::min(oenv.num, nind);
        for (int k = 0; k < nsave; ++k) ptm_indices[i * nind + k] = (int)oenv.atom_indices[k];
        for (int k = nsave; k < nind; ++k) ptm_indices[i * nind + k] = -1;
        const double vals[8] = {(double)res.structure_type, (double)res.ordering_type, res.rmsd, res.interatomic_distance,
                                res.orientation[0], res.orientation[1], res.orientation[2], res.orientation[3]};
        for (int k = 0; k < ncol_out && k < 8; ++k) output[i * ncol_out + k] = vals[k];
    }
    ptm_uninitialize_local(lh);
    if (cached_out) std::memcpy(cached_out, cached.data(), sizeof(uint64_t) * (size_t)N);
    return 0;
}

// the pass-1 neighbour order of one atom as a permutation of its row (0-based; debugging aid of the host tests)
__attribute__((visibility("default"))) void ref_ptm_decode_order(uint64_t code, int8_t *order19)
{
    int8_t corr[PTM_MAX_INPUT_POINTS];
    int dummy = 0;
    ptm_decode_correspondences(PTM_MATCH_FCC, code, corr, &dummy);
    for (int i = 0; i < 19; ++i) order19[i] = corr[i];
}

// solid angles of the Voronoi faces of a point set exactly as extern/ptm/ptm_neighbour_ordering.cpp:57-113 obtains them
// from the library's own cell class (debugging aid of the host tests); nfv receives the number of vertices per face
__attribute__((visibility("default"))) int ref_ptm_solid_angles(int num, const double *pts, double *areas, int *nfv)
{
    ptm_voro::voronoicell_neighbor v;
    double max_norm = 0;
    std::vector<double> nsq(num);
    for (int i = 0; i < num; ++i) {
        nsq[i] = pts[i * 3] * pts[i * 3] + pts[i * 3 + 1] * pts[i * 3 + 1] + pts[i * 3 + 2] * pts[i * 3 + 2];
        max_norm = std::max(max_norm, nsq[i]);
        areas[i] = 0;
        nfv[i] = 0;
    }
    const double k = 10 * std::sqrt(max_norm);
    v.init(-k, k, -k, k, -k, k);
    for (int i = 0; i < num; ++i) v.nplane(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], nsq[i], i);
    std::vector<int> nbr, fv;
    std::vector<double> vert;
    v.neighbors(nbr);
    v.face_vertices(fv);
    v.vertices(0, 0, 0, vert);
    for (size_t i = 0; i < vert.size() / 3; ++i) {
        const double n = std::sqrt(vert[i * 3] * vert[i * 3] + vert[i * 3 + 1] * vert[i * 3 + 1] + vert[i * 3 + 2] * vert[i * 3 + 2]);
        vert[i * 3] /= n; vert[i * 3 + 1] /= n; vert[i * 3 + 2] /= n;
    }
    size_t c = 0;
    for (int f = 0; f < v.number_of_faces(); ++f) {
        const int nv = fv[c++];
        if (nbr[f] >= 0) {
            double sa = 0;
            int u = fv[c], w1 = fv[c + 1];
            for (int i = 2; i < nv; ++i) {
                const int w = fv[c + i];
                sa += ptm::calculate_solid_angle(&vert[u * 3], &vert[w1 * 3], &vert[w * 3]);
                w1 = w;
            }
            areas[nbr[f]] = sa;
            nfv[nbr[f]] = nv;
        }
        c += nv;
    }
    return v.number_of_faces();
}

// ---- read-only views of the reference's literal tables (used by tests to validate GENERATED tables) ----------
static const ptm::refdata_t *ref_struct(int type)
{
    switch (type) {
    case PTM_MATCH_FCC: return &ptm::structure_fcc;
    case PTM_MATCH_HCP: return &ptm::structure_hcp;
    case PTM_MATCH_BCC: return &ptm::structure_bcc;
    case PTM_MATCH_ICO: return &ptm::structure_ico;
    case PTM_MATCH_SC: return &ptm::structure_sc;
    case PTM_MATCH_DCUB: return &ptm::structure_dcub;
    case PTM_MATCH_DHEX: return &ptm::structure_dhex;
    case PTM_MATCH_GRAPHENE: return &ptm::structure_graphene;
    }
    return nullptr;
}
// info[0..5] = num_nbrs, num_facets, max_degree, num_graphs, num_mappings, num_conventional_mappings
__attribute__((visibility("default"))) int ref_ptm_struct_info(int type, int *info)
{
    ptm_initialize_global();
    const ptm::refdata_t *s = ref_struct(type);
    if (!s) return -1;
    info[0] = s->num_nbrs; info[1] = s->num_facets; info[2] = s->max_degree; info[3] = s->num_graphs; info[4] = s->num_mappings;
    info[5] = s->num_conventional_mappings;
    return 0;
}
__attribute__((visibility("default"))) int ref_ptm_graphs(int type, uint64_t *hashes, int *nauts, int8_t *canon, int8_t *facets)
{
    ptm_initialize_global();
    const ptm::refdata_t *s = ref_struct(type);
    if (!s) return -1;
    for (int g = 0; g < s->num_graphs; ++g) {
        hashes[g] = s->graphs[g].hash;
        nauts[g] = s->graphs[g].num_automorphisms;
        std::memcpy(canon + g * PTM_MAX_POINTS, s->graphs[g].canonical_labelling, PTM_MAX_POINTS);
        std::memcpy(facets + g * PTM_MAX_FACETS * 3, s->graphs[g].facets, PTM_MAX_FACETS * 3);
    }
    return 0;
}
// which: 0 = mapping (num_mappings rows), 1 = mapping_conventional; quats: qconventional rows (may be NULL)
__attribute__((visibility("default"))) int ref_ptm_symmetry(int type, int which, int8_t *maps, double *quats, double *points)
{
    const ptm::refdata_t *s = ref_struct(type);
    if (!s) return -1;
    const int n = (which && s->num_conventional_mappings) ? s->num_conventional_mappings : s->num_mappings;
    const int8_t (*m)[PTM_MAX_POINTS] = which ? s->mapping_conventional : s->mapping;
    for (int i = 0; i < n; ++i) std::memcpy(maps + i * PTM_MAX_POINTS, m[i], PTM_MAX_POINTS);
    if (quats && which) { // the generator list map_quaternion_onto_target pairs with mapping_conventional (ptm_map_templates.cpp:21-68)
        const double (*g)[4] = s->qconventional;
        if (type == PTM_MATCH_SC || type == PTM_MATCH_FCC || type == PTM_MATCH_BCC || type == PTM_MATCH_DCUB) g = ptm::generator_cubic;
        if (type == PTM_MATCH_ICO) g = ptm::generator_icosahedral;
        if (g) std::memcpy(quats, g, sizeof(double) * 4 * n);
    }
    if (points) std::memcpy(points, s->points[0], sizeof(double) * 3 * PTM_MAX_POINTS);
    return n;
}
}
