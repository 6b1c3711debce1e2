// ptm_host.cpp — TEST INFRASTRUCTURE: compiles the product's PTM sources (mdapy_amd/csrc/ptm_core.hpp, ptm_tables.hpp)
// for the host so that the CPU test-suite can check table generation and the per-atom algorithm against oracle/_ref
// without a GPU.  Built by tests/test_ptm_host.py with g++ -ffp-contract=off.  Never loaded by the product.
#include "../../mdapy_amd/csrc/ptm_tables.hpp"
#include <cstdio>

using namespace ptmc;

static Tables g_tables;
static bool g_ready = false;

extern "C" {

const char *ptmh_init()
{
    static std::string err;
    if (!g_ready) {
        err = tables_generate(g_tables);
        g_ready = err.empty();
    }
    return err.c_str();
}

// per type: num_nbrs, num_facets, num_graphs, num_maps, num_conv, total automorphisms
void ptmh_type_info(int type, int *out)
{
    const TypeInfo &t = g_tables.types[type];
    out[0] = t.num_nbrs; out[1] = t.num_facets; out[2] = t.num_graphs; out[3] = t.num_maps; out[4] = t.num_conv;
    int na = 0;
    for (int g = t.graph_begin; g < t.graph_begin + t.num_graphs; ++g) na += g_tables.graphs[g].num_aut;
    out[5] = na;
}
void ptmh_graph_hashes(int type, uint64_t *out, int *naut)
{
    const TypeInfo &t = g_tables.types[type];
    for (int g = 0; g < t.num_graphs; ++g) { out[g] = g_tables.graphs[t.graph_begin + g].hash; naut[g] = g_tables.graphs[t.graph_begin + g].num_aut; }
}
void ptmh_generators(int type, double *out)
{
    const TypeInfo &t = g_tables.types[type];
    for (int i = 0; i < t.num_conv; ++i)
        for (int k = 0; k < 4; ++k) out[i * 4 + k] = g_tables.gens[t.gen_begin + i][k];
}
void ptmh_mappings(int type, int conventional, int8_t *out)
{
    const TypeInfo &t = g_tables.types[type];
    const int b = conventional ? t.conv_begin : t.map_begin, n = conventional ? t.num_conv : t.num_maps;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < MAX_PTS; ++k) out[i * MAX_PTS + k] = g_tables.maps[b + i][k];
}
void ptmh_template(int type, double *out)
{
    std::memcpy(out, g_tables.types[type].points, sizeof(double) * MAX_PTS * 3);
}

// both passes on the host.  box9 row-major cell vectors; only orthogonal/triclinic minimum image as box.h does.
namespace {
struct HostFold {
    bool tri;
    const double *h;
    double hi[9];
    int pbc[3];
    void operator()(double &dx, double &dy, double &dz) const
    {
        if (tri) {
            double fx = dx * hi[0] + dy * hi[3] + dz * hi[6], fy = dx * hi[1] + dy * hi[4] + dz * hi[7], fz = dx * hi[2] + dy * hi[5] + dz * hi[8];
            if (pbc[0]) fx -= std::floor(fx + 0.5);
            if (pbc[1]) fy -= std::floor(fy + 0.5);
            if (pbc[2]) fz -= std::floor(fz + 0.5);
            dx = fx * h[0] + fy * h[3] + fz * h[6];
            dy = fx * h[1] + fy * h[4] + fz * h[7];
            dz = fx * h[2] + fy * h[5] + fz * h[8];
        } else {
            if (pbc[0]) dx -= h[0] * std::floor(dx / h[0] + 0.5);
            if (pbc[1]) dy -= h[4] * std::floor(dy / h[4] + 0.5);
            if (pbc[2]) dz -= h[8] * std::floor(dz / h[8] + 0.5);
        }
    }
};
struct HostSrc {
    const double *x, *y, *z;
    int64_t N, M;
    const int *verlet, *types;
    const HostFold *fold;
    const int8_t *orders; // (N,18) from pass 1
    void get(int atom, Env &env) { PolyLocal poly; build_env(x, y, z, N, verlet + (int64_t)atom * M, (int)M, types, atom, *fold, orders + (int64_t)atom * 18, env, poly); }
};
} // namespace

int ptmh_run(const double *x, const double *y, const double *z, int64_t N, const double *box9, const int *boundary, const int *verlet,
             int64_t M, const int *types, int flags, double rmsd_threshold, double *output, int *ptm_indices, int8_t *order_out)
{
    if (!g_ready) return -1;
    HostFold fold;
    fold.tri = false;
    fold.h = box9;
    for (int i = 0; i < 3; ++i) {
        fold.pbc[i] = boundary[i];
        for (int j = 0; j < 3; ++j)
            if (i != j && std::fabs(box9[i * 3 + j]) > 1e-10) fold.tri = true;
    }
    std::memset(fold.hi, 0, sizeof(fold.hi));
    if (fold.tri) {
        const double *m = box9;
        double *hi = fold.hi;
        const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        const double id = 1.0 / det;
        hi[0] = (m[4] * m[8] - m[5] * m[7]) * id;  hi[1] = -(m[1] * m[8] - m[2] * m[7]) * id; hi[2] = (m[1] * m[5] - m[2] * m[4]) * id;
        hi[3] = -(m[3] * m[8] - m[5] * m[6]) * id; hi[4] = (m[0] * m[8] - m[2] * m[6]) * id;  hi[5] = -(m[0] * m[5] - m[2] * m[3]) * id;
        hi[6] = (m[3] * m[7] - m[4] * m[6]) * id;  hi[7] = -(m[0] * m[7] - m[1] * m[6]) * id; hi[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    }
    std::vector<int8_t> orders((size_t)N * 18, (int8_t)-1);
    for (int64_t i = 0; i < N; ++i) { // pass 1: the Voronoi order of every atom's row
        Env env;
        int8_t *orow = orders.data() + i * 18;
        PolyLocal poly;
        build_env(x, y, z, N, verlet + i * M, (int)M, types, (int)i, fold, nullptr, env, poly);
        for (int k = 1; k < env.num; ++k) orow[k - 1] = (int8_t)(env.corr[k] - 1);
    }
    if (order_out) std::memcpy(order_out, orders.data(), orders.size());
    HostSrc src{x, y, z, N, M, verlet, types, &fold, orders.data()};
    for (int64_t i = 0; i < N; ++i) {
        Result r;
        Canon C;
        index_atom<true>(g_tables, flags, src, (int)i, r, C);
        double *o = output + i * 8;
        int type = r.type, ordering = r.ordering;
        if (r.rmsd > rmsd_threshold || type == T_NONE) { type = 0; ordering = 0; }
        o[0] = type; o[1] = ordering; o[2] = r.rmsd; o[3] = r.interatomic;
        o[4] = r.q[0]; o[5] = r.q[1]; o[6] = r.q[2]; o[7] = r.q[3];
        for (int k = 0; k < 18; ++k) ptm_indices[i * 18 + k] = k < r.num_out ? r.ids[k] : -1;
    }
    return 0;
}
}
