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
int ptmh_run(const double *x, const double *y, const double *z, int64_t N, const double *box9, const int *boundary, const int *verlet,
             int64_t M, const int *types, int flags, double rmsd_threshold, double *output, int *ptm_indices, int8_t *order_out)
{
    if (!g_ready) return -1;
    bool tri = false;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            if (i != j && std::fabs(box9[i * 3 + j]) > 1e-10) tri = true;
    double hi[9] = {0};
    if (tri) {
        const double *m = box9;
        const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        const double id = 1.0 / det;
        hi[0] = (m[4] * m[8] - m[5] * m[7]) * id;  hi[1] = -(m[1] * m[8] - m[2] * m[7]) * id; hi[2] = (m[1] * m[5] - m[2] * m[4]) * id;
        hi[3] = -(m[3] * m[8] - m[5] * m[6]) * id; hi[4] = (m[0] * m[8] - m[2] * m[6]) * id;  hi[5] = -(m[0] * m[5] - m[2] * m[3]) * id;
        hi[6] = (m[3] * m[7] - m[4] * m[6]) * id;  hi[7] = -(m[0] * m[7] - m[1] * m[6]) * id; hi[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    }
    for (int64_t i = 0; i < N; ++i) {
        const int *row = verlet + i * M;
        double pts[MAX_IN][3];
        int ids[MAX_IN];
        int cnt = 0;
        for (int j = 0; j < (int)M && j < 18; ++j) {
            const int k = row[j];
            if (k < 0 || k >= N) break;
            if (k == i) continue;
            double dx = x[k] - x[i], dy = y[k] - y[i], dz = z[k] - z[i];
            if (tri) {
                double fx = dx * hi[0] + dy * hi[3] + dz * hi[6], fy = dx * hi[1] + dy * hi[4] + dz * hi[7], fz = dx * hi[2] + dy * hi[5] + dz * hi[8];
                if (boundary[0]) fx -= std::floor(fx + 0.5);
                if (boundary[1]) fy -= std::floor(fy + 0.5);
                if (boundary[2]) fz -= std::floor(fz + 0.5);
                dx = fx * box9[0] + fy * box9[3] + fz * box9[6];
                dy = fx * box9[1] + fy * box9[4] + fz * box9[7];
                dz = fx * box9[2] + fy * box9[5] + fz * box9[8];
            } else {
                if (boundary[0]) dx -= box9[0] * std::floor(dx / box9[0] + 0.5);
                if (boundary[1]) dy -= box9[4] * std::floor(dy / box9[4] + 0.5);
                if (boundary[2]) dz -= box9[8] * std::floor(dz / box9[8] + 0.5);
            }
            pts[cnt][0] = dx; pts[cnt][1] = dy; pts[cnt][2] = dz;
            ids[cnt++] = k;
        }
        int8_t order[MAX_IN];
        order_neighbours(cnt, pts, order);
        if (order_out)
            for (int k = 0; k < 18; ++k) order_out[i * 18 + k] = k < cnt ? order[k] : (int8_t)-1;
        double env[MAX_IN][3];
        int numbers[MAX_IN], atom_ids[MAX_IN];
        env[0][0] = env[0][1] = env[0][2] = 0;
        numbers[0] = types ? types[i] : 0;
        atom_ids[0] = (int)i;
        for (int k = 0; k < cnt; ++k) {
            const int p = order[k];
            env[k + 1][0] = pts[p][0]; env[k + 1][1] = pts[p][1]; env[k + 1][2] = pts[p][2];
            numbers[k + 1] = types ? types[ids[p]] : 0;
            atom_ids[k + 1] = ids[p];
        }
        Result r;
        index_atom(g_tables, flags, cnt + 1, env, numbers, r);
        double *o = output + i * 8;
        int type = r.type, ordering = r.ordering;
        if (r.rmsd > rmsd_threshold || type == T_NONE) { type = 0; ordering = 0; }
        o[0] = type; o[1] = ordering; o[2] = r.rmsd; o[3] = r.interatomic;
        o[4] = r.q[0]; o[5] = r.q[1]; o[6] = r.q[2]; o[7] = r.q[3];
        for (int k = 0; k < 18; ++k) ptm_indices[i * 18 + k] = k < r.num_out ? atom_ids[r.mapping[k]] : -1;
    }
    return 0;
}
}
