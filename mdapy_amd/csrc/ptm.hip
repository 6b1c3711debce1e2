// ptm.hip — polyhedral template matching on gfx950 (SC, FCC, HCP, ICO, BCC, DCUB, DHEX, graphene).
//
// Replaces src/polyhedral_template_matching.cpp:135-318 (get_ptm) together with the part of extern/ptm it drives
// (ptm_preorder_neighbours + ptm_index, see ptm_core.hpp for the per-function map).  The reference runs two
// passes — a serial pre-ordering of every atom's 18 nearest neighbours by Voronoi-face solid angle (:215-255)
// and the OpenMP matching pass (:258-318) — and so do the kernels: k_ptm_order (a thread gathers its row, folds
// the separations with the bit-identical minimum image of common.hpp, orders them, stores 18 bytes) and
// k_ptm_index (templates, (N,8) result row, (N,18) matched-neighbour row).  The split is needed because the
// diamond and graphene templates use the ordered rows of the atom's nearest neighbours as well.
//
// Work per atom is ~1e5 f64 operations on ~4 KB of private state, no HBM traffic to speak of besides the row
// gather (18 x 28 B) and 136 B of output: the kernel is bound by VALU issue / scratch latency, not HBM.  The
// tables (51 KB, generated on the host by ptm_tables.hpp) live in HBM and are read through the scalar/L2 caches.
#include "common.hpp"
#include "ptm_tables.hpp"
#include <mutex>
#include <vector>

namespace mdh {

using ptmc::Tables;


// staged pipeline for the single-shell structure types (ptm_stages.hip)
size_t ptm_match_tables_bytes();
void ptm_compose_match_tables(const ptmc::Tables &T, void *out);
size_t ptm_stage_bytes(int64_t N);
int launch_ptm_order(const double *dx, const double *dy, const double *dz, int64_t N, const DBox &b, const int *dv, int64_t M, int8_t *dord,
                     int *dnbr, unsigned char *redo, int *redo_count, hipStream_t st);
int launch_ptm_stages(const double *dx, const double *dy, const double *dz, int64_t N, const DBox &b, const int *nbr, const int8_t *orders,
                      const int *dtypes, const ptmc::Tables *dt, const void *dmatch, int flags, double rmsd_threshold, double *dout, int ncol,
                      int *dind, int nind, unsigned char *work, hipStream_t st);

static Tables *g_host_tables = nullptr;
const ptmc::Tables *ptm_host_tables() { return g_host_tables; }

// device copy of the tables (+ the composed automorphism table), one per device, created on first use
static int device_tables(const Tables **out, const int8_t **out_autc)
{
    static std::mutex mu;
    static const Tables *dev[64] = {nullptr};
    static const int8_t *dev_autc[64] = {nullptr};
    Tables *&host = g_host_tables;
    int d = 0;
    MDH_HIP(hipGetDevice(&d));
    std::lock_guard<std::mutex> lk(mu);
    if (d < 0 || d >= 64) {
        set_error("mdh_ptm: device ordinal out of range");
        return MDH_ERR_HIP;
    }
    if (!dev[d]) {
        if (!host) {
            Tables *t = new Tables;
            const std::string err = ptmc::tables_generate(*t);
            if (!err.empty()) {
                delete t;
                set_error(err);
                return MDH_ERR_ARG;
            }
            host = t;
        }
        void *p = nullptr;
        MDH_HIP(hipMalloc(&p, sizeof(Tables)));
        MDH_HIP(hipMemcpy(p, host, sizeof(Tables), hipMemcpyHostToDevice));
        dev[d] = static_cast<const Tables *>(p);
        std::vector<int8_t> autc(ptm_match_tables_bytes());
        ptm_compose_match_tables(*host, autc.data());
        MDH_HIP(hipMalloc(&p, autc.size()));
        MDH_HIP(hipMemcpy(p, autc.data(), autc.size(), hipMemcpyHostToDevice));
        dev_autc[d] = static_cast<const int8_t *>(p);
    }
    *out = dev[d];
    *out_autc = dev_autc[d];
    return MDH_OK;
}

// structure string -> PTM_CHECK_* flags, src/polyhedral_template_matching.cpp:168-206
static int parse_structures(const char *s)
{
    static const char *names[] = {"fcc", "hcp", "bcc", "ico", "sc", "dcub", "dhex", "graphene", "all", "default"};
    static const int bits[] = {ptmc::CHECK_FCC, ptmc::CHECK_HCP, ptmc::CHECK_BCC, ptmc::CHECK_ICO, ptmc::CHECK_SC,
                               ptmc::CHECK_DCUB, ptmc::CHECK_DHEX, ptmc::CHECK_GRAPHENE, 255,
                               ptmc::CHECK_FCC | ptmc::CHECK_HCP | ptmc::CHECK_BCC | ptmc::CHECK_ICO};
    auto sep = [](char c) { return c == ' ' || c == ',' || c == '-' || c == '_' || c == '|'; };
    int out = 0;
    while (s && *s) {
        if (sep(*s)) { ++s; continue; }
        bool found = false;
        for (int k = 0; k < 10 && !found; ++k) {
            const size_t len = strlen(names[k]);
            if (strncmp(s, names[k], len) == 0 && (s[len] == 0 || sep(s[len]))) {
                out |= bits[k];
                s += len;
                found = true;
            }
        }
        if (!found)
            ++s;
    }
    return out ? out : (ptmc::CHECK_FCC | ptmc::CHECK_HCP | ptmc::CHECK_BCC | ptmc::CHECK_ICO);
}

} // namespace mdh

using namespace mdh;

namespace mdh { void ptm_debug_order_cap(int cap); }
extern "C" int mdh_debug_set_ptm_order_cap(int cap)
{
    mdh::ptm_debug_order_cap(cap);
    return MDH_OK;
}

extern "C" int mdh_ptm_flags(const char *structure) { return parse_structures(structure); }

extern "C" int mdh_ptm(const char *structure, const double *x, const double *y, const double *z, int64_t N,
                       const double *box9, const double *origin3, const int *boundary3, const int *verlet, int64_t M,
                       const int *types, double rmsd_threshold, double *output, int ncol, int *ptm_indices, int nind,
                       int space, void *stream)
{
    if (N < 0 || M < 0 || ncol < 8 || nind < 0) {
        set_error("mdh_ptm: output needs >= 8 columns and the neighbour list a non-negative width");
        return MDH_ERR_ARG;
    }
    if (N > 2147483647LL) {
        set_error("mdh_ptm: atom ids are 32-bit (verlet_list is int32)");
        return MDH_ERR_ARG;
    }
    const int flags = parse_structures(structure);
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    if (sc.failed())
        return sc.error();
    const Tables *dt = nullptr;
    const int8_t *dautc = nullptr;
    MDH_TRY(device_tables(&dt, &dautc));
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const int *dtp = types ? sc.stage_in(types, (size_t)N, space) : nullptr;
    double *dout = sc.stage(output, (size_t)N * (size_t)ncol, space, false, true);
    int *dind = sc.stage(ptm_indices, (size_t)N * (size_t)nind, space, false, true);
    if (sc.failed())
        return sc.error();
    int8_t *dord = sc.alloc_n<int8_t>((size_t)N * 18);
    int *dnbr = sc.alloc_n<int>((size_t)N * 18);
    unsigned char *dredo = sc.alloc_n<unsigned char>((size_t)N);
    int *dcount = sc.alloc_n<int>(2); // atoms for the second ordering pass; atoms with a face of more than eight vertices
    unsigned char *work = sc.alloc_n<unsigned char>(ptm_stage_bytes(N));
    if (sc.failed())
        return sc.error();
    MDH_TRY(launch_ptm_order(dx, dy, dz, N, b, dv, M, dord, dnbr, dredo, dcount, sc.stream()));
    MDH_TRY(launch_ptm_stages(dx, dy, dz, N, b, dnbr, dord, dtp, dt, dautc, flags, rmsd_threshold, dout, ncol, dind, nind, work, sc.stream()));
    return sc.finish(space);
}

MDH_WARM_UNIT(ptm)
