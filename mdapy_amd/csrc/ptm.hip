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

static constexpr int PTM_BLOCK = 64;

// working set of the canonical form in LDS (ptm_core.hpp: Canon is the private-array twin): per lane 256 + 84 + 16 bytes
// and 16 half-words, element e of lane l at [e * PTM_BLOCK + l]
struct CanonLds {
    static constexpr int BYTES = 256 + 2 * ptmc::MAX_EDGES + ptmc::MAX_NBR; // common | best | index
    static constexpr size_t LDS_BYTES = (size_t)(BYTES + 2 * ptmc::MAX_NBR) * PTM_BLOCK;
    int8_t *b8;        // byte elements, stride PTM_BLOCK
    uint16_t *h16;     // walked[], stride PTM_BLOCK
    int8_t label[ptmc::MAX_PTS];
    __device__ __forceinline__ int cm(int a, int b) const { return b8[(a * 16 + b) * PTM_BLOCK]; }
    __device__ __forceinline__ void cm_set(int a, int b, int v) { b8[(a * 16 + b) * PTM_BLOCK] = (int8_t)v; }
    __device__ __forceinline__ int bs(int i) const { return b8[(256 + i) * PTM_BLOCK]; }
    __device__ __forceinline__ void bs_set(int i, int v) { b8[(256 + i) * PTM_BLOCK] = (int8_t)v; }
    __device__ __forceinline__ int ix(int i) const { return b8[(256 + 2 * ptmc::MAX_EDGES + i) * PTM_BLOCK]; }
    __device__ __forceinline__ void ix_set(int i, int v) { b8[(256 + 2 * ptmc::MAX_EDGES + i) * PTM_BLOCK] = (int8_t)v; }
    __device__ __forceinline__ unsigned wk(int i) const { return h16[i * PTM_BLOCK]; }
    __device__ __forceinline__ void wk_set(int i, unsigned v) { h16[i * PTM_BLOCK] = (uint16_t)v; }
};

#ifdef MDH_PTM_CANON_LDS
static constexpr size_t PTM_INDEX_LDS = CanonLds::LDS_BYTES;
#else
static constexpr size_t PTM_INDEX_LDS = 0;
#endif

template <bool TRI> struct DevFold {
    const DBox &b;
    __device__ __forceinline__ void operator()(double &dx, double &dy, double &dz) const { pbc<TRI>(b, dx, dy, dz); }
};

template <bool TRI> struct DevSrc {
    const double *x, *y, *z;
    int64_t N, M;
    const int *verlet, *types;
    const int8_t *orders;
    const DBox &b;
    __device__ void get(int atom, ptmc::Env &env)
    {
        const DevFold<TRI> fold{b};
        ptmc::PolyLocal unused; // the order is given: no polygon work
        ptmc::build_env(x, y, z, N, verlet + (int64_t)atom * M, (int)M, types, atom, fold, orders + (int64_t)atom * 18, env, unused);
    }
};

// pass 2 (:258-318): match the templates, write the result row and the matched-neighbour row.  SHELL = diamond /
// graphene stages compiled in (9 KB more private memory per lane; the common fcc-hcp-bcc call does without)
template <bool TRI, bool SHELL>
__global__ __launch_bounds__(PTM_BLOCK) void k_ptm_index(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ z, int64_t N, DBox b,
                                                         const int *__restrict__ verlet, int64_t M,
                                                         const int *__restrict__ types, const int8_t *__restrict__ orders,
                                                         const Tables *__restrict__ tables, int flags, double rmsd_threshold,
                                                         double *__restrict__ output, int ncol, int *__restrict__ ptm_indices,
                                                         int nind)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    DevSrc<TRI> src{x, y, z, N, M, verlet, types, orders, b};
    ptmc::Result r;
#ifdef MDH_PTM_CANON_LDS
    extern __shared__ unsigned char canon_lds[];
    CanonLds C;
    C.b8 = reinterpret_cast<int8_t *>(canon_lds) + threadIdx.x;
    C.h16 = reinterpret_cast<uint16_t *>(canon_lds + (size_t)CanonLds::BYTES * PTM_BLOCK) + threadIdx.x;
#else
    // Measured (1 M rattled fcc atoms): with the canonical-form arrays in LDS the kernel keeps 6 instead of 16 waves per
    // CU and the hull phase, still in private memory, loses more than the canonical form gains (76 ms vs 63 ms).
    ptmc::Canon C;
#endif
    ptmc::index_atom<SHELL>(*tables, flags, src, (int)i, r, C);
    int type = r.type, ordering = r.ordering;
    if (r.rmsd > rmsd_threshold || type == ptmc::T_NONE) { // :287-291
        type = 0;
        ordering = 0;
    }
    double *o = output + i * ncol;
    const double vals[8] = {(double)type, (double)ordering, r.rmsd, r.interatomic, r.q[0], r.q[1], r.q[2], r.q[3]};
    for (int k = 0; k < ncol; ++k)
        o[k] = k < 8 ? vals[k] : 0.0;
    int *pi = ptm_indices + (int64_t)i * nind;
    for (int k = 0; k < nind; ++k)
        pi[k] = k < r.num_out && k < ptmc::MAX_PTS ? r.ids[k] : -1;
}

// staged pipeline for the single-shell structure types (ptm_stages.hip)
void ptm_compose_automorphisms(const ptmc::Tables &T, int8_t *autc);
size_t ptm_stage_bytes(int64_t N);
int launch_ptm_order(const double *dx, const double *dy, const double *dz, int64_t N, const DBox &b, const int *dv, int64_t M, int8_t *dord,
                     int *dnbr, unsigned char *redo, int *redo_count, hipStream_t st);
int launch_ptm_stages(const double *dx, const double *dy, const double *dz, int64_t N, const DBox &b, const int *nbr, const int *dtypes,
                      const ptmc::Tables *dt, const int8_t *dautc, int flags, double rmsd_threshold, double *dout, int ncol, int *dind,
                      int nind, unsigned char *work, hipStream_t st);

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
        std::vector<int8_t> autc((size_t)ptmc::MAX_AUTS * ptmc::MAX_PTS);
        ptm_compose_automorphisms(*host, autc.data());
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
    int *dcount = sc.alloc_n<int>(1);
    const bool shell = (flags & (ptmc::CHECK_DCUB | ptmc::CHECK_DHEX | ptmc::CHECK_GRAPHENE)) != 0;
    unsigned char *work = shell ? nullptr : sc.alloc_n<unsigned char>(ptm_stage_bytes(N));
    if (sc.failed())
        return sc.error();
    const dim3 grid(grid_for(N, PTM_BLOCK)), block(PTM_BLOCK);
    MDH_TRY(launch_ptm_order(dx, dy, dz, N, b, dv, M, dord, dnbr, dredo, dcount, sc.stream()));
    if (!shell) {
        MDH_TRY(launch_ptm_stages(dx, dy, dz, N, b, dnbr, dtp, dt, dautc, flags, rmsd_threshold, dout, ncol, dind, nind, work, sc.stream()));
    } else {
        ProfRange pr("k_ptm_index", sc.stream());
#define MDH_PTM_LAUNCH(TRI, SHELL)                                                                                              \
    hipLaunchKernelGGL((k_ptm_index<TRI, SHELL>), grid, block, PTM_INDEX_LDS, sc.stream(), dx, dy, dz, N, b, dv, M, dtp, dord, dt, flags,   \
                       rmsd_threshold, dout, ncol, dind, nind)
        if (b.tri && shell) MDH_PTM_LAUNCH(true, true);
        else if (b.tri) MDH_PTM_LAUNCH(true, false);
        else if (shell) MDH_PTM_LAUNCH(false, true);
        else MDH_PTM_LAUNCH(false, false);
#undef MDH_PTM_LAUNCH
    }
    return sc.finish(space);
}
