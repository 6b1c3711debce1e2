// cna.hip — common neighbour analysis on gfx950 (fixed cutoff, adaptive, diamond).
//
// Replaces src/cna.cpp of the reference: FixedCNA :429-506, AdaptiveCNA
// :289-427, IdentifyDiamond :163-287, over the helpers :16-161.
//
// One thread per atom.  The <=14 listed neighbours are gathered into registers
// (fully unrolled, no scratch), the nn x nn bond matrix is kept as 16-bit rows
// packed four to a 64-bit register so that rows can be fetched by a dynamic
// index without indexing a register array, and the three numbers of every CNA
// signature (#common neighbours, #bonds among them, #bonds of the largest
// connected bond cluster) are obtained with popcount / ctz bit walks.
//
// Minimum image in the unrolled pair loop (orthogonal boxes): when the listed
// positions span less than ~1.5 L on every periodic axis, floor(d/L+0.5) of any
// difference is one of {-1,0,1} and is decided by the two exact thresholds of
// DBox::tn, so the shift L*n is picked by two compares (L*1 = L, L*0 = 0,
// L*-1 = -L are exact) — no division in the hot code.  Atoms that fail the span
// test (unwrapped input) are appended to a to-do list and finished by a second,
// compact kernel that evaluates the reference expression as is (GENERIC = true).
#include "common.hpp"

namespace mdh {

struct Rows { // bond matrix: row a = bits of the neighbours bonded to neighbour a (a < 16)
    uint64_t w0, w1, w2, w3;
    __device__ __forceinline__ unsigned row(int a) const
    {
        const uint64_t w = (a < 8) ? ((a < 4) ? w0 : w1) : ((a < 12) ? w2 : w3);
        return (unsigned)(w >> ((a & 3) << 4)) & 0xffffu;
    }
};

template <int NN>
__device__ __forceinline__ Rows pack_rows(const unsigned (&adj)[NN])
{
    Rows r{0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < NN; ++a) {
        const uint64_t v = (uint64_t)adj[a] << ((a & 3) << 4);
        if (a < 4) r.w0 |= v;
        else if (a < 8) r.w1 |= v;
        else if (a < 12) r.w2 |= v;
        else r.w3 |= v;
    }
    return r;
}

// (ncn, nb, chain) of the bond centre--neighbour ni.  Only neighbours in `limit_mask`
// take part in the bond search (cna.cpp:69-92; the adaptive 12-neighbour pass hands 12, :344).
__device__ __forceinline__ void signature(const Rows &R, int ni, unsigned limit_mask, int &ncn, int &nb, int &chain)
{
    const unsigned common = R.row(ni); // cna.cpp:52-64
    ncn = __popc(common);
    const unsigned pool = common & limit_mask;
    int bonds = 0, maxdeg = 0;
    for (unsigned m = pool; m; m &= m - 1) {
        const int deg = __popc(R.row(__ffs(m) - 1) & pool);
        bonds += deg;
        maxdeg = deg > maxdeg ? deg : maxdeg;
    }
    nb = bonds >> 1;
    // `chain` = number of bonds in the largest connected bond cluster (cna.cpp:97-147).  Shortcuts that are
    // exact consequences of that definition:
    //   nb <= 1           -> chain = nb
    //   nb == 2           -> 2 if the two bonds share an atom (some degree is 2), else 1
    //   k atoms, k bonds, k in {4,5} and every atom of the pool counted (ncn == popc(pool)):
    //                        a graph on k <= 5 vertices with k edges cannot be split (3+1 vertices hold <= 3 edges,
    //                        3+2 hold <= 4, 4+1 hold all of them in the connected 4-part) -> chain = k
    // everything else (e.g. 6 atoms / 6 bonds: two triangles give 3) walks the clusters.
    if (nb <= 1) { chain = nb; return; }
    if (nb == 2) { chain = maxdeg == 2 ? 2 : 1; return; }
    const int npool = __popc(pool);
    if (nb == npool && (npool == 4 || npool == 5)) { chain = nb; return; }
    int best = 0;
    unsigned left = pool;
    while (left) {
        unsigned comp = left & (0u - left), frontier = comp;
        while (frontier) {
            const int a = __ffs(frontier) - 1;
            frontier &= frontier - 1;
            const unsigned grow = R.row(a) & pool & ~comp;
            comp |= grow;
            frontier |= grow;
        }
        int cb = 0;
        for (unsigned m = comp; m; m &= m - 1)
            cb += __popc(R.row(__ffs(m) - 1) & comp);
        cb >>= 1;
        best = cb > best ? cb : best;
        left &= ~comp;
    }
    chain = best;
}

__device__ __forceinline__ double fold(double d, double L, double t_zero, double t_one)
{
    const double shift = (d >= t_one) ? L : ((d >= t_zero) ? 0.0 : -L);
    return d - shift; // == d - L*floor(d/L+0.5)   (box.h:120-124) when n is one of {-1,0,1}
}

// 0: the coordinates spread too far for `fold` (generic variant), 1: every pair has n in {-1,0,1}, 2: every pair has n = 0
// (no neighbour lies across the periodic seam: the minimum image is the plain difference, d - L*0 == d)
template <int NN>
__device__ __forceinline__ int span_class(const DBox &b, const double (&p)[NN], int axis)
{
    double mn = p[0], mx = p[0];
#pragma unroll
    for (int a = 1; a < NN; ++a) {
        mn = fmin(mn, p[a]);
        mx = fmax(mx, p[a]);
    }
    const double span = mx - mn;
    const double lim = fmin(b.tn[axis][3], -b.tn[axis][0]); // |d| below this => n in {-1,0,1}
    const double zero = fmin(b.tn[axis][2], -b.tn[axis][1]); // |d| below this => n == 0
    if (!(span < lim)) // false for NaN as well
        return 0;
    return span < zero ? 2 : 1;
}

// ---- bond matrix among NN listed neighbours: bit c of row a <=> pbcdis_sq(list[a], list[c]) <= cut2
// (cna.cpp:459-466; both ends RAW coordinates, cna.cpp:149-161).  Returns false (orthogonal hot path only)
// when the span test fails and the atom has to be redone by the GENERIC variant.
template <bool TRI, bool GENERIC, int NN>
__device__ __forceinline__ bool bond_rows(const DBox &b, const double *__restrict__ x, const double *__restrict__ y,
                                          const double *__restrict__ z, const int (&ids)[NN], double cut2, Rows &R)
{
    double px[NN], py[NN], pz[NN];
#pragma unroll
    for (int a = 0; a < NN; ++a) {
        const int j = ids[a]; // made safe by the caller (safe_id)
        px[a] = x[j]; py[a] = y[j]; pz[a] = z[j];
    }
    bool plain = false; // no fold needed on any axis
    if (!TRI && !GENERIC) {
        int cls = 2;
        if (b.pbc[0]) cls = min(cls, span_class<NN>(b, px, 0));
        if (b.pbc[1]) cls = min(cls, span_class<NN>(b, py, 1));
        if (b.pbc[2]) cls = min(cls, span_class<NN>(b, pz, 2));
        if (cls == 0)
            return false;
        plain = cls == 2;
    }
    unsigned adj[NN];
#pragma unroll
    for (int a = 0; a < NN; ++a)
        adj[a] = 0;
    if (!TRI && !GENERIC && plain) { // interior atoms (all but the layer at the periodic faces): 11 instructions per pair
#pragma unroll
        for (int a = 0; a < NN; ++a)
#pragma unroll
            for (int c = a + 1; c < NN; ++c) {
                const double dx = px[c] - px[a], dy = py[c] - py[a], dz = pz[c] - pz[a];
                if (dx * dx + dy * dy + dz * dz <= cut2) {
                    adj[a] |= 1u << c;
                    adj[c] |= 1u << a;
                }
            }
        R = pack_rows<NN>(adj);
        return true;
    }
#pragma unroll
    for (int a = 0; a < NN; ++a)
#pragma unroll
        for (int c = a + 1; c < NN; ++c) {
            double d2;
            if (TRI || GENERIC) {
                d2 = pair_d2<TRI>(b, px[a], py[a], pz[a], px[c], py[c], pz[c]);
            } else {
                double dx = px[c] - px[a], dy = py[c] - py[a], dz = pz[c] - pz[a];
                if (b.pbc[0]) dx = fold(dx, b.h[0], b.tn[0][1], b.tn[0][2]);
                if (b.pbc[1]) dy = fold(dy, b.h[4], b.tn[1][1], b.tn[1][2]);
                if (b.pbc[2]) dz = fold(dz, b.h[8], b.tn[2][1], b.tn[2][2]);
                d2 = dx * dx + dy * dy + dz * dz;
            }
            if (d2 <= cut2) {
                adj[a] |= 1u << c;
                adj[c] |= 1u << a;
            }
        }
    R = pack_rows<NN>(adj);
    return true;
}

// ------------------------------------------------------------------ fixed cutoff (cna.cpp:429-506)
// returns the label (0 = none) or -1 when the atom must be redone by the generic variant
template <bool TRI, bool GENERIC, int NN>
__device__ __forceinline__ int fcna_atom(const DBox &b, const double *__restrict__ x, const double *__restrict__ y,
                                         const double *__restrict__ z, const int *__restrict__ row, double cut2, int64_t i,
                                         int64_t N)
{
    int ids[NN];
#pragma unroll
    for (int a = 0; a < NN; ++a)
        ids[a] = safe_id(row[a], i, N);
    Rows R;
    if (!bond_rows<TRI, GENERIC, NN>(b, x, y, z, ids, cut2, R))
        return -1;
    int n421 = 0, n422 = 0, n555 = 0, n444 = 0, n666 = 0;
    for (int ni = 0; ni < NN; ++ni) { // no early exit (cna.cpp:471-494)
        int ncn, nb, ch;
        signature(R, ni, (1u << NN) - 1u, ncn, nb, ch);
        if (ncn == 4 && nb == 2) { n421 += (ch == 1); n422 += (ch == 2); }
        else if (ncn == 5 && nb == 5 && ch == 5) ++n555;
        else if (ncn == 4 && nb == 4 && ch == 4) ++n444;
        else if (ncn == 6 && nb == 6 && ch == 6) ++n666;
    }
    if (n421 == 12) return 1; // cna.cpp:496-503
    if (n421 == 6 && n422 == 6) return 2;
    if (n555 == 12) return 4;
    if (n666 == 8 && n444 == 6) return 3;
    return 0;
}

// to-do list of atoms left to the generic variant: todo[0] = count, todo[1..] = atom ids
__device__ __forceinline__ void defer(int *__restrict__ todo, int64_t i) { todo[1 + atomicAdd(&todo[0], 1)] = (int)i; }

template <bool TRI, bool GENERIC>
__global__ __launch_bounds__(256) void k_fcna(const double *__restrict__ x, const double *__restrict__ y,
                                              const double *__restrict__ z, int64_t N, DBox b,
                                              const int *__restrict__ verlet, int64_t M, const int *__restrict__ nn,
                                              int *__restrict__ pattern, double rc, int *__restrict__ todo)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (GENERIC) {
        if (i >= todo[0])
            return;
        i = todo[1 + i];
    } else if (i >= N) {
        return;
    }
    const int n = nn[i];
    const double cut2 = rc * rc; // cna.cpp:449
    const int *row = verlet + i * M;
    // atoms with nn not in {12,14} keep the caller's value (cna.cpp:456)
    int t = 0;
    if (n == 12 && M >= 12) t = fcna_atom<TRI, GENERIC, 12>(b, x, y, z, row, cut2, i, N);
    else if (n == 14 && M >= 14) t = fcna_atom<TRI, GENERIC, 14>(b, x, y, z, row, cut2, i, N);
    if (t > 0) pattern[i] = t;
    else if (!GENERIC && t < 0) defer(todo, i);
}

// ------------------------------------------------------------------ adaptive (cna.cpp:289-427)
template <bool TRI>
__device__ __forceinline__ double dist2_to(const DBox &b, const double *__restrict__ x, const double *__restrict__ y,
                                           const double *__restrict__ z, double xi, double yi, double zi, int j)
{
    return pair_d2<TRI>(b, xi, yi, zi, x[j], y[j], z[j]); // pbcdis_sq(i, j): exact reference expression
}

// returns the label, or -1 (redo with the generic variant)
template <bool TRI, bool GENERIC>
__device__ __forceinline__ int acna_atom(const DBox &b, const double *__restrict__ x, const double *__restrict__ y,
                                         const double *__restrict__ z, int64_t i, const int *__restrict__ row,
                                         int label, int64_t N)
{
    const double xi = x[i], yi = y[i], zi = z[i];
    int ids[14];
#pragma unroll
    for (int a = 0; a < 14; ++a)
        ids[a] = safe_id(row[a], i, N);
    // ---- 12 nearest neighbours: FCC / HCP / ICO (cna.cpp:309-370)
    double rs = 0.0;
#pragma unroll
    for (int m = 0; m < 12; ++m)
        rs += sqrt(dist2_to<TRI>(b, x, y, z, xi, yi, zi, ids[m]));
    double lc = rs / 12 * (1.0 + sqrt(2.0)) * 0.5; // :319
    {
        int ids12[12];
#pragma unroll
        for (int a = 0; a < 12; ++a)
            ids12[a] = ids[a];
        Rows R;
        if (!bond_rows<TRI, GENERIC, 12>(b, x, y, z, ids12, lc * lc, R))
            return -1;
        int n421 = 0, n422 = 0, n555 = 0;
        for (int ni = 0; ni < 12; ++ni) { // breaks: :334-362
            int ncn, nb, ch;
            signature(R, ni, 0xfffu, ncn, nb, ch);
            if (ncn != 4 && ncn != 5) break;
            if (nb != 2 && nb != 5) break;
            if (ncn == 4 && nb == 2) {
                if (ch == 1) ++n421;
                else if (ch == 2) ++n422;
                else break;
            } else if (ncn == 5 && nb == 5 && ch == 5) ++n555;
            else break;
        }
        if (n421 == 12) label = 1;
        else if (n421 == 6 && n422 == 6) label = 2;
        else if (n555 == 12) label = 4;
    }
    // ---- 14 nearest neighbours: BCC (cna.cpp:372-425), only if still unlabelled
    if (label == 0) {
        rs = 0.0;
#pragma unroll
        for (int m = 0; m < 8; ++m)
            rs += sqrt(dist2_to<TRI>(b, x, y, z, xi, yi, zi, ids[m]) / (3.0 / 4.0));
#pragma unroll
        for (int m = 8; m < 14; ++m)
            rs += sqrt(dist2_to<TRI>(b, x, y, z, xi, yi, zi, ids[m]));
        lc = rs / 14 * (1.0 + sqrt(2.0)) * 0.5;
        Rows R;
        if (!bond_rows<TRI, GENERIC, 14>(b, x, y, z, ids, lc * lc, R))
            return -1;
        int n444 = 0, n666 = 0;
        for (int ni = 0; ni < 14; ++ni) { // :398-421
            int ncn, nb, ch;
            signature(R, ni, 0x3fffu, ncn, nb, ch);
            if (ncn != 4 && ncn != 6) break;
            if (nb != 4 && nb != 6) break;
            if (ncn == 4 && nb == 4 && ch == 4) ++n444;
            else if (ncn == 6 && nb == 6 && ch == 6) ++n666;
            else break;
        }
        if (n666 == 8 && n444 == 6) label = 3;
    }
    return label;
}

template <bool TRI, bool GENERIC>
__global__ __launch_bounds__(256) void k_acna(const double *__restrict__ x, const double *__restrict__ y,
                                              const double *__restrict__ z, int64_t N, DBox b,
                                              const int *__restrict__ verlet, int64_t M, int *__restrict__ pattern,
                                              int *__restrict__ todo)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (GENERIC) {
        if (i >= todo[0])
            return;
        i = todo[1 + i];
    } else if (i >= N) {
        return;
    }
    const int t = acna_atom<TRI, GENERIC>(b, x, y, z, i, verlet + i * M, pattern[i], N);
    if (t >= 0) pattern[i] = t;
    else if (!GENERIC) defer(todo, i);
}

// ------------------------------------------------------------------ diamond, per-atom classification (cna.cpp:184-251)
// the 12 second neighbours are collected first (k_ids_second), so that the classification can be redone
__global__ __launch_bounds__(256) void k_ids_second(int64_t N, const int *__restrict__ verlet, int64_t M,
                                                    int *__restrict__ second)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    int cnt = 0;
    for (int m = 0; m < 4; ++m) { // :188-202; slots that are not reached keep the caller's content
        const int j = safe_id(verlet[i * M + m], i, N);
        int took = 0;
        for (int q = 0; q < 4; ++q) {
            const int k = safe_id(verlet[(int64_t)j * M + q], i, N);
            if (k != (int)i && took < 3) {
                second[i * 12 + cnt] = k;
                ++cnt;
                ++took;
            }
        }
    }
}

template <bool TRI, bool GENERIC>
__global__ __launch_bounds__(256) void k_ids_classify(const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ z, int64_t N, DBox b,
                                                      const int *__restrict__ second, int *__restrict__ pattern,
                                                      int *__restrict__ todo)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (GENERIC) {
        if (i >= todo[0])
            return;
        i = todo[1 + i];
    } else if (i >= N) {
        return;
    }
    int ids[12];
#pragma unroll
    for (int a = 0; a < 12; ++a)
        ids[a] = second[i * 12 + a];
    const double xi = x[i], yi = y[i], zi = z[i];
    double rs = 0.0;
#pragma unroll
    for (int m = 0; m < 12; ++m)
        rs += sqrt(dist2_to<TRI>(b, x, y, z, xi, yi, zi, ids[m]));
    rs /= 12.0;
    const double lc = rs * 1.2071068; // :212
    Rows R;
    if (!bond_rows<TRI, GENERIC, 12>(b, x, y, z, ids, lc * lc, R)) {
        defer(todo, i);
        return;
    }
    int n421 = 0, n422 = 0;
    for (int ni = 0; ni < 12; ++ni) { // :224-245
        int ncn, nb, ch;
        signature(R, ni, 0xfffu, ncn, nb, ch);
        if (ncn != 4) break;
        if (nb != 2) break;
        if (ch == 1) ++n421;
        else if (ch == 2) ++n422;
    }
    if (n421 == 12) pattern[i] = 1;
    else if (n421 == 6 && n422 == 6) pattern[i] = 4;
}

// ---- diamond: the two sequential promotion sweeps (cna.cpp:253-286) made parallel.
// In sweep s an atom j that is still 0 receives the label of the LOWEST-index
// atom i that (a) carries a source label of this sweep and (b) lists j among
// its first four neighbours: exactly the "first writer in index order wins"
// outcome of the serial loop, because a sweep's sources are fixed before it
// starts (sweep 1 sources are 1/4 — never produced by sweep 1; sweep 2 sources
// are 2/5 — never produced by sweep 2).
__global__ __launch_bounds__(256) void k_ids_claim(const int *__restrict__ verlet, int64_t M,
                                                   const int *__restrict__ pattern, int *__restrict__ claim,
                                                   int64_t N, int src_a, int src_b)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    const int t = pattern[i];
    if (t != src_a && t != src_b)
        return;
    for (int q = 0; q < 4; ++q) {
        const int j = safe_id(verlet[i * M + q], i, N);
        if (pattern[j] == 0)
            atomicMin(&claim[j], (int)i);
    }
}

__global__ __launch_bounds__(256) void k_ids_apply(int *__restrict__ pattern, const int *__restrict__ claim,
                                                   int64_t N, int src_a, int dst_a, int dst_b)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N)
        return;
    const int c = claim[j];
    if (c == 0x7fffffff || pattern[j] != 0)
        return;
    pattern[j] = (pattern[c] == src_a) ? dst_a : dst_b;
}

__global__ __launch_bounds__(256) void k_fill_int(int *__restrict__ p, int64_t n, int v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

} // namespace mdh

using namespace mdh;

extern "C" {

int mdh_fcna(const double *x, const double *y, const double *z, int64_t N, const double *box9,
             const double *origin3, const int *boundary3, const int *verlet, int64_t M, const int *nn, int *pattern,
             double rc, int space, void *stream)
{
    if (N < 0 || N >= 2147483647LL || M <= 0) { set_error("mdh_fcna: invalid shape"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const int *dn = sc.stage_in(nn, (size_t)N, space);
    int *dp = sc.stage(pattern, (size_t)N, space, true, true);
    int *todo = sc.alloc_n<int>((size_t)N + 1);
    if (sc.failed())
        return sc.error();
    MDH_HIP(hipMemsetAsync(todo, 0, sizeof(int), st));
    {
        ProfRange pr("k_fcna", st);
        dim3 grid(grid_for(N, 256)), block(256);
        if (b.tri) {
            hipLaunchKernelGGL((k_fcna<true, false>), grid, block, 0, st, dx, dy, dz, N, b, dv, M, dn, dp, rc, todo);
        } else {
            hipLaunchKernelGGL((k_fcna<false, false>), grid, block, 0, st, dx, dy, dz, N, b, dv, M, dn, dp, rc, todo);
            hipLaunchKernelGGL((k_fcna<false, true>), grid, block, 0, st, dx, dy, dz, N, b, dv, M, dn, dp, rc, todo);
        }
    }
    return sc.finish(space);
}

int mdh_acna(const double *x, const double *y, const double *z, int64_t N, const double *box9,
             const double *origin3, const int *boundary3, const int *verlet, int64_t M, int *pattern, int space,
             void *stream)
{
    if (N < 0 || N >= 2147483647LL || M < 14) { set_error("mdh_acna: verlet_list needs at least 14 columns"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    int *dp = sc.stage(pattern, (size_t)N, space, true, true);
    int *todo = sc.alloc_n<int>((size_t)N + 1);
    if (sc.failed())
        return sc.error();
    MDH_HIP(hipMemsetAsync(todo, 0, sizeof(int), st));
    dim3 grid(grid_for(N, 256)), block(256);
    if (b.tri) {
        hipLaunchKernelGGL((k_acna<true, false>), grid, block, 0, st, dx, dy, dz, N, b, dv, M, dp, todo);
    } else {
        hipLaunchKernelGGL((k_acna<false, false>), grid, block, 0, st, dx, dy, dz, N, b, dv, M, dp, todo);
        hipLaunchKernelGGL((k_acna<false, true>), grid, block, 0, st, dx, dy, dz, N, b, dv, M, dp, todo);
    }
    return sc.finish(space);
}

int mdh_ids(const double *x, const double *y, const double *z, int64_t N, const double *box9, const double *origin3,
            const int *boundary3, const int *verlet, int64_t M, int *new_verlet, int *pattern, int space,
            void *stream)
{
    if (N < 0 || N >= 2147483647LL || M < 4) { set_error("mdh_ids: verlet_list needs at least 4 columns"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    int *d2nd = sc.stage(new_verlet, (size_t)(N * 12), space, true, true);
    int *dp = sc.stage(pattern, (size_t)N, space, true, true);
    int *claim = sc.alloc_n<int>((size_t)N);
    int *todo = sc.alloc_n<int>((size_t)N + 1);
    if (sc.failed())
        return sc.error();
    MDH_HIP(hipMemsetAsync(todo, 0, sizeof(int), st));
    dim3 grid(grid_for(N, 256)), block(256);
    hipLaunchKernelGGL(k_ids_second, grid, block, 0, st, N, dv, M, d2nd);
    if (b.tri) {
        hipLaunchKernelGGL((k_ids_classify<true, false>), grid, block, 0, st, dx, dy, dz, N, b, d2nd, dp, todo);
    } else {
        hipLaunchKernelGGL((k_ids_classify<false, false>), grid, block, 0, st, dx, dy, dz, N, b, d2nd, dp, todo);
        hipLaunchKernelGGL((k_ids_classify<false, true>), grid, block, 0, st, dx, dy, dz, N, b, d2nd, dp, todo);
    }
    // sweep 1: 1 -> 2, 4 -> 5 ; sweep 2: 2 -> 3, 5 -> 6
    hipLaunchKernelGGL(k_fill_int, grid, block, 0, st, claim, N, 0x7fffffff);
    hipLaunchKernelGGL(k_ids_claim, grid, block, 0, st, dv, M, dp, claim, N, 1, 4);
    hipLaunchKernelGGL(k_ids_apply, grid, block, 0, st, dp, claim, N, 1, 2, 5);
    hipLaunchKernelGGL(k_fill_int, grid, block, 0, st, claim, N, 0x7fffffff);
    hipLaunchKernelGGL(k_ids_claim, grid, block, 0, st, dv, M, dp, claim, N, 2, 5);
    hipLaunchKernelGGL(k_ids_apply, grid, block, 0, st, dp, claim, N, 2, 3, 6);
    return sc.finish(space);
}
}
