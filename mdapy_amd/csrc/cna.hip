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
#include "cna_core.hpp"

namespace mdh {

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
    if (TRI || GENERIC) {
        R = bond_rows_reg<TRI, NN>(b, px, py, pz, cut2);
        return true;
    }
    const int cls = span_class3<NN>(b, px, py, pz);
    if (cls == 0)
        return false;
    R = bond_rows_ortho<NN>(b, px, py, pz, cut2, cls == 2);
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
    return fcna_label<NN>(R);
}

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

void launch_fcna_all(hipStream_t st, const DBox &b, const double *x, const double *y, const double *z, int64_t N, const int *verlet,
                     int64_t M, const int *nn, int *pattern, double rc, int *todo)
{
    dim3 grid(grid_for(N, 256)), block(256);
    if (b.tri) {
        hipLaunchKernelGGL((k_fcna<true, false>), grid, block, 0, st, x, y, z, N, b, verlet, M, nn, pattern, rc, todo);
    } else {
        hipLaunchKernelGGL((k_fcna<false, false>), grid, block, 0, st, x, y, z, N, b, verlet, M, nn, pattern, rc, todo);
        hipLaunchKernelGGL((k_fcna<false, true>), grid, block, 0, st, x, y, z, N, b, verlet, M, nn, pattern, rc, todo);
    }
}

void launch_fcna_listed(hipStream_t st, const DBox &b, const double *x, const double *y, const double *z, int64_t N, const int *verlet,
                        int64_t M, const int *nn, int *pattern, double rc, int *todo)
{
    dim3 grid(grid_for(N, 256)), block(256); // (the list's length is on the device: threads beyond it leave at once)
    if (b.tri)
        hipLaunchKernelGGL((k_fcna<true, true>), grid, block, 0, st, x, y, z, N, b, verlet, M, nn, pattern, rc, todo);
    else
        hipLaunchKernelGGL((k_fcna<false, true>), grid, block, 0, st, x, y, z, N, b, verlet, M, nn, pattern, rc, todo);
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
        launch_fcna_all(st, b, dx, dy, dz, N, dv, M, dn, dp, rc, todo);
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
