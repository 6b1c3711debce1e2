// order.hip — atom ORDER: how far the order in which atoms were handed in is from a spatial one, a spatial sort, and the
// translation of results between the two index spaces.
//
// The reference's cell build (src/neighbor.cpp:64-100: a linked list per cell) and its consumers do not care in which order atoms
// arrive; on a GPU the order decides whether the gathers of a neighbour's position hit L2 or HBM: an id-sorted dump of a diffused
// system, or a shuffled one, ran the fixed-cutoff CNA at a twelfth of its speed (profiles/r05_order_*).  The host layer
// (mdapy_amd/system.py) therefore keeps a cell-sorted copy of such a system, builds its lists with the original index as the in-cell
// ordering key (rows in the reference's order, mdh_build_neighbor_keyed), runs the analyses in sorted space and translates what the
// user reads — per-atom columns, and the rows when they are asked for — back.  No reference counterpart (there is nothing to do on
// a CPU); every entry is declared in include/mdapy_amd.h.
#include "common.hpp"
#include "grid.hpp"
#include <algorithm>
#include <cmath>

namespace mdh {

// fraction of consecutive atoms (i, i+1) that are NOT in the same or a touching bin of an nb0 x nb1 x nb2 grid over the box
// (fractional coordinates; periodic axes wrap)
struct OrderGrid { int nb[3]; };
template <bool TRI>
__global__ __launch_bounds__(256) void k_order_statistic(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                         int64_t N, DBox b, OrderGrid og, int64_t stride, int64_t samples,
                                                         unsigned long long *__restrict__ far_pairs)
{
    // (a sample of the pairs: every stride-th one — at most 2^18 of them, one atomic per workgroup: a count by every wave of a
    // 10 M-atom system queued 157 k atomics on one word, 1.9 ms)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t * stride;
    bool far = false;
    if (t < samples && i + 1 < N) {
        int c[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const double rx = x[i + s] - b.o[0], ry = y[i + s] - b.o[1], rz = z[i + s] - b.o[2];
            double f[3];
            if (TRI) {
                f[0] = rx * b.hi[0] + ry * b.hi[3] + rz * b.hi[6];
                f[1] = rx * b.hi[1] + ry * b.hi[4] + rz * b.hi[7];
                f[2] = rx * b.hi[2] + ry * b.hi[5] + rz * b.hi[8];
            } else {
                f[0] = rx / b.h[0]; f[1] = ry / b.h[4]; f[2] = rz / b.h[8];
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                double g = f[d];
                if (b.pbc[d]) g = g - floor(g);
                g = fmin(fmax(g, 0.0), 0.999999999);
                c[s][d] = (int)(g * (double)og.nb[d]); // (NaN -> 0)
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            int dd = abs(c[0][d] - c[1][d]);
            if (b.pbc[d]) dd = min(dd, og.nb[d] - dd);
            far = far || dd > 1;
        }
    }
    __shared__ unsigned s_far;
    if (threadIdx.x == 0) s_far = 0;
    __syncthreads();
    const unsigned long long m = __ballot(far);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_far, (unsigned)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0 && s_far) atomicAdd(far_pairs, (unsigned long long)s_far);
}

__global__ __launch_bounds__(256) void k_unpack_sorted(const CellGrid::Packed *__restrict__ pk, int64_t N, double *__restrict__ xs,
                                                       double *__restrict__ ys, double *__restrict__ zs, int *__restrict__ perm)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N)
        return;
    const CellGrid::Packed r = pk[p];
    xs[p] = r.x; ys[p] = r.y; zs[p] = r.z; perm[p] = r.id;
}

// out[p] = in[perm[p]] (gather) or out[perm[p]] = in[p] (scatter); T = 4- or 8-byte words
template <class T, bool SCATTER>
__global__ __launch_bounds__(256) void k_permute(const T *__restrict__ in, const int *__restrict__ perm, int64_t N, T *__restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N)
        return;
    const int q = perm[p];
    if (SCATTER) out[q] = in[p];
    else out[p] = in[q];
}

// three f64 columns through one permutation, four atoms per thread: the permutation read once, twelve gathers in flight per thread
__global__ __launch_bounds__(256) void k_gather3(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                 const int *__restrict__ perm, int64_t N, double *__restrict__ xs, double *__restrict__ ys,
                                                 double *__restrict__ zs)
{
    const int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    int q[4];
    double a[4], b[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = perm[p0 + u < N ? p0 + u : N - 1]; // (unconditional: a branch around a load is followed by a wait for it)
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = x[q[u]]; b[u] = y[q[u]]; c[u] = z[q[u]]; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (p0 + u < N) { xs[p0 + u] = a[u]; ys[p0 + u] = b[u]; zs[p0 + u] = c[u]; }
}

// rows of a list built in sorted space -> the original index space: row p goes to row perm[p], its entries j >= 0 become perm[j]
// (pads stay).  A thread per (row, four slots); rows of a multiple of four slots move in 16-byte pieces.
typedef int RowPiece __attribute__((ext_vector_type(4), aligned(4)));
typedef double DistPiece __attribute__((ext_vector_type(2), aligned(8)));
__global__ __launch_bounds__(256) void k_translate_rows(const int *__restrict__ vs, const double *__restrict__ ds, const int *__restrict__ ns,
                                                        const int *__restrict__ perm, int64_t N, int64_t M, int groups, int *__restrict__ vo,
                                                        double *__restrict__ dso, int *__restrict__ no)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * groups)
        return;
    const int64_t p = t / groups;
    const int g0 = (int)(t - p * groups) * 4;
    const int64_t i = perm[p];
    if (g0 == 0 && ns) no[i] = ns[p];
    const int64_t src = p * M + g0, dst = i * M + g0;
    if (g0 + 4 <= M) {
        const RowPiece v = *reinterpret_cast<const RowPiece *>(vs + src);
        RowPiece o;
        o.x = v.x >= 0 && v.x < N ? perm[v.x] : v.x;
        o.y = v.y >= 0 && v.y < N ? perm[v.y] : v.y;
        o.z = v.z >= 0 && v.z < N ? perm[v.z] : v.z;
        o.w = v.w >= 0 && v.w < N ? perm[v.w] : v.w;
        *reinterpret_cast<RowPiece *>(vo + dst) = o;
        if (ds) {
            const DistPiece a = *reinterpret_cast<const DistPiece *>(ds + src), c = *reinterpret_cast<const DistPiece *>(ds + src + 2);
            *reinterpret_cast<DistPiece *>(dso + dst) = a;
            *reinterpret_cast<DistPiece *>(dso + dst + 2) = c;
        }
    } else {
        for (int s = g0; s < M; ++s) {
            const int v = vs[p * M + s];
            vo[i * M + s] = v >= 0 && v < N ? perm[v] : v;
            if (ds) dso[i * M + s] = ds[p * M + s];
        }
    }
}

static double box_volume(const DBox &b)
{
    const double *h = b.h;
    return std::fabs(h[0] * (h[4] * h[8] - h[5] * h[7]) - h[1] * (h[3] * h[8] - h[5] * h[6]) + h[2] * (h[3] * h[7] - h[4] * h[6]));
}

} // namespace mdh

using namespace mdh;

extern "C" {

int mdh_order_statistic(const double *x, const double *y, const double *z, int64_t N, const double *box9, const double *origin3,
                        const int *boundary3, double *far_fraction, int space, void *stream)
{
    if (N < 0 || N >= 2147483647LL || !far_fraction) { set_error("mdh_order_statistic: invalid N"); return MDH_ERR_ARG; }
    *far_fraction = 0.0;
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N < 2)
        return MDH_OK;
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    unsigned long long *cnt = sc.alloc_n<unsigned long long>(1);
    if (sc.failed())
        return sc.error();
    hipStream_t st = sc.stream();
    MDH_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), st));
    // bins of ~64 atoms: two atoms that follow each other in a spatial order of any kind (a lattice builder, a file written cell
    // by cell, a previous sort) are in touching bins
    const double edge = std::cbrt(64.0 * box_volume(b) / (double)N);
    OrderGrid og;
    for (int d = 0; d < 3; ++d) {
        const double f = std::floor(b.thick[d] / edge);
        og.nb[d] = f >= 1.0 ? (f < 1024.0 ? (int)f : 1024) : 1;
    }
    const int64_t stride = std::max<int64_t>(1, (N - 1) >> 18), samples = (N - 1 + stride - 1) / stride;
    if (b.tri) hipLaunchKernelGGL(k_order_statistic<true>, dim3(grid_for(samples, 256)), dim3(256), 0, st, dx, dy, dz, N, b, og, stride, samples, cnt);
    else hipLaunchKernelGGL(k_order_statistic<false>, dim3(grid_for(samples, 256)), dim3(256), 0, st, dx, dy, dz, N, b, og, stride, samples, cnt);
    unsigned long long host = 0;
    MDH_HIP(hipMemcpyAsync(&host, cnt, sizeof(host), hipMemcpyDeviceToHost, st));
    MDH_HIP(hipStreamSynchronize(st));
    *far_fraction = (double)host / (double)samples;
    return MDH_OK;
}

int mdh_spatial_sort(const double *x, const double *y, const double *z, int64_t N, const double *box9, const double *origin3,
                     const int *boundary3, double *xs, double *ys, double *zs, int *perm, int64_t *n_sorted, int space, void *stream)
{
    if (N < 0 || N >= 2147483647LL || !n_sorted) { set_error("mdh_spatial_sort: invalid N"); return MDH_ERR_ARG; }
    *n_sorted = 0;
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    double *oxs = sc.stage(xs, (size_t)N, space, false, true), *oys = sc.stage(ys, (size_t)N, space, false, true), *ozs = sc.stage(zs, (size_t)N, space, false, true);
    int *operm = sc.stage(perm, (size_t)N, space, false, true);
    if (sc.failed())
        return sc.error();
    hipStream_t st = sc.stream();
    // the cells of a cutoff search at the density's natural scale (2.5 atoms per cell — fcc Cu: 3.09 A, its first-shell cutoff):
    // atoms sorted by these cells are in (nearly) the order any later rc-wide grid of the analyses walks them
    double edge = std::cbrt(2.5 * box_volume(b) / (double)N);
    CellGrid cg;
    for (;;) { // (a grid of at most 2^27 cells)
        MDH_TRY(neighbor_grid_dims(b, edge, cg.g));
        if (cg.g.ncell <= (int64_t(1) << 27))
            break;
        edge *= 1.26;
    }
    MDH_TRY(build_cell_grid(sc, dx, dy, dz, N, b, true, true, cg, nullptr, true, true)); // (scattered input: that is why the caller sorts)
    hipLaunchKernelGGL(k_unpack_sorted, dim3(grid_for(N, 256)), dim3(256), 0, st, cg.pk, N, oxs, oys, ozs, operm);
    int binned = 0;
    MDH_HIP(hipMemcpyAsync(&binned, cg.cell_start + cg.g.ncell, sizeof(int), hipMemcpyDeviceToHost, st));
    MDH_TRY(sc.finish(space));
    MDH_HIP(hipStreamSynchronize(st));
    *n_sorted = binned; // < N: absent atoms (x = NaN) were handed in — perm is no permutation, the caller keeps its own order
    return MDH_OK;
}

int mdh_permute(const void *in, const int *perm, int64_t N, int elem_bytes, int scatter, void *out, int space, void *stream)
{
    if (N < 0 || (elem_bytes != 4 && elem_bytes != 8) || in == out) { set_error("mdh_permute: 4- or 8-byte elements, out of place"); return MDH_ERR_ARG; }
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const int *dp = sc.stage_in(perm, (size_t)N, space);
    hipStream_t st = sc.stream();
    const dim3 grid(grid_for(N, 256)), block(256);
    if (elem_bytes == 4) {
        const unsigned *di = sc.stage_in(static_cast<const unsigned *>(in), (size_t)N, space);
        unsigned *dout = sc.stage(static_cast<unsigned *>(out), (size_t)N, space, false, true);
        if (sc.failed()) return sc.error();
        if (scatter) hipLaunchKernelGGL((k_permute<unsigned, true>), grid, block, 0, st, di, dp, N, dout);
        else hipLaunchKernelGGL((k_permute<unsigned, false>), grid, block, 0, st, di, dp, N, dout);
    } else {
        const unsigned long long *di = sc.stage_in(static_cast<const unsigned long long *>(in), (size_t)N, space);
        unsigned long long *dout = sc.stage(static_cast<unsigned long long *>(out), (size_t)N, space, false, true);
        if (sc.failed()) return sc.error();
        if (scatter) hipLaunchKernelGGL((k_permute<unsigned long long, true>), grid, block, 0, st, di, dp, N, dout);
        else hipLaunchKernelGGL((k_permute<unsigned long long, false>), grid, block, 0, st, di, dp, N, dout);
    }
    MDH_HIP(hipGetLastError());
    return sc.finish(space);
}

int mdh_gather_positions(const double *x, const double *y, const double *z, const int *perm, int64_t N, double *xs, double *ys, double *zs,
                         int space, void *stream)
{
    if (N < 0 || N >= 2147483647LL) { set_error("mdh_gather_positions: invalid N"); return MDH_ERR_ARG; }
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    const int *dp = sc.stage_in(perm, (size_t)N, space);
    double *ox = sc.stage(xs, (size_t)N, space, false, true), *oy = sc.stage(ys, (size_t)N, space, false, true), *oz = sc.stage(zs, (size_t)N, space, false, true);
    if (sc.failed())
        return sc.error();
    hipLaunchKernelGGL(k_gather3, dim3(grid_for((N + 3) / 4, 256)), dim3(256), 0, sc.stream(), dx, dy, dz, dp, N, ox, oy, oz);
    MDH_HIP(hipGetLastError());
    return sc.finish(space);
}

int mdh_translate_rows(const int *verlet_sorted, const double *dist_sorted, const int *nn_sorted, const int *perm, int64_t N, int64_t M,
                       int *verlet, double *dist, int *nn, int space, void *stream)
{
    if (N < 0 || N >= 2147483647LL || M <= 0 || !verlet_sorted || !verlet || !perm || (!dist_sorted) != (!dist) || (!nn_sorted) != (!nn)) {
        set_error("mdh_translate_rows: bad arguments");
        return MDH_ERR_ARG;
    }
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const int *vs = sc.stage_in(verlet_sorted, (size_t)(N * M), space);
    const double *ds = dist_sorted ? sc.stage_in(dist_sorted, (size_t)(N * M), space) : nullptr;
    const int *ns = nn_sorted ? sc.stage_in(nn_sorted, (size_t)N, space) : nullptr;
    const int *dp = sc.stage_in(perm, (size_t)N, space);
    int *vo = sc.stage(verlet, (size_t)(N * M), space, false, true);
    double *dso = dist ? sc.stage(dist, (size_t)(N * M), space, false, true) : nullptr;
    int *no = nn ? sc.stage(nn, (size_t)N, space, false, true) : nullptr;
    if (sc.failed())
        return sc.error();
    const int groups = (int)((M + 3) / 4);
    hipLaunchKernelGGL(k_translate_rows, dim3(grid_for(N * groups, 256)), dim3(256), 0, sc.stream(), vs, ds, ns, dp, N, M, groups, vo, dso, no);
    MDH_HIP(hipGetLastError());
    return sc.finish(space);
}

} // extern "C"
