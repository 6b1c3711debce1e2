// csp.hip — centro-symmetry parameter on gfx950.
//
// Replaces src/centro_symmetry_parameter.cpp:12-94 (get_csp): for every atom the
// K(K-1)/2 values |r_j + r_k|^2 over its first K listed neighbours, of which the
// K/2 smallest are summed in ascending order (partial_sort + sequential sum,
// :79-91).
//
// One thread per atom.  r_j (minimum-image vector to neighbour j) is evaluated
// once per neighbour — the reference re-evaluates it for every pair but from
// identical operands, so the values are bit-identical — and the K/2 smallest
// pair values are kept in a sorted register array.
#include "common.hpp"

namespace mdh {

template <bool TRI, int K>
__device__ __forceinline__ double csp_atom_static(const DBox &b, const Pos4 *__restrict__ pos,
                                                  int64_t i, const int *__restrict__ row, int64_t N)
{
    constexpr int H = K / 2;
    const Pos4 pi = pos[i]; // RAW centre (:46-48)
    const double xi = pi.x, yi = pi.y, zi = pi.z;
    double rx[K], ry[K], rz[K];
    int ids[K];
    load_row<K>(row, ids);
    // all K gathers first, the minimum image afterwards: with both in one loop body the compiler keeps every load behind the
    // branches of the previous neighbour's pbc (it does not hoist loads over control flow) — K dependent memory latencies
#pragma unroll
    for (int a = 0; a < K; ++a) {
        const Pos4 pj = pos[safe_id(ids[a], i, N)];
        rx[a] = pj.x - xi; ry[a] = pj.y - yi; rz[a] = pj.z - zi;
    }
#pragma unroll
    for (int a = 0; a < K; ++a)
        pbc<TRI>(b, rx[a], ry[a], rz[a]);
    double top[H];
#pragma unroll
    for (int q = 0; q < H; ++q)
        top[q] = __builtin_huge_val();
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int c = a + 1; c < K; ++c) {
            const double sx = rx[a] + rx[c], sy = ry[a] + ry[c], sz = rz[a] + rz[c];
            double v = sx * sx + sy * sy + sz * sz; // :73
            if (v < top[H - 1]) {
#pragma unroll
                for (int q = 0; q < H; ++q) { // sorted insert by compare-exchange
                    const double lo = v < top[q] ? v : top[q];
                    const double hi = v < top[q] ? top[q] : v;
                    top[q] = lo;
                    v = hi;
                }
            }
        }
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < H; ++q)
        s += top[q];
    return s;
}

static constexpr int CSP_MAXK = 64;

template <bool TRI>
__device__ double csp_atom_dynamic(const DBox &b, const double *__restrict__ x, const double *__restrict__ y,
                                   const double *__restrict__ z, int64_t i, const int *__restrict__ row, int K, int64_t N)
{
    const int H = K / 2;
    const double xi = x[i], yi = y[i], zi = z[i];
    double rx[CSP_MAXK], ry[CSP_MAXK], rz[CSP_MAXK], top[CSP_MAXK / 2];
    for (int a = 0; a < K; ++a) {
        const int j = safe_id(row[a], i, N);
        double dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
        pbc<TRI>(b, dx, dy, dz);
        rx[a] = dx; ry[a] = dy; rz[a] = dz;
    }
    for (int q = 0; q < H; ++q)
        top[q] = __builtin_huge_val();
    for (int a = 0; a < K; ++a)
        for (int c = a + 1; c < K; ++c) {
            const double sx = rx[a] + rx[c], sy = ry[a] + ry[c], sz = rz[a] + rz[c];
            double v = sx * sx + sy * sy + sz * sz;
            if (H > 0 && v < top[H - 1]) {
                int q = H - 1;
                while (q > 0 && top[q - 1] > v) { top[q] = top[q - 1]; --q; }
                top[q] = v;
            }
        }
    double s = 0.0;
    for (int q = 0; q < H; ++q)
        s += top[q];
    return s;
}

// Two kernels, not one with a branch on K: the general path keeps its pair vectors in scratch memory (1.8 KB per lane), and a
// kernel reserves its scratch whether a launch takes that path or not — the first such launch of a process pays the queue's
// scratch allocation (24 ms for ~1 GB on this chip) and every launch runs at scratch-limited occupancy.
template <bool TRI, int K>
__global__ __launch_bounds__(256) void k_csp(int64_t N, DBox b, const int *__restrict__ verlet, int64_t M,
                                             double *__restrict__ csp, const Pos4 *__restrict__ pos)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    csp[i] = csp_atom_static<TRI, K>(b, pos, i, verlet + i * M, N);
}

template <bool TRI>
__global__ __launch_bounds__(256) void k_csp_any(const double *__restrict__ x, const double *__restrict__ y,
                                                 const double *__restrict__ z, int64_t N, DBox b,
                                                 const int *__restrict__ verlet, int64_t M, int K, double *__restrict__ csp)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    csp[i] = csp_atom_dynamic<TRI>(b, x, y, z, i, verlet + i * M, K, N);
}

} // namespace mdh

using namespace mdh;

extern "C" int mdh_csp(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                       const double *origin3, const int *boundary3, const int *verlet, int64_t M, int num_neigh,
                       double *csp, int space, void *stream)
{
    if (N < 0 || num_neigh <= 0 || (num_neigh & 1) || num_neigh > M || num_neigh > CSP_MAXK) {
        set_error("mdh_csp: num_neigh must be a positive even number <= min(verlet columns, 64)");
        return MDH_ERR_ARG;
    }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    double *dc = sc.stage(csp, (size_t)N, space, false, true);
    if (sc.failed())
        return sc.error();
    const dim3 grid(grid_for(N, 256)), block(256);
    if (num_neigh == 12 || num_neigh == 8) {
        const Pos4 *pos = pack_positions(sc, dx, dy, dz, N);
        if (!pos)
            return sc.error();
        if (num_neigh == 12) {
            if (b.tri) hipLaunchKernelGGL((k_csp<true, 12>), grid, block, 0, sc.stream(), N, b, dv, M, dc, pos);
            else hipLaunchKernelGGL((k_csp<false, 12>), grid, block, 0, sc.stream(), N, b, dv, M, dc, pos);
        } else {
            if (b.tri) hipLaunchKernelGGL((k_csp<true, 8>), grid, block, 0, sc.stream(), N, b, dv, M, dc, pos);
            else hipLaunchKernelGGL((k_csp<false, 8>), grid, block, 0, sc.stream(), N, b, dv, M, dc, pos);
        }
    } else if (b.tri) {
        hipLaunchKernelGGL(k_csp_any<true>, grid, block, 0, sc.stream(), dx, dy, dz, N, b, dv, M, num_neigh, dc);
    } else {
        hipLaunchKernelGGL(k_csp_any<false>, grid, block, 0, sc.stream(), dx, dy, dz, N, b, dv, M, num_neigh, dc);
    }
    return sc.finish(space);
}

MDH_WARM_UNIT(csp)
