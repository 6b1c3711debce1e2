// polycrystal.hip — grain filling for the polycrystal builder on gfx950.
//
// Replaces _polycrystal.transform_and_filter (src/polycrystal.cpp:20-125): every atom of the replicated unit cell is rotated
// about the cell's centre, moved to the grain seed and kept when it lies strictly inside all face planes of the grain's
// Voronoi cell.  The reference evaluates
//     p'_k = ((dx*R(k,0) + dy*R(k,1)) + dz*R(k,2)) + t_k          val_f = ((p'x*a_f + p'y*b_f) + p'z*c_f) + d_f
// left to right (:78-92) and compacts the survivors in input order; both are reproduced bit for bit
// (-ffp-contract=off).  HBM-bound streaming: 24 B read per atom, 24 B written per survivor.
#include "common.hpp"
#include "grid.hpp"

namespace mdh {

static constexpr int PC_MAX_PLANES = 1024; // face planes kept in LDS

struct PcXform { double r[9], c[3], t[3]; };

__device__ __forceinline__ void pc_transform(const PcXform &X, double x, double y, double z, double &px, double &py, double &pz)
{
    const double dx = x - X.c[0], dy = y - X.c[1], dz = z - X.c[2];
    // RTck = R(k,c) (:49-51): p'_k = ((dx*R(k,0) + dy*R(k,1)) + dz*R(k,2)) + t_k = (R d)_k + t_k  (:78-80)
    px = dx * X.r[0] + dy * X.r[1] + dz * X.r[2] + X.t[0];
    py = dx * X.r[3] + dy * X.r[4] + dz * X.r[5] + X.t[1];
    pz = dx * X.r[6] + dy * X.r[7] + dz * X.r[8] + X.t[2];
}

__global__ __launch_bounds__(256) void k_pc_mask(const double *__restrict__ x, const double *__restrict__ y,
                                                 const double *__restrict__ z, int64_t n, PcXform X,
                                                 const double *__restrict__ planes, int nf, unsigned *__restrict__ inside)
{
    __shared__ double pl[PC_MAX_PLANES * 4];
    for (int q = threadIdx.x; q < nf * 4; q += blockDim.x) pl[q] = planes[q];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double px, py, pz;
    pc_transform(X, x[i], y[i], z[i], px, py, pz);
    bool in = true;
    for (int f = 0; f < nf && in; ++f)
        in = !(px * pl[4 * f] + py * pl[4 * f + 1] + pz * pl[4 * f + 2] + pl[4 * f + 3] >= 0.0); // :88-92
    inside[i] = in ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_pc_write(const double *__restrict__ x, const double *__restrict__ y,
                                                  const double *__restrict__ z, int64_t n, PcXform X,
                                                  const unsigned *__restrict__ inside, const int *__restrict__ slot,
                                                  double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !inside[i])
        return;
    double px, py, pz;
    pc_transform(X, x[i], y[i], z[i], px, py, pz);
    double *o = out + (int64_t)slot[i] * 3;
    o[0] = px; o[1] = py; o[2] = pz;
}

} // namespace mdh

using namespace mdh;

// out_pos: (n, 3) row-major capacity; rows [0, *count_host) are written.  rotation9 / center3 / target3 / coeffs (nf, 4): host.
extern "C" int mdh_transform_and_filter(const double *x, const double *y, const double *z, int64_t n, const double *rotation9,
                                        const double *center3, const double *target3, const double *coeffs, int nf,
                                        double *out_pos, int64_t *count_host, int space, void *stream)
{
    if (n < 0 || nf < 0 || nf > PC_MAX_PLANES || n >= 2147483647LL) {
        set_error("mdh_transform_and_filter: bad sizes (at most 1024 planes)");
        return MDH_ERR_ARG;
    }
    if (count_host) *count_host = 0;
    if (n == 0)
        return MDH_OK;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const double *dx = sc.stage_in(x, (size_t)n, space), *dy = sc.stage_in(y, (size_t)n, space), *dz = sc.stage_in(z, (size_t)n, space);
    const double *dpl = sc.stage_in(coeffs, (size_t)nf * 4, MDH_HOST);
    double *dout = sc.stage(out_pos, (size_t)n * 3, space, false, true);
    unsigned *inside = sc.alloc_n<unsigned>((size_t)n);
    int *slot = sc.alloc_n<int>((size_t)n + 1);
    if (sc.failed())
        return sc.error();
    PcXform X;
    for (int q = 0; q < 9; ++q) X.r[q] = rotation9[q];
    for (int q = 0; q < 3; ++q) { X.c[q] = center3[q]; X.t[q] = target3[q]; }
    const dim3 grid(grid_for(n, 256)), block(256);
    hipLaunchKernelGGL(k_pc_mask, grid, block, 0, st, dx, dy, dz, n, X, dpl, nf, inside);
    MDH_TRY(exclusive_scan_u32(sc, inside, slot, n));
    hipLaunchKernelGGL(k_pc_write, grid, block, 0, st, dx, dy, dz, n, X, inside, slot, dout);
    int cnt = 0;
    MDH_HIP(hipMemcpyAsync(&cnt, slot + n, sizeof(int), hipMemcpyDeviceToHost, st));
    MDH_HIP(hipStreamSynchronize(st)); // also: coeffs were staged from caller memory
    if (count_host) *count_host = cnt;
    return sc.finish(space);
}
