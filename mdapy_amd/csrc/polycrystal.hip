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

// ---------------------------------------------------------------------------------------------------------------
// Overlap filter of the graphene-decorated polycrystal (src/neighbor.cpp:489-672): atoms of type 1 (metal) and 2 (carbon)
// with a grain id; pairs closer than the cutoff of their kind lose one partner.  The reference sweeps the atoms in index
// order and lets every still-present atom i act on its still-present higher-numbered neighbours j:
//     metal-metal:  j goes;   carbon-carbon: same grain -> j goes, different grains -> the one with the larger grain id;
//     metal-carbon: the metal goes.
// An atom that was removed before its turn does not act (:548-551); one that removes itself keeps acting until its loop
// ends.  With more than one thread the reference reads and sets the flags concurrently (result depends on the schedule);
// the serial sweep is the one that is defined and it is what is reproduced here: rounds in which an atom acts only when
// it is the lowest-numbered unfinished atom on itself and on all its neighbours within the largest cutoff (atomicMin
// "owner" publish), so that atoms acting in the same round have disjoint footprints and every atom finds exactly the
// flags the serial sweep would show it.
namespace mdh {

__global__ __launch_bounds__(256) void k_ovg_publish(const int *__restrict__ verlet, const int *__restrict__ nn, int64_t N, int64_t M,
                                                     const unsigned char *__restrict__ done, int *__restrict__ owner)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || done[i])
        return;
    atomicMin(&owner[i], (int)i);
    const int n = min(nn[i], (int)M);
    for (int q = 0; q < n; ++q) atomicMin(&owner[verlet[i * M + q]], (int)i);
}

template <bool TRI>
__global__ __launch_bounds__(256) void k_ovg_sweep(const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ z, DBox b, const int *__restrict__ verlet,
                                                   const int *__restrict__ nn, int64_t N, int64_t M, const int *__restrict__ type,
                                                   const int *__restrict__ grain, double rc_mm, double rc_cc, double rc_mc,
                                                   unsigned char *__restrict__ done, const int *__restrict__ owner,
                                                   unsigned char *__restrict__ removed, int *__restrict__ remaining)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || done[i])
        return;
    const int n = min(nn[i], (int)M);
    bool ready = owner[i] == (int)i;
    for (int q = 0; q < n && ready; ++q) ready = owner[verlet[i * M + q]] == (int)i;
    if (!ready) {
        atomicAdd(remaining, 1);
        return;
    }
    done[i] = 1;
    if (removed[i])
        return;
    const int ti = type[i], gi = grain[i];
    double xi = x[i], yi = y[i], zi = z[i];
    if (b.anypbc) // :557-560
        wrap<TRI>(b, xi, yi, zi);
    for (int q = 0; q < n; ++q) {
        const int j = verlet[i * M + q];
        if (j <= (int)i || removed[j])
            continue;
        const int tj = type[j], gj = grain[j];
        double dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi; // the reference's squared distance, bit for bit (:582-586)
        pbc<TRI>(b, dx, dy, dz);
        const double dsq = dx * dx + dy * dy + dz * dz;
        int target = -1;
        if (ti == 1 && tj == 1) {
            if (dsq <= rc_mm * rc_mm) target = j;
        } else if (ti == 2 && tj == 2) {
            if (dsq <= rc_cc * rc_cc) target = (gi != gj) ? ((gi > gj) ? (int)i : j) : j;
        } else if (ti != tj) {
            if (dsq <= rc_mc * rc_mc) target = (ti == 1) ? (int)i : j;
        }
        if (target >= 0) removed[target] = 1;
    }
}

__global__ void k_ovg_keep(const unsigned char *__restrict__ removed, int64_t N, unsigned char *__restrict__ keep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) keep[i] = removed[i] ? 0 : 1;
}

} // namespace mdh

extern "C" int mdh_filter_overlap_atom_with_grain(const double *x, const double *y, const double *z, const int *type, const int *grain_id,
                                                  int64_t N, const double *box9, const double *origin3, const int *boundary3,
                                                  double rc_metal_metal, double rc_cc, double rc_metal_c, unsigned char *keep,
                                                  int space, void *stream)
{
    if (N < 0 || N >= 2147483647LL || !(rc_metal_metal > 0) || !(rc_cc > 0) || !(rc_metal_c > 0)) {
        set_error("mdh_filter_overlap_atom_with_grain: invalid N or cutoffs");
        return MDH_ERR_ARG;
    }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    const double rc_max = fmax(rc_metal_metal, fmax(rc_cc, rc_metal_c));
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    const int *dt = sc.stage_in(type, (size_t)N, space), *dg = sc.stage_in(grain_id, (size_t)N, space);
    unsigned char *dkeep = sc.stage(keep, (size_t)N, space, false, true);
    int *dnn = sc.alloc_n<int>((size_t)N);
    if (sc.failed())
        return sc.error();
    int maxc = 0;
    MDH_TRY(mdh_neighbor_count(dx, dy, dz, N, box9, origin3, boundary3, rc_max, dnn, &maxc, MDH_DEVICE, stream));
    const int64_t M = maxc > 0 ? maxc : 1;
    if ((double)N * (double)M > 4.0e8) { set_error("mdh_filter_overlap_atom_with_grain: the pair list would exceed 4e8 entries"); return MDH_ERR_ARG; }
    int *dv = sc.alloc_n<int>((size_t)(N * M));
    double *dd = sc.alloc_n<double>((size_t)(N * M));
    unsigned char *done = sc.alloc_n<unsigned char>((size_t)N), *removed = sc.alloc_n<unsigned char>((size_t)N);
    int *owner = sc.alloc_n<int>((size_t)N), *remaining = sc.alloc_n<int>(1);
    if (sc.failed())
        return sc.error();
    MDH_TRY(mdh_build_neighbor(dx, dy, dz, N, box9, origin3, boundary3, rc_max, dv, dd, dnn, M, 1, MDH_DEVICE, stream));
    MDH_HIP(hipMemsetAsync(done, 0, (size_t)N, st));
    MDH_HIP(hipMemsetAsync(removed, 0, (size_t)N, st));
    const dim3 grid(grid_for(N, 256)), block(256);
    for (int round = 0; round < 100000; ++round) {
        MDH_HIP(hipMemsetAsync(owner, 0x7f, sizeof(int) * (size_t)N, st));
        MDH_HIP(hipMemsetAsync(remaining, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_ovg_publish, grid, block, 0, st, dv, dnn, N, M, done, owner);
        if (b.tri)
            hipLaunchKernelGGL(k_ovg_sweep<true>, grid, block, 0, st, dx, dy, dz, b, dv, dnn, N, M, dt, dg, rc_metal_metal, rc_cc, rc_metal_c, done,
                               owner, removed, remaining);
        else
            hipLaunchKernelGGL(k_ovg_sweep<false>, grid, block, 0, st, dx, dy, dz, b, dv, dnn, N, M, dt, dg, rc_metal_metal, rc_cc, rc_metal_c, done,
                               owner, removed, remaining);
        int left = 0;
        MDH_HIP(hipMemcpyAsync(&left, remaining, sizeof(int), hipMemcpyDeviceToHost, st));
        MDH_HIP(hipStreamSynchronize(st));
        if (left == 0)
            break;
    }
    hipLaunchKernelGGL(k_ovg_keep, grid, block, 0, st, removed, N, dkeep);
    MDH_HIP(hipGetLastError());
    return sc.finish(space);
}

MDH_WARM_UNIT(polycrystal)
