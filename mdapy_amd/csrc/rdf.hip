// rdf.hip — radial distribution function histograms on gfx950.
//
// Replaces src/radial_distribution_function.cpp: _rdf :22-54, _rdf_single_species
// :56-85 (serial in the reference) and _rdf_streaming :143-317.
//
// Pair counts are accumulated as INTEGERS (u32 in LDS per workgroup, u64 in
// HBM) and converted to f64 once when they are added to the caller's g, so the
// result is independent of the summation order and bit-identical to the
// reference's "+= 1.0" on doubles (exact below 2^53).
#include "common.hpp"
#include "grid.hpp"

namespace mdh {

static constexpr int RDF_LDS_BINS = 8192; // u32 bins kept in LDS (32 KiB); larger histograms go straight to HBM

__device__ __forceinline__ void hist_add(unsigned *lds, unsigned long long *glob, bool use_lds, int64_t bin)
{
    if (use_lds) atomicAdd(&lds[bin], 1u);
    else atomicAdd(&glob[bin], 1ull);
}

__device__ __forceinline__ void hist_flush(unsigned *lds, unsigned long long *glob, bool use_lds, int64_t hsize)
{
    if (!use_lds)
        return;
    __syncthreads();
    for (int64_t q = threadIdx.x; q < hsize; q += blockDim.x) {
        const unsigned v = lds[q];
        if (v) atomicAdd(&glob[q], (unsigned long long)v);
    }
}

// ---- streaming, cell-list path (:174-264)
template <bool TRI>
__global__ __launch_bounds__(256) void k_rdf_cells(const double *__restrict__ xs, const double *__restrict__ ys,
                                                   const double *__restrict__ zs, const int *__restrict__ order,
                                                   const int *__restrict__ cell_start,
                                                   const int *__restrict__ type, int64_t N, DBox b, Grid g,
                                                   double rc, int nbin, int ntype,
                                                   unsigned long long *__restrict__ hist)
{
    __shared__ unsigned lds[RDF_LDS_BINS];
    const int64_t hsize = (int64_t)ntype * ntype * nbin;
    const bool use_lds = hsize <= RDF_LDS_BINS;
    if (use_lds)
        for (int64_t q = threadIdx.x; q < hsize; q += blockDim.x) lds[q] = 0u;
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < N) {
        const int i = order[p];
        double xi = xs[p], yi = ys[p], zi = zs[p];
        if (b.anypbc) // :213-214
            wrap<TRI>(b, xi, yi, zi);
        int c0, c1, c2;
        cell_coords<TRI>(b, g, xi, yi, zi, c0, c1, c2);
        const int it = type[i];
        const double dr = rc / nbin, rcsq = rc * rc; // :158-159
        for (int da = -1; da <= 1; ++da) {          // :223-235: open axes are NOT wrapped
            const int a = b.pbc[0] ? pmod(c0 + da, g.nc[0]) : c0 + da;
            if (a < 0 || a >= g.nc[0]) continue;
            for (int db = -1; db <= 1; ++db) {
                const int bb = b.pbc[1] ? pmod(c1 + db, g.nc[1]) : c1 + db;
                if (bb < 0 || bb >= g.nc[1]) continue;
                for (int dc = -1; dc <= 1; ++dc) {
                    const int cc = b.pbc[2] ? pmod(c2 + dc, g.nc[2]) : c2 + dc;
                    if (cc < 0 || cc >= g.nc[2]) continue;
                    const int64_t cell = ((int64_t)a * g.nc[1] + bb) * g.nc[2] + cc;
                    const int s = cell_start[cell], e = cell_start[cell + 1];
                    for (int q = s; q < e; ++q) {
                        const int j = order[q];
                        if (j == i) continue;
                        double dx = xs[q] - xi, dy = ys[q] - yi, dz = zs[q] - zi;
                        pbc<TRI>(b, dx, dy, dz);
                        const double r2 = dx * dx + dy * dy + dz * dz;
                        if (r2 < rcsq) { // strict, :246
                            const int k = (int)(sqrt(r2) / dr);
                            if (k < nbin)
                                hist_add(lds, hist, use_lds, ((int64_t)it * ntype + type[j]) * nbin + k);
                        }
                    }
                }
            }
        }
    }
    hist_flush(lds, hist, use_lds, hsize);
}

// ---- streaming, all-pairs fallback (:266-305): one thread per atom i, j tiled through LDS
template <bool TRI>
__global__ __launch_bounds__(256) void k_rdf_allpairs(const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ z, const int *__restrict__ type,
                                                      int64_t N, DBox b, double rc, int nbin, int ntype,
                                                      unsigned long long *__restrict__ hist)
{
    __shared__ unsigned lds[RDF_LDS_BINS];
    __shared__ double tx[256], ty[256], tz[256];
    __shared__ int tt[256];
    const int64_t hsize = (int64_t)ntype * ntype * nbin;
    const bool use_lds = hsize <= RDF_LDS_BINS;
    if (use_lds)
        for (int64_t q = threadIdx.x; q < hsize; q += blockDim.x) lds[q] = 0u;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double xi = 0, yi = 0, zi = 0;
    int it = 0;
    if (i < N) {
        xi = x[i]; yi = y[i]; zi = z[i];
        if (b.anypbc) wrap<TRI>(b, xi, yi, zi);
        it = type[i];
    }
    const double dr = rc / nbin, rcsq = rc * rc;
    for (int64_t base = 0; base < N; base += 256) {
        __syncthreads();
        const int64_t j0 = base + threadIdx.x;
        if (j0 < N) { tx[threadIdx.x] = x[j0]; ty[threadIdx.x] = y[j0]; tz[threadIdx.x] = z[j0]; tt[threadIdx.x] = type[j0]; }
        __syncthreads();
        const int lim = (int)((N - base) < 256 ? (N - base) : 256);
        if (i < N)
            for (int t = 0; t < lim; ++t) {
                if (base + t == i) continue;
                double dx = tx[t] - xi, dy = ty[t] - yi, dz = tz[t] - zi;
                pbc<TRI>(b, dx, dy, dz);
                const double r2 = dx * dx + dy * dy + dz * dz;
                if (r2 < rcsq) {
                    const int k = (int)(sqrt(r2) / dr);
                    if (k < nbin)
                        hist_add(lds, hist, use_lds, ((int64_t)it * ntype + tt[t]) * nbin + k);
                }
            }
    }
    hist_flush(lds, hist, use_lds, hsize);
}

// ---- binning of an existing list: _rdf :22-54 (single==0) and _rdf_single_species :56-85 (single==1)
__global__ __launch_bounds__(256) void k_rdf_list(const int *__restrict__ verlet, const double *__restrict__ dist,
                                                  const int *__restrict__ nn, const int *__restrict__ type,
                                                  int64_t N, int64_t M, double rc, int nbin, int ntype, int single,
                                                  unsigned long long *__restrict__ hist)
{
    __shared__ unsigned lds[RDF_LDS_BINS];
    const int64_t hsize = single ? nbin : (int64_t)ntype * ntype * nbin;
    const bool use_lds = hsize <= RDF_LDS_BINS;
    if (use_lds)
        for (int64_t q = threadIdx.x; q < hsize; q += blockDim.x) lds[q] = 0u;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) {
        const double dr = rc / nbin;
        const int n = nn[i];
        const int it = single ? 0 : type[i];
        for (int q = 0; q < n; ++q) {
            const double d = dist[i * M + q];
            const int j = safe_id(verlet[i * M + q], i, N);
            const int k = (int)(d / dr);
            if (!(d < rc) || k >= nbin || k < 0) // k == nbin can only arise from rounding at d -> rc (the reference would write out of bounds there)
                continue;
            if (single) {
                if (j > i) hist_add(lds, hist, use_lds, k);
            } else {
                hist_add(lds, hist, use_lds, ((int64_t)it * ntype + type[j]) * nbin + k);
            }
        }
    }
    hist_flush(lds, hist, use_lds, hsize);
}

// g[q] += scale * count[q]   (:308-316 accumulate into the caller's array; single species adds 2.0 per pair, :81)
__global__ __launch_bounds__(256) void k_hist_to_g(const unsigned long long *__restrict__ hist, double *__restrict__ g,
                                                   int64_t hsize, double scale)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < hsize) g[q] += scale * (double)hist[q];
}

} // namespace mdh

using namespace mdh;

extern "C" {

int mdh_rdf_streaming(const double *x, const double *y, const double *z, const int *type, int64_t N,
                      const double *box9, const double *origin3, const int *boundary3, double *g, int ntype,
                      double rc, int nbin, int space, void *stream)
{
    if (N < 0 || ntype <= 0 || nbin <= 0 || !(rc > 0)) { set_error("mdh_rdf_streaming: invalid argument"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    const int64_t hsize = (int64_t)ntype * ntype * nbin;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    const int *dt = sc.stage_in(type, (size_t)N, space);
    double *dg = sc.stage(g, (size_t)hsize, space, true, true);
    unsigned long long *hist = sc.alloc_n<unsigned long long>((size_t)hsize);
    if (sc.failed())
        return sc.error();
    MDH_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned long long) * (size_t)hsize, st));
    CellGrid cg;
    bool use_cells = true; // :163-172
    double tot = 1.0;
    for (int d = 0; d < 3; ++d) {
        double f = std::floor(b.thick[d] / rc);
        int n = (f < 1.0 || !(f == f)) ? 1 : (f > 2.0e9 ? 2000000000 : (int)f);
        cg.g.nc[d] = n;
        tot *= n;
        if (b.pbc[d] && n < 3) use_cells = false;
    }
    if (use_cells && tot > 2147483000.0) { set_error("mdh_rdf_streaming: cell grid too large"); return MDH_ERR_ARG; }
    if (use_cells) {
        cg.g.ncell = (int64_t)cg.g.nc[0] * cg.g.nc[1] * cg.g.nc[2];
        cg.g.rc_inv = 1.0 / rc;
        cg.g.mode = 1;
        MDH_TRY(build_cell_grid(sc, dx, dy, dz, N, b, true, false, cg));
        if (b.tri)
            hipLaunchKernelGGL(k_rdf_cells<true>, dim3(grid_for(N, 256)), dim3(256), 0, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, dt, N, b, cg.g, rc, nbin, ntype, hist);
        else
            hipLaunchKernelGGL(k_rdf_cells<false>, dim3(grid_for(N, 256)), dim3(256), 0, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, dt, N, b, cg.g, rc, nbin, ntype, hist);
    } else {
        if (b.tri)
            hipLaunchKernelGGL(k_rdf_allpairs<true>, dim3(grid_for(N, 256)), dim3(256), 0, st, dx, dy, dz, dt, N, b, rc, nbin, ntype, hist);
        else
            hipLaunchKernelGGL(k_rdf_allpairs<false>, dim3(grid_for(N, 256)), dim3(256), 0, st, dx, dy, dz, dt, N, b, rc, nbin, ntype, hist);
    }
    hipLaunchKernelGGL(k_hist_to_g, dim3(grid_for(hsize, 256)), dim3(256), 0, st, hist, dg, hsize, 1.0);
    return sc.finish(space);
}

static int rdf_from_list(const int *verlet, const double *dist, const int *nn, const int *type, int64_t N, int64_t M,
                         double *g, int ntype, double rc, int nbin, int single, int space, void *stream)
{
    if (N < 0 || M <= 0 || nbin <= 0 || ntype <= 0) { set_error("mdh_rdf: invalid argument"); return MDH_ERR_ARG; }
    if (N == 0)
        return MDH_OK;
    const int64_t hsize = single ? nbin : (int64_t)ntype * ntype * nbin;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const double *dd = sc.stage_in(dist, (size_t)(N * M), space);
    const int *dn = sc.stage_in(nn, (size_t)N, space);
    const int *dt = single ? nullptr : sc.stage_in(type, (size_t)N, space);
    double *dg = sc.stage(g, (size_t)hsize, space, true, true);
    unsigned long long *hist = sc.alloc_n<unsigned long long>((size_t)hsize);
    if (sc.failed())
        return sc.error();
    MDH_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned long long) * (size_t)hsize, st));
    hipLaunchKernelGGL(k_rdf_list, dim3(grid_for(N, 256)), dim3(256), 0, st, dv, dd, dn, dt, N, M, rc, nbin, ntype, single, hist);
    hipLaunchKernelGGL(k_hist_to_g, dim3(grid_for(hsize, 256)), dim3(256), 0, st, hist, dg, hsize, single ? 2.0 : 1.0);
    return sc.finish(space);
}

int mdh_rdf(const int *verlet, const double *dist, const int *nn, const int *type, int64_t N, int64_t M, double *g,
            int ntype, double rc, int nbin, int space, void *stream)
{
    return rdf_from_list(verlet, dist, nn, type, N, M, g, ntype, rc, nbin, 0, space, stream);
}

int mdh_rdf_single_species(const int *verlet, const double *dist, const int *nn, int64_t N, int64_t M, double *g,
                           double rc, int nbin, int space, void *stream)
{
    return rdf_from_list(verlet, dist, nn, nullptr, N, M, g, 1, rc, nbin, 1, space, stream);
}
}
