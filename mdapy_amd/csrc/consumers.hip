// consumers.hip — per-atom analyses that only read an existing neighbor list, gfx950.
//
//   k_aja      Ackland-Jones analysis        src/ackland_jones_analysis.cpp:9-172
//   k_cnp      common neighbour parameter    src/common_neighbor_parameter.cpp:10-137
//   k_entropy  pair-entropy fingerprint      src/structure_entropy.cpp:9-108
//
// One thread per atom.  All three are bound by the row gather (M x 12 B per atom, plus the neighbours' rows for CNP);
// the arithmetic follows the reference's operation order, so AJA labels are exact and CNP / entropy agree to rounding
// of exp/log (device libm vs glibc).
#include "common.hpp"

namespace mdh {

template <bool TRI>
__global__ __launch_bounds__(256) void k_aja(const double *__restrict__ x, const double *__restrict__ y,
                                             const double *__restrict__ z, int64_t N, DBox b,
                                             const int *__restrict__ verlet, const double *__restrict__ dist, int64_t M,
                                             int *__restrict__ aja)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    const double *di = dist + i * M;
    const int *vi = verlet + i * M;
    double d14[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) d14[j] = di[j];
    double r0 = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) r0 += d14[j] * d14[j];
    r0 /= 6.0;
    const double c145 = 1.45 * r0, c155 = 1.55 * r0;
    int n0 = 0, n1 = 0;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const double r2 = d14[j] * d14[j];
        if (r2 < c155) { ++n1; if (r2 < c145) ++n0; }
    }
    // rows are sorted by distance, so the n0 bonds below 1.45 r0^2 are the first n0 entries (as the reference assumes)
    double rx[14], ry[14], rz[14];
    const double xi = x[i], yi = y[i], zi = z[i];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        if (j < n0) {
            const int q = vi[j];
            double ax = x[q] - xi, ay = y[q] - yi, az = z[q] - zi;
            pbc<TRI>(b, ax, ay, az);
            rx[j] = ax; ry[j] = ay; rz[j] = az;
        } else {
            rx[j] = ry[j] = rz[j] = 0.0;
        }
    }
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
#pragma unroll
    for (int j = 0; j < 14; ++j)
#pragma unroll
        for (int k = j + 1; k < 14; ++k)
            if (k < n0) {
                const double c = (rx[j] * rx[k] + ry[j] * ry[k] + rz[j] * rz[k]) / (d14[j] * d14[k]);
                if (c < -0.945) ++a0; else if (c < -0.915) ++a1; else if (c < -0.755) ++a2; else if (c < -0.195) ++a3;
                else if (c < 0.195) ++a4; else if (c < 0.245) ++a5; else if (c < 0.795) ++a6; else ++a7;
            }
    const double s_cp = fabs(1.0 - a6 / 24.0);
    const int s56m4 = a5 + a6 - a4;
    double s_bcc = s_cp + 1.0;
    if (s56m4 != 0) s_bcc = 0.35 * a4 / (double)s56m4;
    double s_fcc = 0.61 * (abs(a0 + a1 - 6) + a2) / 6.0;
    double s_hcp = (fabs(a0 - 3.0) + abs(a0 + a1 + a2 + a3 - 9)) / 12.0;
    if (a0 == 7) s_bcc = 0.0; else if (a0 == 6) s_fcc = 0.0; else if (a0 <= 3) s_hcp = 0.0;
    int t;
    if (a7 > 0) t = 0;
    else if (a4 < 3) t = (n1 > 13 || n1 < 11) ? 0 : 4;
    else if (s_bcc <= s_cp) t = n1 < 11 ? 0 : 3;
    else if (n1 > 12 || n1 < 11) t = 0;
    else t = s_fcc < s_hcp ? 1 : 2;
    aja[i] = t;
}

template <bool TRI>
__global__ __launch_bounds__(128) void k_cnp(const double *__restrict__ x, const double *__restrict__ y,
                                             const double *__restrict__ z, int64_t N, DBox b,
                                             const int *__restrict__ verlet, const double *__restrict__ dist,
                                             const int *__restrict__ nn, int64_t M, double rc, double *__restrict__ cnp)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    const int ni = nn[i];
    const int *vi = verlet + i * M;
    const double *di = dist + i * M;
    const double xi = x[i], yi = y[i], zi = z[i];
    int cnt = 0;
    double acc = 0.0;
    for (int m = 0; m < ni; ++m) {
        if (!(di[m] <= rc))
            continue;
        const int j = vi[m];
        ++cnt;
        const double xj = x[j], yj = y[j], zj = z[j];
        double rx = 0, ry = 0, rz = 0;
        const int nj = nn[j];
        const int *vj = verlet + (int64_t)j * M;
        const double *dj = dist + (int64_t)j * M;
        for (int s = 0; s < nj; ++s) {
            const int k = vj[s];
            for (int h = 0; h < ni; ++h)
                if (k == vi[h]) { // first match only (:83-120)
                    if (dj[s] <= rc && di[h] <= rc) {
                        const double xk = x[k], yk = y[k], zk = z[k];
                        double ax = xi - xk, ay = yi - yk, az = zi - zk;
                        double bx = xj - xk, by = yj - yk, bz = zj - zk;
                        pbc<TRI>(b, ax, ay, az);
                        pbc<TRI>(b, bx, by, bz);
                        rx += ax + bx; ry += ay + by; rz += az + bz;
                    }
                    break;
                }
        }
        acc += rx * rx + ry * ry + rz * rz;
    }
    cnp[i] = cnt > 0 ? acc / cnt : 1000.0;
}

static constexpr int ENT_MAXBINS = 512;

// the per-bin tables of the reference (:27-38: r_j = j*step, r_j^2, r_j^2*factor with entry 0 := entry 1) are single
// IEEE multiplications, recomputed here with identical results
__global__ __launch_bounds__(128) void k_entropy(const double *__restrict__ dist, const int *__restrict__ nn, int64_t N,
                                                 int64_t M, double rc, double sigma, int use_local, double gd, int nbins,
                                                 double step, double factor, double *__restrict__ entropy)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    const double PI = 3.14159265358979323846;
    const double s2 = sigma * sigma, lvol = 4. / 3. * PI * rc * rc * rc;
    const double *di = dist + i * M;
    const int n = nn[i];
    int nin = 0;
    for (int k = 0; k < n; ++k)
        nin += di[k] <= rc ? 1 : 0;
    double density = gd, fac = 1.0;
    if (use_local) {
        density = nin / lvol;
        fac = gd / density;
    }
    double prev = 0.0, sum = 0.0;
    for (int j = 0; j < nbins; ++j) {
        double g = 0.0;
        const double r = j * step, r2 = r * r;
        const double p = j == 0 ? (step * step) * factor : r2 * factor;
        for (int k = 0; k < n; ++k) {
            const double d = di[k];
            if (d <= rc) {
                const double dl = r - d;
                g += exp(-(dl * dl) / (2.0 * s2)) / p;
            }
        }
        if (use_local) g *= fac;
        const double v = g >= 1e-10 ? (g * log(g) - g + 1.0) * r2 : r2;
        if (j > 0) sum += prev + v;
        prev = v;
    }
    entropy[i] = -PI * density * sum * sigma;
}

} // namespace mdh

using namespace mdh;

extern "C" int mdh_aja(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                       const double *origin3, const int *boundary3, const int *verlet, const double *dist, int64_t M,
                       int *aja, int space, void *stream)
{
    if (N < 0 || M < 14) {
        set_error("mdh_aja: the neighbor list needs at least 14 distance-sorted columns");
        return MDH_ERR_ARG;
    }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const double *dd = sc.stage_in(dist, (size_t)(N * M), space);
    int *da = sc.stage(aja, (size_t)N, space, false, true);
    if (sc.failed())
        return sc.error();
    ProfRange pr("k_aja", sc.stream());
    if (b.tri)
        hipLaunchKernelGGL(k_aja<true>, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), dx, dy, dz, N, b, dv, dd, M, da);
    else
        hipLaunchKernelGGL(k_aja<false>, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), dx, dy, dz, N, b, dv, dd, M, da);
    return sc.finish(space);
}

extern "C" int mdh_cnp(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                       const double *origin3, const int *boundary3, const int *verlet, const double *dist, const int *nn,
                       int64_t M, double *cnp, double rc, int space, void *stream)
{
    if (N < 0 || M <= 0) {
        set_error("mdh_cnp: empty neighbor list");
        return MDH_ERR_ARG;
    }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const double *dd = sc.stage_in(dist, (size_t)(N * M), space);
    const int *dn = sc.stage_in(nn, (size_t)N, space);
    double *dc = sc.stage(cnp, (size_t)N, space, false, true);
    if (sc.failed())
        return sc.error();
    ProfRange pr("k_cnp", sc.stream());
    if (b.tri)
        hipLaunchKernelGGL(k_cnp<true>, dim3(grid_for(N, 128)), dim3(128), 0, sc.stream(), dx, dy, dz, N, b, dv, dd, dn, M, rc, dc);
    else
        hipLaunchKernelGGL(k_cnp<false>, dim3(grid_for(N, 128)), dim3(128), 0, sc.stream(), dx, dy, dz, N, b, dv, dd, dn, M, rc, dc);
    return sc.finish(space);
}

extern "C" int mdh_structure_entropy(double rc, double sigma, int use_local_density, double volume, const double *dist,
                                     const int *nn, int64_t N, int64_t M, double *entropy, int space, void *stream)
{
    const int nbins = (int)floor(rc / sigma) + 1; // :23
    if (N < 0 || M <= 0 || !(rc > 0) || !(sigma > 0) || !(volume > 0) || nbins < 2 || nbins > ENT_MAXBINS) {
        set_error("mdh_structure_entropy: need rc, sigma, volume > 0 and 2 <= floor(rc/sigma)+1 <= 512 bins");
        return MDH_ERR_ARG;
    }
    if (N == 0)
        return MDH_OK;
    const double PI = 3.14159265358979323846;
    const double gd = N / volume;
    const double step = rc / (nbins - 1);
    const double factor = 4. * PI * gd * sqrt(2. * PI * sigma * sigma);
    Scope sc(stream);
    const double *dd = sc.stage_in(dist, (size_t)(N * M), space);
    const int *dn = sc.stage_in(nn, (size_t)N, space);
    double *de = sc.stage(entropy, (size_t)N, space, false, true);
    if (sc.failed())
        return sc.error();
    ProfRange pr("k_entropy", sc.stream());
    hipLaunchKernelGGL(k_entropy, dim3(grid_for(N, 128)), dim3(128), 0, sc.stream(), dd, dn, N, M, rc, sigma, use_local_density ? 1 : 0,
                       gd, nbins, step, factor, de);
    return sc.finish(space);
}
