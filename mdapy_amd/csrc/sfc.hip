// sfc.hip — static structure factor by direct summation on gfx950.
//
// Replaces src/structure_factor.cpp:64-447 (StructureFactorDirect: total / cross) and :451-640
// (StructureFactorDirectPartial: all Ashcroft-Langreth partials at once).
//   F_a(k) = N^-1/2 sum_{i of species a} exp(i k.r_i)      S_ab(k) = Re( conj(F_a) F_b ), averaged over the k-points of a |k| bin
// The reciprocal-lattice points are enumerated on the host exactly as the two reference classes do (they differ:
// :105-200 bounds k_z analytically and tests q^2, :595-630 walks the full grid and tests |k|); the N x n_k phase sums —
// all of the cost — run on the device: a workgroup owns KB k-points, its threads stride over the atoms (each position
// load serves KB sincos pairs), per-species sums are reduced through LDS.  The |k| binning of the n_k results is a
// sequential host loop in the reference's order.
#include "common.hpp"
#include <cmath>
#include <vector>

namespace mdh {

static constexpr int SFC_KB = 4;     // k-points per workgroup
static constexpr int SFC_NT = 256;
static constexpr int SFC_MAXT = 16;  // species

__global__ __launch_bounds__(SFC_NT) void k_sfc_fk(const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ z, const int *__restrict__ type, int64_t n,
                                                   int ntype, const double *__restrict__ kp, int64_t nk, double norm,
                                                   double *__restrict__ fr, double *__restrict__ fi)
{
    const int64_t k0 = (int64_t)blockIdx.x * SFC_KB;
    __shared__ double acc[SFC_KB][SFC_MAXT][2];
    for (int q = threadIdx.x; q < SFC_KB * SFC_MAXT * 2; q += SFC_NT) (&acc[0][0][0])[q] = 0.0;
    __syncthreads();
    double kx[SFC_KB], ky[SFC_KB], kz[SFC_KB];
#pragma unroll
    for (int u = 0; u < SFC_KB; ++u) {
        const int64_t k = k0 + u < nk ? k0 + u : nk - 1;
        kx[u] = kp[3 * k]; ky[u] = kp[3 * k + 1]; kz[u] = kp[3 * k + 2];
    }
    if (ntype == 1) {
        double sr[SFC_KB] = {0, 0, 0, 0}, si[SFC_KB] = {0, 0, 0, 0};
        for (int64_t i = threadIdx.x; i < n; i += SFC_NT) {
            const double xi = x[i], yi = y[i], zi = z[i];
#pragma unroll
            for (int u = 0; u < SFC_KB; ++u) {
                double s, c;
                sincos(kx[u] * xi + ky[u] * yi + kz[u] * zi, &s, &c); // phase as in :222-224
                sr[u] += c; si[u] += s;
            }
        }
#pragma unroll
        for (int u = 0; u < SFC_KB; ++u) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { sr[u] += __shfl_xor(sr[u], d, 64); si[u] += __shfl_xor(si[u], d, 64); }
            if ((threadIdx.x & 63) == 0) { atomicAdd(&acc[u][0][0], sr[u]); atomicAdd(&acc[u][0][1], si[u]); }
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += SFC_NT) {
            const double xi = x[i], yi = y[i], zi = z[i];
            const int t = type[i];
#pragma unroll
            for (int u = 0; u < SFC_KB; ++u) {
                double s, c;
                sincos(kx[u] * xi + ky[u] * yi + kz[u] * zi, &s, &c);
                atomicAdd(&acc[u][t][0], c);
                atomicAdd(&acc[u][t][1], s);
            }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < SFC_KB * ntype; q += SFC_NT) {
        const int u = q / ntype, t = q % ntype;
        if (k0 + u < nk) {
            fr[(int64_t)t * nk + k0 + u] = acc[u][t][0] * norm;
            fi[(int64_t)t * nk + k0 + u] = acc[u][t][1] * norm;
        }
    }
}

struct V3 { double x, y, z; };
static inline double vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline double vnorm(V3 a) { return std::sqrt(vdot(a, a)); }
static const double TWO_PI = 2.0 * 3.14159265358979323846;

static bool reciprocal(const double *h, V3 *b) // :66-103
{
    const double *a = h, *bb = h + 3, *c = h + 6;
    const double vol = a[0] * (bb[1] * c[2] - bb[2] * c[1]) - a[1] * (bb[0] * c[2] - bb[2] * c[0]) + a[2] * (bb[0] * c[1] - bb[1] * c[0]);
    if (std::fabs(vol) < 1e-12)
        return false;
    b[0] = {(bb[1] * c[2] - bb[2] * c[1]) / vol * TWO_PI, (bb[2] * c[0] - bb[0] * c[2]) / vol * TWO_PI, (bb[0] * c[1] - bb[1] * c[0]) / vol * TWO_PI};
    b[1] = {(c[1] * a[2] - c[2] * a[1]) / vol * TWO_PI, (c[2] * a[0] - c[0] * a[2]) / vol * TWO_PI, (c[0] * a[1] - c[1] * a[0]) / vol * TWO_PI};
    b[2] = {(a[1] * bb[2] - a[2] * bb[1]) / vol * TWO_PI, (a[2] * bb[0] - a[0] * bb[2]) / vol * TWO_PI, (a[0] * bb[1] - a[1] * bb[0]) / vol * TWO_PI};
    return true;
}

static void kpoints_total(const V3 *b, double k_max, double k_min, std::vector<double> &kp) // :105-200
{
    const double q_max = k_max / TWO_PI, q_min = k_min / TWO_PI, q_max_sq = q_max * q_max, q_min_sq = q_min * q_min;
    const int nkx = (int)std::ceil(q_max / (vnorm(b[0]) / TWO_PI)), nky = (int)std::ceil(q_max / (vnorm(b[1]) / TWO_PI)),
              nkz = (int)std::ceil(q_max / (vnorm(b[2]) / TWO_PI));
    for (int i = 0; i < nkx; ++i) {
        const V3 kx{b[0].x * i, b[0].y * i, b[0].z * i};
        for (int j = 0; j < nky; ++j) {
            const V3 kxy{kx.x + b[1].x * j, kx.y + b[1].y * j, kx.z + b[1].z * j};
            const double ca = vdot(b[2], b[2]), cb = -2.0 * vdot(kxy, b[2]);
            const double cmin = vdot(kxy, kxy) - k_min * k_min, cmax = vdot(kxy, kxy) - k_max * k_max;
            const double b2a = cb / (2.0 * ca), dmin = b2a * b2a - cmin / ca, dmax = b2a * b2a - cmax / ca;
            if (dmax < 0)
                continue;
            const double zmin = dmin < 0 ? 0.0 : -b2a + std::sqrt(dmin), zmax = -b2a + std::sqrt(dmax);
            int kz0 = (int)std::floor(zmin), kz1 = (int)std::ceil(zmax);
            kz0 = kz0 < 0 ? 0 : kz0;
            kz1 = kz1 > nkz - 1 ? nkz - 1 : kz1;
            for (int k = kz0; k <= kz1; ++k) {
                const V3 kv{kxy.x + b[2].x * k, kxy.y + b[2].y * k, kxy.z + b[2].z * k};
                const double qd = vdot(kv, kv) / (TWO_PI * TWO_PI);
                if (qd <= q_max_sq && qd >= q_min_sq) { kp.push_back(kv.x); kp.push_back(kv.y); kp.push_back(kv.z); }
            }
        }
    }
}

static void kpoints_partial(const V3 *b, double k_max, double k_min, std::vector<double> &kp) // :595-630
{
    const double q_max = k_max / TWO_PI;
    const int nkx = (int)std::ceil(q_max / (vnorm(b[0]) / TWO_PI)), nky = (int)std::ceil(q_max / (vnorm(b[1]) / TWO_PI)),
              nkz = (int)std::ceil(q_max / (vnorm(b[2]) / TWO_PI));
    for (int i = 0; i < nkx; ++i) {
        const V3 kx{b[0].x * i, b[0].y * i, b[0].z * i};
        for (int j = 0; j < nky; ++j) {
            const V3 kxy{kx.x + b[1].x * j, kx.y + b[1].y * j, kx.z + b[1].z * j};
            for (int k = 0; k < nkz; ++k) {
                const V3 kv{kxy.x + b[2].x * k, kxy.y + b[2].y * k, kxy.z + b[2].z * k};
                const double mag = vnorm(kv);
                if (mag > k_min && mag <= k_max) { kp.push_back(kv.x); kp.push_back(kv.y); kp.push_back(kv.z); }
            }
        }
    }
}

static int bin_of(double k, double k_min, double k_max, int bins) // :283-291
{
    if (k < k_min || k >= k_max)
        return -1;
    const int b = (int)((k - k_min) / (k_max - k_min) * bins);
    return b < bins - 1 ? b : bins - 1;
}

// F (ntype, nk) of one point set -> host vectors
static int phase_sums(Scope &sc, const double *dx, const double *dy, const double *dz, const int *dtype, int64_t n, int ntype,
                      const double *dkp, int64_t nk, double norm, std::vector<double> &fr, std::vector<double> &fi)
{
    double *dfr = sc.alloc_n<double>((size_t)ntype * nk), *dfi = sc.alloc_n<double>((size_t)ntype * nk);
    if (sc.failed())
        return sc.error();
    {
        ProfRange pr("k_sfc_fk", sc.stream());
        hipLaunchKernelGGL(k_sfc_fk, dim3((unsigned)((nk + SFC_KB - 1) / SFC_KB)), dim3(SFC_NT), 0, sc.stream(), dx, dy, dz, dtype, n, ntype,
                           dkp, nk, norm, dfr, dfi);
    }
    fr.resize((size_t)ntype * nk);
    fi.resize((size_t)ntype * nk);
    MDH_HIP(hipMemcpyAsync(fr.data(), dfr, sizeof(double) * fr.size(), hipMemcpyDeviceToHost, sc.stream()));
    MDH_HIP(hipMemcpyAsync(fi.data(), dfi, sizeof(double) * fi.size(), hipMemcpyDeviceToHost, sc.stream()));
    MDH_HIP(hipStreamSynchronize(sc.stream()));
    return MDH_OK;
}

} // namespace mdh

using namespace mdh;

// replaces _sfc.compute_sfc_direct (src/structure_factor.cpp:654-680).  sf (bins) is a HOST array whatever `space` says
// about the positions; qx/qy/qz may be NULL (total structure factor); n_total = 0 means n.
extern "C" int mdh_sfc_direct(const double *x, const double *y, const double *z, int64_t n, const double *box9, double *sf_host,
                              int bins, double k_max, double k_min, const double *qx, const double *qy, const double *qz,
                              int64_t nq, unsigned n_total, int space, void *stream)
{
    if (bins <= 0 || !(k_max > 0) || k_min < 0 || !(k_max > k_min) || n <= 0 || (qx && n_total == 0)) {
        set_error("mdh_sfc_direct: need bins > 0, 0 <= k_min < k_max, atoms, and N_total with query points");
        return MDH_ERR_ARG;
    }
    V3 b[3];
    if (!reciprocal(box9, b)) { set_error("Box volume is too small or zero"); return MDH_ERR_BOX; }
    std::vector<double> kp;
    kpoints_total(b, k_max, k_min, kp);
    const int64_t nk = (int64_t)kp.size() / 3;
    if (nk == 0) { set_error("No k-points generated. Check k_min and k_max values."); return MDH_ERR_ARG; }
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)n, space), *dy = sc.stage_in(y, (size_t)n, space), *dz = sc.stage_in(z, (size_t)n, space);
    const double *dkp = sc.stage_in(kp.data(), kp.size(), MDH_HOST);
    if (sc.failed())
        return sc.error();
    if (n_total == 0) n_total = (unsigned)n;
    const double norm = 1.0 / std::sqrt((double)n_total);
    std::vector<double> ar, ai, br, bi;
    MDH_TRY(phase_sums(sc, dx, dy, dz, nullptr, n, 1, dkp, nk, norm, ar, ai));
    if (qx) {
        const double *dqx = sc.stage_in(qx, (size_t)nq, space), *dqy = sc.stage_in(qy, (size_t)nq, space), *dqz = sc.stage_in(qz, (size_t)nq, space);
        if (sc.failed())
            return sc.error();
        MDH_TRY(phase_sums(sc, dqx, dqy, dqz, nullptr, nq, 1, dkp, nk, norm, br, bi));
    }
    std::vector<unsigned> cnt((size_t)bins, 0u);
    for (int q = 0; q < bins; ++q) sf_host[q] = 0.0;
    for (int64_t k = 0; k < nk; ++k) { // :400-427
        const double mag = std::sqrt(kp[3 * k] * kp[3 * k] + kp[3 * k + 1] * kp[3 * k + 1] + kp[3 * k + 2] * kp[3 * k + 2]);
        const int q = bin_of(mag, k_min, k_max, bins);
        if (q < 0)
            continue;
        const double s = qx ? ar[k] * br[k] + ai[k] * bi[k] : ar[k] * ar[k] + ai[k] * ai[k];
        sf_host[q] += s;
        cnt[q]++;
    }
    for (int q = 0; q < bins; ++q) sf_host[q] = cnt[q] ? sf_host[q] / cnt[q] : std::nan("");
    return MDH_OK;
}

// replaces _sfc.compute_sfc_direct_partial (:682-705); out_host (ntype, ntype, bins) HOST array
extern "C" int mdh_sfc_direct_partial(const double *x, const double *y, const double *z, const int *type, int ntype, int64_t n,
                                      const double *box9, double *out_host, int bins, double k_max, double k_min, int space,
                                      void *stream)
{
    if (bins <= 0 || k_min < 0 || !(k_max > k_min) || n <= 0 || ntype <= 0 || ntype > SFC_MAXT) {
        set_error("mdh_sfc_direct_partial: need bins > 0, 0 <= k_min < k_max, atoms and 1..16 species");
        return MDH_ERR_ARG;
    }
    V3 b[3];
    if (!reciprocal(box9, b)) { set_error("Box volume is too small or zero"); return MDH_ERR_BOX; }
    std::vector<double> kp;
    kpoints_partial(b, k_max, k_min, kp);
    const int64_t nk = (int64_t)kp.size() / 3;
    if (nk == 0) { set_error("No k-points generated"); return MDH_ERR_ARG; }
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)n, space), *dy = sc.stage_in(y, (size_t)n, space), *dz = sc.stage_in(z, (size_t)n, space);
    const int *dt = sc.stage_in(type, (size_t)n, space);
    const double *dkp = sc.stage_in(kp.data(), kp.size(), MDH_HOST);
    if (sc.failed())
        return sc.error();
    std::vector<double> fr, fi;
    MDH_TRY(phase_sums(sc, dx, dy, dz, dt, n, ntype, dkp, nk, 1.0 / std::sqrt((double)n), fr, fi));
    std::vector<unsigned> cnt((size_t)bins, 0u);
    std::vector<int> bin((size_t)nk);
    for (int64_t k = 0; k < nk; ++k) {
        const double mag = std::sqrt(kp[3 * k] * kp[3 * k] + kp[3 * k + 1] * kp[3 * k + 1] + kp[3 * k + 2] * kp[3 * k + 2]);
        bin[k] = bin_of(mag, k_min, k_max, bins);
        if (bin[k] >= 0) cnt[bin[k]]++;
    }
    std::vector<double> loc((size_t)bins);
    for (int a = 0; a < ntype; ++a)
        for (int c = a; c < ntype; ++c) { // :520-556
            std::fill(loc.begin(), loc.end(), 0.0);
            for (int64_t k = 0; k < nk; ++k)
                if (bin[k] >= 0)
                    loc[bin[k]] += fr[(size_t)a * nk + k] * fr[(size_t)c * nk + k] + fi[(size_t)a * nk + k] * fi[(size_t)c * nk + k];
            for (int q = 0; q < bins; ++q) {
                const double v = cnt[q] ? loc[q] / cnt[q] : std::nan("");
                out_host[((size_t)a * ntype + c) * bins + q] = v;
                out_host[((size_t)c * ntype + a) * bins + q] = v;
            }
        }
    return MDH_OK;
}

MDH_WARM_UNIT(sfc)
