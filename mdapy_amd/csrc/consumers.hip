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
#include <cstring>

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
    // all fourteen gathers first (a slot past n0 reads a valid atom and is zeroed below), the minimum images afterwards: in one
    // loop body each gather waits behind the branches of the previous neighbour's pbc
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int q = safe_id(vi[j], i, N);
        rx[j] = x[q] - xi; ry[j] = y[q] - yi; rz[j] = z[q] - zi;
    }
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        if (j < n0) pbc<TRI>(b, rx[j], ry[j], rz[j]);
        else rx[j] = ry[j] = rz[j] = 0.0;
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

// one neighbour j of atom i (common_neighbor_parameter.cpp:83-120), rows read entry by entry: the general form
template <bool TRI>
__device__ __forceinline__ double cnp_pair_rows(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                int64_t N, const DBox &b, const int *__restrict__ verlet, const double *__restrict__ dist,
                                                const int *__restrict__ nn, int64_t M, double rc, int64_t i, int j, int ni,
                                                const int *__restrict__ vi, const double *__restrict__ di, double xi, double yi, double zi)
{
    const double xj = x[j], yj = y[j], zj = z[j];
    double rx = 0, ry = 0, rz = 0;
    const int nj = nn[j];
    const int *vj = verlet + (int64_t)j * M;
    const double *dj = dist + (int64_t)j * M;
    for (int s = 0; s < nj; ++s) {
        const int k = safe_id(vj[s], i, N);
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
    return rx * rx + ry * ry + rz * rz;
}

// Rows of up to 16 entries (every cutoff list of a close-packed or bcc crystal) are handled from registers: the atom's own
// ids are loaded once and compared with an unrolled chain (the general form re-reads them ni * nj times per neighbour), a
// neighbour's row arrives as 32 loads in flight, and the positions of the common neighbours of a pair — four of them in
// fcc — are gathered together instead of one dependent load chain per match.  Same sums in the same order as the general
// form, which still takes wider rows (per neighbour) — 20.8 -> see DESIGN.md 3 for the measured effect.
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
    constexpr int W = 16;
    const bool narrow = ni <= W && ni <= M;
    int own[W];          // the atom's row (slots beyond ni: a value no id takes)
    unsigned own_ok = 0; // bit h: di[h] <= rc
    if (narrow) {
#pragma unroll
        for (int h = 0; h < W; ++h) {
            const int hh = min(h, (int)M - 1);
            const int v = vi[hh];
            const double d = di[hh];
            own[h] = h < ni ? v : -0x7fffffff;
            own_ok |= (h < ni && d <= rc) ? 1u << h : 0u;
        }
    }
    for (int m = 0; m < ni; ++m) {
        if (!(narrow ? (own_ok >> m & 1u) != 0 : di[m] <= rc))
            continue;
        int vim = narrow ? own[0] : vi[m];
        if (narrow) {
#pragma unroll
            for (int h = 1; h < W; ++h)
                if (m == h) vim = own[h];
        }
        const int j = safe_id(vim, i, N);
        ++cnt;
        const int nj = nn[j];
        if (!narrow || nj > W || nj > M) {
            acc += cnp_pair_rows<TRI>(x, y, z, N, b, verlet, dist, nn, M, rc, i, j, ni, vi, di, xi, yi, zi);
            continue;
        }
        const double xj = x[j], yj = y[j], zj = z[j];
        const int *vj = verlet + (int64_t)j * M;
        const double *dj = dist + (int64_t)j * M;
        int ks[W];
        unsigned take = 0; // bit s: entry s of j's row is a common neighbour that counts
#pragma unroll
        for (int s = 0; s < W; ++s) { // 32 loads in flight
            const int ss = min(s, (int)M - 1);
            ks[s] = safe_id(vj[ss], i, N);
            const double d = dj[ss];
            take |= (s < nj && d <= rc) ? 1u << s : 0u;
        }
#pragma unroll
        for (int s = 0; s < W; ++s) {
            unsigned hit = 0; // own slots holding this id; the first one decides (:83-120)
#pragma unroll
            for (int h = 0; h < W; ++h)
                hit |= ks[s] == own[h] ? 1u << h : 0u;
            const bool counts = hit != 0 && (own_ok >> __builtin_ctz(hit | 0x10000u) & 1u) != 0;
            if (!counts) take &= ~(1u << s);
        }
        double rx = 0, ry = 0, rz = 0;
#pragma unroll
        for (int half = 0; half < W; half += 8) {
            if ((take >> half & 0xffu) == 0)
                continue;
            double px[8], py[8], pz[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { // the positions of this half's common neighbours, requested together
                const int k = (take >> (half + u) & 1u) ? ks[half + u] : (int)i;
                px[u] = x[k]; py[u] = y[k]; pz[u] = z[k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (take >> (half + u) & 1u) {
                    double ax = xi - px[u], ay = yi - py[u], az = zi - pz[u];
                    double bx = xj - px[u], by = yj - py[u], bz = zj - pz[u];
                    pbc<TRI>(b, ax, ay, az);
                    pbc<TRI>(b, bx, by, bz);
                    rx += ax + bx; ry += ay + by; rz += az + bz;
                }
        }
        acc += rx * rx + ry * ry + rz * rz;
    }
    cnp[i] = cnt > 0 ? acc / cnt : 1000.0;
}

// A wide list (more than 16 columns: the analysis borrowed a list built for a larger cutoff) is narrowed first: every row's
// entries within rc, in their order, into a row of 16, and k_cnp then runs its register form on that.  The general form compared
// the 42 x 42 entries of every pair's rows straight from memory: 418 ms for 4 M atoms of fcc Cu behind build_neighbor(5.0, 50),
// against 8.5 ms on the list of the analysis' own cutoff.  What the narrowing must not change is which of an atom's entries a
// common neighbour is matched to (the first one with that id decides, within rc or not, :83-120): that only differs when an id
// sits twice in a row, i.e. in a box with a periodic width below twice the list's reach — flags[1] returns the largest listed
// distance (as bits) for the caller to check that, flags[0] a row that did not fit; either sends the call down the general form.
constexpr int CNP_NARROW = 16;
__global__ __launch_bounds__(64) void k_cnp_narrow(const int *__restrict__ verlet, const double *__restrict__ dist,
                                                   const int *__restrict__ nn, int64_t N, int64_t M, double rc,
                                                   int *__restrict__ nv, double *__restrict__ nd, int *__restrict__ ncount,
                                                   unsigned long long *__restrict__ flags)
{
    __shared__ int ids[ROW_CHUNK * 64];
    __shared__ double dst[ROW_CHUNK * 64];
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + threadIdx.x;
    const int n = i < N ? min(nn[i], (int)M) : 0;
    const int most = wave_max(n);
    int cnt = 0;
    double far = 0.0;
    for (int c0 = 0; c0 < most; c0 += ROW_CHUNK) {
        __syncthreads();
        stage_row_chunk<true>(verlet, dist, N, M, row0, c0, ids, dst);
        __syncthreads();
        for (int q = 0; q < ROW_CHUNK && c0 + q < n; ++q) {
            const double d = dst[q * 64 + threadIdx.x];
            far = d > far ? d : far;
            if (d <= rc) {
                if (cnt < CNP_NARROW) {
                    nv[i * CNP_NARROW + cnt] = ids[q * 64 + threadIdx.x];
                    nd[i * CNP_NARROW + cnt] = d;
                }
                ++cnt;
            }
        }
    }
    if (i < N) ncount[i] = min(cnt, CNP_NARROW);
    const bool over = cnt > CNP_NARROW;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) far = fmax(far, __shfl_xor(far, s, 64));
    if (__any(over) && threadIdx.x == 0) atomicMax(flags, 1ull);
    if (threadIdx.x == 0) atomicMax(flags + 1, (unsigned long long)__double_as_longlong(far)); // (distances are >= 0: their bits order as they do)
}

static constexpr int ENT_MAXBINS = 512;
static int g_entropy_variant = 0;

// the per-bin tables of the reference (:27-38: r_j = j*step, r_j^2, r_j^2*factor with entry 0 := entry 1) are single
// IEEE multiplications, recomputed here with identical results
// STAGED: the 64 rows of the workgroup are one contiguous piece of the list: loaded with coalesced reads into LDS (entry k
// of row t at [k * 65 + t]) and read from there nbins times.  Unstaged, every one of the nbins * n loads of a wave touches
// 64 different cache lines (a lane's row is M * 8 bytes away from its neighbour's).  Rows too wide for LDS: unstaged.
template <bool STAGED>
__global__ __launch_bounds__(64) void k_entropy(const double *__restrict__ dist, const int *__restrict__ nn, int64_t N,
                                                int64_t M, double rc, double sigma, int use_local, double gd, int nbins,
                                                double step, double factor, double *__restrict__ entropy)
{
    extern __shared__ double ent_rows[];
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + threadIdx.x;
    if (STAGED) {
        const int64_t total = (N - row0 < 64 ? N - row0 : 64) * M;
        for (int64_t e = threadIdx.x; e < total; e += 64) {
            const int r = (int)(e / M), c = (int)(e - (int64_t)r * M);
            ent_rows[c * 65 + r] = dist[row0 * M + e];
        }
        __syncthreads();
    }
    if (i >= N)
        return;
    const double PI = 3.14159265358979323846;
    const double c2 = -1.0 / (2.0 * sigma * sigma), lvol = 4. / 3. * PI * rc * rc * rc;
    const double *di = STAGED ? ent_rows + threadIdx.x : dist + i * M;
    const int stride = STAGED ? 65 : 1;
    const int n = STAGED ? min(nn[i], (int)M) : nn[i]; // (a count beyond the row width is the caller's error: the staged form stays inside the row)
    int nin = 0;
    for (int k = 0; k < n; ++k)
        nin += di[k * stride] <= rc ? 1 : 0;
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
        for (int k = 0; k < n; ++k) { // (the reference divides every term by 2 sigma^2 and by p: one multiplication, one division of the sum)
            const double d = di[k * stride];
            if (d <= rc) {
                const double dl = r - d;
                g += exp((dl * dl) * c2);
            }
        }
        g /= p;
        if (use_local) g *= fac;
        const double v = g >= 1e-10 ? (g * log(g) - g + 1.0) * r2 : r2;
        if (j > 0) sum += prev + v;
        prev = v;
    }
    entropy[i] = -PI * density * sum * sigma;
}


// The same sums by a ladder.  The bins are equally spaced, so for one neighbour at distance d the terms of consecutive bins
// are a Gaussian sampled on a grid: e_0 = exp(-d^2 / 2 sigma^2), e_{j+1} = e_j * rho_j, rho_{j+1} = rho_j * q with
// rho_0 = exp((2 d step - step^2) / 2 sigma^2) and q = exp(-step^2 / sigma^2) — two exponentials per NEIGHBOUR and two
// multiplications per term, where the direct form above spends an exponential and two divisions (~90 double-precision
// instructions) on each of the nbins * n terms: 12.2 ms on 4 M atoms of the published workflow (rc 5.0, sigma 0.2: 26 bins,
// 42 neighbours), the bulk of that call.  Rounding: e_j carries the argument error of e_0 (|arg| * 2^-53, at most 640 here)
// and j steps of the ladder: <= 3e-13 relative on the sums (measured against the oracle: tests), the bar is 1e-6.  Used for
// nbins <= ENT_LADDER_BINS and rc^2 / 2 sigma^2 <= 640 (e_0 stays a normal number); anything else takes the direct kernel.
// Layout as k_sort_rows: one wave per workgroup, 64/L consecutive rows, L lanes to a row (each walks every L-th neighbour,
// the partial sums meet by a butterfly), the rows read with 16-byte loads into LDS.
static constexpr int ENT_LADDER_BINS = 40;
struct EntBins { double inv_p[ENT_LADDER_BINS]; };

template <int L, int NB>
__global__ __launch_bounds__(64) void k_entropy_ladder(const double *__restrict__ dist, const int *__restrict__ nn, int64_t N, int M,
                                                       unsigned inv_m, double rc, double sigma, int use_local, double gd, int nbins,
                                                       double step, double q, EntBins bins, double *__restrict__ entropy)
{
    constexpr int ROWS = 64 / L;
    extern __shared__ double ent_rows[]; // [M][ROWS]
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int rows = (int)((N - row0) < ROWS ? (N - row0) : ROWS);
    const int total = rows * M;
    const int t = threadIdx.x;
    const double *__restrict__ gdist = dist + row0 * M;
    auto slot = [&](int e) {
        const int r = (int)__umulhi((unsigned)e, inv_m);
        return (e - r * M) * ROWS + r;
    };
    if (rows == ROWS && (reinterpret_cast<uintptr_t>(dist) & 15) == 0) { // (ROWS * M is even)
        const double2 *g2 = reinterpret_cast<const double2 *>(gdist);
#pragma unroll 4
        for (int p = t; p < (total >> 1); p += 64) {
            const double2 v = g2[p];
            ent_rows[slot(2 * p)] = v.x; ent_rows[slot(2 * p + 1)] = v.y;
        }
    } else {
        for (int e = t; e < total; e += 64) ent_rows[slot(e)] = gdist[e];
    }
    __syncthreads();
    const int r = t & (ROWS - 1), j = t / ROWS;
    const bool live = r < rows;
    const int64_t i = row0 + r;
    const double PI = 3.14159265358979323846;
    const double c2 = 1.0 / (2.0 * sigma * sigma), lvol = 4. / 3. * PI * rc * rc * rc;
    const double *lr = ent_rows + r;
    double G[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) G[b] = 0.0;
    int nin = 0;
    if (live) {
        const int n = min(nn[i], M); // (a count beyond the row width is the caller's error: stay inside the row)
        for (int k = j; k < n; k += L) {
            const double d = lr[k * ROWS];
            if (d <= rc) {
                ++nin;
                double e = exp(-(d * d) * c2);
                double rho = exp((2.0 * d * step - step * step) * c2);
#pragma unroll
                for (int b8 = 0; b8 < NB; b8 += 8)
                    if (b8 < nbins) { // (uniform: eight bins at a time)
#pragma unroll
                        for (int b = b8; b < b8 + 8; ++b) {
                            G[b] += e;
                            e *= rho;
                            rho *= q;
                        }
                    }
            }
        }
    }
#pragma unroll
    for (int w = ROWS; w < 64; w <<= 1) {
        nin += __shfl_xor(nin, w);
#pragma unroll
        for (int b = 0; b < NB; ++b)
            G[b] += __shfl_xor(G[b], w);
    }
    if (!live || j != 0)
        return;
    double density = gd, fac = 1.0;
    if (use_local) {
        density = nin / lvol;
        fac = gd / density;
    }
    double prev = 0.0, sum = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b)
        if (b < nbins) {
            const double rb = b * step, r2 = rb * rb;
            double g = G[b] * bins.inv_p[b];
            if (use_local) g *= fac;
            const double v = g >= 1e-10 ? (g * log(g) - g + 1.0) * r2 : r2;
            if (b > 0) sum += prev + v;
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
    if (M > CNP_NARROW) { // a wide list: narrowed to the entries within rc where that changes nothing (k_cnp_narrow)
        int *nv = sc.alloc_n<int>((size_t)N * CNP_NARROW), *nc = sc.alloc_n<int>((size_t)N);
        double *nd = sc.alloc_n<double>((size_t)N * CNP_NARROW);
        unsigned long long *flags = sc.alloc_n<unsigned long long>(2);
        if (sc.failed())
            return sc.error();
        MDH_HIP(hipMemsetAsync(flags, 0, 2 * sizeof(unsigned long long), sc.stream()));
        hipLaunchKernelGGL(k_cnp_narrow, dim3(grid_for(N, 64)), dim3(64), 0, sc.stream(), dv, dd, dn, N, M, rc, nv, nd, nc, flags);
        unsigned long long back[2] = {0, 0};
        MDH_HIP(hipMemcpyAsync(back, flags, sizeof(back), hipMemcpyDeviceToHost, sc.stream()));
        MDH_HIP(hipStreamSynchronize(sc.stream()));
        double reach;
        memcpy(&reach, &back[1], sizeof(reach));
        bool twice = false; // can one atom be listed twice in a row ?  Only through two images: a periodic width below twice the reach
        for (int d = 0; d < 3; ++d)
            twice = twice || (b.pbc[d] && b.thick[d] <= 2.0 * reach);
        if (!back[0] && !twice) {
            if (b.tri)
                hipLaunchKernelGGL(k_cnp<true>, dim3(grid_for(N, 128)), dim3(128), 0, sc.stream(), dx, dy, dz, N, b, nv, nd, nc, (int64_t)CNP_NARROW, rc, dc);
            else
                hipLaunchKernelGGL(k_cnp<false>, dim3(grid_for(N, 128)), dim3(128), 0, sc.stream(), dx, dy, dz, N, b, nv, nd, nc, (int64_t)CNP_NARROW, rc, dc);
            return sc.finish(space);
        }
    }
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
    if (g_entropy_variant != 1 && nbins <= ENT_LADDER_BINS && rc * rc / (2.0 * sigma * sigma) <= 640.0 && M >= 2 && M <= 1024) {
        EntBins bins;
        for (int b = 0; b < ENT_LADDER_BINS; ++b) { // :27-38: r_b^2 * factor, entry 0 := entry 1
            const double rb = (b == 0 ? 1 : b) * step;
            bins.inv_p[b] = 1.0 / ((rb * rb) * factor);
        }
        const double q = exp(-(step * step) / (sigma * sigma));
        int L = 1; // lanes to a row: the fewest that keep the rows of a wave within ~13 KB of LDS
        while (L < 8 && (size_t)(64 / L) * M * 8 > 13 * 1024) L <<= 1;
        if (g_entropy_variant >= 2) L = g_entropy_variant == 2 ? 1 : g_entropy_variant == 3 ? 2 : g_entropy_variant == 4 ? 4 : 8;
        const size_t lds = (size_t)(64 / L) * M * 8;
        if (lds <= 64 * 1024) {
            const unsigned inv_m = (unsigned)((0x100000000ull + (uint64_t)M - 1) / (uint64_t)M);
            const dim3 grid(grid_for(N, 64 / L)), block(64);
#define MDH_ENT_LAUNCH(LL, NB)                                                                                                        \
    hipLaunchKernelGGL((k_entropy_ladder<LL, NB>), grid, block, lds, sc.stream(), dd, dn, N, (int)M, inv_m, rc, sigma,              \
                       use_local_density ? 1 : 0, gd, nbins, step, q, bins, de)
#define MDH_ENT_BINS(LL)                                                                                                              \
    if (nbins <= 16) MDH_ENT_LAUNCH(LL, 16); else if (nbins <= 32) MDH_ENT_LAUNCH(LL, 32); else MDH_ENT_LAUNCH(LL, 40)
            switch (L) {
            case 1: MDH_ENT_BINS(1); break;
            case 2: MDH_ENT_BINS(2); break;
            case 4: MDH_ENT_BINS(4); break;
            default: MDH_ENT_BINS(8); break;
            }
#undef MDH_ENT_BINS
#undef MDH_ENT_LAUNCH
            return sc.finish(space);
        }
    }
    const size_t lds = (size_t)M * 65 * sizeof(double);
    if (lds <= 48 * 1024) // (the reference reads nn[i] entries of a row of M: a count beyond M is its caller's error, here as there)
        hipLaunchKernelGGL(k_entropy<true>, dim3(grid_for(N, 64)), dim3(64), lds, sc.stream(), dd, dn, N, M, rc, sigma, use_local_density ? 1 : 0,
                           gd, nbins, step, factor, de);
    else
        hipLaunchKernelGGL(k_entropy<false>, dim3(grid_for(N, 64)), dim3(64), 0, sc.stream(), dd, dn, N, M, rc, sigma, use_local_density ? 1 : 0,
                           gd, nbins, step, factor, de);
    return sc.finish(space);
}

extern "C" int mdh_debug_set_entropy_variant(int v) // 0 automatic, 1 the direct kernel, 2/3/4/5 the ladder with 1/2/4/8 lanes to a row (A/B measurements, tests)
{
    g_entropy_variant = v;
    return MDH_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// atomic temperature                                          src/atomic_temperature.cpp:9-112
// cluster analysis (connected components of the bond graph)   src/cluster.cpp:9-153
// ---------------------------------------------------------------------------------------------------------------
#include "grid.hpp"

namespace mdh {

// (velocity, mass) of an atom as one 32-byte record (x, y, z = velocity, w = mass): a neighbour is two 16-byte requests instead of
// four 8-byte ones from four arrays.  Both sweeps gather every neighbour (42 at rc 5.0 in fcc Cu): 1.3 G requests per call at 4 M
// atoms, each a 64-byte line from L2 whatever it carries — the kernel's bound (12.7 ms with four arrays)
__global__ __launch_bounds__(256) void k_pack_velocity_mass(const double *__restrict__ vx, const double *__restrict__ vy,
                                                            const double *__restrict__ vz, const double *__restrict__ mass, int64_t N,
                                                            Pos4 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = Pos4{vx[i], vy[i], vz[i], mass[i]};
}

__global__ __launch_bounds__(64) void k_atomic_temp(const int *__restrict__ verlet, const double *__restrict__ dist, int64_t N,
                                                    int64_t M, const Pos4 *__restrict__ vm, double rc, double *__restrict__ T)
{
    __shared__ int ids[ROW_CHUNK * 64];
    __shared__ double dst[ROW_CHUNK * 64];
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + threadIdx.x;
    const bool on = i < N;
    constexpr double kb = 1.380649e-23, dim = 3.0, afu = 6.022140857e23;
    constexpr double mass_factor = 1.0 / afu / 1000.0, vel_conv = 1e4;
    const Pos4 self = vm[on ? i : 0];
    const double mi = self.w;
    const double vxi = self.x, vyi = self.y, vzi = self.z;
    double sx = vxi * mi, sy = vyi * mi, sz = vzi * mi, ms = mi;
    int n = 1;
    // Both sweeps take the rows of the workgroup a chunk of 16 columns at a time through LDS (stage_row_chunk), then eight entries
    // at a time: the (velocity, mass) records of the eight neighbours in flight together, the entries used in list order up to the
    // first pad, as the reference's entry-by-entry loop does (same sums, bit for bit).
    auto sweep = [&](auto &&use) {
        bool stop = !on;
        for (int c0 = 0; c0 < M; c0 += ROW_CHUNK) {
            if (!__syncthreads_or(stop ? 0 : 1)) // (also the barrier between the walk of one chunk and the staging of the next)
                break;
            stage_row_chunk<true>(verlet, dist, N, M, row0, c0, ids, dst);
            __syncthreads();
            for (int q0 = 0; q0 < ROW_CHUNK && c0 + q0 < M && !stop; q0 += 8) {
                int js[8];
                double ds[8], mj[8], ux[8], uy[8], uz[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int qq = min(q0 + u, min(ROW_CHUNK, (int)M - c0) - 1);
                    js[u] = ids[qq * 64 + threadIdx.x];
                    ds[u] = dst[qq * 64 + threadIdx.x];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t jj = (unsigned)js[u] < (unsigned)N ? js[u] : i;
                    const Pos4 r = vm[jj];
                    mj[u] = r.w; ux[u] = r.x; uy[u] = r.y; uz[u] = r.z;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (!stop && c0 + q0 + u < M) {
                        if ((unsigned)js[u] >= (unsigned)N) stop = true; // pad (-1) or an index of another system
                        else if (js[u] != i && ds[u] <= rc) use(mj[u], ux[u], uy[u], uz[u]);
                    }
            }
        }
        __syncthreads();
    };
    sweep([&](double mj, double ux, double uy, double uz) { // mass-weighted mean velocity of the neighbourhood (:44-62)
        sx += ux * mj; sy += uy * mj; sz += uz * mj;
        ++n;
        ms += mj;
    });
    const double mx = sx / ms, my = sy / ms, mz = sz / ms;
    double dx = vxi - mx, dy = vyi - my, dz = vzi - mz;
    double ke = 0.0;
    ke += 0.5 * mi * mass_factor * (dx * dx + dy * dy + dz * dz) * vel_conv;
    sweep([&](double mj, double ux, double uy, double uz) { // kinetic energy relative to it (:78-101)
        dx = ux - mx; dy = uy - my; dz = uz - mz;
        ke += 0.5 * mj * mass_factor * (dx * dx + dy * dy + dz * dz) * vel_conv;
    });
    if (on) T[i] = ke * 2.0 / (dim * n * kb);
}

// --- connected components: min-index hooking + pointer jumping -------------------------------------------------
__device__ __forceinline__ int cc_find(int *parent, int i)
{
    int p = parent[i];
    while (p != i) {
        const int g = parent[p];
        if (g != p) parent[i] = g; // path halving (benign race: any ancestor is a valid parent)
        i = p;
        p = g;
    }
    return i;
}

// BY_BOND: an entry > -1 is a bond (get_cluster_by_bond :63); otherwise distance <= rc (get_cluster :32)
// The list is symmetric when this runs (k_cc_fingerprint), so a bond is taken from its larger end only, and the root of the
// atom's own tree is carried from bond to bond (one read confirms it) instead of walked again.  One pass joins every bond's two
// trees — the loop does not leave a bond before both ends have one root or its hook went in, and trees only merge; what follows
// the pass is a check of the labels (k_cc_verify), not a second pass (rounds 2-4 ran one: as long as the first).
template <bool BY_BOND>
__global__ __launch_bounds__(64) void k_cc_hook(const int *__restrict__ verlet, const double *__restrict__ dist,
                                                const int *__restrict__ nn, int64_t N, int64_t M, double rc,
                                                int *__restrict__ parent)
{
    __shared__ int ids[ROW_CHUNK * 64];
    __shared__ double dst[BY_BOND ? 1 : ROW_CHUNK * 64];
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + threadIdx.x;
    const int n = i < N ? min(nn[i], (int)M) : 0;
    const int most = wave_max(n);
    int mine = (int)i; // an ancestor of i: its root as of the last bond
    for (int c0 = 0; c0 < most; c0 += ROW_CHUNK) {
        __syncthreads();
        stage_row_chunk<!BY_BOND>(verlet, dist, N, M, row0, c0, ids, dst);
        __syncthreads();
        for (int q = 0; q < ROW_CHUNK && c0 + q < n; ++q) {
            const int j = ids[q * 64 + threadIdx.x];
            const bool bond = BY_BOND ? (j > -1) : (dst[q * 64 + threadIdx.x] <= rc);
            if (!bond || j < 0 || j >= i)
                continue;
            int a = cc_find(parent, mine), b = cc_find(parent, j);
            while (a != b) { // hook the larger root under the smaller one
                if (a < b) { const int t = a; a = b; b = t; }
                const int old = atomicMin(&parent[a], b);
                if (old == a) break;
                a = cc_find(parent, old < a ? old : a);
                b = cc_find(parent, b);
            }
            mine = a < b ? a : b; // where both trees hang now
        }
    }
}

// after the labels are written: do the two ends of every bond carry one label ?  (the flag stays 0; a 1 sends the caller round again)
template <bool BY_BOND>
__global__ __launch_bounds__(64) void k_cc_verify(const int *__restrict__ verlet, const double *__restrict__ dist,
                                                  const int *__restrict__ nn, int64_t N, int64_t M, double rc,
                                                  const int *__restrict__ cluster, int *__restrict__ flag)
{
    __shared__ int ids[ROW_CHUNK * 64];
    __shared__ double dst[BY_BOND ? 1 : ROW_CHUNK * 64];
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + threadIdx.x;
    const int n = i < N ? min(nn[i], (int)M) : 0;
    const int most = wave_max(n);
    const int ci = i < N ? cluster[i] : 0;
    bool bad = false;
    for (int c0 = 0; c0 < most; c0 += ROW_CHUNK) {
        __syncthreads();
        stage_row_chunk<!BY_BOND>(verlet, dist, N, M, row0, c0, ids, dst);
        __syncthreads();
        for (int q = 0; q < ROW_CHUNK && c0 + q < n; ++q) {
            const int j = ids[q * 64 + threadIdx.x];
            const bool bond = BY_BOND ? (j > -1) : (dst[q * 64 + threadIdx.x] <= rc);
            if (bond && j >= 0 && j < i) bad = bad || cluster[j] != ci;
        }
    }
    if (bad) *flag = 1;
}

// read-only walk to the root (the forest no longer changes once the hooking passes are over)
__device__ __forceinline__ int cc_root(const int *__restrict__ parent, int i)
{
    int p = parent[i];
    while (p != i) {
        i = p;
        p = parent[p];
    }
    return i;
}

// Roots only.  No pointer is written here or in k_cc_label: a path-halving store of one thread (a stale grandparent) landing
// after another thread's "parent[i] = root" left atoms pointing at a non-root, whose rank is not a cluster number (found by
// the randomised sweep: same clusters, ids off by the number of roots skipped).
__global__ __launch_bounds__(256) void k_cc_flatten(const int *__restrict__ parent, int64_t N, unsigned *__restrict__ is_root)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    is_root[i] = parent[i] == (int)i ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_cc_label(const int *__restrict__ parent, const int *__restrict__ rank, int64_t N,
                                                  int *__restrict__ cluster, int *__restrict__ count)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    cluster[i] = rank[cc_root(parent, (int)i)] + 1; // clusters are numbered by their smallest atom index, from 1 (:23-27)
    if (i == 0) *count = rank[N];
}

// Is every bond i -> j also listed as j -> i?  (cutoff lists are; k-nearest lists and one-sided type filters are not)
// By a fingerprint: every directed bond adds a 128-bit hash of its unordered pair, with a plus sign from its smaller end and
// a minus sign from its larger one; both sums are zero when the bonds listed upwards are the bonds listed downwards.  One
// walk over the atom's own row — the test it replaces looked every bond up in the other atom's row (9.2 ms of the 22 ms of
// a 4 M-atom cluster analysis).  A list that is not symmetric (or lists a bond twice in one direction only) leaves a
// non-zero sum unless 128 bits collide, and takes the reference's literal sweep below, which is right for any list.
__device__ __forceinline__ unsigned long long cc_mix(unsigned long long v)
{
    v ^= v >> 30; v *= 0xbf58476d1ce4e5b9ull;
    v ^= v >> 27; v *= 0x94d049bb133111ebull;
    return v ^ (v >> 31);
}
template <bool BY_BOND>
__global__ __launch_bounds__(64) void k_cc_fingerprint(const int *__restrict__ verlet, const double *__restrict__ dist,
                                                       const int *__restrict__ nn, int64_t N, int64_t M, double rc,
                                                       unsigned long long *__restrict__ sums)
{
    __shared__ int ids[ROW_CHUNK * 64];
    __shared__ double dst[BY_BOND ? 1 : ROW_CHUNK * 64];
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + threadIdx.x;
    const int n = i < N ? min(nn[i], (int)M) : 0;
    const int most = wave_max(n);
    unsigned long long s0 = 0, s1 = 0;
    for (int c0 = 0; c0 < most; c0 += ROW_CHUNK) {
        __syncthreads();
        stage_row_chunk<!BY_BOND>(verlet, dist, N, M, row0, c0, ids, dst);
        __syncthreads();
        for (int q = 0; q < ROW_CHUNK && c0 + q < n; ++q) {
            const int j = ids[q * 64 + threadIdx.x];
            const bool bond = BY_BOND ? (j > -1) : (dst[q * 64 + threadIdx.x] <= rc);
            if (!bond || j < 0 || j >= N || j == i)
                continue;
            const unsigned long long lo = j < i ? (unsigned long long)j : (unsigned long long)i, hi = j < i ? (unsigned long long)i : (unsigned long long)j;
            const unsigned long long key = lo * (unsigned long long)N + hi;
            const unsigned long long h0 = cc_mix(key + 0x9e3779b97f4a7c15ull), h1 = cc_mix(key ^ 0xd6e8feb86659fd93ull);
            if (j > i) { s0 += h0; s1 += h1; } else { s0 -= h0; s1 -= h1; }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64);
        s1 += __shfl_xor(s1, d, 64);
    }
    if (threadIdx.x == 0 && (s0 | s1)) {
        atomicAdd(sums, s0);
        atomicAdd(sums + 1, s1);
    }
}

// The reference's sweep taken literally, for lists that are NOT symmetric (src/cluster.cpp:14-53, :56-93): seeds in index
// order, each cluster = the still unlabelled atoms reachable from its seed along DIRECTED bonds; the seed itself is only
// labelled when a member points back at it (or when it has no bond at all), so it may end up in a later cluster or stay -1.
// One workgroup walks the frontiers level by level; the common symmetric case never comes here.
static constexpr int CC_DIR_THREADS = 1024;
template <bool BY_BOND>
__global__ __launch_bounds__(CC_DIR_THREADS) void k_cc_directed(const int *__restrict__ verlet, const double *__restrict__ dist,
                                                                const int *__restrict__ nn, int64_t N, int64_t M, double rc,
                                                                int *__restrict__ cluster, int *__restrict__ qa,
                                                                int *__restrict__ qb, int *__restrict__ count)
{
    __shared__ int s_seed, s_next;
    const int tid = threadIdx.x;
    for (int64_t i = tid; i < N; i += CC_DIR_THREADS) cluster[i] = -1;
    __syncthreads();
    int cid = 0;
    int64_t scan = 0;
    for (;;) {
        // next unlabelled seed at or after `scan`
        int seed = 2147483647;
        for (int64_t base = scan; base < N; base += CC_DIR_THREADS) {
            if (tid == 0) s_seed = 2147483647;
            __syncthreads();
            const int64_t idx = base + tid;
            if (idx < N && cluster[idx] == -1) atomicMin(&s_seed, (int)idx);
            __syncthreads();
            seed = s_seed;
            __syncthreads();
            if (seed != 2147483647)
                break;
        }
        if (seed == 2147483647)
            break;
        scan = (int64_t)seed + 1;
        ++cid;
        if (tid == 0) qa[0] = seed;
        int cnt = 1;
        __syncthreads();
        while (cnt > 0) {
            if (tid == 0) s_next = 0;
            // members without any bond label themselves (:45-48)
            for (int e = tid; e < cnt; e += CC_DIR_THREADS) {
                const int cur = qa[e];
                const int n = nn[cur];
                bool any = false;
                for (int q = 0; q < n && !any; ++q)
                    any = BY_BOND ? (verlet[(int64_t)cur * M + q] > -1) : (dist[(int64_t)cur * M + q] <= rc);
                if (!any) cluster[cur] = cid;
            }
            __syncthreads();
            for (int64_t it = tid; it < (int64_t)cnt * M; it += CC_DIR_THREADS) {
                const int cur = qa[it / M];
                const int q = (int)(it % M);
                if (q >= nn[cur])
                    continue;
                const int nb = verlet[(int64_t)cur * M + q];
                const bool bond = BY_BOND ? (nb > -1) : (dist[(int64_t)cur * M + q] <= rc);
                if (bond && nb >= 0 && nb < N && atomicCAS(&cluster[nb], -1, cid) == -1)
                    qb[atomicAdd(&s_next, 1)] = nb;
            }
            __syncthreads();
            cnt = s_next;
            int *t = qa; qa = qb; qb = t;
            __syncthreads();
        }
    }
    if (tid == 0) *count = cid;
}

__global__ void k_iota(int *p, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int)i;
}

__global__ __launch_bounds__(256) void k_filter_by_type(int *__restrict__ verlet, const double *__restrict__ dist,
                                                        const int *__restrict__ nn, const int *__restrict__ type, int64_t N,
                                                        int64_t M, const int *__restrict__ t1, const int *__restrict__ t2,
                                                        const double *__restrict__ r, int ntype)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    const int n = nn[i], ti = type[i];
    for (int q = 0; q < n; ++q) {
        const int j = verlet[i * M + q];
        if ((unsigned)j >= (unsigned)N)
            continue; // the reference would index type_list(-1) here; a row is filtered once
        const int tj = type[j];
        const double d = dist[i * M + q];
        bool cut = false;
        for (int k = 0; k < ntype; ++k)
            cut = cut || (t1[k] == ti && t2[k] == tj && d > r[k]);
        if (cut) verlet[i * M + q] = -1;
    }
}

} // namespace mdh

extern "C" int mdh_atomic_temperature(const int *verlet, const double *dist, int64_t N, int64_t M, const double *vx,
                                      const double *vy, const double *vz, const double *mass, double *T, double rc, int space,
                                      void *stream)
{
    if (N < 0 || M <= 0) { set_error("mdh_atomic_temperature: empty neighbor list"); return MDH_ERR_ARG; }
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const double *dd = sc.stage_in(dist, (size_t)(N * M), space);
    const double *dvx = sc.stage_in(vx, (size_t)N, space), *dvy = sc.stage_in(vy, (size_t)N, space), *dvz = sc.stage_in(vz, (size_t)N, space);
    const double *dm = sc.stage_in(mass, (size_t)N, space);
    double *dT = sc.stage(T, (size_t)N, space, false, true);
    if (sc.failed())
        return sc.error();
    Pos4 *vm = sc.alloc_n<Pos4>((size_t)N);
    if (!vm)
        return sc.error();
    ProfRange pr("k_atomic_temp", sc.stream());
    hipLaunchKernelGGL(k_pack_velocity_mass, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), dvx, dvy, dvz, dm, N, vm);
    hipLaunchKernelGGL(k_atomic_temp, dim3(grid_for(N, 64)), dim3(64), 0, sc.stream(), dv, dd, N, M, vm, rc, dT);
    return sc.finish(space);
}

// by_bond = 0: get_cluster (dist may not be NULL); by_bond = 1: get_cluster_by_bond (dist unused).
// cluster (N) i32 receives ids 1..n_clusters; *n_clusters_host receives the count (the functions' return value).
extern "C" int mdh_cluster(const int *verlet, const double *dist, const int *nn, int64_t N, int64_t M, double rc, int by_bond,
                           int *cluster, int *n_clusters_host, int space, void *stream)
{
    if (N < 0 || M <= 0 || (!by_bond && !dist)) { set_error("mdh_cluster: empty neighbor list"); return MDH_ERR_ARG; }
    if (n_clusters_host) *n_clusters_host = 0;
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const double *dd = by_bond ? nullptr : sc.stage_in(dist, (size_t)(N * M), space);
    const int *dn = sc.stage_in(nn, (size_t)N, space);
    int *dc = sc.stage(cluster, (size_t)N, space, false, true);
    int *parent = sc.alloc_n<int>((size_t)N);
    unsigned *is_root = sc.alloc_n<unsigned>((size_t)N);
    int *rank = sc.alloc_n<int>((size_t)N + 1);
    int *flag = sc.alloc_n<int>(2);
    if (sc.failed())
        return sc.error();
    const dim3 grid(grid_for(N, 256)), block(256), rgrid(grid_for(N, 64)), rblock(64); // (rgrid: the kernels that stage rows, one wave each)
    {   // directed lists get the reference's sweep itself; symmetric ones (every cutoff list) the union-find below
        unsigned long long *sums = sc.alloc_n<unsigned long long>(2);
        if (!sums)
            return sc.error();
        MDH_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(unsigned long long), st));
        if (by_bond)
            hipLaunchKernelGGL(k_cc_fingerprint<true>, rgrid, rblock, 0, st, dv, dd, dn, N, M, rc, sums);
        else
            hipLaunchKernelGGL(k_cc_fingerprint<false>, rgrid, rblock, 0, st, dv, dd, dn, N, M, rc, sums);
        unsigned long long got[2] = {0, 0};
        MDH_HIP(hipMemcpyAsync(got, sums, sizeof(got), hipMemcpyDeviceToHost, st));
        MDH_HIP(hipStreamSynchronize(st));
        if (got[0] | got[1]) {
            int *qa = sc.alloc_n<int>((size_t)N + 1), *qb = sc.alloc_n<int>((size_t)N + 1);
            if (sc.failed())
                return sc.error();
            if (by_bond)
                hipLaunchKernelGGL(k_cc_directed<true>, dim3(1), dim3(CC_DIR_THREADS), 0, st, dv, dd, dn, N, M, rc, dc, qa, qb, flag + 1);
            else
                hipLaunchKernelGGL(k_cc_directed<false>, dim3(1), dim3(CC_DIR_THREADS), 0, st, dv, dd, dn, N, M, rc, dc, qa, qb, flag + 1);
            int cnt = 0;
            MDH_HIP(hipMemcpyAsync(&cnt, flag + 1, sizeof(int), hipMemcpyDeviceToHost, st));
            MDH_HIP(hipStreamSynchronize(st));
            if (n_clusters_host) *n_clusters_host = cnt;
            return sc.finish(space);
        }
    }
    hipLaunchKernelGGL(k_iota, grid, block, 0, st, parent, N);
    // One hooking pass joins every bond's two trees; the labels are then checked bond by bond (k_cc_verify) and the pass is only
    // repeated — on the forest as it stands — if a bond's ends disagree, which the hooking loop rules out (never seen).  The
    // flags come back in one copy per round: this analysis is not on a hot loop.
    int cnt = 0;
    for (int round = 0; round < 64; ++round) {
        MDH_HIP(hipMemsetAsync(flag, 0, 2 * sizeof(int), st));
        if (by_bond)
            hipLaunchKernelGGL(k_cc_hook<true>, rgrid, rblock, 0, st, dv, dd, dn, N, M, rc, parent);
        else
            hipLaunchKernelGGL(k_cc_hook<false>, rgrid, rblock, 0, st, dv, dd, dn, N, M, rc, parent);
        hipLaunchKernelGGL(k_cc_flatten, grid, block, 0, st, parent, N, is_root);
        MDH_TRY(exclusive_scan_u32(sc, is_root, rank, N));
        hipLaunchKernelGGL(k_cc_label, grid, block, 0, st, parent, rank, N, dc, flag + 1);
        if (by_bond)
            hipLaunchKernelGGL(k_cc_verify<true>, rgrid, rblock, 0, st, dv, dd, dn, N, M, rc, dc, flag);
        else
            hipLaunchKernelGGL(k_cc_verify<false>, rgrid, rblock, 0, st, dv, dd, dn, N, M, rc, dc, flag);
        int back[2] = {0, 0};
        MDH_HIP(hipMemcpyAsync(back, flag, sizeof(back), hipMemcpyDeviceToHost, st));
        MDH_HIP(hipStreamSynchronize(st));
        cnt = back[1];
        if (!back[0])
            break;
    }
    if (n_clusters_host) *n_clusters_host = cnt;
    return sc.finish(space);
}

// replaces _cluster.filter_by_type (src/cluster.cpp:110-148): verlet entries whose pair distance exceeds the cutoff of
// their (type_i, type_j) pair become -1.  t1, t2, r: host arrays of length ntype.
extern "C" int mdh_filter_by_type(int *verlet, const double *dist, const int *nn, const int *type, int64_t N, int64_t M,
                                  const int *t1_host, const int *t2_host, const double *r_host, int ntype, int space,
                                  void *stream)
{
    if (N < 0 || M <= 0 || ntype < 0) { set_error("mdh_filter_by_type: bad sizes"); return MDH_ERR_ARG; }
    if (N == 0 || ntype == 0)
        return MDH_OK;
    Scope sc(stream);
    int *dv = sc.stage(verlet, (size_t)(N * M), space, true, true);
    const double *dd = sc.stage_in(dist, (size_t)(N * M), space);
    const int *dn = sc.stage_in(nn, (size_t)N, space);
    const int *dt = sc.stage_in(type, (size_t)N, space);
    const int *d1 = sc.stage_in(t1_host, (size_t)ntype, MDH_HOST), *d2 = sc.stage_in(t2_host, (size_t)ntype, MDH_HOST);
    const double *dr = sc.stage_in(r_host, (size_t)ntype, MDH_HOST);
    if (sc.failed())
        return sc.error();
    hipLaunchKernelGGL(k_filter_by_type, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), dv, dd, dn, dt, N, M, d1, d2, dr, ntype);
    MDH_HIP(hipStreamSynchronize(sc.stream())); // t1/t2/r are staged from caller memory
    return sc.finish(space);
}

MDH_WARM_UNIT(consumers)
