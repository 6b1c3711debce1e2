// sbo.hip — Steinhardt bond-orientational order parameters on gfx950.
//
// Replaces src/steinhardt_bond_orientation.cpp: get_sq :677-784 over _compute_ql
// :288-576 (stage 1 q_lm, stage 2 Lechner-Dellago neighbour average, stage 3
// q_l / w_l / w_l-hat) and identifySolidLiquid :578-675.
//
// Stage 1: one thread per atom; the 2*nl*(2lmax+1) running sums of an atom live
// in LDS (lane-strided, conflict free) while its neighbours are visited in list
// order — the same sequence of += as the reference, so sums agree to the last
// bit wherever the spherical-harmonic factors do.  The (l,m) normalisation
// sqrt((2l+1)/(4 pi prod)) (:270-279) does not depend on the bond and is
// tabulated once on the host with the reference's expression.
#include "common.hpp"
#include <vector>

namespace mdh {

static int g_sq_variant = 0; // test hook: 1 = generic stage-1 kernel for every degree; 2 = sixteen lanes per atom for l = 4, 6 (measuring variant); 3 = one launch per degree even for the pair (4, 6)
static constexpr int SBO_MAXL = 16;  // entries of llist
static constexpr int SBO_LMAX = 40;  // largest degree (3l+1 must index the 168-entry factorial table)
static constexpr double MY_PI = 3.14159265358979323846;

struct LList { int l[SBO_MAXL]; int n; };

__device__ __forceinline__ double assoc_legendre(int l, int m, double x) // :243-268
{
    double p = 1.0, pm1 = 0.0, pm2 = 0.0;
    if (m != 0) {
        const double sqx = sqrt(1.0 - x * x);
        for (int i = 1; i < m + 1; ++i)
            p *= (2 * i - 1) * sqx;
    }
    for (int i = m + 1; i < l + 1; ++i) {
        pm2 = pm1;
        pm1 = p;
        p = ((2 * i - 1) * x * pm1 - (i + m - 1) * pm2) / (i - m);
    }
    return p;
}

// LDS accumulators: component c of thread t at acc[c * blockDim.x + t]
template <bool TRI, bool LDSACC>
__global__ void k_sq_stage1(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                            int64_t N, DBox b, const int *__restrict__ NL, const double *__restrict__ DL, int64_t M,
                            const int *__restrict__ NN, const double *__restrict__ weight, LList ll, int nnn, int lmax,
                            int use_voronoi, double rc, int use_weight, const double *__restrict__ norm /* [nl][lmax+1] */,
                            double *__restrict__ qlm_r, double *__restrict__ qlm_i)
{
    extern __shared__ double acc[];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    const int nz = 2 * lmax + 1, nl = ll.n;
    const int stride = nl * nz;
    const int bd = blockDim.x, t = threadIdx.x;
    double *gr = qlm_r + i * stride, *gi = qlm_i + i * stride;
    if (LDSACC) // start from the caller's (pre-zeroed) content, like the reference's '+=' on the output arrays
        for (int c = 0; c < stride; ++c) {
            acc[c * bd + t] = gr[c];
            acc[(stride + c) * bd + t] = gi[c];
        }
#define QR(c) (LDSACC ? acc[(c) * bd + t] : gr[c])
#define QI(c) (LDSACC ? acc[(stride + (c)) * bd + t] : gi[c])
    const double EPS = 1e-15;
    const double x1 = x[i], y1 = y[i], z1 = z[i];
    int cnt = NN[i];
    if (!use_voronoi && nnn > 0) // :329-333
        cnt = nnn;
    double wsum = 0.0;
    for (int jj = 0; jj < cnt; ++jj) {
        const int64_t idx = i * M + jj;
        const int j = NL[idx];
        if ((unsigned)j >= (unsigned)N) // pad or foreign index
            continue;
        double dx = x[j] - x1, dy = y[j] - y1, dz = z[j] - z1; // :346-350
        pbc<TRI>(b, dx, dy, dz);
        const double r = DL[idx];
        if (!((r > EPS) && (r <= rc)))
            continue;
        const double w = use_weight ? weight[idx] : 1.0;
        wsum += w;
        const double rinv = 1.0 / r;
        const double ct = dz * rinv;
        double er = dx, ei = dy;
        const double rxy2 = er * er + ei * ei;
        if (rxy2 < EPS * EPS) { er = 1.0; ei = 0.0; }
        else { const double s = 1.0 / sqrt(rxy2); er *= s; ei *= s; }
        for (int il = 0; il < nl; ++il) {
            const int l = ll.l[il];
            const int o = il * nz;
            const double *nrm = norm + il * (lmax + 1);
            QR(o + l) += w * (nrm[0] * assoc_legendre(l, 0, ct));
            double mr = er, mi = ei;
            for (int m = 1; m < l + 1; ++m) {
                const double pf = nrm[m] * assoc_legendre(l, m, ct);
                const double cr = pf * mr, ci = pf * mi;
                const double wr = w * cr, wi = w * ci;
                QR(o + l + m) += wr;
                QI(o + l + m) += wi;
                if (m & 1) { QR(o + l - m) -= wr; QI(o + l - m) += wi; }
                else { QR(o + l - m) += wr; QI(o + l - m) -= wi; }
                const double tr = mr * er - mi * ei, ti = mr * ei + mi * er;
                mr = tr; mi = ti;
            }
        }
    }
    const double f = 1.0 / wsum; // :422 (no guard: NaN/inf for an atom without neighbours)
    for (int il = 0; il < nl; ++il) {
        const int l = ll.l[il];
        for (int m = 0; m < 2 * l + 1; ++m) {
            const int c = il * nz + m;
            QR(c) *= f;
            QI(c) *= f;
        }
    }
    if (LDSACC)
        for (int c = 0; c < stride; ++c) {
            gr[c] = acc[c * bd + t];
            gi[c] = acc[(stride + c) * bd + t];
        }
#undef QR
#undef QI
}

// Stage 1 for ONE degree l known at compile time: the 2l+1 running sums of an atom in registers, the Legendre recurrences
// unrolled (same expressions as assoc_legendre, so the same values; the square root under them is computed once per bond),
// the row segments of the 64 atoms of a workgroup copied in and out through LDS with coalesced accesses.  One launch per
// entry of llist; sums and their order are those of the generic kernel (bond by bond in list order), bit for bit.
template <bool TRI, int L>
__global__ __launch_bounds__(64) void k_sq_stage1_l(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                    int64_t N, DBox b, const int *__restrict__ NL, const double *__restrict__ DL, int64_t M,
                                                    const int *__restrict__ NN, const double *__restrict__ weight, int il, int stride, int nz,
                                                    int lmax, int nnn, int use_voronoi, double rc, int use_weight,
                                                    const double *__restrict__ norm, double *__restrict__ qlm_r, double *__restrict__ qlm_i,
                                                    double *__restrict__ qn, int ncol, int lrt)
{
    constexpr int NM = 2 * L + 1;
    __shared__ double sr[NM][65], si[NM][65];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + t;
    const int rows = (int)(N - row0 < 64 ? N - row0 : 64);
    const int o = il * nz; // this degree's segment of a row: components o .. o + 2l
    // (NM trips at most, unrolled: as a run-time loop every trip waited for its two loads before the next were issued — NM dependent
    // memory latencies in front of every workgroup's work; now the 2 NM loads are in flight together)
    {
        double gr[NM], gi[NM];
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            const int e = t + 64 * q;
            const int r = e / NM, m = e - r * NM;
            const bool in = e < rows * NM;
            gr[q] = in ? qlm_r[(row0 + r) * stride + o + m] : 0.0; // the caller's (pre-zeroed) content: the reference adds onto it
            gi[q] = in ? qlm_i[(row0 + r) * stride + o + m] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            const int e = t + 64 * q;
            const int r = e / NM, m = e - r * NM;
            if (e < rows * NM) { sr[m][r] = gr[q]; si[m][r] = gi[q]; }
        }
    }
    __syncthreads();
    if (i < N) {
        double ar[NM], ai[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) { ar[m] = sr[m][t]; ai[m] = si[m][t]; }
        double nrm[L + 1];
#pragma unroll
        for (int m = 0; m <= L; ++m) nrm[m] = norm[il * (lmax + 1) + m];
        const double EPS = 1e-15;
        const double x1 = x[i], y1 = y[i], z1 = z[i];
        int cnt = NN[i];
        if (!use_voronoi && nnn > 0) // :329-333
            cnt = nnn;
        double wsum = 0.0;
        // The row is read four entries at a time — ids, distances, weights, then the four neighbours' positions, each group of
        // loads in flight together — and the entries are then taken in list order as before (same sums, bit for bit).  One
        // entry per trip meant two dependent memory latencies per neighbour with nothing else to do at 2-3 waves per SIMD:
        // 80 % of the wave-cycles were spent parked.
        int cj[4];
        double cr_[4], cw[4], cx[4], cy[4], cz[4];
        for (int jj = 0; jj < cnt; ++jj) {
            const int u = jj & 3;
            if (u == 0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int64_t idx = i * M + min(jj + v, cnt - 1);
                    cj[v] = NL[idx];
                    cr_[v] = DL[idx];
                    cw[v] = use_weight ? weight[idx] : 1.0;
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int js = (unsigned)cj[v] < (unsigned)N ? cj[v] : (int)i; // (an index outside the system is skipped below)
                    cx[v] = x[js]; cy[v] = y[js]; cz[v] = z[js];
                }
            }
            int j = cj[0];
            double r = cr_[0], w = cw[0], xj = cx[0], yj = cy[0], zj = cz[0];
#pragma unroll
            for (int v = 1; v < 4; ++v)
                if (u == v) { j = cj[v]; r = cr_[v]; w = cw[v]; xj = cx[v]; yj = cy[v]; zj = cz[v]; }
            if ((unsigned)j >= (unsigned)N)
                continue;
            double dx = xj - x1, dy = yj - y1, dz = zj - z1; // :346-350
            pbc<TRI>(b, dx, dy, dz);
            if (!((r > EPS) && (r <= rc)))
                continue;
            wsum += w;
            const double rinv = 1.0 / r;
            const double ct = dz * rinv;
            double er = dx, ei = dy;
            const double rxy2 = er * er + ei * ei;
            if (rxy2 < EPS * EPS) { er = 1.0; ei = 0.0; }
            else { const double sc = 1.0 / sqrt(rxy2); er *= sc; ei *= sc; }
            ar[L] += w * (nrm[0] * assoc_legendre(L, 0, ct));
            double mr = er, mi = ei;
#pragma unroll
            for (int m = 1; m < L + 1; ++m) {
                const double pf = nrm[m] * assoc_legendre(L, m, ct);
                const double cr = pf * mr, ci = pf * mi;
                const double wr = w * cr, wi = w * ci;
                ar[L + m] += wr;
                ai[L + m] += wi;
                if (m & 1) { ar[L - m] -= wr; ai[L - m] += wi; }
                else { ar[L - m] += wr; ai[L - m] -= wi; }
                const double tr = mr * er - mi * ei, ti = mr * ei + mi * er;
                mr = tr; mi = ti;
            }
        }
        const double f = 1.0 / wsum; // :422 (no guard: NaN/inf for an atom without neighbours)
#pragma unroll
        for (int m = 0; m < NM; ++m) { sr[m][t] = ar[m] * f; si[m][t] = ai[m] * f; }
        if (qn) { // q_l of a call without averaging and without w_l: the row is in registers, stage 3 (:520-531) would re-read it
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const double vr = ar[m] * f, vi = ai[m] * f; // the stored values
                s += vr * vr + vi * vi;
            }
            qn[i * ncol + il] = sqrt(4 * MY_PI / (2 * lrt + 1)) * sqrt(s); // (lrt = L at run time: the factor as the kernel of stage 3 computes it)
        }
    }
    __syncthreads();
    for (int e = t; e < rows * NM; e += 64) {
        const int r = e / NM, m = e - r * NM;
        qlm_r[(row0 + r) * stride + o + m] = sr[m][r];
        qlm_i[(row0 + r) * stride + o + m] = si[m][r];
    }
}

// Stage 1 for TWO degrees LA < LB in one launch (q4 and q6, the usual pair): the row, the neighbours' positions and the bond's
// geometry are read and computed once, and the recurrence of assoc_legendre(LB, m, x) passes through P_LA^m on its way up — the same
// operations in the same order as assoc_legendre(LA, m, x) performs, so both degrees' terms, and their sums bond by bond in list
// order, are bit for bit those of two k_sq_stage1_l launches.  (Each of those took 2.8 ms at 10 M atoms whatever the degree: bound
// by the gathers and the q_lm traffic, not by the harmonics.)
template <int LA, int LB>
__device__ __forceinline__ void legendre_pair(int m, double x, double sqx, double &pa, double &pb)
{
    double p = 1.0, pm1 = 0.0, pm2 = 0.0;
    if (m != 0) {
        for (int i = 1; i < m + 1; ++i)
            p *= (2 * i - 1) * sqx;
    }
    pa = p; // (m == LA: the loop below does not reach LA)
#pragma unroll
    for (int i = m + 1; i < LB + 1; ++i) {
        pm2 = pm1;
        pm1 = p;
        p = ((2 * i - 1) * x * pm1 - (i + m - 1) * pm2) / (i - m);
        if (i == LA) pa = p;
    }
    pb = p;
}
template <bool TRI, int LA, int LB>
__global__ __launch_bounds__(64) void k_sq_stage1_pair(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                       int64_t N, DBox b, const int *__restrict__ NL, const double *__restrict__ DL, int64_t M,
                                                       const int *__restrict__ NN, const double *__restrict__ weight, int ila, int ilb, int stride,
                                                       int nz, int lmax, int nnn, int use_voronoi, double rc, int use_weight,
                                                       const double *__restrict__ norm, double *__restrict__ qlm_r, double *__restrict__ qlm_i,
                                                       double *__restrict__ qn, int ncol)
{
    constexpr int NA = 2 * LA + 1, NB = 2 * LB + 1, NM = NA + NB;
    // (one staging buffer used twice — real parts, then imaginary ones: half the LDS, eleven instead of six wavefronts per CU — was
    // measured SLOWER, 4.7 against 4.0 ms at 10 M atoms: the two extra barrier pairs cost more than the occupancy buys)
    __shared__ double sr[NM][65], si[NM][65];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + t;
    const int rows = (int)(N - row0 < 64 ? N - row0 : 64);
    const int oa = ila * nz, ob = ilb * nz;
    { // (unrolled: the 2 NM loads in flight together instead of NM dependent latencies, see k_sq_stage1_l)
        double gr[NM], gi[NM];
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            const int e = t + 64 * q;
            const int r = e / NM, m = e - r * NM;
            const int64_t at = (row0 + r) * stride + (m < NA ? oa + m : ob + (m - NA));
            const bool in = e < rows * NM;
            gr[q] = in ? qlm_r[at] : 0.0; // the caller's (pre-zeroed) content: the reference adds onto it
            gi[q] = in ? qlm_i[at] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            const int e = t + 64 * q;
            const int r = e / NM, m = e - r * NM;
            if (e < rows * NM) { sr[m][r] = gr[q]; si[m][r] = gi[q]; }
        }
    }
    __syncthreads();
    if (i < N) {
        double ar[NM], ai[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) { ar[m] = sr[m][t]; ai[m] = si[m][t]; }
        double na[LA + 1], nb[LB + 1];
#pragma unroll
        for (int m = 0; m <= LA; ++m) na[m] = norm[ila * (lmax + 1) + m];
#pragma unroll
        for (int m = 0; m <= LB; ++m) nb[m] = norm[ilb * (lmax + 1) + m];
        const double EPS = 1e-15;
        const double x1 = x[i], y1 = y[i], z1 = z[i];
        int cnt = NN[i];
        if (!use_voronoi && nnn > 0)
            cnt = nnn;
        double wsum = 0.0;
        int cj[4];
        double cr_[4], cw[4], cx[4], cy[4], cz[4];
        for (int jj = 0; jj < cnt; ++jj) {
            const int u = jj & 3;
            if (u == 0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int64_t idx = i * M + min(jj + v, cnt - 1);
                    cj[v] = NL[idx];
                    cr_[v] = DL[idx];
                    cw[v] = use_weight ? weight[idx] : 1.0;
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int js = (unsigned)cj[v] < (unsigned)N ? cj[v] : (int)i;
                    cx[v] = x[js]; cy[v] = y[js]; cz[v] = z[js];
                }
            }
            int j = cj[0];
            double r = cr_[0], w = cw[0], xj = cx[0], yj = cy[0], zj = cz[0];
#pragma unroll
            for (int v = 1; v < 4; ++v)
                if (u == v) { j = cj[v]; r = cr_[v]; w = cw[v]; xj = cx[v]; yj = cy[v]; zj = cz[v]; }
            if ((unsigned)j >= (unsigned)N)
                continue;
            double dx = xj - x1, dy = yj - y1, dz = zj - z1;
            pbc<TRI>(b, dx, dy, dz);
            if (!((r > EPS) && (r <= rc)))
                continue;
            wsum += w;
            const double rinv = 1.0 / r;
            const double ct = dz * rinv;
            double er = dx, ei = dy;
            const double rxy2 = er * er + ei * ei;
            if (rxy2 < EPS * EPS) { er = 1.0; ei = 0.0; }
            else { const double sc = 1.0 / sqrt(rxy2); er *= sc; ei *= sc; }
            const double sqx = sqrt(1.0 - ct * ct); // (assoc_legendre computes it for every m != 0: the same value)
            double pa, pb;
            legendre_pair<LA, LB>(0, ct, sqx, pa, pb);
            ar[LA] += w * (na[0] * pa);
            ar[NA + LB] += w * (nb[0] * pb);
            double mr = er, mi = ei;
#pragma unroll
            for (int m = 1; m < LB + 1; ++m) {
                legendre_pair<LA, LB>(m, ct, sqx, pa, pb);
                if (m <= LA) {
                    const double pf = na[m] * pa;
                    const double cr = pf * mr, ci = pf * mi;
                    const double wr = w * cr, wi = w * ci;
                    ar[LA + m] += wr;
                    ai[LA + m] += wi;
                    if (m & 1) { ar[LA - m] -= wr; ai[LA - m] += wi; }
                    else { ar[LA - m] += wr; ai[LA - m] -= wi; }
                }
                {
                    const double pf = nb[m] * pb;
                    const double cr = pf * mr, ci = pf * mi;
                    const double wr = w * cr, wi = w * ci;
                    ar[NA + LB + m] += wr;
                    ai[NA + LB + m] += wi;
                    if (m & 1) { ar[NA + LB - m] -= wr; ai[NA + LB - m] += wi; }
                    else { ar[NA + LB - m] += wr; ai[NA + LB - m] -= wi; }
                }
                const double tr = mr * er - mi * ei, ti = mr * ei + mi * er;
                mr = tr; mi = ti;
            }
        }
        const double f = 1.0 / wsum;
#pragma unroll
        for (int m = 0; m < NM; ++m) { sr[m][t] = ar[m] * f; si[m][t] = ai[m] * f; }
        if (qn) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int m = 0; m < NA; ++m) {
                const double vr = ar[m] * f, vi = ai[m] * f;
                s1 += vr * vr + vi * vi;
            }
#pragma unroll
            for (int m = NA; m < NM; ++m) {
                const double vr = ar[m] * f, vi = ai[m] * f;
                s2 += vr * vr + vi * vi;
            }
            const volatile int la_rt = LA, lb_rt = LB; // (the factor as the kernel of stage 3 computes it, from a degree known at run time)
            qn[i * ncol + ila] = sqrt(4 * MY_PI / (2 * la_rt + 1)) * sqrt(s1);
            qn[i * ncol + ilb] = sqrt(4 * MY_PI / (2 * lb_rt + 1)) * sqrt(s2);
        }
    }
    __syncthreads();
    for (int e = t; e < rows * NM; e += 64) {
        const int r = e / NM, m = e - r * NM;
        const int64_t at = (row0 + r) * stride + (m < NA ? oa + m : ob + (m - NA));
        qlm_r[at] = sr[m][r];
        qlm_i[at] = si[m][r];
    }
}

// Stage 1 with SIXTEEN LANES PER ATOM (four atoms to a wavefront) — north_star's "one wavefront per atom" shape in the form that
// wastes the fewest lanes on a 12-bond row: lane b of an atom's group takes the bonds b, b + 16, ... of the row (one bond each for
// the rows of the k-nearest and first-shell lists), evaluates its bond's 2l+1 terms exactly as the lane-per-atom kernel does, and
// the group adds the bonds up with four butterfly steps per component.  Sums run in the butterfly's order, not the list's: equal
// to the lane-per-atom kernel to rounding (~1e-16 relative), not bit for bit.  A MEASURING variant (mdh_debug_set_sq_variant(2),
// tools/sq_wave_ab.py -> profiles/r05_sq_wave.txt): the product path is k_sq_stage1_l.
template <bool TRI, int L>
__global__ __launch_bounds__(256) void k_sq_stage1_g16(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                       int64_t N, DBox b, const int *__restrict__ NL, const double *__restrict__ DL, int64_t M,
                                                       const int *__restrict__ NN, const double *__restrict__ weight, int il, int stride, int nz,
                                                       int lmax, int nnn, int use_voronoi, double rc, int use_weight,
                                                       const double *__restrict__ norm, double *__restrict__ qlm_r, double *__restrict__ qlm_i,
                                                       double *__restrict__ qn, int ncol, int lrt)
{
    constexpr int NM = 2 * L + 1;
    const int sub = threadIdx.x & 15;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= N)
        return; // (whole groups leave: the butterflies below stay inside a group of 16 lanes)
    const int o = il * nz;
    double ar[NM], ai[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) { ar[m] = 0.0; ai[m] = 0.0; }
    double nrm[L + 1];
#pragma unroll
    for (int m = 0; m <= L; ++m) nrm[m] = norm[il * (lmax + 1) + m];
    const double EPS = 1e-15;
    const double x1 = x[i], y1 = y[i], z1 = z[i];
    int cnt = NN[i];
    if (!use_voronoi && nnn > 0)
        cnt = nnn;
    double wsum = 0.0;
    for (int jj = sub; jj < cnt; jj += 16) {
        const int64_t idx = i * M + jj;
        const int j = NL[idx];
        const double r = DL[idx], w = use_weight ? weight[idx] : 1.0;
        if ((unsigned)j >= (unsigned)N)
            continue;
        double dx = x[j] - x1, dy = y[j] - y1, dz = z[j] - z1;
        pbc<TRI>(b, dx, dy, dz);
        if (!((r > EPS) && (r <= rc)))
            continue;
        wsum += w;
        const double rinv = 1.0 / r;
        const double ct = dz * rinv;
        double er = dx, ei = dy;
        const double rxy2 = er * er + ei * ei;
        if (rxy2 < EPS * EPS) { er = 1.0; ei = 0.0; }
        else { const double sc = 1.0 / sqrt(rxy2); er *= sc; ei *= sc; }
        ar[L] += w * (nrm[0] * assoc_legendre(L, 0, ct));
        double mr = er, mi = ei;
#pragma unroll
        for (int m = 1; m < L + 1; ++m) {
            const double pf = nrm[m] * assoc_legendre(L, m, ct);
            const double cr = pf * mr, ci = pf * mi;
            const double wr = w * cr, wi = w * ci;
            ar[L + m] += wr;
            ai[L + m] += wi;
            if (m & 1) { ar[L - m] -= wr; ai[L - m] += wi; }
            else { ar[L - m] += wr; ai[L - m] -= wi; }
            const double tr = mr * er - mi * ei, ti = mr * ei + mi * er;
            mr = tr; mi = ti;
        }
    }
    // the group's bonds added up: four butterfly steps (every lane ends with the totals)
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
        wsum += __shfl_xor(wsum, d, 16);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            ar[m] += __shfl_xor(ar[m], d, 16);
            ai[m] += __shfl_xor(ai[m], d, 16);
        }
    }
    const double f = 1.0 / wsum;
    double mine_r = 0.0, mine_i = 0.0; // lane m of the group owns component m (NM <= 13 components for l <= 6)
#pragma unroll
    for (int m = 0; m < NM; ++m)
        if (sub == m) { mine_r = ar[m]; mine_i = ai[m]; }
    if (sub < NM) {
        const int64_t at = i * stride + o + sub;
        qlm_r[at] = (qlm_r[at] + mine_r) * f; // (onto the caller's pre-zeroed content, as the reference adds)
        qlm_i[at] = (qlm_i[at] + mine_i) * f;
    }
    if (qn && sub == 0) {
        double sacc = 0.0;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const double vr = ar[m] * f, vi = ai[m] * f;
            sacc += vr * vr + vi * vi;
        }
        qn[i * ncol + il] = sqrt(4 * MY_PI / (2 * lrt + 1)) * sqrt(sacc);
    }
}

template <bool TRI>
static bool launch_stage1_l(int l, dim3 grid, hipStream_t st, const double *dx, const double *dy, const double *dz, int64_t N, const DBox &b,
                            const int *dv, const double *dd, int64_t M, const int *dn, const double *dw, int il, int stride, int nz, int lmax,
                            int nnn, int use_voronoi, double rc, int use_weight, const double *dnorm, double *dqr, double *dqi, double *dqn, int ncol)
{
#define MDH_SQ_L(LL)                                                                                                                     \
    case LL:                                                                                                                             \
        hipLaunchKernelGGL((k_sq_stage1_l<TRI, LL>), grid, dim3(64), 0, st, dx, dy, dz, N, b, dv, dd, M, dn, dw, il, stride, nz, lmax, nnn, \
                           use_voronoi, rc, use_weight, dnorm, dqr, dqi, dqn, ncol, LL);                                                 \
        return true;
    if (g_sq_variant == 2 && (l == 4 || l == 6)) { // measuring variant: sixteen lanes per atom
        const dim3 g16(grid_for(N, 16));
        if (l == 4)
            hipLaunchKernelGGL((k_sq_stage1_g16<TRI, 4>), g16, dim3(256), 0, st, dx, dy, dz, N, b, dv, dd, M, dn, dw, il, stride, nz, lmax, nnn, use_voronoi, rc,
                               use_weight, dnorm, dqr, dqi, dqn, ncol, 4);
        else
            hipLaunchKernelGGL((k_sq_stage1_g16<TRI, 6>), g16, dim3(256), 0, st, dx, dy, dz, N, b, dv, dd, M, dn, dw, il, stride, nz, lmax, nnn, use_voronoi, rc,
                               use_weight, dnorm, dqr, dqi, dqn, ncol, 6);
        return true;
    }
    switch (l) {
        MDH_SQ_L(2) MDH_SQ_L(3) MDH_SQ_L(4) MDH_SQ_L(5) MDH_SQ_L(6) MDH_SQ_L(7) MDH_SQ_L(8) MDH_SQ_L(10) MDH_SQ_L(12)
    default:
        return false;
    }
#undef MDH_SQ_L
}

// stage 2 (:439-503): one thread per (atom, component); neighbours added in list order
__global__ __launch_bounds__(256) void k_sq_average(int64_t N, const int *__restrict__ NL, int64_t M,
                                                    const int *__restrict__ NN, LList ll, int nnn, int lmax,
                                                    int use_voronoi, const double *__restrict__ ar,
                                                    const double *__restrict__ ai, double *__restrict__ qlm_r,
                                                    double *__restrict__ qlm_i)
{
    const int nz = 2 * lmax + 1, stride = ll.n * nz;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N * stride)
        return;
    const int64_t i = g / stride;
    const int c = (int)(g % stride);
    const int il = c / nz, m = c % nz;
    if (m >= 2 * ll.l[il] + 1)
        return;
    int cnt = NN[i];
    if (!use_voronoi && nnn > 0)
        cnt = nnn;
    double sr = qlm_r[g], si = qlm_i[g];
    int nb = 1;
    for (int j0 = 0; j0 < cnt; j0 += 4) { // four entries at a time: ids, then the four neighbours' components, in flight together
        int js[4];
        double br[4], bi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            js[u] = NL[i * M + min(j0 + u, cnt - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t jj = (unsigned)js[u] < (unsigned)N ? js[u] : i;
            br[u] = ar[jj * stride + c];
            bi[u] = ai[jj * stride + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + u >= cnt || (unsigned)js[u] >= (unsigned)N)
                continue;
            sr += br[u];
            si += bi[u];
            ++nb;
        }
    }
    const double inv = 1.0 / nb;
    qlm_r[g] = sr * inv;
    qlm_i[g] = si * inv;
}

// stage 3 (:506-575).  STAGED: the q_lm rows of the workgroup's 64 atoms are copied into LDS with coalesced reads first
template <bool STAGED>
__global__ __launch_bounds__(256) void k_sq_final(int64_t N, LList ll, int lmax, int wl, int wlhat,
                                                  const double *__restrict__ cg, const double *__restrict__ qlm_r,
                                                  const double *__restrict__ qlm_i, double *__restrict__ qn, int ncol)
{
    extern __shared__ double rows[];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nz = 2 * lmax + 1, nl = ll.n;
    const int st = nl * nz, bd = blockDim.x + 1, t = threadIdx.x;
    if (STAGED) {
        const int64_t row0 = (int64_t)blockIdx.x * blockDim.x;
        const int64_t nelem = (N - row0 < (int64_t)blockDim.x ? N - row0 : (int64_t)blockDim.x) * st;
        for (int64_t e = t; e < nelem; e += blockDim.x) {
            const int r = (int)(e / st), c = (int)(e - (int64_t)r * st);
            rows[c * bd + r] = qlm_r[row0 * st + e];
            rows[(st + c) * bd + r] = qlm_i[row0 * st + e];
        }
        __syncthreads();
    }
    if (i >= N)
        return;
    struct Row {
        const double *g, *l;
        int bd, t;
        __device__ __forceinline__ double operator[](int c) const { return STAGED ? l[c * bd + t] : g[c]; }
    };
    const Row qr{qlm_r + i * nl * nz, rows, bd, t}, qi{qlm_i + i * nl * nz, rows + (size_t)st * bd, bd, t};
    double *out = qn + i * ncol;
    const double EPS = 1e-15;
    for (int il = 0; il < nl; ++il) {
        const int l = ll.l[il];
        const double nf = sqrt(4 * MY_PI / (2 * l + 1));
        double s = 0.0;
        for (int m = 0; m < 2 * l + 1; ++m)
            s += qr[il * nz + m] * qr[il * nz + m] + qi[il * nz + m] * qi[il * nz + m];
        out[il] = nf * sqrt(s);
    }
    if (wl | wlhat) {
        int c = 0;
        for (int il = 0; il < nl; ++il) {
            const int l = ll.l[il];
            const int off = il * nz;
            double ws = 0.0;
            for (int m1 = 0; m1 < 2 * l + 1; ++m1) {
                const int lo = (l - m1) > 0 ? (l - m1) : 0;
                const int hi = (2 * l + 1) < (3 * l - m1 + 1) ? (2 * l + 1) : (3 * l - m1 + 1);
                for (int m2 = lo; m2 < hi; ++m2) {
                    const int m = m1 + m2 - l;
                    const double a_r = qr[off + m1] * qr[off + m2] - qi[off + m1] * qi[off + m2];
                    const double a_i = qr[off + m1] * qi[off + m2] + qi[off + m1] * qr[off + m2];
                    ws += (a_r * qr[off + m] + a_i * qi[off + m]) * cg[c];
                    ++c;
                }
            }
            const double wf = ws / sqrt(2 * l + 1.0);
            if (wl)
                out[il + nl] = wf;
            if (wlhat) {
                const double q = out[il];
                if (q > EPS) {
                    const double nf = sqrt(4 * MY_PI / (2 * l + 1));
                    const double gfac = nf / q;
                    out[il + (wl ? nl : 0) + nl] = wf * (gfac * gfac * gfac);
                }
            }
        }
    }
}

// identifySolidLiquid pass 1 (:605-643)
__global__ __launch_bounds__(256) void k_solid_bonds(int q6index, const double *__restrict__ Q6,
                                                     const int *__restrict__ verlet, const double *__restrict__ dist,
                                                     const int *__restrict__ nn, int64_t N, int64_t M,
                                                     const double *__restrict__ qlm_r, const double *__restrict__ qlm_i,
                                                     int nl, int nz, double threshold, int n_bond,
                                                     int *__restrict__ solid, int *__restrict__ nbond, int use_voronoi,
                                                     int nnn, double rc)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    const int64_t stride = (int64_t)nl * nz;
    int cnt = nn[i], nsb = 0;
    if (!use_voronoi && nnn > 0)
        cnt = nnn;
    // the atom's own q_6m once, in registers; the row eight entries at a time (ids, distances, the neighbours' Q6 in flight
    // together); a neighbour's 26 components are then one group of loads.  Entry by entry every neighbour cost three
    // dependent memory latencies (id -> Q6 / components -> own components re-read).
    double ar[13], ai[13];
#pragma unroll
    for (int m = 0; m < 13; ++m) {
        ar[m] = qlm_r[i * stride + q6index * nz + m];
        ai[m] = qlm_i[i * stride + q6index * nz + m];
    }
    const double q6i = Q6[i];
    for (int j0 = 0; j0 < cnt; j0 += 8) {
        int js[8];
        double ds[8], qj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t idx = i * M + min(j0 + u, cnt - 1);
            js[u] = verlet[idx];
            ds[u] = dist[idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            qj[u] = Q6[(unsigned)js[u] < (unsigned)N ? js[u] : (int)i];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (j0 + u >= cnt) continue;
            const int j = js[u];
            if ((unsigned)j >= (unsigned)N) continue;
            if (ds[u] > rc) continue;
            const double *br = qlm_r + (int64_t)j * stride + q6index * nz, *bi = qlm_i + (int64_t)j * stride + q6index * nz;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < 13; ++m)
                s += ar[m] * br[m] + ai[m] * bi[m];
            s = s / q6i / qj[u] * 4 * MY_PI / 13;
            if (s > threshold) ++nsb;
        }
    }
    if (nsb >= n_bond) solid[i] = 1;
    nbond[i] = nsb;
}

// pass 2 (:645-674): a solid atom without any solid neighbour becomes liquid.  The reference updates
// the labels in place from concurrent threads; here every atom is judged against the pass-1 labels.
__global__ __launch_bounds__(256) void k_solid_cleanup(const int *__restrict__ verlet, const int *__restrict__ nn,
                                                       int64_t N, int64_t M, const int *__restrict__ snap,
                                                       int *__restrict__ solid, int use_voronoi, int nnn)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || snap[i] != 1)
        return;
    int cnt = nn[i];
    if (!use_voronoi && nnn > 0)
        cnt = nnn;
    for (int jj = 0; jj < cnt; ++jj) {
        const int j = verlet[i * M + jj];
        if ((unsigned)j >= (unsigned)N) continue;
        if (snap[j] == 1) return;
    }
    solid[i] = 0;
}

// host tables ------------------------------------------------------------------
static void factorials(double *f) // h_factorial, :12-181 (n! rounded to double)
{
    long double v = 1.0L;
    f[0] = 1.0;
    for (int n = 1; n < 168; ++n) { v *= (long double)n; f[n] = (double)v; }
}

static void clebsch_gordan(std::vector<double> &cg, const LList &ll) // :188-224
{
    double F[168];
    factorials(F);
    cg.clear();
    for (int il = 0; il < ll.n; ++il) {
        const int l = ll.l[il];
        for (int m1 = 0; m1 < 2 * l + 1; ++m1) {
            const int aa2 = m1 - l;
            for (int m2 = std::max(0, l - m1); m2 < std::min(2 * l + 1, 3 * l - m1 + 1); ++m2) {
                const int bb2 = m2 - l, m = aa2 + bb2 + l;
                double sums = 0.0;
                for (int zz = std::max(0, std::max(-aa2, bb2)); zz < std::min(l, std::min(l - aa2, l + bb2)) + 1; ++zz) {
                    const int ifac = (zz % 2) ? -1 : 1;
                    sums += ifac / (F[zz] * F[l - zz] * F[l - aa2 - zz] * F[l + bb2 - zz] * F[aa2 + zz] * F[-bb2 + zz]);
                }
                const int cc2 = m - l;
                const double sfaccg = std::sqrt(F[l + aa2] * F[l - aa2] * F[l + bb2] * F[l - bb2] * F[l + cc2] * F[l - cc2] * (2 * l + 1));
                const double sfac1 = F[3 * l + 1], sfac2 = F[l];
                const double dcg = std::sqrt(sfac2 * sfac2 * sfac2 / sfac1);
                cg.push_back(sums * dcg * sfaccg);
            }
        }
    }
    if (cg.empty()) cg.push_back(0.0);
}

} // namespace mdh

using namespace mdh;

extern "C" {

int mdh_debug_set_sq_variant(int v)
{
    g_sq_variant = v;
    return MDH_OK;
}

int mdh_get_sq(const double *x, const double *y, const double *z, int64_t N, const double *box9,
               const double *origin3, const int *boundary3, const int *verlet, const double *dist, int64_t M,
               const int *nn, const double *weight, const int *llist_host, int nl, int nnn, int lmax, int wl,
               int wlhat, int average, int use_voronoi, double rc, int use_weight, double *qlm_r, double *qlm_i,
               double *qnarray, int space, void *stream)
{
    if (N < 0 || M <= 0 || nl <= 0 || nl > SBO_MAXL || lmax < 0 || lmax > SBO_LMAX) {
        set_error("mdh_get_sq: need 1 <= len(llist) <= 16 and lmax <= 40");
        return MDH_ERR_ARG;
    }
    LList ll;
    ll.n = nl;
    for (int k = 0; k < SBO_MAXL; ++k) ll.l[k] = 0;
    for (int k = 0; k < nl; ++k) {
        ll.l[k] = llist_host[k];
        if (ll.l[k] < 0 || ll.l[k] > lmax) { set_error("mdh_get_sq: every degree must satisfy 0 <= l <= lmax"); return MDH_ERR_ARG; }
    }
    if (use_weight && !weight) { set_error("mdh_get_sq: use_weight without weight array"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    const int nz = 2 * lmax + 1;
    const int64_t stride = (int64_t)nl * nz;
    const int ncol = nl + (wl ? nl : 0) + (wlhat ? nl : 0);
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const double *dd = sc.stage_in(dist, (size_t)(N * M), space);
    const int *dn = sc.stage_in(nn, (size_t)N, space);
    const double *dw = use_weight ? sc.stage_in(weight, (size_t)(N * M), space) : nullptr;
    double *dqr = sc.stage(qlm_r, (size_t)(N * stride), space, true, true);
    double *dqi = sc.stage(qlm_i, (size_t)(N * stride), space, true, true);
    // stage 3 leaves some w-hat entries untouched (q <= eps), so the caller's content is kept
    double *dqn = sc.stage(qnarray, (size_t)(N * ncol), space, true, true);

    // (l,m) normalisation  sqrt((2l+1)/(4 pi prod_{i=l-m+1}^{l+m} i))   (:270-279)
    std::vector<double> norm((size_t)nl * (lmax + 1), 0.0);
    for (int il = 0; il < nl; ++il)
        for (int m = 0; m <= ll.l[il]; ++m) {
            const int l = ll.l[il];
            double pf = 1.0;
            for (int i = l - m + 1; i < l + m + 1; ++i)
                pf *= i;
            norm[(size_t)il * (lmax + 1) + m] = std::sqrt((2 * l + 1) / (4 * MY_PI * pf));
        }
    std::vector<double> cg;
    if (wl || wlhat) clebsch_gordan(cg, ll);
    else cg.push_back(0.0);
    double *dnorm = sc.alloc_n<double>(norm.size());
    double *dcg = sc.alloc_n<double>(cg.size());
    double *ar = average ? sc.alloc_n<double>((size_t)(N * stride)) : nullptr;
    double *ai = average ? sc.alloc_n<double>((size_t)(N * stride)) : nullptr;
    if (sc.failed())
        return sc.error();
    MDH_HIP(hipMemcpyAsync(dnorm, norm.data(), norm.size() * sizeof(double), hipMemcpyHostToDevice, st));
    MDH_HIP(hipMemcpyAsync(dcg, cg.data(), cg.size() * sizeof(double), hipMemcpyHostToDevice, st));
    MDH_HIP(hipStreamSynchronize(st)); // norm/cg are host temporaries

    // degrees with a compiled instantiation: one register-resident launch per entry of llist; any other degree in the list
    // sends the whole call through the generic kernel (LDS accumulators)
    bool special = g_sq_variant != 1;
    for (int k = 0; k < nl && special; ++k) {
        const int l = ll.l[k];
        special = (l >= 2 && l <= 8) || l == 10 || l == 12;
    }
    // no averaging, no w_l: q_l leaves the stage-1 kernels, stage 3 is not launched (2.1 of 8.2 ms for q4 + q6 of 10 M atoms)
    const bool fused_final = special && !average && !wl && !wlhat;
    double *fq = fused_final ? dqn : nullptr;
    if (special && g_sq_variant == 0 && nl == 2 && ll.l[0] == 4 && ll.l[1] == 6) { // q4 and q6: both degrees in one launch
        const dim3 grid(grid_for(N, 64));
        if (b.tri)
            hipLaunchKernelGGL((k_sq_stage1_pair<true, 4, 6>), grid, dim3(64), 0, st, dx, dy, dz, N, b, dv, dd, M, dn, dw, 0, 1, (int)stride, 2 * lmax + 1, lmax, nnn,
                               use_voronoi, rc, use_weight, dnorm, dqr, dqi, fq, ncol);
        else
            hipLaunchKernelGGL((k_sq_stage1_pair<false, 4, 6>), grid, dim3(64), 0, st, dx, dy, dz, N, b, dv, dd, M, dn, dw, 0, 1, (int)stride, 2 * lmax + 1, lmax, nnn,
                               use_voronoi, rc, use_weight, dnorm, dqr, dqi, fq, ncol);
    } else if (special) {
        const dim3 grid(grid_for(N, 64));
        for (int k = 0; k < nl; ++k) {
            if (b.tri)
                launch_stage1_l<true>(ll.l[k], grid, st, dx, dy, dz, N, b, dv, dd, M, dn, dw, k, (int)stride, 2 * lmax + 1, lmax, nnn, use_voronoi, rc, use_weight, dnorm, dqr, dqi, fq, ncol);
            else
                launch_stage1_l<false>(ll.l[k], grid, st, dx, dy, dz, N, b, dv, dd, M, dn, dw, k, (int)stride, 2 * lmax + 1, lmax, nnn, use_voronoi, rc, use_weight, dnorm, dqr, dqi, fq, ncol);
        }
    } else {
    // block size so that 2*stride doubles per thread fit in 64 KiB of LDS
        int bd = (int)(65536 / (16 * stride)) / 64 * 64;
        if (bd > 256) bd = 256;
        if (bd >= 64) {
            const size_t lds = (size_t)bd * 16 * (size_t)stride;
            if (b.tri)
                hipLaunchKernelGGL((k_sq_stage1<true, true>), dim3(grid_for(N, bd)), dim3(bd), lds, st, dx, dy, dz, N, b, dv, dd, M, dn, dw, ll, nnn, lmax, use_voronoi, rc, use_weight, dnorm, dqr, dqi);
            else
                hipLaunchKernelGGL((k_sq_stage1<false, true>), dim3(grid_for(N, bd)), dim3(bd), lds, st, dx, dy, dz, N, b, dv, dd, M, dn, dw, ll, nnn, lmax, use_voronoi, rc, use_weight, dnorm, dqr, dqi);
        } else {
            if (b.tri)
                hipLaunchKernelGGL((k_sq_stage1<true, false>), dim3(grid_for(N, 64)), dim3(64), 0, st, dx, dy, dz, N, b, dv, dd, M, dn, dw, ll, nnn, lmax, use_voronoi, rc, use_weight, dnorm, dqr, dqi);
            else
                hipLaunchKernelGGL((k_sq_stage1<false, false>), dim3(grid_for(N, 64)), dim3(64), 0, st, dx, dy, dz, N, b, dv, dd, M, dn, dw, ll, nnn, lmax, use_voronoi, rc, use_weight, dnorm, dqr, dqi);
        }
    }
    if (average) {
        MDH_HIP(hipMemcpyAsync(ar, dqr, sizeof(double) * (size_t)(N * stride), hipMemcpyDeviceToDevice, st));
        MDH_HIP(hipMemcpyAsync(ai, dqi, sizeof(double) * (size_t)(N * stride), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_sq_average, dim3(grid_for(N * stride, 256)), dim3(256), 0, st, N, dv, M, dn, ll, nnn, lmax, use_voronoi, ar, ai, dqr, dqi);
    }
    if (fused_final)
        ;
    else if ((size_t)65 * 16 * (size_t)stride <= 60 * 1024)
        hipLaunchKernelGGL(k_sq_final<true>, dim3(grid_for(N, 64)), dim3(64), (size_t)65 * 16 * (size_t)stride, st, N, ll, lmax, wl, wlhat, dcg, dqr, dqi, dqn, ncol);
    else
        hipLaunchKernelGGL(k_sq_final<false>, dim3(grid_for(N, 256)), dim3(256), 0, st, N, ll, lmax, wl, wlhat, dcg, dqr, dqi, dqn, ncol);
    return sc.finish(space);
}

int mdh_identify_solid_liquid(int q6index, const double *Q6, const int *verlet, const double *dist, const int *nn,
                              int64_t N, int64_t M, const double *qlm_r, const double *qlm_i, int nl, int nz,
                              double threshold, int n_bond, int *solidliquid, int *nbond, int use_voronoi, int nnn,
                              double rc, int space, void *stream)
{
    if (N < 0 || M <= 0 || nl <= 0 || nz < 13 || q6index < 0 || q6index >= nl) { set_error("mdh_identify_solid_liquid: invalid argument"); return MDH_ERR_ARG; }
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const int64_t stride = (int64_t)nl * nz;
    const double *dq6 = sc.stage_in(Q6, (size_t)N, space);
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const double *dd = sc.stage_in(dist, (size_t)(N * M), space);
    const int *dn = sc.stage_in(nn, (size_t)N, space);
    const double *dqr = sc.stage_in(qlm_r, (size_t)(N * stride), space);
    const double *dqi = sc.stage_in(qlm_i, (size_t)(N * stride), space);
    int *ds = sc.stage(solidliquid, (size_t)N, space, true, true);
    int *db = sc.stage(nbond, (size_t)N, space, false, true);
    int *snap = sc.alloc_n<int>((size_t)N);
    if (sc.failed())
        return sc.error();
    hipLaunchKernelGGL(k_solid_bonds, dim3(grid_for(N, 256)), dim3(256), 0, st, q6index, dq6, dv, dd, dn, N, M, dqr, dqi, nl, nz, threshold, n_bond, ds, db, use_voronoi, nnn, rc);
    MDH_HIP(hipMemcpyAsync(snap, ds, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_solid_cleanup, dim3(grid_for(N, 256)), dim3(256), 0, st, dv, dn, N, M, snap, ds, use_voronoi, nnn);
    return sc.finish(space);
}
}

MDH_WARM_UNIT(sbo)
