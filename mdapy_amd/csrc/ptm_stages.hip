// ptm_stages.hip — polyhedral template matching as a pipeline of small kernels (gfx950).
//
// Same per-atom algorithm as ptm_core.hpp (which remains the host twin and the table generator; it cites the pieces of
// extern/ptm it restates).  What changes is how the work sits on the machine.  The monolithic kernel kept ~3.7 KB of
// dynamically indexed private arrays per lane (scratch memory: every access a trip to the L2) and, worse, let the 64
// atoms of a wavefront walk different branches of nested data-dependent loops: the canonical form tried 60-72 start
// edges one after the other although a lane needs 1-3 of them, the graph look-up serialised over the table's graphs.
// Here the atom stays on ONE LANE (its work is a serial chain; a wavefront per atom would idle 60 lanes for most of it —
// DESIGN.md 3c has the numbers), but
//   * the work is cut into stages with a small working set each: hull -> canonical form -> template match.  A stage keeps
//     its dynamically indexed state in a per-lane stripe of LDS (element e of lane l at [e * 64 + l]: bank = lane,
//     conflict-free for any index pattern), everything else in registers: no scratch memory at all;
//   * loops over data-dependent work lists are written with a per-lane cursor, so that all lanes of a wavefront run
//     their t-th start edge / t-th automorphism together;
//   * facet normals are not stored (672 B per lane); they are recomputed from the three vertices when a facet is tested.
//     A cheap unnormalised test decides whenever the point is farther than 1e-9 from the plane; only closer calls
//     (coplanar template faces of a perfect crystal) evaluate the reference's normalised expression, bit for bit.
// Stages hand over through HBM in lane-major (structure-of-arrays) buffers: ~500 B per atom, read and written once.
#include "common.hpp"
#include "ptm_core.hpp"
#include <type_traits>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <vector>

namespace mdh {
namespace ptms {

using ptmc::Tables;
constexpr int BLK = 64;
constexpr int NROW = 18;        // neighbours per atom handed over by the ordering pass
constexpr int MAXF = ptmc::MAX_FACETS;

enum { K_SC = 0, K_FCC = 1, K_BCC = 2, NKIND = 3 };
__host__ __device__ constexpr int kind_points(int k) { return k == K_SC ? 7 : k == K_FCC ? 13 : 15; }

template <bool TRI> __device__ __forceinline__ void fold(const DBox &b, double &dx, double &dy, double &dz) { pbc<TRI>(b, dx, dy, dz); }

// ---------------------------------------------------------------------------------------------------------------------
// stage 0: order every atom's row by Voronoi-face solid angle (src/polyhedral_template_matching.cpp:215-255, ptm_core.hpp
// order_neighbours) — ONE FACE PER LANE.  The 18 faces of a cell are independent until the final sort, so a lane clips one
// face's polygon (its own LDS stripe) against the cell's other bisector planes (read from the atom's shared LDS copy of the
// row); the sort is a rank count over the 18 areas.  7 atoms (126 lanes) per 128-thread workgroup.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ORD_THREADS = 128, ORD_APB = ORD_THREADS / NROW; // 7 atoms per workgroup

template <int CAP_, int DIM> struct PolyStripe { // DIM 2: vertices in the coordinates of the face's plane (ptm_core.hpp face_solid_angle_2d)
    static constexpr int CAP = CAP_;
    double *base;
    __device__ __forceinline__ double get(int i, int c) const { return base[(i * DIM + c) * ORD_THREADS]; }
    __device__ __forceinline__ void set(int i, int c, double v) { base[(i * DIM + c) * ORD_THREADS] = v; }
};

struct OrderShared { // per workgroup
    double pts[ORD_APB][NROW][3];
    double nsq[ORD_APB][NROW];
    double area[ORD_APB][NROW];
    int ids[ORD_APB][NROW];
    int raw[ORD_APB][NROW];
    int cnt[ORD_APB];
    int overflow[ORD_APB];
    int big[ORD_APB]; // a face of the atom had more than eight vertices at some point (statistic for the choice of the first pass)
};
template <int CAP, int DIM> constexpr size_t order_lds_bytes() { return sizeof(OrderShared) + (size_t)ORD_THREADS * CAP * DIM * 8; }

// face_solid_angle_2d (ptm_core.hpp) with the polygon's vertices CACHED IN REGISTERS.  The plain form keeps the polygon in the
// lane's LDS stripe only, so the counting pass of every clip — is vertex t beyond the plane? — is a run-time loop of one LDS round
// trip and ~30 instructions (address arithmetic, loop and exec-mask bookkeeping around four multiplications) per vertex, and 13
// of the 17 planes cut nothing.  Here the CAP vertex slots are a register array read with compile-time indices: the counting
// pass is straight-line code, five instructions per slot, no memory access (unused slots hold NaN, which is never "beyond").
// A cut — the only part with data-dependent indices — reads its four edge vertices from the stripe, writes the kept vertices
// to their new slots FROM THE REGISTERS (no dependent read-modify-write chain, any order), then reloads the cache in one batch.
// The outside run's start comes from the bit mask of the counting pass instead of a backward walk.  Per face the same
// operations on the same operands as clip_poly2 / face_solid_angle_2d: the areas are bit for bit those of the host twin.
template <class P> __device__ bool face_solid_angle_2d_regs(int i, int n, const double (*pts)[3], const double *nsq, double k, P &poly, double *out, int *peak)
{
    constexpr int CAP = P::CAP;
    static_assert(CAP <= 31, "one bit per vertex slot");
    *out = 0;
    const double *p = pts[i];
    if (!(nsq[i] > 0))
        return true;
    bool later_twin = false; // (face_solid_angle)
    for (int j = i + 1; j < n; ++j)
        if (nsq[j] == nsq[i])
            later_twin = later_twin || (pts[j][0] == p[0] && pts[j][1] == p[1] && pts[j][2] == p[2]);
    if (later_twin)
        return true;
    double u[3], v[3];
    const double ax = fabs(p[0]), ay = fabs(p[1]), az = fabs(p[2]);
    double e3[3] = {0, 0, 0};
    if (ax <= ay && ax <= az) e3[0] = 1; else if (ay <= az) e3[1] = 1; else e3[2] = 1;
    ptmc::cross3(p, e3, u);
    const double iu = 1.0 / sqrt(ptmc::dot3(u, u));
    u[0] *= iu; u[1] *= iu; u[2] *= iu;
    ptmc::cross3(p, u, v);
    const double iv = 1.0 / sqrt(ptmc::dot3(v, v));
    v[0] *= iv; v[1] *= iv; v[2] *= iv;
    const double c0[3] = {0.5 * p[0], 0.5 * p[1], 0.5 * p[2]};
    const double R = 40 * k + 1;
    const double QNAN = __builtin_nan("");
    double va[CAP], vb[CAP]; // the polygon: slots [0, m) as in the stripe, the rest NaN (va)
#pragma unroll
    for (int t = 0; t < CAP; ++t) { va[t] = QNAN; vb[t] = 0; }
    va[0] = R; vb[0] = R; va[1] = -R; vb[1] = R; va[2] = -R; vb[2] = -R; va[3] = R; vb[3] = -R;
#pragma unroll
    for (int t = 0; t < 4; ++t) { poly.set(t, 0, va[t]); poly.set(t, 1, vb[t]); }
    int m = 4;
    for (int j = 0; j < n && m >= 3; ++j) // nearest planes first (rows are distance-sorted): the square shrinks at once
        if (j != i && !(nsq[j] == nsq[i] && pts[j][0] == p[0] && pts[j][1] == p[1] && pts[j][2] == p[2])) { // a twin's plane is this plane
            const double la = ptmc::dot3(u, pts[j]), lb = ptmc::dot3(v, pts[j]), off = 0.5 * nsq[j] - ptmc::dot3(c0, pts[j]);
            unsigned outmask = 0;
#pragma unroll
            for (int t = 0; t < CAP; ++t) outmask |= ((va[t] * la + vb[t] * lb) - off > 0) ? (1u << t) : 0u;
            const int nout = __builtin_popcount(outmask);
            if (nout == m) {
                m = 0;
            } else if (nout > 0) { // clip_cut2 (ptm_core.hpp)
                const unsigned inside = ~outmask & ((1u << m) - 1u);
                int s = (outmask & 1u) ? 32 - __builtin_clz(inside) : __builtin_ctz(outmask); // first vertex of the outside run (its predecessor is inside)
                if (s == m) s = 0;
                const int e = s + nout >= m ? s + nout - m : s + nout; // first inside vertex after the run
                const int sp = s == 0 ? m - 1 : s - 1, ep = e == 0 ? m - 1 : e - 1;
                double A[2], B[2];
                {
                    const double psp[2] = {poly.get(sp, 0), poly.get(sp, 1)}, ps[2] = {poly.get(s, 0), poly.get(s, 1)};
                    const double pep[2] = {poly.get(ep, 0), poly.get(ep, 1)}, pe[2] = {poly.get(e, 0), poly.get(e, 1)};
                    {
                        const double d0 = (psp[0] * la + psp[1] * lb) - off, d1 = (ps[0] * la + ps[1] * lb) - off;
                        const double t = d0 / (d0 - d1);
                        for (int c = 0; c < 2; ++c) A[c] = psp[c] + t * (ps[c] - psp[c]);
                    }
                    {
                        const double d0 = (pep[0] * la + pep[1] * lb) - off, d1 = (pe[0] * la + pe[1] * lb) - off;
                        const double t = d0 / (d0 - d1);
                        for (int c = 0; c < 2; ++c) B[c] = pep[c] + t * (pe[c] - pep[c]);
                    }
                }
                const int mm = m - nout + 2;
                if (mm > CAP) {
                    m = -1;
                } else {
                    // kept vertex t goes to slot t + delta: the run inside the array (s < e): [0, s) stay, [e, m) move by s + 2 - e;
                    // the run wraps or ends at m: [e, s) move down by e
                    const bool mid = s < e;
                    const int lo = e, hi = mid ? m : s, delta = mid ? s + 2 - e : -e;
                    if (delta != 0) {
#pragma unroll
                        for (int t = 0; t < CAP; ++t)
                            if (t >= lo && t < hi) { poly.set(t + delta, 0, va[t]); poly.set(t + delta, 1, vb[t]); }
                    }
                    const int at = mid ? s : s - e;
                    poly.set(at, 0, A[0]); poly.set(at, 1, A[1]);
                    poly.set(at + 1, 0, B[0]); poly.set(at + 1, 1, B[1]);
                    m = mm;
#pragma unroll
                    for (int t = 0; t < CAP; ++t) {
                        const double ra = poly.get(t, 0), rb = poly.get(t, 1);
                        va[t] = t < m ? ra : QNAN;
                        vb[t] = rb;
                    }
                }
            }
            if (peak && m > *peak) *peak = m;
        }
    bool inside_cube = m >= 3;
    if (inside_cube) {
#pragma unroll
        for (int c = 0; c < CAP; ++c) { // (a NaN slot fails no test: the comparisons are written so that NaN counts as inside)
            const double a = va[c], b = vb[c];
            const bool outside = fabs(c0[0] + a * u[0] + b * v[0]) > k || fabs(c0[1] + a * u[1] + b * v[1]) > k || fabs(c0[2] + a * u[2] + b * v[2]) > k;
            inside_cube = inside_cube && !outside;
        }
    }
    for (int d = 0; d < 3 && m >= 3 && !inside_cube; ++d) { // the bounding cube  |x_d| <= k
        m = ptmc::clip_poly2(poly, m, u[d], v[d], k - c0[d]);
        if (m < 3) break;
        m = ptmc::clip_poly2(poly, m, -u[d], -v[d], k + c0[d]);
    }
    if (m < 0)
        return false;
    if (m < 3)
        return true;
    double first[3], prev[3], cur[3];
    double nfirst = 0, nprev = 0;
    double sa = 0;
    for (int c = 0; c < m; ++c) { // (face_solid_angle_2d: the fan of spherical triangles)
        const double a = poly.get(c, 0), b = poly.get(c, 1);
        cur[0] = c0[0] + a * u[0] + b * v[0]; cur[1] = c0[1] + a * u[1] + b * v[1]; cur[2] = c0[2] + a * u[2] + b * v[2];
        const double ncur = sqrt(ptmc::dot3(cur, cur));
        if (c == 0) { first[0] = cur[0]; first[1] = cur[1]; first[2] = cur[2]; nfirst = ncur; }
        if (c >= 2) {
            double cr[3];
            ptmc::cross3(prev, cur, cr);
            const double num = ptmc::dot3(first, cr);
            const double den = nfirst * nprev * ncur + ptmc::dot3(first, prev) * ncur + ptmc::dot3(cur, first) * nprev + ptmc::dot3(prev, cur) * nfirst;
            sa += fabs(2 * atan2(num, den));
        }
        prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2]; nprev = ncur;
    }
    *out = (sa < ptmc::SLIVER_SR && ptmc::poly_width2(poly, m) < ptmc::SLIVER_WIDTH) ? 0.0 : sa; // (ptm_core.hpp: a needle where a plane grazes the cell is no face)
    return true;
}

// REDO = second pass over the atoms whose polygons outgrew the first pass's storage (flag set), with room for 28 vertices
// ROUNDS (DIM 2; the name is historic): the polygon cached in registers, face_solid_angle_2d_regs
template <bool TRI, int CAP, bool REDO, int DIM, bool ROUNDS = false>
__global__ __launch_bounds__(ORD_THREADS, (CAP == 8 ? (ROUNDS ? 3 : 4) : (ROUNDS && CAP == 10 ? 3 : 1))) void k_ptm_order_faces(const double *__restrict__ x, const double *__restrict__ y,
                                                                 const double *__restrict__ z, int64_t N, DBox b,
                                                                 const int *__restrict__ verlet, int64_t M, int8_t *__restrict__ orders,
                                                                 int *__restrict__ nbr, unsigned char *__restrict__ redo,
                                                                 int *__restrict__ redo_count)
{
    extern __shared__ unsigned char lds[];
    OrderShared &S = *reinterpret_cast<OrderShared *>(lds);
    PolyStripe<CAP, DIM> poly{reinterpret_cast<double *>(lds + sizeof(OrderShared)) + threadIdx.x};
    const int t = threadIdx.x, slot = t / NROW, f = t - slot * NROW;
    if (REDO && *redo_count == 0) // the usual case: a small grid that leaves at once
        return;
  for (int64_t group = blockIdx.x; group * ORD_APB < N; group += gridDim.x) {
    if (REDO) { // uniform per workgroup: skip the group unless one of its atoms asked for the second pass
        bool any = false;
        for (int a = 0; a < ORD_APB; ++a) {
            const int64_t at = group * ORD_APB + a;
            any = any || (at < N && redo[at] != 0);
        }
        if (!any)
            continue;
        __syncthreads();
    }
    const int64_t atom = group * ORD_APB + slot;
    const bool lane_on = slot < ORD_APB && atom < N;
    const bool work = lane_on && (!REDO || redo[atom] != 0);
    // the row as build_env reads it: stop at the first invalid id, skip the atom itself, keep at most 18
    const int scan = M < NROW ? (int)M : NROW;
    if (lane_on) {
        S.raw[slot][f] = f < scan ? verlet[atom * M + f] : -1;
        if (f == 0) { S.overflow[slot] = 0; S.big[slot] = 0; }
    }
    __syncthreads();
    int pos = -1; // this lane's place in the compacted row
    if (lane_on) {
        bool open = true;
        int before = 0;
        for (int a = 0; a < f; ++a) {
            const int j = S.raw[slot][a];
            open = open && j >= 0 && j < N;
            before += (open && j != atom) ? 1 : 0;
        }
        const int j = S.raw[slot][f];
        open = open && j >= 0 && j < N;
        if (open && j != atom) {
            pos = before;
            double dx = x[j] - x[atom], dy = y[j] - y[atom], dz = z[j] - z[atom];
            fold<TRI>(b, dx, dy, dz);
            S.pts[slot][pos][0] = dx; S.pts[slot][pos][1] = dy; S.pts[slot][pos][2] = dz;
            S.nsq[slot][pos] = dx * dx + dy * dy + dz * dz;
            S.ids[slot][pos] = j;
        }
        if (f == NROW - 1) {
            int c = before + ((open && j != atom) ? 1 : 0);
            S.cnt[slot] = c;
        }
    }
    __syncthreads();
    const int cnt = lane_on ? S.cnt[slot] : 0;
    // face f of the compacted row
    if (work && f < cnt) {
        double maxn = 0;
        for (int i = 0; i < cnt; ++i) maxn = fmax(maxn, S.nsq[slot][i]);
        const double k = 10 * sqrt(maxn);
        double a = 0;
        int peak = 0;
        bool fits;
        if constexpr (DIM == 2 && ROUNDS && CAP <= 15) // (the second pass's 28-vertex polygons stay in the stripe: 112 registers of cache)
            fits = face_solid_angle_2d_regs(f, cnt, S.pts[slot], S.nsq[slot], k, poly, &a, (CAP > 8 && !REDO) ? &peak : nullptr);
        else if constexpr (DIM == 2)
            fits = ptmc::face_solid_angle_2d(f, cnt, S.pts[slot], S.nsq[slot], k, poly, &a, (CAP > 8 && !REDO) ? &peak : nullptr);
        else
            fits = ptmc::face_solid_angle(f, cnt, S.pts[slot], S.nsq[slot], k, poly, &a);
        if (!fits)
            S.overflow[slot] = 1;
        if (CAP > 8 && !REDO && peak > 8) S.big[slot] = 1;
        S.area[slot][f] = a;
    }
    __syncthreads();
    if (work) {
        if (S.overflow[slot]) { // (never in the second pass: 28 vertices hold any face of an 18 + 6 plane cell)
            if (!REDO && f == 0) { redo[atom] = 1; atomicAdd(redo_count, 1); }
        } else {
            if (f < cnt) {
                int rank = 0;
                for (int g = 0; g < cnt; ++g) rank += ptmc::face_before(g, f, S.area[slot], S.nsq[slot]) ? 1 : 0;
                orders[atom * NROW + rank] = (int8_t)f;
                nbr[(int64_t)rank * N + atom] = S.ids[slot][f];
            } else {
                orders[atom * NROW + f] = (int8_t)-1;
                nbr[(int64_t)f * N + atom] = -1;
            }
            if (!REDO && f == 0) redo[atom] = 0;
        }
        if (CAP > 8 && !REDO && f == 0 && (S.big[slot] | S.overflow[slot])) atomicAdd(redo_count + 1, 1);
    }
    if (REDO) __syncthreads(); // the shared arrays are reused by the next group
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// stage 1: convex hulls of the first 7 / 13 / 15 points (ptm_core.hpp convex_hull / hull_init / add_facet)
// ---------------------------------------------------------------------------------------------------------------------
// facet word: the oriented triangle with its smallest index first (IB bits each), then two bits for how far the
// oriented triple was rotated to get there, then one for whether the creation order (a,b,c) was swapped to (b,a,c).  The extra
// bits let a later test rebuild the normal from the very operands the facet was created with.
// Up to 16 points (the single-shell hulls: 7, 13, 15 points) a vertex index takes 4 bits: the facet word fits 16 bits, a
// horizon edge 8, and the stripe of a lane 500 bytes instead of 584 — five workgroups per CU instead of four for a kernel
// that runs one wave per SIMD and waits on LDS most of the time.  The 17-point hull of the diamond clusters keeps 5 bits.
template <int NP> struct HullMem {
    static constexpr int IB = NP <= 16 ? 4 : 5;          // bits of a vertex index in a facet word
    static constexpr uint32_t IM = (1u << IB) - 1u;
    typedef typename std::conditional<(NP <= 16), uint16_t, uint32_t>::type FW;
    typedef typename std::conditional<(NP <= 16), uint8_t, uint16_t>::type AW;
    double *P;   // [NP][3]
    FW *F;       // [MAXF]
    uint32_t *E; // [NP-1]: edge marks of the current insertion, row = smaller endpoint, bit hi-1 seen in a visible facet, bit 16+hi-1 in a hidden one
    AW *A;       // [MAXF]: horizon edges waiting to become facets
    static constexpr size_t BYTES = (size_t)BLK * (NP * 3 * 8 + MAXF * sizeof(FW) + (NP - 1) * 4 + MAXF * sizeof(AW));
    __device__ __forceinline__ void pt(int i, double *v) const
    {
        v[0] = P[(i * 3 + 0) * BLK];
        v[1] = P[(i * 3 + 1) * BLK];
        v[2] = P[(i * 3 + 2) * BLK];
    }
};

struct HullState {
    int num_facets, num_prev;
    uint32_t processed;
    bool ok;
    double bary[3];
};

__device__ __forceinline__ void cross(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// sign class of  n . d  against `tol`, n = N/|N| (negated when neg): +1 above, 0 not above.  |N| <= |N0|+|N1|+|N2|, so
// an unnormalised product beyond 1e-9 of that bound settles it; the rest goes through the reference's expression.
__device__ __forceinline__ bool above_plane(const double *N, const double *d, bool neg, double tol)
{
    const double s0 = N[0] * d[0] + N[1] * d[1] + N[2] * d[2];
    const double s = neg ? -s0 : s0;
    const double S = 1e-9 * (fabs(N[0]) + fabs(N[1]) + fabs(N[2]));
    if (s > S)
        return true;
    if (s < -S)
        return false;
    const double nr = sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]); // plane_normal / plane_dist of ptm_core.hpp
    double n[3] = {N[0] / nr, N[1] / nr, N[2] / nr};
    if (neg) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    return n[0] * d[0] + n[1] * d[1] + n[2] * d[2] > tol;
}

// the reference's normalised expression alone (the tail of above_plane)
__device__ __forceinline__ bool above_plane_exact(const double *N, const double *d, bool neg, double tol)
{
    const double nr = sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
    double n[3] = {N[0] / nr, N[1] / nr, N[2] / nr};
    if (neg) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    return n[0] * d[0] + n[1] * d[1] + n[2] * d[2] > tol;
}

template <int NP> __device__ __forceinline__ void raw_normal(const HullMem<NP> &m, int a, int b, int c, double *N, double *pa)
{
    double pb[3], pc[3];
    m.pt(a, pa); m.pt(b, pb); m.pt(c, pc);
    const double u[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    const double v[3] = {pc[0] - pa[0], pc[1] - pa[1], pc[2] - pa[2]};
    cross(u, v, N);
}

// new facet through (a,b,c), oriented away from `inside`; 0xFFFFFFFF when it duplicates one of the facets [first, nf).  (The
// reference compares with every facet of the hull; a facet through the point being inserted can only equal a facet made
// earlier in the SAME insertion — no older facet contains that point — so the caller passes the first facet of the insertion:
// ~3 instead of ~20 dependent LDS reads per new facet.)
template <int NP> __device__ uint32_t make_facet(const HullMem<NP> &m, int a, int b, int c, const double *inside, int nf, int first = 0)
{
    double N[3], pa[3];
    raw_normal(m, a, b, c, N, pa);
    const double d[3] = {pa[0] - inside[0], pa[1] - inside[1], pa[2] - inside[2]};
    const bool flip = above_plane(N, d, false, 0.0);
    int o0 = flip ? b : a, o1 = flip ? a : b, o2 = c;
    int r = 0;
    if (o1 < o0 && o1 <= o2) r = 1;
    if (o2 < o0 && o2 < o1) r = 2;
    if (r == 1 && o2 < o1) r = 2; // (unreachable for distinct indices; keeps the smallest-first rule explicit)
    const int c0 = r == 0 ? o0 : r == 1 ? o1 : o2, c1 = r == 0 ? o1 : r == 1 ? o2 : o0, c2 = r == 0 ? o2 : r == 1 ? o0 : o1;
    constexpr int IB = HullMem<NP>::IB;
    const uint32_t w = (uint32_t)c0 | ((uint32_t)c1 << IB) | ((uint32_t)c2 << (2 * IB));
    bool dup = false;
    for (int j = first; j < nf; ++j)
        dup = dup || ((uint32_t)m.F[j * BLK] & ((1u << (3 * IB)) - 1u)) == w;
    return dup ? 0xFFFFFFFFu : (w | ((uint32_t)r << (3 * IB)) | ((uint32_t)flip << (3 * IB + 2)));
}

// is point q (coordinates) strictly outside facet word w ?
template <int NP> __device__ __forceinline__ bool facet_sees(const HullMem<NP> &m, uint32_t w, const double *q)
{
    constexpr int IB = HullMem<NP>::IB;
    constexpr uint32_t IM = HullMem<NP>::IM;
    const int c0 = w & IM, c1 = (w >> IB) & IM, c2 = (w >> (2 * IB)) & IM, r = (w >> (3 * IB)) & 3;
    const bool flip = (w >> (3 * IB + 2)) & 1;
    const int o0 = r == 0 ? c0 : r == 1 ? c2 : c1, o1 = r == 0 ? c1 : r == 1 ? c0 : c2, o2 = r == 0 ? c2 : r == 1 ? c1 : c0;
    const int a = flip ? o1 : o0, b = flip ? o0 : o1;
    double N[3], pa[3], pb[3], pc[3], pp[3];
    m.pt(a, pa); m.pt(b, pb); m.pt(o2, pc);
    {
        const double u[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
        const double v[3] = {pc[0] - pa[0], pc[1] - pa[1], pc[2] - pa[2]};
        cross(u, v, N);
    }
    // the facet's first vertex c0 is one of the three just read: picked from registers, not read again (a fourth LDS latency per test)
#pragma unroll
    for (int k = 0; k < 3; ++k) pp[k] = c0 == a ? pa[k] : c0 == b ? pb[k] : pc[k];
    const double d[3] = {pp[0] - q[0], pp[1] - q[1], pp[2] - q[2]};
    return above_plane(N, d, flip, ptmc::HULL_TOL);
}

template <int NP> __device__ int hull_start(const HullMem<NP> &m, int num, HullState &h)
{
    h.processed = 0;
    int mn[3], mx[3];
    for (int j = 0; j < 3; ++j) {
        double dmin = 1.7976931348623157e308, dmax = -1.7976931348623157e308;
        int imin = 0, imax = 0;
        for (int i = 0; i < num; ++i) {
            const double d = m.P[(i * 3 + j) * BLK];
            if (d < dmin) { dmin = d; imin = i; }
            if (d > dmax) { dmax = d; imax = i; }
        }
        if (imin == imax)
            return -1;
        mn[j] = imin; mx[j] = imax;
    }
    int a = 0, b = 0;
    {
        double best = 0.0;
        bool any = false;
        for (int j = 0; j < 3; ++j) {
            double p0[3], p1[3];
            m.pt(mn[j], p0); m.pt(mx[j], p1);
            const double dl[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
            const double d = dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2];
            if (d > best) { best = d; a = mn[j]; b = mx[j]; any = true; }
        }
        if (!any)
            return -1;
    }
    double pa[3], pb[3];
    m.pt(a, pa); m.pt(b, pb);
    int c = -1;
    {
        const double ab[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
        const double nab = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2];
        double md = 0.0;
        for (int i = 0; i < num; ++i) {
            if (i == a || i == b) continue;
            double pi[3];
            m.pt(i, pi);
            const double w[3] = {pa[0] - pi[0], pa[1] - pi[1], pa[2] - pi[2]};
            const double dt = w[0] * ab[0] + w[1] * ab[1] + w[2] * ab[2];
            const double dist = ((w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) * nab - dt * dt) / nab;
            if (dist > md) { md = dist; c = i; }
        }
        if (!(md > ptmc::HULL_TOL))
            return -2;
    }
    int d4 = -1;
    {
        double N[3], q[3];
        raw_normal(m, a, b, c, N, q);
        const double nr = sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
        const double n[3] = {N[0] / nr, N[1] / nr, N[2] / nr};
        double md = 0.0;
        for (int i = 0; i < num; ++i) {
            if (i == a || i == b || i == c) continue;
            double pi[3];
            m.pt(i, pi);
            const double dist = fabs(n[0] * (pa[0] - pi[0]) + n[1] * (pa[1] - pi[1]) + n[2] * (pa[2] - pi[2]));
            if (dist > md) { md = dist; d4 = i; }
        }
        if (!(md > ptmc::HULL_TOL))
            return -3;
    }
    const int iv[4] = {a, b, c, d4};
    h.bary[0] = h.bary[1] = h.bary[2] = 0;
    for (int i = 0; i < 4; ++i) {
        h.processed |= 1u << iv[i];
        double p[3];
        m.pt(iv[i], p);
        h.bary[0] += p[0]; h.bary[1] += p[1]; h.bary[2] += p[2];
    }
    h.bary[0] /= 4; h.bary[1] /= 4; h.bary[2] /= 4;
    typedef typename HullMem<NP>::FW FW;
    m.F[0 * BLK] = (FW)make_facet(m, a, b, c, h.bary, 0);
    m.F[1 * BLK] = (FW)make_facet(m, a, b, d4, h.bary, 0);
    m.F[2 * BLK] = (FW)make_facet(m, a, c, d4, h.bary, 0);
    m.F[3 * BLK] = (FW)make_facet(m, b, c, d4, h.bary, 0);
    return 0;
}

// hull of points [0,num); continues the hull of fewer points when h.ok.  0 ok, 1 the centre is on the hull, <0 failure
template <int NP> __device__ int hull_grow(const HullMem<NP> &m, int num, HullState &h)
{
    constexpr int IB = HullMem<NP>::IB;
    constexpr uint32_t IM = HullMem<NP>::IM;
    typedef typename HullMem<NP>::FW FW;
    typedef typename HullMem<NP>::AW AW;
    int num_prev = h.num_prev;
    h.num_prev = num;
    if (!h.ok) {
        const int r = hull_start(m, num, h);
        if (r != 0)
            return r;
        h.num_facets = 4;
        num_prev = 0;
    }
    for (int i = num_prev; i < num; ++i) {
        if ((h.processed >> i) & 1u)
            continue;
        h.processed |= 1u << i;
        double q[3];
        m.pt(i, q);
        for (int a = 0; a < NP - 1; ++a) m.E[a * BLK] = 0;
        int nadd = 0;
        // The walk over the facets, software-pipelined (round 6).  A facet's turn was a CHAIN of LDS round trips — its word, its three
        // vertices, the three returning marks, the word of the facet that takes a dropped one's slot — on a kernel that runs one wavefront
        // per SIMD.  Now the word of the next slot and of the last facet are read a turn ahead, and the marks of a facet are looked at
        // in the NEXT turn, behind the issue of that turn's vertex reads: per facet one exposed round trip (the vertices) instead of four.
        // The LDS executes a wavefront's operations in order, so every mark returns what it returned in the plain loop; the horizon
        // edges are appended in the same order (a facet's before the next facet's marks are even issued).
        uint32_t pw = 0, pe0 = 0, pe1 = 0, pe2 = 0; // the facet whose marks are in flight: its word, what the three marks returned
        bool pending = false;
        auto settle = [&]() { // the horizon edges of the pending facet
            const int a = pw & IM, b = (pw >> IB) & IM, c = (pw >> (2 * IB)) & IM;
            const int hi0 = max(a, b), hi1 = max(b, c), hi2 = max(c, a);
            const bool u = ((pe0 >> (hi0 - 1)) & (pe0 >> (16 + hi0 - 1)) & 1u) != 0;
            const bool v = ((pe1 >> (hi1 - 1)) & (pe1 >> (16 + hi1 - 1)) & 1u) != 0;
            const bool x = ((pe2 >> (hi2 - 1)) & (pe2 >> (16 + hi2 - 1)) & 1u) != 0;
            if (u && nadd < MAXF) { m.A[nadd * BLK] = (AW)(a | (b << IB)); ++nadd; }
            if (v && nadd < MAXF) { m.A[nadd * BLK] = (AW)(b | (c << IB)); ++nadd; }
            if (x && nadd < MAXF) { m.A[nadd * BLK] = (AW)(c | (a << IB)); ++nadd; }
        };
        int j = 0;
        uint32_t w = h.num_facets > 0 ? (uint32_t)m.F[0] : 0u;                                  // the facet in slot j
        uint32_t wlast = h.num_facets > 0 ? (uint32_t)m.F[(h.num_facets - 1) * BLK] : 0u;      // the facet in the last slot
        while (j < h.num_facets) {
            const uint32_t wnext = m.F[min(j + 1, MAXF - 1) * BLK]; // (used if this facet stays; a read past the end is never used)
            const int a = w & IM, b = (w >> IB) & IM, c = (w >> (2 * IB)) & IM;
            // facet_sees, its reads first
            const int r = (w >> (3 * IB)) & 3;
            const bool flip = (w >> (3 * IB + 2)) & 1;
            const int o0 = r == 0 ? a : r == 1 ? c : b, o1 = r == 0 ? b : r == 1 ? a : c, o2 = r == 0 ? c : r == 1 ? b : a;
            const int va = flip ? o1 : o0, vb = flip ? o0 : o1;
            double pa[3], pb[3], pc[3];
            m.pt(va, pa); m.pt(vb, pb); m.pt(o2, pc);
            if (pending) settle(); // (behind the reads above, in front of their use)
            bool vis;
            {
                double Nn[3], pp[3];
                const double uu[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
                const double vv[3] = {pc[0] - pa[0], pc[1] - pa[1], pc[2] - pa[2]};
                cross(uu, vv, Nn);
                // The quick decision from vertex `va` (already in registers): any vertex of the facet gives the same product up to
                // rounding, 1e-16 |N| |p|, and the band S = 1e-9 |N| is seven orders wider — a product outside the band has the sign
                // the reference's expression (from the facet's FIRST vertex, HULL_TOL = 1e-12) has.  Inside the band: that expression.
                const double da[3] = {pa[0] - q[0], pa[1] - q[1], pa[2] - q[2]};
                const double s0 = Nn[0] * da[0] + Nn[1] * da[1] + Nn[2] * da[2];
                const double sg = flip ? -s0 : s0;
                const double S = 1e-9 * (fabs(Nn[0]) + fabs(Nn[1]) + fabs(Nn[2]));
                if (sg > S) {
                    vis = true;
                } else if (sg < -S) {
                    vis = false;
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k) pp[k] = a == va ? pa[k] : a == vb ? pb[k] : pc[k]; // the facet's first vertex
                    const double d[3] = {pp[0] - q[0], pp[1] - q[1], pp[2] - q[2]};
                    vis = above_plane_exact(Nn, d, flip, ptmc::HULL_TOL);
                }
            }
            const uint32_t side = vis ? 0u : 16u;
            const int lo0 = min(a, b), hi0 = max(a, b), lo1 = min(b, c), hi1 = max(b, c), lo2 = min(c, a), hi2 = max(c, a);
            const uint32_t b0 = 1u << (side + hi0 - 1), b1 = 1u << (side + hi1 - 1), b2 = 1u << (side + hi2 - 1);
            pe0 = __hip_atomic_fetch_or(&m.E[lo0 * BLK], b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) | b0;
            pe1 = __hip_atomic_fetch_or(&m.E[lo1 * BLK], b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) | b1;
            pe2 = __hip_atomic_fetch_or(&m.E[lo2 * BLK], b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) | b2;
            pw = w;
            pending = true;
            if (vis) { // drop the facet: the last one takes its slot and is looked at next
                m.F[j * BLK] = (FW)wlast;
                --h.num_facets;
                w = wlast;
                wlast = m.F[max(h.num_facets - 1, 0) * BLK]; // (slot j itself when it is the last one left: just written)
            } else {
                ++j;
                w = wnext;
            }
        }
        if (pending) settle();
        const int first_new = h.num_facets;
        for (int j = 0; j < nadd; ++j) {
            if (h.num_facets >= MAXF)
                return -4;
            const int e = m.A[j * BLK];
            const uint32_t w = make_facet(m, i, e & IM, (e >> IB) & IM, h.bary, h.num_facets, first_new);
            if (w == 0xFFFFFFFFu)
                return -5;
            m.F[h.num_facets * BLK] = (FW)w;
            ++h.num_facets;
        }
    }
    bool centre = false;
    for (int j = 0; j < h.num_facets; ++j) {
        const uint32_t w = m.F[j * BLK];
        centre = centre || (w & IM) == 0; // the smallest index comes first
    }
    return centre ? 1 : 0;
}

struct HullOut {
    uint16_t *facets[NKIND]; // [MAXF][N], vertex indices - 1
    int8_t *status[NKIND];   // [N]: number of facets of a usable hull, -1 otherwise
};

// gather the atom's ordered neighbourhood (centre first), subtract the barycentre of ALL points, scale by the mean
// distance (ptm_core.hpp normalize_vertices); the first NP points go to m.P.  Returns the number of points.
template <bool TRI, int NP>
__device__ int load_normalized(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z, const DBox &b,
                               const int *__restrict__ nbr, int64_t N, int64_t atom, const HullMem<NP> &m)
{
    const double xi = x[atom], yi = y[atom], zi = z[atom];
    double sum[3] = {0, 0, 0};
    double ex[NROW + 1 - NP][3]; // points beyond the hull's reach only count in the barycentre and the scale
    int num = 1;
    bool open = true;
    // Six neighbours at a time: their ids, then their positions, requested together; the minimum images afterwards.  With id,
    // position and fold in one loop body every gather waited behind the branches of the previous neighbour's fold — eighteen
    // dependent memory latencies per atom in a kernel that runs five waves per CU.
#pragma unroll
    for (int k0 = 0; k0 < NROW; k0 += 6) {
        int jj[6];
        bool op[6];
        double gx[6], gy[6], gz[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) jj[u] = nbr[(int64_t)(k0 + u) * N + atom];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            open = open && jj[u] >= 0;
            op[u] = open;
            const int64_t q = open ? (int64_t)jj[u] : atom; // (a closed slot reads the centre: never used)
            gx[u] = x[q] - xi; gy[u] = y[q] - yi; gz[u] = z[q] - zi;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int k = k0 + u;
            double dx = 0, dy = 0, dz = 0;
            if (op[u]) {
                dx = gx[u]; dy = gy[u]; dz = gz[u];
                fold<TRI>(b, dx, dy, dz);
                sum[0] += dx; sum[1] += dy; sum[2] += dz;
                ++num;
            }
            if (k + 1 < NP) {
                m.P[((k + 1) * 3 + 0) * BLK] = dx; m.P[((k + 1) * 3 + 1) * BLK] = dy; m.P[((k + 1) * 3 + 2) * BLK] = dz;
            } else {
                ex[k + 1 - NP][0] = dx; ex[k + 1 - NP][1] = dy; ex[k + 1 - NP][2] = dz;
            }
        }
    }
    static_assert(NROW % 6 == 0, "batches of six");
    const double s[3] = {sum[0] / num, sum[1] / num, sum[2] / num};
    double scale = 0.0;
    m.P[0 * BLK] = 0.0 - s[0]; m.P[1 * BLK] = 0.0 - s[1]; m.P[2 * BLK] = 0.0 - s[2];
#pragma unroll
    for (int i = 1; i <= NROW; ++i) {
        double v[3];
        if (i < NP) { m.pt(i, v); } else { v[0] = ex[i - NP][0]; v[1] = ex[i - NP][1]; v[2] = ex[i - NP][2]; }
        v[0] -= s[0]; v[1] -= s[1]; v[2] -= s[2];
        if (i < num)
            scale += sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (i < NP) { m.P[(i * 3 + 0) * BLK] = v[0]; m.P[(i * 3 + 1) * BLK] = v[1]; m.P[(i * 3 + 2) * BLK] = v[2]; }
    }
    scale /= num;
#pragma unroll
    for (int i = 0; i < NP; ++i)
        for (int c = 0; c < 3; ++c) m.P[(i * 3 + c) * BLK] = m.P[(i * 3 + c) * BLK] / scale;
    return num;
}

template <bool TRI>
__global__ __launch_bounds__(BLK) void k_ptm_hull(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                  int64_t N, DBox b, const int *__restrict__ nbr, int flags, HullOut out)
{
    constexpr int NP = 15;
    extern __shared__ unsigned char lds[];
    const int64_t atom = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (atom >= N)
        return;
    HullMem<NP> m;
    m.P = reinterpret_cast<double *>(lds) + threadIdx.x;
    typedef HullMem<NP> HM;
    m.F = reinterpret_cast<HM::FW *>(lds + (size_t)BLK * NP * 24) + threadIdx.x;
    m.E = reinterpret_cast<uint32_t *>(lds + (size_t)BLK * (NP * 24 + MAXF * sizeof(HM::FW))) + threadIdx.x;
    m.A = reinterpret_cast<HM::AW *>(lds + (size_t)BLK * (NP * 24 + MAXF * sizeof(HM::FW) + (NP - 1) * 4)) + threadIdx.x;
    const int num = load_normalized<TRI, NP>(x, y, z, b, nbr, N, atom, m);
    HullState h;
    h.ok = false; h.num_prev = 0; h.num_facets = 0; h.processed = 0;
    h.bary[0] = h.bary[1] = h.bary[2] = 0;
    for (int kind = 0; kind < NKIND; ++kind) {
        const int np = kind_points(kind);
        const int want = kind == K_SC ? ptmc::CHECK_SC : kind == K_FCC ? (ptmc::CHECK_FCC | ptmc::CHECK_HCP | ptmc::CHECK_ICO) : ptmc::CHECK_BCC;
        if (!(flags & want))
            continue;
        int8_t st = -1;
        if (num >= np) {
            bool prev_ok = h.ok;
            const bool retry = kind != K_FCC; // the reference repeats a failed continued hull from scratch for SC and BCC only
            int ret = 0;
            for (int attempt = 0; attempt < 2; ++attempt) {
                ret = hull_grow(m, np, h);
                h.ok = ret >= 0;
                if (!(retry && prev_ok && !h.ok))
                    break;
                prev_ok = false;
            }
            if (ret == 0) {
                st = (int8_t)h.num_facets;
                for (int j = 0; j < h.num_facets; ++j) {
                    const uint32_t w = m.F[j * BLK];
                    out.facets[kind][(int64_t)j * N + atom] =
                        (uint16_t)(((w & HM::IM) - 1) | ((((w >> HM::IB) & HM::IM) - 1) << 5) | ((((w >> (2 * HM::IB)) & HM::IM) - 1) << 10)); // (5 bits each on the way out)
                }
            }
        }
        out.status[kind][atom] = st;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// stage 2: Weinberg canonical form of the hull triangulation (ptm_core.hpp canonical_form / weinberg), one kind per launch
// ---------------------------------------------------------------------------------------------------------------------
struct CanonOut {
    uint64_t *hash; // [N]
    int8_t *label;  // [17][N]: canonical labelling, label[0] = 0
    int8_t *ok;     // [N]
};

// (The facets are not kept in LDS — the kernel holds them in registers, two to a word — and the tables have NN rows, not 16: 212 instead
// of 280 bytes per lane for the 14-neighbour kinds, twelve workgroups per CU instead of eight for a stage that runs a serial traversal per lane.)
template <int NN> struct CanonMem {
    static constexpr int NF = 2 * NN - 4, NE = 3 * NN - 6;
    uint64_t *C; // [NN]: row a, nibble b = third vertex of the facet left of a->b
    uint16_t *M; // [NN]: first the "edge defined" bits, then the walked bits of a traversal
    int8_t *B;   // [2 NE]: best code
    static constexpr size_t BYTES = (size_t)BLK * (NN * 8 + NN * 2 + 2 * NE);
    __device__ __forceinline__ int cm(int a, int b) const { return (int)((C[a * BLK] >> (4 * b)) & 15u); }
};

// COL: the first four vertices carry colour 1 (the inner atoms of the diamond cluster): a vertex's code is colour * NN + its
// visiting number, so that inner atoms only map onto inner atoms
template <int NN, bool COL>
__global__ __launch_bounds__(BLK, (NN >= 12 && NN <= 14 ? 3 : 1)) void k_ptm_canon(int64_t N, const uint16_t *__restrict__ facets, const int8_t *__restrict__ status,
                                                   int max_degree, int all_degree, CanonOut out)
{
    using Mem = CanonMem<NN>;
    constexpr int NF = Mem::NF, NE = Mem::NE;
    extern __shared__ unsigned char lds[];
    const int64_t atom = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (atom >= N)
        return;
    Mem m;
    m.C = reinterpret_cast<uint64_t *>(lds) + threadIdx.x;
    m.M = reinterpret_cast<uint16_t *>(lds + (size_t)BLK * NN * 8) + threadIdx.x;
    m.B = reinterpret_cast<int8_t *>(lds + (size_t)BLK * (NN * 8 + NN * 2)) + threadIdx.x;
    bool good = status[atom] == NF;
    // the NF facets once, two to a register (round 6): they are read three times below and once per start edge, and every read was a
    // trip to memory inside a run-time loop — 3 NF dependent latencies per atom before the first traversal step.  A facet by a run-time
    // number (the start edge's) is a chain of selects.
    uint32_t fw[NF / 2];
#pragma unroll
    for (int j = 0; j < NF / 2; ++j) {
        const uint32_t lo = good ? facets[(int64_t)(2 * j) * N + atom] : 0u, hi = good ? facets[(int64_t)(2 * j + 1) * N + atom] : 0u;
        fw[j] = lo | (hi << 16);
    }
    static_assert(NF % 2 == 0, "facets in pairs");
    auto facet_at = [&](int j) { return (int)((fw[j >> 1] >> (16 * (j & 1))) & 0xffffu); }; // compile-time j
    auto facet = [&](int j) { // run-time j
        uint32_t w = fw[0];
#pragma unroll
        for (int q = 1; q < NF / 2; ++q) w = (j >> 1) == q ? fw[q] : w;
        return (int)((w >> (16 * (j & 1))) & 0xffffu);
    };
    uint64_t deg = 0; // nibble v = degree of vertex v
    if (good) {
#pragma unroll
        for (int a = 0; a < NN; ++a) { m.C[a * BLK] = 0; m.M[a * BLK] = 0; }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int w = facet_at(j);
            const int a = w & 31, b = (w >> 5) & 31, c = (w >> 10) & 31;
            deg += (1ull << (4 * a)) + (1ull << (4 * b)) + (1ull << (4 * c));
            // every directed edge may appear once (an oriented closed surface)
            const uint16_t ma = m.M[a * BLK], mb = m.M[b * BLK], mc = m.M[c * BLK];
            if (((ma >> b) | (mb >> c) | (mc >> a)) & 1)
                good = false;
            m.M[a * BLK] = (uint16_t)(ma | (1u << b));
            m.M[b * BLK] = (uint16_t)(m.M[b * BLK] | (1u << c));
            m.M[c * BLK] = (uint16_t)(m.M[c * BLK] | (1u << a));
            m.C[a * BLK] |= (uint64_t)c << (4 * b);
            m.C[b * BLK] |= (uint64_t)a << (4 * c);
            m.C[c * BLK] |= (uint64_t)b << (4 * a);
        }
        int mx = 0;
        bool equal = true, all_ok = true;
        const int d0 = (int)(deg & 15u);
        for (int v = 0; v < NN; ++v) {
            const int d = (int)((deg >> (4 * v)) & 15u);
            mx = d > mx ? d : mx;
            equal = equal && d == d0;
            all_ok = all_ok && (all_degree == 0 || d == all_degree);
        }
        equal = equal && !COL;
        good = good && mx <= max_degree && all_ok;
        // start edges: rotation r of facet j at bit 3 j + r, in the order the reference tries them
        uint64_t s_lo = 0;
        uint32_t s_hi = 0;
        if (good) {
            if (equal) {
                s_lo = 1;
            } else {
                uint32_t bestd = 0;
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const int w = facet_at(j);
                    const uint32_t da = (uint32_t)((deg >> (4 * (w & 31))) & 15u), db = (uint32_t)((deg >> (4 * ((w >> 5) & 31))) & 15u),
                                   dc = (uint32_t)((deg >> (4 * ((w >> 10) & 31))) & 15u);
                    const uint32_t k0 = (da << 16) | (db << 8) | dc, k1 = da | (db << 16) | (dc << 8), k2 = (da << 8) | db | (dc << 16);
                    bestd = max(bestd, max(k0, max(k1, k2)));
                }
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const int w = facet_at(j);
                    const uint32_t da = (uint32_t)((deg >> (4 * (w & 31))) & 15u), db = (uint32_t)((deg >> (4 * ((w >> 5) & 31))) & 15u),
                                   dc = (uint32_t)((deg >> (4 * ((w >> 10) & 31))) & 15u);
                    const uint32_t k0 = (da << 16) | (db << 8) | dc, k1 = da | (db << 16) | (dc << 8), k2 = (da << 8) | db | (dc << 16);
                    uint32_t bits = (k0 == bestd ? 1u : 0u) | (k1 == bestd ? 2u : 0u) | (k2 == bestd ? 4u : 0u);
                    const int pos = 3 * j;
                    if (pos < 64) {
                        s_lo |= (uint64_t)bits << pos;
                        if (pos > 61) s_hi |= bits >> (64 - pos);
                    } else {
                        s_hi |= bits << (pos - 64);
                    }
                }
            }
            for (int i = 0; i < 2 * NE; ++i) m.B[i * BLK] = 127;
        }
        uint64_t best_label = 0;
        // all lanes walk their t-th start edge together
        while (true) {
            const bool has = good && (s_lo != 0 || s_hi != 0);
            if (__ballot(has) == 0)
                break;
            if (!has)
                continue;
            int pos;
            if (s_lo) { pos = __builtin_ctzll(s_lo); s_lo &= s_lo - 1; }
            else { pos = 64 + __builtin_ctz(s_hi); s_hi &= s_hi - 1; }
            const int j = pos / 3, r = pos - 3 * j;
            const int w = facet(j);
            const int v0 = w & 31, v1 = (w >> 5) & 31, v2 = (w >> 10) & 31;
            int a = r == 0 ? v0 : r == 1 ? v1 : v2, bq = r == 0 ? v1 : r == 1 ? v2 : v0;
            // one traversal (colours are all 0 for these kinds: a vertex's code is its visiting number)
            for (int v = 0; v < NN; ++v) m.M[v * BLK] = 0;
            uint64_t index = 0;   // nibble v = visiting number
            uint32_t seen = 1u << a;
            int n = 1;            // index[a] = 0
            bool winning = false, alive = true;
            {
                const int first = COL && a < 4 ? NN : 0, cur = m.B[0];
                if (first > cur) alive = false;
                if (first < cur) { m.B[0] = (int8_t)first; winning = true; }
            }
            for (int it = 1; it < 2 * NE && alive; ++it) {
                const bool newv = !((seen >> bq) & 1u);
                if (newv) {
                    index |= (uint64_t)n << (4 * bq);
                    seen |= 1u << bq;
                    ++n;
                }
                const int val = (int)((index >> (4 * bq)) & 15u) + (COL && bq < 4 ? NN : 0);
                const int cur = m.B[it * BLK];
                const uint16_t mb = m.M[bq * BLK];
                int c = m.cm(a, bq);
                if (!winning && val > cur) { alive = false; break; }
                if (winning || val < cur) { winning = true; m.B[it * BLK] = (int8_t)val; }
                if (!newv) {
                    if (!((mb >> a) & 1u)) {
                        c = a; // old vertex reached on a new path: go back
                    } else {
                        while ((mb >> c) & 1u) c = m.cm(c, bq); // right-most edge not yet walked in that direction
                    }
                }
                m.M[a * BLK] = (uint16_t)(m.M[a * BLK] | (1u << bq));
                a = bq;
                bq = c;
            }
            if (alive && winning)
                best_label = index;
        }
        if (good) {
            uint64_t hsh = 0;
            for (int i = 0; i < 2 * NE; ++i) {
                uint64_t e = (uint64_t)(int64_t)m.B[i * BLK];
                e += i % 8;
                e &= 0xF;
                e <<= (4 * i) % 64;
                hsh ^= e;
            }
            out.hash[atom] = hsh;
            out.label[atom] = 0;
            for (int v = 0; v < NN; ++v)
                out.label[(int64_t)(v + 1) * N + atom] = (int8_t)(((best_label >> (4 * v)) & 15u) + 1);
        }
    }
    out.ok[atom] = good ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// two-shell clusters for the diamond / graphene templates (ptm_core.hpp two_shell_env, ptm_multishell.cpp:41-186) and the hull
// of the 16-neighbour diamond cluster with its inner atoms folded in (match_dcub_dhex, ptm_structure_matcher.cpp:194-311)
// ---------------------------------------------------------------------------------------------------------------------
struct ShellSet {
    int *ids;     // [NPT][N]
    double *pts;  // [NPT * 3][N] relative to the centre atom
    int8_t *ok;   // [N]
};

// entry k of atom a's ordered row counts for the clusters only if it is among the 13 nearest (rank <= 12)
__device__ __forceinline__ bool shell_entry(const int *__restrict__ nbr, const int8_t *__restrict__ orders, int64_t N, int a, int k, int *id)
{
    const int j = nbr[(int64_t)k * N + a];
    const int o = orders[(int64_t)a * NROW + k];
    *id = j;
    return j >= 0 && o >= 0 && o <= ptmc::MAX_MULTISHELL - 1;
}

// centre + INNER nearest (Voronoi order, among the 13 nearest) + OUTER neighbours of each of them that are not part of the
// cluster yet, taken rank by rank across the inner atoms
template <bool TRI, int INNER, int OUTER>
__global__ __launch_bounds__(BLK) void k_ptm_shell(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                   int64_t N, DBox b, const int *__restrict__ nbr, const int8_t *__restrict__ orders, ShellSet out)
{
    constexpr int NPT = 1 + INNER + INNER * OUTER, WANT = INNER * OUTER;
    extern __shared__ unsigned char lds[];
    const int64_t atom = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (atom >= N)
        return;
    // Only the ids of the cluster live in LDS (68 bytes per lane).  The centre and the inner atoms' positions are registers;
    // an outer atom's position goes straight to its place in the output and is read back only in the rare case that the same
    // atom id turns up again (an image of it in a small periodic box).  With ids AND positions in LDS (476 bytes per lane)
    // the kernel — a chain of dependent row and position reads per outer atom — ran 1.25 waves per SIMD and was parked 61 %
    // of the time.
    int *OI = reinterpret_cast<int *>(lds) + threadIdx.x; // [NPT]
    const double xi = x[atom], yi = y[atom], zi = z[atom];
    double rp[INNER + 1][3]; // centre, inner atoms: relative to the centre
    OI[0] = (int)atom;
    rp[0][0] = rp[0][1] = rp[0][2] = 0;
    int m = 1;
    for (int k = 0; k < NROW && m < INNER + 1; ++k) {
        int j;
        if (!shell_entry(nbr, orders, N, (int)atom, k, &j))
            continue;
        double dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
        fold<TRI>(b, dx, dy, dz);
        OI[m * BLK] = j;
#pragma unroll
        for (int i = 1; i <= INNER; ++i)
            if (i == m) { rp[i][0] = dx; rp[i][1] = dy; rp[i][2] = dz; }
        ++m;
    }
    bool good = m == INNER + 1;
    int found = 0;
    if (good) {
        double tol;
        {
            const double d[3] = {rp[0][0] - rp[1][0], rp[0][1] - rp[1][1], rp[0][2] - rp[1][2]};
            tol = 1E-5 * sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            tol = tol > 1E-5 ? tol : 1E-5;
        }
        int inner[INNER], lens[INNER], cursor[INNER], counts[INNER];
        double ix[INNER], iy[INNER], iz[INNER];
        int max_len = 0;
#pragma unroll
        for (int i = 0; i < INNER; ++i) {
            inner[i] = OI[(1 + i) * BLK];
            ix[i] = x[inner[i]]; iy[i] = y[inner[i]]; iz[i] = z[inner[i]];
            int n = 1, dummy;
            for (int k = 0; k < NROW; ++k) n += shell_entry(nbr, orders, N, inner[i], k, &dummy) ? 1 : 0;
            lens[i] = n;
            good = good && n >= INNER + 1;
            max_len = n > max_len ? n : max_len;
            cursor[i] = 0;
            counts[i] = 0;
        }
        unsigned filled = 0; // bit s: outer slot s holds an atom
        if (good)
            for (int j = 1; j < max_len && found < WANT; ++j) {
#pragma unroll
                for (int i = 0; i < INNER; ++i) {
                    if (j >= lens[i] || found >= WANT)
                        continue;
                    int id = -1;
                    while (cursor[i] < NROW && !shell_entry(nbr, orders, N, inner[i], cursor[i], &id)) ++cursor[i]; // entry j of this inner atom
                    ++cursor[i];
                    if (counts[i] >= OUTER)
                        continue;
                    double dx = x[id] - ix[i], dy = y[id] - iy[i], dz = z[id] - iz[i];
                    fold<TRI>(b, dx, dy, dz);
                    const double px = dx + rp[1 + i][0], py = dy + rp[1 + i][1], pz = dz + rp[1 + i][2];
                    bool claimed = false;
#pragma unroll
                    for (int k = 0; k < NPT; ++k) {
                        if (k > INNER && !((filled >> k) & 1u))
                            continue;
                        if (id == OI[k * BLK]) {
                            double q[3];
                            if (k <= INNER) { q[0] = rp[k <= INNER ? k : 0][0]; q[1] = rp[k <= INNER ? k : 0][1]; q[2] = rp[k <= INNER ? k : 0][2]; }
                            else { q[0] = out.pts[(int64_t)(k * 3 + 0) * N + atom]; q[1] = out.pts[(int64_t)(k * 3 + 1) * N + atom]; q[2] = out.pts[(int64_t)(k * 3 + 2) * N + atom]; }
                            const double d[3] = {px - q[0], py - q[1], pz - q[2]};
                            if (sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < tol) claimed = true;
                        }
                    }
                    if (claimed)
                        continue;
                    const int slot = 1 + INNER + OUTER * i + counts[i];
                    OI[slot * BLK] = id;
                    out.pts[(int64_t)(slot * 3 + 0) * N + atom] = px; out.pts[(int64_t)(slot * 3 + 1) * N + atom] = py; out.pts[(int64_t)(slot * 3 + 2) * N + atom] = pz;
                    filled |= 1u << slot;
                    ++counts[i];
                    ++found;
                }
            }
    }
    good = good && found == WANT;
    out.ok[atom] = good ? 1 : 0;
    if (good) {
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            out.ids[(int64_t)k * N + atom] = OI[k * BLK];
            if (k <= INNER) {
                out.pts[(int64_t)(k * 3 + 0) * N + atom] = rp[k <= INNER ? k : 0][0];
                out.pts[(int64_t)(k * 3 + 1) * N + atom] = rp[k <= INNER ? k : 0][1];
                out.pts[(int64_t)(k * 3 + 2) * N + atom] = rp[k <= INNER ? k : 0][2];
            }
        }
    }
}

// hull of the normalised diamond cluster, then the reference's surgery on it: a facet spanned by the three outer atoms of
// one inner atom is replaced by the three facets through that inner atom
__global__ __launch_bounds__(BLK) void k_ptm_hull_shell(int64_t N, ShellSet in, int max_degree, uint16_t *__restrict__ facets,
                                                        int8_t *__restrict__ status)
{
    constexpr int NP = 17;
    extern __shared__ unsigned char lds[];
    const int64_t atom = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (atom >= N)
        return;
    HullMem<NP> m;
    m.P = reinterpret_cast<double *>(lds) + threadIdx.x;
    m.F = reinterpret_cast<uint32_t *>(lds + (size_t)BLK * NP * 24) + threadIdx.x;
    m.E = reinterpret_cast<uint32_t *>(lds + (size_t)BLK * (NP * 24 + MAXF * 4)) + threadIdx.x;
    m.A = reinterpret_cast<uint16_t *>(lds + (size_t)BLK * (NP * 24 + MAXF * 4 + (NP - 1) * 4)) + threadIdx.x;
    int8_t st = -1;
    if (in.ok[atom]) {
        double sum[3] = {0, 0, 0};
        for (int i = 0; i < NP; ++i)
            for (int c = 0; c < 3; ++c) {
                const double v = in.pts[(int64_t)(i * 3 + c) * N + atom];
                m.P[(i * 3 + c) * BLK] = v;
                sum[c] += v;
            }
        const double s3[3] = {sum[0] / NP, sum[1] / NP, sum[2] / NP};
        double scale = 0;
        for (int i = 0; i < NP; ++i) {
            double v[3];
            for (int c = 0; c < 3; ++c) { v[c] = m.P[(i * 3 + c) * BLK] - s3[c]; m.P[(i * 3 + c) * BLK] = v[c]; }
            if (i >= 1) scale += sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        }
        scale /= NP;
        for (int i = 0; i < NP * 3; ++i) m.P[i * BLK] = m.P[i * BLK] / scale;
        HullState h;
        h.ok = false; h.num_prev = 0; h.num_facets = 0; h.processed = 0;
        h.bary[0] = h.bary[1] = h.bary[2] = 0;
        bool good = hull_grow(m, NP, h) == 0;
        int nfac = h.num_facets;
        unsigned inverted = 0;
        if (good) {
            for (int j = 0; j < nfac; ++j) { // vertex indices - 1 from here on
                const uint32_t w = m.F[j * BLK];
                const uint32_t a = (w & 31) - 1, bq = ((w >> 5) & 31) - 1, c = ((w >> 10) & 31) - 1;
                m.F[j * BLK] = a | (bq << 5) | (c << 10);
                int n = 0; // an inner atom may lie on the hull, but never two on one facet
                if (a <= 3) { inverted |= 1u << a; ++n; }
                if (bq <= 3) { inverted |= 1u << bq; ++n; }
                if (c <= 3) { inverted |= 1u << c; ++n; }
                good = good && n <= 1;
            }
        }
        const int num_inverted = __builtin_popcount(inverted);
        good = good && nfac == 20 + 2 * num_inverted;
        if (good) { // (sic) the degree bound looks at the first 20 facets only, ptm_structure_matcher.cpp:231
            uint64_t deg = 0;
            for (int j = 0; j < 20; ++j) {
                const uint32_t w = m.F[j * BLK];
                deg += (1ull << (4 * (w & 31))) + (1ull << (4 * ((w >> 5) & 31))) + (1ull << (4 * ((w >> 10) & 31)));
            }
            for (int v = 0; v < 16; ++v) good = good && (int)((deg >> (4 * v)) & 15u) <= max_degree;
        }
        int num_found = 0;
        if (good) {
            for (int j = 0; j < nfac; ++j) {
                const uint32_t w = m.F[j * BLK];
                const int a = w & 31, bq = (w >> 5) & 31, c = (w >> 10) & 31;
                if (a <= 3 || bq <= 3 || c <= 3)
                    continue;
                const int i0 = (a - 4) / 3, i1 = (bq - 4) / 3, i2 = (c - 4) / 3;
                if (i0 == i1 && i0 == i2) {
                    if (num_found + num_inverted >= 4) { good = false; break; }
                    m.A[num_found * BLK] = (uint16_t)w;
                    ++num_found;
                    m.F[j * BLK] = m.F[(nfac - 1) * BLK];
                    --nfac;
                    --j;
                }
            }
        }
        good = good && num_found + num_inverted == 4;
        if (good) {
            for (int t = 0; t < num_found; ++t) {
                const uint32_t w = m.A[t * BLK];
                const uint32_t a = w & 31, bq = (w >> 5) & 31, c = (w >> 10) & 31, i0 = (a - 4) / 3;
                m.F[nfac * BLK] = i0 | (bq << 5) | (c << 10); ++nfac;
                m.F[nfac * BLK] = a | (i0 << 5) | (c << 10); ++nfac;
                m.F[nfac * BLK] = a | (bq << 5) | (i0 << 10); ++nfac;
            }
            // the canonical-form stage repeats the degree bound over all facets (its max_degree argument)
            st = (int8_t)nfac;
            for (int j = 0; j < nfac; ++j) facets[(int64_t)j * N + atom] = (uint16_t)m.F[j * BLK];
        }
    }
    status[atom] = st;
}

// ---------------------------------------------------------------------------------------------------------------------
// stage 3: template look-up, QCP superposition per automorphism, alloy ordering, fundamental-zone remap, outputs
// ---------------------------------------------------------------------------------------------------------------------
enum { K_DC = 3, K_GR = 4, NCANON = 4 };

struct MatchIn {
    const uint64_t *hash[NCANON];
    const int8_t *label[NCANON];
    const int8_t *ok[NCANON];
    ShellSet dc, gr;
};

// The single-shell instance (15 points) keeps a point's identity as its SLOT in the atom's ordered row (0 the atom itself,
// k + 1 its k-th neighbour, -1 none: one byte, resolved through `nbr` where an atom id is wanted) instead of the id: 392
// instead of 448 bytes per lane, six workgroups per CU instead of five for a stage that runs one wave per SIMD and is parked
// on LDS and memory half of the time.  The two-shell instance (17 points of a cluster) keeps ids.
template <int NPM> struct MatchMem {
    static constexpr int NP = NPM;
    static constexpr bool SLOTS = NPM <= 15;
    typedef typename std::conditional<SLOTS, int8_t, int>::type IdT;
    static constexpr int NI = SLOTS ? 16 : 17, NV = SLOTS ? 16 : 20;
    double *P;  // [NP][3] raw separations (centre first)
    IdT *I;     // [NI] the points' slots / atom ids in the same order, later those of the matched atoms in template order
    int8_t *V;  // [NV] inverse canonical labelling of the kind at hand
    static constexpr size_t BYTES = (size_t)BLK * (NP * 24 + NI * sizeof(IdT) + NV);
};

// what the match stage reads besides ptmc::Tables, made once on the host (ptm_compose_match_tables)
struct MatchTables {
    static constexpr int ROW = 32;                 // an automorphism's row: MAX_PTS bytes used, 32-byte aligned for wide loads
    int8_t autc[ptmc::MAX_AUTS][ROW];              // autc[aut][i] = canon_of_its_graph[aut^-1[i]] (see ptm_compose_match_tables)
    uint64_t sorted_hash[ptmc::MAX_GRAPHS];        // per type, its range [graph_begin, graph_begin + num_graphs) sorted by (hash, graph)
    int16_t sorted_graph[ptmc::MAX_GRAPHS];        // ... the graph of that entry
};

struct GraphMap { // template point i -> cluster point V[autc[i]]
    const int8_t *V;
    const int8_t *__restrict__ autc; // one row of MatchTables::autc: 32-byte aligned
    __device__ __forceinline__ int operator()(int i) const { return V[autc[i] * BLK]; }
    // all NPX points at once: the row as two loads (one memory latency instead of one per point), then the NPX look-ups in V
    // together (entries past the template's last point read row byte 0: a valid slot, never used)
    template <int NPX> __device__ __forceinline__ void gather(int np, int *kk) const
    {
        static_assert(NPX <= 20, "five words of the row");
        const uint4 lo = *reinterpret_cast<const uint4 *>(autc);
        const uint32_t hi = *reinterpret_cast<const uint32_t *>(autc + 16);
        const uint32_t w[5] = {lo.x, lo.y, lo.z, lo.w, hi};
#pragma unroll
        for (int i = 0; i < NPX; ++i) kk[i] = V[(int)((w[i >> 2] >> (8 * (i & 3))) & 0xffu) * BLK];
    }
};
struct GrapheneMap { // the 2^3 assignments of each inner atom's two outer atoms: bit 2 swaps points 4,5, bit 1 6,7, bit 0 8,9
    int bits;
    __device__ __forceinline__ int operator()(int i) const { return i < 4 ? i : (((bits >> (2 - ((i - 4) >> 1))) & 1) ? (i ^ 1) : i); }
    template <int NPX> __device__ __forceinline__ void gather(int np, int *kk) const
    {
#pragma unroll
        for (int i = 0; i < NPX; ++i) kk[i] = i < 10 ? (*this)(i) : 0;
    }
};

// rmsd of template `s` onto the points  P[map(i)] - bary  (ptm_core.hpp calc_rmsd)
template <class Mem, class Map>
__device__ double stage_rmsd(const Mem &m, const ptmc::TypeInfo &s, int np, const Map &map, const double *bary, double G1, double G2, double E0,
                             double *q, double *p_scale)
{
    // The correspondence first, all points at once, then the points six at a time (their 18 LDS reads in flight together), the
    // sums in the order i = 0, 1, ... of the plain loop (same bits).  One point per trip of a run-time loop was three DEPENDENT
    // latencies per point — the automorphism's byte from memory, the look-up in V, the point — 13 x 3 per automorphism, 24
    // automorphisms per fcc graph.
    constexpr int NPX = Mem::NP;
    int kk[NPX];
    map.template gather<NPX>(np, kk);
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NPX; c += 6) {
        double px[6], py[6], pz[6];
#pragma unroll
        for (int u = 0; u < 6; ++u)
            if (c + u < NPX) {
                const int k = kk[c + u];
                px[u] = m.P[(k * 3 + 0) * BLK]; py[u] = m.P[(k * 3 + 1) * BLK]; pz[u] = m.P[(k * 3 + 2) * BLK];
            }
#pragma unroll
        for (int u = 0; u < 6; ++u)
            if (c + u < NPX && c + u < np) {
                const int i = c + u;
                const double x1 = s.points[i][0], y1 = s.points[i][1], z1 = s.points[i][2];
                const double x2 = px[u] - bary[0], y2 = py[u] - bary[1], z2 = pz[u] - bary[2];
                A[0] += x1 * x2; A[1] += x1 * y2; A[2] += x1 * z2;
                A[3] += y1 * x2; A[4] += y1 * y2; A[5] += y1 * z2;
                A[6] += z1 * x2; A[7] += z1 * y2; A[8] += z1 * z2;
            }
    }
    double nrmsdsq, rot[9];
    ptmc::qcp_quaternion(A, E0, &nrmsdsq, q);
    ptmc::quat_to_matrix(q, rot);
    // k0 = sum_i (R s_i) . p_i = sum_{j,c} R[j][c] A[c][j]  (A[c][j] = sum_i s_i[c] p_i[j] is the matrix just accumulated): the
    // reference's second pass over the points (ptm_polar.cpp calc_rmsd) is nine multiplications here — the same number up to the
    // order of summation
    double k0 = 0;
    for (int j = 0; j < 3; ++j)
        for (int c = 0; c < 3; ++c) k0 += rot[j * 3 + c] * A[c * 3 + j];
    const double scale = k0 / G2;
    *p_scale = scale;
    return sqrt(fabs(G1 - scale * k0) / np);
}

struct Best {
    int type, aut, kind;
    double rmsd, scale, q[4];
};

template <class Mem> __device__ __forceinline__ void barycentre(const Mem &m, int np, double *bary, double *G2)
{
    bary[0] = bary[1] = bary[2] = 0;
    for (int i = 0; i < np; ++i) { bary[0] += m.P[(i * 3 + 0) * BLK]; bary[1] += m.P[(i * 3 + 1) * BLK]; bary[2] += m.P[(i * 3 + 2) * BLK]; }
    bary[0] /= np; bary[1] /= np; bary[2] /= np;
    double g = 0;
    for (int i = 0; i < np; ++i) {
        const double v[3] = {m.P[(i * 3 + 0) * BLK] - bary[0], m.P[(i * 3 + 1) * BLK] - bary[1], m.P[(i * 3 + 2) * BLK] - bary[2]};
        g += v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    }
    *G2 = g;
}

__device__ __forceinline__ double template_norm(const ptmc::TypeInfo &s, int np)
{
    double G1 = 0;
    for (int i = 0; i < np; ++i) G1 += s.points[i][0] * s.points[i][0] + s.points[i][1] * s.points[i][1] + s.points[i][2] * s.points[i][2];
    return G1;
}

// every automorphism of every table graph of `type` with the atom's hash, in the order of the graphs; the lanes of a wavefront
// walk their t-th candidate together (ptm_core.hpp check_graphs).  The graphs with the atom's hash are found in the type's
// (hash, graph)-sorted list by bisection: the bcc template has 218 graphs, and a walk over them — one dependent memory latency
// each, for every atom whose 15-point hull is a valid triangulation, i.e. nearly every atom of any crystal — was a third of the
// kernel.  Entries of equal hash are in ascending graph order: candidates come in the order of the plain walk.
template <class Mem>
__device__ void try_graphs(const Mem &m, const Tables &T, const MatchTables &MT, int type, int kind, int np, bool live, uint64_t hash,
                           const double *bary, double G2, Best &best)
{
    const ptmc::TypeInfo &s = T.types[type];
    const double G1 = template_norm(s, np);
    const double E0 = (G1 + G2) / 2;
    const int e_end = s.graph_begin + s.num_graphs;
    int e = e_end, j = 0, g = -1, naut = 0, abeg = 0; // entry of the sorted list, automorphism, its graph (-1: none left)
    auto enter = [&]() {
        g = -1;
        if (e < e_end && MT.sorted_hash[e] == hash) {
            g = MT.sorted_graph[e];
            naut = T.graphs[g].num_aut;
            abeg = T.graphs[g].aut_begin;
            j = 0;
        }
    };
    if (live) {
        int lo = s.graph_begin, hi = e_end;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (MT.sorted_hash[mid] < hash) lo = mid + 1; else hi = mid;
        }
        e = lo;
        enter();
    }
    while (true) {
        if (live)
            while (g >= 0 && j >= naut) { ++e; enter(); }
        const bool has = live && g >= 0;
        if (__ballot(has) == 0)
            break;
        if (has) {
            const int aut = abeg + j;
            double q[4], scale = 0;
            const GraphMap map{m.V, MT.autc[aut]};
            const double rmsd = stage_rmsd(m, s, np, map, bary, G1, G2, E0, q, &scale);
            if (rmsd < best.rmsd) {
                best.rmsd = rmsd; best.scale = scale; best.type = type; best.aut = aut; best.kind = kind;
                best.q[0] = q[0]; best.q[1] = q[1]; best.q[2] = q[2]; best.q[3] = q[3];
            }
            ++j;
        }
    }
}

// the atom's own ordered neighbourhood into P / I: separations folded exactly as the ordering pass folded them
template <bool TRI, class Mem>
__device__ __forceinline__ void load_neighbourhood(const Mem &m, const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ z, const DBox &b, const int *__restrict__ nbr, int64_t N, int64_t atom)
{
    const double xi = x[atom], yi = y[atom], zi = z[atom];
    m.P[0 * BLK] = 0; m.P[1 * BLK] = 0; m.P[2 * BLK] = 0;
    typedef typename Mem::IdT IdT;
    m.I[0 * BLK] = Mem::SLOTS ? (IdT)0 : (IdT)atom;
    bool open = true;
    // 15 points serve the largest single-shell template; seven neighbours at a time, ids then positions requested together, the
    // minimum images afterwards (see load_normalized)
#pragma unroll
    for (int k0 = 0; k0 < 14; k0 += 7) {
        int jj[7];
        bool op[7];
        double gx[7], gy[7], gz[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) jj[u] = nbr[(int64_t)(k0 + u) * N + atom];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            open = open && jj[u] >= 0;
            op[u] = open;
            const int64_t q = open ? (int64_t)jj[u] : atom;
            gx[u] = x[q] - xi; gy[u] = y[q] - yi; gz[u] = z[q] - zi;
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int k = k0 + u;
            double dx = 0, dy = 0, dz = 0;
            if (op[u]) {
                dx = gx[u]; dy = gy[u]; dz = gz[u];
                fold<TRI>(b, dx, dy, dz);
            }
            m.P[((k + 1) * 3 + 0) * BLK] = dx; m.P[((k + 1) * 3 + 1) * BLK] = dy; m.P[((k + 1) * 3 + 2) * BLK] = dz;
            m.I[(k + 1) * BLK] = op[u] ? (Mem::SLOTS ? (IdT)(k + 1) : (IdT)jj[u]) : (IdT)-1;
        }
    }
}

template <class Mem> __device__ __forceinline__ void load_cluster(const Mem &m, const ShellSet &set, int npt, int64_t N, int64_t atom)
{
    // six points per trip, their 24 loads together (one point per trip of a run-time loop: 17 x 4 dependent latencies)
    for (int k0 = 0; k0 < npt; k0 += 6) {
        int id[6];
        double p[6][3];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int k = min(k0 + u, npt - 1);
            id[u] = set.ids[(int64_t)k * N + atom];
#pragma unroll
            for (int c = 0; c < 3; ++c) p[u][c] = set.pts[(int64_t)(k * 3 + c) * N + atom];
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int k = k0 + u;
            if (k < npt) {
                m.I[k * BLK] = id[u];
#pragma unroll
                for (int c = 0; c < 3; ++c) m.P[(k * 3 + c) * BLK] = p[u][c];
            }
        }
    }
}

template <bool TRI, bool SHELL>
__global__ __launch_bounds__(BLK) void k_ptm_match(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                   int64_t N, DBox b, const int *__restrict__ nbr, const int *__restrict__ types,
                                                   const Tables *__restrict__ tables, const MatchTables *__restrict__ mtables, int flags,
                                                   MatchIn in, double rmsd_threshold, double *__restrict__ output, int ncol,
                                                   int *__restrict__ ptm_indices, int nind)
{
    using Mem = MatchMem<SHELL ? 17 : 15>;
    constexpr int NP = Mem::NP;
    extern __shared__ unsigned char lds[];
    const int64_t atom = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (atom >= N)
        return;
    Mem m;
    m.P = reinterpret_cast<double *>(lds) + threadIdx.x;
    m.I = reinterpret_cast<typename Mem::IdT *>(lds + (size_t)BLK * NP * 24) + threadIdx.x;
    m.V = reinterpret_cast<int8_t *>(lds + (size_t)BLK * (NP * 24 + Mem::NI * sizeof(typename Mem::IdT))) + threadIdx.x;
    // a stored identity -> atom id (MatchMem)
    auto id_of = [&](int v) { return !Mem::SLOTS ? v : v < 0 ? -1 : v == 0 ? (int)atom : nbr[(int64_t)(v - 1) * N + atom]; };
    const Tables &T = *tables;
    const MatchTables &MT = *mtables;
    load_neighbourhood<TRI>(m, x, y, z, b, nbr, N, atom);
    Best best;
    best.type = ptmc::T_NONE; best.aut = -1; best.kind = -1;
    best.rmsd = INFINITY; best.scale = 0;
    best.q[0] = best.q[1] = best.q[2] = best.q[3] = 0;
    // the canonical forms' flags of all kinds are requested together, and a kind's labelling as one batch of loads: one flag,
    // then one label per trip of a run-time loop, was a chain of up to sixteen dependent memory latencies per kind
    bool live_of[NKIND];
#pragma unroll
    for (int kind = 0; kind < NKIND; ++kind) live_of[kind] = in.ok[kind] != nullptr && in.ok[kind][atom] != 0;
    for (int kind = 0; kind < NKIND; ++kind) {
        const int np = kind_points(kind);
        bool live = live_of[0];
#pragma unroll
        for (int k = 1; k < NKIND; ++k) live = kind == k ? live_of[k] : live;
        if (__ballot(live) == 0)
            continue;
        double bary[3] = {0, 0, 0}, G2 = 0;
        uint64_t hash = 0;
        if (live) {
            hash = in.hash[kind][atom];
            int lab[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) lab[i] = in.label[kind][(int64_t)min(i, np - 1) * N + atom];
#pragma unroll
            for (int i = 0; i < 15; ++i)
                if (i < np) m.V[lab[i] * BLK] = (int8_t)i;
            barycentre(m, np, bary, &G2);
        }
        const int ntypes = kind == K_FCC ? 3 : 1;
        for (int tk = 0; tk < ntypes; ++tk) {
            const int type = kind == K_SC ? ptmc::T_SC : kind == K_BCC ? ptmc::T_BCC : tk == 0 ? ptmc::T_FCC : tk == 1 ? ptmc::T_HCP : ptmc::T_ICO;
            const int bit = type == ptmc::T_SC ? ptmc::CHECK_SC : type == ptmc::T_BCC ? ptmc::CHECK_BCC : type == ptmc::T_FCC ? ptmc::CHECK_FCC
                          : type == ptmc::T_HCP ? ptmc::CHECK_HCP : ptmc::CHECK_ICO;
            if (flags & bit)
                try_graphs(m, T, MT, type, kind, np, live, hash, bary, G2, best);
        }
    }
    if constexpr (SHELL) {
        if (flags & (ptmc::CHECK_DCUB | ptmc::CHECK_DHEX)) { // the 17-point cluster, inner atoms coloured
            const bool live = in.ok[K_DC] != nullptr && in.ok[K_DC][atom] != 0;
            if (__ballot(live) != 0) {
                double bary[3] = {0, 0, 0}, G2 = 0;
                uint64_t hash = 0;
                if (live) {
                    load_cluster(m, in.dc, 17, N, atom);
                    hash = in.hash[K_DC][atom];
                    for (int i = 0; i < 17; ++i) m.V[in.label[K_DC][(int64_t)i * N + atom] * BLK] = (int8_t)i;
                    barycentre(m, 17, bary, &G2);
                }
                if (flags & ptmc::CHECK_DCUB) try_graphs(m, T, MT, ptmc::T_DCUB, K_DC, 17, live, hash, bary, G2, best);
                if (flags & ptmc::CHECK_DHEX) try_graphs(m, T, MT, ptmc::T_DHEX, K_DC, 17, live, hash, bary, G2, best);
            }
        }
        if (flags & ptmc::CHECK_GRAPHENE) { // 10 points, no graph: the eight assignments directly (ptm_core.hpp match_graphene)
            const bool live = in.gr.ok[atom] != 0;
            if (__ballot(live) != 0 && live) {
                load_cluster(m, in.gr, 10, N, atom);
                double bary[3], G2;
                barycentre(m, 10, bary, &G2);
                const ptmc::TypeInfo &s = T.types[ptmc::T_GRAPHENE];
                const double G1 = template_norm(s, 10), E0 = (G1 + G2) / 2;
                for (int t = 0; t < 8; ++t) {
                    const GrapheneMap map{7 - t};
                    double q[4], scale = 0;
                    const double rmsd = stage_rmsd(m, s, 10, map, bary, G1, G2, E0, q, &scale);
                    if (rmsd < best.rmsd) {
                        best.rmsd = rmsd; best.scale = scale; best.type = ptmc::T_GRAPHENE; best.aut = 7 - t; best.kind = K_GR;
                        best.q[0] = q[0]; best.q[1] = q[1]; best.q[2] = q[2]; best.q[3] = q[3];
                    }
                }
            }
        }
    }
    // ---- outputs (ptm_core.hpp index_atom, second half)
    int type = 0, ordering = 0, num_out = 0;
    double o_rmsd = 0, o_inter = 0, o_q[4] = {0, 0, 0, 0};
    if (best.type != ptmc::T_NONE) {
        const ptmc::TypeInfo &s = T.types[best.type];
        const int np = s.num_nbrs + 1;
        // the atoms of the winning cluster, and the map from template points to them
        if constexpr (SHELL) {
            if (best.kind == K_DC) load_cluster(m, in.dc, 17, N, atom);
            else if (best.kind == K_GR) load_cluster(m, in.gr, 10, N, atom);
            else load_neighbourhood<TRI>(m, x, y, z, b, nbr, N, atom);
        }
        int8_t pick[ptmc::MAX_PTS]; // pick[i] = cluster point matched to template point i
        if (best.kind == K_GR) {
            const GrapheneMap map{best.aut};
#pragma unroll
            for (int i = 0; i < ptmc::MAX_PTS; ++i) pick[i] = (int8_t)(i < 10 ? map(i) : 0);
        } else {
            const int8_t *ac = MT.autc[best.aut];
            {
                const auto *lp = in.label[best.kind];
                int lab[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) lab[i] = lp[(int64_t)min(i, np - 1) * N + atom];
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    if (i < np) m.V[lab[i] * BLK] = (int8_t)i;
            }
#pragma unroll
            for (int i = 0; i < ptmc::MAX_PTS; ++i) pick[i] = (int8_t)(i < np ? m.V[ac[i] * BLK] : 0);
        }
        // alloy ordering (ptm_core.hpp alloy_type); without a type column every atom is the same species
        ordering = ptmc::ALLOY_PURE;
        if (types) {
            const int n0 = types[id_of(m.I[0])];
            bool pure = true, none = n0 == -1, binary = true;
            int other = -1;
            for (int i = 1; i < np; ++i) {
                const int ni = types[id_of(m.I[i * BLK])];
                none = none || ni == -1;
                if (ni != n0) {
                    pure = false;
                    if (other == -1) other = ni;
                    else if (ni != other) binary = false;
                }
            }
            if (none) ordering = ptmc::ALLOY_NONE;
            else if (pure) ordering = ptmc::ALLOY_PURE;
            else if (!binary) ordering = ptmc::ALLOY_NONE;
            else {
                uint32_t bin = 0; // bit i: template point i holds the other species
#pragma unroll
                for (int i = 0; i < ptmc::MAX_PTS; ++i)
                    if (i < np && types[id_of(m.I[pick[i] * BLK])] != n0) bin |= 1u << i;
                uint32_t lowest = 0xFFFFFFFFu;
                for (int r = 0; r < s.num_maps; ++r) {
                    const int8_t *mp = T.maps[s.map_begin + r];
                    uint32_t code = 0;
                    for (int i = 0; i < np; ++i) code |= ((bin >> i) & 1u) << mp[i];
                    lowest = code < lowest ? code : lowest;
                }
                ordering = ptmc::ALLOY_NONE;
                if (best.type == ptmc::T_FCC) {
                    if (lowest == 0x00000db6u) ordering = ptmc::ALLOY_L10;
                    if (lowest == 0x00000492u) ordering = ptmc::ALLOY_L12_CU;
                    if (lowest == 0x00001ffeu) ordering = ptmc::ALLOY_L12_AU;
                }
                // shell structure: the inner sites all differ from the centre, the outer ones all agree
                const int inner = best.type == ptmc::T_BCC ? 8 : (best.type == ptmc::T_DCUB || best.type == ptmc::T_DHEX) ? 4
                                : best.type == ptmc::T_GRAPHENE ? 3 : 0;
                if (ordering == ptmc::ALLOY_NONE && inner) {
                    bool shell = true;
                    for (int i = 1; i < inner + 1; ++i) shell = shell && ((bin >> i) & 1u) != (bin & 1u);
                    for (int i = inner + 1; i < np; ++i) shell = shell && ((bin >> i) & 1u) == (bin & 1u);
                    if (shell) ordering = best.type == ptmc::T_BCC ? ptmc::ALLOY_B2 : best.type == ptmc::T_GRAPHENE ? ptmc::ALLOY_BN : ptmc::ALLOY_SIC;
                }
            }
        }
        // fundamental-zone remap (ptm_core.hpp remap_template)
        double mx = 0.0;
        int bi = -1;
        for (int i = 0; i < s.num_conv; ++i) {
            const double *gq = T.gens[s.gen_begin + i];
            const double t = fabs(best.q[0] * gq[0] - best.q[1] * gq[1] - best.q[2] * gq[2] - best.q[3] * gq[3]);
            if (t > mx) { mx = t; bi = i; }
        }
        // matched atoms in template order: ids[perm[i]] = I[pick[i]]
        int ids_local[ptmc::MAX_PTS];
#pragma unroll
        for (int i = 0; i < ptmc::MAX_PTS; ++i) ids_local[i] = i < np ? m.I[pick[i] * BLK] : -1;
        if (bi >= 0) {
            double f[4];
            ptmc::quat_mul(best.q, T.gens[s.gen_begin + bi], f);
            if (f[0] < 0) { f[0] = -f[0]; f[1] = -f[1]; f[2] = -f[2]; f[3] = -f[3]; }
            best.q[0] = f[0]; best.q[1] = f[1]; best.q[2] = f[2]; best.q[3] = f[3];
            const int8_t *perm = T.maps[s.conv_begin + bi];
#pragma unroll
            for (int i = 0; i < ptmc::MAX_PTS; ++i)
                if (i < np) m.I[perm[i] * BLK] = (typename Mem::IdT)ids_local[i];
        } else {
#pragma unroll
            for (int i = 0; i < ptmc::MAX_PTS; ++i)
                if (i < np) m.I[i * BLK] = (typename Mem::IdT)ids_local[i];
        }
        type = best.type;
        o_rmsd = best.rmsd;
        o_inter = ptmc::interatomic_distance(best.type, best.scale);
        o_q[0] = best.q[0]; o_q[1] = best.q[1]; o_q[2] = best.q[2]; o_q[3] = best.q[3];
        num_out = np;
    }
    if (o_rmsd > rmsd_threshold || type == ptmc::T_NONE) { // src/polyhedral_template_matching.cpp:287-291
        type = 0;
        ordering = 0;
    }
    double *o = output + atom * ncol;
    const double vals[8] = {(double)type, (double)ordering, o_rmsd, o_inter, o_q[0], o_q[1], o_q[2], o_q[3]};
    for (int k = 0; k < ncol; ++k)
        o[k] = k < 8 ? vals[k] : 0.0;
    int *pi = ptm_indices + atom * nind;
    for (int k = 0; k < nind; ++k)
        pi[k] = k < num_out ? id_of(m.I[k * BLK]) : -1;
}

} // namespace ptms

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
// autc[aut][i] = canon_of_its_graph[aut^-1[i]]: with it, template point i maps to V[autc[i]] (V = inverse of the atom's
// canonical labelling) — ptm_core.hpp check_graphs builds the same mapping by scattering through `aut`.  And every type's graphs
// sorted by (hash, graph) for the bisection of try_graphs.
size_t ptm_match_tables_bytes() { return sizeof(ptms::MatchTables); }
void ptm_compose_match_tables(const ptmc::Tables &T, void *out)
{
    ptms::MatchTables &M = *static_cast<ptms::MatchTables *>(out);
    std::memset(&M, 0, sizeof(M));
    for (int g = 0; g < T.num_graphs; ++g) {
        const ptmc::Graph &gr = T.graphs[g];
        int np = 0;
        for (int t = 1; t < 9; ++t)
            if (g >= T.types[t].graph_begin && g < T.types[t].graph_begin + T.types[t].num_graphs) np = T.types[t].num_nbrs + 1;
        for (int j = 0; j < gr.num_aut; ++j) {
            const int8_t *aut = T.auts[gr.aut_begin + j];
            int8_t *dst = M.autc[gr.aut_begin + j];
            for (int k = 0; k < np; ++k) dst[aut[k]] = gr.canon[k];
        }
    }
    for (int g = 0; g < ptmc::MAX_GRAPHS; ++g) { M.sorted_hash[g] = ~0ull; M.sorted_graph[g] = (int16_t)g; }
    for (int t = 1; t < 9; ++t) {
        const int g0 = T.types[t].graph_begin, n = T.types[t].num_graphs;
        std::vector<int> idx(n);
        for (int k = 0; k < n; ++k) idx[k] = g0 + k;
        std::sort(idx.begin(), idx.end(), [&](int a, int b) { return T.graphs[a].hash != T.graphs[b].hash ? T.graphs[a].hash < T.graphs[b].hash : a < b; });
        for (int k = 0; k < n; ++k) { M.sorted_hash[g0 + k] = T.graphs[idx[k]].hash; M.sorted_graph[g0 + k] = (int16_t)idx[k]; }
    }
}

static int g_order_cap = 10; // polygon vertices in the first pass (10: faces of up to ten corners — all of a crystal's — stay in it); larger faces take the second pass
static int g_order_dim = 2;  // 2: polygons in the coordinates of their own plane; 3: in space (the form of rounds 1-2, kept for A/B)
static bool g_order_auto = true;
static bool g_order_rounds = true; // the polygon cached in registers (face_solid_angle_2d_regs); false: in the LDS stripe only (A/B, the areas are the same bits)
// Automatic choice of the first pass: eight-vertex polygons run at four waves per SIMD (128 VGPRs, 22 KB of LDS) and are 5 %
// faster on crystals, whose faces stay small; a gas or a glass sends half its atoms to the second pass with them (2.2x
// slower).  Every call counts the atoms that had a face of more than eight vertices; the count of the previous call with the
// same number of atoms — read from pinned memory, never waited for, like the occupancy word of the neighbor build — picks
// eight when fewer than 2 % of the atoms did, else ten (and ten for a first call).
static int *g_order_stat = nullptr; // pinned: [0] atoms with a face of > 8 vertices in the last finished call
static int64_t g_order_stat_n = -1; // ... which had this many atoms
// test / measurement hook (mdh_debug_set_ptm_order_cap): 0 -> automatic; |cap| -> 5, 8, 10 or 15 vertices; a NEGATIVE value selects the 3-D polygons
void ptm_debug_order_cap(int cap)
{
    g_order_rounds = !(cap >= 100); // 100 + cap: the plain clip loop (100 alone: plain loop, automatic cap)
    if (cap >= 100) cap -= 100;
    g_order_auto = cap == 0;
    g_order_dim = cap < 0 ? 3 : 2;
    const int c = cap < 0 ? -cap : cap;
    g_order_cap = cap == 0 ? 10 : c <= 5 ? 5 : c <= 8 ? 8 : c <= 10 ? 10 : 15;
}

template <bool TRI, int CAP, int DIM, bool ROUNDS>
static int launch_ptm_order_as(const double *dx, const double *dy, const double *dz, int64_t N, const DBox &b, const int *dv, int64_t M, int8_t *dord,
                               int *dnbr, unsigned char *redo, int *redo_count, hipStream_t st)
{
    using namespace ptms;
    const dim3 grid((unsigned)((N + ORD_APB - 1) / ORD_APB)), block(ORD_THREADS);
    const dim3 small(grid.x < 1024u ? grid.x : 1024u);
    // the second pass may ask for more than the 64 KB of dynamic LDS a launch gets by default
    MDH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ptm_order_faces<TRI, 28, true, DIM, ROUNDS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(order_lds_bytes<28, DIM>())));
    const size_t lds1 = order_lds_bytes<CAP, DIM>(), lds2 = order_lds_bytes<28, DIM>();
    hipLaunchKernelGGL((k_ptm_order_faces<TRI, CAP, false, DIM, ROUNDS>), grid, block, lds1, st, dx, dy, dz, N, b, dv, M, dord, dnbr, redo, redo_count);
    hipLaunchKernelGGL((k_ptm_order_faces<TRI, 28, true, DIM, ROUNDS>), small, block, lds2, st, dx, dy, dz, N, b, dv, M, dord, dnbr, redo, redo_count);
    MDH_HIP(hipGetLastError());
    return MDH_OK;
}

int launch_ptm_order(const double *dx, const double *dy, const double *dz, int64_t N, const DBox &b, const int *dv, int64_t M, int8_t *dord,
                     int *dnbr, unsigned char *redo, int *redo_count, hipStream_t st)
{
    ProfRange pr("k_ptm_order", st);
    MDH_HIP(hipMemsetAsync(redo_count, 0, 2 * sizeof(int), st));
    if (!g_order_stat) {
        MDH_HIP(hipHostMalloc(reinterpret_cast<void **>(&g_order_stat), sizeof(int), hipHostMallocDefault));
        *g_order_stat = 0;
    }
    if (g_order_auto)
        g_order_cap = (g_order_stat_n == N && (int64_t)(*(volatile int *)g_order_stat) * 50 < N) ? 8 : 10;
    const int ran_cap = g_order_cap;
    struct Note { // after the passes: the statistic of this call on its way to the host
        int *count; int cap; int64_t n; hipStream_t st;
        ~Note() { (void)hipMemcpyAsync(g_order_stat, cap > 8 ? count + 1 : count, sizeof(int), hipMemcpyDeviceToHost, st); g_order_stat_n = n; }
    } note{redo_count, ran_cap, N, st};
#define MDH_ORD(TRI, CAP, DIM)                                                                                                                  \
    do {                                                                                                                                       \
        if (DIM == 2 && g_order_rounds) return launch_ptm_order_as<TRI, CAP, DIM, (DIM == 2)>(dx, dy, dz, N, b, dv, M, dord, dnbr, redo, redo_count, st); \
        return launch_ptm_order_as<TRI, CAP, DIM, false>(dx, dy, dz, N, b, dv, M, dord, dnbr, redo, redo_count, st);                          \
    } while (0)
#define MDH_ORD_CAP(TRI, DIM)                                                                                                                  \
    do {                                                                                                                                       \
        if (g_order_cap == 5) MDH_ORD(TRI, 5, DIM); /* test hook: a first pass so small that most atoms take the second one */                 \
        if (g_order_cap == 8) MDH_ORD(TRI, 8, DIM);                                                                                            \
        if (g_order_cap == 10) MDH_ORD(TRI, 10, DIM);                                                                                          \
        MDH_ORD(TRI, 15, DIM);                                                                                                                 \
    } while (0)
    if (g_order_dim == 2) {
        if (b.tri) MDH_ORD_CAP(true, 2);
        MDH_ORD_CAP(false, 2);
    }
    if (b.tri) MDH_ORD_CAP(true, 3);
    MDH_ORD_CAP(false, 3);
#undef MDH_ORD_CAP
#undef MDH_ORD
}

size_t ptm_stage_bytes(int64_t N)
{
    const size_t n = (size_t)((N + 255) & ~int64_t(255));
    const size_t per_kind = ptms::MAXF * 2 + 1 + 8 + 17 + 1;                 // facets, status, hash, label, ok
    const size_t clusters = (17 * 4 + 17 * 24 + 1) + (10 * 4 + 10 * 24 + 1); // diamond and graphene clusters: ids, points, ok
    return n * (ptms::NCANON * per_kind + clusters) + 4096;
}

int launch_ptm_stages(const double *dx, const double *dy, const double *dz, int64_t N, const DBox &b, const int *nbr, const int8_t *orders,
                      const int *dtypes, const ptmc::Tables *dt, const void *dmatch, int flags, double rmsd_threshold, double *dout, int ncol,
                      int *dind, int nind, unsigned char *work, hipStream_t st)
{
    using namespace ptms;
    const MatchTables *dmt = static_cast<const MatchTables *>(dmatch);
    const size_t n = (size_t)((N + 255) & ~int64_t(255));
    HullOut ho;
    uint16_t *facets[NCANON];
    int8_t *status[NCANON];
    CanonOut co[NCANON];
    MatchIn mi;
    ShellSet dc, gr;
    unsigned char *p = work;
    // 8-byte items first, then 4, 2, 1: every array stays aligned
    for (int k = 0; k < NCANON; ++k) { co[k].hash = reinterpret_cast<uint64_t *>(p); p += n * 8; }
    dc.pts = reinterpret_cast<double *>(p); p += n * 17 * 24;
    gr.pts = reinterpret_cast<double *>(p); p += n * 10 * 24;
    dc.ids = reinterpret_cast<int *>(p); p += n * 17 * 4;
    gr.ids = reinterpret_cast<int *>(p); p += n * 10 * 4;
    for (int k = 0; k < NCANON; ++k) { facets[k] = reinterpret_cast<uint16_t *>(p); p += n * MAXF * 2; }
    for (int k = 0; k < NCANON; ++k) { status[k] = reinterpret_cast<int8_t *>(p); p += n; }
    for (int k = 0; k < NCANON; ++k) { co[k].label = reinterpret_cast<int8_t *>(p); p += n * 17; }
    for (int k = 0; k < NCANON; ++k) { co[k].ok = reinterpret_cast<int8_t *>(p); p += n; }
    dc.ok = reinterpret_cast<int8_t *>(p); p += n;
    gr.ok = reinterpret_cast<int8_t *>(p); p += n;
    for (int k = 0; k < NKIND; ++k) { ho.facets[k] = facets[k]; ho.status[k] = status[k]; }
    const int want[NCANON] = {ptmc::CHECK_SC, ptmc::CHECK_FCC | ptmc::CHECK_HCP | ptmc::CHECK_ICO, ptmc::CHECK_BCC, ptmc::CHECK_DCUB | ptmc::CHECK_DHEX};
    const int single = want[K_SC] | want[K_FCC] | want[K_BCC];
    const bool shell = (flags & (want[K_DC] | ptmc::CHECK_GRAPHENE)) != 0;
    const dim3 grid((unsigned)((N + BLK - 1) / BLK)), block(BLK);
    extern const ptmc::Tables *ptm_host_tables();
    const ptmc::Tables &H = *ptm_host_tables();
    if (flags & single) {
        ProfRange pr("k_ptm_hull", st);
        if (b.tri)
            hipLaunchKernelGGL(k_ptm_hull<true>, grid, block, HullMem<15>::BYTES, st, dx, dy, dz, N, b, nbr, flags, ho);
        else
            hipLaunchKernelGGL(k_ptm_hull<false>, grid, block, HullMem<15>::BYTES, st, dx, dy, dz, N, b, nbr, flags, ho);
    }
    if (flags & want[K_DC]) {
        ProfRange pr("k_ptm_shell", st);
        const size_t lds = (size_t)BLK * 17 * 4;
        if (b.tri)
            hipLaunchKernelGGL((k_ptm_shell<true, 4, 3>), grid, block, lds, st, dx, dy, dz, N, b, nbr, orders, dc);
        else
            hipLaunchKernelGGL((k_ptm_shell<false, 4, 3>), grid, block, lds, st, dx, dy, dz, N, b, nbr, orders, dc);
        hipLaunchKernelGGL(k_ptm_hull_shell, grid, block, HullMem<17>::BYTES, st, N, dc, H.types[ptmc::T_DCUB].max_degree, facets[K_DC], status[K_DC]);
    }
    if (flags & ptmc::CHECK_GRAPHENE) {
        ProfRange pr("k_ptm_shell", st);
        const size_t lds = (size_t)BLK * 10 * 4;
        if (b.tri)
            hipLaunchKernelGGL((k_ptm_shell<true, 3, 2>), grid, block, lds, st, dx, dy, dz, N, b, nbr, orders, gr);
        else
            hipLaunchKernelGGL((k_ptm_shell<false, 3, 2>), grid, block, lds, st, dx, dy, dz, N, b, nbr, orders, gr);
    }
    {
        ProfRange pr("k_ptm_canon", st);
        if (flags & want[K_SC])
            hipLaunchKernelGGL((k_ptm_canon<6, false>), grid, block, CanonMem<6>::BYTES, st, N, facets[K_SC], status[K_SC], H.types[ptmc::T_SC].max_degree, 4, co[K_SC]);
        if (flags & want[K_FCC])
            hipLaunchKernelGGL((k_ptm_canon<12, false>), grid, block, CanonMem<12>::BYTES, st, N, facets[K_FCC], status[K_FCC], H.types[ptmc::T_FCC].max_degree, 0, co[K_FCC]);
        if (flags & want[K_BCC])
            hipLaunchKernelGGL((k_ptm_canon<14, false>), grid, block, CanonMem<14>::BYTES, st, N, facets[K_BCC], status[K_BCC], H.types[ptmc::T_BCC].max_degree, 0, co[K_BCC]);
        if (flags & want[K_DC])
            hipLaunchKernelGGL((k_ptm_canon<16, true>), grid, block, CanonMem<16>::BYTES, st, N, facets[K_DC], status[K_DC], H.types[ptmc::T_DCUB].max_degree, 0, co[K_DC]);
    }
    for (int k = 0; k < NCANON; ++k) {
        const bool on = (flags & want[k]) != 0;
        mi.hash[k] = on ? co[k].hash : nullptr;
        mi.label[k] = on ? co[k].label : nullptr;
        mi.ok[k] = on ? co[k].ok : nullptr;
    }
    mi.dc = dc;
    mi.gr = gr;
    {
        ProfRange pr("k_ptm_match", st);
#define MDH_PTM_MATCH(TRI, SHELL)                                                                                                        \
    hipLaunchKernelGGL((k_ptm_match<TRI, SHELL>), grid, block, MatchMem<SHELL ? 17 : 15>::BYTES, st, dx, dy, dz, N, b, nbr, dtypes, dt, dmt, \
                       flags, mi, rmsd_threshold, dout, ncol, dind, nind)
        if (b.tri && shell) MDH_PTM_MATCH(true, true);
        else if (b.tri) MDH_PTM_MATCH(true, false);
        else if (shell) MDH_PTM_MATCH(false, true);
        else MDH_PTM_MATCH(false, false);
#undef MDH_PTM_MATCH
    }
    return MDH_OK;
}

} // namespace mdh

MDH_WARM_UNIT(ptm_stages)
