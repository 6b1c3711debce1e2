// knn.hip — exact k nearest neighbours (periodic images are distinct candidates) on gfx950.
//
// Replaces src/fast_knn.cpp:846-916 (knn) with its kd-trees (:208-568, :588-794).
// Exact kNN is defined by its result; the candidate set and the distance
// arithmetic are the reference's:
//   wrap    orthogonal  s = floor((p-O)*(1/L)); if (s != 0) p -= s*L            (:688-703, :743-757)
//           triclinic   r = p.inv (NO origin shift); s = floor(r_d); p -= s*row_d (:86-99)
//   images  +-nimages per periodic axis, nimages = 200/clamp(N,50,200), >= 2 if triclinic (:801-841)
//   d2      q = q_wrapped - shift;  d = a - q;  d2 = dx*dx + dy*dy + dz*dz     (:598-603, :421-425, :759-770)
//   self    skipped only when idx == self && d2 == 0.0                          (:641, :525)
// Order under EXACT ties in d2 is traversal dependent in the reference; here ties
// are ordered by atom index.
//
// Search structure: a uniform grid over the (wrapped) atoms; each query walks
// Chebyshev rings of cells around its own cell — image cells beyond the box map
// to (cell, shift) — and stops once the k-th best distance is no larger than
// the distance to anything outside the rings visited so far (k_knn, sorted list
// in LDS).  For k <= 24 the common case — all k within one cell width, i.e. in
// the 27 cells around the query — is served first by k_knn_near with the sorted
// list in registers; the queries it cannot prove go to k_knn through a to-do list.
#include "common.hpp"
#include "grid.hpp"
#include <cstring>
#include <type_traits>
#include <cstdlib>

namespace mdh {

int neighbor_rows_device(Scope &sc, const double *dx, const double *dy, const double *dz, int64_t N, const DBox &b, double rc, int *dv,
                         double *dd, int *dn, int64_t M, const int64_t *dkey, bool ids_only); // neighbor.hip

int g_knn_variant = 0; // 0 = near kernel + general kernel, 1 = general kernel only (tests, A/B), 3 = counting kernel first (measuring variant)

struct KnnGeom {
    int nim[3];    // images per axis (0 on open axes)
    double wmin;   // smallest perpendicular cell width
    int rmax;      // ring index after which every (cell, image) has been visited
};

template <bool TRI>
__global__ __launch_bounds__(256) void k_knn_wrap(const double *__restrict__ x, const double *__restrict__ y,
                                                  const double *__restrict__ z, int64_t N, DBox b,
                                                  double *__restrict__ wx, double *__restrict__ wy,
                                                  double *__restrict__ wz)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    double px = x[i], py = y[i], pz = z[i];
    if (TRI) {
        const double r0 = px * b.hi[0] + py * b.hi[3] + pz * b.hi[6];
        const double r1 = px * b.hi[1] + py * b.hi[4] + pz * b.hi[7];
        const double r2 = px * b.hi[2] + py * b.hi[5] + pz * b.hi[8];
        const double r[3] = {r0, r1, r2};
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (b.pbc[d]) {
                const double s = floor(r[d]);
                if (s != 0.0) { px -= s * b.h[d * 3 + 0]; py -= s * b.h[d * 3 + 1]; pz -= s * b.h[d * 3 + 2]; }
            }
    } else {
        if (b.pbc[0]) { const double s = floor((px - b.o[0]) * (1.0 / b.h[0])); if (s != 0.0) px -= s * b.h[0]; }
        if (b.pbc[1]) { const double s = floor((py - b.o[1]) * (1.0 / b.h[4])); if (s != 0.0) py -= s * b.h[4]; }
        if (b.pbc[2]) { const double s = floor((pz - b.o[2]) * (1.0 / b.h[8])); if (s != 0.0) pz -= s * b.h[8]; }
    }
    wx[i] = px; wy[i] = py; wz[i] = pz;
}

__device__ __forceinline__ int floordiv(int a, int n) { int q = a / n; return (a % n < 0) ? q - 1 : q; }

// ---------------------------------------------------------------------------------------------------------------------------------
// The k nearest from the rows of a CUTOFF build (round 6).  For a system of even density — a crystal, a glass, a liquid: what the
// k-nearest analyses are run on — the k nearest of every atom lie inside a radius r a little beyond the one that holds k + 1 atoms at
// the system's density, and the LDS-tile neighbour kernel (neighbor_lane.hip: single-precision scan of staged cells, ~1 ns per atom)
// lists everything inside r three to four times faster than a thread can walk its 27 cells.  k_knn_rows then sees ~k + 6 candidates
// per query instead of ~110: it recomputes every candidate's squared distance with the k-nearest search's OWN expression (the
// reference wraps both atoms and shifts the query by whole box vectors, fast_knn.cpp:598-603, 759-770; the cutoff build folds the raw
// separation, neighbor.cpp:139-177 — the two round differently across a periodic boundary) and keeps the k smallest by (squared
// distance, index or key).  A query is finished here only if its row is complete (fewer entries than slots), holds k entries and its
// k-th distance lies safely inside r — then nothing outside the row can be nearer; the others are flagged and go through the cell walk
// (k_knn_near with `only`, then k_knn).  Rows are those of the plain search, bit for bit; r only decides who takes which path.
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool TRI>
__global__ __launch_bounds__(256) void k_knn_wrap4(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                   int64_t N, DBox b, Pos4 *__restrict__ w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    double px = x[i], py = y[i], pz = z[i]; // (k_knn_wrap)
    if (TRI) {
        const double r0 = px * b.hi[0] + py * b.hi[3] + pz * b.hi[6];
        const double r1 = px * b.hi[1] + py * b.hi[4] + pz * b.hi[7];
        const double r2 = px * b.hi[2] + py * b.hi[5] + pz * b.hi[8];
        const double r[3] = {r0, r1, r2};
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (b.pbc[d]) {
                const double s = floor(r[d]);
                if (s != 0.0) { px -= s * b.h[d * 3 + 0]; py -= s * b.h[d * 3 + 1]; pz -= s * b.h[d * 3 + 2]; }
            }
    } else {
        if (b.pbc[0]) { const double s = floor((px - b.o[0]) * (1.0 / b.h[0])); if (s != 0.0) px -= s * b.h[0]; }
        if (b.pbc[1]) { const double s = floor((py - b.o[1]) * (1.0 / b.h[4])); if (s != 0.0) py -= s * b.h[4]; }
        if (b.pbc[2]) { const double s = floor((pz - b.o[2]) * (1.0 / b.h[8])); if (s != 0.0) pz -= s * b.h[8]; }
    }
    w[i] = Pos4{px, py, pz, 0.0};
}

// rows (N, M) of a cutoff build with radius r (counts nn keep running past M), K >= k list slots.  flag[i] = 1 and *nflag += 1 for a
// query that must take the cell walk.  key: NULL, or the tie-breaking number of every atom (mdh_knn_keyed)
template <bool TRI, int K, int MROW, bool KEYED>
__global__ __launch_bounds__(256) void k_knn_rows(const Pos4 *__restrict__ w, int64_t N, DBox b, const int *__restrict__ rows, int M,
                                                  const int *__restrict__ nn, double safe2, int k, const int64_t *__restrict__ key,
                                                  int *__restrict__ indices, double *__restrict__ distances,
                                                  unsigned char *__restrict__ flag, int *__restrict__ nflag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    const int cnt = nn[i];
    bool ok = cnt >= k && cnt <= M && cnt <= MROW; // (rows kept from a search with another k may be wider or narrower than MROW)
    double ld[K]; // the K best so far, sorted by (squared distance, tie number); free slots hold +inf
    int li[K];
    typedef typename std::conditional<KEYED, int64_t, int>::type Tie; // (without a key the index itself breaks ties: no third array)
    Tie lt[KEYED ? K : 1];
#pragma unroll
    for (int e = 0; e < K; ++e) { ld[e] = __builtin_huge_val(); li[e] = 0x7fffffff; if (KEYED) lt[e] = (Tie)0x7fffffffffffffffll; }
    auto tie_at = [&](int e) -> Tie { if constexpr (KEYED) return lt[e]; else return (Tie)li[e]; };
    if (ok) {
        const Pos4 q = w[i];
        // the whole row at once (MROW = M ids in 16-byte requests, back to back): a lane's row is one 128-byte line and a wavefront's
        // rows are 64 of them — read four ids at a time as the loop goes, every step fetched all 64 lines again (they do not survive in
        // the CU's 16 KB cache between steps with a dozen wavefronts resident: the kernel ran FASTER with fewer of them)
        int rowv[MROW];
        {
            const int *__restrict__ row = rows + i * (int64_t)M;
#pragma unroll
            for (int q = 0; q < MROW / 4; ++q) {
                RowQuad v = {-1, -1, -1, -1};
                if (4 * q < M) v = *reinterpret_cast<const RowQuad *>(row + 4 * q); // (M: a multiple of four, uniform)
                rowv[4 * q] = v.x; rowv[4 * q + 1] = v.y; rowv[4 * q + 2] = v.z; rowv[4 * q + 3] = v.w;
            }
        }
        static_assert(MROW % 4 == 0, "rows in quads");
#pragma unroll
        for (int e0 = 0; e0 < MROW; e0 += 4) {
            if (e0 >= cnt)
                break;
            int cj[4];
            Pos4 c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) cj[u] = e0 + u < cnt ? rowv[e0 + u] : rowv[e0];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = w[cj[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (e0 + u >= cnt)
                    continue;
                // the image of the candidate nearest to the query: whole box vectors m between the wrapped positions (the box is at
                // least seven cutoffs wide along every periodic vector — the tile kernel's condition, checked by the caller — so the
                // fractional separation is far from +-1/2 and the rounding is not in doubt), then fast_knn.cpp's expression
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
                const double ex = c[u].x - q.x, ey = c[u].y - q.y, ez = c[u].z - q.z;
                if (TRI) {
                    const double f0 = ex * b.hi[0] + ey * b.hi[3] + ez * b.hi[6];
                    const double f1 = ex * b.hi[1] + ey * b.hi[4] + ez * b.hi[7];
                    const double f2 = ex * b.hi[2] + ey * b.hi[5] + ez * b.hi[8];
                    const int m0 = b.pbc[0] ? -(int)rint(f0) : 0, m1 = b.pbc[1] ? -(int)rint(f1) : 0, m2 = b.pbc[2] ? -(int)rint(f2) : 0;
                    s0 = m0 * b.h[0] + m1 * b.h[3] + m2 * b.h[6];
                    s1 = m0 * b.h[1] + m1 * b.h[4] + m2 * b.h[7];
                    s2 = m0 * b.h[2] + m1 * b.h[5] + m2 * b.h[8];
                } else {
                    const int m0 = b.pbc[0] ? -(int)rint(ex * (1.0 / b.h[0])) : 0, m1 = b.pbc[1] ? -(int)rint(ey * (1.0 / b.h[4])) : 0,
                              m2 = b.pbc[2] ? -(int)rint(ez * (1.0 / b.h[8])) : 0;
                    s0 = m0 * b.h[0]; s1 = m1 * b.h[4]; s2 = m2 * b.h[8];
                }
                const double w0 = q.x - s0, w1 = q.y - s1, w2 = q.z - s2;
                const double dx = c[u].x - w0, dy = c[u].y - w1, dz = c[u].z - w2;
                const double d2 = dx * dx + dy * dy + dz * dz;
                const int j = cj[u];
                Tie tj;
                if constexpr (KEYED) tj = key[j]; else tj = j;
                if (!(d2 < ld[K - 1] || (d2 == ld[K - 1] && tj < tie_at(K - 1))))
                    continue;
                int pos = 0; // entries that stay in front of the new one
#pragma unroll
                for (int e = 0; e < K; ++e)
                    pos += (ld[e] < d2 || (ld[e] == d2 && tie_at(e) < tj)) ? 1 : 0;
#pragma unroll
                for (int e = K - 1; e >= 1; --e) {
                    if (e > pos) { ld[e] = ld[e - 1]; li[e] = li[e - 1]; if (KEYED) lt[e] = lt[e - 1]; }
                    else if (e == pos) { ld[e] = d2; li[e] = j; if (KEYED) lt[e] = tj; }
                }
                if (pos == 0) { ld[0] = d2; li[0] = j; if (KEYED) lt[0] = tj; }
            }
        }
        double kth = __builtin_huge_val();
#pragma unroll
        for (int e = 0; e < K; ++e)
            if (e == k - 1) kth = ld[e];
        ok = kth <= safe2; // (the k-th inside r with room to spare: an atom the cutoff build left out is farther than r (1 - 1e-12))
    }
    flag[i] = ok ? 0 : 1;
    if (!ok) {
        atomicAdd(nflag, 1);
        return;
    }
#pragma unroll
    for (int e = 0; e < K; ++e)
        if (e < k) {
            indices[i * (int64_t)k + e] = li[e];
            distances[i * (int64_t)k + e] = sqrt(ld[e]); // :883
        }
}

// The common case — the k nearest all lie within one cell width, in the 27 cells around the query — with the sorted list in
// REGISTERS (K slots, K >= k a template constant): an insertion is a fully unrolled count of the entries that stay in front
// (the position) and one predicated move per slot, ~14 instructions per slot and no memory access, where the list in LDS
// pays two dependent LDS round trips per shifted entry at two waves per SIMD (measured: 15 of 21 ms at k = 18 went into
// shifting).  No LDS at all, so occupancy is set by the registers: k <= 18 is held to 128 VGPRs (four waves per SIMD; 9 - 26
// values spill) — 6.7 -> 6.0 ms at k = 12, 11.5 -> 10.1 ms at k = 18 for 10 M atoms; k = 24 loses with it (20 -> 26 ms) and keeps three.  A query that does not find k candidates within one cell
// width is appended to `todo` and finished by k_knn (any number of rings, list in LDS).
template <bool TRI, int K>
__global__ __launch_bounds__(256, (K <= 18 ? 4 : 3)) void k_knn_near(const double *__restrict__ xs, const double *__restrict__ ys,
                                                  const double *__restrict__ zs, const int *__restrict__ order,
                                                  const int *__restrict__ cell_start, int64_t N, DBox b, DBox bg, Grid g,
                                                  KnnGeom kg, int k, int *__restrict__ indices,
                                                  double *__restrict__ distances, int *__restrict__ todo,
                                                  const int *__restrict__ label, const int *__restrict__ unlabel,
                                                  const int *__restrict__ listed = nullptr, const unsigned char *__restrict__ only = nullptr)
{
    // only (N bytes by atom index, or NULL): the queries whose byte is set — the ones k_knn_rows could not finish
    // label / unlabel (both NULL, or both given): candidates are told apart — the self test, the order under exact ties — by
    // label[q] instead of their index order[q], and a listed label L is written as unlabel[L] (mdh_knn_keyed: the labels are the
    // caller's key, a permutation of 0 .. N-1, so that ties fall as they would in the system the key numbers)
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (listed) { // the queries k_knn_select could not finish (listed[0] of them, positions in the cell-sorted arrays)
        if (p >= listed[0])
            return;
        p = listed[1 + p];
    } else if (p >= N) {
        return;
    }
    const int i = order[p];
    if (only && !only[i])
        return;
    const int self = label ? label[p] : i;
    const int *__restrict__ cand = label ? label : order;
    const double qx = xs[p], qy = ys[p], qz = zs[p]; // wrapped query == stored wrapped self (bitwise)
    int c0, c1, c2;
    cell_coords<TRI>(bg, g, qx, qy, qz, c0, c1, c2);
    double ld[K]; // the K best so far, sorted by (squared distance, id); free slots hold +inf
    int li[K];
#pragma unroll
    for (int e = 0; e < K; ++e) { ld[e] = __builtin_huge_val(); li[e] = 0x7fffffff; }
    const double one_cell = kg.wmin * (1.0 - 1e-9), bound = one_cell * one_cell;
    auto fold_cell = [&](int d, int e, int &a, int &m) {
        m = 0; a = e;
        if (b.pbc[d]) { m = floordiv(e, g.nc[d]); a = e - m * g.nc[d]; return !(m > kg.nim[d] || m < -kg.nim[d]); }
        return e >= 0 && e < g.nc[d];
    };
    for (int col9 = 0; col9 < 9; ++col9) { // nearest columns first: the k-th distance tightens early
        const int da = (0x28161 >> (2 * col9) & 3) - 1, db = (0x22215 >> (2 * col9) & 3) - 1; // (0,0) (-1,0) (1,0) (0,-1) (0,1) (-1,-1) (-1,1) (1,-1) (1,1)
        int a0, m0, a1, m1;
        if (!fold_cell(0, c0 + da, a0, m0) || !fold_cell(1, c1 + db, a1, m1)) continue;
        const int64_t col = ((int64_t)a0 * g.nc[1] + a1) * g.nc[2];
        for (int e2 = c2 - 1; e2 <= c2 + 1;) {
            int a2, m2;
            if (!fold_cell(2, e2, a2, m2)) { ++e2; continue; }
            int len = 1; // cells of this column with the same image number: one contiguous piece of the sorted arrays
            while (e2 + len <= c2 + 1 && a2 + len < g.nc[2]) ++len;
            double s0, s1, s2; // image shift, fast_knn.cpp:822-833 (see k_knn)
            if (TRI) {
                s0 = m0 * b.h[0] + m1 * b.h[3] + m2 * b.h[6];
                s1 = m0 * b.h[1] + m1 * b.h[4] + m2 * b.h[7];
                s2 = m0 * b.h[2] + m1 * b.h[5] + m2 * b.h[8];
            } else {
                s0 = m0 * b.h[0]; s1 = m1 * b.h[4]; s2 = m2 * b.h[8];
            }
            const double w0 = qx - s0, w1 = qy - s1, w2 = qz - s2;
            const int sb = cell_start[col + a2], se = cell_start[col + a2 + len];
            for (int q0 = sb; q0 < se; q0 += 8) {
                double d2s[8];
                int cj[8];
                unsigned todo_mask = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) { // 32 loads in flight together
                    const int q = min(q0 + u, se - 1);
                    const double dx = xs[q] - w0, dy = ys[q] - w1, dz = zs[q] - w2;
                    cj[u] = cand[q];
                    d2s[u] = dx * dx + dy * dy + dz * dz;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool live = q0 + u < se && !(cj[u] == self && d2s[u] == 0.0) && !(d2s[u] > bound) &&
                                      (d2s[u] < ld[K - 1] || (d2s[u] == ld[K - 1] && cj[u] < li[K - 1]));
                    todo_mask |= live ? 1u << u : 0u;
                }
                while (todo_mask) { // every lane inserts ITS next live candidate per round
                    const int u = __builtin_ctz(todo_mask);
                    todo_mask &= todo_mask - 1;
                    double d2 = d2s[0];
                    int j = cj[0];
#pragma unroll
                    for (int v = 1; v < 8; ++v)
                        if (u == v) { d2 = d2s[v]; j = cj[v]; }
                    int pos = 0; // entries that stay in front of the new one
#pragma unroll
                    for (int e = 0; e < K; ++e)
                        pos += (ld[e] < d2 || (ld[e] == d2 && li[e] < j)) ? 1 : 0;
#pragma unroll
                    for (int e = K - 1; e >= 1; --e) {
                        if (e > pos) { ld[e] = ld[e - 1]; li[e] = li[e - 1]; }
                        else if (e == pos) { ld[e] = d2; li[e] = j; }
                    }
                    if (pos == 0) { ld[0] = d2; li[0] = j; }
                }
            }
            e2 += len;
        }
    }
    // k candidates within one cell width: nothing outside the 27 cells can be nearer (the stop test of ring 1)
    bool full = true;
#pragma unroll
    for (int e = 0; e < K; ++e)
        if (e < k && !(ld[e] <= bound)) full = false;
    if (!full) {
        todo[1 + atomicAdd(&todo[0], 1)] = (int)p;
        return;
    }
#pragma unroll
    for (int e = 0; e < K; ++e)
        if (e < k) {
            indices[(int64_t)i * k + e] = unlabel ? unlabel[li[e]] : li[e];
            distances[(int64_t)i * k + e] = sqrt(ld[e]); // :883
        }
}

// The same common case by COUNTING instead of streaming insertion (round 5) — A MEASURING VARIANT (mdh_debug_set_knn_variant(3),
// tools/knn_ab.py -> profiles/r05_knn_counting.txt): rows identical to k_knn_near's, 1.05-1.3x SLOWER; not the product's path.  k_knn_near keeps a sorted list and inserts every
// candidate that beats its k-th entry: ~14 instructions per list slot per insertion, and a wavefront runs as many insertion rounds per
// batch of candidates as its busiest lane needs — 35 k wave-instructions per 64 queries at k = 18, 85 % of them insertions.  Here a
// query walks its 27 cells twice.  Pass 1 counts the candidates inside NT trial radii around the radius a uniform system of this
// density would need for k neighbours (volume steps of 15 %; none beyond one cell width, inside which the 27 cells are complete).  The
// smallest radius that holds at least k — and at most KC — candidates is the query's; pass 2 appends exactly those to an unsorted
// register list (three predicated moves per slot), and every entry finds its place in the row by counting the entries in front of it:
// the row is written straight from the ranks, no sorted list is ever maintained.  Same candidates, same squared distances, same order
// under ties (distance, then id or label) as k_knn_near: the rows are identical.  A query with fewer than k candidates inside one cell
// width, or more than KC inside its radius (shells of a perfect lattice, a surface atom of a cluster whose global density says little),
// is listed for k_knn_near, whose leftovers go on to k_knn as before.
// Trial radii as BUCKETS of the squared distance, found without a comparison: the bits of a positive float grow with its value and are
// a piecewise-linear log2, so bucket(d2) = clamp(mul_hi(bits((float)d2) - base, scale), 0, 7) is a monotone step function of d2 whose
// steps are ~15 % apart in volume.  Monotone is all that is needed: "every candidate of bucket <= j" is a ball around the query for every j.
constexpr int KNN_NT = 6; // buckets 0 .. 5 are counted; 6, 7: farther than any trial radius
struct KnnTrial { int base; unsigned scale; float bound_f; };
template <bool TRI, int KC>
__global__ __launch_bounds__(128) void k_knn_select(const double *__restrict__ xs, const double *__restrict__ ys,
                                                  const double *__restrict__ zs, const int *__restrict__ order,
                                                  const int *__restrict__ cell_start, int64_t N, DBox b, DBox bg, Grid g,
                                                  KnnGeom kg, int k, KnnTrial trial, int *__restrict__ indices,
                                                  double *__restrict__ distances, int *__restrict__ failed,
                                                  const int *__restrict__ label, const int *__restrict__ unlabel)
{
    // the query's list, UNSORTED, in a stripe of LDS (entry e of thread t at [e * 128 + t]: bank = lane): an append is two stores at a
    // per-lane index, which registers cannot do without a predicated move per slot
    __shared__ double s_d[KC * 128];
    __shared__ int s_l[KC * 128];
    const int t = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + t;
    if (p >= N)
        return;
    const int i = order[p];
    const int self = label ? label[p] : i;
    const int *__restrict__ cand = label ? label : order;
    const double qx = xs[p], qy = ys[p], qz = zs[p];
    int c0, c1, c2;
    cell_coords<TRI>(bg, g, qx, qy, qz, c0, c1, c2);
    auto fold_cell = [&](int d, int e, int &a, int &m) {
        m = 0; a = e;
        if (b.pbc[d]) { m = floordiv(e, g.nc[d]); a = e - m * g.nc[d]; return !(m > kg.nim[d] || m < -kg.nim[d]); }
        return e >= 0 && e < g.nc[d];
    };
    auto bucket_of = [&](double d2, bool real) {
        const float f = (float)d2;
        int bk = (int)__umulhi((unsigned)max(__float_as_int(f) - trial.base, 0), trial.scale);
        bk = min(bk, 7);
        // beyond one cell width the 27 cells are not complete: such a candidate belongs to no trial radius
        return (real && f < trial.bound_f) ? bk : 7;
    };
    // the walk of k_knn_near (nearest columns first), eight candidates per trip; visit(d2[8], label[8], bucket[8]).  The query itself
    // (distance 0 in its own cell) is walked like everybody: it is counted, listed and left out when the row is written.
    auto walk = [&](auto &&visit) {
        for (int col9 = 0; col9 < 9; ++col9) {
            const int da = (0x28161 >> (2 * col9) & 3) - 1, db = (0x22215 >> (2 * col9) & 3) - 1;
            int a0, m0, a1, m1;
            if (!fold_cell(0, c0 + da, a0, m0) || !fold_cell(1, c1 + db, a1, m1)) continue;
            const int64_t col = ((int64_t)a0 * g.nc[1] + a1) * g.nc[2];
            for (int e2 = c2 - 1; e2 <= c2 + 1;) {
                int a2, m2;
                if (!fold_cell(2, e2, a2, m2)) { ++e2; continue; }
                int len = 1;
                while (e2 + len <= c2 + 1 && a2 + len < g.nc[2]) ++len;
                double s0, s1, s2;
                if (TRI) {
                    s0 = m0 * b.h[0] + m1 * b.h[3] + m2 * b.h[6];
                    s1 = m0 * b.h[1] + m1 * b.h[4] + m2 * b.h[7];
                    s2 = m0 * b.h[2] + m1 * b.h[5] + m2 * b.h[8];
                } else {
                    s0 = m0 * b.h[0]; s1 = m1 * b.h[4]; s2 = m2 * b.h[8];
                }
                const double w0 = qx - s0, w1 = qy - s1, w2 = qz - s2;
                const int sb = cell_start[col + a2], se = cell_start[col + a2 + len];
                for (int q0 = sb; q0 < se; q0 += 8) {
                    double d2s[8];
                    int cj[8], bk[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int q = min(q0 + u, se - 1);
                        const double dx = xs[q] - w0, dy = ys[q] - w1, dz = zs[q] - w2;
                        cj[u] = cand[q];
                        d2s[u] = dx * dx + dy * dy + dz * dz;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) bk[u] = bucket_of(d2s[u], q0 + u < se);
                    visit(d2s, cj, bk);
                }
                e2 += len;
            }
        }
    };
    // ---- pass 1: candidates per bucket, eight bits each in one register (buckets 0 .. 3) and a second (4, 5; the rest is not counted)
    unsigned acc0 = 0, acc1 = 0;
    walk([&](const double (&)[8], const int (&)[8], const int (&bk)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned one = 1u << ((bk[u] & 3) << 3);
            acc0 += one & (unsigned)((bk[u] - 4) >> 31);                              // bucket < 4
            acc1 += one & (unsigned)(~((bk[u] - 4) >> 31) & ((bk[u] - 6) >> 31));      // bucket 4 or 5
        }
    });
    // the smallest trial radius with k neighbours and the query itself inside
    int sel = -1, want = 0, run = 0;
#pragma unroll
    for (int j = 0; j < KNN_NT; ++j) {
        run += (int)(((j < 4 ? acc0 : acc1) >> ((j & 3) << 3)) & 255u);
        if (sel < 0 && run >= k + 1) { sel = j; want = run; }
    }
    if (sel < 0 || want > KC) { // not k inside one cell width, or a shell too full for the list: the insertion kernel
        failed[1 + atomicAdd(&failed[0], 1)] = (int)p;
        return;
    }
    // ---- pass 2: exactly the candidates of the buckets up to the chosen one, appended as they come
    int n = 0;
    walk([&](const double (&d2s)[8], const int (&cj)[8], const int (&bk)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (bk[u] <= sel) {
                s_d[n * 128 + t] = d2s[u];
                s_l[n * 128 + t] = cj[u];
                ++n;
            }
    });
    // ---- every entry's place in the row: the entries in front of it by (squared distance, label); the query itself is taken out
    double ld[KC];
    int li[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        const bool have = e < n;
        ld[e] = have ? s_d[e * 128 + t] : __builtin_huge_val();
        li[e] = have ? s_l[e * 128 + t] : 0x7fffffff;
    }
    int rank[KC];
#pragma unroll
    for (int e = 0; e < KC; ++e) rank[e] = 0;
#pragma unroll
    for (int e = 0; e < KC; ++e)
#pragma unroll
        for (int f = e + 1; f < KC; ++f) {
            const bool f_first = ld[f] < ld[e] || (ld[f] == ld[e] && li[f] < li[e]);
            rank[e] += f_first ? 1 : 0;
            rank[f] += f_first ? 0 : 1;
        }
    int self_rank = 0x7fffffff; // the query's own entry: label `self` at distance 0 (fast_knn.cpp:641)
#pragma unroll
    for (int e = 0; e < KC; ++e)
        if (e < n && li[e] == self && ld[e] == 0.0) self_rank = min(self_rank, rank[e]);
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        const int r = rank[e] - (rank[e] > self_rank ? 1 : 0);
        if (e < n && rank[e] != self_rank && r < k) {
            indices[(int64_t)i * k + r] = unlabel ? unlabel[li[e]] : li[e];
            distances[(int64_t)i * k + r] = sqrt(ld[e]); // :883
        }
    }
}

// top-k lists live in LDS: entry s of thread t at [s * blockDim + t]
template <bool TRI>
__global__ void k_knn(const double *__restrict__ xs, const double *__restrict__ ys, const double *__restrict__ zs,
                      const int *__restrict__ order, const int *__restrict__ cell_start, int64_t N, DBox b, DBox bg,
                      Grid g, KnnGeom kg, int k, int *__restrict__ indices, double *__restrict__ distances,
                      const int *__restrict__ todo, const int *__restrict__ label, const int *__restrict__ unlabel)
{
    extern __shared__ unsigned char smem[];
    const int bd = blockDim.x, t = threadIdx.x;
    double *td = reinterpret_cast<double *>(smem);                      // [k][bd]
    int *ti = reinterpret_cast<int *>(smem + sizeof(double) * (size_t)k * bd); // [k][bd]
    int64_t p = (int64_t)blockIdx.x * bd + t;
    if (todo) { // the queries k_knn_near could not finish (todo[0] of them, positions in the cell-sorted arrays)
        if (p >= todo[0])
            return;
        p = todo[1 + p];
    } else if (p >= N) {
        return;
    }
    const int i = order[p];
    const int self = label ? label[p] : i; // (labels: as in k_knn_near)
    const int *__restrict__ cand = label ? label : order;
    const double qx = xs[p], qy = ys[p], qz = zs[p]; // wrapped query == stored wrapped self (bitwise)
    int c0, c1, c2;
    cell_coords<TRI>(bg, g, qx, qy, qz, c0, c1, c2);
    int n = 0;
    double worst = __builtin_huge_val();
    int worst_id = 0x7fffffff;
    // the candidates at positions [sb, se) of the cell-sorted arrays, seen from the shifted query w: into the sorted list
    // (squared distance, then id; the k smallest are kept) unless farther than `bound`.  Batches of 8: the 32 loads of a
    // batch are in flight together, and every lane then inserts ITS next live candidate per round — a wave runs
    // max-over-lanes rounds (1-3 of 8) instead of walking all 8 slots with the insertion loop live whenever any lane needs it
    auto scan_range = [&](int sb, int se, double w0, double w1, double w2, double bound) {
        for (int q0 = sb; q0 < se; q0 += 8) {
            double cx[8], cy[8], cz[8];
            int cj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = min(q0 + u, se - 1);
                cx[u] = xs[q]; cy[u] = ys[q]; cz[u] = zs[q]; cj[u] = cand[q];
            }
            double d2s[8];
            unsigned todo = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double dx = cx[u] - w0, dy = cy[u] - w1, dz = cz[u] - w2;
                d2s[u] = dx * dx + dy * dy + dz * dz;
                const bool live = q0 + u < se && !(cj[u] == self && d2s[u] == 0.0) && !(d2s[u] > bound);
                todo |= live ? 1u << u : 0u;
            }
            while (todo) {
                const int u = __builtin_ctz(todo);
                todo &= todo - 1;
                double d2 = d2s[0];
                int j = cj[0];
#pragma unroll
                for (int v = 1; v < 8; ++v)
                    if (u == v) { d2 = d2s[v]; j = cj[v]; }
                if (n == k && !(d2 < worst || (d2 == worst && j < worst_id)))
                    continue;
                int pos = n < k ? n : k - 1;
                while (pos > 0) {
                    const double pd = td[(pos - 1) * bd + t];
                    const int pi = ti[(pos - 1) * bd + t];
                    if (!(pd > d2 || (pd == d2 && pi > j)))
                        break;
                    td[pos * bd + t] = pd;
                    ti[pos * bd + t] = pi;
                    --pos;
                }
                td[pos * bd + t] = d2;
                ti[pos * bd + t] = j;
                if (n < k) ++n;
                if (n == k) { worst = td[(k - 1) * bd + t]; worst_id = ti[(k - 1) * bd + t]; }
            }
        }
    };
    // extended cell index e along axis d -> (stored cell a, image number m); false: no such cell (beyond an open face, or
    // more images away than the reference looks, fast_knn.cpp:806-816)
    auto fold_cell = [&](int d, int e, int &a, int &m) {
        m = 0; a = e;
        if (b.pbc[d]) { m = floordiv(e, g.nc[d]); a = e - m * g.nc[d]; return !(m > kg.nim[d] || m < -kg.nim[d]); }
        return e >= 0 && e < g.nc[d];
    };
    // image shift, fast_knn.cpp:822-833.  An atom stored in cell (a0,a1,a2), seen through the extended cell e = a + m*nc, is
    // the image a + m*L.  Its distance to the query is |a - (q - m*L)|: the reference's shifted query with shift = m*L (:759-763)
    auto shifted = [&](int m0, int m1, int m2, double &w0, double &w1, double &w2) {
        double s0, s1, s2;
        if (TRI) {
            s0 = m0 * b.h[0] + m1 * b.h[3] + m2 * b.h[6];
            s1 = m0 * b.h[1] + m1 * b.h[4] + m2 * b.h[7];
            s2 = m0 * b.h[2] + m1 * b.h[5] + m2 * b.h[8];
        } else {
            s0 = m0 * b.h[0]; s1 = m1 * b.h[4]; s2 = m2 * b.h[8];
        }
        w0 = qx - s0; w1 = qy - s1; w2 = qz - s2;
    };
    // First try: the 27 cells of rings 0 and 1, and only candidates within one cell width (the stop test after ring 1
    // demands k-th distance <= that width, so nothing farther can be in a result found there): a candidate beyond it costs
    // one compare instead of a sorted insertion.  The cells of a column that share an image number are one contiguous
    // piece of the cell-sorted arrays (z runs fastest): 9 pieces of ~3 cells instead of 27 cells, a third of the dependent
    // cell_start -> atoms load chains and batches that are mostly full.  If fewer than k candidates are that close the search
    // starts over without the bound and walks as many rings as it needs — the result is the same set in the same order.
    const double one_cell = kg.wmin * (1.0 - 1e-9);
    bool done = false;
    if (kg.rmax >= 1 && !todo) { // (a query from the to-do list has been through this already)
        const double bound = one_cell * one_cell;
        // nearest columns first — the atom's own, the four that share a face with it, the four diagonal ones: candidates then
        // arrive roughly by increasing distance, so most insertions land near the tail of the sorted list (few entries to
        // shift) and the k-th distance tightens early (far candidates fail one compare instead of being inserted and pushed out
        // again).  The result does not depend on the order: the list is sorted by (distance, id).
        for (int col9 = 0; col9 < 9; ++col9) {
            const int da = (0x28161 >> (2 * col9) & 3) - 1, db = (0x22215 >> (2 * col9) & 3) - 1; // (0,0) (-1,0) (1,0) (0,-1) (0,1) (-1,-1) (-1,1) (1,-1) (1,1)
            {
                int a0, m0, a1, m1;
                if (!fold_cell(0, c0 + da, a0, m0) || !fold_cell(1, c1 + db, a1, m1)) continue;
                const int64_t col = ((int64_t)a0 * g.nc[1] + a1) * g.nc[2];
                for (int e2 = c2 - 1; e2 <= c2 + 1;) {
                    int a2, m2;
                    if (!fold_cell(2, e2, a2, m2)) { ++e2; continue; }
                    int len = 1; // cells of this column with the same image number
                    while (e2 + len <= c2 + 1 && a2 + len < g.nc[2]) ++len;
                    double w0, w1, w2;
                    shifted(m0, m1, m2, w0, w1, w2);
                    scan_range(cell_start[col + a2], cell_start[col + a2 + len], w0, w1, w2, bound);
                    e2 += len;
                }
            }
        }
        done = n == k; // k candidates within one cell width: the stop test of ring 1 has passed
    }
    if (!done) {
        n = 0;
        worst = __builtin_huge_val();
        worst_id = 0x7fffffff;
        for (int R = 0; R <= kg.rmax; ++R) {
            for (int da = -R; da <= R; ++da) {
                int a0, m0;
                if (!fold_cell(0, c0 + da, a0, m0)) continue;
                const int ada = da < 0 ? -da : da;
                for (int db = -R; db <= R; ++db) {
                    int a1, m1;
                    if (!fold_cell(1, c1 + db, a1, m1)) continue;
                    const int adb = db < 0 ? -db : db;
                    const bool shell_ab = (ada == R) || (adb == R);
                    for (int dc = -R; dc <= R; dc += (shell_ab || R == 0) ? 1 : 2 * R) { // interior of the cube was done by earlier rings
                        int a2, m2;
                        if (!fold_cell(2, c2 + dc, a2, m2)) continue;
                        double w0, w1, w2;
                        shifted(m0, m1, m2, w0, w1, w2);
                        const int64_t cell = ((int64_t)a0 * g.nc[1] + a1) * g.nc[2] + a2;
                        scan_range(cell_start[cell], cell_start[cell + 1], w0, w1, w2, __builtin_huge_val());
                    }
                }
            }
            // everything not yet visited lies at Chebyshev cell distance >= R+1, i.e. at least R cell widths away
            if (n == k) {
                const double reach = (double)R * kg.wmin * (1.0 - 1e-9);
                if (worst <= reach * reach)
                    break;
            }
        }
    }
    for (int q = 0; q < n; ++q) {
        indices[(int64_t)i * k + q] = unlabel ? unlabel[ti[q * bd + t]] : ti[q * bd + t];
        distances[(int64_t)i * k + q] = sqrt(td[q * bd + t]); // :883
    }
    for (int q = n; q < k; ++q) { // :885-888
        indices[(int64_t)i * k + q] = -1;
        distances[(int64_t)i * k + q] = -1.0;
    }
}

} // namespace mdh

using namespace mdh;

extern "C" int mdh_debug_set_knn_variant(int v)
{
    g_knn_variant = v;
    return MDH_OK;
}

namespace mdh {
// label[q] = key[order[q]] for the cell-sorted atoms, unlabel[key[i]] = i
__global__ __launch_bounds__(256) void k_knn_labels(const int *__restrict__ order, const int64_t *__restrict__ key, int64_t N,
                                                    int *__restrict__ label, int *__restrict__ unlabel)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N)
        return;
    const int i = order[q];
    const int64_t kv = key[i];
    if (kv < 0 || kv >= N) { label[q] = 0; return; } // (not a permutation: memory-safe, rows meaningless)
    label[q] = (int)kv;
    unlabel[kv] = i;
}
} // namespace mdh

extern "C" int mdh_knn(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                       const double *origin3, const int *boundary3, int k, int *indices, double *distances, int space,
                       void *stream)
{
    return mdh_knn_keyed(x, y, z, N, box9, origin3, boundary3, k, indices, distances, nullptr, space, stream);
}

// key (N) i64, a PERMUTATION of 0 .. N-1, or NULL: the number every atom has in another numbering of the same system (the
// original index of every atom of a cell-sorted copy, mdh_spatial_sort).  Exact ties in distance are then ordered by key instead
// of by index — the rows are the rows mdh_knn gives for the system in the key's numbering, listed neighbour for listed
// neighbour (a perfect lattice is all ties: WHICH four of bcc's six second neighbours are among the twelve nearest depends on it).
extern "C" int mdh_knn_keyed(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                             const double *origin3, const int *boundary3, int k, int *indices, double *distances,
                             const int64_t *key, int space, void *stream)
{
    return mdh_knn_keyed_rows(x, y, z, N, box9, origin3, boundary3, k, indices, distances, key, nullptr, nullptr, 0, nullptr, space, stream);
}

extern "C" int mdh_knn_rows_width(int k) { return k <= 18 ? 32 : 0; } // (one width for every k: rows kept by the caller serve any of them)

// mdh_knn_keyed with the candidate rows of the cutoff build kept BY THE CALLER between searches of the same positions (a System
// that asks for its 12 nearest and then for its 14 nearest: the second search skips the build, half of its time).  rows (N x M i32)
// and counts (N i32): DEVICE buffers (whatever `space` says about the other arguments), M >= mdh_knn_rows_width(k) or the path is
// not taken; *radius (host): in — > 0: rows / counts hold the candidates inside that radius of THESE positions in THIS box (the
// caller vouches for it); 0: build them — out: the radius of what the buffers hold now, 0 when they hold nothing usable.
extern "C" int mdh_knn_keyed_rows(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                  const double *origin3, const int *boundary3, int k, int *indices, double *distances,
                                  const int64_t *key, int *rows_io, int *counts_io, int M_io, double *radius, int space, void *stream)
{
    const double radius_in = radius ? *radius : 0.0;
    if (radius) *radius = 0.0;
    if (N < 0 || N >= 2147483647LL || k <= 0 || k > 64) { set_error("mdh_knn: need 1 <= k <= 64"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    int *di = sc.stage(indices, (size_t)(N * k), space, false, true);
    double *dd = sc.stage(distances, (size_t)(N * k), space, false, true);
    const double vol = std::fabs(b.tri ? (b.h[0] * (b.h[4] * b.h[8] - b.h[5] * b.h[7]) - b.h[1] * (b.h[3] * b.h[8] - b.h[5] * b.h[6]) + b.h[2] * (b.h[3] * b.h[7] - b.h[4] * b.h[6])) : b.h[0] * b.h[4] * b.h[8]);
    const int64_t *dkey = key ? sc.stage_in(key, (size_t)N, space) : nullptr;
    // ---- the k nearest from the rows of a cutoff build (k_knn_rows): large systems in boxes of at least 7.5 radii per periodic vector
    // (the tile kernel's domain, and the nearest image of a listed neighbour is then beyond doubt); MDH_KNN_ROWS=0 or variant != 0: off
    unsigned char *only = nullptr;
    {
        static const bool rows_env = [] { const char *e = std::getenv("MDH_KNN_ROWS"); return !e || std::atoi(e) != 0; }();
      // two attempts at most: with the caller's rows of an earlier search (any k) — and, when those leave more than a few queries
      // unfinished (a radius sized for fewer neighbours), with rows of this k's own radius, built into the caller's buffers
      const double r_own = 1.15 * std::cbrt(3.0 * (double)(k + 1) / (4.0 * 3.14159265358979323846 * ((double)N / vol)));
      // (borrowed rows must reach at least nine tenths of this k's own radius: the rows of a 12-neighbour search serve 14, not 18)
      for (int attempt = (rows_io && counts_io && radius_in >= 0.9 * r_own && M_io == mdh_knn_rows_width(k)) ? 0 : 1; attempt < 2; ++attempt) {
        const bool reuse = attempt == 0;
        const double r = reuse ? radius_in : r_own;
        const char *min_env = std::getenv("MDH_KNN_ROWS_MIN"); // (tests: the path on systems of a few thousand atoms)
        const int64_t min_atoms = min_env ? std::atoll(min_env) : 100000;
        bool fits = rows_env && g_knn_variant == 0 && k <= 18 && N >= min_atoms && r > 0 && std::isfinite(r);
        for (int d = 0; d < 3 && fits; ++d)
            if (b.pbc[d] && !(std::fabs(b.thick[d]) >= 7.5 * r)) fits = false;
        // a system of uneven density (a gas: Poisson counts; a cluster in vacuum) sends many queries on to the cell walk, and the rows
        // were built for nothing: the share of the last search with this (N, k) decides — more than 0.5 %, and the next 15 searches
        // of the signature walk the cells at once (then one more try)
        struct Sig { int64_t N; int k, pbc; double vol; int left_pct, skip; };
        static Sig sigs[16] = {};
        Sig *sig = nullptr;
        const int pbc_bits = (b.pbc[0] ? 1 : 0) | (b.pbc[1] ? 2 : 0) | (b.pbc[2] ? 4 : 0);
        for (auto &e : sigs)
            if (e.N == N && e.k == k && e.pbc == pbc_bits && e.vol == vol) sig = &e;
        if (!sig) {
            static int next = 0;
            sig = &sigs[next++ % 16];
            *sig = Sig{N, k, pbc_bits, vol, 0, 0};
        }
        if (fits && sig->skip > 0) { --sig->skip; fits = false; }
        if (fits) {
            const bool kept = rows_io && counts_io && (reuse || M_io >= mdh_knn_rows_width(k)); // the rows live in the caller's buffers
            const int M = reuse ? M_io : (kept ? M_io : mdh_knn_rows_width(k));
            int *rows = kept ? rows_io : sc.alloc_n<int>((size_t)N * M), *rnn = kept ? counts_io : sc.alloc_n<int>((size_t)N), *nflag = sc.alloc_n<int>(4);
            double *rdist = reuse ? nullptr : sc.alloc_n<double>((size_t)N * M);
            Pos4 *w4 = sc.alloc_n<Pos4>((size_t)N);
            only = sc.alloc_n<unsigned char>((size_t)N);
            static int *pinned = nullptr; // the number of queries left for the cell walk: the one word this path reads back
            if (!pinned && hipHostMalloc(reinterpret_cast<void **>(&pinned), sizeof(int), hipHostMallocDefault) != hipSuccess) pinned = nullptr;
            if (sc.failed() || !pinned)
                return sc.failed() ? sc.error() : MDH_ERR_HIP;
            MDH_HIP(hipMemsetAsync(nflag, 0, sizeof(int), st));
            if (!reuse) {
                ProfRange pr("knn_rows_build", st);
                MDH_TRY(neighbor_rows_device(sc, dx, dy, dz, N, b, r, rows, rdist, rnn, M, nullptr, true));
            }
            if (kept && radius) *radius = r;
            ProfRange pr("knn_rows_select", st);
            const dim3 grid(grid_for(N, 256)), block(256);
            const double safe2 = r * r * (1.0 - 1e-9);
#define MDH_KNN_ROWS_AS(TRI, K, KEYED) hipLaunchKernelGGL((k_knn_rows<TRI, K, 32, KEYED>), grid, block, 0, st, w4, N, b, rows, M, rnn, safe2, k, dkey, di, dd, only, nflag)
#define MDH_KNN_ROWS(K)                                                                                                                  \
    do {                                                                                                                                  \
        if (b.tri) {                                                                                                                      \
            hipLaunchKernelGGL(k_knn_wrap4<true>, grid, block, 0, st, dx, dy, dz, N, b, w4);                                              \
            if (dkey) MDH_KNN_ROWS_AS(true, K, true); else MDH_KNN_ROWS_AS(true, K, false);                                              \
        } else {                                                                                                                          \
            hipLaunchKernelGGL(k_knn_wrap4<false>, grid, block, 0, st, dx, dy, dz, N, b, w4);                                             \
            if (dkey) MDH_KNN_ROWS_AS(false, K, true); else MDH_KNN_ROWS_AS(false, K, false);                                            \
        }                                                                                                                                 \
    } while (0)
            if (k <= 12) MDH_KNN_ROWS(12);
            else if (k <= 14) MDH_KNN_ROWS(14);
            else MDH_KNN_ROWS(18);
#undef MDH_KNN_ROWS
#undef MDH_KNN_ROWS_AS
            MDH_HIP(hipMemcpyAsync(pinned, nflag, sizeof(int), hipMemcpyDeviceToHost, st));
            MDH_HIP(hipStreamSynchronize(st));
            if (*pinned == 0)
                return sc.finish(space);
            if (reuse && (double)*pinned > 0.005 * (double)N) { // borrowed rows that do not reach: this k's own, then
                if (radius) *radius = 0.0;
                only = nullptr;
                continue;
            }
            sig->left_pct = (int)(100.0 * (double)*pinned / (double)N);
            if ((double)*pinned > 0.005 * (double)N) sig->skip = 15; // (a wavefront of the cell walk runs whole if ONE of its queries is left: 2 % left cost as much as all)
        }
        break;
      }
    }
    double *wx = sc.alloc_n<double>((size_t)N), *wy = sc.alloc_n<double>((size_t)N), *wz = sc.alloc_n<double>((size_t)N);
    if (sc.failed())
        return sc.error();
    if (b.tri)
        hipLaunchKernelGGL(k_knn_wrap<true>, dim3(grid_for(N, 256)), dim3(256), 0, st, dx, dy, dz, N, b, wx, wy, wz);
    else
        hipLaunchKernelGGL(k_knn_wrap<false>, dim3(grid_for(N, 256)), dim3(256), 0, st, dx, dy, dz, N, b, wx, wy, wz);

    // images per periodic axis (fast_knn.cpp:806-816)
    KnnGeom kg;
    int nim = 1;
    if (b.anypbc) {
        int64_t cl = N < 50 ? 50 : (N > 200 ? 200 : N);
        nim = (int)(200 / cl);
        if (nim < 1) nim = 1;
        if (nim < 2 && b.tri) nim = 2;
    }
    for (int d = 0; d < 3; ++d) kg.nim[d] = b.pbc[d] ? nim : 0;

    // grid: aim at ~k/3+1 atoms per cell so that ring 1 usually holds the k nearest
    DBox bg = b;
    if (b.tri) bg.o[0] = bg.o[1] = bg.o[2] = 0.0; // the triclinic wrap above is anchored at 0, not at the origin
    CellGrid cg;
    const double per_cell = (double)k / 3.0 + 1.0;
    double wtarget = std::cbrt(vol * per_cell / (double)N);
    if (!(wtarget > 0) || !std::isfinite(wtarget)) wtarget = 1.0;
    double tot = 1.0;
    kg.wmin = __builtin_huge_val();
    kg.rmax = 0;
    for (int d = 0; d < 3; ++d) {
        const double th = std::fabs(b.thick[d]);
        double f = std::floor(th / wtarget);
        int n = (f < 1.0 || !(f == f)) ? 1 : (f > 1024.0 ? 1024 : (int)f);
        cg.g.nc[d] = n;
        tot *= n;
    }
    // keep the grid below ~4 cells per atom (sparse / slab-like systems)
    while (tot > 4.0 * (double)N + 64.0) {
        int dmax = 0;
        for (int d = 1; d < 3; ++d) if (cg.g.nc[d] > cg.g.nc[dmax]) dmax = d;
        if (cg.g.nc[dmax] <= 1) break;
        tot /= cg.g.nc[dmax];
        cg.g.nc[dmax] = (cg.g.nc[dmax] + 1) / 2;
        tot *= cg.g.nc[dmax];
    }
    for (int d = 0; d < 3; ++d) {
        const double w = std::fabs(b.thick[d]) / cg.g.nc[d];
        if (w < kg.wmin) kg.wmin = w;
        const int r = b.pbc[d] ? (kg.nim[d] + 1) * cg.g.nc[d] : cg.g.nc[d] - 1;
        if (r > kg.rmax) kg.rmax = r;
    }
    cg.g.ncell = (int64_t)cg.g.nc[0] * cg.g.nc[1] * cg.g.nc[2];
    cg.g.rc_inv = 0.0;
    cg.g.mode = 1;
    MDH_TRY(build_cell_grid(sc, wx, wy, wz, N, bg, false, false, cg));
    int *label = nullptr, *unlabel = nullptr;
    if (key) {
        label = sc.alloc_n<int>((size_t)N);
        unlabel = sc.alloc_n<int>((size_t)N);
        if (sc.failed())
            return sc.error();
        MDH_HIP(hipMemsetAsync(unlabel, 0, sizeof(int) * (size_t)N, st));
        hipLaunchKernelGGL(k_knn_labels, dim3(grid_for(N, 256)), dim3(256), 0, st, cg.order, dkey, N, label, unlabel);
    }

    // the near kernel (list in registers) where it applies, then the general kernel on what it listed; larger k: the general
    // kernel for every query
    int *todo = nullptr;
    if (k <= 24 && kg.rmax >= 1 && (g_knn_variant == 0 || g_knn_variant == 3)) {
        todo = sc.alloc_n<int>((size_t)N + 1);
        // g_knn_variant 3: the counting kernel first, its leftovers to the insertion kernel (a measuring variant)
        int *failed = nullptr;
        const bool select = g_knn_variant == 3 && N >= 32768 && k >= 4; // (measured slower than the insertion kernel alone: a measuring variant, tools/knn_ab.py)
        if (select) failed = sc.alloc_n<int>((size_t)N + 1);
        if (sc.failed())
            return sc.error();
        MDH_HIP(hipMemsetAsync(todo, 0, sizeof(int), st));
        const dim3 grid(grid_for(N, 256)), block(256);
        const dim3 grid_sel(grid_for(N, 128)), block_sel(128);
        if (select) {
            MDH_HIP(hipMemsetAsync(failed, 0, sizeof(int), st));
            // trial radii: buckets of the squared distance.  Bucket 0 ends at the radius that holds k + 1 atoms in a uniform system of this
            // density; a step of the bucket number is 2^(0.1344) in d2 = 15 % in volume (float bits: 2^23 per octave)
            KnnTrial trial;
            const double r0 = std::cbrt(3.0 * (double)(k + 1) / (4.0 * 3.14159265358979323846 * ((double)N / vol)));
            const float t0 = (float)(r0 * r0);
            int t0_bits;
            std::memcpy(&t0_bits, &t0, sizeof(int));
            const double step = 0.1344 * 8388608.0; // float-bit units per bucket
            trial.base = t0_bits - (int)step;       // bits below t0: bucket 0
            trial.scale = (unsigned)(4294967296.0 / step);
            const double one_cell = kg.wmin * (1.0 - 1e-9);
            trial.bound_f = std::nextafterf((float)(one_cell * one_cell), 0.0f); // (below the double bound: "inside one cell width" for sure)
#define MDH_KNN_SELECT(KC)                                                                                                                \
    do {                                                                                                                                  \
        if (b.tri) hipLaunchKernelGGL((k_knn_select<true, KC>), grid_sel, block_sel, 0, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, N, b, bg, cg.g, kg, k, trial, di, dd, failed, label, unlabel); \
        else hipLaunchKernelGGL((k_knn_select<false, KC>), grid_sel, block_sel, 0, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, N, b, bg, cg.g, kg, k, trial, di, dd, failed, label, unlabel); \
    } while (0)
            if (k <= 12) MDH_KNN_SELECT(16);
            else if (k <= 14) MDH_KNN_SELECT(20);
            else if (k <= 18) MDH_KNN_SELECT(24);
            else MDH_KNN_SELECT(32);
#undef MDH_KNN_SELECT
        }
        const dim3 grid_near(grid.x); // (behind the counting kernel it walks a list, usually short: the other workgroups leave at once)
#define MDH_KNN_NEAR(K)                                                                                                                   \
    do {                                                                                                                                  \
        if (b.tri) hipLaunchKernelGGL((k_knn_near<true, K>), grid_near, block, 0, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, N, b, bg, cg.g, kg, k, di, dd, todo, label, unlabel, failed, only); \
        else hipLaunchKernelGGL((k_knn_near<false, K>), grid_near, block, 0, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, N, b, bg, cg.g, kg, k, di, dd, todo, label, unlabel, failed, only); \
    } while (0)
        if (k <= 12) MDH_KNN_NEAR(12);
        else if (k <= 14) MDH_KNN_NEAR(14);
        else if (k <= 18) MDH_KNN_NEAR(18);
        else MDH_KNN_NEAR(24);
#undef MDH_KNN_NEAR
    }
    int bd = 256;
    while (bd > 64 && (size_t)bd * k * 12 > 65536) bd -= 64;
    const size_t lds = (size_t)bd * k * 12;
    if (b.tri)
        hipLaunchKernelGGL(k_knn<true>, dim3(grid_for(N, bd)), dim3(bd), lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, N, b, bg, cg.g, kg, k, di, dd, todo, label, unlabel);
    else
        hipLaunchKernelGGL(k_knn<false>, dim3(grid_for(N, bd)), dim3(bd), lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, N, b, bg, cg.g, kg, k, di, dd, todo, label, unlabel);
    MDH_HIP(hipGetLastError());
    return sc.finish(space);
}

MDH_WARM_UNIT(knn)
