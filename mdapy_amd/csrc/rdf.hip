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
#include <cstdlib>
#include "grid.hpp"
#include <algorithm>

namespace mdh {

static int g_rdf_variant = 0; // test hook: 1 = thread-per-atom kernel everywhere
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

// ---- streaming, cell-list path on LDS tiles (orthogonal boxes, and triclinic ones periodic along all three vectors).  A workgroup takes one centre cell at a time and pairs its
// atoms with those of the cell itself and of the 13 cells "ahead" of it in the walk order — every unordered pair of atoms
// in neighbouring cells is met exactly once and counted in both directions, (ti, tj) and (tj, ti), which is what the
// reference's full 27-cell walk over ordered pairs adds up to (:223-251), at half the distance evaluations.  The atoms are
// staged into LDS once per cell (coalesced reads of the cell-sorted arrays) as single-precision coordinates relative to the
// centre cell's corner, periodic image folded in; a lane keeps one candidate in registers and walks the centre atoms
// (a broadcast LDS read), so the lanes are busy whatever the cell population.  Single precision only sorts the easy pairs
// into their shells: a pair whose r^2 lies within `tol` of a shell boundary (or of rc^2) is listed and binned after the scan
// with the reference's own double-precision expression, once per direction (raw x[j] - wrapped x[i], minimum image,
// sqrt(r2)/dr, :236-251 — the two directions may round differently), so the counts are the reference's bit for bit.
// |r2_f32 - r2| <= 1.8e-6 (rc^2 + r2) for coordinates inside the 3x3x3-cell frame (cell width < 1.34 rc): three roundings
// to f32 of magnitudes <= 2.7 rc, the subtractions, the FMA chain.
constexpr int RDF_NB = 14, RDF_HITS = 128; // RDF_HITS: a wave's list of hits (< 64 left over + 64 new ones)
__host__ __device__ inline int64_t hsize_of(int ntype, int nbin) { return (int64_t)ntype * ntype * nbin; }

template <bool TRI>
__device__ __forceinline__ void rdf_exact_pair(const double *__restrict__ xs, const double *__restrict__ ys, const double *__restrict__ zs,
                                               const DBox &b, int qi, int qj, int ti, int tj, int ntype, int nbin, double dr, double rcsq,
                                               unsigned *lds)
{
    double xi = xs[qi], yi = ys[qi], zi = zs[qi];
    if (b.anypbc) wrap<TRI>(b, xi, yi, zi);        // :213-214
    double ex = xs[qj] - xi, ey = ys[qj] - yi, ez = zs[qj] - zi;
    pbc<TRI>(b, ex, ey, ez);
    const double e2 = ex * ex + ey * ey + ez * ez;
    if (e2 < rcsq) {                               // strict, :246
        const int kk = (int)(sqrt(e2) / dr);
        if (kk < nbin) atomicAdd(&lds[(ti * ntype + tj) * nbin + kk], 1u);
    }
}

// TRI: fully periodic triclinic boxes.  The same walk over the cells of the fractional grid; a cell's corner and the shift
// to a neighbouring cell are sums of the cell's edge VECTORS (box vector d over nc[d]), the atoms are wrapped through the
// fractional coordinates, and the band is widened by tol_scale = (extent of the 3x3x3-cell frame along the worst Cartesian
// axis) / (2.7 rc): the single-precision error of a coordinate grows with the frame, which a sheared cell stretches.
// One WAVEFRONT per centre cell (four cells in flight per workgroup, sixteen per CU): a cell's set-up — its 14 cell ranges, the
// centre atoms, the first candidates: three dependent trips to memory — is a few microseconds for ~170 wave-trips of pair
// tests, and with a whole workgroup per cell (four barriers per cell, four cells in flight per CU) 69 % of the wave-cycles were
// parked.  A wave needs no workgroup barrier for its own cell; only the histogram is shared.
struct RdfHit { float r2; unsigned qi, qj, tt; }; // tt = type of the centre | type of the candidate << 8
struct RdfWave { // LDS of one wave
    float4 cen[64];          // centre atoms: ux, uy, uz, bits of the position in the sorted arrays
    RdfHit hitq[RDF_HITS];   // pairs inside the cutoff waiting to be binned
    unsigned xq[2 * 128];    // sorted positions of the pairs to bin exactly
    double lo[RDF_NB][3];    // corner of each of the 14 cells
    float shift[RDF_NB][3];  // its offset from the centre cell
    int start[RDF_NB + 2], src[RDF_NB];
    unsigned nq, pad;
    unsigned char etype[64];
    float4 cbuf[128];        // candidates that can reach a centre, waiting for a full wave of them: ux, uy, uz, bits of the position
    unsigned cinfo[128];     // their type | (1 << 8 when they belong to the centre cell itself)
};
static_assert(sizeof(RdfWave) % 16 == 0, "RdfWave keeps float4 alignment");

template <bool TRI>
__global__ __launch_bounds__(256) void k_rdf_tile(const double *__restrict__ xs, const double *__restrict__ ys, const double *__restrict__ zs,
                                                  const int *__restrict__ order, const int *__restrict__ cell_start,
                                                  const int *__restrict__ type, DBox b, Grid g, double rc, int nbin, int ntype,
                                                  unsigned long long *__restrict__ hist, float tol_scale, int probe)
{
    // probe (a measuring switch, MDH_RDF_PROBE=1, tools/rdf_probe.py): 1 = no pair is ever "inside the cutoff" — the kernel walks
    // and tests every pair as always and bins nothing: what the candidate walk and the distance tests cost by themselves
    // 2 = no candidate is dropped by the bounding-box test below (what the cull is worth; same counts)
    extern __shared__ __attribute__((aligned(16))) unsigned char rdf_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t hsize = hsize_of(ntype, nbin);
    unsigned *lds = reinterpret_cast<unsigned *>(rdf_lds); // [ntype^2 nbin] the workgroup's histogram
    RdfWave &W = reinterpret_cast<RdfWave *>(rdf_lds + (((size_t)hsize * 4 + 15) & ~(size_t)15))[wv];
    for (int64_t q = tid; q < hsize; q += 256) lds[q] = 0u;
    __syncthreads();
    const double dr = rc / nbin, rcsq = rc * rc; // :158-159
    const float drf = (float)dr, inv_dr = (float)(1.0 / dr), rc2f = (float)rcsq;
    // edge vectors of a cell: ev[d][c] = component c of box vector d over nc[d] (orthogonal box: the diagonal only)
    double ev[3][3];
    for (int d = 0; d < 3; ++d)
        for (int c = 0; c < 3; ++c) ev[d][c] = (TRI || d == c) ? b.h[3 * d + c] / g.nc[d] : 0.0;
    auto wsync = [] { // LDS written by some lanes of this wave is read by others
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // Pairs inside the cutoff are ~15 % of the pairs tested; binning one (square root, shell bounds, band test, LDS atomics) is
    // ~40 instructions.  Taken on the spot they ran for every trip of the centre loop with a seventh of the lanes active (some
    // lane of the 64 nearly always has a hit): 55 instructions per 64 pair tests.  Instead a hit is pushed — ballot, mbcnt, one
    // 16-byte LDS store — onto the wave's list, and the list is binned 64 hits at a time with every lane busy.
    int nhit = 0; // hits on the wave's list (the same number in every lane)
    auto bin_hit = [&](const RdfHit h) { // one pair inside the cutoff, both directions
        const float r2 = h.r2;
        const int qi = (int)h.qi, qj = (int)h.qj, ti = (int)(h.tt & 255u), tj = (int)(h.tt >> 8);
        const int kb = (int)(__builtin_sqrtf(r2) * inv_dr);
        const float lo = (float)kb * drf, hi = lo + drf;
        const float tol = 4.0e-6f * tol_scale * (rc2f + r2);
        if (r2 - lo * lo > tol && hi * hi - r2 > tol) {
            if (kb < nbin) {
                if (ti == tj) {
                    atomicAdd(&lds[(ti * ntype + tj) * nbin + kb], 2u);
                } else {
                    atomicAdd(&lds[(ti * ntype + tj) * nbin + kb], 1u);
                    atomicAdd(&lds[(tj * ntype + ti) * nbin + kb], 1u);
                }
            }
        } else {
            const unsigned slot = atomicAdd(&W.nq, 1u);
            if (slot < 128u) {
                W.xq[2 * slot] = (unsigned)qi;
                W.xq[2 * slot + 1] = (unsigned)qj;
            } else { // (a full list: this pair right away)
                rdf_exact_pair<TRI>(xs, ys, zs, b, qi, qj, ti, tj, ntype, nbin, dr, rcsq, lds);
                rdf_exact_pair<TRI>(xs, ys, zs, b, qj, qi, tj, ti, ntype, nbin, dr, rcsq, lds);
            }
        }
    };
    for (int64_t cell = (int64_t)blockIdx.x * 4 + wv; cell < g.ncell; cell += (int64_t)gridDim.x * 4) {
        const int c2 = (int)(cell % g.nc[2]), c1 = (int)((cell / g.nc[2]) % g.nc[1]), c0 = (int)(cell / ((int64_t)g.nc[1] * g.nc[2]));
        const int cs = cell_start[cell], ncen_all = cell_start[cell + 1] - cs;
        if (ncen_all == 0)
            continue;
        wsync(); // the tables below are reused
        int n = 0;
        if (lane < RDF_NB) { // the cell itself (entry 0) and the 13 cells after it in the reference's walk order (:223-235)
            const int o = lane == 0 ? 13 : 13 + lane; // position in the 27-cell walk
            const int da = o / 9 - 1, db = (o / 3) % 3 - 1, dc = o % 3 - 1;
            const int a = b.pbc[0] ? pmod(c0 + da, g.nc[0]) : c0 + da, bb = b.pbc[1] ? pmod(c1 + db, g.nc[1]) : c1 + db,
                      cc = b.pbc[2] ? pmod(c2 + dc, g.nc[2]) : c2 + dc;
            int src = 0;
            if (a >= 0 && a < g.nc[0] && bb >= 0 && bb < g.nc[1] && cc >= 0 && cc < g.nc[2]) { // open axes are not wrapped
                const int64_t nb = ((int64_t)a * g.nc[1] + bb) * g.nc[2] + cc;
                src = cell_start[nb];
                n = cell_start[nb + 1] - src;
            }
            W.src[lane] = src;
            for (int c = 0; c < 3; ++c) {
                W.lo[lane][c] = b.o[c] + a * ev[0][c] + bb * ev[1][c] + cc * ev[2][c];
                W.shift[lane][c] = (float)(da * ev[0][c] + db * ev[1][c] + dc * ev[2][c]);
            }
        }
        { // prefix of the 14 populations across the lanes
            int inc = n;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const int t = __shfl_up(inc, d, 64);
                if (lane >= d) inc += t;
            }
            if (lane < RDF_NB) W.start[lane + 1] = inc;
            if (lane == 0) { W.start[0] = 0; W.nq = 0; }
        }
        wsync();
        const int ncand_all = W.start[RDF_NB];
        for (int cbase = 0; cbase < ncen_all; cbase += 64) {
            const int ncen = min(64, ncen_all - cbase);
            wsync();
            if (lane < ncen) { // centre atoms
                const int q = cs + cbase + lane;
                double xi = xs[q], yi = ys[q], zi = zs[q];
                if (b.anypbc)
                    wrap<TRI>(b, xi, yi, zi);
                W.cen[lane] = make_float4((float)(xi - W.lo[0][0]), (float)(yi - W.lo[0][1]), (float)(zi - W.lo[0][2]), __int_as_float(q));
                W.etype[lane] = (unsigned char)type[order[q]];
            }
            wsync();
            // The centres' bounding box (in the centre cell's frame; an atom clamped into an edge cell from outside an open box may lie
            // outside the cell): a candidate farther than the cutoff from the BOX reaches no centre — 24 % of the atoms of the 14
            // cells of rc-wide cells (the cells' corners and edges beyond the rounded box of volume 20.6 rc^3 of 27).  Its distance to
            // the box is worked out per axis with the subtraction of the pair test, squared and summed as the pair test does: rounding
            // is monotone, so the bound is never above the r2 the pair test would find and no pair is lost.  The candidates that
            // pass are gathered into full waves (ballot, mbcnt, LDS), and the centre loop runs once per FULL wave of them.
            float blo[3], bhi[3];
            {
                const float4 ce = W.cen[lane < ncen ? lane : 0];
                blo[0] = bhi[0] = ce.x; blo[1] = bhi[1] = ce.y; blo[2] = bhi[2] = ce.z;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        blo[c] = fminf(blo[c], __shfl_xor(blo[c], d, 64));
                        bhi[c] = fmaxf(bhi[c], __shfl_xor(bhi[c], d, 64));
                    }
            }
            const float reach2 = rc2f * (1.0f + 1.0e-4f * tol_scale);
            int nbuf = 0; // candidates waiting in W.cbuf (the same number in every lane)
            auto pair_up = [&](int nact) { // the first nact (<= 64) waiting candidates against every centre of the chunk
                const bool valid = lane < nact;
                const float4 cb = W.cbuf[lane];
                const unsigned info = W.cinfo[lane];
                const float ux = cb.x, uy = cb.y, uz = cb.z;
                const int qj = __float_as_int(cb.w);
                const unsigned tj = info & 255u;
                const bool same_cell = (info >> 8) != 0u;
                const int ncen_u = __builtin_amdgcn_readfirstlane(ncen); // (the same in every lane: a scalar trip count, no exec-mask loop)
                auto one = [&](const float4 ce, int c) {
                    const float dx = ux - ce.x, dy = uy - ce.y, dz = uz - ce.z;
                    const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    const int qi = __float_as_int(ce.w);
                    // pairs inside the cell once: candidate after centre (and never the atom itself, :240)
                    const bool hit = valid && r2 < reach2 && !(same_cell && qj <= qi) && probe != 1;
                    const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
                    if (hm == 0)
                        return;
                    if (hit) {
                        const int at = nhit + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
                        W.hitq[at] = RdfHit{r2, (unsigned)qi, (unsigned)qj, (unsigned)W.etype[c] | (tj << 8)};
                    }
                    nhit = __builtin_amdgcn_readfirstlane(nhit + (int)__popcll(hm)); // (kept in a scalar register: the test below is a scalar branch)
                    if (nhit >= 64) { // a full wave of hits: bin them, move the rest down
                        wsync();
                        bin_hit(W.hitq[lane]);
                        const int rest = nhit - 64;
                        RdfHit mv = W.hitq[lane];
                        if (lane < rest) mv = W.hitq[64 + lane];
                        wsync();
                        if (lane < rest) W.hitq[lane] = mv;
                        nhit = rest;
                    }
                };
                int c = 0;
                for (; c + 1 < ncen_u; c += 2) { // two centres a trip: both broadcast reads in flight before the first test (four: 5.0 -> 6.4 ms)
                    const float4 ce0 = W.cen[c], ce1 = W.cen[c + 1];
                    one(ce0, c);
                    one(ce1, c + 1);
                }
                if (c < ncen_u) one(W.cen[c], c);
            };
            for (int gbase = 0; gbase < ncand_all; gbase += 64) {
                const int gv = gbase + lane; // this lane's candidate: atom gv of the 14 cells laid end to end
                const bool valid = gv < ncand_all;
                int k = 0;
                if (valid)
                    while (gv >= W.start[k + 1]) ++k;
                const int qj = valid ? W.src[k] + (gv - W.start[k]) : W.src[0];
                double xj = xs[qj], yj = ys[qj], zj = zs[qj];
                if (b.anypbc)
                    wrap<TRI>(b, xj, yj, zj);
                const float ux = (float)(xj - W.lo[k][0]) + W.shift[k][0], uy = (float)(yj - W.lo[k][1]) + W.shift[k][1],
                            uz = (float)(zj - W.lo[k][2]) + W.shift[k][2];
                const unsigned tj = (unsigned)type[order[qj]];
                // distance to the centres' box, axis by axis as the pair test subtracts (candidate minus centre): never above a pair's
                const float ex = fmaxf(fmaxf(ux - bhi[0], blo[0] - ux), 0.0f), ey = fmaxf(fmaxf(uy - bhi[1], blo[1] - uy), 0.0f),
                            ez = fmaxf(fmaxf(uz - bhi[2], blo[2] - uz), 0.0f);
                const float box2 = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                const bool keep = valid && (box2 < reach2 || probe == 2); // (probe 2, a measuring switch: nobody is left out)
                const unsigned long long km = __ballot(keep);
                if (keep) {
                    const int at = nbuf + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
                    W.cbuf[at] = make_float4(ux, uy, uz, __int_as_float(qj));
                    W.cinfo[at] = tj | (k == 0 ? 256u : 0u);
                }
                nbuf += __popcll(km);
                if (nbuf >= 64) { // a full wave of candidates
                    wsync();
                    pair_up(64);
                    const int rest = nbuf - 64;
                    float4 mv = W.cbuf[lane];
                    unsigned mi = W.cinfo[lane];
                    if (lane < rest) { mv = W.cbuf[64 + lane]; mi = W.cinfo[64 + lane]; }
                    wsync();
                    if (lane < rest) { W.cbuf[lane] = mv; W.cinfo[lane] = mi; }
                    nbuf = rest;
                }
            }
            wsync();
            if (nbuf > 0) pair_up(nbuf); // what is left
        }
        wsync(); // what is left on the wave's list
        if (lane < nhit) bin_hit(W.hitq[lane]);
        nhit = 0;
        wsync();
        const int nq = (int)min(W.nq, 128u);
        for (int e = lane; e < 2 * nq; e += 64) { // the pairs near a shell boundary, each direction as the reference bins it
            const int qa = (int)W.xq[2 * (e >> 1) + (e & 1)], qb = (int)W.xq[2 * (e >> 1) + 1 - (e & 1)];
            rdf_exact_pair<TRI>(xs, ys, zs, b, qa, qb, type[order[qa]], type[order[qb]], ntype, nbin, dr, rcsq, lds);
        }
    }
    __syncthreads();
    for (int64_t q = tid; q < hsize; q += 256) {
        const unsigned v = lds[q];
        if (v) atomicAdd(&hist[q], (unsigned long long)v);
    }
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
// The counts do not depend on the order the entries are visited in, so the list is read as what it is in memory — one array
// of N * M entries: a workgroup takes 256 consecutive rows and its lanes walk that piece entry by entry (entry e belongs to
// row e / M), every load of the wave one contiguous 256- or 512-byte piece.  A lane per row read its row 4 / 8 bytes at a time,
// M entries away from its neighbour's lane: 6.1 ms for the 42-of-50-wide list of 4 M atoms at rc 5 (the list was fetched many
// times over: common.hpp stage_row_chunk).
constexpr int RDF_LIST_ROWS = 256;
__global__ __launch_bounds__(256) void k_rdf_list(const int *__restrict__ verlet, const double *__restrict__ dist,
                                                  const int *__restrict__ nn, const int *__restrict__ type,
                                                  int64_t N, int64_t M, double rc, int nbin, int ntype, int single,
                                                  unsigned long long *__restrict__ hist)
{
    __shared__ unsigned lds[RDF_LDS_BINS];
    __shared__ int row_n[RDF_LIST_ROWS], row_t[RDF_LIST_ROWS];
    const int64_t hsize = single ? nbin : (int64_t)ntype * ntype * nbin;
    const bool use_lds = hsize <= RDF_LDS_BINS;
    if (use_lds)
        for (int64_t q = threadIdx.x; q < hsize; q += blockDim.x) lds[q] = 0u;
    const int64_t row0 = (int64_t)blockIdx.x * RDF_LIST_ROWS;
    const int rows = (int)((N - row0) < RDF_LIST_ROWS ? (N - row0) : RDF_LIST_ROWS);
    if ((int)threadIdx.x < rows) {
        row_n[threadIdx.x] = nn[row0 + threadIdx.x];
        row_t[threadIdx.x] = single ? 0 : type[row0 + threadIdx.x];
    }
    __syncthreads();
    const double dr = rc / nbin;
    const unsigned m = (unsigned)M, total = (unsigned)rows * m; // (rows * M < 2^31: mdh_rdf_list refuses wider rows)
    const int *__restrict__ gv = verlet + row0 * M;
    const double *__restrict__ gd = dist + row0 * M;
    for (unsigned e = threadIdx.x; e < total; e += 256) {
        const unsigned r = e / m, q = e - r * m;
        if ((int)q >= row_n[r])
            continue;
        const double d = gd[e];
        const int k = (int)(d / dr);
        if (!(d < rc) || k >= nbin || k < 0) // k == nbin can only arise from rounding at d -> rc (the reference would write out of bounds there)
            continue;
        const int64_t i = row0 + r;
        const int j = safe_id(gv[e], i, N);
        if (single) {
            if (j > i) hist_add(lds, hist, use_lds, k);
        } else {
            hist_add(lds, hist, use_lds, ((int64_t)row_t[r] * ntype + type[j]) * nbin + k);
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

int mdh_debug_set_rdf_variant(int v)
{
    g_rdf_variant = v;
    return MDH_OK;
}

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
        const size_t tile_lds = (((size_t)hsize * 4 + 15) & ~(size_t)15) + 4 * sizeof(RdfWave);
        // the frame of 3x3x3 cells along the worst Cartesian axis, in units of the orthogonal kernel's 2.7 rc
        double tol_scale = 1.0;
        const bool tri_tile = b.tri != 0; // (periodic or open along any vector: the kernel wraps cell indices and atoms along the periodic ones only)
        if (tri_tile) {
            for (int c = 0; c < 3; ++c) {
                double ext = 0;
                for (int d = 0; d < 3; ++d) ext += 2.0 * std::fabs(b.h[3 * d + c]) / cg.g.nc[d];
                tol_scale = std::max(tol_scale, ext / (2.7 * rc));
            }
        }
        if ((!b.tri || (tri_tile && tol_scale < 64.0)) && ntype <= 255 && hsize <= RDF_LDS_BINS && g_rdf_variant == 0) {
            ProfRange pr("k_rdf_tile", st);
            const char *probe_env = std::getenv("MDH_RDF_PROBE");
            const int rdf_probe = probe_env ? std::atoi(probe_env) : 0;
            const unsigned blocks = (unsigned)std::min<int64_t>((cg.g.ncell + 3) / 4, 256 * 8); // a wave per cell, four to a workgroup
            if (b.tri)
                hipLaunchKernelGGL(k_rdf_tile<true>, dim3(blocks), dim3(256), tile_lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, dt, b, cg.g, rc, nbin, ntype, hist, (float)tol_scale, rdf_probe);
            else
                hipLaunchKernelGGL(k_rdf_tile<false>, dim3(blocks), dim3(256), tile_lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, dt, b, cg.g, rc, nbin, ntype, hist, 1.0f, rdf_probe);
        } else if (b.tri)
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
    if (N < 0 || M <= 0 || M >= (1 << 23) || nbin <= 0 || ntype <= 0) { set_error("mdh_rdf: invalid argument"); return MDH_ERR_ARG; }
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

MDH_WARM_UNIT(rdf)
