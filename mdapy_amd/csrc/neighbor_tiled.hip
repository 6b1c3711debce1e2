// neighbor_tiled.hip — LDS-tiled 27-cell neighbor scan for orthogonal boxes (gfx950).
//
// Same result, bit for bit, as the thread-per-atom kernel in neighbor.hip (and therefore as
// src/neighbor.cpp:102-187 of the reference); this is the fast path for the common case.
//
// One workgroup owns a tile of T x T x T cells.  It stages the atoms of the (T+2)^3 halo cells
// (raw x,y,z + atom id) from the cell-sorted arrays into LDS with coalesced loads — every cell of
// the grid is read from HBM/L2 once per neighbouring tile instead of once per neighbouring ATOM —
// then each thread takes one centre atom of the tile and walks its 27 cells out of LDS in the
// reference's order (cells (i,j,k)-lexicographic, atoms of a cell by descending id).  The three
// z-cells of one (i,j) column are contiguous in LDS, so the walk is 9 runs of candidates, each
// processed four at a time (independent f64 chains, predicated; no per-candidate branch).
//
// Minimum image.  With all atoms handed over inside the box and >= 7 cells on every periodic axis
// the image number n = floor(d/L + 0.5) of a (centre, candidate) pair is decided by the pair of
// CELLS: candidates of an adjacent cell are < 3 rc <= 3L/7 away after the right shift, so n is 0
// inside the box and +-1 across the periodic seam — far from the +-L/2 decision points.  The kernel
// then evaluates the reference's  d - L*n  with that n (same operands, same two roundings), which
// removes 3 divisions / 3 floors per candidate.  When the precondition does not hold (flag from the
// binning pass, or < 7 cells) the exact threshold search of common.hpp::pbc_axis is used instead.
//
// Output.  A hit is first recorded in LDS as a 2-byte ticket (LDS index of the candidate + which of
// the run's three z-cells it sits in).  After the scan the workgroup turns tickets into rows
// cooperatively: consecutive lanes write consecutive slots of a row, so a wave store covers whole
// 64 B / 128 B row segments instead of 64 scattered rows; the distance is recomputed from the same
// operands (identical bits).  With fill_pads the same pass writes the -1 / rc+1 pads.
#include "common.hpp"
#include "grid.hpp"

namespace mdh {

static constexpr int HALO_CAP = 1024; // atoms a tile's halo may hold in LDS (28 B each); ticket index is 10 bits

static constexpr int NT = 256;       // threads per workgroup
static constexpr int MAX_NH = 512;   // halo cells a tile may have
static constexpr int MAX_COLS = 64;  // (x,y) columns of centre cells a tile may have

// Tile shape (cells): TXY x TXY x TZ, chosen on the host from the mean cell population so that the halo fills
// (but does not overflow) the LDS budget and the number of centre atoms is close to a multiple of NT.
struct TileShape { int txy, tz; };

__device__ __forceinline__ int excl_scan_block(int v, int *scratch, int nthreads, int *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) scratch[w] = inc;
    __syncthreads();
    int off = 0, tot = 0;
    for (int k = 0; k < (nthreads >> 6); ++k) {
        if (k < w) off += scratch[k];
        tot += scratch[k];
    }
    __syncthreads();
    *total = tot;
    return off + inc - v;
}

// d2 of one (centre, candidate) pair; CELLSHIFT: sx,sy,sz = L*n for the pair's cells
template <bool CELLSHIFT>
__device__ __forceinline__ double pair_d2_tiled(const DBox &b, double xj, double yj, double zj, double xi, double yi,
                                                double zi, double sx, double sy, double sz)
{
    double dx = xj - xi, dy = yj - yi, dz = zj - zi; // raw x[j] - wrapped centre (neighbor.cpp:164-166)
    if (CELLSHIFT) {
        dx = dx - sx; // == xij - L*floor(xij/L+0.5)   (box.h:120-124) with n known from the cells
        dy = dy - sy;
        dz = dz - sz;
    } else {
        pbc<false>(b, dx, dy, dz);
    }
    return dx * dx + dy * dy + dz * dz;
}

// L * n for the 2-bit image code (n+1)
__device__ __forceinline__ double img_shift(double L, int code) { return L * (double)(code - 1); }

template <bool CELLSHIFT, int MODE>
__global__ __launch_bounds__(NT) void k_neighbor_tiled(
    const double *__restrict__ xs, const double *__restrict__ ys, const double *__restrict__ zs,
    const int *__restrict__ order, const int *__restrict__ cell_start, DBox b, Grid g, double rc,
    int *__restrict__ verlet, double *__restrict__ dist, int *__restrict__ nn, int M, int mp_shift,
    int *__restrict__ flags, unsigned char *__restrict__ tile_flag, int nt0, int nt1, int nt2, int want_moved,
    TileShape ts)
{
    const int TXY = ts.txy, TZ = ts.tz;
    const int HXY = TXY + 2, HZ = TZ + 2, NH = HXY * HXY * HZ, NCOL = TXY * TXY;
    // which of the two minimum-image variants serves this call is decided on the device (no host sync)
    // want_moved: 0 = run only if every atom came in wrapped, 1 = run only if not, -1 = always run
    if (want_moved >= 0 && (flags[0] != 0) != (want_moved != 0))
        return;

    extern __shared__ unsigned char smem[];
    double *lx = reinterpret_cast<double *>(smem);
    double *ly = lx + HALO_CAP;
    double *lz = ly + HALO_CAP;
    double *cxi = lz + HALO_CAP; // wrapped centre coordinates of this pass [NT]
    double *cyi = cxi + NT;
    double *czi = cyi + NT;
    int *lid = reinterpret_cast<int *>(czi + NT); // [HALO_CAP]
    int *crow = lid + HALO_CAP;                   // global atom id of the centre [NT]
    int *cinfo = crow + NT;                       // min(count, M) | xy image code << 8 | (z codes) << 12  [NT]
    unsigned short *tick = reinterpret_cast<unsigned short *>(cinfo + NT); // [NT][M]
    __shared__ int h_off[MAX_NH + 1];
    __shared__ int h_img[MAX_NH]; // (nx+1) | (ny+1)<<2 | (nz+1)<<4
    __shared__ int c_off[MAX_COLS + 1];
    __shared__ int scan_tmp[4];

    // XCD-aware tile order: block b runs on XCD b%8; give every XCD one contiguous chunk of tiles so that
    // neighbouring tiles (which share halo cells) meet in the same L2.
    const int ntiles = nt0 * nt1 * nt2;
    const int per = (ntiles + 7) / 8;
    const int tile_id = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile_id >= ntiles)
        return;
    const int t2 = tile_id % nt2, t1 = (tile_id / nt2) % nt1, t0 = tile_id / (nt2 * nt1);
    const int T0 = t0 * TXY, T1 = t1 * TXY, T2 = t2 * TZ;
    const int tid = threadIdx.x;

    // ---- halo cell table: source range, LDS offset, image code.  Thread t owns halo cells 2t and 2t+1
    // (adjacent in z, hence adjacent in memory).
    int cnt2[2] = {0, 0}, src2[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int h = 2 * tid + u;
        if (h >= NH)
            continue;
        int cnt = 0, src = 0;
        const int hz = h % HZ, hy = (h / HZ) % HXY, hx = h / (HZ * HXY);
        const int g0 = T0 + hx - 1, g1 = T1 + hy - 1, g2 = T2 + hz - 1;
        int img = 1 | (1 << 2) | (1 << 4);
        if (g0 >= -1 && g0 <= g.nc[0] && g1 >= -1 && g1 <= g.nc[1] && g2 >= -1 && g2 <= g.nc[2]) {
            const int a0 = pmod(g0, g.nc[0]), a1 = pmod(g1, g.nc[1]), a2 = pmod(g2, g.nc[2]);
            const int64_t c = ((int64_t)a0 * g.nc[1] + a1) * g.nc[2] + a2;
            src = cell_start[c];
            cnt = cell_start[c + 1] - src;
            // image of the candidate cell seen from an in-grid centre cell: below the box -> raw coordinates are
            // ~+L away (n = +1); above -> n = -1.  Open axes are never folded (box.h:120-124).
            const int n0 = b.pbc[0] ? (g0 < 0 ? 1 : (g0 >= g.nc[0] ? -1 : 0)) : 0;
            const int n1 = b.pbc[1] ? (g1 < 0 ? 1 : (g1 >= g.nc[1] ? -1 : 0)) : 0;
            const int n2 = b.pbc[2] ? (g2 < 0 ? 1 : (g2 >= g.nc[2] ? -1 : 0)) : 0;
            img = (n0 + 1) | ((n1 + 1) << 2) | ((n2 + 1) << 4);
        }
        h_img[h] = img;
        cnt2[u] = cnt;
        src2[u] = src;
    }
    int total;
    const int off0 = excl_scan_block(cnt2[0] + cnt2[1], scan_tmp, NT, &total);
    if (2 * tid < NH) h_off[2 * tid] = off0;
    if (2 * tid + 1 < NH) h_off[2 * tid + 1] = off0 + cnt2[0];
    if (tid == 0) h_off[NH] = total;
    if (total > HALO_CAP) { // leave this tile to the thread-per-atom kernel
        if (tid == 0) {
            tile_flag[tile_id] = 1;
            atomicAdd(&flags[2], 1);
        }
        return;
    }
    // ---- stage the halo atoms (each thread copies its cell: neighbouring threads read neighbouring memory)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int cnt = cnt2[u], src = src2[u], off = off0 + (u ? cnt2[0] : 0);
        int k = 0;
        for (; k + 4 <= cnt; k += 4) { // four independent loads in flight per array
            double a[4], bb[4], c[4];
            int d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = xs[src + k + u]; bb[u] = ys[src + k + u]; c[u] = zs[src + k + u]; d[u] = order[src + k + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { lx[off + k + u] = a[u]; ly[off + k + u] = bb[u]; lz[off + k + u] = c[u]; lid[off + k + u] = d[u]; }
        }
        for (; k < cnt; ++k) {
            lx[off + k] = xs[src + k];
            ly[off + k] = ys[src + k];
            lz[off + k] = zs[src + k];
            lid[off + k] = order[src + k];
        }
    }
    // ---- centre runs: one contiguous LDS run per (x,y) column of the tile, clipped to the grid
    const int zlo = 1, zhi = min(TZ, g.nc[2] - T2); // interior hz in [1, zhi]
    __syncthreads();                                  // h_off complete
    if (tid < 64) { // one wave: per-column centre counts -> exclusive prefix
        int v = 0;
        if (tid < NCOL) {
            const int hx = tid / TXY + 1, hy = tid % TXY + 1;
            const bool ok = (T0 + hx - 1 < g.nc[0]) && (T1 + hy - 1 < g.nc[1]) && zhi >= 1;
            v = ok ? (h_off[(hx * HXY + hy) * HZ + zhi + 1] - h_off[(hx * HXY + hy) * HZ + zlo]) : 0;
        }
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int t = __shfl_up(inc, d, 64);
            if (tid >= d) inc += t;
        }
        if (tid < NCOL) c_off[tid + 1] = inc;
        if (tid == 0) c_off[0] = 0;
    }
    __syncthreads();
    const int ncentres = c_off[NCOL];
    const double rcsq = rc * rc; // neighbor.cpp:127
    const double pad = rc + 1.0;
    const int MP = 1 << mp_shift; // smallest power of two >= M: slots of a row handled by MP adjacent lanes

    for (int base = 0; base < ncentres; base += NT) {
        const int q = base + tid;
        if (q < ncentres) {
            int col = 0;
            for (int k = 1; k < NCOL; ++k)
                col += (q >= c_off[k]) ? 1 : 0;
            const int hx = col / TXY + 1, hy = col % TXY + 1;
            const int colbase = (hx * HXY + hy) * HZ;
            const int li = h_off[colbase + zlo] + (q - c_off[col]); // LDS index of the centre atom
            int hz = zlo;
            while (hz < zhi && li >= h_off[colbase + hz + 1]) ++hz;
            double xi = lx[li], yi = ly[li], zi = lz[li];
            if (b.anypbc) // neighbor.cpp:139-142
                wrap<false>(b, xi, yi, zi);
            // z image codes of the three cells of every run of this centre (only the seam cells differ from 0)
            const int zc_m = (h_img[colbase + hz - 1] >> 4) & 3, zc_p = (h_img[colbase + hz + 1] >> 4) & 3;
            const double sz_m = img_shift(b.h[8], zc_m), sz_p = img_shift(b.h[8], zc_p);
            unsigned short *my = tick + (size_t)tid * M;
            int hits = 0;
            for (int da = -1; da <= 1; ++da)       // neighbor.cpp:147-151
                for (int db = -1; db <= 1; ++db) {
                    const int cb = ((hx + da) * HXY + (hy + db)) * HZ + hz;
                    const int xy = h_img[cb] & 15;
                    const double sx = img_shift(b.h[0], xy & 3), sy = img_shift(b.h[4], xy >> 2);
                    const int k0 = h_off[cb - 1], k1 = h_off[cb], k2 = h_off[cb + 1], k3 = h_off[cb + 2];
                    for (int k = k0; k < k3; k += 4) {
                        double d2[4];
                        int kk[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            kk[u] = min(k + u, k3 - 1); // clamped: the tail re-reads the last candidate, masked below
                            const double sz = CELLSHIFT ? (kk[u] < k1 ? sz_m : (kk[u] < k2 ? 0.0 : sz_p)) : 0.0;
                            d2[u] = pair_d2_tiled<CELLSHIFT>(b, lx[kk[u]], ly[kk[u]], lz[kk[u]], xi, yi, zi, sx, sy, sz);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool hit = (k + u < k3) && (kk[u] != li) && (d2[u] <= rcsq);
                            if (hit) {
                                if (hits < M) // 10 bits LDS index | 2 bits z-cell of the run | 4 bits xy image code
                                    my[hits] = (unsigned short)(kk[u] | ((kk[u] < k1 ? 0 : (kk[u] < k2 ? 1 : 2)) << 10) | (xy << 12));
                                ++hits;
                            }
                        }
                    }
                }
            const int i = lid[li];
            nn[i] = hits; // keeps counting past M (neighbor.cpp:172-177)
            crow[tid] = i;
            cinfo[tid] = (hits < M ? hits : M) | (zc_m << 16) | (zc_p << 18);
            cxi[tid] = xi; cyi[tid] = yi; czi[tid] = zi;
        }
        __syncthreads();
        // ---- tickets -> rows: MP adjacent lanes serve the slots of one centre
        const int nrows = min(NT, ncentres - base);
        const int e = tid & (MP - 1);
        for (int c = tid >> mp_shift; c < nrows; c += (NT >> mp_shift)) {
            if (e >= M)
                continue;
            const int info = cinfo[c];
            const int64_t o = (int64_t)crow[c] * M + e;
            if (e < (info & 0xffff)) {
                const unsigned tk = tick[c * M + e];
                const int k = (int)(tk & 1023u), zsel = (int)((tk >> 10) & 3u), xy = (int)(tk >> 12);
                const int zc = zsel == 0 ? ((info >> 16) & 3) : (zsel == 1 ? 1 : ((info >> 18) & 3));
                const double d2 = pair_d2_tiled<CELLSHIFT>(b, lx[k], ly[k], lz[k], cxi[c], cyi[c], czi[c],
                                                           img_shift(b.h[0], xy & 3), img_shift(b.h[4], xy >> 2),
                                                           CELLSHIFT ? (zsel == 1 ? 0.0 : img_shift(b.h[8], zc)) : 0.0);
                verlet[o] = lid[k];
                dist[o] = sqrt(d2);
            } else if (MODE == 2) {
                verlet[o] = -1;
                dist[o] = pad;
            }
        }
        __syncthreads();
    }
}

static size_t tiled_lds_bytes(int64_t M)
{
    return (size_t)HALO_CAP * 28 + (size_t)NT * (24 + 4 + 4) + (size_t)NT * (size_t)M * 2;
}

// pick the tile shape for a mean cell population `pop`
static TileShape choose_shape(double pop, const Grid &g)
{
    TileShape best{0, 0};
    double best_score = -1.0;
    for (int txy = 2; txy <= 8; ++txy)
        for (int tz = 2; tz <= 16; ++tz) {
            const int nh = (txy + 2) * (txy + 2) * (tz + 2);
            if (nh > MAX_NH || txy * txy > MAX_COLS)
                continue;
            if (nh * pop > 0.86 * HALO_CAP) // head-room for density fluctuations; overflowing tiles fall back
                continue;
            const double c = txy * txy * tz * pop;                       // centre atoms per tile
            const double util = c / (std::ceil(c / NT) * NT);           // lane utilisation of the scan
            const double reuse = (double)(txy * txy * tz) / (double)nh; // centre cells per staged cell
            const double score = util * (0.35 + reuse);
            if (score > best_score) { best_score = score; best = TileShape{txy, tz}; }
        }
    (void)g;
    return best;
}

TiledPlan plan_tiled(const DBox &b, const Grid &g, int64_t N, int64_t M)
{
    TiledPlan p{0, 0, false};
    if (b.tri || g.mode != 0 || N <= 0 || M <= 0)
        return p;
    const double pop = (double)N / (double)g.ncell; // mean atoms per cell
    const TileShape sh = choose_shape(pop, g);
    if (!sh.txy)
        return p;
    if (tiled_lds_bytes(M) > 62 * 1024) // ticket rows must fit next to the halo (<= 64 KiB: two workgroups per CU)
        return p;
    int mp = 1;
    while (mp < M) mp <<= 1;
    if (mp > NT)
        return p;
    p.tile = sh.txy;
    p.tile_z = sh.tz;
    p.cellshift = true;
    for (int d = 0; d < 3; ++d)
        if (b.pbc[d] && g.nc[d] < 7)
            p.cellshift = false;
    return p;
}

template <bool CS>
static void launch_one(hipStream_t st, const CellGrid &cg, const DBox &b, double rc, int *verlet, double *dist, int *nn,
                       int M, bool fill_pads, unsigned char *tile_flag, const int *nt, int want_moved, TileShape ts)
{
    const int ntiles = nt[0] * nt[1] * nt[2];
    const int per = (ntiles + 7) / 8;
    dim3 grid((unsigned)(per * 8)), block(NT);
    const size_t lds = tiled_lds_bytes(M);
    int mp_shift = 0;
    while ((1 << mp_shift) < M) ++mp_shift;
    if (fill_pads)
        hipLaunchKernelGGL((k_neighbor_tiled<CS, 2>), grid, block, lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, b, cg.g, rc, verlet, dist, nn, M, mp_shift, cg.flags, tile_flag, nt[0], nt[1], nt[2], want_moved, ts);
    else
        hipLaunchKernelGGL((k_neighbor_tiled<CS, 1>), grid, block, lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, b, cg.g, rc, verlet, dist, nn, M, mp_shift, cg.flags, tile_flag, nt[0], nt[1], nt[2], want_moved, ts);
}

int launch_neighbor_tiled(Scope &sc, const CellGrid &cg, const TiledPlan &plan, int64_t N, const DBox &b, double rc,
                          int *verlet, double *dist, int *nn, int64_t M, bool fill_pads, TileFilter &tf)
{
    const TileShape ts{plan.tile, plan.tile_z};
    int nt[3];
    for (int d = 0; d < 3; ++d) {
        const int T = d == 2 ? ts.tz : ts.txy;
        nt[d] = (cg.g.nc[d] + T - 1) / T;
    }
    const int64_t ntiles = (int64_t)nt[0] * nt[1] * nt[2];
    unsigned char *tile_flag = sc.alloc_n<unsigned char>((size_t)ntiles);
    if (sc.failed())
        return sc.error();
    hipStream_t st = sc.stream();
    MDH_HIP(hipMemsetAsync(tile_flag, 0, (size_t)ntiles, st));
    // Two launches, one of which returns at once on the device flag: per-cell image shifts when every atom
    // came in wrapped (and the grid allows it), the exact threshold search otherwise.
    if (plan.cellshift) launch_one<true>(st, cg, b, rc, verlet, dist, nn, (int)M, fill_pads, tile_flag, nt, 0, ts);
    launch_one<false>(st, cg, b, rc, verlet, dist, nn, (int)M, fill_pads, tile_flag, nt, plan.cellshift ? 1 : -1, ts);
    MDH_HIP(hipGetLastError());
    tf.flag = tile_flag;
    tf.any = cg.flags + 2;
    tf.tile = ts.txy;
    tf.tile_z = ts.tz;
    tf.nt[0] = nt[0]; tf.nt[1] = nt[1]; tf.nt[2] = nt[2];
    return MDH_OK;
}

} // namespace mdh
