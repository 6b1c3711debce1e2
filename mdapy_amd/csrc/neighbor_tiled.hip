// neighbor_tiled.hip — LDS-tiled 27-cell neighbor scan for orthogonal boxes (gfx950).
//
// Same result, bit for bit, as the thread-per-atom kernel in neighbor.hip (and therefore as
// src/neighbor.cpp:102-187 of the reference); this is the fast path for the common case.
//
// One workgroup owns a tile of TXY x TXY x TZ cells (shape chosen on the host from the mean cell
// population).  It stages the atoms of the halo cells (raw x,y,z + atom id) from the cell-sorted
// arrays into LDS with coalesced loads — every cell of the grid is read from HBM/L2 once per
// neighbouring tile instead of once per neighbouring ATOM — then each thread takes one centre atom
// of the tile and walks its 27 cells out of LDS in the reference's order (cells (i,j,k)-
// lexicographic, atoms of a cell by descending id).  The three z-cells of one (i,j) column are
// contiguous in LDS, so the walk is 9 runs of candidates, each processed four at a time
// (independent f64 chains).
//
// Minimum image.  n = floor(d/L + 0.5) of a (centre, candidate) pair is known without evaluating it
// when there are >= 7 cells on every periodic axis: candidates of an adjacent cell are < 3 rc <= 3L/7
// away after the right shift, far from the +-L/2 decision points, so
//        n = n_cell + m_atom
// where n_cell in {-1,0,1} says whether the candidate's cell was reached across the periodic seam and
// m_atom in {-1,0,1} is the whole number of box lengths by which the candidate's RAW coordinate differs
// from its wrapped one (0 for input that is already wrapped; recorded by the binning pass).  The
// kernel then evaluates the reference's  d - L*n  with that n (same operands, same roundings; L*n is
// exact for |n| <= 2), removing 3 divisions and 3 floors per candidate.  Tiles away from the seam
// whose atoms all have m = 0 (almost all of them) skip the shift altogether: d - L*0 == d.  When the
// precondition fails (flag from the binning pass, or < 7 cells) the exact threshold search of
// common.hpp::pbc_axis is used instead.
//
// Output.  A hit is first recorded in LDS as a 2-byte ticket (LDS index of the candidate).  After the
// scan the workgroup turns tickets into rows cooperatively: consecutive lanes write consecutive
// slots of a row, so a wave store covers whole 64 B / 128 B row segments instead of 64 scattered
// rows; the distance is recomputed from the same operands (identical bits).  With fill_pads the
// same pass writes the -1 / rc+1 pads.
//
// Partly empty boxes (vacuum around a slab or a particle, a rank's slab of a decomposed system).  The tile shape is sized
// for the population of the OCCUPIED region (occupied_cells_hint: counted on the device, read by the next call), the XCD
// chunks are cut from the list of tiles that hold centre atoms (k_tile_live + scan), and the launch grid follows the same
// estimate — a workgroup takes further tiles of its chunk when the estimate was too small.  A box that is full of atoms
// takes the straight-line instance (LIST = false): workgroup b owns tile b.
#include "common.hpp"
#include "grid.hpp"
#include <algorithm>
#include <mutex>
#include <vector>

namespace mdh {

#ifndef MDH_HALO_CAP
#define MDH_HALO_CAP 1024
#endif
static constexpr int HALO_CAP = MDH_HALO_CAP; // atoms a tile's halo may hold in LDS (30 B each)
static constexpr int NT = 256;        // threads per workgroup
#ifndef MDH_MAX_NH
#define MDH_MAX_NH 512
#endif
static constexpr int MAX_NH = MDH_MAX_NH;    // halo cells a tile may have
static constexpr int MAX_COLS = 64;   // (x,y) columns of centre cells a tile may have
static constexpr int TICK_STRIDE = NT + 2; // ticket slot stride (u16 units): 516 B -> consecutive slots shift by one bank
static constexpr int NEUTRAL = img::CELL_NEUTRAL; // a halo cell's image code "no shift" (grid.hpp img::)

// Tile shape (cells): TXY x TXY x TZ
struct TileShape { int txy, tz; };

__device__ __forceinline__ int excl_scan_block(int v, int *scratch, int *total)
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
#pragma unroll
    for (int k = 0; k < (NT >> 6); ++k) {
        if (k < w) off += scratch[k];
        tot += scratch[k];
    }
    __syncthreads();
    *total = tot;
    return off + inc - v;
}

// L * n for a staged atom's combined image number n = n_cell + m_atom in [-2, 2], stored as n + 2 in three bits per axis
// cell code cc and atom code ca -> combined code (grid.hpp img::)
__device__ __forceinline__ int combine_codes(int cc, int ca) { return img::combine(cc, ca); }

// squared distance of one (centre, candidate) pair.
//   PBCMODE 0: no shift at all (tile away from the seam, all atoms wrapped): d - L*0 == d
//   PBCMODE 1: d - L*n with n from the image codes
//   PBCMODE 2: exact threshold / division search (common.hpp)
template <int PBCMODE>
__device__ __forceinline__ double pair_d2_tiled(const DBox &b, double xj, double yj, double zj, double xi, double yi,
                                                double zi, int sh)
{
    double dx = xj - xi, dy = yj - yi, dz = zj - zi; // raw x[j] - wrapped centre (neighbor.cpp:164-166)
    if (PBCMODE == 1) {
        dx = dx - b.h[0] * (double)img::axis(sh, 0); // == xij - L*floor(xij/L+0.5)   (box.h:120-124)
        dy = dy - b.h[4] * (double)img::axis(sh, 1);
        dz = dz - b.h[8] * (double)img::axis(sh, 2);
    } else if (PBCMODE == 2) {
        pbc<false>(b, dx, dy, dz);
    }
    return dx * dx + dy * dy + dz * dz;
}

struct TileLds {
    double *lx, *ly, *lz;        // staged raw positions [HALO_CAP]
    int *lid;                    // staged atom ids [HALO_CAP]
    unsigned short *lsh;         // combined image code of a staged atom as seen from this tile [HALO_CAP]
};

// one run of candidates [k0, k3) for centre li; tickets (LDS indices) appended to my[]
template <int PBCMODE, bool SELF>
__device__ __forceinline__ void scan_run(const DBox &b, const TileLds &L, int k0, int k3, int li, double xi, double yi,
                                         double zi, double rcsq, unsigned short *__restrict__ my, int M, int &hits)
{
    // four independent chains per step; the last step re-reads the run's last candidate for the lanes past the
    // end (clamped index) and masks them out.  Tickets are stored slot-major with a padded stride
    // (my[slot * TICK_STRIDE]) so that the lanes of a wave hit different LDS banks.
    // `slot` walks the centre's ticket column (stride TICK_STRIDE) instead of being recomputed from `hits`.
    unsigned short *slot = my + (hits < M ? hits : M) * TICK_STRIDE;
    for (int k = k0; k < k3; k += 4) {
        double d2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // lanes past the end of the run read whatever follows in LDS (the arrays are contiguous) and are masked out
            const int qq = k + u;
            int sh = 0;
            if (PBCMODE == 1) sh = L.lsh[qq];
            d2[u] = pair_d2_tiled<PBCMODE>(b, L.lx[qq], L.ly[qq], L.lz[qq], xi, yi, zi, sh);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool hit = (d2[u] <= rcsq) && (k + u < k3) && (!SELF || (k + u != li));
            if (hit) {
                if (hits < M) { *slot = (unsigned short)(k + u); slot += TICK_STRIDE; }
                ++hits;
            }
        }
    }
}

template <int PBCMODE>
__device__ __forceinline__ int scan_centre(const DBox &b, const TileLds &L, const unsigned short *__restrict__ h_off,
                                           int HXY, int HZ, int hx, int hy, int hz, int li, double xi, double yi,
                                           double zi, double rcsq, unsigned short *__restrict__ my, int M)
{
    // bounds of the nine runs first (18 independent LDS reads, one wait), then the runs in reference order
    int k0s[9], k3s[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) { // r = (da+1)*3 + (db+1): neighbor.cpp:147-151
        const int cb = ((hx + r / 3 - 1) * HXY + (hy + r % 3 - 1)) * HZ + hz;
        k0s[r] = h_off[cb - 1]; // cells hz-1, hz, hz+1 of this column are contiguous
        k3s[r] = h_off[cb + 2];
    }
    int hits = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        if (r == 4)
            scan_run<PBCMODE, true>(b, L, k0s[r], k3s[r], li, xi, yi, zi, rcsq, my, M, hits);
        else
            scan_run<PBCMODE, false>(b, L, k0s[r], k3s[r], li, xi, yi, zi, rcsq, my, M, hits);
    }
    return hits;
}

// CELLSHIFT: image numbers from codes (PBCMODE 0/1 chosen per tile); otherwise the exact search (PBCMODE 2)
// LIST: the workgroup takes its tiles from the list of live tiles and may take more than one (partly empty box); otherwise
// workgroup b owns tile b of a box that is full of atoms (straight-line code)
template <bool CELLSHIFT, int MODE, bool LIST>
__global__ __launch_bounds__(NT) void k_neighbor_tiled(
    const double *__restrict__ xs, const double *__restrict__ ys, const double *__restrict__ zs,
    const int *__restrict__ order, const unsigned short *__restrict__ mvs, const int *__restrict__ cell_start, DBox b,
    Grid g, double rc, int *__restrict__ verlet, double *__restrict__ dist, int *__restrict__ nn, int M, int mp_shift,
    int *__restrict__ flags, unsigned char *__restrict__ tile_flag, int nt0, int nt1, int nt2, int want_moved,
    TileShape ts, const int *__restrict__ tile_list, const int *__restrict__ n_live)
{
    const int TXY = ts.txy, TZ = ts.tz;
    const int HXY = TXY + 2, HZ = TZ + 2, NH = HXY * HXY * HZ, NCOL = TXY * TXY;
    // which of the two minimum-image variants serves this call is decided on the device (no host sync)
    // want_moved: 0 = run only if the image codes are valid, 1 = run only if not, -1 = always run
    if (want_moved >= 0 && (flags[0] != 0) != (want_moved != 0))
        return;

    extern __shared__ unsigned char smem[];
    TileLds L;
    L.lx = reinterpret_cast<double *>(smem);
    L.ly = L.lx + HALO_CAP;
    L.lz = L.ly + HALO_CAP;
    double *cxi = L.lz + HALO_CAP; // wrapped centre coordinates of this pass [NT]
    double *cyi = cxi + NT;
    double *czi = cyi + NT;
    L.lid = reinterpret_cast<int *>(czi + NT);
    int *crow = L.lid + HALO_CAP;                 // global atom id of the centre [NT]
    int *ccnt = crow + NT;                        // min(count, M) [NT]
    L.lsh = reinterpret_cast<unsigned short *>(ccnt + NT);
    unsigned short *tick = L.lsh + HALO_CAP;      // [M][TICK_STRIDE]
    __shared__ unsigned short h_off[MAX_NH + 2];
    __shared__ int c_off[MAX_COLS + 1];
    __shared__ int scan_tmp[4];

    // XCD-aware tile order: block b runs on XCD b%8; give every XCD one contiguous chunk of the tiles THAT HOLD CENTRE ATOMS
    // (tile_list, in tile order) so that neighbouring tiles (which share halo cells) meet in the same L2 and an empty part
    // of the box (vacuum, the other ranks' slabs of a decomposed system) leaves no XCD idle.
    const int nlive = (LIST && tile_list) ? *n_live : nt0 * nt1 * nt2; // no list: every tile is live
    const int per = (nlive + 7) / 8;
    const int tid = threadIdx.x;
    // the grid is sized from the last known occupancy; a workgroup takes further tiles of its XCD's chunk if that was too few
    for (int jt = (int)(blockIdx.x >> 3); jt < per; jt += (int)(gridDim.x >> 3)) {
    const int slot = (blockIdx.x & 7) * per + jt;
    if (slot >= nlive)
        break;
    const int tile_id = (LIST && tile_list) ? tile_list[slot] : slot;
    const int t2 = tile_id % nt2, t1 = (tile_id / nt2) % nt1, t0 = tile_id / (nt2 * nt1);
    const int T0 = t0 * TXY, T1 = t1 * TXY, T2 = t2 * TZ;

    // ---- halo cell table: source range, LDS offset, image code.  Thread t owns halo cells 2t and 2t+1
    // (adjacent in z, hence adjacent in memory).
    const float inv_hz = 1.0f / (float)HZ, inv_hxy = 1.0f / (float)HXY;
    int cnt2[2] = {0, 0}, src2[2] = {0, 0}, img2[2] = {NEUTRAL, NEUTRAL};
    bool general = false; // does this tile need image shifts at all?
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int h = 2 * tid + u;
        if (h >= NH)
            continue;
        int cnt = 0, src = 0;
        // h < 512: quotients by the (runtime) tile extents through exact float reciprocals instead of integer division
        const int hcol = (int)(((float)h + 0.5f) * inv_hz); // h / HZ
        const int hz = h - hcol * HZ;
        const int hx = (int)(((float)hcol + 0.5f) * inv_hxy); // hcol / HXY
        const int hy = hcol - hx * HXY;
        const int g0 = T0 + hx - 1, g1 = T1 + hy - 1, g2 = T2 + hz - 1;
        int img = NEUTRAL;
        if (g0 >= -1 && g0 <= g.nc[0] && g1 >= -1 && g1 <= g.nc[1] && g2 >= -1 && g2 <= g.nc[2]) {
            // g in [-1, nc]: the positive modulo (neighbor.cpp:18-22) is one conditional add / subtract
            const int a0 = g0 < 0 ? g0 + g.nc[0] : (g0 >= g.nc[0] ? g0 - g.nc[0] : g0);
            const int a1 = g1 < 0 ? g1 + g.nc[1] : (g1 >= g.nc[1] ? g1 - g.nc[1] : g1);
            const int a2 = g2 < 0 ? g2 + g.nc[2] : (g2 >= g.nc[2] ? g2 - g.nc[2] : g2);
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
        img2[u] = img;
        general = general || (img != NEUTRAL && cnt > 0);
        cnt2[u] = cnt;
        src2[u] = src;
    }
    int total;
    const int off0 = excl_scan_block(cnt2[0] + cnt2[1], scan_tmp, &total);
    if (total > HALO_CAP) { // leave this tile to the thread-per-atom kernel
        if (tid == 0) {
            tile_flag[tile_id] = 1;
            atomicAdd(&flags[2], 1);
        }
        continue; // (excl_scan_block ended with a barrier)
    }
    if (2 * tid < NH) h_off[2 * tid] = (unsigned short)off0;
    if (2 * tid + 1 < NH) h_off[2 * tid + 1] = (unsigned short)(off0 + cnt2[0]);
    if (tid == 0) { h_off[NH] = (unsigned short)total; h_off[NH + 1] = (unsigned short)total; }
    // ---- stage the halo atoms (each thread copies its two cells: neighbouring threads read neighbouring memory)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int cnt = cnt2[u], src = src2[u], off = off0 + (u ? cnt2[0] : 0);
        int k = 0;
        for (; k + 4 <= cnt; k += 4) { // four independent loads in flight per array
            double a[4], bb[4], c[4];
            int d[4];
            int m[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) { a[v] = xs[src + k + v]; bb[v] = ys[src + k + v]; c[v] = zs[src + k + v]; d[v] = order[src + k + v]; m[v] = mvs[src + k + v]; }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                L.lx[off + k + v] = a[v]; L.ly[off + k + v] = bb[v]; L.lz[off + k + v] = c[v]; L.lid[off + k + v] = d[v];
                L.lsh[off + k + v] = (unsigned short)combine_codes(img2[u], m[v]);
                general = general || (m[v] != img::ATOM_NEUTRAL);
            }
        }
        for (; k < cnt; ++k) {
            const int m = mvs[src + k];
            L.lx[off + k] = xs[src + k]; L.ly[off + k] = ys[src + k]; L.lz[off + k] = zs[src + k]; L.lid[off + k] = order[src + k];
            L.lsh[off + k] = (unsigned short)combine_codes(img2[u], m);
            general = general || (m != img::ATOM_NEUTRAL);
        }
    }
    const int tile_general = __syncthreads_or(general ? 1 : 0); // also publishes h_off / staged atoms
    // ---- centre runs: one contiguous LDS run per (x,y) column of the tile, clipped to the grid
    const int zlo = 1, zhi = min(TZ, g.nc[2] - T2); // interior hz in [1, zhi]
    if (tid < 64) { // one wave: per-column centre counts -> exclusive prefix
        int v = 0;
        if (tid < NCOL) {
            const int hx = tid / TXY + 1, hy = tid % TXY + 1;
            const bool ok = (T0 + hx - 1 < g.nc[0]) && (T1 + hy - 1 < g.nc[1]) && zhi >= 1;
            v = ok ? ((int)h_off[(hx * HXY + hy) * HZ + zhi + 1] - (int)h_off[(hx * HXY + hy) * HZ + zlo]) : 0;
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
            int col = 0; // largest column with c_off[col] <= q
            for (int step = 32; step > 0; step >>= 1)
                if (col + step < NCOL && q >= c_off[col + step]) col += step;
            const int hx = col / TXY + 1, hy = col % TXY + 1;
            const int colbase = (hx * HXY + hy) * HZ;
            const int li = (int)h_off[colbase + zlo] + (q - c_off[col]); // LDS index of the centre atom
            int hz = zlo;
            while (hz < zhi && li >= (int)h_off[colbase + hz + 1]) ++hz;
            double xi = L.lx[li], yi = L.ly[li], zi = L.lz[li];
            if (b.anypbc) // neighbor.cpp:139-142
                wrap<false>(b, xi, yi, zi);
            unsigned short *my = tick + tid; // slot s of this centre at my[s * TICK_STRIDE]
            int hits;
            if (!CELLSHIFT) hits = scan_centre<2>(b, L, h_off, HXY, HZ, hx, hy, hz, li, xi, yi, zi, rcsq, my, M);
            else if (tile_general) hits = scan_centre<1>(b, L, h_off, HXY, HZ, hx, hy, hz, li, xi, yi, zi, rcsq, my, M);
            else hits = scan_centre<0>(b, L, h_off, HXY, HZ, hx, hy, hz, li, xi, yi, zi, rcsq, my, M);
            const int i = L.lid[li];
            nn[i] = hits; // keeps counting past M (neighbor.cpp:172-177)
            crow[tid] = i;
            ccnt[tid] = hits < M ? hits : M;
            cxi[tid] = xi; cyi[tid] = yi; czi[tid] = zi;
        }
        __syncthreads();
        // ---- tickets -> rows: MP adjacent lanes serve the slots of one centre
        const int nrows = min(NT, ncentres - base);
        const int e = tid & (MP - 1);
        for (int c = tid >> mp_shift; c < nrows; c += (NT >> mp_shift)) {
            if (e >= M)
                continue;
            const int64_t o = (int64_t)crow[c] * M + e;
            if (e < ccnt[c]) {
                const int k = tick[e * TICK_STRIDE + c];
                double d2;
                if (!CELLSHIFT) d2 = pair_d2_tiled<2>(b, L.lx[k], L.ly[k], L.lz[k], cxi[c], cyi[c], czi[c], 0);
                else if (tile_general) d2 = pair_d2_tiled<1>(b, L.lx[k], L.ly[k], L.lz[k], cxi[c], cyi[c], czi[c], L.lsh[k]);
                else d2 = pair_d2_tiled<0>(b, L.lx[k], L.ly[k], L.lz[k], cxi[c], cyi[c], czi[c], 0);
                verlet[o] = L.lid[k];
                dist[o] = sqrt(d2);
            } else if (MODE == 2) {
                verlet[o] = -1;
                dist[o] = pad;
            }
        }
        __syncthreads();
    }
    if (!LIST)
        break;
    if (jt + (int)(gridDim.x >> 3) < per) __syncthreads(); // LDS is reused by the next tile
    } // tiles of this workgroup
}

// tiles with at least one centre atom: flag (one thread per tile), then an order-preserving compaction
__global__ __launch_bounds__(256) void k_tile_live(const int *__restrict__ cell_start, Grid g, int nt0, int nt1, int nt2, TileShape ts,
                                                   unsigned *__restrict__ live)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt0 * nt1 * nt2)
        return;
    const int t2 = t % nt2, t1 = (t / nt2) % nt1, t0 = t / (nt2 * nt1);
    const int z0 = t2 * ts.tz, z1 = min(z0 + ts.tz, g.nc[2]);
    bool any = false;
    for (int a = t0 * ts.txy; a < min((t0 + 1) * ts.txy, g.nc[0]) && !any; ++a)
        for (int c = t1 * ts.txy; c < min((t1 + 1) * ts.txy, g.nc[1]) && !any; ++c) {
            const int64_t col = ((int64_t)a * g.nc[1] + c) * g.nc[2];
            any = cell_start[col + z1] > cell_start[col + z0]; // the z-run of a column is contiguous
        }
    live[t] = any ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_tile_compact(const unsigned *__restrict__ live, const int *__restrict__ slot, int ntiles,
                                                      int *__restrict__ tile_list)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntiles && live[t]) tile_list[slot[t]] = t;
}

static size_t tiled_lds_bytes(int64_t M)
{
    return (size_t)HALO_CAP * (24 + 4 + 2) + (size_t)NT * (24 + 4 + 4) + (size_t)TICK_STRIDE * (size_t)M * 2;
}

// pick the tile shape for a mean cell population `pop`
static TileShape choose_shape(double pop)
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
    return best;
}

// ---------------------------------------------------------------------------------------------------------------
// How many cells hold atoms?  The tile shape is sized for the population of the OCCUPIED cells: a box that is mostly empty
// (vacuum around a slab or a particle, the other ranks' slabs of a decomposed system) would otherwise get tiles whose
// halo overflows LDS wherever the atoms are.  The count is taken on the device in every call; the host uses the value the
// previous call with the same (N, grid) left in pinned memory — an MD-style sequence of calls never waits for it — and
// waits only the first time it sees a new (N, grid).  A stale value costs speed, never correctness (overflowing tiles
// fall back to the thread-per-atom kernel).
// counted in blocks of 4 x 4 x 4 cells (a block counts with all its cells as soon as one of them holds an atom): the empty
// cells that a lattice leaves between its occupied ones belong to the occupied region, vacuum does not
__global__ __launch_bounds__(256) void k_count_occupied(const int *__restrict__ cell_start, Grid g, int *__restrict__ out)
{
    const int nb0 = (g.nc[0] + 3) >> 2, nb1 = (g.nc[1] + 3) >> 2, nb2 = (g.nc[2] + 3) >> 2;
    const int64_t nblk = (int64_t)nb0 * nb1 * nb2;
    int mine = 0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nblk; q += (int64_t)gridDim.x * blockDim.x) {
        const int b2 = (int)(q % nb2), b1 = (int)((q / nb2) % nb1), b0 = (int)(q / ((int64_t)nb2 * nb1));
        const int x1 = min(b0 * 4 + 4, g.nc[0]), y1 = min(b1 * 4 + 4, g.nc[1]), z0 = b2 * 4, z1 = min(z0 + 4, g.nc[2]);
        bool any = false;
        for (int a = b0 * 4; a < x1 && !any; ++a)
            for (int c = b1 * 4; c < y1 && !any; ++c) {
                const int64_t col = ((int64_t)a * g.nc[1] + c) * g.nc[2];
                any = cell_start[col + z1] > cell_start[col + z0];
            }
        if (any) mine += (x1 - b0 * 4) * (y1 - b1 * 4) * (z1 - z0);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

namespace {
struct OccEntry { int64_t N, ncell; int device; int *host; unsigned calls; };
std::mutex g_occ_mu;
std::vector<OccEntry> g_occ;
} // namespace

int occupied_cells_hint(Scope &sc, const CellGrid &cg, int64_t N, int64_t *occupied)
{
    hipStream_t st = sc.stream();
    int device = 0;
    (void)hipGetDevice(&device);
    std::lock_guard<std::mutex> lk(g_occ_mu);
    auto count = [&](int *host_dst) -> int {
        int *dcnt = sc.alloc_n<int>(1);
        if (sc.failed())
            return sc.error();
        MDH_HIP(hipMemsetAsync(dcnt, 0, sizeof(int), st));
        const int blocks = (int)std::min<int64_t>((cg.g.ncell / 64 + 255) / 256 + 1, 4096);
        hipLaunchKernelGGL(k_count_occupied, dim3(blocks), dim3(256), 0, st, cg.cell_start, cg.g, dcnt);
        MDH_HIP(hipMemcpyAsync(host_dst, dcnt, sizeof(int), hipMemcpyDeviceToHost, st));
        return MDH_OK;
    };
    for (auto &e : g_occ)
        if (e.N == N && e.ncell == cg.g.ncell && e.device == device) {
            const int last = *(volatile int *)e.host;
            *occupied = last > 0 ? last : cg.g.ncell;
            if ((++e.calls & 7u) == 0) // the occupied region of a running simulation drifts slowly: recount every 8th call
                MDH_TRY(count(e.host));
            return MDH_OK;
        }
    int *host = nullptr;
    if (g_occ.size() >= 64) { // keep the table small: the oldest signature hands its pinned word on (never freed: a copy
        host = g_occ.front().host; // enqueued on some other stream may still land in it — a wrong hint at worst)
        g_occ.erase(g_occ.begin());
    } else {
        MDH_HIP(hipHostMalloc(reinterpret_cast<void **>(&host), sizeof(int), hipHostMallocDefault));
    }
    MDH_TRY(count(host));
    MDH_HIP(hipStreamSynchronize(st));
    *occupied = *host > 0 ? *host : cg.g.ncell;
    g_occ.push_back(OccEntry{N, cg.g.ncell, device, host, 0u});
    return MDH_OK;
}

TiledPlan plan_tiled(const DBox &b, const Grid &g, int64_t N, int64_t M, int64_t occupied_cells)
{
    TiledPlan p{0, 0, false, false, 0};
    if (b.tri || g.mode != 0 || N <= 0 || M <= 0)
        return p;
    const double pop = (double)N / (double)(occupied_cells > 0 ? occupied_cells : g.ncell); // mean atoms per occupied cell
    const TileShape sh = choose_shape(pop);
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
    p.full = occupied_cells >= g.ncell;
    p.occupied = occupied_cells > 0 ? occupied_cells : g.ncell;
    p.cellshift = true;
    for (int d = 0; d < 3; ++d)
        if (b.pbc[d] && g.nc[d] < 7)
            p.cellshift = false;
    return p;
}

template <bool CS>
static void launch_one(hipStream_t st, const CellGrid &cg, const DBox &b, double rc, int *verlet, double *dist, int *nn,
                       int M, bool fill_pads, unsigned char *tile_flag, const int *nt, int want_moved, TileShape ts,
                       const int *tile_list, const int *n_live, int64_t est_live, bool standby)
{
    const int ntiles = nt[0] * nt[1] * nt[2];
    int per = (ntiles + 7) / 8;
    if (est_live > 0) // tiles expected to hold atoms (+25 %); more than that and workgroups loop (k_neighbor_tiled)
        per = std::max(1, std::min(per, (int)((est_live + est_live / 4 + 7) / 8)));
    if (standby) // the variant that the device flag will most likely send home: three workgroups per CU, looping if it does run
        per = std::min(per, 96);
    dim3 grid((unsigned)(per * 8)), block(NT);
    const size_t lds = tiled_lds_bytes(M);
    int mp_shift = 0;
    while ((1 << mp_shift) < M) ++mp_shift;
    const bool list = tile_list != nullptr || standby || per * 8 < ntiles; // full box, one tile per workgroup: the straight-line kernel
    if (fill_pads) {
        if (list) hipLaunchKernelGGL((k_neighbor_tiled<CS, 2, true>), grid, block, lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.mvs, cg.cell_start, b, cg.g, rc, verlet, dist, nn, M, mp_shift, cg.flags, tile_flag, nt[0], nt[1], nt[2], want_moved, ts, tile_list, n_live);
        else hipLaunchKernelGGL((k_neighbor_tiled<CS, 2, false>), grid, block, lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.mvs, cg.cell_start, b, cg.g, rc, verlet, dist, nn, M, mp_shift, cg.flags, tile_flag, nt[0], nt[1], nt[2], want_moved, ts, tile_list, n_live);
    } else {
        if (list) hipLaunchKernelGGL((k_neighbor_tiled<CS, 1, true>), grid, block, lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.mvs, cg.cell_start, b, cg.g, rc, verlet, dist, nn, M, mp_shift, cg.flags, tile_flag, nt[0], nt[1], nt[2], want_moved, ts, tile_list, n_live);
        else hipLaunchKernelGGL((k_neighbor_tiled<CS, 1, false>), grid, block, lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.mvs, cg.cell_start, b, cg.g, rc, verlet, dist, nn, M, mp_shift, cg.flags, tile_flag, nt[0], nt[1], nt[2], want_moved, ts, tile_list, n_live);
    }
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
    unsigned *live = sc.alloc_n<unsigned>((size_t)ntiles);
    int *slot = sc.alloc_n<int>((size_t)ntiles + 1);
    int *tile_list = sc.alloc_n<int>((size_t)ntiles);
    if (sc.failed())
        return sc.error();
    hipStream_t st = sc.stream();
    MDH_HIP(hipMemsetAsync(tile_flag, 0, (size_t)ntiles, st));
    // live tiles expected from the last known occupancy: occupied cells / cells per tile (tiles are cut partly empty at the
    // surface of the occupied region, hence the head-room in launch_one)
    const int64_t est_live = plan.full ? 0 : std::max<int64_t>(1, plan.occupied / std::max(1, ts.txy * ts.txy * ts.tz) * 2);
    if (plan.full) { // the occupancy count says that no 4x4x4 block of cells is empty: all tiles are live, no list needed
        tile_list = nullptr;
    } else {
        hipLaunchKernelGGL(k_tile_live, dim3(grid_for(ntiles, 256)), dim3(256), 0, st, cg.cell_start, cg.g, nt[0], nt[1], nt[2], ts, live);
        MDH_TRY(exclusive_scan_u32(sc, live, slot, ntiles)); // slot[ntiles] = number of live tiles
        hipLaunchKernelGGL(k_tile_compact, dim3(grid_for(ntiles, 256)), dim3(256), 0, st, live, slot, (int)ntiles, tile_list);
    }
    // Two launches, one of which returns at once on the device flag: image numbers from the cell / atom codes when
    // the binning pass found them valid (and the grid allows it), the exact threshold search otherwise.
    if (plan.cellshift) launch_one<true>(st, cg, b, rc, verlet, dist, nn, (int)M, fill_pads, tile_flag, nt, 0, ts, tile_list, slot + ntiles, est_live, false);
    launch_one<false>(st, cg, b, rc, verlet, dist, nn, (int)M, fill_pads, tile_flag, nt, plan.cellshift ? 1 : -1, ts, tile_list, slot + ntiles, est_live, plan.cellshift);
    MDH_HIP(hipGetLastError());
    tf.flag = tile_flag;
    tf.any = cg.flags + 2;
    tf.tile = ts.txy;
    tf.tile_z = ts.tz;
    tf.nt[0] = nt[0]; tf.nt[1] = nt[1]; tf.nt[2] = nt[2];
    return MDH_OK;
}

} // namespace mdh

MDH_WARM_UNIT(neighbor_tiled)
