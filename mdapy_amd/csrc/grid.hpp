// grid.hpp — the rc-wide cell grid shared by the neighbor build, kNN and RDF kernels.
#pragma once
#include "common.hpp"
#include <cmath>

namespace mdh {

// mode 0: cells of width rc anchored at the origin, last cell absorbs the remainder (neighbor.cpp:29-62)
// mode 1: nc equal cells across the box, floor((x-o)/L*nc)  (radial_distribution_function.cpp:109-141; also kNN)
struct Grid {
    int nc[3];
    int64_t ncell;
    double rc_inv;
    int mode;
};

// Image codes (orthogonal boxes).  An atom handed in outside the box on a periodic axis has raw = wrapped + m L, m a whole number
// of box lengths (an unwrapped trajectory: atoms that have diffused through the faces); the cell grid sees the wrapped atom, the
// reference's distances are computed from the raw one (neighbor.cpp:164-166) and then folded by L * floor(d / L + 0.5)
// (box.h:120-124).  The tile kernels take that image number from codes instead of a division per pair:
//   atom code      (m + 15) per axis, 5 bits each, m in [-14, 14] (more: the build's flags[0], the thread-per-atom kernel)
//   cell code      (n + 1) per axis, 2 bits each: the candidate's cell seen from the tile, n in {-1, 0, 1}
//   combined code  (n + m + 16) per axis, 5 bits each — fits the 16-bit LDS entry of a staged atom
namespace img {
constexpr int MAX_M = 14;
constexpr int ATOM_NEUTRAL = 15 | (15 << 5) | (15 << 10);
constexpr int CELL_NEUTRAL = 1 | (1 << 2) | (1 << 4);
constexpr int NEUTRAL = 16 | (16 << 5) | (16 << 10); // combined code "no shift"
__host__ __device__ __forceinline__ int combine(int cc, int ca)
{
    return ((cc & 3) + (ca & 31)) | ((((cc >> 2) & 3) + ((ca >> 5) & 31)) << 5) | ((((cc >> 4) & 3) + ((ca >> 10) & 31)) << 10);
}
__host__ __device__ __forceinline__ int axis(int code, int d) { return ((code >> (5 * d)) & 31) - 16; } // image number of a combined code
} // namespace img

// device buffers produced by build_cell_grid (owned by the Scope that built them)
struct CellGrid {
    Grid g;
    int win_lo = 0, win_hi = 0; // planes [win_lo, win_hi) of axis 0 the grid was built over (a promised window, neighbor.hip); 0, 0: all
    int cen_lo = 0, cen_hi = 0; // planes [cen_lo, cen_hi) of axis 0 whose atoms want rows (mdh_hint_centre_window: a slab's own atoms; the rest of the window is halo); 0, 0: all
    mutable bool flags_fresh = false; // flags[] were zeroed by build_cell_grid and nobody has used them yet (the first neighbor pass skips its own memset: every hipMemsetAsync is a 5 us launch)
    int *cell_start; // [ncell+1] exclusive prefix of the per-cell populations
    int *order;      // [N] atom ids, cell-major, DESCENDING id inside a cell
    double *xs, *ys, *zs; // [N] raw positions in `order`
    // device flags written while binning:
    //   flags[0] != 0 : some atom's raw coordinate differs from its wrapped one by more than `slack` on a
    //                   periodic axis (unwrapped input) -> per-cell image shifts are not valid
    //   flags[1]      : largest cell population
    int *flags;
    unsigned short *mvs; // [N] per-atom image code (raw vs wrapped coordinate; img::), in `order`
    // [N] the same five values as one 32-byte record per atom (neighbor builds): the tile kernel stages an atom with two
    // 16-byte requests instead of five small ones
    struct Packed { double x, y, z; int id, code; };
    Packed *pk; // when set, xs / ys / zs / mvs are NOT filled (ensure_unpacked() does that for the kernels that want them)
    // INDIRECT (neighbor builds of input that comes in some spatial order): no sorted copy of the atoms at all — atom q of the cell
    // order is atom order[q] of the CALLER's arrays (ix, iy, iz; image code imv[order[q]] when flags[4] says that any atom has
    // one).  pk, xs, ys, zs, mvs are all null; the kernels read through `order` (the tile kernel stages a cell's atoms with four
    // small gathers per atom where the record took two 16-byte reads — and the grid build is one pass over 56 B per atom shorter)
    const double *ix = nullptr, *iy = nullptr, *iz = nullptr;
    const unsigned short *imv = nullptr;
};
// a cell-sorted atom, from either representation
struct SortedView {
    const double *xs, *ys, *zs;
    const int *order;
    const CellGrid::Packed *pk;
    bool indirect; // xs, ys, zs are the caller's arrays, read at order[q] (CellGrid::ix)
    __device__ __forceinline__ void get(int64_t q, double &x, double &y, double &z, int &id) const
    {
        if (pk) { const CellGrid::Packed r = pk[q]; x = r.x; y = r.y; z = r.z; id = r.id; }
        else if (indirect) { id = order[q]; x = xs[id]; y = ys[id]; z = zs[id]; }
        else { x = xs[q]; y = ys[q]; z = zs[q]; id = order[q]; }
    }
    __device__ __forceinline__ int id_of(int64_t q) const { return pk ? pk[q].id : order[q]; }
};
inline SortedView view_of(const CellGrid &cg)
{
    if (!cg.pk && !cg.xs && cg.ix) return SortedView{cg.ix, cg.iy, cg.iz, cg.order, nullptr, true};
    return SortedView{cg.xs, cg.ys, cg.zs, cg.order, cg.pk, false};
}
int ensure_unpacked(Scope &sc, CellGrid &cg, int64_t N); // xs, ys, zs, mvs from pk (no-op when they exist)

// neighbor_tiled.hip: the round-1 LDS-tiled kernel (double-precision scan, any run length); serves the cells too full for neighbor_lane.hip
struct TiledPlan {
    int tile;        // cells per tile edge in x and y; 0 = not applicable
    int tile_z;      // cells per tile edge in z
    bool cellshift;  // every periodic axis has >= 7 cells: per-cell image shifts may replace the minimum-image search
    bool full;       // every 4x4x4 block of cells holds atoms (last known occupancy): all tiles are live
    int64_t occupied; // cells of the occupied region (last known)
};
// what the LDS-tiled kernel leaves to the thread-per-atom kernel: tiles whose halo did not fit in LDS
// (flag[t] != 0; *any counts them).  flag == nullptr: no filtering (the thread-per-atom kernel does everything).
struct TileFilter {
    const unsigned char *flag = nullptr;
    const int *any = nullptr;
    const int *moved = nullptr; // != nullptr and *moved != 0: the tiled kernel stood down (unwrapped input), take every atom
    const int *list = nullptr;  // != nullptr: ids of the flagged tiles, *any of them (k_neighbor_tiles walks the list; the flag scan over all atoms is skipped)
    int list_cap = 0;
    int *cna_todo = nullptr;    // != nullptr (fused neighbor + fixed CNA): the mop-up kernels list the atoms they take here, count first
    int tile = 1, tile_z = 1;
    int nt[3] = {1, 1, 1};
};
// occupied_cells: cells that hold atoms (occupied_cells_hint); 0 = assume all of them
TiledPlan plan_tiled(const DBox &b, const Grid &g, int64_t N, int64_t M, int64_t occupied_cells);
int occupied_cells_hint(Scope &sc, const CellGrid &cg, int64_t N, int64_t *occupied);
int launch_neighbor_tiled(Scope &sc, const CellGrid &cg, const TiledPlan &plan, int64_t N, const DBox &b, double rc,
                          int *verlet, double *dist, int *nn, int64_t M, bool fill_pads, TileFilter &tf);

// neighbor_lane.hip: LDS tiles, one thread per centre atom, single-precision pruning (see the file header)
struct GridStats {
    static constexpr int NBIN = 99; // v[0] = cells of the occupied region; v[1 + len] = 3-cell z-runs of that length (98: longer than 96)
    int v[NBIN];
    int last_listed = -1;   // tiles the first pass of the previous build with this (N, grid) listed for the second (-1: not known)
    int *listed_sink = nullptr; // pinned host word the second pass of THIS build writes its count to (device-visible)
};
struct LanePlan {
    int txy, tz;      // tile shape in cells; txy == 0: not applicable
    int cap;          // atoms a tile's halo may hold in LDS
    bool tk8;         // rows of at most 16 slots: one-byte tickets, lean LDS layout, rows written by the centre's lane
    int wgs;          // workgroups per CU the LDS budget was cut for
    int rw;           // rows (centres) a wave works on at a time: 64, fewer for long rows in dense cells
    int nw = 4;       // waves per workgroup: 4, or 8 (tiles of up to 512 halo cells, two workgroups per CU)
    float mid, T;     // single-precision scan: the constant c subtracted from d2 (a little below rc^2) and the width W of the band above it
    int last_listed = -1; int *listed_sink = nullptr; // GridStats: sizes the second pass's grid
    bool full;        // every 4x4x4 block of cells holds atoms (last known statistics): all tiles are live
    int64_t occupied; // cells of the occupied region (last known)
};
int grid_stats_hint(Scope &sc, const CellGrid &cg, int64_t N, GridStats *out);
LanePlan plan_lane(const DBox &b, const Grid &g, int64_t N, int64_t M, const GridStats &gs, double rc, bool fcna, bool count);
int lane_last_listed(); // tiles listed for the slice pass as last seen when a plan was made (mdh_debug_counters)
// pattern != nullptr: the fixed-cutoff CNA label (cna.cpp:429-506, same rc) of every centre the kernel takes is written too
int launch_neighbor_lane(Scope &sc, const CellGrid &cg, const LanePlan &plan, int64_t N, const DBox &b, double rc,
                         int *verlet, double *dist, int *nn, int64_t M, bool fill_pads, bool count, int *max_count,
                         TileFilter &tf, int *pattern = nullptr);
// cna.hip: fixed-cutoff CNA from finished lists on the caller's stream — of all atoms, or of the atoms listed in todo
// (todo[0] = count, device side) with the reference expression
// done != nullptr: todo sits in a kept block (Scope::KEEP_TODO) — the kernel that walks the list clears its counters when it leaves
void launch_fcna_all(hipStream_t st, const DBox &b, const double *x, const double *y, const double *z, int64_t N, const int *verlet,
                     int64_t M, const int *nn, int *pattern, double rc, int *todo, int *done = nullptr, const Pos4 *pos = nullptr, const int *use_pos = nullptr);
void launch_fcna_listed(hipStream_t st, const DBox &b, const double *x, const double *y, const double *z, int64_t N, const int *verlet,
                        int64_t M, const int *nn, int *pattern, double rc, int *todo, int *done = nullptr);

__host__ __device__ __forceinline__ int pmod(int a, int n) // neighbor.cpp:18-22
{
    int r = a % n;
    return r < 0 ? r + n : r;
}

// cell coordinates of an already wrapped position (neighbor.cpp:29-62)
template <bool TRI>
__device__ __forceinline__ void cell_coords(const DBox &b, const Grid &g, double x, double y, double z, int &c0,
                                            int &c1, int &c2)
{
    double f0, f1, f2;
    if (TRI) {
        double dx = x - b.o[0], dy = y - b.o[1], dz = z - b.o[2];
        double nx = dx * b.hi[0] + dy * b.hi[3] + dz * b.hi[6];
        double ny = dx * b.hi[1] + dy * b.hi[4] + dz * b.hi[7];
        double nz = dx * b.hi[2] + dy * b.hi[5] + dz * b.hi[8];
        if (g.mode == 0) {
            f0 = floor(nx * b.thick[0] * g.rc_inv);
            f1 = floor(ny * b.thick[1] * g.rc_inv);
            f2 = floor(nz * b.thick[2] * g.rc_inv);
        } else {
            f0 = floor(nx * g.nc[0]);
            f1 = floor(ny * g.nc[1]);
            f2 = floor(nz * g.nc[2]);
        }
    } else if (g.mode == 0) {
        f0 = floor((x - b.o[0]) * g.rc_inv);
        f1 = floor((y - b.o[1]) * g.rc_inv);
        f2 = floor((z - b.o[2]) * g.rc_inv);
    } else {
        f0 = floor((x - b.o[0]) / b.h[0] * g.nc[0]);
        f1 = floor((y - b.o[1]) / b.h[4] * g.nc[1]);
        f2 = floor((z - b.o[2]) / b.h[8] * g.nc[2]);
    }
    // static_cast<int> then clamp to [0, nc-1] (neighbor.cpp:58-61); done in
    // floating point first so that huge values saturate instead of wrapping.
    f0 = fmin(fmax(f0, 0.0), (double)(g.nc[0] - 1));
    f1 = fmin(fmax(f1, 0.0), (double)(g.nc[1] - 1));
    f2 = fmin(fmax(f2, 0.0), (double)(g.nc[2] - 1));
    c0 = (int)f0; // NaN -> fmax(NaN,0)=0
    c1 = (int)f1;
    c2 = (int)f2;
}

// fills cg.g for the cutoff neighbor search: nc = max(floor(thickness/rc), 3) (neighbor.cpp:203-206)
int neighbor_grid_dims(const DBox &b, double rc, Grid &g);

// Bins the atoms into cg.g (dims/mode set by the caller) and produces cell_start/order/xs,ys,zs.
//   wrap_first : wrap a position into the primary cell before binning when any axis is periodic
//   sort_desc  : order every cell's atoms by descending id (reference row order); otherwise the
//                order inside a cell is whatever the atomic counters produced
int build_cell_grid(Scope &sc, const double *x, const double *y, const double *z, int64_t N, const DBox &b,
                    bool wrap_first, bool sort_desc, CellGrid &cg, const int64_t *sort_key = nullptr, bool packed = false,
                    bool scattered = false);

// neighbor.hip: out[0..n] = exclusive prefix sums of in[0..n), out[n] = total
int exclusive_scan_u32(Scope &sc, const unsigned *in, int *out, int64_t n);

} // namespace mdh
