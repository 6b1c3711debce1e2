// voronoi.hip — Voronoi cell volume, face count and cavity radius per atom on gfx950.
//
// Replaces src/voronoi.cpp:16-147 (get_voronoi_volume_number_radius / _tri), which hand every atom to voro++
// (extern/voro++).  Here ONE WAVEFRONT builds one cell: the neighbours within a search radius (cell-list neighbor
// build of this library, rows sorted by distance) become half-space constraints in LDS, lane f clips face f out of its
// own plane against the others (voro_core.hpp; polygon in a per-lane LDS stripe), and the wave reduces volume, face
// count and farthest vertex.  A cell is complete when no unseen atom can cut it, i.e. 2 R_max <= search radius; the
// host enlarges the radius (x1.4) and repeats while a device counter reports incomplete cells.
//   volume   = sum_f area_f h_f / 3          faces = non-degenerate faces, walls of open axes included
//   radius   = 2 R_max                       (voro++ keeps cells doubled: sqrt(max_radius_squared()) is twice the
//                                             farthest-vertex distance, and that is what the reference returns)
#include "common.hpp"
#include "grid.hpp"
#include <cstdio>
#include <cstdlib>
#include "voro_core.hpp"
#include "../../include/mdapy_amd.h"

namespace mdh {

int sort_rows_any_tie_order(int *dv, double *dd, int64_t N, int64_t M, void *stream); // neighbor.hip
static int g_listed_passes = 0; // passes of the last call that went over the atoms with open cells only (mdh_debug_counters out4[3])
int voro_listed_passes() { return g_listed_passes; }
static constexpr int VORO_MAXC = 250;  // neighbours one cell may consider
// ... in a pass over the atoms whose cell is still open (few atoms, long rows: the surface of a void looks at 10^3 neighbours
// before the far side closes its cell; 12 KB of polygons + 48 bytes per constraint = 108 KB of LDS, one wave per CU)
static constexpr int VORO_MAXC_LISTED = 2048;
static constexpr int VORO_LANES = 64;

struct PolyLdsV { // vertex i, plane coordinate c (voro_core.hpp voronoi_face_2d) of lane l at [(i*2 + c) * 64 + l]
    static constexpr int CAP = 12;
    double *base;
    __device__ __forceinline__ double get(int i, int c) const { return base[(i * 2 + c) * VORO_LANES]; }
    __device__ __forceinline__ void set(int i, int c, double x) { base[(i * 2 + c) * VORO_LANES] = x; }
};

// optional per-face output of the cell-info call: polygons relative to the atom
struct CellOut {
    int *nv = nullptr;      // (N, W) vertices per face, 0 = no face in this slot
    double *area = nullptr; // (N, W)
    double *vert = nullptr; // (N, W, V, 3)
    int W = 0, V = 0;
    int *need_v = nullptr;  // largest polygon seen when it did not fit V
};

template <bool TRI>
__global__ __launch_bounds__(VORO_LANES) void k_voronoi(const double *__restrict__ x, const double *__restrict__ y,
                                                        const double *__restrict__ z, int64_t N, DBox b,
                                                        const int *__restrict__ verlet, const int *__restrict__ nn, int64_t M,
                                                        double rc, double *__restrict__ volume, int *__restrict__ nfaces,
                                                        double *__restrict__ radius, int *__restrict__ incomplete,
                                                        int *__restrict__ row_id, double *__restrict__ row_dist,
                                                        double *__restrict__ row_area, int W, double a_thr, double r_thr,
                                                        int64_t n_orig, int *__restrict__ max_faces, DBox b0,
                                                        const unsigned char *__restrict__ dropped, CellOut co,
                                                        const int *__restrict__ subset, int *__restrict__ unfinished, int maxc)
{
    // subset != nullptr: a refinement pass — workgroup r builds the cell of atom subset[r] from row r of (verlet, nn), which
    // were made for the listed atoms only; unfinished: the atoms whose cell is still open are listed for the next pass
    const int64_t row = blockIdx.x;
    const int64_t i = subset ? subset[row] : row;
    const int lane = threadIdx.x;
    if (dropped && dropped[i % n_orig]) { // not in the reference's container (see k_mark_outside): no cell, no row
        if (lane == 0) {
            volume[i] = 0.0;
            nfaces[i] = 0;
            radius[i] = 0.0;
        }
        if (row_id && i < n_orig)
            for (int slot = lane; slot < W; slot += VORO_LANES) {
                const int64_t o = i * (int64_t)W + slot;
                row_id[o] = -1;
                row_dist[o] = 10000.0;
                row_area[o] = 0.0;
            }
        return;
    }
    // LDS of the wave, sized by the launch for the rows it was given (ncap = min(M, VORO_MAXC) + 6 constraints): polygons 12 KB
    // + 48 bytes per constraint — 16 KB and eight waves per CU at the usual 80-neighbour rows (36.9 KB and four waves with
    // 16-vertex polygons in space and tables for 256 constraints)
    extern __shared__ __attribute__((aligned(16))) double voro_lds[];
    const int ncap = (int)(M < maxc ? M : maxc) + 6; // maxc: VORO_MAXC, or VORO_MAXC_LISTED in the passes over the open cells only
    double *poly_lds = voro_lds;                                         // [CAP * 2 * 64]
    double (*nrm)[3] = reinterpret_cast<double (*)[3]>(poly_lds + PolyLdsV::CAP * 2 * VORO_LANES); // [ncap][3]
    double *off = reinterpret_cast<double *>(nrm + ncap), *dist = off + ncap; // [ncap] each
    double *farea = dist + ncap;                                         // [ncap] area of face f (0: no face)
    const double xi = x[i], yi = y[i], zi = z[i];
    // rows are sorted by distance: a crowded atom uses its VORO_MAXC nearest neighbours and is complete within THEIR reach
    const int n = min(min(nn[row], (int)M), maxc);
    const double big = 4 * rc;
    // constraints 0..5: walls of open axes (orthogonal boxes), otherwise the bounding cube
    if (lane < 6) {
        const int a = lane >> 1;
        const bool up = (lane & 1) == 0;
        double o = big;
        if (!TRI && !b.pbc[a]) {
            const double p = (a == 0 ? xi : (a == 1 ? yi : zi)) - b.o[a];
            const double len = a == 0 ? b.h[0] : (a == 1 ? b.h[4] : b.h[8]);
            o = up ? len - p : p;
        }
        nrm[lane][0] = nrm[lane][1] = nrm[lane][2] = 0.0;
        nrm[lane][a] = up ? 1.0 : -1.0;
        off[lane] = o;
        dist[lane] = o;
    }
    for (int c = lane; c < n; c += VORO_LANES) {
        const int j = verlet[row * M + c];
        double dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
        pbc<TRI>(b, dx, dy, dz);
        const double d2 = dx * dx + dy * dy + dz * dz;
        nrm[6 + c][0] = dx; nrm[6 + c][1] = dy; nrm[6 + c][2] = dz;
        off[6 + c] = 0.5 * d2;
        dist[6 + c] = 0.5 * sqrt(d2);
    }
    __syncthreads();
    const int nc = n + 6;
    double vol = 0.0, mr2 = 0.0, asum = 0.0;
    int nf = 0;
    int co_base = 0; // faces of this cell already written to the cell-info rows
    for (int f0 = 0; f0 < nc; f0 += VORO_LANES) {
        const int f = f0 + lane;
        bool have = false;
        voroc::FaceResult2 r;
        r.area = 0.0; r.maxr2 = 0.0; r.overflow = false; r.nv = 0;
        PolyLdsV fast{poly_lds + lane};
        ptmc::PolyLocal slow;
        bool in_slow = false;
        if (f < nc) {
            farea[f] = 0.0;
            if (!(f < 6 && (TRI || b.pbc[f >> 1]))) { // the bounding cube is not a face
                r = voroc::voronoi_face_2d(fast, f, nc, nrm, off, dist, 6, big);
                if (r.overflow) { // a face with more than 12 vertices: private storage holds 28
                    r = voroc::voronoi_face_2d(slow, f, nc, nrm, off, dist, 6, big);
                    in_slow = true;
                }
                if (!r.overflow && (in_slow ? voroc::face_exists(r, slow, dist[f]) : voroc::face_exists(r, fast, dist[f]))) {
                    vol += r.area * dist[f] / 3.0;
                    ++nf;
                    mr2 = fmax(mr2, r.maxr2);
                    farea[f] = r.area;
                    asum += r.area;
                    have = true;
                }
            }
        }
        if (co.nv && i < n_orig) { // polygon of every face, walls included, in constraint order (walls, then nearest first)
            const unsigned long long m = __ballot(have);
            if (have) {
                const int slot = co_base + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < co.W) {
                    const int64_t o = i * (int64_t)co.W + slot;
                    co.nv[o] = r.nv;
                    co.area[o] = r.area;
                    if (r.nv > co.V) atomicMax(co.need_v, r.nv);
                    for (int c = 0; c < r.nv && c < co.V; ++c) {
                        double p3[3];
                        if (in_slow) r.vertex(slow, c, p3); else r.vertex(fast, c, p3);
                        for (int d = 0; d < 3; ++d) co.vert[(o * co.V + c) * 3 + d] = p3[d];
                    }
                }
            }
            co_base += __popcll(m);
        }
    }
    if (co.nv && i < n_orig) // slots past the last face (also those of an earlier attempt with a smaller search radius)
        for (int slot = co_base + lane; slot < co.W; slot += VORO_LANES) {
            co.nv[i * (int64_t)co.W + slot] = 0;
            co.area[i * (int64_t)co.W + slot] = 0.0;
        }
    // wave reduction (fixed butterfly order: deterministic)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        vol += __shfl_xor(vol, d, 64);
        asum += __shfl_xor(asum, d, 64);
        nf += __shfl_xor(nf, d, 64);
        mr2 = fmax(mr2, __shfl_xor(mr2, d, 64));
    }
    if (lane == 0) {
        const double rmax = sqrt(mr2);
        volume[i] = vol;
        nfaces[i] = nf;
        radius[i] = 2.0 * rmax;
        const double reach = nn[row] > maxc ? 2.0 * dist[6 + n - 1] : rc; // every atom closer than `reach` has been seen
        // no face at all: nothing within the search radius cut the bounding cube (a lone atom of a tiny periodic cell) —
        // the cell is not known yet, whatever the vertex distances of an empty face list say
        if (2.0 * rmax > reach || nn[row] > M || nf == 0) {
            const int slot = atomicAdd(incomplete, 1);
            if (unfinished) unfinished[slot] = (int)i;
        } else if (max_faces && i < n_orig) {
            // (only a cell that is COMPLETE in this pass counts: a cell still open has more faces now than it will have at a wider
            // radius, and the width the rows are handed over at is max(neighbor_number) of the finished cells, voronoi.cpp:307-447)
            atomicMax(max_faces, nf);
        }
    }
    // Voronoi neighbour rows (src/voronoi.cpp:307-447): the faces shared with atoms — walls have no partner — whose area
    // exceeds max(a_thr, r_thr * total face area), nearest first, padded with -1 / 10000 / 0
    if (row_id && i < n_orig) {
        __syncthreads();
        double amin = a_thr > 0 ? a_thr : 0.0;
        if (r_thr > 0) amin = asum * r_thr;
        if (a_thr > amin) amin = a_thr;
        int base = 0;
        for (int f0 = 6; f0 < nc; f0 += VORO_LANES) {
            const int f = f0 + lane;
            const bool keep = f < nc && farea[f] > 0.0 && farea[f] > amin;
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < W) {
                    const int64_t o = i * (int64_t)W + slot;
                    const int64_t j = verlet[row * M + (f - 6)] % n_orig; // image of a replicated system -> original atom
                    row_id[o] = (int)j;
                    // the reference reports the MINIMUM-IMAGE distance of the pair in the caller's box (src/voronoi.cpp:419-424,
                    // :277-282), also for a face that a thin box makes the cell share with a farther image of j
                    double ddx = x[j] - xi, ddy = y[j] - yi, ddz = z[j] - zi;
                    pbc<TRI>(b0, ddx, ddy, ddz);
                    row_dist[o] = sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
                    row_area[o] = farea[f];
                }
            }
            base += __popcll(m);
        }
        for (int slot = base + lane; slot < W; slot += VORO_LANES) {
            const int64_t o = i * (int64_t)W + slot;
            row_id[o] = -1;
            row_dist[o] = 10000.0;
            row_area[o] = 0.0;
        }
    }
}



// The reference stores an atom in its voro++ container only if, on every OPEN axis, its block index
//   step_int((x - origin) * (1 / (L / n)))   (extern/voro++/src/container_3d.cc:439-456, v_base_3d.cc:21-22)
// lies in [0, n), n = int(L * cbrt(N / (4.6 V)) + 1) (src/voronoi.cpp:36-44); other atoms get no cell (their outputs keep
// the caller's zeros) and cut nobody's cell.  Such atoms are flagged and moved far away along that axis, out of everybody's reach.
__global__ void k_mark_outside(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z, int64_t N,
                               DBox b, int n0, int n1, int n2, double *__restrict__ ox, double *__restrict__ oy,
                               double *__restrict__ oz, unsigned char *__restrict__ dropped)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    double p[3] = {x[i], y[i], z[i]};
    const int nb[3] = {n0, n1, n2};
    bool out = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (b.pbc[a])
            continue;
        const double len = b.h[a * 4];
        const double sp = 1.0 / (len / (double)nb[a]);
        const double t = (p[a] - b.o[a]) * sp;
        const bool inside = !(t < 0) && t < 2147483647.0 && (int)t < nb[a]; // step_int: int(t) - 1 for t < 0
        if (!inside && !out) {
            out = true;
            p[a] = b.o[a] + 1.0e4 * len + 1.0e4;
        }
    }
    ox[i] = p[0]; oy[i] = p[1]; oz[i] = p[2];
    dropped[i] = out ? 1 : 0;
}

// images of the atoms along the periodic axes flagged in rep[]: image (a,b,c) of atom i at index ((a*ny + b)*nz + c)*N + i,
// the (0,0,0) image first — so the first N rows of every result belong to the original atoms
__global__ void k_replicate(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z, int64_t N,
                            int ny, int nz, int64_t total, const double *__restrict__ h9, double *__restrict__ ox,
                            double *__restrict__ oy, double *__restrict__ oz)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total)
        return;
    const int64_t i = t % N, img = t / N;
    const int c = (int)(img % nz), bq = (int)((img / nz) % ny), a = (int)(img / ((int64_t)nz * ny));
    ox[t] = x[i] + a * h9[0] + bq * h9[3] + c * h9[6];
    oy[t] = y[i] + a * h9[1] + bq * h9[4] + c * h9[7];
    oz[t] = z[i] + a * h9[2] + bq * h9[5] + c * h9[8];
}

// Rows of the LISTED atoms only, for the refinement passes: one wavefront per listed atom walks the 27 cells of an rc-wide
// grid in the reference's order (neighbor.cpp:147-151; inside a cell as the sorted arrays have it) and appends what lies
// within rc — the same candidates, in the same order, with the same distance expression as mdh_build_neighbor gives that atom
// (raw x[j] - wrapped x[i], minimum image, sqrt).  rows == nullptr: counts only.
template <bool TRI>
__global__ __launch_bounds__(64) void k_rows_of_listed(SortedView sv, const int *__restrict__ cell_start,
                                                       const double *__restrict__ x, const double *__restrict__ y,
                                                       const double *__restrict__ z, DBox b, Grid g, double rc,
                                                       const int *__restrict__ listed, int *__restrict__ rows,
                                                       double *__restrict__ dists, int *__restrict__ counts, int64_t M,
                                                       int *__restrict__ max_count)
{
    const int64_t r = blockIdx.x;
    const int i = listed[r];
    const int lane = threadIdx.x;
    double xi = x[i], yi = y[i], zi = z[i];
    if (b.anypbc)
        wrap<TRI>(b, xi, yi, zi); // neighbor.cpp:139-142
    int c0, c1, c2;
    cell_coords<TRI>(b, g, xi, yi, zi, c0, c1, c2);
    const double rcsq = rc * rc;
    int cnt = 0;
    for (int a = c0 - 1; a <= c0 + 1; ++a)
        for (int bb = c1 - 1; bb <= c1 + 1; ++bb)
            for (int cc = c2 - 1; cc <= c2 + 1; ++cc) {
                const int64_t cell = ((int64_t)pmod(a, g.nc[0]) * g.nc[1] + pmod(bb, g.nc[1])) * g.nc[2] + pmod(cc, g.nc[2]);
                const int s = cell_start[cell], e = cell_start[cell + 1];
                for (int q0 = s; q0 < e; q0 += 64) {
                    const int q = q0 + lane;
                    bool hit = false;
                    int j = -1;
                    double d2 = 0.0;
                    if (q < e) {
                        double xq, yq, zq;
                        sv.get(q, xq, yq, zq, j);
                        double dx = xq - xi, dy = yq - yi, dz = zq - zi;
                        pbc<TRI>(b, dx, dy, dz);
                        d2 = dx * dx + dy * dy + dz * dz;
                        hit = j != i && d2 <= rcsq;
                    }
                    const unsigned long long m = __ballot(hit);
                    if (hit && rows) {
                        const int slot = cnt + __popcll(m & ((1ull << lane) - 1ull));
                        if (slot < M) {
                            rows[r * M + slot] = j;
                            dists[r * M + slot] = sqrt(d2);
                        }
                    }
                    cnt += __popcll(m);
                }
            }
    if (rows) // pads as mdh_build_neighbor(fill_pads) leaves them
        for (int slot = cnt + lane; slot < M; slot += 64) {
            rows[r * M + slot] = -1;
            dists[r * M + slot] = rc + 1.0;
        }
    if (lane == 0) {
        counts[r] = cnt;
        if (max_count) atomicMax(max_count, cnt);
    }
}

// cells of all `N` atoms of one (possibly replicated) system; 1 = some cells reach beyond half a period (caller replicates)
static int voronoi_solve(void *stream, const double *dx, const double *dy, const double *dz, int64_t N, const double *box9,
                         const double *origin3, const int *boundary3, double *dvol, int *dnf, double *drad, int *dnn, int *dflag,
                         bool *too_small, int *row_id, double *row_dist, double *row_area, int W, double a_thr, double r_thr,
                         int64_t n_orig, int *dmaxf, const double *box9_orig, const unsigned char *dropped, const CellOut &co)
{
    *too_small = false;
    DBox b, b0;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    MDH_TRY(make_box(b0, box9_orig, origin3, boundary3));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double vol = fabs(b.h[0] * (b.h[4] * b.h[8] - b.h[5] * b.h[7]) - b.h[1] * (b.h[3] * b.h[8] - b.h[5] * b.h[6]) +
                            b.h[2] * (b.h[3] * b.h[7] - b.h[4] * b.h[6]));
    double rc_cap = 1.0e300; // minimum-image rows hold every atom once: the search radius must stay below half a period
    for (int a = 0; a < 3; ++a)
        if (b.pbc[a]) rc_cap = fmin(rc_cap, 0.5 * b.thick[a] * (1.0 - 1e-9));
    double rc = fmin(2.2 * cbrt(vol / (double)N), rc_cap);
    Scope keep(stream);
    int *lists[2] = {keep.alloc_n<int>((size_t)N), keep.alloc_n<int>((size_t)N)}; // atoms whose cell is still open, this pass / the next
    if (keep.failed())
        return keep.error();
    for (int attempt = 0; attempt < 16; ++attempt) {
        int maxc = 0;
        if (std::getenv("MDH_VORO_DEBUG")) fprintf(stderr, "voronoi full pass: N %lld rc %g cap %g\n", (long long)N, rc, rc_cap);
        MDH_TRY(mdh_neighbor_count(dx, dy, dz, N, box9, origin3, boundary3, rc, dnn, &maxc, MDH_DEVICE, stream));
        const int64_t M = maxc > 0 ? maxc : 1;
        if ((double)N * (double)M > 4.0e8) { // 4.8 GB of rows: a homogeneous system needs ~60 entries per atom
            set_error("mdh_voronoi_volume_number_radius: the search list would exceed 4e8 entries (extremely inhomogeneous system, e.g. a cluster in a periodic vacuum)");
            return MDH_ERR_ARG;
        }
        Scope inner(stream);
        int *dv = inner.alloc_n<int>((size_t)(N * M));
        double *dd = inner.alloc_n<double>((size_t)(N * M));
        if (inner.failed())
            return inner.error();
        MDH_TRY(mdh_build_neighbor(dx, dy, dz, N, box9, origin3, boundary3, rc, dv, dd, dnn, M, 1, MDH_DEVICE, stream));
        MDH_TRY(sort_rows_any_tie_order(dv, dd, N, M, stream));
        MDH_HIP(hipMemsetAsync(dflag, 0, sizeof(int), st));
        if (dmaxf) MDH_HIP(hipMemsetAsync(dmaxf, 0, sizeof(int), st));
        {
            ProfRange pr("k_voronoi", st);
            const size_t voro_lds_bytes = ((size_t)PolyLdsV::CAP * 2 * VORO_LANES + (size_t)((M < VORO_MAXC ? M : VORO_MAXC) + 6) * 6) * sizeof(double);
            if (b.tri)
                hipLaunchKernelGGL(k_voronoi<true>, dim3((unsigned)N), dim3(VORO_LANES), voro_lds_bytes, st, dx, dy, dz, N, b, dv, dnn, M, rc, dvol, dnf, drad, dflag, row_id, row_dist, row_area, W, a_thr, r_thr, n_orig, dmaxf, b0, dropped, co, (const int *)nullptr, lists[0], VORO_MAXC);
            else
                hipLaunchKernelGGL(k_voronoi<false>, dim3((unsigned)N), dim3(VORO_LANES), voro_lds_bytes, st, dx, dy, dz, N, b, dv, dnn, M, rc, dvol, dnf, drad, dflag, row_id, row_dist, row_area, W, a_thr, r_thr, n_orig, dmaxf, b0, dropped, co, (const int *)nullptr, lists[0], VORO_MAXC);
        }
        int bad = 0;
        MDH_HIP(hipMemcpyAsync(&bad, dflag, sizeof(int), hipMemcpyDeviceToHost, st));
        MDH_HIP(hipStreamSynchronize(st));
        if (bad == 0)
            return MDH_OK;
        if (rc >= rc_cap) {
            *too_small = true;
            return MDH_OK;
        }
        rc = fmin(rc * 1.4, rc_cap);
        // A few open cells — the atoms of a surface, of a void's rim — do not send everybody to a wider search: while they are
        // less than a quarter of the atoms, only THEY get rows at the wider radius (k_rows_of_listed) and their cells rebuilt,
        // pass after pass, each pass listing who is still open.  (A slab of 256 k atoms took 47 ms against 9 for the periodic
        // crystal, a free cluster 213 ms, and a triclinic slab — whose open direction is a vacuum three box lengths wide,
        // voronoi.py — was refused for the size of everybody's rows.)
        if (lists[0] && (int64_t)bad * 4 <= N) {
            int cur = 0;
            bool stalled = false; // the previous listed pass closed no cell
            while (bad > 0) {
                const int64_t nl = bad;
                ++g_listed_passes;
                if (std::getenv("MDH_VORO_DEBUG")) fprintf(stderr, "voronoi subset pass: N %lld listed %lld rc %g cap %g\n", (long long)N, (long long)nl, rc, rc_cap);
                CellGrid cg;
                MDH_TRY(neighbor_grid_dims(b, rc, cg.g));
                // (a listed atom looks at 27 cells of N / ncell atoms: surfaces facing a vacuum many cells wide — the widened open
                // direction of a triclinic slab — would test 10^11 pairs before the row limit below stops them)
                if ((double)nl * 27.0 * (double)N / (double)cg.g.ncell > 2.0e10) {
                    set_error("mdh_voronoi_volume_number_radius: the search list would exceed 4e8 entries (extremely inhomogeneous system, e.g. a cluster in a periodic vacuum)");
                    return MDH_ERR_ARG;
                }
                Scope inner(stream);
                MDH_TRY(build_cell_grid(inner, dx, dy, dz, N, b, true, true, cg));
                int *cnts = inner.alloc_n<int>((size_t)nl);
                int *dmax = inner.alloc_n<int>(1);
                if (inner.failed())
                    return inner.error();
                MDH_HIP(hipMemsetAsync(dmax, 0, sizeof(int), st));
                const SortedView sv = view_of(cg);
                if (b.tri) hipLaunchKernelGGL(k_rows_of_listed<true>, dim3((unsigned)nl), dim3(64), 0, st, sv, cg.cell_start, dx, dy, dz, b, cg.g, rc, lists[cur], (int *)nullptr, (double *)nullptr, cnts, (int64_t)0, dmax);
                else hipLaunchKernelGGL(k_rows_of_listed<false>, dim3((unsigned)nl), dim3(64), 0, st, sv, cg.cell_start, dx, dy, dz, b, cg.g, rc, lists[cur], (int *)nullptr, (double *)nullptr, cnts, (int64_t)0, dmax);
                int maxc = 0;
                MDH_HIP(hipMemcpyAsync(&maxc, dmax, sizeof(int), hipMemcpyDeviceToHost, st));
                MDH_HIP(hipStreamSynchronize(st));
                const int64_t M = maxc > 0 ? maxc : 1;
                // (rows wider than the block sort takes are refused; so are rows of more than 2048 candidates — cells that reach past
                // five neighbour shells — once a pass has closed none of its cells: the surface of a slab looking across a vacuum
                // as wide as the slab, which only rows of 10^4..10^5 entries would close, seconds per pass.  A void in a crystal
                // closes some cells with every pass and goes on)
                if ((double)nl * (double)M > 4.0e8 || M > 8192 || (M > 2048 && stalled)) {
                    set_error("mdh_voronoi_volume_number_radius: the search list would exceed 4e8 entries (extremely inhomogeneous system, e.g. a cluster in a periodic vacuum): " +
                              std::to_string((long long)nl) + " cells still open at a search radius of " + std::to_string(rc) + " with rows of " + std::to_string((long long)M) + " candidates");
                    return MDH_ERR_ARG;
                }
                int *dv = inner.alloc_n<int>((size_t)(nl * M));
                double *dd = inner.alloc_n<double>((size_t)(nl * M));
                if (inner.failed())
                    return inner.error();
                if (b.tri) hipLaunchKernelGGL(k_rows_of_listed<true>, dim3((unsigned)nl), dim3(64), 0, st, sv, cg.cell_start, dx, dy, dz, b, cg.g, rc, lists[cur], dv, dd, cnts, M, (int *)nullptr);
                else hipLaunchKernelGGL(k_rows_of_listed<false>, dim3((unsigned)nl), dim3(64), 0, st, sv, cg.cell_start, dx, dy, dz, b, cg.g, rc, lists[cur], dv, dd, cnts, M, (int *)nullptr);
                MDH_TRY(sort_rows_any_tie_order(dv, dd, nl, M, stream));
                MDH_HIP(hipMemsetAsync(dflag, 0, sizeof(int), st));
                {
                    ProfRange pr("k_voronoi", st);
                    const size_t voro_lds_bytes = ((size_t)PolyLdsV::CAP * 2 * VORO_LANES + (size_t)((M < VORO_MAXC_LISTED ? M : VORO_MAXC_LISTED) + 6) * 6) * sizeof(double);
                    if (voro_lds_bytes > 48 * 1024)
                        MDH_HIP(hipFuncSetAttribute(b.tri ? (const void *)k_voronoi<true> : (const void *)k_voronoi<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)voro_lds_bytes));
                    if (b.tri)
                        hipLaunchKernelGGL(k_voronoi<true>, dim3((unsigned)nl), dim3(VORO_LANES), voro_lds_bytes, st, dx, dy, dz, N, b, dv, cnts, M, rc, dvol, dnf, drad, dflag, row_id, row_dist, row_area, W, a_thr, r_thr, n_orig, dmaxf, b0, dropped, co, lists[cur], lists[1 - cur], VORO_MAXC_LISTED);
                    else
                        hipLaunchKernelGGL(k_voronoi<false>, dim3((unsigned)nl), dim3(VORO_LANES), voro_lds_bytes, st, dx, dy, dz, N, b, dv, cnts, M, rc, dvol, dnf, drad, dflag, row_id, row_dist, row_area, W, a_thr, r_thr, n_orig, dmaxf, b0, dropped, co, lists[cur], lists[1 - cur], VORO_MAXC_LISTED);
                }
                MDH_HIP(hipMemcpyAsync(&bad, dflag, sizeof(int), hipMemcpyDeviceToHost, st));
                MDH_HIP(hipStreamSynchronize(st));
                cur = 1 - cur;
                stalled = bad == (int)nl;
                if (bad == 0)
                    return MDH_OK;
                if (rc >= rc_cap) {
                    *too_small = true;
                    return MDH_OK;
                }
                rc = fmin(rc * 1.4, rc_cap);
            }
        }
    }
    set_error("mdh_voronoi_volume_number_radius: search radius did not converge");
    return MDH_ERR_ARG;
}

} // namespace mdh

using namespace mdh;

// shared driver: volumes always; neighbour rows (width W, original atom ids) when row_id != nullptr; *max_faces_host
// receives the largest face count (walls included) of the N atoms
static int voronoi_driver(const double *x, const double *y, const double *z, int64_t N, const double *box9, const double *origin3,
                          const int *boundary3, double *volume, int *nfaces, double *radius, int *row_id, double *row_dist,
                          double *row_area, int W, double a_thr, double r_thr, int *max_faces_host, int space, void *stream,
                          const CellOut &co = CellOut())
{
    g_listed_passes = 0;
    if (N < 0 || N >= 2147483647LL) { set_error("mdh_voronoi_volume_number_radius: invalid N"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    double *dvol = sc.stage(volume, (size_t)N, space, false, true);
    int *dnf = sc.stage(nfaces, (size_t)N, space, false, true);
    double *drad = sc.stage(radius, (size_t)N, space, false, true);
    int *dflag = sc.alloc_n<int>(2);
    int *drid = row_id ? sc.stage(row_id, (size_t)N * (size_t)W, space, false, true) : nullptr;
    double *drd = row_id ? sc.stage(row_dist, (size_t)N * (size_t)W, space, false, true) : nullptr;
    double *dra = row_id ? sc.stage(row_area, (size_t)N * (size_t)W, space, false, true) : nullptr;
    if (sc.failed())
        return sc.error();
    unsigned char *ddrop = nullptr;
    if (!b.tri && !(b.pbc[0] && b.pbc[1] && b.pbc[2])) {
        const double vol = b.h[0] * b.h[4] * b.h[8];
        const double ilscale = pow((double)N / (4.6 * vol), 1 / 3.0); // src/voronoi.cpp:39-43
        const int n0 = (int)(b.h[0] * ilscale + 1), n1 = (int)(b.h[4] * ilscale + 1), n2 = (int)(b.h[8] * ilscale + 1);
        double *sx = sc.alloc_n<double>((size_t)N), *sy = sc.alloc_n<double>((size_t)N), *sz = sc.alloc_n<double>((size_t)N);
        ddrop = sc.alloc_n<unsigned char>((size_t)N);
        if (sc.failed())
            return sc.error();
        hipLaunchKernelGGL(k_mark_outside, dim3(grid_for(N, 256)), dim3(256), 0, st, dx, dy, dz, N, b, n0, n1, n2, sx, sy, sz, ddrop);
        dx = sx; dy = sy; dz = sz;
    }
    // Cells wider than half a period cannot be described by minimum-image rows: the periodic axes that are too thin are
    // replicated (x3 per round, the original atoms first) and the cells of the original atoms are taken from the copy.
    int rep[3] = {1, 1, 1};
    for (int round = 0; round < 4; ++round) {
        const int64_t total = N * rep[0] * rep[1] * rep[2];
        if (total >= 50000000LL) {
            set_error("mdh_voronoi_volume_number_radius: the replicated system would be too large");
            return MDH_ERR_ARG;
        }
        double big9[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) big9[r * 3 + c] = box9[r * 3 + c] * rep[r];
        Scope work(stream);
        const double *px = dx, *py = dy, *pz = dz;
        double *wv = dvol, *wr = drad;
        int *wn = dnf;
        int *dnn = work.alloc_n<int>((size_t)total);
        if (total != N) {
            double *rx = work.alloc_n<double>((size_t)total), *ry = work.alloc_n<double>((size_t)total), *rz = work.alloc_n<double>((size_t)total);
            double *h9 = work.alloc_n<double>(9);
            wv = work.alloc_n<double>((size_t)total); wr = work.alloc_n<double>((size_t)total); wn = work.alloc_n<int>((size_t)total);
            if (work.failed())
                return work.error();
            MDH_HIP(hipMemcpyAsync(h9, box9, sizeof(double) * 9, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_replicate, dim3(grid_for(total, 256)), dim3(256), 0, st, dx, dy, dz, N, rep[1], rep[2], total, h9, rx, ry, rz);
            MDH_HIP(hipStreamSynchronize(st)); // box9 is caller memory
            px = rx; py = ry; pz = rz;
        }
        if (work.failed())
            return work.error();
        bool too_small = false;
        MDH_TRY(voronoi_solve(stream, px, py, pz, total, big9, origin3, boundary3, wv, wn, wr, dnn, dflag, &too_small, drid, drd, dra, W, a_thr,
                              r_thr, N, dflag + 1, box9, ddrop, co));
        if (!too_small) {
            if (total != N) {
                MDH_HIP(hipMemcpyAsync(dvol, wv, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, st));
                MDH_HIP(hipMemcpyAsync(dnf, wn, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice, st));
                MDH_HIP(hipMemcpyAsync(drad, wr, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, st));
                MDH_HIP(hipStreamSynchronize(st));
            }
            if (max_faces_host) {
                MDH_HIP(hipMemcpyAsync(max_faces_host, dflag + 1, sizeof(int), hipMemcpyDeviceToHost, st));
                MDH_HIP(hipStreamSynchronize(st));
            }
            return sc.finish(space);
        }
        // replicate the thinnest periodic axis (and any other that is not at least 1.5 times thicker)
        double tmin = 1.0e300;
        for (int a = 0; a < 3; ++a)
            if (b.pbc[a]) tmin = fmin(tmin, b.thick[a] * rep[a]);
        for (int a = 0; a < 3; ++a)
            if (b.pbc[a] && b.thick[a] * rep[a] < 1.5 * tmin) rep[a] *= 3;
    }
    set_error("mdh_voronoi_volume_number_radius: cells still reach beyond half the replicated box (extremely dilute system)");
    return MDH_ERR_ARG;
}

extern "C" int mdh_voronoi_volume_number_radius(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                                const double *origin3, const int *boundary3, double *volume, int *nfaces,
                                                double *radius, int space, void *stream)
{
    return voronoi_driver(x, y, z, N, box9, origin3, boundary3, volume, nfaces, radius, nullptr, nullptr, nullptr, 0, -1.0, -1.0,
                          nullptr, space, stream);
}

// replaces _voronoi.get_voronoi_neighbor (src/voronoi.cpp:307-447) in two calls, like the exact-width neighbor build:
//   mdh_voronoi_neighbor_count: neighbor_number (N) = faces per cell (walls included, as voro++ reports them) and the
//                               row width = their maximum;
//   mdh_voronoi_neighbor:       verlet / distance / face_area (N, width): the faces shared with atoms whose area exceeds
//                               max(a_thr, r_thr * total face area), NEAREST FIRST (voro++'s face order is an internal
//                               detail of that library), then -1 / 10000 / 0.
extern "C" int mdh_voronoi_neighbor_count(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                          const double *origin3, const int *boundary3, int *neighbor_number, int *width_host,
                                          int space, void *stream)
{
    if (!width_host) { set_error("mdh_voronoi_neighbor_count: width_host is NULL"); return MDH_ERR_ARG; }
    *width_host = 0;
    if (N <= 0)
        return N == 0 ? MDH_OK : MDH_ERR_ARG;
    Scope sc(stream);
    double *v = sc.alloc_n<double>((size_t)N), *r = sc.alloc_n<double>((size_t)N);
    int *nf = sc.stage(neighbor_number, (size_t)N, space, false, true);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    if (sc.failed())
        return sc.error();
    MDH_TRY(voronoi_driver(dx, dy, dz, N, box9, origin3, boundary3, v, nf, r, nullptr, nullptr, nullptr, 0, -1.0, -1.0, width_host,
                           MDH_DEVICE, stream));
    return sc.finish(space);
}

extern "C" int mdh_voronoi_neighbor(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                    const double *origin3, const int *boundary3, double a_face_area_threshold,
                                    double r_face_area_threshold, int *verlet, double *distance, double *face_area, int width,
                                    int space, void *stream)
{
    if (N < 0 || width <= 0) { set_error("mdh_voronoi_neighbor: invalid shape"); return MDH_ERR_ARG; }
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    double *v = sc.alloc_n<double>((size_t)N), *r = sc.alloc_n<double>((size_t)N);
    int *nf = sc.alloc_n<int>((size_t)N);
    if (sc.failed())
        return sc.error();
    // scratch volumes are device arrays whatever the caller's space: run the driver on device copies of the inputs
    if (space == MDH_HOST) {
        const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
        int *dv = sc.stage(verlet, (size_t)N * width, space, false, true);
        double *dd = sc.stage(distance, (size_t)N * width, space, false, true), *da = sc.stage(face_area, (size_t)N * width, space, false, true);
        if (sc.failed())
            return sc.error();
        MDH_TRY(voronoi_driver(dx, dy, dz, N, box9, origin3, boundary3, v, nf, r, dv, dd, da, width, a_face_area_threshold,
                               r_face_area_threshold, nullptr, MDH_DEVICE, stream));
        return sc.finish(space);
    }
    return voronoi_driver(x, y, z, N, box9, origin3, boundary3, v, nf, r, verlet, distance, face_area, width, a_face_area_threshold,
                          r_face_area_threshold, nullptr, MDH_DEVICE, stream);
}

// replaces the geometry of _voronoi.get_cell_info (src/voronoi.cpp:449-540): for every cell its faces (walls of open axes
// included) as polygons.  face_nv (N, W): vertices of face slot s (0: no face), face_area (N, W), face_vert (N, W, V, 3):
// (rows of `from` columns -> rows of `to` <= from columns: the leading columns of every row)
__global__ __launch_bounds__(256) void k_narrow_rows(const int *__restrict__ v, const double *__restrict__ d, const double *__restrict__ a,
                                                     int64_t N, int from, int to, int *__restrict__ vo, double *__restrict__ d_out,
                                                     double *__restrict__ ao)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * to)
        return;
    const int64_t r = e / to;
    const int c = (int)(e - r * to);
    vo[e] = v[r * from + c];
    d_out[e] = d[r * from + c];
    ao[e] = a[r * from + c];
}

// Both of the above from ONE construction of the cells: the rows are made `width` columns wide on the device; if no cell has
// more faces than that, the caller's buffers receive them as (N, *width_out) arrays — *width_out = the width
// mdh_voronoi_neighbor_count reports, the buffers must hold N * width entries — and neighbor_number the face counts; otherwise
// nothing is written but *width_out (> width): call again with that width.
extern "C" int mdh_voronoi_neighbor_rows(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                         const double *origin3, const int *boundary3, double a_face_area_threshold,
                                         double r_face_area_threshold, int *verlet, double *distance, double *face_area, int width,
                                         int *neighbor_number, int *width_out, int space, void *stream)
{
    if (N < 0 || width <= 0 || !width_out) { set_error("mdh_voronoi_neighbor_rows: invalid shape"); return MDH_ERR_ARG; }
    *width_out = 0;
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    double *v = sc.alloc_n<double>((size_t)N), *r = sc.alloc_n<double>((size_t)N);
    int *nf = sc.stage(neighbor_number, (size_t)N, space, false, true);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    int *wv = sc.alloc_n<int>((size_t)N * width);
    double *wd = sc.alloc_n<double>((size_t)N * width), *wa = sc.alloc_n<double>((size_t)N * width);
    if (sc.failed())
        return sc.error();
    int maxf = 0;
    MDH_TRY(voronoi_driver(dx, dy, dz, N, box9, origin3, boundary3, v, nf, r, wv, wd, wa, width, a_face_area_threshold,
                           r_face_area_threshold, &maxf, MDH_DEVICE, stream)); // (maxf is on the host when the driver returns)
    const int w = maxf > 1 ? maxf : 1;
    *width_out = w;
    if (w > width)
        return sc.finish(space);
    int *ov = sc.stage(verlet, (size_t)N * w, space, false, true);
    double *od = sc.stage(distance, (size_t)N * w, space, false, true), *oa = sc.stage(face_area, (size_t)N * w, space, false, true);
    if (sc.failed())
        return sc.error();
    hipLaunchKernelGGL(k_narrow_rows, dim3(grid_for(N * w, 256)), dim3(256), 0, sc.stream(), wv, wd, wa, N, width, w, ov, od, oa);
    return sc.finish(space);
}

// polygon vertices RELATIVE TO THE ATOM in polygon order; W >= the width reported by mdh_voronoi_neighbor_count.
// *need_v_host > V on return: some polygon has more vertices than V (call again with that V); volume / radius as in
// mdh_voronoi_volume_number_radius.  Faces are listed walls first, then nearest neighbour first (voro++ lists them in the
// order of its internal vertex graph); every face is clipped on its own, so a vertex shared by three faces appears in
// each of them with coordinates equal to rounding.  Host (numpy) outputs only: this is a small-system call.
extern "C" int mdh_voronoi_cell_info(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                     const double *origin3, const int *boundary3, int W, int V, int *nfaces, int *face_nv,
                                     double *face_area, double *face_vert, double *volume, double *radius, int *need_v_host,
                                     int space, void *stream)
{
    if (N < 0 || W <= 0 || V < 3 || !need_v_host) { set_error("mdh_voronoi_cell_info: invalid shape"); return MDH_ERR_ARG; }
    *need_v_host = 0;
    if (N == 0)
        return MDH_OK;
    if ((double)N * W * V * 24.0 > 4.0e9) {
        set_error("mdh_voronoi_cell_info: the polygon table would exceed 4 GB (this call is meant for small systems)");
        return MDH_ERR_ARG;
    }
    Scope sc(stream);
    CellOut co;
    co.W = W; co.V = V;
    co.nv = sc.stage(face_nv, (size_t)N * W, space, false, true);
    co.area = sc.stage(face_area, (size_t)N * W, space, false, true);
    co.vert = sc.stage(face_vert, (size_t)N * W * V * 3, space, false, true);
    co.need_v = sc.alloc_n<int>(1);
    double *v = sc.stage(volume, (size_t)N, space, false, true), *r = sc.stage(radius, (size_t)N, space, false, true);
    int *nf = sc.stage(nfaces, (size_t)N, space, false, true);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    if (sc.failed())
        return sc.error();
    hipStream_t st = sc.stream();
    MDH_HIP(hipMemsetAsync(co.nv, 0, sizeof(int) * (size_t)N * W, st));
    MDH_HIP(hipMemsetAsync(co.area, 0, sizeof(double) * (size_t)N * W, st));
    MDH_HIP(hipMemsetAsync(co.vert, 0, sizeof(double) * (size_t)N * W * V * 3, st));
    MDH_HIP(hipMemsetAsync(co.need_v, 0, sizeof(int), st));
    MDH_TRY(voronoi_driver(dx, dy, dz, N, box9, origin3, boundary3, v, nf, r, nullptr, nullptr, nullptr, 0, -1.0, -1.0, nullptr,
                           MDH_DEVICE, stream, co));
    MDH_HIP(hipMemcpyAsync(need_v_host, co.need_v, sizeof(int), hipMemcpyDeviceToHost, st));
    MDH_HIP(hipStreamSynchronize(st));
    return sc.finish(space);
}

// distance column of neighbour rows recomputed as the reference does it: sqrt of box.pbc(x[j] - x[i]) with the box and the
// boundary flags of the CALL (src/voronoi.cpp:277-282 for the triclinic variant, whose positions are the caller's unrotated
// ones while the box is the LAMMPS-aligned one); -1 entries get 10000
template <bool TRI>
__global__ __launch_bounds__(256) void k_row_distance(const int *__restrict__ verlet, int64_t N, int W, const double *__restrict__ x,
                                                      const double *__restrict__ y, const double *__restrict__ z, DBox b,
                                                      double *__restrict__ dist)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * W)
        return;
    const int64_t i = t / W;
    const int j = verlet[t];
    double d = 10000.0;
    if (j >= 0 && j < N) {
        double dx = x[j] - x[i], dy = y[j] - y[i], dz = z[j] - z[i];
        pbc<TRI>(b, dx, dy, dz);
        d = sqrt(dx * dx + dy * dy + dz * dz);
    }
    dist[t] = d;
}

extern "C" int mdh_voronoi_row_distance(const int *verlet, int64_t N, int width, const double *x, const double *y, const double *z,
                                        const double *box9, const double *origin3, const int *boundary3, double *distance, int space,
                                        void *stream)
{
    if (N < 0 || width <= 0) { set_error("mdh_voronoi_row_distance: bad sizes"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const int *dv = sc.stage_in(verlet, (size_t)N * (size_t)width, space);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    double *dd = sc.stage(distance, (size_t)N * (size_t)width, space, false, true);
    if (sc.failed())
        return sc.error();
    const dim3 grid(grid_for(N * width, 256)), block(256);
    if (b.tri)
        hipLaunchKernelGGL(k_row_distance<true>, grid, block, 0, sc.stream(), dv, N, width, dx, dy, dz, b, dd);
    else
        hipLaunchKernelGGL(k_row_distance<false>, grid, block, 0, sc.stream(), dv, N, width, dx, dy, dz, b, dd);
    return sc.finish(space);
}

MDH_WARM_UNIT(voronoi)
