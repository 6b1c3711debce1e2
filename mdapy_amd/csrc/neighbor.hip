// neighbor.hip — cell-list cutoff neighbor search on gfx950.
//
// Replaces src/neighbor.cpp of the reference (build_cell :64-100,
// build_verlet_list :102-187, build_neighbor :351-388, the exact-width variant
// :189-349, sort_verlet_by_distance :745-775, wrap_positions :675-702,
// average_by_neighbor :704-743).
//
// Data layout in HBM:
//   caller:  x,y,z f64[N] (SoA, original atom order); verlet int32[N][M],
//            dist f64[N][M], nn int32[N]   — rows in ORIGINAL atom order.
//   scratch: cell_count u32[ncell] -> cell_start i32[ncell+1] (exclusive scan),
//            rank i32[N] (slot handed out by the atomic bin counter),
//            order i32[N]  (atom ids sorted by cell; inside a cell DESCENDING id,
//                           the order in which the reference's head-inserted
//                           linked list is walked),
//            xs,ys,zs f64[N] (raw positions gathered into cell order, so a
//                           cell's atoms — and the 3 cells of a z-run — are
//                           contiguous and loads are coalesced).
#include "common.hpp"
#include "grid.hpp"
#include "cna_core.hpp"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

namespace mdh {

static int *g_moved_probe = nullptr; // pinned: flags[0] of the last tracked neighbor pass (mdh_debug_track_counters)
// 1: neighbor builds of input in spatial order keep no sorted copy of the atoms (CellGrid::ix); 0: the 32-byte records always
static std::atomic<int> g_indirect{[] { const char *e = std::getenv("MDH_INDIRECT"); return e ? std::atoi(e) : 1; }()};
// Row width of the build the NEXT packed cell grid of this thread is for (0: not known).  Rows of more than 16 slots go to the tile
// kernel's wide instance, which hides the indirect staging's gathers badly (two or three workgroups per CU): such a build keeps
// the records (the 12-nearest search's cutoff build, 4.7 atoms per cell, rows of 24: 2.15 ms with records, 2.22 without)
static thread_local int g_next_rows = 0;
int g_neighbor_variant = 0; // 0 = automatic, 1 = force the thread-per-atom kernel, 2 = force the round-1 LDS-tiled kernel (A/B measurements, tests)

// ----------------------------------------------------------------------------
// cell assignment: wrap, bin, take a slot from the cell's atomic counter
// ----------------------------------------------------------------------------
struct CellPlanes { int p0, p1, p2, p3; int *bad; }; // planes [p0, p1) and [p2, p3) of axis 0 hold every atom (bad == nullptr: not promised; else a pinned host word)
// K atoms per lane: a wave takes 64 * K consecutive atoms as K slices of 64 (slice k: atom base + 64 k + lane, so adjacent lanes
// still hold adjacent atoms and the runs below are found per slice).  The kernel is a chain of dependent memory trips — position
// loads, the returning atomic, the stores — at full occupancy (26-34 VGPRs; 87 % of the wave-cycles waiting,
// profiles/r05_step_counters.json): with K > 1 the loads of a lane's K atoms are in flight together, and so are its K atomics.
template <bool TRI, int K>
__global__ __launch_bounds__(256) void k_assign(const double *__restrict__ x, const double *__restrict__ y,
                                                const double *__restrict__ z, int64_t N, DBox b, Grid g,
                                                int wrap_first, int *__restrict__ cell_id, int *__restrict__ rank,
                                                unsigned *__restrict__ cell_count, unsigned *__restrict__ ctl, unsigned gen,
                                                double slack, unsigned short *__restrict__ mv, CellPlanes win,
                                                CellGrid::Packed *__restrict__ rec, int drop_absent)
{
    const int lane = threadIdx.x & 63;
    const int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 * K) + lane;
    bool moved = false, outside = false, coded = false;
    double xr[K], yr[K], zr[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { // (all loads of the lane first)
        const int64_t i = i0 + 64 * k;
        xr[k] = yr[k] = zr[k] = 0.0;
        if (i < N) { xr[k] = x[i]; yr[k] = y[i]; zr[k] = z[i]; }
    }
    int cells[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int64_t i = i0 + 64 * k;
        int cell = -1 - lane; // lanes past the end: distinct negative values, no run, no atomic
        // an atom whose x is NaN is ABSENT: it takes no cell, appears in nobody's row and gets no row of its own (the unused slots of a
        // decomposed system's fixed-size ghost block, slab.hip k_slab_append_static; the reference has no meaning for such input)
        // (only the neighbor builds — drop_absent — know what to do without such an atom: their kernels walk cells, and their per-atom
        // passes end at the number of atoms binned; every other user of the grid bins a NaN as it always did, into cell 0)
        const bool absent = drop_absent && i < N && xr[k] != xr[k];
        if (absent) {
            cell_id[i] = -1;
            if (mv) mv[i] = (unsigned short)img::ATOM_NEUTRAL;
        }
        if (i < N && !absent) {
            double xi = xr[k], yi = yr[k], zi = zr[k];
            int code = img::ATOM_NEUTRAL; // (m + 15) per axis: raw = wrapped + m*L
            if (wrap_first && b.anypbc) { // neighbor.cpp:88-91
                wrap<TRI>(b, xi, yi, zi);
                if (!TRI) {
                    // whole box lengths between the raw and the wrapped coordinate (an unwrapped trajectory: a few); more than
                    // img::MAX_M of them, or a coordinate that is not wrapped + m L to within `slack`, invalidates the image codes
                    // for this call (flags[0])
                    const double raw[3] = {xr[k], yr[k], zr[k]}, wrp[3] = {xi, yi, zi};
                    code = 0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        double m = 0.0;
                        if (b.pbc[d] && raw[d] != wrp[d]) { // already wrapped (the common case): m = 0, no division
                            m = rint((raw[d] - wrp[d]) / b.h[d * 4]);
                            if (!(fabs(m) <= (double)img::MAX_M) || !(fabs(raw[d] - m * b.h[d * 4] - wrp[d]) <= slack)) { moved = true; m = 0.0; }
                        }
                        code |= ((int)m + 15) << (5 * d);
                    }
                }
            }
            if (mv) mv[i] = (unsigned short)code;
            // scattered input (mdh_spatial_sort): the atom as ONE 32-byte record in input order — the gather then reads one random
            // sector per atom instead of three (x, y, z) or four (the image code)
            if (rec) rec[i] = CellGrid::Packed{xr[k], yr[k], zr[k], (int)i, code};
            coded = coded || code != img::ATOM_NEUTRAL;
            int c0, c1, c2;
            cell_coords<TRI>(b, g, xi, yi, zi, c0, c1, c2);
            cell = (c0 * g.nc[1] + c1) * g.nc[2] + c2; // neighbor.cpp:24-27 (ncell < 2^31 checked on the host)
            if (win.bad && !((c0 >= win.p0 && c0 < win.p1) || (c0 >= win.p2 && c0 < win.p3))) {
                // an atom outside the window of planes the caller promised (mdh_hint_cell_window): the counters out there were
                // never zeroed — it takes no slot and is not scattered (cell -1); the build is reported broken (win.bad), its
                // rows are not to be used, and nothing is written out of bounds
                outside = true;
                cell = -1 - lane;
            }
            cell_id[i] = cell;
        }
        cells[k] = cell;
    }
    // One returning atomic per RUN of adjacent lanes in the same cell instead of one per atom: atoms usually arrive in some
    // spatial order (a lattice builder, a file written cell by cell, a previous sort), so neighbouring lanes share cells;
    // the slot inside a cell is arbitrary anyway (k_sort_cells restores the reference's order).  Unordered input pays a
    // ballot and two shuffles.
    unsigned base[K];
    int first[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { // (the K atomics of the lane in flight together)
        const int cell = cells[k];
        const int prev = __shfl_up(cell, 1, 64);
        const bool head = lane == 0 || prev != cell || cell < 0;
        const unsigned long long heads = __ballot(head);
        first[k] = 63 - __builtin_clzll(heads & ((2ull << lane) - 1ull));                   // head of this lane's run
        const unsigned long long later = lane == 63 ? 0ull : (heads >> (lane + 1));
        const int last = later ? lane + __builtin_ctzll(later) : 63;                        // last lane of the run, as seen from its head
        base[k] = 0;
        if (head && cell >= 0)
            base[k] = atomicAdd(&cell_count[cell], (unsigned)(last - lane + 1));
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const unsigned bs = __shfl(base[k], first[k], 64);
        if (cells[k] >= 0)
            rank[i0 + 64 * k] = (int)(bs + (unsigned)(lane - first[k]));
    }
    // what this kernel finds out about the input goes into generation-stamped control words (no memset per build): the scan
    // that follows turns them into the build's flags[0] (unwrapped input) and flags[4] (image codes present)
    if (__any(moved) && lane == 0)
        ctl[1] = gen;
    if (__any(coded) && lane == 0)
        ctl[2] = gen; // some atom was handed in outside the box: the gather has to read the image codes (else they are all neutral)
    if (__any(outside) && lane == 0)
        *win.bad = 1; // (pinned host memory: read by the next build of the thread / mdh_cell_window_check)
}

// ----------------------------------------------------------------------------
// exclusive prefix sum of the bin counters (three small kernels)
// ----------------------------------------------------------------------------
static constexpr int SCAN_BLOCK = 256;
static constexpr int SCAN_ITEMS = 4; // per thread -> 1024 per block (small inputs); SCAN_ITEMS_BIG for large ones
static constexpr int SCAN_ITEMS_BIG = 32; // 8192 per block: every block takes a ticket from ONE word, ~90 of them per microsecond —
                                          // with 1024 per block the 3 925 tickets of a 4 M-cell grid were 43 of the scan's 62 us
static constexpr int64_t SCAN_BIG_FROM = 1 << 19; // items from which the big blocks are used

__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads = 4 waves); returns exclusive prefix, total via *tot
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned *tot)
{
    __shared__ unsigned wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned inc = wave_incl_scan(v, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned off = 0, t = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < w) off += wsum[k];
        t += wsum[k];
    }
    __syncthreads();
    *tot = t;
    return off + inc - v;
}

// ----------------------------------------------------------------------------
// Single-pass exclusive scan (decoupled look-back: Merrill & Garland, "Single-pass parallel prefix scan with decoupled
// look-back", NVIDIA NVR-2016-002) — ONE launch instead of three: at a few thousand atoms a build is a chain of dependent
// launches of ~4 us each, whatever they do.  A block takes a ticket (so that every predecessor it waits for is already
// running), scans its 1024 items, publishes its total, and wave 0 collects the totals / inclusive prefixes of the blocks
// before it, 64 at a time.  Control words live in a kept block (Scope::KEEP_SCAN): ctl[0] the ticket counter (reset by
// the block that takes the last ticket), ctl[1], ctl[2] the stamps of k_assign, status words from byte 256 on:
// generation (30 bits) | state (2: 1 = block total, 2 = inclusive prefix) | value (32) — a word of an earlier launch
// carries an older generation and reads as "not there yet", so nothing is cleared between launches.
// REZERO: the input is a build's bin counters in a KEEP_ZERO block: every counter is cleared as it is read.
// flags != nullptr: the first block also writes the build's eight device flags (grid.hpp) from the stamps.
// ----------------------------------------------------------------------------
static std::atomic<unsigned> g_scan_gen{0};
static unsigned next_scan_gen()
{
    unsigned g = (++g_scan_gen) & 0x3fffffffu;
    if (g == 0) { // 2^30 launches: old status words could repeat a generation — start over from clean control blocks
        reset_kept_blocks(Scope::KEEP_SCAN);
        g = (++g_scan_gen) & 0x3fffffffu;
    }
    return g;
}

template <bool REZERO, int ITEMS = SCAN_ITEMS>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_onepass(unsigned *__restrict__ in, int *__restrict__ out, int64_t n,
                                                             unsigned *__restrict__ ctl, unsigned gen, int *__restrict__ flags)
{
    constexpr int SCAN_ITEMS = ITEMS; // (shadows the namespace constant: the body below is written for any multiple of four)
    __shared__ unsigned s_blk, s_excl;
    unsigned long long *status = reinterpret_cast<unsigned long long *>(ctl + 64);
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_blk = atomicAdd(&ctl[0], 1u);
    __syncthreads();
    const unsigned blk = s_blk, nblk = gridDim.x;
    if (tid == 0) {
        if (blk == nblk - 1) ctl[0] = 0; // every ticket has been taken: ready for the next launch
        if (blk == 0 && flags) {
            flags[0] = ctl[1] == gen ? 1 : 0; flags[1] = 0; flags[2] = 0; flags[3] = 0;
            flags[4] = ctl[2] == gen ? 1 : 0; flags[5] = 0; flags[6] = 0; flags[7] = 0;
        }
    }
    const int64_t base = ((int64_t)blk * SCAN_BLOCK + tid) * SCAN_ITEMS;
    unsigned v[SCAN_ITEMS], s = 0;
    const bool vec = base + SCAN_ITEMS <= n && ((reinterpret_cast<uintptr_t>(in + base) | reinterpret_cast<uintptr_t>(out + base)) & 15u) == 0;
    if (vec) {
#pragma unroll
        for (int c = 0; c < SCAN_ITEMS / 4; ++c) {
            const uint4 q = reinterpret_cast<const uint4 *>(in + base)[c];
            v[4 * c] = q.x; v[4 * c + 1] = q.y; v[4 * c + 2] = q.z; v[4 * c + 3] = q.w;
        }
        if (REZERO) {
#pragma unroll
            for (int c = 0; c < SCAN_ITEMS / 4; ++c) reinterpret_cast<uint4 *>(in + base)[c] = make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            v[k] = (base + k < n) ? in[base + k] : 0u;
            if (REZERO && base + k < n) in[base + k] = 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) s += v[k];
    unsigned tot;
    unsigned ex = block_excl_scan(s, &tot);
    const unsigned long long stamp = (unsigned long long)gen << 34;
    if (tid == 0) // this block's total (block 0: its inclusive prefix) for the blocks behind it
        __hip_atomic_store(&status[blk], stamp | ((unsigned long long)(blk == 0 ? 2u : 1u) << 32) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        unsigned excl = 0;
        if (blk > 0) {
            int64_t j = (int64_t)blk - 1; // nearest predecessor
            for (;;) {
                const int64_t idx = j - lane;
                unsigned long long st;
                for (;;) {
                    st = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (stamp | (2ull << 32));
                    const bool there = (st >> 34) == gen && ((st >> 32) & 3u) != 0;
                    if (__all(there))
                        break;
                    __builtin_amdgcn_s_sleep(2);
                }
                const unsigned long long pm = __ballot(((st >> 32) & 3u) == 2u);
                const int first = pm ? __builtin_ctzll(pm) : 64; // nearest block that already knows its inclusive prefix
                unsigned val = lane <= first ? (unsigned)(st & 0xffffffffull) : 0u;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d, 64);
                excl += val;
                if (pm)
                    break;
                j -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&status[blk], stamp | (2ull << 32) | (unsigned long long)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    ex += s_excl;
    if (vec) {
#pragma unroll
        for (int c = 0; c < SCAN_ITEMS / 4; ++c) {
            int4 o;
            o.x = (int)ex; o.y = (int)(ex + v[4 * c]); o.z = (int)(ex + v[4 * c] + v[4 * c + 1]); o.w = (int)(ex + v[4 * c] + v[4 * c + 1] + v[4 * c + 2]);
            reinterpret_cast<int4 *>(out + base)[c] = o;
            ex += v[4 * c] + v[4 * c + 1] + v[4 * c + 2] + v[4 * c + 3];
        }
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            if (base + k < n) out[base + k] = (int)ex;
            ex += v[k];
        }
    }
    if (blk == nblk - 1 && tid == 0) out[n] = (int)(s_excl + tot); // grand total
}

static size_t scan_ctl_bytes(int64_t n)
{
    const int64_t per = (int64_t)SCAN_BLOCK * SCAN_ITEMS;
    return 256 + (size_t)((n + per - 1) / per) * 8;
}
// out[0..n] = exclusive prefix of in[0..n), out[n] = total; rezero: clear in[] on the way (bin counters of a kept block)
static void launch_scan_gen(hipStream_t st, unsigned *in, int *out, int64_t n, unsigned *ctl, unsigned gen, bool rezero, int *flags)
{
    const bool big = n >= SCAN_BIG_FROM;
    const int64_t per = (int64_t)SCAN_BLOCK * (big ? SCAN_ITEMS_BIG : SCAN_ITEMS);
    const dim3 grid((unsigned)std::max<int64_t>(1, (n + per - 1) / per)), block(SCAN_BLOCK);
    if (big) {
        if (rezero) hipLaunchKernelGGL((k_scan_onepass<true, SCAN_ITEMS_BIG>), grid, block, 0, st, in, out, n, ctl, gen, flags);
        else hipLaunchKernelGGL((k_scan_onepass<false, SCAN_ITEMS_BIG>), grid, block, 0, st, in, out, n, ctl, gen, flags);
    } else {
        if (rezero) hipLaunchKernelGGL((k_scan_onepass<true, SCAN_ITEMS>), grid, block, 0, st, in, out, n, ctl, gen, flags);
        else hipLaunchKernelGGL((k_scan_onepass<false, SCAN_ITEMS>), grid, block, 0, st, in, out, n, ctl, gen, flags);
    }
}
static void launch_scan(hipStream_t st, unsigned *in, int *out, int64_t n, unsigned *ctl, bool rezero, int *flags)
{
    launch_scan_gen(st, in, out, n, ctl, next_scan_gen(), rezero, flags);
}

// out[0..n] = exclusive prefix sums of in[0..n) (out[n] = total); for other translation units (grid.hpp)
int exclusive_scan_u32(Scope &sc, const unsigned *in, int *out, int64_t n)
{
    unsigned *ctl = static_cast<unsigned *>(sc.alloc_kept(scan_ctl_bytes(n), Scope::KEEP_SCAN));
    if (sc.failed())
        return sc.error();
    launch_scan(sc.stream(), const_cast<unsigned *>(in), out, n, ctl, false, nullptr);
    MDH_HIP(hipGetLastError());
    return MDH_OK;
}

__global__ __launch_bounds__(256) void k_scatter(const int *__restrict__ cell_id, const int *__restrict__ rank,
                                                 const int *__restrict__ cell_start, int *__restrict__ order,
                                                 int64_t N)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    const int c = cell_id[i];
    if (c >= 0) // (< 0: an atom outside a promised cell window, k_assign)
        order[cell_start[c] + rank[i]] = (int)i;
}

// The atomic counters hand out slots in arbitrary order; put every cell's
// atoms into DESCENDING id order (what a walk of the reference's linked list
// sees, neighbor.cpp:97-98) so that rows come out in reference order and the
// result is deterministic.  One thread per cell; cells hold a handful of atoms.
// key != nullptr: descending key[id] instead of descending id (a decomposed system: key = global atom id, so that the rows of
// a slab come out in the order the whole system's rows have)
// Cells of up to eight atoms (nearly all of them at cell width rc) are sorted in registers: the ids, then the keys, loaded as
// one batch, a sorting network with fixed indices, the ids stored back.  The insertion sort below — every comparison a
// dependent read of order[] and, with a key, of key[order[]] — was a chain of 6-10 memory latencies per cell: 32 us of the
// headline build, 77 us of a slab's (random 8-byte key reads).
template <int W, typename K>
__device__ __forceinline__ void sort_cell_net(int *__restrict__ order, int s, int n, const int64_t *__restrict__ key)
{
    int id[W];
    K k[W];
#pragma unroll
    for (int u = 0; u < W; ++u) id[u] = order[s + min(u, n - 1)];
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const K v = key ? (K)key[id[u]] : (K)id[u];
        k[u] = u < n ? v : (sizeof(K) == 8 ? (K)INT64_MIN : (K)INT32_MIN); // pads sink to the end (descending order)
    }
    auto ce = [&](int a, int b) { // k[a] >= k[b] afterwards
        const bool sw = k[a] < k[b];
        const K ka = sw ? k[b] : k[a], kb = sw ? k[a] : k[b];
        const int ia = sw ? id[b] : id[a], ib = sw ? id[a] : id[b];
        k[a] = ka; k[b] = kb; id[a] = ia; id[b] = ib;
    };
    if (W == 4) {
        ce(0, 1); ce(2, 3); ce(0, 2); ce(1, 3); ce(1, 2);
    } else { // Batcher's odd-even merge sort of eight
        ce(0, 1); ce(2, 3); ce(4, 5); ce(6, 7);
        ce(0, 2); ce(1, 3); ce(4, 6); ce(5, 7);
        ce(1, 2); ce(5, 6);
        ce(0, 4); ce(1, 5); ce(2, 6); ce(3, 7);
        ce(2, 4); ce(3, 5);
        ce(1, 2); ce(3, 4); ce(5, 6);
    }
#pragma unroll
    for (int u = 0; u < W; ++u)
        if (u < n) order[s + u] = id[u];
}

// tmp: N ints of scratch indexed like `order` (the rank array of k_assign, free once the atoms are scattered), for cells of
// more than eight atoms without a key: the ids are copied there and every atom is PLACED at the number of larger ids of its
// cell — n^2 independent, cached reads instead of the insertion sort's chain of dependent ones (dense cells, rc = 5 A: 11 atoms
// per cell, up to 50 in the fat last cells: 239 -> 204 us at 10 M atoms, 131 -> 94 us at 3.4 M)
__global__ __launch_bounds__(256) void k_sort_cells(const int *__restrict__ cell_start, int *__restrict__ order,
                                                    int64_t ncell, const int64_t *__restrict__ key, int *__restrict__ tmp)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell)
        return;
    const int s = cell_start[c], e = cell_start[c + 1];
    const int n = e - s;
    if (n <= 1)
        return;
    if (n <= 4) {
        if (key) sort_cell_net<4, int64_t>(order, s, n, key);
        else sort_cell_net<4, int>(order, s, n, nullptr);
        return;
    }
    if (n <= 8) {
        if (key) sort_cell_net<8, int64_t>(order, s, n, key);
        else sort_cell_net<8, int>(order, s, n, nullptr);
        return;
    }
    if (!key && tmp) {
        for (int a = s; a < e; ++a) tmp[a] = order[a];
        for (int a = s; a < e; ++a) {
            const int mine = tmp[a];
            int larger = 0;
            for (int q = s; q < e; ++q) larger += tmp[q] > mine ? 1 : 0;
            order[s + larger] = mine; // (ids are distinct: every slot of the cell is written once)
        }
        return;
    }
    for (int a = s + 1; a < e; ++a) {
        int v = order[a], q = a - 1;
        const int64_t kv = key ? key[v] : (int64_t)v;
        while (q >= s && (key ? key[order[q]] : (int64_t)order[q]) < kv) {
            order[q + 1] = order[q];
            --q;
        }
        order[q + 1] = v;
    }
}

// The same order for grids of many atoms per cell (N / ncell > 6: rc = 5 A in a metal, 11 atoms per cell): EIGHT lanes per cell.
// The cell's ids are staged in LDS (64 per cell; a fuller cell is sorted by its first lane as above), then lane l
// ranks the atoms l, l + 8, ... by counting the larger ids of its cell — n / 8 trips of n LDS reads instead of n * n dependent
// global ones in a single lane (rc = 6 A, 18 atoms per cell, 4 M atoms: 240 us, as long as k_assign, k_scatter and k_gather together).
constexpr int SORT_DENSE_CAP = 64;
__global__ __launch_bounds__(256) void k_sort_cells_dense(const int *__restrict__ cell_start, int *__restrict__ order, int64_t ncell,
                                                          int *__restrict__ tmp)
{
    __shared__ int ids[32 * SORT_DENSE_CAP];
    const int sub = threadIdx.x & 7, lc = threadIdx.x >> 3;
    const int64_t c = (int64_t)blockIdx.x * 32 + lc;
    int s = 0, n = 0;
    if (c < ncell) {
        s = cell_start[c];
        n = cell_start[c + 1] - s;
    }
    const bool staged = n > 1 && n <= SORT_DENSE_CAP, big = n > SORT_DENSE_CAP;
    if (staged)
        for (int a = sub; a < n; a += 8) ids[lc * SORT_DENSE_CAP + a] = order[s + a];
    // a fuller cell — the LAST cell of an axis takes the remainder of the box (neighbor.cpp:58-61) and is up to twice as wide: the
    // corner cell of a 256 k-atom box at rc = 5 A holds 84 atoms where the mean is 12 — goes through the free `rank` array instead,
    // still eight lanes to the cell (one lane, n * n loads: 330 us for that one cell, as long as the rest of the call)
    if (big)
        for (int a = sub; a < n; a += 8) tmp[s + a] = order[s + a];
    __syncthreads(); // (workgroup scope: the copies in LDS and in HBM are visible to the cell's other lanes)
    if (staged) {
        for (int a = sub; a < n; a += 8) {
            const int mine = ids[lc * SORT_DENSE_CAP + a];
            int larger = 0;
            for (int q = 0; q < n; ++q) larger += ids[lc * SORT_DENSE_CAP + q] > mine ? 1 : 0;
            order[s + larger] = mine; // (ids are distinct: every slot of the cell is written once)
        }
    } else if (big) {
        for (int a = sub; a < n; a += 8) {
            const int mine = tmp[s + a];
            int larger = 0;
            for (int q = 0; q < n; ++q) larger += tmp[s + q] > mine ? 1 : 0;
            order[s + larger] = mine;
        }
    }
}

__global__ __launch_bounds__(256) void k_gather(const double *__restrict__ x, const double *__restrict__ y,
                                                const double *__restrict__ z, const int *__restrict__ order,
                                                double *__restrict__ xs, double *__restrict__ ys,
                                                double *__restrict__ zs, int64_t N,
                                                const unsigned short *__restrict__ mv, unsigned short *__restrict__ mvs,
                                                CellGrid::Packed *__restrict__ pk, const int *__restrict__ any_code,
                                                const int *__restrict__ n_binned)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N)
        return;
    if (n_binned && p >= *n_binned) // a windowed build that dropped atoms outside its window: order[] ends at the atoms binned
        return;
    const int i = order[p];

    const double a = x[i], b = y[i], c = z[i];
    // the image codes are a fourth scattered read per atom (one byte each); k_assign says whether any of them is not neutral
    const unsigned short m = *any_code ? mv[i] : (unsigned short)img::ATOM_NEUTRAL;
    if (pk) {
        pk[p] = CellGrid::Packed{a, b, c, i, (int)m};
        return;
    }
    xs[p] = a;
    ys[p] = b;
    zs[p] = c;
    mvs[p] = m;
}

// the gather of scattered input: whole records (written in input order by k_assign), one random 32-byte read per atom
__global__ __launch_bounds__(256) void k_gather_records(const CellGrid::Packed *__restrict__ rec, const int *__restrict__ order,
                                                        CellGrid::Packed *__restrict__ pk, int64_t N, const int *__restrict__ n_binned)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N || p >= *n_binned)
        return;
    pk[p] = rec[order[p]];
}

__global__ __launch_bounds__(256) void k_unpack(const CellGrid::Packed *__restrict__ pk, int64_t N, double *__restrict__ xs,
                                                double *__restrict__ ys, double *__restrict__ zs, unsigned short *__restrict__ mvs)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N)
        return;
    const CellGrid::Packed r = pk[p];
    xs[p] = r.x; ys[p] = r.y; zs[p] = r.z; mvs[p] = (unsigned short)r.code;
}

int ensure_unpacked(Scope &sc, CellGrid &cg, int64_t N)
{
    if ((!cg.pk && !cg.ix) || cg.xs)
        return MDH_OK;
    cg.xs = sc.alloc_n<double>((size_t)N);
    cg.ys = sc.alloc_n<double>((size_t)N);
    cg.zs = sc.alloc_n<double>((size_t)N);
    cg.mvs = sc.alloc_n<unsigned short>((size_t)N);
    if (sc.failed())
        return sc.error();
    if (cg.pk) // from the records
        hipLaunchKernelGGL(k_unpack, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), cg.pk, N, cg.xs, cg.ys, cg.zs, cg.mvs);
    else // an indirect grid: the gather its build left out
        hipLaunchKernelGGL(k_gather, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), cg.ix, cg.iy, cg.iz, cg.order, cg.xs, cg.ys, cg.zs, N, cg.imv,
                           cg.mvs, (CellGrid::Packed *)nullptr, cg.flags + 4, cg.cell_start + cg.g.ncell);
    MDH_HIP(hipGetLastError());
    return MDH_OK;
}

// ----------------------------------------------------------------------------
// Cell window (decomposed systems).  A rank's atoms — its slab and the halo — occupy a few planes of the GLOBAL cell grid the
// neighbor build works on (global box, global cells: the rows equal the undivided system's), and the passes over ALL cells
// (bin counters zeroed, three scan kernels, the in-cell sort) then cost more than the passes over the atoms.  The caller, who
// knows where its atoms are, promises a window of fractional coordinates along one axis (mdh_hint_cell_window, consumed by
// the next build on this thread); those passes run over the window's planes (one more on each side) only, and the prefix
// array outside them is filled with the constants a full scan would have left there, so that every reader of cell_start is
// served as before.  An atom binned outside the promised window breaks the promise: counted on the device, reported by
// the next call of this thread that builds a grid.
// ----------------------------------------------------------------------------
struct CellWindow { int axis; double lo, hi; bool set; };
static thread_local CellWindow g_window{0, 0.0, 0.0, false};
// ... and inside the window the stretch that holds the atoms whose rows are wanted (mdh_hint_centre_window: a rank's OWN slab; what
// lies between it and the window's ends are ghosts — candidates of the tile kernel, never its centres: their rows are not made)
static thread_local CellWindow g_centre{0, 0.0, 0.0, false};
static thread_local int *g_window_violations = nullptr; // pinned host word of the previous windowed build

// out[a..b) = *v (a value that is on the device only)
// (16-byte stores over the aligned middle — the region behind a slab's window is most of the global grid, 100 MB on eight ranks —
// launched by fill_from() below)
__global__ __launch_bounds__(256) void k_fill_from(int *__restrict__ out, int64_t a, int64_t b, const int *__restrict__ v)
{
    const int val = *v;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b - a < 8) {
        if (a + i < b) out[a + i] = val;
        return;
    }
    const int64_t a4 = (a + 3) & ~(int64_t)3, b4 = b & ~(int64_t)3; // a4 <= b4: out is 16-byte aligned at multiples of four
    const int64_t q = a4 + 4 * i;
    if (q + 4 <= b4) *reinterpret_cast<int4 *>(out + q) = make_int4(val, val, val, val);
    if (i < 4) {
        if (a + i < a4) out[a + i] = val;
        if (b4 + i < b) out[b4 + i] = val;
    }
}
// a one-piece window [a0, a1] of the grid: zeros in front of it, *v (the atoms binned, cell_start[a1]) behind it — ONE launch behind
// the window's scan instead of a memset in front of it and a fill behind (a memset is two 5 us nodes on the stream)
__global__ __launch_bounds__(256) void k_fill_outside(int *__restrict__ out, int64_t a0, int64_t a1, int64_t n1, const int *__restrict__ v)
{
    const int val = *v;
    const int64_t q = 4 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    const int64_t head4 = (a0 + 3) >> 2 << 2;               // the head [0, a0) rounded up to whole quads (the overshoot is fixed below)
    const int64_t t0 = (a1 + 1 + 3) & ~(int64_t)3;          // first aligned index behind the window
    if (q < head4) {
        if (q + 4 <= a0) *reinterpret_cast<int4 *>(out + q) = make_int4(0, 0, 0, 0);
        else for (int64_t i = q; i < a0; ++i) out[i] = 0;
        return;
    }
    const int64_t r = q - head4 + t0;                       // quads behind the window
    if (r == t0) for (int64_t i = a1 + 1; i < t0 && i < n1; ++i) out[i] = val;
    if (r + 4 <= n1) *reinterpret_cast<int4 *>(out + r) = make_int4(val, val, val, val);
    else for (int64_t i = r; i < n1; ++i) out[i] = val;
}
static void fill_from(hipStream_t st, int *out, int64_t a, int64_t b, const int *v)
{
    if (b > a)
        hipLaunchKernelGGL(k_fill_from, dim3(grid_for((b - a) / 4 + 8, 256)), dim3(256), 0, st, out, a, b, v);
}

// out[a..b) += *v  (v outside [a, b)); the entry `skip`, if in range, is left alone
__global__ __launch_bounds__(256) void k_add_from(int *__restrict__ out, int64_t a, int64_t b, const int *__restrict__ v, int64_t skip)
{
    const int64_t i = a + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b && i != skip) out[i] += *v;
}

// atoms binned outside planes [p0, p1) u [p2, p3) of axis 0 (cells are a0-major)
int neighbor_grid_dims(const DBox &b, double rc, Grid &g)
{
    double nc_total = 1.0;
    for (int d = 0; d < 3; ++d) { // neighbor.cpp:203-206
        double f = std::floor(b.thick[d] / rc);
        if (!(f < 2147483647.0)) { set_error("cell grid too large (box thickness / rc overflows int)"); return MDH_ERR_ARG; }
        int n = (int)f;
        g.nc[d] = n > 3 ? n : 3;
        nc_total *= (double)g.nc[d];
    }
    if (nc_total > 2147483000.0) {
        set_error("cell grid too large: " + std::to_string(nc_total) + " cells (the reference indexes cells with int32)");
        return MDH_ERR_ARG;
    }
    g.ncell = (int64_t)g.nc[0] * g.nc[1] * g.nc[2];
    g.rc_inv = 1.0 / rc; // neighbor.cpp:78
    g.mode = 0;
    return MDH_OK;
}

// Do the atoms come in a spatial order?  One workgroup samples 1 024 pairs of consecutive atoms (i, i+1): far = more than two
// bins of ~64 atoms apart along some axis (fractional coordinates, periodic axes wrapped); more than a quarter far -> *flag = 1 (a word
// of pinned host memory, order_hint()): the NEXT builds of this (N, grid) move whole 32-byte records (k_assign writes them in input
// order, the gather reads one random sector per atom instead of three or four: 598 -> ~250 us at 10 M shuffled atoms).  A lattice
// builder's order, a file written cell by cell, a sorted copy: 0.  Launched on the first and every eighth build of a signature.
__global__ __launch_bounds__(1024) void k_order_far_flag(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                         int64_t N, DBox b, double nb0, double nb1, double nb2, int *__restrict__ flag)
{
    __shared__ int s_far;
    if (threadIdx.x == 0) s_far = 0;
    __syncthreads();
    const int64_t step = (N - 1) / 1024 > 0 ? (N - 1) / 1024 : 1;
    const int64_t i = (int64_t)threadIdx.x * step;
    bool far = false;
    if (i + 1 < N) {
        const double dx = x[i + 1] - x[i], dy = y[i + 1] - y[i], dz = z[i + 1] - z[i];
        double f[3] = {dx * b.hi[0] + dy * b.hi[3] + dz * b.hi[6], dx * b.hi[1] + dy * b.hi[4] + dz * b.hi[7], dx * b.hi[2] + dy * b.hi[5] + dz * b.hi[8]};
        const double nb[3] = {nb0, nb1, nb2};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (b.pbc[d]) f[d] -= rint(f[d]);
            far = far || !(fabs(f[d]) * nb[d] <= 2.0); // (NaN counts as far)
        }
    }
    const unsigned long long m = __ballot(far);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_far, __popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) *flag = 4 * s_far > 1024 ? 1 : 0;
}

int build_cell_grid(Scope &sc, const double *x, const double *y, const double *z, int64_t N, const DBox &b,
                    bool wrap_first, bool sort_desc, CellGrid &cg, const int64_t *sort_key, bool packed, bool scattered)
{
    const Grid &g = cg.g;
    hipStream_t st = sc.stream();

    // bin counters in a kept block (all zero whenever idle: the scan clears what it reads), the scan's control words in
    // another; the eight device flags are plain scratch, written by the scan — a build enqueues no hipMemsetAsync
    unsigned *cell_count = static_cast<unsigned *>(sc.alloc_kept(sizeof(unsigned) * (size_t)g.ncell, Scope::KEEP_ZERO));
    unsigned *ctl = static_cast<unsigned *>(sc.alloc_kept(scan_ctl_bytes(g.ncell), Scope::KEEP_SCAN));
    cg.flags = sc.alloc_n<int>(8);
    cg.cell_start = sc.alloc_n<int>((size_t)g.ncell + 1);
    int *cell_id = sc.alloc_n<int>((size_t)N);
    int *rank = sc.alloc_n<int>((size_t)N);
    cg.order = sc.alloc_n<int>((size_t)N + 4); // (four spare entries: the tile kernel reads a cell's first four ids as one 16-byte request)
    unsigned short *mv = sc.alloc_n<unsigned short>((size_t)N);
    cg.xs = cg.ys = cg.zs = nullptr;
    cg.mvs = nullptr;
    cg.pk = nullptr;
    // scattered (with packed): the caller knows that the atoms come in no spatial order
    CellGrid::Packed *rec = nullptr;
    int *rec_flag = nullptr;
    cg.ix = cg.iy = cg.iz = nullptr;
    cg.imv = nullptr;
    if (packed) {
        // records: always for a caller that knows (scattered); for a large system otherwise when the last sample of this (N, grid) said so
        if (!scattered && N >= (int64_t(1) << 18)) {
            const OrderHint h = order_hint(1, N, g.ncell, x);
            scattered = h.word && *(volatile int *)h.word != 0;
            if (h.word && h.sample) rec_flag = h.word;
        }
        // input in some spatial order: no sorted copy, the kernels read through `order` (CellGrid::ix); MDH_INDIRECT=0 /
        // mdh_debug_set_indirect(0): the records always — an A/B switch, and how the tests reach both paths on one input
        // (not for dense cells — six atoms and more, the wide instance's ground: two workgroups per CU hide the staging's
        // dependent gathers badly, build_neighbor(5.0, 50) at 10 M atoms 4.48 -> 4.60 ms; profiles/r06_cell_grid_ab.txt)
        const bool indirect = g_indirect.load(std::memory_order_relaxed) != 0 && !scattered && sort_desc && (double)N <= 6.0 * (double)g.ncell && g_next_rows <= 16;
        g_next_rows = 0;
        if (!indirect) cg.pk = sc.alloc_n<CellGrid::Packed>((size_t)N);
        if (scattered) rec = sc.alloc_n<CellGrid::Packed>((size_t)N);
    } else {
        cg.xs = sc.alloc_n<double>((size_t)N);
        cg.ys = sc.alloc_n<double>((size_t)N);
        cg.zs = sc.alloc_n<double>((size_t)N);
        cg.mvs = sc.alloc_n<unsigned short>((size_t)N);
    }
    if (sc.failed())
        return sc.error();

    // window of planes along axis 0 (orthogonal boxes, rc-wide cells): [p0, p1) and, when it wraps around the ring, [p2, p3)
    int p0 = 0, p1 = g.nc[0], p2 = 0, p3 = 0;
    bool windowed = false;
    if (g_window_violations && *(volatile int *)g_window_violations != 0) {
        *g_window_violations = 0;
        g_window.set = false;
        set_error("an earlier neighbor build on this thread found atoms outside the cell window it had been promised (mdh_hint_cell_window)");
        return MDH_ERR_ARG;
    }
    if (g_window.set) {
        const CellWindow w = g_window;
        g_window.set = false; // one build
        // (only the neighbor builds — the callers of the packed record — know what a window leaves undone; a hint that meets
        // any other grid build is dropped)
        if (packed && w.axis == 0 && !b.tri && g.mode == 0 && g.nc[0] >= 16 && w.hi > w.lo && w.hi - w.lo < 0.75) {
            const double L = b.h[0];
            int lo = (int)std::floor(w.lo * L * g.rc_inv) - 1, hi = (int)std::ceil(w.hi * L * g.rc_inv) + 1; // one plane of margin
            if (hi - lo < g.nc[0] - 2) {
                windowed = true;
                if (lo < 0) { p0 = 0; p1 = std::min(hi, g.nc[0]); p2 = g.nc[0] + lo; p3 = g.nc[0]; }         // wraps below
                else if (hi > g.nc[0]) { p0 = 0; p1 = hi - g.nc[0]; p2 = lo; p3 = g.nc[0]; }                   // wraps above
                else { p0 = lo; p1 = hi; p2 = p3 = 0; }
                if (p2 < p1 && p3 > p2) { windowed = false; p0 = 0; p1 = g.nc[0]; p2 = p3 = 0; }               // (the pieces meet: everything)
            }
        }
    }
    const int64_t plane = (int64_t)g.nc[1] * g.nc[2];
    cg.win_lo = cg.win_hi = 0;
    if (windowed && p3 <= p2) { cg.win_lo = p0; cg.win_hi = p1; } // one piece: the tile kernel runs over its range of tiles
    cg.cen_lo = cg.cen_hi = 0;
    if (g_centre.set) {
        const CellWindow c = g_centre;
        g_centre.set = false; // one build
        if (packed && c.axis == 0 && !b.tri && g.mode == 0 && c.hi > c.lo && c.lo >= 0.0 && c.hi <= 1.0) {
            // the planes an atom of fraction [lo, hi) can be binned into (cell_coords: floor((x - o) rc_inv), clamped); the ends moved
            // out by 1e-9 of their value — far more than the roundings that separate the caller's fraction of an atom from the grid's
            // (x - o) rc_inv, far less than a plane: an atom ON the slab's face is inside whichever way it was rounded
            const double L = b.h[0];
            cg.cen_lo = std::max(0, std::min(g.nc[0] - 1, (int)std::floor(c.lo * L * g.rc_inv * (1.0 - 1e-9) - 1e-9)));
            cg.cen_hi = std::min(g.nc[0], (int)std::floor(c.hi * L * g.rc_inv * (1.0 + 1e-9) + 1e-9) + 1);
        }
    }
    cg.flags_fresh = true;
    // slack for the raw-vs-wrapped consistency flag: far above rounding, far below a cell width
    const double slack = 0.01 / (g.rc_inv > 0 ? g.rc_inv : 1.0);
    // the promise is checked where the atoms are binned (a word of pinned host memory the kernel writes) and read by the next
    // build of this thread or by mdh_cell_window_check
    CellPlanes win{p0, p1, p2, p3, nullptr};
    if (windowed) {
        if (!g_window_violations) MDH_HIP(hipHostMalloc(reinterpret_cast<void **>(&g_window_violations), sizeof(int), hipHostMallocDefault));
        *g_window_violations = 0;
        win.bad = g_window_violations;
    }
    if (rec_flag) {
        const double *h = b.h;
        const double vol = std::fabs(h[0] * (h[4] * h[8] - h[5] * h[7]) - h[1] * (h[3] * h[8] - h[5] * h[6]) + h[2] * (h[3] * h[7] - h[4] * h[6]));
        const double edge = std::cbrt(64.0 * vol / (double)N);
        double nb[3];
        for (int d = 0; d < 3; ++d) nb[d] = std::max(1.0, std::floor(b.thick[d] / edge));
        hipLaunchKernelGGL(k_order_far_flag, dim3(1), dim3(1024), 0, st, x, y, z, N, b, nb[0], nb[1], nb[2], rec_flag);
    }
    const unsigned gen = next_scan_gen(); // stamps of this build's k_assign; the (first) scan below is launched with the same value
    // atoms per lane (A/B: MDH_ASSIGN_K = 1, 2, 4; small systems keep one atom per lane: they need the workgroups to fill the chip)
    static const int assign_k_env = [] { const char *e = std::getenv("MDH_ASSIGN_K"); return e ? std::atoi(e) : 0; }();
    // (measured at 10 M atoms: 145 -> 120 us on a lattice, 162 -> 162 on a polycrystal — 10 M runs of one atom, the atomics' own
    // throughput — 413 -> 440 on a shuffled frame, which therefore keeps one: profiles/r05_assign_k.txt)
    const int assign_k = assign_k_env > 0 ? assign_k_env : ((N >= (int64_t)1 << 20 && !scattered) ? 4 : 1);
#define MDH_ASSIGN(TRI, K) hipLaunchKernelGGL((k_assign<TRI, K>), dim3(grid_for(N, 256 * K)), dim3(256), 0, st, x, y, z, N, b, g, (int)wrap_first, cell_id, rank, cell_count, ctl, gen, slack, mv, win, rec, packed ? 1 : 0)
    if (b.tri) {
        if (assign_k >= 4) MDH_ASSIGN(true, 4); else if (assign_k >= 2) MDH_ASSIGN(true, 2); else MDH_ASSIGN(true, 1);
    } else {
        if (assign_k >= 4) MDH_ASSIGN(false, 4); else if (assign_k >= 2) MDH_ASSIGN(false, 2); else MDH_ASSIGN(false, 1);
    }
#undef MDH_ASSIGN
    auto scan_piece = [&](int64_t from, int64_t to, unsigned use_gen, int *flags) {
        launch_scan_gen(st, cell_count + from, cg.cell_start + from, to - from, ctl, use_gen, true, flags); // [to] = the piece's total
    };
    if (!windowed) {
        scan_piece(0, g.ncell, gen, cg.flags);
    } else {
        // the pieces in index order: [p0, p1) then [p2, p3); a piece is scanned on its own, the atoms before it added by the
        // fill / by a second add pass; cell_start elsewhere = what a full scan leaves: the atoms binned so far.  The counters
        // outside the window are zero (kept block; atoms out there take no slot) and stay untouched.
        const int64_t a0 = p0 * plane, a1 = p1 * plane, b0 = p2 * plane, b1 = p3 * plane;
        // (constant fills through the runtime's fill kernel: 16-byte stores, 5 us per 10 MB against 15 of a store per thread)
        hipError_t fill_err = hipSuccess;
        auto fill_const = [&](int64_t from, int64_t to, int v) {
            if (to > from && fill_err == hipSuccess)
                fill_err = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cg.cell_start + from), v, (size_t)(to - from), st);
        };
        if (b1 > b0) fill_const(0, a0, 0);
        // (from a multiple of four cells on: the scan moves 16 bytes per access only from an aligned start — 66 against 26 us for the
        // 3.6 M cells of a 10 M-atom slab; the up to three counters in front of the window are zero, and zero is what the cells in
        // front of the window hold)
        scan_piece(a0 & ~(int64_t)3, a1, gen, cg.flags);
        if (b1 <= b0) { // one piece: everything outside it in one launch
            const int64_t n1 = g.ncell + 1, quads = ((a0 + 3) >> 2) + ((n1 - std::min(n1, (a1 + 1 + 3) & ~(int64_t)3) + 3) >> 2) + 1;
            hipLaunchKernelGGL(k_fill_outside, dim3(grid_for(quads, 256)), dim3(256), 0, st, cg.cell_start, a0, a1, n1, cg.cell_start + a1);
        } else if (b1 > b0) {
            // second piece: offsets start at the first piece's total, which sits on the device in cell_start[a1]
            fill_from(st, cg.cell_start, a1 + 1, b0, cg.cell_start + a1);
            scan_piece(b0, b1, next_scan_gen(), nullptr);
            hipLaunchKernelGGL(k_add_from, dim3(grid_for(b1 - b0 + 1, 256)), dim3(256), 0, st, cg.cell_start, b0, b1 + 1, cg.cell_start + a1, (int64_t)-1);
            // behind the window: the number of atoms BINNED (cell_start[b1], on the device) — N unless the promise was broken; with
            // the constant N a cell behind a broken window spanned the records [n_binned, N), which k_scatter / k_gather never wrote
            if (b1 < g.ncell)
                fill_from(st, cg.cell_start, b1 + 1, g.ncell + 1, cg.cell_start + b1);
        }
        MDH_HIP(fill_err);
    }
    sc.keep_confirm(cell_count); // every counter a binned atom touched has been read and cleared by a scan enqueued above
    hipLaunchKernelGGL(k_scatter, dim3(grid_for(N, 256)), dim3(256), 0, st, cell_id, rank, cg.cell_start, cg.order, N);
    if (sort_desc) {
        if (!windowed && !sort_key && (double)N > 6.0 * (double)g.ncell) {
            hipLaunchKernelGGL(k_sort_cells_dense, dim3(grid_for(g.ncell, 32)), dim3(256), 0, st, cg.cell_start, cg.order, g.ncell, rank);
        } else if (!windowed) {
            hipLaunchKernelGGL(k_sort_cells, dim3(grid_for(g.ncell, 256)), dim3(256), 0, st, cg.cell_start, cg.order, g.ncell, sort_key, rank);
        } else {
            hipLaunchKernelGGL(k_sort_cells, dim3(grid_for((p1 - p0) * plane, 256)), dim3(256), 0, st, cg.cell_start + p0 * plane, cg.order, (p1 - p0) * plane, sort_key, rank);
            if (p3 > p2)
                hipLaunchKernelGGL(k_sort_cells, dim3(grid_for((p3 - p2) * plane, 256)), dim3(256), 0, st, cg.cell_start + p2 * plane, cg.order, (p3 - p2) * plane, sort_key, rank);
        }
    }
    // (the atoms binned = the grid's total, on the device: all N unless absent atoms were handed in or a window's promise was broken)
    const int *n_binned = cg.cell_start + g.ncell;
    if (packed && !cg.pk) { cg.ix = x; cg.iy = y; cg.iz = z; cg.imv = mv; } // indirect: nothing is gathered
    else if (rec) hipLaunchKernelGGL(k_gather_records, dim3(grid_for(N, 256)), dim3(256), 0, st, rec, cg.order, cg.pk, N, n_binned);
    else hipLaunchKernelGGL(k_gather, dim3(grid_for(N, 256)), dim3(256), 0, st, x, y, z, cg.order, cg.xs, cg.ys, cg.zs, N, mv, cg.mvs, cg.pk, cg.flags + 4, n_binned);
    MDH_HIP(hipGetLastError());
    return MDH_OK;
}

// ----------------------------------------------------------------------------
// 27-cell scan, one thread per centre atom (centres taken in cell order so the
// lanes of a wave share their candidate cells through L1/L2).
//   MODE 0: count only            (first pass of the exact-width variant)
//   MODE 1: reference semantics   (write valid slots only, caller pre-filled pads)
//   MODE 2: also write the pads   (-1, rc+1)
// ----------------------------------------------------------------------------
// one centre atom (position p of the cell-sorted arrays): the reference's 27-cell walk, neighbor.cpp:139-177
template <bool TRI, int MODE>
__device__ __forceinline__ int neighbor_one(const SortedView &sv, const int *__restrict__ cell_start, const DBox &b,
                                            const Grid &g, double rc, int *__restrict__ verlet, double *__restrict__ dist,
                                            int *__restrict__ nn, int64_t M, int64_t p, double xi, double yi, double zi, int c0, int c1,
                                            int c2)
{
    int cnt = 0;
    const int i = sv.id_of(p);
    const double rcsq = rc * rc; // neighbor.cpp:127
    const int64_t row = (int64_t)i * M;
    const bool zrun = (c2 >= 1) && (c2 + 1 < g.nc[2]); // the three z-cells are one contiguous run
    for (int a = c0 - 1; a <= c0 + 1; ++a) {            // neighbor.cpp:147-151
        const int ca = pmod(a, g.nc[0]);
        for (int bb = c1 - 1; bb <= c1 + 1; ++bb) {
            const int64_t base = ((int64_t)ca * g.nc[1] + pmod(bb, g.nc[1])) * g.nc[2];
            for (int seg = 0; seg < (zrun ? 1 : 3); ++seg) {
                int s, e;
                if (zrun) {
                    s = cell_start[base + c2 - 1];
                    e = cell_start[base + c2 + 2];
                } else {
                    const int cc = pmod(c2 - 1 + seg, g.nc[2]);
                    s = cell_start[base + cc];
                    e = cell_start[base + cc + 1];
                }
                // four candidates per trip, their loads issued together: with one candidate per trip every one of them is a
                // dependent L2 round trip (the loop carries the row count through a store), and a thread of the mop-up kernel in
                // a fat cell at the box's far faces walks a thousand of them — 0.54 ms for the 3 % of a 3.4 M-atom box at rc = 5 A
                for (int q0 = s; q0 < e; q0 += 4) {
                    double xq[4], yq[4], zq[4];
                    int jq[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) // (past the end of the piece: its last candidate again, not looked at)
                        sv.get(min(q0 + u, e - 1), xq[u], yq[u], zq[u], jq[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (q0 + u >= e || jq[u] == i)
                            continue;
                        double dx = xq[u] - xi, dy = yq[u] - yi, dz = zq[u] - zi; // raw x[j] - wrapped centre, :164-166
                        pbc<TRI>(b, dx, dy, dz);
                        const double d2 = dx * dx + dy * dy + dz * dz;
                        if (d2 <= rcsq) {
                            if (MODE != 0 && cnt < M) {
                                verlet[row + cnt] = jq[u];
                                dist[row + cnt] = sqrt(d2);
                            }
                            ++cnt;
                        }
                    }
                }
            }
        }
    }
    nn[i] = cnt;
    if (MODE == 2) {
        const double pad = rc + 1.0;
        for (int64_t n = cnt; n < M; ++n) {
            verlet[row + n] = -1;
            dist[row + n] = pad;
        }
    }
    return cnt;
}

// The same walk by a whole wavefront for ONE atom: 64 candidates per trip, the hits' slots from a ballot (candidate order = row
// order, as above).  For the listed tiles of the mop-up kernel: their atoms sit in the fat last cells of the box, a thread walks
// 300 ... 1000 candidates there four at a time (80 ... 380 dependent trips), a wave 9 runs of one to three trips.
template <bool TRI, int MODE>
__device__ __forceinline__ int neighbor_one_wave(const SortedView &sv, const int *__restrict__ cell_start, const DBox &b,
                                                 const Grid &g, double rc, int *__restrict__ verlet, double *__restrict__ dist,
                                                 int *__restrict__ nn, int64_t M, int64_t p, double xi, double yi, double zi, int c0,
                                                 int c1, int c2)
{
    const int lane = (int)(threadIdx.x & 63);
    int cnt = 0;
    const int i = sv.id_of(p);
    const double rcsq = rc * rc; // neighbor.cpp:127
    const int64_t row = (int64_t)i * M;
    const bool zrun = (c2 >= 1) && (c2 + 1 < g.nc[2]);
    for (int a = c0 - 1; a <= c0 + 1; ++a) { // neighbor.cpp:147-151
        const int ca = pmod(a, g.nc[0]);
        for (int bb = c1 - 1; bb <= c1 + 1; ++bb) {
            const int64_t base = ((int64_t)ca * g.nc[1] + pmod(bb, g.nc[1])) * g.nc[2];
            for (int seg = 0; seg < (zrun ? 1 : 3); ++seg) {
                int s, e;
                if (zrun) {
                    s = cell_start[base + c2 - 1];
                    e = cell_start[base + c2 + 2];
                } else {
                    const int cc = pmod(c2 - 1 + seg, g.nc[2]);
                    s = cell_start[base + cc];
                    e = cell_start[base + cc + 1];
                }
                for (int q0 = s; q0 < e; q0 += 64) {
                    const int q = q0 + lane;
                    bool hit = false;
                    int j = -1;
                    double d2 = 0.0;
                    if (q < e) {
                        double xq, yq, zq;
                        sv.get(q, xq, yq, zq, j);
                        double dx = xq - xi, dy = yq - yi, dz = zq - zi; // raw x[j] - wrapped centre, :164-166
                        pbc<TRI>(b, dx, dy, dz);
                        d2 = dx * dx + dy * dy + dz * dz;
                        hit = j != i && d2 <= rcsq;
                    }
                    const unsigned long long m = __ballot(hit);
                    if (MODE != 0 && hit) {
                        const int slot = cnt + __popcll(m & ((1ull << lane) - 1ull));
                        if (slot < M) {
                            verlet[row + slot] = j;
                            dist[row + slot] = sqrt(d2);
                        }
                    }
                    cnt += __popcll(m);
                }
            }
        }
    }
    if (lane == 0) nn[i] = cnt;
    if (MODE == 2) {
        const double pad = rc + 1.0;
        for (int64_t n = cnt + lane; n < M; n += 64) {
            verlet[row + n] = -1;
            dist[row + n] = pad;
        }
    }
    return cnt;
}

template <bool TRI, int MODE>
__device__ __forceinline__ void neighbor_atoms_body(const SortedView &sv,
                                                  const int *__restrict__ cell_start, int64_t N, const DBox &b, const Grid &g,
                                                  double rc, int *__restrict__ verlet, double *__restrict__ dist,
                                                  int *__restrict__ nn, int64_t M, int *__restrict__ max_count,
                                                  const TileFilter &tf)
{
    const bool take_all = tf.moved && *tf.moved != 0; // the tiled kernel stood down: this kernel does the whole call
    if (tf.flag && !take_all && (tf.list || *tf.any == 0)) // nothing to mop up here (flagged tiles go to k_neighbor_tiles when listed)
        return;
    int cnt = 0;
    // the grid is capped (a stand-by launch then costs a few thousand workgroups that leave at once, not N / 256 of them):
    // a workgroup strides over the atoms
    // (cell_start[ncell] = the atoms the grid holds: N, or fewer after a windowed build that dropped atoms outside its window —
    // the records behind them were never written)
    N = min(N, (int64_t)cell_start[g.ncell]);
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < N; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = base + threadIdx.x;
        bool mine = p < N;
        int c0 = 0, c1 = 0, c2 = 0;
        double xi = 0, yi = 0, zi = 0;
        if (mine) {
            { int idp; sv.get(p, xi, yi, zi, idp); }
            if (b.anypbc) // neighbor.cpp:139-142
                wrap<TRI>(b, xi, yi, zi);
            cell_coords<TRI>(b, g, xi, yi, zi, c0, c1, c2);
            if (tf.flag) { // fallback pass: only atoms of tiles the LDS-tiled kernel could not hold (or all of them when it stood down)
                const int t = ((c0 / tf.tile) * tf.nt[1] + (c1 / tf.tile)) * tf.nt[2] + (c2 / tf.tile_z);
                mine = take_all || tf.flag[t] != 0;
            }
        }
        if (mine) {
            cnt = max(cnt, neighbor_one<TRI, MODE>(sv, cell_start, b, g, rc, verlet, dist, nn, M, p, xi, yi, zi, c0, c1, c2));
            if (tf.cna_todo) defer(tf.cna_todo, sv.id_of(p));
        }
    }
    if (MODE == 0) {
        int m = cnt;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            int t = __shfl_xor(m, d, 64);
            m = t > m ? t : m;
        }
        if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(max_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(max_count, m); // (read first: a single word serialises ~90 atomics per microsecond)
    }
}

// mop-up of the tiles the wave kernel listed (halo over the LDS budget, atoms far outside the box): a workgroup per listed
// tile, its threads over the tile's centre atoms — the cost follows the number of listed tiles, not N
template <bool TRI, int MODE>
__global__ __launch_bounds__(256) void k_neighbor(SortedView sv, const int *__restrict__ cell_start, int64_t N, DBox b, Grid g, double rc,
                                                  int *__restrict__ verlet, double *__restrict__ dist, int *__restrict__ nn, int64_t M,
                                                  int *__restrict__ max_count, TileFilter tf)
{
    neighbor_atoms_body<TRI, MODE>(sv, cell_start, N, b, g, rc, verlet, dist, nn, M, max_count, tf);
}

template <bool TRI, int MODE>
__device__ __forceinline__ void neighbor_tiles_body(const SortedView &sv,
                                                        const int *__restrict__ cell_start, const DBox &b, const Grid &g, double rc,
                                                        int *__restrict__ verlet, double *__restrict__ dist, int *__restrict__ nn,
                                                        int64_t M, int *__restrict__ max_count, const TileFilter &tf)
{
    if (tf.moved && *tf.moved != 0) // k_neighbor takes the whole call
        return;
    const int nlist = min(*tf.any, tf.list_cap);
    int best = 0;
    // a workgroup per (listed tile, column of the tile), a wavefront per atom of the column's z-run (neighbor_one_wave)
    const int ncol = tf.tile * tf.tile;
    const int wave = (int)(threadIdx.x >> 6), nwave = (int)(blockDim.x >> 6);
    // (... and a column's atoms in MOP_CHUNKS interleaved shares, a workgroup each: the corner column of the box holds 84 atoms in
    // one cell where the mean is 12 — 21 atoms per wave was the whole kernel's critical path, 440 us)
    constexpr int MOP_CHUNKS = 8;
    for (int64_t w8 = blockIdx.x; w8 < (int64_t)nlist * ncol * MOP_CHUNKS; w8 += gridDim.x) {
        const int64_t w = w8 / MOP_CHUNKS;
        const int chunk = (int)(w8 - w * MOP_CHUNKS);
        const int t = tf.list[w / ncol], colq = (int)(w % ncol);
        const int t2 = t % tf.nt[2], t1 = (t / tf.nt[2]) % tf.nt[1], t0 = t / (tf.nt[2] * tf.nt[1]);
        const int z0 = t2 * tf.tile_z, z1 = min(z0 + tf.tile_z, g.nc[2]);
        const int a = t0 * tf.tile + colq / tf.tile, c = t1 * tf.tile + colq % tf.tile;
        if (a < g.nc[0] && c < g.nc[1]) {
            const int64_t col = ((int64_t)a * g.nc[1] + c) * g.nc[2];
            const int s = cell_start[col + z0], e = cell_start[col + z1]; // the z-run of a column is contiguous
            for (int p = s + chunk * nwave + wave; p < e; p += nwave * MOP_CHUNKS) {
                double xi, yi, zi;
                { int idp; sv.get(p, xi, yi, zi, idp); }
                if (b.anypbc)
                    wrap<TRI>(b, xi, yi, zi);
                int c0, c1, c2;
                cell_coords<TRI>(b, g, xi, yi, zi, c0, c1, c2);
                best = max(best, neighbor_one_wave<TRI, MODE>(sv, cell_start, b, g, rc, verlet, dist, nn, M, p, xi, yi, zi, c0, c1, c2));
                if (tf.cna_todo && (threadIdx.x & 63) == 0) defer(tf.cna_todo, sv.id_of(p));
            }
        }
    }
    if (MODE == 0) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d, 64));
        if ((threadIdx.x & 63) == 0 && best > __hip_atomic_load(max_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(max_count, best);
    }
}

template <bool TRI, int MODE>
__global__ __launch_bounds__(256) void k_neighbor_tiles(SortedView sv, const int *__restrict__ cell_start, DBox b, Grid g, double rc,
                                                        int *__restrict__ verlet, double *__restrict__ dist, int *__restrict__ nn, int64_t M,
                                                        int *__restrict__ max_count, TileFilter tf)
{
    neighbor_tiles_body<TRI, MODE>(sv, cell_start, b, g, rc, verlet, dist, nn, M, max_count, tf);
}

// the two stand-bys behind a tile kernel that lists its leftovers, as ONE launch (a launch that finds nothing to do costs
// ~4 us): the whole call atom by atom if the tile kernel stood down (unwrapped input), else the listed tiles
template <bool TRI, int MODE>
__global__ __launch_bounds__(256) void k_neighbor_mop(SortedView sv, const int *__restrict__ cell_start, int64_t N, DBox b, Grid g, double rc,
                                                      int *__restrict__ verlet, double *__restrict__ dist, int *__restrict__ nn, int64_t M,
                                                      int *__restrict__ max_count, TileFilter tf)
{
    if (tf.moved && *tf.moved != 0) neighbor_atoms_body<TRI, MODE>(sv, cell_start, N, b, g, rc, verlet, dist, nn, M, max_count, tf);
    else neighbor_tiles_body<TRI, MODE>(sv, cell_start, b, g, rc, verlet, dist, nn, M, max_count, tf);
}

template <int MODE>
static void launch_neighbor(hipStream_t st, const CellGrid &cg, int64_t N, const DBox &b, double rc, int *verlet,
                            double *dist, int *nn, int64_t M, int *max_count, TileFilter tf = TileFilter{})
{
    // behind a tile kernel that lists its leftovers this launch only stands by for unwrapped input (device flag): a small grid
    // then, whose workgroups stride over the atoms if they do have to take the call (10 -> 3 us when they leave at once)
    dim3 grid(std::min(grid_for(N, 256), tf.list ? 2048 : 8192)), block(256);
    if (tf.list) { // behind a tile kernel with a list (at most list_cap entries): both stand-bys in one launch
        if (b.tri)
            hipLaunchKernelGGL((k_neighbor_mop<true, MODE>), grid, block, 0, st, view_of(cg), cg.cell_start, N, b, cg.g, rc, verlet, dist, nn, M, max_count, tf);
        else
            hipLaunchKernelGGL((k_neighbor_mop<false, MODE>), grid, block, 0, st, view_of(cg), cg.cell_start, N, b, cg.g, rc, verlet, dist, nn, M, max_count, tf);
        return;
    }
    if (b.tri)
        hipLaunchKernelGGL((k_neighbor<true, MODE>), grid, block, 0, st, view_of(cg), cg.cell_start, N, b, cg.g, rc, verlet, dist, nn, M, max_count, tf);
    else
        hipLaunchKernelGGL((k_neighbor<false, MODE>), grid, block, 0, st, view_of(cg), cg.cell_start, N, b, cg.g, rc, verlet, dist, nn, M, max_count, tf);
}

// ----------------------------------------------------------------------------
// small row-wise helpers
// ----------------------------------------------------------------------------
// neighbor.cpp:745-775: selection of the first k entries by strict '<' over all M columns.
// One wave per workgroup, 64/L consecutive rows of it, L lanes to a row (L = 1 ... 16, the smallest that keeps the LDS copy
// of the rows near 10 KB: 12+ waves per CU, where 64 rows of 50 entries — build_neighbor(5.0, 50), the published workflow —
// left one wave per SIMD and the list read at 0.8 TB/s).  The rows are one contiguous piece of memory: read with 16-byte
// loads into LDS (element c of row r at [c * ROWS + r]; lane j * ROWS + r walks columns a+1+j, a+1+j+L, ...: consecutive
// words, conflict-free), selected there — every lane starts from entry a and only takes a strictly smaller one, the L
// partial results meet by (distance, column): the first of the smallest, as the serial loop has it — and only written back
// when something moved: the rows of a k-nearest search arrive sorted, and every analysis that borrows them "sorts" them
// again (the reference does the same); for those the kernel is one read of the list.
template <int L>
__global__ __launch_bounds__(64) void k_sort_rows(int *__restrict__ verlet, double *__restrict__ dist, int64_t N, int M, int k,
                                                  unsigned inv_m)
{
    constexpr int ROWS = 64 / L;
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    double *ld = reinterpret_cast<double *>(sort_lds);        // [M][ROWS]
    int *lv = reinterpret_cast<int *>(ld + (size_t)M * ROWS); // [M][ROWS]
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int rows = (int)((N - row0) < ROWS ? (N - row0) : ROWS);
    const int total = rows * M;
    const int t = threadIdx.x;
    double *__restrict__ gd = dist + row0 * M;
    int *__restrict__ gv = verlet + row0 * M;
    // e -> (row, column): e < 2^16 and M < 2^16, so the high word of e * ceil(2^32 / M) is e / M exactly
    auto slot = [&](int e) {
        const int r = (int)__umulhi((unsigned)e, inv_m);
        return (e - r * M) * ROWS + r;
    };
    const bool vec = rows == ROWS && ((reinterpret_cast<uintptr_t>(dist) | reinterpret_cast<uintptr_t>(verlet)) & 15) == 0; // (ROWS * M is a multiple of 4)
    if (vec) {
        {
            const double2 *gd2 = reinterpret_cast<const double2 *>(gd);
            const int4 *gv4 = reinterpret_cast<const int4 *>(gv);
#pragma unroll 4
            for (int p = t; p < (total >> 1); p += 64) {
                const double2 v = gd2[p];
                ld[slot(2 * p)] = v.x; ld[slot(2 * p + 1)] = v.y;
            }
#pragma unroll 4
            for (int p = t; p < (total >> 2); p += 64) {
                const int4 v = gv4[p];
                lv[slot(4 * p)] = v.x; lv[slot(4 * p + 1)] = v.y; lv[slot(4 * p + 2)] = v.z; lv[slot(4 * p + 3)] = v.w;
            }
        }
    } else {
        for (int e = t; e < total; e += 64) { const int s = slot(e); ld[s] = gd[e]; lv[s] = gv[e]; }
    }
    __syncthreads();
    const int r = t & (ROWS - 1), j = t / ROWS;
    const bool live = r < rows;
    const double *lr = ld + r;
    // One walk first: entries 0 ... p-1 stay where they are if they ascend and nothing behind them is smaller — rows that
    // arrive sorted (a k-nearest list; the same list sorted for the analysis before this one) are done after this walk,
    // rows sorted to 12 and now wanted to 14 start at 12.  (s: the smallest entry behind a; '<' only, as the selection.)
    int first = k;
    if (live) {
        double s = __builtin_inf();
        for (int c = k + j; c < M; c += L) {
            const double v = lr[c * ROWS];
            if (v < s) s = v;
        }
#pragma unroll
        for (int w = ROWS; w < 64; w <<= 1) {
            const double o = __shfl_xor(s, w);
            if (o < s) s = o;
        }
        for (int a = k - 1; a >= 0; --a) {
            const double v = lr[a * ROWS];
            if (s < v) first = a;
            if (v < s) s = v;
        }
    }
#pragma unroll
    for (int w = 1; w < 64; w <<= 1) {
        const int o = __shfl_xor(first, w);
        first = o < first ? o : first;
    }
    bool moved = false;
    for (int a = first; a < k; ++a) {
        int best = a;
        double db = live ? lr[a * ROWS] : 0.0;
        if (live) {
            int c = a + 1 + j;
            for (; c + 3 * L < M; c += 4 * L) {
                const double v0 = lr[c * ROWS], v1 = lr[(c + L) * ROWS], v2 = lr[(c + 2 * L) * ROWS], v3 = lr[(c + 3 * L) * ROWS];
                if (v0 < db) { db = v0; best = c; }
                if (v1 < db) { db = v1; best = c + L; }
                if (v2 < db) { db = v2; best = c + 2 * L; }
                if (v3 < db) { db = v3; best = c + 3 * L; }
            }
            for (; c < M; c += L) {
                const double v = lr[c * ROWS];
                if (v < db) { db = v; best = c; }
            }
        }
#pragma unroll
        for (int w = ROWS; w < 64; w <<= 1) { // the other lanes of this row are w, 2w, ... lanes away
            const double od = __shfl_xor(db, w);
            const int ob = __shfl_xor(best, w);
            if (od < db || (od == db && ob < best)) { db = od; best = ob; }
        }
        if (live && j == 0 && best != a) {
            const double td = ld[a * ROWS + r];
            ld[a * ROWS + r] = db; ld[best * ROWS + r] = td;
            const int tv = lv[a * ROWS + r];
            lv[a * ROWS + r] = lv[best * ROWS + r]; lv[best * ROWS + r] = tv;
            moved = true;
        }
        if (L > 1) __syncthreads(); // (one wave: the writes above are in LDS before the next column walk of the row's other lanes)
    }
    if (!__syncthreads_or(moved ? 1 : 0))
        return;
    if (vec) {
        double2 *gd2 = reinterpret_cast<double2 *>(gd);
        int4 *gv4 = reinterpret_cast<int4 *>(gv);
#pragma unroll 4
        for (int p = t; p < (total >> 1); p += 64)
            gd2[p] = make_double2(ld[slot(2 * p)], ld[slot(2 * p + 1)]);
#pragma unroll 4
        for (int p = t; p < (total >> 2); p += 64)
            gv4[p] = make_int4(lv[slot(4 * p)], lv[slot(4 * p + 1)], lv[slot(4 * p + 2)], lv[slot(4 * p + 3)]);
    } else {
        for (int e = t; e < total; e += 64) { const int s = slot(e); gd[e] = ld[s]; gv[e] = lv[s]; }
    }
}

// the round-2 form of the same (64 rows per workgroup, a lane to a row): kept for A/B measurements (mdh_debug_set_neighbor_variant(3))
constexpr int SORT_ROWS = 64;
__global__ __launch_bounds__(SORT_ROWS) void k_sort_rows_r2(int *__restrict__ verlet, double *__restrict__ dist, int64_t N,
                                                            int64_t M, int k)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    double *ld = reinterpret_cast<double *>(sort_lds);            // [M][64]
    int *lv = reinterpret_cast<int *>(ld + (size_t)M * SORT_ROWS); // [M][64]
    const int64_t row0 = (int64_t)blockIdx.x * SORT_ROWS;
    const int rows = (int)((N - row0) < SORT_ROWS ? (N - row0) : SORT_ROWS);
    const int64_t total = (int64_t)rows * M;
    const int t = threadIdx.x;
    for (int64_t e = t; e < total; e += SORT_ROWS) {
        const int r = (int)(e / M), c = (int)(e - (int64_t)r * M);
        ld[c * SORT_ROWS + r] = dist[row0 * M + e];
        lv[c * SORT_ROWS + r] = verlet[row0 * M + e];
    }
    __syncthreads();
    bool moved = false;
    if (t < rows) {
        for (int a = 0; a < k; ++a) {
            int best = a;
            double db = ld[a * SORT_ROWS + t];
            for (int c = a + 1; c < M; ++c) {
                const double v = ld[c * SORT_ROWS + t];
                if (v < db) { db = v; best = c; }
            }
            if (best != a) {
                const double td = ld[a * SORT_ROWS + t];
                ld[a * SORT_ROWS + t] = db; ld[best * SORT_ROWS + t] = td;
                const int tv = lv[a * SORT_ROWS + t];
                lv[a * SORT_ROWS + t] = lv[best * SORT_ROWS + t]; lv[best * SORT_ROWS + t] = tv;
                moved = true;
            }
        }
    }
    if (!__syncthreads_or(moved ? 1 : 0))
        return;
    for (int64_t e = t; e < total; e += SORT_ROWS) {
        const int r = (int)(e / M), c = (int)(e - (int64_t)r * M);
        dist[row0 * M + e] = ld[c * SORT_ROWS + r];
        verlet[row0 * M + e] = lv[c * SORT_ROWS + r];
    }
}

// the same in place in HBM, for rows too wide for the LDS copy
__global__ __launch_bounds__(256) void k_sort_rows_wide(int *__restrict__ verlet, double *__restrict__ dist, int64_t N,
                                                        int64_t M, int k)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    int *v = verlet + i * M;
    double *d = dist + i * M;
    for (int a = 0; a < k; ++a) {
        int best = a;
        double db = d[a];
        for (int c = a + 1; c < M; ++c) {
            double t = d[c];
            if (t < db) { db = t; best = c; }
        }
        if (best != a) {
            double td = d[a]; d[a] = db; d[best] = td;
            int tv = v[a]; v[a] = v[best]; v[best] = tv;
        }
    }
}

// whole rows of up to SORT_BLOCK_MAX entries, one workgroup per row: a bitonic network over (distance, index) keys in LDS
// (the selection sort above is quadratic in the row length: a 36 000-wide row — surface atoms of a slab looking across
// its vacuum — took minutes)
constexpr int SORT_BLOCK_MAX = 8192;
constexpr int SORT_LDS_WIDEST = 1024; // the selection kernel above with 16 lanes to a row: 4 rows of 1024 entries in 48 KB
__global__ __launch_bounds__(256) void k_sort_rows_block(int *__restrict__ verlet, double *__restrict__ dist, int64_t M, int P)
{
    extern __shared__ double sort_block_lds[];
    double *ld = sort_block_lds;
    int *lv = (int *)(sort_block_lds + P);
    const int64_t row = blockIdx.x;
    const int t = threadIdx.x;
    for (int c = t; c < P; c += 256) {
        ld[c] = c < M ? dist[row * M + c] : 1.0e300;
        lv[c] = c < M ? verlet[row * M + c] : 0x7fffffff;
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = t; e < (P >> 1); e += 256) {
                const int lo = ((e & ~(stride - 1)) << 1) | (e & (stride - 1)), hi = lo | stride;
                const bool up = (lo & size) == 0;
                const double a = ld[lo], c = ld[hi];
                const int va = lv[lo], vc = lv[hi];
                const bool gt = a > c || (a == c && va > vc);
                if (gt == up) { ld[lo] = c; ld[hi] = a; lv[lo] = vc; lv[hi] = va; }
            }
            __syncthreads();
        }
    for (int c = t; c < M; c += 256) {
        dist[row * M + c] = ld[c];
        verlet[row * M + c] = lv[c];
    }
}

template <bool TRI>
__global__ __launch_bounds__(256) void k_wrap(double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
                                              int64_t N, DBox b)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    double xi = x[i], yi = y[i], zi = z[i];
    wrap<TRI>(b, xi, yi, zi); // neighbor.cpp:695 (unconditional)
    x[i] = xi; y[i] = yi; z[i] = zi;
}

// (the rows of the workgroup a chunk at a time through LDS: common.hpp stage_row_chunk)
__global__ __launch_bounds__(64) void k_average(double rc, const int *__restrict__ verlet,
                                                const double *__restrict__ dist, const int *__restrict__ nn,
                                                int64_t N, int64_t M, const double *__restrict__ value,
                                                double *__restrict__ out, int include_self)
{
    __shared__ int ids[ROW_CHUNK * 64];
    __shared__ double dst[ROW_CHUNK * 64];
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + threadIdx.x;
    const bool on = i < N;
    double s = 0.0;
    int cnt = 0;
    if (on && include_self) { s += value[i]; ++cnt; }
    const int n = on ? min(nn[i], (int)M) : 0;
    const int most = wave_max(n);
    for (int c0 = 0; c0 < most; c0 += ROW_CHUNK) {
        __syncthreads();
        stage_row_chunk<true>(verlet, dist, N, M, row0, c0, ids, dst);
        __syncthreads();
        // neighbor.cpp:729-736 (sequential sum in list order); the values of a chunk's entries requested together
        double val[ROW_CHUNK];
#pragma unroll
        for (int q = 0; q < ROW_CHUNK; ++q)
            val[q] = (c0 + q < n && dst[q * 64 + threadIdx.x] <= rc) ? value[safe_id(ids[q * 64 + threadIdx.x], i, N)] : 0.0;
#pragma unroll
        for (int q = 0; q < ROW_CHUNK; ++q)
            if (c0 + q < n && dst[q * 64 + threadIdx.x] <= rc) { s += val[q]; ++cnt; }
    }
    if (on) out[i] = cnt > 0 ? s / cnt : 0.0;
}

// filter_overlap_atom (neighbor.cpp:390-486): keep[j] = 0 iff some atom i < j lies within rc of j.  The reference lets
// every centre i mark its higher-numbered neighbours; here every atom j looks for a lower-numbered i and evaluates the
// very expression centre i would: raw x[j] - wrapped x[i], folded, squared, compared with rc^2.  The 27-cell
// neighbourhood is symmetric, so the same pairs are examined.
template <bool TRI>
__global__ __launch_bounds__(256) void k_filter_overlap(const double *__restrict__ xs, const double *__restrict__ ys,
                                                        const double *__restrict__ zs, const int *__restrict__ order,
                                                        const int *__restrict__ cell_start, int64_t N, DBox b, Grid g,
                                                        double rc, unsigned char *__restrict__ keep)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N)
        return;
    const double xr = xs[p], yr = ys[p], zr = zs[p]; // raw position of j
    double xw = xr, yw = yr, zw = zr;
    if (b.anypbc)
        wrap<TRI>(b, xw, yw, zw);
    int c0, c1, c2;
    cell_coords<TRI>(b, g, xw, yw, zw, c0, c1, c2);
    const int j = order[p];
    const double rcsq = rc * rc;
    bool hit = false;
    for (int a = c0 - 1; a <= c0 + 1 && !hit; ++a) {
        const int ca = pmod(a, g.nc[0]);
        for (int bb = c1 - 1; bb <= c1 + 1 && !hit; ++bb) {
            const int64_t base = ((int64_t)ca * g.nc[1] + pmod(bb, g.nc[1])) * g.nc[2];
            for (int cc = c2 - 1; cc <= c2 + 1 && !hit; ++cc) {
                const int64_t cell = base + pmod(cc, g.nc[2]);
                for (int q = cell_start[cell]; q < cell_start[cell + 1]; ++q) {
                    if (order[q] >= j)
                        continue;
                    double xi = xs[q], yi = ys[q], zi = zs[q]; // the lower-numbered atom is the centre: wrapped (:430-436)
                    if (b.anypbc)
                        wrap<TRI>(b, xi, yi, zi);
                    double dx = xr - xi, dy = yr - yi, dz = zr - zi;
                    pbc<TRI>(b, dx, dy, dz);
                    if (dx * dx + dy * dy + dz * dz <= rcsq) { hit = true; break; }
                }
            }
        }
    }
    keep[j] = hit ? 0 : 1;
}

__global__ __launch_bounds__(256) void k_max_i32(const int *__restrict__ v, int64_t n, int *__restrict__ out)
{
    int m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = max(m, v[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
}

// last exact row width seen for a (N, grid) signature; set < 0: query only
static int width_hint(int64_t N, int64_t ncell, int set)
{
    struct Entry { int64_t N, ncell; int width; };
    static std::mutex mu;
    static std::vector<Entry> table;
    std::lock_guard<std::mutex> lk(mu);
    for (auto &e : table)
        if (e.N == N && e.ncell == ncell) {
            if (set >= 0) e.width = set;
            return e.width;
        }
    if (set > 0) {
        if (table.size() >= 64) table.erase(table.begin());
        table.push_back(Entry{N, ncell, set});
    }
    return 0;
}

// One pass over a built cell grid: mode 0 = counts only (nn, *dmax), 1 = reference semantics (caller's pads), 2 = pads written.
// The tile kernel where it applies, the round-1 tiled kernel for cells too full for it, the thread-per-atom code for the rest.
// pattern != nullptr (mode 1 or 2): fixed-cutoff CNA labels as well.  Fused into the tile kernel where that runs (the
// leftovers of its mop-up kernels are listed in todo and labelled from the finished rows); *fused = false: nothing was
// labelled, the caller runs the analysis on the lists
static int neighbor_pass(Scope &sc, const CellGrid &cg, const DBox &b, int64_t N, double rc, int *dv, double *dd, int *dn,
                         int64_t M, int mode, int *dmax, int *pattern = nullptr, int *todo = nullptr, bool *fused = nullptr)
{
    hipStream_t st = sc.stream();
    if (!cg.flags_fresh) MDH_HIP(hipMemsetAsync(cg.flags + 2, 0, sizeof(int) * 2, st)); // the tile lists of this pass (flags[0], unwrapped input, stays)
    cg.flags_fresh = false;
    TileFilter tf{};
    bool done = false;
    if (g_neighbor_variant == 0) { // tile kernel (orthogonal and triclinic boxes); the thread-per-atom code below then only mops up what it listed
        GridStats gs;
        MDH_TRY(grid_stats_hint(sc, cg, N, &gs));
        const bool cna = pattern && mode != 0;
        const LanePlan lp = plan_lane(b, cg.g, N, mode == 0 ? 1 : M, gs, rc, cna, mode == 0);
        if (lp.txy) {
            if (cna) tf.cna_todo = todo;
            MDH_TRY(launch_neighbor_lane(sc, cg, lp, N, b, rc, dv, dd, dn, mode == 0 ? 1 : M, mode == 2, mode == 0, dmax, tf, cna ? pattern : nullptr));
            done = true;
            if (cna && fused) *fused = true;
        }
    }
    if (!done && mode != 0 && g_neighbor_variant != 1 && !b.tri) { // cells too full for the kernel above (or forced): the round-1 tiled kernel
        int64_t occ = 0;
        MDH_TRY(occupied_cells_hint(sc, cg, N, &occ));
        const TiledPlan plan = plan_tiled(b, cg.g, N, M, occ);
        if (plan.tile) {
            MDH_TRY(ensure_unpacked(sc, const_cast<CellGrid &>(cg), N));
            MDH_TRY(launch_neighbor_tiled(sc, cg, plan, N, b, rc, dv, dd, dn, M, mode == 2, tf));
        }
    }
    if (mode == 0) launch_neighbor<0>(st, cg, N, b, rc, nullptr, nullptr, dn, 1, dmax, tf);
    else if (mode == 2) launch_neighbor<2>(st, cg, N, b, rc, dv, dd, dn, M, nullptr, tf);
    else launch_neighbor<1>(st, cg, N, b, rc, dv, dd, dn, M, nullptr, tf);
    if (g_moved_probe) // mdh_debug_track_counters(1): the build's "image codes not valid" flag, for mdh_debug_counters
        MDH_HIP(hipMemcpyAsync(g_moved_probe, cg.flags, sizeof(int), hipMemcpyDeviceToHost, st));
    MDH_HIP(hipGetLastError());
    return MDH_OK;
}

// For other units of the library (knn.hip): the rows of a cutoff build on device arrays — its own cell grid, pads written (-1 / rc + 1),
// counts that keep running past M — enqueued on the Scope's stream.  ids_only: the caller does not read the distances (dd must still
// be a buffer of N x M: the kernels for rows of <= 16 slots and the mop-up code write them regardless).
void lane_ids_only(bool on); // neighbor_lane.hip
int neighbor_rows_device(Scope &sc, const double *dx, const double *dy, const double *dz, int64_t N, const DBox &b, double rc, int *dv,
                         double *dd, int *dn, int64_t M, const int64_t *dkey, bool ids_only)
{
    CellGrid cg;
    MDH_TRY(neighbor_grid_dims(b, rc, cg.g));
    g_next_rows = (int)std::min<int64_t>(M, 1 << 20);
    MDH_TRY(build_cell_grid(sc, dx, dy, dz, N, b, true, true, cg, dkey, true));
    // pads written (the tile kernel then stores whole 16-byte groups; leaving the pads out measured SLOWER: 2.96 against 2.54 ms
    // at 10 M atoms, rc 3.8, 24 slots); distances wanted or not (rows of more than 16 slots: the wide instance skips them)
    static const bool want_dist = [] { const char *e = std::getenv("MDH_KNN_ROWS_DIST"); return e && std::atoi(e) != 0; }(); // A/B
    lane_ids_only(ids_only && !want_dist);
    const int rcode = neighbor_pass(sc, cg, b, N, rc, dv, dd, dn, M, 2, nullptr);
    lane_ids_only(false);
    return rcode;
}

int moved_probe(int enable) // enable > 0: start tracking; 0: stop; < 0: the last value (-1: none)
{
    if (enable > 0 && !g_moved_probe) {
        if (hipHostMalloc(reinterpret_cast<void **>(&g_moved_probe), sizeof(int), hipHostMallocDefault) != hipSuccess) { g_moved_probe = nullptr; return -1; }
        *g_moved_probe = -1;
    } else if (enable == 0 && g_moved_probe) {
        int *p = g_moved_probe;
        g_moved_probe = nullptr;
        (void)hipDeviceSynchronize();
        (void)hipHostFree(p);
    }
    return g_moved_probe ? *(volatile int *)g_moved_probe : -1;
}

} // namespace mdh

using namespace mdh;

extern "C" {

int mdh_build_neighbor(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                       const double *origin3, const int *boundary3, double rc, int *verlet, double *dist, int *nn,
                       int64_t max_neigh, int fill_pads, int space, void *stream)
{
    return mdh_build_neighbor_keyed(x, y, z, N, box9, origin3, boundary3, rc, verlet, dist, nn, max_neigh, fill_pads, nullptr,
                                    space, stream);
}

// key (N) i64, or NULL: the atoms of a cell are listed by descending key instead of descending index
int mdh_build_neighbor_keyed(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                             const double *origin3, const int *boundary3, double rc, int *verlet, double *dist, int *nn,
                             int64_t max_neigh, int fill_pads, const int64_t *key, int space, void *stream)
{
    if (N < 0 || N >= 2147483647LL || !(rc > 0) || max_neigh <= 0) { set_error("mdh_build_neighbor: invalid N, rc or max_neigh"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    // host space + reference semantics: pads come from the caller's buffers, so they are uploaded too
    int *dv = sc.stage(verlet, (size_t)(N * max_neigh), space, !fill_pads, true);
    double *dd = sc.stage(dist, (size_t)(N * max_neigh), space, !fill_pads, true);
    int *dn = sc.stage(nn, (size_t)N, space, false, true);
    const int64_t *dkey = key ? sc.stage_in(key, (size_t)N, space) : nullptr;
    if (sc.failed())
        return sc.error();
    CellGrid cg;
    MDH_TRY(neighbor_grid_dims(b, rc, cg.g));
    {
        ProfRange pr("cell_grid", sc.stream());
        g_next_rows = (int)std::min<int64_t>(max_neigh, 1 << 20);
        MDH_TRY(build_cell_grid(sc, dx, dy, dz, N, b, true, true, cg, dkey, true));
    }
    {
        ProfRange pr("k_neighbor", sc.stream());
        MDH_TRY(neighbor_pass(sc, cg, b, N, rc, dv, dd, dn, max_neigh, fill_pads ? 2 : 1, nullptr));
    }
    return sc.finish(space);
}

// mdh_build_neighbor followed by mdh_fcna with the same rc, in one pass over the tiles where the tile kernel applies
int mdh_build_neighbor_fcna(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                            const double *origin3, const int *boundary3, double rc, int *verlet, double *dist, int *nn,
                            int64_t max_neigh, int fill_pads, int *pattern, const int64_t *key, int space, void *stream)
{
    if (N < 0 || N >= 2147483647LL || !(rc > 0) || max_neigh <= 0) { set_error("mdh_build_neighbor_fcna: invalid N, rc or max_neigh"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    int *dv = sc.stage(verlet, (size_t)(N * max_neigh), space, !fill_pads, true);
    double *dd = sc.stage(dist, (size_t)(N * max_neigh), space, !fill_pads, true);
    int *dn = sc.stage(nn, (size_t)N, space, false, true);
    int *dp = sc.stage(pattern, (size_t)N, space, true, true); // atoms without 12 or 14 neighbours keep the caller's value (cna.cpp:456)
    // the to-do list of the labels (count first) in a kept block whose two counters — workgroups done (word 0), length (word 64) — are
    // zero whenever it is idle (the kernel that walks the list clears them when it leaves, cna.hip k_fcna): no memset per call
    int *done = static_cast<int *>(sc.alloc_kept(sizeof(int) * ((size_t)N + 1 + 64), Scope::KEEP_TODO));
    int *todo = done ? done + 64 : nullptr;
    const int64_t *dkey = key ? sc.stage_in(key, (size_t)N, space) : nullptr;
    if (sc.failed())
        return sc.error();
    CellGrid cg;
    MDH_TRY(neighbor_grid_dims(b, rc, cg.g));
    {
        ProfRange pr("cell_grid", sc.stream());
        g_next_rows = (int)std::min<int64_t>(max_neigh, 1 << 20);
        MDH_TRY(build_cell_grid(sc, dx, dy, dz, N, b, true, true, cg, dkey, true));
    }
    bool fused = false;
    {
        ProfRange pr("k_neighbor", sc.stream());
        MDH_TRY(neighbor_pass(sc, cg, b, N, rc, dv, dd, dn, max_neigh, fill_pads ? 2 : 1, nullptr, dp, todo, &fused));
    }
    {
        ProfRange pr("k_fcna", sc.stream());
        if (fused) launch_fcna_listed(sc.stream(), b, dx, dy, dz, N, dv, max_neigh, dn, dp, rc, todo, done);
        else launch_fcna_all(sc.stream(), b, dx, dy, dz, N, dv, max_neigh, dn, dp, rc, todo, done);
        MDH_HIP(hipGetLastError());
    }
    sc.keep_confirm(done);
    return sc.finish(space);
}

int mdh_hint_centre_window(int axis, double frac_lo, double frac_hi)
{
    g_centre = CellWindow{axis, frac_lo, frac_hi, true};
    return MDH_OK;
}

int mdh_hint_cell_window(int axis, double frac_lo, double frac_hi)
{
    g_window = CellWindow{axis, frac_lo, frac_hi, true};
    return MDH_OK;
}

int mdh_cell_window_check(void *stream)
{
    if (!g_window_violations)
        return MDH_OK;
    MDH_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    if (*(volatile int *)g_window_violations != 0) {
        *g_window_violations = 0;
        set_error("the last neighbor build on this thread found atoms outside the cell window it had been promised (mdh_hint_cell_window): its rows are incomplete");
        return MDH_ERR_ARG;
    }
    return MDH_OK;
}

int mdh_debug_set_neighbor_variant(int v)
{
    g_neighbor_variant = v;
    return MDH_OK;
}

int mdh_debug_set_indirect(int on) { return g_indirect.exchange(on ? 1 : 0); }

int mdh_neighbor_count(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                       const double *origin3, const int *boundary3, double rc, int *nn, int *max_count, int space,
                       void *stream)
{
    if (N < 0 || N >= 2147483647LL || !(rc > 0) || !max_count) { set_error("mdh_neighbor_count: invalid N or rc"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    *max_count = 0;
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    int *dn = sc.stage(nn, (size_t)N, space, false, true);
    int *dmax = sc.alloc_n<int>(1);
    if (sc.failed())
        return sc.error();
    MDH_HIP(hipMemsetAsync(dmax, 0, sizeof(int), sc.stream()));
    CellGrid cg;
    MDH_TRY(neighbor_grid_dims(b, rc, cg.g));
    MDH_TRY(build_cell_grid(sc, dx, dy, dz, N, b, true, true, cg, nullptr, true));
    MDH_TRY(neighbor_pass(sc, cg, b, N, rc, nullptr, nullptr, dn, 1, 0, dmax));
    MDH_HIP(hipMemcpyAsync(max_count, dmax, sizeof(int), hipMemcpyDeviceToHost, sc.stream()));
    MDH_TRY(sc.finish(space));
    MDH_HIP(hipStreamSynchronize(sc.stream()));
    return MDH_OK;
}

// _neighbor.build_neighbor_without_max_neigh in one call: ONE cell grid serves the counting pass and the build
int mdh_build_neighbor_exact(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                             const double *origin3, const int *boundary3, double rc, int *nn, int64_t *width,
                             mdh_alloc_rows_fn alloc, void *user, int space, void *stream)
{
    return mdh_build_neighbor_exact_keyed(x, y, z, N, box9, origin3, boundary3, rc, nn, width, alloc, user, nullptr, space, stream);
}

// key (N) i64, or NULL: as in mdh_build_neighbor_keyed
int mdh_build_neighbor_exact_keyed(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                   const double *origin3, const int *boundary3, double rc, int *nn, int64_t *width,
                                   mdh_alloc_rows_fn alloc, void *user, const int64_t *key, int space, void *stream)
{
    return mdh_build_neighbor_exact_fcna(x, y, z, N, box9, origin3, boundary3, rc, nn, width, alloc, user, nullptr, key, space, stream);
}

// pattern (N) i32 or NULL: the fixed-cutoff CNA labels of the same cutoff as well (mdh_build_neighbor_fcna at the exact width)
int mdh_build_neighbor_exact_fcna(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                  const double *origin3, const int *boundary3, double rc, int *nn, int64_t *width,
                                  mdh_alloc_rows_fn alloc, void *user, int *pattern, const int64_t *key, int space, void *stream)
{
    if (N < 0 || N >= 2147483647LL || !(rc > 0) || !width || !alloc) { set_error("mdh_build_neighbor_exact: invalid N, rc, width or allocator"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    int *verlet = nullptr;
    double *dist = nullptr;
    *width = 1; // neighbor.cpp:301-304: at least one column
    if (N == 0) {
        if (alloc(user, 0, 1, &verlet, &dist) != 0) { set_error("mdh_build_neighbor_exact: the row allocator failed"); return MDH_ERR_NOMEM; }
        return MDH_OK;
    }
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    int *dn = sc.stage(nn, (size_t)N, space, false, true);
    int *dmax = sc.alloc_n<int>(1);
    const int64_t *dkey = key ? sc.stage_in(key, (size_t)N, space) : nullptr;
    int *dp = pattern ? sc.stage(pattern, (size_t)N, space, true, true) : nullptr; // atoms without 12 or 14 neighbours keep the caller's value (cna.cpp:456)
    int *done = pattern ? static_cast<int *>(sc.alloc_kept(sizeof(int) * ((size_t)N + 1 + 64), Scope::KEEP_TODO)) : nullptr; // (as in mdh_build_neighbor_fcna)
    int *todo = done ? done + 64 : nullptr;
    if (sc.failed())
        return sc.error();
    hipStream_t st = sc.stream();
    MDH_HIP(hipMemsetAsync(dmax, 0, sizeof(int), st));
    CellGrid cg;
    MDH_TRY(neighbor_grid_dims(b, rc, cg.g));
    {
        ProfRange pr("cell_grid", st);
        g_next_rows = width_hint(N, cg.g.ncell, -1); // the width the last build of this (N, grid) found (0: none yet)
        MDH_TRY(build_cell_grid(sc, dx, dy, dz, N, b, true, true, cg, dkey, true));
    }
    // the labels of a build: inside the tile kernel where that ran (its leftovers listed in todo), from the finished rows otherwise
    bool fused = false;
    auto labels = [&](const int *dv, int64_t M) {
        if (!dp)
            return;
        ProfRange pr("k_fcna", st);
        if (fused) launch_fcna_listed(st, b, dx, dy, dz, N, dv, M, dn, dp, rc, todo, done);
        else launch_fcna_all(st, b, dx, dy, dz, N, dv, M, dn, dp, rc, todo, done);
        sc.keep_confirm(done); // (the list's walker leaves the counters zero)
    };
    // Width hint: the largest count the previous call with the same (N, grid) found.  A sequence of calls on one system (a
    // trajectory, the same analysis repeated) almost always finds the same maximum again, so the rows are built at that
    // width at once and the counts written by the build confirm it — the counting pass is skipped.  A wrong hint costs one
    // wasted build (the counts are then known) and is replaced; results never depend on it.
    int hmax = 0;
    const int hint = space == MDH_DEVICE ? width_hint(N, cg.g.ncell, -1) : 0; // (host buffers: a discarded first allocation would still be a copy-back target)
    bool built = false;
    if (hint > 0) {
        if (alloc(user, N, hint, &verlet, &dist) != 0 || !verlet || !dist) { set_error("mdh_build_neighbor_exact: the row allocator failed"); return MDH_ERR_NOMEM; }
        int *dv = sc.stage(verlet, (size_t)(N * hint), space, false, true);
        double *dd = sc.stage(dist, (size_t)(N * hint), space, false, true);
        if (sc.failed())
            return sc.error();
        {
            ProfRange pr("k_neighbor", st);
            MDH_TRY(neighbor_pass(sc, cg, b, N, rc, dv, dd, dn, hint, 2, nullptr, dp, todo, &fused));
            hipLaunchKernelGGL(k_max_i32, dim3(1024), dim3(256), 0, st, dn, N, dmax);
            MDH_HIP(hipMemcpyAsync(&hmax, dmax, sizeof(int), hipMemcpyDeviceToHost, st));
        }
        // (enqueued before the width is known: with a confirmed hint — the usual case — the call has no idle gap; after a wrong one the
        // second build labels again, and a label depends on the atom's neighbours only, not on the width of the rows: an atom the
        // wasted pass labelled had its 12 or 14 neighbours listed in full, anything else it left to the list)
        labels(dv, hint);
        MDH_HIP(hipStreamSynchronize(st));
        built = (hmax > 1 ? hmax : 1) == hint;
    } else {
        ProfRange pr("k_neighbor_count", st);
        MDH_TRY(neighbor_pass(sc, cg, b, N, rc, nullptr, nullptr, dn, 1, 0, dmax));
        MDH_HIP(hipMemcpyAsync(&hmax, dmax, sizeof(int), hipMemcpyDeviceToHost, st));
        MDH_HIP(hipStreamSynchronize(st));
    }
    const int64_t M = hmax > 1 ? hmax : 1;
    *width = M;
    width_hint(N, cg.g.ncell, (int)(M <= 4096 ? M : 0));
    if (!built) {
        if (alloc(user, N, M, &verlet, &dist) != 0 || !verlet || !dist) { set_error("mdh_build_neighbor_exact: the row allocator failed"); return MDH_ERR_NOMEM; }
        int *dv = sc.stage(verlet, (size_t)(N * M), space, false, true);
        double *dd = sc.stage(dist, (size_t)(N * M), space, false, true);
        if (sc.failed())
            return sc.error();
        {
            ProfRange pr("k_neighbor", st);
            fused = false;
            MDH_TRY(neighbor_pass(sc, cg, b, N, rc, dv, dd, dn, M, 2, nullptr, dp, todo, &fused));
        }
        labels(dv, M);
    }
    MDH_HIP(hipGetLastError());
    return sc.finish(space);
}

int mdh_filter_overlap_atom(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                            const double *origin3, const int *boundary3, double rc, unsigned char *keep, int space,
                            void *stream)
{
    if (N < 0 || N >= 2147483647LL || !(rc > 0)) { set_error("mdh_filter_overlap_atom: invalid N or rc"); return MDH_ERR_ARG; }
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N == 0)
        return MDH_OK;
    Scope sc(stream);
    const double *dx = sc.stage_in(x, (size_t)N, space), *dy = sc.stage_in(y, (size_t)N, space), *dz = sc.stage_in(z, (size_t)N, space);
    unsigned char *dk = sc.stage(keep, (size_t)N, space, false, true);
    if (sc.failed())
        return sc.error();
    CellGrid cg;
    MDH_TRY(neighbor_grid_dims(b, rc, cg.g));
    MDH_TRY(build_cell_grid(sc, dx, dy, dz, N, b, true, false, cg));
    if (b.tri)
        hipLaunchKernelGGL(k_filter_overlap<true>, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, N, b, cg.g, rc, dk);
    else
        hipLaunchKernelGGL(k_filter_overlap<false>, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), cg.xs, cg.ys, cg.zs, cg.order, cg.cell_start, N, b, cg.g, rc, dk);
    return sc.finish(space);
}

int mdh_sort_verlet_by_distance(int *verlet, double *dist, int64_t N, int64_t M, int sort_num, int space, void *stream)
{
    if (N < 0 || M <= 0) { set_error("mdh_sort_verlet_by_distance: invalid shape"); return MDH_ERR_ARG; }
    if (N == 0 || sort_num <= 0)
        return MDH_OK;
    Scope sc(stream);
    int *dv = sc.stage(verlet, (size_t)(N * M), space, true, true);
    double *dd = sc.stage(dist, (size_t)(N * M), space, true, true);
    if (sc.failed())
        return sc.error();
    const int k = (int)(sort_num < M ? sort_num : M);
    if (M == 1)
        return sc.finish(space);
    if (M > SORT_LDS_WIDEST) { // (the reference's selection sort, neighbor.cpp: its order among EQUAL distances — a perfect lattice — is part of the result)
        hipLaunchKernelGGL(k_sort_rows_wide, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), dv, dd, N, M, k);
        return sc.finish(space);
    }
    if (g_neighbor_variant == 3 && (size_t)M * SORT_ROWS * 12 <= 60 * 1024) {
        hipLaunchKernelGGL(k_sort_rows_r2, dim3(grid_for(N, SORT_ROWS)), dim3(SORT_ROWS), (size_t)M * SORT_ROWS * 12, sc.stream(), dv, dd, N, M, k);
        return sc.finish(space);
    }
    int L = 1; // lanes to a row: the fewest that keep the rows of a wave within ~10 KB of LDS
    while (L < 16 && (size_t)(64 / L) * M * 12 > 10 * 1024) L <<= 1;
    const size_t lds = (size_t)(64 / L) * M * 12;
    const unsigned inv_m = (unsigned)((0x100000000ull + (uint64_t)M - 1) / (uint64_t)M);
    const dim3 grid(grid_for(N, 64 / L)), block(64);
    switch (L) {
    case 1: hipLaunchKernelGGL(k_sort_rows<1>, grid, block, lds, sc.stream(), dv, dd, N, (int)M, k, inv_m); break;
    case 2: hipLaunchKernelGGL(k_sort_rows<2>, grid, block, lds, sc.stream(), dv, dd, N, (int)M, k, inv_m); break;
    case 4: hipLaunchKernelGGL(k_sort_rows<4>, grid, block, lds, sc.stream(), dv, dd, N, (int)M, k, inv_m); break;
    case 8: hipLaunchKernelGGL(k_sort_rows<8>, grid, block, lds, sc.stream(), dv, dd, N, (int)M, k, inv_m); break;
    default: hipLaunchKernelGGL(k_sort_rows<16>, grid, block, lds, sc.stream(), dv, dd, N, (int)M, k, inv_m); break;
    }
    return sc.finish(space);
}

extern "C++" {
namespace mdh {
// Whole rows in HBM by ascending distance, equal distances by id — NOT the reference's order among equal distances (that is the
// selection sort above, quadratic in the row length): for the Voronoi search lists, whose cells do not depend on the order of
// equidistant planes.  Rows of 161 ... 8192 entries: one workgroup per row, a bitonic network in LDS.
int sort_rows_any_tie_order(int *dv, double *dd, int64_t N, int64_t M, void *stream)
{
    if (N <= 0 || M <= 0)
        return MDH_OK;
    if (M <= 80 || M > SORT_BLOCK_MAX) // (whole rows: the selection is quadratic in the row length, the network is not)
        return mdh_sort_verlet_by_distance(dv, dd, N, M, (int)M, MDH_DEVICE, stream);
    int P = 256;
    while (P < M) P <<= 1;
    const size_t bytes = (size_t)P * 12;
    if (bytes > 48 * 1024)
        MDH_HIP(hipFuncSetAttribute((const void *)k_sort_rows_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL(k_sort_rows_block, dim3((unsigned)N), dim3(256), bytes, static_cast<hipStream_t>(stream), dv, dd, M, P);
    MDH_HIP(hipGetLastError());
    return MDH_OK;
}
} // namespace mdh
} // extern "C++"

int mdh_wrap_positions(double *x, double *y, double *z, int64_t N, const double *box9, const double *origin3,
                       const int *boundary3, int space, void *stream)
{
    DBox b;
    MDH_TRY(make_box(b, box9, origin3, boundary3));
    if (N <= 0)
        return MDH_OK;
    Scope sc(stream);
    double *dx = sc.stage(x, (size_t)N, space, true, true), *dy = sc.stage(y, (size_t)N, space, true, true), *dz = sc.stage(z, (size_t)N, space, true, true);
    if (sc.failed())
        return sc.error();
    if (b.tri)
        hipLaunchKernelGGL(k_wrap<true>, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), dx, dy, dz, N, b);
    else
        hipLaunchKernelGGL(k_wrap<false>, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), dx, dy, dz, N, b);
    return sc.finish(space);
}

int mdh_average_by_neighbor(double rc, const int *verlet, const double *dist, const int *nn, int64_t N, int64_t M,
                            const double *value, double *value_ave, int include_self, int space, void *stream)
{
    if (N <= 0)
        return MDH_OK;
    Scope sc(stream);
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const double *dd = sc.stage_in(dist, (size_t)(N * M), space);
    const int *dn = sc.stage_in(nn, (size_t)N, space);
    const double *dval = sc.stage_in(value, (size_t)N, space);
    double *dout = sc.stage(value_ave, (size_t)N, space, false, true);
    if (sc.failed())
        return sc.error();
    hipLaunchKernelGGL(k_average, dim3(grid_for(N, 64)), dim3(64), 0, sc.stream(), rc, dv, dd, dn, N, M, dval, dout, include_self);
    return sc.finish(space);
}
}

MDH_WARM_UNIT(neighbor)
