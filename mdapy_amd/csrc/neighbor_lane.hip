// neighbor_lane.hip — cutoff neighbor search on LDS tiles, one thread per centre atom, single-precision pruning with
// hit masks in registers (gfx950).
//
// Same result, bit for bit, as the thread-per-atom kernel in neighbor.hip and therefore as src/neighbor.cpp:102-187 of
// the reference (ids, order inside a row, counts, distances).  This is the fast path of mdh_build_neighbor /
// mdh_neighbor_count / mdh_build_neighbor_exact for orthogonal and triclinic boxes, periodic or open along any of their vectors.
//
// A workgroup owns a tile of TXY x TXY x TZ cells and stages the atoms of the halo ((TXY+2)^2 x (TZ+2) cells, ONE CELL
// PER THREAD, two 16-byte loads per atom from the cell-sorted 32-byte records) into LDS once: raw doubles for the values that are written,
// and single-precision coordinates relative to the tile corner with the periodic image shift folded in
// (u = x - (X0 + L n), n = image number of the candidate's cell as seen from the tile + the atom's own image code) for the
// values that only decide.  The staging pass also lists the tile's centre atoms and tabulates, per halo cell, the LDS range
// of the 3-cell z-run around it (cells (i+da, j+db, k-1..k+1) are contiguous), so a centre finds each of its 9 runs with
// one 4-byte LDS read.
//
// Scan (hand-written, scan_run_asm).  A thread walks its 9 runs in the reference's order (neighbor.cpp:147-151), four
// candidates per trip: four 16-byte LDS reads in flight, then per candidate 3 subtractions, an FMA chain ending in
// e = d2 - c (c a little below rc^2), v_alignbit shifting e's SIGN into the hit mask, and — for two candidates at a time —
// an unsigned v_min3 of e's bits that tracks the smallest non-negative e seen.  No per-candidate branch, ticket store,
// address arithmetic or VCC traffic: 7.5 VALU instructions per candidate, all register-only single-precision / integer
// operations (2.7 cycles each on this chip against 4.4 for anything that touches SGPRs / VCC or is double precision;
// tools/ubench).  |e_f32 - e_exact| <= tol (host, from the tile extent): e < 0 is a hit, e > W a miss; a thread that saw
// 0 <= e <= W redoes ITS masks with the reference's own double-precision expression.  Single precision only prunes: every
// distance that is WRITTEN is recomputed in double precision from the raw coordinates exactly as the reference does (raw
// x[j] - wrapped x[i], minimum image, (dx*dx + dy*dy) + dz*dz, sqrt), so rows are bit-identical.
//
// Centres and waves.  The tile's centres go out in chunks of 64 (one per lane); chunk c is taken by wave (c + tile) mod 4,
// which scans it, expands its masks into tickets in ITS rows of LDS and writes its rows — no workgroup barrier behind the
// staging.
//
// Output, two instances (template TK8).  TK8 (rows of <= 16 slots, runs of <= 32 candidates; <= 128 VGPRs and ~38 KB of LDS:
// FOUR workgroups per CU): one-byte tickets (run << 5 | position in the run; run 8 is the last of the walk, so its tickets
// are the last of the row and a count says which they are); the centre's own lane decodes them, recomputes the distances
// in double precision and holds its whole row in registers; the four lanes of a quad exchange 16-byte pieces (quad_transpose)
// and store 16 bytes each: a quad writes 64 contiguous bytes of every row array.  Wide (!TK8: dense cells, rows of up to 128
// slots, runs of up to 96 candidates as three masks): two-byte tickets, rows streamed four slots at a time, fewer rows per
// wave (rw) where LDS is short.
//
// One tile per workgroup, straight-line code (template LOOP = false); the form that walks a list of tiles (LOOP = true)
// only for the second pass and for what a tile list longer than the host expected leaves over: its loop-carried uniform
// state costs 140 scalar-register spills.  Tiles whose halo does not fit the LDS budget are listed and taken by a SECOND
// launch of the same kernel on one-cell slices of those tiles; what is left after that (a dense blob, atoms far outside the
// box on an open axis, a run longer than the instance's masks) is listed again for the thread-per-atom code
// (k_neighbor_tiles).  Not taken at all (thread-per-atom kernel / round-1 tiled kernel, same results): fewer than 7 cells on
// a periodic axis or 4 on an open one, atoms more than 14 box lengths outside an orthogonal periodic box (device flag; nearer
// ones carry their image number in their record, grid.hpp img::), max_neigh > 128 (> 64 where a cell holds more than 19.5 atoms),
// grids where more than 5 % of the runs hold 89 atoms or more.
//
// Measured (10 061 824-atom FCC Cu, rc = 0.854 a, M = 16; DESIGN.md 3a has the counters, the per-phase time stamps of the
// MDH_STAMPS build and the tables of what moved the kernel and what was built and not kept): round-1 tiled kernel 1.78 ms,
// round 2 of this kernel 1.14 ms, now 0.92 ms.
#include "common.hpp"
#include "grid.hpp"
#include "cna_core.hpp"
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace mdh {
namespace lane {

// A workgroup is NW waves (template parameter of the kernel): 4 — 256 threads, four workgroups per CU — or 8 for the big-tile
// instance — 512 threads, two per CU.  A tile has at most one halo cell per thread.
constexpr int cen_cap(int nw) { return 80 * nw; } // centre atoms a tile may hold
// tickets of a row + the spare slot, rounded up (rows are read back four tickets at a time).  (cna_rows: 28 bytes, room for a centre's
// bond rows — built, and dropped: the 2 KB cost the headline tile 49 atoms of room, 169 of its 51 200 tiles went to the slice pass
// (33 us), and since cna_counts_words fetches no row by a computed index nothing needs them)
__host__ __device__ constexpr int ticket_row(int M, bool cna_rows) { return (cna_rows && ((M + 4) & ~3) < 28) ? 28 : ((M + 4) & ~3); }
static constexpr int NEUTRAL = img::CELL_NEUTRAL;  // a halo cell's image code "no shift" (grid.hpp img::)
static constexpr int NEUTRAL3 = img::NEUTRAL;      // combined code "no shift"

// exact n / d for 32-bit n by multiply and shift (Granlund & Montgomery, "Division by invariant integers using
// multiplication", fig. 4.1): the tile and halo-cell coordinates of a workgroup come from divisions by launch constants, and a
// division is ~30 instructions in front of a tile's first load
struct FastDiv {
    unsigned mul, sh1, sh2, d;
    __device__ __forceinline__ unsigned div(unsigned n) const
    {
        const unsigned t = __umulhi(mul, n);
        return (t + ((n - t) >> sh1)) >> sh2;
    }
};
static FastDiv make_fastdiv(unsigned d)
{
    FastDiv f{0, 0, 0, d};
    unsigned l = 0;
    while ((1ull << l) < d) ++l; // ceil(log2 d)
    f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 1 ? l - 1 : 0;
    return f;
}
// txy, tz: tile shape in cells; by_nt2, by_nt1: divisions of a tile id by the tile counts along z and y of this launch;
// mhz, mhxy: ceil(2^16 / (tz + 2)), ceil(2^16 / (txy + 2)) — thread id / HZ as (tid * mhz) >> 16, exact for tid < 512
struct Shape { int txy, tz; FastDiv by_nt2, by_nt1; unsigned mhz, mhxy; };
static Shape make_shape(int txy, int tz, int nt1, int nt2)
{
    return Shape{txy, tz, make_fastdiv((unsigned)nt2), make_fastdiv((unsigned)nt1), (65536u + (unsigned)tz + 1u) / ((unsigned)tz + 2u),
                 (65536u + (unsigned)txy + 1u) / ((unsigned)txy + 2u)};
}
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef int Int4 __attribute__((ext_vector_type(4))); // 16 bytes of a row (nontemporal vector stores take clang vector types)

static int g_last_plan[8]; // test hook (mdh_debug_neighbor_plan)
static int g_last_listed = -1; // tiles the previous build with the last plan's (N, grid) listed for the slice pass (mdh_debug_counters)
#ifdef MDH_STAMPS
__device__ unsigned long long g_stamps[65536 * 8];
#define STAMP(k) do { if (tid == 0 && !parent && tile.id < 65536) g_stamps[tile.id * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

// workgroup barrier that orders LDS accesses only: global stores issued before it stay in flight (__syncthreads() would
// wait for their acknowledgement)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NW>
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
    for (int k = 0; k < NW; ++k) {
        if (k < w) off += scratch[k];
        tot += scratch[k];
    }
    // (no second barrier: the one call per tile is followed by several barriers before `scratch` is written again)
    *total = tot;
    return off + inc - v;
}

// cell code cc and atom code ca -> combined code (grid.hpp img::)
__device__ __forceinline__ int combine_codes(int cc, int ca) { return img::combine(cc, ca); }

// the reference's squared distance of one pair: raw x[j] - wrapped x[i] (neighbor.cpp:164-166), minimum image
// d - L*floor(d/L+0.5) with the image number n taken from the code (box.h:120-124; L*n exact, d - L*0 == d), then
// (dx*dx + dy*dy) + dz*dz (neighbor.cpp:170)
// KIND 0: both atoms in the box and the pair inside it (no image); 1: orthogonal box, image number from the code;
// 2: triclinic box, the reference's fold through fractional coordinates (box.h:99-114) — the image is whatever it rounds to
template <int KIND>
__device__ __forceinline__ double exact_d2(const DBox &b, double xj, double yj, double zj, double xi, double yi, double zi,
                                           int code)
{
    double dx = xj - xi, dy = yj - yi, dz = zj - zi;
    if (KIND == 1) {
        dx = dx - b.h[0] * (double)img::axis(code, 0);
        dy = dy - b.h[4] * (double)img::axis(code, 1);
        dz = dz - b.h[8] * (double)img::axis(code, 2);
    }
    if (KIND == 2)
        pbc<true>(b, dx, dy, dz);
    return dx * dx + dy * dy + dz * dz;
}

// One run of candidates for one centre per lane, hand-scheduled (fixed scratch registers v100-v121).  Four candidates per
// trip: four 16-byte LDS reads in flight, then per candidate 3 subtractions, an FMA chain that ends in e = d2 - c (c a little
// below rc^2), v_alignbit shifting e's SIGN into the hit mask, and an unsigned v_min of e's bits that tracks the smallest
// NON-NEGATIVE e the lane has seen (negative floats are the large unsigned numbers).  Eight instructions, none of which
// touches VCC or an SGPR: e < 0 is a hit for sure, e > W a miss for sure, 0 <= e <= W (decided once per centre from the
// tracked minimum) sends the lane to the double-precision pass.  Lanes whose run is exhausted leave EXEC; the others go on.
// a: LDS byte address of the run's first candidate; rem: its length; bit (L4-1-j) of `mask` is candidate j, L4 = length
// rounded up to 4.  Candidates past the end of a run are other staged atoms: their bits are masked by the caller, a small
// e of theirs only costs a redundant double-precision pass.  (Trips of 8, 12 and 16 candidates with the reads re-issued as
// they are consumed were measured too: what counts is the number of candidate SLOTS a wave walks — 12 for the 10-atom runs
// of the headline lattice with trips of 4 or 12, 16 with 8 or 16 — not the per-trip overhead)
#define MDH_CAND2(X1, Y1, Z1, X2, Y2, Z2)                                                                                            \
    "v_sub_f32 v116, " X1 ", %[sx]\n\t"                                                                                              \
    "v_sub_f32 v119, " X2 ", %[sx]\n\t"                                                                                              \
    "v_sub_f32 v117, " Y1 ", %[sy]\n\t"                                                                                              \
    "v_sub_f32 v120, " Y2 ", %[sy]\n\t"                                                                                              \
    "v_sub_f32 v118, " Z1 ", %[sz]\n\t"                                                                                              \
    "v_sub_f32 v121, " Z2 ", %[sz]\n\t"                                                                                              \
    "v_fma_f32 v116, v116, v116, %[negc]\n\t"                                                                                        \
    "v_fma_f32 v119, v119, v119, %[negc]\n\t"                                                                                        \
    "v_fmac_f32 v116, v117, v117\n\t"                                                                                                \
    "v_fmac_f32 v119, v120, v120\n\t"                                                                                                \
    "v_fmac_f32 v116, v118, v118\n\t"                                                                                                \
    "v_fmac_f32 v119, v121, v121\n\t"                                                                                                \
    "v_alignbit_b32 %[m], %[m], v116, 31\n\t"                                                                                        \
    "v_alignbit_b32 %[m], %[m], v119, 31\n\t"                                                                                        \
    "v_min3_u32 %[w], %[w], v116, v119\n\t"
__device__ __forceinline__ void scan_run_asm(unsigned a, int rem, float sx, float sy, float sz, float negc, unsigned &mask,
                                             unsigned &w)
{
    unsigned long long save;
    unsigned m;
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "v_mov_b32 %[m], 0\n"
                 ".Lscan_top_%=:\n\t"
                 "v_cmp_lt_i32 vcc, 0, %[rem]\n\t"
                 "s_and_b64 exec, exec, vcc\n\t"
                 "s_cbranch_execz .Lscan_end_%=\n\t"
                 "ds_read_b128 v[100:103], %[a]\n\t"
                 "ds_read_b128 v[104:107], %[a] offset:16\n\t"
                 "ds_read_b128 v[108:111], %[a] offset:32\n\t"
                 "ds_read_b128 v[112:115], %[a] offset:48\n\t"
                 "v_add_u32 %[a], 64, %[a]\n\t"
                 "v_add_u32 %[rem], -4, %[rem]\n\t"
                 "s_waitcnt lgkmcnt(2)\n\t" MDH_CAND2("v100", "v101", "v102", "v104", "v105", "v106")
                 "s_waitcnt lgkmcnt(0)\n\t" MDH_CAND2("v108", "v109", "v110", "v112", "v113", "v114")
                 "s_branch .Lscan_top_%=\n"
                 ".Lscan_end_%=:\n\t"
                 "s_mov_b64 exec, %[save]\n\t"
                 : [m] "=&v"(m), [w] "+v"(w), [a] "+v"(a), [rem] "+v"(rem), [save] "=&s"(save)
                 : [sx] "v"(sx), [sy] "v"(sy), [sz] "v"(sz), [negc] "v"(negc)
                 : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110",
                   "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121");
    mask = m;
}
// TWELVE candidate slots of a run, straight-line: no trip counter, no EXEC mask, addresses as instruction offsets; the reads
// of the later slots are issued as the registers of the earlier ones are consumed.  Bit (11 - j) of `mask` is candidate j.
// For tiles whose runs all hold <= 12 candidates (the headline lattice: cells of 1, 2 or 4 atoms, runs of 2 ... 12, and
// with 64 centres in a wave some lane always has a run of 10 or 12: the trip loop of scan_run_asm ran three trips for every
// run anyway and paid a compare, two address updates and the EXEC bookkeeping for each).
__device__ __forceinline__ void scan_run12_asm(unsigned a, float sx, float sy, float sz, float negc, unsigned &mask, unsigned &w)
{
    unsigned m;
    asm volatile("v_mov_b32 %[m], 0\n\t"
                 "ds_read_b128 v[100:103], %[a]\n\t"
                 "ds_read_b128 v[104:107], %[a] offset:16\n\t"
                 "ds_read_b128 v[108:111], %[a] offset:32\n\t"
                 "ds_read_b128 v[112:115], %[a] offset:48\n\t"
                 "s_waitcnt lgkmcnt(2)\n\t" MDH_CAND2("v100", "v101", "v102", "v104", "v105", "v106")
                 "ds_read_b128 v[100:103], %[a] offset:64\n\t"
                 "ds_read_b128 v[104:107], %[a] offset:80\n\t"
                 "s_waitcnt lgkmcnt(2)\n\t" MDH_CAND2("v108", "v109", "v110", "v112", "v113", "v114")
                 "ds_read_b128 v[108:111], %[a] offset:96\n\t"
                 "ds_read_b128 v[112:115], %[a] offset:112\n\t"
                 "s_waitcnt lgkmcnt(2)\n\t" MDH_CAND2("v100", "v101", "v102", "v104", "v105", "v106")
                 "ds_read_b128 v[100:103], %[a] offset:128\n\t"
                 "ds_read_b128 v[104:107], %[a] offset:144\n\t"
                 "s_waitcnt lgkmcnt(2)\n\t" MDH_CAND2("v108", "v109", "v110", "v112", "v113", "v114")
                 "ds_read_b128 v[108:111], %[a] offset:160\n\t"
                 "ds_read_b128 v[112:115], %[a] offset:176\n\t"
                 "s_waitcnt lgkmcnt(2)\n\t" MDH_CAND2("v100", "v101", "v102", "v104", "v105", "v106")
                 "s_waitcnt lgkmcnt(0)\n\t" MDH_CAND2("v108", "v109", "v110", "v112", "v113", "v114")
                 : [m] "=&v"(m), [w] "+v"(w)
                 : [a] "v"(a), [sx] "v"(sx), [sy] "v"(sy), [sz] "v"(sz), [negc] "v"(negc)
                 : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",
                   "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121");
    mask = m;
}
#undef MDH_CAND2
// slots a run of `len` candidates takes in its hit mask: whole trips of four
__device__ __forceinline__ int run_slots(int len)
{
    return (len + 3) & ~3;
}

// IEEE double-precision square root.  For 2^-767 <= x < 2^1000 this is the compiler's own expansion of sqrt(x) (v_rsq_f64
// seed, one Goldschmidt step, two correction steps) minus its input scaling, which is the identity there: bit-identical
// results, seven instructions fewer, and no per-lane branch — the caller asks sqrt_fast_ok() for the whole wave and takes
// the library path when any lane is outside (smaller arguments, 0, inf, NaN).
__device__ __forceinline__ bool sqrt_fast_ok(double x)
{
    return x >= 0x1p-767 && x < 0x1p+1000;
}
__device__ __forceinline__ double sqrt_fast(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d0 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d0, h, g);
    const double d1 = __builtin_fma(-g, g, x);
    return __builtin_fma(d1, h, g);
}

// the same run decided by the reference's double-precision expression (threads with a pair inside the decision band)
template <bool SELF, bool TRI>
__device__ __forceinline__ unsigned scan_run_f64(const double2 *__restrict__ lxy, const double *__restrict__ lz,
                                                 const unsigned short *__restrict__ lsh, const DBox &b, double rcsq, int k0, int len,
                                                 int li, double xi, double yi, double zi)
{
    unsigned m = 0;
    const int S = run_slots(len);
    for (int j = 0; j < S; ++j) {
        const int k = k0 + j;
        bool h = false;
        if (j < len) {
            const double2 cj = lxy[k];
            const double d2 = exact_d2<TRI ? 2 : 1>(b, cj.x, cj.y, lz[k], xi, yi, zi, TRI ? 0 : lsh[k]);
            h = (d2 <= rcsq) && (!SELF || k != li); // neighbor.cpp:162,171
        }
        m = m + m + (h ? 1u : 0u);
    }
    return m;
}

// fixed-cutoff CNA of one centre (FixedCNA, cna.cpp:429-506) from the tile: its NN tickets name the listed neighbours in row
// order, their RAW positions are staged.  plain: no staged atom of this tile carries an image shift, and
// two neighbours of one centre are less than 2 rc < L / 2 apart (>= 7 cells per periodic axis): the minimum image of every
// pair is the plain difference, d - L * 0 == d as the reference computes it.  Tiles at a periodic face classify the spread
// of the positions as k_fcna does; -1: spread too far (atoms handed in outside the box), the atom goes to the to-do list
template <bool TRI, int NN, class Index>
__device__ __forceinline__ int lane_fcna(const DBox &b, Index index_of, const double2 *__restrict__ lxy,
                                         const double *__restrict__ lz, bool plain, double cut2)
{
    double px[NN], py[NN], pz[NN];
#pragma unroll
    for (int a = 0; a < NN; ++a) {
        const int k = index_of(a); // LDS index of the a-th listed neighbour
        const double2 c = lxy[k];
        px[a] = c.x; py[a] = c.y; pz[a] = lz[k];
    }
    Rows R;
    if (TRI) {
        R = bond_rows_reg<true, NN>(b, px, py, pz, cut2);
    } else {
        const int cls = plain ? 2 : span_class3<NN>(b, px, py, pz);
        if (cls == 0)
            return -1;
        R = bond_rows_ortho<NN>(b, px, py, pz, cut2, cls == 2);
    }
    return fcna_label<NN>(R);
}

// The same label with the pair tests in SINGLE precision on the tile's staged coordinates (f4: relative to the tile's corner, every
// staged atom already in the image the tile sees — a difference of two of them IS the minimum image while the box is at least
// seven cells wide, which the tile kernel asks for anyway): no second read of the positions, 36 registers instead of 72, ten
// register-only instructions per pair (cna.hip fcna_atom_f32).  The decision band is the scan's own: both tests compare a squared
// distance of two staged atoms with rc^2 (neighbor.cpp:160 and cna.cpp:459-466 use the same `<=`), and the bound behind (negc, W)
// holds for any two atoms of the 3 x 3 x 3 cells around a centre.  -1: a pair inside the band — the atom goes on the to-do list and
// is finished by the double-precision kernel with the reference's expression (cna.hip k_fcna<TRI, true>), as mdh_fcna does it.
// (the fused form had only ever been measured with the double-precision tests.)
template <int NN, class Index>
__device__ __forceinline__ int lane_fcna_f32(Index index_of, const float4 *__restrict__ f4, float negc, float W)
{
    float ux[NN], uy[NN], uz[NN];
#pragma unroll
    for (int a = 0; a < NN; ++a) {
        const float4 c = f4[index_of(a)];
        ux[a] = c.x; uy[a] = c.y; uz[a] = c.z;
    }
    unsigned adj[NN];
#pragma unroll
    for (int a = 0; a < NN; ++a)
        adj[a] = 0;
    unsigned w = 0x7f7fffffu; // bits of the smallest non-negative d2 - c seen
    pair_tests_f32<NN, NN, 0>(ux, uy, uz, negc, adj, w);
    if (w <= __float_as_uint(W))
        return -1;
    return fcna_label_words<NN, RowsReg<NN>>(adj, RowsReg<NN>(adj)); // (the label fetches no row by a computed index: cna_counts_words)
}

// 4 x 4 transpose of 16-byte pieces among the four lanes of a quad: lane u's piece v <-> lane v's piece u.  Two butterfly
// stages (partner lane u^1, then u^2), each a bitwise select (v_bfi_b32 with a lane-parity mask in a VGPR), one quad_perm DPP
// move and two selects per dword: 64 register-only instructions.  odd1 / odd2: all ones in lanes with bit 0 / bit 1 set.
__device__ __forceinline__ int bit_select(int m, int a, int b) // (a & m) | (b & ~m): one register-only instruction
{
    int d; // (spelled out: from the C expression the compiler derives v_cndmask with an SGPR-pair mask, ~23 cycles apiece here)
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ void quad_transpose(int (&P)[4][4], int odd1, int odd2)
{
#pragma unroll
    for (int pp = 0; pp < 4; pp += 2)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int x = bit_select(odd1, P[pp][w], P[pp + 1][w]);            // odd lanes hand over piece pp, even lanes piece pp+1
            const int y = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
            P[pp + 1][w] = bit_select(odd1, P[pp + 1][w], y);
            P[pp][w] = bit_select(odd1, y, P[pp][w]);
        }
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int x = bit_select(odd2, P[pp][w], P[pp + 2][w]);
            const int y = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
            P[pp + 2][w] = bit_select(odd2, P[pp + 2][w], y);
            P[pp][w] = bit_select(odd2, y, P[pp][w]);
        }
}

// Where a tile's atoms come from.  Records (pk): the cell-sorted 32-byte records of k_gather / k_gather_records — two 16-byte
// reads per atom.  IND (CellGrid::ix, input that comes in some spatial order): no sorted copy exists; atom q of the cell order is
// atom order[q] of the caller's arrays — the id, then x, y, z (and the image code, where any atom has one) of that atom.  The
// staging pays four small gathers per atom, three tiles stage every atom (0.05-0.09 ms at 10 M atoms); the grid build is
// k_gather shorter (0.13 ms, 623 MB written and read back): profiles/r06_cell_grid_ab.txt.
struct IndirectSrc { const double *x, *y, *z; const unsigned short *mv; const int *order; };
__device__ __forceinline__ CellGrid::Packed load_atom(const IndirectSrc &src, int id, int coded)
{
    // the image code only where some atom of the call has one (`coded` is the same in every lane, and nearly always 0: the branch
    // is jumped over, the wait the compiler puts behind a guarded load is never reached)
    int code = (int)img::ATOM_NEUTRAL;
    if (coded) code = src.mv[id];
    return CellGrid::Packed{src.x[id], src.y[id], src.z[id], id, code};
}
struct __attribute__((packed, aligned(4))) Ids4 { int v[4]; }; // four consecutive ids of the cell order as ONE 16-byte request (any 4-byte address)
template <bool IND>
__device__ __forceinline__ CellGrid::Packed load_record(const CellGrid::Packed *__restrict__ pk, const IndirectSrc &src, int q, int coded)
{
    if (!IND)
        return pk[q];
    return load_atom(src, src.order[q], coded);
}

// COUNT: nn and the largest count only (first pass of the exact-width variant)
// parent != nullptr: second pass over the tiles the first pass listed (halo over the LDS budget): the same tiling cut into
// nsub slices along z (this launch's TZ = parent's TZ / nsub); what still does not fit goes to `flagged` (counter
// flags[flag_slot]) and from there to the thread-per-atom code
// TRI: triclinic box (open vectors as in the orthogonal case: no cell beyond the face is staged, atoms outside the box sit in the
// edge cells and are checked against the halo in cell units).  Cells are parallelepipeds in fractional coordinates; the staged
// single-precision coordinates are Cartesian, relative to the tile's corner, of the WRAPPED atom shifted by the lattice
// vectors its cell is away from the tile; every decision inside the band and every written distance goes through the
// reference's fractional fold.
// LOOP: the workgroup walks every (gridDim/8)-th tile of its XCD's chunk, from the jt0-th on (the slice pass; what a tile
// list longer than the host expected leaves over); false: one tile per workgroup, straight-line code (measurably faster:
// 1.15 against 1.23 ms on the headline build — the loop-carried state costs scalar-register spills in every phase)
// FCNA: the fixed-cutoff CNA label of every centre as well (mdh_build_neighbor_fcna)
// TK8 (rows of at most 16 slots): one-byte tickets, no row-word / wrapped-centre tables, rows written by the centre's own lane,
// and (without the fused CNA) the kernel held to 128 VGPRs: a workgroup then needs under 40 KB of LDS and FOUR share a CU.
// false: two-byte tickets and the slot-per-lane write-out (rows of up to 64 slots)
// NW: waves per workgroup.  4: tiles of up to 256 halo cells, four workgroups per CU.  8 (one-byte instance without the fused CNA
// only): tiles of up to 512 halo cells, two workgroups per CU — the same sixteen waves per CU, but a tile of 6 x 6 x 5 cells
// stages 2.5 halo cells per centre cell instead of 3.15, its per-wave front phases serve twice the centres, and its ~440 centres
// fill seven chunks of 64 better than ~196 fill three or four
// IND: the atoms are read through the cell-sorted id list from the caller's arrays (load_record)
template <bool COUNT, bool TRI, bool LOOP, bool FCNA, bool TK8, int NW = 4, bool IND = false>
#ifdef MDH_FCNA_F64
#define MDH_FCNA_LEAN false // the double-precision pair tests of the fused label need ~175 VGPRs: three workgroups per CU
#else
#define MDH_FCNA_LEAN true  // the single-precision ones fit the 128 of the plain instance: four
#endif
// (the walked instances that also label — the slice pass, a tile list longer than its launch: rare — take the registers they want:
// held to 128 they spill a dozen to scratch memory, and a launch that reserves scratch is slower to start even when it finds no work)
__global__ __launch_bounds__(NW * 64, (TK8 && (!FCNA || (MDH_FCNA_LEAN && !LOOP))) ? 16 / NW : 1) void k_neighbor_lane(
    const CellGrid::Packed *__restrict__ pk, const int *__restrict__ cell_start, DBox b,
    Grid g, double rc, float negc, float W, int *__restrict__ verlet, double *__restrict__ dist, int *__restrict__ nn,
    int M, int write_pads, int cap, int *__restrict__ flags, unsigned char *__restrict__ tile_flag, int nt0,
    int nt1, int nt2, Shape ts, const int *__restrict__ tile_list, const int *__restrict__ n_live, int list_mode,
    int *__restrict__ max_count, int *__restrict__ flagged, const int *__restrict__ parent, int parent_nt2, int nsub,
    int flag_slot, int *__restrict__ pattern, int *__restrict__ cna_todo, int jt0, int rw, int tile_base, int *__restrict__ listed_sink,
    int cen_lo, int cen_hi, IndirectSrc isrc)
{
    const int TXY = ts.txy, TZ = ts.tz;
    const int coded = IND ? flags[4] : 0; // (uniform) some atom was handed in outside the box: the image codes are read as well
    const int HXY = TXY + 2, HZ = TZ + 2, NH = HXY * HXY * HZ;
    constexpr int CENC = cen_cap(NW);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // Tickets: (run r << JB) | position of the candidate in its run, decoded where a row is written through the run table hr.
    // Two bytes, or ONE (TK8): five bits of position (a run's mask has 32 slots) and three of run number — run 8 is the last of
    // the walk, so its tickets are the last of the row and a count of them says which they are.
    typedef typename std::conditional<TK8, unsigned char, unsigned short>::type Ticket;
    constexpr int JB = TK8 ? 5 : 8;
    float4 *f4 = reinterpret_cast<float4 *>(smem);                  // [cap] staged (ux, uy, uz, bits of the atom id); the scan reads up to 11 entries past a run's end: into lxy, masked
    double2 *lxy = reinterpret_cast<double2 *>(f4 + cap);           // [cap] staged raw x, y
    double *lz = reinterpret_cast<double *>(lxy + cap);             // [cap] staged raw z
    unsigned *cen = reinterpret_cast<unsigned *>(lz + cap);         // [CEN_CAP] centre atoms: LDS index | halo cell << 11
    unsigned short *lsh = reinterpret_cast<unsigned short *>(cen + CENC); // [cap] combined image code of a staged atom seen from this tile
    Ticket *tk = reinterpret_cast<Ticket *>(lsh + cap + (cap & 1)); // [NT][TKS] tickets (slot M swallows the hits past M); wave w owns rows 64 w ...
    const unsigned f4_lds = (unsigned)(unsigned long)(lds_byte *)smem;
    // halo cell: population — needed from the block scan's barrier to the run table only, so it lives in the ticket rows, cell t
    // in the rows of wave t / 64 (the wave that writes it: a wave walking a list of tiles may be a tile ahead of the others,
    // whose tickets it must not touch)
    const int TKS = ticket_row(M, false);
    const int wstride = max(rw * TKS, 256 / (int)sizeof(Ticket)); // tickets of one wave's rows (at least its 64 populations)
    auto hc = [&](int t) -> unsigned & { return reinterpret_cast<unsigned *>(tk + (size_t)(t >> 6) * wstride)[t & 63]; };
    __shared__ unsigned hr[NW * 64 + 2]; // 3-cell z-run centred on the cell: LDS offset | length << 16
    __shared__ int scan_tmp[NW];
    __shared__ int s_flag[4];

    const double rcsq = rc * rc, pad = rc + 1.0; // neighbor.cpp:127; pads neighbor.py:125-129
    const double cw = rc;                        // cell width of the rc-wide grid (neighbor.cpp:29-62)
    int vmax = 0;

    // XCD-aware tile order: block b runs on XCD b%8; every XCD gets one contiguous chunk of the tiles THAT HOLD CENTRE ATOMS
    // (neighbouring tiles share halo cells through the same L2; vacuum leaves no XCD idle)
    const int nlive = parent ? min(*n_live, nt0 * nt1 * parent_nt2) * nsub : ((list_mode && tile_list) ? *n_live : nt0 * nt1 * nt2);
    const int per = (nlive + 7) / 8;
    if (parent && listed_sink && blockIdx.x == 0 && tid == 0)
        *listed_sink = min(*n_live, nt0 * nt1 * parent_nt2); // (pinned host memory: the next build sizes this pass's grid with it)
    struct Tile { int id, T0, T1, T2; bool empty; };
    struct Halo { int cnt, src, img, hz; bool edge, centre; };
    auto tile_of = [&](int slot) {
        Tile t;
        int t0, t1, t2;
        if (parent) {
            const int pt = parent[slot / nsub]; // tile of the first pass
            t2 = (pt % parent_nt2) * nsub + slot % nsub;
            t1 = (pt / parent_nt2) % nt1;
            t0 = pt / (parent_nt2 * nt1);
            t.id = (t0 * nt1 + t1) * nt2 + t2;
        } else {
            t.id = (list_mode && tile_list) ? tile_list[slot] : slot + tile_base; // (tile_base: first tile of a window of the grid, else 0)
            const unsigned col = ts.by_nt2.div((unsigned)t.id);
            t2 = t.id - (int)col * nt2;
            t0 = (int)ts.by_nt1.div(col);
            t1 = (int)col - t0 * nt1;
        }
        t.T0 = t0 * TXY; t.T1 = t1 * TXY; t.T2 = t2 * TZ;
        t.empty = t.T2 >= g.nc[2]; // (a slice beyond the grid: the parent tile was a clipped one)
        return t;
    };
    // halo cell of this thread: source range, image code, is it a centre cell
    auto halo_of = [&](const Tile &t) {
        Halo h{0, 0, NEUTRAL, 0, false, false}; // edge: first / last cell of an open axis (atoms outside the box are clamped into it, neighbor.cpp:58-61)
        if (tid < NH && !t.empty) {
            const int hcol = (int)(((unsigned)tid * ts.mhz) >> 16); // tid / HZ
            h.hz = tid - hcol * HZ;
            const int hx = (int)(((unsigned)hcol * ts.mhxy) >> 16), hy = hcol - hx * HXY;
            const int g0 = t.T0 + hx - 1, g1 = t.T1 + hy - 1, g2 = t.T2 + h.hz - 1;
            // a cell beyond an OPEN face is the far side of the box in the reference's modulo walk (neighbor.cpp:18-27); with
            // >= 4 cells on the axis its atoms are >= 2 rc from every centre of this tile: no hits, not staged
            const bool in0 = b.pbc[0] ? (g0 >= -1 && g0 <= g.nc[0]) : (g0 >= 0 && g0 < g.nc[0]);
            const bool in1 = b.pbc[1] ? (g1 >= -1 && g1 <= g.nc[1]) : (g1 >= 0 && g1 < g.nc[1]);
            const bool in2 = b.pbc[2] ? (g2 >= -1 && g2 <= g.nc[2]) : (g2 >= 0 && g2 < g.nc[2]);
            if (in0 && in1 && in2) {
                const int a0 = g0 < 0 ? g0 + g.nc[0] : (g0 >= g.nc[0] ? g0 - g.nc[0] : g0);
                const int a1 = g1 < 0 ? g1 + g.nc[1] : (g1 >= g.nc[1] ? g1 - g.nc[1] : g1);
                const int a2 = g2 < 0 ? g2 + g.nc[2] : (g2 >= g.nc[2] ? g2 - g.nc[2] : g2);
                const int c = (a0 * g.nc[1] + a1) * g.nc[2] + a2; // (the host refuses grids of 2^31 cells or more)
                h.src = cell_start[c];
                h.cnt = cell_start[c + 1] - h.src;
                // image of the candidate cell seen from an in-grid centre cell: below the box -> raw coordinates are ~+L
                // away (n = +1); above -> n = -1
                const int n0 = g0 < 0 ? 1 : (g0 >= g.nc[0] ? -1 : 0);
                const int n1 = g1 < 0 ? 1 : (g1 >= g.nc[1] ? -1 : 0);
                const int n2 = g2 < 0 ? 1 : (g2 >= g.nc[2] ? -1 : 0);
                h.img = (n0 + 1) | ((n1 + 1) << 2) | ((n2 + 1) << 4);
                h.edge = (!b.pbc[0] && (a0 == 0 || a0 == g.nc[0] - 1)) || (!b.pbc[1] && (a1 == 0 || a1 == g.nc[1] - 1)) ||
                         (!b.pbc[2] && (a2 == 0 || a2 == g.nc[2] - 1));
                // (cen_lo, cen_hi: the planes of axis 0 whose atoms want rows — all of them, or the planes of a decomposed system's own slab:
                // the ghost planes around it are staged as candidates, never scanned as centres)
                h.centre = hx >= 1 && hx <= TXY && hy >= 1 && hy <= TXY && h.hz >= 1 && h.hz <= TZ && g0 < g.nc[0] && g1 < g.nc[1] && g2 < g.nc[2] &&
                           g0 >= cen_lo && g0 < cen_hi;
            }
        }
        return h;
    };
    // the first four atoms of a cell (twenty independent loads), requested BEFORE the workgroup scan: their latency overlaps
    // the scan's two barriers instead of following them (the loads do not need the LDS offsets, only the stores do)
    // (unconditional loads, nothing computed from them here: a branch around them, or a narrowing of the code, makes the compiler
    // copy registers behind an s_waitcnt right after the loads, and the latency is paid before the scan after all; a thread
    // without atoms loads record 0 and never looks at it)
    auto request = [&](const Halo &h, double (&ra)[4], double (&rb)[4], double (&rc4)[4], int (&rd)[4], int (&rm)[4]) {
        Ids4 ids{};
        if (IND) ids = *reinterpret_cast<const Ids4 *>(isrc.order + h.src); // (order[] has four spare entries behind its last atom)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int q = h.cnt > 0 ? h.src + min(v, h.cnt - 1) : 0;
            // (IND: a slot past the cell's end repeats its first atom; a cell without atoms reads atom 0)
            const CellGrid::Packed a = IND ? load_atom(isrc, v < h.cnt ? ids.v[v] : (h.cnt > 0 ? ids.v[0] : 0), coded) : load_record<false>(pk, isrc, q, coded);
            ra[v] = a.x; rb[v] = a.y; rc4[v] = a.z; rd[v] = a.id; rm[v] = a.code;
        }
    };
    for (int jt = jt0 + (int)(blockIdx.x >> 3); jt < per; jt += (int)(gridDim.x >> 3)) {
        const int slot = (int)(blockIdx.x & 7) * per + jt;
        if (slot >= nlive)
            break;
        const Tile tile = tile_of(slot);
        STAMP(0);
        Halo cur = halo_of(tile);
#ifdef MDH_STAMPS
        asm volatile("" : "+v"(cur.src), "+v"(cur.cnt));
        STAMP(1);
#endif
        double pa[4], pb[4], pc[4];
        int pd[4], pm[4];
        request(cur, pa, pb, pc, pd, pm);
        if (flags[0] != 0) // unwrapped input: the image codes are not valid, the thread-per-atom kernel takes the whole call (asked
            return;        // here, behind the tile's first loads: the flag is two dependent scalar loads away)
        const int tile_id = tile.id, T0 = tile.T0, T1 = tile.T1, T2 = tile.T2;
        const int cnt = cur.cnt, src = cur.src, img = cur.img, hz = cur.hz;
        const bool edge = cur.edge, centre_cell = cur.centre;
        if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; s_flag[2] = 0; s_flag[3] = 0; }
        hc(tid) = (unsigned)cnt; // the neighbours in z need it for their run (published by the scan's barrier)

        int total2;
        const int off2 = excl_scan_block<NW>(cnt | (centre_cell ? cnt << 16 : 0), scan_tmp, &total2); // both prefixes in one scan (each < 2^15)
        const int total = total2 & 0xffff, ncentres = total2 >> 16;
        const int off0 = off2 & 0xffff, coff = off2 >> 16;
        bool ok = !(total > cap || ncentres > CENC); // else: listed for the next pass
        const bool no_centres = ncentres == 0;          // (uniform) a tile of ghost planes only: nothing to stage, nothing to list
        if (no_centres) ok = false;
        STAMP(2);
        // the 3-cell run around every cell that can be a column entry of a centre's walk (cells tid-1, tid, tid+1 are adjacent in z
        // and in LDS)
        if (ok && tid < NH && hz >= 1 && hz <= HZ - 2) {
            const unsigned k0 = (unsigned)off0 - hc(tid - 1), len = hc(tid - 1) + (unsigned)cnt + hc(tid + 1);
            hr[tid] = k0 | (len << 16);
            if (len > (TK8 ? 32u : 96u)) s_flag[2] = 1; // a run's hit mask is one 32-bit register (length rounded up to 4); three in the wide instance
            if (TK8 && len > 12u) s_flag[3] = 1;        // no short-run scan for this tile (eight slots + up to four leftovers per run)
        }
        // ---- stage this cell's atoms
        if (ok && cnt > 0) {
            double X0, Y0, Z0, XS, YS, ZS;
            const int code0 = combine_codes(img, img::ATOM_NEUTRAL); // an atom inside the box (image number 0): the cell's own shift
            if (TRI) { // corner of the tile's halo and the cell's lattice shift, through the cell vectors (rows of h)
                // (cell k of axis d starts at the fraction k * rc / thickness_d: cells of perpendicular width rc, grid.hpp)
                const double f0 = (double)(T0 - 1) * (cw / b.thick[0]) + (double)img::axis(code0, 0);
                const double f1 = (double)(T1 - 1) * (cw / b.thick[1]) + (double)img::axis(code0, 1);
                const double f2 = (double)(T2 - 1) * (cw / b.thick[2]) + (double)img::axis(code0, 2);
                XS = b.o[0] + f0 * b.h[0] + f1 * b.h[3] + f2 * b.h[6];
                YS = b.o[1] + f0 * b.h[1] + f1 * b.h[4] + f2 * b.h[7];
                ZS = b.o[2] + f0 * b.h[2] + f1 * b.h[5] + f2 * b.h[8];
                X0 = XS; Y0 = YS; Z0 = ZS;
            } else {
                X0 = b.o[0] + (double)(T0 - 1) * cw; Y0 = b.o[1] + (double)(T1 - 1) * cw; Z0 = b.o[2] + (double)(T2 - 1) * cw;
                XS = X0 + b.h[0] * (double)img::axis(code0, 0); YS = Y0 + b.h[4] * (double)img::axis(code0, 1);
                ZS = Z0 + b.h[8] * (double)img::axis(code0, 2);
            }
            const float flo = (float)(-1.5 * cw);
            const float fhx = (float)(((double)HXY + 1.5) * cw), fhz = (float)(((double)HZ + 1.5) * cw);
            bool general = code0 != NEUTRAL3, far = false;
            // four atoms of the cell: into the staging arrays
            auto stage4 = [&](int k, const double (&a)[4], const double (&bb)[4], const double (&c)[4], const int (&d)[4], const int (&m)[4]) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (k + v < cnt) {
                        int code = code0;
                        float ux, uy, uz;
                        if (TRI) { // the atom as the cell grid saw it (wrapped along the periodic vectors), in the tile's frame
                            double wx = a[v], wy = bb[v], wz = c[v];
                            wrap<true>(b, wx, wy, wz);
                            const double rx = wx - XS, ry = wy - YS, rz = wz - ZS;
                            ux = (float)rx; uy = (float)ry; uz = (float)rz;
                            if (edge) { // first / last cell of an OPEN vector: an atom clamped into it from outside the box may be anywhere —
                                // its place in units of cells along the three vectors, counted from the halo's corner
                                const double s0 = (rx * b.hi[0] + ry * b.hi[3] + rz * b.hi[6]) * (b.thick[0] / cw);
                                const double s1 = (rx * b.hi[1] + ry * b.hi[4] + rz * b.hi[7]) * (b.thick[1] / cw);
                                const double s2 = (rx * b.hi[2] + ry * b.hi[5] + rz * b.hi[8]) * (b.thick[2] / cw);
                                far = far || !(s0 >= -1.5 && s0 <= (double)HXY + 1.5 && s1 >= -1.5 && s1 <= (double)HXY + 1.5 && s2 >= -1.5 && s2 <= (double)HZ + 1.5);
                            }
                        } else {
                            ux = (float)(a[v] - XS); uy = (float)(bb[v] - YS); uz = (float)(c[v] - ZS);
                        }
                        if (!TRI && m[v] != img::ATOM_NEUTRAL) { // an atom handed in outside the box on a periodic axis: its own image number on top
                            code = combine_codes(img, m[v]);
                            ux = (float)((a[v] - b.h[0] * (double)img::axis(code, 0)) - X0);
                            uy = (float)((bb[v] - b.h[4] * (double)img::axis(code, 1)) - Y0);
                            uz = (float)((c[v] - b.h[8] * (double)img::axis(code, 2)) - Z0);
                            general = true;
                        }
                        // the decision band assumes coordinates inside the tile's halo; an atom clamped into an edge cell from
                        // far outside the box (open axis) sends the tile to the thread-per-atom code
                        if (!TRI && edge)
                            far = far || !(ux >= flo && ux <= fhx && uy >= flo && uy <= fhx && uz >= flo && uz <= fhz);
                        const int p = off0 + k + v;
#ifdef MDH_EXP_GATHER // measuring build (make gather): the staged atom carries its record's index, not its id
                        f4[p] = make_float4(ux, uy, uz, __int_as_float(src + k + v));
#else
                        f4[p] = make_float4(ux, uy, uz, __int_as_float(d[v]));
#endif
                        lxy[p] = make_double2(a[v], bb[v]);
                        lz[p] = c[v];
                        lsh[p] = (unsigned short)code;
                        if (centre_cell) cen[coff + k + v] = (unsigned)p | ((unsigned)tid << 11);
                    }
                }
            };
            if (TK8) { // cells of the one-byte instance rarely hold more than the four atoms requested in front of the scan
                for (int k = 0; k < cnt; k += 4) {
                    double a[4], bb[4], c[4];
                    int d[4], m[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) { // twenty independent loads in flight
                        if (k == 0) { a[v] = pa[v]; bb[v] = pb[v]; c[v] = pc[v]; d[v] = pd[v]; m[v] = pm[v]; }
                        else {
                            const int q = src + min(k + v, cnt - 1);
                            const CellGrid::Packed r = load_record<IND>(pk, isrc, q, coded);
                            a[v] = r.x; bb[v] = r.y; c[v] = r.z; d[v] = r.id; m[v] = r.code;
                        }
                    }
                    stage4(k, a, bb, c, d, m);
                }
            } else { // dense cells (ten atoms and more): the next four atoms travel while these four are staged (the wide instance
                     // runs two workgroups per CU: registers to spare)
                double a[4], bb[4], c[4];
                int d[4], m[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) { a[v] = pa[v]; bb[v] = pb[v]; c[v] = pc[v]; d[v] = pd[v]; m[v] = pm[v]; }
                for (int k = 0; k < cnt; k += 4) {
                    double na[4], nb[4], nc[4];
                    int nd[4], nm[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) { // (unconditional: past the cell's end the last atom again, never staged)
                        const int q = src + min(k + 4 + v, cnt - 1);
                        const CellGrid::Packed r = load_record<IND>(pk, isrc, q, coded);
                        na[v] = r.x; nb[v] = r.y; nc[v] = r.z; nd[v] = r.id; nm[v] = r.code;
                    }
                    stage4(k, a, bb, c, d, m);
#pragma unroll
                    for (int v = 0; v < 4; ++v) { a[v] = na[v]; bb[v] = nb[v]; c[v] = nc[v]; d[v] = nd[v]; m[v] = nm[v]; }
                }
            }
            if (general) s_flag[0] = 1;
            if (far) s_flag[1] = 1;
        }
        __syncthreads(); // publishes the run table, the staged atoms, the centre list and the flags
        STAMP(3);
        if (ok && (s_flag[1] | s_flag[2]))
            ok = false;
        if (!ok && !no_centres && tid == 0) { // list this tile for the next pass
            if (tile_flag) tile_flag[tile_id] = 1;
            flagged[atomicAdd(&flags[flag_slot], 1)] = tile_id;
        }
        const bool general_tile = s_flag[0] != 0;
        const bool short_runs = TK8 && s_flag[3] == 0; // every run of the tile holds <= 12 candidates
        // ---- From here on the four waves do not meet again: a wave takes whole chunks of the tile's centres, scans
        // them, leaves their tickets in ITS rows of tk and writes their rows itself — LDS traffic inside one wave is ordered,
        // no workgroup barrier — so one wave's scan overlaps another's write-out.
        const int wv = tid >> 6;
        Ticket *tkw = tk + (size_t)wv * wstride; // rw: rows a wave works on at a time (64; fewer where rows are long and centres few: dense cells)
        const int A2 = HXY * HZ;
        // halo cell of run r relative to the centre's cell: ((r / 3 - 1) * HXY + (r % 3 - 1)) * HZ, neighbor.cpp:147-151
        auto run_cell = [&](int cbv, int r) {
            const int r3 = (r * 11) >> 5;
            return cbv + __mul24(r3, A2 - 3 * HZ) + __mul24(r, HZ) - (A2 + HZ);
        };
        // Centres go out in chunks of rw (a full wave of them); chunk c is taken by wave (c + jt) mod 4 — a tile of 150 centres
        // keeps three waves busy with full registers and leaves the fourth free, and the rotation spreads the free one over the
        // CU's four SIMDs from tile to tile (an even split, 38 lanes in each of four waves, costs four chunk passes for three)
        for (int cbase = (int)((unsigned)(wv - jt) & (unsigned)(NW - 1)) * rw; ok && cbase < ncentres; cbase += NW * rw) {
            const int q = cbase + lane;
            const bool mine = lane < rw && q < ncentres; // this lane holds a centre
            int kept = 0, kept8 = 0, cb = 0, id = 0; // (lanes without a centre: no row)
            double xi = 0, yi = 0, zi = 0;
            Ticket *my = tkw + (size_t)lane * TKS;
            if (mine) {
                const unsigned cv = cen[q];
                const int li = (int)(cv & 2047u);
                cb = (int)(cv >> 11);
                const float4 s = f4[li];
                // WIDE (the two-byte instance): runs of up to 96 candidates — cells of ten atoms and more — as three masks, the
                // first 32 candidates of a run in mk, the next 32 in mk2, the rest in mk3 (the last cell of an axis takes the
                // remainder of the box, neighbor.cpp:58-61, and holds up to twice the atoms of the others: with two masks the
                // tiles along three faces of a 3.4 M-atom box at rc = 5 A went to the slice pass and on to the thread-per-atom
                // code, 0.35 of 1.98 ms)
                constexpr bool WIDE = !TK8;
                unsigned hv[9], mk[9], mk2[WIDE ? 9 : 1], mk3[WIDE ? 9 : 1];
#pragma unroll
                for (int r = 0; r < 9; ++r) // neighbor.cpp:147-151: r = (da+1)*3 + (db+1)
                    hv[r] = hr[cb + ((r / 3 - 1) * HXY + (r % 3 - 1)) * HZ];
                unsigned w = 0x7f7fffffu; // bits of the smallest non-negative d2 - c this centre has seen
                // TK8: the masks are kept TOP-ALIGNED (bit 31 - j = candidate j of the run)
                if (TK8 && short_runs) {
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        const int len = (int)(hv[r] >> 16);
                        scan_run12_asm(f4_lds + ((hv[r] & 0xffffu) << 4), s.x, s.y, s.z, negc, mk[r], w);
                        mk[r] = (mk[r] << 20) & (~0u << ((32 - len) & 31)) & (len > 0 ? ~0u : 0u); // slots past the end of the run
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        const int len = (int)(hv[r] >> 16), la = WIDE ? min(len, 32) : len;
                        scan_run_asm(f4_lds + ((hv[r] & 0xffffu) << 4), la, s.x, s.y, s.z, negc, mk[r], w);
                        mk[r] &= ~0u << (run_slots(la) - la); // slots past the end of the run
                        if (TK8) mk[r] <<= (32 - run_slots(la)) & 31; // (an empty run: mask 0, shift 0)
                        if (WIDE) {
                            const int lb = min(len - la, 32), lc = len - la - lb;
                            mk2[r] = 0;
                            mk3[r] = 0;
                            if (__builtin_amdgcn_ballot_w64(lb > 0)) {
                                scan_run_asm(f4_lds + (((hv[r] & 0xffffu) + 32u) << 4), lb, s.x, s.y, s.z, negc, mk2[r], w);
                                mk2[r] &= lb > 0 ? ~0u << (run_slots(lb) - lb) : 0u;
                                if (__builtin_amdgcn_ballot_w64(lc > 0)) {
                                    scan_run_asm(f4_lds + (((hv[r] & 0xffffu) + 64u) << 4), lc, s.x, s.y, s.z, negc, mk3[r], w);
                                    mk3[r] &= lc > 0 ? ~0u << (run_slots(lc) - lc) : 0u;
                                }
                            }
                        }
                    }
                }
                {   // the centre itself sits in run 4 with d2 = 0: not a neighbour (neighbor.cpp:162)
                    const int len = (int)(hv[4] >> 16), idx = li - (int)(hv[4] & 0xffffu);
                    if (TK8) mk[4] &= ~(0x80000000u >> idx);
                    else if (idx < 32) mk[4] &= ~(1u << (run_slots(min(len, 32)) - 1 - idx));
                    else if (idx < 64) mk2[4] &= ~(1u << (run_slots(min(len - 32, 32)) - 1 - (idx - 32)));
                    else mk3[4] &= ~(1u << (run_slots(len - 64) - 1 - (idx - 64)));
                }
                STAMP(4);
                {
                    const double2 ci = lxy[li];
                    xi = ci.x; yi = ci.y; zi = lz[li];
                    if (b.anypbc) { // neighbor.cpp:139-142
                        if (TRI || general_tile) { // an atom may have been handed in outside the box: the whole expression
                            wrap<TRI>(b, xi, yi, zi);
                        } else { // every staged atom lies inside the box: floor((x - o) / L) = 0, the wrap is o + (x - o) - L * 0 (box.h:158-176)
                            if (b.pbc[0]) xi = b.o[0] + (xi - b.o[0]);
                            if (b.pbc[1]) yi = b.o[1] + (yi - b.o[1]);
                            if (b.pbc[2]) zi = b.o[2] + (zi - b.o[2]);
                        }
                    }
                }
                if (__builtin_expect(w <= __float_as_uint(W), 0)) { // a pair inside the decision band: this centre again in double precision
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        const int k0 = (int)(hv[r] & 0xffffu), len = (int)(hv[r] >> 16), la = WIDE ? min(len, 32) : len;
                        if (r == 4) mk[r] = scan_run_f64<true, TRI>(lxy, lz, lsh, b, rcsq, k0, la, li, xi, yi, zi);
                        else mk[r] = scan_run_f64<false, TRI>(lxy, lz, lsh, b, rcsq, k0, la, li, xi, yi, zi);
                        if (TK8) mk[r] <<= (32 - run_slots(la)) & 31;
                        if (WIDE) {
                            const int lb = min(len - la, 32), lc = len - la - lb;
                            if (r == 4) mk2[r] = scan_run_f64<true, TRI>(lxy, lz, lsh, b, rcsq, k0 + 32, lb, li, xi, yi, zi);
                            else mk2[r] = scan_run_f64<false, TRI>(lxy, lz, lsh, b, rcsq, k0 + 32, lb, li, xi, yi, zi);
                            if (r == 4) mk3[r] = scan_run_f64<true, TRI>(lxy, lz, lsh, b, rcsq, k0 + 64, lc, li, xi, yi, zi);
                            else mk3[r] = scan_run_f64<false, TRI>(lxy, lz, lsh, b, rcsq, k0 + 64, lc, li, xi, yi, zi);
                        }
                    }
                }
                int hits = 0;
#pragma unroll
                for (int r = 0; r < 9; ++r) hits += __builtin_popcount(mk[r]) + (WIDE ? __builtin_popcount(mk2[r]) + __builtin_popcount(mk3[r]) : 0);
#ifdef MDH_EXP_GATHER
                id = pk[__float_as_int(s.w)].id;
#else
                id = __float_as_int(s.w);
#endif
                nn[id] = hits; // keeps counting past M (neighbor.cpp:172-177)
                if (COUNT) {
                    vmax = max(vmax, hits);
                } else {
                    // masks -> tickets in walk order: bit (S-1-j) of a run's mask is its candidate j (S = 32 in the one-byte
                    // instance: top-aligned).  Branch-free steps: a step
                    // of a run that has no hit left writes into the row's spare slot M (v_ffbh of 0 is -1: the slot index
                    // saturates, the cleared bit is one of an empty mask).  Three steps per run cover nearly every run; a loop
                    // takes what is left.  (With a branch per step — v_cmp, s_and_saveexec, s_cbranch — the 36 steps of a centre
                    // were 13 % of a tile's time.)
                    int sl = 0;
                    auto step = [&](unsigned &m, int jb) {
                        int z;
                        asm("v_ffbh_u32 %0, %1" : "=v"(z) : "v"(m)); // leading zeros; -1 for 0
                        const int none = z >> 31;                      // -1: no hit left in this run
                        const unsigned slot = min((unsigned)(sl | none), (unsigned)M);
                        my[slot] = (Ticket)(jb + z);
                        sl += none + 1;
                        m &= ~(0x80000000u >> (z & 31));
                    };
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        const int len = (int)(hv[r] >> 16), la = WIDE ? min(len, 32) : len;
                        const int jb = TK8 ? ((r & 7) << JB) : (r << JB) + (run_slots(la) - 32); // + clz(m) = the hit's ticket
                        unsigned m = mk[r];
                        step(m, jb);
                        step(m, jb);
                        step(m, jb);
                        while (__builtin_amdgcn_ballot_w64(m != 0)) step(m, jb);
                        if (WIDE) { // candidates 32.. of the run: position 32 + j; 64..: position 64 + j
                            const int lb = min(len - la, 32), lc = len - la - lb;
                            unsigned m2 = mk2[r], m3 = mk3[r];
                            const int jb2 = (r << JB) + run_slots(lb), jb3 = (r << JB) + 32 + run_slots(lc);
                            while (__builtin_amdgcn_ballot_w64(m2 != 0)) step(m2, jb2);
                            while (__builtin_amdgcn_ballot_w64(m3 != 0)) step(m3, jb3);
                        }
                    }
                    {   // listed hits, and how many of them belong to run 8 (the last ones of the row)
                        const int n8 = __builtin_popcount(mk[8]);
                        kept = hits < M ? hits : M;
                        kept8 = max(0, min(n8, kept - (hits - n8)));
                    }
                    if (FCNA) { // atoms without 12 or 14 neighbours keep the caller's value (cna.cpp:456)
                        auto index_of = [&](int a) {
                            const unsigned t = my[a];
                            const int r = (TK8 && a >= kept - kept8) ? 8 : (int)(t >> JB);
                            return (int)(hr[run_cell(cb, r)] & 0xffffu) + (int)(t & ((1u << JB) - 1u));
                        };
                        int label = 0;
#ifdef MDH_FCNA_F64 // measuring build (make fcna64): the pair tests in double precision on the raw coordinates, as until round 5
                        if (hits == 12 && M >= 12) label = lane_fcna<TRI, 12>(b, index_of, lxy, lz, !general_tile, rcsq);
                        else if (hits == 14 && M >= 14) label = lane_fcna<TRI, 14>(b, index_of, lxy, lz, !general_tile, rcsq);
#else
                        if (hits == 12 && M >= 12) label = lane_fcna_f32<12>(index_of, f4, negc, W);
                        else if (hits == 14 && M >= 14) label = lane_fcna_f32<14>(index_of, f4, negc, W);
#endif
                        if (label > 0) pattern[id] = label;
                        else if (label < 0) defer(cna_todo, id);
                    }
                }
            }
            if (TK8 && !COUNT) {
                // ---- M <= 16: the centre's lane works out its own row — every listed neighbour's distance, in registers — and
                // writes it with 16-byte stores: no row word, no per-slot copy of the centre, a third of the instructions of the
                // slot-per-lane write-out below.  The tickets go through LDS only to turn a per-lane slot number into a register
                // number: written at slot `sl`, read back as the row's four words.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                STAMP(5);
                int maxk = kept;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) maxk = max(maxk, __shfl_xor(maxk, d, 64));
                const unsigned *myw = reinterpret_cast<const unsigned *>(my);
                unsigned tw[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) tw[v] = (4 * v < M) ? myw[v] : 0u;
                int idv[16];
                double dv[16];
                // four slots at a time: their run-table reads go out together, then their twelve position reads, then four
                // independent distance chains — two LDS round trips per group instead of two per slot
#pragma unroll
                for (int g0 = 0; g0 < 16; g0 += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        idv[g0 + u] = -1; // pads neighbor.py:125-129
                        dv[g0 + u] = pad;
                    }
                    if (g0 < maxk) { // (uniform)
                        int k[4];
                        bool h[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int sx = g0 + u;
                            h[u] = sx < kept;
                            const unsigned t = (tw[sx >> 2] >> (8 * (sx & 3))) & 255u;
                            int r = (sx >= kept - kept8) ? 8 : (int)(t >> 5);
                            r = h[u] ? r : 4; // a lane whose row is shorter: the first atom of its own run, result unused
                            k[u] = (int)(hr[run_cell(cb, r)] & 0xffffu) + (h[u] ? (int)(t & 31u) : 0);
                        }
                        double2 cj[4];
                        double zj[4], d2[4];
                        int nid[4], sh[4];
#ifdef MDH_EXP_GATHER // the hits' raw doubles through L2 in one batched gather per group of four
                        int qx[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) qx[u] = mine ? __float_as_int(f4[k[u]].w) : 0; // (a lane without a centre reads whatever LDS holds)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const CellGrid::Packed rj = pk[qx[u]];
                            cj[u] = make_double2(rj.x, rj.y);
                            zj[u] = rj.z;
                            nid[u] = rj.id;
                            sh[u] = (!TRI && general_tile) ? (int)lsh[k[u]] : 0;
                        }
#else
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            cj[u] = lxy[k[u]];
                            zj[u] = lz[k[u]];
                            nid[u] = __float_as_int(f4[k[u]].w);
                            sh[u] = (!TRI && general_tile) ? (int)lsh[k[u]] : 0;
                        }
#endif
                        bool slow = false;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (TRI) d2[u] = exact_d2<2>(b, cj[u].x, cj[u].y, zj[u], xi, yi, zi, 0);
                            else if (general_tile) d2[u] = exact_d2<1>(b, cj[u].x, cj[u].y, zj[u], xi, yi, zi, sh[u]);
                            else d2[u] = exact_d2<0>(b, cj[u].x, cj[u].y, zj[u], xi, yi, zi, 0);
                            d2[u] = h[u] ? d2[u] : 1.0;
                            slow = slow || !sqrt_fast_ok(d2[u]);
                        }
                        double rr[4]; // neighbor.cpp:174 — the four roots as one straight-line block: their dependent chains interleave
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) == 0, 1)) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) rr[u] = sqrt_fast(d2[u]);
                        } else {
#pragma unroll
                            for (int u = 0; u < 4; ++u) rr[u] = sqrt(d2[u]);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            idv[g0 + u] = h[u] ? nid[u] : -1;
                            dv[g0 + u] = h[u] ? rr[u] : pad;
                        }
                    }
                }
                // ---- the rows go out.  Whole rows in 16-byte pieces: first the four lanes of a quad exchange pieces (lane u gets
                // piece u of each of the quad's four rows), so that one store instruction writes 64 consecutive bytes per quad —
                // 16 line requests per instruction instead of 64: with every lane storing into its own row the address unit
                // of the CU, at about two cycles per request, was a quarter of the kernel's time.
                {
                    const int rid = mine ? id : -1;
#ifdef MDH_EXP_SCATTER // measuring build (make scatter): every lane stores its own row, 16 bytes at a time (64 line requests per store instruction)
                    if (write_pads && (M & 3) == 0) {
                        if (rid >= 0) {
                            const uint64_t row = (uint64_t)(unsigned)id * (unsigned)M;
#pragma unroll
                            for (int pp = 0; pp < 4; ++pp)
                                if (4 * pp < M)
                                    *reinterpret_cast<Int4 *>(verlet + row + 4 * pp) = Int4{idv[4 * pp], idv[4 * pp + 1], idv[4 * pp + 2], idv[4 * pp + 3]};
#pragma unroll
                            for (int piece = 0; piece < 8; ++piece)
                                if (2 * piece < M)
                                    *reinterpret_cast<Int4 *>(dist + row + 2 * piece) = Int4{__double2loint(dv[2 * piece]), __double2hiint(dv[2 * piece]),
                                                                                              __double2loint(dv[2 * piece + 1]), __double2hiint(dv[2 * piece + 1])};
                        }
                    } else
#endif
                    if (write_pads && (M & 3) == 0) {
                        const int odd1 = -(lane & 1), odd2 = -((lane >> 1) & 1), u4 = lane & 3;
                        int P[4][4];
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                            for (int w = 0; w < 4; ++w) P[pp][w] = idv[4 * pp + w];
                        quad_transpose(P, odd1, odd2);
                        int rq[4]; // the quad's four row numbers
                        rq[0] = __builtin_amdgcn_mov_dpp(rid, 0x00, 0xF, 0xF, true); // quad_perm [v,v,v,v]
                        rq[1] = __builtin_amdgcn_mov_dpp(rid, 0x55, 0xF, 0xF, true);
                        rq[2] = __builtin_amdgcn_mov_dpp(rid, 0xAA, 0xF, 0xF, true);
                        rq[3] = __builtin_amdgcn_mov_dpp(rid, 0xFF, 0xF, 0xF, true);
                        if (4 * u4 < M) {
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                if (rq[v] >= 0)
                                    *reinterpret_cast<Int4 *>(verlet + (uint64_t)(unsigned)rq[v] * (unsigned)M + 4 * u4) = Int4{P[v][0], P[v][1], P[v][2], P[v][3]};
                        }
#pragma unroll
                        for (int half = 0; half < 2; ++half) { // distance pieces 0..3, then 4..7 (two doubles each)
#pragma unroll
                            for (int pp = 0; pp < 4; ++pp) {
                                P[pp][0] = __double2loint(dv[8 * half + 2 * pp]); P[pp][1] = __double2hiint(dv[8 * half + 2 * pp]);
                                P[pp][2] = __double2loint(dv[8 * half + 2 * pp + 1]); P[pp][3] = __double2hiint(dv[8 * half + 2 * pp + 1]);
                            }
                            quad_transpose(P, odd1, odd2);
                            const int piece = 4 * half + u4;
                            if (2 * piece < M) {
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                    if (rq[v] >= 0)
                                        *reinterpret_cast<Int4 *>(dist + (uint64_t)(unsigned)rq[v] * (unsigned)M + 2 * piece) = Int4{P[v][0], P[v][1], P[v][2], P[v][3]};
                            }
                        }
                    } else if (rid >= 0) { // reference semantics (the caller's pads stay) or a row width that is no multiple of four
                        const uint64_t row = (uint64_t)(unsigned)id * (unsigned)M;
#pragma unroll
                        for (int sx = 0; sx < 16; ++sx)
                            if (sx < M && (sx < kept || write_pads)) {
                                verlet[row + sx] = idv[sx];
                                dist[row + sx] = dv[sx];
                            }
                    }
                }
                STAMP(6);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                STAMP(7);
            } else if (!COUNT) {
                // ---- rows of up to 64 slots (and any row in cells of many atoms): the same lane-per-centre form, four slots at
                // a time — tickets read back as one word pair, run-table reads, twelve position reads, four distance chains —
                // each group's part of the row stored at once (16-byte stores where the row allows).
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                STAMP(5);
                int maxk = kept;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) maxk = max(maxk, __shfl_xor(maxk, d, 64));
                const uint64_t row = (uint64_t)(unsigned)id * (unsigned)M;
                const int gend = write_pads ? M : min(M, maxk);
                const bool ids_only = (write_pads & 2) != 0; // (uniform) lane_ids_only: no distance is computed or stored
                for (int g0 = 0; g0 < gend; g0 += 4) {
                    int idv[4];
                    double dv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        idv[u] = -1; // pads neighbor.py:125-129
                        dv[u] = pad;
                    }
                    if (g0 < maxk) { // (uniform)
                        const uint2 t4 = *reinterpret_cast<const uint2 *>(my + g0); // tickets g0 .. g0+3
                        int k[4];
                        bool h[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            h[u] = g0 + u < kept;
                            const unsigned t = ((u < 2 ? t4.x : t4.y) >> (16 * (u & 1))) & 0xffffu;
                            const int r = h[u] ? (int)(t >> JB) : 4; // a lane whose row is shorter: the first atom of its own run, result unused
                            k[u] = (int)(hr[run_cell(cb, r)] & 0xffffu) + (h[u] ? (int)(t & ((1u << JB) - 1u)) : 0);
                        }
                        double2 cj[4];
                        double zj[4], d2[4];
                        int nid[4], sh[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            cj[u] = lxy[k[u]];
                            zj[u] = lz[k[u]];
#ifdef MDH_EXP_GATHER
                            nid[u] = mine ? pk[__float_as_int(f4[k[u]].w)].id : 0;
#else
                            nid[u] = __float_as_int(f4[k[u]].w);
#endif
                            sh[u] = (!TRI && general_tile) ? (int)lsh[k[u]] : 0;
                        }
                        bool slow = false;
                        double rr[4] = {pad, pad, pad, pad}; // neighbor.cpp:174
                        if (!ids_only) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (TRI) d2[u] = exact_d2<2>(b, cj[u].x, cj[u].y, zj[u], xi, yi, zi, 0);
                            else if (general_tile) d2[u] = exact_d2<1>(b, cj[u].x, cj[u].y, zj[u], xi, yi, zi, sh[u]);
                            else d2[u] = exact_d2<0>(b, cj[u].x, cj[u].y, zj[u], xi, yi, zi, 0);
                            d2[u] = h[u] ? d2[u] : 1.0;
                            slow = slow || !sqrt_fast_ok(d2[u]);
                        }
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) == 0, 1)) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) rr[u] = sqrt_fast(d2[u]);
                        } else {
#pragma unroll
                            for (int u = 0; u < 4; ++u) rr[u] = sqrt(d2[u]);
                        }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            idv[u] = h[u] ? nid[u] : -1;
                            dv[u] = h[u] ? rr[u] : pad;
                        }
                    }
                    if (mine) {
                        if (write_pads && g0 + 4 <= M) { // (any alignment: the hardware takes 16-byte stores at 4-byte addresses)
                            *reinterpret_cast<Int4 *>(verlet + row + g0) = Int4{idv[0], idv[1], idv[2], idv[3]};
                            if (!ids_only) {
                                Int4 *dp = reinterpret_cast<Int4 *>(dist + row + g0);
                                dp[0] = Int4{__double2loint(dv[0]), __double2hiint(dv[0]), __double2loint(dv[1]), __double2hiint(dv[1])};
                                dp[1] = Int4{__double2loint(dv[2]), __double2hiint(dv[2]), __double2loint(dv[3]), __double2hiint(dv[3])};
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (g0 + u < M && (g0 + u < kept || write_pads)) {
                                    verlet[row + g0 + u] = idv[u];
                                    if (!ids_only) dist[row + g0 + u] = dv[u];
                                }
                        }
                    }
                }
                STAMP(6);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                STAMP(7);
            }
        }
        if (!LOOP)
            break;
        if (jt + (int)(gridDim.x >> 3) < per) lds_barrier(); // LDS is reused by the next tile
    } // tiles of this workgroup
    if (COUNT) {
        int m = vmax;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
        // one word takes ~90 atomics per microsecond: 200 000 waves must not all queue on it.  Almost every wave finds the
        // maximum already there (a plain device-scope read), the few that raise it use the atomic.
        if (lane == 0 && m > __hip_atomic_load(max_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_count, m);
    }
}

// tiles with at least one centre atom: flag (one thread per tile), then an order-preserving compaction
__global__ __launch_bounds__(256) void k_tile_live(const int *__restrict__ cell_start, Grid g, int nt0, int nt1, int nt2, Shape ts,
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

// ---------------------------------------------------------------------------------------------------------------
// Grid statistics that size the launch: how many cells belong to the occupied region (counted in 4x4x4-cell blocks: the
// empty cells a lattice leaves between occupied ones belong to it, vacuum does not) and the histogram of the lengths of the
// 3-cell z-runs — a run's hit mask is one 32-bit register.  Counted on the device; the host uses the values the previous call
// with the same (N, grid) left in pinned memory — an MD-style sequence of calls never waits — and waits only the first
// time it sees a new (N, grid).  A stale value costs speed, never correctness.
// out[0] = occupied cells; out[1 + len] = number of runs of that length (len 0..96; out[98] = longer)
__global__ __launch_bounds__(256) void k_grid_stats(const int *__restrict__ cell_start, Grid g, int *__restrict__ out)
{
    __shared__ int hist[GridStats::NBIN];
    for (int k = threadIdx.x; k < GridStats::NBIN; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    const int nb0 = (g.nc[0] + 3) >> 2, nb1 = (g.nc[1] + 3) >> 2, nb2 = (g.nc[2] + 3) >> 2;
    const int64_t nblk = (int64_t)nb0 * nb1 * nb2;
    int mine = 0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nblk; q += (int64_t)gridDim.x * blockDim.x) {
        const int b2 = (int)(q % nb2), b1 = (int)((q / nb2) % nb1), b0 = (int)(q / ((int64_t)nb2 * nb1));
        const int x1 = min(b0 * 4 + 4, g.nc[0]), y1 = min(b1 * 4 + 4, g.nc[1]), z0 = b2 * 4, z1 = min(z0 + 4, g.nc[2]);
        bool any = false;
        for (int a = b0 * 4; a < x1; ++a)
            for (int c = b1 * 4; c < y1; ++c) {
                const int64_t col = ((int64_t)a * g.nc[1] + c) * g.nc[2];
                any = any || cell_start[col + z1] > cell_start[col + z0];
            }
        if (any) {
            mine += (x1 - b0 * 4) * (y1 - b1 * 4) * (z1 - z0);
            for (int a = b0 * 4; a < x1; ++a)
                for (int c = b1 * 4; c < y1; ++c) {
                    const int64_t col = ((int64_t)a * g.nc[1] + c) * g.nc[2];
                    for (int k = z0; k < z1; ++k) {
                        const int len = cell_start[col + min(k + 2, g.nc[2])] - cell_start[col + max(k - 1, 0)];
                        atomicAdd(&hist[1 + min(len, 97)], 1);
                    }
                }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&hist[0], mine);
    __syncthreads();
    for (int k = threadIdx.x; k < GridStats::NBIN; k += blockDim.x)
        if (hist[k]) atomicAdd(&out[k], hist[k]);
}

namespace {
struct StatEntry { int64_t N, ncell; int device; int *host; unsigned calls; };
std::mutex g_stat_mu;
std::vector<StatEntry> g_stat;
} // namespace

} // namespace lane

int grid_stats_hint(Scope &sc, const CellGrid &cg, int64_t N, GridStats *out)
{
    using namespace lane;
    hipStream_t st = sc.stream();
    int device = 0;
    (void)hipGetDevice(&device);
    std::lock_guard<std::mutex> lk(g_stat_mu);
    auto count = [&](int *host_dst) -> int {
        int *dcnt = sc.alloc_n<int>(GridStats::NBIN);
        if (sc.failed())
            return sc.error();
        MDH_HIP(hipMemsetAsync(dcnt, 0, sizeof(int) * GridStats::NBIN, st));
        const int blocks = (int)std::min<int64_t>((cg.g.ncell / 64 + 255) / 256 + 1, 2048);
        hipLaunchKernelGGL(k_grid_stats, dim3(blocks), dim3(256), 0, st, cg.cell_start, cg.g, dcnt);
        MDH_HIP(hipMemcpyAsync(host_dst, dcnt, sizeof(int) * GridStats::NBIN, hipMemcpyDeviceToHost, st));
        return MDH_OK;
    };
    auto read = [&](const int *host) {
        // the copy may be landing right now: a torn read mixes two generations of a slowly drifting statistic
        for (int k = 0; k < GridStats::NBIN; ++k) out->v[k] = ((const volatile int *)host)[k];
        if (out->v[0] <= 0) out->v[0] = (int)std::min<int64_t>(cg.g.ncell, 2147483647);
        out->last_listed = ((const volatile int *)host)[GridStats::NBIN];
        out->listed_sink = const_cast<int *>(host) + GridStats::NBIN;
    };
    for (auto &e : g_stat)
        if (e.N == N && e.ncell == cg.g.ncell && e.device == device) {
            read(e.host);
            if ((++e.calls & 7u) == 0) // the occupied region of a running simulation drifts slowly: recount every 8th call
                MDH_TRY(count(e.host));
            return MDH_OK;
        }
    int *host = nullptr;
    if (g_stat.size() >= 64) { // keep the table small: the oldest signature hands its pinned block on (never freed: a copy
        host = g_stat.front().host; // enqueued on some other stream may still land in it — a wrong hint at worst)
        g_stat.erase(g_stat.begin());
    } else {
        MDH_HIP(hipHostMalloc(reinterpret_cast<void **>(&host), sizeof(int) * (GridStats::NBIN + 1), hipHostMallocDefault));
    }
    host[GridStats::NBIN] = -1; // tiles listed for the second pass: written by the device (k_neighbor_lane, second pass)
    MDH_TRY(count(host));
    MDH_HIP(hipStreamSynchronize(st));
    read(host);
    g_stat.push_back(StatEntry{N, cg.g.ncell, device, host, 0u});
    return MDH_OK;
}

namespace lane {

// tk8: one-byte tickets, else two-byte ones; rows of (M + 1) tickets rounded up to a multiple of four; rw rows per wave
static size_t lds_bytes(int cap, int64_t M, bool tk8, int rw, int nw = 4, bool fcna = false)
{
    const size_t wave = std::max<size_t>((size_t)rw * (size_t)ticket_row((int)M, false) * (tk8 ? 1 : 2), 256); // (the kernel's wstride)
    const size_t tk = (size_t)nw * wave;
    return (size_t)cap * 16 + (size_t)cap * 16 + (size_t)cap * 8 + (size_t)cen_cap(nw) * 4 + (size_t)(cap + (cap & 1)) * 2 + ((tk + 15) & ~(size_t)15);
}

} // namespace lane

// reason (negative) the tile kernel cannot take a call with this box, grid and row width; 0: it can
static int lane_refusal(const DBox &b, const Grid &g, int64_t M)
{
    if (g.mode != 0 || M <= 0 || M > 128) return -1;
    for (int d = 0; d < 3; ++d)
        if (g.nc[d] < (b.pbc[d] ? 7 : 4)) return -3; // image numbers from the cell pair need >= 7 cells; skipping the far side of an open axis >= 4
    return 0;
}

static LanePlan plan_lane_fresh(const DBox &b, const Grid &g, int64_t N, int64_t M, const GridStats &gs, double rc, bool fcna, bool count);

// The plan is a function of the box, the grid, N, M, rc and the run-length / occupancy statistics — in a trajectory the same
// from call to call — and the search below walks ~800 tile shapes: a few microseconds of a step that is 70 us of host time at
// 4 000 atoms.  The last plan of the thread is kept with its inputs and handed out again when they are the same.
LanePlan plan_lane(const DBox &b, const Grid &g, int64_t N, int64_t M, const GridStats &gs, double rc, bool fcna, bool count)
{
    struct Key { int nc[3], pbc[3], tri; int64_t N, M; double rc, o[3], h[9], thick[3]; int fcna, count; int v[GridStats::NBIN]; };
    struct Memo { bool set = false; Key key; LanePlan plan; int hook[8]; };
    static thread_local Memo memo;
    Key k;
    std::memset(&k, 0, sizeof(k)); // (padding bytes too: the keys are compared as bytes)
    for (int d = 0; d < 3; ++d) { k.nc[d] = g.nc[d]; k.pbc[d] = b.pbc[d]; k.o[d] = b.o[d]; k.thick[d] = b.thick[d]; }
    for (int d = 0; d < 9; ++d) k.h[d] = b.h[d];
    k.tri = b.tri; k.N = N; k.M = M; k.rc = rc; k.fcna = fcna; k.count = count;
    for (int q = 0; q < GridStats::NBIN; ++q) k.v[q] = gs.v[q];
    if (memo.set && std::memcmp(&k, &memo.key, sizeof(k)) == 0) {
        LanePlan p = memo.plan;
        p.last_listed = gs.last_listed; // (the two fields that are not the plan's own)
        p.listed_sink = gs.listed_sink;
        lane::g_last_listed = gs.listed_sink ? *(volatile int *)gs.listed_sink : gs.last_listed;
        for (int q = 0; q < 8; ++q) lane::g_last_plan[q] = memo.hook[q];
        return p;
    }
    const LanePlan p = plan_lane_fresh(b, g, N, M, gs, rc, fcna, count);
    memo.key = k;
    memo.plan = p;
    for (int q = 0; q < 8; ++q) memo.hook[q] = lane::g_last_plan[q];
    memo.set = true;
    return p;
}

static LanePlan plan_lane_fresh(const DBox &b, const Grid &g, int64_t N, int64_t M, const GridStats &gs, double rc, bool fcna, bool count)
{
    using namespace lane;
    LanePlan p{};
    for (int k = 0; k < 8; ++k) g_last_plan[k] = 0;
    if (N <= 0) { g_last_plan[6] = -1; return p; }
    if (const int why = lane_refusal(b, g, M)) { g_last_plan[6] = why; return p; }
    if (!(rc > 1e-12 && rc < 1e12)) { g_last_plan[6] = -4; return p; }
    int64_t runs = 0, over32 = 0, over96 = 0;
    for (int k = 1; k < GridStats::NBIN; ++k) runs += gs.v[k];
    for (int len = 29; len <= 97; ++len) over32 += gs.v[1 + len]; // (a little below the limits: the statistics are the previous call's)
    for (int len = 89; len <= 97; ++len) over96 += gs.v[1 + len];
    const bool long_runs = runs > 0 && (double)over32 > 0.002 * (double)runs; // many runs would not fit one 32-bit hit mask: the wide instance (three)
    // cells so full that more than a few per cent of the runs would not fit three masks (fewer — the wider last cell of a small box — go to the mop-up kernel)
    if (runs > 0 && (double)over96 > 0.05 * (double)runs) { g_last_plan[6] = -5; g_last_plan[5] = (int)over96; g_last_plan[4] = (int)runs; return p; }
    const int64_t occ = gs.v[0] > 0 ? gs.v[0] : g.ncell;
    const double pop = (double)N / (double)occ; // mean atoms per cell of the occupied region
    // rows of 65 ... 128 slots: measured against the round-1 tiled kernel on 4 M atoms of rattled fcc Cu (tools/rc_sweep.py,
    // profiles/r04_rc_sweep.txt) — ahead while a cell holds fewer than ~20 atoms (rc 5.6 / 5.8 / 6.0 A: 7.6 -> 5.5, ~8.3 -> 5.9,
    // 8.8 -> 5.7 ms), behind beyond (rc 6.5 A, 24 atoms per cell: a third of the tiles hold a run of more than 96 candidates and
    // go to the mop-up code, 10.9 -> 12.7 ms)
    if (M > 64 && pop > 19.5) { g_last_plan[6] = -7; g_last_plan[5] = (int)(1000.0 * pop); return p; }
    static const int cap_env = [] { const char *e = std::getenv("MDH_LANE_CAP"); return e ? std::atoi(e) : 0; }();
    static const int wgs_env = [] { const char *e = std::getenv("MDH_LANE_WGS"); return e ? std::atoi(e) : 0; }(); // A/B: workgroups per CU the LDS is cut for
    // rows of at most 16 slots in cells of a few atoms: one-byte tickets, the lean LDS layout, rows written by the centre's lane;
    // four workgroups per CU where the instance keeps to 128 VGPRs (not the fused CNA)
    static const int tk8_env = [] { const char *e = std::getenv("MDH_LANE_TK8"); return e ? std::atoi(e) : 1; }(); // A/B: 0 = the slot-per-lane write-out always
    const bool tk8 = (count && !long_runs) || (tk8_env && M <= 16 && !long_runs);
    const int max_wgs = (tk8 && (!fcna || MDH_FCNA_LEAN)) ? 4 : 3;
    // LDS budget: four workgroups per CU (the 128-VGPR instance only), else three, two, one, if the tile that allows is not
    // much worse than what fewer would get.  Rows of many slots in cells of many atoms (rc = 5 A, 50 slots: the reference's
    // own benchmark call) leave few centres per tile: the ticket rows are sized for them (rw rows per wave), not for 64.
    // eight waves per workgroup (tiles of up to 512 halo cells, two workgroups per CU): the one-byte instance without the fused
    // CNA, where four workgroups of four waves would share the CU anyway
#ifdef MDH_LANE_NW8
    static const int nw_env = [] { const char *e = std::getenv("MDH_LANE_NW"); return e ? std::atoi(e) : 0; }(); // A/B: 4 or 8
#else
    constexpr int nw_env = 0;
#endif
    // A/B (tools/measure_r05.sh lane_tiles): MDH_LANE_TILE="txy,tz" forces the tile shape where it fits
    static const int tile_env = [] { const char *e = std::getenv("MDH_LANE_TILE"); int a = 0, c = 0; return (e && std::sscanf(e, "%d,%d", &a, &c) == 2) ? a * 100 + c : 0; }();
    Shape best{0, 0};
    int best_cap = 0, best_wgs = 0, best_rw = 64, best_nw = 4;
    double best_score = -1.0;
    bool stop = false;
    for (int nw = 8; nw >= 4 && !stop; nw -= 4)
    for (int wgs = (nw == 8 ? 2 : max_wgs); wgs >= 1; --wgs) {
        if (nw == 8 && (!(tk8 && !fcna) || nw_env != 8 || wgs < 2)) // measured 3 % slower than four waves: on request only
            continue;
        if (nw == 4 && nw_env == 8 && tk8 && !fcna)
            continue;
        if (wgs_env > 0 && nw == 4 && wgs != std::min(wgs_env, max_wgs))
            continue;
        const int nthr = nw * 64;
        // static tables (1.1 KB: the run table, scan scratch, flags) and a margin.  LDS is handed out in 512-byte granules: 40 960 B
        // in all give four workgroups per CU (measured: 40 544 do, 41 216 do not; 52.9 KB three, 54.3 KB not)
        const long budget = nw == 8 ? 160 * 1024 / 2 - 2112 - 64 : 160 * 1024 / wgs - 1088 - (wgs == 4 ? 64 : 1600);
        for (int txy = 1; txy <= 8; ++txy)
            for (int tz = 1; tz <= 24; ++tz) {
                const int nh = (txy + 2) * (txy + 2) * (tz + 2);
                if (nh > nthr)
                    continue;
                if (tile_env && (txy != tile_env / 100 || tz != tile_env % 100))
                    continue;
                const int ncc = txy * txy * tz;
                const double c = ncc * pop;                                 // centre atoms per tile
                if (c * 1.15 > cen_cap(nw))
                    continue;
                int rw = 64;
                if (!tk8) rw = std::min(64, std::max(8, ((int)std::ceil(c * 1.15 / nw) + 3) & ~3)); // (a row is a multiple of eight bytes: any count keeps the waves' blocks aligned)
                const long fixed = (long)lds_bytes(0, M, tk8, rw, nw, fcna);
                int cap = (int)((budget - fixed - 2) / 42);
                if (cap_env > 0) cap = cap_env;
                cap = std::min(cap, 2040); // (a centre's LDS index takes 11 bits of its table entry)
                if (cap < 64 || nh * pop > 0.875 * cap) // head-room for density fluctuations; what overflows goes to the slice pass
                    continue;
                const double passes = std::ceil(c * 1.15 / (nw * rw)); // (head-room: a second pass for a handful of centres is a waste)
                const double util = c / (passes * nthr);                   // lane utilisation of the scan
                const double reuse = (double)ncc / (double)nh;             // centre cells per staged cell
                const int wpc = wgs * nw;                                  // waves per CU
                const double score = util * (0.35 + reuse) * (wpc == 16 ? 1.1 : (wpc == 12 ? 1.0 : (wpc == 8 ? 0.85 : (M > 64 ? 0.45 : 0.6))));
                // the big tile must leave the chip full: at least four workgroups' worth of tiles per CU
                if (nw == 8 && (double)occ / ncc < 4.0 * 256.0)
                    continue;
                if (score > best_score) { best_score = score; best = Shape{txy, tz}; best_cap = cap; best_wgs = wgs; best_rw = rw; best_nw = nw; }
            }
        if (cap_env > 0) {
            stop = true;
            break;
        }
    }
    if (!best.txy) { g_last_plan[6] = -6; return p; }
    // Decision band of the single-precision scan (file header).  E bounds the staged coordinates (far-atom check of the
    // kernel), du the error of one staged coordinate: rounding to f32 plus what the double-precision shift can lose;
    // |d2_f32 - d2| <= (2 du)(2 sqrt(3) |d| + 6 du) + 5 * 2^-24 max(d2, rc^2) for the three subtractions and the FMA chain.
    const int hmax = std::max(best.txy, best.tz) + 2;
    double E = ((double)hmax + 1.5) * rc;
    double big = E;
    if (b.tri) { // the halo is a parallelepiped: its edges along the cell vectors, coordinates bounded by their sum
        double ext = 0, span = 0;
        for (int d = 0; d < 3; ++d) {
            const double len = std::sqrt(b.h[3 * d] * b.h[3 * d] + b.h[3 * d + 1] * b.h[3 * d + 1] + b.h[3 * d + 2] * b.h[3 * d + 2]);
            ext += ((double)((d == 2 ? best.tz : best.txy) + 2) + 1.5) * len * rc / b.thick[d]; // a cell is rc / thickness of the vector
            span += len;
        }
        E = ext;
        big = E + 3.0 * span;
        for (int d = 0; d < 3; ++d) big += std::fabs(b.o[d]);
        big *= 4.0; // the wrap and the frame shift go through the fractional coordinates: a few more roundings
    } else {
        // (a raw coordinate may be img::MAX_M + 1 box lengths away from the box: the shift L * n loses at most an ulp of THAT)
        for (int d = 0; d < 3; ++d) big = std::max(big, std::fabs(b.o[d]) + (double)(img::MAX_M + 2) * std::fabs(b.h[d * 4]) + E);
    }
    const double du = std::ldexp(E, -24) * 1.01 + std::ldexp(big, -49);
    const double rcsq = rc * rc;
    const double tol = 2.0 * (11.0 * du * rc + 12.0 * std::ldexp(rcsq, -24)); // twice the bound: a wider band costs nothing
    // e = d2 - c in single precision is within tol of the exact value.  c: the largest float <= rc^2 - tol, so that e < 0 is
    // a hit for sure; W: the smallest float >= (rc^2 - c) + tol, so that e > W is a miss for sure
    float c = (float)(rcsq - tol);
    while ((double)c > rcsq - tol) c = std::nextafterf(c, -INFINITY);
    const double want = (rcsq - (double)c) + tol;
    float W = (float)want;
    while ((double)W < want) W = std::nextafterf(W, INFINITY);
    p.mid = c;
    p.T = W;
    p.txy = best.txy;
    p.tz = best.tz;
    p.cap = best_cap;
    p.tk8 = tk8;
    p.wgs = best_wgs;
    p.rw = best_rw;
    p.nw = best_nw;
    p.occupied = occ;
    p.last_listed = gs.last_listed;
    p.listed_sink = gs.listed_sink;
    g_last_listed = gs.listed_sink ? *(volatile int *)gs.listed_sink : gs.last_listed; // (the freshest value the device has delivered)
    // "full": (nearly) every 4x4x4 block of cells holds atoms — all tiles are launched, no list of live ones is made.  A couple of
    // per cent of empty blocks (the corner cell of a 100^3-cell fcc box holds no lattice site) cost a workgroup each that finds no
    // centre and leaves; the list costs three launches
    p.full = (double)occ >= 0.98 * (double)g.ncell;
    g_last_plan[0] = p.txy; g_last_plan[1] = p.tz; g_last_plan[2] = p.cap; g_last_plan[3] = (int)lds_bytes(p.cap, M, tk8, p.rw, p.nw, fcna);
    g_last_plan[4] = p.full | (p.tk8 ? 2 : 0) | (p.wgs << 2) | (p.nw == 8 ? 32 : 0); g_last_plan[5] = (int)(1000.0 * pop); g_last_plan[6] = (int)std::min<int64_t>(occ, 2147483647); g_last_plan[7] = 1;
    return p;
}

// count == true: nn and *max_count only (first pass of the exact-width variant); M is then 1
// The next build's wide instance writes ids and counts only (knn.hip: the rows of a cutoff build as candidates of a k-nearest search;
// the distances would be 8 M bytes per atom written and ~300 instructions per four slots computed for nobody).  The other kernels of
// a build (rows of <= 16 slots, the mop-up code) write distances as always: the caller still passes a buffer.
static thread_local bool g_lane_ids_only = false;
void lane_ids_only(bool on) { g_lane_ids_only = on; }

int launch_neighbor_lane(Scope &sc, const CellGrid &cg, const LanePlan &plan, int64_t N, const DBox &b, double rc,
                         int *verlet, double *dist, int *nn, int64_t M, bool fill_pads, bool count, int *max_count,
                         TileFilter &tf, int *pattern)
{
    using namespace lane;
    int nt[3];
    for (int d = 0; d < 3; ++d) {
        const int T = d == 2 ? plan.tz : plan.txy;
        nt[d] = (cg.g.nc[d] + T - 1) / T;
    }
    const int64_t ntiles = (int64_t)nt[0] * nt[1] * nt[2];
    const Shape ts = make_shape(plan.txy, plan.tz, nt[1], nt[2]);
    const int nsub = ts.tz; // second pass: one-cell slices along z
    unsigned *live = sc.alloc_n<unsigned>((size_t)ntiles);
    int *slot = sc.alloc_n<int>((size_t)ntiles + 1);
    int *tile_list = sc.alloc_n<int>((size_t)ntiles);
    int *flagged = sc.alloc_n<int>((size_t)ntiles);                   // tiles of the first pass for the second (each listed at most once)
    int *flagged2 = sc.alloc_n<int>((size_t)ntiles * (size_t)nsub);   // slices of the second pass for the thread-per-atom code
    if (sc.failed())
        return sc.error();
    hipStream_t st = sc.stream();
    int per = (int)((ntiles + 7) / 8);
    int list_mode = 0, tile_base = 0, nt0_run = nt[0];
    if (plan.full) { // the statistics say that no 4x4x4 block of cells is empty: all tiles are live, workgroup b owns tile b
        tile_list = nullptr;
    } else if (cg.win_hi > cg.win_lo) {
        // the caller promised a window of planes along axis 0 (a slab of a decomposed system, mdh_hint_cell_window) and the cell
        // grid was built over it: the tiles that can hold atoms are ONE range of tile numbers (axis 0 is the slowest index) —
        // no list, no pass over the tiles of the global grid to make one
        tile_list = nullptr;
        // (tiles of the planes that hold atoms wanting rows: the window, or the centre window inside it)
        const int w_lo = cg.cen_hi > cg.cen_lo ? std::max(cg.win_lo, cg.cen_lo) : cg.win_lo, w_hi = cg.cen_hi > cg.cen_lo ? std::min(cg.win_hi, cg.cen_hi) : cg.win_hi;
        const int t_lo = w_lo / ts.txy, t_hi = (std::max(w_hi, w_lo + 1) - 1) / ts.txy;
        tile_base = t_lo * nt[1] * nt[2];
        nt0_run = t_hi - t_lo + 1;
        per = (int)(((int64_t)nt0_run * nt[1] * nt[2] + 7) / 8);
    } else {
        hipLaunchKernelGGL(k_tile_live, dim3(grid_for(ntiles, 256)), dim3(256), 0, st, cg.cell_start, cg.g, nt[0], nt[1], nt[2], ts, live);
        MDH_TRY(exclusive_scan_u32(sc, live, slot, ntiles)); // slot[ntiles] = number of live tiles
        hipLaunchKernelGGL(k_tile_compact, dim3(grid_for(ntiles, 256)), dim3(256), 0, st, live, slot, (int)ntiles, tile_list);
        // live tiles expected from the last known occupancy (+25 %); a workgroup takes further tiles of its chunk if that was too few
        const int64_t est_live = std::max<int64_t>(1, plan.occupied / std::max(1, ts.txy * ts.txy * ts.tz) * 2);
        per = std::max(1, std::min(per, (int)((est_live + est_live / 4 + 7) / 8)));
        list_mode = 1;
    }
    const dim3 grid((unsigned)(per * 8));
    const size_t lds1 = lds_bytes(plan.cap, count ? 1 : M, plan.tk8, plan.rw, plan.nw, pattern != nullptr); // first pass (four or eight waves per workgroup)
    const size_t lds2 = lds_bytes(plan.cap, count ? 1 : M, plan.tk8, plan.rw, 4, pattern != nullptr);       // slice pass: always four
    const int cen_lo = cg.cen_hi > cg.cen_lo ? cg.cen_lo : 0, cen_hi = cg.cen_hi > cg.cen_lo ? cg.cen_hi : 0x7fffffff;
    const int Mi = (int)M, wp = fill_pads ? (g_lane_ids_only ? 3 : 1) : 0; // bit 1: the wide instance neither computes nor stores the distances (lane_ids_only)
    const float negc = -plan.mid;
    const int nt2b = nt[2] * nsub;
    // second pass: workgroups walk the listed tiles' slices.  Nothing was listed by the previous build with this (N, grid)
    // (a lattice, nearly always): a small stand-by grid — it walks whatever turns up this time, slowly but correctly; 22 us of
    // every build went into 1024 workgroups that found an empty list
    const dim3 grid2(plan.last_listed == 0 ? 64u : 1024u);
    const Shape ts2 = make_shape(ts.txy, 1, nt[1], nt2b);
    const bool indirect = !cg.pk; // CellGrid::ix
    const lane::IndirectSrc isrc{cg.ix, cg.iy, cg.iz, cg.imv, cg.order};
#define MDH_LANE_PASS_I(COUNT, TRI, LOOP, FCNA, TK8, NW, IND, GRID, JT0, ...)                                                             \
    do {                                                                                                                                  \
        if (lds > 60 * 1024) /* above the default dynamic-LDS limit: raise it for the instance about to run */                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_neighbor_lane<COUNT, TRI, LOOP, FCNA, TK8, NW, IND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_neighbor_lane<COUNT, TRI, LOOP, FCNA, TK8, NW, IND>), GRID, dim3((NW) * 64), lds, st, cg.pk, cg.cell_start, b, \
                           cg.g, rc, negc, plan.T, verlet, dist, nn, Mi, wp, plan.cap, cg.flags, nullptr, __VA_ARGS__, pattern, tf.cna_todo, JT0, plan.rw, tile_base, plan.listed_sink, \
                           cen_lo, cen_hi, isrc); \
    } while (0)
#define MDH_LANE_PASS(COUNT, TRI, LOOP, FCNA, TK8, NW, GRID, JT0, ...)                                                                         \
    do {                                                                                                                                  \
        const size_t lds = (NW) == 8 ? lds1 : lds2;                                                                                       \
        if (indirect) MDH_LANE_PASS_I(COUNT, TRI, LOOP, FCNA, TK8, NW, true, GRID, JT0, __VA_ARGS__);                                     \
        else MDH_LANE_PASS_I(COUNT, TRI, LOOP, FCNA, TK8, NW, false, GRID, JT0, __VA_ARGS__);                                             \
    } while (0)
    // first pass: one tile per workgroup — all tiles, or the list of live ones, whose length only the device knows: the grid
    // is cut for the expected number and a walked launch stands by for what a longer list leaves over (it leaves at once
    // otherwise); second pass: one-cell slices of what the first listed
#define MDH_LANE_LAUNCH_NW(COUNT, TRI, FCNA, TK8, NW)                                                                                     \
    do {                                                                                                                                  \
        if (list_mode) {                                                                                                                  \
            MDH_LANE_PASS(COUNT, TRI, false, FCNA, TK8, NW, grid, 0, nt[0], nt[1], nt[2], ts, tile_list, slot + ntiles, 1, max_count, flagged, nullptr, 0, 1, 2); \
            if ((int64_t)per * 8 < ntiles)                                                                                                \
                MDH_LANE_PASS(COUNT, TRI, true, FCNA, TK8, NW, dim3(512), per, nt[0], nt[1], nt[2], ts, tile_list, slot + ntiles, 1, max_count, flagged, nullptr, 0, 1, 2); \
        } else {                                                                                                                          \
            MDH_LANE_PASS(COUNT, TRI, false, FCNA, TK8, NW, grid, 0, nt0_run, nt[1], nt[2], ts, nullptr, slot + ntiles, 0, max_count, flagged, nullptr, 0, 1, 2); \
        }                                                                                                                                 \
        MDH_LANE_PASS(COUNT, TRI, true, FCNA, TK8, 4, grid2, 0, nt[0], nt[1], nt2b, ts2, nullptr, cg.flags + 2, 1, max_count, flagged2, flagged, nt[2], nsub, 3); \
    } while (0)
#define MDH_LANE_LAUNCH(COUNT, TRI, FCNA, TK8) MDH_LANE_LAUNCH_NW(COUNT, TRI, FCNA, TK8, 4)
#ifdef MDH_LANE_NW8 // measuring build (make nw8): the eight-wave instances are compiled in and MDH_LANE_NW=8 selects them
    if (plan.nw == 8) { // (one-byte instance without the fused CNA: plan_lane)
        if (count) { if (b.tri) MDH_LANE_LAUNCH_NW(true, true, false, true, 8); else MDH_LANE_LAUNCH_NW(true, false, false, true, 8); }
        else { if (b.tri) MDH_LANE_LAUNCH_NW(false, true, false, true, 8); else MDH_LANE_LAUNCH_NW(false, false, false, true, 8); }
    } else
#endif
    if (count) {
        if (plan.tk8) { if (b.tri) MDH_LANE_LAUNCH(true, true, false, true); else MDH_LANE_LAUNCH(true, false, false, true); }
        else { if (b.tri) MDH_LANE_LAUNCH(true, true, false, false); else MDH_LANE_LAUNCH(true, false, false, false); }
    } else if (plan.tk8) {
        if (pattern) { if (b.tri) MDH_LANE_LAUNCH(false, true, true, true); else MDH_LANE_LAUNCH(false, false, true, true); }
        else { if (b.tri) MDH_LANE_LAUNCH(false, true, false, true); else MDH_LANE_LAUNCH(false, false, false, true); }
    } else {
        if (pattern) { if (b.tri) MDH_LANE_LAUNCH(false, true, true, false); else MDH_LANE_LAUNCH(false, false, true, false); }
        else { if (b.tri) MDH_LANE_LAUNCH(false, true, false, false); else MDH_LANE_LAUNCH(false, false, false, false); }
    }
#undef MDH_LANE_PASS
#undef MDH_LANE_PASS_I
#undef MDH_LANE_LAUNCH
#undef MDH_LANE_LAUNCH_NW
    MDH_HIP(hipGetLastError());
    // what the two passes listed for the thread-per-atom code (k_neighbor_tiles), in the tiling of the second pass
    tf.flag = reinterpret_cast<const unsigned char *>(flagged2); // (non-null: "a tiled kernel ran"; the per-tile byte flags are not used with a list)
    tf.any = cg.flags + 3;
    tf.moved = cg.flags;
    tf.list = flagged2;
    tf.list_cap = (int)std::min<int64_t>(ntiles * nsub, 2147483647);
    tf.tile = ts2.txy;
    tf.tile_z = ts2.tz;
    tf.nt[0] = nt[0]; tf.nt[1] = nt[1]; tf.nt[2] = nt2b;
    return MDH_OK;
}

} // namespace mdh

#ifdef MDH_STAMPS
extern "C" int mdh_debug_lane_stamps(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mdh::lane::g_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : 1;
}
#endif
namespace mdh { int lane_last_listed() { return lane::g_last_listed; } }
extern "C" int mdh_debug_neighbor_plan(int *plan8)
{
    for (int k = 0; k < 8; ++k) plan8[k] = mdh::lane::g_last_plan[k];
    mdh::lane::g_last_plan[7] = 0; // [7] = 1: the plan was made since the last query
    return MDH_OK;
}

MDH_WARM_UNIT(neighbor_lane)
