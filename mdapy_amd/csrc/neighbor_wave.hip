// neighbor_wave.hip — cutoff neighbor search, one WAVEFRONT per centre cell, one LANE per candidate (gfx950).
//
// Same result, bit for bit, as the thread-per-atom kernel in neighbor.hip and therefore as src/neighbor.cpp:102-187 of
// the reference (ids, order inside a row, counts, distances).  This is the fast path of mdh_build_neighbor /
// mdh_neighbor_count for orthogonal boxes.
//
// Work decomposition.  A workgroup (4 waves) owns a tile of TXY x TXY x TZ cells and stages the atoms of the halo
// ((TXY+2)^2 x (TZ+2) cells, one cell per thread, coalesced loads from the cell-sorted arrays) into LDS once.  After ONE
// barrier every wave works alone: it takes centre cells of the tile round-robin and, for a centre cell, lays the
// candidates of its 27 cells out ACROSS THE LANES in the reference's visiting order — the 27 cells are 9 z-runs
// (cells (i+da, j+db, k-1..k+1), contiguous in LDS); run r gets S consecutive lanes of group r / (64/S), so a lane
// finds its candidate with two 2-byte LDS reads and an add.  A centre atom is then tested against 64 candidates per
// instruction; the hit mask comes out of the compare as a 64-bit scalar, the slot of a hit inside the row is
// popcount-below-lane (v_mbcnt), i.e. the reference's order (neighbor.cpp:147-177) falls out of the lane order with
// no per-candidate loop control, no per-hit branch and no address arithmetic per candidate.
//
// Arithmetic.  The scan DECIDES in single precision on coordinates relative to the tile corner with the periodic
// image shift already folded in at staging (u = (x - L n) - X0, n = image number of the candidate's cell as seen from
// the tile + the atom's own recorded image code).  |d2_f32 - d2_exact| is bounded by `tol` (host, from the tile extent);
// a pair with d2_f32 < rc^2 - tol is a hit, > rc^2 + tol a miss, and the (rare) pairs in between are decided by the
// reference's own double-precision expression on the spot.  Every distance that is WRITTEN is recomputed in double
// precision from the raw coordinates exactly as the reference does (raw x[j] - wrapped x[i], minimum image,
// (dx*dx + dy*dy) + dz*dz, sqrt), so rows are bit-identical; single precision only prunes.
//
// Output.  Hits are recorded as 2-byte tickets (LDS index of the candidate) in a per-wave buffer of 64 centres; when
// it is full the wave turns tickets into rows cooperatively — MP adjacent lanes write the MP slots of one row, pads
// included — so wave stores cover whole rows.  No workgroup barrier after staging.
//
// What this kernel does not take (left to the thread-per-atom kernel, same results): triclinic boxes, fewer than 7
// cells on a periodic axis / 4 on an open one, unwrapped input (device flag), max_neigh > 64, tiles whose halo
// overflows LDS or holds atoms far outside the box on an open axis (flagged per tile on the device).
#include "common.hpp"
#include "grid.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace mdh {
namespace wave {

static constexpr int NT = 256;      // threads per workgroup (4 waves)
static constexpr int MAX_NH = 256;  // halo cells of a tile: one per thread
static constexpr int NEUTRAL = 1 | (1 << 2) | (1 << 4);   // image code "no shift": (n+1) per axis, 2 bits each
static constexpr int NEUTRAL3 = 2 | (2 << 3) | (2 << 6);  // combined code "no shift": (n+2) per axis, 3 bits each

struct Shape { int txy, tz; };
typedef float float2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;

__device__ int g_dbg[4]; // [0] centre cells on the slow path, [1] tiles that overflowed LDS, [2] tiles with far atoms, [3] pairs decided in double precision
static int g_last_plan[8];

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// v[lane l] = val (both uniform).  The s_nop covers a lane select / value that a VALU instruction (v_readlane) has just written
__device__ __forceinline__ void set_lane(int &v, int val, int l) { int keep; asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(v), "=&s"(keep) : "s"(val), "s"(l)); }

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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

// cell code cc and atom code ca (each (n+1) per axis in 2 bits) -> combined code ((n_cell + m_atom) + 2 per axis in 3 bits)
__device__ __forceinline__ int combine_codes(int cc, int ca)
{
    return ((cc & 3) + (ca & 3)) | ((((cc >> 2) & 3) + ((ca >> 2) & 3)) << 3) | ((((cc >> 4) & 3) + ((ca >> 4) & 3)) << 6);
}

// the reference's squared distance of one pair: raw x[j] - wrapped x[i] (neighbor.cpp:164-166), minimum image
// d - L*floor(d/L+0.5) with the image number n taken from the code (box.h:120-124; L*n exact, d - L*0 == d), then
// (dx*dx + dy*dy) + dz*dz (neighbor.cpp:170)
template <bool GENERAL>
__device__ __forceinline__ double exact_d2(const DBox &b, double xj, double yj, double zj, double xi, double yi, double zi,
                                           int code)
{
    double dx = xj - xi, dy = yj - yi, dz = zj - zi;
    if (GENERAL) {
        dx = dx - b.h[0] * (double)((code & 7) - 2);
        dy = dy - b.h[4] * (double)(((code >> 3) & 7) - 2);
        dz = dz - b.h[8] * (double)(((code >> 6) & 7) - 2);
    }
    return dx * dx + dy * dy + dz * dz;
}

// everything a wave needs to know about its tile
struct Ctx {
    const float4 *f4;            // [cap+64] staged (ux, uy, uz, bits of the atom id)
    const double2 *lxy;          // [cap] staged raw x, y
    const double *lz;            // [cap] staged raw z
    const unsigned short *lsh;   // [cap] combined image code of a staged atom as seen from this tile
    const unsigned *hc;          // [NH] halo cell: LDS offset | population << 16
    const unsigned *hr;          // [NH] 3-cell z-run centred on the cell: LDS offset | length << 16
    double *cxw, *cyw, *czw;     // per wave [64]: wrapped centre of a ticket column
    int *crow, *ccnt;            // per wave [64]: atom id / min(count, M) of a ticket column
    unsigned short *ccol;        // per wave [64]: LDS index of the centre of a ticket column
    unsigned short *tk;          // per wave [64][M] + slack: tickets
    unsigned tk_lds;             // LDS byte address of tk
    float lo, hi;                // decision band around rc^2 in single precision
    double rcsq, pad;
    int HXY, HZ, M, mp_shift;
    bool general, write_pads;
    int *verlet;
    double *dist;
    int *nn;
};

// the reference's own test for one pair, for the pairs the single-precision band cannot decide
__device__ __forceinline__ bool exact_hit(const Ctx &C, const DBox &b, int k, int li)
{
    asm volatile("" : "+v"(k), "+v"(li)); // rare path: keep its address arithmetic out of the loops around it
    const double2 ci = C.lxy[li];
    double xi = ci.x, yi = ci.y, zi = C.lz[li];
    if (b.anypbc) // neighbor.cpp:139-142
        wrap<false>(b, xi, yi, zi);
    const double2 cj = C.lxy[k];
    const double d2 = exact_d2<true>(b, cj.x, cj.y, C.lz[k], xi, yi, zi, C.lsh[k]);
    return d2 <= C.rcsq; // neighbor.cpp:171
}

__device__ __forceinline__ bool eval_hit(const Ctx &C, const DBox &b, float ux, float uy, float uz, float sx, float sy,
                                         float sz, bool ok, int idx, int li)
{
    const float dx = ux - sx, dy = uy - sy, dz = uz - sz;
    const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    bool su = ok && (d2 < C.lo);
    const bool mb = ok && (d2 <= C.hi);
    if (__builtin_expect(__ballot(mb && !su) != 0ull, 0)) {
        if (mb && !su) {
            su = exact_hit(C, b, idx, li);
            atomicAdd(&g_dbg[3], 1);
        }
    }
    return su;
}

// per-wave ticket state
struct Cols {
    int count; // ticket columns in use (uniform)
    int cntv;  // lane c: hits of column c (keeps counting past M, neighbor.cpp:172-177)
    int vmax;  // lane-wise running maximum of the counts (COUNT mode)
};

// tickets -> rows for the columns collected so far
template <bool COUNT>
__device__ __forceinline__ void flush(const Ctx &C, const DBox &b, Cols &K)
{
    const int lane = threadIdx.x & 63;
    const int rows = K.count;
    if (lane < rows) {
        const int li = C.ccol[lane];
        const int id = __float_as_int(C.f4[li].w);
        C.nn[id] = K.cntv;
        if (COUNT) {
            K.vmax = max(K.vmax, K.cntv);
        } else {
            C.crow[lane] = id;
            C.ccnt[lane] = min(K.cntv, C.M);
            const double2 ci = C.lxy[li];
            double xi = ci.x, yi = ci.y, zi = C.lz[li];
            if (b.anypbc) // neighbor.cpp:139-142
                wrap<false>(b, xi, yi, zi);
            C.cxw[lane] = xi; C.cyw[lane] = yi; C.czw[lane] = zi;
        }
    }
    if (!COUNT) {
        wave_sync();
        const int MP = 1 << C.mp_shift; // smallest power of two >= M: slots of a row handled by MP adjacent lanes
        const int e = lane & (MP - 1);
        if (e < C.M) {
            for (int c = lane >> C.mp_shift; c < rows; c += (64 >> C.mp_shift)) {
                const int64_t o = (int64_t)C.crow[c] * C.M + e;
                if (e < C.ccnt[c]) {
                    const int k = C.tk[c * C.M + e];
                    const double2 cj = C.lxy[k];
                    double d2;
                    if (C.general) d2 = exact_d2<true>(b, cj.x, cj.y, C.lz[k], C.cxw[c], C.cyw[c], C.czw[c], C.lsh[k]);
                    else d2 = exact_d2<false>(b, cj.x, cj.y, C.lz[k], C.cxw[c], C.cyw[c], C.czw[c], 0);
                    C.verlet[o] = __float_as_int(C.f4[k].w);
                    C.dist[o] = sqrt(d2); // neighbor.cpp:174
                } else if (C.write_pads) {
                    C.verlet[o] = -1;
                    C.dist[o] = C.pad;
                }
            }
        }
        wave_sync();
    }
    K.count = 0;
}

// Centre cell with a run longer than S (or more centres than one group holds): the runs one after the other in windows
// of 64 lanes, every window tested against all centres of the cell (lane n of `me` / `myc` holds centre n and its count).
template <bool COUNT>
__device__ __forceinline__ void slow_cell(const Ctx &C, const DBox &b, Cols &K, int cb, int c0, int ncen)
{
    const int lane = threadIdx.x & 63;
    if (lane == 0) atomicAdd(&g_dbg[0], 1);
    for (int n0 = 0; n0 < ncen;) {
        if (K.count == 64)
            flush<COUNT>(C, b, K);
        const int chunk = min(ncen - n0, 64 - K.count);
        if (lane < chunk)
            C.ccol[K.count + lane] = (unsigned short)(c0 + n0 + lane);
        const float4 me = C.f4[c0 + n0 + min(lane, chunk - 1)];
        int myc = 0;
        for (int r = 0; r < 9; ++r) { // neighbor.cpp:147-151
            const unsigned hv = (unsigned)uni((int)C.hr[cb + ((r / 3 - 1) * C.HXY + (r % 3 - 1)) * C.HZ]);
            const int k0 = (int)(hv & 0xffffu), k3 = k0 + (int)(hv >> 16);
            for (int w0 = k0; w0 < k3; w0 += 64) {
                const int idx = w0 + lane;
                const float4 q = C.f4[idx];
                for (int n = 0; n < chunk; ++n) {
                    const int li = c0 + n0 + n;
                    const bool h = eval_hit(C, b, q.x, q.y, q.z, lane_f(me.x, n), lane_f(me.y, n), lane_f(me.z, n), idx < k3 && idx != li, idx, li);
                    const unsigned long long hm = __ballot(h);
                    const int base = __builtin_amdgcn_readlane(myc, n);
                    if (!COUNT) {
                        const int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
                        if (h && pos < C.M) C.tk[(K.count + n) * C.M + pos] = (unsigned short)idx;
                    }
                    set_lane(myc, base + __builtin_popcountll(hm), n);
                }
            }
        }
        const int moved = __shfl(myc, lane - K.count, 64); // counts of centres 0..chunk-1 -> ticket columns count..count+chunk-1
        if (lane >= K.count && lane < K.count + chunk) K.cntv = moved;
        K.count += chunk;
        n0 += chunk;
    }
}

// One centre against two groups of 64 candidates, hand-scheduled: 6 packed f32 instructions give both groups' squared
// distances, the compares leave the hit masks in VCC, v_mbcnt turns a mask into the slot of each hit, and the tickets are
// stored under EXEC = hit mask.  Returns the hit counts of the groups and the union of the pairs inside the decision band.
// ok0 / ok1: lanes that hold a candidate (self excluded); tk_col: LDS byte address of the ticket column.
// Tickets past M spill into the following columns (written later) or the slack behind the last one.
template <bool COUNT>
__device__ __forceinline__ void centre2(float2v ux, float2v uy, float2v uz, float sx, float sy, float sz, float lo, float hi,
                                        unsigned long long ok0, unsigned long long ok1, int idx0, int idx1, unsigned tk_col,
                                        int &h0, int &h1, unsigned long long &unc)
{
    const unsigned long long SX = (unsigned)__float_as_int(sx), SY = (unsigned)__float_as_int(sy), SZ = (unsigned)__float_as_int(sz);
    unsigned long long save, t;
    unsigned tk1;
    if (COUNT) {
        asm volatile(
            "s_nop 1\n\t"
            "v_pk_add_f32 v[118:119], %[UX], %[SX] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 v[120:121], %[UY], %[SY] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 v[122:123], %[UZ], %[SZ] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_mul_f32 v[118:119], v[118:119], v[118:119]\n\t"
            "v_pk_fma_f32 v[118:119], v[120:121], v[120:121], v[118:119]\n\t"
            "v_pk_fma_f32 v[118:119], v[122:123], v[122:123], v[118:119]\n\t"
            "v_cmp_ge_f32 vcc, %[HI], v118\n\t"
            "s_and_b64 %[UNC], vcc, %[OK0]\n\t"
            "v_cmp_gt_f32 vcc, %[LO], v118\n\t"
            "s_and_b64 vcc, vcc, %[OK0]\n\t"
            "s_xor_b64 %[UNC], %[UNC], vcc\n\t"
            "s_bcnt1_i32_b64 %[H0], vcc\n\t"
            "v_cmp_ge_f32 vcc, %[HI], v119\n\t"
            "s_and_b64 %[T], vcc, %[OK1]\n\t"
            "v_cmp_gt_f32 vcc, %[LO], v119\n\t"
            "s_and_b64 vcc, vcc, %[OK1]\n\t"
            "s_xor_b64 %[T], %[T], vcc\n\t"
            "s_or_b64 %[UNC], %[UNC], %[T]\n\t"
            "s_bcnt1_i32_b64 %[H1], vcc\n\t"
            : [H0] "=&s"(h0), [H1] "=&s"(h1), [UNC] "=&s"(unc), [T] "=&s"(t)
            : [UX] "v"(ux), [UY] "v"(uy), [UZ] "v"(uz), [SX] "s"(SX), [SY] "s"(SY), [SZ] "s"(SZ), [LO] "s"(lo), [HI] "s"(hi),
              [OK0] "s"(ok0), [OK1] "s"(ok1)
            : "vcc", "scc", "v118", "v119", "v120", "v121", "v122", "v123");
    } else {
        asm volatile(
            "s_nop 1\n\t"
            "v_pk_add_f32 v[118:119], %[UX], %[SX] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 v[120:121], %[UY], %[SY] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 v[122:123], %[UZ], %[SZ] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_mul_f32 v[118:119], v[118:119], v[118:119]\n\t"
            "v_pk_fma_f32 v[118:119], v[120:121], v[120:121], v[118:119]\n\t"
            "v_pk_fma_f32 v[118:119], v[122:123], v[122:123], v[118:119]\n\t"
            "s_mov_b64 %[SAVE], exec\n\t"
            "v_cmp_ge_f32 vcc, %[HI], v118\n\t"
            "s_and_b64 %[UNC], vcc, %[OK0]\n\t"
            "v_cmp_gt_f32 vcc, %[LO], v118\n\t"
            "s_and_b64 vcc, vcc, %[OK0]\n\t"
            "s_xor_b64 %[UNC], %[UNC], vcc\n\t"
            "v_mbcnt_lo_u32_b32 v124, vcc_lo, 0\n\t"
            "v_mbcnt_hi_u32_b32 v124, vcc_hi, v124\n\t"
            "s_bcnt1_i32_b64 %[H0], vcc\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "v_lshl_add_u32 v124, v124, 1, %[TK]\n\t"
            "ds_write_b16 v124, %[IDX0]\n\t"
            "s_mov_b64 exec, %[SAVE]\n\t"
            "v_cmp_ge_f32 vcc, %[HI], v119\n\t"
            "s_and_b64 %[T], vcc, %[OK1]\n\t"
            "v_cmp_gt_f32 vcc, %[LO], v119\n\t"
            "s_and_b64 vcc, vcc, %[OK1]\n\t"
            "s_xor_b64 %[T], %[T], vcc\n\t"
            "s_or_b64 %[UNC], %[UNC], %[T]\n\t"
            "v_mbcnt_lo_u32_b32 v125, vcc_lo, 0\n\t"
            "v_mbcnt_hi_u32_b32 v125, vcc_hi, v125\n\t"
            "s_bcnt1_i32_b64 %[H1], vcc\n\t"
            "s_lshl1_add_u32 %[TK1], %[H0], %[TK]\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "v_lshl_add_u32 v125, v125, 1, %[TK1]\n\t"
            "ds_write_b16 v125, %[IDX1]\n\t"
            "s_mov_b64 exec, %[SAVE]\n\t"
            : [H0] "=&s"(h0), [H1] "=&s"(h1), [UNC] "=&s"(unc), [T] "=&s"(t), [SAVE] "=&s"(save), [TK1] "=&s"(tk1)
            : [UX] "v"(ux), [UY] "v"(uy), [UZ] "v"(uz), [SX] "s"(SX), [SY] "s"(SY), [SZ] "s"(SZ), [LO] "s"(lo), [HI] "s"(hi),
              [OK0] "s"(ok0), [OK1] "s"(ok1), [IDX0] "v"(idx0), [IDX1] "v"(idx1), [TK] "s"(tk_col)
            : "vcc", "scc", "memory", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125");
    }
}

// S lanes per run, 64/S runs per group, NG groups cover the 9 runs
template <int S, int NG, bool COUNT>
__global__ __launch_bounds__(NT) void k_neighbor_wave(
    const double *__restrict__ xs, const double *__restrict__ ys, const double *__restrict__ zs,
    const int *__restrict__ order, const unsigned char *__restrict__ mvs, const int *__restrict__ cell_start, DBox b,
    Grid g, double rc, float lo, float hi, int *__restrict__ verlet, double *__restrict__ dist, int *__restrict__ nn,
    int M, int mp_shift, int write_pads, int cap, int *__restrict__ flags, unsigned char *__restrict__ tile_flag, int nt0,
    int nt1, int nt2, Shape ts, const int *__restrict__ tile_list, const int *__restrict__ n_live, int list_mode,
    int *__restrict__ max_count, int *__restrict__ flagged)
{
    static_assert(S * (64 / S) <= 64 && NG * (64 / S) >= 9, "groups must cover the 9 runs");
    constexpr int RPR = 64 / S;           // runs per group
    constexpr int G4 = 4 / RPR;           // group that holds run 4 (the centre's own column)
    constexpr int B4 = (4 % RPR) * S;     // first lane of run 4 inside its group
    if (flags[0] != 0) // unwrapped input: the image codes are not valid, the thread-per-atom kernel takes the whole call
        return;
    const int TXY = ts.txy, TZ = ts.tz;
    const int HXY = TXY + 2, HZ = TZ + 2, NH = HXY * HXY * HZ;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = uni(tid >> 6);
    float4 *f4 = reinterpret_cast<float4 *>(smem);
    double2 *lxy = reinterpret_cast<double2 *>(f4 + cap + 64);
    double *lz = reinterpret_cast<double *>(lxy + cap);
    const int tk_bytes = ((64 * M + NG * 64) * 2 + 7) & ~7;
    const int wave_bytes = 3 * 64 * 8 + 2 * 64 * 4 + 64 * 2 + tk_bytes;
    unsigned char *wbase = reinterpret_cast<unsigned char *>(lz + cap) + (size_t)wv * wave_bytes;
    unsigned short *lsh = reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(lz + cap) + (size_t)4 * wave_bytes);
    __shared__ unsigned hc[MAX_NH + 2];
    __shared__ unsigned hr[MAX_NH + 2];
    __shared__ int scan_tmp[4];
    __shared__ int s_flag[2];

    Ctx C;
    C.f4 = f4; C.lxy = lxy; C.lz = lz; C.lsh = lsh; C.hc = hc; C.hr = hr;
    C.cxw = reinterpret_cast<double *>(wbase); C.cyw = C.cxw + 64; C.czw = C.cyw + 64;
    C.crow = reinterpret_cast<int *>(C.czw + 64); C.ccnt = C.crow + 64;
    C.ccol = reinterpret_cast<unsigned short *>(C.ccnt + 64);
    C.tk = C.ccol + 64;
    C.tk_lds = (unsigned)(unsigned long)(lds_byte *)smem + (unsigned)(reinterpret_cast<unsigned char *>(C.tk) - smem);
    C.lo = lo; C.hi = hi; C.rcsq = rc * rc; C.pad = rc + 1.0; // neighbor.cpp:127; pads neighbor.py:125-129
    C.HXY = HXY; C.HZ = HZ; C.M = M; C.mp_shift = mp_shift; C.write_pads = write_pads != 0;
    C.verlet = verlet; C.dist = dist; C.nn = nn;

    // lane -> (run, offset inside the run) of every group; the same for all cells
    const int rs = lane / S, off = lane - rs * S;
    int rel[NG], offv[NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int r = gi * RPR + rs;
        const bool lok = rs < RPR && r < 9;
        const int rr = lok ? r : 4;
        rel[gi] = ((rr / 3 - 1) * HXY + (rr % 3 - 1)) * HZ; // neighbor.cpp:147-151: r = (da+1)*3 + (db+1)
        offv[gi] = lok ? off : 0x7fff;                     // a lane without a run never holds a candidate
    }
    Cols K{0, 0, 0};

    // XCD-aware tile order: block b runs on XCD b%8; every XCD gets one contiguous chunk of the tiles THAT HOLD CENTRE ATOMS
    const int nlive = (list_mode && tile_list) ? *n_live : nt0 * nt1 * nt2;
    const int per = (nlive + 7) / 8;
    const double cw = rc; // cell width of the rc-wide grid (neighbor.cpp:29-62)
    for (int jt = (int)(blockIdx.x >> 3); jt < per; jt += (int)(gridDim.x >> 3)) {
        const int slot = (blockIdx.x & 7) * per + jt;
        if (slot >= nlive)
            break;
        const int tile_id = (list_mode && tile_list) ? tile_list[slot] : slot;
        const int t2 = tile_id % nt2, t1 = (tile_id / nt2) % nt1, t0 = tile_id / (nt2 * nt1);
        const int T0 = t0 * TXY, T1 = t1 * TXY, T2 = t2 * TZ;
        if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }

        // ---- halo cell of this thread: source range, image code
        int cnt = 0, src = 0, img = NEUTRAL, hz = 0;
        bool edge = false; // first / last cell of an open axis: atoms outside the box are clamped into it (neighbor.cpp:58-61)
        if (tid < NH) {
            const int hcol = tid / HZ;
            hz = tid - hcol * HZ;
            const int hx = hcol / HXY, hy = hcol - hx * HXY;
            const int g0 = T0 + hx - 1, g1 = T1 + hy - 1, g2 = T2 + hz - 1;
            // a cell beyond an OPEN face is the far side of the box in the reference's modulo walk (neighbor.cpp:18-27); with
            // >= 4 cells on the axis its atoms are >= 2 rc from every centre of this tile: no hits, not staged
            const bool in0 = b.pbc[0] ? (g0 >= -1 && g0 <= g.nc[0]) : (g0 >= 0 && g0 < g.nc[0]);
            const bool in1 = b.pbc[1] ? (g1 >= -1 && g1 <= g.nc[1]) : (g1 >= 0 && g1 < g.nc[1]);
            const bool in2 = b.pbc[2] ? (g2 >= -1 && g2 <= g.nc[2]) : (g2 >= 0 && g2 < g.nc[2]);
            if (in0 && in1 && in2) {
                const int a0 = g0 < 0 ? g0 + g.nc[0] : (g0 >= g.nc[0] ? g0 - g.nc[0] : g0);
                const int a1 = g1 < 0 ? g1 + g.nc[1] : (g1 >= g.nc[1] ? g1 - g.nc[1] : g1);
                const int a2 = g2 < 0 ? g2 + g.nc[2] : (g2 >= g.nc[2] ? g2 - g.nc[2] : g2);
                const int64_t c = ((int64_t)a0 * g.nc[1] + a1) * g.nc[2] + a2;
                src = cell_start[c];
                cnt = cell_start[c + 1] - src;
                // image of the candidate cell seen from an in-grid centre cell: below the box -> raw coordinates are ~+L
                // away (n = +1); above -> n = -1
                const int n0 = g0 < 0 ? 1 : (g0 >= g.nc[0] ? -1 : 0);
                const int n1 = g1 < 0 ? 1 : (g1 >= g.nc[1] ? -1 : 0);
                const int n2 = g2 < 0 ? 1 : (g2 >= g.nc[2] ? -1 : 0);
                img = (n0 + 1) | ((n1 + 1) << 2) | ((n2 + 1) << 4);
                edge = (!b.pbc[0] && (a0 == 0 || a0 == g.nc[0] - 1)) || (!b.pbc[1] && (a1 == 0 || a1 == g.nc[1] - 1)) ||
                       (!b.pbc[2] && (a2 == 0 || a2 == g.nc[2] - 1));
            }
        }
        int total;
        const int off0 = excl_scan_block(cnt, scan_tmp, &total);
        if (total > cap) { // leave this tile to the thread-per-atom kernel
            if (tid == 0) {
                tile_flag[tile_id] = 1;
                flagged[atomicAdd(&flags[2], 1)] = tile_id;
                atomicAdd(&g_dbg[1], 1);
            }
            continue; // (excl_scan_block ended with a barrier)
        }
        if (tid < NH) hc[tid] = (unsigned)off0 | ((unsigned)cnt << 16);
        // ---- stage this cell's atoms: raw doubles, image code, and single-precision coordinates relative to the tile
        // corner with the image shift folded in
        if (cnt > 0) {
            const double X0 = b.o[0] + (double)(T0 - 1) * cw, Y0 = b.o[1] + (double)(T1 - 1) * cw, Z0 = b.o[2] + (double)(T2 - 1) * cw;
            const int code0 = combine_codes(img, NEUTRAL); // an atom inside the box (image code 0): the cell's own shift
            const double XS = X0 + b.h[0] * (double)((code0 & 7) - 2), YS = Y0 + b.h[4] * (double)(((code0 >> 3) & 7) - 2),
                         ZS = Z0 + b.h[8] * (double)(((code0 >> 6) & 7) - 2);
            const float flo = (float)(-1.5 * cw);
            const float fhx = (float)(((double)HXY + 1.5) * cw), fhz = (float)(((double)HZ + 1.5) * cw);
            bool general = code0 != NEUTRAL3, far = false;
            for (int k = 0; k < cnt; k += 4) {
                double a[4], bb[4], c[4];
                int d[4];
                unsigned char m[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int q = src + min(k + v, cnt - 1);
                    a[v] = xs[q]; bb[v] = ys[q]; c[v] = zs[q]; d[v] = order[q]; m[v] = mvs[q];
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (k + v < cnt) {
                        int code = code0;
                        float ux = (float)(a[v] - XS), uy = (float)(bb[v] - YS), uz = (float)(c[v] - ZS);
                        if (m[v] != NEUTRAL) { // an atom handed in outside the box on a periodic axis: its own image number on top
                            code = combine_codes(img, m[v]);
                            ux = (float)((a[v] - b.h[0] * (double)((code & 7) - 2)) - X0);
                            uy = (float)((bb[v] - b.h[4] * (double)(((code >> 3) & 7) - 2)) - Y0);
                            uz = (float)((c[v] - b.h[8] * (double)(((code >> 6) & 7) - 2)) - Z0);
                            general = true;
                        }
                        // the decision band assumes coordinates inside the tile's halo; an atom clamped into an edge cell from
                        // far outside the box (open axis) sends the tile to the thread-per-atom kernel
                        if (edge)
                            far = far || !(ux >= flo && ux <= fhx && uy >= flo && uy <= fhx && uz >= flo && uz <= fhz);
                        const int p = off0 + k + v;
                        f4[p] = make_float4(ux, uy, uz, __int_as_float(d[v]));
                        lxy[p] = make_double2(a[v], bb[v]);
                        lz[p] = c[v];
                        lsh[p] = (unsigned short)code;
                    }
                }
            }
            if (general) s_flag[0] = 1;
            if (far) s_flag[1] = 1;
        }
        __syncthreads(); // publishes hc, the staged atoms and the two flags
        if (s_flag[1]) {
            if (tid == 0) {
                tile_flag[tile_id] = 1;
                flagged[atomicAdd(&flags[2], 1)] = tile_id;
                atomicAdd(&g_dbg[2], 1);
            }
            if (list_mode) __syncthreads();
            continue;
        }
        C.general = s_flag[0] != 0;
        // the 3-cell run around every cell that can be a centre's neighbour column entry (cells hz-1..hz+1 are contiguous in LDS)
        if (tid < NH && hz >= 1 && hz <= HZ - 2) {
            const unsigned lo_c = hc[tid - 1], hi_c = hc[tid + 1];
            const unsigned k0 = lo_c & 0xffffu, k3 = (hi_c & 0xffffu) + (hi_c >> 16);
            hr[tid] = k0 | ((k3 - k0) << 16);
        }
        __syncthreads();

        // ---- this wave's centre cells, round-robin over the tile in (x, y, z)-lexicographic order
        {
            const int ncc = TXY * TXY * TZ;
            int cz = wv % TZ, tq = wv / TZ;
            int cy = tq % TXY, cx = tq / TXY;
            for (int q = wv; q < ncc; q += 4) {
                if (T0 + cx < g.nc[0] && T1 + cy < g.nc[1] && T2 + cz < g.nc[2]) {
                    const int cb = ((cx + 1) * HXY + (cy + 1)) * HZ + cz + 1;
                    const unsigned vc = (unsigned)uni((int)hc[cb]);
                    const int c0 = (int)(vc & 0xffffu), ncen = (int)(vc >> 16);
                    if (ncen > 0) {
                        int idx[NG];
                        unsigned long long ok[NG], over = 0ull;
#pragma unroll
                        for (int gi = 0; gi < NG; ++gi) {
                            const unsigned hv = hr[cb + rel[gi]];
                            const unsigned len = hv >> 16;
                            idx[gi] = (int)(hv & 0xffffu) + off;
                            ok[gi] = __builtin_amdgcn_uicmp((unsigned)offv[gi], len, 36 /* ult */);
                            over |= __builtin_amdgcn_uicmp(len, (unsigned)S, 34 /* ugt */);
                        }
                        if (__builtin_expect(over != 0ull, 0)) {
                            slow_cell<COUNT>(C, b, K, cb, c0, ncen);
                        } else {
                            if (K.count + ncen > 64)
                                flush<COUNT>(C, b, K);
                            float ux[NG], uy[NG], uz[NG];
#pragma unroll
                            for (int gi = 0; gi < NG; ++gi) {
                                const float4 v = f4[idx[gi]];
                                ux[gi] = v.x; uy[gi] = v.y; uz[gi] = v.z;
                            }
                            // the centres are lanes of run 4: lane sl0 + n holds centre n
                            const int sl0 = B4 + (c0 - __builtin_amdgcn_readlane(idx[G4], B4));
                            if (lane < ncen)
                                C.ccol[K.count + lane] = (unsigned short)(c0 + lane);
                            if (NG == 2) {
                                const float2v UX = {ux[0], ux[NG - 1]}, UY = {uy[0], uy[NG - 1]}, UZ = {uz[0], uz[NG - 1]};
                                for (int n = 0; n < ncen; ++n) {
                                    const int sl = sl0 + n;
                                    const float sx = lane_f(ux[G4], sl), sy = lane_f(uy[G4], sl), sz = lane_f(uz[G4], sl);
                                    const unsigned long long self = 1ull << sl;
                                    const unsigned long long ok0 = G4 == 0 ? ok[0] & ~self : ok[0], ok1 = G4 == 0 ? ok[NG - 1] : ok[NG - 1] & ~self;
                                    int h0, h1;
                                    unsigned long long unc;
                                    centre2<COUNT>(UX, UY, UZ, sx, sy, sz, lo, hi, ok0, ok1, idx[0], idx[NG - 1],
                                                   C.tk_lds + (unsigned)((K.count + n) * M * 2), h0, h1, unc);
                                    int hits = h0 + h1;
                                    if (__builtin_expect(unc != 0ull, 0)) { // a pair inside the decision band: this centre again, every pair checked
                                        const int li = c0 + n;
                                        hits = 0;
#pragma unroll
                                        for (int gi = 0; gi < NG; ++gi) {
                                            const bool okl = ((gi == 0 ? ok0 : ok1) >> lane) & 1ull;
                                            const bool h = eval_hit(C, b, ux[gi], uy[gi], uz[gi], sx, sy, sz, okl, idx[gi], li);
                                            const unsigned long long hm = __ballot(h);
                                            if (!COUNT) {
                                                const int pos = hits + __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
                                                if (h && pos < M) C.tk[(K.count + n) * M + pos] = (unsigned short)idx[gi];
                                            }
                                            hits += __builtin_popcountll(hm);
                                        }
                                    }
                                    { int keep; asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(K.cntv), "=&s"(keep) : "s"(hits), "s"(K.count + n)); }
                                }
                            } else {
                                for (int n = 0; n < ncen; ++n) {
                                    const int sl = sl0 + n, li = c0 + n;
                                    const float sx = lane_f(ux[G4], sl), sy = lane_f(uy[G4], sl), sz = lane_f(uz[G4], sl);
                                    const unsigned long long self = 1ull << sl;
                                    int hits = 0;
#pragma unroll
                                    for (int gi = 0; gi < NG; ++gi) {
                                        const unsigned long long okm = gi == G4 ? ok[gi] & ~self : ok[gi];
                                        const bool h = eval_hit(C, b, ux[gi], uy[gi], uz[gi], sx, sy, sz, (okm >> lane) & 1ull, idx[gi], li);
                                        const unsigned long long hm = __ballot(h);
                                        if (!COUNT) {
                                            const int pos = hits + __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
                                            if (h && pos < M) C.tk[(K.count + n) * M + pos] = (unsigned short)idx[gi];
                                        }
                                        hits += __builtin_popcountll(hm);
                                    }
                                    set_lane(K.cntv, hits, K.count + n);
                                }
                            }
                            K.count += ncen;
                        }
                    }
                }
                cz += 4;
                while (cz >= TZ) {
                    cz -= TZ;
                    if (++cy == TXY) { cy = 0; ++cx; }
                }
            }
            if (K.count)
                flush<COUNT>(C, b, K);
        }
        if (!list_mode)
            break;
        if (jt + (int)(gridDim.x >> 3) < per) __syncthreads(); // LDS is reused by the next tile
    } // tiles of this workgroup
    if (COUNT) {
        int m = K.vmax;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
        if (lane == 0 && m > 0) atomicMax(max_count, m);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_neighbor_lane: same staging, one THREAD per centre atom.  A thread walks its 9 runs out of LDS in the reference's
// order with one 16-byte LDS read and ~9 single-precision VALU instructions per candidate, and records the hits of a
// run as a BIT MASK in a register (mask = 2*mask + hit: one add-with-carry on the compare's VCC) — no per-candidate
// branch, ticket store or address arithmetic.  After the scan the masks are expanded into 2-byte tickets (a few hits per
// run), and the workgroup writes the rows cooperatively exactly as above (double-precision recompute, MP lanes per row).
// Pairs inside the single-precision decision band make the thread redo its masks with the reference's own expression.
// ---------------------------------------------------------------------------------------------------------------
static constexpr int CEN_CAP = 512; // centre atoms a tile may hold

// One run of candidates for one centre per lane, hand-scheduled.  Four candidates per trip: four 16-byte LDS reads in
// flight, then per candidate 3 subtractions, an FMA chain that ends in d' = d2 - rc^2, one compare whose VCC is shifted
// into the trip's hit nibble by an add-with-carry, and one v_min that tracks the smallest |d'| the lane has seen (the
// decision-band test happens once per centre).  Lanes whose run is exhausted leave EXEC; the others go on.
// a: LDS byte address of the run's first candidate; rem: its length; bit (L4-1-j) of `mask` is candidate j, L4 = length
// rounded up to 4.  Candidates past the end of a run are other staged atoms: their hit bits are masked, a small |d'| of
// theirs only costs a redundant double-precision pass.
__device__ __forceinline__ void scan_run_asm(unsigned a, int rem, float sx, float sy, float sz, float nmid, float negT,
                                             unsigned &mask, float &w)
{
    unsigned long long save;
    unsigned m;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_mov_b32 %[m], 0\n"
        ".Lscan_top_%=:\n\t"
        "v_cmp_lt_i32 vcc, 0, %[rem]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lscan_end_%=\n\t"
        "ds_read_b128 v[130:133], %[a]\n\t"
        "ds_read_b128 v[134:137], %[a] offset:16\n\t"
        "ds_read_b128 v[138:141], %[a] offset:32\n\t"
        "ds_read_b128 v[142:145], %[a] offset:48\n\t"
        "v_min_u32 v150, 4, %[rem]\n\t"
        "v_sub_u32 v151, 4, v150\n\t"
        "v_bfm_b32 v150, v150, v151\n\t"        // ((1 << valid) - 1) << (4 - valid): the valid candidates of this trip
        "v_add_u32 %[a], 64, %[a]\n\t"
        "v_add_u32 %[rem], -4, %[rem]\n\t"
        "v_mov_b32 v149, 0\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_sub_f32 v146, v130, %[sx]\n\t"
        "v_sub_f32 v147, v131, %[sy]\n\t"
        "v_sub_f32 v148, v132, %[sz]\n\t"
        "v_fma_f32 v146, v146, v146, %[nmid]\n\t"
        "v_fmac_f32 v146, v147, v147\n\t"
        "v_fmac_f32 v146, v148, v148\n\t"
        "v_cmp_gt_f32 vcc, %[negT], v146\n\t"
        "v_addc_co_u32 v149, vcc, v149, v149, vcc\n\t"
        "v_min_f32_e64 %[w], %[w], |v146|\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_sub_f32 v146, v134, %[sx]\n\t"
        "v_sub_f32 v147, v135, %[sy]\n\t"
        "v_sub_f32 v148, v136, %[sz]\n\t"
        "v_fma_f32 v146, v146, v146, %[nmid]\n\t"
        "v_fmac_f32 v146, v147, v147\n\t"
        "v_fmac_f32 v146, v148, v148\n\t"
        "v_cmp_gt_f32 vcc, %[negT], v146\n\t"
        "v_addc_co_u32 v149, vcc, v149, v149, vcc\n\t"
        "v_min_f32_e64 %[w], %[w], |v146|\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_sub_f32 v146, v138, %[sx]\n\t"
        "v_sub_f32 v147, v139, %[sy]\n\t"
        "v_sub_f32 v148, v140, %[sz]\n\t"
        "v_fma_f32 v146, v146, v146, %[nmid]\n\t"
        "v_fmac_f32 v146, v147, v147\n\t"
        "v_fmac_f32 v146, v148, v148\n\t"
        "v_cmp_gt_f32 vcc, %[negT], v146\n\t"
        "v_addc_co_u32 v149, vcc, v149, v149, vcc\n\t"
        "v_min_f32_e64 %[w], %[w], |v146|\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sub_f32 v146, v142, %[sx]\n\t"
        "v_sub_f32 v147, v143, %[sy]\n\t"
        "v_sub_f32 v148, v144, %[sz]\n\t"
        "v_fma_f32 v146, v146, v146, %[nmid]\n\t"
        "v_fmac_f32 v146, v147, v147\n\t"
        "v_fmac_f32 v146, v148, v148\n\t"
        "v_cmp_gt_f32 vcc, %[negT], v146\n\t"
        "v_addc_co_u32 v149, vcc, v149, v149, vcc\n\t"
        "v_min_f32_e64 %[w], %[w], |v146|\n\t"
        "v_and_b32 v149, v149, v150\n\t"
        "v_lshl_or_b32 %[m], %[m], 4, v149\n\t"
        "s_branch .Lscan_top_%=\n"
        ".Lscan_end_%=:\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        : [m] "=&v"(m), [w] "+v"(w), [a] "+v"(a), [rem] "+v"(rem), [save] "=&s"(save)
        : [sx] "v"(sx), [sy] "v"(sy), [sz] "v"(sz), [nmid] "v"(nmid), [negT] "s"(negT)
        : "vcc", "scc", "memory", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141",
          "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151");
    mask = m;
}

// the same run decided by the reference's double-precision expression (threads with a pair inside the decision band)
template <bool SELF>
__device__ __forceinline__ unsigned scan_run_f64(const Ctx &C, const DBox &b, int k0, int len, int li, double xi, double yi, double zi)
{
    unsigned m = 0;
    const int L4 = (len + 3) & ~3;
    for (int j = 0; j < L4; ++j) {
        const int k = k0 + j;
        bool h = false;
        if (j < len) {
            const double2 cj = C.lxy[k];
            const double d2 = exact_d2<true>(b, cj.x, cj.y, C.lz[k], xi, yi, zi, C.lsh[k]);
            h = (d2 <= C.rcsq) && (!SELF || k != li); // neighbor.cpp:162,171
        }
        m = m + m + (h ? 1u : 0u);
    }
    return m;
}

template <bool COUNT>
__global__ __launch_bounds__(NT) void k_neighbor_lane(
    const double *__restrict__ xs, const double *__restrict__ ys, const double *__restrict__ zs,
    const int *__restrict__ order, const unsigned char *__restrict__ mvs, const int *__restrict__ cell_start, DBox b,
    Grid g, double rc, float lo, float hi, int *__restrict__ verlet, double *__restrict__ dist, int *__restrict__ nn,
    int M, int mp_shift, int write_pads, int cap, int *__restrict__ flags, unsigned char *__restrict__ tile_flag, int nt0,
    int nt1, int nt2, Shape ts, const int *__restrict__ tile_list, const int *__restrict__ n_live, int list_mode,
    int *__restrict__ max_count, int *__restrict__ flagged, int dbg, const int *__restrict__ parent, int parent_nt2, int nsub,
    int flag_slot)
{
    // parent != nullptr: second pass over the tiles the first pass listed (halo over the LDS budget): the same tiling cut into
    // nsub slices along z (this launch's TZ = parent's TZ / nsub); what still does not fit goes to `flagged` (counter
    // flags[flag_slot]) and from there to the thread-per-atom code
    if (flags[0] != 0) // unwrapped input: the image codes are not valid, the thread-per-atom kernel takes the whole call
        return;
    const int TXY = ts.txy, TZ = ts.tz;
    const int HXY = TXY + 2, HZ = TZ + 2, NH = HXY * HXY * HZ;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    float4 *f4 = reinterpret_cast<float4 *>(smem);
    double2 *lxy = reinterpret_cast<double2 *>(f4 + cap + 8);
    double *lz = reinterpret_cast<double *>(lxy + cap);
    double *cxw = lz + cap, *cyw = cxw + NT, *czw = cyw + NT; // wrapped centre of this pass's rows
    int *crow = reinterpret_cast<int *>(czw + NT), *ccnt = crow + NT;
    unsigned *cen = reinterpret_cast<unsigned *>(ccnt + NT);       // [CEN_CAP] centre atoms of the tile: LDS index | halo cell << 16
    unsigned short *tk = reinterpret_cast<unsigned short *>(cen + CEN_CAP); // [NT][M] tickets
    unsigned short *lsh = tk + (size_t)NT * M + (((size_t)NT * M) & 1);
    const unsigned f4_lds = (unsigned)(unsigned long)(lds_byte *)smem;
    __shared__ unsigned hc[MAX_NH + 2];
    __shared__ unsigned hr[MAX_NH + 2];
    __shared__ int scan_tmp[4];
    __shared__ int s_flag[3];

    Ctx C;
    C.f4 = f4; C.lxy = lxy; C.lz = lz; C.lsh = lsh; C.hc = hc; C.hr = hr;
    C.cxw = cxw; C.cyw = cyw; C.czw = czw; C.crow = crow; C.ccnt = ccnt; C.ccol = nullptr; C.tk = tk; C.tk_lds = 0;
    C.lo = lo; C.hi = hi; C.rcsq = rc * rc; C.pad = rc + 1.0; // neighbor.cpp:127; pads neighbor.py:125-129
    C.HXY = HXY; C.HZ = HZ; C.M = M; C.mp_shift = mp_shift; C.write_pads = write_pads != 0;
    C.verlet = verlet; C.dist = dist; C.nn = nn;
    int vmax = 0;

    const int nlive = parent ? min(*n_live, nt0 * nt1 * parent_nt2) * nsub : ((list_mode && tile_list) ? *n_live : nt0 * nt1 * nt2);
    const int per = (nlive + 7) / 8;
    const double cw = rc;
    for (int jt = (int)(blockIdx.x >> 3); jt < per; jt += (int)(gridDim.x >> 3)) {
        const int slot = (blockIdx.x & 7) * per + jt;
        if (slot >= nlive)
            break;
        int tile_id, t0, t1, t2;
        if (parent) {
            const int pt = parent[slot / nsub]; // tile of the first pass
            t2 = (pt % parent_nt2) * nsub + slot % nsub;
            t1 = (pt / parent_nt2) % nt1;
            t0 = pt / (parent_nt2 * nt1);
            tile_id = (t0 * nt1 + t1) * nt2 + t2;
        } else {
            tile_id = (list_mode && tile_list) ? tile_list[slot] : slot;
            t2 = tile_id % nt2; t1 = (tile_id / nt2) % nt1; t0 = tile_id / (nt2 * nt1);
        }
        const int T0 = t0 * TXY, T1 = t1 * TXY, T2 = t2 * TZ;
        if (T2 >= g.nc[2]) // (a slice beyond the grid: the parent tile was a clipped one)
            continue;
        if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; s_flag[2] = 0; }

        // ---- halo cell of this thread: source range, image code, is it a centre cell
        int cnt = 0, src = 0, img = NEUTRAL, hz = 0;
        bool edge = false, centre_cell = false;
        if (tid < NH) {
            const int hcol = tid / HZ;
            hz = tid - hcol * HZ;
            const int hx = hcol / HXY, hy = hcol - hx * HXY;
            const int g0 = T0 + hx - 1, g1 = T1 + hy - 1, g2 = T2 + hz - 1;
            const bool in0 = b.pbc[0] ? (g0 >= -1 && g0 <= g.nc[0]) : (g0 >= 0 && g0 < g.nc[0]);
            const bool in1 = b.pbc[1] ? (g1 >= -1 && g1 <= g.nc[1]) : (g1 >= 0 && g1 < g.nc[1]);
            const bool in2 = b.pbc[2] ? (g2 >= -1 && g2 <= g.nc[2]) : (g2 >= 0 && g2 < g.nc[2]);
            if (in0 && in1 && in2) {
                const int a0 = g0 < 0 ? g0 + g.nc[0] : (g0 >= g.nc[0] ? g0 - g.nc[0] : g0);
                const int a1 = g1 < 0 ? g1 + g.nc[1] : (g1 >= g.nc[1] ? g1 - g.nc[1] : g1);
                const int a2 = g2 < 0 ? g2 + g.nc[2] : (g2 >= g.nc[2] ? g2 - g.nc[2] : g2);
                const int64_t c = ((int64_t)a0 * g.nc[1] + a1) * g.nc[2] + a2;
                src = cell_start[c];
                cnt = cell_start[c + 1] - src;
                const int n0 = g0 < 0 ? 1 : (g0 >= g.nc[0] ? -1 : 0);
                const int n1 = g1 < 0 ? 1 : (g1 >= g.nc[1] ? -1 : 0);
                const int n2 = g2 < 0 ? 1 : (g2 >= g.nc[2] ? -1 : 0);
                img = (n0 + 1) | ((n1 + 1) << 2) | ((n2 + 1) << 4);
                edge = (!b.pbc[0] && (a0 == 0 || a0 == g.nc[0] - 1)) || (!b.pbc[1] && (a1 == 0 || a1 == g.nc[1] - 1)) ||
                       (!b.pbc[2] && (a2 == 0 || a2 == g.nc[2] - 1));
                centre_cell = hx >= 1 && hx <= TXY && hy >= 1 && hy <= TXY && hz >= 1 && hz <= TZ && g0 < g.nc[0] && g1 < g.nc[1] && g2 < g.nc[2];
            }
        }
        int total2;
        const int off2 = excl_scan_block(cnt | (centre_cell ? cnt << 16 : 0), scan_tmp, &total2); // both prefixes in one scan (each < 2^15)
        const int total = total2 & 0xffff, ncentres = total2 >> 16;
        const int off0 = off2 & 0xffff, coff = off2 >> 16;
        if (total > cap || ncentres > CEN_CAP) { // leave this tile to the thread-per-atom code
            if (tid == 0) {
                if (tile_flag) tile_flag[tile_id] = 1;
                flagged[atomicAdd(&flags[flag_slot], 1)] = tile_id;
                atomicAdd(&g_dbg[1], 1);
            }
            continue; // (excl_scan_block ended with a barrier)
        }
        if (tid < NH) hc[tid] = (unsigned)off0 | ((unsigned)cnt << 16);
        if (cnt > 0) {
            const double X0 = b.o[0] + (double)(T0 - 1) * cw, Y0 = b.o[1] + (double)(T1 - 1) * cw, Z0 = b.o[2] + (double)(T2 - 1) * cw;
            const int code0 = combine_codes(img, NEUTRAL);
            const double XS = X0 + b.h[0] * (double)((code0 & 7) - 2), YS = Y0 + b.h[4] * (double)(((code0 >> 3) & 7) - 2),
                         ZS = Z0 + b.h[8] * (double)(((code0 >> 6) & 7) - 2);
            const float flo = (float)(-1.5 * cw);
            const float fhx = (float)(((double)HXY + 1.5) * cw), fhz = (float)(((double)HZ + 1.5) * cw);
            bool general = code0 != NEUTRAL3, far = false;
            for (int k = 0; k < cnt; k += 4) {
                double a[4], bb[4], c[4];
                int d[4];
                unsigned char m[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int q = src + min(k + v, cnt - 1);
                    a[v] = xs[q]; bb[v] = ys[q]; c[v] = zs[q]; d[v] = order[q]; m[v] = mvs[q];
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (k + v < cnt) {
                        int code = code0;
                        float ux = (float)(a[v] - XS), uy = (float)(bb[v] - YS), uz = (float)(c[v] - ZS);
                        if (m[v] != NEUTRAL) {
                            code = combine_codes(img, m[v]);
                            ux = (float)((a[v] - b.h[0] * (double)((code & 7) - 2)) - X0);
                            uy = (float)((bb[v] - b.h[4] * (double)(((code >> 3) & 7) - 2)) - Y0);
                            uz = (float)((c[v] - b.h[8] * (double)(((code >> 6) & 7) - 2)) - Z0);
                            general = true;
                        }
                        if (edge)
                            far = far || !(ux >= flo && ux <= fhx && uy >= flo && uy <= fhx && uz >= flo && uz <= fhz);
                        const int p = off0 + k + v;
                        f4[p] = make_float4(ux, uy, uz, __int_as_float(d[v]));
                        lxy[p] = make_double2(a[v], bb[v]);
                        lz[p] = c[v];
                        lsh[p] = (unsigned short)code;
                        if (centre_cell) cen[coff + k + v] = (unsigned)p | ((unsigned)tid << 16);
                    }
                }
            }
            if (general) s_flag[0] = 1;
            if (far) s_flag[1] = 1;
        }
        __syncthreads(); // publishes hc, the staged atoms, the centre list and the flags
        if (tid < NH && hz >= 1 && hz <= HZ - 2) {
            const unsigned lo_c = hc[tid - 1], hi_c = hc[tid + 1];
            const unsigned k0 = lo_c & 0xffffu, k3 = (hi_c & 0xffffu) + (hi_c >> 16);
            hr[tid] = k0 | ((k3 - k0) << 16);
            if (k3 - k0 > 32u) s_flag[2] = 1; // a run's hit mask is one 32-bit register (length rounded up to 4)
        }
        __syncthreads();
        if (s_flag[1] | s_flag[2]) {
            if (tid == 0) {
                if (tile_flag) tile_flag[tile_id] = 1;
                flagged[atomicAdd(&flags[flag_slot], 1)] = tile_id;
                atomicAdd(&g_dbg[2], 1);
            }
            if (list_mode) __syncthreads();
            continue;
        }
        C.general = s_flag[0] != 0;
        if (dbg == 1) { if (list_mode) __syncthreads(); if (!list_mode) break; continue; }

        for (int base = 0; base < ncentres; base += NT) {
            const int q = base + tid;
            if (q < ncentres) {
                const unsigned cv = cen[q];
                const int li = (int)(cv & 0xffffu), cb = (int)(cv >> 16);
                const float4 s = f4[li];
                unsigned hv[9], mk[9];
#pragma unroll
                for (int r = 0; r < 9; ++r) { // neighbor.cpp:147-151: r = (da+1)*3 + (db+1)
                    hv[r] = hr[cb + ((r / 3 - 1) * HXY + (r % 3 - 1)) * HZ];
                }
                // lo = -(rc^2) and hi = T in this kernel: d' = d2 - rc^2 is a sure hit below -T, a sure miss above +T
                float w = 3.0e38f;
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    if (dbg == 3) { mk[r] = (hv[r] >> 16) ? 1u : 0u; continue; }
                    scan_run_asm(f4_lds + ((hv[r] & 0xffffu) << 4), (int)(hv[r] >> 16), s.x, s.y, s.z, lo, -hi, mk[r], w);
                }
                {   // the centre itself sits in run 4 with d2 = 0: not a neighbour (neighbor.cpp:162)
                    const int L4 = ((int)(hv[4] >> 16) + 3) & ~3;
                    mk[4] &= ~(1u << (L4 - 1 - (li - (int)(hv[4] & 0xffffu))));
                }
                const bool unc = !(w > hi);
                const double2 ci = lxy[li];
                double xi = ci.x, yi = ci.y, zi = lz[li];
                if (b.anypbc) // neighbor.cpp:139-142
                    wrap<false>(b, xi, yi, zi);
                if (__builtin_expect(unc, 0)) { // a pair inside the decision band: this centre again in double precision
                    atomicAdd(&g_dbg[3], 1);
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        if (r == 4) mk[r] = scan_run_f64<true>(C, b, (int)(hv[r] & 0xffffu), (int)(hv[r] >> 16), li, xi, yi, zi);
                        else mk[r] = scan_run_f64<false>(C, b, (int)(hv[r] & 0xffffu), (int)(hv[r] >> 16), li, xi, yi, zi);
                    }
                }
                int hits = 0;
#pragma unroll
                for (int r = 0; r < 9; ++r) hits += __builtin_popcount(mk[r]);
                const int id = __float_as_int(s.w);
                nn[id] = hits; // keeps counting past M (neighbor.cpp:172-177)
                if (COUNT) {
                    vmax = max(vmax, hits);
                } else {
                    // masks -> tickets in walk order: bit (len-1-j) of a run's mask is its candidate j
                    unsigned short *my = tk + (size_t)tid * M;
                    int sl = 0;
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        const int k0 = (int)(hv[r] & 0xffffu), L4 = ((int)(hv[r] >> 16) + 3) & ~3;
                        unsigned m = mk[r];
                        while (m) {
                            const int bpos = 31 - __builtin_clz(m);
                            if (sl < M) my[sl] = (unsigned short)(k0 + (L4 - 1 - bpos));
                            ++sl;
                            m &= ~(1u << bpos);
                        }
                    }
                    crow[tid] = id;
                    ccnt[tid] = hits < M ? hits : M;
                    cxw[tid] = xi; cyw[tid] = yi; czw[tid] = zi;
                }
            }
            if (!COUNT && dbg != 2) {
                __syncthreads();
                // ---- tickets -> rows: MP adjacent lanes serve the slots of one centre
                const int nrows = min(NT, ncentres - base);
                const int MP = 1 << mp_shift;
                const int e = tid & (MP - 1);
                if (e < M) {
                    for (int c = tid >> mp_shift; c < nrows; c += (NT >> mp_shift)) {
                        const int64_t o = (int64_t)crow[c] * M + e;
                        if (e < ccnt[c]) {
                            const int k = tk[c * M + e];
                            const double2 cj = lxy[k];
                            double d2;
                            if (C.general) d2 = exact_d2<true>(b, cj.x, cj.y, lz[k], cxw[c], cyw[c], czw[c], lsh[k]);
                            else d2 = exact_d2<false>(b, cj.x, cj.y, lz[k], cxw[c], cyw[c], czw[c], 0);
                            verlet[o] = __float_as_int(f4[k].w);
                            dist[o] = sqrt(d2); // neighbor.cpp:174
                        } else if (write_pads) {
                            verlet[o] = -1;
                            dist[o] = C.pad;
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (!list_mode)
            break;
        if (jt + (int)(gridDim.x >> 3) < per) __syncthreads(); // LDS is reused by the next tile
    } // tiles of this workgroup
    if (COUNT) {
        int m = vmax;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
        if (lane == 0 && m > 0) atomicMax(max_count, m);
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
// 3-cell z-runs — what a group of S lanes has to hold.  Counted on the device; the host uses the values the previous call
// with the same (N, grid) left in pinned memory — an MD-style sequence of calls never waits — and waits only the first
// time it sees a new (N, grid).  A stale value costs speed, never correctness.
// out[0] = occupied cells; out[1 + len] = number of runs of that length (len 0..64; out[66] = longer)
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
                        atomicAdd(&hist[1 + min(len, 65)], 1);
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

} // namespace wave

int grid_stats_hint(Scope &sc, const CellGrid &cg, int64_t N, GridStats *out)
{
    using namespace wave;
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
        MDH_HIP(hipHostMalloc(reinterpret_cast<void **>(&host), sizeof(int) * GridStats::NBIN, hipHostMallocDefault));
    }
    MDH_TRY(count(host));
    MDH_HIP(hipStreamSynchronize(st));
    read(host);
    g_stat.push_back(StatEntry{N, cg.g.ncell, device, host, 0u});
    return MDH_OK;
}

namespace wave {

static size_t lds_bytes(int cap, int64_t M, int NG)
{
    const size_t tk_bytes = (((size_t)64 * M + (size_t)NG * 64) * 2 + 7) & ~(size_t)7;
    const size_t wave_bytes = 3 * 64 * 8 + 2 * 64 * 4 + 64 * 2 + tk_bytes;
    return (size_t)(cap + 64) * 16 + (size_t)cap * 24 + 4 * wave_bytes + (size_t)cap * 2;
}

static size_t lds_bytes_lane(int cap, int64_t M)
{
    size_t tk = (size_t)NT * M;
    tk += tk & 1;
    return (size_t)(cap + 8) * 16 + (size_t)cap * 24 + (size_t)NT * 32 + (size_t)CEN_CAP * 4 + tk * 2 + (size_t)cap * 2;
}

static const int kS[6] = {7, 12, 16, 21, 32, 64};
static const int kNG[6] = {1, 2, 3, 3, 5, 9};

} // namespace wave

WavePlan plan_wave(const DBox &b, const Grid &g, int64_t N, int64_t M, const GridStats &gs, double rc)
{
    using namespace wave;
    WavePlan p{};
    if (b.tri || g.mode != 0 || N <= 0 || M <= 0 || M > 64)
        return p;
    for (int d = 0; d < 3; ++d)
        if (g.nc[d] < (b.pbc[d] ? 7 : 4)) // image numbers from the cell pair need >= 7 cells; skipping the far side of an open axis >= 4
            return p;
    if (!(rc > 1e-12 && rc < 1e12))
        return p;
    static const int cap_env = [] { const char *e = std::getenv("MDH_WAVE_CAP"); return e ? std::atoi(e) : 0; }();
    const int cap = cap_env > 0 ? cap_env : 768;
    const int64_t occ = gs.v[0] > 0 ? gs.v[0] : g.ncell;
    const double pop = (double)N / (double)occ; // mean atoms per cell of the occupied region
    Shape best{0, 0};
    double best_score = -1.0;
    for (int txy = 1; txy <= 8; ++txy)
        for (int tz = 1; tz <= 24; ++tz) {
            const int nh = (txy + 2) * (txy + 2) * (tz + 2);
            if (nh > MAX_NH || nh * pop > 0.86 * cap) // head-room for density fluctuations; overflowing tiles fall back
                continue;
            const int ncc = txy * txy * tz;
            if (ncc < 8) // two centre cells per wave at least
                continue;
            const double score = (double)ncc / (double)nh + 1e-4 * tz;
            if (score > best_score) { best_score = score; best = Shape{txy, tz}; }
        }
    if (!best.txy)
        return p;
    // lanes per run: the cheapest of the instantiated layouts.  A centre cell costs NG rounds per centre when its 9 runs
    // fit S lanes each, and one round per run and 64 candidates (>= 9) otherwise.
    int64_t runs = 0;
    for (int k = 1; k < GridStats::NBIN; ++k) runs += gs.v[k];
    int pick = -1;
    double best_cost = 1e300;
    static const int s_env = [] { const char *e = std::getenv("MDH_WAVE_S"); return e ? std::atoi(e) : 0; }();
    for (int c = 0; c < 6; ++c) {
        double p_run = 0.0; // fraction of runs longer than S
        if (runs > 0) {
            int64_t longer = 0;
            for (int len = kS[c] + 1; len <= 65; ++len) longer += gs.v[1 + len];
            p_run = (double)longer / (double)runs;
        } else {
            p_run = 3.0 * pop + 3.0 * std::sqrt(3.0 * pop) > kS[c] ? 0.5 : 0.0;
        }
        const double p_cell = 1.0 - std::pow(1.0 - p_run, 9.0);
        const double slow = 9.0 + 27.0 * pop / 64.0 + 6.0; // rounds + reloading the candidates per centre
        const double cost = (1.0 - p_cell) * kNG[c] + p_cell * slow;
        if (s_env > 0 ? kS[c] == s_env : cost < best_cost) { best_cost = cost; pick = c; if (s_env > 0) break; }
    }
    if (pick < 0)
        return p;
    if (wave::lds_bytes(cap, M, kNG[pick]) > 160 * 1024 - 4096)
        return p;
    // decision band of the single-precision scan (file header).  E bounds the staged coordinates (far-atom check of the
    // kernel), du the error of one staged coordinate: rounding to f32 plus what the double-precision shift can lose.
    const int hmax = std::max(best.txy, best.tz) + 2;
    const double E = ((double)hmax + 1.5) * rc;
    double big = E;
    for (int d = 0; d < 3; ++d) big = std::max(big, std::fabs(b.o[d]) + 2.0 * std::fabs(b.h[d * 4]) + E);
    const double du = std::ldexp(E, -24) * 1.01 + std::ldexp(big, -49);
    const double rcsq = rc * rc;
    const double tol = 2.0 * (11.0 * du * rc + 12.0 * std::ldexp(rcsq, -24));
    float lo = (float)(rcsq - tol), hi = (float)(rcsq + tol);
    while ((double)lo > rcsq - tol) lo = std::nextafterf(lo, -INFINITY);
    while ((double)hi < rcsq + tol) hi = std::nextafterf(hi, INFINITY);
    p.txy = best.txy;
    p.tz = best.tz;
    p.cap = cap;
    p.S = kS[pick];
    p.NG = kNG[pick];
    p.lo = lo;
    p.hi = hi;
    p.mid = (float)rcsq;
    {   // |fl32(d2 - mid) - (d2_exact - rc^2)| <= tol/2 + |mid - rc^2|: the same bound, centred
        float T = (float)(tol + std::fabs((double)p.mid - rcsq));
        while ((double)T < tol + std::fabs((double)p.mid - rcsq)) T = std::nextafterf(T, INFINITY);
        p.T = T;
    }
    p.occupied = occ;
    p.full = occ >= g.ncell;
    g_last_plan[0] = p.txy; g_last_plan[1] = p.tz; g_last_plan[2] = p.cap; g_last_plan[3] = p.S; g_last_plan[4] = p.NG;
    g_last_plan[5] = p.full; g_last_plan[6] = (int)(1000.0 * pop); g_last_plan[7] = (int)std::min<int64_t>(occ, 2147483647);
    return p;
}

namespace wave {

template <int S, int NG>
static void launch_sn(bool count, dim3 grid, size_t lds, hipStream_t st, const CellGrid &cg, const DBox &b, double rc, const WavePlan &p,
                      int *verlet, double *dist, int *nn, int M, int mp_shift, int write_pads, unsigned char *tile_flag, const int *nt,
                      Shape ts, const int *tile_list, const int *n_live, int list_mode, int *max_count, int *flagged)
{
    if (count)
        hipLaunchKernelGGL((k_neighbor_wave<S, NG, true>), grid, dim3(NT), lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.mvs, cg.cell_start, b, cg.g, rc, p.lo, p.hi, verlet, dist, nn, M, mp_shift, write_pads, p.cap, cg.flags, tile_flag, nt[0], nt[1], nt[2], ts, tile_list, n_live, list_mode, max_count, flagged);
    else
        hipLaunchKernelGGL((k_neighbor_wave<S, NG, false>), grid, dim3(NT), lds, st, cg.xs, cg.ys, cg.zs, cg.order, cg.mvs, cg.cell_start, b, cg.g, rc, p.lo, p.hi, verlet, dist, nn, M, mp_shift, write_pads, p.cap, cg.flags, tile_flag, nt[0], nt[1], nt[2], ts, tile_list, n_live, list_mode, max_count, flagged);
}

} // namespace wave

// count == true: nn and *max_count only (first pass of the exact-width variant); M is then 1
int launch_neighbor_wave(Scope &sc, const CellGrid &cg, const WavePlan &plan, int64_t N, const DBox &b, double rc,
                         int *verlet, double *dist, int *nn, int64_t M, bool fill_pads, bool count, int *max_count,
                         TileFilter &tf)
{
    using namespace wave;
    const Shape ts{plan.txy, plan.tz};
    int nt[3];
    for (int d = 0; d < 3; ++d) {
        const int T = d == 2 ? ts.tz : ts.txy;
        nt[d] = (cg.g.nc[d] + T - 1) / T;
    }
    const int64_t ntiles = (int64_t)nt[0] * nt[1] * nt[2];
    unsigned char *tile_flag = sc.alloc_n<unsigned char>((size_t)ntiles);
    unsigned *live = sc.alloc_n<unsigned>((size_t)ntiles);
    int *slot = sc.alloc_n<int>((size_t)ntiles + 1);
    int *tile_list = sc.alloc_n<int>((size_t)ntiles);
    int *flagged = sc.alloc_n<int>((size_t)ntiles); // ids of the tiles left to the thread-per-atom code (every tile is flagged at most once)
    if (sc.failed())
        return sc.error();
    hipStream_t st = sc.stream();
    MDH_HIP(hipMemsetAsync(tile_flag, 0, (size_t)ntiles, st));
    int per = (int)((ntiles + 7) / 8);
    int list_mode = 0;
    if (plan.full) { // the statistics say that no 4x4x4 block of cells is empty: all tiles are live, workgroup b owns tile b
        tile_list = nullptr;
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
    const size_t lds = lds_bytes(plan.cap, count ? 1 : M, plan.NG);
    int mp_shift = 0;
    while ((1 << mp_shift) < M) ++mp_shift;
    const int Mi = (int)M, wp = fill_pads ? 1 : 0;
    static const int kernel_env = [] { const char *e = std::getenv("MDH_NB_KERNEL"); return e ? std::atoi(e) : 0; }();
    static const int dbg_env = [] { const char *e = std::getenv("MDH_DBG_PHASE"); return e ? std::atoi(e) : 0; }();
    if (kernel_env == 1) { // thread-per-centre variant on the same staging
        const size_t ldsl = lds_bytes_lane(plan.cap, count ? 1 : M);
        const float lane_lo = -plan.mid, lane_hi = plan.T; // k_neighbor_lane: -(rc^2) and the half-width of the decision band
        // second pass: the listed tiles again, cut into one-cell slices along z (a fifth of the halo); what overflows even
        // then (a dense blob) is listed once more, for the thread-per-atom code
        const int nsub = ts.tz;
        const Shape ts2{ts.txy, 1};
        const int nt2b = nt[2] * nsub;
        int *flagged2 = sc.alloc_n<int>((size_t)ntiles * (size_t)nsub);
        if (sc.failed())
            return sc.error();
        const void *fn = count ? reinterpret_cast<const void *>(&k_neighbor_lane<true>) : reinterpret_cast<const void *>(&k_neighbor_lane<false>);
        if (ldsl > 60 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsl);
        if (count) {
            hipLaunchKernelGGL((k_neighbor_lane<true>), grid, dim3(NT), ldsl, st, cg.xs, cg.ys, cg.zs, cg.order, cg.mvs, cg.cell_start, b, cg.g, rc, lane_lo, lane_hi, verlet, dist, nn, Mi, mp_shift, wp, plan.cap, cg.flags, tile_flag, nt[0], nt[1], nt[2], ts, tile_list, slot + ntiles, list_mode, max_count, flagged, dbg_env, nullptr, 0, 1, 2);
            hipLaunchKernelGGL((k_neighbor_lane<true>), dim3(1024), dim3(NT), ldsl, st, cg.xs, cg.ys, cg.zs, cg.order, cg.mvs, cg.cell_start, b, cg.g, rc, lane_lo, lane_hi, verlet, dist, nn, Mi, mp_shift, wp, plan.cap, cg.flags, nullptr, nt[0], nt[1], nt2b, ts2, nullptr, cg.flags + 2, 1, max_count, flagged2, dbg_env, flagged, nt[2], nsub, 3);
        } else {
            hipLaunchKernelGGL((k_neighbor_lane<false>), grid, dim3(NT), ldsl, st, cg.xs, cg.ys, cg.zs, cg.order, cg.mvs, cg.cell_start, b, cg.g, rc, lane_lo, lane_hi, verlet, dist, nn, Mi, mp_shift, wp, plan.cap, cg.flags, tile_flag, nt[0], nt[1], nt[2], ts, tile_list, slot + ntiles, list_mode, max_count, flagged, dbg_env, nullptr, 0, 1, 2);
            hipLaunchKernelGGL((k_neighbor_lane<false>), dim3(1024), dim3(NT), ldsl, st, cg.xs, cg.ys, cg.zs, cg.order, cg.mvs, cg.cell_start, b, cg.g, rc, lane_lo, lane_hi, verlet, dist, nn, Mi, mp_shift, wp, plan.cap, cg.flags, nullptr, nt[0], nt[1], nt2b, ts2, nullptr, cg.flags + 2, 1, max_count, flagged2, dbg_env, flagged, nt[2], nsub, 3);
        }
        MDH_HIP(hipGetLastError());
        tf.flag = tile_flag; tf.any = cg.flags + 3; tf.moved = cg.flags; tf.list = flagged2;
        tf.list_cap = (int)std::min<int64_t>(ntiles * nsub, 2147483647);
        tf.tile = ts2.txy; tf.tile_z = ts2.tz; tf.nt[0] = nt[0]; tf.nt[1] = nt[1]; tf.nt[2] = nt2b;
        return MDH_OK;
    }
    const bool big_lds = lds > 60 * 1024; // above the default dynamic-LDS limit: raise it for the instance about to run
#define MDH_WAVE_CASE(S_, NG_)                                                                                                  \
    case S_: {                                                                                                                  \
        if (big_lds) {                                                                                                          \
            if (count) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_neighbor_wave<S_, NG_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            else (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_neighbor_wave<S_, NG_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
        }                                                                                                                       \
        launch_sn<S_, NG_>(count, grid, lds, st, cg, b, rc, plan, verlet, dist, nn, Mi, mp_shift, wp, tile_flag, nt, ts, tile_list, slot + ntiles, list_mode, max_count, flagged); \
        break;                                                                                                                  \
    }
    switch (plan.S) {
        MDH_WAVE_CASE(7, 1)
        MDH_WAVE_CASE(12, 2)
        MDH_WAVE_CASE(16, 3)
        MDH_WAVE_CASE(21, 3)
        MDH_WAVE_CASE(32, 5)
        MDH_WAVE_CASE(64, 9)
    default:
        set_error("internal: no wave kernel for this lane layout");
        return MDH_ERR_ARG;
    }
#undef MDH_WAVE_CASE
    MDH_HIP(hipGetLastError());
    tf.flag = tile_flag;
    tf.any = cg.flags + 2;
    tf.moved = cg.flags;
    tf.list = flagged;
    tf.list_cap = (int)std::min<int64_t>(ntiles, 2147483647);
    tf.tile = ts.txy;
    tf.tile_z = ts.tz;
    tf.nt[0] = nt[0]; tf.nt[1] = nt[1]; tf.nt[2] = nt[2];
    return MDH_OK;
}

} // namespace mdh

extern "C" int mdh_debug_wave_info(int *plan8, int *dbg4)
{
    for (int k = 0; k < 8; ++k) plan8[k] = mdh::wave::g_last_plan[k];
    MDH_HIP(hipMemcpyFromSymbol(dbg4, HIP_SYMBOL(mdh::wave::g_dbg), sizeof(int) * 4));
    int zero[4] = {0, 0, 0, 0};
    MDH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(mdh::wave::g_dbg), zero, sizeof(int) * 4));
    return MDH_OK;
}
