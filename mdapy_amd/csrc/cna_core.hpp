// cna_core.hpp — the register-level pieces of the common neighbour analysis shared by cna.hip (one thread per atom, the
// listed neighbours gathered from global memory) and neighbor_lane.hip (fixed-cutoff CNA fused into the neighbour tile
// kernel, the neighbours' positions read from the LDS tile): bond matrix rows, the three numbers of a CNA signature, the
// fixed-cutoff label.  Reference: src/cna.cpp:16-161 (helpers), :429-506 (FixedCNA).
#pragma once
#include "common.hpp"

namespace mdh {

struct Rows { // bond matrix: row a = bits of the neighbours bonded to neighbour a (a < 16)
    uint64_t w0, w1, w2, w3;
    __device__ __forceinline__ unsigned row(int a) const
    {
        const uint64_t w = (a < 8) ? ((a < 4) ? w0 : w1) : ((a < 12) ? w2 : w3);
        return (unsigned)(w >> ((a & 3) << 4)) & 0xffffu;
    }
};

template <int NN>
__device__ __forceinline__ Rows pack_rows(const unsigned (&adj)[NN])
{
    Rows r{0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < NN; ++a) {
        const uint64_t v = (uint64_t)adj[a] << ((a & 3) << 4);
        if (a < 4) r.w0 |= v;
        else if (a < 8) r.w1 |= v;
        else if (a < 12) r.w2 |= v;
        else r.w3 |= v;
    }
    return r;
}

// The same rows in LDS, one column per thread (row a of thread t at base[a * stride]): a row fetched by a dynamic index is one
// ds_read_u16 instead of the ~8 select / shift instructions the packed registers need — the signatures of an atom fetch
// ~60 rows, and the kernels that use this keep no other data in LDS.
struct RowsLds {
    const unsigned short *base;
    int stride;
    __device__ __forceinline__ unsigned row(int a) const { return base[a * stride]; }
};
template <int NN>
__device__ __forceinline__ RowsLds spill_rows(const Rows &R, unsigned short *base, int stride)
{
#pragma unroll
    for (int a = 0; a < NN; ++a) base[a * stride] = (unsigned short)R.row(a); // (static a: a shift and a mask)
    return RowsLds{base, stride};
}

// The rows where they are — NN registers — for a caller with no LDS to spare (the fused kernel of neighbor_lane.hip): a row fetched
// by a computed index is a chain of NN - 1 selects; only the cluster walks of signature() fetch rows that way (six bonds on six
// atoms, the (6,6,6) of bcc), cna_counts_words() decides everything else from the packed words.
template <int NN>
struct RowsReg {
    unsigned v[NN]; // (a copy, every index a constant: stays in registers)
    __device__ __forceinline__ explicit RowsReg(const unsigned (&adj)[NN])
    {
#pragma unroll
        for (int k = 0; k < NN; ++k) v[k] = adj[k];
    }
    __device__ __forceinline__ unsigned row(int a) const
    {
        unsigned r = v[0];
#pragma unroll
        for (int k = 1; k < NN; ++k) {
            unsigned t = v[k];
            asm("" : "+v"(t)); // (opaque: left to itself the compiler turns the chain into an indexed array in scratch memory)
            r = (a == k) ? t : r;
        }
        return r;
    }
};

// (ncn, nb, chain) of the bond centre--neighbour ni.  Only neighbours in `limit_mask`
// take part in the bond search (cna.cpp:69-92; the adaptive 12-neighbour pass hands 12, :344).
template <class RT>
__device__ __forceinline__ void signature(const RT &R, int ni, unsigned limit_mask, int &ncn, int &nb, int &chain)
{
    const unsigned common = R.row(ni); // cna.cpp:52-64
    ncn = __popc(common);
    const unsigned pool = common & limit_mask;
    int bonds = 0, maxdeg = 0;
    for (unsigned m = pool; m; m &= m - 1) {
        const int deg = __popc(R.row(__ffs(m) - 1) & pool);
        bonds += deg;
        maxdeg = deg > maxdeg ? deg : maxdeg;
    }
    nb = bonds >> 1;
    // `chain` = number of bonds in the largest connected bond cluster (cna.cpp:97-147).  Shortcuts that are
    // exact consequences of that definition:
    //   nb <= 1           -> chain = nb
    //   nb == 2           -> 2 if the two bonds share an atom (some degree is 2), else 1
    //   k atoms, k bonds, k in {4,5} and every atom of the pool counted (ncn == popc(pool)):
    //                        a graph on k <= 5 vertices with k edges cannot be split (3+1 vertices hold <= 3 edges,
    //                        3+2 hold <= 4, 4+1 hold all of them in the connected 4-part) -> chain = k
    // everything else (e.g. 6 atoms / 6 bonds: two triangles give 3) walks the clusters.
    if (nb <= 1) { chain = nb; return; }
    if (nb == 2) { chain = maxdeg == 2 ? 2 : 1; return; }
    const int npool = __popc(pool);
    if (nb == npool && (npool == 4 || npool == 5)) { chain = nb; return; }
    int best = 0;
    unsigned left = pool;
    while (left) {
        unsigned comp = left & (0u - left), frontier = comp;
        while (frontier) {
            const int a = __ffs(frontier) - 1;
            frontier &= frontier - 1;
            const unsigned grow = R.row(a) & pool & ~comp;
            comp |= grow;
            frontier |= grow;
        }
        int cb = 0;
        for (unsigned m = comp; m; m &= m - 1)
            cb += __popc(R.row(__ffs(m) - 1) & comp);
        cb >>= 1;
        best = cb > best ? cb : best;
        left &= ~comp;
    }
    chain = best;
}

__device__ __forceinline__ double fold(double d, double L, double t_zero, double t_one)
{
    const double shift = (d >= t_one) ? L : ((d >= t_zero) ? 0.0 : -L);
    return d - shift; // == d - L*floor(d/L+0.5)   (box.h:120-124) when n is one of {-1,0,1}
}

// 0: the coordinates spread too far for `fold` (generic variant), 1: every pair has n in {-1,0,1}, 2: every pair has n = 0
// (no neighbour lies across the periodic seam: the minimum image is the plain difference, d - L*0 == d)
template <int NN>
__device__ __forceinline__ int span_class(const DBox &b, const double (&p)[NN], int axis)
{
    double mn = p[0], mx = p[0];
#pragma unroll
    for (int a = 1; a < NN; ++a) {
        mn = fmin(mn, p[a]);
        mx = fmax(mx, p[a]);
    }
    const double span = mx - mn;
    const double lim = fmin(b.tn[axis][3], -b.tn[axis][0]); // |d| below this => n in {-1,0,1}
    const double zero = fmin(b.tn[axis][2], -b.tn[axis][1]); // |d| below this => n == 0
    if (!(span < lim)) // false for NaN as well
        return 0;
    return span < zero ? 2 : 1;
}

// bond matrix among NN neighbours whose RAW positions are in registers: bit c of row a <=> pbcdis_sq(a, c) <= cut2
// (cna.cpp:459-466, both ends raw coordinates cna.cpp:149-161), the reference expression pair by pair
template <bool TRI, int NN>
__device__ __forceinline__ Rows bond_rows_reg(const DBox &b, const double (&px)[NN], const double (&py)[NN],
                                              const double (&pz)[NN], double cut2)
{
    unsigned adj[NN];
#pragma unroll
    for (int a = 0; a < NN; ++a)
        adj[a] = 0;
#pragma unroll
    for (int a = 0; a < NN; ++a)
#pragma unroll
        for (int c = a + 1; c < NN; ++c) {
            const double d2 = pair_d2<TRI>(b, px[a], py[a], pz[a], px[c], py[c], pz[c]);
            if (d2 <= cut2) {
                adj[a] |= 1u << c;
                adj[c] |= 1u << a;
            }
        }
    return pack_rows<NN>(adj);
}

// orthogonal box, positions classified by span_class3(): plain (class 2: every pair has image number 0, the minimum image
// is the plain difference, 11 instructions per pair) or class 1 (the shift L*n picked by the two exact thresholds of
// DBox::tn).  One function with one `adj` and an early return on purpose: written as two functions behind a branch the
// compiler hoists the pairs' common subtractions above it and needs four times the registers.
template <int NN>
__device__ __forceinline__ Rows bond_rows_ortho(const DBox &b, const double (&px)[NN], const double (&py)[NN],
                                                const double (&pz)[NN], double cut2, bool plain)
{
    unsigned adj[NN];
#pragma unroll
    for (int a = 0; a < NN; ++a)
        adj[a] = 0;
    if (plain) {
#pragma unroll
        for (int a = 0; a < NN; ++a)
#pragma unroll
            for (int c = a + 1; c < NN; ++c) {
                const double dx = px[c] - px[a], dy = py[c] - py[a], dz = pz[c] - pz[a];
                if (dx * dx + dy * dy + dz * dz <= cut2) {
                    adj[a] |= 1u << c;
                    adj[c] |= 1u << a;
                }
            }
        return pack_rows<NN>(adj);
    }
#pragma unroll
    for (int a = 0; a < NN; ++a)
#pragma unroll
        for (int c = a + 1; c < NN; ++c) {
            double dx = px[c] - px[a], dy = py[c] - py[a], dz = pz[c] - pz[a];
            if (b.pbc[0]) dx = fold(dx, b.h[0], b.tn[0][1], b.tn[0][2]);
            if (b.pbc[1]) dy = fold(dy, b.h[4], b.tn[1][1], b.tn[1][2]);
            if (b.pbc[2]) dz = fold(dz, b.h[8], b.tn[2][1], b.tn[2][2]);
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 <= cut2) {
                adj[a] |= 1u << c;
                adj[c] |= 1u << a;
            }
        }
    return pack_rows<NN>(adj);
}

// how the NN positions spread over the periodic axes of an orthogonal box (the smallest span_class() of the three)
template <int NN>
__device__ __forceinline__ int span_class3(const DBox &b, const double (&px)[NN], const double (&py)[NN], const double (&pz)[NN])
{
    int cls = 2;
    if (b.pbc[0]) cls = min(cls, span_class<NN>(b, px, 0));
    if (b.pbc[1]) cls = min(cls, span_class<NN>(b, py, 1));
    if (b.pbc[2]) cls = min(cls, span_class<NN>(b, pz, 2));
    return cls;
}

// fixed-cutoff label from the bond matrix of the NN listed neighbours (cna.cpp:471-503; no early exit): 0 = none
template <int NN, class RT>
__device__ __forceinline__ int fcna_label(const RT &R)
{
    int n421 = 0, n422 = 0, n555 = 0, n444 = 0, n666 = 0;
    for (int ni = 0; ni < NN; ++ni) {
        int ncn, nb, ch;
        signature(R, ni, (1u << NN) - 1u, ncn, nb, ch);
        if (ncn == 4 && nb == 2) { n421 += (ch == 1); n422 += (ch == 2); }
        else if (ncn == 5 && nb == 5 && ch == 5) ++n555;
        else if (ncn == 4 && nb == 4 && ch == 4) ++n444;
        else if (ncn == 6 && nb == 6 && ch == 6) ++n666;
    }
    if (n421 == 12) return 1; // cna.cpp:496-503
    if (n421 == 6 && n422 == 6) return 2;
    if (n555 == 12) return 4;
    if (n666 == 8 && n444 == 6) return 3;
    return 0;
}

// The same label from the bond rows held in registers, two 16-bit rows to a word, without fetching a row by a computed index:
// for the bond centre--ni the rows of its common neighbours are selected by masks, so 2 * (bonds among them) is one popcount
// per word and the atoms those bonds touch are the OR of the masked words.  That decides every signature with at most two
// bonds (two bonds share an atom <=> they touch three atoms) and those of the shortcuts of signature(); the rest — six bonds
// on six atoms, the (6,6,6) of bcc — walks the clusters through the LDS copy of the rows.
// six bonds among six common neighbours (the (6,6,x) of bcc): are they ONE cluster (x = 6)?  The cluster of the lowest atom that has
// a bond is grown five times — a cluster on six atoms has no longer path — through the packed rows, the rows of the atoms reached so
// far selected by masks as in cna_counts_words: no row is fetched by a computed index, so no caller needs its rows in LDS (the tile
// kernel has none to spare: DESIGN 3a), and a lane that meets no such signature never comes here.
template <int NW>
__device__ __forceinline__ bool six_bonds_one_cluster(const unsigned (&P)[NW], unsigned common, unsigned bonded)
{
    unsigned reach = bonded & (0u - bonded); // bonded: the common neighbours that have a bond (16 bits)
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        unsigned acc = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const unsigned lo = (unsigned)(((int)(reach << (31 - 2 * k))) >> 31) & 0xffffu;
            const unsigned hi = (unsigned)(((int)(reach << (30 - 2 * k))) >> 31) & 0xffff0000u;
            acc |= P[k] & (lo | hi);
        }
        reach |= (acc | (acc >> 16)) & common & 0xffffu;
    }
    return (bonded & ~reach) == 0u;
}
struct CnaCounts { int n421, n422, n555, n444, n666; };
// MAXO: signatures of none of the five kinds after which no label is possible any more — every label asks for 12 (14) bonds of
// the listed kinds: one stray bond decides a 12-neighbour atom, three a 14-neighbour one (n421 == 12 needs 12 of the 14).  A lane
// past that number skips the rest of its bonds; the counts it returns are then incomplete, and every caller's label tests fail on
// them as they would on the complete ones.  On a lattice nothing is skipped; in a liquid nearly every atom is decided by its
// first bonds and a wave leaves once its last lane is (the general cluster walks below were most of the kernel's time there).
template <int NN, int MAXO = NN, class RT = RowsLds>
__device__ __forceinline__ CnaCounts cna_counts_words(const unsigned (&adj)[NN], const RT &L)
{
    constexpr int NW = (NN + 1) / 2;
    unsigned P[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) P[k] = adj[2 * k] | ((2 * k + 1 < NN) ? (adj[2 * k + 1] << 16) : 0u);
    // the five counts and `others` as 4-bit fields of one word (each <= NN <= 14): a signature adds ONE constant picked by
    // key = ncn + 32 nb — (4,2,x) 68, (5,5,5) 165, (4,4,4) 132, (6,6,6) 198 — instead of a chain of compared-and-branched
    // increments (thirty instructions per bond, half of this function, before)
    unsigned packed = 0;
#pragma unroll
    for (int ni = 0; ni < NN; ++ni) {
        if (MAXO < NN && (packed >> 20) > (unsigned)MAXO)
            continue;
        const unsigned common = adj[ni]; // cna.cpp:52-64
        const unsigned ncn = (unsigned)__popc(common);
        const unsigned c2 = common | (common << 16);
        unsigned twice = 0, touched = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const unsigned lo = (unsigned)(((int)(common << (31 - 2 * k))) >> 31) & 0xffffu;         // row 2k is a common neighbour
            const unsigned hi = (unsigned)(((int)(common << (30 - 2 * k))) >> 31) & 0xffff0000u;     // row 2k+1
            const unsigned xk = P[k] & c2 & (lo | hi);
            twice += __popc(xk);
            touched |= xk;
        }
        // chain (cna.cpp:97-147) by the shortcuts of signature(): two bonds make a chain of 2 iff they share an atom (three atoms
        // touched); k bonds on k = 4 or 5 atoms are one cluster.  Only (6,6,x) — bcc — has to look at the clusters, and only its
        // answer is used: every other signature with three bonds or more counts as `other` whatever its chain is.  (L, the rows
        // for a fetch by a computed index, is not used any more: kept for the callers' sake.)
        const unsigned key = ncn + (twice << 4); // (twice = 2 nb: the rows are symmetric; ncn < 16: no two signatures share a key)
        const unsigned three = (__popc((touched | (touched >> 16)) & 0xffffu) == 3) ? 1u << 4 : 1u;
        unsigned inc = 1u << 20;
        inc = key == 68u ? three : inc;
        inc = key == 165u ? 1u << 8 : inc;
        inc = key == 132u ? 1u << 12 : inc;
        if (key == 198u) // (6,6,x): x = 6 iff the six bonds are one cluster (cna.cpp:97-147)
            inc = six_bonds_one_cluster<NW>(P, common, (touched | (touched >> 16)) & 0xffffu) ? 1u << 16 : 1u << 20;
        packed += inc;
    }
    return CnaCounts{(int)(packed & 15u), (int)((packed >> 4) & 15u), (int)((packed >> 8) & 15u), (int)((packed >> 12) & 15u), (int)((packed >> 16) & 15u)};
}
template <int NN, class RT = RowsLds>
__device__ __forceinline__ int fcna_label_words(const unsigned (&adj)[NN], const RT &L)
{
    const CnaCounts c = cna_counts_words<NN, NN - 12, RT>(adj, L); // (12 neighbours: no stray bond; 14: two, n421 == 12 of 14)
    if (c.n421 == 12) return 1; // cna.cpp:496-503
    if (c.n421 == 6 && c.n422 == 6) return 2;
    if (c.n555 == 12) return 4;
    if (c.n666 == 8 && c.n444 == 6) return 3;
    return 0;
}

// pair number p -> (a, c), a < c, in the order (0,1) (0,2) ... (NN-2,NN-1)
template <int NN> constexpr int pair_a(int p) { int a = 0; while (p >= NN - 1 - a) { p -= NN - 1 - a; ++a; } return a; }
template <int NN> constexpr int pair_c(int p) { int a = 0; while (p >= NN - 1 - a) { p -= NN - 1 - a; ++a; } return a + 1 + p; }
// the NN (NN - 1) / 2 single-precision pair tests of the CNA kernels (the first NN of NV vectors), two pairs to a block: e = |u_c - u_a|^2 - c for both, ONE v_min3_u32
// that tracks the smallest non-negative e seen, the sign of each e shifted into both bond rows — 9.5 register-only instructions per pair
template <int NN, int NV, int P>
__device__ __forceinline__ void pair_tests_f32(const float (&ux)[NV], const float (&uy)[NV], const float (&uz)[NV], float negc,
                                               unsigned (&adj)[NN], unsigned &w)
{
    constexpr int NP = NN * (NN - 1) / 2;
    if constexpr (P + 1 < NP) {
        constexpr int a0 = pair_a<NN>(P), c0 = pair_c<NN>(P), a1 = pair_a<NN>(P + 1), c1 = pair_c<NN>(P + 1);
        float t0, t1, t2, s0, s1, s2;
        // (a row that both pairs touch — (a,c) (a,c+1), or (NN-3,NN-1) (NN-2,NN-1) — must be ONE operand of the block: bound twice it
        // would be two registers, and one of the two bits would be lost)
#define MDH_PAIR2_HEAD                                                                                                                 \
            "v_sub_f32 %[t0], %[xc], %[xa]\n\t"                                                                                        \
            "v_sub_f32 %[t1], %[yc], %[ya]\n\t"                                                                                        \
            "v_sub_f32 %[t2], %[zc], %[za]\n\t"                                                                                        \
            "v_sub_f32 %[s0], %[xd], %[xb]\n\t"                                                                                        \
            "v_sub_f32 %[s1], %[yd], %[yb]\n\t"                                                                                        \
            "v_sub_f32 %[s2], %[zd], %[zb]\n\t"                                                                                        \
            "v_fma_f32 %[t0], %[t0], %[t0], %[negc]\n\t"                                                                               \
            "v_fma_f32 %[s0], %[s0], %[s0], %[negc]\n\t"                                                                               \
            "v_fmac_f32 %[t0], %[t1], %[t1]\n\t"                                                                                       \
            "v_fmac_f32 %[s0], %[s1], %[s1]\n\t"                                                                                       \
            "v_fmac_f32 %[t0], %[t2], %[t2]\n\t"                                                                                       \
            "v_fmac_f32 %[s0], %[s2], %[s2]\n\t"                                                                                       \
            "v_min3_u32 %[w], %[w], %[t0], %[s0]\n\t"                                                                                  \
            "v_lshrrev_b32 %[t0], 31, %[t0]\n\t"                                                                                       \
            "v_lshrrev_b32 %[s0], 31, %[s0]\n\t"
#define MDH_PAIR2_IN                                                                                                                   \
            [xc] "v"(ux[c0]), [xa] "v"(ux[a0]), [yc] "v"(uy[c0]), [ya] "v"(uy[a0]), [zc] "v"(uz[c0]), [za] "v"(uz[a0]),                 \
            [xd] "v"(ux[c1]), [xb] "v"(ux[a1]), [yd] "v"(uy[c1]), [yb] "v"(uy[a1]), [zd] "v"(uz[c1]), [zb] "v"(uz[a1]),                 \
            [negc] "v"(negc), [sc] "n"(c0), [sa] "n"(a0), [sd] "n"(c1), [sb] "n"(a1)
        if constexpr (a0 == a1) {
            asm(MDH_PAIR2_HEAD
                "v_lshl_or_b32 %[ra], %[t0], %[sc], %[ra]\n\t"
                "v_lshl_or_b32 %[rc], %[t0], %[sa], %[rc]\n\t"
                "v_lshl_or_b32 %[ra], %[s0], %[sd], %[ra]\n\t"
                "v_lshl_or_b32 %[rd], %[s0], %[sb], %[rd]"
                : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2), [w] "+v"(w),
                  [ra] "+v"(adj[a0]), [rc] "+v"(adj[c0]), [rd] "+v"(adj[c1])
                : MDH_PAIR2_IN);
        } else if constexpr (c0 == c1) {
            asm(MDH_PAIR2_HEAD
                "v_lshl_or_b32 %[ra], %[t0], %[sc], %[ra]\n\t"
                "v_lshl_or_b32 %[rc], %[t0], %[sa], %[rc]\n\t"
                "v_lshl_or_b32 %[rb], %[s0], %[sd], %[rb]\n\t"
                "v_lshl_or_b32 %[rc], %[s0], %[sb], %[rc]"
                : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2), [w] "+v"(w),
                  [ra] "+v"(adj[a0]), [rc] "+v"(adj[c0]), [rb] "+v"(adj[a1])
                : MDH_PAIR2_IN);
        } else {
            static_assert(a0 != c1 && c0 != a1, "two pairs of a block share a row that the block binds twice");
            asm(MDH_PAIR2_HEAD
                "v_lshl_or_b32 %[ra], %[t0], %[sc], %[ra]\n\t"
                "v_lshl_or_b32 %[rc], %[t0], %[sa], %[rc]\n\t"
                "v_lshl_or_b32 %[rb], %[s0], %[sd], %[rb]\n\t"
                "v_lshl_or_b32 %[rd], %[s0], %[sb], %[rd]"
                : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2), [w] "+v"(w),
                  [ra] "+v"(adj[a0]), [rc] "+v"(adj[c0]), [rb] "+v"(adj[a1]), [rd] "+v"(adj[c1])
                : MDH_PAIR2_IN);
        }
#undef MDH_PAIR2_HEAD
#undef MDH_PAIR2_IN
        pair_tests_f32<NN, NV, P + 2>(ux, uy, uz, negc, adj, w);
    } else if constexpr (P < NP) {
        constexpr int a = pair_a<NN>(P), c = pair_c<NN>(P);
        float t0, t1, t2;
        asm("v_sub_f32 %[t0], %[xc], %[xa]\n\t"
            "v_sub_f32 %[t1], %[yc], %[ya]\n\t"
            "v_sub_f32 %[t2], %[zc], %[za]\n\t"
            "v_fma_f32 %[t0], %[t0], %[t0], %[negc]\n\t"
            "v_fmac_f32 %[t0], %[t1], %[t1]\n\t"
            "v_fmac_f32 %[t0], %[t2], %[t2]\n\t"
            "v_min_u32 %[w], %[w], %[t0]\n\t"
            "v_lshrrev_b32 %[t0], 31, %[t0]\n\t"
            "v_lshl_or_b32 %[ra], %[t0], %[sc], %[ra]\n\t"
            "v_lshl_or_b32 %[rc], %[t0], %[sa], %[rc]"
            : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [w] "+v"(w), [ra] "+v"(adj[a]), [rc] "+v"(adj[c])
            : [xc] "v"(ux[c]), [xa] "v"(ux[a]), [yc] "v"(uy[c]), [ya] "v"(uy[a]), [zc] "v"(uz[c]), [za] "v"(uz[a]), [negc] "v"(negc),
              [sc] "n"(c), [sa] "n"(a));
    }
}

// to-do list of atoms left to a later kernel: todo[0] = count, todo[1..] = atom ids
__device__ __forceinline__ void defer(int *__restrict__ todo, int64_t i) { todo[1 + atomicAdd(&todo[0], 1)] = (int)i; }

} // namespace mdh
