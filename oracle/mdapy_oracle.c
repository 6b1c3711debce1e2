/*
 * mdapy_oracle.c — TEST INFRASTRUCTURE ONLY.  Not shipped, not on the product path.
 *
 * A plain-C (C99 + optional OpenMP) restatement of the algorithms of mdapy's
 * neighbor-list + per-atom structural-analysis hot path.  It exists so that
 * the HIP kernels in mdapy_amd/csrc can be checked for parity on a machine
 * where the reference itself is not present.  Only tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py may load this library.
 *
 * Every function cites the reference file:line whose arithmetic (operation
 * order, comparison operators, sentinel conventions) it follows; paths are
 * relative to the reference checkout (mushroomfire/mdapy 1.0.8a1).
 *
 * Build: see oracle/Makefile  (gcc -O3 -std=c99 -fopenmp -ffp-contract=off, no -march:
 * the reference is built without FMA, CMakeLists.txt:126-133).
 *
 * Parity pinning: tests/test_oracle_golden.py checks this file against the
 * reference's own golden vectors (tests/golden/, copied data files of the
 * reference test-suite) — see DESIGN.md §"Oracle".
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Box  (src/box.h:8-244)                                                     */
/* ------------------------------------------------------------------------- */
typedef struct {
    double h[9];      /* rows a,b,c                                   box.h:12 */
    double hi[9];     /* inverse                                      box.h:13 */
    double o[3];
    double thick[3];
    int pbc[3];
    int tri;
} obox;

static double obox_volume(const obox *b) /* box.h:22-35 */
{
    const double *d = b->h;
    if (b->tri)
        return d[0] * (d[4] * d[8] - d[5] * d[7]) - d[1] * (d[3] * d[8] - d[5] * d[6]) +
               d[2] * (d[3] * d[7] - d[4] * d[6]);
    return d[0] * d[4] * d[8];
}

static double obox_thickness(const obox *b, int dir) /* box.h:54-89 */
{
    if (!b->tri)
        return b->h[dir * 4];
    double V = obox_volume(b);
    const double *A = b->h, *B = b->h + 3, *C = b->h + 6;
    const double *p, *q;
    if (dir == 0) { p = B; q = C; }
    else if (dir == 1) { p = A; q = C; }
    else { p = A; q = B; }
    double m = p[1] * q[2] - p[2] * q[1];
    double n = p[2] * q[0] - p[0] * q[2];
    double k = p[0] * q[1] - p[1] * q[0];
    return V / sqrt(m * m + n * n + k * k);
}

/* returns 0 ok, -1 singular triclinic box (box.h:185-186 throws) */
static int obox_init(obox *b, const double *box9, const double *origin, const int *boundary) /* box.h:208-244 */
{
    memset(b, 0, sizeof(*b));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            b->h[i * 3 + j] = box9[i * 3 + j];
            if (i != j && fabs(box9[i * 3 + j]) > 1e-10)
                b->tri = 1;
        }
    if (b->h[0] < 0 || b->h[4] < 0 || b->h[8] < 0)
        b->tri = 1;
    if (b->tri) { /* box.h:182-203 adjugate * 1/det */
        double det = obox_volume(b);
        if (fabs(det) < 1e-12)
            return -1;
        double id = 1.0 / det;
        const double *m = b->h;
        b->hi[0] = (m[4] * m[8] - m[5] * m[7]) * id;
        b->hi[1] = -(m[1] * m[8] - m[2] * m[7]) * id;
        b->hi[2] = (m[1] * m[5] - m[2] * m[4]) * id;
        b->hi[3] = -(m[3] * m[8] - m[5] * m[6]) * id;
        b->hi[4] = (m[0] * m[8] - m[2] * m[6]) * id;
        b->hi[5] = -(m[0] * m[5] - m[2] * m[3]) * id;
        b->hi[6] = (m[3] * m[7] - m[4] * m[6]) * id;
        b->hi[7] = -(m[0] * m[7] - m[1] * m[6]) * id;
        b->hi[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    } else {
        b->hi[0] = 1.0 / b->h[0];
        b->hi[4] = 1.0 / b->h[4];
        b->hi[8] = 1.0 / b->h[8];
    }
    for (int i = 0; i < 3; ++i) {
        b->o[i] = origin[i];
        b->pbc[i] = boundary[i];
    }
    for (int i = 0; i < 3; ++i)
        b->thick[i] = obox_thickness(b, i);
    return 0;
}

/* minimum image, box.h:94-126 */
static inline void obox_pbc(const obox *b, double *dx, double *dy, double *dz)
{
    if (b->tri) {
        const double *hi = b->hi, *h = b->h;
        double fx = *dx * hi[0] + *dy * hi[3] + *dz * hi[6];
        double fy = *dx * hi[1] + *dy * hi[4] + *dz * hi[7];
        double fz = *dx * hi[2] + *dy * hi[5] + *dz * hi[8];
        if (b->pbc[0]) fx -= floor(fx + 0.5);
        if (b->pbc[1]) fy -= floor(fy + 0.5);
        if (b->pbc[2]) fz -= floor(fz + 0.5);
        *dx = fx * h[0] + fy * h[3] + fz * h[6];
        *dy = fx * h[1] + fy * h[4] + fz * h[7];
        *dz = fx * h[2] + fy * h[5] + fz * h[8];
    } else {
        if (b->pbc[0]) *dx -= b->h[0] * floor(*dx / b->h[0] + 0.5);
        if (b->pbc[1]) *dy -= b->h[4] * floor(*dy / b->h[4] + 0.5);
        if (b->pbc[2]) *dz -= b->h[8] * floor(*dz / b->h[8] + 0.5);
    }
}

/* wrap into the primary cell, box.h:131-177 */
static inline void obox_wrap(const obox *b, double *x, double *y, double *z)
{
    if (b->tri) {
        const double *hi = b->hi, *h = b->h;
        double dx = *x - b->o[0], dy = *y - b->o[1], dz = *z - b->o[2];
        double fx = dx * hi[0] + dy * hi[3] + dz * hi[6];
        double fy = dx * hi[1] + dy * hi[4] + dz * hi[7];
        double fz = dx * hi[2] + dy * hi[5] + dz * hi[8];
        if (b->pbc[0]) fx -= floor(fx);
        if (b->pbc[1]) fy -= floor(fy);
        if (b->pbc[2]) fz -= floor(fz);
        *x = b->o[0] + fx * h[0] + fy * h[3] + fz * h[6];
        *y = b->o[1] + fx * h[1] + fy * h[4] + fz * h[7];
        *z = b->o[2] + fx * h[2] + fy * h[5] + fz * h[8];
    } else {
        if (b->pbc[0]) { double d = *x - b->o[0]; *x = b->o[0] + d - b->h[0] * floor(d / b->h[0]); }
        if (b->pbc[1]) { double d = *y - b->o[1]; *y = b->o[1] + d - b->h[4] * floor(d / b->h[4]); }
        if (b->pbc[2]) { double d = *z - b->o[2]; *z = b->o[2] + d - b->h[8] * floor(d / b->h[8]); }
    }
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int pmod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; } /* neighbor.cpp:18-22 */

/* ------------------------------------------------------------------------- */
/* Cell grid of the cutoff neighbor search (src/neighbor.cpp:24-100,203-206)  */
/* ------------------------------------------------------------------------- */
typedef struct {
    int nc[3];
    int64_t ncell;
    int64_t *start;  /* ncell+1 */
    int *atoms;      /* N, each cell's atoms in DESCENDING index order: the
                        reference inserts at the list head while i ascends
                        (neighbor.cpp:79-99), so a walk sees high indices first */
} ogrid;

/* cell coordinates of an (already wrapped) position, neighbor.cpp:29-62 */
static inline void cell_of(const obox *b, double rc_inv, const int *nc, double x, double y, double z, int *c)
{
    if (b->tri) {
        double dx = x - b->o[0], dy = y - b->o[1], dz = z - b->o[2];
        const double *hi = b->hi;
        double fx = dx * hi[0] + dy * hi[3] + dz * hi[6];
        double fy = dx * hi[1] + dy * hi[4] + dz * hi[7];
        double fz = dx * hi[2] + dy * hi[5] + dz * hi[8];
        c[0] = (int)floor(fx * b->thick[0] * rc_inv);
        c[1] = (int)floor(fy * b->thick[1] * rc_inv);
        c[2] = (int)floor(fz * b->thick[2] * rc_inv);
    } else {
        c[0] = (int)floor((x - b->o[0]) * rc_inv);
        c[1] = (int)floor((y - b->o[1]) * rc_inv);
        c[2] = (int)floor((z - b->o[2]) * rc_inv);
    }
    for (int d = 0; d < 3; ++d)
        c[d] = imax(0, imin(c[d], nc[d] - 1));
}

static inline void center_of(const obox *b, const double *x, const double *y, const double *z, int64_t i,
                             double *xi, double *yi, double *zi)
{
    *xi = x[i]; *yi = y[i]; *zi = z[i];
    if (b->pbc[0] || b->pbc[1] || b->pbc[2]) /* neighbor.cpp:88-91,139-142 */
        obox_wrap(b, xi, yi, zi);
}

static int ogrid_build(ogrid *g, const obox *b, double rc, const double *x, const double *y, const double *z, int64_t N)
{
    for (int d = 0; d < 3; ++d) /* neighbor.cpp:203-206 */
        g->nc[d] = imax((int)floor(b->thick[d] / rc), 3);
    g->ncell = (int64_t)g->nc[0] * g->nc[1] * g->nc[2];
    g->start = (int64_t *)calloc((size_t)g->ncell + 1, sizeof(int64_t));
    g->atoms = (int *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int));
    int64_t *cid = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    if (!g->start || !g->atoms || !cid)
        return -2;
    const double rc_inv = 1.0 / rc;
    for (int64_t i = 0; i < N; ++i) {
        double xi, yi, zi;
        int c[3];
        center_of(b, x, y, z, i, &xi, &yi, &zi);
        cell_of(b, rc_inv, g->nc, xi, yi, zi, c);
        cid[i] = ((int64_t)c[0] * g->nc[1] + c[1]) * g->nc[2] + c[2]; /* neighbor.cpp:24-27 */
        g->start[cid[i] + 1]++;
    }
    for (int64_t c = 0; c < g->ncell; ++c)
        g->start[c + 1] += g->start[c];
    /* fill from the back while i ascends => descending index inside a cell */
    int64_t *fill = (int64_t *)malloc((size_t)g->ncell * sizeof(int64_t));
    if (!fill)
        return -2;
    for (int64_t c = 0; c < g->ncell; ++c)
        fill[c] = g->start[c + 1];
    for (int64_t i = 0; i < N; ++i)
        g->atoms[--fill[cid[i]]] = (int)i;
    free(fill);
    free(cid);
    return 0;
}

static void ogrid_free(ogrid *g)
{
    free(g->start);
    free(g->atoms);
}

/*
 * Cutoff neighbor list, src/neighbor.cpp:102-187 (build_verlet_list) and
 * :351-388 (build_neighbor).  Caller pre-fills verlet=-1, dist=rc+1, nn=0
 * (src/mdapy/neighbor.py:125-129).  The count keeps running past max_neigh
 * (:172-177) so that the host can report overflow.  verlet/dist may be NULL
 * (count only; used for the exact-width variant :189-349).
 */
ORC_API int orc_build_neighbor(const double *x, const double *y, const double *z, int64_t N,
                               const double *box9, const double *origin, const int *boundary, double rc,
                               int *verlet, double *dist, int *nn, int64_t max_neigh, int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
    ogrid g;
    int rcode = ogrid_build(&g, &b, rc, x, y, z, N);
    if (rcode)
        return rcode;
    const double rc_inv = 1.0 / rc, rcsq = rc * rc;
    (void)num_t;
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        double xi, yi, zi;
        int c[3];
        center_of(&b, x, y, z, i, &xi, &yi, &zi);
        cell_of(&b, rc_inv, g.nc, xi, yi, zi, c);
        int64_t cnt = 0;
        for (int a = c[0] - 1; a <= c[0] + 1; ++a)        /* neighbor.cpp:147-151 */
            for (int bb = c[1] - 1; bb <= c[1] + 1; ++bb)
                for (int cc = c[2] - 1; cc <= c[2] + 1; ++cc) {
                    int64_t cell = ((int64_t)pmod(a, g.nc[0]) * g.nc[1] + pmod(bb, g.nc[1])) * g.nc[2] + pmod(cc, g.nc[2]);
                    for (int64_t p = g.start[cell]; p < g.start[cell + 1]; ++p) {
                        int j = g.atoms[p];
                        if (j == i)
                            continue;
                        double dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi; /* raw x[j], wrapped centre :164-166 */
                        obox_pbc(&b, &dx, &dy, &dz);
                        double d2 = dx * dx + dy * dy + dz * dz;
                        if (d2 <= rcsq) {
                            if (verlet && cnt < max_neigh) {
                                verlet[i * max_neigh + cnt] = j;
                                dist[i * max_neigh + cnt] = sqrt(d2);
                            }
                            ++cnt;
                        }
                    }
                }
        nn[i] = (int)cnt;
    }
    ogrid_free(&g);
    return 0;
}

/* src/neighbor.cpp:745-775 — selection of the first k by strict '<' over all M columns */
ORC_API void orc_sort_verlet_by_distance(int *verlet, double *dist, int64_t N, int64_t M, int k, int num_t)
{
    const int64_t kk = k < M ? k : M;
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        int *v = verlet + i * M;
        double *d = dist + i * M;
        for (int64_t a = 0; a < kk; ++a) {
            int64_t best = a;
            for (int64_t c = a + 1; c < M; ++c)
                if (d[c] < d[best])
                    best = c;
            if (best != a) {
                double td = d[a]; d[a] = d[best]; d[best] = td;
                int tv = v[a]; v[a] = v[best]; v[best] = tv;
            }
        }
    }
}

/* src/neighbor.cpp:675-702 */
ORC_API int orc_wrap_positions(double *x, double *y, double *z, int64_t N, const double *box9, const double *origin,
                               const int *boundary, int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < N; ++i)
        obox_wrap(&b, &x[i], &y[i], &z[i]);
    return 0;
}

/* src/neighbor.cpp:704-743 */
ORC_API void orc_average_by_neighbor(double rc, const int *verlet, const double *dist, const int *nn, int64_t N,
                                     int64_t M, const double *value, double *out, int include_self, int num_t)
{
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        double s = 0.0;
        int cnt = 0;
        if (include_self) { s += value[i]; ++cnt; }
        for (int64_t j = 0; j < nn[i]; ++j)
            if (dist[i * M + j] <= rc) { s += value[verlet[i * M + j]]; ++cnt; }
        out[i] = cnt > 0 ? s / cnt : 0.0;
    }
}

/* ------------------------------------------------------------------------- */
/* Common neighbour analysis  (src/cna.cpp)                                   */
/* ------------------------------------------------------------------------- */
static inline double pair_d2(const obox *b, const double *x, const double *y, const double *z, int i, int j)
{ /* cna.cpp:149-161 — both ends RAW coordinates */
    double dx = x[j] - x[i], dy = y[j] - y[i], dz = z[j] - z[i];
    obox_pbc(b, &dx, &dy, &dz);
    return dx * dx + dy * dy + dz * dz;
}

/* adjacency among the nn listed neighbours: bit b of adj[a] <=> d2(a,b) <= cut2   (cna.cpp:16-48,459-466) */
static void bond_matrix(const obox *b, const double *x, const double *y, const double *z, const int *ids, int nn,
                        double cut2, uint32_t *adj)
{
    for (int a = 0; a < nn; ++a)
        adj[a] = 0;
    for (int a = 0; a < nn; ++a)
        for (int c = a + 1; c < nn; ++c)
            if (pair_d2(b, x, y, z, ids[a], ids[c]) <= cut2) {
                adj[a] |= 1u << c;
                adj[c] |= 1u << a;
            }
}

/*
 * CNA signature of the bond centre--neighbour `ni`:
 *   ncn   = # common neighbours                           (cna.cpp:52-64)
 *   nb    = # bonds among the common neighbours; only neighbours with
 *           index < nlimit take part                     (cna.cpp:69-92; the adaptive
 *           12-neighbour pass hands 12 here while sizing for 14, :344)
 *   chain = # bonds in the largest connected bond cluster (cna.cpp:97-147)
 */
static void cna_signature(const uint32_t *adj, int ni, int nlimit, int *ncn, int *nb, int *chain)
{
    uint32_t common = adj[ni];
    *ncn = __builtin_popcount(common);
    uint32_t pool = common & (nlimit >= 32 ? 0xffffffffu : ((1u << nlimit) - 1u));
    int bonds = 0;
    for (uint32_t m = pool; m; m &= m - 1) {
        int a = __builtin_ctz(m);
        bonds += __builtin_popcount(adj[a] & pool);
    }
    *nb = bonds / 2;
    /* largest connected component, measured in bonds */
    int best = 0;
    uint32_t left = pool;
    while (left) {
        uint32_t comp = left & (~left + 1u), frontier = comp;
        while (frontier) {
            int a = __builtin_ctz(frontier);
            frontier &= frontier - 1;
            uint32_t grow = adj[a] & pool & ~comp;
            comp |= grow;
            frontier |= grow;
        }
        int cb = 0;
        for (uint32_t m = comp; m; m &= m - 1)
            cb += __builtin_popcount(adj[__builtin_ctz(m)] & comp);
        cb /= 2;
        if (cb > best)
            best = cb;
        left &= ~comp;
    }
    *chain = best;
}

/* src/cna.cpp:429-506 FixedCNA.  pattern must be pre-zeroed by the caller. */
ORC_API int orc_fcna(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                     const double *origin, const int *boundary, const int *verlet, int64_t M, const int *nn,
                     int *pattern, double rc, int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
    const double cut2 = rc * rc;
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        int n = nn[i];
        if (n != 12 && n != 14)
            continue;
        uint32_t adj[32];
        bond_matrix(&b, x, y, z, verlet + i * M, n, cut2, adj);
        int n421 = 0, n422 = 0, n555 = 0, n444 = 0, n666 = 0;
        for (int ni = 0; ni < n; ++ni) { /* no early exit: cna.cpp:471-494 */
            int ncn, nb, ch;
            cna_signature(adj, ni, n, &ncn, &nb, &ch);
            if (ncn == 4 && nb == 2) { if (ch == 1) n421++; else if (ch == 2) n422++; }
            else if (ncn == 5 && nb == 5 && ch == 5) n555++;
            else if (ncn == 4 && nb == 4 && ch == 4) n444++;
            else if (ncn == 6 && nb == 6 && ch == 6) n666++;
        }
        if (n421 == 12) pattern[i] = 1;                     /* cna.cpp:496-503 */
        else if (n421 == 6 && n422 == 6) pattern[i] = 2;
        else if (n555 == 12) pattern[i] = 4;
        else if (n666 == 8 && n444 == 6) pattern[i] = 3;
    }
    return 0;
}

/* src/cna.cpp:289-427 AdaptiveCNA: verlet rows must be distance-sorted, >=14 wide. */
ORC_API int orc_acna(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                     const double *origin, const int *boundary, const int *verlet, int64_t M, int *pattern, int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const int *row = verlet + i * M;
        uint32_t adj[32];
        double rs = 0.0;
        for (int m = 0; m < 12; ++m)                         /* :312-317 */
            rs += sqrt(pair_d2(&b, x, y, z, (int)i, row[m]));
        double lc = rs / 12 * (1.0 + sqrt(2.0)) * 0.5;       /* :319 */
        bond_matrix(&b, x, y, z, row, 12, lc * lc, adj);
        int n421 = 0, n422 = 0, n555 = 0;
        for (int ni = 0; ni < 12; ++ni) {                    /* breaks: :334-362 */
            int ncn, nb, ch;
            cna_signature(adj, ni, 12, &ncn, &nb, &ch);
            if (ncn != 4 && ncn != 5) break;
            if (nb != 2 && nb != 5) break;
            if (ncn == 4 && nb == 2) {
                if (ch == 1) n421++;
                else if (ch == 2) n422++;
                else break;
            } else if (ncn == 5 && nb == 5 && ch == 5) n555++;
            else break;
        }
        if (n421 == 12) pattern[i] = 1;
        else if (n421 == 6 && n422 == 6) pattern[i] = 2;
        else if (n555 == 12) pattern[i] = 4;
        if (pattern[i] != 0)
            continue;
        rs = 0.0;                                            /* :372-387 */
        for (int m = 0; m < 8; ++m)
            rs += sqrt(pair_d2(&b, x, y, z, (int)i, row[m]) / (3.0 / 4.0));
        for (int m = 8; m < 14; ++m)
            rs += sqrt(pair_d2(&b, x, y, z, (int)i, row[m]));
        lc = rs / 14 * (1.0 + sqrt(2.0)) * 0.5;
        bond_matrix(&b, x, y, z, row, 14, lc * lc, adj);
        int n444 = 0, n666 = 0;
        for (int ni = 0; ni < 14; ++ni) {                    /* :398-421 */
            int ncn, nb, ch;
            cna_signature(adj, ni, 14, &ncn, &nb, &ch);
            if (ncn != 4 && ncn != 6) break;
            if (nb != 4 && nb != 6) break;
            if (ncn == 4 && nb == 4 && ch == 4) n444++;
            else if (ncn == 6 && nb == 6 && ch == 6) n666++;
            else break;
        }
        if (n666 == 8 && n444 == 6) pattern[i] = 3;
    }
    return 0;
}

/* src/cna.cpp:163-287 IdentifyDiamond.  verlet rows: >=4 nearest neighbours, sorted. */
ORC_API int orc_ids(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                    const double *origin, const int *boundary, const int *verlet, int64_t M, int *second /* N x 12 */,
                    int *pattern, int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        int *s2 = second + i * 12;
        int cnt = 0;
        for (int m = 0; m < 4; ++m) {                        /* :188-202 */
            int j = verlet[i * M + m], took = 0;
            for (int q = 0; q < 4; ++q) {
                int k = verlet[(int64_t)j * M + q];
                if (k != i && took < 3) { s2[cnt++] = k; ++took; }
            }
        }
        double rs = 0.0;
        for (int m = 0; m < 12; ++m)
            rs += sqrt(pair_d2(&b, x, y, z, (int)i, s2[m]));
        rs /= 12.0;
        double lc = rs * 1.2071068;                          /* :212 */
        uint32_t adj[32];
        bond_matrix(&b, x, y, z, s2, 12, lc * lc, adj);
        int n421 = 0, n422 = 0;
        for (int ni = 0; ni < 12; ++ni) {                    /* :224-245 */
            int ncn, nb, ch;
            cna_signature(adj, ni, 12, &ncn, &nb, &ch);
            if (ncn != 4) break;
            if (nb != 2) break;
            if (ch == 1) n421++;
            else if (ch == 2) n422++;
        }
        if (n421 == 12) pattern[i] = 1;
        else if (n421 == 6 && n422 == 6) pattern[i] = 4;
    }
    /* two sequential, order dependent sweeps (:253-286) */
    for (int64_t i = 0; i < N; ++i) {
        int t = pattern[i];
        if (t != 1 && t != 4) continue;
        for (int q = 0; q < 4; ++q) {
            int j = verlet[i * M + q];
            if (pattern[j] == 0) pattern[j] = (t == 1) ? 2 : 5;
        }
    }
    for (int64_t i = 0; i < N; ++i) {
        int t = pattern[i];
        if (t != 2 && t != 5) continue;
        for (int q = 0; q < 4; ++q) {
            int j = verlet[i * M + q];
            if (pattern[j] == 0) pattern[j] = (t == 2) ? 3 : 6;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Centro-symmetry parameter  (src/centro_symmetry_parameter.cpp:12-94)       */
/* ------------------------------------------------------------------------- */
static int cmp_double(const void *a, const void *b)
{
    double p = *(const double *)a, q = *(const double *)b;
    return (p > q) - (p < q);
}

ORC_API int orc_csp(const double *x, const double *y, const double *z, int64_t n_atoms, const double *box9,
                    const double *origin, const int *boundary, const int *verlet, int64_t M, int K, double *csp,
                    int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
    const int npair = K * (K - 1) / 2, half = K / 2;
#pragma omp parallel num_threads(num_t > 0 ? num_t : 1)
    {
        double *pd = (double *)malloc(sizeof(double) * (size_t)(npair > 0 ? npair : 1));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n_atoms; ++i) {
            const double xi = x[i], yi = y[i], zi = z[i];  /* RAW centre :46-48 */
            int p = 0;
            for (int a = 0; a < K; ++a)
                for (int c = a + 1; c < K; ++c) {
                    int j = verlet[i * M + a], k = verlet[i * M + c];
                    double ax = x[j] - xi, ay = y[j] - yi, az = z[j] - zi;
                    double bx = x[k] - xi, by = y[k] - yi, bz = z[k] - zi;
                    obox_pbc(&b, &ax, &ay, &az);
                    obox_pbc(&b, &bx, &by, &bz);
                    double sx = ax + bx, sy = ay + by, sz = az + bz;
                    pd[p++] = sx * sx + sy * sy + sz * sz;
                }
            qsort(pd, (size_t)npair, sizeof(double), cmp_double); /* partial_sort :79-81: same smallest values */
            double s = 0.0;
            for (int q = 0; q < half; ++q)
                s += pd[q];
            csp[i] = s;
        }
        free(pd);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Steinhardt bond orientation (src/steinhardt_bond_orientation.cpp)          */
/* ------------------------------------------------------------------------- */
static double fact_tab[168];
static int fact_ready = 0;
static void fact_init(void) /* :12-181 h_factorial: n! as doubles */
{
    if (fact_ready)
        return;
    long double f = 1.0L;
    fact_tab[0] = 1.0;
    for (int n = 1; n < 168; ++n) {
        f *= (long double)n;
        fact_tab[n] = (double)f;
    }
    fact_ready = 1;
}

static double assoc_legendre(int l, int m, double x) /* :243-268 */
{
    if (l < m)
        return 0.0;
    double p = 1.0, pm1 = 0.0, pm2 = 0.0;
    if (m != 0) {
        double sqx = sqrt(1.0 - x * x);
        for (int i = 1; i < m + 1; ++i)
            p *= (2 * i - 1) * sqx;
    }
    for (int i = m + 1; i < l + 1; ++i) {
        pm2 = pm1;
        pm1 = p;
        p = ((2 * i - 1) * x * pm1 - (i + m - 1) * pm2) / (i - m);
    }
    return p;
}

static double polar_prefactor(int l, int m, double ct) /* :270-286 */
{
    const double PI = 3.14159265358979323846;
    int ma = m < 0 ? -m : m;
    double pf = 1.0;
    for (int i = l - ma + 1; i < l + ma + 1; ++i)
        pf *= i;
    pf = sqrt((2 * l + 1) / (4 * PI * pf)) * assoc_legendre(l, ma, ct);
    if ((m < 0) && (m % 2))
        pf = -pf;
    return pf;
}

static int cg_count(const int *llist, int nl) /* :226-241 */
{
    int c = 0;
    for (int il = 0; il < nl; ++il) {
        int l = llist[il];
        for (int m1 = 0; m1 < 2 * l + 1; ++m1)
            for (int m2 = imax(0, l - m1); m2 < imin(2 * l + 1, 3 * l - m1 + 1); ++m2)
                ++c;
    }
    return c;
}

static void cg_fill(double *cg, const int *llist, int nl) /* :188-224 */
{
    fact_init();
    int c = 0;
    for (int il = 0; il < nl; ++il) {
        int l = llist[il];
        for (int m1 = 0; m1 < 2 * l + 1; ++m1) {
            int aa2 = m1 - l;
            for (int m2 = imax(0, l - m1); m2 < imin(2 * l + 1, 3 * l - m1 + 1); ++m2) {
                int bb2 = m2 - l;
                int m = aa2 + bb2 + l;
                double sums = 0.0;
                for (int zz = imax(0, imax(-aa2, bb2)); zz < imin(l, imin(l - aa2, l + bb2)) + 1; ++zz) {
                    int ifac = (zz % 2) ? -1 : 1;
                    sums += ifac / (fact_tab[zz] * fact_tab[l - zz] * fact_tab[l - aa2 - zz] * fact_tab[l + bb2 - zz] *
                                    fact_tab[aa2 + zz] * fact_tab[-bb2 + zz]);
                }
                int cc2 = m - l;
                double sfaccg = sqrt(fact_tab[l + aa2] * fact_tab[l - aa2] * fact_tab[l + bb2] * fact_tab[l - bb2] *
                                     fact_tab[l + cc2] * fact_tab[l - cc2] * (2 * l + 1));
                double sfac1 = fact_tab[3 * l + 1];
                double sfac2 = fact_tab[l];
                double dcg = sqrt(sfac2 * sfac2 * sfac2 / sfac1);
                cg[c++] = sums * dcg * sfaccg;
            }
        }
    }
}

/*
 * get_sq / _compute_ql  (:288-576, :677-784).
 * qlm_r/qlm_i (N, nl, 2*lmax+1) must be pre-zeroed; qn (N, ncol).
 * weight may be NULL when use_weight == 0.
 */
ORC_API int orc_get_sq(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                       const double *origin, const int *boundary, const int *NL, const double *DL, int64_t M,
                       const int *NN, const double *weight, const int *llist, int nl, int nnn, int lmax, int wl,
                       int wlhat, int average, int use_voronoi, double rc, int use_weight, double *qlm_r,
                       double *qlm_i, double *qn, int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
    const double EPS = 1e-15, PI = 3.14159265358979323846;
    const int nz = 2 * lmax + 1;
    const int64_t stride = (int64_t)nl * nz;
    int ncol = nl + (wl ? nl : 0) + (wlhat ? nl : 0);
    double *cg = NULL;
    if (wl || wlhat) {
        cg = (double *)malloc(sizeof(double) * (size_t)imax(1, cg_count(llist, nl)));
        cg_fill(cg, llist, nl);
    }
    /* stage 1 :323-436 */
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(dynamic, 16)
    for (int64_t i = 0; i < N; ++i) {
        double wsum = 0.0;
        int cnt = NN[i];
        if (!use_voronoi && nnn > 0)
            cnt = nnn;
        double *qr = qlm_r + i * stride, *qi = qlm_i + i * stride;
        for (int jj = 0; jj < cnt; ++jj) {
            int64_t idx = i * M + jj;
            int j = NL[idx];
            if (j < 0)
                continue;
            double dx = x[j] - x[i], dy = y[j] - y[i], dz = z[j] - z[i];
            obox_pbc(&b, &dx, &dy, &dz);
            double r = DL[idx];
            if (!(r > EPS && r <= rc))
                continue;
            double w = use_weight ? weight[idx] : 1.0;
            wsum += w;
            double rinv = 1.0 / r;
            double ct = dz * rinv;
            double er = dx, ei = dy;
            double rxy2 = er * er + ei * ei;
            if (rxy2 < EPS * EPS) { er = 1.0; ei = 0.0; }
            else { double s = 1.0 / sqrt(rxy2); er *= s; ei *= s; }
            for (int il = 0; il < nl; ++il) {
                int l = llist[il];
                double *pr = qr + il * nz, *pi = qi + il * nz;
                pr[l] += w * polar_prefactor(l, 0, ct);
                double mr = er, mi = ei;
                for (int m = 1; m < l + 1; ++m) {
                    double pf = polar_prefactor(l, m, ct);
                    double cr = pf * mr, ci = pf * mi;
                    double wr = w * cr, wi = w * ci;
                    pr[l + m] += wr;
                    pi[l + m] += wi;
                    if (m & 1) { pr[l - m] -= wr; pi[l - m] += wi; }
                    else { pr[l - m] += wr; pi[l - m] -= wi; }
                    double tr = mr * er - mi * ei, ti = mr * ei + mi * er;
                    mr = tr; mi = ti;
                }
            }
        }
        double f = 1.0 / wsum; /* no guard: NaN/inf for neighbour-less atoms :422 */
        for (int il = 0; il < nl; ++il) {
            int l = llist[il];
            for (int m = 0; m < 2 * l + 1; ++m) { qr[il * nz + m] *= f; qi[il * nz + m] *= f; }
        }
    }
    /* stage 2 :439-503 */
    if (average) {
        double *ar = (double *)malloc(sizeof(double) * (size_t)(N * stride + 1));
        double *ai = (double *)malloc(sizeof(double) * (size_t)(N * stride + 1));
        memcpy(ar, qlm_r, sizeof(double) * (size_t)(N * stride));
        memcpy(ai, qlm_i, sizeof(double) * (size_t)(N * stride));
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(dynamic, 16)
        for (int64_t i = 0; i < N; ++i) {
            int cnt = NN[i];
            if (!use_voronoi && nnn > 0)
                cnt = nnn;
            int nb = 1;
            double *qr = qlm_r + i * stride, *qi = qlm_i + i * stride;
            for (int jj = 0; jj < cnt; ++jj) {
                int j = NL[i * M + jj];
                if (j < 0)
                    continue;
                for (int il = 0; il < nl; ++il) {
                    int l = llist[il];
                    for (int m = 0; m < 2 * l + 1; ++m) {
                        qr[il * nz + m] += ar[(int64_t)j * stride + il * nz + m];
                        qi[il * nz + m] += ai[(int64_t)j * stride + il * nz + m];
                    }
                }
                ++nb;
            }
            double inv = 1.0 / nb;
            for (int il = 0; il < nl; ++il) {
                int l = llist[il];
                for (int m = 0; m < 2 * l + 1; ++m) { qr[il * nz + m] *= inv; qi[il * nz + m] *= inv; }
            }
        }
        free(ar);
        free(ai);
    }
    /* stage 3 :506-575 */
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const double *qr = qlm_r + i * stride, *qi = qlm_i + i * stride;
        double *out = qn + i * ncol;
        for (int il = 0; il < nl; ++il) {
            int l = llist[il];
            double nf = sqrt(4 * PI / (2 * l + 1));
            double s = 0.0;
            for (int m = 0; m < 2 * l + 1; ++m)
                s += qr[il * nz + m] * qr[il * nz + m] + qi[il * nz + m] * qi[il * nz + m];
            out[il] = nf * sqrt(s);
        }
        if (wl || wlhat) {
            int c = 0;
            for (int il = 0; il < nl; ++il) {
                int l = llist[il];
                const double *pr = qr + il * nz, *pi = qi + il * nz;
                double ws = 0.0;
                for (int m1 = 0; m1 < 2 * l + 1; ++m1)
                    for (int m2 = imax(0, l - m1); m2 < imin(2 * l + 1, 3 * l - m1 + 1); ++m2) {
                        int m = m1 + m2 - l;
                        double ar_ = pr[m1] * pr[m2] - pi[m1] * pi[m2];
                        double ai_ = pr[m1] * pi[m2] + pi[m1] * pr[m2];
                        ws += (ar_ * pr[m] + ai_ * pi[m]) * cg[c];
                        ++c;
                    }
                double wf = ws / sqrt(2 * l + 1.0);
                if (wl)
                    out[il + nl] = wf;
                if (wlhat) {
                    double q = out[il];
                    if (q > EPS) {
                        double nf = sqrt(4 * PI / (2 * l + 1));
                        double g = nf / q;
                        out[il + (wl ? nl : 0) + nl] = wf * (g * g * g);
                    }
                }
            }
        }
    }
    free(cg);
    return 0;
}

/* identifySolidLiquid (:578-675).  The second pass reads labels while other
 * threads clear them in the reference (racy by construction); this restatement
 * evaluates it against the pass-1 labels, which is what a single-threaded run
 * of the reference gives only when no two isolated-solid atoms are neighbours;
 * tests use cases where both readings agree. */
ORC_API void orc_identify_solid_liquid(int q6index, const double *Q6, const int *verlet, const double *dist,
                                       const int *nn, int64_t N, int64_t M, const double *qlm_r, const double *qlm_i,
                                       int nl, int nz, double threshold, int n_bond, int *solid, int *nbond,
                                       int use_voronoi, int nnn, double rc, int num_t)
{
    const double PI = 3.14159265358979323846;
    const int64_t stride = (int64_t)nl * nz;
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        int cnt = nn[i], nsb = 0;
        if (!use_voronoi && nnn > 0)
            cnt = nnn;
        for (int jj = 0; jj < cnt; ++jj) {
            int j = verlet[i * M + jj];
            if (j < 0) continue;
            if (dist[i * M + jj] > rc) continue;
            double s = 0.0;
            for (int m = 0; m < 13; ++m)
                s += qlm_r[i * stride + q6index * nz + m] * qlm_r[(int64_t)j * stride + q6index * nz + m] +
                     qlm_i[i * stride + q6index * nz + m] * qlm_i[(int64_t)j * stride + q6index * nz + m];
            s = s / Q6[i] / Q6[j] * 4 * PI / 13;
            if (s > threshold) ++nsb;
        }
        if (nsb >= n_bond) solid[i] = 1;
        nbond[i] = nsb;
    }
    int *snap = (int *)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
    memcpy(snap, solid, sizeof(int) * (size_t)N);
    for (int64_t i = 0; i < N; ++i) {
        if (snap[i] != 1) continue;
        int cnt = nn[i], any = 0;
        if (!use_voronoi && nnn > 0)
            cnt = nnn;
        for (int jj = 0; jj < cnt; ++jj) {
            int j = verlet[i * M + jj];
            if (j < 0) continue;
            if (snap[j] == 1) { any = 1; break; }
        }
        if (!any) solid[i] = 0;
    }
    free(snap);
}

/* ------------------------------------------------------------------------- */
/* Radial distribution function (src/radial_distribution_function.cpp)        */
/* ------------------------------------------------------------------------- */
/* _rdf :22-54 — g (Nt,Nt,nbin) accumulated */
ORC_API void orc_rdf(const int *verlet, const double *dist, const int *nn, const int *type, int64_t N, int64_t M,
                     double *g, int Nt, double rc, int nbin)
{
    const double dr = rc / nbin;
    for (int64_t i = 0; i < N; ++i)
        for (int q = 0; q < nn[i]; ++q) {
            double d = dist[i * M + q];
            if (d < rc) {
                int j = verlet[i * M + q];
                int k = (int)(d / dr);
                g[((int64_t)type[i] * Nt + type[j]) * nbin + k] += 1.0;
            }
        }
}

/* _rdf_single_species :56-85 */
ORC_API void orc_rdf_single(const int *verlet, const double *dist, const int *nn, int64_t N, int64_t M, double *g,
                            double rc, int nbin)
{
    const double dr = rc / nbin;
    for (int64_t i = 0; i < N; ++i)
        for (int q = 0; q < nn[i]; ++q) {
            int j = verlet[i * M + q];
            double d = dist[i * M + q];
            if (j > i && d < rc)
                g[(int)(d / dr)] += 2.0;
        }
}

/* _rdf_streaming :143-317 */
ORC_API int orc_rdf_streaming(const double *x, const double *y, const double *z, const int *type, int64_t N,
                              const double *box9, const double *origin, const int *boundary, double *g, int Nt,
                              double rc, int nbin, int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
    const double dr = rc / nbin, rcsq = rc * rc;
    int nc[3], use_cells = 1;
    for (int d = 0; d < 3; ++d) { /* :163-172 */
        nc[d] = imax(1, (int)floor(b.thick[d] / rc));
        if (b.pbc[d] && nc[d] < 3)
            use_cells = 0;
    }
    const int any_pbc = b.pbc[0] || b.pbc[1] || b.pbc[2];
    const int64_t hsize = (int64_t)Nt * Nt * nbin;
    uint64_t *hist = (uint64_t *)calloc((size_t)hsize, sizeof(uint64_t)); /* integer counts: order independent */
    int64_t ncell = (int64_t)nc[0] * nc[1] * nc[2];
    int64_t *start = NULL;
    int *atoms = NULL;
    if (use_cells) {
        start = (int64_t *)calloc((size_t)ncell + 1, sizeof(int64_t));
        atoms = (int *)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
        int64_t *cid = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
        for (int64_t i = 0; i < N; ++i) {
            double xi = x[i], yi = y[i], zi = z[i];
            if (any_pbc) obox_wrap(&b, &xi, &yi, &zi);
            int c[3];
            if (b.tri) { /* cell_index_for :109-141 */
                double dx = xi - b.o[0], dy = yi - b.o[1], dz = zi - b.o[2];
                double fx = dx * b.hi[0] + dy * b.hi[3] + dz * b.hi[6];
                double fy = dx * b.hi[1] + dy * b.hi[4] + dz * b.hi[7];
                double fz = dx * b.hi[2] + dy * b.hi[5] + dz * b.hi[8];
                c[0] = (int)floor(fx * nc[0]); c[1] = (int)floor(fy * nc[1]); c[2] = (int)floor(fz * nc[2]);
            } else {
                c[0] = (int)floor((xi - b.o[0]) / b.h[0] * nc[0]);
                c[1] = (int)floor((yi - b.o[1]) / b.h[4] * nc[1]);
                c[2] = (int)floor((zi - b.o[2]) / b.h[8] * nc[2]);
            }
            for (int d = 0; d < 3; ++d) c[d] = imax(0, imin(c[d], nc[d] - 1));
            cid[i] = ((int64_t)c[0] * nc[1] + c[1]) * nc[2] + c[2];
            start[cid[i] + 1]++;
        }
        for (int64_t c = 0; c < ncell; ++c) start[c + 1] += start[c];
        int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)ncell);
        for (int64_t c = 0; c < ncell; ++c) fill[c] = start[c];
        for (int64_t i = 0; i < N; ++i) atoms[fill[cid[i]]++] = (int)i;
        /* keep cid for the scan below */
#pragma omp parallel num_threads(num_t > 0 ? num_t : 1)
        {
            uint64_t *loc = (uint64_t *)calloc((size_t)hsize, sizeof(uint64_t));
#pragma omp for schedule(dynamic, 64)
            for (int64_t i = 0; i < N; ++i) {
                double xi = x[i], yi = y[i], zi = z[i];
                if (any_pbc) obox_wrap(&b, &xi, &yi, &zi);
                int64_t ci = cid[i];
                int c2 = (int)(ci % nc[2]), c1 = (int)((ci / nc[2]) % nc[1]), c0 = (int)(ci / ((int64_t)nc[1] * nc[2]));
                for (int da = -1; da <= 1; ++da) { /* :223-235 non-periodic axes are NOT wrapped */
                    int a = b.pbc[0] ? pmod(c0 + da, nc[0]) : c0 + da;
                    if (a < 0 || a >= nc[0]) continue;
                    for (int db = -1; db <= 1; ++db) {
                        int bb = b.pbc[1] ? pmod(c1 + db, nc[1]) : c1 + db;
                        if (bb < 0 || bb >= nc[1]) continue;
                        for (int dc = -1; dc <= 1; ++dc) {
                            int cc = b.pbc[2] ? pmod(c2 + dc, nc[2]) : c2 + dc;
                            if (cc < 0 || cc >= nc[2]) continue;
                            int64_t cell = ((int64_t)a * nc[1] + bb) * nc[2] + cc;
                            for (int64_t p = start[cell]; p < start[cell + 1]; ++p) {
                                int j = atoms[p];
                                if (j == i) continue;
                                double dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
                                obox_pbc(&b, &dx, &dy, &dz);
                                double r2 = dx * dx + dy * dy + dz * dz;
                                if (r2 < rcsq) {
                                    int k = (int)(sqrt(r2) / dr);
                                    if (k < nbin)
                                        loc[((int64_t)type[i] * Nt + type[j]) * nbin + k]++;
                                }
                            }
                        }
                    }
                }
            }
#pragma omp critical
            for (int64_t q = 0; q < hsize; ++q) hist[q] += loc[q];
            free(loc);
        }
        free(fill);
        free(cid);
    } else { /* all pairs :266-305 */
#pragma omp parallel num_threads(num_t > 0 ? num_t : 1)
        {
            uint64_t *loc = (uint64_t *)calloc((size_t)hsize, sizeof(uint64_t));
#pragma omp for schedule(dynamic, 64)
            for (int64_t i = 0; i < N; ++i) {
                double xi = x[i], yi = y[i], zi = z[i];
                if (any_pbc) obox_wrap(&b, &xi, &yi, &zi);
                for (int64_t j = 0; j < N; ++j) {
                    if (j == i) continue;
                    double dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
                    obox_pbc(&b, &dx, &dy, &dz);
                    double r2 = dx * dx + dy * dy + dz * dz;
                    if (r2 < rcsq) {
                        int k = (int)(sqrt(r2) / dr);
                        if (k < nbin)
                            loc[((int64_t)type[i] * Nt + type[j]) * nbin + k]++;
                    }
                }
            }
#pragma omp critical
            for (int64_t q = 0; q < hsize; ++q) hist[q] += loc[q];
            free(loc);
        }
    }
    for (int64_t q = 0; q < hsize; ++q) g[q] += (double)hist[q]; /* :308-316 '+=' */
    free(hist);
    free(start);
    free(atoms);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Warren-Cowley parameter (src/warren_cowley_parameter.cpp:9-80)             */
/* ------------------------------------------------------------------------- */
ORC_API void orc_wcp(const int *verlet, const int *nn, const int *type, int64_t N, int64_t M, int Nt, double *wcp)
{
    int64_t *zmn = (int64_t *)calloc((size_t)Nt * Nt, sizeof(int64_t));
    int64_t *zm = (int64_t *)calloc((size_t)Nt, sizeof(int64_t));
    double *conc = (double *)calloc((size_t)Nt, sizeof(double));
    for (int64_t i = 0; i < N; ++i) {
        int ti = type[i];
        conc[ti] += 1.0;
        zm[ti] += nn[i];
        for (int q = 0; q < nn[i]; ++q)
            zmn[(int64_t)ti * Nt + type[verlet[i * M + q]]]++;
    }
    for (int t = 0; t < Nt; ++t)
        conc[t] /= (double)N;
    for (int a = 0; a < Nt; ++a)
        for (int c = 0; c < Nt; ++c)
            wcp[a * Nt + c] = (conc[c] > 0 && zm[a] > 0) ? 1.0 - (double)zmn[a * Nt + c] / (conc[c] * (double)zm[a]) : 0.0;
    free(zmn);
    free(zm);
    free(conc);
}

/* ------------------------------------------------------------------------- */
/* Exact k nearest neighbours (src/fast_knn.cpp:846-916, orthogonal + triclinic)
 * The reference walks a kd-tree; exact kNN is defined by its result, so this
 * restatement enumerates every (atom, image) candidate with the reference's
 * distance arithmetic:
 *   wrap:   s=floor((p-O)*(1/L)); if (s!=0) p-=s*L            (:688-703, :743-757)   [orthogonal]
 *   images: +-nimages per periodic axis, nimages=200/clamp(N,50,200) (>=2 if triclinic) (:801-841)
 *   query:  q = q_wrapped - shift;  d2 = (a-q).(a-q)            (:598-603,:759-770)
 *   self excluded only when idx==self && d2==0.0              (:641)
 * Ties in d2 are ordered by (d2, index, image order) here; the reference's
 * order under exact ties is traversal dependent.
 * Triclinic wrap uses fractional coordinates (KdTree::build :212-260).
 * ------------------------------------------------------------------------- */
ORC_API int orc_knn(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                    const double *origin, const int *boundary, int k, int *indices, double *distances, int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
    int any = b.pbc[0] || b.pbc[1] || b.pbc[2];
    int nim = 1;
    if (any) {
        int64_t cl = N < 50 ? 50 : (N > 200 ? 200 : N);
        nim = (int)(200 / cl);
        if (nim < 1) nim = 1;
        if (nim < 2 && b.tri) nim = 2;
    }
    int nx = b.pbc[0] ? nim : 0, ny = b.pbc[1] ? nim : 0, nzz = b.pbc[2] ? nim : 0;
    int ns = (2 * nx + 1) * (2 * ny + 1) * (2 * nzz + 1);
    double *sh = (double *)malloc(sizeof(double) * 3 * (size_t)ns);
    int c = 0;
    for (int iz = -nzz; iz <= nzz; ++iz)
        for (int iy = -ny; iy <= ny; ++iy)
            for (int ix = -nx; ix <= nx; ++ix) {
                if (b.tri) {
                    sh[3 * c + 0] = ix * b.h[0] + iy * b.h[3] + iz * b.h[6];
                    sh[3 * c + 1] = ix * b.h[1] + iy * b.h[4] + iz * b.h[7];
                    sh[3 * c + 2] = ix * b.h[2] + iy * b.h[5] + iz * b.h[8];
                } else {
                    sh[3 * c + 0] = ix * b.h[0];
                    sh[3 * c + 1] = iy * b.h[4];
                    sh[3 * c + 2] = iz * b.h[8];
                }
                ++c;
            }
    double *wx = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    double *wy = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    double *wz = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    for (int64_t i = 0; i < N; ++i) {
        double px = x[i], py = y[i], pz = z[i];
        if (b.tri) { /* wrap_triclinic :86-99 — reduced coords WITHOUT origin shift */
            double r[3] = {px * b.hi[0] + py * b.hi[3] + pz * b.hi[6], px * b.hi[1] + py * b.hi[4] + pz * b.hi[7],
                           px * b.hi[2] + py * b.hi[5] + pz * b.hi[8]};
            for (int d = 0; d < 3; ++d)
                if (b.pbc[d]) {
                    double s = floor(r[d]);
                    if (s != 0.0) { px -= s * b.h[d * 3]; py -= s * b.h[d * 3 + 1]; pz -= s * b.h[d * 3 + 2]; }
                }
        } else {
            if (b.pbc[0]) { double s = floor((px - b.o[0]) * (1.0 / b.h[0])); if (s != 0.0) px -= s * b.h[0]; }
            if (b.pbc[1]) { double s = floor((py - b.o[1]) * (1.0 / b.h[4])); if (s != 0.0) py -= s * b.h[4]; }
            if (b.pbc[2]) { double s = floor((pz - b.o[2]) * (1.0 / b.h[8])); if (s != 0.0) pz -= s * b.h[8]; }
        }
        wx[i] = px; wy[i] = py; wz[i] = pz;
    }
#pragma omp parallel num_threads(num_t > 0 ? num_t : 1)
    {
        double *bd = (double *)malloc(sizeof(double) * (size_t)k);
        int *bi = (int *)malloc(sizeof(int) * (size_t)k);
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < N; ++i) {
            int n = 0;
            for (int s = 0; s < ns; ++s) {
                double q0 = wx[i] - sh[3 * s], q1 = wy[i] - sh[3 * s + 1], q2 = wz[i] - sh[3 * s + 2];
                for (int64_t j = 0; j < N; ++j) {
                    double dx = wx[j] - q0, dy = wy[j] - q1, dz = wz[j] - q2;
                    double d2 = dx * dx + dy * dy + dz * dz;
                    if (j == i && d2 == 0.0)
                        continue;
                    if (n == k && !(d2 < bd[k - 1] || (d2 == bd[k - 1] && (int)j < bi[k - 1])))
                        continue;
                    int pos = n < k ? n : k - 1;
                    while (pos > 0 && (bd[pos - 1] > d2 || (bd[pos - 1] == d2 && bi[pos - 1] > (int)j))) {
                        bd[pos] = bd[pos - 1];
                        bi[pos] = bi[pos - 1];
                        --pos;
                    }
                    bd[pos] = d2;
                    bi[pos] = (int)j;
                    if (n < k) ++n;
                }
            }
            for (int q = 0; q < n; ++q) { indices[i * k + q] = bi[q]; distances[i * k + q] = sqrt(bd[q]); }
            for (int q = n; q < k; ++q) { indices[i * k + q] = -1; distances[i * k + q] = -1.0; }
        }
        free(bd);
        free(bi);
    }
    free(sh); free(wx); free(wy); free(wz);
    return 0;
}

/* src/repeat_cell.cpp:19-61 */
ORC_API void orc_repeat_cell(double *newp, const double *box9, const double *oldp, int64_t n_old, int nx, int ny,
                             int nz)
{
    for (int64_t cell = 0; cell < (int64_t)nx * ny * nz; ++cell) {
        int ix = (int)(cell / ((int64_t)ny * nz));
        int64_t t = cell % ((int64_t)ny * nz);
        int iy = (int)(t / nz), iz = (int)(t % nz);
        double sx = ix * box9[0] + iy * box9[3] + iz * box9[6];
        double sy = ix * box9[1] + iy * box9[4] + iz * box9[7];
        double sz = ix * box9[2] + iy * box9[5] + iz * box9[8];
        for (int64_t i = 0; i < n_old; ++i) {
            int64_t o = (cell * n_old + i) * 3;
            newp[o] = oldp[i * 3] + sx;
            newp[o + 1] = oldp[i * 3 + 1] + sy;
            newp[o + 2] = oldp[i * 3 + 2] + sz;
        }
    }
}

ORC_API int orc_num_procs(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

/* ====================================================================== list consumers
 * Ackland-Jones analysis                                  src/ackland_jones_analysis.cpp:9-172
 * rows must hold >= 14 neighbours sorted by distance (system.py:1620-1636) */
ORC_API int orc_aja(const double *x, const double *y, const double *z, int64_t n_atoms, const double *box9,
                    const double *origin, const int *boundary, const int *verlet, const double *dist, int64_t M,
                    int *aja, int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < n_atoms; ++i) {
        const double *di = dist + i * M;
        const int *vi = verlet + i * M;
        double r0 = 0.0;
        for (int j = 0; j < 6; ++j) r0 += di[j] * di[j];               /* :44-52 */
        r0 /= 6.0;
        const double c145 = 1.45 * r0, c155 = 1.55 * r0;
        int n0 = 0, n1 = 0;
        for (int j = 0; j < 14; ++j) {                                   /* :55-72 */
            const double r2 = di[j] * di[j];
            if (r2 < c155) { ++n1; if (r2 < c145) ++n0; }
        }
        int al[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const double xi = x[i], yi = y[i], zi = z[i];
        for (int j = 0; j < n0; ++j) {                                   /* :80-126: bond-angle histogram */
            double ax = x[vi[j]] - xi, ay = y[vi[j]] - yi, az = z[vi[j]] - zi;
            obox_pbc(&b, &ax, &ay, &az);
            for (int k = j + 1; k < n0; ++k) {
                double bx = x[vi[k]] - xi, by = y[vi[k]] - yi, bz = z[vi[k]] - zi;
                obox_pbc(&b, &bx, &by, &bz);
                const double c = (ax * bx + ay * by + az * bz) / (di[j] * di[k]);
                if (c < -0.945) al[0]++; else if (c < -0.915) al[1]++; else if (c < -0.755) al[2]++;
                else if (c < -0.195) al[3]++; else if (c < 0.195) al[4]++; else if (c < 0.245) al[5]++;
                else if (c < 0.795) al[6]++; else al[7]++;
            }
        }
        const double s_cp = fabs(1.0 - al[6] / 24.0);                    /* :129-150 */
        const int s56m4 = al[5] + al[6] - al[4];
        double s_bcc = s_cp + 1.0;
        if (s56m4 != 0) s_bcc = 0.35 * al[4] / (double)s56m4;
        double s_fcc = 0.61 * (abs(al[0] + al[1] - 6) + al[2]) / 6.0;
        double s_hcp = (fabs(al[0] - 3.0) + abs(al[0] + al[1] + al[2] + al[3] - 9)) / 12.0;
        if (al[0] == 7) s_bcc = 0.0; else if (al[0] == 6) s_fcc = 0.0; else if (al[0] <= 3) s_hcp = 0.0;
        int t;                                                           /* :152-170 */
        if (al[7] > 0) t = 0;
        else if (al[4] < 3) t = (n1 > 13 || n1 < 11) ? 0 : 4;
        else if (s_bcc <= s_cp) t = n1 < 11 ? 0 : 3;
        else if (n1 > 12 || n1 < 11) t = 0;
        else t = s_fcc < s_hcp ? 1 : 2;
        aja[i] = t;
    }
    return 0;
}

/* common neighbour parameter                                 src/common_neighbor_parameter.cpp:10-137 */
ORC_API int orc_cnp(const double *x, const double *y, const double *z, int64_t n_atoms, const double *box9,
                    const double *origin, const int *boundary, const int *verlet, const double *dist, const int *nn,
                    int64_t M, double *cnp, double rc, int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(dynamic, 64)
    for (int64_t i = 0; i < n_atoms; ++i) {
        int cnt = 0;
        double acc = 0.0;
        const int ni = nn[i];
        const int *vi = verlet + i * M;
        const double *di = dist + i * M;
        for (int m = 0; m < ni; ++m) {
            if (!(di[m] <= rc)) continue;                                /* :59 */
            const int j = vi[m];
            ++cnt;
            double rx = 0, ry = 0, rz = 0;
            const int nj = nn[j];
            const int *vj = verlet + (int64_t)j * M;
            const double *dj = dist + (int64_t)j * M;
            for (int s = 0; s < nj; ++s)
                for (int h = 0; h < ni; ++h)
                    if (vj[s] == vi[h]) {                                /* first match only, :83-120 */
                        if (dj[s] <= rc && di[h] <= rc) {
                            const int k = vj[s];
                            double ax = x[i] - x[k], ay = y[i] - y[k], az = z[i] - z[k];
                            double bx = x[j] - x[k], by = y[j] - y[k], bz = z[j] - z[k];
                            obox_pbc(&b, &ax, &ay, &az);
                            obox_pbc(&b, &bx, &by, &bz);
                            rx += ax + bx; ry += ay + by; rz += az + bz;
                        }
                        break;
                    }
            acc += rx * rx + ry * ry + rz * rz;
        }
        cnp[i] = cnt > 0 ? acc / cnt : 1000.0;                           /* :127-134 */
    }
    return 0;
}

/* structural (pair) entropy fingerprint                      src/structure_entropy.cpp:9-108 */
ORC_API int orc_structure_entropy(double rc, double sigma, int use_local_density, double volume, const double *dist,
                                  const int *nn, int64_t n_atoms, int64_t M, double *entropy, int num_t)
{
    const double PI = 3.14159265358979323846;
    const int nbins = (int)floor(rc / sigma) + 1;
    const double gd = n_atoms / volume;
    double *rl = (double *)malloc(sizeof(double) * (size_t)nbins), *rl2 = (double *)malloc(sizeof(double) * (size_t)nbins),
           *pre = (double *)malloc(sizeof(double) * (size_t)nbins);
    const double step = rc / (nbins - 1);
    const double factor = 4. * PI * gd * sqrt(2. * PI * sigma * sigma);
    for (int j = 0; j < nbins; ++j) { rl[j] = j * step; rl2[j] = rl[j] * rl[j]; pre[j] = rl2[j] * factor; }
    pre[0] = pre[1];
    const double s2 = sigma * sigma, lvol = 4. / 3. * PI * rc * rc * rc;
#pragma omp parallel num_threads(num_t > 0 ? num_t : 1)
    {
        double *g = (double *)malloc(sizeof(double) * (size_t)nbins);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n_atoms; ++i) {
            int nin = 0;
            for (int j = 0; j < nbins; ++j) {
                double a = 0.0;
                for (int k = 0; k < nn[i]; ++k) {
                    const double d = dist[i * M + k];
                    if (d <= rc) {
                        const double dl = rl[j] - d;
                        a += exp(-(dl * dl) / (2.0 * s2)) / pre[j];
                        if (j == 0) ++nin;
                    }
                }
                g[j] = a;
            }
            double density = gd;
            if (use_local_density) {
                density = nin / lvol;
                const double fac = gd / density;
                for (int j = 0; j < nbins; ++j) g[j] *= fac;
            }
            double prev = 0.0, sum = 0.0;
            for (int j = 0; j < nbins; ++j) {
                const double v = g[j] >= 1e-10 ? (g[j] * log(g[j]) - g[j] + 1.0) * rl2[j] : rl2[j];
                if (j > 0) sum += prev + v;
                prev = v;
            }
            entropy[i] = -PI * density * sum * sigma;
        }
        free(g);
    }
    free(rl); free(rl2); free(pre);
    return 0;
}

/* atomic temperature                                          src/atomic_temperature.cpp:9-112 */
ORC_API int orc_atomic_temperature(const int *verlet, const double *dist, int64_t n_atoms, int64_t M, const double *vx,
                                   const double *vy, const double *vz, const double *mass, double *T, double rc, int num_t)
{
    const double kb = 1.380649e-23, dim = 3.0, afu = 6.022140857e23;
    const double mass_factor = 1.0 / afu / 1000.0, vel_conv = 1e4;
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < n_atoms; ++i) {
        const int *vi = verlet + i * M;
        const double *di = dist + i * M;
        const double mi = mass[i];
        double sx = vx[i] * mi, sy = vy[i] * mi, sz = vz[i] * mi, ms = mi;
        int n = 1;
        for (int q = 0; q < M; ++q) {
            const int j = vi[q];
            if (j < 0) break;
            if (j != i && di[q] <= rc) { sx += vx[j] * mass[j]; sy += vy[j] * mass[j]; sz += vz[j] * mass[j]; ++n; ms += mass[j]; }
        }
        const double mx = sx / ms, my = sy / ms, mz = sz / ms;
        double dx = vx[i] - mx, dy = vy[i] - my, dz = vz[i] - mz;
        double ke = 0.0;
        ke += 0.5 * mi * mass_factor * (dx * dx + dy * dy + dz * dz) * vel_conv;
        for (int q = 0; q < M; ++q) {
            const int j = vi[q];
            if (j < 0) break;
            if (j != i && di[q] <= rc) {
                dx = vx[j] - mx; dy = vy[j] - my; dz = vz[j] - mz;
                ke += 0.5 * mass[j] * mass_factor * (dx * dx + dy * dy + dz * dz) * vel_conv;
            }
        }
        T[i] = ke * 2.0 / (dim * n * kb);
    }
    return 0;
}

/* cluster analysis: serial breadth-first flood fill            src/cluster.cpp:9-106 (get_cluster / get_cluster_by_bond) */
ORC_API int orc_cluster(const int *verlet, const double *dist, const int *nn, int64_t n_atoms, int64_t M, double rc,
                        int by_bond, int *cluster)
{
    int *queue = (int *)malloc(sizeof(int) * (size_t)(n_atoms > 0 ? n_atoms * 2 + 2 : 2));
    int cid = 0;
    for (int64_t seed = 0; seed < n_atoms; ++seed) {
        if (cluster[seed] != -1) continue;
        int64_t head = 0, tail = 0, cap = n_atoms * 2 + 2;
        queue[tail++] = (int)seed;
        ++cid;
        while (head < tail) {
            const int cur = queue[head++];
            int nl = 0;
            for (int j = 0; j < nn[cur]; ++j) {
                const int nb = verlet[(int64_t)cur * M + j];
                const int bond = by_bond ? (nb > -1) : (dist[(int64_t)cur * M + j] <= rc);
                if (bond) {
                    ++nl;
                    if (cluster[nb] == -1) { cluster[nb] = cid; if (tail < cap) queue[tail++] = nb; }
                }
            }
            if (nl == 0) cluster[cur] = cid;
        }
    }
    free(queue);
    return cid;
}

/* src/cluster.cpp:108-148 */
ORC_API int orc_filter_by_type(int *verlet, const double *dist, const int *nn, const int *type, int64_t n_atoms, int64_t M,
                               const int *t1, const int *t2, const double *r, int ntype)
{
    for (int64_t i = 0; i < n_atoms; ++i)
        for (int q = 0; q < nn[i]; ++q) {
            const int j = verlet[i * M + q];
            if (j < 0) continue;
            for (int k = 0; k < ntype; ++k)
                if (t1[k] == type[i] && t2[k] == type[j] && dist[i * M + q] > r[k]) verlet[i * M + q] = -1;
        }
    return 0;
}

/* FCC planar faults (stacking faults / twin boundaries among HCP-labelled atoms of an FCC crystal)
 *                                                              src/identify_fcc_planar_faults.cpp:9-245
 * fault: 0 non-hcp, 1 other, 2 intrinsic SF, 3 twin boundary, 4 multi-layer SF, 5 extrinsic SF */
static int pft_stacked(int a, int b, const int *hn)
{
    static const int basal[6] = {0, 1, 5, 6, 7, 8};
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j)
            if (hn[a * 12 + basal[i]] == hn[b * 12 + basal[j]]) return 0;
    return 1;
}
static int pft_bsearch(const int *arr, int n, int value)
{
    int left = 0, right = n - 1, mid = 0;
    while (left <= right) {
        mid = (left + right) / 2;
        if (value < arr[mid]) right = mid - 1;
        else if (value > arr[mid]) left = mid + 1;
        else break;
    }
    return mid;
}
ORC_API int orc_identify_sftb_fcc(const int *hcp_idx, int64_t n_hcp, int *hn, const int *ptm12, const int *stype,
                                  int64_t n_atoms, int *fault, int identify_esf)
{
    static const int layer_dir[12] = {0, 0, -1, -1, -1, 0, 0, 0, 0, 1, 1, 1};
    static const int basal[6] = {0, 1, 5, 6, 7, 8}, oop[6] = {2, 3, 4, 9, 10, 11};
    (void)n_atoms;
    for (int64_t i = 0; i < n_hcp; ++i)                                   /* :72-86 */
        for (int j = 0; j < 12; ++j) {
            const int b = ptm12[(int64_t)hcp_idx[i] * 12 + j];
            hn[i * 12 + j] = stype[b] == 2 ? pft_bsearch(hcp_idx, (int)n_hcp, b) : -b - 1;
        }
    for (int64_t i = 0; i < n_hcp; ++i) {                                 /* :88-137 */
        int nb = 0, np_ = 0, nn_ = 0, fp = 0, fn = 0;
        for (int j = 0; j < 12; ++j) {
            const int q = hn[i * 12 + j];
            if (q >= 0) {
                if (layer_dir[j] == 0) ++nb;
                else if (pft_stacked((int)i, q, hn)) { if (layer_dir[j] == 1) ++np_; else ++nn_; }
            } else if (layer_dir[j] != 0) {
                if (stype[-q - 1] == 1) { if (layer_dir[j] > 0) ++fp; else ++fn; }
            }
        }
        int f;
        if ((np_ != 0 && nn_ == 0) || (np_ == 0 && nn_ != 0)) f = 2;
        else if (nb >= 1 && np_ == 0 && nn_ == 0 && fp != 0 && fn != 0) f = 3;
        else if (np_ != 0 && nn_ != 0) f = 4;
        else f = 1;
        fault[hcp_idx[i]] = f;
    }
    for (int64_t i = 0; i < n_hcp; ++i) {                                 /* :139-180: serial, order dependent */
        const int a = hcp_idx[i];
        if (fault[a] == 3 || fault[a] == 1) {
            int nisf = 0, ntw = 0;
            for (int jj = 0; jj < 6; ++jj) {
                const int q = hn[i * 12 + basal[jj]];
                if (q >= 0) { const int nf = fault[hcp_idx[q]]; if (nf == 2) ++nisf; else if (nf == 3) ++ntw; }
            }
            if (nisf != 0 && ntw == 0) fault[a] = 2; else if (nisf == 0 && ntw != 0) fault[a] = 3;
        } else if (fault[a] == 4) {
            for (int jj = 0; jj < 6; ++jj) {
                const int q = hn[i * 12 + oop[jj]];
                if (q >= 0 && fault[hcp_idx[q]] == 2) fault[hcp_idx[q]] = 4;
            }
        }
    }
    if (!identify_esf) return 0;
    for (int64_t i = 0; i < n_hcp; ++i) {                                 /* :185-217 */
        const int a = hcp_idx[i];
        if (fault[a] != 3) continue;
        for (int j = 0; j < 12; ++j) {
            const int jn = ptm12[(int64_t)a * 12 + j];
            if (stype[jn] != 1) continue;
            int fc = 0, hc = 0;
            for (int k = 0; k < 12; ++k) {
                const int t = stype[ptm12[(int64_t)jn * 12 + k]];
                if (t == 1) ++fc; else if (t == 2) ++hc;
            }
            if (fc >= 5 && fc <= 6 && hc >= 5 && hc <= 6) { fault[a] = 5; break; }
        }
    }
    return 0;
}

/* _neighbor.filter_overlap_atom_with_grain                             src/neighbor.cpp:489-672, run with ONE thread
 * (the reference's parallel loop reads and sets the removal flags concurrently; its serial order is the defined one).
 * Brute force over pairs instead of the cell list: the cell list only prunes, it does not change which pairs interact. */
ORC_API int orc_filter_overlap_atom_with_grain(const double *x, const double *y, const double *z, const int *type, const int *grain,
                                               int64_t N, const double *box9, const double *origin, const int *boundary,
                                               double rc_mm, double rc_cc, double rc_mc, unsigned char *keep)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
    unsigned char *removed = (unsigned char *)calloc((size_t)(N > 0 ? N : 1), 1);
    const double q_mm = rc_mm * rc_mm, q_cc = rc_cc * rc_cc, q_mc = rc_mc * rc_mc;
    const int anypbc = boundary[0] || boundary[1] || boundary[2];
    for (int64_t i = 0; i < N; ++i) {
        if (removed[i]) continue; /* :548-551 */
        double xi = x[i], yi = y[i], zi = z[i];
        if (anypbc) obox_wrap(&b, &xi, &yi, &zi); /* :557-560 */
        for (int64_t j = i + 1; j < N; ++j) {
            if (removed[j]) continue;
            double dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
            obox_pbc(&b, &dx, &dy, &dz);
            const double d2 = dx * dx + dy * dy + dz * dz;
            int64_t target = -1;
            if (type[i] == 1 && type[j] == 1) {
                if (d2 <= q_mm) target = j;
            } else if (type[i] == 2 && type[j] == 2) {
                if (d2 <= q_cc) target = (grain[i] != grain[j]) ? ((grain[i] > grain[j]) ? i : j) : j;
            } else if (type[i] != type[j]) {
                if (d2 <= q_mc) target = (type[i] == 1) ? i : j;
            }
            if (target >= 0) removed[target] = 1;
        }
    }
    for (int64_t i = 0; i < N; ++i) keep[i] = removed[i] ? 0 : 1;
    free(removed);
    return 0;
}

/* _polycrystal.transform_and_filter                                  src/polycrystal.cpp:20-125
 * out (n,3) capacity; returns the number of survivors (input order) */
ORC_API int64_t orc_transform_and_filter(const double *x, const double *y, const double *z, int64_t n, const double *R,
                                         const double *center, const double *target, const double *coeffs, int nf, double *out)
{
    const double RT00 = R[0], RT01 = R[3], RT02 = R[6]; /* RTck = R(k,c), :49-51 */
    const double RT10 = R[1], RT11 = R[4], RT12 = R[7];
    const double RT20 = R[2], RT21 = R[5], RT22 = R[8];
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double dx = x[i] - center[0], dy = y[i] - center[1], dz = z[i] - center[2];
        const double px = dx * RT00 + dy * RT10 + dz * RT20 + target[0];
        const double py = dx * RT01 + dy * RT11 + dz * RT21 + target[1];
        const double pz = dx * RT02 + dy * RT12 + dz * RT22 + target[2];
        int inside = 1;
        for (int f = 0; f < nf; ++f) {
            const double *pl = coeffs + 4 * f;
            const double val = px * pl[0] + py * pl[1] + pz * pl[2] + pl[3];
            if (val >= 0.0) { inside = 0; break; }
        }
        if (inside) { out[3 * cnt] = px; out[3 * cnt + 1] = py; out[3 * cnt + 2] = pz; ++cnt; }
    }
    return cnt;
}

/* overlap filter of the polycrystal builder                   src/neighbor.cpp:390-486
 * keep[j] = 0 iff a centre i < j finds j within rc (centre wrapped, j raw, minimum image) */
ORC_API int orc_filter_overlap_atom(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                    const double *origin, const int *boundary, double rc, unsigned char *keep, int num_t)
{
    obox b;
    if (obox_init(&b, box9, origin, boundary))
        return -1;
    ogrid g;
    int rcode = ogrid_build(&g, &b, rc, x, y, z, N);
    if (rcode)
        return rcode;
    const double rc_inv = 1.0 / rc, rcsq = rc * rc;
    for (int64_t i = 0; i < N; ++i) keep[i] = 1;
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        double xi, yi, zi;
        int c[3];
        center_of(&b, x, y, z, i, &xi, &yi, &zi);
        cell_of(&b, rc_inv, g.nc, xi, yi, zi, c);
        for (int a = c[0] - 1; a <= c[0] + 1; ++a)
            for (int bb = c[1] - 1; bb <= c[1] + 1; ++bb)
                for (int cc = c[2] - 1; cc <= c[2] + 1; ++cc) {
                    int64_t cell = ((int64_t)pmod(a, g.nc[0]) * g.nc[1] + pmod(bb, g.nc[1])) * g.nc[2] + pmod(cc, g.nc[2]);
                    for (int64_t p = g.start[cell]; p < g.start[cell + 1]; ++p) {
                        int j = g.atoms[p];
                        if (j <= i)
                            continue;                                  /* :448 only the higher index is marked */
                        double dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
                        obox_pbc(&b, &dx, &dy, &dz);
                        if (dx * dx + dy * dy + dz * dz <= rcsq)
                            keep[j] = 0;                               /* benign race: every writer stores 0 */
                    }
                }
    }
    ogrid_free(&g);
    return 0;
}

/* ====================================================================== static structure factor, direct summation
 *                                                                          src/structure_factor.cpp:64-447 (total / cross)
 *                                                                          src/structure_factor.cpp:451-640 (all partials)
 * The two classes enumerate the reciprocal-lattice points differently (:122-200 vs :595-630); both are restated. */
typedef struct { double x, y, z; } sfc_v3;
static void sfc_recip(const double *h, sfc_v3 *b)                          /* :66-103 */
{
    const double *a = h, *bb = h + 3, *c = h + 6;
    const double vol = a[0] * (bb[1] * c[2] - bb[2] * c[1]) - a[1] * (bb[0] * c[2] - bb[2] * c[0]) + a[2] * (bb[0] * c[1] - bb[1] * c[0]);
    const double tp = 2.0 * 3.14159265358979323846;
    b[0].x = (bb[1] * c[2] - bb[2] * c[1]) / vol * tp; b[0].y = (bb[2] * c[0] - bb[0] * c[2]) / vol * tp; b[0].z = (bb[0] * c[1] - bb[1] * c[0]) / vol * tp;
    b[1].x = (c[1] * a[2] - c[2] * a[1]) / vol * tp;   b[1].y = (c[2] * a[0] - c[0] * a[2]) / vol * tp;   b[1].z = (c[0] * a[1] - c[1] * a[0]) / vol * tp;
    b[2].x = (a[1] * bb[2] - a[2] * bb[1]) / vol * tp; b[2].y = (a[2] * bb[0] - a[0] * bb[2]) / vol * tp; b[2].z = (a[0] * bb[1] - a[1] * bb[0]) / vol * tp;
}
static double sfc_norm(sfc_v3 v) { return sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
static double sfc_dot(sfc_v3 a, sfc_v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* k-points of StructureFactorDirect::generateKPoints (:105-200); returns the count, fills kp (3 doubles each) if not NULL */
ORC_API int64_t orc_sfc_kpoints_total(const double *box9, double k_max, double k_min, double *kp)
{
    const double tp = 2.0 * 3.14159265358979323846;
    sfc_v3 b[3];
    sfc_recip(box9, b);
    const double q_max = k_max / tp, q_min = k_min / tp, q_max_sq = q_max * q_max, q_min_sq = q_min * q_min;
    const int nkx = (int)ceil(q_max / (sfc_norm(b[0]) / tp)), nky = (int)ceil(q_max / (sfc_norm(b[1]) / tp)), nkz = (int)ceil(q_max / (sfc_norm(b[2]) / tp));
    int64_t n = 0;
    for (int i = 0; i < nkx; ++i) {
        const sfc_v3 kx = {b[0].x * i, b[0].y * i, b[0].z * i};
        for (int ky = 0; ky < nky; ++ky) {
            const sfc_v3 kxy = {kx.x + b[1].x * ky, kx.y + b[1].y * ky, kx.z + b[1].z * ky};
            const double ca = sfc_dot(b[2], b[2]), cb = -2.0 * sfc_dot(kxy, b[2]);
            const double cmin = sfc_dot(kxy, kxy) - k_min * k_min, cmax = sfc_dot(kxy, kxy) - k_max * k_max;
            const double b2a = cb / (2.0 * ca), dmin = b2a * b2a - cmin / ca, dmax = b2a * b2a - cmax / ca;
            if (dmax < 0) continue;
            const double zmin = dmin < 0 ? 0.0 : -b2a + sqrt(dmin), zmax = -b2a + sqrt(dmax);
            int kz0 = (int)floor(zmin), kz1 = (int)ceil(zmax);
            if (kz0 < 0) kz0 = 0;
            if (kz1 > nkz - 1) kz1 = nkz - 1;
            for (int kz = kz0; kz <= kz1; ++kz) {
                const sfc_v3 k = {kxy.x + b[2].x * kz, kxy.y + b[2].y * kz, kxy.z + b[2].z * kz};
                const double qd = sfc_dot(k, k) / (tp * tp);
                if (qd <= q_max_sq && qd >= q_min_sq) {
                    if (kp) { kp[3 * n] = k.x; kp[3 * n + 1] = k.y; kp[3 * n + 2] = k.z; }
                    ++n;
                }
            }
        }
    }
    return n;
}
/* k-points of StructureFactorDirectPartial::helper_kpoints (:595-630) */
ORC_API int64_t orc_sfc_kpoints_partial(const double *box9, double k_max, double k_min, double *kp)
{
    const double tp = 2.0 * 3.14159265358979323846;
    sfc_v3 b[3];
    sfc_recip(box9, b);
    const double q_max = k_max / tp;
    const int nkx = (int)ceil(q_max / (sfc_norm(b[0]) / tp)), nky = (int)ceil(q_max / (sfc_norm(b[1]) / tp)), nkz = (int)ceil(q_max / (sfc_norm(b[2]) / tp));
    int64_t n = 0;
    for (int i = 0; i < nkx; ++i)
        for (int j = 0; j < nky; ++j)
            for (int k = 0; k < nkz; ++k) {
                const sfc_v3 kv = {b[0].x * i + b[1].x * j + b[2].x * k, b[0].y * i + b[1].y * j + b[2].y * k, b[0].z * i + b[1].z * j + b[2].z * k};
                /* the reference forms (bx*i + by*j) + bz*k: same association */
                const double mag = sfc_norm(kv);
                if (mag > k_min && mag <= k_max) {
                    if (kp) { kp[3 * n] = kv.x; kp[3 * n + 1] = kv.y; kp[3 * n + 2] = kv.z; }
                    ++n;
                }
            }
    return n;
}
static int sfc_bin(double k, double k_min, double k_max, int bins)        /* :283-291 */
{
    if (k < k_min || k >= k_max) return -1;
    int b = (int)((k - k_min) / (k_max - k_min) * bins);
    return b < bins - 1 ? b : bins - 1;
}
/* total (qx == NULL) or cross structure factor between the point sets (:293-447) */
ORC_API int orc_sfc_direct(const double *x, const double *y, const double *z, int64_t n, const double *box9, double *sf, int bins,
                           double k_max, double k_min, const double *qx, const double *qy, const double *qz, int64_t nq,
                           unsigned n_total, int num_t)
{
    const int64_t nk = orc_sfc_kpoints_total(box9, k_max, k_min, NULL);
    if (nk <= 0) return -2;
    double *kp = (double *)malloc(sizeof(double) * 3 * (size_t)nk);
    orc_sfc_kpoints_total(box9, k_max, k_min, kp);
    if (n_total == 0) n_total = (unsigned)n;
    const double norm = 1.0 / sqrt((double)n_total);
    double *sk = (double *)malloc(sizeof(double) * (size_t)nk);
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(dynamic)
    for (int64_t k = 0; k < nk; ++k) {
        double ar = 0, ai = 0, br = 0, bi = 0;
        for (int64_t i = 0; i < n; ++i) { const double a = kp[3 * k] * x[i] + kp[3 * k + 1] * y[i] + kp[3 * k + 2] * z[i]; ar += cos(a); ai += sin(a); }
        ar *= norm; ai *= norm;
        if (qx) {
            for (int64_t i = 0; i < nq; ++i) { const double a = kp[3 * k] * qx[i] + kp[3 * k + 1] * qy[i] + kp[3 * k + 2] * qz[i]; br += cos(a); bi += sin(a); }
            br *= norm; bi *= norm;
        } else { br = ar; bi = ai; }
        sk[k] = ar * br + ai * bi;                                          /* Re(conj(F1) F2) */
    }
    unsigned *cnt = (unsigned *)calloc((size_t)bins, sizeof(unsigned));
    for (int b = 0; b < bins; ++b) sf[b] = 0.0;
    for (int64_t k = 0; k < nk; ++k) {
        const double mag = sqrt(kp[3 * k] * kp[3 * k] + kp[3 * k + 1] * kp[3 * k + 1] + kp[3 * k + 2] * kp[3 * k + 2]);
        const int b = sfc_bin(mag, k_min, k_max, bins);
        if (b >= 0) { sf[b] += sk[k]; cnt[b]++; }
    }
    for (int b = 0; b < bins; ++b) sf[b] = cnt[b] ? sf[b] / cnt[b] : NAN;
    free(kp); free(sk); free(cnt);
    return 0;
}
/* all Ashcroft-Langreth partials, out (ntype, ntype, bins) (:451-560) */
ORC_API int orc_sfc_direct_partial(const double *x, const double *y, const double *z, const int *type, int ntype, int64_t n,
                                   const double *box9, double *out, int bins, double k_max, double k_min, int num_t)
{
    const int64_t nk = orc_sfc_kpoints_partial(box9, k_max, k_min, NULL);
    if (nk <= 0) return -2;
    double *kp = (double *)malloc(sizeof(double) * 3 * (size_t)nk);
    orc_sfc_kpoints_partial(box9, k_max, k_min, kp);
    double *fr = (double *)calloc((size_t)ntype * nk, sizeof(double)), *fi = (double *)calloc((size_t)ntype * nk, sizeof(double));
    const double norm = 1.0 / sqrt((double)n);
#pragma omp parallel for num_threads(num_t > 0 ? num_t : 1) schedule(dynamic)
    for (int64_t k = 0; k < nk; ++k) {
        double ar[64] = {0}, ai[64] = {0};
        for (int64_t i = 0; i < n; ++i) { const double a = kp[3 * k] * x[i] + kp[3 * k + 1] * y[i] + kp[3 * k + 2] * z[i]; ar[type[i]] += cos(a); ai[type[i]] += sin(a); }
        for (int t = 0; t < ntype; ++t) { fr[(size_t)t * nk + k] = ar[t] * norm; fi[(size_t)t * nk + k] = ai[t] * norm; }
    }
    unsigned *cnt = (unsigned *)calloc((size_t)bins, sizeof(unsigned));
    int *bin = (int *)malloc(sizeof(int) * (size_t)nk);
    for (int64_t k = 0; k < nk; ++k) {
        const double mag = sqrt(kp[3 * k] * kp[3 * k] + kp[3 * k + 1] * kp[3 * k + 1] + kp[3 * k + 2] * kp[3 * k + 2]);
        bin[k] = sfc_bin(mag, k_min, k_max, bins);
        if (bin[k] >= 0) cnt[bin[k]]++;
    }
    for (int a = 0; a < ntype; ++a)
        for (int b = a; b < ntype; ++b) {
            double *loc = (double *)calloc((size_t)bins, sizeof(double));
            for (int64_t k = 0; k < nk; ++k)
                if (bin[k] >= 0) loc[bin[k]] += fr[(size_t)a * nk + k] * fr[(size_t)b * nk + k] + fi[(size_t)a * nk + k] * fi[(size_t)b * nk + k];
            for (int q = 0; q < bins; ++q) {
                const double v = cnt[q] ? loc[q] / cnt[q] : NAN;
                out[((size_t)a * ntype + b) * bins + q] = v;
                out[((size_t)b * ntype + a) * bins + q] = v;
            }
            free(loc);
        }
    free(kp); free(fr); free(fi); free(cnt); free(bin);
    return 0;
}
