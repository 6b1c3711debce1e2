/*
 * mdapy_amd.h — C ABI of libmdapy_amd.so: MI355X (gfx950) HIP implementation of
 * mdapy's neighbor-list + per-atom structural-analysis hot path.
 *
 * This is the drop-in boundary: each entry point replaces one
 * function of one of the reference's nanobind extension modules
 * (mushroomfire/mdapy, CMakeLists.txt:71-100).  The reference interface that
 * each one replaces is cited as  file:line  relative to the reference tree.
 *
 * Conventions (mirroring SURVEY.md §8b):
 *   - plain pointers + sizes, no torch / numpy types;
 *   - `int` is int32, `double` is IEEE binary64, index products are 64-bit;
 *   - box9 = 3x3 row-major box matrix (rows a,b,c), origin3, boundary3 (0/1) are
 *     ALWAYS host pointers (they are 15 numbers);
 *   - every other array pointer of one call lives in the SAME memory space,
 *     given by `space`: MDH_HOST (pageable/pinned host memory: the library
 *     stages through HBM and returns when the results are back on the host) or
 *     MDH_DEVICE (HBM of the current HIP device, e.g. torch tensor.data_ptr():
 *     the call is enqueued on `stream` and returns without synchronising unless
 *     it has to hand a scalar back to the host);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - outputs are caller-allocated and, unless stated otherwise, caller
 *     initialised exactly as the reference's Python layer initialises them;
 *   - return value 0 = success; <0 = error, text via mdh_last_error().
 *     MDH_ERR_BOX corresponds to the reference's only C++ throw on this path
 *     (src/box.h:185-186 "The volume of the box is zero.") and is surfaced as
 *     RuntimeError by the Python layer, like nanobind does.
 */
#ifndef MDAPY_AMD_H
#define MDAPY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDH_HOST 0
#define MDH_DEVICE 1

#define MDH_OK 0
#define MDH_ERR_BOX (-1)   /* singular triclinic box                       */
#define MDH_ERR_HIP (-2)   /* a HIP runtime call failed / no gfx950 device */
#define MDH_ERR_ARG (-3)   /* invalid argument                             */
#define MDH_ERR_NOMEM (-4) /* device allocation failed                     */

/* ---- runtime ---------------------------------------------------------- */
const char *mdh_last_error(void);
int mdh_version(void);
/* number of visible HIP devices (0 when there is no GPU; never an error) */
int mdh_device_count(void);
/* select the device of this thread's later calls and load the library's code objects on it (mdh_warm) */
int mdh_set_device(int device);
/*
 * Load every code object of the library on the current device, once per device and process (an empty launch per
 * translation unit; ~30-100 ms in a fresh process): afterwards the FIRST mdh_build_neighbor / mdh_csp / ... of the
 * process costs about what the following ones do.  The reference has no counterpart: a CPU extension module is
 * mapped as a whole at import (CMakeLists.txt:71-100).  The library calls it by itself at its first compute
 * call on a device (MDAPY_HIP_WARM=0 in the environment switches that off) and in mdh_set_device; a binding may call it
 * earlier.  No-op when the device has been warmed before.
 */
int mdh_warm(void);
/*
 * min_max2[0] / [1] (host memory) = smallest / largest entry of an int32 array of n > 0 entries; waits for the stream.
 * Host policy of src/mdapy/system.py:1262-1263, 1987-1988 ("does every atom have k neighbours in the list I hold?" is
 * `neighbor_number.min() >= k` there, a numpy reduction): with the counts resident in HBM the reduction runs there.
 */
int mdh_min_max_i32(const int *v, int64_t n, int *min_max2, int space, void *stream);
/*
 * Dense species codes of a numeric label column (host policy of src/mdapy/radial_distribution_function.py:83-110 and
 * warren_cowley_parameter.py:76-112: `type - 1` / element -> sorted index, a numpy pass over N labels there): every value of
 * [low, low + span), span <= 4096, that occurs gets the next code in ascending order; codes[i] = code of v[i] (0 for a value
 * outside the range), counts_span[k] (host memory, span entries) = how often low + k occurs.  Waits for the stream.
 */
int mdh_dense_codes_i32(const int *v, int64_t n, int low, int span, int *codes, int64_t *counts_span, int space, void *stream);
/* free the cached per-device scratch buffers */
int mdh_release_workspace(void);
/* bytes currently held by the scratch cache of the current device */
int64_t mdh_workspace_bytes(void);
/*
 * Per-kernel timing with HIP events recorded on the stream each kernel is launched on
 * (used by bench.py for the roofline figure).  mdh_prof_enable(1) starts collecting (2: the
 * range "k_neighbor" only — an event pair costs ~8 us of stream time per range and call),
 * mdh_prof_reset() drops what was collected, mdh_prof_report() synchronises the recorded
 * events and writes lines "name count total_ms\n" into buf (returns bytes written, <0 on error).
 */
int mdh_prof_enable(int on);
int mdh_prof_reset(void);
int mdh_prof_report(char *buf, int buflen);
/* A/B switch for measurements and tests: 0 = automatic kernel choice (default), 1 = force the
 * thread-per-atom neighbor kernel, 2 = force the round-1 LDS-tiled kernel where it applies.  Results are identical. */
int mdh_debug_set_neighbor_variant(int variant);
/* A/B switch for measurements and tests: 1 (default; MDH_INDIRECT in the environment) = a neighbor build of input that comes in
 * some spatial order keeps no cell-sorted copy of the atoms, its kernels read them through the cell-sorted id list;
 * 0 = the cell-sorted 32-byte records always (what unordered input gets either way).  Returns the previous value.  Results are identical. */
int mdh_debug_set_indirect(int on);
/* test hook: the tile plan of the last neighbor build that took the LDS-tile kernel (neighbor_lane.hip):
 * plan8 = {tile cells in x/y, in z, halo atoms per tile, LDS bytes, box full of atoms, 1000 * atoms per cell,
 * cells of the occupied region, 1 if a plan was made since the last query}. */
int mdh_debug_neighbor_plan(int *plan8);
/* test hook: fixed-cutoff CNA (mdh_fcna, src/cna.cpp:429-506): 0 = single-precision pair tests with a decision band where the
 * box allows (default; atoms inside the band are finished with the reference's double-precision expression), 1 = the
 * double-precision kernel everywhere.  Labels are identical. */
int mdh_debug_set_fcna_variant(int variant);
/* measurement hook (bench.py `extra.*.todo_fraction`): mdh_debug_track_counters(1) makes every mdh_fcna copy the length of its
 * to-do list (atoms with a pair inside the single-precision decision band, finished in double precision) into pinned memory
 * (one 4-byte copy per call, off by default); mdh_debug_counters(out4) — after the caller synchronised the stream — returns
 * {to-do atoms of the last tracked mdh_fcna or -1, tiles the previous neighbor build with the last plan's (N, grid) listed for
 * the one-cell-slice pass or -1, the "image codes not valid" flag of the last tracked neighbor build (1: an orthogonal box with atoms
 * more than 14 box lengths outside it — the thread-per-atom kernel took the call; 0: the tile kernel did where the box and the
 * grid allow one; -1: none tracked), passes of the last Voronoi call that went over the atoms with a still-open cell only}. */
int mdh_debug_track_counters(int on);
int mdh_debug_counters(int64_t *out4);
/* test hook: 0 = LDS-tile kernel for the streaming RDF where it applies (default), 1 = thread-per-atom kernel everywhere */
int mdh_debug_set_rdf_variant(int variant);
/* k nearest neighbours: 0 = the near kernel (sorted list in registers, 27 cells) followed by the general kernel on the queries it
 * lists; 1 = the general kernel for every query (A/B measurements, tests; results identical) */
int mdh_debug_set_knn_variant(int variant);
/* test hook: 0 = per-degree register-resident stage 1 of the Steinhardt parameters where compiled (default), 1 = generic kernel */
int mdh_debug_set_sq_variant(int variant);
/* test hook: structure entropy, 0 = the ladder kernel where it applies (<= 40 bins, rc^2 / 2 sigma^2 <= 640) with the lane count
 * picked by the row width (default), 1 = the direct kernel (an exponential per term), 2 / 3 / 4 / 5 = the ladder with 1 / 2 / 4 / 8
 * lanes to a row */
int mdh_debug_set_entropy_variant(int variant);
/* test hook: vertex capacity of the first pass of the PTM neighbour ordering (10 default, 15; 5 sends most atoms through
 * the second, 28-vertex pass, whose results must be the same) */
int mdh_debug_set_ptm_order_cap(int cap);
/* test hook: out4[k] = smallest double d with floor(d/L + 0.5) >= k-1 (k = 0..3) for a periodic orthogonal
 * axis of length L — the exact decision points that let the kernels replace floor(d/L+0.5) by compares. */
int mdh_debug_image_thresholds(double L, double *out4);

/* ---- input side: the atom table of a text file ---------------------------- */
/*
 * replaces the per-atom block parsing of BuildSystem.read_dump / read_xyz:
 *   src/mdapy/load_save.py:141-175 (dump rows -> typed columns), :825-845 (extended-XYZ rows)
 * `text` = nbytes of rows separated by '\n' (fields separated by blanks / tabs; '\r' ignored), in host memory or in
 * HBM (text_space); the first nrows rows are parsed, extra fields of a row are ignored as the reference's
 * row.split()[:ncol] does.  kinds[c]: 0 float64, 1 int32, 2 up to 8 characters packed little-endian into an int64
 * (element / typelabel columns), 3 skip.  columns[c] = array of nrows elements of that type (ignored for kind 3).
 * Floating-point fields are converted with correct rounding (== Python float(), what the reference's readers
 * produce).  Fields the device cannot decide — more than 19 significant digits whose truncation straddles a rounding
 * boundary, nan / inf, integers written as 3.0, strings longer than 8 bytes — are reported in redo (2 * redo_cap
 * int64): redo[2k] = row * ncol + column, redo[2k+1] = (byte offset of the field << 16) | its length, for
 * k < min(status4[0], redo_cap); the caller converts those on the host.
 * status4: [0] fields to redo, [1] rows missing or with fewer than ncol fields, [2] fields that are no numbers,
 * [3] rows present in the text.
 */
int mdh_parse_table(const char *text, int64_t nbytes, int text_space, int64_t nrows, int ncol, const int *kinds,
                    void *const *columns, int64_t *redo, int64_t redo_cap, int64_t *status4, int space, void *stream);
/* test hooks (run on the host, no GPU needed): entry q (-342..308) of the generated 128-bit table of powers of five
 * (high, low word); the field converter itself: 0 converted, 1 undecided (host must redo), 2 not a number */
int mdh_debug_text_pow5(int q, uint64_t *out2);
int mdh_debug_parse_double(const char *s, int64_t len, double *out);

/* ---- _neighbor -------------------------------------------------------- */
/*
 * replaces _neighbor.build_neighbor                       src/neighbor.cpp:351-388
 * (cell build :64-100 + build_verlet_list :102-187).
 * verlet (N,max_neigh) int32, dist (N,max_neigh) f64, nn (N) int32.
 * fill_pads == 0: reference semantics — only the first min(nn[i],max_neigh)
 *   slots of a row are written; the caller has pre-filled -1 / rc+1
 *   (src/mdapy/neighbor.py:125-129).
 * fill_pads != 0: the kernel also writes the pads (-1, rc+1.0) so that the
 *   caller may hand over uninitialised memory (saves one full pass over HBM).
 * nn[i] keeps counting past max_neigh (neighbor.cpp:172-177).
 */
int mdh_build_neighbor(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                       const double *origin3, const int *boundary3, double rc, int *verlet, double *dist, int *nn,
                       int64_t max_neigh, int fill_pads, int space, void *stream);
/* The same with an ordering key (multi-GPU extension): inside a cell the atoms are listed by descending key[i]
 * (i64, N) instead of descending index — the reference's rule (neighbor.cpp:97-98) applied to the GLOBAL atom ids of a
 * slab's owned + ghost atoms, so that every row equals the row of the undivided system.  key == NULL: mdh_build_neighbor.
 * ABSENT atoms (extension, all mdh_build_neighbor* entries): an atom whose x is NaN takes no cell — it appears in nobody's row, and
 * its own row and count are NOT written (the unused slots of a decomposed step's fixed-size ghost block,
 * mdh_slab_append_ghosts_static; the reference has no meaning for NaN input). */
int mdh_build_neighbor_keyed(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                             const double *origin3, const int *boundary3, double rc, int *verlet, double *dist, int *nn,
                             int64_t max_neigh, int fill_pads, const int64_t *key, int space, void *stream);

/* mdh_build_neighbor followed by mdh_fcna(rc) — what System.cal_common_neighbor_analysis(rc, max_neigh) runs when it has no list
 * yet (neighbor.cpp:351 then cna.cpp:429-506) — as ONE pass over the LDS tiles: a centre's 12 or 14 neighbours are still staged
 * when its row is written, so the bond matrix is taken from LDS (single-precision pair tests on the tile's staged coordinates with
 * the scan's own decision band; a pair inside the band sends the atom to a to-do list that the double-precision kernel of mdh_fcna
 * finishes) instead of 12-14 gathers per atom from HBM and a second read of the rows.
 * verlet / dist / nn exactly as mdh_build_neighbor leaves them; pattern (N) i32 exactly as mdh_fcna leaves it
 * (caller-initialised; atoms without 12 or 14 neighbours keep their value) — bit for bit, tests/test_gpu_parity.py
 * test_fused_*.  key: as in mdh_build_neighbor_keyed, or NULL.  Where the tile kernel does not apply the two steps run one after
 * the other inside the call.
 * Measured on MI355X (10 061 824-atom fcc Cu, M = 16, profiles/r05_fused_ab.txt): 1.51 ms against 1.71 ms for the two calls
 * (0.88; 0.87 - 0.96 on lattices rattled by 0.05 - 0.3 A).  The form of rounds 2-4, with double-precision pair tests on the raw
 * coordinates (175 VGPRs, three workgroups per CU, scratch memory), took 2.9 ms and was not used; this one keeps the 124 VGPRs and
 * four workgroups per CU of the plain instance.  The package calls it wherever the reference would build a list and label from it. */
int mdh_build_neighbor_fcna(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                            const double *origin3, const int *boundary3, double rc, int *verlet, double *dist, int *nn,
                            int64_t max_neigh, int fill_pads, int *pattern, const int64_t *key, int space, void *stream);

/* Decomposed systems (multi-GPU extension): a promise that every atom of the NEXT neighbor build of this thread —
 * a rank's slab and its halo, in the global box — has its wrapped fractional coordinate along `axis` in [frac_lo, frac_hi]
 * (the interval may leave [0,1): a slab at the periodic seam).  The passes over ALL cells of the global grid then run over the
 * window's planes only; results are the same.  Only axis 0 of orthogonal boxes is taken (otherwise ignored).  An atom outside
 * the window breaks the promise: detected on the device, the atom is left out of the grid (nothing is written out of bounds),
 * and the broken promise is reported as MDH_ERR_ARG by this thread's next build or by mdh_cell_window_check(). */
int mdh_hint_cell_window(int axis, double frac_lo, double frac_hi);
/* With a cell window: the stretch [frac_lo, frac_hi) of the same axis that holds the atoms whose rows the caller WANTS (a rank's own
 * slab; the rest of the window is its ghost halo).  The next neighbor build of this thread lists the ghosts as neighbours but makes
 * no rows for atoms binned outside the stretch's planes: their counts, rows and labels are left as the
 * caller's buffers held them — pre-zero the counts.  The reference has no counterpart (src/mdapy/parallel.py:1-53 is OpenMP thread
 * control); rows of the atoms inside the stretch are those of mdh_build_neighbor (src/neighbor.cpp:102-187).  Axis 0 of orthogonal boxes. */
int mdh_hint_centre_window(int axis, double frac_lo, double frac_hi);
/* Waits for `stream` and returns MDH_ERR_ARG if the last windowed build of this thread found atoms outside its window (the
 * build itself stays memory-safe: such atoms take no slot and are left out, so its rows are incomplete); MDH_OK otherwise.
 * For callers that want the CURRENT step to fail instead of the next build (one stream synchronisation). */
int mdh_cell_window_check(void *stream);

/* Halo selection of the slab decomposition (multi-GPU extension): one pass over the owned atoms; up / down (n) i32
 * receive the indices of the atoms whose wrapped fractional coordinate f along the decomposed axis (hi3 = that column of the
 * inverse box) satisfies f >= up_from / f < down_below, counts_host[2] their numbers.  Order of the indices: unspecified.
 * gid (n) i64 + up_pack / down_pack (capacity,4) f64 (or all NULL): rows (x, y, z, id) of the selected atoms, in the same order.
 * up / down hold `capacity` entries; a count above it says that the buffers were too small (only the first `capacity`
 * selections were stored): call again with larger ones. */
int mdh_slab_halo_select(const double *x, const double *y, const double *z, int64_t n, const double *origin3_host,
                         const double *hi3_host, double up_from, double down_below, int *up, int *down, int64_t *counts_host,
                         const int64_t *gid, double *up_pack, double *down_pack, int64_t capacity, int space, void *stream);
/* The same selection packed into the two messages of the halo exchange, without a word to the host: msg_up / msg_down are
 * device buffers of 1 + (4 + nextra) * cap doubles — [0] the number of selected atoms, then rows x, y, z, the nextra (<= 4)
 * extra f64 columns, the global id, `cap` columns each.  A count above cap: the message holds the first cap atoms only
 * (both ends see it in the header).  Device memory only; nothing synchronises. */
int mdh_slab_halo_messages(const double *x, const double *y, const double *z, int64_t n, const double *origin3_host,
                           const double *hi3_host, double up_from, double down_below, const int64_t *gid,
                           const double *const *extras_host_array_of_device_pointers, int nextra, double *msg_up,
                           double *msg_down, int64_t cap, void *stream);
/* The receiving side: the nl + nr atoms of the two incoming messages (same layout, ncol = 3 + nextra f64 columns) are written
 * behind the owned atoms, columns[k][n_owned ...] and gid[n_owned ...] (the id row converted back to i64), the left
 * neighbour's atoms first.  One launch; the caller has read nl and nr from the message headers.  Device memory only. */
int mdh_slab_append_ghosts(const double *msg_from_left, const double *msg_from_right, int64_t cap, int64_t nl, int64_t nr,
                           double *const *columns_host_array_of_device_pointers, int ncol, int64_t *gid, int64_t n_owned,
                           void *stream);
/* The receiving side with NOTHING read by the host (the decomposed step's fast path): the ghost block behind the owned atoms has
 * the fixed size 2 cap — the left message's atoms at n_owned + [0, cap), the right message's at n_owned + cap + [0, cap), their
 * counts taken from the message headers on the device; the slots behind a message's atoms are marked ABSENT (x = NaN, y = z =
 * extras = 0, id = -1).  An absent atom is given no cell by the cell-grid build of mdh_build_neighbor* (it appears in no row and
 * its own row is not written: the caller zeroes the counts array).  A header above cap raises a flag that
 * mdh_slab_overflow_check() reports (MDH_ERR_ARG, flag cleared) once that append has run on the device.  No reference
 * counterpart (multi-GPU extension). */
int mdh_slab_append_ghosts_static(const double *msg_from_left, const double *msg_from_right, int64_t cap,
                                  double *const *columns_host_array_of_device_pointers, int ncol, int64_t *gid, int64_t n_owned,
                                  void *stream);
int mdh_slab_overflow_check(void);

/* ---- atom order (csrc/order.hip).  The reference's linked-list cell build (src/neighbor.cpp:64-100) and its consumers are
 * indifferent to the order in which atoms arrive; these kernels' gathers are not.  No reference counterpart: the host layer
 * (mdapy_amd/system.py) uses the four entries to keep a cell-sorted copy of a system that was handed in in no spatial order.
 *
 * mdh_order_statistic: *far_fraction (host) = fraction of consecutive atoms (i, i+1) that do not lie in the same or in touching
 * bins of a grid of ~64-atom bins.  ~0 for a lattice builder's order, a file written cell by cell or a sorted system, ~1 for a
 * shuffled one.  Synchronises `stream`. */
int mdh_order_statistic(const double *x, const double *y, const double *z, int64_t N, const double *box9, const double *origin3,
                        const int *boundary3, double *far_fraction, int space, void *stream);
/* mdh_spatial_sort: perm (N) i32 = the atoms in cell order (cells of 2.5 atoms on average, the walk of src/neighbor.cpp:18-62,
 * descending index inside a cell), xs / ys / zs (N) f64 = the raw positions in that order.  *n_sorted (host) = N, or fewer when
 * absent atoms (x = NaN) were handed in — perm is then no permutation.  Synchronises `stream`. */
int mdh_spatial_sort(const double *x, const double *y, const double *z, int64_t N, const double *box9, const double *origin3,
                     const int *boundary3, double *xs, double *ys, double *zs, int *perm, int64_t *n_sorted, int space, void *stream);
/* mdh_permute: out[p] = in[perm[p]] (scatter == 0) or out[perm[p]] = in[p] (scatter != 0) for N elements of 4 or 8 bytes. */
int mdh_permute(const void *in, const int *perm, int64_t N, int elem_bytes, int scatter, void *out, int space, void *stream);
/* xs[p] = x[perm[p]] (and y, z): the three position columns through one permutation in one pass — the next frame of a trajectory read
 * through the previous frame's cell order (System's twin; no reference counterpart, see mdh_spatial_sort).  perm must hold indices in [0, N). */
int mdh_gather_positions(const double *x, const double *y, const double *z, const int *perm, int64_t N, double *xs, double *ys, double *zs,
                         int space, void *stream);
/* mdh_translate_rows: a list built on the sorted copy (row p = atom perm[p], entries = sorted indices) as the list of the
 * original order: verlet[perm[p]][s] = perm[verlet_sorted[p][s]] (pads < 0 stay), dist and nn rows moved along (both NULL or both
 * given, pairwise).  With rows built by mdh_build_neighbor_keyed(key = perm) the result is the list mdh_build_neighbor builds on
 * the original order, bit for bit. */
int mdh_translate_rows(const int *verlet_sorted, const double *dist_sorted, const int *nn_sorted, const int *perm, int64_t N,
                       int64_t M, int *verlet, double *dist, int *nn, int space, void *stream);

/*
 * first half of _neighbor.build_neighbor_without_max_neigh  src/neighbor.cpp:189-349:
 * counts only; *max_count (host int) receives max(nn) (0 when N == 0).  The
 * caller then allocates (N, max(max_count,1)) arrays and calls
 * mdh_build_neighbor(..., fill_pads=1).  Synchronises `stream`.
 */
int mdh_neighbor_count(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                       const double *origin3, const int *boundary3, double rc, int *nn, int *max_count, int space,
                       void *stream);

/*
 * replaces _neighbor.build_neighbor_without_max_neigh       src/neighbor.cpp:189-349 in ONE call: the cell grid is built
 * once and serves the counting pass and the build.  The reference returns freshly allocated arrays owned by capsules
 * (:312-317); here the caller supplies the allocator: after counting, alloc(user, N, M, &verlet, &dist) must hand back
 * (N, M) int32 / f64 arrays in the memory space of the call (M = max(max count, 1), also stored in *width); the rows are
 * then written pads included (-1 / rc+1, :320-329).  nn (N) int32.  Synchronises `stream` once (between the passes).
 * MDH_DEVICE calls remember the width found for a (N, grid) signature: the next call with the same signature builds at that
 * width at once and lets the counts of the build confirm it (no counting pass); if they do not, alloc is called a SECOND
 * time with the right width and the first arrays are to be dropped by the caller — *width and the arrays of the LAST
 * alloc call are the result.
 */
typedef int (*mdh_alloc_rows_fn)(void *user, int64_t N, int64_t M, int **verlet, double **dist);
int mdh_build_neighbor_exact(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                             const double *origin3, const int *boundary3, double rc, int *nn, int64_t *width,
                             mdh_alloc_rows_fn alloc, void *user, int space, void *stream);
/* the same with an in-cell ordering key (N) i64 or NULL, as mdh_build_neighbor_keyed takes it (a sorted copy of a system: key = the
 * original index, mdh_spatial_sort; a slab of a decomposed system: key = the global id) */
int mdh_build_neighbor_exact_keyed(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                   const double *origin3, const int *boundary3, double rc, int *nn, int64_t *width,
                                   mdh_alloc_rows_fn alloc, void *user, const int64_t *key, int space, void *stream);
/* the same, and the fixed-cutoff CNA labels of this cutoff in the same pass over the tiles (mdh_build_neighbor_fcna at the exact
 * width): what System.cal_common_neighbor_analysis(rc) runs when it has no list yet — build_neighbor(rc, max_neigh=None), then
 * fcna (src/neighbor.cpp:189-349, then src/cna.cpp:429-506).  pattern (N) i32, caller-initialised, as mdh_fcna leaves it; NULL:
 * no labels (mdh_build_neighbor_exact_keyed).  key as above, or NULL. */
int mdh_build_neighbor_exact_fcna(const double *x, const double *y, const double *z, int64_t N, const double *box9,
                                  const double *origin3, const int *boundary3, double rc, int *nn, int64_t *width,
                                  mdh_alloc_rows_fn alloc, void *user, int *pattern, const int64_t *key, int space, void *stream);

/* replaces _neighbor.sort_verlet_by_distance               src/neighbor.cpp:745-775 */
int mdh_sort_verlet_by_distance(int *verlet, double *dist, int64_t N, int64_t M, int sort_num, int space,
                                void *stream);

/* replaces _neighbor.wrap_positions                        src/neighbor.cpp:675-702 */
int mdh_wrap_positions(double *x, double *y, double *z, int64_t N, const double *box9, const double *origin3,
                       const int *boundary3, int space, void *stream);

/* replaces _neighbor.average_by_neighbor                   src/neighbor.cpp:704-743 */
int mdh_average_by_neighbor(double rc, const int *verlet, const double *dist, const int *nn, int64_t N, int64_t M,
                            const double *value, double *value_ave, int include_self, int space, void *stream);

/* ---- _cna ------------------------------------------------------------- */
/* replaces _cna.fcna (FixedCNA)                            src/cna.cpp:429-506; pattern pre-zeroed */
int mdh_fcna(const double *x, const double *y, const double *z, int64_t N, const double *box9,
             const double *origin3, const int *boundary3, const int *verlet, int64_t M, const int *nn, int *pattern,
             double rc, int space, void *stream);

/* replaces _cna.acna (AdaptiveCNA)                         src/cna.cpp:289-427; rows distance sorted, M >= 14 */
int mdh_acna(const double *x, const double *y, const double *z, int64_t N, const double *box9,
             const double *origin3, const int *boundary3, const int *verlet, int64_t M, int *pattern, int space,
             void *stream);

/* replaces _cna.ids (IdentifyDiamond)                      src/cna.cpp:163-287; new_verlet (N,12) */
int mdh_ids(const double *x, const double *y, const double *z, int64_t N, const double *box9, const double *origin3,
            const int *boundary3, const int *verlet, int64_t M, int *new_verlet, int *pattern, int space,
            void *stream);

/* ---- _csp ------------------------------------------------------------- */
/* replaces _csp.get_csp                                    src/centro_symmetry_parameter.cpp:12-94 */
int mdh_csp(const double *x, const double *y, const double *z, int64_t N, const double *box9, const double *origin3,
            const int *boundary3, const int *verlet, int64_t M, int num_neigh, double *csp, int space,
            void *stream);

/* ---- _sbo ------------------------------------------------------------- */
/* replaces _sbo.get_sq                                     src/steinhardt_bond_orientation.cpp:677-784
 * qlm_r/qlm_i (N,nl,2*lmax+1) pre-zeroed, qnarray (N,ncol); weight may be NULL when !use_weight.
 * llist is a HOST pointer (nl small ints). */
int mdh_get_sq(const double *x, const double *y, const double *z, int64_t N, const double *box9,
               const double *origin3, const int *boundary3, const int *verlet, const double *dist, int64_t M,
               const int *nn, const double *weight, const int *llist_host, int nl, int nnn, int lmax, int wl,
               int wlhat, int average, int use_voronoi, double rc, int use_weight, double *qlm_r, double *qlm_i,
               double *qnarray, int space, void *stream);

/* replaces _sbo.identifySolidLiquid                        src/steinhardt_bond_orientation.cpp:578-675 */
int mdh_identify_solid_liquid(int q6index, const double *Q6, const int *verlet, const double *dist, const int *nn,
                              int64_t N, int64_t M, const double *qlm_r, const double *qlm_i, int nl, int nz,
                              double threshold, int n_bond, int *solidliquid, int *nbond, int use_voronoi, int nnn,
                              double rc, int space, void *stream);

/* ---- _rdf ------------------------------------------------------------- */
/* replaces _rdf._rdf                                       src/radial_distribution_function.cpp:22-54 */
int mdh_rdf(const int *verlet, const double *dist, const int *nn, const int *type, int64_t N, int64_t M, double *g,
            int ntype, double rc, int nbin, int space, void *stream);
/* replaces _rdf._rdf_single_species                        src/radial_distribution_function.cpp:56-85 */
int mdh_rdf_single_species(const int *verlet, const double *dist, const int *nn, int64_t N, int64_t M, double *g,
                           double rc, int nbin, int space, void *stream);
/* replaces _rdf._rdf_streaming                             src/radial_distribution_function.cpp:143-317 */
int mdh_rdf_streaming(const double *x, const double *y, const double *z, const int *type, int64_t N,
                      const double *box9, const double *origin3, const int *boundary3, double *g, int ntype,
                      double rc, int nbin, int space, void *stream);

/* ---- _wcp ------------------------------------------------------------- */
/* replaces _wcp.get_wcp                                    src/warren_cowley_parameter.cpp:9-80; wcp (ntype,ntype) */
int mdh_wcp(const int *verlet, const int *nn, const int *type, int64_t N, int64_t M, int ntype, double *wcp,
            int space, void *stream);

/* extension (multi-GPU): the integer reductions of get_wcp (:26-55) over the rows with rows[i] != 0 (NULL = all rows);
 * counts (ntype*ntype + 2*ntype) u64 = Z_mn | Z_m | atoms per type.  Ranks all-reduce them, then apply :57-75. */
int mdh_wcp_counts(const int *verlet, const int *nn, const int *type, const unsigned char *rows, int64_t N, int64_t M,
                   int ntype, unsigned long long *counts, int space, void *stream);

/* ---- _fast_knn -------------------------------------------------------- */
/* replaces _fast_knn.knn                                   src/fast_knn.cpp:846-916 */
int mdh_knn(const double *x, const double *y, const double *z, int64_t N, const double *box9, const double *origin3,
            const int *boundary3, int k, int *indices, double *distances, int space, void *stream);
/* the same with key (N) i64, a PERMUTATION of 0 .. N-1, or NULL: the number every atom has in another numbering of the same
 * system (mdh_spatial_sort's perm for a cell-sorted copy).  Exact ties in distance are ordered by key instead of by index, so the
 * rows are those mdh_knn gives for the system in the key's numbering, neighbour for neighbour — on a perfect lattice, where the
 * k-th distance is a tie, WHICH neighbours are listed depends on it.  A key that is no permutation: memory-safe, rows meaningless. */
int mdh_knn_keyed(const double *x, const double *y, const double *z, int64_t N, const double *box9, const double *origin3,
                  const int *boundary3, int k, int *indices, double *distances, const int64_t *key, int space, void *stream);
/* mdh_knn_keyed (src/fast_knn.cpp:846-916) with the candidate rows of its cutoff build kept BY THE CALLER between searches of the
 * same positions: a System that asks for its 12 nearest (centro-symmetry) and then for its 14 nearest (adaptive CNA) builds the rows
 * once.  rows_io (N x M_io) i32 and counts_io (N) i32 are DEVICE buffers whatever `space` says; M_io a multiple of four,
 * >= mdh_knn_rows_width(k) for a search that is to FILL them.  *radius (host): in — > 0: the buffers hold every atom's neighbours
 * inside that radius for THESE positions in THIS box (the caller vouches for it), the build is skipped; 0: not yet — out: the radius
 * of what the buffers hold now, 0 when they hold nothing usable (a small system, a box too thin, k > 18: the cell walk took the
 * call).  Results are those of mdh_knn_keyed in every case; a query the rows cannot finish takes the cell walk. */
int mdh_knn_keyed_rows(const double *x, const double *y, const double *z, int64_t N, const double *box9, const double *origin3,
                       const int *boundary3, int k, int *indices, double *distances, const int64_t *key, int *rows_io, int *counts_io,
                       int M_io, double *radius, int space, void *stream);
/* slots per atom a search for k neighbours wants in rows_io (0: the rows path does not serve this k) */
int mdh_knn_rows_width(int k);

/* ---- _repeat_cell ----------------------------------------------------- */
/* replaces _repeat_cell.repeat_cell                        src/repeat_cell.cpp:19-61; new_pos flat (n_old*nx*ny*nz*3) */
int mdh_repeat_cell(double *new_pos, const double *old_box9_host, const double *old_pos, int64_t n_old, int nx,
                    int ny, int nz, int space, void *stream);

/* ---- _ptm --------------------------------------------------------------- */
/* replaces _ptm.get_ptm                                     src/polyhedral_template_matching.cpp:135-318
 * (both passes: ptm_preorder_neighbours :215-255 and ptm_index :258-318 of extern/ptm).
 * structure: the reference's structure string ("fcc-hcp-bcc", "default", ...; separators " ,-_|", :168-206).
 * verlet (N,M) int32 rows sorted by distance (the caller passes the 18 nearest neighbours); types may be NULL.
 * output (N,ncol>=8) f64: type, alloy ordering, rmsd, interatomic distance, orientation quaternion w x y z;
 * ptm_indices (N,nind) i32: matched atoms in template order, -1 padded (:296-305).
 * All eight structure types of the library are built: sc, fcc, hcp, ico, bcc, and the two-shell ones dcub, dhex, graphene. */
int mdh_ptm(const char *structure, const double *x, const double *y, const double *z, int64_t N, const double *box9_host,
            const double *origin3_host, const int *boundary3_host, const int *verlet, int64_t M, const int *types,
            double rmsd_threshold, double *output, int ncol, int *ptm_indices, int nind, int space, void *stream);
/* the PTM_CHECK_* bit mask the structure string selects (:168-206) */
int mdh_ptm_flags(const char *structure);

/* ---- list consumers ---------------------------------------- */
/* replaces _aja.compute_aja                                 src/ackland_jones_analysis.cpp:9-172
 * rows: >= 14 neighbours sorted by distance; aja (N) i32: 0 other, 1 fcc, 2 hcp, 3 bcc, 4 ico */
int mdh_aja(const double *x, const double *y, const double *z, int64_t N, const double *box9_host,
            const double *origin3_host, const int *boundary3_host, const int *verlet, const double *dist, int64_t M,
            int *aja, int space, void *stream);
/* replaces _cnp.compute_cnp                                 src/common_neighbor_parameter.cpp:10-137; cnp (N) f64 */
int mdh_cnp(const double *x, const double *y, const double *z, int64_t N, const double *box9_host,
            const double *origin3_host, const int *boundary3_host, const int *verlet, const double *dist, const int *nn,
            int64_t M, double *cnp, double rc, int space, void *stream);
/* replaces _structure_entropy.calculate_structure_entropy   src/structure_entropy.cpp:9-108; entropy (N) f64 */
int mdh_structure_entropy(double rc, double sigma, int use_local_density, double volume, const double *dist,
                          const int *nn, int64_t N, int64_t M, double *entropy, int space, void *stream);

/* replaces _atomtemp.compute_temp                           src/atomic_temperature.cpp:9-112; velocities in m/s, mass g/mol */
int mdh_atomic_temperature(const int *verlet, const double *dist, int64_t N, int64_t M, const double *vx, const double *vy,
                           const double *vz, const double *mass, double *T, double rc, int space, void *stream);
/* replaces _cluster.get_cluster (by_bond=0, :9-56) and _cluster.get_cluster_by_bond (by_bond=1, :58-106).
 * cluster (N) i32 receives ids 1..n numbered by smallest member index; *n_clusters_host = n (the functions' return value).
 * Connected components of the (symmetric) bond graph by min-index union-find instead of the serial flood fill. */
int mdh_cluster(const int *verlet, const double *dist, const int *nn, int64_t N, int64_t M, double rc, int by_bond,
                int *cluster, int *n_clusters_host, int space, void *stream);
/* replaces _cluster.filter_by_type                           src/cluster.cpp:108-148; t1/t2/r host arrays of length ntype */
int mdh_filter_by_type(int *verlet, const double *dist, const int *nn, const int *type, int64_t N, int64_t M,
                       const int *t1_host, const int *t2_host, const double *r_host, int ntype, int space, void *stream);

/* replaces _fccpft.identify_sftb_fcc                       src/identify_fcc_planar_faults.cpp:54-245
 * hcp_indices (n_hcp) ascending atom ids with structure type 2; hcp_neighbors (n_hcp,12) caller scratch (written);
 * ptm_indices (N,12) = columns 1..12 of the PTM neighbour rows; fault_types (N) pre-zeroed, HCP entries written:
 * 1 other, 2 intrinsic SF, 3 twin boundary, 4 multi-layer SF, 5 extrinsic SF (only with identify_esf). */
int mdh_identify_sftb_fcc(const int *hcp_indices, int64_t n_hcp, int *hcp_neighbors, const int *ptm_indices,
                          const int *structure_types, int64_t N, int *fault_types, int identify_esf, int space, void *stream);

/* replaces _neighbor.filter_overlap_atom                   src/neighbor.cpp:390-486 (polycrystal builder)
 * keep (N) u8: 0 for every atom that has a lower-numbered atom within rc, else 1 */
int mdh_filter_overlap_atom(const double *x, const double *y, const double *z, int64_t N, const double *box9_host,
                            const double *origin3_host, const int *boundary3_host, double rc, unsigned char *keep,
                            int space, void *stream);

/* replaces _neighbor.filter_overlap_atom_with_grain          src/neighbor.cpp:489-672 (graphene-decorated polycrystals)
 * type: 1 metal / 2 carbon, grain_id per atom.  keep (N) u8 = the result of the reference's sweep run SERIALLY (index order;
 * metal-metal: the higher index goes; carbon-carbon: same grain -> higher index, different grains -> larger grain id;
 * metal-carbon: the metal goes; removed atoms do not act).  With several threads the reference's result depends on the schedule. */
int mdh_filter_overlap_atom_with_grain(const double *x, const double *y, const double *z, const int *type, const int *grain_id,
                                       int64_t N, const double *box9_host, const double *origin3_host, const int *boundary3_host,
                                       double rc_metal_metal, double rc_cc, double rc_metal_c, unsigned char *keep, int space,
                                       void *stream);

/* replaces _polycrystal.transform_and_filter                src/polycrystal.cpp:20-125 (polycrystal builder)
 * p' = R (p - center) + target (rotation9 = row-major R, i.e. (p - center) @ R.T); kept when
 * a*p'x + b*p'y + c*p'z + d < 0 for every row (a,b,c,d) of coeffs (nf <= 1024, host array); survivors in input order in
 * rows [0, *count_host) of out_pos (capacity (n,3)). */
int mdh_transform_and_filter(const double *x, const double *y, const double *z, int64_t n, const double *rotation9_host,
                             const double *center3_host, const double *target3_host, const double *coeffs_host, int nf,
                             double *out_pos, int64_t *count_host, int space, void *stream);

/* ---- _voronoi (volume functions) ------------------------------ */
/* replaces _voronoi.get_voronoi_volume_number_radius         src/voronoi.cpp:16-71 (and, called with a LAMMPS-aligned
 * triclinic box and all-periodic boundary, get_voronoi_volume_number_radius_tri :73-147).
 * volume (N) f64, nfaces (N) i32 (walls of open axes count, as voro++'s number_of_faces), radius (N) f64 =
 * sqrt(max_radius_squared) of voro++ = twice the largest vertex distance.  Cells are built by half-space clipping from
 * the neighbours within an automatically enlarged search radius; returns MDH_ERR_ARG when a cell would reach beyond half
 * a periodic box length (replicate the system first). */
int mdh_voronoi_volume_number_radius(const double *x, const double *y, const double *z, int64_t N, const double *box9_host,
                                     const double *origin3_host, const int *boundary3_host, double *volume, int *nfaces,
                                     double *radius, int space, void *stream);

/* replaces _voronoi.get_voronoi_neighbor                      src/voronoi.cpp:307-447, in two calls:
 * count: neighbor_number (N) i32 = faces per cell (walls included), *width_host = their maximum;
 * rows:  verlet / distance / face_area (N, width): faces shared with atoms whose area exceeds
 *        max(a_thr, r_thr * total face area of the cell), nearest first (voro++'s face order is internal to that
 *        library; consumers skip -1 entries), padded with -1 / 10000 / 0. */
int mdh_voronoi_neighbor_count(const double *x, const double *y, const double *z, int64_t N, const double *box9_host,
                               const double *origin3_host, const int *boundary3_host, int *neighbor_number, int *width_host,
                               int space, void *stream);
int mdh_voronoi_neighbor(const double *x, const double *y, const double *z, int64_t N, const double *box9_host,
                         const double *origin3_host, const int *boundary3_host, double a_face_area_threshold,
                         double r_face_area_threshold, int *verlet, double *distance, double *face_area, int width, int space,
                         void *stream);

/* mdh_voronoi_neighbor_count and mdh_voronoi_neighbor from ONE construction of the cells (the reference's function builds its
 * container once, src/voronoi.cpp:307-447): rows are made `width` columns wide on the device; if no cell has more faces, verlet /
 * distance / face_area (buffers of N * width entries) receive them as (N, *width_out_host) arrays, *width_out_host being the
 * width mdh_voronoi_neighbor_count reports, and neighbor_number (N) the face counts; otherwise only *width_out_host (> width) is
 * written: call again with that width. */
int mdh_voronoi_neighbor_rows(const double *x, const double *y, const double *z, int64_t N, const double *box9_host,
                              const double *origin3_host, const int *boundary3_host, double a_face_area_threshold,
                              double r_face_area_threshold, int *verlet, double *distance, double *face_area, int width,
                              int *neighbor_number, int *width_out_host, int space, void *stream);

/* geometry of _voronoi.get_cell_info                          src/voronoi.cpp:449-540: per cell, every face (walls of open axes
 * included) as a polygon.  face_nv (N,W) i32 vertices per face slot (0: none), face_area (N,W), face_vert (N,W,V,3) vertices
 * relative to the atom in polygon order, nfaces / volume / radius (N) as in mdh_voronoi_volume_number_radius.
 * W >= width of mdh_voronoi_neighbor_count; *need_v_host > V on return: repeat with that V. */
int mdh_voronoi_cell_info(const double *x, const double *y, const double *z, int64_t N, const double *box9_host,
                          const double *origin3_host, const int *boundary3_host, int W, int V, int *nfaces, int *face_nv,
                          double *face_area, double *face_vert, double *volume, double *radius, int *need_v_host, int space,
                          void *stream);

/* distance column of _voronoi.get_voronoi_neighbor_tri         src/voronoi.cpp:277-282: sqrt(box.pbc(x[j] - x[i])) with the
 * caller's UNROTATED positions and the LAMMPS-aligned box / boundary flags of that call; -1 entries -> 10000. */
int mdh_voronoi_row_distance(const int *verlet, int64_t N, int width, const double *x, const double *y, const double *z,
                             const double *box9_host, const double *origin3_host, const int *boundary3_host, double *distance,
                             int space, void *stream);

/* ---- _sfc (static structure factor, direct summation) ------------ */
/* replaces _sfc.compute_sfc_direct                          src/structure_factor.cpp:654-680 (StructureFactorDirect :64-447)
 * sf_host (bins) is always a host array; qx/qy/qz NULL = total S(k), otherwise the cross term between the two point sets
 * (n_total required then).  Empty bins are NaN. */
int mdh_sfc_direct(const double *x, const double *y, const double *z, int64_t n, const double *box9_host, double *sf_host, int bins,
                   double k_max, double k_min, const double *qx, const double *qy, const double *qz, int64_t nq,
                   unsigned n_total, int space, void *stream);
/* replaces _sfc.compute_sfc_direct_partial                  src/structure_factor.cpp:682-705 (:451-640)
 * out_host (ntype, ntype, bins) host array of Ashcroft-Langreth partials; type 0-based, ntype <= 16 */
int mdh_sfc_direct_partial(const double *x, const double *y, const double *z, const int *type, int ntype, int64_t n,
                           const double *box9_host, double *out_host, int bins, double k_max, double k_min, int space,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MDAPY_AMD_H */
