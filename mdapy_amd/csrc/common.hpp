// common.hpp — shared host/device definitions of libmdapy_amd (gfx950 only).
//
// Numerical contract: every kernel whose output is compared
// bit-exactly with the reference is compiled with -ffp-contract=off and uses
// IEEE '/', sqrt and floor; the operation ORDER of each formula follows the
// reference line cited next to it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/mdapy_amd.h"

namespace mdh {

// ----------------------------------------------------------------------------
// Simulation box as the kernels see it (reference: struct Box, src/box.h:8-17).
// Passed by value as a kernel argument (lands in SGPRs / the kernarg segment).
// ----------------------------------------------------------------------------
struct DBox {
    double h[9];      // rows a,b,c
    double hi[9];     // inverse (triclinic: adjugate/det, box.h:182-203; ortho: 1/L on the diagonal)
    double o[3];      // origin
    double thick[3];  // perpendicular thickness, box.h:54-89
    // Orthogonal minimum image n(d) = floor(d/L + 0.5) is a monotone step
    // function of d.  tn[a][k] is the smallest double d with n(d) >= k-1
    // (k = 0..3, i.e. n >= -1, 0, 1, 2), found on the host by bisection with
    // the very same IEEE expression, so `d >= tn` reproduces the reference's
    // floor(d/L+0.5) bit for bit without a division (see pbc_axis()).
    double tn[3][4];
    int pbc[3];
    int tri;
    int anypbc;
};

// host: build DBox from the C-ABI box arguments; returns MDH_OK / MDH_ERR_BOX
int make_box(DBox &b, const double *box9, const double *origin3, const int *boundary3);

// ----------------------------------------------------------------------------
// device-side geometry
// ----------------------------------------------------------------------------
__device__ __forceinline__ double pbc_axis(double d, double L, const double *t)
{
    // reference: xij -= L * floor(xij / L + 0.5)   (box.h:120-124)
    double n;
    if (d >= t[0] && d < t[3])
        n = (d >= t[2]) ? 1.0 : ((d >= t[1]) ? 0.0 : -1.0);
    else
        n = floor(d / L + 0.5);
    return d - L * n;
}

template <bool TRI>
__device__ __forceinline__ void pbc(const DBox &b, double &dx, double &dy, double &dz)
{
    if (TRI) { // box.h:99-114
        double fx = dx * b.hi[0] + dy * b.hi[3] + dz * b.hi[6];
        double fy = dx * b.hi[1] + dy * b.hi[4] + dz * b.hi[7];
        double fz = dx * b.hi[2] + dy * b.hi[5] + dz * b.hi[8];
        if (b.pbc[0]) fx -= floor(fx + 0.5);
        if (b.pbc[1]) fy -= floor(fy + 0.5);
        if (b.pbc[2]) fz -= floor(fz + 0.5);
        dx = fx * b.h[0] + fy * b.h[3] + fz * b.h[6];
        dy = fx * b.h[1] + fy * b.h[4] + fz * b.h[7];
        dz = fx * b.h[2] + fy * b.h[5] + fz * b.h[8];
    } else {
        if (b.pbc[0]) dx = pbc_axis(dx, b.h[0], b.tn[0]);
        if (b.pbc[1]) dy = pbc_axis(dy, b.h[4], b.tn[1]);
        if (b.pbc[2]) dz = pbc_axis(dz, b.h[8], b.tn[2]);
    }
}

// floor(d / L) of box.h:158-176.  For 0 <= d < L (1 - 1e-12) the rounded quotient is below 1, so the floor is 0 without
// evaluating the f64 division (~35 instructions); atoms inside the box — nearly all of them — take this path.
__device__ __forceinline__ double wrap_count(double d, double L)
{
    if (d >= 0.0 && d < L * 0.999999999999)
        return 0.0;
    return floor(d / L);
}

template <bool TRI>
__device__ __forceinline__ void wrap(const DBox &b, double &x, double &y, double &z)
{
    if (TRI) { // box.h:133-156
        double dx = x - b.o[0], dy = y - b.o[1], dz = z - b.o[2];
        double fx = dx * b.hi[0] + dy * b.hi[3] + dz * b.hi[6];
        double fy = dx * b.hi[1] + dy * b.hi[4] + dz * b.hi[7];
        double fz = dx * b.hi[2] + dy * b.hi[5] + dz * b.hi[8];
        if (b.pbc[0]) fx -= floor(fx);
        if (b.pbc[1]) fy -= floor(fy);
        if (b.pbc[2]) fz -= floor(fz);
        x = b.o[0] + fx * b.h[0] + fy * b.h[3] + fz * b.h[6];
        y = b.o[1] + fx * b.h[1] + fy * b.h[4] + fz * b.h[7];
        z = b.o[2] + fx * b.h[2] + fy * b.h[5] + fz * b.h[8];
    } else { // box.h:158-176
        if (b.pbc[0]) { double d = x - b.o[0]; x = b.o[0] + d - b.h[0] * wrap_count(d, b.h[0]); }
        if (b.pbc[1]) { double d = y - b.o[1]; y = b.o[1] + d - b.h[4] * wrap_count(d, b.h[4]); }
        if (b.pbc[2]) { double d = z - b.o[2]; z = b.o[2] + d - b.h[8] * wrap_count(d, b.h[8]); }
    }
}

// squared minimum-image distance between two RAW positions (src/cna.cpp:149-161)
template <bool TRI>
__device__ __forceinline__ double pair_d2(const DBox &b, double xi, double yi, double zi, double xj, double yj,
                                          double zj)
{
    double dx = xj - xi, dy = yj - yi, dz = zj - zi;
    pbc<TRI>(b, dx, dy, dz);
    return dx * dx + dy * dy + dz * dz;
}

// ----------------------------------------------------------------------------
// host runtime: errors, per-device scratch cache, host<->HBM staging
// ----------------------------------------------------------------------------
void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define MDH_HIP(call)                                                         \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess)                                                \
            return ::mdh::hip_fail(e__, #call, __FILE__, __LINE__);           \
    } while (0)

#define MDH_TRY(expr)                                                         \
    do {                                                                      \
        int rc__ = (expr);                                                    \
        if (rc__ != MDH_OK)                                                   \
            return rc__;                                                      \
    } while (0)

// A call-scoped view of the scratch cache.  Buffers handed out live until the
// Scope is destroyed; they go back to the per-device cache (not to the driver),
// so steady-state calls perform no hipMalloc/hipFree at all.
class Scope {
  public:
    explicit Scope(void *stream);
    ~Scope();
    hipStream_t stream() const { return stream_; }
    // device scratch, 256-byte aligned; nullptr + error set on failure
    void *alloc(size_t bytes);
    template <class T> T *alloc_n(size_t n) { return static_cast<T *>(alloc((n ? n : 1) * sizeof(T))); }
    // Kept blocks: device memory with an invariant that holds whenever the block is idle, so that a call needs no
    // hipMemsetAsync (a ~7 us node of its own on the stream) to bring it into its starting state.  A kept block is only ever
    // handed out for its own tag, zero-filled when it is created.
    //   KEEP_ZERO : all zero when idle (the bin counters of a cell grid: the scan that reads them writes the zeros back)
    //   KEEP_SCAN : control words of the single-pass scan (a ticket that its last block resets, generation-stamped status
    //               words: a stale one is never mistaken for a fresh one)
    //   KEEP_TODO : word 0 (blocks done) and word 64 (length of a to-do list) zero when idle (reset by the list's last reader)
    // keep_confirm(p): the kernels that restore p's invariant have been enqueued; a KEEP_ZERO / KEEP_TODO block that leaves
    // its Scope unconfirmed (an error path) is zero-filled on the stream before it returns to the cache.
    enum Keep { KEEP_NONE = 0, KEEP_ZERO = 1, KEEP_SCAN = 2, KEEP_TODO = 3 };
    void *alloc_kept(size_t bytes, Keep tag);
    void keep_confirm(void *p);

    // Stage a caller array.  space==MDH_DEVICE: returns the pointer itself.
    // space==MDH_HOST: returns a device copy (uploaded when `in`), and remembers
    // to download it at finish() when `out`.
    template <class T> T *stage(T *p, size_t n, int space, bool in, bool out)
    {
        return static_cast<T *>(stage_raw(const_cast<void *>(static_cast<const void *>(p)), n * sizeof(T), space, in, out));
    }
    template <class T> const T *stage_in(const T *p, size_t n, int space)
    {
        return static_cast<const T *>(stage_raw(const_cast<T *>(p), n * sizeof(T), space, true, false));
    }
    // download staged outputs (host space) and, for host space, synchronise.
    int finish(int space);
    bool failed() const { return failed_; }
    int error() const { return err_; } // MDH_ERR_HIP (no device / HIP failure) or MDH_ERR_NOMEM

  private:
    void *stage_raw(void *p, size_t bytes, int space, bool in, bool out);
    struct Out { void *host; void *dev; size_t bytes; };
    hipStream_t stream_;
    int device_;
    bool failed_ = false;
    int err_ = MDH_ERR_NOMEM;
    static const int kMaxHeld = 64;
    void *held_[kMaxHeld];
    size_t held_bytes_[kMaxHeld];
    unsigned char held_keep_[kMaxHeld]; // Keep tag | 0x80 once confirmed
    int nheld_ = 0;
    void *alloc_impl(size_t bytes, int tag);
    Out outs_[16];
    int nouts_ = 0;
};

void reset_kept_blocks(int tag); // runtime.hip

// A word of pinned host memory per (tag, a, b, array) signature — a shape and the device array the question is about — that a sampling
// kernel of some call writes and the NEXT calls with the same signature read on the host without waiting for anything; a new array of a
// known shape starts from the latest answer for that shape ("do this system's atoms come in a spatial order?", "do these rows name
// neighbours far away in memory?").  A stale or missing answer costs speed, never correctness: both ways of doing the work give
// the same result.  sample: true on the first and every eighth call of the signature — the caller launches its sampling kernel then.
struct OrderHint { int *word; bool sample; };
OrderHint order_hint(int tag, int64_t a, int64_t b, const void *array); // runtime.hip; word == nullptr if pinned memory could not be had

// Positions as one 32-byte record per atom (x, y, z, unused): a neighbour's position is then two 16-byte requests instead of
// three 8-byte ones — the list consumers that gather 12 - 18 neighbours per atom are bound by the number of lane requests their
// gathers make (about two cycles each in the CU's address unit), not by bytes.  Packed into scratch at the start of a call:
// 0.1 ms per 10 M atoms.
struct __attribute__((aligned(16))) Pos4 { double x, y, z, w; };
// the first NI entries of a list row in 16-byte requests (rows start at any 4-byte address: the hardware takes unaligned ones)
typedef int RowQuad __attribute__((ext_vector_type(4), aligned(4)));
template <int NI>
__device__ __forceinline__ void load_row(const int *__restrict__ row, int (&ids)[NI])
{
#pragma unroll
    for (int q = 0; q < NI / 4; ++q) {
        const RowQuad v = *reinterpret_cast<const RowQuad *>(row + 4 * q);
        ids[4 * q] = v.x; ids[4 * q + 1] = v.y; ids[4 * q + 2] = v.z; ids[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int a = NI & ~3; a < NI; ++a) ids[a] = row[a];
}
const Pos4 *pack_positions(Scope &sc, const double *x, const double *y, const double *z, int64_t N);

// Rows of a list for a thread-per-row kernel, ROW_CHUNK columns at a time through LDS.  A lane that walks its own row reads 4 or
// 8 bytes at a time from an address M entries away from its neighbour's: every load instruction of the wave touches 64 cache
// lines, and with rows of 40-50 entries the lines are gone from L1 and L2 before the lane comes back for their next entry — the
// list is fetched 16-32 times (a 2 GB list walked by `k_cc_hook`, `k_average`, `k_atomic_temp`: 5-7 ms each at 4 M atoms).  Here
// the 64 rows of a one-wave workgroup are read a chunk at a time in 16-byte requests (four lanes to a row's 16 ids, eight to
// its 16 distances: whole 64- and 128-byte pieces), entry c of row t lands at [c * 64 + t], and lane t walks its row from there in
// list order, as before.  Workgroups of 64 threads; `ids` and `dst` are ROW_CHUNK * 64 entries each.
constexpr int ROW_CHUNK = 16;
typedef double RowPair __attribute__((ext_vector_type(2), aligned(8)));
template <bool WITH_DIST>
__device__ __forceinline__ void stage_row_chunk(const int *__restrict__ verlet, const double *__restrict__ dist, int64_t N, int64_t M,
                                                int64_t row0, int c0, int *__restrict__ ids, double *__restrict__ dst)
{
    const int t = threadIdx.x;
#pragma unroll
    for (int e = t; e < 64 * (ROW_CHUNK / 4); e += 64) {
        const int row = e / (ROW_CHUNK / 4), c = (e % (ROW_CHUNK / 4)) * 4;
        const int64_t i = row0 + row;
        if (i < N && c0 + c < M) {
            const int *src = verlet + i * M + c0 + c;
            if (c0 + c + 3 < M) {
                const RowQuad v = *reinterpret_cast<const RowQuad *>(src);
                ids[(c + 0) * 64 + row] = v.x; ids[(c + 1) * 64 + row] = v.y; ids[(c + 2) * 64 + row] = v.z; ids[(c + 3) * 64 + row] = v.w;
            } else {
                for (int k = 0; c0 + c + k < M; ++k) ids[(c + k) * 64 + row] = src[k];
            }
        }
    }
    if (WITH_DIST) {
#pragma unroll
        for (int e = t; e < 64 * (ROW_CHUNK / 2); e += 64) {
            const int row = e / (ROW_CHUNK / 2), c = (e % (ROW_CHUNK / 2)) * 2;
            const int64_t i = row0 + row;
            if (i < N && c0 + c < M) {
                const double *src = dist + i * M + c0 + c;
                if (c0 + c + 1 < M) {
                    const RowPair v = *reinterpret_cast<const RowPair *>(src);
                    dst[(c + 0) * 64 + row] = v.x; dst[(c + 1) * 64 + row] = v.y;
                } else {
                    dst[c * 64 + row] = src[0];
                }
            }
        }
    }
}
// the largest of a per-lane count over the wave (how many chunks the workgroup has to stage)
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}

// An index taken from a caller's list, made safe to dereference: an entry outside [0, n) (a pad of a k-nearest list in a
// system of fewer than k+1 atoms, a list that belongs to another system) reads atom `fallback` instead of faulting the
// GPU.  The reference reads out of bounds there (undefined behaviour), so any defined result is as good as its.
__device__ __forceinline__ int safe_id(int j, int64_t fallback, int64_t n) { return (unsigned)j < (unsigned)n ? j : (int)fallback; }

inline int grid_for(int64_t n, int block) { return (int)((n + block - 1) / block); }

// Every translation unit of the library is one code object, loaded by the HIP runtime at the first launch of one of its
// kernels (1-2.5 ms apiece on an MI355X box, 65 ms for the first of the process).  MDH_WARM_UNIT(name) gives a unit an empty
// kernel; mdh_warm() (runtime.hip) launches them all, so that the loading happens where a caller expects set-up cost — the
// first use of a device — and not inside its first build_neighbor, cal_centro_symmetry_parameter, ...
#define MDH_WARM_UNIT(name)                                                                                                    \
    namespace mdh {                                                                                                            \
    __global__ void k_warm_##name() {}                                                                                         \
    void warm_##name(hipStream_t st) { hipLaunchKernelGGL(k_warm_##name, dim3(1), dim3(64), 0, st); }                          \
    }

// Scoped HIP-event pair around a kernel launch (no-op unless mdh_prof_enable(1)); prof.hip
class ProfRange {
  public:
    ProfRange(const char *name, hipStream_t st);
    ~ProfRange();

  private:
    const char *name_;
    hipStream_t st_;
    hipEvent_t a_;
};

} // namespace mdh
