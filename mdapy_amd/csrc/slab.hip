// slab.hip — halo selection of the slab decomposition in one pass over the owned atoms.
//
// For every owned atom the wrapped fractional coordinate along the decomposed axis,
//     f = (x-o0)*hi[0] + (y-o1)*hi[1] + (z-o2)*hi[2];  f -= floor(f);  if (f >= 1) f -= 1
// (the expression of mdapy_amd/distributed.py::SlabDecomposition.frac, evaluated in the same order, no FMA), and the two
// selections  f >= up_from  (layer sent to the right neighbour)  and  f < down_below  (sent to the left one).  The indices
// are appended through one atomic per wavefront; their order is irrelevant (the receiver sorts ghosts by global id).
// HBM-bound: 24 B read per atom, 4 B written per selected atom.
#include "common.hpp"

namespace mdh {

__global__ __launch_bounds__(256) void k_slab_select(const double *__restrict__ x, const double *__restrict__ y,
                                                     const double *__restrict__ z, int64_t n, double o0, double o1, double o2,
                                                     double h0, double h1, double h2, double up_from, double down_below,
                                                     int *__restrict__ up, int *__restrict__ down, int *__restrict__ counts,
                                                     const int64_t *__restrict__ gid, double *__restrict__ up_pack,
                                                     double *__restrict__ down_pack, int capacity)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool is_up = false, is_down = false;
    if (i < n) {
        double f = (x[i] - o0) * h0 + (y[i] - o1) * h1 + (z[i] - o2) * h2;
        f = f - floor(f);
        if (f >= 1.0) f = f - 1.0;
        is_up = f >= up_from;
        is_down = f < down_below;
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long mu = __ballot(is_up), md = __ballot(is_down);
    int bu = 0, bd = 0;
    if (lane == 0) {
        if (mu) bu = atomicAdd(&counts[0], __popcll(mu));
        if (md) bd = atomicAdd(&counts[1], __popcll(md));
    }
    bu = __shfl(bu, 0, 64);
    bd = __shfl(bd, 0, 64);
    if (is_up) {
        const int s = bu + __popcll(mu & ((1ull << lane) - 1ull));
        if (s < capacity) { // (the counters keep running: the host sees that the buffers were too small and comes back)
            up[s] = (int)i;
            if (up_pack) { double *o = up_pack + (int64_t)s * 4; o[0] = x[i]; o[1] = y[i]; o[2] = z[i]; o[3] = (double)gid[i]; }
        }
    }
    if (is_down) {
        const int s = bd + __popcll(md & ((1ull << lane) - 1ull));
        if (s < capacity) {
            down[s] = (int)i;
            if (down_pack) { double *o = down_pack + (int64_t)s * 4; o[0] = x[i]; o[1] = y[i]; o[2] = z[i]; o[3] = (double)gid[i]; }
        }
    }
}

// The same selection written straight into the two messages of the exchange (device memory, wire layout: [0] = number of
// atoms as a double, then `width` rows of `cap` columns — x, y, z, the extra per-atom columns, the global id last), and no
// word goes to the host: the receiver reads the count out of the message it gets.
struct SlabExtras { const double *p[4]; int n; };
__global__ __launch_bounds__(256) void k_slab_messages(const double *__restrict__ x, const double *__restrict__ y,
                                                       const double *__restrict__ z, int64_t n, double o0, double o1, double o2,
                                                       double h0, double h1, double h2, double up_from, double down_below,
                                                       int *__restrict__ counts, const int64_t *__restrict__ gid, SlabExtras ex,
                                                       double *__restrict__ msg_up, double *__restrict__ msg_down, int cap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool is_up = false, is_down = false;
    if (i < n) {
        double f = (x[i] - o0) * h0 + (y[i] - o1) * h1 + (z[i] - o2) * h2;
        f = f - floor(f);
        if (f >= 1.0) f = f - 1.0;
        is_up = f >= up_from;
        is_down = f < down_below;
    }
    // slots: a wave's selections are counted by a ballot, the workgroup's waves take their places from two LDS counters, and ONE
    // thread per workgroup and direction goes to the global counter (a slab's layers are contiguous stretches of its atoms: with one
    // global atomic per wave the ~1 700 waves that hold the layers queued on two words — 32 us of a 390 us step at 1.26 M atoms)
    __shared__ int s_cnt[2], s_base[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned long long mu = __ballot(is_up), md = __ballot(is_down);
    int bu = 0, bd = 0;
    if (lane == 0) {
        if (mu) bu = atomicAdd(&s_cnt[0], __popcll(mu));
        if (md) bd = atomicAdd(&s_cnt[1], __popcll(md));
    }
    __syncthreads();
    if (threadIdx.x < 2) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]) : 0;
    __syncthreads();
    bu = __shfl(bu, 0, 64) + s_base[0];
    bd = __shfl(bd, 0, 64) + s_base[1];
    auto put = [&](double *msg, int s) {
        if (s >= cap) // (the counter keeps running: the count in the header tells both ends that the message was too small)
            return;
        double *o = msg + 1 + s;
        o[0] = x[i];
        o[(int64_t)cap] = y[i];
        o[2 * (int64_t)cap] = z[i];
        for (int k = 0; k < ex.n; ++k) o[(3 + k) * (int64_t)cap] = ex.p[k][i];
        o[(3 + ex.n) * (int64_t)cap] = (double)gid[i];
    };
    if (is_up) put(msg_up, bu + __popcll(mu & ((1ull << lane) - 1ull)));
    if (is_down) put(msg_down, bd + __popcll(md & ((1ull << lane) - 1ull)));
}

__global__ void k_slab_headers(int *__restrict__ counts, double *__restrict__ msg_up, double *__restrict__ msg_down)
{
    msg_up[0] = (double)counts[0];
    msg_down[0] = (double)counts[1];
    counts[0] = counts[1] = 0;
}

} // namespace mdh

using namespace mdh;

// msg_up / msg_down: device memory, 1 + (4 + nextra) * cap doubles each.  extras: nextra (<= 4) device pointers to f64 columns
// of the owned atoms.  Everything stays on the stream: no synchronisation.
extern "C" int mdh_slab_halo_messages(const double *x, const double *y, const double *z, int64_t n, const double *origin3,
                                      const double *hi3, double up_from, double down_below, const int64_t *gid,
                                      const double *const *extras, int nextra, double *msg_up, double *msg_down, int64_t cap,
                                      void *stream)
{
    if (n < 0 || n >= 2147483647LL || cap < 0 || cap >= 2147483647LL || nextra < 0 || nextra > 4 || !gid || !msg_up || !msg_down) {
        set_error("mdh_slab_halo_messages: bad arguments");
        return MDH_ERR_ARG;
    }
    Scope sc(stream);
    hipStream_t st = sc.stream();
    // (two counters in a kept block that is zero whenever idle: k_slab_headers clears what it reads — no memset launch per exchange)
    int *dc = static_cast<int *>(sc.alloc_kept(2 * sizeof(int), Scope::KEEP_ZERO));
    if (sc.failed())
        return sc.error();
    SlabExtras ex{{nullptr, nullptr, nullptr, nullptr}, nextra};
    for (int k = 0; k < nextra; ++k) ex.p[k] = extras[k];
    if (n > 0)
        hipLaunchKernelGGL(k_slab_messages, dim3(grid_for(n, 256)), dim3(256), 0, st, x, y, z, n, origin3[0], origin3[1], origin3[2],
                           hi3[0], hi3[1], hi3[2], up_from, down_below, dc, gid, ex, msg_up, msg_down, (int)cap);
    hipLaunchKernelGGL(k_slab_headers, dim3(1), dim3(1), 0, st, dc, msg_up, msg_down);
    sc.keep_confirm(dc);
    MDH_HIP(hipGetLastError());
    return MDH_OK;
}

// The receiving side of the same messages: the nl + nr atoms of the two incoming messages (layout of mdh_slab_halo_messages:
// [count, row 0, row 1, ..., id row], rows `cap` long) are written behind the owned atoms — column k of the message to
// cols[k][n_owned ...] (f64), the id row to gid[n_owned ...] (i64; ids < 2^53 are exact in f64) — left message first.  One
// launch instead of two copies per column; nl and nr are the counts the caller has already read from the headers.
namespace mdh {
struct SlabColumns { double *p[7]; int n; };
__global__ __launch_bounds__(256) void k_slab_append(const double *__restrict__ msg_l, const double *__restrict__ msg_r, int64_t cap, int64_t nl,
                                                     int64_t nr, SlabColumns cols, int64_t *__restrict__ gid, int64_t n_owned)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nl + nr)
        return;
    const double *m = i < nl ? msg_l : msg_r;
    const int64_t j = i < nl ? i : i - nl;
    for (int k = 0; k < cols.n; ++k) cols.p[k][n_owned + i] = m[1 + (int64_t)k * cap + j];
    gid[n_owned + i] = (int64_t)m[1 + (int64_t)cols.n * cap + j];
}
} // namespace mdh

// The same with NOTHING read by the host: the ghost block behind the owned atoms has the fixed size 2 cap — left message at
// n_owned + [0, cap), right message at n_owned + cap + [0, cap) — the counts come from the message headers on the device, and the
// slots behind them are marked absent (x = NaN: build_cell_grid gives such an atom no cell; y = z = extras = 0, id = -1).  A
// header that announces more atoms than `cap` (the sender kept its first cap) sets a word of pinned host memory that
// mdh_slab_overflow_check reads.
namespace mdh {
static thread_local int *g_slab_overflow = nullptr;
__global__ __launch_bounds__(256) void k_slab_append_static(const double *__restrict__ msg_l, const double *__restrict__ msg_r, int64_t cap,
                                                            SlabColumns cols, int64_t *__restrict__ gid, int64_t n_owned, int *__restrict__ overflow)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * cap)
        return;
    const bool left = i < cap;
    const double *m = left ? msg_l : msg_r;
    const int64_t j = left ? i : i - cap;
    int64_t cnt = (int64_t)m[0];
    if (cnt > cap) {
        if (j == 0) *overflow = 1;
        cnt = cap;
    }
    if (j < cnt) {
        for (int k = 0; k < cols.n; ++k) cols.p[k][n_owned + i] = m[1 + (int64_t)k * cap + j];
        gid[n_owned + i] = (int64_t)m[1 + (int64_t)cols.n * cap + j];
    } else {
        cols.p[0][n_owned + i] = __builtin_nan("");
        for (int k = 1; k < cols.n; ++k) cols.p[k][n_owned + i] = 0.0;
        gid[n_owned + i] = -1;
    }
}
} // namespace mdh

extern "C" int mdh_slab_append_ghosts_static(const double *msg_left, const double *msg_right, int64_t cap, double *const *columns, int ncol,
                                             int64_t *gid, int64_t n_owned, void *stream)
{
    if (!msg_left || !msg_right || !columns || !gid || ncol < 3 || ncol > 7 || cap < 0 || n_owned < 0) {
        set_error("mdh_slab_append_ghosts_static: bad arguments");
        return MDH_ERR_ARG;
    }
    if (!g_slab_overflow) {
        MDH_HIP(hipHostMalloc(reinterpret_cast<void **>(&g_slab_overflow), sizeof(int), hipHostMallocDefault));
        *g_slab_overflow = 0;
    }
    SlabColumns c{{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, ncol};
    for (int k = 0; k < ncol; ++k) c.p[k] = columns[k];
    if (cap > 0)
        hipLaunchKernelGGL(k_slab_append_static, dim3(grid_for(2 * cap, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), msg_left, msg_right, cap, c,
                           gid, n_owned, g_slab_overflow);
    MDH_HIP(hipGetLastError());
    return MDH_OK;
}

// MDH_ERR_ARG (and the flag cleared) if a halo message appended by this thread's mdh_slab_append_ghosts_static calls announced more
// atoms than its capacity since the last check; sees the appends that have COMPLETED on the device (no synchronisation here)
extern "C" int mdh_slab_overflow_check(void)
{
    if (g_slab_overflow && *(volatile int *)g_slab_overflow != 0) {
        *g_slab_overflow = 0;
        set_error("a halo message announced more atoms than the agreed capacity: the ghost block of that step was cut short");
        return MDH_ERR_ARG;
    }
    return MDH_OK;
}

extern "C" int mdh_slab_append_ghosts(const double *msg_left, const double *msg_right, int64_t cap, int64_t nl, int64_t nr,
                                      double *const *columns, int ncol, int64_t *gid, int64_t n_owned, void *stream)
{
    if (!msg_left || !msg_right || !columns || !gid || ncol < 3 || ncol > 7 || cap < 0 || nl < 0 || nr < 0 || nl > cap || nr > cap || n_owned < 0) {
        set_error("mdh_slab_append_ghosts: bad arguments");
        return MDH_ERR_ARG;
    }
    SlabColumns c{{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, ncol};
    for (int k = 0; k < ncol; ++k) c.p[k] = columns[k];
    if (nl + nr > 0)
        hipLaunchKernelGGL(k_slab_append, dim3(grid_for(nl + nr, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), msg_left, msg_right, cap, nl, nr, c,
                           gid, n_owned);
    MDH_HIP(hipGetLastError());
    return MDH_OK;
}

// up / down: (capacity) i32 each; counts_host[0..1] receive the numbers of selected atoms — if one of them exceeds `capacity`
// the buffers hold the first `capacity` selections only and the caller repeats the call with larger ones (a slab's halo is a
// few per cent of its atoms: buffers sized for all of them would be ~72 B per owned atom of transient memory per step).  hi3 = column `axis` of the
// inverse box matrix, origin3: host arrays.  gid (n) i64 with up_pack / down_pack (n, 4) f64, or all three NULL: the
// selected atoms' (x, y, z, id) rows, ready to be sent (ids < 2^53 are exact in f64).  Synchronises the stream (the counts
// size the exchange that follows).
extern "C" int mdh_slab_halo_select(const double *x, const double *y, const double *z, int64_t n, const double *origin3,
                                    const double *hi3, double up_from, double down_below, int *up, int *down,
                                    int64_t *counts_host, const int64_t *gid, double *up_pack, double *down_pack, int64_t capacity,
                                    int space, void *stream)
{
    if ((gid || up_pack || down_pack) && !(gid && up_pack && down_pack)) {
        set_error("mdh_slab_halo_select: gid, up_pack and down_pack go together");
        return MDH_ERR_ARG;
    }
    if (n < 0 || n >= 2147483647LL || !counts_host || capacity < 0) { set_error("mdh_slab_halo_select: bad arguments"); return MDH_ERR_ARG; }
    if (capacity > n) capacity = n;
    counts_host[0] = counts_host[1] = 0;
    if (n == 0)
        return MDH_OK;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const double *dx = sc.stage_in(x, (size_t)n, space), *dy = sc.stage_in(y, (size_t)n, space), *dz = sc.stage_in(z, (size_t)n, space);
    int *du = sc.stage(up, (size_t)capacity, space, false, true), *dd = sc.stage(down, (size_t)capacity, space, false, true);
    int *dc = sc.alloc_n<int>(2);
    const int64_t *dg = gid ? sc.stage_in(gid, (size_t)n, space) : nullptr;
    double *dup = up_pack ? sc.stage(up_pack, (size_t)capacity * 4, space, false, true) : nullptr;
    double *ddp = down_pack ? sc.stage(down_pack, (size_t)capacity * 4, space, false, true) : nullptr;
    if (sc.failed())
        return sc.error();
    MDH_HIP(hipMemsetAsync(dc, 0, 2 * sizeof(int), st));
    hipLaunchKernelGGL(k_slab_select, dim3(grid_for(n, 256)), dim3(256), 0, st, dx, dy, dz, n, origin3[0], origin3[1], origin3[2],
                       hi3[0], hi3[1], hi3[2], up_from, down_below, du, dd, dc, dg, dup, ddp, (int)capacity);
    int c[2] = {0, 0};
    MDH_HIP(hipMemcpyAsync(c, dc, sizeof(c), hipMemcpyDeviceToHost, st));
    MDH_HIP(hipStreamSynchronize(st));
    counts_host[0] = c[0];
    counts_host[1] = c[1];
    return sc.finish(space);
}

MDH_WARM_UNIT(slab)
