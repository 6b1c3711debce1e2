// runtime.hip — error reporting, per-device scratch cache, host<->HBM staging,
// and the host-side construction of the kernel-visible box.
#include "common.hpp"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

extern "C" int mdh_warm(void);

namespace mdh {

static thread_local std::string g_err;

void set_error(const std::string &msg) { g_err = msg; }

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what + " (" + file + ":" + std::to_string(line) + ")";
    return MDH_ERR_HIP;
}

// ----------------------------------------------------------------------------
// scratch cache: blocks are handed out best-fit and returned at Scope exit
//
// A DEVICE-space call returns while its kernels are still queued, and its Scope gives the blocks back at that moment.
// Reuse from the SAME stream is ordered by the stream; reuse from another stream (another host thread, a side stream of the
// caller) is not.  Every Scope therefore records ONE event on its stream when it ends; the blocks it held remember that
// event, and a later Scope on a different stream makes its stream wait for it before touching the block.
// ----------------------------------------------------------------------------
struct DoneEvent { hipEvent_t ev; hipStream_t stream; int refs; };
struct Block { void *p; size_t bytes; bool busy; int device; DoneEvent *done; int tag; }; // tag: Scope::Keep (0: plain scratch)
static std::mutex g_mu;
static std::vector<Block> g_blocks;
static std::vector<DoneEvent *> g_event_pool;

static DoneEvent *event_acquire(hipStream_t st) // g_mu held
{
    DoneEvent *e = nullptr;
    if (!g_event_pool.empty()) {
        e = g_event_pool.back();
        g_event_pool.pop_back();
    } else {
        e = new DoneEvent{nullptr, nullptr, 0};
        if (hipEventCreateWithFlags(&e->ev, hipEventDisableTiming) != hipSuccess) { delete e; return nullptr; }
    }
    e->stream = st;
    e->refs = 0;
    return e;
}

static void event_release(DoneEvent *e) // g_mu held
{
    if (e && --e->refs == 0)
        g_event_pool.push_back(e);
}

Scope::Scope(void *stream) : stream_(static_cast<hipStream_t>(stream))
{
    device_ = 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) { // no CPU fallback: fail loudly
        set_error(std::string("HIP error: no ROCm device available (") + hipGetErrorString(e) + "); mdapy_amd needs an AMD GPU (gfx950)");
        failed_ = true;
        err_ = MDH_ERR_HIP;
        return;
    }
    (void)hipGetDevice(&device_);
    // the first call of the process on this device loads every code object of the library (mdh_warm): the set-up cost of a
    // process belongs to its first touch of a device, wherever the caller selected it, not to whichever kernels come first
    static std::atomic<unsigned long long> warmed{0};
    static const bool want = [] { const char *e = std::getenv("MDAPY_HIP_WARM"); return !(e && e[0] == '0'); }();
    if (want && device_ < 64 && !((warmed.load(std::memory_order_relaxed) >> device_) & 1ull)) {
        warmed.fetch_or(1ull << device_, std::memory_order_relaxed);
        (void)mdh_warm();
    }
}

Scope::~Scope()
{
    if (nheld_ == 0)
        return;
    // a kept block whose invariant was not restored (an error path between its first and its last kernel): zero-fill it, in
    // stream order, before the event that marks the end of this call
    for (int i = 0; i < nheld_; ++i)
        if (held_keep_[i] == KEEP_ZERO || held_keep_[i] == KEEP_TODO)
            (void)hipMemsetAsync(held_[i], 0, held_keep_[i] == KEEP_TODO ? std::min<size_t>(512, held_bytes_[i]) : held_bytes_[i], stream_);
    std::lock_guard<std::mutex> lk(g_mu);
    DoneEvent *done = event_acquire(stream_);
    if (done && hipEventRecord(done->ev, stream_) != hipSuccess) { // cannot mark the end of this call's work: wait for it instead
        (void)hipStreamSynchronize(stream_);
        g_event_pool.push_back(done);
        done = nullptr;
    } else if (!done) {
        (void)hipStreamSynchronize(stream_);
    }
    for (int i = 0; i < nheld_; ++i)
        for (auto &b : g_blocks)
            if (b.p == held_[i]) {
                event_release(b.done);
                b.done = done;
                if (done) ++done->refs;
                b.busy = false;
                break;
            }
    if (done && done->refs == 0)
        g_event_pool.push_back(done);
}

void *Scope::alloc(size_t bytes) { return alloc_impl(bytes, KEEP_NONE); }

void *Scope::alloc_kept(size_t bytes, Keep tag) { return alloc_impl(bytes, (int)tag); }

void Scope::keep_confirm(void *p)
{
    for (int i = 0; i < nheld_; ++i)
        if (held_[i] == p) held_keep_[i] |= 0x80;
}

void *Scope::alloc_impl(size_t bytes, int tag)
{
    if (failed_)
        return nullptr;
    if (nheld_ >= kMaxHeld) { set_error("internal: too many scratch buffers in one call"); failed_ = true; return nullptr; }
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    std::lock_guard<std::mutex> lk(g_mu);
    // Best fit among the idle blocks this stream used last (or nobody did): no waiting.  A block last used on ANOTHER stream
    // costs a wait for that stream's call to end — the caller's side stream packing halo messages would hold up the neighbor
    // build on the main stream for a few counters' worth of scratch — so small requests get a block of their own instead
    // (once per stream and size), and only large ones share across streams.
    int best = -1, best_any = -1;
    for (size_t i = 0; i < g_blocks.size(); ++i) {
        const Block &b = g_blocks[i];
        if (b.busy || b.device != device_ || b.bytes < bytes || b.tag != tag)
            continue;
        if (best_any < 0 || b.bytes < g_blocks[best_any].bytes)
            best_any = (int)i;
        if ((!b.done || b.done->stream == stream_) && (best < 0 || b.bytes < g_blocks[best].bytes))
            best = (int)i;
    }
    const bool fits = best >= 0 && g_blocks[best].bytes <= 2 * bytes + (1u << 20);
    if (!fits && bytes > (size_t(4) << 20))
        best = best_any;
    if (tag != KEEP_NONE && best < 0)
        best = best_any; // kept blocks: any idle one of the tag that is large enough (only the part a call touches matters)
    // reuse only if the block is not grossly oversized (keeps big list buffers from being pinned by tiny requests)
    if (best >= 0 && (tag != KEEP_NONE || g_blocks[best].bytes <= 2 * bytes + (1u << 20))) {
        Block &b = g_blocks[best];
        if (b.done && b.done->stream != stream_) // last used on another stream: its work there comes first
            (void)hipStreamWaitEvent(stream_, b.done->ev, 0);
        b.busy = true;
        held_[nheld_] = b.p;
        held_keep_[nheld_] = (unsigned char)tag;
        held_bytes_[nheld_++] = b.bytes;
        return b.p;
    }
    if (tag != KEEP_NONE) // a new kept block: with head-room, so that a slowly growing system does not make one per size
        bytes = (bytes + bytes / 8 + 4095) & ~size_t(4095);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // drop idle cached blocks of this device and retry once (hipFree waits for the device: nothing is in flight after it)
        for (size_t i = 0; i < g_blocks.size();) {
            if (!g_blocks[i].busy && g_blocks[i].device == device_) { (void)hipFree(g_blocks[i].p); event_release(g_blocks[i].done); g_blocks.erase(g_blocks.begin() + i); }
            else ++i;
        }
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
        set_error(std::string("device allocation of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
        failed_ = true;
        return nullptr;
    }
    if (tag != KEEP_NONE) { // every kept block starts zero-filled (stream-ordered: its first user is this call)
        e = hipMemsetAsync(p, 0, bytes, stream_);
        if (e != hipSuccess) { (void)hipFree(p); hip_fail(e, "hipMemsetAsync(kept block)", __FILE__, __LINE__); failed_ = true; err_ = MDH_ERR_HIP; return nullptr; }
    }
    g_blocks.push_back(Block{p, bytes, true, device_, nullptr, tag});
    held_[nheld_] = p;
    held_keep_[nheld_] = (unsigned char)tag;
    held_bytes_[nheld_++] = bytes;
    return p;
}

// every cached block of one Keep tag back to all-zero (waits for the device; for the rare generation wrap of the scan)
void reset_kept_blocks(int tag)
{
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &b : g_blocks)
        if (b.tag == tag) (void)hipMemset(b.p, 0, b.bytes);
}

void *Scope::stage_raw(void *p, size_t bytes, int space, bool in, bool out)
{
    if (space == MDH_DEVICE || p == nullptr)
        return p;
    void *d = alloc(bytes);
    if (!d)
        return nullptr;
    if (in && bytes) {
        hipError_t e = hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, stream_);
        if (e != hipSuccess) { hip_fail(e, "hipMemcpyAsync(H2D)", __FILE__, __LINE__); failed_ = true; err_ = MDH_ERR_HIP; return nullptr; }
    }
    if (out) {
        if (nouts_ >= 16) { set_error("internal: too many staged outputs"); failed_ = true; return nullptr; }
        outs_[nouts_++] = Out{p, d, bytes};
    }
    return d;
}

__global__ __launch_bounds__(256) void k_pack_positions(const double *__restrict__ x, const double *__restrict__ y,
                                                        const double *__restrict__ z, int64_t N, Pos4 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = Pos4{x[i], y[i], z[i], 0.0};
}

const Pos4 *pack_positions(Scope &sc, const double *x, const double *y, const double *z, int64_t N)
{
    Pos4 *out = sc.alloc_n<Pos4>((size_t)N);
    if (!out)
        return nullptr;
    hipLaunchKernelGGL(k_pack_positions, dim3(grid_for(N, 256)), dim3(256), 0, sc.stream(), x, y, z, N, out);
    return out;
}

OrderHint order_hint(int tag, int64_t a, int64_t b, const void *array)
{
    struct Entry { int tag; int64_t a, b; const void *array; int device; int *word; unsigned calls; };
    static std::mutex mu;
    static std::vector<Entry> table;
    int device = 0;
    (void)hipGetDevice(&device);
    std::lock_guard<std::mutex> lk(mu);
    int guess = 0; // the latest answer for this shape, whatever the array: frames of one trajectory look alike
    for (size_t k = table.size(); k-- > 0;) {
        Entry &e = table[k];
        if (e.tag != tag || e.a != a || e.b != b || e.device != device)
            continue;
        if (e.array == array)
            return OrderHint{e.word, (++e.calls & 7u) == 0};
        if (guess == 0) guess = *(volatile int *)e.word != 0 ? 1 : -1;
    }
    int *word = nullptr;
    if (table.size() >= 64) { // the oldest signature hands its word on (never freed: a kernel enqueued earlier may still write it — a wrong hint at worst)
        word = table.front().word;
        table.erase(table.begin());
    } else if (hipHostMalloc(reinterpret_cast<void **>(&word), sizeof(int), hipHostMallocDefault) != hipSuccess) {
        return OrderHint{nullptr, false};
    }
    *word = guess > 0 ? 1 : 0;
    table.push_back(Entry{tag, a, b, array, device, word, 0u});
    return OrderHint{word, true};
}

int Scope::finish(int space)
{
    if (failed_)
        return err_;
    MDH_HIP(hipGetLastError());
    if (space == MDH_HOST) {
        for (int i = 0; i < nouts_; ++i)
            if (outs_[i].bytes)
                MDH_HIP(hipMemcpyAsync(outs_[i].host, outs_[i].dev, outs_[i].bytes, hipMemcpyDeviceToHost, stream_));
        MDH_HIP(hipStreamSynchronize(stream_));
    }
    return MDH_OK;
}

// ----------------------------------------------------------------------------
// box
// ----------------------------------------------------------------------------
static double det3(const double *d, bool tri) // box.h:22-35
{
    if (tri)
        return d[0] * (d[4] * d[8] - d[5] * d[7]) - d[1] * (d[3] * d[8] - d[5] * d[6]) + d[2] * (d[3] * d[7] - d[4] * d[6]);
    return d[0] * d[4] * d[8];
}

// total order preserving map double <-> int64 (for bisection over representable doubles)
static int64_t to_key(double v)
{
    int64_t k;
    std::memcpy(&k, &v, 8);
    return k < 0 ? (int64_t)(0x8000000000000000ull - (uint64_t)k) : k;
}
static double from_key(int64_t k)
{
    if (k < 0) k = (int64_t)(0x8000000000000000ull - (uint64_t)k);
    double v;
    std::memcpy(&v, &k, 8);
    return v;
}

// smallest double d with floor(d / L + 0.5) >= n   (L > 0)
static double image_threshold(double L, double n)
{
    auto f = [L](double d) { return std::floor(d / L + 0.5); };
    int64_t lo = to_key(-8.0 * L), hi = to_key(8.0 * L); // f(lo) < n <= f(hi) for n in [-1,2]
    while (lo + 1 < hi) { // keys of opposite sign: hi - lo would overflow int64
        const int64_t mid = (lo >> 1) + (hi >> 1) + (lo & hi & 1);
        if (f(from_key(mid)) >= n) hi = mid; else lo = mid;
    }
    return from_key(hi);
}

int make_box(DBox &b, const double *box9, const double *origin3, const int *boundary3)
{
    std::memset(&b, 0, sizeof(b));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            b.h[i * 3 + j] = box9[i * 3 + j];
            if (i != j && std::fabs(box9[i * 3 + j]) > 1e-10) // box.h:216-218
                b.tri = 1;
        }
    if (b.h[0] < 0 || b.h[4] < 0 || b.h[8] < 0) // box.h:221-222
        b.tri = 1;
    if (b.tri) {
        double det = det3(b.h, true);
        if (std::fabs(det) < 1e-12) { // box.h:185-186
            set_error("The volume of the box is zero.");
            return MDH_ERR_BOX;
        }
        const double id = 1.0 / det;
        const double *m = b.h;
        b.hi[0] = (m[4] * m[8] - m[5] * m[7]) * id;
        b.hi[1] = -(m[1] * m[8] - m[2] * m[7]) * id;
        b.hi[2] = (m[1] * m[5] - m[2] * m[4]) * id;
        b.hi[3] = -(m[3] * m[8] - m[5] * m[6]) * id;
        b.hi[4] = (m[0] * m[8] - m[2] * m[6]) * id;
        b.hi[5] = -(m[0] * m[5] - m[2] * m[3]) * id;
        b.hi[6] = (m[3] * m[7] - m[4] * m[6]) * id;
        b.hi[7] = -(m[0] * m[7] - m[1] * m[6]) * id;
        b.hi[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    } else {
        b.hi[0] = 1.0 / b.h[0];
        b.hi[4] = 1.0 / b.h[4];
        b.hi[8] = 1.0 / b.h[8];
    }
    for (int i = 0; i < 3; ++i) {
        b.o[i] = origin3[i];
        b.pbc[i] = boundary3[i] ? 1 : 0;
    }
    b.anypbc = b.pbc[0] || b.pbc[1] || b.pbc[2];
    for (int dir = 0; dir < 3; ++dir) { // box.h:54-89
        if (!b.tri) { b.thick[dir] = b.h[dir * 4]; continue; }
        const double V = det3(b.h, true);
        const double *A = b.h, *B = b.h + 3, *C = b.h + 6;
        const double *p = dir == 0 ? B : A;
        const double *q = dir == 2 ? B : C;
        const double mm = p[1] * q[2] - p[2] * q[1];
        const double nn = p[2] * q[0] - p[0] * q[2];
        const double kk = p[0] * q[1] - p[1] * q[0];
        b.thick[dir] = V / std::sqrt(mm * mm + nn * nn + kk * kk);
    }
    for (int a = 0; a < 3; ++a) {
        const double L = b.h[a * 4];
        if (!b.tri && b.pbc[a] && L > 0 && std::isfinite(L)) {
            for (int k = 0; k < 4; ++k)
                b.tn[a][k] = image_threshold(L, (double)(k - 1));
        } else { // empty fast range => kernels always take the exact division path
            b.tn[a][0] = 1.0; b.tn[a][1] = 0.0; b.tn[a][2] = 0.0; b.tn[a][3] = 0.0;
        }
    }
    return MDH_OK;
}

} // namespace mdh

namespace mdh {
void warm_prof(hipStream_t st);
void warm_neighbor(hipStream_t st);
void warm_neighbor_tiled(hipStream_t st);
void warm_neighbor_lane(hipStream_t st);
void warm_cna(hipStream_t st);
void warm_csp(hipStream_t st);
void warm_sbo(hipStream_t st);
void warm_rdf(hipStream_t st);
void warm_wcp(hipStream_t st);
void warm_knn(hipStream_t st);
void warm_repeat(hipStream_t st);
void warm_ptm(hipStream_t st);
void warm_ptm_stages(hipStream_t st);
void warm_consumers(hipStream_t st);
void warm_pft(hipStream_t st);
void warm_voronoi(hipStream_t st);
void warm_sfc(hipStream_t st);
void warm_polycrystal(hipStream_t st);
void warm_slab(hipStream_t st);
void warm_text(hipStream_t st);
}
namespace mdh {
__global__ void k_warm_runtime() {}
// a launch that needs 256 bytes of private (scratch) memory per lane: the queue's scratch area is allocated at the first launch
// that needs one, and again when a later kernel needs more per lane (1-17 ms on this chip); the stand-by kernels of the
// fixed-cutoff CNA (244 B: its double-precision to-do pass) and the k-nearest search (up to 120 B of spills) are such kernels
__global__ void k_warm_scratch(int n, int *out)
{
    volatile int a[64];
    for (int i = 0; i < n; ++i) a[(i * 7) & 63] = i;
    if (n > 0) out[0] = a[n & 63];
}
// smallest and largest entry of an int32 array: out[0], out[1] preset to INT_MAX / INT_MIN
__global__ __launch_bounds__(256) void k_min_max_i32(const int *__restrict__ v, int64_t n, int *__restrict__ out)
{
    int lo = 2147483647, hi = -2147483647 - 1;
    // 16 bytes per lane and trip where the array allows (it is 256-byte aligned when it is one of the package's own): 54 us for
    // 10 M entries with one entry per lane and trip, a fifth of that so
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(v) & 15u) == 0) ? n / 4 : 0;
    const int4 *v4 = reinterpret_cast<const int4 *>(v);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int4 q = v4[i];
        lo = min(min(lo, q.x), min(q.y, min(q.z, q.w)));
        hi = max(max(hi, q.x), max(q.y, max(q.z, q.w)));
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = v[i];
        lo = min(lo, x);
        hi = max(hi, x);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, __shfl_xor(lo, d, 64));
        hi = max(hi, __shfl_xor(hi, d, 64));
    }
    // one pair of atomics per workgroup: 4096 waves each reading and updating the two words took 43-80 us on 4 M entries
    // (operations on one address queue up in L2), the array itself is 5 us of traffic
    __shared__ int s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(out, min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])));
        atomicMax(out + 1, max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3])));
    }
}
__global__ void k_min_max_init(int *out) { out[0] = 2147483647; out[1] = -2147483647 - 1; }
// how often each value of [low, low + span) occurs (span <= 4096: a workgroup counts in LDS)
__global__ __launch_bounds__(256) void k_value_counts(const int *__restrict__ v, int64_t n, int low, int span, unsigned long long *__restrict__ counts)
{
    __shared__ unsigned h[4096];
    for (int k = threadIdx.x; k < span; k += blockDim.x) h[k] = 0u;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned d = (unsigned)(v[i] - low);
        if (d < (unsigned)span) atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < span; k += blockDim.x)
        if (h[k]) atomicAdd(&counts[k], (unsigned long long)h[k]);
}
__global__ __launch_bounds__(256) void k_apply_lut(const int *__restrict__ v, int64_t n, int low, int span, const int *__restrict__ lut, int *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const unsigned d = (unsigned)(v[i] - low);
    out[i] = d < (unsigned)span ? lut[d] : 0;
}
}

// ----------------------------------------------------------------------------
// C ABI: runtime
// ----------------------------------------------------------------------------
extern "C" {

const char *mdh_last_error(void) { return mdh::g_err.c_str(); }

int mdh_version(void) { return 100; }

int mdh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int mdh_warm(void)
{
    static std::mutex mu;
    static unsigned long long done = 0; // one bit per device
    int device = 0;
    MDH_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> lk(mu);
    if (device < 64 && ((done >> device) & 1ull))
        return MDH_OK;
    hipLaunchKernelGGL(mdh::k_warm_runtime, dim3(1), dim3(64), 0, nullptr);
    hipLaunchKernelGGL(mdh::k_warm_scratch, dim3(1), dim3(64), 0, nullptr, 0, static_cast<int *>(nullptr));
    mdh::warm_prof(nullptr);
    mdh::warm_neighbor(nullptr);
    mdh::warm_neighbor_tiled(nullptr);
    mdh::warm_neighbor_lane(nullptr);
    mdh::warm_cna(nullptr);
    mdh::warm_csp(nullptr);
    mdh::warm_sbo(nullptr);
    mdh::warm_rdf(nullptr);
    mdh::warm_wcp(nullptr);
    mdh::warm_knn(nullptr);
    mdh::warm_repeat(nullptr);
    mdh::warm_ptm(nullptr);
    mdh::warm_ptm_stages(nullptr);
    mdh::warm_consumers(nullptr);
    mdh::warm_pft(nullptr);
    mdh::warm_voronoi(nullptr);
    mdh::warm_sfc(nullptr);
    mdh::warm_polycrystal(nullptr);
    mdh::warm_slab(nullptr);
    mdh::warm_text(nullptr);
    MDH_HIP(hipGetLastError());
    MDH_HIP(hipStreamSynchronize(nullptr));
    if (device < 64) done |= 1ull << device;
    return MDH_OK;
}

int mdh_min_max_i32(const int *v, int64_t n, int *min_max2, int space, void *stream)
{
    if (n <= 0 || !min_max2) { mdh::set_error("mdh_min_max_i32: empty array"); return MDH_ERR_ARG; }
    mdh::Scope sc(stream);
    const int *dv = sc.stage_in(v, (size_t)n, space);
    int *out = sc.alloc_n<int>(2);
    if (sc.failed())
        return sc.error();
    hipStream_t st = sc.stream();
    // the answer lands in page-locked memory of this thread (a copy into pageable memory goes through the runtime's staging path)
    static thread_local int *pinned = nullptr;
    if (!pinned) MDH_HIP(hipHostMalloc(reinterpret_cast<void **>(&pinned), 2 * sizeof(int), hipHostMallocDefault));
    hipLaunchKernelGGL(mdh::k_min_max_init, dim3(1), dim3(1), 0, st, out);
    hipLaunchKernelGGL(mdh::k_min_max_i32, dim3((unsigned)std::min<int64_t>(512, (n + 1023) / 1024)), dim3(256), 0, st, dv, n, out);
    MDH_HIP(hipMemcpyAsync(pinned, out, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    MDH_HIP(hipStreamSynchronize(st));
    min_max2[0] = pinned[0];
    min_max2[1] = pinned[1];
    return MDH_OK;
}

int mdh_dense_codes_i32(const int *v, int64_t n, int low, int span, int *codes, int64_t *counts_span, int space, void *stream)
{
    if (n <= 0 || span <= 0 || span > 4096 || !codes || !counts_span) { mdh::set_error("mdh_dense_codes_i32: need n > 0 and 0 < span <= 4096"); return MDH_ERR_ARG; }
    mdh::Scope sc(stream);
    const int *dv = sc.stage_in(v, (size_t)n, space);
    int *dc = sc.stage(codes, (size_t)n, space, false, true);
    unsigned long long *dcnt = sc.alloc_n<unsigned long long>((size_t)span);
    int *dlut = sc.alloc_n<int>((size_t)span);
    if (sc.failed())
        return sc.error();
    hipStream_t st = sc.stream();
    static thread_local unsigned long long *pinned = nullptr; // counts down, lut up: page-locked, one block per thread
    if (!pinned) MDH_HIP(hipHostMalloc(reinterpret_cast<void **>(&pinned), 4096 * (sizeof(unsigned long long) + sizeof(int)), hipHostMallocDefault));
    int *lut = reinterpret_cast<int *>(pinned + 4096);
    MDH_HIP(hipMemsetAsync(dcnt, 0, sizeof(unsigned long long) * (size_t)span, st));
    hipLaunchKernelGGL(mdh::k_value_counts, dim3((unsigned)std::min<int64_t>(2048, (n + 2047) / 2048)), dim3(256), 0, st, dv, n, low, span, dcnt);
    MDH_HIP(hipMemcpyAsync(pinned, dcnt, sizeof(unsigned long long) * (size_t)span, hipMemcpyDeviceToHost, st));
    MDH_HIP(hipStreamSynchronize(st));
    int next = 0;
    for (int k = 0; k < span; ++k) {
        counts_span[k] = (int64_t)pinned[k];
        lut[k] = pinned[k] ? next++ : 0;
    }
    MDH_HIP(hipMemcpyAsync(dlut, lut, sizeof(int) * (size_t)span, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mdh::k_apply_lut, dim3(mdh::grid_for(n, 256)), dim3(256), 0, st, dv, n, low, span, dlut, dc);
    if (space == MDH_DEVICE)
        MDH_HIP(hipStreamSynchronize(st)); // (the pinned lut block is reused by this thread's next call)
    return sc.finish(space);
}

int mdh_set_device(int device)
{
    MDH_HIP(hipSetDevice(device));
    return mdh_warm();
}

int mdh_release_workspace(void)
{
    std::lock_guard<std::mutex> lk(mdh::g_mu);
    for (size_t i = 0; i < mdh::g_blocks.size();) {
        if (!mdh::g_blocks[i].busy) { (void)hipFree(mdh::g_blocks[i].p); mdh::event_release(mdh::g_blocks[i].done); mdh::g_blocks.erase(mdh::g_blocks.begin() + i); }
        else ++i;
    }
    return MDH_OK;
}

// test hook: the four minimum-image decision thresholds of an orthogonal periodic axis of length L
int mdh_debug_image_thresholds(double L, double *out4)
{
    mdh::DBox b;
    const double box9[9] = {L, 0, 0, 0, L, 0, 0, 0, L}, o[3] = {0, 0, 0};
    const int p[3] = {1, 1, 1};
    MDH_TRY(mdh::make_box(b, box9, o, p));
    for (int k = 0; k < 4; ++k) out4[k] = b.tn[0][k];
    return MDH_OK;
}

int64_t mdh_workspace_bytes(void)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mdh::g_mu);
    int64_t s = 0;
    for (auto &b : mdh::g_blocks)
        if (b.device == dev) s += (int64_t)b.bytes;
    return s;
}
}
