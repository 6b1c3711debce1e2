// pft.hip — FCC planar faults on gfx950: stacking faults and twin boundaries among the HCP-labelled atoms of an FCC
// crystal, from the PTM labels and PTM neighbour rows.
//
// Replaces src/identify_fcc_planar_faults.cpp:54-245 (identify_sftb_fcc).  Passes 1, 2 and 4 are per-atom and run one
// thread per HCP atom.  Pass 3 of the reference is a SERIAL sweep in index order whose iterations read and write the
// labels of their neighbours (:139-180) — its result depends on the order.  It is reproduced exactly with
// priority-ordered rounds: iteration k touches only {k} + N(k); it may run as soon as every lower-numbered iteration that
// touches any of those atoms has finished.  Per round, every pending j publishes atomicMin(owner[x], j) for x in
// {j} + N(j); k is ready iff it owns all of {k} + N(k).  Ready iterations of one round have disjoint footprints, so they
// can run concurrently, and every iteration sees exactly the labels the serial sweep would have shown it.
#include "common.hpp"

namespace mdh {

__device__ __constant__ int c_layer_dir[12] = {0, 0, -1, -1, -1, 0, 0, 0, 0, 1, 1, 1};
__device__ __constant__ int c_basal[6] = {0, 1, 5, 6, 7, 8};
__device__ __constant__ int c_oop[6] = {2, 3, 4, 9, 10, 11};

__device__ __forceinline__ int pft_bsearch(const int *__restrict__ arr, int n, int value) // :30-45 (returns mid even on a miss)
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

__global__ __launch_bounds__(256) void k_pft_map(const int *__restrict__ hcp_idx, int n_hcp, int *__restrict__ hn,
                                                 const int *__restrict__ ptm12, const int *__restrict__ stype)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_hcp * 12)
        return;
    const int i = t / 12, j = t % 12;
    const int b = ptm12[(int64_t)hcp_idx[i] * 12 + j];
    hn[t] = stype[b] == 2 ? pft_bsearch(hcp_idx, n_hcp, b) : -b - 1; // :72-86
}

__global__ __launch_bounds__(256) void k_pft_classify(const int *__restrict__ hcp_idx, int n_hcp, const int *__restrict__ hn,
                                                      const int *__restrict__ stype, int *__restrict__ fault)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hcp)
        return;
    int mine[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) mine[a] = hn[i * 12 + c_basal[a]];
    int nb = 0, np = 0, nn = 0, fp = 0, fn = 0;
    for (int j = 0; j < 12; ++j) { // :88-124
        const int q = hn[i * 12 + j];
        const int dir = c_layer_dir[j];
        if (q >= 0) {
            if (dir == 0) {
                ++nb;
            } else {
                bool stacked = true; // no common basal neighbour (are_stacked, :9-23)
                for (int b = 0; b < 6; ++b) {
                    const int theirs = hn[q * 12 + c_basal[b]];
#pragma unroll
                    for (int a = 0; a < 6; ++a) stacked = stacked && (mine[a] != theirs);
                }
                if (stacked) { if (dir == 1) ++np; else ++nn; }
            }
        } else if (dir != 0) {
            if (stype[-q - 1] == 1) { if (dir > 0) ++fp; else ++fn; }
        }
    }
    int f;
    if ((np != 0 && nn == 0) || (np == 0 && nn != 0)) f = 2;      // :126-137
    else if (nb >= 1 && np == 0 && nn == 0 && fp != 0 && fn != 0) f = 3;
    else if (np != 0 && nn != 0) f = 4;
    else f = 1;
    fault[hcp_idx[i]] = f;
}

// ---- pass 3: ordered sweep in rounds ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pft_publish(int n_hcp, const int *__restrict__ hn, const unsigned char *__restrict__ done,
                                                     int *__restrict__ owner)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_hcp || done[j])
        return;
    atomicMin(&owner[j], j);
    for (int c = 0; c < 12; ++c) {
        const int q = hn[j * 12 + c];
        if (q >= 0) atomicMin(&owner[q], j);
    }
}

__global__ __launch_bounds__(256) void k_pft_sweep(const int *__restrict__ hcp_idx, int n_hcp, const int *__restrict__ hn,
                                                   unsigned char *__restrict__ done, const int *__restrict__ owner,
                                                   int *__restrict__ fault, int *__restrict__ remaining)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_hcp || done[k])
        return;
    bool ready = owner[k] == k;
    for (int c = 0; c < 12 && ready; ++c) {
        const int q = hn[k * 12 + c];
        if (q >= 0) ready = owner[q] == k;
    }
    if (!ready) {
        atomicAdd(remaining, 1);
        return;
    }
    const int a = hcp_idx[k];
    const int f = fault[a];
    if (f == 3 || f == 1) { // :143-165
        int nisf = 0, ntw = 0;
        for (int jj = 0; jj < 6; ++jj) {
            const int q = hn[k * 12 + c_basal[jj]];
            if (q >= 0) {
                const int nf = fault[hcp_idx[q]];
                if (nf == 2) ++nisf; else if (nf == 3) ++ntw;
            }
        }
        if (nisf != 0 && ntw == 0) fault[a] = 2;
        else if (nisf == 0 && ntw != 0) fault[a] = 3;
    } else if (f == 4) { // :166-179
        for (int jj = 0; jj < 6; ++jj) {
            const int q = hn[k * 12 + c_oop[jj]];
            if (q >= 0 && fault[hcp_idx[q]] == 2) fault[hcp_idx[q]] = 4;
        }
    }
    done[k] = 1;
}

__global__ __launch_bounds__(256) void k_pft_esf(const int *__restrict__ hcp_idx, int n_hcp, const int *__restrict__ ptm12,
                                                 const int *__restrict__ stype, int *__restrict__ fault)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hcp)
        return;
    const int a = hcp_idx[i];
    if (fault[a] != 3) // :188: only twin-boundary atoms are examined, and the pass only turns 3 into 5
        return;
    for (int j = 0; j < 12; ++j) {
        const int jn = ptm12[(int64_t)a * 12 + j];
        if (stype[jn] != 1)
            continue;
        int fc = 0, hc = 0;
        for (int k = 0; k < 12; ++k) {
            const int t = stype[ptm12[(int64_t)jn * 12 + k]];
            fc += t == 1; hc += t == 2;
        }
        if (fc >= 5 && fc <= 6 && hc >= 5 && hc <= 6) { fault[a] = 5; return; }
    }
}

__global__ void k_fill_i32(int *p, int n, int v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

} // namespace mdh

using namespace mdh;

// replaces _fccpft.identify_sftb_fcc                     src/identify_fcc_planar_faults.cpp:54-245
// hcp_indices (n_hcp) ascending; hcp_neighbors (n_hcp,12) scratch owned by the caller (written); ptm_indices (N,12);
// structure_types (N); fault_types (N) pre-zeroed by the caller, entries of HCP atoms are written.
extern "C" int mdh_identify_sftb_fcc(const int *hcp_indices, int64_t n_hcp, int *hcp_neighbors, const int *ptm_indices,
                                     const int *structure_types, int64_t N, int *fault_types, int identify_esf, int space,
                                     void *stream)
{
    if (n_hcp < 0 || N < 0 || n_hcp > N || N > 2147483647LL / 12) {
        set_error("mdh_identify_sftb_fcc: bad sizes");
        return MDH_ERR_ARG;
    }
    if (n_hcp == 0)
        return MDH_OK;
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const int nh = (int)n_hcp;
    const int *dh = sc.stage_in(hcp_indices, (size_t)n_hcp, space);
    int *dhn = sc.stage(hcp_neighbors, (size_t)n_hcp * 12, space, false, true);
    const int *dp = sc.stage_in(ptm_indices, (size_t)N * 12, space);
    const int *ds = sc.stage_in(structure_types, (size_t)N, space);
    int *df = sc.stage(fault_types, (size_t)N, space, true, true);
    unsigned char *done = sc.alloc_n<unsigned char>((size_t)n_hcp);
    int *owner = sc.alloc_n<int>((size_t)n_hcp);
    int *remaining = sc.alloc_n<int>(1);
    if (sc.failed())
        return sc.error();
    const dim3 block(256), grid(grid_for(n_hcp, 256));
    hipLaunchKernelGGL(k_pft_map, dim3(grid_for(n_hcp * 12, 256)), block, 0, st, dh, nh, dhn, dp, ds);
    hipLaunchKernelGGL(k_pft_classify, grid, block, 0, st, dh, nh, dhn, ds, df);
    MDH_HIP(hipMemsetAsync(done, 0, (size_t)n_hcp, st));
    for (int64_t round = 0; round <= n_hcp; ++round) { // every round retires at least the lowest pending iteration
        hipLaunchKernelGGL(k_fill_i32, grid, block, 0, st, owner, nh, 0x7fffffff);
        MDH_HIP(hipMemsetAsync(remaining, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_pft_publish, grid, block, 0, st, nh, dhn, done, owner);
        hipLaunchKernelGGL(k_pft_sweep, grid, block, 0, st, dh, nh, dhn, done, owner, df, remaining);
        int left = 0;
        MDH_HIP(hipMemcpyAsync(&left, remaining, sizeof(int), hipMemcpyDeviceToHost, st));
        MDH_HIP(hipStreamSynchronize(st));
        if (left == 0)
            break;
    }
    if (identify_esf)
        hipLaunchKernelGGL(k_pft_esf, grid, block, 0, st, dh, nh, dp, ds, df);
    MDH_HIP(hipGetLastError());
    return sc.finish(space);
}

MDH_WARM_UNIT(pft)
