// wcp.hip — Warren-Cowley short-range-order parameter on gfx950.
//
// Replaces src/warren_cowley_parameter.cpp:9-80 (get_wcp).  Z_mn, Z_m and the
// per-type atom counts are integer reductions: u32 partials in LDS per
// workgroup, u64 totals in HBM (the reference uses int32 totals, which agree
// whenever they do not overflow), then alpha_ab = 1 - Z_ab / (c_b * Z_a).
#include "common.hpp"

namespace mdh {

static constexpr int WCP_MAXT = 64; // LDS budget: (T*T + 2T) u32

__global__ __launch_bounds__(256) void k_wcp_count(const int *__restrict__ verlet, const int *__restrict__ nn,
                                                   const int *__restrict__ type, const unsigned char *__restrict__ rows,
                                                   int64_t N, int64_t M, int T, unsigned long long *__restrict__ tot)
{
    extern __shared__ unsigned lds[]; // [T*T] Zmn, [T] Zm, [T] count
    const int words = T * T + 2 * T;
    for (int q = threadIdx.x; q < words; q += blockDim.x) lds[q] = 0u;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N && (!rows || rows[i])) { // rows: optional mask of the rows to count (owned atoms of a multi-GPU slab)
        const int ti = type[i], n = nn[i];
        atomicAdd(&lds[T * T + T + ti], 1u);
        atomicAdd(&lds[T * T + ti], (unsigned)n);
        // eight entries of the row at a time: their ids, then their types, requested together (entry by entry it was two
        // dependent memory latencies per neighbour: 94 % of the wave-cycles parked); with up to four types the counts stay in
        // registers until the row is done — 256 threads adding into T*T LDS words serialise on them
        unsigned c[4] = {0u, 0u, 0u, 0u};
        const bool few = T <= 4;
        for (int q0 = 0; q0 < n; q0 += 8) {
            int tj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tj[u] = safe_id(verlet[i * M + min(q0 + u, n - 1)], i, N);
#pragma unroll
            for (int u = 0; u < 8; ++u) tj[u] = type[tj[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (q0 + u >= n) continue;
                if (few) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) c[k] += tj[u] == k ? 1u : 0u;
                } else {
                    atomicAdd(&lds[ti * T + tj[u]], 1u);
                }
            }
        }
        if (few) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < T && c[k]) atomicAdd(&lds[ti * T + k], c[k]);
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < words; q += blockDim.x) {
        const unsigned v = lds[q];
        if (v) atomicAdd(&tot[q], (unsigned long long)v);
    }
}

__global__ void k_wcp_final(const unsigned long long *__restrict__ tot, int64_t N, int T, double *__restrict__ wcp)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= T * T)
        return;
    const int a = q / T, c = q % T;
    const double conc = (double)tot[T * T + T + c] / (double)N; // :57-59
    const unsigned long long zm = tot[T * T + a];
    wcp[q] = (conc > 0 && zm > 0) ? 1.0 - (double)tot[q] / (conc * (double)zm) : 0.0; // :66-75
}

} // namespace mdh

using namespace mdh;

extern "C" int mdh_wcp(const int *verlet, const int *nn, const int *type, int64_t N, int64_t M, int ntype,
                       double *wcp, int space, void *stream)
{
    if (N <= 0 || M <= 0 || ntype <= 0 || ntype > WCP_MAXT) { set_error("mdh_wcp: need N>0, M>0 and 1 <= ntype <= 64"); return MDH_ERR_ARG; }
    Scope sc(stream);
    hipStream_t st = sc.stream();
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const int *dn = sc.stage_in(nn, (size_t)N, space);
    const int *dt = sc.stage_in(type, (size_t)N, space);
    double *dw = sc.stage(wcp, (size_t)ntype * ntype, space, false, true);
    const int words = ntype * ntype + 2 * ntype;
    unsigned long long *tot = sc.alloc_n<unsigned long long>((size_t)words);
    if (sc.failed())
        return sc.error();
    MDH_HIP(hipMemsetAsync(tot, 0, sizeof(unsigned long long) * (size_t)words, st));
    hipLaunchKernelGGL(k_wcp_count, dim3(grid_for(N, 256)), dim3(256), sizeof(unsigned) * (size_t)words, st, dv, dn, dt, nullptr, N, M, ntype, tot);
    hipLaunchKernelGGL(k_wcp_final, dim3(grid_for(ntype * ntype, 64)), dim3(64), 0, st, tot, N, ntype, dw);
    return sc.finish(space);
}

// Extension for the multi-GPU path (no counterpart in the reference): the raw integer reductions of get_wcp
// (src/warren_cowley_parameter.cpp:26-55) over the rows selected by `rows` (NULL = all): counts[0..T*T) = Z_mn,
// [T*T..T*T+T) = Z_m, [T*T+T..T*T+2T) = atoms per type.  Ranks all-reduce these and apply :57-75 once.
extern "C" int mdh_wcp_counts(const int *verlet, const int *nn, const int *type, const unsigned char *rows, int64_t N,
                              int64_t M, int ntype, unsigned long long *counts, int space, void *stream)
{
    if (N < 0 || M <= 0 || ntype <= 0 || ntype > WCP_MAXT) { set_error("mdh_wcp_counts: need N>=0, M>0 and 1 <= ntype <= 64"); return MDH_ERR_ARG; }
    Scope sc(stream);
    if (sc.failed())
        return sc.error();
    hipStream_t st = sc.stream();
    const int words = ntype * ntype + 2 * ntype;
    const int *dv = sc.stage_in(verlet, (size_t)(N * M), space);
    const int *dn = sc.stage_in(nn, (size_t)N, space);
    const int *dt = sc.stage_in(type, (size_t)N, space);
    const unsigned char *dr = rows ? sc.stage_in(rows, (size_t)N, space) : nullptr;
    unsigned long long *dc = sc.stage(counts, (size_t)words, space, false, true);
    if (sc.failed())
        return sc.error();
    MDH_HIP(hipMemsetAsync(dc, 0, sizeof(unsigned long long) * (size_t)words, st));
    if (N > 0)
        hipLaunchKernelGGL(k_wcp_count, dim3(grid_for(N, 256)), dim3(256), sizeof(unsigned) * (size_t)words, st, dv, dn, dt, dr, N, M, ntype, dc);
    return sc.finish(space);
}

MDH_WARM_UNIT(wcp)
