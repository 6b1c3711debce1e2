// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction of the ops the neighbor kernels are made of.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s, double ds)
{
    float a = threadIdx.x * 1e-3f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
    double da = a, db = b, dc = c, dd = d;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 pa = {a, b}, pb = {c, d}, pc = {b, c}, pd = {d, a};
    int ia = threadIdx.x, ib = ia + 1, ic = ia + 2, id = ia + 3;
    unsigned long long m = 0;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s));) }
        if (OP == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));) }
        if (OP == 2) { REP16(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd));) }
        if (OP == 3) { REP16(asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd));) }
        if (OP == 4) { REP16(asm volatile("v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd));) }
        if (OP == 5) { REP16(asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_sub_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 6) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0" : : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");) }
        if (OP == 7) { REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0\n v_mbcnt_hi_u32_b32 %1, -1, %1\n v_mbcnt_lo_u32_b32 %2, -1, %2\n v_mbcnt_hi_u32_b32 %3, -1, %3" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id));) }
        if (OP == 8) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id));) }
        if (OP == 9) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id));) }
        if (OP == 10) { REP16(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9" : : "v"(ia), "v"(ib), "v"(ic), "v"(id) : "s20", "s21", "s22", "s23");) }
        if (OP == 11) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %2\n v_lshl_add_u32 %2, %2, 1, %3\n v_lshl_add_u32 %3, %3, 1, %0" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id));) }
        if (OP == 12) { REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %3, %0" : : "v"(da), "v"(db), "v"(dc), "v"(dd) : "vcc");) }
        if (OP == 13) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));) }
        if (OP == 14) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");) }
        if (OP == 15) { REP16(asm volatile("s_and_b64 s[20:21], s[20:21], vcc\n s_bcnt1_i32_b64 s22, s[20:21]\n s_add_i32 s23, s23, s22\n s_lshl_b32 s22, s23, 1" : : : "s20", "s21", "s22", "s23", "scc");) }
        if (OP == 16) { REP16(asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd));) }
        if (OP == 17) { REP16(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f64_f32 %4, %2\n v_cvt_f64_f32 %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(da), "+v"(db));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (float)(da + db + dc + dd) + pa.x + pb.x + pc.y + pd.y + ia + ib + ic + id + (float)m;
}
template <int OP> double run(float *out, int blocks, int threads)
{
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(out, 10, 1.0f, 1.0);
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(out, iters, 1.0f, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = waves per SIMD * iters * 64
    const double waves_per_simd = (double)blocks * threads / 64.0 / 1024.0;
    return ms * 1e-3 * 2.4e9 / (waves_per_simd * iters * 64.0); // cycles (at 2.4 GHz) per wave-instruction per SIMD
}
int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_sub_f32", "v_cmp_lt_f32", "v_mbcnt", "v_mul_lo_u32", "v_mad_u32_u24", "v_readlane_b32", "v_lshl_add_u32", "v_cmp_lt_f64", "v_pk_add_f32", "v_cndmask_b32", "salu(4)", "v_sqrt/rsq_f64", "v_cvt f32<->f64"};
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int threads = cfg == 0 ? 64 : 256, blocks = cfg == 0 ? 1024 : (cfg == 1 ? 256 : 1024); // 1 wave/SIMD (x1 per CU... ), 1 wave/SIMD, 4 waves/SIMD
        printf("config: %d blocks x %d threads (%.2f waves/SIMD)\n", blocks, threads, (double)blocks * threads / 64 / 1024);
        double r[18];
        r[0] = run<0>(out, blocks, threads); r[1] = run<1>(out, blocks, threads); r[2] = run<2>(out, blocks, threads); r[3] = run<3>(out, blocks, threads);
        r[4] = run<4>(out, blocks, threads); r[5] = run<5>(out, blocks, threads); r[6] = run<6>(out, blocks, threads); r[7] = run<7>(out, blocks, threads);
        r[8] = run<8>(out, blocks, threads); r[9] = run<9>(out, blocks, threads); r[10] = run<10>(out, blocks, threads); r[11] = run<11>(out, blocks, threads);
        r[12] = run<12>(out, blocks, threads); r[13] = run<13>(out, blocks, threads); r[14] = run<14>(out, blocks, threads); r[15] = run<15>(out, blocks, threads);
        r[16] = run<16>(out, blocks, threads); r[17] = run<17>(out, blocks, threads);
        for (int i = 0; i < 18; ++i) printf("  %-18s %6.2f cycles/wave-instr/SIMD (2.4 GHz assumed)\n", names[i], r[i]);
    }
    return 0;
}
