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
    __shared__ float lds[4096];
    int la = (threadIdx.x * 64) & 16383;
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 qa = {a,b,c,d}, qb = qa, qc = qa, qd = qa;
    lds[threadIdx.x] = a;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :  : "memory");) }
        if (OP == 1) { REP16(asm volatile("v_fmac_f32 %0, %1, %1\n v_fmac_f32 %1, %2, %2\n v_fmac_f32 %2, %3, %3\n v_fmac_f32 %3, %0, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :  : "memory");) }
        if (OP == 2) { REP16(asm volatile("v_subrev_f32 %0, %4, %0\n v_subrev_f32 %1, %4, %1\n v_subrev_f32 %2, %4, %2\n v_subrev_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s) : "memory");) }
        if (OP == 3) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) :  : "memory");) }
        if (OP == 4) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) :  : "memory");) }
        if (OP == 5) { REP16(asm volatile("v_writelane_b32 %0, s20, 3\n v_writelane_b32 %1, s20, 4\n v_writelane_b32 %2, s20, 5\n v_writelane_b32 %3, s20, 6" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) :  : "s20", "memory");) }
        if (OP == 6) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %0, s[20:21]" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) :  : "s20","s21", "memory");) }
        if (OP == 7) { REP16(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cmp_lt_f32_e64 s[22:23], %1, %2\n v_cmp_lt_f32_e64 s[24:25], %2, %3\n v_cmp_lt_f32_e64 s[26:27], %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :  : "s20","s21","s22","s23","s24","s25","s26","s27", "memory");) }
        if (OP == 8) { REP16(asm volatile("v_pk_add_f32 %0, %0, s[20:21] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, s[20:21] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %2, s[20:21] op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, s[20:21] op_sel_hi:[1,0]" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) :  : "s20","s21", "memory");) }
        if (OP == 9) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) :  : "memory");) }
        if (OP == 10) { REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, s20, 0\n v_mbcnt_hi_u32_b32 %0, s21, %0\n v_mbcnt_lo_u32_b32 %2, s22, 0\n v_mbcnt_hi_u32_b32 %2, s23, %2" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) :  : "s20","s21","s22","s23", "memory");) }
        if (OP == 11) { REP16(asm volatile("ds_write_b16 %0, %1\n ds_write_b16 %0, %2 offset:2\n ds_write_b16 %0, %3 offset:4\n ds_write_b16 %0, %1 offset:6" : "+v"(la), "+v"(ib), "+v"(ic), "+v"(id) :  : "memory");) }
        if (OP == 12) { REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)" : "+v"(qa), "+v"(qb), "+v"(qc), "+v"(qd) : "v"(la) : "memory");) }
        if (OP == 13) { REP16(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:16\n ds_read_b32 %2, %4 offset:32\n ds_read_b32 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(la) : "memory");) }
        if (OP == 14) { REP16(asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, s[22:23]\n s_mov_b64 exec, s[20:21]\n s_nop 0" :  :  : "s20","s21", "memory");) }
        if (OP == 15) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0" :  : "v"(ia), "v"(ib), "v"(ic), "v"(id) : "vcc", "memory");) }
        if (OP == 16) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %1, %1, %2, %1\n v_fma_f32 %2, %2, %3, %2\n v_fma_f32 %3, %3, %0, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :  : "memory");) }
        if (OP == 17) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :  : "memory");) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (float)(da + db + dc + dd) + pa.x + pb.x + pc.y + pd.y + ia + ib + ic + id + (float)m + qa.x + qb.y + qc.z + qd.w + lds[(threadIdx.x + 1) & 4095];
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
    const char *names[] = {"v_mul_f32 e32", "v_fmac_f32 e32", "v_subrev_f32 sgpr", "v_add_u32 e32", "v_mov_b32", "v_writelane_b32", "v_cndmask e64 sgpr", "v_cmp_lt_f32 e64->s", "v_pk_add_f32 sgpr opsel", "v_pk_mul_f32", "v_mbcnt lo sgpr", "ds_write_b16 x4", "ds_read_b128 x4", "ds_read_b32 x4", "exec swap x4", "v_cmp_lt_u32 e32", "v_fma_f32 vop3 vgpr", "v_add_f32 e32"};
    for (int cfg = 1; cfg < 4; ++cfg) {
        const int threads = 256, blocks = cfg == 1 ? 512 : (cfg == 2 ? 1536 : 2048); // 1 wave/SIMD (x1 per CU... ), 1 wave/SIMD, 4 waves/SIMD
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
