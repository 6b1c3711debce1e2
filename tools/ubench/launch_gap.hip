// launch_gap.hip — what a DEPENDENT kernel launch costs on this box: a chain of K trivial kernels on one stream, as plain
// launches and as one hipGraph launch (stream-captured), per chain.  hipcc --offload-arch=gfx950 -O2 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_tick(int *p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 16, R = argc > 2 ? atoi(argv[2]) : 2000, G = argc > 3 ? atoi(argv[3]) : 64;
    int *d; CK(hipMalloc(&d, 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto chain = [&]() { for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_tick, dim3(G), dim3(256), 0, st, d, k); };
    for (int r = 0; r < 50; ++r) chain();
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; ++r) chain();
    CK(hipStreamSynchronize(st));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
    printf("stream : %d kernels per chain, %.2f us per chain = %.2f us per kernel\n", K, us, us / K);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    chain();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
    printf("graph  : %d kernels per chain, %.2f us per chain = %.2f us per kernel\n", K, us, us / K);
    // a chain with a memset node and a 4-byte D2H copy in it (what a build has besides kernels)
    int *h; CK(hipHostMalloc(reinterpret_cast<void **>(&h), 64, hipHostMallocDefault));
    auto chain2 = [&]() {
        hipMemsetAsync(d, 0, 1024, st);
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_tick, dim3(G), dim3(256), 0, st, d, k);
        hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, st);
    };
    for (int r = 0; r < 50; ++r) chain2();
    CK(hipStreamSynchronize(st));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; ++r) chain2();
    CK(hipStreamSynchronize(st));
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
    printf("stream + memset + 4-byte D2H: %.2f us per chain\n", us);
    return 0;
}
