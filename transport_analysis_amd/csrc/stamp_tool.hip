// stamp_tool.hip — DIAGNOSTIC ONLY (not part of libta_hip.so): runs the flagship
// accumulate kernel with s_memtime stamps per phase and prints the shares.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast stamp_tool.hip -o /tmp/stamp_tool
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "fft_kernels.hpp"
#include "plans.hpp"
using namespace ta;
#ifndef STAMP_PLAN
#define STAMP_PLAN Plan<256, 5, 16, 16, 8>
#endif
using P = STAMP_PLAN;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 10000;
    const long A = argc > 2 ? atol(argv[2]) : 20000;
    const int nwg = argc > 3 ? atoi(argv[3]) : 256;
    const long C = A * 3;
    double* vel; CK(hipMalloc(&vel, sizeof(double) * T * C));
    std::vector<double> h((size_t)T * C);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 500.0 - 1.0;
    CK(hipMemcpy(vel, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    std::vector<cd> tw(4 * P::M + 2, cd{0.0, 0.0});
    for (int n = 0; n < 2 * P::M; ++n) tw[n] = cd{cos(M_PI * n / P::M), -sin(M_PI * n / P::M)};
    { const int R0 = StageInfo<P, 0>::R; const long L0 = P::M / R0;
      for (int B = 0; B < 2; ++B) for (long q = 0; q < R0; ++q) for (long u = 0; u < L0; ++u) {
          long n = (u * (2 * q + B)) % (2L * P::M); tw[(2 + B) * (size_t)P::M + q * L0 + u] = cd{cos(M_PI * n / P::M), -sin(M_PI * n / P::M)}; } }
    cd* d_tw; CK(hipMalloc(&d_tw, sizeof(cd) * tw.size()));
    CK(hipMemcpy(d_tw, tw.data(), sizeof(cd) * tw.size(), hipMemcpyHostToDevice));
    const size_t accn = (size_t)nwg * 2 * acc_quads<P>() * 2 * P::NT;
    double* partial; CK(hipMalloc(&partial, sizeof(double) * accn));
    CK(hipMemset(partial, 0, sizeof(double) * accn));
    unsigned long long* st; CK(hipMalloc(&st, 8 * sizeof(unsigned long long) * nwg));
    const size_t lds = (size_t)P::lds_elems() * sizeof(cd);
    auto kern = k_fft_accum<WithLanding<P>, true, true>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(P::NT), lds, 0, vel, C, 2L, T, C, d_tw, partial, 0, st, 0, 0L, (double*)nullptr, 0L);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rep %d: %.3f ms\n", rep, ms);
    }
    std::vector<unsigned long long> hs(8 * nwg);
    CK(hipMemcpy(hs.data(), st, sizeof(unsigned long long) * hs.size(), hipMemcpyDeviceToHost));
    double sum[8] = {0};
    for (int w = 0; w < nwg; ++w) for (int i = 0; i < 8; ++i) sum[i] += (double)hs[8 * w + i];
    const char* names[8] = {"wait+firstA", "bar+accin+midA", "lastA+accout", "bar+firstB", "bar+accin+midB(+gather)", "lastB(+gather)+accout", "-", "iters"};
    double tot = 0; for (int i = 0; i < 6; ++i) tot += sum[i];
    for (int i = 0; i < 8; ++i)
        printf("%-16s %12.0f cycles/iter  (%.1f%%)\n", names[i], sum[i] / sum[7], i < 6 ? 100.0 * sum[i] / tot : 0.0);
    return 0;
}
