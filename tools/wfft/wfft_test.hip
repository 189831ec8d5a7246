// wfft_test.hip — standalone check + timing of k_wfft_accum (tools only; not part of the library).
//   wfft_test check            : few pairs, compare the accumulated spectrum with a CPU transform
//   wfft_test time <n_pairs> [T] [reps] [stamp]
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../transport_analysis_amd/csrc/wfft.hpp"

using namespace ta;
#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            exit(2);                                                                \
        }                                                                           \
    } while (0)

typedef std::complex<long double> cl;
static void fft_rec(std::vector<cl>& a) {
    const size_t n = a.size();
    if (n == 1) return;
    int r = (n % 2 == 0) ? 2 : (n % 5 == 0) ? 5 : (int)n;
    const size_t m = n / r;
    std::vector<std::vector<cl>> sub(r, std::vector<cl>(m));
    for (size_t i = 0; i < n; ++i) sub[i % r][i / r] = a[i];
    for (auto& s : sub) fft_rec(s);
    const long double pi = 3.141592653589793238462643383279502884L;
    for (size_t k = 0; k < n; ++k) {
        cl acc = 0;
        for (int j = 0; j < r; ++j) {
            const long double ang = -2 * pi * (long double)((j * k) % n) / (long double)n;
            acc += sub[j][k % m] * cl(cosl(ang), sinl(ang));
        }
        a[k] = acc;
    }
}

__global__ void k_fill(double* p, size_t n, unsigned long long seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 3.4641016151377544;  // unit variance
    }
}

template <int R0>
static int run(int argc, char** argv) {
    using P = WPlan<R0>;
    const int M = P::M;
    const bool check = argc < 2 || !strcmp(argv[1], "check");
    const long n_pairs = check ? 11 : atol(argv[2]);
    const int T = argc > 3 ? atoi(argv[3]) : (M == 10240 ? 10000 : M - 37);
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const bool stamp = argc > 5 && atoi(argv[5]);
    const bool touch = !(argc > 6 && atoi(argv[6]) == 0);
    const bool inter = !(argc > 7 && atoi(argv[7]) == 0);
    const long pitch = T;
    int ncu = 256;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    ncu = prop.multiProcessorCount;
    const bool half = getenv("WF_HALF") && R0 == 20;
    const int split = getenv("WF_SPLIT") ? atoi(getenv("WF_SPLIT")) : 0;  // 1: plain, 2: interleaved
    int nwg = (int)std::min<long>(ncu, n_pairs);
    if (half) nwg = std::max(16, (int)std::min<long>(2 * ncu, 2 * n_pairs) / 16 * 16);
    if (split) nwg = std::max(16, (int)std::min<long>(ncu, 2 * n_pairs) / 16 * 16);
    // twiddles
    std::vector<cd> tw(wf_table_elems(R0));
    wf_fill_table(R0, tw.data());
    cd* d_tw;
    double *d_pm, *d_acc;
    unsigned long long* d_st;
    const size_t n_el = (size_t)n_pairs * pitch * 2;
    CK(hipMalloc(&d_tw, tw.size() * sizeof(cd)));
    CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cd), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pm, n_el * 8));
    CK(hipMalloc(&d_acc, (size_t)nwg * 2 * M * 8));
    CK(hipMalloc(&d_st, (size_t)nwg * 8 * 8));
    CK(hipMemset(d_acc, 0, (size_t)nwg * 2 * M * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_whalf_accum<false>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)WHalf::kLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_whalf_accum<true>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)WHalf::kLds));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d_pm, n_el, 12345ull);
    CK(hipDeviceSynchronize());
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wfft_accum<P, false>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wfft_accum<P, false, false>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wfft_accum<P, true, false, false>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wfft_accum<P, true, false, true>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wfft_accum<P, false, false, false>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wfft_accum<P, true>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wsplit_accum<P, false, false>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wsplit_accum<P, false, true>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wsplit_accum<P, true, true>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
    auto launch = [&]() {
        if (split && stamp)
            hipLaunchKernelGGL((k_wsplit_accum<P, true, true>), dim3(nwg), dim3(P::NT), P::kLds, 0, d_pm, pitch, T,
                               n_pairs, d_tw, d_acc, d_st);
        else if (split == 2)
            hipLaunchKernelGGL((k_wsplit_accum<P, false, true>), dim3(nwg), dim3(P::NT), P::kLds, 0, d_pm, pitch, T,
                               n_pairs, d_tw, d_acc, d_st);
        else if (split)
            hipLaunchKernelGGL((k_wsplit_accum<P, false, false>), dim3(nwg), dim3(P::NT), P::kLds, 0, d_pm, pitch, T,
                               n_pairs, d_tw, d_acc, d_st);
        else if (half && stamp)
            hipLaunchKernelGGL((k_whalf_accum<true>), dim3(nwg), dim3(256), WHalf::kLds, 0, d_pm, pitch, T, n_pairs,
                               d_tw, d_acc, d_st);
        else if (half)
            hipLaunchKernelGGL((k_whalf_accum<false>), dim3(nwg), dim3(256), WHalf::kLds, 0, d_pm, pitch, T, n_pairs,
                               d_tw, d_acc, d_st);
        else if (stamp && !inter)
            hipLaunchKernelGGL((k_wfft_accum<P, true, false, false>), dim3(nwg), dim3(P::NT), P::kLds, 0, d_pm, pitch, T,
                               n_pairs, d_tw, d_acc, d_st);
        else if (stamp && !touch)
            hipLaunchKernelGGL((k_wfft_accum<P, true, false, true>), dim3(nwg), dim3(P::NT), P::kLds, 0, d_pm, pitch, T,
                               n_pairs, d_tw, d_acc, d_st);
        else if (!inter)
            hipLaunchKernelGGL((k_wfft_accum<P, false, false, false>), dim3(nwg), dim3(P::NT), P::kLds, 0, d_pm, pitch, T,
                               n_pairs, d_tw, d_acc, d_st);
        else if (stamp)
            hipLaunchKernelGGL((k_wfft_accum<P, true>), dim3(nwg), dim3(P::NT), P::kLds, 0, d_pm, pitch, T,
                               n_pairs, d_tw, d_acc, d_st);
        else if (!touch)
            hipLaunchKernelGGL((k_wfft_accum<P, false, false>), dim3(nwg), dim3(P::NT), P::kLds, 0, d_pm, pitch, T,
                               n_pairs, d_tw, d_acc, d_st);
        else
            hipLaunchKernelGGL((k_wfft_accum<P, false>), dim3(nwg), dim3(P::NT), P::kLds, 0, d_pm, pitch, T,
                               n_pairs, d_tw, d_acc, d_st);
        CK(hipGetLastError());
    };
    launch();
    CK(hipDeviceSynchronize());
    if (check) {
        std::vector<double> h((size_t)n_pairs * pitch * 2), acc((size_t)nwg * 2 * M);
        CK(hipMemcpy(h.data(), d_pm, h.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(acc.data(), d_acc, acc.size() * 8, hipMemcpyDeviceToHost));
        std::vector<long double> ref(2 * (size_t)M, 0.0L);
        for (long p = 0; p < n_pairs; ++p) {
            std::vector<cl> a(2 * (size_t)M, cl(0, 0));
            for (int t = 0; t < T; ++t) a[t] = cl(h[(p * pitch + t) * 2], h[(p * pitch + t) * 2 + 1]);
            fft_rec(a);
            for (size_t k = 0; k < a.size(); ++k) ref[k] += std::norm(a[k]);
        }
        long double mx = 0, err = 0;
        for (size_t k = 0; k < ref.size(); ++k) {
            long double got = 0;
            for (int w = 0; w < nwg; ++w) got += acc[(size_t)w * 2 * M + k];
            mx = std::max(mx, fabsl(ref[k]));
            err = std::max(err, fabsl(got - ref[k]));
        }
        printf("R0=%d M=%d T=%d pairs=%ld  max|ref|=%Lg  max err=%Lg  rel=%Lg  %s\n", R0, M, T, n_pairs, mx,
               err, err / mx, err / mx < 1e-12L ? "OK" : "FAIL");
        return err / mx < 1e-12L ? 0 : 1;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
        sum += ms;
    }
    const double bytes = (double)n_pairs * T * 16.0;
    printf("R0=%d M=%d T=%d pairs=%ld nwg=%d: best %.3f ms  mean %.3f ms  %.1f GB/s  (x150000 pairs: %.2f ms)\n", R0,
           M, T, n_pairs, nwg, best, sum / reps, bytes / best * 1e-6, best * 150000.0 / n_pairs);
    if (stamp) {
        std::vector<unsigned long long> st((size_t)nwg * 8);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        for (int h = 0; h < 2; ++h) {
            double s[4] = {0, 0, 0, 0};
            for (int w = 0; w < nwg; ++w)
                for (int i = 0; i < 4; ++i) s[i] += (double)st[(size_t)w * 8 + h * 4 + i];
            const double per = (double)nwg * ((double)n_pairs / nwg);
            printf("cycles/pair (wave %d): S1A %.0f  S2A %.0f  S1B %.0f  S2B %.0f  total %.0f\n", 4 * h, s[0] / per,
                   s[1] / per, s[2] / per, s[3] / per, (s[0] + s[1] + s[2] + s[3]) / per);
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const char* r = getenv("WF_R0");
    const int R0 = r ? atoi(r) : 20;
    switch (R0) {
        case 20: return run<20>(argc, argv);
        case 16: return run<16>(argc, argv);
        case 10: return run<10>(argc, argv);
        case 8: return run<8>(argc, argv);
        case 5: return run<5>(argc, argv);
        case 4: return run<4>(argc, argv);
        case 2: return run<2>(argc, argv);
    }
    fprintf(stderr, "unsupported WF_R0\n");
    return 2;
}
