// wfft_test.hip — standalone check + timing of the kernels of csrc/wfft.hpp (tools only; not part
// of the library).  Plan from the environment: WF_R0 (first-stage radix, default 20), WF_R (outer
// radix, default 1).
//   wfft_test check            : 11 pairs; the summed spectrum of the forward kernel against a
//                                long-double transform, and the inverse kernel's lag sums against
//                                the direct sums of products
//   wfft_test inv <n_items> [T] [reps]
//                              : inverse kernel on n_items random spectra (by-particle shape: one
//                                workgroup per spectrum, grid = compute units); built with
//                                -DWF_INV_STAMP=1 also the cycles per phase
//   wfft_test time <n_pairs> [T] [reps] [stamp]
//                              : forward kernel, lag-sum mode; stamp = 1 adds the in-kernel s_memtime
//                                split (S1 incl. waiting for rows / S2 incl. the barrier)
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifdef WF_ABLATIONS  // build.sh: csrc/wfft.hpp with ablations.patch applied (the WF_ABL / WF_INV_STAMP instrumentation that left the product header in round 6)
#include "wfft_abl.hpp"
#else
#include "../../transport_analysis_amd/csrc/wfft.hpp"
#endif
#include "wfused.hpp"

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

__global__ void k_fill32(float* p, size_t n, unsigned long long seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = (float)(((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 3.4641016151377544);
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

// ---- the fused by-particle kernel (csrc/wfused.hpp) --------------------------------------------------
//   wfft_test fcheck [T] [f32]        : 7 atoms, dim = 1, 2, 3: every atom's lags against direct sums of products
//   wfft_test ftime <n_atoms> [T] [reps] [D] [f32] [percu]
template <int R0>
static int run_fused(int argc, char** argv) {
    if constexpr (R0 % 2 != 0 || R0 < 4) {
        fprintf(stderr, "fused: even WF_R0 >= 4 only\n");
        return 2;
    } else {
        using P = WPlan<R0>;
        const int M2 = R0 * 512;  // padded half length: the series may have up to M2 frames
        const bool check = !strcmp(argv[1], "fcheck");
        const int T = check ? (argc > 2 && atoi(argv[2]) > 0 ? atoi(argv[2]) : (M2 == 10240 ? 10000 : M2 - 37)) : (argc > 3 ? atoi(argv[3]) : (M2 == 10240 ? 10000 : M2 - 37));
        const bool f32 = check ? (argc > 3 && atoi(argv[3])) : (argc > 6 && atoi(argv[6]));
        const long pitch = (T + 7) / 8 * 8, ld = pitch;
        hipDeviceProp_t prop;
        CK(hipGetDeviceProperties(&prop, 0));
        std::vector<cd> tw(wf_table_elems(R0, 1));
        wf_fill_table(R0, 1, tw.data());
        cd* d_tw;
        CK(hipMalloc(&d_tw, tw.size() * sizeof(cd)));
        CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cd), hipMemcpyHostToDevice));
        std::vector<double> rn(T);
        wfused_fill_rnorm(R0, T, rn.data());
        double* d_rn;
        CK(hipMalloc(&d_rn, rn.size() * 8));
        CK(hipMemcpy(d_rn, rn.data(), rn.size() * 8, hipMemcpyHostToDevice));
        auto launch = [&](const void* pm, long n_atoms, int D, double* d_out, int nwg) {
            auto go = [&](auto kern) {
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)P::kLds));
                hipLaunchKernelGGL(kern, dim3(nwg), dim3(P::NT), P::kLds, 0, (const double*)pm, pitch, T, n_atoms, D, d_tw,
                                   d_rn, d_out, ld);
                CK(hipGetLastError());
            };
            if (f32) go(k_wfused_bp<P, true>);
            else go(k_wfused_bp<P, false>);
        };
        auto per_cu = [&]() {
            int n = 1;
            if (f32) {
                auto kern = k_wfused_bp<P, true>;
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, P::NT, P::kLds) != hipSuccess || n < 1) n = 1;
            } else {
                auto kern = k_wfused_bp<P, false>;
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, P::NT, P::kLds) != hipSuccess || n < 1) n = 1;
            }
            return n;
        };
        if (check) {
            int bad = 0;
            for (int D = 1; D <= 3; ++D) {
                const long A = 7, n_cols = A * D, n_pairs = (n_cols + 1) / 2;
                const size_t n_el = (size_t)n_pairs * pitch * 2;
                void* d_pm;
                double* d_out;
                CK(hipMalloc(&d_pm, n_el * (f32 ? 4 : 8)));
                CK(hipMalloc(&d_out, (size_t)A * ld * 8));
                CK(hipMemset(d_out, 0xff, (size_t)A * ld * 8));
                if (f32) hipLaunchKernelGGL(k_fill32, dim3(256), dim3(256), 0, 0, (float*)d_pm, n_el, 4242ull + D);
                else hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, (double*)d_pm, n_el, 4242ull + D);
                CK(hipDeviceSynchronize());
                launch(d_pm, A, D, d_out, 3);  // three workgroups: grid-stride over the groups of atoms
                CK(hipDeviceSynchronize());
                std::vector<double> h(n_el), o((size_t)A * ld);
                if (f32) {
                    std::vector<float> hf(n_el);
                    CK(hipMemcpy(hf.data(), d_pm, n_el * 4, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < n_el; ++i) h[i] = hf[i];
                } else CK(hipMemcpy(h.data(), d_pm, n_el * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost));
                long double mx = 0, err = 0;
                for (long a = 0; a < A; ++a)
                    for (int n = 0; n < T; n += (n < 40 || n > T - 40) ? 1 : std::max(1, T / 61)) {
                        long double sref = 0;
                        for (int d = 0; d < D; ++d) {
                            const long c = a * D + d, p = c / 2, hf = c & 1;
                            for (int t = 0; t + n < T; ++t)
                                sref += (long double)h[(p * pitch + t) * 2 + hf] * h[(p * pitch + t + n) * 2 + hf];
                        }
                        sref /= (long double)(T - n);
                        mx = std::max(mx, fabsl(sref));
                        const long double e = fabsl((long double)o[a * ld + n] - sref);
                        if (!(e <= err)) err = e;  // (NaN propagates)
                    }
                const bool ok = err / mx < 1e-11L;
                printf("fused R0=%d T=%d D=%d atoms=%ld %s  max|ref|=%Lg  max err=%Lg  rel=%Lg  %s\n", R0, T, D, A,
                       f32 ? "f32" : "f64", mx, err, err / mx, ok ? "OK" : "FAIL");
                bad += !ok;
                CK(hipFree(d_pm));
                CK(hipFree(d_out));
            }
            return bad ? 1 : 0;
        }
        const long A = atol(argv[2]);
        const int reps = argc > 4 ? atoi(argv[4]) : 5;
        const int D = argc > 5 ? atoi(argv[5]) : 3;
        int pc = per_cu();
        if (argc > 7 && atoi(argv[7]) > 0) pc = std::min(pc, atoi(argv[7]));
        const long n_groups = D & 1 ? (A + 1) / 2 : A;
        const int nwg = (int)std::min<long>((long)prop.multiProcessorCount * pc, n_groups);
        const long n_pairs = (A * D + 1) / 2;
        const size_t n_el = (size_t)n_pairs * pitch * 2;
        void* d_pm;
        double* d_out;
        CK(hipMalloc(&d_pm, n_el * (f32 ? 4 : 8)));
        CK(hipMalloc(&d_out, (size_t)A * ld * 8));
        if (f32) hipLaunchKernelGGL(k_fill32, dim3(4096), dim3(256), 0, 0, (float*)d_pm, n_el, 12345ull);
        else hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)d_pm, n_el, 12345ull);
        CK(hipDeviceSynchronize());
        launch(d_pm, A, D, d_out, nwg);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        float best = 1e30f, sum = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, 0));
            launch(d_pm, A, D, d_out, nwg);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
            sum += ms;
        }
#if WFU_STAMP
        {
            unsigned long long z[16];
            CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(wfu_stamps), sizeof(z)));
            const double per = (double)A * (reps + 1);
            const char* nm[10] = {"S1", "S1 barrier", "S2 (+row requests)", "S2 barrier", "inv: acc -> LDS + barrier", "inv: reads, barrier, sub-series", "inv: barrier", "inv: radix + Q to LDS", "inv: barrier + untangle + stores", "inv: end barrier"};
            double tot = 0;
            for (int i = 0; i < 10; ++i) tot += (double)z[i];
            for (int i = 0; i < 10; ++i) printf("  cycles per atom (wave 0): %-36s %9.0f  %5.1f %%\n", nm[i], z[i] / per, 100.0 * z[i] / tot);
            printf("  total %.0f cycles per atom\n", tot / per);
        }
#endif
        printf("fused R0=%d T=%d atoms=%ld D=%d %s nwg=%d (%d per CU): best %.3f ms  mean %.3f ms = %.2f us per atom and CU  (x100000 atoms: %.2f ms)\n",
               R0, T, A, D, f32 ? "f32" : "f64", nwg, pc, best, sum / reps, best * 1e3 * prop.multiProcessorCount / A,
               best * 100000.0 / A);
        return 0;
    }
}

// ---- the forward kernel in by-particle mode with the in-kernel stamps (round 6: why a unit costs more there than in
// lag-sum mode):   wfft_test bptime <n_atoms> [T] [reps] [D]
template <int R0>
static int run_bptime(int argc, char** argv) {
    using P = WPlan<R0>;
    const int M = P::M, L = 2 * M;
    const long A = argc > 2 ? atol(argv[2]) : 25600;
    const int T = argc > 3 ? atoi(argv[3]) : (M == 10240 ? 10000 : M - 37);
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const int D = argc > 5 ? atoi(argv[5]) : 3;
    const long pitch = (T + 7) / 8 * 8, n_pairs = (A * D + 1) / 2;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const long groups = D & 1 ? (A + 1) / 2 : A;
    const int nwg = (int)std::max<long>(16, std::min<long>(prop.multiProcessorCount, 2 * groups) / 16 * 16);
    std::vector<cd> tw(wf_table_elems(R0, 1));
    wf_fill_table(R0, 1, tw.data());
    cd* d_tw;
    double *d_pm, *d_acc;
    unsigned long long* d_st;
    CK(hipMalloc(&d_tw, tw.size() * sizeof(cd)));
    CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cd), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pm, (size_t)n_pairs * pitch * 16));
    CK(hipMalloc(&d_acc, (size_t)A * L * 8));
    CK(hipMalloc(&d_st, (size_t)nwg * 16 * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d_pm, (size_t)n_pairs * pitch * 2, 12345ull);
    CK(hipDeviceSynchronize());
    auto kern = k_wsplit_accum<P, true, false, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r <= reps; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(P::NT), P::kLds, 0, d_pm, pitch, T, A, d_tw, d_acc, D, 1, d_st);
        CK(hipGetLastError());
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) best = std::min(best, ms);
    }
    std::vector<unsigned long long> st((size_t)nwg * 16);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
#ifdef WF_MIX  // (built with tools/wfft/mixed_units.patch applied: 3 units per pair of atoms at dim 3, 1 per pair at dim 1)
    const long units_per_atom_x2 = WF_MIX ? (D == 3 ? 3 : D == 1 ? 1 : 2) : (D == 3 ? 4 : 2);
#else
    const long units_per_atom_x2 = D == 3 ? 4 : 2;
#endif
    const double unit_passes = (double)A * units_per_atom_x2 / 2.0 * 2.0;  // all workgroups together
    for (int h = 0; h < 2; ++h) {
        double sacc[4] = {0, 0, 0, 0}, tail = 0;
        for (int w = 0; w < nwg; ++w) {
            tail += (double)st[(size_t)w * 16 + 8 + h];
            for (int i = 0; i < 4; ++i) sacc[i] += (double)st[(size_t)w * 16 + h * 4 + i];
        }
        printf("by-particle forward R0=%d T=%d atoms=%ld D=%d nwg=%d: cycles per unit and pass (wave %d): S1 %.0f  S2 %.0f (row requests + barrier wait %.0f)  total %.0f  clock %.0f MHz\n",
               R0, T, A, D, nwg, 4 * h, sacc[0] / unit_passes, sacc[1] / unit_passes, tail / unit_passes, (sacc[0] + sacc[1]) / unit_passes,
               sacc[2] / sacc[3] * 100.0);
    }
    printf("by-particle forward: best %.3f ms (x100000 atoms: %.2f ms)\n", best, best * 100000.0 / A);
    return 0;
}

template <int R0>
static int run(int R, int argc, char** argv) {
    if (argc > 1 && (!strcmp(argv[1], "fcheck") || !strcmp(argv[1], "ftime"))) return run_fused<R0>(argc, argv);
    if (argc > 1 && !strcmp(argv[1], "bptime")) return run_bptime<R0>(argc, argv);
    using P = WPlan<R0>;
    const int M = P::M, L = 2 * R * M;
    if (argc > 1 && !strcmp(argv[1], "inv")) {
        const long n_items = argc > 2 ? atol(argv[2]) : 2048;
        const int Ti = argc > 3 ? atoi(argv[3]) : (R * M == 10240 ? 10000 : R * M - 37);
        const int reps_i = argc > 4 ? atoi(argv[4]) : 5;
        const long ld = (Ti + 7) / 8 * 8;
        hipDeviceProp_t pr;
        CK(hipGetDeviceProperties(&pr, 0));
        std::vector<cd> twv(wf_table_elems(R0, R));
        wf_fill_table(R0, R, twv.data());
        cd* dtw;
        double *dspec, *dout;
        CK(hipMalloc(&dtw, twv.size() * sizeof(cd)));
        CK(hipMemcpy(dtw, twv.data(), twv.size() * sizeof(cd), hipMemcpyHostToDevice));
        CK(hipMalloc(&dspec, (size_t)n_items * L * 8));
        CK(hipMalloc(&dout, (size_t)n_items * ld * 8));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, dspec, (size_t)n_items * L, 777ull);
        CK(hipDeviceSynchronize());
        auto go = [&](auto kern) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)P::kLds));
            hipLaunchKernelGGL(kern, dim3(std::min<long>(pr.multiProcessorCount, n_items)), dim3(P::NT), P::kLds, 0, dspec,
                               Ti, n_items, dtw, dout, ld, R);
            CK(hipGetLastError());
        };
        auto launch_inv = [&]() {
            if (R > 1) go(k_winverse<P, true, 0>);
            else go(k_winverse<P, false, (P::NS1 < 2 ? P::NS1 : 2)>);  // the library's default prefetch depth
        };
        launch_inv();
        CK(hipDeviceSynchronize());
#if WF_INV_STAMP
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(wf_inv_stamps), z, sizeof(z)));
#endif
        hipEvent_t a0, a1;
        CK(hipEventCreate(&a0));
        CK(hipEventCreate(&a1));
        float best = 1e30f;
        for (int r = 0; r < reps_i; ++r) {
            CK(hipEventRecord(a0, 0));
            launch_inv();
            CK(hipEventRecord(a1, 0));
            CK(hipEventSynchronize(a1));
            float ms;
            CK(hipEventElapsedTime(&ms, a0, a1));
            best = std::min(best, ms);
        }
        printf("inverse R0=%d R=%d T=%d items=%ld: best %.3f ms = %.2f us per item and CU  (x100000 items: %.2f ms)\n", R0, R, Ti,
               n_items, best, best * 1e3 * pr.multiProcessorCount / n_items, best * 100000.0 / n_items);
#if WF_INV_STAMP
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(wf_inv_stamps), sizeof(z)));
        const double it = (double)z[4];
        printf("cycles per item (wave 0): sub-series stage incl. spectrum wait %.0f  radix-R0 %.0f  Q to LDS %.0f  untangle + stores %.0f  total %.0f\n",
               z[0] / it, z[1] / it, z[2] / it, z[3] / it, (z[0] + z[1] + z[2] + z[3]) / it);
#endif
        return 0;
    }
    const bool check = argc < 2 || !strcmp(argv[1], "check");
    const long n_pairs = check ? 11 : atol(argv[2]);
    const int T = argc > 3 ? atoi(argv[3]) : (R * M == 10240 ? 10000 : R * M - 37);
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const bool stamp = argc > 5 && atoi(argv[5]);
    const long pitch = (T + 7) / 8 * 8;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, gran = 16 * R;
    int per_cu = 1;  // resident workgroups per compute unit, as the library sizes its grid
    {
        auto kern = k_wsplit_accum<P, false, false, false>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)P::kLds));
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, P::NT, P::kLds) != hipSuccess || per_cu < 1)
            per_cu = 1;
        if (getenv("WF_PERCU")) per_cu = std::min(per_cu, atoi(getenv("WF_PERCU")));  // fewer workgroups per unit than fit
    }
    const int nwg = std::max<long>(gran, std::min<long>((long)ncu * per_cu, 2 * R * n_pairs) / gran * gran);
    const int n_tuples = nwg / (2 * R);
    std::vector<cd> tw(wf_table_elems(R0, R));
    wf_fill_table(R0, R, tw.data());
    cd* d_tw;
    double *d_pm, *d_acc, *d_spec, *d_lag;
    unsigned long long* d_st;
    const size_t n_el = (size_t)n_pairs * pitch * 2;
    CK(hipMalloc(&d_tw, tw.size() * sizeof(cd)));
    CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(cd), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pm, n_el * 8));
    CK(hipMalloc(&d_acc, (size_t)n_tuples * L * 8));
    CK(hipMalloc(&d_spec, (size_t)L * 8));
    CK(hipMalloc(&d_lag, (size_t)T * 8));
    CK(hipMalloc(&d_st, (size_t)nwg * 16 * 8));
    if (getenv("WF_ZERO") && atoi(getenv("WF_ZERO")))  // all-zero input: the same cycles at a fraction of the power
        CK(hipMemset(d_pm, 0, n_el * 8));
    else
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d_pm, n_el, 12345ull);
    CK(hipDeviceSynchronize());
    auto launch = [&]() {
        auto go = [&](auto kern) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)P::kLds));
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(P::NT), P::kLds, 0, d_pm, pitch, T, n_pairs, d_tw, d_acc, 0, R,
                               d_st);
        };
        if (R > 1) stamp ? go(k_wsplit_accum<P, false, true, true>) : go(k_wsplit_accum<P, false, true, false>);
        else stamp ? go(k_wsplit_accum<P, false, false, true>) : go(k_wsplit_accum<P, false, false, false>);
        CK(hipGetLastError());
    };
    launch();
    CK(hipDeviceSynchronize());
    if (check) {
        std::vector<double> h(n_el), acc((size_t)n_tuples * L);
        CK(hipMemcpy(h.data(), d_pm, h.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(acc.data(), d_acc, acc.size() * 8, hipMemcpyDeviceToHost));
        std::vector<long double> ref((size_t)L, 0.0L);
        for (long p = 0; p < n_pairs; ++p) {
            std::vector<cl> a((size_t)L, cl(0, 0));
            for (int t = 0; t < T; ++t) a[t] = cl(h[(p * pitch + t) * 2], h[(p * pitch + t) * 2 + 1]);
            fft_rec(a);
            for (size_t k = 0; k < a.size(); ++k) ref[k] += std::norm(a[k]);
        }
        // spectrum layout [pass c][q][cc / 2][lane][cc & 1]: bin k = 2R (q + R0 s) + c, s = (lane >> 3) + 8 (lane & 7) + 64 cc
        std::vector<double> spec((size_t)L, 0.0);
        long double mx = 0, err = 0;
        for (int c = 0; c < 2 * R; ++c)
            for (int q = 0; q < R0; ++q)
                for (int cc = 0; cc < 8; ++cc)
                    for (int lane = 0; lane < 64; ++lane) {
                        const size_t i = (size_t)c * M + (((q * 4 + cc / 2) * 64 + lane) * 2 + (cc & 1));
                        long double got = 0;
                        for (int w = 0; w < n_tuples; ++w) got += acc[(size_t)w * L + i];
                        spec[i] = (double)got;
                        const int sb = (lane >> 3) + 8 * (lane & 7) + 64 * cc;
                        const size_t k = (size_t)2 * R * (q + R0 * sb) + c;
                        mx = std::max(mx, fabsl(ref[k]));
                        err = std::max(err, fabsl(got - ref[k]));
                    }
        const bool ok_f = err / mx < 1e-12L;
        printf("forward  R0=%d R=%d L=%d T=%d pairs=%ld  max|ref|=%Lg  max err=%Lg  rel=%Lg  %s\n", R0, R, L, T, n_pairs,
               mx, err, err / mx, ok_f ? "OK" : "FAIL");
        // inverse kernel on the summed spectrum: lag sums against sum_p sum_t conj(z[t]) z[t+n] / (T - n)
        CK(hipMemcpy(d_spec, spec.data(), spec.size() * 8, hipMemcpyHostToDevice));
        auto inv = [&](auto kern) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)P::kLds));
            hipLaunchKernelGGL(kern, dim3(1), dim3(P::NT), P::kLds, 0, d_spec, T, 1L, d_tw, d_lag, 0L, R);
        };
        if (R > 1) inv(k_winverse<P, true, 0>);
        else inv(k_winverse<P, false, 0>);
        CK(hipDeviceSynchronize());
        std::vector<double> lag(T);
        CK(hipMemcpy(lag.data(), d_lag, lag.size() * 8, hipMemcpyDeviceToHost));
        long double lmx = 0, lerr = 0;
        for (int n = 0; n < T; n += std::max(1, T / 97)) {
            long double sref = 0;
            for (long p = 0; p < n_pairs; ++p)
                for (int t = 0; t + n < T; ++t)
                    sref += (long double)h[(p * pitch + t) * 2] * h[(p * pitch + t + n) * 2] +
                            (long double)h[(p * pitch + t) * 2 + 1] * h[(p * pitch + t + n) * 2 + 1];
            sref /= (long double)(T - n);
            lmx = std::max(lmx, fabsl(sref));
            lerr = std::max(lerr, fabsl((long double)lag[n] - sref));
        }
        const bool ok_i = lerr / lmx < 1e-11L;
        printf("inverse  max|ref|=%Lg  max err=%Lg  rel=%Lg  %s\n", lmx, lerr, lerr / lmx, ok_i ? "OK" : "FAIL");
        return ok_f && ok_i ? 0 : 1;
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
    printf("R0=%d R=%d L=%d T=%d pairs=%ld nwg=%d: best %.3f ms  mean %.3f ms  %.1f GB/s  (x150000 pairs: %.2f ms)\n", R0,
           R, L, T, n_pairs, nwg, best, sum / reps, bytes / best * 1e-6, best * 150000.0 / n_pairs);
    if (stamp) {
        std::vector<unsigned long long> st((size_t)nwg * 16);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        for (int h = 0; h < 2; ++h) {
            double s[4] = {0, 0, 0, 0}, tail = 0;
            for (int w = 0; w < nwg; ++w) tail += (double)st[(size_t)w * 16 + 8 + h];
            for (int w = 0; w < nwg; ++w)
                for (int i = 0; i < 4; ++i) s[i] += (double)st[(size_t)w * 16 + h * 4 + i];
            const double per = (double)nwg * ((double)n_pairs / n_tuples);  // unit-passes
            // two stamps per unit: S1 incl. waiting for its rows, S2 incl. the barrier before it
            {  // [2] / [3] = the kernel's span in shader cycles / 100 MHz ticks
                printf("cycles per unit and pass (wave %d): S1 %.0f  S2 %.0f (of which row requests + barrier wait %.0f)  total %.0f   in-kernel clock %.0f MHz\n", 4 * h,
                       s[0] / per, s[1] / per, tail / per, (s[0] + s[1]) / per, s[2] / s[3] * 100.0);
            }
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const char* r = getenv("WF_R0");
    const int R0 = r ? atoi(r) : 20;
    const int R = getenv("WF_R") ? atoi(getenv("WF_R")) : 1;
#ifdef WF_ONLY_R0  // quick builds: one plan
    if (R0 == WF_ONLY_R0) return run<WF_ONLY_R0>(R, argc, argv);
#else
    switch (R0) {
        case 20: return run<20>(R, argc, argv);
        case 18: return run<18>(R, argc, argv);
        case 16: return run<16>(R, argc, argv);
        case 14: return run<14>(R, argc, argv);
        case 12: return run<12>(R, argc, argv);
        case 10: return run<10>(R, argc, argv);
        case 9: return run<9>(R, argc, argv);
        case 8: return run<8>(R, argc, argv);
        case 7: return run<7>(R, argc, argv);
        case 6: return run<6>(R, argc, argv);
        case 5: return run<5>(R, argc, argv);
        case 4: return run<4>(R, argc, argv);
        case 3: return run<3>(R, argc, argv);
        case 2: return run<2>(R, argc, argv);
    }
#endif
    fprintf(stderr, "unsupported WF_R0\n");
    return 2;
}
