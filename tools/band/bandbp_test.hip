// bandbp_test.hip — harness for bandbp_kernels.hpp (windowed VACF with the by-particle array on the FP64 matrix cores).
//   bandbp_test check            small shapes (every dim) against a CPU double loop
//   bandbp_test time T A [reps]  synthetic slab, dim = 3: ms per launch, TFLOP/s (2 flop per term)
//   bandbp_test hcheck | htime T A [reps]   the same for the Einstein-Helfand form (k_band_bp_helf)
//   bandbp_test hltime T A [per]   its lag sums alone; tcheck | ttime T A [per]   the float32 form (k_band32_tp)
// build: tools/band/buildbp.sh [SUFFIX] [-DBP_NW=8]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../transport_analysis_amd/csrc/band32tp_kernels.hpp"
using namespace ta;

#ifndef BP_NW
#define BP_NW 8
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static double rnd(unsigned long long& s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

__global__ void k_fill(double* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 99) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = ((double)((z ^ (z >> 31)) >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 2.0;
    }
}

__global__ void k_fill32(float* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 99) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = (float)(((double)((z ^ (z >> 31)) >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 2.0 + 30.0);
    }
}

template <int D>
static void launch(int nwg, const double* pm, long pitch, int T, long A, double* out, long ld) {
    static unsigned long long* counter = nullptr;
    if (!counter) CK(hipMalloc(&counter, 64));
    CK(hipMemsetAsync(counter, 0, 64, 0));
    hipLaunchKernelGGL((k_band_bp_vacf<D, BP_NW, false>), dim3(nwg), dim3(64 * BP_NW), 0, 0, pm, pitch, T, A, out, ld, counter, 1, (double*)nullptr);
}
static void launch_d(int D, int nwg, const double* pm, long pitch, int T, long A, double* out, long ld) {
    if (D == 1) launch<1>(nwg, pm, pitch, T, A, out, ld);
    else if (D == 2) launch<2>(nwg, pm, pitch, T, A, out, ld);
    else launch<3>(nwg, pm, pitch, T, A, out, ld);
}

template <int D>
static void launch_h(int nwg, const double* pm, long pitch, int T, long A, double* out, long ld) {
    static unsigned long long* counter = nullptr;
    if (!counter) CK(hipMalloc(&counter, 64));
    CK(hipMemsetAsync(counter, 0, 64, 0));
    hipLaunchKernelGGL((k_band_bp_helf<D, BP_NW, false>), dim3(nwg), dim3(64 * BP_NW), 0, 0, pm, pitch, T, A, 1.0, out, ld, counter, 1, (double*)nullptr);
}
static void launch_hd(int D, int nwg, const double* pm, long pitch, int T, long A, double* out, long ld) {
    if (D == 1) launch_h<1>(nwg, pm, pitch, T, A, out, ld);
    else if (D == 2) launch_h<2>(nwg, pm, pitch, T, A, out, ld);
    else launch_h<3>(nwg, pm, pitch, T, A, out, ld);
}

// Helfand form: kind 0 noise around 1000, 1 a drifting random walk, 2 tiny values, 3 a pure cubic trend
static int hcheck_one(int T, long A, int D, int nwg, int kind) {
    const long n_cols = D * A, n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
    std::vector<double> h((size_t)n_pairs * pitch * 2);
    unsigned long long s = 99 + T * 17 + A + 7 * D + kind;
    for (long pr = 0; pr < n_pairs; ++pr)
        for (int c = 0; c < 2; ++c) {
            double walk = 0;
            for (long t = 0; t < pitch; ++t) {
                const double x = (double)t / 100.0 + c;
                double v = kind == 0 ? rnd(s) + 1000.0 : kind == 1 ? (walk += rnd(s) + 0.05) + 300.0 : kind == 2 ? 1e-12 * (rnd(s) + 3.0)
                                                                                                            : x * x * x + 5.0 * pr;
                h[(pr * pitch + t) * 2 + c] = v;  // (rows between T and the pitch hold data too: nothing may read them)
            }
        }
    if (n_cols & 1)
        for (long t = 0; t < pitch; ++t) h[((n_pairs - 1) * pitch + t) * 2 + 1] = 0.0;
    double *pm, *out;
    CK(hipMalloc(&pm, h.size() * 8));
    CK(hipMemcpy(pm, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 8 * (size_t)A * pitch));
    CK(hipMemset(out, 0, 8 * (size_t)A * pitch));
    launch_hd(D, nwg, pm, pitch, T, A, out, pitch);
    CK(hipDeviceSynchronize());
    std::vector<double> got((size_t)A * pitch);
    CK(hipMemcpy(got.data(), out, got.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0, worst_rel = 0;
    long wa = -1;
    int wk = -1;
    for (long a = 0; a < A; ++a) {
        double scale = 0;
        std::vector<double> ref(T, 0.0);
        for (int k = 1; k < T; ++k) {
            double acc = 0;
            for (int d = 0; d < D; ++d) {
                const long c = D * a + d;
                const double* col = h.data() + (c >> 1) * pitch * 2 + (c & 1);
                for (int i = 0; i + k < T; ++i) {
                    const double df = col[2 * i] - col[2 * (i + k)];
                    acc += df * df;
                }
            }
            ref[k] = acc / (T - k);
            scale = std::max(scale, ref[k]);
        }
        for (int k = 0; k < T; ++k) {
            const double e = std::fabs(got[a * pitch + k] - ref[k]) / (scale > 0 ? scale : 1.0);
            if (!(e <= worst)) worst = e, wa = a, wk = k;
            if (ref[k] > 0) worst_rel = std::max(worst_rel, std::fabs(got[a * pitch + k] - ref[k]) / ref[k]);
        }
    }
    const bool ok = worst < 1e-11 && worst_rel < (kind == 3 ? 1e-9 : 1e-10);
    printf("helfand kind %d T=%6d A=%5ld D=%d nwg=%3d : worst %.2e of the particle's scale (particle %ld lag %d), lag by lag %.2e %s\n", kind, T, A,
           D, nwg, worst, wa, wk, worst_rel, ok ? "ok" : "FAIL");
    (void)hipFree(pm), (void)hipFree(out);
    return ok ? 0 : 1;
}

// ---- float32, time-packed (band32tp_kernels.hpp) ----
#ifndef TP_NW
#define TP_NW 12
#endif
template <int D, bool LAGS>
static void launch_t(int nwg, const float* pm, long pitch, int T, long A, double* out, long ld, int per, double* partial) {
    static unsigned long long* counter = nullptr;
    if (!counter) CK(hipMalloc(&counter, 64));
    CK(hipMemsetAsync(counter, 0, 64, 0));
    hipLaunchKernelGGL((k_band32_tp<D, TP_NW, LAGS>), dim3(nwg), dim3(64 * TP_NW), 0, 0, pm, pitch, T, A, 1.0, out, ld, counter, per, partial);
}
static int tcheck_one(int T, long A, int D, int nwg, int kind, int per) {  // per = 0: by particle; > 0: lag sums with `per` particles per unit
    const long n_cols = D * A, n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
    std::vector<float> h((size_t)n_pairs * pitch * 2);
    unsigned long long s = 99 + T * 17 + A + 7 * D + kind;
    for (long pr = 0; pr < n_pairs; ++pr)
        for (int c = 0; c < 2; ++c) {
            double walk = 0;
            for (long t = 0; t < pitch; ++t) {
                double v = kind == 0 ? rnd(s) + 1000.0 : kind == 1 ? (walk += rnd(s) + 0.05) + 300.0 : 1e-12 * (rnd(s) + 3.0);
                h[(pr * pitch + t) * 2 + c] = (float)v;
            }
        }
    if (n_cols & 1)
        for (long t = 0; t < pitch; ++t) h[((n_pairs - 1) * pitch + t) * 2 + 1] = 0.0f;
    float* pm;
    double *out, *partial = nullptr, *lagsum = nullptr;
    CK(hipMalloc(&pm, h.size() * 4));
    CK(hipMemcpy(pm, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 8 * (size_t)A * pitch));
    CK(hipMemset(out, 0, 8 * (size_t)A * pitch));
    const int n_groups = ((T + 15) / 16 + 15) / 16;
    const long n_pb = per ? (A + per - 1) / per : 0;
    if (per) {
        CK(hipMalloc(&partial, 8 * (size_t)n_pb * n_groups * kBandPartial));
        CK(hipMalloc(&lagsum, 8 * (size_t)T));
        if (D == 1) launch_t<1, true>(nwg, pm, pitch, T, A, nullptr, 0, per, partial);
        else if (D == 2) launch_t<2, true>(nwg, pm, pitch, T, A, nullptr, 0, per, partial);
        else launch_t<3, true>(nwg, pm, pitch, T, A, nullptr, 0, per, partial);
        hipLaunchKernelGGL(k_bandbp_gather, dim3((T + 15) / 16), dim3(256), 0, 0, partial, n_pb, n_groups, T, 1.0, 1, lagsum);
    } else {
        if (D == 1) launch_t<1, false>(nwg, pm, pitch, T, A, out, pitch, 1, nullptr);
        else if (D == 2) launch_t<2, false>(nwg, pm, pitch, T, A, out, pitch, 1, nullptr);
        else launch_t<3, false>(nwg, pm, pitch, T, A, out, pitch, 1, nullptr);
    }
    CK(hipDeviceSynchronize());
    std::vector<double> got((size_t)A * pitch), gl(T, 0.0);
    CK(hipMemcpy(got.data(), out, got.size() * 8, hipMemcpyDeviceToHost));
    if (per) CK(hipMemcpy(gl.data(), lagsum, 8 * (size_t)T, hipMemcpyDeviceToHost));
    double worst = 0;
    long wa = -1;
    int wk = -1;
    std::vector<double> tot(T, 0.0);
    for (long a = 0; a < A; ++a) {
        double scale = 0;
        std::vector<double> ref(T, 0.0);
        for (int k = 1; k < T; ++k) {
            double acc = 0;
            for (int d = 0; d < D; ++d) {
                const long c = D * a + d;
                const float* col = h.data() + (c >> 1) * pitch * 2 + (c & 1);
                for (int i = 0; i + k < T; ++i) {
                    const double df = (double)col[2 * i] - (double)col[2 * (i + k)];
                    acc += df * df;
                }
            }
            ref[k] = acc / (T - k);
            tot[k] += ref[k];
            scale = std::max(scale, ref[k]);
        }
        if (!per)
            for (int k = 0; k < T; ++k) {
                const double e = std::fabs(got[a * pitch + k] - ref[k]) / (scale > 0 ? scale : 1.0);
                if (!(e <= worst)) worst = e, wa = a, wk = k;
            }
    }
    if (per) {
        double scale = 0;
        for (int k = 0; k < T; ++k) scale = std::max(scale, tot[k]);
        for (int k = 0; k < T; ++k) {
            const double e = std::fabs(gl[k] - tot[k]) / (scale > 0 ? scale : 1.0);
            if (!(e <= worst)) worst = e, wk = k;
        }
    }
    if (getenv("TP_DEBUG") && !per) {
        const long a = wa < 0 ? 0 : wa;
        std::vector<double> ref(T, 0.0);
        for (int k = 1; k < T; ++k) {
            double acc = 0;
            for (int d = 0; d < D; ++d) {
                const long c = D * a + d;
                const float* col = h.data() + (c >> 1) * pitch * 2 + (c & 1);
                for (int i = 0; i + k < T; ++i) { const double df = (double)col[2 * i] - (double)col[2 * (i + k)]; acc += df * df; }
            }
            ref[k] = acc / (T - k);
        }
        for (int k = 0; k < T; ++k) {
            const double e = (got[a * pitch + k] - ref[k]) * (T - k);
            if (std::fabs(e) > 1e-3 * std::fabs(ref[k] * (T - k))) printf("   lag %4d (off %3d): got-ref (unnormalised) %.6g  ref %.6g\n", k, k & 255, e, ref[k] * (T - k));
        }
    }
    const bool ok = worst < 2e-6;
    printf("float32 %s kind %d T=%6d A=%5ld D=%d nwg=%3d : worst %.2e of the scale (particle %ld lag %d) %s\n", per ? "lag sums   " : "by particle", kind, T, A, D,
           nwg, worst, wa, wk, ok ? "ok" : "FAIL");
    (void)hipFree(pm), (void)hipFree(out), (void)hipFree(partial), (void)hipFree(lagsum);
    return ok ? 0 : 1;
}

static int check_one(int T, long A, int D, int nwg) {
    const long n_cols = D * A, n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
    std::vector<double> h((size_t)n_pairs * pitch * 2);
    unsigned long long s = 99 + T * 17 + A + 7 * D;
    // the rows between T and the pitch hold garbage on purpose: nothing may read them as data
    for (auto& x : h) x = rnd(s) + 0.25;
    if (n_cols & 1)
        for (long t = 0; t < pitch; ++t) h[((n_pairs - 1) * pitch + t) * 2 + 1] = 0.0;
    double *pm, *out;
    CK(hipMalloc(&pm, h.size() * 8));
    CK(hipMemcpy(pm, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 8 * (size_t)A * pitch));
    CK(hipMemset(out, 0, 8 * (size_t)A * pitch));
    launch_d(D, nwg, pm, pitch, T, A, out, pitch);
    CK(hipDeviceSynchronize());
    std::vector<double> got((size_t)A * pitch);
    CK(hipMemcpy(got.data(), out, got.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    long wa = -1;
    int wk = -1;
    for (long a = 0; a < A; ++a) {
        double scale = 0;
        std::vector<double> ref(T, 0.0);
        for (int k = 0; k < T; ++k) {
            double acc = 0;
            for (int d = 0; d < D; ++d) {
                const long c = D * a + d;
                const double* col = h.data() + (c >> 1) * pitch * 2 + (c & 1);
                for (int i = 0; i + k < T; ++i) acc += col[2 * i] * col[2 * (i + k)];
            }
            ref[k] = acc / (T - k);
            scale = std::max(scale, std::fabs(ref[k]));
        }
        for (int k = 0; k < T; ++k) {
            const double e = std::fabs(got[a * pitch + k] - ref[k]) / (scale > 0 ? scale : 1.0);
            if (!(e <= worst)) worst = e, wa = a, wk = k;
        }
    }
    const bool ok = worst < 1e-12;
    printf("T=%6d A=%5ld D=%d nwg=%3d : worst %.2e of the particle's scale (particle %ld lag %d) %s\n", T, A, D, nwg, worst, wa, wk, ok ? "ok" : "FAIL");
    (void)hipFree(pm), (void)hipFree(out);
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "check";
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    if (!strcmp(mode, "check")) {
        int bad = 0;
        const int shapes[][2] = {{1, 2}, {2, 3}, {15, 4}, {16, 5}, {17, 3}, {63, 2}, {64, 3}, {65, 2}, {239, 2}, {240, 3}, {241, 2}, {255, 3},
                                 {257, 7}, {480, 3}, {481, 5}, {511, 2}, {513, 2}, {1000, 9}, {2049, 3}, {5000, 2}, {300, 700}};
        for (int D = 1; D <= 3; ++D)
            for (auto& sh : shapes) bad += check_one(sh[0], sh[1], D, D == 3 ? 256 : 7);
        printf(bad ? "FAILED %d\n" : "all ok\n", bad);
        return bad ? 1 : 0;
    }
    if (!strcmp(mode, "hcheck")) {
        int bad = 0;
        const int shapes[][2] = {{1, 2}, {2, 3}, {15, 4}, {16, 5}, {17, 3}, {63, 2}, {64, 3}, {65, 2}, {239, 2}, {240, 3}, {241, 2}, {255, 3},
                                 {257, 7}, {449, 3}, {480, 3}, {481, 5}, {511, 2}, {513, 2}, {1000, 9}, {2049, 3}, {5000, 2}, {300, 700}};
        for (int D = 1; D <= 3; ++D)
            for (auto& sh : shapes) bad += hcheck_one(sh[0], sh[1], D, D == 3 ? 256 : 7, 0);
        for (int kind : {1, 2, 3}) bad += hcheck_one(1000, 5, 3, 256, kind) + hcheck_one(4100, 2, 3, 256, kind) + hcheck_one(700, 3, 2, 3, kind);
        printf(bad ? "FAILED %d\n" : "all ok\n", bad);
        return bad ? 1 : 0;
    }
    if (!strcmp(mode, "tdebug")) return tcheck_one(argc > 2 ? atoi(argv[2]) : 1000, argc > 3 ? atol(argv[3]) : 1, argc > 4 ? atoi(argv[4]) : 1, 7, 0, 0);
    if (!strcmp(mode, "tcheck")) {
        int bad = 0;
        const int shapes[][2] = {{1, 2}, {2, 3}, {15, 4}, {16, 5}, {17, 3}, {63, 2}, {64, 3}, {65, 2}, {239, 2}, {240, 3}, {241, 2}, {255, 3},
                                 {257, 7}, {449, 3}, {480, 3}, {481, 5}, {511, 2}, {513, 2}, {1000, 9}, {2049, 3}, {5000, 2}, {300, 700}};
        for (int D = 1; D <= 3; ++D)
            for (auto& sh : shapes) bad += tcheck_one(sh[0], sh[1], D, D == 3 ? 256 : 7, 0, 0) + tcheck_one(sh[0], sh[1], D, D == 3 ? 256 : 7, 0, D == 2 ? 1 : 4);
        for (int kind : {1, 2}) bad += tcheck_one(1000, 5, 3, 256, kind, 0) + tcheck_one(4100, 2, 3, 256, kind, 0) + tcheck_one(4100, 7, 3, 256, kind, 3);
        printf(bad ? "FAILED %d\n" : "all ok\n", bad);
        return bad ? 1 : 0;
    }
    if (!strcmp(mode, "ttime")) {  // float32: by particle (per = 0) or lag sums with `per` particles per unit
        const int T = argc > 2 ? atoi(argv[2]) : 20000;
        const long A = argc > 3 ? atol(argv[3]) : 25000;
        const int per = argc > 4 ? atoi(argv[4]) : 0;
        const long n_cols = 3 * A, n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
        const int n_groups = ((T + 15) / 16 + 15) / 16;
        const long n_pb = per ? (A + per - 1) / per : 1;
        float* pm;
        double *out, *partial, *lagsum;
        CK(hipMalloc(&pm, (size_t)n_pairs * pitch * 8));
        hipLaunchKernelGGL(k_fill32, dim3(4096), dim3(256), 0, 0, pm, (size_t)n_pairs * pitch * 2);
        CK(hipMalloc(&out, 8 * (size_t)A * pitch));
        CK(hipMalloc(&partial, 8 * (size_t)n_pb * n_groups * kBandPartial));
        CK(hipMalloc(&lagsum, 8 * (size_t)T));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, 0));
            if (per) {
                launch_t<3, true>(prop.multiProcessorCount, pm, pitch, T, A, nullptr, 0, per, partial);
                hipLaunchKernelGGL(k_bandbp_gather, dim3((T + 15) / 16), dim3(256), 0, 0, partial, n_pb, n_groups, T, 1.0, 1, lagsum);
            } else {
                CK(hipMemsetAsync(out, 0, 8 * (size_t)A * pitch, 0));
                launch_t<3, false>(prop.multiProcessorCount, pm, pitch, T, A, out, pitch, 1, nullptr);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) printf("  Helfand float32 time-packed (%s, %d waves per workgroup) %d x %ld x 3: %.3f ms  %.1f TFLOP/s (2 flop per term)\n",
                          per ? "lag sums" : "by particle", TP_NW, T, A, ms, 2.0 * (double)T * (T - 1) / 2 * n_cols / (ms * 1e-3) / 1e12);
        }
        return 0;
    }
    if (!strcmp(mode, "ltime")) {  // windowed VACF lag sums alone (per_unit particles per unit)
        const int T = argc > 2 ? atoi(argv[2]) : 5000;
        const long A = argc > 3 ? atol(argv[3]) : 50000;
        const int per = argc > 4 ? atoi(argv[4]) : 16;
        const long n_cols = 3 * A, n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
        const int n_groups = ((T + 15) / 16 + 15) / 16;
        const long n_pb = (A + per - 1) / per;
        double *pm, *partial, *lagsum;
        unsigned long long* counter;
        CK(hipMalloc(&pm, (size_t)n_pairs * pitch * 16));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, pm, (size_t)n_pairs * pitch * 2);
        CK(hipMalloc(&partial, 8 * (size_t)n_pb * n_groups * kBandPartial));
        CK(hipMalloc(&lagsum, 8 * (size_t)T));
        CK(hipMalloc(&counter, 64));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, 0));
            CK(hipMemsetAsync(counter, 0, 64, 0));
            hipLaunchKernelGGL((k_band_bp_vacf<3, BP_NW, true>), dim3(prop.multiProcessorCount), dim3(64 * BP_NW), 0, 0, pm, pitch, T, A,
                               (double*)nullptr, 0L, counter, per, partial);
            hipLaunchKernelGGL(k_bandbp_gather, dim3((T + 15) / 16), dim3(256), 0, 0, partial, n_pb, n_groups, T, 1.0, 0, lagsum);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) printf("  windowed VACF lag sums (FP64 matrix cores, %d particles per unit) %d x %ld x 3: %.3f ms  %.1f TFLOP/s\n", per, T, A, ms,
                          2.0 * (double)T * (T + 1) / 2 * n_cols / (ms * 1e-3) / 1e12);
        }
        return 0;
    }
    if (!strcmp(mode, "hltime")) {  // lag sums alone through the same kernel (per_unit particles per unit)
        const int T = argc > 2 ? atoi(argv[2]) : 20000;
        const long A = argc > 3 ? atol(argv[3]) : 25000;
        const int per = argc > 4 ? atoi(argv[4]) : 16;
        const long n_cols = 3 * A, n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
        const int n_groups = ((T + 15) / 16 + 15) / 16;
        const long n_pb = (A + per - 1) / per;
        double *pm, *partial, *lagsum;
        unsigned long long* counter;
        CK(hipMalloc(&pm, (size_t)n_pairs * pitch * 16));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, pm, (size_t)n_pairs * pitch * 2);
        CK(hipMalloc(&partial, 8 * (size_t)n_pb * n_groups * kBandPartial));
        CK(hipMalloc(&lagsum, 8 * (size_t)T));
        CK(hipMalloc(&counter, 64));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, 0));
            CK(hipMemsetAsync(counter, 0, 64, 0));
            hipLaunchKernelGGL((k_band_bp_helf<3, BP_NW, true>), dim3(prop.multiProcessorCount), dim3(64 * BP_NW), 0, 0, pm, pitch, T, A, 1.0,
                               (double*)nullptr, 0L, counter, per, partial);
            hipLaunchKernelGGL(k_bandbp_gather, dim3((T + 15) / 16), dim3(256), 0, 0, partial, n_pb, n_groups, T, 1.0, 1, lagsum);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) printf("  Helfand lag sums (FP64 matrix cores, %d particles per unit) %d x %ld x 3: %.3f ms  %.1f TFLOP/s (2 flop per term)\n", per, T, A,
                          ms, 2.0 * (double)T * (T - 1) / 2 * n_cols / (ms * 1e-3) / 1e12);
        }
        return 0;
    }
    const bool helf = !strcmp(mode, "htime");
    const int T = argc > 2 ? atoi(argv[2]) : 5000;
    const long A = argc > 3 ? atol(argv[3]) : 50000;
    const int reps = argc > 4 ? atoi(argv[4]) : 3;
    const long n_cols = 3 * A, n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
    double *pm, *out;
    CK(hipMalloc(&pm, (size_t)n_pairs * pitch * 16));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, pm, (size_t)n_pairs * pitch * 2);
    CK(hipMalloc(&out, 8 * (size_t)A * pitch));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int wg_per_cu = getenv("BP_WG_PER_CU") ? atoi(getenv("BP_WG_PER_CU")) : 1;
    for (int r = 0; r <= reps; ++r) {
        CK(hipEventRecord(e0, 0));
        CK(hipMemsetAsync(out, 0, 8 * (size_t)A * pitch, 0));
        if (helf) launch_h<3>(prop.multiProcessorCount * wg_per_cu, pm, pitch, T, A, out, pitch);
        else launch<3>(prop.multiProcessorCount * wg_per_cu, pm, pitch, T, A, out, pitch);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) printf("  by-particle %s (FP64 matrix cores) %d x %ld x 3: %.3f ms  %.1f TFLOP/s (2 flop per term)\n", helf ? "Helfand" : "windowed VACF", T, A, ms,
                      2.0 * (double)T * (T + 1) / 2 * n_cols / (ms * 1e-3) / 1e12);
    }
    return 0;
}
