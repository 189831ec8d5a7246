// band_test.hip — harness for band_kernels.hpp (MFMA FP64 windowed VACF lag sums).
//   band_test check            small shapes against a plain CPU double loop
//   band_test time T A [D]     synthetic slab, ms per launch and useful TFLOP/s (2 * T(T+1)/2 * A*D flop)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "band_kernels.hpp"
using namespace ta;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct DevPlan {
    BandPlan p;
    BandPiece* pieces = nullptr;
    int *slot_begin = nullptr, *slot_pieces = nullptr, *group_begin = nullptr;
    double* partial = nullptr;
    void upload() {
        CK(hipMalloc(&pieces, sizeof(BandPiece) * p.pieces.size()));
        CK(hipMemcpy(pieces, p.pieces.data(), sizeof(BandPiece) * p.pieces.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&slot_begin, 4 * p.slot_begin.size()));
        CK(hipMemcpy(slot_begin, p.slot_begin.data(), 4 * p.slot_begin.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&slot_pieces, 4 * std::max<size_t>(1, p.slot_pieces.size())));
        CK(hipMemcpy(slot_pieces, p.slot_pieces.data(), 4 * p.slot_pieces.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&group_begin, 4 * p.group_begin.size()));
        CK(hipMemcpy(group_begin, p.group_begin.data(), 4 * p.group_begin.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&partial, sizeof(double) * (size_t)p.n_labels * p.pieces.size() * kBandPartial));
    }
    void free_all() { (void)hipFree(pieces), (void)hipFree(slot_begin), (void)hipFree(slot_pieces), (void)hipFree(group_begin), (void)hipFree(partial); }
};

static bool g_helf = false;  // "helfand" anywhere on the command line: squared differences of the slab's columns
static void run(DevPlan& d, const double* pm, long pitch, int T, long n_cols, double* lagsum, int nwg) {
    const long n_pairs = (n_cols + 1) / 2;
    if (g_helf)
        hipLaunchKernelGGL(k_band_lags<true>, dim3(nwg), dim3(512), 0, 0, pm, pitch, T, n_pairs, d.p.n_labels, d.p.n_ph,
                           d.pieces, (int)d.p.pieces.size(), d.slot_begin, d.slot_pieces, d.partial);
    else
        hipLaunchKernelGGL(k_band_lags<false>, dim3(nwg), dim3(512), 0, 0, pm, pitch, T, n_pairs, d.p.n_labels, d.p.n_ph,
                           d.pieces, (int)d.p.pieces.size(), d.slot_begin, d.slot_pieces, d.partial);
    hipLaunchKernelGGL(k_band_gather, dim3((T + 255) / 256), dim3(256), 0, 0, d.partial, d.p.n_labels, (int)d.p.pieces.size(),
                       d.p.n_ph, d.p.per_phase, d.group_begin, d.p.n_groups, T, g_helf ? -2.0 : 1.0, g_helf ? 1 : 0, lagsum);
}

static double rnd(unsigned long long& s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return ((double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 2.0;
}

static int check_one(int T, long n_cols, int nwg) {
    const long n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
    std::vector<double> h((size_t)n_pairs * pitch * 2);
    unsigned long long s = 1234567 + T * 31 + n_cols;
    for (auto& v : h) v = rnd(s) + (g_helf ? 1000.0 : 0.0);  // (Helfand: a large common offset the centring must remove) pad rows hold garbage on purpose; an unpaired last column is paired with zeros
    if (n_cols & 1)
        for (long t = 0; t < pitch; ++t) h[((n_pairs - 1) * pitch + t) * 2 + 1] = 0.0;
    std::vector<double> ref(T, 0.0);
    for (long c = 0; c < n_cols; ++c) {
        const double* col = h.data() + (c >> 1) * pitch * 2 + (c & 1);
        for (int k = 0; k < T; ++k) {
            double a = 0;
            if (g_helf)
                for (int i = 0; i + k < T; ++i) a += (col[2 * i] - col[2 * (i + k)]) * (col[2 * i] - col[2 * (i + k)]);
            else
                for (int i = 0; i + k < T; ++i) a += col[2 * i] * col[2 * (i + k)];
            ref[k] += a;
        }
    }
    double* pm;
    double* out;
    CK(hipMalloc(&pm, h.size() * 8));
    CK(hipMemcpy(pm, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 8 * T));
    DevPlan d;
    const int nl = 8;
    d.p = band_plan(T, (nwg / nl) * 8, nl);
    d.upload();
    run(d, pm, pitch, T, n_cols, out, nwg);
    CK(hipDeviceSynchronize());
    std::vector<double> got(T);
    CK(hipMemcpy(got.data(), out, 8 * T, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int k = 0; k < T; ++k) scale = std::max(scale, std::fabs(ref[k] / (T - k)));
    int wk = -1;
    for (int k = 0; k < T; ++k) {
        const double e = std::fabs(got[k] - ref[k] / (T - k)) / scale;
        if (e > worst) worst = e, wk = k;
    }
    printf("T=%6d cols=%6ld nwg=%4d groups=%3d n_ph=%2d pieces=%5zu max/mean cost %.1f/%.1f : worst %.2e at lag %d %s\n", T, n_cols, nwg,
           d.p.n_groups, d.p.n_ph, d.p.pieces.size(), d.p.max_cost, d.p.mean_cost, worst, wk, worst < 1e-12 ? "ok" : "FAIL");
    d.free_all();
    (void)hipFree(pm), (void)hipFree(out);
    return worst < 1e-12 ? 0 : 1;
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

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "check";
    for (int k = 1; k < argc; ++k)
        if (!strcmp(argv[k], "helfand")) g_helf = true, argc = k;  // last argument
    if (!strcmp(mode, "check")) {
        int bad = 0;
        const int shapes[][3] = {{1, 3, 256},    {5, 7, 256},     {16, 8, 256},   {17, 9, 256},  {100, 30, 256}, {255, 64, 256},
                                 {256, 5, 256},  {257, 33, 256},  {513, 16, 256}, {1000, 24, 64}, {1000, 24, 256}, {2049, 10, 256},
                                 {4100, 9, 256}, {5000, 16, 256}, {300, 2000, 256}, {9000, 8, 256}};
        for (auto& s : shapes) bad += check_one(s[0], s[1], s[2]);
        printf(bad ? "FAILED %d\n" : "all ok\n", bad);
        return bad ? 1 : 0;
    }
    if (!strcmp(mode, "plan")) {  // host only: how even the cut of the band is
        for (int T : {1, 100, 512, 1000, 2000, 5000, 8192, 10000, 20000, 50000, 100000, 200000}) {
            const BandPlan p = band_plan(T, 256, 8);
            size_t longest = 0;
            for (int s = 0; s < p.slots; ++s) longest = std::max<size_t>(longest, p.slot_begin[s + 1] - p.slot_begin[s]);
            printf("T=%7d groups %5d n_ph %2d pieces/phase %5d, per slot <= %zu pieces, cost max/mean %.1f / %.1f (%.3f)\n", T, p.n_groups,
                   p.n_ph, p.per_phase, longest, p.max_cost, p.mean_cost, p.max_cost / p.mean_cost);
        }
        return 0;
    }
    const int T = argc > 2 ? atoi(argv[2]) : 5000;
    const long A = argc > 3 ? atol(argv[3]) : 50000;
    const int D = argc > 4 ? atoi(argv[4]) : 3;
    const int reps = argc > 5 ? atoi(argv[5]) : 5;
    const long n_cols = A * D, n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nwg = prop.multiProcessorCount;
    double *pm, *out;
    CK(hipMalloc(&pm, (size_t)n_pairs * pitch * 16));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, pm, (size_t)n_pairs * pitch * 2);
    CK(hipMalloc(&out, 8 * T));
    DevPlan d;
    const int labels = getenv("BAND_LABELS") ? atoi(getenv("BAND_LABELS")) : 8;
    d.p = band_plan(T, (nwg / labels) * 8, labels, getenv("BAND_NPH") ? atoi(getenv("BAND_NPH")) : 0);
    d.upload();
    printf("T=%d A=%ld D=%d: groups %d, n_ph %d, pieces %zu, slot cost max/mean %.1f / %.1f steps per octet\n", T, A, D, d.p.n_groups,
           d.p.n_ph, d.p.pieces.size(), d.p.max_cost, d.p.mean_cost);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    run(d, pm, pitch, T, n_cols, out, nwg);
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        run(d, pm, pitch, T, n_cols, out, nwg);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = (g_helf ? 3.0 : 2.0) * (double)T * (T + 1) / 2 * (double)n_cols;
        printf("  %.3f ms  %.1f useful TFLOP/s\n", ms, flop / (ms * 1e-3) / 1e12);
    }
    std::vector<double> h(8);
    CK(hipMemcpy(h.data(), out, 64, hipMemcpyDeviceToHost));
    printf("  lagsum[0..3] = %.6f %.6f %.6f %.6f (expect lag 0 ~ cols/3 = %.1f)\n", h[0], h[1], h[2], h[3], n_cols / 3.0);
    return 0;
}
