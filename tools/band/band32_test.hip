// band32_test.hip — harness for band32_kernels.hpp (Einstein-Helfand lag sums on the FP32 matrix cores).
//   band32_test check            small shapes against a CPU double loop over the float32-rounded slab
//   band32_test time T A [D] [reps]   synthetic slab: ms per launch, useful TFLOP/s (3 flop per term, the
//                                reference's count) and the matrix pipe's issued TFLOP/s
// build: tools/band/build32.sh [SUFFIX] [-DB32_NW=4|8] [-DB32_PF=1|2|4|8] [-DB32_NS=..]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "band32_kernels.hpp"
using namespace ta;

#ifndef B32_NW  // defaults = what band32.hip launches: 8 waves per workgroup (two per SIMD), requests 2 steps ahead
#define B32_NW 8
#endif
#ifndef B32_PF
#define B32_PF 2
#endif
#ifndef B32_NS
#define B32_NS (2 * B32_PF)
#endif

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

static unsigned long long* g_stamps = nullptr;
static void run(DevPlan& d, const float* pm, long pitch, int T, long n_cols, double* lagsum, int nwg) {
    const long n_pairs = (n_cols + 1) / 2;
    hipLaunchKernelGGL((k_band32_lags<B32_NW, B32_PF, B32_NS>), dim3(nwg), dim3(64 * B32_NW), 0, 0, pm, pitch, T, n_pairs, d.p.n_labels,
                       d.p.n_ph, d.pieces, (int)d.p.pieces.size(), d.slot_begin, d.slot_pieces, d.partial, g_stamps);
    hipLaunchKernelGGL(k_band_gather, dim3((T + 255) / 256), dim3(256), 0, 0, d.partial, d.p.n_labels, (int)d.p.pieces.size(),
                       d.p.n_ph, d.p.per_phase, d.group_begin, d.p.n_groups, T, -2.0, 1, lagsum);
}

static double rnd(unsigned long long& s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return ((double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 2.0;
}

// kind 0: noise + a large offset; 1: a random walk with drift (smooth: short lags far below the scale);
// 2: noise scaled by 1e-12 (the result must not depend on the unit); 3: a pure cubic trend
static int check_one(int T, long n_cols, int nwg, int kind) {
    const long n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
    std::vector<float> h((size_t)n_pairs * pitch * 2);
    unsigned long long s = 1234567 + T * 31 + n_cols + 77 * kind;
    for (long pr = 0; pr < n_pairs; ++pr)
        for (int c = 0; c < 2; ++c) {
            double walk = 0;
            for (long t = 0; t < pitch; ++t) {
                double v;
                if (kind == 0) v = rnd(s) + 1000.0;
                else if (kind == 1) v = (walk += rnd(s) + 0.05) + 300.0;
                else if (kind == 2) v = 1e-12 * (rnd(s) + 3.0);
                else v = 0.5 * (double)t * t * t * (1.0 + 0.5 * c + pr);
                h[(pr * pitch + t) * 2 + c] = (float)v;  // pad rows hold garbage on purpose
            }
        }
    if (n_cols & 1)
        for (long t = 0; t < pitch; ++t) h[((n_pairs - 1) * pitch + t) * 2 + 1] = 0.0f;
    std::vector<double> ref(T, 0.0);
    for (long c = 0; c < n_cols; ++c) {
        const float* col = h.data() + (c >> 1) * pitch * 2 + (c & 1);
        for (int k = 0; k < T; ++k) {
            double a = 0;
            for (int i = 0; i + k < T; ++i) {
                const double df = (double)col[2 * i] - (double)col[2 * (i + k)];
                a += df * df;
            }
            ref[k] += a;
        }
    }
    float* pm;
    double* out;
    CK(hipMalloc(&pm, h.size() * 4));
    CK(hipMemcpy(pm, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 8 * T));
    DevPlan d;
    const int nl = 8;
    d.p = band_plan(T, (nwg / nl) * B32_NW, nl);
    d.upload();
    run(d, pm, pitch, T, n_cols, out, nwg);
    CK(hipDeviceSynchronize());
    std::vector<double> got(T);
    CK(hipMemcpy(got.data(), out, 8 * T, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0, worst_rel = 0;
    for (int k = 0; k < T; ++k) scale = std::max(scale, std::fabs(ref[k] / (T - k)));
    int wk = -1, wrk = -1;
    for (int k = 0; k < T; ++k) {
        const double want = ref[k] / (T - k);
        const double e = scale > 0 ? std::fabs(got[k] - want) / scale : std::fabs(got[k]);
        if (e > worst) worst = e, wk = k;
        if (k > 0 && want > 0) {
            const double r = std::fabs(got[k] - want) / want;
            if (r > worst_rel) worst_rel = r, wrk = k;
        }
    }
    const bool ok = worst < 2e-6 && got[0] == 0.0;
    printf("kind %d T=%6d cols=%6ld nwg=%4d groups=%3d n_ph=%2d pieces=%5zu : worst %.2e of the scale at lag %d, lag by lag %.2e at %d %s\n",
           kind, T, n_cols, nwg, d.p.n_groups, d.p.n_ph, d.p.pieces.size(), worst, wk, worst_rel, wrk, ok ? "ok" : "FAIL");
    if (!ok && getenv("B32_VERBOSE")) {
        for (int k = 0; k < T && k < 20; ++k) printf("    lag %d: got %.9g want %.9g\n", k, got[k], ref[k] / (T - k));
    }
    d.free_all();
    (void)hipFree(pm), (void)hipFree(out);
    return ok ? 0 : 1;
}

__global__ void k_fill(float* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 99) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = (float)(((double)((z ^ (z >> 31)) >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 2.0 + 30.0);
    }
}

// by-particle kernel: every particle's own mean squared differences, atom-major output
static int bp_check_one(int T, long A, int kind) {
    const long n_cols = 3 * A, n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
    std::vector<float> h((size_t)n_pairs * pitch * 2);
    unsigned long long s = 99 + T * 17 + A + 7 * kind;
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
    double* out;
    CK(hipMalloc(&pm, h.size() * 4));
    CK(hipMemcpy(pm, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 8 * (size_t)A * pitch));
    CK(hipMemset(out, 0xff, 8 * (size_t)A * pitch));
    hipLaunchKernelGGL((k_band32_bp<B32_NW, B32_PF, B32_NS>), dim3(256), dim3(64 * B32_NW), 0, 0, pm, pitch, T, A, 1.0, out, pitch);
    CK(hipDeviceSynchronize());
    std::vector<double> got((size_t)A * pitch);
    CK(hipMemcpy(got.data(), out, got.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    long wa = -1;
    int wk = -1;
    for (long a = 0; a < A; ++a) {
        double scale = 0;
        std::vector<double> ref(T, 0.0);
        for (int k = 1; k < T; ++k) {
            double acc = 0;
            for (int d = 0; d < 3; ++d) {
                const long c = 3 * a + d;
                const float* col = h.data() + (c >> 1) * pitch * 2 + (c & 1);
                for (int i = 0; i + k < T; ++i) {
                    const double df = (double)col[2 * i] - (double)col[2 * (i + k)];
                    acc += df * df;
                }
            }
            ref[k] = acc / (T - k);
            scale = std::max(scale, ref[k]);
        }
        for (int k = 0; k < T; ++k) {
            const double e = std::fabs(got[a * pitch + k] - ref[k]) / (scale > 0 ? scale : 1.0);
            if (!(e <= worst)) worst = e, wa = a, wk = k;
        }
    }
    const bool ok = worst < 2e-6;
    printf("by-particle kind %d T=%6d A=%5ld : worst %.2e of the particle's scale (particle %ld lag %d) %s\n", kind, T, A, worst, wa, wk, ok ? "ok" : "FAIL");
    (void)hipFree(pm), (void)hipFree(out);
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "check";
    if (!strcmp(mode, "bpcheck")) {
        int bad = 0;
        const int shapes[][2] = {{1, 2}, {2, 3}, {15, 4}, {16, 5}, {17, 3}, {239, 2}, {240, 3}, {241, 2}, {255, 3}, {257, 7}, {480, 3},
                                 {481, 5}, {1000, 9}, {2049, 3}, {5000, 2}, {300, 700}};
        for (auto& sh : shapes) bad += bp_check_one(sh[0], sh[1], 0);
        for (int kind : {1, 2}) bad += bp_check_one(1000, 5, kind) + bp_check_one(4100, 2, kind);
        printf(bad ? "FAILED %d\n" : "all ok\n", bad);
        return bad ? 1 : 0;
    }
    if (!strcmp(mode, "bptime")) {
        const int T = argc > 2 ? atoi(argv[2]) : 20000;
        const long A = argc > 3 ? atol(argv[3]) : 25000;
        const int reps = argc > 4 ? atoi(argv[4]) : 2;
        const long n_cols = 3 * A, n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
        hipDeviceProp_t prop;
        CK(hipGetDeviceProperties(&prop, 0));
        float* pm;
        double* out;
        CK(hipMalloc(&pm, (size_t)n_pairs * pitch * 8));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, pm, (size_t)n_pairs * pitch * 2);
        CK(hipMalloc(&out, 8 * (size_t)A * pitch));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int wg_per_cu = getenv("B32_WG_PER_CU") ? atoi(getenv("B32_WG_PER_CU")) : 1;
        for (int r = 0; r <= reps; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_band32_bp<B32_NW, B32_PF, B32_NS>), dim3(prop.multiProcessorCount * wg_per_cu), dim3(64 * B32_NW), 0, 0, pm,
                               pitch, T, A, 1.0, out, pitch);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) printf("  by-particle Helfand (float32 matrix cores) %d x %ld x 3: %.3f ms  %.1f TFLOP/s (3 flop per term)\n", T, A, ms,
                          3.0 * (double)T * (T - 1) / 2 * n_cols / (ms * 1e-3) / 1e12);
        }
        return 0;
    }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int wg_per_cu = getenv("B32_WG_PER_CU") ? atoi(getenv("B32_WG_PER_CU")) : 1;
    const int nwg_full = prop.multiProcessorCount * wg_per_cu;
    printf("# band32_test NW=%d PF=%d NS=%d, %d workgroups of %d threads\n", B32_NW, B32_PF, B32_NS, nwg_full, 64 * B32_NW);
    if (!strcmp(mode, "check")) {
        int bad = 0;
        const int shapes[][3] = {{1, 3, 256},    {2, 6, 256},     {5, 7, 256},     {16, 8, 256},   {17, 9, 256},   {100, 30, 256},
                                 {255, 64, 256}, {256, 5, 256},   {257, 33, 256},  {513, 16, 256}, {1000, 24, 64}, {1000, 24, 256},
                                 {2049, 10, 256}, {4100, 9, 256}, {5000, 16, 256}, {300, 2000, 256}, {9000, 8, 256}};
        for (auto& s : shapes) bad += check_one(s[0], s[1], s[2], 0);
        for (int kind : {1, 2, 3})
            for (auto& s : {shapes[5], shapes[9], shapes[12], shapes[14]}) bad += check_one(s[0], s[1], s[2], kind);
        printf(bad ? "FAILED %d\n" : "all ok\n", bad);
        return bad ? 1 : 0;
    }
    const int T = argc > 2 ? atoi(argv[2]) : 20000;
    const long A = argc > 3 ? atol(argv[3]) : 25000;
    const int D = argc > 4 ? atoi(argv[4]) : 3;
    const int reps = argc > 5 ? atoi(argv[5]) : 3;
    const long n_cols = A * D, n_pairs = (n_cols + 1) / 2, pitch = (T + 7) / 8 * 8;
    float* pm;
    double* out;
    CK(hipMalloc(&pm, (size_t)n_pairs * pitch * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, pm, (size_t)n_pairs * pitch * 2);
    CK(hipMalloc(&out, 8 * T));
    DevPlan d;
    d.p = band_plan(T, (nwg_full / 8) * B32_NW, 8, getenv("BAND_NPH") ? atoi(getenv("BAND_NPH")) : 0);
    d.upload();
    printf("T=%d A=%ld D=%d: groups %d, n_ph %d, pieces %zu, slot cost max/mean %.1f / %.1f steps per sextet\n", T, A, D, d.p.n_groups,
           d.p.n_ph, d.p.pieces.size(), d.p.max_cost, d.p.mean_cost);
    CK(hipMalloc(&g_stamps, 16 * (size_t)nwg_full * B32_NW));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    run(d, pm, pitch, T, n_cols, out, nwg_full);
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        run(d, pm, pitch, T, n_cols, out, nwg_full);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double terms = (double)T * (T - 1) / 2 * (double)n_cols;
        printf("  %.3f ms  %.1f reference TFLOP/s (3 per term), matrix pipe issues %.1f TFLOP/s (2 per term on 6 of 8 slots)\n", ms,
               3.0 * terms / (ms * 1e-3) / 1e12, 2.0 * (8.0 / 6.0) * terms / (ms * 1e-3) / 1e12);
    }
    {   // the clock the kernel held and its cycles per step (busiest wave), from the last launch's stamps
        std::vector<unsigned long long> st(2 * (size_t)nwg_full * B32_NW);
        CK(hipMemcpy(st.data(), g_stamps, st.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, ticks = 0, cmax = 0;
        for (size_t w = 0; w < st.size() / 2; ++w) cyc += (double)st[2 * w], ticks += (double)st[2 * w + 1], cmax = std::max(cmax, (double)st[2 * w]);
        const long n_sext = (n_pairs + 2) / 3;
        const double steps_busiest = d.p.max_cost * (double)((n_sext + 7) / 8) / d.p.n_ph;
        printf("  in-kernel clock %.0f MHz (mean over waves), busiest wave %.3e cycles = %.0f per step (%.0f MFMA cycles of them)\n",
               cyc / ticks * 100.0, cmax, cmax / steps_busiest, 32.0 * 32.0);
    }
    std::vector<double> hh(4);
    CK(hipMemcpy(hh.data(), out, 32, hipMemcpyDeviceToHost));
    printf("  lagsum[0..3] = %.6f %.6f %.6f %.6f (uniform [-1,1) noise: lags >= 1 ~ 2/3 cols = %.1f)\n", hh[0], hh[1], hh[2], hh[3], 2.0 / 3.0 * n_cols);
    return 0;
}
