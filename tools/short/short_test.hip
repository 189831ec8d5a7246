// short_test.hip — k_short (transport_analysis_amd/csrc/short_kernels.hpp) alone on a synthetic slab: timing of one
// variant (TMAX from -DTM=32|64, MODE from -DMD=0|1, dim from -DDM=1|2|3), optionally with parts of the kernel compiled out
// (-DSHORT_ABL=bits, tools/short/ablations.patch).   usage: short_test T n_atoms by_particle(0|1) [reps [pitch]]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "short_kernels.hpp"

#ifndef TM
#define TM 64
#endif
#ifndef MD
#define MD 0
#endif
#ifndef DM
#define DM 3
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_fill(double* p, size_t n, unsigned long long seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31;
        p[i] = (double)(long long)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}

int main(int argc, char** argv) {
    using namespace ta;
    const int T = argc > 1 ? atoi(argv[1]) : 64;
    const long A = argc > 2 ? atol(argv[2]) : 7812480;
    const int by_particle = argc > 3 ? atoi(argv[3]) : 1;
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const int D = DM;
    if (T > TM) { printf("T > TM\n"); return 1; }
    const long pitch = argc > 5 ? atol(argv[5]) : (T + 7) / 8 * 8, n_pairs = (A * D + 1) / 2;
    const size_t n_el = (size_t)n_pairs * pitch * 2;
    double *vel, *pos = nullptr, *masses = nullptr, *bp = nullptr, *partial;
    CK(hipMalloc(&vel, n_el * 8));
    k_fill<<<4096, 256>>>(vel, n_el, 1);
    if (MD == 1) {
        CK(hipMalloc(&pos, n_el * 8));
        k_fill<<<4096, 256>>>(pos, n_el, 77);
        CK(hipMalloc(&masses, A * 8));
        k_fill<<<4096, 256>>>(masses, A, 5);
    }
    if (by_particle) CK(hipMalloc(&bp, (size_t)T * A * 8));
    const size_t lds = by_particle ? ShortCfg<TM, true>::kLds : ShortCfg<TM, false>::kLds;
    auto fn = by_particle ? k_short<TM, MD, DM, true> : k_short<TM, MD, DM, false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nwg = prop.multiProcessorCount * per_cu;
    CK(hipMalloc(&partial, (size_t)nwg * 4 * T * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms(reps);
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(fn, dim3(nwg), dim3(256), lds, 0, vel, pos, masses, pitch, T, A, 1.0, bp, A, partial);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[r], e0, e1));
    }
    std::vector<double> h(T);
    CK(hipMemcpy(h.data(), partial, T * 8, hipMemcpyDeviceToHost));
    const double gb = ((double)T * A * D * 8 * (MD == 1 ? 2 : 1) + (by_particle ? (double)T * A * 8 : 0)) / 1e9;
    float best = ms[1];
    for (int r = 1; r < reps; ++r) best = ms[r] < best ? ms[r] : best;
    printf("TM=%d MD=%d ABL=%d T=%d A=%ld bp=%d wg/CU=%d: %.3f ms (%.2f TB/s)  [partial0 %.6g]\n", TM, MD,
#ifdef SHORT_ABL
           SHORT_ABL,
#else
           0,
#endif
           T, A, by_particle, per_cu, best, gb / best, h[0]);
    return 0;
}
