// dft8_bench.hip — FP64 issue efficiency of the radix-8 butterfly code of csrc/wfft.hpp in
// isolation (registers only, no LDS): cycles per sub-series' worth of arithmetic (3 DFT8 + 14
// twiddle products + |.|^2 accumulation = ~236 FP64 instructions) at 1 and 2 waves per SIMD,
// one stream or two interleaved streams per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../transport_analysis_amd/csrc/fft_engine.hpp"
using namespace ta;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void body(cd (&v)[8], const cd (&twa)[7], const cd (&twb)[7], double (&acc)[8]) {
    Dft<8>::run(v);
#pragma unroll
    for (int a = 1; a < 8; ++a) v[a] = cmul(v[a], twa[a - 1]);
    Dft<8>::run(v);
#pragma unroll
    for (int a = 1; a < 8; ++a) v[a] = cmul(v[a], twb[a - 1]);
    Dft<8>::run(v);
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = fma(v[c].y, v[c].y, fma(v[c].x, v[c].x, acc[c]));
}

template <int STREAMS>
__global__ void __launch_bounds__(512) k(unsigned long long* out, double* sink, int n) {
    const int tid = threadIdx.x;
    cd twa[7], twb[7], v[STREAMS][8];
    double acc[STREAMS][8];
#pragma unroll
    for (int i = 0; i < 7; ++i) { twa[i] = cd{cos(0.01 * (tid + i)), sin(0.01 * (tid + i))}; twb[i] = cd{cos(0.02 * (tid + i)), sin(0.02 * (tid + i))}; }
#pragma unroll
    for (int s = 0; s < STREAMS; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[s][i] = cd{1e-3 * (tid + i + s), 2e-3 * (tid - i)}; acc[s][i] = 0; }
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int s = 0; s < STREAMS; ++s) {
            body(v[s], twa, twb, acc[s]);
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[s][i].x = v[s][i].x * 1e-3 + 1.0; v[s][i].y = v[s][i].y * 1e-3 - 1.0; }
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if ((tid & 63) == 0 && blockIdx.x == 0) { out[2 * (tid >> 6)] = t0; out[2 * (tid >> 6) + 1] = t1; }
    double s = 0;
#pragma unroll
    for (int q = 0; q < STREAMS; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[q][i];
    sink[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

template <int STREAMS>
int run(int threads, unsigned long long* d, double* s) {
    const int n = 2000;
    hipLaunchKernelGGL((k<STREAMS>), dim3(256), dim3(threads), 0, 0, d, s, n);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<STREAMS>), dim3(256), dim3(threads), 0, 0, d, s, n);
    CK(hipDeviceSynchronize());
    unsigned long long h[16];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const int lastw = threads / 64 - 1;
    printf("streams=%d waves/SIMD=%d: cycles per sub-series body: first wave %.0f, last wave %.0f (252 FP64 instr each incl. rescale)\n",
           STREAMS, threads / 256, (double)(h[1] - h[0]) / n / STREAMS, (double)(h[2 * lastw + 1] - h[2 * lastw]) / n / STREAMS);
    return 0;
}

int main() {
    unsigned long long* d; double* s;
    CK(hipMalloc(&d, 16 * 8)); CK(hipMalloc(&s, 256 * 512 * 8));
    run<1>(256, d, s); run<1>(512, d, s); run<2>(256, d, s); run<2>(512, d, s); run<3>(512, d, s);
    return 0;
}
