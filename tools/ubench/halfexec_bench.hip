// halfexec_bench.hip — does a wave64 instruction with only lanes 0..31 (or 0..15) enabled issue
// faster than with all 64?  (Balancing 2.5 butterfly rounds per wave as 2 full + 1 half-wave
// round only pays if it does.)  v_fmac_f64, ds_write_b128, ds_read_b128; one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(unsigned long long* out, double* sink, int n, int active) {
    extern __shared__ double4 lds[];
    const int lane = threadIdx.x & 63;
    const double f = threadIdx.x * 0.5 + 1.0;
    double a[8], y[16], x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = f + i; x[i] = 1.0 + 1e-9 * (i + f); }
#pragma unroll
    for (int i = 0; i < 16; ++i) y[i] = 1e-9 * (i + 1) * f;
    double2 v = make_double2(f, f + 1);
    double2* l2 = reinterpret_cast<double2*>(lds);
    unsigned long long t0 = 0, t1 = 0;
    __syncthreads();
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    if (lane < active) {
        for (int it = 0; it < n; ++it) {
            if (KIND == 0) {
#pragma unroll
                for (int b = 0; b < 8; ++b)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(x[b]), "v"(y[i + b]));
            } else if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 64; ++i) l2[threadIdx.x + 256 * (i & 15)] = v;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int i = 0; i < 64; ++i) {
                    double2 r = l2[threadIdx.x + 256 * (i & 15)];
                    asm volatile("" :: "v"(r.x), "v"(r.y));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (lane == 0 && blockIdx.x == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
    double s = v.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
int run(const char* name, unsigned long long* d, double* s) {
    const int n = 500;
    printf("%-18s", name);
    for (int active : {64, 48, 32, 16}) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 65536, 0, d, s, n, active);
        CK(hipDeviceSynchronize());
        unsigned long long hh[8], lo = ~0ull, hi = 0;
        CK(hipMemcpy(hh, d, 16 * 4, hipMemcpyDeviceToHost));
        for (int w = 0; w < 4; ++w) { if (hh[2 * w] < lo) lo = hh[2 * w]; if (hh[2 * w + 1] > hi) hi = hh[2 * w + 1]; }
        printf("  %2d lanes: %6.2f cyc/instr", active, (double)(hi - lo) / ((double)n * 64));
    }
    printf("\n");
    return 0;
}

int main() {
    unsigned long long* d; double* s;
    CK(hipMalloc(&d, 1024)); CK(hipMalloc(&s, sizeof(double) * 256 * 256));
    CK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    run<0>("v_fmac_f64", d, s);
    run<1>("ds_write_b128", d, s);
    run<2>("ds_read_b128", d, s);
    return 0;
}
