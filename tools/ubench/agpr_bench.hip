// agpr_bench.hip — cost of v_accvgpr_read/write, f64 VALU and ds_write_b128 for ONE wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k(unsigned long long* out, double* sink, int n) {
    __shared__ double4 lds[1024];
    unsigned acc = threadIdx.x;
    double f = threadIdx.x * 0.5, g = 1.0001, h = 0.25;
    unsigned long long t0, t1, t2, t3, t4, t5;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) { unsigned x; asm volatile("v_accvgpr_write_b32 a8, %1\n v_accvgpr_read_b32 %0, a8" : "=v"(x) : "v"(acc) : "a8"); acc = x + 1; }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) { unsigned x, y; asm volatile("v_accvgpr_read_b32 %0, a8\n v_accvgpr_read_b32 %1, a9" : "=v"(x), "=v"(y)); acc += x ^ y; }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2));
    double a0 = f, a1 = f + 1, a2 = f + 2, a3 = f + 3;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) { a0 = fma(a0, g, h); a1 = fma(a1, g, h); a2 = fma(a2, g, h); a3 = fma(a3, g, h); }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t3));
    double b0 = f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) b0 = fma(b0, g, h);
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t4));
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) lds[(threadIdx.x + 64 * k2) & 1023] = make_double4(a0, a1, a2, a3);
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t5));
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; }
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + acc + lds[threadIdx.x].x;
}
int main() {
    unsigned long long* d; double* s; CK(hipMalloc(&d, 64)); CK(hipMalloc(&s, 8 * 256 * 256));
    const int n = 1000;
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, s, n); CK(hipDeviceSynchronize());
    unsigned long long h[5]; CK(hipMemcpy(h, d, 40, hipMemcpyDeviceToHost));
    printf("accvgpr write+read dependent pair : %.1f cycles/pair\n", (double)h[0] / (n * 16));
    printf("accvgpr read x2 (independent)     : %.1f cycles/2 reads\n", (double)h[1] / (n * 16));
    printf("f64 fma, 4 independent chains      : %.1f cycles/fma\n", (double)h[2] / (n * 32));
    printf("f64 fma, 1 dependent chain         : %.1f cycles/fma\n", (double)h[3] / (n * 32));
    printf("ds_write_b128 x16 + wait (4 waves) : %.1f cycles/write\n", (double)h[4] / (n * 16));
    return 0;
}
