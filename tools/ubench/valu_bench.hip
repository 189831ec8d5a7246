// valu_bench.hip — sustained issue rate of f64 / f32 / packed-f32 VALU instructions per SIMD at
// 1, 2 and 4 waves per SIMD (gfx950).  Prints shader cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

// KIND 0: v_fma_f64 (3 distinct operands, 8 accumulators x 8 multiplier pairs, like the direct tile)
// KIND 1: v_fma_f32   KIND 2: v_pk_fma_f32   KIND 3: v_add_f64   KIND 4: v_mul_f64
// KIND 5: v_fmac_f64 with x shared by 8 consecutive instructions (exactly the tile's pattern)
template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, double* sink, int n) {
    const double f = threadIdx.x * 0.5 + 1.0;
    double a[8], y[16], x[8];
    float af[8], yf[16], xf[8];
    v2f ap[8], yp[16], xp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = f + i; x[i] = 1.0 + 1e-9 * (i + f); af[i] = (float)a[i]; xf[i] = (float)x[i];
        ap[i] = v2f{af[i], af[i] + 1}; xp[i] = v2f{xf[i], xf[i]};
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { y[i] = 1e-9 * (i + 1) * f; yf[i] = (float)y[i]; yp[i] = v2f{yf[i], yf[i]}; }
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0 || KIND == 5) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(x[b]), "v"(y[i + b]));
                if (KIND == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(af[i]) : "v"(xf[b]), "v"(yf[i + b]));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ap[i]) : "v"(xp[b]), "v"(yp[i + b]));
                if (KIND == 3) asm volatile("v_add_f64 %0, %0, %2" : "+v"(a[i]) : "v"(x[b]), "v"(y[i + b]));
                if (KIND == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x[b]), "v"(y[i + b]));
                if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %2" : "+v"(ap[i]) : "v"(xp[b]), "v"(yp[i + b]));
                if (KIND == 8) asm volatile("v_fma_f64 %0, %0, 1.0, %2" : "+v"(a[i]) : "v"(x[b]), "v"(y[i + b]));
                if (KIND == 9) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x[b]), "v"(y[i + b]), "v"(y[(i + b + 3) & 15]));
                if (KIND == 10) asm volatile("v_fma_f64 %0, %0, %2, 0" : "+v"(a[i]) : "v"(x[b]), "v"(y[i + b]));
                if (KIND == 11) { if (i & 1) asm volatile("v_add_f64 %0, %0, %2" : "+v"(a[i]) : "v"(x[b]), "v"(y[i + b])); else asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(x[b]), "v"(y[i + b])); }
                if (KIND == 12) asm volatile("v_add_f64 %0, %1, -%2" : "=v"(a[i]) : "v"(x[b]), "v"(y[i + b]));
                if (KIND == 7) asm volatile("v_sub_f32 %0, %0, %2" : "+v"(af[i]) : "v"(xf[b]), "v"(yf[i + b]));
            }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + af[i] + ap[i].x + ap[i].y;
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
int run(const char* name, unsigned long long* d, double* s) {
    const int n = 2000;
    printf("%-34s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, d, s, n);
        CK(hipDeviceSynchronize());
        unsigned long long hh[32], lo = ~0ull, hi = 0;
        CK(hipMemcpy(hh, d, 16 * 4 * wps, hipMemcpyDeviceToHost));
        for (int w = 0; w < 4 * wps; ++w) { if (hh[2 * w] < lo) lo = hh[2 * w]; if (hh[2 * w + 1] > hi) hi = hh[2 * w + 1]; }
        const unsigned long long h = hi - lo;
        // one wave issues n*64 instructions; a SIMD hosts wps waves
        printf("  %d w/SIMD: %5.2f cyc/instr/SIMD", wps, (double)h / ((double)n * 64 * wps));
    }
    printf("\n");
    return 0;
}

template <int KIND>
int clock_under_load(const char* name, unsigned long long* d, double* s) {
    // long run on every CU at 4 waves/SIMD: shader clock = s_memtime ticks / hipEvent time
    const int n = 100000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, d, s, n);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hh[32], lo = ~0ull, hi = 0;
    CK(hipMemcpy(hh, d, 16 * 16, hipMemcpyDeviceToHost));
    for (int w = 0; w < 16; ++w) { if (hh[2 * w] < lo) lo = hh[2 * w]; if (hh[2 * w + 1] > hi) hi = hh[2 * w + 1]; }
    printf("%-20s %8.2f ms, %llu ticks -> %.0f MHz tick rate; %.2f ticks/instr/SIMD; %.2f ns/instr/SIMD\n", name, ms,
           hi - lo, (double)(hi - lo) / (ms * 1e3), (double)(hi - lo) / ((double)n * 64 * 4), ms * 1e6 / ((double)n * 64 * 4));
    return 0;
}

int main() {
    unsigned long long* d; double* s;
    CK(hipMalloc(&d, 1024)); CK(hipMalloc(&s, 8 * 1024 * 256));
    run<0>("v_fmac_f64 (8x8 tile pattern)", d, s);
    run<3>("v_add_f64", d, s);
    run<4>("v_mul_f64", d, s);
    run<8>("v_fma_f64 d, a, 1.0, b  (add via fma)", d, s);
    run<9>("v_fma_f64 d, a, b, c", d, s);
    run<10>("v_fma_f64 d, a, b, 0   (mul via fma)", d, s);
    run<11>("alternating v_add_f64 / v_fmac_f64", d, s);
    run<1>("v_fmac_f32", d, s);
    run<7>("v_sub_f32", d, s);
    run<2>("v_pk_fma_f32", d, s);
    run<6>("v_pk_add_f32", d, s);
    clock_under_load<0>("long v_fmac_f64", d, s);
    clock_under_load<1>("long v_fmac_f32", d, s);
    clock_under_load<2>("long v_pk_fma_f32", d, s);
    return 0;
}
