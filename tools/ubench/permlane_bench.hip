// permlane_bench.hip — issue cost of the cross-lane register moves a register-file transposition
// would be built from (gfx950): v_permlane32_swap, v_permlane16_swap, v_mov_b32_dpp (row_ror with a
// bank mask, quad_perm), v_cndmask_b32 with a DPP operand, next to v_fma_f64.  One 512-thread
// workgroup per CU (2 waves per SIMD), N independent instructions per iteration; prints cycles
// per wave-instruction as seen by wave 0 and by its SIMD partner (wave 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned long long* out, unsigned* sink, int n) {
    unsigned r[16];
    double d[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * (i + 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = 1.0 + threadIdx.x * 1e-9 * (i + 1);
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[2 * i]), "+v"(r[2 * i + 1]));
                if (MODE == 1) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[2 * i]), "+v"(r[2 * i + 1]));
                if (MODE == 2) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(r[2 * i]) : "v"(r[2 * i + 1]));
                if (MODE == 3) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[2 * i]) : "v"(r[2 * i + 1]));
                if (MODE == 4) asm volatile("v_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r[2 * i]) : "v"(r[2 * i + 1]), "v"(r[2 * i]) : "vcc");
                if (MODE == 5) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 3) & 7]));
                if (MODE == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(r[2 * i]) : "v"(r[2 * i + 1]));
            }
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += r[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += (unsigned)d[i];
    sink[blockIdx.x * 512 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
int run(const char* name, int ncu) {
    unsigned long long* d_out;
    unsigned* d_sink;
    CK(hipMalloc(&d_out, ncu * 8 * 8));
    CK(hipMalloc(&d_sink, ncu * 512 * 4));
    const int n = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(512), 0, 0, d_out, d_sink, n);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(512), 0, 0, d_out, d_sink, n);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-44s cycles per wave-instruction: wave0 %.2f  wave4 %.2f\n", name, (double)h[0] / (n * 32.0), (double)h[4] / (n * 32.0));
    hipFree(d_out);
    hipFree(d_sink);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    run<5>("v_fma_f64", ncu);
    run<6>("v_mov_b32", ncu);
    run<0>("v_permlane32_swap_b32", ncu);
    run<1>("v_permlane16_swap_b32", ncu);
    run<2>("v_mov_b32_dpp row_ror:8 bank_mask:0xc", ncu);
    run<3>("v_mov_b32_dpp quad_perm", ncu);
    run<4>("v_cndmask_b32_dpp quad_perm", ncu);
    return 0;
}
