// lds_valu_bench.hip — do LDS stores (ds_write_b128) overlap with FP64 VALU work on gfx950?
// One 512-thread workgroup per CU (2 waves per SIMD).  Per iteration a wave issues NW
// ds_write_b128 (+ NR ds_read_b128) and NF v_fmac_f64, in one of several arrangements.
// Prints cycles per iteration (wave 0) for: VALU only, LDS only, both in every wave
// (sequential blocks / interleaved), and role-split waves (waves 0-3 VALU, waves 4-7 LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

// MODE 0: fma only; 1: lds only; 2: every wave: [NW writes][NR reads] then NF fma (blocks)
// 3: every wave interleaved: one write per NF/NW fma; 4: role split; 5: every wave: writes, fma, NO wait
template <int MODE, int NW, int NR, int NF>
__global__ void __launch_bounds__(512) k(unsigned long long* out, double* sink, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    double a[8], x[8], y[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = tid + i; x[i] = 1.0 + 1e-9 * (i + tid); }
#pragma unroll
    for (int i = 0; i < 16; ++i) y[i] = 1e-9 * (i + 1) * tid;
    d2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = d2{a[i], x[i]};
    // each wave owns 16 KiB: writes 8 x 1 KiB rows, conflict-free
    unsigned addr = (unsigned)(wave * 16384 + (tid & 63) * 16);
    const bool do_f = MODE == 0 || MODE == 2 || MODE == 3 || MODE == 5 || (MODE == 4 && wave < 4);
    const bool do_l = MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || (MODE == 4 && wave >= 4);
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < n; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v[w & 7]), "n"((w & 7) * 1024));
#pragma unroll
                for (int f = 0; f < NF / NW; ++f)
                    asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[f & 7]) : "v"(x[(f + w) & 7]), "v"(y[(f + w) & 15]));
            }
#pragma unroll
            for (int r = 0; r < NR; ++r)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r & 7]) : "v"(addr), "n"((r & 7) * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            if (do_l) {
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v[w & 7]), "n"((w & 7) * 1024));
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r & 7]) : "v"(addr), "n"((r & 7) * 1024));
            }
            if (do_f) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[f & 7]) : "v"(x[(f >> 3) & 7]), "v"(y[f & 15]));
            }
            if (do_l && MODE != 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if ((tid & 63) == 0 && blockIdx.x == 0) { out[2 * wave] = t0; out[2 * wave + 1] = t1; }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + v[i].x + v[i].y;
    sink[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

template <int MODE, int NW, int NR, int NF>
int run(const char* name, unsigned long long* d, double* s) {
    const int n = 2000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NW, NR, NF>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipLaunchKernelGGL((k<MODE, NW, NR, NF>), dim3(256), dim3(512), 131072, 0, d, s, n);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<MODE, NW, NR, NF>), dim3(256), dim3(512), 131072, 0, d, s, n);
    CK(hipDeviceSynchronize());
    unsigned long long h[16];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-44s NW=%2d NR=%2d NF=%3d  cycles/iter: wave0 %7.1f  wave4 %7.1f\n", name, NW, NR, NF,
           (double)(h[1] - h[0]) / n, (double)(h[9] - h[8]) / n);
    return 0;
}

int main() {
    unsigned long long* d;
    double* s;
    CK(hipMalloc(&d, 16 * 8));
    CK(hipMalloc(&s, 256 * 512 * 8));
    run<0, 8, 8, 84>("fma only", d, s);
    run<1, 8, 0, 84>("writes only", d, s);
    run<1, 8, 8, 84>("writes + reads only", d, s);
    run<2, 8, 0, 84>("all waves: writes, then fma, wait", d, s);
    run<2, 8, 8, 84>("all waves: writes+reads, then fma, wait", d, s);
    run<5, 8, 0, 84>("all waves: writes, fma, no wait", d, s);
    run<3, 8, 0, 80>("all waves: interleaved 1 write / 10 fma", d, s);
    run<3, 8, 8, 80>("all waves: interleaved + reads", d, s);
    run<4, 8, 0, 84>("split: waves0-3 fma, waves4-7 writes", d, s);
    run<4, 8, 8, 84>("split: waves0-3 fma, waves4-7 w+r", d, s);
    run<0, 16, 0, 168>("fma only", d, s);
    run<2, 16, 0, 168>("all waves: writes, then fma, wait", d, s);
    run<3, 16, 0, 160>("all waves: interleaved 1 write / 10 fma", d, s);
    run<2, 20, 0, 400>("S1-like: 20 writes then 400 fma", d, s);
    run<3, 20, 0, 400>("S1-like interleaved 1 write / 20 fma", d, s);
    run<0, 20, 0, 400>("fma only 400", d, s);
    return 0;
}
