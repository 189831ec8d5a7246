// energy_bench.hip — what one FP64 FMA and one LDS byte cost in joules on MI355X, for the energy
// model of DESIGN.md section 6.0.  Three kernels, each launched back to back for ~3 s on random
// operands while tools/ubench/energy_bench.sh samples rocm-smi:
//   fma   : independent v_fma_f64 (8 accumulators x 8 multiplier pairs per lane), 2 waves per SIMD
//   lds   : each wave exchanges 8 KiB with itself: 8 ds_write_b128 + 8 ds_read_b128 per round (the
//           forward kernel's exchange), 2 waves per SIMD
//   mfma  : v_mfma_f64_16x16x4_f64 back to back, 8 independent accumulators, 2 waves per SIMD
//           (mfma4: 4 accumulators, 1 wave per SIMD) — the matrix pipe's FP64 rate and its power
//   copy  : 16 bytes per lane streamed from one 12 GB buffer to another (HBM read + write)
// Prints per kernel: ms per launch, the in-kernel clock (delta s_memtime / delta s_memrealtime), and
// operations per second; power is in the script's rocm-smi samples.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(512) k_fma(const double* __restrict__ in, double* __restrict__ out, int n,
                                             unsigned long long* clk) {
    const int tid = threadIdx.x;
    const size_t g = (size_t)blockIdx.x * 512 + tid;
    double a[8], x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = in[g * 24 + i], x[i] = in[g * 24 + 8 + i], y[i] = in[g * 24 + 16 + i];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(x[(i + b) & 7], y[b], a[i]);
        // keep the values bounded: one more FMA per accumulator pulls them back
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], -0.5, x[i]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[g] = s;
    if (tid == 0) clk[2 * blockIdx.x] = t1 - t0, clk[2 * blockIdx.x + 1] = r1 - r0;
}

typedef double d4 __attribute__((ext_vector_type(4)));
// v_mfma_f64_16x16x4_f64 back to back: NACC independent 16x16 accumulators per wave, operands from
// registers (random values scaled so the accumulators stay bounded).  2048 flop per instruction.
template <int NACC>
__global__ void __launch_bounds__(512) k_mfma(const double* __restrict__ in, double* __restrict__ out, int n,
                                              unsigned long long* clk) {
    const int tid = threadIdx.x;
    const size_t g = (size_t)blockIdx.x * 512 + tid;
    double a[4], b[4];
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = in[g * 24 + i] * 0.01, b[i] = in[g * 24 + 4 + i] * 0.01;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(i + k) & 3], b[k], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[g] = s;
    if (tid == 0) clk[2 * blockIdx.x] = t1 - t0, clk[2 * blockIdx.x + 1] = r1 - r0;
}

__global__ void __launch_bounds__(512) k_lds(const double* __restrict__ in, double* __restrict__ out, int n,
                                             unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t g = (size_t)blockIdx.x * 512 + tid;
    d2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = d2{in[g * 16 + 2 * i], in[g * 16 + 2 * i + 1]};
    d2* reg = reinterpret_cast<d2*>(smem) + wave * 512;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int a = 0; a < 8; ++a) reg[a * 64 + (lane ^ (8 * (a & 1)))] = v[a];
        __builtin_amdgcn_wave_barrier();
        const int hi = lane >> 3, lo = lane & 7, base = hi * 64 + (lo ^ (8 * (hi & 1)));
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) v[n1] = reg[base ^ (8 * n1)];
        __builtin_amdgcn_wave_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[g] = s;
    if (tid == 0) clk[2 * blockIdx.x] = t1 - t0, clk[2 * blockIdx.x + 1] = r1 - r0;
}

__global__ void __launch_bounds__(256) k_copy(const d2* __restrict__ in, d2* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = in[i];
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
    const char* what = argc > 1 ? argv[1] : "fma";
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nwg = prop.multiProcessorCount;  // one 512-thread workgroup per compute unit: 2 waves per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    if (!strcmp(what, "copy")) {
        const size_t n = (size_t)12 << 26;  // 12 GiB of 16-byte elements... 12 * 2^30 / 16
        d2 *a, *b;
        CK(hipMalloc(&a, n * 16));
        CK(hipMalloc(&b, n * 16));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)a, n * 2);
        CK(hipDeviceSynchronize());
        float total = 0, ms = 0;
        int launches = 0;
        while (total < seconds * 1e3) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_copy, dim3(nwg * 8), dim3(256), 0, 0, a, b, n);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            total += ms, launches += 20;
        }
        printf("copy: %.3f ms per launch of %.1f GB read + %.1f GB written = %.1f GB/s moved\n", total / launches, n * 16 / 1e9,
               n * 16 / 1e9, 2.0 * n * 16 / (total / launches * 1e-3) / 1e9);
        return 0;
    }
    const bool lds = !strcmp(what, "lds");
    const bool mfma = !strcmp(what, "mfma"), mfma4 = !strcmp(what, "mfma4");  // 8 / 4 accumulators; 8 or 4 waves per CU
    const int wgthreads = mfma4 ? 256 : 512;
    double *in, *out;
    unsigned long long* clk;
    const size_t nthr = (size_t)nwg * 512;
    CK(hipMalloc(&in, nthr * 24 * 8));
    CK(hipMalloc(&out, nthr * 8));
    CK(hipMalloc(&clk, (size_t)nwg * 16));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, in, nthr * 24);
    CK(hipDeviceSynchronize());
    const int n = lds ? 20000 : (mfma || mfma4) ? 4000 : 40000;
    if (lds) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    float total = 0, ms = 0;
    int launches = 0;
    while (total < seconds * 1e3) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 4; ++i) {
            if (lds) hipLaunchKernelGGL(k_lds, dim3(nwg), dim3(512), 65536, 0, in, out, n, clk);
            else if (mfma) hipLaunchKernelGGL(k_mfma<8>, dim3(nwg), dim3(512), 0, 0, in, out, n, clk);
            else if (mfma4) hipLaunchKernelGGL(k_mfma<4>, dim3(nwg), dim3(wgthreads), 0, 0, in, out, n, clk);
            else hipLaunchKernelGGL(k_fma, dim3(nwg), dim3(512), 0, 0, in, out, n, clk);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        total += ms, launches += 4;
    }
    unsigned long long h[2 * 256 * 2];
    CK(hipMemcpy(h, clk, (size_t)nwg * 16, hipMemcpyDeviceToHost));
    double c = 0, r = 0;
    for (int w = 0; w < nwg; ++w) c += (double)h[2 * w], r += (double)h[2 * w + 1];
    const double per = total / launches * 1e-3;
    if (mfma || mfma4) {
        const double nw = (double)nwg * wgthreads / 64, ninst = nw * n * 4 * (mfma ? 8 : 4);
        printf("%s: %.3f ms per launch, in-kernel clock %.0f MHz, %.3e v_mfma_f64_16x16x4 per second = %.1f TFLOP/s (%.1f cycles per instruction and SIMD)\n",
               what, per * 1e3, c / r * 100.0, ninst / per, ninst * 2048 / per / 1e12,
               per * (c / r * 1e8) / (ninst / (nwg * 4.0)));
    } else if (lds)
        printf("lds: %.3f ms per launch, in-kernel clock %.0f MHz, %.3e bytes stored + as many loaded per second (%.1f TB/s each way)\n",
               per * 1e3, c / r * 100.0, (double)nthr * n * 8 * 16 / per, (double)nthr * n * 8 * 16 / per / 1e12);
    else
        printf("fma: %.3f ms per launch, in-kernel clock %.0f MHz, %.3e FP64 FMAs per second (%.1f TFLOP/s)\n", per * 1e3,
               c / r * 100.0, (double)nthr * n * 72 / per, 2.0 * nthr * n * 72 / per / 1e12);
    return 0;
}
