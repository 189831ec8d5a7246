// direct.hip — launchers for the all-lag direct correlators and small helpers.
#include "direct_kernels.hpp"
#include "fft_engine.hpp"
#include "ta_internal.hpp"

namespace ta {

// ---- sum of partial rows in a fixed order ------------------------------------
// out[i] = sum_w partial[w][i], i < n.  A workgroup takes 16 adjacent elements (128-byte row
// segments); thread (g, i) adds the rows w = g, g + 16, ... in two interleaved chains, then the
// 16 group sums are added in a fixed tree: the result does not depend on the launch.
__global__ void __launch_bounds__(256)
    k_sum_partials(const double* __restrict__ partial, int n_parts, long n, double* __restrict__ out) {
    __shared__ double red[16][17];
    const int tid = threadIdx.x, g = tid >> 4, e = tid & 15;
    const long i = (long)blockIdx.x * 16 + e;
    double s0 = 0.0, s1 = 0.0;
    if (i < n) {
        int w = g;
        for (; w + 16 < n_parts; w += 32) {
            s0 += partial[(long)w * n + i];
            s1 += partial[(long)(w + 16) * n + i];
        }
        if (w < n_parts) s0 += partial[(long)w * n + i];
    }
    red[g][e] = s0 + s1;
    __syncthreads();
    for (int h = 8; h > 0; h >>= 1) {
        if (g < h) red[g][e] += red[g + h][e];
        __syncthreads();
    }
    if (g == 0 && i < n) out[i] = red[0][e];
}

// compiled chunk sizes (lags per chunk); the launcher picks the one that fills the CU best
#define TA_DIRECT_CHUNKS(X) X(8) X(10)

size_t direct_lds_bytes(int T, bool f32, int L) {
    const int nchunks = (T + L - 1) / L;
    return (size_t)(nchunks + 3) * group_stride_dwords(L, f32 ? 1 : 2) * 4;
}

// src_f32: float32 slabs (only with the float32 arithmetic path: f32 must be set too)
static const void* direct_kernel(int mode, bool f32, int L, bool gs, bool src_f32) {
    if (src_f32 && !f32) return nullptr;
#define TA_K(M, LL, GS, R, S) reinterpret_cast<const void*>(k_direct<M, LL, GS, R, S>)
#define TA_PICK(M, LL)                                                                               \
    if (src_f32) return gs ? TA_K(M, LL, true, float, float) : TA_K(M, LL, false, float, float);      \
    if (f32) return gs ? TA_K(M, LL, true, float, double) : TA_K(M, LL, false, float, double);        \
    return gs ? TA_K(M, LL, true, double, double) : TA_K(M, LL, false, double, double);
#define TA_X(LL)                                  \
    if (L == LL) {                                \
        if (mode == MODE_VACF) {                  \
            TA_PICK(MODE_VACF, LL)                \
        }                                         \
        TA_PICK(MODE_HELFAND, LL)                 \
    }
    TA_DIRECT_CHUNKS(TA_X)
#undef TA_X
#undef TA_PICK
#undef TA_K
    return nullptr;
}

bool direct_chunk_supported(int L) {
#define TA_X(LL) if (L == LL) return true;
    TA_DIRECT_CHUNKS(TA_X)
#undef TA_X
    return false;
}

hipError_t launch_direct(int mode, bool f32, bool src_f32, int L, const void* vel, const void* pos,
                         const double* masses, long ld_row, int T, long n_atoms, int D,
                         double scale, double* by_particle, long ld_bp, double* ts_partial, int nwg,
                         int nt, size_t lds_bytes, void* stage_buf, int gnt, hipStream_t st) {
    const bool gs = stage_buf != nullptr;  // long trajectory: column staged in global memory
    const void* fn = direct_kernel(mode, f32, L, gs, src_f32);
    if (!fn) return hipErrorInvalidValue;
    if (!gs) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    void* args[] = {&vel, &pos, &masses, &ld_row, &T, &n_atoms, &D, &scale, &by_particle,
                    &ld_bp, &ts_partial, &stage_buf, &gnt};
    return hipLaunchKernel(fn, dim3(nwg), dim3(nt), args, gs ? 0 : lds_bytes, st);
}

int direct_max_wg_per_cu(int mode, bool f32, int L, int nt, size_t lds_bytes, bool global_stage) {
    const void* fn = direct_kernel(mode, f32, L, global_stage, false);
    if (!fn) return 1;
    int n = 0;
    if (!global_stage)
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, nt,
                                                                global_stage ? 0 : lds_bytes);
    return (e != hipSuccess || n < 1) ? 1 : n;
}

hipError_t launch_sum_partials(const double* partial, int n_parts, long n, double* out,
                               hipStream_t st) {
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, partial, n_parts, n, out);
    return hipGetLastError();
}

// out[n] = sum_a bp[n*ld + a]: the lag sums of a by-particle result (one workgroup per lag).
__global__ void __launch_bounds__(256) k_row_sums(const double* __restrict__ bp, long n_cols,
                                                  long ld, double* __restrict__ out) {
    __shared__ double part[4];
    const double* row = bp + (long)blockIdx.x * ld;
    double s0 = 0.0, s1 = 0.0;
    long a = threadIdx.x;
    for (; a + 256 < n_cols; a += 512) {
        s0 += row[a];
        s1 += row[a + 256];
    }
    if (a < n_cols) s0 += row[a];
    double s = s0 + s1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

hipError_t launch_row_sums(const double* bp, long n_rows, long n_cols, long ld, double* out,
                           hipStream_t st) {
    hipLaunchKernelGGL(k_row_sums, dim3((unsigned)n_rows), dim3(256), 0, st, bp, n_cols, ld, out);
    return hipGetLastError();
}

__global__ void k_widen_f32(const float* __restrict__ in, double* __restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (double)in[i];
}

hipError_t launch_widen_f32(const float* in, double* out, long n, hipStream_t st) {
    const int nt = 256;
    long blocks = (n + nt - 1) / nt;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_widen_f32, dim3((unsigned)blocks), dim3(nt), 0, st, in, out, n);
    return hipGetLastError();
}

}  // namespace ta
