// direct.hip — launchers for the all-lag direct correlators and small helpers.
#include "direct_kernels.hpp"
#include "fft_kernels.hpp"
#include "ta_internal.hpp"

namespace ta {

// ---- cross-workgroup sum of the partial spectra -------------------------------
// spec[i] = sum_w partial[w][i],  i < 2*M (pass A then pass B)
__global__ void k_sum_partials(const double* __restrict__ partial, int n_parts, long n,
                               double* __restrict__ spec) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int w = 0; w < n_parts; ++w) s += partial[(long)w * n + i];
    spec[i] = s;
}


constexpr int kL = 8;

int direct_chunk() { return kL; }

size_t direct_lds_bytes(int T) {
    const int nchunks = (T + kL - 1) / kL;
    return (size_t)(nchunks + 3) * (kL + 2) * sizeof(double);
}

template <int MODE>
static hipError_t launch_mode(const double* vel, const double* pos, const double* masses,
                              long ld_row, int T, long n_atoms, int D, double scale,
                              double* by_particle, long ld_bp, double* ts_partial, int nwg, int nt,
                              size_t lds_bytes, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_direct<MODE, kL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_direct<MODE, kL>), dim3(nwg), dim3(nt), lds_bytes, st, vel, pos, masses,
                       ld_row, T, n_atoms, D, scale, by_particle, ld_bp, ts_partial);
    return hipGetLastError();
}

hipError_t launch_direct(int mode, const double* vel, const double* pos, const double* masses,
                         long ld_row, int T, long n_atoms, int D, double scale,
                         double* by_particle, long ld_bp, double* ts_partial, int nwg, int nt,
                         size_t lds_bytes, hipStream_t st) {
    if (mode == MODE_VACF)
        return launch_mode<MODE_VACF>(vel, pos, masses, ld_row, T, n_atoms, D, scale, by_particle,
                                      ld_bp, ts_partial, nwg, nt, lds_bytes, st);
    return launch_mode<MODE_HELFAND>(vel, pos, masses, ld_row, T, n_atoms, D, scale, by_particle,
                                     ld_bp, ts_partial, nwg, nt, lds_bytes, st);
}

int direct_max_wg_per_cu(int mode, int nt, size_t lds_bytes) {
    int n = 0;
    hipError_t e;
    if (mode == MODE_VACF) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_direct<MODE_VACF, kL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_direct<MODE_VACF, kL>, nt, lds_bytes);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_direct<MODE_HELFAND, kL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_direct<MODE_HELFAND, kL>, nt, lds_bytes);
    }
    if (e != hipSuccess || n < 1) n = 1;
    return n;
}

hipError_t launch_sum_partials(const double* partial, int n_parts, long n, double* out,
                               hipStream_t st) {
    const int nt = 256;
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((n + nt - 1) / nt)), dim3(nt), 0, st,
                       partial, n_parts, n, out);
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
