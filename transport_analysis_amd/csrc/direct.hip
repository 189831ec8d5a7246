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


// Sum the per-workgroup accumulator blocks and restore the transform's bin order.
// Block layout per pass: [quad][tid][2 doubles]; double index d = m*R + q of a thread sits
// in quad d/2, component d%2.  spec[pass*M + u*R + q], u = tid + m*NT.
__global__ void k_sum_partials_perm(const double* __restrict__ partial, int n_parts, int M, int NT,
                                    int R, int K, int TASKS, double* __restrict__ spec) {
    const int quads = (K * R + 1) / 2;
    const long blk = (long)quads * 2 * NT;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // index within [2][blk]
    if (i >= 2 * blk) return;
    const int pass = (int)(i / blk);
    const long r = i - pass * blk;
    const int comp = (int)(r & 1);
    const long qt = r >> 1;  // quad*NT + tid
    const int tid = (int)(qt % NT);
    const int d = 2 * (int)(qt / NT) + comp;
    if (d >= K * R) return;
    const int m = d / R, q = d % R;
    const int u = tid + m * NT;
    if (u >= TASKS) return;
    double s = 0.0;
    for (int w = 0; w < n_parts; ++w) s += partial[(long)w * 2 * blk + i];
    spec[(long)pass * M + (long)u * R + q] = s;
}

hipError_t launch_sum_partials_perm(const double* partial, int n_parts, int M, int NT, int R, int K,
                                    int TASKS, double* out, hipStream_t st) {
    const int nt = 256;
    const long n = 2L * ((K * R + 1) / 2) * 2 * NT;
    hipLaunchKernelGGL(k_sum_partials_perm, dim3((unsigned)((n + nt - 1) / nt)), dim3(nt), 0, st,
                       partial, n_parts, M, NT, R, K, TASKS, out);
    return hipGetLastError();
}

constexpr int kL = 8;

int direct_chunk() { return kL; }

size_t direct_lds_bytes(int T, bool f32) {
    const int nchunks = (T + kL - 1) / kL;
    return f32 ? (size_t)(nchunks + 3) * group_stride<kL, float>() * sizeof(float)
               : (size_t)(nchunks + 3) * group_stride<kL, double>() * sizeof(double);
}

template <int MODE, typename Real>
static hipError_t launch_mode(const double* vel, const double* pos, const double* masses,
                              long ld_row, int T, long n_atoms, int D, double scale,
                              double* by_particle, long ld_bp, double* ts_partial, int nwg, int nt,
                              size_t lds_bytes, void* stage_buf, int gnt, hipStream_t st) {
    if (stage_buf) {  // long trajectory: column staged in global memory, no LDS
        hipLaunchKernelGGL((k_direct<MODE, kL, true, Real>), dim3(nwg), dim3(nt), 0, st, vel, pos,
                           masses, ld_row, T, n_atoms, D, scale, by_particle, ld_bp, ts_partial,
                           stage_buf, gnt);
        return hipGetLastError();
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_direct<MODE, kL, false, Real>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_direct<MODE, kL, false, Real>), dim3(nwg), dim3(nt), lds_bytes, st, vel,
                       pos, masses, ld_row, T, n_atoms, D, scale, by_particle, ld_bp, ts_partial,
                       (void*)nullptr, gnt);
    return hipGetLastError();
}

hipError_t launch_direct(int mode, bool f32, const double* vel, const double* pos,
                         const double* masses, long ld_row, int T, long n_atoms, int D,
                         double scale, double* by_particle, long ld_bp, double* ts_partial, int nwg,
                         int nt, size_t lds_bytes, void* stage_buf, int gnt, hipStream_t st) {
#define TA_GO(M, R)                                                                              \
    return launch_mode<M, R>(vel, pos, masses, ld_row, T, n_atoms, D, scale, by_particle, ld_bp, \
                             ts_partial, nwg, nt, lds_bytes, stage_buf, gnt, st)
    if (mode == MODE_VACF) {
        if (f32) TA_GO(MODE_VACF, float);
        TA_GO(MODE_VACF, double);
    }
    if (f32) TA_GO(MODE_HELFAND, float);
    TA_GO(MODE_HELFAND, double);
#undef TA_GO
}

template <int MODE, bool GS, typename Real>
static int occ(int nt, size_t lds_bytes) {
    int n = 0;
    if (!GS)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_direct<MODE, kL, GS, Real>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_direct<MODE, kL, GS, Real>, nt,
                                                                GS ? 0 : lds_bytes);
    return (e != hipSuccess || n < 1) ? 1 : n;
}

int direct_max_wg_per_cu(int mode, bool f32, int nt, size_t lds_bytes, bool global_stage) {
    const int key = (mode == MODE_VACF ? 0 : 4) + (f32 ? 2 : 0) + (global_stage ? 1 : 0);
    switch (key) {
        case 0: return occ<MODE_VACF, false, double>(nt, lds_bytes);
        case 1: return occ<MODE_VACF, true, double>(nt, lds_bytes);
        case 2: return occ<MODE_VACF, false, float>(nt, lds_bytes);
        case 3: return occ<MODE_VACF, true, float>(nt, lds_bytes);
        case 4: return occ<MODE_HELFAND, false, double>(nt, lds_bytes);
        case 5: return occ<MODE_HELFAND, true, double>(nt, lds_bytes);
        case 6: return occ<MODE_HELFAND, false, float>(nt, lds_bytes);
        default: return occ<MODE_HELFAND, true, float>(nt, lds_bytes);
    }
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
