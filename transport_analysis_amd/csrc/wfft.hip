// wfft.hip — launchers of the pair-major power-spectrum kernels (wfft.hpp) and the kernels that
// turn the summed spectrum into the lag-indexed sums (timeseries path of
// VelocityAutocorr._conclude_fft, /root/reference/transport_analysis/velocityautocorr.py:208-215).
#include "wfft.hpp"

#include <vector>

#include "ta_internal.hpp"

namespace ta {
namespace {

constexpr int kR0s[] = {1, 2, 4, 5, 8, 10, 16, 20};

template <int R0>
hipError_t launch_accum_r0(int nwg, hipStream_t st, const double* pm, long pitch, int T, long n_pairs,
                           const cd* tw, double* accg) {
    using P = WPlan<R0>;
    auto kern = k_wfft_accum<P, false, WF_TOUCH_DEFAULT, WF_INTER_DEFAULT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(P::NT), P::kLds, st, pm, pitch, T, n_pairs, tw, accg, nullptr);
    return hipGetLastError();
}

// pass-split kernel (one pass per workgroup, couples b / b + 8): grid a multiple of 16
template <int R0>
hipError_t launch_split_r0(int nwg, hipStream_t st, const double* pm, long pitch, int T, long n_pairs,
                           const cd* tw, double* accg) {
    using P = WPlan<R0>;
    auto kern = k_wsplit_accum<P, false, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(P::NT), P::kLds, st, pm, pitch, T, n_pairs, tw, accg, nullptr, 0);
    return hipGetLastError();
}

// by-particle, two kernels: the pass-split forward kernel leaves every atom's power spectrum in
// `spec` ([atom][2][R0][8][64] doubles), the inverse kernel turns each into the atom's lags
template <int R0>
hipError_t launch_bp2_r0(int nwg_fwd, int nwg_inv, hipStream_t st, const double* pm, long pitch, int T,
                         long n_atoms, int D, const cd* tw, double* spec, double* out, long ld, int pf) {
    using P = WPlan<R0>;
    constexpr int NSA = (R0 + P::NW - 1) / P::NW;
    auto fwd = k_wsplit_accum<P, false, true, true>;
    auto inv = pf <= 0 ? k_wbp_inverse<P, 0> : pf == 1 || NSA == 1 ? k_wbp_inverse<P, 1>
               : pf == 2 || NSA == 2 ? k_wbp_inverse<P, (NSA < 2 ? NSA : 2)> : k_wbp_inverse<P, NSA>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fwd),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(inv), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)P::kLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fwd, dim3(nwg_fwd), dim3(P::NT), P::kLds, st, pm, pitch, T, n_atoms, tw, spec, nullptr, D);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(inv, dim3(nwg_inv), dim3(P::NT), P::kLds, st, spec, T, n_atoms, tw, out, ld);
    return hipGetLastError();
}

template <int R0>
hipError_t launch_bp_r0(int nwg, hipStream_t st, const double* pm, long pitch, int T, long n_atoms, int D,
                        const cd* tw, double* out, long ld) {
    using P = WPlan<R0>;
    auto kern = k_wbp<P, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(P::NT), P::kLds, st, pm, pitch, T, n_atoms, D, tw, out, ld, nullptr);
    return hipGetLastError();
}

template <int R0>
int max_wg_r0() {
    using P = WPlan<R0>;
    auto kern = k_wfft_accum<P, false, WF_TOUCH_DEFAULT, WF_INTER_DEFAULT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)P::kLds);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, P::NT, P::kLds) != hipSuccess || n < 1) n = 1;
    return n;
}

// spec[k] = sum over workgroups of their natural-order accumulator blocks, k < L2 = 2M
__global__ void __launch_bounds__(256)
    k_wf_sum(const double* __restrict__ partial, int n_parts, int L2, double* __restrict__ spec) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= L2) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int w = 0;
    for (; w + 3 < n_parts; w += 4) {
        s0 += partial[(long)w * L2 + k];
        s1 += partial[(long)(w + 1) * L2 + k];
        s2 += partial[(long)(w + 2) * L2 + k];
        s3 += partial[(long)(w + 3) * L2 + k];
    }
    for (; w < n_parts; ++w) s0 += partial[(long)w * L2 + k];
    spec[k] = (s0 + s1) + (s2 + s3);
}

__global__ void __launch_bounds__(256)
    k_wf_sum_perm(const double* __restrict__ partial, int n_parts, int M, const int* __restrict__ perm,
                  double* __restrict__ spec) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // pass*M + p
    if (i >= 2 * M) return;
    const int pass = i / M, p = i - pass * M;
    const long k = 2L * perm[p] + pass;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int w = 0;
    for (; w + 3 < n_parts; w += 4) {
        s0 += partial[(long)w * 2 * M + k];
        s1 += partial[(long)(w + 1) * 2 * M + k];
        s2 += partial[(long)(w + 2) * 2 * M + k];
        s3 += partial[(long)(w + 3) * 2 * M + k];
    }
    for (; w < n_parts; ++w) s0 += partial[(long)w * 2 * M + k];
    spec[i] = (s0 + s1) + (s2 + s3);
}

// The lag sums are the real part of the inverse transform of the (real) summed spectrum:
//   lagsum[n] = (1 / (2M (T - n))) * sum_{k < 2M} P[k] cos(pi k n / M),
// cosine even in k about M: fold P[k] + P[2M - k] (k < M) first, then one workgroup per lag,
// the table index (k n) mod 2M advanced by a fixed step per thread, fixed-order tree sum.
__global__ void k_wf_fold(double* __restrict__ spec, int L2) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > 0 && k < L2 / 2) spec[k] += spec[L2 - k];
}

__global__ void __launch_bounds__(256)
    k_wf_lags(const double* __restrict__ spec, const cd* __restrict__ tw2, int L2, int T,
              double* __restrict__ lagsum) {
    __shared__ double red[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    int idx = (int)(((long)tid * n) % L2);
    const int step = (int)((256L * n) % L2);
    double s = 0.0;
    for (int k = tid; k <= L2 / 2; k += 256) {
        s += spec[k] * tw2[idx].x;
        idx += step;
        if (idx >= L2) idx -= L2;
    }
    red[tid] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) red[tid] += red[tid + h];
        __syncthreads();
    }
    if (tid == 0) lagsum[n] = red[0] / ((double)L2 * (double)(T - n));  // L2 (T-n) < 2^53: exact
}

}  // namespace

bool wfft_choose(long n_frames, int* R0) {
    for (int r : kR0s)
        if ((long)r * 512 >= n_frames) {
            *R0 = r;
            return true;
        }
    return false;
}

size_t wfft_table_elems(int R0) { return wf_table_elems(R0); }

void wfft_fill_table(int R0, cd* a) { wf_fill_table(R0, a); }

int wfft_max_wg_per_cu(int R0) {
    switch (R0) {
        case 1: return 2;  // 64 KiB of LDS per 256-thread workgroup
        case 2: return max_wg_r0<2>();
        case 4: return max_wg_r0<4>();
        case 5: return max_wg_r0<5>();
        case 8: return max_wg_r0<8>();
        case 10: return max_wg_r0<10>();
        case 16: return max_wg_r0<16>();
        case 20: return max_wg_r0<20>();
    }
    return 1;
}

// R0 = 1 (n_frames <= 512): independent waves, 4 per workgroup; accg [4 nwg][1024]
hipError_t launch_w1_accum(int nwg, hipStream_t st, const double* pm, long pitch, int T, long n_pairs,
                           const cd* tw, double* accg) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_w1_accum),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)W1::kLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_w1_accum, dim3(nwg), dim3(W1::NT), W1::kLds, st, pm, pitch, T, n_pairs, tw, accg);
    return hipGetLastError();
}

hipError_t launch_w1_bp(int nwg, hipStream_t st, const double* pm, long pitch, int T, long n_atoms, int D,
                        const cd* tw, double* out, long ld) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_w1_bp),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)W1::kLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_w1_bp, dim3(nwg), dim3(W1::NT), W1::kLds, st, pm, pitch, T, n_atoms, D, tw, out, ld);
    return hipGetLastError();
}

hipError_t launch_wfft_accum(int R0, int nwg, hipStream_t st, const double* pm, long pitch, int T,
                             long n_pairs, const cd* tw, double* accg) {
    switch (R0) {
        case 1: return launch_w1_accum(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 2: return launch_accum_r0<2>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 4: return launch_accum_r0<4>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 5: return launch_accum_r0<5>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 8: return launch_accum_r0<8>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 10: return launch_accum_r0<10>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 16: return launch_accum_r0<16>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 20: return launch_accum_r0<20>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_wfft_sum_perm(const double* partial, int n_parts, int M, const int* perm, double* spec,
                                hipStream_t st) {
    hipLaunchKernelGGL(k_wf_sum_perm, dim3((2 * M + 255) / 256), dim3(256), 0, st, partial, n_parts, M, perm, spec);
    return hipGetLastError();
}

hipError_t launch_wfft_by_particle(int R0, int nwg, hipStream_t st, const double* pm, long pitch, int T,
                                   long n_atoms, int D, const cd* tw, double* out, long ld) {
    switch (R0) {
        case 1: return launch_w1_bp(nwg, st, pm, pitch, T, n_atoms, D, tw, out, ld);
        case 2: return launch_bp_r0<2>(nwg, st, pm, pitch, T, n_atoms, D, tw, out, ld);
        case 4: return launch_bp_r0<4>(nwg, st, pm, pitch, T, n_atoms, D, tw, out, ld);
        case 5: return launch_bp_r0<5>(nwg, st, pm, pitch, T, n_atoms, D, tw, out, ld);
        case 8: return launch_bp_r0<8>(nwg, st, pm, pitch, T, n_atoms, D, tw, out, ld);
        case 10: return launch_bp_r0<10>(nwg, st, pm, pitch, T, n_atoms, D, tw, out, ld);
        case 16: return launch_bp_r0<16>(nwg, st, pm, pitch, T, n_atoms, D, tw, out, ld);
        case 20: return launch_bp_r0<20>(nwg, st, pm, pitch, T, n_atoms, D, tw, out, ld);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_wfft_split(int R0, int nwg, hipStream_t st, const double* pm, long pitch, int T,
                             long n_pairs, const cd* tw, double* accg) {
    if (nwg < 16 || nwg % 16) return hipErrorInvalidValue;
    switch (R0) {
        case 2: return launch_split_r0<2>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 4: return launch_split_r0<4>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 5: return launch_split_r0<5>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 8: return launch_split_r0<8>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 10: return launch_split_r0<10>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 16: return launch_split_r0<16>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
        case 20: return launch_split_r0<20>(nwg, st, pm, pitch, T, n_pairs, tw, accg);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_wfft_by_particle2(int R0, int nwg_fwd, int nwg_inv, hipStream_t st, const double* pm,
                                    long pitch, int T, long n_atoms, int D, const cd* tw, double* spec,
                                    double* out, long ld, int pf) {
    if (nwg_fwd < 16 || nwg_fwd % 16 || nwg_inv < 1) return hipErrorInvalidValue;
    switch (R0) {
        case 2: return launch_bp2_r0<2>(nwg_fwd, nwg_inv, st, pm, pitch, T, n_atoms, D, tw, spec, out, ld, pf);
        case 4: return launch_bp2_r0<4>(nwg_fwd, nwg_inv, st, pm, pitch, T, n_atoms, D, tw, spec, out, ld, pf);
        case 5: return launch_bp2_r0<5>(nwg_fwd, nwg_inv, st, pm, pitch, T, n_atoms, D, tw, spec, out, ld, pf);
        case 8: return launch_bp2_r0<8>(nwg_fwd, nwg_inv, st, pm, pitch, T, n_atoms, D, tw, spec, out, ld, pf);
        case 10: return launch_bp2_r0<10>(nwg_fwd, nwg_inv, st, pm, pitch, T, n_atoms, D, tw, spec, out, ld, pf);
        case 16: return launch_bp2_r0<16>(nwg_fwd, nwg_inv, st, pm, pitch, T, n_atoms, D, tw, spec, out, ld, pf);
        case 20: return launch_bp2_r0<20>(nwg_fwd, nwg_inv, st, pm, pitch, T, n_atoms, D, tw, spec, out, ld, pf);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_wfft_finish(int R0, const double* partial, int n_parts, const cd* tw, int T,
                              double* spec, double* lagsum, hipStream_t st) {
    const int L2 = 2 * R0 * 512;
    hipLaunchKernelGGL(k_wf_sum, dim3((L2 + 255) / 256), dim3(256), 0, st, partial, n_parts, L2, spec);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_wf_fold, dim3((L2 / 2 + 255) / 256), dim3(256), 0, st, spec, L2);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(k_wf_lags, dim3(T), dim3(256), 0, st, spec, tw, L2, T, lagsum);
    return hipGetLastError();
}

}  // namespace ta
