// wfft.hip — launchers of the pair-major power-spectrum kernels (wfft.hpp) and the kernels that
// turn the summed spectrum into the lag-indexed sums (timeseries path of
// VelocityAutocorr._conclude_fft, /root/reference/transport_analysis/velocityautocorr.py:208-215).
#include "wfft.hpp"

#include <algorithm>
#include <atomic>
#include <vector>

#include "ta_internal.hpp"

namespace ta {
namespace {

constexpr int kR0s[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20};
constexpr int kOuter[] = {1, 2, 3, 4, 5, 8, 16};

// the dynamic LDS limit of a kernel is set once per (kernel, device), not on every launch.
// Contexts of several devices may launch from several host threads: the flags are atomics
// (setting the attribute twice is harmless, skipping it is not), and a device index outside the
// cache takes the uncached path every time.
constexpr int kMaxDev = 16;
using DevFlag = std::atomic<bool>;
using DevCount = std::atomic<int>;
inline int cur_dev() {  // -1: not cacheable
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return -1;
    return d < 0 || d >= kMaxDev ? -1 : d;
}
template <class K>
hipError_t set_lds(K kern, size_t bytes, DevFlag* done /* [kMaxDev], one array per kernel */) {
    const int d = cur_dev();
    if (d >= 0 && done[d].load(std::memory_order_acquire)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && d >= 0) done[d].store(true, std::memory_order_release);
    return e;
}

template <int R0>
hipError_t launch_forward_r0(int R, bool byp, bool src_f32, int nwg, hipStream_t st, const double* pm, long pitch, int T,
                             long n_units, int D, const cd* tw, double* accg) {
    using P = WPlan<R0>;
    if (src_f32 && R > 1) return hipErrorInvalidValue;  // float32 slabs: no outer radix (the caller widens)
    auto kern = src_f32 ? (byp ? k_wsplit_accum<P, true, false, false, true> : k_wsplit_accum<P, false, false, false, true>)
                : byp   ? (R > 1 ? k_wsplit_accum<P, true, true> : k_wsplit_accum<P, true, false>)
                        : (R > 1 ? k_wsplit_accum<P, false, true> : k_wsplit_accum<P, false, false>);
    static DevFlag done[6][kMaxDev];
    hipError_t e = set_lds(kern, P::kLds, done[src_f32 ? 4 + (byp ? 1 : 0) : (byp ? 2 : 0) + (R > 1 ? 1 : 0)]);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(P::NT), P::kLds, st, pm, pitch, T, n_units, tw, accg, D, R, nullptr);
    return hipGetLastError();
}

// diagnostic build of the lag-sum forward kernel (R = 1) with in-kernel s_memtime / s_memrealtime
// stamps (ta_clock_probe): 16 unsigned 64-bit slots per workgroup
template <int R0>
hipError_t launch_forward_stamp_r0(int nwg, hipStream_t st, const double* pm, long pitch, int T, long n_units,
                                   const cd* tw, double* accg, unsigned long long* stamps) {
    using P = WPlan<R0>;
    auto kern = k_wsplit_accum<P, false, false, true>;
    static DevFlag done[kMaxDev];
    hipError_t e = set_lds(kern, P::kLds, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(P::NT), P::kLds, st, pm, pitch, T, n_units, tw, accg, 1, 1, stamps);
    return hipGetLastError();
}

template <int R0>
hipError_t launch_inverse_r0(int R, int nwg, hipStream_t st, const double* spec, int T, long n_items, const cd* tw,
                             double* out, long ld, int pf) {
    using P = WPlan<R0>;
    constexpr int NSA = P::NS1;
    const int variant = R > 1 ? 0 : pf <= 0 ? 1 : pf == 1 ? 2 : pf == 2 ? 3 : 4;
    auto kern = R > 1    ? k_winverse<P, true, 0>
                : pf <= 0 ? k_winverse<P, false, 0>
                : pf == 1 ? k_winverse<P, false, 1>
                : pf == 2 ? k_winverse<P, false, (NSA < 2 ? NSA : 2)>
                          : k_winverse<P, false, NSA>;
    static DevFlag done[5][kMaxDev];
    hipError_t e = set_lds(kern, P::kLds, done[variant]);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(P::NT), P::kLds, st, spec, T, n_items, tw, out, ld, R);
    return hipGetLastError();
}

// resident workgroups per compute unit of plan R0: the smallest over the forward kernel's four
// variants (the grid of each must be fully resident: a tuple's workgroups wait for nobody, but
// they share rows through the L2 only while they run together); asked once per device
template <int R0>
int max_wg_r0() {
    using P = WPlan<R0>;
    static DevCount cached[kMaxDev];
    const int d = cur_dev();
    if (d >= 0) {
        const int c = cached[d].load(std::memory_order_acquire);
        if (c > 0) return c;
    }
    int best = 1 << 30;
    auto ask = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)P::kLds);
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, P::NT, P::kLds) != hipSuccess || n < 1) n = 1;
        best = std::min(best, n);
    };
    ask(k_wsplit_accum<P, false, false>);
    ask(k_wsplit_accum<P, false, true>);
    ask(k_wsplit_accum<P, true, false>);
    ask(k_wsplit_accum<P, true, true>);
    ask(k_wsplit_accum<P, false, false, false, true>);
    ask(k_wsplit_accum<P, true, false, false, true>);
    if (d >= 0) cached[d].store(best, std::memory_order_release);
    return best;
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

// Up to 10240 frames: the smallest on-chip length R0 * 512 >= n_frames.  Beyond: an outer radix R
// in front of one of the five largest on-chip plans; a pass then forms its rows from R strided
// rows each, work that grows like R^2 per pair next to the transforms' R log: the plan with the
// smallest M' (1 + 0.15 R) wins (measured at 12 GB: 25000 frames as 5 x 5120: 12.9 ms, 30000 frames
// as 3 x 10240: 8.1 ms).
bool wfft_choose(long n_frames, int* R0, int* R) {
    for (int r : kR0s)
        if ((long)r * 512 >= n_frames) {
            *R0 = r, *R = 1;
            return true;
        }
    double best = 0.0;
    for (int ro : kOuter)
        for (int r : {12, 14, 16, 18, 20}) {
            const long m = (long)ro * r * 512;
            if (ro == 1 || m < n_frames) continue;
            const double cost = (double)m * (1.0 + 0.15 * ro);
            if (best == 0.0 || cost < best) best = cost, *R0 = r, *R = ro;
        }
    return best != 0.0;
}

size_t wfft_table_elems(int R0, int R) { return wf_table_elems(R0, R); }

void wfft_fill_table(int R0, int R, cd* a) { wf_fill_table(R0, R, a); }

// threads per workgroup of the forward / inverse kernels of plan R0 (512-point kernels: a wave)
int wfft_threads(int R0) {
    switch (R0) {
        case 2: return WPlan<2>::NT;
        case 3: return WPlan<3>::NT;
        case 4: return WPlan<4>::NT;
        case 5: return WPlan<5>::NT;
        case 6: return WPlan<6>::NT;
        case 7: return WPlan<7>::NT;
        case 8: return WPlan<8>::NT;
        case 9: return WPlan<9>::NT;
        case 10: return WPlan<10>::NT;
        case 12: return WPlan<12>::NT;
        case 14: return WPlan<14>::NT;
        case 16: return WPlan<16>::NT;
        case 18: return WPlan<18>::NT;
        case 20: return WPlan<20>::NT;
    }
    return W1::NT;  // R0 = 1: four independent waves per workgroup
}

int wfft_max_wg_per_cu(int R0) {
    switch (R0) {
        case 1: return 2;  // 64 KiB of LDS per 256-thread workgroup
        case 2: return max_wg_r0<2>();
        case 3: return max_wg_r0<3>();
        case 4: return max_wg_r0<4>();
        case 5: return max_wg_r0<5>();
        case 6: return max_wg_r0<6>();
        case 7: return max_wg_r0<7>();
        case 8: return max_wg_r0<8>();
        case 9: return max_wg_r0<9>();
        case 10: return max_wg_r0<10>();
        case 12: return max_wg_r0<12>();
        case 14: return max_wg_r0<14>();
        case 16: return max_wg_r0<16>();
        case 18: return max_wg_r0<18>();
        case 20: return max_wg_r0<20>();
    }
    return 1;
}

// R0 = 1 (n_frames <= 512): independent waves, 4 per workgroup; accg [4 nwg][1024]
hipError_t launch_w1_accum(int nwg, hipStream_t st, const double* pm, long pitch, int T, long n_pairs,
                           const cd* tw, double* accg) {
    // n_frames <= 256: pass A's 512 bins are pad enough, pass B is skipped; <= 128 / 64 / 32: 2 / 4 / 8 pairs share a transform
    // (wfft.hpp: w1_two_passes, k_w1_accum)
    static DevFlag done[5][kMaxDev];
    auto go = [&](auto kern, DevFlag* flag) {
        hipError_t e = set_lds(kern, W1::kLds, flag);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(W1::NT), W1::kLds, st, pm, pitch, T, n_pairs, tw, accg);
        return hipGetLastError();
    };
    if (T <= 32) return go(k_w1_accum<true, 8>, done[4]);
    if (T <= 64) return go(k_w1_accum<true, 4>, done[3]);
    if (T <= 128) return go(k_w1_accum<true, 2>, done[2]);
    if (T <= 256) return go(k_w1_accum<true, 1>, done[1]);
    return go(k_w1_accum<false, 1>, done[0]);
}

hipError_t launch_w1_bp(int nwg, hipStream_t st, const double* pm, long pitch, int T, long n_atoms, int D,
                        const cd* tw, double* out, long ld) {
    static DevFlag done[kMaxDev], done1[kMaxDev];
    hipError_t e = T <= 256 ? set_lds(k_w1_bp<true>, W1::kLds, done1) : set_lds(k_w1_bp<false>, W1::kLds, done);
    if (e != hipSuccess) return e;
    if (T <= 256) hipLaunchKernelGGL(k_w1_bp<true>, dim3(nwg), dim3(W1::NT), W1::kLds, st, pm, pitch, T, n_atoms, D, tw, out, ld);
    else hipLaunchKernelGGL(k_w1_bp<false>, dim3(nwg), dim3(W1::NT), W1::kLds, st, pm, pitch, T, n_atoms, D, tw, out, ld);
    return hipGetLastError();
}

// (src_f32: the slab holds float32 elements, 8-byte rows; plans without an outer radix only)
// Forward kernel (R0 > 1): nwg a multiple of 16 R.  Lag-sum mode (by_particle false): n_units
// column pairs, accg [nwg / 2R][L] partial spectra; by-particle mode: n_units atoms of D columns,
// accg [n_units][L].  L = 2 R R0 512 doubles per spectrum.
hipError_t launch_wfft_forward(int R0, int R, bool by_particle, bool src_f32, int nwg, hipStream_t st, const double* pm,
                               long pitch, int T, long n_units, int D, const cd* tw, double* accg) {
    if (R < 1 || nwg < 16 * R || nwg % (16 * R)) return hipErrorInvalidValue;
    switch (R0) {
        case 2: return launch_forward_r0<2>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 3: return launch_forward_r0<3>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 4: return launch_forward_r0<4>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 5: return launch_forward_r0<5>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 6: return launch_forward_r0<6>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 7: return launch_forward_r0<7>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 8: return launch_forward_r0<8>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 9: return launch_forward_r0<9>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 10: return launch_forward_r0<10>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 12: return launch_forward_r0<12>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 14: return launch_forward_r0<14>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 16: return launch_forward_r0<16>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 18: return launch_forward_r0<18>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
        case 20: return launch_forward_r0<20>(R, by_particle, src_f32, nwg, st, pm, pitch, T, n_units, D, tw, accg);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_wfft_forward_stamp(int R0, int nwg, hipStream_t st, const double* pm, long pitch, int T,
                                     long n_units, const cd* tw, double* accg, unsigned long long* stamps) {
    if (nwg < 16 || nwg % 16) return hipErrorInvalidValue;
    switch (R0) {
        case 8: return launch_forward_stamp_r0<8>(nwg, st, pm, pitch, T, n_units, tw, accg, stamps);
        case 10: return launch_forward_stamp_r0<10>(nwg, st, pm, pitch, T, n_units, tw, accg, stamps);
        case 12: return launch_forward_stamp_r0<12>(nwg, st, pm, pitch, T, n_units, tw, accg, stamps);
        case 16: return launch_forward_stamp_r0<16>(nwg, st, pm, pitch, T, n_units, tw, accg, stamps);
        case 20: return launch_forward_stamp_r0<20>(nwg, st, pm, pitch, T, n_units, tw, accg, stamps);
    }
    return hipErrorInvalidValue;
}

// Inverse kernel (R0 > 1): the lag values of n_items spectra, out[item * ld + lag].
hipError_t launch_wfft_inverse(int R0, int R, int nwg, hipStream_t st, const double* spec, int T, long n_items,
                               const cd* tw, double* out, long ld, int prefetch) {
    if (R < 1 || nwg < 1) return hipErrorInvalidValue;
    switch (R0) {
        case 2: return launch_inverse_r0<2>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 3: return launch_inverse_r0<3>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 4: return launch_inverse_r0<4>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 5: return launch_inverse_r0<5>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 6: return launch_inverse_r0<6>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 7: return launch_inverse_r0<7>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 8: return launch_inverse_r0<8>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 9: return launch_inverse_r0<9>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 10: return launch_inverse_r0<10>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 12: return launch_inverse_r0<12>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 14: return launch_inverse_r0<14>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 16: return launch_inverse_r0<16>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 18: return launch_inverse_r0<18>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
        case 20: return launch_inverse_r0<20>(R, nwg, st, spec, T, n_items, tw, out, ld, prefetch);
    }
    return hipErrorInvalidValue;
}

// R0 = 1 lag sums: the 1024 summed bins -> lag sums by the cosine sum (T <= 512 lags)
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
