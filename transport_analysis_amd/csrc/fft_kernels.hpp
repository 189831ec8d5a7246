// fft_kernels.hpp — the on-chip plan kernels that are still in use: the ONE inverse transform
// per launch that turns a summed power spectrum into lag sums (k_fft_finalize; replaces the
// per-atom inverse FFTs of tidynamics.acf, /root/reference/transport_analysis/velocityautocorr.py:
// 208-215) and the stage helpers the outer-radix path (fft_long.hip) builds on.  The power-spectrum
// accumulation itself lives in wfft.hpp (pair-major slabs, wave-local transforms); round 1's
// gather kernel with its hand-assigned AGPR slots is gone.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "fft_engine.hpp"

namespace ta {

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    [&]<int... I>(std::integer_sequence<int, I...>) {
        (f(std::integral_constant<int, I>{}), ...);
    }(std::make_integer_sequence<int, N>{});
}

// Same, accumulating into the workgroup's accumulator block in global memory (L2/MALL
// resident, [quad][thread] x double2): a butterfly's R accumulators are loaded one butterfly
// ahead and stored right after the update, so only 2*R of them are ever in registers.
template <class P, class Hook>
__device__ __forceinline__ void last_stage_acc_global(const cd* __restrict__ lds,
                                                      double* __restrict__ blk, int tid,
                                                      Hook&& after_task) {
    using SI = StageInfo<P, P::S - 1>;
    static_assert(SI::R % 2 == 0, "accumulators move as double2");
    constexpr int Q = SI::R / 2;  // double2 per butterfly
    // global address space kept explicit: a laundered generic pointer would turn these into
    // flat loads/stores, whose waits drain every load in flight
    typedef tw_d2 __attribute__((address_space(1)))* gptr;
    tw_d2 buf[2][Q];
    auto row = [&](int quad) {
        double* r = blk + (long)quad * 2 * P::NT;
        asm volatile("" : "+s"(r));  // uniform row base: SGPR base + one shared VGPR offset
        return (gptr)r;
    };
#pragma unroll
    for (int h = 0; h < Q; ++h) buf[0][h] = row(h)[tid];
    static_for<SI::K>([&](auto mm) {
        constexpr int m = decltype(mm)::value;
        const int u = tid + m * P::NT;
        if constexpr (m + 1 < SI::K) {
#pragma unroll
            for (int h = 0; h < Q; ++h) buf[(m + 1) & 1][h] = row((m + 1) * Q + h)[tid];
        }
        if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
            const int sb = sw(u * SI::R);
            cd v[SI::R];
#pragma unroll
            for (int j = 0; j < SI::R; ++j) v[j] = lds[sw_off(sb, j)];
            Dft<SI::R>::run(v);
#pragma unroll
            for (int h = 0; h < Q; ++h) {
                tw_d2 a = buf[m & 1][h];
                a.x += norm2(v[2 * h]);
                a.y += norm2(v[2 * h + 1]);
                row(m * Q + h)[tid] = a;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        after_task(m);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// The workgroup's accumulator blocks in global memory are laid out [quad][thread] x 2 doubles,
// so a wave's accesses are contiguous; a thread only ever touches its own slots.
template <class P>
constexpr int acc_quads() {
    return (StageInfo<P, P::S - 1>::K * StageInfo<P, P::S - 1>::R * 2 + 3) / 4;
}
template <class P, int s>
__device__ __forceinline__ cd stage_seed(const cd* __restrict__ tw2, int tid) {
    using SI = StageInfo<P, s>;
    if constexpr (stage_seedable<P, s>()) return tw_lane(tw2, (tid % SI::L) * SI::TWSTEP);
    else return cd{1.0, 0.0};
}

// mid stages 1..S-2 (plans have S <= 5, so at most three of them).  `hook(slot)` is
// called once per butterfly round with a compile-time slot number.
template <class P, int s, int SLOT0, class Hook>
__device__ __forceinline__ void mid_stages_seeded(cd* lds, const cd* tw2, const cd (&seed)[4],
                                                  int tid, Hook&& hook) {
    if constexpr (s <= P::S - 2) {
        if constexpr (stage_seedable<P, s>()) {
            fwd_stage_lds_seeded<P, s>(lds, seed[s], tid, [&](int m) { hook(SLOT0 + m); });
            __syncthreads();
            mid_stages_seeded<P, s + 1, SLOT0 + StageInfo<P, s>::K>(lds, tw2, seed, tid, hook);
        } else {
            fwd_stage_lds<P, s>(lds, tw2, tid);
            __syncthreads();
            mid_stages_seeded<P, s + 1, SLOT0>(lds, tw2, seed, tid, hook);
        }
    }
}

// Shared epilogue: LDS holds q = IDFT_M(P_A + i P_B) in natural order.
// lag-n value = Re(a + conj(W_{2M}^n) * b) / (2M) / (T-n), with
// a = (q[n] + conj(q[M-n]))/2, b = (q[n] - conj(q[M-n]))/(2i).
template <class P>
__device__ __forceinline__ double lag_value(const cd* __restrict__ lds,
                                            const cd* __restrict__ tw2, int n, int T) {
    const cd qn = lds[sw(n)];
    const cd qm = lds[sw((P::M - n) % P::M)];
    const cd a = {0.5 * (qn.x + qm.x), 0.5 * (qn.y - qm.y)};
    // (qn - conj(qm)) / (2i) = ( (qn.y + qm.y) - i (qn.x - qm.x) ) / 2
    const cd b = {0.5 * (qn.y + qm.y), -0.5 * (qn.x - qm.x)};
    const cd w = tw_lane(tw2, n);  // exp(-i pi n / M); need its conjugate
    const double re = a.x + (b.x * w.x + b.y * w.y);
    return re / (2.0 * (double)P::M * (double)(T - n));  // 2M(T-n) < 2^53: exact product
}

// ---- K3 (timeseries path): one inverse transform of the summed spectrum --------
// spec: [n_slices][2][M] (pass A bins, pass B bins; the slices are partial sums over disjoint
// sets of workgroups), digit-reversed order.
template <class P>
__global__ void __launch_bounds__(P::NT)
    k_fft_finalize(const double* __restrict__ spec, const cd* __restrict__ tw2, int T,
                   double* __restrict__ lagsum, int n_slices) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    const int tid = threadIdx.x;
    // 4 elements x n_slices x 2 independent loads in flight per thread
    for (int i0 = tid; i0 < P::M; i0 += 4 * P::NT) {
        double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
        for (int y = 0; y < n_slices; ++y) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * P::NT;
                if (i < P::M) {
                    a[k] += spec[(long)y * 2 * P::M + i];
                    b[k] += spec[(long)y * 2 * P::M + P::M + i];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * P::NT;
            if (i < P::M) lds[sw(i)] = cd{a[k], b[k]};
        }
    }
    __syncthreads();
    inv_all_stages<P, P::S - 1>(lds, tw2, tid);
    for (int n = tid; n < T; n += P::NT) lagsum[n] = lag_value<P>(lds, tw2, n, T);
}

}  // namespace ta
