// direct_kernels.hpp — all-lag direct correlators (K4, K5+K6 of SURVEY.md 8a).
//
// MODE_VACF    : VelocityAutocorr._conclude_simple
//                (/root/reference/transport_analysis/velocityautocorr.py:217-238)
//                B[k,n] = 1/(T-k) sum_i sum_d v[i,n,d] v[i+k,n,d],  k = 0..T-1
// MODE_HELFAND : ViscosityHelfand._conclude
//                (/root/reference/transport_analysis/viscosity.py:201-233)
//                P = (m*v)*x ; H[k,n] = scale/(D (T-k)) sum_i sum_d (P[i]-P[i+k])^2,
//                k = 1..T-1, H[0,n] = 0.  The difference is formed first, exactly as
//                the reference does (no prefix-sum expansion: SURVEY.md 7.3-5).
//
// One workgroup takes one atom at a time.  For each of its D columns the whole
// time series (T float64, P formed on the fly for Helfand) is staged in LDS; a
// thread owns a PAIR of lag chunks (L consecutive lags each): chunk j and chunk
// nchunks-1-j, so every thread does the same number of multiply-adds although
// lag k has T-k terms.  Inside a chunk the thread register-tiles L lags x L time
// steps: per step it reads L new series values for the lagged window and L values
// for the leading window from LDS and issues L*L FMAs (f64 VALU; this path is
// FP64-FMA bound, not HBM bound: algorithmic intensity ~ T/8 flop per byte).
//
// LDS layout: groups of L values followed by 16 B of padding, so lane-strided
// 16-byte reads (lane stride = one group) hit 16 distinct bank slots.
#pragma once
#include <hip/hip_runtime.h>

namespace ta {

enum { MODE_VACF = 0, MODE_HELFAND = 1 };

template <int L>
__device__ __forceinline__ int lds_slot(int e) {  // element index -> f64 slot
    return (e / L) * (L + 2) + (e % L);
}

template <int L>
__device__ __forceinline__ void lds_read_group(const double* __restrict__ s, int e,
                                               double (&out)[L]) {
    // e is a multiple of L: one padded group, 16-byte aligned
    const double2* p = reinterpret_cast<const double2*>(s + (e / L) * (L + 2));
#pragma unroll
    for (int i = 0; i < L / 2; ++i) {
        double2 t = p[i];
        out[2 * i] = t.x;
        out[2 * i + 1] = t.y;
    }
}

// Accumulate one chunk (lags k0..k0+L-1) of one staged column into acc[L].
// n_valid = number of staged samples (T); the LDS copy is zero-padded beyond it.
template <int MODE, int L>
__device__ __forceinline__ void chunk_accumulate(const double* __restrict__ s, int T, int k0,
                                                 double (&acc)[L]) {
    // window w[0..2L): series values at i+k0 .. i+k0+2L-1 ; x[0..L): values at i..i+L-1
    double w[2 * L];
    {
        double lo[L];
        lds_read_group<L>(s, k0, lo);
#pragma unroll
        for (int a = 0; a < L; ++a) w[a] = lo[a];
    }
    const int span = T - k0;  // number of leading indices i with at least lag k0 valid
    const int nblk = (span + L - 1) / L;
    // blocks whose L x L tile is valid for every lag of the chunk: i + 2L - 2 + k0 < T
    int nfull = (T - k0 - 2 * L + 1) / L + 1;
    if (T - k0 - 2 * L + 1 < 0) nfull = 0;
    if (nfull > nblk) nfull = nblk;
    if (MODE == MODE_VACF) nfull = nblk;  // zero padding makes out-of-range products vanish
    int blk = 0;
    for (; blk < nfull; ++blk) {
        const int i = blk * L;
        double x[L], hi[L];
        lds_read_group<L>(s, i, x);
        lds_read_group<L>(s, i + k0 + L, hi);
#pragma unroll
        for (int a = 0; a < L; ++a) w[L + a] = hi[a];
        if (MODE == MODE_VACF) {
#pragma unroll
            for (int b = 0; b < L; ++b)
#pragma unroll
                for (int a = 0; a < L; ++a) acc[a] = fma(x[b], w[a + b], acc[a]);
        } else {
#pragma unroll
            for (int b = 0; b < L; ++b)
#pragma unroll
                for (int a = 0; a < L; ++a) {
                    const double d = x[b] - w[a + b];
                    acc[a] = fma(d, d, acc[a]);
                }
        }
#pragma unroll
        for (int a = 0; a < L; ++a) w[a] = w[L + a];
    }
    if (MODE == MODE_HELFAND) {
        // ragged end of the chunk (at most two blocks): a pair (i+b, i+b+k) only counts
        // while i+b+k < T.  Straight from the staged column, rolled over b: small code.
        for (; blk < nblk; ++blk) {
            const int i = blk * L;
#pragma unroll 1
            for (int b = 0; b < L; ++b) {
                const double xb = s[lds_slot<L>(i + b)];
#pragma unroll
                for (int a = 0; a < L; ++a) {
                    const int e = i + b + k0 + a;
                    const double d = xb - s[lds_slot<L>(e)];
                    if (e < T) acc[a] = fma(d, d, acc[a]);
                }
            }
        }
    }
}

// grid.x: persistent workgroups over atoms.  ts_partial: [gridDim.x][T] (zeroed by caller).
// GLOBAL_STAGE: trajectories too long for LDS (one column = (T/L+3)*(L+2)*8 bytes > 160 KiB)
// stage the column in this workgroup's slice of `stage_buf` instead ([gridDim.x][n_slots]
// float64, L1/L2 resident): same code, lower rate, no limit on n_frames.
template <int MODE, int L, bool GLOBAL_STAGE>
__global__ void __launch_bounds__(1024)
    k_direct(const double* __restrict__ vel, const double* __restrict__ pos,
             const double* __restrict__ masses, long ld_row, int T, long n_atoms, int D,
             double scale, double* __restrict__ by_particle, long ld_bp,
             double* __restrict__ ts_partial, double* __restrict__ stage_buf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* s = reinterpret_cast<double*>(smem_raw);
    if constexpr (GLOBAL_STAGE)
        s = stage_buf + (long)blockIdx.x * ((long)((T + L - 1) / L + 3) * (L + 2));
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nchunks = (T + L - 1) / L;
    const int npairs = (nchunks + 1) / 2;
    // staged length: up to the end of the window any chunk can touch, zero padded
    const int n_stage = (nchunks + 3) * L;
    double* ts_out = ts_partial + (long)blockIdx.x * T;

    for (long atom = blockIdx.x; atom < n_atoms; atom += gridDim.x) {
        const double mass = (MODE == MODE_HELFAND) ? masses[atom] : 1.0;
        for (int pp0 = 0; pp0 < npairs; pp0 += nt) {
            const int pp = pp0 + tid;
            const bool active = pp < npairs;
            const int j1 = pp, j2 = nchunks - 1 - pp;
            double acc1[L], acc2[L];
#pragma unroll
            for (int a = 0; a < L; ++a) acc1[a] = acc2[a] = 0.0;
            for (int d = 0; d < D; ++d) {
                __syncthreads();  // previous column fully consumed
                for (int e = tid; e < n_stage; e += nt) {
                    double val = 0.0;
                    if (e < T) {
                        const long g = (long)e * ld_row + atom * D + d;
                        val = vel[g];
                        if (MODE == MODE_HELFAND) val = (mass * val) * pos[g];
                    }
                    s[lds_slot<L>(e)] = val;
                }
                __syncthreads();
                if (active) {
                    chunk_accumulate<MODE, L>(s, T, j1 * L, acc1);
                    if (j2 != j1) chunk_accumulate<MODE, L>(s, T, j2 * L, acc2);
                }
            }
            if (active) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int jc = h ? j2 : j1;
                    if (h && j2 == j1) break;
#pragma unroll
                    for (int a = 0; a < L; ++a) {
                        const int k = jc * L + a;
                        if (k < T) {
                            double val = (h ? acc2[a] : acc1[a]) / (double)(T - k);
                            if (MODE == MODE_HELFAND) val = (k == 0) ? 0.0 : (val / (double)D) * scale;
                            if (by_particle) by_particle[(long)k * ld_bp + atom] = val;
                            ts_out[k] += val;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace ta
