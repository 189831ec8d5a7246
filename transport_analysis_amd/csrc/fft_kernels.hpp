// fft_kernels.hpp — FFT-VACF kernels (K1..K3 of SURVEY.md section 8a).
//
// Replaces VelocityAutocorr._conclude_fft + tidynamics.acf
// (/root/reference/transport_analysis/velocityautocorr.py:208-215).
//
// Maths.  For a real column x of n_frames = T samples the reference needs
//   acf[k] = sum_i x[i] x[i+k] / (T-k),  k < T,
// which tidynamics evaluates as Re IFFT(|FFT(x padded to L)|^2) with L >= 2T-1.
// Here L = 2M with M >= T the smallest length of the form 2^a or 5*2^a, and:
//   * two adjacent real columns are packed into one complex series z = x + i*y;
//     Re IFFT(|Z|^2) = acf_x + acf_y exactly (the cross terms are odd), and the
//     reference only ever needs sums over columns (dims, then atoms);
//   * the 2M-point transform of a series whose upper half is zero splits into two
//     M-point transforms: even bins = FFT_M(z) ("pass A"), odd bins =
//     FFT_M(z[t] * exp(-i pi t / M)) ("pass B").  With the first radix R0 and
//     t = u + j*M/R0 the twist factors as exp(-i pi u/M) * W_{2 R0}^j: a
//     lane-uniform constant per input, and the lane-dependent part merges into
//     the stage twiddle: output q of butterfly u is scaled by W_{2M}^{u (2q + B)},
//     B = 0 for pass A and 1 for pass B;
//   * the timeseries is linear in the power spectra, so a workgroup accumulates
//     |Z|^2 over all its column pairs in registers (in the transform's own
//     digit-reversed order) and ONE inverse transform per launch, not per atom,
//     turns the summed spectrum into the lag-indexed sum.
// tw2 is the single twiddle table W_{2M}^n = exp(-i pi n / M), n < 2M.
// Algorithmic HBM bytes: every input element is needed once: T*A*D*8 bytes.
#pragma once
#include "fft_engine.hpp"

namespace ta {

// ---- input access -----------------------------------------------------------
// z[t] = col[t*ld_row] + i*col[t*ld_row + 1] (imaginary part 0 when !has2).
template <bool VEC>
__device__ __forceinline__ cd load_z(const double* __restrict__ col, long ld_row, int t,
                                     bool has2) {
    const double* p = col + (long)t * ld_row;
    if constexpr (VEC) {
        const double2 v = *reinterpret_cast<const double2*>(p);
        return {v.x, v.y};
    } else {
        cd r;
        r.x = p[0];
        r.y = has2 ? p[1] : 0.0;
        return r;
    }
}

// First forward stage fused with the global load and (PASSB) the twist.
template <class P, bool VEC, bool PASSB>
__device__ __forceinline__ void fwd_first_stage(cd* __restrict__ lds, const cd* __restrict__ tw2,
                                                const double* __restrict__ col, long ld_row,
                                                int T, bool has2, int tid, int flags = 0) {
    using SI = StageInfo<P, 0>;
    cd v[SI::K][SI::R];
    // all of the thread's loads first: K*R independent requests in flight
#pragma unroll
    for (int m = 0; m < SI::K; ++m) {
        const int u = tid + m * P::NT;
#pragma unroll
        for (int j = 0; j < SI::R; ++j) {
            const int t = u + j * SI::L;
            cd z = {0.0, 0.0};
            if ((SI::TASKS % P::NT == 0 || u < SI::TASKS) && t < T) {
                if (flags & 1) z = cd{(double)t, 1.0};  // timing diagnostics only
                else z = load_z<VEC>(col, ld_row, t, has2);
            }
            v[m][j] = z;
        }
    }
#pragma unroll
    for (int m = 0; m < SI::K; ++m) {
        const int u = tid + m * P::NT;
        if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
            if constexpr (PASSB) {
                // lane-uniform part of the twist: W_{2 R0}^j = tw2[j * L]
#pragma unroll
                for (int j = 1; j < SI::R; ++j) v[m][j] = cmul(v[m][j], tw2[j * SI::L]);
            }
            Dft<SI::R>::run(v[m]);
#pragma unroll
            for (int q = PASSB ? 0 : 1; q < SI::R; ++q)
                v[m][q] = cmul(v[m][q], tw2[u * (2 * q + (PASSB ? 1 : 0))]);
#pragma unroll
            for (int q = 0; q < SI::R; ++q) lds[sw(u + q * SI::L)] = v[m][q];
        }
    }
}

// Last forward stage fused with |.|^2 accumulation into registers.
template <class P>
__device__ __forceinline__ void fwd_last_stage_acc(
    const cd* __restrict__ lds,
    double (&acc)[StageInfo<P, P::S - 1>::K][StageInfo<P, P::S - 1>::R], int tid) {
    using SI = StageInfo<P, P::S - 1>;
    static_assert(SI::L == 1, "last stage must have unit stride");
#pragma unroll
    for (int m = 0; m < SI::K; ++m) {
        const int u = tid + m * P::NT;
        if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
            cd v[SI::R];
#pragma unroll
            for (int j = 0; j < SI::R; ++j) v[j] = lds[sw(u * SI::R + j)];
            Dft<SI::R>::run(v);
#pragma unroll
            for (int q = 0; q < SI::R; ++q) acc[m][q] += norm2(v[q]);
        }
    }
}

// One full forward pass (A or B) of one column pair, accumulated into acc.
template <class P, bool VEC, bool PASSB>
__device__ __forceinline__ void forward_pass_acc(
    cd* lds, const cd* tw2, const double* col, long ld_row, int T, bool has2,
    double (&acc)[StageInfo<P, P::S - 1>::K][StageInfo<P, P::S - 1>::R], int tid, int flags = 0) {
    // The per-thread twiddles are the same for every column pair; left alone, LICM
    // hoists ~50 complex values per thread out of the pair loop and spills them.
    // Laundering the (wave-uniform) table pointer keeps them as L1/L2-served loads.
    // Same for the per-lane gather addresses and table offsets (all functions of
    // tid and ld_row only): recomputing them per pass is cheaper than spilling.
    asm volatile("" : "+s"(tw2), "+s"(ld_row), "+v"(tid));
    fwd_first_stage<P, VEC, PASSB>(lds, tw2, col, ld_row, T, has2, tid, flags);
    __syncthreads();
    if (flags & 2) return;  // timing diagnostics only: load + first stage
    fwd_mid_stages<P, 1>(lds, tw2, tid);
    fwd_last_stage_acc<P>(lds, acc, tid);
    __syncthreads();
}

// ---- K1+K2: accumulate power spectra over column pairs ------------------------
// Persistent workgroups.  A workgroup runs ONE pass type (A: even bins, B: odd
// bins) over its share of the column pairs, so it carries a single accumulator
// set; the A and the B workgroup of a pair are dealt to the same XCD (blockIdx % 8)
// and walk the pairs in the same order, so the second reader of a line finds it in
// that XCD's L2.  Workgroups of one XCD take consecutive pairs (adjacent columns).
// partial: [2][n_slots][P::M] float64 in the transform's digit-reversed bin order,
// n_slots = gridDim.x / 2.  Requires gridDim.x even (and a multiple of 16 for the
// XCD-aware walk; otherwise the walk degrades to a plain interleave).
template <class P, bool VEC, bool PASSB>
__device__ __forceinline__ void accum_body(cd* lds, const double* __restrict__ vel, long ld_row,
                                           int T, long n_cols, const cd* __restrict__ tw2,
                                           double* __restrict__ out, int slot, int n_slots,
                                           long pair_stride, int flags) {
    using SL = StageInfo<P, P::S - 1>;
    const int tid = threadIdx.x;
    double acc[SL::K][SL::R];
#pragma unroll
    for (int m = 0; m < SL::K; ++m)
#pragma unroll
        for (int q = 0; q < SL::R; ++q) acc[m][q] = 0.0;
    const long n_pairs = (n_cols + 1) / 2;
    for (long pair = slot; pair < n_pairs; pair += n_slots) {
        const bool has2 = 2 * pair + 1 < n_cols;
        forward_pass_acc<P, VEC, PASSB>(lds, tw2, vel + pair * pair_stride, ld_row, T, has2, acc,
                                        tid, flags);
    }
#pragma unroll
    for (int m = 0; m < SL::K; ++m) {
        const int u = tid + m * P::NT;
        if (SL::TASKS % P::NT == 0 || u < SL::TASKS) {
#pragma unroll
            for (int q = 0; q < SL::R; ++q) out[u * SL::R + q] = acc[m][q];
        }
    }
}

template <class P, bool VEC>
__global__ void __launch_bounds__(P::NT)
    k_fft_accum(const double* __restrict__ vel, long ld_row, long pair_stride, int T, long n_cols,
                const cd* __restrict__ tw2, double* __restrict__ partial, int flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    const int nwg = gridDim.x, wg = blockIdx.x;
    const int n_slots = nwg / 2;
    int pass, slot;
    if (nwg % 16 == 0) {
        // blocks b and b+8 share an XCD: (xcd, r) with r = b / 8; pass = r & 1.
        const int xcd = wg % 8, r = wg / 8, per_xcd = nwg / 16;
        pass = r & 1;
        slot = xcd * per_xcd + (r >> 1);
    } else {
        pass = wg & 1;
        slot = wg >> 1;
    }
    double* out = partial + ((long)pass * n_slots + slot) * P::M;
    if (pass == 0)
        accum_body<P, VEC, false>(lds, vel, ld_row, T, n_cols, tw2, out, slot, n_slots, pair_stride, flags);
    else
        accum_body<P, VEC, true>(lds, vel, ld_row, T, n_cols, tw2, out, slot, n_slots, pair_stride, flags);
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
    const cd w = tw2[n];  // exp(-i pi n / M); need its conjugate
    const double re = a.x + (b.x * w.x + b.y * w.y);
    return re / (2.0 * (double)P::M) / (double)(T - n);
}

// ---- K3 (timeseries path): one inverse transform of the summed spectrum --------
// spec: [2][M] (pass A bins, pass B bins), digit-reversed order.
template <class P>
__global__ void __launch_bounds__(P::NT)
    k_fft_finalize(const double* __restrict__ spec, const cd* __restrict__ tw2, int T,
                   double* __restrict__ lagsum) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    const int tid = threadIdx.x;
    for (int i = tid; i < P::M; i += P::NT) lds[sw(i)] = cd{spec[i], spec[P::M + i]};
    __syncthreads();
    inv_all_stages<P, P::S - 1>(lds, tw2, tid);
    for (int n = tid; n < T; n += P::NT) lagsum[n] = lag_value<P>(lds, tw2, n, T);
}

// ---- by-particle path: per-atom spectra, inverse transform per atom ------------
// One workgroup per atom at a time: forward passes over the atom's columns
// (D=1: (x,0); D=2: (x,y); D=3: (x,y),(z,0)), inverse, scatter into
// by_particle[:, atom], and keep a running per-lag sum for the timeseries.
// ts_partial: [gridDim.x][T].
template <class P>
__global__ void __launch_bounds__(P::NT)
    k_fft_by_particle(const double* __restrict__ vel, long ld_row, int T, long n_atoms, int D,
                      const cd* __restrict__ tw2, double* __restrict__ by_particle, long ld_bp,
                      double* __restrict__ ts_partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    using SL = StageInfo<P, P::S - 1>;
    const int tid = threadIdx.x;
    double* ts = ts_partial + (long)blockIdx.x * T;  // zeroed by the caller

    for (long atom = blockIdx.x; atom < n_atoms; atom += gridDim.x) {
        double accA[SL::K][SL::R], accB[SL::K][SL::R];
#pragma unroll
        for (int m = 0; m < SL::K; ++m)
#pragma unroll
            for (int q = 0; q < SL::R; ++q) accA[m][q] = accB[m][q] = 0.0;
        const double* base = vel + atom * D;
        for (int c = 0; c < D; c += 2) {
            const bool has2 = c + 1 < D;
            forward_pass_acc<P, false, false>(lds, tw2, base + c, ld_row, T, has2, accA, tid);
            forward_pass_acc<P, false, true>(lds, tw2, base + c, ld_row, T, has2, accB, tid);
        }
#pragma unroll
        for (int m = 0; m < SL::K; ++m) {
            const int u = tid + m * P::NT;
            if (SL::TASKS % P::NT == 0 || u < SL::TASKS) {
#pragma unroll
                for (int q = 0; q < SL::R; ++q)
                    lds[sw(u * SL::R + q)] = cd{accA[m][q], accB[m][q]};
            }
        }
        __syncthreads();
        inv_all_stages<P, P::S - 1>(lds, tw2, tid);
        for (int n = tid; n < T; n += P::NT) {
            const double val = lag_value<P>(lds, tw2, n, T);
            by_particle[(long)n * ld_bp + atom] = val;
            ts[n] += val;
        }
        __syncthreads();
    }
}

}  // namespace ta
