// helfand_fft.hip — optional O(T log T) evaluation of the Einstein-Helfand lag sums
// (option "helfand_fft", timeseries path; SURVEY.md section 8(f) row f4: the reference has
// no such path, /root/reference/transport_analysis/viscosity.py:201-233 always runs the
// O(T^2) double loop, which k_direct<MODE_HELFAND> restates and which stays the default).
//
// With P[i] = (m v[i]) x[i] per column (the reference's evaluation order),
//   sum_{i<T-k} (P[i] - P[i+k])^2 = S1(k) - 2 S2(k),
//   S2(k) = sum_i P[i] P[i+k]                       -> the FFT VACF lag sums of the P slab
//   S1(k) = sum_{i<T-k} P[i]^2 + sum_{i>=k} P[i]^2  -> prefix sums of Q[i] = sum_cols P[i]^2
// so the sum over atoms of the windowed mean squared difference is
//   (C[T-k] + C[T] - C[k]) / (T-k) - 2 * lagsum_fft[k],   C = exclusive prefix sums of Q.
// Accuracy: the two terms are each ~2 sum P^2 / (T-k) and their difference is formed in
// float64: absolute error ~1e-16 * sum P^2 / (T-k), i.e. ~1e-15 of the series' scale, but the
// RELATIVE error of a lag whose mean squared difference is far below P^2 (short lags of a
// smooth P) grows by that ratio -- which is why this is an option and not the default.
#include <hip/hip_runtime.h>

#include "ta_internal.hpp"

namespace ta {
namespace {

// Pair-major slabs in and out (layout.hip).  A workgroup walks whole column pairs along time
// (coalesced 16-byte rows): P[t, pair] = ((m v) x) for both columns, and the pair's
// contributions P.x^2 + P.y^2 to Q[t], summed over the workgroup's pairs, go to its own row of
// Qpart ([gridDim.x][T], every element written; summed over workgroups in a fixed order afterwards).
__global__ void __launch_bounds__(256)
    k_helfand_product(const double* __restrict__ vel, const double* __restrict__ pos,
                      const double* __restrict__ masses, long pitch, long T, long n_cols, int D,
                      double* __restrict__ P, double* __restrict__ Qpart) {
    const long n_pairs = (n_cols + 1) / 2;
    double* q = Qpart + (long)blockIdx.x * T;
    // time in pieces of 1024 rows, the workgroup's pairs inside: a thread's four contributions
    // to Q stay in registers across the pairs and are stored once per piece (the row of Qpart
    // read and written once per pair was a third of this kernel's traffic)
    for (long t0 = 0; t0 < T; t0 += 1024) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (long pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
            const long c = 2 * pair;
            const double m0 = masses[c / D];
            const bool two = c + 1 < n_cols;
            const double m1 = two ? masses[(c + 1) / D] : 0.0;
            const double2* v = reinterpret_cast<const double2*>(vel) + pair * pitch;
            const double2* x = reinterpret_cast<const double2*>(pos) + pair * pitch;
            double2* p = reinterpret_cast<double2*>(P) + pair * pitch;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long t = t0 + threadIdx.x + 256 * i;
                if (t < T) {
                    const double2 vv = v[t], xx = x[t];
                    double2 r;
                    r.x = (m0 * vv.x) * xx.x;
                    r.y = two ? (m1 * vv.y) * xx.y : 0.0;
                    p[t] = r;
                    acc[i] += r.x * r.x + r.y * r.y;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long t = t0 + threadIdx.x + 256 * i;
            if (t < T) q[t] = acc[i];
        }
    }
}

// Single workgroup: C = exclusive prefix sums of Q (C[0] = 0 ... C[T]), then
// out[k] = factor * ((C[T-k] + C[T] - C[k]) / (T-k) - 2 s2n[k]), out[0] = 0 exactly
// (viscosity.py:205-233 leaves row 0 at 0).
__global__ void __launch_bounds__(256)
    k_helfand_combine(const double* __restrict__ Q, const double* __restrict__ s2n, double* __restrict__ C,
                      int T, double factor, double* __restrict__ out) {
    __shared__ double part[256];
    const int tid = threadIdx.x;
    const int chunk = (T + 255) / 256;
    const int lo = tid * chunk < T ? tid * chunk : T, hi = lo + chunk < T ? lo + chunk : T;
    double s = 0.0;
    for (int i = lo; i < hi; ++i) s += Q[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        double run = 0.0;
        for (int i = 0; i < 256; ++i) {
            const double v = part[i];
            part[i] = run;
            run += v;
        }
    }
    __syncthreads();
    double run = part[tid];
    for (int i = lo; i < hi; ++i) {
        C[i] = run;
        run += Q[i];
    }
    if (lo < T && hi == T) C[T] = run;  // the thread that owns the last non-empty range
    __threadfence_block();
    __syncthreads();
    const double total = C[T];
    for (int k = tid; k < T; k += 256) {
        if (k == 0) {
            out[0] = 0.0;
        } else {
            const double s1 = C[T - k] + (total - C[k]);
            out[k] = factor * (s1 / (double)(T - k) - 2.0 * s2n[k]);
        }
    }
}

// By-particle variant.  k_helfand_product_bp: P slab and Qa[t, n] = sum_d P[t, n, d]^2 (into the
// (T+1, n_atoms) prefix array's rows 1..T).  k_helfand_combine_bp: in-place prefix sums over
// time per atom (coalesced over atoms), then
// bp[k, n] = factor * ((C[T-k] + C[T] - C[k]) / (T-k) - 2 bp[k, n]) with bp holding the FFT
// by-particle autocorrelation of P on entry; row 0 is set to exactly 0.
__global__ void __launch_bounds__(256)
    k_helfand_product_bp(const double* __restrict__ vel, const double* __restrict__ pos,
                         const double* __restrict__ masses, long pitch, long T, long n_atoms, int D,
                         double* __restrict__ P, double* __restrict__ Ca) {
    // 64 atoms x 64 frames per workgroup; a thread walks 16 consecutive frames of its atom
    // (pair-major slabs: consecutive rows of a column share cache lines), lanes = atoms, so
    // the Ca rows are written 512 bytes at a time
    const long n = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (n >= n_atoms) return;
    const long t0 = (long)blockIdx.y * 64 + (threadIdx.x >> 6) * 16;
    const double m = masses[n];
    for (long t = t0; t < t0 + 16 && t < T; ++t) {
        double s = 0.0;
        for (int d = 0; d < D; ++d) {
            const long c = n * D + d;
            const long g = ((c >> 1) * pitch + t) * 2 + (c & 1);
            const double val = (m * vel[g]) * pos[g];
            P[g] = val;
            s += val * val;
        }
        Ca[(t + 1) * n_atoms + n] = s;
    }
}

// 64 atoms x 4 time quarters per workgroup: every thread scans its quarter of its atom's
// column (local prefix sums in place), the quarters' totals meet in LDS, and the combine
// pass adds the quarter offsets on the fly.  Lanes of a wave are consecutive atoms: 512-byte
// contiguous accesses.
__global__ void __launch_bounds__(256)
    k_helfand_combine_bp(double* __restrict__ Ca, long n_atoms, int T, double factor,
                         double* __restrict__ bp, long ld_bp) {
    __shared__ double tot[4][64];
    const int a = threadIdx.x & 63, c = threadIdx.x >> 6;
    const long n = (long)blockIdx.x * 64 + a;
    const bool live = n < n_atoms;
    const int Tc = (T + 3) / 4;  // rows 1..T of Ca in four quarters of Tc rows
    const int lo = 1 + c * Tc, hi = (lo + Tc - 1 < T) ? lo + Tc - 1 : T;
    double run = 0.0;
    if (live) {
        if (c == 0) Ca[n] = 0.0;
        for (int t = lo; t <= hi; ++t) {
            run += Ca[(long)t * n_atoms + n];
            Ca[(long)t * n_atoms + n] = run;
        }
    }
    tot[c][a] = run;
    __threadfence_block();
    __syncthreads();
    if (!live) return;
    const double o1 = tot[0][a], o2 = o1 + tot[1][a], o3 = o2 + tot[2][a], total = o3 + tot[3][a];
    auto C = [&](int t) -> double {  // exclusive prefix sum C[t], t in [0, T]
        if (t == 0) return 0.0;
        const int q = (t - 1) / Tc;
        const double off = q == 0 ? 0.0 : q == 1 ? o1 : q == 2 ? o2 : o3;
        return Ca[(long)t * n_atoms + n] + off;
    };
    // lags split over the four threads of the atom
    const int Kc = (T + 3) / 4;
    const int k0 = c * Kc, k1 = (k0 + Kc < T) ? k0 + Kc : T;
    for (int k = k0; k < k1; ++k) {
        if (k == 0) {
            bp[n] = 0.0;
        } else {
            const double s1 = C(T - k) + (total - C(k));
            bp[(long)k * ld_bp + n] = factor * (s1 / (double)(T - k) - 2.0 * bp[(long)k * ld_bp + n]);
        }
    }
}

}  // namespace

hipError_t launch_helfand_product_bp(const double* vel, const double* pos, const double* masses,
                                     long pitch, long T, long n_atoms, int D, double* P, double* Ca,
                                     hipStream_t st) {
    hipLaunchKernelGGL(k_helfand_product_bp, dim3((unsigned)((n_atoms + 63) / 64), (unsigned)((T + 63) / 64)),
                       dim3(256), 0, st, vel, pos, masses, pitch, T, n_atoms, D, P, Ca);
    return hipGetLastError();
}

hipError_t launch_helfand_combine_bp(double* Ca, long n_atoms, int T, double factor, double* bp,
                                     long ld_bp, hipStream_t st) {
    hipLaunchKernelGGL(k_helfand_combine_bp, dim3((unsigned)((n_atoms + 63) / 64)), dim3(256), 0, st, Ca,
                       n_atoms, T, factor, bp, ld_bp);
    return hipGetLastError();
}

// Qpart: [n_parts][T], every element written; the caller sums it over n_parts into Q.
hipError_t launch_helfand_product(const double* vel, const double* pos, const double* masses,
                                  long pitch, long T, long n_cols, int D, double* P, double* Qpart,
                                  int n_parts, hipStream_t st) {
    hipLaunchKernelGGL(k_helfand_product, dim3((unsigned)n_parts), dim3(256), 0, st, vel, pos, masses, pitch,
                       T, n_cols, D, P, Qpart);
    return hipGetLastError();
}

hipError_t launch_helfand_combine(const double* Q, const double* s2n, double* C, int T, double factor,
                                  double* out, hipStream_t st) {
    hipLaunchKernelGGL(k_helfand_combine, dim3(1), dim3(256), 0, st, Q, s2n, C, T, factor, out);
    return hipGetLastError();
}

}  // namespace ta
