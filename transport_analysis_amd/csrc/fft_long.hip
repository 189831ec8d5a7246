// fft_long.hip — FFT VACF lag sums for trajectories longer than the largest on-chip
// transform (n_frames > 10240), timeseries path.
//
// Replaces the same reference code as fft_kernels.hpp (VelocityAutocorr._conclude_fft +
// tidynamics.acf, /root/reference/transport_analysis/velocityautocorr.py:208-215) where a
// column no longer fits one workgroup's LDS.
//
// Maths.  Pad to 2M' with M' = R*M >= n_frames, M an on-chip plan length (8192 or 10240) and R
// the outer radix (2, 4, 8, 16).  Bin k = 2R*s + c (c < 2R, s < M) of the 2M'-point transform of
// the zero-padded pair series z is one output of an M-point transform:
//     Z[2R s + c] = FFT_M(u_c)[s],
//     u_c[t] = W_{2M'}^{c t} * sum_{j<R} z[t + M j] * W_{2R}^{c j},     t < M
// (a radix-2R decimation-in-frequency step done while the column is read).  So a workgroup
// runs 2R passes per column pair, each the on-chip M-point transform of a series it forms
// from R strided rows per element, and adds |.|^2 into that pass's accumulator block in
// global memory.  The lag sums are linear in the summed spectrum; with P[k] summed over pairs
//     lagsum[n] = (1 / (2M' (T - n))) * sum_k P[k] cos(pi k n / M'),     n < T,
// evaluated directly (k_long_lags, over the spectrum folded about k = M'): one launch per
// analysis, O(T * M') table look-ups, where an inverse transform of length 2M' would not fit on
// chip either.
//
// Cost: the column is gathered once per pair; the 2R derived series cross L2 twice (2R*M*16
// bytes written and read back per pair) and nothing is software-pipelined: O(T log T) with a
// larger constant than the on-chip path, meant to replace the O(T^2) direct correlator beyond
// the on-chip limit.
#include <hip/hip_runtime.h>

#include <vector>

#include "fft_kernels.hpp"
#include "plans.hpp"
#include "ta_internal.hpp"

namespace ta {
namespace {

__device__ __forceinline__ cd cfma(cd acc, cd a, cd b) {  // acc + a*b
    return {acc.x + (a.x * b.x - a.y * b.y), acc.y + (a.x * b.y + a.y * b.x)};
}

// Phase 0 for FOUR adjacent column pairs at once (8 complete columns, 16-byte aligned rows):
// a lane's four 16-byte loads of a row fall into the same 64 bytes, so the texture addresser
// sees one scattered line per row and three hits on it instead of four scattered requests --
// the "wide gather" of DESIGN.md section 6, possible here because the data goes to scratch
// anyway and does not have to be parked in registers.
template <class P, int ROUT>
__device__ __forceinline__ void long_phase0_quad(const double* __restrict__ col, long ld_row, long pair_stride, int T,
                                                 const double* __restrict__ zeros,
                                                 const cd* __restrict__ twL, cd* __restrict__ scr, int tid) {
    using S0 = StageInfo<P, 0>;
    static_assert(S0::TASKS % P::NT == 0, "every thread owns K first-stage butterflies");
    static_assert(ROUT <= 4, "register budget: G*ROUT*4 loads of 16 bytes in flight");
    constexpr int NE = S0::K * S0::R;
    constexpr int G = 4 / ROUT;
    constexpr int L2 = 2 * ROUT * P::M;
    constexpr long QS = 2L * ROUT * P::M;  // scratch elements per pair
    asm volatile("" : "+v"(tid));
#pragma unroll 1
    for (int e0 = 0; e0 < NE; e0 += G) {
        cd z[G][ROUT][4];
        int t1s[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int e = e0 + g < NE ? e0 + g : NE - 1;
            const int m = e / S0::R, j0 = e - m * S0::R;
            const int t1 = tid + m * P::NT + j0 * S0::L;
            t1s[g] = t1;
#pragma unroll
            for (int j = 0; j < ROUT; ++j) {
                const int t = t1 + P::M * j;
                const double* p = t < T ? col + (long)t * ld_row : zeros;  // 64 zero bytes there
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double2 v = *reinterpret_cast<const double2*>(p + (t < T ? q * pair_stride : 0));
                    z[g][j][q] = cd{v.x, v.y};
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll 1
            for (int c = 0; c < 2 * ROUT; ++c) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cd a = z[g][0][q];
#pragma unroll
                    for (int j = 1; j < ROUT; ++j) a = cfma(a, z[g][j][q], twL[(c * j * P::M) % L2]);
                    scr[q * QS + (long)c * P::M + t1s[g]] = a;
                }
            }
        }
    }
}

// accg: [gridDim.x][2*Rout][ACC_BLK] float64, zeroed by the caller; block layout as in
// k_fft_accum ([quad][thread] x 2 doubles).  twL: W_{2M'}^n = exp(-i pi n / M'), n < 2M'.
// scratch: [gridDim.x][4][2*Rout][M] complex (a work unit is four adjacent pairs): the column is gathered ONCE per pair (phase 0: R
// strided rows per element, all in flight together) and the 2R partial sums
// a_c[t] = sum_j z[t + M j] W_{2R}^{c j} go to the workgroup's scratch (lane = row: contiguous
// 1 KB stores), from which pass c reads its series back contiguously -- a thread reads exactly
// the elements it wrote.  Per pair: R*M scattered 16-byte requests instead of 2R*R*M.
constexpr int kMaxRout = 16;

// Phase 0 for a compile-time outer radix: the R rows of G = 16/R elements (16 scattered loads)
// are in flight together before any of them is used; left to a rolled loop every element's
// loads would expose a full memory latency (measured 364k cycles per pair at R = 2, 3/4 of
// the kernel).
template <class P, int ROUT, bool WIDE>
__device__ __forceinline__ void long_phase0(const double* __restrict__ col, long ld_row, int T,
                                            bool two, const double* __restrict__ zeros,
                                            const cd* __restrict__ twL, cd* __restrict__ scr, int tid) {
    using S0 = StageInfo<P, 0>;
    static_assert(S0::TASKS % P::NT == 0, "every thread owns K first-stage butterflies");
    constexpr int NE = S0::K * S0::R;  // elements per thread
    constexpr int G = kMaxRout / ROUT;
    constexpr int L2 = 2 * ROUT * P::M;
    asm volatile("" : "+v"(tid));  // row offsets are formed per pair, not hoisted and spilled
#pragma unroll 1
    for (int e0 = 0; e0 < NE; e0 += G) {
        cd z[G][ROUT];
        int t1s[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int e = e0 + g < NE ? e0 + g : NE - 1;  // the tail repeats the last element
            const int m = e / S0::R, j0 = e - m * S0::R;
            const int t1 = tid + m * P::NT + j0 * S0::L;
            t1s[g] = t1;
#pragma unroll
            for (int j = 0; j < ROUT; ++j) {
                const int t = t1 + P::M * j;
                // rows past the end are the zero padding: read the zero block instead
                const double* p = t < T ? col + (long)t * ld_row : zeros;
                if constexpr (WIDE) {  // no branch between the loads: they all go out first
                    const double2 v = *reinterpret_cast<const double2*>(p);
                    z[g][j] = cd{v.x, v.y};
                } else {
                    const double im = p[two ? 1 : 0];  // no partner column: imaginary part zero
                    z[g][j] = cd{p[0], two ? im : 0.0};
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll 1
            for (int c = 0; c < 2 * ROUT; ++c) {
                cd a = z[g][0];
#pragma unroll
                for (int j = 1; j < ROUT; ++j) a = cfma(a, z[g][j], twL[(c * j * P::M) % L2]);  // W_{2R}^{c j}
                scr[(long)c * P::M + t1s[g]] = a;
            }
        }
    }
}

template <class P>
__global__ void __launch_bounds__(P::NT)
    k_fft_accum_long(const double* __restrict__ vel, long ld_row, long pair_stride, int T, long n_cols, int Rout,
                     const cd* __restrict__ tw2, const cd* __restrict__ twL,
                     double* __restrict__ accg, cd* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    using S0 = StageInfo<P, 0>;
    constexpr long ACC_BLK = (long)acc_quads<P>() * 2 * P::NT;
    const int nwg = gridDim.x, wg = blockIdx.x;
    int slot = wg;
    if (nwg % 8 == 0) slot = (wg % 8) * (nwg / 8) + wg / 8;  // XCD-aware walk, as k_fft_accum
    const int tid = threadIdx.x;
    const int L2 = 2 * Rout * P::M;  // table length 2M'
    double* blk0 = accg + (long)wg * 2 * Rout * ACC_BLK;
    cd* scr = scratch + (long)wg * 4 * 2 * Rout * P::M;  // four pairs' worth
    const double* zeros = reinterpret_cast<const double*>(tw2 + 4 * P::M);  // 64 zero bytes

    cd seed[4];
    seed[0] = cd{1.0, 0.0};
    seed[1] = stage_seed<P, (P::S > 2 ? 1 : 0)>(tw2, tid);
    seed[2] = stage_seed<P, (P::S > 3 ? 2 : 0)>(tw2, tid);
    seed[3] = stage_seed<P, (P::S > 4 ? 3 : 0)>(tw2, tid);

    const bool slab16 = ((reinterpret_cast<unsigned long long>(vel) | ((unsigned long long)ld_row * 8)) & 15) == 0;
    const long n_pairs = (n_cols + 1) / 2;
    const long n_quads = (n_pairs + 3) / 4;  // work unit: four adjacent column pairs
    const long QS = 2L * Rout * P::M;        // scratch elements per pair
    auto no_hook = [](int) {};
    for (long quad = slot; quad < n_quads; quad += nwg) {
      const long pair0 = 4 * quad;
      const int npq = (int)(n_pairs - pair0 < 4 ? n_pairs - pair0 : 4);
      // ---- phase 0: gather once, partial sums of all 2R passes -> scratch
      const bool quadwide = slab16 && Rout <= 4 && 2 * (pair0 + 4) <= n_cols;
      if (quadwide) {
          if (Rout == 2) long_phase0_quad<P, 2>(vel + pair0 * pair_stride, ld_row, pair_stride, T, zeros, twL, scr, tid);
          else long_phase0_quad<P, 4>(vel + pair0 * pair_stride, ld_row, pair_stride, T, zeros, twL, scr, tid);
      }
      for (int pq = 0; pq < npq; ++pq) {
        const long pair = pair0 + pq;
        cd* scr_p = scr + pq * QS;
        if (quadwide) continue;
        const double* col = vel + pair * pair_stride;
        const bool two = 2 * pair + 1 < n_cols;  // an odd last column has no partner
        const bool wide = two && slab16;         // one 16-byte load per row
#define TA_P0(R)                                                                        \
    if (wide) long_phase0<P, R, true>(col, ld_row, T, two, zeros, twL, scr_p, tid);    \
    else long_phase0<P, R, false>(col, ld_row, T, two, zeros, twL, scr_p, tid);
        switch (Rout) {
            case 2: TA_P0(2) break;
            case 4: TA_P0(4) break;
            case 8: TA_P0(8) break;
            default: TA_P0(16) break;
        }
#undef TA_P0
      }
      __threadfence();
      for (int pq = 0; pq < npq; ++pq) {
        const cd* scr_p = scr + pq * QS;
        for (int c = 0; c < 2 * Rout; ++c) {
            // ---- first stage of pass c: u_c[t] = W_{2M'}^{c t} a_c[t].  Per butterfly u the inputs
            // are scaled by the wave-uniform W_{2M'}^{c j0 L} (hoisted: tj), the outputs by
            // W_{2M'}^{c u} * W_M^{u q} = h g^q (one table entry each per butterfly, powers by
            // repeated multiplication); the next butterfly's operands are loaded while the
            // current one is computed.
            const cd* __restrict__ ac = scr_p + (long)c * P::M;
            // per-butterfly offsets depend on tid only: formed here, per pass, or LICM hoists the
            // lot out of both loops and spills it
            int tl = tid;
            asm volatile("" : "+v"(tl));
            cd tj[S0::R];
#pragma unroll
            for (int j0 = 0; j0 < S0::R; ++j0) tj[j0] = twL[(c * j0 * S0::L) % L2];
            cd wbuf[2][S0::R], hb[2], gb[2];
            auto fetch = [&](auto mc, int which) {
                constexpr int m = decltype(mc)::value;
                const int u = tl + m * P::NT;
                if (S0::TASKS % P::NT == 0 || u < S0::TASKS) {
#pragma unroll
                    for (int j0 = 0; j0 < S0::R; ++j0) wbuf[which][j0] = ac[u + j0 * S0::L];
                    hb[which] = twL[(unsigned)(u * c) % (unsigned)L2];  // W_{2M'}^{c u}
                    gb[which] = tw2[2 * u];                              // W_M^u
                }
            };
            fetch(std::integral_constant<int, 0>{}, 0);
            static_for<S0::K>([&](auto mm) {
                constexpr int m = decltype(mm)::value;
                const int u = tl + m * P::NT;
                if constexpr (m + 1 < S0::K) fetch(std::integral_constant<int, m + 1>{}, (m + 1) & 1);
                // one butterfly ahead, not all of them (the scheduler would cluster every load
                // of the unrolled stage at its top and spill)
                __builtin_amdgcn_sched_barrier(0);
                if (S0::TASKS % P::NT == 0 || u < S0::TASKS) {
                    cd w[S0::R];
                    w[0] = wbuf[m & 1][0];
#pragma unroll
                    for (int j0 = 1; j0 < S0::R; ++j0) w[j0] = cmul(wbuf[m & 1][j0], tj[j0]);
                    Dft<S0::R>::run(w);
                    const int sb = sw(u);
                    cd te = hb[m & 1];
                    const cd g = gb[m & 1];
#pragma unroll
                    for (int q = 0; q < S0::R; ++q) {
                        lds[sw_off(sb, q * S0::L)] = cmul(w[q], te);
                        te = cmul(te, g);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            __syncthreads();
            mid_stages_seeded<P, 1, 0>(lds, tw2, seed, tid, no_hook);
            last_stage_acc_global<P>(lds, blk0 + (long)c * ACC_BLK, tid, no_hook);
            __syncthreads();
        }
      }
    }
}

// Sum the workgroups' blocks (fixed order) and put the spectrum into natural bin order:
// P[2R*perm[p] + c] = sum_w block[w][c][(m, q)], p = u*R_last + q, u = tid + m*NT.
__global__ void k_long_spectrum(const double* __restrict__ partial, int n_parts, int n_pass, int NT,
                                int R, int K, int TASKS, const int* __restrict__ perm,
                                double* __restrict__ spec) {
    const int quads = (K * R + 1) / 2;
    const long blk = (long)quads * 2 * NT;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // index within [n_pass][blk]
    if (i >= (long)n_pass * blk) return;
    const int c = (int)(i / blk);
    const long r = i - c * blk;
    const int comp = (int)(r & 1);
    const long qt = r >> 1;  // quad*NT + tid
    const int tid = (int)(qt % NT);
    const int d = 2 * (int)(qt / NT) + comp;
    if (d >= K * R) return;
    const int m = d / R, q = d % R;
    const int u = tid + m * NT;
    if (u >= TASKS) return;
    double s = 0.0;
    for (int w = 0; w < n_parts; ++w) s += partial[(long)w * n_pass * blk + i];
    spec[(long)n_pass * perm[u * R + q] + c] = s;
}

// The cosine is even in k about L2/2: fold P[k] + P[L2-k] once (in place, k <= L2/2), which
// halves the table look-ups of every lag.
__global__ void k_long_fold(double* __restrict__ spec, int L2) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > 0 && k < L2 / 2) spec[k] += spec[L2 - k];
}

// lagsum[n] = sum_k P[k] cos(2 pi k n / L2) / (L2 (T - n)) over the folded spectrum (k <= L2/2):
// one workgroup per lag, the table index (k n) mod L2 advanced by a fixed step per thread,
// tree reduction in fixed order.
__global__ void __launch_bounds__(256)
    k_long_lags(const double* __restrict__ spec, const cd* __restrict__ twL, int L2, int T,
                double* __restrict__ lagsum) {
    __shared__ double red[256];
    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    int idx = (int)(((long)tid * n) % L2);
    const int step = (int)((256L * n) % L2);
    double s = 0.0;
    for (int k = tid; k <= L2 / 2; k += 256) {
        s += spec[k] * twL[idx].x;
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

template <class P>
hipError_t launch_accum(int nwg, hipStream_t st, const double* vel, long ld_row, long pair_stride, int T, long n_cols,
                        int Rout, const cd* tw2, const cd* twL, double* accg, cd* scratch) {
    const size_t lds = (size_t)P::lds_elems() * sizeof(cd);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fft_accum_long<P>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_fft_accum_long<P>), dim3(nwg), dim3(P::NT), lds, st, vel, ld_row, pair_stride, T, n_cols,
                       Rout, tw2, twL, accg, scratch);
    return hipGetLastError();
}

using PlanA = Plan<256, 16, 8, 8, 8>;      // M = 8192
using PlanB = Plan<256, 5, 16, 16, 8>;     // M = 10240

template <class P>
void digit_perm(std::vector<int>& perm) {
    // position p = q0*(M/r0) + q1*(M/(r0 r1)) + ... holds frequency s = q0 + r0*(q1 + r1*(...))
    perm.assign(P::M, 0);
    for (int s = 0; s < P::M; ++s) {
        int rest = s, p = 0, block = P::M;
        for (int st = 0; st < P::S; ++st) {
            const int r = P::radix(st);
            block /= r;
            p += (rest % r) * block;
            rest /= r;
        }
        perm[p] = s;
    }
}

}  // namespace

bool fft_long_choose(long n_frames, int* M, int* Rout) {
    long best = 0;
    for (int R = 2; R <= 16; R *= 2)
        for (int m : {PlanA::M, PlanB::M}) {
            const long mp = (long)R * m;
            if (mp >= n_frames && (best == 0 || mp < best)) {
                best = mp;
                *M = m;
                *Rout = R;
            }
        }
    return best != 0;
}

void fft_long_perm(int M, std::vector<int>& perm) {
    if (M == PlanA::M) digit_perm<PlanA>(perm);
    else digit_perm<PlanB>(perm);
}

size_t fft_long_acc_block(int M) {  // doubles per workgroup and pass
    return (size_t)(M == PlanA::M ? acc_quads<PlanA>() * 2 * PlanA::NT : acc_quads<PlanB>() * 2 * PlanB::NT);
}

hipError_t launch_fft_long_accum(int M, int nwg, hipStream_t st, const double* vel, long ld_row, long pair_stride, int T,
                                 long n_cols, int Rout, const cd* tw2, const cd* twL, double* accg,
                                 cd* scratch) {
    if (Rout != 2 && Rout != 4 && Rout != 8 && Rout != 16) return hipErrorInvalidValue;
    if (M == PlanA::M)
        return launch_accum<PlanA>(nwg, st, vel, ld_row, pair_stride, T, n_cols, Rout, tw2, twL, accg, scratch);
    return launch_accum<PlanB>(nwg, st, vel, ld_row, pair_stride, T, n_cols, Rout, tw2, twL, accg, scratch);
}

hipError_t launch_fft_long_finish(int M, int Rout, const double* partial, int n_parts, const int* perm,
                                  const cd* twL, int T, double* spec, double* lagsum, hipStream_t st) {
    int NT, R, K, TASKS;
    if (M == PlanA::M) {
        using SL = StageInfo<PlanA, PlanA::S - 1>;
        NT = PlanA::NT, R = SL::R, K = SL::K, TASKS = SL::TASKS;
    } else {
        using SL = StageInfo<PlanB, PlanB::S - 1>;
        NT = PlanB::NT, R = SL::R, K = SL::K, TASKS = SL::TASKS;
    }
    const int n_pass = 2 * Rout;
    const long n = (long)n_pass * ((K * R + 1) / 2) * 2 * NT;
    hipLaunchKernelGGL(k_long_spectrum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial,
                       n_parts, n_pass, NT, R, K, TASKS, perm, spec);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_long_fold, dim3((unsigned)((n_pass * M / 2 + 255) / 256)), dim3(256), 0, st, spec,
                       n_pass * M);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_long_lags, dim3(T), dim3(256), 0, st, spec, twL, n_pass * M, T, lagsum);
    return hipGetLastError();
}

}  // namespace ta
