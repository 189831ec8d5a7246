// wfft.hpp — per-wave FFT kernels on pair-major slabs (gfx950): power spectra of column pairs
// (forward kernel) and the lag values of a power spectrum (inverse kernel).
//
// Replaces the per-atom tidynamics.acf loop of VelocityAutocorr._conclude_fft
// (/root/reference/transport_analysis/velocityautocorr.py:208-215): both the mean over atoms
// (results.timeseries) and the per-atom array (results.vacf_by_particle).
//
// Input layout ("pair-major", produced by ta_stage_commit / ta_relayout_dev): column pair p
// (columns 2p, 2p+1 of the (n_atoms*dim) columns) is one contiguous array of `pitch` rows of
// 16 bytes, row t = (x[t], y[t]) = the complex sample z[t] = x[t] + i y[t].  A wave reads 1 KiB
// of consecutive rows per load instruction; rows past n_frames read as zero through the buffer
// descriptor's bounds check, which IS the zero padding.
//
// Transform.  The series is padded to L = 2 R M points, M = R0 * 512 the on-chip length
// (R0 in {2, ..., 10, 12, 14, 16, 18, 20}) and R the outer radix (1 up to 10240 frames, then 2, 3, 4, 5, 8, 16).
// Bin k = 2R s + c of the L-point transform is output s of an M-point transform ("pass" c < 2R):
//     Z[2R s + c] = FFT_M(u_c)[s],   u_c[t] = W_L^{c t} sum_{jo < R} z[t + M jo] W_2R^{c jo},  t < M
// (R = 1: pass A = even bins of the zero-padded series, pass B = odd bins).  A pass is
//   S1  one radix-R0 butterfly per thread (thread u: rows u + 512 j of u_c, formed while the
//       rows are read), output q scaled by W_L^{u (2R q + c)} and written to LDS as sub-series q
//       (512 values, 8 KiB);
//   S2  512-point transforms of the R0 sub-series, ONE WAVE each, radix 8 x 8 x 8 with the data
//       of a lane in registers and two exchanges through the sub-series' own 8 KiB of LDS: no
//       workgroup barrier inside S2, so the waves of a SIMD drift apart and the LDS stores of
//       one hide under the arithmetic of the other; the second and third stage take their
//       twiddles BEFORE the butterfly, in tangent form (wf_dft8_tw); the last stage adds |.|^2
//       into the wave's register accumulators (bin s = q + R0 (a + 8 b + 64 cc) of the pass).
// A workgroup runs ONE pass; the 2R workgroups of a "tuple" (same XCD) walk the same units.
// n_frames <= 512 has its own wave-independent kernels (end of this file).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "fft_engine.hpp"

namespace ta {

#ifndef WF_NW_R0
#define WF_NW_R0 0  // experiment: plan R0 = WF_NW_R0 runs with WF_NW_VAL waves per workgroup
#define WF_NW_VAL 0
#endif

template <int LO, int HI, class F>
__device__ __forceinline__ void static_for_range(F&& f) {
    [&]<int... I>(std::integer_sequence<int, I...>) {
        (f(std::integral_constant<int, LO + I>{}), ...);
    }(std::make_integer_sequence<int, (HI > LO ? HI - LO : 0)>{});
}

typedef unsigned int wf_u32x4 __attribute__((ext_vector_type(4)));

#include "wfft_twist.inc"

// ---- first-stage DFTs that fft_engine.hpp does not have: radix 3 and the prime-factor
// butterflies 2x3, 4x3, 2x5, 4x5 (no internal twiddles) --------------------------------------------
// N = N1 N2 coprime: n = (N2 n1 + N1 n2) mod N, k1 = k mod N1, k2 = k mod N2:
// X[k] = sum W_N1^{n1 k1} W_N2^{n2 k2} x[n]
template <>
struct Dft<3> {
    static __device__ __forceinline__ void run(cd (&v)[3]) {
        constexpr double kS3 = 0.86602540378443864676372317075294;  // sin(2 pi / 3)
        const cd t = v[1] + v[2], d = v[1] - v[2];
        const cd m = cd{v[0].x - 0.5 * t.x, v[0].y - 0.5 * t.y};
        const cd e = cd{kS3 * d.y, -kS3 * d.x};  // -i sin(2 pi / 3) (v1 - v2)
        v[0] = v[0] + t;
        v[1] = m + e;
        v[2] = m - e;
    }
};

// odd N by its conjugate-symmetric pairs: with s_j = x_j + x_{N-j}, d_j = x_j - x_{N-j} (j <= N/2)
// X_k, X_{N-k} = x_0 + sum_j cos(2 pi j k / N) s_j  -/+  i sum_j sin(2 pi j k / N) d_j
template <int N>
struct DftOddTables;
template <>
struct DftOddTables<7> {
    static constexpr double c[3] = {0.62348980185873353053, -0.22252093395631440429, -0.90096886790241912624};
    static constexpr double s[3] = {0.78183148246802980871, 0.97492791218182360702, 0.43388373911755812048};
};
template <>
struct DftOddTables<9> {
    static constexpr double c[4] = {0.76604444311897803520, 0.17364817766693034885, -0.5, -0.93969262078590838405};
    static constexpr double s[4] = {0.64278760968653932632, 0.98480775301220805937, 0.86602540378443864676,
                                    0.34202014332566873304};
};
template <int N>
__device__ __forceinline__ void dft_odd(cd (&v)[N]) {
    constexpr int H = N / 2;
    using Tb = DftOddTables<N>;
    cd sp[H], dm[H];
#pragma unroll
    for (int j = 1; j <= H; ++j) {
        sp[j - 1] = v[j] + v[N - j];
        dm[j - 1] = v[j] - v[N - j];
    }
    const cd x0 = v[0];
    cd sum = x0;
#pragma unroll
    for (int j = 0; j < H; ++j) sum = sum + sp[j];
    v[0] = sum;
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        double rx = x0.x, ry = x0.y, ix = 0.0, iy = 0.0;
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            const int m = (j * k) % N;                      // cos / sin of 2 pi m / N by its mirror m <= N/2
            const double cm = m == 0 ? 1.0 : Tb::c[(m <= H ? m : N - m) - 1];
            const double sm = m == 0 ? 0.0 : m <= H ? Tb::s[m - 1] : -Tb::s[N - m - 1];
            rx = fma(cm, sp[j - 1].x, rx);
            ry = fma(cm, sp[j - 1].y, ry);
            ix = fma(sm, dm[j - 1].x, ix);
            iy = fma(sm, dm[j - 1].y, iy);
        }
        // -i (ix + i iy) = iy - i ix
        v[k] = cd{rx + iy, ry - ix};
        v[N - k] = cd{rx - iy, ry + ix};
    }
}
template <>
struct Dft<7> {
    static __device__ __forceinline__ void run(cd (&v)[7]) { dft_odd<7>(v); }
};
template <>
struct Dft<9> {
    static __device__ __forceinline__ void run(cd (&v)[9]) { dft_odd<9>(v); }
};

// 2 x N2 prime-factor butterflies for odd N2 (as Dft<10>)
template <int N2>
__device__ __forceinline__ void dft_2x(cd (&v)[2 * N2]) {
    constexpr int N = 2 * N2;
    cd s[2][N2];
#pragma unroll
    for (int j2 = 0; j2 < N2; ++j2) {
        const cd a = v[(2 * j2) % N], b = v[(N2 + 2 * j2) % N];
        s[0][j2] = a + b;
        s[1][j2] = a - b;
    }
    Dft<N2>::run(s[0]);
    Dft<N2>::run(s[1]);
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = s[q % 2][q % N2];
}
template <>
struct Dft<14> {
    static __device__ __forceinline__ void run(cd (&v)[14]) { dft_2x<7>(v); }
};
template <>
struct Dft<18> {
    static __device__ __forceinline__ void run(cd (&v)[18]) { dft_2x<9>(v); }
};

template <>
struct Dft<6> {
    static __device__ __forceinline__ void run(cd (&v)[6]) {
        cd s[2][3];
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
            const cd a = v[(2 * j2) % 6], b = v[(3 + 2 * j2) % 6];
            s[0][j2] = a + b;
            s[1][j2] = a - b;
        }
        Dft<3>::run(s[0]);
        Dft<3>::run(s[1]);
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = s[q % 2][q % 3];
    }
};

template <>
struct Dft<12> {
    static __device__ __forceinline__ void run(cd (&v)[12]) {
        cd s[4][3];
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
            cd t[4] = {v[(4 * j2) % 12], v[(3 + 4 * j2) % 12], v[(6 + 4 * j2) % 12], v[(9 + 4 * j2) % 12]};
            Dft<4>::run(t);
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) s[k1][j2] = t[k1];
        }
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) Dft<3>::run(s[k1]);
#pragma unroll
        for (int q = 0; q < 12; ++q) v[q] = s[q % 4][q % 3];
    }
};

template <>
struct Dft<10> {
    static __device__ __forceinline__ void run(cd (&v)[10]) {
        cd s[2][5];
#pragma unroll
        for (int j2 = 0; j2 < 5; ++j2) {
            const cd a = v[(2 * j2) % 10], b = v[(5 + 2 * j2) % 10];
            s[0][j2] = a + b;
            s[1][j2] = a - b;
        }
        Dft<5>::run(s[0]);
        Dft<5>::run(s[1]);
#pragma unroll
        for (int q = 0; q < 10; ++q) v[q] = s[q % 2][q % 5];
    }
};

template <>
struct Dft<20> {
    static __device__ __forceinline__ void run(cd (&v)[20]) {
        cd s[4][5];
#pragma unroll
        for (int j2 = 0; j2 < 5; ++j2) {
            cd t[4] = {v[(4 * j2) % 20], v[(5 + 4 * j2) % 20], v[(10 + 4 * j2) % 20], v[(15 + 4 * j2) % 20]};
            Dft<4>::run(t);
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) s[k1][j2] = t[k1];
        }
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) Dft<5>::run(s[k1]);
#pragma unroll
        for (int q = 0; q < 20; ++q) v[q] = s[q % 4][q % 5];
    }
};

// Twiddle table of plan (R0, R) (host side; shared by the library and tools/wfft), L = 2 R M:
//   [0, L)            W_L^n = exp(-2 pi i n / L)
//   [L, L + 896)      wave-local stage twiddles of the inverse kernel [14][64]: rows 0..6
//                     W_512^{lane (r+1)}, rows 7..13 W_64^{(lane & 7) (r-6)}
//   [L + 896, L + 1408)  forward kernel, pre-twiddled radix-8 stages in tangent form [8][64]: entry
//                     (tau, c) stands for the factor w = c (1 + i tau).  Rows 0..3: w = g^4, g^2, g,
//                     g W_8 with g = W_64^{lane >> 3} (second stage); rows 4..7 the same with
//                     g = W_512^{8 (lane & 7) + (lane >> 3)} (third stage).
constexpr int kWfStageRows = 14 + 8;  // table rows of 64 entries behind W_L^n
inline size_t wf_table_elems(int R0, int R) { return 2 * (size_t)R * R0 * 512 + kWfStageRows * 64; }
inline cd wf_tan_form(long num, long den) {  // exp(-2 pi i num / den) as (tau, c)
    const long double pi = 3.141592653589793238462643383279502884L;
    num %= den;
    if (4 * num == den) return cd{-0x1p200, 0x1p-200};      // -i: c (1 + i tau) = 2^-200 - i, exact to 2^-200
    if (4 * num == 3 * den) return cd{0x1p200, 0x1p-200};   // +i
    if (num == 0) return cd{0.0, 1.0};
    if (2 * num == den) return cd{0.0, -1.0};
    const long double h = 2.0L * pi * (long double)num / (long double)den;
    const long double c = cosl(h), sn = -sinl(h);
    return cd{(double)(sn / c), (double)c};
}
inline void wf_fill_table(int R0, int R, cd* a) {
    const long L = 2L * R * R0 * 512;
    const long double pi = 3.141592653589793238462643383279502884L;
    for (long n = 0; n < L; ++n) {
        if (n == 0) a[n] = cd{1.0, 0.0};
        else if (2 * n == L) a[n] = cd{-1.0, 0.0};
        else if (4 * n == L) a[n] = cd{0.0, -1.0};
        else if (4 * n == 3 * L) a[n] = cd{0.0, 1.0};
        else {
            const long double h = 2.0L * pi * (long double)n / (long double)L;
            a[n] = cd{(double)cosl(h), (double)-sinl(h)};
        }
    }
    for (int r = 1; r < 8; ++r)
        for (int l = 0; l < 64; ++l) {
            a[L + (r - 1) * 64 + l] = a[(L / 512) * l * r];       // W_512^{l r}
            a[L + (6 + r) * 64 + l] = a[(L / 64) * (l & 7) * r];  // W_64^{(l & 7) r}
        }
    cd* t = a + L + 14 * 64;
    for (int l = 0; l < 64; ++l) {
        const long gb = 8 * (l >> 3), gc = 8 * (l & 7) + (l >> 3);  // g = W_512^gb, W_512^gc
        const long g[2] = {gb, gc};
        for (int st = 0; st < 2; ++st) {
            t[(4 * st + 0) * 64 + l] = wf_tan_form(4 * g[st], 512);
            t[(4 * st + 1) * 64 + l] = wf_tan_form(2 * g[st], 512);
            t[(4 * st + 2) * 64 + l] = wf_tan_form(g[st], 512);
            t[(4 * st + 3) * 64 + l] = wf_tan_form(g[st] + 64, 512);  // g W_8
        }
    }
}

template <int R0_>
struct WPlan {
    static constexpr int R0 = R0_;
    static constexpr int N1 = 512;          // sub-series length = one wave's transform
    // forward kernel: with fewer than 8 sub-series, 8 waves would mostly idle through S2 (and all
    // land on the same SIMDs): small plans run in small workgroups, several per compute unit, and a
    // thread runs K1 of the 512 first-stage butterflies.  Waves per workgroup by same-box A/B
    // (tools/wfft, 24 GB of input): R0 = 2: 2 (9.3 ms against 18.9 ms with 8), R0 = 3: 2 (3.0 vs
    // 3.6 ms with 3, 3.4 with 1), R0 = 4: 2 (3.78 vs 3.98 ms with 4), R0 = 5: 4 (10.4 vs 11.5 with 8,
    // 12.5 with 5), R0 = 6: 4 (6.1 vs 6.7 with 6, 8.7 with 3), R0 = 7: 4 (5.10 vs 5.17 with 8, per
    // 12 GB); from R0 = 8 on 8 waves (R0 = 9 with 3 / 4: 8.2 / 5.7 vs 5.0; R0 = 14 with 7: 9.6 vs 5.0;
    // R0 = 18 with 6: 19.9 vs 4.8; R0 = 8 with 4:
    // 8.3 vs 7.6 ms; R0 = 10 with 4 / 5: 10.8 / 14.3 vs 9.7; R0 = 12 with 4 / 6: 13.4 / 19.4 vs 11.8;
    // R0 = 20 with 4: 11.6 vs 9.2).
    static constexpr int NW = R0 == WF_NW_R0 ? WF_NW_VAL : R0 <= 4 ? 2 : R0 <= 7 ? 4 : 8;
    static constexpr int NT = 64 * NW;
    static constexpr int K1 = (N1 + NT - 1) / NT;
    // (measured: trading the resident stage twiddles for a fourth wave per SIMD in the small plans
    // costs 5-10 %: 1000 frames 9.96 vs 9.26 ms, 2000 frames 10.56 vs 9.62 ms per 24 GB)
    // Row requests of the next unit spread over S2 instead of one burst behind it (the burst: 20
    // requests of 1 KiB per wave, 16 cycles each in the texture addresser, queue up in front of the
    // barrier).  Measured per plan, same box (tools/wfft, -DWF_SPREAD_ALL=0/1): pays where ONE
    // 8-wave workgroup owns the compute unit and two sub-series plus all row registers fit;
    // the plans with several workgroups per unit overlap across workgroups already and lose
    // 3-6 %; R0 = 18, 20 have no registers for it (three sub-series in flight, 96 registers).
#ifndef WF_SPREAD_ALL
    static constexpr bool kSpreadRows = R0 >= 9 && R0 <= 16;
#else
    static constexpr bool kSpreadRows = WF_SPREAD_ALL && (R0 < 18 || R0 == WF_NW_R0);  // (a 4-wave experiment plan has 512 registers)
#endif
    // One of two (three) sub-series in flight takes its first exchange through the register file
    // (gfx950 permlane swaps + DPP) instead of the LDS: same-box A/B per 30000 pairs R0 = 16: 1.63
    // -> 1.575 ms, R0 = 18: 1.91 -> 1.84, R0 = 20: 2.07 -> 2.02 ms; both sub-series that way: R0 = 16 +-0,
    // R0 = 20 +5 % (spills).  (-DWF_REG_EXCHANGE=0/1 overrides.)
#ifndef WF_REG_EXCHANGE
    static constexpr bool kRegExchange = R0 >= 16;
#else
    static constexpr bool kRegExchange = WF_REG_EXCHANGE;
#endif
    // ... and where a wave runs its sub-series one at a time (one full slot per wave), every one of
    // them, on the plans where the same-box A/B says so (per ~24 GB: R0 = 2 -4 %, 7 -2 %, 9 -1.5 %,
    // 10 -2.4 %, 12 -1.5..-3 %, 14 -3.7 %; R0 = 3..6 +-1 %, R0 = 8 +5..13 %: two workgroups per unit
    // there, the vector pipe is the busier one)
#ifndef WF_REG_EXCHANGE_SINGLE
    static constexpr bool kRegExchangeSingle = R0 == 2 || R0 == 7 || R0 == 9 || R0 == 10 || R0 == 12 || R0 == 14;
#else
    static constexpr bool kRegExchangeSingle = (WF_REG_EXCHANGE_SINGLE >> (R0 - 1)) & 1;  // bit mask over R0
#endif
#ifndef WF_MINW_R0
    // R0 = 2 lives on four waves per SIMD -- several workgroups per compute unit -- and sits just
    // under the 128 registers that takes: hold it there
    static __host__ __device__ constexpr int min_waves(bool /*byp*/, bool lng) { return R0 == 2 ? 4 : 1; }
#else  // experiment: plan R0 = WF_MINW_R0 is compiled for WF_MINW_VAL waves per SIMD
    static __host__ __device__ constexpr int min_waves(bool, bool) { return R0 == WF_MINW_R0 ? WF_MINW_VAL : 1; }
#endif
    static constexpr bool kTwResident = true;
    static constexpr int M = R0 * N1;
    // sub-series of a wave (forward kernel): consecutive ones, q = sub_base(wave) + s, s < sub_count(wave);
    // the first R0 % NW waves own one more than the others, NS1 at most
    static constexpr int NS1 = (R0 + NW - 1) / NW;
    static constexpr int NLO = R0 / NW, REM = R0 % NW;
    static __host__ __device__ constexpr int sub_count(int wave) { return NLO + (wave < REM ? 1 : 0); }
    static __host__ __device__ constexpr int sub_base(int wave) { return wave * NLO + (wave < REM ? wave : REM); }
    static constexpr size_t kLds = (size_t)M * sizeof(cd);

    // ---- by-particle mode, a unit that is a single REAL column (z of an even atom, x of an odd one):
    // its transform is Hermitian, Z[L - k] = conj Z[k], so |Z|^2 of bin (pass c, sub-series q,
    // position r) equals that of its mirror: pass 0: (0, R0 - q, 511 - r) (q = 0: (0, 0, 512 - r));
    // pass c > 0: (2R - c, R0 - 1 - q, 511 - r).  The lag values only see the EVEN part of the power
    // spectrum (lag[n] = sum_k P[k] cos(2 pi k n / L)), so of every mirror pair of sub-series ONE is
    // transformed, with weight 2, and the other not at all (a sub-series that is its own mirror:
    // weight 1): half the second-stage work and half the first-stage stores of such a unit.  Which
    // member of a pair is taken is chosen per plan so that the waves (and the SIMDs: waves w, w + 4)
    // share the work evenly; every pass c > 0 takes the same set, which is then also the
    // complement of the mirror image of what pass 2R - c takes.
    struct RealSel {
        unsigned qmask[2];  // [pass > 0]: bit q = sub-series q is transformed
        unsigned q1mask[2];  // ... with weight 1 (its own mirror) instead of 2
    };
    static __host__ __device__ constexpr int owner(int q) {
        for (int w = 0; w < NW; ++w)
            if (q >= sub_base(w) && q < sub_base(w) + sub_count(w)) return w;
        return 0;
    }
    static __host__ __device__ constexpr RealSel real_selection() {
        RealSel r{{0u, 0u}, {0u, 0u}};
        for (int t = 0; t < 2; ++t) {
            int load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, simd[4] = {0, 0, 0, 0};
            auto take = [&](int q, bool self) {
                r.qmask[t] |= 1u << q;
                if (self) r.q1mask[t] |= 1u << q;
                ++load[owner(q)];
                ++simd[owner(q) % 4];
            };
            const int span = t == 0 ? R0 : R0 - 1;  // mirror of q: (span - q) mod R0
            // the sub-series that are their own mirror
            for (int q = 0; q < R0; ++q)
                if ((span - q + R0) % R0 == q) take(q, true);
            for (int q = 0; q < R0; ++q) {
                const int m = (span - q + R0) % R0;
                if (m <= q) continue;  // each pair once (q < m)
                const int wa = owner(q), wb = owner(m);
                const int ca = 16 * load[wa] + simd[wa % 4], cb = 16 * load[wb] + simd[wb % 4];
                take(cb < ca ? m : q, false);
            }
        }
        return r;
    }
    static constexpr RealSel kRealSel = real_selection();
};

__device__ __forceinline__ cd wf_load(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uni_off) {
    // rows past the end of the series read as zero: the buffer's bounds check IS the padding.
    // lane_off: per-lane byte offset (VGPR), uni_off: wave-uniform byte offset (SGPR operand)
    const wf_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uni_off, 0);
    return __builtin_bit_cast(cd, v);
}

// (post-twiddle form, kept for the 512-frame kernels at the end of this file)
// One wave's 512-point DIF transform of the sub-series at `reg` (natural order in, the lane's
// eight outputs c = 0..7 are bins s = (lane>>3) + 8 (lane&7) + 64 c), |.|^2 added to acc.
// The two exchanges stay inside the sub-series' own 8 KiB; their layouts are conflict-free for
// ds_write_b128 / ds_read_b128 (verified exhaustively against the lane groups of the LDS):
//   exchange 1: (a, l)      at a*64 + (l ^ 8 (a&1))
//   exchange 2: (a, b, n0)  at (8a + b)*8 + (n0 ^ a ^ b)
// A wave's own DS operations execute in order, so a read issued after a write of the same wave
// sees it: no barrier, only the compiler has to keep the order (wave_barrier).
struct WfSubPost {
    int lane, hi, lo;
    __device__ __forceinline__ explicit WfSubPost(int lane_) : lane(lane_) {
        // LDS addresses depend on the lane only: re-formed per call (a few integer
        // operations), otherwise they are hoisted out of the pair loop and spilled
        asm volatile("" : "+v"(lane));
        hi = lane >> 3;
        lo = lane & 7;
    }
    __device__ __forceinline__ void read_a(const cd* __restrict__ reg, cd (&v)[8]) const {
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) v[n2] = reg[64 * n2 + lane];
    }
    __device__ __forceinline__ void stage_a(cd* __restrict__ reg, cd (&v)[8], const cd (&twa)[7]) const {
        Dft<8>::run(v);
#pragma unroll
        for (int a = 1; a < 8; ++a) v[a] = cmul(v[a], twa[a - 1]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int a = 0; a < 8; ++a) reg[a * 64 + (lane ^ (8 * (a & 1)))] = v[a];
        __builtin_amdgcn_wave_barrier();
        const int base = hi * 64 + (lo ^ (8 * (hi & 1)));  // (8 n1 + lo) ^ 8 (hi&1), n1 = 0
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) v[n1] = reg[base ^ (8 * n1)];
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void stage_b(cd* __restrict__ reg, cd (&v)[8], const cd (&twb)[7]) const {
        Dft<8>::run(v);
#pragma unroll
        for (int b = 1; b < 8; ++b) v[b] = cmul(v[b], twb[b - 1]);
        __builtin_amdgcn_wave_barrier();
        const int x = lo ^ hi;
#pragma unroll
        for (int b = 0; b < 8; ++b) reg[(hi * 8 + b) * 8 + (x ^ b)] = v[b];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n0 = 0; n0 < 8; ++n0) v[n0] = reg[lane * 8 + (n0 ^ x)];
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void stage_c(cd (&v)[8], double (&acc)[8]) const {
        Dft<8>::run(v);
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = fma(v[c].y, v[c].y, fma(v[c].x, v[c].x, acc[c]));
        __builtin_amdgcn_wave_barrier();
    }
};

// ---- pre-twiddled radix-8 butterfly in tangent form ---------------------------------------------
// y[b] = sum_k x[k] g^k W_8^{k b}: three levels of radix-2 butterflies A +- w B whose factor
// w = c (1 + i tau) (times -i where the level asks for it) costs 2 FMAs for q = (1 + i tau) B and
// 4 for A +- c q: 72 FMAs per butterfly, twiddles included, against 52 + 28 for an untwiddled
// butterfly followed by seven complex multiplications; and 8 constants per lane instead of 14.
// ROT = 1: the factor is -i w.
template <int ROT>
__device__ __forceinline__ void wf_bf(cd& A, cd& B, const cd t) {
    const double qx = fma(-t.x, B.y, B.x), qy = fma(t.x, B.x, B.y);
    const cd a = A;
    if constexpr (ROT == 0) {
        A = cd{fma(t.y, qx, a.x), fma(t.y, qy, a.y)};
        B = cd{fma(-t.y, qx, a.x), fma(-t.y, qy, a.y)};
    } else {  // -i q = (qy, -qx)
        A = cd{fma(t.y, qy, a.x), fma(-t.y, qx, a.y)};
        B = cd{fma(-t.y, qy, a.x), fma(t.y, qx, a.y)};
    }
}
// t[0..3] = tangent forms of g^4, g^2, g, g W_8; natural order in and out
__device__ __forceinline__ void wf_dft8_tw(cd (&v)[8], const cd (&t)[4]) {
    wf_bf<0>(v[0], v[4], t[0]);
    wf_bf<0>(v[1], v[5], t[0]);
    wf_bf<0>(v[2], v[6], t[0]);
    wf_bf<0>(v[3], v[7], t[0]);
    wf_bf<0>(v[0], v[2], t[1]);  // E0, E2
    wf_bf<0>(v[1], v[3], t[1]);  // O0, O2 (pending g)
    wf_bf<1>(v[4], v[6], t[1]);  // E1, E3
    wf_bf<1>(v[5], v[7], t[1]);  // O1, O3 (pending g)
    wf_bf<0>(v[0], v[1], t[2]);  // y0, y4
    wf_bf<1>(v[2], v[3], t[2]);  // y2, y6
    wf_bf<0>(v[4], v[5], t[3]);  // y1, y5
    wf_bf<1>(v[6], v[7], t[3]);  // y3, y7
    const cd y4 = v[1], y6 = v[3], y1 = v[4], y3 = v[6];
    v[1] = y1, v[3] = y3, v[4] = y4, v[6] = y6;
}

// One wave's 512-point transform of the sub-series at `reg` (natural order in, the lane's eight
// outputs c = 0..7 are bins s = (lane>>3) + 8 (lane&7) + 64 c), |.|^2 added to acc.  Index
// n = 64 n2 + 8 n1 + n0, bin s = a + 8 b + 64 c:
//   stage a: radix 8 over n2 (no twiddle), lane = (n1, n0)        -> exchange 1 -> lane (a, n0)
//   stage b: radix 8 over n1, pre-twiddle (W_64^a)^n1             -> exchange 2 -> lane (a, b)
//   stage c: radix 8 over n0, pre-twiddle (W_512^{a + 8 b})^n0  (the factor W_512^{a n0} left
//            over from stage b's twiddle W_512^{a (8 n1 + n0)} and W_64^{b n0} in one)
// The two exchanges stay inside the sub-series' own 8 KiB; their layouts are conflict-free for
// ds_write_b128 / ds_read_b128 (verified exhaustively against the lane groups of the LDS):
//   exchange 1: (a, l)      at a*64 + (l ^ 8 (a&1))
//   exchange 2: (a, b, n0)  at (8a + b)*8 + (n0 ^ a ^ b)
// A wave's own DS operations execute in order, so a read issued after a write of the same wave
// sees it: no barrier, only the compiler has to keep the order (wave_barrier).
struct WfTw {
    cd b[4], c[4];  // tangent-form constants of the second and third stage (table rows 14..21)
};
// Register-file transposition between the register index (3 bits) and lane bits 3..5 of eight
// complex values per lane (gfx950 cross-lane swaps; experiment, see WfSub::stage_a).
typedef unsigned wf_u32pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wf_transpose_hi(cd (&v)[8]) {
    unsigned w[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const wf_u32x4 t = __builtin_bit_cast(wf_u32x4, v[a]);
        w[a][0] = t.x, w[a][1] = t.y, w[a][2] = t.z, w[a][3] = t.w;
    }
    // register bit 2 <-> lane bit 5
#pragma unroll
    for (int a = 0; a < 8; ++a)
        if (!(a & 4))
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const wf_u32pair r = __builtin_amdgcn_permlane32_swap(w[a][d], w[a | 4][d], false, false);
                w[a][d] = r.x, w[a | 4][d] = r.y;
            }
    // register bit 1 <-> lane bit 4
#pragma unroll
    for (int a = 0; a < 8; ++a)
        if (!(a & 2))
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const wf_u32pair r = __builtin_amdgcn_permlane16_swap(w[a][d], w[a | 2][d], false, false);
                w[a][d] = r.x, w[a | 2][d] = r.y;
            }
    // register bit 0 <-> lane bit 3
#pragma unroll
    for (int a = 0; a < 8; ++a)
        if (!(a & 1))
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned lo = w[a][d], hi = w[a | 1][d];
                w[a][d] = __builtin_amdgcn_update_dpp(lo, hi, 0x128, 0xF, 0xC, false);      // lanes with bit 3 set
                w[a | 1][d] = __builtin_amdgcn_update_dpp(hi, lo, 0x128, 0xF, 0x3, false);  // lanes with bit 3 clear
            }
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        wf_u32x4 t;
        t.x = w[a][0], t.y = w[a][1], t.z = w[a][2], t.w = w[a][3];
        v[a] = __builtin_bit_cast(cd, t);
    }
}

// LDS byte addresses of a wave's first sub-series, formed once per launch: every access of S2 is
// one of these registers plus an instruction offset (the wave's sub-series s sits s * 8 KiB
// further on).  Exchange 1 writes a*64 + (l ^ 8 (a&1)): base `we` for even a, `wo` for odd a;
// it reads (hi*64 + lo) + 8 (n1 ^ (hi&1)): base `re` for even n1, `ro` for odd n1.  Exchange 2's
// swizzle n0 ^ a ^ b costs one integer operation per element.
struct WfAddr {
    // x16 = 16 (hi ^ lo); exchange 2's bases: b2w = q_bytes + 1024 hi, b2r = b2w + 128 lo
    unsigned we, wo, re, ro, x16, b2w, b2r;
    __device__ __forceinline__ void init(int lane, unsigned q_bytes) {
        const int hi = lane >> 3, lo = lane & 7;
        we = q_bytes + 16u * (unsigned)lane;
        wo = q_bytes + 16u * (unsigned)(lane ^ 8);
        const unsigned rb = q_bytes + 16u * (unsigned)(hi * 64 + lo);
        re = rb + 128u * (unsigned)(hi & 1);
        ro = rb - 128u * (unsigned)(hi & 1);
        x16 = 16u * (unsigned)(hi ^ lo);
        b2w = q_bytes + 1024u * (unsigned)hi;
        b2r = b2w + 128u * (unsigned)lo;
    }
    // exchange 2, element k of the lane: written at (8 hi + k) 8 + (x ^ k), read at (8 hi + lo) 8 + (k ^ x):
    // one xor-add per element, formed where it is used (the same register serves all the wave's
    // sub-series, which differ by an instruction offset); 128 k goes into the instruction offset
    __device__ __forceinline__ unsigned w2(int k) const { return (x16 ^ (16u * (unsigned)k)) + b2w; }
    __device__ __forceinline__ unsigned r2(int k) const { return (x16 ^ (16u * (unsigned)k)) + b2r; }
};
struct WfSub {
    const WfAddr& ad;
    unsigned char* smem;
    __device__ __forceinline__ WfSub(const WfAddr& ad_, unsigned char* smem_) : ad(ad_), smem(smem_) {}
    __device__ __forceinline__ cd& at(unsigned base, unsigned off) const {
        // every element of a sub-series sits on a 16-byte boundary; said out loud, because `cd` only
        // promises 8 and the compiler otherwise turns part of the accesses of the three-in-flight
        // path (R0 = 18, 20) into ds_read2_b64, which the exchange layouts are not conflict-free for
        // (SQ_LDS_BANK_CONFLICT 3.8e8 of 2.1e9 LDS cycles per 24 GB at R0 = 20; none at R0 <= 16)
        return *reinterpret_cast<cd*>(__builtin_assume_aligned(smem + (base + off), 16));
    }
    // SO: byte offset of the sub-series behind the wave's first one
    template <unsigned SO>
    __device__ __forceinline__ void read_a(cd (&v)[8]) const {
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) v[n2] = at(ad.we, SO + 1024u * n2);
    }
    // each exchange in two halves: after the `_w` half the eight values live in LDS only, the
    // `_r` half brings the lane's next eight back -- whatever runs in between needs no register
    // of this sub-series
    template <unsigned SO>
    __device__ __forceinline__ void stage_a_w(cd (&v)[8]) const {
        Dft<8>::run(v);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int a = 0; a < 8; ++a) at(a & 1 ? ad.wo : ad.we, SO + 1024u * a) = v[a];
        __builtin_amdgcn_wave_barrier();
    }
    template <unsigned SO>
    __device__ __forceinline__ void stage_a_r(cd (&v)[8]) const {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) v[n1] = at(n1 & 1 ? ad.ro : ad.re, SO + 128u * n1);
        __builtin_amdgcn_wave_barrier();
    }
    // XV: exchange 1 (register index a <-> lane bits 3..5) in the register file instead of the
    // LDS -- three rounds of pairwise swaps: v_permlane32_swap (lane bit 5), v_permlane16_swap
    // (bit 4), v_mov_b32_dpp row_ror:8 under a bank mask (bit 3); 80 instructions, 112 issue slots
    // (the permlane swaps take two) against 8 stores + 8 loads
    template <unsigned SO, bool XV = false>
    __device__ __forceinline__ void stage_a(cd (&v)[8]) const {
        if constexpr (XV) {
            Dft<8>::run(v);
            wf_transpose_hi(v);
        } else {
            stage_a_w<SO>(v);
            stage_a_r<SO>(v);
        }
    }
    template <unsigned SO>
    __device__ __forceinline__ void stage_b_w(cd (&v)[8], const WfTw& tw) const {
        wf_dft8_tw(v, tw.b);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int b = 0; b < 8; ++b) at(ad.w2(b), SO + 128u * b) = v[b];
        __builtin_amdgcn_wave_barrier();
    }
    template <unsigned SO>
    __device__ __forceinline__ void stage_b_r(cd (&v)[8]) const {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n0 = 0; n0 < 8; ++n0) v[n0] = at(ad.r2(n0), SO);
        __builtin_amdgcn_wave_barrier();
    }
    template <unsigned SO>
    __device__ __forceinline__ void stage_b(cd (&v)[8], const WfTw& tw) const {
        stage_b_w<SO>(v, tw);
        stage_b_r<SO>(v);
    }
    __device__ __forceinline__ void stage_c(cd (&v)[8], const WfTw& tw, double (&acc)[8]) const {
        wf_dft8_tw(v, tw.c);
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = fma(v[c].y, v[c].y, fma(v[c].x, v[c].x, acc[c]));
    }
    // ... with a weight (1 or 2: the sub-series of a real column that stand for their mirror images too)
    __device__ __forceinline__ void stage_c_w(cd (&v)[8], const WfTw& tw, double (&acc)[8], double wgt) const {
        wf_dft8_tw(v, tw.c);
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = fma(wgt, fma(v[c].y, v[c].y, v[c].x * v[c].x), acc[c]);
    }
};

constexpr unsigned kWfSubBytes = 512 * sizeof(cd);  // one sub-series in LDS

// `hook(part)`, part = integral_constant 0..3, is called at four points spread over the wave's
// S2: the forward kernel requests a quarter of the next unit's rows at each (issued in one burst
// behind S2, the 20 requests of every wave queue up in front of the barrier: the texture
// addresser takes 16 cycles per 1-KiB request, 2.5k cycles per workgroup and unit)
struct WfNoHook {
    template <class T>
    __device__ __forceinline__ void operator()(T) const {}
};
template <int I>
using wf_part = std::integral_constant<int, I>;

// sub-series S0 of the wave (its first one is 0)
template <int S0, bool XV = false, class Hook = WfNoHook>
__device__ __forceinline__ void wf_sub512(const WfSub& w, const WfTw& tw, double (&acc)[8], Hook&& hook = Hook{}) {
    cd v[8];
    w.read_a<S0 * kWfSubBytes>(v);
    hook(wf_part<0>{});
    w.template stage_a<S0 * kWfSubBytes, XV>(v);
    hook(wf_part<1>{});
    w.stage_b<S0 * kWfSubBytes>(v, tw);
    hook(wf_part<2>{});
    w.stage_c(v, tw, acc);
    hook(wf_part<3>{});
}

// ... with the weight of WfSub::stage_c_w (by-particle mode, real columns)
template <int S0, bool XV = false>
__device__ __forceinline__ void wf_sub512_w(const WfSub& w, const WfTw& tw, double (&acc)[8], double wgt) {
    cd v[8];
    w.read_a<S0 * kWfSubBytes>(v);
    w.template stage_a<S0 * kWfSubBytes, XV>(v);
    w.stage_b<S0 * kWfSubBytes>(v, tw);
    w.stage_c_w(v, tw, acc, wgt);
}

// Two sub-series (S0, S0 + 1) interleaved: while one's butterflies run, the other's exchange
// (eight 16-byte stores, eight loads, ~300 cycles of LDS round trip) is in flight.
// XV: the second sub-series takes exchange 1 through the register file (WfSub::stage_a) while
// the first one's is in the LDS pipe: one sixth of the wave's LDS exchange traffic becomes vector
// work (WPlan::kRegExchange; both of them that way is slower: the vector pipe becomes the limit)
template <int S0, bool XV = false, class Hook = WfNoHook>
__device__ __forceinline__ void wf_sub512_x2(const WfSub& w, const WfTw& tw, double (&acc0)[8],
                                             double (&acc1)[8], Hook&& hook = Hook{}) {
    constexpr unsigned A = S0 * kWfSubBytes, B = A + kWfSubBytes;
    cd v0[8], v1[8];
    w.read_a<A>(v0);
    w.read_a<B>(v1);
    __builtin_amdgcn_wave_barrier();
    hook(wf_part<0>{});
    w.template stage_a<A>(v0);
    w.template stage_a<B, XV>(v1);
    hook(wf_part<1>{});
    w.stage_b<A>(v0, tw);
    w.stage_b<B>(v1, tw);
    hook(wf_part<2>{});
    w.stage_c(v0, tw, acc0);
    w.stage_c(v1, tw, acc1);
    hook(wf_part<3>{});
}

// Three sub-series interleaved (the waves that own one more than the others: they then finish
// together with their SIMD partner's two instead of running a third alone).
template <bool XV = false, class Hook = WfNoHook>
__device__ __forceinline__ void wf_sub512_x3(const WfSub& w, const WfTw& tw, double (&acc0)[8],
                                             double (&acc1)[8], double (&acc2)[8], Hook&& hook = Hook{}) {
    constexpr unsigned A = 0, B = kWfSubBytes, C = 2 * kWfSubBytes;
    cd v0[8], v1[8], v2[8];
    w.read_a<A>(v0);
    w.read_a<B>(v1);
    w.read_a<C>(v2);
    __builtin_amdgcn_wave_barrier();
    hook(wf_part<0>{});
    w.template stage_a<A>(v0);
    w.template stage_a<B>(v1);
    w.template stage_a<C, XV>(v2);
    hook(wf_part<1>{});
    w.stage_b<A>(v0, tw);
    w.stage_b<B>(v1, tw);
    w.stage_b<C>(v2, tw);
    hook(wf_part<2>{});
    w.stage_c(v0, tw, acc0);
    w.stage_c(v1, tw, acc1);
    w.stage_c(v2, tw, acc2);
    hook(wf_part<3>{});
}

// ================================================================================================
// pm: pair-major slab, pair p at pm + p*pitch*2 doubles; T rows are valid, the rest of the
// transform length is zero padding.  tw2: the plan's table (wf_fill_table): W_L^n, n < L, then
// the stage constants.  accg: partial spectra (see below).
//
// Per unit: S1 (rows -> radix-R0 butterfly -> output twiddles -> LDS), barrier, S2 (the wave's
// sub-series, |.|^2 into its accumulators), barrier.  The rows of the NEXT unit are requested
// during S2 -- spread over it where the plan has the registers (WPlan::kSpreadRows), else in one
// burst behind it -- and land while the slowest wave finishes.  The tangent-form constants of the
// second and third radix-8 stage (32 registers per lane) stay resident for the whole launch.
// Forward kernel: |transform|^2 of column units, accumulated per pass in registers.
//
// grid: a multiple of 16 R blocks, all resident.  Block b: XCD x = b & 7 (round-robin placement,
// speed only), i = b >> 3, pass c = i mod 2R, tuple (b & 7) + 8 (i / 2R).  The 2R blocks of a tuple
// sit on one XCD and walk the same units at the same pace, so the later readers of a row find it
// in that XCD's L2 and every input byte leaves HBM once.  (Both passes in one workgroup would
// read a pair twice half a pair-period apart: with 32 compute units x 160 KiB in flight per
// 4 MiB L2 the second read mostly misses -- measured 1.8x the algorithmic bytes.)  One pass per
// workgroup also halves the accumulators per thread (24 per lane at R0 = 20), which is what lets
// the next unit's rows, the RESIDENT stage twiddles and two or three sub-series in flight fit
// 256 registers without spilling.
//
// Lag-sum mode (BYP = false): units are the n_units column pairs of the slab, tuple t takes pairs
//   t, t + n_tuples, ...; at the end the pass's accumulators go to accg[t][c][q][cc / 2][lane][cc & 1]
//   (a row of L doubles per tuple, summed over tuples afterwards).
// By-particle mode (BYP = true): units are the atoms' column units (wf_unit_of); a tuple takes
//   whole atoms (pairs of adjacent atoms when the number of columns per atom is odd), and after
//   an atom's last unit the pass's accumulators -- the atom's power spectrum in this pass -- go
//   to accg[atom][c][...] in the same order and start again from zero.
// Either way a spectrum is L doubles, [pass][q][cc / 2][lane][cc & 1], as k_winverse reads them back.
__device__ __forceinline__ void wf_unit_of(long atom, int k, int D, long* pair, int* kind) {
    // kind 2 = both columns of the pair (complex series), 0 / 1 = only that half (real series)
    const long c0 = atom * D;
    if (D == 2) {
        *pair = atom, *kind = 2;
    } else if (D == 1) {
        *pair = c0 >> 1, *kind = (int)(c0 & 1);
    } else if ((c0 & 1) == 0) {  // even first column: (x, y) aligned, then z alone
        *pair = (c0 >> 1) + k, *kind = k == 0 ? 2 : 0;
    } else {                     // x alone (second half of a pair), then (y, z) aligned
        *pair = (c0 >> 1) + k, *kind = k == 0 ? 1 : 2;
    }
}

typedef unsigned int wf_u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned int))));

__device__ __forceinline__ cd wf_cfma(cd acc, cd a, cd b) {  // acc + a b
    return {acc.x + (a.x * b.x - a.y * b.y), acc.y + (a.x * b.y + a.y * b.x)};
}

template <bool S, class A, class B>
__device__ __forceinline__ auto& wf_pick(A& a, B& b) {
    if constexpr (S) return a;
    else return b;
}

// SRC32: the pair-major slab holds float32 elements (8-byte rows (x[t], y[t]): "stage_device_f32",
// what MDAnalysis data is at the source): half the HBM bytes -- a third of this kernel's energy is
// the HBM read -- and half the registers for the next unit's rows, which
// is what lets R0 = 18, 20 spread their row requests over S2 like the smaller plans.  The rows are
// widened to float64 (exactly) when the first stage picks them up; everything after is the same
// arithmetic on the same values.  (Without an outer radix; longer trajectories are widened into a
// float64 scratch slab first.)
template <class P, bool BYP = false, bool LONG = false, bool STAMP = false, bool SRC32 = false>
__global__ void __launch_bounds__(P::NT, P::min_waves(BYP, LONG))
    k_wsplit_accum(const double* __restrict__ pm, long pitch, int T, long n_units,
                   const cd* __restrict__ tw2, double* __restrict__ accg, int D, int R_arg,
                   unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    constexpr int R0 = P::R0, N1 = P::N1, NW = P::NW, NS1 = P::NS1, M = P::M;
    static_assert(!(SRC32 && LONG), "float32 slabs: plans without an outer radix only");
    const int R = LONG ? R_arg : 1, npass = 2 * R, L = npass * M;
    const int tid = threadIdx.x, wave = tid >> 6;
    int lane = tid & 63;
    const int bi = blockIdx.x >> 3;
    const int pass = bi % npass;
    const long tuple = (blockIdx.x & 7) + 8 * (bi / npass), n_tuples = gridDim.x / npass;
    const int upa = BYP ? (D == 3 ? 2 : 1) : 1;  // units per atom (by-particle mode)
    // with an odd number of columns per atom, atoms 2i and 2i + 1 share a column pair (the last
    // column of one, the first of the other): a tuple takes both, so the shared rows come from
    // the L2 the second time
    const int grp = BYP && (D & 1) ? 2 : 1;

    double acc[NS1][8];
#pragma unroll
    for (int s = 0; s < NS1; ++s)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[s][c] = 0.0;
    // STAMP (diagnostic builds only): st_acc[0] / [1] = shader cycles in S1 / S2, [2] / [3] = the
    // kernel's whole span in shader cycles (s_memtime) and in 100 MHz ticks (s_memrealtime): their
    // quotient x 100 MHz is the clock the kernel ran at (MI355X_MICROARCH.md, DVFS item 6)
    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_prev = 0, st_t0 = 0, st_r0 = 0, st_tail = 0, st_s2end = 0;
    if constexpr (STAMP) {
        st_prev = st_t0 = __builtin_amdgcn_s_memtime();
        st_r0 = __builtin_amdgcn_s_memrealtime();
    }
#define WF_STAMP(i)                                                   \
    if constexpr (STAMP) {                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[i] += now_ - st_prev;                                  \
        st_prev = now_;                                               \
    }
    const __amdgpu_buffer_rsrc_t twr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<cd*>(tw2), 0, (L + kWfStageRows * 64) * 16, 0x00020000);
    // the tuple's units in order: lag-sum mode pairs tuple + i n_tuples; by-particle mode the
    // units k < upa of its atoms.  A unit past the end gets an empty buffer: its loads return
    // zeros and are never used.
    auto unit_rsrc = [&](long item, int k, int* kind) {
        long pair = item;
        const bool live = item < n_units;
        *kind = 2;
        if constexpr (BYP) wf_unit_of(live ? item : 0, k, D, &pair, kind);
        if constexpr (SRC32)
            return __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(reinterpret_cast<const float*>(pm) + (live ? pair : 0) * pitch * 2), 0,
                live ? T * 8 : 0, 0x00020000);
        else
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pm + (live ? pair : 0) * pitch * 2), 0,
                                                     live ? T * 16 : 0, 0x00020000);
    };
    // row u + 512 j + M jo of the unit (a single real column: that half of the row, imaginary part 0);
    // SRC32: the raw 8 bytes (one column: its 4 bytes in .x), widened by wf_widen when S1 starts
    using RowT = std::conditional_t<SRC32, wf_u32x2, cd>;
    auto load_row = [&](__amdgpu_buffer_rsrc_t rs, int kd, int u, unsigned row_off) -> RowT {
        if constexpr (SRC32) {
            // the whole 8-byte row also for a single column (wf_widen picks its half): 4-byte requests
            // fetch the same lines at half the rate
            return __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)u * 8u, row_off * 8u, 0);
        } else {
            if (!BYP || kd == 2) return wf_load(rs, (unsigned)u * 16u, row_off * 16u);
            return cd{__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                                                     rs, (unsigned)u * 16u + (unsigned)kd * 8u, row_off * 16u, 0)),
                      0.0};
        }
    };
    [[maybe_unused]] auto wf_widen = [&](wf_u32x2 r, int kd) {
        if (!BYP || kd == 2)
            return cd{(double)__builtin_bit_cast(float, (unsigned)r[0]), (double)__builtin_bit_cast(float, (unsigned)r[1])};
        return cd{(double)__builtin_bit_cast(float, (unsigned)(kd ? r[1] : r[0])), 0.0};
    };
    // thread tid runs the first-stage butterflies u = tid + NT k, k < K1 (u < 512)
    constexpr int K1 = P::K1, NT = P::NT;
    RowT xx[K1][R0];
    auto issue_loads = [&](__amdgpu_buffer_rsrc_t rs, int kd) {
#pragma unroll
        for (int k1 = 0; k1 < K1; ++k1) {
            const int u = tid + NT * k1;  // rows past 512 j + 511 belong to the next j: never loaded
            if (K1 * NT != N1 && u >= N1) continue;
            if (!BYP || kd == 2) {
#pragma unroll
                for (int j = 0; j < R0; ++j) xx[k1][j] = load_row(rs, 2, u, (unsigned)(N1 * j));
            } else {
#pragma unroll
                for (int j = 0; j < R0; ++j) xx[k1][j] = load_row(rs, kd, u, (unsigned)(N1 * j));
            }
        }
    };
#ifndef WF_LOAD_PARTS
#define WF_LOAD_PARTS 4  // 1: all of the next unit's rows in one burst behind S2 (as before round 3)
#endif
    // Rows with linear index i = k1 R0 + j.  Plans with kSpreadRows request them along S2, a quarter
    // at each hook point of the wave's first call; the others in one burst behind S2.
    constexpr int NR = K1 * R0;
    constexpr bool kSpread = P::kSpreadRows || (SRC32 && R0 >= 18);  // float32 rows: 2 registers each
#ifndef WF_SPREAD32_TAIL
#define WF_SPREAD32_TAIL 0  // float32 rows, R0 = 18, 20: this many of the rows stay behind S2.  Same-box A/B at
                            // 10000 x 100000 x 3 (float64 slab 8.69 ms): 20 (none along S2) 8.78 ms, 8: 8.30, 4: 8.24, 0: 7.85 ms
                            // (28 bytes of scratch and still the fastest)
#endif
    constexpr int NSPREAD = WF_LOAD_PARTS != 4 ? 0 : (kSpread ? (SRC32 && R0 >= 18 ? NR - WF_SPREAD32_TAIL : NR) : 0);
    auto issue_row = [&](auto ii, __amdgpu_buffer_rsrc_t rs, int kd) {
        constexpr int i = decltype(ii)::value, k1 = i / R0, j = i % R0;
        const int u = tid + NT * k1;
        if (K1 * NT != N1 && u >= N1) return;
        if (!BYP || kd == 2) xx[k1][j] = load_row(rs, 2, u, (unsigned)(N1 * j));
        else xx[k1][j] = load_row(rs, kd, u, (unsigned)(N1 * j));
    };
    auto issue_loads_part = [&](auto part_c, __amdgpu_buffer_rsrc_t rs, int kd) {
        constexpr int PART = decltype(part_c)::value;
        static_for_range<PART * NSPREAD / 4, (PART + 1) * NSPREAD / 4>([&](auto ii) { issue_row(ii, rs, kd); });
    };
    auto issue_loads_tail = [&](__amdgpu_buffer_rsrc_t rs, int kd) {
        static_for_range<NSPREAD, NR>([&](auto ii) { issue_row(ii, rs, kd); });
    };
    int kind = 2, nkind = 2;
    __amdgpu_buffer_rsrc_t crs = unit_rsrc(tuple * grp, 0, &kind);
    issue_loads(crs, kind);
    // the wave-local stage twiddles stay in registers for the whole launch where one pass's
    // accumulators leave room for them (14 fewer loads per wave and unit); the small plans trade
    // them for a fourth wave per SIMD and load them again before every S2
    WfTw stw;
    auto load_stage_tw = [&]() {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            stw.b[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(L + (14 + a) * 64) * 16u);
            stw.c[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(L + (18 + a) * 64) * 16u);
        }
    };
    // (the by-particle variant with an outer radix on the two largest plans loads them again after
    // every first stage instead: 32 registers it does not have while the outer rows are folded in)
    constexpr bool kTwRes = P::kTwResident && !(BYP && LONG && R0 >= 18);
    if constexpr (kTwRes) load_stage_tw();
    WfAddr wad;
    wad.init(lane, (unsigned)P::sub_base(wave) * kWfSubBytes);
    const WfSub wsub(wad, smem_raw);
    int k = 0;  // unit of the atom (by-particle mode)
    bool zero_pending = false;
    for (long item = tuple * grp; item < n_units;) {
        // ---- S1: radix-R0 butterflies u = tid + NT k over rows u + 512 j of u_c (jo = 0 requested
        // during the previous S2); g = W_M^u, h = W_L^{c u}
#ifndef WF_REAL_S1_SKIP
#define WF_REAL_S1_SKIP 0  // 1: a real column's unwanted sub-series are not stored either (wave-uniform branches
                           // around the output twiddles and LDS stores; costs registers: spills at R0 = 20)
#endif
        const unsigned qsel = (WF_REAL_S1_SKIP && BYP && kind != 2) ? P::kRealSel.qmask[pass ? 1 : 0] : ~0u;
#pragma unroll
        for (int k1 = 0; k1 < K1; ++k1) {
        const int u = tid + NT * k1;
        if (K1 * NT != N1 && u >= N1) continue;
        [[maybe_unused]] cd xwide[SRC32 ? R0 : 1];
        if constexpr (SRC32) {
#pragma unroll
            for (int j = 0; j < R0; ++j) xwide[j] = wf_widen(xx[k1][j], kind);
        }
        cd(&x)[R0] = wf_pick<SRC32>(xwide, xx[k1]);
        cd g, g2, h;
        auto load_seeds = [&]() {
            g = wf_load(twr, (unsigned)(u * R) * 32u, 0u);
            g2 = wf_load(twr, (unsigned)(u * R) * 64u, 0u);
            h = wf_load(twr, (unsigned)(u * pass) * 16u, 0u);
        };
        if constexpr (!LONG) load_seeds();
        if constexpr (LONG) {
            // u_c[u + 512 j] = sum_jo z[u + 512 j + M jo] U(j, jo), U = W_L^{c (512 j + M jo)} lane-uniform,
            // its index advanced by c 512 per j and c M per jo (mod L: one conditional subtraction)
            const int sj = pass * N1, so = pass * M;
            if (pass) {
                int idx = 0;
#pragma unroll
                for (int j = 1; j < R0; ++j) {
                    idx += sj;
                    idx -= idx >= L ? L : 0;
                    x[j] = cmul(x[j], tw_uniform(tw2, idx));
                }
            }
            auto outer_rows = [&](auto single) {
                int base = 0;
                for (int jo = 1; jo < R; ++jo) {
                    base += so;
                    base -= base >= L ? L : 0;
                    if (jo * M >= T) break;  // nothing but padding from here on
                    int idx = base;
                    // NZ rows in flight at a time (these requests are NOT prefetched: each group is a
                    // round trip to L2 / HBM in the middle of S1, so as many per group as the registers
                    // next to x[] allow: ten -- R0 = 20, R = 2: 2.57 -> 2.46 ms per 15000 pairs, R = 3:
                    // 3.28 -> 3.00 ms, same box; all twenty spill; the by-particle variant, which had
                    // two: 20000 x 25000 x 3 with the per-particle array 18.4 -> 16.5 ms)
                    constexpr int NZ = 10;
#pragma unroll
                    for (int j0 = 0; j0 < R0; j0 += NZ) {
                        cd z[NZ];
#pragma unroll
                        for (int i = 0; i < NZ; ++i)
                            if (j0 + i < R0)
                                z[i] = load_row(crs, decltype(single)::value ? kind : 2, u, (unsigned)(N1 * (j0 + i) + M * jo));
#pragma unroll
                        for (int i = 0; i < NZ; ++i)
                            if (j0 + i < R0) {
                                x[j0 + i] = wf_cfma(x[j0 + i], z[i], tw_uniform(tw2, idx));
                                idx += sj;
                                idx -= idx >= L ? L : 0;
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            if (BYP && kind != 2) outer_rows(std::true_type{});
            else outer_rows(std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            load_seeds();  // after the row loads: their registers are free again
        } else if (pass) {
            // pass B twist, lane-uniform part W_{2 R0}^j: literals (wfft_twist.inc) instead of 19
            // scalar loads per butterfly
#pragma unroll
            for (int j = 1; j < R0; ++j) x[j] = cmul(x[j], cd{WfTwist<R0>::re(j), WfTwist<R0>::im(j)});
        }
        Dft<R0>::run(x);
        {
            // output twiddles W_L^{u (2R q + c)} = h g^q: two chains (even / odd q) by g^2, each
            // output stored as soon as it is scaled (the 20 stores of a wave take ~260 LDS-path
            // cycles: issued in one burst at the end they are fully exposed)
            cd te = pass ? h : cd{1.0, 0.0};
            cd to = pass ? cmul(h, g) : g;
            if (pass) x[0] = cmul(x[0], te);
            if (!(WF_REAL_S1_SKIP && BYP) || (qsel & 1u)) lds[u] = x[0];
            if constexpr (R0 > 1) {
                x[1] = cmul(x[1], to);
                if (!(WF_REAL_S1_SKIP && BYP) || (qsel & 2u)) lds[N1 + u] = x[1];
            }
#pragma unroll
            for (int q = 2; q < R0; ++q) {
                if (q & 1) {
                    to = cmul(to, g2);
                    x[q] = cmul(x[q], to);
                } else {
                    te = cmul(te, g2);
                    x[q] = cmul(x[q], te);
                }
                // (only the LDS store is skipped: branches around the products as well cost registers)
                if (!(WF_REAL_S1_SKIP && BYP) || (qsel >> q & 1u)) lds[q * N1 + u] = x[q];
            }
        }
        }
        if constexpr (!kTwRes) {
            asm volatile("" : "+v"(lane));  // keeps the loads (and their registers) inside the loop
            load_stage_tw();
        }
        WF_STAMP(0)
        __syncthreads();
        if (BYP && zero_pending) {
#pragma unroll
            for (int s = 0; s < NS1; ++s)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[s][c] = 0.0;
            zero_pending = false;
        }
        // ---- S2: sub-series q = wave + NW s, two or three in flight per wave; the next unit's
        // rows are requested after the last one (before the barrier)
        const bool last_of_item = !BYP || k == upa - 1;
        long nitem = item;
        if (last_of_item) {
            if (BYP && grp == 2 && (item & 1) == 0 && item + 1 < n_units) nitem = item + 1;
            else nitem = (item & ~(long)(grp - 1)) + grp * n_tuples;
        }
        const int nk = last_of_item ? 0 : k + 1;
        const __amdgpu_buffer_rsrc_t nrs = unit_rsrc(nitem, nk, &nkind);
        // the next unit's rows: a quarter at each of four points of the wave's (first) S2 call
        auto row_hook = [&](auto part_c) {
#if WF_LOAD_PARTS == 4
            issue_loads_part(part_c, nrs, nkind);
#endif
        };
        if (BYP && kind != 2) {
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            // a real column: the wave's share of P::kRealSel, one sub-series at a time
            // (the next unit's row requests that would ride along S2 go out first)
            static_for_range<0, 4>([&](auto part_c) { row_hook(part_c); });
            const unsigned qs = P::kRealSel.qmask[pass ? 1 : 0] >> P::sub_base(wv);
            const unsigned q1 = P::kRealSel.q1mask[pass ? 1 : 0] >> P::sub_base(wv);
            // (measured: the waves with two selected sub-series running them interleaved instead -- 13.07
            // against 13.04 ms at 10000 x 100000 x 3 -- changes nothing: the unit waits for its first stage)
            static_for_range<0, NS1>([&](auto ss) {
                constexpr int s = decltype(ss)::value;
                if ((s < P::NLO || wv < P::REM) && (qs >> s & 1u))
                    wf_sub512_w<s, P::kRegExchangeSingle>(wsub, stw, acc[s], (q1 >> s & 1u) ? 1.0 : 2.0);
            });
        } else
        if constexpr (NS1 == 3 && P::NLO == 2 && P::REM != 0) {
            // two full slots and a partial third: the waves that own three sub-series take them
            // three at a time, the others two at a time (no row requests along S2 in these plans:
            // NSPREAD = 0.  Measured instead, R0 = 20, same box: every wave two at a time with 8 of
            // the 20 requests spread, the third alone afterwards -- correct, 2.12 against 2.11 ms.)
            if constexpr (SRC32) {  // room for the next unit's (float32) rows: requested along S2
                if (wave < P::REM) wf_sub512_x3<P::kRegExchange>(wsub, stw, acc[0], acc[1], acc[2], row_hook);
                else wf_sub512_x2<0, P::kRegExchange>(wsub, stw, acc[0], acc[1], row_hook);
            } else {
                if (wave < P::REM) wf_sub512_x3<P::kRegExchange>(wsub, stw, acc[0], acc[1], acc[2]);
                else wf_sub512_x2<0, P::kRegExchange>(wsub, stw, acc[0], acc[1]);
            }
        } else {
            static_for_range<0, NS1>([&](auto ss) {
                constexpr int s = decltype(ss)::value;
                constexpr bool full = s < P::NLO;                       // every wave has this slot
                constexpr bool nfull = s + 1 < P::NLO;                  // ... and the next one
                constexpr bool head = full && nfull && (s % 2 == 0);
                constexpr bool tail = full && s > 0 && (s % 2 == 1);
                // (the row requests ride on the wave's first call: every wave has slot 0)
                if constexpr (head) {
                    if constexpr (s == 0) wf_sub512_x2<s, P::kRegExchange>(wsub, stw, acc[s], acc[s + 1], row_hook);
                    else wf_sub512_x2<s, P::kRegExchange>(wsub, stw, acc[s], acc[s + 1]);
                } else if constexpr (!tail) {
                    if constexpr (s == 0) wf_sub512<s, P::kRegExchangeSingle>(wsub, stw, acc[s], row_hook);
                    else if (full || wave < P::REM) wf_sub512<s, P::kRegExchangeSingle>(wsub, stw, acc[s]);
                }
            });
        }
        if constexpr (STAMP) st_s2end = __builtin_amdgcn_s_memtime();  // behind the wave's own sub-series
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        auto store_acc = [&](long row) {  // accg[row][pass][q][cc / 2][lane][cc & 1]
            const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
                accg + (row * npass + pass) * (long)M, 0, M * 8, 0x00020000);
#pragma unroll
            for (int s = 0; s < NS1; ++s) {
                const int q = P::sub_base(wv) + s;
                if (s < P::NLO || wv < P::REM) {
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2) {
                        // 16 bytes per lane (cc = 2 c2, 2 c2 + 1): half the store instructions here and half
                        // the load instructions in k_winverse (by-particle step 23.6 -> 23.0 ms)
                        const cd two{acc[s][2 * c2], acc[s][2 * c2 + 1]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wf_u32x4, two), sr, (unsigned)lane * 16u,
                                                               (unsigned)((q * 4 + c2) * 64) * 16u, 0);
                    }
                }
            }
        };
        if constexpr (BYP) {
            // the atom's power spectrum in this pass; the accumulators restart from zero before the
            // next unit's S2
            if (last_of_item) {
                store_acc(item);
                zero_pending = true;
            }
        } else {
            if (nitem >= n_units) store_acc(tuple);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_loads_tail(nrs, nkind);
        kind = nkind, item = nitem, k = nk, crs = nrs;
        if constexpr (STAMP) st_tail += __builtin_amdgcn_s_memtime() - st_s2end;  // the row requests behind S2
        WF_STAMP(1)
        __syncthreads();
    }
    if constexpr (!BYP) {
        // a tuple without units still owes its (zero) row of the partial spectra
        if (tuple * grp >= n_units) {
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
                accg + (tuple * npass + pass) * (long)M, 0, M * 8, 0x00020000);
#pragma unroll
            for (int s = 0; s < NS1; ++s) {
                const int q = P::sub_base(wv) + s;
                if (s < P::sub_count(wv)) {
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wf_u32x4, cd{0.0, 0.0}), sr,
                                                               (unsigned)lane * 16u, (unsigned)((q * 4 + c2) * 64) * 16u, 0);
                }
            }
        }
    }
    if constexpr (STAMP) {
        st_acc[2] = __builtin_amdgcn_s_memtime() - st_t0;
        st_acc[3] = __builtin_amdgcn_s_memrealtime() - st_r0;
        if (lane == 0 && (wave == 0 || wave == NW / 2)) {  // 16 slots per workgroup: [0..3] wave 0, [4..7] wave NW/2, [8], [9] tails
            for (int i = 0; i < 4; ++i) stamps[16 * (long)blockIdx.x + (wave ? 4 : 0) + i] = st_acc[i];
            stamps[16 * (long)blockIdx.x + 8 + (wave ? 1 : 0)] = st_tail;
        }
    }
#undef WF_STAMP
}

// ================================================================================================
// Inverse kernel: the lag values of power spectra (one per atom for results.vacf_by_particle,
// velocityautocorr.py:145-147, 210-214; the spectrum summed over atoms for results.timeseries).
//
// A workgroup takes whole spectra (grid-stride).  With P real,
//     lag[n] = (1 / (L (T - n))) Re sum_k P[k] W_L^{k n},   k = 2R s + c:
//     lag[n] = (1 / (L (T - n))) Re sum_{c < 2R} W_L^{c n} Q_c[n mod M],   Q_c = FFT_M(P_c).
// Two passes share one complex M-point transform, Q = FFT_M(P_c + i P_c'), run by the TRANSPOSE
// of the forward algorithm (the DFT matrix is symmetric) -- input in the accumulators' own bin
// order, output in natural order:
//   sub-series q (registers cc = 0..7 of lane 8a + b: bin a + 8b + 64cc)
//     -> DFT8 over cc, x W_64^{b n0} -> exchange -> DFT8 over b -> exchange, x W_512^{l a}
//     -> DFT8 over a -> G_q[64 n2 + l] to LDS           (same exchange layouts as forward)
//   thread u: Q[u + 512 j'] = DFT_R0 over q of G_q[u] W_M^{uq}
// and with Qm[n] = Q[M - n] (one more trip through LDS, natural order) the two are separated:
//     Q_c = (Q + conj Qm) / 2,   Q_c' = (Q - conj Qm) / (2i).
// R = 1: lag[n] = ( Re(Q+Qm)/2 + cos(pi n/M) Im(Q+Qm)/2 - sin(pi n/M) Re(Q-Qm)/2 ) / (2M (T - n)).
// R > 1: the R transforms of a spectrum run one after the other, each adding its two passes'
// terms to the spectrum's output row (read-modify-write by the thread that owns the lag).
// Output: out[item * ld + n], atom-major (512-byte stores), transposed afterwards for the
// by-particle array.
struct WfSubT {
    int lane, hi, lo;
    __device__ __forceinline__ explicit WfSubT(int lane_) : lane(lane_) {
        asm volatile("" : "+v"(lane));
        hi = lane >> 3;
        lo = lane & 7;
    }
    // v[c] (bins of this lane) -> the sub-series' transform G[64 n2 + lane] stored at reg
    __device__ __forceinline__ void run(cd* __restrict__ reg, cd (&v)[8], const cd (&twa)[7],
                                        const cd (&twb)[7]) const {
        Dft<8>::run(v);  // over c -> n0
#pragma unroll
        for (int n0 = 1; n0 < 8; ++n0) v[n0] = cmul(v[n0], twb[n0 - 1]);  // W_64^{b n0}, b = lane & 7
        __builtin_amdgcn_wave_barrier();
        const int x = lo ^ hi;
#pragma unroll
        for (int n0 = 0; n0 < 8; ++n0) reg[(hi * 8 + n0) * 8 + (x ^ n0)] = v[n0];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int b = 0; b < 8; ++b) v[b] = reg[lane * 8 + (b ^ x)];  // lane = (a, n0)
        __builtin_amdgcn_wave_barrier();
        Dft<8>::run(v);  // over b -> n1
        __builtin_amdgcn_wave_barrier();
        const int base = hi * 64 + (lo ^ (8 * (hi & 1)));
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) reg[base ^ (8 * n1)] = v[n1];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int a = 0; a < 8; ++a) v[a] = reg[a * 64 + (lane ^ (8 * (a & 1)))];  // lane = l = 8 n1 + n0
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int a = 1; a < 8; ++a) v[a] = cmul(v[a], twa[a - 1]);  // W_512^{l a}
        Dft<8>::run(v);  // over a -> n2
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) reg[64 * n2 + lane] = v[n2];
        __builtin_amdgcn_wave_barrier();
    }
};



template <class P, bool LONG = false, int PF = 0>
__global__ void __launch_bounds__(P::NT)
    k_winverse(const double* __restrict__ spec, int T, long n_items, const cd* __restrict__ tw2,
               double* __restrict__ out, long ld, int R_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    // the forward kernel's workgroup: NW waves (one per sub-series up to 8), K1 first-stage
    // butterflies u = tid + NT k per thread
    constexpr int R0 = P::R0, N1 = P::N1, NW = P::NW, NT = P::NT, K1 = P::K1, M = P::M, NSA = P::NS1;
    static_assert(!LONG || PF == 0, "the spectrum prefetch is for the single-transform case");
    static_assert(PF <= NSA, "prefetch depth counts sub-series of a wave");
    const int R = LONG ? R_arg : 1, L = 2 * R * M;
    int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t twr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<cd*>(tw2), 0, (L + kWfStageRows * 64) * 16, 0x00020000);
    // the spectra of the first PF of this wave's sub-series are requested an item ahead (during
    // the previous item's first stage and untangling); the rest when their turn comes
    cd v[NSA][8];
    auto load_spec = [&](long item, int cp, int s) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int q = wv + NW * s;
        const bool live = item < n_items && q < R0;
        const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double*>(spec + (live ? item : 0) * (long)L + (long)(2 * cp) * M), 0, live ? 2 * M * 8 : 0,
            0x00020000);
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {  // 16 bytes per lane and load: elements cc = 2 c2, 2 c2 + 1 of a pass
            const cd a = wf_load(sr, (unsigned)lane * 16u, (unsigned)((q * 4 + c2) * 64) * 16u);
            const cd b = wf_load(sr, (unsigned)lane * 16u, (unsigned)(M / 2 + (q * 4 + c2) * 64) * 16u);
            v[s][2 * c2] = cd{a.x, b.x};
            v[s][2 * c2 + 1] = cd{a.y, b.y};
        }
    };
    // A thread owns the same lags n = u + 512 j for every spectrum: the normalisation 1 / (2M (T - n))
    // is formed ONCE per launch (an exact product, one division) and applied as a multiplication --
    // twenty FP64 divisions (~30 instructions each) per thread and spectrum were 15 % of the
    // kernel's vector instructions.  (Within one ulp of the quotient; the parity bar is 1e-10.)
    // (plans from R0 = 9 on: one workgroup per compute unit whatever the register count; the smaller
    // ones keep the division and their two or more workgroups per unit)
    constexpr bool kHoistNorm = !LONG && R0 >= 9;
    double rnorm[kHoistNorm ? K1 : 1][kHoistNorm ? R0 : 1];
    if constexpr (kHoistNorm) {
#pragma unroll
        for (int k1 = 0; k1 < K1; ++k1)
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                const int n = tid + NT * k1 + N1 * j;
                rnorm[k1][j] = n < T ? 1.0 / (2.0 * (double)M * (double)(T - n)) : 0.0;
            }
    }
#pragma unroll
    for (int s = 0; s < PF; ++s) load_spec(blockIdx.x, 0, s);
    for (long item = blockIdx.x; item < n_items; item += gridDim.x) {
        for (int cp = 0; cp < R; ++cp) {
            // per-thread offsets and LDS addresses are re-formed per transform: hoisted out of the
            // loops they would be spilled
            asm volatile("" : "+v"(tid), "+v"(lane));
            {
                cd twa[7], twb[7];
#pragma unroll
                for (int a = 0; a < 7; ++a) {
                    twa[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(L + a * 64) * 16u);
                    twb[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(L + (7 + a) * 64) * 16u);
                }
#pragma unroll
                for (int s = PF; s < NSA; ++s) load_spec(item, cp, s);
                const WfSubT wt(lane);
#pragma unroll
                for (int s = 0; s < NSA; ++s) {
                    const int q = wave + NW * s;
                    if (NW * s + NW - 1 < R0 || q < R0) wt.run(lds + q * N1, v[s], twa, twb);
                }
            }
            __syncthreads();
            if constexpr (PF > 0) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < PF; ++s) load_spec(item + gridDim.x, 0, s);
                __builtin_amdgcn_sched_barrier(0);
            }
            cd xx[K1][R0];
#pragma unroll
            for (int k1 = 0; k1 < K1; ++k1) {
                const int u = tid + NT * k1;
                if (K1 * NT != N1 && u >= N1) continue;
                cd(&x)[R0] = xx[k1];
                const cd g = wf_load(twr, (unsigned)(u * R) * 32u, 0u), g2 = wf_load(twr, (unsigned)(u * R) * 64u, 0u);
#pragma unroll
                for (int q = 0; q < R0; ++q) x[q] = lds[q * N1 + u];
                cd te = cd{1.0, 0.0}, to = g;
                if constexpr (R0 > 1) x[1] = cmul(x[1], to);
#pragma unroll
                for (int q = 2; q < R0; ++q) {
                    if (q & 1) {
                        to = cmul(to, g2);
                        x[q] = cmul(x[q], to);
                    } else {
                        te = cmul(te, g2);
                        x[q] = cmul(x[q], te);
                    }
                }
                Dft<R0>::run(x);  // x[j'] = Q[u + 512 j']
            }
            // No barrier here: thread u has read G_q[u] for every q -- column u of every 8 KiB block --
            // and writes Q[u + 512 j] into the SAME column of the same blocks; nobody else touches it.
            // (Round 4: the barrier that stood here cost the waves of a SIMD their skew twice.)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k1 = 0; k1 < K1; ++k1) {
                const int u = tid + NT * k1;
                if (K1 * NT != N1 && u >= N1) continue;
#pragma unroll
                for (int j = 0; j < R0; ++j) lds[u + N1 * j] = xx[k1][j];
            }
            __syncthreads();
            double* o = out + item * ld;
#pragma unroll
            for (int k1 = 0; k1 < K1; ++k1) {
                const int u = tid + NT * k1;
                if (K1 * NT != N1 && u >= N1) continue;
                cd(&x)[R0] = xx[k1];
                // Q[M - n] for n = u + 512 j: element (512 - u) + 512 (R0 - 1 - j), u > 0
                const int mu = u == 0 ? 0 : N1 - u;
                if constexpr (!LONG) {
                    const cd h = wf_load(twr, (unsigned)u * 16u, 0u);
#pragma unroll
                    for (int j = 0; j < R0; ++j) {
                        const int n = u + N1 * j;
                        const int jm = u == 0 ? (R0 - j) % R0 : R0 - 1 - j;
                        const cd qm = lds[mu + N1 * jm];
                        // W_2M^n = W_2M^u * W_{2 R0}^j = cos - i sin (the second factor lane-uniform)
                        // (the lane-uniform factor as a literal, wfft_twist.inc: a scalar load here shares its
                        // counter with the LDS loads of Qm and made every step wait for all of them)
                        const cd w = j == 0 ? h : cmul(h, cd{WfTwist<R0>::re(j), WfTwist<R0>::im(j)});
                        const double ar = 0.5 * (x[j].x + qm.x), br = 0.5 * (x[j].y + qm.y),
                                     bi = -0.5 * (x[j].x - qm.x);
                        const double lagv = ar + (w.x * br - w.y * bi);
                        if constexpr (kHoistNorm) {
                            if (n < T) o[n] = lagv * rnorm[k1][j];  // 1 / (2M (T - n))
                        } else {
                            if (n < T) o[n] = lagv / (2.0 * (double)M * (double)(T - n));  // 2M(T-n) < 2^53: exact product
                        }
                    }
                } else {
                    // passes c0 = 2 cp (real parts of the input) and c1 = c0 + 1 (imaginary parts):
                    // lag[n] += Re(W_L^{c0 n} Q_c0[n0]) + Re(W_L^{c1 n} Q_c1[n0]), n = n0 + M jo,
                    // W_L^{c n} = W_L^{c u} (per thread) x W_L^{c (512 j + M jo)} (lane-uniform, index mod L)
                    const int c0 = 2 * cp, c1 = c0 + 1;
                    const cd h0 = wf_load(twr, (unsigned)(u * c0) * 16u, 0u),
                             h1 = wf_load(twr, (unsigned)(u * c1) * 16u, 0u);
                    const bool first = cp == 0, last = cp == R - 1;
                    int j0 = 0, j1 = 0;  // c (512 j) mod L
#pragma unroll
                    for (int j = 0; j < R0; ++j) {
                        const int jm = u == 0 ? (R0 - j) % R0 : R0 - 1 - j;
                        const cd qm = lds[mu + N1 * jm];
                        const cd qa = cd{0.5 * (x[j].x + qm.x), 0.5 * (x[j].y - qm.y)};
                        const cd qb = cd{0.5 * (x[j].y + qm.y), -0.5 * (x[j].x - qm.x)};
                        int i0 = j0, i1 = j1;
                        for (int jo = 0; jo < R; ++jo) {
                            const int n = u + N1 * j + M * jo;
                            const cd w0 = cmul(h0, tw_uniform(tw2, i0)), w1 = cmul(h1, tw_uniform(tw2, i1));
                            double a = (w0.x * qa.x - w0.y * qa.y) + (w1.x * qb.x - w1.y * qb.y);
                            if (n < T) {
                                if (!first) a += o[n];
                                o[n] = last ? a / ((double)L * (double)(T - n)) : a;  // L (T-n) < 2^53: exact product
                            }
                            i0 += c0 * M, i0 -= i0 >= L ? L : 0;
                            i1 += c1 * M, i1 -= i1 >= L ? L : 0;
                        }
                        j0 += c0 * N1, j0 -= j0 >= L ? L : 0;
                        j1 += c1 * N1, j1 -= j1 >= L ? L : 0;
                    }
                }
            }
            __syncthreads();  // Q consumed before the next transform's sub-series overwrite the LDS
        }
    }
}

// ================================================================================================
// n_frames <= 512 (M = 512, R0 = 1): the whole padded series is ONE sub-series, so a wave works
// alone -- rows straight from global memory into the lane's eight registers, pass B's twist
// W_1024^t from eight resident per-lane factors, both passes' sub-transforms interleaved in two
// private 8 KiB LDS regions, no workgroup barrier anywhere.
struct W1 {
    static constexpr int NT = 256, NWV = 4, M = 512;
    static constexpr size_t kLds = (size_t)NWV * 2 * 512 * sizeof(cd);  // two regions per wave
};

// SINGLE (n_frames <= 256): the 512 bins of pass A alone are a long enough pad (512 >= 2 n_frames - 1).  With pass A's power
// spectrum doubled and pass B's left at zero, everything downstream — the sum over workgroups, the inverse transform with its
// 1 / 1024 — computes (1 / 512) sum_s P_A[s] W_512^{-s n}: the same lags from half the transforms.
template <bool SINGLE>
__device__ __forceinline__ void w1_two_passes(cd* __restrict__ regA, cd* __restrict__ regB, int lane,
                                              const cd (&twa)[7], const cd (&twb)[7], const cd (&v)[8],
                                              const cd (&tB)[8], double (&accA)[8], double (&accB)[8]) {
    const WfSubPost w(lane);
    if constexpr (SINGLE) {
        cd a[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) a[n2] = v[n2];
        w.stage_a(regA, a, twa);
        w.stage_b(regA, a, twb);
        w.stage_c(a, accA);
        return;
    }
    cd a[8], b[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) {
        a[n2] = v[n2];
        b[n2] = cmul(v[n2], tB[n2]);
    }
    w.stage_a(regA, a, twa);
    w.stage_a(regB, b, twa);
    w.stage_b(regA, a, twb);
    w.stage_b(regB, b, twb);
    w.stage_c(a, accA);
    w.stage_c(b, accB);
}

// lag sums: accg [gridDim.x * 4][1024], natural bin order (bin 2 s + B), one row per wave
// PACK (lag sums of short series, SINGLE only): the autocorrelations of several series ADD, and series that sit 512 / PACK
// rows apart in one transform do not meet at lags < n_frames as long as 512 / PACK >= 2 n_frames - 1: a wave transforms
// PACK column pairs at once — pair PACK p + q in rows 512 q / PACK ... of the 512 — 2 / 4 / 8 up to 128 / 64 / 32 frames.
template <bool SINGLE, int PACK>
__global__ void __launch_bounds__(W1::NT)
    k_w1_accum(const double* __restrict__ pm, long pitch, int T, long n_pairs,
               const cd* __restrict__ tw2, double* __restrict__ accg) {
    static_assert(PACK == 1 || (SINGLE && (PACK == 2 || PACK == 4 || PACK == 8)), "packing needs the one-pass form");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long gw = (long)blockIdx.x * W1::NWV + wave, nw = (long)gridDim.x * W1::NWV;
    cd* regA = lds + wave * 1024;
    cd* regB = regA + 512;
    const __amdgpu_buffer_rsrc_t twr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cd*>(tw2), 0, (2 * W1::M + kWfStageRows * 64) * 16, 0x00020000);
    cd twa[7], twb[7], tB[8];
#pragma unroll
    for (int a = 0; a < 7; ++a) {
        twa[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * W1::M + a * 64) * 16u);
        twb[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * W1::M + (7 + a) * 64) * 16u);
    }
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) tB[n2] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(64 * n2) * 16u);  // W_1024^t
    double accA[8], accB[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) accA[c] = accB[c] = 0.0;
    constexpr int kPer = 8 / PACK;  // registers (blocks of 64 rows) per packed pair
    for (long p = gw; p * PACK < n_pairs; p += nw) {
        cd v[8];
#pragma unroll
        for (int q = 0; q < PACK; ++q) {
            const long pair = p * PACK + q;  // (a pair past the end: an empty resource, zeros)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<double*>(pm + (pair < n_pairs ? pair : 0) * pitch * 2), 0, pair < n_pairs ? T * 16 : 0, 0x00020000);
#pragma unroll
            for (int j = 0; j < kPer; ++j) v[q * kPer + j] = wf_load(rs, (unsigned)lane * 16u, (unsigned)(64 * j) * 16u);
        }
        w1_two_passes<SINGLE>(regA, regB, lane, twa, twb, v, tB, accA, accB);
    }
    double* out = accg + gw * 2 * W1::M;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int sb = (lane >> 3) + 8 * (lane & 7) + 64 * c;
        out[2 * sb] = SINGLE ? 2.0 * accA[c] : accA[c];
        out[2 * sb + 1] = SINGLE ? 0.0 : accB[c];
    }
}

// by-particle: a wave takes whole atoms; out[atom * ld + lag] (atom-major, as k_wbp)
template <bool SINGLE>
__global__ void __launch_bounds__(W1::NT)
    k_w1_bp(const double* __restrict__ pm, long pitch, int T, long n_atoms, int D,
            const cd* __restrict__ tw2, double* __restrict__ out, long ld) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long gw = (long)blockIdx.x * W1::NWV + wave, nw = (long)gridDim.x * W1::NWV;
    cd* regA = lds + wave * 1024;
    cd* regB = regA + 512;
    const __amdgpu_buffer_rsrc_t twr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cd*>(tw2), 0, (2 * W1::M + kWfStageRows * 64) * 16, 0x00020000);
    cd twa[7], twb[7], tB[8];
#pragma unroll
    for (int a = 0; a < 7; ++a) {
        twa[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * W1::M + a * 64) * 16u);
        twb[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * W1::M + (7 + a) * 64) * 16u);
    }
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) tB[n2] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(64 * n2) * 16u);
    const int n_units = D == 3 ? 2 : 1;
    // the power spectrum of one particle (its units' spectra added) into acc
    auto forward = [&](long atom, double (&accA)[8], double (&accB)[8]) __attribute__((always_inline)) {
        // units as in k_wbp: the atom's aligned column pair (kind 2) and/or a single column
        auto unit_of = [&](int k, long* pair, int* kind) {
            const long c0 = atom * D;
            if (D == 2) *pair = atom, *kind = 2;
            else if (D == 1) *pair = c0 >> 1, *kind = (int)(c0 & 1);
            else if ((c0 & 1) == 0) *pair = (c0 >> 1) + k, *kind = k == 0 ? 2 : 0;
            else *pair = (c0 >> 1) + k, *kind = k == 0 ? 1 : 2;
        };
        if (SINGLE && n_units == 2 && T <= 128) {
            // both units of the atom in ONE transform, 256 rows apart (their autocorrelations add: k_w1_accum's packing)
            cd v[8];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                long pair;
                int kind;
                unit_of(k, &pair, &kind);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<double*>(pm + pair * pitch * 2), 0, T * 16, 0x00020000);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cd x = wf_load(rs, (unsigned)lane * 16u, (unsigned)(64 * j) * 16u);
                    if (kind != 2) x = cd{kind ? x.y : x.x, 0.0};
                    v[4 * k + j] = x;
                }
            }
            w1_two_passes<SINGLE>(regA, regB, lane, twa, twb, v, tB, accA, accB);
            return;
        }
        for (int k = 0; k < n_units; ++k) {
            long pair;
            int kind;
            unit_of(k, &pair, &kind);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<double*>(pm + pair * pitch * 2), 0, T * 16, 0x00020000);
            cd v[8];
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) {
                v[n2] = wf_load(rs, (unsigned)lane * 16u, (unsigned)(64 * n2) * 16u);
                if (kind != 2) v[n2] = cd{kind ? v[n2].y : v[n2].x, 0.0};
            }
            w1_two_passes<SINGLE>(regA, regB, lane, twa, twb, v, tB, accA, accB);
        }
    };
    if constexpr (SINGLE) {
        // One pass leaves the imaginary half of the inverse transform's input free: TWO particles share it (the transform of
        // P_a + i P_b, real spectra: Re q_a[n] = (Q[n].re + Q[-n].re) / 2, Re q_b[n] = (Q[n].im + Q[-n].im) / 2).
        for (long ap = gw; 2 * ap < n_atoms; ap += nw) {
            const long a0 = 2 * ap, a1 = a0 + 1;
            double acc0[8], acc1[8], unused[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc0[c] = acc1[c] = unused[c] = 0.0;
            forward(a0, acc0, unused);
            if (a1 < n_atoms) forward(a1, acc1, unused);
            cd v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = cd{2.0 * acc0[c], 2.0 * acc1[c]};  // (doubled: w1_two_passes)
            const WfSubT wt(lane);
            wt.run(regA, v, twa, twb);
            double* o0 = out + a0 * ld;
            double* o1 = out + a1 * ld;
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) {
                const int n = 64 * n2 + lane;
                const cd q = regA[n], qm = regA[(W1::M - n) % W1::M];
                if (n < T) {
                    const double norm = 1.0 / (2.0 * (double)W1::M * (double)(T - n));
                    o0[n] = 0.5 * (q.x + qm.x) * norm;
                    if (a1 < n_atoms) o1[n] = 0.5 * (q.y + qm.y) * norm;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        for (long atom = gw; atom < n_atoms; atom += nw) {
            double accA[8], accB[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) accA[c] = accB[c] = 0.0;
            for (int k = 0; k < n_units; ++k) {  // (spelled out, not `forward`: the compiler then stays at two waves per SIMD)
                const long c0 = atom * D;
                long pair;
                int kind;
                if (D == 2) pair = atom, kind = 2;
                else if (D == 1) pair = c0 >> 1, kind = (int)(c0 & 1);
                else if ((c0 & 1) == 0) pair = (c0 >> 1) + k, kind = k == 0 ? 2 : 0;
                else pair = (c0 >> 1) + k, kind = k == 0 ? 1 : 2;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<double*>(pm + pair * pitch * 2), 0, T * 16, 0x00020000);
                cd v[8];
#pragma unroll
                for (int n2 = 0; n2 < 8; ++n2) {
                    v[n2] = wf_load(rs, (unsigned)lane * 16u, (unsigned)(64 * n2) * 16u);
                    if (kind != 2) v[n2] = cd{kind ? v[n2].y : v[n2].x, 0.0};
                }
                w1_two_passes<false>(regA, regB, lane, twa, twb, v, tB, accA, accB);
            }
            // lag values: transposed transform of P_A + i P_B (R0 = 1: its output IS Q in natural order)
            cd v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = cd{accA[c], accB[c]};
            const WfSubT wt(lane);
            wt.run(regA, v, twa, twb);
            double* o = out + atom * ld;
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) {
                const int n = 64 * n2 + lane;
                const cd q = regA[n], qm = regA[(W1::M - n) % W1::M];
                const cd w = tB[n2];  // W_1024^n = cos - i sin
                const double ar = 0.5 * (q.x + qm.x), br = 0.5 * (q.y + qm.y), bi = -0.5 * (q.x - qm.x);
                const double L = ar + (w.x * br - w.y * bi);
                if (n < T) o[n] = L / (2.0 * (double)W1::M * (double)(T - n));
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace ta

