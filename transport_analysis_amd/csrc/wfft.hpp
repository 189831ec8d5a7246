// wfft.hpp — power-spectrum accumulation over column pairs, pair-major slabs (gfx950).
//
// Replaces the per-atom tidynamics.acf loop of VelocityAutocorr._conclude_fft
// (/root/reference/transport_analysis/velocityautocorr.py:208-215) on the timeseries path.
//
// Input layout ("pair-major", produced by ta_stage_commit / ta_relayout_dev): column pair p
// (columns 2p, 2p+1 of the (n_atoms*dim) columns) is one contiguous array of `pitch` rows of
// 16 bytes, row t = (x[t], y[t]) = the complex sample z[t] = x[t] + i y[t].  A workgroup reads
// a pair as 1 KiB-per-wave coalesced loads, every byte exactly once from HBM (the second pass
// re-reads it while it is still in the XCD's L2).
//
// Transform.  The 2M-point transform of the zero-padded series (M = R0 * 512 >= n_frames) is
// split into pass A (even bins, FFT_M(z)) and pass B (odd bins, FFT_M(z W_2M^t)); each pass is
//   S1  one radix-R0 butterfly per thread (thread u: rows u + 512 j), output q scaled by
//       W_2M^{u (2q + B)} and written to LDS as sub-series q (512 values, 8 KiB);
//   S2  512-point transforms of the R0 sub-series, ONE WAVE each, radix 8 x 8 x 8 with the data
//       of a lane in registers and two exchanges through the sub-series' own 8 KiB of LDS: no
//       workgroup barrier inside S2, so the waves of a SIMD drift apart and the LDS stores of
//       one hide under the arithmetic of the other; the last radix-8 stage adds |.|^2 into the
//       wave's register accumulators (bin k = 2 (q + R0 s) + B, s = a + 8 b + 64 c).
// Sub-transform i = B*R0 + q belongs to wave i % 8, so with R0 = 20 every wave owns exactly
// five (pass, q) slots and 40 accumulators per lane.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "fft_engine.hpp"

namespace ta {

#ifndef WF_ABL
#define WF_ABL 0  // timing ablations (wrong results): 1 no S2, 2 no S1 arithmetic, 3 no row loads,
                  // 4 no S2 arithmetic (LDS traffic only), 5 no S2 exchanges (arithmetic only)
#endif

template <int LO, int HI, class F>
__device__ __forceinline__ void static_for_range(F&& f) {
    [&]<int... I>(std::integer_sequence<int, I...>) {
        (f(std::integral_constant<int, LO + I>{}), ...);
    }(std::make_integer_sequence<int, (HI > LO ? HI - LO : 0)>{});
}

typedef unsigned int wf_u32x4 __attribute__((ext_vector_type(4)));

// ---- first-stage DFTs that fft_engine.hpp does not have: prime-factor 2x5 and 4x5 -----------
// n = (5 n1 + N1 n2) mod N, k1 = k mod N1, k2 = k mod 5: X[k] = sum W_N1^{n1 k1} W_5^{n2 k2} x[n]
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

// Twiddle table of plan R0 (host side; shared by the library and tools/wfft):
//   [0, 2M)                      W_2M^n = exp(-i pi n / M)
//   [2M, 2M + 896)               wave-local stage twiddles [14][64]: rows 0..6 W_512^{lane (r+1)},
//                                rows 7..13 W_64^{(lane & 7) (r-6)}
//   [2M + 896, 4M + 896)         first-stage output twiddles [B][q][u] = W_2M^{u (2q + B)}
inline size_t wf_table_elems(int R0) { return 4 * (size_t)R0 * 512 + 14 * 64; }
inline void wf_fill_table(int R0, cd* a) {
    const long M = (long)R0 * 512;
    const long double pi = 3.141592653589793238462643383279502884L;
    for (long n = 0; n < 2 * M; ++n) {
        if (n == 0) a[n] = cd{1.0, 0.0};
        else if (n == M) a[n] = cd{-1.0, 0.0};
        else if (2 * n == M) a[n] = cd{0.0, -1.0};
        else if (2 * n == 3 * M) a[n] = cd{0.0, 1.0};
        else {
            const long double h = pi * (long double)n / (long double)M;
            a[n] = cd{(double)cosl(h), (double)-sinl(h)};
        }
    }
    for (int r = 1; r < 8; ++r)
        for (int l = 0; l < 64; ++l) {
            a[2 * M + (r - 1) * 64 + l] = a[2 * R0 * l * r];
            a[2 * M + (6 + r) * 64 + l] = a[16 * R0 * (l & 7) * r];
        }
    for (int B = 0; B < 2; ++B)
        for (long q = 0; q < R0; ++q)
            for (long u = 0; u < 512; ++u)
                a[2 * M + 896 + (B * R0 + q) * 512 + u] = a[(u * (2 * q + B)) % (2 * M)];
}

template <int R0_>
struct WPlan {
    static constexpr int R0 = R0_;
    static constexpr int N1 = 512;          // sub-series length = one wave's transform
    static constexpr int NT = 512;          // threads: one first-stage butterfly each
    static constexpr int NW = NT / 64;
    static constexpr int M = R0 * N1;
    static constexpr int NS = (2 * R0 + NW - 1) / NW;  // (pass, q) slots per wave
    static constexpr size_t kLds = (size_t)M * sizeof(cd);
    // slots a pass can touch (over all waves): i = wave + NW*s in [B R0, (B+1) R0)
    static constexpr int slot_lo(int B) { return B * R0 < NW ? 0 : (B * R0 - (NW - 1) + NW - 1) / NW; }
    static constexpr int slot_hi(int B) { return ((B + 1) * R0 - 1) / NW; }
    // waves that own no sub-series in the pass's conditional slot (a contiguous range)
    static constexpr int idle_first(int B) {
        for (int w = 0; w < NW; ++w)
            if (!wave_has(B, w, slot_lo(B)) || !wave_has(B, w, slot_hi(B))) return w;
        return 0;
    }
    static constexpr int idle_waves(int B) {
        int n = 0;
        for (int w = 0; w < NW; ++w)
            if (!wave_has(B, w, slot_lo(B)) || !wave_has(B, w, slot_hi(B))) ++n;
        return n == NW ? 0 : n;
    }
    static constexpr bool wave_has(int B, int w, int s) { return w + NW * s >= B * R0 && w + NW * s < (B + 1) * R0; }
    // slot s belongs to pass B for every wave
    static constexpr bool slot_always(int B, int s) { return NW * s >= B * R0 && NW - 1 + NW * s < (B + 1) * R0; }
};

__device__ __forceinline__ cd wf_load(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uni_off) {
    // rows past the end of the series read as zero: the buffer's bounds check IS the padding.
    // lane_off: per-lane byte offset (VGPR), uni_off: wave-uniform byte offset (SGPR operand)
    const wf_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uni_off, 0);
    return __builtin_bit_cast(cd, v);
}

// One wave's 512-point DIF transform of the sub-series at `reg` (natural order in, the lane's
// eight outputs c = 0..7 are bins s = (lane>>3) + 8 (lane&7) + 64 c), |.|^2 added to acc.
// The two exchanges stay inside the sub-series' own 8 KiB; their layouts are conflict-free for
// ds_write_b128 / ds_read_b128 (verified exhaustively against the lane groups of the LDS):
//   exchange 1: (a, l)      at a*64 + (l ^ 8 (a&1))
//   exchange 2: (a, b, n0)  at (8a + b)*8 + (n0 ^ a ^ b)
// A wave's own DS operations execute in order, so a read issued after a write of the same wave
// sees it: no barrier, only the compiler has to keep the order (wave_barrier).
struct WfSub {
    int lane, hi, lo;
    __device__ __forceinline__ explicit WfSub(int lane_) : lane(lane_) {
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
#if WF_ABL != 4
        Dft<8>::run(v);
#pragma unroll
        for (int a = 1; a < 8; ++a) v[a] = cmul(v[a], twa[a - 1]);
#endif
        __builtin_amdgcn_wave_barrier();
#if WF_ABL != 5
#pragma unroll
        for (int a = 0; a < 8; ++a) reg[a * 64 + (lane ^ (8 * (a & 1)))] = v[a];
        __builtin_amdgcn_wave_barrier();
        const int base = hi * 64 + (lo ^ (8 * (hi & 1)));  // (8 n1 + lo) ^ 8 (hi&1), n1 = 0
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) v[n1] = reg[base ^ (8 * n1)];
#endif
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void stage_b(cd* __restrict__ reg, cd (&v)[8], const cd (&twb)[7]) const {
#if WF_ABL != 4
        Dft<8>::run(v);
#pragma unroll
        for (int b = 1; b < 8; ++b) v[b] = cmul(v[b], twb[b - 1]);
#endif
        __builtin_amdgcn_wave_barrier();
#if WF_ABL != 5
        const int x = lo ^ hi;
#pragma unroll
        for (int b = 0; b < 8; ++b) reg[(hi * 8 + b) * 8 + (x ^ b)] = v[b];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n0 = 0; n0 < 8; ++n0) v[n0] = reg[lane * 8 + (n0 ^ x)];
#endif
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void stage_c(cd (&v)[8], double (&acc)[8]) const {
#if WF_ABL != 4
        Dft<8>::run(v);
#endif
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = fma(v[c].y, v[c].y, fma(v[c].x, v[c].x, acc[c]));
        __builtin_amdgcn_wave_barrier();
    }
};

__device__ __forceinline__ void wf_sub512(cd* __restrict__ reg, int lane, const cd (&twa)[7],
                                          const cd (&twb)[7], double (&acc)[8]) {
    const WfSub w(lane);
    cd v[8];
    w.read_a(reg, v);
    w.stage_a(reg, v, twa);
    w.stage_b(reg, v, twb);
    w.stage_c(v, acc);
}

// Two sub-series interleaved: while one's butterflies run, the other's exchange (eight 16-byte
// stores, eight loads, ~300 cycles of LDS round trip) is in flight.
__device__ __forceinline__ void wf_sub512_x2(cd* __restrict__ reg0, cd* __restrict__ reg1, int lane,
                                             const cd (&twa)[7], const cd (&twb)[7],
                                             double (&acc0)[8], double (&acc1)[8]) {
    const WfSub w(lane);
    cd v0[8], v1[8];
    w.read_a(reg0, v0);
    w.read_a(reg1, v1);
    __builtin_amdgcn_wave_barrier();
    w.stage_a(reg0, v0, twa);
    w.stage_a(reg1, v1, twa);
    w.stage_b(reg0, v0, twb);
    w.stage_b(reg1, v1, twb);
    w.stage_c(v0, acc0);
    w.stage_c(v1, acc1);
}

// Three sub-series interleaved (the waves that own one more than the others: they then finish
// together with their SIMD partner's two instead of running a third alone).
__device__ __forceinline__ void wf_sub512_x3(cd* __restrict__ reg0, cd* __restrict__ reg1,
                                             cd* __restrict__ reg2, int lane, const cd (&twa)[7],
                                             const cd (&twb)[7], double (&acc0)[8], double (&acc1)[8],
                                             double (&acc2)[8]) {
    const WfSub w(lane);
    cd v0[8], v1[8], v2[8];
    w.read_a(reg0, v0);
    w.read_a(reg1, v1);
    w.read_a(reg2, v2);
    __builtin_amdgcn_wave_barrier();
    w.stage_a(reg0, v0, twa);
    w.stage_a(reg1, v1, twa);
    w.stage_a(reg2, v2, twa);
    w.stage_b(reg0, v0, twb);
    w.stage_b(reg1, v1, twb);
    w.stage_b(reg2, v2, twb);
    w.stage_c(v0, acc0);
    w.stage_c(v1, acc1);
    w.stage_c(v2, acc2);
}

// pm: pair-major slab, pair p at pm + p*pitch*2 doubles; T rows are valid, the rest of the
// transform length is zero padding.  accg: [gridDim.x][2M] float64, natural bin order.
// tw2: W_2M^n, n < 2M, followed (at tw2 + 2M) by the wave-local stage twiddles [14][64]:
// rows 0..6 = W_512^{lane (r+1)}, rows 7..13 = W_64^{(lane&7) (r-6)}.
//
// Software pipeline over (pair, pass): the 20 row loads of the NEXT pass are issued before the
// barrier that ends the current one (they land while the slowest wave finishes), the 14 stage
// twiddles of a wave are re-loaded from L2 after each first stage instead of being kept across
// it: a first-stage butterfly's 80 data registers, the 80 accumulator registers and the 56
// twiddle registers do not fit 256 together.
// what the library instantiates (measured per build on the GPU: tools/wfft/wfft_test)
#ifndef WF_TOUCH_DEFAULT
#define WF_TOUCH_DEFAULT false
#endif
#ifndef WF_PRE
#define WF_PRE 0  // first-stage butterfly before the barrier that frees the LDS (measured: 3-7 % slower)
#endif
#ifndef WF_SI
#define WF_SI 1   // first-stage stores interleaved with the output twiddles
#endif
#ifndef WF_INTER_DEFAULT
#define WF_INTER_DEFAULT false
#endif
template <class P, bool STAMP = false, bool TOUCH = WF_TOUCH_DEFAULT, bool INTER = WF_INTER_DEFAULT>
__global__ void __launch_bounds__(P::NT)
    k_wfft_accum(const double* __restrict__ pm, long pitch, int T, long n_pairs,
                 const cd* __restrict__ tw2, double* __restrict__ accg,
                 unsigned long long* __restrict__ stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    constexpr int R0 = P::R0, N1 = P::N1, NS = P::NS, NW = P::NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    double acc[NS][8];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[s][c] = 0.0;

    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_prev = 0;
    if constexpr (STAMP) st_prev = __builtin_amdgcn_s_memtime();
#define WF_STAMP(i)                                                   \
    if constexpr (STAMP) {                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[i] += now_ - st_prev;                                  \
        st_prev = now_;                                               \
    }

    auto rsrc_of = [&](long p) {
        // a pair past the end gets an empty buffer: its loads return zeros and are never used
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pm + (p < n_pairs ? p : 0) * pitch * 2), 0,
                                                 p < n_pairs ? T * 16 : 0, 0x00020000);
    };
    // x: the rows of a first-stage butterfly; g = W_2M^{2u}, g2 = g^2, h = W_2M^{u}: seeds of
    // its output twiddles (per-thread constants, re-loaded with the rows rather than held
    // across S2, where the registers are short)
    // (table reads as buffer loads: lane offset in one VGPR, row offset in an SGPR: no per-load
    // address registers for the compiler to hoist out of the loop and spill)
    const __amdgpu_buffer_rsrc_t twr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cd*>(tw2), 0, (2 * P::M + 14 * 64) * 16, 0x00020000);
    cd x[R0], g, g2, h;
    auto issue_loads = [&](__amdgpu_buffer_rsrc_t rs) {
#pragma unroll
        for (int j = 0; j < R0; ++j) x[j] = wf_load(rs, (unsigned)tid * 16u, (unsigned)(N1 * j) * 16u);
        g = wf_load(twr, (unsigned)tid * 32u, 0u);
        g2 = wf_load(twr, (unsigned)tid * 64u, 0u);
        h = wf_load(twr, (unsigned)tid * 16u, 0u);
    };
    issue_loads(rsrc_of(blockIdx.x));

    for (long p = blockIdx.x; p < n_pairs; p += gridDim.x) {
        auto one_pass = [&](auto BB) {
            constexpr int B = decltype(BB)::value;
            // ---- S1: radix-R0 butterfly u = tid over rows u + 512 j (loaded one pass ahead)
            if constexpr (B == 1) {
                // pass B twist, lane-uniform part: W_{2 R0}^j = tw2[j * 512]
#pragma unroll
                for (int j = 1; j < R0; ++j) x[j] = cmul(x[j], tw_uniform(tw2, j * N1));
            }
            Dft<R0>::run(x);
            // Everything above touches registers only, so it may run BEFORE the barrier that
            // ends the previous pass's S2 (WF_PRE): a wave that owns one sub-series fewer in
            // that pass does its butterfly while the others finish theirs.
            WF_STAMP(2 * B)
#if WF_PRE
            __syncthreads();  // every wave has read the previous pass's sub-series: LDS is free
#endif
            {
                // output twiddles W_2M^{u(2q+B)} = h^B g^q: two chains (even / odd q) by g^2;
                // WF_SI: each output is stored as soon as it is scaled
                const cd gg = g, gg2 = g2, hh = h;
                cd te = B ? hh : cd{1.0, 0.0};
                cd to = B ? cmul(hh, gg) : gg;
                if constexpr (B == 1) x[0] = cmul(x[0], te);
                if (WF_SI) lds[tid] = x[0];
                if constexpr (R0 > 1) {
                    x[1] = cmul(x[1], to);
                    if (WF_SI) lds[N1 + tid] = x[1];
                }
#pragma unroll
                for (int q = 2; q < R0; ++q) {
                    if (q & 1) {
                        to = cmul(to, gg2);
                        x[q] = cmul(x[q], to);
                    } else {
                        te = cmul(te, gg2);
                        x[q] = cmul(x[q], te);
                    }
                    if (WF_SI) lds[q * N1 + tid] = x[q];
                }
                if (!WF_SI) {
#pragma unroll
                    for (int q = 0; q < R0; ++q) lds[q * N1 + tid] = x[q];
                }
            }
            // this wave's stage twiddles for S2 (dead during S1); the scheduling barriers keep
            // the loads from being hoisted over the code before them (which would make their
            // destination registers live there)
            __builtin_amdgcn_sched_barrier(0);
            cd twa[7], twb[7];
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                twa[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * P::M + a * 64) * 16u);
                twb[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * P::M + (7 + a) * 64) * 16u);
            }
            __syncthreads();  // the sub-series are complete
            // ---- S2: this wave's sub-series of the pass
            static_for_range<P::slot_lo(B), P::slot_hi(B) + 1>([&](auto ss) {
                constexpr int s = decltype(ss)::value;
                constexpr int lo_s = P::slot_lo(B);
                // slots that every wave owns in this pass are taken two at a time
                constexpr int first_all = P::slot_always(B, lo_s) ? lo_s : lo_s + 1;
                constexpr bool in_pair_run = P::slot_always(B, s) && s >= first_all;
                constexpr int k = s - first_all;  // position inside the run of always-slots
                constexpr bool pair_head = INTER && in_pair_run && k % 2 == 0 && s + 1 <= P::slot_hi(B) && P::slot_always(B, s + 1);
                constexpr bool pair_tail = INTER && in_pair_run && k % 2 == 1 && P::slot_always(B, s - 1);
                const int i = wave + NW * s;
                if constexpr (pair_head) {
                    wf_sub512_x2(lds + (i - B * R0) * N1, lds + (i + NW - B * R0) * N1, lane, twa, twb, acc[s],
                                 acc[s + 1]);
                } else if constexpr (!pair_tail) {
                    if (P::slot_always(B, s) || (i >= B * R0 && i < (B + 1) * R0))
                        wf_sub512(lds + (i - B * R0) * N1, lane, twa, twb, acc[s]);
                }
            });
            // rows of the next pass: the same pair again (pass B), or the next pair
            __builtin_amdgcn_sched_barrier(0);
            const __amdgpu_buffer_rsrc_t nrs = rsrc_of(B == 0 ? p : p + gridDim.x);
            if constexpr (B == 1 && TOUCH && P::idle_waves(B) > 0) {
                // The waves that own one sub-series fewer in this pass are done early: they
                // pull the next pair's lines into L2 (one dword per 128-byte line), so that the
                // row loads issued below by the late waves do not wait for HBM.
                constexpr int NI = P::idle_waves(B), KT = (P::M * 16 / 128 + NI * 64 - 1) / (NI * 64);
                const int rank = wave - P::idle_first(B);
                if (rank >= 0 && rank < NI) {
                    unsigned t[KT];
#pragma unroll
                    for (int k = 0; k < KT; ++k)
                        t[k] = __builtin_amdgcn_raw_buffer_load_b32(nrs, (unsigned)(rank * 64 + lane) * 128u,
                                                                    (unsigned)(k * NI * 64) * 128u, 0);
                    issue_loads(nrs);
#pragma unroll
                    for (int k = 0; k < KT; ++k) asm volatile("" ::"v"(t[k]));
                } else {
                    issue_loads(nrs);
                }
            } else {
                issue_loads(nrs);
            }
            WF_STAMP(2 * B + 1)
#if !WF_PRE
            __syncthreads();
#endif
        };
        one_pass(std::integral_constant<int, 0>{});
        one_pass(std::integral_constant<int, 1>{});
    }
    // accumulators -> natural bin order
    double* out = accg + (long)blockIdx.x * 2 * P::M;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = wave + NW * s;
        if (i < 2 * R0) {
            const int B = i / R0, q = i - B * R0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int sb = (lane >> 3) + 8 * (lane & 7) + 64 * c;
                out[2 * (q + R0 * sb) + B] = acc[s][c];
            }
        }
    }
    if constexpr (STAMP) {
        if (lane == 0 && (wave == 0 || wave == 4))
            for (int i = 0; i < 4; ++i) stamps[8 * (long)blockIdx.x + (wave / 4) * 4 + i] = st_acc[i];
    }
#undef WF_STAMP
}


// ================================================================================================
// Pass-split variant: a workgroup does ONE pass (A: even bins, B: odd bins) of every pair of its
// couple; blocks b and b + 8 (same XCD under round-robin placement, speed only) form a couple
// and walk the same pair list at the same pace, so the second reader of a pair finds it in the
// XCD's L2 and every input byte leaves HBM once.  (k_wfft_accum reads a pair twice from one
// compute unit, half a pair-period apart: with 32 compute units x 160 KiB in flight per 4 MiB
// L2 the second read mostly misses -- measured 1.8x the algorithmic bytes at the L2's fabric
// side.)  Half the accumulators per thread (one pass: 24 per lane instead of 40 at R0 = 20),
// which is what lets the first-stage rows, the RESIDENT stage twiddles and two or three
// sub-series in flight fit 256 registers without spilling.
//
// grid: a multiple of 16 blocks; block b: pass B = (b >> 3) & 1, couple c = (b & 7) + 8 (b >> 4).
// Lag-sum mode (BYP = false): units are the n_units column pairs of the slab; accg:
//   [n_couples][2M] natural bin order, the A block writes the even bins of its couple's row, the
//   B block the odd ones.
// By-particle forward mode (BYP = true): units are the atoms' column units (wf_unit_of); a couple
//   takes whole atoms (pairs of adjacent atoms when the number of columns per atom is odd), and
//   after an atom's last unit the pass's accumulators -- the atom's power spectrum in this
//   pass -- go to accg as [atom][pass][q][c][lane] (the order k_wbp_inverse reads them back in)
//   and start again from zero.
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

template <class P, bool STAMP = false, bool INTER = true, bool BYP = false>
__global__ void __launch_bounds__(P::NT)
    k_wsplit_accum(const double* __restrict__ pm, long pitch, int T, long n_units,
                   const cd* __restrict__ tw2, double* __restrict__ accg,
                   unsigned long long* __restrict__ stamps = nullptr, int D = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    constexpr int R0 = P::R0, N1 = P::N1, NW = P::NW;
    constexpr int NS1 = (R0 + NW - 1) / NW;  // sub-series per wave: q = wave + 8 s
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int passB = (blockIdx.x >> 3) & 1;
    const long couple = (blockIdx.x & 7) + 8 * (blockIdx.x >> 4), n_couples = gridDim.x / 2;
    const int upa = BYP ? (D == 3 ? 2 : 1) : 1;  // units per atom (by-particle mode)
    // with an odd number of columns per atom, atoms 2i and 2i + 1 share a column pair (the last
    // column of one, the first of the other): a couple takes both, so the shared rows come from
    // the L2 the second time
    const int grp = BYP && (D & 1) ? 2 : 1;

    double acc[NS1][8];
#pragma unroll
    for (int s = 0; s < NS1; ++s)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[s][c] = 0.0;
    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_prev = 0;
    if constexpr (STAMP) st_prev = __builtin_amdgcn_s_memtime();
#define WF_STAMP(i)                                                   \
    if constexpr (STAMP) {                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[i] += now_ - st_prev;                                  \
        st_prev = now_;                                               \
    }
    const __amdgpu_buffer_rsrc_t twr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cd*>(tw2), 0, (4 * P::M + 14 * 64) * 16, 0x00020000);
    // the couple's units in order: lag-sum mode pairs couple + i n_couples; by-particle mode the
    // units k < upa of atoms couple + a n_couples.  A unit past the end gets an empty buffer: its
    // loads return zeros and are never used.
    auto unit_rsrc = [&](long item, int k, int* kind) {
        long pair = item;
        const bool live = item < n_units;
        *kind = 2;
        if constexpr (BYP) wf_unit_of(live ? item : 0, k, D, &pair, kind);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pm + (live ? pair : 0) * pitch * 2), 0,
                                                 live ? T * 16 : 0, 0x00020000);
    };
    cd x[R0];
    auto issue_loads = [&](__amdgpu_buffer_rsrc_t rs, int kd) {
        if (!BYP || kd == 2) {
#pragma unroll
            for (int j = 0; j < R0; ++j) x[j] = wf_load(rs, (unsigned)tid * 16u, (unsigned)(N1 * j) * 16u);
        } else {  // a single real column: that half of every row, imaginary part zero
#pragma unroll
            for (int j = 0; j < R0; ++j)
                x[j] = cd{__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                                                         rs, (unsigned)tid * 16u + (unsigned)kd * 8u,
                                                         (unsigned)(N1 * j) * 16u, 0)),
                          0.0};
        }
    };
    int kind = 2, nkind = 2;
    {
        const __amdgpu_buffer_rsrc_t rs0 = unit_rsrc(couple * grp, 0, &kind);
        issue_loads(rs0, kind);
    }
    // the wave-local stage twiddles stay in registers for the whole launch (one pass's
    // accumulators leave room for them: 14 fewer loads per wave and unit)
    cd twa[7], twb[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) {
        twa[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * P::M + a * 64) * 16u);
        twb[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * P::M + (7 + a) * 64) * 16u);
    }
    int k = 0;  // unit of the atom (by-particle mode)
    for (long item = couple * grp; item < n_units;) {
        // ---- S1: radix-R0 butterfly u = tid over rows u + 512 j (requested during the previous S2)
        const cd g = wf_load(twr, (unsigned)tid * 32u, 0u), g2 = wf_load(twr, (unsigned)tid * 64u, 0u),
                 h = wf_load(twr, (unsigned)tid * 16u, 0u);
#if WF_ABL != 2
        if (passB) {
            // pass B twist, lane-uniform part: W_{2 R0}^j = tw2[j * 512]
#pragma unroll
            for (int j = 1; j < R0; ++j) x[j] = cmul(x[j], tw_uniform(tw2, j * N1));
        }
        Dft<R0>::run(x);
#endif
        {
            // output twiddles W_2M^{u(2q+B)} = h^B g^q: two chains (even / odd q) by g^2, each
            // output stored as soon as it is scaled (the 20 stores of a wave take ~260 LDS-path
            // cycles: issued in one burst at the end they are fully exposed)
            cd te = passB ? h : cd{1.0, 0.0};
            cd to = passB ? cmul(h, g) : g;
            if (passB) x[0] = cmul(x[0], te);
            lds[tid] = x[0];
            if constexpr (R0 > 1) {
                x[1] = cmul(x[1], to);
                lds[N1 + tid] = x[1];
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
                lds[q * N1 + tid] = x[q];
            }
        }
        WF_STAMP(0)
        __syncthreads();
        // ---- S2: sub-series q = wave + 8 s, two or three in flight per wave; the next unit's
        // rows are requested after the last one (before the barrier)
        const bool last_of_item = !BYP || k == upa - 1;
        long nitem = item;
        if (last_of_item) {
            if (BYP && grp == 2 && (item & 1) == 0 && item + 1 < n_units) nitem = item + 1;
            else nitem = (item & ~(long)(grp - 1)) + grp * n_couples;
        }
        const int nk = last_of_item ? 0 : k + 1;
        const __amdgpu_buffer_rsrc_t nrs = unit_rsrc(nitem, nk, &nkind);
#if WF_ABL == 1
        if (T < 0)
#endif
        if constexpr (INTER && NS1 == 3 && NW * 2 + NW - 1 >= R0 && NW * 1 + NW - 1 < R0) {
            // two full slots and a partial third: the waves that own three sub-series take them
            // three at a time, the others two at a time
            if (wave + 2 * NW < R0)
                wf_sub512_x3(lds + wave * N1, lds + (wave + NW) * N1, lds + (wave + 2 * NW) * N1, lane, twa, twb,
                             acc[0], acc[1], acc[2]);
            else
                wf_sub512_x2(lds + wave * N1, lds + (wave + NW) * N1, lane, twa, twb, acc[0], acc[1]);
        } else {
            static_for_range<0, NS1>([&](auto ss) {
                constexpr int s = decltype(ss)::value;
                constexpr bool full = NW * s + NW - 1 < R0;             // every wave has this slot
                constexpr bool nfull = NW * (s + 1) + NW - 1 < R0;      // ... and the next one
                constexpr bool pfull = s > 0 && NW * (s - 1) + NW - 1 < R0;
                constexpr bool head = INTER && full && nfull && (s % 2 == 0);
                constexpr bool tail = INTER && full && pfull && (s % 2 == 1);
                const int q = wave + NW * s;
                if constexpr (head) {
                    wf_sub512_x2(lds + q * N1, lds + (q + NW) * N1, lane, twa, twb, acc[s], acc[s + 1]);
                } else if constexpr (!tail) {
                    if (full || q < R0) wf_sub512(lds + q * N1, lane, twa, twb, acc[s]);
                }
            });
        }
        if constexpr (BYP) {
            if (last_of_item) {
                // the atom's power spectrum in this pass: [atom][pass][q][c][lane], then from zero
                const long atom = item;
                const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
                    accg + ((atom * 2 + passB) * R0) * (8 * 64), 0, R0 * 512 * 8, 0x00020000);
                const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
                for (int s = 0; s < NS1; ++s) {
                    const int q = wv + NW * s;
                    if (NW * s + NW - 1 < R0 || q < R0) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wf_u32x2, acc[s][c]), sr,
                                                                  (unsigned)lane * 8u, (unsigned)((q * 8 + c) * 64) * 8u,
                                                                  0);
                            acc[s][c] = 0.0;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#if WF_ABL != 3
        issue_loads(nrs, nkind);
#endif
        kind = nkind, item = nitem, k = nk;
        WF_STAMP(1)
        __syncthreads();
    }
    if constexpr (!BYP) {
        double* out = accg + couple * 2 * P::M;
#pragma unroll
        for (int s = 0; s < NS1; ++s) {
            const int q = wave + NW * s;
            if (q < R0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int sb = (lane >> 3) + 8 * (lane & 7) + 64 * c;
                    out[2 * (q + R0 * sb) + passB] = acc[s][c];
                }
            }
        }
    }
    if constexpr (STAMP) {
        if (lane == 0 && (wave == 0 || wave == 4))
            for (int i = 0; i < 4; ++i) stamps[8 * (long)blockIdx.x + (wave / 4) * 4 + i] = st_acc[i];
    }
#undef WF_STAMP
}

// ================================================================================================
// M = 10240, two workgroups per compute unit ("half passes").
//
// k_wfft_accum holds a whole pass (160 KiB) in LDS, so a compute unit runs ONE workgroup whose
// waves move through the phases together: the first stage (vector-arithmetic bound) and the
// sub-series transforms (LDS bound) never overlap.  Here a workgroup is 256 threads with 80 KiB
// of LDS and does ONE pass (A: even bins or B: odd bins, by block index) in two halves: half H
// holds the ten sub-series q = H, H+2, ..., H+18 (the prime-factor 4 x 5 butterfly splits by
// q mod 4 in {H, H+2} at no extra arithmetic: a DFT4 needs 4 of its 8 additions for two of
// its outputs).  Two such workgroups share a compute unit and drift apart, so one's first
// stage runs under the other's exchanges.  The price: a pair's rows are read four times (twice
// per pass), three of them from L2; the A and B workgroups of a couple (blocks b and b + 8:
// same XCD under round-robin placement) walk the same pair list at the same pace.
template <int H>
__device__ __forceinline__ void dft20_half(const cd (&x)[20], cd (&y)[10]) {
    cd s0[5], s1[5];  // outputs k1 = H and H + 2 of the five DFT4 (over j1), indexed by j2
#pragma unroll
    for (int j2 = 0; j2 < 5; ++j2) {
        const cd a0 = x[(4 * j2) % 20], a1 = x[(5 + 4 * j2) % 20], a2 = x[(10 + 4 * j2) % 20],
                 a3 = x[(15 + 4 * j2) % 20];
        if constexpr (H == 0) {
            const cd t0 = a0 + a2, t2 = a1 + a3;
            s0[j2] = t0 + t2;
            s1[j2] = t0 - t2;
        } else {
            const cd t1 = a0 - a2, t3 = mul_mi(a1 - a3);
            s0[j2] = t1 + t3;
            s1[j2] = t1 - t3;
        }
    }
    Dft<5>::run(s0);
    Dft<5>::run(s1);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const int q = 2 * r + H;
        y[r] = (q % 4 == H) ? s0[q % 5] : s1[q % 5];
    }
}

struct WHalf {
    static constexpr int R0 = 20, N1 = 512, NT = 256, NW = 4, M = R0 * N1, NSUB = 10, NS = 5;
    static constexpr size_t kLds = (size_t)NSUB * N1 * sizeof(cd);
    // sub-series i = 10 H + r of a pass belongs to wave i % 4, slot i / 4
    static constexpr int slot_lo(int H) { return H * NSUB < NW ? 0 : (H * NSUB - (NW - 1) + NW - 1) / NW; }
    static constexpr int slot_hi(int H) { return ((H + 1) * NSUB - 1) / NW; }
};

// grid: a multiple of 16 blocks; block b: pass B = (b >> 3) & 1, couple c = (b & 7) + 8 (b >> 4);
// accg: [n_couples][2M], the A block writes the even bins of its couple's row, the B block the odd.
template <bool STAMP = false>
__global__ void __launch_bounds__(256, 2)
    k_whalf_accum(const double* __restrict__ pm, long pitch, int T, long n_pairs,
                  const cd* __restrict__ tw2, double* __restrict__ accg,
                  unsigned long long* __restrict__ stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    using P = WHalf;
    constexpr int N1 = P::N1, NS = P::NS, NW = P::NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int passB = (blockIdx.x >> 3) & 1;
    const long couple = (blockIdx.x & 7) + 8 * (blockIdx.x >> 4), n_couples = gridDim.x / 2;

    double acc[NS][8];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[s][c] = 0.0;
    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_prev = 0;
    if constexpr (STAMP) st_prev = __builtin_amdgcn_s_memtime();
#define WF_STAMP(i)                                                   \
    if constexpr (STAMP) {                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[i] += now_ - st_prev;                                  \
        st_prev = now_;                                               \
    }
    const __amdgpu_buffer_rsrc_t twr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cd*>(tw2), 0, (2 * P::M + 14 * 64) * 16, 0x00020000);
    auto rsrc_of = [&](long p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pm + (p < n_pairs ? p : 0) * pitch * 2), 0,
                                                 p < n_pairs ? T * 16 : 0, 0x00020000);
    };
    // rows of butterfly u and the seeds of its output twiddles: g = W_2M^{2u}, g2 = g^2, h = W_2M^u
    cd x[20], g, g2, h;
    auto issue_loads = [&](__amdgpu_buffer_rsrc_t rs, int u) {
#pragma unroll
        for (int j = 0; j < 20; ++j) x[j] = wf_load(rs, (unsigned)u * 16u, (unsigned)(N1 * j) * 16u);
        g = wf_load(twr, (unsigned)u * 32u, 0u);
        g2 = wf_load(twr, (unsigned)u * 64u, 0u);
        h = wf_load(twr, (unsigned)u * 16u, 0u);
    };
    issue_loads(rsrc_of(couple), tid);

    for (long p = couple; p < n_pairs; p += n_couples) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(p);
        auto one_half = [&](auto HH) {
            constexpr int H = decltype(HH)::value;
            // ---- S1: two butterflies per thread (u = tid, tid + 256), ten outputs each
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                const int u = tid + 256 * rep;
                if (rep == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_loads(rs, u);
                }
                if (passB) {
                    // pass B twist, lane-uniform part: W_40^j = tw2[j * 512]
#pragma unroll
                    for (int j = 1; j < 20; ++j) x[j] = cmul(x[j], tw_uniform(tw2, j * N1));
                }
                cd y[10];
                dft20_half<H>(x, y);
                // output twiddles W_2M^{u (2q + B)}, q = 2r + H: start h^B g^H, step g^2
                cd w = passB ? h : cd{1.0, 0.0};
                if constexpr (H == 1) w = passB ? cmul(w, g) : g;
#pragma unroll
                for (int r = 0; r < 10; ++r) {
                    if (r > 0) w = cmul(w, g2);
                    if (r > 0 || H == 1 || passB) y[r] = cmul(y[r], w);
                    lds[r * N1 + u] = y[r];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            cd twa[7], twb[7];
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                twa[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * P::M + a * 64) * 16u);
                twb[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * P::M + (7 + a) * 64) * 16u);
            }
            WF_STAMP(2 * H)
            __syncthreads();
            // ---- S2: this wave's sub-series of the half
            static_for_range<P::slot_lo(H), P::slot_hi(H) + 1>([&](auto ss) {
                constexpr int s = decltype(ss)::value;
                const int i = wave + NW * s;
                if (i >= H * P::NSUB && i < (H + 1) * P::NSUB)
                    wf_sub512(lds + (i - H * P::NSUB) * N1, lane, twa, twb, acc[s]);
            });
            // first butterfly of the next half: the same pair again, or the next pair
            __builtin_amdgcn_sched_barrier(0);
            issue_loads(H == 0 ? rs : rsrc_of(p + n_couples), tid);
            WF_STAMP(2 * H + 1)
            __syncthreads();
        };
        one_half(std::integral_constant<int, 0>{});
        one_half(std::integral_constant<int, 1>{});
    }
    // accumulators -> natural bin order: bin 2 (q + 20 sb) + B, q = 2r + H, i = 10 H + r
    double* out = accg + couple * 2 * P::M;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = wave + NW * s;
        const int H = i / P::NSUB, r = i - H * P::NSUB, q = 2 * r + H;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int sb = (lane >> 3) + 8 * (lane & 7) + 64 * c;
            out[2 * (q + 20 * sb) + passB] = acc[s][c];
        }
    }
    if constexpr (STAMP) {
        if (lane == 0 && wave == 0)
            for (int i = 0; i < 4; ++i) stamps[8 * (long)blockIdx.x + i] = st_acc[i];
    }
#undef WF_STAMP
}


// ================================================================================================
// By-particle mode: results.vacf_by_particle (velocityautocorr.py:145-147, 210-214).
//
// A workgroup takes whole atoms.  An atom's columns are transformed as one or two UNITS (its
// 16-byte aligned column pair as a complex series, a single column as a real one), both passes
// each, and |.|^2 is accumulated in registers as in the lag-sum kernels -- sub-series q of BOTH
// passes belongs to wave q % 8, so a lane holds P_A[j] and P_B[j] of the same bins j.  After the
// atom's last unit those accumulators ARE its power spectrum; its lag values are one M-point
// transform of P_A + i P_B, run by the TRANSPOSE of the forward algorithm (the DFT matrix is
// symmetric): input in the accumulators' own bin order, output in natural order:
//   sub-series q (registers c = 0..7 of lane 8a + b: bin a + 8b + 64c)
//     -> DFT8 over c, x W_64^{b n0} -> exchange -> DFT8 over b -> exchange, x W_512^{l a}
//     -> DFT8 over a -> G_q[64 n2 + l] to LDS           (same exchange layouts as forward)
//   thread u: Q[u + 512 j'] = DFT_R0 over q of G_q[u] W_M^{uq}
// and with Qm[n] = Q[M - n] (one more trip through LDS, natural order):
//   lag[n] = ( Re(Q+Qm)/2 + cos(pi n/M) Im(Q+Qm)/2 - sin(pi n/M) Re(Q-Qm)/2 ) / (2M (T - n)).
// Output: atom-major scratch out[atom * ld + n] (512-byte stores), transposed afterwards.
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

// pm: pair-major slab of the shard (n_atoms * D columns); out: [n_atoms][ld] atom-major lags.
template <class P, bool STAMP = false>
__global__ void __launch_bounds__(P::NT)
    k_wbp(const double* __restrict__ pm, long pitch, int T, long n_atoms, int D,
          const cd* __restrict__ tw2, double* __restrict__ out, long ld,
          unsigned long long* __restrict__ stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    constexpr int R0 = P::R0, N1 = P::N1, NW = P::NW, M = P::M;
    constexpr int NSA = (R0 + NW - 1) / NW;  // sub-series per wave and pass: q = wave + 8 s
    int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t twr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cd*>(tw2), 0, (4 * M + 14 * 64) * 16, 0x00020000);
    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_prev = 0;
    if constexpr (STAMP) st_prev = __builtin_amdgcn_s_memtime();
#define WF_STAMP(i)                                                   \
    if constexpr (STAMP) {                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[i] += now_ - st_prev;                                  \
        st_prev = now_;                                               \
    }
    // units of an atom: (pair index, kind) with kind 2 = both columns of the pair (complex
    // series), 0 / 1 = only that half (real series)
    const int n_units = D == 3 ? 2 : 1;
    auto unit_of = [&](long atom, int k, long* pair, int* kind) {
        const long c0 = atom * D;
        if (D == 2) {
            *pair = atom, *kind = 2;
        } else if (D == 1) {
            *pair = c0 >> 1, *kind = (int)(c0 & 1);
        } else if ((c0 & 1) == 0) {  // even first column: (x, y) aligned, z alone
            if (k == 0) *pair = c0 >> 1, *kind = 2;
            else *pair = (c0 >> 1) + 1, *kind = 0;
        } else {                     // x alone (second half of a pair), (y, z) aligned
            if (k == 0) *pair = c0 >> 1, *kind = 1;
            else *pair = (c0 >> 1) + 1, *kind = 2;
        }
    };
    cd x[R0], g, g2, h;
    auto issue_loads = [&](long pair) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double*>(pm + pair * pitch * 2), 0, T * 16, 0x00020000);
#pragma unroll
        for (int j = 0; j < R0; ++j) x[j] = wf_load(rs, (unsigned)tid * 16u, (unsigned)(N1 * j) * 16u);
    };
    auto load_seeds = [&]() {  // after the butterfly: its temporaries are dead by then
        g = wf_load(twr, (unsigned)tid * 32u, 0u);
        g2 = wf_load(twr, (unsigned)tid * 64u, 0u);
        h = wf_load(twr, (unsigned)tid * 16u, 0u);
    };
    auto load_stage_tw = [&](cd (&twa)[7], cd (&twb)[7]) {
#pragma unroll
        for (int a = 0; a < 7; ++a) {
            twa[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * M + a * 64) * 16u);
            twb[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * M + (7 + a) * 64) * 16u);
        }
    };

    for (long atom = blockIdx.x; atom < n_atoms; atom += gridDim.x) {
        double accA[NSA][8], accB[NSA][8];
#pragma unroll
        for (int s = 0; s < NSA; ++s)
#pragma unroll
            for (int c = 0; c < 8; ++c) accA[s][c] = accB[s][c] = 0.0;
        for (int k = 0; k < n_units; ++k) {
            long pair;
            int kind;
            unit_of(atom, k, &pair, &kind);
            auto one_pass = [&](auto BB, double (&acc)[NSA][8]) {
                constexpr int B = decltype(BB)::value;
                // per-thread offsets and LDS addresses are re-formed per pass: hoisted out of the
                // atom loop they would be spilled
                asm volatile("" : "+v"(tid), "+v"(lane));
                // (requesting the rows a pass ahead costs more in scratch traffic than the
                // latency it hides here: measured 24.7 -> 31.2 ms at 10000 x 100000 x 3)
                issue_loads(pair);
                if (kind != 2) {  // a single real column: the other half of the rows is not ours
#pragma unroll
                    for (int j = 0; j < R0; ++j) x[j] = cd{kind ? x[j].y : x[j].x, 0.0};
                }
                if constexpr (B == 1) {
#pragma unroll
                    for (int j = 1; j < R0; ++j) x[j] = cmul(x[j], tw_uniform(tw2, j * N1));
                }
                Dft<R0>::run(x);
                __builtin_amdgcn_sched_barrier(0);
                load_seeds();
                {
                    cd te = B ? h : cd{1.0, 0.0};
                    cd to = B ? cmul(h, g) : g;
                    if constexpr (B == 1) x[0] = cmul(x[0], te);
                    lds[tid] = x[0];
                    if constexpr (R0 > 1) {
                        x[1] = cmul(x[1], to);
                        lds[N1 + tid] = x[1];
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
                        lds[q * N1 + tid] = x[q];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                cd twa[7], twb[7];
                load_stage_tw(twa, twb);
                WF_STAMP(0)
                __syncthreads();
#pragma unroll
                for (int s = 0; s < NSA; ++s) {
                    const int q = wave + NW * s;
                    if (NW * s + NW - 1 < R0 || q < R0) wf_sub512(lds + q * N1, lane, twa, twb, acc[s]);
                }
                WF_STAMP(1)
                __syncthreads();
            };
            one_pass(std::integral_constant<int, 0>{}, accA);
            one_pass(std::integral_constant<int, 1>{}, accB);
        }
        // ---- the atom's lag values: transposed transform of P_A + i P_B
        asm volatile("" : "+v"(tid), "+v"(lane));
        {
            cd twa[7], twb[7];
            load_stage_tw(twa, twb);
            const WfSubT wt(lane);
#pragma unroll
            for (int s = 0; s < NSA; ++s) {
                const int q = wave + NW * s;
                if (NW * s + NW - 1 < R0 || q < R0) {
                    cd v[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = cd{accA[s][c], accB[s][c]};
                    wt.run(lds + q * N1, v, twa, twb);
                }
            }
        }
        __syncthreads();
        {
            load_seeds();  // g = W_M^u
#pragma unroll
            for (int q = 0; q < R0; ++q) x[q] = lds[q * N1 + tid];
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
            Dft<R0>::run(x);  // x[j'] = Q[tid + 512 j']
        }
        __syncthreads();  // every thread has read its G values: LDS free for Q in natural order
#pragma unroll
        for (int j = 0; j < R0; ++j) lds[tid + N1 * j] = x[j];
        __syncthreads();
        {
            double* o = out + atom * ld;
            // Q[M - n] for n = u + 512 j: element (512 - u) + 512 (R0 - 1 - j), u > 0
            const int mu = tid == 0 ? 0 : N1 - tid;
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                const int n = tid + N1 * j;
                const int jm = tid == 0 ? (R0 - j) % R0 : R0 - 1 - j;
                const cd qm = lds[mu + N1 * jm];
                // W_2M^n = W_2M^u * W_{2 R0}^j = cos - i sin (the second factor lane-uniform)
                const cd w = j == 0 ? h : cmul(h, tw_uniform(tw2, j * N1));
                const double ar = 0.5 * (x[j].x + qm.x), br = 0.5 * (x[j].y + qm.y), bi = -0.5 * (x[j].x - qm.x);
                const double L = ar + (w.x * br - w.y * bi);
                if (n < T) o[n] = L / (2.0 * (double)M * (double)(T - n));  // 2M(T-n) < 2^53: exact product
            }
        }
        __syncthreads();  // Q consumed before the next atom's first stage overwrites the LDS
    }
    if constexpr (STAMP) {
        if (lane == 0 && wave == 0)
            for (int i = 0; i < 4; ++i) stamps[8 * (long)blockIdx.x + i] = st_acc[i];
    }
#undef WF_STAMP
}


// By-particle inverse: a workgroup per atom (grid-stride).  spec: [atom][2][R0][8][64] from
// k_wsplit_accum<BYP>; out[atom * ld + lag], atom-major.  The transposed transform of
// P_A + i P_B as in k_wbp (which see), with registers to spare: two sub-series in flight.
template <class P, int PF = 0, int ABL = 0>
__global__ void __launch_bounds__(P::NT)
    k_wbp_inverse(const double* __restrict__ spec, int T, long n_atoms, const cd* __restrict__ tw2,
                  double* __restrict__ out, long ld) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    constexpr int R0 = P::R0, N1 = P::N1, NW = P::NW, M = P::M;
    constexpr int NSA = (R0 + NW - 1) / NW;
    int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t twr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cd*>(tw2), 0, (4 * M + 14 * 64) * 16, 0x00020000);
    // the spectra of the first PF of this wave's sub-series are requested an atom ahead (during
    // the previous atom's first stage and untangling); the rest when their turn comes
    cd v[NSA][8];
    auto load_spec = [&](long atom, int s) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int q = wv + NW * s;
        const bool live = atom < n_atoms && q < R0 && ABL != 5;
        const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double*>(spec + ((live ? atom : 0) * 2 * R0) * (8 * 64)), 0, live ? 2 * R0 * 512 * 8 : 0,
            0x00020000);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            v[s][c].x = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                                                       sr, (unsigned)lane * 8u, (unsigned)((q * 8 + c) * 64) * 8u, 0));
            v[s][c].y = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                                                       sr, (unsigned)lane * 8u,
                                                       (unsigned)(((R0 + q) * 8 + c) * 64) * 8u, 0));
        }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) load_spec(blockIdx.x, s);
    for (long atom = blockIdx.x; atom < n_atoms; atom += gridDim.x) {
        // per-thread offsets and LDS addresses are re-formed per atom: hoisted out of the loop
        // they would be spilled
        asm volatile("" : "+v"(tid), "+v"(lane));
        {
            cd twa[7], twb[7];
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                twa[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * M + a * 64) * 16u);
                twb[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * M + (7 + a) * 64) * 16u);
            }
#pragma unroll
            for (int s = PF; s < NSA; ++s) load_spec(atom, s);
            const WfSubT wt(lane);
#pragma unroll
            for (int s = 0; s < NSA; ++s) {
                const int q = wave + NW * s;
                if (NW * s + NW - 1 < R0 || q < R0) {
                    if constexpr (ABL == 1) {  // ablation: no sub-transforms
#pragma unroll
                        for (int c = 0; c < 8; ++c) lds[q * N1 + 64 * c + lane] = v[s][c];
                    } else {
                        wt.run(lds + q * N1, v[s], twa, twb);
                    }
                }
            }
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < PF; ++s) load_spec(atom + gridDim.x, s);
        __builtin_amdgcn_sched_barrier(0);
        cd x[R0];
        const cd g = wf_load(twr, (unsigned)tid * 32u, 0u), g2 = wf_load(twr, (unsigned)tid * 64u, 0u),
                 h = wf_load(twr, (unsigned)tid * 16u, 0u);
#pragma unroll
        for (int q = 0; q < R0; ++q) x[q] = lds[q * N1 + tid];
        if constexpr (ABL != 2) {
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
        }
        if constexpr (ABL != 2) Dft<R0>::run(x);  // x[j'] = Q[tid + 512 j']
        __syncthreads();  // every thread has read its G values: LDS free for Q in natural order
#pragma unroll
        for (int j = 0; j < R0; ++j) lds[tid + N1 * j] = x[j];
        __syncthreads();
        {
            double* o = out + atom * ld;
            const int mu = tid == 0 ? 0 : N1 - tid;
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                const int n = tid + N1 * j;
                if constexpr (ABL == 3) {  // ablation: no untangling
                    if (n < T) o[n] = x[j].x;
                    continue;
                }
                const int jm = tid == 0 ? (R0 - j) % R0 : R0 - 1 - j;
                const cd qm = lds[mu + N1 * jm];
                const cd w = j == 0 ? h : cmul(h, tw_uniform(tw2, j * N1));  // W_2M^n = cos - i sin
                const double ar = 0.5 * (x[j].x + qm.x), br = 0.5 * (x[j].y + qm.y), bi = -0.5 * (x[j].x - qm.x);
                const double L = ar + (w.x * br - w.y * bi);
                if constexpr (ABL == 4) {  // ablation: no division
                    if (n < T) o[n] = L * (2.0 * (double)M * (double)(T - n));
                    continue;
                }
                if (n < T) o[n] = L / (2.0 * (double)M * (double)(T - n));  // 2M(T-n) < 2^53: exact product
            }
        }
        __syncthreads();  // Q consumed before the next atom's sub-series overwrite the LDS
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

__device__ __forceinline__ void w1_two_passes(cd* __restrict__ regA, cd* __restrict__ regB, int lane,
                                              const cd (&twa)[7], const cd (&twb)[7], const cd (&v)[8],
                                              const cd (&tB)[8], double (&accA)[8], double (&accB)[8]) {
    const WfSub w(lane);
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
__global__ void __launch_bounds__(W1::NT)
    k_w1_accum(const double* __restrict__ pm, long pitch, int T, long n_pairs,
               const cd* __restrict__ tw2, double* __restrict__ accg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long gw = (long)blockIdx.x * W1::NWV + wave, nw = (long)gridDim.x * W1::NWV;
    cd* regA = lds + wave * 1024;
    cd* regB = regA + 512;
    const __amdgpu_buffer_rsrc_t twr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cd*>(tw2), 0, (4 * W1::M + 14 * 64) * 16, 0x00020000);
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
    for (long p = gw; p < n_pairs; p += nw) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double*>(pm + p * pitch * 2), 0, T * 16, 0x00020000);
        cd v[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) v[n2] = wf_load(rs, (unsigned)lane * 16u, (unsigned)(64 * n2) * 16u);
        w1_two_passes(regA, regB, lane, twa, twb, v, tB, accA, accB);
    }
    double* out = accg + gw * 2 * W1::M;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int sb = (lane >> 3) + 8 * (lane & 7) + 64 * c;
        out[2 * sb] = accA[c];
        out[2 * sb + 1] = accB[c];
    }
}

// by-particle: a wave takes whole atoms; out[atom * ld + lag] (atom-major, as k_wbp)
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
        const_cast<cd*>(tw2), 0, (4 * W1::M + 14 * 64) * 16, 0x00020000);
    cd twa[7], twb[7], tB[8];
#pragma unroll
    for (int a = 0; a < 7; ++a) {
        twa[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * W1::M + a * 64) * 16u);
        twb[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(2 * W1::M + (7 + a) * 64) * 16u);
    }
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) tB[n2] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(64 * n2) * 16u);
    const int n_units = D == 3 ? 2 : 1;
    for (long atom = gw; atom < n_atoms; atom += nw) {
        double accA[8], accB[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) accA[c] = accB[c] = 0.0;
        for (int k = 0; k < n_units; ++k) {
            // units as in k_wbp: the atom's aligned column pair (kind 2) and/or a single column
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
            w1_two_passes(regA, regB, lane, twa, twb, v, tB, accA, accB);
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

}  // namespace ta
