// bandbp_kernels.hpp — the windowed VACF WITH its by-particle array on the FP64 matrix cores (gfx950
// v_mfma_f64_16x16x4_f64).
//
// Quantity: VelocityAutocorr._conclude_simple (/root/reference/transport_analysis/velocityautocorr.py:217-238),
// the class default vacf_by_particle[k, n] = sum_{i < T-k} sum_d v[i, n, d] v[i+k, n, d] / (T - k): every
// particle's own lag sums, which band_kernels.hpp (lag sums over all particles) cannot give and k_direct
// (direct_kernels.hpp) computes on the vector units at 0.63 of their peak.
//
// The contraction of ONE particle is short (dim <= 3 columns), so the instruction's four k-slots are filled
// from the time axis instead: with time cut into blocks of 16 frames, the k-slot s of a "super-step" S is block
// 4 S + s, the A operand of column c is v[64 S + lane, c] — 64 consecutive frames, lane = 16 s + i — and the B
// operand for block lag d is the same column 16 d frames later, v[64 S + 16 d + lane, c]:
//   C_d[m, n] += sum_s v[64 S + 16 s + m, c] v[64 S + 16 s + 16 d + n, c]      (lag 16 d + n - m)
// is one MFMA with all four k-slots doing arithmetic, `dim` MFMAs per super-step and block lag.  (The
// Einstein-Helfand form needs the norms in a k-slot of their own and stays at 3 of 4: band32_kernels.hpp.)
//
// A unit = (particle, block lags 16 g ... 16 g + 15) runs the whole band of those block lags and writes lags
// 256 g ... 256 g + 240; a lag 256 g + 241 ... 255 has one half in this unit's last block lag and the other in the next
// unit's first: both halves are ADDED into the zeroed output (two terms into a zero: the same sum in either order).  The 16 B operands of a super-step are 64-frame windows 16 frames apart: the column
// lives in a per-wave LDS ring (8 chunks of 64 frames and a copy of the first chunk behind the last, so that a
// window never wraps) and every operand is ONE ds_read_b64 at an immediate offset — no vector instruction per
// operand (FP64 MFMAs and vector instructions do not overlap on gfx950).  Rows come in by ordinary buffer loads one
// super-step ahead (registers, then a ds_write at the top of the next super-step): nothing here is hidden from the
// compiler, no inline assembly, no hand-placed wait.
#pragma once
#include "band_kernels.hpp"

namespace ta {

constexpr int kBpChunks = 8;                        // ring: chunks of 64 frames (super-step S reads chunks S ... S + 4)
constexpr int kBpRingFrames = 64 * (kBpChunks + 1);  // + the copy of ring position 0 behind position 7

// pm: pair-major float64 slab of n_atoms * D columns.  bp_am[particle * ld_am + lag] = S[lag] / (T - lag), every lag
// < T written or added to: bp_am and *next_unit must be ZERO on entry.  grid: any number of workgroups of 64 NW threads.
template <int D, int NW>
__global__ void __launch_bounds__(64 * NW)
    k_band_bp_vacf(const double* __restrict__ pm, long pitch, int T, long n_atoms, double* __restrict__ bp_am, long ld_am,
                   unsigned long long* __restrict__ next_unit) {
    static_assert(D >= 1 && D <= 3, "a particle's columns lie in at most two column pairs");
    constexpr int DR = D < 2 ? 2 : D;  // (one column's ring is smaller than the epilogue's scratch)
    static_assert(DR * kBpRingFrames >= 2 * 16 * 32, "the epilogue's scratch reuses the ring");
    __shared__ double ringB[NW][DR][kBpRingFrames];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nblk = (T + 15) / 16, n_groups = (nblk + 15) / 16;
    const long n_units = n_atoms * n_groups;
    double(*rB)[kBpRingFrames] = ringB[wave];
    for (;;) {
        // units differ in length (the band of block lag 16 g is nblk - 16 g blocks long): a wave takes the next one when it
        // is done with its own (the counter only grows: every wave leaves the loop)
        unsigned long long taken = 0;
        if (lane == 0) taken = atomicAdd(next_unit, 1ull);
        const long u = (long)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(taken >> 32)) << 32) |
                              (unsigned)__builtin_amdgcn_readfirstlane((int)taken));
        if (u >= n_units) break;
        const long atom = __builtin_amdgcn_readfirstlane((int)(u / n_groups));
        const int g = __builtin_amdgcn_readfirstlane((int)(u - atom * n_groups)), d0 = 16 * g;
        // one buffer resource per column, cut off behind frame T - 1: frames past the end of the series read as zeros
        // (the rows behind them belong to the padding or to the next pair) and every column takes the same offsets
        __amdgpu_buffer_rsrc_t rs[D];
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const long col = (long)D * atom + c;
            rs[c] = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pm + (col >> 1) * pitch * 2 + (col & 1)), 0,
                                                      16 * T - 8 * (int)(col & 1), 0x00020000);
        }
        auto load = [&](int c, int f0) -> double {  // frames f0 ... f0 + 63 of column c, one per lane
            return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs[c], (f0 + lane) * 16, 0, 0));
        };
        const int n_super = (nblk - d0 + 3) / 4, fB = 16 * d0;
        band_d4 acc[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[d] = band_d4{0.0, 0.0, 0.0, 0.0};
        // the ring's first five chunks; chunk 5 and the first A chunk wait in registers
        double sb[D], sa[D];
#pragma unroll
        for (int c = 0; c < D; ++c) {
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const double x = load(c, fB + 64 * n);
                rB[c][64 * n + lane] = x;
                if (n == 0) rB[c][64 * kBpChunks + lane] = x;
            }
            sb[c] = load(c, fB + 64 * 5);
            sa[c] = load(c, 0);  // the A operand IS the load: lane = frame
        }
        int S = 0;
        for (bool more = true; more;) {
#pragma unroll
            for (int a = 0; a < kBpChunks; ++a) {  // S % 8 == a: ring positions are immediates
                const int pos = (a + 5) % kBpChunks;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    rB[c][64 * pos + lane] = sb[c];  // chunk S + 5 (first read by super-step S + 1)
                    if (pos == 0) rB[c][64 * kBpChunks + lane] = sb[c];
                }
                double A[D];
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    A[c] = sa[c];
                    sb[c] = load(c, fB + 64 * (S + 6));
                    sa[c] = load(c, 64 * (S + 1));
                }
                __builtin_amdgcn_sched_barrier(0);  // the requests stay here, a super-step ahead of their use
#pragma unroll
                for (int c = 0; c < D; ++c)
#pragma unroll
                    for (int d = 0; d < 16; ++d)
                        acc[d] = TA_BAND_MFMA(A[c], rB[c][(64 * a + 16 * d) % (64 * kBpChunks) + lane], acc[d]);
                if (++S == n_super) {
                    more = false;
                    break;
                }
            }
        }
        // diagonals: acc[d] register r of lane l is C_d[m = 4 r + (l >> 4)][n = l & 15], lag 16 (d0 + d) + n - m.
        // A block goes to the LDS skewed — row m, column n - m + 15 — so that a diagonal is a column: its sum is 16
        // reads without a bound (the cells no row writes stay zero), eight per half wave.
        __builtin_amdgcn_wave_barrier();
        double* blk = &rB[0][0];   // [16][32]
        double* dsum = blk + 512;  // [16][32]: the 31 diagonal sums of every block lag
#pragma unroll
        for (int q = 0; q < 8; ++q) blk[64 * q + lane] = 0.0;
        __builtin_amdgcn_wave_barrier();
        const int skew = (lane & 15) - (lane >> 4) + 15, half = lane >> 5, col = lane & 31;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
#pragma unroll
            for (int r = 0; r < 4; ++r) blk[(4 * r + (lane >> 4)) * 32 + skew - 4 * r] = acc[d][r];
            __builtin_amdgcn_wave_barrier();
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += blk[(8 * half + j) * 32 + col];
            s = band_sum_halves(s);
            if (lane < 32) dsum[d * 32 + lane] = s;
            __builtin_amdgcn_wave_barrier();
        }
        double* out = bp_am + atom * ld_am;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int off = lane + 64 * k;  // lag 256 g + off = 16 (d0 + d) + e: block lag d with e >= 0, d + 1 with e - 16
            const long lag = 256L * g + off;
            if (lag < T) {
                const int d = off >> 4, e = off & 15;
                double s = dsum[d * 32 + e + 15];
                if (d < 15 || e == 0) {
                    if (e >= 1) s += dsum[(d + 1) * 32 + e - 16 + 15];
                    out[lag] = s / (double)(T - lag);
                } else {  // the other half is the next unit's first block lag
                    unsafeAtomicAdd(out + lag, s / (double)(T - lag));
                }
            }
        }
        if (g > 0 && lane >= 1 && lane < 16) {  // ... and this unit's first block lag completes the previous unit's last 15 lags
            const long lag = 256L * g - 16 + lane;
            if (lag < T) unsafeAtomicAdd(out + lag, dsum[lane - 16 + 15] / (double)(T - lag));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace ta
