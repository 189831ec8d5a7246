// short_kernels.hpp — short trajectories (n_frames <= 64): all lags of a column in ONE LANE's registers.
//
// The quantities of direct_kernels.hpp (velocityautocorr.py:217-238 and, mathematically the same, :208-215 with
// tidynamics.acf; viscosity.py:201-233) for trajectories so short that neither a 512-point transform per column
// pair (wfft.hpp: 29 ms per 12 GB at 32 frames with the by-particle array) nor a workgroup per particle
// (direct_kernels.hpp: 116 ms) fits the problem: millions of particles, a few dozen frames.
//
// A lane owns one COLUMN c = atom * D + d: its n_frames values x[0 .. T) in registers (an 8-byte load per frame from
// the pair-major slab: element (t, c) at ((c >> 1) pitch + t) 2 + (c & 1) -- the two lanes of a pair ask for the two
// halves of a row in the same instruction, and the 8 rows of a 128-byte line in 8 consecutive ones; Helfand:
// P = (m v) x formed on the way in, the reference's order) and one accumulator per lag.  The double loop
//     for j < T:  for i <= j:  acc[j - i] += x[i] x[j]              (Helfand: i < j, (x[i] - x[j])^2: difference first)
// is unrolled in full, so every index is a register name; the trip over j ends at the run-time n_frames (a scalar
// branch per j), so the work is exactly T (T + 1) / 2 terms per column.  No LDS, no barrier inside the loop.
// A wave takes 64 adjacent columns (D = 3: 63 = 21 particles).  The vector ALU can name 256 registers (beyond them the
// compiler parks values in accumulation registers and moves them back and forth: 1.6 moves per FMA at 64 frames): up to
// 32 frames x and all accumulators fit; beyond, the lag sums alone are done in two blocks of 32 lags.
// Lag sums alone (BP = false), up to 32 frames: the accumulators are never reset -- a lane adds up all its columns; one
// wave reduction at the end.  Otherwise, per tile (and block of lags), eight lags at a time (one basic block: the
// latencies of its LDS reads, lane shifts and stores overlap): with the by-particle array (BP) a particle's D lanes are
// added by lane shifts and the d = 0 lane stores by_particle[lag, atom] STRAIGHT into the caller's (n_frames, ld) array
// (adjacent lanes, adjacent particles: no atom-major scratch, no transposition); every lane (in the two-block form every
// second, after one lane exchange) adds its accumulators into cells of its own in the LDS that live across the wave's
// tiles (tot[wave][lag][cell]).  Either way partial[wave][lag] is added in a fixed order by k_sum_partials --
// deterministic for a given grid.
// (A lane per PARTICLE instead -- whole 16-byte pair rows, no lane shifts, 512-byte stores -- needs two columns and the
// accumulators at once: it spills inside the FMA loop, 5 - 26 ms where this form takes 2 - 5.)
//
// Roofline: HBM.  Bytes per launch = T A D 8 read (+ T A 8 written with the by-particle array); arithmetic
// T (T + 1) / 2 FMAs per column is 0.7 ms (32 frames) ... 1.3 ms (64 frames) per 12 GB of input at the FP64 peak.
#pragma once
#include <hip/hip_runtime.h>

#include "direct_kernels.hpp"

namespace ta {

constexpr int kShortWaves = 4;  // waves per workgroup (one per SIMD)

// the compiled variants (measured, profiles/r06_short.txt): up to 32 frames x and every accumulator fit 256 registers at two
// waves per SIMD; up to 64 frames the lag sums alone run two blocks of 32 lags (two waves per SIMD, two lanes per lag-sum cell:
// 128 KiB of cells would not fit twice), with the by-particle array all 64 at one wave per SIMD (5.3 against 6.0 ms per 12 GB;
// a 48-frame variant of the one-block form spills and gains nothing)
template <int TMAX, bool BP>
struct ShortCfg {
    static constexpr bool BLOCKED = TMAX > 32 && !BP;
    static constexpr int WAVES_PER_SIMD = TMAX > 32 && BP ? 1 : 2;
    static constexpr int LB = BLOCKED ? 32 : TMAX;  // lags per block
    static constexpr int RQ = BLOCKED ? 2 : 1;      // lanes per lag-sum cell
    static constexpr bool PERSIST = !BP && !BLOCKED;  // lag sums alone, one block: the accumulators are never reset
    static constexpr size_t kLds = PERSIST ? 0 : sizeof(double) * (64 / RQ) * kShortWaves * TMAX;
};

template <int D>
constexpr int short_cols_per_wave() { return D == 3 ? 63 : 64; }

// partial: [gridDim.x * kShortWaves][T]; bp: (T, ld_bp) with BP; factor: 1 (VACF) or scale / D (Helfand)
template <int TMAX, int MODE, int D, bool BP>
__global__ void __launch_bounds__(64 * kShortWaves)
    __attribute__((amdgpu_waves_per_eu(ShortCfg<TMAX, BP>::WAVES_PER_SIMD, ShortCfg<TMAX, BP>::WAVES_PER_SIMD)))
    k_short(const double* __restrict__ vel, const double* __restrict__ pos, const double* __restrict__ masses, long pitch,
            int T, long n_atoms, double factor, double* __restrict__ bp, long ld_bp, double* __restrict__ partial) {
    using Cfg = ShortCfg<TMAX, BP>;
    constexpr int CW = short_cols_per_wave<D>(), AW = CW / D, LB = Cfg::LB, RQ = Cfg::RQ, CELLS = 64 / RQ;
    constexpr bool PERSIST = Cfg::PERSIST;
    __shared__ double rn[TMAX];            // factor / (T - k)
    extern __shared__ double short_lds[];  // !PERSIST: tot[wave][k][cell]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)threadIdx.x < TMAX) rn[threadIdx.x] = (int)threadIdx.x < T ? factor / (double)(T - (int)threadIdx.x) : 0.0;
    __syncthreads();
    const long n_tiles = (n_atoms + AW - 1) / AW;
    const int d = lane % D;
    double* tot = short_lds + (wave * TMAX) * CELLS + lane / RQ;
    const bool owns_cell = lane % RQ == 0;
    if (!PERSIST && owns_cell) {
#pragma unroll
        for (int k = 0; k < TMAX; ++k) tot[k * CELLS] = 0.0;
    }
    double acc[LB];
#pragma unroll
    for (int k = 0; k < LB; ++k) acc[k] = 0.0;

    for (long tile = (long)blockIdx.x * kShortWaves + wave; tile < n_tiles; tile += (long)gridDim.x * kShortWaves) {
        const long atom = tile * AW + lane / D;
        const bool active = lane < CW && atom < n_atoms;
        const long c = atom * D + d;
        double x[TMAX];
        if (active) {
            const double* pv = vel + ((c >> 1) * pitch) * 2 + (c & 1);
            if (MODE == MODE_HELFAND) {
                const double* pp = pos + ((c >> 1) * pitch) * 2 + (c & 1);
                const double m = masses[atom];
#pragma unroll
                for (int t8 = 0; t8 < TMAX; t8 += 8)
                    if (t8 < T) {  // pitch is n_frames rounded up to 8: the rows exist
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[t8 + u] = (m * pv[(t8 + u) * 2]) * pp[(t8 + u) * 2];
                    }
            } else {
#pragma unroll
                for (int t8 = 0; t8 < TMAX; t8 += 8)
                    if (t8 < T) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[t8 + u] = pv[(t8 + u) * 2];
                    }
            }
        } else {
#pragma unroll
            for (int t = 0; t < TMAX; ++t) x[t] = 0.0;
        }
#pragma unroll
        for (int k0 = 0; k0 < TMAX; k0 += LB)
            if (k0 < T) {
                if (!PERSIST) {
#pragma unroll
                    for (int k = 0; k < LB; ++k) acc[k] = 0.0;
                }
#pragma unroll
                for (int j = k0; j < TMAX; ++j)
                    if (j < T) {
#pragma unroll
                        for (int i = 0; i <= j; ++i) {
                            const int k = j - i - k0;  // lag within the block
                            if (k < 0 || k >= LB) continue;
                            if (MODE == MODE_HELFAND) {
                                if (i == j) continue;
                                const double df = x[i] - x[j];
                                acc[k] = __builtin_fma(df, df, acc[k]);
                            } else {
                                acc[k] = __builtin_fma(x[i], x[j], acc[k]);
                            }
                        }
                    }
                if (!PERSIST) {
#pragma unroll
                    for (int k8 = 0; k8 < LB; k8 += 8)
                        if (k0 + k8 < T) {
                            // (the table cells are re-read per tile: hoisted out of the tile loop, the TMAX values would
                            // take 2 TMAX registers for the length of the kernel)
                            int kk = k0 + k8;
                            asm volatile("" : "+v"(kk));
                            double cell[8], r[8], v[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                cell[u] = tot[(k0 + k8 + u) * CELLS];
                                if (RQ == 2) cell[u] += __shfl_xor(acc[k8 + u], 1, 64);
                                if (BP) {
                                    r[u] = rn[kk + u];
                                    v[u] = acc[k8 + u];
                                    if (D >= 2) v[u] += __shfl_down(acc[k8 + u], 1, 64);
                                    if (D == 3) v[u] += __shfl_down(acc[k8 + u], 2, 64);
                                }
                            }
                            if (owns_cell) {
#pragma unroll
                                for (int u = 0; u < 8; ++u) tot[(k0 + k8 + u) * CELLS] = cell[u] + acc[k8 + u];
                            }
                            if (BP && active && d == 0) {
                                double* row = bp + (long)(k0 + k8) * ld_bp + atom;
                                if (k0 + k8 + 8 <= T) {
#pragma unroll
                                    for (int u = 0; u < 8; ++u) row[(long)u * ld_bp] = v[u] * r[u];
                                } else {
#pragma unroll
                                    for (int u = 0; u < 8; ++u)
                                        if (k0 + k8 + u < T) row[(long)u * ld_bp] = v[u] * r[u];
                                }
                            }
                        }
                }
            }
    }
    double* prow = partial + ((long)blockIdx.x * kShortWaves + wave) * T;
#pragma unroll
    for (int k = 0; k < TMAX; ++k)
        if (k < T) {
            double s = 0.0;
            if constexpr (PERSIST) s = acc[k];
            else if (lane < CELLS) s = short_lds[(wave * TMAX + k) * CELLS + lane];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) prow[k] = s * rn[k];
        }
}

}  // namespace ta
