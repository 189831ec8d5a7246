// band32tp_kernels.hpp — Einstein-Helfand on the FP32 matrix cores with the k-slots filled from the TIME axis:
// BASELINE configs[4]'s "float32 path", with and without results.visc_by_particle
// (/root/reference/transport_analysis/viscosity.py:201-233; P = (m v) x rounded ONCE to float32, float32 products,
// float64 accumulation: the "direct_f32" contract, 2e-6 of the series' scale).
//
// The structure is bandbp_kernels.hpp's k_band_bp_helf (read that file first): a unit = 16 block lags of one particle
// (or, for lag sums alone, of a run of particles summed in the same accumulators); the A operand of column c is 64
// consecutive frames — lane = frame, straight from the load — the B operand for block lag d the same column 16 d frames
// later, ONE ds_read_b32 at an immediate offset of a per-wave LDS ring that holds the columns centred on a reference row
// (moved every 256 frames) and, as one more column, their squared norms; |b|^2 is one vector add per block lag and
// super-step, |a|^2 one running sum per lane.  All four k-slots of v_mfma_f32_16x16x4_f32 do arithmetic: 3 MFMAs per
// super-step and block lag at dim = 3, where band32_kernels.hpp's column-packed forms spend 4 (a slot of four for the
// norms) and prepare every window fragment with vector instructions.
// What float32 adds: the accumulators (and the norm sums) hold at most kBand32tpFlush super-steps — 12 products per
// element each — and go through Band32Diag's diagonal sums into float64 (five doubles per lane); rows are requested
// TWO super-steps ahead (a super-step is 1536 matrix cycles, shorter than a memory round trip).
#pragma once
#include <utility>

#include "band_common.hpp"
#include "bandbp_kernels.hpp"

namespace ta {

#ifndef TA_BAND32TP_FLUSH
#define TA_BAND32TP_FLUSH 64
#endif
#ifndef TA_BAND32TP_ABL  // timing ablations (wrong results), bit mask: 1 no norm-column adds, 2 no re-centring, 4 no flush in the loop,
#define TA_BAND32TP_ABL 0  // 8 no MFMAs, 16 no window reads (operands = the A registers)
#endif
constexpr int kBand32tpFlush = TA_BAND32TP_FLUSH;  // super-steps between flushes (a multiple of 8)

__device__ __forceinline__ float band32_first_lane(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

// P32: pair-major float32 product slab (8-byte rows) of n_atoms * D columns.  grid: any number of workgroups of 64 NW threads.
// !LAGS: bp_am[particle * ld_am + lag] = factor * sum_i sum_d (dP)^2 / (T - lag), lag 0 exactly 0; bp_am and next_unit[0 .. 7]
// must be ZERO on entry (two units add their halves of the lags at a unit boundary).
// LAGS: partial[(g * n_pb + block) * kBandPartial + q] = the sum for lag 256 g - 15 + q over the unit's `per_unit`
// particles (every element written; k_bandbp_gather adds them in a fixed order).
template <int D, int NW, bool LAGS>
__global__ void __launch_bounds__(64 * NW)
    k_band32_tp(const float* __restrict__ P32, long pitch, int T, long n_atoms, double factor, double* __restrict__ bp_am, long ld_am,
                unsigned long long* __restrict__ next_unit, int per_unit, double* __restrict__ partial) {
    static_assert(D >= 1 && D <= 3, "a particle's columns lie in at most two column pairs");
    static_assert(kBand32tpFlush % kBpChunks == 0, "flushes happen between passes of the ring");
    constexpr int NR = D + 1;  // rings: the centred columns and the rows' squared norms
    __shared__ float ringB[NW][NR][kBpRingFrames];
    __shared__ __attribute__((aligned(16))) float diag[NW][32 * 17 + 16 * 32 + 16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nblk = (T + 15) / 16, n_groups = (nblk + 15) / 16;
    const long n_pb = LAGS ? (n_atoms + per_unit - 1) / per_unit : n_atoms;
    float(*rB)[kBpRingFrames] = ringB[wave];
    Band32Diag sums;
    sums.init(diag[wave], lane);
    float* nas = diag[wave] + 32 * 17 + 16 * 32;  // NA[16] of a flush
    for (;;) {
        long pb;
        int g;
        if (!band_take_unit(next_unit, lane, n_groups, n_pb, &pb, &g)) break;
        const int d0 = 16 * g;
        const int n_super = (nblk - d0 + 3) / 4, fB = 16 * d0;
        band_f4 acc[16];
        float nbacc[16], na = 0.0f;
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[d] = band_f4{0.0f, 0.0f, 0.0f, 0.0f}, nbacc[d] = 0.0f;
        sums.clear();
        int since = 0;  // passes of the ring since the last flush
        // float32 sums -> float64: (a - b)^2 summed = NA[m] + NB_d[n] - 2 acc_d[m][n], formed in float32 (each term a sum of
        // at most kBand32tpFlush x 12 squares), its 31 diagonals added up and accumulated in float64
        auto flush = [&]() __attribute__((always_inline)) {
            const float tot = band32_sum_rows(na);  // every lane: NA[lane & 15]
            if (lane < 16) nas[lane] = tot;
            TA_LDS_ORDER();
            const band_f4 na_m = *reinterpret_cast<const band_f4*>(nas + 4 * (lane >> 4));  // rows m = 4 (lane >> 4) + r
            float nbd[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) nbd[d] = band32_sum_rows(nbacc[d]), nbacc[d] = 0.0f;  // NB_d[lane & 15]
#pragma unroll
            for (int d = 0; d < 16; ++d)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) acc[d][r4] = __builtin_fmaf(-2.0f, acc[d][r4], na_m[r4] + nbd[d]);
            na = 0.0f;
            sums.flush(acc);
        };
        const long atom_lo = LAGS ? pb * per_unit : pb, atom_hi = LAGS ? (atom_lo + per_unit < n_atoms ? atom_lo + per_unit : n_atoms) : pb + 1;
        for (long atom = atom_lo; atom < atom_hi; ++atom) {
            // one buffer resource per column, cut off behind frame T - 1: frames past the end of the series read as zeros
            __amdgpu_buffer_rsrc_t rs[D];
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const long col = (long)D * atom + c;
                rs[c] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P32 + (col >> 1) * pitch * 2 + (col & 1)), 0,
                                                          8 * T - 4 * (int)(col & 1), 0x00020000);
            }
            auto load = [&](int c, int f0) __attribute__((always_inline)) -> float {  // frames f0 ... f0 + 63 of column c, one per lane
                return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs[c], (f0 + lane) * 8, 0, 0));
            };
            float sa[2][D], sb[2][D], r[D];
            // raw rows of chunk `pos` (frames fbase ...) -> centred columns and their norms in the ring
            auto write_chunk = [&]<bool TAIL>(int pos, int fbase, const float(&raw)[D]) __attribute__((always_inline)) {
                float nb = 0.0f;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    float b = raw[c] - r[c];
                    if (TAIL && !(fbase + lane < T)) b = 0.0f;
                    rB[c][64 * pos + lane] = b;
                    if (pos == 0) rB[c][64 * kBpChunks + lane] = b;
                    nb = __builtin_fmaf(b, b, nb);
                }
                rB[D][64 * pos + lane] = nb;
                if (pos == 0) rB[D][64 * kBpChunks + lane] = nb;
            };
#pragma unroll
            for (int c = 0; c < D; ++c) sa[0][c] = load(c, 0), sa[1][c] = load(c, 64), r[c] = band32_first_lane(sa[0][c]);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                float x[D];
#pragma unroll
                for (int c = 0; c < D; ++c) x[c] = load(c, fB + 64 * n);
                write_chunk.template operator()<true>(n, fB + 64 * n, x);
            }
#pragma unroll
            for (int c = 0; c < D; ++c) sb[0][c] = load(c, fB + 64 * 5), sb[1][c] = load(c, fB + 64 * 6);
            TA_LDS_ORDER();
            int S = 0;
            // one super-step; a == S % 8 (a constant once unrolled)
            auto body = [&]<bool TAIL, int a>() __attribute__((always_inline)) {
                if (a == 0 && !(TA_BAND32TP_ABL & 4) && ++since == kBand32tpFlush / kBpChunks) {
                    flush();
                    since = 0;
                }
                // The ring is written chunk by chunk and read window by window: a lane reads what OTHER lanes wrote.  The compiler
                // sees each thread's own addresses only (a write at 64 q + lane and a read at 16 x + lane never alias for it) and
                // would move reads across the writes: wave_barrier() pins its order on both sides of the writes (no instruction:
                // LDS operations of one wave complete in order).
                TA_LDS_ORDER();
                if (a % 4 == 0 && S != 0 && !(TA_BAND32TP_ABL & 2)) {  // a new pass: a new reference row, which the ring's live chunks follow
                    float delta[D];
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        const float rn = band32_first_lane(sa[a & 1][c]);
                        delta[c] = r[c] - rn, r[c] = rn;
                    }
#pragma unroll
                    for (int q = 0; q <= kBpChunks; ++q) {
                        const int k = ((q % kBpChunks) - a + kBpChunks) % kBpChunks;  // position q holds chunk S + k
                        if (k > 4) continue;  // (k = 5 ... 7: chunks behind the window, about to be overwritten)
                        if (q == kBpChunks && k == 0) continue;  // the copy of position 0 serves windows that start in position 7: chunk S + k - 1
                        const int fbase = fB + 64 * (S + k);
                        float nb = 0.0f;
#pragma unroll
                        for (int c = 0; c < D; ++c) {
                            float b = rB[c][64 * q + lane] + delta[c];
                            if (TAIL && !(fbase + lane < T)) b = 0.0f;
                            rB[c][64 * q + lane] = b;
                            nb = __builtin_fmaf(b, b, nb);
                        }
                        rB[D][64 * q + lane] = nb;
                    }
                }
                write_chunk.template operator()<TAIL>((a + 5) % kBpChunks, fB + 64 * (S + 5), sb[a & 1]);  // chunk S + 5
                TA_LDS_ORDER();
                float A[D], asq = 0.0f;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    A[c] = sa[a & 1][c] - r[c];
                    if (TAIL && !(64 * S + lane < T)) A[c] = 0.0f;
                    asq = __builtin_fmaf(A[c], A[c], asq);
                }
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    sb[a & 1][c] = load(c, fB + 64 * (S + 7));
                    sa[a & 1][c] = load(c, 64 * (S + 2));
                }
                __builtin_amdgcn_sched_barrier(0);  // the requests stay here, two super-steps ahead of their use
                if constexpr (!TAIL) {
                    na += asq;
                    // block lag by block lag: a window of every column and of the norms, their MFMAs, the norm add.  (Measured at
                    // 20000 x 25000 x 3, profiles/r05_band32tp_ablation.txt: reading a column's 16 windows as one batch ahead of its
                    // MFMAs 258 ms, this order 256; requesting block lag d + 1's windows before the MFMAs of d with the scheduler
                    // fenced per block lag 309.  The SLP vectoriser is off for this file (csrc/Makefile): it packed the norm adds into
                    // v_pk_add_f32 through register moves and kept their operands alive across super-steps — 256 registers and
                    // spills; without it 215 - 223 and 245 ms, and three waves per SIMD fit without a spill in the loop: 238.)
#pragma unroll
                    for (int d = 0; d < 16; ++d) {
                        const int w = (64 * a + 16 * d) % (64 * kBpChunks) + lane;
                        if (!(TA_BAND32TP_ABL & 8)) {
#pragma unroll
                            for (int c = 0; c < D; ++c) acc[d] = TA_BAND32_MFMA(A[c], (TA_BAND32TP_ABL & 16) ? A[(c + d) % D] : rB[c][w], acc[d]);
                        }
                        if (!(TA_BAND32TP_ABL & 17)) nbacc[d] += rB[D][w];
                    }
                } else {
                    const float Ah = -0.5f * asq;
#pragma unroll
                    for (int d = 0; d < 16; ++d) {
                        const int j0 = 64 * S + 16 * (d0 + d);  // the window's first frame
                        if (j0 >= T) continue;                  // (wave-uniform) nothing of this block lag is left
                        const int w = (64 * a + 16 * d) % (64 * kBpChunks) + lane;
#pragma unroll
                        for (int c = 0; c < D; ++c) acc[d] = TA_BAND32_MFMA(A[c], rB[c][w], acc[d]);
                        nbacc[d] += rB[D][w];
                        acc[d] = TA_BAND32_MFMA(Ah, j0 + lane < T ? 1.0f : 0.0f, acc[d]);  // -|a|^2 / 2 where the pair's later frame exists
                    }
                }
            };
            // whole passes of 8 super-steps none of whose requests reaches frame T: S + 7 is the farthest chunk touched
            const int n_ok = (T - fB) / 64 - 7;
            const int S_bulk = (n_ok < n_super ? (n_ok > 0 ? n_ok : 0) : n_super) / kBpChunks * kBpChunks;
            // (a is a template argument: the two-deep request registers are indexed by it)
            auto pass_bulk = [&]<int... a>(std::integer_sequence<int, a...>) __attribute__((always_inline)) { ((body.template operator()<false, a>(), ++S), ...); };
            auto pass_tail = [&]<int... a>(std::integer_sequence<int, a...>) __attribute__((always_inline)) {
                return ((body.template operator()<true, a>(), ++S != n_super) && ...);  // false: the unit's last super-step is done
            };
            while (S < S_bulk) pass_bulk(std::make_integer_sequence<int, kBpChunks>{});
            if (S < n_super)
                while (pass_tail(std::make_integer_sequence<int, kBpChunks>{})) {
                }
            TA_LDS_ORDER();  // (the next particle's rows overwrite the ring)
        }
        flush();
        // sums.s[k]: lag slot q = lane + 64 k, lag 256 g - 15 + q: diagonal e >= 0 of block lag d plus e - 16 of d + 1
        if constexpr (LAGS) {
            double* out = partial + ((long)g * n_pb + pb) * kBandPartial;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int q = lane + 64 * k;
                if (q < kBandPartial) out[q] = q - 15 <= 255 ? sums.s[k] : 0.0;
            }
        } else {
            double* out = bp_am + pb * ld_am;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int off = lane + 64 * k - 15;
                const long lag = 256L * g + off;
                if (off <= 255 && lag >= 0 && lag < T) {
                    const double val = factor * sums.s[k] / (double)(T - lag);
                    if (off >= 0 && off <= 240) out[lag] = lag == 0 ? 0.0 : val;  // both halves in this unit
                    else unsafeAtomicAdd(out + lag, val);                       // a lag at a unit boundary: one half each
                }
            }
        }
    }
}

}  // namespace ta
