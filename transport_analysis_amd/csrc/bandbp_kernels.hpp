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
#include "band_common.hpp"

namespace ta {

// Lanes of a wave exchange data through the LDS here (ring: written chunk by chunk, read window by window; epilogues).  The
// hardware completes one wave's LDS operations in order; the COMPILER orders only what may alias within a thread (a write at
// 64 q + lane and a read at 16 x + lane never do) — fences for its IR passes, wave_barrier for its schedulers; no instruction.
#define TA_LDS_ORDER()                                              \
    do {                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
        __builtin_amdgcn_wave_barrier();                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
    } while (0)

// Work distribution of the time-packed kernels: a unit = (block of particles pb, block lag group g).  Units differ in length
// (the band of group g is nblk - 16 g blocks long), so waves TAKE them: one counter per XCD (next_unit[0 .. 7]; workgroups are
// dealt round-robin to the XCDs, blockIdx % 8), and XCD x takes the particle blocks x, x + 8, ...: the ~nblk/16 units of a
// particle, which read the same rows, run at about the same time behind ONE L2.  The counters only grow: every wave leaves.
// (Longest units first — all blocks at group 0, then group 1, ... — loses the shared rows and is 2 - 5 % slower.)
#ifndef TA_BANDBP_XCD
#define TA_BANDBP_XCD 1
#endif
constexpr int kBpCounters = 8;
__device__ __forceinline__ bool band_take_unit(unsigned long long* next_unit, int lane, int n_groups, long n_pb, long* pb, int* g) {
    const int n_lab = TA_BANDBP_XCD ? ((int)gridDim.x < kBpCounters ? (int)gridDim.x : kBpCounters) : 1;
    const int x = (int)(blockIdx.x % (unsigned)n_lab);
    unsigned long long taken = 0;
    if (lane == 0) taken = atomicAdd(next_unit + x, 1ull);
    const long v = (long)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(taken >> 32)) << 32) |
                          (unsigned)__builtin_amdgcn_readfirstlane((int)taken));
    const long pbl = v / n_groups;
    *pb = x + (long)n_lab * pbl;
    *g = (int)(v - pbl * n_groups);
    return *pb < n_pb;
}

constexpr int kBpChunks = 8;                        // ring: chunks of 64 frames (super-step S reads chunks S ... S + 4)
constexpr int kBpRingFrames = 64 * (kBpChunks + 1);  // + the copy of ring position 0 behind position 7

// pm: pair-major float64 slab of n_atoms * D columns.  bp_am[particle * ld_am + lag] = S[lag] / (T - lag), every lag
// < T written or added to: bp_am and next_unit[0 .. 7] must be ZERO on entry.  grid: any number of workgroups of 64 NW threads.
// LAGS (lag sums alone): a unit is a block lag group of `per_unit` consecutive particles summed in the same accumulators;
// it writes partial[(g * n_pb + block) * kBandPartial + q] = the sum for lag 256 g - 15 + q (k_bandbp_gather adds them up).
template <int D, int NW, bool LAGS>
__global__ void __launch_bounds__(64 * NW)
    k_band_bp_vacf(const double* __restrict__ pm, long pitch, int T, long n_atoms, double* __restrict__ bp_am, long ld_am,
                   unsigned long long* __restrict__ next_unit, int per_unit, double* __restrict__ partial) {
    static_assert(D >= 1 && D <= 3, "a particle's columns lie in at most two column pairs");
    constexpr int DR = D < 2 ? 2 : D;  // (one column's ring is smaller than the epilogue's scratch)
    static_assert(DR * kBpRingFrames >= 2 * 16 * 32, "the epilogue's scratch reuses the ring");
    __shared__ double ringB[NW][DR][kBpRingFrames];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nblk = (T + 15) / 16, n_groups = (nblk + 15) / 16;
    const long n_pb = LAGS ? (n_atoms + per_unit - 1) / per_unit : n_atoms;
    double(*rB)[kBpRingFrames] = ringB[wave];
    for (;;) {
        long pb;
        int g;
        if (!band_take_unit(next_unit, lane, n_groups, n_pb, &pb, &g)) break;
        const int d0 = 16 * g;
        const int n_super = (nblk - d0 + 3) / 4, fB = 16 * d0;
        band_d4 acc[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[d] = band_d4{0.0, 0.0, 0.0, 0.0};
        const long atom_lo = LAGS ? pb * per_unit : pb, atom_hi = LAGS ? (atom_lo + per_unit < n_atoms ? atom_lo + per_unit : n_atoms) : pb + 1;
        for (long atom = atom_lo; atom < atom_hi; ++atom) {
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
        TA_LDS_ORDER();
        int S = 0;
        // one super-step; a == S % 8 (a constant once unrolled: ring positions are immediates).  TAIL: the last super-steps of
        // a unit skip the block lags whose whole window lies behind the series (the band's triangular end)
        auto body = [&]<bool TAIL>(int a) __attribute__((always_inline)) {
            const int pos = (a + 5) % kBpChunks;
            // The ring is written chunk by chunk and read window by window: a lane reads what OTHER lanes wrote.  The compiler
            // sees each thread's own addresses only (a write at 64 q + lane and a read at 16 x + lane never alias for it) and
            // may move reads across the writes: TA_LDS_ORDER() pins its order on both sides of the writes (no instruction:
            // LDS operations of one wave complete in order).
            TA_LDS_ORDER();
#pragma unroll
            for (int c = 0; c < D; ++c) {
                rB[c][64 * pos + lane] = sb[c];  // chunk S + 5 (first read by super-step S + 1)
                if (pos == 0) rB[c][64 * kBpChunks + lane] = sb[c];
            }
            TA_LDS_ORDER();
            double A[D];
#pragma unroll
            for (int c = 0; c < D; ++c) {
                A[c] = sa[c];
                sb[c] = load(c, fB + 64 * (S + 6));
                sa[c] = load(c, 64 * (S + 1));
            }
            __builtin_amdgcn_sched_barrier(0);  // the requests stay here, a super-step ahead of their use
#pragma unroll
            for (int d = 0; d < 16; ++d) {  // block lag by block lag
                if (TAIL && 64 * S + 16 * (d0 + d) >= T) continue;  // (wave-uniform)
#pragma unroll
                for (int c = 0; c < D; ++c)
                    acc[d] = TA_BAND_MFMA(A[c], rB[c][(64 * a + 16 * d) % (64 * kBpChunks) + lane], acc[d]);
            }
        };
        // whole passes of 8 super-steps in which every block lag's window still starts inside the series
        const int n_ok = (T - fB - 240 + 63) / 64;  // super-steps S with 64 S + 16 (d0 + 15) < T
        const int S_bulk = (n_ok < n_super ? (n_ok > 0 ? n_ok : 0) : n_super) / kBpChunks * kBpChunks;
        while (S < S_bulk) {
#pragma unroll
            for (int a = 0; a < kBpChunks; ++a) {
                body.template operator()<false>(a);
                ++S;
            }
        }
        for (bool more = S < n_super; more;) {
#pragma unroll
            for (int a = 0; a < kBpChunks; ++a) {
                body.template operator()<true>(a);
                if (++S == n_super) {
                    more = false;
                    break;
                }
            }
        }
        TA_LDS_ORDER();  // (the next particle's rows overwrite the ring)
        }
        // diagonals: acc[d] register r of lane l is C_d[m = 4 r + (l >> 4)][n = l & 15], lag 16 (d0 + d) + n - m.
        // A block goes to the LDS skewed — row m, column n - m + 15 — so that a diagonal is a column: its sum is 16
        // reads without a bound (the cells no row writes stay zero), eight per half wave.
        TA_LDS_ORDER();
        double* blk = &rB[0][0];   // [16][32]
        double* dsum = blk + 512;  // [16][32]: the 31 diagonal sums of every block lag
#pragma unroll
        for (int q = 0; q < 8; ++q) blk[64 * q + lane] = 0.0;
        TA_LDS_ORDER();
        const int skew = (lane & 15) - (lane >> 4) + 15, half = lane >> 5, col = lane & 31;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
#pragma unroll
            for (int r = 0; r < 4; ++r) blk[(4 * r + (lane >> 4)) * 32 + skew - 4 * r] = acc[d][r];
            TA_LDS_ORDER();
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += blk[(8 * half + j) * 32 + col];
            s = band_sum_halves(s);
            if (lane < 32) dsum[d * 32 + lane] = s;
            TA_LDS_ORDER();
        }
        if constexpr (LAGS) {
            double* out = partial + ((long)g * n_pb + pb) * kBandPartial;
            for (int q = lane; q < kBandPartial; q += 64) {
                // slot q is lag offset q - 15 = 16 d + e: (d, e >= 0) and (d + 1, e - 16)
                const int off = q - 15;
                const int d = off >= 0 ? off >> 4 : -1, e = off - 16 * d;  // e in [0, 15] (off < 0: 1..15)
                double s = 0.0;
                if (off <= 255) {
                    if (d >= 0) s = dsum[d * 32 + e + 15];
                    if (e >= 1 && d + 1 <= 15) s += dsum[(d + 1) * 32 + e - 16 + 15];
                }
                out[q] = s;
            }
            TA_LDS_ORDER();
            continue;
        }
        double* out = bp_am + pb * ld_am;
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
        TA_LDS_ORDER();
    }
}

// ---- Einstein-Helfand WITH the by-particle array (float64) ---------------------------------------------------
// results.visc_by_particle of /root/reference/transport_analysis/viscosity.py:201-233, the class default:
//   H[k, n] = factor * sum_{i < T-k} sum_d (P[i, n, d] - P[i+k, n, d])^2 / (T - k),   P = (m v) x (the product slab).
// Same units, ring and k-slots as k_band_bp_vacf; (a - b)^2 = |a|^2 + |b|^2 - 2 a.b with a = P[i] - r, b = P[j] - r for a
// reference row r near frame i (the first A frame of every pass of four super-steps, 256 frames: the three terms stay of
// the size of the result for all but the shortest lags of a smooth series, as in band_kernels.hpp):
//  * the ring holds the CENTRED columns and, as one more column, the rows' squared norms |b|^2; a new reference moves
//    the ring's contents by r_old - r_new;
//  * a.b: dim MFMAs per super-step and block lag, all four k-slots doing arithmetic (the norms need no slot here);
//  * |b|^2 does not depend on the A row: per block lag ONE vector add of the norm column's window (nbacc), reduced
//    over the k-slots at the end; |a|^2 does not depend on the block lag: one running sum per lane (na);
//  * super-steps that touch frames >= T ("tail": the last ~14 of a unit) zero the centred values of such frames and
//    put |a|^2 through the product with the B side's validity instead (-|a|^2/2 x valid: one more MFMA per block
//    lag), and skip block lags whose whole window lies behind the series.
__device__ __forceinline__ double band_first_lane(double x) {
    const band_u2 w = __builtin_bit_cast(band_u2, x);
    return __builtin_bit_cast(double, band_u2{(unsigned)__builtin_amdgcn_readfirstlane((int)w.x),
                                              (unsigned)__builtin_amdgcn_readfirstlane((int)w.y)});
}

// P: pair-major float64 product slab of n_atoms * D columns.  grid: any number of workgroups of 64 NW threads.
// !LAGS: bp_am[particle * ld_am + lag] = factor * H-sum / (T - lag), lag 0 exactly 0; bp_am and next_unit[0 .. 7] must be ZERO on entry.
// LAGS (lag sums alone, results.visc_by_particle not asked for): a unit is a block lag group of `per_unit` consecutive
// particles, whose sums stay in the same accumulators; it writes partial[(g * n_pb + block) * kBandPartial + q] = the sum for
// lag 256 g - 15 + q over its particles (every element written; k_bandbp_gather adds them in a fixed order).
template <int D, int NW, bool LAGS>
__global__ void __launch_bounds__(64 * NW)
    k_band_bp_helf(const double* __restrict__ P, long pitch, int T, long n_atoms, double factor, double* __restrict__ bp_am,
                   long ld_am, unsigned long long* __restrict__ next_unit, int per_unit, double* __restrict__ partial) {
    static_assert(D >= 1 && D <= 3, "a particle's columns lie in at most two column pairs");
    constexpr int NR = D + 1;  // rings: the centred columns and the rows' squared norms
    static_assert(NR * kBpRingFrames >= 2 * 16 * 32 + 16, "the epilogue's scratch reuses the ring");
    __shared__ double ringB[NW][NR][kBpRingFrames];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nblk = (T + 15) / 16, n_groups = (nblk + 15) / 16;
    const long n_pb = LAGS ? (n_atoms + per_unit - 1) / per_unit : n_atoms;
    double(*rB)[kBpRingFrames] = ringB[wave];
    for (;;) {
        long pb;
        int g;
        if (!band_take_unit(next_unit, lane, n_groups, n_pb, &pb, &g)) break;
        const int d0 = 16 * g;
        const int n_super = (nblk - d0 + 3) / 4, fB = 16 * d0;
        band_d4 acc[16];
        double nbacc[16], na = 0.0;
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[d] = band_d4{0.0, 0.0, 0.0, 0.0}, nbacc[d] = 0.0;
        const long atom_lo = LAGS ? pb * per_unit : pb, atom_hi = LAGS ? (atom_lo + per_unit < n_atoms ? atom_lo + per_unit : n_atoms) : pb + 1;
        for (long atom = atom_lo; atom < atom_hi; ++atom) {
        __amdgpu_buffer_rsrc_t rs[D];
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const long col = (long)D * atom + c;
            rs[c] = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P + (col >> 1) * pitch * 2 + (col & 1)), 0,
                                                      16 * T - 8 * (int)(col & 1), 0x00020000);
        }
        auto load = [&](int c, int f0) -> double {  // frames f0 ... f0 + 63 of column c, one per lane; zeros behind the series
            return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs[c], (f0 + lane) * 16, 0, 0));
        };
        double sa[D], sb[D], r[D];
        // raw rows of chunk `pos` (frames fbase ...) -> centred columns and their norms in the ring
        auto write_chunk = [&]<bool TAIL>(int pos, int fbase, const double(&raw)[D]) {
            double nb = 0.0;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double b = raw[c] - r[c];
                if (TAIL && !(fbase + lane < T)) b = 0.0;
                rB[c][64 * pos + lane] = b;
                if (pos == 0) rB[c][64 * kBpChunks + lane] = b;
                nb = __builtin_fma(b, b, nb);
            }
            rB[D][64 * pos + lane] = nb;
            if (pos == 0) rB[D][64 * kBpChunks + lane] = nb;
        };
#pragma unroll
        for (int c = 0; c < D; ++c) sa[c] = load(c, 0), r[c] = band_first_lane(sa[c]);
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            double x[D];
#pragma unroll
            for (int c = 0; c < D; ++c) x[c] = load(c, fB + 64 * n);
            write_chunk.template operator()<true>(n, fB + 64 * n, x);
        }
#pragma unroll
        for (int c = 0; c < D; ++c) sb[c] = load(c, fB + 64 * 5);
        TA_LDS_ORDER();
        int S = 0;
        // one super-step; a == S % 8 (a constant once unrolled)
        auto body = [&]<bool TAIL>(int a) {
            TA_LDS_ORDER();  // (ring writes below, window reads above and further down: see k_band_bp_vacf)
            if (a % 4 == 0 && S != 0) {  // a new pass: a new reference row, which the ring's contents follow
                double delta[D];
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    const double rn = band_first_lane(sa[c]);
                    delta[c] = r[c] - rn, r[c] = rn;
                }
#pragma unroll
                for (int q = 0; q <= kBpChunks; ++q) {
                    const int k = ((q % kBpChunks) - a + kBpChunks) % kBpChunks;  // position q holds chunk S + k
                    if (k > 4) continue;                     // (k = 5 ... 7: chunks behind the window, about to be overwritten)
                    if (q == kBpChunks && k == 0) continue;  // the copy of position 0 serves windows that start in position 7: chunk S + k - 1
                    const int fbase = fB + 64 * (S + k);
                    double nb = 0.0;
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        double b = rB[c][64 * q + lane] + delta[c];
                        if (TAIL && !(fbase + lane < T)) b = 0.0;
                        rB[c][64 * q + lane] = b;
                        nb = __builtin_fma(b, b, nb);
                    }
                    rB[D][64 * q + lane] = nb;
                }
            }
            double A[D], asq = 0.0;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                A[c] = sa[c] - r[c];
                if (TAIL && !(64 * S + lane < T)) A[c] = 0.0;
                asq = __builtin_fma(A[c], A[c], asq);
            }
            write_chunk.template operator()<TAIL>((a + 5) % kBpChunks, fB + 64 * (S + 5), sb);  // chunk S + 5 (first read by super-step S + 1)
            TA_LDS_ORDER();
#pragma unroll
            for (int c = 0; c < D; ++c) {
                sb[c] = load(c, fB + 64 * (S + 6));
                sa[c] = load(c, 64 * (S + 1));
            }
            __builtin_amdgcn_sched_barrier(0);  // the requests stay here, a super-step ahead of their use
            if constexpr (!TAIL) {
                na += asq;
                // block lag by block lag: the windows of every column and of the norms, their MFMAs, the norm add (column by column
                // with the norm adds behind the last MFMA: 490 ms at 20000 x 25000 x 3; the norm adds after the first column: 476;
                // this order: 462)
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    const int w = (64 * a + 16 * d) % (64 * kBpChunks) + lane;
#pragma unroll
                    for (int c = 0; c < D; ++c) acc[d] = TA_BAND_MFMA(A[c], rB[c][w], acc[d]);
                    nbacc[d] += rB[D][w];
                }
            } else {
                const double Ah = -0.5 * asq;
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    const int j0 = 64 * S + 16 * (d0 + d);  // the window's first frame
                    if (j0 >= T) continue;                  // (wave-uniform) nothing of this block lag is left
                    const int w = (64 * a + 16 * d) % (64 * kBpChunks) + lane;
#pragma unroll
                    for (int c = 0; c < D; ++c) acc[d] = TA_BAND_MFMA(A[c], rB[c][w], acc[d]);
                    nbacc[d] += rB[D][w];
                    acc[d] = TA_BAND_MFMA(Ah, j0 + lane < T ? 1.0 : 0.0, acc[d]);  // -|a|^2 / 2 where the pair's later frame exists
                }
            }
        };
        // whole passes of 8 super-steps none of whose requests reaches frame T: S + 6 is the farthest chunk touched
        const int n_ok = (T - fB) / 64 - 6;
        const int S_bulk = (n_ok < n_super ? (n_ok > 0 ? n_ok : 0) : n_super) / kBpChunks * kBpChunks;
        while (S < S_bulk) {
#pragma unroll
            for (int a = 0; a < kBpChunks; ++a) {
                body.template operator()<false>(a);
                ++S;
            }
        }
        for (bool more = S < n_super; more;) {
#pragma unroll
            for (int a = 0; a < kBpChunks; ++a) {
                body.template operator()<true>(a);
                if (++S == n_super) {
                    more = false;
                    break;
                }
            }
        }
        TA_LDS_ORDER();  // (the next particle's rows overwrite the ring)
        }
        // (a - b)^2 summed = NA[m] + NB_d[n] - 2 acc_d[m][n]; diagonals as above
        double* blk = &rB[0][0];   // [16][32]
        double* dsum = blk + 512;  // [16][32]
        double* nas = dsum + 512;  // [16]
        double na_m[4];
        {
            const double tot = band_sum_rows(na);  // every lane: NA[lane & 15]
            if (lane < 16) nas[lane] = tot;
            TA_LDS_ORDER();
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) na_m[r4] = nas[4 * r4 + (lane >> 4)];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) blk[64 * q + lane] = 0.0;
        TA_LDS_ORDER();
        const int skew = (lane & 15) - (lane >> 4) + 15, half = lane >> 5, col = lane & 31;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const double nbd = band_sum_rows(nbacc[d]);  // NB_d[lane & 15]
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                blk[(4 * r4 + (lane >> 4)) * 32 + skew - 4 * r4] = __builtin_fma(-2.0, acc[d][r4], na_m[r4] + nbd);
            TA_LDS_ORDER();
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += blk[(8 * half + j) * 32 + col];
            s = band_sum_halves(s);
            if (lane < 32) dsum[d * 32 + lane] = s;
            TA_LDS_ORDER();
        }
        if constexpr (LAGS) {
            double* out = partial + ((long)g * n_pb + pb) * kBandPartial;
            for (int q = lane; q < kBandPartial; q += 64) {
                // slot q is lag offset q - 15 = 16 d + e: (d, e >= 0) and (d + 1, e - 16)
                const int off = q - 15;
                const int d = off >= 0 ? off >> 4 : -1, e = off - 16 * d;  // e in [0, 15] (off < 0: 1..15)
                double s = 0.0;
                if (off <= 255) {
                    if (d >= 0) s = dsum[d * 32 + e + 15];
                    if (e >= 1 && d + 1 <= 15) s += dsum[(d + 1) * 32 + e - 16 + 15];
                }
                out[q] = s;
            }
            TA_LDS_ORDER();
            continue;
        }
        const long atom = pb;
        double* out = bp_am + atom * ld_am;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int off = lane + 64 * k;
            const long lag = 256L * g + off;
            if (lag < T) {
                const int d = off >> 4, e = off & 15;
                double s = dsum[d * 32 + e + 15];
                if (d < 15 || e == 0) {
                    if (e >= 1) s += dsum[(d + 1) * 32 + e - 16 + 15];
                    out[lag] = lag == 0 ? 0.0 : factor * s / (double)(T - lag);
                } else {
                    unsafeAtomicAdd(out + lag, factor * s / (double)(T - lag));
                }
            }
        }
        if (g > 0 && lane >= 1 && lane < 16) {
            const long lag = 256L * g - 16 + lane;
            if (lag < T) unsafeAtomicAdd(out + lag, factor * dsum[lane - 16 + 15] / (double)(T - lag));
        }
        TA_LDS_ORDER();
    }
}

// lagsum[k] = factor * (sum over the particle blocks of the halves that hold lag k) / (T - k), in a fixed order: a block
// takes 16 lags, its 16 slices of threads each every 16th particle block, their sums are added slice by slice.
// zero_lag0: lagsum[0] = 0 exactly (viscosity.py:205-233 leaves row 0 at 0).  grid: ceil(T / 16) blocks of 256.
static __global__ void __launch_bounds__(256)
    k_bandbp_gather(const double* __restrict__ partial, long n_pb, int n_groups, int T, double factor, int zero_lag0,
                    double* __restrict__ lagsum) {
    __shared__ double red[16][17];
    const int kk = threadIdx.x & 15, sl = threadIdx.x >> 4, k = blockIdx.x * 16 + kk;
    double s = 0.0;
    if (k < T)
        for (int h = 0; h < 2; ++h) {  // group g holds lag offsets -15 ... 255 from 256 g
            const int g = (k >> 8) + h, off = k - 256 * g;
            if (g >= n_groups || off < -15) continue;
            const double* p = partial + (long)g * n_pb * kBandPartial + off + 15;
            for (long b = sl; b < n_pb; b += 16) s += p[b * kBandPartial];
        }
    red[sl][kk] = s;
    __syncthreads();
    if (sl == 0 && k < T) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][kk];
        lagsum[k] = (zero_lag0 && k == 0) ? 0.0 : factor * (t / (double)(T - k));
    }
}

}  // namespace ta
