// mid_kernels.hpp — the O(T^2) correlators between the short kernels and the matrix-core ones (65 ... 512 frames).
//
// The quantities of direct_kernels.hpp (velocityautocorr.py:217-238; viscosity.py:201-233, difference first).  There, a
// thread tiles L x L (lags x time steps, L = 8) and reads 2 L values from the LDS per tile: 2 bytes per FMA, the LDS'
// whole bandwidth.  Here a lane owns a BLOCK of 16 consecutive lags of one column and walks the time axis with a sliding
// window in registers: at step i it needs x[i] and ONE new window value x[i + k0 + 16], for 16 FMAs -- 1 byte per FMA,
// read 16 steps at a time (a group: 16 x 16 terms, fully unrolled, so that the window's rotation is a renaming of
// registers).  Lag block b has T - 16 b steps: a lane takes blocks b and NB - 1 - b, so every lane walks ~T + 16 steps.
//
// A workgroup stages a tile of NC adjacent columns (a multiple of dim: whole particles) from the pair-major slab into the
// LDS (xs[column][TS], zero padded to the next multiple of 16 plus 16: out-of-range products of the windowed VACF vanish;
// Helfand forms P = (m v) x on the way in and masks its out-of-range terms in a block's last groups), lanes = (column,
// pair of blocks); then the accumulators go back through the LDS: by_particle[lag, atom] (the sum over the particle's dim
// columns) is stored straight into the caller's (n_frames, ld) array, and thread k adds up lag k over the tile's columns in
// a register that lives across the tiles: partial[workgroup][lag], added in a fixed order by k_sum_partials.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "direct_kernels.hpp"

namespace ta {

constexpr int kMidLB = 16;       // lags per block
constexpr int kMidThreads = 256;

// one group: steps s < LB of lags u < LB; the window x[i0 + k0 + (s + u)] is wa for s + u < LB and wb beyond (the next
// group swaps their roles: no register moves).  MASKED (Helfand): term (s, u) counts while s + u < nvalid = T - (i0 + k0)
template <int MODE, bool MASKED>
__device__ __forceinline__ void mid_group(const double (&xi)[kMidLB], const double (&wa)[kMidLB], const double (&wb)[kMidLB],
                                          double (&acc)[kMidLB], int nvalid) {
    constexpr int LB = kMidLB;
#pragma unroll
    for (int s = 0; s < LB; ++s) {
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const double wv = s + u < LB ? wa[s + u] : wb[s + u - LB];
            if (MODE == MODE_HELFAND) {
                double df = xi[s] - wv;
                if (MASKED) df = (s + u < nvalid) ? df : 0.0;
                acc[u] = __builtin_fma(df, df, acc[u]);
            } else {
                acc[u] = __builtin_fma(xi[s], wv, acc[u]);
            }
        }
    }
}

// groups of a block whose every term counts: i0 + k0 + 2 LB - 2 < T
__device__ __forceinline__ int mid_full_groups(int T, int k0) {
    const int r = T - k0 - 2 * kMidLB + 1;
    return r < 0 ? 0 : r / kMidLB + 1;
}

// n_full: Helfand, groups every lane of the WAVE runs unmasked (the minimum over its lanes of mid_full_groups): two loops
// with one form of the group each (both forms side by side in one loop cost the compiler 1 KiB of scratch per lane)
template <int MODE>
__device__ __forceinline__ void mid_block(const double* __restrict__ xc, int T, int k0, int n_full, double (&acc)[kMidLB]) {
    constexpr int LB = kMidLB;
    double w0[LB], w1[LB], xi[LB];
    auto load16 = [&](const double* src, double(&dst)[LB]) __attribute__((always_inline)) {
        const double2* p2 = reinterpret_cast<const double2*>(src);
#pragma unroll
        for (int q = 0; q < LB / 2; ++q) {
            const double2 r = p2[q];
            dst[2 * q] = r.x;
            dst[2 * q + 1] = r.y;
        }
    };
    load16(xc + k0, w0);
    // two groups per trip: (w0, w1) then (w1, w0).  Helfand: pairs of groups every lane of the wave runs unmasked first, then
    // the masked form (both forms side by side in one loop cost the compiler scratch)
    int i0 = 0;
    auto pair = [&](auto masked) __attribute__((always_inline)) {
        constexpr bool M = decltype(masked)::value;
        load16(xc + i0, xi);
        load16(xc + i0 + k0 + LB, w1);
        mid_group<MODE, M>(xi, w0, w1, acc, T - (i0 + k0));
        if (i0 + LB < T - k0) {
            load16(xc + i0 + LB, xi);
            load16(xc + i0 + k0 + 2 * LB, w0);
            mid_group<MODE, M>(xi, w1, w0, acc, T - (i0 + LB + k0));
        }
    };
    if (MODE == MODE_HELFAND) {
        for (; i0 + 2 * LB <= n_full * LB; i0 += 2 * LB) pair(std::false_type{});
        for (; i0 < T - k0; i0 += 2 * LB) pair(std::true_type{});
    } else {
        for (; i0 < T - k0; i0 += 2 * LB) pair(std::false_type{});
    }
}

// NC: columns per tile (a multiple of D, <= NCL = 1 << ncl_log2 lanes per block pair); TS: LDS stride of a column in doubles
// (>= roundup(T, 16) + 16, TS % 4 == 2); dynamic LDS: (NCL * TS + T) doubles; partial: [gridDim.x][T]
template <int MODE>
__global__ void __launch_bounds__(kMidThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
    k_mid(const double* __restrict__ vel, const double* __restrict__ pos, const double* __restrict__ masses, long pitch, int T,
          long n_atoms, int D, int NC, int ncl_log2, int TS, double factor, double* __restrict__ bp, long ld_bp,
          double* __restrict__ partial) {
    constexpr int LB = kMidLB;
    extern __shared__ __attribute__((aligned(16))) double mid_lds[];
    const int NCL = 1 << ncl_log2;
    double* xs = mid_lds;
    double* rn = mid_lds + (long)NCL * TS;  // factor / (T - k)
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, n_waves = nt >> 6;
    for (int k = tid; k < T; k += nt) rn[k] = factor / (double)(T - k);
    const long n_cols = n_atoms * D;
    const long n_tiles = (n_cols + NC - 1) / NC;
    const int NB = (T + LB - 1) / LB, NP = (NB + 1) / 2;
    const int j = tid & (NCL - 1), bpi = tid >> ncl_log2;
    const int b1 = bpi, b2 = NB - 1 - bpi;
    double tot0 = 0.0, tot1 = 0.0;  // lags tid and tid + nt, over all tiles of this workgroup
    // zero padding, once: staging writes rows < T, and the lags >= T that the results put there are sums of nothing
    for (int idx = tid; idx < NCL * (TS - T); idx += nt) {
        const int jj = idx / (TS - T), t = T + idx - jj * (TS - T);
        xs[jj * TS + t] = 0.0;
    }

    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long c0 = tile * NC;
        const int ncv = (int)min((long)NC, n_cols - c0);  // columns of this tile
        __syncthreads();                                  // the previous tile's results are read
        // ---- stage: pairs p_lo .. p_hi, a wave per pair, lanes along the time axis (16-byte rows)
        const long p_lo = c0 >> 1, p_hi = (c0 + ncv - 1) >> 1;
        // (items = (pair, 64-row chunk) of this wave, a batch at a time: the loads of a batch are in flight together)
        const int tchunks = (T + 63) >> 6;
        const int my_pairs = (int)((p_hi - p_lo - wave + n_waves) / n_waves);  // pairs p_lo + wave + n_waves i <= p_hi
        const int n_items = wave <= p_hi - p_lo ? my_pairs * tchunks : 0;
        constexpr int SB = 8;  // items per batch
        for (int i0 = 0; i0 < n_items; i0 += SB) {
            double2 r[SB], q[SB];
            int je8[SB], t8[SB];
            double me8[SB], mo8[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int it = i0 + u;
                const int ip = it / tchunks, tc = it - ip * tchunks;
                const long p = p_lo + wave + (long)n_waves * ip;
                const int t = lane + 64 * tc;
                je8[u] = (int)(2 * p - c0);
                t8[u] = it < n_items && t < T ? t : -1;
                if (t8[u] >= 0) {
                    r[u] = (reinterpret_cast<const double2*>(vel) + p * pitch)[t];
                    if (MODE == MODE_HELFAND) {
                        q[u] = (reinterpret_cast<const double2*>(pos) + p * pitch)[t];
                        const int je = je8[u];
                        me8[u] = je >= 0 && je < ncv ? masses[(2 * p) / D] : 0.0;
                        mo8[u] = je + 1 >= 0 && je + 1 < ncv ? masses[(2 * p + 1) / D] : 0.0;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                if (t8[u] < 0) continue;
                const int je = je8[u], jo = je + 1;
                double vx = r[u].x, vy = r[u].y;
                if (MODE == MODE_HELFAND) {
                    vx = (me8[u] * vx) * q[u].x;
                    vy = (mo8[u] * vy) * q[u].y;
                }
                if (je >= 0 && je < ncv) xs[je * TS + t8[u]] = vx;
                if (jo >= 0 && jo < ncv) xs[jo * TS + t8[u]] = vy;
            }
        }
        __syncthreads();
        // ---- compute: lane = (column j, blocks b1 and b2 = NB - 1 - b1)
        double acc1[LB], acc2[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) acc1[u] = acc2[u] = 0.0;
        const bool active = j < ncv && bpi < NP;
        int nf1 = 0, nf2 = 0;
        if (MODE == MODE_HELFAND) {  // (outside the divergent part: the lane exchanges want every lane)
            nf1 = active ? mid_full_groups(T, b1 * LB) : 1 << 30;
            nf2 = active && b2 > b1 ? mid_full_groups(T, b2 * LB) : 1 << 30;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                nf1 = min(nf1, __shfl_xor(nf1, off, 64));
                nf2 = min(nf2, __shfl_xor(nf2, off, 64));
            }
        }
        if (active) {
            mid_block<MODE>(xs + j * TS, T, b1 * LB, nf1, acc1);
            if (b2 > b1) mid_block<MODE>(xs + j * TS, T, b2 * LB, nf2, acc2);
        }
        __syncthreads();  // every window is read: the columns become their lags
        if (active) {
            double* res = xs + j * TS;
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                res[b1 * LB + u] = acc1[u];
                if (b2 > b1) res[b2 * LB + u] = acc2[u];
            }
        }
        __syncthreads();
        // ---- results: by particle (a particle's D columns added), and lag k over the tile's columns
        if (bp) {
            const int na = ncv / D;  // (NC and n_cols are multiples of D)
            const long atom0 = c0 / D;
            const unsigned long long inv = 0x100000000ull / (unsigned)na + 1;  // idx / na for idx < 2^16 (na <= 64, T <= 512)
            for (int idx = tid; idx < na * T; idx += nt) {
                const int k = (int)(((unsigned long long)idx * inv) >> 32), a = idx - k * na;
                double v = 0.0;
                for (int d = 0; d < D; ++d) v += xs[(a * D + d) * TS + k];
                if (MODE == MODE_HELFAND && k == 0) v = 0.0;
                bp[(long)k * ld_bp + atom0 + a] = v * rn[k];
            }
        }
        {
            // (eight columns at a time: independent reads, two chains of adds per lag)
            const int k0 = tid < T ? tid : 0, k1 = tid + nt < T ? tid + nt : 0;
            double s0 = 0.0, s1 = 0.0, r0 = 0.0, r1 = 0.0;
            int c = 0;
            for (; c + 8 <= ncv; c += 8) {
                double a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a[u] = xs[(c + u) * TS + k0];
                    b[u] = xs[(c + u) * TS + k1];
                }
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    s0 += a[u];
                    r0 += a[u + 1];
                    s1 += b[u];
                    r1 += b[u + 1];
                }
            }
            for (; c < ncv; ++c) {
                s0 += xs[c * TS + k0];
                s1 += xs[c * TS + k1];
            }
            tot0 += s0 + r0;  // (a thread without a lag adds up lag 0 and never stores it)
            tot1 += s1 + r1;
        }
    }
    double* prow = partial + (long)blockIdx.x * T;
    if (tid < T) prow[tid] = (MODE == MODE_HELFAND && tid == 0) ? 0.0 : tot0 * rn[tid];
    if (tid + nt < T) prow[tid + nt] = tot1 * rn[tid + nt];
}

}  // namespace ta
