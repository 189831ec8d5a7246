// band_common.hpp — what the matrix-core correlators of this library share: vector types, the MFMA spellings, the sums over a
// wave's four 16-lane rows by register swaps, the float32 kernels' diagonal flush into float64 (Band32Diag) and the float32
// product slab P = float32((m v) x) of the Einstein-Helfand float32 option (viscosity.py:205-215).  Used by
// bandbp_kernels.hpp / band32tp_kernels.hpp (the library's kernels: k-slots from the time axis) and by the column-packed
// forms that live under tools/band/ since round 6 (band_kernels.hpp, band32_kernels.hpp: the parity tests' second opinion,
// built on demand).
#pragma once
#include <hip/hip_runtime.h>

namespace ta {

typedef double band_d2 __attribute__((ext_vector_type(2)));
typedef double band_d4 __attribute__((ext_vector_type(4)));
typedef float band_f2 __attribute__((ext_vector_type(2)));
typedef float band_f4 __attribute__((ext_vector_type(4)));

constexpr int kBandPartial = 272;  // lags 16 d0 - 15 ... 16 d0 + 255, padded to a multiple of 16

#define TA_BAND_MFMA(A, B, C) __builtin_amdgcn_mfma_f64_16x16x4f64((A), (B), (C), 0, 0, 0)
#define TA_BAND32_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

typedef unsigned band_u2 __attribute__((ext_vector_type(2)));

// s0 + s1 + s2 + s3 of the four 16-lane rows, lane by lane, in every row: two register-to-register
// swaps (lane bit 5, lane bit 4) instead of trips through the LDS crossbar
__device__ __forceinline__ double band_sum_halves(double s) {  // lane l: s[l] + s[l ^ 32]
    const band_u2 w = __builtin_bit_cast(band_u2, s);
    const band_u2 x0 = __builtin_amdgcn_permlane32_swap(w.x, w.x, false, false);
    const band_u2 x1 = __builtin_amdgcn_permlane32_swap(w.y, w.y, false, false);
    return __builtin_bit_cast(double, band_u2{x0.x, x1.x}) + __builtin_bit_cast(double, band_u2{x0.y, x1.y});
}
__device__ __forceinline__ double band_sum_row_pairs(double u) {  // lane l: u[l] + u[l ^ 16]
    const band_u2 v = __builtin_bit_cast(band_u2, u);
    const band_u2 y0 = __builtin_amdgcn_permlane16_swap(v.x, v.x, false, false);
    const band_u2 y1 = __builtin_amdgcn_permlane16_swap(v.y, v.y, false, false);
    return __builtin_bit_cast(double, band_u2{y0.x, y1.x}) + __builtin_bit_cast(double, band_u2{y0.y, y1.y});
}
__device__ __forceinline__ double band_sum_rows(double s) { return band_sum_row_pairs(band_sum_halves(s)); }

// s0 + s1 + s2 + s3 of the four 16-lane rows, in every row: two register swaps (lane bit 5, lane bit 4).
// Inline assembly because hipcc 7.2 (clang 22) mis-selects the SECOND result of
// __builtin_amdgcn_permlane{32,16}_swap when it is used as a 32-bit scalar ("x.x + x.y" becomes
// v_add_f32 v1, v1, v1; tools/band/dma_probe.hip has the three-line reproducer) — the 64-bit use in
// band_kernels.hpp is compiled correctly.  "s_nop 1": the swap reads VGPRs a VALU instruction may have just
// written (the compiler pads its own swaps the same way; it does not look inside inline assembly).
__device__ __forceinline__ float band32_sum_rows(float s) {
    float a = s, b = s;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a += b;  // s[l] + s[l ^ 32]
    b = a;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;  // ... + the same of lane l ^ 16
}

// Where the float32 accumulators go every so many steps (kBand32Flush / kBand32tpFlush of the kernel).  Only the 31 diagonals of a 16 x 16 accumulator
// matter (element (m, n) of block lag d belongs to lag 16 d + n - m), so a flush sums the diagonals in float32
// — at most 16 values each — through a wave-private LDS image and adds the 272 lag values of the wave's 16
// block lags into float64: five doubles per lane instead of the 64 a lane would need to keep every element
// (128 registers: what held this kernel to one wave per SIMD).
//   D    [32][17] floats, element (m, n) at [n - m + 15][m]: a diagonal is a row; entries no element maps to
//        (and row 31) are zeroed once and never written;
//   ds   [16][32] floats, ds[d][e + 15] = diagonal e of accumulator d;
//   s[k] lag slot lane + 64 k of this piece: slot q holds lag offset q - 15 = 16 d + e from block lag d0, i.e.
//        diagonal e >= 0 of accumulator d plus diagonal e - 16 of accumulator d + 1.
struct Band32Diag {
    float* D;
    float* ds;
    double s[5];
    int lane;
    __device__ __forceinline__ void init(float* lds, int lane_) {
        D = lds, ds = lds + 32 * 17, lane = lane_;
        for (int q = lane; q < 32 * 17; q += 64) D[q] = 0.0f;
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] = 0.0;
    }
    // acc (float32, C layout row m = 4 (lane >> 4) + r, column n = lane & 15) -> s; acc is zeroed
    __device__ __forceinline__ void flush(band_f4 (&acc)[16]) {
        const int g = lane >> 4, n = lane & 15, t = lane & 31, hh = lane >> 5;
        float* wr = D + (n - 4 * g + 15) * 17 + 4 * g - 48;  // register r at wr[16 (3 - r)]
        const float* rd = D + t * 17 + 8 * hh;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
#pragma unroll
            for (int r = 0; r < 4; ++r) wr[16 * (3 - r)] = acc[d][r];
            acc[d] = band_f4{0.0f, 0.0f, 0.0f, 0.0f};
            __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave complete in order: this pins the compiler's order)
            float u = ((rd[0] + rd[1]) + (rd[2] + rd[3])) + ((rd[4] + rd[5]) + (rd[6] + rd[7]));
            float v = u;
            asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(v));
            if (lane < 32) ds[d * 32 + t] = u + v;
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int off = lane + 64 * k - 15;            // -15 ... 304
            const int d = (off + 16) / 16 - 1, e = off - 16 * d;  // e in [0, 15]
            float v = 0.0f;
            if (d >= 0 && d <= 15) v = ds[d * 32 + e + 15];
            if (e >= 1 && d + 1 >= 0 && d + 1 <= 15) v += ds[(d + 1) * 32 + e - 1];
            s[k] += (double)v;
        }
        __builtin_amdgcn_wave_barrier();
    }
};

// P = (m v) x per column, the reference's evaluation order (viscosity.py:212-219), formed in float64 and
// rounded once to float32; pair-major slabs in (float64 16-byte rows or float32 8-byte rows), pair-major
// float32 out.  A thread takes two consecutive rows of a pair: 16-byte stores.
template <typename SRC>
__global__ void __launch_bounds__(256)
    k_helfand_product32(const SRC* __restrict__ vel, const SRC* __restrict__ pos, const double* __restrict__ masses, long pitch,
                        long T, long n_cols, int D, float* __restrict__ P) {
    typedef SRC src4 __attribute__((ext_vector_type(4)));
    const long n_pairs = (n_cols + 1) / 2, half = pitch / 2;  // pitch is a multiple of 8
    for (long pair = blockIdx.y; pair < n_pairs; pair += gridDim.y) {
        const long c = 2 * pair;
        const double m0 = masses[c / D];
        const bool two = c + 1 < n_cols;
        const double m1 = two ? masses[(c + 1) / D] : 0.0;
        const src4* v = reinterpret_cast<const src4*>(vel) + pair * half;
        const src4* x = reinterpret_cast<const src4*>(pos) + pair * half;
        band_f4* p = reinterpret_cast<band_f4*>(P) + pair * half;
        for (long q = (long)blockIdx.x * 256 + threadIdx.x; 2 * q < T; q += (long)gridDim.x * 256) {
            const src4 vv = v[q], xx = x[q];
            band_f4 out;
            out.x = (float)((m0 * (double)vv.x) * (double)xx.x);
            out.y = two ? (float)((m1 * (double)vv.y) * (double)xx.y) : 0.0f;
            out.z = (float)((m0 * (double)vv.z) * (double)xx.z);
            out.w = two ? (float)((m1 * (double)vv.w) * (double)xx.w) : 0.0f;
            if (2 * q + 1 >= T) out.z = out.w = 0.0f;
            p[q] = out;
        }
    }
}

}  // namespace ta
