// layout.hip — the device-side slab layout ("pair-major") and the kernels that produce it.
//
// The reference stages velocities/positions as (n_frames, n_atoms, dim) row-major float64
// (/root/reference/transport_analysis/velocityautocorr.py:150-152,192-194, viscosity.py:128-134,
// 191-199): one atom's consecutive samples are n_atoms*dim*8 bytes apart, which is the worst
// possible layout for kernels that walk the TIME axis of a column.  The build owns the staging
// hooks, so frames are re-laid out on the device as they are committed:
//
//   column c = atom*dim + d of the shard;  pair p = c / 2;  n_pairs = ceil(n_cols / 2)
//   element (t, c)  ->  slab[(p * pitch + t) * 2 + (c & 1)],     pitch = n_frames rounded up to 8
//
// i.e. every pair of adjacent columns is ONE contiguous array of `pitch` rows of 16 bytes
// (x[t], y[t]); an odd last column is paired with zeros.  A wave reading consecutive rows of a
// pair moves 1 KiB per load instruction; a column pair is the FFT kernels' complex series as it
// stands.  The transposition runs on the committed chunk while the next chunk crosses PCIe.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

#include "ta_internal.hpp"

namespace ta {
namespace {

// src: frame-major rows [0, t_count) x columns [0, n_cols), row stride ld_row elements of SrcT.
// dst rows t_dst0 + [0, t_count) of every pair are written.  64 x 64 tiles through LDS: reads
// are 512 B (256 B for float32) per row segment, writes 1 KiB per pair (64 rows x 16 B).
template <typename SrcT, typename DstT>
__global__ void __launch_bounds__(256)
    k_relayout(const SrcT* __restrict__ src, long ld_row, long n_cols, long t_count,
               DstT* __restrict__ dst, long pitch, long t_dst0) {
    __shared__ __attribute__((aligned(16))) double tile[64][66];
    const int tid = threadIdx.x;
    const long c0 = (long)blockIdx.x * 64, r0 = (long)blockIdx.y * 64;
    const long n_pairs = (n_cols + 1) / 2;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = i * 4 + (tid >> 6), c = tid & 63;
        double v = 0.0;
        if (r0 + r < t_count && c0 + c < n_cols) v = (double)src[(r0 + r) * ld_row + c0 + c];
        tile[r][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = k * 256 + tid;
        const int r = idx & 63, pp = idx >> 6;
        const long pair = c0 / 2 + pp;
        if (pair < n_pairs && r0 + r < t_count) {
            const double2 v = *reinterpret_cast<const double2*>(&tile[r][2 * pp]);
            DstT* d = dst + (pair * pitch + t_dst0 + r0 + r) * 2;
            if constexpr (sizeof(DstT) == 8) *reinterpret_cast<double2*>(d) = v;
            else *reinterpret_cast<float2*>(d) = float2{(float)v.x, (float)v.y};
        }
    }
}

// float64 -> float64 with 16-byte accesses on BOTH sides (ld_row even, bases 16-byte aligned, so that
// the two columns of a pair are one aligned 16-byte element of the source row): a tile of 64 rows
// x 64 pairs of 16-byte elements through LDS (rows padded to 65 elements: the transposed
// ds_read_b128 is conflict-free), 1 KiB per wave instruction in (64 pairs of one row) and out
// (64 rows of one pair).  The frame-major *_dev inputs take this path: the 8-byte reads of the
// generic kernel ran at 4.5 TB/s (read + write).
constexpr int kWideLds = 64 * 65 * 16;
__global__ void __launch_bounds__(256)
    k_relayout_wide(const double* __restrict__ src, long ld_row, long n_cols, long t_count,
                    double* __restrict__ dst, long pitch, long t_dst0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wide_raw[];
    double2(*tile)[65] = reinterpret_cast<double2(*)[65]>(wide_raw);
    const int tid = threadIdx.x, q = tid >> 6, l = tid & 63;
    const long p0 = (long)blockIdx.x * 64, r0 = (long)blockIdx.y * 64;
    const long n_pairs = (n_cols + 1) / 2;
    double2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // row 4 i + q of the tile, pair l
        const long t = r0 + 4 * i + q, c = 2 * (p0 + l);
        v[i] = double2{0.0, 0.0};
        if (t < t_count) {
            const double* s = src + t * ld_row + c;
            if (c + 1 < n_cols) v[i] = *reinterpret_cast<const double2*>(s);
            else if (c < n_cols) v[i].x = s[0];  // an odd last column is paired with zeros
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[4 * i + q][l] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // pair 4 i + q of the tile, row l
        const long pair = p0 + 4 * i + q, t = r0 + l;
        if (pair < n_pairs && t < t_count)
            *reinterpret_cast<double2*>(dst + (pair * pitch + t_dst0 + t) * 2) = tile[l][4 * i + q];
    }
}

// float32 -> float32 (MDAnalysis' dtype on both sides: a float32 frame-major device tensor into a float32
// slab, "stage_device_f32") with 16-byte accesses on both sides: a source element is two pairs of one
// row, a destination element two rows of one pair.  Tile: 64 rows x 64 pairs of 8 bytes through LDS.
// Needs ld_row % 4 == 0, a 16-byte aligned source, even t_dst0 (the slab pitch is a multiple of 8).
__global__ void __launch_bounds__(256)
    k_relayout_wide32(const float* __restrict__ src, long ld_row, long n_cols, long t_count,
                      float* __restrict__ dst, long pitch, long t_dst0) {
    __shared__ __attribute__((aligned(16))) float2 tile[64][65];
    const int tid = threadIdx.x, q = tid >> 5, l = tid & 31;
    const long p0 = (long)blockIdx.x * 64, r0 = (long)blockIdx.y * 64;
    const long n_pairs = (n_cols + 1) / 2;
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // row 8 i + q of the tile, pairs 2 l and 2 l + 1
        const long t = r0 + 8 * i + q, c = 2 * (p0 + 2 * l);
        v[i] = float4{0.f, 0.f, 0.f, 0.f};
        if (t < t_count) {
            const float* s = src + t * ld_row + c;
            if (c + 3 < n_cols) v[i] = *reinterpret_cast<const float4*>(s);
            else {  // the row's last columns: an odd last column is paired with a zero
                if (c < n_cols) v[i].x = s[0];
                if (c + 1 < n_cols) v[i].y = s[1];
                if (c + 2 < n_cols) v[i].z = s[2];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        tile[8 * i + q][2 * l] = float2{v[i].x, v[i].y};
        tile[8 * i + q][2 * l + 1] = float2{v[i].z, v[i].w};
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // pair 8 i + q of the tile, rows 2 l and 2 l + 1
        const long pair = p0 + 8 * i + q, t = r0 + 2 * l;
        if (pair < n_pairs && t < t_count) {
            const float2 a = tile[2 * l][8 * i + q], b = tile[2 * l + 1][8 * i + q];
            float* d = dst + (pair * pitch + t_dst0 + t) * 2;
            if (t + 1 < t_count) *reinterpret_cast<float4*>(d) = float4{a.x, a.y, b.x, b.y};
            else *reinterpret_cast<float2*>(d) = a;
        }
    }
}

// The inverse, for callers that want a frame-major copy back (tests, diagnostics).
template <typename PmT>
__global__ void __launch_bounds__(256)
    k_unlayout(const PmT* __restrict__ pm, long pitch, long n_cols, long t_count,
               double* __restrict__ dst, long ld_row) {
    const long n_pairs = (n_cols + 1) / 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs * t_count;
         i += (long)gridDim.x * blockDim.x) {
        const long pair = i / t_count, t = i - pair * t_count;
        const PmT* q = pm + (pair * pitch + t) * 2;
        const double2 v = double2{(double)q[0], (double)q[1]};
        dst[t * ld_row + 2 * pair] = v.x;
        if (2 * pair + 1 < n_cols) dst[t * ld_row + 2 * pair + 1] = v.y;
    }
}

// By-particle results leave the correlators atom-major (src[atom * src_ld + lag], contiguous
// stores); the caller's array is (n_frames, ld_bp) like results.vacf_by_particle
// (velocityautocorr.py:145-147).  64 x 64 tiles through LDS, both sides coalesced; with
// `partial` the tile also adds its 64 atoms per lag: partial[tile_a][lag] (the mean over atoms,
// velocityautocorr.py:214, summed over tiles in a fixed order afterwards).
template <bool WIDE>
__global__ void __launch_bounds__(256)
    k_bp_transpose(const double* __restrict__ src, long src_ld, long n_atoms, long T,
                   double* __restrict__ bp, long ld_bp, double* __restrict__ partial) {
    __shared__ double tile[64][65];
    const int tid = threadIdx.x;
    const long a0 = (long)blockIdx.x * 64, t0 = (long)blockIdx.y * 64;
    if constexpr (WIDE) {
        // 16 bytes per lane on both sides (src_ld, ld_bp even, bases 16-byte aligned): a row of 64
        // doubles is read / written by 32 lanes
        const int r = tid >> 5, c2 = tid & 31;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int a = i * 8 + r, t = 2 * c2;
            double2 v = double2{0.0, 0.0};
            if (a0 + a < n_atoms) {
                const double* p = src + (a0 + a) * src_ld + t0 + t;
                if (t0 + t + 1 < T) v = *reinterpret_cast<const double2*>(p);
                else if (t0 + t < T) v.x = p[0];
            }
            tile[a][t] = v.x;
            tile[a][t + 1] = v.y;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = i * 8 + r, a = 2 * c2;
            if (t0 + t < T) {
                double* p = bp + (t0 + t) * ld_bp + a0 + a;
                if (a0 + a + 1 < n_atoms) *reinterpret_cast<double2*>(p) = double2{tile[a][t], tile[a + 1][t]};
                else if (a0 + a < n_atoms) p[0] = tile[a][t];
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int a = i * 4 + (tid >> 6), t = tid & 63;
            double v = 0.0;
            if (a0 + a < n_atoms && t0 + t < T) v = src[(a0 + a) * src_ld + t0 + t];
            tile[a][t] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = i * 4 + (tid >> 6), a = tid & 63;
            if (a0 + a < n_atoms && t0 + t < T) bp[(t0 + t) * ld_bp + a0 + a] = tile[a][t];
        }
    }
    if (partial && tid < 64 && t0 + tid < T) {
        double s = 0.0;
#pragma unroll 8
        for (int a = 0; a < 64; ++a) s += tile[a][tid];  // rows past n_atoms hold zeros
        partial[(long)blockIdx.x * T + t0 + tid] = s;
    }
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    unsigned long long z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Benchmark generator (include/ta_hip.h, ta_stage_synth): value of element idx is a function of
// (seed, idx) only, integer arithmetic plus ONE float64 multiply, so NumPy reproduces it bit for
// bit (oracle/synth.py): the sum of the eight 16-bit fields of two splitmix64 words, centred
// and scaled to unit variance (Irwin-Hall, n = 8).
__device__ __forceinline__ double synth_value(unsigned long long seed, unsigned long long idx) {
    const unsigned long long a = splitmix64(seed + 2 * idx), b = splitmix64(seed + 2 * idx + 1);
    long s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (long)((a >> (16 * k)) & 0xFFFF) + (long)((b >> (16 * k)) & 0xFFFF);
    return (double)(s - 262140) * 0x1.3988e1412ed76p-16;  // 1 / sqrt(8 (65536^2 - 1) / 12) = 1.8688123650118534e-05
}

// element (t, c) of the shard = synth(seed, t * n_cols_total + col_offset + c)
template <typename PmT>
__global__ void __launch_bounds__(256)
    k_synth(PmT* __restrict__ pm, long pitch, long n_cols, long T, unsigned long long seed,
            long col_offset, long n_cols_total) {
    const long n_pairs = (n_cols + 1) / 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs * T;
         i += (long)gridDim.x * blockDim.x) {
        const long pair = i / T, t = i - pair * T;
        const long c = 2 * pair;
        double2 v;
        v.x = synth_value(seed, (unsigned long long)(t * n_cols_total + col_offset + c));
        v.y = c + 1 < n_cols ? synth_value(seed, (unsigned long long)(t * n_cols_total + col_offset + c + 1)) : 0.0;
        PmT* q = pm + (pair * pitch + t) * 2;  // float32 slabs hold the value rounded once
        q[0] = (PmT)v.x, q[1] = (PmT)v.y;
    }
}

}  // namespace

hipError_t launch_relayout(const void* src, bool src_f32, long ld_row, long n_cols, long t_count,
                           void* dst, bool dst_f32, long pitch, long t_dst0, hipStream_t st) {
    if (t_count <= 0 || n_cols <= 0) return hipSuccess;
    if (!src_f32 && !dst_f32 && ld_row % 2 == 0 && ((uintptr_t)src & 15) == 0) {
        static std::atomic<bool> set[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !set[dev].load(std::memory_order_acquire)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_relayout_wide),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kWideLds);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) set[dev].store(true, std::memory_order_release);
        }
        const long n_pairs = (n_cols + 1) / 2;
        const dim3 wgrid((unsigned)((n_pairs + 63) / 64), (unsigned)((t_count + 63) / 64));
        hipLaunchKernelGGL(k_relayout_wide, wgrid, dim3(256), kWideLds, st, (const double*)src, ld_row, n_cols, t_count,
                           (double*)dst, pitch, t_dst0);
        return hipGetLastError();
    }
    if (src_f32 && dst_f32 && ld_row % 4 == 0 && ((uintptr_t)src & 15) == 0 && t_dst0 % 2 == 0 && pitch % 2 == 0) {
        const long n_pairs = (n_cols + 1) / 2;
        const dim3 wgrid((unsigned)((n_pairs + 63) / 64), (unsigned)((t_count + 63) / 64));
        hipLaunchKernelGGL(k_relayout_wide32, wgrid, dim3(256), 0, st, (const float*)src, ld_row, n_cols, t_count,
                           (float*)dst, pitch, t_dst0);
        return hipGetLastError();
    }
    const dim3 grid((unsigned)((n_cols + 63) / 64), (unsigned)((t_count + 63) / 64));
#define TA_GO(S, Dt) \
    hipLaunchKernelGGL((k_relayout<S, Dt>), grid, dim3(256), 0, st, (const S*)src, ld_row, n_cols, t_count, (Dt*)dst, pitch, t_dst0)
    if (src_f32 && dst_f32) TA_GO(float, float);
    else if (src_f32) TA_GO(float, double);
    else if (dst_f32) TA_GO(double, float);
    else TA_GO(double, double);
#undef TA_GO
    return hipGetLastError();
}

hipError_t launch_unlayout(const void* pm, bool pm_f32, long pitch, long n_cols, long t_count, double* dst,
                           long ld_row, hipStream_t st) {
    if (t_count <= 0 || n_cols <= 0) return hipSuccess;
    if (pm_f32)
        hipLaunchKernelGGL(k_unlayout<float>, dim3(2048), dim3(256), 0, st, (const float*)pm, pitch, n_cols, t_count,
                           dst, ld_row);
    else
        hipLaunchKernelGGL(k_unlayout<double>, dim3(2048), dim3(256), 0, st, (const double*)pm, pitch, n_cols,
                           t_count, dst, ld_row);
    return hipGetLastError();
}

hipError_t launch_bp_transpose(const double* src, long src_ld, long n_atoms, long T, double* bp, long ld_bp,
                               double* partial, hipStream_t st) {
    if (T <= 0 || n_atoms <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_atoms + 63) / 64), (unsigned)((T + 63) / 64));
    const bool wide = src_ld % 2 == 0 && ld_bp % 2 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)bp & 15) == 0;
    if (wide)
        hipLaunchKernelGGL(k_bp_transpose<true>, grid, dim3(256), 0, st, src, src_ld, n_atoms, T, bp, ld_bp, partial);
    else
        hipLaunchKernelGGL(k_bp_transpose<false>, grid, dim3(256), 0, st, src, src_ld, n_atoms, T, bp, ld_bp, partial);
    return hipGetLastError();
}

hipError_t launch_synth(void* pm, bool pm_f32, long pitch, long n_cols, long T, unsigned long long seed,
                        long col_offset, long n_cols_total, hipStream_t st) {
    if (T <= 0 || n_cols <= 0) return hipSuccess;
    if (pm_f32)
        hipLaunchKernelGGL(k_synth<float>, dim3(4096), dim3(256), 0, st, (float*)pm, pitch, n_cols, T, seed,
                           col_offset, n_cols_total);
    else
        hipLaunchKernelGGL(k_synth<double>, dim3(4096), dim3(256), 0, st, (double*)pm, pitch, n_cols, T, seed,
                           col_offset, n_cols_total);
    return hipGetLastError();
}

}  // namespace ta
