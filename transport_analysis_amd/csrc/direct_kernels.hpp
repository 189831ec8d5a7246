// direct_kernels.hpp — all-lag direct correlators (K4, K5+K6 of SURVEY.md 8a).
//
// MODE_VACF    : VelocityAutocorr._conclude_simple
//                (/root/reference/transport_analysis/velocityautocorr.py:217-238)
//                B[k,n] = 1/(T-k) sum_i sum_d v[i,n,d] v[i+k,n,d],  k = 0..T-1
// MODE_HELFAND : ViscosityHelfand._conclude
//                (/root/reference/transport_analysis/viscosity.py:201-233)
//                P = (m*v)*x ; H[k,n] = scale/(D (T-k)) sum_i sum_d (P[i]-P[i+k])^2,
//                k = 1..T-1, H[0,n] = 0.  The difference is formed first, exactly as
//                the reference does (no prefix-sum expansion: SURVEY.md 7.3-5).
//
// One workgroup takes one atom at a time.  For each of its D columns the whole
// time series (T float64, P formed on the fly for Helfand) is staged in LDS; a
// thread owns a PAIR of lag chunks (L consecutive lags each): chunk j and chunk
// nchunks-1-j, so every thread does the same number of multiply-adds although
// lag k has T-k terms.  Inside a chunk the thread register-tiles L lags x L time
// steps: per step it reads L new series values for the lagged window and L values
// for the leading window from LDS and issues L*L FMAs (f64 VALU; this path is
// FP64-FMA bound, not HBM bound: algorithmic intensity ~ T/8 flop per byte).
//
// LDS layout: groups of L values followed by 16 B of padding, so lane-strided
// 16-byte reads (lane stride = one group) hit 16 distinct bank slots.
#pragma once
#include <hip/hip_runtime.h>

namespace ta {

enum { MODE_VACF = 0, MODE_HELFAND = 1 };

// Staged column layout: groups of L values padded to a stride of 4 (mod 8) dwords: groups
// stay 16-byte aligned for ds_read_b128 and the 16 lanes a b128 read services together
// (consecutive chunks) fall on 16 different 4-dword bank sets.
constexpr int group_stride_dwords(int L, int elem_dwords) {
    int s = L * elem_dwords;
    while (s % 8 != 4) ++s;
    return s;
}
template <int L, typename Real>
constexpr int group_stride() {  // in elements
    return group_stride_dwords(L, (int)sizeof(Real) / 4) / ((int)sizeof(Real) / 4);
}

template <int L, typename Real>
__device__ __forceinline__ int lds_slot(int e) {  // element index -> slot
    return (e / L) * group_stride<L, Real>() + (e % L);
}

template <int L>
__device__ __forceinline__ void lds_read_group(const double* __restrict__ s, int e,
                                               double (&out)[L]) {
    // e is a multiple of L: one padded group, 16-byte aligned
    const double2* p = reinterpret_cast<const double2*>(s + (e / L) * group_stride<L, double>());
#pragma unroll
    for (int i = 0; i < L / 2; ++i) {
        double2 t = p[i];
        out[2 * i] = t.x;
        out[2 * i + 1] = t.y;
    }
}

template <int L>
__device__ __forceinline__ void lds_read_group(const float* __restrict__ s, int e,
                                               float (&out)[L]) {
    const float* g = s + (e / L) * group_stride<L, float>();
    const float4* p = reinterpret_cast<const float4*>(g);
#pragma unroll
    for (int i = 0; i < L / 4; ++i) {
        float4 t = p[i];
        out[4 * i] = t.x;
        out[4 * i + 1] = t.y;
        out[4 * i + 2] = t.z;
        out[4 * i + 3] = t.w;
    }
    if constexpr (L % 4 == 2) {
        const float2 t = *reinterpret_cast<const float2*>(g + L - 2);
        out[L - 2] = t.x;
        out[L - 1] = t.y;
    }
}

// Accumulate one chunk (lags k0..k0+L-1) of one staged column into acc[L].
// The staged copy is zero-padded beyond T.
template <int MODE, int L>
__device__ __forceinline__ void chunk_accumulate(const double* __restrict__ s, int T, int k0,
                                                 double (&acc)[L]) {
    // per block: x[0..L) = values at i..i+L-1 and the window lo ++ hi = values at
    // i+k0 .. i+k0+2L-1.  Blocks go two at a time so that the window halves swap roles
    // instead of being copied.
    const int span = T - k0;  // number of leading indices i with at least lag k0 valid
    const int nblk = (span + L - 1) / L;
    // blocks whose L x L tile is valid for every lag of the chunk: i + 2L - 2 + k0 < T
    int nfull = (T - k0 - 2 * L + 1) / L + 1;
    if (T - k0 - 2 * L + 1 < 0) nfull = 0;
    if (nfull > nblk) nfull = nblk;
    if (MODE == MODE_VACF) nfull = nblk;  // zero padding makes out-of-range products vanish
#define TA_TILE_F64(LO, HI)                                                          \
    _Pragma("unroll") for (int b = 0; b < L; ++b) _Pragma("unroll") for (int a = 0; a < L; ++a) { \
        const double wv = (a + b < L) ? LO[(a + b) % L] : HI[(a + b) % L];           \
        if (MODE == MODE_VACF) {                                                     \
            acc[a] = fma(x[b], wv, acc[a]);                                          \
        } else {                                                                     \
            const double dd = x[b] - wv;                                             \
            acc[a] = fma(dd, dd, acc[a]);                                            \
        }                                                                            \
    }
    double wa[L], wb[L], x[L];
    lds_read_group<L>(s, k0, wa);
    int blk = 0;
    for (; blk + 1 < nfull; blk += 2) {
        const int i = blk * L;
        lds_read_group<L>(s, i, x);
        lds_read_group<L>(s, i + k0 + L, wb);
        TA_TILE_F64(wa, wb)
        lds_read_group<L>(s, i + L, x);
        lds_read_group<L>(s, i + k0 + 2 * L, wa);
        TA_TILE_F64(wb, wa)
    }
    if (blk < nfull) {
        lds_read_group<L>(s, blk * L, x);
        lds_read_group<L>(s, blk * L + k0 + L, wb);
        TA_TILE_F64(wa, wb)
        ++blk;
    }
#undef TA_TILE_F64
    if (MODE == MODE_HELFAND) {
        // ragged end of the chunk (at most two blocks): a pair (i+b, i+b+k) only counts
        // while i+b+k < T.  Straight from the staged column, rolled over b: small code.
        for (; blk < nblk; ++blk) {
            const int i = blk * L;
#pragma unroll 1
            for (int b = 0; b < L; ++b) {
                const double xb = s[lds_slot<L, double>(i + b)];
#pragma unroll
                for (int a = 0; a < L; ++a) {
                    const int e = i + b + k0 + a;
                    const double d = xb - s[lds_slot<L, double>(e)];
                    if (e < T) acc[a] = fma(d, d, acc[a]);
                }
            }
        }
    }
}

// ---- float32 path: packed math (v_pk_add_f32 / v_pk_fma_f32, two float32 per lane) ------
typedef float v2f __attribute__((ext_vector_type(2)));

template <int L>
__device__ __forceinline__ void lds_read_pairs(const float* __restrict__ s, int e,
                                               v2f (&out)[L / 2]) {
    // e is a multiple of L: one padded group of L floats as L/2 register pairs
    float t[L];
    lds_read_group<L>(s, e, t);
#pragma unroll
    for (int k = 0; k < L / 2; ++k) out[k] = v2f{t[2 * k], t[2 * k + 1]};
}

// One L x L tile on register pairs.  X[bp] = (x[2bp], x[2bp+1]); the window w[0..2L) is
// lo ++ hi as natural pairs E[k] = (w[2k], w[2k+1]); odd lags need O[k] = (w[2k+1], w[2k+2]).
// part[a] holds two half-sums of lag a (even and odd b), added together at the flush.
template <int MODE, int L>
__device__ __forceinline__ void tile_f32(const v2f (&X)[L / 2], const v2f (&lo)[L / 2],
                                         const v2f (&hi)[L / 2], v2f (&part)[L]) {
    v2f E[L], O[L - 1];
#pragma unroll
    for (int k = 0; k < L / 2; ++k) {
        E[k] = lo[k];
        E[L / 2 + k] = hi[k];
    }
#pragma unroll
    for (int k = 0; k < L - 1; ++k) O[k] = __builtin_shufflevector(E[k], E[k + 1], 1, 2);
#pragma unroll
    for (int bp = 0; bp < L / 2; ++bp)
#pragma unroll
        for (int a = 0; a < L; ++a) {
            const int idx = a + 2 * bp;
            const v2f W = (a & 1) ? O[(idx - 1) / 2] : E[idx / 2];
            if (MODE == MODE_VACF) {
                part[a] = __builtin_elementwise_fma(X[bp], W, part[a]);
            } else {
                const v2f d = X[bp] - W;
                part[a] = __builtin_elementwise_fma(d, d, part[a]);
            }
        }
}

// float32 chunk: products / squared differences and 8L-term sums per lag in float32 (two
// packed half-sums), added into the float64 accumulators every 8 blocks.  Blocks go two at
// a time so the window halves swap roles instead of being copied.
template <int MODE, int L>
__device__ __forceinline__ void chunk_accumulate(const float* __restrict__ s, int T, int k0,
                                                 double (&acc)[L]) {
    static_assert(L % 2 == 0, "packed float32 tile needs an even chunk");
    const int span = T - k0;
    const int nblk = (span + L - 1) / L;
    int nfull = (T - k0 - 2 * L + 1) / L + 1;
    if (T - k0 - 2 * L + 1 < 0) nfull = 0;
    if (nfull > nblk) nfull = nblk;
    if (MODE == MODE_VACF) nfull = nblk;
    v2f part[L];
#pragma unroll
    for (int a = 0; a < L; ++a) part[a] = v2f{0.f, 0.f};
    auto flush = [&]() {
#pragma unroll
        for (int a = 0; a < L; ++a) {
            acc[a] += (double)(part[a].x + part[a].y);
            part[a] = v2f{0.f, 0.f};
        }
    };
    v2f A[L / 2], B[L / 2], X[L / 2];
    lds_read_pairs<L>(s, k0, A);
    int blk = 0;
    for (; blk + 1 < nfull; blk += 2) {
        const int i = blk * L;
        lds_read_pairs<L>(s, i, X);
        lds_read_pairs<L>(s, i + k0 + L, B);
        tile_f32<MODE, L>(X, A, B, part);
        lds_read_pairs<L>(s, i + L, X);
        lds_read_pairs<L>(s, i + k0 + 2 * L, A);
        tile_f32<MODE, L>(X, B, A, part);
        if (((blk + 2) & 7) == 0) flush();
    }
    if (blk < nfull) {
        lds_read_pairs<L>(s, blk * L, X);
        lds_read_pairs<L>(s, blk * L + k0 + L, B);
        tile_f32<MODE, L>(X, A, B, part);
        ++blk;
    }
    flush();
    if (MODE == MODE_HELFAND) {
        for (; blk < nblk; ++blk) {  // ragged end, as in the float64 version
            const int i = blk * L;
#pragma unroll 1
            for (int b = 0; b < L; ++b) {
                const float xb = s[lds_slot<L, float>(i + b)];
#pragma unroll
                for (int a = 0; a < L; ++a) {
                    const int e = i + b + k0 + a;
                    const float d = xb - s[lds_slot<L, float>(e)];
                    if (e < T) acc[a] = fma((double)d, (double)d, acc[a]);
                }
            }
        }
    }
}

// grid.x: persistent workgroups.  A workgroup is n_groups = blockDim.x / gnt column groups of
// gnt threads; group g of workgroup b works on atoms (b*n_groups + g) + k*gridDim.x*n_groups,
// each group with its own staged column, so that a compute unit's 16 wave slots are filled by
// ONE workgroup whose waves the hardware spreads evenly over the four SIMDs.
// ts_partial: [gridDim.x * n_groups][T] (zeroed by caller).
// GLOBAL_STAGE: trajectories too long for LDS (one column = (T/L+3) padded groups > 160 KiB)
// stage the column in the group's slice of `stage_buf` instead ([gridDim.x * n_groups][n_slots],
// L1/L2 resident): same code, lower rate, no limit on n_frames.
// SrcT: element type of the pair-major slabs (float64, or float32 device slabs: "stage_device_f32")
template <int MODE, int L, bool GLOBAL_STAGE, typename Real, typename SrcT = double>
__global__ void __launch_bounds__(1024)
    k_direct(const SrcT* __restrict__ vel, const SrcT* __restrict__ pos,
             const double* __restrict__ masses, long pitch, int T, long n_atoms, int D,
             double scale, double* __restrict__ by_particle, long ld_bp,
             double* __restrict__ ts_partial, void* __restrict__ stage_buf, int gnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n_groups = blockDim.x / gnt;
    const int grp = threadIdx.x / gnt, tid = threadIdx.x - grp * gnt, nt = gnt;
    const int nchunks = (T + L - 1) / L;
    const int npairs = (nchunks + 1) / 2;
    const long col_slots = (long)(nchunks + 3) * group_stride<L, Real>();
    Real* s = reinterpret_cast<Real*>(smem_raw) + grp * col_slots;
    if constexpr (GLOBAL_STAGE)
        s = reinterpret_cast<Real*>(stage_buf) + ((long)blockIdx.x * n_groups + grp) * col_slots;
    // staged length: up to the end of the window any chunk can touch, zero padded
    const int n_stage = (nchunks + 3) * L;
    double* ts_out = ts_partial + ((long)blockIdx.x * n_groups + grp) * T;

    for (long atom0 = (long)blockIdx.x * n_groups; atom0 < n_atoms;
         atom0 += (long)gridDim.x * n_groups) {
        const long atom = atom0 + grp;
        const bool valid = atom < n_atoms;  // idle groups still take part in the barriers
        const double mass = (MODE == MODE_HELFAND && valid) ? masses[atom] : 1.0;
        for (int pp0 = 0; pp0 < npairs; pp0 += nt) {
            const int pp = pp0 + tid;
            const bool active = valid && pp < npairs;
            const int j1 = pp, j2 = nchunks - 1 - pp;
            double acc1[L], acc2[L];
#pragma unroll
            for (int a = 0; a < L; ++a) acc1[a] = acc2[a] = 0.0;
            for (int d = 0; d < D; ++d) {
                __syncthreads();  // previous column fully consumed
                // pair-major slab (layout.hip): column c, row e at ((c/2)*pitch + e)*2 + (c&1):
                // consecutive lanes read consecutive rows, 16 bytes apart
                const long c = atom * D + d;
                const long cbase = (c >> 1) * pitch * 2 + (c & 1);
                for (int e = tid; e < n_stage; e += nt) {
                    double val = 0.0;
                    if (valid && e < T) {
                        const long g = cbase + 2L * e;
                        val = (double)vel[g];
                        if (MODE == MODE_HELFAND) val = (mass * val) * (double)pos[g];
                    }
                    s[lds_slot<L, Real>(e)] = (Real)val;
                }
                __syncthreads();
                if (active) {
                    chunk_accumulate<MODE, L>(s, T, j1 * L, acc1);
                    if (j2 != j1) chunk_accumulate<MODE, L>(s, T, j2 * L, acc2);
                }
            }
            if (active) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int jc = h ? j2 : j1;
                    if (h && j2 == j1) break;
#pragma unroll
                    for (int a = 0; a < L; ++a) {
                        const int k = jc * L + a;
                        if (k < T) {
                            double val = (h ? acc2[a] : acc1[a]) / (double)(T - k);
                            if (MODE == MODE_HELFAND) val = (k == 0) ? 0.0 : (val / (double)D) * scale;
                            // atom-major scratch (ld_bp = row pitch >= T), transposed afterwards
                            if (by_particle) by_particle[(long)atom * ld_bp + k] = val;
                            ts_out[k] += val;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace ta
