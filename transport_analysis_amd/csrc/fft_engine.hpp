// fft_engine.hpp — workgroup-level complex-f64 FFT on LDS for gfx950.
//
// One workgroup owns one length-M complex series in LDS (16 B per element,
// M*16 <= 160 KiB).  The forward transform is an in-place decimation-in-
// frequency (DIF) pass structure: natural-order input, digit-reversed output.
// The inverse is the exact transpose (DIT): it consumes the digit-reversed
// order and produces natural order, so spectra accumulated in the forward
// transform's own output order never need a permutation.
//
// Stage s works on sub-blocks of size N_s = M / (R_0..R_{s-1}); with L = N_s/R_s
// butterfly u = blk*L + b reads i_j = blk*N_s + b + j*L (j < R_s), computes the
// R_s-point DFT, multiplies output q by W_{N_s}^{q*b} and writes back to the same
// R_s slots.  Consecutive lanes touch consecutive elements whenever L >= 64; the
// small-L stages stride by R_s elements, which the XOR swizzle below keeps
// conflict-free for ds_read_b128/ds_write_b128 (16 lanes per LDS cycle group,
// each group covering all residues of lane%16).
#pragma once
#include <hip/hip_runtime.h>

namespace ta {

struct cd {
    double x, y;
};

__device__ __forceinline__ cd operator+(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd a, cd b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ cd cmulc(cd a, cd b) {  // a * conj(b)
    return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}
__device__ __forceinline__ cd mul_mi(cd a) { return {a.y, -a.x}; }  // a * (-i)
__device__ __forceinline__ cd mul_pi(cd a) { return {-a.y, a.x}; }  // a * (+i)
__device__ __forceinline__ double norm2(cd a) { return a.x * a.x + a.y * a.y; }

// LDS element swizzle: keeps strided (power-of-two) element access conflict-free
// for 16-byte accesses; a permutation inside every aligned 16-element block.
__device__ __forceinline__ int sw(int i) { return i ^ ((i >> 4) & 15); }
// sw(base + c) from sb = sw(base), for a constant c whose one-bits are zero in base (every
// butterfly: base = blk*N + b with b < L, c = j*L, N = R*L).  sw is linear over disjoint bit
// fields: sw(base + c) = sw(base) ^ sw(c); the bits of sw(c) above the low nibble are c's own
// and cannot meet a one-bit of sb, so they are an ADD and fold into the DS instruction's
// immediate offset.  A butterfly's R addresses then cost one v_xor per distinct low nibble of
// sw(j*L) instead of a shift, a bit-op and an add each.
__device__ __forceinline__ int sw_off(int sb, int c) {
    const int s = c ^ ((c >> 4) & 15);
    return (sb ^ (s & 15)) + (s & ~15);
}

constexpr double kR2 = 0.70710678118654752440084436210485;   // sqrt(1/2)
constexpr double kC8 = 0.92387953251128675612818318939679;   // cos(pi/8)
constexpr double kS8 = 0.38268343236508977172845998403040;   // sin(pi/8)
constexpr double kC5a = 0.30901699437494742410229341718282;  // cos(2pi/5)
constexpr double kC5b = -0.80901699437494742410229341718282; // cos(4pi/5)
constexpr double kS5a = 0.95105651629515357211643933337938;  // sin(2pi/5)
constexpr double kS5b = 0.58778525229247312916870595463907;  // sin(4pi/5)

// ---- forward DFTs on registers: v[q] <- sum_j v[j] exp(-2 pi i j q / R) -------
template <int R>
struct Dft;

template <>
struct Dft<2> {
    static __device__ __forceinline__ void run(cd (&v)[2]) {
        cd a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <>
struct Dft<4> {
    static __device__ __forceinline__ void run(cd (&v)[4]) {
        cd t0 = v[0] + v[2], t1 = v[0] - v[2];
        cd t2 = v[1] + v[3], t3 = mul_mi(v[1] - v[3]);
        v[0] = t0 + t2;
        v[1] = t1 + t3;
        v[2] = t0 - t2;
        v[3] = t1 - t3;
    }
};

template <>
struct Dft<5> {
    static __device__ __forceinline__ void run(cd (&v)[5]) {
        cd t1 = v[1] + v[4], t2 = v[2] + v[3], t3 = v[1] - v[4], t4 = v[2] - v[3];
        cd y0 = v[0] + t1 + t2;
        cd a1 = {v[0].x + kC5a * t1.x + kC5b * t2.x, v[0].y + kC5a * t1.y + kC5b * t2.y};
        cd a2 = {v[0].x + kC5b * t1.x + kC5a * t2.x, v[0].y + kC5b * t1.y + kC5a * t2.y};
        cd b1 = {kS5a * t3.x + kS5b * t4.x, kS5a * t3.y + kS5b * t4.y};
        cd b2 = {kS5b * t3.x - kS5a * t4.x, kS5b * t3.y - kS5a * t4.y};
        cd ib1 = mul_mi(b1), ib2 = mul_mi(b2);  // -i*b
        v[0] = y0;
        v[1] = a1 + ib1;
        v[4] = a1 - ib1;
        v[2] = a2 + ib2;
        v[3] = a2 - ib2;
    }
};

template <>
struct Dft<8> {
    static __device__ __forceinline__ void run(cd (&v)[8]) {
        cd e[4] = {v[0], v[2], v[4], v[6]};
        cd o[4] = {v[1], v[3], v[5], v[7]};
        Dft<4>::run(e);
        Dft<4>::run(o);
        cd o1 = {kR2 * (o[1].x + o[1].y), kR2 * (o[1].y - o[1].x)};   // * (1-i)/sqrt2
        cd o2 = mul_mi(o[2]);
        cd o3 = {kR2 * (o[3].y - o[3].x), -kR2 * (o[3].x + o[3].y)};  // * (-1-i)/sqrt2
        v[0] = e[0] + o[0];
        v[4] = e[0] - o[0];
        v[1] = e[1] + o1;
        v[5] = e[1] - o1;
        v[2] = e[2] + o2;
        v[6] = e[2] - o2;
        v[3] = e[3] + o3;
        v[7] = e[3] - o3;
    }
};

// multiply by W16^n = exp(-2 pi i n / 16) for the n that occur in a 4x4 split
template <int N>
__device__ __forceinline__ cd mul_w16(cd a) {
    if constexpr (N == 0) return a;
    else if constexpr (N == 1) return {kC8 * a.x + kS8 * a.y, kC8 * a.y - kS8 * a.x};
    else if constexpr (N == 2) return {kR2 * (a.x + a.y), kR2 * (a.y - a.x)};
    else if constexpr (N == 3) return {kS8 * a.x + kC8 * a.y, kS8 * a.y - kC8 * a.x};
    else if constexpr (N == 4) return mul_mi(a);
    else if constexpr (N == 6) return {kR2 * (a.y - a.x), -kR2 * (a.x + a.y)};
    else if constexpr (N == 9) return {-kC8 * a.x - kS8 * a.y, kS8 * a.x - kC8 * a.y};
    else return a;
}

template <>
struct Dft<16> {
    static __device__ __forceinline__ void run(cd (&v)[16]) {
        // j = 4*j1 + j2 ; q = q1 + 4*q2
        cd c0[4] = {v[0], v[4], v[8], v[12]};
        cd c1[4] = {v[1], v[5], v[9], v[13]};
        cd c2[4] = {v[2], v[6], v[10], v[14]};
        cd c3[4] = {v[3], v[7], v[11], v[15]};
        Dft<4>::run(c0);
        Dft<4>::run(c1);
        Dft<4>::run(c2);
        Dft<4>::run(c3);
        // twiddle W16^(j2*q1), then DFT4 over j2 for each q1
        cd r0[4] = {c0[0], c1[0], c2[0], c3[0]};
        cd r1[4] = {c0[1], mul_w16<1>(c1[1]), mul_w16<2>(c2[1]), mul_w16<3>(c3[1])};
        cd r2[4] = {c0[2], mul_w16<2>(c1[2]), mul_w16<4>(c2[2]), mul_w16<6>(c3[2])};
        cd r3[4] = {c0[3], mul_w16<3>(c1[3]), mul_w16<6>(c2[3]), mul_w16<9>(c3[3])};
        Dft<4>::run(r0);
        Dft<4>::run(r1);
        Dft<4>::run(r2);
        Dft<4>::run(r3);
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            v[0 + 4 * q2] = r0[q2];
            v[1 + 4 * q2] = r1[q2];
            v[2 + 4 * q2] = r2[q2];
            v[3 + 4 * q2] = r3[q2];
        }
    }
};

// inverse (unnormalised) DFT through the swap trick: idft(v) = swap(dft(swap(v)))
template <int R>
__device__ __forceinline__ void idft(cd (&v)[R]) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
        double t = v[j].x;
        v[j].x = v[j].y;
        v[j].y = t;
    }
    Dft<R>::run(v);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        double t = v[j].x;
        v[j].x = v[j].y;
        v[j].y = t;
    }
}

// ---- plans ---------------------------------------------------------------------
// A plan is a compile-time list of radices whose product is M, and the
// workgroup size NT that runs it.
template <int NT_, int... Rs>
struct Plan {
    static constexpr int NT = NT_;
    static constexpr int S = sizeof...(Rs);
    static constexpr int R[sizeof...(Rs)] = {Rs...};
    static constexpr int M = (Rs * ...);
    static constexpr int radix(int s) { return R[s]; }
    static constexpr int block(int s) {  // N_s
        int n = M;
        for (int i = 0; i < s; ++i) n /= R[i];
        return n;
    }
    static constexpr int lds_elems() { return (M + 15) / 16 * 16; }
};

template <class P, int s>
struct StageInfo {
    static constexpr int R = P::radix(s);
    static constexpr int N = P::block(s);
    static constexpr int L = N / R;
    static constexpr int TASKS = P::M / R;
    static constexpr int K = (TASKS + P::NT - 1) / P::NT;  // tasks per thread (max)
    static constexpr int TWSTEP = 2 * P::M / N;            // table stride: W_N^n = tw2[n*TWSTEP]
};

// One in-LDS forward stage (not first, not last): read, DFT, twiddle, write back.
template <class P, int s>
__device__ __forceinline__ void fwd_stage_lds(cd* __restrict__ lds, const cd* __restrict__ tw2,
                                              int tid) {
    using SI = StageInfo<P, s>;
#pragma unroll 1
    for (int m = 0; m < SI::K; ++m) {
        const int u = tid + m * P::NT;
        if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
            const int blk = u / SI::L, b = u - blk * SI::L;
            const int base = blk * SI::N + b;
            const int sb = sw(base);
            cd v[SI::R];
#pragma unroll
            for (int j = 0; j < SI::R; ++j) v[j] = lds[sw_off(sb, j * SI::L)];
            Dft<SI::R>::run(v);
            if (SI::L > 1) {
#pragma unroll
                for (int q = 1; q < SI::R; ++q) v[q] = cmul(v[q], tw2[q * b * SI::TWSTEP]);
            }
#pragma unroll
            for (int q = 0; q < SI::R; ++q) lds[sw_off(sb, q * SI::L)] = v[q];
        }
    }
}

// Inverse of stage s (conj twiddle, inverse DFT), in place.
template <class P, int s>
__device__ __forceinline__ void inv_stage_lds(cd* __restrict__ lds, const cd* __restrict__ tw2,
                                              int tid) {
    using SI = StageInfo<P, s>;
#pragma unroll 1
    for (int m = 0; m < SI::K; ++m) {
        const int u = tid + m * P::NT;
        if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
            const int blk = u / SI::L, b = u - blk * SI::L;
            const int base = blk * SI::N + b;
            const int sb = sw(base);
            cd v[SI::R];
#pragma unroll
            for (int q = 0; q < SI::R; ++q) v[q] = lds[sw_off(sb, q * SI::L)];
            if (SI::L > 1) {
#pragma unroll
                for (int q = 1; q < SI::R; ++q) v[q] = cmulc(v[q], tw2[q * b * SI::TWSTEP]);
            }
            idft<SI::R>(v);
#pragma unroll
            for (int j = 0; j < SI::R; ++j) lds[sw_off(sb, j * SI::L)] = v[j];
        }
    }
}

// Table reads that must not become flat loads: after the pointer laundering in the pair loop
// the compiler no longer knows the address space of tw2 and would emit flat_load + a wait for
// vmcnt(0) AND lgkmcnt(0), i.e. for every gather load in flight.  Wave-uniform entries go
// through the constant address space (s_load, counted by lgkmcnt only), per-lane entries
// through the global one.
typedef double tw_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cd tw_uniform(const cd* tw2, int idx) {
    const tw_d2 t = ((const tw_d2 __attribute__((address_space(4)))*)tw2)[idx];
    return cd{t.x, t.y};
}
__device__ __forceinline__ cd tw_lane(const cd* tw2, int idx) {
    const tw_d2 t = ((const tw_d2 __attribute__((address_space(1)))*)tw2)[idx];
    return cd{t.x, t.y};
}

template <class P, int s>
constexpr bool stage_seedable() {
    return StageInfo<P, s>::L > 1 && (P::NT % StageInfo<P, s>::L == 0);
}

// Inverse stage without strided table gathers (15 scattered 16-byte loads per radix-16
// butterfly are L2-request-bound): a seedable stage forms conj(W^{q b}) = conj(seed)^q from
// one table entry per thread, like the forward pass; stage 0 (run last, b = u) reads the
// forward pass-A first-stage table [q][u] = W_M^{q u} at tw2 + 2M, lane-contiguous.
template <class P, int s>
__device__ __forceinline__ void inv_stage_lds_fast(cd* __restrict__ lds,
                                                   const cd* __restrict__ tw2, int tid) {
    using SI = StageInfo<P, s>;
    if constexpr (s == 0 && SI::L > 1) {
        const cd* __restrict__ tbl = tw2 + 2 * P::M;
#pragma unroll 1
        for (int m = 0; m < SI::K; ++m) {
            const int u = tid + m * P::NT;
            if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
                const int sb = sw(u);
                cd v[SI::R];
#pragma unroll
                for (int q = 0; q < SI::R; ++q) v[q] = lds[sw_off(sb, q * SI::L)];
#pragma unroll
                for (int q = 1; q < SI::R; ++q) v[q] = cmulc(v[q], tw_lane(tbl, q * SI::L + u));
                idft<SI::R>(v);
#pragma unroll
                for (int j = 0; j < SI::R; ++j) lds[sw_off(sb, j * SI::L)] = v[j];
            }
        }
    } else if constexpr (stage_seedable<P, s>()) {
        const cd seed = tw_lane(tw2, (tid % SI::L) * SI::TWSTEP);
        const cd seed2 = cmul(seed, seed);
#pragma unroll 1
        for (int m = 0; m < SI::K; ++m) {
            const int u = tid + m * P::NT;
            if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
                const int blk = u / SI::L, b = u - blk * SI::L;
                const int base = blk * SI::N + b;
                const int sb = sw(base);
                cd v[SI::R];
#pragma unroll
                for (int q = 0; q < SI::R; ++q) v[q] = lds[sw_off(sb, q * SI::L)];
                cd wo = seed, we = seed2;  // seed^q for the current odd / even q
                v[1] = cmulc(v[1], wo);
                if constexpr (SI::R > 2) v[2] = cmulc(v[2], we);
#pragma unroll
                for (int q = 3; q < SI::R; ++q) {
                    if (q & 1) {
                        wo = cmul(wo, seed2);
                        v[q] = cmulc(v[q], wo);
                    } else {
                        we = cmul(we, seed2);
                        v[q] = cmulc(v[q], we);
                    }
                }
                idft<SI::R>(v);
#pragma unroll
                for (int j = 0; j < SI::R; ++j) lds[sw_off(sb, j * SI::L)] = v[j];
            }
        }
    } else {
        inv_stage_lds<P, s>(lds, tw2, tid);
    }
}

template <class P, int s>
__device__ __forceinline__ void inv_all_stages(cd* lds, const cd* tw2, int tid) {
    // per-thread twiddle seeds and addresses depend on tid only: inside a loop over atoms LICM
    // would hoist them out and keep (spill) them across the whole forward pipeline
    asm volatile("" : "+v"(tid));
    inv_stage_lds_fast<P, s>(lds, tw2, tid);
    __syncthreads();
    if constexpr (s > 0) inv_all_stages<P, s - 1>(lds, tw2, tid);
}

// In-LDS forward stage whose per-thread twiddle base b = u % L is the same for all
// of the thread's butterflies (NT % L == 0): no table loads.  The twiddles
// W^{q b} = seed^q are formed by repeated multiplication on the fly (two interleaved
// chains, odd and even q, so consecutive products are independent; the error grows
// by ~1 ulp per step, <= 8 steps per chain: far inside the 1e-10 budget).
template <class P, int s, class Hook>
__device__ __forceinline__ void fwd_stage_lds_seeded(cd* __restrict__ lds, cd seed, int tid,
                                                     Hook&& after_task) {
    using SI = StageInfo<P, s>;
    static_assert(P::NT % SI::L == 0, "seeded stage needs NT % L == 0");
    // The seed is the same for every column pair, so LICM would hoist the whole power
    // chain (R-1 complex values per stage) out of the pair loop and spill it; laundering
    // the seed keeps the chain where it is used.
    asm volatile("" : "+v"(seed.x), "+v"(seed.y));
    asm volatile("" : "+v"(tid));  // likewise the LDS addresses
    const cd seed2 = cmul(seed, seed);
    const double c2 = 2.0 * seed2.x;
#pragma unroll
    for (int m = 0; m < SI::K; ++m) {
        const int u = tid + m * P::NT;
        if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
            const int blk = u / SI::L, b = u - blk * SI::L;
            const int base = blk * SI::N + b;
            const int sb = sw(base);
            cd v[SI::R];
#pragma unroll
            for (int j = 0; j < SI::R; ++j) v[j] = lds[sw_off(sb, j * SI::L)];
            Dft<SI::R>::run(v);
            // seed^q by the three-term recurrence w_{q+2} = 2 cos(2 theta) w_q - w_{q-2} (two
            // FMAs per power instead of a complex product; odd and even q are independent
            // chains; <= 7 steps each, error growth ~q^2 ulp: far inside the 1e-10 budget)
            cd wo = seed, we = seed2;              // seed^q for the current odd / even q
            cd po = cd{seed.x, -seed.y}, pe = cd{1.0, 0.0};  // seed^(q-2)
            v[1] = cmul(v[1], wo);
            if constexpr (SI::R > 2) v[2] = cmul(v[2], we);
#pragma unroll
            for (int q = 3; q < SI::R; ++q) {
                if (q & 1) {
                    const cd n = {c2 * wo.x - po.x, c2 * wo.y - po.y};
                    po = wo;
                    wo = n;
                    v[q] = cmul(v[q], wo);
                } else {
                    const cd n = {c2 * we.x - pe.x, c2 * we.y - pe.y};
                    pe = we;
                    we = n;
                    v[q] = cmul(v[q], we);
                }
            }
#pragma unroll
            for (int q = 0; q < SI::R; ++q) lds[sw_off(sb, q * SI::L)] = v[q];
        }
        // all lanes (also those without a butterfly in this round) run the hook
        __builtin_amdgcn_sched_barrier(0);
        after_task(m);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace ta
