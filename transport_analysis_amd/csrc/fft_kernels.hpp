// fft_kernels.hpp — FFT-VACF kernels (K1..K3 of SURVEY.md section 8a).
//
// Replaces VelocityAutocorr._conclude_fft + tidynamics.acf
// (/root/reference/transport_analysis/velocityautocorr.py:208-215).
//
// Maths.  For a real column x of n_frames = T samples the reference needs
//   acf[k] = sum_i x[i] x[i+k] / (T-k),  k < T,
// which tidynamics evaluates as Re IFFT(|FFT(x padded to L)|^2) with L >= 2T-1.
// Here L = 2M with M >= T the smallest length of the form 2^a or 5*2^a, and:
//   * two adjacent real columns are packed into one complex series z = x + i*y;
//     Re IFFT(|Z|^2) = acf_x + acf_y exactly (the cross terms are odd), and the
//     reference only ever needs sums over columns (dims, then atoms);
//   * the 2M-point transform of a series whose upper half is zero splits into two
//     M-point transforms: even bins = FFT_M(z) ("pass A"), odd bins =
//     FFT_M(z[t] * exp(-i pi t / M)) ("pass B").  With the first radix R0 and
//     t = u + j*M/R0 the twist factors as exp(-i pi u/M) * W_{2 R0}^j: a
//     lane-uniform constant per input, and the lane-dependent part merges into
//     the stage twiddle: output q of butterfly u is scaled by W_{2M}^{u (2q + B)},
//     B = 0 for pass A and 1 for pass B;
//   * the timeseries is linear in the power spectra, so a workgroup accumulates
//     |Z|^2 over all its column pairs in registers (in the transform's own
//     digit-reversed order) and ONE inverse transform per launch, not per atom,
//     turns the summed spectrum into the lag-indexed sum.
// tw2 is the single twiddle table W_{2M}^n = exp(-i pi n / M), n < 2M.
// Algorithmic HBM bytes: every input element is needed once: T*A*D*8 bytes.
#pragma once
#include <hip/hip_runtime.h>
#include "agpr_slots.inc"
namespace ta {
template <class P>
__device__ __forceinline__ void agpr_fence();
}
#define TA_AGPR_FENCE_HOOK() ::ta::agpr_fence<P>()
#include "fft_engine.hpp"

namespace ta {

// ---- K1+K2: accumulate power spectra over column pairs ------------------------
//
// Register plan.  At one wave per SIMD a wave owns 512 registers, of which VALU
// instructions can only address the lower 256 (v0..v255); the upper half (a0..a255,
// the AGPRs) is reachable by v_accvgpr_read/write and by global loads/stores.  The
// gathered column pair (up to 40 complex f64 = 160 dwords per thread) and one
// accumulator set (up to 48 f64 = 96 dwords) are "cold" for most of an iteration and
// are kept there EXPLICITLY, at fixed register numbers, through the accessors of
// agpr_slots.inc (inline asm).  The compiler DOES use AGPRs of its own under pressure and cannot
// be told the manual slots are live: code-free clobber fences keep its live ranges below the
// manual range and tools/check_agpr.py (a Makefile step on the generated ISA) fails the build
// if any compiler-owned instruction reads or writes inside it.
// That makes the software pipeline deterministic: the gather is issued as
// global_load_dwordx4 with an AGPR destination, nobody waits for it or spills it, and
// the compiler's own 256 VGPRs are left for one butterfly's working set.
#include <type_traits>
#include <utility>

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    [&]<int... I>(std::integer_sequence<int, I...>) {
        (f(std::integral_constant<int, I>{}), ...);
    }(std::make_integer_sequence<int, N>{});
}

// Manual slots sit at the TOP of the AGPR file: [agpr_base<P>(), 256) = gathered pair
// (4 dwords per complex value) followed by the accumulators (2 dwords per bin).  Everything
// below is left to the compiler: under register pressure it parks values of its own in the
// lowest free AGPRs and cannot be told that the manual slots are live.  The fences below
// keep its live ranges out of the manual range and tools/check_agpr.py verifies that at
// build time.
//
// "Landing" plans (one wave per SIMD, NT >= 256): the area after the pair is not an
// accumulator set but a LANDING ZONE for the first-stage butterflies K0-KL..K0-1 of the NEXT
// pair (KL = K0/2), filled by gather loads issued during pass A, when the pair area is still
// needed and the vector-memory pipe would otherwise idle; pass B refills the pair slots of
// butterflies 0..K0-KL-1 as it frees them and copies the landed half over once its own
// butterflies K0-KL.. are done.  Half of the gather's pipe time thus moves out of pass B.
// Both accumulator sets of these plans live in VGPRs while they are used (pass A's set is
// loaded from / stored to the workgroup's global block around pass A's last stage).
template <class P>
constexpr bool landing() {
    // measured per plan: the 128-thread 5*2^a plan spills with the landing zone and is faster without
    return P::kLanding && StageInfo<P, 0>::K >= 2 && (P::NT >= 256 || P::radix(0) != 5);
}
template <class P>
constexpr int land_tasks() { return landing<P>() ? StageInfo<P, 0>::K / 2 : 0; }
template <class P>
constexpr int land_elems() { return land_tasks<P>() * StageInfo<P, 0>::R; }
template <class P>
constexpr int agpr_acc_dwords() {
    if (landing<P>()) return 4 * land_elems<P>();
    return (StageInfo<P, P::S - 1>::K * StageInfo<P, P::S - 1>::R * 2 + 3) / 4 * 4;
}
template <class P>
constexpr int agpr_base() {  // multiple of 16
    return (256 - 4 * StageInfo<P, 0>::K * StageInfo<P, 0>::R - agpr_acc_dwords<P>()) / 16 * 16;
}
template <class P>
constexpr int agpr_acc_base() { return agpr_base<P>() + 4 * StageInfo<P, 0>::K * StageInfo<P, 0>::R; }

template <int I>
__device__ __forceinline__ double ag_read_f64() {
    return __hiloint2double((int)AG<I + 1>::r(), (int)AG<I>::r());
}
template <int I>
__device__ __forceinline__ void ag_write_f64(double x) {
    AG<I>::w((unsigned)__double2loint(x));
    AG<I + 1>::w((unsigned)__double2hiint(x));
}

template <int D, int S>
__device__ __forceinline__ void ag_mov() {  // aD = aS
    asm volatile("v_accvgpr_mov_b32 a[%0], a[%1]" ::"n"(D), "n"(S));
}

// Issue gather loads [LO, HI) (flat index f = m*R0 + j -> a[4f..4f+3]) of one column pair.
// Addressing: a lane's row tid sits lane_off = tid*ld_row*8 bytes into the column (computed once
// per kernel), element f adds the wave-uniform (m*NT + j*L)*ld_row*8 (scalar multiply), so a
// load costs one 64-bit vector add.  Rows past the end (t >= T, the zero padding) read the
// 32 zero bytes behind the twiddle tables instead (compare + select on the address): the
// parked registers then hold the padded series itself and nothing is masked when they are
// read (twice per pair).  Non-VEC (rows not 16-byte aligned or an odd last column): two
// 8-byte loads, or one plus a zero imaginary part.
// kind (wave-uniform, non-VEC kernels): 0 = one column (imaginary part zero), 1 = two columns,
// two 8-byte loads, 2 = two columns at a 16-byte aligned address, one load.
// FMAX: only elements f < FMAX are loaded; DSTOFF: dword offset added to the destination slot
// (the landing zone is the pair area shifted by 4*land_elems dwords).
template <class P, bool VEC, int LO, int HI, int FMAX = 1 << 20, int DSTOFF = 0>
__device__ __forceinline__ void gather_issue_range(const double* __restrict__ col, long ld_row,
                                                   int T, int kind, int tid,
                                                   unsigned long lane_off,
                                                   const cd* __restrict__ zeros) {
    using SI = StageInfo<P, 0>;
    // form this piece's addresses here and now (hoisted out, the 40 row addresses of a
    // pair would occupy 80 VGPRs for the whole pass, their uniform parts 80 SGPRs)
    asm volatile("" : "+s"(ld_row), "+s"(T));
    static_for<(HI > LO ? HI - LO : 0)>([&](auto i) {
        constexpr int f = LO + decltype(i)::value;
        if constexpr (f < SI::K * SI::R && f < FMAX) {
            constexpr int m = f / SI::R, j = f % SI::R;
            constexpr int dst = agpr_base<P>() + 4 * f + DSTOFF;
            constexpr int r0 = m * P::NT + j * SI::L;  // row of lane 0
            // uniform part on the scalar unit (left alone the compiler folds it into a
            // quarter-rate v_mad_u64_u32 per load)
            unsigned long base = reinterpret_cast<unsigned long>(col) + (unsigned long)r0 * ((unsigned long)ld_row * 8ul);
            asm volatile("" : "+s"(base));
            const double* p = reinterpret_cast<const double*>(base + lane_off);
            if (T - r0 < P::NT) {  // wave-uniform: this element reaches into the padding
                asm volatile("");  // keep it a scalar branch (two selects per load otherwise)
                if (!(tid < T - r0)) p = reinterpret_cast<const double*>(zeros);
            }
            if constexpr (VEC) {
                ag_load4<dst>(p);
            } else if (kind == 2) {
                ag_load4<dst>(p);
            } else {
                ag_load2<dst>(p);
                if (kind == 1) {
                    ag_load2<dst + 2>(p + 1);
                } else {
                    AG<dst + 2>::w(0u);
                    AG<dst + 3>::w(0u);
                }
            }
        }
    });
}

// First stage from the parked pair (read-only: pass B reads it again).
// Twiddles W_2M^{u(2q+B)}, u = tid + m*NT, without table gathers: g = W^{2u} = W^{2 tid} *
// W^{2 m NT} and h = W^{uB} = W^{tid B} * W^{m NT B} (per-lane factor loaded once per pass
// while no gather is in flight, wave-uniform factor by scalar load), then the powers
// h, h g, h g^2, ... by repeated multiplication (two interleaved chains; <= R0/2 steps each).
template <class P, bool PASSB, class Hook>
__device__ __forceinline__ void first_stage_from_agpr(cd* __restrict__ lds,
                                                      const cd* __restrict__ tw2, int T, int tid,
                                                      Hook&& after_task) {
    using SI = StageInfo<P, 0>;
    const cd G = tw_lane(tw2, 2 * tid);                           // W_2M^{2 tid}
    const cd H = PASSB ? tw_lane(tw2, tid) : cd{1.0, 0.0};        // W_2M^{tid}
    static_for<SI::K>([&](auto mm) {
        constexpr int m = decltype(mm)::value;
        const int u = tid + m * P::NT;
        if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
            cd w[SI::R];
            static_for<SI::R>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                constexpr int a = agpr_base<P>() + 4 * (m * SI::R + j);
                w[j] = cd{ag_read_f64<a>(), ag_read_f64<a + 2>()};  // zero-padded by the gather
            });
            if constexpr (PASSB) {
                // lane-uniform part of the twist: W_{2 R0}^j = tw2[j * L]
#pragma unroll
                for (int j = 1; j < SI::R; ++j) w[j] = cmul(w[j], tw_uniform(tw2, j * SI::L));
            }
            agpr_fence<P>();
            Dft<SI::R>::run(w);
            agpr_fence<P>();
            constexpr int eg = (2 * m * P::NT) % (2 * P::M), eh = (m * P::NT) % (2 * P::M);
            const cd g = m == 0 ? G : cmul(G, tw_uniform(tw2, eg));
            cd h = H;
            if constexpr (PASSB && m > 0) h = cmul(H, tw_uniform(tw2, eh));
            const cd g2 = cmul(g, g);
            cd te = h, to = cmul(h, g);  // h g^q for the current even / odd q
            if constexpr (PASSB) w[0] = cmul(w[0], te);
            if constexpr (SI::R > 1) w[1] = cmul(w[1], to);
#pragma unroll
            for (int q = 2; q < SI::R; ++q) {
                if (q & 1) {
                    to = cmul(to, g2);
                    w[q] = cmul(w[q], to);
                } else {
                    te = cmul(te, g2);
                    w[q] = cmul(w[q], te);
                }
            }
            agpr_fence<P>();
            const int sb = sw(u);
#pragma unroll
            for (int q = 0; q < SI::R; ++q) lds[sw_off(sb, q * SI::L)] = w[q];
        }
        agpr_fence<P>();
        __builtin_amdgcn_sched_barrier(0);
        after_task(m);  // butterfly m's slots of the parked pair are dead from here on (pass B)
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Last forward stage fused with |.|^2 accumulation into the parked accumulators.
template <class P, class Hook>
__device__ __forceinline__ void last_stage_acc_agpr(const cd* __restrict__ lds, int tid,
                                                    Hook&& after_task) {
    using SI = StageInfo<P, P::S - 1>;
    asm volatile("" : "+v"(tid));  // LDS addresses re-formed here, not carried (and spilled) across stages
    static_for<SI::K>([&](auto mm) {
        constexpr int m = decltype(mm)::value;
        const int u = tid + m * P::NT;
        if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
            const int sb = sw(u * SI::R);
            cd v[SI::R];
#pragma unroll
            for (int j = 0; j < SI::R; ++j) v[j] = lds[sw_off(sb, j)];
            agpr_fence<P>();
            Dft<SI::R>::run(v);
            agpr_fence<P>();
            static_for<SI::R>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                constexpr int a = agpr_acc_base<P>() + 2 * (m * SI::R + q);
                ag_write_f64<a>(ag_read_f64<a>() + norm2(v[q]));
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        after_task(m);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Same, accumulating into an ordinary register array (the second accumulator set).
template <class P, class Hook>
__device__ __forceinline__ void last_stage_acc_regs(
    const cd* __restrict__ lds,
    double (&acc)[StageInfo<P, P::S - 1>::K][StageInfo<P, P::S - 1>::R], int tid, Hook&& after_task) {
    using SI = StageInfo<P, P::S - 1>;
    if constexpr (!P::kLanding) asm volatile("" : "+v"(tid));  // see fwd_stage_lds_seeded
    static_for<SI::K>([&](auto mm) {
        constexpr int m = decltype(mm)::value;
        const int u = tid + m * P::NT;
        if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
            const int sb = sw(u * SI::R);
            cd v[SI::R];
#pragma unroll
            for (int j = 0; j < SI::R; ++j) v[j] = lds[sw_off(sb, j)];
            agpr_fence<P>();
            Dft<SI::R>::run(v);
            agpr_fence<P>();
#pragma unroll
            for (int q = 0; q < SI::R; ++q) acc[m][q] += norm2(v[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
        after_task(m);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Same, accumulating into the workgroup's accumulator block in global memory (L2/MALL
// resident, [quad][thread] x double2): a butterfly's R accumulators are loaded one butterfly
// ahead and stored right after the update, so only 2*R of them are ever in registers.
template <class P, class Hook>
__device__ __forceinline__ void last_stage_acc_global(const cd* __restrict__ lds,
                                                      double* __restrict__ blk, int tid,
                                                      Hook&& after_task) {
    using SI = StageInfo<P, P::S - 1>;
    static_assert(SI::R % 2 == 0, "accumulators move as double2");
    constexpr int Q = SI::R / 2;  // double2 per butterfly
    // global address space kept explicit: a laundered generic pointer would turn these into
    // flat loads/stores, whose waits drain every load in flight
    typedef tw_d2 __attribute__((address_space(1)))* gptr;
    tw_d2 buf[2][Q];
    auto row = [&](int quad) {
        double* r = blk + (long)quad * 2 * P::NT;
        asm volatile("" : "+s"(r));  // uniform row base: SGPR base + one shared VGPR offset
        return (gptr)r;
    };
#pragma unroll
    for (int h = 0; h < Q; ++h) buf[0][h] = row(h)[tid];
    static_for<SI::K>([&](auto mm) {
        constexpr int m = decltype(mm)::value;
        const int u = tid + m * P::NT;
        if constexpr (m + 1 < SI::K) {
#pragma unroll
            for (int h = 0; h < Q; ++h) buf[(m + 1) & 1][h] = row((m + 1) * Q + h)[tid];
        }
        if (SI::TASKS % P::NT == 0 || u < SI::TASKS) {
            const int sb = sw(u * SI::R);
            cd v[SI::R];
#pragma unroll
            for (int j = 0; j < SI::R; ++j) v[j] = lds[sw_off(sb, j)];
            agpr_fence<P>();
            Dft<SI::R>::run(v);
            agpr_fence<P>();
#pragma unroll
            for (int h = 0; h < Q; ++h) {
                tw_d2 a = buf[m & 1][h];
                a.x += norm2(v[2 * h]);
                a.y += norm2(v[2 * h + 1]);
                row(m * Q + h)[tid] = a;
            }
        }
        agpr_fence<P>();
        __builtin_amdgcn_sched_barrier(0);
        after_task(m);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// The workgroup's accumulator blocks in global memory (one per pass; what the launch hands to
// k_sum_partials_perm, and where the landing pipeline streams pass A's set) are laid out
// [quad][thread] x 2 doubles, so a wave's accesses are contiguous; a thread only ever touches
// its own slots, so program order is the only ordering needed.
template <class P>
constexpr int acc_quads() {
    return (StageInfo<P, P::S - 1>::K * StageInfo<P, P::S - 1>::R * 2 + 3) / 4;
}
template <class P>
__device__ __forceinline__ void agpr_fence() {
    static_assert(agpr_base<P>() >= 0, "manual AGPR slots exceed the AGPR file");
    agpr_fence_from<agpr_base<P>()>();
}

// Store the AGPR-resident accumulator set to a block (end of the kernel).
template <class P>
__device__ __forceinline__ void acc_swap_out(double* __restrict__ blk, int tid) {
    asm volatile("s_nop 4" ::: "memory");  // VALU AGPR writes -> VMEM store data (no auto padding in asm)
    static_for<acc_quads<P>()>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        ag_store4<agpr_acc_base<P>() + 4 * q>(blk + 2 * ((long)q * P::NT + tid));
    });
}

template <class P, int s>
__device__ __forceinline__ cd stage_seed(const cd* __restrict__ tw2, int tid) {
    using SI = StageInfo<P, s>;
    if constexpr (stage_seedable<P, s>()) return tw_lane(tw2, (tid % SI::L) * SI::TWSTEP);
    else return cd{1.0, 0.0};
}

// Number of butterfly rounds a thread runs in the seeded mid stages s..S-2 (MIDSLOTS)
// and in the last stage: the slots between which the next pair's gather loads are spread.
template <class P, int s>
constexpr int mid_slots_from() {
    if constexpr (s <= P::S - 2)
        return (stage_seedable<P, s>() ? StageInfo<P, s>::K : 0) + mid_slots_from<P, s + 1>();
    else
        return 0;
}

// mid stages 1..S-2 (plans have S <= 5, so at most three of them).  `hook(slot)` is
// called once per butterfly round with a compile-time slot number.
template <class P, int s, int SLOT0, class Hook>
__device__ __forceinline__ void mid_stages_seeded(cd* lds, const cd* tw2, const cd (&seed)[4],
                                                  int tid, Hook&& hook) {
    if constexpr (s <= P::S - 2) {
        if constexpr (stage_seedable<P, s>()) {
            fwd_stage_lds_seeded<P, s>(lds, seed[s], tid, [&](int m) { hook(SLOT0 + m); });
            __syncthreads();
            mid_stages_seeded<P, s + 1, SLOT0 + StageInfo<P, s>::K>(lds, tw2, seed, tid, hook);
        } else {
            fwd_stage_lds<P, s>(lds, tw2, tid);
            __syncthreads();
            mid_stages_seeded<P, s + 1, SLOT0>(lds, tw2, seed, tid, hook);
        }
    }
}

template <class P>
constexpr bool plan_all_mid_seedable() {
    bool ok = true;
    if constexpr (P::S > 2) ok = ok && stage_seedable<P, 1>();
    if constexpr (P::S > 3) ok = ok && stage_seedable<P, 2>();
    if constexpr (P::S > 4) ok = ok && stage_seedable<P, 3>();
    return ok;
}

// Shared epilogue: LDS holds q = IDFT_M(P_A + i P_B) in natural order.
// lag-n value = Re(a + conj(W_{2M}^n) * b) / (2M) / (T-n), with
// a = (q[n] + conj(q[M-n]))/2, b = (q[n] - conj(q[M-n]))/(2i).
template <class P>
__device__ __forceinline__ double lag_value(const cd* __restrict__ lds,
                                            const cd* __restrict__ tw2, int n, int T) {
    const cd qn = lds[sw(n)];
    const cd qm = lds[sw((P::M - n) % P::M)];
    const cd a = {0.5 * (qn.x + qm.x), 0.5 * (qn.y - qm.y)};
    // (qn - conj(qm)) / (2i) = ( (qn.y + qm.y) - i (qn.x - qm.x) ) / 2
    const cd b = {0.5 * (qn.y + qm.y), -0.5 * (qn.x - qm.x)};
    const cd w = tw_lane(tw2, n);  // exp(-i pi n / M); need its conjugate
    const double re = a.x + (b.x * w.x + b.y * w.y);
    return re / (2.0 * (double)P::M * (double)(T - n));  // 2M(T-n) < 2^53: exact product
}

// STAMP (diagnostic builds only): lane 0 accumulates s_memtime deltas per phase.
#define TA_STAMP(idx)                                                              \
    if constexpr (STAMP) {                                                         \
        __builtin_amdgcn_sched_barrier(0);                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();              \
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */                      \
        __builtin_amdgcn_sched_barrier(0);                                         \
        st_acc[idx] += now_ - st_prev;                                             \
        st_prev = now_;                                                            \
    }

// Persistent workgroups (one wave per SIMD at the big plans).  A workgroup gathers a
// column pair ONCE and runs both passes from the parked registers (A: even bins, B: odd
// bins), so every input element is requested from L2 once.  With a grid that is a
// multiple of 8, blocks b and b+8 share an XCD and an XCD's workgroups take consecutive
// pairs, so the 8 pairs of a 128-byte line go through one L2 and the line leaves HBM once.
//
// Plain pipeline (by-particle mode, the 5*2^a plans below 256 threads):
//   wait gather -> first stage A -> mid A -> last A (accumulators in AGPRs)
//   -> first stage B -> mid B -> last B (accumulators in VGPRs), with the NEXT pair's gather
//   loads spread between pass B's butterfly rounds (its first stage frees the pair slots).
// Landing pipeline (landing<P>(), see above): the gather is split over both passes.
// The 16-byte strided gather is request-bound (a pure gather of this shape peaks at
// ~1.9-2.3 TB/s on MI355X, ~0.25 lane-requests/clk/CU) and every load costs the issuing wave
// ~400-500 cycles of issue back-pressure; issued as one burst the 40 loads per thread would
// stall the wave for longer than a whole pass takes.
//
// accg: [gridDim.x][2][acc_quads*2*NT] float64 (pass A block, pass B block), zeroed by
// the caller; k_sum_partials_perm restores the transform's digit-reversed bin order.
//
// BYP (by-particle mode): a workgroup takes whole atoms (atom = slot + k*gridDim.x; units of
// an atom: (x,y) then (z,0) for D = 3, one unit for D <= 2); after an atom's last unit the
// two accumulator sets are the atom's own power spectrum: they go to LDS, one inverse
// transform, and the lag values are written to
// by_particle[:, atom] while the NEXT atom's first unit is already in flight into the parked
// registers (the lag sums over atoms are row sums of by_particle: k_row_sums).
template <class P, bool VEC, bool STAMP = false, bool BYP = false>
__global__ void __launch_bounds__(P::NT)
    k_fft_accum(const double* __restrict__ vel, long ld_row, long pair_stride, int T, long n_cols,
                const cd* __restrict__ tw2, double* __restrict__ accg, int flags,
                unsigned long long* __restrict__ stamps = nullptr, int D = 0, long n_atoms = 0,
                double* __restrict__ by_particle = nullptr, long ld_bp = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    using SL = StageInfo<P, P::S - 1>;
    using S0 = StageInfo<P, 0>;
    static_assert(P::S <= 5, "seed array sized for S <= 5");
    static_assert(agpr_acc_base<P>() + agpr_acc_dwords<P>() <= 256, "manual AGPR slots exceed a255");
    static_assert(agpr_base<P>() >= 16, "leave at least a0..a15 to the compiler");
    asm volatile("; TA_AGPR_MANUAL_RANGE %0 %1" ::"n"(agpr_base<P>()), "n"(256));  // tools/check_agpr.py
    constexpr long ACC_BLK = (long)acc_quads<P>() * 2 * P::NT;
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = 0;
    if constexpr (STAMP) st_prev = __builtin_amdgcn_s_memtime();

    const int nwg = gridDim.x, wg = blockIdx.x;
    int slot = wg;
    if (nwg % 8 == 0) slot = (wg % 8) * (nwg / 8) + wg / 8;
    int tid = threadIdx.x;
    double* blkA = accg + (long)wg * 2 * ACC_BLK;
    double* blkB = blkA + ACC_BLK;

    cd seed[4];
    seed[0] = cd{1.0, 0.0};
    seed[1] = stage_seed<P, (P::S > 2 ? 1 : 0)>(tw2, tid);
    seed[2] = stage_seed<P, (P::S > 3 ? 2 : 0)>(tw2, tid);
    seed[3] = stage_seed<P, (P::S > 4 ? 3 : 0)>(tw2, tid);

    constexpr int NLOAD = S0::K * S0::R;
    constexpr int MIDSLOTS = mid_slots_from<P, 1>();
    // gather slots: pass B's first-stage butterflies (each frees its own R0 registers), then
    // the butterfly rounds of its mid and last stages.  A mid-stage round lasts about twice as
    // long as a first- or last-stage round, so it carries twice the loads: the gather is
    // L2-request-bound and anything issued faster than it drains only blocks the wave.
    constexpr int NSLOT = S0::K + MIDSLOTS + SL::K;
    constexpr int UNIT = (NLOAD + S0::K + 2 * MIDSLOTS + SL::K - 1) / (S0::K + 2 * MIDSLOTS + SL::K);
    static_assert(UNIT <= S0::R, "a first-stage slot may only refill registers already consumed");
    static_assert(NSLOT <= 32, "extend the TA_PIECE list");

    // pass A accumulators: manual AGPR slots (landing plans: the workgroup's global block,
    // in VGPRs only around pass A's last stage); pass B accumulators: ordinary registers
    if constexpr (!landing<P>())
        static_for<SL::K * SL::R>([&](auto dd) { ag_write_f64<agpr_acc_base<P>() + 2 * decltype(dd)::value>(0.0); });
    double accB[SL::K][SL::R];
#pragma unroll
    for (int m = 0; m < SL::K; ++m)
#pragma unroll
        for (int q = 0; q < SL::R; ++q) accB[m][q] = 0.0;

    // work units of this workgroup: unit i = column pair slot + i*nwg, or (BYP) unit
    // i % ppa of atom slot + (i / ppa)*nwg
    const int ppa = BYP ? (D + 1) / 2 : 1;
    const long n_mine = BYP ? (slot < n_atoms ? ((n_atoms - slot + nwg - 1) / nwg) * ppa : 0)
                            : ((n_cols + 1) / 2 > slot ? ((n_cols + 1) / 2 - slot + nwg - 1) / nwg : 0);
    // BYP, D = 3: the two-column unit is the 16-byte ALIGNED pair of the atom's three columns,
    // (x,y) for even atoms and (y,z) for odd ones (the sum over the atom's columns does not
    // care), so it is one load per row whenever the slab itself is 16-byte aligned.
    const bool slab16 = ((reinterpret_cast<unsigned long long>(vel) | ((unsigned long long)ld_row * 8)) & 15) == 0;
    // column c of the shard starts at vel + (c/2)*pair_stride + (c&1) (rows ld_row elements
    // apart): pair-major slabs have ld_row = 2, pair_stride = 2*pitch (layout.hip)
    auto col_ptr = [&](long c) -> const double* { return vel + (c >> 1) * pair_stride + (c & 1); };
    auto unit_col = [&](long i) -> const double* {
        if constexpr (BYP) {
            const long atom = slot + (i / ppa) * nwg;
            if (D == 3) {
                const int odd = (int)(atom & 1);
                return col_ptr(atom * 3 + ((i % ppa) == 0 ? odd : (odd ? 0 : 2)));
            }
            return col_ptr(atom * D);
        } else {
            return vel + (slot + i * nwg) * pair_stride;
        }
    };
    auto unit_kind = [&](long i) -> int {
        if constexpr (BYP) {
            if (D == 3) return (i % ppa) == 0 ? (slab16 ? 2 : 1) : 0;
            if (D == 2) return slab16 ? 2 : 1;
            return 0;
        } else {
            // a pair of a 16-byte aligned slab is one 16-byte load (pair_stride = 2: every
            // pair starts at an even column); an odd last column has no partner
            return 2 * (slot + i * nwg) + 1 < n_cols ? (slab16 && pair_stride % 2 == 0 ? 2 : 1) : 0;
        }
    };
    long unit = 0;
    // byte offset of this lane's row tid inside any column; 32 zero bytes behind the tables
    const unsigned long lane_off = (unsigned long)(unsigned)tid * ((unsigned long)ld_row * 8ul);
    const cd* zeros = tw2 + 4 * P::M;
    if (unit < n_mine)
        gather_issue_range<P, VEC, 0, NLOAD>(unit_col(0), ld_row, T, unit_kind(0), tid, lane_off, zeros);
    auto no_hook = [](int) {};
    while (unit < n_mine) {
        // per-lane addresses and table offsets depend on tid/ld_row only: keep LICM from
        // hoisting (and spilling) them out of the pair loop
        asm volatile("" : "+s"(tw2), "+s"(ld_row), "+v"(tid));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the gathered pair has landed
        const long next = unit + 1;
        const bool more = next < n_mine;
        const double* ncol = unit_col(more ? next : unit);
        const int nkind = unit_kind(more ? next : unit);
        if constexpr (landing<P>()) {
            // ================= landing plans =================
            constexpr int KL = land_tasks<P>(), NL = land_elems<P>(), NB = NLOAD - NL;
            constexpr int WSLOTS = S0::K + 2 * MIDSLOTS + SL::K;  // mid rounds count twice
            constexpr int UA = (NL + WSLOTS - 1) / WSLOTS, UB = (NB + WSLOTS - 1) / WSLOTS;
            static_assert(UB <= S0::R, "a first-stage slot may only refill registers already consumed");
#define TA_W(S) ((S) <= S0::K ? (S) : (S) <= S0::K + MIDSLOTS ? S0::K + 2 * ((S)-S0::K) \
                                                             : S0::K + 2 * MIDSLOTS + ((S)-S0::K - MIDSLOTS))
            // pass A: elements NB.. of the next unit -> landing zone
            auto hookA = [&](int slot_) {
                if (more) {
#define TA_PIECE(S)                                                                            \
    if (slot_ == S)                                                                            \
        gather_issue_range<P, VEC, NB + TA_W(S) * UA, NB + TA_W((S) + 1) * UA, NLOAD, 4 * NL>( \
            ncol, ld_row, T, nkind, tid, lane_off, zeros);
                    TA_PIECE(0) TA_PIECE(1) TA_PIECE(2) TA_PIECE(3) TA_PIECE(4) TA_PIECE(5)
                    TA_PIECE(6) TA_PIECE(7) TA_PIECE(8) TA_PIECE(9) TA_PIECE(10) TA_PIECE(11)
                    TA_PIECE(12) TA_PIECE(13) TA_PIECE(14) TA_PIECE(15) TA_PIECE(16) TA_PIECE(17)
                    TA_PIECE(18) TA_PIECE(19) TA_PIECE(20) TA_PIECE(21) TA_PIECE(22) TA_PIECE(23)
                    TA_PIECE(24) TA_PIECE(25) TA_PIECE(26) TA_PIECE(27) TA_PIECE(28) TA_PIECE(29)
                    TA_PIECE(30) TA_PIECE(31)
#undef TA_PIECE
                }
            };
            // pass B: elements 0..NB-1 of the next unit -> pair slots of the butterflies already
            // done; after butterfly m >= K0-KL the landed elements of that butterfly move over
            auto hookB = [&](int slot_) {
                if (more) {
#define TA_PIECE(S)                                                                          \
    if (slot_ == S) {                                                                        \
        gather_issue_range<P, VEC, TA_W(S) * UB, TA_W((S) + 1) * UB, NB>(ncol, ld_row, T,    \
                                                     nkind, tid, lane_off, zeros);          \
        if constexpr ((S) >= S0::K - KL && (S) < S0::K) {                                    \
            static_for<4 * S0::R>([&](auto dd) {                                             \
                constexpr int a = agpr_base<P>() + 4 * (S)*S0::R + decltype(dd)::value;      \
                ag_mov<a, a + 4 * NL>();                                                     \
            });                                                                              \
        }                                                                                    \
    }
                    TA_PIECE(0) TA_PIECE(1) TA_PIECE(2) TA_PIECE(3) TA_PIECE(4) TA_PIECE(5)
                    TA_PIECE(6) TA_PIECE(7) TA_PIECE(8) TA_PIECE(9) TA_PIECE(10) TA_PIECE(11)
                    TA_PIECE(12) TA_PIECE(13) TA_PIECE(14) TA_PIECE(15) TA_PIECE(16) TA_PIECE(17)
                    TA_PIECE(18) TA_PIECE(19) TA_PIECE(20) TA_PIECE(21) TA_PIECE(22) TA_PIECE(23)
                    TA_PIECE(24) TA_PIECE(25) TA_PIECE(26) TA_PIECE(27) TA_PIECE(28) TA_PIECE(29)
                    TA_PIECE(30) TA_PIECE(31)
#undef TA_PIECE
                }
            };
#undef TA_W
            // ---- pass A: even bins
            first_stage_from_agpr<P, false>(lds, tw2, T, tid, hookA);
            TA_STAMP(0)
            __syncthreads();
            mid_stages_seeded<P, 1, S0::K>(lds, tw2, seed, tid, hookA);
            TA_STAMP(1)
            last_stage_acc_global<P>(lds, blkA, tid, [&](int m) { hookA(S0::K + MIDSLOTS + m); });
            TA_STAMP(2)
            __syncthreads();
            // ---- pass B: odd bins.  Everything issued so far (the landing loads, the
            // accumulator block) must be complete before landed data is moved.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            first_stage_from_agpr<P, true>(lds, tw2, T, tid, hookB);
            TA_STAMP(3)
            __syncthreads();
            mid_stages_seeded<P, 1, S0::K>(lds, tw2, seed, tid, hookB);
            TA_STAMP(4)
            last_stage_acc_regs<P>(lds, accB, tid, [&](int m) { hookB(S0::K + MIDSLOTS + m); });
            TA_STAMP(5)
            __syncthreads();
        } else {
        // ---- pass A: even bins
        first_stage_from_agpr<P, false>(lds, tw2, T, tid, no_hook);
        TA_STAMP(0)
        __syncthreads();
        mid_stages_seeded<P, 1, 0>(lds, tw2, seed, tid, no_hook);
        TA_STAMP(1)
        last_stage_acc_agpr<P>(lds, tid, no_hook);
        TA_STAMP(2)
        __syncthreads();
        // ---- pass B: odd bins; after its first stage the parked pair is dead and is
        // refilled with the next pair while pass B's butterflies run
        auto hook = [&](int slot_) {
            if (more) {
#define TA_LO(S) ((S) <= S0::K ? (S)*UNIT                                                   \
                 : (S) <= S0::K + MIDSLOTS ? S0::K * UNIT + ((S)-S0::K) * 2 * UNIT           \
                                           : S0::K * UNIT + MIDSLOTS * 2 * UNIT + ((S)-S0::K - MIDSLOTS) * UNIT)
#define TA_PIECE(S)                                                                         \
    if (slot_ == S)                                                                         \
        gather_issue_range<P, VEC, TA_LO(S), TA_LO((S) + 1)>(ncol, ld_row, T, nkind, tid, lane_off, zeros);
                TA_PIECE(0) TA_PIECE(1) TA_PIECE(2) TA_PIECE(3) TA_PIECE(4) TA_PIECE(5)
                TA_PIECE(6) TA_PIECE(7) TA_PIECE(8) TA_PIECE(9) TA_PIECE(10) TA_PIECE(11)
                TA_PIECE(12) TA_PIECE(13) TA_PIECE(14) TA_PIECE(15) TA_PIECE(16) TA_PIECE(17)
                TA_PIECE(18) TA_PIECE(19) TA_PIECE(20) TA_PIECE(21) TA_PIECE(22) TA_PIECE(23)
                TA_PIECE(24) TA_PIECE(25) TA_PIECE(26) TA_PIECE(27) TA_PIECE(28) TA_PIECE(29)
                TA_PIECE(30) TA_PIECE(31)
#undef TA_PIECE
#undef TA_LO
            }
        };
        first_stage_from_agpr<P, true>(lds, tw2, T, tid, hook);
        TA_STAMP(3)
        __syncthreads();
        mid_stages_seeded<P, 1, S0::K>(lds, tw2, seed, tid, hook);
        TA_STAMP(4)
        last_stage_acc_regs<P>(lds, accB, tid, [&](int m) { hook(S0::K + MIDSLOTS + m); });
        TA_STAMP(5)
        __syncthreads();
        }  // !landing
        if constexpr (STAMP) st_acc[7] += 1;
        if constexpr (BYP) {
            if ((int)(unit % ppa) == ppa - 1) {
                const long atom = slot + (unit / ppa) * nwg;
                // ---- this atom's spectrum -> LDS (digit-reversed order), accumulators reset
                if constexpr (landing<P>()) {
                    // pass A's set is in the workgroup's block, pass B's in registers
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    double2* ba = reinterpret_cast<double2*>(blkA);
#pragma unroll
                    for (int m = 0; m < SL::K; ++m) {
                        const int u = tid + m * P::NT;
#pragma unroll
                        for (int q = 0; q < SL::R; q += 2) {
                            const long at = (long)((m * SL::R + q) / 2) * P::NT + tid;
                            const double2 va = ba[at];
                            ba[at] = make_double2(0.0, 0.0);
                            if (SL::TASKS % P::NT == 0 || u < SL::TASKS) {
                                lds[sw(u * SL::R + q)] = cd{va.x, accB[m][q]};
                                lds[sw(u * SL::R + q + 1)] = cd{va.y, accB[m][q + 1]};
                            }
                            accB[m][q] = accB[m][q + 1] = 0.0;
                        }
                    }
                } else {
                    static_for<SL::K>([&](auto mm) {
                        constexpr int m = decltype(mm)::value;
                        const int u = tid + m * P::NT;
                        if (SL::TASKS % P::NT == 0 || u < SL::TASKS) {
                            static_for<SL::R>([&](auto qq) {
                                constexpr int q = decltype(qq)::value;
                                constexpr int a = agpr_acc_base<P>() + 2 * (m * SL::R + q);
                                lds[sw(u * SL::R + q)] = cd{ag_read_f64<a>(), accB[m][q]};
                                ag_write_f64<a>(0.0);
                                accB[m][q] = 0.0;
                            });
                        }
                    });
                }
                agpr_fence<P>();
                __syncthreads();
                inv_all_stages<P, P::S - 1>(lds, tw2, tid);
                // lag values of this atom; the atom mean is a row sum over by_particle afterwards
                // (k_row_sums), like the reference's mean(axis=1)
                for (int n0 = tid; n0 < T; n0 += 4 * P::NT) {
                    agpr_fence<P>();
                    double val[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int n = n0 + k * P::NT;
                        val[k] = lag_value<P>(lds, tw2, n < T ? n : 0, T);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int n = n0 + k * P::NT;
                        // atom-major scratch (ld_bp = row pitch >= T): a wave stores 512 contiguous
                        // bytes; k_bp_transpose turns it into the caller's (n_frames, n_atoms) array
                        if (n < T) by_particle[(long)atom * ld_bp + n] = val[k];
                    }
                }
                agpr_fence<P>();
                __syncthreads();
            }
        }
        unit = next;
    }
    if constexpr (BYP) return;
    // write both accumulator sets to this workgroup's block ([quad][thread] layout)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        if constexpr (!landing<P>()) acc_swap_out<P>(blkA, tid);
        double2* b = reinterpret_cast<double2*>(blkB);
#pragma unroll
        for (int m = 0; m < SL::K; ++m)
#pragma unroll
            for (int q = 0; q < SL::R; q += 2)
                b[(long)((m * SL::R + q) / 2) * P::NT + tid] = make_double2(accB[m][q], accB[m][q + 1]);
    }
    if constexpr (STAMP) {
        if (threadIdx.x == 0)
            for (int i = 0; i < 8; ++i) stamps[8 * (long)wg + i] = st_acc[i];
    }
}

// ---- K3 (timeseries path): one inverse transform of the summed spectrum --------
// spec: [n_slices][2][M] (pass A bins, pass B bins; the slices are partial sums over disjoint
// sets of workgroups), digit-reversed order.
template <class P>
__global__ void __launch_bounds__(P::NT)
    k_fft_finalize(const double* __restrict__ spec, const cd* __restrict__ tw2, int T,
                   double* __restrict__ lagsum, int n_slices) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    const int tid = threadIdx.x;
    // 4 elements x n_slices x 2 independent loads in flight per thread
    for (int i0 = tid; i0 < P::M; i0 += 4 * P::NT) {
        double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
        for (int y = 0; y < n_slices; ++y) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * P::NT;
                if (i < P::M) {
                    a[k] += spec[(long)y * 2 * P::M + i];
                    b[k] += spec[(long)y * 2 * P::M + P::M + i];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * P::NT;
            if (i < P::M) lds[sw(i)] = cd{a[k], b[k]};
        }
    }
    __syncthreads();
    inv_all_stages<P, P::S - 1>(lds, tw2, tid);
    for (int n = tid; n < T; n += P::NT) lagsum[n] = lag_value<P>(lds, tw2, n, T);
}

}  // namespace ta
