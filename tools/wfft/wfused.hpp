// wfused.hpp (tools/wfft: an EXPERIMENT, not part of the library; outcome in profiles/r06_bp_fused.txt) — FFT VACF by particle in ONE kernel (gfx950): an atom's forward transforms, its power
// spectrum and the inverse transform never leave the compute unit.
//
// Replaces, for results.vacf_by_particle of VelocityAutocorr._conclude_fft
// (/root/reference/transport_analysis/velocityautocorr.py:145-147, 208-215), the chain
// k_wsplit_accum (by-particle mode) -> per-atom power spectra in HBM -> k_winverse of wfft.hpp: those two
// kernels moved 2 L doubles per atom through HBM (16 GB written and read back at 10000 x 100000 x 3).
//
// Same transform as wfft.hpp with the padded length L = 2 R0 512 cut FOUR ways instead of two:
//     bin k = 4 s + c,   Z[4 s + c] = FFT_M(u_c)[s],   M = RH 512,  RH = R0 / 2,
//     u_c[t] = W_L^{c t} (z[t] + W_4^c z[t + M]),   t < M,   c < 4
// (the outer radix R = 2 of wfft.hpp).  ONE workgroup runs all four passes of a unit from ONE read of its
// rows: thread u holds rows u + 512 jj, jj < R0 (a_j = row j, b_j = row j + RH), and the LDS -- R0
// sub-series of 512 points, as in plan R0 -- holds TWO passes at a time:
//     half h: S1 passes 2h, 2h+1 (radix-RH butterflies of a_j + W_4^c b_j) -> slots [pass & 1][q]
//             S2 the R0 slots, one wave each, |.|^2 into acc[h] (wfft.hpp's second stage as it stands)
// so every input byte is requested once, by one workgroup, and all four passes' accumulators of the atom
// stay in registers.  After the atom's last unit the accumulators change hands through the LDS (the wave
// that owns pass 2h's sub-series q also needs pass 2h+1's), and the two complex M-point transforms
// Q_h = FFT_M(P_2h + i P_2h+1) run as in k_winverse (transposed algorithm, untangled against the mirrored
// values); thread u then owns the lags u + 512 j + M jo, sums the four passes' terms in registers and
// writes the atom's row of the atom-major scratch that k_bp_transpose turns into (n_frames, n_atoms).
#pragma once
#ifndef WF_ABLATIONS
#include "../../transport_analysis_amd/csrc/wfft.hpp"
#endif

namespace ta {

// W_{2 R0}^m, any m >= 0, from the table of pass B's twist (j < R0; the other half by its sign)
template <int R0>
__host__ __device__ constexpr cd wf_twist_any(int m) {
    m %= 2 * R0;
    if (m >= R0) return cd{-WfTwist<R0>::re(m - R0), -WfTwist<R0>::im(m - R0)};
    return cd{WfTwist<R0>::re(m), WfTwist<R0>::im(m)};
}
// a W_4^n
template <int N>
__device__ __forceinline__ cd wf_mul_w4(cd a) {
    if constexpr ((N & 3) == 0) return a;
    else if constexpr ((N & 3) == 1) return mul_mi(a);
    else if constexpr ((N & 3) == 2) return cd{-a.x, -a.y};
    else return mul_pi(a);
}

// rnorm[n] = 1 / (L (T - n)): the lag normalisation of the fused kernel as a table (L (T - n) < 2^53: exact product,
// one rounding in the division; the kernel multiplies: within one ulp of the quotient, as k_winverse's hoisted form)
inline void wfused_fill_rnorm(int R0, long T, double* a) {
    const double L = 2.0 * R0 * 512.0;
    for (long n = 0; n < T; ++n) a[n] = 1.0 / (L * (double)(T - n));
}

#ifndef WFU_SINGLE_H0
#define WFU_SINGLE_H0 0  // bit 0 / bit 1: float64 / float32 rows: the first half's second stage (rows alive) runs one sub-series at a time
#endif
#ifndef WFU_TW_RESIDENT
#define WFU_TW_RESIDENT 0  // 1: the second stage's tangent-form constants stay in registers (32) for the whole launch
#endif
#ifndef WFU_ABL
#define WFU_ABL 0  // timing / register-pressure ablations (wrong results): 1 no inverse, 2 no S2 of half 0, 4 no S2 of half 1, 8 no S1
#endif
#ifndef WFU_STAMP
#define WFU_STAMP 0  // diagnostic builds (tools/wfft): shader cycles per phase, summed over workgroups (wave 0) into wfu_stamps
#endif
#if WFU_STAMP
__device__ unsigned long long wfu_stamps[16];
#define WFU_T(i)                                                      \
    {                                                                 \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        wst[i] += now_ - wprev;                                       \
        wprev = now_;                                                 \
    }
#else
#define WFU_T(i)
#endif
#ifndef WFU_XV
#define WFU_XV 0  // 1: one sub-series in flight takes its first exchange through the register file (32 more registers)
#endif
#ifndef WFU_X3
#define WFU_X3 0  // 1: the waves that own three slots run them three at a time (96 registers in flight), else two and one
#endif

template <class P, bool SRC32>
__global__ void __launch_bounds__(P::NT, 1)
    k_wfused_bp(const double* __restrict__ pm, long pitch, int T, long n_atoms, int D,
                const cd* __restrict__ tw2, const double* __restrict__ rnorm, double* __restrict__ out, long ld) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    double* ldsd = reinterpret_cast<double*>(smem_raw);
    constexpr int R0 = P::R0, RH = R0 / 2, N1 = P::N1, NW = P::NW, NS1 = P::NS1, NT = P::NT, K1 = P::K1;
    static_assert(R0 % 2 == 0 && R0 >= 4, "the four-pass form needs an even number of sub-series slots");
    using PH = WPlan<RH>;  // (its real-column selection: the mirror structure of RH sub-series per pass)
    constexpr int M = RH * N1, L = 4 * M;
    const int tid = threadIdx.x, wave = tid >> 6;
    int lane = tid & 63;
    const int upa = D == 3 ? 2 : 1;   // units per atom
    const int grp = D & 1 ? 2 : 1;    // atoms 2i, 2i + 1 share a column pair when an atom has an odd number of columns
    const long n_groups = (n_atoms + grp - 1) / grp;

    const __amdgpu_buffer_rsrc_t twr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<cd*>(tw2), 0, (L + kWfStageRows * 64) * 16, 0x00020000);
    // rnorm[n] = 1 / (L (n_frames - n)), n < n_frames (wfused_fill_rnorm; past the end: zeros from the bounds check)
    const __amdgpu_buffer_rsrc_t rnr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(rnorm), 0, T * 8, 0x00020000);
    auto unit_rsrc = [&](long atom, int k, int* kind) {
        long pair = 0;
        const bool live = atom < n_atoms;
        *kind = 2;
        wf_unit_of(live ? atom : 0, k, D, &pair, kind);
        if constexpr (SRC32)
            return __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(reinterpret_cast<const float*>(pm) + (live ? pair : 0) * pitch * 2), 0,
                live ? T * 8 : 0, 0x00020000);
        else
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pm + (live ? pair : 0) * pitch * 2), 0,
                                                     live ? T * 16 : 0, 0x00020000);
    };
    using RowT = std::conditional_t<SRC32, wf_u32x2, cd>;
    auto load_row = [&](__amdgpu_buffer_rsrc_t rs, int kd, int u, unsigned row_off) -> RowT {
        if constexpr (SRC32) {
            return __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)u * 8u, row_off * 8u, 0);
        } else {
            if (kd == 2) return wf_load(rs, (unsigned)u * 16u, row_off * 16u);
            return cd{__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                                                     rs, (unsigned)u * 16u + (unsigned)kd * 8u, row_off * 16u, 0)),
                      0.0};
        }
    };
    auto row_value = [&](RowT r, int kd) -> cd {
        if constexpr (SRC32) {
            if (kd == 2)
                return cd{(double)__builtin_bit_cast(float, (unsigned)r[0]), (double)__builtin_bit_cast(float, (unsigned)r[1])};
            return cd{(double)__builtin_bit_cast(float, (unsigned)(kd ? r[1] : r[0])), 0.0};
        } else {
            return r;
        }
    };
    RowT xx[K1][R0];
    auto issue_loads = [&](__amdgpu_buffer_rsrc_t rs, int kd) {
#pragma unroll
        for (int k1 = 0; k1 < K1; ++k1) {
            const int u = tid + NT * k1;
            if (K1 * NT != N1 && u >= N1) continue;
            if (kd == 2) {
#pragma unroll
                for (int j = 0; j < R0; ++j) xx[k1][j] = load_row(rs, 2, u, (unsigned)(N1 * j));
            } else {
#pragma unroll
                for (int j = 0; j < R0; ++j) xx[k1][j] = load_row(rs, kd, u, (unsigned)(N1 * j));
            }
        }
    };
    WfTw stw;
    auto load_stage_tw = [&]() {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            stw.b[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(L + (14 + a) * 64) * 16u);
            stw.c[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(L + (18 + a) * 64) * 16u);
        }
    };
#if WFU_STAMP
    unsigned long long wst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, wprev = __builtin_amdgcn_s_memtime();
#endif
    double acc[2][NS1][8];
    auto zero_acc = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < NS1; ++s)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[h][s][c] = 0.0;
    };

    // ---- S1 of half H: passes c = 2H, 2H + 1 of the unit whose rows are in xx ----------------------------
    auto first_stage = [&](auto hc, int kind) {
        constexpr int H = decltype(hc)::value;
        if constexpr (SRC32 && H == 1) {
            // the raw rows again, not the first half's widened copies of them (kept alive across its S2 they cost 80 registers)
#pragma unroll
            for (int k1 = 0; k1 < K1; ++k1)
#pragma unroll
                for (int j = 0; j < R0; ++j) asm volatile("" : "+v"(xx[k1][j]));
        }
#pragma unroll
        for (int k1 = 0; k1 < K1; ++k1) {
            int u = tid + NT * k1;
            if (K1 * NT != N1 && u >= N1) continue;
            // (per-thread table entries and LDS addresses are re-formed per call: hoisted out of the loop over units they
            // would be spilled)
            asm volatile("" : "+v"(u));
            const cd g = wf_load(twr, (unsigned)u * 64u, 0u), g2 = wf_load(twr, (unsigned)u * 128u, 0u);  // W_M^u, W_M^2u
            static_for_range<0, 2>([&](auto ppc) {
                constexpr int pp = decltype(ppc)::value, c = 2 * H + pp;
                cd x[RH];
#pragma unroll
                for (int j = 0; j < RH; ++j) {
                    const cd a = row_value(xx[k1][j], kind), b = row_value(xx[k1][j + RH], kind);
                    cd v = a + wf_mul_w4<c>(b);
                    if (c != 0 && j != 0) v = cmul(v, wf_twist_any<R0>(c * j));
                    x[j] = v;
                }
                Dft<RH>::run(x);
                // output twiddles W_L^{u (4 q + c)} = h_c g^q, two chains (even / odd q) by g^2
                cd te = cd{1.0, 0.0}, to = g;
                if constexpr (c != 0) {
                    const cd h = wf_load(twr, (unsigned)(u * c) * 16u, 0u);
                    te = h;
                    to = cmul(h, g);
                    x[0] = cmul(x[0], te);
                }
                cd* dst = lds + pp * RH * N1 + u;
                dst[0] = x[0];
                if constexpr (RH > 1) {
                    x[1] = cmul(x[1], to);
                    dst[N1] = x[1];
                }
#pragma unroll
                for (int q = 2; q < RH; ++q) {
                    if (q & 1) {
                        to = cmul(to, g2);
                        x[q] = cmul(x[q], to);
                    } else {
                        te = cmul(te, g2);
                        x[q] = cmul(x[q], te);
                    }
                    dst[q * N1] = x[q];
                }
                __builtin_amdgcn_sched_barrier(0);  // one pass at a time: interleaved, their work arrays double the pressure
            });
        }
    };

    // ---- S2 of half H over the R0 slots (wfft.hpp's second stage) ---------------------------------------
    auto second_stage = [&](auto hc, int kind) {
        constexpr int H = decltype(hc)::value;
        auto& a = acc[H];
        // the lane's LDS addresses (and, unless resident, the stage constants) are formed inside the call: hoisted out of
        // the loop over units they are spilled and reloaded one by one
        asm volatile("" : "+v"(lane));
        WfAddr wad;
        wad.init(lane, (unsigned)P::sub_base(wave) * kWfSubBytes);
        const WfSub wsub(wad, smem_raw);
        if constexpr (!WFU_TW_RESIDENT) load_stage_tw();
        if (!(WFU_ABL & 16) && kind != 2) {
            // a real column: of every mirror pair of sub-series one is transformed, with weight 2 (WPlan::kRealSel)
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            const unsigned m0 = PH::kRealSel.qmask[H ? 1 : 0], m1 = PH::kRealSel.qmask[1];
            const unsigned o0 = PH::kRealSel.q1mask[H ? 1 : 0], o1 = PH::kRealSel.q1mask[1];
            const unsigned qs = (m0 | (m1 << RH)) >> P::sub_base(wv);   // bit s: slot sub_base + s is transformed
            const unsigned q1 = (o0 | (o1 << RH)) >> P::sub_base(wv);  // ... with weight 1
            static_for_range<0, NS1>([&](auto ss) {
                constexpr int s = decltype(ss)::value;
                if ((s < P::NLO || wv < P::REM) && (qs >> s & 1u))
                    wf_sub512_w<s, (WFU_XV && P::kRegExchangeSingle)>(wsub, stw, a[s], (q1 >> s & 1u) ? 1.0 : 2.0);
            });
            return;
        }
        constexpr bool kSingles = ((WFU_SINGLE_H0 >> (SRC32 ? 1 : 0)) & 1) && H == 0;
        if constexpr (!kSingles && WFU_X3 && NS1 == 3 && P::NLO == 2 && P::REM != 0) {
            if (wave < P::REM) wf_sub512_x3<(WFU_XV && P::kRegExchange)>(wsub, stw, a[0], a[1], a[2]);
            else wf_sub512_x2<0, (WFU_XV && P::kRegExchange)>(wsub, stw, a[0], a[1]);
        } else if constexpr (kSingles) {
            static_for_range<0, NS1>([&](auto ss) {
                constexpr int s = decltype(ss)::value;
                if (s < P::NLO || wave < P::REM) wf_sub512<s, false>(wsub, stw, a[s]);
            });
        } else {
            static_for_range<0, NS1>([&](auto ss) {
                constexpr int s = decltype(ss)::value;
                constexpr bool full = s < P::NLO;
                constexpr bool nfull = s + 1 < P::NLO;
                constexpr bool head = full && nfull && (s % 2 == 0);
                constexpr bool tail = full && s > 0 && (s % 2 == 1);
                if constexpr (head) {
                    wf_sub512_x2<s, (WFU_XV && P::kRegExchange)>(wsub, stw, a[s], a[s + 1]);
                } else if constexpr (!tail) {
                    if (full || wave < P::REM) wf_sub512<s, (WFU_XV && P::kRegExchangeSingle)>(wsub, stw, a[s]);
                }
            });
        }
    };

    // ---- the atom's lags from its four passes' accumulators ---------------------------------------------
    // nrs / nkind: the rows of the next unit (the next atom's first) are requested in the middle of it
    auto inverse = [&](long atom, __amdgpu_buffer_rsrc_t nrs, int nkind) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        asm volatile("" : "+v"(lane));
        // (a) accumulators -> LDS as doubles [h][slot][cc][lane] (every wave is past its S2: the barrier behind it)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < NS1; ++s)
                if (s < P::NLO || wv < P::REM) {
                    const int slot = P::sub_base(wv) + s;
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) ldsd[((h * R0 + slot) * 8 + cc) * 64 + lane] = acc[h][s][cc];
                }
        __syncthreads();
        WFU_T(4)
        // (b) complex sub-series i = h RH + q, i < R0, consecutive ones per wave: P_2h + i P_2h+1
        cd v[NS1][8];
#pragma unroll
        for (int s = 0; s < NS1; ++s)
            if (s < P::NLO || wv < P::REM) {
                const int i = P::sub_base(wv) + s, h = i >= RH ? 1 : 0, q = i - h * RH;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc)
                    v[s][cc] = cd{ldsd[((h * R0 + q) * 8 + cc) * 64 + lane], ldsd[((h * R0 + RH + q) * 8 + cc) * 64 + lane]};
            }
        __syncthreads();
        {
            cd twa[7], twb[7];
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                twa[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(L + a * 64) * 16u);
                twb[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(L + (7 + a) * 64) * 16u);
            }
            const WfSubT wt(lane);
#pragma unroll
            for (int s = 0; s < NS1; ++s)
                if (s < P::NLO || wv < P::REM) wt.run(lds + (P::sub_base(wv) + s) * N1, v[s], twa, twb);
        }
        WFU_T(5)
        __syncthreads();
        WFU_T(6)
        __builtin_amdgcn_sched_barrier(0);
        issue_loads(nrs, nkind);  // the next atom's first rows land during the rest of this one
        __builtin_amdgcn_sched_barrier(0);
        // (c) thread u: Q_h[u + 512 j'] = DFT_RH over q of G_q[u] W_M^{u q}, back into column u of the same blocks
        double* o = out + atom * ld;
#pragma unroll
        for (int k1 = 0; k1 < K1; ++k1) {
            int u = tid + NT * k1;
            if (K1 * NT != N1 && u >= N1) continue;
            asm volatile("" : "+v"(u));
            const cd g = wf_load(twr, (unsigned)u * 64u, 0u), g2 = wf_load(twr, (unsigned)u * 128u, 0u);
            cd x[2][RH];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int q = 0; q < RH; ++q) x[h][q] = lds[(h * RH + q) * N1 + u];
                cd te = cd{1.0, 0.0}, to = g;
                if constexpr (RH > 1) x[h][1] = cmul(x[h][1], to);
#pragma unroll
                for (int q = 2; q < RH; ++q) {
                    if (q & 1) {
                        to = cmul(to, g2);
                        x[h][q] = cmul(x[h][q], to);
                    } else {
                        te = cmul(te, g2);
                        x[h][q] = cmul(x[h][q], te);
                    }
                }
                Dft<RH>::run(x[h]);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < RH; ++j) lds[(h * RH + j) * N1 + u] = x[h][j];
            WFU_T(7)
            if constexpr (K1 == 1) __syncthreads();
            else {
                // (several butterflies per thread: every column is written before any mirror is read)
                if (k1 == K1 - 1) __syncthreads();
                continue;
            }
            // (d) untangle against the mirrored values, the four passes' terms of lag n = u + 512 j + M jo
            const int mu = u == 0 ? 0 : N1 - u;
            const cd h1 = wf_load(twr, (unsigned)u * 16u, 0u), h2 = wf_load(twr, (unsigned)u * 32u, 0u),
                     h3 = wf_load(twr, (unsigned)u * 48u, 0u);
            double rn[RH][2];
#pragma unroll
            for (int j = 0; j < RH; ++j)
#pragma unroll
                for (int jo = 0; jo < 2; ++jo)
                    rn[j][jo] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rnr, (unsigned)u * 8u, (unsigned)(N1 * j + M * jo) * 8u, 0));
#pragma unroll
            for (int j = 0; j < RH; ++j) {
                const int jm = u == 0 ? (RH - j) % RH : RH - 1 - j;
                double lagv[2] = {0.0, 0.0};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const cd qm = lds[(h * RH + jm) * N1 + mu];
                    const cd xq = x[h][j];
                    const cd qa = cd{0.5 * (xq.x + qm.x), 0.5 * (xq.y - qm.y)};    // Q of pass 2h
                    const cd qb = cd{0.5 * (xq.y + qm.y), -0.5 * (xq.x - qm.x)};  // Q of pass 2h + 1
                    // W_L^{c n} = W_L^{c u} W_{2 R0}^{c j} W_4^{c jo}
                    const cd wa = h == 0 ? cd{1.0, 0.0} : cmul(h2, wf_twist_any<R0>(2 * j));
                    const cd wb = h == 0 ? cmul(h1, wf_twist_any<R0>(j)) : cmul(h3, wf_twist_any<R0>(3 * j));
                    const cd ta = cmul(wa, qa), tb = cmul(wb, qb);
                    // jo = 0: W_4^0; jo = 1: W_4^c = 1, -i, -1, +i for c = 0..3
                    if (h == 0) {
                        lagv[0] += ta.x + tb.x;
                        lagv[1] += ta.x + tb.y;  // Re(-i tb) = tb.y
                    } else {
                        lagv[0] += ta.x + tb.x;
                        lagv[1] += -ta.x - tb.y;  // Re(-ta) + Re(+i tb)
                    }
                }
#pragma unroll
                for (int jo = 0; jo < 2; ++jo) {
                    const int n = u + N1 * j + M * jo;
                    if (n < T) o[n] = lagv[jo] * rn[j][jo];  // 1 / (L (T - n))
                }
            }
        }
        if constexpr (K1 > 1) {
            // several butterflies per thread: the untangling in a second sweep (Q re-read from the LDS)
#pragma unroll
            for (int k1 = 0; k1 < K1; ++k1) {
                int u = tid + NT * k1;
                if (K1 * NT != N1 && u >= N1) continue;
                asm volatile("" : "+v"(u));
                const int mu = u == 0 ? 0 : N1 - u;
                const cd h1 = wf_load(twr, (unsigned)u * 16u, 0u), h2 = wf_load(twr, (unsigned)u * 32u, 0u),
                         h3 = wf_load(twr, (unsigned)u * 48u, 0u);
                double rn[RH][2];
#pragma unroll
                for (int j = 0; j < RH; ++j)
#pragma unroll
                    for (int jo = 0; jo < 2; ++jo)
                        rn[j][jo] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rnr, (unsigned)u * 8u, (unsigned)(N1 * j + M * jo) * 8u, 0));
#pragma unroll
                for (int j = 0; j < RH; ++j) {
                    const int jm = u == 0 ? (RH - j) % RH : RH - 1 - j;
                    double lagv[2] = {0.0, 0.0};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const cd qm = lds[(h * RH + jm) * N1 + mu];
                        const cd xq = lds[(h * RH + j) * N1 + u];
                        const cd qa = cd{0.5 * (xq.x + qm.x), 0.5 * (xq.y - qm.y)};
                        const cd qb = cd{0.5 * (xq.y + qm.y), -0.5 * (xq.x - qm.x)};
                        const cd wa = h == 0 ? cd{1.0, 0.0} : cmul(h2, wf_twist_any<R0>(2 * j));
                        const cd wb = h == 0 ? cmul(h1, wf_twist_any<R0>(j)) : cmul(h3, wf_twist_any<R0>(3 * j));
                        const cd ta = cmul(wa, qa), tb = cmul(wb, qb);
                        if (h == 0) {
                            lagv[0] += ta.x + tb.x;
                            lagv[1] += ta.x + tb.y;
                        } else {
                            lagv[0] += ta.x + tb.x;
                            lagv[1] += -ta.x - tb.y;
                        }
                    }
#pragma unroll
                    for (int jo = 0; jo < 2; ++jo) {
                        const int n = u + N1 * j + M * jo;
                        if (n < T) o[n] = lagv[jo] * rn[j][jo];
                    }
                }
            }
        }
        WFU_T(8)
        __syncthreads();  // Q consumed before the next atom's first stage overwrites the LDS
        WFU_T(9)
    };

    // ---- the workgroup's groups of atoms ------------------------------------------------------------------
    long g0 = blockIdx.x;
    if (g0 >= n_groups) return;
    int kind = 2, nkind = 2;
    long atom = g0 * grp;
    int k = 0;
    __amdgpu_buffer_rsrc_t crs = unit_rsrc(atom, 0, &kind);
    issue_loads(crs, kind);
    if constexpr (WFU_TW_RESIDENT) load_stage_tw();
    zero_acc();
    for (;;) {
        // the unit behind this one: the atom's next unit, the group's second atom, the next group's first
        const bool last_of_atom = k == upa - 1;
        long natom = atom;
        int nk = k + 1;
        if (last_of_atom) {
            nk = 0;
            if (grp == 2 && (atom & 1) == 0 && atom + 1 < n_atoms) natom = atom + 1;
            else {
                g0 += gridDim.x;
                natom = g0 < n_groups ? g0 * grp : n_atoms;  // (past the end: an empty resource, zeros)
            }
        }
        const __amdgpu_buffer_rsrc_t nrs = unit_rsrc(natom, nk, &nkind);
        if constexpr (!(WFU_ABL & 8)) first_stage(std::integral_constant<int, 0>{}, kind);
        WFU_T(0)
        __syncthreads();
        WFU_T(1)
        if constexpr (!(WFU_ABL & 2)) second_stage(std::integral_constant<int, 0>{}, kind);
        WFU_T(2)
        __syncthreads();
        WFU_T(3)
        if constexpr (!(WFU_ABL & 8)) first_stage(std::integral_constant<int, 1>{}, kind);
        WFU_T(0)
        __syncthreads();
        WFU_T(1)
        if constexpr (!(WFU_ABL & 4)) second_stage(std::integral_constant<int, 1>{}, kind);
        __builtin_amdgcn_sched_barrier(0);
        if (!last_of_atom) issue_loads(nrs, nkind);
        WFU_T(2)
        __syncthreads();
        WFU_T(3)
        if (last_of_atom) {
            if constexpr (!(WFU_ABL & 1)) inverse(atom, nrs, nkind);
            else {
                issue_loads(nrs, nkind);
                double z = 0.0;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int s_ = 0; s_ < NS1; ++s_)
#pragma unroll
                        for (int c = 0; c < 8; ++c) z += acc[h][s_][c];
                out[atom * ld + tid] = z;
            }
            zero_acc();
        }
        if (natom >= n_atoms) break;
        atom = natom, k = nk, kind = nkind, crs = nrs;
    }
#if WFU_STAMP
    if (tid == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(&wfu_stamps[i], wst[i]);
#endif
}

}  // namespace ta
