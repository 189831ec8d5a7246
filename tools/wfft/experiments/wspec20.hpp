// EXPERIMENT (round 3, tools only): wave-specialised forward kernel for the plan R0 = 20, lag sums,
// no outer radix.  Included by tools/wfft/wfft_test.hip after csrc/wfft.hpp; run with WF_SPEC=1.
//
// The radix-20 first stage splits into its even and its odd outputs (two radix-10 halves):
//     X[2m]     = DFT10_m( x_j + x_{j+10} )
//     X[2m + 1] = DFT10_m( (x_j - x_{j+10}) W_20^j )                      (pass A)
// and with pass B's twist t_j = W_40^j on the rows (t_{j+10} = -i t_j):
//     X[2m]     = DFT10_m( (x_j - i x_{j+10}) W_40^j ),   X[2m + 1] = DFT10_m( (x_j + i x_{j+10}) W_40^{3j} ).
// A half is 10 sub-series = 80 KiB of LDS, so both halves of the LDS hold different half-units:
// waves 0..3 ("producers", 256 threads, two butterflies each) compute half-unit h + 1 in
// registers and store it while waves 4..7 ("consumers") run the ten sub-series of half-unit h;
// one barrier per half-unit; every SIMD hosts one producer and one consumer wave.  The producers
// read a unit's rows twice (once per half: the second time from L2).
//
// RESULT (correct: harness check passes; same box, 30000 pairs): 2.93 ms against 2.13 ms for the
// library kernel.  Stamps per unit and pass: producers 14.7k cycles of work + 12.4k waiting at the
// barrier, consumers 26.5k of work.  Four consumer waves are too few: a wave's sub-series is a
// chain of dependent LDS round trips (read, exchange, exchange) with at most 15 LDS operations
// in flight per wave (lgkmcnt is four bits), so S2's rate scales with the number of waves that run
// it -- eight in the library kernel, four here -- and neither pipe is busy: LDS ~26 %, vector
// ~37 % per SIMD.  What S2 wants is MORE waves in flight, not fewer with a partner doing S1.
#pragma once

namespace ta {

// exp(-i pi k / 20), any k >= 0 (literals)
template <int K>
__device__ __forceinline__ cd wspec_w40() {
    constexpr int k = K % 40;
    if constexpr (k < 20) return cd{WfTwist<20>::re(k), WfTwist<20>::im(k)};
    else return cd{-WfTwist<20>::re(k - 20), -WfTwist<20>::im(k - 20)};
}

template <bool STAMP = false>
__global__ void __launch_bounds__(512, 1)
    k_wspec20(const double* __restrict__ pm, long pitch, int T, long n_units, const cd* __restrict__ tw2,
              double* __restrict__ accg, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd* lds = reinterpret_cast<cd*>(smem_raw);
    constexpr int R0 = 20, N1 = 512, M = R0 * N1, L = 2 * M, HQ = 10;
    constexpr unsigned kRegion = HQ * kWfSubBytes;  // 80 KiB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bi = blockIdx.x >> 3;
    const int pass = bi & 1;
    const long tuple = (blockIdx.x & 7) + 8 * (bi >> 1), n_tuples = gridDim.x / 2;
    const __amdgpu_buffer_rsrc_t twr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<cd*>(tw2), 0, (L + kWfStageRows * 64) * 16, 0x00020000);
    // the tuple's units: pairs tuple, tuple + n_tuples, ...; n_mine of them; 2 n_mine half-units
    const long n_mine = tuple < n_units ? (n_units - tuple + n_tuples - 1) / n_tuples : 0;
    const long n_half = 2 * n_mine;
    auto unit_rsrc = [&](long i) {  // i-th unit of the tuple; past the end: an empty buffer (zeros)
        const bool live = i < n_mine;
        const long pair = tuple + i * n_tuples;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pm + (live ? pair : 0) * pitch * 2), 0,
                                                 live ? T * 16 : 0, 0x00020000);
    };
    unsigned long long st_acc[2] = {0, 0}, st_prev = 0;
    if constexpr (STAMP) st_prev = __builtin_amdgcn_s_memtime();

    if (wave < 4) {
        // ------------------------------------------------------------------ producers
        // thread p: butterflies u = p and p + 256; half-unit hh = 2 i + h goes to region hh & 1
        const int p = tid;
        cd xx[2][R0];
        cd g2[2], s0[2][2];  // per butterfly: g^2, and the first output twiddle of each half (h g^half)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int u = p + 256 * b;
            const cd g = wf_load(twr, (unsigned)u * 32u, 0u);
            g2[b] = wf_load(twr, (unsigned)u * 64u, 0u);
            const cd h = wf_load(twr, (unsigned)(u * pass) * 16u, 0u);
            s0[b][0] = pass ? h : cd{1.0, 0.0};
            s0[b][1] = pass ? cmul(h, g) : g;
        }
        auto issue = [&](int b, long hh) {
            const __amdgpu_buffer_rsrc_t rs = unit_rsrc(hh >> 1);
            const int u = p + 256 * b;
#pragma unroll
            for (int j = 0; j < R0; ++j) xx[b][j] = wf_load(rs, (unsigned)u * 16u, (unsigned)(N1 * j) * 16u);
        };
        auto produce = [&](int b, int half, unsigned region) {
            cd(&x)[R0] = xx[b];
            cd(&y)[10] = reinterpret_cast<cd(&)[10]>(xx[b]);  // in place: y[j] from x[j], x[j + 10]
            static_for_range<0, 10>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                const cd a = x[j], c = x[j + 10];
                cd r;
                if (!pass) {
                    r = half ? a - c : a + c;
                    if (half && j) r = cmul(r, cd{WfTwist<10>::re(j), WfTwist<10>::im(j)});  // W_20^j
                } else {
                    // a -/+ i c, then W_40^j (even half) or W_40^{3j} (odd half)
                    r = half ? cd{a.x - c.y, a.y + c.x} : cd{a.x + c.y, a.y - c.x};
                    if (j) r = cmul(r, half ? wspec_w40<3 * j>() : wspec_w40<j>());
                }
                y[j] = r;
            });
            Dft<10>::run(y);
            const int u = p + 256 * b;
            cd t = s0[b][half];
            cd* dst = lds + region / sizeof(cd) + u;
            if (pass || half) y[0] = cmul(y[0], t);
            dst[0] = y[0];
#pragma unroll
            for (int m = 1; m < 10; ++m) {
                t = cmul(t, g2[b]);
                y[m] = cmul(y[m], t);
                dst[m * N1] = y[m];
            }
        };
        if (n_half > 0) {
            issue(0, 0);
            issue(1, 0);
            // half-unit 0 into region 0 (the consumers wait)
            produce(0, 0, 0u);
            issue(0, 1);
            produce(1, 0, 0u);
            issue(1, 1);
        }
        __syncthreads();
        for (long hh = 0; hh < n_half; ++hh) {
            // consumers: half-unit hh in region hh & 1; producers: half-unit hh + 1 into the other
            const unsigned region = ((hh + 1) & 1) ? kRegion : 0u;
            const int half = (int)((hh + 1) & 1);
            produce(0, half, region);  // (past the end: zeros from an empty buffer, never read)
            __builtin_amdgcn_sched_barrier(0);
            issue(0, hh + 2);
            produce(1, half, region);
            __builtin_amdgcn_sched_barrier(0);
            issue(1, hh + 2);
            if constexpr (STAMP) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                st_acc[0] += now - st_prev;
                st_prev = now;
            }
            __syncthreads();
            if constexpr (STAMP) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                st_acc[1] += now - st_prev;
                st_prev = now;
            }
        }
    } else {
        // ------------------------------------------------------------------ consumers
        // consumer wave cw: sub-series qi of a half: cw 0: 0..2, 1: 3..5, 2: 6..7, 3: 8..9
        const int cw = wave - 4;
        const int qb = cw < 2 ? 3 * cw : 6 + 2 * (cw - 2);
        const bool three = cw < 2;
        WfTw stw;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            stw.b[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(L + (14 + a) * 64) * 16u);
            stw.c[a] = wf_load(twr, (unsigned)lane * 16u, (unsigned)(L + (18 + a) * 64) * 16u);
        }
        double acc[2][3][8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[h][s][c] = 0.0;
        __syncthreads();
        for (long hh = 0; hh < n_half; ++hh) {
            auto run_half = [&](auto hc) {
                constexpr int h = decltype(hc)::value;
                WfAddr wad;
                wad.init(lane, (h ? kRegion : 0u) + (unsigned)qb * kWfSubBytes);
                const WfSub wsub(wad, smem_raw);
                wf_sub512_x2<0, true>(wsub, stw, acc[h][0], acc[h][1]);
                __builtin_amdgcn_sched_barrier(0);
                if (three) wf_sub512<2, true>(wsub, stw, acc[h][2]);
            };
            if (hh & 1) run_half(std::integral_constant<int, 1>{});
            else run_half(std::integral_constant<int, 0>{});
            if constexpr (STAMP) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                st_acc[0] += now - st_prev;
                st_prev = now;
            }
            __syncthreads();
            if constexpr (STAMP) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                st_acc[1] += now - st_prev;
                st_prev = now;
            }
        }
        // accg[tuple][pass][q][cc / 2][lane][cc & 1], q = 2 qi + half
        const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
            accg + (tuple * 2 + pass) * (long)M, 0, M * 8, 0x00020000);
        const int cwu = __builtin_amdgcn_readfirstlane(cw);
        const int qbu = cwu < 2 ? 3 * cwu : 6 + 2 * (cwu - 2);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (s == 2 && cwu >= 2) continue;
                const int q = 2 * (qbu + s) + h;
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) {
                    const cd two{acc[h][s][2 * c2], acc[h][s][2 * c2 + 1]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wf_u32x4, two), sr, (unsigned)lane * 16u,
                                                           (unsigned)((q * 4 + c2) * 64) * 16u, 0);
                }
            }
    }
    if constexpr (STAMP) {
        if (lane == 0 && (wave == 0 || wave == 4))
            for (int i = 0; i < 2; ++i) stamps[8 * (long)blockIdx.x + (wave ? 4 : 0) + i] = st_acc[i];
    }
}

}  // namespace ta
