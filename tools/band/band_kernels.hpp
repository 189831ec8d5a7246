// band_kernels.hpp — lag SUMS of the O(T^2) correlators on the FP64 matrix cores (gfx950
// v_mfma_f64_16x16x4_f64): the windowed VACF (below) and, further down, the Einstein-Helfand squared
// differences on the product slab.
//
// Quantity: VelocityAutocorr._conclude_simple summed over particles
// (/root/reference/transport_analysis/velocityautocorr.py:217-238):
//   S[k] = sum_{columns c} sum_{i < T-k} v[i, c] v[i+k, c],   lagsum[k] = S[k] / (T - k)
// (the per-particle array of that method needs every particle's own sums: k_direct, direct_kernels.hpp).
//
// With G[t, s] = sum_c v[t, c] v[s, c] (the Gram matrix of the frames), S[k] is the sum of G's k-th
// diagonal: GEMM-shaped work, 2 T^2/2 K flop.  G is never formed.  Time is cut into blocks of 16
// frames; C_d = sum_I G[block I, block I+d] (a 16 x 16 matrix per block lag d) is what an MFMA
// accumulator holds, accumulated over I AND over columns — the contraction runs over (I, c) — and
// only at the very end the 31 diagonals of every C_d are summed into lags 16 d - 15 ... 16 d + 15.
//
// A wave owns 16 consecutive block lags d0 .. d0+15 (16 accumulators, 128 registers) and a range of
// blocks [I0, I1) (a "piece" of the band).  Per step I it needs the fragment F_I (16 frames x 4 columns,
// one float64 per lane) as the A operand and F_{I+d0} .. F_{I+d0+15} as B operands: the B window
// slides by ONE fragment per step, so a step is 2 loads (16 bytes per lane: two columns of a
// pair-major row, i.e. fragments of two k-sets) for 32 MFMAs, straight from the slab through L2
// — no LDS, no barrier in the loop.  Columns go in octets (4 column pairs = one 16-byte load per
// lane); the workgroups with the same blockIdx % n_labels (one XCD, as the hardware deals them)
// sweep the same octets in the same order, their waves covering the whole band between them, so
// an octet (T x 64 bytes) is read by one XCD only.  (Nothing paces the waves, though: measured, 90 % of
// the L2 requests miss and are served by the Infinity Cache / HBM — 2.2 TB/s, no time lost while
// the matrix pipe is the limit)
//
// Output: partial[label][piece][272] (lags 16 d0 - 15 ... 16 d0 + 255 of that piece), every element
// written; k_band_gather adds them up in a fixed order: results do not depend on scheduling.
#pragma once
#include <hip/hip_runtime.h>

#include "../../transport_analysis_amd/csrc/band_common.hpp"

#include <algorithm>
#include <vector>

namespace ta {



struct BandPiece {
    int d0;     // first block lag of the 16 accumulators (a multiple of 16)
    int i0, i1; // blocks [i0, i1) of the A operand
    int phase;  // takes octets label + n_labels * (phase + n_ph * j)
};

// ---- host: cut the band into pieces of equal cost ------------------------------------------
struct BandPlan {
    int T = 0, nblk = 0, n_groups = 0, n_ph = 1, n_labels = 8, slots = 0, per_phase = 0;
    std::vector<BandPiece> pieces;  // [phase][piece within the phase], sorted by group, then i0
    std::vector<int> group_begin;   // [n_groups + 1] into one phase's pieces
    std::vector<int> slot_begin;    // [slots + 1] into slot_pieces
    std::vector<int> slot_pieces;   // piece indices of every wave slot
    double max_cost = 0, mean_cost = 0;  // steps per octet sweep of the busiest / the average slot
};

// slots: wave slots per label (workgroups per label x 8).  Group g = block lags 16 g .. 16 g + 15 has
// nblk - 16 g steps per octet; a visit (one piece on one octet) costs its steps plus kVisit for
// filling the window.  A label keeps n_ph octets in flight (phases), so that a wave slot's share of
// the band is long enough (>= 64 steps where the band allows it) for the visits' overhead not to
// count.  The band, group after group, is one sequence of steps cut into equal shares, one per wave
// slot; a share that crosses a group boundary is two (or more) pieces with their own accumulators.
inline BandPlan band_plan(int T, int slots, int n_labels, int force_ph = 0) {
    constexpr double kVisit = 1.5;
    BandPlan p;
    p.T = T;
    p.nblk = (T + 15) / 16;
    p.n_groups = (p.nblk + 15) / 16;
    p.n_labels = n_labels;
    p.slots = slots;
    auto steps = [&](int g) { return p.nblk - 16 * g; };
    long total = 0;
    for (int g = 0; g < p.n_groups; ++g) total += steps(g);
    // octets in flight per label: their rows (T x 64 bytes each) should stay in one L2 (4 MiB)
    int n_ph = 1;
    while (2 * n_ph <= slots && n_ph < 64 && (double)total * n_ph / slots < 64.0 &&
           (double)T * 64.0 * (2 * n_ph) <= 3.0 * 1048576.0)
        n_ph *= 2;
    if (force_ph > 0) n_ph = force_ph;  // (experiments: tools/band/band_test)
    while (slots % n_ph) n_ph /= 2;
    p.n_ph = n_ph;
    const int wslots = slots / n_ph;  // wave slots of one phase
    // the band as one sequence of steps (group after group), cut into wslots equal shares; a share
    // that crosses a group boundary is two (or more) pieces
    std::vector<BandPiece> one;
    std::vector<std::vector<int>> lists(wslots);
    p.group_begin.assign(p.n_groups + 1, 0);
    {
        int g = 0, i = 0;  // next step to hand out
        for (int w = 0; w < wslots; ++w) {
            long need = (long)((double)total * (w + 1) / wslots + 0.5) - (long)((double)total * w / wslots + 0.5);
            while (need > 0 && g < p.n_groups) {
                const int take = (int)std::min<long>(need, steps(g) - i);
                lists[w].push_back((int)one.size());
                one.push_back({16 * g, i, i + take, 0});
                need -= take, i += take;
                if (i == steps(g)) {
                    ++g, i = 0;
                    if (g <= p.n_groups) p.group_begin[g] = (int)one.size();
                }
            }
        }
        for (; g < p.n_groups; ++g) p.group_begin[g + 1] = (int)one.size();
    }
    p.per_phase = (int)one.size();
    for (int ph = 0; ph < n_ph; ++ph)
        for (BandPiece q : one) {
            q.phase = ph;
            p.pieces.push_back(q);
        }
    // slot = workgroup * 8 + wave of a label: phase ph owns slots [ph wslots, (ph + 1) wslots)
    p.slot_begin.assign(slots + 1, 0);
    for (int s = 0; s < slots; ++s) {
        p.slot_begin[s] = (int)p.slot_pieces.size();
        const int ph = s / wslots, w = s % wslots;
        double load = 0;
        for (int idx : lists[w]) {
            p.slot_pieces.push_back(ph * p.per_phase + idx);
            load += one[idx].i1 - one[idx].i0 + kVisit;
        }
        p.max_cost = std::max(p.max_cost, load), p.mean_cost += load / slots;
    }
    p.slot_begin[slots] = (int)p.slot_pieces.size();
    return p;
}

// ---- device ---------------------------------------------------------------------------------
typedef unsigned band_u4 __attribute__((ext_vector_type(4)));

// One octet of columns = 4 adjacent column pairs of the pair-major float64 slab behind ONE buffer
// resource (pairs past the end of the slab fall outside it).  A lane reads row t of pair
// kk = lane >> 4 at byte offset (kk pitch + t) * 16; frames past the end of the series get an offset
// outside the resource instead, so the bounds check returns zeros.  (A slab's unpaired last
// column is paired with zeros by everything that writes slabs: layout.hip.)
//
// Rows are requested by LDS-DMA (buffer_load_dwordx4 ... lds, inline assembly with hand-placed s_waitcnt)
// into a per-wave ring in LDS, kBandPF steps ahead: lane l's 16 bytes land at slot + 16 l, which IS the
// fragment layout, and the fragments are read from the ring one step ahead by ordinary LDS loads that the
// compiler schedules and waits for itself.  Round 4 requested into REGISTERS from inline assembly (the
// compiler sinks visible loads to their first use, behind the step's exit test: a memory latency per step):
// the compiler then believes the destination is defined at the asm statement, and a copy or a spill of it
// before the hand-placed wait reads stale data — nothing forbade that (the float32 form, under more
// register pressure, showed exactly such copies).  A load whose target is the LDS has no register to copy.
// What is left to inline assembly is checked in the generated code by tools/check_isa.py at build time.
constexpr int kBandPF = 2;  // steps a row request runs ahead of its use
constexpr int kBandNS = 4;  // ring slots (A block + window block, 2 KiB each); 16 % kBandNS == 0, kBandNS >= 2 kBandPF
struct BandSrc {
    band_u4 rs;                  // buffer resource of the octet (LDS-DMA, inline assembly)
    __amdgpu_buffer_rsrc_t crs;  // the same for the compiler-visible loads of a visit's first 15 fragments
    unsigned lane_off;           // (kk pitch + i) * 16
    int T, i;                    // frames; this lane's frame inside a block (lane & 15)
    bool slot;                   // Helfand form: lane group 3 carries the norms instead of columns (reads zeros)
    __device__ __forceinline__ unsigned offset(int b) const {
        unsigned off = lane_off + (unsigned)b * 256u;
        if (16 * b + 16 > T) off = 16 * b + i < T ? off : 0xfffffff0u;  // wave-uniform: a block at the end of the series
        return off;
    }
    // "s_nop 4": an SGPR of the resource (or M0) may have just been written by a VALU instruction (v_readlane
    // restoring a spilled SGPR): 5 wait states before a VMEM instruction reads it, and the hazard recogniser
    // does not look inside inline assembly.  It also covers the one wait state between the M0 write and the DMA.
    __device__ __forceinline__ void dma(unsigned lds_addr, int b) const {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(offset(b)), "s"(rs)
                     : "memory");
    }
    __device__ __forceinline__ band_d2 load_now(int b) const {  // compiler-visible: waited for at its use
        return __builtin_bit_cast(band_d2, __builtin_amdgcn_raw_buffer_load_b128(crs, offset(b), 0, 0));
    }
    // step x's two requests: rows of block x into the slot's first KiB, of block x + dw into its second
    __device__ __forceinline__ void request_step(unsigned ring_addr, int slot_idx, int x, int dw) const {
        dma(ring_addr + 2048u * (unsigned)slot_idx, x);
        dma(ring_addr + 2048u * (unsigned)slot_idx + 1024u, x + dw);
    }
};
#define TA_BAND_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")


// one piece on one octet: acc[d] += sum_{I in [i0, i1)} F_I^T F_{I + d0 + d}
// ring: this wave's LDS ring (kBandNS slots of 128 rows).  Step x's rows — A block x and the window's newest
// block x + d0 + 15 — are requested kBandPF steps ahead into slot (x - i0) % kBandNS and read into registers
// during step x - 1.
__device__ __forceinline__ void band_visit(const BandSrc& src, band_d2* ring, int d0, int i0, int i1, band_d4 (&acc)[16]) {
    static_assert(16 % kBandNS == 0 && kBandNS >= 2 * kBandPF && kBandPF >= 1, "slots are indexed by the unrolled step");
    const int lane = (int)(threadIdx.x & 63);
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)ring);  // LDS byte address
    band_d2 W[16];
#pragma unroll
    for (int d = 0; d < 15; ++d) W[d] = src.load_now(i0 + d0 + d);
#pragma unroll
    for (int s = 0; s < kBandPF; ++s) src.request_step(ring_addr, s % kBandNS, i0 + s, d0 + 15);
    TA_BAND_WAIT(2 * (kBandPF - 1));
    band_d2 anext = ring[lane], wnext = ring[64 + lane];
    int I = i0;
    for (bool more = true; more;) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            // window slot (j + d) & 15 holds F_{I + d0 + d}
            const band_d2 A = anext;
            W[(j + 15) & 15] = wnext;
            src.request_step(ring_addr, (j + kBandPF) % kBandNS, I + kBandPF, d0 + 15);
            TA_BAND_WAIT(2 * (kBandPF - 1));  // the requests of step I + 1 have landed
            anext = ring[128 * ((j + 1) % kBandNS) + lane];
            wnext = ring[128 * ((j + 1) % kBandNS) + 64 + lane];
#pragma unroll
            for (int d = 0; d < 16; ++d) acc[d] = TA_BAND_MFMA(A.x, W[(j + d) & 15].x, acc[d]);
#pragma unroll
            for (int d = 0; d < 16; ++d) acc[d] = TA_BAND_MFMA(A.y, W[(j + d) & 15].y, acc[d]);
            if (++I == i1) {  // one common tail
                more = false;
                break;
            }
        }
    }
    TA_BAND_WAIT(0);  // the requests past the piece: the ring is reused by the next visit
}

// ---- Einstein-Helfand form: sum over column c of (P[i,c] - P[j,c])^2 for every pair of frames --------
// (viscosity.py:201-233 summed over particles; P = (m v) x, the product slab of helfand_fft.hip.)
// (a - b)^2 = a^2 + b^2 - 2 a b with a = P[i] - r, b = P[j] - r for a reference row r close to
// frame i: the reference is the first frame of the A block every kBandRef steps, so |a| is the
// series' variation over < 16 kBandRef frames and |b| that over the lag — the three terms are
// each of the size of the result for all but the shortest lags of a smooth series (a pure trend at lag
// 1: 2 (16 kBandRef)^2 times larger, 1e-12 relative), which the plain expansion S1 - 2 S2 of the
// helfand_fft option cannot promise.  Three of the four 16-lane groups carry column pairs; the fourth
// carries, as its "columns", (valid_A[m], -nA[m]/2) in the A operand and (-nB[n]/2, valid_B[n]) in the
// B operand (nA, nB: squared norms of the centred rows over the group's six columns): the same 32
// MFMAs per step then accumulate  a.b - nB/2 - nA/2 = -(a - b)^2 / 2  for every pair of valid frames
// and nothing for the others; the accumulators stay of the size of the result.
#ifndef TA_BAND_REF
#define TA_BAND_REF 16
#endif
#ifndef TA_BAND_ABL  // timing ablations (wrong results): 1 no re-centring, 2 no preparation at all
#define TA_BAND_ABL 0
#endif
constexpr int kBandRef = TA_BAND_REF;


struct BandHelf {
    int T, i;
    bool slot;
    double slot_one;  // 1 in the fourth lane group, 0 elsewhere
    // Every VALU instruction of this kernel costs matrix-pipe time (measured: FP64 MFMAs and vector
    // instructions do not overlap on gfx950, interleaved or not), so the common case — a block that lies
    // inside the series — is kept to the fewest instructions; blocks that reach the end take the
    // branch with the per-lane selects (wave-uniform).
    // The reference row carries (-1, 0) in the fourth lane group, whose rows read as zeros: "row - r" puts
    // (1, 0) there, which IS the A operand's (valid, -nA/2 = 0) of a step without invalid pairs.
    __device__ __forceinline__ band_d2 first_row(band_d2 raw) const {
        const int src_lane = (int)(threadIdx.x & 48);
        band_d2 r = band_d2{__shfl(raw.x, src_lane), __shfl(raw.y, src_lane)};
        if (slot) r = band_d2{-1.0, 0.0};
        return r;
    }
    // B operand of block b from rows that are already centred except for `shift` (rows - r, or old + delta)
    // `exact_one`: the fourth group holds exactly (1, 0); its square is taken off BEFORE the sum over the lane
    // groups (fma(c.x, c.x, -slot_one) = 0 there), never after it: 1 + (a norm of 1e-20) - 1 is 0 — round 4's
    // form did that and lost every digit of small-valued data (P in other units: 70 % wrong at P ~ 1e-10;
    // tests/test_gpu_parity.py::test_helfand_matrix_cores_do_not_depend_on_the_unit)
    template <bool exact_one>
    __device__ __forceinline__ band_d2 finish_b(band_d2 c, int b) const {
        if (16 * b + 16 <= T) {
            double n;
            if constexpr (exact_one) n = -0.5 * band_sum_rows(__builtin_fma(c.y, c.y, __builtin_fma(c.x, c.x, -slot_one)));
            else n = -0.5 * band_sum_rows(slot ? 0.0 : c.x * c.x + c.y * c.y);
            if (slot) c = band_d2{n, 1.0};
        } else {
            const bool valid = 16 * b + i < T;
            if (!valid) c = band_d2{0.0, 0.0};
            const double n = -0.5 * band_sum_rows(slot ? 0.0 : c.x * c.x + c.y * c.y);
            if (slot) c = band_d2{n, valid ? 1.0 : 0.0};
        }
        return c;
    }
    __device__ __forceinline__ band_d2 prep_b(band_d2 raw, band_d2 r, int b) const { return finish_b<true>(raw - r, b); }
    __device__ __forceinline__ band_d2 recentre_b(band_d2 old, band_d2 delta, int b) const { return finish_b<false>(old + delta, b); }
    // A operand of a step whose window reaches the end of the series: (valid, -nA/2) through the product
    __device__ __forceinline__ band_d2 prep_a_edge(band_d2 raw, band_d2 r, int b) const {
        const bool valid = 16 * b + i < T;
        band_d2 c = raw - r;
        if (!valid) c = band_d2{0.0, 0.0};
        const double n = -0.5 * band_sum_rows(slot ? 0.0 : c.x * c.x + c.y * c.y);
        if (slot) c = band_d2{valid ? 1.0 : 0.0, n};
        return c;
    }
    // ... and of every other step: all its pairs are valid, the rows' norms would add the same -nA[m]/2 to
    // all 16 accumulators, so they are summed per lane in `na` (fourth group: a count, dropped in the
    // epilogue) and subtracted once
    __device__ __forceinline__ band_d2 prep_a_bulk(band_d2 raw, band_d2 r, double& na) const {
        const band_d2 c = raw - r;
        na += c.x * c.x + c.y * c.y;
        return c;
    }
};

// The reference row (frame 16 I of every column, every kBandRef steps) is lane i = 0 of the A fragment itself.
// Requests and the ring: as band_visit.
__device__ __forceinline__ void band_visit_helfand(const BandSrc& src, band_d2* ring, int d0, int i0, int i1, band_d4 (&acc)[16],
                                                   double& na) {
    const BandHelf h{src.T, src.i, src.slot, src.slot ? 1.0 : 0.0};
    const int lane = (int)(threadIdx.x & 63);
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)ring);
    band_d2 W[16], r;
#pragma unroll
    for (int d = 0; d < 15; ++d) W[d] = src.load_now(i0 + d0 + d);
#pragma unroll
    for (int s = 0; s < kBandPF; ++s) src.request_step(ring_addr, s % kBandNS, i0 + s, d0 + 15);
    TA_BAND_WAIT(2 * (kBandPF - 1));
    band_d2 anext = ring[lane], wnext = ring[64 + lane];
    r = h.first_row(anext);
#pragma unroll
    for (int d = 0; d < 15; ++d) {  // (the 16th is prepared by the first step, like every step's newest)
        W[d] = h.prep_b(W[d], r, i0 + d0 + d);
        if (d % 3 == 2) __builtin_amdgcn_sched_barrier(0);  // three chains at a time: registers
    }
    int I = i0;
    for (bool more = true; more;) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            // this step's operands from the rows read during the previous step ...
            if (TA_BAND_ABL == 0 && j % kBandRef == 0 && I != i0) {  // new reference: the 15 older window fragments follow it
                const band_d2 rn = h.first_row(anext);
                const band_d2 delta = r - rn;
                r = rn;
#pragma unroll
                for (int d = 0; d < 15; ++d) {
                    W[(j + d) & 15] = h.recentre_b(W[(j + d) & 15], delta, I + d0 + d);
                    if (d % 3 == 2) __builtin_amdgcn_sched_barrier(0);
                }
            }
            band_d2 A;
            if (TA_BAND_ABL >= 2) A = anext;
            else if (16 * (I + d0 + 16) <= src.T) A = h.prep_a_bulk(anext, r, na);
            else A = h.prep_a_edge(anext, r, I);
            W[(j + 15) & 15] = TA_BAND_ABL < 2 ? h.prep_b(wnext, r, I + d0 + 15) : wnext;
            // ... then the next step's rows leave the ring (their latency is this step's MFMAs)
            src.request_step(ring_addr, (j + kBandPF) % kBandNS, I + kBandPF, d0 + 15);
            TA_BAND_WAIT(2 * (kBandPF - 1));
            anext = ring[128 * ((j + 1) % kBandNS) + lane];
            wnext = ring[128 * ((j + 1) % kBandNS) + 64 + lane];
#pragma unroll
            for (int d = 0; d < 16; ++d) acc[d] = TA_BAND_MFMA(A.x, W[(j + d) & 15].x, acc[d]);
#pragma unroll
            for (int d = 0; d < 16; ++d) acc[d] = TA_BAND_MFMA(A.y, W[(j + d) & 15].y, acc[d]);
            if (++I == i1) {
                more = false;
                break;
            }
        }
    }
    TA_BAND_WAIT(0);
}

// pm: pair-major float64 slab (layout.hip).  grid: n_labels * (slots / 8) workgroups of 512.
// HELF: pm is the product slab P; columns go in groups of three pairs, results are -1/2 the squared differences
template <bool HELF>
__global__ void __launch_bounds__(512)
    k_band_lags(const double* __restrict__ pm, long pitch, int T, long n_pairs, int n_labels, int n_ph,
                const BandPiece* __restrict__ pieces, int n_pieces, const int* __restrict__ slot_begin,
                const int* __restrict__ slot_pieces, double* __restrict__ partial) {
    // per wave: one accumulator block as [m][n] with a row stride of 17, and the 16 x 31 diagonal sums
    // (LDS operations of one wave complete in order: wave_barrier only pins the compiler's order)
    __shared__ double red[8][272 + 16 * 32];
    __shared__ band_d2 rings[8][kBandNS * 128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int label = blockIdx.x % n_labels;
    const int slot = (blockIdx.x / n_labels) * 8 + wave;
    constexpr int kPairs = HELF ? 3 : 4;  // column pairs per visit
    const long n_oct = (n_pairs + kPairs - 1) / kPairs;
    const int kk = lane >> 4;
    double* blk = red[wave];
    double* dsum = blk + 272;
    const int pb = __builtin_amdgcn_readfirstlane(slot_begin[slot]), pe = __builtin_amdgcn_readfirstlane(slot_begin[slot + 1]);
    for (int pi = pb; pi < pe; ++pi) {
        const int idx = __builtin_amdgcn_readfirstlane(slot_pieces[pi]);
        const BandPiece pc = pieces[idx];
        const int d0 = __builtin_amdgcn_readfirstlane(pc.d0), i0 = __builtin_amdgcn_readfirstlane(pc.i0),
                  i1 = __builtin_amdgcn_readfirstlane(pc.i1), phase = __builtin_amdgcn_readfirstlane(pc.phase);
        band_d4 acc[16];
        double na = 0.0;  // (HELF) squared norms of the A rows of the steps that did not put them through the product
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[d] = band_d4{0.0, 0.0, 0.0, 0.0};
        for (long o = label + (long)n_labels * phase; o < n_oct; o += (long)n_labels * n_ph) {
            const long left = n_pairs - kPairs * o;  // pairs of this octet that exist
            BandSrc src;
            const unsigned long long base = reinterpret_cast<unsigned long long>(pm + kPairs * o * pitch * 2);
            // raw buffer resource: base, stride 0, num_records in bytes, the gfx9 data format word
            const unsigned n_bytes = (unsigned)((left < kPairs ? left : kPairs) * pitch) * 16u;
            src.rs = band_u4{(unsigned)base, (unsigned)(base >> 32) & 0xffffu, n_bytes, 0x00020000u};
            src.crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(pm + kPairs * o * pitch * 2), 0, (int)n_bytes, 0x00020000);
            // (Helfand form: the fourth lane group reads just past the resource's last byte: zeros)
            src.lane_off = (unsigned)(kk * pitch + (lane & 15)) * 16u;
            if (HELF && kk == 3) src.lane_off = (unsigned)(3 * pitch) * 16u + (unsigned)(lane & 15) * 16u;
            src.T = T;
            src.i = lane & 15;
            src.slot = HELF && kk == 3;
            if constexpr (HELF) band_visit_helfand(src, rings[wave], d0, i0, i1, acc, na);
            else band_visit(src, rings[wave], d0, i0, i1, acc);
        }
        // diagonals: acc[d] register r of lane l is C_d[m = 4 r + (l >> 4)][n = l & 15], lag 16 (d0 + d) + n - m
        double na_m[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (HELF) {  // nA[m] / 2 of the lane's four rows m = 4 r + (l >> 4), through the LDS
            const double tot = band_sum_rows(kk == 3 ? 0.0 : na);
            if (lane < 16) dsum[lane] = 0.5 * tot;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r) na_m[r] = dsum[4 * r + (lane >> 4)];
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int d = 0; d < 16; ++d) {
#pragma unroll
            for (int r = 0; r < 4; ++r) blk[(4 * r + (lane >> 4)) * 17 + (lane & 15)] = acc[d][r] - na_m[r];
            __builtin_amdgcn_wave_barrier();
            if (lane < 31) {
                const int e = lane - 15;
                const int m_lo = e < 0 ? -e : 0, m_hi = e > 0 ? 16 - e : 16;
                double s = 0.0;
                for (int m = m_lo; m < m_hi; ++m) s += blk[m * 17 + m + e];
                dsum[d * 32 + lane] = s;
            }
            __builtin_amdgcn_wave_barrier();
        }
        double* out = partial + ((long)label * n_pieces + idx) * kBandPartial;
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
        __builtin_amdgcn_wave_barrier();
    }
}

// lagsum[k] = factor * (sum over labels and over the pieces that hold lag k) / (T - k), fixed order;
// zero_lag0: lagsum[0] = 0 exactly (viscosity.py:205-233 leaves row 0 at 0)
static __global__ void __launch_bounds__(256)  // (static: this header is included by band.hip and band32.hip)
    k_band_gather(const double* __restrict__ partial, int n_labels, int n_pieces, int n_ph, int per_phase,
                  const int* __restrict__ group_begin, int n_groups, int T, double factor, int zero_lag0,
                  double* __restrict__ lagsum) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= T) return;
    double s = 0.0;
    for (int h = 0; h < 2; ++h) {
        // group g holds lag offsets -15 ... 255 from 256 g
        const int g = (k >> 8) + h;
        const int off = k - 256 * g;
        if (g >= n_groups || off < -15) continue;
        for (int ph = 0; ph < n_ph; ++ph)
            for (int li = group_begin[g]; li < group_begin[g + 1]; ++li)
                for (int lab = 0; lab < n_labels; ++lab)
                    s += partial[((long)lab * n_pieces + ph * per_phase + li) * kBandPartial + off + 15];
    }
    lagsum[k] = (zero_lag0 && k == 0) ? 0.0 : factor * (s / (double)(T - k));
}

}  // namespace ta
