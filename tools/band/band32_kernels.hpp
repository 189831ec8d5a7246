// band32_kernels.hpp — Einstein-Helfand lag SUMS on the FP32 matrix cores (gfx950 v_mfma_f32_16x16x4_f32):
// BASELINE configs[4]'s "float32 path" without the by-particle array.
//
// Quantity: ViscosityHelfand._conclude summed over particles
// (/root/reference/transport_analysis/viscosity.py:201-233): for every lag k
//   S[k] = sum_{columns c} sum_{i < T-k} (P[i, c] - P[i+k, c])^2,  P = (m v) x rounded ONCE to float32
// (what the float32 vector kernel, direct_kernels.hpp, rounds too: the "direct_f32" contract is float32
// values and products, float64 accumulation, 2e-6 of the series' scale).
//
// Structure: band_kernels.hpp's Helfand form (read that file's comments first) — blocks of 16 frames, a
// wave owns 16 block lags and a range of blocks, rows centred on a frame close to the A block so that
// (a - b)^2 = a^2 + b^2 - 2 a b has no cancellation, the squared norms carried by the fourth 16-lane group —
// with these differences:
//   * the product slab is float32 (8-byte rows: half the bytes the float64 form pulls through the fabric
//     at the same number of requests per MFMA), centring, norms and fragments are float32;
//   * the MFMA is v_mfma_f32_16x16x4_f32: 2048 flop in 32 cycles per SIMD, twice the FP64 form's rate, and
//     its result rows are m = 4 (lane >> 4) + r (the FP64 form: (lane >> 4) + 4 r);
//   * float32 accumulators hold at most kFlush steps (128 products each) and are then added into
//     float64 accumulators in registers: the vector path's own discipline (float32 block sums, float64
//     accumulation) — sums over 10^6 terms in float32 would lose five digits;
//   * a step of 1024 MFMA cycles is shorter than a memory round trip, so rows are requested PF steps ahead —
//     not into registers but into a per-wave LDS ring, by LDS-DMA (buffer_load_dwordx4 ... lds): one request
//     per step brings the A block and the window's new block (32 lanes x two 8-byte rows each), no register
//     is the target of a load in flight (the float64 form's inline-assembly loads rely on the compiler not
//     copying such a register; under this kernel's register pressure it does), and the fragments are read
//     from the ring one step ahead with ordinary LDS loads the compiler schedules and waits for itself.
#pragma once
#include "band_kernels.hpp"  // (tools/band: includes csrc/band_common.hpp)

namespace ta {


// One sextet of columns = 3 adjacent column pairs of the pair-major FLOAT32 slab (8-byte rows) behind one
// buffer resource.  A request is ONE LDS-DMA instruction: lane L < 32 fetches rows 2 (L & 7), + 1 of block
// `lo` of pair L >> 3, lane L >= 32 the same rows of block `hi` of pair (L - 32) >> 3, 16 bytes each, landing
// at ring slot + 16 L: the slot is [2 blocks][4 lane groups][16 frames] of 8-byte rows, i.e. fragment lane
// l of the A block at [l], of the window's new block at [64 + l].  The fourth lane group (L & 24 == 24: the
// norm slot) and everything past the resource (pairs that do not exist) fetch out of range: zeros.  Rows
// past the end of the series inside the resource are NOT zeros (pad rows, the next pair's rows): the
// consumers mask by frame number wherever a block is not entirely inside the series.
struct BandSrc32 {
    band_u4 rs;          // raw buffer resource for the LDS-DMA (inline assembly)
    __amdgpu_buffer_rsrc_t crs;  // the same for compiler-visible loads (the visit's first 15 fragments)
    unsigned dma_off;    // this lane's byte offset for block 0 of its half (out of range in the fourth lane group)
    unsigned dma_hi;     // 128 in lanes >= 32, else 0: block `hi` = `lo` + delta
    unsigned lane_off;   // (kk pitch + i) * 8 of the fragment lane, out of range in the fourth lane group
    int T, i;
    bool slot;
    // "s_nop 4": an SGPR of the resource restored by v_readlane just before (hazard recogniser does not look
    // inside inline assembly; tools/check_isa.py verifies the nop is there); s_nop 0: M0 write -> LDS-DMA
    __device__ __forceinline__ void dma(unsigned lds_addr, unsigned voff) const {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs)
                     : "memory");
    }
    __device__ __forceinline__ band_f2 load_now(int b) const {  // compiler-visible, waited for at its use
        return __builtin_bit_cast(band_f2, __builtin_amdgcn_raw_buffer_load_b64(crs, lane_off + (unsigned)b * 128u, 0, 0));
    }
};
#define TA_BAND32_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")


__device__ __forceinline__ band_f2 band32_sub(band_f2 a, band_f2 b) { return a - b; }
__device__ __forceinline__ band_f2 band32_add(band_f2 a, band_f2 b) { return a + b; }

struct BandHelf32 {
    int T, i;
    bool slot;
    float slot_one;  // 1 in the fourth lane group, 0 elsewhere
    // The reference row carries (-1, 0) in the fourth lane group, whose rows read as zeros: "row - r" puts
    // (1, 0) there = the A operand's (valid, -nA/2 = 0) of a step without invalid pairs.  The squared norm
    // starts from -slot_one, so that exact 1 never enters the sum over the lane groups (a norm of 1e-12
    // beside it would lose its digits: the result must not depend on the unit of P).
    __device__ __forceinline__ band_f2 first_row(band_f2 raw) const {
        const int src_lane = (int)(threadIdx.x & 48);
        band_f2 r = band_f2{__shfl(raw.x, src_lane), __shfl(raw.y, src_lane)};
        if (slot) r = band_f2{-1.0f, 0.0f};
        return r;
    }
    __device__ __forceinline__ float norm2(band_f2 c) const {  // columns: |c|^2; fourth group with c = (1, 0): exactly 0
        return __builtin_fmaf(c.y, c.y, __builtin_fmaf(c.x, c.x, -slot_one));
    }
    // B operand of block b from rows that are centred already (rows - r, or old + delta).
    // one_in_slot: the fourth group holds exactly (1, 0) (rows - r); inside: block b lies inside the series
    template <bool one_in_slot, bool inside>
    __device__ __forceinline__ band_f2 finish_b(band_f2 c, int b) const {
        if constexpr (inside) {
            const float n = -0.5f * band32_sum_rows(one_in_slot ? norm2(c) : (slot ? 0.0f : c.x * c.x + c.y * c.y));
            if (slot) c = band_f2{n, 1.0f};
        } else {
            const bool valid = 16 * b + i < T;
            if (!valid) c = band_f2{0.0f, 0.0f};
            const float n = -0.5f * band32_sum_rows(slot ? 0.0f : c.x * c.x + c.y * c.y);
            if (slot) c = band_f2{n, valid ? 1.0f : 0.0f};
        }
        return c;
    }
    // The operands of a step, for any block (inside the series or reaching past its end): rows centred, frames
    // past the end zeroed, -1/2 the squared norm over the sextet in the fourth lane group beside the
    // validity flag — (valid, -nA/2) for the A operand, (-nB/2, valid) for the B operand.  (raw - r) is (1, 0) in
    // the fourth lane group, so masking alone leaves the flag in c.x.
    __device__ __forceinline__ band_f2 centred(band_f2 raw, band_f2 r, int b, float& n) const {
        band_f2 c = band32_sub(raw, r);
        const bool valid = i < T - 16 * b;
        if (!valid) c = band_f2{0.0f, 0.0f};
        n = -0.5f * band32_sum_rows(slot ? 0.0f : c.x * c.x + c.y * c.y);
        return c;
    }
    // A operand of a step whose whole window lies inside the series: its rows' norms would add the same -nA[m]/2
    // to all 16 accumulators, so they are summed per lane (fourth group: exactly 0) and subtracted once in the
    // epilogue
    __device__ __forceinline__ band_f2 prep_a_bulk(band_f2 raw, band_f2 r, float& na) const {
        const band_f2 c = band32_sub(raw, r);
        na += norm2(c);
        return c;
    }
    __device__ __forceinline__ band_f2 prep_a(band_f2 raw, band_f2 r, int b) const {
        float n;
        band_f2 c = centred(raw, r, b, n);
        if (slot) c.y = n;
        return c;
    }
    __device__ __forceinline__ band_f2 prep_b(band_f2 raw, band_f2 r, int b) const {
        float n;
        band_f2 c = centred(raw, r, b, n);
        if (slot) c = band_f2{n, c.x};
        return c;
    }
};

#ifndef TA_BAND32_ABL  // timing ablations (wrong results), bit mask: 1 no request / ring read in the loop, 2 no preparation of
#define TA_BAND32_ABL 0  // the window's new fragment, 4 no re-centring, 8 no flush, 16 no preparation of the A operand
#endif
#ifndef TA_BAND32_FLUSH  // (harness: 16 ... 512 measured, tools/band/build32.sh; 20000 x 25000 x 3: 394 / 346 / 337 / 333 / 332 ms
#define TA_BAND32_FLUSH 128  // at 16 / 64 / 128 / 256 / 512, the lag sums equal to 1e-10 between 16 and 512)
#endif
constexpr int kBand32Flush = TA_BAND32_FLUSH;  // steps a float32 accumulator holds before it is added into float64


// one piece on one sextet.  ring: this wave's LDS ring, NS slots of 128 rows (1 KiB); the request of step x
// (A block x, window block x + d0 + 15) is issued PF steps ahead into slot (x - i0) % NS and read into
// registers one step ahead.  Every 16 steps the reference row moves to the A block's first frame and the 15
// older window fragments follow it.
// (Measured, docs/HISTORY.md "Round 5: the column-packed float32 forms": v_mfma_f32_16x16x4_f32 and vector instructions do NOT overlap on gfx950 any more
// than the FP64 form's do — making step x + 1's operands between step x's MFMAs, one MFMA : two vector
// instructions, took 424 ms where this form took 389 (20000 x 25000 x 3, one wave per SIMD): a vector
// instruction costs its time wherever it stands, and with one wave per SIMD that time is its latency (the
// window fragment's preparation is one dependent chain).  So the loop is kept to the fewest instructions —
// blocks inside the series take a path without masks, the A rows' norms are summed per lane instead of going
// through the product — and the kernel runs TWO waves per SIMD, which the diagonal flush above makes room for.)
template <int PF, int NS>
__device__ __forceinline__ void band32_visit(const BandSrc32& src, band_f2* ring, int d0, int i0, int i1, band_f4 (&acc)[16],
                                             Band32Diag& sums, double& na, int& since) {
    static_assert(16 % NS == 0 && NS >= 2 * PF && PF >= 1, "slots are indexed by the unrolled step; a slot is rewritten PF steps after it was read");
    static_assert(kBand32Flush % 16 == 0, "flushes happen between passes of the ring");
    const BandHelf32 h{src.T, src.i, src.slot, src.slot ? 1.0f : 0.0f};
    const int lane = (int)(threadIdx.x & 63);
    // LDS byte address of the ring (low half of the flat address; the same in every lane of the wave)
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)ring);
    const unsigned voff0 = src.dma_off + src.dma_hi * (unsigned)(d0 + 15);
    band_f2 W[16], r;
    float na32 = 0.0f;
#pragma unroll
    for (int d = 0; d < 15; ++d) W[d] = src.load_now(i0 + d0 + d);
#pragma unroll
    for (int s = 0; s < PF; ++s) src.dma(ring_addr + 1024u * (s % NS), voff0 + 128u * (unsigned)(i0 + s));
    TA_BAND32_WAIT(PF - 1);
    band_f2 anext = ring[lane], wnext = ring[64 + lane];
    r = h.first_row(anext);
    if (16 * (i0 + d0 + 15) <= src.T) {  // the 15 blocks lie inside the series
#pragma unroll
        for (int d = 0; d < 15; ++d) W[d] = h.template finish_b<true, true>(band32_sub(W[d], r), 0);
    } else {
#pragma unroll
        for (int d = 0; d < 15; ++d) W[d] = h.template finish_b<true, false>(band32_sub(W[d], r), i0 + d0 + d);
    }
    int I = i0;
    for (bool more = true; more;) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j == 0 && !(TA_BAND32_ABL & 8) && ++since == kBand32Flush / 16) {  // (a visit's first pass counts: a piece of
                sums.flush(acc);                                                    // short visits must not outrun the limit)
                since = 0;
            }
            // this step's operands from the rows read during the previous step ...
            if (j == 0 && I != i0 && !(TA_BAND32_ABL & 4)) {  // a new pass: a new reference row that the 15 older window fragments follow
                const band_f2 rn = h.first_row(anext);
                const band_f2 delta = band32_sub(r, rn);
                r = rn;
                if (16 * (I + d0 + 15) <= src.T) {
#pragma unroll
                    for (int d = 0; d < 15; ++d) W[d] = h.template finish_b<false, true>(band32_add(W[d], delta), 0);
                } else {
#pragma unroll
                    for (int d = 0; d < 15; ++d) W[d] = h.template finish_b<false, false>(band32_add(W[d], delta), I + d0 + d);
                }
            }
            band_f2 A;
            if (TA_BAND32_ABL & 18) {
                A = (TA_BAND32_ABL & 16) ? anext : h.prep_a_bulk(anext, r, na32);
                W[(j + 15) & 15] = (TA_BAND32_ABL & 2) ? wnext : h.template finish_b<true, true>(band32_sub(wnext, r), 0);
            } else if (16 * (I + d0 + 16) <= src.T) {  // this step's whole window inside the series
                A = h.prep_a_bulk(anext, r, na32);
                W[(j + 15) & 15] = h.template finish_b<true, true>(band32_sub(wnext, r), 0);
            } else {
                A = h.prep_a(anext, r, I);
                W[(j + 15) & 15] = h.prep_b(wnext, r, I + d0 + 15);
            }
            // ... then the next step's rows leave the ring (their latency is this step's MFMAs)
            if (!(TA_BAND32_ABL & 1)) {
                src.dma(ring_addr + 1024u * ((j + PF) % NS), voff0 + 128u * (unsigned)(I + PF));
                TA_BAND32_WAIT(PF - 1);  // the request of step I + 1 has landed
                anext = ring[128 * ((j + 1) % NS) + lane];
                wnext = ring[128 * ((j + 1) % NS) + 64 + lane];
            }
#pragma unroll
            for (int d = 0; d < 16; ++d) acc[d] = TA_BAND32_MFMA(A.x, W[(j + d) & 15].x, acc[d]);
#pragma unroll
            for (int d = 0; d < 16; ++d) acc[d] = TA_BAND32_MFMA(A.y, W[(j + d) & 15].y, acc[d]);
            if (++I == i1) {  // (one common tail)
                more = false;
                break;
            }
        }
    }
    TA_BAND32_WAIT(0);  // the requests past the piece: the ring is reused by the next visit
    na += (double)na32;  // (float32 over one visit's steps: a few hundred terms per lane)
}

// pm: pair-major FLOAT32 product slab (8-byte rows).  grid: n_labels * (slots / NW) workgroups of 64 NW threads.
// Results: partial[label][piece][272] = -1/2 the squared differences (k_band_gather applies the factor -2).
template <int NW, int PF, int NS>
__global__ void __launch_bounds__(64 * NW)
    k_band32_lags(const float* __restrict__ pm, long pitch, int T, long n_pairs, int n_labels, int n_ph,
                  const BandPiece* __restrict__ pieces, int n_pieces, const int* __restrict__ slot_begin,
                  const int* __restrict__ slot_pieces, double* __restrict__ partial, unsigned long long* __restrict__ stamps) {
    // stamps (diagnostics; NULL in the library): per wave, shader cycles and 100 MHz ticks from entry to exit
    __shared__ float diag[NW][32 * 17 + 16 * 32];
    __shared__ double na_half[NW][16];
    __shared__ band_f2 rings[NW][NS * 128];
    const unsigned long long t_c0 = stamps ? __builtin_amdgcn_s_memtime() : 0, t_r0 = stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int label = blockIdx.x % n_labels;
    const int slot = (blockIdx.x / n_labels) * NW + wave;
    const long n_sext = (n_pairs + 2) / 3;
    const int kk = lane >> 4;
    Band32Diag sums;
    sums.init(diag[wave], lane);
    const int pb = __builtin_amdgcn_readfirstlane(slot_begin[slot]), pe = __builtin_amdgcn_readfirstlane(slot_begin[slot + 1]);
    for (int pi = pb; pi < pe; ++pi) {
        const int idx = __builtin_amdgcn_readfirstlane(slot_pieces[pi]);
        const BandPiece pc = pieces[idx];
        const int d0 = __builtin_amdgcn_readfirstlane(pc.d0), i0 = __builtin_amdgcn_readfirstlane(pc.i0),
                  i1 = __builtin_amdgcn_readfirstlane(pc.i1), phase = __builtin_amdgcn_readfirstlane(pc.phase);
        band_f4 acc[16];
        double na = 0.0;  // squared norms of the A rows of the steps that did not put them through the product
        // passes since the last flush; the second wave of a SIMD (waves w and w + NW / 2 share one) starts half a
        // period in, so that the two do not stand in their flushes — LDS round trips — at the same time
        int since = (NW >= 8 && wave >= NW / 2) ? kBand32Flush / 32 : 0;
        sums.clear();
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[d] = band_f4{0.0f, 0.0f, 0.0f, 0.0f};
        for (long o = label + (long)n_labels * phase; o < n_sext; o += (long)n_labels * n_ph) {
            const long left = n_pairs - 3 * o;  // pairs of this sextet that exist
            BandSrc32 src;
            const unsigned long long base = reinterpret_cast<unsigned long long>(pm + 3 * o * pitch * 2);
            const unsigned n_bytes = (unsigned)((left < 3 ? left : 3) * pitch) * 8u;
            src.rs = band_u4{(unsigned)base, (unsigned)(base >> 32) & 0xffffu, n_bytes, 0x00020000u};
            src.crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pm + 3 * o * pitch * 2), 0, (int)n_bytes, 0x00020000);
            const int dk = (lane >> 3) & 3;  // the DMA lane's pair (3: the norm slot's zeros)
            src.dma_off = dk == 3 ? 0x80000000u : (unsigned)(dk * pitch + 2 * (lane & 7)) * 8u;
            src.dma_hi = lane >= 32 ? 128u : 0u;
            src.lane_off = kk == 3 ? 0x80000000u : (unsigned)(kk * pitch + (lane & 15)) * 8u;
            src.T = T;
            src.i = lane & 15;
            src.slot = kk == 3;
            band32_visit<PF, NS>(src, rings[wave], d0, i0, i1, acc, sums, na, since);
        }
        sums.flush(acc);
        // the A norms left out of the products: nA[m] / 2 of block row m, on every diagonal that row m lies on
        {
            const double t0 = band_sum_rows(na);  // (fourth lane group: exactly 0)
            if (lane < 16) na_half[wave][lane] = 0.5 * t0;
            __builtin_amdgcn_wave_barrier();
        }
        auto corr = [&](int e) {  // sum of nA[m] / 2 over the rows of diagonal e
            const int m_lo = e < 0 ? -e : 0, m_hi = e > 0 ? 16 - e : 16;
            double c = 0.0;
            for (int m = m_lo; m < m_hi; ++m) c += na_half[wave][m];
            return c;
        };
        double* out = partial + ((long)label * n_pieces + idx) * kBandPartial;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int q = lane + 64 * k;
            if (q < kBandPartial) {
                const int off = q - 15;
                const int d = (off + 16) / 16 - 1, e = off - 16 * d;
                double c = 0.0;
                if (d >= 0 && d <= 15) c = corr(e);
                if (e >= 1 && d + 1 >= 0 && d + 1 <= 15) c += corr(e - 16);
                out[q] = off <= 255 ? sums.s[k] - c : 0.0;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (stamps && lane == 0) {
        stamps[2 * ((long)blockIdx.x * NW + wave)] = __builtin_amdgcn_s_memtime() - t_c0;
        stamps[2 * ((long)blockIdx.x * NW + wave) + 1] = __builtin_amdgcn_s_memrealtime() - t_r0;
    }
}

// ---- Einstein-Helfand WITH the by-particle array (dim = 3) on the FP32 matrix cores ------------------------
// results.visc_by_particle of viscosity.py:201-233 under the float32 option: every particle's own mean squared
// differences.  The contraction of a product is then over ONE particle's three columns: lane groups 0..2 carry
// x, y, z (ONE float32 per lane, a single MFMA per block lag and step), the fourth carries valid_A | -nB/2 as
// above: 3 of the instruction's 4 k-slots do arithmetic.  (The FP64 form of this, tools/band/helfand_bp_mfma.patch,
// tied with the vector kernel; here the matrix instruction is twice as fast and the float32 vector kernel needs a
// subtraction and an FMA per term.)
// A unit = (particle, 16 block lags starting at 15 g): it owns lags 240 g ... 240 g + 239 — a lag needs block lags
// d and d + 1, so consecutive units overlap by one block lag (16/15 of the MFMAs) instead of adding up halves —
// runs the whole band of those block lags, and writes its 240 values into the atom-major scratch that
// k_bp_transpose turns into (n_frames, n_particles).  Units are dealt round-robin to the waves: the ~T/240 units of
// a particle run on different waves at about the same time, on the same 12 T bytes.
// Rows: two LDS-DMA requests of one dword per lane and step (A block, window block): lane (k, i) fetches column
// 3 p + k of frame i — 4 bytes at ((c >> 1) pitch + t) * 8 + (c & 1) * 4 — which lands at slot + 4 lane: the
// fragment itself.
struct BandSrc32s {
    band_u4 rs;         // the two column pairs that hold the particle's three columns
    unsigned lane_off;  // byte offset of the lane's column at frame i (fourth lane group: out of range)
    int T, i;
    bool slot;
    __device__ __forceinline__ void dma(unsigned lds_addr, int b) const {
        const unsigned off = lane_off + (unsigned)b * 128u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(off), "s"(rs)
                     : "memory");
    }
};

struct BandHelf32s {
    int T, i;
    bool slot;
    float slot_one;
    __device__ __forceinline__ float first_row(float raw) const {
        const float r = __shfl(raw, (int)(threadIdx.x & 48));
        return slot ? -1.0f : r;  // "row - r" = 1 = valid_A in the fourth lane group (its rows read as zeros)
    }
    // B operand: centred row, fourth lane group -nB/2 (inside: the block lies inside the series; one_in_slot: the
    // fourth group holds exactly 1, taken off before the sum over the lane groups)
    template <bool one_in_slot, bool inside>
    __device__ __forceinline__ float finish_b(float c, int b) const {
        if constexpr (!inside)
            if (!(i < T - 16 * b)) c = 0.0f;
        const float q = (one_in_slot && inside) ? __builtin_fmaf(c, c, -slot_one) : (slot ? 0.0f : c * c);
        const float n = -0.5f * band32_sum_rows(q);
        return slot ? n : c;
    }
};

// one unit: acc[d] += sum over I in [0, nsteps) and the particle's columns of -(a - b)^2 / 2 for block lags d0 + d
template <int PF, int NS>
__device__ __forceinline__ void band32_unit_bp(const BandSrc32s& src, float* ring, int d0, int nsteps, band_f4 (&acc)[16],
                                               Band32Diag& sums, double& na) {
    static_assert(16 % NS == 0 && NS >= 2 * PF && PF >= 1, "slots are indexed by the unrolled step");
    const BandHelf32s h{src.T, src.i, src.slot, src.slot ? 1.0f : 0.0f};
    const int lane = (int)(threadIdx.x & 63);
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)ring);
    auto request_step = [&](int slot_idx, int x) {  // step x: A block x, window block x + d0 + 15
        src.dma(ring_addr + 512u * (unsigned)slot_idx, x);
        src.dma(ring_addr + 512u * (unsigned)slot_idx + 256u, x + d0 + 15);
    };
    float W[16], r, na32 = 0.0f;
    // the first 15 window fragments through the ring as well: one block per 64-float half slot, 2 NS at a time
#pragma unroll
    for (int base = 0; base < 15; base += 2 * NS) {
#pragma unroll
        for (int q = 0; q < 2 * NS; ++q)
            if (base + q < 15) src.dma(ring_addr + 256u * (unsigned)q, d0 + base + q);
        TA_BAND32_WAIT(0);
#pragma unroll
        for (int q = 0; q < 2 * NS; ++q)
            if (base + q < 15) W[base + q] = ring[64 * q + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read before the next requests may overwrite the half slots
    }
#pragma unroll
    for (int s = 0; s < PF; ++s) request_step(s % NS, s);
    TA_BAND32_WAIT(2 * (PF - 1));
    float anext = ring[lane], wnext = ring[64 + lane];
    r = h.first_row(anext);
    if (16 * (d0 + 15) <= src.T) {
#pragma unroll
        for (int d = 0; d < 15; ++d) W[d] = h.template finish_b<true, true>(W[d] - r, 0);
    } else {
#pragma unroll
        for (int d = 0; d < 15; ++d) W[d] = h.template finish_b<true, false>(W[d] - r, d0 + d);
    }
    int I = 0, since = 0;
    for (bool more = true; more;) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j == 0 && ++since == kBand32Flush / 16) {
                sums.flush(acc);
                since = 0;
            }
            if (j == 0 && I != 0) {  // a new pass: a new reference row that the 15 older window fragments follow
                const float rn = h.first_row(anext);
                const float delta = r - rn;
                r = rn;
                if (16 * (I + d0 + 15) <= src.T) {
#pragma unroll
                    for (int d = 0; d < 15; ++d) W[d] = h.template finish_b<false, true>(W[d] + delta, 0);
                } else {
#pragma unroll
                    for (int d = 0; d < 15; ++d) W[d] = h.template finish_b<false, false>(W[d] + delta, I + d0 + d);
                }
            }
            const bool bulk = 16 * (I + d0 + 16) <= src.T;  // every pair of this step is valid
            float A = anext - r, na_half = 0.0f;
            if (bulk) {
                na32 = __builtin_fmaf(A, A, na32 - h.slot_one);  // (fourth lane group: 1 - 1 = 0 exactly)
                W[(j + 15) & 15] = h.template finish_b<true, true>(wnext - r, 0);
            } else {
                const bool valid = src.i < src.T - 16 * I;
                if (!valid) A = 0.0f;
                na_half = -0.5f * band32_sum_rows(src.slot ? 0.0f : A * A);
                W[(j + 15) & 15] = h.template finish_b<true, false>(wnext - r, I + d0 + 15);
            }
            request_step((j + PF) % NS, I + PF);
            TA_BAND32_WAIT(2 * (PF - 1));
            anext = ring[128 * ((j + 1) % NS) + lane];
            wnext = ring[128 * ((j + 1) % NS) + 64 + lane];
#pragma unroll
            for (int d = 0; d < 16; ++d) acc[d] = TA_BAND32_MFMA(A, W[(j + d) & 15], acc[d]);
            if (!bulk) {  // the window reaches the end of the series: -nA[m]/2 only where frame n exists
                const float A2 = src.slot ? na_half : 0.0f;
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    const float B2 = (src.slot && src.i < src.T - 16 * (I + d0 + d)) ? 1.0f : 0.0f;
                    acc[d] = TA_BAND32_MFMA(A2, B2, acc[d]);
                }
            }
            if (++I == nsteps) {
                more = false;
                break;
            }
        }
    }
    TA_BAND32_WAIT(0);
    na += (double)na32;
}

// P32: pair-major float32 product slab, dim = 3.  bp_am[particle * ld_am + lag] = factor * sum_i sum_d (dP)^2 / (T - lag),
// lag 0: exactly 0.  grid: any number of workgroups of 64 NW threads.
template <int NW, int PF, int NS>
__global__ void __launch_bounds__(64 * NW)
    k_band32_bp(const float* __restrict__ P32, long pitch, int T, long n_atoms, double factor, double* __restrict__ bp_am, long ld_am) {
    __shared__ float diag[NW][32 * 17 + 16 * 32];
    __shared__ double na_half[NW][16];
    __shared__ float rings[NW][NS * 128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kk = lane >> 4;
    Band32Diag sums;
    sums.init(diag[wave], lane);
    const int nblk = (T + 15) / 16, n_groups = (nblk + 14) / 15;
    const long n_units = n_atoms * n_groups, n_waves = (long)gridDim.x * NW;
    for (long u = (long)blockIdx.x * NW + wave; u < n_units; u += n_waves) {
        // (uniform per wave; readfirstlane tells the compiler)
        const long atom = __builtin_amdgcn_readfirstlane((int)(u / n_groups));
        const int g = __builtin_amdgcn_readfirstlane((int)(u - atom * n_groups)), d0 = 15 * g;
        const long c0 = 3 * atom, p0 = c0 >> 1;  // first column, first pair
        const unsigned long long base = reinterpret_cast<unsigned long long>(P32 + p0 * pitch * 2);
        BandSrc32s src;
        src.rs = band_u4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu)),
                         (unsigned)(2 * pitch) * 8u, 0x00020000u};
        const long c = c0 + kk;  // this lane group's column
        src.lane_off = kk == 3 ? 0x80000000u : (unsigned)(((c >> 1) - p0) * pitch + (lane & 15)) * 8u + (unsigned)(c & 1) * 4u;
        src.T = T;
        src.i = lane & 15;
        src.slot = kk == 3;
        band_f4 acc[16];
        double na = 0.0;
        sums.clear();
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[d] = band_f4{0.0f, 0.0f, 0.0f, 0.0f};
        band32_unit_bp<PF, NS>(src, rings[wave], d0, nblk - d0, acc, sums, na);
        sums.flush(acc);
        {
            const double t0 = band_sum_rows(na);
            if (lane < 16) na_half[wave][lane] = 0.5 * t0;
            __builtin_amdgcn_wave_barrier();
        }
        auto corr = [&](int e) {
            const int m_lo = e < 0 ? -e : 0, m_hi = e > 0 ? 16 - e : 16;
            double cc = 0.0;
            for (int m = m_lo; m < m_hi; ++m) cc += na_half[wave][m];
            return cc;
        };
        double* out = bp_am + atom * ld_am;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int off = lane + 64 * k - 15;  // lag 240 g + off, complete for 0 <= off < 240
            const long lag = 240L * g + off;
            if (off >= 0 && off < 240 && lag < T) {
                const int d = off >> 4, e = off & 15;
                double cc = corr(e);
                if (e >= 1) cc += corr(e - 16);
                out[lag] = lag == 0 ? 0.0 : factor * -2.0 * (sums.s[k] - cc) / (double)(T - lag);
                (void)d;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}


}  // namespace ta
