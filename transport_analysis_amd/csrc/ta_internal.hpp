// ta_internal.hpp — shared between the translation units of libta_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "fft_engine.hpp"

namespace ta {

struct FftArgs {        // k_fft_finalize
    int T;
    const cd* tw2;        // [0,2M): W_{2M}^n; [2M,3M): first-stage table [q][u] = W_M^{q u}; ...
    const double* spec;   // [n_slices][2][M], the plan's digit-reversed bin order
    int n_slices;
    double* lagsum;       // [T]
};

// An on-chip plan (M = 2^a or 5*2^a, 16..10240): today only its inverse transform is used
// (k_fft_finalize: ONE per launch) plus, for M = 8192 / 10240, the outer-radix path's stages.
struct PlanEntry {
    int M, NT, S;
    int R_first;                          // first radix: layout of the first-stage twiddle table
    size_t lds_bytes;
    hipError_t (*finalize)(hipStream_t st, const FftArgs& a);
    void (*perm)(std::vector<int>& out);  // output position -> frequency (digit reversal)
};

const std::vector<PlanEntry>& plans_pow2();
const std::vector<PlanEntry>& plans_five();

// direct.hip
// vel / pos: pair-major slabs (layout.hip) of `pitch` rows per pair
hipError_t launch_direct(int mode, bool f32, int L, const double* vel, const double* pos,
                         const double* masses, long pitch, int T, long n_atoms, int D,
                         double scale, double* by_particle, long ld_bp, double* ts_partial, int nwg,
                         int nt, size_t lds_bytes, void* stage_buf, int gnt, hipStream_t st);
bool direct_chunk_supported(int L);       // lags per chunk compiled in (8, 10)
size_t direct_lds_bytes(int T, bool f32, int L);
int direct_max_wg_per_cu(int mode, bool f32, int L, int nt, size_t lds_bytes, bool global_stage);

hipError_t launch_row_sums(const double* bp, long n_rows, long n_cols, long ld, double* out,
                           hipStream_t st);
hipError_t launch_sum_partials(const double* partial, int n_parts, long n, double* out,
                               hipStream_t st);
// fft_long.hip: FFT lag sums beyond the on-chip transform length (timeseries path)
bool fft_long_choose(long n_frames, int* M, int* Rout);  // smallest M' = Rout*M >= n_frames
void fft_long_perm(int M, std::vector<int>& perm);       // position -> frequency of plan M's output
size_t fft_long_acc_block(int M);                        // doubles per workgroup and pass
hipError_t launch_fft_long_accum(int M, int nwg, hipStream_t st, const double* vel, long ld_row, long pair_stride, int T,
                                 long n_cols, int Rout, const cd* tw2, const cd* twL, double* accg,
                                 cd* scratch /* [nwg][4][2*Rout][M] */);
hipError_t launch_fft_long_finish(int M, int Rout, const double* partial, int n_parts, const int* perm,
                                  const cd* twL, int T, double* spec, double* lagsum, hipStream_t st);

// helfand_fft.hip: optional FFT evaluation of the Helfand lag sums
// pair-major slabs in, product slab P out in the same layout; Qpart [n_parts][T] zeroed by caller
hipError_t launch_helfand_product(const double* vel, const double* pos, const double* masses,
                                  long pitch, long T, long n_cols, int D, double* P, double* Qpart,
                                  int n_parts, hipStream_t st);
hipError_t launch_helfand_combine(const double* Q, const double* s2n, double* C, int T, double factor,
                                  double* out, hipStream_t st);
hipError_t launch_helfand_product_bp(const double* vel, const double* pos, const double* masses,
                                     long pitch, long T, long n_atoms, int D, double* P, double* Ca,
                                     hipStream_t st);  // Ca: (T+1, n_atoms)
hipError_t launch_helfand_combine_bp(double* Ca, long n_atoms, int T, double factor, double* bp,
                                     long ld_bp, hipStream_t st);

hipError_t launch_widen_f32(const float* in, double* out, long n, hipStream_t st);

// layout.hip: frame-major (n_frames, ld_row) float32/float64 rows -> pair-major slab rows
hipError_t launch_relayout(const void* src, bool src_f32, long ld_row, long n_cols, long t_count,
                           double* dst, long pitch, long t_dst0, hipStream_t st);
hipError_t launch_unlayout(const double* pm, long pitch, long n_cols, long t_count, double* dst,
                           long ld_row, hipStream_t st);
// atom-major by-particle scratch -> (n_frames, ld_bp); partial: [ceil(n_atoms/64)][T] or NULL
hipError_t launch_bp_transpose(const double* src, long src_ld, long n_atoms, long T, double* bp, long ld_bp,
                               double* partial, hipStream_t st);
hipError_t launch_synth(double* pm, long pitch, long n_cols, long T, unsigned long long seed,
                        long col_offset, long n_cols_total, hipStream_t st);

// wfft.hip: power-spectrum accumulation on pair-major slabs, M = R0 * 512
bool wfft_choose(long n_frames, int* R0);
size_t wfft_table_elems(int R0);
void wfft_fill_table(int R0, cd* table);
int wfft_max_wg_per_cu(int R0);
hipError_t launch_wfft_accum(int R0, int nwg, hipStream_t st, const double* pm, long pitch, int T,
                             long n_pairs, const cd* tw, double* accg /* [nwg][2M], natural order */);
// by-particle mode (k_wbp): lag values of every atom, atom-major out[atom * ld + lag]
hipError_t launch_wfft_by_particle(int R0, int nwg, hipStream_t st, const double* pm, long pitch, int T,
                                   long n_atoms, int D, const cd* tw, double* out, long ld);
// pass-split form: nwg a multiple of 16, accg [nwg/2][2M] (every element written by the launch)
// two-kernel by-particle evaluation: spec holds n_atoms * 2 * R0 * 512 doubles of scratch
hipError_t launch_wfft_by_particle2(int R0, int nwg_fwd, int nwg_inv, hipStream_t st, const double* pm,
                                    long pitch, int T, long n_atoms, int D, const cd* tw, double* spec,
                                    double* out, long ld, int prefetch);
hipError_t launch_wfft_split(int R0, int nwg, hipStream_t st, const double* pm, long pitch, int T,
                             long n_pairs, const cd* tw, double* accg);
hipError_t launch_wfft_finish(int R0, const double* partial, int n_parts, const cd* tw, int T,
                              double* spec /* [2M] */, double* lagsum, hipStream_t st);
// spec[pass*M + p] = sum over workgroups of bin 2*perm[p] + pass: the natural-order blocks summed
// into the [2][M] digit-reversed layout that the on-chip plan's k_fft_finalize consumes
hipError_t launch_wfft_sum_perm(const double* partial, int n_parts, int M, const int* perm, double* spec,
                                hipStream_t st);

}  // namespace ta
