// ta_internal.hpp — shared between the translation units of libta_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/ta_hip.h"
#include "fft_engine.hpp"

struct ta_ctx;

namespace ta {

// No exception leaves an extern "C" function (every call returns an int status).  Every entry point's
// body runs inside guard(report, body): std::bad_alloc becomes TA_E_NOMEM, anything else TA_E_HIP, the message goes
// through `report(code, text)` -- the file's fail / gfail, which records it for ta_last_error and returns the code
// (and must not throw itself: they swallow a failing string copy).
template <class Report, class Body>
int guard(Report&& report, Body&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try {
            return report(TA_E_NOMEM, "out of host memory (std::bad_alloc inside the library)");
        } catch (...) {
            return TA_E_NOMEM;
        }
    } catch (const std::exception& e) {
        try {
            return report(TA_E_HIP, std::string("internal error: ") + e.what());
        } catch (...) {
            return TA_E_HIP;
        }
    } catch (...) {
        try {
            return report(TA_E_HIP, "internal error: unknown exception");
        } catch (...) {
            return TA_E_HIP;
        }
    }
}

// api.hip, for group.hip (several contexts driven by one host thread): one context's share of a
// host-facing call queued on its streams (which: 0 FFT VACF, 1 windowed VACF, 2 Helfand), the sum
// over its atoms left on the device; host_wait blocks until that work and its copies are done
int host_launch(ta_ctx* ctx, int which, const double* h_masses, double scale, double* h_bp, int64_t ld_host,
                double** d_total);
int host_wait(ta_ctx* ctx);
hipStream_t ctx_stream(ta_ctx* ctx);
int ctx_device(const ta_ctx* ctx);
int64_t ctx_staged_frames(const ta_ctx* ctx);
int ctx_fail(ta_ctx* ctx, int code, const std::string& msg);  // records the message, returns code
void host_zero(void* p, size_t bytes);  // stage_host.hip: memset 0 on several threads, streaming stores
// stage_host.hip: an anonymous, zero-filled, huge-page backed mapping that is page-locked (hipHostRegister) chunk by
// chunk on demand; copies out of it must not span two chunks
struct HostBlock {
    static constexpr size_t kChunk = (size_t)64 << 20;
    char* base = nullptr;
    size_t bytes = 0;               // mapped length: a multiple of kChunk
    std::vector<unsigned char> locked;  // per chunk
};
int host_block_map(size_t bytes, HostBlock* b);                    // 0, or -1 when the mapping fails
hipError_t host_block_lock(HostBlock& b, size_t b0, size_t b1);    // page-lock the chunks covering [b0, b1)
void host_block_unmap(HostBlock& b);

// direct.hip
// vel / pos: pair-major slabs (layout.hip) of `pitch` rows per pair, float64 or (src_f32, with the
// float32 arithmetic path only) float32 elements
hipError_t launch_direct(int mode, bool f32, bool src_f32, int L, const void* vel, const void* pos,
                         const double* masses, long pitch, int T, long n_atoms, int D,
                         double scale, double* by_particle, long ld_bp, double* ts_partial, int nwg,
                         int nt, size_t lds_bytes, void* stage_buf, int gnt, hipStream_t st);
bool direct_chunk_supported(int L);       // lags per chunk compiled in (8, 10)
size_t direct_lds_bytes(int T, bool f32, int L);
int direct_max_wg_per_cu(int mode, bool f32, int L, int nt, size_t lds_bytes, bool global_stage);

// band32tp.hip: the float32 product slab of the Einstein-Helfand float32 option: P32 = float32((m v) x) in a pair-major float32
// slab (vel / pos: pair-major float64 slabs, or float32 ones with src_f32)
hipError_t launch_helfand_product32(const void* vel, const void* pos, bool src_f32, const double* masses, long pitch, long T,
                                    long n_cols, int D, float* P32, hipStream_t st);

// short.hip: trajectories of up to short_max_frames() frames, a lane per column (short_kernels.hpp); bp (n_frames, ld_bp) or
// NULL; partial [nwg * short_waves()][T] lag sums per wave (normalised); factor 1 (VACF) or scale / D (Helfand)
int short_max_frames();
int short_waves();
int short_grid(int n_cu, int mode, int T, long n_atoms, int D, bool by_particle);
hipError_t launch_short(int mode, int nwg, const double* vel, const double* pos, const double* masses, long pitch, int T,
                        long n_atoms, int D, double factor, double* bp, long ld_bp, double* partial, hipStream_t st);

// mid.hip: 65 ... mid_max_frames() frames, a lane per (column, pair of 16-lag blocks) walking a sliding window (mid_kernels.hpp);
// bp (n_frames, ld_bp) or NULL; partial [nwg][T]; factor 1 (VACF) or scale / D (Helfand)
struct MidShape {
    int ncl_log2, nc, ts, threads;
    size_t lds;
};
int mid_max_frames();
MidShape mid_shape(int T, int D, int ncl_log2 = 0);  // ncl_log2 3..6: lanes per pair of blocks forced (0: by n_frames)
int mid_grid(int n_cu, int mode, int T, long n_atoms, int D, int ncl_log2);
hipError_t launch_mid(int mode, int nwg, const double* vel, const double* pos, const double* masses, long pitch, int T,
                      long n_atoms, int D, double factor, double* bp, long ld_bp, double* partial, int ncl_log2, hipStream_t st);

hipError_t launch_row_sums(const double* bp, long n_rows, long n_cols, long ld, double* out,
                           hipStream_t st);
// bandbp.hip: the windowed VACF with its by-particle array on the FP64 matrix cores (atom-major scratch, zeroed inside)
hipError_t launch_band_bp_vacf(int n_cu, const double* pm, long pitch, int T, long n_atoms, int D, double* bp_am, long ld_am,
                               unsigned long long* next_unit, hipStream_t st);
hipError_t launch_band_bp_vacf_lags(int n_cu, const double* pm, long pitch, int T, long n_atoms, int D, double* partial,
                                    unsigned long long* next_unit, double* lagsum, hipStream_t st);
hipError_t launch_band_bp_helf(int n_cu, const double* P, long pitch, int T, long n_atoms, int D, double factor, double* bp_am,
                               long ld_am, unsigned long long* next_unit, hipStream_t st);
int band_bp_helf_block(int n_cu, int T, long n_atoms);
size_t band_bp_helf_partial_doubles(int n_cu, int T, long n_atoms);
// band32tp.hip: the float32 option's Helfand forms, k-slots from the time axis
hipError_t launch_band32_tp_bp(int n_cu, const float* P32, long pitch, int T, long n_atoms, int D, double factor, double* bp_am, long ld_am,
                               unsigned long long* next_unit, hipStream_t st);
hipError_t launch_band32_tp_lags(int n_cu, const float* P32, long pitch, int T, long n_atoms, int D, double factor, double* partial,
                                 unsigned long long* next_unit, double* lagsum, hipStream_t st);
hipError_t launch_band_bp_helf_lags(int n_cu, const double* P, long pitch, int T, long n_atoms, int D, double factor, double* partial,
                                    unsigned long long* next_unit, double* lagsum, hipStream_t st);
hipError_t launch_sum_partials(const double* partial, int n_parts, long n, double* out,
                               hipStream_t st);
// helfand_fft.hip: optional FFT evaluation of the Helfand lag sums
// pair-major slabs in, product slab P out in the same layout; Qpart [n_parts][T] written in full
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
// (dst_f32 / pm_f32: the slab holds float32 elements)
hipError_t launch_relayout(const void* src, bool src_f32, long ld_row, long n_cols, long t_count,
                           void* dst, bool dst_f32, long pitch, long t_dst0, hipStream_t st);
hipError_t launch_unlayout(const void* pm, bool pm_f32, long pitch, long n_cols, long t_count, double* dst,
                           long ld_row, hipStream_t st);
// atom-major by-particle scratch -> (n_frames, ld_bp); partial: [ceil(n_atoms/64)][T] or NULL
hipError_t launch_bp_transpose(const double* src, long src_ld, long n_atoms, long T, double* bp, long ld_bp,
                               double* partial, hipStream_t st);
hipError_t launch_synth(void* pm, bool pm_f32, long pitch, long n_cols, long T, unsigned long long seed,
                        long col_offset, long n_cols_total, hipStream_t st);

// wfft.hip: FFT evaluation on pair-major slabs, padded length L = 2 R R0 512 (wfft.hpp)
bool wfft_choose(long n_frames, int* R0, int* R);  // smallest R R0 512 >= n_frames
size_t wfft_table_elems(int R0, int R);
void wfft_fill_table(int R0, int R, cd* table);
int wfft_max_wg_per_cu(int R0);
int wfft_threads(int R0);
// forward kernel (R0 > 1), nwg a multiple of 16 R: lag-sum mode n_units column pairs ->
// accg [nwg / 2R][L] partial spectra; by-particle mode n_units atoms -> accg [n_units][L]
// src_f32: pm points at a float32 slab (8-byte rows); R = 1 only
hipError_t launch_wfft_forward(int R0, int R, bool by_particle, bool src_f32, int nwg, hipStream_t st, const double* pm,
                               long pitch, int T, long n_units, int D, const cd* tw, double* accg);
// the lag-sum forward kernel built with in-kernel clock stamps (plans R0 = 8, 10, 12, 16, 20 without an
// outer radix): stamps[16 * workgroup + ...], see wfft.hpp (ta_clock_probe)
hipError_t launch_wfft_forward_stamp(int R0, int nwg, hipStream_t st, const double* pm, long pitch, int T,
                                     long n_units, const cd* tw, double* accg, unsigned long long* stamps);
// inverse kernel (R0 > 1): lag values of n_items spectra, out[item * ld + lag]
hipError_t launch_wfft_inverse(int R0, int R, int nwg, hipStream_t st, const double* spec, int T, long n_items,
                               const cd* tw, double* out, long ld, int prefetch);
// n_frames <= 512 (R0 = 1): lag-sum kernel (accg [4 nwg][1024], natural bin order) with its
// finish (sum, cosine sums), and the fused by-particle kernel (atom-major out[atom * ld + lag])
hipError_t launch_w1_accum(int nwg, hipStream_t st, const double* pm, long pitch, int T, long n_pairs,
                           const cd* tw, double* accg);
hipError_t launch_w1_bp(int nwg, hipStream_t st, const double* pm, long pitch, int T, long n_atoms, int D,
                        const cd* tw, double* out, long ld);
hipError_t launch_wfft_finish(int R0, const double* partial, int n_parts, const cd* tw, int T,
                              double* spec /* [2M] */, double* lagsum, hipStream_t st);

}  // namespace ta
