// band_tools.hpp — launchers of the column-packed matrix-core forms (band.hip, band32.hip: tools only since round 6; the
// library's own forms fill the k-slots from the time axis, csrc/bandbp_kernels.hpp, csrc/band32tp_kernels.hpp)
#pragma once
#include <hip/hip_runtime.h>

namespace ta {

// band.hip: lag SUMS on the FP64 matrix cores (band_kernels.hpp); *cache: the plan's device copy, kept by the caller
//   windowed VACF: lagsum[k] = factor * sum over columns c and origins i of pm[i, c] pm[i + k, c] / (n_frames - k)
//   helfand (pm = the product slab P): factor * sum of (P[i, c] - P[i + k, c])^2 / (n_frames - k), lagsum[0] = 0
struct BandCache;
hipError_t launch_band_lags(BandCache** cache, int n_cu, bool helfand, const double* pm, long pitch, int T, long n_cols,
                            double factor, double* lagsum, hipStream_t st);
void band_cache_free(BandCache* c);
struct BandPiece;
hipError_t band_tables(BandCache** cache, int n_cu, int T, hipStream_t st, int* nwg, int* n_ph, int* n_pieces, int* per_phase,
                       int* n_groups, const BandPiece** pieces, const int** slot_begin, const int** slot_pieces,
                       const int** group_begin, double** partial);
// band32.hip: the Helfand lag sums on the FP32 matrix cores (band32_kernels.hpp): P32 = float32((m v) x) in a
// pair-major float32 slab (vel / pos: pair-major float64 slabs, or float32 ones with src_f32), then
//   lagsum[k] = factor * sum of (P32[i, c] - P32[i + k, c])^2 / (n_frames - k), lagsum[0] = 0
hipError_t launch_band32_bp(int n_cu, const float* pm32, long pitch, int T, long n_atoms, double factor, double* bp_am, long ld_am,
                            hipStream_t st);
hipError_t launch_band32_lags(BandCache** cache, int n_cu, const float* pm32, long pitch, int T, long n_cols, double factor,
                              double* lagsum, hipStream_t st);


}  // namespace ta
