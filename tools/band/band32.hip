// band32.hip — launcher of the FP32 matrix-core evaluation of the Einstein-Helfand lag sums (band32_kernels.hpp):
// BASELINE configs[4]'s float32 path without the by-particle array
// (/root/reference/transport_analysis/viscosity.py:201-233 summed over particles, P rounded once to float32).
// The cut of the band is band.hip's (same wave slots per XCD: 8 waves per workgroup, one workgroup per CU).
#include "band32_kernels.hpp"

#include "../../include/ta_hip.h"
#include "../../transport_analysis_amd/csrc/ta_internal.hpp"
#include "band_tools.hpp"

namespace ta {

// lagsum[k] = factor * sum over columns and origins of (P32[i, c] - P32[i + k, c])^2 / (n_frames - k), lagsum[0] = 0
hipError_t launch_band32_lags(BandCache** cache, int n_cu, const float* pm32, long pitch, int T, long n_cols, double factor,
                              double* lagsum, hipStream_t st) {
    constexpr int kLabels = 8, kWaves = 8;
    int nwg = 0, n_ph = 0, n_pieces = 0, per_phase = 0, n_groups = 0;
    const BandPiece* pieces = nullptr;
    const int *slot_begin = nullptr, *slot_pieces = nullptr, *group_begin = nullptr;
    double* partial = nullptr;
    hipError_t e = band_tables(cache, n_cu, T, st, &nwg, &n_ph, &n_pieces, &per_phase, &n_groups, &pieces, &slot_begin, &slot_pieces,
                               &group_begin, &partial);
    if (e != hipSuccess) return e;
    static_assert(kWaves == 8, "band_tables cuts the band for 8 wave slots per workgroup");
    hipLaunchKernelGGL((k_band32_lags<kWaves, 2, 4>), dim3(nwg), dim3(64 * kWaves), 0, st, pm32, pitch, T, (n_cols + 1) / 2, kLabels,
                       n_ph, pieces, n_pieces, slot_begin, slot_pieces, partial, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(k_band_gather, dim3((T + 255) / 256), dim3(256), 0, st, partial, kLabels, n_pieces, n_ph, per_phase, group_begin,
                       n_groups, T, -2.0 * factor, 1, lagsum);
    return hipGetLastError();
}

// the float32 option WITH the by-particle array, dim = 3: bp_am[particle * ld_am + lag] (atom-major scratch), every lag written
hipError_t launch_band32_bp(int n_cu, const float* pm32, long pitch, int T, long n_atoms, double factor, double* bp_am, long ld_am,
                            hipStream_t st) {
    constexpr int kWaves = 12;  // three per SIMD (156 registers; harness: 8 / 12 / 16 waves 423 / 389 / 388 ms, 16 with scratch)
    hipLaunchKernelGGL((k_band32_bp<kWaves, 2, 4>), dim3(std::max(1, n_cu)), dim3(64 * kWaves), 0, st, pm32, pitch, T, n_atoms, factor,
                       bp_am, ld_am);
    return hipGetLastError();
}

}  // namespace ta
