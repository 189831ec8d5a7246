// bandbp.hip — launchers of the FP64 matrix-core evaluation of the O(T^2) correlators WITH their by-particle arrays
// (bandbp_kernels.hpp): the class-default outputs of VelocityAutocorr._conclude_simple
// (/root/reference/transport_analysis/velocityautocorr.py:217-238) and ViscosityHelfand._conclude (viscosity.py:201-233).
#include "bandbp_kernels.hpp"

#include "../../include/ta_hip.h"
#include "ta_internal.hpp"

namespace ta {

// bp_am[particle * ld_am + lag] = sum_{i, d} v[i, particle, d] v[i + lag, particle, d] / (n_frames - lag) (atom-major scratch of
// n_atoms * ld_am doubles, zeroed here: two units add their halves of some lags); next_unit: 8 bytes of device memory
hipError_t launch_band_bp_vacf(int n_cu, const double* pm, long pitch, int T, long n_atoms, int D, double* bp_am, long ld_am,
                               unsigned long long* next_unit, hipStream_t st) {
    constexpr int kWaves = 8;  // two per SIMD (194 registers; the rings of 8 waves take 111 KiB of the CU's LDS)
    hipError_t e = hipMemsetAsync(bp_am, 0, sizeof(double) * (size_t)n_atoms * (size_t)ld_am, st);
    if (e == hipSuccess) e = hipMemsetAsync(next_unit, 0, sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    const dim3 grid(std::max(1, n_cu)), block(64 * kWaves);
    if (D == 1) hipLaunchKernelGGL((k_band_bp_vacf<1, kWaves>), grid, block, 0, st, pm, pitch, T, n_atoms, bp_am, ld_am, next_unit);
    else if (D == 2) hipLaunchKernelGGL((k_band_bp_vacf<2, kWaves>), grid, block, 0, st, pm, pitch, T, n_atoms, bp_am, ld_am, next_unit);
    else if (D == 3) hipLaunchKernelGGL((k_band_bp_vacf<3, kWaves>), grid, block, 0, st, pm, pitch, T, n_atoms, bp_am, ld_am, next_unit);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// the Einstein-Helfand form on the float64 product slab P: bp_am[particle * ld_am + lag] = factor * sum (dP)^2 / (n_frames - lag)
hipError_t launch_band_bp_helf(int n_cu, const double* P, long pitch, int T, long n_atoms, int D, double factor, double* bp_am,
                               long ld_am, unsigned long long* next_unit, hipStream_t st) {
    constexpr int kWaves = 8;  // two per SIMD; four rings per wave (three centred columns and their norms): 147 KiB of LDS
    hipError_t e = hipMemsetAsync(bp_am, 0, sizeof(double) * (size_t)n_atoms * (size_t)ld_am, st);
    if (e == hipSuccess) e = hipMemsetAsync(next_unit, 0, sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    const dim3 grid(std::max(1, n_cu)), block(64 * kWaves);
    if (D == 1) hipLaunchKernelGGL((k_band_bp_helf<1, kWaves>), grid, block, 0, st, P, pitch, T, n_atoms, factor, bp_am, ld_am, next_unit);
    else if (D == 2) hipLaunchKernelGGL((k_band_bp_helf<2, kWaves>), grid, block, 0, st, P, pitch, T, n_atoms, factor, bp_am, ld_am, next_unit);
    else if (D == 3) hipLaunchKernelGGL((k_band_bp_helf<3, kWaves>), grid, block, 0, st, P, pitch, T, n_atoms, factor, bp_am, ld_am, next_unit);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace ta
