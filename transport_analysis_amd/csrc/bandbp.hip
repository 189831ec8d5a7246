// bandbp.hip — launchers of the FP64 matrix-core evaluation of the O(T^2) correlators WITH their by-particle arrays
// (bandbp_kernels.hpp): the class-default outputs of VelocityAutocorr._conclude_simple
// (/root/reference/transport_analysis/velocityautocorr.py:217-238) and ViscosityHelfand._conclude (viscosity.py:201-233).
#include "bandbp_kernels.hpp"

#include "../../include/ta_hip.h"
#include "ta_internal.hpp"

namespace ta {

int band_bp_helf_block(int n_cu, int T, long n_atoms);

// bp_am[particle * ld_am + lag] = sum_{i, d} v[i, particle, d] v[i + lag, particle, d] / (n_frames - lag) (atom-major scratch of
// n_atoms * ld_am doubles, zeroed here: two units add their halves of some lags); next_unit: 64 bytes of device memory (one counter per XCD)
hipError_t launch_band_bp_vacf(int n_cu, const double* pm, long pitch, int T, long n_atoms, int D, double* bp_am, long ld_am,
                               unsigned long long* next_unit, hipStream_t st) {
    constexpr int kWaves = 8;  // two per SIMD (194 registers; the rings of 8 waves take 111 KiB of the CU's LDS)
    hipError_t e = hipMemsetAsync(bp_am, 0, sizeof(double) * (size_t)n_atoms * (size_t)ld_am, st);
    if (e == hipSuccess) e = hipMemsetAsync(next_unit, 0, sizeof(unsigned long long) * kBpCounters, st);
    if (e != hipSuccess) return e;
    const dim3 grid(std::max(1, n_cu)), block(64 * kWaves);
    double* none = nullptr;
    if (D == 1) hipLaunchKernelGGL((k_band_bp_vacf<1, kWaves, false>), grid, block, 0, st, pm, pitch, T, n_atoms, bp_am, ld_am, next_unit, 1, none);
    else if (D == 2) hipLaunchKernelGGL((k_band_bp_vacf<2, kWaves, false>), grid, block, 0, st, pm, pitch, T, n_atoms, bp_am, ld_am, next_unit, 1, none);
    else if (D == 3) hipLaunchKernelGGL((k_band_bp_vacf<3, kWaves, false>), grid, block, 0, st, pm, pitch, T, n_atoms, bp_am, ld_am, next_unit, 1, none);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ... and its lag sums alone (by_particle=False): lagsum[k] = sum over particles, columns and origins of v[i] v[i + k] / (n_frames - k);
// units and scratch as launch_band_bp_helf_lags
hipError_t launch_band_bp_vacf_lags(int n_cu, const double* pm, long pitch, int T, long n_atoms, int D, double* partial,
                                    unsigned long long* next_unit, double* lagsum, hipStream_t st) {
    constexpr int kWaves = 8;
    const int per = band_bp_helf_block(n_cu, T, n_atoms), n_groups = ((T + 15) / 16 + 15) / 16;
    const long n_pb = (n_atoms + per - 1) / per;
    hipError_t e = hipMemsetAsync(next_unit, 0, sizeof(unsigned long long) * kBpCounters, st);
    if (e != hipSuccess) return e;
    const dim3 grid(std::max(1, n_cu)), block(64 * kWaves);
    double* none = nullptr;
    if (D == 1) hipLaunchKernelGGL((k_band_bp_vacf<1, kWaves, true>), grid, block, 0, st, pm, pitch, T, n_atoms, none, 0L, next_unit, per, partial);
    else if (D == 2) hipLaunchKernelGGL((k_band_bp_vacf<2, kWaves, true>), grid, block, 0, st, pm, pitch, T, n_atoms, none, 0L, next_unit, per, partial);
    else if (D == 3) hipLaunchKernelGGL((k_band_bp_vacf<3, kWaves, true>), grid, block, 0, st, pm, pitch, T, n_atoms, none, 0L, next_unit, per, partial);
    else return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bandbp_gather, dim3((T + 15) / 16), dim3(256), 0, st, partial, n_pb, n_groups, T, 1.0, 0, lagsum);
    return hipGetLastError();
}

// the Einstein-Helfand form on the float64 product slab P: bp_am[particle * ld_am + lag] = factor * sum (dP)^2 / (n_frames - lag)
hipError_t launch_band_bp_helf(int n_cu, const double* P, long pitch, int T, long n_atoms, int D, double factor, double* bp_am,
                               long ld_am, unsigned long long* next_unit, hipStream_t st) {
    constexpr int kWaves = 8;  // two per SIMD; four rings per wave (three centred columns and their norms): 147 KiB of LDS
    hipError_t e = hipMemsetAsync(bp_am, 0, sizeof(double) * (size_t)n_atoms * (size_t)ld_am, st);
    if (e == hipSuccess) e = hipMemsetAsync(next_unit, 0, sizeof(unsigned long long) * kBpCounters, st);
    if (e != hipSuccess) return e;
    const dim3 grid(std::max(1, n_cu)), block(64 * kWaves);
    double* none = nullptr;
    if (D == 1) hipLaunchKernelGGL((k_band_bp_helf<1, kWaves, false>), grid, block, 0, st, P, pitch, T, n_atoms, factor, bp_am, ld_am, next_unit, 1, none);
    else if (D == 2) hipLaunchKernelGGL((k_band_bp_helf<2, kWaves, false>), grid, block, 0, st, P, pitch, T, n_atoms, factor, bp_am, ld_am, next_unit, 1, none);
    else if (D == 3) hipLaunchKernelGGL((k_band_bp_helf<3, kWaves, false>), grid, block, 0, st, P, pitch, T, n_atoms, factor, bp_am, ld_am, next_unit, 1, none);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ... and its lag sums alone (by_particle=False): lagsum[k] = factor * sum over particles, columns and origins of (dP)^2 / (n_frames - k),
// lagsum[0] = 0.  A unit = one group of 16 block lags of band_bp_helf_block() consecutive particles.
// partial: band_bp_helf_partial_doubles() doubles of scratch.
int band_bp_helf_block(int n_cu, int T, long n_atoms) {
    // particles per unit of the lag-sum forms: one epilogue per unit and 272 partial sums (at 4: 9 % of the slab's bytes);
    // measured at 5000 x 50000 x 3 / 20000 x 25000 x 3: 1: 55.0 ms, 4: 51.5 / 456, 16: 52.3 / 462 (longer units leave waves
    // idle at the end).  Fewer when there are not ~8 units per wave otherwise.
    const long n_groups = ((T + 15) / 16 + 15) / 16, want = 8L * 8 * std::max(1, n_cu);
    return (int)std::max<long>(1, std::min<long>(4, n_atoms * n_groups / want));
}
size_t band_bp_helf_partial_doubles(int n_cu, int T, long n_atoms) {
    const long per = band_bp_helf_block(n_cu, T, n_atoms), n_pb = (n_atoms + per - 1) / per, n_groups = ((T + 15) / 16 + 15) / 16;
    return (size_t)n_pb * n_groups * kBandPartial;
}
hipError_t launch_band_bp_helf_lags(int n_cu, const double* P, long pitch, int T, long n_atoms, int D, double factor, double* partial,
                                    unsigned long long* next_unit, double* lagsum, hipStream_t st) {
    constexpr int kWaves = 8;
    const int per = band_bp_helf_block(n_cu, T, n_atoms), n_groups = ((T + 15) / 16 + 15) / 16;
    const long n_pb = (n_atoms + per - 1) / per;
    hipError_t e = hipMemsetAsync(next_unit, 0, sizeof(unsigned long long) * kBpCounters, st);
    if (e != hipSuccess) return e;
    const dim3 grid(std::max(1, n_cu)), block(64 * kWaves);
    double* none = nullptr;
    if (D == 1) hipLaunchKernelGGL((k_band_bp_helf<1, kWaves, true>), grid, block, 0, st, P, pitch, T, n_atoms, factor, none, 0L, next_unit, per, partial);
    else if (D == 2) hipLaunchKernelGGL((k_band_bp_helf<2, kWaves, true>), grid, block, 0, st, P, pitch, T, n_atoms, factor, none, 0L, next_unit, per, partial);
    else if (D == 3) hipLaunchKernelGGL((k_band_bp_helf<3, kWaves, true>), grid, block, 0, st, P, pitch, T, n_atoms, factor, none, 0L, next_unit, per, partial);
    else return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bandbp_gather, dim3((T + 15) / 16), dim3(256), 0, st, partial, n_pb, n_groups, T, factor, 1, lagsum);
    return hipGetLastError();
}

}  // namespace ta
