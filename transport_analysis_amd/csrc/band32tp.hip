// band32tp.hip — launchers of the time-packed FP32 matrix-core evaluation of the Einstein-Helfand squared differences
// (band32tp_kernels.hpp): BASELINE configs[4]'s float32 path, with and without results.visc_by_particle
// (/root/reference/transport_analysis/viscosity.py:201-233, P rounded once to float32).
#include "band32tp_kernels.hpp"

#include <algorithm>

#include "../../include/ta_hip.h"
#include "ta_internal.hpp"

namespace ta {

// P32 = float32((m v) x), pair-major 8-byte rows; vel / pos: pair-major float64 or (src_f32) float32 slabs
hipError_t launch_helfand_product32(const void* vel, const void* pos, bool src_f32, const double* masses, long pitch, long T,
                                    long n_cols, int D, float* P32, hipStream_t st) {
    const long n_pairs = (n_cols + 1) / 2;
    // grid.x covers the rows two at a time, grid.y walks the pairs
    const unsigned gx = (unsigned)std::max<long>(1, std::min<long>(64, (T / 2 + 255) / 256));
    const unsigned gy = (unsigned)std::max<long>(1, std::min<long>(n_pairs, 65535));
    if (src_f32)
        hipLaunchKernelGGL(k_helfand_product32<float>, dim3(gx, gy), dim3(256), 0, st, (const float*)vel, (const float*)pos, masses,
                           pitch, T, n_cols, D, P32);
    else
        hipLaunchKernelGGL(k_helfand_product32<double>, dim3(gx, gy), dim3(256), 0, st, (const double*)vel, (const double*)pos,
                           masses, pitch, T, n_cols, D, P32);
    return hipGetLastError();
}


namespace {
constexpr int kWaves32tp = 12;  // three per SIMD (168 registers: 8 / 12 waves 245 / 238 ms; the rings and flush images of 12 waves take 158 KiB of LDS)

template <bool LAGS>
hipError_t launch_tp(int n_cu, const float* P32, long pitch, int T, long n_atoms, int D, double factor, double* bp_am, long ld_am,
                     unsigned long long* next_unit, int per_unit, double* partial, hipStream_t st) {
    const dim3 grid(std::max(1, n_cu)), block(64 * kWaves32tp);
    if (D == 1) hipLaunchKernelGGL((k_band32_tp<1, kWaves32tp, LAGS>), grid, block, 0, st, P32, pitch, T, n_atoms, factor, bp_am, ld_am, next_unit, per_unit, partial);
    else if (D == 2) hipLaunchKernelGGL((k_band32_tp<2, kWaves32tp, LAGS>), grid, block, 0, st, P32, pitch, T, n_atoms, factor, bp_am, ld_am, next_unit, per_unit, partial);
    else if (D == 3) hipLaunchKernelGGL((k_band32_tp<3, kWaves32tp, LAGS>), grid, block, 0, st, P32, pitch, T, n_atoms, factor, bp_am, ld_am, next_unit, per_unit, partial);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
}  // namespace

// the float32 option WITH the by-particle array: bp_am[particle * ld_am + lag] (atom-major scratch, zeroed here), every lag written
hipError_t launch_band32_tp_bp(int n_cu, const float* P32, long pitch, int T, long n_atoms, int D, double factor, double* bp_am, long ld_am,
                               unsigned long long* next_unit, hipStream_t st) {
    hipError_t e = hipMemsetAsync(bp_am, 0, sizeof(double) * (size_t)n_atoms * (size_t)ld_am, st);
    if (e == hipSuccess) e = hipMemsetAsync(next_unit, 0, sizeof(unsigned long long) * kBpCounters, st);
    if (e != hipSuccess) return e;
    return launch_tp<false>(n_cu, P32, pitch, T, n_atoms, D, factor, bp_am, ld_am, next_unit, 1, nullptr, st);
}

// ... and its lag sums alone; partial: band_bp_helf_partial_doubles() doubles of scratch (the same units as the float64 form)
hipError_t launch_band32_tp_lags(int n_cu, const float* P32, long pitch, int T, long n_atoms, int D, double factor, double* partial,
                                 unsigned long long* next_unit, double* lagsum, hipStream_t st) {
    const int per = band_bp_helf_block(n_cu, T, n_atoms), n_groups = ((T + 15) / 16 + 15) / 16;
    const long n_pb = (n_atoms + per - 1) / per;
    hipError_t e = hipMemsetAsync(next_unit, 0, sizeof(unsigned long long) * kBpCounters, st);
    if (e != hipSuccess) return e;
    e = launch_tp<true>(n_cu, P32, pitch, T, n_atoms, D, factor, nullptr, 0, next_unit, per, partial, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bandbp_gather, dim3((T + 15) / 16), dim3(256), 0, st, partial, n_pb, n_groups, T, factor, 1, lagsum);
    return hipGetLastError();
}

}  // namespace ta
