// short.hip — launcher of the short-trajectory kernels (short_kernels.hpp).
#include <algorithm>

#include "short_kernels.hpp"
#include "ta_internal.hpp"

namespace ta {

namespace {
struct ShortFn {
    const void* fn;
    size_t lds;
};
template <int TMAX, bool BP>
ShortFn short_kernel(int mode, int D) {
#define TA_K(M, DD) ShortFn{reinterpret_cast<const void*>(k_short<TMAX, M, DD, BP>), ShortCfg<TMAX, BP>::kLds}
    if (mode == MODE_VACF) return D == 1 ? TA_K(MODE_VACF, 1) : D == 2 ? TA_K(MODE_VACF, 2) : TA_K(MODE_VACF, 3);
    return D == 1 ? TA_K(MODE_HELFAND, 1) : D == 2 ? TA_K(MODE_HELFAND, 2) : TA_K(MODE_HELFAND, 3);
#undef TA_K
}
ShortFn short_pick(int mode, int T, int D, bool bp) {
    if (T <= 32) return bp ? short_kernel<32, true>(mode, D) : short_kernel<32, false>(mode, D);
    return bp ? short_kernel<64, true>(mode, D) : short_kernel<64, false>(mode, D);
}
}  // namespace

int short_max_frames() { return 64; }
int short_waves() { return kShortWaves; }

int short_grid(int n_cu, int mode, int T, long n_atoms, int D, bool by_particle) {
    const ShortFn k = short_pick(mode, T, D, by_particle);
    int per_cu = 0;
    (void)hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k.fn, 64 * kShortWaves, k.lds) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    const long aw = (D == 3 ? 63 : 64) / D, n_tiles = (n_atoms + aw - 1) / aw;
    const long want = (n_tiles + kShortWaves - 1) / kShortWaves;
    return (int)std::max<long>(1, std::min<long>((long)n_cu * per_cu, want));
}

hipError_t launch_short(int mode, int nwg, const double* vel, const double* pos, const double* masses, long pitch, int T,
                        long n_atoms, int D, double factor, double* bp, long ld_bp, double* partial, hipStream_t st) {
    if (T < 1 || T > short_max_frames() || D < 1 || D > 3) return hipErrorInvalidValue;
    const ShortFn k = short_pick(mode, T, D, bp != nullptr);
    hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.lds);
    if (e != hipSuccess) return e;
    void* args[] = {&vel, &pos, &masses, &pitch, &T, &n_atoms, &factor, &bp, &ld_bp, &partial};
    return hipLaunchKernel(k.fn, dim3(nwg), dim3(64 * kShortWaves), args, k.lds, st);
}

}  // namespace ta
