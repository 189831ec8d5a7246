// mid.hip — launcher of the mid-length O(T^2) kernels (mid_kernels.hpp).
#include <algorithm>

#include "mid_kernels.hpp"
#include "ta_internal.hpp"

namespace ta {

int mid_max_frames() { return 512; }

// launch shape for n_frames: lanes per pair of lag blocks (a power of two), columns per tile, LDS stride, threads, LDS bytes
MidShape mid_shape(int T, int D, int ncl_log2) {
    MidShape s;
    s.ncl_log2 = T <= 128 ? 6 : T <= 256 ? 5 : 4;
    if (ncl_log2 >= 3 && ncl_log2 <= 6) s.ncl_log2 = ncl_log2;  // ("mid_ncl": tools/mid_shapes.py)
    const int ncl = 1 << s.ncl_log2;
    s.nc = ncl / D * D;
    const int nb = (T + kMidLB - 1) / kMidLB, np = (nb + 1) / 2;
    s.ts = (T + kMidLB - 1) / kMidLB * kMidLB + kMidLB + 2;
    s.threads = (ncl * np + 63) / 64 * 64;
    s.lds = sizeof(double) * ((size_t)ncl * s.ts + T);
    return s;
}

static const void* mid_kernel(int mode) {
    return mode == MODE_VACF ? reinterpret_cast<const void*>(k_mid<MODE_VACF>) : reinterpret_cast<const void*>(k_mid<MODE_HELFAND>);
}

int mid_grid(int n_cu, int mode, int T, long n_atoms, int D, int ncl_log2) {
    const MidShape s = mid_shape(T, D, ncl_log2);
    const void* fn = mid_kernel(mode);
    int per_cu = 0;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, s.threads, s.lds) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    const long n_tiles = (n_atoms * D + s.nc - 1) / s.nc;
    return (int)std::max<long>(1, std::min<long>((long)n_cu * per_cu, n_tiles));
}

hipError_t launch_mid(int mode, int nwg, const double* vel, const double* pos, const double* masses, long pitch, int T,
                      long n_atoms, int D, double factor, double* bp, long ld_bp, double* partial, int ncl_log2, hipStream_t st) {
    if (T < 1 || T > mid_max_frames() || D < 1 || D > 3) return hipErrorInvalidValue;
    const MidShape s = mid_shape(T, D, ncl_log2);
    if (s.threads > kMidThreads || 2 * s.threads < T) return hipErrorInvalidValue;
    const void* fn = mid_kernel(mode);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.lds);
    if (e != hipSuccess) return e;
    int nc = s.nc, lanes_log2 = s.ncl_log2, ts = s.ts;
    void* args[] = {&vel, &pos, &masses, &pitch, &T, &n_atoms, &D, &nc, &lanes_log2, &ts, &factor, &bp, &ld_bp, &partial};
    return hipLaunchKernel(fn, dim3(nwg), dim3(s.threads), args, s.lds, st);
}

}  // namespace ta
