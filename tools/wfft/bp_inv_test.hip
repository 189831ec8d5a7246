// Timing harness of k_wbp_inverse (by-particle inverse kernel) with ablations: which part of
// an atom's 16 us is arithmetic, LDS traffic, spectrum loads, division.
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=fast bp_inv_test.hip -o bp_inv_test
#include "../../transport_analysis_amd/csrc/wfft.hpp"

#include <cstdio>
#include <vector>

using namespace ta;
#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                           \
        }                                                                       \
    } while (0)

template <int PF, int ABL>
int run(const char* name, const double* spec, int T, long n_atoms, const cd* tw, double* out, long ld, int nwg) {
    using P = WPlan<20>;
    auto kern = k_wbp_inverse<P, PF, ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::kLds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(P::NT), P::kLds, 0, spec, T, n_atoms, tw, out, ld);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %8.3f ms  %6.2f us/atom/CU\n", name, ms, ms * 1e3 / ((double)n_atoms / nwg));
    return 0;
}

int main() {
    const int R0 = 20, T = 10000;
    const long n_atoms = 16384, ld = 10000;
    std::vector<cd> tab(wf_table_elems(R0));
    wf_fill_table(R0, tab.data());
    cd* tw;
    CK(hipMalloc((void**)&tw, tab.size() * sizeof(cd)));
    CK(hipMemcpy(tw, tab.data(), tab.size() * sizeof(cd), hipMemcpyHostToDevice));
    double *spec, *out;
    const size_t ns = (size_t)n_atoms * 2 * R0 * 512;
    CK(hipMalloc((void**)&spec, ns * 8));
    CK(hipMemset(spec, 0x3c, ns * 8));
    CK(hipMalloc((void**)&out, (size_t)n_atoms * ld * 8));
    const int nwg = 256;
    if (run<0, 0>("full, no prefetch", spec, T, n_atoms, tw, out, ld, nwg)) return 1;
    run<1, 0>("prefetch 1 sub-series", spec, T, n_atoms, tw, out, ld, nwg);
    run<2, 0>("prefetch 2", spec, T, n_atoms, tw, out, ld, nwg);
    run<3, 0>("prefetch 3", spec, T, n_atoms, tw, out, ld, nwg);
    run<0, 1>("no sub-transforms", spec, T, n_atoms, tw, out, ld, nwg);
    run<0, 2>("no first-stage butterfly", spec, T, n_atoms, tw, out, ld, nwg);
    run<0, 3>("no untangling", spec, T, n_atoms, tw, out, ld, nwg);
    run<0, 4>("no division", spec, T, n_atoms, tw, out, ld, nwg);
    run<0, 5>("no spectrum loads", spec, T, n_atoms, tw, out, ld, nwg);
    run<3, 5>("no spectrum loads (pf 3)", spec, T, n_atoms, tw, out, ld, nwg);
    return 0;
}
