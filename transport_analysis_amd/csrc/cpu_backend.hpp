// cpu_backend.hpp — the opt-in CPU backend (cpu_backend.cpp) as api.hip sees it
#pragma once
#include <cstdint>
#include <vector>

namespace ta {
namespace cpu {

struct State {
    int64_t T = 0, A = 0;
    int D = 0, dtype = 0, threads = 1;
    std::vector<void*> slabs;  // (n_frames, n_atoms, dim) host slabs, float32 or float64 elements (owned by the context)
};

bool supported();        // the build targets AVX2 + FMA
int hardware_threads();  // OpenMP's default team size
// slab `slab` = columns [col_offset, col_offset + n_atoms dim) of the synthetic tensor of ta_stage_synth
void synth(const State& s, int slab, unsigned long long seed, int64_t col_offset, int64_t n_cols_total);
// timeseries: (n_frames,) SUMS over atoms; by_particle: (n_frames, n_atoms) or NULL.  Return TA_OK / TA_E_NOMEM.
int vacf_fft(const State& s, double* timeseries, double* by_particle);
int vacf_direct(const State& s, double* timeseries, double* by_particle);
int helfand(const State& s, const double* masses, double scale, double* timeseries, double* by_particle);

}  // namespace cpu
}  // namespace ta
