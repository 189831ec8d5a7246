// band.hip — launcher of the FP64 matrix-core evaluation of the lag SUMS of the O(T^2) correlators
// (band_kernels.hpp): windowed VACF (/root/reference/transport_analysis/velocityautocorr.py:217-238 summed
// over particles) and, on the product slab P = (m v) x, the Einstein-Helfand mean squared differences
// (viscosity.py:201-233 summed over particles).  The cut of the band depends on n_frames and the grid only; its device copy and
// the partial-sum buffer are cached per context.
#include "band_kernels.hpp"

#include "ta_internal.hpp"

namespace ta {

struct BandCache {
    BandPlan plan;
    int nwg = 0;
    BandPiece* pieces = nullptr;
    int *slot_begin = nullptr, *slot_pieces = nullptr, *group_begin = nullptr;
    double* partial = nullptr;
};

static void band_release(BandCache* c) {
    if (!c) return;
    for (void* p : {(void*)c->pieces, (void*)c->slot_begin, (void*)c->slot_pieces, (void*)c->group_begin, (void*)c->partial})
        if (p) (void)hipFree(p);
    c->pieces = nullptr, c->slot_begin = c->slot_pieces = c->group_begin = nullptr, c->partial = nullptr;
}

void band_cache_free(BandCache* c) {
    band_release(c);
    delete c;
}

template <typename V>
static hipError_t band_upload(V** dst, const std::vector<V>& src) {
    const size_t bytes = sizeof(V) * std::max<size_t>(1, src.size());
    hipError_t e = hipMalloc((void**)dst, bytes);
    if (e != hipSuccess) return e;
    return src.empty() ? hipSuccess : hipMemcpy(*dst, src.data(), sizeof(V) * src.size(), hipMemcpyHostToDevice);
}

hipError_t launch_band_lags(BandCache** cache, int n_cu, bool helfand, const double* pm, long pitch, int T, long n_cols,
                            double factor, double* lagsum, hipStream_t st) {
    constexpr int kLabels = 8;  // one per XCD, as the hardware deals workgroups round-robin to them
    const int nwg = std::max(kLabels, n_cu / kLabels * kLabels);
    if (!*cache) *cache = new BandCache;
    BandCache& c = **cache;
    if (c.plan.T != T || c.nwg != nwg || !c.partial) {
        hipError_t e = hipStreamSynchronize(st);  // a launch that still reads the old tables
        if (e != hipSuccess) return e;
        band_release(&c);
        c.plan = band_plan(T, nwg / kLabels * 8, kLabels);
        c.nwg = nwg;
        if ((e = band_upload(&c.pieces, c.plan.pieces)) != hipSuccess) return e;
        if ((e = band_upload(&c.slot_begin, c.plan.slot_begin)) != hipSuccess) return e;
        if ((e = band_upload(&c.slot_pieces, c.plan.slot_pieces)) != hipSuccess) return e;
        if ((e = band_upload(&c.group_begin, c.plan.group_begin)) != hipSuccess) return e;
        e = hipMalloc((void**)&c.partial, sizeof(double) * (size_t)kLabels * c.plan.pieces.size() * kBandPartial);
        if (e != hipSuccess) {
            c.plan.T = 0;
            return e;
        }
    }
    const long n_pairs = (n_cols + 1) / 2;
    const int n_pieces = (int)c.plan.pieces.size();
    if (helfand)  // accumulators hold -1/2 the squared differences
        hipLaunchKernelGGL(k_band_lags<true>, dim3(nwg), dim3(512), 0, st, pm, pitch, T, n_pairs, kLabels, c.plan.n_ph, c.pieces,
                           n_pieces, c.slot_begin, c.slot_pieces, c.partial);
    else
        hipLaunchKernelGGL(k_band_lags<false>, dim3(nwg), dim3(512), 0, st, pm, pitch, T, n_pairs, kLabels, c.plan.n_ph, c.pieces,
                           n_pieces, c.slot_begin, c.slot_pieces, c.partial);
    hipLaunchKernelGGL(k_band_gather, dim3((T + 255) / 256), dim3(256), 0, st, c.partial, kLabels, n_pieces, c.plan.n_ph,
                       c.plan.per_phase, c.group_begin, c.plan.n_groups, T, helfand ? -2.0 * factor : factor, helfand ? 1 : 0,
                       lagsum);
    return hipGetLastError();
}

}  // namespace ta
