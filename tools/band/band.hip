// band.hip — launcher of the FP64 matrix-core evaluation of the lag SUMS of the O(T^2) correlators
// (band_kernels.hpp): windowed VACF (/root/reference/transport_analysis/velocityautocorr.py:217-238 summed
// over particles) and, on the product slab P = (m v) x, the Einstein-Helfand mean squared differences
// (viscosity.py:201-233 summed over particles).  The cut of the band depends on n_frames and the grid only; its device copy and
// the partial-sum buffer are cached per context.
#include "band_kernels.hpp"

#include <new>

#include "../../include/ta_hip.h"
#include "../../transport_analysis_amd/csrc/ta_internal.hpp"
#include "band_tools.hpp"

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

// the cut of the band for (T, this device) and its device copies, built on first use and kept in *cache
hipError_t band_tables(BandCache** cache, int n_cu, int T, hipStream_t st, int* nwg_out, int* n_ph, int* n_pieces, int* per_phase,
                       int* n_groups, const BandPiece** pieces, const int** slot_begin, const int** slot_pieces,
                       const int** group_begin, double** partial) {
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
        e = band_upload(&c.pieces, c.plan.pieces);
        if (e == hipSuccess) e = band_upload(&c.slot_begin, c.plan.slot_begin);
        if (e == hipSuccess) e = band_upload(&c.slot_pieces, c.plan.slot_pieces);
        if (e == hipSuccess) e = band_upload(&c.group_begin, c.plan.group_begin);
        if (e == hipSuccess)
            e = hipMalloc((void**)&c.partial, sizeof(double) * (size_t)kLabels * c.plan.pieces.size() * kBandPartial);
        if (e != hipSuccess) {  // nothing half-built stays behind
            band_release(&c);
            c.plan.T = 0;
            return e;
        }
    }
    *nwg_out = nwg, *n_ph = c.plan.n_ph, *n_pieces = (int)c.plan.pieces.size(), *per_phase = c.plan.per_phase;
    *n_groups = c.plan.n_groups, *pieces = c.pieces, *slot_begin = c.slot_begin, *slot_pieces = c.slot_pieces;
    *group_begin = c.group_begin, *partial = c.partial;
    return hipSuccess;
}

hipError_t launch_band_lags(BandCache** cache, int n_cu, bool helfand, const double* pm, long pitch, int T, long n_cols,
                            double factor, double* lagsum, hipStream_t st) {
    constexpr int kLabels = 8;
    int nwg = 0, n_ph = 0, n_pieces = 0, per_phase = 0, n_groups = 0;
    const BandPiece* pieces = nullptr;
    const int *slot_begin = nullptr, *slot_pieces = nullptr, *group_begin = nullptr;
    double* partial = nullptr;
    hipError_t e = band_tables(cache, n_cu, T, st, &nwg, &n_ph, &n_pieces, &per_phase, &n_groups, &pieces, &slot_begin, &slot_pieces,
                               &group_begin, &partial);
    if (e != hipSuccess) return e;
    const long n_pairs = (n_cols + 1) / 2;
    if (helfand)  // accumulators hold -1/2 the squared differences
        hipLaunchKernelGGL(k_band_lags<true>, dim3(nwg), dim3(512), 0, st, pm, pitch, T, n_pairs, kLabels, n_ph, pieces, n_pieces,
                           slot_begin, slot_pieces, partial);
    else
        hipLaunchKernelGGL(k_band_lags<false>, dim3(nwg), dim3(512), 0, st, pm, pitch, T, n_pairs, kLabels, n_ph, pieces, n_pieces,
                           slot_begin, slot_pieces, partial);
    hipLaunchKernelGGL(k_band_gather, dim3((T + 255) / 256), dim3(256), 0, st, partial, kLabels, n_pieces, n_ph, per_phase, group_begin,
                       n_groups, T, helfand ? -2.0 * factor : factor, helfand ? 1 : 0, lagsum);
    return hipGetLastError();
}

}  // namespace ta

// Host only: the cut of the band for n_frames on a device of n_cu compute units, checked cell by
// cell (every step of every group belongs to exactly one piece of every phase, every piece to
// exactly one wave slot of its phase).
static int band_plan_info_impl(int64_t n_frames, int n_cu, int* n_pieces, int* octets_in_flight, double* max_over_mean);
extern "C" int ta_band_plan_info(int64_t n_frames, int n_cu, int* n_pieces, int* octets_in_flight, double* max_over_mean) {
    try {  // (no exception leaves an extern "C" function)
        return band_plan_info_impl(n_frames, n_cu, n_pieces, octets_in_flight, max_over_mean);
    } catch (const std::bad_alloc&) {
        return TA_E_NOMEM;
    } catch (...) {
        return TA_E_UNSUPPORTED;
    }
}
static int band_plan_info_impl(int64_t n_frames, int n_cu, int* n_pieces, int* octets_in_flight, double* max_over_mean) {
    using namespace ta;
    if (n_frames < 1 || n_frames >= ((int64_t)1 << 24) || n_cu < 8) return TA_E_INVALID;
    constexpr int kLabels = 8;
    const int nwg = n_cu / kLabels * kLabels;
    const BandPlan p = band_plan((int)n_frames, nwg / kLabels * 8, kLabels);
    const int per = p.per_phase;
    if ((int)p.pieces.size() != per * p.n_ph || (int)p.slot_begin.size() != p.slots + 1) return TA_E_UNSUPPORTED;
    // coverage of the band by one phase's pieces: within a group they are sorted by i0 (band_plan hands the steps out
    // in order), so "every step in exactly one piece" is "the intervals tile [0, steps of the group)"
    for (int g = 0; g < p.n_groups; ++g) {
        int next = 0;
        for (int li = p.group_begin[g]; li < p.group_begin[g + 1]; ++li) {
            const BandPiece& q = p.pieces[li];
            if (q.d0 != 16 * g || q.i0 != next || q.i1 <= q.i0) return TA_E_UNSUPPORTED;
            next = q.i1;
        }
        if (next != p.nblk - 16 * g) return TA_E_UNSUPPORTED;
    }
    // every piece of every phase in exactly one slot, of its own phase
    std::vector<int> owner(p.pieces.size(), 0);
    const int wslots = p.slots / p.n_ph;
    for (int s = 0; s < p.slots; ++s)
        for (int k = p.slot_begin[s]; k < p.slot_begin[s + 1]; ++k) {
            const int idx = p.slot_pieces[k];
            if (idx < 0 || idx >= (int)p.pieces.size() || p.pieces[idx].phase != s / wslots || idx / per != s / wslots)
                return TA_E_UNSUPPORTED;
            ++owner[idx];
        }
    for (int c : owner)
        if (c != 1) return TA_E_UNSUPPORTED;
    if (n_pieces) *n_pieces = (int)p.pieces.size();
    if (octets_in_flight) *octets_in_flight = p.n_ph;
    if (max_over_mean) *max_over_mean = p.mean_cost > 0 ? p.max_cost / p.mean_cost : 1.0;
    return TA_OK;
}
