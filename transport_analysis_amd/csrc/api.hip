// api.hip — C-ABI entry points of libta_hip.so (declared in include/ta_hip.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/ta_hip.h"
#include "direct_kernels.hpp"
#include "ta_internal.hpp"

using namespace ta;

namespace {

thread_local std::string g_tls_error;

struct Tables {
    cd* tw2 = nullptr;    // W_{2M}^n = exp(-i pi n / M), n < 2M
    int* perm = nullptr;  // the plan's output position -> frequency (made on first use)
};

struct LongTables {     // fft_long.hip
    cd* twL = nullptr;  // W_{2M'}^n, n < 2M'
    int* perm = nullptr;  // plan M's output position -> frequency
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct ta_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // device->host copies of by-particle blocks (host_compute)
    std::string err;
    std::map<int, Tables> tables;
    std::map<long, LongTables> long_tables;  // keyed by M'
    std::map<int, cd*> wf_tables;            // wfft.hip tables, keyed by R0
    DevBuf partial, spec, ts_partial, out_lagsum, out_bp, masses, bounce, stage_buf, long_scratch, helf_p, helf_small;
    DevBuf pm_in[2];  // pair-major copies of frame-major *_dev inputs
    DevBuf bp_scratch;  // atom-major by-particle results before the transposition
    DevBuf bp_spec;     // per-atom power spectra of one block of atoms (two-kernel by-particle path)
    // staging: pinned host slabs keep the reference's (n_frames, n_atoms, dim) layout, the
    // device slabs are pair-major (layout.hip) with st_pitch rows per column pair
    int64_t st_T = 0, st_A = 0, st_pitch = 0;
    int st_D = 0, st_dtype = TA_F64, st_nslabs = 0;
    std::vector<void*> h_slabs;
    std::vector<double*> d_slabs;
    // timing: a ring of event quadruples, one per compute call (start, main kernel start,
    // main kernel end, end), so a caller can time K calls back to back and read all K
    // durations afterwards (ta_timing_history) instead of synchronising inside its loop
    static constexpr int kRing = 64;
    hipEvent_t ring[kRing][4] = {};
    hipEvent_t* ev = ring[0];
    hipEvent_t ev_stage = nullptr;  // orders a caller's stream behind the staging stream
    long n_calls = 0;  // compute calls completed (their events recorded)
    bool timing_valid = false;
    // options
    int64_t opt_fft_nwg = 0;
    int64_t opt_direct_nwg = 0;
    int64_t opt_fft_debug = 0;
    int64_t opt_direct_f32 = 0;
    int64_t opt_direct_groups = 0;
    int64_t opt_direct_chunk = 0;
    int64_t opt_helfand_fft = 0;
    int64_t opt_bp_block = 0;
    int64_t opt_bp_spec_atoms = 0;
    int64_t opt_bp_prefetch = 2;
};

namespace {

int fail(ta_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    g_tls_error = msg;
    return code;
}

#define TA_HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(ctx, (_e == hipErrorOutOfMemory) ? TA_E_NOMEM : TA_E_HIP,              \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                    \
    } while (0)

int ensure(ta_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return TA_OK;
    if (b.p) {
        hipFree(b.p);
        b.p = nullptr;
        b.bytes = 0;
    }
    if (bytes == 0) bytes = 16;
    TA_HIP_TRY(ctx, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return TA_OK;
}

inline int64_t pm_pitch(int64_t n_frames) { return (n_frames + 7) / 8 * 8; }
inline size_t pm_bytes(int64_t n_frames, int64_t n_cols) {
    return (size_t)((n_cols + 1) / 2) * (size_t)pm_pitch(n_frames) * 16;
}

int get_wf_table(ta_ctx* ctx, int R0, cd** out) {
    auto it = ctx->wf_tables.find(R0);
    if (it != ctx->wf_tables.end()) {
        *out = it->second;
        return TA_OK;
    }
    std::vector<cd> a(wfft_table_elems(R0));
    wfft_fill_table(R0, a.data());
    cd* d = nullptr;
    TA_HIP_TRY(ctx, hipMalloc((void**)&d, sizeof(cd) * a.size()));
    hipError_t e = hipMemcpy(d, a.data(), sizeof(cd) * a.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(d);
        return fail(ctx, TA_E_HIP, std::string("twiddle table upload: ") + hipGetErrorString(e));
    }
    ctx->wf_tables[R0] = d;
    *out = d;
    return TA_OK;
}

int get_tables(ta_ctx* ctx, int M, int R0, Tables* out) {
    auto it = ctx->tables.find(M);
    if (it != ctx->tables.end()) {
        *out = it->second;
        return TA_OK;
    }
    // [0,2M): W_2M^n.  [2M,3M) and [3M,4M): the first-stage twiddles of pass A and pass B,
    // W_2M^{u(2q+B)} stored [q][u] (u < M/R0) so that a wave's 64 consecutive butterflies read
    // 1 KB contiguous (from the main table the same values sit 2q+B elements apart: one L2
    // request per lane).  [4M,4M+4): zeros.
    std::vector<cd> a(4 * (size_t)M + 4, cd{0.0, 0.0});  // + 64 zero bytes: the gathers' padding rows
    const long double pi = 3.141592653589793238462643383279502884L;
    auto w2m = [&](long n) {
        n %= 2L * M;
        if (n == 0) return cd{1.0, 0.0};
        if (n == M) return cd{-1.0, 0.0};
        if (2 * n == M) return cd{0.0, -1.0};
        if (2 * n == 3L * M) return cd{0.0, 1.0};
        long double h = pi * (long double)n / (long double)M;
        return cd{(double)cosl(h), (double)-sinl(h)};
    };
    for (long n = 0; n < 2L * M; ++n) a[n] = w2m(n);
    const long L0 = M / R0;
    for (int B = 0; B < 2; ++B)
        for (long q = 0; q < R0; ++q)
            for (long u = 0; u < L0; ++u) a[(2 + B) * (size_t)M + q * L0 + u] = w2m(u * (2 * q + B));
    Tables t;
    TA_HIP_TRY(ctx, hipMalloc((void**)&t.tw2, sizeof(cd) * (4 * (size_t)M + 4)));
    hipError_t e = hipMemcpy(t.tw2, a.data(), sizeof(cd) * (4 * (size_t)M + 4), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(t.tw2);
        return fail(ctx, TA_E_HIP, std::string("twiddle table upload: ") + hipGetErrorString(e));
    }
    ctx->tables[M] = t;
    *out = t;
    return TA_OK;
}

int get_long_tables(ta_ctx* ctx, int M, int Rout, LongTables* out) {
    const long Mp = (long)M * Rout;
    auto it = ctx->long_tables.find(Mp);
    if (it != ctx->long_tables.end()) {
        *out = it->second;
        return TA_OK;
    }
    std::vector<cd> a(2 * (size_t)Mp);
    const long double pi = 3.141592653589793238462643383279502884L;
    for (long n = 0; n < 2 * Mp; ++n) {
        if (n == 0) a[n] = cd{1.0, 0.0};
        else if (n == Mp) a[n] = cd{-1.0, 0.0};
        else if (2 * n == Mp) a[n] = cd{0.0, -1.0};
        else if (2 * n == 3 * Mp) a[n] = cd{0.0, 1.0};
        else {
            const long double h = pi * (long double)n / (long double)Mp;
            a[n] = cd{(double)cosl(h), (double)-sinl(h)};
        }
    }
    std::vector<int> perm;
    fft_long_perm(M, perm);
    LongTables t;
    hipError_t e = hipMalloc((void**)&t.twL, sizeof(cd) * a.size());
    if (e == hipSuccess) e = hipMalloc((void**)&t.perm, sizeof(int) * perm.size());
    if (e == hipSuccess) e = hipMemcpy(t.twL, a.data(), sizeof(cd) * a.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t.perm, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (t.twL) hipFree(t.twL);
        if (t.perm) hipFree(t.perm);
        return fail(ctx, e == hipErrorOutOfMemory ? TA_E_NOMEM : TA_E_HIP,
                    std::string("long-transform tables: ") + hipGetErrorString(e));
    }
    ctx->long_tables[Mp] = t;
    *out = t;
    return TA_OK;
}

const PlanEntry* plan_of_length(int M) {
    for (const auto* tab : {&plans_pow2(), &plans_five()})
        for (const auto& p : *tab)
            if (p.M == M) return &p;
    return nullptr;
}

// FFT lag sums for n_frames beyond the largest on-chip plan (fft_long.hip); timeseries only.
int fft_long_impl(ta_ctx* ctx, const double* d_pm, int64_t pitch, int64_t T, int64_t A, int D,
                  double* d_lagsum, hipStream_t st, int M, int Rout) {
    const PlanEntry* plan = plan_of_length(M);
    if (!plan) return fail(ctx, TA_E_INVALID, "no on-chip plan for the long transform");
    int rc;
    Tables tb;
    LongTables lt;
    if ((rc = get_tables(ctx, M, plan->R_first, &tb))) return rc;
    if ((rc = get_long_tables(ctx, M, Rout, &lt))) return rc;
    const int64_t n_quads = ((A * D + 1) / 2 + 3) / 4;  // a workgroup takes four adjacent pairs
    int64_t nwg = ctx->opt_fft_nwg > 0 ? ctx->opt_fft_nwg : (int64_t)ctx->n_cu;
    nwg = std::max<int64_t>(1, std::min(nwg, n_quads));
    if (nwg >= 8) nwg -= nwg % 8;  // XCD-aware walk
    const size_t blk = fft_long_acc_block(M);
    const size_t acc_bytes = sizeof(double) * (size_t)nwg * 2 * Rout * blk;
    if ((rc = ensure(ctx, ctx->partial, acc_bytes))) return rc;
    if ((rc = ensure(ctx, ctx->spec, sizeof(double) * 2 * (size_t)Rout * M))) return rc;
    if ((rc = ensure(ctx, ctx->long_scratch, sizeof(cd) * (size_t)nwg * 4 * 2 * Rout * M))) return rc;
    TA_HIP_TRY(ctx, hipMemsetAsync(ctx->partial.p, 0, acc_bytes, st));
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
    // pair-major slab: rows 2 elements apart, pairs 2*pitch apart, every pair complete (an odd
    // last column is stored next to zeros)
    TA_HIP_TRY(ctx, launch_fft_long_accum(M, (int)nwg, st, d_pm, 2, 2 * pitch, (int)T, 2 * ((A * D + 1) / 2),
                                          Rout, tb.tw2, lt.twL, (double*)ctx->partial.p,
                                          (cd*)ctx->long_scratch.p));
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
    TA_HIP_TRY(ctx, launch_fft_long_finish(M, Rout, (const double*)ctx->partial.p, (int)nwg, lt.perm,
                                           lt.twL, (int)T, (double*)ctx->spec.p, d_lagsum, st));
    return TA_OK;
}

int check_shape(ta_ctx* ctx, int64_t T, int64_t A, int D, int64_t ld_row) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (T < 1 || A < 1 || D < 1 || D > 3)
        return fail(ctx, TA_E_INVALID, "need n_frames >= 1, n_atoms >= 1, 1 <= dim <= 3");
    if (ld_row < A * D) return fail(ctx, TA_E_INVALID, "ld_row smaller than n_atoms*dim");
    if (T > (int64_t)1 << 30) return fail(ctx, TA_E_INVALID, "n_frames too large");
    return TA_OK;
}

int direct_impl(ta_ctx* ctx, int mode, const double* d_vel, const double* d_pos,
                const double* d_masses, int64_t T, int64_t A, int D, int64_t pitch, double scale,
                double* d_lagsum, double* d_bp, int64_t ld_bp, hipStream_t st) {
    const bool f32 = ctx->opt_direct_f32 != 0;
    // Shape of the launch.  A thread owns one chunk pair (2L lags); a column group = W waves;
    // a workgroup = G groups working on G atoms at once, so that ONE workgroup fills a CU's
    // 16 wave slots (G*W <= 16) and its waves are dealt evenly to the 4 SIMDs.  The column
    // must be resident next to the compute units: in LDS when it fits (float64: <= 16376
    // frames, float32: <= 27296), otherwise in an L2-resident per-group staging buffer
    // (slower, any length).  L (8 or 10 lags per chunk) is the one that wastes fewer lanes
    // and SIMD slots for this n_frames.
    const size_t lds_cap = 160 * 1024;
    struct Shape { int L, W, G; size_t col; bool gs; double eff; } best{0, 0, 0, 0, false, -1.0};
    for (int L : {8, 10}) {
        if (ctx->opt_direct_chunk > 0 && L != ctx->opt_direct_chunk) continue;
        if (!direct_chunk_supported(L)) continue;
        Shape c;
        c.L = L;
        c.col = direct_lds_bytes((int)T, f32, L);
        c.gs = c.col > lds_cap;
        const int npairs = ((int)((T + L - 1) / L) + 1) / 2;
        c.W = std::min(16, (npairs + 63) / 64);
        c.G = 16 / c.W;
        if (!c.gs) c.G = (int)std::min<size_t>(c.G, lds_cap / c.col);
        if (ctx->opt_direct_groups > 0) c.G = (int)std::min<int64_t>(c.G, ctx->opt_direct_groups);
        c.G = (int)std::max<int64_t>(1, std::min<int64_t>(c.G, A));
        const int rounds = (npairs + c.W * 64 - 1) / (c.W * 64);
        const int waves = c.G * c.W;
        c.eff = (double)npairs / ((double)rounds * c.W * 64) *   // active lanes
                (double)waves / (4.0 * ((waves + 3) / 4)) *      // SIMD balance
                (1.0 - 0.6 / L);                                 // per-tile overhead
        if (c.eff > best.eff) best = c;
    }
    if (best.eff < 0) return fail(ctx, TA_E_INVALID, "direct_chunk option: unsupported chunk size");
    const int L = best.L, W = best.W, G = best.G;
    const size_t col = best.col;
    const bool global_stage = best.gs;
    const int gnt = W * 64, nt = G * gnt;
    const size_t lds = global_stage ? 0 : col * (size_t)G;
    const int per_cu = direct_max_wg_per_cu(mode, f32, L, nt, lds, global_stage);
    int64_t nwg = ctx->opt_direct_nwg > 0 ? ctx->opt_direct_nwg : (int64_t)ctx->n_cu * per_cu;
    nwg = std::max<int64_t>(1, std::min<int64_t>(nwg, (A + G - 1) / G));
    const size_t rows = (size_t)nwg * G;
    int rc = ensure(ctx, ctx->ts_partial, sizeof(double) * rows * T);
    if (rc) return rc;
    void* stage_buf = nullptr;
    if (global_stage) {
        if ((rc = ensure(ctx, ctx->stage_buf, col * rows))) return rc;
        stage_buf = ctx->stage_buf.p;
    }
    // by-particle values leave the kernel atom-major (contiguous stores) and are transposed
    // into the caller's (n_frames, ld_bp) array afterwards
    double* bp_am = nullptr;
    const int64_t Tp = pm_pitch(T);
    if (d_bp) {
        if ((rc = ensure(ctx, ctx->bp_scratch, sizeof(double) * (size_t)A * Tp))) return rc;
        bp_am = (double*)ctx->bp_scratch.p;
    }
    TA_HIP_TRY(ctx, hipMemsetAsync(ctx->ts_partial.p, 0, sizeof(double) * rows * T, st));
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
    TA_HIP_TRY(ctx, launch_direct(mode, f32, L, d_vel, d_pos, d_masses, pitch, (int)T, A, D, scale, bp_am,
                                  Tp, (double*)ctx->ts_partial.p, (int)nwg, nt, lds, stage_buf,
                                  gnt, st));
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
    TA_HIP_TRY(ctx, launch_sum_partials((const double*)ctx->ts_partial.p, (int)rows, T, d_lagsum, st));
    if (d_bp) TA_HIP_TRY(ctx, launch_bp_transpose(bp_am, Tp, A, T, d_bp, ld_bp, nullptr, st));
    return TA_OK;
}


// ---- compute on pair-major slabs (every entry point ends up here) --------------------------
// FFT VACF.  Lag sums only (d_bp == NULL): T <= 512 the small on-chip plans, T <= 10240 the
// wave-local kernels of wfft.hpp, T <= 163840 the outer-radix path; with a by-particle array
// the on-chip plans up to 10240 frames; everything else the direct correlator (same quantity:
// velocityautocorr.py:217-238 == :208-215 mathematically).
int fft_impl(ta_ctx* ctx, const double* pm, int64_t pitch, int64_t T, int64_t A, int D,
             double* d_lagsum, double* d_bp, int64_t ld_bp, hipStream_t st) {
    int rc;
    const int64_t n_cols = A * D, n_pairs = (n_cols + 1) / 2;
    int R0 = 0;
    if (!d_bp && wfft_choose((long)T, &R0)) {
        cd* tw = nullptr;
        if ((rc = get_wf_table(ctx, R0, &tw))) return rc;
        const int L2 = 2 * R0 * 512;
        int64_t nwg = ctx->opt_fft_nwg > 0 ? ctx->opt_fft_nwg : (int64_t)ctx->n_cu * wfft_max_wg_per_cu(R0);
        // pass-split form (one pass per workgroup, the two workgroups of a couple share an
        // XCD's L2: every input byte leaves HBM once) whenever 16 workgroups have work
        const bool split = R0 > 1 && std::min<int64_t>(nwg, 2 * n_pairs) >= 16 && ctx->opt_fft_debug != 2;
        if (split) nwg = std::min<int64_t>(nwg, 2 * n_pairs) / 16 * 16;
        else if (R0 == 1) nwg = std::max<int64_t>(1, std::min(nwg, (n_pairs + 3) / 4));  // a wave per pair
        else nwg = std::max<int64_t>(1, std::min(nwg, n_pairs));
        int n_parts = (int)(split ? nwg / 2 : R0 == 1 ? 4 * nwg : nwg);
        if ((rc = ensure(ctx, ctx->partial, sizeof(double) * (size_t)n_parts * L2))) return rc;
        if ((rc = ensure(ctx, ctx->spec, sizeof(double) * (size_t)L2))) return rc;
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
        if (split)
            TA_HIP_TRY(ctx, launch_wfft_split(R0, (int)nwg, st, pm, pitch, (int)T, n_pairs, tw,
                                              (double*)ctx->partial.p));
        else
            TA_HIP_TRY(ctx, launch_wfft_accum(R0, (int)nwg, st, pm, pitch, (int)T, n_pairs, tw,
                                              (double*)ctx->partial.p));
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
        nwg = n_parts;
        // the summed spectrum -> lag sums: ONE inverse transform per launch, run by the on-chip
        // plan of the same length (its finalize kernel consumes the digit-reversed order, which
        // the sum over workgroups produces on the way)
        const PlanEntry* fin = plan_of_length(R0 * 512);
        if (!fin) {
            TA_HIP_TRY(ctx, launch_wfft_finish(R0, (const double*)ctx->partial.p, (int)nwg, tw, (int)T,
                                               (double*)ctx->spec.p, d_lagsum, st));
            return TA_OK;
        }
        Tables tb;
        if ((rc = get_tables(ctx, fin->M, fin->R_first, &tb))) return rc;
        if (!tb.perm) {
            std::vector<int> perm;
            fin->perm(perm);
            int* d = nullptr;
            TA_HIP_TRY(ctx, hipMalloc((void**)&d, sizeof(int) * perm.size()));
            hipError_t e = hipMemcpy(d, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                hipFree(d);
                return fail(ctx, TA_E_HIP, std::string("permutation upload: ") + hipGetErrorString(e));
            }
            ctx->tables[fin->M].perm = tb.perm = d;
        }
        TA_HIP_TRY(ctx, launch_wfft_sum_perm((const double*)ctx->partial.p, (int)nwg, fin->M, tb.perm,
                                             (double*)ctx->spec.p, st));
        FftArgs fa{(int)T, tb.tw2, (const double*)ctx->spec.p, 1, d_lagsum};
        TA_HIP_TRY(ctx, fin->finalize(st, fa));
        return TA_OK;
    }
    if (d_bp && wfft_choose((long)T, &R0)) {
        // by-particle mode on the wave-local machinery (k_wbp): per-atom lag values to an
        // atom-major scratch (512-byte stores), then the transposition into the caller's
        // (n_frames, ld_bp) array, which also adds up its 64 atoms per lag
        cd* tw = nullptr;
        if ((rc = get_wf_table(ctx, R0, &tw))) return rc;
        const int64_t Tp = pm_pitch(T), n_tiles = (A + 63) / 64;
        int64_t nwg = ctx->opt_fft_nwg > 0 ? ctx->opt_fft_nwg : (int64_t)ctx->n_cu * wfft_max_wg_per_cu(R0);
        nwg = std::max<int64_t>(1, std::min(nwg, R0 == 1 ? (A + 3) / 4 : A));
        if ((rc = ensure(ctx, ctx->bp_scratch, sizeof(double) * (size_t)A * Tp))) return rc;
        if ((rc = ensure(ctx, ctx->ts_partial, sizeof(double) * (size_t)n_tiles * T))) return rc;
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
        // two kernels per block of atoms where a block fills the chip: the pass-split forward
        // kernel (couples of workgroups share their rows through the XCD's L2; half the
        // accumulators per thread, so rows are prefetched and sub-series interleaved as on the
        // lag-sum path) leaves each atom's power spectrum in scratch, the inverse kernel turns
        // it into the atom's lags.  Blocks start on an even atom, so on a pair boundary.
        const int64_t fwd_wg = std::min<int64_t>((int64_t)ctx->n_cu * wfft_max_wg_per_cu(R0), 2 * A) / 16 * 16;
        if (R0 > 1 && fwd_wg >= 16 && ctx->opt_fft_debug != 3) {
            const size_t spec_per_atom = sizeof(double) * 2 * (size_t)R0 * 512;
            const int64_t CA = std::min<int64_t>(A, ctx->opt_bp_spec_atoms > 0 ? (ctx->opt_bp_spec_atoms + 1) / 2 * 2 : 16384);
            if ((rc = ensure(ctx, ctx->bp_spec, spec_per_atom * (size_t)CA))) return rc;
            for (int64_t a0 = 0; a0 < A; a0 += CA) {
                const int64_t ca = std::min(CA, A - a0);
                const int64_t groups = D & 1 ? (ca + 1) / 2 : ca;  // a couple of workgroups per group of atoms
                const int64_t fw = std::max<int64_t>(16, std::min<int64_t>(fwd_wg, 2 * groups) / 16 * 16);
                TA_HIP_TRY(ctx, launch_wfft_by_particle2(R0, (int)fw, (int)std::min<int64_t>(nwg, ca), st,
                                                         pm + (a0 * D / 2) * pitch * 2, pitch, (int)T, ca, D, tw,
                                                         (double*)ctx->bp_spec.p,
                                                         (double*)ctx->bp_scratch.p + a0 * Tp, Tp,
                                                         (int)ctx->opt_bp_prefetch));
            }
        } else {
            TA_HIP_TRY(ctx, launch_wfft_by_particle(R0, (int)nwg, st, pm, pitch, (int)T, A, D, tw,
                                                    (double*)ctx->bp_scratch.p, Tp));
        }
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
        TA_HIP_TRY(ctx, launch_bp_transpose((const double*)ctx->bp_scratch.p, Tp, A, T, d_bp, ld_bp,
                                            (double*)ctx->ts_partial.p, st));
        TA_HIP_TRY(ctx, launch_sum_partials((const double*)ctx->ts_partial.p, (int)n_tiles, T, d_lagsum, st));
        return TA_OK;
    }
    // beyond the on-chip lengths: lag sums by the outer-radix path, anything else by the direct
    // correlator (same quantity: velocityautocorr.py:217-238 == :208-215 mathematically)
    int long_M = 0, long_R = 0;
    if (!d_bp && fft_long_choose((long)T, &long_M, &long_R))
        return fft_long_impl(ctx, pm, pitch, T, A, D, d_lagsum, st, long_M, long_R);
    return direct_impl(ctx, MODE_VACF, pm, nullptr, nullptr, T, A, D, pitch, 1.0, d_lagsum, d_bp, ld_bp, st);
}

// Helfand mean squared differences (viscosity.py:201-233); the "helfand_fft" option evaluates
// them as S1 - 2 S2 (helfand_fft.hip) where an FFT path exists for the request.
int helfand_impl(ta_ctx* ctx, const double* pm_vel, const double* pm_pos, const double* d_masses,
                 int64_t pitch, int64_t T, int64_t A, int D, double scale, double* d_lagsum,
                 double* d_bp, int64_t ld_bp, hipStream_t st) {
    int rc;
    const int64_t n_cols = A * D, n_pairs = (n_cols + 1) / 2;
    const bool fft_ok = ctx->opt_helfand_fft && T >= 2;
    int r0 = 0, lm = 0, lr = 0;
    const bool has_fft_ts = wfft_choose((long)T, &r0) || fft_long_choose((long)T, &lm, &lr);
    if (fft_ok && !d_bp && has_fft_ts) {
        const int n_parts = (int)std::min<int64_t>(1024, n_pairs);
        if ((rc = ensure(ctx, ctx->helf_p, pm_bytes(T, n_cols)))) return rc;
        if ((rc = ensure(ctx, ctx->helf_small, sizeof(double) * ((size_t)n_parts * T + 3 * (size_t)T + 1)))) return rc;
        double* P = (double*)ctx->helf_p.p;
        double* Qpart = (double*)ctx->helf_small.p;
        double* Q = Qpart + (size_t)n_parts * T;
        double* S2 = Q + T;
        double* C = S2 + T;
        TA_HIP_TRY(ctx, hipMemsetAsync(Qpart, 0, sizeof(double) * (size_t)n_parts * T, st));
        TA_HIP_TRY(ctx, launch_helfand_product(pm_vel, pm_pos, d_masses, pitch, T, n_cols, D, P, Qpart, n_parts, st));
        TA_HIP_TRY(ctx, launch_sum_partials(Qpart, n_parts, T, Q, st));
        if ((rc = fft_impl(ctx, P, pitch, T, A, D, S2, nullptr, 0, st))) return rc;
        TA_HIP_TRY(ctx, launch_helfand_combine(Q, S2, C, (int)T, scale / (double)D, d_lagsum, st));
        return TA_OK;
    }
    if (fft_ok && d_bp && wfft_choose((long)T, &r0)) {
        if ((rc = ensure(ctx, ctx->helf_p, pm_bytes(T, n_cols)))) return rc;
        if ((rc = ensure(ctx, ctx->helf_small, sizeof(double) * ((size_t)T + 1) * A))) return rc;
        double* P = (double*)ctx->helf_p.p;
        double* Ca = (double*)ctx->helf_small.p;
        if (n_cols & 1)  // the unpaired last column's partner is never written by the product kernel
            TA_HIP_TRY(ctx, hipMemsetAsync(P + (size_t)(n_pairs - 1) * pitch * 2, 0, (size_t)pitch * 16, st));
        TA_HIP_TRY(ctx, launch_helfand_product_bp(pm_vel, pm_pos, d_masses, pitch, T, A, D, P, Ca, st));
        if ((rc = fft_impl(ctx, P, pitch, T, A, D, d_lagsum, d_bp, ld_bp, st))) return rc;
        TA_HIP_TRY(ctx, launch_helfand_combine_bp(Ca, A, (int)T, scale / (double)D, d_bp, ld_bp, st));
        TA_HIP_TRY(ctx, launch_row_sums(d_bp, T, A, ld_bp, d_lagsum, st));
        return TA_OK;
    }
    return direct_impl(ctx, MODE_HELFAND, pm_vel, pm_pos, d_masses, T, A, D, pitch, scale, d_lagsum, d_bp,
                       ld_bp, st);
}

enum { W_FFT = 0, W_DIRECT = 1, W_HELFAND = 2 };

// one compute call on pair-major slabs, bracketed by the timing events
int compute_pm(ta_ctx* ctx, int which, const double* pm_vel, const double* pm_pos, const double* d_masses,
               int64_t pitch, int64_t T, int64_t A, int D, double scale, double* d_lagsum, double* d_bp,
               int64_t ld_bp, hipStream_t st, bool record_start) {
    int rc;
    ctx->timing_valid = false;
    if (record_start) {
        ctx->ev = ctx->ring[ctx->n_calls % ta_ctx::kRing];
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[0], st));
    }
    // paths without a dominant kernel of their own re-record ev[1]/ev[2] inside
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
    if (which == W_FFT) rc = fft_impl(ctx, pm_vel, pitch, T, A, D, d_lagsum, d_bp, ld_bp, st);
    else if (which == W_DIRECT)
        rc = direct_impl(ctx, MODE_VACF, pm_vel, nullptr, nullptr, T, A, D, pitch, 1.0, d_lagsum, d_bp, ld_bp, st);
    else rc = helfand_impl(ctx, pm_vel, pm_pos, d_masses, pitch, T, A, D, scale, d_lagsum, d_bp, ld_bp, st);
    if (rc) return rc;
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[3], st));
    ctx->timing_valid = true;
    ++ctx->n_calls;
    return TA_OK;
}

// frame-major device input of a *_dev entry point -> the context's pair-major scratch slab
int relayout_input(ta_ctx* ctx, int k, const double* d_src, int64_t T, int64_t n_cols, int64_t ld_row,
                   hipStream_t st, const double** out) {
    int rc = ensure(ctx, ctx->pm_in[k], pm_bytes(T, n_cols));
    if (rc) return rc;
    TA_HIP_TRY(ctx, launch_relayout(d_src, false, ld_row, n_cols, T, (double*)ctx->pm_in[k].p, pm_pitch(T), 0, st));
    *out = (const double*)ctx->pm_in[k].p;
    return TA_OK;
}

int dev_entry(ta_ctx* ctx, int which, const double* d_vel, const double* d_pos, const double* d_masses,
              int64_t T, int64_t A, int D, int64_t ld_row, double scale, double* d_lagsum, double* d_bp,
              int64_t ld_bp, void* stream) {
    int rc = check_shape(ctx, T, A, D, ld_row);
    if (rc) return rc;
    if (!d_vel || !d_lagsum || (which == W_HELFAND && (!d_pos || !d_masses)))
        return fail(ctx, TA_E_INVALID, "null device pointer");
    if (d_bp && ld_bp < A) return fail(ctx, TA_E_INVALID, "ld_bp smaller than n_atoms");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the legacy default stream, as for any HIP call
    ctx->timing_valid = false;
    ctx->ev = ctx->ring[ctx->n_calls % ta_ctx::kRing];
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[0], st));
    const double *pv = nullptr, *px = nullptr;
    if ((rc = relayout_input(ctx, 0, d_vel, T, A * D, ld_row, st, &pv))) return rc;
    if (which == W_HELFAND && (rc = relayout_input(ctx, 1, d_pos, T, A * D, ld_row, st, &px))) return rc;
    return compute_pm(ctx, which, pv, px, d_masses, pm_pitch(T), T, A, D, scale, d_lagsum, d_bp, ld_bp, st, false);
}

// frames committed by ta_stage_commit travel on the context's own stream: a caller's stream that
// is about to touch the slabs waits for them (a no-op when nothing is pending)
int order_after_staging(ta_ctx* ctx, hipStream_t st) {
    if (st == ctx->stream) return TA_OK;
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
    TA_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_stage, 0));
    return TA_OK;
}

int staged_entry(ta_ctx* ctx, int which, const double* d_masses, double scale, double* d_lagsum,
                 double* d_bp, int64_t ld_bp, void* stream) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    const int need = which == W_HELFAND ? 2 : 1;
    if (ctx->st_nslabs < need) return fail(ctx, TA_E_STATE, "slabs have not been staged");
    if (!d_lagsum || (which == W_HELFAND && !d_masses)) return fail(ctx, TA_E_INVALID, "null device pointer");
    if (d_bp && ld_bp < ctx->st_A) return fail(ctx, TA_E_INVALID, "ld_bp smaller than n_atoms");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = order_after_staging(ctx, (hipStream_t)stream);
    if (rc) return rc;
    return compute_pm(ctx, which, ctx->d_slabs[0], need == 2 ? ctx->d_slabs[1] : nullptr, d_masses,
                      ctx->st_pitch, ctx->st_T, ctx->st_A, ctx->st_D, scale, d_lagsum, d_bp, ld_bp,
                      (hipStream_t)stream, true);
}

}  // namespace

extern "C" {

int ta_abi_version(void) { return 2; }

int ta_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ta_last_error(const ta_ctx* ctx) { return ctx ? ctx->err.c_str() : g_tls_error.c_str(); }

int ta_ctx_create(int device, ta_ctx** out) {
    if (!out) return fail(nullptr, TA_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1)
        return fail(nullptr, TA_E_HIP,
                    "no usable HIP device (this library has no CPU fallback): " +
                        std::string(e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
    if (device < 0 || device >= n) return fail(nullptr, TA_E_INVALID, "device index out of range");
    ta_ctx* ctx = new (std::nothrow) ta_ctx();
    if (!ctx) return fail(nullptr, TA_E_NOMEM, "out of host memory");
    ctx->device = device;
    hipDeviceProp_t prop;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    for (auto& q : ctx->ring)
        for (auto& ev : q)
            if (e == hipSuccess) e = hipEventCreate(&ev);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_stage, hipEventDisableTiming);
    if (e != hipSuccess) {
        const std::string msg = std::string("context setup: ") + hipGetErrorString(e);
        ta_ctx_destroy(ctx);  // frees whatever was created
        return fail(nullptr, TA_E_HIP, msg);
    }
    ctx->n_cu = prop.multiProcessorCount;
    *out = ctx;
    return TA_OK;
}

int ta_stage_free(ta_ctx* ctx) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (void* h : ctx->h_slabs)
        if (h) hipHostFree(h);
    for (double* d : ctx->d_slabs)
        if (d) hipFree(d);
    ctx->h_slabs.clear();
    ctx->d_slabs.clear();
    ctx->st_nslabs = 0;
    ctx->st_T = ctx->st_A = ctx->st_pitch = 0;
    return TA_OK;
}

int ta_ctx_destroy(ta_ctx* ctx) {
    if (!ctx) return TA_OK;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    ta_stage_free(ctx);
    for (auto& kv : ctx->tables) {
        hipFree(kv.second.tw2);
        if (kv.second.perm) hipFree(kv.second.perm);
    }
    for (auto& kv : ctx->wf_tables) hipFree(kv.second);
    for (auto& kv : ctx->long_tables) {
        hipFree(kv.second.twL);
        hipFree(kv.second.perm);
    }
    for (DevBuf* b : {&ctx->partial, &ctx->spec, &ctx->ts_partial, &ctx->out_lagsum, &ctx->out_bp,
                      &ctx->masses, &ctx->bounce, &ctx->stage_buf, &ctx->long_scratch, &ctx->helf_p,
                      &ctx->helf_small, &ctx->pm_in[0], &ctx->pm_in[1], &ctx->bp_scratch, &ctx->bp_spec})
        if (b->p) hipFree(b->p);
    for (auto& q : ctx->ring)
        for (auto& ev : q)
            if (ev) hipEventDestroy(ev);
    if (ctx->ev_stage) hipEventDestroy(ctx->ev_stage);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return TA_OK;
}

int ta_trim(ta_ctx* ctx) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    for (DevBuf* b : {&ctx->partial, &ctx->spec, &ctx->ts_partial, &ctx->out_bp, &ctx->bounce, &ctx->stage_buf,
                      &ctx->long_scratch, &ctx->helf_p, &ctx->helf_small, &ctx->pm_in[0], &ctx->pm_in[1],
                      &ctx->bp_scratch, &ctx->bp_spec})
        if (b->p) {
            hipFree(b->p);
            b->p = nullptr;
            b->bytes = 0;
        }
    return TA_OK;
}

int ta_set_option(ta_ctx* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return fail(ctx, TA_E_INVALID, "null argument");
    if (!strcmp(key, "fft_nwg")) ctx->opt_fft_nwg = value;
    else if (!strcmp(key, "direct_nwg")) ctx->opt_direct_nwg = value;
    else if (!strcmp(key, "fft_debug")) ctx->opt_fft_debug = value;
    else if (!strcmp(key, "direct_f32")) ctx->opt_direct_f32 = value;
    else if (!strcmp(key, "direct_groups")) ctx->opt_direct_groups = value;
    else if (!strcmp(key, "direct_chunk")) ctx->opt_direct_chunk = value;
    else if (!strcmp(key, "helfand_fft")) ctx->opt_helfand_fft = value;
    else if (!strcmp(key, "bp_block")) ctx->opt_bp_block = value;
    else if (!strcmp(key, "bp_spec_atoms")) ctx->opt_bp_spec_atoms = value;
    else if (!strcmp(key, "bp_prefetch")) ctx->opt_bp_prefetch = value;
    else return fail(ctx, TA_E_INVALID, std::string("unknown option ") + key);
    return TA_OK;
}

int ta_fft_plan_info(int64_t n_frames, int64_t* m_out, int* n_threads, int* n_stages) {
    int R0 = 0;
    if (wfft_choose((long)n_frames, &R0)) {
        // first-stage radix R0 (none up to 512 frames), then one wave per 512-point sub-series (8 x 8 x 8)
        if (m_out) *m_out = (int64_t)R0 * 512;
        if (n_threads) *n_threads = R0 == 1 ? 64 : 512;
        if (n_stages) *n_stages = R0 == 1 ? 3 : 4;
        return TA_OK;
    }
    int long_M = 0, long_R = 0;
    if (fft_long_choose((long)n_frames, &long_M, &long_R)) {
        // outer radix step + on-chip transform (lag sums only; fft_long.hip)
        const PlanEntry* q = plan_of_length(long_M);
        if (m_out) *m_out = (int64_t)long_M * long_R;
        if (n_threads) *n_threads = q ? q->NT : 0;
        if (n_stages) *n_stages = q ? q->S + 1 : 0;
        return TA_OK;
    }
    return fail(nullptr, TA_E_UNSUPPORTED, "n_frames exceeds the largest FFT plan");
}

/* ------------------------------------------------------------------ staging */
static int stage_alloc_common(ta_ctx* ctx, int64_t n_frames, int64_t n_atoms, int dim, int dtype,
                              int n_slabs, void** h_slabs) {
    int rc = check_shape(ctx, n_frames, n_atoms, dim, n_atoms * dim);
    if (rc) return rc;
    if (n_slabs < 1 || n_slabs > 4) return fail(ctx, TA_E_INVALID, "bad slab count");
    if (dtype != TA_F32 && dtype != TA_F64) return fail(ctx, TA_E_INVALID, "bad dtype");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ta_stage_free(ctx);
    const size_t n = (size_t)n_frames * n_atoms * dim;
    const size_t esz = dtype == TA_F32 ? 4 : 8;
    const size_t dbytes = pm_bytes(n_frames, n_atoms * dim);
    for (int i = 0; i < n_slabs; ++i) {
        void* h = nullptr;
        double* d = nullptr;
        hipError_t e = hipSuccess;
        if (h_slabs) e = hipHostMalloc(&h, n * esz, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void**)&d, dbytes);
        if (e != hipSuccess) {
            if (h) hipHostFree(h);
            ta_stage_free(ctx);
            return fail(ctx, TA_E_NOMEM, std::string("staging allocation failed: ") + hipGetErrorString(e));
        }
        if (h) memset(h, 0, n * esz);  // the reference starts from np.zeros (velocityautocorr.py:150)
        ctx->h_slabs.push_back(h);
        ctx->d_slabs.push_back(d);
        // frames never committed read as zeros, like the reference's np.zeros slab
        TA_HIP_TRY(ctx, hipMemsetAsync(d, 0, dbytes, ctx->stream));
        if (h_slabs) h_slabs[i] = h;
    }
    ctx->st_T = n_frames;
    ctx->st_A = n_atoms;
    ctx->st_D = dim;
    ctx->st_dtype = dtype;
    ctx->st_nslabs = n_slabs;
    ctx->st_pitch = pm_pitch(n_frames);
    // the zero fill ran on the context's stream; later fills may come on any stream
    TA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TA_OK;
}

int ta_stage_alloc(ta_ctx* ctx, int64_t n_frames, int64_t n_atoms, int dim, int dtype, int n_slabs,
                   void** h_slabs) {
    if (!h_slabs) return fail(ctx, TA_E_INVALID, "h_slabs is NULL");
    return stage_alloc_common(ctx, n_frames, n_atoms, dim, dtype, n_slabs, h_slabs);
}

int ta_stage_alloc_device(ta_ctx* ctx, int64_t n_frames, int64_t n_atoms, int dim, int n_slabs) {
    return stage_alloc_common(ctx, n_frames, n_atoms, dim, TA_F64, n_slabs, nullptr);
}

int ta_stage_commit(ta_ctx* ctx, int64_t frame_lo, int64_t frame_hi) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (ctx->st_nslabs == 0) return fail(ctx, TA_E_STATE, "ta_stage_alloc has not been called");
    if (frame_lo < 0 || frame_hi > ctx->st_T || frame_lo > frame_hi)
        return fail(ctx, TA_E_INVALID, "frame range out of bounds");
    if (!ctx->h_slabs[0]) return fail(ctx, TA_E_STATE, "device-only slabs: use ta_stage_commit_dev");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t row = (size_t)ctx->st_A * ctx->st_D;
    const size_t esz = ctx->st_dtype == TA_F32 ? 4 : 8;
    if (frame_hi == frame_lo) return TA_OK;
    // frames cross PCIe in their native width into a landing buffer (<= 64 MiB) and are
    // transposed into the pair-major slab on the device (float32 widened on the way)
    const int64_t per = std::max<int64_t>(1, (int64_t)(((size_t)64 << 20) / (row * esz)));
    const int64_t chunk = std::min<int64_t>(per, frame_hi - frame_lo);
    int rc = ensure(ctx, ctx->bounce, (size_t)chunk * row * esz);
    if (rc) return rc;
    for (int i = 0; i < ctx->st_nslabs; ++i) {
        for (int64_t f = frame_lo; f < frame_hi; f += chunk) {
            const int64_t m = std::min(chunk, frame_hi - f);
            const char* src = (const char*)ctx->h_slabs[i] + (size_t)f * row * esz;
            TA_HIP_TRY(ctx, hipMemcpyAsync(ctx->bounce.p, src, (size_t)m * row * esz, hipMemcpyHostToDevice,
                                           ctx->stream));
            TA_HIP_TRY(ctx, launch_relayout(ctx->bounce.p, ctx->st_dtype == TA_F32, (long)row, (long)row, m,
                                            ctx->d_slabs[i], ctx->st_pitch, f, ctx->stream));
        }
    }
    return TA_OK;
}

int ta_stage_commit_dev(ta_ctx* ctx, int slab, const void* d_src, int dtype, int64_t ld_row,
                        int64_t frame_lo, int64_t frame_hi, void* stream) {
    if (!ctx || !d_src) return fail(ctx, TA_E_INVALID, "null argument");
    if (slab < 0 || slab >= ctx->st_nslabs) return fail(ctx, TA_E_INVALID, "no such slab");
    if (dtype != TA_F32 && dtype != TA_F64) return fail(ctx, TA_E_INVALID, "bad dtype");
    if (frame_lo < 0 || frame_hi > ctx->st_T || frame_lo > frame_hi)
        return fail(ctx, TA_E_INVALID, "frame range out of bounds");
    if (ld_row < ctx->st_A * ctx->st_D) return fail(ctx, TA_E_INVALID, "ld_row smaller than n_atoms*dim");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc_ = order_after_staging(ctx, (hipStream_t)stream)) return rc_;
    TA_HIP_TRY(ctx, launch_relayout(d_src, dtype == TA_F32, ld_row, ctx->st_A * ctx->st_D, frame_hi - frame_lo,
                                    ctx->d_slabs[slab], ctx->st_pitch, frame_lo, (hipStream_t)stream));
    return TA_OK;
}

int ta_stage_synth(ta_ctx* ctx, int slab, uint64_t seed, int64_t col_offset, int64_t n_cols_total,
                   void* stream) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (slab < 0 || slab >= ctx->st_nslabs) return fail(ctx, TA_E_INVALID, "no such slab");
    if (col_offset < 0 || col_offset + ctx->st_A * ctx->st_D > n_cols_total)
        return fail(ctx, TA_E_INVALID, "column block outside the synthetic tensor");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc_ = order_after_staging(ctx, (hipStream_t)stream)) return rc_;
    TA_HIP_TRY(ctx, launch_synth(ctx->d_slabs[slab], ctx->st_pitch, ctx->st_A * ctx->st_D, ctx->st_T, seed,
                                 col_offset, n_cols_total, (hipStream_t)stream));
    return TA_OK;
}

int ta_stage_read_dev(ta_ctx* ctx, int slab, double* d_dst, int64_t ld_row, void* stream) {
    if (!ctx || !d_dst) return fail(ctx, TA_E_INVALID, "null argument");
    if (slab < 0 || slab >= ctx->st_nslabs) return fail(ctx, TA_E_INVALID, "no such slab");
    if (ld_row < ctx->st_A * ctx->st_D) return fail(ctx, TA_E_INVALID, "ld_row smaller than n_atoms*dim");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc_ = order_after_staging(ctx, (hipStream_t)stream)) return rc_;
    TA_HIP_TRY(ctx, launch_unlayout(ctx->d_slabs[slab], ctx->st_pitch, ctx->st_A * ctx->st_D, ctx->st_T, d_dst,
                                    ld_row, (hipStream_t)stream));
    return TA_OK;
}

int ta_stage_device(ta_ctx* ctx, int slab, double** d_slab, int64_t* pitch_rows, int64_t* n_pairs) {
    if (!ctx || !d_slab) return fail(ctx, TA_E_INVALID, "null argument");
    if (slab < 0 || slab >= ctx->st_nslabs) return fail(ctx, TA_E_INVALID, "no such slab");
    *d_slab = ctx->d_slabs[slab];
    if (pitch_rows) *pitch_rows = ctx->st_pitch;
    if (n_pairs) *n_pairs = (ctx->st_A * ctx->st_D + 1) / 2;
    return TA_OK;
}

/* --------------------------------------------------------- device compute */
int ta_vacf_fft_dev(ta_ctx* ctx, const double* d_vel, int64_t T, int64_t A, int D, int64_t ld_row,
                    double* d_lagsum, double* d_bp, int64_t ld_bp, void* stream) {
    return dev_entry(ctx, W_FFT, d_vel, nullptr, nullptr, T, A, D, ld_row, 1.0, d_lagsum, d_bp, ld_bp, stream);
}

int ta_vacf_direct_dev(ta_ctx* ctx, const double* d_vel, int64_t T, int64_t A, int D, int64_t ld_row,
                       double* d_lagsum, double* d_bp, int64_t ld_bp, void* stream) {
    return dev_entry(ctx, W_DIRECT, d_vel, nullptr, nullptr, T, A, D, ld_row, 1.0, d_lagsum, d_bp, ld_bp, stream);
}

int ta_helfand_msd_dev(ta_ctx* ctx, const double* d_vel, const double* d_pos, const double* d_masses,
                       int64_t T, int64_t A, int D, int64_t ld_row, double scale, double* d_lagsum,
                       double* d_bp, int64_t ld_bp, void* stream) {
    return dev_entry(ctx, W_HELFAND, d_vel, d_pos, d_masses, T, A, D, ld_row, scale, d_lagsum, d_bp, ld_bp, stream);
}

int ta_vacf_fft_staged(ta_ctx* ctx, double* d_lagsum, double* d_bp, int64_t ld_bp, void* stream) {
    return staged_entry(ctx, W_FFT, nullptr, 1.0, d_lagsum, d_bp, ld_bp, stream);
}

int ta_vacf_direct_staged(ta_ctx* ctx, double* d_lagsum, double* d_bp, int64_t ld_bp, void* stream) {
    return staged_entry(ctx, W_DIRECT, nullptr, 1.0, d_lagsum, d_bp, ld_bp, stream);
}

int ta_helfand_msd_staged(ta_ctx* ctx, const double* d_masses, double scale, double* d_lagsum,
                          double* d_bp, int64_t ld_bp, void* stream) {
    return staged_entry(ctx, W_HELFAND, d_masses, scale, d_lagsum, d_bp, ld_bp, stream);
}

int ta_last_timing(ta_ctx* ctx, float* total_ms, float* main_kernel_ms) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (!ctx->timing_valid) return fail(ctx, TA_E_STATE, "no completed compute call to time");
    TA_HIP_TRY(ctx, hipEventSynchronize(ctx->ev[3]));
    float t = 0.f, m = 0.f;
    TA_HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[3]));
    TA_HIP_TRY(ctx, hipEventElapsedTime(&m, ctx->ev[1], ctx->ev[2]));
    if (total_ms) *total_ms = t;
    if (main_kernel_ms) *main_kernel_ms = m;
    return TA_OK;
}

int ta_timing_history(ta_ctx* ctx, int max_n, float* total_ms, float* main_kernel_ms, int* n_out) {
    if (!ctx || !n_out) return fail(ctx, TA_E_INVALID, "null argument");
    const long have = std::min<long>(ctx->n_calls, ta_ctx::kRing);
    const int n = (int)std::min<long>(have, std::max(0, max_n));
    for (int i = 0; i < n; ++i) {  // chronological: oldest of the last n first
        hipEvent_t* q = ctx->ring[(ctx->n_calls - n + i) % ta_ctx::kRing];
        TA_HIP_TRY(ctx, hipEventSynchronize(q[3]));
        float t = 0.f, m = 0.f;
        TA_HIP_TRY(ctx, hipEventElapsedTime(&t, q[0], q[3]));
        TA_HIP_TRY(ctx, hipEventElapsedTime(&m, q[1], q[2]));
        if (total_ms) total_ms[i] = t;
        if (main_kernel_ms) main_kernel_ms[i] = m;
    }
    *n_out = n;
    return TA_OK;
}

/* ------------------------------------------------- host-facing (blocking) */
static int host_compute(ta_ctx* ctx, int which, const double* h_masses, double scale,
                        double* h_ts, double* h_bp) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (!h_ts) return fail(ctx, TA_E_INVALID, "h_timeseries is NULL");
    const int need = which == W_HELFAND ? 2 : 1;
    if (ctx->st_nslabs < need) return fail(ctx, TA_E_STATE, "slabs have not been staged");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t T = ctx->st_T, A = ctx->st_A;
    int rc = ensure(ctx, ctx->out_lagsum, sizeof(double) * T);
    if (rc) return rc;
    double* d_bp = nullptr;
    if (h_bp) {
        if ((rc = ensure(ctx, ctx->out_bp, sizeof(double) * (size_t)T * A))) return rc;
        d_bp = (double*)ctx->out_bp.p;
    }
    double* d_ls = (double*)ctx->out_lagsum.p;
    const double* d_m = nullptr;
    if (which == W_HELFAND) {
        if (!h_masses) return fail(ctx, TA_E_INVALID, "h_masses is NULL");
        if ((rc = ensure(ctx, ctx->masses, sizeof(double) * A))) return rc;
        TA_HIP_TRY(ctx, hipMemcpyAsync(ctx->masses.p, h_masses, sizeof(double) * A, hipMemcpyHostToDevice,
                                       ctx->stream));
        d_m = (const double*)ctx->masses.p;
    }
    // With a by-particle array the device->host copy (8 GB at 10000 x 100000) is several times
    // the compute: atoms go in blocks, the copy of block c (a strided 2-D copy into the caller's
    // (n_frames, n_atoms) array, on a second stream) runs under the compute of block c + 1.
    const int64_t CH = ctx->opt_bp_block > 0 ? (ctx->opt_bp_block + 63) / 64 * 64 : 16384;
    if (h_bp && A >= 2 * CH) {
        const int64_t n_blocks = (A + CH - 1) / CH;
        if ((rc = ensure(ctx, ctx->out_lagsum, sizeof(double) * T * n_blocks))) return rc;
        d_ls = (double*)ctx->out_lagsum.p;
        if (!ctx->copy_stream) TA_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        std::vector<double> part((size_t)T * n_blocks);
        const int D = ctx->st_D;
        for (int64_t b = 0; b < n_blocks; ++b) {
            const int64_t lo = b * CH, hi = std::min(A, lo + CH);
            const int64_t pair_lo = lo * D / 2;  // lo is a multiple of 64: a pair boundary
            const double* v = ctx->d_slabs[0] + pair_lo * ctx->st_pitch * 2;
            const double* x = need == 2 ? ctx->d_slabs[1] + pair_lo * ctx->st_pitch * 2 : nullptr;
            if ((rc = compute_pm(ctx, which, v, x, d_m ? d_m + lo : nullptr, ctx->st_pitch, T, hi - lo, D, scale,
                                 d_ls + b * T, d_bp + lo, A, ctx->stream, true)))
                return rc;
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
            TA_HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_stage, 0));
            TA_HIP_TRY(ctx, hipMemcpy2DAsync(h_bp + lo, sizeof(double) * A, d_bp + lo, sizeof(double) * A,
                                             sizeof(double) * (hi - lo), T, hipMemcpyDeviceToHost,
                                             ctx->copy_stream));
        }
        TA_HIP_TRY(ctx, hipMemcpyAsync(part.data(), d_ls, sizeof(double) * T * n_blocks, hipMemcpyDeviceToHost,
                                       ctx->stream));
        TA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        TA_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
        const double n_at = (double)A;
        for (int64_t k = 0; k < T; ++k) {
            double sum = 0.0;
            for (int64_t b = 0; b < n_blocks; ++b) sum += part[(size_t)b * T + k];
            h_ts[k] = sum / n_at;
        }
        return TA_OK;
    }
    if ((rc = staged_entry(ctx, which, d_m, scale, d_ls, d_bp, A, (void*)ctx->stream))) return rc;
    TA_HIP_TRY(ctx, hipMemcpyAsync(h_ts, d_ls, sizeof(double) * T, hipMemcpyDeviceToHost, ctx->stream));
    if (h_bp)
        TA_HIP_TRY(ctx, hipMemcpyAsync(h_bp, d_bp, sizeof(double) * (size_t)T * A, hipMemcpyDeviceToHost,
                                       ctx->stream));
    TA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const double n_at = (double)A;  // mean over atoms (velocityautocorr.py:214,237)
    for (int64_t k = 0; k < T; ++k) h_ts[k] /= n_at;
    return TA_OK;
}

int ta_vacf_fft(ta_ctx* ctx, double* h_ts, double* h_bp) { return host_compute(ctx, W_FFT, nullptr, 1.0, h_ts, h_bp); }
int ta_vacf_direct(ta_ctx* ctx, double* h_ts, double* h_bp) { return host_compute(ctx, W_DIRECT, nullptr, 1.0, h_ts, h_bp); }
int ta_helfand_msd(ta_ctx* ctx, const double* h_masses, double scale, double* h_ts, double* h_bp) {
    return host_compute(ctx, W_HELFAND, h_masses, scale, h_ts, h_bp);
}

}  // extern "C"
