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
    cd* tw2 = nullptr;  // W_{2M}^n = exp(-i pi n / M), n < 2M
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
    std::string err;
    std::map<int, Tables> tables;
    std::map<long, LongTables> long_tables;  // keyed by M'
    DevBuf partial, spec, ts_partial, out_lagsum, out_bp, masses, f32_stage, stage_buf, long_scratch, helf_p, helf_small;
    // staging
    int64_t st_T = 0, st_A = 0;
    int st_D = 0, st_dtype = TA_F64, st_nslabs = 0;
    std::vector<void*> h_slabs;
    std::vector<double*> d_slabs;
    // timing
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool timing_valid = false;
    // options
    int64_t opt_fft_nwg = 0;
    int64_t opt_direct_nwg = 0;
    int64_t opt_fft_debug = 0;
    int64_t opt_direct_f32 = 0;
    int64_t opt_direct_groups = 0;
    int64_t opt_direct_chunk = 0;
    int64_t opt_helfand_fft = 0;
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

const PlanEntry* find_plan(int64_t n_frames) {
    const PlanEntry* best = nullptr;
    for (const auto* tab : {&plans_pow2(), &plans_five()})
        for (const auto& p : *tab)
            if (p.M >= n_frames && (!best || p.M < best->M)) best = &p;
    return best;
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
    TA_HIP_TRY(ctx, hipMemcpy(t.tw2, a.data(), sizeof(cd) * (4 * (size_t)M + 4), hipMemcpyHostToDevice));
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
    TA_HIP_TRY(ctx, hipMalloc((void**)&t.twL, sizeof(cd) * a.size()));
    TA_HIP_TRY(ctx, hipMemcpy(t.twL, a.data(), sizeof(cd) * a.size(), hipMemcpyHostToDevice));
    TA_HIP_TRY(ctx, hipMalloc((void**)&t.perm, sizeof(int) * perm.size()));
    TA_HIP_TRY(ctx, hipMemcpy(t.perm, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice));
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
int fft_long_impl(ta_ctx* ctx, const double* d_vel, int64_t T, int64_t A, int D, int64_t ld_row,
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
    TA_HIP_TRY(ctx, launch_fft_long_accum(M, (int)nwg, st, d_vel, ld_row, (int)T, A * D, Rout, tb.tw2,
                                          lt.twL, (double*)ctx->partial.p, (cd*)ctx->long_scratch.p));
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
                const double* d_masses, int64_t T, int64_t A, int D, int64_t ld_row, double scale,
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
    TA_HIP_TRY(ctx, hipMemsetAsync(ctx->ts_partial.p, 0, sizeof(double) * rows * T, st));
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
    TA_HIP_TRY(ctx, launch_direct(mode, f32, L, d_vel, d_pos, d_masses, ld_row, (int)T, A, D, scale, d_bp,
                                  ld_bp, (double*)ctx->ts_partial.p, (int)nwg, nt, lds, stage_buf,
                                  gnt, st));
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
    TA_HIP_TRY(ctx, launch_sum_partials((const double*)ctx->ts_partial.p, (int)rows, T, d_lagsum, st));
    return TA_OK;
}

}  // namespace

extern "C" {

int ta_abi_version(void) { return 1; }

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
    TA_HIP_TRY(ctx, hipSetDevice(device));
    hipDeviceProp_t prop;
    TA_HIP_TRY(ctx, hipGetDeviceProperties(&prop, device));
    ctx->n_cu = prop.multiProcessorCount;
    TA_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    for (auto& ev : ctx->ev) TA_HIP_TRY(ctx, hipEventCreate(&ev));
    *out = ctx;
    return TA_OK;
}

int ta_stage_free(ta_ctx* ctx) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    hipSetDevice(ctx->device);
    for (void* h : ctx->h_slabs)
        if (h) hipHostFree(h);
    for (double* d : ctx->d_slabs)
        if (d) hipFree(d);
    ctx->h_slabs.clear();
    ctx->d_slabs.clear();
    ctx->st_nslabs = 0;
    ctx->st_T = ctx->st_A = 0;
    return TA_OK;
}

int ta_ctx_destroy(ta_ctx* ctx) {
    if (!ctx) return TA_OK;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    ta_stage_free(ctx);
    for (auto& kv : ctx->tables) {
        hipFree(kv.second.tw2);
    }
    for (auto& kv : ctx->long_tables) {
        hipFree(kv.second.twL);
        hipFree(kv.second.perm);
    }
    for (DevBuf* b : {&ctx->partial, &ctx->spec, &ctx->ts_partial, &ctx->out_lagsum, &ctx->out_bp,
                      &ctx->masses, &ctx->f32_stage, &ctx->stage_buf, &ctx->long_scratch, &ctx->helf_p,
                      &ctx->helf_small})
        if (b->p) hipFree(b->p);
    for (auto& ev : ctx->ev)
        if (ev) hipEventDestroy(ev);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
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
    else return fail(ctx, TA_E_INVALID, std::string("unknown option ") + key);
    return TA_OK;
}

int ta_fft_plan_info(int64_t n_frames, int64_t* m_out, int* n_threads, int* n_stages) {
    const PlanEntry* p = find_plan(n_frames);
    int long_M = 0, long_R = 0;
    if (!p && fft_long_choose((long)n_frames, &long_M, &long_R)) {
        // outer radix step + on-chip transform (lag sums only; fft_long.hip)
        const PlanEntry* q = plan_of_length(long_M);
        if (m_out) *m_out = (int64_t)long_M * long_R;
        if (n_threads) *n_threads = q ? q->NT : 0;
        if (n_stages) *n_stages = q ? q->S + 1 : 0;
        return TA_OK;
    }
    if (!p) return fail(nullptr, TA_E_UNSUPPORTED, "n_frames exceeds the largest FFT plan");
    if (m_out) *m_out = p->M;
    if (n_threads) *n_threads = p->NT;
    if (n_stages) *n_stages = p->S;
    return TA_OK;
}

/* ------------------------------------------------------------------ staging */
int ta_stage_alloc(ta_ctx* ctx, int64_t n_frames, int64_t n_atoms, int dim, int dtype, int n_slabs,
                   void** h_slabs) {
    int rc = check_shape(ctx, n_frames, n_atoms, dim, n_atoms * dim);
    if (rc) return rc;
    if (!h_slabs || n_slabs < 1 || n_slabs > 4) return fail(ctx, TA_E_INVALID, "bad slab count");
    if (dtype != TA_F32 && dtype != TA_F64) return fail(ctx, TA_E_INVALID, "bad dtype");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ta_stage_free(ctx);
    const size_t n = (size_t)n_frames * n_atoms * dim;
    const size_t esz = dtype == TA_F32 ? 4 : 8;
    for (int i = 0; i < n_slabs; ++i) {
        void* h = nullptr;
        double* d = nullptr;
        hipError_t e = hipHostMalloc(&h, n * esz, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void**)&d, n * sizeof(double));
        if (e != hipSuccess) {
            if (h) hipHostFree(h);
            ta_stage_free(ctx);
            return fail(ctx, TA_E_NOMEM, std::string("staging allocation failed: ") + hipGetErrorString(e));
        }
        memset(h, 0, n * esz);  // the reference starts from np.zeros (velocityautocorr.py:150)
        ctx->h_slabs.push_back(h);
        ctx->d_slabs.push_back(d);
        h_slabs[i] = h;
    }
    ctx->st_T = n_frames;
    ctx->st_A = n_atoms;
    ctx->st_D = dim;
    ctx->st_dtype = dtype;
    ctx->st_nslabs = n_slabs;
    return TA_OK;
}

int ta_stage_commit(ta_ctx* ctx, int64_t frame_lo, int64_t frame_hi) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (ctx->st_nslabs == 0) return fail(ctx, TA_E_STATE, "ta_stage_alloc has not been called");
    if (frame_lo < 0 || frame_hi > ctx->st_T || frame_lo > frame_hi)
        return fail(ctx, TA_E_INVALID, "frame range out of bounds");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t row = (size_t)ctx->st_A * ctx->st_D;
    const size_t n = (size_t)(frame_hi - frame_lo) * row;
    if (n == 0) return TA_OK;
    for (int i = 0; i < ctx->st_nslabs; ++i) {
        double* dst = ctx->d_slabs[i] + (size_t)frame_lo * row;
        if (ctx->st_dtype == TA_F64) {
            const double* src = (const double*)ctx->h_slabs[i] + (size_t)frame_lo * row;
            TA_HIP_TRY(ctx, hipMemcpyAsync(dst, src, n * 8, hipMemcpyHostToDevice, ctx->stream));
        } else {
            const size_t chunk = std::min(n, (size_t)64 << 20);  // elements per staging round
            int rc = ensure(ctx, ctx->f32_stage, chunk * 4);
            if (rc) return rc;
            const float* src = (const float*)ctx->h_slabs[i] + (size_t)frame_lo * row;
            for (size_t off = 0; off < n; off += chunk) {
                const size_t m = std::min(chunk, n - off);
                TA_HIP_TRY(ctx, hipMemcpyAsync(ctx->f32_stage.p, src + off, m * 4,
                                               hipMemcpyHostToDevice, ctx->stream));
                TA_HIP_TRY(ctx, launch_widen_f32((const float*)ctx->f32_stage.p, dst + off, (long)m,
                                                 ctx->stream));
            }
        }
    }
    return TA_OK;
}

int ta_stage_device(ta_ctx* ctx, int slab, double** d_slab) {
    if (!ctx || !d_slab) return fail(ctx, TA_E_INVALID, "null argument");
    if (slab < 0 || slab >= ctx->st_nslabs) return fail(ctx, TA_E_INVALID, "no such slab");
    *d_slab = ctx->d_slabs[slab];
    return TA_OK;
}

/* --------------------------------------------------------- device compute */
int ta_vacf_fft_dev(ta_ctx* ctx, const double* d_vel, int64_t T, int64_t A, int D, int64_t ld_row,
                    double* d_lagsum, double* d_bp, int64_t ld_bp, void* stream) {
    int rc = check_shape(ctx, T, A, D, ld_row);
    if (rc) return rc;
    if (!d_vel || !d_lagsum) return fail(ctx, TA_E_INVALID, "null device pointer");
    if (d_bp && ld_bp < A) return fail(ctx, TA_E_INVALID, "ld_bp smaller than n_atoms");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the legacy default stream, as for any HIP call
    ctx->timing_valid = false;
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[0], st));
    const PlanEntry* plan = find_plan(T);
    int long_M = 0, long_R = 0;
    if (!plan && !d_bp && fft_long_choose((long)T, &long_M, &long_R)) {
        // longer than the largest on-chip transform, lag sums only: outer radix step while
        // the column is read, on-chip transforms of the derived series (fft_long.hip)
        rc = fft_long_impl(ctx, d_vel, T, A, D, ld_row, d_lagsum, st, long_M, long_R);
        if (rc) return rc;
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[3], st));
        ctx->timing_valid = true;
        return TA_OK;
    }
    if (!plan) {
        // by-particle output (or longer than 16 x 10240 frames): the direct correlator computes
        // the same quantity (velocityautocorr.py:217-238 == :208-215 mathematically)
        rc = direct_impl(ctx, MODE_VACF, d_vel, nullptr, nullptr, T, A, D, ld_row, 1.0, d_lagsum,
                         d_bp, ld_bp, st);
        if (rc) return rc;
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[3], st));
        ctx->timing_valid = true;
        return TA_OK;
    }
    Tables tb;
    if ((rc = get_tables(ctx, plan->M, plan->R_first, &tb))) return rc;
    FftArgs a{};
    a.vel = d_vel;
    a.ld_row = ld_row;
    a.pair_stride = 2;
    a.flags = (int)ctx->opt_fft_debug;
    a.T = (int)T;
    a.n_cols = A * D;
    a.n_atoms = A;
    a.D = D;
    a.tw2 = tb.tw2;
    if (!d_bp) {
        // the all-16-byte-loads instantiation needs every pair complete: an odd column count
        // (possible with an even ld_row when the shard is a column block of a wider slab) takes
        // the general one, which still uses 16-byte loads wherever a pair is aligned
        const bool vec = (ld_row % 2 == 0) && ((uintptr_t)d_vel % 16 == 0) && ((A * D) % 2 == 0);
        const int64_t n_pairs = (A * D + 1) / 2;
        int64_t nwg = ctx->opt_fft_nwg > 0 ? ctx->opt_fft_nwg
                                           : (int64_t)ctx->n_cu * plan->max_wg_per_cu(vec ? 0 : 1);
        nwg = std::max<int64_t>(1, std::min(nwg, n_pairs));
        if (nwg >= 8) nwg -= nwg % 8;  // XCD-aware walk
        const size_t acc_blk = (size_t)((plan->K_last * plan->R_last + 1) / 2) * 2 * plan->NT;
        const size_t acc_bytes = sizeof(double) * (size_t)nwg * 2 * acc_blk;
        if ((rc = ensure(ctx, ctx->partial, acc_bytes))) return rc;
        const int n_slices = (int)std::min<int64_t>(4, nwg);
        if ((rc = ensure(ctx, ctx->spec, sizeof(double) * 2 * plan->M * n_slices))) return rc;
        a.partial = (double*)ctx->partial.p;
        TA_HIP_TRY(ctx, hipMemsetAsync(a.partial, 0, acc_bytes, st));
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
        TA_HIP_TRY(ctx, plan->accum(vec, (int)nwg, st, a));
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
        TA_HIP_TRY(ctx, launch_sum_partials_perm(a.partial, (int)nwg, plan->M, plan->NT, plan->R_last,
                                                 plan->K_last, plan->TASKS_last,
                                                 (double*)ctx->spec.p, n_slices, st));
        a.spec = (const double*)ctx->spec.p;
        a.n_slices = n_slices;
        a.lagsum = d_lagsum;
        TA_HIP_TRY(ctx, plan->finalize(st, a));
    } else {
        int64_t nwg = ctx->opt_fft_nwg > 0 ? ctx->opt_fft_nwg
                                           : (int64_t)ctx->n_cu * plan->max_wg_per_cu(2);
        nwg = std::max<int64_t>(1, std::min(nwg, A));
        if (nwg >= 8) nwg -= nwg % 8;  // XCD-aware walk
        // accumulator swap blocks of the small plans (as in the timeseries path)
        const size_t acc_blk = (size_t)((plan->K_last * plan->R_last + 1) / 2) * 2 * plan->NT;
        const size_t acc_bytes = sizeof(double) * (size_t)nwg * 2 * acc_blk;
        if ((rc = ensure(ctx, ctx->partial, acc_bytes))) return rc;
        a.partial = (double*)ctx->partial.p;
        a.by_particle = d_bp;
        a.ld_bp = ld_bp;
        TA_HIP_TRY(ctx, hipMemsetAsync(a.partial, 0, acc_bytes, st));
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
        TA_HIP_TRY(ctx, plan->by_particle((int)nwg, st, a));
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
        // lag sums = row sums of the by-particle array (velocityautocorr.py:214)
        TA_HIP_TRY(ctx, launch_row_sums(d_bp, T, A, ld_bp, d_lagsum, st));
    }
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[3], st));
    ctx->timing_valid = true;
    return TA_OK;
}

int ta_vacf_direct_dev(ta_ctx* ctx, const double* d_vel, int64_t T, int64_t A, int D,
                       int64_t ld_row, double* d_lagsum, double* d_bp, int64_t ld_bp, void* stream) {
    int rc = check_shape(ctx, T, A, D, ld_row);
    if (rc) return rc;
    if (!d_vel || !d_lagsum) return fail(ctx, TA_E_INVALID, "null device pointer");
    if (d_bp && ld_bp < A) return fail(ctx, TA_E_INVALID, "ld_bp smaller than n_atoms");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the legacy default stream, as for any HIP call
    ctx->timing_valid = false;
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[0], st));
    rc = direct_impl(ctx, MODE_VACF, d_vel, nullptr, nullptr, T, A, D, ld_row, 1.0, d_lagsum, d_bp,
                     ld_bp, st);
    if (rc) return rc;
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[3], st));
    ctx->timing_valid = true;
    return TA_OK;
}

int ta_helfand_msd_dev(ta_ctx* ctx, const double* d_vel, const double* d_pos, const double* d_masses,
                       int64_t T, int64_t A, int D, int64_t ld_row, double scale, double* d_lagsum,
                       double* d_bp, int64_t ld_bp, void* stream) {
    int rc = check_shape(ctx, T, A, D, ld_row);
    if (rc) return rc;
    if (!d_vel || !d_pos || !d_masses || !d_lagsum)
        return fail(ctx, TA_E_INVALID, "null device pointer");
    if (d_bp && ld_bp < A) return fail(ctx, TA_E_INVALID, "ld_bp smaller than n_atoms");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the legacy default stream, as for any HIP call
    ctx->timing_valid = false;
    if (ctx->opt_helfand_fft && !d_bp && T >= 2) {
        // option "helfand_fft" (lag sums only): S1 from prefix sums, S2 = FFT lag sums of the
        // product slab P = (m v) x (helfand_fft.hip)
        const size_t n = (size_t)T * A * D;
        if ((rc = ensure(ctx, ctx->helf_p, sizeof(double) * n))) return rc;
        if ((rc = ensure(ctx, ctx->helf_small, sizeof(double) * (3 * (size_t)T + 1)))) return rc;
        double* P = (double*)ctx->helf_p.p;
        double* Q = (double*)ctx->helf_small.p;
        double* S2 = Q + T;
        double* C = S2 + T;
        TA_HIP_TRY(ctx, launch_helfand_product(d_vel, d_pos, d_masses, ld_row, T, A * D, D, P, Q, st));
        if ((rc = ta_vacf_fft_dev(ctx, P, T, A, D, A * D, S2, nullptr, 0, stream))) return rc;
        TA_HIP_TRY(ctx, launch_helfand_combine(Q, S2, C, (int)T, scale / (double)D, d_lagsum, st));
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[3], st));
        return TA_OK;  // timing: ev[0..2] set by the FFT call, ev[3] after the combine
    }
    if (ctx->opt_helfand_fft && d_bp && T >= 2 && find_plan(T)) {
        // the same per atom (on-chip FFT lengths only; longer trajectories take the direct
        // correlator below): bp <- FFT by-particle autocorrelation of P, then S1 - 2 S2 in place
        const size_t n = (size_t)T * A * D;
        if ((rc = ensure(ctx, ctx->helf_p, sizeof(double) * n))) return rc;
        if ((rc = ensure(ctx, ctx->helf_small, sizeof(double) * ((size_t)T + 1) * A))) return rc;
        double* P = (double*)ctx->helf_p.p;
        double* Ca = (double*)ctx->helf_small.p;
        TA_HIP_TRY(ctx, launch_helfand_product_bp(d_vel, d_pos, d_masses, ld_row, T, A, D, P, Ca, st));
        if ((rc = ta_vacf_fft_dev(ctx, P, T, A, D, A * D, d_lagsum, d_bp, ld_bp, stream))) return rc;
        TA_HIP_TRY(ctx, launch_helfand_combine_bp(Ca, A, (int)T, scale / (double)D, d_bp, ld_bp, st));
        TA_HIP_TRY(ctx, launch_row_sums(d_bp, T, A, ld_bp, d_lagsum, st));
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[3], st));
        return TA_OK;
    }
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[0], st));
    rc = direct_impl(ctx, MODE_HELFAND, d_vel, d_pos, d_masses, T, A, D, ld_row, scale, d_lagsum,
                     d_bp, ld_bp, st);
    if (rc) return rc;
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[3], st));
    ctx->timing_valid = true;
    return TA_OK;
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

/* ------------------------------------------------- host-facing (blocking) */
static int host_compute(ta_ctx* ctx, int which, const double* h_masses, double scale,
                        double* h_ts, double* h_bp) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (!h_ts) return fail(ctx, TA_E_INVALID, "h_timeseries is NULL");
    const int need = which == 2 ? 2 : 1;
    if (ctx->st_nslabs < need) return fail(ctx, TA_E_STATE, "slabs have not been staged");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t T = ctx->st_T, A = ctx->st_A;
    const int D = ctx->st_D;
    int rc = ensure(ctx, ctx->out_lagsum, sizeof(double) * T);
    if (rc) return rc;
    double* d_bp = nullptr;
    if (h_bp) {
        if ((rc = ensure(ctx, ctx->out_bp, sizeof(double) * (size_t)T * A))) return rc;
        d_bp = (double*)ctx->out_bp.p;
    }
    double* d_ls = (double*)ctx->out_lagsum.p;
    if (which == 0)
        rc = ta_vacf_fft_dev(ctx, ctx->d_slabs[0], T, A, D, A * D, d_ls, d_bp, A, (void*)ctx->stream);
    else if (which == 1)
        rc = ta_vacf_direct_dev(ctx, ctx->d_slabs[0], T, A, D, A * D, d_ls, d_bp, A, (void*)ctx->stream);
    else {
        if (!h_masses) return fail(ctx, TA_E_INVALID, "h_masses is NULL");
        if ((rc = ensure(ctx, ctx->masses, sizeof(double) * A))) return rc;
        TA_HIP_TRY(ctx, hipMemcpyAsync(ctx->masses.p, h_masses, sizeof(double) * A,
                                       hipMemcpyHostToDevice, ctx->stream));
        rc = ta_helfand_msd_dev(ctx, ctx->d_slabs[0], ctx->d_slabs[1], (const double*)ctx->masses.p,
                                T, A, D, A * D, scale, d_ls, d_bp, A, (void*)ctx->stream);
    }
    if (rc) return rc;
    TA_HIP_TRY(ctx, hipMemcpyAsync(h_ts, d_ls, sizeof(double) * T, hipMemcpyDeviceToHost, ctx->stream));
    if (h_bp)
        TA_HIP_TRY(ctx, hipMemcpyAsync(h_bp, d_bp, sizeof(double) * (size_t)T * A,
                                       hipMemcpyDeviceToHost, ctx->stream));
    TA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const double n_at = (double)A;  // mean over atoms (velocityautocorr.py:214,237)
    for (int64_t k = 0; k < T; ++k) h_ts[k] /= n_at;
    return TA_OK;
}

int ta_vacf_fft(ta_ctx* ctx, double* h_ts, double* h_bp) { return host_compute(ctx, 0, nullptr, 1.0, h_ts, h_bp); }
int ta_vacf_direct(ta_ctx* ctx, double* h_ts, double* h_bp) { return host_compute(ctx, 1, nullptr, 1.0, h_ts, h_bp); }
int ta_helfand_msd(ta_ctx* ctx, const double* h_masses, double scale, double* h_ts, double* h_bp) {
    return host_compute(ctx, 2, h_masses, scale, h_ts, h_bp);
}

}  // extern "C"
