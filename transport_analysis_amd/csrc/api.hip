// api.hip — C-ABI entry points of libta_hip.so (declared in include/ta_hip.h).
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ta_hip.h"
#include "cpu_backend.hpp"
#include "direct_kernels.hpp"
#include "ta_internal.hpp"

using namespace ta;

namespace {

thread_local std::string g_tls_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct ta_ctx {
    // a CPU context (ta_ctx_create(TA_DEVICE_CPU, ...): the opt-in backend of cpu_backend.cpp) owns host slabs only;
    // no HIP call is ever made on its behalf and every device-facing entry point rejects it (TA_NO_CPU)
    bool is_cpu = false;
    ta::cpu::State cpu;
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // device->host copies of by-particle blocks (host_compute)
    std::string err;
    std::map<int, cd*> wf_tables;  // wfft.hip twiddle tables, keyed by 64 * R0 + R
    DevBuf partial, spec, ts_partial, out_lagsum, out_bp, masses, bounce, stage_buf, helf_p, helf_small;
    DevBuf pm_in[2];  // pair-major copies of frame-major *_dev inputs
    DevBuf bp_scratch;  // atom-major by-particle results before the transposition
    DevBuf bp_spec;     // per-atom power spectra of one block of atoms (two-kernel by-particle path)
    DevBuf unit_counter;  // k_band_bp_vacf's work counter
    // staging: pinned host slabs keep the reference's (n_frames, n_atoms, dim) layout, the
    // device slabs are pair-major (layout.hip) with st_pitch rows per column pair
    int64_t st_T = 0, st_A = 0, st_pitch = 0;
    int st_D = 0, st_dtype = TA_F64, st_nslabs = 0;
    bool st_dev_f32 = false;  // device slabs hold float32 elements ("stage_device_f32")
    std::vector<void*> h_slabs;
    std::vector<HostBlock> h_blocks;  // the mapping behind h_slabs[i] (base == NULL: a hipHostMalloc block, the fallback)
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
    // kernel timeline of the last compute call ("timeline" option): an event before every
    // launch, a last one at the end; segment i = [mark i, mark i + 1) belongs to name i
    struct Mark {
        const char* name;
        hipEvent_t ev;
    };
    std::vector<Mark> marks;
    std::vector<hipEvent_t> mark_pool;
    size_t marks_used = 0;
    int64_t opt_timeline = 0;
    // staging: second landing buffer + stream, so that a piece crosses PCIe while the one before
    // it is transposed
    DevBuf bounce2;
    hipStream_t relayout_stream = nullptr;
    hipEvent_t ev_piece[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    // options
    int64_t opt_fft_nwg = 0;
    int64_t opt_direct_nwg = 0;
    int64_t opt_direct_f32 = 0;
    int64_t opt_direct_groups = 0;
    int64_t opt_direct_chunk = 0;
    int64_t opt_direct_mfma = 1;  // windowed VACF lag sums without the by-particle array: matrix-core kernel
    int64_t opt_helfand_fft = 0;
    int64_t opt_bp_block = 0;
    int64_t opt_bp_spec_atoms = 0;
    int64_t opt_bp_prefetch = 2;
    // "short_max": trajectories of up to this many frames (<= 64) take the register-resident kernels of short_kernels.hpp
    // wherever float64 slabs are asked for a by-particle array or an O(T^2) form (0: never); "short_lags_max": the FFT
    // path's lag sums alone as well, up to this many frames (per 12 GB: 2.1 against 3.9 ms at 32 frames, 3.8 against 4.6 at
    // 48, 4.1 against 3.8 at 64: profiles/r06_short.txt)
    int64_t opt_short_max = 64, opt_short_lags_max = 48;
    int64_t opt_mid_max = 512, opt_mid_all = 0, opt_mid_ncl = 0;  // k_mid (mid_kernels.hpp) under "direct_mfma" 1: see direct_impl
    int64_t opt_direct_subwave = 1;  // "direct_subwave": k_direct's column groups may be 16 or 32 lanes (under ~640 frames)
    int64_t opt_stage_device_f32 = 0;
    int64_t opt_fail_alloc_after = 0, opt_fail_throw_after = 0;  // test hooks of ensure()
    // ta_stage_commit hands its frame range to a worker thread that makes the HIP calls (copies in pieces, the
    // transposition launches): the caller's frame loop never waits on the runtime — which it did, for as long
    // as another thread's hipHostMalloc of the by-particle result held the runtime's lock (0.17 s of a 0.6 s
    // loop at 10000 x 50000 x 3).  Everything that touches the slabs or the streams joins the queue first
    // (commit_flush); an error of a queued commit is returned there.
    int64_t opt_async_commit = 1;
    int64_t opt_lock_ahead = 1;  // "lock_ahead": the commit worker page-locks the chunks behind the one it committed
    std::thread cq_thread;
    std::mutex cq_m;
    std::condition_variable cq_cv;
    std::deque<std::pair<int64_t, int64_t>> cq;
    bool cq_stop = false, cq_busy = false;
    int cq_rc = TA_OK;
    std::string cq_err;
};

int commit_flush(ta_ctx* ctx);  // (defined with ta_stage_commit)
static void commit_stop(ta_ctx* ctx);
static void commit_worker(ta_ctx* ctx);

namespace {

// (the commit worker thread reports errors too: the context's message is written and read under a lock)
std::mutex g_err_m;
int fail(ta_ctx* ctx, int code, const std::string& msg) noexcept {
    try {  // (a failing copy of the message must not turn an error return into an exception)
        if (ctx) {
            std::lock_guard<std::mutex> lk(g_err_m);
            ctx->err = msg;
        }
        g_tls_error = msg;
    } catch (...) {
    }
    return code;
}

#define TA_NO_CPU(ctx)                                                                                                   \
    do {                                                                                                                  \
        if ((ctx) && (ctx)->is_cpu)                                                                                       \
            return fail(ctx, TA_E_UNSUPPORTED, "not available on the CPU backend (device pointers, streams and kernel timings belong to GPU contexts)"); \
    } while (0)

#define TA_HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(ctx, (_e == hipErrorOutOfMemory) ? TA_E_NOMEM : TA_E_HIP,              \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                    \
    } while (0)

int ensure(ta_ctx* ctx, DevBuf& b, size_t bytes) {
    // test hooks ("fail_alloc_after" / "fail_throw_after" n): the n-th call from now throws what a failing host allocation /
    // any other library exception would, so that the tests can see the C boundary turn it into a status
    if (ctx->opt_fail_alloc_after > 0 && --ctx->opt_fail_alloc_after == 0) throw std::bad_alloc();
    if (ctx->opt_fail_throw_after > 0 && --ctx->opt_fail_throw_after == 0) throw std::runtime_error("fail_throw_after");
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

// timeline mark: the work queued on `st` from here to the next mark is `name`'s
void tl_mark(ta_ctx* ctx, const char* name, hipStream_t st) {
    if (!ctx->opt_timeline) return;
    if (ctx->marks_used == ctx->mark_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return;
        ctx->mark_pool.push_back(e);
    }
    hipEvent_t e = ctx->mark_pool[ctx->marks_used++];
    if (hipEventRecord(e, st) == hipSuccess) ctx->marks.push_back({name, e});
}
void tl_reset(ta_ctx* ctx) {
    ctx->marks.clear();
    ctx->marks_used = 0;
}

inline int64_t pm_pitch(int64_t n_frames) { return (n_frames + 7) / 8 * 8; }
inline size_t pm_bytes(int64_t n_frames, int64_t n_cols, bool f32 = false) {
    return (size_t)((n_cols + 1) / 2) * (size_t)pm_pitch(n_frames) * (f32 ? 8 : 16);
}

int get_wf_table(ta_ctx* ctx, int R0, int R, cd** out) {
    const int key = 64 * R0 + R;
    auto it = ctx->wf_tables.find(key);
    if (it != ctx->wf_tables.end()) {
        *out = it->second;
        return TA_OK;
    }
    std::vector<cd> a(wfft_table_elems(R0, R));
    wfft_fill_table(R0, R, a.data());
    cd* d = nullptr;
    TA_HIP_TRY(ctx, hipMalloc((void**)&d, sizeof(cd) * a.size()));
    hipError_t e = hipMemcpy(d, a.data(), sizeof(cd) * a.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(d);
        return fail(ctx, TA_E_HIP, std::string("twiddle table upload: ") + hipGetErrorString(e));
    }
    ctx->wf_tables[key] = d;
    *out = d;
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

// Short trajectories (short_kernels.hpp): a lane per column, every lag in its registers; the by-particle array is written in
// place, the lag sums leave as one row per wave
bool short_applies(const ta_ctx* ctx, int64_t T) { return T <= ctx->opt_short_max && T <= short_max_frames(); }

int short_impl(ta_ctx* ctx, int mode, const double* d_vel, const double* d_pos, const double* d_masses, int64_t T, int64_t A,
               int D, int64_t pitch, double scale, double* d_lagsum, double* d_bp, int64_t ld_bp, hipStream_t st) {
    const int nwg = short_grid(ctx->n_cu, mode, (int)T, A, D, d_bp != nullptr);
    const int rows = nwg * short_waves();
    int rc = ensure(ctx, ctx->ts_partial, sizeof(double) * (size_t)rows * T);
    if (rc) return rc;
    tl_mark(ctx, "k_short", st);
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
    TA_HIP_TRY(ctx, launch_short(mode, nwg, d_vel, d_pos, d_masses, pitch, (int)T, A, D,
                                 mode == MODE_HELFAND ? scale / (double)D : 1.0, d_bp, ld_bp, (double*)ctx->ts_partial.p, st));
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
    tl_mark(ctx, "k_sum_partials", st);
    TA_HIP_TRY(ctx, launch_sum_partials((const double*)ctx->ts_partial.p, rows, T, d_lagsum, st));
    return TA_OK;
}

// 65 ... 512 frames (mid_kernels.hpp): a lane per (column, pair of 16-lag blocks), sliding window in registers
int mid_impl(ta_ctx* ctx, int mode, const double* d_vel, const double* d_pos, const double* d_masses, int64_t T, int64_t A, int D,
             int64_t pitch, double scale, double* d_lagsum, double* d_bp, int64_t ld_bp, hipStream_t st) {
    const int nwg = mid_grid(ctx->n_cu, mode, (int)T, A, D, (int)ctx->opt_mid_ncl);
    int rc = ensure(ctx, ctx->ts_partial, sizeof(double) * (size_t)nwg * T);
    if (rc) return rc;
    tl_mark(ctx, "k_mid", st);
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
    TA_HIP_TRY(ctx, launch_mid(mode, nwg, d_vel, d_pos, d_masses, pitch, (int)T, A, D, mode == MODE_HELFAND ? scale / (double)D : 1.0,
                               d_bp, ld_bp, (double*)ctx->ts_partial.p, (int)ctx->opt_mid_ncl, st));
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
    tl_mark(ctx, "k_sum_partials", st);
    TA_HIP_TRY(ctx, launch_sum_partials((const double*)ctx->ts_partial.p, nwg, T, d_lagsum, st));
    return TA_OK;
}

int direct_impl(ta_ctx* ctx, int mode, const void* d_vel, const void* d_pos,
                const double* d_masses, int64_t T, int64_t A, int D, int64_t pitch, double scale,
                double* d_lagsum, double* d_bp, int64_t ld_bp, hipStream_t st, bool src_f32 = false) {
    const bool f32 = ctx->opt_direct_f32 != 0;  // src_f32 only comes with it (compute_pm)
    if (!f32 && !src_f32 && ctx->opt_direct_mfma == 1 && short_applies(ctx, T))  // ("direct_mfma" 0 / 3 force a form)
        return short_impl(ctx, mode, (const double*)d_vel, (const double*)d_pos, d_masses, T, A, D, pitch, scale, d_lagsum, d_bp,
                          ld_bp, st);
    // k_mid where it wins (profiles/r06_direct_mid_sweep.txt): the windowed VACF from 97 to 512 frames (9.4 against 16.5 ms per
    // 12 GB at 128 frames with the by-particle array, 21.9 against 25.8 at 512), Helfand from 97 to 128 (18 against 23);
    // "mid_max" 0: never; "mid_all" 1: wherever the kernel can run (65 ... 512 frames, both quantities: the parity tests)
    if (!f32 && !src_f32 && ctx->opt_direct_mfma == 1 && T > 64 && T <= ctx->opt_mid_max && T <= mid_max_frames() &&
        (ctx->opt_mid_all || (T >= 97 && (mode == MODE_VACF || T <= 128))))
        return mid_impl(ctx, mode, (const double*)d_vel, (const double*)d_pos, d_masses, T, A, D, pitch, scale, d_lagsum, d_bp, ld_bp,
                        st);
    // The O(T^2) correlators run on the matrix cores wherever that wins: FP64 (bandbp_kernels.hpp) and, for the float32
    // option's Helfand forms, FP32 (band32tp_kernels.hpp: P rounded once to float32 like the float32 vector kernel's staged
    // values, float32 products, float64 accumulation) -- the k-slots of the MFMA filled from the time axis.  "direct_mfma":
    // 1 = by trajectory length (these kernels pay a ring fill and an epilogue per particle (block) and lag group; the vector
    // kernel, whose column groups are 8 - 32 lanes under ~640 frames, wins the windowed VACF up to 512 frames, Helfand float64
    // up to 351, float32 up to 447: profiles/r06_direct_mid_sweep.txt, 12 GB of input at every length),
    // 3 = always, 0 = vector kernels (the parity tests' second opinion).  (The column-packed forms of rounds 4-5 --
    // "direct_mfma" 2, inline-assembly LDS-DMA -- live under tools/band/ since round 6.)
    const bool band_ok = !d_bp && !f32 && !src_f32 && ctx->opt_direct_mfma && T < ((int64_t)1 << 24);
    const int mf = (int)ctx->opt_direct_mfma;
    auto time_packed = [&](int64_t from_frames) { return mf == 3 || (mf == 1 && T >= from_frames); };
    if (!d_bp && f32 && mode == MODE_HELFAND && time_packed(448) && T < ((int64_t)1 << 24)) {
        const int64_t n_cols = A * D;
        if (ensure(ctx, ctx->helf_p, pm_bytes(T, n_cols, true)) == TA_OK &&
            ensure(ctx, ctx->bp_scratch, sizeof(double) * band_bp_helf_partial_doubles(ctx->n_cu, (int)T, A)) == TA_OK &&
            ensure(ctx, ctx->unit_counter, 64) == TA_OK) {
            tl_mark(ctx, "k_helfand_product32", st);
            TA_HIP_TRY(ctx, launch_helfand_product32(d_vel, d_pos, src_f32, d_masses, pitch, T, n_cols, D, (float*)ctx->helf_p.p, st));
            tl_mark(ctx, "k_band32_tp", st);
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
            TA_HIP_TRY(ctx, launch_band32_tp_lags(ctx->n_cu, (const float*)ctx->helf_p.p, pitch, (int)T, A, D, scale / (double)D,
                                                  (double*)ctx->bp_scratch.p, (unsigned long long*)ctx->unit_counter.p, d_lagsum, st));
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
            return TA_OK;
        }
        (void)hipGetLastError();  // out of memory for the product slab: the vector kernel needs none
    }
    // ... and with the by-particle array
    if (d_bp && f32 && mode == MODE_HELFAND && time_packed(448) && T < ((int64_t)1 << 24)) {
        const int64_t n_cols = A * D, Tp = pm_pitch(T), n_tiles = (A + 63) / 64;
        if (ensure(ctx, ctx->helf_p, pm_bytes(T, n_cols, true)) == TA_OK &&
            ensure(ctx, ctx->bp_scratch, sizeof(double) * (size_t)A * Tp) == TA_OK &&
            ensure(ctx, ctx->ts_partial, sizeof(double) * (size_t)n_tiles * T) == TA_OK &&
            ensure(ctx, ctx->unit_counter, 64) == TA_OK) {
            tl_mark(ctx, "k_helfand_product32", st);
            TA_HIP_TRY(ctx, launch_helfand_product32(d_vel, d_pos, src_f32, d_masses, pitch, T, n_cols, D, (float*)ctx->helf_p.p, st));
            tl_mark(ctx, "k_band32_tp", st);
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
            TA_HIP_TRY(ctx, launch_band32_tp_bp(ctx->n_cu, (const float*)ctx->helf_p.p, pitch, (int)T, A, D, scale / (double)D,
                                                (double*)ctx->bp_scratch.p, Tp, (unsigned long long*)ctx->unit_counter.p, st));
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
            tl_mark(ctx, "k_bp_transpose", st);
            TA_HIP_TRY(ctx, launch_bp_transpose((const double*)ctx->bp_scratch.p, Tp, A, T, d_bp, ld_bp, (double*)ctx->ts_partial.p, st));
            tl_mark(ctx, "k_sum_partials", st);
            TA_HIP_TRY(ctx, launch_sum_partials((const double*)ctx->ts_partial.p, (int)n_tiles, T, d_lagsum, st));
            return TA_OK;
        }
        (void)hipGetLastError();
    }
    // ... and the windowed VACF with its by-particle array (the class default) on the FP64 matrix cores: the k-slots
    // are filled from the time axis (bandbp_kernels.hpp)
    if (d_bp && !f32 && !src_f32 && mode == MODE_VACF && time_packed(513) && T < ((int64_t)1 << 24)) {
        const int64_t Tp = pm_pitch(T), n_tiles = (A + 63) / 64;
        if (ensure(ctx, ctx->bp_scratch, sizeof(double) * (size_t)A * Tp) == TA_OK &&
            ensure(ctx, ctx->ts_partial, sizeof(double) * (size_t)n_tiles * T) == TA_OK &&
            ensure(ctx, ctx->unit_counter, 64) == TA_OK) {
            tl_mark(ctx, "k_band_bp_vacf", st);
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
            TA_HIP_TRY(ctx, launch_band_bp_vacf(ctx->n_cu, (const double*)d_vel, pitch, (int)T, A, D, (double*)ctx->bp_scratch.p, Tp,
                                                (unsigned long long*)ctx->unit_counter.p, st));
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
            tl_mark(ctx, "k_bp_transpose", st);
            TA_HIP_TRY(ctx, launch_bp_transpose((const double*)ctx->bp_scratch.p, Tp, A, T, d_bp, ld_bp, (double*)ctx->ts_partial.p, st));
            tl_mark(ctx, "k_sum_partials", st);
            TA_HIP_TRY(ctx, launch_sum_partials((const double*)ctx->ts_partial.p, (int)n_tiles, T, d_lagsum, st));
            return TA_OK;
        }
        (void)hipGetLastError();
    }
    // ... and the Einstein-Helfand by-particle array (float64) the same way, on the product slab
    if (d_bp && !f32 && !src_f32 && mode == MODE_HELFAND && time_packed(352) && T < ((int64_t)1 << 24)) {
        const int64_t n_cols = A * D, n_pairs = (n_cols + 1) / 2, Tp = pm_pitch(T), n_tiles = (A + 63) / 64;
        const int n_parts = (int)std::min<int64_t>(1024, n_pairs);
        if (ensure(ctx, ctx->helf_p, pm_bytes(T, n_cols)) == TA_OK &&
            ensure(ctx, ctx->helf_small, sizeof(double) * (size_t)n_parts * T) == TA_OK &&
            ensure(ctx, ctx->bp_scratch, sizeof(double) * (size_t)A * Tp) == TA_OK &&
            ensure(ctx, ctx->ts_partial, sizeof(double) * (size_t)n_tiles * T) == TA_OK &&
            ensure(ctx, ctx->unit_counter, 64) == TA_OK) {
            tl_mark(ctx, "k_helfand_product", st);
            TA_HIP_TRY(ctx, launch_helfand_product((const double*)d_vel, (const double*)d_pos, d_masses, pitch, T, n_cols, D,
                                                   (double*)ctx->helf_p.p, (double*)ctx->helf_small.p, n_parts, st));
            tl_mark(ctx, "k_band_bp_helf", st);
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
            TA_HIP_TRY(ctx, launch_band_bp_helf(ctx->n_cu, (const double*)ctx->helf_p.p, pitch, (int)T, A, D, scale / (double)D,
                                                (double*)ctx->bp_scratch.p, Tp, (unsigned long long*)ctx->unit_counter.p, st));
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
            tl_mark(ctx, "k_bp_transpose", st);
            TA_HIP_TRY(ctx, launch_bp_transpose((const double*)ctx->bp_scratch.p, Tp, A, T, d_bp, ld_bp, (double*)ctx->ts_partial.p, st));
            tl_mark(ctx, "k_sum_partials", st);
            TA_HIP_TRY(ctx, launch_sum_partials((const double*)ctx->ts_partial.p, (int)n_tiles, T, d_lagsum, st));
            return TA_OK;
        }
        (void)hipGetLastError();
    }
    // windowed VACF lag sums alone: the by-particle kernel with a unit's particles summed in its accumulators (work handed out
    // by a counter: 51.4 ms at 5000 x 50000 x 3)
    if (band_ok && mode == MODE_VACF && time_packed(513) &&
        ensure(ctx, ctx->bp_scratch, sizeof(double) * band_bp_helf_partial_doubles(ctx->n_cu, (int)T, A)) == TA_OK &&
        ensure(ctx, ctx->unit_counter, 64) == TA_OK) {
        tl_mark(ctx, "k_band_bp_vacf", st);
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
        TA_HIP_TRY(ctx, launch_band_bp_vacf_lags(ctx->n_cu, (const double*)d_vel, pitch, (int)T, A, D, (double*)ctx->bp_scratch.p,
                                                 (unsigned long long*)ctx->unit_counter.p, d_lagsum, st));
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
        return TA_OK;
    }
    if (band_ok && mode == MODE_HELFAND && time_packed(352)) {
        // the product slab P = (m v) x first (T*A*D*8 bytes more; without them: the vector kernel); then the kernel of the
        // by-particle form with the particles of a unit summed in its accumulators (k-slots from the time axis: all four do
        // arithmetic: 463 ms per configs[4] share).
        const int64_t n_cols = A * D, n_pairs = (n_cols + 1) / 2;
        const int n_parts = (int)std::min<int64_t>(1024, n_pairs);
        if (ensure(ctx, ctx->helf_p, pm_bytes(T, n_cols)) == TA_OK &&
            ensure(ctx, ctx->helf_small, sizeof(double) * (size_t)n_parts * T) == TA_OK &&
            ensure(ctx, ctx->bp_scratch, sizeof(double) * band_bp_helf_partial_doubles(ctx->n_cu, (int)T, A)) == TA_OK &&
            ensure(ctx, ctx->unit_counter, 64) == TA_OK) {
            tl_mark(ctx, "k_helfand_product", st);
            TA_HIP_TRY(ctx, launch_helfand_product((const double*)d_vel, (const double*)d_pos, d_masses, pitch, T, n_cols, D,
                                                   (double*)ctx->helf_p.p, (double*)ctx->helf_small.p, n_parts, st));
            tl_mark(ctx, "k_band_bp_helf", st);
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
            TA_HIP_TRY(ctx, launch_band_bp_helf_lags(ctx->n_cu, (const double*)ctx->helf_p.p, pitch, (int)T, A, D, scale / (double)D,
                                                     (double*)ctx->bp_scratch.p, (unsigned long long*)ctx->unit_counter.p, d_lagsum, st));
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
            return TA_OK;
        }
        (void)hipGetLastError();  // out of memory for the product slab: not an error of the call
    }
    // Shape of the launch.  A thread owns one chunk pair (2L lags); a column group = W waves;
    // a workgroup = G groups working on G atoms at once, so that ONE workgroup fills a CU's
    // 16 wave slots (G*W <= 16) and its waves are dealt evenly to the 4 SIMDs.  The column
    // must be resident next to the compute units: in LDS when it fits (float64: <= 16376
    // frames, float32: <= 27296), otherwise in an L2-resident per-group staging buffer
    // (slower, any length).  L (8 or 10 lags per chunk) is the one that wastes fewer lanes
    // and SIMD slots for this n_frames.
    const size_t lds_cap = 160 * 1024;
    // Trajectories under ~1000 frames have fewer chunk pairs than a wave has lanes: a column group is then 16 or 32 LANES
    // ("direct_subwave" 1, the default), several groups per wave, up to 64 atoms per workgroup.
    struct Shape { int L, GT, G; size_t col; bool gs; double eff; } best{0, 0, 0, 0, false, -1.0};
    for (int L : {8, 10}) {
        if (ctx->opt_direct_chunk > 0 && L != ctx->opt_direct_chunk) continue;
        if (!direct_chunk_supported(L)) continue;
        Shape c;
        c.L = L;
        c.col = direct_lds_bytes((int)T, f32, L);
        c.gs = c.col > lds_cap;
        const int npairs = ((int)((T + L - 1) / L) + 1) / 2;
        c.GT = 64 * std::min(16, (npairs + 63) / 64);  // threads per column group
        if (ctx->opt_direct_subwave && npairs <= 32) c.GT = npairs <= 8 ? 8 : npairs <= 16 ? 16 : 32;
        c.G = 1024 / c.GT;
        if (!c.gs) c.G = (int)std::min<size_t>(c.G, lds_cap / c.col);
        if (ctx->opt_direct_groups > 0) c.G = (int)std::min<int64_t>(c.G, ctx->opt_direct_groups);
        c.G = (int)std::max<int64_t>(1, std::min<int64_t>(c.G, A));
        if (c.GT < 64) c.G = std::max(64 / c.GT, c.G / (64 / c.GT) * (64 / c.GT));  // whole waves
        const int rounds = (npairs + c.GT - 1) / c.GT;
        const int waves = (c.G * c.GT + 63) / 64;
        c.eff = (double)npairs / ((double)rounds * c.GT) *       // active lanes
                (double)waves / (4.0 * ((waves + 3) / 4)) *      // SIMD balance
                (1.0 - 0.6 / L);                                 // per-tile overhead
        if (c.GT < 64)  // sub-wave groups: the lanes in use decide; at equal use 8 lags per chunk are 5 - 15 % ahead
            c.eff = 2.0 + (double)npairs / ((double)rounds * c.GT) * (L == 8 ? 1.0 : 0.93);
        if (c.eff > best.eff) best = c;
    }
    if (best.eff < 0) return fail(ctx, TA_E_INVALID, "direct_chunk option: unsupported chunk size");
    const int L = best.L, G = best.G;
    const size_t col = best.col;
    const bool global_stage = best.gs;
    const int gnt = best.GT, nt = G * gnt;
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
    tl_mark(ctx, "memset", st);
    TA_HIP_TRY(ctx, hipMemsetAsync(ctx->ts_partial.p, 0, sizeof(double) * rows * T, st));
    tl_mark(ctx, "k_direct", st);
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
    TA_HIP_TRY(ctx, launch_direct(mode, f32, src_f32, L, d_vel, d_pos, d_masses, pitch, (int)T, A, D, scale, bp_am,
                                  Tp, (double*)ctx->ts_partial.p, (int)nwg, nt, lds, stage_buf,
                                  gnt, st));
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
    tl_mark(ctx, "k_sum_partials", st);
    TA_HIP_TRY(ctx, launch_sum_partials((const double*)ctx->ts_partial.p, (int)rows, T, d_lagsum, st));
    if (d_bp) {
        tl_mark(ctx, "k_bp_transpose", st);
        TA_HIP_TRY(ctx, launch_bp_transpose(bp_am, Tp, A, T, d_bp, ld_bp, nullptr, st));
    }
    return TA_OK;
}


// ---- compute on pair-major slabs (every entry point ends up here) --------------------------
// FFT VACF (wfft.hpp): n_frames <= 512 the wave-independent 512-point kernels; up to 10240 frames
// one on-chip transform per pass (R = 1), up to 163840 frames an outer radix R <= 16 in front of
// it; beyond that the direct correlator (same quantity: velocityautocorr.py:217-238 == :208-215
// mathematically).  Lag sums: forward kernel -> partial spectra per tuple of workgroups -> their
// sum -> ONE inverse transform.  By-particle array: per block of atoms, forward kernel -> the
// atoms' power spectra in scratch -> inverse kernel -> atom-major lags; then the transposition
// into the caller's (n_frames, ld_bp) array, which also adds up its 64 atoms per lag.
// pm_f32: the slab holds float32 elements (fft_reads_f32 says for which lengths the kernels take it)
bool fft_reads_f32(int64_t T) {
    int R0 = 0, R = 1;
    return wfft_choose((long)T, &R0, &R) && R == 1 && R0 > 1;
}

int fft_impl(ta_ctx* ctx, const double* pm, int64_t pitch, int64_t T, int64_t A, int D,
             double* d_lagsum, double* d_bp, int64_t ld_bp, hipStream_t st, bool pm_f32 = false) {
    int rc;
    const int64_t n_cols = A * D, n_pairs = (n_cols + 1) / 2;
    int R0 = 0, R = 1;
    if (!wfft_choose((long)T, &R0, &R))
        return direct_impl(ctx, MODE_VACF, pm, nullptr, nullptr, T, A, D, pitch, 1.0, d_lagsum, d_bp, ld_bp, st);
    if (!pm_f32 && short_applies(ctx, T) && (d_bp || T <= ctx->opt_short_lags_max))
        return short_impl(ctx, MODE_VACF, pm, nullptr, nullptr, T, A, D, pitch, 1.0, d_lagsum, d_bp, ld_bp, st);
    cd* tw = nullptr;
    if ((rc = get_wf_table(ctx, R0, R, &tw))) return rc;
    const int64_t Tp = pm_pitch(T), n_tiles = (A + 63) / 64;
    const int64_t L = 2L * R * R0 * 512;  // doubles per spectrum
    const int64_t cap = ctx->opt_fft_nwg > 0 ? ctx->opt_fft_nwg : (int64_t)ctx->n_cu * wfft_max_wg_per_cu(R0);
    // forward grid: whole tuples of 2R workgroups on each of the 8 XCDs, no more tuples than groups of units
    auto forward_grid = [&](int64_t n_groups) {
        const int64_t gran = 16 * R;
        return std::max<int64_t>(gran, std::min<int64_t>(cap, 2 * R * n_groups) / gran * gran);
    };
    if (d_bp) {
        if ((rc = ensure(ctx, ctx->bp_scratch, sizeof(double) * (size_t)A * Tp))) return rc;
        if ((rc = ensure(ctx, ctx->ts_partial, sizeof(double) * (size_t)n_tiles * T))) return rc;
    }
    if (R0 == 1) {
        if (!d_bp) {
            const int64_t nwg = std::max<int64_t>(1, std::min(cap, (n_pairs + 3) / 4));  // a wave per pair
            const int n_parts = (int)(4 * nwg);
            if ((rc = ensure(ctx, ctx->partial, sizeof(double) * (size_t)n_parts * L))) return rc;
            if ((rc = ensure(ctx, ctx->spec, sizeof(double) * (size_t)L))) return rc;
            tl_mark(ctx, "k_w1_accum", st);
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
            TA_HIP_TRY(ctx, launch_w1_accum((int)nwg, st, pm, pitch, (int)T, n_pairs, tw, (double*)ctx->partial.p));
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
            tl_mark(ctx, "k_wf_sum+k_wf_fold+k_wf_lags", st);
            TA_HIP_TRY(ctx, launch_wfft_finish(R0, (const double*)ctx->partial.p, n_parts, tw, (int)T,
                                               (double*)ctx->spec.p, d_lagsum, st));
            return TA_OK;
        }
        const int64_t nwg = std::max<int64_t>(1, std::min(cap, (A + 3) / 4));  // a wave per atom
        tl_mark(ctx, "k_w1_bp", st);
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
        TA_HIP_TRY(ctx, launch_w1_bp((int)nwg, st, pm, pitch, (int)T, A, D, tw, (double*)ctx->bp_scratch.p, Tp));
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
    } else if (!d_bp) {
        const int64_t nwg = forward_grid(n_pairs), n_tuples = nwg / (2 * R);
        if ((rc = ensure(ctx, ctx->partial, sizeof(double) * (size_t)n_tuples * L))) return rc;
        if ((rc = ensure(ctx, ctx->spec, sizeof(double) * (size_t)L))) return rc;
        tl_mark(ctx, "k_wsplit_accum", st);
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
        TA_HIP_TRY(ctx, launch_wfft_forward(R0, R, false, pm_f32, (int)nwg, st, pm, pitch, (int)T, n_pairs, D, tw,
                                            (double*)ctx->partial.p));
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
        // the summed spectrum -> lag sums: ONE inverse transform per launch
        tl_mark(ctx, "k_sum_partials", st);
        TA_HIP_TRY(ctx, launch_sum_partials((const double*)ctx->partial.p, (int)n_tuples, L, (double*)ctx->spec.p, st));
        tl_mark(ctx, "k_winverse", st);
        TA_HIP_TRY(ctx, launch_wfft_inverse(R0, R, 1, st, (const double*)ctx->spec.p, (int)T, 1, tw, d_lagsum, 0, 0));
        return TA_OK;
    } else {
        // blocks of atoms sized by the spectrum scratch (2.5 GiB unless the bp_spec_atoms option
        // says otherwise); a block starts on an even atom, so on a column-pair boundary
        const size_t spec_bytes = sizeof(double) * (size_t)L;
        int64_t CA = ctx->opt_bp_spec_atoms > 0 ? ctx->opt_bp_spec_atoms : (int64_t)(((size_t)5 << 29) / spec_bytes);
        CA = std::min<int64_t>(A, std::max<int64_t>(2, (CA + 1) / 2 * 2));
        {  // equal blocks instead of full ones and a remainder
            const int64_t n_blocks = (A + CA - 1) / CA;
            CA = std::min<int64_t>(CA, ((A + n_blocks - 1) / n_blocks + 1) / 2 * 2);
        }
        if ((rc = ensure(ctx, ctx->bp_spec, spec_bytes * (size_t)CA))) return rc;
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
        for (int64_t a0 = 0; a0 < A; a0 += CA) {
            const int64_t ca = std::min(CA, A - a0);
            const int64_t groups = D & 1 ? (ca + 1) / 2 : ca;  // a tuple of workgroups per group of atoms
            tl_mark(ctx, "k_wsplit_accum", st);
            // (a block starts on a column-pair boundary; pairs of a float32 slab are 8-byte rows)
            const double* pm_blk = pm_f32 ? (const double*)((const float*)pm + (a0 * D / 2) * pitch * 2)
                                          : pm + (a0 * D / 2) * pitch * 2;
            TA_HIP_TRY(ctx, launch_wfft_forward(R0, R, true, pm_f32, (int)forward_grid(groups), st, pm_blk, pitch,
                                                (int)T, ca, D, tw, (double*)ctx->bp_spec.p));
            tl_mark(ctx, "k_winverse", st);
            TA_HIP_TRY(ctx, launch_wfft_inverse(R0, R, (int)std::min<int64_t>(cap, ca), st,
                                                (const double*)ctx->bp_spec.p, (int)T, ca, tw,
                                                (double*)ctx->bp_scratch.p + a0 * Tp, Tp, (int)ctx->opt_bp_prefetch));
        }
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
    }
    tl_mark(ctx, "k_bp_transpose", st);
    TA_HIP_TRY(ctx, launch_bp_transpose((const double*)ctx->bp_scratch.p, Tp, A, T, d_bp, ld_bp,
                                        (double*)ctx->ts_partial.p, st));
    tl_mark(ctx, "k_sum_partials", st);
    TA_HIP_TRY(ctx, launch_sum_partials((const double*)ctx->ts_partial.p, (int)n_tiles, T, d_lagsum, st));
    return TA_OK;
}

// Helfand mean squared differences (viscosity.py:201-233); the "helfand_fft" option evaluates
// them as S1 - 2 S2 (helfand_fft.hip) where an FFT path exists for the request.
int helfand_impl(ta_ctx* ctx, const double* pm_vel, const double* pm_pos, const double* d_masses,
                 int64_t pitch, int64_t T, int64_t A, int D, double scale, double* d_lagsum,
                 double* d_bp, int64_t ld_bp, hipStream_t st) {
    int rc;
    const int64_t n_cols = A * D, n_pairs = (n_cols + 1) / 2;
    const bool fft_ok = ctx->opt_helfand_fft && T >= 2;
    int r0 = 0, ro = 0;
    const bool has_fft = wfft_choose((long)T, &r0, &ro);
    if (fft_ok && !d_bp && has_fft) {
        const int n_parts = (int)std::min<int64_t>(1024, n_pairs);
        if ((rc = ensure(ctx, ctx->helf_p, pm_bytes(T, n_cols)))) return rc;
        if ((rc = ensure(ctx, ctx->helf_small, sizeof(double) * ((size_t)n_parts * T + 3 * (size_t)T + 1)))) return rc;
        double* P = (double*)ctx->helf_p.p;
        double* Qpart = (double*)ctx->helf_small.p;
        double* Q = Qpart + (size_t)n_parts * T;
        double* S2 = Q + T;
        double* C = S2 + T;
        tl_mark(ctx, "k_helfand_product", st);
        TA_HIP_TRY(ctx, launch_helfand_product(pm_vel, pm_pos, d_masses, pitch, T, n_cols, D, P, Qpart, n_parts, st));
        tl_mark(ctx, "k_sum_partials", st);
        TA_HIP_TRY(ctx, launch_sum_partials(Qpart, n_parts, T, Q, st));
        if ((rc = fft_impl(ctx, P, pitch, T, A, D, S2, nullptr, 0, st))) return rc;
        tl_mark(ctx, "k_helfand_combine", st);
        TA_HIP_TRY(ctx, launch_helfand_combine(Q, S2, C, (int)T, scale / (double)D, d_lagsum, st));
        return TA_OK;
    }
    if (fft_ok && d_bp && has_fft) {
        if ((rc = ensure(ctx, ctx->helf_p, pm_bytes(T, n_cols)))) return rc;
        if ((rc = ensure(ctx, ctx->helf_small, sizeof(double) * ((size_t)T + 1) * A))) return rc;
        double* P = (double*)ctx->helf_p.p;
        double* Ca = (double*)ctx->helf_small.p;
        if (n_cols & 1)  // the unpaired last column's partner is never written by the product kernel
            TA_HIP_TRY(ctx, hipMemsetAsync(P + (size_t)(n_pairs - 1) * pitch * 2, 0, (size_t)pitch * 16, st));
        tl_mark(ctx, "k_helfand_product", st);
        TA_HIP_TRY(ctx, launch_helfand_product_bp(pm_vel, pm_pos, d_masses, pitch, T, A, D, P, Ca, st));
        if ((rc = fft_impl(ctx, P, pitch, T, A, D, d_lagsum, d_bp, ld_bp, st))) return rc;
        tl_mark(ctx, "k_helfand_combine", st);
        TA_HIP_TRY(ctx, launch_helfand_combine_bp(Ca, A, (int)T, scale / (double)D, d_bp, ld_bp, st));
        tl_mark(ctx, "k_row_sums", st);
        TA_HIP_TRY(ctx, launch_row_sums(d_bp, T, A, ld_bp, d_lagsum, st));
        return TA_OK;
    }
    return direct_impl(ctx, MODE_HELFAND, pm_vel, pm_pos, d_masses, T, A, D, pitch, scale, d_lagsum, d_bp,
                       ld_bp, st);
}

enum { W_FFT = 0, W_DIRECT = 1, W_HELFAND = 2 };

// one compute call on pair-major slabs, bracketed by the timing events
int compute_pm(ta_ctx* ctx, int which, const void* pm_vel_any, const void* pm_pos_any, const double* d_masses,
               int64_t pitch, int64_t T, int64_t A, int D, double scale, double* d_lagsum, double* d_bp,
               int64_t ld_bp, hipStream_t st, bool record_start, bool pm_f32 = false) {
    int rc;
    ctx->timing_valid = false;
    if (record_start) {
        ctx->ev = ctx->ring[ctx->n_calls % ta_ctx::kRing];
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[0], st));
        tl_reset(ctx);
    }
    // float32 device slabs are read as they are by the float32 direct correlators; every other
    // evaluation works on a float64 copy (same layout) in the context's scratch slabs
    const bool direct_on_f32 = pm_f32 && ctx->opt_direct_f32 &&
                               (which == W_DIRECT || (which == W_HELFAND && !(ctx->opt_helfand_fft && T >= 2)));
    // ... and by the FFT kernels of the plans without an outer radix (513 ... 10240 frames)
    const bool fft_on_f32 = pm_f32 && which == W_FFT && fft_reads_f32(T);
    if (pm_f32 && !direct_on_f32 && !fft_on_f32) {
        const size_t n_el = (size_t)((A * D + 1) / 2) * (size_t)pitch * 2;
        const void* src[2] = {pm_vel_any, pm_pos_any};
        for (int k = 0; k < 2; ++k) {
            if (!src[k]) continue;
            if ((rc = ensure(ctx, ctx->pm_in[k], n_el * sizeof(double)))) return rc;
            tl_mark(ctx, "k_widen_f32", st);
            TA_HIP_TRY(ctx, launch_widen_f32((const float*)src[k], (double*)ctx->pm_in[k].p, (long)n_el, st));
        }
        pm_vel_any = ctx->pm_in[0].p;
        if (pm_pos_any) pm_pos_any = ctx->pm_in[1].p;
    }
    const double* pm_vel = (const double*)pm_vel_any;
    const double* pm_pos = (const double*)pm_pos_any;
    // paths without a dominant kernel of their own re-record ev[1]/ev[2] inside
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
    if (which == W_FFT) rc = fft_impl(ctx, pm_vel, pitch, T, A, D, d_lagsum, d_bp, ld_bp, st, fft_on_f32);
    else if (which == W_DIRECT)
        rc = direct_impl(ctx, MODE_VACF, pm_vel_any, nullptr, nullptr, T, A, D, pitch, 1.0, d_lagsum, d_bp, ld_bp, st,
                         direct_on_f32);
    else if (direct_on_f32)
        rc = direct_impl(ctx, MODE_HELFAND, pm_vel_any, pm_pos_any, d_masses, T, A, D, pitch, scale, d_lagsum, d_bp,
                         ld_bp, st, true);
    else rc = helfand_impl(ctx, pm_vel, pm_pos, d_masses, pitch, T, A, D, scale, d_lagsum, d_bp, ld_bp, st);
    if (rc) return rc;
    tl_mark(ctx, "end", st);
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
    tl_mark(ctx, "k_relayout", st);
    TA_HIP_TRY(ctx, launch_relayout(d_src, false, ld_row, n_cols, T, ctx->pm_in[k].p, false, pm_pitch(T), 0, st));
    *out = (const double*)ctx->pm_in[k].p;
    return TA_OK;
}

int dev_entry(ta_ctx* ctx, int which, const double* d_vel, const double* d_pos, const double* d_masses,
              int64_t T, int64_t A, int D, int64_t ld_row, double scale, double* d_lagsum, double* d_bp,
              int64_t ld_bp, void* stream) {
    TA_NO_CPU(ctx);
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
    tl_reset(ctx);
    const double *pv = nullptr, *px = nullptr;
    if ((rc = relayout_input(ctx, 0, d_vel, T, A * D, ld_row, st, &pv))) return rc;
    if (which == W_HELFAND && (rc = relayout_input(ctx, 1, d_pos, T, A * D, ld_row, st, &px))) return rc;
    return compute_pm(ctx, which, pv, px, d_masses, pm_pitch(T), T, A, D, scale, d_lagsum, d_bp, ld_bp, st, false);
}

// frames committed by ta_stage_commit travel on the context's own stream: a caller's stream that
// is about to touch the slabs waits for them (a no-op when nothing is pending)
int order_after_staging(ta_ctx* ctx, hipStream_t st) {
    if (int rc = commit_flush(ctx)) return rc;  // queued commits have made their calls on ctx->stream
    if (st == ctx->stream) return TA_OK;
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
    TA_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_stage, 0));
    return TA_OK;
}

int staged_entry(ta_ctx* ctx, int which, const double* d_masses, double scale, double* d_lagsum,
                 double* d_bp, int64_t ld_bp, void* stream) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    TA_NO_CPU(ctx);
    const int need = which == W_HELFAND ? 2 : 1;
    if (ctx->st_nslabs < need) return fail(ctx, TA_E_STATE, "slabs have not been staged");
    if (!d_lagsum || (which == W_HELFAND && !d_masses)) return fail(ctx, TA_E_INVALID, "null device pointer");
    if (d_bp && ld_bp < ctx->st_A) return fail(ctx, TA_E_INVALID, "ld_bp smaller than n_atoms");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = order_after_staging(ctx, (hipStream_t)stream);
    if (rc) return rc;
    return compute_pm(ctx, which, ctx->d_slabs[0], need == 2 ? ctx->d_slabs[1] : nullptr, d_masses,
                      ctx->st_pitch, ctx->st_T, ctx->st_A, ctx->st_D, scale, d_lagsum, d_bp, ld_bp,
                      (hipStream_t)stream, true, ctx->st_dev_f32);
}

}  // namespace

extern "C" {

int ta_abi_version(void) { return 6; }

int ta_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ta_last_error(const ta_ctx* ctx) {
    if (!ctx) return g_tls_error.c_str();
    // a copy made under the lock, owned by the calling thread until its next call
    thread_local std::string copy;
    try {
        std::lock_guard<std::mutex> lk(g_err_m);
        copy = ctx->err;
    } catch (...) {
        return "out of memory while reading the error message";
    }
    return copy.c_str();
}

int ta_ctx_create(int device, ta_ctx** out) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(nullptr, c_, m_); }, [&]() -> int {
    if (!out) return fail(nullptr, TA_E_INVALID, "out is NULL");
    *out = nullptr;
    if (device == TA_DEVICE_CPU) {  // the opt-in CPU backend: no HIP call at all
        if (!ta::cpu::supported()) return fail(nullptr, TA_E_UNSUPPORTED, "the CPU backend is built for hosts with AVX2 and FMA");
        ta_ctx* c = new (std::nothrow) ta_ctx();
        if (!c) return fail(nullptr, TA_E_NOMEM, "out of host memory");
        c->is_cpu = true;
        c->device = TA_DEVICE_CPU;
        c->n_cu = 0;
        c->cpu.threads = ta::cpu::hardware_threads();
        *out = c;
        return TA_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1)
        return fail(nullptr, TA_E_HIP,
                    "no usable HIP device (this library has no CPU fallback): " +
                        std::string(e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
    if (device < 0 || device >= n) return fail(nullptr, TA_E_INVALID, "device index out of range (TA_DEVICE_CPU = -1 asks for the CPU backend)");
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
    });
}

int ta_stage_free(ta_ctx* ctx) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (ctx->is_cpu) {
        for (auto& b : ctx->h_blocks)
            if (b.base) host_block_unmap(b);
        ctx->h_blocks.clear();
        ctx->h_slabs.clear();
        ctx->cpu.slabs.clear();
        ctx->st_nslabs = 0;
        ctx->st_T = ctx->st_A = ctx->st_pitch = 0;
        return TA_OK;
    }
    (void)commit_flush(ctx);  // (an error of a commit into slabs that are going away is dropped with them)
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->relayout_stream) hipStreamSynchronize(ctx->relayout_stream);
    for (size_t i = 0; i < ctx->h_slabs.size(); ++i) {
        if (i < ctx->h_blocks.size() && ctx->h_blocks[i].base) host_block_unmap(ctx->h_blocks[i]);
        else if (ctx->h_slabs[i]) hipHostFree(ctx->h_slabs[i]);
    }
    ctx->h_blocks.clear();
    for (double* d : ctx->d_slabs)
        if (d) hipFree(d);
    ctx->h_slabs.clear();
    ctx->d_slabs.clear();
    ctx->st_nslabs = 0;
    ctx->st_T = ctx->st_A = ctx->st_pitch = 0;
    return TA_OK;
    });
}

int ta_ctx_destroy(ta_ctx* ctx) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx) return TA_OK;
    if (ctx->is_cpu) {
        ta_stage_free(ctx);
        delete ctx;
        return TA_OK;
    }
    (void)commit_flush(ctx);
    commit_stop(ctx);
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    ta_stage_free(ctx);
    for (auto& kv : ctx->wf_tables) hipFree(kv.second);
    for (DevBuf* b : {&ctx->partial, &ctx->spec, &ctx->ts_partial, &ctx->out_lagsum, &ctx->out_bp,
                      &ctx->masses, &ctx->bounce, &ctx->stage_buf, &ctx->helf_p,
                      &ctx->helf_small, &ctx->pm_in[0], &ctx->pm_in[1], &ctx->bp_scratch, &ctx->bp_spec,
                      &ctx->bounce2, &ctx->unit_counter})
        if (b->p) hipFree(b->p);
    for (auto& q : ctx->ring)
        for (auto& ev : q)
            if (ev) hipEventDestroy(ev);
    if (ctx->ev_stage) hipEventDestroy(ctx->ev_stage);
    for (hipEvent_t e : ctx->mark_pool) hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) {
        if (ctx->ev_piece[i]) hipEventDestroy(ctx->ev_piece[i]);
        if (ctx->ev_done[i]) hipEventDestroy(ctx->ev_done[i]);
    }
    if (ctx->relayout_stream) hipStreamDestroy(ctx->relayout_stream);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return TA_OK;
    });
}

int ta_trim(ta_ctx* ctx) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (ctx->is_cpu) return TA_OK;  // (no workspaces outlive a call)
    if (int rc = commit_flush(ctx)) return rc;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    for (DevBuf* b : {&ctx->partial, &ctx->spec, &ctx->ts_partial, &ctx->out_bp, &ctx->bounce, &ctx->bounce2, &ctx->stage_buf,
                      &ctx->helf_p, &ctx->helf_small, &ctx->pm_in[0], &ctx->pm_in[1],
                      &ctx->bp_scratch, &ctx->bp_spec})
        if (b->p) {
            hipFree(b->p);
            b->p = nullptr;
            b->bytes = 0;
        }
    return TA_OK;
    });
}

int ta_set_option(ta_ctx* ctx, const char* key, int64_t value) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx || !key) return fail(ctx, TA_E_INVALID, "null argument");
    if (!strcmp(key, "fft_nwg")) ctx->opt_fft_nwg = value;
    else if (!strcmp(key, "direct_nwg")) ctx->opt_direct_nwg = value;
    else if (!strcmp(key, "direct_f32")) ctx->opt_direct_f32 = value;
    else if (!strcmp(key, "direct_groups")) ctx->opt_direct_groups = value;
    else if (!strcmp(key, "direct_chunk")) ctx->opt_direct_chunk = value;
    else if (!strcmp(key, "direct_mfma")) {
        if (value != 0 && value != 1 && value != 3)
            return fail(ctx, TA_E_INVALID, "direct_mfma: 0 vector kernels, 1 by trajectory length (default), 3 matrix cores always "
                                           "(2, the column-packed forms, left the library in round 6: tools/band/)");
        ctx->opt_direct_mfma = value;
    }
    else if (!strcmp(key, "helfand_fft")) ctx->opt_helfand_fft = value;
    else if (!strcmp(key, "bp_block")) ctx->opt_bp_block = value;
    else if (!strcmp(key, "bp_spec_atoms")) ctx->opt_bp_spec_atoms = value;
    else if (!strcmp(key, "lock_ahead")) ctx->opt_lock_ahead = value;
    else if (!strcmp(key, "cpu_threads")) {  // CPU backend: OpenMP team size (0: the runtime's default)
        if (value < 0 || value > 4096) return fail(ctx, TA_E_INVALID, "cpu_threads: 0 (default) .. 4096");
        ctx->cpu.threads = value > 0 ? (int)value : ta::cpu::hardware_threads();
    }
    else if (!strcmp(key, "fail_alloc_after")) ctx->opt_fail_alloc_after = value;
    else if (!strcmp(key, "fail_throw_after")) ctx->opt_fail_throw_after = value;
    else if (!strcmp(key, "bp_prefetch")) ctx->opt_bp_prefetch = value;
    else if (!strcmp(key, "short_max")) ctx->opt_short_max = value;
    else if (!strcmp(key, "direct_subwave")) ctx->opt_direct_subwave = value;
    else if (!strcmp(key, "mid_max")) ctx->opt_mid_max = value;
    else if (!strcmp(key, "mid_all")) ctx->opt_mid_all = value;
    else if (!strcmp(key, "mid_ncl")) ctx->opt_mid_ncl = value;
    else if (!strcmp(key, "short_lags_max")) ctx->opt_short_lags_max = value;
    else if (!strcmp(key, "stage_device_f32")) ctx->opt_stage_device_f32 = value;
    else if (!strcmp(key, "timeline")) ctx->opt_timeline = value;
    else if (!strcmp(key, "async_commit")) {
        if (int rc = commit_flush(ctx)) return rc;
        ctx->opt_async_commit = value;
    }
    else return fail(ctx, TA_E_INVALID, std::string("unknown option ") + key);
    return TA_OK;
    });
}

int ta_fft_plan_info(int64_t n_frames, int64_t* m_out, int* n_threads, int* n_stages) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(nullptr, c_, m_); }, [&]() -> int {
    int R0 = 0, R = 1;
    if (!wfft_choose((long)n_frames, &R0, &R))
        return fail(nullptr, TA_E_UNSUPPORTED, "n_frames exceeds the largest FFT plan");
    // [outer radix R while the rows are read,] first-stage radix R0 (none up to 512 frames), then
    // one wave per 512-point sub-series (8 x 8 x 8)
    if (m_out) *m_out = (int64_t)R * R0 * 512;
    if (n_threads) *n_threads = wfft_threads(R0);
    if (n_stages) *n_stages = (R0 == 1 ? 3 : 4) + (R > 1 ? 1 : 0);
    return TA_OK;
    });
}

/* --------------------------------------------- pinned host memory for results */
}  // extern "C"

namespace {
// result arrays handed out by ta_host_alloc*: mapping -> page-locked in one piece (a 2-D device->host copy spans all of
// it); ta_host_free finds them here by address
std::mutex g_host_m;
std::map<void*, size_t> g_host_blocks;  // base -> mapped length
}  // namespace

extern "C" {

int ta_host_alloc_on(int device, int64_t n_bytes, void** h_out) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(nullptr, c_, m_); }, [&]() -> int {
    if (!h_out || n_bytes < 0) return fail(nullptr, TA_E_INVALID, "bad argument");
    *h_out = nullptr;
    // the allocating thread may be a fresh helper thread whose current device is 0: bind it to the
    // analysis' own GPU first, so that no context is created on a device the rank does not use
    if (device >= 0) {
        const hipError_t es = hipSetDevice(device);
        if (es != hipSuccess)
            return fail(nullptr, TA_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(es));
    }
    // an anonymous mapping page-locked by ONE hipHostRegister (portable: usable by every device's copy engines; the
    // result outlives the context): 4 GiB in ~0.2 s where hipHostMalloc takes 0.5-0.9 s with the runtime's lock held
    const size_t len = ((size_t)std::max<int64_t>(n_bytes, 16) + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
    void* m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m != MAP_FAILED) {
        (void)madvise(m, len, MADV_HUGEPAGE);
        if (hipHostRegister(m, len, hipHostRegisterPortable) == hipSuccess) {
            std::lock_guard<std::mutex> lk(g_host_m);
            g_host_blocks[m] = len;
            *h_out = m;
            return TA_OK;
        }
        (void)hipGetLastError();
        munmap(m, len);
    }
    void* h = nullptr;
    const hipError_t e = hipHostMalloc(&h, (size_t)std::max<int64_t>(n_bytes, 16), hipHostMallocPortable);
    if (e != hipSuccess)
        return fail(nullptr, TA_E_NOMEM, std::string("pinned host allocation failed: ") + hipGetErrorString(e));
    *h_out = h;
    return TA_OK;
    });
}

int ta_host_alloc(int64_t n_bytes, void** h_out) { return ta_host_alloc_on(-1, n_bytes, h_out); }

int ta_host_free(void* h) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(nullptr, c_, m_); }, [&]() -> int {
    if (!h) return TA_OK;
    size_t len = 0;
    {
        std::lock_guard<std::mutex> lk(g_host_m);
        auto it = g_host_blocks.find(h);
        if (it != g_host_blocks.end()) len = it->second, g_host_blocks.erase(it);
    }
    if (len) {
        const hipError_t e = hipHostUnregister(h);
        munmap(h, len);
        return e == hipSuccess ? TA_OK : fail(nullptr, TA_E_HIP, std::string("hipHostUnregister: ") + hipGetErrorString(e));
    }
    const hipError_t e = hipHostFree(h);
    return e == hipSuccess ? TA_OK : fail(nullptr, TA_E_HIP, std::string("hipHostFree: ") + hipGetErrorString(e));
    });
}

/* ------------------------------------------------------------------ staging */
static int stage_alloc_common(ta_ctx* ctx, int64_t n_frames, int64_t n_atoms, int dim, int dtype,
                              int n_slabs, void** h_slabs) {
    int rc = check_shape(ctx, n_frames, n_atoms, dim, n_atoms * dim);
    if (rc) return rc;
    if (n_slabs < 1 || n_slabs > 4) return fail(ctx, TA_E_INVALID, "bad slab count");
    if (dtype != TA_F32 && dtype != TA_F64) return fail(ctx, TA_E_INVALID, "bad dtype");
    if (ctx->is_cpu) {
        // host slabs in the reference's (n_frames, n_atoms, dim) layout, zero-filled mappings like the GPU contexts' (never
        // page-locked); the CPU backend reads them where they are
        if (!h_slabs) return fail(ctx, TA_E_UNSUPPORTED, "the CPU backend has no device slabs");
        ta_stage_free(ctx);
        const size_t bytes = (size_t)n_frames * n_atoms * dim * (dtype == TA_F32 ? 4 : 8);
        for (int i = 0; i < n_slabs; ++i) {
            HostBlock blk;
            if (host_block_map(bytes, &blk) != 0) {
                ta_stage_free(ctx);
                return fail(ctx, TA_E_NOMEM, "staging allocation failed: no host memory for the slab");
            }
            ctx->h_slabs.push_back(blk.base);
            ctx->h_blocks.push_back(blk);
            ctx->cpu.slabs.push_back(blk.base);
            h_slabs[i] = blk.base;
        }
        ctx->st_T = ctx->cpu.T = n_frames;
        ctx->st_A = ctx->cpu.A = n_atoms;
        ctx->st_D = ctx->cpu.D = dim;
        ctx->st_dtype = ctx->cpu.dtype = dtype;
        ctx->st_nslabs = n_slabs;
        ctx->st_pitch = pm_pitch(n_frames);
        return TA_OK;
    }
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ta_stage_free(ctx);
    const size_t n = (size_t)n_frames * n_atoms * dim;
    const size_t esz = dtype == TA_F32 ? 4 : 8;
    // device slabs hold float32 when the option asks for it and nothing wider is coming in
    const bool dev_f32 = ctx->opt_stage_device_f32 && (dtype == TA_F32 || !h_slabs);
    const size_t dbytes = pm_bytes(n_frames, n_atoms * dim, dev_f32);
    for (int i = 0; i < n_slabs; ++i) {
        void* h = nullptr;
        double* d = nullptr;
        hipError_t e = hipSuccess;
        HostBlock blk;
        if (h_slabs) {
            // zero-filled like the reference's np.zeros (velocityautocorr.py:150); page-locked as it is committed
            if (host_block_map(n * esz, &blk) == 0) h = blk.base;
            else {  // no mapping to be had: the runtime's allocator
                e = hipHostMalloc(&h, n * esz, hipHostMallocDefault);
                if (e == hipSuccess) host_zero(h, n * esz);
            }
        }
        if (e == hipSuccess) e = hipMalloc((void**)&d, dbytes);
        if (e != hipSuccess) {
            if (blk.base) host_block_unmap(blk);
            else if (h) hipHostFree(h);
            ta_stage_free(ctx);
            return fail(ctx, TA_E_NOMEM, std::string("staging allocation failed: ") + hipGetErrorString(e));
        }
        ctx->h_slabs.push_back(h);
        ctx->h_blocks.push_back(blk);
        ctx->d_slabs.push_back(d);
        // frames never committed read as zeros, like the reference's np.zeros slab
        TA_HIP_TRY(ctx, hipMemsetAsync(d, 0, dbytes, ctx->stream));
        if (h_slabs) h_slabs[i] = h;
    }
    ctx->st_T = n_frames;
    ctx->st_A = n_atoms;
    ctx->st_D = dim;
    ctx->st_dtype = dtype;
    ctx->st_dev_f32 = dev_f32;
    ctx->st_nslabs = n_slabs;
    ctx->st_pitch = pm_pitch(n_frames);
    // the zero fill ran on the context's stream; later fills may come on any stream
    TA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (h_slabs && ctx->opt_async_commit && ctx->opt_lock_ahead) {
        // the worker starts page-locking the slabs' first chunks while the caller sets up its frame loop (an empty job)
        if (!ctx->cq_thread.joinable()) ctx->cq_thread = std::thread(commit_worker, ctx);
        {
            std::lock_guard<std::mutex> lk(ctx->cq_m);
            ctx->cq.emplace_back(0, 0);
        }
        ctx->cq_cv.notify_all();
    }
    return TA_OK;
}

int ta_stage_alloc(ta_ctx* ctx, int64_t n_frames, int64_t n_atoms, int dim, int dtype, int n_slabs,
                   void** h_slabs) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!h_slabs) return fail(ctx, TA_E_INVALID, "h_slabs is NULL");
    return stage_alloc_common(ctx, n_frames, n_atoms, dim, dtype, n_slabs, h_slabs);
    });
}

int ta_stage_alloc_device(ta_ctx* ctx, int64_t n_frames, int64_t n_atoms, int dim, int n_slabs) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    return stage_alloc_common(ctx, n_frames, n_atoms, dim, TA_F64, n_slabs, nullptr);
    });
}

}  // extern "C"

static int stage_commit_now(ta_ctx* ctx, int64_t frame_lo, int64_t frame_hi) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (ctx->st_nslabs == 0) return fail(ctx, TA_E_STATE, "ta_stage_alloc has not been called");
    if (frame_lo < 0 || frame_hi > ctx->st_T || frame_lo > frame_hi)
        return fail(ctx, TA_E_INVALID, "frame range out of bounds");
    if (!ctx->h_slabs[0]) return fail(ctx, TA_E_STATE, "device-only slabs: use ta_stage_commit_dev");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t row = (size_t)ctx->st_A * ctx->st_D;
    const size_t esz = ctx->st_dtype == TA_F32 ? 4 : 8;
    if (frame_hi == frame_lo) return TA_OK;
    // frames cross PCIe in their native width into one of two landing buffers (<= 64 MiB each) and
    // are transposed into the pair-major slab on the device (float32 widened on the way), on a
    // second stream: piece i + 1 crosses PCIe while piece i is transposed
    const int64_t per = std::max<int64_t>(1, (int64_t)(((size_t)64 << 20) / (row * esz)));
    const int64_t chunk = std::min<int64_t>(per, frame_hi - frame_lo);
    int rc = ensure(ctx, ctx->bounce, (size_t)chunk * row * esz);
    if (!rc) rc = ensure(ctx, ctx->bounce2, (size_t)chunk * row * esz);
    if (rc) return rc;
    if (!ctx->relayout_stream) {
        TA_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->relayout_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            TA_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_piece[i], hipEventDisableTiming));
            TA_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_done[i], hipEventDisableTiming));
        }
    }
    void* land[2] = {ctx->bounce.p, ctx->bounce2.p};
    bool used[2] = {false, false};
    // the slabs may still be read by work queued earlier on the context's stream
    TA_HIP_TRY(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
    TA_HIP_TRY(ctx, hipStreamWaitEvent(ctx->relayout_stream, ctx->ev_stage, 0));
    int piece = 0;
    for (int i = 0; i < ctx->st_nslabs; ++i) {
        for (int64_t f = frame_lo; f < frame_hi; f += chunk, ++piece) {
            const int b = piece & 1;
            const int64_t m = std::min(chunk, frame_hi - f);
            const char* src = (const char*)ctx->h_slabs[i] + (size_t)f * row * esz;
            if (used[b]) TA_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_done[b], 0));  // buffer free again
            HostBlock& blk = ctx->h_blocks[i];
            if (blk.base) {
                // page-lock the chunks these frames live in (first commit of each: 64 MiB at 23 GB/s), then copy in
                // segments that stay inside one chunk
                const size_t s0 = (size_t)f * row * esz, s1 = s0 + (size_t)m * row * esz;
                TA_HIP_TRY(ctx, host_block_lock(blk, s0, s1));
                for (size_t p = s0; p < s1;) {
                    const size_t e = std::min(s1, (p / HostBlock::kChunk + 1) * HostBlock::kChunk);
                    TA_HIP_TRY(ctx, hipMemcpyAsync((char*)land[b] + (p - s0), blk.base + p, e - p, hipMemcpyHostToDevice, ctx->stream));
                    p = e;
                }
            } else
                TA_HIP_TRY(ctx, hipMemcpyAsync(land[b], src, (size_t)m * row * esz, hipMemcpyHostToDevice, ctx->stream));
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev_piece[b], ctx->stream));
            TA_HIP_TRY(ctx, hipStreamWaitEvent(ctx->relayout_stream, ctx->ev_piece[b], 0));
            TA_HIP_TRY(ctx, launch_relayout(land[b], ctx->st_dtype == TA_F32, (long)row, (long)row, m,
                                            ctx->d_slabs[i], ctx->st_dev_f32, ctx->st_pitch, f, ctx->relayout_stream));
            TA_HIP_TRY(ctx, hipEventRecord(ctx->ev_done[b], ctx->relayout_stream));
            used[b] = true;
        }
    }
    // whatever follows on the context's stream sees the transposed frames
    for (int b = 0; b < 2; ++b)
        if (used[b]) TA_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_done[b], 0));
    return TA_OK;
}

// Page-locking ahead of the frame loop (round 6): behind a commit the worker registers the next chunks of every slab --
// pages no frame has touched yet, which the registration faults in -- so that the filling threads find their pages present
// and the commits that follow find their chunks registered (before: every chunk was registered by the commit that first
// needed it, with the loop's threads page-faulting on it meanwhile: frame fills 0.12 s -> 0.42 s at 10000 x 50000 x 3).
static void stage_lock_ahead(ta_ctx* ctx, int64_t frame_from) {
    constexpr size_t kAhead = 3;
    const size_t row = (size_t)ctx->st_A * ctx->st_D, esz = ctx->st_dtype == TA_F32 ? 4 : 8;
    const size_t s0 = (size_t)frame_from * row * esz;
    for (size_t i = 0; i < ctx->h_blocks.size(); ++i) {
        HostBlock& blk = ctx->h_blocks[i];
        if (blk.base && s0 < blk.bytes) (void)host_block_lock(blk, s0, std::min(blk.bytes, s0 + kAhead * HostBlock::kChunk));
    }
}

static void commit_worker(ta_ctx* ctx) {
    (void)hipSetDevice(ctx->device);
    std::unique_lock<std::mutex> lk(ctx->cq_m);
    for (;;) {
        ctx->cq_cv.wait(lk, [&] { return ctx->cq_stop || !ctx->cq.empty(); });
        if (ctx->cq.empty()) return;  // stop requested and nothing left
        const auto job = ctx->cq.front();
        ctx->cq.pop_front();
        ctx->cq_busy = true;
        lk.unlock();
        // (an exception on this thread would end the process: it becomes the queued commit's error)
        const int rc = ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
            const int r = job.second > job.first ? stage_commit_now(ctx, job.first, job.second) : TA_OK;  // (an empty job: lock ahead only)
            if (r == TA_OK && ctx->opt_lock_ahead) stage_lock_ahead(ctx, job.second);
            return r;
        });
        lk.lock();
        if (rc && ctx->cq_rc == TA_OK) {  // the first failure is the one reported
            std::lock_guard<std::mutex> el(g_err_m);
            ctx->cq_rc = rc, ctx->cq_err = ctx->err;
        }
        ctx->cq_busy = false;
        ctx->cq_cv.notify_all();
    }
}

// every queued commit has made its HIP calls (their work is queued on the context's streams); returns the
// first error one of them hit
int commit_flush(ta_ctx* ctx) {
    if (!ctx->cq_thread.joinable()) return TA_OK;
    std::unique_lock<std::mutex> lk(ctx->cq_m);
    ctx->cq_cv.wait(lk, [&] { return ctx->cq.empty() && !ctx->cq_busy; });
    const int rc = ctx->cq_rc;
    if (rc) {
        const std::string msg = ctx->cq_err;
        ctx->cq_rc = TA_OK;
        lk.unlock();
        return fail(ctx, rc, "queued ta_stage_commit: " + msg);
    }
    return TA_OK;
}

static void commit_stop(ta_ctx* ctx) {
    if (!ctx->cq_thread.joinable()) return;
    {
        std::lock_guard<std::mutex> lk(ctx->cq_m);
        ctx->cq_stop = true;
    }
    ctx->cq_cv.notify_all();
    ctx->cq_thread.join();
    ctx->cq_stop = false;
}

extern "C" {

int ta_stage_commit(ta_ctx* ctx, int64_t frame_lo, int64_t frame_hi) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (ctx->st_nslabs == 0) return fail(ctx, TA_E_STATE, "ta_stage_alloc has not been called");
    if (frame_lo < 0 || frame_hi > ctx->st_T || frame_lo > frame_hi)
        return fail(ctx, TA_E_INVALID, "frame range out of bounds");
    if (!ctx->h_slabs[0]) return fail(ctx, TA_E_STATE, "device-only slabs: use ta_stage_commit_dev");
    if (frame_hi == frame_lo) return TA_OK;
    if (ctx->is_cpu) return TA_OK;  // the CPU backend reads the host slab in place
    if (!ctx->opt_async_commit) {
        if (int rc = commit_flush(ctx)) return rc;
        return stage_commit_now(ctx, frame_lo, frame_hi);
    }
    if (!ctx->cq_thread.joinable()) ctx->cq_thread = std::thread(commit_worker, ctx);
    {
        std::lock_guard<std::mutex> lk(ctx->cq_m);
        ctx->cq.emplace_back(frame_lo, frame_hi);
    }
    ctx->cq_cv.notify_all();
    return TA_OK;
    });
}

int ta_stage_commit_dev(ta_ctx* ctx, int slab, const void* d_src, int dtype, int64_t ld_row,
                        int64_t frame_lo, int64_t frame_hi, void* stream) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx || !d_src) return fail(ctx, TA_E_INVALID, "null argument");
    TA_NO_CPU(ctx);
    if (slab < 0 || slab >= ctx->st_nslabs) return fail(ctx, TA_E_INVALID, "no such slab");
    if (dtype != TA_F32 && dtype != TA_F64) return fail(ctx, TA_E_INVALID, "bad dtype");
    if (frame_lo < 0 || frame_hi > ctx->st_T || frame_lo > frame_hi)
        return fail(ctx, TA_E_INVALID, "frame range out of bounds");
    if (ld_row < ctx->st_A * ctx->st_D) return fail(ctx, TA_E_INVALID, "ld_row smaller than n_atoms*dim");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc_ = order_after_staging(ctx, (hipStream_t)stream)) return rc_;
    TA_HIP_TRY(ctx, launch_relayout(d_src, dtype == TA_F32, ld_row, ctx->st_A * ctx->st_D, frame_hi - frame_lo,
                                    ctx->d_slabs[slab], ctx->st_dev_f32, ctx->st_pitch, frame_lo,
                                    (hipStream_t)stream));
    return TA_OK;
    });
}

int ta_stage_synth(ta_ctx* ctx, int slab, uint64_t seed, int64_t col_offset, int64_t n_cols_total,
                   void* stream) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (slab < 0 || slab >= ctx->st_nslabs) return fail(ctx, TA_E_INVALID, "no such slab");
    if (col_offset < 0 || col_offset + ctx->st_A * ctx->st_D > n_cols_total)
        return fail(ctx, TA_E_INVALID, "column block outside the synthetic tensor");
    if (ctx->is_cpu) {  // the same generator into the host slab
        ta::cpu::synth(ctx->cpu, slab, seed, col_offset, n_cols_total);
        return TA_OK;
    }
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc_ = order_after_staging(ctx, (hipStream_t)stream)) return rc_;
    TA_HIP_TRY(ctx, launch_synth(ctx->d_slabs[slab], ctx->st_dev_f32, ctx->st_pitch, ctx->st_A * ctx->st_D, ctx->st_T, seed,
                                 col_offset, n_cols_total, (hipStream_t)stream));
    return TA_OK;
    });
}

int ta_stage_read_dev(ta_ctx* ctx, int slab, double* d_dst, int64_t ld_row, void* stream) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx || !d_dst) return fail(ctx, TA_E_INVALID, "null argument");
    TA_NO_CPU(ctx);
    if (slab < 0 || slab >= ctx->st_nslabs) return fail(ctx, TA_E_INVALID, "no such slab");
    if (ld_row < ctx->st_A * ctx->st_D) return fail(ctx, TA_E_INVALID, "ld_row smaller than n_atoms*dim");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc_ = order_after_staging(ctx, (hipStream_t)stream)) return rc_;
    TA_HIP_TRY(ctx, launch_unlayout(ctx->d_slabs[slab], ctx->st_dev_f32, ctx->st_pitch, ctx->st_A * ctx->st_D, ctx->st_T, d_dst,
                                    ld_row, (hipStream_t)stream));
    return TA_OK;
    });
}

int ta_stage_device(ta_ctx* ctx, int slab, double** d_slab, int64_t* pitch_rows, int64_t* n_pairs) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx || !d_slab) return fail(ctx, TA_E_INVALID, "null argument");
    TA_NO_CPU(ctx);
    if (slab < 0 || slab >= ctx->st_nslabs) return fail(ctx, TA_E_INVALID, "no such slab");
    *d_slab = ctx->d_slabs[slab];
    if (pitch_rows) *pitch_rows = ctx->st_pitch;
    if (n_pairs) *n_pairs = (ctx->st_A * ctx->st_D + 1) / 2;
    return TA_OK;
    });
}

/* --------------------------------------------------------- device compute */
int ta_vacf_fft_dev(ta_ctx* ctx, const double* d_vel, int64_t T, int64_t A, int D, int64_t ld_row,
                    double* d_lagsum, double* d_bp, int64_t ld_bp, void* stream) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    return dev_entry(ctx, W_FFT, d_vel, nullptr, nullptr, T, A, D, ld_row, 1.0, d_lagsum, d_bp, ld_bp, stream);
    });
}

int ta_vacf_direct_dev(ta_ctx* ctx, const double* d_vel, int64_t T, int64_t A, int D, int64_t ld_row,
                       double* d_lagsum, double* d_bp, int64_t ld_bp, void* stream) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    return dev_entry(ctx, W_DIRECT, d_vel, nullptr, nullptr, T, A, D, ld_row, 1.0, d_lagsum, d_bp, ld_bp, stream);
    });
}

int ta_helfand_msd_dev(ta_ctx* ctx, const double* d_vel, const double* d_pos, const double* d_masses,
                       int64_t T, int64_t A, int D, int64_t ld_row, double scale, double* d_lagsum,
                       double* d_bp, int64_t ld_bp, void* stream) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    return dev_entry(ctx, W_HELFAND, d_vel, d_pos, d_masses, T, A, D, ld_row, scale, d_lagsum, d_bp, ld_bp, stream);
    });
}

int ta_vacf_fft_staged(ta_ctx* ctx, double* d_lagsum, double* d_bp, int64_t ld_bp, void* stream) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    return staged_entry(ctx, W_FFT, nullptr, 1.0, d_lagsum, d_bp, ld_bp, stream);
    });
}

int ta_vacf_direct_staged(ta_ctx* ctx, double* d_lagsum, double* d_bp, int64_t ld_bp, void* stream) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    return staged_entry(ctx, W_DIRECT, nullptr, 1.0, d_lagsum, d_bp, ld_bp, stream);
    });
}

int ta_helfand_msd_staged(ta_ctx* ctx, const double* d_masses, double scale, double* d_lagsum,
                          double* d_bp, int64_t ld_bp, void* stream) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    return staged_entry(ctx, W_HELFAND, d_masses, scale, d_lagsum, d_bp, ld_bp, stream);
    });
}

int ta_last_timing(ta_ctx* ctx, float* total_ms, float* main_kernel_ms) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    TA_NO_CPU(ctx);
    if (!ctx->timing_valid) return fail(ctx, TA_E_STATE, "no completed compute call to time");
    TA_HIP_TRY(ctx, hipEventSynchronize(ctx->ev[3]));
    float t = 0.f, m = 0.f;
    TA_HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[3]));
    TA_HIP_TRY(ctx, hipEventElapsedTime(&m, ctx->ev[1], ctx->ev[2]));
    if (total_ms) *total_ms = t;
    if (main_kernel_ms) *main_kernel_ms = m;
    return TA_OK;
    });
}

int ta_timing_history(ta_ctx* ctx, int max_n, float* total_ms, float* main_kernel_ms, int* n_out) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx || !n_out) return fail(ctx, TA_E_INVALID, "null argument");
    TA_NO_CPU(ctx);
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
    });
}

int ta_clock_probe(ta_ctx* ctx, int n_launches, double* mhz, double* cycles_per_unit_pass, double* ms_per_launch) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    TA_NO_CPU(ctx);
    if (ctx->st_nslabs < 1) return fail(ctx, TA_E_STATE, "slabs have not been staged");
    if (n_launches < 1) return fail(ctx, TA_E_INVALID, "need at least one launch");
    if (ctx->st_dev_f32) return fail(ctx, TA_E_UNSUPPORTED, "clock probe: float64 device slabs only");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t T = ctx->st_T, n_pairs = (ctx->st_A * ctx->st_D + 1) / 2;
    int R0 = 0, R = 1;
    if (!wfft_choose((long)T, &R0, &R) || R != 1 || !(R0 == 8 || R0 == 10 || R0 == 12 || R0 == 16 || R0 == 20))
        return fail(ctx, TA_E_UNSUPPORTED, "clock probe: plans R0 = 8, 10, 12, 16, 20 without an outer radix only");
    cd* tw = nullptr;
    int rc = get_wf_table(ctx, R0, R, &tw);
    if (rc) return rc;
    const int64_t cap = ctx->opt_fft_nwg > 0 ? ctx->opt_fft_nwg : (int64_t)ctx->n_cu * wfft_max_wg_per_cu(R0);
    const int64_t nwg = std::max<int64_t>(16, std::min<int64_t>(cap, 2 * n_pairs) / 16 * 16), n_tuples = nwg / 2;
    const int64_t L = 2L * R0 * 512;
    if ((rc = ensure(ctx, ctx->partial, sizeof(double) * (size_t)n_tuples * L))) return rc;
    DevBuf st;
    if ((rc = ensure(ctx, st, sizeof(unsigned long long) * 16 * (size_t)nwg))) return rc;
    if (int rc_ = order_after_staging(ctx, ctx->stream)) {
        hipFree(st.p);
        return rc_;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;  // its own events: the probe is not a compute call and leaves the timing ring alone
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    for (int i = 0; i < n_launches && e == hipSuccess; ++i)
        e = launch_wfft_forward_stamp(R0, (int)nwg, ctx->stream, ctx->d_slabs[0], ctx->st_pitch, (int)T, n_pairs, tw,
                                      (double*)ctx->partial.p, (unsigned long long*)st.p);
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    std::vector<unsigned long long> h(16 * (size_t)nwg);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), st.p, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    hipFree(st.p);
    if (e != hipSuccess) return fail(ctx, TA_E_HIP, std::string("clock probe: ") + hipGetErrorString(e));
    // the last launch's stamps, wave 0 of every workgroup: [0] S1, [1] S2 cycles, [2] the kernel's span
    // in shader cycles, [3] in 100 MHz ticks
    double cyc = 0.0, span = 0.0, ticks = 0.0;
    for (int64_t w = 0; w < nwg; ++w) {
        cyc += (double)h[16 * w + 0] + (double)h[16 * w + 1];
        span += (double)h[16 * w + 2];
        ticks += (double)h[16 * w + 3];
    }
    const double unit_passes = 2.0 * (double)n_pairs;  // every pair in both passes, over all workgroups
    if (mhz) *mhz = ticks > 0 ? span / ticks * 100.0 : 0.0;
    if (cycles_per_unit_pass) *cycles_per_unit_pass = cyc / unit_passes;
    if (ms_per_launch) *ms_per_launch = ms / n_launches;
    return TA_OK;
    });
}

int ta_kernel_timeline(ta_ctx* ctx, int max_n, const char** names, float* ms, int* n_out) {
    return ta::guard([&](int c_, const std::string& m_) { return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx || !n_out) return fail(ctx, TA_E_INVALID, "null argument");
    TA_NO_CPU(ctx);
    *n_out = 0;
    if (ctx->marks.size() < 2) return TA_OK;  // option off, or no call yet
    TA_HIP_TRY(ctx, hipEventSynchronize(ctx->marks.back().ev));
    std::vector<std::pair<const char*, float>> agg;  // by name, in order of first appearance
    for (size_t i = 0; i + 1 < ctx->marks.size(); ++i) {
        float d = 0.f;
        TA_HIP_TRY(ctx, hipEventElapsedTime(&d, ctx->marks[i].ev, ctx->marks[i + 1].ev));
        size_t j = 0;
        while (j < agg.size() && strcmp(agg[j].first, ctx->marks[i].name)) ++j;
        if (j == agg.size()) agg.push_back({ctx->marks[i].name, 0.f});
        agg[j].second += d;
    }
    const int n = (int)std::min<size_t>(agg.size(), (size_t)std::max(0, max_n));
    for (int i = 0; i < n; ++i) {
        if (names) names[i] = agg[i].first;
        if (ms) ms[i] = agg[i].second;
    }
    *n_out = n;
    return TA_OK;
    });
}

/* ------------------------------------------------- host-facing (blocking) */
}  // extern "C"

namespace ta {
// One context's share of a host-facing call, queued but not waited for: compute on the staged
// slabs, by-particle blocks copied into the caller's host array (row stride ld_host elements: the
// caller's array may be wider than this context's block of atoms -- the column range of one GPU
// in a multi-device group), and the lag-indexed SUM over this context's atoms left on the device
// in *d_total ((n_frames,) float64, valid once host_wait has returned or for work queued behind it
// on ctx->stream).  h_masses: this context's atoms.
int host_launch(ta_ctx* ctx, int which, const double* h_masses, double scale, double* h_bp, int64_t ld_host,
                double** d_total) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    TA_NO_CPU(ctx);
    const int need = which == W_HELFAND ? 2 : 1;
    if (ctx->st_nslabs < need) return fail(ctx, TA_E_STATE, "slabs have not been staged");
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t T = ctx->st_T, A = ctx->st_A;
    if (h_bp && ld_host < A) return fail(ctx, TA_E_INVALID, "host row stride smaller than n_atoms");
    // With a by-particle array the device->host copy (8 GB at 10000 x 100000) is several times
    // the compute: atoms go in blocks, the copy of block c (a strided 2-D copy into the caller's
    // (n_frames, ld_host) array, on a second stream) runs under the compute of block c + 1.
    const int64_t CH = ctx->opt_bp_block > 0 ? (ctx->opt_bp_block + 63) / 64 * 64 : 16384;
    const bool blocked = h_bp && A >= 2 * CH;
    const int64_t n_blocks = blocked ? (A + CH - 1) / CH : 1;
    // rows [0, n_blocks): per-block lag sums; row n_blocks: their sum (one block: row 0 is the sum)
    int rc = ensure(ctx, ctx->out_lagsum, sizeof(double) * T * (n_blocks + 1));
    if (rc) return rc;
    double* d_ls = (double*)ctx->out_lagsum.p;
    double* d_bp = nullptr;
    if (h_bp) {
        if ((rc = ensure(ctx, ctx->out_bp, sizeof(double) * (size_t)T * A))) return rc;
        d_bp = (double*)ctx->out_bp.p;
    }
    const double* d_m = nullptr;
    if (which == W_HELFAND) {
        if (!h_masses) return fail(ctx, TA_E_INVALID, "h_masses is NULL");
        if ((rc = ensure(ctx, ctx->masses, sizeof(double) * A))) return rc;
        TA_HIP_TRY(ctx, hipMemcpyAsync(ctx->masses.p, h_masses, sizeof(double) * A, hipMemcpyHostToDevice,
                                       ctx->stream));
        d_m = (const double*)ctx->masses.p;
    }
    if (!blocked) {
        if ((rc = staged_entry(ctx, which, d_m, scale, d_ls, d_bp, A, (void*)ctx->stream))) return rc;
        if (h_bp)
            TA_HIP_TRY(ctx, hipMemcpy2DAsync(h_bp, sizeof(double) * ld_host, d_bp, sizeof(double) * A,
                                             sizeof(double) * A, T, hipMemcpyDeviceToHost, ctx->stream));
        *d_total = d_ls;
        return TA_OK;
    }
    if (!ctx->copy_stream) TA_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    // the blocks go to compute_pm directly (staged_entry is not on this path): join the commit worker here, or the
    // first block is queued ahead of frames whose copies the worker has not issued yet
    if ((rc = order_after_staging(ctx, ctx->stream))) return rc;
    const int D = ctx->st_D;
    for (int64_t b = 0; b < n_blocks; ++b) {
        const int64_t lo = b * CH, hi = std::min(A, lo + CH);
        const int64_t pair_lo = lo * D / 2;  // lo is a multiple of 64: a pair boundary
        const size_t off = (size_t)pair_lo * ctx->st_pitch * (ctx->st_dev_f32 ? 8 : 16);  // bytes
        const void* v = (const char*)ctx->d_slabs[0] + off;
        const void* x = need == 2 ? (const char*)ctx->d_slabs[1] + off : nullptr;
        if ((rc = compute_pm(ctx, which, v, x, d_m ? d_m + lo : nullptr, ctx->st_pitch, T, hi - lo, D, scale,
                             d_ls + b * T, d_bp + lo, A, ctx->stream, true, ctx->st_dev_f32)))
            return rc;
        TA_HIP_TRY(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
        TA_HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_stage, 0));
        TA_HIP_TRY(ctx, hipMemcpy2DAsync(h_bp + lo, sizeof(double) * ld_host, d_bp + lo, sizeof(double) * A,
                                         sizeof(double) * (hi - lo), T, hipMemcpyDeviceToHost,
                                         ctx->copy_stream));
    }
    // the blocks' lag sums, added on the device in a fixed order
    TA_HIP_TRY(ctx, launch_sum_partials(d_ls, (int)n_blocks, T, d_ls + n_blocks * T, ctx->stream));
    *d_total = d_ls + n_blocks * T;
    return TA_OK;
}

int host_wait(ta_ctx* ctx) {
    TA_HIP_TRY(ctx, hipSetDevice(ctx->device));
    TA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream) TA_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    return TA_OK;
}
hipStream_t ctx_stream(ta_ctx* ctx) { return ctx->stream; }
int ctx_device(const ta_ctx* ctx) { return ctx->device; }
int64_t ctx_staged_frames(const ta_ctx* ctx) { return ctx->st_nslabs ? ctx->st_T : 0; }
int ctx_fail(ta_ctx* ctx, int code, const std::string& msg) { return fail(ctx, code, msg); }
int ctx_host_slab(ta_ctx* ctx, int slab, void** h, int64_t* T, int64_t* A, int* D, int* dtype) {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (slab < 0 || slab >= ctx->st_nslabs) return fail(ctx, TA_E_INVALID, "no such slab");
    if (!ctx->h_slabs[slab]) return fail(ctx, TA_E_STATE, "device-only slabs have no host side to fill");
    *h = ctx->h_slabs[slab], *T = ctx->st_T, *A = ctx->st_A, *D = ctx->st_D, *dtype = ctx->st_dtype;
    return TA_OK;
}
}  // namespace ta

extern "C" {

static int host_compute(ta_ctx* ctx, int which, const double* h_masses, double scale,
                        double* h_ts, double* h_bp) {
    // (an exception after the launch: the queued kernels and copies into the caller's arrays finish before the error returns)
    return ta::guard([&](int c_, const std::string& m_) { if (ctx) (void)host_wait(ctx); return fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx) return fail(nullptr, TA_E_INVALID, "null context");
    if (!h_ts) return fail(ctx, TA_E_INVALID, "h_timeseries is NULL");
    if (ctx->is_cpu) {
        const int need = which == W_HELFAND ? 2 : 1;
        if (ctx->st_nslabs < need) return fail(ctx, TA_E_STATE, "slabs have not been staged");
        if (which == W_HELFAND && !h_masses) return fail(ctx, TA_E_INVALID, "h_masses is NULL");
        const int rc = which == W_FFT      ? ta::cpu::vacf_fft(ctx->cpu, h_ts, h_bp)
                       : which == W_DIRECT ? ta::cpu::vacf_direct(ctx->cpu, h_ts, h_bp)
                                           : ta::cpu::helfand(ctx->cpu, h_masses, scale, h_ts, h_bp);
        if (rc) return fail(ctx, rc, "CPU backend: out of host memory");
        const double n_at = (double)ctx->st_A;  // mean over atoms (velocityautocorr.py:214,237; viscosity.py:233)
        for (int64_t k = 0; k < ctx->st_T; ++k) h_ts[k] /= n_at;
        return TA_OK;
    }
    double* d_total = nullptr;
    int rc = host_launch(ctx, which, h_masses, scale, h_bp, ctx->st_A, &d_total);
    if (rc) return rc;
    const int64_t T = ctx->st_T;
    TA_HIP_TRY(ctx, hipMemcpyAsync(h_ts, d_total, sizeof(double) * T, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = host_wait(ctx))) return rc;
    const double n_at = (double)ctx->st_A;  // mean over atoms (velocityautocorr.py:214,237)
    for (int64_t k = 0; k < T; ++k) h_ts[k] /= n_at;
    return TA_OK;
    });
}

int ta_vacf_fft(ta_ctx* ctx, double* h_ts, double* h_bp) { return host_compute(ctx, W_FFT, nullptr, 1.0, h_ts, h_bp); }
int ta_vacf_direct(ta_ctx* ctx, double* h_ts, double* h_bp) { return host_compute(ctx, W_DIRECT, nullptr, 1.0, h_ts, h_bp); }
int ta_helfand_msd(ta_ctx* ctx, const double* h_masses, double scale, double* h_ts, double* h_bp) {
    return host_compute(ctx, W_HELFAND, h_masses, scale, h_ts, h_bp);
}

}  // extern "C"
