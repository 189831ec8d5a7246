// stage_host.hip — the per-frame slab fill on the host, natively (include/ta_hip.h: ta_stage_frame,
// ta_group_stage_frame).  Replaces the reference's per-frame statements
//     self._velocities[self._frame_index] = self.atomgroup.velocities[:, self._dim]
// (/root/reference/transport_analysis/velocityautocorr.py:192-194; viscosity.py:189-199 for velocities and
// positions), where `atomgroup.velocities` is itself a gather `ts.velocities[atomgroup.ix]` into a temporary:
// one pass from the Timestep's own (n_atoms_universe, 3) array — gather by atom index, column selection,
// conversion to the slab's element type — straight into the pinned staging slab, on a few host threads.
// ctypes releases the GIL for the duration of the call.
#include <hip/hip_runtime.h>
#include <emmintrin.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/ta_hip.h"
#include "ta_internal.hpp"

namespace {

struct Job {
    const void* src = nullptr;
    void* dst = nullptr;          // frame row of the slab: (n_atoms, n_col) elements
    const int64_t* index = nullptr;  // source row of atom a, or NULL: atom_lo + a
    int64_t atom_lo = 0, n_atoms = 0, ld_row = 0;
    int col0 = 0, col_step = 1, n_col = 0;
    bool src_f32 = true, dst_f32 = true;
};

// dst[0 .. n) = (D) src[0 .. n), both contiguous.  Same types: 16-byte non-temporal stores (the slab is written
// once and read by the DMA engine, never by this core: no read-for-ownership of its lines, no cache pollution;
// glibc's memcpy only does that above tens of MB)
template <typename S, typename D>
void copy_block(const S* src, D* dst, int64_t n) {
    if constexpr (std::is_same_v<S, D>) {
        const char* s = (const char*)src;
        char* d = (char*)dst;
        size_t bytes = (size_t)n * sizeof(S);
        const size_t head = std::min<size_t>(bytes, (16 - ((uintptr_t)d & 15)) & 15);
        memcpy(d, s, head);
        s += head, d += head, bytes -= head;
        const size_t body = bytes & ~(size_t)63;
        for (size_t k = 0; k < body; k += 64) {
            const __m128i x0 = _mm_loadu_si128((const __m128i*)(s + k)), x1 = _mm_loadu_si128((const __m128i*)(s + k + 16));
            const __m128i x2 = _mm_loadu_si128((const __m128i*)(s + k + 32)), x3 = _mm_loadu_si128((const __m128i*)(s + k + 48));
            _mm_stream_si128((__m128i*)(d + k), x0);
            _mm_stream_si128((__m128i*)(d + k + 16), x1);
            _mm_stream_si128((__m128i*)(d + k + 32), x2);
            _mm_stream_si128((__m128i*)(d + k + 48), x3);
        }
        memcpy(d + body, s + body, bytes - body);
        _mm_sfence();  // the streamed lines are globally visible before this thread reports (the GPU's DMA reads them next)
    } else {
        for (int64_t k = 0; k < n; ++k) dst[k] = (D)src[k];
    }
}

template <typename S, typename D>
void copy_rows(const Job& j, int64_t a0, int64_t a1) {
    const S* src = (const S*)j.src;
    D* dst = (D*)j.dst + a0 * j.n_col;
    const bool dense = j.col_step == 1 && !j.index;
    if (dense && j.n_col == j.ld_row && j.col0 == 0) {  // whole rows of consecutive atoms: one block
        copy_block(src + (j.atom_lo + a0) * j.ld_row, dst, (a1 - a0) * j.n_col);
        return;
    }
    for (int64_t a = a0; a < a1; ++a) {
        const S* s = src + (j.index ? j.index[a] : j.atom_lo + a) * j.ld_row + j.col0;
        for (int k = 0; k < j.n_col; ++k) *dst++ = (D)s[(int64_t)k * j.col_step];
    }
}

void run_chunk(const Job& j, int64_t a0, int64_t a1) {
    if (a1 <= a0) return;
    if (j.src_f32) {
        if (j.dst_f32) copy_rows<float, float>(j, a0, a1);
        else copy_rows<float, double>(j, a0, a1);
    } else {
        if (j.dst_f32) copy_rows<double, float>(j, a0, a1);
        else copy_rows<double, double>(j, a0, a1);
    }
}

// A few helper threads shared by the process.  Frames arrive every ~50 us while a trajectory is staged: a
// helper spins on the generation counter for a short while after its last job before it goes to sleep on the
// condition variable (waking a sleeping thread costs about as much as a whole frame's copy).  The atoms of a
// call are split statically, one slice per participant (the caller is one): no shared counter to fight over.
class Pool {
public:
    static Pool& get() {
        static Pool p;
        return p;
    }
    int helpers() const { return (int)threads_.size(); }
    // jobs[0 .. n_jobs): returns when all of them are copied
    void run(const Job* jobs, int n_jobs) {
        int64_t total = 0;
        for (int i = 0; i < n_jobs; ++i) total += jobs[i].n_atoms;
        const int parts = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)threads_.size() + 1, total / 4096));
        if (parts == 1) {
            slice(jobs, n_jobs, total, 0, 1);
            return;
        }
        std::lock_guard<std::mutex> one_caller(call_);  // one staging call at a time uses the helpers
        jobs_ = jobs, n_jobs_ = n_jobs, total_ = total, parts_ = parts;
        done_.v.store(0, std::memory_order_relaxed);
        gen_.v.fetch_add(1, std::memory_order_release);
        if (sleepers_.v.load(std::memory_order_acquire) > 0) {
            { std::lock_guard<std::mutex> lk(m_); }
            cv_.notify_all();
        }
        slice(jobs, n_jobs, total, 0, parts);
        const int n = (int)threads_.size();
        while (done_.v.load(std::memory_order_acquire) < n) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }

private:
    struct alignas(64) Line {
        std::atomic<long> v{0};
    };
    // participant p of `parts`: atoms [total p / parts, total (p + 1) / parts) of the jobs laid end to end
    static void slice(const Job* jobs, int n_jobs, int64_t total, int p, int parts) {
        int64_t lo = total * p / parts, hi = total * (p + 1) / parts, base = 0;
        for (int i = 0; i < n_jobs && base < hi; ++i) {
            const int64_t a0 = std::max<int64_t>(0, lo - base), a1 = std::min<int64_t>(jobs[i].n_atoms, hi - base);
            run_chunk(jobs[i], a0, a1);
            base += jobs[i].n_atoms;
        }
    }
    Pool() {
        int n = 3;
        if (const char* e = getenv("TA_AMD_STAGE_THREADS")) n = atoi(e) - 1;
        const int hw = (int)std::thread::hardware_concurrency();
        n = std::max(0, std::min(n, std::max(0, hw / 2 - 1)));
        for (int i = 0; i < n; ++i) threads_.emplace_back([this, i] { loop(i + 1); });
    }
    ~Pool() {
        stop_.store(true);
        gen_.v.fetch_add(1, std::memory_order_release);
        { std::lock_guard<std::mutex> lk(m_); }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void loop(int me) {
        long seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            int spins = 0;
            while (gen_.v.load(std::memory_order_acquire) == seen) {
                if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
                    std::unique_lock<std::mutex> lk(m_);
                    sleepers_.v.fetch_add(1, std::memory_order_acq_rel);
                    cv_.wait(lk, [&] { return gen_.v.load(std::memory_order_acquire) != seen; });
                    sleepers_.v.fetch_sub(1, std::memory_order_acq_rel);
                    break;
                }
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            seen = gen_.v.load(std::memory_order_acquire);
            if (stop_.load()) return;
            if (me < parts_) slice(jobs_, n_jobs_, total_, me, parts_);
            done_.v.fetch_add(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_, call_;
    std::condition_variable cv_;
    Line gen_, done_, sleepers_;
    const Job* jobs_ = nullptr;
    int n_jobs_ = 0, parts_ = 1;
    int64_t total_ = 0;
    std::atomic<bool> stop_{false};
};

int check_args(ta_ctx* ctx, const void* h_src, int src_dtype, int64_t ld_row, int col0, int col_step, int n_col) {
    if (!h_src) return ta::ctx_fail(ctx, TA_E_INVALID, "h_src is NULL");
    if (src_dtype != TA_F32 && src_dtype != TA_F64) return ta::ctx_fail(ctx, TA_E_INVALID, "bad dtype");
    if (n_col < 1 || n_col > 3 || col0 < 0 || col_step < 1 || col0 + (int64_t)(n_col - 1) * col_step >= ld_row)
        return ta::ctx_fail(ctx, TA_E_INVALID, "columns col0 + k col_step (k < n_col) must lie inside a source row of ld_row elements");
    return TA_OK;
}

}  // namespace

namespace ta {
// api.hip: the pinned slab `slab` of the context as (host pointer, frames, atoms, dim, element type), or an error
int ctx_host_slab(ta_ctx* ctx, int slab, void** h, int64_t* T, int64_t* A, int* D, int* dtype);
}  // namespace ta

extern "C" {

int ta_stage_frame(ta_ctx* ctx, int slab, int64_t frame, const void* h_src, int src_dtype, int64_t ld_row, int col0,
                   int col_step, int n_col, int64_t atom_lo, const int64_t* h_index, int64_t n_atoms) {
    return ta::guard([&](int c_, const std::string& m_) { return ta::ctx_fail(ctx, c_, m_); }, [&]() -> int {
    if (!ctx) return ta::ctx_fail(nullptr, TA_E_INVALID, "null context");
    void* h = nullptr;
    int64_t T = 0, A = 0;
    int D = 0, dtype = 0;
    int rc = ta::ctx_host_slab(ctx, slab, &h, &T, &A, &D, &dtype);
    if (rc) return rc;
    if ((rc = check_args(ctx, h_src, src_dtype, ld_row, col0, col_step, n_col))) return rc;
    if (frame < 0 || frame >= T) return ta::ctx_fail(ctx, TA_E_INVALID, "frame out of range");
    if (n_col != D || n_atoms != A || atom_lo < 0) return ta::ctx_fail(ctx, TA_E_INVALID, "n_col / n_atoms do not match the staged slab");
    Job j;
    j.src = h_src, j.index = h_index, j.atom_lo = atom_lo, j.n_atoms = n_atoms, j.ld_row = ld_row;
    j.col0 = col0, j.col_step = col_step, j.n_col = n_col;
    j.src_f32 = src_dtype == TA_F32, j.dst_f32 = dtype == TA_F32;
    j.dst = (char*)h + (size_t)frame * A * D * (j.dst_f32 ? 4 : 8);
    Pool::get().run(&j, 1);
    return TA_OK;
    });
}

int ta_group_stage_frame(ta_group* g, int slab, int64_t frame, const void* h_src, int src_dtype, int64_t ld_row, int col0,
                         int col_step, int n_col, int64_t atom_lo, const int64_t* h_index, int64_t n_atoms) {
    return ta::guard([&](int c_, const std::string& m_) { return ta::ctx_fail(nullptr, c_, m_); }, [&]() -> int {
    const int n = ta_group_size(g);
    if (n < 1) return TA_E_INVALID;
    std::vector<Job> jobs;
    for (int i = 0; i < n; ++i) {
        ta_ctx* c = nullptr;
        int rc = ta_group_member(g, i, &c, nullptr);
        if (rc) return rc;
        int64_t lo = 0, hi = 0;
        if ((rc = ta_group_shard(g, n_atoms, i, &lo, &hi))) return rc;
        if (hi == lo) continue;
        void* h = nullptr;
        int64_t T = 0, A = 0;
        int D = 0, dtype = 0;
        if ((rc = ta::ctx_host_slab(c, slab, &h, &T, &A, &D, &dtype))) return rc;
        if ((rc = check_args(c, h_src, src_dtype, ld_row, col0, col_step, n_col))) return rc;
        if (frame < 0 || frame >= T || n_col != D || A != hi - lo || atom_lo < 0)
            return ta::ctx_fail(c, TA_E_INVALID, "frame / n_col / n_atoms do not match the group's staged slabs");
        Job j;
        j.src = h_src, j.index = h_index ? h_index + lo : nullptr, j.atom_lo = atom_lo + lo, j.n_atoms = hi - lo, j.ld_row = ld_row;
        j.col0 = col0, j.col_step = col_step, j.n_col = n_col;
        j.src_f32 = src_dtype == TA_F32, j.dst_f32 = dtype == TA_F32;
        j.dst = (char*)h + (size_t)frame * A * D * (j.dst_f32 ? 4 : 8);
        jobs.push_back(j);
    }
    if (!jobs.empty()) Pool::get().run(jobs.data(), (int)jobs.size());
    return TA_OK;
    });
}

int ta_stage_threads(void) {
    try {  // (the first call starts the helper threads)
        return Pool::get().helpers() + 1;
    } catch (...) {
        return 1;
    }
}

}  // extern "C"

namespace ta {
// ---- host blocks: where pinned slabs and result arrays live -------------------------------------------------
// hipHostMalloc of 6 GiB takes 0.7-1.4 s on the boxes of this pool (and hipHostFree another 0.5-1.0 s), all of it
// inside _prepare, before the first frame.  An anonymous mapping is there at once, zero-filled by the kernel (the
// reference's np.zeros: no memset), backed by transparent huge pages where the system allows, and
// hipHostRegister page-locks it at 23 GB/s — chunk by chunk, when a chunk is first committed (on the commit
// worker's thread, under the frame loop), or in one piece for a result array.  A hipMemcpy may not span two
// separately registered ranges ("invalid argument"), so copies out of a slab are cut at chunk boundaries
// (tools/ubench/hostreg_probe.hip, profiles/r05_hostreg_probe.txt).
int host_block_map(size_t bytes, HostBlock* b) {
    // whole 64 MiB chunks for a slab of several; a small slab (a few atoms by a few hundred frames) maps, zeroes and
    // page-locks its own size rounded up to 2 MiB only
    const size_t want = std::max<size_t>(bytes, 1);
    const size_t unit = want < HostBlock::kChunk ? ((size_t)2 << 20) : HostBlock::kChunk;
    const size_t len = (want + unit - 1) / unit * unit;
    void* m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return -1;
    (void)madvise(m, len, MADV_HUGEPAGE);
    b->base = (char*)m, b->bytes = len;
    b->locked.assign((len + HostBlock::kChunk - 1) / HostBlock::kChunk, 0);
    return 0;
}
// page-lock the chunks that cover [b0, b1).  A chunk the runtime refuses to register (RLIMIT_MEMLOCK, a cgroup limit)
// stays pageable -- marked 2, not tried again: copies out of it are ordinary pageable hipMemcpyAsync calls, slower,
// same bytes -- instead of failing the analysis after its frames have been read.
hipError_t host_block_lock(HostBlock& b, size_t b0, size_t b1) {
    for (size_t k = b0 / HostBlock::kChunk; k < b.locked.size() && k * HostBlock::kChunk < b1; ++k) {
        if (b.locked[k]) continue;
        const size_t off = k * HostBlock::kChunk, len = std::min(HostBlock::kChunk, b.bytes - off);
        const hipError_t e = hipHostRegister(b.base + off, len, hipHostRegisterPortable);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            static std::atomic<bool> said{false};
            if (!said.exchange(true))
                fprintf(stderr, "transport_analysis_amd: hipHostRegister of a staging chunk failed (%s): copying from pageable memory\n",
                        hipGetErrorString(e));
            b.locked[k] = 2;
            continue;
        }
        b.locked[k] = 1;
    }
    return hipSuccess;
}
void host_block_unmap(HostBlock& b) {
    if (!b.base) return;
    for (size_t k = 0; k < b.locked.size(); ++k)
        if (b.locked[k] == 1) {
            const hipError_t e = hipHostUnregister(b.base + k * HostBlock::kChunk);
            if (e != hipSuccess)  // the pages would stay pinned behind a mapping that is going away: say so
                fprintf(stderr, "transport_analysis_amd: hipHostUnregister(%p): %s\n", (void*)(b.base + k * HostBlock::kChunk), hipGetErrorString(e));
        }
    munmap(b.base, b.bytes);
    b.base = nullptr, b.bytes = 0, b.locked.clear();
}

// api.hip (ta_stage_alloc): the reference's np.zeros for a fresh pinned slab, on the staging threads with
// streaming stores (6 GB at 10000 x 50000 x 3 x float32: a single memset is a third of _prepare)
void host_zero(void* p, size_t bytes) {
    static const float zeros[4] = {0.f, 0.f, 0.f, 0.f};
    (void)zeros;
    // one "frame" of 16-byte rows per 1 MiB piece, as jobs for the pool: float -> float copy_block does not apply
    // (no source), so the pieces are zeroed by plain tasks
    const size_t piece = (size_t)8 << 20;
    const size_t n = (bytes + piece - 1) / piece;
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (;;) {
            const size_t k = next.fetch_add(1, std::memory_order_relaxed);
            if (k >= n) return;
            char* d = (char*)p + k * piece;
            const size_t len = std::min(piece, bytes - k * piece);
            const size_t head = std::min<size_t>(len, (16 - ((uintptr_t)d & 15)) & 15);
            memset(d, 0, head);
            const size_t body = (len - head) & ~(size_t)63;
            const __m128i z = _mm_setzero_si128();
            for (size_t q = 0; q < body; q += 64) {
                _mm_stream_si128((__m128i*)(d + head + q), z);
                _mm_stream_si128((__m128i*)(d + head + q + 16), z);
                _mm_stream_si128((__m128i*)(d + head + q + 32), z);
                _mm_stream_si128((__m128i*)(d + head + q + 48), z);
            }
            memset(d + head + body, 0, len - head - body);
        }
    };
    const int helpers = bytes >= ((size_t)64 << 20) ? std::min(7, std::max(0, (int)std::thread::hardware_concurrency() / 2 - 1)) : 0;
    std::vector<std::thread> th;
    for (int i = 0; i < helpers; ++i) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    _mm_sfence();
}
}  // namespace ta

extern "C" {

}  // extern "C"
