// group.hip — several GPUs behind ONE call ("ta_ctx_create(device_ids[], n_dev, &ctx)
// ... multi-GPU fan-out and the RCCL reduce happen inside the call"; 8(e): atoms are the sharded
// axis, one reduce of the (n_frames,) lag sums, by-particle blocks copied into column ranges of one
// host array).  One host thread, one frame loop: member i owns a ta_ctx on device_ids[i], stages
// and correlates atoms [A i / n, A (i + 1) / n) and the members' lag sums are added once:
//   n_dev = 1                      no reduce (and librccl is never loaded);
//   n_dev > 1, distinct devices    ncclReduce(sum, float64, n_frames) in one RCCL group call, the
//                                  communicators from ncclCommInitAll (one process, n devices);
//                                  librccl.so is dlopen'ed on first use;
//   otherwise (two members on one GPU -- what a one-GPU box can test -- or RCCL failing)
//                                  the members' vectors are copied to member 0's device
//                                  (hipMemcpyPeerAsync) and added there in member order.
// "reduce_mode" (ta_group_set_option; $TA_AMD_GROUP_REDUCE = auto | peer | rccl) overrides the choice:
// 1 = peer copies whatever the devices, 2 = RCCL or an error ("force_rccl" 1 is the same), which also
// runs the collective for ONE member: the only form of this branch a one-GPU box can execute.
// Replaces, for several GPUs at once, the same reference code as the single-context calls:
// velocityautocorr.py:142-153,178-238, viscosity.py:111-142,167-233 (the atom mean at
// velocityautocorr.py:214,237 / viscosity.py:233 is the reduce + one division).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>  // types and prototypes only: librccl.so is dlopen'ed (struct Rccl)
#else  // a ROCm install without the RCCL development headers: the few declarations this file needs, as RCCL 2.x publishes them
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclFloat64 = 8 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count);
ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/ta_hip.h"
#include "ta_internal.hpp"

using namespace ta;

namespace {

thread_local std::string g_group_tls_error;

// The RCCL entry points of the reduce, resolved from librccl.so when first needed (the library does
// not link librccl: a one-GPU user never loads it).  <rccl/rccl.h> is included for its TYPES only: every
// pointer below is declared as decltype(&ncclX), so a call through it is compiled against the
// header's own prototype, and the static_asserts pin the prototypes this file was written against
// (a header that reorders or retypes an argument fails the build instead of corrupting a call).
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;  // (optional: a library without it destroys the communicator instead)
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load(std::string* why) {
        if (handle) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!handle) {
            const char* e = dlerror();
            *why = std::string("librccl.so not loadable: ") + (e ? e : "?");
            return false;
        }
        auto sym = [&](const char* n) { return dlsym(handle, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
        CommCount = (decltype(CommCount))sym("ncclCommCount");
        Reduce = (decltype(Reduce))sym("ncclReduce");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !Reduce || !GroupStart || !GroupEnd) {
            *why = "librccl.so lacks an entry point of the reduce";
            dlclose(handle);
            handle = nullptr;
            return false;
        }
        return true;
    }
    std::string what(ncclResult_t rc) const { return GetErrorString ? GetErrorString(rc) : "failed"; }
};
Rccl g_rccl;
static_assert(std::is_same_v<decltype(&ncclCommInitAll), ncclResult_t (*)(ncclComm_t*, int, const int*)>);
static_assert(std::is_same_v<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>);
static_assert(std::is_same_v<decltype(&ncclCommCount), ncclResult_t (*)(const ncclComm_t, int*)>);
static_assert(std::is_same_v<decltype(&ncclReduce), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t,
                                                                     int, ncclComm_t, hipStream_t)>);
static_assert(std::is_same_v<decltype(&ncclGroupStart), ncclResult_t (*)()>);
static_assert(std::is_same_v<decltype(&ncclGroupEnd), ncclResult_t (*)()>);
static_assert(std::is_same_v<decltype(&ncclGetErrorString), const char* (*)(ncclResult_t)>);
static_assert(ncclFloat64 == 8 && ncclSum == 0 && ncclSuccess == 0);

// how the members' lag sums are added: "reduce_mode" option, $TA_AMD_GROUP_REDUCE = auto | peer | rccl
enum { RED_AUTO = 0, RED_PEER = 1, RED_RCCL = 2 };

}  // namespace

struct ta_group {
    std::vector<int> devices;
    std::vector<ta_ctx*> ctx;
    std::vector<int64_t> lo, hi;  // atom range of member i (after ta_group_stage_alloc)
    int64_t T = 0, A = 0;
    int D = 0, n_slabs = 0;
    bool distinct = true;          // all members on different devices
    std::vector<ncclComm_t> comms;  // RCCL communicators (distinct devices), created on first reduce
    bool rccl_tried = false;
    int reduce_mode = RED_AUTO;     // "reduce_mode" option / $TA_AMD_GROUP_REDUCE
    int rccl_ranks = 0;             // ncclCommCount of the communicator the last RCCL reduce ran on
    std::string reduce_kind = "none";
    std::string reduce_note;        // why RED_AUTO fell back to peer copies (empty: it did not)
    std::string err;
    // on the reducing member's device: [n + 1][T] rows for the copy-and-add reduce; events to order the devices
    double* d_rows = nullptr;
    size_t rows_bytes = 0;
    int rows_device = -1;
    std::vector<hipEvent_t> ev;
};

namespace {

int gfail(ta_group* g, int code, const std::string& msg) noexcept {
    try {
        if (g) g->err = msg;
        g_group_tls_error = msg;
    } catch (...) {
    }
    return code;
}
#define TAG_TRY(g, expr)                                                                             \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return gfail(g, e_ == hipErrorOutOfMemory ? TA_E_NOMEM : TA_E_HIP,                       \
                         std::string(#expr) + ": " + hipGetErrorString(e_));                         \
    } while (0)
// a member's failure, with its message
int mfail(ta_group* g, int i, int rc) {
    return gfail(g, rc, "device member " + std::to_string(i) + ": " + ta_last_error(g->ctx[i]));
}

void shard(int64_t A, int i, int n, int64_t* lo, int64_t* hi) {  // as transport_analysis_amd/dist.py:atom_shard
    *lo = A * i / n;
    *hi = A * (i + 1) / n;
}

// The (n_frames,) sums of the members `who` (d_tot[j] on member who[j]'s device, ready on that member's
// stream) -> their sum on member who[0]'s device, in d_tot[0], ready on that member's stream.
// RED_AUTO: RCCL when every member holds atoms on a device of its own (n > 1), peer copies when that
// fails (reduce_note says why) or does not apply; RED_PEER: peer copies; RED_RCCL: RCCL or an error --
// also for ONE member (ncclCommInitAll(1) + ncclReduce onto itself: what a one-GPU box can execute of
// this branch, tests/test_gpu_dist.py::test_group_forced_rccl_one_member).
int reduce_members(ta_group* g, const std::vector<int>& who, std::vector<double*>& d_tot) {
    const int n = (int)g->ctx.size(), m = (int)who.size();
    const int64_t T = g->T;
    const int mode = g->reduce_mode;
    if (m <= 1 && mode != RED_RCCL) {
        g->reduce_kind = "none";
        return TA_OK;
    }
    const bool rccl_applies = m == n && g->distinct;
    if (mode == RED_RCCL && !rccl_applies)
        return gfail(g, TA_E_UNSUPPORTED, "reduce_mode rccl: every member needs atoms and a device of its own");
    if (mode != RED_PEER && rccl_applies && !g->rccl_tried) {
        g->rccl_tried = true;
        std::string why;
        if (g_rccl.load(&why)) {
            g->comms.assign(n, nullptr);
            const ncclResult_t rc = g_rccl.CommInitAll(g->comms.data(), n, g->devices.data());
            if (rc != ncclSuccess) {
                g->comms.clear();
                why = "ncclCommInitAll: " + g_rccl.what(rc);
            }
        }
        g->reduce_note = why;
    }
    if (mode != RED_PEER && rccl_applies && !g->comms.empty()) {
        // ONE collective: every member's vector is added into member 0's (in place on the root).  Nothing
        // returns between ncclGroupStart and ncclGroupEnd: an open group on this thread would swallow
        // every later RCCL call made from it (the caller's torch.distributed ones included).
        ncclResult_t rc = g_rccl.GroupStart();
        hipError_t he = hipSuccess;
        int launched = 0;  // calls RCCL accepted inside the group
        if (rc == ncclSuccess) {
            for (int i = 0; i < n && rc == ncclSuccess && he == hipSuccess; ++i) {
                he = hipSetDevice(g->devices[i]);
                if (he == hipSuccess) {
                    rc = g_rccl.Reduce(d_tot[i], d_tot[i], (size_t)T, ncclFloat64, ncclSum, 0, g->comms[i],
                                       ctx_stream(g->ctx[i]));
                    if (rc == ncclSuccess) ++launched;
                }
            }
            const ncclResult_t rc2 = g_rccl.GroupEnd();
            if (rc == ncclSuccess) rc = rc2;
        }
        if (rc == ncclSuccess && he == hipSuccess) {
            int cnt = 0;
            if (g_rccl.CommCount && g_rccl.CommCount(g->comms[0], &cnt) == ncclSuccess) g->rccl_ranks = cnt;
            g->reduce_kind = "rccl";
            return TA_OK;
        }
        const std::string why = he != hipSuccess ? std::string("hipSetDevice: ") + hipGetErrorString(he)
                                                 : "ncclReduce: " + g_rccl.what(rc);
        if (mode == RED_RCCL) return gfail(g, TA_E_HIP, why);
        // RED_AUTO.  Falling back to peer copies is only safe when NOTHING of the collective was enqueued: the reduce is
        // in place on the root's vector, and ncclGroupEnd launches whatever calls were accepted before the failing one.
        // That is the case when ncclGroupStart itself failed, or the very first member's hipSetDevice / ncclReduce did
        // (`launched` counts the calls RCCL accepted).  Otherwise the members' sums may be half-reduced and their streams
        // may wait on ranks that never arrive: the communicators are aborted and the call fails.
        if (launched > 0) {
            for (ncclComm_t c : g->comms)
                if (c) (void)(g_rccl.CommAbort ? g_rccl.CommAbort(c) : g_rccl.CommDestroy(c));
            g->comms.clear();
            g->reduce_note = why;
            return gfail(g, TA_E_HIP, "RCCL reduce failed after " + std::to_string(launched) + " of " + std::to_string(n) +
                                          " ranks had joined (" + why + "): the lag sums are not trustworthy; re-run with reduce_mode 1 (peer copies)");
        }
        for (ncclComm_t c : g->comms)
            if (c) (void)g_rccl.CommDestroy(c);
        g->comms.clear();
        g->reduce_note = why;
    } else if (mode == RED_RCCL) {
        return gfail(g, TA_E_HIP, "reduce_mode rccl: " + (g->reduce_note.empty() ? std::string("no communicator") : g->reduce_note));
    }
    if (m <= 1) {
        g->reduce_kind = "none";
        return TA_OK;
    }
    // copy-and-add on the first member's device, in member order (bitwise reproducible)
    const int root = who[0];
    TAG_TRY(g, hipSetDevice(g->devices[root]));
    const size_t need = sizeof(double) * (size_t)T * (n + 1);
    if (g->rows_bytes < need || g->rows_device != g->devices[root]) {
        if (g->d_rows) {
            (void)hipSetDevice(g->rows_device);
            (void)hipFree(g->d_rows);
            (void)hipSetDevice(g->devices[root]);
        }
        g->d_rows = nullptr, g->rows_bytes = 0;
        TAG_TRY(g, hipMalloc((void**)&g->d_rows, need));
        g->rows_bytes = need, g->rows_device = g->devices[root];
    }
    if (g->ev.empty()) {
        g->ev.assign(n, nullptr);
        for (int i = 0; i < n; ++i) {
            TAG_TRY(g, hipSetDevice(g->devices[i]));
            TAG_TRY(g, hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming));
        }
    }
    hipStream_t s0 = ctx_stream(g->ctx[root]);
    for (int j = 1; j < m; ++j) {  // member who[j]'s sum is ready on its own stream: the root's stream waits for it
        TAG_TRY(g, hipSetDevice(g->devices[who[j]]));
        TAG_TRY(g, hipEventRecord(g->ev[who[j]], ctx_stream(g->ctx[who[j]])));
    }
    TAG_TRY(g, hipSetDevice(g->devices[root]));
    for (int j = 0; j < m; ++j) {
        if (j) TAG_TRY(g, hipStreamWaitEvent(s0, g->ev[who[j]], 0));
        TAG_TRY(g, hipMemcpyPeerAsync(g->d_rows + (size_t)j * T, g->devices[root], d_tot[j], g->devices[who[j]],
                                      sizeof(double) * T, s0));
    }
    TAG_TRY(g, launch_sum_partials(g->d_rows, m, T, g->d_rows + (size_t)n * T, s0));
    d_tot[0] = g->d_rows + (size_t)n * T;
    g->reduce_kind = "peer-copy";
    return TA_OK;
}

// everything queued so far on the members `who` has finished (their kernels and their copies into the
// caller's host arrays): called before an error is returned, so that the caller may free those arrays
void drain_members(ta_group* g, const std::vector<int>& who) {
    for (int i : who) (void)host_wait(g->ctx[i]);
}

int group_compute_impl(ta_group* g, int which, const double* h_masses, double scale, double* h_ts, double* h_bp) {
    if (!g) return gfail(nullptr, TA_E_INVALID, "null group");
    if (!h_ts) return gfail(g, TA_E_INVALID, "h_timeseries is NULL");
    if (g->T == 0) return gfail(g, TA_E_STATE, "slabs have not been staged");
    if (which == 2 && !h_masses) return gfail(g, TA_E_INVALID, "h_masses is NULL");
    const int n = (int)g->ctx.size();
    std::vector<double*> d_tot;
    std::vector<int> who;  // members that hold atoms (all of them unless there are more devices than atoms)
    int rc;
    for (int i = 0; i < n; ++i) {
        if (g->hi[i] == g->lo[i]) continue;
        double* d = nullptr;
        rc = host_launch(g->ctx[i], which, h_masses ? h_masses + g->lo[i] : nullptr, scale,
                         h_bp ? h_bp + g->lo[i] : nullptr, g->A, &d);
        who.push_back(i);  // (a failed launch may have queued work already)
        if (rc) {
            rc = mfail(g, i, rc);
            drain_members(g, who);  // earlier members' copies into h_bp are in flight: not after we return
            return rc;
        }
        d_tot.push_back(d);
    }
    if ((rc = reduce_members(g, who, d_tot))) {
        drain_members(g, who);
        return rc;
    }
    const int root = who[0];
    hipError_t he = hipSetDevice(g->devices[root]);
    if (he == hipSuccess)
        he = hipMemcpyAsync(h_ts, d_tot[0], sizeof(double) * g->T, hipMemcpyDeviceToHost, ctx_stream(g->ctx[root]));
    if (he != hipSuccess) {
        drain_members(g, who);
        return gfail(g, TA_E_HIP, std::string("timeseries copy: ") + hipGetErrorString(he));
    }
    rc = TA_OK;
    for (int i : who) {  // every member is waited for, whatever an earlier one reports
        const int r = host_wait(g->ctx[i]);
        if (r && !rc) rc = mfail(g, i, r);
    }
    if (rc) return rc;
    const double n_at = (double)g->A;  // mean over ALL atoms (velocityautocorr.py:214,237; viscosity.py:233)
    for (int64_t k = 0; k < g->T; ++k) h_ts[k] /= n_at;
    return TA_OK;
}

// ... behind the guard of the entry points: an exception after the members' launches drains them first (the caller may
// free its arrays as soon as the call returns)
int group_compute(ta_group* g, int which, const double* h_masses, double scale, double* h_ts, double* h_bp) {
    return ta::guard(
        [&](int c_, const std::string& m_) {
            if (g)
                for (ta_ctx* c : g->ctx)
                    if (c) (void)host_wait(c);
            return gfail(g, c_, m_);
        },
        [&]() -> int { return group_compute_impl(g, which, h_masses, scale, h_ts, h_bp); });
}

}  // namespace

extern "C" {

int ta_group_create(const int* device_ids, int n_dev, ta_group** out) {
    return ta::guard([&](int c_, const std::string& m_) { return gfail(nullptr, c_, m_); }, [&]() -> int {
    if (!out) return gfail(nullptr, TA_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!device_ids || n_dev < 1 || n_dev > 64) return gfail(nullptr, TA_E_INVALID, "need 1..64 device ids");
    for (int i = 0; i < n_dev; ++i)
        if (device_ids[i] < 0) return gfail(nullptr, TA_E_UNSUPPORTED, "device groups are made of GPU contexts (TA_DEVICE_CPU is a context of its own)");
    ta_group* g = new (std::nothrow) ta_group();
    if (!g) return gfail(nullptr, TA_E_NOMEM, "out of host memory");
    for (int i = 0; i < n_dev; ++i) {
        ta_ctx* c = nullptr;
        const int rc = ta_ctx_create(device_ids[i], &c);
        if (rc) {
            const std::string msg = std::string("device ") + std::to_string(device_ids[i]) + ": " + ta_last_error(nullptr);
            ta_group_destroy(g);
            return gfail(nullptr, rc, msg);
        }
        g->devices.push_back(device_ids[i]);
        g->ctx.push_back(c);
        for (int j = 0; j < i; ++j)
            if (device_ids[j] == device_ids[i]) g->distinct = false;
    }
    g->lo.assign(n_dev, 0);
    g->hi.assign(n_dev, 0);
    if (const char* e = getenv("TA_AMD_GROUP_REDUCE")) {
        if (!strcmp(e, "peer")) g->reduce_mode = RED_PEER;
        else if (!strcmp(e, "rccl")) g->reduce_mode = RED_RCCL;
        else if (strcmp(e, "auto") && *e) {
            ta_group_destroy(g);
            return gfail(nullptr, TA_E_INVALID, "TA_AMD_GROUP_REDUCE must be auto, peer or rccl");
        }
    }
    // peer access for the copy-and-add reduce (harmless when RCCL does the reduce; errors ignored:
    // hipMemcpyPeerAsync works without it, through the host)
    for (int i = 1; i < n_dev; ++i)
        if (device_ids[i] != device_ids[0]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device_ids[0], device_ids[i]) == hipSuccess && can) {
                (void)hipSetDevice(device_ids[0]);
                (void)hipDeviceEnablePeerAccess(device_ids[i], 0);
                (void)hipGetLastError();
            }
        }
    *out = g;
    return TA_OK;
    });
}

int ta_group_destroy(ta_group* g) {
    return ta::guard([&](int c_, const std::string& m_) { return gfail(g, c_, m_); }, [&]() -> int {
    if (!g) return TA_OK;
    for (size_t i = 0; i < g->comms.size(); ++i)
        if (g->comms[i] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g->comms[i]);
    for (size_t i = 0; i < g->ev.size(); ++i)
        if (g->ev[i]) {
            (void)hipSetDevice(g->devices[i]);
            (void)hipEventDestroy(g->ev[i]);
        }
    if (g->d_rows) {
        (void)hipSetDevice(g->rows_device);
        (void)hipFree(g->d_rows);
    }
    for (ta_ctx* c : g->ctx) ta_ctx_destroy(c);
    delete g;
    return TA_OK;
    });
}

const char* ta_group_last_error(const ta_group* g) { return g ? g->err.c_str() : g_group_tls_error.c_str(); }

int ta_group_size(const ta_group* g) { return g ? (int)g->ctx.size() : 0; }

int ta_group_member(ta_group* g, int i, ta_ctx** ctx, int* device) {
    return ta::guard([&](int c_, const std::string& m_) { return gfail(g, c_, m_); }, [&]() -> int {
    if (!g || i < 0 || i >= (int)g->ctx.size()) return gfail(g, TA_E_INVALID, "no such member");
    if (ctx) *ctx = g->ctx[i];
    if (device) *device = g->devices[i];
    return TA_OK;
    });
}

int ta_group_shard(const ta_group* g, int64_t n_atoms, int i, int64_t* atom_lo, int64_t* atom_hi) {
    return ta::guard([&](int c_, const std::string& m_) { return gfail(const_cast<ta_group*>(g), c_, m_); }, [&]() -> int {
    if (!g || i < 0 || i >= (int)g->ctx.size() || n_atoms < 0 || !atom_lo || !atom_hi)
        return gfail(const_cast<ta_group*>(g), TA_E_INVALID, "bad argument");
    shard(n_atoms, i, (int)g->ctx.size(), atom_lo, atom_hi);
    return TA_OK;
    });
}

const char* ta_group_reduce_kind(const ta_group* g) { return g ? g->reduce_kind.c_str() : ""; }

const char* ta_group_reduce_note(const ta_group* g) { return g ? g->reduce_note.c_str() : ""; }

int ta_group_rccl_ranks(const ta_group* g) { return g ? g->rccl_ranks : 0; }

int ta_group_set_option(ta_group* g, const char* key, int64_t value) {
    return ta::guard([&](int c_, const std::string& m_) { return gfail(g, c_, m_); }, [&]() -> int {
    if (!g || !key) return gfail(g, TA_E_INVALID, "null argument");
    // the group's own options; every other key goes to the members' contexts
    if (!strcmp(key, "reduce_mode") || !strcmp(key, "force_rccl")) {
        const int mode = !strcmp(key, "force_rccl") ? (value ? RED_RCCL : RED_AUTO) : (int)value;
        if (mode < RED_AUTO || mode > RED_RCCL) return gfail(g, TA_E_INVALID, "reduce_mode: 0 auto, 1 peer copies, 2 RCCL");
        g->reduce_mode = mode;
        return TA_OK;
    }
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const int rc = ta_set_option(g->ctx[i], key, value);
        if (rc) return mfail(g, (int)i, rc);
    }
    return TA_OK;
    });
}

int ta_group_stage_alloc(ta_group* g, int64_t n_frames, int64_t n_atoms, int dim, int dtype, int n_slabs,
                         void** h_slabs) {
    return ta::guard([&](int c_, const std::string& m_) { return gfail(g, c_, m_); }, [&]() -> int {
    if (!g) return gfail(nullptr, TA_E_INVALID, "null group");
    if (!h_slabs) return gfail(g, TA_E_INVALID, "h_slabs is NULL");
    if (n_frames < 1 || n_atoms < 1 || dim < 1 || dim > 3 || n_slabs < 1 || n_slabs > 4)
        return gfail(g, TA_E_INVALID, "need n_frames >= 1, n_atoms >= 1, 1 <= dim <= 3, 1 <= n_slabs <= 4");
    const int n = (int)g->ctx.size();
    g->T = 0;
    for (int i = 0; i < n; ++i) {
        shard(n_atoms, i, n, &g->lo[i], &g->hi[i]);
        for (int s = 0; s < n_slabs; ++s) h_slabs[i * n_slabs + s] = nullptr;
        if (g->hi[i] == g->lo[i]) {  // more devices than atoms
            ta_stage_free(g->ctx[i]);
            continue;
        }
        const int rc = ta_stage_alloc(g->ctx[i], n_frames, g->hi[i] - g->lo[i], dim, dtype, n_slabs,
                                      h_slabs + (size_t)i * n_slabs);
        if (rc) return mfail(g, i, rc);
    }
    g->T = n_frames, g->A = n_atoms, g->D = dim, g->n_slabs = n_slabs;
    return TA_OK;
    });
}

int ta_group_stage_alloc_device(ta_group* g, int64_t n_frames, int64_t n_atoms, int dim, int n_slabs) {
    return ta::guard([&](int c_, const std::string& m_) { return gfail(g, c_, m_); }, [&]() -> int {
    if (!g) return gfail(nullptr, TA_E_INVALID, "null group");
    if (n_frames < 1 || n_atoms < 1 || dim < 1 || dim > 3 || n_slabs < 1 || n_slabs > 4)
        return gfail(g, TA_E_INVALID, "need n_frames >= 1, n_atoms >= 1, 1 <= dim <= 3, 1 <= n_slabs <= 4");
    const int n = (int)g->ctx.size();
    g->T = 0;
    for (int i = 0; i < n; ++i) {
        shard(n_atoms, i, n, &g->lo[i], &g->hi[i]);
        if (g->hi[i] == g->lo[i]) {
            ta_stage_free(g->ctx[i]);
            continue;
        }
        const int rc = ta_stage_alloc_device(g->ctx[i], n_frames, g->hi[i] - g->lo[i], dim, n_slabs);
        if (rc) return mfail(g, i, rc);
    }
    g->T = n_frames, g->A = n_atoms, g->D = dim, g->n_slabs = n_slabs;
    return TA_OK;
    });
}

int ta_group_stage_synth(ta_group* g, int slab, uint64_t seed, int64_t col_offset, int64_t n_cols_total) {
    return ta::guard([&](int c_, const std::string& m_) { return gfail(g, c_, m_); }, [&]() -> int {
    if (!g) return gfail(nullptr, TA_E_INVALID, "null group");
    if (g->T == 0) return gfail(g, TA_E_STATE, "slabs have not been staged");
    for (size_t i = 0; i < g->ctx.size(); ++i) {  // member i: its own columns of the ONE synthetic tensor
        if (g->hi[i] == g->lo[i]) continue;
        const int rc = ta_stage_synth(g->ctx[i], slab, seed, col_offset + g->lo[i] * g->D, n_cols_total,
                                      (void*)ctx_stream(g->ctx[i]));
        if (rc) return mfail(g, (int)i, rc);
    }
    return TA_OK;
    });
}

int ta_group_stage_commit(ta_group* g, int64_t frame_lo, int64_t frame_hi) {
    return ta::guard([&](int c_, const std::string& m_) { return gfail(g, c_, m_); }, [&]() -> int {
    if (!g) return gfail(nullptr, TA_E_INVALID, "null group");
    if (g->T == 0) return gfail(g, TA_E_STATE, "ta_group_stage_alloc has not been called");
    for (size_t i = 0; i < g->ctx.size(); ++i) {  // queued on each device's own streams: the devices overlap
        if (g->hi[i] == g->lo[i]) continue;
        const int rc = ta_stage_commit(g->ctx[i], frame_lo, frame_hi);
        if (rc) return mfail(g, (int)i, rc);
    }
    return TA_OK;
    });
}

int ta_group_stage_free(ta_group* g) {
    return ta::guard([&](int c_, const std::string& m_) { return gfail(g, c_, m_); }, [&]() -> int {
    if (!g) return gfail(nullptr, TA_E_INVALID, "null group");
    for (ta_ctx* c : g->ctx) ta_stage_free(c);
    g->T = g->A = 0;
    return TA_OK;
    });
}

int ta_group_vacf_fft(ta_group* g, double* h_ts, double* h_bp) { return group_compute(g, 0, nullptr, 1.0, h_ts, h_bp); }
int ta_group_vacf_direct(ta_group* g, double* h_ts, double* h_bp) { return group_compute(g, 1, nullptr, 1.0, h_ts, h_bp); }
int ta_group_helfand_msd(ta_group* g, const double* h_masses, double scale, double* h_ts, double* h_bp) {
    return ta::guard([&](int c_, const std::string& m_) { return gfail(g, c_, m_); }, [&]() -> int {
    return group_compute(g, 2, h_masses, scale, h_ts, h_bp);
    });
}

}  // extern "C"
