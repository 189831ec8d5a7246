/* ta_hip.h — C-ABI of the MI355X time-correlation library (libta_hip.so).
 *
 * This is the drop-in boundary for transport-analysis's time-correlation hot
 * path.  The reference has no native seam of its own (it is pure Python); the
 * entry points below are what a ctypes binding inside the reference's
 * `_prepare` / `_single_frame` / `_conclude` hooks calls, and each one cites
 * the reference code it replaces (paths relative to
 * /root/reference/transport_analysis).  See INTEGRATION.md for the binding.
 *
 * Conventions
 *   - every call returns int: 0 = TA_OK, negative = error (TA_E_*);
 *     ta_last_error() returns a human-readable message for the last failure
 *     on that context (or on the calling thread when ctx is NULL).
 *   - no C++ exception crosses this boundary: every entry point's body runs inside a catch-all
 *     (ta::guard, csrc/ta_internal.hpp); std::bad_alloc comes back as TA_E_NOMEM, any other
 *     exception as TA_E_HIP, both with a message; work already queued on behalf of the call is
 *     drained first, so the caller may free its arrays.  (Test hooks of ta_set_option:
 *     "fail_alloc_after" / "fail_throw_after" n make the n-th call of the library's allocation
 *     helper throw, tests/test_gpu_parity.py::test_exception_inside_the_library_becomes_a_status.)
 *   - HOST slabs (and the frame-major d_* inputs of the *_dev calls) are the reference's
 *     layout: (n_frames, n_atoms, dim) row-major (velocityautocorr.py:150-152,
 *     viscosity.py:128-134).  The library's own DEVICE slabs are "pair-major": the staging
 *     calls transpose frames as they are committed, so that a column pair (columns 2p, 2p+1
 *     of the n_atoms*dim columns) is one contiguous array of 16-byte rows (x[t], y[t]):
 *         element (t, c) -> slab[((c / 2) * pitch + t) * 2 + (c & 1)],  pitch = n_frames
 *     rounded up to 8; an odd last column is paired with zeros.  Every kernel reads that
 *     layout (coalesced along time); frame-major *_dev inputs are transposed into a scratch
 *     slab first (one more pass over the data and a second copy of it).
 *   - outputs are caller-owned.  "lagsum" outputs are lag-indexed SUMS over the
 *     atoms handled by this call, already divided by the frame-pair count
 *     (n_frames - lag): lagsum[k] = sum_n by_particle[k, n].  The caller divides
 *     by the total atom count after the (optional) cross-GPU reduce; this is the
 *     only cross-atom operation of the path (velocityautocorr.py:214,237,
 *     viscosity.py:233).
 *   - by_particle outputs are (n_frames, ld_bp) row-major float64 with
 *     ld_bp >= n_atoms (results.vacf_by_particle / results.visc_by_particle,
 *     velocityautocorr.py:145-147, viscosity.py:117-119); pass NULL to skip.
 *   - pointers named d_* are DEVICE pointers valid on the context's GPU;
 *     pointers named h_* are host pointers.
 *   - `stream` is a hipStream_t (void*); NULL = the legacy default (null) stream, exactly
 *     as for a HIP runtime call, so work queued by the caller on its default stream
 *     (PyTorch's default stream is the null stream) is ordered with the library's.
 *     The *_dev entry points are asynchronous on that stream; host-facing entry points
 *     run on a private non-blocking stream of the context and block until results are
 *     in the host buffers.
 *   - one context per analysis object; calls on one context are not re-entrant.
 *   - there is NO CPU fallback: without a usable GPU ta_ctx_create fails.
 */
#ifndef TA_HIP_H
#define TA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TA_OK 0
#define TA_E_INVALID -1   /* bad argument (shape, NULL, unsupported dim) */
#define TA_E_NOMEM -2     /* host or device allocation failed */
#define TA_E_HIP -3       /* a HIP runtime call or kernel launch failed */
#define TA_E_STATE -4     /* call order violated (e.g. compute before staging) */
#define TA_E_UNSUPPORTED -5

#define TA_F32 0
#define TA_F64 1

/* ta_ctx_create(TA_DEVICE_CPU, ...): the OPT-IN CPU backend behind the same symbols (csrc/cpu_backend.cpp, C++/OpenMP,
 * SURVEY.md section 8(b)): host slabs only, ta_stage_alloc / ta_stage_frame / ta_stage_commit (a no-op) / ta_vacf_fft /
 * ta_vacf_direct / ta_helfand_msd / ta_stage_synth (into the host slab) / ta_set_option ("cpu_threads") / ta_stage_free /
 * ta_trim work as documented below and
 * compute on the host cores; every device-facing call (ta_stage_alloc_device, *_dev, *_staged, ta_stage_commit_dev,
 * timings, ta_group_*) returns TA_E_UNSUPPORTED.  It is never chosen on the caller's behalf: every other
 * device index is a GPU and fails without one.                                                              */
#define TA_DEVICE_CPU (-1)

typedef struct ta_ctx ta_ctx;

/* ---- context ---------------------------------------------------------- */
int ta_ctx_create(int device, ta_ctx **out);
int ta_ctx_destroy(ta_ctx *ctx);
/* the returned string is a per-thread copy: valid until the calling thread's next ta_last_error() */
const char *ta_last_error(const ta_ctx *ctx);
/* number of visible HIP devices (0 when none / runtime unusable) */
int ta_device_count(void);
/* library ABI version, bumped on any signature change */
int ta_abi_version(void);

/* ---- staging: replaces the per-frame slab fills ------------------------
 * VelocityAutocorr._prepare/_single_frame (velocityautocorr.py:142-153,178-194)
 * ViscosityHelfand._prepare/_single_frame (viscosity.py:111-142,167-199)
 *
 * ta_stage_alloc allocates n_slabs pinned host slabs of (n_frames, n_atoms,
 * dim) elements of `dtype` (TA_F32 is lossless for MDAnalysis data, which is
 * float32 at the source, and halves the PCIe bytes) plus float64 pair-major
 * device slabs; h_slabs[i] receives the host pointers, which the Python side
 * wraps as NumPy arrays and fills frame by frame.  ta_stage_commit moves
 * frames [frame_lo, frame_hi) of every slab host->device asynchronously in
 * their native width (<= 64 MiB pieces) and transposes them into the device
 * slabs there (the transposition of one piece runs while the next one crosses
 * PCIe).  The rows of a committed range must not be written again before the next compute / ta_stage_* call
 * on the context has returned (the copies are asynchronous, and by default queued to a worker thread:
 * option "async_commit").  Frames never committed read as zeros (np.zeros in the reference).
 * ta_stage_alloc_device: device slabs only, for data that is already on the
 * GPU; fill them with ta_stage_commit_dev (frame-major float32/float64 rows
 * [frame_lo, frame_hi) at d_src, row stride ld_row elements; asynchronous on
 * `stream`) or ta_stage_synth.  ta_stage_read_dev writes a frame-major
 * float64 copy of a slab (diagnostics / tests).  ta_stage_device returns the
 * pair-major device slab i, its pitch (rows per pair) and its pair count
 * (valid until the next ta_stage_alloc* / ta_stage_free / ta_ctx_destroy).
 * The pinned host views die with ta_stage_free / the next ta_stage_alloc.   */
int ta_stage_alloc(ta_ctx *ctx, int64_t n_frames, int64_t n_atoms, int dim, int dtype,
                   int n_slabs, void **h_slabs);
int ta_stage_alloc_device(ta_ctx *ctx, int64_t n_frames, int64_t n_atoms, int dim, int n_slabs);
int ta_stage_commit(ta_ctx *ctx, int64_t frame_lo, int64_t frame_hi);
/* The per-frame fill itself, natively: replaces
 *     self._velocities[i] = self.atomgroup.velocities[:, self._dim]        (velocityautocorr.py:192-194,
 *     viscosity.py:189-199; `atomgroup.velocities` is a gather ts.velocities[atomgroup.ix] into a temporary)
 * by ONE pass from the Timestep's own array into row `frame` of pinned slab `slab`:
 *     slab[frame, a, k] = (slab dtype) h_src[row(a) * ld_row + col0 + k * col_step],   a < n_atoms, k < n_col,
 *     row(a) = h_index ? h_index[a] : atom_lo + a
 * h_src: the frame's (n_atoms_universe, ld_row) float32 / float64 (src_dtype) rows, ld_row = 3 for an MDAnalysis
 * Timestep; n_col and n_atoms must equal the slab's dim and atom count.  The copy runs on a few host threads
 * ($TA_AMD_STAGE_THREADS, default 4 including the caller; ta_stage_threads() says how many); a ctypes caller's
 * GIL is released meanwhile.  ta_group_stage_frame: the same into every member's slab, member i taking atoms
 * [lo_i, hi_i) of the n_atoms (h_index + lo_i, or atom_lo + lo_i).  Callers keep the reference's guards
 * (ts.has_velocities etc.) in front of it; a source that is not a dense float array is staged through the
 * NumPy view of ta_stage_alloc instead.                                                        */
int ta_stage_frame(ta_ctx *ctx, int slab, int64_t frame, const void *h_src, int src_dtype, int64_t ld_row,
                   int col0, int col_step, int n_col, int64_t atom_lo, const int64_t *h_index, int64_t n_atoms);
int ta_stage_threads(void);
int ta_stage_commit_dev(ta_ctx *ctx, int slab, const void *d_src, int dtype, int64_t ld_row,
                        int64_t frame_lo, int64_t frame_hi, void *stream);
int ta_stage_read_dev(ta_ctx *ctx, int slab, double *d_dst, int64_t ld_row, void *stream);
int ta_stage_device(ta_ctx *ctx, int slab, double **d_slab, int64_t *pitch_rows, int64_t *n_pairs);
int ta_stage_free(ta_ctx *ctx);
/* Benchmark input (a stateless counter-based generator, so that the CPU baseline
 * and every GPU shard materialise the same tensor without shipping it): element (t, c) of the
 * slab = synth(seed, t * n_cols_total + col_offset + c), where synth(seed, i) is the sum of the
 * eight 16-bit fields of splitmix64(seed + 2 i) and splitmix64(seed + 2 i + 1), minus 262140,
 * times 1/sqrt(8 (65536^2 - 1) / 12): zero mean, unit variance, bit-identical in NumPy
 * (oracle/synth.py).  Asynchronous on `stream`.                                            */
int ta_stage_synth(ta_ctx *ctx, int slab, uint64_t seed, int64_t col_offset, int64_t n_cols_total,
                   void *stream);
/* Release the context's cached workspaces.  They are sized by the largest call so far and kept
 * until this call or ta_ctx_destroy: partial spectra (<= 42 MB), with a by-particle array the
 * atom-major scratch (n_atoms * n_frames * 8 bytes) and the power spectra of one block of atoms
 * (2.5 GiB unless "bp_spec_atoms" says otherwise), the pair-major copies of frame-major *_dev
 * inputs (the input's size, twice for Helfand), the 64 MiB landing buffer of ta_stage_commit
 * and the product slab of the "helfand_fft" option (the input's size).                        */
int ta_trim(ta_ctx *ctx);

/* ---- pinned host memory for result arrays ---------------------------------
 * results.vacf_by_particle / results.visc_by_particle (velocityautocorr.py:145-147,
 * viscosity.py:117-119) are (n_frames, n_atoms) float64: 8 GB at 10000 x 100000.  Into a pageable
 * array the device->host copy of a host-facing call runs at ~16 GB/s the first time (the runtime
 * pins the pages on first use), into pinned memory at the link's rate.  ta_host_alloc returns
 * page-locked host memory that does NOT belong to a context (the result outlives the analysis
 * object); the Python side wraps it as the NumPy array it hands out and frees it with
 * ta_host_free when the last view dies.                                                   */
int ta_host_alloc(int64_t n_bytes, void **h_out);
/* the same with the calling thread bound to `device` first (hipSetDevice): a helper thread that
 * page-locks the array while frames are staged would otherwise create a HIP context on GPU 0 from
 * every rank.  device < 0: the calling thread's current device, as ta_host_alloc.            */
int ta_host_alloc_on(int device, int64_t n_bytes, void **h_out);
int ta_host_free(void *h);

/* ---- compute on staged slabs (host-facing, blocking) -------------------
 * ta_vacf_fft     : VelocityAutocorr._conclude_fft    (velocityautocorr.py:208-215,
 *                   incl. tidynamics.acf at :211-213)
 * ta_vacf_direct  : VelocityAutocorr._conclude_simple (velocityautocorr.py:217-238)
 * ta_helfand_msd  : ViscosityHelfand._conclude        (viscosity.py:201-233);
 *                   slab 0 = velocities, slab 1 = positions; `scale` is
 *                   1 / (2 * kB * mean(volumes) * temp_avg) (viscosity.py:229-231)
 * h_timeseries: (n_frames,) = mean over atoms; h_by_particle: (n_frames, n_atoms)
 * or NULL.  ta_vacf_fft beyond 163840 frames (the largest FFT plan) is evaluated by the
 * direct correlator (same quantity: the reference asserts their equality).      */
int ta_vacf_fft(ta_ctx *ctx, double *h_timeseries, double *h_by_particle);
int ta_vacf_direct(ta_ctx *ctx, double *h_timeseries, double *h_by_particle);
int ta_helfand_msd(ta_ctx *ctx, const double *h_masses, double scale, double *h_timeseries,
                   double *h_by_particle);

/* ---- compute on caller-provided device memory (asynchronous) -----------
 * Same arithmetic as above on a device-resident FRAME-MAJOR shard: d_vel / d_pos are
 * (n_frames, n_atoms, dim) float64 with row stride ld_row elements between
 * frames (ld_row >= n_atoms*dim; == for a dense slab, larger when the shard is a
 * column block of a wider slab).  The shard is first transposed into a pair-major
 * scratch slab owned by the context (see Conventions; ta_stage_alloc_device +
 * ta_stage_commit_dev + ta_*_staged avoid the copy).  d_lagsum: (n_frames,) SUM over
 * this shard's atoms.  d_by_particle: (n_frames, ld_bp) or NULL.                   */
int ta_vacf_fft_dev(ta_ctx *ctx, const double *d_vel, int64_t n_frames, int64_t n_atoms,
                    int dim, int64_t ld_row, double *d_lagsum, double *d_by_particle,
                    int64_t ld_bp, void *stream);
int ta_vacf_direct_dev(ta_ctx *ctx, const double *d_vel, int64_t n_frames, int64_t n_atoms,
                       int dim, int64_t ld_row, double *d_lagsum, double *d_by_particle,
                       int64_t ld_bp, void *stream);
int ta_helfand_msd_dev(ta_ctx *ctx, const double *d_vel, const double *d_pos,
                       const double *d_masses, int64_t n_frames, int64_t n_atoms, int dim,
                       int64_t ld_row, double scale, double *d_lagsum,
                       double *d_by_particle, int64_t ld_bp, void *stream);

/* ---- compute on the staged (pair-major) slabs, device outputs, asynchronous on `stream` ----
 * Same arithmetic and outputs as the *_dev calls, on the slabs of ta_stage_alloc*: no
 * transposition, no second copy.  d_masses: (n_atoms,) float64 device array.              */
int ta_vacf_fft_staged(ta_ctx *ctx, double *d_lagsum, double *d_by_particle, int64_t ld_bp,
                       void *stream);
int ta_vacf_direct_staged(ta_ctx *ctx, double *d_lagsum, double *d_by_particle, int64_t ld_bp,
                          void *stream);
int ta_helfand_msd_staged(ta_ctx *ctx, const double *d_masses, double scale, double *d_lagsum,
                          double *d_by_particle, int64_t ld_bp, void *stream);

/* ---- several GPUs behind one call (one process, one frame loop) ---------------------------
 * the multi-GPU fan-out and the reduce happen INSIDE the call.  A group owns
 * one context per entry of device_ids; member i stages and correlates the contiguous atom range
 * [n_atoms i / n_dev, n_atoms (i + 1) / n_dev) (ta_group_shard; the same split as one process per
 * GPU under torch.distributed), so the host fills n_dev pinned slabs per frame -- each GPU's column
 * block of the reference's slab (velocityautocorr.py:150-152,192-194; viscosity.py:128-134,189-199)
 * -- in ONE pass over the trajectory.  A compute call queues every member's kernels on its own
 * device, adds the members' (n_frames,) lag sums ONCE and divides by the total atom count
 * (velocityautocorr.py:214,237; viscosity.py:233); by-particle blocks are copied device->host
 * straight into the column range [lo_i, hi_i) of the caller's ONE (n_frames, n_atoms) array.
 * The reduce: n_dev = 1 none (librccl is not loaded); n_dev > 1 on distinct devices ncclReduce in
 * one RCCL group call (librccl.so dlopen'ed on first use; communicators from ncclCommInitAll);
 * members that share a device, or RCCL failing: peer copies to the first member + a sum in
 * member order.  ta_group_reduce_kind names what the last call used: "none" | "rccl" | "peer-copy";
 * ta_group_reduce_note says why an automatic choice fell back from RCCL to peer copies ("" when it
 * did not), ta_group_rccl_ranks how many ranks the communicator of the last RCCL reduce had
 * (ncclCommCount).  ta_group_set_option keys of the group itself (every other key goes to the
 * members' contexts, as ta_set_option):
 *   "reduce_mode" 0|1|2 : 0 automatic (above; also $TA_AMD_GROUP_REDUCE=auto), 1 peer copies whatever
 *                      the devices (=peer), 2 RCCL or an error (=rccl) -- with ONE member this runs
 *                      ncclCommInitAll(1) + ncclReduce onto itself, the form of the RCCL branch a
 *                      one-GPU box can execute;  "force_rccl" 1 = "reduce_mode" 2.
 * h_slabs of ta_group_stage_alloc: n_dev * n_slabs pointers, member i's slab s at [i * n_slabs + s]
 * (NULL for a member without atoms: more devices than atoms), each (n_frames, hi_i - lo_i, dim).
 * h_masses of ta_group_helfand_msd: all n_atoms.  Options go to every member.                */
typedef struct ta_group ta_group;
int ta_group_create(const int *device_ids, int n_dev, ta_group **out);
int ta_group_destroy(ta_group *g);
const char *ta_group_last_error(const ta_group *g);
int ta_group_size(const ta_group *g);
int ta_group_member(ta_group *g, int i, ta_ctx **ctx, int *device);
int ta_group_shard(const ta_group *g, int64_t n_atoms, int i, int64_t *atom_lo, int64_t *atom_hi);
const char *ta_group_reduce_kind(const ta_group *g);
const char *ta_group_reduce_note(const ta_group *g);
int ta_group_rccl_ranks(const ta_group *g);
int ta_group_set_option(ta_group *g, const char *key, int64_t value);
int ta_group_stage_alloc(ta_group *g, int64_t n_frames, int64_t n_atoms, int dim, int dtype,
                         int n_slabs, void **h_slabs);
int ta_group_stage_commit(ta_group *g, int64_t frame_lo, int64_t frame_hi);
int ta_group_stage_frame(ta_group *g, int slab, int64_t frame, const void *h_src, int src_dtype, int64_t ld_row,
                         int col0, int col_step, int n_col, int64_t atom_lo, const int64_t *h_index, int64_t n_atoms);
/* device slabs only + the benchmark generator on every member's columns of the one synthetic
 * tensor (member i: col_offset + lo_i * dim), as ta_stage_alloc_device / ta_stage_synth          */
int ta_group_stage_alloc_device(ta_group *g, int64_t n_frames, int64_t n_atoms, int dim, int n_slabs);
int ta_group_stage_synth(ta_group *g, int slab, uint64_t seed, int64_t col_offset, int64_t n_cols_total);
int ta_group_stage_free(ta_group *g);
int ta_group_vacf_fft(ta_group *g, double *h_timeseries, double *h_by_particle);
int ta_group_vacf_direct(ta_group *g, double *h_timeseries, double *h_by_particle);
int ta_group_helfand_msd(ta_group *g, const double *h_masses, double scale, double *h_timeseries,
                         double *h_by_particle);

/* ---- instrumentation ----------------------------------------------------
 * Device time of the last *_dev / host-facing compute call on this context,
 * measured with hipEvents recorded on the stream the kernels were launched on.
 * total_ms covers the whole launch sequence; main_kernel_ms only the dominant
 * kernel (FFT accumulate pass / direct correlator).  Blocks until the events
 * have completed.                                                            */
int ta_last_timing(ta_ctx *ctx, float *total_ms, float *main_kernel_ms);
/* the same for the last min(max_n, 64, calls so far) compute calls on this context, oldest
 * first (a caller times K calls back to back and reads the K durations afterwards);
 * *n_out = number of entries written.  Blocks until those calls have completed.      */
int ta_timing_history(ta_ctx *ctx, int max_n, float *total_ms, float *main_kernel_ms, int *n_out);
/* With the "timeline" option on, every compute call records an event before each of its kernel
 * launches.  ta_kernel_timeline returns, for the last compute call, the device time per kernel
 * NAME in order of first appearance (a kernel launched once per block of atoms is summed):
 * names[i] (static strings owned by the library), ms[i], *n_out entries (<= max_n).  The sum is
 * the call's total_ms.  Blocks until the call has completed.                              */
int ta_kernel_timeline(ta_ctx *ctx, int max_n, const char **names, float *ms, int *n_out);
/* The clock the headline kernel actually runs at (MI355X lowers it under load; board power and
 * the driver's sclk are not the test).  Launches a DIAGNOSTIC build of the lag-sum forward kernel
 * (in-kernel s_memtime / s_memrealtime stamps; the product kernels execute no stamp) n_launches
 * times back to back on the staged float64 slab -- ask for >= 2 s worth -- and reports, from the
 * last launch: *mhz = delta s_memtime / delta s_memrealtime x 100 MHz (mean over workgroups),
 * *cycles_per_unit_pass = shader cycles one workgroup spends per column pair and pass, and
 * *ms_per_launch (events around all launches; the stamped build is a few per cent slower than the
 * product kernel).  Plans R0 = 8, 10, 12, 16, 20 without an outer radix (n_frames in (3584, 4096],
 * (4608, 5120], (5120, 6144], (7168, 8192], (9216, 10240]); otherwise TA_E_UNSUPPORTED.        */
int ta_clock_probe(ta_ctx *ctx, int n_launches, double *mhz, double *cycles_per_unit_pass,
                   double *ms_per_launch);
/* FFT length bookkeeping for a given n_frames: *m_out = padded half-length M
 * (the transform computes a 2M-point correlation, 2M >= 2*n_frames-1): M = R * R0 * 512 with
 * R0 in {1,...,10,12,14,16,18,20} and the outer radix R = 1 up to 10240 frames (one on-chip
 * transform per pass), R in {2,3,4,5,8,16} with R0 in {12,14,16,18,20} up to 163840 frames (n_stages
 * counts the outer step).
 * Beyond that ta_vacf_fft* compute the same quantity with the direct correlator and this
 * call returns TA_E_UNSUPPORTED.                                              */
int ta_fft_plan_info(int64_t n_frames, int64_t *m_out, int *n_threads, int *n_stages);
/* options (key, value):
 *   "direct_f32" 0|1 : direct correlators (ta_vacf_direct*, ta_helfand_msd*) round the staged values (Helfand:
 *                      P = (m v) x, formed in float64) ONCE to float32, form products / squared differences
 *                      in float32 and accumulate in float64 (BASELINE configs[4]'s float32 path; within 2e-6
 *                      of the series' scale).  Helfand runs on the FP32 matrix cores (band32tp_kernels.hpp:
 *                      v_mfma_f32_16x16x4_f32 on a float32 product slab, T*A*D*4 bytes more), with or without
 *                      the by-particle array, any dim: 2.2x the float32 vector kernel; the windowed VACF stays on
 *                      the vector kernel.  Default 0 = float64.
 *   "direct_mfma" 1|0|3: ta_vacf_direct* and ta_helfand_msd* on the matrix cores (float64: FP64,
 *                      v_mfma_f64_16x16x4_f64; under "direct_f32": FP32 for Helfand) -- bandbp_kernels.hpp,
 *                      band32tp_kernels.hpp: the instruction's k-slots are filled from the time axis, a particle's
 *                      columns live in a per-wave LDS ring; with or without the by-particle arrays; Helfand from
 *                      products of rows centred on a nearby frame (every lag and particle within 1e-9 of the
 *                      difference-first vector kernel, pure trend included; needs T*A*D*8 (float32: *4) bytes for
 *                      the product slab, else the vector kernel runs).
 *                      1 (default) = by n_frames (these kernels fill a ring and run an epilogue per particle and lag
 *                      group; the vector kernel packs 2 - 8 particles into a wave under ~640 frames): windowed VACF from
 *                      513 frames, Helfand float64 from 352, its float32 option from 448 (below: the vector kernel; up
 *                      to 64 frames see "short_max").  0 = the vector kernels everywhere; 3 = matrix cores always.
 *                      (2, the column-packed forms of rounds 4-5 with their inline-assembly LDS-DMA, is rejected
 *                      since round 6: those kernels are tools/band/, built on demand as a second opinion.)
 *   "short_max" n    : trajectories of up to n frames (default and maximum 64; 0 = never) take the register-resident
 *                      kernels of short_kernels.hpp wherever float64 arithmetic on float64 slabs is asked for a
 *                      by-particle array (all three quantities) or an O(T^2) form ("direct_mfma" 1 only: 0 and 3 force
 *                      their forms): a lane per column, every lag in its registers, the by-particle array written in
 *                      place.  "short_lags_max" n (default 48): the lag sums of ta_vacf_fft* alone as well, up to n frames.
 *   "helfand_fft" 0|1: ta_helfand_msd* evaluate the mean squared differences in O(T log T)
 *                      (n_frames <= 163840, else as default): sum (P[i]-P[i+k])^2 = S1(k) - 2 S2(k), S2 by the FFT
 *                      lag sums of the product slab P = (m v) x, S1 by prefix sums.  An
 *                      extension (the reference has only the O(T^2) loop,
 *                      viscosity.py:201-233, which stays the default): ~1e-15 of the series'
 *                      scale, but the relative error of lags whose mean squared difference
 *                      is far below P^2 grows by that ratio.  Needs T*A*D*8 bytes more.
 *   "bp_block" n     : host-facing calls with a by-particle array process atoms in blocks of n
 *                      (rounded up to 64; default 16384) so that a block's device->host copy
 *                      runs under the next block's compute;
 *   "fft_nwg", "direct_nwg" : persistent workgroup counts (0 = automatic);
 *   "direct_chunk" 0|8|10, "direct_groups" n : force the direct correlators' lags per chunk /
 *                      cap the atoms a workgroup works on at once (0 = automatic);
 *   "mid_max" n      : trajectories of 97 ... n frames (default and maximum 512; 0 = never) take k_mid (mid_kernels.hpp: a
 *                      lane per column and pair of 16-lag blocks, a sliding window in registers) for the windowed VACF, and
 *                      for the Einstein-Helfand sums up to 128 frames, under "direct_mfma" 1 and float64 arithmetic;
 *                      "mid_all" 1: wherever the kernel can run (65 ... 512 frames, both quantities); "mid_ncl" 3..6:
 *                      log2 of the lanes per pair of lag blocks, i.e. the columns per tile (tools/mid_shapes.py; 0: by length);
 *   "direct_subwave" 1|0 : the vector kernel's column groups may be 8, 16 or 32 lanes where a column has that few
 *                      pairs of lag chunks (under ~640 frames): several particles per wave, 3x at 65 ... 256 frames
 *                      (0: a whole wave per column, as before round 6);
 *   "stage_device_f32" 0|1 : device slabs allocated AFTERWARDS hold float32 elements when the host
 *                      slabs are TA_F32 (or there are none: ta_stage_alloc_device): half the
 *                      device footprint (BASELINE configs[4]: 12 GB instead of 24 GB per GPU).
 *                      The float32 direct correlators ("direct_f32" 1) read them as they are, and
 *                      so do the FFT kernels for 513 ... 10240 frames (8-byte rows widened exactly
 *                      in the first stage: 3-14 % faster than from float64 slabs, float64
 *                      arithmetic, results equal to float64 slabs of the same values to rounding);
 *                      every other evaluation first widens them into float64 scratch slabs.  ta_stage_device
 *                      then returns a pointer to float rows (8 bytes per pair row).
 *   "bp_spec_atoms" n : FFT path with a by-particle array: atoms per block of power spectra
 *                      (scratch = n * 16 * M bytes; 0 = as many as fit 2.5 GiB);
 *   "bp_prefetch" 0..3 : sub-series of the next atom's spectrum the inverse kernel requests
 *                      ahead (default 2);
 *   "timeline" 0|1   : record an event before every kernel launch of a compute call
 *                      (ta_kernel_timeline);
 *   "async_commit" 1|0 : ta_stage_commit hands its frame range to a worker thread of the context, which
 *                      makes the HIP calls (the caller's frame loop never waits on the runtime, e.g. while
 *                      another thread page-locks a result array); every call that touches the slabs joins
 *                      the queue first and returns a queued commit's error.  0: the calls are made by
 *                      ta_stage_commit itself.   Unknown keys return TA_E_INVALID.          */
int ta_set_option(ta_ctx *ctx, const char *key, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* TA_HIP_H */
