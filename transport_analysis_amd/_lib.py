"""ctypes binding of libta_hip.so (the C-ABI declared in include/ta_hip.h).

There is no CPU fallback: if the shared library is missing, or no GPU is
usable, every compute entry point raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libta_hip.so")
_CSRC = os.path.join(_HERE, "csrc")

TA_F32, TA_F64 = 0, 1

#: every symbol include/ta_hip.h declares
EXPORTS = (
    "ta_abi_version", "ta_device_count", "ta_last_error", "ta_ctx_create", "ta_ctx_destroy",
    "ta_stage_alloc", "ta_stage_alloc_device", "ta_stage_commit", "ta_stage_frame", "ta_stage_threads", "ta_stage_commit_dev",
    "ta_stage_read_dev", "ta_stage_device", "ta_stage_free", "ta_stage_synth", "ta_trim",
    "ta_vacf_fft", "ta_vacf_direct", "ta_helfand_msd",
    "ta_vacf_fft_dev", "ta_vacf_direct_dev", "ta_helfand_msd_dev",
    "ta_vacf_fft_staged", "ta_vacf_direct_staged", "ta_helfand_msd_staged",
    "ta_last_timing", "ta_timing_history", "ta_kernel_timeline", "ta_clock_probe", "ta_fft_plan_info",
    "ta_set_option",
    "ta_host_alloc", "ta_host_alloc_on", "ta_host_free",
    "ta_group_create", "ta_group_destroy", "ta_group_last_error", "ta_group_size", "ta_group_member",
    "ta_group_shard", "ta_group_reduce_kind", "ta_group_reduce_note", "ta_group_rccl_ranks", "ta_group_set_option", "ta_group_stage_alloc",
    "ta_group_stage_commit", "ta_group_stage_frame", "ta_group_stage_free", "ta_group_stage_alloc_device", "ta_group_stage_synth", "ta_group_vacf_fft", "ta_group_vacf_direct",
    "ta_group_helfand_msd",
)


class TAError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, code, message):
        super().__init__(f"libta_hip error {code}: {message}")
        self.code = code


def build(force=False):
    """Compile libta_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", _CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", _CSRC])
    return _SO


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 with
    the same SONAME as /opt/rocm's; whichever is loaded first serves both libraries, and
    torch.cuda fails ("No HIP GPUs are available") on top of a runtime that is not its own.
    So when torch is installed but not loaded yet, its runtime is loaded first -- without
    importing torch -- and libta_hip.so binds to it exactly as it does when the application
    imported torch before us."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the system runtime


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(
            f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (hipcc --offload-arch=gfx950). transport_analysis_amd has no CPU fallback."
        )
    _share_hip_runtime_with_torch()
    L = ctypes.CDLL(_SO)
    vp, i64, ci, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
    L.ta_abi_version.restype = ci
    L.ta_device_count.restype = ci
    L.ta_last_error.restype = ctypes.c_char_p
    L.ta_last_error.argtypes = [vp]
    L.ta_ctx_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.ta_ctx_destroy.argtypes = [vp]
    L.ta_stage_alloc.argtypes = [vp, i64, i64, ci, ci, ci, ctypes.POINTER(vp)]
    L.ta_stage_commit.argtypes = [vp, i64, i64]
    L.ta_stage_frame.argtypes = [vp, ci, i64, vp, ci, i64, ci, ci, ci, i64, vp, i64]
    L.ta_group_stage_frame.argtypes = [vp, ci, i64, vp, ci, i64, ci, ci, ci, i64, vp, i64]
    L.ta_stage_alloc_device.argtypes = [vp, i64, i64, ci, ci]
    L.ta_stage_commit_dev.argtypes = [vp, ci, vp, ci, i64, i64, i64, vp]
    L.ta_stage_read_dev.argtypes = [vp, ci, vp, i64, vp]
    L.ta_stage_device.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.ta_stage_free.argtypes = [vp]
    L.ta_stage_synth.argtypes = [vp, ci, ctypes.c_uint64, i64, i64, vp]
    L.ta_trim.argtypes = [vp]
    L.ta_vacf_fft_staged.argtypes = [vp, vp, vp, i64, vp]
    L.ta_vacf_direct_staged.argtypes = [vp, vp, vp, i64, vp]
    L.ta_helfand_msd_staged.argtypes = [vp, vp, dbl, vp, vp, i64, vp]
    L.ta_vacf_fft.argtypes = [vp, vp, vp]
    L.ta_vacf_direct.argtypes = [vp, vp, vp]
    L.ta_helfand_msd.argtypes = [vp, vp, dbl, vp, vp]
    L.ta_vacf_fft_dev.argtypes = [vp, vp, i64, i64, ci, i64, vp, vp, i64, vp]
    L.ta_vacf_direct_dev.argtypes = [vp, vp, i64, i64, ci, i64, vp, vp, i64, vp]
    L.ta_helfand_msd_dev.argtypes = [vp, vp, vp, vp, i64, i64, ci, i64, dbl, vp, vp, i64, vp]
    L.ta_last_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.ta_timing_history.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                    ctypes.POINTER(ci)]
    L.ta_kernel_timeline.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float),
                                     ctypes.POINTER(ci)]
    L.ta_clock_probe.argtypes = [vp, ci, ctypes.POINTER(dbl), ctypes.POINTER(dbl), ctypes.POINTER(dbl)]
    L.ta_host_alloc.argtypes = [i64, ctypes.POINTER(vp)]
    L.ta_host_alloc_on.argtypes = [ctypes.c_int, i64, ctypes.POINTER(vp)]
    L.ta_host_free.argtypes = [vp]
    L.ta_fft_plan_info.argtypes = [i64, ctypes.POINTER(i64), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.ta_set_option.argtypes = [vp, ctypes.c_char_p, i64]
    L.ta_group_create.argtypes = [ctypes.POINTER(ci), ci, ctypes.POINTER(vp)]
    L.ta_group_destroy.argtypes = [vp]
    L.ta_group_last_error.argtypes = [vp]
    L.ta_group_last_error.restype = ctypes.c_char_p
    L.ta_group_size.argtypes = [vp]
    L.ta_group_member.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(ci)]
    L.ta_group_shard.argtypes = [vp, i64, ci, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.ta_group_reduce_kind.argtypes = [vp]
    L.ta_group_reduce_kind.restype = ctypes.c_char_p
    L.ta_group_reduce_note.argtypes = [vp]
    L.ta_group_reduce_note.restype = ctypes.c_char_p
    L.ta_group_rccl_ranks.argtypes = [vp]
    L.ta_group_set_option.argtypes = [vp, ctypes.c_char_p, i64]
    L.ta_group_stage_alloc.argtypes = [vp, i64, i64, ci, ci, ci, ctypes.POINTER(vp)]
    L.ta_group_stage_commit.argtypes = [vp, i64, i64]
    L.ta_group_stage_free.argtypes = [vp]
    L.ta_group_stage_alloc_device.argtypes = [vp, i64, i64, ci, ci]
    L.ta_group_stage_synth.argtypes = [vp, ci, ctypes.c_uint64, i64, i64]
    L.ta_group_vacf_fft.argtypes = [vp, vp, vp]
    L.ta_group_vacf_direct.argtypes = [vp, vp, vp]
    L.ta_group_helfand_msd.argtypes = [vp, vp, dbl, vp, vp]
    for name in EXPORTS:
        if name not in ("ta_last_error", "ta_group_last_error", "ta_group_reduce_kind", "ta_group_reduce_note"):
            getattr(L, name).restype = ci
    _lib = L
    return L


def device_count():
    return int(lib().ta_device_count())


def fft_plan_info(n_frames):
    m, nt, ns = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
    rc = lib().ta_fft_plan_info(int(n_frames), ctypes.byref(m), ctypes.byref(nt), ctypes.byref(ns))
    if rc != 0:
        return None
    return {"M": m.value, "n_threads": nt.value, "n_stages": ns.value}


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(None)


def frame_source(arr):
    """(pointer, dtype code, ld_row) of a frame's (n_atoms, n_coord) array if ta_stage_frame can read it in
    place -- float32 / float64, unit stride along the coordinates, a non-negative whole-element row stride --
    else None (the caller then stages through the NumPy view of the slab)."""
    if not isinstance(arr, np.ndarray) or arr.ndim != 2 or arr.dtype not in (np.float32, np.float64):
        return None
    isz = arr.dtype.itemsize
    if arr.shape[1] > 1 and arr.strides[1] != isz:
        return None
    if arr.strides[0] < isz * arr.shape[1] or arr.strides[0] % isz:
        return None
    return arr.ctypes.data, (TA_F32 if arr.dtype == np.float32 else TA_F64), arr.strides[0] // isz


def atom_rows(ix):
    """(atom_lo, index array or None, n) for ta_stage_frame from an AtomGroup's atom indices: a group of
    consecutive atoms is a block (no index array), anything else is gathered by index."""
    ix = np.ascontiguousarray(ix, dtype=np.int64)
    n = int(ix.size)
    if n == 0:
        return 0, None, 0
    if int(ix[-1]) - int(ix[0]) + 1 == n and (n == 1 or bool(np.all(np.diff(ix) == 1))):
        return int(ix[0]), None, n
    return 0, ix, n


class _PinnedBlock:
    """Owner of one ta_host_alloc block: freed when the last NumPy view of it is gone."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                lib().ta_host_free(ctypes.c_void_p(self.ptr))
                self.ptr = 0
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64, device=-1):
    """np.empty in page-locked host memory (ta_host_alloc_on): the home of results.vacf_by_particle /
    results.visc_by_particle, so that the device->host copy runs at the link's rate the first
    time.  The memory lives as long as the array or any view of it.  `device`: the GPU the calling
    thread is bound to before allocating (-1: its current one).  Raises TAError when the
    allocation fails; `result_empty` is the variant that falls back to pageable memory."""
    shape = tuple(int(x) for x in np.atleast_1d(shape))
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    p = ctypes.c_void_p()
    rc = lib().ta_host_alloc_on(int(device), nbytes, ctypes.byref(p))
    if rc != 0:
        raise TAError(rc, lib().ta_last_error(None).decode())
    buf = (ctypes.c_char * max(nbytes, 1)).from_address(p.value)
    buf._owner = _PinnedBlock(p.value)  # the ctypes array is the NumPy array's base
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)


def pinned_results_enabled():
    """$TA_AMD_PINNED_RESULTS=0 keeps result arrays in pageable memory (np.empty, as the reference's
    np.zeros): for hosts whose RLIMIT_MEMLOCK / cgroup leaves no room for 8 GB more of pinned pages
    beside the staging slabs."""
    return os.environ.get("TA_AMD_PINNED_RESULTS", "1") != "0"


def result_empty(shape, device=-1):
    """Home of a by-particle result: pinned when possible, else -- with a warning -- a pageable
    np.empty (the device->host copy is then slower the first time, the values are the same).  The
    analysis must not be lost in _conclude, after every frame has been read, because page-locking
    failed (hipHostMalloc under a memlock limit)."""
    if not pinned_results_enabled():
        return np.empty(tuple(int(x) for x in np.atleast_1d(shape)), dtype=np.float64)
    try:
        return pinned_empty(shape, device=device)
    except TAError as e:
        import warnings

        warnings.warn(f"page-locked result array unavailable ({e}); using pageable memory "
                      "(slower device->host copy, same values)", RuntimeWarning, stacklevel=2)
        return np.empty(tuple(int(x) for x in np.atleast_1d(shape)), dtype=np.float64)


class PinnedResult:
    """The home of a by-particle result, page-locked on a helper thread while the frames are staged
    (page-locking 8 GB takes about as long as copying them); `get()` joins.  The helper thread is
    bound to the analysis' GPU; a failed pinned allocation degrades to pageable memory with a
    warning (result_empty)."""

    def __init__(self, shape, device=-1):
        import threading

        self._arr, self._err = None, None

        def work():
            try:
                self._arr = result_empty(shape, device=device)
            except Exception as e:  # surfaced by get()
                self._err = e

        self._thread = threading.Thread(target=work, daemon=True)
        self._thread.start()

    def get(self):
        self._thread.join()
        if self._err is not None:
            raise self._err
        return self._arr


DEVICE_CPU = -1  # ta_hip.h: TA_DEVICE_CPU


def device_index(device):
    """A GPU index, or DEVICE_CPU for "cpu" / -1: the opt-in CPU backend behind the same C symbols
    (csrc/cpu_backend.cpp).  Nothing picks it on the caller's behalf."""
    if isinstance(device, str):
        if device.strip().lower() == "cpu":
            return DEVICE_CPU
        return int(device)
    return int(device)


class _PlainHome:
    """result_home of a CPU context: an ordinary array (there is no device to copy from)"""

    def __init__(self, shape):
        self._shape = tuple(int(x) for x in np.atleast_1d(shape))

    def get(self):
        return np.empty(self._shape, dtype=np.float64)


class Context:
    """One ta_ctx: owns a stream, plan tables, workspaces and the staged slabs.  `device`: a GPU index, or
    "cpu" / DEVICE_CPU for the opt-in CPU backend (host slabs, OpenMP; the device-pointer calls are unsupported)."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p(None)
        L = lib()
        device = device_index(device)
        rc = L.ta_ctx_create(device, ctypes.byref(self._h))
        if rc != 0:
            raise TAError(rc, L.ta_last_error(None).decode())
        self.device = device
        self.is_cpu = device == DEVICE_CPU
        self._slabs = []

    # -- plumbing -------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise TAError(rc, lib().ta_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._drop_views()
            if not getattr(self, "_borrowed", False):  # a group member's context belongs to its group
                lib().ta_ctx_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self._check(lib().ta_set_option(self._h, key.encode(), int(value)))

    def result_home(self, shape):
        """Start page-locking the by-particle result array of `shape`; `.get()` returns it."""
        if getattr(self, "is_cpu", False):
            return _PlainHome(shape)
        return PinnedResult(shape, device=self.device)

    # -- staging --------------------------------------------------------
    def stage_alloc(self, n_frames, n_atoms, dim, n_slabs=1, dtype=np.float64):
        """Pinned host slabs (n_frames, n_atoms, dim) as NumPy views + device twins."""
        code = TA_F64 if np.dtype(dtype) == np.float64 else TA_F32
        ptrs = (ctypes.c_void_p * n_slabs)()
        self._drop_views()
        self._check(lib().ta_stage_alloc(self._h, n_frames, n_atoms, dim, code, n_slabs, ptrs))
        ct = ctypes.c_double if code == TA_F64 else ctypes.c_float
        n = int(n_frames) * int(n_atoms) * int(dim)
        out = []
        for p in ptrs:
            buf = (ct * n).from_address(p)
            out.append(np.frombuffer(buf, dtype=dtype).reshape(n_frames, n_atoms, dim))
        self._slabs = out
        self.shape = (int(n_frames), int(n_atoms), int(dim))
        return out

    def stage_commit(self, frame_lo, frame_hi):
        self._check(lib().ta_stage_commit(self._h, int(frame_lo), int(frame_hi)))

    def stage_frame(self, slab, frame, source, cols, rows):
        """slab[frame] = source[rows][:, cols] natively (ta_stage_frame).  source: frame_source(...) of the
        Timestep's array; cols: the dim_type's column list (an arithmetic progression); rows: atom_rows(...)."""
        ptr, code, ld = source
        lo, index, n = rows
        step = cols[1] - cols[0] if len(cols) > 1 else 1
        self._check(lib().ta_stage_frame(self._h, int(slab), int(frame), ctypes.c_void_p(ptr), code, int(ld), int(cols[0]),
                                         int(step), len(cols), int(lo), _ptr(index), int(n)))

    def stage_alloc_device(self, n_frames, n_atoms, dim, n_slabs=1):
        """Device slabs only (pair-major), for data that is already on the GPU."""
        self._drop_views()
        self._check(lib().ta_stage_alloc_device(self._h, n_frames, n_atoms, dim, n_slabs))
        self.shape = (int(n_frames), int(n_atoms), int(dim))

    def stage_commit_dev(self, slab, d_src, ld_row, frame_lo, frame_hi, dtype=np.float64, stream=0):
        code = TA_F64 if np.dtype(dtype) == np.float64 else TA_F32
        self._check(lib().ta_stage_commit_dev(self._h, slab, d_src, code, int(ld_row), int(frame_lo),
                                              int(frame_hi), stream or None))

    def stage_synth(self, slab, seed, col_offset, n_cols_total, stream=0):
        self._check(lib().ta_stage_synth(self._h, slab, int(seed), int(col_offset), int(n_cols_total),
                                         stream or None))

    def stage_read_dev(self, slab, d_dst, ld_row, stream=0):
        self._check(lib().ta_stage_read_dev(self._h, slab, d_dst, int(ld_row), stream or None))

    def stage_device(self, slab):
        """(device pointer, rows per column pair, number of pairs) of the pair-major slab."""
        p, pitch, n_pairs = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        self._check(lib().ta_stage_device(self._h, slab, ctypes.byref(p), ctypes.byref(pitch),
                                          ctypes.byref(n_pairs)))
        return p.value, pitch.value, n_pairs.value

    def trim(self):
        self._check(lib().ta_trim(self._h))

    def _drop_views(self):
        # the pinned memory behind the NumPy views goes away with the slabs: writes through a
        # stale view must not land in freed memory silently
        for a in self._slabs:
            try:
                a.setflags(write=False)
            except Exception:
                pass
        self._slabs = []

    def stage_free(self):
        self._drop_views()
        self._check(lib().ta_stage_free(self._h))

    # -- host-facing compute -------------------------------------------
    def _host(self, fn, by_particle, *extra, out=None):
        """by_particle: False, True (a pinned (n_frames, n_atoms) array is allocated here) or
        `out` = the caller's (n_frames, n_atoms) float64 C-contiguous array (pinned_empty)."""
        T, A, _ = getattr(self, "shape", None) or (1, 1, 1)  # unstaged: the library reports it
        ts = np.empty(T, dtype=np.float64)
        bp = None
        if out is not None:
            if out.shape != (T, A) or out.dtype != np.float64 or not out.flags.c_contiguous:
                raise ValueError("out must be a C-contiguous float64 array of shape (n_frames, n_atoms)")
            bp = out
        elif by_particle:
            bp = np.empty((T, A), dtype=np.float64) if getattr(self, "is_cpu", False) else result_empty((T, A), device=self.device)
        self._check(fn(self._h, *extra, _ptr(ts), _ptr(bp)))
        return ts, bp

    def vacf_fft(self, by_particle=False, out=None):
        return self._host(lib().ta_vacf_fft, by_particle, out=out)

    def vacf_direct(self, by_particle=False, out=None):
        return self._host(lib().ta_vacf_direct, by_particle, out=out)

    def helfand_msd(self, masses, scale, by_particle=False, out=None):
        m = np.ascontiguousarray(masses, dtype=np.float64)
        return self._host(lib().ta_helfand_msd, by_particle, _ptr(m), ctypes.c_double(scale), out=out)

    # -- device-pointer compute (asynchronous) --------------------------
    def vacf_fft_dev(self, d_vel, n_frames, n_atoms, dim, ld_row, d_lagsum, d_bp=0, ld_bp=0, stream=0):
        self._check(lib().ta_vacf_fft_dev(self._h, d_vel, n_frames, n_atoms, dim, ld_row, d_lagsum,
                                          d_bp or None, ld_bp, stream or None))

    def vacf_direct_dev(self, d_vel, n_frames, n_atoms, dim, ld_row, d_lagsum, d_bp=0, ld_bp=0, stream=0):
        self._check(lib().ta_vacf_direct_dev(self._h, d_vel, n_frames, n_atoms, dim, ld_row, d_lagsum,
                                             d_bp or None, ld_bp, stream or None))

    def helfand_msd_dev(self, d_vel, d_pos, d_masses, n_frames, n_atoms, dim, ld_row, scale, d_lagsum,
                        d_bp=0, ld_bp=0, stream=0):
        self._check(lib().ta_helfand_msd_dev(self._h, d_vel, d_pos, d_masses, n_frames, n_atoms, dim,
                                             ld_row, scale, d_lagsum, d_bp or None, ld_bp,
                                             stream or None))

    # -- compute on the staged slabs, device outputs (asynchronous) ------
    def vacf_fft_staged(self, d_lagsum, d_bp=0, ld_bp=0, stream=0):
        self._check(lib().ta_vacf_fft_staged(self._h, d_lagsum, d_bp or None, ld_bp, stream or None))

    def vacf_direct_staged(self, d_lagsum, d_bp=0, ld_bp=0, stream=0):
        self._check(lib().ta_vacf_direct_staged(self._h, d_lagsum, d_bp or None, ld_bp, stream or None))

    def helfand_msd_staged(self, d_masses, scale, d_lagsum, d_bp=0, ld_bp=0, stream=0):
        self._check(lib().ta_helfand_msd_staged(self._h, d_masses, scale, d_lagsum, d_bp or None, ld_bp,
                                                stream or None))

    def timing_history(self, max_n=64):
        """[(total_ms, main_kernel_ms)] of the last compute calls, oldest first."""
        n = ctypes.c_int()
        t = (ctypes.c_float * max_n)()
        m = (ctypes.c_float * max_n)()
        self._check(lib().ta_timing_history(self._h, max_n, t, m, ctypes.byref(n)))
        return [(t[i], m[i]) for i in range(n.value)]

    def kernel_timeline(self, max_n=32):
        """[(kernel name, ms)] of the last compute call (needs set_option("timeline", 1))."""
        n = ctypes.c_int()
        names = (ctypes.c_char_p * max_n)()
        ms = (ctypes.c_float * max_n)()
        self._check(lib().ta_kernel_timeline(self._h, max_n, names, ms, ctypes.byref(n)))
        return [(names[i].decode(), ms[i]) for i in range(n.value)]

    def clock_probe(self, n_launches):
        """{"mhz", "cycles_per_unit_pass", "ms_per_launch"} of the stamped lag-sum forward kernel
        launched n_launches times back to back on the staged slab (ta_clock_probe)."""
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._check(lib().ta_clock_probe(self._h, int(n_launches), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"mhz": a.value, "cycles_per_unit_pass": b.value, "ms_per_launch": c.value}

    def last_timing(self):
        t, m = ctypes.c_float(), ctypes.c_float()
        self._check(lib().ta_last_timing(self._h, ctypes.byref(t), ctypes.byref(m)))
        return t.value, m.value


class Group:
    """Several GPUs behind one object (ta_group): ONE frame loop fills every GPU's column block of
    the slab, a compute call fans out, reduces the lag sums once inside the library (RCCL for
    distinct devices; `reduce_kind` says what ran) and copies by-particle blocks into the column
    ranges of one host array.  Same methods as `Context` where the analysis classes use them;
    `stage_alloc` returns, per slab, the list of per-member views, and `shards` the members' atom
    ranges [(lo, hi), ...] (an empty range: more devices than atoms, its view is None)."""

    def __init__(self, devices):
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one GPU")
        self._h = ctypes.c_void_p(None)
        L = lib()
        ids = (ctypes.c_int * len(devices))(*devices)
        rc = L.ta_group_create(ids, len(devices), ctypes.byref(self._h))
        if rc != 0:
            raise TAError(rc, L.ta_group_last_error(None).decode())
        self.devices = devices
        self.device = devices[0]
        self.shards = []
        self._slabs = []

    def _check(self, rc):
        if rc != 0:
            raise TAError(rc, lib().ta_group_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._drop_views()
            for ref in self.__dict__.get("_member_views", []):
                c = ref()
                if c is not None:  # a borrowed member context must not outlive the group's handle
                    c._h = ctypes.c_void_p(None)
            self._member_views = []
            lib().ta_group_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _drop_views(self):
        for per_member in self._slabs:
            for a in per_member:
                if a is not None:
                    try:
                        a.setflags(write=False)
                    except Exception:
                        pass
        self._slabs = []

    def set_option(self, key, value):
        self._check(lib().ta_group_set_option(self._h, key.encode(), int(value)))

    @property
    def reduce_kind(self):
        return lib().ta_group_reduce_kind(self._h).decode()

    @property
    def reduce_note(self):
        """why an automatic reduce fell back from RCCL to peer copies ("" when it did not)"""
        return lib().ta_group_reduce_note(self._h).decode()

    @property
    def rccl_ranks(self):
        """ranks of the communicator the last RCCL reduce ran on (ncclCommCount)"""
        return int(lib().ta_group_rccl_ranks(self._h))

    def member_context(self, i):
        """Member i's context as a non-owning `Context` (timing history, options of one member)."""
        h, dev = ctypes.c_void_p(), ctypes.c_int()
        self._check(lib().ta_group_member(self._h, int(i), ctypes.byref(h), ctypes.byref(dev)))
        c = Context.__new__(Context)
        c._h, c.device, c._slabs, c._borrowed, c.is_cpu = h, dev.value, [], True, False
        c._group = self  # the member's context lives as long as its group: the view keeps the group alive ...
        import weakref

        self.__dict__.setdefault("_member_views", []).append(weakref.ref(c))  # ... and close() invalidates the view
        return c

    def shard(self, n_atoms, i):
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        self._check(lib().ta_group_shard(self._h, int(n_atoms), int(i), ctypes.byref(lo), ctypes.byref(hi)))
        return lo.value, hi.value

    def result_home(self, shape):
        return PinnedResult(shape, device=self.device)

    def stage_alloc(self, n_frames, n_atoms, dim, n_slabs=1, dtype=np.float64):
        """-> [slab][member] NumPy views (n_frames, hi_i - lo_i, dim) of the members' pinned slabs."""
        code = TA_F64 if np.dtype(dtype) == np.float64 else TA_F32
        n_dev = len(self.devices)
        ptrs = (ctypes.c_void_p * (n_dev * n_slabs))()
        self._drop_views()
        self._check(lib().ta_group_stage_alloc(self._h, n_frames, n_atoms, dim, code, n_slabs, ptrs))
        ct = ctypes.c_double if code == TA_F64 else ctypes.c_float
        self.shards = [self.shard(n_atoms, i) for i in range(n_dev)]
        out = []
        for s in range(n_slabs):
            views = []
            for i, (lo, hi) in enumerate(self.shards):
                p = ptrs[i * n_slabs + s]
                if hi == lo or not p:
                    views.append(None)
                    continue
                n = int(n_frames) * (hi - lo) * int(dim)
                buf = (ct * n).from_address(p)
                views.append(np.frombuffer(buf, dtype=dtype).reshape(n_frames, hi - lo, dim))
            out.append(views)
        self._slabs = out
        self.shape = (int(n_frames), int(n_atoms), int(dim))
        return out

    def stage_commit(self, frame_lo, frame_hi):
        self._check(lib().ta_group_stage_commit(self._h, int(frame_lo), int(frame_hi)))

    def stage_frame(self, slab, frame, source, cols, rows):
        """every member's slab[frame] = its atoms of source[rows][:, cols] (ta_group_stage_frame)"""
        ptr, code, ld = source
        lo, index, n = rows
        step = cols[1] - cols[0] if len(cols) > 1 else 1
        self._check(lib().ta_group_stage_frame(self._h, int(slab), int(frame), ctypes.c_void_p(ptr), code, int(ld),
                                               int(cols[0]), int(step), len(cols), int(lo), _ptr(index), int(n)))

    def stage_alloc_device(self, n_frames, n_atoms, dim, n_slabs=1):
        self._drop_views()
        self._check(lib().ta_group_stage_alloc_device(self._h, n_frames, n_atoms, dim, n_slabs))
        self.shards = [self.shard(n_atoms, i) for i in range(len(self.devices))]
        self.shape = (int(n_frames), int(n_atoms), int(dim))

    def stage_synth(self, slab, seed, col_offset, n_cols_total):
        self._check(lib().ta_group_stage_synth(self._h, slab, int(seed), int(col_offset), int(n_cols_total)))

    def stage_free(self):
        self._drop_views()
        self._check(lib().ta_group_stage_free(self._h))

    def _host(self, fn, by_particle, *extra, out=None):
        T, A, _ = getattr(self, "shape", None) or (1, 1, 1)
        ts = np.empty(T, dtype=np.float64)
        bp = None
        if out is not None:
            if out.shape != (T, A) or out.dtype != np.float64 or not out.flags.c_contiguous:
                raise ValueError("out must be a C-contiguous float64 array of shape (n_frames, n_atoms)")
            bp = out
        elif by_particle:
            bp = result_empty((T, A), device=self.device)
        self._check(fn(self._h, *extra, _ptr(ts), _ptr(bp)))
        return ts, bp

    def vacf_fft(self, by_particle=False, out=None):
        return self._host(lib().ta_group_vacf_fft, by_particle, out=out)

    def vacf_direct(self, by_particle=False, out=None):
        return self._host(lib().ta_group_vacf_direct, by_particle, out=out)

    def helfand_msd(self, masses, scale, by_particle=False, out=None):
        m = np.ascontiguousarray(masses, dtype=np.float64)
        return self._host(lib().ta_group_helfand_msd, by_particle, _ptr(m), ctypes.c_double(scale), out=out)

