"""Velocity autocorrelation function on MI355X — drop-in for
``transport_analysis.velocityautocorr.VelocityAutocorr``.

Same constructor, ``run()``, results and helper methods as the reference
(/root/reference/transport_analysis/velocityautocorr.py:72-422); the arithmetic
of ``_conclude_fft`` (:208-215) and ``_conclude_simple`` (:217-238) runs in
hand-written HIP kernels behind the C-ABI of ``include/ta_hip.h``.  There is no
CPU fallback.
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib
from ._base import (stage_columns, AnalysisBase, NoDataError, UpdatingAtomGroup, native_rows, parse_dim_type,
                    stage_frame_native)

#: frames staged on the host before an asynchronous host->device copy is queued
_COMMIT_BYTES = 32 << 20


class VelocityAutocorr(AnalysisBase):
    r"""Velocity autocorrelation function (VACF) of an AtomGroup.

    Parameters
    ----------
    atomgroup : AtomGroup
        Particles to analyse.  An ``UpdatingAtomGroup`` raises ``TypeError``.
    dim_type : {'xyz', 'xy', 'yz', 'xz', 'x', 'y', 'z'}
        Dimensions included in the VACF (case-insensitive, order-sensitive).
    fft : bool
        ``True``: FFT algorithm (reference: ``tidynamics.acf`` per atom);
        ``False``: direct "windowed" algorithm.  Both give the same quantity.
    by_particle : bool, keyword-only, default True
        ``True`` materialises ``results.vacf_by_particle`` (n_frames, n_atoms)
        as the reference does (``fft=False``: on the FP64 matrix cores, a particle's
        column in a per-wave LDS ring).  ``False`` computes only
        ``results.timeseries`` (``fft=True``: power spectra are summed
        over atoms on the GPU before the single inverse transform, 2.4x faster;
        ``fft=False``: the lag sums are the diagonal sums of the frames' Gram matrix,
        on the FP64 matrix cores, no faster than with the array) and
        ``results.vacf_by_particle`` is ``None``.
    device : int or "cpu", keyword-only
        GPU index (default: ``$TA_AMD_DEVICE`` or 0; with ``distributed=True``:
        ``$TA_AMD_DEVICE``, else ``$LOCAL_RANK``, else torch's current device).  ``"cpu"`` (or
        ``$TA_AMD_DEVICE=cpu``) asks for the library's opt-in CPU backend -- C++/OpenMP behind the
        same C symbols (``csrc/cpu_backend.cpp``); it is never selected on the caller's behalf: without
        a GPU every other value fails loudly.
    stage_dtype : numpy dtype, keyword-only
        Element type of the pinned staging slab.  Default: the dtype MDAnalysis hands the
        velocities out in (float32) -- lossless, half the PCIe bytes of the reference's
        float64 slab; the device slab and all arithmetic are float64 either way.
    device_float32 : bool or None, keyword-only, default None
        Keep float32-staged frames as float32 in the DEVICE slab as well (half the device memory:
        12 GB instead of 24 GB at 10000 frames x 100000 atoms; the FFT kernels read the 8-byte rows
        and widen them exactly, 3-14 % faster than from float64 slabs; same values in, results
        equal to float64 slabs within 1e-15 of the scale).  ``None``: on when the staging slab is
        float32, ``fft=True`` and the trajectory has 513 ... 10240 frames (the plans that read
        float32 rows; any other evaluation of float32 device slabs first widens them).
    devices : sequence of int, keyword-only
        Several GPUs from ONE process and ONE pass over the trajectory: the
        atoms are split into contiguous blocks, one per GPU; every frame's columns go straight
        into each GPU's pinned slab, the kernels run on all GPUs at once, the lag sums are reduced
        once inside the library (RCCL) and ``results.vacf_by_particle`` is ONE
        ``(n_frames, n_particles)`` array whose column ranges the GPUs fill.  The script is the
        reference's, unchanged.  Exclusive with ``distributed``.
    distributed : bool, keyword-only, default False
        One process per GPU under ``torch.distributed`` (e.g. ``torchrun``): every rank runs
        the same script on the same AtomGroup, stages and correlates only its contiguous block
        of atoms (and asks the trajectory for that block's velocities only) and ONE all-reduce of
        the lag sums gives ``results.timeseries`` (the mean over ALL atoms) on every rank.
        ``results.vacf_by_particle`` then holds this rank's atoms only,
        ``results.particle_range = (lo, hi)``.

    Attributes
    ----------
    results.timeseries : (n_frames,) float64 — VACF averaged over particles,
        lag index k = 0..n_frames-1, units (Å/ps)^2.
    results.vacf_by_particle : (n_frames, n_particles) float64 or None
    dim_fac, n_frames, n_particles, times, frames, start, stop, step
    """

    def __init__(self, atomgroup, dim_type="xyz", fft=True, **kwargs):
        self._want_by_particle = bool(kwargs.pop("by_particle", True))
        self._distributed = bool(kwargs.pop("distributed", False))
        self._stage_dtype = kwargs.pop("stage_dtype", None)
        self._device_f32 = kwargs.pop("device_float32", None)
        devices = kwargs.pop("devices", None)
        self._devices = None if devices is None else [int(d) for d in devices]
        if self._devices is not None and self._distributed:
            raise ValueError("devices=[...] (one process, several GPUs) and distributed=True "
                             "(one process per GPU) are exclusive")
        device = kwargs.pop("device", None)
        if device is None and self._devices:
            device = self._devices[0]
        if device is None:
            if self._distributed:  # one process per GPU: this rank's own device
                from .dist import default_device

                device = default_device()
            else:
                device = os.environ.get("TA_AMD_DEVICE", 0)
        self._device = _lib.device_index(device)
        super().__init__(atomgroup.universe.trajectory, **kwargs)

        if isinstance(atomgroup, UpdatingAtomGroup):
            raise TypeError("UpdatingAtomGroups are not valid for VACF computation")

        self.dim_type = dim_type.lower()
        self._dim, self.dim_fac = parse_dim_type(self.dim_type)
        self.fft = fft

        self.atomgroup = atomgroup
        self.n_particles = len(self.atomgroup)
        self._run_called = False
        self._ctx = None

    _parse_dim_type = staticmethod(parse_dim_type)

    # MDAnalysis >= 2.8 parallel-analysis protocol: frames are staged into ONE device slab per
    # analysis object and every lag couples all frames, so a frame-split backend cannot apply;
    # the data-parallel axis of this path is atoms (distributed=True), not frames.
    _analysis_algorithm_is_parallelizable = False

    @classmethod
    def get_supported_backends(cls):
        return ("serial",)

    def _pick_stage_dtype(self):
        """float32 when the trajectory hands out float32 (MDAnalysis always does); the
        reference upcasts into a float64 slab (:150-152,192-194), which the device slab is."""
        if self._stage_dtype is not None:
            return np.dtype(self._stage_dtype)
        try:
            dt = np.asarray(self.atomgroup.velocities).dtype
        except Exception:  # no velocities: _single_frame raises NoDataError, as the reference does
            return np.dtype(np.float64)
        return np.dtype(np.float32) if dt == np.float32 else np.dtype(np.float64)

    # ------------------------------------------------------------ hooks
    def _prepare(self):
        """Pinned host slab + device slab instead of ``np.zeros`` (:142-153)."""
        if self._ctx is None:
            self._ctx = _lib.Group(self._devices) if self._devices is not None else _lib.Context(self._device)
        self._lo, self._hi = 0, self.n_particles
        self._source = self.atomgroup  # whose velocities a frame is read from
        if self._distributed:
            from .dist import shard_of_this_rank

            _, _, self._lo, self._hi = shard_of_this_rank(self.n_particles)
            self.results.particle_range = (self._lo, self._hi)
            # this rank's block only: the trajectory gathers hi - lo atoms per frame, not all of them
            self._source = self.atomgroup[self._lo:self._hi]
        self._n_local = self._hi - self._lo
        dtype = self._pick_stage_dtype()
        dev32 = self._device_f32
        if dev32 is None:  # float32 staging stays float32 on the device where the FFT kernels read it as it is
            dev32 = dtype == np.float32 and bool(self.fft) and 512 < self.n_frames <= 10240
        self._ctx.set_option("stage_device_f32", int(bool(dev32) and dtype == np.float32))
        if self._devices is not None:
            # one pinned slab per GPU, each holding that GPU's column block; filled in ONE frame loop
            (views,) = self._ctx.stage_alloc(self.n_frames, self.n_particles, self.dim_fac, n_slabs=1, dtype=dtype)
            self._velocities = views
            self._targets = [(v, lo, hi) for v, (lo, hi) in zip(views, self._ctx.shards) if hi > lo]
            self.results.device_ranges = list(self._ctx.shards)
        else:
            (self._velocities,) = self._ctx.stage_alloc(
                self.n_frames, max(self._n_local, 1), self.dim_fac, n_slabs=1, dtype=dtype)
            # (columns of the source group: the distributed source is the block itself)
            self._targets = [(self._velocities, 0, self._n_local)] if self._distributed else \
                [(self._velocities, self._lo, self._hi)]
        # the per-frame fill reads the Timestep's own array natively (ta_stage_frame) where it can
        self._rows = native_rows(self._source) if self._n_local else None
        frame_bytes = max(1, self._n_local * self.dim_fac * dtype.itemsize)
        self._commit_every = max(1, _COMMIT_BYTES // frame_bytes)
        self._committed = 0
        self.results.vacf_by_particle = None
        # the (n_frames, n_particles) result array (:145-147) lives in pinned host memory, page-locked
        # on a helper thread while the frames are staged
        self._bp_home = None
        if self._want_by_particle and self._n_local and not self._device_reduce():
            self._bp_home = self._ctx.result_home((self.n_frames, self._n_local))
        # results.timeseries is not set here (reference: :153)

    def _single_frame(self):
        """Stage one frame of the selected velocity columns (:178-194)."""
        if not self._ts.has_velocities:
            raise NoDataError("VACF computation requires velocities in the trajectory")
        i = self._frame_index
        if self._n_local and not stage_frame_native(self._ctx, 0, i, self._ts, "velocities", self._dim, self._rows):
            vel = np.asarray(self._source.velocities)
            for view, lo, hi in self._targets:
                stage_columns(view[i], vel, lo, hi, self._dim)
        if i + 1 - self._committed >= self._commit_every:
            self._ctx.stage_commit(self._committed, i + 1)
            self._committed = i + 1

    def _conclude(self):
        if self._committed < self.n_frames:
            self._ctx.stage_commit(self._committed, self.n_frames)
            self._committed = self.n_frames
        if self.fft:
            self._conclude_fft()
        else:
            self._conclude_simple()

    def _conclude_fft(self):
        self._compute("fft")

    def _conclude_simple(self):
        self._compute("direct")

    def _device_reduce(self):
        if not self._distributed:
            return False
        from .dist import uses_device_reduce

        return uses_device_reduce()

    def _compute(self, which):
        if self._distributed:
            from .dist import staged_timeseries_on_device

            if self._device_reduce():  # RCCL: the lag sums stay on the GPU through the reduce
                ts, bp = staged_timeseries_on_device(self._ctx, which, self.n_frames, self._n_local,
                                                     self.n_particles, self._device,
                                                     by_particle=self._want_by_particle)
                self.results.vacf_by_particle = bp
                self.results.timeseries = ts
                self._run_called = True
                return
        fn = self._ctx.vacf_fft if which == "fft" else self._ctx.vacf_direct
        home = self._bp_home.get() if self._bp_home is not None else None
        self._bp_home = None
        ts, bp = fn(by_particle=self._want_by_particle, out=home)
        self._store(ts, bp)

    def _store(self, ts, bp):
        if self._distributed:
            from .dist import allreduce_mean_over_atoms

            if self._n_local == 0:  # more ranks than atoms: this rank contributes nothing
                ts, bp = np.zeros(self.n_frames), (None if bp is None else bp[:, :0])
            ts = allreduce_mean_over_atoms(ts, self._n_local, self.n_particles, self._device)
        self.results.vacf_by_particle = bp
        self.results.timeseries = ts
        self._run_called = True

    # --------------------------------------------- post-processing (host)
    def _window(self, start, stop, step):
        stop = self.n_frames if stop == 0 else stop
        sl = slice(start, stop, step)
        return self.times[sl], self.results.timeseries[sl]

    def plot_vacf(self, start=0, stop=0, step=1, xlabel="Time (ps)",
                  ylabel="Velocity Autocorrelation Function (Å^2 / ps^2)"):
        """Plot the VACF (:240-285).  Returns the list of Line2D from ``Axes.plot``."""
        if not self._run_called:
            raise RuntimeError("Analysis must be run prior to plotting")
        import matplotlib.pyplot as plt

        t, y = self._window(start, stop, step)
        _, ax = plt.subplots()
        ax.set_xlabel(xlabel)
        ax.set_ylabel(ylabel)
        return ax.plot(t, y)

    def self_diffusivity_gk(self, start=0, stop=0, step=1):
        """Green-Kubo self-diffusivity, trapezoid rule, divided by dim_fac (:287-322)."""
        if not self._run_called:
            raise RuntimeError("Analysis must be run prior to computing self-diffusivity")
        from scipy import integrate

        t, y = self._window(start, stop, step)
        return integrate.trapezoid(y, t) / self.dim_fac

    def self_diffusivity_gk_odd(self, start=0, stop=0, step=1):
        """Green-Kubo self-diffusivity, Simpson rule (:324-360)."""
        if not self._run_called:
            raise RuntimeError("Analysis must be run prior to computing self-diffusivity")
        from scipy import integrate

        t, y = self._window(start, stop, step)
        return integrate.simpson(y=y, x=t) / self.dim_fac

    def plot_running_integral(self, start=0, stop=0, step=1, initial=0, xlabel="Time (ps)",
                              ylabel="Running Integral of the VACF (Å^2 / ps)"):
        """Plot the cumulative trapezoid integral of the VACF / dim_fac (:362-422)."""
        if not self._run_called:
            raise RuntimeError("Analysis must be run prior to plotting")
        import matplotlib.pyplot as plt
        from scipy import integrate

        t, y = self._window(start, stop, step)
        running = integrate.cumulative_trapezoid(y, t, initial=initial) / self.dim_fac
        _, ax = plt.subplots()
        ax.set_xlabel(xlabel)
        ax.set_ylabel(ylabel)
        return ax.plot(t, running)
