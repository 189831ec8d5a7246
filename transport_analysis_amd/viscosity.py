"""Einstein-Helfand viscosity function on MI355X — drop-in for
``transport_analysis.viscosity.ViscosityHelfand``
(/root/reference/transport_analysis/viscosity.py:26-272).  The windowed
mean-squared-difference accumulator of ``_conclude`` (:201-233) runs in a
hand-written HIP kernel; the fit (:235-245) stays on the host.  No CPU fallback.
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib
from ._base import (stage_columns, BOLTZMANN, AnalysisBase, NoDataError, UpdatingAtomGroup, native_rows,
                    parse_dim_type, stage_frame_native)

_COMMIT_BYTES = 32 << 20


class ViscosityHelfand(AnalysisBase):
    r"""Viscosity function via the Einstein-Helfand method.

    Parameters
    ----------
    atomgroup : AtomGroup
    temp_avg : float — average temperature (K), default 300.
    dim_type : {'xyz', 'xy', 'yz', 'xz', 'x', 'y', 'z'}
    linear_fit_window : (int, int) or None — lag-index window for the slope fit.
    by_particle : bool, keyword-only, default True — materialise
        ``results.visc_by_particle``; ``False`` computes the timeseries only.  Both run on the
        matrix cores (FP64; ``float32=True``: FP32), every lag to the path's accuracy.
    device : int or "cpu", keyword-only — GPU index (default ``$TA_AMD_DEVICE`` or 0); ``"cpu"`` asks
        for the library's opt-in CPU backend (C++/OpenMP behind the same C symbols; nothing selects it
        on the caller's behalf).
    distributed : bool, keyword-only, default False — one process per GPU under
        ``torch.distributed``: each rank handles its contiguous block of atoms, one all-reduce of
        the lag sums gives ``results.timeseries`` on every rank; ``results.visc_by_particle``
        holds this rank's atoms only (``results.particle_range``).
    float32 : bool, keyword-only, default False — form the mass-weighted
        velocity-position products in float64, then evaluate the squared differences and
        their block sums in float32 (accumulated into float64): 2e-6 of the series' scale
        instead of 1e-10 relative, 1.9x the throughput.  The products are rounded to float32
        ONCE before differences are formed, so the shortest lags of a smooth, trending series see
        that rounding (6e-8 of |P|) against a small difference: on the reference's step trajectory
        1e-5 relative at lag 1, 2e-6 at lags 2-4, 3e-7 from lag 5 on; a fit window that starts at
        lag 5 or later is unaffected.  (The CPU backend computes this option in float64.)

    fft : bool, keyword-only, default False — an extension (the reference has only the
        O(n_frames^2) loop): evaluate the mean squared differences in
        O(n_frames log n_frames) as ``S1(k) - 2 S2(k)`` (``S2`` = FFT autocorrelation of the
        products ``m v x``, ``S1`` from prefix sums), with or without the per-particle array, for
        n_frames <= 163840 (longer trajectories fall back to the direct correlator).  Accurate to ~1e-15 of the series' scale;
        lags whose mean squared difference is far below the squared products themselves lose
        relative accuracy by that ratio.

    Attributes
    ----------
    results.timeseries : (n_frames,) — viscosity function averaged over particles
        (lag 0 is exactly 0).
    results.visc_by_particle : (n_frames, n_particles) or None
    results.viscosity : slope of the linear fit (only with linear_fit_window).
    """

    def __init__(self, atomgroup, temp_avg=300.0, dim_type="xyz", linear_fit_window=None,
                 **kwargs):
        self._want_by_particle = bool(kwargs.pop("by_particle", True))
        self._distributed = bool(kwargs.pop("distributed", False))
        self._stage_dtype = kwargs.pop("stage_dtype", None)
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
        self._float32 = bool(kwargs.pop("float32", False))
        self._fft = bool(kwargs.pop("fft", False))
        if self._fft and self._float32:
            raise ValueError("fft=True and float32=True are exclusive")
        super().__init__(atomgroup.universe.trajectory, **kwargs)

        if isinstance(atomgroup, UpdatingAtomGroup):
            raise TypeError("UpdatingAtomGroups are not valid for viscosity computation")

        self.temp_avg = temp_avg
        self.dim_type = dim_type.lower()
        self.linear_fit_window = linear_fit_window
        self._dim, self.dim_fac = parse_dim_type(self.dim_type)

        self.atomgroup = atomgroup
        self.n_particles = len(self.atomgroup)
        self._ctx = None

    _parse_dim_type = staticmethod(parse_dim_type)

    # see VelocityAutocorr: atoms, not frames, are this path's parallel axis
    _analysis_algorithm_is_parallelizable = False

    @classmethod
    def get_supported_backends(cls):
        return ("serial",)

    def _pick_stage_dtype(self):
        """float32 when the trajectory hands out float32 velocities AND positions (MDAnalysis
        does): lossless, half the PCIe bytes; the device slabs are float64 (:128-134)."""
        if self._stage_dtype is not None:
            return np.dtype(self._stage_dtype)
        try:
            f32 = (np.asarray(self.atomgroup.velocities).dtype == np.float32
                   and np.asarray(self.atomgroup.positions).dtype == np.float32)
        except Exception:  # missing data: _single_frame raises NoDataError, as the reference does
            return np.dtype(np.float64)
        return np.dtype(np.float32) if f32 else np.dtype(np.float64)

    def _prepare(self):
        """Two pinned slabs (velocities, positions) + volumes + masses (:111-142)."""
        if self._ctx is None:
            self._ctx = _lib.Group(self._devices) if self._devices is not None else _lib.Context(self._device)
        self._ctx.set_option("direct_f32", int(self._float32))
        self._ctx.set_option("helfand_fft", int(self._fft))
        # float32 path: the device slabs keep float32 staging as float32 (half the footprint; the
        # kernels round the staged value to float32 anyway, so the results do not change)
        self._ctx.set_option("stage_device_f32", int(self._float32))
        self._lo, self._hi = 0, self.n_particles
        self._source = self.atomgroup  # whose velocities / positions a frame is read from
        if self._distributed:
            from .dist import shard_of_this_rank

            _, _, self._lo, self._hi = shard_of_this_rank(self.n_particles)
            self.results.particle_range = (self._lo, self._hi)
            # this rank's block only: the trajectory gathers hi - lo atoms per frame, not all of them
            self._source = self.atomgroup[self._lo:self._hi]
        self._n_local = self._hi - self._lo
        dtype = self._pick_stage_dtype()
        if self._devices is not None:
            # one pair of pinned slabs per GPU (its column block); filled in ONE frame loop
            vviews, xviews = self._ctx.stage_alloc(self.n_frames, self.n_particles, self.dim_fac, n_slabs=2,
                                                   dtype=dtype)
            self._velocities, self._positions = vviews, xviews
            self._targets = [(v, x, lo, hi) for v, x, (lo, hi) in zip(vviews, xviews, self._ctx.shards) if hi > lo]
            self.results.device_ranges = list(self._ctx.shards)
        else:
            self._velocities, self._positions = self._ctx.stage_alloc(
                self.n_frames, max(self._n_local, 1), self.dim_fac, n_slabs=2, dtype=dtype)
            self._targets = [(self._velocities, self._positions, 0, self._n_local)] if self._distributed else \
                [(self._velocities, self._positions, self._lo, self._hi)]
        self._volumes = np.zeros(self.n_frames)
        self._masses = np.asarray(self.atomgroup.masses, dtype=np.float64)[self._lo:self._hi]
        if self._n_local == 0:
            self._masses = np.ones(1)
        self.boltzmann = BOLTZMANN
        frame_bytes = max(1, 2 * self._n_local * self.dim_fac * dtype.itemsize)
        self._commit_every = max(1, _COMMIT_BYTES // frame_bytes)
        # the per-frame fill reads the Timestep's own arrays natively (ta_stage_frame) where it can
        self._rows = native_rows(self._source) if self._n_local else None
        self._committed = 0
        self.results.visc_by_particle = None
        # pinned home of the (n_frames, n_particles) result (:117-119), page-locked on a helper thread
        self._bp_home = None
        if self._want_by_particle and self._n_local:
            device_reduce = False
            if self._distributed:
                from .dist import uses_device_reduce

                device_reduce = uses_device_reduce()
            if not device_reduce:
                self._bp_home = self._ctx.result_home((self.n_frames, self._n_local))

    def _single_frame(self):
        """Stage volume, velocities and positions of one frame (:167-199)."""
        ts = self._ts
        if not (ts.has_velocities and ts.has_positions and ts.volume != 0):
            raise NoDataError(
                "Helfand viscosity computation requires "
                "velocities, positions, and box volume in the trajectory"
            )
        i = self._frame_index
        self._volumes[i] = ts.volume
        if self._n_local:
            if not stage_frame_native(self._ctx, 0, i, ts, "velocities", self._dim, self._rows):
                vel = np.asarray(self._source.velocities)
                for vview, xview, lo, hi in self._targets:
                    stage_columns(vview[i], vel, lo, hi, self._dim)
            if not stage_frame_native(self._ctx, 1, i, ts, "positions", self._dim, self._rows):
                pos = np.asarray(self._source.positions)
                for vview, xview, lo, hi in self._targets:
                    stage_columns(xview[i], pos, lo, hi, self._dim)
        if i + 1 - self._committed >= self._commit_every:
            self._ctx.stage_commit(self._committed, i + 1)
            self._committed = i + 1

    def _conclude(self):
        if self._committed < self.n_frames:
            self._ctx.stage_commit(self._committed, self.n_frames)
            self._committed = self.n_frames
        self._vol_avg = np.average(self._volumes)
        # everything is divided by 2 kB <V> T (:229-231)
        scale = 1.0 / (2 * self.boltzmann * self._vol_avg * self.temp_avg)
        device_reduce = False
        if self._distributed:
            from .dist import staged_timeseries_on_device, uses_device_reduce

            device_reduce = uses_device_reduce()
        if device_reduce:  # RCCL: the lag sums stay on the GPU through the reduce
            ts, bp = staged_timeseries_on_device(self._ctx, "helfand", self.n_frames, self._n_local,
                                                 self.n_particles, self._device, masses=self._masses,
                                                 scale=scale, by_particle=self._want_by_particle)
        else:
            home = self._bp_home.get() if self._bp_home is not None else None
            self._bp_home = None
            ts, bp = self._ctx.helfand_msd(self._masses, scale, by_particle=self._want_by_particle, out=home)
        if self._distributed and not device_reduce:
            from .dist import allreduce_mean_over_atoms

            if self._n_local == 0:
                ts, bp = np.zeros(self.n_frames), (None if bp is None else bp[:, :0])
            ts = allreduce_mean_over_atoms(ts, self._n_local, self.n_particles, self._device)
        self.results.visc_by_particle = bp
        self.results.timeseries = ts

        if self.linear_fit_window is not None:
            # the reference fits against lagtimes = arange(1, n_frames): its x axis
            # is one ahead of the timeseries index; the slope is unaffected (:235-245)
            lagtimes = np.arange(1, self.n_frames)
            lo, hi = self.linear_fit_window[0], self.linear_fit_window[1]
            fit = np.polyfit(lagtimes[lo:hi], self.results.timeseries[lo:hi], 1)
            self.results.viscosity = fit[0]

    def plot_viscosity_function(self):
        """Plot the viscosity function against lag index, marking the fit window (:247-272)."""
        import matplotlib.pyplot as plt

        plt.plot(np.arange(0, self.n_frames), self.results.timeseries, label="Viscosity Function")
        if self.linear_fit_window is not None:
            lo, hi = self.linear_fit_window[0], self.linear_fit_window[1]
            plt.axvline(lo, color="red", linestyle="--", label="Fit Start")
            plt.axvline(hi, color="blue", linestyle="--", label="Fit End")
        plt.xlabel("Lag-time")
        plt.ylabel("Viscosity Function")
        plt.title("Viscosity Function vs Lag-time")
        plt.legend()
        plt.show()
