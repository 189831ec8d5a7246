"""Upper boundary of the hot path: the MDAnalysis ``AnalysisBase`` contract.

When MDAnalysis is importable the analysis classes subclass the real
``MDAnalysis.analysis.base.AnalysisBase`` and use its ``Results``,
``UpdatingAtomGroup``, ``NoDataError`` and ``units.constants``.  When it is not
(the build container and the GPU box have no MDAnalysis), the minimal
stand-ins below provide the same template-method protocol the reference relies
on (/root/reference/transport_analysis/velocityautocorr.py:120,142-206):
``run(start, stop, step, frames)`` -> ``_setup_frames`` -> ``_prepare`` ->
per frame ``_single_frame`` (with ``_frame_index``, ``_ts``, ``frames``,
``times`` maintained) -> ``_conclude`` -> ``self``.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - exercised only where MDAnalysis exists
    from MDAnalysis.analysis.base import AnalysisBase, Results
    from MDAnalysis.core.groups import UpdatingAtomGroup
    from MDAnalysis.exceptions import NoDataError
    from MDAnalysis.units import constants as _mda_constants

    HAVE_MDANALYSIS = True
    try:
        BOLTZMANN = _mda_constants["Boltzmann_constant"]
    except KeyError:  # MDAnalysis < 2.6 spelling (viscosity.py:138-142)
        BOLTZMANN = _mda_constants["Boltzman_constant"]
except ImportError:
    HAVE_MDANALYSIS = False
    #: kJ/(mol K); MDAnalysis.units.constants["Boltzmann_constant"]
    BOLTZMANN = 8.314462159e-3

    class NoDataError(ValueError):
        """Raised when a trajectory lacks data an analysis needs."""

    class UpdatingAtomGroup:  # only ever used for isinstance checks
        pass

    class Results(dict):
        """dict with attribute access, like MDAnalysis.analysis.base.Results."""

        def __getattr__(self, key):
            try:
                return self[key]
            except KeyError as err:
                raise AttributeError(f"'Results' object has no attribute '{key}'") from err

        def __setattr__(self, key, value):
            self[key] = value

        def __delattr__(self, key):
            try:
                del self[key]
            except KeyError as err:
                raise AttributeError(f"'Results' object has no attribute '{key}'") from err

    class AnalysisBase:
        """Serial frame loop with the hooks of MDAnalysis' AnalysisBase."""

        def __init__(self, trajectory, verbose=False, **kwargs):
            self._trajectory = trajectory
            self._verbose = verbose
            self.results = Results()

        def _setup_frames(self, trajectory, start=None, stop=None, step=None, frames=None):
            if frames is not None:
                if not all(opt is None for opt in (start, stop, step)):
                    raise ValueError("start/stop/step cannot be combined with frames")
                index = list(frames)
                self.start = self.stop = self.step = None
            else:
                n = len(trajectory)
                rng = range(*slice(start, stop, step).indices(n))
                index = list(rng)
                self.start, self.stop, self.step = rng.start, rng.stop, rng.step
            self._frame_indices = index
            self.n_frames = len(index)
            self.frames = np.zeros(self.n_frames, dtype=int)
            self.times = np.zeros(self.n_frames)

        def _prepare(self):
            pass

        def _single_frame(self):
            raise NotImplementedError

        def _conclude(self):
            pass

        def run(self, start=None, stop=None, step=None, frames=None, verbose=None, **kwargs):
            self._setup_frames(self._trajectory, start=start, stop=stop, step=step, frames=frames)
            self._prepare()
            for i, idx in enumerate(self._frame_indices):
                ts = self._trajectory[idx]
                self._frame_index = i
                self._ts = ts
                self.frames[i] = ts.frame
                self.times[i] = ts.time
                self._single_frame()
            self._conclude()
            return self


_DIM_KEYS = {
    "x": [0],
    "y": [1],
    "z": [2],
    "xy": [0, 1],
    "xz": [0, 2],
    "yz": [1, 2],
    "xyz": [0, 1, 2],
}


def parse_dim_type(dim_str):
    """Column indices and dimensionality factor for a (lower-cased) dim_type.

    Same table and error text as the reference
    (velocityautocorr.py:155-176, viscosity.py:144-165); order matters, so
    "yx" is invalid."""
    try:
        cols = _DIM_KEYS[dim_str]
    except KeyError:
        raise ValueError(
            "invalid dim_type: {} specified, please specify one of xyz, "
            "xy, xz, yz, x, y, z".format(dim_str)
        )
    return list(cols), len(cols)


def stage_columns(dst, src, lo, hi, dim):
    """dst[: hi - lo] = src[lo:hi][:, dim] (the reference's per-frame slab fill,
    velocityautocorr.py:192-194, viscosity.py:189-199) without the temporary the fancy index
    makes: every dim_type of the table is an arithmetic progression of columns, i.e. a slice
    (one strided copy straight into the pinned slab; 15x faster for "xyz" at 50000 atoms)."""
    n = hi - lo
    step = dim[1] - dim[0] if len(dim) > 1 else 1
    if step > 0 and all(b - a == step for a, b in zip(dim, dim[1:])):
        dst[:n] = src[lo:hi, dim[0]:dim[-1] + 1:step]
    else:  # not reachable from parse_dim_type's table
        dst[:n] = src[lo:hi][:, dim]


def native_rows(group):
    """How ta_stage_frame finds the group's atoms in a Timestep's arrays: (first atom, index array or None,
    count, largest index) from MDAnalysis' ``AtomGroup.ix`` (the stand-in's ``indices``); None when the
    group has neither or $TA_AMD_NATIVE_STAGING=0 (the frames are then staged through NumPy views)."""
    import os

    from . import _lib

    if os.environ.get("TA_AMD_NATIVE_STAGING", "1") == "0":
        return None
    ix = getattr(group, "ix", None)
    if ix is None:
        ix = getattr(group, "indices", None)
    if ix is None:
        return None
    ix = np.asarray(ix)
    if ix.ndim != 1 or ix.size == 0 or not np.issubdtype(ix.dtype, np.integer) or int(ix.min()) < 0:
        return None
    lo, index, n = _lib.atom_rows(ix)
    return (lo, index, n), int(ix.max())


def stage_frame_native(ctx, slab, frame, ts, attr, cols, rows):
    """slab[frame] = ts.<attr>[group's atoms][:, cols] in one native pass (ta_stage_frame: gather, column
    selection, conversion, on a few host threads, GIL released) -- what the reference's
    ``self.atomgroup.<attr>[:, self._dim]`` computes through two temporaries
    (velocityautocorr.py:192-194, viscosity.py:189-199).  False when the Timestep's array is not
    something the native call can read in place (the caller falls back to the NumPy path)."""
    from . import _lib

    if rows is None or not hasattr(ctx, "stage_frame"):
        return False
    try:
        arr = getattr(ts, attr)
    except Exception:
        return False
    src = _lib.frame_source(arr)
    if src is None or rows[1] >= arr.shape[0] or cols[-1] >= arr.shape[1]:
        return False
    ctx.stage_frame(slab, frame, src, cols, rows[0])
    return True
