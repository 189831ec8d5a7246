"""A tiny in-memory stand-in for the MDAnalysis objects the hot path reads.

NOT part of the product path: it exists so the API can be exercised (tests,
smoke, examples) on machines without MDAnalysis, the way the reference's tests
build in-memory trajectories with ``mda.Universe.empty(..., velocities=True)``
(/root/reference/transport_analysis/tests/test_velocityautocorr.py:46-57).  It
provides exactly what the hooks touch: ``atomgroup.universe.trajectory``,
``len(atomgroup)``, ``.velocities`` / ``.positions`` (fresh float32 copies),
``.masses``, and per-frame ``ts.has_velocities``, ``ts.has_positions``,
``ts.volume``, ``ts.frame``, ``ts.time``.
"""
from __future__ import annotations

import numpy as np


class Timestep:
    def __init__(self, traj, frame):
        self._traj = traj
        self.frame = frame
        self.time = traj.time_offset + frame * traj.dt

    @property
    def has_velocities(self):
        return self._traj._vel is not None

    @property
    def has_positions(self):
        return self._traj._pos is not None

    @property
    def velocities(self):
        return self._traj._vel[self.frame]

    @property
    def positions(self):
        return self._traj._pos[self.frame]

    @property
    def dimensions(self):
        return self._traj.dimensions

    @property
    def volume(self):
        d = self._traj.dimensions
        if d is None:
            return 0.0
        return float(d[0] * d[1] * d[2])  # orthorhombic boxes only


class MemoryTrajectory:
    def __init__(self, positions, velocities, dimensions, dt, time_offset=0.0):
        self._pos = positions
        self._vel = velocities
        self.dimensions = dimensions
        self.dt = dt
        self.time_offset = time_offset
        ref = positions if positions is not None else velocities
        self.n_frames = 0 if ref is None else ref.shape[0]
        self.ts = Timestep(self, 0) if self.n_frames else None

    def __len__(self):
        return self.n_frames

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.n_frames))]
        if i < 0:
            i += self.n_frames
        if not 0 <= i < self.n_frames:
            raise IndexError(i)
        self.ts = Timestep(self, i)
        return self.ts

    def __iter__(self):
        for i in range(self.n_frames):
            yield self[i]


class AtomGroup:
    def __init__(self, universe, indices):
        self.universe = universe
        self.indices = np.asarray(indices, dtype=np.int64)
        n = universe._n_atoms
        # the whole universe in order: a frame is handed out by one plain copy, not a gather
        self._everything = len(self.indices) == n and bool(np.array_equal(self.indices, np.arange(n)))

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, item):
        return AtomGroup(self.universe, np.atleast_1d(self.indices[item]))

    @property
    def n_atoms(self):
        return len(self.indices)

    @property
    def ix(self):
        """the atoms' indices in the universe (MDAnalysis' AtomGroup.ix)"""
        return self.indices

    def _frame(self, arr, what):
        if arr is None:
            from ._base import NoDataError

            raise NoDataError(f"This Timestep has no {what}")
        frame = arr[self.universe.trajectory.ts.frame]
        if self._everything:
            return np.array(frame, dtype=np.float32)
        return np.array(frame[self.indices], dtype=np.float32)

    @property
    def velocities(self):
        return self._frame(self.universe.trajectory._vel, "velocities")

    @property
    def positions(self):
        return self._frame(self.universe.trajectory._pos, "positions")

    @property
    def masses(self):
        return np.array(self.universe._masses[self.indices], dtype=np.float64)


class ArrayUniverse:
    """``positions`` / ``velocities``: (n_frames, n_atoms, 3) arrays or None."""

    def __init__(self, positions=None, velocities=None, masses=None, dimensions=None, dt=1.0,
                 n_atoms=None, n_frames=None):
        # like an MDAnalysis Timestep, the trajectory holds float32 coordinates: ts.velocities / ts.positions
        # and AtomGroup.velocities / .positions hand out the same values (no copy when float32 came in)
        if positions is not None:
            positions = np.ascontiguousarray(positions, dtype=np.float32)
        if velocities is not None:
            velocities = np.ascontiguousarray(velocities, dtype=np.float32)
        ref = positions if positions is not None else velocities
        if ref is None:
            if n_atoms is None or n_frames is None:
                raise ValueError("need arrays, or n_atoms and n_frames")
            positions = np.zeros((n_frames, n_atoms, 3), dtype=np.float32)
            ref = positions
        self._n_atoms = ref.shape[1]
        self._masses = (np.ones(self._n_atoms) if masses is None
                        else np.asarray(masses, dtype=np.float64))
        self.trajectory = MemoryTrajectory(positions, velocities, dimensions, dt)
        self.atoms = AtomGroup(self, np.arange(self._n_atoms))
