"""NumPy restatement of the reference's time-correlation arithmetic.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Every function cites
the lines of ``/root/reference/transport_analysis`` whose arithmetic it
restates; the functions take plain arrays instead of going through
``AnalysisBase``.

Parity pinning
--------------
* ``vacf_windowed`` and ``helfand`` restate pure-NumPy bodies of the reference.
  ``tests/golden/make_golden.py`` executed the reference's *own* methods
  (``VelocityAutocorr._conclude_simple``, ``ViscosityHelfand._conclude``) and
  its own known-answer generators (``characteristic_poly``,
  ``characteristic_poly_helfand``) in the build container and committed their
  outputs; ``tests/test_oracle.py`` checks this file against them.
* ``vacf_fft`` goes through ``tidynamics.acf`` in the reference
  (``velocityautocorr.py:208-215``).  tidynamics (``>=1.0.0``, unpinned:
  ``pyproject.toml:22``) is a third-party dependency that is NOT under
  ``/root/reference`` and is not installed anywhere we can run, so its
  published algorithm is restated here (``tidynamics_acf``) and pinned by the
  reference's own fixtures for that seam: the closed-form step-trajectory
  polynomial (``tests/test_velocityautocorr.py:79-123,454-483``) and the
  N=10 vector printed in ``docs/tutorials/vacf_testing_examples.ipynb``.
"""

from __future__ import annotations

import numpy as np

# MDAnalysis.units.constants["Boltzmann_constant"], kJ/(mol K).  Pinned to 12
# digits by the Helfand toy-system outputs of the reference
# (docs/tutorials/helfand_dev_toy_system.ipynb).
BOLTZMANN_KJ_PER_MOL_K = 8.314462159e-3

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
    """velocityautocorr.py:155-176 / viscosity.py:144-165."""
    if dim_str not in _DIM_KEYS:
        raise ValueError(
            "invalid dim_type: {} specified, please specify one of xyz, "
            "xy, xz, yz, x, y, z".format(dim_str)
        )
    cols = _DIM_KEYS[dim_str]
    return list(cols), len(cols)


# --------------------------------------------------------------------------
# tidynamics.acf, restated (tidynamics 1.x, _correlation.py).  Call site:
# velocityautocorr.py:211-213.
# --------------------------------------------------------------------------
def tidynamics_n_fft(n):
    """Power of two tidynamics pads to before doubling: 2**ceil(log2(n+1))."""
    e = int(np.ceil(np.log2(n + 1)))
    p = 2**e
    if n == p:
        return n
    if n < p:
        return p
    return 2 ** (e + 1)


def tidynamics_autocorrelation_1d(series):
    n = len(series)
    n_fft = tidynamics_n_fft(n)
    padded = np.zeros(2 * n_fft)
    padded[:n] = series
    spectrum = np.fft.fft(padded)
    corr = np.fft.ifft(spectrum * spectrum.conj())[:n].real
    return corr / (n - np.arange(n))


def tidynamics_acf(data):
    """(T,) -> (T,), or (T, D) -> (T,) summed over the D columns."""
    data = np.asarray(data)
    if data.ndim == 1:
        return tidynamics_autocorrelation_1d(data)
    total = tidynamics_autocorrelation_1d(data[:, 0])
    for j in range(1, data.shape[1]):
        total = total + tidynamics_autocorrelation_1d(data[:, j])
    return total


# --------------------------------------------------------------------------
# VACF
# --------------------------------------------------------------------------
def vacf_fft(velocities):
    """velocityautocorr.py:208-215 (``_conclude_fft``).

    velocities: (T, A, D) float64.  Returns (vacf_by_particle (T, A),
    timeseries (T,)).  Per-atom Python loop, as in the reference.
    """
    v = np.asarray(velocities, dtype=np.float64)
    n_frames, n_particles, _ = v.shape
    by_particle = np.zeros((n_frames, n_particles))
    for n in range(n_particles):
        by_particle[:, n] = tidynamics_acf(v[:, n, :])
    return by_particle, by_particle.mean(axis=1)


def vacf_fft_batched(velocities, atom_block=512):
    """Same quantity as ``vacf_fft`` with the atom loop vectorised.

    Not a restatement of reference control flow: a faster checker for
    mid-sized parity cases; ``tests/test_oracle.py`` ties it to ``vacf_fft``.
    """
    v = np.asarray(velocities, dtype=np.float64)
    n_frames, n_particles, _ = v.shape
    length = 2 * tidynamics_n_fft(n_frames)
    norm = (n_frames - np.arange(n_frames))[:, None]
    by_particle = np.empty((n_frames, n_particles))
    for lo in range(0, n_particles, atom_block):
        blk = v[:, lo : lo + atom_block, :]
        spec = np.fft.fft(blk, n=length, axis=0)
        power = (spec.real**2 + spec.imag**2).sum(axis=2)
        corr = np.fft.ifft(power, axis=0)[:n_frames].real
        by_particle[:, lo : lo + atom_block] = corr / norm
    return by_particle, by_particle.mean(axis=1)


def vacf_windowed(velocities):
    """velocityautocorr.py:217-238 (``_conclude_simple``), lag 0..T-1."""
    v = np.asarray(velocities, dtype=np.float64)
    n_frames, n_particles, _ = v.shape
    by_particle = np.zeros((n_frames, n_particles))
    for lag in range(n_frames):
        prod = v[: n_frames - lag, :, :] * v[lag:, :, :]
        per_frame = np.sum(prod, axis=-1)
        by_particle[lag, :] = np.mean(per_frame, axis=0)
    return by_particle, by_particle.mean(axis=1)


# --------------------------------------------------------------------------
# Helfand viscosity function
# --------------------------------------------------------------------------
def helfand(
    velocities,
    positions,
    masses,
    volumes,
    temp_avg=300.0,
    boltzmann=BOLTZMANN_KJ_PER_MOL_K,
):
    """viscosity.py:201-233 (``ViscosityHelfand._conclude`` up to the fit).

    velocities, positions: (T, A, D) float64; masses: (A,); volumes: (T,).
    Returns (visc_by_particle (T, A), timeseries (T,)).  Row 0 stays 0.
    """
    v = np.asarray(velocities, dtype=np.float64)
    x = np.asarray(positions, dtype=np.float64)
    n_frames, n_particles, _ = v.shape
    m = np.asarray(masses, dtype=np.float64).reshape((1, n_particles, 1))
    vol_avg = np.average(np.asarray(volumes, dtype=np.float64))

    by_particle = np.zeros((n_frames, n_particles))
    for lag in range(1, n_frames):
        diff = m * v[:-lag, :, :] * x[:-lag, :, :] - m * v[lag:, :, :] * x[lag:, :, :]
        sq = np.square(diff).mean(axis=-1)
        by_particle[lag, :] = np.mean(sq, axis=0)
    by_particle = by_particle / (2 * boltzmann * vol_avg * temp_avg)
    return by_particle, by_particle.mean(axis=1)


def helfand_fit(timeseries, fit_window):
    """viscosity.py:235-245: slope over ``lagtimes = arange(1, T)`` (the
    reference's x axis starts at 1 while the timeseries index starts at 0)."""
    n_frames = len(timeseries)
    lagtimes = np.arange(1, n_frames)
    s, e = fit_window[0], fit_window[1]
    return np.polyfit(lagtimes[s:e], timeseries[s:e], 1)[0]


# --------------------------------------------------------------------------
# Synthetic inputs shared by tests and bench.py.
# --------------------------------------------------------------------------
def synthetic_velocities(n_frames, n_atoms, dim=3, seed=20250824):
    rng = np.random.Generator(np.random.Philox(seed))
    return rng.standard_normal((n_frames, n_atoms, dim))


def synthetic_helfand(n_frames, n_atoms, dim=3, seed=20250829, box=60.0):
    rng = np.random.Generator(np.random.Philox(seed))
    v = rng.standard_normal((n_frames, n_atoms, dim))
    x0 = rng.uniform(0.0, box, size=(1, n_atoms, dim))
    x = x0 + 0.002 * np.cumsum(v, axis=0)
    masses = np.resize(np.array([15.999, 1.008, 1.008]), n_atoms)
    volumes = np.full(n_frames, box**3)
    return v, x, masses, volumes
