"""GPU parity: HIP kernels (through the C-ABI) against the CPU oracle and the
golden vectors produced by the reference.  Tolerance: scale-relative 1e-10
(BASELINE.json north_star), stated per assertion."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, scale_rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-10
TOL_F32 = 2e-6  # float32 products / block sums, float64 accumulation


def g(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="module")
def ctx():
    from transport_analysis_amd import _lib

    assert _lib.device_count() >= 1, "no GPU visible: the HIP path cannot run"
    c = _lib.Context(0)
    yield c
    c.close()


def run_vacf(ctx, v, fft, by_particle):
    T, A, D = v.shape
    (slab,) = ctx.stage_alloc(T, A, D, n_slabs=1)
    slab[...] = v
    ctx.stage_commit(0, T)
    ts, bp = (ctx.vacf_fft if fft else ctx.vacf_direct)(by_particle=by_particle)
    return ts, bp


def run_helfand(ctx, v, x, m, scale, by_particle):
    T, A, D = v.shape
    sv, sx = ctx.stage_alloc(T, A, D, n_slabs=2)
    sv[...] = v
    sx[...] = x
    ctx.stage_commit(0, T)
    return ctx.helfand_msd(m, scale, by_particle=by_particle)


def step(nstep, start=0, stop=None, step_=1, cols=(0, 1, 2)):
    t = np.arange(nstep, dtype=np.float64)[start:stop:step_]
    v = np.repeat(t[:, None, None], 3, axis=2)[:, :, list(cols)]
    x = np.repeat((t * t / 2)[:, None, None], 3, axis=2)[:, :, list(cols)]
    return v, x


# ------------------------------------------------------------------ VACF
@pytest.mark.parametrize("fft", [True, False])
@pytest.mark.parametrize("tag", ["T7_A1_D1", "T64_A5_D2", "T200_A33_D3"])
def test_vacf_golden_random(ctx, tag, fft):
    v = g(f"rand_vel_{tag}.npy")
    kind = "fft" if fft else "windowed"
    want_bp, want_ts = g(f"ref_vacf_{kind}_bp_{tag}.npy"), g(f"ref_vacf_{kind}_ts_{tag}.npy")
    ts, bp = run_vacf(ctx, v, fft, True)
    assert bp.shape == want_bp.shape and ts.shape == want_ts.shape
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL
    ts2, bp2 = run_vacf(ctx, v, fft, False)  # timeseries-only fast path
    assert bp2 is None
    assert scale_rel_err(ts2, want_ts) < TOL


@pytest.mark.parametrize("fft", [True, False])
@pytest.mark.parametrize("d", [1, 2, 3])
def test_vacf_step_kat_full(ctx, d, fft):
    # test_velocityautocorr.py:331-340 / :454-469 (N=5001); decimal=4 (windowed), 3 (FFT)
    v, _ = step(5001, cols=range(d))
    poly = g(f"kat_vacf_poly_N5001_D{d}.npy")
    for by_particle in (False, True):
        ts, _ = run_vacf(ctx, v, fft, by_particle)
        np.testing.assert_almost_equal(ts, poly, decimal=3 if fft else 4)
        assert scale_rel_err(ts, poly) < TOL


@pytest.mark.parametrize("fft", [True, False])
@pytest.mark.parametrize("d,cols", [(1, [1]), (2, [0, 2]), (3, [0, 1, 2])])
def test_vacf_step_kat_sliced(ctx, d, cols, fft):
    v, _ = step(5001, 10, 1000, 10, cols)
    poly = g(f"kat_vacf_poly_10_1000_10_D{d}.npy")
    ts, bp = run_vacf(ctx, v, fft, True)
    np.testing.assert_almost_equal(ts, poly, decimal=3 if fft else 4)
    assert scale_rel_err(ts, poly) < TOL
    assert scale_rel_err(bp[:, 0], poly) < TOL


def test_vacf_n10_notebook(ctx):
    const = json.load(open(os.path.join(GOLDEN, "reference_constants.json")))
    v, _ = step(10)
    for fft in (True, False):
        ts, _ = run_vacf(ctx, v, fft, False)
        np.testing.assert_allclose(ts, const["notebook_poly_step_N10"], rtol=0, atol=1e-10)


SHAPES = [(1, 1, 1), (2, 3, 3), (5, 2, 2), (16, 7, 3), (17, 4, 1), (33, 9, 2), (100, 40, 3),
          (129, 3, 3), (257, 33, 3), (640, 11, 2), (641, 5, 3), (1000, 37, 3), (1025, 6, 1),
          (2049, 3, 3), (2561, 4, 2), (4097, 2, 3), (5121, 2, 1), (10000, 3, 3), (10240, 2, 2),
          (30, 5, 3), (64, 3, 2), (77, 4, 3), (200, 6, 1), (400, 5, 3), (2000, 3, 3),
          # first-stage radices 7, 9, 14, 18 (odd prime-power butterflies)
          (3500, 5, 3), (4600, 3, 2), (7000, 4, 3), (9100, 3, 1),
          # both sides of the one-pass limit (256 frames: 512 bins are pad enough) and of the wave-local transform (512)
          (255, 4, 3), (256, 5, 3), (256, 3, 1), (512, 3, 3), (513, 2, 3)]


def test_vacf_fft_short_series_share_a_transform(ctx):
    """Lag sums of short trajectories: up to 256 frames one pass of the 512-point transform, up to 128 / 64 / 32 frames
    2 / 4 / 8 column pairs in ONE transform (autocorrelations add; the series sit 512 / PACK rows apart and do not meet at
    lags < n_frames) — every length on both sides of the limits, pair counts that do not fill the last transform."""
    from oracle import numpy_oracle as orc

    for T in (1, 2, 3, 16, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257):
        for A, D in ((1, 1), (3, 3), (7, 2), (33, 3), (64, 1)):
            v = orc.synthetic_velocities(T, A, D, seed=T * 7 + A)
            ts, _ = run_vacf(ctx, v, True, False)
            want_bp, want_ts = orc.vacf_fft_batched(v)
            assert scale_rel_err(ts, want_ts) < TOL, (T, A, D)
            # by particle: up to 128 frames the two units of a dim = 3 particle share its transform
            ts2, bp = ctx.vacf_fft(by_particle=True)
            assert scale_rel_err(bp, want_bp) < TOL and scale_rel_err(ts2, want_ts) < TOL, (T, A, D)


@pytest.mark.parametrize("T,A,D", SHAPES)
def test_vacf_fft_vs_oracle_shapes(ctx, T, A, D):
    """Every FFT plan size incl. ragged/odd column counts, both output modes."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=1000 + T)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    ts, bp = run_vacf(ctx, v, True, True)
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL
    ts2, _ = run_vacf(ctx, v, True, False)
    assert scale_rel_err(ts2, want_ts) < TOL


@pytest.mark.parametrize("T,A,D", [(1, 2, 3), (3, 1, 1), (8, 5, 3), (9, 4, 2), (31, 6, 3),
                                   (500, 9, 3), (1000, 33, 3), (4099, 3, 2)])
def test_vacf_direct_vs_oracle_shapes(ctx, T, A, D):
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=2000 + T)
    want_bp, want_ts = orc.vacf_fft_batched(v)  # same quantity (reference asserts equality)
    if T <= 1000:
        want_bp, want_ts = orc.vacf_windowed(v)
    ts, bp = run_vacf(ctx, v, False, True)
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL
    ts2, _ = run_vacf(ctx, v, False, False)
    assert scale_rel_err(ts2, want_ts) < TOL


@pytest.mark.parametrize("T,A,D", [(1, 1, 1), (2, 1, 1), (15, 3, 1), (16, 5, 3), (17, 2, 2), (240, 4, 3), (241, 3, 3),
                                   (255, 7, 3), (256, 3, 3), (257, 9, 1), (271, 4, 3), (272, 4, 3), (273, 5, 2),
                                   (511, 6, 3), (513, 11, 3), (1000, 37, 3), (2049, 8, 3), (4100, 3, 3),
                                   (5000, 21, 3), (9000, 3, 1), (300, 2001, 1), (300, 2003, 3)])
@pytest.mark.parametrize("form", [3])
def test_vacf_direct_lag_sums_on_the_matrix_cores(ctx, T, A, D, form):
    """Windowed VACF without the by-particle array = diagonal sums of the frames' Gram matrix on the FP64 matrix cores:
    "direct_mfma" 3 (the default from 112 frames) k_band_bp_vacf with a unit's particles summed in its
    accumulators (k-slots from the time axis, bandbp_kernels.hpp) — against the oracle and against the vector kernel
    ("direct_mfma" 0; the column-packed second form of rounds 4-5 is tools/band/ since round 6); frame counts on
    both sides of the 16-frame blocks and the 256-lag groups, odd column counts (the unpaired column's zero
    partner), few and many columns."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=2600 + T)
    want_ts = orc.vacf_windowed(v)[1] if T <= 1000 else orc.vacf_fft_batched(v)[1]
    ctx.set_option("timeline", 1)
    ctx.set_option("direct_mfma", form)
    try:
        ts_m, _ = run_vacf(ctx, v, False, False)
        assert [n for n, _ in ctx.kernel_timeline()] == ["k_band_bp_vacf"]
        ts_again, _ = ctx.vacf_direct(by_particle=False)
        assert np.array_equal(ts_m, ts_again)  # fixed summation order: the same bits every launch
        ctx.set_option("direct_mfma", 0)
        ts_v, _ = ctx.vacf_direct(by_particle=False)
        assert "k_direct" in [n for n, _ in ctx.kernel_timeline()]
    finally:
        ctx.set_option("direct_mfma", 1)
        ctx.set_option("timeline", 0)
    assert scale_rel_err(ts_m, want_ts) < TOL
    assert scale_rel_err(ts_m, ts_v) < 1e-12


@pytest.mark.parametrize("T,A,D", [(1, 1, 1), (2, 2, 3), (15, 3, 1), (16, 5, 3), (17, 2, 2), (63, 3, 3), (64, 4, 3), (65, 2, 3),
                                   (240, 4, 3), (241, 3, 3), (255, 7, 3), (256, 3, 3), (257, 9, 1), (271, 4, 2),
                                   (272, 4, 3), (511, 6, 3), (513, 11, 3), (1000, 37, 3), (2049, 8, 3), (4100, 3, 3),
                                   (5000, 21, 3), (9000, 3, 1), (300, 2001, 3), (300, 1001, 1)])
def test_vacf_direct_by_particle_on_the_matrix_cores(ctx, T, A, D):
    """Windowed VACF WITH the by-particle array (the class default): k_band_bp_vacf (bandbp_kernels.hpp: the
    MFMA's k-slots filled from the time axis, a particle's column in a per-wave LDS ring) against the oracle
    and against the vector kernel it replaces ("direct_mfma" 0); frame counts on both sides of the 16-frame
    blocks, the 64-frame chunks and the 256-lag units, every dim, odd particle counts (columns that start in the
    second half of a pair), and the same bits on every launch although two units add their halves of some lags."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=2800 + T)
    want_bp, want_ts = orc.vacf_windowed(v) if T <= 1000 else orc.vacf_fft_batched(v)
    ctx.set_option("timeline", 1)
    ctx.set_option("direct_mfma", 3)  # (the default, 1, takes the vector kernel below 144 frames)
    try:
        ts_m, bp_m = run_vacf(ctx, v, False, True)
        assert "k_band_bp_vacf" in [n for n, _ in ctx.kernel_timeline()]
        ts_again, bp_again = ctx.vacf_direct(by_particle=True)
        assert np.array_equal(bp_m, bp_again) and np.array_equal(ts_m, ts_again)
        ctx.set_option("direct_mfma", 0)
        ts_v, bp_v = ctx.vacf_direct(by_particle=True)
        assert "k_direct" in [n for n, _ in ctx.kernel_timeline()]
    finally:
        ctx.set_option("direct_mfma", 1)
        ctx.set_option("timeline", 0)
    assert bp_m.shape == (T, A)
    assert scale_rel_err(bp_m, want_bp) < TOL and scale_rel_err(ts_m, want_ts) < TOL
    assert scale_rel_err(bp_m, bp_v) < 1e-12 and scale_rel_err(ts_m, ts_v) < 1e-12
    # every particle on its own scale (a quiet particle next to a loud one)
    for n in range(0, A, max(1, A // 7)):
        assert scale_rel_err(bp_m[:, n], want_bp[:, n]) < TOL


@pytest.mark.parametrize("T,A,D", [(1, 1, 1), (2, 2, 3), (15, 3, 1), (16, 5, 3), (17, 2, 2), (241, 3, 3), (256, 3, 3),
                                   (257, 9, 1), (272, 4, 3), (273, 5, 2), (513, 11, 3), (1000, 37, 3), (2049, 8, 3),
                                   (4100, 3, 3), (5000, 7, 3), (300, 2001, 1), (300, 2001, 3)])
@pytest.mark.parametrize("form", [3])
def test_helfand_lag_sums_on_the_matrix_cores(ctx, T, A, D, form):
    """Einstein-Helfand mean squared differences without the by-particle array on the FP64 matrix cores ("direct_mfma" 3 =
    the default at every length): k_band_bp_helf with a unit's particles summed in its accumulators (k-slots from
    the time axis), on the product slab with rows centred on a nearby frame, against the oracle (viscosity.py:201-233: difference
    first) and against the vector kernel ("direct_mfma" 0); positions with a large offset and a drift, so that P
    is far from zero-mean."""
    from oracle import numpy_oracle as orc

    v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=2700 + T)
    x = x + 50.0 + 0.01 * np.arange(T)[:, None, None]
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
    want_ts = orc.helfand(v, x, m, vol, 300.0)[1]
    ctx.set_option("timeline", 1)
    ctx.set_option("direct_mfma", form)
    try:
        ts_m, _ = run_helfand(ctx, v, x, m, scale, False)
        assert [n for n, _ in ctx.kernel_timeline()] == ["k_helfand_product", "k_band_bp_helf"]
        ts_again, _ = ctx.helfand_msd(m, scale, by_particle=False)
        assert np.array_equal(ts_m, ts_again)
        ctx.set_option("direct_mfma", 0)
        ts_v, _ = ctx.helfand_msd(m, scale, by_particle=False)
        assert "k_direct" in [n for n, _ in ctx.kernel_timeline()]
    finally:
        ctx.set_option("direct_mfma", 1)
        ctx.set_option("timeline", 0)
    assert ts_m[0] == 0.0
    assert scale_rel_err(ts_m, want_ts) < TOL
    assert scale_rel_err(ts_m, ts_v) < 1e-11
    if T > 4:  # lag by lag, not only against the largest value: the centring keeps short lags exact
        rel = np.abs(ts_m[1:] - ts_v[1:]) / np.abs(ts_v[1:])
        assert rel.max() < 1e-9, (rel.argmax() + 1, rel.max())


@pytest.mark.parametrize("T,A,D", [(1, 1, 1), (2, 2, 3), (15, 3, 1), (16, 5, 3), (17, 2, 2), (63, 3, 3), (65, 2, 3), (241, 3, 3),
                                   (256, 3, 3), (257, 9, 1), (272, 4, 3), (273, 5, 2), (449, 3, 3), (513, 11, 3),
                                   (1000, 37, 3), (2049, 8, 3), (4100, 3, 3), (5000, 7, 3), (300, 2001, 3), (300, 999, 1)])
def test_helfand_by_particle_on_the_matrix_cores(ctx, T, A, D):
    """Einstein-Helfand WITH visc_by_particle (the class default), float64: k_band_bp_helf (bandbp_kernels.hpp: k-slots
    from the time axis, centred columns and their norms in a per-wave LDS ring) against the oracle
    (viscosity.py:201-233: difference first) and against the vector kernel ("direct_mfma" 0), particle by particle
    and lag by lag; P far from zero-mean (positions with an offset and a drift); frame counts on both sides of
    the 16-frame blocks, the 64-frame chunks, the 256-lag units and the 8-chunk ring; same bits every launch."""
    from oracle import numpy_oracle as orc

    v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=2900 + T)
    x = x + 50.0 + 0.01 * np.arange(T)[:, None, None]
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
    want_bp, want_ts = orc.helfand(v, x, m, vol, 300.0)
    ctx.set_option("timeline", 1)
    ctx.set_option("direct_mfma", 3)  # (the default, 1, takes k_short up to 64 frames and the vector kernel up to 351)
    try:
        ts_m, bp_m = run_helfand(ctx, v, x, m, scale, True)
        assert "k_band_bp_helf" in [n for n, _ in ctx.kernel_timeline()]
        ts_again, bp_again = ctx.helfand_msd(m, scale, by_particle=True)
        assert np.array_equal(bp_m, bp_again) and np.array_equal(ts_m, ts_again)
        ctx.set_option("direct_mfma", 0)
        ts_v, bp_v = ctx.helfand_msd(m, scale, by_particle=True)
        assert "k_direct" in [n for n, _ in ctx.kernel_timeline()]
    finally:
        ctx.set_option("direct_mfma", 1)
        ctx.set_option("timeline", 0)
    assert bp_m.shape == (T, A) and not bp_m[0].any()
    assert scale_rel_err(bp_m, want_bp) < TOL and scale_rel_err(ts_m, want_ts) < TOL
    assert scale_rel_err(bp_m, bp_v) < 1e-12
    if T > 1:  # lag by lag, particle by particle: short lags are orders of magnitude below the long ones
        rel = np.abs(bp_m[1:] - bp_v[1:]) / np.abs(bp_v[1:])
        assert rel.max() < 1e-9, (np.unravel_index(rel.argmax(), rel.shape), rel.max())


def test_helfand_by_particle_matrix_cores_on_a_pure_trend_and_in_other_units(ctx):
    """v = t, x = t^2 / 2 (P = m t^3 / 2: nine orders of magnitude, small lag-1 differences) by particle, every lag
    against the vector kernel; and P scaled by 1e-12 ... 1e+8: the result scales with the square."""
    from oracle import numpy_oracle as orc

    T = 3000
    v, x = step(T)
    m = np.array([1.0, 2.0])
    v = np.repeat(v[:, :1], 2, axis=1) * np.array([1.0, 0.5])[None, :, None]
    x = np.repeat(x[:, :1], 2, axis=1)
    _, bp_m = run_helfand(ctx, v, x, m, 1.0, True)
    ctx.set_option("direct_mfma", 0)
    try:
        _, bp_v = ctx.helfand_msd(m, 1.0, by_particle=True)
    finally:
        ctx.set_option("direct_mfma", 1)
    rel = np.abs(bp_m[1:] - bp_v[1:]) / np.abs(bp_v[1:])
    assert rel.max() < 1e-10, (np.unravel_index(rel.argmax(), rel.shape), rel.max())
    T, A = 700, 13
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=98)
    _, base = run_helfand(ctx, v, x, m, 1.0, True)
    for sv, sx in ((1e-6, 1e-6), (1e4, 1e4), (1e-3, 1.0)):
        _, got = run_helfand(ctx, v * sv, x * sx, m, 1.0, True)
        assert scale_rel_err(got / (sv * sx) ** 2, base) < TOL, (sv, sx)


@pytest.mark.parametrize("form", [3])
def test_helfand_matrix_cores_on_a_pure_trend(ctx, form):
    """The case the plain expansion S1 - 2 S2 loses (3.6e-9 on the reference's own
    step trajectory): v = t, x = t^2 / 2, so P = m t^3 / 2 grows by nine orders of magnitude while
    the lag-1 differences stay small.  Every lag against the vector kernel (difference first); both
    matrix-core forms ("direct_mfma" 1 / 2)."""
    T = 3000
    v, x = step(T)
    m = np.array([1.0, 2.0])
    v = np.repeat(v[:, :1], 2, axis=1) * np.array([1.0, 0.5])[None, :, None]
    x = np.repeat(x[:, :1], 2, axis=1)
    ctx.set_option("direct_mfma", form)
    ts_m, _ = run_helfand(ctx, v, x, m, 1.0, False)
    ctx.set_option("direct_mfma", 0)
    try:
        ts_v, _ = ctx.helfand_msd(m, 1.0, by_particle=False)
    finally:
        ctx.set_option("direct_mfma", 1)
    rel = np.abs(ts_m[1:] - ts_v[1:]) / np.abs(ts_v[1:])
    assert rel.max() < 1e-10, (rel.argmax() + 1, rel.max())


@pytest.mark.parametrize("T,A,D", [(1, 1, 1), (2, 2, 3), (15, 3, 1), (16, 5, 3), (17, 2, 2), (241, 3, 3), (256, 3, 3),
                                   (257, 9, 1), (272, 4, 3), (273, 5, 2), (513, 11, 3), (1000, 37, 3), (2049, 8, 3),
                                   (4100, 3, 3), (5000, 7, 3), (300, 2001, 1), (300, 2001, 3)])
@pytest.mark.parametrize("form", [3])
def test_helfand_float32_lag_sums_on_the_matrix_cores(ctx, T, A, D, form):
    """BASELINE configs[4]'s float32 path without the by-particle array on the FP32 matrix cores ("direct_mfma" 3 = the
    default at every length): k_band32_tp (band32tp_kernels.hpp: k-slots from the time axis, a unit's particles summed in its
    accumulators); float32 accumulators flushed into float64 — against the oracle (viscosity.py:201-233) at the float32 path's bar, 2e-6 of the series'
    scale, on the shapes of the float64 form's test: both sides of the 16-frame blocks and the 256-lag
    groups, ragged column counts (sextets with one and two pairs, an unpaired last column), P far from
    zero-mean.  Same bits every launch; and the float32 vector kernel ("direct_mfma" 0) agrees."""
    from oracle import numpy_oracle as orc

    v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=2700 + T)
    x = x + 50.0 + 0.01 * np.arange(T)[:, None, None]
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
    want_ts = orc.helfand(v, x, m, vol, 300.0)[1]
    ctx.set_option("direct_f32", 1)
    ctx.set_option("timeline", 1)
    ctx.set_option("direct_mfma", form)
    try:
        ts_m, _ = run_helfand(ctx, v, x, m, scale, False)
        assert [n for n, _ in ctx.kernel_timeline()] == ["k_helfand_product32", "k_band32_tp"]
        ts_again, _ = ctx.helfand_msd(m, scale, by_particle=False)
        assert np.array_equal(ts_m, ts_again)  # fixed summation order
        ctx.set_option("direct_mfma", 0)
        ts_v, _ = ctx.helfand_msd(m, scale, by_particle=False)
        assert "k_direct" in [n for n, _ in ctx.kernel_timeline()]
    finally:
        ctx.set_option("direct_mfma", 1)
        ctx.set_option("direct_f32", 0)
        ctx.set_option("timeline", 0)
    assert ts_m[0] == 0.0
    assert scale_rel_err(ts_m, want_ts) < TOL_F32
    assert scale_rel_err(ts_m, ts_v) < TOL_F32
    if T > 16:
        assert scale_rel_err(ts_m, want_ts) > 0.0  # really float32 arithmetic


@pytest.mark.parametrize("T,A", [(1, 2), (2, 3), (16, 5), (17, 3), (63, 3), (65, 2), (239, 2), (240, 3), (241, 2), (257, 7), (449, 3),
                                 (481, 5), (513, 2), (1000, 9), (2049, 3), (4100, 2), (300, 300)])
@pytest.mark.parametrize("form", [3])
def test_helfand_float32_by_particle_on_the_matrix_cores(ctx, T, A, form):
    """The float32 option WITH results.visc_by_particle (the class default output): k_band32_tp (k-slots from the time
    axis; any dim; "direct_mfma" 3 = the default at every length) — against the oracle at 2e-6 of the
    scale, frame counts on both sides of the 16-frame blocks, the 64-frame chunks and the 240- / 256-lag units; the
    timeseries is the mean of the by-particle array."""
    from oracle import numpy_oracle as orc

    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=2900 + T)
    x = x + 50.0 + 0.01 * np.arange(T)[:, None, None]
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
    want_bp, want_ts = orc.helfand(v, x, m, vol, 300.0)
    ctx.set_option("direct_f32", 1)
    ctx.set_option("timeline", 1)
    ctx.set_option("direct_mfma", form)
    try:
        ts, bp = run_helfand(ctx, v, x, m, scale, True)
        assert [n for n, _ in ctx.kernel_timeline()] == ["k_helfand_product32", "k_band32_tp",
                                                         "k_bp_transpose", "k_sum_partials"]
        ts2, bp2 = ctx.helfand_msd(m, scale, by_particle=True)
        assert np.array_equal(bp, bp2) and np.array_equal(ts, ts2)
        _, bp_d2 = run_helfand(ctx, v[:, :, :2], x[:, :, :2], m, scale, True)
        assert "k_band32_tp" in [n for n, _ in ctx.kernel_timeline()]
        _, bp_d1 = run_helfand(ctx, v[:, :, 1:2], x[:, :, 1:2], m, scale, True)
    finally:
        ctx.set_option("direct_mfma", 1)
        ctx.set_option("direct_f32", 0)
        ctx.set_option("timeline", 0)
    assert scale_rel_err(bp_d1, orc.helfand(v[:, :, 1:2], x[:, :, 1:2], m, vol, 300.0)[0]) < TOL_F32
    assert np.all(bp[0] == 0.0) and ts[0] == 0.0
    assert scale_rel_err(bp, want_bp) < TOL_F32
    assert scale_rel_err(ts, want_ts) < TOL_F32
    np.testing.assert_allclose(ts, bp.mean(axis=1), rtol=1e-12, atol=1e-13 * np.abs(ts).max())
    assert scale_rel_err(bp_d2, orc.helfand(v[:, :, :2], x[:, :, :2], m, vol, 300.0)[0]) < TOL_F32


def test_helfand_float32_matrix_cores_trend_units_and_float32_slabs(ctx):
    """Three properties of the float32 matrix-core Helfand path.  (a) The pure cubic trend (v = t, x = t^2 / 2:
    P grows by nine orders of magnitude): every lag within 2e-6 of the SCALE of the float64 difference-first
    result — lag by lag the float32 rounding of P itself (6e-8 |P|) dominates the short lags, for the vector
    float32 kernel too, which is why the float32 bar is a scale-relative one.  (b) The result does not depend
    on the unit of P: inputs scaled by 1e-6 and 1e+5 give results scaled by the square of the product, to
    the same 2e-6 (the norm slot's exact 1 never meets a norm in a sum).  (c) float32 device slabs
    ("stage_device_f32") feed the product kernel as they are: same bits as float64 slabs of the same
    float32-representable values."""
    from oracle import numpy_oracle as orc

    T = 3000
    v, x = step(T)
    m = np.array([1.0, 2.0])
    v = np.repeat(v[:, :1], 2, axis=1) * np.array([1.0, 0.5])[None, :, None]
    x = np.repeat(x[:, :1], 2, axis=1)
    ts64, _ = run_helfand(ctx, v, x, m, 1.0, False)
    ctx.set_option("direct_f32", 1)
    try:
        ts32, _ = run_helfand(ctx, v, x, m, 1.0, False)
        assert scale_rel_err(ts32, ts64) < TOL_F32
        # (b) units
        T2, A2 = 700, 13
        v2, x2, m2, vol = orc.synthetic_helfand(T2, A2, 3, seed=99)
        base, _ = run_helfand(ctx, v2, x2, m2, 1.0, False)
        for sv, sx in ((1e-6, 1.0), (1e5, 1e5), (1e-4, 1e-4)):
            got, _ = run_helfand(ctx, v2 * sv, x2 * sx, m2, 1.0, False)
            assert scale_rel_err(got / (sv * sx) ** 2, base) < TOL_F32, (sv, sx)
        # (c) float32 device slabs, both matrix-core forms
        v32, x32 = v2.astype(np.float32), x2.astype(np.float32)
        for form, kernel in ((3, "k_band32_tp"),):
            ctx.set_option("direct_mfma", form)
            want, _ = run_helfand(ctx, v32.astype(np.float64), x32.astype(np.float64), m2, 1.0, False)
            ctx.set_option("stage_device_f32", 1)
            try:
                sv_, sx_ = ctx.stage_alloc(T2, A2, 3, n_slabs=2, dtype=np.float32)
                sv_[...] = v32
                sx_[...] = x32
                ctx.stage_commit(0, T2)
                ctx.set_option("timeline", 1)
                got, _ = ctx.helfand_msd(m2, 1.0, by_particle=False)
                assert [n for n, _ in ctx.kernel_timeline()] == ["k_helfand_product32", kernel]
            finally:
                ctx.set_option("timeline", 0)
                ctx.set_option("stage_device_f32", 0)
                ctx.stage_free()
            assert np.array_equal(got, want)
    finally:
        ctx.set_option("direct_mfma", 1)
        ctx.set_option("direct_f32", 0)


@pytest.mark.parametrize("T,A,D", [(100, 700, 3), (1000, 300, 2), (2561, 90, 3), (5121, 70, 3),
                                   (10000, 50, 1), (640, 203, 3)])
def test_fft_many_units_per_workgroup(ctx, T, A, D):
    """Persistent workgroups walking several column pairs / atoms each (8 workgroups only):
    the software pipeline across units and, by-particle, the per-atom inverse + reset."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=5000 + T)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    ctx.set_option("fft_nwg", 8)
    try:
        ts, bp = run_vacf(ctx, v, True, True)
        ts2, _ = run_vacf(ctx, v, True, False)
    finally:
        ctx.set_option("fft_nwg", 0)
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL
    np.testing.assert_allclose(ts, bp.mean(axis=1), rtol=1e-12, atol=1e-13 * np.abs(ts).max())
    assert scale_rel_err(ts2, want_ts) < TOL


@pytest.mark.parametrize("T,A,D,nwg", [(300, 77, 3, 2), (64, 100, 2, 1), (1000, 41, 3, 3)])
def test_direct_column_groups_ragged(ctx, T, A, D, nwg):
    """Workgroups of several column groups with an atom count that is not a multiple of the
    group count, several rounds per workgroup, both correlators and both arithmetic paths."""
    from oracle import numpy_oracle as orc

    v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=6000 + T)
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
    want_vbp, want_vts = orc.vacf_windowed(v)
    want_hbp, want_hts = orc.helfand(v, x, m, vol, 300.0)
    ctx.set_option("direct_nwg", nwg)
    try:
        for f32, tol in ((0, TOL), (1, TOL_F32)):
            ctx.set_option("direct_f32", f32)
            for chunk in (8, 10):
                ctx.set_option("direct_chunk", chunk)
                ts, bp = run_vacf(ctx, v, False, True)
                assert scale_rel_err(bp, want_vbp) < tol and scale_rel_err(ts, want_vts) < tol
                ts, bp = run_helfand(ctx, v, x, m, scale, True)
                assert scale_rel_err(bp, want_hbp) < tol and scale_rel_err(ts, want_hts) < tol
    finally:
        ctx.set_option("direct_nwg", 0)
        ctx.set_option("direct_f32", 0)
        ctx.set_option("direct_chunk", 0)


def test_device_entry_points_on_a_column_block(ctx):
    """ta_*_dev on a shard that is a column block of a wider resident slab (ld_row > n_atoms*dim)
    with a padded by-particle leading dimension: the multi-GPU calling convention."""
    import torch

    from oracle import numpy_oracle as orc

    T, A_all, D, lo, hi = 257, 40, 3, 7, 29
    v, x, m, vol = orc.synthetic_helfand(T, A_all, D, seed=77)
    A = hi - lo
    dv = torch.from_numpy(v).cuda()
    dx = torch.from_numpy(x).cuda()
    dm = torch.from_numpy(m[lo:hi].copy()).cuda()
    ld_row, ld_bp = A_all * D, A + 5
    st = torch.cuda.current_stream().cuda_stream
    off = lo * D * 8
    want_bp, _ = orc.vacf_fft_batched(v[:, lo:hi])
    hbp, _ = orc.helfand(v[:, lo:hi], x[:, lo:hi], m[lo:hi], vol, 300.0)
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
    for which in ("fft", "direct", "helfand"):
        lag = torch.zeros(T, dtype=torch.float64, device="cuda")
        bp = torch.full((T, ld_bp), -7.0, dtype=torch.float64, device="cuda")
        if which == "fft":
            ctx.vacf_fft_dev(dv.data_ptr() + off, T, A, D, ld_row, lag.data_ptr(), bp.data_ptr(), ld_bp, st)
        elif which == "direct":
            ctx.vacf_direct_dev(dv.data_ptr() + off, T, A, D, ld_row, lag.data_ptr(), bp.data_ptr(), ld_bp, st)
        else:
            ctx.helfand_msd_dev(dv.data_ptr() + off, dx.data_ptr() + off, dm.data_ptr(), T, A, D, ld_row,
                                scale, lag.data_ptr(), bp.data_ptr(), ld_bp, st)
        torch.cuda.synchronize()
        got = bp.cpu().numpy()
        want = hbp if which == "helfand" else want_bp
        assert scale_rel_err(got[:, :A], want) < TOL
        assert np.all(got[:, A:] == -7.0)  # the padding columns are not touched
        assert scale_rel_err(lag.cpu().numpy(), want.sum(axis=1)) < TOL
        lag2 = torch.zeros(T, dtype=torch.float64, device="cuda")  # lag sums without by_particle
        if which == "fft":
            ctx.vacf_fft_dev(dv.data_ptr() + off, T, A, D, ld_row, lag2.data_ptr(), 0, 0, st)
        elif which == "direct":
            ctx.vacf_direct_dev(dv.data_ptr() + off, T, A, D, ld_row, lag2.data_ptr(), 0, 0, st)
        else:
            ctx.helfand_msd_dev(dv.data_ptr() + off, dx.data_ptr() + off, dm.data_ptr(), T, A, D, ld_row,
                                scale, lag2.data_ptr(), 0, 0, st)
        torch.cuda.synchronize()
        assert scale_rel_err(lag2.cpu().numpy(), want.sum(axis=1)) < TOL


@pytest.mark.parametrize("T,A_all,D,lo,hi", [(13, 18, 1, 0, 1), (300, 8, 3, 1, 4), (5200, 6, 3, 2, 5),
                                           (9000, 4, 1, 1, 4), (640, 10, 2, 3, 8)])
def test_fft_shard_with_odd_column_count_in_an_even_slab(ctx, T, A_all, D, lo, hi):
    """A column block whose column count is odd while the slab's row length is even (found by
    tests/stress_gpu.py): the last column has no partner and must not pick up the neighbouring
    atom's data through a 16-byte load."""
    import torch

    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A_all, D, seed=9000 + T)
    dv = torch.from_numpy(v).cuda()
    A = hi - lo
    want_bp, _ = orc.vacf_fft_batched(v[:, lo:hi])
    st = torch.cuda.current_stream().cuda_stream
    for with_bp in (False, True):
        lag = torch.zeros(T, dtype=torch.float64, device="cuda")
        bp = torch.zeros((T, A), dtype=torch.float64, device="cuda")
        ctx.vacf_fft_dev(dv.data_ptr() + lo * D * 8, T, A, D, A_all * D, lag.data_ptr(),
                         bp.data_ptr() if with_bp else 0, A, st)
        torch.cuda.synchronize()
        assert scale_rel_err(lag.cpu().numpy(), want_bp.sum(axis=1)) < TOL
        if with_bp:
            assert scale_rel_err(bp.cpu().numpy(), want_bp) < TOL


def test_cabi_error_behaviour(ctx):
    """Negative status + message instead of a crash (include/ta_hip.h conventions)."""
    import torch

    from transport_analysis_amd import _lib

    c = _lib.Context(0)
    with pytest.raises(_lib.TAError) as e:  # compute before staging
        c.vacf_fft()
    assert e.value.code == -4
    buf = torch.zeros(64, dtype=torch.float64, device="cuda")
    for args in ((buf.data_ptr(), 4, 2, 4, 8),      # dim 4
                 (buf.data_ptr(), 4, 2, 3, 5),      # ld_row < n_atoms*dim
                 (buf.data_ptr(), 0, 2, 3, 6),      # no frames
                 (0, 4, 2, 3, 6)):                  # null slab
        with pytest.raises(_lib.TAError) as e:
            c.vacf_fft_dev(args[0], args[1], args[2], args[3], args[4], buf.data_ptr())
        assert e.value.code == -1 and str(e.value)
    with pytest.raises(_lib.TAError):  # by-particle leading dimension too small
        c.vacf_direct_dev(buf.data_ptr(), 4, 2, 3, 6, buf.data_ptr(), buf.data_ptr(), 1)
    with pytest.raises(_lib.TAError):
        c.set_option("no_such_option", 1)
    c.close()


@pytest.mark.parametrize("fft", [True, False])
def test_vacf_long_trajectory(ctx, fft):
    """n_frames beyond the on-chip limits (one on-chip transform stops at 10240 frames, an
    LDS-resident column of the direct correlator at 16376): outer radix for the FFT, the column
    staged in global memory for the direct correlator."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(17001, 3, 3, seed=17)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    ts, bp = run_vacf(ctx, v, fft, True)
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL


@pytest.mark.parametrize("T,A,D", [(10241, 3, 3), (12000, 5, 3), (12000, 7, 3), (20000, 11, 2), (16385, 2, 2), (20000, 4, 3),
                                   (20481, 3, 1), (33000, 2, 3), (50000, 1, 3), (70000, 1, 2),
                                   (90000, 1, 1), (140000, 1, 2), (163840, 1, 1)])
def test_vacf_fft_long_trajectory(ctx, T, A, D):
    """fft=True with n_frames beyond the largest on-chip transform: outer radix 2/4/8/16 while the
    rows are read + on-chip transforms per pass (csrc/wfft.hpp), lag sums and the by-particle
    array; covers every outer radix, an odd column count (unpaired last column), single-column
    units in either half of a pair, rows past the end."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=T % 1000 + A)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    ts, bp = run_vacf(ctx, v, True, False)
    assert bp is None
    assert ts.shape == want_ts.shape
    assert scale_rel_err(ts, want_ts) < TOL
    ts, bp = run_vacf(ctx, v, True, True)
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL


def test_vacf_beyond_the_fft_plans(ctx):
    """n_frames > 163840: ta_vacf_fft computes the same quantity with the O(T^2) correlators on the matrix
    cores — 10241 block lags, 641 groups of 16: the time-packed kernel's 641 units of one particle (lag sums and
    by particle), and under "direct_mfma" 0 the vector kernel (the column staged in an L2-resident buffer: it does not
    fit the LDS) — against the FFT oracle."""
    from oracle import numpy_oracle as orc

    T, A, D = 163841, 1, 2
    v = orc.synthetic_velocities(T, A, D, seed=77)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    ctx.set_option("timeline", 1)
    try:
        ts, bp = run_vacf(ctx, v, True, False)
        assert [n for n, _ in ctx.kernel_timeline()] == ["k_band_bp_vacf"]
        ctx.set_option("direct_mfma", 0)
        ts2, _ = ctx.vacf_fft(by_particle=False)
        assert "k_direct" in [n for n, _ in ctx.kernel_timeline()]
    finally:
        ctx.set_option("direct_mfma", 1)
        ctx.set_option("timeline", 0)
    assert scale_rel_err(ts, want_ts) < TOL and scale_rel_err(ts2, want_ts) < TOL
    ts, bp = ctx.vacf_fft(by_particle=True)
    assert scale_rel_err(bp, want_bp) < TOL and scale_rel_err(ts, want_ts) < TOL


def test_vacf_fft_long_trajectory_many_pairs_and_step_kat(ctx):
    """More column pairs than workgroups (several pairs per workgroup, accumulator blocks
    carried across pairs), and the reference's closed-form step trajectory at N = 12001."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(10500, 400, 3, seed=5)
    _, want_ts = orc.vacf_fft_batched(v)
    ts, _ = run_vacf(ctx, v, True, False)
    assert scale_rel_err(ts, want_ts) < TOL
    # the reference's step trajectory v[t] = t (test_velocityautocorr.py:59-93), N = 12001:
    # lag k is 3 * sum_i i (i + k) / (N - k), exact in integers
    n = 12001
    vs, _ = step(n)
    ts, _ = run_vacf(ctx, vs, True, False)
    lags = (0, 1, 7, 6000, 12000)
    exact = np.array([3.0 * sum(i * (i + kk) for i in range(n - kk)) / (n - kk) for kk in lags])
    got = ts[list(lags)]
    assert np.max(np.abs(got - exact) / np.max(np.abs(exact))) < TOL


def test_helfand_long_trajectory(ctx):
    from oracle import numpy_oracle as orc

    T = 16500
    v, x, m, vol = orc.synthetic_helfand(T, 2, 2, seed=18)
    # direct oracle at this length is O(T^2) slab work: check a handful of lags exactly
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
    ts, bp = run_helfand(ctx, v, x, m, scale, True)
    P = m[None, :, None] * v * x
    for lag in (1, 2, 777, 8000, T - 1):
        want = np.mean(np.square(P[:-lag] - P[lag:]).mean(axis=-1), axis=0) * scale
        np.testing.assert_allclose(bp[lag], want, rtol=1e-11)
        np.testing.assert_allclose(ts[lag], want.mean(), rtol=1e-11)
    assert ts[0] == 0.0


def test_vacf_config2_full_size(ctx):
    """BASELINE config[1]: 1000 x 10000 x 3 float64, FFT path vs the oracle."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(1000, 10000, 3, seed=20250826)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    ts, _ = run_vacf(ctx, v, True, False)
    assert scale_rel_err(ts, want_ts) < TOL
    ts, bp = run_vacf(ctx, v, True, True)
    assert scale_rel_err(ts, want_ts) < TOL
    assert scale_rel_err(bp, want_bp) < TOL
    tsd, _ = run_vacf(ctx, v, False, False)
    assert scale_rel_err(tsd, want_ts) < TOL


def _torch_lags(v, lags):
    """sum over atoms and dims of <v(t) v(t+k)>, by slab products on the GPU (float64)."""
    T = v.shape[0]
    return [float((v[: T - k] * v[k:]).sum().item()) / (T - k) for k in lags]


@pytest.mark.parametrize("mode,T,A", [("fft", 10000, 100000), ("direct", 5000, 50000)])
def test_vacf_baseline_full_size_properties(ctx, mode, T, A):
    """BASELINE configs[2] (FFT, 10000 x 100000 x 3) and configs[3] (windowed, 5000 x 50000 x 3)
    at FULL size through the device-pointer entry points, checked by size-independent
    properties: selected lags against slab products, lag 0 = mean squared speed, the two
    algorithms agree, and v -> 2v gives EXACTLY 4x (power-of-two scaling is exact in binary
    floating point through every step of both algorithms)."""
    import torch

    D = 3
    gen = torch.Generator(device="cuda")
    gen.manual_seed(20250824 + (3 if mode == "fft" else 4))
    v = torch.randn((T, A, D), dtype=torch.float64, device="cuda", generator=gen)
    st = torch.cuda.current_stream().cuda_stream
    call = ctx.vacf_fft_dev if mode == "fft" else ctx.vacf_direct_dev
    lag1 = torch.zeros(T, dtype=torch.float64, device="cuda")
    call(v.data_ptr(), T, A, D, A * D, lag1.data_ptr(), 0, 0, st)
    torch.cuda.synchronize()
    lags = [0, 1, 2, 7, T // 3, T // 2, T - 2, T - 1]
    want = _torch_lags(v, lags)
    scale = want[0]
    for k, w in zip(lags, want):
        assert abs(float(lag1[k].item()) - w) < TOL * scale, (k, float(lag1[k].item()), w)
    v *= 2.0
    lag2 = torch.zeros(T, dtype=torch.float64, device="cuda")
    call(v.data_ptr(), T, A, D, A * D, lag2.data_ptr(), 0, 0, st)
    torch.cuda.synchronize()
    assert torch.equal(lag2, 4.0 * lag1)
    if mode == "direct":  # the other algorithm on the same data (the reference asserts equality)
        lag3 = torch.zeros(T, dtype=torch.float64, device="cuda")
        ctx.vacf_fft_dev(v.data_ptr(), T, A, D, A * D, lag3.data_ptr(), 0, 0, st)
        torch.cuda.synchronize()
        assert float((lag3 - lag2).abs().max().item()) < TOL * 4.0 * scale
    del v
    torch.cuda.empty_cache()


def test_helfand_float32_baseline_shape_properties(ctx):
    """BASELINE configs[4]'s per-GPU share in time (20000 frames, float32 path) on a 2000-atom
    block: selected lags against slab differences in float64, lag 0 exactly 0, and the
    float64 path on the same data within the float32 tolerance."""
    import torch

    T, A, D = 20000, 2000, 3
    gen = torch.Generator(device="cuda")
    gen.manual_seed(20250824 + 5)
    v = torch.randn((T, A, D), dtype=torch.float64, device="cuda", generator=gen)
    x = 30.0 + 0.002 * torch.cumsum(v, dim=0)
    m = torch.tensor([15.999, 1.008, 1.008], dtype=torch.float64, device="cuda").repeat((A + 2) // 3)[:A].contiguous()
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for f32 in (1, 0):
        ctx.set_option("direct_f32", f32)
        try:
            lag = torch.zeros(T, dtype=torch.float64, device="cuda")
            ctx.helfand_msd_dev(v.data_ptr(), x.data_ptr(), m.data_ptr(), T, A, D, A * D, 1.0,
                                lag.data_ptr(), 0, 0, st)
            torch.cuda.synchronize()
        finally:
            ctx.set_option("direct_f32", 0)
        out[f32] = lag
    P = m[None, :, None] * v * x
    scale = None
    for k in (1, 2, 100, T // 2, T - 1):
        w = float(((P[: T - k] - P[k:]) ** 2).sum().item()) / (T - k) / D
        scale = scale or w
        assert abs(float(out[0][k].item()) - w) < TOL * max(w, scale)
        assert abs(float(out[1][k].item()) - w) < TOL_F32 * max(w, scale)
    assert float(out[0][0].item()) == 0.0 and float(out[1][0].item()) == 0.0
    del v, x, P
    torch.cuda.empty_cache()


def test_vacf_linearity_and_lag0(ctx):
    """Size-independent properties: lag 0 equals the mean squared speed; the lag
    sums of two atom blocks add up to the lag sum of their union."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(3000, 64, 3, seed=5)
    ts, _ = run_vacf(ctx, v, True, False)
    np.testing.assert_allclose(ts[0], np.mean(np.sum(v * v, axis=2)), rtol=1e-12)
    a, _ = run_vacf(ctx, v[:, :20], True, False)
    b, _ = run_vacf(ctx, v[:, 20:], True, False)
    assert scale_rel_err((20 * a + 44 * b) / 64, ts) < 1e-12


# --------------------------------------------------------------- Helfand
@pytest.mark.parametrize("tag", ["T9_A1_D1", "T50_A6_D2", "T120_A17_D3"])
def test_helfand_golden_random(ctx, tag):
    from transport_analysis_amd._base import BOLTZMANN

    z = np.load(os.path.join(GOLDEN, f"rand_helfand_in_{tag}.npz"))
    scale = 1.0 / (2 * BOLTZMANN * np.average(z["vol"]) * 313.0)
    ts, bp = run_helfand(ctx, z["v"], z["x"], z["m"], scale, True)
    want_bp, want_ts = g(f"ref_helfand_bp_{tag}.npy"), g(f"ref_helfand_ts_{tag}.npy")
    assert ts[0] == 0.0 and np.all(bp[0] == 0.0)
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL
    np.testing.assert_allclose(ts, want_ts, rtol=1e-7)  # the reference's own bar
    ts2, bp2 = run_helfand(ctx, z["v"], z["x"], z["m"], scale, False)
    assert bp2 is None and scale_rel_err(ts2, want_ts) < TOL


@pytest.mark.parametrize("d", [1, 2, 3])
def test_helfand_step_kat(ctx, d):
    # test_viscosity.py:180-208 (assert_allclose rtol=1e-7), N=5001 and sliced
    from transport_analysis_amd._base import BOLTZMANN

    scale = 1.0 / (2 * BOLTZMANN * 8.0 * 300.0)
    v, x = step(5001, cols=range(d))
    ts, _ = run_helfand(ctx, v, x, np.array([16.0]), scale, False)
    want = g(f"kat_helfand_poly_N5001_D{d}.npy")
    np.testing.assert_allclose(ts, want, rtol=1e-7)
    assert scale_rel_err(ts, want) < TOL
    v, x = step(5001, 10, 1000, 10, range(d))
    ts, _ = run_helfand(ctx, v, x, np.array([16.0]), scale, True)
    np.testing.assert_allclose(ts, g(f"kat_helfand_poly_10_1000_10_D{d}.npy"), rtol=1e-7)


@pytest.mark.parametrize("T,A,D", [(2, 3, 1), (8, 2, 3), (17, 5, 2), (300, 21, 3), (1001, 7, 3)])
def test_helfand_vs_oracle_shapes(ctx, T, A, D):
    from oracle import numpy_oracle as orc

    v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=3000 + T)
    want_bp, want_ts = orc.helfand(v, x, m, vol, 300.0)
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
    ts, bp = run_helfand(ctx, v, x, m, scale, True)
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL


@pytest.mark.parametrize("T,A,D", [(2, 3, 1), (17, 5, 2), (300, 21, 3), (1001, 7, 3), (5000, 3, 3)])
def test_float32_direct_paths_vs_oracle(ctx, T, A, D):
    """ta_set_option("direct_f32", 1): BASELINE configs[4]'s float32 Helfand path (and the
    same switch on the direct VACF).  Tolerance: 2e-6 of the series scale."""
    from oracle import numpy_oracle as orc

    v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=4000 + T)
    want_bp, want_ts = orc.helfand(v, x, m, vol, 300.0)
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
    ctx.set_option("direct_f32", 1)
    try:
        ts, bp = run_helfand(ctx, v, x, m, scale, True)
        assert ts[0] == 0.0 and np.all(bp[0] == 0.0)
        assert scale_rel_err(bp, want_bp) < TOL_F32
        assert scale_rel_err(ts, want_ts) < TOL_F32
        assert scale_rel_err(ts, want_ts) > 0.0 or T < 8  # really the float32 kernel
        want_bp, want_ts = orc.vacf_windowed(v)
        ts, bp = run_vacf(ctx, v, False, True)
        assert scale_rel_err(bp, want_bp) < TOL_F32
        assert scale_rel_err(ts, want_ts) < TOL_F32
    finally:
        ctx.set_option("direct_f32", 0)
    ts, _ = run_helfand(ctx, v, x, m, scale, False)
    assert scale_rel_err(ts, orc.helfand(v, x, m, vol, 300.0)[1]) < TOL  # switch is off again


@pytest.mark.parametrize("T,A,D", [(2, 3, 1), (9, 1, 1), (50, 6, 2), (300, 21, 3), (1001, 7, 3),
                                   (5000, 5, 3), (10240, 2, 2), (12000, 3, 3), (16500, 2, 3)])
def test_helfand_fft_option_vs_oracle(ctx, T, A, D):
    """Option "helfand_fft" (extension, csrc/helfand_fft.hip): lag sums as S1 - 2 S2 with S2 from
    the FFT path (on-chip and long-trajectory plans) -- same series as the reference's O(T^2)
    loop to the north-star tolerance in the scale-relative metric; lag 0 exactly 0."""
    from oracle import numpy_oracle as orc

    v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=T + A)
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
    want_bp, want_ts = orc.helfand(v, x, m, vol, 300.0)
    ctx.set_option("helfand_fft", 1)
    try:
        ts, bp = run_helfand(ctx, v, x, m, scale, False)
    finally:
        ctx.set_option("helfand_fft", 0)
    assert bp is None and ts.shape == want_ts.shape
    assert ts[0] == 0.0
    assert scale_rel_err(ts, want_ts) < TOL
    ctx.set_option("helfand_fft", 1)
    try:
        ts, bp = run_helfand(ctx, v, x, m, scale, True)  # per atom
    finally:
        ctx.set_option("helfand_fft", 0)
    assert np.all(bp[0] == 0.0)
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL


def test_helfand_fft_option_step_kat_and_class(ctx):
    """The reference's step trajectory (test_viscosity.py:115-132 / :180-208, N = 5001, m = 16,
    V = 8, T = 300): a smooth P, where short lags are far below P^2 -- the FFT option stays inside
    1e-10 of the series' scale (not of each value: see include/ta_hip.h).  The class takes
    fft=True with and without the per-particle array."""
    from transport_analysis_amd import ViscosityHelfand
    from transport_analysis_amd._base import BOLTZMANN
    from transport_analysis_amd._mini_mda import ArrayUniverse

    want = g("kat_helfand_poly_N5001_D3.npy")
    v, x = step(5001)
    ctx.set_option("helfand_fft", 1)
    try:
        ts, _ = run_helfand(ctx, v, x, np.array([16.0]), 1.0 / (2 * BOLTZMANN * 8.0 * 300.0), False)
    finally:
        ctx.set_option("helfand_fft", 0)
    assert ts[0] == 0.0
    assert scale_rel_err(ts, want) < TOL
    u = ArrayUniverse(positions=x[:300].astype(np.float32), velocities=v[:300].astype(np.float32),
                      masses=np.full(1, 16.0), dimensions=[2, 2, 2, 90, 90, 90])
    ref = ViscosityHelfand(u.atoms, by_particle=False).run()
    got = ViscosityHelfand(u.atoms, fft=True, by_particle=False).run()
    assert got.results.visc_by_particle is None
    assert scale_rel_err(got.results.timeseries, ref.results.timeseries) < TOL
    got = ViscosityHelfand(u.atoms, fft=True).run()
    assert scale_rel_err(got.results.visc_by_particle[:, 0], ref.results.timeseries) < TOL


@pytest.mark.parametrize("T", [20000, 28000])
def test_float32_helfand_long_trajectory(ctx, T):
    """20000 frames (configs[4]) fits LDS as float32; 28000 takes the L2-staged variant."""
    from oracle import numpy_oracle as orc

    v, x, m, vol = orc.synthetic_helfand(T, 2, 3, seed=19)
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
    ctx.set_option("direct_f32", 1)
    try:
        ts, bp = run_helfand(ctx, v, x, m, scale, True)
    finally:
        ctx.set_option("direct_f32", 0)
    P = m[None, :, None] * v * x
    for lag in (1, 2, 777, 8000, T - 1):
        want = np.mean(np.square(P[:-lag] - P[lag:]).mean(axis=-1), axis=0) * scale
        np.testing.assert_allclose(bp[lag], want, rtol=5e-6, atol=5e-6 * scale * np.max(np.abs(P)) ** 2)
    assert ts[0] == 0.0


def test_f32_staging_is_lossless(ctx):
    """MDAnalysis hands out float32; staging float32 and widening on the device
    must give exactly what staging the upcast float64 gives."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(300, 17, 3, seed=9).astype(np.float32)
    (slab,) = ctx.stage_alloc(300, 17, 3, n_slabs=1, dtype=np.float32)
    slab[...] = v
    ctx.stage_commit(0, 100)
    ctx.stage_commit(100, 300)
    ts32, _ = ctx.vacf_fft(by_particle=False)
    ts64, _ = run_vacf(ctx, v.astype(np.float64), True, False)
    np.testing.assert_array_equal(ts32, ts64)


# ------------------------------------------------------------------ layout / staging / generator
@pytest.mark.parametrize("T,A,D", [(1, 1, 1), (7, 3, 3), (64, 5, 1), (65, 33, 3), (300, 129, 2), (1000, 7, 3)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pair_major_layout_roundtrip(ctx, T, A, D, dtype):
    """ta_stage_commit (host -> pair-major device slab) and ta_stage_commit_dev (device ->
    pair-major) followed by ta_stage_read_dev give back the frames bit for bit: odd column
    counts (unpaired last column), partial tiles, float32 widening, piecewise commits."""
    import torch

    rng = np.random.default_rng(T * 1000 + A)
    v = rng.standard_normal((T, A, D)).astype(dtype)
    (slab,) = ctx.stage_alloc(T, A, D, n_slabs=1, dtype=dtype)
    slab[...] = v
    cut = T // 3
    ctx.stage_commit(0, cut)
    ctx.stage_commit(cut, T)
    out = torch.empty((T, A * D), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ctx.stage_read_dev(0, out.data_ptr(), A * D, st)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(T, A, D), v.astype(np.float64))
    ptr, pitch, n_pairs = ctx.stage_device(0)
    assert ptr and pitch >= T and pitch % 8 == 0 and n_pairs == (A * D + 1) // 2
    # device-side source with a row stride (a column block of a wider tensor)
    wide = torch.from_numpy(rng.standard_normal((T, A * D + 5))).cuda()
    ctx.stage_alloc_device(T, A, D, n_slabs=1)
    ctx.stage_commit_dev(0, wide.data_ptr() + 2 * 8, A * D + 5, 0, T, stream=st)
    ctx.stage_read_dev(0, out.data_ptr(), A * D, st)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), wide.cpu().numpy()[:, 2:2 + A * D])


@pytest.mark.parametrize("T,A,D,ld,off", [(130, 7, 3, 26, 2), (64, 64, 2, 128, 0), (257, 33, 3, 100, 0), (1, 1, 2, 2, 0),
                                           (200, 129, 1, 130, 0)])
def test_relayout_16_byte_path(ctx, T, A, D, ld, off):
    """Frame-major float64 rows with an EVEN row stride and a 16-byte aligned base take the
    transposition with 16-byte accesses on both sides (k_relayout_wide): odd column counts (the last
    column paired with zeros), partial tiles in both directions, a column block of a wider tensor."""
    import torch

    rng = np.random.default_rng(T + 7 * A)
    wide = torch.from_numpy(rng.standard_normal((T, ld))).cuda()
    st = torch.cuda.current_stream().cuda_stream
    ctx.stage_alloc_device(T, A, D, n_slabs=1)
    assert (wide.data_ptr() + off * 8) % 16 == 0 and ld % 2 == 0
    ctx.stage_commit_dev(0, wide.data_ptr() + off * 8, ld, 0, T, stream=st)
    out = torch.empty((T, A * D), dtype=torch.float64, device="cuda")
    ctx.stage_read_dev(0, out.data_ptr(), A * D, st)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), wide.cpu().numpy()[:, off:off + A * D])


@pytest.mark.parametrize("T,A,D,ld,off,lo", [(130, 7, 3, 28, 4, 0), (64, 64, 2, 128, 0, 0), (257, 33, 3, 100, 0, 0), (1, 1, 2, 4, 0, 0),
                                              (200, 129, 1, 132, 0, 0), (131, 10, 3, 32, 0, 64), (90, 9, 3, 28, 0, 33)])
def test_relayout_float32_16_byte_path(ctx, T, A, D, ld, off, lo):
    """float32 frame-major rows into a float32 device slab ("stage_device_f32"): row stride a multiple of
    4 and a 16-byte aligned base take k_relayout_wide32 (16-byte accesses on both sides, two rows of
    a pair per store): odd column and frame counts, partial tiles, a column block of a wider tensor,
    frames committed in two pieces (an odd first frame of the second piece falls back to the generic
    kernel)."""
    import torch

    rng = np.random.default_rng(T + 11 * A)
    wide = torch.from_numpy(rng.standard_normal((T, ld)).astype(np.float32)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    ctx.set_option("stage_device_f32", 1)
    try:
        ctx.stage_alloc_device(T, A, D, n_slabs=1)
        assert (wide.data_ptr() + off * 4) % 16 == 0 and ld % 4 == 0
        for a, b in ((0, lo), (lo, T)):
            if b > a:
                ctx.stage_commit_dev(0, wide.data_ptr() + off * 4 + a * ld * 4, ld, a, b, dtype="float32", stream=st)
        out = torch.empty((T, A * D), dtype=torch.float64, device="cuda")
        ctx.stage_read_dev(0, out.data_ptr(), A * D, st)
        torch.cuda.synchronize()
    finally:
        ctx.set_option("stage_device_f32", 0)
    assert np.array_equal(out.cpu().numpy(), wide.cpu().numpy()[:, off:off + A * D].astype(np.float64))


def test_clock_probe(ctx):
    """ta_clock_probe: the stamped build of the lag-sum forward kernel reports a plausible shader
    clock and cycle count on the staged slab; plans without a stamped build say so."""
    from transport_analysis_amd import _lib

    ctx.stage_alloc_device(10000, 600, 3, n_slabs=1)
    ctx.stage_synth(0, 99, 0, 1800)
    p = ctx.clock_probe(3)
    assert 300.0 < p["mhz"] < 2600.0 and p["cycles_per_unit_pass"] > 1000 and p["ms_per_launch"] > 0
    ctx.stage_alloc_device(700, 50, 3, n_slabs=1)  # R0 = 2: no stamped build
    ctx.stage_synth(0, 99, 0, 150)
    with pytest.raises(_lib.TAError, match="clock probe"):
        ctx.clock_probe(1)


@pytest.mark.parametrize("T,A,D,off,tot", [(50, 7, 3, 0, 21), (33, 5, 3, 6, 40), (128, 64, 1, 10, 100)])
def test_synthetic_generator_bit_exact(ctx, T, A, D, off, tot):
    """ta_stage_synth == oracle.synth (NumPy) bit for bit: the CPU baseline and every GPU shard
    see the same tensor."""
    import torch

    from oracle import synth

    ctx.stage_alloc_device(T, A, D, n_slabs=1)
    st = torch.cuda.current_stream().cuda_stream
    ctx.stage_synth(0, 20250824 + 3, off, tot, st)
    out = torch.empty((T, A * D), dtype=torch.float64, device="cuda")
    ctx.stage_read_dev(0, out.data_ptr(), A * D, st)
    torch.cuda.synchronize()
    want = synth.synthetic_block(20250824 + 3, T, tot, off, off + A * D)
    assert np.array_equal(out.cpu().numpy(), want)
    assert abs(want.mean()) < 0.2 and 0.7 < want.std() < 1.3


@pytest.mark.parametrize("fft", [True, False])
def test_classes_float32_and_float64_staging_bit_equal(fft):
    """The drop-in classes stage float32 when the trajectory hands out float32 (MDAnalysis
    does): results are bit-identical to float64 staging (the reference's slab dtype,
    velocityautocorr.py:150-152), for VACF and Helfand."""
    from oracle import numpy_oracle as orc
    from transport_analysis_amd import VelocityAutocorr, ViscosityHelfand
    from transport_analysis_amd._mini_mda import ArrayUniverse

    T, A = 600, 9
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=99)
    u = ArrayUniverse(positions=x, velocities=v, masses=m, dimensions=[60, 60, 60, 90, 90, 90])
    a32 = VelocityAutocorr(u.atoms, fft=fft, device_float32=False).run()
    a64 = VelocityAutocorr(u.atoms, fft=fft, stage_dtype=np.float64).run()
    assert a32._velocities.dtype == np.float32 and a64._velocities.dtype == np.float64
    assert np.array_equal(a32.results.timeseries, a64.results.timeseries)
    assert np.array_equal(a32.results.vacf_by_particle, a64.results.vacf_by_particle)
    # default: float32 staging stays float32 on the device (fft, 513 ... 10240 frames): same values,
    # a separately compiled kernel -- equal to rounding
    auto = VelocityAutocorr(u.atoms, fft=fft).run()
    assert scale_rel_err(auto.results.timeseries, a64.results.timeseries) < 1e-14
    assert scale_rel_err(auto.results.vacf_by_particle, a64.results.vacf_by_particle) < 1e-14
    if fft:
        h32 = ViscosityHelfand(u.atoms).run()
        h64 = ViscosityHelfand(u.atoms, stage_dtype=np.float64).run()
        assert h32._positions.dtype == np.float32
        assert np.array_equal(h32.results.timeseries, h64.results.timeseries)
        assert np.array_equal(h32.results.visc_by_particle, h64.results.visc_by_particle)


def test_helfand_config4_full_per_gpu_share(ctx):
    """BASELINE configs[4] at ONE GPU's full share: 20000 frames x 25000 atoms x 3, float32 path
    on float32 device slabs,
    staged pair-major (viscosity.py:210-226 is the loop replaced).  Size-independent checks:
    lag 0 exactly 0; selected lags against float64 slab differences over the whole block; the
    lag sums of the two half blocks add up to the whole's."""
    import torch

    T, A, D = 20000, 25000, 3
    st = torch.cuda.current_stream().cuda_stream
    m = torch.tensor([15.999, 1.008, 1.008], dtype=torch.float64, device="cuda").repeat((A + 2) // 3)[:A].contiguous()

    def stage(lo, hi):
        n = hi - lo
        ctx.stage_alloc_device(T, n, D, n_slabs=2)
        ctx.stage_synth(0, 20250824 + 5, lo * D, A * D, st)
        fm = torch.empty((T, n * D), dtype=torch.float64, device="cuda")
        ctx.stage_read_dev(0, fm.data_ptr(), n * D, st)
        xm = 30.0 + 0.002 * torch.cumsum(fm, dim=0)
        ctx.stage_commit_dev(1, xm.data_ptr(), n * D, 0, T, stream=st)
        ctx.stage_read_dev(1, xm.data_ptr(), n * D, st)  # as stored: rounded to float32
        torch.cuda.synchronize()
        return fm, xm

    ctx.set_option("direct_f32", 1)
    ctx.set_option("stage_device_f32", 1)  # float32 device slabs: 2 x 6 GB instead of 2 x 12 GB
    try:
        v, x = stage(0, A)
        whole = torch.zeros(T, dtype=torch.float64, device="cuda")
        ctx.helfand_msd_staged(m.data_ptr(), 1.0, whole.data_ptr(), 0, A, st)
        torch.cuda.synchronize()
        assert float(whole[0].item()) == 0.0
        P = (m[None, :, None] * v.view(T, A, D)) * x.view(T, A, D)
        scale = None
        for k in (1, 7, 1000, T // 2, T - 1):
            w = float(((P[: T - k] - P[k:]) ** 2).sum().item()) / (T - k) / D
            scale = scale or w
            assert abs(float(whole[k].item()) - w) < TOL_F32 * max(w, scale), (k, float(whole[k].item()), w)
        del v, x, P
        torch.cuda.empty_cache()
        parts = torch.zeros(T, dtype=torch.float64, device="cuda")
        for lo, hi in ((0, A // 2), (A // 2, A)):
            v, x = stage(lo, hi)
            del v, x
            part = torch.zeros(T, dtype=torch.float64, device="cuda")
            ctx.helfand_msd_staged(m[lo:hi].contiguous().data_ptr(), 1.0, part.data_ptr(), 0, hi - lo, st)
            torch.cuda.synchronize()
            parts += part
        assert float((parts - whole).abs().max().item()) < TOL_F32 * float(whole.abs().max().item())
    finally:
        ctx.set_option("direct_f32", 0)
        ctx.set_option("stage_device_f32", 0)
        ctx.stage_free()
        ctx.trim()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("T,A,D", [(600, 19, 1), (1000, 37, 3), (1030, 24, 2), (2100, 21, 3), (2560, 18, 1),
                                   (3000, 17, 3), (4200, 16, 2), (6000, 13, 3), (8192, 12, 1), (10000, 11, 3),
                                   (12000, 9, 3), (20000, 11, 1), (20480, 8, 2), (33000, 5, 3), (50000, 4, 1),
                                   (3300, 14, 3), (4500, 12, 2), (6500, 11, 3), (9000, 10, 1)])
@pytest.mark.parametrize("spec_atoms", [0, 6])
def test_vacf_by_particle_blocks_of_atoms(ctx, T, A, D, spec_atoms):
    """The by-particle FFT evaluation (forward kernel leaving per-atom power spectra, inverse
    kernel; blocks of `spec_atoms` atoms) for every first-stage radix, outer radices 1, 2, 4, 8 and
    every column-unit kind (D = 1: single columns of either half of a pair; D = 3: an aligned pair
    plus a single column, alternating), against the oracle.  velocityautocorr.py:196-215."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=4000 + T)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    try:
        ctx.set_option("bp_spec_atoms", spec_atoms)
        ts, bp = run_vacf(ctx, v, True, True)
    finally:
        ctx.set_option("bp_spec_atoms", 0)
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL


@pytest.mark.parametrize("T,A", [(10000, 100000), (20000, 25000)])
def test_vacf_by_particle_config2_full_size(ctx, T, A):
    """The reference's default output at BASELINE configs[2]'s size on one GPU: 10000 x 100000 x 3
    with the (n_frames, n_atoms) by-particle array (8 GB), and at configs[4]'s per-GPU shape
    (20000 frames: outer radix 2, 4 GB).  The mean over atoms of the array is the lag-sum path's
    timeseries (velocityautocorr.py:214), blocks of atoms agree with the oracle, and every row of
    the array is written."""
    import torch

    from oracle import numpy_oracle as orc
    from oracle import synth

    D = 3
    st = torch.cuda.current_stream().cuda_stream
    ctx.stage_alloc_device(T, A, D, n_slabs=1)
    ctx.stage_synth(0, 20250824 + 3, 0, A * D, st)
    bp = torch.full((T, A), float("nan"), dtype=torch.float64, device="cuda")
    lag_bp = torch.zeros(T, dtype=torch.float64, device="cuda")
    lag_ts = torch.zeros(T, dtype=torch.float64, device="cuda")
    ctx.vacf_fft_staged(lag_bp.data_ptr(), bp.data_ptr(), A, st)
    ctx.vacf_fft_staged(lag_ts.data_ptr(), 0, 0, st)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(bp).any().item())
    scale = float(lag_ts.abs().max().item())
    assert float((lag_bp - lag_ts).abs().max().item()) < TOL * scale
    assert float((bp.sum(dim=1) - lag_ts).abs().max().item()) < TOL * scale
    for lo in (0, A // 2 - 1, A - 4):
        v = synth.synthetic_block(20250824 + 3, T, A * D, lo * D, (lo + 4) * D).reshape(T, 4, D)
        want, _ = orc.vacf_fft_batched(v)
        got = bp[:, lo:lo + 4].cpu().numpy()
        assert scale_rel_err(got, want) < TOL
    del bp
    ctx.stage_free()
    ctx.trim()
    torch.cuda.empty_cache()


def test_blocked_by_particle_waits_for_the_commit_worker(ctx):
    """A host-facing call whose atoms go in blocks does not pass through staged_entry: it joins the
    commit worker itself.  A commit large enough to still be page-locking its chunks and issuing its
    copies when the compute call arrives (200 MB, several chunks) must be complete in every block."""
    from oracle import numpy_oracle as orc

    T, A, D = 4096, 2048, 3
    rng = np.random.default_rng(17)
    v = rng.standard_normal((T, A, D))
    ctx.set_option("bp_block", 256)
    try:
        ts, bp = run_vacf(ctx, v, True, True)
    finally:
        ctx.set_option("bp_block", 0)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    assert scale_rel_err(ts, want_ts) < TOL
    assert scale_rel_err(bp, want_bp) < TOL


@pytest.mark.parametrize("mode,D", [("fft", 3), ("fft", 1), ("direct", 2), ("helfand", 3)])
def test_host_path_by_particle_in_atom_blocks(ctx, mode, D):
    """Host-facing calls with a by-particle array process atoms in blocks (copy of block c
    under the compute of block c + 1): same results as one block, block edges on pair
    boundaries for every dim."""
    from oracle import numpy_oracle as orc

    T, A = 700, 200
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=5)
    v, x = v[:, :, :D], x[:, :, :D]
    ctx.set_option("bp_block", 64)
    try:
        if mode == "helfand":
            ts, bp = run_helfand(ctx, v, x, m, 0.37, True)
            want_bp, want_ts = orc.helfand(v, x, m, np.ones(T), temp_avg=1.0, boltzmann=0.5)
            want_bp, want_ts = want_bp * 0.37, want_ts * 0.37
        else:
            ts, bp = run_vacf(ctx, v, mode == "fft", True)
            want_bp, want_ts = orc.vacf_fft_batched(v)
    finally:
        ctx.set_option("bp_block", 0)
    assert bp.shape == (T, A)
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL


@pytest.mark.parametrize("T,A,D", [(513, 700, 3), (1100, 1301, 1), (2600, 333, 2), (5200, 257, 3), (9000, 301, 3),
                                   (10240, 130, 1)])
def test_vacf_fft_many_units_per_workgroup(ctx, T, A, D):
    """Enough atoms that the persistent workgroups walk several pairs / atoms each (couples of the
    pass-split kernel, atom loop of the by-particle kernel), odd pair counts, every dim."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=4000 + T)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    ts, bp = run_vacf(ctx, v, True, True)
    assert scale_rel_err(bp, want_bp) < TOL
    assert scale_rel_err(ts, want_ts) < TOL
    ts2, _ = run_vacf(ctx, v, True, False)
    assert scale_rel_err(ts2, want_ts) < TOL


# ------------------------------------------------------------------ float32 device slabs
@pytest.mark.parametrize("T,A,D", [(300, 7, 3), (1000, 33, 3), (2500, 5, 2), (17001, 2, 1)])
def test_float32_device_slabs_change_nothing(ctx, T, A, D):
    """"stage_device_f32": float32 host slabs kept as float32 on the device (half the footprint,
    BASELINE configs[4]'s float32 path).  The float32 direct correlators read them as they are and
    every other evaluation widens them first: all results are bit-equal to those from float64
    device slabs of the same (float32-representable) values.  viscosity.py:201-233,
    velocityautocorr.py:208-238."""
    from oracle import numpy_oracle as orc

    v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=77 + T)
    v, x = v.astype(np.float32), x.astype(np.float32)
    scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)

    def run_all(dev_f32):
        out = {}
        ctx.set_option("stage_device_f32", dev_f32)
        sv, sx = ctx.stage_alloc(T, A, D, n_slabs=2, dtype=np.float32)
        sv[...] = v
        sx[...] = x
        ctx.stage_commit(0, T)
        for f32 in (1, 0):
            ctx.set_option("direct_f32", f32)
            out[f"helfand{f32}"] = ctx.helfand_msd(m, scale, by_particle=True)
            out[f"direct{f32}"] = ctx.vacf_direct(by_particle=True)
        out["fft"] = ctx.vacf_fft(by_particle=True)
        ctx.set_option("helfand_fft", 1)
        out["helfand_fft"] = ctx.helfand_msd(m, scale, by_particle=False)
        ctx.set_option("helfand_fft", 0)
        return out

    try:
        a, b = run_all(1), run_all(0)
    finally:
        for key in ("stage_device_f32", "direct_f32", "helfand_fft"):
            ctx.set_option(key, 0)
    for key in a:
        if key == "fft" and 512 < T <= 10240:
            # these plans read the float32 rows themselves (no widening pass): the same values through
            # a separately compiled kernel -- equal to rounding (FMA contraction may differ by plan)
            assert scale_rel_err(a[key][0], b[key][0]) < 1e-14, key
            assert scale_rel_err(a[key][1], b[key][1]) < 1e-14, key
            continue
        assert np.array_equal(a[key][0], b[key][0]), key
        if a[key][1] is not None:
            assert np.array_equal(a[key][1], b[key][1]), key
    want_bp, want_ts = orc.helfand(v.astype(np.float64), x.astype(np.float64), m, vol, 300.0) if T <= 2500 else (None, None)
    if want_ts is not None:
        assert scale_rel_err(a["helfand1"][0], want_ts) < TOL_F32
        assert scale_rel_err(a["helfand0"][0], want_ts) < TOL


@pytest.mark.parametrize("T,A,D", [(513, 5, 3), (1024, 3, 1), (1500, 9, 2), (2048, 7, 3), (2600, 4, 3), (3100, 3, 3),
                                   (3600, 2, 3), (4100, 3, 3), (4700, 2, 2), (5200, 3, 3), (6200, 2, 3), (7200, 2, 3),
                                   (8200, 2, 3), (9300, 3, 3), (10240, 3, 3), (10000, 70, 3)])
def test_fft_reads_float32_device_slabs(ctx, T, A, D):
    """Every FFT plan without an outer radix reads float32 device slabs as they are (8-byte rows,
    widened exactly in the first stage): lag sums and the by-particle array against the oracle at
    the float64 bar -- complex units, single real columns of both parities, odd column counts."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=5 * T + A).astype(np.float32)
    try:
        ctx.set_option("stage_device_f32", 1)
        (slab,) = ctx.stage_alloc(T, A, D, n_slabs=1, dtype=np.float32)
        slab[...] = v
        ctx.stage_commit(0, T)
        ts, bp = ctx.vacf_fft(by_particle=True)
        ts2, _ = ctx.vacf_fft(by_particle=False)
    finally:
        ctx.set_option("stage_device_f32", 0)
    want_bp, want_ts = orc.vacf_fft_batched(v.astype(np.float64))
    assert scale_rel_err(ts, want_ts) < TOL and scale_rel_err(ts2, want_ts) < TOL
    assert scale_rel_err(bp, want_bp) < TOL


def test_float32_device_slabs_staged_api(ctx):
    """Device-only float32 slabs: ta_stage_synth rounds once to float32, ta_stage_commit_dev /
    ta_stage_read_dev round-trip float32 data exactly, and the staged float32 Helfand evaluation
    equals the one on float64 slabs holding the same rounded values."""
    import torch

    T, A, D = 4000, 300, 3
    st = torch.cuda.current_stream().cuda_stream
    m = torch.linspace(1.0, 16.0, A, dtype=torch.float64, device="cuda")
    res = []
    try:
        ctx.set_option("direct_f32", 1)
        for dev_f32 in (1, 0):
            ctx.set_option("stage_device_f32", dev_f32)
            ctx.stage_alloc_device(T, A, D, n_slabs=2)
            ctx.stage_synth(0, 99, 0, A * D, st)
            fm = torch.empty((T, A * D), dtype=torch.float64, device="cuda")
            ctx.stage_read_dev(0, fm.data_ptr(), A * D, st)
            torch.cuda.synchronize()
            if dev_f32:
                assert bool((fm == fm.float().double()).all().item())  # float32 values
                v32 = fm.clone()
            else:
                ctx.stage_commit_dev(0, v32.data_ptr(), A * D, 0, T, stream=st)  # same rounded values
            xm = (30.0 + 0.002 * torch.cumsum(v32, dim=0)).float()
            ctx.stage_commit_dev(1, xm.data_ptr(), A * D, 0, T, dtype=np.float32, stream=st)
            back = torch.empty((T, A * D), dtype=torch.float64, device="cuda")
            ctx.stage_read_dev(1, back.data_ptr(), A * D, st)
            torch.cuda.synchronize()
            assert bool((back == xm.double()).all().item())
            lag = torch.zeros(T, dtype=torch.float64, device="cuda")
            bp = torch.zeros((T, A), dtype=torch.float64, device="cuda")
            ctx.helfand_msd_staged(m.data_ptr(), 1.0, lag.data_ptr(), bp.data_ptr(), A, st)
            torch.cuda.synchronize()
            res.append((lag.cpu().numpy(), bp.cpu().numpy()))
    finally:
        ctx.set_option("stage_device_f32", 0)
        ctx.set_option("direct_f32", 0)
        ctx.stage_free()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert float(np.max(np.abs(res[0][0]))) > 0


@pytest.mark.parametrize("T,A", [(1000, 400000), (1500, 260000), (2500, 160000), (3000, 130000), (6000, 66000)])
def test_vacf_fft_small_plans_at_scale(ctx, T, A):
    """The plans below 8 sub-series (R0 = 2, 3, 5, 6: workgroups of 2 or 4 waves, several per
    compute unit) and R0 = 12 on ~10 GB each: the lag sums of the whole block equal the sum over
    its two halves (size-independent linearity), and selected lags equal plain torch reductions
    over the staged tensor.  velocityautocorr.py:208-215."""
    import torch

    D = 3
    st = torch.cuda.current_stream().cuda_stream
    seed = 20250824 + 11

    def lagsums(lo, hi, keep=False):
        ctx.stage_alloc_device(T, hi - lo, D, n_slabs=1)
        ctx.stage_synth(0, seed, lo * D, A * D, st)
        out = torch.zeros(T, dtype=torch.float64, device="cuda")
        ctx.vacf_fft_staged(out.data_ptr(), 0, 0, st)
        fm = None
        if keep:
            fm = torch.empty((T, (hi - lo) * D), dtype=torch.float64, device="cuda")
            ctx.stage_read_dev(0, fm.data_ptr(), (hi - lo) * D, st)
        torch.cuda.synchronize()
        return out, fm

    try:
        whole, fm = lagsums(0, A, keep=True)
        scale = float((fm * fm).sum().item()) / T
        for k in (0, 1, T // 3, T - 1):
            ref = float((fm[: T - k] * fm[k:]).sum().item()) / (T - k)
            assert abs(float(whole[k].item()) - ref) < TOL * scale, (k, float(whole[k].item()), ref)
        del fm
        torch.cuda.empty_cache()
        cut = A // 2 + 1  # an odd number of columns in the first half
        a, _ = lagsums(0, cut)
        b, _ = lagsums(cut, A)
        assert float((a + b - whole).abs().max().item()) < TOL * scale
    finally:
        ctx.stage_free()
        ctx.trim()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("T,A,D", [(1000, 9, 3), (2500, 10, 2), (6000, 9, 3), (10000, 8, 1)])
def test_vacf_by_particle_prefetch_depths(ctx, T, A, D):
    """Every spectrum-prefetch depth of the inverse kernel ("bp_prefetch" 0..3; plans with 1, 2 and 3
    sub-series per wave) gives the same per-particle array bit for bit: the depth only moves
    loads."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=5000 + T)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    got = []
    try:
        for pf in (0, 1, 2, 3):
            ctx.set_option("bp_prefetch", pf)
            got.append(run_vacf(ctx, v, True, True))
    finally:
        ctx.set_option("bp_prefetch", 2)
    for ts, bp in got:
        assert np.array_equal(bp, got[0][1]) and np.array_equal(ts, got[0][0])
    assert scale_rel_err(got[0][1], want_bp) < TOL
    assert scale_rel_err(got[0][0], want_ts) < TOL


# ------------------------------------------- round 3: boundary additions
def test_kernel_timeline_sums_to_the_call(ctx):
    """ta_kernel_timeline ("timeline" option): per-kernel device time of the last call, named,
    inside the call's total (events on the launch stream); off by default."""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(3000, 300, 3, seed=3)
    run_vacf(ctx, v, True, True)
    assert ctx.kernel_timeline() == []  # option off: nothing recorded
    ctx.set_option("timeline", 1)
    try:
        ts, bp = ctx.vacf_fft(by_particle=True)
        tl = ctx.kernel_timeline()
        names = [n for n, _ in tl]
        assert names == ["k_wsplit_accum", "k_winverse", "k_bp_transpose", "k_sum_partials"]
        total, _ = ctx.last_timing()
        # the marks sit inside the call's own start / end events
        assert 0.5 * total < sum(ms for _, ms in tl) <= 1.02 * total
        ts2, _ = ctx.vacf_direct(by_particle=False)
        assert [n for n, _ in ctx.kernel_timeline()] == ["k_band_bp_vacf"]  # lag sums alone: matrix cores
        ctx.vacf_direct(by_particle=True)
        assert [n for n, _ in ctx.kernel_timeline()] == ["k_band_bp_vacf", "k_bp_transpose", "k_sum_partials"]
        ctx.set_option("direct_mfma", 0)
        ctx.vacf_direct(by_particle=True)
        assert [n for n, _ in ctx.kernel_timeline()] == ["memset", "k_direct", "k_sum_partials", "k_bp_transpose"]
    finally:
        ctx.set_option("direct_mfma", 1)
        ctx.set_option("timeline", 0)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    assert scale_rel_err(ts, want_ts) < TOL and scale_rel_err(bp, want_bp) < TOL


def test_direct_forms_by_trajectory_length(ctx):
    """"direct_mfma" 1 (the default) picks the form by n_frames: the matrix-core kernels pay a ring fill and an epilogue
    per particle and lag group, and the vector kernels pack several particles into a wave under ~640 frames ("direct_subwave";
    k_mid, test_mid_length_kernel), so vector kernels run the windowed VACF up to 512 frames (k_mid from 97), Einstein-Helfand
    float64 up to 351 (k_mid from 97 to 128) and its float32 option up to 447 — thresholds from
    profiles/r06_direct_mid_sweep.txt (up to 64 frames: k_short, test_short_trajectory_kernels).
    Whatever is picked agrees with the forced forms, whole-wave column groups ("direct_subwave" 0) agree with the packed
    ones, and "direct_mfma" 2 is rejected."""
    from oracle import numpy_oracle as orc
    from transport_analysis_amd import _lib

    def names():
        return [n for n, _ in ctx.kernel_timeline() if n.startswith(("k_band", "k_direct", "k_mid"))]

    with pytest.raises(_lib.TAError, match="tools/band"):
        ctx.set_option("direct_mfma", 2)
    ctx.set_option("timeline", 1)
    try:
        for T, vacf, helf, helf32 in ((80, "k_direct", "k_direct", "k_direct"), (100, "k_mid", "k_mid", "k_direct"),
                                      (351, "k_mid", "k_direct", "k_direct"), (352, "k_mid", "k_band_bp_helf", "k_direct"),
                                      (512, "k_mid", "k_band_bp_helf", "k_band32_tp"),
                                      (513, "k_band_bp_vacf", "k_band_bp_helf", "k_band32_tp"),
                                      (1600, "k_band_bp_vacf", "k_band_bp_helf", "k_band32_tp")):
            v, x, m, vol = orc.synthetic_helfand(T, 37, 3, seed=41 + T)
            ts_d, bp_d = run_vacf(ctx, v, False, True)
            assert names() == [vacf], (T, names())
            ts_l, _ = ctx.vacf_direct(by_particle=False)
            assert names() == [vacf], (T, names())
            hs_b, hb = run_helfand(ctx, v, x, m, 1.0, True)
            assert names() == [helf], (T, names())
            hs_l, _ = ctx.helfand_msd(m, 1.0, by_particle=False)
            assert names() == [helf], (T, names())
            ctx.set_option("direct_f32", 1)
            fs_b, fb = ctx.helfand_msd(m, 1.0, by_particle=True)
            assert names() == [helf32], (T, names())
            fs_l, _ = ctx.helfand_msd(m, 1.0, by_particle=False)
            assert names() == [helf32], (T, names())
            ctx.set_option("direct_f32", 0)
            if T <= 640:  # the same kernel with a whole wave per column (the lags per chunk may differ: 8 or 10)
                ctx.set_option("direct_mfma", 0)
                packed = ctx.helfand_msd(m, 1.0, by_particle=True)
                ctx.set_option("direct_subwave", 0)
                whole = ctx.helfand_msd(m, 1.0, by_particle=True)
                assert scale_rel_err(packed[1], whole[1]) < 1e-12 and scale_rel_err(packed[0], whole[0]) < 1e-12
                ctx.set_option("direct_subwave", 1)
                ctx.set_option("direct_mfma", 1)
            for form in (3, 0):
                ctx.set_option("direct_mfma", form)
                assert scale_rel_err(ctx.helfand_msd(m, 1.0, by_particle=False)[0], hs_l) < 1e-11
                assert scale_rel_err(ctx.helfand_msd(m, 1.0, by_particle=True)[1], hb) < 1e-11
                ctx.set_option("direct_f32", 1)
                assert scale_rel_err(ctx.helfand_msd(m, 1.0, by_particle=False)[0], fs_l) < TOL_F32
                ctx.set_option("direct_f32", 0)
                run_vacf(ctx, v, False, False)
                assert scale_rel_err(ctx.vacf_direct(by_particle=False)[0], ts_l) < 1e-12
                assert scale_rel_err(ctx.vacf_direct(by_particle=True)[1], bp_d) < 1e-12
                run_helfand(ctx, v, x, m, 1.0, False)  # (both slabs back for the next round)
            ctx.set_option("direct_mfma", 1)
    finally:
        ctx.set_option("direct_mfma", 1)
        ctx.set_option("direct_subwave", 1)
        ctx.set_option("direct_f32", 0)
        ctx.set_option("timeline", 0)


def test_pinned_result_home(ctx):
    """The by-particle array of a host-facing call lives in ta_host_alloc memory: written through
    a caller-provided `out`, outliving the context's slabs, freed with its last view."""
    import gc
    import weakref

    from oracle import numpy_oracle as orc
    from transport_analysis_amd import _lib

    T, A = 700, 130
    v = orc.synthetic_velocities(T, A, 3, seed=8)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    home = _lib.pinned_empty((T, A))
    assert home.shape == (T, A) and home.dtype == np.float64 and home.flags.c_contiguous
    (slab,) = ctx.stage_alloc(T, A, 3)
    slab[...] = v
    ctx.stage_commit(0, T)
    ts, bp = ctx.vacf_fft(out=home)
    assert bp is home and scale_rel_err(bp, want_bp) < TOL and scale_rel_err(ts, want_ts) < TOL
    ts, bp2 = ctx.vacf_direct(by_particle=True)  # allocated by the binding: pinned as well
    assert scale_rel_err(bp2, want_bp) < TOL
    with pytest.raises(ValueError):
        ctx.vacf_fft(out=np.empty((T, A + 1)))
    ctx.stage_free()
    view = bp2[5:9, 3:7]
    root = bp2
    while isinstance(root, np.ndarray):  # reshape -> frombuffer -> the ctypes block
        root = root.base
    owner = weakref.ref(root._owner)
    del bp2, root
    gc.collect()
    assert owner() is not None  # a view keeps the block alive
    assert scale_rel_err(view, want_bp[5:9, 3:7]) < TOL
    del view
    gc.collect()
    assert owner() is None
    z = _lib.pinned_empty((0, 4))
    assert z.shape == (0, 4)


def test_stage_commit_through_both_landing_buffers(ctx):
    """ta_stage_commit with more than two 64 MiB pieces per call and frames committed in several
    calls: pieces alternate between two landing buffers, the transposition runs on its own stream;
    every frame must land where the one-piece path puts it (read back bit for bit)."""
    import torch

    T, A, D = 120, 90000, 3  # 2.16 MB per float64 frame: 31 frames per piece
    rng = np.random.default_rng(17)
    (slab,) = ctx.stage_alloc(T, A, D)
    slab[...] = rng.standard_normal((T, A, D))
    keep = slab.copy()
    ctx.stage_commit(0, 70)   # three pieces
    ctx.stage_commit(70, 75)  # one small piece right behind
    ctx.stage_commit(75, T)   # two pieces
    back = torch.empty((T, A * D), dtype=torch.float64, device="cuda")
    ctx.stage_read_dev(0, back.data_ptr(), A * D, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(back.cpu().numpy().reshape(T, A, D), keep)
    ts, _ = ctx.vacf_fft(by_particle=False)
    ref = (keep[:1] * keep[:1]).sum() / A  # lag T-1 has one term: <v0 . v_{T-1}>
    assert ts[0] == pytest.approx((keep * keep).sum() / (T * A), rel=1e-12)
    assert ts[T - 1] == pytest.approx((keep[0] * keep[T - 1]).sum() / A, rel=1e-10, abs=1e-12 * abs(ref))
    ctx.stage_free()


# ------------------------------------------------------------------ native per-frame staging
@pytest.mark.parametrize("src_dtype", [np.float32, np.float64])
@pytest.mark.parametrize("stage_dtype", [np.float32, np.float64])
def test_stage_frame_native_equals_numpy_fill(ctx, src_dtype, stage_dtype):
    """ta_stage_frame (the reference's `slab[i] = atomgroup.velocities[:, dim]`, velocityautocorr.py:192-194,
    as one native pass from the Timestep's array): every dim_type's columns, a block of consecutive atoms
    and a gather by index, float32 / float64 on both sides, a padded source row — bit-equal to the NumPy
    expression, frame by frame, through several threads' chunks (70000 atoms) and below one chunk."""
    from transport_analysis_amd import _lib

    rng = np.random.default_rng(11)
    for n_univ, ld in ((70001, 3), (300, 4)):
        wide = rng.standard_normal((n_univ, ld)).astype(src_dtype)
        frame_arr = wide[:, :3] if ld == 4 else wide  # (ld 4: rows padded, as a view of a wider array)
        for cols in ([0], [1], [2], [0, 1], [0, 2], [1, 2], [0, 1, 2]):
            for ix in (np.arange(5, n_univ - 7), rng.permutation(n_univ)[: n_univ // 2], np.array([n_univ - 1])):
                n = len(ix)
                (slab,) = ctx.stage_alloc(3, n, len(cols), n_slabs=1, dtype=stage_dtype)
                src = _lib.frame_source(frame_arr)
                assert src is not None and src[2] == ld
                ctx.stage_frame(0, 1, src, cols, _lib.atom_rows(ix))
                want = frame_arr[ix][:, cols].astype(stage_dtype)
                assert np.array_equal(slab[1], want)
                assert not slab[0].any() and not slab[2].any()  # only the frame asked for
    assert _lib.frame_source(np.zeros((4, 3), dtype=np.float32)[:, ::-1]) is None  # negative stride: NumPy path
    assert _lib.frame_source(np.zeros((4, 3), dtype=np.int32)) is None
    with pytest.raises(_lib.TAError):  # a slab of 2 columns cannot take 3
        (slab,) = ctx.stage_alloc(2, 5, 2, n_slabs=1)
        ctx.stage_frame(0, 0, _lib.frame_source(np.zeros((9, 3))), [0, 1, 2], _lib.atom_rows(np.arange(5)))
    ctx.stage_free()


@pytest.mark.parametrize("subset", [False, True])
def test_classes_native_and_numpy_staging_bit_equal(subset, monkeypatch):
    """Both classes stage every frame through ta_stage_frame (from ts.velocities / ts.positions, gathering
    by the group's atom indices) and, with $TA_AMD_NATIVE_STAGING=0, through the NumPy views as before: the
    staged slabs and every result are bit-equal; a sub-group that is not a block of consecutive atoms takes
    the index path; devices=[0, 0] fills both members' slabs from one call."""
    from oracle import numpy_oracle as orc
    from transport_analysis_amd import VelocityAutocorr, ViscosityHelfand
    from transport_analysis_amd._mini_mda import ArrayUniverse

    T, A = 300, 41
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=5)
    u = ArrayUniverse(positions=x.astype(np.float32), velocities=v.astype(np.float32), masses=m,
                      dimensions=[60, 60, 60, 90, 90, 90])
    ag = u.atoms[np.array([3, 4, 9, 1, 30, 17, 18, 19])] if subset else u.atoms
    out = {}
    for native in ("1", "0"):
        monkeypatch.setenv("TA_AMD_NATIVE_STAGING", native)
        a = VelocityAutocorr(ag, dim_type="xz", fft=True).run()
        assert (a._rows is not None) == (native == "1")
        h = ViscosityHelfand(ag).run()
        g = VelocityAutocorr(ag, fft=False, devices=[0, 0]).run()
        out[native] = (np.array(a._velocities), a.results.timeseries, a.results.vacf_by_particle,
                       np.array(h._velocities), np.array(h._positions), h.results.timeseries, h.results.visc_by_particle,
                       np.concatenate([np.array(s) for s in g._velocities if s is not None], axis=1), g.results.timeseries)
    for got, want in zip(out["1"], out["0"]):
        assert np.array_equal(got, want)
    if subset:
        assert out["1"][0].shape == (T, 8, 2)


def test_helfand_matrix_cores_do_not_depend_on_the_unit(ctx):
    """The float64 matrix-core Helfand lag sums for P scaled by 1e-12 ... 1e+8 (velocities and positions in other
    units): the result scales with the square, to the float64 bar.  (The norm slot's exact 1 must never be
    added to a squared norm: beside 1, a norm of 1e-20 has no digits left.)"""
    from oracle import numpy_oracle as orc

    T, A = 700, 13
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=98)
    base, _ = run_helfand(ctx, v, x, m, 1.0, False)
    ctx.set_option("timeline", 1)
    try:
        ctx.set_option("direct_mfma", 3)  # the time-packed form (the default from 896 frames)
        base3, _ = run_helfand(ctx, v, x, m, 1.0, False)
        assert scale_rel_err(base3, base) < 1e-12
        for sv, sx in ((1e-6, 1e-6), (1e4, 1e4), (1e-3, 1.0)):
            got, _ = run_helfand(ctx, v * sv, x * sx, m, 1.0, False)
            assert [n for n, _ in ctx.kernel_timeline()] == ["k_helfand_product", "k_band_bp_helf"]
            assert scale_rel_err(got / (sv * sx) ** 2, base) < TOL, (sv, sx)
    finally:
        ctx.set_option("direct_mfma", 1)
        ctx.set_option("timeline", 0)


# ------------------------------------------------------------------ the C boundary and C++ exceptions
@pytest.mark.parametrize("hook,code", [("fail_alloc_after", -2), ("fail_throw_after", -3)])  # TA_E_NOMEM, TA_E_HIP
@pytest.mark.parametrize("entry", ["vacf_fft", "vacf_fft_bp", "vacf_direct", "helfand"])
def test_exception_inside_the_library_becomes_a_status(hook, code, entry):
    """every call returns an int status.  A std::bad_alloc / any other exception thrown inside a
    compute call (the allocation helper throws on request: options fail_alloc_after / fail_throw_after) comes
    back as TA_E_NOMEM / TA_E_HIP with a message, the interpreter lives, and the context works afterwards."""
    from oracle import numpy_oracle as orc
    from transport_analysis_amd import _lib

    c = _lib.Context(0)
    try:
        T, A, D = 700, 9, 3
        v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=11)
        sv, sx = c.stage_alloc(T, A, D, n_slabs=2)
        sv[...] = v
        sx[...] = x
        c.stage_commit(0, T)
        calls = {
            "vacf_fft": lambda: c.vacf_fft(by_particle=False),
            "vacf_fft_bp": lambda: c.vacf_fft(by_particle=True),
            "vacf_direct": lambda: c.vacf_direct(by_particle=True),
            "helfand": lambda: c.helfand_msd(m, 1.0, by_particle=True),
        }
        want = calls[entry]()  # (also makes every later allocation a re-use: the hook counts calls, not bytes)
        c.set_option(hook, 1)
        with pytest.raises(_lib.TAError) as ei:
            calls[entry]()
        assert ei.value.code == code
        assert ("bad_alloc" in str(ei.value)) if code == -2 else ("fail_throw_after" in str(ei.value))
        got = calls[entry]()  # the hook has fired once; the context is intact
        for a, b in zip(want, got):
            if a is not None:
                assert np.array_equal(a, b)
    finally:
        c.close()


def test_exception_inside_a_group_call_becomes_a_status():
    from oracle import numpy_oracle as orc
    from transport_analysis_amd import _lib

    g = _lib.Group([0, 0])
    try:
        T, A, D = 300, 10, 3
        v = orc.synthetic_velocities(T, A, D, seed=5)
        (views,) = g.stage_alloc(T, A, D, n_slabs=1)
        for view, (lo, hi) in zip(views, g.shards):
            if view is not None:
                view[...] = v[:, lo:hi]
        g.stage_commit(0, T)
        want = g.vacf_fft(by_particle=True)
        g.member_context(1).set_option("fail_alloc_after", 1)  # the second member's next allocation-helper call throws
        with pytest.raises(_lib.TAError) as ei:
            g.vacf_fft(by_particle=True)
        assert ei.value.code == -2  # TA_E_NOMEM
        got = g.vacf_fft(by_particle=True)
        assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])
    finally:
        g.close()


@pytest.mark.parametrize("D", [1, 2, 3])
@pytest.mark.parametrize("T", [1, 2, 7, 8, 9, 31, 32, 33, 50, 63, 64])
def test_short_trajectory_kernels(ctx, T, D):
    """Up to 64 frames the by-particle arrays of all three quantities and the O(T^2) lag sums come from k_short
    (short_kernels.hpp: a lane per column, every lag in its registers, the by-particle array written in place): against
    the oracle (velocityautocorr.py:208-215 / :217-238, viscosity.py:201-233), against the kernels it replaces
    ("short_max" 0: the 512-point transforms, the workgroup-per-particle correlators, the matrix-core Helfand form),
    particle by particle and lag by lag; particle counts on both sides of a wave's 64 / 32 / 21 particles and of a
    workgroup's four waves; same bits every launch; 65 frames take the old kernels."""
    from oracle import numpy_oracle as orc

    def names():
        return [n for n, _ in ctx.kernel_timeline() if not n.startswith(("k_sum", "end", "memset"))]

    ctx.set_option("timeline", 1)
    try:
        for A in (1, 2, 20, 21, 22, 64, 65, 85, 257, 1000):
            v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=7000 + 13 * T + A)
            x = x + 50.0
            scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
            want_bp, want_ts = orc.vacf_windowed(v)
            want_hb, want_hs = orc.helfand(v, x, m, vol, 300.0)
            got = {}
            for short in (64, 0):
                ctx.set_option("short_max", short)
                hs, hb = run_helfand(ctx, v, x, m, scale, True)
                if short:
                    assert names() == ["k_short"], names()
                    again = ctx.helfand_msd(m, scale, by_particle=True)
                    assert np.array_equal(again[0], hs) and np.array_equal(again[1], hb)
                hs_l, _ = ctx.helfand_msd(m, scale, by_particle=False)
                ts_d, bp_d = ctx.vacf_direct(by_particle=True)
                ts_dl, _ = ctx.vacf_direct(by_particle=False)
                if short:
                    assert names() == ["k_short"], names()
                ts_f, bp_f = ctx.vacf_fft(by_particle=True)
                assert names() == (["k_short"] if short else ["k_w1_bp", "k_bp_transpose"]), names()
                ts_fl, _ = ctx.vacf_fft(by_particle=False)  # ("short_lags_max": the FFT path's lag sums alone, up to 48 frames)
                assert ("k_short" in names()) == bool(short and T <= 48), names()
                got[short] = (hs, hb, hs_l, ts_d, bp_d, ts_dl, ts_f, bp_f, ts_fl)
                assert hb.shape == (T, A) and not hb[0].any()
                for a, w in ((hb, want_hb), (hs, want_hs), (hs_l, want_hs), (bp_d, want_bp), (ts_d, want_ts),
                             (ts_dl, want_ts), (bp_f, want_bp), (ts_f, want_ts), (ts_fl, want_ts)):
                    assert scale_rel_err(a, w) < TOL, (T, A, D, short)
            for a, b in zip(got[64], got[0]):
                assert scale_rel_err(a, b) < 1e-12, (T, A, D)
            if T > 1:  # lag by lag, particle by particle (Helfand: short lags are far below the long ones)
                rel = np.abs(got[64][1][1:] - got[0][1][1:]) / np.abs(got[0][1][1:])
                assert rel.max() < 1e-9, (T, A, D, rel.max())
        ctx.set_option("short_max", 64)
        ctx.set_option("short_lags_max", 64)
        ts_s, _ = ctx.vacf_fft(by_particle=False)
        assert names() == ["k_short"] and scale_rel_err(ts_s, want_ts) < TOL
        ctx.set_option("short_lags_max", 48)
        v = orc.synthetic_velocities(65, 30, D, seed=65)
        run_vacf(ctx, v, False, True)
        assert "k_short" not in names()
    finally:
        ctx.set_option("short_max", 64)
        ctx.set_option("short_lags_max", 48)
        ctx.set_option("timeline", 0)


@pytest.mark.parametrize("D", [1, 2, 3])
@pytest.mark.parametrize("T", [65, 80, 96, 127, 128, 129, 200, 255, 256, 257, 300, 400, 511, 512])
def test_mid_length_kernel(ctx, T, D):
    """65 ... 512 frames: k_mid (mid_kernels.hpp: a lane per column and pair of 16-lag blocks, a sliding window in registers)
    for the windowed VACF and the Einstein-Helfand sums, with and without the by-particle array, wherever it can run
    ("mid_all" 1; by default: the windowed VACF from 97 frames, Helfand from 97 to 128) — against the oracle
    (velocityautocorr.py:217-238, viscosity.py:201-233), against the kernels it replaces ("mid_max" 0), particle by
    particle and lag by lag; frame counts on both sides of the 16-lag blocks and of the three launch shapes (64 / 32 / 16
    columns per tile), particle counts on both sides of a tile; same bits every launch; 513 frames take the old kernels."""
    from oracle import numpy_oracle as orc

    def names():
        return [n for n, _ in ctx.kernel_timeline() if not n.startswith(("k_sum", "end", "memset", "k_bp_transpose", "k_helfand_product"))]

    ctx.set_option("timeline", 1)
    try:
        for A in (1, 5, 6, 10, 11, 21, 22, 64, 150):
            v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=9000 + 13 * T + A)
            x = x + 50.0
            scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
            want_bp, want_ts = orc.vacf_windowed(v)
            want_hb, want_hs = orc.helfand(v, x, m, vol, 300.0)
            got = {}
            for mid in (512, 0):
                ctx.set_option("mid_max", mid)
                ctx.set_option("mid_all", 1)
                hs, hb = run_helfand(ctx, v, x, m, scale, True)
                if mid:
                    assert names() == ["k_mid"], names()
                    again = ctx.helfand_msd(m, scale, by_particle=True)
                    assert np.array_equal(again[0], hs) and np.array_equal(again[1], hb)
                else:
                    assert "k_mid" not in names()
                hs_l, _ = ctx.helfand_msd(m, scale, by_particle=False)
                ts_d, bp_d = ctx.vacf_direct(by_particle=True)
                ts_dl, _ = ctx.vacf_direct(by_particle=False)
                assert ("k_mid" in names()) == bool(mid)
                got[mid] = (hs, hb, hs_l, ts_d, bp_d, ts_dl)
                assert hb.shape == (T, A) and not hb[0].any()
                for a, w in ((hb, want_hb), (hs, want_hs), (hs_l, want_hs), (bp_d, want_bp), (ts_d, want_ts), (ts_dl, want_ts)):
                    assert scale_rel_err(a, w) < TOL, (T, A, D, mid)
            for a, b in zip(got[512], got[0]):
                assert scale_rel_err(a, b) < 1e-12, (T, A, D)
            rel = np.abs(got[512][1][1:] - got[0][1][1:]) / np.abs(got[0][1][1:])  # Helfand, lag by lag, particle by particle
            assert rel.max() < 1e-9, (T, A, D, rel.max())
        # the default choice
        ctx.set_option("mid_max", 512)
        ctx.set_option("mid_all", 0)
        ctx.vacf_direct(by_particle=True)
        assert ("k_mid" in names()) == (T >= 97), names()
        ctx.helfand_msd(m, scale, by_particle=True)
        assert ("k_mid" in names()) == (97 <= T <= 128), names()
        v = orc.synthetic_velocities(513, 30, D, seed=65)
        run_vacf(ctx, v, False, True)
        assert "k_mid" not in names()
    finally:
        ctx.set_option("mid_max", 512)
        ctx.set_option("mid_all", 0)
        ctx.set_option("timeline", 0)
