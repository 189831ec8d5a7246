"""The library's opt-in CPU backend (transport_analysis_amd/csrc/cpu_backend.cpp) through the real C-ABI, on the
GPU-less container: the golden / known-answer parity set of the GPU path (reference-generated fixtures, the
reference's step polynomials, the notebook vector), ragged shapes against the oracle, and that it is OPT-IN:
nothing reaches it unless the caller names it.  Tolerance: 1e-10 of the series' scale."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, scale_rel_err
from transport_analysis_amd import _lib

TOL = 1e-10


def g(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context("cpu")
    yield c
    c.close()


def run_vacf(ctx, v, fft, by_particle, dtype=np.float64):
    T, A, D = v.shape
    (slab,) = ctx.stage_alloc(T, A, D, n_slabs=1, dtype=dtype)
    slab[...] = v
    ctx.stage_commit(0, T)
    return (ctx.vacf_fft if fft else ctx.vacf_direct)(by_particle=by_particle)


def run_helfand(ctx, v, x, m, scale, by_particle, dtype=np.float64):
    T, A, D = v.shape
    sv, sx = ctx.stage_alloc(T, A, D, n_slabs=2, dtype=dtype)
    sv[...] = v
    sx[...] = x
    ctx.stage_commit(0, T)
    return ctx.helfand_msd(m, scale, by_particle=by_particle)


def step(nstep, start=0, stop=None, step_=1, cols=(0, 1, 2)):
    t = np.arange(nstep, dtype=np.float64)[start:stop:step_]
    v = np.repeat(t[:, None, None], 3, axis=2)[:, :, list(cols)]
    x = np.repeat((t * t / 2)[:, None, None], 3, axis=2)[:, :, list(cols)]
    return v, x


def test_cpu_backend_is_opt_in():
    """Without a GPU the default path fails loudly; the CPU backend runs only for a caller who names it."""
    from transport_analysis_amd import VelocityAutocorr
    from transport_analysis_amd._mini_mda import ArrayUniverse

    u = ArrayUniverse(velocities=np.ones((4, 2, 3)), positions=np.ones((4, 2, 3)))
    if _lib.device_count() == 0:
        with pytest.raises(_lib.TAError, match="no usable HIP device"):
            _lib.Context(0)
        os.environ.pop("TA_AMD_DEVICE", None)
        with pytest.raises(_lib.TAError, match="no usable HIP device"):
            VelocityAutocorr(u.atoms).run()
    a = VelocityAutocorr(u.atoms, device="cpu").run()
    assert a._ctx.is_cpu and a.results.timeseries.shape == (4,)
    np.testing.assert_allclose(a.results.timeseries, [3.0, 3.0, 3.0, 3.0], rtol=1e-14)
    assert _lib.device_index("cpu") == _lib.DEVICE_CPU == -1 and _lib.device_index("2") == 2


def test_device_facing_calls_are_unsupported(ctx):
    ctx.stage_alloc(4, 2, 3)
    for call in (lambda: ctx.stage_alloc_device(4, 2, 3), lambda: ctx.stage_device(0), lambda: ctx.timing_history(),
                 lambda: ctx.vacf_fft_staged(0), lambda: ctx.kernel_timeline(),
                 lambda: ctx.vacf_fft_dev(1, 4, 2, 3, 6, 1)):
        with pytest.raises(_lib.TAError) as ei:
            call()
        assert ei.value.code == -5, ei.value  # TA_E_UNSUPPORTED
    with pytest.raises(_lib.TAError):
        _lib.Group([-1])
    ctx.stage_free()
    with pytest.raises(_lib.TAError, match="not been staged"):
        ctx.vacf_fft()


@pytest.mark.parametrize("fft", [True, False])
@pytest.mark.parametrize("tag", ["T7_A1_D1", "T64_A5_D2", "T200_A33_D3"])
def test_vacf_golden_random(ctx, tag, fft):
    v = g(f"rand_vel_{tag}.npy")
    kind = "fft" if fft else "windowed"
    want_bp, want_ts = g(f"ref_vacf_{kind}_bp_{tag}.npy"), g(f"ref_vacf_{kind}_ts_{tag}.npy")
    ts, bp = run_vacf(ctx, v, fft, True)
    assert bp.shape == want_bp.shape and ts.shape == want_ts.shape
    assert scale_rel_err(bp, want_bp) < TOL and scale_rel_err(ts, want_ts) < TOL
    ts2, bp2 = run_vacf(ctx, v, fft, False)
    assert bp2 is None and np.array_equal(ts, ts2)


@pytest.mark.parametrize("fft", [True, False])
@pytest.mark.parametrize("d", [1, 2, 3])
def test_vacf_step_kat_full_and_sliced(ctx, d, fft):
    # test_velocityautocorr.py:331-340 / :454-469 (N = 5001) and :342-360 / :471-483 (start 10, stop 1000, step 10)
    v, _ = step(5001, cols=range(d))
    poly = g(f"kat_vacf_poly_N5001_D{d}.npy")
    ts, _ = run_vacf(ctx, v, fft, False)
    np.testing.assert_almost_equal(ts, poly, decimal=3 if fft else 4)
    assert scale_rel_err(ts, poly) < TOL
    cols = {1: [1], 2: [0, 2], 3: [0, 1, 2]}[d]
    v, _ = step(5001, 10, 1000, 10, cols)
    poly = g(f"kat_vacf_poly_10_1000_10_D{d}.npy")
    ts, bp = run_vacf(ctx, v, fft, True)
    assert scale_rel_err(ts, poly) < TOL and scale_rel_err(bp[:, 0], poly) < TOL


def test_vacf_n10_notebook(ctx):
    const = json.load(open(os.path.join(GOLDEN, "reference_constants.json")))
    v, _ = step(10)
    for fft in (True, False):
        ts, _ = run_vacf(ctx, v, fft, False)
        np.testing.assert_allclose(ts, const["notebook_poly_step_N10"], rtol=0, atol=1e-10)


@pytest.mark.parametrize("T,A,D", [(1, 1, 1), (2, 3, 3), (5, 2, 2), (16, 7, 3), (17, 9, 1), (33, 17, 2), (100, 40, 3),
                                   (129, 3, 3), (257, 33, 3), (1000, 37, 3), (1025, 6, 1), (2049, 3, 3), (4097, 2, 3)])
def test_vacf_fft_vs_oracle_shapes(ctx, T, A, D):
    """power-of-two pads on both sides of a length, odd column counts (the unpaired column), atom counts that do not
    fill the last block of eight or the last pair of a block; float32 and float64 staging give the same bits"""
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=1000 + T).astype(np.float32).astype(np.float64)
    want_bp, want_ts = orc.vacf_fft_batched(v)
    ts, bp = run_vacf(ctx, v, True, True)
    assert scale_rel_err(bp, want_bp) < TOL and scale_rel_err(ts, want_ts) < TOL
    ts32, bp32 = run_vacf(ctx, v, True, True, dtype=np.float32)
    assert np.array_equal(ts, ts32) and np.array_equal(bp, bp32)


@pytest.mark.parametrize("T,A,D", [(1, 2, 3), (3, 1, 1), (8, 5, 3), (9, 4, 2), (31, 6, 3), (500, 9, 3), (1000, 33, 3)])
def test_vacf_direct_vs_oracle_shapes(ctx, T, A, D):
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(T, A, D, seed=2000 + T)
    want_bp, want_ts = orc.vacf_windowed(v)
    ts, bp = run_vacf(ctx, v, False, True)
    assert scale_rel_err(bp, want_bp) < TOL and scale_rel_err(ts, want_ts) < TOL


@pytest.mark.parametrize("tag", ["T9_A1_D1", "T50_A6_D2", "T120_A17_D3"])
def test_helfand_golden_random(ctx, tag):
    from transport_analysis_amd._base import BOLTZMANN

    z = np.load(os.path.join(GOLDEN, f"rand_helfand_in_{tag}.npz"))
    scale = 1.0 / (2 * BOLTZMANN * np.average(z["vol"]) * 313.0)
    ts, bp = run_helfand(ctx, z["v"], z["x"], z["m"], scale, True)
    want_bp, want_ts = g(f"ref_helfand_bp_{tag}.npy"), g(f"ref_helfand_ts_{tag}.npy")
    assert ts[0] == 0.0 and np.all(bp[0] == 0.0)
    assert scale_rel_err(bp, want_bp) < TOL and scale_rel_err(ts, want_ts) < TOL
    np.testing.assert_allclose(ts, want_ts, rtol=1e-7)  # the reference's own bar


@pytest.mark.parametrize("d", [1, 2, 3])
def test_helfand_step_kat(ctx, d):
    # test_viscosity.py:180-208 (assert_allclose rtol=1e-7), N = 5001 and sliced
    from transport_analysis_amd._base import BOLTZMANN

    scale = 1.0 / (2 * BOLTZMANN * 8.0 * 300.0)
    v, x = step(5001, cols=range(d))
    ts, _ = run_helfand(ctx, v, x, np.array([16.0]), scale, False)
    np.testing.assert_allclose(ts, g(f"kat_helfand_poly_N5001_D{d}.npy"), rtol=1e-7)
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
    assert scale_rel_err(bp, want_bp) < TOL and scale_rel_err(ts, want_ts) < TOL


def test_results_do_not_depend_on_the_thread_count(ctx):
    from oracle import numpy_oracle as orc

    v = orc.synthetic_velocities(700, 45, 3, seed=77)
    got = []
    for n in (1, 3, 0):  # 0: OpenMP's default team
        ctx.set_option("cpu_threads", n)
        got.append(run_vacf(ctx, v, True, True) + run_vacf(ctx, v, False, True))
    for other in got[1:]:
        for a, b in zip(got[0], other):
            assert np.array_equal(a, b)


def test_stage_synth_is_the_oracles_generator(ctx):
    """ta_stage_synth on a CPU context fills the host slab with the benchmark tensor: bit for bit oracle/synth.py (and,
    tests/test_gpu_parity.py, the GPU's)"""
    from oracle import synth

    T, A, D = 37, 5, 3
    for dtype in (np.float64, np.float32):
        (slab,) = ctx.stage_alloc(T, A, D, dtype=dtype)
        ctx.stage_synth(0, 20250827, 6, 40)
        want = synth.synthetic_block(20250827, T, 40, 6, 6 + A * D).reshape(T, A, D)
        assert np.array_equal(slab, want.astype(dtype))
