"""API parity with the reference's own tests
(/root/reference/transport_analysis/tests/test_velocityautocorr.py,
test_viscosity.py), on in-memory trajectories.  Every test runs twice: on the CPU
with an oracle-backed context (host logic only) and, marked gpu, on the real HIP
path through the C-ABI."""
import json
import os

import matplotlib

matplotlib.use("Agg")
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_almost_equal, assert_approx_equal
from scipy import integrate

from conftest import GOLDEN, scale_rel_err
from transport_analysis_amd import VelocityAutocorr as VACF
from transport_analysis_amd import ViscosityHelfand as VH
from transport_analysis_amd import _base, _lib
from transport_analysis_amd._mini_mda import ArrayUniverse

NSTEP = 5001
DIMS = [("xyz", 3), ("xy", 2), ("xz", 2), ("yz", 2), ("x", 1), ("y", 1), ("z", 1)]
CONST = json.load(open(os.path.join(GOLDEN, "reference_constants.json")))


def g(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(params=["oracle-backed", "cpu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    if request.param == "oracle-backed":
        from fake_backend import OracleContext, OracleGroup

        monkeypatch.setattr(_lib, "Context", OracleContext)
        monkeypatch.setattr(_lib, "Group", OracleGroup)
    elif request.param == "cpu":
        # the library's own opt-in CPU backend (csrc/cpu_backend.cpp) through the real C-ABI: what a user asks for with
        # device="cpu" / $TA_AMD_DEVICE=cpu
        monkeypatch.setenv("TA_AMD_DEVICE", "cpu")
    else:
        assert _lib.device_count() >= 1
    return request.param


@pytest.fixture(scope="module")
def step_vtraj():
    # test_velocityautocorr.py:46-57 / test_viscosity.py:56-86
    t = np.arange(NSTEP, dtype=np.float64)
    v = np.repeat(t[:, None, None], 3, axis=2)
    x = np.repeat((t * t / 2)[:, None, None], 3, axis=2)
    return ArrayUniverse(positions=x, velocities=v, masses=[16.0],
                         dimensions=[2, 2, 2, 90, 90, 90])


@pytest.fixture(scope="module")
def water():
    rng = np.random.default_rng(5)
    v = rng.standard_normal((10, 30, 3)).astype(np.float32)
    x = (10 + 0.01 * np.cumsum(v, axis=0)).astype(np.float32)
    return ArrayUniverse(positions=x, velocities=v, masses=np.full(30, 15.999),
                         dimensions=[20, 20, 20, 90, 90, 90], dt=0.5)


# -------------------------------------------------------------- construction
def test_ag_accepted(water):
    VACF(water.atoms[:10], fft=False)
    VH(water.atoms[:10])


def test_no_velocities(backend):
    u = ArrayUniverse(n_atoms=10, n_frames=5)
    with pytest.raises(_base.NoDataError, match="VACF computation requires velocities"):
        VACF(u.atoms, fft=False).run()
    with pytest.raises(_base.NoDataError, match="Helfand viscosity computation requires"):
        VH(u.atoms).run()


def test_zero_volume_is_no_data(backend, water):
    u = ArrayUniverse(positions=water.trajectory._pos, velocities=water.trajectory._vel)
    with pytest.raises(_base.NoDataError, match="Helfand viscosity computation requires"):
        VH(u.atoms).run()


def test_updating_ag_rejected(water):
    class Updating(_base.UpdatingAtomGroup):
        universe = water

        def __len__(self):
            return 3

    with pytest.raises(TypeError, match="UpdatingAtomGroups are not valid"):
        VACF(Updating(), fft=False)
    with pytest.raises(TypeError, match="UpdatingAtomGroups are not valid"):
        VH(Updating())


@pytest.mark.parametrize("dimtype", ["foo", "bar", "yx", "zyx"])
def test_dimtype_error(water, dimtype):
    with pytest.raises(ValueError, match=f"invalid dim_type: {dimtype}"):
        VACF(water.atoms, dim_type=dimtype)
    with pytest.raises(ValueError, match=f"invalid dim_type: {dimtype}"):
        VH(water.atoms, dim_type=dimtype)


def test_dimtype_is_lowercased(water):
    assert VACF(water.atoms, dim_type="XZ")._dim == [0, 2]


def test_must_run_first(step_vtraj):
    v = VACF(step_vtraj.atoms, fft=False)
    for call in (v.plot_vacf, v.self_diffusivity_gk, v.self_diffusivity_gk_odd,
                 v.plot_running_integral):
        with pytest.raises(RuntimeError, match="Analysis must be run"):
            call()
    assert "timeseries" not in v.results


# -------------------------------------------------------------------- VACF
def test_fft_vs_simple_and_attributes(backend, water):
    ag = water.atoms[::3]
    a = VACF(ag, fft=False).run()
    b = VACF(ag, fft=True).run()
    assert a.n_frames == 10 and a.n_particles == 10 and a.dim_fac == 3
    assert a.results.vacf_by_particle.shape == (10, 10)
    assert a.results.timeseries.dtype == np.float64
    assert_allclose(a.times, 0.5 * np.arange(10))
    assert_almost_equal(a.results.timeseries, b.results.timeseries, decimal=4)
    assert_almost_equal(a.results.vacf_by_particle, b.results.vacf_by_particle, decimal=4)
    assert_allclose(a.results.timeseries, a.results.vacf_by_particle.mean(axis=1), rtol=1e-12)
    c = VACF(ag, fft=True, by_particle=False).run()
    assert c.results.vacf_by_particle is None
    assert_allclose(c.results.timeseries, b.results.timeseries, rtol=1e-10, atol=1e-12)


def test_second_run_restages(backend, water):
    v = VACF(water.atoms, fft=True)
    first = v.run().results.timeseries.copy()
    second = v.run(start=2).results.timeseries
    assert len(first) == 10 and len(second) == 8
    assert_allclose(v.run().results.timeseries, first, rtol=1e-12)


@pytest.mark.parametrize("tdim,tdim_factor", DIMS)
@pytest.mark.parametrize("fft", [False, True])
def test_step_vtraj_all_dims(backend, step_vtraj, tdim, tdim_factor, fft):
    v = VACF(step_vtraj.atoms, dim_type=tdim, fft=fft).run()
    poly = g(f"kat_vacf_poly_N5001_D{tdim_factor}.npy")
    assert_almost_equal(v.results.timeseries, poly, decimal=3 if fft else 4)
    # Green-Kubo numbers quoted in the reference's tests (:378): 8 significant figures
    assert_approx_equal(v.self_diffusivity_gk(), CONST["gk_trapezoid_step_N5001"], significant=8)
    assert_approx_equal(v.self_diffusivity_gk_odd(), CONST["gk_simpson_step_N5001"], significant=8)
    sl = slice(10, 1000, 10)
    want = integrate.simpson(y=poly[sl], x=np.arange(NSTEP)[sl]) / tdim_factor
    assert_approx_equal(v.self_diffusivity_gk(start=10, stop=1000, step=10), want, significant=6)
    want = integrate.trapezoid(poly[sl], np.arange(NSTEP)[sl]) / tdim_factor
    assert_approx_equal(v.self_diffusivity_gk_odd(start=10, stop=1000, step=10), want, significant=6)


@pytest.mark.parametrize("tdim,tdim_factor", DIMS)
@pytest.mark.parametrize("fft", [False, True])
def test_start_stop_step_all_dims(backend, step_vtraj, tdim, tdim_factor, fft):
    v = VACF(step_vtraj.atoms, dim_type=tdim, fft=fft).run(start=10, stop=1000, step=10)
    assert v.n_frames == 99
    assert_allclose(v.frames, np.arange(10, 1000, 10))
    poly = g(f"kat_vacf_poly_10_1000_10_D{tdim_factor}.npy")
    assert_almost_equal(v.results.timeseries, poly, decimal=3 if fft else 4)


def test_plots(backend, water):
    v = VACF(water.atoms, fft=False).run()
    (line,) = v.plot_vacf()
    x, y = line.get_xydata().T
    assert_allclose(x, v.times)
    assert_allclose(y, v.results.timeseries)
    assert line.axes.get_xlabel() == "Time (ps)"
    assert line.axes.get_ylabel() == "Velocity Autocorrelation Function (Å^2 / ps^2)"
    (line,) = v.plot_vacf(start=1, stop=9, step=2, xlabel="a", ylabel="b")
    x, y = line.get_xydata().T
    assert_allclose(x, v.times[1:9:2])
    assert_allclose(y, v.results.timeseries[1:9:2])
    assert (line.axes.get_xlabel(), line.axes.get_ylabel()) == ("a", "b")
    (line,) = v.plot_running_integral()
    x, y = line.get_xydata().T
    want = np.zeros(v.n_frames)
    for i in range(1, v.n_frames):
        want[i] = integrate.trapezoid(v.results.timeseries[: i + 1], v.times[: i + 1]) / v.dim_fac
    assert_allclose(x, v.times)
    assert_allclose(y, want)
    assert line.axes.get_ylabel() == "Running Integral of the VACF (Å^2 / ps)"
    (line,) = v.plot_running_integral(start=1, stop=9, step=2)
    assert len(line.get_xydata()) == 4


# ----------------------------------------------------------------- Helfand
@pytest.mark.parametrize("tdim,tdim_factor", DIMS)
def test_helfand_step_all_dims(backend, step_vtraj, tdim, tdim_factor):
    # test_viscosity.py:180-208, assert_allclose default rtol=1e-7
    vh = VH(step_vtraj.atoms, dim_type=tdim).run(start=10, stop=1000, step=10)
    assert_allclose(vh.results.timeseries, g(f"kat_helfand_poly_10_1000_10_D{tdim_factor}.npy"))
    assert vh.results.timeseries[0] == 0.0
    assert vh.boltzmann == CONST["boltzmann_kJ_per_mol_K"]


def test_helfand_full_kat(backend, step_vtraj):
    vh = VH(step_vtraj.atoms, dim_type="x").run()
    assert_allclose(vh.results.timeseries, g("kat_helfand_poly_N5001_D1.npy"))


@pytest.mark.parametrize("tdim,tdim_factor", DIMS)
def test_lag_sums_only_on_the_reference_kats(backend, step_vtraj, tdim, tdim_factor):
    """by_particle=False (timeseries only) of both O(T^2) analyses on the reference's step
    trajectory: on the GPU these are the matrix-core kernels (windowed VACF: diagonal sums of the
    frames' Gram matrix; Helfand: products of rows centred on a nearby frame) — the same tolerances
    as the reference's own assertions (test_velocityautocorr.py:269-279, test_viscosity.py:180-208),
    and the Helfand one lag by lag since its short lags are what a careless expansion loses."""
    v = VACF(step_vtraj.atoms, dim_type=tdim, fft=False, by_particle=False).run()
    assert v.results.vacf_by_particle is None
    assert_almost_equal(v.results.timeseries, g(f"kat_vacf_poly_N5001_D{tdim_factor}.npy"), decimal=4)
    vh = VH(step_vtraj.atoms, dim_type=tdim, by_particle=False).run(start=10, stop=1000, step=10)
    assert vh.results.visc_by_particle is None and vh.results.timeseries[0] == 0.0
    assert_allclose(vh.results.timeseries, g(f"kat_helfand_poly_10_1000_10_D{tdim_factor}.npy"))
    vh = VH(step_vtraj.atoms, dim_type=tdim, by_particle=False).run()
    assert_allclose(vh.results.timeseries, g(f"kat_helfand_poly_N5001_D{tdim_factor}.npy"))


def test_helfand_float32_switch(backend, step_vtraj):
    """float32=True selects the library's float32 squared-difference path (configs[4])."""
    vh = VH(step_vtraj.atoms, dim_type="xy", float32=True).run(start=10, stop=1000, step=10)
    want = g("kat_helfand_poly_10_1000_10_D2.npy")
    # the float32 path's bar: 2e-6 of the series' scale (P is rounded once to float32: on this pure trend that rounding,
    # 6e-8 |P|, is what the shortest lags see; lag by lag they agree to a few 1e-6)
    assert np.max(np.abs(vh.results.timeseries - want)) < 2e-6 * np.max(np.abs(want))
    assert_allclose(vh.results.timeseries, want, rtol=5e-5)
    # per lag: only the shortest lags of this pure trend see P's float32 rounding (measured on the GPU path: 1e-5 at lag 1,
    # 2e-6 at lags 2-4, 3e-7 from lag 5 on; tools/f32_lag_accuracy.py) -- from lag 5 on the bound is that of the scale
    assert_allclose(vh.results.timeseries[5:], want[5:], rtol=2e-6)
    assert vh.results.timeseries[0] == 0.0
    if backend == "oracle-backed":
        assert vh._ctx.options["direct_f32"] == 1
    vh = VH(step_vtraj.atoms, dim_type="xy").run(start=10, stop=1000, step=10)
    assert_allclose(vh.results.timeseries, g("kat_helfand_poly_10_1000_10_D2.npy"))


def test_helfand_fft_switch(backend, step_vtraj):
    """fft=True (an extension: the reference has only the O(T^2) loop) selects the library's
    S1 - 2 S2 evaluation of the mean squared differences; it excludes float32."""
    vh = VH(step_vtraj.atoms, dim_type="xy", fft=True, by_particle=False).run(start=10, stop=1000, step=10)
    want = g("kat_helfand_poly_10_1000_10_D2.npy")
    assert np.max(np.abs(vh.results.timeseries - want)) < 1e-10 * np.max(np.abs(want))
    assert vh.results.timeseries[0] == 0.0 and vh.results.visc_by_particle is None
    if backend == "oracle-backed":
        assert vh._ctx.options["helfand_fft"] == 1
    vh = VH(step_vtraj.atoms, dim_type="xy", fft=True).run(start=10, stop=1000, step=10)
    assert vh.results.visc_by_particle.shape == (99, 1)
    assert np.max(np.abs(vh.results.visc_by_particle[:, 0] - want)) < 1e-10 * np.max(np.abs(want))
    with pytest.raises(ValueError):
        VH(step_vtraj.atoms, fft=True, by_particle=False, float32=True)
    vh = VH(step_vtraj.atoms, dim_type="xy", by_particle=False).run(start=10, stop=1000, step=10)
    if backend == "oracle-backed":
        assert vh._ctx.options["helfand_fft"] == 0


@pytest.mark.parametrize("tag", ["T50_A6_D2", "T120_A17_D3"])
def test_helfand_fit_and_volume(backend, tag):
    z = np.load(os.path.join(GOLDEN, f"rand_helfand_in_{tag}.npz"))
    T, A, D = z["v"].shape
    pad = lambda a: np.concatenate([a, np.zeros((T, A, 3 - D))], axis=2)  # noqa: E731
    side = float(np.cbrt(np.average(z["vol"])))
    u = ArrayUniverse(positions=pad(z["x"]), velocities=pad(z["v"]), masses=z["m"],
                      dimensions=[side, side, side, 90, 90, 90])
    # float32 trajectory data: compare against the oracle on the same rounded inputs
    from oracle import numpy_oracle as orc

    v32 = pad(z["v"]).astype(np.float32).astype(np.float64)[:, :, :D]
    x32 = pad(z["x"]).astype(np.float32).astype(np.float64)[:, :, :D]
    want_bp, want_ts = orc.helfand(v32, x32, z["m"], np.full(T, side**3), 313.0)
    vh = VH(u.atoms, temp_avg=313.0, dim_type="xyz"[:D], linear_fit_window=(2, T - 2)).run()
    assert_allclose(vh.results.timeseries, want_ts, rtol=1e-9)
    assert_allclose(vh.results.visc_by_particle, want_bp, rtol=1e-9, atol=1e-12 * want_bp.max())
    assert_allclose(vh.results.viscosity, orc.helfand_fit(want_ts, (2, T - 2)), rtol=1e-8)
    vh.plot_viscosity_function()


def test_mdanalysis_parallel_backend_declaration():
    """MDAnalysis >= 2.8: run(backend=...) is only honoured by classes that declare themselves
    parallelizable; these two stage all frames into one device slab, so they declare the
    serial backend only (their parallel axis is atoms: distributed=True)."""
    from transport_analysis_amd import VelocityAutocorr, ViscosityHelfand

    for cls in (VelocityAutocorr, ViscosityHelfand):
        assert cls.get_supported_backends() == ("serial",)
        assert cls._analysis_algorithm_is_parallelizable is False


@pytest.mark.parametrize("device", ["cpu", pytest.param(0, marks=pytest.mark.gpu)])
def test_ncbox_water_golden_vectors(device):
    """BASELINE configs[0] ("CPU path, plumbing, no GPU"): VelocityAutocorr(fft=True) on MDAnalysisTests' PRM_NCBOX/TRJ_NCBOX
    water box through the REAL MDAnalysis AnalysisBase -- on the library's opt-in CPU backend (device="cpu": runs
    wherever MDAnalysis is installed, GPU or not) and, marked gpu, on the HIP path.  Both vectors the reference
    prints: its module docstring (velocityautocorr.py:39-43, resname WAT and resid 1-5) and
    docs/tutorials/vacf_testing_examples.ipynb:52-55 (name O and resname WAT and resid 1-10);
    and its own FFT == windowed assertion (tests/test_velocityautocorr.py:297-315)."""
    mda = pytest.importorskip("MDAnalysis")
    pytest.importorskip("MDAnalysisTests")
    from MDAnalysisTests.datafiles import PRM_NCBOX, TRJ_NCBOX

    from transport_analysis_amd import VelocityAutocorr

    import json

    const = json.load(open(os.path.join(GOLDEN, "reference_constants.json")))
    u = mda.Universe(PRM_NCBOX, TRJ_NCBOX)
    for sel, key in (("resname WAT and resid 1-5", "ncbox_vacf_fft_WAT_resid_1_5"),
                     ("name O and resname WAT and resid 1-10", "ncbox_vacf_fft_O_resid_1_10")):
        ag = u.select_atoms(sel)
        fft = VelocityAutocorr(ag, fft=True, device=device).run()
        np.testing.assert_allclose(fft.results.timeseries, const[key], rtol=1e-7)
        win = VelocityAutocorr(ag, fft=False, device=device).run()
        np.testing.assert_almost_equal(fft.results.timeseries, win.results.timeseries, decimal=4)
        np.testing.assert_almost_equal(fft.results.vacf_by_particle, win.results.vacf_by_particle, decimal=4)


@pytest.mark.parametrize("dim_type", ["x", "y", "z", "xy", "xz", "yz", "xyz"])
@pytest.mark.parametrize("src_dtype,dst_dtype", [(np.float32, np.float32), (np.float32, np.float64),
                                                 (np.float64, np.float64)])
def test_stage_columns_equals_the_reference_fill(dim_type, src_dtype, dst_dtype):
    """The per-frame slab fill (velocityautocorr.py:192-194, viscosity.py:189-199:
    slab[i] = atomgroup.velocities[:, dim]) done by slices: same values for every dim_type,
    for an atom sub-range (distributed runs) and across the float32 -> float64 upcast."""
    from transport_analysis_amd._base import parse_dim_type, stage_columns

    dim, fac = parse_dim_type(dim_type)
    rng = np.random.default_rng(4)
    src = rng.standard_normal((37, 3)).astype(src_dtype)
    for lo, hi in ((0, 37), (5, 29), (36, 37)):
        dst = np.full((40, fac), -7.0, dtype=dst_dtype)
        stage_columns(dst, src, lo, hi, dim)
        assert np.array_equal(dst[: hi - lo], src[lo:hi][:, dim].astype(dst_dtype))
        assert np.all(dst[hi - lo:] == -7.0)


# ---- several GPUs from one process: devices=[...] -------------------------
def test_group_partition_matches_the_per_process_one():
    """The C-ABI's split (ta_group_shard: floor(A i / n)) is transport_analysis_amd.dist.atom_shard:
    contiguous, covering, and the same whether the GPUs sit behind one process or one each."""
    from fake_backend import OracleGroup
    from transport_analysis_amd.dist import atom_shard

    for A in (1, 2, 7, 64, 100000):
        for n in (1, 2, 3, 8):
            g = OracleGroup(list(range(n)))
            edges = [g.shard(A, i) for i in range(n)]
            assert edges == [atom_shard(A, i, n) for i in range(n)]
            assert edges[0][0] == 0 and edges[-1][1] == A
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("fft", [True, False])
def test_devices_one_frame_loop_column_ranges(backend, water, devices, fft):
    if backend == "cpu":
        pytest.skip("device groups are GPU contexts (ta_group_*: TA_E_UNSUPPORTED on the CPU backend)")
    """devices=[...]: ONE pass over the trajectory fills every member's column block, the result is
    the single-context one -- timeseries over ALL atoms, ONE (n_frames, n_particles) by-particle
    array whose column ranges the members fill.  On the GPU box the members share the one GPU
    (the copy-and-add reduce); devices=[0] must equal the plain context bit for bit."""
    ag = water.atoms
    ref = VACF(ag, fft=fft).run()
    calls = {"n": 0}
    orig = type(ag).velocities.fget

    def counting(self):
        calls["n"] += 1
        return orig(self)

    type(ag).velocities = property(counting)
    try:
        a = VACF(ag, fft=fft, devices=devices).run()
    finally:
        type(ag).velocities = property(orig)
    # one gather per frame (+ the dtype probe of _prepare), however many devices
    assert calls["n"] <= ag.universe.trajectory.n_frames + 1
    assert a.results.vacf_by_particle.shape == ref.results.vacf_by_particle.shape
    n = len(devices)
    assert a.results.device_ranges == [(30 * i // n, 30 * (i + 1) // n) for i in range(n)]
    if len(devices) == 1:
        assert np.array_equal(a.results.timeseries, ref.results.timeseries)
        assert np.array_equal(a.results.vacf_by_particle, ref.results.vacf_by_particle)
    else:
        assert scale_rel_err(a.results.timeseries, ref.results.timeseries) < 1e-12
        assert scale_rel_err(a.results.vacf_by_particle, ref.results.vacf_by_particle) < 1e-12
    h_ref = VH(ag, linear_fit_window=(2, 8)).run()
    h = VH(ag, linear_fit_window=(2, 8), devices=devices).run()
    assert scale_rel_err(h.results.timeseries, h_ref.results.timeseries) < 1e-12
    assert scale_rel_err(h.results.visc_by_particle, h_ref.results.visc_by_particle) < 1e-12
    np.testing.assert_allclose(h.results.viscosity, h_ref.results.viscosity, rtol=1e-9)


def test_devices_more_gpus_than_atoms_and_exclusive_with_distributed(backend):
    if backend == "cpu":
        pytest.skip("device groups are GPU contexts")
    rng = np.random.default_rng(11)
    v = rng.standard_normal((12, 2, 3)).astype(np.float32)
    u = ArrayUniverse(velocities=v, positions=np.cumsum(v, axis=0), masses=[1.0, 2.0],
                      dimensions=[5, 5, 5, 90, 90, 90])
    ref = VACF(u.atoms, fft=True).run()
    a = VACF(u.atoms, fft=True, devices=[0, 0, 0]).run()  # member 0 holds no atom
    assert a.results.device_ranges == [(0, 0), (0, 1), (1, 2)]
    assert scale_rel_err(a.results.timeseries, ref.results.timeseries) < 1e-12
    assert scale_rel_err(a.results.vacf_by_particle, ref.results.vacf_by_particle) < 1e-12
    with pytest.raises(ValueError, match="exclusive"):
        VACF(u.atoms, devices=[0], distributed=True)
    with pytest.raises(ValueError, match="exclusive"):
        VH(u.atoms, devices=[0], distributed=True)

