"""The oracle, pinned against vectors produced by the reference itself
(tests/golden/make_golden.py) and the constants printed in its tests/notebooks."""
import json
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_almost_equal, assert_approx_equal
from scipy import integrate

from oracle import numpy_oracle as orc
from conftest import GOLDEN, REPO, scale_rel_err


def g(name):
    return np.load(os.path.join(GOLDEN, name))


def step(nstep, start=0, stop=None, step_=1, cols=(0, 1, 2)):
    t = np.arange(nstep, dtype=np.float64)[start:stop:step_]
    v = np.repeat(t[:, None, None], 3, axis=2)[:, :, list(cols)]
    x = np.repeat((t * t / 2)[:, None, None], 3, axis=2)[:, :, list(cols)]
    return v, x


CONST = json.load(open(os.path.join(GOLDEN, "reference_constants.json")))
DIMS = [("xyz", 3), ("xy", 2), ("xz", 2), ("yz", 2), ("x", 1), ("y", 1), ("z", 1)]


@pytest.mark.parametrize("key", ["foo", "bar", "yx", "zyx"])
def test_dim_type_error(key):
    with pytest.raises(ValueError, match=f"invalid dim_type: {key}"):
        orc.parse_dim_type(key)


def test_dim_type_table():
    assert orc.parse_dim_type("xz") == ([0, 2], 2)
    assert orc.parse_dim_type("xyz") == ([0, 1, 2], 3)
    assert orc.parse_dim_type("y") == ([1], 1)


def test_tidynamics_pad_rule():
    # N=10->16, 1000->1024, 1024->2048, 5001->8192, 1e4->16384
    for n, want in ((10, 16), (1000, 1024), (1024, 2048), (5001, 8192), (10**4, 16384)):
        assert orc.tidynamics_n_fft(n) == want


def test_notebook_n10_vectors():
    v, _ = step(10)
    _, ts_w = orc.vacf_windowed(v)
    _, ts_f = orc.vacf_fft(v)
    assert_allclose(ts_w, CONST["notebook_poly_step_N10"], rtol=0, atol=1e-12)
    # printed FFT line: identical except the roundoff residue at the last lag
    assert_allclose(ts_f, CONST["notebook_fft_step_N10"], rtol=0, atol=2e-12)
    assert_allclose(ts_w, g("ref_vacf_windowed_step_N10.npy"), rtol=0, atol=0)
    assert_allclose(ts_f, g("ref_vacf_fft_step_N10.npy"), rtol=0, atol=0)
    assert_allclose(g("kat_vacf_poly_N10_D3.npy"), CONST["notebook_poly_step_N10"])


@pytest.mark.parametrize("dim,d", DIMS)
def test_windowed_vs_reference_poly_sliced(dim, d):
    # test_velocityautocorr.py:342-360 (start=10, stop=1000, step=10 -> 99 frames)
    cols, _ = orc.parse_dim_type(dim)
    v, _ = step(5001, 10, 1000, 10, cols)
    assert v.shape[0] == 99
    _, ts = orc.vacf_windowed(v)
    assert_almost_equal(ts, g(f"kat_vacf_poly_10_1000_10_D{d}.npy"), decimal=4)
    _, tsf = orc.vacf_fft(v)
    assert_almost_equal(tsf, g(f"kat_vacf_poly_10_1000_10_D{d}.npy"), decimal=3)


@pytest.mark.parametrize("d", [1, 2, 3])
def test_windowed_sliced_equals_reference_method(d):
    cols = {1: [0], 2: [0, 2], 3: [0, 1, 2]}[d]
    v, _ = step(5001, 10, 1000, 10, cols)
    _, ts = orc.vacf_windowed(v)
    assert_allclose(ts, g(f"ref_vacf_windowed_step_10_1000_10_D{d}.npy"), rtol=0, atol=0)


@pytest.mark.parametrize("d", [1, 2, 3])
def test_fft_vs_reference_poly_full(d):
    # test_velocityautocorr.py:96-123 and :454-469 (N=5001, decimal=4 / 3)
    v, _ = step(5001, cols=range(d))
    _, ts = orc.vacf_fft(v)
    poly = g(f"kat_vacf_poly_N5001_D{d}.npy")
    assert_almost_equal(ts, poly, decimal=4)
    assert scale_rel_err(ts, poly) < 1e-11
    # Green-Kubo constants quoted in the reference's tests (:378)
    gk = integrate.trapezoid(ts, np.arange(5001.0)) / d
    assert_approx_equal(gk, CONST["gk_trapezoid_step_N5001"], significant=8)
    gk_odd = integrate.simpson(y=ts, x=np.arange(5001.0)) / d
    assert_approx_equal(gk_odd, CONST["gk_simpson_step_N5001"], significant=8)


@pytest.mark.parametrize("tag", ["T7_A1_D1", "T64_A5_D2", "T200_A33_D3"])
def test_random_vs_reference_methods(tag):
    v = g(f"rand_vel_{tag}.npy")
    bp, ts = orc.vacf_windowed(v)
    assert_allclose(bp, g(f"ref_vacf_windowed_bp_{tag}.npy"), rtol=0, atol=0)
    assert_allclose(ts, g(f"ref_vacf_windowed_ts_{tag}.npy"), rtol=0, atol=0)
    bpf, tsf = orc.vacf_fft(v)
    assert_allclose(bpf, g(f"ref_vacf_fft_bp_{tag}.npy"), rtol=0, atol=0)
    assert_allclose(tsf, g(f"ref_vacf_fft_ts_{tag}.npy"), rtol=0, atol=0)
    # the reference asserts FFT == windowed to decimal=4 (:305-315)
    assert scale_rel_err(bpf, bp) < 1e-12
    bpb, tsb = orc.vacf_fft_batched(v, atom_block=4)
    assert scale_rel_err(bpb, bpf) < 1e-13 and scale_rel_err(tsb, tsf) < 1e-13


@pytest.mark.parametrize("dim,d", DIMS)
def test_helfand_vs_reference_poly_sliced(dim, d):
    # test_viscosity.py:191-208, default rtol=1e-7
    cols, _ = orc.parse_dim_type(dim)
    v, x = step(5001, 10, 1000, 10, cols)
    _, ts = orc.helfand(v, x, [16.0], np.full(99, 8.0), 300.0)
    assert_allclose(ts, g(f"kat_helfand_poly_10_1000_10_D{d}.npy"))


def test_helfand_full_kat_subset():
    # full N=5001 KAT (test_viscosity.py:180-189) at D=1 (the slab loop is O(T^2))
    v, x = step(5001, cols=[0])
    _, ts = orc.helfand(v, x, [16.0], np.full(5001, 8.0), 300.0)
    assert_allclose(ts, g("kat_helfand_poly_N5001_D1.npy"))
    assert ts[0] == 0.0


def test_helfand_n10_and_notebook():
    v, x = step(10)
    _, ts = orc.helfand(v, x, [16.0], np.full(10, 8.0), 300.0)
    assert_allclose(ts, g("ref_helfand_step_N10.npy"), rtol=0, atol=0)
    # stale notebook vector is the sum over dims = 3x the current mean
    want = np.array(CONST["helfand_notebook_step_N10_sum_over_dims"]) / 3.0
    assert_allclose(ts[:3], want, rtol=1e-11)


@pytest.mark.parametrize("tag", ["T9_A1_D1", "T50_A6_D2", "T120_A17_D3"])
def test_helfand_random_vs_reference_method(tag):
    z = np.load(os.path.join(GOLDEN, f"rand_helfand_in_{tag}.npz"))
    bp, ts = orc.helfand(z["v"], z["x"], z["m"], z["vol"], 313.0)
    assert_allclose(bp, g(f"ref_helfand_bp_{tag}.npy"), rtol=0, atol=0)
    assert_allclose(ts, g(f"ref_helfand_ts_{tag}.npy"), rtol=0, atol=0)
    T = len(ts)
    slope = orc.helfand_fit(ts, (2, T - 2))
    assert_allclose(slope, g(f"ref_helfand_visc_{tag}.npy")[0], rtol=1e-13)


# ---- the plain-C port against the same reference vectors ------------------
@pytest.mark.parametrize("tag", ["T7_A1_D1", "T64_A5_D2", "T200_A33_D3"])
def test_c_oracle_vacf(tag):
    from oracle import c_oracle

    v = g(f"rand_vel_{tag}.npy")
    bp, ts = c_oracle.vacf_windowed(v)
    assert scale_rel_err(bp, g(f"ref_vacf_windowed_bp_{tag}.npy")) < 1e-14
    assert scale_rel_err(ts, g(f"ref_vacf_windowed_ts_{tag}.npy")) < 1e-14
    bp, ts = c_oracle.vacf_fft(v, n_threads=2)
    assert scale_rel_err(bp, g(f"ref_vacf_fft_bp_{tag}.npy")) < 1e-13
    assert scale_rel_err(ts, g(f"ref_vacf_fft_ts_{tag}.npy")) < 1e-13
    # throughput variant used by bench.py's all-cores CPU line: same numbers, lag sums only
    lagsum = c_oracle.vacf_fft_lagsum(v, n_threads=3)
    assert scale_rel_err(lagsum / v.shape[1], g(f"ref_vacf_fft_ts_{tag}.npy")) < 1e-13


@pytest.mark.parametrize("tag", ["T9_A1_D1", "T50_A6_D2", "T120_A17_D3"])
def test_c_oracle_helfand(tag):
    from oracle import c_oracle

    z = np.load(os.path.join(GOLDEN, f"rand_helfand_in_{tag}.npz"))
    bp, ts = c_oracle.helfand(z["v"], z["x"], z["m"], z["vol"], 313.0)
    assert_allclose(bp, g(f"ref_helfand_bp_{tag}.npy"), rtol=1e-13, atol=0)
    assert_allclose(ts, g(f"ref_helfand_ts_{tag}.npy"), rtol=1e-13, atol=0)


def test_c_oracle_step_kat():
    from oracle import c_oracle

    v, _ = step(5001)
    _, ts = c_oracle.vacf_fft(v)
    assert_almost_equal(ts, g("kat_vacf_poly_N5001_D3.npy"), decimal=3)
    _, ts = c_oracle.vacf_windowed(v)
    assert_almost_equal(ts, g("kat_vacf_poly_N5001_D3.npy"), decimal=4)


def test_parallel_numpy_oracle_matches_one_core():
    """oracle/parallel.py (bench.py's all-cores CPU line): blocks of atoms in child processes add
    up to the one-process result on the same synthetic tensor."""
    from oracle import numpy_oracle as orc
    from oracle import parallel, synth

    T, A, D = 300, 11, 3
    lag, seconds, n = parallel.vacf_fft_all_cores(7, T, A * D + 5, A, D, n_workers=3)
    v = synth.synthetic_block(7, T, A * D + 5, 0, A * D).reshape(T, A, D)
    bp, _ = orc.vacf_fft(v)
    assert n == 3 and seconds > 0
    np.testing.assert_allclose(lag, bp.sum(axis=1), rtol=0, atol=1e-12 * float(np.max(np.abs(bp))) * A)
    assert parallel.usable_cpus() >= 1


def test_golden_files_reproduce_from_the_reference():
    """tests/golden/make_golden.py --check: a fresh run of the reference's own code (build
    container only: /root/reference does not exist on the GPU box) reproduces every committed
    array bit for bit and every constant -- the fixtures and their generator do not drift."""
    if not os.path.isdir("/root/reference/transport_analysis"):
        pytest.skip("the reference is only present in the build container")
    import subprocess
    import sys

    res = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), "--check"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]


def test_c_oracle_kats_under_sanitizers():
    """ASan/UBSan on the CPU C restatement.  `make -C oracle/c asan` builds
    ta_oracle.c + kat_main.c with -fsanitize=address,undefined and runs the known-answer cases
    (step-trajectory closed form, FFT == windowed on ragged shapes, Helfand against a long-double
    loop); any sanitizer report or mismatch fails the make."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    cdir = os.path.join(REPO, "oracle", "c")
    r = subprocess.run(["make", "-s", "-C", cdir, "-B", "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle KATs under ASan/UBSan: ok" in r.stdout
